"""ctypes binding of oracle/agatha_oracle.c.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product path (agatha_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libagatha_oracle.so")
_REFLIB = os.path.join(_HERE, "_ref", "libagatha_refshim.so")


class Params(C.Structure):
    """Field order of gasal_subst_scores (reference AGAThA/src/gasal.h:165-173)."""
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap_open", C.c_int32),
                ("gap_extend", C.c_int32), ("slice_width", C.c_int32), ("z_threshold", C.c_int32),
                ("band_width", C.c_int32)]


def make_params(m=2, x=4, q=4, r=2, s=3, z=400, w=751):
    return Params(m, x, q, r, s, z, w)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("agatha_oracle.c", "agatha_lanes_model.c", "ksw_style_avx2.c", "seq_ops_ref.c", "Makefile")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.agatha_oracle_batch.argtypes = [C.c_void_p] * 6 + [C.c_int, C.POINTER(Params), C.c_int, C.c_int,
                                                                C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.agatha_oracle_batch.restype = None
        _lib.agatha_oracle_pack.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.agatha_nominal_cells.argtypes = [C.c_int, C.c_int, C.c_int]
        _lib.agatha_nominal_cells.restype = C.c_int64
        _lib.agatha_lanes_batch.argtypes = [C.c_void_p] * 6 + [C.c_int, C.POINTER(Params), C.c_int, C.c_int, C.c_int,
                                                               C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        _lib.agatha_lanes_batch.restype = None
        _lib.agatha_lanes16_batch.argtypes = [C.c_void_p] * 6 + [C.c_int, C.POINTER(Params), C.c_int, C.c_int, C.c_int] + \
                                                [C.c_void_p] * 5
        _lib.agatha_lanes16_batch.restype = None
        _lib.ksw_style_batch.argtypes = [C.c_void_p] * 6 + [C.c_int, C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p,
                                                            C.c_void_p, C.POINTER(C.c_int)]
        _lib.ksw_style_batch.restype = None
        _lib.ksw_style_batch_stops.argtypes = [C.c_void_p] * 6 + [C.c_int, C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p,
                                                                  C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        _lib.ksw_style_batch_stops.restype = None
        _lib.agatha_cells_upto_diag.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.agatha_cells_upto_diag.restype = C.c_int64
        _lib.agatha_traceback_batch.argtypes = [C.c_void_p] * 6 + [C.c_int, C.POINTER(Params), C.c_int] + [C.c_void_p] * 6
        _lib.agatha_traceback_batch.restype = None
        for f in (_lib.agatha_ref_seq_ops_as_written, _lib.agatha_seq_ops_product_semantics):
            f.argtypes = [C.c_void_p] * 4 + [C.c_int]
            f.restype = None
    return _lib


from agatha_amd.workload import make_batch  # noqa: E402  (wire-format helper lives in the product package)


MODEL_SLICES, MODEL_STEPS, MODEL_EXACTBAND = 0, 1, 2


def seq_ops(packed, lens, offsets, ops, as_written):
    """Per-sequence reverse / complement of one side of a packed batch (oracle/seq_ops_ref.c): as_written=True restates the
    reference's gasal_reversecomplement_kernel (pack_rc_seqs.h:56-212) exactly as its text behaves, False the semantics the
    product implements (reverse exactly len bases, padding stays behind).  Returns a new array."""
    out = np.ascontiguousarray(packed, np.uint32).copy()
    lens, offsets = np.ascontiguousarray(lens, np.uint32), np.ascontiguousarray(offsets, np.uint32)
    ops = np.ascontiguousarray(ops, np.uint8)
    f = lib().agatha_ref_seq_ops_as_written if as_written else lib().agatha_seq_ops_product_semantics
    f(out.ctypes.data, lens.ctypes.data, offsets.ctypes.data, ops.ctypes.data, len(lens))
    return out


def align_batch(qbuf, tbuf, qoff, toff, qlen, tlen, params, wide=False, model=MODEL_SLICES, threads=1):
    n = len(qlen)
    qbuf = np.ascontiguousarray(qbuf, np.uint8)
    tbuf = np.ascontiguousarray(tbuf, np.uint8)
    arrs = [np.ascontiguousarray(a, np.uint32) for a in (qoff, toff, qlen, tlen)]
    out = np.zeros((3, n), np.int32)
    lib().agatha_oracle_batch(qbuf.ctypes.data, tbuf.ctypes.data, *[a.ctypes.data for a in arrs], n,
                              C.byref(params), int(wide), int(model), int(threads),
                              out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data)
    return out[0], out[1], out[2]


def lanes_batch(qbuf, tbuf, qoff, toff, qlen, tlen, params, G, S, threads=1):
    """CPU emulation of the HIP kernel's lane/slot schedule (oracle/agatha_lanes_model.c)."""
    n = len(qlen)
    qbuf = np.ascontiguousarray(qbuf, np.uint8)
    tbuf = np.ascontiguousarray(tbuf, np.uint8)
    arrs = [np.ascontiguousarray(a, np.uint32) for a in (qoff, toff, qlen, tlen)]
    out = np.zeros((3, n), np.int32)
    rc = C.c_int(0)
    lib().agatha_lanes_batch(qbuf.ctypes.data, tbuf.ctypes.data, *[a.ctypes.data for a in arrs], n,
                             C.byref(params), int(G), int(S), int(threads),
                             out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, C.byref(rc))
    if rc.value:
        raise ValueError("G*S too small for this band")
    return out[0], out[1], out[2]


def lanes16_batch(qbuf, tbuf, qoff, toff, qlen, tlen, params, G, S, threads=1, value_step_margin=0):
    """CPU emulation of the packed-int16 kernel's arithmetic (no band masks, cut constants, bail-out to int32).
    Returns (score, qend, tend, kind, stats): kind 0 = int16 path, 1 = fell back to the int32 model, -1 = window too
    small; stats = (min rep, max rep, largest out-of-band rep, smallest in-band rep)."""
    n = len(qlen)
    qbuf = np.ascontiguousarray(qbuf, np.uint8)
    tbuf = np.ascontiguousarray(tbuf, np.uint8)
    arrs = [np.ascontiguousarray(a, np.uint32) for a in (qoff, toff, qlen, tlen)]
    out = np.zeros((4, n), np.int32)
    st = np.zeros(4, np.int32)
    # value_step_margin > 0: emulate the kernel's value steps (all but a pair's last `margin` steps only track the values of the
    # anti-diagonal maxima); kind 2 = the pair had to be started over on key steps
    lib().agatha_lanes16_set_margin(int(value_step_margin))
    lib().agatha_lanes16_batch(qbuf.ctypes.data, tbuf.ctypes.data, *[a.ctypes.data for a in arrs], n,
                               C.byref(params), int(G), int(S), int(threads),
                               out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, out[3].ctypes.data,
                               st.ctypes.data)
    return out[0], out[1], out[2], out[3], st


def ksw_style_batch(qbuf, tbuf, qoff, toff, qlen, tlen, params, threads=1):
    """Anti-diagonal AVX2 CPU kernel (oracle/ksw_style_avx2.c), exact-band semantics; returns (score, qend, tend, n_fallback)
    where n_fallback counts pairs outside the int16 domain that went through the scalar exact-band model instead."""
    n = len(qlen)
    qbuf = np.ascontiguousarray(qbuf, np.uint8)
    tbuf = np.ascontiguousarray(tbuf, np.uint8)
    arrs = [np.ascontiguousarray(a, np.uint32) for a in (qoff, toff, qlen, tlen)]
    out = np.zeros((3, n), np.int32)
    fb = C.c_int(0)
    lib().ksw_style_batch(qbuf.ctypes.data, tbuf.ctypes.data, *[a.ctypes.data for a in arrs], n, C.byref(params),
                          int(threads), out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, C.byref(fb))
    return out[0], out[1], out[2], fb.value


def effective_cells_batch(qbuf, tbuf, qoff, toff, qlen, tlen, params, threads=1):
    """SURVEY.md 8(d) "effective cells": the in-band cells (exact band) on the cell anti-diagonals up to the one each pair's
    z-drop walk ended on (agatha_kernel.h:297-309,319-322 end the reference's loop there), from the AVX2 port (scalar exact-band
    model outside its int16 domain).  Returns (per-pair effective cells int64[n], per-pair stop anti-diagonal, (score, qend,
    tend), n_fallback); a pair that never stops has stop = Q + R - 2 and effective = nominal cells."""
    n = len(qlen)
    qbuf = np.ascontiguousarray(qbuf, np.uint8)
    tbuf = np.ascontiguousarray(tbuf, np.uint8)
    arrs = [np.ascontiguousarray(a, np.uint32) for a in (qoff, toff, qlen, tlen)]
    out = np.zeros((4, n), np.int32)
    fb = C.c_int(0)
    lib().ksw_style_batch_stops(qbuf.ctypes.data, tbuf.ctypes.data, *[a.ctypes.data for a in arrs], n, C.byref(params),
                                int(threads), out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, out[3].ctypes.data,
                                C.byref(fb))
    w = int(params.band_width)
    f = lib().agatha_cells_upto_diag
    eff = np.fromiter((f(int(q), int(r), w, int(d)) for q, r, d in zip(arrs[2], arrs[3], out[3])), np.int64, n)
    return eff, out[3], (out[0], out[1], out[2]), fb.value


def align_pairs(queries, targets, params, **kw):
    qb, qo, ql = make_batch(queries)
    tb, to, tl = make_batch(targets)
    return align_batch(qb, tb, qo, to, ql, tl, params, **kw)


def start_positions(queries, targets, params, qend, tend, threads=1):
    """Start positions (the reference declares query_batch_start / target_batch_start, gasal.h:89-90, and leaves them NULL,
    res.cpp:27-28; GASAL2's WITH_START idea, gasal.h:36): the SAME banded extension run backwards from the end cell --
    on the reversed prefixes q[0..qend], t[0..tend], z-drop off -- ends in the cell where the best-scoring alignment that
    ends in (qend, tend) begins.  Returns (query_start, target_start, backward_score)."""
    rq = [bytes(q[:int(e) + 1])[::-1] for q, e in zip(queries, qend)]
    rt = [bytes(t[:int(e) + 1])[::-1] for t, e in zip(targets, tend)]
    back = Params(params.match, params.mismatch, params.gap_open, params.gap_extend, params.slice_width, -1, params.band_width)
    s, q, t = align_pairs(rq, rt, back, wide=True, model=MODEL_SLICES, threads=threads)
    return np.asarray(qend, np.int32) - q, np.asarray(tend, np.int32) - t, s


def traceback_batch(qbuf, tbuf, qoff, toff, qlen, tlen, params, threads=1):
    """Alignment paths (agatha_model_traceback: the cigar / n_cigar_ops members the reference declares, gasal.h:91-92, and
    never fills).  Returns (score, query_end, target_end, cigar, n_ops): pair k's bytes are
    cigar[qoff[k] + toff[k] : ... + n_ops[k]], each (count << 2) | op, op 0 match / 1 mismatch / 2 D / 3 I."""
    n = len(qlen)
    qbuf, tbuf = np.ascontiguousarray(qbuf, np.uint8), np.ascontiguousarray(tbuf, np.uint8)
    qoff, toff, qlen, tlen = (np.ascontiguousarray(a, np.uint32) for a in (qoff, toff, qlen, tlen))
    score, qend, tend, nops = (np.zeros(n, np.int32) for _ in range(4))
    off = qoff.astype(np.uint64) + toff.astype(np.uint64)
    cigar = np.zeros(int(qbuf.size + tbuf.size) + 16, np.uint8)
    lib().agatha_traceback_batch(qbuf.ctypes.data, tbuf.ctypes.data, qoff.ctypes.data, toff.ctypes.data, qlen.ctypes.data,
                                 tlen.ctypes.data, n, C.byref(params), threads, score.ctypes.data, qend.ctypes.data,
                                 tend.ctypes.data, cigar.ctypes.data, off.ctypes.data, nops.ctypes.data)
    return score, qend, tend, cigar, nops


def traceback_pairs(queries, targets, params, threads=1):
    """-> (score, query_end, target_end, [cigar bytes of each pair])"""
    qb, qo, ql = make_batch(queries)
    tb, to, tl = make_batch(targets)
    s, qe, te, cig, nops = traceback_batch(qb, tb, qo, to, ql, tl, params, threads)
    off = qo.astype(np.int64) + to.astype(np.int64)
    return s, qe, te, [cig[o:o + k].tobytes() if k >= 0 else None for o, k in zip(off, nops)]    # None: no path


def cigar_rescore(cigar, query, target, params):
    """Independent check of a cigar: walks both sequences from the origin and returns (score, query bases used, target
    bases used) under the affine scoring of the recurrence (a gap of length L costs gap_open + L * gap_extend; a base
    against N, or N against anything, scores -1: gasal_kernels.h:48-50).  Raises if an op byte contradicts the bases."""
    i = j = 0
    score = 0
    prev = -1
    for byte in cigar:
        op, cnt = byte & 3, byte >> 2
        assert cnt >= 1
        if op in (0, 1):
            for _ in range(cnt):
                a, b = query[i] & 15, (target[j] & 15 if j < len(target) else 14)     # target_end can lie in the N padding
                same = a == b and a != 14
                assert same == (op == 0), (i, j, op)
                score += -1 if (a == 14 or b == 14) else (params.match if a == b else -params.mismatch)
                i += 1; j += 1
        else:
            score -= cnt * params.gap_extend + (0 if prev == op else params.gap_open)
            if op == 2: j += cnt
            else: i += cnt
        prev = op
    return score, i, j


def pack(unpacked):
    unpacked = np.ascontiguousarray(unpacked, np.uint8)
    assert unpacked.size % 8 == 0
    out = np.zeros(unpacked.size // 8, np.uint32)
    lib().agatha_oracle_pack(unpacked.ctypes.data, unpacked.size, out.ctypes.data)
    return out


def nominal_cells(Q, R, w):
    return int(lib().agatha_nominal_cells(int(Q), int(R), int(w)))


def nominal_cells_np(qlen, tlen, w):
    """Vectorised nominal in-band cell count: sum_i (min(R-1,i+w)-max(0,i-w)+1)."""
    tot = 0
    for Q, R in zip(np.asarray(qlen, np.int64), np.asarray(tlen, np.int64)):
        i = np.arange(Q, dtype=np.int64)
        hi = np.minimum(R - 1, i + w)
        lo = np.maximum(0, i - w)
        tot += int(np.clip(hi - lo + 1, 0, None).sum())
    return tot


# ---- the reference kernel itself under the CPU warp emulator (only where oracle/_ref was built) ----
_ref = None


def have_ref():
    return os.path.exists(_REFLIB)


def ref_align_batch(qbuf, tbuf, qoff, toff, qlen, tlen, params, blocks=0, threads_per_block=256):
    """Run the unmodified reference agatha_kernel on CPU (oracle/ref_shim).  Only available in the
    build container after `make -C oracle ref`."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REFLIB)
        _ref.refshim_align.argtypes = [C.c_void_p] * 6 + [C.c_int, C.POINTER(Params), C.c_int, C.c_int,
                                                          C.c_void_p, C.c_void_p, C.c_void_p]
        _ref.refshim_align.restype = C.c_int
    n = len(qlen)
    qbuf = np.ascontiguousarray(qbuf, np.uint8)
    tbuf = np.ascontiguousarray(tbuf, np.uint8)
    arrs = [np.ascontiguousarray(a, np.uint32) for a in (qoff, toff, qlen, tlen)]
    out = np.zeros((3, n), np.int32)
    rc = _ref.refshim_align(qbuf.ctypes.data, tbuf.ctypes.data, *[a.ctypes.data for a in arrs], n,
                            C.byref(params), int(blocks), int(threads_per_block),
                            out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data)
    if rc != 0:
        raise RuntimeError("reference shim reported deadlock/error rc=%d" % rc)
    return out[0], out[1], out[2]
