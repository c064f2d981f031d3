"""GPU: alignment paths (cigar / n_cigar_ops of gasal.h:91-92) through the C-ABI against the oracle's
traceback, plus the size-independent check: a path re-scored from the sequences alone gives the reported score and uses
exactly query_end + 1 / target_end + 1 bases."""
import numpy as np
import pytest

from oracle import oracle as O, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import agatha_amd
    e = agatha_amd.Engine(0)
    yield e
    e.close()


def _pairs(seed, n, lo, hi, sub=0.05, ins=0.04, dele=0.04, n_every=7):
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for k in range(n):
        ln = int(rng.integers(lo, hi))
        q = synth.random_seq(rng, ln)
        t = synth.mutate(rng, q, sub, ins, dele)
        if n_every and k % n_every == 0:
            q = q.copy()
            q[rng.integers(0, ln)] = ord('N')
        qs.append(bytes(q))
        ts.append(bytes(t))
    return qs, ts


def _traceback(eng, qs, ts, scratch_bytes=None, stats=False, **p):
    """stats: also (the int16 kernel's step statistics of the call, the pair kinds it left)"""
    import agatha_amd
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload()
        b.pack()
        got = b.align_traceback(agatha_amd.Scores.make(**p), scratch_bytes=scratch_bytes)
        return (got, b.step_stats(), eng.last_int16_config()) if stats else got
    finally:
        b.free()


def _check(qs, ts, got, threads=8, **p):
    params = O.make_params(**p)
    s, qe, te, cig = O.traceback_pairs(qs, ts, params, threads=threads)
    assert (got[0] == s).all() and (got[1] == qe).all() and (got[2] == te).all()
    for k, (a, b) in enumerate(zip(got[3], cig)):
        assert a == b, (k, p, None if a is None else a[:8], None if b is None else b[:8])
    paths = 0
    for k, c in enumerate(got[3]):
        if c is None or s[k] <= 0:
            continue
        paths += 1
        assert O.cigar_rescore(c, qs[k], ts[k], params) == (s[k], qe[k] + 1, te[k] + 1)
    return paths


@pytest.mark.parametrize("w,z,lo,hi,n", [(751, 400, 1, 3000, 48), (16, 400, 1, 300, 300), (3, -1, 1, 60, 400),
                                         (100, 50, 1, 800, 200), (0, 400, 1, 40, 60), (40, -1, 1, 500, 120),
                                         (200, 400, 500, 4000, 40), (1500, 400, 3000, 9000, 6)])
def test_paths_match_oracle(eng, w, z, lo, hi, n):
    qs, ts = _pairs(1000 + w, n, lo, hi)
    got = _traceback(eng, qs, ts, w=w, z=z)
    assert _check(qs, ts, got, w=w, z=z) > n // 2


def test_paths_in_several_passes(eng):
    """A scratch area that holds 5 pairs of the longest kind: the batch goes through in passes and nothing changes."""
    import agatha_amd
    qs, ts = _pairs(77, 64, 200, 1500)
    p = dict(w=64, z=400)
    lib = eng.lib
    import ctypes as C
    small = lib.agatha_amd_traceback_scratch_bytes(64, max(map(len, qs)), max(map(len, ts)), C.byref(agatha_amd.Scores.make(**p)), 5)
    whole = _traceback(eng, qs, ts, **p)
    parts = _traceback(eng, qs, ts, scratch_bytes=small, **p)
    assert all((a == b).all() for a, b in zip(whole[:3], parts[:3])) and whole[3] == parts[3]
    _check(qs, ts, parts, **p)


def test_broken_pairs_and_other_scorings(eng):
    """z-dropped pairs (unrelated tails), empty alignments, other letters, a scoring outside the byte profile."""
    rng = np.random.default_rng(5)
    qs, ts = _pairs(6, 80, 50, 1200)
    for k in range(0, 80, 3):                                   # unrelated second half: z-drop ends the extension early
        t = bytearray(ts[k]); h = len(t) // 2
        t[h:] = bytes(synth.random_seq(rng, len(t) - h)); ts[k] = bytes(t)
    qs[1] = b"ACGT" * 30; ts[1] = b"TGCA" * 30                    # no positive cell: empty alignment
    qs[2] = b"ACGTRYACGTAC" * 20; ts[2] = b"ACGTRYACGTAC" * 20   # letters outside ACGTN
    for p in (dict(m=2, x=4, q=4, r=2, z=100, w=200), dict(m=1, x=4, q=6, r=2, z=400, w=64),
              dict(m=200, x=300, q=400, r=100, z=20000, w=100)):
        got = _traceback(eng, qs, ts, **p)
        _check(qs, ts, got, **p)
    assert got[3][1] in (b"", None) or got[0][1] > 0


def test_length_hints_that_are_too_small_cost_results_not_memory(eng):
    """Code areas are sized from the TRUE lengths, the number of passes from the hints: with a pair much longer than the hints
    the passes run out and the pairs left over get AGATHA_AMD_BAD_RESULT / AGATHA_AMD_NO_PATH -- every other pair is exact,
    nothing is written outside the scratch."""
    import ctypes as C
    import agatha_amd
    from agatha_amd.engine import _DevBuf, _chk
    qs, ts = _pairs(91, 24, 300, 600, n_every=0)
    long_q = bytes(synth.random_seq(np.random.default_rng(3), 12000))
    qs[7], ts[7] = long_q, long_q
    p = dict(w=64, z=400)
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    lib = eng.lib
    sc = agatha_amd.Scores.make(**p)
    try:
        b.upload(); b.pack()
        hint = 640                                              # true for every pair but number 7
        nbytes = lib.agatha_amd_traceback_scratch_bytes(24, hint, hint, C.byref(sc), 0)
        guard = 1 << 20                                         # canary behind the scratch
        scratch, cig, nops = _DevBuf(lib, nbytes + guard), _DevBuf(lib, b.qbytes + b.tbytes + 16), _DevBuf(lib, 4 * 24)
        canary = np.full(guard, 0xA5, np.uint8)
        _chk(lib, lib.agatha_amd_memcpy_h2d_async(eng.stream, scratch.ptr + nbytes, canary.ctypes.data, guard))
        m = b.d_meta
        _chk(lib, lib.agatha_amd_align_traceback(eng.stream, b.d_pk_q.ptr, b.d_pk_t.ptr, m[2].ptr, m[3].ptr, m[0].ptr, m[1].ptr, 24,
                                                 hint, hint, C.byref(sc), b.d_res[0].ptr, b.d_res[1].ptr, b.d_res[2].ptr, cig.ptr,
                                                 nops.ptr, b.d_ws.ptr, b.ws_bytes, scratch.ptr, nbytes))
        h_n = np.zeros(24, np.uint32)
        h_c = np.zeros(b.qbytes + b.tbytes, np.uint8)
        back = np.zeros(guard, np.uint8)
        _chk(lib, lib.agatha_amd_memcpy_d2h_async(eng.stream, h_n.ctypes.data, nops.ptr, 96))
        _chk(lib, lib.agatha_amd_memcpy_d2h_async(eng.stream, h_c.ctypes.data, cig.ptr, h_c.nbytes))
        _chk(lib, lib.agatha_amd_memcpy_d2h_async(eng.stream, back.ctypes.data, scratch.ptr + nbytes, guard))
        b.download()
        eng.synchronize()
        res = b.res_host.copy()
        for d in (scratch, cig, nops):
            d.free()
    finally:
        b.free()
    assert (back == 0xA5).all()
    s, qe, te, cigs = O.traceback_pairs(qs, ts, O.make_params(**p), threads=4)
    off = qo.astype(np.int64) + to.astype(np.int64)
    refused = 0
    for k in range(24):
        if res[0][k] == np.iinfo(np.int32).min:
            assert h_n[k] == 0xFFFFFFFF
            refused += 1
        else:
            assert (res[0][k], res[1][k], res[2][k]) == (s[k], qe[k], te[k])
            assert h_c[off[k]:off[k] + h_n[k]].tobytes() == cigs[k]
    assert 1 <= refused <= 20


def test_mixed_lengths_share_the_code_area(eng):
    """One 6 kb pair among many short ones: sized by the longest pair the area would hold 4 pairs per pass; sized by the true
    lengths everything fits one pass (checked through the plan the device leaves at the head of the scratch)."""
    import ctypes as C
    import agatha_amd
    from agatha_amd.engine import _DevBuf, _chk
    qs, ts = _pairs(17, 60, 100, 400)
    big = synth.random_seq(np.random.default_rng(9), 6000)
    qs[30], ts[30] = bytes(big), bytes(synth.mutate(np.random.default_rng(10), big, 0.03, 0.03, 0.03))
    p = dict(w=100, z=400)
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    lib = eng.lib
    sc = agatha_amd.Scores.make(**p)
    try:
        b.upload(); b.pack()
        nbytes = lib.agatha_amd_traceback_scratch_bytes(60, b.max_qlen, b.max_tlen, C.byref(sc), 4)
        got = b.align_traceback(sc, scratch_bytes=nbytes)           # (allocates its own scratch of that size)
        scratch, cig, nops = _DevBuf(lib, nbytes), _DevBuf(lib, b.qbytes + b.tbytes + 16), _DevBuf(lib, 4 * 60)
        m = b.d_meta
        _chk(lib, lib.agatha_amd_align_traceback(eng.stream, b.d_pk_q.ptr, b.d_pk_t.ptr, m[2].ptr, m[3].ptr, m[0].ptr, m[1].ptr, 60,
                                                 b.max_qlen, b.max_tlen, C.byref(sc), b.d_res[0].ptr, b.d_res[1].ptr, b.d_res[2].ptr,
                                                 cig.ptr, nops.ptr, b.d_ws.ptr, b.ws_bytes, scratch.ptr, nbytes))
        plan = np.zeros(1, np.int32)
        head = ((8 * 60 + 255) // 256) * 256 + ((4 * 60 + 255) // 256) * 256
        _chk(lib, lib.agatha_amd_memcpy_d2h_async(eng.stream, plan.ctypes.data, scratch.ptr + head, 4))
        eng.synchronize()
        for d in (scratch, cig, nops):
            d.free()
    finally:
        b.free()
    assert plan[0] == 1
    _check(qs, ts, got, **p)


def test_paths_match_the_golden_fixture(eng):
    """The committed vectors of tests/golden/traceback_paths.json through the C-ABI."""
    import json
    import os
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "traceback_paths.json")))
    for case in doc["cases"]:
        qs = [q.encode() for q in case["queries"]]
        ts = [t.encode() for t in case["targets"]]
        got = _traceback(eng, qs, ts, **case["params"])
        assert [int(v) for v in got[0]] == case["score"] and [int(v) for v in got[1]] == case["query_end"]
        assert [int(v) for v in got[2]] == case["target_end"]
        assert [None if c is None else c.hex() for c in got[3]] == case["bytes"]


def test_single_pair_and_tiny_batches(eng):
    for n in (1, 2, 3):
        qs, ts = _pairs(200 + n, n, 1, 50)
        got = _traceback(eng, qs, ts, w=8, z=-1)
        _check(qs, ts, got, w=8, z=-1)


@pytest.mark.parametrize("w,z,lo,hi,n,shape", [(751, 400, 2000, 9000, 40, (16, 6)), (400, 400, 1000, 5000, 40, (16, 6)),
                                                 (1000, -1, 3000, 9000, 12, (32, 6)), (1528, 400, 4000, 9000, 8, (32, 6))])
def test_the_int16_kernel_records_the_codes(eng, w, z, lo, hi, n, shape):
    """Bands of 49..192 blocks with scores the packed kernel takes: the pass runs on align16_kernel<.., true> (one pair with an N in
    its query per seven and whatever it abandons go to the int32 kernel of the same pass, which writes the same layout); every
    path byte against the oracle, and the same bytes with the int16 kernel switched off."""
    import agatha_amd
    qs, ts = _pairs(4000 + w, n, lo, hi)
    rng = np.random.default_rng(w)
    for k in range(1, n, 5):                                     # unrelated tails: z-drop (or a long gap) inside the band
        t = bytearray(ts[k]); h = len(t) * 2 // 3
        t[h:] = bytes(synth.random_seq(rng, len(t) - h)); ts[k] = bytes(t)
    got, st, cfg = _traceback(eng, qs, ts, stats=True, w=w, z=z)
    # (round 6: the pass runs value steps like the score-only kernel; a pair a value step cannot decide starts over and writes the same codes again)
    assert cfg == shape and st[3] >= n - (n + 6) // 7 - 2 and st[0] > st[1] > 0
    assert _check(qs, ts, got, threads=16, w=w, z=z) > n // 2
    # ... and with key steps only (the pass of rounds 4-5): the same bytes
    with agatha_amd.debug_options(tb_value_steps=0):
        gotk, stk, _ = _traceback(eng, qs, ts, stats=True, w=w, z=z)
    assert stk[0] == 0 and stk[1] > 0
    assert all((a == b).all() for a, b in zip(got[:3], gotk[:3])) and got[3] == gotk[3]
    agatha_amd.set_debug_option("no_int16", 1)
    try:
        ref, st32, _ = _traceback(eng, qs, ts, stats=True, w=w, z=z)
    finally:
        agatha_amd.set_debug_option("no_int16", 0)
    assert st32[3] == 0
    assert all((a == b).all() for a, b in zip(got[:3], ref[:3])) and got[3] == ref[3]


def test_int16_codes_in_several_passes(eng):
    """The scratch holds three pairs of the longest kind: the int16 kernel of every pass takes the pairs of that pass only."""
    import ctypes as C
    import agatha_amd
    qs, ts = _pairs(78, 40, 1500, 4000)
    p = dict(w=751, z=400)
    small = eng.lib.agatha_amd_traceback_scratch_bytes(40, max(map(len, qs)), max(map(len, ts)), C.byref(agatha_amd.Scores.make(**p)), 3)
    whole = _traceback(eng, qs, ts, **p)
    parts, st, cfg = _traceback(eng, qs, ts, scratch_bytes=small, stats=True, **p)
    assert cfg == (16, 6) and st[3] >= 30
    assert all((a == b).all() for a, b in zip(whole[:3], parts[:3])) and whole[3] == parts[3]
    _check(qs, ts, parts, **p)
