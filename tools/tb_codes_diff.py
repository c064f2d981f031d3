"""Developer tool (GPU box): the cell codes of ONE pair as the int16 traceback kernel and as the int32 one record them, cell by
cell (the cells the walk may look at: rows of the query, inside the band).  Usage: python3 tools/tb_codes_diff.py seed n lo hi k w z"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import agatha_amd                                    # noqa: E402
from agatha_amd.engine import _DevBuf, _chk          # noqa: E402
from oracle import oracle as O, synth                # noqa: E402

seed, n, lo, hi, k, w, z = (int(x) for x in sys.argv[1:8])
rng = np.random.default_rng(seed)
qs, ts = [], []
for kk in range(n):
    ln = int(rng.integers(lo, hi))
    q = synth.random_seq(rng, ln)
    t = synth.mutate(rng, q, 0.05, 0.04, 0.04)
    if kk % 7 == 0:
        q = q.copy(); q[rng.integers(0, ln)] = ord('N')
    qs.append(bytes(q)); ts.append(bytes(t))
qs, ts = [qs[k]], [ts[k]]
Q, R = len(qs[0]), len(ts[0])
print("pair", k, "Q", Q, "R", R)
eng = agatha_amd.Engine(0)
lib = eng.lib
sc = agatha_amd.Scores.make(w=w, z=z)
qb, qo, ql = O.make_batch(qs)
tb, to, tl = O.make_batch(ts)
b = eng.batch(qb, tb, qo, to, ql, tl)
b.upload(); b.pack(); eng.synchronize()
nbytes = lib.agatha_amd_traceback_scratch_bytes(1, b.max_qlen, b.max_tlen, C.byref(sc), 0)
m = b.d_meta
out = []
for no16 in (0, 1):
    agatha_amd.set_debug_option("no_int16", no16)
    scratch = _DevBuf(lib, nbytes)
    cig = _DevBuf(lib, b.qbytes + b.tbytes + 16)
    nops = _DevBuf(lib, 4)
    _chk(lib, lib.agatha_amd_align_traceback(eng.stream, b.d_pk_q.ptr, b.d_pk_t.ptr, m[2].ptr, m[3].ptr, m[0].ptr, m[1].ptr, 1,
                                             b.max_qlen, b.max_tlen, C.byref(sc), b.d_res[0].ptr, b.d_res[1].ptr, b.d_res[2].ptr,
                                             cig.ptr, nops.ptr, b.d_ws.ptr, b.ws_bytes, scratch.ptr, nbytes))
    h = np.zeros(nbytes // 4, np.uint32)
    _chk(lib, lib.agatha_amd_memcpy_d2h_async(eng.stream, h.ctypes.data, scratch.ptr, h.nbytes))
    b.download(); eng.synchronize()
    print("no_int16", no16, "result", [int(b.res_host[j][0]) for j in range(3)], "kinds", b.pair_kinds())
    out.append(h)
agatha_amd.set_debug_option("no_int16", 0)
head = (8 + 255) // 256 * 256 + (4 + 255) // 256 * 256 + 256
W = (w + 7) // 8
pql, prl = (Q + 7) // 8, (R + 7) // 8
GS = 96 if W + 1 <= 96 else 192
a16, a32 = out[0][head // 4:], out[1][head // 4:]
bad = 0
for i in range(pql + prl - 1):
    for r in range(max(0, i - pql + 1), min(prl - 1, i) + 1):
        q = i - r
        if q < max(0, r - W) or q > min(pql - 1, r + W):
            continue
        base = (i * GS + r % GS) * 8
        for il in range(min(8, Q - 8 * q)):
            x, y = int(a16[base + il]), int(a32[base + il])
            if x == y:
                continue
            for jl in range(8):
                ci, cj = 8 * q + il, 8 * r + jl
                if abs(ci - cj) > w:
                    continue
                cx, cy = (x >> (4 * jl)) & 15, (y >> (4 * jl)) & 15
                if cx != cy:
                    bad += 1
                    if bad <= 40:
                        print(f"cell ({ci},{cj}) block ({q},{r}) step {i}: int16 {cx} int32 {cy}")
print("differing cells", bad)
