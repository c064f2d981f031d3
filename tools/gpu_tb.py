"""Traceback pass on a slice of the headline workload: wall time of agatha_amd_align_traceback (scratch allocated up front)
next to plain agatha_amd_align.  Usage: python3 tools/gpu_tb.py [n_pairs]   (run under rocprofv3 --kernel-trace --stats for
the split between the recording kernel and the walk)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import agatha_amd                                    # noqa: E402
from agatha_amd import workload as W                 # noqa: E402
from agatha_amd.engine import _DevBuf, _chk          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
qs, ts = W.cfg_c1(n=n)
qb, qo, ql = W.make_batch(qs)
tb, to, tl = W.make_batch(ts)
eng = agatha_amd.Engine(0)
lib = eng.lib
sc = agatha_amd.Scores.make()
b = eng.batch(qb, tb, qo, to, ql, tl)
b.upload(); b.pack(); eng.synchronize()
cells = W.nominal_cells_total(ql, tl, 751)
per = lib.agatha_amd_traceback_pair_bytes(b.max_qlen, b.max_tlen, C.byref(sc))
nbytes = lib.agatha_amd_traceback_scratch_bytes(n, b.max_qlen, b.max_tlen, C.byref(sc), 0)
scratch = _DevBuf(lib, nbytes)
cig = _DevBuf(lib, b.qbytes + b.tbytes + 16)
nops = _DevBuf(lib, 4 * n)
m = b.d_meta


def tb():
    _chk(lib, lib.agatha_amd_align_traceback(eng.stream, b.d_pk_q.ptr, b.d_pk_t.ptr, m[2].ptr, m[3].ptr, m[0].ptr, m[1].ptr, n,
                                             b.max_qlen, b.max_tlen, C.byref(sc), b.d_res[0].ptr, b.d_res[1].ptr, b.d_res[2].ptr,
                                             cig.ptr, nops.ptr, b.d_ws.ptr, b.ws_bytes, scratch.ptr, nbytes))


for name, fn in (("align", lambda: b.align(sc)), ("align_traceback", tb)):
    fn(); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt * 1e3:.2f} ms  {cells / dt / 1e9:.1f} GCUPS  (n={n}, code area {per * n / 2**30:.2f} GiB)")
