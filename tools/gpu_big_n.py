import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import agatha_amd
from agatha_amd import workload as WL
from oracle import oracle as O
eng = agatha_amd.Engine(0)
for n, lo, hi in ((200000, 50, 300), (60000, 900, 1100)):
    qs, ts = WL.make_pairs(5, n, lambda r: int(r.integers(lo, hi)), 0.03, 0.03, 0.04)
    qb, qo, ql = WL.make_batch(qs); tb, to, tl = WL.make_batch(ts)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    t0 = time.time(); exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=16); t1 = time.time()
    b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack()
    e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(agatha_amd.Scores.make(**p)); eng.record(e1); b.download(); eng.synchronize()
    ok = all(np.array_equal(b.res_host[k], exp[k]) for k in range(3))
    print(n, "pairs ok" if ok else "MISMATCH", "kernel ms", round(eng.elapsed_ms(e0, e1), 2), b.kernel_choice(), b.pair_kinds(), "oracle s", round(t1 - t0, 1), flush=True)
    b.free()
