"""Developer tool (GPU box): BASELINE configs[4] (mixed lengths, log-uniform 1-100 kb, 30 % broken pairs) under the split between
the two int16 shapes: kernel-only time per option set.   python tools/c4_split_sweep.py [pairs] "opt=v,opt=v" ..."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload, shard
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
eng = agatha_amd.Engine(0)
t0 = time.time()
qs, ts = workload.cfg_c4(n=n)
qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
del qs, ts
b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
sc = agatha_amd.Scores.make()
cells = float(shard.nominal_cells(ql, tl, 751).sum())
steps = ((ql + 7) // 8 + (tl + 7) // 8).astype(np.int64)
print(f"{n} pairs generated in {time.time() - t0:.0f} s; steps total {steps.sum():.3e} longest {steps.max()}; pairs over 50/75/90 kb: "
      f"{int((ql > 50000).sum())} / {int((ql > 75000).sum())} / {int((ql > 90000).sum())}", flush=True)
ref = None
for spec in sys.argv[2:] or [""]:
    opts = dict(a.split("=") for a in spec.split(",") if a)
    old = {k: agatha_amd.get_debug_option(k) for k in opts}
    for k, v in opts.items(): agatha_amd.set_debug_option(k, int(v))
    ms = []
    for rep in range(3):
        e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms.append(eng.elapsed_ms(e0, e1))
    b.download(); eng.synchronize()
    res = np.stack([np.asarray(a).copy() for a in b.res_host[:3]])
    same = "" if ref is None else (" results identical" if np.array_equal(res, ref) else " RESULTS DIFFER")
    if ref is None: ref = res
    print(f"{spec or 'default':40s} ms {' '.join('%.1f' % m for m in ms)} -> {cells / min(ms[1:]) / 1e9:.2f} TCUPS  choice {b.kernel_choice()} split {b.split_info()} kinds {b.pair_kinds()}{same}", flush=True)
    for k, v in old.items(): agatha_amd.set_debug_option(k, v)
b.free()
