"""Developer tool (GPU box): a batch of C1 pairs with a few ultra-long ones among them -- the batch the split between the two int16
shapes is for.  python tools/mix_sweep.py [n_c1] [n_long] "opt=v" ..."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload
n1, n3 = int(sys.argv[1]), int(sys.argv[2])
eng = agatha_amd.Engine(0)
qs, ts = workload.cfg_c1(n=n1)
ql_, tl_ = workload.cfg_c3(n=n3)
rng = np.random.default_rng(5)
pos = sorted(rng.choice(n1, n3, replace=False).tolist())
for k, p in enumerate(pos):
    qs.insert(p + k, ql_[k]); ts.insert(p + k, tl_[k])
qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
sc = agatha_amd.Scores.make()
from agatha_amd import shard
cells = float(shard.nominal_cells(ql, tl, 751).sum())
for spec in sys.argv[3:] or [""]:
    opts = dict(a.split("=") for a in spec.split(",") if a)
    old = {k: agatha_amd.get_debug_option(k) for k in opts}
    for k, v in opts.items(): agatha_amd.set_debug_option(k, int(v))
    ms = []
    for rep in range(4):
        e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms.append(eng.elapsed_ms(e0, e1))
    print(f"{spec or 'default':24s} min {min(ms[1:]):.2f} ms = {cells / min(ms[1:]) / 1e9:.2f} TCUPS  choice {b.kernel_choice()} split {b.split_info()} sched {b.schedule_info()[0]} kinds {b.pair_kinds()}", flush=True)
    for k, v in old.items(): agatha_amd.set_debug_option(k, v)
b.free()
