"""Developer tool (GPU box): the same long pairs with more and more one-base pairs beside them -- does the number of workgroups of the launch change
what a long pair's step costs (placement of the workgroups on the CUs)?   python3 tools/pad_sweep.py [cfg] [pairs] [w] [pads comma separated]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload
cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg_c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
pads = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "0,256,768,1792,3944").split(",")]
eng = agatha_amd.Engine(0)
qs0, ts0 = getattr(workload, cfgname)(n=n)
sc = agatha_amd.Scores.make(w=w)
for pad in pads:
    qs = list(qs0) + [b"A"] * pad; ts = list(ts0) + [b"A"] * pad
    qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
    ms = []
    for rep in range(4):
        e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms.append(eng.elapsed_ms(e0, e1))
    print(f"{n} pairs of {cfgname} + {pad:5d} one-base pairs: align min {min(ms[1:]):.2f} median {np.median(ms[1:]):.2f} ms  choice {b.kernel_choice()} split {b.split_info()} steps {b.step_stats()[:2]}", flush=True)
    b.free()
