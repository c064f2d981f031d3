"""Developer tool (GPU box): the int16 kernel's step statistics on C0 / C1 pairs at a scoring under window options.
python tools/gpu_win_stats.py cfg pairs "opt=v,opt=v" ...   (SCORING=m,x,q,r)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload
cfgname, n = sys.argv[1], int(sys.argv[2])
eng = agatha_amd.Engine(0)
qs, ts = getattr(workload, cfgname)(n=n)
qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
m_, x_, q_, r_ = (int(v) for v in os.environ.get("SCORING", "1,4,6,2").split(","))
sc = agatha_amd.Scores.make(m=m_, x=x_, q=q_, r=r_, w=751)
for spec in sys.argv[3:] or [""]:
    opts = dict(a.split("=") for a in spec.split(",") if a)
    saved = {k: agatha_amd.get_debug_option(k) for k in opts}
    for k, v in opts.items():
        agatha_amd.set_debug_option(k, int(v))
    b.align(sc); eng.synchronize()
    st = b.step_stats()
    print(f"{spec or 'default':36s} value/key {st[0]}/{st[1]} started over {st[2]} back to checkpoint {st[15]} pairs started {st[3]}; "
          f"value step not calm {st[4]}, key step needs a cell it does not know {st[5]}, ended without the cell {st[6]}; first not-calm {list(st[7:14])}", flush=True)
    for k, v in saved.items():
        agatha_amd.set_debug_option(k, v)
b.free()
