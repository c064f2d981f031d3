"""Developer tool (GPU box): kernel time of one batch under debug options.  python tools/opt_sweep.py cfg pairs "opt=v,opt=v" "opt=v" ..."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload
cfgname, n = sys.argv[1], int(sys.argv[2])
eng = agatha_amd.Engine(0)
qs, ts = getattr(workload, cfgname)(n=n)
qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
W = int(os.environ.get("BAND", {"cfg_c2": 500, "cfg_c3": 1500}.get(cfgname, 751)))
# SCORING=m,x,q,r (default the library's m2 x4 q4 r2; the reference's bench command, AGAThA.sh:44, is 1,4,6,2)
m_, x_, q_, r_ = (int(v) for v in os.environ.get("SCORING", "2,4,4,2").split(","))
sc = agatha_amd.Scores.make(m=m_, x=x_, q=q_, r=r_, w=W)
defaults = {}
for spec in sys.argv[3:] or [""]:
    opts = dict(a.split("=") for a in spec.split(",") if a)
    for k, v in opts.items():
        defaults.setdefault(k, agatha_amd.get_debug_option(k))
        agatha_amd.set_debug_option(k, int(v))
    ms = []
    for rep in range(6):
        e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms.append(eng.elapsed_ms(e0, e1))
    st = b.step_stats()
    print(f"{spec or 'default':40s} min {min(ms[1:]):.2f} median {np.median(ms[1:]):.2f} ms  choice {b.kernel_choice()} sched {b.schedule_info()[0]} split {b.split_info()} value/key steps {st[0]}/{st[1]} over {st[2]} back {st[15]}", flush=True)
    for k in opts:
        agatha_amd.set_debug_option(k, defaults[k])
b.free()
