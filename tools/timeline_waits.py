"""Developer tool (GPU box, library built with -DAGATHA16_DIAG): per wave of the int16 kernel on C1 (+ N runs: N_RUN_FRAC) the
steps, value steps and rounds spent with a lane group waiting for a suspended pair; which waves end last and why."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
eng = agatha_amd.Engine(0)
qs, ts = workload.cfg_c1(n=n)
if os.environ.get("N_RUN_FRAC"):
    qs = workload.add_n_runs(qs, float(os.environ["N_RUN_FRAC"]), seed=7)
qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
sc = agatha_amd.Scores.make()
agatha_amd.set_debug_option("timeline", 1)
for rep in range(2):
    b.align(sc); eng.synchronize()
t = b.timeline().astype(np.int64)
idx = np.nonzero(t[:, 1] != 0)[0]
t = t[idx]
t0 = t[:, 0].min()
en = (t[:, 1] - t0) / 100.0
steps, fast, waits = t[:, 4], t[:, 5], t[:, 6]
print("waves", len(t), "end us median %.0f max %.0f" % (np.median(en), en.max()))
print("steps: median", int(np.median(steps)), "max", int(steps.max()), " waves over median+20:", int((steps > np.median(steps) + 20).sum()))
print("key steps per wave: median", int(np.median(steps - fast)), "p90", int(np.percentile(steps - fast, 90)), "max", int((steps - fast).max()))
print("wait rounds per wave: median", int(np.median(waits)), "max", int(waits.max()), "waves with waits:", int((waits > 0).sum()))
o = np.argsort(-en)[:12]
for k in o:
    print("  wave %4d end %.0f us steps %d key %d wait rounds %d  us/step %.2f" % (idx[k], en[k], steps[k], steps[k] - fast[k], waits[k], (t[k, 1] - t[k, 0]) / 100.0 / steps[k]))
print("corr(end, key steps) %.2f  corr(end, steps) %.2f  corr(end, waits) %.2f" % (np.corrcoef(en, steps - fast)[0, 1], np.corrcoef(en, steps)[0, 1], np.corrcoef(en, waits)[0, 1]))
