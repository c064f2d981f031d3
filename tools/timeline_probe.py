"""Developer tool (GPU box): wave end times / steps of the int16 kernel on C1, optionally with N runs in a fraction of the queries.
    python tools/timeline_probe.py [n_run_frac] [pairs]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
eng = agatha_amd.Engine(0)
qs, ts = workload.cfg_c1(n=n)
if frac > 0: qs = workload.add_n_runs(qs, frac, seed=7)
qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
sc = agatha_amd.Scores.make()
agatha_amd.set_debug_option("timeline", 1)
for rep in range(2):
    e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms = eng.elapsed_ms(e0, e1)
t = b.timeline().astype(np.int64); t = t[t[:, 1] != 0]
t0 = t[:, 0].min(); st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
steps = t[:, 4]
print(f"n_run_frac={frac} align={ms:.2f} ms waves={len(t)} schedule={b.schedule_info()} steps={b.step_stats()[:4]}")
print("  end us: min %.0f p10 %.0f median %.0f p90 %.0f p99 %.0f max %.0f" % (en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), np.percentile(en, 99), en.max()))
print("  steps per wave: min %d median %d p99 %d max %d" % (steps.min(), np.median(steps), np.percentile(steps, 99), steps.max()))
us = (en - st) / np.maximum(steps, 1)
print("  us per step: min %.2f median %.2f p90 %.2f p99 %.2f max %.2f" % (us.min(), np.median(us), np.percentile(us, 90), np.percentile(us, 99), us.max()))
late = np.argsort(en)[-8:]
for j in late: print("   late wave %d: end %.0f steps %d us/step %.2f" % (j, en[j], steps[j], us[j]))
b.free()
