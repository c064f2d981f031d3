"""Developer tool (GPU box): what a value step and a key step of the packed-int16 kernel cost, from the per-wave timeline (debug option
"timeline": start, end, steps, value steps of every wave): least-squares fit of a wave's life against its two step counts, and where
the kernel's time goes between the mean wave and the last one.   python tools/timeline_steps.py [pairs] [cfg] [option=value ...]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfgname = sys.argv[2] if len(sys.argv) > 2 else "cfg_c1"
opts = dict(a.split("=") for a in sys.argv[3:])
eng = agatha_amd.Engine(0)
qs, ts = getattr(workload, cfgname)(n=n)
qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
m_, x_, q_, r_ = (int(v) for v in os.environ.get("SCORING", "2,4,4,2").split(","))
sc = agatha_amd.Scores.make(m=m_, x=x_, q=q_, r=r_, w=int(opts.pop("w", 751)))
agatha_amd.set_debug_option("timeline", 1)
for k, v in opts.items():
    agatha_amd.set_debug_option(k, int(v))
for rep in range(3):
    e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms = eng.elapsed_ms(e0, e1)
t = b.timeline().astype(np.int64)
t = t[t[:, 1] != 0]
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
steps, fast = t[:, 4].astype(float), t[:, 5].astype(float)
keys = steps - fast
life = en - st
A = np.stack([fast, keys, np.ones_like(fast)], 1)
coef, *_ = np.linalg.lstsq(A, life, rcond=None)
print(f"align {ms:.2f} ms, {len(t)} waves, schedule {b.schedule_info()}, kernel choice {b.kernel_choice()}, step stats {b.step_stats()[:4]}")
print("wave end us: min %.0f p10 %.0f median %.0f p90 %.0f p99 %.0f max %.0f ; start max %.0f" % (en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), np.percentile(en, 99), en.max(), st.max()))
print("lazy value steps (one pair per wave: no lower bound, no reduction, no test): %.1f %% of the value steps" % (100.0 * t[:, 7].sum() / max(fast.sum(), 1)))
print("steps per wave: min %d median %d max %d ; key steps per wave: min %d median %d p90 %d max %d" % (steps.min(), np.median(steps), steps.max(), keys.min(), np.median(keys), np.percentile(keys, 90), keys.max()))
print("fit life = %.3f us x value steps + %.3f us x key steps + %.0f us ; mean life %.0f us = %.1f %% of the kernel" % (coef[0], coef[1], coef[2], life.mean(), 100 * life.mean() / en.max()))
late = en > np.percentile(en, 99)
print("the last 1 %% of the waves: key steps median %d (all: %d), steps median %d, us per step %.2f (all: %.2f)" % (np.median(keys[late]), np.median(keys), np.median(steps[late]), np.median(life[late] / steps[late]), np.median(life / steps)))
hw = t[:, 2]; wid = hw & 15; xcc = t[:, 3] & 15
for q in (0, 1):
    m = (wid & 1) == q
    print("  SIMD slot %d: end median %.0f max %.0f, us/step median %.3f" % (q, np.median(en[m]), en[m].max(), np.median((life / steps)[m])))
for x in range(8):
    m = xcc == x
    if m.any(): print("  xcc %d: end median %.0f max %.0f us/step %.3f" % (x, np.median(en[m]), en[m].max(), np.median((life / steps)[m])))
if os.environ.get("TIMELINE_DUMP"):
    np.save(os.environ["TIMELINE_DUMP"], t)
# waves by their number of key steps: how long they live
for lo_, hi_ in ((0, 100), (100, 200), (200, 300), (300, 400), (400, 600), (600, 900), (900, 100000)):
    m = (keys >= lo_) & (keys < hi_)
    if m.any(): print("  waves with %4d..%-5d key steps: %5d, life median %.0f us, end median %.0f max %.0f, us per value step (life - 1.35 x key) %.3f" % (lo_, hi_, m.sum(), np.median(life[m]), np.median(en[m]), en[m].max(), np.median((life[m]) / (fast[m] + 1.35 * keys[m]))))
print("corr(end, key steps) %.2f corr(end, steps) %.2f" % (np.corrcoef(en, keys)[0, 1], np.corrcoef(en, steps)[0, 1]))
b.free()
