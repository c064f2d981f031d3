"""Developer tool (GPU box): where the two waves of a <128, 1> workgroup (one pair on two waves) land -- same SIMD or not -- and what a step costs
either way (the per-wave timeline's HW_ID: SIMD = bits 5:4, CU = bits 11:8, SE = bits 15:13 on gfx9).   python3 tools/x2_placement.py [pairs] [cfg] [w]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfgname = sys.argv[2] if len(sys.argv) > 2 else "cfg_c3"
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
eng = agatha_amd.Engine(0)
qs, ts = getattr(workload, cfgname)(n=n)
pad = max(0, 4200 - n)           # (a workspace for <= 4 096 pairs holds no timeline area: one-base pairs make up the number; they end in their dry step)
qs = list(qs) + [b"A"] * pad; ts = list(ts) + [b"A"] * pad
qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
sc = agatha_amd.Scores.make(w=w)
agatha_amd.set_debug_option("timeline", 1)
for rep in range(2):
    e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms = eng.elapsed_ms(e0, e1)
t = b.timeline().astype(np.int64)
print(f"align {ms:.2f} ms, kernel choice {b.kernel_choice()}, split {b.split_info()}, rows {len(t)}, non-empty {(t[:, 1] != 0).sum()}")
hw = t[:, 2]
simd, cu, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, t[:, 3] & 15
life = (t[:, 1] - t[:, 0]) / 100.0
steps = np.maximum(t[:, 4], 1)
a, c = np.arange(0, len(t) - 1, 2), np.arange(1, len(t), 2)
ok = (t[a, 1] != 0) & (t[c, 1] != 0) & (t[a, 4] > 1000)
a, c = a[ok], c[ok]
same_cu = (cu[a] == cu[c]) & (se[a] == se[c]) & (xcc[a] == xcc[c])
same_simd = same_cu & (simd[a] == simd[c])
print("workgroups", len(a), "both waves on one CU", int(same_cu.sum()), "on one SIMD", int(same_simd.sum()))
for name, m in (("same SIMD", same_simd), ("different SIMDs", ~same_simd)):
    if m.any(): print("  %-16s %4d workgroups: us per step median %.3f (steps median %d)" % (name, m.sum(), np.median((life[a] / steps[a])[m]), np.median(steps[a][m])))
t0 = t[t[:, 1] != 0, 0].min()
lw = np.concatenate([a, c])
print("long pairs' waves: start us min %.0f max %.0f, end us min %.0f median %.0f max %.0f" % ((t[lw, 0].min() - t0) / 100.0, (t[lw, 0].max() - t0) / 100.0, (t[lw, 1].min() - t0) / 100.0, np.median(t[lw, 1] - t0) / 100.0, (t[lw, 1].max() - t0) / 100.0))
print("all waves: last end us %.0f" % ((t[t[:, 1] != 0, 1].max() - t0) / 100.0))
keyl = ((xcc[lw] * 8 + se[lw]) * 16 + cu[lw]) * 4 + simd[lw]
ul, cl = np.unique(keyl, return_counts=True)
cul, ccl = np.unique(keyl >> 2, return_counts=True)
print("long pairs' waves: SIMDs used", len(ul), "waves per SIMD histogram", np.bincount(cl).tolist(), "; CUs used", len(cul), "waves per CU histogram", np.bincount(ccl).tolist())
# how many waves share a SIMD at all (other workgroups)
key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
live = t[:, 1] != 0
u, cnt = np.unique(key[live], return_counts=True)
print("SIMDs in use", len(u), "waves per SIMD in use: max", cnt.max(), "histogram", np.bincount(cnt).tolist())
b.free()
