"""Developer tool (GPU box): per-wave timeline of the packed-int16 kernel on the C1 batch (debug option "timeline").
Prints when waves start and end, grouped by XCD / CU / SIMD, for the static schedule and for the work queue."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfgname = sys.argv[2] if len(sys.argv) > 2 else "cfg_c1"
eng = agatha_amd.Engine(0)
qs, ts = getattr(workload, cfgname)(n=n)
if os.environ.get("N_RUN_FRAC"):                     # (a run of N in that fraction of the DP-row sequences, like bench.py --n-run-frac)
    qs = workload.add_n_runs(qs, float(os.environ["N_RUN_FRAC"]), seed=7)
qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
sc = agatha_amd.Scores.make()
agatha_amd.set_debug_option("timeline", 1)
modes = [(-1, -1, 0), (1, -1, 0), (-1, 0, 0)] if cfgname != 'cfg_c1' else [(0, -1, 0), (0, -1, 8), (0, 0, 8)]
for nomig, pb, duty in modes:
    agatha_amd.set_debug_option("no_migrate", nomig)
    agatha_amd.set_debug_option("prio_slice", pb)
    agatha_amd.set_debug_option("prio_duty", duty)
    print("duty", duty, end=" ")
    for rep in range(2):
        e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms = eng.elapsed_ms(e0, e1)
    t = b.timeline().astype(np.int64)
    t = t[t[:, 1] != 0]
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0          # microseconds
    hw = t[:, 2]; simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 3; xcc = t[:, 3] & 15
    wid = hw & 15
    key = xcc * 1000000 + se * 10000 + cu * 100 + simd
    cukey = xcc * 10000 + se * 100 + cu
    print(f"prio_slice={pb} wave slot ids:", dict(zip(*np.unique(wid, return_counts=True))), "fast waves by slot parity:", [float(np.round(np.median(((t[:, 1] - t[:, 0]) / 100.0 / np.maximum(t[:, 4], 1))[(wid & 1) == q]), 2)) for q in (0, 1)], "by workgroup half:", [float(np.round(np.median(((t[:, 1] - t[:, 0]) / 100.0 / np.maximum(t[:, 4], 1))[(np.arange(len(t)) // 4 >= len(t) // 8) == q]), 2)) for q in (False, True)])
    nw = len(t)
    same = sum(int(key[j] == key[j + nw // 2]) for j in range(nw // 2)) if nw % 8 == 0 else -1
    print("  waves (b, w) and (b + half, w) on the same SIMD:", same, "of", nw // 2)
    bywave = {}
    for j in range(nw):
        bywave.setdefault(int(key[j]), []).append(j)
    rel = {}
    for kk_, js in bywave.items():
        if len(js) == 2:
            a_, b_ = sorted(js)
            r_ = ((b_ // 4) - (a_ // 4), (a_ % 4), (b_ % 4))
            rel[r_] = rel.get(r_, 0) + 1
    print("  SIMD partners (block distance, wave of first, wave of second) -> count:", sorted(rel.items(), key=lambda x: -x[1])[:12])
    cus = {}
    for j in range(nw):
        cus.setdefault(int(cukey[j]), set()).add(j // 4)
    print("  workgroups per CU:", sorted(set(tuple(sorted(v)) for v in list(cus.values())[:6])), "block distance of the two workgroups of a CU:", sorted(set(max(v) - min(v) for v in cus.values() if len(v) == 2))[:10])
    print(f"no_migrate={nomig} align={ms:.2f} ms waves={len(t)} schedule={b.schedule_info()}")
    print("  start us: min %.0f max %.0f   end us: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (st.min(), st.max(), en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
    print("  mean life %.0f us = %.1f%% of the kernel; steps per wave: min %d median %d max %d; pairs per wave (x4 groups): %d..%d" % ((en - st).mean(), 100 * (en - st).mean() / en.max(), t[:, 4].min(), np.median(t[:, 4]), t[:, 4].max(), t[:, 5].min(), t[:, 5].max()))
    us_per_step = (en - st) / np.maximum(t[:, 4], 1)
    print("  us per step: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f" % (us_per_step.min(), np.percentile(us_per_step, 10), np.median(us_per_step), np.percentile(us_per_step, 90), us_per_step.max()))
    key = xcc * 1000000 + se * 10000 + cu * 100 + simd
    cukey = xcc * 10000 + se * 100 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print("  distinct (xcc,se,cu,simd):", len(uniq), "waves per SIMD histogram:", dict(zip(*np.unique(cnt, return_counts=True))))
    cukey = xcc * 10000 + se * 100 + cu
    u2, c2 = np.unique(cukey, return_counts=True)
    print("  distinct CUs:", len(u2), "waves per CU histogram:", dict(zip(*np.unique(c2, return_counts=True))))
    for x in range(8):
        m = xcc == x
        if m.any(): print("   xcc %d: waves %d  end min %.0f median %.0f max %.0f  us/step median %.2f" % (x, m.sum(), en[m].min(), np.median(en[m]), en[m].max(), np.median(us_per_step[m])))
    for q in (0, 1):
        m = (wid & 1) == q
        print("   slot %d: end min %.0f median %.0f max %.0f" % (q, en[m].min(), np.median(en[m]), en[m].max()))
    # per CU: spread of the end times of its 8 waves, and of the CU means
    cu_mean = np.array([en[cukey == u].mean() for u in u2]); cu_max = np.array([en[cukey == u].max() for u in u2])
    print("   per-CU mean end: min %.0f median %.0f max %.0f ; per-CU max end: min %.0f median %.0f max %.0f" % (cu_mean.min(), np.median(cu_mean), cu_mean.max(), cu_max.min(), np.median(cu_max), cu_max.max()))
    # speed against how many waves share the SIMD
    per = dict(zip(uniq, cnt))
    share = np.array([per[k] for k in key])
    for c in sorted(set(share)):
        m = share == c
        print("   waves on SIMDs holding %d wave(s): %d, us/step median %.2f, end median %.0f" % (c, m.sum(), np.median(us_per_step[m]), np.median(en[m])))
b.free()
