"""Developer tool (GPU box): C1 pairs a share of whose reads have a BURST of errors in the middle (a few hundred bases at 35-45 % error: the
score dips by a few hundred and recovers) -- z-drop comes into reach of a value step's bounds without firing, the pair goes back to a
checkpoint and, finding nothing wrong, carries on.  Kernel time and step statistics.   python3 tools/gpu_dips.py [pairs] [share] [burst bases]"""
import sys, numpy as np
sys.path.insert(0, ".")
import agatha_amd
from agatha_amd import workload as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
share = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
import os
RANDOM = os.environ.get("DIPS_RANDOM") == "1"          # the burst is unrelated sequence instead of a 40 % error stretch
eng = agatha_amd.Engine(0)
import os
CFG = os.environ.get("CFG", "cfg_c1")          # (CFG=cfg_c2 BAND=500: the HiFi shape, two register pairs per lane, checkpoints with bookkeeping)
_m, _x, _q, _r = (int(v) for v in os.environ.get("SCORING", "2,4,4,2").split(","))      # (SCORING=1,4,6,2: the reference's bench command)
sc = agatha_amd.Scores.make(m=_m, x=_x, q=_q, r=_r, w=int(os.environ.get("BAND", "751")))
qs0, ts0 = getattr(W, CFG)(n=n)
for burst in ([int(sys.argv[3])] if len(sys.argv) > 3 and sys.argv[3].isdigit() else ([0, 60, 100, 140, 180, 0] if RANDOM else [0, 150, 250, 350, 500, 0])):
    rng = np.random.default_rng(11)
    ts = []
    for t in ts0:
        # (a read too short to hold the burst between its first and last fifth stays as it is: the bundled-dataset shape has 200-base reads)
        if burst and rng.random() < share and len(t) * 4 // 5 - burst > len(t) // 5:
            a = np.frombuffer(t, np.uint8).copy()
            at = int(rng.integers(len(a) // 5, len(a) * 4 // 5 - burst))
            seg = W.random_seq(rng, burst) if RANDOM else W.mutate(rng, a[at:at + burst], 0.15, 0.12, 0.13)
            t = np.concatenate([a[:at], seg, a[at + burst:]]).tobytes()
        ts.append(t)
    qb, qo, ql = W.make_batch(qs0); tb, to, tl = W.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
    b.align(sc); eng.synchronize()
    ms = []
    for _ in range(3):
        e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms.append(eng.elapsed_ms(e0, e1))
    st = b.step_stats()
    b.download(); eng.synchronize()
    full = int(np.sum(b.res_host[1] + b.res_host[2] + 2 >= 0.97 * (ql.astype(np.int64) + tl.astype(np.int64))))
    print(f"burst {burst:4d} bases in {share:.0%} of the reads: align {min(ms):6.2f} ms  value steps {st[0]} key steps {st[1]} started over {st[2]} back to checkpoint {st[15]} "
          f"not calm on values {st[4]}; pairs aligned to their end {full} of {n}", flush=True)
    if "--timeline" in sys.argv:
        agatha_amd.set_debug_option("timeline", 1)
        b.align(sc); eng.synchronize()
        t = b.timeline().astype(np.int64); t = t[t[:, 1] != 0]
        en = (t[:, 1] - t[:, 0].min()) / 100.0
        print("      waves", len(t), "end us p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; steps per wave p50 %d p99 %d max %d; key steps per wave p50 %d max %d" % (
            np.percentile(en, 10), np.median(en), np.percentile(en, 90), np.percentile(en, 99), en.max(), np.median(t[:, 4]), np.percentile(t[:, 4], 99), t[:, 4].max(),
            np.median(t[:, 4] - t[:, 5]), (t[:, 4] - t[:, 5]).max()), flush=True)
        agatha_amd.set_debug_option("timeline", 0)
    b.free()
