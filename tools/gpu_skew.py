"""Developer tool (GPU box): C1-like pairs whose two sequences differ in length (the target is cut to a fraction of the read, or
the read to a fraction of the target), or a share of whose reads have an unrelated tail: kernel time and how many pairs the int16
kernel starts over / takes back to a checkpoint (the distances and reasons are counted by -DAGATHA16_DIAG builds only).
Usage: python3 tools/gpu_skew.py [pairs]"""
import sys, numpy as np
sys.path.insert(0, ".")
import agatha_amd
from agatha_amd import workload as W
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10000
eng = agatha_amd.Engine(0)
import os
CFG = os.environ.get("CFG", "cfg_c1")          # (CFG=cfg_c2 BAND=500: the HiFi shape, two register pairs per lane, checkpoints with bookkeeping)
_m, _x, _q, _r = (int(v) for v in os.environ.get("SCORING", "2,4,4,2").split(","))      # (SCORING=1,4,6,2: the reference's bench command)
sc = agatha_amd.Scores.make(m=_m, x=_x, q=_q, r=_r, w=int(os.environ.get("BAND", "751")))
qs0, ts0 = getattr(W, CFG)(n=n)
rng = np.random.default_rng(5)
for name, fq, ft in (("equal", 1.0, 1.0), ("target 90 %", 1.0, 0.9), ("target 75 %", 1.0, 0.75), ("query 90 %", 0.9, 1.0), ("query 75 %", 0.75, 1.0),
                     ("1 % broken", -0.01, 1.0), ("5 % broken", -0.05, 1.0), ("30 % broken", -0.3, 1.0)):
    if fq < 0:          # that fraction of the pairs: the read's tail (from a random point on) is unrelated sequence -- z-drop ends the extension there
        qs, ts = list(qs0), []
        for t in ts0:
            if rng.random() < -fq:
                a = np.frombuffer(t, np.uint8).copy(); h = int(rng.integers(len(a) // 10, len(a)))
                a[h:] = W.random_seq(rng, len(a) - h); t = a.tobytes()
            ts.append(t)
    else:
        qs = [q[:max(1, int(len(q) * fq))] for q in qs0]
        ts = [t[:max(1, int(len(t) * ft))] for t in ts0]
    qb, qo, ql = W.make_batch(qs); tb, to, tl = W.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
    b.align(sc); eng.synchronize()
    ms = []
    for _ in range(3):
        e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms.append(eng.elapsed_ms(e0, e1))
    st = b.step_stats()
    cells = W.nominal_cells_total(ql, tl, int(os.environ.get("BAND", "751")))
    print(f"{name:12s} align {min(ms):7.2f} ms  {cells / min(ms) / 1e6:7.1f} GCUPS  value steps {st[0]} key steps {st[1]} started over {st[2]} back to checkpoint {st[15]} "
          f"to int32 {st[24]} ended without the cell {st[6]} not calm on values {st[4]} key step without the cell {st[5]} back by n x 256 steps {st[16:24]} started over at n x 512 steps {st[25:33]} why (second time, no checkpoints, c0 < span, c0 <= first, slot invalid) {st[33:38]} suspended with an older checkpoint {st[38]}")
    if "--timeline" in sys.argv:
        agatha_amd.set_debug_option("timeline", 1)
        b.align(sc); eng.synchronize()
        t = b.timeline().astype(np.int64); t = t[t[:, 1] != 0]
        en = (t[:, 1] - t[:, 0].min()) / 100.0
        print("      waves", len(t), "end us p50 %.0f p90 %.0f p99 %.0f max %.0f; steps per wave p50 %d p90 %d p99 %d max %d" % (
            np.median(en), np.percentile(en, 90), np.percentile(en, 99), en.max(), np.median(t[:, 4]), np.percentile(t[:, 4], 90), np.percentile(t[:, 4], 99), t[:, 4].max()))
        agatha_amd.set_debug_option("timeline", 0)
    b.free()
