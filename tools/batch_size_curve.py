"""Developer tool (GPU box): throughput of agatha_amd_align as a function of the batch size -- the reference's CLI cuts batches of `-a 8192`
(AGAThA/src/args_parser.cpp:23, test_prog.cpp:287) and a mapper sends whatever it has, so the 10 000-pair headline is one point of a curve.
Per size: the first n pairs of ONE generated batch (so that every size sees the same length distribution), kernel time by HIP events around
the align call (sort, schedule, record and the align kernels: what the reference's raw.log brackets), nominal TCUPS, the shape the device
chose, static schedule / split, and the time of the same call with every candidate forced in turn where FORCE=1.

    python3 tools/batch_size_curve.py [cfg_c1|cfg_c2|cfg_c0] [sizes comma separated]        (SCORING=m,x,q,r  BAND=w  FORCE=1)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload, shard

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg_c1"
sizes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 512, 1024, 1808, 2048, 4096, 8192, 10000, 16384, 20000]
W = int(os.environ.get("BAND", {"cfg_c2": 500, "cfg_c3": 1500}.get(cfg, 751)))
m_, x_, q_, r_ = (int(v) for v in os.environ.get("SCORING", "2,4,4,2").split(","))
sc = agatha_amd.Scores.make(m=m_, x=x_, q=q_, r=r_, w=W)
eng = agatha_amd.Engine(0)
qs_all, ts_all = getattr(workload, cfg)(n=max(sizes))
force = os.environ.get("FORCE") == "1"
ref_rate = None
rows = []
print(f"# {cfg} band {W} scoring m{m_} x{x_} q{q_} r{r_}; sizes {sizes}")
print(f"# {'pairs':>6s} {'kernel ms':>10s} {'TCUPS':>7s} {'vs 10000':>8s}  shape (kind, lanes/pair, slots/lane)  static-schedule  split")
for n in sizes:
    qs, ts = qs_all[:n], ts_all[:n]
    qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
    cells = float(np.sum(shard.nominal_cells(ql, tl, W)))
    b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
    def timed(reps=5):
        ms = []
        for _ in range(reps + 1):
            e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms.append(eng.elapsed_ms(e0, e1))
        return float(np.median(ms[1:]))
    t = timed()
    choice, sched, split = b.kernel_choice(), b.schedule_info(), b.split_info()
    rate = cells / t / 1e9
    rows.append((n, t, rate, choice, sched, split))
    line = f"  {n:6d} {t:10.3f} {rate:7.3f} {'':8s}  {choice}  {sched[0]} (T {sched[1]}, groups {sched[2]})  {split}"
    if force:
        alts = []
        for fc in range(4):
            agatha_amd.set_debug_option("force_choice", fc)
            try:
                tt = timed(3); alts.append(f"cand{fc}:{tt:.2f}ms/{b.kernel_choice()[1:]}")
            except Exception as e:                          # (a candidate that does not exist for this window)
                alts.append(f"cand{fc}:-")
        agatha_amd.set_debug_option("force_choice", -1)
        line += "   forced: " + " ".join(alts)
    print(line, flush=True)
    b.free()
ref = next((r for r in rows if r[0] == 10000), rows[-1])
print("# relative to the %d-pair rate (%.3f TCUPS):" % (ref[0], ref[2]))
for n, t, rate, *_ in rows:
    print(f"#   {n:6d} pairs  {rate / ref[2]:.3f}" + ("   <-- below 0.75" if n >= 2048 and rate / ref[2] < 0.75 else ""))
