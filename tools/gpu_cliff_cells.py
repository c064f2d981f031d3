"""Developer tool (GPU box): kernel time of cells of tools/cliff_sweep.py's grid -- the ones the CPU sweep names as the worst -- next to the
clean batch of the same shape and scoring (error 1 %, nothing cut, no N runs).  A cell is 'shape scoring err cut nfrac'.
    python tools/gpu_cliff_cells.py "C1 m1x4q6r2 0.15 1.0 0.0" ..."""
import os, sys, zlib, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload as wl
SC = {"m2x4q4r2": (2, 4, 4, 2), "m1x4q6r2": (1, 4, 6, 2), "m1x9q16r2": (1, 9, 16, 2), "m1x19q39r3": (1, 19, 39, 3), "m2x8q12r2": (2, 8, 12, 2)}
SHAPES = {"C0": (lambda rng: int(np.clip(np.rint(rng.normal(3000, 1000)), 200, 8000)), 751, 20000),
          "C1": (lambda rng: int(np.clip(np.rint(rng.normal(10000, 1000)), 8000, 12000)), 751, 10000),
          "C2": (lambda rng: int(rng.integers(15000, 20001)), 500, 12500)}
eng = agatha_amd.Engine(0)
SCALE = float(os.environ.get("PAIRS_SCALE", "1.0"))


def run(shape, scoring, err, cut, nf):
    lfn, band, n = SHAPES[shape]
    n = int(n * SCALE)
    seed = 0xC11FF + zlib.crc32(repr((shape, scoring, err, cut, nf)).encode()) % 100000
    qs, ts = wl.make_pairs(seed, n, lfn, 0.3 * err, 0.3 * err, 0.4 * err)
    if cut < 1.0:
        ts = [t[:max(1, int(len(t) * cut))] for t in ts]
    if nf > 0:
        qs = wl.add_n_runs(qs, nf, seed=seed + 1)
    qb, qo, ql = wl.make_batch(qs); tb, to, tl = wl.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
    m, x, q, r = SC[scoring]
    sc = agatha_amd.Scores.make(m=m, x=x, q=q, r=r, w=band)
    ms = []
    for _ in range(4):
        e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms.append(eng.elapsed_ms(e0, e1))
    st = b.step_stats() + b.flat_stats(); kinds = b.pair_kinds(); choice = b.kernel_choice(); sched = b.schedule_info()[0]
    if os.environ.get("TIMELINE"):
        agatha_amd.set_debug_option("timeline", 1)
        b.align(sc); eng.synchronize()
        t = b.timeline().astype(np.int64); t = t[t[:, 1] != 0]
        en = (t[:, 1] - t[:, 0].min()) / 100.0
        print("      waves", len(t), "end us p50 %.0f p90 %.0f p99 %.0f max %.0f; steps per wave p50 %d p90 %d p99 %d max %d; value steps per wave p50 %d max %d" % (
            np.median(en), np.percentile(en, 90), np.percentile(en, 99), en.max(), np.median(t[:, 4]), np.percentile(t[:, 4], 90), np.percentile(t[:, 4], 99), t[:, 4].max(), np.median(t[:, 5]), t[:, 5].max()))
        late = np.argsort(en)[-5:]
        print("      the five last waves: end", en[late].astype(int), "steps", t[late, 4], "value steps", t[late, 5])
        agatha_amd.set_debug_option("timeline", 0)
    b.free()
    return min(ms[1:]), st, kinds, choice, sched


for spec in sys.argv[1:]:
    shape, scoring, err, cut, nf = spec.split()
    err, cut, nf = float(err), float(cut), float(nf)
    t_clean, st0, _, _, _ = run(shape, scoring, 0.01, 1.0, 0.0)
    t, st, kinds, choice, sched = run(shape, scoring, err, cut, nf)
    print(f"{shape} {scoring:11s} err {err:.2f} cut {cut:.2f} N-run {nf:.2f}: {t:7.2f} ms = {t / t_clean:5.2f} x the clean batch ({t_clean:6.2f} ms); {choice} static schedule {sched}; "
          f"value/key wave-steps {st[0]}/{st[1]}, started over {st[2]}, back to a checkpoint {st[15]}, pairs to int32 {kinds[2]}; flat {st[41]} of {st[40]} pairs asked, {st[42]} young pairs restarted on key steps", flush=True)
