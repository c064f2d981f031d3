"""CPU only: a standing sweep for performance cliffs of the packed-int16 kernel's speculative value steps.

The kernel's results are exact whatever happens; its SPEED depends on how many pairs a value step could not decide and had to be
started over (on key steps, from a checkpoint or from their first step), and on how many pairs leave for the int32 kernel.  The
decision logic is emulated lane for lane by oracle/agatha_lanes_model.c (agatha_model_lanes16 with a margin); this tool runs it over

    {scorings} x {error rates} x {target cut to a fraction of its length} x {N-run fraction} x {C0, C1, C2 shapes}

and prints, per cell, the share of pairs started over (kind 2), handed to the int32 kernel (kind 1) and -- from the plain oracle
-- z-dropped, so that a cell with many pairs started over on reads that do NOT break stands out.  Every result is also checked
against the oracle (a mismatch is a bug, not a cliff).  Minutes on 8 cores; no GPU.

    python tools/cliff_sweep.py [--pairs-scale 1.0] [--quick] [--out profiles/r05_v1/cliff_sweep.txt] [--margin 12]
    python tools/cliff_sweep.py --bursts [--shapes C1] [--out profiles/r05_v2/cliff_sweep_bursts.txt]      (reads with a burst of errors; round 5)
"""
import argparse
import itertools
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O                      # noqa: E402
from agatha_amd import workload as wl               # noqa: E402

# (name, m, x, q, r): AGAThA / minimap2 map-ont defaults; the reference's bench command (AGAThA.sh:44) = minimap2 map-hifi's first
# gap cost; minimap2 asm10 and asm5; minimap2 sr
SCORINGS = [("m2x4q4r2", 2, 4, 4, 2), ("m1x4q6r2", 1, 4, 6, 2), ("m1x9q16r2", 1, 9, 16, 2), ("m1x19q39r3", 1, 19, 39, 3), ("m2x8q12r2", 2, 8, 12, 2)]
ERRORS = [0.01, 0.05, 0.10, 0.15]
CUTS = [1.0, 0.9, 0.75]
NFRACS = [0.0, 0.02]
# shape: (name, length law, band, pairs per cell)
SHAPES = [("C0", lambda rng: int(np.clip(np.rint(rng.normal(3000, 1000)), 200, 8000)), 751, 240),
          ("C1", lambda rng: int(np.clip(np.rint(rng.normal(10000, 1000)), 8000, 12000)), 751, 96),
          ("C2", lambda rng: int(rng.integers(15000, 20001)), 500, 48)]


BENCH_PAIRS = {"C0": 20000, "C1": 10000, "C2": 12500}       # pairs of a bench batch of the shape (bench.py CONFIGS), on 8 192 lane groups of 16 lanes


def shape_gs(band):
    """lanes / slots of the int16 throughput shape for this band (align16_kernel.hip: kCfgs16)"""
    win = (band + 7) // 8 + 1
    for g, s in ((16, 2), (16, 4), (16, 6), (32, 4), (32, 6), (64, 4), (64, 6)):
        if g * s >= win:
            return g, s
    raise ValueError(band)


def make_cell(seed, n, length_fn, err, cut, nfrac):
    sub, ins, dele = 0.3 * err, 0.3 * err, 0.4 * err
    qs, ts = wl.make_pairs(seed, n, length_fn, sub, ins, dele)
    if cut < 1.0:
        ts = [t[:max(1, int(len(t) * cut))] for t in ts]
    if nfrac > 0:
        qs = wl.add_n_runs(qs, nfrac, seed=seed + 1)
    return qs, ts


def run_cell(qs, ts, scoring, band, margin, threads):
    _, m, x, q, r = scoring
    p = O.make_params(m=m, x=x, q=q, r=r, w=band)
    qb, qo, ql = wl.make_batch(qs)
    tb, to, tl = wl.make_batch(ts)
    G, S = shape_gs(band)
    import ctypes as C
    asked, flat = C.c_int.in_dll(O.lib(), "agatha_lanes16_asked"), C.c_int.in_dll(O.lib(), "agatha_lanes16_flat")
    asked.value = flat.value = 0
    got = O.lanes16_batch(qb, tb, qo, to, ql, tl, p, G, S, threads=threads, value_step_margin=margin)
    n_asked, n_flat = asked.value, flat.value
    exp = O.align_batch(qb, tb, qo, to, ql, tl, p, wide=True, model=O.MODEL_STEPS, threads=threads)
    bad = int(sum(1 for k in range(len(ql)) if any(int(got[j][k]) != int(exp[j][k]) for j in range(3))))
    kind = got[3]
    # z-dropped (by the plain oracle's count of the anti-diagonals it walked): a pair that ends before its last anti-diagonal
    lib = O.lib()
    import ctypes as C
    lib.agatha_steps_stats_batch.argtypes = [C.c_void_p] * 6 + [C.c_int, C.POINTER(O.Params), C.c_int, C.c_void_p]
    st = np.zeros((len(ql), 7), np.int32)
    lib.agatha_steps_stats_batch(qb.ctypes.data, tb.ctypes.data, qo.ctypes.data, to.ctypes.data, ql.ctypes.data, tl.ctypes.data,
                                 len(ql), C.byref(p), threads, st.ctypes.data)
    zd = st[:, 4] != 0
    over = kind == 2
    return dict(n=len(ql), over=int(over.sum()), over_clean=int((over & ~zd).sum()), back=int((kind == 1).sum()), zdrop=int(zd.sum()),
                mismatch=bad, ineligible=int((kind == 1).sum()) == len(ql), asked=n_asked, flat=n_flat)


def bursts(a):
    """Reads with a stretch of 40 % errors (tools/gpu_dips.py is the GPU side: profiles/r05_v2/probes_m1_before.txt, probation.txt): the score dips
    and recovers; where the dip brings z-drop within reach of a value step's bounds the pair goes back to a checkpoint.  Cost of a cell =
    value steps + 1.4 key steps of everything the model ran (a step run twice counts twice), relative to the same reads without a burst."""
    import ctypes as C
    lib = O.lib()
    span, prob, left = (C.c_int.in_dll(lib, n) for n in ("agatha_lanes16_ck_span", "agatha_lanes16_probation", "agatha_lanes16_left_probation"))
    C.c_int.in_dll(lib, "agatha_lanes16_ck_slots").value = a.slots
    steps = (C.c_longlong * 2).in_dll(lib, "agatha_lanes16_steps")
    per_pair = C.c_void_p.in_dll(lib, "agatha_lanes16_pair_steps")
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    emit(f"# tools/cliff_sweep.py --bursts margin={a.margin}: a burst of errors (sub 15 % ins 12 % del 13 %) in EVERY read of a cell, a checkpoint every {a.span} steps in {a.slots} slots (the kernel: 256, 2),")
    emit("# probation off / on (DESIGN.md 3.6).  back% = pairs that went back to a checkpoint, ret% = returns to value steps per pair that went back,")
    emit("# over% = pairs that (also) started from their first step, cost = (value steps + 1.4 key steps) / the same cell without a burst; every result checked against the oracle")
    emit("# tail% = what ONE such pair costs the wave that holds it on the static schedule, where the kernel ends with its last wave: the steps the pair runs beyond")
    emit("#   its run without the burst (the way back to the checkpoint, run twice) + 0.4 per key step (the whole wave runs the key form while one pair wants it), worst pair")
    emit("#   of the cell, over the steps of a lane group (a bench batch of the shape on 8 192 lane groups: 2.4 / 1.2 / 1.5 pairs).  An ESTIMATE: the chip measured + 19 -> + 10 % (250 bases),")
    emit("#   + 26 -> + 24 % (350), + 11 % (500, C1 at m1 x4 q6 r2 with the burst in every tenth read; profiles/r05_v2/probation.txt) -- the pool of rests takes some of")
    emit("#   it back, waves with two such pairs add to it.  A cell whose tail% is far from 0 is a cliff on the chip whatever its cost column says.")
    emit(f"{'shape':5s} {'scoring':11s} {'burst':>5s} {'pairs':>5s} | {'back%':>6s} {'over%':>6s} {'cost':>6s} {'tail%':>6s} | {'back%':>6s} {'ret%':>6s} {'over%':>6s} {'cost':>6s} {'tail%':>6s}   (probation off | on)")
    shapes = [sh for sh in SHAPES if sh[0] in a.shapes.split(",")]
    for (sname, lfn, band, npairs), sc in itertools.product(shapes, SCORINGS[:3] if not a.quick else SCORINGS[:2]):
        n = max(8, int(npairs * a.pairs_scale))
        _, m, x, q, r = sc
        p = O.make_params(m=m, x=x, q=q, r=r, w=band)
        G, S = shape_gs(band)
        base_cost = None
        base_pair = None
        for burst in (0, 150, 250, 350, 500):
            seed = 0xB0857 + zlib.crc32(repr((sname, sc[0])).encode()) % 100000           # (the same reads for every burst length of a row)
            qs, ts0 = wl.make_pairs(seed, n, lfn, 0.03, 0.03, 0.04)
            rng = np.random.default_rng(seed + 1)
            ts = []
            for t in ts0:
                arr = np.frombuffer(t, np.uint8).copy()
                at = int(rng.integers(len(arr) // 5, max(len(arr) // 5 + 1, len(arr) * 4 // 5 - burst)))
                ts.append(np.concatenate([arr[:at], wl.mutate(rng, arr[at:at + burst], 0.15, 0.12, 0.13), arr[at + burst:]]).tobytes() if burst else t)
            qb, qo, ql = wl.make_batch(qs)
            tb, to, tl = wl.make_batch(ts)
            exp = O.align_batch(qb, tb, qo, to, ql, tl, p, wide=True, model=O.MODEL_STEPS, threads=a.threads)
            cells = []
            for on in (0, 1):
                span.value, prob.value = a.span, on
                left.value = 0; steps[0] = steps[1] = 0
                pp = np.zeros((n, 2), np.int64)
                per_pair.value = pp.ctypes.data
                got = O.lanes16_batch(qb, tb, qo, to, ql, tl, p, G, S, threads=a.threads, value_step_margin=a.margin)
                per_pair.value = None
                span.value = prob.value = 0
                assert all((np.asarray(g) == np.asarray(e)).all() for g, e in zip(got[:3], exp)), "the int16 model disagrees with the oracle"
                back, over = int(((got[3] == 3) | (got[3] == 4)).sum()), int(((got[3] == 2) | (got[3] == 4)).sum())
                cells.append([100.0 * back / n, 100.0 * left.value / max(back, 1), 100.0 * over / n, steps[0] + 1.4 * steps[1], pp])
            if burst == 0:
                base_cost = (cells[0][3], cells[1][3])
                base_pair = (cells[0][4], cells[1][4])
            tails = []
            for on in (0, 1):
                st_, cl_ = cells[on][4], base_pair[on]
                extra = (st_.sum(1) - cl_.sum(1)) + 0.4 * (st_[:, 1] - cl_[:, 1])
                tails.append(100.0 * max(float(extra.max()), 0.0) / (BENCH_PAIRS[sname] / 8192.0 * float(cl_.sum(1).mean())))
            emit(f"{sname:5s} {sc[0]:11s} {burst:5d} {n:5d} | {cells[0][0]:6.1f} {cells[0][2]:6.1f} {cells[0][3] / base_cost[0]:6.3f} {tails[0]:6.1f} | {cells[1][0]:6.1f} {cells[1][1]:6.1f} {cells[1][2]:6.1f} {cells[1][3] / base_cost[1]:6.3f} {tails[1]:6.1f}")
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs-scale", type=float, default=1.0)
    ap.add_argument("--quick", action="store_true", help="two scorings, two error rates, no cut, no N runs (a smoke run)")
    ap.add_argument("--out", default="")
    ap.add_argument("--margin", type=int, default=12, help="fast_margin of the kernel (debug option of the same name)")
    ap.add_argument("--threads", type=int, default=max(1, len(os.sched_getaffinity(0))))
    ap.add_argument("--shapes", default="C0,C1,C2")
    ap.add_argument("--old-window", action="store_true", help="round 4's rule for the window of key steps (3/2 (slack + 7 ge) i / best, no prior, no cap): the 'before' of round 5")
    ap.add_argument("--span", type=int, default=256, help="--bursts: steps between two checkpoints (the kernel: 256 for pairs of 2 048 .. 4 095 steps, debug option ck_shift 28; 27 = 128, 26 = 64)")
    ap.add_argument("--slots", type=int, default=2, help="--bursts: a what-if -- a ring of that many checkpoints instead of the kernel's two (agatha_lanes16_ck_slots)")
    ap.add_argument("--bursts", action="store_true", help="the other table (round 5): reads with a burst of errors (a dip of the score that recovers), with the kernel's checkpoints, probation off and on: who goes back, who returns to value steps, what the batch costs in steps")
    a = ap.parse_args()
    if a.bursts:
        return bursts(a)
    if a.old_window:
        import ctypes as C
        C.c_int.in_dll(O.lib(), "agatha_lanes16_old_window").value = 1
    scorings, errors, cuts, nfracs = SCORINGS, ERRORS, CUTS, NFRACS
    if a.quick:
        scorings, errors, cuts, nfracs = SCORINGS[:2], [0.05, 0.10], [1.0], [0.0]
    shapes = [s for s in SHAPES if s[0] in a.shapes.split(",")]
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    emit(f"# tools/cliff_sweep.py margin={a.margin} pairs-scale={a.pairs_scale}{' --old-window (round 4 rule)' if a.old_window else ''}: oracle/agatha_lanes_model.c (agatha_model_lanes16), every result checked against the oracle")
    emit("# started over = a value step could not decide / the pair ended without the cell of its maximum (kernel: back to a checkpoint or to its first step, on key steps);")
    emit("# 'clean' = of pairs the oracle does NOT z-drop; handed back = left for the int32 kernel; columns in % of the cell's pairs")
    emit("# flat% = pairs that say, at their 64th..127th step, that their score hardly rises (of those that were asked); at 30 % the kernel takes the batch for a FLAT one")
    emit("# and runs it on key steps (1.3-1.4 x its clean batch, profiles/r05_v1/cliff_cells_after.txt): what the model then counts as started over does not happen")
    emit(f"{'shape':5s} {'scoring':11s} {'err':>4s} {'cut':>5s} {'Nrun':>5s} {'pairs':>5s} {'zdrop%':>7s} {'over%':>7s} {'over-clean%':>11s} {'back%':>6s} {'flat%':>6s} {'mismatch':>8s}")
    t0 = time.time()
    worst = []
    for (sname, lfn, band, npairs), sc, err, cut, nf in itertools.product(shapes, scorings, errors, cuts, nfracs):
        n = max(8, int(npairs * a.pairs_scale))
        seed = 0xC11FF + zlib.crc32(repr((sname, sc[0], err, cut, nf)).encode()) % 100000         # (the same pairs in every run of a cell)
        qs, ts = make_cell(seed, n, lfn, err, cut, nf)
        r = run_cell(qs, ts, sc, band, a.margin, a.threads)
        pc = lambda v: 100.0 * v / r["n"]
        flat_pc = 100.0 * r["flat"] / max(r["asked"], 1)
        emit(f"{sname:5s} {sc[0]:11s} {err:4.2f} {cut:5.2f} {nf:5.2f} {r['n']:5d} {pc(r['zdrop']):7.1f} {pc(r['over']):7.2f} {pc(r['over_clean']):11.2f} {pc(r['back']):6.1f} {flat_pc:6.1f} {r['mismatch']:8d}" + ("  FLAT BATCH" if flat_pc > 30.0 and not a.old_window else ""))
        worst.append((0.0 if (flat_pc > 30.0 and not a.old_window) else pc(r["over_clean"]), sname, sc[0], err, cut, nf))
        assert r["mismatch"] == 0, "the int16 model disagrees with the oracle"
    worst.sort(reverse=True)
    emit(f"# {len(worst)} cells in {time.time() - t0:.0f} s; the ten worst by pairs started over although they do not z-drop (flat batches aside):")
    for w in worst[:10]:
        emit(f"#   {w[0]:6.2f} %  {w[1]} {w[2]} err {w[3]} cut {w[4]} N-run {w[5]}")
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
