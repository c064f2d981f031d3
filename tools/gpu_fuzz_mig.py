"""Developer tool (GPU box): randomised parity sweep of the preemptive static schedule of the packed-int16 kernel (pairs that
are suspended by one lane group and resumed by another) against the oracle.

    python tools/gpu_fuzz_mig.py [seconds] [seed]

Every trial draws scores, band, slice width, z-drop and a batch of 8 200 ... 40 000 mostly short pairs (similar, noisy, broken,
unrelated; Ns; a few long ones) -- more pairs than lane groups, so the schedule is in force -- and now and then forces the
take-over path (odd lane groups start late, short timeout).  The same batch is also run on the work queue."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload as WL
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = agatha_amd.Engine(0)
t_end = time.time() + budget
trials = bad = used = 0
while time.time() < t_end:
    w = int(rng.choice([16, 17, 24, 30, 40, 64, 100, 250, 500, int(rng.integers(16, 520))]))
    m = int(rng.choice([1, 2, 3, 5])); x = int(rng.choice([1, 3, 4, 6, 9])); q = int(rng.choice([0, 1, 4, 6, 20])); r = int(rng.choice([1, 2, 3]))
    s = int(rng.choice([1, 2, 3, 4, 7])); z = int(rng.choice([-1, 0, 20, 100, 400]))
    p = dict(m=m, x=x, q=q, r=r, s=s, z=z, w=w)
    n = int(rng.integers(8200, 40000))
    lmax = int(rng.choice([60, 200, 500, 1200]))
    qs, ts = [], []
    for k in range(n):
        L = int(rng.integers(1, lmax)) if rng.random() > 0.002 else int(rng.integers(2000, 9000))
        ref = WL.random_seq(rng, L)
        mode = int(rng.integers(0, 5))
        if mode == 0: rd = WL.mutate(rng, ref, 0.03, 0.03, 0.04)
        elif mode == 1: rd = WL.mutate(rng, ref, 0.15, 0.1, 0.1)
        elif mode == 2: rd = WL.random_seq(rng, int(rng.integers(1, lmax)))
        elif mode == 3:
            bp = int(rng.integers(0, L)); rd = np.concatenate([WL.mutate(rng, ref[:bp], 0.02, 0.02, 0.02), WL.random_seq(rng, int(rng.integers(1, lmax)))])
        else: rd = ref.copy()
        if rd.size == 0: rd = WL.random_seq(rng, 1)
        if rng.random() < 0.02:
            ref = ref.copy(); ref[rng.integers(0, ref.size)] = ord("N")
        if rng.random() < 0.01:
            rd = rd.copy(); rd[rng.integers(0, rd.size)] = ord("R")
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    qb, qo, ql = WL.make_batch(qs); tb, to, tl = WL.make_batch(ts)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=16)
    force_takeover = rng.random() < 0.25
    # (no_migrate = -1: the static schedule whenever it is possible -- since round 4 the device does not choose it for a batch whose
    #  longest pair is several times the average one, which these batches with their few long pairs are)
    for mode, opts in (("static", dict(no_migrate=-1, mig_timeout_us=500, mig_fresh_timeout_us=int(rng.choice([100, 500, 2000])), mig_test_delay_us=int(rng.integers(2000, 30000))) if force_takeover else dict(no_migrate=-1)),
                       ("queue", dict(no_migrate=1))):
        vs = dict(fast_margin=int(rng.choice([0, 2, 16, 40])), ck_min_steps=int(rng.choice([0, 16, 4096])))       # value steps, checkpoints
        vs["static_ck"] = int(rng.integers(0, 2)); vs["fast_anchor"] = int(rng.integers(0, 2))
        with agatha_amd.debug_options(force_int16=1, **vs, **opts):
            b = eng.batch(qb, tb, qo, to, ql, tl)
            try:
                b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p)); b.download(); eng.synchronize()
                got = [b.res_host[j].copy() for j in range(3)]
                info = b.schedule_info(); choice = b.kernel_choice()
            finally:
                b.free()
        if mode == "static" and info[0]: used += 1
        diff = [i for i in range(len(ql)) if any(int(exp[j][i]) != int(got[j][i]) for j in range(3))]
        if diff:
            bad += 1
            i = diff[0]
            print("MISMATCH", mode, p, "n", n, "takeover", force_takeover, info, choice, "pairs", diff[:6], "first: Q", int(ql[i]), "R", int(tl[i]),
                  "exp", [int(exp[j][i]) for j in range(3)], "got", [int(got[j][i]) for j in range(3)], flush=True)
    trials += 1
print("migration fuzz trials", trials, "with the static schedule in force", used, "mismatching runs", bad)
