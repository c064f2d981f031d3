"""Developer tool (GPU box): randomised parity sweep on LONG reads (10-80 kb: chains of thousands of steps), the regime of the one-pair-per-wave
shapes <64, P> and <128, 1> -- rebases, checkpoints at their real spans, going back to them, probation, lazy value steps, the E hand-off in
registers and its passage through LDS at every save -- which tools/gpu_fuzz.py (reads of up to 4 kb) only touches with forced options.

    python tools/gpu_fuzz_long.py [seconds] [seed]

Every trial: 24-96 pairs of one length class, a mix of clean, noisy (10-15 % errors), broken (an unrelated tail from a random point on) and
bursty reads (a 150-500-base burst of errors), scoring / z / band drawn at random from what a mapper uses, lazy_max and ck_min_steps varied;
the device's own choice of shape and every int16 candidate forced in turn against the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload as WL
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = agatha_amd.Engine(0)
t_end = time.time() + budget
trials = runs = bad = 0
seen = {}
while time.time() < t_end:
    w = int(rng.choice([751, 751, 500, 1000, 1500, 2000]))
    m, x, q, r = [(2, 4, 4, 2), (1, 4, 6, 2), (2, 8, 12, 2), (1, 19, 39, 3), (3, 5, 0, 1)][int(rng.integers(0, 5))]
    z = int(rng.choice([400, 400, 120, 1000, -1]))
    p = dict(m=m, x=x, q=q, r=r, s=int(rng.choice([1, 3])), z=z, w=w)
    lo, hi = [(10000, 20000), (20000, 40000), (40000, 80000)][int(rng.integers(0, 3))]
    n = int(rng.choice([24, 48, 96])) if hi <= 40000 else int(rng.choice([12, 24]))
    qs, ts = [], []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        ref = WL.random_seq(rng, L)
        mode = int(rng.integers(0, 5))
        if mode == 0: rd = WL.mutate(rng, ref, 0.03, 0.03, 0.04)
        elif mode == 1: rd = WL.mutate(rng, ref, 0.06, 0.04, 0.04)
        elif mode == 2:
            bp = int(rng.integers(L // 10, L)); rd = np.concatenate([WL.mutate(rng, ref[:bp], 0.03, 0.03, 0.04), WL.random_seq(rng, L - bp)])
        elif mode == 3:
            blen = int(rng.choice([150, 250, 350, 500])); at = int(rng.integers(L // 5, L * 4 // 5 - blen))
            a = WL.mutate(rng, ref, 0.03, 0.03, 0.04); at = min(at, max(0, len(a) - blen - 1))
            rd = np.concatenate([a[:at], WL.mutate(rng, a[at:at + blen], 0.15, 0.12, 0.13), a[at + blen:]])
        else:
            k = int(rng.integers(1, w)); pos = int(rng.integers(0, L))
            rd = np.concatenate([ref[:pos], WL.random_seq(rng, k), ref[pos:]]) if rng.random() < 0.5 else np.concatenate([ref[:pos], ref[min(L, pos + k):]])
            rd = WL.mutate(rng, rd, 0.02, 0.02, 0.02)
        if rd.size == 0: rd = WL.random_seq(rng, 1)
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    qb, qo, ql = WL.make_batch(qs); tb, to, tl = WL.make_batch(ts)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=16)
    vs = dict(lazy_max=int(rng.choice([0, 2, 8, 8])), ck_min_steps=int(rng.choice([384, 384, 64, 100000])), fast_margin=int(rng.choice([12, 12, 3, 40])))
    for mode, opts in (("choice", dict(vs)), ("cand0", dict(vs, force_int16=1, force_choice=0)), ("cand1", dict(vs, force_int16=1, force_choice=1)), ("cand2", dict(vs, force_int16=1, force_choice=2))):
        try:
            with agatha_amd.debug_options(**opts):
                b = eng.batch(qb, tb, qo, to, ql, tl)
                try:
                    b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p)); b.download(); eng.synchronize()
                    got = [b.res_host[j].copy() for j in range(3)]
                    ch, st = b.kernel_choice(), b.step_stats()
                finally:
                    b.free()
        except agatha_amd.AgathaError:
            continue                                   # (a candidate that does not exist for this band)
        runs += 1
        key = (ch[0], ch[1], ch[2])
        s_ = seen.setdefault(key, [0, 0, 0, 0]); s_[0] += 1; s_[1] += int(st[2]); s_[2] += int(st[15]); s_[3] += int(st[23])
        diff = [i for i in range(len(ql)) if any(int(exp[j][i]) != int(got[j][i]) for j in range(3))]
        if diff:
            bad += 1
            i = diff[0]
            print("MISMATCH", mode, p, vs, "n", len(ql), "choice", ch, "pairs", diff[:6], "first: Q", int(ql[i]), "R", int(tl[i]),
                  "exp", [int(exp[j][i]) for j in range(3)], "got", [int(got[j][i]) for j in range(3)], flush=True)
    trials += 1
print("long-read fuzz trials", trials, "runs", runs, "mismatching runs", bad)
for k, v in sorted(seen.items()): print("  shape", k, "runs", v[0], "pairs started over", v[1], "back to a checkpoint", v[2], "lazy value wave-steps", v[3])
