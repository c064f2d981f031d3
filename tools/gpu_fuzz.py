"""Developer tool (GPU box): randomised parity sweep of all kernels against the oracle.

    python tools/gpu_fuzz.py [seconds] [seed]

Every trial draws scoring parameters, band, slice width and z-drop at random, builds a mixed batch (similar pairs, noisy
pairs, unrelated pairs, broken pairs, large indels, ragged lengths, Ns) and aligns it three ways: device's choice,
int16 kernel forced, int32 kernels only.  Any difference from the oracle is printed with the parameters that caused it.
"""
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
trials = bad = tb_runs = 0
while time.time() < t_end:
    w = int(rng.choice([16, 17, 30, 64, 100, 250, 500, 751, 760, 1000, 1500, int(rng.integers(16, 1700))]))
    m = int(rng.choice([1, 2, 3, 5, 16])); x = int(rng.choice([1, 3, 4, 6, 9, 32])); q = int(rng.choice([0, 1, 4, 6, 20, 64])); r = int(rng.choice([1, 2, 3, 16]))
    s = int(rng.choice([1, 2, 3, 4, 7, 20])); z = int(rng.choice([-1, 0, 1, 50, 400, 5000]))
    p = dict(m=m, x=x, q=q, r=r, s=s, z=z, w=w)
    lmax = int(rng.choice([300, 1500, 4000]))
    qs, ts = [], []
    for _ in range(int(rng.choice([8, 40, 80]))):
        L = int(rng.integers(1, lmax))
        ref = WL.random_seq(rng, L)
        mode = int(rng.integers(0, 6))
        if mode == 0: rd = WL.mutate(rng, ref, 0.03, 0.03, 0.04)
        elif mode == 1: rd = WL.mutate(rng, ref, 0.15, 0.1, 0.1)
        elif mode == 2: rd = WL.random_seq(rng, int(rng.integers(1, lmax)))
        elif mode == 3:
            bp = int(rng.integers(0, L)); rd = np.concatenate([WL.mutate(rng, ref[:bp], 0.02, 0.02, 0.02), WL.random_seq(rng, int(rng.integers(1, 800)))])
        elif mode == 4:
            k = int(rng.integers(0, min(L, 2 * w) + 1)); pos = int(rng.integers(0, L))
            rd = np.concatenate([ref[:pos], WL.random_seq(rng, k), ref[pos:]]) if rng.random() < 0.5 else np.concatenate([ref[:pos], ref[min(L, pos + k):]])
        else:
            rd = ref.copy()
        if rd.size == 0: rd = WL.random_seq(rng, 1)
        if rng.random() < 0.15:
            rd = rd.copy(); rd[rng.random(rd.size) < 0.03] = ord("N")
        if rng.random() < 0.1:
            ref = ref.copy(); ref[rng.random(ref.size) < 0.03] = ord("N")
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    qb, qo, ql = WL.make_batch(qs); tb, to, tl = WL.make_batch(ts)
    if os.environ.get("FUZZ_ONLY") and trials != int(os.environ["FUZZ_ONLY"]):
        vs = dict(fast_margin=int(rng.choice([0, 1, 3, 8, 16, 64])), ck_min_steps=int(rng.choice([0, 16, 256, 4096])))
        [rng.integers(0, 2) for _ in range(3)]
        trials += 1
        continue
    exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=16)
    # (value steps of the int16 kernel: the window of key steps at a pair's end and the checkpoints vary as well)
    vs = dict(fast_margin=int(rng.choice([0, 1, 3, 8, 16, 64])), ck_min_steps=int(rng.choice([0, 16, 256, 4096])))
    vs["lazy_max"] = (0, 1, 8, 8)[(trials >> 1) & 3]      # (lazy value steps of the one-pair-per-wave shapes, round 6)
    vs["fast_anchor"] = trials & 1            # (where the window of key steps is anchored: the shorter sequence's corner / the pair's end)
    for k_, v_ in os.environ.items():          # (tools/fuzz_one.py: override an option / a parameter of the trial)
        if k_.startswith("FUZZ_FORCE_"): vs[k_[11:].lower()] = int(v_)
        if k_.startswith("FUZZ_PARAM_"): p[k_[11:].lower()] = int(v_)
    if any(k_.startswith("FUZZ_PARAM_") for k_ in os.environ):
        exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=16)
    for mode, opts in (("choice", dict(vs)), ("int16", dict(vs, force_int16=1)), ("int32", {"no_int16": 1})):
        if os.environ.get("FUZZ_VERBOSE"): print("trial", trials, mode, p, vs, "n", len(ql), "lmax", lmax, flush=True)
        with agatha_amd.debug_options(**opts):
            hint = bool(rng.integers(0, 2))
            got = eng.align_host_batch(qb, tb, qo, to, ql, tl, agatha_amd.Scores.make(**p), use_len_hint=hint)
        diff = [i for i in range(len(ql)) if any(int(exp[j][i]) != int(got[j][i]) for j in range(3))]
        if diff:
            bad += 1
            i = diff[0]
            print("MISMATCH", mode, p, "n", len(ql), "pairs", diff[:6], "first: Q", int(ql[i]), "R", int(tl[i]),
                  "exp", [int(exp[j][i]) for j in range(3)], "got", [int(got[j][i]) for j in range(3)], "int16cfg", eng.last_int16_config(), flush=True)
            if os.environ.get("FUZZ_VERBOSE"):
                with agatha_amd.debug_options(**opts):
                    b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p), use_len_hint=hint); b.download(); eng.synchronize()
                    print("   hint", hint, "choice", b.kernel_choice(), "again:", [int(b.res_host[j][i]) for j in range(3)], "step_stats", b.step_stats(), flush=True); b.free()
    if trials % int(os.environ.get("FUZZ_TB_EVERY", "3")) == 0:
        # the traceback pass on the same batch: scores, ends and every path byte against the oracle's walk
        es, eq, et, ecig, en = O.traceback_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), threads=16)
        b = eng.batch(qb, tb, qo, to, ql, tl)
        try:
            b.upload(); b.pack()
            gs, gq, gt, gc = b.align_traceback(agatha_amd.Scores.make(**p))
        finally:
            b.free()
        off = qo.astype(np.int64) + to.astype(np.int64)
        diff = [i for i in range(len(ql)) if (int(gs[i]), int(gq[i]), int(gt[i])) != (int(es[i]), int(eq[i]), int(et[i])) or
                gc[i] != (None if en[i] < 0 else ecig[off[i]:off[i] + en[i]].tobytes())]
        tb_runs += 1
        if diff:
            bad += 1
            i = diff[0]
            print("TRACEBACK MISMATCH", p, "n", len(ql), "pairs", diff[:6], "first: Q", int(ql[i]), "R", int(tl[i]), "exp", int(es[i]), int(eq[i]),
                  int(et[i]), int(en[i]), "got", int(gs[i]), int(gq[i]), int(gt[i]), None if gc[i] is None else len(gc[i]), flush=True)
    trials += 1
print("fuzz trials", trials, "(traceback on", tb_runs, "of them) mismatching runs", bad)
