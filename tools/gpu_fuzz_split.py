"""Developer tool (GPU box): randomised parity sweep of the SPLIT of a mixed batch between the two packed-int16 shapes (round 4):
every trial draws scores, band, z-drop and a batch of many short pairs with a few long ones among them (some broken, some with N),
runs it with the device's choice and with no_split = 1, demands identical results, and checks the long pairs and a sample of the
short ones against the oracle.      python tools/gpu_fuzz_split.py [seconds] [seed]"""
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
    w = int(rng.choice([100, 250, 500, 751, 760, int(rng.integers(64, 1500))]))
    m, x, q, r = [(2, 4, 4, 2), (1, 4, 6, 2), (2, 3, 5, 1), (3, 6, 4, 2)][int(rng.integers(0, 4))]
    p = dict(m=m, x=x, q=q, r=r, s=int(rng.choice([1, 3, 4])), z=int(rng.choice([-1, 100, 400])), w=w)
    n_short, n_long = int(rng.integers(5000, 22000)), int(rng.integers(1, 48))
    lmax = int(rng.choice([400, 1200, 3000]))
    qs, ts = [], []
    where = set(rng.choice(n_short + n_long, n_long, replace=False).tolist())
    for k in range(n_short + n_long):
        L = int(rng.integers(8000, 40000)) if k in where else int(rng.integers(30, lmax))
        ref = WL.random_seq(rng, L)
        u = rng.random()
        if u < 0.7: rd = WL.mutate(rng, ref, 0.03, 0.03, 0.04)
        elif u < 0.85: rd = np.concatenate([WL.mutate(rng, ref[:L // 2], 0.02, 0.02, 0.02), WL.random_seq(rng, L - L // 2)])
        else: rd = WL.mutate(rng, ref, 0.12, 0.1, 0.1)
        if rng.random() < 0.02:
            ref = ref.copy(); a = int(rng.integers(0, ref.size)); ref[a:a + 20] = ord("N")
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    qb, qo, ql = WL.make_batch(qs); tb, to, tl = WL.make_batch(ts)
    sc = agatha_amd.Scores.make(**p)
    res = {}
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload(); b.pack()
        forced = int(rng.choice([0, 0, 1, 7, 64, 300, 900]))          # (most batches are too small for the cost model to split them)
        for mode, opt in (("split", 0), ("one", 1)):
            agatha_amd.set_debug_option("no_split", opt)
            agatha_amd.set_debug_option("force_split", forced if mode == "split" else 0)
            b.align(sc); b.download(); eng.synchronize()
            res[mode] = [b.res_host[j].copy() for j in range(3)]
            if mode == "split":
                info = b.split_info(); choice = b.kernel_choice()
    finally:
        agatha_amd.set_debug_option("no_split", 0); agatha_amd.set_debug_option("force_split", 0)
        b.free()
    if info[0] > 0: used += 1
    same = all((a == c).all() for a, c in zip(res["split"], res["one"]))
    k = np.unique(np.concatenate([np.asarray(sorted(where)), rng.choice(len(qs), 300, replace=False)]))
    sb = WL.make_batch([qs[i] for i in k]), WL.make_batch([ts[i] for i in k])
    exp = O.align_batch(sb[0][0], sb[1][0], sb[0][1], sb[1][1], sb[0][2], sb[1][2], O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=16)
    ok = all((np.asarray(a)[k] == e).all() for a, e in zip(res["split"], exp))
    if not (same and ok):
        bad += 1
        print("MISMATCH", p, "n", n_short, n_long, "split", info, choice, "split == one shape:", same, "sample == oracle:", ok, flush=True)
    trials += 1
print("split fuzz trials", trials, "with a split", used, "mismatching", bad)
