"""Developer tool (CPU only): randomised check of the int16 kernel's ARITHMETIC MODEL (oracle/agatha_lanes_model.c) against
the oracle: results, int16 range, zone separation, bail-out rate; and of its DECISION model (value steps, checkpoints, probation).  python tools/model16_fuzz.py [trials] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from tests.test_oracle import _mixed_pairs

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
SC = [(2, 4, 4, 2), (1, 4, 6, 2), (2, 3, 5, 1), (3, 5, 0, 1), (1, 1, 1, 1), (16, 32, 64, 16), (16, 0, 0, 16), (0, 32, 64, 0), (5, 4, 10, 16), (16, 32, 0, 0), (1, 19, 39, 3), (2, 32, 64, 2), (1, 32, 10, 1), (2, 24, 4, 2)]
gmin, gmax, garb, rmin, nb, npairs, bad = 0, 0, -10**9, 10**9, 0, 0, 0
for t in range(trials):
    w = int(rng.choice([16, 17, 18, 19, 20, 21, 22, 23, 24, 33, 47, 64, 100, 200, 333, 751]))
    z = int(rng.choice([-1, 0, 20, 100, 400, 2000]))
    s = int(rng.choice([1, 2, 3, 5, 8]))
    m, x, q, r = SC[int(rng.integers(0, len(SC)))]
    prm = O.make_params(m, x, q, r, s, z, w)
    qs, ts = _mixed_pairs(rng, 16, int(rng.choice([300, 2500, 6000])))
    qb, qo, ql = O.make_batch(qs); tb, to, tl = O.make_batch(ts)
    W = (w + 7) // 8
    G, S = [c for c in ((16, 2), (16, 4), (16, 6), (32, 4), (32, 6), (64, 4)) if c[0] * c[1] >= W + 1][0]
    e = O.align_batch(qb, tb, qo, to, ql, tl, prm, wide=True, model=O.MODEL_STEPS, threads=8)
    sc, qe, te, kind, st = O.lanes16_batch(qb, tb, qo, to, ql, tl, prm, G, S, threads=8)
    ok = np.array_equal(e[0], sc) and np.array_equal(e[1], qe) and np.array_equal(e[2], te) and (kind >= 0).all()
    if not ok:
        bad += 1
        print("MISMATCH trial", t, "w", w, "z", z, "s", s, "scores", (m, x, q, r), "pairs", np.nonzero((e[0] != sc) | (e[1] != qe) | (e[2] != te))[0][:5])
    nb += int((kind == 1).sum()); npairs += len(kind)
    # the DECISION model on the same batch (value steps with a random margin, checkpoints every 0 / 64 / 128 / 256 steps, probation on or
    # off -- round 5): whatever a pair goes through, its result is the oracle's
    import ctypes as C
    span_, prob_ = C.c_int.in_dll(O.lib(), "agatha_lanes16_ck_span"), C.c_int.in_dll(O.lib(), "agatha_lanes16_probation")
    slots_ = C.c_int.in_dll(O.lib(), "agatha_lanes16_ck_slots")
    slots_.value = int(rng.choice([2, 2, 4, 8]))          # (2 = the kernel's two slots; a ring of more is the what-if of DESIGN.md 6 item 0)
    lazy_, lazy_any_ = C.c_int.in_dll(O.lib(), "agatha_lanes16_lazy_max"), C.c_int.in_dll(O.lib(), "agatha_lanes16_lazy_any_shape")
    lazy_.value, lazy_any_.value = int(rng.choice([0, 3, 8, 8])), 1        # (lazy value steps, round 6: the kernel uses them where a wave holds one pair; their arithmetic is the same on every shape)
    span_.value, prob_.value, margin = int(rng.choice([0, 64, 128, 256])), int(rng.integers(0, 2)), int(rng.choice([1, 4, 12, 40]))
    sc2, qe2, te2, kind2, _ = O.lanes16_batch(qb, tb, qo, to, ql, tl, prm, G, S, threads=8, value_step_margin=margin)
    if not (np.array_equal(e[0], sc2) and np.array_equal(e[1], qe2) and np.array_equal(e[2], te2)):
        bad += 1
        print("MISMATCH (decisions) trial", t, "w", w, "z", z, "s", s, "scores", (m, x, q, r), "span", span_.value, "slots", slots_.value, "probation", prob_.value, "margin", margin,
              "pairs", np.nonzero((e[0] != sc2) | (e[1] != qe2) | (e[2] != te2))[0][:5])
    kinds_seen = kinds_seen + np.bincount(kind2, minlength=5)[:5] if "kinds_seen" in dir() else np.bincount(kind2, minlength=5)[:5]
    span_.value = prob_.value = 0; slots_.value = 2; lazy_.value, lazy_any_.value = 8, 0
    if (kind == 0).any():
        gmin = min(gmin, st[0]); gmax = max(gmax, st[1]); garb = max(garb, st[2]); rmin = min(rmin, st[3])
print("trials", trials, "mismatching trials", bad, "pairs", npairs, "bailed", nb)
print("decision model: pairs as they came / refused / from their first step / back to a checkpoint / back and over again:", [int(v) for v in kinds_seen])
print("rep range [%d, %d]  largest garbage %d  smallest in-band %d" % (gmin, gmax, garb, rmin))
