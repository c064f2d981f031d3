#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors.npz: seeded inputs + the outputs of the REFERENCE agatha_kernel executed
under the CPU warp emulator (oracle/ref_shim, `make -C oracle ref`; needs /root/reference, build container only).

The fixture is data only: ASCII sequences in the GASAL host-batch wire format, offsets, lengths, the seven
scoring/band parameters, and the three int32 result arrays the reference produced.  Run from the repo root:
    python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle as O, synth  # noqa: E402

assert O.have_ref(), "build the shim first: make -C oracle ref"

groups = []


def add(name, qs, ts, **params):
    P = O.make_params(**params)
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    s, q, t = O.ref_align_batch(qb, tb, qo, to, ql, tl, P)
    # the plain-C oracle must agree before anything is written
    s2, q2, t2 = O.align_batch(qb, tb, qo, to, ql, tl, P, wide=False, model=O.MODEL_SLICES)
    assert (s == s2).all() and (q == q2).all() and (t == t2).all(), name
    groups.append((name, params, qb, tb, qo, to, ql, tl, np.stack([s, q, t])))
    print(f"{name}: n={len(ql)} maxlen={max(ql.max(), tl.max())} score range {s.min()}..{s.max()}")


rng = np.random.default_rng(20241002)
A = dict(m=2, x=4, q=4, r=2)
B = dict(m=1, x=4, q=6, r=2)
# small bands (block-granular band / lost-diagonal quirks), mixed z, short reads, with Ns
for w in (5, 8, 9, 13, 16, 17, 24, 33, 40):
    for sc, scn in ((A, "A"), (B, "B")):
        z = int(rng.choice([-1, 0, 20, 60, 400]))
        e = float(rng.uniform(0.02, 0.12))
        qs, ts = synth.make_pairs(int(rng.integers(1 << 30)), 40, lambda r: int(np.exp(r.uniform(0, np.log(400)))),
                                  e, e, e, n_rate=0.01)
        add(f"small_w{w}_{scn}_z{z}", qs, ts, s=3, z=z, w=w, **sc)
# medium reads, band 100 / 751, both scoring sets, z-drop firing (broken pairs)
for w, sc, scn, z in ((100, A, "A", 100), (751, A, "A", 400), (751, B, "B", 400), (100, B, "B", 50)):
    qs, ts = synth.cfg_c4(n=48, seed=int(rng.integers(1 << 30)), lo=100, hi=3000)
    add(f"mixed_w{w}_{scn}_z{z}", qs, ts, s=3, z=z, w=w, **sc)
# length-asymmetric pairs (band leaves the matrix: empty-slice stop rule)
qs, ts = synth.make_pairs(77, 32, lambda r: int(r.integers(200, 1200)), 0.03, 0.03, 0.03)
ts = [t[: max(1, len(t) // 3)] for t in ts]
add("asym_w40_A", qs, ts, s=3, z=400, w=40, **A)
add("asym_w40_A_swapped", ts, qs, s=3, z=400, w=40, **A)
# headline shape: ~10 kb ONT-like, band 751, z 400 (BASELINE.json configs[1]), a handful of pairs
qs, ts = synth.cfg_c1(n=6, seed=0xA6A70001)
add("c1_ont10k_w751_A", qs, ts, s=3, z=400, w=751, **A)
qs, ts = synth.cfg_c1(n=4, seed=0xA6A70011)
add("c1_ont10k_w751_B", qs, ts, s=3, z=400, w=751, **B)
# HiFi-like 15-20 kb at m=1 (in the reference's 16-bit domain), band 500
qs, ts = synth.cfg_c2(n=3, seed=0xA6A70002)
add("c2_hifi_w500_B", qs, ts, s=3, z=400, w=500, **B)

out = {}
names = []
for name, params, qb, tb, qo, to, ql, tl, res in groups:
    names.append(name)
    out[name + "/params"] = np.array([params[k] for k in ("m", "x", "q", "r", "s", "z", "w")], np.int32)
    out[name + "/qbatch"], out[name + "/tbatch"] = qb, tb
    out[name + "/qoff"], out[name + "/toff"], out[name + "/qlen"], out[name + "/tlen"] = qo, to, ql, tl
    out[name + "/expect"] = res
out["names"] = np.array(names)
path = os.path.join(os.path.dirname(__file__), "ref_vectors.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes;", sum(len(g[6]) for g in groups), "pairs")
