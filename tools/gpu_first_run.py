#!/usr/bin/env python3
"""Developer tool: quick GPU diagnostics (parity on golden vectors with per-pair mismatch dump + a first timing)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import agatha_amd  # noqa: E402
from oracle import oracle as O, synth  # noqa: E402
from helpers import load_ref_vectors, load_kats  # noqa: E402

out = {}
eng = agatha_amd.Engine(0)
print("lib", eng.lib.agatha_amd_version())
bad_total = 0
for g in load_ref_vectors():
    s, q, t = eng.align_host_batch(g["qbatch"], g["tbatch"], g["qoff"], g["toff"], g["qlen"], g["tlen"],
                                   agatha_amd.Scores.make(**g["params"]))
    exp = g["expect"]
    bad = np.nonzero((s != exp[0]) | (q != exp[1]) | (t != exp[2]))[0]
    bad_total += bad.size
    print(g["name"], "cfg", eng.last_config(), "n", len(s), "bad", bad.size)
    for k in bad[:5]:
        print("   pair", k, "Q", g["qlen"][k], "R", g["tlen"][k], "got", (s[k], q[k], t[k]), "exp", tuple(exp[:, k]))
print("TOTAL BAD", bad_total)
out["golden_bad"] = int(bad_total)

# timing on a slice of config C1
n = int(os.environ.get("N_PAIRS", "2048"))
qs, ts = synth.cfg_c1(n=n)
qb, qo, ql = O.make_batch(qs)
tb, to, tl = O.make_batch(ts)
sc = agatha_amd.Scores.make()
b = eng.batch(qb, tb, qo, to, ql, tl)
b.upload(); b.pack(); eng.synchronize()
cells = O.nominal_cells_np(ql, tl, 751)
for rep in range(3):
    e0, e1 = eng.event(), eng.event()
    eng.record(e0); b.align(sc); eng.record(e1)
    ms = eng.elapsed_ms(e0, e1)
    print(f"align n={n} {ms:.2f} ms  {cells / ms / 1e6:.1f} GCUPS cfg={eng.last_config()}")
out["c1_n"] = n; out["c1_ms"] = ms; out["c1_gcups"] = cells / ms / 1e6
b.download(); eng.synchronize()
t0 = time.time()
k = min(n, 24)
exp = O.align_batch(qb, tb, qo[:k], to[:k], ql[:k], tl[:k], O.make_params(), wide=True, model=0, threads=8)
dt = time.time() - t0
ok = all((b.res_host[j][:k] == exp[j]).all() for j in range(3))
print("c1 subset parity", ok, f"oracle {k} pairs in {dt:.1f}s")
out["c1_parity"] = bool(ok)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "first_run.json"), "w"))
