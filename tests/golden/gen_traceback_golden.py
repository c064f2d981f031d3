#!/usr/bin/env python3
"""Generate tests/golden/traceback_paths.json: a few dozen small pairs with the scores, end cells and alignment paths
(CIGAR text) of oracle/agatha_oracle.c: agatha_model_traceback.  The reference never fills cigar / n_cigar_ops
(gasal.h:91-92, res.cpp:27-28), so there is no reference output to record: this pins the DEFINITION (tie-breaks: diagonal >
E > F, open before extend; gaps open from a cell's diagonal term; the no-path rule) against later changes of the oracle and
of the kernels.  Data only: ASCII sequences, parameters, expected results.  Run from the repo root:
    python tests/golden/gen_traceback_golden.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle as O, synth  # noqa: E402


def text(c):
    if c is None:
        return "!"
    if not c:
        return "*"
    out, run, op = [], 0, c[0] & 3
    for b in c:
        if (b & 3) != op:
            out.append(f"{run}{'=XDI'[op]}")
            run, op = 0, b & 3
        run += b >> 2
    out.append(f"{run}{'=XDI'[op]}")
    return "".join(out)


rng = np.random.default_rng(0x7B7B)
cases = []
for p in (dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751), dict(m=1, x=4, q=6, r=2, s=3, z=100, w=40), dict(m=2, x=4, q=4, r=2, s=1, z=-1, w=3),
          dict(m=5, x=4, q=10, r=1, s=7, z=60, w=16)):
    qs, ts = [], []
    for k in range(12):
        ln = int(rng.integers(1, 260))
        q = synth.random_seq(rng, ln)
        t = synth.mutate(rng, q, 0.06, 0.05, 0.05)
        if k % 4 == 3:                                              # unrelated tail: z-drop / early end
            t = np.concatenate([t[:len(t) // 2], synth.random_seq(rng, int(rng.integers(1, 120)))])
        if k % 5 == 4:
            q = q.copy(); q[rng.integers(0, ln)] = ord("N")
        qs.append(bytes(q)); ts.append(bytes(t))
    s, qe, te, cig = O.traceback_pairs(qs, ts, O.make_params(**p))
    cases.append({"params": p, "queries": [q.decode() for q in qs], "targets": [t.decode() for t in ts],
                  "score": [int(v) for v in s], "query_end": [int(v) for v in qe], "target_end": [int(v) for v in te],
                  "cigar": [text(c) for c in cig], "bytes": [None if c is None else c.hex() for c in cig]})
json.dump({"source": "oracle/agatha_oracle.c: agatha_model_traceback", "cases": cases},
          open(os.path.join(os.path.dirname(__file__), "traceback_paths.json"), "w"))
print(sum(len(c["queries"]) for c in cases), "pairs")
