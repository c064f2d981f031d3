"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PKEYS = ("m", "x", "q", "r", "s", "z", "w")


def load_kats():
    return json.load(open(os.path.join(GOLDEN, "kat_appendix_e.json")))


def load_ref_vectors():
    """Groups of the fixture written by tests/golden/gen_golden.py (reference kernel outputs)."""
    z = np.load(os.path.join(GOLDEN, "ref_vectors.npz"))
    out = []
    for name in z["names"]:
        name = str(name)
        p = dict(zip(PKEYS, (int(v) for v in z[name + "/params"])))
        out.append(dict(name=name, params=p, qbatch=z[name + "/qbatch"], tbatch=z[name + "/tbatch"],
                        qoff=z[name + "/qoff"], toff=z[name + "/toff"], qlen=z[name + "/qlen"],
                        tlen=z[name + "/tlen"], expect=z[name + "/expect"]))
    return out


def in_reference_domain(qlen, tlen, match):
    """Reference validity domain: lengths < 32768 and every H < 32768."""
    m = int(max(np.max(qlen), np.max(tlen)))
    return m < 32768 and match * int(min(np.max(qlen), np.max(tlen))) < 32768


def oracle_batch(g, **kw):
    return O.align_batch(g["qbatch"], g["tbatch"], g["qoff"], g["toff"], g["qlen"], g["tlen"],
                         O.make_params(**g["params"]), **kw)
