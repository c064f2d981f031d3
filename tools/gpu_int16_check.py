"""Developer tool (GPU box): packed-int16 kernel vs oracle on a few workloads, with timing.  Run under `timeout`."""
import os, sys, time
os.environ.setdefault("AGATHA_AMD_FORCE_INT16", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agatha_amd import engine as E
from agatha_amd import workload as WL
from oracle import oracle as O


def run(name, qs, ts, prm_tuple, check=True):
    m, x, q, r, s, z, w = prm_tuple
    qb, qo, ql = WL.make_batch(qs)
    tb, to, tl = WL.make_batch(ts)
    eng = E.Engine(0)
    sc = E.Scores(m, x, q, r, s, z, w)
    t0 = time.time()
    got = eng.align_host_batch(qb, tb, qo, to, ql, tl, sc)
    dt = time.time() - t0
    cfg16 = eng.last_int16_config()
    msg = f"{name}: n={len(ql)} int16={cfg16} int32={eng.last_config()} wall={dt*1e3:.1f} ms"
    if check:
        exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(m, x, q, r, s, z, w), wide=True, model=O.MODEL_SLICES, threads=16)
        bad = [i for i in range(len(ql)) if any(int(exp[j][i]) != int(got[j][i]) for j in range(3))]
        msg += f" mismatches={len(bad)}"
        for i in bad[:5]:
            msg += f"\n   pair {i} Q={ql[i]} R={tl[i]} exp={[int(exp[j][i]) for j in range(3)]} got={[int(got[j][i]) for j in range(3)]}"
    print(msg, flush=True)


if __name__ == "__main__":
    rng = np.random.default_rng(5)
    qs, ts = WL.cfg_c1(n=64)
    run("c1-64 w751", qs, ts, (2, 4, 4, 2, 3, 400, 751))
    run("c1-64 w760 (t0=0)", qs, ts, (2, 4, 4, 2, 3, 400, 760))
    run("c1-64 w751 s1", qs, ts, (2, 4, 4, 2, 1, 400, 751))
    run("c1-64 w751 z-1", qs, ts, (2, 4, 4, 2, 3, -1, 751))
    qs, ts = WL.cfg_c4(n=300, lo=100, hi=20000)
    run("c4-300 w751", qs, ts, (2, 4, 4, 2, 3, 400, 751))
    run("c4-300 w760 m1x4", qs, ts, (1, 4, 6, 2, 2, 100, 760))
    qs, ts = WL.make_pairs(7, 200, lambda g: int(g.integers(1, 3000)), 0.05, 0.05, 0.05, n_rate=0.02)
    run("short ragged w751", qs, ts, (2, 4, 4, 2, 3, 400, 751))
    qs, ts = WL.cfg_c1(n=2000)
    run("c1-2000 w751", qs, ts, (2, 4, 4, 2, 3, 400, 751))
