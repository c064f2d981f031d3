#!/usr/bin/env python3
"""Developer tool: kernel-only throughput of the BASELINE.json workload shapes on one GPU (not the bench contract)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import agatha_amd  # noqa: E402
from agatha_amd import shard  # noqa: E402
from agatha_amd import workload as synth  # noqa: E402
from agatha_amd import workload as O  # noqa: E402  (make_batch only)

eng = agatha_amd.Engine(0)
out = {}
cfgs = [
    ("C0 3kb w751", lambda: synth.cfg_c0(n=20000), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)),
    ("C1 10kb w751", lambda: synth.cfg_c1(n=10000), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)),
    ("C2 HiFi 15-20kb w500 m1", lambda: synth.cfg_c2(n=12000), dict(m=1, x=4, q=6, r=2, s=3, z=400, w=500)),
    ("C3 100kb w1500", lambda: synth.cfg_c3(n=256), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=1500)),
    ("C4 mixed 1-100kb zdrop", lambda: synth.cfg_c4(n=6000), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)),
]
for name, gen, p in cfgs:
    t0 = time.time()
    qs, ts = gen()
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    del qs, ts
    cells = int(shard.nominal_cells(ql, tl, p["w"]).sum())
    b = eng.batch(qb, tb, qo, to, ql, tl)
    b.upload(); b.pack(); eng.synchronize()
    sc = agatha_amd.Scores.make(**p)
    ms = []
    for rep in range(3):
        e0, e1 = eng.event(), eng.event()
        eng.record(e0); b.align(sc); eng.record(e1)
        ms.append(eng.elapsed_ms(e0, e1))
    b.download(); eng.synchronize()
    best = min(ms)
    out[name] = dict(pairs=len(ql), nominal_cells=cells, ms=best, gcups_nominal=cells / best / 1e6,
                     pairs_per_s=len(ql) / best * 1e3, cfg=eng.last_config(), choice=b.kernel_choice(),
                     kinds=b.pair_kinds(), gen_s=round(time.time() - t0, 1))
    print(name, out[name], flush=True)
    b.free()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_configs.json"), "w"), indent=1)
