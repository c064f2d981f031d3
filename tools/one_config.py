"""Developer tool (GPU box): one BASELINE workload shape, kernel-only, a few launches (for rocprofv3 --pmc runs).
    python3 tools/one_config.py C0|C1|C2|C3|C4 [pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agatha_amd
from agatha_amd import workload as synth, shard
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
gen, p = {"C3": (lambda n: synth.cfg_c3(n=n or 256), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=1500)),
          "C4": (lambda n: synth.cfg_c4(n=n or 6000), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)),
          "C1": (lambda n: synth.cfg_c1(n=n or 10000), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)),
          "C0": (lambda n: synth.cfg_c0(n=n or 20000), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)),
          "C2": (lambda n: synth.cfg_c2(n=n or 12500), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=500))}[name]
qs, ts = gen(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
qb, qo, ql = synth.make_batch(qs); tb, to, tl = synth.make_batch(ts)
cells = int(shard.nominal_cells(ql, tl, p["w"]).sum())
steps = int(((ql + 7) // 8 + (tl + 7) // 8).sum())
eng = agatha_amd.Engine(0)
b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
sc = agatha_amd.Scores.make(**p)
for rep in range(3):
    e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms = eng.elapsed_ms(e0, e1)
    print(name, b.kernel_choice(), round(ms, 2), "ms", round(cells / ms / 1e6, 1), "GCUPS", "pair-steps", steps, "cells", cells, "steps", b.step_stats(), "kinds", b.pair_kinds(), flush=True)
b.free()
