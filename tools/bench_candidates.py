import os, sys, time, numpy as np
sys.path.insert(0,'/root/repo' if os.path.isdir('/root/repo') else '.')
sys.path.insert(0, os.getcwd())
import agatha_amd
from agatha_amd import workload as synth, shard
eng = agatha_amd.Engine(0)
for name, gen, p in (("C3", lambda: synth.cfg_c3(n=256), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=1500)),
                     ("C4", lambda: synth.cfg_c4(n=6000), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)),
                     ("C1-2000", lambda: synth.cfg_c1(n=2000), dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751))):
    qs, ts = gen()
    qb, qo, ql = synth.make_batch(qs); tb, to, tl = synth.make_batch(ts)
    cells = int(shard.nominal_cells(ql, tl, p["w"]).sum())
    b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
    sc = agatha_amd.Scores.make(**p)
    for c in ("", "0", "1", "2", "3"):
        agatha_amd.set_debug_option("force_choice", int(c) if c else -1)
        ms = []
        for rep in range(2):
            e0, e1 = eng.event(), eng.event(); eng.record(e0); b.align(sc); eng.record(e1); ms.append(eng.elapsed_ms(e0, e1))
        print(name, "force", c or "model", b.kernel_choice(), round(min(ms), 2), "ms", round(cells / min(ms) / 1e6, 1), "GCUPS", flush=True)
    b.free()
