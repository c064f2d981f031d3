import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import agatha_amd
from agatha_amd import workload
eng = agatha_amd.Engine(0)
qs, ts = workload.cfg_c1(n=10000)
qb, qo, ql = workload.make_batch(qs); tb, to, tl = workload.make_batch(ts)
b = eng.batch(qb, tb, qo, to, ql, tl); b.upload(); b.pack(); eng.synchronize()
sc = agatha_amd.Scores.make()
for rep in range(2):
    b.align(sc); eng.synchronize()
print("stats", list(b.step_stats()))
print("kinds", b.pair_kinds(), "sched", b.schedule_info())
b.free()
