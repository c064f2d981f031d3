"""CPU only: per-pair step statistics of the BASELINE workload shapes (oracle/agatha_oracle.c: agatha_steps_stats_batch) --
how far before a pair's end the running maximum rises for the last time, and on how many steps z-drop could fire judging
by the anti-diagonal maxima alone.  The numbers behind the margin of the int16 kernel's value-only fast path."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from agatha_amd import workload as wl

lib = oracle.lib()
lib.agatha_steps_stats_batch.argtypes = [C.c_void_p] * 6 + [C.c_int, C.POINTER(oracle.Params), C.c_int, C.c_void_p]
lib.agatha_steps_stats_batch.restype = None


def stats(qs, ts, params, threads=8):
    qb, qo, ql = wl.make_batch(qs); tb, to, tl = wl.make_batch(ts)
    n = len(ql)
    st = np.zeros((n, 7), np.int32)
    lib.agatha_steps_stats_batch(qb.ctypes.data, tb.ctypes.data, qo.ctypes.data, to.ctypes.data, ql.ctypes.data, tl.ctypes.data,
                                 n, C.byref(params), threads, st.ctypes.data)
    return st


def report(name, st):
    run, last, nz, first, zd, raises, total = st.T
    tail = run - 1 - last                       # steps between the last rise and the last step run
    print(f"{name}: pairs {len(st)}  steps/pair {run.mean():.0f} of {total.mean():.0f}  z-dropped {zd.mean()*100:.1f}%  "
          f"steps that raise the maximum {100*raises.sum()/run.sum():.1f}%")
    for m in (8, 16, 32, 64, 128):
        print(f"   last rise more than {m:3d} steps before the end: {100*np.mean((tail > m) & (last >= 0)):.2f}% of pairs"
              f"  (not z-dropped: {100*np.mean((tail > m) & (last >= 0) & (zd == 0)):.2f}%)")
    print(f"   pairs with a step on which z-drop could fire by value: {100*np.mean(nz > 0):.1f}%; such steps {100*nz.sum()/run.sum():.3f}% of all;"
          f" mean steps from the first one to the end {np.mean((run - first)[first >= 0]) if (first >= 0).any() else 0:.1f}")
    never = (last < 0)
    print(f"   pairs whose maximum never rises: {100*never.mean():.2f}%")


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    report("C1", stats(*wl.cfg_c1(n), oracle.make_params(w=751)))
    report("C0", stats(*wl.cfg_c0(n), oracle.make_params(w=751)))
    report("C2", stats(*wl.cfg_c2(n // 2), oracle.make_params(m=1, x=4, q=6, r=2, w=500)))
    report("C4", stats(*wl.cfg_c4(n, hi=30000), oracle.make_params(w=751)))
