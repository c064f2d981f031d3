"""CPU stand-in for agatha_amd's Engine / DeviceBatch / Scores with the interface bench.py uses (tests/test_bench_strong_leg.py):
"aligning" a pair gives a triple that depends only on the pair's own lengths and first bases, so every rank -- and a single rank
that sees the whole batch -- computes the same numbers.  Nothing here aligns anything; what is under test is bench.py's N > 1
plumbing (shards, collectives, the strong leg, the JSON line)."""
import ctypes as C
import time

import numpy as np


class Scores:
    @classmethod
    def make(cls, **kw):
        return cls()


def _fake(qb, tb, qo, to, ql, tl):
    ql, tl = np.asarray(ql, np.int64), np.asarray(tl, np.int64)
    q0 = np.asarray(qb)[np.asarray(qo, np.int64)].astype(np.int64)
    t0 = np.asarray(tb)[np.asarray(to, np.int64)].astype(np.int64)
    return ((3 * ql + tl + q0 + 7 * t0) % 100003).astype(np.int32), (ql - 1).astype(np.int32), (tl - 1).astype(np.int32)


class Batch:
    def __init__(self, host):
        self.host = host
        self.n = len(host[4])
        self.ptrs = None
        self.res_host = [np.zeros(self.n, np.int32) for _ in range(3)]
        self.res = None

    def use_result_pointers(self, ptrs):
        self.ptrs = list(ptrs)

    def upload(self, stream=None): pass
    def pack(self, stream=None): pass

    def align(self, scores, stream=None):
        self.res = _fake(*self.host)
        if self.ptrs:
            for p, r in zip(self.ptrs, self.res):
                C.memmove(p, r.ctypes.data, 4 * self.n)

    def download(self, stream=None):
        for j in range(3):
            self.res_host[j][:] = self.res[j]

    def pair_kinds(self, stream=None): return (self.n, 0, 0)
    def schedule_info(self, stream=None): return (0, 0, 0)
    def step_stats(self, stream=None): return [0] * 40
    def kernel_choice(self, stream=None): return ("int16", 16, 4)
    def free(self): pass


class Engine:
    def __init__(self, device=0): pass
    def batch(self, *host): return Batch(host)
    def synchronize(self): pass
    def event(self): return [0.0]
    def set_kernel_events(self, e0, e1):
        if e0 is not None:
            e0[0] = time.perf_counter(); e1[0] = e0[0] + 1e-3
    def elapsed_ms(self, e0, e1): return (e1[0] - e0[0]) * 1e3
    def last_config(self): return (32, 2)
    def align_host_batch(self, qb, tb, qo, to, ql, tl, scores): return _fake(qb, tb, qo, to, ql, tl)
