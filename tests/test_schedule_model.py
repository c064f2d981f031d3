"""CPU-only: the preemptive static schedule of the packed-int16 kernel (schedule_kernel in agatha_amd/csrc/align_kernel.hip and
the acquisition loop of align16_body.inc), restated in a few lines of numpy and checked for the properties the device code
relies on.  The GPU tests (tests/test_gpu_int16.py::test_pairs_migrate_between_lane_groups ...) run the real thing; this
file pins the ARITHMETIC of the rule: McNaughton's wrap-around over the sorted pairs."""
import numpy as np
import pytest


def step_count(Q, R, sw):
    """Steps the kernel executes for a pair: a dry step, whole slices of `sw` block anti-diagonals, the final check step."""
    total = (Q + 7) // 8 + (R + 7) // 8 - 1
    return -(-total // sw) * sw + 2


PAIR_OVERHEAD = 16         # kMigPairOverheadSteps * (64 / G): what starting a pair costs its lane group, in steps (counted by the schedule)


def schedule(p, m):
    """p: step counts in sorted order (0 = a pair this kernel skips).  Returns T and, per lane group, its list of segments
    (pair, first step, last step + 1, kind) in the order the group runs them: kind 'early' = first part of the pair that
    crosses INTO its interval (suspended at the end), 'whole', 'late' = rest of the pair that crosses OUT (resumed)."""
    cum = np.concatenate([[0], np.cumsum(p)])
    P, pm = int(cum[-1]), int(p.max())
    T = max(pm, -(-P // m), 1)
    groups = []
    for g in range(m):
        lo, hi = g * T, (g + 1) * T
        segs = []
        j = int(np.searchsorted(cum[1:], lo, side="right"))        # first pair with cum[j + 1] > lo
        while j < len(p) and cum[j] < hi:
            c, c1 = int(cum[j]), int(cum[j + 1])
            if c1 > c:
                if c < lo:
                    segs.append((j, 0, c1 - lo, "early"))
                elif c1 <= hi:
                    segs.append((j, 0, c1 - c, "whole"))
                else:
                    segs.append((j, c1 - hi, c1 - c, "late"))
            j += 1
        groups.append(segs)
    return T, groups


@pytest.mark.parametrize("seed", range(6))
def test_every_step_of_every_pair_runs_exactly_once_and_in_order(seed):
    rng = np.random.default_rng(seed)
    m = int(rng.choice([4, 16, 64, 512]))
    n = int(rng.integers(m + 1, 6 * m))
    sw = int(rng.choice([1, 3, 7]))
    Q = rng.integers(1, 3000, n); R = rng.integers(1, 3000, n)
    p = np.array([step_count(int(q), int(r), sw) + PAIR_OVERHEAD for q, r in zip(Q, R)])
    p[rng.random(n) < 0.05] = 0                                     # pairs of another kind: skipped
    p = -np.sort(-p)                                                # longest first
    T, groups = schedule(p, m)
    assert T >= p.max() and T * m >= p.sum()
    covered = {j: [] for j in range(n) if p[j] > 0}
    when = {}
    for g, segs in enumerate(groups):
        t = 0
        kinds = [s[3] for s in segs]
        assert kinds.count("early") <= 1 and kinds.count("late") <= 1
        if "early" in kinds:
            assert kinds[0] == "early"                              # a group starts with the pair it has to suspend
        if "late" in kinds:
            assert kinds[-1] == "late"                              # and ends with the pair it resumes
        for j, a, b, kind in segs:
            covered[j].append((a, b))
            when[(j, kind)] = (t, t + b - a)                        # local time of the group, in steps
            t += b - a
        assert t <= T                                               # no lane group runs more than T steps
    for j, parts in covered.items():
        parts.sort()
        assert parts[0][0] == 0 and parts[-1][1] == p[j] and len(parts) <= 2
        if len(parts) == 2:
            assert parts[0][1] == parts[1][0]
            # the first part ends (its group's local time) before the second part starts: p_j <= T keeps them apart
            assert when[(j, "early")][1] <= when[(j, "late")][0]


def test_step_count_formula():
    assert step_count(8, 8, 3) == 3 + 2 and step_count(10000, 10000, 3) == 2499 + 2 and step_count(1, 1, 1) == 3
