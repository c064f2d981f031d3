"""CPU-only: round 5's additions to the preemptive static schedule -- the POOL of the suspended pairs' rests and the deal of the intervals to
the physical lane groups (schedule_kernel in agatha_amd/csrc/align_kernel.hip, the acquisition loop of align16_acquire.inc) -- restated in
tools/sched_sim.py and checked for what the device code relies on.  (tests/test_schedule_model.py pins McNaughton's wrap-around itself; the
GPU tests run the real thing.)"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import sched_sim as SS      # noqa: E402


def _batch(seed, m, rounds):
    rng = np.random.default_rng(seed)
    n = int(m * rounds)
    tot = rng.integers(300, 3000, n)
    p = -(-tot // 3) * 3 + 2 + SS.PAIR_OVERHEAD
    p[rng.random(n) < 0.03] = 0                     # pairs of another kind: skipped
    return -np.sort(-p)


@pytest.mark.parametrize("seed", range(4))
def test_pool_and_permutation_are_what_schedule_kernel_builds(seed):
    m = 512
    p = _batch(seed, m, [1.2, 1.7, 2.4, 5.0][seed])
    S = SS.build(p, m, num_cus=16)
    T, cum, rest = S["T"], S["cum"], S["rest"]
    assert T >= p.max() and T * m >= p.sum()
    # every boundary that cuts a pair is in the pool exactly once, longest rest first (by the 2 048 bins of the counting sort)
    cutting = [b for b in range(1, m) if rest[b] > 0]
    assert sorted(S["pool"]) == cutting
    bins = [2047 - (int(rest[b]) * 2047) // T for b in S["pool"]]
    assert bins == sorted(bins)
    for b in cutting:
        j = int(S["cross"][b])
        assert cum[j] < b * T < cum[j + 1] and rest[b] == b * T - cum[j]
    # the deal: a permutation; a wave's four lane groups own intervals with neighbouring rests, consecutive waves of the order share a CU
    assert sorted(S["perm"]) == list(range(m))
    own = np.array([rest[g + 1] for g in S["perm"]])
    gpb = m // 32
    spread = [np.ptp(own[w * 4:(w + 1) * 4]) for w in range(m // 4)]
    assert np.median(spread) <= T / (m // 8), (np.median(spread), T)          # (intervals dealt at random would spread a wave's rests over T / 2)
    # rank of an interval in the order of the sort (bin of its rest, then its number: the device's order inside a bin is that of its atomics)
    order = sorted(range(m), key=lambda g: (2047 - (int(rest[g + 1]) * 2047) // T, g))
    rank = {g: u for u, g in enumerate(order)}
    for cu in range(16):
        mine = sorted(rank[int(g)] for g in np.concatenate([S["perm"][cu * gpb:(cu + 1) * gpb], S["perm"][(cu + 16) * gpb:(cu + 17) * gpb]]))
        assert mine == list(range(cu * 2 * gpb, (cu + 1) * 2 * gpb))           # one CU (workgroups cu and cu + half): 2 gpb consecutive ranks


@pytest.mark.parametrize("seed,rounds", [(0, 1.3), (1, 2.2), (2, 4.0)])
def test_every_step_runs_once_nobody_waits_for_ever_and_on_time_is_mcnaughton(seed, rounds):
    m = 256
    # pairs of similar lengths (the line is full: T = ceil(P / m)), every wave-step costs the same, nobody is late: each lane group is done with
    # its fixed part after T - (its own rest) steps and finds that rest on top of the pool (ties inside a bin of the sort aside: then a
    # neighbour's, a step or two longer or shorter), and every wave ends at T
    p = SS.c1_like(n=int(m * rounds), seed=seed)
    S = SS.build(p, m, num_cus=8)
    assert S["T"] == -(-int(p.sum()) // m)
    end, executed, c = SS.simulate(S, window=0, value_us=1.0, key_us=1.0, lone_speedup=1.0, num_cus=8)
    assert (executed == p).all()
    assert end.max() <= S["T"] + 4 and end.min() >= S["T"] - 8 - p.sum() % m, (end.min(), end.max(), S["T"])
    assert c["taken_whole"] == 0 and c["own_rest"] >= 0.8 * c["draws"], c
    # the forms cost what they cost on the chip, pairs have windows of key steps, waves drift: the pool hands out other rests, every step still
    # runs exactly once, and the launch ends within a few percent of its lower bound
    end2, executed2, c2 = SS.simulate(S, window=60, num_cus=8)
    assert (executed2 == p).all()
    assert end2.max() < 1.08 * S["T"] * SS.VALUE_US, (end2.max(), S["T"] * SS.VALUE_US)
    # pairs of all lengths (T is the longest pair, most of the line is empty, idle lane groups take the longest rests at once and wait for
    # them): every step still runs exactly once and nobody runs beyond T
    p = _batch(seed, m, rounds)
    S = SS.build(p, m, num_cus=8)
    end, executed, c = SS.simulate(S, window=0, value_us=1.0, key_us=1.0, lone_speedup=1.0, num_cus=8)
    assert (executed == p).all()
    assert end.max() <= S["T"] + 4              # (a rest of the same bin of the sort, a step or two longer than its own)



def test_a_first_part_nobody_started_is_taken_whole_and_its_owner_skips_it():
    m = 256
    p = _batch(5, m, 1.5)
    S = SS.build(p, m, num_cus=8)
    start = np.zeros(m // 4)
    start[1::2] = 0.6 * S["T"]                        # every second wave becomes resident late (a second persistent grid behind another one)
    end, executed, c = SS.simulate(S, window=0, value_us=1.0, key_us=1.0, lone_speedup=1.0, num_cus=8, start_us=start)
    assert c["taken_whole"] > 0
    assert (executed == p).all()                      # taken whole = run once, by the lane group that took it
    assert end.max() < 1.7 * S["T"]                   # (half the chip idles for 0.6 T: the others take over what they can)


def test_pairs_that_run_extra_steps_cost_the_launch_its_worst_wave():
    """What the simulation is for: a pair that goes back to a checkpoint runs e extra steps; the batch's steps grow by
    a fraction of a percent, the launch by e / T -- and by less when the trip back is shorter or the key steps behind it end (probation)."""
    m = 1024
    p = SS.c1_like(n=1250, seed=1)
    S = SS.build(p, m, num_cus=32)
    base, _, _ = SS.simulate(S, window=170, num_cus=32)
    rng = np.random.default_rng(2)
    res = {}
    for name, back, on in (("off", 300, 0), ("on", 300, 1), ("on, short way back", 100, 1)):
        pieces = {}
        for j in np.nonzero(rng.random(len(p)) < 0.10)[0]:
            steps = int(p[j]); g = int(rng.integers(steps // 5, steps * 4 // 5))
            pieces[int(j)] = SS.burst_pieces(steps, 170, g, max(g - back, 1), g + 120 if on else steps)
        end, executed, _ = SS.simulate(S, pieces=pieces, window=170, num_cus=32)
        extra = executed.sum() / p.sum() - 1.0
        res[name] = (end.max() / base.max() - 1.0, extra)
    assert all(v[1] < 0.02 for v in res.values()), res                  # the batch's steps: + 1 %
    assert res["off"][0] > 0.15 and res["off"][0] > res["on"][0] > res["on, short way back"][0] > 0.03, res      # the launch: + 10 .. 40 %
