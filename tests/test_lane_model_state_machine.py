"""CPU: the state machine around the packed-int16 kernel's speculative value steps, driven through every transition
by the CPU model of the kernel's decisions (oracle/agatha_lanes_model.c, agatha_model_lanes16 with a margin and, new in round 5,
with checkpoints: agatha_lanes16_ck_span).  On the GPU only the fuzzers reach these paths end to end; here each one is forced and
its RESULT is compared with the oracle:

    value steps -> window of key steps -> result                                   (kind 0: as it came)
    a value step is not calm (z-drop in reach on a broken read)    -> gives up
    the pair ends without the cell of its maximum (window too narrow) -> gives up
        gives up, no checkpoint yet                   -> starts from its first step on key steps           (kind 2)
        gives up, bound has risen since the newer checkpoint by > slack + 14 ge -> back to the NEWER one   (kind 3, counts[0])
        gives up, it has not                          -> back to the OLDER one                              (kind 3, counts[1])
        goes back, key steps from there never place the maximum -> gives up again -> first step            (kind 4)

What the GPU adds on top of this (suspension and resume between lane groups, the fallback state of a suspended pair) restores
states of the SAME computation; tests/test_gpu_int16.py and tools/gpu_fuzz_mig.py cover those on the chip."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from agatha_amd import workload as W


def _batch(seed, n):
    rng = np.random.default_rng(seed)
    qs, ts = W.make_pairs(seed, n, lambda r: int(r.integers(600, 4000)), 0.03, 0.03, 0.04)
    out = []
    for t in ts:
        a = np.frombuffer(t, np.uint8).copy()
        if rng.random() < 0.33:         # a third of the reads: an unrelated tail from a random point on (z-drop ends them there)
            h = int(rng.integers(len(a) // 10, len(a)))
            a[h:] = W.random_seq(rng, len(a) - h)
        out.append(a.tobytes())
    return O.make_batch(qs), O.make_batch(out)


@pytest.fixture()
def knobs():
    lib = O.lib()
    k = dict(span=C.c_int.in_dll(lib, "agatha_lanes16_ck_span"), counts=(C.c_int * 2).in_dll(lib, "agatha_lanes16_ck_counts"),
             cap_min=C.c_int.in_dll(lib, "agatha_lanes16_win_cap_min"), cap_div=C.c_int.in_dll(lib, "agatha_lanes16_win_cap_div"))
    saved = (k["span"].value, k["cap_min"].value, k["cap_div"].value)
    yield k
    k["span"].value, k["cap_min"].value, k["cap_div"].value = saved
    k["counts"][0] = k["counts"][1] = 0


@pytest.mark.parametrize("scoring", [dict(m=1, x=4, q=6, r=2, z=400), dict(m=2, x=4, q=4, r=2, z=400), dict(m=1, x=4, q=6, r=2, z=-1)])
def test_every_transition_gives_the_oracles_result(knobs, scoring):
    (qb, qo, ql), (tb, to, tl) = _batch(11, 220)
    p = O.make_params(**scoring)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, p, wide=True, model=O.MODEL_STEPS, threads=8)
    seen = np.zeros(6, np.int64)
    newer = older = 0
    # (no checkpoints; a checkpoint every 64 / 128 steps; the window as it is / capped at 24 steps so that pairs end without the cell)
    for span, cap in ((0, (128, 16)), (64, (128, 16)), (64, (24, 1 << 20)), (128, (24, 1 << 20))):
        knobs["span"].value = span
        knobs["cap_min"].value, knobs["cap_div"].value = cap
        knobs["counts"][0] = knobs["counts"][1] = 0
        got = O.lanes16_batch(qb, tb, qo, to, ql, tl, p, 16, 6, threads=8, value_step_margin=12)
        for g, e in zip(got[:3], exp):
            bad = np.nonzero(np.asarray(g) != np.asarray(e))[0]
            assert bad.size == 0, (scoring, span, cap, bad[:4], got[3][bad[:4]])
        kinds = np.bincount(got[3] + 1, minlength=6)
        assert kinds[0] == 0 and kinds[2] == 0                      # nothing refused, nothing left to the int32 model
        if span == 0:
            assert kinds[4] == 0 and kinds[5] == 0                  # no checkpoints: a pair that gives up starts from its first step
        seen += kinds
        newer += knobs["counts"][0]; older += knobs["counts"][1]
    # every transition was taken by some pair of some run
    assert seen[1] > 0 and seen[3] > 0 and seen[4] > 0, seen        # as it came / from its first step / back to a checkpoint
    assert older > 0
    if scoring["z"] >= 0 or scoring["m"] == 1:
        assert seen[5] > 0, seen                                    # went back, gave up again, started over


def test_both_checkpoint_rules_are_exercised(knobs):
    (qb, qo, ql), (tb, to, tl) = _batch(12, 160)
    p = O.make_params(m=1, x=4, q=6, r=2, z=400)
    knobs["span"].value = 128
    knobs["cap_min"].value, knobs["cap_div"].value = 24, 1 << 20
    knobs["counts"][0] = knobs["counts"][1] = 0
    got = O.lanes16_batch(qb, tb, qo, to, ql, tl, p, 16, 6, threads=8, value_step_margin=12)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, p, wide=True, model=O.MODEL_STEPS, threads=8)
    assert all((np.asarray(g) == np.asarray(e)).all() for g, e in zip(got[:3], exp))
    assert knobs["counts"][0] > 0 and knobs["counts"][1] > 0, list(knobs["counts"])      # the newer and the older checkpoint


def test_probation_returns_a_pair_to_value_steps_and_changes_no_result(knobs):
    """Round 5 ("probation"; align16_acquire.inc / align16_step_maxima.inc, PROB): a pair that goes back to a checkpoint runs
    key steps until 32 steps behind the step it gave up on, on while z-drop is within reach of what a value step can tell, and then returns
    to value steps; a pair that gives up on probation starts from its first step on key steps for good.  Reads with a 350-base burst of
    errors at the reference's scoring (match 1: the dip brings z-drop within reach, the read recovers): most pairs go back, most of those
    return, the key steps of the batch more than halve, and every result is the oracle's -- with probation and without.  Then the
    broken-read batch of the other tests through every span with probation on: pairs that give up on probation, results the oracle's."""
    lib = O.lib()
    prob, left = C.c_int.in_dll(lib, "agatha_lanes16_probation"), C.c_int.in_dll(lib, "agatha_lanes16_left_probation")
    steps = (C.c_longlong * 2).in_dll(lib, "agatha_lanes16_steps")
    qs, ts0 = W.make_pairs(5, 48, lambda r: int(r.integers(8000, 12000)), 0.03, 0.03, 0.04)
    rng = np.random.default_rng(11)
    ts = []
    for t in ts0:
        a = np.frombuffer(t, np.uint8).copy()
        at = int(rng.integers(len(a) // 5, len(a) * 4 // 5 - 350))
        ts.append(np.concatenate([a[:at], W.mutate(rng, a[at:at + 350], 0.15, 0.12, 0.13), a[at + 350:]]).tobytes())
    (qb, qo, ql), (tb, to, tl) = O.make_batch(qs), O.make_batch(ts)
    p = O.make_params(m=1, x=4, q=6, r=2, z=400, w=751)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, p, wide=True, model=O.MODEL_STEPS, threads=8)
    knobs["span"].value = 256                      # the kernel's span for pairs of 2 048 .. 4 095 steps
    seen = {}
    try:
        for on in (0, 1):
            prob.value = on
            left.value = 0; steps[0] = steps[1] = 0
            got = O.lanes16_batch(qb, tb, qo, to, ql, tl, p, 16, 6, threads=8, value_step_margin=12)
            assert all((np.asarray(g) == np.asarray(e)).all() for g, e in zip(got[:3], exp)), on
            seen[on] = (int((got[3] == 3).sum()), left.value, steps[0], steps[1])
        went_back, returned, _, key_on = seen[1]
        assert seen[0][0] >= 24 and went_back >= 24, seen             # half the pairs and more go back to a checkpoint
        assert seen[0][1] == 0 and returned >= went_back // 2, seen   # ... and most of those leave their probation
        assert key_on < 0.6 * seen[0][3], seen                        # the batch's key steps (44 k -> 20 k when this was written)
        # broken reads (z-drop ends a third of them): pairs that give up on probation start from their first step; every span, exact
        (qb, qo, ql), (tb, to, tl) = _batch(11, 220)
        p = O.make_params(m=1, x=4, q=6, r=2, z=400)
        exp = O.align_batch(qb, tb, qo, to, ql, tl, p, wide=True, model=O.MODEL_STEPS, threads=8)
        kinds = np.zeros(6, np.int64)
        for span, cap in ((64, (128, 16)), (64, (24, 1 << 20)), (128, (24, 1 << 20)), (256, (128, 16))):
            knobs["span"].value = span
            knobs["cap_min"].value, knobs["cap_div"].value = cap
            got = O.lanes16_batch(qb, tb, qo, to, ql, tl, p, 16, 6, threads=8, value_step_margin=12)
            for g, e in zip(got[:3], exp):
                bad = np.nonzero(np.asarray(g) != np.asarray(e))[0]
                assert bad.size == 0, (span, cap, bad[:4], got[3][bad[:4]])
            kinds += np.bincount(got[3] + 1, minlength=6)
        assert kinds[4] > 0 and kinds[5] > 0, kinds                   # went back (and stayed) / went back, gave up again, started over
    finally:
        prob.value = 0


@pytest.mark.parametrize("scoring", [dict(m=2, x=4, q=4, r=2, z=400), dict(m=1, x=4, q=6, r=2, z=400), dict(m=2, x=4, q=4, r=2, z=120), dict(m=2, x=4, q=4, r=2, z=-1)])
def test_lazy_value_steps_change_no_result(knobs, scoring):
    """Round 6 ("lazy value steps"; align16_step_blocks.inc / align16_step_maxima.inc: `lazy`, `skip_until`, `acc_hi`): where a wave
    holds one pair a calm test that passed with room to spare answers for up to lazy_max - 1 steps behind it; the kernel computes no lower bound,
    reduces nothing and tests nothing on those.  The lane model mirrors the rule (agatha_lanes16_lazy_max; on every shape with
    agatha_lanes16_lazy_any_shape, the kernel's choice being G >= 64): clean reads and reads that z-drop ends, with and without checkpoints, the
    window as it is and capped so that pairs end without the cell of their maximum -- every result the oracle's whatever lazy_max, and the shape the
    kernel uses it on (64 lanes, two slots per lane) as well as the throughput shape."""
    lib = O.lib()
    lazy, any_shape = C.c_int.in_dll(lib, "agatha_lanes16_lazy_max"), C.c_int.in_dll(lib, "agatha_lanes16_lazy_any_shape")
    (qb, qo, ql), (tb, to, tl) = _batch(23, 120)
    w = 751 if scoring["z"] != 120 else 300
    p = O.make_params(w=w, **scoring)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, p, wide=True, model=O.MODEL_STEPS, threads=8)
    saved = (lazy.value, any_shape.value)
    n_lazy = C.c_longlong.in_dll(lib, "agatha_lanes16_lazy_steps")
    try:
        for G, S in ((64, 2), (16, 6)):
            if G * S * 8 < w + 8: continue
            for lm in (0, 2, 8):
                for span, cap in ((0, (128, 16)), (64, (128, 16)), (128, (24, 1 << 20))):
                    lazy.value, any_shape.value = lm, 1
                    n_lazy.value = 0
                    knobs["span"].value = span
                    knobs["cap_min"].value, knobs["cap_div"].value = cap
                    got = O.lanes16_batch(qb, tb, qo, to, ql, tl, p, G, S, threads=8, value_step_margin=12)
                    assert all((np.asarray(g) == np.asarray(e)).all() for g, e in zip(got[:3], exp)), (G, S, lm, span, cap)
                    if lm == 0: assert n_lazy.value == 0
                    if lm == 8 and scoring["z"] in (400, -1): assert n_lazy.value > 1000, n_lazy.value
    finally:
        lazy.value, any_shape.value = saved
