"""GPU: the five BASELINE.json workload shapes, at sizes the oracle can check, plus size-independent
properties on the full 10 k-pair headline batch."""
import numpy as np
import pytest

from oracle import oracle as O, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import agatha_amd
    e = agatha_amd.Engine(0)
    yield e
    e.close()


def _run(eng, qs, ts, **p):
    import agatha_amd
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    got = eng.align_host_batch(qb, tb, qo, to, ql, tl, agatha_amd.Scores.make(**p))
    return (qb, tb, qo, to, ql, tl), got


def _check(batch, got, threads=8, **p):
    exp = O.align_batch(*batch, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=threads)
    for a, b in zip(got, exp):
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, (p, bad[:5], a[bad[:5]], b[bad[:5]])


def test_c0_bundled_dataset_standin(eng):
    qs, ts = synth.cfg_c0(n=400)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    batch, got = _run(eng, qs, ts, **p)
    _check(batch, got, **p)


def test_c2_hifi_band500_both_scorings(eng):
    qs, ts = synth.cfg_c2(n=96)
    for p in (dict(m=1, x=4, q=6, r=2, s=3, z=400, w=500),       # inside the reference's 16-bit domain
              dict(m=2, x=4, q=4, r=2, s=3, z=400, w=500)):      # scores reach 40 000: int32 ("wide") semantics
        batch, got = _run(eng, qs, ts, **p)
        assert eng.last_config() in ((32, 2), (64, 1))
        _check(batch, got, **p)
    assert got[0].max() > 32767


def _run_batch(eng, qs, ts, **p):
    """like _run, plus (kernel choice, schedule info, step statistics)"""
    import agatha_amd
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p)); b.download(); eng.synchronize()
        got = [b.res_host[j].copy() for j in range(3)]
        info = (b.kernel_choice(), b.schedule_info(), b.step_stats(), b.pair_kinds())
    finally:
        b.free()
    return (qb, tb, qo, to, ql, tl), got, info


def test_c2_at_the_size_of_one_gpu_share(eng):
    """C2 at 9 000 pairs (BASELINE: 100 k over 8 GPUs): more pairs than the <16, 2> shape has lane groups, so the static
    schedule runs on HiFi shapes inside the suite -- pairs suspended by one lane group and resumed by another, value steps,
    checkpoints (4 400 steps per pair); 500 pairs spread over the batch against the oracle, the rest through determinism."""
    qs, ts = synth.cfg_c2(n=9000)
    p = dict(m=1, x=4, q=6, r=2, s=3, z=400, w=500)
    batch, got, (choice, sched, st, kinds) = _run_batch(eng, qs, ts, **p)
    assert choice == ("int16", 16, 4) and sched[0]
    assert st[0] > 10 * st[1] > 0 and kinds[2] == 0
    k = np.sort(np.random.default_rng(2).choice(9000, 500, replace=False))
    sb = O.make_batch([qs[i] for i in k]), O.make_batch([ts[i] for i in k])
    exp = O.align_batch(sb[0][0], sb[1][0], sb[0][1], sb[1][1], sb[0][2], sb[1][2], O.make_params(**p), wide=True,
                        model=O.MODEL_SLICES, threads=16)
    assert all((np.asarray(a)[k] == b).all() for a, b in zip(got, exp))
    _, again, _ = _run_batch(eng, qs, ts, **p)
    assert all((a == b).all() for a, b in zip(got, again))


def test_c3_ultralong_band1500(eng):
    """64 pairs of ~100 kb at band 1500: the device picks the shape that puts one pair on two cooperating waves, and with 64
    pairs on 1024 workgroup slots every pair... runs alone; the queue of that shape is drawn from in the test below"""
    qs, ts = synth.cfg_c3(n=64)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=1500)
    batch, got, (choice, sched, st, kinds) = _run_batch(eng, qs, ts, **p)
    assert choice == ("int16", 128, 2) and kinds[2] == 0
    assert max(len(q) for q in qs) > 65536
    _check(batch, got, threads=16, **p)


def test_c3_shape_draws_from_its_queue(eng):
    """The two-wave shape with more pairs than workgroups (1 100 pairs of 13-15 kb at band 1500 on 1024 workgroups: the last ones
    come from the queue), several workgroups per CU (the waves of a pair drift apart in time), scores that move the base."""
    import agatha_amd
    qs, ts = synth.make_pairs(77, 1100, lambda r: int(r.integers(13000, 15000)), 0.03, 0.03, 0.04)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=1500)
    agatha_amd.set_debug_option("force_choice", 1)
    try:
        batch, got, (choice, sched, st, kinds) = _run_batch(eng, qs, ts, **p)
    finally:
        agatha_amd.set_debug_option("force_choice", -1)
    assert choice == ("int16", 128, 2) and kinds[2] == 0 and st[2] == 0
    k = np.sort(np.random.default_rng(3).choice(1100, 300, replace=False))
    sb = O.make_batch([qs[i] for i in k]), O.make_batch([ts[i] for i in k])
    exp = O.align_batch(sb[0][0], sb[1][0], sb[0][1], sb[1][1], sb[0][2], sb[1][2], O.make_params(**p), wide=True,
                        model=O.MODEL_SLICES, threads=16)
    assert all((np.asarray(a)[k] == b).all() for a, b in zip(got, exp))


def test_c4_mixed_lengths_heavy_zdrop(eng):
    """BASELINE's full length range, 1 k - 100 k, 30 % broken / high-error pairs: z-drop on pairs far beyond the reference's
    16-bit domain is checked against the (wide) oracle."""
    qs, ts = synth.cfg_c4(n=120, lo=1000, hi=100000)
    assert max(len(q) for q in qs) > 80000
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    batch, got = _run(eng, qs, ts, **p)
    _check(batch, got, **p)
    qs, ts = synth.cfg_c4(n=400, lo=1000, hi=40000)
    batch, got = _run(eng, qs, ts, **p)
    _check(batch, got, **p)
    # a good share of these pairs must actually have been cut short by z-drop
    ql, tl = batch[4].astype(np.int64), batch[5].astype(np.int64)
    assert np.mean(got[1] + got[2] + 2 < 0.8 * (ql + tl)) > 0.15


def test_c1_full_batch_properties(eng):
    """10 000 pairs (the bench workload): determinism, independence from batch composition, and a sampled oracle check."""
    import agatha_amd
    qs, ts = synth.cfg_c1(n=10000)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    batch, got = _run(eng, qs, ts, **p)
    _, again = _run(eng, qs, ts, **p)
    assert all((a == b).all() for a, b in zip(got, again))                      # atomic queue order must not matter
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(10000, 600, replace=False))
    _, sub = _run(eng, [qs[i] for i in pick], [ts[i] for i in pick], **p)       # same pairs in a different batch
    assert all((a[pick] == b).all() for a, b in zip(got, sub))
    # ALL 10 000 pairs against the block-granular oracle (the restatement of the reference kernel, not the exact-band port):
    # they ran on a static schedule, i.e. most lane groups suspended one pair and resumed another, on value steps with key
    # steps at the pairs' ends
    exp = O.align_batch(*batch, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=16)
    assert all((a == b).all() for a, b in zip(got, exp))
    # checksum of checksums for the record (recomputable from the seeds)
    assert int(got[0].sum()) > 0 and (got[1] < batch[4]).all() and (got[2] < batch[5] + 8).all()


def test_identity_pairs_score_linearly(eng):
    rng = np.random.default_rng(5)
    seqs = [synth.random_seq(rng, n).tobytes() for n in (1, 8, 9, 1000, 12345, 40000)]
    _, got = _run(eng, seqs, seqs, m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    for s, L, q, t in zip(got[0], map(len, seqs), got[1], got[2]):
        assert (s, q, t) == (2 * L, L - 1, L - 1)


def _sampled_check(batch_lists, got, k, p):
    qs, ts = batch_lists
    sb = O.make_batch([qs[i] for i in k]), O.make_batch([ts[i] for i in k])
    exp = O.align_batch(sb[0][0], sb[1][0], sb[0][1], sb[1][1], sb[0][2], sb[1][2], O.make_params(**p), wide=True,
                        model=O.MODEL_SLICES, threads=16)
    assert all((np.asarray(a)[k] == b).all() for a, b in zip(got, exp))


@pytest.mark.parametrize("cut", ["target", "query"])
def test_c1_pairs_of_unequal_lengths_on_the_static_schedule(eng, cut):
    """10 000 C1 pairs with one sequence cut to 85 %: the maximum of such an extension rises for the last time where the SHORTER
    sequence ends, 190 steps before the pair's last step.  The window of key steps at a pair's end is anchored there (a window
    anchored at the last step made every one of these pairs start over: 70 ms instead of 27): no pair is started over, taken
    back or handed to the int32 kernel, and 400 sampled pairs are the oracle's."""
    qs, ts = synth.cfg_c1(n=10000)
    if cut == "target":
        ts = [t[:int(len(t) * 0.85)] for t in ts]
    else:
        qs = [q[:int(len(q) * 0.85)] for q in qs]
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    batch, got, (choice, sched, st, kinds) = _run_batch(eng, qs, ts, **p)
    assert choice == ("int16", 16, 6) and sched[0]
    assert st[2] == 0 and st[15] == 0 and st[24] == 0 and kinds[2] == 0 and st[0] > 4 * st[1]
    _sampled_check((qs, ts), got, np.sort(np.random.default_rng(8).choice(10000, 400, replace=False)), p)


def test_c1_with_broken_pairs_on_the_static_schedule(eng):
    """2 % of 10 000 C1 reads have an unrelated tail: z-drop ends those extensions, on a value step that cannot decide it.  On the
    static schedule such a pair goes back to a checkpoint in place (it ends early, so its lane group has the steps to spare); none
    goes to the int32 kernel behind.  All the broken pairs and 300 others against the oracle."""
    qs, ts = synth.cfg_c1(n=10000)
    rng = np.random.default_rng(21)
    broken = np.sort(rng.choice(10000, 200, replace=False))
    ts = list(ts)
    for i in broken:
        a = np.frombuffer(ts[i], np.uint8).copy()
        h = int(rng.integers(len(a) // 10, len(a)))
        a[h:] = synth.random_seq(rng, len(a) - h)
        ts[i] = a.tobytes()
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    batch, got, (choice, sched, st, kinds) = _run_batch(eng, qs, ts, **p)
    assert choice == ("int16", 16, 6) and sched[0]
    assert st[15] > 100 and st[24] == 0 and kinds[2] == 0
    others = np.setdiff1d(np.random.default_rng(9).choice(10000, 300, replace=False), broken)
    _sampled_check((qs, ts), got, np.sort(np.concatenate([broken, others])), p)
    ql, tl = batch[4].astype(np.int64), batch[5].astype(np.int64)
    assert np.mean((got[1] + got[2] + 2 < 0.95 * (ql + tl))[broken]) > 0.7          # they did end early


def test_targets_shorter_than_their_queries_by_more_than_the_band_stay_on_the_int16_kernel(eng):
    """HiFi-like pairs (band 500: cut diagonal -4) whose target is cut to 90 %: the band leaves the matrix through the last column a
    few hundred rows below the target's end, and the last block of the column holds only padded columns -- cells that derive from the
    reference's -infinity (agatha_kernel.h:207-215).  An anti-diagonal with nothing else on it used to trip the int16 kernel's range
    check (1 163 of 9 000 such pairs went to the int32 kernel: 65 instead of 31 ms); it now ends the pair, whose result is final.
    Asserted: no pair is abandoned, 300 pairs equal the oracle; the same with the QUERIES cut."""
    qs, ts = synth.cfg_c2(n=3000)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=500)
    for cut_targets in (True, False):
        q2 = [q if cut_targets else q[:int(len(q) * 0.9)] for q in qs]
        t2 = [t[:int(len(t) * 0.9)] if cut_targets else t for t in ts]
        batch, got, (choice, sched, st, kinds) = _run_batch(eng, q2, t2, **p)
        assert choice[0] == "int16" and kinds[2] == 0 and st[24] == 0, (choice, kinds)
        _sampled_check((q2, t2), got, np.sort(np.random.default_rng(4).choice(3000, 300, replace=False)), p)


def test_c1_with_many_broken_pairs_goes_back_to_checkpoints_not_to_first_steps(eng):
    """30 % of 10 000 C1 reads have an unrelated tail (tools/gpu_skew.py's hardest batch): thousands of pairs go back to a checkpoint,
    many of them after the schedule has moved them to another lane group.  Round 4: a pair goes back to the NEWER of its two
    checkpoints when the bound it keeps of its maximum allows it (debug option ck_newer), and a pair that is resumed on key steps by a
    wave on value steps keeps its exact keys -- before, such a pair started over from its first step far into its steps.  Asserted:
    every broken pair and 300 others equal the oracle; the whole batch is the same with ck_newer = 0; few pairs start over from their
    first step (they do so only within their first checkpoint spans) and the key steps stay a fraction of the value steps."""
    import agatha_amd
    qs, ts = synth.cfg_c1(n=10000)
    rng = np.random.default_rng(5)
    ts = list(ts)
    broken = []
    for i in range(10000):
        if rng.random() < 0.3:
            a = np.frombuffer(ts[i], np.uint8).copy()
            h = int(rng.integers(len(a) // 10, len(a)))
            a[h:] = synth.random_seq(rng, len(a) - h)
            ts[i] = a.tobytes()
            broken.append(i)
    broken = np.array(broken)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    batch, got, (choice, sched, st, kinds) = _run_batch(eng, qs, ts, **p)
    assert choice == ("int16", 16, 6) and sched[0]
    assert st[15] > 2000 and st[24] == 0 and kinds[2] == 0
    assert st[2] < 200 and st[1] < st[0] // 4, (st[2], st[1], st[0])
    others = np.setdiff1d(np.random.default_rng(9).choice(10000, 300, replace=False), broken)
    _sampled_check((qs, ts), got, np.sort(np.concatenate([broken, others])), p)
    old = agatha_amd.get_debug_option("ck_newer")
    agatha_amd.set_debug_option("ck_newer", 0)
    try:
        _, got0, (_, _, st0, _) = _run_batch(eng, qs, ts, **p)
    finally:
        agatha_amd.set_debug_option("ck_newer", old)
    assert all(np.array_equal(a, b) for a, b in zip(got, got0))
    assert st0[15] > 2000


def test_a_few_long_pairs_among_many_short_ones_are_split_between_the_two_int16_shapes(eng):
    """Round 4 (the purpose of the reference's uneven bucketing, agatha_kernel.h:113, and subwarp rejoining, :365-408, re-derived
    for a batch of mixed lengths): 20 000 bundled-dataset-like pairs with 12 pairs of 30 kb among them.  One shape per launch puts
    everything on the latency shape (the short pairs at 60 % of the throughput shape's rate) or makes the long pairs crawl on the
    throughput shape; the device now sends the long ones to <64, 1>, one pair per wave on a second stream, and the rest to <16, 3>,
    side by side.  Asserted: both shapes ran (agatha_amd_split_info), every long pair and 1 500 short ones spread over the batch
    equal the oracle, the whole batch equals a run with the split switched off."""
    import agatha_amd
    qs, ts = synth.cfg_c0(n=20000)
    lq, lt = synth.make_pairs(77, 12, lambda r: int(r.integers(29000, 31000)), 0.03, 0.03, 0.04)
    pos = np.sort(np.random.default_rng(3).choice(20000, 12, replace=False))
    for k, at in enumerate(pos):
        qs.insert(int(at) + k, lq[k]); ts.insert(int(at) + k, lt[k])
    where_long = [int(at) + k for k, at in enumerate(pos)]
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p)); b.download(); eng.synchronize()
        got = [b.res_host[j].copy() for j in range(3)]
        choice, split, st = b.kernel_choice(), b.split_info(), b.step_stats()
        agatha_amd.set_debug_option("no_split", 1)
        try:
            b.align(agatha_amd.Scores.make(**p)); b.download(); eng.synchronize()
            one_shape = [b.res_host[j].copy() for j in range(3)]
            assert b.split_info()[0] == 0
        finally:
            agatha_amd.set_debug_option("no_split", 0)
    finally:
        b.free()
    # the throughput shape + the 12 long pairs (and whichever of the longest short ones the cost model sends along) on <64, 1>
    assert choice == ("int16", 16, 6) and 12 <= split[0] <= 512 and split[1:] == (64, 2), (choice, split)
    assert all((a == c).all() for a, c in zip(got, one_shape))
    k = np.unique(np.concatenate([np.asarray(where_long), np.random.default_rng(4).choice(len(qs), 1500, replace=False)]))
    sb = O.make_batch([qs[i] for i in k]), O.make_batch([ts[i] for i in k])
    exp = O.align_batch(sb[0][0], sb[1][0], sb[0][1], sb[1][1], sb[0][2], sb[1][2], O.make_params(**p), wide=True,
                        model=O.MODEL_SLICES, threads=16)
    assert all((np.asarray(a)[k] == e).all() for a, e in zip(got, exp))
