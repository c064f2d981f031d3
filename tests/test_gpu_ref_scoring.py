"""GPU: the reference's ONE bench command (AGAThA.sh:44: `manual -p -m 1 -x 4 -q 6 -r 2 -s 3 -z 400 -w 751`; timed region
AGAThA/src/gasal_align.cu:219-236) on the BASELINE shapes.  At match 1 the score of a 10 %-error read rises by a point and a half
per step with a deviation of five, so where the maximum rises for the last time is a matter of a hundred steps, not of ten: round
4's window of key steps lost the cell of the maximum in 0.75 % of clean C1 pairs and 3.6 % of clean C0 pairs (they were started
over: + 12 % / + 23 % on the kernel, profiles/r05_v0).  Round 5's window follows the pair's rate of rise AND the variance its
error rate implies (align16_body.inc, widen_window); these tests pin what that bought: nothing is started over on clean C1
pairs, almost nothing on C0, the results are the oracle's, and the CPU model of the kernel's decisions
(oracle/agatha_lanes_model.c) tells the same story as the chip's counters on the same batch."""
import numpy as np
import pytest

from oracle import oracle as O, synth

pytestmark = pytest.mark.gpu

REF = dict(m=1, x=4, q=6, r=2, s=3, z=400, w=751)          # AGAThA.sh:44


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
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p)); b.download(); eng.synchronize()
        got = [b.res_host[j].copy() for j in range(3)]
        info = dict(choice=b.kernel_choice(), sched=b.schedule_info(), st=b.step_stats(), kinds=b.pair_kinds())
    finally:
        b.free()
    return (qb, tb, qo, to, ql, tl), got, info


def test_clean_c1_pairs_at_the_reference_scoring_are_never_started_over(eng):
    qs, ts = synth.cfg_c1(n=10000)
    batch, got, info = _run(eng, qs, ts, **REF)
    st = info["st"]
    assert info["choice"][0] == "int16" and info["sched"][0], info       # the headline path: packed int16, static schedule
    assert st[2] == 0 and st[15] == 0, f"pairs started over {st[2]}, taken back to a checkpoint {st[15]}"
    assert info["kinds"][2] == 0                                          # nothing left for the int32 kernel
    # key wave-steps stay a small share although every pair now ends on ~170 of them (the intervals of the schedule are dealt so
    # that the pair ends of a wave coincide: schedule_kernel)
    assert st[1] < 0.12 * (st[0] + st[1]), st[:2]
    pick = np.sort(np.random.default_rng(5).choice(10000, 1500, replace=False))
    sub = [np.ascontiguousarray(a[pick]) for a in batch[2:]]
    exp = O.align_batch(batch[0], batch[1], *sub, O.make_params(**REF), wide=True, model=O.MODEL_SLICES, threads=16)
    assert all((g[pick] == e).all() for g, e in zip(got, exp))


def test_clean_c0_pairs_at_the_reference_scoring(eng):
    qs, ts = synth.cfg_c0(n=20000)
    batch, got, info = _run(eng, qs, ts, **REF)
    st = info["st"]
    assert info["choice"][0] == "int16"
    # <= 0.2 % (round 4: 3.6 %); what is left are reads whose score peaks well before their end without z-dropping
    assert st[2] + st[15] <= 40, f"pairs started over {st[2]}, taken back to a checkpoint {st[15]} of 20 000"
    pick = np.sort(np.random.default_rng(6).choice(20000, 3000, replace=False))
    sub = [np.ascontiguousarray(a[pick]) for a in batch[2:]]
    exp = O.align_batch(batch[0], batch[1], *sub, O.make_params(**REF), wide=True, model=O.MODEL_SLICES, threads=16)
    assert all((g[pick] == e).all() for g, e in zip(got, exp))


def test_the_lane_model_and_the_chip_agree_on_who_is_started_over(eng):
    """The same 4 000 C0 pairs through the kernel and through the CPU model of its decisions (oracle/agatha_lanes_model.c), twice: with
    round 5's window, and with the window capped at 45 steps -- about what round 4's rule gave at this scoring -- so that there IS
    something to count.  Both must see (almost) nothing started over with the first and a like number with the second (the kernel
    looks at a pair's window when its WAVE's step counter passes a multiple of 64, the model when the pair's own does: not the same
    pairs, the same rate); the results never depend on the window."""
    import ctypes as C
    import agatha_amd
    qs, ts = synth.cfg_c0(n=4000)
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    p = O.make_params(**REF)
    lib = O.lib()
    cap_min, cap_div = C.c_int.in_dll(lib, "agatha_lanes16_win_cap_min"), C.c_int.in_dll(lib, "agatha_lanes16_win_cap_div")
    saved = (cap_min.value, cap_div.value)
    model = {}
    try:
        for name, (cm, cd) in (("round5", saved), ("narrow", (45, 1 << 20))):
            cap_min.value, cap_div.value = cm, cd
            model[name] = int((O.lanes16_batch(qb, tb, qo, to, ql, tl, p, 16, 6, threads=16, value_step_margin=12)[3] == 2).sum())
    finally:
        cap_min.value, cap_div.value = saved
    exp = O.align_batch(qb, tb, qo, to, ql, tl, p, wide=True, model=O.MODEL_SLICES, threads=16)
    chip = {}
    defaults = {k: agatha_amd.get_debug_option(k) for k in ("win_cap_min", "win_cap_div", "flat_detect")}
    try:
        # (flat_detect off: with a cap of 45 steps every pair would need "more than four times the window it may have", and the batch
        #  would be taken for one whose scores do not rise -- what is counted here are the decisions of single pairs)
        agatha_amd.set_debug_option("flat_detect", 0)
        for name, (cm, cd) in (("round5", (defaults["win_cap_min"], defaults["win_cap_div"])), ("narrow", (45, 1 << 20))):
            agatha_amd.set_debug_option("win_cap_min", cm); agatha_amd.set_debug_option("win_cap_div", cd)
            _, got, info = _run(eng, qs, ts, **REF)
            assert all((g == e).all() for g, e in zip(got, exp)), name          # results never depend on the window
            chip[name] = int(info["st"][2] + info["st"][15])
    finally:
        for k, v in defaults.items():
            agatha_amd.set_debug_option(k, v)
    assert model["round5"] <= 8 and chip["round5"] <= 8, (model, chip)
    # (133 and 133 when this was written: the model decides what the kernel decides)
    assert model["narrow"] >= 60 and abs(chip["narrow"] - model["narrow"]) <= max(10, model["narrow"] // 4), (model, chip)


def test_a_flat_batch_is_noticed_and_runs_on_key_steps(eng):
    """Reads whose score hardly rises at the scoring (15 % errors at match 1: the expected gain per base is negative) have no window
    that holds the last rise of their maximum; half of them used to end without its cell, go back to a checkpoint, fail again and
    start from their first step (a 10 kb batch: 66 ms against 27 for clean reads, where key steps all along take 35).  The kernel
    notices: the pairs say at their 64th..127th step whether they are flat, and when 30 % are the young pairs start over on key
    steps and later pairs start on them.  Here: 9 000 pairs of ~5 kb at 15 % errors on the static schedule -- most say flat, most
    are restarted young, (almost) nobody fails late, results = the oracle's; the clean batch beside it is left alone."""
    import agatha_amd
    qs, ts = synth.make_pairs(77, 9000, lambda r: int(np.clip(np.rint(r.normal(5000, 400)), 4000, 6000)), 0.045, 0.045, 0.06)
    batch, got, info = _run(eng, qs, ts, **REF)
    st = info["st"]
    qb, tb, qo, to, ql, tl = batch
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**REF)); eng.synchronize()
        flat = b.flat_stats()
    finally:
        b.free()
    assert info["choice"][0] == "int16" and info["sched"][0]
    assert flat[1] > 0.3 * flat[0] and flat[0] > 4000, flat                # most pairs say: flat
    assert flat[2] > 4000, flat                                            # ... and the young ones started over on key steps
    assert st[15] + max(st[2] - flat[2], 0) < 0.03 * len(qs), (st[:16], flat)     # hardly anybody fails late any more
    assert st[1] > 5 * st[0]                                               # the batch ran on key steps
    pick = np.sort(np.random.default_rng(8).choice(len(qs), 1200, replace=False))
    sub = [np.ascontiguousarray(a[pick]) for a in batch[2:]]
    exp = O.align_batch(batch[0], batch[1], *sub, O.make_params(**REF), wide=True, model=O.MODEL_SLICES, threads=16)
    assert all((g[pick] == e).all() for g, e in zip(got, exp))
    # the clean batch of the same shape: nobody is restarted
    qs2, ts2 = synth.make_pairs(78, 9000, lambda r: int(np.clip(np.rint(r.normal(5000, 400)), 4000, 6000)), 0.03, 0.03, 0.04)
    batch2, got2, info2 = _run(eng, qs2, ts2, **REF)
    b = eng.batch(*batch2)
    try:
        b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**REF)); eng.synchronize()
        flat2 = b.flat_stats()
    finally:
        b.free()
    assert flat2[2] == 0 and flat2[1] < 0.2 * max(flat2[0], 1), flat2
    assert info2["st"][2] + info2["st"][15] <= 9, info2["st"][:16]


def test_a_read_with_a_burst_of_errors_returns_to_value_steps(eng):
    """A 350-base stretch of 40 % errors in every tenth 10 kb read: at match 1 the score falls by a few hundred inside it, z-drop (400)
    comes within reach of what a value step knows, the pair goes back to a checkpoint and walks the dip on key steps -- and, round 5,
    returns to value steps once z-drop is comfortably out of reach again ("probation"; until then it stayed on key steps to its end:
    34 instead of 27 ms for the batch).  Here: most pairs that go back do return, the batch runs fewer key steps than with the debug
    option probation = 0, and the results are the oracle's either way."""
    import agatha_amd
    qs, ts0 = synth.cfg_c1(n=10000)
    rng = np.random.default_rng(11)
    ts, hit = [], []
    for j, t in enumerate(ts0):
        if rng.random() < 0.10:
            a = np.frombuffer(t, np.uint8).copy()
            at = int(rng.integers(len(a) // 5, len(a) * 4 // 5 - 350))
            t = np.concatenate([a[:at], synth.mutate(rng, a[at:at + 350], 0.15, 0.12, 0.13), a[at + 350:]]).tobytes()
            hit.append(j)
        ts.append(t)
    runs = {}
    try:
        for on in (1, 0):
            agatha_amd.set_debug_option("probation", on)
            runs[on] = _run(eng, qs, ts, **REF)
    finally:
        agatha_amd.set_debug_option("probation", 1)
    batch, got, info = runs[1]
    st, st_off = info["st"], runs[0][2]["st"]
    assert info["choice"][0] == "int16" and info["sched"][0]
    assert st[15] > 300 and st_off[15] > 300, (st[:16], st_off[:16])               # hundreds of pairs go back to a checkpoint
    assert st[38] > 0.3 * st[15] and st_off[38] == 0, (st[:16], st_off[:16])        # ... and a good part of them leave their probation (the CPU model: 60 %)
    assert st[1] < 0.85 * st_off[1], (st[:2], st_off[:2])                           # fewer key wave-steps for it
    r = np.random.default_rng(9)
    pick = np.sort(np.concatenate([r.choice(np.array(hit), 700, replace=False),
                                   r.choice(np.setdiff1d(np.arange(10000), np.array(hit)), 500, replace=False)]))
    sub = [np.ascontiguousarray(a[pick]) for a in batch[2:]]
    exp = O.align_batch(batch[0], batch[1], *sub, O.make_params(**REF), wide=True, model=O.MODEL_SLICES, threads=16)
    for on in (1, 0):
        assert all((g[pick] == e).all() for g, e in zip(runs[on][1], exp)), on


@pytest.mark.parametrize("burst", [150, 350])
def test_bundled_dataset_shape_with_bursts_of_errors_at_the_reference_scoring(eng, burst):
    """Round 6: the same bursts in the 3 kb reads of the bundled-dataset shape (configs[0], the work queue: 2.4 rounds of pairs), where a read has
    one checkpoint behind it or none -- dozens to hundreds of pairs start from their first step, and the ones that do so after the queue has run
    dry are the kernel's tail (profiles/r06_v1/bursts.txt: + 25 ... + 33 % on the kernel).  Such a pair now leaves for the clean-up launch of the
    latency shape (one pair per wave behind the kernel) instead of running 750 key steps alone.  Results: the oracle's, with the launch and
    without it (debug option cleanup_min_steps = 0)."""
    import agatha_amd
    qs, ts0 = synth.cfg_c0(n=20000)
    rng = np.random.default_rng(11)
    ts, hit = [], []
    for j, t in enumerate(ts0):
        if rng.random() < 0.10 and len(t) * 4 // 5 - burst > len(t) // 5:
            a = np.frombuffer(t, np.uint8).copy()
            at = int(rng.integers(len(a) // 5, len(a) * 4 // 5 - burst))
            t = np.concatenate([a[:at], synth.mutate(rng, a[at:at + burst], 0.15, 0.12, 0.13), a[at + burst:]]).tobytes()
            hit.append(j)
        ts.append(t)
    runs = {}
    try:
        for cl in (384, 0):
            agatha_amd.set_debug_option("cleanup_min_steps", cl)
            runs[cl] = _run(eng, qs, ts, **REF)
    finally:
        agatha_amd.set_debug_option("cleanup_min_steps", 384)
    batch, got, info = runs[384]
    assert info["choice"][0] == "int16" and not info["sched"][0], info          # the work queue
    st = info["st"]
    assert st[2] + st[15] >= 20, st[:16]                                          # pairs do start over or go back: the case is the one meant
    r = np.random.default_rng(9)
    pick = np.sort(np.concatenate([r.choice(np.array(hit), min(600, len(hit)), replace=False),
                                   r.choice(np.setdiff1d(np.arange(20000), np.array(hit)), 600, replace=False)]))
    sub = [np.ascontiguousarray(a[pick]) for a in batch[2:]]
    exp = O.align_batch(batch[0], batch[1], *sub, O.make_params(**REF), wide=True, model=O.MODEL_SLICES, threads=16)
    for cl in (384, 0):
        assert all((g[pick] == e).all() for g, e in zip(runs[cl][1], exp)), cl
    assert all((a == b).all() for a, b in zip(runs[384][1], runs[0][1]))        # ... and all 20 000 agree between the two runs


def _dip_then_break(n=96, seed=23):
    """30 kb reads with a 250-base burst of errors in the middle (the score dips by ~200 and recovers) and, 550 bases behind it, an unrelated
    tail (z-drop ends the pair there): two give-ups less than a checkpoint span apart, the first one survivable."""
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for _ in range(n):
        L = int(rng.integers(28000, 32000))
        ref = synth.random_seq(rng, L)
        rd = synth.mutate(rng, ref, 0.03, 0.03, 0.04)
        at = int(rng.integers(L * 2 // 5, L * 3 // 5))
        a = rd.copy()
        seg = synth.mutate(rng, a[at:at + 250], 0.15, 0.12, 0.13)
        brk = at + 250 + int(rng.integers(500, 700))
        rd2 = np.concatenate([a[:at], seg, a[at + 250:brk], synth.random_seq(rng, max(L - brk, 64))])
        qs.append(ref.tobytes()); ts.append(rd2.tobytes())
    return qs, ts


def test_a_pair_that_went_back_once_finds_its_checkpoints_again(eng):
    """Round 6 (profiles/r06_v2/c4_book_lost.txt): on the shapes that keep book of their checkpoints in a register (one or two register pairs
    per lane: the latency shapes, HiFi bands) going back to a checkpoint emptied the book; a pair that then left its probation and gave up a
    second time before the next checkpoint was due started from its FIRST step although both slots held a checkpoint of it -- the one
    clean-then-broken 100 kb read of BASELINE configs[4] ran 42 000 steps, 109 instead of 63 ms for the batch.  Here: a dip, a recovery, a
    break within one span; nobody starts from the first step far into the pair, and the results are the oracle's."""
    import agatha_amd
    qs, ts = _dip_then_break()
    agatha_amd.set_debug_option("force_choice", 1)                 # the int16 latency shape: one pair per wave, book-keeping checkpoints
    try:
        batch, got, info = _run(eng, qs, ts, **REF)
    finally:
        agatha_amd.set_debug_option("force_choice", -1)
    assert info["choice"][0] == "int16" and info["choice"][1] >= 64, info
    st = info["st"]
    assert st[15] >= len(qs) // 2, st[:16]                         # the dips send pairs back to a checkpoint ...
    assert st[2] == 0, st[:16]                                     # ... and the breaks behind them do not send anybody to the first step (round 5's library: 1 of these 96)
    exp = O.align_batch(*batch, O.make_params(**REF), wide=True, model=O.MODEL_SLICES, threads=16)
    assert all((g == e).all() for g, e in zip(got, exp))

