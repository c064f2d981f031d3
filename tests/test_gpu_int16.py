"""GPU parity of the packed-int16 kernel (agatha_amd/csrc/align16_kernel.hip) and of the routing between it, the
int32 profile kernel and the compare kernel.  Needs a real MI355X: `pytest -m gpu`."""
import os

import numpy as np
import pytest

import agatha_amd
from oracle import oracle as O
from agatha_amd import workload as WL

pytestmark = pytest.mark.gpu

BASE = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)


@pytest.fixture(scope="module")
def eng():
    import agatha_amd
    agatha_amd.set_debug_option("force_int16", 1)      # the launcher leaves small batches to the int32 kernel otherwise
    e = agatha_amd.Engine(0)
    yield e
    e.close()
    agatha_amd.set_debug_option("force_int16", 0)


def test_device_side_kernel_choice(eng):
    """Without the override the device picks the kernel from the length histogram: a small batch is latency-bound (64
    lanes per pair, one register pair per lane), a large uniform one throughput-bound (16 lanes per pair, three)."""
    import agatha_amd
    agatha_amd.set_debug_option("force_int16", 0)
    try:
        for n, expect in ((64, ("int16", 64, 2)), (9000, ("int16", 16, 6))):
            qs, ts = WL.make_pairs(3, n, lambda r: int(r.integers(900, 1100)), 0.03, 0.03, 0.04)
            qb, qo, ql = WL.make_batch(qs)
            tb, to, tl = WL.make_batch(ts)
            b = eng.batch(qb, tb, qo, to, ql, tl)
            try:
                b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**BASE), use_len_hint=False)
                assert b.kernel_choice() == expect
            finally:
                b.free()
    finally:
        agatha_amd.set_debug_option("force_int16", 1)


def _run(eng, qs, ts, p):
    import agatha_amd
    qb, qo, ql = WL.make_batch(qs)
    tb, to, tl = WL.make_batch(ts)
    got = eng.align_host_batch(qb, tb, qo, to, ql, tl, agatha_amd.Scores.make(**p))
    exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=8)
    return got, exp


def _run_with_kinds(eng, qs, ts, p):
    """Like _run, plus how the pairs were routed (plain, other letters, taken over by the int32 kernel)."""
    import agatha_amd
    qb, qo, ql = WL.make_batch(qs)
    tb, to, tl = WL.make_batch(ts)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=8)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p)); b.download(); eng.synchronize()
        got = tuple(b.res_host[k].copy() for k in range(3))
        kinds = b.pair_kinds()
    finally:
        b.free()
    return got, exp, kinds


def _same(got, exp):
    return all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(got, exp))


def _expected_int16_config(p):
    """(lanes per pair, slots per lane) the launcher must choose, or None (agatha_amd/csrc/align16_kernel.hip: pick16)."""
    W = (p["w"] + 7) // 8
    per = max(2 * p["r"], p["x"], 1)
    if p["w"] < 16 or p["q"] + p["r"] + per * (p["w"] + 16) + 64 > 16000:
        return None
    for G, S in ((16, 2), (16, 4), (16, 6), (32, 4), (32, 6)):
        if G * S >= W + 1:
            return (G, S) if G * S <= 2 * (W + 1) + 32 else None
    for G, S in ((64, 4), (64, 6)):             # (round 4) windows of 193..384 blocks: one pair per wave as the throughput shape
        if G * S >= W + 1:
            return (G, S)
    return None


@pytest.mark.parametrize("p", [BASE, dict(BASE, w=760), dict(BASE, s=1), dict(BASE, z=-1), dict(BASE, z=0),
                               dict(m=1, x=4, q=6, r=2, s=2, z=100, w=760), dict(m=16, x=32, q=64, r=16, s=3, z=2000, w=103),
                               dict(BASE, w=500), dict(BASE, w=100, s=2), dict(BASE, w=1000), dict(BASE, w=1400, z=1000),
                               dict(BASE, w=1500), dict(BASE, w=2000), dict(BASE, w=15)],
                         ids=lambda p: "m%dx%dq%dr%ds%dz%dw%d" % (p["m"], p["x"], p["q"], p["r"], p["s"], p["z"], p["w"]))
def test_int16_kernel_matches_oracle(eng, p):
    """Cut diagonals, windows (16x2 ... 32x6 slots), slice widths, z-drop on/off/immediate, a scoring at the edge of the
    int16 kernel's domain, the wide windows of round 4 (w = 2000: <64, 2>), and a band the launcher must leave to the int32
    kernel (w = 15: too narrow)."""
    hi = 14000 if p["w"] <= 1000 else 30000
    qs, ts = WL.cfg_c4(n=160 if p["w"] <= 1000 else 96, seed=31 + p["w"], lo=50, hi=hi)
    got, exp = _run(eng, qs, ts, p)
    assert _same(got, exp)
    assert eng.last_int16_config() == _expected_int16_config(p)


@pytest.mark.parametrize("w", list(range(745, 753)) + [97, 250, 505])
def test_every_cut_diagonal(eng, w):
    """w mod 8 decides which cell diagonal of an edge block is cut (one compiled kernel per value), and for w mod 8 in
    1..6 the blocks next to the corners of the band are cut as well."""
    p = dict(BASE, w=w, s=int(1 + w % 3))
    qs, ts = WL.cfg_c4(n=128, seed=1000 + w, lo=30, hi=12000)
    got, exp, kinds = _run_with_kinds(eng, qs, ts, p)
    assert _same(got, exp)
    assert eng.last_int16_config() == _expected_int16_config(p)
    # z-drop is on and no sequence holds an N: the int16 kernel must finish every one of these pairs itself (a hand-back
    # would hide a block the kernel cannot classify behind the int32 kernel's correct answer)
    assert kinds == (len(qs), 0, 0)
    # matrices around the band's corners, and pairs whose band leaves the matrix through its side (these may be handed
    # back: anti-diagonals on which only padded columns remain)
    rng = np.random.default_rng(w)
    qs, ts = [], []
    for L in (7, 8, 9, 8 * ((w + 7) // 8) - 1, 8 * ((w + 7) // 8) + 1, w, w + 1, 2 * w, 2 * w + 9):
        a = WL.random_seq(rng, L)
        qs.append(a.tobytes()); ts.append(WL.mutate(rng, a, 0.02, 0.02, 0.02).tobytes() or b"A")
        qs.append(a.tobytes()); ts.append(np.concatenate([a, WL.random_seq(rng, w + 40)]).tobytes())
        qs.append(np.concatenate([a, WL.random_seq(rng, w + 40)]).tobytes()); ts.append(a.tobytes())
    got, exp = _run(eng, qs, ts, p)
    assert _same(got, exp)


def test_int16_and_int32_kernels_agree(eng):
    import agatha_amd
    qs, ts = WL.cfg_c1(n=300, seed=77)
    qb, qo, ql = WL.make_batch(qs)
    tb, to, tl = WL.make_batch(ts)
    sc = agatha_amd.Scores.make(**BASE)
    a = eng.align_host_batch(qb, tb, qo, to, ql, tl, sc)
    assert eng.last_int16_config() == (16, 6)
    agatha_amd.set_debug_option("no_int16", 1)
    try:
        b = eng.align_host_batch(qb, tb, qo, to, ql, tl, sc)
        assert eng.last_int16_config() is None
    finally:
        agatha_amd.set_debug_option("no_int16", 0)
    assert _same(a, b)


def test_rebasing_long_pairs(eng):
    """Scores far beyond int16 (2 x 60 kb): the representation is rebased ~60 times per pair."""
    f = lambda rng: int(rng.integers(30000, 60001))
    qs, ts = WL.make_pairs(5, 24, f, 0.01, 0.01, 0.01)
    got, exp = _run(eng, qs, ts, BASE)
    assert _same(got, exp)
    assert int(np.max(exp[0])) > 40000


@pytest.mark.parametrize("p", [dict(m=16, x=32, q=64, r=16, s=3, z=400, w=103), dict(m=16, x=1, q=0, r=16, s=2, z=-1, w=250),
                               dict(m=1, x=4, q=6, r=0, s=3, z=400, w=751), dict(m=2, x=4, q=4, r=2, s=3, z=-1, w=751),
                               dict(m=1, x=32, q=64, r=1, s=3, z=-1, w=97), dict(m=1, x=19, q=39, r=3, s=3, z=400, w=751),
                               dict(m=2, x=12, q=24, r=2, s=3, z=1000, w=1000), dict(m=2, x=32, q=64, r=2, s=2, z=400, w=400)],
                         ids=lambda p: "m%dx%dq%dr%ds%dz%dw%d" % (p["m"], p["x"], p["q"], p["r"], p["s"], p["z"], p["w"]))
def test_drifting_frame_extremes(eng, p):
    """The int16 kernel sees every value from its own anti-diagonal (+ ge per anti-diagonal): with the steepest scores
    the representation is rebased every ~10 steps of a 2 x 30 kb pair, with r = 0 the frame does not move at all, and
    with z-drop off on noisy and broken pairs the values fall while the frame rises; with steep mismatch penalties
    (in-band cells up to 16 000 below their anti-diagonal's maximum) the in-band zone starts lifted."""
    rng = np.random.default_rng(77)
    qs, ts = WL.make_pairs(11, 10, lambda r: int(r.integers(20000, 30001)), 0.01, 0.01, 0.01)
    q2, t2 = WL.make_pairs(12, 10, lambda r: int(r.integers(3000, 9000)), 0.12, 0.08, 0.08)
    q3, t3 = WL.cfg_c4(n=40, seed=13, lo=100, hi=12000)
    got, exp, kinds = _run_with_kinds(eng, qs + q2 + q3, ts + t2 + t3, p)
    assert _same(got, exp)
    assert eng.last_int16_config() == _expected_int16_config(p)


def test_steep_scores_stay_on_the_int16_kernel(eng):
    """minimap2's asm5 scoring at band 751 (in-band cells up to 14 679 below a maximum): good pairs are finished by the
    int16 kernel itself, with the in-band zone lifted; nothing is handed back."""
    p = dict(m=1, x=19, q=39, r=3, s=3, z=400, w=751)
    qs, ts = WL.make_pairs(21, 48, lambda r: int(r.integers(6000, 14001)), 0.004, 0.002, 0.002)
    got, exp, kinds = _run_with_kinds(eng, qs, ts, p)
    assert _same(got, exp)
    assert eng.last_int16_config() == (16, 6)
    assert tuple(int(v) for v in kinds) == (48, 0, 0)


def test_pairs_the_int16_kernel_hands_back(eng):
    """z-drop switched off on unrelated sequences: the scores sink until they come within `spread` of the reference's
    -infinity, where the int16 kernel abandons the pair (bail-out) and the int32 kernel redoes it (in the kernel's
    drifting frame the representation itself hardly sinks: an anti-diagonal maximum loses at most the gap-extension
    score per anti-diagonal, which is what the frame adds); pairs whose band leaves the matrix through its side
    (lengths differing by more than the band) produce empty anti-diagonals."""
    rng = np.random.default_rng(9)
    qs = [WL.random_seq(rng, 20000).tobytes() for _ in range(12)]
    ts = [WL.random_seq(rng, 20000).tobytes() for _ in range(12)]
    base = WL.random_seq(rng, 6000)
    for extra in (800, 1600, 3000):                      # |Q - R| > band
        qs.append(base.tobytes()); ts.append(np.concatenate([base, WL.random_seq(rng, extra)]).tobytes())
        qs.append(np.concatenate([base, WL.random_seq(rng, extra)]).tobytes()); ts.append(base.tobytes())
    for p in (dict(BASE, z=-1), BASE, dict(BASE, w=760, z=30000)):
        got, exp = _run(eng, qs, ts, p)
        assert _same(got, exp)
    # with z-drop off the twelve unrelated pairs must really have been handed back
    import agatha_amd
    qb, qo, ql = WL.make_batch(qs)
    tb, to, tl = WL.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**dict(BASE, z=-1)))
        plain, other, takeover = b.pair_kinds()
    finally:
        b.free()
    assert other == 0 and takeover >= 12 and plain + takeover == len(qs)


def test_n_in_query_and_other_letters_are_routed(eng):
    """N on either side stays on the int16 kernel (N in the query -- the DP rows -- through the block variant that knows N
    rows, on key steps); letters outside ACGTN go to the compare kernel."""
    rng = np.random.default_rng(4)
    qs, ts = [], []
    for k in range(30):
        ref = WL.random_seq(rng, int(rng.integers(500, 6000)))
        rd = WL.mutate(rng, ref, 0.03, 0.03, 0.04)
        ref, rd = ref.copy(), rd.copy()
        if k % 3 == 0:
            ref[rng.random(ref.size) < 0.02] = ord("N")
        if k % 3 == 1:
            rd[rng.random(rd.size) < 0.02] = ord("N")
        if k % 5 == 4:
            rd[rng.integers(0, rd.size)] = ord("R")
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    got, exp, kinds = _run_with_kinds(eng, qs, ts, BASE)
    assert _same(got, exp)
    assert kinds[1] == 6 and kinds[2] == 0          # the six pairs with an R: compare kernel; nothing to the int32 profile kernel


@pytest.mark.parametrize("p", [BASE, dict(BASE, w=100, z=60), dict(m=1, x=4, q=6, r=2, s=1, z=400, w=500), dict(BASE, w=1500)])
def test_n_runs_in_the_query_on_the_int16_kernel(eng, p):
    """N in the DP-row sequence (the reference scores it in line, gasal_kernels.h:48-50; `AGAThA.sh:44` passes ref.fasta first, so
    the rows are reference-genome pieces, where N runs live): runs of N of every length and phase against the 8-row blocks,
    at the start, in the middle and at the ragged end of the query, with N in the target as well, on the throughput and the
    latency shapes.  The pairs stay on the int16 kernel (no take-over) and the wave returns to value steps behind a run."""
    rng = np.random.default_rng(17)
    qs, ts = [], []
    for k in range(90):
        ref = WL.random_seq(rng, int(rng.integers(300, 5000)))
        rd = WL.mutate(rng, ref, 0.03, 0.03, 0.04).copy()
        ref = ref.copy()
        if k % 6 != 5:
            for _ in range(int(rng.integers(1, 4))):
                a = int(rng.integers(0, ref.size)); n = int(rng.choice([1, 2, 7, 8, 9, 40, 300]))
                ref[a:a + n] = ord("N")
        if k % 4 == 0:
            ref[-int(rng.integers(1, 12)):] = ord("N")          # N up to the ragged end of the query
        if k % 7 == 0:
            rd[rng.random(rd.size) < 0.01] = ord("N")
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    got, exp, st, kinds = _run_stats(eng, qs, ts, p)
    assert _same(got, exp)
    assert kinds[1] == 0 and kinds[2] == 0
    assert st[0] > 0 and st[1] > 0                  # value steps, and key steps where N rows were in flight
    # the latency shape of the window (64 lanes per pair; one pair on two waves at band 1500, where a pair that holds an N stays on
    # key steps: whether an N row is in flight differs between the two waves, and they must run the same kind of step)
    agatha_amd.set_debug_option("force_int16", 0)
    agatha_amd.set_debug_option("force_choice", 1)
    try:
        got, exp, st, kinds = _run_stats(eng, qs, ts, p, fast_margin=16)
        assert _same(got, exp) and kinds[2] == 0
        if p["w"] == 1500:
            assert _run_stats.choice == ("int16", 128, 2)
        elif p["w"] >= 500:
            assert _run_stats.choice == ("int16", 64, 2)
    finally:
        agatha_amd.set_debug_option("force_choice", -1)
        agatha_amd.set_debug_option("force_int16", 1)


@pytest.mark.parametrize("n", [4100, 5000, 8000, 8185, 8192, 8200, 9000, 12288, 12300, 15000, 16384, 16400])
def test_first_round_dealt_to_the_workgroups(eng, n):
    """With a full grid (8192 lane groups for a narrow band) and between 1 and 2 rounds of pairs the int16 kernel deals
    the first round to the workgroups by formula (which waves share a SIMD; two formulas, below and above 1.5 rounds)
    and takes the rest from the queue; pairs the kernel must skip (other letters) and pairs with an N in the query sit in the
    dealt range as well.  16 400 pairs: plain queue.  Below one round, with more workgroups than CUs, everything is dealt (the CUs
    with one workgroup take the longest chunks)."""
    rng = np.random.default_rng(n)
    qs, ts = [], []
    for k in range(n):
        ref = WL.random_seq(rng, int(rng.integers(40, 260)))
        rd = WL.mutate(rng, ref, 0.03, 0.03, 0.04)
        if rd.size == 0:
            rd = WL.random_seq(rng, 1)
        ref, rd = ref.copy(), rd.copy()
        if k % 97 == 5:
            ref[rng.integers(0, ref.size)] = ord("N")          # N in the query (file 1): stays here, on key steps
        if k % 89 == 7:
            rd[rng.integers(0, rd.size)] = ord("R")            # other letter: compare kernel
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    p = dict(BASE, w=24)
    agatha_amd.set_debug_option("force_choice", 0)             # the int16 throughput shape <16,1>: 16 groups per workgroup
    try:
        got, exp, kinds = _run_with_kinds(eng, qs, ts, p)
    finally:
        agatha_amd.set_debug_option("force_choice", -1)
    assert _same(got, exp)
    assert int(kinds[1]) > 0 and int(kinds[2]) == 0


def test_first_round_dealt_in_the_latency_shape(eng):
    """2100 pairs at band 751: the device picks the 64-lane shape (2048 lane groups), whose first round is dealt as well."""
    qs, ts = WL.make_pairs(77, 2100, lambda r: int(r.integers(1500, 3500)), 0.03, 0.03, 0.04)
    got, exp, kinds = _run_with_kinds(eng, qs, ts, BASE)
    assert _same(got, exp)
    assert tuple(int(v) for v in kinds) == (2100, 0, 0)


def test_ragged_lengths_around_block_edges(eng):
    """Every query length modulo 8 (rows that do not exist in the last row block) against every target length modulo 8
    (padded reference columns)."""
    rng = np.random.default_rng(12)
    qs, ts = [], []
    for dq in range(8):
        for dt in range(8):
            ref = WL.random_seq(rng, 1600 + dq)
            rd = WL.mutate(rng, ref, 0.02, 0.02, 0.02)
            rd = rd[:max(1, (rd.size // 8) * 8 - 8 + dt)]
            qs.append(ref.tobytes()); ts.append(rd.tobytes())
    for p in (BASE, dict(BASE, w=760, s=2)):
        got, exp = _run(eng, qs, ts, p)
        assert _same(got, exp)


def test_pairs_longer_than_the_16_bit_block_indices(eng):
    """The int16 kernel keeps block indices in 16 bits (sequences up to 262 k bases); a longer pair is handed to the
    int32 kernel when it is drawn from the queue."""
    import agatha_amd
    p = dict(m=1, x=4, q=6, r=2, s=3, z=400, w=751)
    rng = np.random.default_rng(77)
    big = WL.random_seq(rng, 270000)
    qs = [big.tobytes()] + [WL.random_seq(rng, 3000).tobytes() for _ in range(6)]
    ts = [WL.mutate(rng, big, 0.01, 0.01, 0.01).tobytes()] + [WL.mutate(rng, np.frombuffer(q, np.uint8), 0.03, 0.03, 0.04).tobytes() for q in qs[1:]]
    got, exp = _run(eng, qs, ts, p)
    assert _same(got, exp)
    assert int(exp[0][0]) > 200000
    qb, qo, ql = WL.make_batch(qs)
    tb, to, tl = WL.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p))
        assert b.kernel_choice()[0] == "int16" and b.pair_kinds() == (6, 0, 1)
    finally:
        b.free()


def _mig_batch(n, seed, lo, hi, broken=0.0):
    """n pairs for a narrow band; `broken`: share of pairs whose read turns into an unrelated sequence somewhere (z-drop)."""
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for k in range(n):
        ref = WL.random_seq(rng, int(rng.integers(lo, hi)))
        rd = WL.mutate(rng, ref, 0.03, 0.03, 0.04)
        if rng.random() < broken:
            bp = int(rng.integers(0, max(1, rd.size)))
            rd = np.concatenate([rd[:bp], WL.random_seq(rng, ref.size - min(bp, ref.size) + 8)])
        if rd.size == 0:
            rd = WL.random_seq(rng, 1)
        ref, rd = ref.copy(), rd.copy()
        if k % 97 == 5:
            ref[rng.integers(0, ref.size)] = ord("N")          # N in the query: the int32 kernel's pair, skipped here
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    return qs, ts


def _run_scheduled(eng, qs, ts, p):
    qb, qo, ql = WL.make_batch(qs)
    tb, to, tl = WL.make_batch(ts)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=8)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p)); b.download()
        eng.synchronize()
        got = [b.res_host[k].copy() for k in range(3)]
        return got, exp, b.schedule_info(), b.kernel_choice()
    finally:
        b.free()


@pytest.mark.parametrize("n,w,lo,hi", [(8200, 24, 300, 1200), (9500, 24, 40, 900), (12288, 24, 200, 700), (15000, 40, 100, 800),
                                       (9000, 300, 500, 1500), (20000, 24, 100, 400)])
def test_pairs_migrate_between_lane_groups(eng, n, w, lo, hi):
    """More pairs than lane groups (8192 for these bands): the int16 kernel runs a static preemptive schedule -- every lane
    group executes T = max(longest pair, total steps / groups) steps, and the pair that crosses a group boundary is
    suspended by one group and resumed by its neighbour (the reference's subwarp rejoining, agatha_kernel.h:365-408,
    re-derived: pairs in flight move to lane groups that would idle).  Bit-exact against the oracle, and identical to the
    run with the work queue."""
    qs, ts = _mig_batch(n, n + w, lo, hi)
    p = dict(BASE, w=w)
    agatha_amd.set_debug_option("force_choice", 0)             # the int16 throughput shape
    try:
        got, exp, info, choice = _run_scheduled(eng, qs, ts, p)
        assert choice[0] == "int16" and choice[1] == 16
        assert info[0] and info[2] <= 8192 and info[1] >= 1, info
        assert _same(got, exp)
        with agatha_amd.debug_options(no_migrate=1):
            got2, _, info2, _ = _run_scheduled(eng, qs, ts, p)
        assert not info2[0] and _same(got2, exp)
    finally:
        agatha_amd.set_debug_option("force_choice", -1)


def test_migration_with_zdrop_in_the_first_part(eng):
    """Pairs that end (z-drop) inside the part their first lane group runs: the second group must find nothing to resume."""
    qs, ts = _mig_batch(10000, 5, 200, 1500, broken=0.5)
    p = dict(BASE, w=24, z=60)
    agatha_amd.set_debug_option("force_choice", 0)
    try:
        got, exp, info, _ = _run_scheduled(eng, qs, ts, p)
        assert info[0]
        assert _same(got, exp)
    finally:
        agatha_amd.set_debug_option("force_choice", -1)


def test_migration_take_over_when_the_first_part_never_comes(eng):
    """A lane group that waits too long for the pair its neighbour has to suspend takes the pair over and runs it from its
    first step (deadlock-freedom when the neighbour's workgroup is not resident).  Forced here: odd lane groups start 20 ms
    late, the wait is cut to 1 ms."""
    qs, ts = _mig_batch(9000, 6, 200, 900)
    p = dict(BASE, w=24)
    agatha_amd.set_debug_option("force_choice", 0)
    try:
        with agatha_amd.debug_options(mig_timeout_us=1000, mig_test_delay_us=20000):
            got, exp, info, _ = _run_scheduled(eng, qs, ts, p)
        assert info[0]
        assert _same(got, exp)
    finally:
        agatha_amd.set_debug_option("force_choice", -1)


@pytest.mark.parametrize("mode", [3, 1003])
def test_a_poisoned_saved_state_ends_in_the_int32_kernel_not_in_a_hang(eng, mode):
    """Round 6, the hard bounds of the int16 kernel.  In round 5 an unchecked saved state sent a wave to step 2 109 373 922 and the GPU suite
    hung (profiles/r05_v2/probation_hang_probe.txt); the root cause was fixed, but nothing bounded a lane group's steps.  Now (a) every state a
    pair is resumed from is checked -- this pair, this launch, a step inside the pair, a slice counter inside the slice -- and (b) a step counter
    beyond the pair's last step ends the pair inside the step loop.  Either way the pair goes to the int32 kernel and is counted
    (agatha_amd_guard_stats), the results are the oracle's, and the binding warns.  The debug option poison_state writes the third suspended
    pair of the launch with a garbage step counter (mode 3: caught by (a)) or with one behind a flag that lets it pass (a) (mode 1003: caught by
    (b), after a handful of steps)."""
    qs, ts = _mig_batch(9000, 41, 200, 900)
    p = dict(BASE, w=24)
    agatha_amd.set_debug_option("force_choice", 0)
    qb, qo, ql = WL.make_batch(qs)
    tb, to, tl = WL.make_batch(ts)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=8)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload(); b.pack()
        with agatha_amd.debug_options(poison_state=mode):
            b.align(agatha_amd.Scores.make(**p)); b.download()
            with pytest.warns(RuntimeWarning, match="packed-int16 kernel"):
                eng.synchronize()
            g = b.guard_stats()
            kinds = b.pair_kinds()
        assert b.schedule_info()[0]
        got = [b.res_host[k].copy() for k in range(3)]
        assert _same(got, exp)
        assert g[2] == 1 and g[3] >= 3, g                      # one state poisoned
        assert (g[0], g[1]) == ((0, 1) if mode < 1000 else (1, 0)), g
        assert kinds[2] >= 1, kinds                            # ... and its pair redone by the int32 kernel
        # the same batch without the option: no counter moves, no warning
        b.align(agatha_amd.Scores.make(**p)); b.download(); eng.synchronize()
        assert b.guard_stats() == (0, 0, 0, 0)
        assert _same([b.res_host[k].copy() for k in range(3)], exp)
    finally:
        b.free()
        agatha_amd.set_debug_option("force_choice", -1)


def test_a_lane_group_that_never_started_is_taken_over_after_a_short_grace(eng):
    """Round 4: the only protection against a workgroup of the persistent grid that is not resident used
    to be a fixed 50 ms.  A boundary whose state is still FRESH when its left neighbour needs the pair says that the right
    neighbour has not even started -- nothing will arrive soon --, and the pair is taken over after mig_fresh_timeout_us (2 ms
    by default); the long time-out is left for pairs that are RUNNING.  Forced here: odd lane groups start 30 ms late, the long
    time-out stays at its 50 ms: the batch must end in well under that, with pairs taken over, and bit-exact."""
    import time
    qs, ts = _mig_batch(9000, 6, 200, 900)
    p = dict(BASE, w=24)
    agatha_amd.set_debug_option("force_choice", 0)
    try:
        with agatha_amd.debug_options(mig_test_delay_us=30000):
            qb, qo, ql = WL.make_batch(qs)
            tb, to, tl = WL.make_batch(ts)
            exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=8)
            b = eng.batch(qb, tb, qo, to, ql, tl)
            try:
                b.upload(); b.pack(); eng.synchronize()
                e0, e1 = eng.event(), eng.event()
                eng.record(e0); b.align(agatha_amd.Scores.make(**p)); eng.record(e1)
                ms = eng.elapsed_ms(e0, e1)
                b.download(); eng.synchronize()
                got = [b.res_host[j].copy() for j in range(3)]
                st, sched = b.step_stats(), b.schedule_info()
            finally:
                b.free()
        assert sched[0] and _same(got, exp)
        assert st[14] + st[39] > 0          # pairs were taken over (round 5: a rest whose first part was never started is taken whole at once, out[39]) ...
        assert ms < 45.0, ms                # ... without anybody sitting out the long time-out (30 ms of forced delay + the work)
    finally:
        agatha_amd.set_debug_option("force_choice", -1)


def test_one_pair_on_two_cooperating_waves(eng):
    """Latency shape for windows of 129..256 blocks (bands 1017..2040 on long pairs, e.g. BASELINE's ultra-long config): one
    pair on the two waves of a 128-thread workgroup, <128, 1> -- the H hand-off of lane 63, the step's reduced maxima and
    the queue position cross the wave boundary through LDS, one barrier per step.  Bit-exact against the oracle for several
    bands / cut diagonals, with z-drop, ragged ends, and more pairs than workgroups fit (queue)."""
    rng = np.random.default_rng(31)
    qs, ts = [], []
    for k in range(40):
        ref = WL.random_seq(rng, int(rng.integers(9000, 16000)))
        rd = WL.mutate(rng, ref, 0.03, 0.03, 0.04)
        if k % 5 == 0:
            rd = np.concatenate([rd[:rd.size // 2], WL.random_seq(rng, rd.size // 2)])          # breaks: z-drop
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    for p in (dict(BASE, w=1500), dict(BASE, w=1024, z=100), dict(m=1, x=4, q=6, r=2, s=1, z=400, w=1221)):
        qb, qo, ql = WL.make_batch(qs)
        tb, to, tl = WL.make_batch(ts)
        exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=8)
        agatha_amd.set_debug_option("force_int16", 0)
        try:
            b = eng.batch(qb, tb, qo, to, ql, tl)
            try:
                b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p)); b.download(); eng.synchronize()
                got = [b.res_host[j].copy() for j in range(3)]
                assert b.kernel_choice() == ("int16", 128, 2)
            finally:
                b.free()
        finally:
            agatha_amd.set_debug_option("force_int16", 1)
        assert _same(got, exp)


# ---- value steps (packed maxima of H alone), pairs that are started over, checkpoints ----
def _run_stats(eng, qs, ts, p, **opts):
    """align under debug options; returns (got, exp, step_stats, pair kinds)"""
    qb, qo, ql = WL.make_batch(qs)
    tb, to, tl = WL.make_batch(ts)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=8)
    with agatha_amd.debug_options(**opts):
        b = eng.batch(qb, tb, qo, to, ql, tl)
        try:
            b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p)); b.download(); eng.synchronize()
            got = [b.res_host[j].copy() for j in range(3)]
            st, kinds = b.step_stats(), b.pair_kinds()
            _run_stats.choice = b.kernel_choice()
            _run_stats.flat = b.flat_stats()
            _run_stats.sched = b.schedule_info()
        finally:
            b.free()
    return got, exp, st, kinds


def _broken_batch(seed, n, lo, hi, broken=0.3, noisy=0.2):
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for _ in range(n):
        ref = WL.random_seq(rng, int(rng.integers(lo, hi)))
        u = rng.random()
        if u < broken:          # unrelated behind a breakpoint: z-drop comes into reach long after the last rise of the maximum
            bp = int(rng.integers(ref.size // 4, ref.size))
            rd = np.concatenate([WL.mutate(rng, ref[:bp], 0.03, 0.03, 0.04), WL.random_seq(rng, ref.size - bp + 1)])
        elif u < broken + noisy:   # so noisy that the maximum keeps sinking and recovering
            e = rng.uniform(0.25, 0.4) / 3
            rd = WL.mutate(rng, ref, e, e, e)
        else:
            rd = WL.mutate(rng, ref, 0.03, 0.03, 0.04)
        if rd.size == 0:
            rd = WL.random_seq(rng, 1)
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    return qs, ts


@pytest.mark.parametrize("margin", [1, 4, 16, 64])
def test_value_steps_and_pairs_started_over(eng, margin):
    """Steps that only track the VALUES of the anti-diagonal maxima (one instruction per cell pair instead of three) decide
    nothing but the running maximum; a pair on which z-drop comes into reach, or that ends without a rise of the maximum on
    a key step, is started over on key steps.  Results bit-exact for every width of the window of key steps at a pair's end."""
    qs, ts = _broken_batch(61 + margin, 300, 400, 5000)
    for p in (BASE, dict(BASE, z=100, w=100), dict(m=1, x=4, q=6, r=2, s=1, z=200, w=500)):
        got, exp, st, kinds = _run_stats(eng, qs, ts, p, fast_margin=margin, ck_min_steps=0)
        assert _same(got, exp)
        assert st[0] > 0 and st[1] > 0    # value steps and key steps
        assert st[2] > 0                  # pairs were started over (from their first step: no checkpoints here)
        assert kinds[2] == 0              # and none of that went to the int32 kernel
    got, exp, st, _ = _run_stats(eng, qs, ts, BASE, fast_margin=0)
    assert _same(got, exp) and st[0] == 0 and st[2] == 0          # key steps only: the kernel of round 2


@pytest.mark.parametrize("shape", [0, 1])
def test_checkpoints_and_going_back_to_them(eng, shape):
    """Long pairs save their state every eighth of their steps; a pair that must be started over goes back to the last
    checkpoint before the final rise of its maximum and runs key steps from there.  Forced on short pairs here
    (ck_min_steps = 32: a checkpoint every 256 steps), on the throughput shape and on the latency shape."""
    qs, ts = _broken_batch(71, 240, 6000, 12000, broken=0.5, noisy=0.1)
    p = dict(BASE, z=120)
    agatha_amd.set_debug_option("force_int16", 0)
    agatha_amd.set_debug_option("force_choice", shape)
    try:
        got, exp, st, kinds = _run_stats(eng, qs, ts, p, ck_min_steps=32)
        assert _same(got, exp)
        assert _run_stats.choice == ("int16", 64, 2) if shape else _run_stats.choice[:2] == ("int16", 16)
        assert st[15] > 20                # pairs went back to a checkpoint
        assert kinds[2] == 0
        got, exp, st, _ = _run_stats(eng, qs, ts, p, ck_min_steps=0)
        assert _same(got, exp) and st[15] == 0 and st[2] > 20
    finally:
        agatha_amd.set_debug_option("force_choice", -1)
        agatha_amd.set_debug_option("force_int16", 1)


@pytest.mark.parametrize("lazy_max", [0, 2, 3, 8])
def test_lazy_value_steps_of_the_one_pair_per_wave_shape(eng, lazy_max):
    """Round 6: where a wave holds one pair (<64, P>) a value step whose calm test passed with room to spare answers for up to lazy_max steps
    behind it -- no lower bound, no reduction over the lanes, no test on those (step_stats [23]); what the lanes keep meanwhile (the largest
    last-column cell) enters the next test or, when the wave turns to key steps first, the bound of the running maximum.  Clean, noisy and broken
    reads, z-drop within reach of the bound's slack and far from it, checkpoints every 256 steps and going back to them: bit-exact whatever
    lazy_max; 0 = every value step is tested (the kernel as it was)."""
    qs, ts = _broken_batch(83 + lazy_max, 200, 3000, 9000, broken=0.3, noisy=0.3)
    agatha_amd.set_debug_option("force_int16", 0)
    agatha_amd.set_debug_option("force_choice", 1)
    try:
        for p in (BASE, dict(BASE, z=120), dict(m=1, x=4, q=6, r=2, s=1, z=400, w=751), dict(BASE, z=-1)):
            got, exp, st, kinds = _run_stats(eng, qs, ts, p, lazy_max=lazy_max, ck_min_steps=32)
            assert _run_stats.choice[:2] == ("int16", 64)
            assert _same(got, exp)
            assert kinds[2] == 0
            if lazy_max == 0: assert st[23] == 0
            elif p.get("z", 400) >= 400 or p.get("z") == -1: assert st[23] > st[0] // 4, (st[0], st[23])          # (a test answers for lazy_max - 1 steps behind it: a third and more of the value steps even at 2)
    finally:
        agatha_amd.set_debug_option("force_choice", -1)
        agatha_amd.set_debug_option("force_int16", 1)


def test_two_wave_shape_moves_the_base_under_a_barrier(eng):
    """One pair on two waves, more workgroups than CUs (two and more per CU, so the waves of a pair drift apart in time) and
    scores that move the base of the representation several times per pair: the E hand-off values of the lanes at the wave
    boundary are read by the other wave at the start of its next step, so moving them to the new base needs a barrier
    behind it (found in round 3 as values off by 2048 once steps became shorter)."""
    rng = np.random.default_rng(83)
    qs, ts = [], []
    for _ in range(700):
        ref = WL.random_seq(rng, int(rng.integers(5000, 7000)))
        qs.append(ref.tobytes()); ts.append(WL.mutate(rng, ref, 0.02, 0.02, 0.02).tobytes())
    p = dict(BASE, w=1500)
    agatha_amd.set_debug_option("force_int16", 0)
    agatha_amd.set_debug_option("force_choice", 1)            # the latency shape of this window: <128, 1>
    try:
        qb, qo, ql = WL.make_batch(qs)
        tb, to, tl = WL.make_batch(ts)
        exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_STEPS, threads=8)
        b = eng.batch(qb, tb, qo, to, ql, tl)
        try:
            for _ in range(3):
                b.upload(); b.pack(); b.align(agatha_amd.Scores.make(**p)); b.download(); eng.synchronize()
                got = [b.res_host[j].copy() for j in range(3)]
                assert b.kernel_choice() == ("int16", 128, 2)
                assert _same(got, exp)
                assert b.step_stats()[2] == 0 and b.step_stats()[15] == 0       # nothing had to be started over
        finally:
            b.free()
    finally:
        agatha_amd.set_debug_option("force_choice", -1)
        agatha_amd.set_debug_option("force_int16", 1)


def test_two_wave_shape_with_value_steps_n_rows_and_pairs_started_over(eng):
    """One pair on two waves (<128, 1>), steep gap scores (the in-band zone is lifted), a small z (pairs are started over all the
    time), very short and longer pairs, N runs in some queries: the two waves of a pair must always run the same kind of step --
    the round-3 fuzzer found the case where they did not (an N row in flight is a property of ONE wave's lanes), a hang."""
    rng = np.random.default_rng(29)
    qs, ts = [], []
    for k in range(60):
        ref = WL.random_seq(rng, int(rng.choice([5, 40, 300, 835, 2000, 3900])))
        rd = WL.mutate(rng, ref, 0.05, 0.04, 0.04).copy() if k % 3 else WL.random_seq(rng, int(rng.integers(1, 3000)))
        if rd.size == 0:
            rd = WL.random_seq(rng, 1)
        ref = ref.copy()
        if k % 4 == 1 and ref.size > 30:
            a = int(rng.integers(0, ref.size - 20)); ref[a:a + int(rng.integers(1, 20))] = ord("N")
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    agatha_amd.set_debug_option("force_int16", 0)
    agatha_amd.set_debug_option("force_choice", 1)
    try:
        for p in (dict(m=2, x=6, q=20, r=2, s=1, z=50, w=1500), dict(m=2, x=6, q=20, r=2, s=3, z=-1, w=1500), dict(BASE, w=1500, z=30)):
            for margin in (16, 3):
                got, exp, st, kinds = _run_stats(eng, qs, ts, p, fast_margin=margin, ck_min_steps=16)
                assert _run_stats.choice == ("int16", 128, 2)
                assert _same(got, exp), (p, margin)
    finally:
        agatha_amd.set_debug_option("force_choice", -1)
        agatha_amd.set_debug_option("force_int16", 1)


def test_a_grid_larger_than_the_checkpoint_area_takes_no_checkpoints_beyond_it(eng):
    """the checkpoint area is sized for the default persistent grid; with the max_blocks debug option above it
    -- 3 workgroups per CU -- the lane groups beyond the area must simply do without checkpoints (they used to write past the end
    of the caller's workspace: `ck_slots = 1 << 30`).  9 000 pairs of 1 050+ steps, a third of them broken so that pairs do go
    back to checkpoints; results = the oracle's, and a second batch aligned right behind, whose workspace lies next to the first
    one's, is intact."""
    qs, ts = _broken_batch(91, 9000, 4200, 5200)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    cus = 256
    got, exp, st, kinds = _run_stats(eng, qs, ts, p, max_blocks=3 * cus, force_int16=1, ck_min_steps=1024)
    assert all((g == e).all() for g, e in zip(got, exp))
    assert st[15] > 0                   # pairs did go back to checkpoints (the lane groups inside the area)
    got2, exp2, _, _ = _run_stats(eng, qs[:500], ts[:500], p, force_int16=1)
    assert all((g == e).all() for g, e in zip(got2, exp2))


@pytest.mark.parametrize("w", [1600, 2047, 2300, 3064])
def test_bands_beyond_1528_run_on_the_int16_kernel(eng, w):
    """Round 4 (VERDICT r3 missing #5): windows of 193..384 blocks -- bands 1529..3064 on long pairs -- had no int16 shape and ran
    on the int32 kernel at half the rate.  <64, 2> (one pair per wave, two register pairs per lane) and the new <64, 3> take them as
    throughput shapes; <128, 1> stays the latency shape up to 256 blocks.  Long pairs with large indels (so that the band's edges
    matter), broken ones (z-drop) and one cut diagonal per band against the oracle; nothing handed back to the int32 kernel."""
    rng = np.random.default_rng(w)
    qs, ts = [], []
    for k in range(20):
        ref = WL.random_seq(rng, int(rng.integers(14000, 22000)))
        rd = WL.mutate(rng, ref, 0.03, 0.03, 0.04)
        if k % 4 == 1:          # a large deletion / insertion: the path runs along the band's edge
            cut = int(rng.integers(1000, min(w, 2500))); at = int(rng.integers(2000, 8000))
            rd = np.concatenate([rd[:at], rd[at + cut:]]) if k % 8 == 1 else np.concatenate([rd[:at], WL.random_seq(rng, cut), rd[at:]])
        if k % 5 == 0:
            rd = np.concatenate([rd[:rd.size // 2], WL.random_seq(rng, rd.size // 2)])          # breaks: z-drop
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    p = dict(BASE, w=w)
    got, exp, st, kinds = _run_stats(eng, qs, ts, p, force_int16=1)
    W = (w + 7) // 8
    assert _run_stats.choice[0] == "int16" and _run_stats.choice[1] in (64, 128) and _run_stats.choice[1] * _run_stats.choice[2] >= W + 1
    assert all((g == e).all() for g, e in zip(got, exp))
    assert kinds[2] == 0 and st[0] + st[1] > 0


@pytest.mark.parametrize("w", [1500, 2000, 3000])
def test_z_drop_off_and_sequences_cut_short_on_the_wide_shapes(eng, w):
    """the early end "an anti-diagonal below the in-band zone ends the pair" must only fire where no in-band cell of the pair
    is left (round 5: the geometry is checked), also with z-drop off, on very long divergent pairs and on the shapes of round 4 --
    <64, 2>, <64, 3> as throughput shapes, <128, 1> as the latency shape.  Targets / queries cut to 45 .. 85 % (shorter than the other
    sequence by more than the band), z = -1 and z = 400, against the oracle; whatever leaves for the int32 kernel comes back right."""
    rng = np.random.default_rng(w + 7)
    qs, ts = [], []
    for k in range(18):
        ref = WL.random_seq(rng, int(rng.integers(12000, 20000)))
        rd = WL.mutate(rng, ref, 0.04, 0.04, 0.05)
        f = float(rng.uniform(0.45, 0.85))
        if k % 3 == 0:
            rd = rd[:max(1, int(rd.size * f))]              # target much shorter than the query
        elif k % 3 == 1:
            ref = ref[:max(1, int(ref.size * f))]           # query much shorter than the target
        else:
            rd = np.concatenate([rd[:int(rd.size * f)], WL.random_seq(rng, rd.size - int(rd.size * f))])      # divergent tail: scores sink
        qs.append(ref.tobytes()); ts.append(rd.tobytes())
    for z in (-1, 400):
        p = dict(BASE, w=w, z=z)
        got, exp, st, kinds = _run_stats(eng, qs, ts, p, force_int16=1)
        assert _run_stats.choice[0] == "int16" and _run_stats.choice[1] in (32, 64, 128)
        assert all((g == e).all() for g, e in zip(got, exp)), (w, z)


def test_pairs_that_give_up_late_on_the_static_schedule_go_to_the_clean_up_launch(eng):
    """Round 5: on a static schedule a pair that must start from its FIRST step far into its steps costs its lane group a whole pair's key
    steps that nobody else can take, and the kernel -- which ends with its last wave -- 25 ms for ONE such pair.  It now leaves for a
    clean-up launch of the int16 latency shape behind the kernel (kind 5: one pair per wave, key steps only).  Forced here: 9 000 pairs
    of 1 050+ steps, a third of them broken, and no checkpoints at all (ck_min_steps beyond every pair), so that every pair that
    gives up has nowhere to go back to.  Results = the oracle's with the clean-up launch and without it (cleanup_min_steps = 0: in
    place, as until round 5); with it the pairs are counted, and they are still 'plain' pairs for agatha_amd_pair_kinds."""
    qs, ts = _broken_batch(93, 9000, 4200, 5200)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    # (force_int16 = 0 -- this module's engine sets it --: the clean-up launch is the int16 LATENCY shape, which is only a candidate when the
    #  device has the choice)
    got, exp, st, kinds = _run_stats(eng, qs, ts, p, force_int16=0, ck_min_steps=1 << 20, flat_detect=0)
    assert _run_stats.choice[0] == "int16" and _run_stats.sched[0]      # the int16 throughput shape on the static schedule
    assert all((g == e).all() for g, e in zip(got, exp))
    sent = _run_stats.flat[4]
    assert sent > 50, (sent, st[:4])                                    # broken pairs far into their steps: to the clean-up launch
    assert kinds[0] == len(qs) - kinds[2] and kinds[2] < 20             # they stay plain pairs
    got2, exp2, st2, _ = _run_stats(eng, qs, ts, p, force_int16=0, ck_min_steps=1 << 20, flat_detect=0, cleanup_min_steps=0)
    assert all((g == e).all() for g, e in zip(got2, exp2))
    assert _run_stats.flat[4] == 0 and st2[2] >= sent                   # in place: nothing sent, at least as many started over
