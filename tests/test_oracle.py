"""CPU-only: the oracle against the reference's known answers, and its two schedules against each other."""
import os

import numpy as np
import pytest

from oracle import oracle as O, synth
from helpers import load_kats, load_ref_vectors, oracle_batch


@pytest.mark.parametrize("model", [O.MODEL_SLICES, O.MODEL_STEPS])
@pytest.mark.parametrize("wide", [False, True])
def test_appendix_e_kats(model, wide):
    doc = load_kats()
    for c in doc["cases"]:
        s, q, t = O.align_pairs([c["query"]], [c["target"]], O.make_params(**c["params"]), wide=wide, model=model)
        assert [int(s[0]), int(q[0]), int(t[0])] == c["expect"], (c["id"], c["params"], c["note"])


def test_int16_saturation_kat():
    sat = load_kats()["saturation"]
    rng = np.random.default_rng(sat["seed"])
    seq = "".join(rng.choice(list("ACGT"), sat["length"]))
    for model in (O.MODEL_SLICES, O.MODEL_STEPS):
        s, q, t = O.align_pairs([seq], [seq], O.make_params(), wide=False, model=model)
        assert [int(s[0]), int(q[0]), int(t[0])] == sat["expect_faithful"]
        s, q, t = O.align_pairs([seq], [seq], O.make_params(), wide=True, model=model)
        assert [int(s[0]), int(q[0]), int(t[0])] == sat["expect_wide"]


@pytest.mark.parametrize("g", load_ref_vectors(), ids=lambda g: g["name"])
def test_reference_vectors(g):
    """Outputs of the reference kernel (run under oracle/ref_shim when the fixture was generated)."""
    for wide, model in ((False, O.MODEL_SLICES), (True, O.MODEL_STEPS)):
        if model == O.MODEL_STEPS and g["qlen"].max() > 12000:
            pass
        s, q, t = oracle_batch(g, wide=wide, model=model, threads=4)
        exp = g["expect"]
        assert (s == exp[0]).all() and (q == exp[1]).all() and (t == exp[2]).all()


def test_slices_equals_steps_random():
    rng = np.random.default_rng(11)
    for _ in range(60):
        w = int(rng.choice([0, 1, 3, 5, 8, 9, 16, 17, 33, 64, 100, 751]))
        P = O.make_params(m=2, x=4, q=4, r=int(rng.choice([1, 2])), s=int(rng.choice([1, 2, 3, 5, 7])),
                          z=int(rng.choice([-1, 0, 20, 100, 400])), w=w)
        e = float(rng.uniform(0, 0.15))
        qs, ts = synth.make_pairs(int(rng.integers(1 << 30)), 16, lambda r: int(np.exp(r.uniform(0, np.log(1200)))),
                                  e, e, e, n_rate=0.01)
        if rng.random() < 0.4:
            ts = [t[: max(1, len(t) // 3)] for t in ts]
        for wide in (False, True):
            a = O.align_pairs(qs, ts, P, wide=wide, model=O.MODEL_SLICES)
            b = O.align_pairs(qs, ts, P, wide=wide, model=O.MODEL_STEPS)
            assert all((x == y).all() for x, y in zip(a, b))


def test_exact_band_model_agrees_at_wide_bands():
    """At the BASELINE bands the textbook |i-j|<=w model agrees with the block-granular reference semantics."""
    qs, ts = synth.make_pairs(5, 24, lambda r: int(r.integers(500, 3000)), 0.03, 0.03, 0.04)
    for w in (101, 751):
        P = O.make_params(w=w)
        a = O.align_pairs(qs, ts, P, wide=True, model=O.MODEL_SLICES)
        b = O.align_pairs(qs, ts, P, model=O.MODEL_EXACTBAND)
        assert all((x == y).all() for x, y in zip(a, b))


def test_pack_layout():
    buf, off, ln = O.make_batch([b"ACGTNACG", b"TT"])
    p = O.pack(buf)
    # first base in bits 31-28; A=1 C=3 G=7 T=4 N=14 (reference pack_rc_seqs.h:21-33)
    assert p[0] == 0x1374E137 and p[1] == 0x44EEEEEE


def test_nominal_cells_formula():
    L, w = 10000, 751
    assert O.nominal_cells(L, L, w) == L * (2 * w + 1) - w * (w + 1) == 14465248
    assert O.nominal_cells_np([L], [L], w) == 14465248


def test_ksw_style_avx2_equals_exact_band_model():
    """The SIMD CPU baseline (oracle/ksw_style_avx2.c) is bit-identical to the scalar exact-band model."""
    rng = np.random.default_rng(8)
    for _ in range(80):
        w = int(rng.choice([0, 1, 5, 8, 9, 16, 33, 64, 100, 751]))
        P = O.make_params(m=int(rng.choice([1, 2, 3])), x=4, q=int(rng.choice([2, 4, 6])), r=int(rng.choice([1, 2])),
                          z=int(rng.choice([-1, 0, 40, 400])), w=w)
        e = float(rng.uniform(0, 0.15))
        qs, ts = synth.make_pairs(int(rng.integers(1 << 30)), 12, lambda r: int(np.exp(r.uniform(0, np.log(2000)))),
                                  e, e, e, n_rate=0.01)
        if rng.random() < 0.4:
            ts = [t[: max(1, len(t) // 3)] for t in ts]
        qb, qo, ql = O.make_batch(qs)
        tb, to, tl = O.make_batch(ts)
        a = O.align_batch(qb, tb, qo, to, ql, tl, P, model=O.MODEL_EXACTBAND)
        b = O.ksw_style_batch(qb, tb, qo, to, ql, tl, P, threads=2)
        assert all((x == y).all() for x, y in zip(a, b[:3]))


def test_ksw_style_matches_reference_vectors_at_baseline_band():
    for g in load_ref_vectors():
        if g["params"]["w"] < 500:
            continue
        r = O.ksw_style_batch(g["qbatch"], g["tbatch"], g["qoff"], g["toff"], g["qlen"], g["tlen"],
                              O.make_params(**g["params"]), threads=4)
        assert all((x == y).all() for x, y in zip(r[:3], g["expect"])), g["name"]


def _mixed_pairs(rng, n, lmax):
    from agatha_amd.workload import random_seq, mutate
    qs, ts = [], []
    for _ in range(n):
        L = int(rng.integers(1, lmax))
        ref = random_seq(rng, L)
        mode = int(rng.integers(0, 4))
        if mode == 0:
            rd = mutate(rng, ref, 0.03, 0.03, 0.04)
        elif mode == 1:
            rd = mutate(rng, ref, 0.15, 0.1, 0.1)
        elif mode == 2:
            rd = random_seq(rng, int(rng.integers(1, lmax)))
        else:
            bp = int(rng.integers(0, L))
            rd = np.concatenate([mutate(rng, ref[:bp], 0.02, 0.02, 0.02), random_seq(rng, int(rng.integers(1, 600)))])
        if rd.size == 0:
            rd = random_seq(rng, 1)
        if rng.random() < 0.2:
            ref = ref.copy()
            ref[rng.random(ref.size) < 0.05] = ord("N")
        qs.append(ref.tobytes())
        ts.append(rd.tobytes())
    return qs, ts


def test_lane_schedule_model_equals_oracle():
    """The CPU emulation of the int32 kernel's lane/slot schedule is bit-identical to the slice-wise oracle."""
    rng = np.random.default_rng(21)
    for w, G, S in ((24, 16, 1), (100, 16, 1), (751, 32, 3)):
        prm = O.make_params(2, 4, 4, 2, 3, 400, w)
        qs, ts = _mixed_pairs(rng, 12, 1500)
        qb, qo, ql = O.make_batch(qs)
        tb, to, tl = O.make_batch(ts)
        e = O.align_batch(qb, tb, qo, to, ql, tl, prm, wide=True, model=O.MODEL_SLICES, threads=4)
        g = O.lanes_batch(qb, tb, qo, to, ql, tl, prm, G, S, threads=4)
        for a, b in zip(e, g):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_int16_kernel_model_equals_oracle(seed):
    """The packed-int16 kernel's arithmetic (oracle/agatha_lanes_model.c: rebased int16 representation, band cut by
    constants instead of per-cell tests, bail-out to int32) gives the oracle's results, keeps every value inside
    int16 and keeps its three value zones disjoint."""
    rng = np.random.default_rng(seed)
    fell_back = 0
    for trial in range(10):
        w = int(rng.choice([16, 17, 23, 24, 33, 47, 64, 100, 200, 751]))
        z = int(rng.choice([-1, 0, 20, 100, 400, 2000]))
        s = int(rng.choice([1, 2, 3, 5, 8]))
        m, x, q, r = [(2, 4, 4, 2), (1, 4, 6, 2), (2, 3, 5, 1), (3, 5, 0, 1), (1, 1, 1, 1), (16, 32, 64, 16)][int(rng.integers(0, 6))]
        prm = O.make_params(m, x, q, r, s, z, w)
        qs, ts = _mixed_pairs(rng, 16, 2500)
        qb, qo, ql = O.make_batch(qs)
        tb, to, tl = O.make_batch(ts)
        W = (w + 7) // 8
        G, S = [c for c in ((16, 2), (16, 4), (16, 6), (32, 4), (32, 6), (64, 4)) if c[0] * c[1] >= W + 1][0]
        e = O.align_batch(qb, tb, qo, to, ql, tl, prm, wide=True, model=O.MODEL_STEPS, threads=4)
        sc, qe, te, kind, st = O.lanes16_batch(qb, tb, qo, to, ql, tl, prm, G, S, threads=4)
        assert np.array_equal(e[0], sc) and np.array_equal(e[1], qe) and np.array_equal(e[2], te)
        assert (kind >= 0).all()
        fell_back += int((kind == 1).sum())
        if (kind == 0).any():
            assert st[0] >= -32768 and st[1] <= 32767          # no int16 wrap anywhere
            assert st[2] < -22000 and st[3] >= -22000          # out-of-band cells below, in-band cells above L16_GLO
    assert fell_back < 10 * 16 // 2


@pytest.mark.parametrize("margin", [1, 4, 16])
def test_value_steps_decide_nothing_but_the_running_maximum(margin):
    """The int16 kernel's value steps (align16_body.inc, FAST) in the lane model: all but a pair's first step and its last
    `margin` (+ 1/128 of its) steps compute no maxima inside the blocks; they bound the running maximum from above and every
    anti-diagonal maximum from below by the cells of each block's last row and column (round 4; oracle/agatha_lanes_model.c has the
    argument), and rely on the calm test (the lower bounds within z of the upper bound, inside their zone, inside the pair) to
    decide nothing but the running maximum; pairs that meet a step they cannot decide, or end without the cell of their
    maximum, are started over on key steps (kind 2).  Whatever the margin, the results are the oracle's; clean pairs are never
    started over with a window of 16 key steps (+ what the pair's own rate of rise asks for: the bound is up to 7 mismatches +
    7 gap extensions above the maximum and the key steps must see the score rise by more than that); broken pairs (z-drop) are."""
    rng = np.random.default_rng(100 + margin)
    started_over = clean_started_over = 0
    for trial in range(6):
        w = int(rng.choice([24, 47, 100, 200, 751]))
        z = int(rng.choice([-1, 20, 100, 400]))
        s = int(rng.choice([1, 3, 5]))
        m, x, q, r = [(2, 4, 4, 2), (1, 4, 6, 2), (2, 3, 5, 1)][int(rng.integers(0, 3))]
        prm = O.make_params(m, x, q, r, s, z, w)
        qs, ts = _mixed_pairs(rng, 12, 2500)
        clean = synth.make_pairs(int(rng.integers(1, 1000)), 6, lambda g: int(g.integers(600, 2500)), 0.03, 0.03, 0.04)
        qs, ts = qs + clean[0], ts + clean[1]
        qb, qo, ql = O.make_batch(qs)
        tb, to, tl = O.make_batch(ts)
        W = (w + 7) // 8
        G, S = [c for c in ((16, 2), (16, 4), (16, 6), (32, 4), (32, 6), (64, 4)) if c[0] * c[1] >= W + 1][0]
        e = O.align_batch(qb, tb, qo, to, ql, tl, prm, wide=True, model=O.MODEL_STEPS, threads=4)
        sc, qe, te, kind, st = O.lanes16_batch(qb, tb, qo, to, ql, tl, prm, G, S, threads=4, value_step_margin=margin)
        assert np.array_equal(e[0], sc) and np.array_equal(e[1], qe) and np.array_equal(e[2], te)
        started_over += int((kind == 2).sum())
        if z < 0 or z >= 400:                  # (with a small z even a clean pair dips far enough below its maximum)
            clean_started_over += int((kind[-6:] == 2).sum())
    assert started_over > 0
    if margin >= 16:
        assert clean_started_over == 0


def _seq_ops_fixture():
    import json
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seq_ops_as_written.json")))
    seqs = [s.encode() for s in d["seqs"]]
    return seqs, np.asarray(d["ops"], np.uint8), np.asarray(d["packed_after"], np.uint32)


def _py_transform(seq, op, pad_first):
    """Reverse / complement of the string.  pad_first: reverse the PADDED sequence, as the reference's kernel does as
    written (the padding Ns end up in front)."""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    s = seq + b"N" * ((-len(seq)) % 8) if pad_first else seq
    s = s[::-1] if op & 1 else s
    return s.translate(comp) if op & 2 else s


def test_int16_model_ends_a_pair_whose_real_cells_have_run_out():
    """A target (or query) shorter than the other sequence by more than the band: the band leaves the matrix through the last column
    (row) and the blocks behind it hold only padded columns, cells that derive from the reference's -infinity (agatha_kernel.h:207-215).
    The int16 kernel ends such a pair on the first anti-diagonal whose maximum lies below its in-band zone -- the result is final --
    instead of abandoning it to the int32 kernel (align16_body.inc, round 4); its arithmetic model does the same here: no pair bails
    out, every result is the oracle's, at bands whose cut diagonal differs, with and without z-drop."""
    rng = np.random.default_rng(3)
    for w in (20, 47, 133):
        for cut, side in ((0.5, 0), (0.9, 0), (0.8, 1)):
            qs, ts = synth.make_pairs(int(rng.integers(1, 10 ** 6)), 12, lambda r: int(r.integers(600, 2000)), 0.03, 0.03, 0.04)
            if side == 0:
                ts = [t[:max(8, int(len(t) * cut))] for t in ts]
            else:
                qs = [q[:max(8, int(len(q) * cut))] for q in qs]
            qb, qo, ql = O.make_batch(qs)
            tb, to, tl = O.make_batch(ts)
            W = (w + 7) // 8
            G, S = [c for c in ((16, 2), (16, 4), (16, 6)) if c[0] * c[1] >= W + 1][0]
            for z in (400, -1):
                prm = O.make_params(2, 4, 4, 2, 3, z, w)
                e = O.align_batch(qb, tb, qo, to, ql, tl, prm, wide=True, model=O.MODEL_STEPS, threads=4)
                sc, qe, te, kind, st = O.lanes16_batch(qb, tb, qo, to, ql, tl, prm, G, S, threads=4)
                assert (kind == 0).all(), (w, cut, side, z, kind)
                assert np.array_equal(e[0], sc) and np.array_equal(e[1], qe) and np.array_equal(e[2], te), (w, cut, side, z)


def test_seq_ops_reference_as_written_and_product_semantics():
    """f2 (reverse / complement op path): oracle/seq_ops_ref.c restates the reference's gasal_reversecomplement_kernel
    (pack_rc_seqs.h:56-212) as written and the semantics the product implements.  Pinned here: (1) the restatement
    reproduces the committed fixture; (2) as written, the reference reverses the PADDED sequence (its count of padding
    bases compares a nibble with 0x4E and is always 0, :113-116), so padding Ns rotate to the front; (3) the product
    semantics reverse exactly len bases; (4) the two agree exactly when len % 8 == 0 or the op does not reverse."""
    seqs, ops, after = _seq_ops_fixture()
    buf, offs, lens = O.make_batch(seqs)
    packed = O.pack(buf)
    ref = O.seq_ops(packed, lens, offs, ops, as_written=True)
    assert (ref == after).all()                                                                    # (1)
    padded = [_py_transform(s, int(o), True) for s, o in zip(seqs, ops)]
    assert (ref == O.pack(O.make_batch(padded)[0])).all()                                          # (2)
    prod = O.seq_ops(packed, lens, offs, ops, as_written=False)
    assert (prod == O.pack(O.make_batch([_py_transform(s, int(o), False) for s, o in zip(seqs, ops)])[0])).all()   # (3)
    n_diff = 0
    for s, o, off in zip(seqs, ops, offs):                                                         # (4)
        w0, w1 = int(off) // 8, (int(off) + len(s) + 7) // 8
        same = (ref[w0:w1] == prod[w0:w1]).all()
        assert same == (len(s) % 8 == 0 or not (o & 1)), (len(s), int(o))
        n_diff += not same
    assert n_diff > 20


def test_start_positions_definition():
    """f4: the start of an alignment = where the same extension, run backwards from the end cell on the reversed prefixes,
    ends (GASAL2's WITH_START idea, gasal.h:36; the reference declares the result members and leaves them NULL).  A clean
    pair starts at (0, 0); junk in front of either sequence is skipped by the start, although the extension itself,
    anchored at (0, 0), pays a gap for it."""
    rng = np.random.default_rng(3)
    core = synth.random_seq(rng, 400).tobytes()
    junk = synth.random_seq(rng, 30).tobytes()
    P = O.make_params(w=100, z=400)
    qs, ts = [core, junk + core, core, b"ACGT"], [core, core, junk + core, b"TTTT"]
    s, qe, te = O.align_pairs(qs, ts, P, wide=True)
    qs_, ts_, back = O.start_positions(qs, ts, P, qe, te)
    assert s.tolist()[0] == 800 and qs_.tolist() == [0, 30, 0, 0] and ts_.tolist() == [0, 0, 30, 0]
    assert (back >= s).all() and back.tolist()[1] == 800


def _full_dp_best_path_score(q, t, P, qe, te):
    """Unbanded Gotoh DP in plain Python (small inputs only): the best score of an alignment of q[0..qe] with t[0..te] that
    starts at the origin and ends in the cell (qe, te) -- an upper bound for any path the banded walk can report, reached
    when the band is wider than the sequences."""
    NEG = -10 ** 9
    a, b, go, ge = P.match, P.mismatch, P.gap_open, P.gap_extend
    n, m = qe + 1, te + 1
    H = [[NEG] * (m + 1) for _ in range(n + 1)]
    E = [[NEG] * (m + 1) for _ in range(n + 1)]
    F = [[NEG] * (m + 1) for _ in range(n + 1)]
    H[0][0] = 0
    for j in range(1, m + 1):
        E[0][j] = H[0][j] = -(go + ge * j)
    for i in range(1, n + 1):
        F[i][0] = H[i][0] = -(go + ge * i)
        for j in range(1, m + 1):
            x, y = q[i - 1] & 15, t[j - 1] & 15
            s = -1 if (x == 14 or y == 14) else (a if x == y else -b)
            E[i][j] = max(E[i][j - 1] - ge, H[i][j - 1] - go - ge)
            F[i][j] = max(F[i - 1][j] - ge, H[i - 1][j] - go - ge)
            H[i][j] = max(H[i - 1][j - 1] + s, E[i][j], F[i][j])
    return H[n][m]


def test_traceback_paths_rescore_to_the_reported_score():
    """agatha_model_traceback (cigar / n_cigar_ops are declared and never filled by the reference,
    gasal.h:91-92, so there is no reference vector for them): scores and end cells equal the scoring model's, every path
    re-scored from the sequences alone gives the reported score and uses query_end + 1 / target_end + 1 bases, long runs
    are split at 63, and with a band wider than the sequences the path is optimal among all alignments ending in that cell."""
    rng = np.random.default_rng(11)
    total = nopath = 0
    for w, z, n, L in ((751, 400, 30, 2500), (16, 400, 150, 300), (3, -1, 300, 60), (100, 50, 150, 800), (0, 400, 50, 40)):
        P = O.make_params(w=w, z=z)
        qs, ts = [], []
        for k in range(n):
            ln = int(rng.integers(1, L))
            q = synth.random_seq(rng, ln)
            t = synth.mutate(rng, q, 0.05, 0.04, 0.04)
            if k % 7 == 0:
                q = q.copy(); q[rng.integers(0, ln)] = ord('N')
            qs.append(bytes(q)); ts.append(bytes(t))
        s, qe, te, cig = O.traceback_pairs(qs, ts, P, threads=4)
        s2, qe2, te2 = O.align_pairs(qs, ts, P, wide=True, model=O.MODEL_SLICES, threads=4)
        assert (s == s2).all() and (qe == qe2).all() and (te == te2).all()
        for k in range(n):
            total += 1
            if cig[k] is None:
                nopath += 1
                continue
            if s[k] <= 0:
                assert cig[k] == b""
                continue
            assert all(1 <= (b >> 2) <= 63 for b in cig[k])
            assert O.cigar_rescore(cig[k], qs[k], ts[k], P) == (s[k], qe[k] + 1, te[k] + 1)
    assert nopath <= total // 100          # only scores that came through a skipped band-edge cell have no path
    # a 200-base identity: 63 + 63 + 63 + 11 matches
    s, qe, te, cig = O.traceback_pairs([b"ACGTTGCA" * 25], [b"ACGTTGCA" * 25], O.make_params())
    assert cig[0] == bytes([63 << 2, 63 << 2, 63 << 2, 11 << 2]) and s[0] == 400
    # optimality inside a band that holds the whole matrix
    P = O.make_params(w=400, z=-1)
    qs, ts = [], []
    for k in range(40):
        q = synth.random_seq(rng, int(rng.integers(5, 70)))
        qs.append(bytes(q)); ts.append(bytes(synth.mutate(rng, q, 0.1, 0.08, 0.08)))
    s, qe, te, cig = O.traceback_pairs(qs, ts, P, threads=4)
    for k in range(40):
        if s[k] > 0:
            assert _full_dp_best_path_score(qs[k], ts[k], P, int(qe[k]), int(te[k])) == s[k]


def test_traceback_golden_fixture():
    """tests/golden/traceback_paths.json (generated by tests/golden/gen_traceback_golden.py) pins the definition of the paths."""
    import json
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "traceback_paths.json")))
    n = 0
    for case in doc["cases"]:
        qs = [q.encode() for q in case["queries"]]
        ts = [t.encode() for t in case["targets"]]
        s, qe, te, cig = O.traceback_pairs(qs, ts, O.make_params(**case["params"]))
        assert [int(v) for v in s] == case["score"] and [int(v) for v in qe] == case["query_end"]
        assert [int(v) for v in te] == case["target_end"]
        assert [None if c is None else c.hex() for c in cig] == case["bytes"]
        n += len(qs)
    assert n == 48
