"""GPU parity: the HIP path (through the C-ABI of libagatha_amd.so) against the oracle and the golden vectors.
Everything here needs a real MI355X: run with `pytest -m gpu`."""
import os

import numpy as np
import pytest

from oracle import oracle as O, synth
from helpers import load_kats, load_ref_vectors, GOLDEN as GOLDEN_DIR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import agatha_amd
    e = agatha_amd.Engine(0)        # raises if the HIP library or the GPU is missing: no fallback
    yield e
    e.close()


def _scores(p):
    import agatha_amd
    return agatha_amd.Scores.make(**p)


def _gpu(eng, qs, ts, p, **kw):
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    return eng.align_host_batch(qb, tb, qo, to, ql, tl, _scores(p), **kw)


def test_pack_matches_reference_layout(eng):
    rng = np.random.default_rng(0)
    seqs = [bytes(rng.choice(list(b"ACGTNacgtn"), size=int(n))) for n in (1, 7, 8, 9, 63, 1000, 4097)]
    qb, qo, ql = O.make_batch(seqs)
    tb, to, tl = O.make_batch(seqs[::-1])
    b = eng.batch(qb, tb, qo, to, ql, tl)
    b.upload(); b.pack()
    pq, pt = b.packed_host()
    b.free()
    assert (pq == O.pack(qb)).all() and (pt == O.pack(tb)).all()


def test_appendix_e_kats(eng):
    doc = load_kats()
    for c in doc["cases"]:
        s, q, t = _gpu(eng, [c["query"]], [c["target"]], c["params"])
        assert [int(s[0]), int(q[0]), int(t[0])] == c["expect"], (c["id"], c["params"], c["note"])


def test_wide_semantics_beyond_int16(eng):
    """Outside the reference's 16-bit domain the engine keeps int32 arithmetic: Appendix E #14."""
    sat = load_kats()["saturation"]
    rng = np.random.default_rng(sat["seed"])
    seq = "".join(rng.choice(list("ACGT"), sat["length"]))
    s, q, t = _gpu(eng, [seq], [seq], dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751))
    assert [int(s[0]), int(q[0]), int(t[0])] == sat["expect_wide"]


@pytest.mark.parametrize("g", load_ref_vectors(), ids=lambda g: g["name"])
def test_reference_vectors(eng, g):
    """Bit-exact against outputs of the reference kernel (tests/golden/gen_golden.py)."""
    s, q, t = eng.align_host_batch(g["qbatch"], g["tbatch"], g["qoff"], g["toff"], g["qlen"], g["tlen"],
                                   _scores(g["params"]))
    exp = g["expect"]
    bad = np.nonzero((s != exp[0]) | (q != exp[1]) | (t != exp[2]))[0]
    assert bad.size == 0, (g["name"], bad[:8], s[bad[:8]], exp[0][bad[:8]])


@pytest.mark.parametrize("seed", range(6))
def test_random_batches_vs_oracle(eng, seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(6):
        w = int(rng.choice([0, 1, 5, 8, 9, 16, 17, 33, 64, 100, 248, 500, 751, 1500]))
        p = dict(m=int(rng.choice([1, 2, 3])), x=int(rng.choice([2, 4, 5])), q=int(rng.choice([2, 4, 6])),
                 r=int(rng.choice([1, 2])), s=int(rng.choice([1, 2, 3, 5, 7])),
                 z=int(rng.choice([-1, 0, 20, 100, 400])), w=w)
        e = float(rng.uniform(0, 0.15))
        n = int(rng.choice([1, 3, 17, 64, 200]))
        maxlen = int(rng.choice([60, 500, 4000]))
        qs, ts = synth.make_pairs(int(rng.integers(1 << 30)), n, lambda r: int(np.exp(r.uniform(0, np.log(maxlen)))),
                                  e, e, e, n_rate=0.01 if rng.random() < 0.3 else 0.0)
        if rng.random() < 0.3:
            ts = [t[: max(1, len(t) // int(rng.integers(2, 5)))] for t in ts]
        qb, qo, ql = O.make_batch(qs)
        tb, to, tl = O.make_batch(ts)
        exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=4)
        for hint in (True, False):          # with / without the length hint: different lane-group shapes
            got = eng.align_host_batch(qb, tb, qo, to, ql, tl, _scores(p), use_len_hint=hint)
            for a, b_ in zip(got, exp):
                assert (a == b_).all(), (p, hint, eng.last_config())


def test_rejects_bad_arguments(eng):
    import agatha_amd
    qb, qo, ql = O.make_batch([b"ACGT"])
    # a huge band is fine while the sequences are short (window = min(W+1, ceil(Q/8), ceil(R/8)))...
    s, q, t = eng.align_host_batch(qb, qb, qo, qo, ql, ql, _scores(dict(w=10 ** 6)))
    assert (int(s[0]), int(q[0]), int(t[0])) == (8, 3, 3)
    # ...and refused, loudly, once nothing bounds the window below what is compiled
    with pytest.raises(agatha_amd.AgathaError):
        eng.align_host_batch(qb, qb, qo, qo, ql, ql, _scores(dict(w=10 ** 6)), use_len_hint=False)
    with pytest.raises(agatha_amd.AgathaError):
        eng.align_host_batch(qb[:4], qb, qo, qo, ql, ql, _scores({}))                 # bytes not a multiple of 8


def test_score_range_guard(eng):
    """Scores travel as H << K in 32-bit keys (K = 10 at band 751): a pair that could exceed 2^(30-K) is refused -- the whole
    call with AGATHA_AMD_ERANGE when the length hints prove it, that pair alone (AGATHA_AMD_BAD_RESULT, -1, -1) when the caller
    gave no hints; the other pairs of the batch are aligned as usual.  Also on the negative side when z-drop is off."""
    import agatha_amd
    rng = np.random.default_rng(1)
    long_ = synth.random_seq(rng, 12000).tobytes()
    short = synth.random_seq(rng, 900).tobytes()
    qs, ts = [long_, short], [long_, short]
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    for p in (dict(m=100, x=4, q=4, r=2, s=3, z=400, w=751),          # 12 000 x 100 > 2^20
              dict(m=2, x=100, q=4, r=2, s=3, z=-1, w=751)):          # z-drop off: -100 per mismatch can run away as well
        with pytest.raises(agatha_amd.AgathaError, match="range"):
            eng.align_host_batch(qb, tb, qo, to, ql, tl, _scores(p))
        s, q, t = eng.align_host_batch(qb, tb, qo, to, ql, tl, _scores(p), use_len_hint=False)
        assert (int(s[0]), int(q[0]), int(t[0])) == (-2 ** 31, -1, -1)
        exp = O.align_pairs([short], [short], O.make_params(**p), wide=True)
        assert (int(s[1]), int(q[1]), int(t[1])) == tuple(int(v[0]) for v in exp)


def _transform(seq, op):
    comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    s = seq[::-1] if op & 1 else seq
    return s.translate(comp) if op & 2 else s


def test_reverse_complement_ops(eng):
    """Header-char op codes (0 forward, 1 reverse, 2 complement, 3 reverse-complement): aligning with ops must equal
    aligning the transformed strings.  Lengths include non-multiples of 8 and sequences with Ns."""
    rng = np.random.default_rng(21)
    qs, ts = synth.make_pairs(17, 64, lambda r: int(r.integers(1, 700)), 0.03, 0.03, 0.03, n_rate=0.01)
    qops = rng.integers(0, 4, len(qs)).astype(np.uint8)
    tops = rng.integers(0, 4, len(ts)).astype(np.uint8)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=100, w=64)
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    got = eng.align_host_batch(qb, tb, qo, to, ql, tl, _scores(p), qops=qops, tops=tops)
    exp = O.align_pairs([_transform(q, o) for q, o in zip(qs, qops)], [_transform(t, o) for t, o in zip(ts, tops)],
                        O.make_params(**p), wide=True)
    for a, b in zip(got, exp):
        assert (a == b).all()
    # the packed words themselves
    b_ = eng.batch(qb, tb, qo, to, ql, tl)
    b_.upload(); b_.pack(); b_.seq_ops(qops, tops)
    pq, pt = b_.packed_host()
    b_.free()
    eq, _, _ = O.make_batch([_transform(q, o) for q, o in zip(qs, qops)])
    assert (pq == O.pack(eq)).all()


def test_edge_shapes(eng):
    """Empty-ish and ragged inputs: single pair, length-1 sequences, one side much longer, many tiny pairs."""
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    cases = [([b"A"], [b"A"]), ([b"A"], [b"ACGTACGTACGT" * 50]), ([b"ACGT" * 2000], [b"AC"]),
             ([b"N" * 9], [b"N" * 17]), ([b"ACGTACGTA"], [b"ACGTACGTA"])]
    for qs, ts in cases:
        got = _gpu(eng, qs, ts, p)
        exp = O.align_pairs(qs, ts, O.make_params(**p), wide=True)
        assert all((a == b).all() for a, b in zip(got, exp)), (qs[0][:12], ts[0][:12])
    rng = np.random.default_rng(3)
    qs = [synth.random_seq(rng, int(n)).tobytes() for n in rng.integers(1, 40, 20000)]
    ts = [synth.random_seq(rng, int(n)).tobytes() for n in rng.integers(1, 40, 20000)]
    got = _gpu(eng, qs, ts, p)
    exp = O.align_pairs(qs, ts, O.make_params(**p), wide=True, threads=8)
    assert all((a == b).all() for a, b in zip(got, exp))


def test_wrong_length_hint_is_refused_per_pair(eng):
    """A batch whose hint claims short reads but holds a long pair: that pair comes back as BAD_RESULT, not garbage."""
    import agatha_amd
    qs, ts = synth.make_pairs(9, 4, lambda r: 6000)
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    b.max_qlen, b.max_tlen = 64, 64                  # lie: window of 8 blocks -> 16-lane groups, band needs 95
    b.upload(); b.pack(); b.align(agatha_amd.Scores.make()); b.download(); eng.synchronize()
    assert (b.res_host[0] == -2 ** 31).all() and (b.res_host[1] == -1).all()
    b.free()


def test_letters_outside_acgtn(eng):
    """Any other letter only matches a letter with the same low nibble (reference: `ASCII & 0xF`); pairs holding such
    letters take the kernel's compare path, the others the score-profile path -- mixed in one batch here."""
    rng = np.random.default_rng(77)
    qs, ts = synth.make_pairs(31, 48, lambda r: int(r.integers(50, 1500)), 0.03, 0.03, 0.03, n_rate=0.01)
    alphabet = np.frombuffer(b"RYKMBDHVryswU*", dtype=np.uint8)
    qs2, ts2 = [], []
    for k, (q, t) in enumerate(zip(qs, ts)):
        q, t = np.frombuffer(q, np.uint8).copy(), np.frombuffer(t, np.uint8).copy()
        if k % 3 == 0:                      # a third of the pairs get exotic letters on one or both sides
            q[rng.random(q.size) < 0.02] = rng.choice(alphabet)
            if k % 2 == 0:
                t[rng.random(t.size) < 0.02] = rng.choice(alphabet)
        qs2.append(q.tobytes()); ts2.append(t.tobytes())
    for p in (dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751), dict(m=1, x=4, q=6, r=2, s=3, z=50, w=33),
              dict(m=200, x=300, q=4, r=2, s=3, z=400, w=100)):      # last: scores that do not fit the byte profile
        got = _gpu(eng, qs2, ts2, p)
        exp = O.align_pairs(qs2, ts2, O.make_params(**p), wide=True, threads=4)
        assert all((a == b).all() for a, b in zip(got, exp)), p


def test_prepacked_input(eng):
    """The reference's `isPacked` data format (ctors.cpp:65-73): the caller ships 4-bit packed words, no pack kernel runs.
    Packed here on the host by the oracle's restatement of gasal_pack_kernel."""
    import agatha_amd
    qs, ts = synth.make_pairs(11, 300, lambda r: int(r.integers(1, 5000)), 0.03, 0.03, 0.04, n_rate=0.01)
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(**p), wide=True, model=O.MODEL_SLICES, threads=8)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload_packed(O.pack(qb), O.pack(tb))
        b.align(agatha_amd.Scores.make(**p))
        b.download()
        eng.synchronize()
        for k in range(3):
            assert np.array_equal(b.res_host[k], exp[k])
    finally:
        b.free()


def test_seq_ops_kernel_against_the_reference_as_written(eng):
    """f2: agatha::seq_ops_kernel against the fixture produced by the restatement of the reference's
    gasal_reversecomplement_kernel (pack_rc_seqs.h:56-212, oracle/seq_ops_ref.c).  (a) For len % 8 == 0 (and for ops that
    do not reverse) the kernel's packed words EQUAL the reference's; (b) for len % 8 != 0 with a reversal they differ in
    exactly the documented way: the reference, as written, rotates the padding Ns to the front, the kernel reverses the
    len real bases and keeps the padding behind them (= oracle's product semantics)."""
    import json
    d = json.load(open(os.path.join(GOLDEN_DIR, "seq_ops_as_written.json")))
    seqs = [s.encode() for s in d["seqs"]]
    ops = np.asarray(d["ops"], np.uint8)
    ref = np.asarray(d["packed_after"], np.uint32)
    buf, offs, lens = O.make_batch(seqs)
    b = eng.batch(buf, buf, offs, offs, lens, lens)
    try:
        b.upload(); b.pack(); b.seq_ops(ops, None)
        got, untouched = b.packed_host()
    finally:
        b.free()
    assert (untouched == O.pack(buf)).all()
    prod = O.seq_ops(O.pack(buf), lens, offs, ops, as_written=False)
    assert (got == prod).all()
    n_equal = n_diff = 0
    for s, o, off in zip(seqs, ops, offs):
        w0, w1 = int(off) // 8, (int(off) + len(s) + 7) // 8
        if len(s) % 8 == 0 or not (o & 1):
            assert (got[w0:w1] == ref[w0:w1]).all(), (len(s), int(o))              # (a)
            n_equal += 1
        else:
            assert (got[w0:w1] != ref[w0:w1]).any(), (len(s), int(o))              # (b)
            pad = (-len(s)) % 8
            nib = lambda words: [(int(w) >> (28 - 4 * k)) & 15 for w in words for k in range(8)]
            assert nib(ref[w0:w1])[:pad] == [14] * pad and nib(got[w0:w1])[len(s):] == [14] * pad
            n_diff += 1
    assert n_equal > 40 and n_diff > 40


def test_start_positions_against_the_oracle(eng):
    """f4: agatha_amd_align_starts (reverse-prefix kernel + the ordinary align kernels run backwards + starts kernel) against
    the oracle's definition, on pairs with junk in front of either sequence, ragged lengths, z-dropped pairs (their end
    cell lies before the break) and pairs without any positive score."""
    import agatha_amd
    rng = np.random.default_rng(8)
    qs, ts = [], []
    for k in range(300):
        core = synth.random_seq(rng, int(rng.integers(1, 2500)))
        rd = synth.mutate(rng, core, 0.03, 0.03, 0.04)
        if rd.size == 0:
            rd = synth.random_seq(rng, 1)
        a, b = core.tobytes(), rd.tobytes()
        if k % 3 == 1:
            a = synth.random_seq(rng, int(rng.integers(1, 40))).tobytes() + a
        if k % 3 == 2:
            b = synth.random_seq(rng, int(rng.integers(1, 40))).tobytes() + b
        if k % 7 == 0:
            b = b[:len(b) // 2] + synth.random_seq(rng, len(b) // 2 + 1).tobytes()          # breaks: z-drop
        qs.append(a); ts.append(b)
    qs += [b"ACGT", b"A"]; ts += [b"TTTT", b"C"]
    for p in (dict(m=2, x=4, q=4, r=2, s=3, z=100, w=64), dict(m=1, x=4, q=6, r=2, s=3, z=400, w=751)):
        qb, qo, ql = O.make_batch(qs)
        tb, to, tl = O.make_batch(ts)
        P = O.make_params(**p)
        es, eq, et = O.align_batch(qb, tb, qo, to, ql, tl, P, wide=True, threads=8)
        xq, xt, _ = O.start_positions(qs, ts, P, eq, et, threads=8)
        b = eng.batch(qb, tb, qo, to, ql, tl)
        try:
            b.upload(); b.pack(); b.align(_scores(p)); b.download()
            gq, gt = b.align_starts(_scores(p))
            assert (b.res_host[0] == es).all() and (b.res_host[1] == eq).all() and (b.res_host[2] == et).all()
        finally:
            b.free()
        assert (gq == xq).all() and (gt == xt).all()
        assert (gq > 0).sum() > 50 and (gt > 0).sum() > 50 and (gq >= 0).all() and (gq <= eq).all() and (gt <= et).all()


def test_two_bit_input_with_n_mask(eng):
    """2-bit codes + N mask (agatha_amd_pack2_host / agatha_amd_unpack2; `north_star`'s "2-bit-packed reference/query", SURVEY.md
    8 f3's second half): the device words equal the pack kernel's, and the batch aligns to the oracle's results -- with runs of
    N in both sequences (the mask), lower-case letters, ragged lengths."""
    import agatha_amd
    rng = np.random.default_rng(12)
    qs, ts = synth.cfg_c4(n=64, seed=5, lo=50, hi=3000)
    qs = [bytes(q) for q in qs]
    ts = [bytes(t) for t in ts]
    for k in range(0, 64, 5):               # runs of N, lower case
        q = bytearray(qs[k]); a = int(rng.integers(0, max(1, len(q) - 40))); q[a:a + 30] = b"N" * min(30, len(q) - a); qs[k] = bytes(q)
        ts[k] = ts[k].lower() if k % 2 else ts[k][:len(ts[k]) // 2] + b"n" * 7 + ts[k][len(ts[k]) // 2:]
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    exp = O.align_batch(qb, tb, qo, to, ql, tl, O.make_params(), wide=True, model=O.MODEL_SLICES, threads=4)
    cq, mq, oq = agatha_amd.pack2_host(qb)
    ct, mt, ot = agatha_amd.pack2_host(tb)
    assert oq == 0 and ot == 0
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload_packed2(cq, mq, ct, mt)
        pq, pt = b.packed_host()
        assert (pq == O.pack(qb)).all() and (pt == O.pack(tb)).all()
        b.align(agatha_amd.Scores.make()); b.download(); eng.synchronize()
        got = [b.res_host[j].copy() for j in range(3)]
    finally:
        b.free()
    for g, e in zip(got, exp):
        assert (np.asarray(g) == np.asarray(e)).all()
