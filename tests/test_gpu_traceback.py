"""GPU: alignment paths (cigar / n_cigar_ops of gasal.h:91-92) through the C-ABI against the oracle's
traceback, plus the size-independent check: a path re-scored from the sequences alone gives the reported score and uses
exactly query_end + 1 / target_end + 1 bases."""
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


def _pairs(seed, n, lo, hi, sub=0.05, ins=0.04, dele=0.04, n_every=7):
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for k in range(n):
        ln = int(rng.integers(lo, hi))
        q = synth.random_seq(rng, ln)
        t = synth.mutate(rng, q, sub, ins, dele)
        if n_every and k % n_every == 0:
            q = q.copy()
            q[rng.integers(0, ln)] = ord('N')
        qs.append(bytes(q))
        ts.append(bytes(t))
    return qs, ts


def _traceback(eng, qs, ts, scratch_bytes=None, **p):
    import agatha_amd
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    b = eng.batch(qb, tb, qo, to, ql, tl)
    try:
        b.upload()
        b.pack()
        return b.align_traceback(agatha_amd.Scores.make(**p), scratch_bytes=scratch_bytes)
    finally:
        b.free()


def _check(qs, ts, got, threads=8, **p):
    params = O.make_params(**p)
    s, qe, te, cig = O.traceback_pairs(qs, ts, params, threads=threads)
    assert (got[0] == s).all() and (got[1] == qe).all() and (got[2] == te).all()
    for k, (a, b) in enumerate(zip(got[3], cig)):
        assert a == b, (k, p, None if a is None else a[:8], None if b is None else b[:8])
    paths = 0
    for k, c in enumerate(got[3]):
        if c is None or s[k] <= 0:
            continue
        paths += 1
        assert O.cigar_rescore(c, qs[k], ts[k], params) == (s[k], qe[k] + 1, te[k] + 1)
    return paths


@pytest.mark.parametrize("w,z,lo,hi,n", [(751, 400, 1, 3000, 48), (16, 400, 1, 300, 300), (3, -1, 1, 60, 400),
                                         (100, 50, 1, 800, 200), (0, 400, 1, 40, 60), (40, -1, 1, 500, 120),
                                         (200, 400, 500, 4000, 40), (1500, 400, 3000, 9000, 6)])
def test_paths_match_oracle(eng, w, z, lo, hi, n):
    qs, ts = _pairs(1000 + w, n, lo, hi)
    got = _traceback(eng, qs, ts, w=w, z=z)
    assert _check(qs, ts, got, w=w, z=z) > n // 2


def test_paths_in_several_passes(eng):
    """A scratch area that holds 5 pairs: the batch goes through in passes and nothing changes."""
    import agatha_amd
    qs, ts = _pairs(77, 64, 200, 1500)
    p = dict(w=64, z=400)
    lib = eng.lib
    import ctypes as C
    per = lib.agatha_amd_traceback_pair_bytes(max(map(len, qs)), max(map(len, ts)), C.byref(agatha_amd.Scores.make(**p)))
    whole = _traceback(eng, qs, ts, **p)
    parts = _traceback(eng, qs, ts, scratch_bytes=5 * per, **p)
    assert all((a == b).all() for a, b in zip(whole[:3], parts[:3])) and whole[3] == parts[3]
    _check(qs, ts, parts, **p)


def test_broken_pairs_and_other_scorings(eng):
    """z-dropped pairs (unrelated tails), empty alignments, other letters, a scoring outside the byte profile."""
    rng = np.random.default_rng(5)
    qs, ts = _pairs(6, 80, 50, 1200)
    for k in range(0, 80, 3):                                   # unrelated second half: z-drop ends the extension early
        t = bytearray(ts[k]); h = len(t) // 2
        t[h:] = bytes(synth.random_seq(rng, len(t) - h)); ts[k] = bytes(t)
    qs[1] = b"ACGT" * 30; ts[1] = b"TGCA" * 30                    # no positive cell: empty alignment
    qs[2] = b"ACGTRYACGTAC" * 20; ts[2] = b"ACGTRYACGTAC" * 20   # letters outside ACGTN
    for p in (dict(m=2, x=4, q=4, r=2, z=100, w=200), dict(m=1, x=4, q=6, r=2, z=400, w=64),
              dict(m=200, x=300, q=400, r=100, z=20000, w=100)):
        got = _traceback(eng, qs, ts, **p)
        _check(qs, ts, got, **p)
    assert got[3][1] in (b"", None) or got[0][1] > 0
