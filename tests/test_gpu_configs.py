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


def test_c3_ultralong_band1500(eng):
    qs, ts = synth.cfg_c3(n=16)
    p = dict(m=2, x=4, q=4, r=2, s=3, z=400, w=1500)
    batch, got = _run(eng, qs, ts, **p)
    assert eng.last_config() == (64, 3)
    assert max(len(q) for q in qs) > 65536 or max(len(q) for q in qs) > 32768
    _check(batch, got, **p)


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
    # 1 000 of the 10 000 pairs against the block-granular oracle (the restatement of the reference kernel, not the
    # exact-band port), spread over the whole batch: they ran on a static schedule, i.e. some of them were suspended by one
    # lane group and resumed by another
    k = np.sort(rng.choice(10000, 1000, replace=False))
    sb = O.make_batch([qs[i] for i in k]), O.make_batch([ts[i] for i in k])
    exp = O.align_batch(sb[0][0], sb[1][0], sb[0][1], sb[1][1], sb[0][2], sb[1][2], O.make_params(**p), wide=True,
                        model=O.MODEL_SLICES, threads=16)
    assert all((a[k] == b).all() for a, b in zip(got, exp))
    # checksum of checksums for the record (recomputable from the seeds)
    assert int(got[0].sum()) > 0 and (got[1] < batch[4]).all() and (got[2] < batch[5] + 8).all()


def test_identity_pairs_score_linearly(eng):
    rng = np.random.default_rng(5)
    seqs = [synth.random_seq(rng, n).tobytes() for n in (1, 8, 9, 1000, 12345, 40000)]
    _, got = _run(eng, seqs, seqs, m=2, x=4, q=4, r=2, s=3, z=400, w=751)
    for s, L, q, t in zip(got[0], map(len, seqs), got[1], got[2]):
        assert (s, q, t) == (2 * L, L - 1, L - 1)
