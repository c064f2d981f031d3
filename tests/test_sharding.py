"""CPU-only: multi-GPU sharding logic, including a real 2-process gather over the gloo backend."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as O, synth
from agatha_amd import shard


def test_nominal_cells_closed_form_matches_oracle():
    rng = np.random.default_rng(0)
    Q = rng.integers(1, 12000, 40)
    R = rng.integers(1, 12000, 40)
    for w in (0, 5, 100, 751):
        got = shard.nominal_cells(Q, R, w)
        exp = np.array([O.nominal_cells(int(q), int(r), w) for q, r in zip(Q, R)])
        assert (got == exp).all()


def test_lpt_partition_is_a_balanced_bijection():
    rng = np.random.default_rng(1)
    cost = rng.integers(1, 10 ** 6, 1000)
    for world in (1, 2, 4, 8):
        parts = shard.lpt_partition(cost, world)
        allidx = np.sort(np.concatenate(parts))
        assert (allidx == np.arange(1000)).all()
        loads = np.array([cost[p].sum() for p in parts])
        assert loads.max() <= loads.mean() * 1.02 + cost.max()


def test_take_pairs_roundtrip():
    qs, ts = synth.make_pairs(3, 20, lambda r: int(r.integers(1, 300)))
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    idx = np.array([1, 5, 6, 19])
    sqb, stb, sqo, sto, sql, stl = shard.take_pairs(qb, tb, qo, to, ql, tl, idx)
    for k, i in enumerate(idx):
        assert bytes(sqb[sqo[k]:sqo[k] + sql[k]]) == qs[i] and bytes(stb[sto[k]:sto[k] + stl[k]]) == ts[i]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    qs, ts = synth.cfg_c4(n=61, seed=5, lo=50, hi=1500)           # every rank builds the same job
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    P = O.make_params(w=100, z=100)
    parts = shard.lpt_partition(shard.nominal_cells(ql, tl, 100), world)
    sub = shard.take_pairs(qb, tb, qo, to, ql, tl, parts[rank])
    # the per-shard aligner is the oracle here (no GPU in this container); the sharding/gather code is the product's
    local = O.align_batch(*sub, P, wide=True)
    full = shard.gather_results(local, parts[rank], len(ql), dist)
    exp = np.stack(O.align_batch(qb, tb, qo, to, ql, tl, P, wide=True))
    ret[rank] = bool((full == exp).all())
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_gloo_gather_restores_input_order():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def _strong_worker(rank, world, port, ret):
    """The strong-scaling path of bench.py (--scaling strong) over gloo: LPT partition of ONE batch, every rank aligns
    its share (the oracle stands in for the GPU here), shard.gather_results_tensor restores input order on every rank."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    qs, ts = synth.cfg_c4(n=97, seed=11, lo=50, hi=2500)
    qb, qo, ql = O.make_batch(qs)
    tb, to, tl = O.make_batch(ts)
    P = O.make_params(w=100, z=100)
    cost = shard.nominal_cells(ql, tl, 100)
    parts = shard.lpt_partition(cost, world)
    loads = np.array([cost[p].sum() for p in parts])
    sub = shard.take_pairs(qb, tb, qo, to, ql, tl, parts[rank])
    local = torch.from_numpy(np.stack(O.align_batch(*sub, P, wide=True)).astype(np.int32))
    full = shard.gather_results_tensor(local, torch.from_numpy(parts[rank]), len(ql), dist, torch).numpy()
    exp = np.stack(O.align_batch(qb, tb, qo, to, ql, tl, P, wide=True))
    ret[rank] = bool((full == exp).all()) and bool(loads.max() <= loads.mean() * 1.05 + cost.max())
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_path_two_process_gloo():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_strong_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def _chunk_worker(rank, world, port, ret):
    """bench.py --scaling strong on a shape with a chunked generator: every rank draws all the lengths, deals the CHUNKS of 64
    consecutive pairs by LPT, generates ONLY its own chunks (a random stream per chunk), aligns them (the oracle stands in for the
    GPU) and the one all-gather restores input order; compared with the whole batch generated and aligned in one piece."""
    import torch
    import torch.distributed as dist
    from agatha_amd import workload
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, seed, gen = 300, 41, "cfg_c0"
    lens = workload.chunked_lengths(gen, n, seed)
    nch = (n + workload.CHUNK - 1) // workload.CHUNK
    cost = np.add.reduceat(shard.nominal_cells(lens, lens, 100), np.arange(0, n, workload.CHUNK))
    parts = shard.lpt_partition(cost, world)
    qb, tb, qo, to, ql, tl, ids = workload.chunked_pairs(gen, seed, lens, parts[rank])
    P = O.make_params(w=100, z=100)
    local = torch.from_numpy(np.stack(O.align_batch(qb, tb, qo, to, ql, tl, P, wide=True)).astype(np.int32))
    full = shard.gather_results_tensor(local, torch.from_numpy(ids), n, dist, torch).numpy()
    whole = workload.chunked_pairs(gen, seed, lens, range(nch))
    exp = np.stack(O.align_batch(*whole[:6], P, wide=True))
    own = sorted(int(c) for c in parts[rank])
    ret[rank] = bool((full == exp).all()) and bool((whole[6] == np.arange(n)).all()) and len(ids) == sum(min(n, (c + 1) * 64) - c * 64 for c in own)
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_with_per_rank_chunks_two_process_gloo():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_chunk_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def test_chunked_generator_is_the_same_whatever_the_shard():
    """A chunk's pairs do not depend on which other chunks are generated with it, the lengths follow the shape's law, reads are the
    references through the shape's error channel (lengths within a few per cent), the wire format is the GASAL host batch."""
    from agatha_amd import workload
    lens = workload.chunked_lengths("cfg_c1", 1000, 7)
    assert lens.min() >= 8000 and lens.max() <= 12000 and abs(lens.mean() - 10000) < 200
    a = workload.chunked_pairs("cfg_c1", 7, lens, [5])
    b = workload.chunked_pairs("cfg_c1", 7, lens, [2, 5, 9])
    k = 64
    assert (a[4] == b[4][k:2 * k]).all() and (a[5] == b[5][k:2 * k]).all() and (a[6] == b[6][k:2 * k]).all()
    assert bytes(a[0]) == bytes(b[0][b[2][k]:b[2][2 * k]]) and bytes(a[1]) == bytes(b[1][b[3][k]:b[3][2 * k]])
    assert (a[2] % 8 == 0).all() and (a[4] == lens[320:384]).all()
    assert np.abs(a[5].astype(np.int64) - a[4]).max() < 0.05 * a[4].max()
    assert set(np.unique(a[0]).tolist()) <= set(b"ACGTN")


def test_fasta_index_reads_only_byte_ranges(tmp_path):
    """agatha_amd.multi_gpu: the vectorised FASTA index (offsets, op codes, lengths without decoding a sequence) and the
    per-record reads agree with the plain sequential reader, for ragged line widths, CRLF, empty lines and every op code."""
    from agatha_amd import multi_gpu as M
    rng = np.random.default_rng(4)
    recs = [(">", b""), ("<", b"A")] + [("></+"[k % 4], synth.random_seq(rng, int(rng.integers(1, 700))).tobytes()) for k in range(60)]
    for variant in range(3):
        path = tmp_path / f"v{variant}.fa"
        with open(path, "wb") as f:
            for k, (op, s) in enumerate(recs):
                eol = b"\r\n" if variant == 1 else b"\n"
                f.write(op.encode() + b"rec%d extra text" % k + eol)
                width = [61, 80, 7][variant]
                for i in range(0, len(s), width):
                    f.write(s[i:i + width] + eol)
                if variant == 2 and k % 5 == 0:
                    f.write(eol)                                   # empty line
            if variant == 0:
                f.seek(-1, 2); f.truncate()                        # no newline at the end of the file
        seqs, ops = M.read_fasta(str(path))
        iops, start, end, length = M.fasta_index(str(path), chunk=997)
        assert (iops == ops).all() and length.tolist() == [len(s) for s in seqs]
        ids = [0, 1, 5, 17, 61]
        assert M.read_records(str(path), start, end, ids) == [seqs[i] for i in ids]
