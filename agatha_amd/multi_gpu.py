"""Align one FASTA pair set on all GPUs of a node: one process per GPU, pairs sharded, results gathered over RCCL.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        -m agatha_amd.multi_gpu -m 2 -x 4 -q 4 -r 2 -s 3 -z 400 -w 751 ref.fasta query.fasta > score.log

Same scoring flags and output lines as the single-GPU CLI (`agatha_amd/manual -p`, reference test_prog.cpp:361-369),
always in input order.  The reference is single-GPU (its gasal_set_device hook is never called, test_prog.cpp:31):
this front end is the "shard the batch over the 8 GPUs, gather 12 bytes per pair" step of the design.
Every rank INDEXES both files (a vectorised scan for line starts: record offsets, op codes and lengths, no sequence is
decoded), takes its LPT share of the pairs by nominal cells, reads only the byte ranges of its own records, runs the
ordinary single-GPU hot path on them and joins one all-gather.
"""
import argparse
import os
import sys

import numpy as np

from . import shard, workload
from .engine import Engine, Scores

_OPS = b"></+"


def read_fasta(path):
    """Records start with one of > < / + (op code 0..3, test_prog.cpp:83-92); sequence lines are concatenated."""
    seqs, ops, cur = [], [], None
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if not line:
                continue
            k = _OPS.find(line[:1])
            if k >= 0:
                if cur is not None:
                    seqs.append(b"".join(cur))
                ops.append(k)
                cur = []
            elif cur is not None:
                cur.append(line)
            else:
                raise SystemExit("Batch1 and target_batch files should be fasta having same number of sequences")
    if cur is not None:
        seqs.append(b"".join(cur))
    return seqs, np.asarray(ops, np.uint8)


def fasta_index(path, chunk=1 << 26):
    """Record table of a FASTA file without decoding it: (ops uint8[n], start int64[n], end int64[n], length int64[n]) --
    op code of the header character, byte range of the record's sequence lines, number of bases.  Vectorised: newline
    positions are collected chunk by chunk, the first byte of every line tells header lines from sequence lines."""
    size = os.path.getsize(path)
    if size == 0:
        return np.zeros(0, np.uint8), np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64)
    if True:
        buf = np.memmap(path, dtype=np.uint8, mode="r")
        nls, ncr = [], []
        for lo in range(0, size, chunk):
            part = buf[lo:lo + chunk]
            nls.append(np.flatnonzero(part == 10).astype(np.int64) + lo)
            cr = np.flatnonzero(part == 13).astype(np.int64) + lo
            if cr.size:
                ncr.append(cr)
        nl = np.concatenate(nls)
        starts = np.concatenate([[0], nl + 1]).astype(np.int64)
        starts = starts[starts < size]                                   # line starts
        ends = np.concatenate([nl, [size]])[:starts.size]                # position of each line's newline (or EOF)
        first = buf[starts]
        nonempty = ends > starts
        hdr = np.isin(first, np.frombuffer(_OPS, np.uint8)) & nonempty
        hidx = np.flatnonzero(hdr)
        if hidx.size == 0 or (nonempty[:hidx[0]].any()):
            raise SystemExit("Batch1 and target_batch files should be fasta having same number of sequences")
        ops = np.searchsorted(np.sort(np.frombuffer(_OPS, np.uint8)), first[hidx])
        ops = np.argsort(np.frombuffer(_OPS, np.uint8))[ops].astype(np.uint8)      # index into "></+"
        rstart = np.minimum(ends[hidx] + 1, size)                        # first byte after the header line
        rend = np.concatenate([starts[hidx[1:]], [size]])                # start of the next header (or EOF)
        # bases = bytes of the region minus its newlines (and carriage returns)
        nl_before = np.searchsorted(nl, rend) - np.searchsorted(nl, rstart)
        length = rend - rstart - nl_before
        if ncr:
            crs = np.concatenate(ncr)
            length -= np.searchsorted(crs, rend) - np.searchsorted(crs, rstart)
        del buf, first
    return ops, rstart, rend, length.astype(np.int64)


def read_records(path, start, end, ids):
    """The sequences of the records `ids` only (byte ranges from fasta_index), newlines stripped."""
    out = []
    with open(path, "rb") as f:
        for i in ids:
            f.seek(int(start[i]))
            out.append(f.read(int(end[i] - start[i])).replace(b"\n", b"").replace(b"\r", b""))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(add_help=True)
    for flag, dest, default in (("-m", "m", 2), ("-x", "x", 4), ("-q", "q", 4), ("-r", "r", 2), ("-s", "s", 3),
                                ("-z", "z", 400), ("-w", "w", 751), ("-a", "a", 1 << 20)):
        ap.add_argument(flag, dest=dest, type=int, default=default)
    ap.add_argument("-c", dest="ops", action="store_true", help="apply the reverse/complement header op codes")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("query_batch_fasta")
    ap.add_argument("target_batch_fasta")
    a = ap.parse_args(argv)

    sys.stdout.flush()
    saved_stdout = os.dup(1)          # RCCL may print its banner on fd 1: results only go to the real stdout
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    dev = None
    if world > 1 or "WORLD_SIZE" in os.environ:
        import torch
        import torch.distributed as dist
        if a.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dev = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    qops, qstart, qend, ql = fasta_index(a.query_batch_fasta)
    tops, tstart, tend, tl = fasta_index(a.target_batch_fasta)
    if len(ql) != len(tl) or len(ql) == 0:
        raise SystemExit("Batch1 and target_batch files should be fasta having same number of sequences")
    n = len(ql)
    mine = shard.lpt_partition(shard.nominal_cells(ql, tl, a.w), world)[rank]

    eng = Engine(local_rank)
    scores = Scores.make(m=a.m, x=a.x, q=a.q, r=a.r, s=a.s, z=a.z, w=a.w)
    local = [np.zeros(len(mine), np.int32) for _ in range(3)]
    for lo in range(0, len(mine), a.a):                 # batches of -a pairs, as the reference CLI cuts them
        idx = mine[lo:lo + a.a]
        qb, qo, qlen = workload.make_batch(read_records(a.query_batch_fasta, qstart, qend, idx))     # this rank's byte ranges only
        tb, to, tlen = workload.make_batch(read_records(a.target_batch_fasta, tstart, tend, idx))
        res = eng.align_host_batch(qb, tb, qo, to, qlen, tlen, scores,
                                   qops=qops[idx] if a.ops else None, tops=tops[idx] if a.ops else None)
        for k in range(3):
            local[k][lo:lo + len(idx)] = res[k]
    full = shard.gather_results(local, mine, n, dist, device=dev)
    if rank == 0:
        with os.fdopen(os.dup(saved_stdout), "w") as out:
            for k in range(n):
                out.write(f"{full[0, k]}\tquery_batch_end={full[1, k]}\ttarget_batch_end={full[2, k]}\n")
    eng.close()
    if dist is not None and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
