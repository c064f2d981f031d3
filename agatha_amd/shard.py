"""Sharding a batch of independent sequence pairs over the GPUs of one node, and gathering the results.

The reference is single-GPU (its gasal_set_device hook is never called, interfaces.cpp:86-116 / test_prog.cpp:31).
Pairs are independent, so the path shards with no data-path exchange: every rank aligns its own subset and the only
collective is one all-gather of 3 x int32 per pair (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests).
"""
import numpy as np


def nominal_cells(qlen, tlen, w):
    """In-band DP cells per pair, sum_i clip(min(R-1, i+w) - max(0, i-w) + 1, 0) over the rows i < Q, in
    closed form and vectorised over the pairs: the summand is piecewise linear in i with breakpoints at w + 1, R - w and
    R + w, so each pair is at most four trapezoids."""
    Q = np.asarray(qlen, np.int64)
    R = np.asarray(tlen, np.int64)
    w = np.int64(w)

    def f(i):           # the summand at row i (arrays)
        return np.clip(np.minimum(R - 1, i + w) - np.maximum(0, i - w) + 1, 0, None)

    zero = np.zeros_like(Q)
    # rows are cut at the sorted, clipped breakpoints; on every piece [a, b) the summand is linear (or identically 0)
    cuts = np.sort(np.stack([zero, np.clip(w + 1, 0, Q), np.clip(R - w, 0, Q), np.clip(R + w, 0, Q), np.clip(Q, 0, None)]), axis=0)
    total = np.zeros_like(Q)
    for k in range(4):
        a, b = cuts[k], cuts[k + 1]
        n = np.clip(b - a, 0, None)
        total += (f(a) + f(np.maximum(b - 1, a))) * n // 2          # trapezoid: (first + last) * count / 2 (always even)
    return np.where((Q > 0) & (R > 0), total, 0)


def lpt_partition(cost, world):
    """Longest-processing-time-first: deal pairs, most expensive first, to the least loaded rank (a heap: O(n log G)).
    Returns a list of index arrays (original pair ids per rank, ascending)."""
    import heapq
    cost = np.asarray(cost, np.int64)
    order = np.argsort(-cost, kind="stable")
    heap = [(0, g) for g in range(world)]
    owner = np.empty(cost.size, np.int32)
    for idx, c in zip(order.tolist(), cost[order].tolist()):
        load, g = heap[0]
        owner[idx] = g
        heapq.heapreplace(heap, (load + c, g))
    return [np.flatnonzero(owner == g).astype(np.int64) for g in range(world)]


def take_pairs(qbuf, tbuf, qoff, toff, qlen, tlen, idx):
    """Sub-batch (GASAL wire format) holding the pairs `idx`, re-packed contiguously."""
    idx = np.asarray(idx, np.int64)

    def side(buf, off, ln):
        # one gather over the whole side: byte j of the output comes from (source offset - destination offset of its pair) + j
        sizes = (np.asarray(ln, np.int64)[idx] + 7) // 8 * 8
        noff = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64) if len(idx) else np.zeros(0, np.int64)
        total = int(sizes.sum())
        src = np.repeat(np.asarray(off, np.int64)[idx] - noff, sizes) + np.arange(total, dtype=np.int64)
        return np.asarray(buf)[src], noff.astype(np.uint32), np.asarray(ln, np.uint32)[idx]
    qb, qo, ql = side(qbuf, qoff, qlen)
    tb, to, tl = side(tbuf, toff, tlen)
    return qb, tb, qo, to, ql, tl


def gather_results(local, idx, n_total, dist=None, device=None):
    """All-gather the per-rank results (three int32 arrays for the pairs `idx`) and scatter them back to input order.
    `dist` is torch.distributed (already initialised) or None for a single process."""
    local = np.stack([np.asarray(a, np.int32) for a in local])          # (3, n_local)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = np.zeros((3, n_total), np.int32)
        out[:, idx] = local
        return out
    import torch
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[1]], dtype=torch.int64, device=dev))
    nmax = int(max(int(c.item()) for c in counts))
    pad = torch.full((4, nmax), -1, dtype=torch.int32, device=dev)      # row 3 carries the original pair ids
    pad[:3, :local.shape[1]] = torch.from_numpy(local).to(dev)
    pad[3, :local.shape[1]] = torch.from_numpy(np.asarray(idx, np.int32)).to(dev)
    allr = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(allr, pad)
    out = np.zeros((3, n_total), np.int32)
    for g in range(world):
        a = allr[g].cpu().numpy()
        k = int(counts[g].item())
        out[:, a[3, :k]] = a[:3, :k]
    return out


def gather_results_tensor(local, idx, n_total, dist, torch):
    """The strong-scaling exchange on tensors (GPU tensors over RCCL in bench.py, CPU tensors over gloo in the tests):
    `local` is this rank's (3, n_local) int32 result tensor, `idx` the (n_local,) int64 original pair ids.  One
    all-gather of 4 x nmax int32 per rank (results + ids, padded to the largest shard), then a scatter into input order.
    Returns the (3, n_total) tensor on every rank."""
    world = dist.get_world_size()
    n_local = int(local.shape[1])
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    counts[dist.get_rank()] = n_local
    dist.all_reduce(counts)
    nmax = int(counts.max().item())
    pad = torch.full((4, nmax), -1, dtype=torch.int32, device=local.device)
    pad[:3, :n_local] = local
    pad[3, :n_local] = idx.to(torch.int32)
    flat = torch.empty((world * 4, nmax), dtype=torch.int32, device=local.device)      # concatenation along dim 0 (NCCL and gloo)
    dist.all_gather_into_tensor(flat, pad)
    allr = flat.view(world, 4, nmax)
    ids = allr[:, 3, :].reshape(-1).to(torch.int64)
    vals = allr[:, :3, :].permute(1, 0, 2).reshape(3, -1)
    keep = ids >= 0
    out = torch.zeros((3, n_total), dtype=torch.int32, device=local.device)
    out[:, ids[keep]] = vals[:, keep]
    return out
