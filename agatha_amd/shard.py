"""Sharding a batch of independent sequence pairs over the GPUs of one node, and gathering the results.

The reference is single-GPU (its gasal_set_device hook is never called, interfaces.cpp:86-116 / test_prog.cpp:31).
Pairs are independent, so the path shards with no data-path exchange: every rank aligns its own subset and the only
collective is one all-gather of 3 x int32 per pair (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests).
"""
import numpy as np


def nominal_cells(qlen, tlen, w):
    """In-band DP cells per pair, closed form of sum_i (min(R-1, i+w) - max(0, i-w) + 1)."""
    Q = np.asarray(qlen, np.int64)
    R = np.asarray(tlen, np.int64)
    w = int(w)
    out = np.zeros(Q.shape, np.int64)
    for k in range(Q.size):          # vectorised per pair over rows would cost O(sum Q); use the piecewise closed form
        q, r = int(Q[k]), int(R[k])
        if q <= 0 or r <= 0:
            continue
        # rows 0..q-1; hi(i) = min(r-1, i+w), lo(i) = max(0, i-w); count = hi - lo + 1 when >= 1
        i = np.arange(q, dtype=np.int64) if q < 4096 else None
        if i is not None:
            c = np.minimum(r - 1, i + w) - np.maximum(0, i - w) + 1
            out[k] = int(np.clip(c, 0, None).sum())
        else:
            out[k] = _cells_closed(q, r, w)
    return out


def _cells_closed(q, r, w):
    # sum over rows of clip(min(r-1, i+w) - max(0, i-w) + 1, 0): piecewise linear, evaluate by segments
    pts = sorted(set([0, q, max(0, min(q, w)), max(0, min(q, r - 1 - w)), max(0, min(q, r + w))]))
    total = 0
    for a, b in zip(pts[:-1], pts[1:]):
        if b <= a:
            continue
        # the summand is linear on [a, b-1]: evaluate at both ends
        def f(i):
            return max(0, min(r - 1, i + w) - max(0, i - w) + 1)
        fa, fb = f(a), f(b - 1)
        total += (fa + fb) * (b - a) // 2
    return total


def lpt_partition(cost, world):
    """Longest-processing-time-first: deal pairs, most expensive first, to the least loaded rank.
    Returns a list of index arrays (original pair ids per rank)."""
    cost = np.asarray(cost, np.int64)
    order = np.argsort(-cost, kind="stable")
    loads = np.zeros(world, np.int64)
    parts = [[] for _ in range(world)]
    for idx in order:
        g = int(np.argmin(loads))
        parts[g].append(int(idx))
        loads[g] += int(cost[idx])
    return [np.asarray(sorted(p), np.int64) for p in parts]


def take_pairs(qbuf, tbuf, qoff, toff, qlen, tlen, idx):
    """Sub-batch (GASAL wire format) holding the pairs `idx`, re-packed contiguously."""
    def side(buf, off, ln):
        sizes = (np.asarray(ln, np.int64)[idx] + 7) // 8 * 8
        noff = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint32) if len(idx) else np.zeros(0, np.uint32)
        out = np.empty(int(sizes.sum()), np.uint8)
        for k, i in enumerate(idx):
            out[noff[k]:noff[k] + sizes[k]] = buf[off[i]:off[i] + sizes[k]]
        return out, noff, np.asarray(ln, np.uint32)[idx]
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
