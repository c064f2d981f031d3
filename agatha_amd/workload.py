"""Workload helpers of the product package: the GASAL host-batch wire format and the synthetic read-pair generators of
SURVEY.md 8(d) (BASELINE.json configs).  Pure numpy, no alignment arithmetic: bench.py, the tools and the tests all
build their inputs here; nothing in this module touches oracle/.
"""
import numpy as np


def make_batch(seqs):
    """GASAL host-batch wire format (reference host_batch.cpp:79-154): ASCII, each sequence padded with 'N' to a
    multiple of 8 bytes; returns (bytes array, byte offsets, true lengths)."""
    offs, lens, total = [], [], 0
    for s in seqs:
        offs.append(total)
        lens.append(len(s))
        total += (len(s) + 7) & ~7
    buf = np.full(max(total, 8), ord("N"), dtype=np.uint8)
    for s, o in zip(seqs, offs):
        if len(s):
            buf[o:o + len(s)] = np.frombuffer(s if isinstance(s, bytes) else s.encode(), dtype=np.uint8)
    return buf[:total] if total else buf[:0], np.asarray(offs, np.uint32), np.asarray(lens, np.uint32)


def nominal_cells_total(qlen, tlen, w):
    """Sum over pairs of the nominal in-band cells sum_i (min(R-1,i+w) - max(0,i-w) + 1)."""
    tot = 0
    w = int(w)
    for Q, R in zip(np.asarray(qlen, np.int64), np.asarray(tlen, np.int64)):
        i = np.arange(Q, dtype=np.int64)
        tot += int(np.clip(np.minimum(R - 1, i + w) - np.maximum(0, i - w) + 1, 0, None).sum())
    return tot

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_seq(rng, n):
    return _ACGT[rng.integers(0, 4, size=int(n))]


def mutate(rng, ref, sub, ins, dele):
    """Independent per-base channel {substitution, insertion (uniform base), deletion}."""
    n = ref.size
    u = rng.random(n)
    out = []
    keep = u >= dele                       # deletion
    is_sub = (u >= dele) & (u < dele + sub)
    base = ref.copy()
    if is_sub.any():
        shift = rng.integers(1, 4, size=int(is_sub.sum()))
        idx = np.searchsorted(_ACGT, base[is_sub])          # ACGT is sorted in ASCII
        base[is_sub] = _ACGT[(idx + shift) % 4]
    is_ins = rng.random(n) < ins
    ins_base = random_seq(rng, n)
    # interleave: for each position emit [ins_base if is_ins] + [base if keep]
    cnt = is_ins.astype(np.int64) + keep.astype(np.int64)
    pos = np.cumsum(cnt) - cnt
    out = np.empty(int(cnt.sum()), dtype=np.uint8)
    out[pos[is_ins]] = ins_base[is_ins]
    out[(pos + is_ins)[keep]] = base[keep]
    return out


def make_pairs(seed, n, length_fn, sub=0.03, ins=0.03, dele=0.04, n_rate=0.0):
    """Returns two lists of bytes objects: file-1 sequences (DP rows, 'query_batch') = the reference pieces,
    file-2 sequences (DP columns, 'target_batch') = the mutated reads."""
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for _ in range(n):
        L = int(length_fn(rng))
        ref = random_seq(rng, L)
        read = mutate(rng, ref, sub, ins, dele)
        if read.size == 0:
            read = random_seq(rng, 1)
        if n_rate > 0:
            ref = ref.copy()
            ref[rng.random(ref.size) < n_rate] = ord("N")
        qs.append(ref.tobytes())
        ts.append(read.tobytes())
    return qs, ts


def cfg_c1(n=10000, seed=0xA6A70001):
    """10 k synthetic ONT pairs, ~10 kb, band 751 (BASELINE.json configs[1])."""
    f = lambda rng: np.clip(np.rint(rng.normal(10000, 1000)), 8000, 12000)
    return make_pairs(seed, n, f, 0.03, 0.03, 0.04)


def cfg_c0(n=2000, seed=0xA6A70000):
    """Bundled-dataset stand-in (the real dataset is absent from the reference tree)."""
    f = lambda rng: np.clip(np.rint(rng.normal(3000, 1000)), 200, 8000)
    return make_pairs(seed, n, f, 0.04, 0.03, 0.03)


def cfg_c2(n=100000, seed=0xA6A70002):
    f = lambda rng: rng.integers(15000, 20001)
    return make_pairs(seed, n, f, 0.002, 0.004, 0.004)


def cfg_c3(n=256, seed=0xA6A70003):
    f = lambda rng: max(1000, np.rint(rng.normal(100000, 10000)))
    return make_pairs(seed, n, f, 0.03, 0.03, 0.04)


def cfg_c4(n=20000, seed=0xA6A70004, lo=1000, hi=100000):
    """Mixed lengths, 30 % high-error / broken pairs that force z-drop."""
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for _ in range(n):
        L = int(np.exp(rng.uniform(np.log(lo), np.log(hi))))
        ref = random_seq(rng, L)
        if rng.random() < 0.7:
            read = mutate(rng, ref, 0.03, 0.03, 0.04)
        elif rng.random() < 0.5:
            e = rng.uniform(0.25, 0.45) / 3
            read = mutate(rng, ref, e, e, e)
        else:
            bp = int(rng.integers(0, L))
            read = np.concatenate([mutate(rng, ref[:bp], 0.03, 0.03, 0.04), random_seq(rng, L - bp)])
        if read.size == 0:
            read = random_seq(rng, 1)
        qs.append(ref.tobytes())
        ts.append(read.tobytes())
    return qs, ts


def add_n_runs(seqs, frac, seed=1, lo=50, hi=1000):
    """A run of N (lo..hi bases, uniform position) in a fraction of the sequences: what reference-genome pieces look like
    around assembly gaps.  Returns a new list."""
    rng = np.random.default_rng(seed)
    out = []
    for s in seqs:
        if rng.random() < frac and len(s) > 8:
            a = np.frombuffer(s, dtype=np.uint8).copy()
            n = int(min(rng.integers(lo, hi + 1), a.size - 1))
            at = int(rng.integers(0, a.size - n + 1))
            a[at:at + n] = ord("N")
            s = a.tobytes()
        out.append(s)
    return out


# ---- the same workload shapes, generated in CHUNKS of consecutive pairs with a random stream per chunk, for the strong-scaling
# ---- form of the bench: a rank only generates the chunks it owns (bench.py --scaling strong, agatha_amd/shard.py).  A chunk's
# ---- pairs are made with whole-array operations (all reference bases of the chunk at once, one pass of the error channel).
CHUNK = 64
_LEN_LAWS = {
    "cfg_c0": lambda rng, n: np.clip(np.rint(rng.normal(3000, 1000, n)), 200, 8000),
    "cfg_c1": lambda rng, n: np.clip(np.rint(rng.normal(10000, 1000, n)), 8000, 12000),
    "cfg_c2": lambda rng, n: rng.integers(15000, 20001, n),
    "cfg_c3": lambda rng, n: np.maximum(1000, np.rint(rng.normal(100000, 10000, n))),
}
_CHANNELS = {"cfg_c0": (0.04, 0.03, 0.03), "cfg_c1": (0.03, 0.03, 0.04), "cfg_c2": (0.002, 0.004, 0.004), "cfg_c3": (0.03, 0.03, 0.04)}


def chunked_lengths(gen, n, seed):
    """Reference lengths of all n pairs (cheap: one draw per pair), identical on every rank."""
    return _LEN_LAWS[gen](np.random.default_rng([seed, 0x1e47]), n).astype(np.int64)


def chunked_pairs(gen, seed, lens, chunk_ids):
    """The pairs of the given chunks (CHUNK consecutive pairs each) in the GASAL wire format:
    (qbuf, tbuf, qoff, toff, qlen, tlen, pair ids)."""
    sub, ins, dele = _CHANNELS[gen]
    n = len(lens)
    def one(c):
        lo, hi = int(c) * CHUNK, min(n, (int(c) + 1) * CHUNK)
        L = lens[lo:hi]
        rng = np.random.default_rng([seed, 0xc4a9, int(c)])
        total = int(L.sum())
        ref = _ACGT[rng.integers(0, 4, size=total, dtype=np.uint8)]
        u = rng.random(total, dtype=np.float32)
        keep = u >= dele
        is_sub = keep & (u < dele + sub)
        base = ref.copy()
        ns = int(is_sub.sum())
        if ns:
            base[is_sub] = _ACGT[(np.searchsorted(_ACGT, base[is_sub]) + rng.integers(1, 4, size=ns)) % 4]
        is_ins = rng.random(total, dtype=np.float32) < ins
        ins_base = _ACGT[rng.integers(0, 4, size=total, dtype=np.uint8)]
        cnt = is_ins.astype(np.int64) + keep.astype(np.int64)
        pos = np.cumsum(cnt) - cnt
        read = np.empty(int(cnt.sum()), np.uint8)
        read[pos[is_ins]] = ins_base[is_ins]
        read[(pos + is_ins)[keep]] = base[keep]
        starts = np.concatenate([[0], np.cumsum(L)[:-1]])
        rl = np.add.reduceat(cnt, starts)
        return (ref, L), (read, rl), L, np.maximum(rl, 1), np.arange(lo, hi, dtype=np.int64)      # (an empty read gets one base, as in make_pairs)

    # numpy releases the GIL in these array passes: a few threads make a rank's shard in a fraction of the time
    from concurrent.futures import ThreadPoolExecutor
    import os
    chunk_ids = list(chunk_ids)
    workers = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4, len(chunk_ids)))
    with ThreadPoolExecutor(workers) as ex:
        done = list(ex.map(one, chunk_ids))
    qparts = [d[0] for d in done]; tparts = [d[1] for d in done]
    ql = [d[2] for d in done]; tl = [d[3] for d in done]; ids = [d[4] for d in done]

    def assemble(parts, lens_true):
        lt = np.concatenate(lens_true).astype(np.int64)
        sizes = (lt + 7) // 8 * 8
        off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        buf = np.full(int(sizes.sum()), ord("N"), np.uint8)
        k = 0
        for flat, raw in parts:             # one slice copy per sequence (no index arrays of the batch's size)
            src = 0
            for r in raw.tolist():
                if r:
                    buf[off[k]:off[k] + r] = flat[src:src + r]
                else:
                    buf[off[k]] = ord("A")
                src += r
                k += 1
        return buf, off.astype(np.uint32), lt.astype(np.uint32)

    if not qparts:
        z8, z32 = np.zeros(0, np.uint8), np.zeros(0, np.uint32)
        return z8, z8, z32, z32, z32, z32, np.zeros(0, np.int64)
    qb, qo, qlen = assemble(qparts, ql)
    tb, to, tlen = assemble(tparts, tl)
    return qb, tb, qo, to, qlen, tlen, np.concatenate(ids)
