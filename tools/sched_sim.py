"""CPU only: the preemptive static schedule of the packed-int16 kernel as it is since round 5 -- McNaughton's wrap-around, the intervals
dealt to the physical lane groups by the length of their rests, the POOL of the suspended pairs' rests (schedule_kernel in
agatha_amd/csrc/align_kernel.hip, the acquisition loop of align16_acquire.inc) -- restated in numpy, and an event simulation of a launch on
it: four lane groups per wave step together, a wave-step costs what its slowest form costs (key steps while ANY of its pairs wants them),
a lane group that is done with the fixed part of its interval draws the longest rest nobody has taken, waits for one that is still RUNNING,
takes one that is FRESH whole.  What it is for:

  * tests/test_schedule_pool_model.py pins the properties the device code relies on (every step of every pair runs exactly once, nobody
    waits for ever, a launch in which everybody is on time is McNaughton's schedule, a first part nobody started is taken whole);
  * `python tools/sched_sim.py` prices what the step counts of a batch cannot show: the kernel ends with its LAST wave, so pairs that run
    extra steps (a trip back to a checkpoint, key steps behind it) cost the launch the worst wave's surplus, not the batch's (DESIGN.md 3.6,
    6 item 0).  C1 with a burst of errors in every tenth read, probation off and on, against the chip's numbers.

The simulation knows nothing of SIMDs, XCDs or the instruction cache: two costs per wave-step (value / key form, measured: 8.2 / 11.5 us with
two waves per SIMD), every wave resident from t = 0 unless told otherwise."""
import heapq
import sys

import numpy as np

PAIR_OVERHEAD = 16          # kMigPairOverheadSteps * (64 / G) for G = 16
VALUE_US, KEY_US = 8.2, 11.5
FRESH, RUNNING, SAVED, DONE, STOLEN = range(5)


def build(p, m, num_cus=256, identity=False, bins=2048):
    """schedule_kernel: p = step counts in sorted order (0 = a pair this kernel skips), m lane groups.  Returns a dict: T, cum, rest[b] (b = 0..m),
    cross[b] (the pair across boundary b, -1), pool (boundaries by rest, longest first), perm (physical lane group -> interval)."""
    p = np.asarray(p, np.int64)
    n = len(p)
    cum = np.concatenate([[0], np.cumsum(p)])
    P, pm = int(cum[-1]), int(p.max())
    T = max(pm, -(-P // m), 1)
    rest = np.zeros(m + 1, np.int64)
    cross = np.full(m + 1, -1, np.int64)
    for b in range(1, m):
        at = b * T
        j = int(np.searchsorted(cum[1:], at, side="right"))        # first pair with cum[j + 1] > at
        if j < n and cum[j] < at:
            rest[b] = at - cum[j]; cross[b] = j
    bin_of = lambda r: (bins - 1) - (r * (bins - 1)) // T           # descending: bin 0 holds the longest rests
    # (a counting sort by bin; inside a bin the device's order is that of its atomics -- here: by boundary)
    pool = [b for b in sorted(range(m), key=lambda b: (bin_of(int(rest[b])), b)) if rest[b] > 0]
    perm = np.arange(m)
    gpb = m // (2 * num_cus) if m // (2 * num_cus) > 0 else 1
    if not identity and m == 2 * num_cus * gpb:
        order = sorted(range(m), key=lambda g: (bin_of(int(rest[g + 1])), g))      # rank u -> interval
        for u, g in enumerate(order):
            cu, within = u // (2 * gpb), u % (2 * gpb)
            blk = cu if within < gpb else cu + num_cus
            perm[blk * gpb + within % gpb] = g
    return dict(T=T, cum=cum, rest=rest, cross=cross, pool=pool, perm=perm, n=n, m=m, p=p)


def plain_pieces(steps, window):
    """a pair as it comes: value steps, then its window of key steps.  Pieces are (count, key form?, position of the first step)."""
    w = min(window, steps)
    out = []
    if steps - w > 0: out.append((steps - w, False, 0))
    if w > 0: out.append((w, True, steps - w))
    return out


def burst_pieces(steps, window, give_up, back_to, key_until):
    """a pair that gives up on a value step at `give_up`, goes back to its checkpoint at `back_to` and runs key steps until `key_until` (its
    probation; = steps: for good), then value steps and its window"""
    out = [(give_up, False, 0)]
    key_until = min(max(key_until, give_up), steps)
    out.append((key_until - back_to, True, back_to))
    tail = steps - key_until
    w = min(window, tail)
    if tail - w > 0: out.append((tail - w, False, key_until))
    if w > 0: out.append((w, True, steps - w))
    return out


def cut(pieces, stop_at):
    """the first part of a pair (until its position reaches stop_at for the first time) and its rest"""
    first, rest, done = [], [], False
    for cnt, key, pos in pieces:
        if done: rest.append((cnt, key, pos)); continue
        if pos + cnt >= stop_at > pos:
            a = stop_at - pos
            if a > 0: first.append((a, key, pos))
            if cnt - a > 0: rest.append((cnt - a, key, pos + a))
            done = True
        else: first.append((cnt, key, pos))
    return first, rest


def simulate(S, pieces=None, window=60, start_us=None, value_us=VALUE_US, key_us=KEY_US, groups_per_wave=4, lone_speedup=1.48, num_cus=256, log=None):
    """Run a launch on schedule S.  pieces[j]: what pair j executes (default: plain_pieces of its scheduled steps).  start_us[w]: when wave w
    becomes resident.  lone_speedup: a wave whose SIMD partner (the same wave slot of the CU's other workgroup) has ended issues alone -- at 74 %
    of the SIMD's rate instead of half of it (align16_step_blocks.inc, the issue-priority turns).  Returns per-wave end times (us), the steps
    executed per pair, and counters."""
    T, cum, rest, cross, p, m, n = S["T"], S["cum"], S["rest"], S["cross"], S["p"], S["m"], S["n"]
    if pieces is None: pieces = {}
    piece_of = lambda j: pieces.get(j) or plain_pieces(int(p[j]), window)
    nw = -(-m // groups_per_wave)
    state = np.full(m + 1, FRESH)                 # of the pair across boundary b (its first part belongs to interval b)
    saved_rest = {}                               # b -> pieces left when its first part was suspended
    pool = list(S["pool"])
    head = [0]
    executed = np.zeros(n, np.int64)              # steps executed per pair (tests: exactly its pieces' total)
    counters = dict(taken_whole=0, waits=0, own_rest=0, draws=0)

    class Group:
        __slots__ = ("interval", "queue", "cur", "waiting", "exhausted", "hold", "fixed_done", "steps")
    groups = []
    for ph in range(m):
        g = Group(); g.interval = int(S["perm"][ph]); g.queue = None; g.cur = []; g.waiting = None; g.exhausted = False; g.hold = None; g.fixed_done = False; g.steps = 0
        groups.append(g)

    def fixed_part(iv):
        """segments of interval iv in order: ('early', b=iv) first part of the pair across its lower boundary, then whole pairs"""
        lo, hi = iv * T, (iv + 1) * T
        segs = []
        j = int(np.searchsorted(cum[1:], lo, side="right"))
        while j < n and cum[j] < hi:
            c, c1 = int(cum[j]), int(cum[j + 1])
            if c1 > c:
                if c < lo: segs.append(("early", j, c1 - lo))
                elif c1 <= hi: segs.append(("whole", j, 0))
                # (the pair that crosses out: its rest is in the pool)
            j += 1
        return segs

    for g in groups: g.queue = fixed_part(g.interval)

    def next_segment(g, now):
        """give lane group g something to run, or mark it waiting / exhausted"""
        while True:
            if g.queue:
                kind, j, stop_at = g.queue.pop(0)
                if kind == "whole":
                    g.cur = [(c, k, pos, j) for c, k, pos in piece_of(j)]; g.hold = None
                    return
                b = g.interval
                if state[b] != FRESH: continue                     # taken whole by somebody else: skip it
                state[b] = RUNNING
                first, rst = cut(piece_of(j), stop_at)
                g.cur = [(c, k, pos, j) for c, k, pos in first]; g.hold = ("early", b, rst)
                if not g.cur: finish(g, now); continue
                return
            # the pool
            if g.waiting is None:
                if head[0] >= len(pool): g.exhausted = True; return
                b = pool[head[0]]; head[0] += 1; counters["draws"] += 1
                if b == g.interval + 1: counters["own_rest"] += 1
            else: b = g.waiting
            j = int(cross[b])
            if state[b] == FRESH:
                state[b] = STOLEN; counters["taken_whole"] += 1
                g.cur = [(c, k, pos, j) for c, k, pos in piece_of(j)]; g.hold = None; g.waiting = None
                return
            if state[b] == RUNNING:
                if g.waiting is None: counters["waits"] += 1
                g.waiting = b; return
            g.waiting = None
            if state[b] == DONE: continue
            g.cur = [(c, k, pos, j) for c, k, pos in saved_rest.pop(b)]; g.hold = None
            if not g.cur: continue
            return

    def finish(g, now):
        if g.hold is not None:
            _, b, rst = g.hold
            if rst: saved_rest[b] = rst; state[b] = SAVED
            else: state[b] = DONE
            g.hold = None

    t_wave = np.zeros(nw) if start_us is None else np.asarray(start_us, float).copy()
    heap = [(float(t_wave[w]), w) for w in range(nw)]
    heapq.heapify(heap)
    ended = np.zeros(nw)
    is_over = np.zeros(nw, bool)
    wpb = max(1, (m // (2 * num_cus)) // groups_per_wave) if m >= 2 * num_cus * groups_per_wave else 0     # waves per workgroup
    def partner(w):
        if not wpb or m != 2 * num_cus * wpb * groups_per_wave: return -1
        blk, wv = divmod(w, wpb)
        return (blk + num_cus if blk < num_cus else blk - num_cus) * wpb + wv
    poll = 32
    while heap:
        now, w = heapq.heappop(heap)
        gs = groups[w * groups_per_wave:(w + 1) * groups_per_wave]
        for g in gs:
            if not g.cur and not g.exhausted: next_segment(g, now)
        active = [g for g in gs if g.cur]
        if not active:
            if all(g.exhausted for g in gs): ended[w] = now; is_over[w] = True; continue
            heapq.heappush(heap, (now + poll * value_us * 0.25, w)); continue          # (everybody waits: the wave sleeps and looks again)
        k = min(g.cur[0][0] for g in active)
        if any(g.waiting is not None for g in gs): k = min(k, poll)                    # a group that waits looks again every few steps
        key = any(g.cur[0][1] for g in active)
        pw = partner(w)
        now += k * (key_us if key else value_us) / (lone_speedup if pw >= 0 and is_over[pw] else 1.0)
        for g in active:
            c, kf, pos, j = g.cur[0]
            executed[j] += k; g.steps += k
            if c == k:
                g.cur.pop(0)
                if not g.cur: finish(g, now)
            else: g.cur[0] = (c - k, kf, pos + k, j)
        if log is not None: log.append((w, now, k, key))
        heapq.heappush(heap, (now, w))
    return ended, executed, counters


def c1_like(n=10000, seed=3, sw=3):
    """step counts of a C1-like batch (10 kb +- 1 kb, 8-12 kb; the read 1 % shorter), sorted as the kernel sorts them: longest first"""
    rng = np.random.default_rng(seed)
    L = np.clip(np.rint(rng.normal(10000, 1000, n)), 8000, 12000).astype(np.int64)
    tot = (L + 7) // 8 + ((L * 99) // 100 + 7) // 8 - 1
    p = -(-tot // sw) * sw + 2 + PAIR_OVERHEAD
    return -np.sort(-np.asarray(p))


def main():
    m = 8192
    p = c1_like()
    S = build(p, m)
    T, cum = S["T"], S["cum"]
    base, _, c0 = simulate(S, window=170)             # (the reference's scoring: ~170 key steps at a pair's end)
    cross_of = {int(S["cross"][b]): b for b in range(1, m) if S["cross"][b] >= 0}
    print(f"C1-like, 10 000 pairs on {m} lane groups: T = {T} steps; {len(cross_of)} pairs lie across a boundary -- their last steps are somebody's last steps: they END AT T by construction")
    print(f"clean launch: last wave {base.max() / 1e3:.2f} ms, mean wave {base.mean() / 1e3:.2f} ms (the chip: 26.6 / 25.3), own rest drawn {c0['own_rest']} of {c0['draws']}, waits {c0['waits']}")
    bursty = np.nonzero(np.random.default_rng(5).random(len(p)) < 0.10)[0]

    def run(share, on, back=None, dip=90, sel=None):
        r2 = np.random.default_rng(7)
        pieces = {}
        for j in bursty:
            if r2.random() >= share: continue
            steps = int(p[j])
            g = int(r2.integers(steps // 5, steps * 4 // 5)) + 20
            c = (g // 256) * 256 - (256 if r2.random() < 0.5 else 0) if back is None else g - back        # the older / the newer checkpoint, span 256
            if c < 256 and back is None: c = max(g - 300, 1)
            if sel is not None and not sel(int(j), g): continue
            pieces[int(j)] = burst_pieces(steps, 170, g, max(c, 1), g + 33 + dip if on else steps)
        end, executed, _ = simulate(S, pieces=pieces, window=170)
        return len(pieces), 100 * (end.max() / base.max() - 1), 100 * (end.mean() / base.mean() - 1), 100 * (executed.sum() / p.sum() - 1)

    print("\na burst of errors in every tenth read: the pair gives up ~20 steps into the dip, goes back to the older / newer checkpoint (span 256), runs key steps")
    print("for good (probation off) or until the dip is behind it (on).  Share of such reads that go back: the chip's counters (54 and 887 of ~1 000).")
    print(f"{'':36s} {'pairs':>5s} {'steps':>8s} {'mean wave':>10s} {'LAST WAVE':>10s}    the chip's kernel (profiles/r05_v2/probation.txt)")
    chip = {(250, 0): "+ 19 %", (250, 1): "+ 10 %", (350, 0): "+ 26 %", (350, 1): "+ 24 %"}
    for burst, dip, share in ((250, 60, 0.054), (350, 90, 0.9)):
        for on in (0, 1):
            k, tail, mean, extra = run(share, on, dip=dip)
            print(f"  {burst} bases, probation {'on ' if on else 'off'}              {k:5d} {extra:+7.1f} % {mean:+8.1f} % {tail:+8.1f} %    {chip[(burst, on)]}")
    print("\nwhere the delay falls does not matter -- the pool cannot give back what a chain of steps has lost (350 bases, probation on):")
    in_rest = lambda j, g: j in cross_of and g >= int(cum[j + 1] - cross_of[j] * T)
    for name, sel in (("give-up in the pair's rest (its last steps, behind the boundary)", in_rest), ("give-up before that (first part, whole pair)", lambda j, g: not in_rest(j, g))):
        k, tail, mean, extra = run(0.9, 1, sel=sel)
        print(f"  {name:66s} {k:5d} pairs, mean wave {mean:+5.1f} %, last wave {tail:+6.1f} %")
    print("\n... the length of the way back does (350 bases, probation on):")
    for back in (300, 150, 50):
        k, tail, mean, extra = run(0.9, 1, back=back)
        print(f"  {back:3d} steps back: mean wave {mean:+5.1f} %, last wave {tail:+6.1f} %")


if __name__ == "__main__":
    sys.exit(main())
