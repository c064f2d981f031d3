"""Developer tool (CPU): the two step loops of one int16 instantiation in the ISA listing tools/isa16_one.sh leaves in /tmp/isa16/one.s: for each
loop (the innermost loops that hold the three 500-instruction block pairs) the basic blocks in layout order with their VALU / LDS / SALU / memory
counts, and the totals of the blocks on the straight path (reached by falling through or by a forward branch over at most a few blocks; the
out-of-line blocks that branch BACK into the loop are listed apart: they are the rarely taken paths).   python3 tools/isa16_loops.py [one.s]"""
import re, sys, collections
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa16/one.s"
out, skip = [], False
for l in open(path).read().split("\n"):
    t = l.strip()
    if t.startswith(".if "): skip = not eval(t[4:]); continue
    if t == ".endif": skip = False; continue
    if not skip: out.append(l)
s = "\n".join(out)
i = s.index("_ZN6agatha14align16_kernel"); i = s.index(":\n", i)
fn = s[i:]; fn = fn[:fn.index(".Lfunc_end")]
blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", fn)
names = [b.split(":")[0] for b in blocks]
idx = {n: k for k, n in enumerate(names)}
info = []
for b in blocks:
    lines = b.split("\n")
    body = [l.strip() for l in lines[1:] if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    c = collections.Counter(x.split()[0] for x in body)
    brs = [(x.split()[0], idx.get(x.split()[-1])) for x in body if x.startswith(("s_cbranch", "s_branch"))]
    hdr = lines[0]
    info.append(dict(n=len(body), valu=sum(v for k, v in c.items() if k.startswith("v_")), lds=sum(v for k, v in c.items() if k.startswith("ds_")),
                     salu=sum(v for k, v in c.items() if k.startswith("s_")), mem=sum(v for k, v in c.items() if k.startswith(("global_", "scratch_", "flat_", "buffer_"))),
                     rl=c["v_readlane_b32"] + c["v_writelane_b32"], brs=brs, loop=re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", hdr)))
big = [k for k, d in enumerate(info) if d["valu"] >= 500]          # the block pairs
# group the block pairs by loop header
groups = collections.defaultdict(list)
for k in big:
    m = info[k]["loop"]
    groups[m.group(1) if m else "?"].append(k)
for hdr, ks in groups.items():
    lo = idx.get("." + "L" + hdr, min(ks))
    # the loop's straight part: from its header to the last block that branches back to the header region before the out-of-line blocks start
    hi = max(ks)
    while hi + 1 < len(info) and not (info[hi]["brs"] and any(t is not None and t <= lo + 1 for _, t in info[hi]["brs"])): hi += 1
    tot = collections.Counter()
    print("loop", hdr, "blocks", lo, "..", hi, "(block pairs at", ks, ")")
    for k in range(lo, hi + 1):
        d = info[k]
        for f in ("valu", "lds", "salu", "mem", "rl"): tot[f] += d[f]
    print("   straight part: VALU %d (of it v_readlane/v_writelane %d)  LDS %d  SALU %d  memory %d" % (tot["valu"], tot["rl"], tot["lds"], tot["salu"], tot["mem"]))
    seg, last = [], lo
    for k in ks + [hi + 1]:
        seg.append((last, k - 1)); last = k + 1
    for (a, b_), label in zip(seg, ["before pair %d" % (len(ks) - 1 - j) for j in range(len(ks))] + ["after the last pair"]):
        print("   %-20s VALU %4d  LDS %3d  SALU %3d  memory %2d   (blocks %d..%d)" % (label, sum(info[k]["valu"] for k in range(a, b_ + 1)), sum(info[k]["lds"] for k in range(a, b_ + 1)),
              sum(info[k]["salu"] for k in range(a, b_ + 1)), sum(info[k]["mem"] for k in range(a, b_ + 1)), a, b_))
    print("   block pairs: VALU", [info[k]["valu"] for k in ks])
