"""Developer tool (GPU box): summary of a rocprofv3 PC-sampling run (--pc-sampling-beta-enabled, csv output): samples per instruction
of the align16 kernel, by opcode class and by code-object offset (bins of 256 bytes), so that the hot regions of a step can be matched
against the ISA listing of tools/isa16_one.sh.   python3 tools/pcs_summary.py <dir>"""
import collections, csv, glob, os, sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pcsample/pcs"
files = [f for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)]
print("csv files:", [os.path.relpath(f, d) + " (%d B)" % os.path.getsize(f) for f in files])
for f in files:
    if "pc_sampl" not in os.path.basename(f).lower():
        continue
    with open(f, newline="") as fh:
        rd = csv.reader(fh)
        hdr = next(rd, None)
        print(os.path.basename(f), "columns:", hdr)
        if hdr is None:
            continue
        col = {c.lower(): i for i, c in enumerate(hdr)}
        ins_i = next((i for c, i in col.items() if "instruction" == c or c == "instruction_comment" or c.startswith("instruction")), None)
        off_i = next((i for c, i in col.items() if "offset" in c), None)
        n = 0
        by_op, by_bin, by_ins = collections.Counter(), collections.Counter(), collections.Counter()
        first = []
        for row in rd:
            n += 1
            if len(first) < 5:
                first.append(row)
            ins = row[ins_i] if ins_i is not None and ins_i < len(row) else ""
            op = ins.split()[0] if ins else "?"
            by_op[op] += 1
            if off_i is not None and off_i < len(row):
                try:
                    o = int(row[off_i], 0)
                    by_bin[o >> 8] += 1
                    by_ins[(o, ins)] += 1
                except ValueError:
                    pass
        print("samples", n)
        for r in first:
            print("  ", r)
        cls = collections.Counter()
        for op, c in by_op.items():
            k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "scratch_", "flat_", "buffer_")) else "other"
            cls[k] += c
        print("by class:", dict(cls))
        print("top opcodes:", by_op.most_common(25))
        print("top 256-byte bins (offset >> 8: samples):", sorted(by_bin.items(), key=lambda kv: -kv[1])[:60])
        with open(os.path.join(d, "pcs_by_offset.txt"), "w") as out:
            for (o, ins), c in sorted(by_ins.items()):
                out.write("%8d %6d %s\n" % (o, c, ins))
        print("per-instruction table ->", os.path.join(d, "pcs_by_offset.txt"))
