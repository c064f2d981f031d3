"""Developer tool: mean per-launch PMC counters of one kernel from a rocprofv3 counter_collection.csv."""
import csv, glob, collections, sys
pat, name = sys.argv[1], sys.argv[2]
ms = float(sys.argv[3]) if len(sys.argv) > 3 else None
agg = collections.defaultdict(list)
for f in glob.glob(pat):
    for r in csv.DictReader(open(f)):
        if name in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print({k: "%.3e" % v for k, v in sorted(m.items())})
cells = 1.367e11
if "SQ_INSTS_VALU" in m:
    print("VALU lane-ops/cell %.2f" % (m["SQ_INSTS_VALU"] * 64 / cells), "SALU/VALU %.2f" % (m.get("SQ_INSTS_SALU", 0) / m["SQ_INSTS_VALU"]))
    if ms:
        print("VALU issue fraction %.2f" % (m["SQ_INSTS_VALU"] * 4 / 1024 / (ms * 1e-3 * 2.37e9)))
