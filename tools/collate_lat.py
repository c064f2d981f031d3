"""Developer tool: summaries of tools/gpu_profile_lat.sh (gpurun_out/prof_<tag>_<C3|C4>_*) -> profiles/<dest>/
    python tools/collate_lat.py <tag> <dest>"""
import csv, glob, collections, json, os, shutil, sys
tag, dest = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles", dest); os.makedirs(out, exist_ok=True)
src = os.path.join(root, "gpurun_out")
res = {}
for c in ("C3", "C4"):
    stats = os.path.join(src, f"prof_{tag}_{c}_trace", "trace_kernel_stats.csv")
    if not os.path.exists(stats): continue
    shutil.copy(stats, os.path.join(out, f"{c}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    dom = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    name = dom["Name"]
    agg = collections.defaultdict(list)
    for d in glob.glob(os.path.join(src, f"prof_{tag}_{c}_pmc*")):
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"] == name: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
    avg_ms = float(dom["AverageNs"]) / 1e6
    d = {"kernel": name.split("(")[0].replace("void ", ""), "avg_ms": avg_ms, "calls": int(dom["Calls"]), "pmc_mean_per_launch": m,
         "line": open(os.path.join(src, f"prof_{tag}_{c}_trace.txt")).read().strip().splitlines()[-1][:300]}
    if "SQ_INSTS_VALU" in m and "SQ_WAVE_CYCLES" in m:
        # (SQ_WAVE_CYCLES counts in units of 4 cycles, as one wave64 VALU instruction takes: the ratio is the share of a wave's life in which it issues VALU)
        d["valu_issue_share_of_wave_cycles"] = m["SQ_INSTS_VALU"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else None
    if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
        d["wait_share_of_wave_cycles"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else None
    res[c] = d
    print(c, d["kernel"], "avg ms", round(avg_ms, 2), {k: "%.3g" % v for k, v in m.items()})
json.dump(res, open(os.path.join(out, "latency_regimes.json"), "w"), indent=1)
