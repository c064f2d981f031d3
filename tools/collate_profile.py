"""Developer tool: turn the rocprofv3 outputs of tools/gpu_profile.sh (gpurun_out/prof_<tag>_*) into the committed
summaries under profiles/<dest>/ and refresh profiles/latest_pmc.json (HBM bytes per launch of the dominant kernel).

    python tools/collate_profile.py <tag> <dest> [pairs]
"""
import csv, glob, collections, json, os, shutil, sys

tag, dest = sys.argv[1], sys.argv[2]
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles", dest)
os.makedirs(out, exist_ok=True)
src = os.path.join(root, "gpurun_out")

stats = os.path.join(src, f"prof_{tag}_trace", "trace_kernel_stats.csv")
shutil.copy(stats, os.path.join(out, "kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
dom = max(rows, key=lambda r: float(r["TotalDurationNs"]))
kname = dom["Name"]
short = kname.split("(")[0].replace("void ", "")

per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(src, f"prof_{tag}_pmc*"))):
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "align" in r["Kernel_Name"]:
                per_kernel[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in per_kernel.items()}
json.dump({"source": f"rocprofv3 --pmc, separate passes, python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline ({pairs} pairs); "
                     "mean per launch", "kernels": summary}, open(os.path.join(out, "pmc_mean_per_launch.json"), "w"), indent=1)
b = os.path.join(src, f"prof_{tag}_trace.json")
if os.path.exists(b):
    shutil.copy(b, os.path.join(out, "bench_under_rocprof.json"))
m = summary.get(short, {})
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    json.dump({"source": f"profiles/{dest} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 2 --warmup 1, {pairs} pairs)",
               "kernel": short, "pairs": pairs, "config": "C1", "scoring": "m2x4q4r2",
               "preemptive_schedule": bool(json.load(open(b)).get("config", {}).get("preemptive_schedule_rank0", {}).get("used", False)) if os.path.exists(b) else False,
               "hbm_bytes_per_launch": (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024,
               "fetch_size_kb": m["FETCH_SIZE"], "write_size_kb": m["WRITE_SIZE"],
               "valu_insts_per_launch": m.get("SQ_INSTS_VALU"), "lds_bank_conflict_cycles": m.get("SQ_LDS_BANK_CONFLICT"),
               "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half of a streamed read"},
              open(os.path.join(root, "profiles", "latest_pmc.json"), "w"), indent=1)
print("dominant kernel:", short, "avg ms", float(dom["AverageNs"]) / 1e6)
for k, v in summary.items():
    print(k, {c: "%.3g" % x for c, x in v.items()})
