"""Developer tool: the pipeline leg of a bench.py JSON line, one row per run.  python tools/print_pipeline.py bench.json"""
import json, sys
b = json.load(open(sys.argv[1]))
p = b["gasal_api"]["pipeline"]
print("kernel_ms", round(b["kernel_ms"], 2), "value", round(b["value"], 1), b["unit"])
for r in p["runs"]:
    print("host threads", r["host_threads"], "host format", {False: "ASCII", True: "4-bit (-k)", 2: "2-bit + N mask (-K)"}[r["host_packed"]],
          "end to end", round(r["end_to_end_gcups"], 1), "GCUPS =", round(r["vs_kernel_only"], 3), "x kernel-only, loop", round(r["loop_s"], 3), "s")
print("pairs taken over with -p:", p["pairs_taken_over_with_p"])
