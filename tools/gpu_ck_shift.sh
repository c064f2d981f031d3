# checkpoint spacing of the int16 kernel (debug option ck_shift): broken-read batches and HBM bytes per launch (run through gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gasal-api"
for s in 28 29 30; do
  export AGATHA_AMD_CK_SHIFT=$s
  echo "== ck_shift $s"
  timeout 300 python3 tools/gpu_skew.py 10000 2>&1 | cut -c1-150
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ck${s}_f -o pmc -- $BENCH > /dev/null 2> gpurun_out/ck${s}_f.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/ck${s}_w -o pmc -- $BENCH > gpurun_out/ck${s}_bench.json 2> gpurun_out/ck${s}_w.err
  python3 - <<PY
import csv, glob
for tag in ("f", "w"):
    for f in glob.glob("gpurun_out/ck${s}_%s/**/*counter_collection.csv" % tag, recursive=True):
        tot = {}
        for r in csv.DictReader(open(f)):
            if "align16_kernel" in r["Kernel_Name"]:
                tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in tot.items(): print("ck_shift ${s}", k, "launches", len(v), "mean per launch", sum(v) / len(v))
PY
  tail -c 400 gpurun_out/ck${s}_bench.json
done
