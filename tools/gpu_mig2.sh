cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-m}
for nm in 0 1; do
AGATHA_AMD_NO_MIGRATE=$nm timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_m${nm}_$TAG.json 2> gpurun_out/bench_m${nm}_$TAG.err
python3 -c "
import json; b=json.load(open('gpurun_out/bench_m${nm}_$TAG.json')); print('no_migrate=$nm GCUPS',round(b['value'],1),'kernel_ms',round(b['kernel_ms'],2),b['config'].get('preemptive_schedule_rank0'))"
AGATHA_AMD_NO_MIGRATE=$nm rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_m${nm}_$TAG -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_m${nm}_$TAG.err
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_m${nm}_$TAG/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "align16_kernel<16, 3, -1>" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in agg.items()}
print({k:"%.3e"%v for k,v in m.items()})
if "SQ_INSTS_VALU" in m: print("VALU lane-ops/cell %.2f"%(m["SQ_INSTS_VALU"]*64/1.4366e11))
PY
done
