# round 4: parity tests + bench + one PMC pass (instruction counts); usage: bash tools/gpu_r04.sh TAG [pytest args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r04}; shift
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 "$@" > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gasal-api > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_$TAG -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gasal-api > /dev/null 2> gpurun_out/pmc_$TAG.err
tail -15 gpurun_out/pytest_gpu_$TAG.log
python3 - <<PY
import json,csv,glob,collections
b=json.load(open("gpurun_out/bench_$TAG.json")); print("GCUPS",round(b["value"],1),"kernel_ms",round(b["kernel_ms"],2),"kernel",b["config"]["kernel"], b["config"]["int16_steps_rank0"])
agg=collections.defaultdict(list)
kn=b["config"]["kernel"].replace("agatha::","")
base,args=kn.split("<"); args=args.rstrip(">").split(",")
for f in glob.glob("gpurun_out/pmc_$TAG/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if base+"<"+", ".join(args)+"," in n.replace("true","1") or base+"<"+", ".join(args)+">" in n:
            if ", true>" in n: continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in agg.items()}
cells=1.4366e11
print({k:"%.3e"%v for k,v in m.items()})
if "SQ_INSTS_VALU" in m: print("VALU lane-ops/cell %.2f"%(m["SQ_INSTS_VALU"]*64/cells), "SALU/VALU %.2f"%(m["SQ_INSTS_SALU"]/m["SQ_INSTS_VALU"]))
PY
