cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-c}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout 300 python bench.py --steps 5 --warmup 1 --scaling strong --no-cpu-baseline > gpurun_out/bench_strong1_$TAG.json 2> gpurun_out/bench_strong1_$TAG.err
for c in C0 C2 C3 C4; do timeout 600 python bench.py --config $c --steps 3 --warmup 1 > gpurun_out/bench_${c}_$TAG.json 2> gpurun_out/bench_${c}_$TAG.err; done
python3 - <<PY
import json
for f in ("strong1","C0","C2","C3","C4"):
    try:
        b=json.load(open("gpurun_out/bench_%s_$TAG.json"%f)); print(f,"GCUPS",round(b["value"],1),"kernel_ms",round(b["kernel_ms"],2),b["config"]["kernel"],b["scaling"],b["config"].get("preemptive_schedule_rank0"),"cpu",round(b.get("cpu_baseline",{}).get("value",0),1),b.get("cpu_baseline",{}).get("gpu_results_checked"))
    except Exception as e: print(f,"failed",e)
PY
