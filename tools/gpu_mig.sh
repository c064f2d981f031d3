# GPU check of the preemptive schedule: the migration tests, then the int16 suite, then C1 bench with and without migration
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-m}
timeout 300 python -m pytest tests/test_gpu_int16.py -x -q -k "migrat" > gpurun_out/pytest_mig_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_mig_$TAG.log
tail -3 gpurun_out/pytest_mig_$TAG.log
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mig_$TAG.json 2> gpurun_out/bench_mig_$TAG.err
AGATHA_AMD_NO_MIGRATE=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_nomig_$TAG.json 2> gpurun_out/bench_nomig_$TAG.err
python3 - <<PY
import json
for f in ("mig","nomig"):
    try:
        b=json.load(open("gpurun_out/bench_%s_$TAG.json"%f)); print(f,"GCUPS",round(b["value"],1),"kernel_ms",round(b["kernel_ms"],2),"checked",b.get("gpu_results_checked"))
    except Exception as e: print(f,"failed",e)
PY
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -3 gpurun_out/pytest_gpu_$TAG.log
