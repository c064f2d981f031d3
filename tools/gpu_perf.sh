# quick perf + parity check of the int16 kernel: int16/migration tests, C1 bench (5 steps), C0/C2 kernel times
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-p}
timeout 900 python -m pytest tests/test_gpu_int16.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q > gpurun_out/pytest_perf_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_perf_$TAG.log
tail -3 gpurun_out/pytest_perf_$TAG.log
for c in C1 C0 C2; do
timeout 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$c kernel_ms',round(b['kernel_ms'],2),'kernel GCUPS',round(b['kernel_gcups_rank0'],1),'step GCUPS',round(b['value'],1),b['config']['kernel'],b['config'].get('preemptive_schedule_rank0'),b['config'].get('int16_steps_rank0'))"
done
