# full GPU check: whole gpu suite, bench (3 steps) with and without the preemptive schedule, a batch-size sweep
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-f}
if [ -z "$NOTEST" ]; then timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -3 gpurun_out/pytest_gpu_$TAG.log; fi
timeout 300 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python3 -c "
import json; b=json.load(open('gpurun_out/bench_$TAG.json')); print('bench GCUPS',round(b['value'],1),'kernel_ms',round(b['kernel_ms'],2),b['config'].get('preemptive_schedule_rank0'), 'cpu', b.get('cpu_baseline',{}).get('value'), b.get('cpu_baseline',{}).get('gpu_results_checked'))"
for n in ${SIZES:-4500 8192 9000 11000 12288 13000 16384 20000 32768}; do
for nm in 0 1 2; do
AGATHA_AMD_PRIO_SLICE=$([ $nm = 2 ] && echo 15 || echo -1) AGATHA_AMD_NO_MIGRATE=$([ $nm = 0 ] && echo 0 || echo 1) timeout 200 python bench.py --pairs $n --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('pairs $n no_migrate=$nm kernel_ms',round(b['kernel_ms'],2),'kernel GCUPS',round(b['kernel_gcups_rank0'],1),b['config'].get('preemptive_schedule_rank0'))"
done; done
