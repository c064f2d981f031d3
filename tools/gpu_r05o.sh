cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/opt_sweep.py cfg_c1 10000 "" > gpurun_out/r05o_smoke.txt 2>&1 || { echo "SMOKE FAILED"; cat gpurun_out/r05o_smoke.txt; exit 1; }
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 400 > gpurun_out/pytest_r05o.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05o.log
tail -4 gpurun_out/pytest_r05o.log
timeout 400 python tools/gpu_fuzz.py 200 51 > gpurun_out/fuzz_r05o.txt 2>&1; tail -2 gpurun_out/fuzz_r05o.txt
timeout 400 python tools/gpu_fuzz_mig.py 150 52 > gpurun_out/fuzz_mig_r05o.txt 2>&1; tail -2 gpurun_out/fuzz_mig_r05o.txt
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r05o.json 2> gpurun_out/bench_r05o.err; python -c "
import json; b=json.load(open('gpurun_out/bench_r05o.json')); print('C1 m2', round(b['value'],1), 'GCUPS kernel_ms', round(b['kernel_ms'],2), b['config']['int16_steps_rank0']); print('gasal_api ref cmd', b.get('gasal_api',{}).get('reference_bench_command')); print('pipeline', {k:v for k,v in b.get('gasal_api',{}).get('pipeline',{}).items() if k in ('best_end_to_end_gcups','best_config','best_host_packed_vs_kernel_only','steady_state')}); print('cpu', b.get('cpu_baseline',{}).get('value'), b.get('cpu_baseline',{}).get('gpu_results_checked'))"
timeout 900 python bench.py --steps 10 --warmup 2 --scoring m1x4q6r2 --no-pipeline > gpurun_out/bench_r05o_m1.json 2> gpurun_out/bench_r05o_m1.err; python -c "
import json; b=json.load(open('gpurun_out/bench_r05o_m1.json')); print('C1 m1', round(b['value'],1), 'GCUPS kernel_ms', round(b['kernel_ms'],2), b['config']['int16_steps_rank0']); print('cpu', b.get('cpu_baseline',{}).get('value'), b.get('cpu_baseline',{}).get('gpu_results_checked'))"
