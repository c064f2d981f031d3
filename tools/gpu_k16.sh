# int16-kernel change check: its tests + the BASELINE-shape tests, then the bench line (run through gpurun)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_int16.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
python bench.py --steps 5 --warmup 1 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('bench GCUPS',round(b['value'],1),'kernel_ms',round(b['kernel_ms'],2), b['cpu_baseline'].get('gpu_results_checked'))"
