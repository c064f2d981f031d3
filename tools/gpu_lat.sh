# latency-regime check: int16 tests, C1 bench line, C3 / C4 / one-round kernel-only (run through gpurun)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_int16.py tests/test_gpu_configs.py -x -q 2>&1 | tail -2
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('C1 GCUPS',round(b['value'],1),'kernel_ms',round(b['kernel_ms'],2))"
for c in C3 C4 C0 C2; do timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$c GCUPS',round(b['value'],1),'kernel_ms',round(b['kernel_ms'],2),b['config']['kernel'])"; done
timeout 200 python bench.py --pairs 8192 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('8192 pairs kernel_ms',round(b['kernel_ms'],2))"
