cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { timeout 300 python bench.py --n-run-frac $1 --steps 5 --warmup 1 --no-cpu-baseline --no-gasal-api 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); s=b['config'].get('int16_steps_rank0'); print('$2 n-run-frac $1 kernel_ms',round(b['kernel_ms'],3),s['value_wave_steps'],s['key_wave_steps'],s['pairs_started_over'],s['pairs_started'],s['debug'][:3])"; }
run 0 default; run 0.02 default; run 0.02 default
AGATHA_AMD_FAST_MARGIN=0 run 0 keyonly; AGATHA_AMD_FAST_MARGIN=0 run 0.02 keyonly
AGATHA_AMD_NO_MIGRATE=1 run 0 queue; AGATHA_AMD_NO_MIGRATE=1 run 0.02 queue
