# C1 with and without N runs in 2 % of the DP-row sequences (kernel time, how the pairs were routed)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for f in 0 0.02 0.10; do
timeout 300 python bench.py --n-run-frac $f --steps 5 --warmup 1 --no-cpu-baseline --no-gasal-api 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('n-run-frac $f kernel_ms',round(b['kernel_ms'],3),'GCUPS',round(b['value'],1),b['config']['pairs_plain_other_letters_int32_takeover_rank0'],b['config'].get('int16_steps_rank0'))"
done
