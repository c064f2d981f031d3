# A/B of the value-step path of the int16 kernel: C1 / C0 / C2 kernel time with fast_margin = 0 (key steps only) and the default
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for fm in 0 16 ${FAST_MARGINS:-}; do
for c in ${CONFIGS:-C1 C0 C2}; do
AGATHA_AMD_FAST_MARGIN=$fm timeout 300 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('margin $fm $c kernel_ms',round(b['kernel_ms'],2),'kernel GCUPS',round(b['kernel_gcups_rank0'],1),b['config']['kernel'],b['config'].get('preemptive_schedule_rank0',{}).get('used'),b['config'].get('int16_steps_rank0'),b['config']['pairs_plain_other_letters_int32_takeover_rank0'])"
done
done
for nm in 1; do
AGATHA_AMD_FAST_MARGIN=0 AGATHA_AMD_NO_MIGRATE=1 timeout 300 python bench.py --config C1 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('margin 0 no_migrate C1 kernel_ms',round(b['kernel_ms'],2),b['config'].get('int16_steps_rank0'))"
done
