cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in ${CFGS:-C0 C2}; do for mode in "0 -1" "1 -1" "1 15" "1 0"; do set -- $mode
AGATHA_AMD_NO_MIGRATE=$1 AGATHA_AMD_PRIO_SLICE=$2 timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$c no_migrate=$1 prio=$2 kernel_ms',round(b['kernel_ms'],2),'kernel GCUPS',round(b['kernel_gcups_rank0'],1),'step GCUPS',round(b['value'],1),b['config']['kernel'],b['config'].get('preemptive_schedule_rank0'))"
done; done
