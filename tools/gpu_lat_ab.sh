# latency regimes (C3, C4, C3 at 1024 pairs, one round of C1) with key steps only and with value steps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for fm in 0 16; do
for c in "C3" "C4" "C3 1024"; do
AGATHA_AMD_FAST_MARGIN=$fm timeout 600 python3 tools/one_config.py $c 2>/dev/null | tail -1 | sed "s/^/margin $fm /"
done
AGATHA_AMD_FAST_MARGIN=$fm timeout 200 python bench.py --pairs 8192 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('margin $fm 8192 pairs kernel_ms',round(b['kernel_ms'],2),round(b['kernel_gcups_rank0'],1),b['config'].get('int16_steps_rank0'))"
done
