# checkpoint threshold sweep: C0 / C1 kernel time and the broken-pair batches of tools/gpu_skew.py
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for ck in 2048 1024 512; do
echo "ck_min_steps $ck"
for c in C1 C0; do AGATHA_AMD_CK_MIN_STEPS=$ck python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-gasal-api 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('  $c kernel_ms',round(b['kernel_ms'],3))"; done
AGATHA_AMD_CK_MIN_STEPS=$ck python3 tools/gpu_skew.py 10000 --timeline 2>&1 | tail -6 | cut -c1-330
done
