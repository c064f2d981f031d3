# A/B of two builds of the HIP library in one call (same box): ./ab_old.so against the tree's, C1 and C0, alternating
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" "$GRAFT_REPO_ROOT/ab_old.so"; do
for c in C1 C0; do
AGATHA_AMD_LIB=$lib timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-gasal-api 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('lib=${lib:-tree} $c kernel_ms',round(b['kernel_ms'],3))"
done; done; done
