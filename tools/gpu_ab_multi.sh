# A/B of several builds of the HIP library in one call: ./ab_*.so against the tree's, C1 and C0
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" $(ls $GRAFT_REPO_ROOT/ab_*.so); do
for c in ${CONFIGS:-C1 C0}; do
AGATHA_AMD_LIB=$lib timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-gasal-api 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('lib=$(basename ${lib:-tree}) $c kernel_ms',round(b['kernel_ms'],3))"
done; done; done
