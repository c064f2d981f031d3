# A/B of ./ab_*.so against the tree's library on C1 plain and with N runs in 2 % of the queries (kernel ms)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" $(ls $GRAFT_REPO_ROOT/ab_*.so); do
for nf in 0 0.02; do
AGATHA_AMD_LIB=$lib timeout 300 python bench.py --n-run-frac $nf --steps 6 --warmup 2 --no-cpu-baseline --no-gasal-api 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('lib=$(basename ${lib:-tree}) n-run-frac $nf kernel_ms',round(b['kernel_ms'],3))"
done; done; done
