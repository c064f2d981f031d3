# traceback wall time (tools/gpu_tb.py) for every ./ab_*.so and the tree's library, in one call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=${1:-6000}
for rep in 1 2; do
for lib in "" $(ls $GRAFT_REPO_ROOT/ab_*.so 2>/dev/null); do
echo "lib=$(basename ${lib:-tree}) $(AGATHA_AMD_LIB=$lib timeout 300 python3 tools/gpu_tb.py $N 2>&1 | tail -1)"
done; done
