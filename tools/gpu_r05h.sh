cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05h_skew_pool.txt
: > $O
for np_ in 0 1; do
  echo "== no_pool=$np_" >> $O
  AGATHA_AMD_NO_POOL=$np_ timeout 400 python tools/gpu_skew.py 10000 2>&1 | cut -c1-190 >> $O
done
cat $O
