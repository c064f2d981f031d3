cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05c_timeline.txt
: > $O
for sc in 2,4,4,2 1,4,6,2; do
  for opt in "" "mig_identity=1" "fast_margin=0"; do
    echo "== C1 10000 scoring $sc $opt" >> $O
    SCORING=$sc timeout 300 python tools/timeline_steps.py 10000 cfg_c1 $opt >> $O 2>&1
  done
done
cat $O
