# round 5, first look: the reference's bench scoring (AGAThA.sh:44: m1 x4 q6 r2) on C1 / C0 under the window options
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05a_ref_scoring.txt
: > $O
for sc in 2,4,4,2 1,4,6,2; do
  echo "== C1 10000 pairs, scoring $sc" >> $O
  SCORING=$sc timeout 600 python tools/opt_sweep.py cfg_c1 10000 "" "fast_margin=0" "fast_margin=50" "fast_margin=100" "fast_margin=200" "fast_margin=300" >> $O 2>&1
  echo "== C0 20000 pairs, scoring $sc" >> $O
  SCORING=$sc timeout 600 python tools/opt_sweep.py cfg_c0 20000 "" "fast_margin=0" "fast_margin=50" "fast_margin=100" "fast_margin=200" >> $O 2>&1
done
cat $O
