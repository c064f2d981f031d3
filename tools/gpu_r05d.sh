cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05d_wincap.txt
: > $O
echo "== C1 10000 pairs, scoring 2,4,4,2" >> $O
SCORING=2,4,4,2 timeout 600 python tools/opt_sweep.py cfg_c1 10000 "" >> $O 2>&1
echo "== C1 10000 pairs, scoring 1,4,6,2" >> $O
SCORING=1,4,6,2 timeout 600 python tools/opt_sweep.py cfg_c1 10000 "" "win_cap_div=8" "win_cap_div=16" "win_cap_div=24" "win_cap_div=32,win_cap_min=64" "mig_identity=1" >> $O 2>&1
echo "== C0 20000 pairs, scoring 1,4,6,2" >> $O
SCORING=1,4,6,2 timeout 600 python tools/opt_sweep.py cfg_c0 20000 "" "win_cap_min=64" "win_cap_min=128" "win_cap_min=160" >> $O 2>&1
echo "== timeline C1 m1 default" >> $O
SCORING=1,4,6,2 timeout 300 python tools/timeline_steps.py 10000 cfg_c1 >> $O 2>&1
cat $O
