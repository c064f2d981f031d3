cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05y_probes_m1.txt
: > $O
echo "== C1, reference scoring m1 x4 q6 r2: unequal lengths and broken reads" >> $O
SCORING=1,4,6,2 timeout 400 python tools/gpu_skew.py 10000 2>&1 | cut -c1-170 >> $O
echo "== C1, reference scoring: bursts of errors" >> $O
SCORING=1,4,6,2 timeout 400 python tools/gpu_dips.py 10000 2>&1 | cut -c1-170 >> $O
echo "== C2 (HiFi, band 500), reference scoring" >> $O
SCORING=1,4,6,2 CFG=cfg_c2 BAND=500 timeout 500 python tools/gpu_skew.py 9000 2>&1 | cut -c1-170 >> $O
cat $O
