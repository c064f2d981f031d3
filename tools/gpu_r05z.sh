cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05z_probation.txt
: > $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O 2>&1 || { cat $O; exit 1; }
echo "== C1, reference scoring: bursts of errors (probation on)" >> $O
SCORING=1,4,6,2 timeout 150 python tools/gpu_dips.py 10000 2>&1 | cut -c1-170 >> $O || { echo "TIMED OUT / FAILED" >> $O; cat $O; exit 1; }
echo "== tests" >> $O
timeout 700 python -m pytest tests/test_gpu_ref_scoring.py tests/test_gpu_int16.py tests/test_gpu_configs.py -x -q 2>&1 | tail -8 >> $O
timeout 200 python tools/gpu_fuzz_mig.py 120 11 2>&1 | tail -3 >> $O
echo "== C1, default scoring: broken reads / unequal lengths" >> $O
timeout 200 python tools/gpu_skew.py 10000 2>&1 | cut -c1-170 >> $O
cat $O
