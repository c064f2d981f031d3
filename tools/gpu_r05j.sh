cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05j.txt
: > $O
timeout 120 python tools/opt_sweep.py cfg_c1 10000 "" >> $O 2>&1 || { echo "SMOKE FAILED rc=$?" >> $O; cat $O; exit 1; }
timeout 600 python -m pytest tests/test_gpu_int16.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_r05j.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05j.log
tail -3 gpurun_out/pytest_r05j.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r05j -o p -- python3 $GRAFT_REPO_ROOT/tools/opt_sweep.py cfg_c1 10000 "" > $GRAFT_REPO_ROOT/gpurun_out/prof_r05j.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_r05j -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-160
cat $O
