cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r05i -o p -- python3 $GRAFT_REPO_ROOT/tools/opt_sweep.py cfg_c1 10000 "" > $GRAFT_REPO_ROOT/gpurun_out/prof_r05i.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_r05i -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-220 | head -30
tail -2 gpurun_out/prof_r05i.log
