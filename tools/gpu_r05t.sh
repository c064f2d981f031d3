cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/opt_sweep.py cfg_c1 10000 "" > gpurun_out/r05t.txt 2>&1 || { echo "SMOKE FAILED"; cat gpurun_out/r05t.txt; exit 1; }
timeout 200 python tools/opt_sweep.py cfg_c2 12500 "" >> gpurun_out/r05t.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_int16.py tests/test_gpu_configs.py -m gpu -q -x --timeout 300 -k "migrat or schedule or take_over or never_started or c1_full or c2_at or static or pool or clean_up" > gpurun_out/pytest_r05t.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05t.log
tail -3 gpurun_out/pytest_r05t.log
for rep in 1 2; do for lib in "" $GRAFT_REPO_ROOT/ab_r4.so; do
  echo "lib=$(basename ${lib:-tree})" >> gpurun_out/r05t.txt
  AGATHA_AMD_LIB=$lib timeout 300 python tools/one_config.py C3 1024 2>&1 | tail -2 | cut -c1-120 >> gpurun_out/r05t.txt
done; done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r05t -o p -- python3 $GRAFT_REPO_ROOT/tools/opt_sweep.py cfg_c1 10000 "" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_r05t -name "*kernel_stats.csv" | head -1); head -7 $f | cut -c1-140 >> gpurun_out/r05t.txt
cat gpurun_out/r05t.txt
