cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/opt_sweep.py cfg_c1 10000 "" "cleanup_min_steps=0" > gpurun_out/r05r_smoke.txt 2>&1 || { echo "SMOKE FAILED"; cat gpurun_out/r05r_smoke.txt; exit 1; }
cat gpurun_out/r05r_smoke.txt
timeout 600 python -m pytest tests/test_gpu_int16.py -m gpu -q -x --timeout 300 -k "clean_up or migrat or broken or checkpoint or take_over or never_started" > gpurun_out/pytest_r05r.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05r.log
tail -5 gpurun_out/pytest_r05r.log
O=gpurun_out/r05r_skew_cleanup.txt
: > $O
for cm in 384 0; do
  echo "== cleanup_min_steps=$cm" >> $O
  AGATHA_AMD_CLEANUP_MIN_STEPS=$cm timeout 300 python tools/gpu_skew.py 10000 2>&1 | grep -E "equal|broken" | cut -c1-150 >> $O
  AGATHA_AMD_CLEANUP_MIN_STEPS=$cm TIMELINE=1 timeout 300 python tools/gpu_cliff_cells.py "C1 m1x4q6r2 0.15 1.0 0.02" "C0 m1x4q6r2 0.15 1.0 0.02" 2>&1 | grep -E "^C|waves|five" | tail -8 | cut -c1-260 >> $O
done
cat $O
