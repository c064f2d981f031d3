cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/opt_sweep.py cfg_c1 10000 "" > gpurun_out/r05n_smoke.txt 2>&1 || { echo "SMOKE FAILED"; cat gpurun_out/r05n_smoke.txt; exit 1; }
timeout 900 python tools/gpu_cliff_cells.py "C1 m1x4q6r2 0.15 1.0 0.02" "C1 m1x4q6r2 0.15 1.0 0.0" "C2 m1x4q6r2 0.15 1.0 0.02" "C1 m1x9q16r2 0.05 1.0 0.02" "C1 m2x8q12r2 0.15 1.0 0.0" "C0 m1x4q6r2 0.15 1.0 0.02" "C1 m1x4q6r2 0.10 1.0 0.0" > gpurun_out/r05n_cliff_cells.txt 2>&1
cat gpurun_out/r05n_cliff_cells.txt
timeout 900 python -m pytest tests/test_gpu_int16.py tests/test_gpu_ref_scoring.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_r05n.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05n.log
tail -3 gpurun_out/pytest_r05n.log
