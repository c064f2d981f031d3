cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/opt_sweep.py cfg_c1 10000 "" > /dev/null 2>&1 || { echo "SMOKE FAILED"; exit 1; }
timeout 600 python -m pytest tests/test_gpu_ref_scoring.py -m gpu -q --timeout 300 > gpurun_out/pytest_r05x.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05x.log
tail -12 gpurun_out/pytest_r05x.log
timeout 600 python tools/gpu_cliff_cells.py "C0 m1x4q6r2 0.10 1.0 0.0" "C1 m1x4q6r2 0.10 1.0 0.0" "C1 m1x9q16r2 0.05 1.0 0.02" "C0 m1x9q16r2 0.05 1.0 0.0" "C1 m1x4q6r2 0.15 1.0 0.0" "C0 m2x8q12r2 0.15 0.9 0.0" > gpurun_out/r05x_cells.txt 2>&1; cut -c1-300 gpurun_out/r05x_cells.txt
