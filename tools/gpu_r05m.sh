cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/opt_sweep.py cfg_c1 10000 "" > gpurun_out/r05m_smoke.txt 2>&1 || { echo "SMOKE FAILED"; cat gpurun_out/r05m_smoke.txt; exit 1; }
cat gpurun_out/r05m_smoke.txt
timeout 900 python tools/gpu_cliff_cells.py "C0 m1x4q6r2 0.15 0.9 0.0" "C0 m1x4q6r2 0.15 1.0 0.02" "C1 m1x4q6r2 0.15 1.0 0.02" "C1 m1x4q6r2 0.15 1.0 0.0" "C0 m1x4q6r2 0.15 0.75 0.02" "C2 m1x4q6r2 0.15 1.0 0.02" "C0 m1x9q16r2 0.05 1.0 0.0" "C0 m2x8q12r2 0.15 0.9 0.0" "C1 m1x9q16r2 0.05 1.0 0.02" "C1 m2x8q12r2 0.15 1.0 0.0" "C1 m1x4q6r2 0.10 1.0 0.0" "C1 m2x4q4r2 0.10 1.0 0.0" "C0 m1x4q6r2 0.10 1.0 0.0" "C1 m1x19q39r3 0.01 1.0 0.0" > gpurun_out/r05m_cliff_cells.txt 2>&1
cat gpurun_out/r05m_cliff_cells.txt
for sc in 2,4,4,2 1,4,6,2; do
  SCORING=$sc timeout 200 python tools/opt_sweep.py cfg_c1 10000 "" >> gpurun_out/r05m_smoke.txt 2>&1
  SCORING=$sc timeout 200 python tools/opt_sweep.py cfg_c0 20000 "" >> gpurun_out/r05m_smoke.txt 2>&1
done
cat gpurun_out/r05m_smoke.txt
