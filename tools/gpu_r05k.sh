cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python tools/gpu_cliff_cells.py "C0 m1x4q6r2 0.15 0.9 0.0" "C0 m1x4q6r2 0.15 1.0 0.02" "C1 m1x4q6r2 0.15 1.0 0.02" "C1 m1x4q6r2 0.15 1.0 0.0" "C0 m1x4q6r2 0.15 0.75 0.02" "C2 m1x4q6r2 0.15 1.0 0.02" "C0 m1x9q16r2 0.05 1.0 0.0" "C0 m2x8q12r2 0.15 0.9 0.0" "C1 m1x9q16r2 0.05 1.0 0.02" "C1 m2x8q12r2 0.15 1.0 0.0" "C1 m1x4q6r2 0.10 1.0 0.0" "C1 m2x4q4r2 0.10 1.0 0.0" "C0 m1x4q6r2 0.10 1.0 0.0" > gpurun_out/r05k_cliff_cells.txt 2>&1
cat gpurun_out/r05k_cliff_cells.txt
