cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r05v2
timeout 120 python tools/opt_sweep.py cfg_c1 10000 "" > gpurun_out/r05s_smoke.txt 2>&1 || { echo "SMOKE FAILED"; cat gpurun_out/r05s_smoke.txt; exit 1; }
cat gpurun_out/r05s_smoke.txt
timeout 1200 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/pytest_r05s.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05s.log
tail -4 gpurun_out/pytest_r05s.log
for c in C2 C3 C4; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_${c}_trace -o trace -- python3 tools/one_config.py $c > gpurun_out/prof_${TAG}_${c}_trace.txt 2> gpurun_out/prof_${TAG}_${c}_trace.err
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/prof_${TAG}_${c}_pmc1 -o pmc -- python3 tools/one_config.py $c > /dev/null 2> gpurun_out/prof_${TAG}_${c}_pmc1.err
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/prof_${TAG}_${c}_pmc2 -o pmc -- python3 tools/one_config.py $c > /dev/null 2> gpurun_out/prof_${TAG}_${c}_pmc2.err
  tail -1 gpurun_out/prof_${TAG}_${c}_trace.txt | cut -c1-200
done
c=C0
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/prof_${TAG}_${c}_pmc2 -o pmc -- python3 tools/one_config.py $c > /dev/null 2> gpurun_out/prof_${TAG}_${c}_pmc2.err
timeout 600 python tools/gpu_cliff_cells.py "C1 m1x4q6r2 0.15 1.0 0.02" "C1 m1x4q6r2 0.15 1.0 0.02" "C1 m1x4q6r2 0.15 1.0 0.02" "C0 m1x4q6r2 0.15 1.0 0.02" > gpurun_out/r05s_flat_n_cells.txt 2>&1
cat gpurun_out/r05s_flat_n_cells.txt | cut -c1-250
