cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/opt_sweep.py cfg_c1 10000 "" > /dev/null 2>&1 || { echo "SMOKE FAILED"; exit 1; }
timeout 500 python tools/gpu_fuzz.py 300 77 > gpurun_out/fuzz_r05u.txt 2>&1; tail -2 gpurun_out/fuzz_r05u.txt
timeout 500 python tools/gpu_fuzz_mig.py 300 78 > gpurun_out/fuzz_mig_r05u.txt 2>&1; tail -2 gpurun_out/fuzz_mig_r05u.txt
timeout 300 python tools/gpu_fuzz_split.py 120 79 > gpurun_out/fuzz_split_r05u.txt 2>&1; tail -2 gpurun_out/fuzz_split_r05u.txt
AGATHA_AMD_FLAT_PERCENT=1 AGATHA_AMD_CLEANUP_MIN_STEPS=1 timeout 400 python tools/gpu_fuzz_mig.py 200 80 > gpurun_out/fuzz_mig_flat_r05u.txt 2>&1; tail -2 gpurun_out/fuzz_mig_flat_r05u.txt
