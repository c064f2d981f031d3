cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05p_ab.txt
: > $O
for rep in 1 2; do
  for lib in "" $GRAFT_REPO_ROOT/ab_r4.so; do
    echo "== lib=$(basename ${lib:-tree}) rep $rep" >> $O
    AGATHA_AMD_LIB=$lib timeout 200 python tools/opt_sweep.py cfg_c1 10000 "" 2>&1 | cut -c1-170 >> $O
    if [ -z "$lib" ]; then
      timeout 200 python tools/opt_sweep.py cfg_c1 10000 "no_pool=1,mig_identity=1,flat_detect=0" "flat_detect=0" "no_pool=1" "mig_identity=1,no_pool=1" 2>&1 | cut -c1-170 >> $O
    fi
    AGATHA_AMD_LIB=$lib timeout 200 python tools/opt_sweep.py cfg_c0 20000 "" 2>&1 | cut -c1-170 >> $O
    AGATHA_AMD_LIB=$lib timeout 200 python tools/opt_sweep.py cfg_c2 12500 "" 2>&1 | cut -c1-170 >> $O
  done
done
cat $O
timeout 900 python -m pytest tests/test_gpu_int16.py tests/test_gpu_traceback.py tests/test_gpu_cli.py tests/test_gpu_parity.py -m gpu -q --timeout 400 > gpurun_out/pytest_r05p.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05p.log
tail -3 gpurun_out/pytest_r05p.log
