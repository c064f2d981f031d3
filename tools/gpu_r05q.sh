cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05q_ab.txt
: > $O
timeout 120 python tools/opt_sweep.py cfg_c1 10000 "" > /dev/null 2>&1 || { echo "SMOKE FAILED"; exit 1; }
for rep in 1 2; do
  for lib in "" $GRAFT_REPO_ROOT/ab_r4.so; do
    echo "== lib=$(basename ${lib:-tree}) rep $rep" >> $O
    AGATHA_AMD_LIB=$lib timeout 200 python tools/opt_sweep.py cfg_c1 10000 "" 2>&1 | cut -c1-170 >> $O
    AGATHA_AMD_LIB=$lib timeout 200 python tools/opt_sweep.py cfg_c0 20000 "" 2>&1 | cut -c1-170 >> $O
    AGATHA_AMD_LIB=$lib timeout 200 python tools/opt_sweep.py cfg_c2 12500 "" 2>&1 | cut -c1-170 >> $O
  done
done
cat $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/pytest_r05q.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05q.log
tail -4 gpurun_out/pytest_r05q.log
