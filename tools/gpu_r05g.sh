cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_int16.py tests/test_gpu_configs.py tests/test_gpu_ref_scoring.py -m gpu -q -x > gpurun_out/pytest_r05g.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05g.log
tail -6 gpurun_out/pytest_r05g.log
O=gpurun_out/r05g_pool.txt
: > $O
for sc in 2,4,4,2 1,4,6,2; do
  echo "== C1 10000 pairs, scoring $sc" >> $O
  SCORING=$sc timeout 600 python tools/opt_sweep.py cfg_c1 10000 "" "no_pool=1" "" "no_pool=1" "no_pool=1,mig_identity=1" >> $O 2>&1
  echo "== C2 12500 pairs, scoring $sc" >> $O
  SCORING=$sc timeout 600 python tools/opt_sweep.py cfg_c2 12500 "" "no_pool=1" >> $O 2>&1
done
echo "== timeline C1 m2 default" >> $O
timeout 300 python tools/timeline_steps.py 10000 cfg_c1 >> $O 2>&1
echo "== timeline C1 m2 no_pool" >> $O
timeout 300 python tools/timeline_steps.py 10000 cfg_c1 no_pool=1 >> $O 2>&1
cat $O
