# round 5: parity tests that touch the static schedule / value steps, then the two scorings with the interval permutation on / off
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_int16.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/pytest_r05b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05b.log
tail -5 gpurun_out/pytest_r05b.log
O=gpurun_out/r05b_ref_scoring.txt
: > $O
for sc in 2,4,4,2 1,4,6,2; do
  echo "== C1 10000 pairs, scoring $sc" >> $O
  SCORING=$sc timeout 600 python tools/opt_sweep.py cfg_c1 10000 "" "mig_identity=1" "" "mig_identity=1" "fast_margin=0" >> $O 2>&1
  echo "== C0 20000 pairs, scoring $sc" >> $O
  SCORING=$sc timeout 600 python tools/opt_sweep.py cfg_c0 20000 "" "fast_margin=0" >> $O 2>&1
  echo "== C2 12500 pairs, scoring $sc" >> $O
  SCORING=$sc timeout 600 python tools/opt_sweep.py cfg_c2 12500 "" "mig_identity=1" >> $O 2>&1
done
cat $O
