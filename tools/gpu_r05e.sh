cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05e_window.txt
: > $O
for sc in 2,4,4,2 1,4,6,2; do
  echo "== C1 10000 pairs, scoring $sc" >> $O
  SCORING=$sc timeout 600 python tools/opt_sweep.py cfg_c1 10000 "" "win_cap_div=12" "win_cap_div=24" >> $O 2>&1
  echo "== C0 20000 pairs, scoring $sc" >> $O
  SCORING=$sc timeout 600 python tools/opt_sweep.py cfg_c0 20000 "" "win_cap_min=96" "win_cap_min=160" >> $O 2>&1
  echo "== C2 12500 pairs, scoring $sc" >> $O
  SCORING=$sc timeout 600 python tools/opt_sweep.py cfg_c2 12500 "" >> $O 2>&1
done
cat $O
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r05e.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05e.log
tail -4 gpurun_out/pytest_r05e.log
