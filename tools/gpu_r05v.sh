cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05v_ckshift.txt
: > $O
for cs in 28 29 30; do
  echo "== ck_shift=$cs" >> $O
  AGATHA_AMD_CK_SHIFT=$cs timeout 300 python tools/gpu_skew.py 10000 2>&1 | grep -E "equal|broken" | cut -c1-150 >> $O
done
cat $O
