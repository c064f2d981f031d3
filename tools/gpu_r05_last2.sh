cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_last2.txt
: > $O
timeout 40 python tools/gpu_probation_loop.py 5 1 >> $O 2>&1; echo "loop rc=$?" >> $O
grep -q "loop rc=0" $O || { cat $O; exit 1; }
timeout 205 python -m pytest tests/test_gpu_int16.py tests/test_gpu_ref_scoring.py -x -q 2>&1 | tail -6 >> $O
cat $O
