cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_last3.txt
: > $O
timeout 30 python tools/gpu_probation_loop.py 3 1 >> $O 2>&1; echo "loop rc=$?" >> $O
grep -q "loop rc=0" $O || { cat $O; exit 1; }
timeout 85 python -m pytest tests/test_gpu_ref_scoring.py tests/test_gpu_int16.py -x -q -k "burst_of_errors or checkpoints_and or started_over or give_up_late" 2>&1 | tail -4 >> $O
cat $O
