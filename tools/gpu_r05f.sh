cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ref_scoring.py -m gpu -q -x > gpurun_out/pytest_r05f.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r05f.log
tail -30 gpurun_out/pytest_r05f.log
# the 1-GPU denominator of the strong leg (100 000 HiFi pairs in ONE batch)
timeout 1500 python tools/record_strong_1gpu.py gpurun_out/strong_1gpu.json 100000 10 > gpurun_out/strong_1gpu.log 2>&1
tail -3 gpurun_out/strong_1gpu.log
