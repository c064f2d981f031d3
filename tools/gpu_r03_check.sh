# round 3: whole GPU suite, then both fuzzers for a while (value steps are the default of the int16 kernel)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout 400 python tools/gpu_fuzz.py ${FUZZ_S:-150} ${SEED:-31} > gpurun_out/fuzz_$TAG.txt 2>&1; tail -4 gpurun_out/fuzz_$TAG.txt
timeout 400 python tools/gpu_fuzz_mig.py ${FUZZ_S:-150} ${SEED:-32} > gpurun_out/fuzz_mig_$TAG.txt 2>&1; tail -4 gpurun_out/fuzz_mig_$TAG.txt
