set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench rc=$?" >> gpurun_out/bench1.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err; echo "rc=$?" >> gpurun_out/bench_dist1.err
rocminfo | grep -E "Compute Unit|Max Clock|Marketing" | head -8 > gpurun_out/rocminfo.txt
nproc >> gpurun_out/rocminfo.txt; lscpu | grep "Model name" >> gpurun_out/rocminfo.txt
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/bench1.json; tail -3 gpurun_out/bench1.err; cat gpurun_out/bench_dist1.json; tail -5 gpurun_out/bench_dist1.err
