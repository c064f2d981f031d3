# round-3 baseline: the VALU-roof microbenchmarks (recorded again: profiles/r03_*), then the quick perf check
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_base
hipcc -O2 --offload-arch=gfx950 tools/microbench/valu_rate.hip -o /tmp/valu_rate && timeout 120 /tmp/valu_rate > gpurun_out/r03_base/valu_rate.txt 2>&1
hipcc -O2 --offload-arch=gfx950 tools/microbench/dep_chain.hip -o /tmp/dep_chain && timeout 120 /tmp/dep_chain > gpurun_out/r03_base/dep_chain.txt 2>&1
tail -5 gpurun_out/r03_base/valu_rate.txt
bash tools/gpu_perf.sh r03base
