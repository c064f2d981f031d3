# rocprofv3 passes for the bench command (run through gpurun).  Summaries land in gpurun_out/prof_*.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r01}
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gasal-api"
cat /sys/fs/cgroup/cpu.max > gpurun_out/cpu_quota.txt 2>&1; cat /sys/fs/cgroup/cpuset.cpus.effective >> gpurun_out/cpu_quota.txt 2>&1
python3 -c "import os; print(len(os.sched_getaffinity(0)))" >> gpurun_out/cpu_quota.txt
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_trace -o trace -- $BENCH > gpurun_out/prof_${TAG}_trace.json 2> gpurun_out/prof_${TAG}_trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/prof_${TAG}_pmc1 -o pmc -- $BENCH > /dev/null 2> gpurun_out/prof_${TAG}_pmc1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_${TAG}_pmc2 -o pmc -- $BENCH > /dev/null 2> gpurun_out/prof_${TAG}_pmc2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_${TAG}_pmc3 -o pmc -- $BENCH > /dev/null 2> gpurun_out/prof_${TAG}_pmc3.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_INSTS_VALU --output-format csv -d gpurun_out/prof_${TAG}_pmc4 -o pmc -- $BENCH > /dev/null 2> gpurun_out/prof_${TAG}_pmc4.err
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_I8 --output-format csv -d gpurun_out/prof_${TAG}_pmc5 -o pmc -- $BENCH > /dev/null 2> gpurun_out/prof_${TAG}_pmc5.err
find gpurun_out -name "*.csv" | head -40
du -sh gpurun_out
