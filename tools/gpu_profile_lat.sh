# rocprofv3 passes for the latency regimes: C3 (256 pairs, <128,1>) and C4 (6000 pairs, <64,1>), kernel-only tool tools/one_config.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03}
for c in C3 C4; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_${c}_trace -o trace -- python3 tools/one_config.py $c > gpurun_out/prof_${TAG}_${c}_trace.txt 2> gpurun_out/prof_${TAG}_${c}_trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/prof_${TAG}_${c}_pmc1 -o pmc -- python3 tools/one_config.py $c > /dev/null 2> gpurun_out/prof_${TAG}_${c}_pmc1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_${TAG}_${c}_pmc2 -o pmc -- python3 tools/one_config.py $c > /dev/null 2> gpurun_out/prof_${TAG}_${c}_pmc2.err
tail -1 gpurun_out/prof_${TAG}_${c}_trace.txt | cut -c1-200
done
