# round 5: the tracked measurements -- bench lines of every BASELINE shape, the reference's bench scoring, rocprofv3 stats + PMC of the
# headline bench command and of the other shapes' kernels.   bash tools/gpu_r05_final.sh TAG
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05}
timeout 120 python tools/opt_sweep.py cfg_c1 10000 "" > /dev/null 2>&1 || { echo "SMOKE FAILED"; exit 1; }
bash tools/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
: > gpurun_out/bench_configs_$TAG.jsonl
for c in C0 C2 C3 C4; do
  timeout 900 python bench.py --config $c --steps 5 --warmup 2 --no-gasal-api >> gpurun_out/bench_configs_$TAG.jsonl 2> gpurun_out/bench_$c_$TAG.err
done
timeout 600 python bench.py --config C4 --pairs 20000 --steps 3 --warmup 1 --no-gasal-api --no-cpu-baseline >> gpurun_out/bench_configs_$TAG.jsonl 2>> gpurun_out/bench_C4_$TAG.err
timeout 600 python bench.py --config C3 --pairs 1024 --steps 3 --warmup 1 --no-gasal-api --no-cpu-baseline >> gpurun_out/bench_configs_$TAG.jsonl 2>> gpurun_out/bench_C3_$TAG.err
for sc in m1x4q6r2; do
  timeout 900 python bench.py --steps 10 --warmup 2 --scoring $sc --no-pipeline > gpurun_out/bench_ref_scoring_$TAG.json 2> gpurun_out/bench_ref_scoring_$TAG.err
  timeout 900 python bench.py --config C0 --steps 10 --warmup 2 --scoring $sc --no-pipeline > gpurun_out/bench_ref_scoring_C0_$TAG.json 2>> gpurun_out/bench_ref_scoring_$TAG.err
done
for c in C0 C2 C3 C4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_${c}_trace -o trace -- python3 tools/one_config.py $c > gpurun_out/prof_${TAG}_${c}_trace.txt 2> gpurun_out/prof_${TAG}_${c}_trace.err
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/prof_${TAG}_${c}_pmc1 -o pmc -- python3 tools/one_config.py $c > /dev/null 2> gpurun_out/prof_${TAG}_${c}_pmc1.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/prof_${TAG}_${c}_pmc2 -o pmc -- python3 tools/one_config.py $c > /dev/null 2> gpurun_out/prof_${TAG}_${c}_pmc2.err
  tail -1 gpurun_out/prof_${TAG}_${c}_trace.txt | cut -c1-200
done
python3 -c "
import json
for l in open('gpurun_out/bench_configs_$TAG.jsonl'):
    b=json.loads(l); print(b['config']['workload'][:40], round(b['value'],1),'GCUPS kernel_ms',round(b['kernel_ms'],2), b['config']['kernel'], (b.get('effective_cells') or {}).get('ratio'))
for f in ('gpurun_out/bench_ref_scoring_$TAG.json','gpurun_out/bench_ref_scoring_C0_$TAG.json'):
    b=json.load(open(f)); print(b['config']['workload'][:60], round(b['value'],1),'GCUPS kernel_ms',round(b['kernel_ms'],2), b['config']['int16_steps_rank0']['pairs_started_over'])
"
du -sh gpurun_out
