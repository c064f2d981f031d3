# round 3, final tree: the profile of the bench command, the other workload shapes, the traceback pass, the N-run variant (run through gpurun)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r03v4}
bash tools/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
# what the checkpoints write: the same counters with checkpoints off
AGATHA_AMD_CK_MIN_STEPS=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_nock_w -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gasal-api > /dev/null 2> gpurun_out/prof_${TAG}_nock_w.err
AGATHA_AMD_CK_MIN_STEPS=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_nock_f -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gasal-api > /dev/null 2> gpurun_out/prof_${TAG}_nock_f.err
python3 tools/bench_configs.py > gpurun_out/bench_configs_$TAG.json 2> gpurun_out/bench_configs_$TAG.err; tail -3 gpurun_out/bench_configs_$TAG.json | cut -c1-600
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cut -c1-400 gpurun_out/bench_$TAG.json
python3 bench.py --pairs 8192 --no-cpu-baseline --no-gasal-api 2>/dev/null > gpurun_out/bench_8192_$TAG.json; python3 -c "
import json; b=json.load(open('gpurun_out/bench_8192_$TAG.json')); print('8192 pairs kernel_ms', b['kernel_ms'], 'value', b['value'])"
for c in C0 C2 C3 C4; do python3 bench.py --config $c --no-cpu-baseline --no-gasal-api 2>/dev/null > gpurun_out/bench_${c}_$TAG.json; python3 -c "
import json; b=json.load(open('gpurun_out/bench_${c}_$TAG.json')); print('$c kernel_ms', b['kernel_ms'], 'value', b['value'], b['unit'])"; done
bash tools/gpu_tb.sh 6000 > gpurun_out/tb_6000_$TAG.txt 2>&1; cp gpurun_out/tb_prof/tb_kernel_stats.csv gpurun_out/tb_6000_kernel_stats_$TAG.csv; tail -9 gpurun_out/tb_6000_$TAG.txt | head -6
bash tools/gpu_tb.sh 2000 > gpurun_out/tb_2000_$TAG.txt 2>&1; cp gpurun_out/tb_prof/tb_kernel_stats.csv gpurun_out/tb_2000_kernel_stats_$TAG.csv; tail -9 gpurun_out/tb_2000_$TAG.txt | head -3
AGATHA_AMD_NO_INT16=1 python3 tools/gpu_tb.py 6000 2>&1 | tail -1 | tee gpurun_out/tb_6000_int32_$TAG.txt
bash tools/gpu_nrun2.sh > gpurun_out/nrun_$TAG.txt 2>&1; cat gpurun_out/nrun_$TAG.txt
