#!/bin/bash
# Developer tool: ONE parameterised wrapper for everything that runs on the GPU box through gpurun (round 6; the rounds before kept a
# one-shot script per call).   gpurun --timeout S -- 'bash tools/gpu_run.sh <recipe> [args]'
# Everything is written under gpurun_out/<TAG>/ (TAG = $GPU_TAG, default the recipe's name); A/B libraries live in ab/ (git-ignored,
# pushed to the box).  Recipes may be chained: tools/gpu_run.sh suite + bench + profile r06_v1
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
BENCHQ="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-gasal-api"

kernel_ms() {   # kernel_ms <label> [bench args]: kernel time of one bench leg as one line
    local label=$1; shift
    timeout 400 $BENCHQ "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    b=json.loads(sys.stdin.read()); print('$label', 'kernel_ms', round(b['kernel_ms'],3), 'ms_per_step', round(b['ms_per_step'],3), 'value', round(b['value'],1))
except Exception as e: print('$label', 'FAILED', e)"
}

recipe() {
    local r=$1; shift
    local out=gpurun_out/${GPU_TAG:-$r}; mkdir -p $out
    case $r in
    suite)      # the GPU test suite, as the driver runs it
        timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $out/suite.txt ;;
    tests)      # tests <pytest args>: a part of it, optionally on another library (AGATHA_AMD_LIB)
        timeout 900 python3 -m pytest "$@" -x -q 2>&1 | tail -15 | tee -a $out/tests.txt ;;
    bench)      # the bench line as the driver runs it
        timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; tail -c 1500 $out/bench_default.json ;;
    quick)      # kernel time of C1 (and the configs in $CONFIGS) on the tree's library and on every ab/*.so, alternating, $REPS times
        for rep in $(seq 1 ${REPS:-2}); do
          for lib in "" $(ls ab/*.so 2>/dev/null); do
            for c in ${CONFIGS:-C1}; do
              AGATHA_AMD_LIB=${lib:+$PWD/$lib} kernel_ms "lib=$(basename ${lib:-tree}) $c" --config $c
            done; done; done | tee -a $out/quick.txt ;;
    profile)    # rocprofv3 --stats + the PMC passes of the bench command -> gpurun_out/prof_<TAG>_*; collate with tools/collate_profile.py
        bash tools/gpu_profile.sh ${1:-r06} ;;
    pmc1)       # issued VALU instructions per launch only (one pass)
        rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc1 -o pmc -- $BENCHQ > /dev/null 2> $out/pmc1.err
        python3 - <<PYEOF | tee $out/pmc1.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$out/pmc1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "align16_kernel" in r["Kernel_Name"] and float(r["Counter_Value"]) > 1e6: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print({k: "%.4e (%d launches)" % (v, len(agg[k])) for k, v in sorted(m.items())})
if "SQ_INSTS_VALU" in m: print("issued VALU lane-ops per cell (C1, 1.43658e11 cells): %.3f" % (m["SQ_INSTS_VALU"] * 64 / 1.43658e11))
PYEOF
        ;;
    pmcx)       # pmcx <config> <counter> ...: one counter pass of `bench.py --config <config> $PMCX_ARGS`, mean per launch of the int16 kernel (where a lone wave's time goes: the WAIT / ACTIVE / LEVEL counters)
        local c=$1; shift
        rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/pmcx_$c -o pmc -- $BENCHQ --config $c $PMCX_ARGS > /dev/null 2>> $out/pmcx.err
        python3 - <<PYEOF | tee -a $out/pmcx.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$out/pmcx_$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "align16_kernel" in r["Kernel_Name"]: agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, n), v in sorted(agg.items()):
    if max(v) > 0: print("$c %-42s %-24s mean %.4e max %.4e (%d launches)" % (k, n, sum(v) / len(v), max(v), len(v)))
PYEOF
        ;;
    pcsample)   # PC sampling of the headline kernel (beta; may be refused on the box)
        timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval ${1:-500} \
            --kernel-trace --output-format csv -d $out/pcs -o pcs -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gasal-api > $out/pcs.json 2> $out/pcs.err
        tail -5 $out/pcs.err; ls -la $out/pcs 2>/dev/null | head; python3 tools/pcs_summary.py $out/pcs 2>&1 | tail -60 | tee $out/pcs_summary.txt ;;
    dips)       # dips <lib|tree> <scoring> <cfg> <band> [pairs]: reads with a burst of errors (tools/gpu_dips.py)
        local lib=$1 sc=$2 cfg=$3 band=$4 n=${5:-10000}
        [ "$lib" = tree ] && lib= || lib=$PWD/ab/$lib
        echo "== dips lib=${lib:-tree} scoring=$sc cfg=$cfg band=$band n=$n" | tee -a $out/dips.txt
        AGATHA_AMD_LIB=$lib SCORING=$sc CFG=$cfg BAND=$band timeout 600 python3 tools/gpu_dips.py $n 0.1 2>&1 | tee -a $out/dips.txt ;;
    skew)       # skew <lib|tree> <scoring>: unequal lengths / broken reads (tools/gpu_skew.py)
        local lib=$1 sc=$2
        [ "$lib" = tree ] && lib= || lib=$PWD/ab/$lib
        echo "== skew lib=${lib:-tree} scoring=$sc" | tee -a $out/skew.txt
        AGATHA_AMD_LIB=$lib SCORING=$sc timeout 600 python3 tools/gpu_skew.py 2>&1 | tee -a $out/skew.txt ;;
    fuzz)       # fuzz <lib|tree> <seconds>: the randomised parity sweeps (static schedule, mixed short reads, long reads)
        local lib=$1 s=${2:-60}
        [ "$lib" = tree ] && lib= || lib=$PWD/ab/$lib
        AGATHA_AMD_LIB=$lib timeout $((s + 120)) python3 tools/gpu_fuzz_mig.py $s 2>&1 | tail -3 | tee -a $out/fuzz.txt
        AGATHA_AMD_LIB=$lib timeout $((s + 120)) python3 tools/gpu_fuzz.py $s 2>&1 | tail -3 | tee -a $out/fuzz.txt
        AGATHA_AMD_LIB=$lib timeout $((s / 2 + 180)) python3 tools/gpu_fuzz_long.py $((s / 2)) 2>&1 | tail -14 | tee -a $out/fuzz.txt ;;
    curve)      # throughput as a function of the batch size (tools/batch_size_curve.py)
        timeout 900 python3 tools/batch_size_curve.py "$@" 2>&1 | tee $out/batch_size_curve.txt ;;
    configs)    # bench lines of the other BASELINE shapes
        for c in C0 C2 C3 C4; do timeout 600 python3 bench.py --config $c --steps 4 --warmup 1 --no-gasal-api 2>/dev/null | tail -1 >> $out/bench_configs.jsonl; done
        python3 -c "
import json
for l in open('$out/bench_configs.jsonl'):
    b=json.loads(l); print(b['config'].get('workload','?')[:40], 'kernel_ms', round(b.get('kernel_ms',0),2), 'value', round(b['value'],1))" ;;
    tl)         # tl [pairs] [cfg]: the wave timeline of one batch on the tree's library and on every ab/*.so (kernel end = the last wave's end)
        for lib in "" $(ls ab/*.so 2>/dev/null); do
            echo "== lib=$(basename ${lib:-tree})" | tee -a $out/tl.txt
            AGATHA_AMD_LIB=${lib:+$PWD/$lib} timeout 180 python3 tools/timeline_steps.py ${1:-10000} ${2:-cfg_c1} 2>&1 | head -4 | tee -a $out/tl.txt
        done ;;
    lib)        # lib <name|tree>: the library (ab/<name>) every later recipe of the chain loads
        if [ "$1" = tree ]; then unset AGATHA_AMD_LIB; else export AGATHA_AMD_LIB=$PWD/ab/$1; fi ;;
    py)         # py <script> [args]: any tool of this directory
        timeout ${PY_TIMEOUT:-600} python3 "$@" 2>&1 | tee -a $out/py.txt ;;
    *) echo "unknown recipe $r"; return 2 ;;
    esac
}

# recipes separated by '+'
args=()
for a in "$@" +; do
    if [ "$a" = + ]; then [ ${#args[@]} -gt 0 ] && recipe "${args[@]}"; args=(); else args+=("$a"); fi
done
