cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in "" $GRAFT_REPO_ROOT/ab_v5.so; do
name=$(basename ${lib:-tree})
AGATHA_AMD_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3_$name -o t -- python3 bench.py --config C3 --steps 3 --warmup 1 --no-cpu-baseline --no-gasal-api > /dev/null 2>&1
python3 - $name <<'PY'
import csv,glob,sys
for f in glob.glob(f'gpurun_out/c3_{sys.argv[1]}/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:5]: print(sys.argv[1], r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
