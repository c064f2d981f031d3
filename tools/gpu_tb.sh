# traceback pass: wall time next to plain align, then the per-kernel split (run through gpurun)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
N=${1:-2000}
python3 tools/gpu_tb.py $N 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tb_prof -o tb -- python3 tools/gpu_tb.py $N > gpurun_out/tb_prof.log 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/tb_prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:100], r['Calls'], r['AverageNs'], r['Percentage'])
PY
