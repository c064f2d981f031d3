# Developer tool (GPU box): kernel trace of the stream / batch manager over a sustained feed -- where the time between two align
# kernels goes.  usage: bash tools/gpu_pipe_trace.sh [host threads] [extra manual flags]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NT=${1:-2}; shift
D=/tmp/pipe_trace; mkdir -p $D
python3 - <<PY
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from agatha_amd import workload
import gasal_api_timing as g
qs, ts = workload.cfg_c1(n=16384)
g.write_fasta("$D/ref.fasta", qs); g.write_fasta("$D/query.fasta", ts)
PY
cd /tmp
AGATHA_AMD_REPEAT=8 AGATHA_AMD_LOOP_STATS=$D/loop.txt timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pipe_trace_$NT -o p -- $GRAFT_REPO_ROOT/agatha_amd/manual "$@" -m 2 -x 4 -q 4 -r 2 -s 3 -z 400 -w 751 -a 8192 -n $NT $D/ref.fasta $D/query.fasta > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pipe_trace_$NT.err
cd $GRAFT_REPO_ROOT
cat $D/loop.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pipe_trace_$NT/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))) for r in rows]
ks.sort()
al = [k for k in ks if "align16_kernel<16, 3" in k[2]]
t0 = al[0][0]
print("align16 kernels:", len(al), "span ms %.2f" % ((al[-1][1] - t0) / 1e6), "sum of kernel ms %.2f" % (sum(e - s for s, e, _, _ in al) / 1e6))
for i, (s, e, n, q) in enumerate(al):
    gap = (s - al[i - 1][1]) / 1e3 if i else 0.0
    between = collections.Counter()
    if i:
        for s2, e2, n2, q2 in ks:
            if s2 >= al[i - 1][1] and e2 <= s and "align16_kernel<16, 3" not in n2:
                between[n2.split("(")[0].replace("agatha::", "")[:28]] += (e2 - s2) / 1e3
    print("  #%2d stream %s start %.2f ms dur %.2f ms gap before %.0f us; kernels inside the gap (us): %s" % (i, q, (s - t0) / 1e6, (e - s) / 1e6, gap, dict((k, round(v)) for k, v in between.most_common(6))))
mc = glob.glob("gpurun_out/pipe_trace_$NT/*memory_copy_trace.csv")
if mc:
    rows = list(csv.DictReader(open(mc[0])))
    tot = collections.Counter(); 
    for r in rows: tot[r.get("Direction", r.get("Kind", "?"))] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print("memory copies (ms):", dict((k, round(v, 1)) for k, v in tot.items()), "count", len(rows))
PY
