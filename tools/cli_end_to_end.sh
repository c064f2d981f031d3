# Developer tool (run through gpurun): the `manual` CLI end to end on the C1 workload written as FASTA.
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, time
sys.path.insert(0, ".")
from agatha_amd import workload as synth
qs, ts = synth.cfg_c1(n=10000)
for path, seqs in (("/tmp/ref.fasta", qs), ("/tmp/query.fasta", ts)):
    with open(path, "w") as f:
        for k, s in enumerate(seqs):
            f.write(f">>> {k+1}\n{s.decode()}\n")
PY
ls -la /tmp/ref.fasta /tmp/query.fasta
rm -f /tmp/raw.log
for A in 8192 10000; do
  for N in 1 2; do
    T0=$(date +%s.%N); ./agatha_amd/manual -p -m 2 -x 4 -q 4 -r 2 -s 3 -z 400 -w 751 -a $A -n $N /tmp/ref.fasta /tmp/query.fasta /tmp/raw_${A}_${N}.log > /tmp/score_${A}_${N}.log; T1=$(date +%s.%N); python3 -c "print('wall %.2f s (a=$A n=$N)' % ($T1 - $T0))"
    echo "kernel ms per batch:"; cat /tmp/raw_${A}_${N}.log | tr '\n' ' '; echo; wc -l /tmp/score_${A}_${N}.log; md5sum /tmp/score_${A}_${N}.log
  done
done
T0=$(date +%s.%N); ./agatha_amd/manual -m 2 -x 4 -q 4 -r 2 -s 3 -z 400 -w 751 /tmp/ref.fasta /tmp/query.fasta; T1=$(date +%s.%N); python3 -c "print('wall %.2f s (no -p)' % ($T1 - $T0))"
