# instruction cache counters of the int16 kernel on C1, plain and with N runs (run through gpurun)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for nf in 0 0.02; do
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/icache_$nf -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gasal-api --n-run-frac $nf > /dev/null 2> gpurun_out/icache_$nf.err
python3 - $nf <<'PY'
import csv,glob,sys,collections
nf=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(f'gpurun_out/icache_{nf}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:48]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        if r['Counter_Name']=='SQC_ICACHE_REQ': cnt[k]+=1
for k,v in acc.items():
    if 'align16' in k and v.get('SQC_ICACHE_REQ',0)>1e6:
        n=max(cnt[k],1)
        print('n-run-frac',nf,k,'launches',n,{c: round(x/n) for c,x in v.items()}, 'miss rate %.4f'%(v['SQC_ICACHE_MISSES']/max(v['SQC_ICACHE_REQ'],1)))
PY
done
