cd $GRAFT_REPO_ROOT
for N in 4096 8192 10000 12288 16384 20000 40000; do
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --pairs $N 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$N', round(d['value'],1), round(d['kernel_ms'],2))"
done
