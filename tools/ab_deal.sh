# developer tool (GPU box): A/B of the dealt first round (AGATHA_AMD_NO_DEAL=1: plain queue) over batch sizes
cd $GRAFT_REPO_ROOT
for n in ${SIZES:-8192 9000 10000 11000 12000 12288 13000 14000 15000 16384}; do for d in 0 1; do AGATHA_AMD_NO_DEAL=$d python bench.py --pairs $n --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print($n, 'no_deal=$d', round(b[\"value\"],1), round(b[\"kernel_ms\"],2))"; done; done
for d in 0 1; do echo no_deal=$d; AGATHA_AMD_NO_DEAL=$d python tools/bench_configs.py 2>&1 | grep "C0\|C2" | cut -c1-125; done
