cd $GRAFT_REPO_ROOT
for B in 512 480 448 417 400 384 334 320 256; do
  AGATHA_AMD_MAX_BLOCKS=$B python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$B', round(d['value'],1), round(d['kernel_ms'],2))"
done
