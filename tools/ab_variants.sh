cd $GRAFT_REPO_ROOT
for V in $VARIANTS; do
  if [ -f agatha_amd/libagatha_amd_var_$V.so ]; then
  AGATHA_AMD_LIB=$GRAFT_REPO_ROOT/agatha_amd/libagatha_amd_var_$V.so python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$V', round(d['value'],1), round(d['kernel_ms'],2))"
  fi
done
