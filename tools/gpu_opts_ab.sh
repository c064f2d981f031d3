# A/B of debug options (environment) on the tree's library in one call: OPTS="NAME=V,NAME=V;NAME=V;..." (first entry may be empty)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
IFS=';' read -ra SETS <<< "${OPTS:-;AGATHA_AMD_CK_MIN_STEPS=4096;AGATHA_AMD_FAST_MARGIN=10;AGATHA_AMD_FAST_MARGIN=10,AGATHA_AMD_CK_MIN_STEPS=4096}"
for rep in 1 2; do
for set in "${SETS[@]}"; do
for c in ${CONFIGS:-C1 C0 C2 C4}; do
( IFS=','; for kv in $set; do export "$kv"; done
  timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-gasal-api 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('opts=[$set] $c kernel_ms',round(b['kernel_ms'],3),'value',round(b['value'],4))" )
done; done; done
