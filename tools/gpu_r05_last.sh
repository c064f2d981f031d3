cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_last.txt
: > $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O 2>&1 || { cat $O; exit 1; }
timeout 1150 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 >> $O
timeout 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gasal-api --no-pipeline 2>&1 | tail -1 | cut -c1-900 >> $O
cat $O
