cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python bench.py > gpurun_out/bench_default_r05w.json 2> gpurun_out/bench_default_r05w.err ) 2> gpurun_out/bench_default_r05w.time
cat gpurun_out/bench_default_r05w.time | tail -3
python -c "
import json; b=json.load(open('gpurun_out/bench_default_r05w.json'))
print('value', round(b['value'],1), b['unit'], 'ms_per_step', round(b['ms_per_step'],3), 'kernel_ms', round(b['kernel_ms'],3), 'dtype', b['dtype'])
print('roofline', {k:(round(v,5) if isinstance(v,float) else v) for k,v in b['roofline'].items() if k in ('achieved','frac','traffic','traffic_bytes_per_launch')})
print('roofline_valu', {k:(round(v,4) if isinstance(v,float) else v) for k,v in b['roofline_valu'].items() if k in ('frac','ops_per_cell','frac_vs_guide_2cycle_issue','frac_at_round2_count_of_5_per_cell','issued_lane_ops_per_cell')})
print('cpu', b['cpu_baseline']['value'], b['cpu_baseline']['cores'], b['cpu_baseline']['gpu_results_checked'])
g=b['gasal_api']; print('ref cmd', g['reference_bench_command']['kernel_ms_per_batch'], g['reference_bench_command']['gcups_by_raw_log'], g['reference_bench_command']['pairs_started_over'])
print('pipeline', g['pipeline']['best_end_to_end_gcups'], g['pipeline']['best_host_packed_vs_kernel_only'], g['pipeline']['steady_state'])
print([ (r['host_threads'], r['host_packed'], round(r['end_to_end_gcups'])) for r in g['pipeline']['runs']])
"
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -6
