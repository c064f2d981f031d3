"""Developer tool (GPU box): the stream / batch manager's sustained feed (tools/gasal_api_timing.py time_pipeline: 16 batches of 8 192 C1 pairs through
`manual`, 1 / 2 / 4 host threads, host ASCII / -k / -K) under several limits on the batches whose align kernels may be on the chip at once
(AGATHA_AMD_MAX_INFLIGHT, libgasal_amd's gate; 0 = none: round 5's behaviour).   python3 tools/pipe_sweep.py [limits, comma separated]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gasal_api_timing as G
from agatha_amd import workload
limits = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 2, 3]
kernel_gcups = float(os.environ.get("KERNEL_GCUPS", "5480"))
qs, ts = workload.cfg_c1(n=16384)
for lim in limits:
    os.environ["AGATHA_AMD_MAX_INFLIGHT"] = str(lim)
    r = G.time_pipeline(qs, ts, dict(m=2, x=4, q=4, r=2), 751, 400, kernel_gcups=kernel_gcups)
    print("limit %d:" % lim, " ".join("%dthr%s=%.0f" % (x["host_threads"], {False: "", True: "-k", 2: "-K"}[x["host_packed"]], x["end_to_end_gcups"]) for x in r["runs"]),
          "| steady %.0f (%s)" % (r["steady_state"]["marginal_gcups"], r["best_config"]), flush=True)
