"""bench.py at N > 1, end to end on the CPU: two ranks under torch.distributed.run with the gloo backend and a stand-in for the
engine (tests/bench_stub_engine.py).  What is pinned: the weak leg and the `strong` object of the ONE JSON line -- ONE batch of
BASELINE configs[2] sharded over the ranks by LPT, at least 10 timed steps behind 2 warm-up steps by default, the WHOLE batch
compared with a single-rank run (strong_scaling_check), the recorded 1-GPU time beside it when profiles/strong_1gpu.json matches,
and that an empty shard stops every rank instead of hanging the others.  (A rank of the real bench needs a GPU: the driver runs
that.)"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def run_bench(args, timeout=600):
    env = dict(os.environ, AGATHA_BENCH_BACKEND="gloo", AGATHA_BENCH_ENGINE="tests.bench_stub_engine", AGATHA_BENCH_CHECK_PIECE="128",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *args]
    return subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_strong_leg_over_two_gloo_ranks():
    p = run_bench(["--steps", "2", "--warmup", "1", "--pairs", "48", "--strong-pairs", "300", "--strong-steps", "3"])
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 2 and out["config"]["n_ranks_seen_by_rccl"] == 2
    assert out["value"] > 0 and out["config"]["pairs_per_gpu"] == 48
    st = out["strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["n_ranks_seen_by_rccl"] == 2
    assert st["steps"] == 3 and st["warmup"] >= 2                      # max(--steps, --strong-steps); never fewer than 2 warm-up steps
    assert st["strong_scaling_check"].startswith("300/300 pairs (the whole batch"), st["strong_scaling_check"]
    assert 0 < st["pairs_rank0"] < 300 and 1.0 <= st["shard_imbalance_max_over_mean"] < 1.5
    assert st["value"] > 0 and st["ms_per_step"] > 0
    # the recorded single-GPU time is for the batch BASELINE.json names (100 000 pairs): not this one
    assert "speedup_vs_recorded_1gpu" in st and st["speedup_vs_recorded_1gpu"] is None


def test_default_strong_leg_has_ten_steps():
    sys.path.insert(0, ROOT)
    import bench
    src = open(bench.__file__).read()
    assert 'add_argument("--strong-steps", type=int, default=10' in src
    assert "max(a.steps, a.strong_steps)" in src and "max(a.warmup, 2)" in src


def test_an_empty_shard_stops_every_rank():
    # 64 pairs = ONE chunk for two ranks: the second rank's share is empty; both must exit (non-zero), neither may hang
    p = run_bench(["--steps", "1", "--warmup", "0", "--pairs", "16", "--strong-pairs", "64", "--strong-steps", "1"], timeout=300)
    assert p.returncode != 0
    assert "without pairs" in p.stderr.decode()
