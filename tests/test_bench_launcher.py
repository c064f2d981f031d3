"""bench.py --gpus N started as a plain process (the way the driver starts --gpus 1): the launcher branch must start N ranks with
torch.distributed.run as a CHILD process, before torch or the HIP library is imported in the parent, and hand on the child's exit
code; rank 0's line must say n_gpus: N.  CPU only: the ranks are a stub worker (a rank of the real bench needs a GPU)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent('''
    import json, os, sys
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    open(os.path.join(os.environ["STUB_OUT"], "rank%d" % rank), "w").write(" ".join(sys.argv[1:]))
    if rank == 0:
        print(json.dumps({"n_gpus": world, "argv": sys.argv[1:], "master": os.environ.get("MASTER_ADDR")}), flush=True)
    sys.exit(int(os.environ.get("STUB_RC", "0")))
''')


def run_parent(tmp_path, n, rc=0):
    stub = tmp_path / "stub_worker.py"
    stub.write_text(STUB)
    out = tmp_path / "ranks"
    out.mkdir(exist_ok=True)
    code = textwrap.dedent(f'''
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        rc = bench.maybe_spawn_ranks(["--gpus", "{n}", "--steps", "2", "--warmup", "1"])
        assert "torch" not in sys.modules and "agatha_amd" not in sys.modules, "the parent touched torch / the HIP library"
        sys.exit(100 if rc is None else rc)
    ''')
    env = dict(os.environ, AGATHA_BENCH_WORKER=str(stub), STUB_OUT=str(out), STUB_RC=str(rc))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    return p, out


def test_gpus_n_starts_n_ranks_as_children(tmp_path):
    p, out = run_parent(tmp_path, 3)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert sorted(os.listdir(out)) == ["rank0", "rank1", "rank2"]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 3 and line["master"] == "127.0.0.1"
    assert line["argv"] == ["--gpus", "3", "--steps", "2", "--warmup", "1"]          # the ranks see the same flags


def test_child_exit_code_is_handed_on(tmp_path):
    p, _ = run_parent(tmp_path, 2, rc=7)
    assert p.returncode != 0 and p.returncode != 100


def test_single_gpu_and_ranks_do_not_spawn():
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    fake = lambda cmd, env=None: calls.append(cmd) or subprocess.CompletedProcess(cmd, 0)
    assert bench.maybe_spawn_ranks(["--steps", "3"], run=fake) is None
    assert bench.maybe_spawn_ranks(["--gpus", "1"], run=fake) is None
    old = os.environ.get("WORLD_SIZE")
    os.environ["WORLD_SIZE"] = "8"
    try:
        assert bench.maybe_spawn_ranks(["--gpus", "8"], run=fake) is None       # already a rank of a launcher
    finally:
        if old is None:
            del os.environ["WORLD_SIZE"]
        else:
            os.environ["WORLD_SIZE"] = old
    assert calls == []
    old = os.environ.pop("WORLD_SIZE", None)
    try:
        assert bench.maybe_spawn_ranks(["--gpus=4", "--config", "C2"], run=fake) == 0
    finally:
        if old is not None:
            os.environ["WORLD_SIZE"] = old
    cmd = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[-3:] == ["--gpus=4", "--config", "C2"]
