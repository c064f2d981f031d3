"""The product's real entry point timed by the reference's own protocol (AGAThA.sh:44,52 + gasal_align.cu:219-236): the CLI
`agatha_amd/manual -p ...` reads two FASTA files, pushes batches of -a pairs through gasal_aln_async on two streams per host
thread (-n threads), the library appends every batch's kernel milliseconds to raw.log, and the sum of raw.log is what
AGAThA.sh averages into time.json.  Used by bench.py (the "gasal_api" object) and as a tool:

    python tools/gasal_api_timing.py [C1|C0] [pairs]        # -a 8192 / 10000, -n 1 / 2; one JSON object per line
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MANUAL = os.path.join(ROOT, "agatha_amd", "manual")


def write_fasta(path, seqs):
    """The reference's two-line records (test_prog.cpp:94-141): a header whose first character is the op code, one sequence line."""
    with open(path, "wb") as f:
        for k, s in enumerate(seqs):
            f.write(b">" + str(k + 1).encode() + b"\n" + s + b"\n")


def run_cli(ref_fa, query_fa, scoring, w, z, a, n_threads, workdir, extra=()):
    raw = os.path.join(workdir, f"raw_{a}_{n_threads}.log")
    stats = os.path.join(workdir, f"stats_{a}_{n_threads}.txt")
    score = os.path.join(workdir, f"score_{a}_{n_threads}.log")
    for p in (raw, stats):
        if os.path.exists(p):
            os.remove(p)
    cmd = [MANUAL, "-p", "-m", str(scoring["m"]), "-x", str(scoring["x"]), "-q", str(scoring["q"]), "-r", str(scoring["r"]),
           "-s", "3", "-z", str(z), "-w", str(w), "-a", str(a), "-n", str(n_threads), *extra, ref_fa, query_fa, raw]
    env = dict(os.environ, AGATHA_AMD_RAW_STATS=stats)
    t0 = time.time()
    with open(score, "wb") as out:
        subprocess.check_call(cmd, stdout=out, env=env)
    wall = time.time() - t0
    ms = [float(x) for x in open(raw).read().split()]
    per_batch = [dict(zip(("pairs", "kernel_ms", "value_wave_steps", "key_wave_steps", "pairs_started_over", "went_back_to_checkpoint", "pairs_taken_over"),
                          (int(v[0]), float(v[1]), *map(int, v[2:])))) for v in (l.split() for l in open(stats)) if len(v) == 7]
    return dict(cmd=" ".join(os.path.basename(c) if c.startswith("/") else c for c in cmd), batches=len(ms), kernel_ms_per_batch=[round(x, 3) for x in ms],
                kernel_ms_sum=sum(ms), wall_s=wall, pairs_taken_over=sum(b["pairs_taken_over"] for b in per_batch), per_batch=per_batch, score_log=score)


def time_config(qs, ts, scoring, w, z, combos=((10000, 1),), keep_dir=None):
    d = keep_dir or tempfile.mkdtemp(prefix="agatha_cli_")
    ref_fa, query_fa = os.path.join(d, "ref.fasta"), os.path.join(d, "query.fasta")
    write_fasta(ref_fa, qs); write_fasta(query_fa, ts)
    out = []
    for a, n in combos:
        r = run_cli(ref_fa, query_fa, scoring, w, z, a, n, d)
        r.update(a=a, n_threads=n, pairs=len(qs))
        out.append(r)
    if keep_dir is None:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)
    return out


def time_pipeline(qs, ts, scoring, w, z, kernel_gcups=None, batch=8192, batches=16, combos=((1, False), (2, False), (4, False), (2, True), (4, True), (2, 2), (4, 2))):
    """The stream / batch manager over a sustained feed (wall time of {host fill, H2D, pack, sort, align, D2H},
    stream-overlapped; reference gasal_align.cu:144-162,254-266 + test_prog.cpp:273-375): `batches` batches of `batch` pairs --
    the first 2 * batch pairs of the batch at hand, run over again (AGATHA_AMD_REPEAT) -- through `manual` WITHOUT -p (production
    mode: nothing printed, no events), 2 storages per host thread, -n 1 / 2 / 4 host threads, host ASCII, host-packed 4-bit words (-k; host_packed true) and 2-bit codes + N mask (-K; host_packed 2).
    The loop's seconds come from the CLI itself (AGATHA_AMD_LOOP_STATS; the FASTA parse is not in them).  One more run with -p
    collects the per-batch kernel times and the pairs taken over after the time-out."""
    from agatha_amd import shard
    import numpy as np
    n_file = min(len(qs), 2 * batch)
    rep = max(1, (batch * batches) // n_file)
    d = tempfile.mkdtemp(prefix="agatha_pipe_")
    ref_fa, query_fa = os.path.join(d, "ref.fasta"), os.path.join(d, "query.fasta")
    write_fasta(ref_fa, qs[:n_file]); write_fasta(query_fa, ts[:n_file])
    cells_file = float(shard.nominal_cells(np.array([len(x) for x in qs[:n_file]]), np.array([len(x) for x in ts[:n_file]]), w).sum())
    # (the reference's parser scans options only up to argc - 4, args_parser.cpp:117: without -p the LAST option before the two
    #  file names must be one that takes a value, so the flag -k goes first)
    base = ["-m", str(scoring["m"]), "-x", str(scoring["x"]), "-q", str(scoring["q"]), "-r", str(scoring["r"]),
            "-s", "3", "-z", str(z), "-w", str(w), "-a", str(batch)]
    runs = []
    try:
        for n_threads, packed in combos:
            stats = os.path.join(d, "loop.txt")
            if os.path.exists(stats):
                os.remove(stats)
            cmd = [MANUAL] + (["-K"] if packed == 2 else ["-k"] if packed else []) + base + ["-n", str(n_threads), ref_fa, query_fa]
            env = dict(os.environ, AGATHA_AMD_REPEAT=str(rep), AGATHA_AMD_LOOP_STATS=stats)
            t0 = time.time()
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, env=env)
            wall = time.time() - t0
            sec, pairs, nb, _ = open(stats).read().split()
            sec = float(sec)
            gc = cells_file * rep / sec / 1e9
            runs.append(dict(host_threads=n_threads, host_packed=packed, storages=2 * n_threads, batches=int(nb), pairs=int(pairs),
                             loop_s=sec, process_wall_s=wall, end_to_end_gcups=gc,
                             vs_kernel_only=(gc / kernel_gcups) if kernel_gcups else None))
        # The loop's seconds hold the ramp of the FIRST batch (its host fill and H2D: ~20 ms before the first kernel starts -- 6 % of a
        # 16-batch run, profiles/r05_v1/pipeline_kernel_trace.txt) and the drain of the last one.  The steady state is what a longer feed
        # adds per batch: the best configuration once more with three times the batches, (t3 - t1) / (2 x batches).
        best0 = max(runs, key=lambda x: x["end_to_end_gcups"])
        stats = os.path.join(d, "loop.txt")
        if os.path.exists(stats):
            os.remove(stats)
        cmd = [MANUAL] + (["-K"] if best0["host_packed"] == 2 else ["-k"] if best0["host_packed"] else []) + base + ["-n", str(best0["host_threads"]), ref_fa, query_fa]
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, env=dict(os.environ, AGATHA_AMD_REPEAT=str(3 * rep), AGATHA_AMD_LOOP_STATS=stats))
        sec3 = float(open(stats).read().split()[0])
        steady = None
        if sec3 > best0["loop_s"]:
            gc_ss = cells_file * 2 * rep / (sec3 - best0["loop_s"]) / 1e9
            steady = dict(batches=3 * batches, loop_s=sec3, marginal_gcups=gc_ss, vs_kernel_only=(gc_ss / kernel_gcups) if kernel_gcups else None,
                          ramp_and_drain_ms=1e3 * (best0["loop_s"] - (sec3 - best0["loop_s"]) / 2.0))
        # -p once: per-batch kernel milliseconds + taken-over counts of a sustained run (2 threads, host ASCII)
        r = run_cli(ref_fa, query_fa, scoring, w, z, batch, 2, d)       # (one pass over the file: 2 batches per thread-pair)
        taken = r["pairs_taken_over"]
        kms = r["kernel_ms_per_batch"]
    finally:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)
    best = max(runs, key=lambda x: x["end_to_end_gcups"])
    best_k = max((x for x in runs if x["host_packed"]), key=lambda x: x["end_to_end_gcups"], default=None)
    # the share of the H2D copies that is hidden: per batch the stream carries copy + kernels; what the loop takes per batch
    # beyond the kernel time is what was not hidden (negative: the batches overlap on the GPU)
    return dict(protocol="agatha_amd/manual (no -p), AGATHA_AMD_REPEAT: %d batches of %d pairs through gasal_aln_async, 2 storages per host thread; "
                         "seconds of the batch loop (host fill + H2D + pack + sort + align + D2H, overlapped)" % (batches, batch),
                runs=runs, best_end_to_end_gcups=best["end_to_end_gcups"], best_config={k: best[k] for k in ("host_threads", "host_packed")},
                best_host_packed_vs_kernel_only=(best_k["vs_kernel_only"] if best_k else None), steady_state=steady,
                kernel_ms_per_batch_with_p=kms, pairs_taken_over_with_p=taken)


if __name__ == "__main__":
    from agatha_amd import workload
    name = sys.argv[1] if len(sys.argv) > 1 else "C1"
    gen, scoring, w, z, n0 = {"C1": (workload.cfg_c1, dict(m=2, x=4, q=4, r=2), 751, 400, 10000),
                              "C0": (workload.cfg_c0, dict(m=2, x=4, q=4, r=2), 751, 400, 20000)}[name]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else n0
    qs, ts = gen(n=n)
    for r in time_config(qs, ts, scoring, w, z, combos=((8192, 1), (8192, 2), (10000, 1), (10000, 2))):
        r.pop("per_batch"); r.pop("score_log")
        print(json.dumps(dict(config=name, **r)), flush=True)
