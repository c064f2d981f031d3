"""GPU: the GASAL-compatible C++ host layer + `manual` CLI end to end (FASTA -> batches -> score lines)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MANUAL = os.path.join(ROOT, "agatha_amd", "manual")


def write_fasta(path, seqs, width=70, header=">>> "):
    with open(path, "w") as f:
        for k, s in enumerate(seqs):
            s = s.decode() if isinstance(s, bytes) else s
            f.write(f"{header}{k + 1}\n")
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")


def parse(out):
    res = []
    for line in out.strip().splitlines():
        a, b, c = line.split("\t")
        assert b.startswith("query_batch_end=") and c.startswith("target_batch_end=")
        res.append((int(a), int(b.split("=")[1]), int(c.split("=")[1])))
    return np.array(res, np.int64)


def match_batches(got, batches):
    """Lines are in input order inside a batch; batches are printed in the order their streams are seen to finish (two
    streams per host thread, exactly as the reference's test_prog.cpp:355-374), so match batch by batch."""
    pos, unused = 0, list(range(len(batches)))
    while pos < len(got):
        hit = [k for k in unused if (got[pos:pos + len(batches[k])] == batches[k]).all()]
        assert hit, f"printed block at line {pos} matches no batch"
        pos += len(batches[hit[0]])
        unused.remove(hit[0])
    assert not unused


@pytest.mark.parametrize("threads,align_num", [(1, 8192), (1, 100), (3, 64)])
def test_cli_matches_oracle(tmp_path, threads, align_num):
    qs, ts = synth.cfg_c4(n=300, seed=99, lo=100, hi=4000)
    f1, f2, raw = tmp_path / "ref.fasta", tmp_path / "query.fasta", tmp_path / "raw.log"
    write_fasta(f1, qs, width=61)
    write_fasta(f2, ts, width=80)          # different line counts per record: the reader must not care
    cmd = [MANUAL, "-p", "-m", "1", "-x", "4", "-q", "6", "-r", "2", "-s", "3", "-z", "400", "-w", "751",
           "-a", str(align_num), "-n", str(threads), str(f1), str(f2), str(raw)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = parse(r.stdout)
    exp = np.stack(O.align_pairs(qs, ts, O.make_params(m=1, x=4, q=6, r=2, s=3, z=400, w=751), wide=True,
                                 threads=4), axis=1)
    assert got.shape == exp.shape
    # Lines are in input order inside a batch; batches are printed in the order their streams are seen to finish
    # (two streams per host thread, exactly as the reference's test_prog.cpp:355-374), so match batch by batch.
    per_thread = -(-300 // threads)
    batches = []
    for t in range(threads):
        lo, hi = t * per_thread, min(300, (t + 1) * per_thread)
        batches += [exp[k:min(k + align_num, hi)] for k in range(lo, hi, align_num)]
    match_batches(got, batches)
    lines = open(raw).read().split()
    n_batches = sum(-(-c // align_num) for c in ([300] if threads == 1 else [100, 100, 100]))
    assert len(lines) == n_batches and all(float(x) > 0 for x in lines)


def test_cli_silent_without_p(tmp_path):
    qs, ts = synth.make_pairs(3, 10, lambda r: 200)
    f1, f2 = tmp_path / "a.fa", tmp_path / "b.fa"
    write_fasta(f1, qs, header=">")
    write_fasta(f2, ts, header=">")
    r = subprocess.run([MANUAL, "-w", "100", str(f1), str(f2)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout == ""


def test_cli_op_codes_with_c_flag(tmp_path):
    """-c (extension): header characters > < / + select forward / reverse / complement / reverse-complement."""
    qs, ts = synth.make_pairs(8, 40, lambda r: int(r.integers(20, 900)))
    ops_q = [">", "<", "/", "+"] * 10
    ops_t = ["+", ">", "<", "/"] * 10
    f1, f2, raw = tmp_path / "a.fa", tmp_path / "b.fa", tmp_path / "raw.log"
    for path, seqs, ops in ((f1, qs, ops_q), (f2, ts, ops_t)):
        with open(path, "w") as f:
            for k, (s, o) in enumerate(zip(seqs, ops)):
                f.write(f"{o}{k}\n{s.decode()}\n")
    r = subprocess.run([MANUAL, "-p", "-c", "-w", "100", "-z", "200", str(f1), str(f2), str(raw)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    comp = bytes.maketrans(b"ACGT", b"TGCA")

    def tr(s, o):
        k = "></+".index(o)
        s = s[::-1] if k & 1 else s
        return s.translate(comp) if k & 2 else s
    exp = np.stack(O.align_pairs([tr(s, o) for s, o in zip(qs, ops_q)], [tr(s, o) for s, o in zip(ts, ops_t)],
                                 O.make_params(w=100, z=200), wide=True), axis=1)
    assert (parse(r.stdout) == exp).all()


def test_multi_gpu_front_end_single_rank(tmp_path):
    """agatha_amd.multi_gpu under torch.distributed.run (1 rank here; the 8-GPU form is the same code path)."""
    import sys
    qs, ts = synth.cfg_c4(n=120, seed=3, lo=100, hi=5000)
    f1, f2 = tmp_path / "ref.fasta", tmp_path / "query.fasta"
    write_fasta(f1, qs)
    write_fasta(f2, ts)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", "-m", "agatha_amd.multi_gpu", "-m", "1", "-x", "4", "-q", "6", "-r", "2", "-w", "100",
           "-z", "100", "-a", "50", str(f1), str(f2)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    exp = np.stack(O.align_pairs(qs, ts, O.make_params(m=1, x=4, q=6, r=2, s=3, z=100, w=100), wide=True, threads=4), axis=1)
    assert (parse(r.stdout) == exp).all()


REF_CLIENT = os.path.join(ROOT, "oracle", "_ref", "ref_test_prog")


@pytest.mark.skipif(not os.path.exists(REF_CLIENT), reason="oracle/_ref/ref_test_prog not built (make -C oracle ref, build container)")
def test_reference_client_and_manual_behave_alike(tmp_path):
    """Drop-in check of the GASAL boundary with the reference's OWN client: AGAThA/test_prog/test_prog.cpp, compiled in the
    build container against include/ + libgasal_amd.so (one documented edit, tests/test_abi_and_host.py), and this repo's
    `manual` print the same score.log on the C0 stand-in and the library writes one raw.log line per batch for both
    (the reference writes that line inside gasal_aln_async, gasal_align.cu:218-236)."""
    qs, ts = synth.cfg_c0(n=600)
    f1, f2 = tmp_path / "ref.fasta", tmp_path / "query.fasta"
    # the reference client reads its two files in line lock-step (test_prog.cpp:94): one line per record
    write_fasta(f1, qs, width=10 ** 9, header=">")
    write_fasta(f2, ts, width=10 ** 9, header=">")
    flags = ["-p", "-m", "1", "-x", "4", "-q", "6", "-r", "2", "-s", "3", "-z", "400", "-w", "751"]      # AGAThA.sh:44
    exp = np.stack(O.align_pairs(qs, ts, O.make_params(m=1, x=4, q=6, r=2, s=3, z=400, w=751), wide=True, threads=4), axis=1)
    for align_num, n_batches in ((8192, 1), (250, 3)):
        outs = {}
        for name, exe in (("manual", MANUAL), ("reference_client", REF_CLIENT)):
            raw = tmp_path / f"raw_{name}_{align_num}.log"
            r = subprocess.run([exe] + flags + ["-a", str(align_num), str(f1), str(f2), str(raw)], capture_output=True,
                               text=True, timeout=600)
            assert r.returncode == 0, (name, r.stderr[-2000:])
            lines = open(raw).read().split()
            assert len(lines) == n_batches and all(float(x) > 0 for x in lines), (name, lines)
            outs[name] = r.stdout
        if n_batches == 1:
            assert outs["manual"] == outs["reference_client"]                      # score.log, byte for byte
            assert (parse(outs["manual"]) == exp).all()
        else:
            for o in outs.values():
                match_batches(parse(o), [exp[k:k + align_num] for k in range(0, 600, align_num)])


def test_cli_prepacked_batches_through_the_gasal_api(tmp_path):
    """f3: `manual -k` creates its storages with isPacked (ctors.cpp:65-73), packs every sequence on the host
    (gasal_host_batch_fill_packed -> agatha_amd_pack_host) and gasal_aln_async skips the pack kernel (gasal_align.cu:174):
    half the H2D bytes, identical score lines.  Several batches, pages that grow, lengths of every residue mod 8.
    `manual -K` does the same in the 2-bit + N-mask format (gasal_host_batch_fill_packed2 -> agatha_amd_pack2_host, expanded on
    the device by agatha_amd_unpack2): 3/8 of the bytes, identical score lines, N runs included; other letters are refused."""
    qs, ts = synth.cfg_c4(n=400, seed=12, lo=50, hi=6000)
    qs = synth.add_n_runs(qs, 0.1, seed=3, lo=1, hi=40)
    ts = synth.add_n_runs(ts, 0.1, seed=4, lo=1, hi=40)
    f1, f2 = tmp_path / "ref.fasta", tmp_path / "query.fasta"
    write_fasta(f1, qs, header=">")
    write_fasta(f2, ts, header=">")
    outs = []
    for n_extra, extra in enumerate(([], ["-k"], ["-K"])):
        raw = tmp_path / ("raw%d.log" % n_extra)
        r = subprocess.run([MANUAL, "-p"] + extra + ["-m", "1", "-x", "4", "-q", "6", "-r", "2", "-z", "400", "-w", "751", "-a", "150",
                            str(f1), str(f2), str(raw)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert len(open(raw).read().split()) == 3
        outs.append(parse(r.stdout))
    exp = np.stack(O.align_pairs(qs, ts, O.make_params(m=1, x=4, q=6, r=2, s=3, z=400, w=751), wide=True, threads=4), axis=1)
    for o in outs:
        match_batches(o, [exp[k:k + 150] for k in range(0, 400, 150)])
    # ops + pre-packed together are refused, as documented
    r = subprocess.run([MANUAL, "-p", "-k", "-c", "-w", "100", str(f1), str(f2), str(tmp_path / "r.log")], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "isPacked" in r.stderr
    # a letter the 2-bit format has no code for
    qs2 = list(qs[:20]); qs2[7] = qs2[7][:10] + b"R" + qs2[7][11:]
    f3, f4 = tmp_path / "ref2.fasta", tmp_path / "query2.fasta"
    write_fasta(f3, qs2, header=">"); write_fasta(f4, ts[:20], header=">")
    r = subprocess.run([MANUAL, "-p", "-K", "-w", "100", str(f3), str(f4), str(tmp_path / "r2.log")], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "2-bit format" in r.stderr
    r = subprocess.run([MANUAL, "-p", "-k", "-w", "100", str(f3), str(f4), str(tmp_path / "r3.log")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_cli_start_positions(tmp_path):
    """f4 through the GASAL API: `manual -p -S` allocates and fills host_res->query_batch_start / target_batch_start (NULL in
    the reference, res.cpp:27-28) and prints them behind the ends."""
    rng = np.random.default_rng(2)
    qs, ts = synth.make_pairs(9, 120, lambda r: int(r.integers(30, 3000)))
    qs = [synth.random_seq(rng, int(rng.integers(1, 30))).tobytes() + q if k % 2 else q for k, q in enumerate(qs)]
    f1, f2, raw = tmp_path / "a.fa", tmp_path / "b.fa", tmp_path / "raw.log"
    write_fasta(f1, qs, header=">")
    write_fasta(f2, ts, header=">")
    r = subprocess.run([MANUAL, "-p", "-S", "-w", "200", "-z", "300", "-a", "50", str(f1), str(f2), str(raw)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = []
    for line in r.stdout.strip().splitlines():
        f = line.split("\t")
        assert [x.split("=")[0] for x in f[1:]] == ["query_batch_end", "target_batch_end", "query_batch_start", "target_batch_start"]
        rows.append([int(f[0])] + [int(x.split("=")[1]) for x in f[1:]])
    got = np.array(rows, np.int64)
    P = O.make_params(w=200, z=300)
    es, eq, et = O.align_pairs(qs, ts, P, wide=True, threads=4)
    xq, xt, _ = O.start_positions(qs, ts, P, eq, et, threads=4)
    exp = np.stack([es, eq, et, xq, xt], axis=1)
    match_batches(got, [exp[k:k + 50] for k in range(0, 120, 50)])
    assert (exp[:, 3] > 0).sum() > 30


def cigar_text(c):
    """CIGAR text of a path in the byte format of include/agatha_amd.h (runs the format split are merged again)."""
    if c is None:
        return "!"
    if not c:
        return "*"
    out, run, op = [], 0, c[0] & 3
    for b in c:
        if (b & 3) != op:
            out.append(f"{run}{'=XDI'[op]}")
            run, op = 0, b & 3
        run += b >> 2
    out.append(f"{run}{'=XDI'[op]}")
    return "".join(out)


def test_cli_traceback(tmp_path):
    """f4 through the GASAL API: `manual -p -T` fills host_res->cigar / n_cigar_ops (declared and left NULL by the reference,
    gasal.h:91-92) and prints every path as CIGAR text; three batches per stream, so the arrays are reused."""
    qs, ts = synth.make_pairs(19, 130, lambda r: int(r.integers(1, 2500)))
    qs[5], ts[5] = b"ACGT" * 25, b"TGCA" * 25                       # empty alignment: "*"
    f1, f2, raw = tmp_path / "a.fa", tmp_path / "b.fa", tmp_path / "raw.log"
    write_fasta(f1, qs, header=">")
    write_fasta(f2, ts, header=">")
    r = subprocess.run([MANUAL, "-p", "-T", "-w", "120", "-z", "300", "-a", "25", str(f1), str(f2), str(raw)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    P = O.make_params(w=120, z=300)
    es, eq, et, cig = O.traceback_pairs(qs, ts, P, threads=4)
    exp = [f"{s}\tquery_batch_end={a}\ttarget_batch_end={b}\tcigar={cigar_text(c)}" for s, a, b, c in zip(es, eq, et, cig)]
    got = r.stdout.strip().splitlines()
    assert len(got) == len(exp)
    blocks, unused = [exp[k:k + 25] for k in range(0, 130, 25)], list(range(6))
    pos = 0
    while pos < len(got):                                           # batches print in the order their streams finish
        hit = [k for k in unused if got[pos:pos + len(blocks[k])] == blocks[k]]
        assert hit, (pos, got[pos], )
        pos += len(blocks[hit[0]])
        unused.remove(hit[0])
    assert sum("D" in e or "I" in e for e in exp) > 60 and any(e.endswith("cigar=*") for e in exp)


def test_cli_batch_larger_than_one_round_of_lane_groups(tmp_path):
    """`manual -a 12000` on 11 000 short pairs at a narrow band: one GASAL batch holds more pairs than the int16 kernel has
    lane groups (8192), so the C++ layer's workspace carries the areas of the preemptive schedule and pairs are suspended
    and resumed across lane groups behind gasal_aln_async -- same score lines as the oracle, also with start positions."""
    qs, ts = synth.make_pairs(21, 11000, lambda r: int(r.integers(60, 700)), 0.03, 0.03, 0.04)
    f1, f2, raw = tmp_path / "a.fa", tmp_path / "b.fa", tmp_path / "raw.log"
    write_fasta(f1, qs, header=">", width=10 ** 9)
    write_fasta(f2, ts, header=">", width=10 ** 9)
    P = O.make_params(w=24, z=100)
    es, eq, et = O.align_pairs(qs, ts, P, wide=True, threads=8)
    r = subprocess.run([MANUAL, "-p", "-w", "24", "-z", "100", "-a", "12000", str(f1), str(f2), str(raw)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (parse(r.stdout) == np.stack([es, eq, et], axis=1)).all()
    assert len(open(raw).read().split()) == 1


@pytest.mark.parametrize("threads,gflag", [(1, []), (2, ["-g", "1"])])
def test_cli_two_storages_with_batches_larger_than_one_round(tmp_path, threads, gflag):
    """The CLI's default shape: two streams (storages) per host thread, each holding a batch of more pairs than the int16
    kernel has lane groups, so two persistent full-chip grids on static schedules are enqueued back to back.  The second
    grid's workgroups only become resident as the first one's leave; a lane group that waited 50 ms for a neighbour that is
    not resident would take the pair over (MIG_STOLEN) -- correct, but a cliff: the statistics line of every batch
    (AGATHA_AMD_RAW_STATS) must show none.  Also run as `-g 1 -n 2` (device selection through gasal_set_device, two host
    threads = four storages)."""
    n, a = 9500 * 2 * threads, 9500
    qs, ts = synth.make_pairs(33, n, lambda r: int(r.integers(300, 900)), 0.03, 0.03, 0.04)
    f1, f2, raw, stats = tmp_path / "a.fa", tmp_path / "b.fa", tmp_path / "raw.log", tmp_path / "stats.txt"
    write_fasta(f1, qs, header=">", width=10 ** 9)
    write_fasta(f2, ts, header=">", width=10 ** 9)
    P = O.make_params(w=200, z=100)
    exp = np.stack(O.align_pairs(qs, ts, P, wide=True, threads=8), axis=1)
    r = subprocess.run([MANUAL, "-p", "-w", "200", "-z", "100", "-a", str(a), "-n", str(threads)] + gflag + [str(f1), str(f2), str(raw)],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, AGATHA_AMD_RAW_STATS=str(stats), AGATHA_AMD_FORCE_INT16="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    match_batches(parse(r.stdout), [exp[k:k + a] for k in range(0, n, a)])
    lines = [l.split() for l in open(stats)]
    assert len(lines) == 2 * threads == len(open(raw).read().split())
    assert all(int(l[0]) == a for l in lines)
    assert sum(int(l[6]) for l in lines) == 0, lines            # no pair was taken over after a time-out
    assert all(int(l[2]) + int(l[3]) > 0 for l in lines), lines         # and the batches ran on the int16 kernel


def test_cli_start_positions_of_ultra_long_reads(tmp_path):
    """`manual -S` on reads of BASELINE's ultra-long shape (~130 kb, band 1500): the backward pass runs without z-drop, and
    must not arm the range check that z < 0 implies for scores that sink without bound (round 2 refused such batches with
    AGATHA_AMD_ERANGE although the forward pass had accepted them); also a packed (-k) batch whose sequences overflow the first
    host page (capacity of packed storages is accounted in packed bytes)."""
    rng = np.random.default_rng(5)
    qs, ts = [], []
    for L in (131000, 128500, 3000):
        ref = synth.random_seq(rng, L)
        qs.append(synth.random_seq(rng, 17).tobytes() + ref.tobytes())          # the alignment starts behind the origin
        ts.append(synth.mutate(rng, ref, 0.03, 0.03, 0.04).tobytes())
    f1, f2, raw = tmp_path / "a.fa", tmp_path / "b.fa", tmp_path / "raw.log"
    write_fasta(f1, qs, header=">", width=10 ** 9)
    write_fasta(f2, ts, header=">", width=10 ** 9)
    P = O.make_params(w=1500, z=400)
    es, eq, et = O.align_pairs(qs, ts, P, wide=True, threads=4)
    xq, xt, _ = O.start_positions(qs, ts, P, eq, et, threads=4)
    exp = np.stack([es, eq, et, xq, xt], axis=1)
    for extra in ([], ["-k"]):
        r = subprocess.run([MANUAL, "-p", "-S", "-w", "1500", "-z", "400"] + extra + [str(f1), str(f2), str(raw)], capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        got = np.array([[int(f[0])] + [int(x.split("=")[1]) for x in f[1:]] for f in (l.split("\t") for l in r.stdout.strip().splitlines())], np.int64)
        assert (got == exp).all()
    assert exp[0, 0] > 100000


@pytest.mark.parametrize("threads,packed", [(1, False), (2, True), (2, 2)])
def test_cli_sustained_feed_of_sixteen_batches(tmp_path, threads, packed):
    """The stream / batch manager behind gasal_aln_async over a sustained feed ({host fill, H2D, pack, sort, align,
    D2H} stream-overlapped; reference gasal_align.cu:144-162,254-266 + test_prog.cpp:273-375): the pairs of the two files run eight
    times over (AGATHA_AMD_REPEAT), sixteen batches of 9 000 pairs through two storages per host thread, host ASCII, host-packed
    (-k) and in the 2-bit + N-mask format (-K; packed = 2).  Every batch prints the oracle's lines; no batch shows a pair taken over after the 50 ms time-out (each batch is a
    persistent full-chip grid on a static schedule enqueued behind the other storage's); the CLI reports the seconds of its batch
    loop (AGATHA_AMD_LOOP_STATS), which is what bench.py's `gasal_api.pipeline` turns into end-to-end GCUPS."""
    a, rep = 9000, 8
    n = 2 * a
    qs, ts = synth.make_pairs(41, n, lambda r: int(r.integers(300, 900)), 0.03, 0.03, 0.04)
    f1, f2, raw, stats, loop = (tmp_path / x for x in ("a.fa", "b.fa", "raw.log", "stats.txt", "loop.txt"))
    write_fasta(f1, qs, header=">", width=10 ** 9)
    write_fasta(f2, ts, header=">", width=10 ** 9)
    P = O.make_params(w=200, z=100)
    exp = np.stack(O.align_pairs(qs, ts, P, wide=True, threads=8), axis=1)
    cmd = [MANUAL] + (["-K"] if packed == 2 else ["-k"] if packed else []) + ["-p", "-w", "200", "-z", "100", "-a", str(a), "-n", str(threads), str(f1), str(f2), str(raw)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, AGATHA_AMD_RAW_STATS=str(stats), AGATHA_AMD_LOOP_STATS=str(loop), AGATHA_AMD_REPEAT=str(rep),
                                AGATHA_AMD_FORCE_INT16="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    got = parse(r.stdout)
    assert len(got) == n * rep
    # the virtual file is the two real batches over and over: thread t starts at pair t * (n * rep / threads), a multiple of a
    match_batches(got, [exp[(k % n):(k % n) + a] for k in range(0, n * rep, a)])
    lines = [l.split() for l in open(stats)]
    assert len(lines) == 2 * rep and all(int(l[0]) == a for l in lines)
    assert sum(int(l[6]) for l in lines) == 0, lines            # no pair was taken over after a time-out
    sec, pairs, batches, nthr = open(loop).read().split()
    assert int(pairs) == n * rep and int(batches) == 2 * rep and int(nthr) == threads and 0 < float(sec) < 60
