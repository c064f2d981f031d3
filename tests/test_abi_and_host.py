"""CPU-only: the C-ABI library loads and exports every symbol include/agatha_amd.h declares; the C++ host layer
(GASAL API + CLI) builds, links and honours the reference's argument contract.  No compute calls (no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "agatha_amd", "libagatha_amd.so")
MANUAL = os.path.join(ROOT, "agatha_amd", "manual")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not (os.path.exists(LIB) and os.path.exists(MANUAL)):
        import __graft_entry__
        __graft_entry__.build()


def test_every_declared_symbol_is_exported():
    import agatha_amd
    lib = agatha_amd.load_library()
    hdr = open(os.path.join(ROOT, "include", "agatha_amd.h")).read()
    names = set(re.findall(r"\b(agatha_amd_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    from agatha_amd import engine
    assert set(engine.EXPORTS) == names


def test_abi_struct_matches_reference_layout():
    import ctypes as C
    import agatha_amd
    # seven int32 in the order of the reference's gasal_subst_scores (gasal.h:165-173)
    assert [f[0] for f in agatha_amd.Scores._fields_] == ["match", "mismatch", "gap_open", "gap_extend", "slice_width",
                                                           "z_threshold", "band_width"]
    assert C.sizeof(agatha_amd.Scores) == 28


def test_library_reports_limits_without_a_gpu():
    import agatha_amd
    lib = agatha_amd.load_library()
    assert lib.agatha_amd_max_band() >= 1500          # BASELINE config 3 needs band 1500
    assert lib.agatha_amd_workspace_bytes(4096) < (1 << 20)
    # batches that can exceed one round of lane groups also carry the areas of the preemptive schedule (suspended pairs)
    assert lib.agatha_amd_workspace_bytes(8192) < (160 << 20)         # (two states per lane-group boundary since round 4) the reference: 0.98 GB per stream (ctors.cpp:89)
    assert lib.agatha_amd_strerror(-2).decode().startswith("band")


def test_engine_fails_loudly_without_gpu():
    import agatha_amd
    if agatha_amd.load_library().agatha_amd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(agatha_amd.AgathaError):
        agatha_amd.Engine(0)


def test_gasal_host_library_exports_the_reference_api():
    out = subprocess.check_output(["nm", "-DC", os.path.join(ROOT, "agatha_amd", "libgasal_amd.so")], text=True)
    for sym in ["gasal_copy_subst_scores", "gasal_init_gpu_storage_v", "gasal_init_streams", "gasal_host_batch_fill",
                "gasal_host_alns_resize", "gasal_op_fill", "gasal_aln_async", "gasal_is_aln_async_done",
                "gasal_destroy_streams", "gasal_destroy_gpu_storage_v", "gasal_set_device", "gasal_host_batch_reset",
                "gasal_res_new_host", "Parameters::parse", "gasal_host_batch_fill_packed(", "gasal_host_batch_fill_packed2("]:
        assert sym in out, sym


def test_cli_argument_contract(tmp_path):
    # --help anywhere prints usage and exits 0 (args_parser.cpp:101-110)
    r = subprocess.run([MANUAL, "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage" in r.stderr
    # argc < 4 is refused (args_parser.cpp:112): a bare `manual q.fa t.fa` fails
    r = subprocess.run([MANUAL, "a.fa", "b.fa"], capture_output=True, text=True)
    assert r.returncode == 1 and "Not enough" in r.stderr
    # missing files -> WRONG_FILES, exit 1
    r = subprocess.run([MANUAL, "-m", "2", str(tmp_path / "nope1.fa"), str(tmp_path / "nope2.fa")], capture_output=True, text=True)
    assert r.returncode == 1 and "File error" in r.stderr
    # multi-letter option packs are refused
    r = subprocess.run([MANUAL, "-sp", "1", str(tmp_path / "a"), str(tmp_path / "b")], capture_output=True, text=True)
    assert r.returncode == 1 and "Wrong argument" in r.stderr


def test_avg_time_contract(tmp_path):
    """Post-processing of the driver script (reference misc/avg_time.py:14-44): sum of the raw log / iterations, merged
    into the JSON; "NaN" for a missing or empty raw file."""
    import json
    from agatha_amd import avg_time
    raw = tmp_path / "raw.log"
    out = tmp_path / "time.json"
    raw.write_text("1.5\n2.5\n4\n")
    assert avg_time.main(["AGAThA", "test", str(raw), str(out), "2"]) == 0
    assert json.loads(out.read_text()) == {"AGAThA": {"test": 4.0}}
    (tmp_path / "empty.log").write_text("")
    avg_time.main(["AGAThA", "empty", str(tmp_path / "empty.log"), str(out), "3"])
    avg_time.main(["other", "gone", str(tmp_path / "missing.log"), str(out), "3"])
    assert json.loads(out.read_text()) == {"AGAThA": {"test": 4.0, "empty": "NaN"}, "other": {"gone": "NaN"}}


REF_CLIENT_SRC = "/root/reference/AGAThA/test_prog/test_prog.cpp"


@pytest.mark.skipif(not os.path.exists(REF_CLIENT_SRC), reason="reference tree absent (build container only)")
def test_reference_client_builds_against_this_boundary(tmp_path):
    """The reference's OWN client (AGAThA/test_prog/test_prog.cpp) compiles against include/ and links libgasal_amd.so
    with exactly ONE documented edit: its line 25, a bare cudaDeviceSynchronize() -- a CUDA
    runtime name the reference's gasal.h:8 drags in and this library does not provide -- is dropped.  Unedited it must
    fail on that name and on nothing else.  (`make -C oracle ref` builds the same binary into oracle/_ref/ for the GPU
    half of this check, tests/test_gpu_cli.py.)"""
    src = open(REF_CLIENT_SRC).read()
    assert src.count("cudaDeviceSynchronize();") == 1 and "cuda" not in src.replace("cudaDeviceSynchronize();", "").lower()
    base = ["g++", "-O1", "-std=c++17", "-fopenmp", "-w", "-I" + os.path.join(ROOT, "include"),
            "-I" + os.path.dirname(REF_CLIENT_SRC)]
    link = ["-L" + os.path.join(ROOT, "agatha_amd"), "-lgasal_amd", "-lagatha_amd",
            "-Wl,-rpath," + os.path.join(ROOT, "agatha_amd")]
    # (a) unchanged: one error, the CUDA runtime name
    r = subprocess.run(base + ["-fsyntax-only", REF_CLIENT_SRC], capture_output=True, text=True)
    errors = [l for l in r.stderr.splitlines() if "error:" in l]
    assert r.returncode != 0 and len(errors) == 1 and "cudaDeviceSynchronize" in errors[0], r.stderr
    # (b) with the documented edit: compiles, links, and honours the argument contract without a GPU
    edited = tmp_path / "ref_test_prog.cpp"
    edited.write_text("\n".join(l for l in src.split("\n") if l.strip() != "cudaDeviceSynchronize();"))
    exe = tmp_path / "ref_test_prog"
    r = subprocess.run(base + ["-o", str(exe), str(edited)] + link, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), "a.fa", "b.fa"], capture_output=True, text=True)
    assert r.returncode == 1 and "Not enough" in r.stderr
    edited.unlink()          # no reference text stays behind, not even under pytest's tmp dir


def test_host_packer_matches_the_pack_layout():
    """agatha_amd_pack_host (AVX2): the reference's 4-bit layout (pack_rc_seqs.h:21-33), checked against the oracle's
    restatement for every tail length of the vector loop, any letter, both cases."""
    import numpy as np
    import agatha_amd
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    letters = np.frombuffer(b"ACGTNacgtnRYKM", np.uint8)
    for n in [8 * k for k in range(1, 20)] + [4096, 100008]:
        a = letters[rng.integers(0, letters.size, n)].copy()
        assert (agatha_amd.pack_host(a) == O.pack(a)).all(), n


def test_two_bit_host_packer_round_trips_through_the_four_bit_layout():
    """agatha_amd_pack2_host (2-bit codes + N mask, 3 bits per base: north_star's "2-bit-packed" input): unpacking its output on
    the CPU by the rule the device kernel uses gives the 4-bit words of the ordinary packing (pack_rc_seqs.h:21-33) wherever the
    letters are ACGT or N in any case; every other letter becomes N and is counted."""
    import numpy as np
    import agatha_amd
    rng = np.random.default_rng(8)
    a = rng.choice(np.frombuffer(b"ACGTNacgtn", np.uint8), 8 * 1001).astype(np.uint8)
    codes, nmask, other = agatha_amd.pack2_host(a)
    assert other == 0 and codes.dtype == np.uint16 and nmask.dtype == np.uint8 and codes.size == nmask.size == 1001
    lut = np.array([1, 3, 7, 4], np.uint32)
    words = np.zeros(1001, np.uint32)
    for k in range(8):
        c = (codes.astype(np.uint32) >> (14 - 2 * k)) & 3
        nib = np.where((nmask >> (7 - k)) & 1, 14, lut[c]).astype(np.uint32)
        words |= nib << np.uint32(28 - 4 * k)
    assert (words == agatha_amd.pack_host(a)).all()
    b = a.copy(); b[5] = ord("R"); b[77] = ord("q"); b[78] = ord("-")          # ('q' & 15 = 1 like 'A': the letter itself is looked at)
    c2, m2, other2 = agatha_amd.pack2_host(b)
    assert other2 == 3 and (m2[0] >> (7 - 5)) & 1 and (m2[9] >> (7 - 5)) & 1 and (m2[9] >> (7 - 6)) & 1
