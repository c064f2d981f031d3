"""ctypes binding of include/agatha_amd.h (the C-ABI of libagatha_amd.so).

Mirrors the reference's host flow for one batch (AGAThA/src/gasal_align.cu:27-273):
H2D of the unpacked ASCII batch -> pack -> (sort) -> align -> D2H of three int32 result arrays.
There is deliberately no CPU path here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBNAME = "libagatha_amd.so"


class AgathaError(RuntimeError):
    pass


class Scores(C.Structure):
    """agatha_amd_scores == the reference's gasal_subst_scores (AGAThA/src/gasal.h:165-173)."""
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap_open", C.c_int32),
                ("gap_extend", C.c_int32), ("slice_width", C.c_int32), ("z_threshold", C.c_int32),
                ("band_width", C.c_int32)]

    @classmethod
    def make(cls, m=2, x=4, q=4, r=2, s=3, z=400, w=751):
        """Defaults of the reference CLI (args_parser.cpp:12-22)."""
        return cls(m, x, q, r, s, z, w)


def library_path():
    return os.environ.get("AGATHA_AMD_LIB") or os.path.join(_HERE, _LIBNAME)     # override: developer A/B builds only


def build_library(force=False):
    """Compile the HIP library in-tree with hipcc (cross-compiles for gfx950 without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "-s", "clean"])
    subprocess.check_call(["make", "-C", src, "-s", "-j8", "all"])
    return library_path()


_lib = None


def load_library():
    """Load libagatha_amd.so; raises AgathaError (never falls back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise AgathaError(f"{path} is missing: build it with `make -C agatha_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(path)
    vp, u32p, i32p = C.c_void_p, C.c_void_p, C.c_void_p
    lib.agatha_amd_strerror.restype = C.c_char_p
    lib.agatha_amd_strerror.argtypes = [C.c_int]
    lib.agatha_amd_last_error.restype = C.c_char_p
    lib.agatha_amd_version.restype = C.c_char_p
    lib.agatha_amd_device_count.restype = C.c_int
    lib.agatha_amd_set_device.argtypes = [C.c_int]
    lib.agatha_amd_max_band.restype = C.c_int
    lib.agatha_amd_workspace_bytes.restype = C.c_size_t
    lib.agatha_amd_workspace_bytes.argtypes = [C.c_uint32]
    lib.agatha_amd_workspace_bytes_long.restype = C.c_size_t
    lib.agatha_amd_workspace_bytes_long.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    lib.agatha_amd_pack.argtypes = [vp, vp, C.c_uint32, u32p]
    lib.agatha_amd_pack_host.argtypes = [vp, C.c_size_t, u32p]
    lib.agatha_amd_seq_ops.argtypes = [vp, vp, u32p, u32p, u32p, vp, C.c_uint32]
    lib.agatha_amd_align.argtypes = [vp, u32p, u32p, u32p, u32p, u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.POINTER(Scores), i32p, i32p, i32p, vp, C.c_size_t]
    lib.agatha_amd_starts_scratch_bytes.restype = C.c_size_t
    lib.agatha_amd_starts_scratch_bytes.argtypes = [C.c_uint32] * 3
    lib.agatha_amd_align_starts.argtypes = [vp, u32p, u32p, u32p, u32p] + [C.c_uint32] * 5 + [C.POINTER(Scores)] + [i32p] * 4 + \
        [vp, C.c_size_t, vp, C.c_size_t]
    lib.agatha_amd_traceback_pair_bytes.restype = C.c_size_t
    lib.agatha_amd_traceback_pair_bytes.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(Scores)]
    lib.agatha_amd_traceback_scratch_bytes.restype = C.c_size_t
    lib.agatha_amd_traceback_scratch_bytes.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Scores), C.c_uint32]
    lib.agatha_amd_align_traceback.argtypes = [vp, u32p, u32p, u32p, u32p, u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.POINTER(Scores), i32p, i32p, i32p, vp, u32p, vp, C.c_size_t, vp, C.c_size_t]
    lib.agatha_amd_set_kernel_events.argtypes = [vp, vp]
    lib.agatha_amd_set_kernel_events.restype = None
    lib.agatha_amd_last_config.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.agatha_amd_last_config.restype = None
    lib.agatha_amd_last_int16_config.restype = C.c_int
    lib.agatha_amd_set_debug_option.argtypes = [C.c_char_p, C.c_int]
    lib.agatha_amd_get_debug_option.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    lib.agatha_amd_pair_kinds.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.agatha_amd_kernel_choice.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_int)]
    lib.agatha_amd_timeline.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32]
    lib.agatha_amd_schedule_info.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_int)]
    lib.agatha_amd_split_info.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_int)]
    lib.agatha_amd_pack2_host.argtypes = [vp, C.c_size_t, vp, vp]
    lib.agatha_amd_pack2_host.restype = C.c_long
    lib.agatha_amd_unpack2.argtypes = [vp, vp, vp, C.c_uint32, vp]
    lib.agatha_amd_step_stats.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint)]
    if hasattr(lib, "agatha_amd_flat_stats"):        # (developer A/B runs load older builds through AGATHA_AMD_LIB)
        lib.agatha_amd_flat_stats.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint)]
    if hasattr(lib, "agatha_amd_guard_stats"):
        lib.agatha_amd_guard_stats.argtypes = [vp, vp, C.c_uint32, vp, C.c_int]
    lib.agatha_amd_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
    lib.agatha_amd_free.argtypes = [vp]
    lib.agatha_amd_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    lib.agatha_amd_host_free.argtypes = [vp]
    lib.agatha_amd_memcpy_h2d_async.argtypes = [vp, vp, vp, C.c_size_t]
    lib.agatha_amd_memcpy_d2h_async.argtypes = [vp, vp, vp, C.c_size_t]
    lib.agatha_amd_stream_create.argtypes = [C.POINTER(vp)]
    lib.agatha_amd_stream_destroy.argtypes = [vp]
    lib.agatha_amd_stream_synchronize.argtypes = [vp]
    lib.agatha_amd_stream_query.argtypes = [vp]
    lib.agatha_amd_event_create.argtypes = [C.POINTER(vp)]
    lib.agatha_amd_event_destroy.argtypes = [vp]
    lib.agatha_amd_event_record.argtypes = [vp, vp]
    lib.agatha_amd_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    _lib = lib
    return lib


EXPORTS = [
    "agatha_amd_strerror", "agatha_amd_last_error", "agatha_amd_version", "agatha_amd_device_count",
    "agatha_amd_set_device", "agatha_amd_max_band", "agatha_amd_workspace_bytes", "agatha_amd_workspace_bytes_long", "agatha_amd_pack", "agatha_amd_pack_host",
    "agatha_amd_seq_ops", "agatha_amd_align", "agatha_amd_starts_scratch_bytes", "agatha_amd_align_starts", "agatha_amd_traceback_pair_bytes",
    "agatha_amd_traceback_scratch_bytes",
    "agatha_amd_align_traceback", "agatha_amd_set_debug_option", "agatha_amd_get_debug_option", "agatha_amd_set_kernel_events", "agatha_amd_last_config", "agatha_amd_last_int16_config", "agatha_amd_pair_kinds", "agatha_amd_kernel_choice", "agatha_amd_schedule_info", "agatha_amd_split_info", "agatha_amd_pack2_host", "agatha_amd_unpack2", "agatha_amd_step_stats", "agatha_amd_flat_stats", "agatha_amd_guard_stats", "agatha_amd_timeline", "agatha_amd_malloc", "agatha_amd_free",
    "agatha_amd_host_alloc", "agatha_amd_host_free", "agatha_amd_memcpy_h2d_async",
    "agatha_amd_memcpy_d2h_async", "agatha_amd_stream_create", "agatha_amd_stream_destroy",
    "agatha_amd_stream_synchronize", "agatha_amd_stream_query", "agatha_amd_event_create",
    "agatha_amd_event_destroy", "agatha_amd_event_record", "agatha_amd_event_elapsed_ms", "agatha_amd_stream_wait_event", "agatha_amd_get_device",
]


def pack_host(unpacked):
    """ASCII host batch (multiple of 8 bytes) -> packed words, on the host (agatha_amd_pack_host: AVX2)."""
    lib = load_library()
    u = np.ascontiguousarray(unpacked, np.uint8)
    out = np.empty(u.size // 8, np.uint32)
    _chk(lib, lib.agatha_amd_pack_host(u.ctypes.data, u.size, out.ctypes.data))
    return out


def pack2_host(unpacked):
    """2-bit codes + N mask of a padded ASCII batch (agatha_amd_pack2_host): -> (codes uint16[n/8], nmask uint8[n/8], letters that
    were neither ACGT nor N and became N)."""
    lib = load_library()
    u = np.ascontiguousarray(unpacked, np.uint8)
    if u.size % 8:
        raise AgathaError("the batch must be padded to a multiple of 8 bytes")
    codes, nmask = np.zeros(u.size // 8, np.uint16), np.zeros(u.size // 8, np.uint8)
    other = lib.agatha_amd_pack2_host(u.ctypes.data, u.size, codes.ctypes.data, nmask.ctypes.data)
    if other < 0:
        raise AgathaError(f"agatha_amd_pack2_host: {other}")
    return codes, nmask, int(other)


def set_debug_option(name, value):
    """Routing knobs of agatha_amd_align for tests and A/B runs (include/agatha_amd.h: agatha_amd_set_debug_option)."""
    lib = load_library()
    _chk(lib, lib.agatha_amd_set_debug_option(name.encode(), int(value)))


def get_debug_option(name):
    lib = load_library()
    v = C.c_int(0)
    _chk(lib, lib.agatha_amd_get_debug_option(name.encode(), C.byref(v)))
    return int(v.value)


class debug_options:
    """with debug_options(force_int16=1): ...   -- sets the options, restores the previous values on exit."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: get_debug_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_debug_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_debug_option(k, v)
        return False


def _chk(lib, rc):
    if rc != 0:
        raise AgathaError(f"{lib.agatha_amd_strerror(rc).decode()} ({rc}): {lib.agatha_amd_last_error().decode()}")


class _DevBuf:
    def __init__(self, lib, nbytes):
        self.lib, self.nbytes = lib, int(nbytes)
        p = C.c_void_p()
        _chk(lib, lib.agatha_amd_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    def free(self):
        if self.ptr and not getattr(self, "external", False):
            self.lib.agatha_amd_free(self.ptr)
        self.ptr = None


class DeviceBatch:
    """One batch resident in HBM: unpacked + packed sequences, the four metadata arrays, the three result
    arrays and the workspace -- the device half of the reference's gasal_gpu_storage_t (gasal.h:97-155)."""

    def __init__(self, eng, qbuf, tbuf, qoff, toff, qlen, tlen):
        lib = eng.lib
        self.eng = eng
        self.n = int(len(qlen))
        self.host = [np.ascontiguousarray(qbuf, np.uint8), np.ascontiguousarray(tbuf, np.uint8)] + \
                    [np.ascontiguousarray(a, np.uint32) for a in (qoff, toff, qlen, tlen)]
        qb, tb = self.host[0], self.host[1]
        if qb.size % 8 or tb.size % 8 or qb.size == 0 or tb.size == 0 or self.n == 0:
            raise AgathaError("batch bytes must be non-zero multiples of 8 and n_alns > 0 "
                              "(reference gasal_align.cu:33-53)")
        self.qbytes, self.tbytes = qb.size, tb.size
        self.max_qlen = int(self.host[4].max())
        self.max_tlen = int(self.host[5].max())
        self.d_unp_q = _DevBuf(lib, qb.size + 16)
        self.d_unp_t = _DevBuf(lib, tb.size + 16)
        self.d_pk_q = _DevBuf(lib, qb.size // 2 + 16)
        self.d_pk_t = _DevBuf(lib, tb.size // 2 + 16)
        self.d_meta = [_DevBuf(lib, 4 * self.n) for _ in range(4)]
        self.d_res = [_DevBuf(lib, 4 * self.n) for _ in range(3)]
        self.ws_bytes = lib.agatha_amd_workspace_bytes_long(self.n, max(self.max_qlen, 1), max(self.max_tlen, 1))
        self.d_ws = _DevBuf(lib, self.ws_bytes)
        # results land in pinned host memory (as the reference's cudaHostAlloc'd host_res, ctors.cpp): the D2H copy is
        # then really asynchronous and the host can enqueue the next batch while this one runs
        hp = C.c_void_p()
        _chk(lib, lib.agatha_amd_host_alloc(C.byref(hp), 12 * self.n + 16))
        self._res_pinned = hp.value
        # (behind the results: the four guard counters of the int16 kernel, agatha_amd_guard_stats -- copied with every download, looked at
        #  when the stream is next waited for)
        self._guard = np.ctypeslib.as_array((C.c_uint32 * 4).from_address(hp.value + 12 * self.n))
        self._guard[:] = 0
        self._guard_pending = False
        self.res_host = np.ctypeslib.as_array((C.c_int32 * (3 * self.n)).from_address(hp.value)).reshape(3, self.n)
        self.res_host[:] = 0

    def use_result_pointers(self, ptrs):
        """Write results into caller-owned device arrays (e.g. torch tensors used for the RCCL gather)."""
        for b in self.d_res:
            b.free()
        self.d_res = []
        for p in ptrs:
            d = _DevBuf.__new__(_DevBuf)
            d.lib, d.nbytes, d.ptr, d.external = self.eng.lib, 4 * self.n, int(p), True
            self.d_res.append(d)

    def upload(self, stream=None):
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        bufs = [self.d_unp_q, self.d_unp_t] + self.d_meta
        for d, h in zip(bufs, self.host):
            _chk(lib, lib.agatha_amd_memcpy_h2d_async(st, d.ptr, h.ctypes.data, h.nbytes))

    def upload_packed(self, packed_q, packed_t, stream=None):
        """Pre-packed input (the reference's `isPacked` batches, ctors.cpp:65-73): 4-bit codes, 8 bases per uint32, first
        base in bits 31-28, every sequence starting at word offset byte_offset / 8.  Skips the unpacked upload and the
        pack kernel; reverse/complement ops are not available on such a batch (they work on the unpacked bytes)."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        pq = np.ascontiguousarray(packed_q, np.uint32)
        pt = np.ascontiguousarray(packed_t, np.uint32)
        if pq.size * 8 != self.qbytes or pt.size * 8 != self.tbytes:
            raise AgathaError("packed batches must hold one uint32 per 8 bytes of the unpacked batch layout")
        self._packed_host = (pq, pt)          # keep alive until the copies have run
        _chk(lib, lib.agatha_amd_memcpy_h2d_async(st, self.d_pk_q.ptr, pq.ctypes.data, pq.nbytes))
        _chk(lib, lib.agatha_amd_memcpy_h2d_async(st, self.d_pk_t.ptr, pt.ctypes.data, pt.nbytes))
        for d, h in zip(self.d_meta, self.host[2:]):
            _chk(lib, lib.agatha_amd_memcpy_h2d_async(st, d.ptr, h.ctypes.data, h.nbytes))

    def upload_packed2(self, codes_q, nmask_q, codes_t, nmask_t, stream=None):
        """2-bit codes + N mask (agatha_amd_pack2_host: one uint16 + one byte per 8 bases of the padded batch layout; 3 bits per
        base over PCIe): the two arrays of each side go to the (otherwise unused) unpacked device buffer and agatha_amd_unpack2
        turns them into the 4-bit words the kernels read, in the place of pack().  ACGT + N only."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        keep = []
        for codes, nmask, nbytes, d_unp, d_pk in ((codes_q, nmask_q, self.qbytes, self.d_unp_q, self.d_pk_q), (codes_t, nmask_t, self.tbytes, self.d_unp_t, self.d_pk_t)):
            c = np.ascontiguousarray(codes, np.uint16); m = np.ascontiguousarray(nmask, np.uint8)
            if c.size * 8 != nbytes or m.size * 8 != nbytes:
                raise AgathaError("2-bit batches must hold one uint16 and one mask byte per 8 bytes of the unpacked batch layout")
            keep += [c, m]
            off_m = (c.nbytes + 255) // 256 * 256           # codes at the start of the unpacked buffer, the mask behind them (3/8 of its size)
            _chk(lib, lib.agatha_amd_memcpy_h2d_async(st, d_unp.ptr, c.ctypes.data, c.nbytes))
            _chk(lib, lib.agatha_amd_memcpy_h2d_async(st, d_unp.ptr + off_m, m.ctypes.data, m.nbytes))
            _chk(lib, lib.agatha_amd_unpack2(st, d_unp.ptr, d_unp.ptr + off_m, nbytes, d_pk.ptr))
        self._packed_host = tuple(keep)
        for d, h in zip(self.d_meta, self.host[2:]):
            _chk(lib, lib.agatha_amd_memcpy_h2d_async(st, d.ptr, h.ctypes.data, h.nbytes))

    def pack(self, stream=None):
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        _chk(lib, lib.agatha_amd_pack(st, self.d_unp_q.ptr, self.qbytes, self.d_pk_q.ptr))
        _chk(lib, lib.agatha_amd_pack(st, self.d_unp_t.ptr, self.tbytes, self.d_pk_t.ptr))

    def seq_ops(self, qops=None, tops=None, stream=None):
        """Apply per-sequence reverse/complement op codes (0..3) to the packed batch; call after pack()."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        for ops, unp, pk, ln, of in ((qops, self.d_unp_q, self.d_pk_q, self.d_meta[2], self.d_meta[0]),
                                     (tops, self.d_unp_t, self.d_pk_t, self.d_meta[3], self.d_meta[1])):
            if ops is None:
                continue
            h = np.ascontiguousarray(ops, np.uint8)
            d = _DevBuf(lib, max(self.n, 1))
            self._tmp = getattr(self, "_tmp", []) + [d, h]
            _chk(lib, lib.agatha_amd_memcpy_h2d_async(st, d.ptr, h.ctypes.data, self.n))
            _chk(lib, lib.agatha_amd_seq_ops(st, unp.ptr, pk.ptr, ln.ptr, of.ptr, d.ptr, self.n))

    def align(self, scores, stream=None, use_len_hint=True):
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        m = self.d_meta
        _chk(lib, lib.agatha_amd_align(st, self.d_pk_q.ptr, self.d_pk_t.ptr, m[2].ptr, m[3].ptr, m[0].ptr, m[1].ptr,
                                       self.n, self.max_qlen if use_len_hint else 0,
                                       self.max_tlen if use_len_hint else 0, C.byref(scores),
                                       self.d_res[0].ptr, self.d_res[1].ptr, self.d_res[2].ptr,
                                       self.d_ws.ptr, self.ws_bytes))

    def align_starts(self, scores, stream=None):
        """Start positions of the alignments the last align() found (agatha_amd_align_starts): returns two int32 arrays
        (query start, target start); synchronises."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        m = self.d_meta
        nb = lib.agatha_amd_starts_scratch_bytes(self.qbytes, self.tbytes, self.n)
        scratch = _DevBuf(lib, nb)
        out = [_DevBuf(lib, 4 * self.n) for _ in range(2)]
        try:
            _chk(lib, lib.agatha_amd_align_starts(st, self.d_pk_q.ptr, self.d_pk_t.ptr, m[0].ptr, m[1].ptr, self.n, self.qbytes,
                                                  self.tbytes, self.max_qlen, self.max_tlen, C.byref(scores), self.d_res[1].ptr,
                                                  self.d_res[2].ptr, out[0].ptr, out[1].ptr, self.d_ws.ptr, self.ws_bytes,
                                                  scratch.ptr, nb))
            h = np.zeros((2, self.n), np.int32)
            for k in range(2):
                _chk(lib, lib.agatha_amd_memcpy_d2h_async(st, h[k].ctypes.data, out[k].ptr, 4 * self.n))
            _chk(lib, lib.agatha_amd_stream_synchronize(st))
            return h[0], h[1]
        finally:
            scratch.free()
            for o in out:
                o.free()

    def align_traceback(self, scores, stream=None, scratch_bytes=None):
        """Scores, end cells and alignment paths in one call (agatha_amd_align_traceback).  Returns (score, query_end,
        target_end, cigars): cigars[k] is the bytes object of pair k ((count << 2) | op; 0 match, 1 mismatch, 2 D, 3 I), b''
        for an empty alignment, None where the library reports AGATHA_AMD_NO_PATH.  scratch_bytes: device memory for the
        cell codes (default: the whole batch in one pass, capped at 8 GiB; a smaller one means more passes).  Synchronises."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        m = self.d_meta
        per = lib.agatha_amd_traceback_pair_bytes(self.max_qlen, self.max_tlen, C.byref(scores))
        if per == 0:
            raise AgathaError("traceback: band too wide or empty batch")
        if scratch_bytes is None:        # the whole batch in one pass, or passes of about 8 GiB
            ppp = max(1, min(self.n, (8 << 30) // per))
            scratch_bytes = lib.agatha_amd_traceback_scratch_bytes(self.n, self.max_qlen, self.max_tlen, C.byref(scores), ppp)
        scratch = _DevBuf(lib, scratch_bytes)
        cig = _DevBuf(lib, self.qbytes + self.tbytes + 16)
        nops = _DevBuf(lib, 4 * self.n)
        try:
            _chk(lib, lib.agatha_amd_align_traceback(st, self.d_pk_q.ptr, self.d_pk_t.ptr, m[2].ptr, m[3].ptr, m[0].ptr, m[1].ptr,
                                                     self.n, self.max_qlen, self.max_tlen, C.byref(scores), self.d_res[0].ptr,
                                                     self.d_res[1].ptr, self.d_res[2].ptr, cig.ptr, nops.ptr, self.d_ws.ptr,
                                                     self.ws_bytes, scratch.ptr, scratch_bytes))
            h_cig = np.zeros(self.qbytes + self.tbytes, np.uint8)
            h_n = np.zeros(self.n, np.uint32)
            _chk(lib, lib.agatha_amd_memcpy_d2h_async(st, h_cig.ctypes.data, cig.ptr, h_cig.nbytes))
            _chk(lib, lib.agatha_amd_memcpy_d2h_async(st, h_n.ctypes.data, nops.ptr, h_n.nbytes))
            self.download(st)
            _chk(lib, lib.agatha_amd_stream_synchronize(st))
            res = self.res_host.copy()
            off = self.host[2].astype(np.int64) + self.host[3].astype(np.int64)
            cigars = [None if k == 0xFFFFFFFF else h_cig[o:o + k].tobytes() for o, k in zip(off, h_n.astype(np.int64))]
            return res[0], res[1], res[2], cigars
        finally:
            scratch.free()
            cig.free()
            nops.free()

    def kernel_choice(self, stream=None):
        """("int32" | "int16", lanes per pair, slots per lane) of the kernel the device chose for the plain pairs."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        c = (C.c_int * 3)()
        _chk(lib, lib.agatha_amd_kernel_choice(st, self.d_ws.ptr, self.n, c))
        return ("int16" if c[0] else "int32", int(c[1]), int(c[2]))

    def schedule_info(self, stream=None):
        """(static schedule used, steps per lane group, lane groups used) of the last align(): see agatha_amd_schedule_info."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        c = (C.c_int * 3)()
        _chk(lib, lib.agatha_amd_schedule_info(st, self.d_ws.ptr, self.n, c))
        return bool(c[0]), int(c[1]), int(c[2])

    def split_info(self, stream=None):
        """(pairs that ran on the int16 latency shape beside the throughput shape, its lanes per pair, its slots per lane) of the last
        align(): see agatha_amd_split_info; (0, 0, 0) when one shape took the batch."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        c = (C.c_int * 3)()
        _chk(lib, lib.agatha_amd_split_info(st, self.d_ws.ptr, self.n, c))
        return int(c[0]), int(c[1]), int(c[2])

    def step_stats(self, stream=None):
        """(value wave-steps, key wave-steps, pairs started over, pairs started) of the int16 kernel in the last align()."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        c = (C.c_uint * 40)()
        _chk(lib, lib.agatha_amd_step_stats(st, self.d_ws.ptr, self.n, c))
        return tuple(int(v) for v in c)

    def flat_stats(self, stream=None):
        """(pairs that said whether they are flat, flat ones, young pairs started over on key steps, gate ticks, pairs sent to the clean-up launch, positions it looked at) of the int16 kernel's last align()."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        c = (C.c_uint * 6)()
        _chk(lib, lib.agatha_amd_flat_stats(st, self.d_ws.ptr, self.n, c))
        return tuple(int(v) for v in c)

    def timeline(self, stream=None, max_waves=4096):
        """Per-wave records of the int16 kernel (debug option "timeline"): array [waves, 8] of uint32, see agatha_amd_timeline."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        out = np.zeros((max_waves, 8), np.uint32)
        nw = lib.agatha_amd_timeline(st, self.d_ws.ptr, self.n, out.ctypes.data, max_waves)
        if nw < 0:
            _chk(lib, nw)
        return out[:nw]

    def pair_kinds(self, stream=None):
        """(plain, other letters, taken over by the int32 profile kernel) pair counts of the last align()."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        c = (C.c_uint32 * 3)()
        _chk(lib, lib.agatha_amd_pair_kinds(st, self.d_ws.ptr, self.n, c))
        return int(c[0]), int(c[1]), int(c[2])

    def download(self, stream=None):
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        for k in range(3):
            _chk(lib, lib.agatha_amd_memcpy_d2h_async(st, self.res_host[k].ctypes.data, self.d_res[k].ptr, 4 * self.n))
        if hasattr(lib, "agatha_amd_guard_stats") and getattr(self, "_guard", None) is not None:
            _chk(lib, lib.agatha_amd_guard_stats(st, self.d_ws.ptr, self.n, self._guard.ctypes.data, 0))
            if st == self.eng.stream:       # (another stream: Engine.synchronize() does not wait for it -- ask guard_stats())
                self._guard_pending = True
                self.eng._guarded.add(self)

    def guard_stats(self, stream=None):
        """(pairs that left for the int32 kernel because their step counter had run past their last step, ... because the state they were
        to be resumed from failed its check, states poisoned by the debug option poison_state, suspended states counted while it is on) of
        the int16 kernel in the last align(): agatha_amd_guard_stats.  Waits for the stream."""
        lib = self.eng.lib
        st = stream if stream is not None else self.eng.stream
        c = (C.c_uint * 4)()
        _chk(lib, lib.agatha_amd_guard_stats(st, self.d_ws.ptr, self.n, C.addressof(c), 1))
        return tuple(int(v) for v in c)

    def _check_guard(self):
        if self._guard_pending and self._guard is not None:
            self._guard_pending = False
            g = [int(v) for v in self._guard]
            if g[0] or g[1]:
                import warnings
                warnings.warn(f"agatha_amd: the packed-int16 kernel ended {g[0]} pair(s) whose step counter had run past their last step and refused "
                              f"{g[1]} saved state(s) that failed their check ({g[2]} poisoned on purpose by the debug option poison_state); those pairs "
                              "were redone by the int32 kernel and the results are right, but device memory the kernel owns was overwritten or its "
                              "state machine has a bug -- please report (agatha_amd_guard_stats)", RuntimeWarning, stacklevel=3)

    def packed_host(self):
        """Copies of the packed device buffers (for the pack-kernel parity test)."""
        lib = self.eng.lib
        out = []
        for d, nb in ((self.d_pk_q, self.qbytes // 2), (self.d_pk_t, self.tbytes // 2)):
            h = np.zeros(nb // 4, np.uint32)
            _chk(lib, lib.agatha_amd_memcpy_d2h_async(self.eng.stream, h.ctypes.data, d.ptr, nb))
            out.append(h)
        self.eng.synchronize()
        return out

    def free(self):
        if getattr(self, "_res_pinned", None):
            self.res_host = None
            self._guard = None
            self.eng._guarded.discard(self)
            self.eng.lib.agatha_amd_host_free(self._res_pinned)
            self._res_pinned = None
        for b in [self.d_unp_q, self.d_unp_t, self.d_pk_q, self.d_pk_t, self.d_ws] + self.d_meta + self.d_res:
            b.free()
        for b in getattr(self, "_tmp", []):
            if isinstance(b, _DevBuf):
                b.free()


class Engine:
    """Owns one device + one stream; `align_host_batch` is the whole hot path for one batch."""

    def __init__(self, device=0):
        self.lib = load_library()
        if self.lib.agatha_amd_device_count() <= 0:
            raise AgathaError("no HIP device visible: the alignment path has no CPU fallback")
        _chk(self.lib, self.lib.agatha_amd_set_device(int(device)))
        self.device = int(device)
        s = C.c_void_p()
        _chk(self.lib, self.lib.agatha_amd_stream_create(C.byref(s)))
        self.stream = s.value
        self._guarded = set()           # batches whose guard counters are on their way to the host (DeviceBatch.download)

    def synchronize(self):
        _chk(self.lib, self.lib.agatha_amd_stream_synchronize(self.stream))
        for b in list(self._guarded):
            b._check_guard()
        self._guarded.clear()

    def last_config(self):
        g, s = C.c_int(0), C.c_int(0)
        self.lib.agatha_amd_last_config(C.byref(g), C.byref(s))
        return g.value, s.value

    def last_int16_config(self):
        """(lanes per pair, slots per lane) of the packed-int16 kernel if it was a candidate in the last align, else
        None; DeviceBatch.kernel_choice() tells which candidate the device picked."""
        v = int(self.lib.agatha_amd_last_int16_config())
        return (v >> 8, v & 255) if v else None

    def batch(self, qbuf, tbuf, qoff, toff, qlen, tlen):
        return DeviceBatch(self, qbuf, tbuf, qoff, toff, qlen, tlen)

    def align_host_batch(self, qbuf, tbuf, qoff, toff, qlen, tlen, scores, use_len_hint=True, qops=None, tops=None):
        """ASCII host batch in the GASAL wire format -> (score, query_end, target_end) int32 arrays."""
        b = self.batch(qbuf, tbuf, qoff, toff, qlen, tlen)
        try:
            b.upload(); b.pack()
            if qops is not None or tops is not None:
                b.seq_ops(qops, tops)
            b.align(scores, use_len_hint=use_len_hint); b.download()
            self.synchronize()
            return b.res_host[0].copy(), b.res_host[1].copy(), b.res_host[2].copy()
        finally:
            b.free()

    def event(self):
        e = C.c_void_p()
        _chk(self.lib, self.lib.agatha_amd_event_create(C.byref(e)))
        return e.value

    def record(self, ev, stream=None):
        _chk(self.lib, self.lib.agatha_amd_event_record(ev, stream if stream is not None else self.stream))

    def set_kernel_events(self, e0, e1):
        self.lib.agatha_amd_set_kernel_events(e0, e1)

    def elapsed_ms(self, e0, e1):
        ms = C.c_float(0)
        _chk(self.lib, self.lib.agatha_amd_event_elapsed_ms(e0, e1, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if self.stream:
            self.lib.agatha_amd_stream_destroy(self.stream)
            self.stream = None
