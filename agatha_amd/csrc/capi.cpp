// capi.cpp -- the C-ABI of include/agatha_amd.h on top of the HIP kernels.
#include "../../include/agatha_amd.h"
#include "kernels.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <cstdlib>

namespace {

thread_local char g_err[256] = "";
thread_local int g_lastG = 0, g_lastS = 0, g_last16 = 0;
thread_local hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;

int hip_fail(hipError_t e, const char* what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s (HIP error %d)", what, hipGetErrorString(e), (int)e);
    return AGATHA_AMD_EHIP;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)

// Debug / A-B options (agatha_amd_set_debug_option).  Process-global, read with relaxed atomics on the hot path; the
// environment variables AGATHA_AMD_<NAME> only give the initial values, read once.
struct DebugOption { const char* name; const char* env; std::atomic<int> value; };
DebugOption g_opts[] = {
    {"max_blocks", "AGATHA_AMD_MAX_BLOCKS", {0}},      // > 0: cap of the persistent grids
    {"no_deal", "AGATHA_AMD_NO_DEAL", {0}},            // 1: no dealt first round, every pair from the work queue
    {"no_int16", "AGATHA_AMD_NO_INT16", {0}},          // 1: the packed-int16 kernel is not a candidate
    {"force_int16", "AGATHA_AMD_FORCE_INT16", {0}},    // 1: ... is the only candidate (when the scores allow it)
    {"force_choice", "AGATHA_AMD_FORCE_CHOICE", {-1}}, // >= 0: candidate index that takes the plain pairs
    {"no_migrate", "AGATHA_AMD_NO_MIGRATE", {0}},      // 1: pairs never move between lane groups (no preemptive schedule)
};
enum { OPT_MAX_BLOCKS, OPT_NO_DEAL, OPT_NO_INT16, OPT_FORCE_INT16, OPT_FORCE_CHOICE, OPT_NO_MIGRATE, OPT_COUNT };
std::once_flag g_opts_once;
void init_opts()
{
    std::call_once(g_opts_once, [] {
        for (DebugOption& o : g_opts) { const char* e = getenv(o.env); if (e && *e) o.value.store(atoi(e), std::memory_order_relaxed); }
    });
}
int opt(int which) { init_opts(); return g_opts[which].value.load(std::memory_order_relaxed); }

constexpr uint32_t kBuckets = 16384;     // sort buckets of 32 bases of (query + target) length
constexpr size_t kAlign = 256;
size_t round_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

int num_cus()
{
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        hipDeviceProp_t prop;
        cached = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cached_dev = dev;
    }
    return cached;
}

}  // namespace

extern "C" {

const char* agatha_amd_strerror(int code)
{
    switch (code) {
        case AGATHA_AMD_OK: return "ok";
        case AGATHA_AMD_EINVAL: return "invalid argument";
        case AGATHA_AMD_EBAND: return "band wider than the largest compiled window";
        case AGATHA_AMD_EWORKSPACE: return "workspace too small";
        case AGATHA_AMD_EHIP: return "HIP runtime error";
        case AGATHA_AMD_ERANGE: return "scores can exceed the int32 range of the kernel for these lengths/band (pass length hints, lower -m or -w)";
        default: return "unknown error";
    }
}
const char* agatha_amd_last_error(void) { return g_err; }
const char* agatha_amd_version(void) { return "agatha_amd 0.1 (gfx950)"; }

int agatha_amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int agatha_amd_set_device(int device) { HIPCHK(hipSetDevice(device)); return 0; }

int agatha_amd_max_band(void) { return (agatha::max_window_blocks() - 1) * 8; }

size_t agatha_amd_workspace_bytes(uint32_t max_n_alns)
{
    return round_up(sizeof(uint32_t) * (size_t)max_n_alns) + round_up(sizeof(uint32_t) * kBuckets) + kAlign +
           round_up(sizeof(agatha::AlignLaunch)) + round_up((size_t)max_n_alns);
}

int agatha_amd_pack(void* stream, const uint8_t* d_unpacked, uint32_t nbytes, uint32_t* d_packed)
{
    if (!d_unpacked || !d_packed || nbytes == 0 || (nbytes % 8) != 0) return AGATHA_AMD_EINVAL;
    HIPCHK(agatha::launch_pack(d_unpacked, nbytes, d_packed, (hipStream_t)stream));
    return 0;
}

int agatha_amd_seq_ops(void* stream, const uint8_t* d_unpacked, uint32_t* d_packed, const uint32_t* d_lens,
                       const uint32_t* d_offsets, const uint8_t* d_ops, uint32_t n_seqs)
{
    if (!d_unpacked || !d_packed || !d_lens || !d_offsets || !d_ops || n_seqs == 0) return AGATHA_AMD_EINVAL;
    HIPCHK(agatha::launch_seq_ops(d_unpacked, d_packed, d_lens, d_offsets, d_ops, n_seqs, (hipStream_t)stream));
    return 0;
}

int agatha_amd_align(void* stream, const uint32_t* d_packed_query, const uint32_t* d_packed_target,
                     const uint32_t* d_query_lens, const uint32_t* d_target_lens,
                     const uint32_t* d_query_offsets, const uint32_t* d_target_offsets,
                     uint32_t n_alns, uint32_t max_query_len, uint32_t max_target_len,
                     const agatha_amd_scores* sc, int32_t* d_aln_score, int32_t* d_query_batch_end,
                     int32_t* d_target_batch_end, void* d_workspace, size_t workspace_bytes)
{
    if (!d_packed_query || !d_packed_target || !d_query_lens || !d_target_lens || !d_query_offsets ||
        !d_target_offsets || !sc || !d_aln_score || !d_query_batch_end || !d_target_batch_end || !d_workspace)
        return AGATHA_AMD_EINVAL;
    if (n_alns == 0 || n_alns > 0x7fffffffu) return AGATHA_AMD_EINVAL;
    if (sc->slice_width < 1 || sc->band_width < 0 || sc->gap_extend < 0) return AGATHA_AMD_EINVAL;
    if (workspace_bytes < agatha_amd_workspace_bytes(n_alns)) return AGATHA_AMD_EWORKSPACE;

    // blocks that can be live on one block-anti-diagonal: min(W + 1, ceil(Q/8), ceil(R/8))
    const long W = ((long)sc->band_width + 7) / 8;
    long window = W + 1;
    if (max_query_len) window = std::min(window, ((long)max_query_len + 7) / 8);
    if (max_target_len) window = std::min(window, ((long)max_target_len + 7) / 8);
    window = std::max(window, 1L);
    if (window > agatha::max_window_blocks()) return AGATHA_AMD_EBAND;
    // scores are carried as H << K in int32: the largest possible score must stay below 2^(30-K)
    {
        const int K = agatha::key_bits_for_window((int)window);
        const long lmax = (max_query_len && max_target_len) ? std::min<long>(max_query_len, max_target_len)
                          : (long)std::max(max_query_len, max_target_len);
        const long top = std::max<long>(lmax, 1) * std::max(sc->match, 1) + 16384 + 2L * (sc->band_width + 8) * sc->gap_extend;
        if (K < 0 || top >= (1L << (30 - K))) return AGATHA_AMD_ERANGE;
    }

    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)d_workspace;
    uint32_t* order = (uint32_t*)ws;             ws += round_up(sizeof(uint32_t) * (size_t)n_alns);
    uint32_t* hist = (uint32_t*)ws;              ws += round_up(sizeof(uint32_t) * kBuckets);
    unsigned int* queue = (unsigned int*)ws;     ws += kAlign;
    agatha::AlignLaunch* rec = (agatha::AlignLaunch*)ws;   ws += round_up(sizeof(agatha::AlignLaunch));
    uint8_t* exotic = (uint8_t*)ws;

    // inside the 256-byte queue block: [0..3] queue heads, [8] step totals, [10] kernel choice, [12] pair-kind counters
    float* totals = (float*)(queue + 8);
    int* choice = (int*)(queue + 10);
    HIPCHK(hipMemsetAsync(queue, 0, kAlign, st));
    HIPCHK(agatha::launch_sort(d_query_lens, d_target_lens, (int)n_alns, hist, kBuckets, order, totals, st));

    agatha::AlignLaunch L;
    L.packed_q = d_packed_query; L.packed_t = d_packed_target;
    L.qlens = d_query_lens; L.tlens = d_target_lens; L.qoffs = d_query_offsets; L.toffs = d_target_offsets;
    L.order = order; L.n = (int)n_alns; L.queue = queue;
    L.score = d_aln_score; L.qend = d_query_batch_end; L.tend = d_target_batch_end;
    L.p = {sc->match, sc->mismatch, sc->gap_open, sc->gap_extend, sc->slice_width, sc->z_threshold, sc->band_width};
    L.num_cus = num_cus();
    L.exotic = exotic;
    L.kind_counts = queue + 12;
    L.force_cmp = (sc->match < -128 || sc->match > 127 || sc->mismatch < -127 || sc->mismatch > 128) ? 1 : 0;
    HIPCHK(agatha::launch_exotic(L, st));
    L.max_blocks_override = opt(OPT_MAX_BLOCKS);
    L.no_deal = opt(OPT_NO_DEAL) ? 1 : 0;
    // Candidates for the plain pairs: the packed-int16 kernel when the scores and the band allow it, the int32 kernel
    // in its throughput shape and in its latency shape; which one runs is decided on the device from the batch's length
    // histogram (record_kernel).  Debug options no_int16 / force_int16 / force_choice override it (A/B runs, tests).
    L.force_choice = opt(OPT_FORCE_CHOICE);
    L.choice = choice; L.totals = totals;
    HIPCHK(agatha::plan_align(L, (int)window, opt(OPT_NO_INT16) != 0, opt(OPT_FORCE_INT16) != 0));
    g_last16 = (L.ncand > 0 && L.cand[0].kind == 1) ? ((L.cand[0].G << 8) | L.cand[0].S) : 0;
    L.self_dev = rec;
    HIPCHK(agatha::launch_record(L, rec, st));     // device copy of the record, queue head reset, kernel choice; stream-ordered
    if (g_ev0) HIPCHK(hipEventRecord(g_ev0, st));
    HIPCHK(agatha::launch_align(L, (int)window, &g_lastG, &g_lastS, st));
    if (g_ev1) HIPCHK(hipEventRecord(g_ev1, st));
    return 0;
}

int agatha_amd_set_debug_option(const char* name, int value)
{
    if (!name) return AGATHA_AMD_EINVAL;
    init_opts();
    for (DebugOption& o : g_opts)
        if (strcmp(o.name, name) == 0) { o.value.store(value, std::memory_order_relaxed); return 0; }
    return AGATHA_AMD_EINVAL;
}

int agatha_amd_get_debug_option(const char* name, int* value)
{
    if (!name || !value) return AGATHA_AMD_EINVAL;
    init_opts();
    for (DebugOption& o : g_opts)
        if (strcmp(o.name, name) == 0) { *value = o.value.load(std::memory_order_relaxed); return 0; }
    return AGATHA_AMD_EINVAL;
}

void agatha_amd_set_kernel_events(void* ev_begin, void* ev_end) { g_ev0 = (hipEvent_t)ev_begin; g_ev1 = (hipEvent_t)ev_end; }

void agatha_amd_last_config(int* G, int* S) { if (G) *G = g_lastG; if (S) *S = g_lastS; }

int agatha_amd_last_int16_config(void) { return g_last16; }

int agatha_amd_kernel_choice(void* stream, const void* d_workspace, uint32_t n_alns, int out[3])
{
    if (!d_workspace || !out || n_alns == 0) return AGATHA_AMD_EINVAL;
    const char* ws = (const char*)d_workspace;
    ws += round_up(sizeof(uint32_t) * (size_t)n_alns) + round_up(sizeof(uint32_t) * kBuckets);
    int choice = 0;
    agatha::AlignLaunch rec;
    hipError_t e = hipMemcpyAsync(&choice, ws + 10 * sizeof(unsigned int), sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&rec, ws + kAlign, sizeof(rec), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "agatha_amd_kernel_choice");
    if (choice < 0 || choice >= rec.ncand) return AGATHA_AMD_EINVAL;
    out[0] = rec.cand[choice].kind; out[1] = rec.cand[choice].G; out[2] = rec.cand[choice].S;
    return 0;
}

int agatha_amd_pair_kinds(void* stream, const void* d_workspace, uint32_t n_alns, uint32_t counts[3])
{
    if (!d_workspace || !counts || n_alns == 0) return AGATHA_AMD_EINVAL;
    const char* ws = (const char*)d_workspace;
    ws += round_up(sizeof(uint32_t) * (size_t)n_alns) + round_up(sizeof(uint32_t) * kBuckets) + kAlign + round_up(sizeof(agatha::AlignLaunch));
    uint8_t* h = (uint8_t*)malloc(n_alns);
    if (!h) return AGATHA_AMD_EINVAL;
    hipError_t e = hipMemcpyAsync(h, ws, n_alns, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { free(h); return hip_fail(e, "agatha_amd_pair_kinds"); }
    counts[0] = counts[1] = counts[2] = 0;
    for (uint32_t k = 0; k < n_alns; k++) if (h[k] < 3) counts[h[k]]++;
    free(h);
    return 0;
}

int agatha_amd_malloc(void** d_ptr, size_t bytes) { if (!d_ptr) return AGATHA_AMD_EINVAL; HIPCHK(hipMalloc(d_ptr, bytes ? bytes : 1)); return 0; }
int agatha_amd_free(void* d_ptr) { if (d_ptr) HIPCHK(hipFree(d_ptr)); return 0; }
int agatha_amd_host_alloc(void** h_ptr, size_t bytes) { if (!h_ptr) return AGATHA_AMD_EINVAL; HIPCHK(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault)); return 0; }
int agatha_amd_host_free(void* h_ptr) { if (h_ptr) HIPCHK(hipHostFree(h_ptr)); return 0; }
int agatha_amd_memcpy_h2d_async(void* stream, void* d_dst, const void* h_src, size_t bytes)
{ if (bytes) HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream)); return 0; }
int agatha_amd_memcpy_d2h_async(void* stream, void* h_dst, const void* d_src, size_t bytes)
{ if (bytes) HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream)); return 0; }
int agatha_amd_stream_create(void** stream) { if (!stream) return AGATHA_AMD_EINVAL; hipStream_t s; HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); *stream = (void*)s; return 0; }
int agatha_amd_stream_destroy(void* stream) { if (stream) HIPCHK(hipStreamDestroy((hipStream_t)stream)); return 0; }
int agatha_amd_stream_synchronize(void* stream) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return 0; }
int agatha_amd_stream_query(void* stream)
{
    hipError_t e = hipStreamQuery((hipStream_t)stream);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 1; }
    return hip_fail(e, "hipStreamQuery");
}
int agatha_amd_event_create(void** event) { if (!event) return AGATHA_AMD_EINVAL; hipEvent_t e; HIPCHK(hipEventCreate(&e)); *event = (void*)e; return 0; }
int agatha_amd_event_destroy(void* event) { if (event) HIPCHK(hipEventDestroy((hipEvent_t)event)); return 0; }
int agatha_amd_event_record(void* event, void* stream) { HIPCHK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream)); return 0; }
int agatha_amd_event_elapsed_ms(void* start, void* stop, float* ms)
{
    if (!ms) return AGATHA_AMD_EINVAL;
    HIPCHK(hipEventSynchronize((hipEvent_t)stop));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

}  // extern "C"
