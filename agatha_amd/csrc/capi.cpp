// capi.cpp -- the C-ABI of include/agatha_amd.h on top of the HIP kernels.
#include "../../include/agatha_amd.h"
#include "kernels.h"

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <mutex>
#include <cmath>
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
    {"no_migrate", "AGATHA_AMD_NO_MIGRATE", {0}},      // 1: pairs never move between lane groups (no preemptive schedule); -1: always when possible
    {"mig_timeout_us", "AGATHA_AMD_MIG_TIMEOUT_US", {50000}},   // wait for a suspended pair this long, then take it over
    {"mig_fresh_timeout_us", "AGATHA_AMD_MIG_FRESH_TIMEOUT_US", {2000}},   // ... and this long when the other lane group has not even started the pair (its workgroup is not resident)
    {"mig_test_delay_us", "AGATHA_AMD_MIG_TEST_DELAY_US", {0}}, // tests: odd lane groups start this late
    {"prio_slice", "AGATHA_AMD_PRIO_SLICE", {-1}},     // > 0: SIMD partners alternate issue priority every 2^n ticks (10 ns each); -1: 2^15 on a static schedule, off otherwise; 0: off
    {"prio_duty", "AGATHA_AMD_PRIO_DUTY", {0}},        // slices out of 16 in which the wave in slot 0 of its SIMD is favoured; 0 = automatic (8, or by the waves' step counts); -1: also no priority by the length of a wave's pair on the latency shapes' work queue
    {"timeline", "AGATHA_AMD_TIMELINE", {0}},          // 1: every wave of the int16 kernel records when and where it ran
    {"force_split", "AGATHA_AMD_FORCE_SPLIT", {0}},    // > 0: this many pairs (the longest) on the latency shape beside the throughput shape, whatever the cost model says (tests)
    {"ck_newer", "AGATHA_AMD_CK_NEWER", {1}},          // int16 kernel: 1 = a pair that must go back takes the newer of its two checkpoints when the bound of its maximum has risen far enough behind it (three register pairs per lane) / the checkpoint before "keep" when it has hardly risen behind "keep" (one or two), 0 = always the older one / always "keep" (round 3)
    {"ck_shift", "AGATHA_AMD_CK_SHIFT", {28}},         // int16 kernel: a pair of s steps takes a checkpoint every 2^(n - clz(s)) steps (28: every eighth to sixteenth of the pair, 29: quarter to eighth)
    {"lat_blocks", "AGATHA_AMD_LAT_BLOCKS", {0}},      // > 0: a batch split by length keeps its long pairs on this many workgroups of the latency shape (experiments; 0: no cap)
    {"no_split", "AGATHA_AMD_NO_SPLIT", {0}},          // 1: a batch of mixed lengths is never split between the two int16 shapes (one shape per launch, as before round 4)
    {"prio_fine", "AGATHA_AMD_PRIO_FINE", {0}},        // quarters of a slice added to slot 0's share of the issue priority (static schedule)
    {"fast_margin", "AGATHA_AMD_FAST_MARGIN", {12}},   // int16 kernel: value steps except in a window of key steps at a pair's end that starts n + steps / 128 before the corner of the shorter sequence; 0: key steps only
    {"fast_anchor", "AGATHA_AMD_FAST_ANCHOR", {1}},   // int16 kernel: 1 = the window of key steps is anchored at the corner of the shorter sequence, 0 = at the pair's last step (experiments)
    {"static_ck", "AGATHA_AMD_STATIC_CK", {1}},   // int16 kernel, static schedule, three register pairs per lane: 1 = checkpoints there as well (a pair that must be started over goes back in place), 0 = none (such a pair goes to the int32 kernel behind)
    {"win_cap_min", "AGATHA_AMD_WIN_CAP_MIN", {128}},    // int16 kernel: the adaptive part of the window of key steps at a pair's end is capped at max(win_cap_min, steps of the pair / win_cap_div)
    {"win_cap_div", "AGATHA_AMD_WIN_CAP_DIV", {16}},
    {"flat_detect", "AGATHA_AMD_FLAT_DETECT", {1}},     // int16 kernel: 1 = when most pairs of a batch say (at their 64th..127th step) that their score hardly rises, young pairs start over on key steps and later pairs start on them
    {"flat_percent", "AGATHA_AMD_FLAT_PERCENT", {30}},  // ... when more than this share of the pairs are flat (clean 10 %-error reads at match 1: 4-8 %; the flat batches of profiles/r05_v1: 49-88 %)
    {"cleanup_min_steps", "AGATHA_AMD_CLEANUP_MIN_STEPS", {384}},   // int16 kernel, static schedule: a pair that must start from its first step at step g of its t steps, 2 g > t + this, goes to the clean-up launch of the latency shape; 0 = it starts over in place (until round 5)
    {"probation", "AGATHA_AMD_PROBATION", {1}},         // int16 kernel: 1 = a pair that went back to a checkpoint (or its first step) runs key steps until z-drop is comfortably out of reach again, then value steps and a window; 0 = key steps for good (until round 5)
    {"no_pool", "AGATHA_AMD_NO_POOL", {0}},             // static schedule: 1 = every lane group resumes the pair that crosses out of its own interval (until round 4); 0 = the rests of the suspended pairs are a pool, longest first, for whoever is done with its fixed part
    {"mig_identity", "AGATHA_AMD_MIG_IDENTITY", {0}},   // static schedule: 1 = lane group g owns interval g of the line of pairs (until round 4); 0 = intervals whose pairs end together share a wave (schedule_kernel)
    {"ck_min_steps", "AGATHA_AMD_CK_MIN_STEPS", {384}},    // int16 kernel: pairs of at least this many steps take checkpoints (0: none do; 1024 until late in round 4: 3 kb pairs with broken reads among them, 22 -> 19 ms)
    {"poison_state", "AGATHA_AMD_POISON_STATE", {0}},     // tests: n > 0 = the int16 kernel writes the n-th suspended pair of a launch with a garbage step counter; 1000 + n = the same behind a flag that lets it pass the resume check, for the bound inside the step loop (see AlignLaunch::poison_state, agatha_amd_guard_stats)
    {"lazy_max", "AGATHA_AMD_LAZY_MAX", {8}},             // int16 kernel, one pair per wave: lazy value steps -- a calm test that passed with room to spare answers for up to this many steps behind it (no lower bound, no reduction, no test on those); 0 = every value step is tested
    {"tb_value_steps", "AGATHA_AMD_TB_VALUE_STEPS", {1}},  // traceback pass of the int16 kernel: value steps (1, round 6) or key steps only (0)
};
enum { OPT_MAX_BLOCKS, OPT_NO_DEAL, OPT_NO_INT16, OPT_FORCE_INT16, OPT_FORCE_CHOICE, OPT_NO_MIGRATE, OPT_MIG_TIMEOUT_US, OPT_MIG_FRESH_TIMEOUT_US, OPT_MIG_TEST_DELAY_US, OPT_PRIO_SLICE, OPT_PRIO_DUTY, OPT_TIMELINE, OPT_FORCE_SPLIT, OPT_CK_NEWER, OPT_CK_SHIFT, OPT_LAT_BLOCKS, OPT_NO_SPLIT, OPT_PRIO_FINE, OPT_FAST_MARGIN, OPT_FAST_ANCHOR, OPT_STATIC_CK, OPT_WIN_CAP_MIN, OPT_WIN_CAP_DIV, OPT_FLAT_DETECT, OPT_FLAT_PERCENT, OPT_CLEANUP_MIN_STEPS, OPT_PROBATION, OPT_NO_POOL, OPT_MIG_IDENTITY, OPT_CK_MIN_STEPS, OPT_POISON_STATE, OPT_LAZY_MAX, OPT_TB_VALUE_STEPS, OPT_COUNT };
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
constexpr size_t kQueueBytes = 2 * kAlign;       // the queue block: 64 counters of round 1-5 (queue heads, totals, choice, kinds, step statistics) + the guard counters of round 6
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

// A second stream beside every stream agatha_amd_align is called on, for the latency shape of a batch that the device splits by
// length between the two packed-int16 shapes (record_kernel), with the two events of the fork and the join.  Streams made by
// agatha_amd_stream_create get theirs there (the GASAL layer's: nothing is created inside a timed batch); a foreign stream
// (torch's, the null stream) gets one at its first call.  Kept until agatha_amd_stream_destroy / for the life of the process.
struct AuxStream { hipStream_t s; hipEvent_t fork, join; };
// (keyed by device AND stream: the null stream -- torch's default stream too -- is the same handle on every device, and an event or a
//  stream of one device must never be used on another: a process that drives several GPUs gets one entry per device.  Two host threads
//  that align on the SAME stream at the same time would share one fork / join pair; HIP streams are not meant to be fed by two threads
//  at once without the caller's own ordering, and the header says so: one caller per stream.)
struct AuxKey { int dev; void* stream; bool operator==(const AuxKey& o) const { return dev == o.dev && stream == o.stream; } };
struct AuxKeyHash { size_t operator()(const AuxKey& k) const { return std::hash<void*>()(k.stream) ^ ((size_t)(unsigned)k.dev * 0x9E3779B97F4A7C15ull); } };
std::mutex g_aux_mutex;
std::unordered_map<AuxKey, AuxStream, AuxKeyHash> g_aux;
int current_device() { int dev = 0; if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; } return dev; }
AuxStream* aux_stream(void* main_stream)
{
    const AuxKey key{current_device(), main_stream};
    std::lock_guard<std::mutex> lock(g_aux_mutex);
    auto it = g_aux.find(key);
    if (it != g_aux.end()) return &it->second;
    AuxStream a{nullptr, nullptr, nullptr};
    bool ok = hipStreamCreateWithFlags(&a.s, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&a.join, hipEventDisableTiming) == hipSuccess;
    if (!ok) {          // nothing half-made is kept: the caller runs everything on its own stream instead
        (void)hipGetLastError();
        if (a.join) (void)hipEventDestroy(a.join);
        if (a.fork) (void)hipEventDestroy(a.fork);
        if (a.s) (void)hipStreamDestroy(a.s);
        return nullptr;
    }
    return &g_aux.emplace(key, a).first->second;
}
void aux_stream_drop(void* main_stream)
{
    const AuxKey key{current_device(), main_stream};
    std::lock_guard<std::mutex> lock(g_aux_mutex);
    auto it = g_aux.find(key);
    if (it == g_aux.end()) return;
    (void)hipStreamDestroy(it->second.s); (void)hipEventDestroy(it->second.fork); (void)hipEventDestroy(it->second.join);
    g_aux.erase(it);
}

// Batches of more pairs than this can be larger than one round of the packed-int16 kernel's lane groups: their workspace
// also holds the areas of the preemptive schedule (prefix sums, boundary states, suspended pair state: ~68 MiB).
constexpr uint32_t kMigMinPairs = 4096;
size_t base_workspace_bytes(uint32_t n)
{
    return round_up(sizeof(uint32_t) * (size_t)n) + round_up(sizeof(uint32_t) * kBuckets) + kQueueBytes +
           round_up(sizeof(agatha::AlignLaunch)) + round_up((size_t)n) + round_up(sizeof(int) * agatha::kSimdStepsInts);
}
size_t mig_workspace_bytes(uint32_t n)
{
    return round_up(sizeof(uint32_t) * ((size_t)n + 1)) + round_up(sizeof(int) * (agatha::kMigMaxSlots + 1)) + round_up(sizeof(int) * agatha::kMigMaxSlots) + round_up(sizeof(int) * (2 * agatha::kMigMaxSlots + 4)) +
           round_up(sizeof(uint32_t) * agatha::kTimelineWaves * agatha::kTimelineDwords) + agatha::kMigBufBytes;
}

// Checkpoint area of the packed-int16 kernel (two slots of a suspended pair's size per lane group that can be in flight: the
// largest of its shapes' needs, never more lane groups than pairs)
struct CkShape { int G, P; };
const CkShape kCkShapes[] = {{16, 1}, {16, 2}, {16, 3}, {32, 2}, {32, 3}, {64, 1}, {64, 2}, {128, 1}, {64, 3}};
size_t ck_groups(int G, uint32_t n)
{
    const size_t cus = (size_t)num_cus();
    const size_t groups = G == 128 ? cus * 4 : cus * 8 * (size_t)(64 / G);
    return std::min<size_t>(groups, (size_t)n + 16);       // (a grid of whole workgroups: up to 16 lane groups each)
}
// (two parts: the shapes with fewer than 64 lanes per pair, then the ones with 64 and more -- a batch split by length between a
//  throughput and a latency shape runs both at the same time)
size_t ck_part_bytes(uint32_t n, bool lat)
{
    size_t need = 0;
    for (const CkShape& s : kCkShapes)
        if ((s.G >= 64) == lat) need = std::max(need, ck_groups(s.G, n) * (size_t)agatha::kCkSlots16 * (size_t)agatha::align16_mig_fields(s.P) * (size_t)s.G * sizeof(uint32_t));
    return round_up(need);
}
size_t ck_workspace_bytes(uint32_t n) { return ck_part_bytes(n, false) + ck_part_bytes(n, true); }

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
    return base_workspace_bytes(max_n_alns) + (max_n_alns > kMigMinPairs ? mig_workspace_bytes(max_n_alns) : 0);
}

size_t agatha_amd_workspace_bytes_long(uint32_t max_n_alns, uint32_t max_query_len, uint32_t max_target_len)
{
    const bool is_long = !max_query_len || !max_target_len ||
                         ((size_t)max_query_len + 7) / 8 + ((size_t)max_target_len + 7) / 8 >= (size_t)std::max(opt(OPT_CK_MIN_STEPS), 1);
    return agatha_amd_workspace_bytes(max_n_alns) + ((is_long && opt(OPT_CK_MIN_STEPS) > 0) ? ck_workspace_bytes(max_n_alns) : 0);
}

int agatha_amd_pack(void* stream, const uint8_t* d_unpacked, uint32_t nbytes, uint32_t* d_packed)
{
    if (!d_unpacked || !d_packed || nbytes == 0 || (nbytes % 8) != 0) return AGATHA_AMD_EINVAL;
    HIPCHK(agatha::launch_pack(d_unpacked, nbytes, d_packed, (hipStream_t)stream));
    return 0;
}

// ---- host-side packing (for pre-packed batches: halves the H2D bytes; reference ctors.cpp:65-73, gasal_align.cu:174) ----
static void pack_host_scalar(const uint8_t* a, size_t nbytes, uint32_t* out)
{
    for (size_t w = 0; w < nbytes / 8; w++) {
        uint32_t v = 0;
        for (int k = 0; k < 8; k++) v |= (uint32_t)(a[8 * w + k] & 15u) << (28 - 4 * k);      // pack_rc_seqs.h:21-33
        out[w] = v;
    }
}

__attribute__((target("avx2"))) static void pack_host_avx2(const uint8_t* a, size_t nbytes, uint32_t* out)
{
    // 64 bases -> 8 words per iteration: low nibbles, pairs combined with one multiply-add (16 * even + odd), 16-bit sums
    // narrowed to bytes, bytes of every word reversed (the first base sits in the word's top nibble)
    const __m256i lo4 = _mm256_set1_epi8(0x0F), coef = _mm256_set1_epi16(0x0110);
    const __m256i rev = _mm256_setr_epi8(3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9, 8, 15, 14, 13, 12, 3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9, 8, 15, 14, 13, 12);
    size_t i = 0;
    for (; i + 64 <= nbytes; i += 64) {
        const __m256i v0 = _mm256_and_si256(_mm256_loadu_si256((const __m256i*)(a + i)), lo4);
        const __m256i v1 = _mm256_and_si256(_mm256_loadu_si256((const __m256i*)(a + i + 32)), lo4);
        const __m256i p0 = _mm256_maddubs_epi16(v0, coef), p1 = _mm256_maddubs_epi16(v1, coef);
        __m256i b = _mm256_permute4x64_epi64(_mm256_packus_epi16(p0, p1), 0xD8);      // undo the per-lane interleave of packus
        _mm256_storeu_si256((__m256i*)(out + i / 8), _mm256_shuffle_epi8(b, rev));
    }
    pack_host_scalar(a + i, nbytes - i, out + i / 8);
}

int agatha_amd_pack_host(const uint8_t* h_unpacked, size_t nbytes, uint32_t* h_packed)
{
    if (!h_unpacked || !h_packed || (nbytes % 8) != 0) return AGATHA_AMD_EINVAL;
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2) pack_host_avx2(h_unpacked, nbytes, h_packed);
    else pack_host_scalar(h_unpacked, nbytes, h_packed);
    return 0;
}

// code of the low nibble of an ASCII letter (any case): A 1 -> 0, C 3 -> 1, G 7 -> 2, T 4 -> 3; everything else, N included: mask
static long pack2_host_scalar(const uint8_t* h_unpacked, size_t nbytes, uint16_t* h_codes, uint8_t* h_nmask)
{
    static const int8_t code_of[16] = {-1, 0, -1, 1, 3, -1, -1, 2, -1, -1, -1, -1, -1, -1, -1, -1};
    long other = 0;
    for (size_t w = 0; w < nbytes / 8; w++) {
        uint32_t codes = 0, mask = 0;
        for (int k = 0; k < 8; k++) {
            const uint8_t ch = h_unpacked[8 * w + k];
            const int c = code_of[ch & 15u];
            // (the low nibble alone cannot tell 'A' from 'Q': the letter itself must be one of ACGT, any case)
            const uint8_t up = (uint8_t)(ch & 0xDFu);
            const bool acgt = c >= 0 && (up == 'A' || up == 'C' || up == 'G' || up == 'T');
            if (acgt) codes |= (uint32_t)c << (14 - 2 * k);
            else { mask |= 1u << (7 - k); if (up != 'N') other++; }
        }
        h_codes[w] = (uint16_t)codes; h_nmask[w] = (uint8_t)mask;
    }
    return other;
}

__attribute__((target("avx2,popcnt"))) static long pack2_host_avx2(const uint8_t* a, size_t nbytes, uint16_t* h_codes, uint8_t* h_nmask)
{
    // 32 bases -> 4 code words + 4 mask bytes per iteration.  Two table lookups on the low nibble give the 2-bit code and the letter
    // that nibble must belong to (upper case); bases that are not that letter are masked.  The codes of a word are combined by two
    // multiply-adds (4 * even + odd, then 16 * even + odd: one byte per four bases) and a shift; the mask bits come from movemask,
    // bit-reversed per byte (the first base sits in the top bit).
    static uint8_t rev8[256];
    static const bool rev_ready = [] { for (int v = 0; v < 256; v++) { int r = 0; for (int b = 0; b < 8; b++) if (v & (1 << b)) r |= 0x80 >> b; rev8[v] = (uint8_t)r; } return true; }();
    (void)rev_ready;
    const __m256i lo4 = _mm256_set1_epi8(0x0F), upper = _mm256_set1_epi8((char)0xDF), letter_n = _mm256_set1_epi8('N');
    const __m256i code_lut = _mm256_setr_epi8(0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
    // (an entry without a letter holds a value whose low nibble differs from its index: it never equals a byte that selected it)
    const __m256i letter_lut = _mm256_setr_epi8(1, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0, 1, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i coef8 = _mm256_set1_epi16(0x0104), coef16 = _mm256_set1_epi32(0x00010010), low16 = _mm256_set1_epi64x(0xFFFF);
    long other = 0;
    size_t i = 0;
    for (; i + 32 <= nbytes; i += 32) {
        const __m256i ch = _mm256_loadu_si256((const __m256i*)(a + i));
        const __m256i nib = _mm256_and_si256(ch, lo4), up = _mm256_and_si256(ch, upper);
        const __m256i valid = _mm256_cmpeq_epi8(up, _mm256_shuffle_epi8(letter_lut, nib));
        const __m256i code = _mm256_and_si256(_mm256_shuffle_epi8(code_lut, nib), valid);
        const uint32_t inval = ~(uint32_t)_mm256_movemask_epi8(valid);
        const uint32_t is_n = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(up, letter_n));
        other += __builtin_popcount(inval & ~is_n);
        const __m256i p = _mm256_maddubs_epi16(code, coef8);              // 16-bit lanes: 4 c(2j) + c(2j+1)
        const __m256i q = _mm256_madd_epi16(p, coef16);                   // 32-bit lanes: 16 p(2k) + p(2k+1) = the byte of four bases
        const __m256i w = _mm256_and_si256(_mm256_or_si256(_mm256_slli_epi64(q, 8), _mm256_srli_epi64(q, 32)), low16);   // 64-bit lanes: one word each
        const size_t wi = i / 8;
        h_codes[wi] = (uint16_t)_mm256_extract_epi64(w, 0); h_codes[wi + 1] = (uint16_t)_mm256_extract_epi64(w, 1);
        h_codes[wi + 2] = (uint16_t)_mm256_extract_epi64(w, 2); h_codes[wi + 3] = (uint16_t)_mm256_extract_epi64(w, 3);
        h_nmask[wi] = rev8[inval & 0xFF]; h_nmask[wi + 1] = rev8[(inval >> 8) & 0xFF];
        h_nmask[wi + 2] = rev8[(inval >> 16) & 0xFF]; h_nmask[wi + 3] = rev8[inval >> 24];
    }
    return other + pack2_host_scalar(a + i, nbytes - i, h_codes + i / 8, h_nmask + i / 8);
}

long agatha_amd_pack2_host(const uint8_t* h_unpacked, size_t nbytes, uint16_t* h_codes, uint8_t* h_nmask)
{
    if (!h_unpacked || !h_codes || !h_nmask || (nbytes % 8) != 0) return AGATHA_AMD_EINVAL;
    static const bool have_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
    return have_avx2 ? pack2_host_avx2(h_unpacked, nbytes, h_codes, h_nmask) : pack2_host_scalar(h_unpacked, nbytes, h_codes, h_nmask);
}

int agatha_amd_unpack2(void* stream, const uint16_t* d_codes, const uint8_t* d_nmask, uint32_t nbytes, uint32_t* d_packed)
{
    if (!d_codes || !d_nmask || !d_packed || nbytes == 0 || (nbytes % 8) != 0) return AGATHA_AMD_EINVAL;
    HIPCHK(agatha::launch_unpack2(d_codes, d_nmask, nbytes / 8, d_packed, (hipStream_t)stream));
    return 0;
}

int agatha_amd_seq_ops(void* stream, const uint8_t* d_unpacked, uint32_t* d_packed, const uint32_t* d_lens,
                       const uint32_t* d_offsets, const uint8_t* d_ops, uint32_t n_seqs)
{
    if (!d_unpacked || !d_packed || !d_lens || !d_offsets || !d_ops || n_seqs == 0) return AGATHA_AMD_EINVAL;
    HIPCHK(agatha::launch_seq_ops(d_unpacked, d_packed, d_lens, d_offsets, d_ops, n_seqs, (hipStream_t)stream));
    return 0;
}

// blocks that can be live on one block-anti-diagonal: min(W + 1, ceil(Q/8), ceil(R/8))
static long window_blocks(const agatha_amd_scores* sc, uint32_t max_query_len, uint32_t max_target_len)
{
    const long W = ((long)sc->band_width + 7) / 8;
    long window = W + 1;
    if (max_query_len) window = std::min(window, ((long)max_query_len + 7) / 8);
    if (max_target_len) window = std::min(window, ((long)max_target_len + 7) / 8);
    return std::max(window, 1L);
}

// the traceback pass (every pair through the compare kernel, which also records the cell codes, then the walk)
struct TracebackArgs {
    void* scratch; size_t scratch_bytes;
    uint8_t* cigar; uint32_t* n_ops;
};
static size_t tb_header_bytes(uint32_t n) { return round_up(8 * (size_t)n) + round_up(4 * (size_t)n) + kAlign; }

// tb != nullptr: the traceback pass instead of the scoring kernels
static int align_impl(void* stream, const uint32_t* d_packed_query, const uint32_t* d_packed_target,
                      const uint32_t* d_query_lens, const uint32_t* d_target_lens,
                      const uint32_t* d_query_offsets, const uint32_t* d_target_offsets,
                      uint32_t n_alns, uint32_t max_query_len, uint32_t max_target_len,
                      const agatha_amd_scores* sc, int32_t* d_aln_score, int32_t* d_query_batch_end,
                      int32_t* d_target_batch_end, void* d_workspace, size_t workspace_bytes, const TracebackArgs* tb)
{
    if (!d_packed_query || !d_packed_target || !d_query_lens || !d_target_lens || !d_query_offsets ||
        !d_target_offsets || !sc || !d_aln_score || !d_query_batch_end || !d_target_batch_end || !d_workspace)
        return AGATHA_AMD_EINVAL;
    if (n_alns == 0 || n_alns > 0x7fffffffu) return AGATHA_AMD_EINVAL;
    if (sc->slice_width < 1 || sc->band_width < 0 || sc->gap_extend < 0) return AGATHA_AMD_EINVAL;
    if (workspace_bytes < base_workspace_bytes(n_alns)) return AGATHA_AMD_EWORKSPACE;

    const long window = window_blocks(sc, max_query_len, max_target_len);
    if (window > agatha::max_window_blocks()) return AGATHA_AMD_EBAND;
    // scores are carried as H << K in int32: the largest possible score -- and, with z-drop off, the deepest negative one --
    // must stay below 2^(30-K).  With length hints the whole call is refused here; without them (0 = unknown) the device
    // checks every pair and writes AGATHA_AMD_BAD_RESULT for the ones that do not fit (exotic_kernel, kind 3).
    const int Kbits = tb ? agatha::tb_key_bits((int)window) : agatha::key_bits_for_window((int)window);
    if (Kbits < 0) return AGATHA_AMD_ERANGE;
    const long long score_limit = 1ll << (30 - Kbits);
    if (max_query_len || max_target_len) {
        const long lmin = (max_query_len && max_target_len) ? std::min<long>(max_query_len, max_target_len)
                          : (long)std::max(max_query_len, max_target_len);
        const long lmax = (long)std::max(max_query_len, max_target_len);
        const long long top = (long long)std::max<long>(lmin, 1) * std::max(sc->match, 1) + 16384 + 2ll * (sc->band_width + 8) * sc->gap_extend;
        const long long per = std::max(sc->mismatch, 2 * sc->gap_extend);
        const long long low = sc->z_threshold < 0 ? (long long)lmax * std::max<long long>(per, 1) + sc->gap_open + 16384 : 0;
        if (top >= score_limit || low >= score_limit) return AGATHA_AMD_ERANGE;
    }

    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)d_workspace;
    uint32_t* order = (uint32_t*)ws;             ws += round_up(sizeof(uint32_t) * (size_t)n_alns);
    uint32_t* hist = (uint32_t*)ws;              ws += round_up(sizeof(uint32_t) * kBuckets);
    unsigned int* queue = (unsigned int*)ws;     ws += kQueueBytes;
    agatha::AlignLaunch* rec = (agatha::AlignLaunch*)ws;   ws += round_up(sizeof(agatha::AlignLaunch));
    uint8_t* exotic = (uint8_t*)ws;                          ws += round_up((size_t)n_alns);
    int* simd_steps = (int*)ws;                              ws += round_up(sizeof(int) * agatha::kSimdStepsInts);
    // areas of the preemptive schedule, present when the caller sized the workspace for a batch this large
    const bool mig = n_alns > kMigMinPairs && workspace_bytes >= base_workspace_bytes(n_alns) + mig_workspace_bytes(n_alns) &&
                     opt(OPT_NO_MIGRATE) <= 0;
    uint32_t* cum = (uint32_t*)ws;                           ws += round_up(sizeof(uint32_t) * ((size_t)n_alns + 1));
    int* mig_state = (int*)ws;                               ws += round_up(sizeof(int) * (agatha::kMigMaxSlots + 1));
    int* mig_perm = (int*)ws;                                ws += round_up(sizeof(int) * agatha::kMigMaxSlots);
    int* mig_late = (int*)ws;                                ws += round_up(sizeof(int) * (2 * agatha::kMigMaxSlots + 4));
    uint32_t* timeline = (uint32_t*)ws;                      ws += round_up(sizeof(uint32_t) * agatha::kTimelineWaves * agatha::kTimelineDwords);
    uint32_t* mig_buf = (uint32_t*)ws;
    // the checkpoint area lies behind everything else (behind the schedule's areas when the workspace holds them)
    const size_t ck_off = base_workspace_bytes(n_alns) + ((n_alns > kMigMinPairs && workspace_bytes >= base_workspace_bytes(n_alns) + mig_workspace_bytes(n_alns)) ? mig_workspace_bytes(n_alns) : 0);
    const bool have_ck = workspace_bytes >= ck_off + ck_workspace_bytes(n_alns) && opt(OPT_CK_MIN_STEPS) > 0;

    // inside the 256-byte queue block: [0..3] queue heads, [8] step totals, [10] kernel choice, [12] pair-kind counters,
    // [16..18] the schedule, [20..23] step statistics of the int16 kernel
    float* totals = (float*)(queue + 8);
    int* choice = (int*)(queue + 10);
    HIPCHK(hipMemsetAsync(queue, 0, kQueueBytes, st));
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
    L.step_stats = queue + 20;
    L.guard_stats = queue + 64;
    L.poison_state = opt(OPT_POISON_STATE);
    L.lazy_max = std::min(std::max(opt(OPT_LAZY_MAX), 0), 8);
    L.tb_value_steps = opt(OPT_TB_VALUE_STEPS) != 0;
    L.score_limit = score_limit;
    L.force_cmp = (tb || sc->match < -128 || sc->match > 127 || sc->mismatch < -127 || sc->mismatch > 128) ? 1 : 0;
    L.tb_codes = nullptr; L.tb_off = nullptr; L.tb_pass = nullptr; L.tb_plan = nullptr; L.tb_lanes = 0;
    HIPCHK(agatha::launch_exotic(L, st));
    int tb_passes = 0, tb_gs = 0, G16 = 0, P16 = 0;
    bool tb16 = false;
    if (tb) {
        // scratch = [word offset of every pair's codes | its pass | the plan | the code area]; the number of passes launched
        // is the worst case (every pair as long as the hints): a pass the plan did not need returns at once, so that the
        // call stays asynchronous
        tb_gs = agatha::tb_group_slots((int)window);
        const size_t pair_bytes = agatha_amd_traceback_pair_bytes(max_query_len, max_target_len, sc);
        if (!tb_gs || !pair_bytes) return AGATHA_AMD_EBAND;
        const size_t head = tb_header_bytes(n_alns);
        if (tb->scratch_bytes < head + pair_bytes) return AGATHA_AMD_EWORKSPACE;
        const size_t cap = (tb->scratch_bytes - head) & ~(size_t)255;
        const size_t per_pass = cap / pair_bytes;
        const size_t need_passes = ((size_t)n_alns + per_pass - 1) / per_pass;
        if (need_passes > 4096) return AGATHA_AMD_EWORKSPACE;          // (three launches per pass: give the call more scratch)
        tb_passes = (int)need_passes;
        char* p = (char*)tb->scratch;
        unsigned long long* off = (unsigned long long*)p;   p += round_up(8 * (size_t)n_alns);
        int* pass = (int*)p;                                 p += round_up(4 * (size_t)n_alns);
        int* plan = (int*)p;                                 p += kAlign;
        L.tb_codes = (uint32_t*)p; L.tb_off = off; L.tb_pass = pass; L.tb_plan = plan;
        tb16 = opt(OPT_NO_INT16) == 0 && agatha::align16_tb_config(L.p, (int)window, tb_gs, &G16, &P16);
        L.tb_lanes = tb16 ? G16 : 0;           // (the layout of the code words: both kernels of a pass and the walk read it)
        HIPCHK(agatha::launch_tb_plan(L, tb_gs, (unsigned long long)(cap / 4), tb_passes, off, pass, plan, st));
    }
    L.mig_enabled = 0; L.mig_slots = 0; L.cum = cum; L.sched = (int*)(queue + 16); L.mig_state = mig_state; L.mig_buf = mig_buf;
    L.mig_slot_dwords = 0; L.mig_fallback = 0; L.mig_perm = nullptr; L.mig_late = nullptr; L.mig_rest = nullptr; L.mig_identity = opt(OPT_MIG_IDENTITY) ? 1 : 0;
    L.timeline = nullptr;
    L.simd_steps = simd_steps;
    HIPCHK(hipMemsetAsync(simd_steps, 0, sizeof(int) * agatha::kSimdStepsInts, st));
    L.prio_slice_bits = opt(OPT_PRIO_SLICE);
    L.prio_duty = opt(OPT_PRIO_DUTY);
    L.prio_fine = opt(OPT_PRIO_FINE);
    if (opt(OPT_TIMELINE) && workspace_bytes >= base_workspace_bytes(n_alns) + mig_workspace_bytes(n_alns)) {
        L.timeline = timeline;
        HIPCHK(hipMemsetAsync(timeline, 0, sizeof(uint32_t) * agatha::kTimelineWaves * agatha::kTimelineDwords, st));
    }
    L.mig_timeout_ticks = 100u * (unsigned)std::max(opt(OPT_MIG_TIMEOUT_US), 0);
    L.mig_fresh_timeout_ticks = 100u * (unsigned)std::max(std::min(opt(OPT_MIG_FRESH_TIMEOUT_US), opt(OPT_MIG_TIMEOUT_US)), 0);
    L.mig_test_delay_ticks = 100u * (unsigned)std::max(opt(OPT_MIG_TEST_DELAY_US), 0);
    L.fast_margin = std::max(opt(OPT_FAST_MARGIN), 0);
    L.probation = opt(OPT_PROBATION) ? 1 : 0;
    L.flat_detect = opt(OPT_FLAT_DETECT) ? 1 : 0; L.flat_percent = std::max(opt(OPT_FLAT_PERCENT), 0);
    L.win_cap_min = std::max(opt(OPT_WIN_CAP_MIN), 0); L.win_cap_div = std::max(opt(OPT_WIN_CAP_DIV), 1);
    {
        // the window of key steps a pair starts with, before it has shown its own rate of rise (align16_body.inc, widen_window): the steps a
        // read with 15 % errors needs to rise by more than the slack of a value step's bound, 3.5 sigma
        const double m_ = sc->match, pen = std::max<double>(sc->match + sc->mismatch, 0.5 * sc->match + sc->gap_open + sc->gap_extend), e0 = 0.15;
        // (+ 8 m where the upper bound comes from the blocks' last columns alone: align16_body.inc, `slack`)
        const bool one_cell_ok = sc->z_threshold < 0 || 40 * sc->gap_open <= sc->z_threshold;
        const double X = 7.0 * std::max(sc->mismatch, 1) + (one_cell_ok ? 8.0 * std::max(sc->match, 0) : 0.0) + 7.0 * sc->gap_extend, mu = 4.0 * (m_ - e0 * pen), V = 4.0 * e0 * pen * pen, k2 = 12.0;
        double n = 4096.0;
        if (mu > 0.0) { const double r = (std::sqrt(k2 * V) + std::sqrt(k2 * V + 4.0 * mu * X)) / (2.0 * mu); n = std::min(4096.0, r * r); }
        L.win_prior = (int)std::ceil(n);
    }
    L.ck_buf = have_ck ? (uint32_t*)((char*)d_workspace + ck_off) : nullptr;
    // (what the caller's workspace really holds behind ck_off: the kernel derives from it how many of ITS lane groups fit -- a grid
    //  made larger than the default with the max_blocks debug option must not write checkpoints past the end of the workspace)
    L.ck_dwords = have_ck ? (unsigned long long)((workspace_bytes - ck_off) / sizeof(uint32_t)) : 0ull;
    L.ck_lat_off = (unsigned long long)(ck_part_bytes(n_alns, false) / sizeof(uint32_t));
    L.ck_min_steps = opt(OPT_CK_MIN_STEPS);
    L.static_ck = opt(OPT_STATIC_CK) ? 1 : 0;
    L.fast_anchor = opt(OPT_FAST_ANCHOR) ? 1 : 0;
    { static std::atomic<int> launches{1}; L.launch_id = launches.fetch_add(1, std::memory_order_relaxed); }
    L.max_blocks_override = opt(OPT_MAX_BLOCKS);
    L.no_deal = opt(OPT_NO_DEAL) ? 1 : 0;
    // Candidates for the plain pairs: the packed-int16 kernel when the scores and the band allow it, the int32 kernel
    // in its throughput shape and in its latency shape; which one runs is decided on the device from the batch's length
    // histogram (record_kernel).  Debug options no_int16 / force_int16 / force_choice override it (A/B runs, tests).
    L.force_choice = opt(OPT_FORCE_CHOICE);
    L.force_split = opt(OPT_FORCE_SPLIT);
    L.lat_blocks = opt(OPT_LAT_BLOCKS);
    L.ck_shift = opt(OPT_CK_SHIFT);
    L.ck_newer = opt(OPT_CK_NEWER);
    L.choice = choice; L.totals = totals;
    // (the int16 kernel keeps three flags of a pair in the top bits of its index)
    HIPCHK(agatha::plan_align(L, (int)window, opt(OPT_NO_INT16) != 0 || n_alns >= (1u << 28), opt(OPT_FORCE_INT16) != 0));
    g_last16 = (L.ncand > 0 && L.cand[0].kind == 1) ? ((L.cand[0].G << 8) | L.cand[0].S) : 0;
    // (the clean-up launch needs an int16 latency shape behind the int16 throughput shape)
    L.cleanup_min_steps = std::max(opt(OPT_CLEANUP_MIN_STEPS), 0);
    L.cleanup_ok = (!tb && L.cleanup_min_steps > 0 && L.ncand >= 2 && L.cand[0].kind == 1 && L.cand[1].kind == 1) ? 1 : 0;
    if (mig && L.ncand > 0 && L.cand[0].kind == 1 && L.cand[0].G < 64) {
        const int slots = L.cand[0].capacity, dwords = agatha::align16_mig_fields(L.cand[0].S / 2) * L.cand[0].G;
        if (slots <= agatha::kMigMaxSlots && (size_t)slots * dwords * sizeof(uint32_t) <= agatha::kMigBufBytes) {
            // (room for a second state per boundary: the fallback of a pair that is suspended with a bound for its maximum)
            const bool two = (size_t)slots * 2 * dwords * sizeof(uint32_t) <= agatha::kMigBufBytes;
            L.mig_enabled = opt(OPT_NO_MIGRATE) < 0 ? 2 : 1; L.mig_slots = slots; L.mig_perm = mig_perm; L.mig_late = opt(OPT_NO_POOL) ? nullptr : mig_late; L.mig_rest = (uint32_t*)(mig_late + agatha::kMigMaxSlots + 2); L.mig_slot_dwords = two ? 2 * dwords : dwords; L.mig_fallback = two ? 1 : 0;
            HIPCHK(hipMemsetAsync(mig_state, 0, sizeof(int) * ((size_t)slots + 1), st));
            HIPCHK(agatha::launch_schedule(L, st));
        }
    }
    L.self_dev = rec;
    // device copy of the record, queue head reset, kernel choice (one shape, or the batch split by length between the two int16
    // shapes: debug option no_split); stream-ordered
    HIPCHK(agatha::launch_record(L, rec, st, hist, kBuckets, (!tb && opt(OPT_NO_SPLIT) == 0) ? 1 : 0));
    if (tb) {
        // the pairs with plain letters run on the int16 kernel where a shape with the pass's slot count exists (bands of 49..192
        // blocks), the int32 kernel behind it takes the rest of the pass: other letters, N in the query, pairs it abandoned
        g_last16 = tb16 ? ((G16 << 8) | (2 * P16)) : 0;
        for (int pass = 0; pass < tb_passes; pass++) {
            if (pass > 0) HIPCHK(agatha::launch_record(L, rec, st));       // queue heads back to 0
            if (tb16) HIPCHK(agatha::launch_align16_tb(L, G16, P16, pass, st));
            HIPCHK(agatha::launch_align_tb(L, (int)window, pass, st));
            HIPCHK(agatha::launch_backtrace(L, tb_gs, pass, tb->cigar, tb->n_ops, st));
        }
        return 0;
    }
    if (g_ev0) HIPCHK(hipEventRecord(g_ev0, st));
    AuxStream* ax = (opt(OPT_NO_SPLIT) == 0) ? aux_stream(stream) : nullptr;
    HIPCHK(agatha::launch_align(L, (int)window, &g_lastG, &g_lastS, st, ax ? ax->s : nullptr, ax ? ax->fork : nullptr, ax ? ax->join : nullptr));
    if (g_ev1) HIPCHK(hipEventRecord(g_ev1, st));
    return 0;
}

int agatha_amd_align(void* stream, const uint32_t* d_packed_query, const uint32_t* d_packed_target,
                     const uint32_t* d_query_lens, const uint32_t* d_target_lens,
                     const uint32_t* d_query_offsets, const uint32_t* d_target_offsets,
                     uint32_t n_alns, uint32_t max_query_len, uint32_t max_target_len,
                     const agatha_amd_scores* sc, int32_t* d_aln_score, int32_t* d_query_batch_end,
                     int32_t* d_target_batch_end, void* d_workspace, size_t workspace_bytes)
{
    return align_impl(stream, d_packed_query, d_packed_target, d_query_lens, d_target_lens, d_query_offsets, d_target_offsets,
                      n_alns, max_query_len, max_target_len, sc, d_aln_score, d_query_batch_end, d_target_batch_end,
                      d_workspace, workspace_bytes, nullptr);
}

size_t agatha_amd_traceback_pair_bytes(uint32_t max_query_len, uint32_t max_target_len, const agatha_amd_scores* sc)
{
    if (!sc || !max_query_len || !max_target_len || sc->band_width < 0) return 0;
    const int gs = agatha::tb_group_slots((int)window_blocks(sc, max_query_len, max_target_len));
    if (!gs) return 0;
    // one 32-byte code block per (step, slot of the lane group); steps = block anti-diagonals of the longest pair
    const size_t steps = ((size_t)max_query_len + 7) / 8 + ((size_t)max_target_len + 7) / 8;
    return steps * (size_t)gs * 32;
}

int agatha_amd_align_traceback(void* stream, const uint32_t* d_packed_query, const uint32_t* d_packed_target,
                               const uint32_t* d_query_lens, const uint32_t* d_target_lens,
                               const uint32_t* d_query_offsets, const uint32_t* d_target_offsets,
                               uint32_t n_alns, uint32_t max_query_len, uint32_t max_target_len,
                               const agatha_amd_scores* sc, int32_t* d_aln_score, int32_t* d_query_batch_end,
                               int32_t* d_target_batch_end, uint8_t* d_cigar, uint32_t* d_n_cigar_ops,
                               void* d_workspace, size_t workspace_bytes, void* d_scratch, size_t scratch_bytes)
{
    if (!d_cigar || !d_n_cigar_ops || !d_scratch || !sc || n_alns == 0) return AGATHA_AMD_EINVAL;
    if (!max_query_len || !max_target_len) return AGATHA_AMD_EINVAL;        // they bound the number of passes
    TracebackArgs tb = {d_scratch, scratch_bytes, d_cigar, d_n_cigar_ops};
    return align_impl(stream, d_packed_query, d_packed_target, d_query_lens, d_target_lens, d_query_offsets, d_target_offsets,
                      n_alns, max_query_len, max_target_len, sc, d_aln_score, d_query_batch_end, d_target_batch_end,
                      d_workspace, workspace_bytes, &tb);
}

size_t agatha_amd_traceback_scratch_bytes(uint32_t n_alns, uint32_t max_query_len, uint32_t max_target_len,
                                          const agatha_amd_scores* sc, uint32_t pairs_per_pass)
{
    const size_t per = agatha_amd_traceback_pair_bytes(max_query_len, max_target_len, sc);
    if (!per || !n_alns) return 0;
    if (pairs_per_pass == 0 || pairs_per_pass > n_alns) pairs_per_pass = n_alns;
    return tb_header_bytes(n_alns) + 256 + per * (size_t)pairs_per_pass;
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

size_t agatha_amd_starts_scratch_bytes(uint32_t query_batch_bytes, uint32_t target_batch_bytes, uint32_t max_n_alns)
{
    // reversed packed prefixes of both sides (4 bit per base of the unpacked layout), their lengths, the backward results
    return round_up((size_t)query_batch_bytes / 2 + 16) + round_up((size_t)target_batch_bytes / 2 + 16) +
           5 * round_up(sizeof(uint32_t) * (size_t)max_n_alns);
}

int agatha_amd_align_starts(void* stream, const uint32_t* d_packed_query, const uint32_t* d_packed_target,
                            const uint32_t* d_query_offsets, const uint32_t* d_target_offsets, uint32_t n_alns,
                            uint32_t query_batch_bytes, uint32_t target_batch_bytes, uint32_t max_query_len, uint32_t max_target_len,
                            const agatha_amd_scores* sc, const int32_t* d_query_batch_end, const int32_t* d_target_batch_end,
                            int32_t* d_query_batch_start, int32_t* d_target_batch_start, void* d_workspace,
                            size_t workspace_bytes, void* d_scratch, size_t scratch_bytes)
{
    if (!d_packed_query || !d_packed_target || !d_query_offsets || !d_target_offsets || !sc || !d_query_batch_end ||
        !d_target_batch_end || !d_query_batch_start || !d_target_batch_start || !d_workspace || !d_scratch || n_alns == 0)
        return AGATHA_AMD_EINVAL;
    if (scratch_bytes < agatha_amd_starts_scratch_bytes(query_batch_bytes, target_batch_bytes, n_alns)) return AGATHA_AMD_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)d_scratch;
    uint32_t* rq = (uint32_t*)p;   p += round_up((size_t)query_batch_bytes / 2 + 16);
    uint32_t* rt = (uint32_t*)p;   p += round_up((size_t)target_batch_bytes / 2 + 16);
    const size_t arr = round_up(sizeof(uint32_t) * (size_t)n_alns);
    uint32_t* rql = (uint32_t*)p;  p += arr;
    uint32_t* rtl = (uint32_t*)p;  p += arr;
    int32_t* bs = (int32_t*)p;     p += arr;
    int32_t* bq = (int32_t*)p;     p += arr;
    int32_t* bt = (int32_t*)p;
    HIPCHK(agatha::launch_reverse_prefix(d_packed_query, rq, d_query_offsets, d_query_batch_end, rql, n_alns, st));
    HIPCHK(agatha::launch_reverse_prefix(d_packed_target, rt, d_target_offsets, d_target_batch_end, rtl, n_alns, st));
    agatha_amd_scores back = *sc;
    // The alignment is known to exist: no z-drop on the way back.  Not z = -1, which would arm the range check for scores
    // that sink without bound (it refuses reads of ~127 kb at band 1500): from the end cell back along the alignment every
    // partial score is >= 0 (the end cell holds the maximum over all prefixes), and cells off the path lie at most a band
    // width below it, so a threshold no drop can reach means the same and checks nothing.
    back.z_threshold = 1 << 24;
    const int rc = agatha_amd_align(stream, rq, rt, rql, rtl, d_query_offsets, d_target_offsets, n_alns, max_query_len, max_target_len,
                                    &back, bs, bq, bt, d_workspace, workspace_bytes);
    if (rc != 0) return rc;
    HIPCHK(agatha::launch_starts(d_query_batch_end, d_target_batch_end, bq, bt, d_query_batch_start, d_target_batch_start, n_alns, st));
    return 0;
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
    if (e == hipSuccess) e = hipMemcpyAsync(&rec, ws + kQueueBytes, sizeof(rec), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "agatha_amd_kernel_choice");
    if (choice < 0 || choice >= rec.ncand) return AGATHA_AMD_EINVAL;
    out[0] = rec.cand[choice].kind; out[1] = rec.cand[choice].G; out[2] = rec.cand[choice].S;
    return 0;
}

int agatha_amd_split_info(void* stream, const void* d_workspace, uint32_t n_alns, int out[3])
{
    if (!d_workspace || !out || n_alns == 0) return AGATHA_AMD_EINVAL;
    const char* ws = (const char*)d_workspace;
    ws += round_up(sizeof(uint32_t) * (size_t)n_alns) + round_up(sizeof(uint32_t) * kBuckets);
    unsigned int n_long = 0;
    agatha::AlignLaunch rec;
    hipError_t e = hipMemcpyAsync(&n_long, ws + 11 * sizeof(unsigned int), sizeof(n_long), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&rec, ws + kQueueBytes, sizeof(rec), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "agatha_amd_split_info");
    out[0] = (int)n_long; out[1] = out[2] = 0;
    if (n_long && rec.ncand >= 2) { out[1] = rec.cand[1].G; out[2] = rec.cand[1].S; }
    return 0;
}

int agatha_amd_schedule_info(void* stream, const void* d_workspace, uint32_t n_alns, int out[3])
{
    if (!d_workspace || !out || n_alns == 0) return AGATHA_AMD_EINVAL;
    const char* ws = (const char*)d_workspace;
    ws += round_up(sizeof(uint32_t) * (size_t)n_alns) + round_up(sizeof(uint32_t) * kBuckets);
    hipError_t e = hipMemcpyAsync(out, ws + 16 * sizeof(unsigned int), 3 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "agatha_amd_schedule_info");
    return 0;
}

int agatha_amd_step_stats(void* stream, const void* d_workspace, uint32_t n_alns, unsigned int out[40])
{
    if (!d_workspace || !out || n_alns == 0) return AGATHA_AMD_EINVAL;
    const char* ws = (const char*)d_workspace;
    ws += round_up(sizeof(uint32_t) * (size_t)n_alns) + round_up(sizeof(uint32_t) * kBuckets);
    hipError_t e = hipMemcpyAsync(out, ws + 20 * sizeof(unsigned int), 40 * sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "agatha_amd_step_stats");
    return 0;
}

int agatha_amd_guard_stats(void* stream, const void* d_workspace, uint32_t n_alns, unsigned int out[4], int synchronise)
{
    if (!d_workspace || !out || n_alns == 0) return AGATHA_AMD_EINVAL;
    const char* ws = (const char*)d_workspace;
    ws += round_up(sizeof(uint32_t) * (size_t)n_alns) + round_up(sizeof(uint32_t) * kBuckets);
    hipError_t e = hipMemcpyAsync(out, ws + 64 * sizeof(unsigned int), 4 * sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess && synchronise) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "agatha_amd_guard_stats");
    return 0;
}

int agatha_amd_flat_stats(void* stream, const void* d_workspace, uint32_t n_alns, unsigned int out[6])
{
    if (!d_workspace || !out || n_alns == 0) return AGATHA_AMD_EINVAL;
    const char* ws = (const char*)d_workspace;
    ws += round_up(sizeof(uint32_t) * (size_t)n_alns) + round_up(sizeof(uint32_t) * kBuckets);
    hipError_t e = hipMemcpyAsync(out, ws + 60 * sizeof(unsigned int), 4 * sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out + 4, ws + 14 * sizeof(unsigned int), 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "agatha_amd_flat_stats");
    return 0;
}

int agatha_amd_timeline(void* stream, const void* d_workspace, uint32_t n_alns, uint32_t* out, uint32_t max_waves)
{
    if (!d_workspace || !out || n_alns == 0) return AGATHA_AMD_EINVAL;
    // the timeline area is part of the schedule's areas: only workspaces sized for more than kMigMinPairs pairs have it
    if (n_alns <= kMigMinPairs) { snprintf(g_err, sizeof(g_err), "agatha_amd_timeline: workspaces for <= %u pairs hold no timeline area", kMigMinPairs); return AGATHA_AMD_EWORKSPACE; }
    const char* ws = (const char*)d_workspace;
    ws += base_workspace_bytes(n_alns) + round_up(sizeof(uint32_t) * ((size_t)n_alns + 1)) + round_up(sizeof(int) * (agatha::kMigMaxSlots + 1)) + round_up(sizeof(int) * agatha::kMigMaxSlots) + round_up(sizeof(int) * (2 * agatha::kMigMaxSlots + 4));
    const uint32_t nw = std::min<uint32_t>(max_waves, agatha::kTimelineWaves);
    hipError_t e = hipMemcpyAsync(out, ws, sizeof(uint32_t) * agatha::kTimelineDwords * nw, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "agatha_amd_timeline");
    return (int)nw;
}

int agatha_amd_pair_kinds(void* stream, const void* d_workspace, uint32_t n_alns, uint32_t counts[3])
{
    if (!d_workspace || !counts || n_alns == 0) return AGATHA_AMD_EINVAL;
    const char* ws = (const char*)d_workspace;
    ws += round_up(sizeof(uint32_t) * (size_t)n_alns) + round_up(sizeof(uint32_t) * kBuckets) + kQueueBytes + round_up(sizeof(agatha::AlignLaunch));
    uint8_t* h = (uint8_t*)malloc(n_alns);
    if (!h) return AGATHA_AMD_EINVAL;
    hipError_t e = hipMemcpyAsync(h, ws, n_alns, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { free(h); return hip_fail(e, "agatha_amd_pair_kinds"); }
    counts[0] = counts[1] = counts[2] = 0;
    for (uint32_t k = 0; k < n_alns; k++) { const int kd = (h[k] & 0x7f) == 5 ? 0 : (h[k] & 0x7f); if (kd < 3) counts[kd]++; }      // (5: a plain pair the clean-up launch took)
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
int agatha_amd_stream_create(void** stream) { if (!stream) return AGATHA_AMD_EINVAL; hipStream_t s; HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); *stream = (void*)s; (void)aux_stream((void*)s); return 0; }
int agatha_amd_stream_destroy(void* stream) { if (stream) { aux_stream_drop(stream); HIPCHK(hipStreamDestroy((hipStream_t)stream)); } return 0; }
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
int agatha_amd_stream_wait_event(void* stream, void* event) { HIPCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0)); return 0; }
int agatha_amd_get_device(void) { int d = -1; HIPCHK(hipGetDevice(&d)); return d; }
int agatha_amd_event_elapsed_ms(void* start, void* stop, float* ms)
{
    if (!ms) return AGATHA_AMD_EINVAL;
    HIPCHK(hipEventSynchronize((hipEvent_t)stop));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

}  // extern "C"
