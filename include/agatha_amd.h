/*
 * agatha_amd.h -- C-ABI of the MI355X guided-alignment engine (libagatha_amd.so).
 *
 * This is the drop-in boundary for the hot path of readwrite112/AGAThA: every entry point replaces one
 * CUDA launch site of the reference's gasal_aln_async() (AGAThA/src/gasal_align.cu).  All pointers
 * prefixed d_ are DEVICE pointers (HIP), `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  Every call is asynchronous on `stream` and allocates nothing.  Return value: 0 on success,
 * a negative AGATHA_AMD_E* code otherwise (agatha_amd_strerror()).  No torch / C++ types cross this line.
 *
 * Data formats are the reference's own:
 *   unpacked batch : ASCII, each sequence padded with 'N' to a multiple of 8 bytes (host_batch.cpp:79-154)
 *   packed batch   : 8 bases per uint32, base k of a word in bits 31-4k..28-4k  (pack_rc_seqs.h:21-33)
 *   offsets        : BYTE offsets into the unpacked batch (multiples of 8); lens: true lengths
 *   results        : three int32[n]: score, 0-based inclusive end on the query (DP rows) and target (DP columns)
 */
#ifndef AGATHA_AMD_H
#define AGATHA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same seven fields, same order, as the reference's gasal_subst_scores (AGAThA/src/gasal.h:165-173) */
typedef struct agatha_amd_scores {
    int32_t match;        /* -m */
    int32_t mismatch;     /* -x (penalty, positive) */
    int32_t gap_open;     /* -q */
    int32_t gap_extend;   /* -r */
    int32_t slice_width;  /* -s: only decides WHEN z-drop is tested (reference semantics); any value >= 1 is computed with an exact
                             ring of anti-diagonal maxima.  The REFERENCE is only defined for s + 1 a power of two (1, 3, 7 ...):
                             its ring index is `& (8 (s + 1) - 1)` (agatha_kernel.h:29,83,295) and aliases slots otherwise, so
                             parity with it is claimed, and tested, for those values only */
    int32_t z_threshold;  /* -z: < 0 disables z-drop */
    int32_t band_width;   /* -w */
} agatha_amd_scores;

enum {
    AGATHA_AMD_OK = 0,
    AGATHA_AMD_EINVAL = -1,      /* bad argument (NULL pointer, size not a multiple of 8, n == 0 ...) */
    AGATHA_AMD_EBAND = -2,       /* band wider than the largest compiled window (agatha_amd_max_band()) */
    AGATHA_AMD_EWORKSPACE = -3,  /* workspace too small, see agatha_amd_workspace_bytes() */
    AGATHA_AMD_EHIP = -4,        /* HIP runtime error, text in agatha_amd_last_error() */
    AGATHA_AMD_ERANGE = -5       /* match * min(max lengths) -- or, with z-drop off, mismatch * max(max lengths) -- does not fit the
                                    kernel's 2^(30-K) score range (K = 8..13 by band); without length hints the pairs that do not
                                    fit get AGATHA_AMD_BAD_RESULT instead */
};

const char* agatha_amd_strerror(int code);
const char* agatha_amd_last_error(void);     /* thread-local text of the last HIP failure */
const char* agatha_amd_version(void);

/* device selection: replaces gasal_set_device (interfaces.cpp:86-116) */
int agatha_amd_device_count(void);
int agatha_amd_set_device(int device);

/* largest band_width the compiled kernels accept when sequences are longer than the band */
int agatha_amd_max_band(void);

/* bytes of device scratch agatha_amd_align() needs for up to max_n_alns pairs (replaces the reference's
 * 0.98 GB/stream global_buffer + pinned host_buffer, ctors.cpp:89-90: this is ~5 B per pair + 64 KiB up to 4096 pairs;
 * larger batches add ~136 MiB for the pairs that are suspended and resumed by another lane group when the batch is larger
 * than one round of lane groups -- a caller that passes less gets the work queue instead, never an error) */
size_t agatha_amd_workspace_bytes(uint32_t max_n_alns);
/* The same plus, when the sequences can be long (ceil(query / 8) + ceil(target / 8) >= 384 -- the debug option ck_min_steps --, or a length given as 0 =
 * unknown), the checkpoint area of the packed-int16 kernel: two slots of ~8-58 KiB per lane group in flight (at most
 * ~200 MiB), where a long pair's state is saved every eighth of its steps.  A long pair that must be started over -- z-drop
 * came into reach on a step that only tracked the maxima's values -- then goes back one checkpoint instead of to its first
 * step.  A caller that passes the smaller size gets the same results, only slower on such pairs. */
size_t agatha_amd_workspace_bytes_long(uint32_t max_n_alns, uint32_t max_query_len, uint32_t max_target_len);

/* ASCII -> packed.  Replaces the gasal_pack_kernel launch (gasal_align.cu:174-185; kernel pack_rc_seqs.h:13-53).
 * nbytes must be a multiple of 8; d_unpacked 16-byte aligned; d_packed holds nbytes/8 words. */
int agatha_amd_pack(void* stream, const uint8_t* d_unpacked, uint32_t nbytes, uint32_t* d_packed);

/* The same packing on the HOST (AVX2 when the CPU has it): for callers that ship pre-packed batches -- the reference's
 * isPacked storages (ctors.cpp:65-73, gasal_align.cu:174), which halve the H2D bytes -- and skip agatha_amd_pack().
 * Plain host pointers, nbytes a multiple of 8, h_packed holds nbytes/8 words.  Synchronous. */
int agatha_amd_pack_host(const uint8_t* h_unpacked, size_t nbytes, uint32_t* h_packed);

/* 2-bit codes + N mask (round 4; north_star's "2-bit-packed reference/query"): per 8 bases of the padded batch one
 * uint16 of codes (A 0, C 1, G 2, T 3; base k in bits 15-2k..14-2k) and one byte of mask (bit 7-k: base k is N, padding, or any
 * letter outside ACGT -- the format carries ACGT + N only): 3 bits per base over PCIe instead of 4 (agatha_amd_pack_host) or 8.
 * agatha_amd_pack2_host packs on the host (nbytes a multiple of 8; h_codes holds nbytes/8 uint16, h_nmask nbytes/8 bytes) and
 * returns the number of letters that were neither ACGT nor N and went into the mask as N (>= 0; a caller that must keep such
 * letters -- they score as mismatches against everything in the reference, gasal_kernels.h:48-50 -- ships ASCII or 4-bit words
 * instead), or a negative error code.  agatha_amd_unpack2 turns the two device arrays into the nbytes/8 4-bit words every
 * kernel reads; use it in the place of agatha_amd_pack(). */
long agatha_amd_pack2_host(const uint8_t* h_unpacked, size_t nbytes, uint16_t* h_codes, uint8_t* h_nmask);
int agatha_amd_unpack2(void* stream, const uint16_t* d_codes, const uint8_t* d_nmask, uint32_t nbytes, uint32_t* d_packed);

/* Per-sequence reverse / complement of one side of a batch, AFTER agatha_amd_pack().  Replaces the
 * gasal_reversecomplement_kernel launch (gasal_align.cu:199-213; kernel pack_rc_seqs.h:56-212).  d_ops[k] bit 0 = reverse,
 * bit 1 = complement (operation_on_seq, gasal.h:66-71); sequences with op 0 are left untouched.  Needs the unpacked ASCII
 * of the batch still resident (it re-derives the affected packed words from it). */
int agatha_amd_seq_ops(void* stream, const uint8_t* d_unpacked, uint32_t* d_packed, const uint32_t* d_lens,
                       const uint32_t* d_offsets, const uint8_t* d_ops, uint32_t n_seqs);

/* Result triple (AGATHA_AMD_BAD_RESULT, -1, -1): written for a pair whose true lengths exceed the max_*_len hints badly
 * enough that the chosen lane group cannot hold its band (correct hints, or 0 = unknown, never produce that), and, when no
 * hints were given, for a pair so long that its scores could leave the kernels' 2^(30-K) range (hundreds of kilobases at
 * wide bands; with hints such a call is refused with AGATHA_AMD_ERANGE). */
#define AGATHA_AMD_BAD_RESULT INT32_MIN

/* Sort + align one batch.  Replaces agatha_kernel_launcher (gasal_align.cu:10-23): the agatha_sort kernel,
 * its D2H / host std::sort / H2D round trip, and agatha_kernel itself.
 * max_query_len / max_target_len: upper bounds of the lengths in this batch (0 = unknown); they only let
 * short batches run on a narrower lane group and never change results.
 * Threads: one caller per (device, stream) at a time -- the call forks a helper stream that belongs to that pair (kept per device:
 * the null stream is the same handle on every GPU) and joins it again before it returns; different streams, or the same stream
 * handle on different devices, may be driven from different host threads concurrently. */
int agatha_amd_align(void* stream,
                     const uint32_t* d_packed_query, const uint32_t* d_packed_target,
                     const uint32_t* d_query_lens, const uint32_t* d_target_lens,
                     const uint32_t* d_query_offsets, const uint32_t* d_target_offsets,
                     uint32_t n_alns, uint32_t max_query_len, uint32_t max_target_len,
                     const agatha_amd_scores* scores,
                     int32_t* d_aln_score, int32_t* d_query_batch_end, int32_t* d_target_batch_end,
                     void* d_workspace, size_t workspace_bytes);

/* Start positions of the alignments agatha_amd_align() found: fills the result members the reference declares and leaves
 * NULL (query_batch_start / target_batch_start, gasal.h:89-90, res.cpp:27-28; GASAL2's WITH_START, gasal.h:36).  The same
 * banded extension is run BACKWARDS from every end cell -- on the reversed prefixes q[0..query_end], t[0..target_end],
 * z-drop off -- and the cell it ends in is where the best-scoring alignment that ends in (query_end, target_end) begins
 * (0-based inclusive, like the ends).  Call after agatha_amd_align() on the same stream, with its end arrays; the packed
 * batches must still be resident.  query_batch_bytes / target_batch_bytes: sizes of the UNPACKED layout (what
 * gasal_aln_async takes).  d_scratch: agatha_amd_starts_scratch_bytes() of device memory; d_workspace as for
 * agatha_amd_align (the same one may be used).  Costs about one more agatha_amd_align of the batch.
 * NOTE: these are the starts of a LOCAL-style trimming of the extension -- where the best-scoring alignment ending in the end
 * cell begins; it may begin after the origin and, banded around the end cell's diagonal, may leave the forward band.  They are
 * NOT the first cell of the path agatha_amd_align_traceback() reports: that path is the extension alignment itself and
 * always starts at the origin (0, 0).  A caller that prints both (`manual -S -T`) shows two different alignments of the
 * same end cell. */
size_t agatha_amd_starts_scratch_bytes(uint32_t query_batch_bytes, uint32_t target_batch_bytes, uint32_t max_n_alns);
int agatha_amd_align_starts(void* stream, const uint32_t* d_packed_query, const uint32_t* d_packed_target,
                            const uint32_t* d_query_offsets, const uint32_t* d_target_offsets, uint32_t n_alns,
                            uint32_t query_batch_bytes, uint32_t target_batch_bytes, uint32_t max_query_len, uint32_t max_target_len,
                            const agatha_amd_scores* scores, const int32_t* d_query_batch_end, const int32_t* d_target_batch_end,
                            int32_t* d_query_batch_start, int32_t* d_target_batch_start, void* d_workspace,
                            size_t workspace_bytes, void* d_scratch, size_t scratch_bytes);

/* Alignment paths: fills the result members the reference declares and never fills (cigar / n_cigar_ops, gasal.h:91-92,
 * res.cpp:27-28), in the byte format GASAL2 publishes for them: pair k's path is d_n_cigar_ops[k] bytes starting at
 * d_cigar + d_query_offsets[k] + d_target_offsets[k] (so d_cigar needs query_batch_bytes + target_batch_bytes of the
 * unpacked layout), one byte per run, (count << 2) | op with op 0 = match, 1 = mismatch, 2 = D (target base against a
 * gap), 3 = I (query base against a gap), count <= 63, longer runs split greedily from the start; the first byte is the
 * alignment's first column -- always the origin, this being an extension alignment -- and the last its end cell.
 * The call is a complete agatha_amd_align() of the batch (it also writes score / ends, bit-identical to that call) through
 * a variant of the kernel that records a 4-bit code per cell, followed by the walk back from every end cell; it replaces
 * that call rather than following it.  The path is the one the recurrence took: ties diagonal > E > F, open before extend.
 * d_n_cigar_ops[k] = 0 for an empty alignment (score 0), AGATHA_AMD_NO_PATH for a pair whose result is
 * AGATHA_AMD_BAD_RESULT or whose score came through a cell the reference's block-granular band skips (the stale-register
 * reads of agatha_kernel.h:33-35: such a score belongs to no alignment; rare, needs a path along the band edge).
 * d_scratch: agatha_amd_traceback_scratch_bytes(n_alns, max lens, scores, pairs_per_pass) of device memory: 16 bytes per pair
 * + the code area.  Every pair's codes take (row blocks + column blocks) x 32 bytes x the lane-group size, from its TRUE
 * lengths (agatha_amd_traceback_pair_bytes() is that for the longest possible pair: 8.6 MB for 10 kb x 10 kb at band 751);
 * the device packs the pairs into the area in input order and starts it over when the next pair does not fit, so a small
 * scratch (pairs_per_pass < n_alns; at least n_alns / 4096) costs passes, not correctness, and a batch of mixed lengths needs
 * far fewer passes than its longest pair suggests.  max_query_len / max_target_len are REQUIRED here (non-zero, true upper bounds):
 * the host launches the number of passes they imply without waiting for the device's plan (passes the plan does not need
 * return at once); should a pair be longer than the hints, the pairs left over get AGATHA_AMD_BAD_RESULT / AGATHA_AMD_NO_PATH,
 * as does a pair whose codes do not fit the area even alone.  d_workspace as for agatha_amd_align.
 * The recording kernel is the packed-int16 one for bands of 49..192 blocks (w = 377..1528) with scores it takes, the int32 one
 * otherwise and for what the int16 kernel leaves (other letters, N in the query, abandoned pairs): results and bytes do not
 * depend on which (debug option no_int16; agatha_amd_step_stats() after the call counts the pairs the int16 kernel started). */
#define AGATHA_AMD_NO_PATH 0xFFFFFFFFu
size_t agatha_amd_traceback_pair_bytes(uint32_t max_query_len, uint32_t max_target_len, const agatha_amd_scores* scores);
/* pairs_per_pass: 0 = the whole batch in one pass (for pairs as long as the hints) */
size_t agatha_amd_traceback_scratch_bytes(uint32_t n_alns, uint32_t max_query_len, uint32_t max_target_len,
                                          const agatha_amd_scores* scores, uint32_t pairs_per_pass);
int agatha_amd_align_traceback(void* stream,
                               const uint32_t* d_packed_query, const uint32_t* d_packed_target,
                               const uint32_t* d_query_lens, const uint32_t* d_target_lens,
                               const uint32_t* d_query_offsets, const uint32_t* d_target_offsets,
                               uint32_t n_alns, uint32_t max_query_len, uint32_t max_target_len,
                               const agatha_amd_scores* scores,
                               int32_t* d_aln_score, int32_t* d_query_batch_end, int32_t* d_target_batch_end,
                               uint8_t* d_cigar, uint32_t* d_n_cigar_ops,
                               void* d_workspace, size_t workspace_bytes, void* d_scratch, size_t scratch_bytes);

/* Optional: a hipEvent_t pair (as void*) that the NEXT agatha_amd_align() calls of this thread record directly
 * around the alignment kernel launch (excluding the sort); pass NULLs to switch it off.  Used by bench.py for
 * the per-kernel duration of the roofline line. */
void agatha_amd_set_kernel_events(void* ev_begin, void* ev_end);

/* lane-group shape the last agatha_amd_align() of this thread used (diagnostics for bench.py / DESIGN.md) */
void agatha_amd_last_config(int* lanes_per_pair, int* slots_per_lane);

/* (lanes_per_pair << 8) | slots_per_lane of the packed-int16 kernel if it was a CANDIDATE in the last
 * agatha_amd_align() of this thread (scores and band inside its domain), 0 if not.  Whether it ran is the device's
 * choice (agatha_amd_kernel_choice); debug options no_int16 / force_int16 below override it. */
int agatha_amd_last_int16_config(void);

/* Debug / A-B options of the routing inside agatha_amd_align (never needed for correct results; tests and the tuning
 * tools use them).  Process-global; the environment variable AGATHA_AMD_<NAME IN CAPITALS> gives the initial value and
 * is read ONCE, at the first use of the library -- the hot path never calls getenv.  Names:
 *   "no_int16" (1: the packed-int16 kernel is not a candidate), "force_int16" (1: it is the only candidate when the
 *   scores and the band allow it), "force_choice" (>= 0: index of the candidate that takes the plain pairs, -1 = model),
 *   "no_deal" (1: no dealt first round), "no_migrate" (1: pairs never move between lane groups), "max_blocks"
 *   (> 0: cap of the persistent grids), "mig_timeout_us" (how long a lane group waits for a pair another group has to
 *   suspend before it takes the pair over, default 50000; "mig_fresh_timeout_us", default 2000, when that group has not even
 *   started the pair: its workgroup is not resident), "mig_test_delay_us" (tests: odd lane groups start late), "no_split" (1: a
 *   batch of mixed lengths is never split between the two int16 shapes), "prio_fine", "fast_margin", "ck_min_steps", "static_ck",
 *   "ck_shift" / "ck_newer" (spacing of the int16 kernel's checkpoints; which of the two a pair goes back to), "force_split" /
 *   "lat_blocks" (experiments with the split),
 *   "timeline" (1: waves record when and where they ran, agatha_amd_timeline), "prio_slice" / "prio_duty" (the
 *   time-sliced issue priority of the two waves that share a SIMD: slice length 2^n x 10 ns, -1 = automatic, 0 = off).
 * Returns AGATHA_AMD_EINVAL for an unknown name. */
int agatha_amd_set_debug_option(const char* name, int value);
int agatha_amd_get_debug_option(const char* name, int* value);

/* Diagnostics: which candidate kernel the device chose for the plain pairs of the last agatha_amd_align() on this
 * workspace: out[0] = 0 int32 profile kernel / 1 packed-int16 kernel, out[1] = lanes per pair, out[2] = slots per lane.
 * (Candidates: the int16 kernel if scores and band allow it, the int32 kernel with the smallest lane group that holds the
 * band, and the int32 kernel with 64 lanes per pair; the device picks the one with the smallest estimated time from
 * the batch's length histogram -- latency of the longest pair against throughput over the whole batch.)  Synchronises
 * the stream. */
int agatha_amd_kernel_choice(void* stream, const void* d_workspace, uint32_t n_alns, int out[3]);

/* Diagnostics: a batch of mixed lengths may be split between the two shapes of the packed-int16 kernel (round 4; the purpose of
 * the reference's uneven bucketing and subwarp rejoining, agatha_kernel.h:113 / :365-408, re-derived): out[0] = the number of pairs
 * -- the longest ones -- that ran on the latency shape, side by side with the rest on the throughput shape agatha_amd_kernel_choice
 * reports (0: one shape took them all); out[1], out[2] = lanes per pair and slots per lane of that latency shape.  Debug option
 * "no_split" = 1 switches the split off.  Synchronises the stream. */
int agatha_amd_split_info(void* stream, const void* d_workspace, uint32_t n_alns, int out[3]);

/* Diagnostics: the preemptive schedule of the last agatha_amd_align() on this workspace.  out[0] = 1 if the packed-int16
 * throughput kernel ran the batch on a static schedule in which pairs move between lane groups (more pairs than lane
 * groups, up to a few rounds; the reference's subwarp rejoining, agatha_kernel.h:365-408, re-derived), out[1] = steps every
 * lane group executes, out[2] = lane groups used.  All 0 when the work queue was used.  Synchronises the stream. */
int agatha_amd_schedule_info(void* stream, const void* d_workspace, uint32_t n_alns, int out[3]);

/* Diagnostics: what the packed-int16 kernel's steps were in the last agatha_amd_align() on this workspace: out[0] = wave-steps
 * without anti-diagonal maxima inside the blocks ("value steps": the running maximum and the anti-diagonal maxima are bounded by
 * the last row and column of every block), out[1] = wave-steps with H : column keys ("key steps": a pair's last steps,
 * pairs that were started over), out[2] = pairs started over on key steps (z-drop came into reach on a value step, or the
 * pair ended without knowing the cell of its maximum) from their first step, out[3] = pairs started, out[15] = pairs taken back to a
 * checkpoint instead, out[24] = pairs handed to the int32 kernel instead (static schedule without checkpoints: debug option
 * static_ck = 0); out[23] = (builds without -DAGATHA16_DIAG) the value steps among out[0] on which nothing was reduced or tested (round 6, "lazy" value steps of the shapes with one
 * pair per wave: the last test answers for them; debug option lazy_max); out[4..6] why (a value step that was not calm / a key step that needed the cell / the end of a pair without it);
 * in builds with -DAGATHA16_DIAG only (tools/gpu_skew.py; zero otherwise): out[16..23] = how far back the checkpoint lay, in units of 256 steps; out[25..32] = how far the pairs that started from their first
 * step had come, in units of 512 steps, and out[33..37] why the ones beyond 1024 steps had no checkpoint to go back to (second time
 * / pair too short for checkpoints / before its second checkpoint / resumed pair: the state it was resumed from is used / slot
 * overwritten), out[38] = pairs that were suspended with an older checkpoint instead of their present state.
 * In builds WITHOUT -DAGATHA16_DIAG out[39] = rests of suspended pairs that a lane group took whole from the pool because their first part had
 * not been started yet (round 5; no work is lost by that, unlike out[14], the pairs taken over after a time-out).
 * Likewise out[38] = pairs that, having gone back to a checkpoint or to their first step, returned to value steps (round 5, "probation").
 * Synchronises the stream. */
int agatha_amd_step_stats(void* stream, const void* d_workspace, uint32_t n_alns, unsigned int out[40]);

/* Diagnostics: the packed-int16 kernel's look at the batch (round 5; DESIGN.md 3.6, "flat batches"): out[0] = pairs that have said,
 * between their 64th and 128th step, whether their score rises fast enough for a window of key steps at their end, out[1] = of those,
 * the ones that are flat (they would need more than four times the window a pair may have), out[2] = young pairs that were
 * started over on key steps because most of the batch is flat, out[3] = 10 ns ticks the launch of the throughput shape waited behind
 * the latency shape on the helper stream (split_gate_kernel; 0: no second int16 shape was a candidate), out[4] = pairs that gave up far into
 * their steps on a static schedule and were aligned by the clean-up launch of the latency shape instead of starting over in place (debug
 * option cleanup_min_steps), out[5] = positions of the sorted order that launch looked at.  Synchronises the stream. */
int agatha_amd_flat_stats(void* stream, const void* d_workspace, uint32_t n_alns, unsigned int out[6]);

/* The packed-int16 kernel's hard bounds (round 6): out[0] = pairs that left for the int32 kernel because their step counter had run past the
 * pair's last step, out[1] = ... because the state they were to be resumed from (a suspended pair, a checkpoint, a fallback) failed its check,
 * out[2] = states poisoned on purpose (debug option poison_state: tests), out[3] = suspended states counted while that option is on.  out[0] + out[1] > 0 without out[2] means that memory
 * the kernel owns was overwritten or that its state machine has a bug; the results of the call are right all the same (the int32 kernel redid
 * those pairs) and libgasal_amd / the Python binding print a warning.  Before round 6 such a state was a hung GPU (profiles/r05_v2/
 * probation_hang_probe.txt).  synchronise != 0: waits for the stream; 0: only enqueues the copy (out must then be pinned host memory that stays
 * valid until the stream has passed it: the Python binding reads it behind the results of every batch without a wait of its own). */
int agatha_amd_guard_stats(void* stream, const void* d_workspace, uint32_t n_alns, unsigned int out[4], int synchronise);

/* Diagnostics (debug option "timeline" = 1, workspace sized for > 4096 pairs): where and when every wave of the packed-int16
 * kernel ran in the last agatha_amd_align() on this workspace.  8 dwords per wave (wave = 4 * workgroup + wave in
 * workgroup): start and end in ticks of the 100 MHz real-time counter, HW_ID, XCC_ID, steps executed, pairs started, 2
 * spare.  Returns the number of waves copied (<= max_waves) or a negative code.  Synchronises the stream. */
int agatha_amd_timeline(void* stream, const void* d_workspace, uint32_t n_alns, uint32_t* out, uint32_t max_waves);

/* Diagnostics: how the last agatha_amd_align() on this workspace routed its n_alns pairs.  counts[0] = plain pairs
 * (aligned by the packed-int16 kernel when it ran, else by the int32 profile kernel), counts[1] = pairs with letters
 * outside ACGTN (compare kernel), counts[2] = pairs the int32 profile kernel took over (N in the query, or handed
 * back by the int16 kernel).  Synchronises the stream. */
int agatha_amd_pair_kinds(void* stream, const void* d_workspace, uint32_t n_alns, uint32_t counts[3]);

/* Thin device-memory helpers so that non-HIP hosts (ctypes, cgo, JNI) can drive the library without
 * linking the HIP runtime themselves.  Synchronous except the *_async copies. */
int agatha_amd_malloc(void** d_ptr, size_t bytes);
int agatha_amd_free(void* d_ptr);
int agatha_amd_host_alloc(void** h_ptr, size_t bytes);   /* pinned */
int agatha_amd_host_free(void* h_ptr);
int agatha_amd_memcpy_h2d_async(void* stream, void* d_dst, const void* h_src, size_t bytes);
int agatha_amd_memcpy_d2h_async(void* stream, void* h_dst, const void* d_src, size_t bytes);
int agatha_amd_stream_create(void** stream);
int agatha_amd_stream_destroy(void* stream);
int agatha_amd_stream_synchronize(void* stream);
int agatha_amd_stream_query(void* stream);               /* 0 = idle, 1 = busy, <0 = error */
/* timing on the launch stream (the reference's -p mode, gasal_align.cu:219-236, done per stream) */
int agatha_amd_event_create(void** event);
int agatha_amd_event_destroy(void* event);
int agatha_amd_event_record(void* event, void* stream);
int agatha_amd_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on stop */
/* work enqueued on `stream` after this call waits for `event` (as last recorded); the host does not (hipStreamWaitEvent).  libgasal_amd's
 * batch manager uses it to keep at most a few batches' align kernels on the chip at once however many host threads feed it. */
int agatha_amd_stream_wait_event(void* stream, void* event);
/* the device agatha_amd_set_device() selected for the calling thread (>= 0), or a negative error code */
int agatha_amd_get_device(void);

#ifdef __cplusplus
}
#endif
#endif /* AGATHA_AMD_H */
