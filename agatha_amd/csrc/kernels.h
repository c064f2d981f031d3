// kernels.h -- private interface between the C-ABI (capi.cpp) and the HIP kernels (align_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace agatha {

// field order of the reference's gasal_subst_scores (AGAThA/src/gasal.h:165-173)
struct AlignParams { int32_t match, mismatch, gap_open, gap_extend, slice_width, z_threshold, band_width; };

// one candidate kernel for the plain (kind-0) pairs of a launch; the choice is made on the device from the length
// histogram of the batch (record_kernel), so that the call stays asynchronous
struct KernelChoice {
    int kind;                      // 0 = int32 profile kernel, 1 = packed-int16 kernel
    int G, S;                      // lanes per pair, slots per lane
    float t_lat, t_load;           // microseconds per step of one pair: alone on its SIMD / with the chip full
    int capacity;                  // lane groups of its persistent grid
};

struct AlignLaunch {
    const uint32_t *packed_q, *packed_t, *qlens, *tlens, *qoffs, *toffs, *order;
    int n;
    unsigned int* queue;
    uint8_t* exotic;               // per pair kind (low 7 bits; bit 7 = plain pair whose query holds an N): 0 = plain, 1 = holds letters outside ACGTN (compare kernel),
                                   // 2 = abandoned by the packed-int16 kernel (int32 profile kernel takes it),
                                   // 3 = scores out of the kernels' range (no kernel takes it: AGATHA_AMD_BAD_RESULT)
                                   // 4 = (traceback pass) done by the int16 kernel; 5 = gave up on the static schedule, taken by the clean-up launch of the int16 latency shape
    int ncand;                     // candidates for the kind-0 pairs, in launch order
    KernelChoice cand[4];
    int* choice;                   // device: index of the candidate that takes the kind-0 pairs
    int force_choice;              // >= 0: candidate index to use regardless of the model (AGATHA_AMD_FORCE_CHOICE, experiments)
    int force_split;               // > 0: split the batch between the two int16 shapes with this many pairs on the latency shape, whatever the model says (tests, fuzzers)
    int ck_newer;                  // int16 kernel: 1 = a pair goes back to the newer of its two checkpoints when its bound has risen enough behind it (shapes without bookkeeping), to the checkpoint before "keep" when it has not (shapes with it); 0 = round 3's rule (debug option ck_newer)
    int ck_shift;                  // int16 kernel: 2^(ck_shift - clz(steps of the pair)) steps between two checkpoints (debug option ck_shift)
    int lat_blocks;                // > 0: workgroups of the latency shape that take the long pairs of a split batch (debug option lat_blocks, experiments; 0: no cap)
    float* totals;                 // device: [0] sum of steps over the batch, [1] steps of the longest pair (sort_scan_kernel)
    unsigned int* kind_counts;     // device: [0] pairs of kind 1, [1] pairs of kind 2 (kernels with nothing to do return at once)
    int force_cmp;                 // 1 = scores do not fit the byte profile: compare path for every pair
    long long score_limit;         // 2^(30-K): scores (and, with z-drop off, their negatives) must stay below it in the H << K keys
    int32_t *score, *qend, *tend;
    AlignParams p;
    int num_cus;
    int max_blocks_override;       // > 0: cap of the persistent grid (tuning knob, AGATHA_AMD_MAX_BLOCKS)
    int no_deal;                   // 1 = every pair from the queue, no dealt first round (AGATHA_AMD_NO_DEAL, A/B runs)
    const AlignLaunch* self_dev;   // device copy of this record (lives in the workspace)
    // ---- preemptive static schedule of the packed-int16 throughput shape (pairs migrate between lane groups) ----
    int mig_enabled;               // workspace holds the areas below and the option is on (2: use the schedule whenever it is possible: A/B runs)
    int mig_slots;                 // lane groups of the full persistent grid of candidate 0 (its capacity)
    uint32_t* cum;                 // device: [n + 1] exclusive prefix sums of the pairs' step counts in sorted order
    int* sched;                    // device: [0] 1 = static schedule in force, [1] T = steps per lane group, [2] groups used
    int* mig_state;                // device: [mig_slots + 1] state of the pair that crosses each group boundary
    int* mig_perm;                 // device: [mig_slots] the interval of the line of pairs each physical lane group owns (schedule_kernel; nullptr: its own index)
    uint32_t* mig_buf;             // device: suspended state of those pairs, mig_slot_dwords per boundary
    int mig_slot_dwords;           // (stride of a boundary's states in mig_buf)
    int* mig_late;                 // device: the pool of the suspended pairs' rests: [0] head (atomic), [1] how many, [2 ..] the boundaries, longest rest first (schedule_kernel; nullptr: no pool, every lane group resumes the pair that crosses out of its own interval)
    uint32_t* mig_rest;            // device: [mig_slots + 1] scratch of schedule_kernel: the rest of the pair across every boundary
    int mig_identity;              // 1: lane group g owns interval g (debug option mig_identity: the schedule without the permutation, A/B runs)
    int mig_fallback;              // 1: the stride holds TWO states, the suspended one and a fallback (the older checkpoint of a pair that was suspended with a bound for its maximum)
    unsigned int mig_timeout_ticks;  // 100 MHz ticks a group waits for a pair that another group is RUNNING to be suspended before it takes the pair over
    unsigned int mig_fresh_timeout_ticks;   // ... and for a pair the other group has not even started (its workgroup is not resident)
    uint32_t* timeline;            // device (debug option "timeline"): per wave of the int16 kernel {start, end (100 MHz ticks), HW_ID, XCC_ID, steps, pairs}
    int prio_slice_bits;           // > 0: the two waves of a SIMD take turns at high issue priority, in slices of 2^bits ticks of the 100 MHz clock
    int* simd_steps;               // device: [2 workgroups][CU][SIMD] step counts the waves of a dealt single round tell each other
    int prio_duty;                 // of every 16 slices, the wave in slot 0 of its SIMD is the favoured one in this many
    int prio_fine;                 // ... plus this many quarters of a slice (debug option prio_fine: the arbiter is not quite even-handed)
    unsigned int mig_test_delay_ticks;   // tests: odd lane groups sleep this long before they start (forces the take-over)
    // packed-int16 kernel, the hard bounds of round 6 (align16_step_maxima.inc, align16_acquire.inc): [0] pairs handed to the int32 kernel because their
    // step counter had run past the pair's last step, [1] ... because the state they were to be resumed from (a suspended pair, a checkpoint, a
    // fallback) failed its check (this pair, this launch, 0 <= step <= steps of the pair, the slice counter in range), [2] saved states poisoned
    // by the debug option poison_state (tests).  Non-zero [0] / [1] means memory the kernel owns was corrupted or its state machine has a bug:
    // the results are still right (the int32 kernel redoes those pairs), and the host says so loudly.
    unsigned int* guard_stats;
    int tb_value_steps;            // traceback pass on the int16 kernel: 1 = value steps as in the score-only kernel (round 6), 0 = key steps only (debug option tb_value_steps)
    int lazy_max;                  // packed-int16 kernel, one pair per wave, value steps: a calm test that passed with room to spare answers for at most this many steps behind it (debug option lazy_max; 0 = every value step is tested)
    int poison_state;              // tests (debug option poison_state): n > 0 = the n-th suspended state of the launch is written with a garbage step counter (n + 1000: behind a flag that lets it pass the resume check, so that only the bound inside the step loop can end the pair)
    unsigned int* step_stats;      // device: [0] value wave-steps, [1] key wave-steps, [2] pairs started over, [3] pairs started, [23] lazy value wave-steps (int16 kernel)
    uint32_t* ck_buf;              // device: checkpoints of the int16 kernel's long pairs, two slots of a suspended pair's size per lane group (nullptr: none)
    unsigned long long ck_dwords;  // dwords of that area ...
    unsigned long long ck_lat_off; // ... of which the first ck_lat_off belong to the shapes with fewer than 64 lanes per pair, the rest to the others (a batch split by
                                   // length runs one of each at the same time); a shape <G, P> has room for (its part) / (2 * mig_fields(P) * G) lane groups
    int ck_min_steps;              // pairs of fewer steps take no checkpoints
    int fast_anchor;               // 1: the window of key steps starts before the corner of the shorter sequence (default); 0: before the pair's last step (experiments)
    int static_ck;                 // on a static schedule the three-register-pair shapes: 1 = checkpoints and going back to them in place, 0 = none, a pair that must start over goes to the int32 kernel
    int launch_id;                 // a number per agatha_amd_align call (24 bits are stored with every checkpoint: a slot's content must be this call's)
    int cleanup_ok, cleanup_min_steps;   // packed-int16 kernel, static schedule: a pair that must start from its first step at step g of its t, 2 g > t + cleanup_min_steps, leaves for the clean-up launch of the latency shape (kind 5; debug option cleanup_min_steps, 0 = never)
    int probation;                 // packed-int16 kernel: 1 = a pair that went back to a checkpoint returns to value steps once z-drop is out of reach again (debug option probation; 0 = key steps for good, until round 5)
    int flat_percent;              // ... when more than this share of the pairs that have said so are flat (debug option flat_percent)
    int flat_detect;               // packed-int16 kernel: 1 = a batch whose pairs are mostly flat (their score hardly rises) runs on key steps (debug option flat_detect)
    int win_prior;                 // packed-int16 kernel: the adaptive part of the window of key steps a pair STARTS with: what a read with 15 % errors needs at this scoring (capi.cpp)
    int win_cap_min, win_cap_div;  // packed-int16 kernel: the adaptive part of a pair's window of key steps is at most max(win_cap_min, steps / win_cap_div) steps
    int fast_margin;               // packed-int16 kernel: > 0 = value steps (align16_body.inc) except in a pair's last fast_margin steps; 0 = key steps only
    // ---- traceback pass (align_tb.hip): the compare kernel also records a 4-bit code per computed cell ----
    uint32_t* tb_codes;            // device: the code area; pair k's words start at tb_off[k]: 8 words (one per block row, a
                                   // nibble per column) for (step, column block mod G*S) at ((step * G*S) + slot) * 8
    const unsigned long long* tb_off;   // device: word offset of each pair's codes inside its pass (tb_plan_kernel)
    const int* tb_pass;            // device: the pass a pair belongs to (the code area is reused pass after pass); -1 = its codes
                                   // do not fit the area at all (AGATHA_AMD_BAD_RESULT)
    const int* tb_plan;            // device: [0] number of passes
    // layout of a step's G*S*8 code words: 0 = block-major (slot, row); L > 0 = the int16 kernel's, lane-major for its L lanes per
    // pair: [register pair][half][rows 0-3 | 4-7][lane][4 words], so that one store instruction of a lane group is contiguous
    int tb_lanes;
};

// states of a boundary (the pair whose steps are split between lane groups b - 1 and b)
enum { MIG_FRESH = 0, MIG_RUNNING = 1, MIG_SAVED = 2, MIG_DONE = 3, MIG_STOLEN = 4 };
// dwords of suspended state per lane of the packed-int16 kernel <G, P>: RC, H, F, XH, CORNER of the P register pairs, the 7
// carried accumulators, the lane's (2P + 1) * 9 sixteen-bit E hand-off values (two per dword), 10 per-pair scalars
constexpr int mig_fields(int P) { return P * 26 + 7 + ((2 * P + 1) * 9 + 1) / 2 + 10; }
// largest `spread` (how far below an anti-diagonal maximum an in-band cell can be) the packed-int16 kernel is offered for
constexpr int kAlign16MaxSpread = 16000;
// ... and the largest scores (agatha16_scores_ok); the kernel's one 32-bit subtract of two packed halves rests on them (align16_block.inc, row_cells8)
constexpr int kAlign16MaxMatch = 16, kAlign16MaxMismatch = 32, kAlign16MaxGapOpen = 64, kAlign16MaxGapExtend = 16;
// what a pair costs its lane group beyond its own steps (finding it, loading its lengths and reference words, building the
// score profiles: ~4 dependent memory round trips), in steps, per lane group of the wave -- every start stalls all the
// groups of its wave, so a pair costs its group kMigPairOverheadSteps * (64 / G) steps; the schedule counts that, so that
// lane groups holding many tiny pairs are not the last to finish
// checkpoint slots per lane group in the checkpoint area (every shape's stride).  (Round 6: a RING of eight slots 64 steps apart, the CPU
// model's favourite of round 5, was run on the chip -- profiles/r06_v0/ring8_*: reads with a 350-base burst 35.2 against 33.5 ms, the clean
// batch + 1 %, the flat-batch test fails (6 364 pairs go back to a checkpoint) -- and taken out again; two slots it is.)
constexpr int kCkSlots16 = 2;
constexpr int kMigPairOverheadSteps = 4;
constexpr int kMigMaxSlots = 16384;
constexpr int kSimdStepsInts = 2 * 4 * 1024;          // (up to 1024 CUs)
constexpr int kTimelineWaves = 4096, kTimelineDwords = 8;
constexpr size_t kMigBufBytes = (size_t)136 << 20;      // (two states of <= 8.2 KB per lane group boundary, 8 192 of them)

// window_blocks = blocks that can be live on one block-anti-diagonal.  plan_align fills L.cand / L.ncand (int16 kernel if
// usable and not disabled, int32 throughput shape = smallest (G, S) covering the window, int32 latency shape = 64 lanes
// per pair with fewer slots); launch_align launches them (+ the compare kernel) in that order.
hipError_t plan_align(AlignLaunch& L, int window_blocks, bool disable16, bool force16);
// (aux: a second stream for the latency shape of the packed-int16 kernel, so that a batch split by length runs its two shapes
//  side by side; fork / join: two events the caller owns; nullptr: everything on st)
hipError_t launch_align(const AlignLaunch& L, int window_blocks, int* G_out, int* S_out, hipStream_t st, hipStream_t aux = nullptr,
                        hipEvent_t fork = nullptr, hipEvent_t join = nullptr);
int max_window_blocks();
// packed-int16 kernel (align16_kernel.hip): the (G, P) of its throughput shape for this window and, if there is one with
// fewer blocks per lane, of its latency shape (64 lanes per pair; *GL = 0 if none); and its launcher
bool align16_config(const AlignParams& p, int window_blocks, int* G, int* P, int* GL, int* PL);
hipError_t launch_align16(const AlignLaunch& L, int G, int P, int kid, hipStream_t st);
// Traceback pass on the int16 kernel: align16_tb_config says whether a shape with exactly `group_slots` slots exists for these
// scores (the code layout of a pass is shared with the int32 traceback kernel); launch_align16_tb runs the pairs of `pass`
// that have plain letters and no N in the query, and marks the ones it finished (kind 4 in L.exotic).
bool align16_tb_config(const AlignParams& p, int window_blocks, int group_slots, int* G, int* P);
hipError_t launch_align16_tb(const AlignLaunch& L, int G, int P, int pass, hipStream_t st);
int key_bits_for_window(int window_blocks);
// writes L into *rec on the device (by-value kernel argument: no host-memory lifetime to care about) and zeroes *L.queue
hipError_t launch_exotic(const AlignLaunch& L, hipStream_t st);
// (hist: the sort's cumulative length histogram, for the split of a mixed batch between the two int16 shapes; nullptr / 0: none)
hipError_t launch_record(const AlignLaunch& L, AlignLaunch* rec, hipStream_t st, const uint32_t* hist = nullptr, uint32_t nbuckets = 0,
                         int allow_split = 0);
// step counts, their prefix sums and the decision static / dynamic (after launch_sort and launch_exotic)
hipError_t launch_schedule(const AlignLaunch& L, hipStream_t st);
// dwords of suspended state per lane of the packed-int16 kernel <G, P>
int align16_mig_fields(int P);
hipError_t launch_sort(const uint32_t* qlens, const uint32_t* tlens, int n, uint32_t* hist, uint32_t nbuckets,
                       uint32_t* order, float* totals, hipStream_t st);
hipError_t launch_seq_ops(const uint8_t* unpacked, uint32_t* packed, const uint32_t* lens, const uint32_t* offsets,
                          const uint8_t* ops, uint32_t n, hipStream_t st);
hipError_t launch_reverse_prefix(const uint32_t* packed, uint32_t* rev, const uint32_t* offsets, const int32_t* ends,
                                 uint32_t* rev_lens, uint32_t n, hipStream_t st);
hipError_t launch_starts(const int32_t* qend, const int32_t* tend, const int32_t* bq, const int32_t* bt, int32_t* qstart,
                         int32_t* tstart, uint32_t n, hipStream_t st);
// Traceback pass (align_tb.hip).  tb_group_slots: G*S of the shape it uses for this window (0: band too wide);
// tb_key_bits: the K of that shape (score range check); launch_align_tb: the compare kernel with code recording, for every
// pair of the given pass (L.force_cmp must be 1, L.tb_codes set); launch_backtrace: one wave per pair of that pass walks
// the codes from (qend, tend) to the origin and writes GASAL2-style bytes at cigar + qoffs[pair] + toffs[pair].
int tb_group_slots(int window_blocks);
int tb_key_bits(int window_blocks);
// launch_tb_plan: sizes every pair's code area from its true lengths and packs the pairs, in input order, into passes over
// cap_words words, at most max_passes of them (off / pass / plan as in AlignLaunch; pairs that do not fit get AGATHA_AMD_BAD_RESULT).
hipError_t launch_tb_plan(const AlignLaunch& L, int group_slots, unsigned long long cap_words, int max_passes,
                          unsigned long long* off, int* pass, int* plan, hipStream_t st);
hipError_t launch_align_tb(const AlignLaunch& L, int window_blocks, int pass, hipStream_t st);
hipError_t launch_backtrace(const AlignLaunch& L, int group_slots, int pass, uint8_t* cigar, uint32_t* n_ops, hipStream_t st);
hipError_t launch_pack(const uint8_t* unpacked, uint32_t nbytes, uint32_t* packed, hipStream_t st);
// 2-bit codes + N mask (one uint16 + one byte per 8 bases) -> nwords 4-bit words
hipError_t launch_unpack2(const uint16_t* codes, const uint8_t* nmask, uint32_t nwords, uint32_t* packed, hipStream_t st);

}  // namespace agatha
