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
    uint8_t* exotic;               // per pair kind: 0 = plain, 1 = holds letters outside ACGTN (compare kernel),
                                   // 2 = abandoned by the packed-int16 kernel (int32 profile kernel takes it)
    int ncand;                     // candidates for the kind-0 pairs, in launch order
    KernelChoice cand[4];
    int* choice;                   // device: index of the candidate that takes the kind-0 pairs
    int force_choice;              // >= 0: candidate index to use regardless of the model (AGATHA_AMD_FORCE_CHOICE, experiments)
    float* totals;                 // device: [0] sum of steps over the batch, [1] steps of the longest pair (sort_scan_kernel)
    unsigned int* kind_counts;     // device: [0] pairs of kind 1, [1] pairs of kind 2 (kernels with nothing to do return at once)
    int force_cmp;                 // 1 = scores do not fit the byte profile: compare path for every pair
    int32_t *score, *qend, *tend;
    AlignParams p;
    int num_cus;
    int max_blocks_override;       // > 0: cap of the persistent grid (tuning knob, AGATHA_AMD_MAX_BLOCKS)
    int no_deal;                   // 1 = every pair from the queue, no dealt first round (AGATHA_AMD_NO_DEAL, A/B runs)
    const AlignLaunch* self_dev;   // device copy of this record (lives in the workspace)
};

// window_blocks = blocks that can be live on one block-anti-diagonal.  plan_align fills L.cand / L.ncand (int16 kernel if
// usable and not disabled, int32 throughput shape = smallest (G, S) covering the window, int32 latency shape = 64 lanes
// per pair with fewer slots); launch_align launches them (+ the compare kernel) in that order.
hipError_t plan_align(AlignLaunch& L, int window_blocks, bool disable16, bool force16);
hipError_t launch_align(const AlignLaunch& L, int window_blocks, int* G_out, int* S_out, hipStream_t st);
int max_window_blocks();
// packed-int16 kernel (align16_kernel.hip): the (G, P) of its throughput shape for this window and, if there is one with
// fewer blocks per lane, of its latency shape (64 lanes per pair; *GL = 0 if none); and its launcher
bool align16_config(const AlignParams& p, int window_blocks, int* G, int* P, int* GL, int* PL);
hipError_t launch_align16(const AlignLaunch& L, int G, int P, int kid, hipStream_t st);
int key_bits_for_window(int window_blocks);
// writes L into *rec on the device (by-value kernel argument: no host-memory lifetime to care about) and zeroes *L.queue
hipError_t launch_exotic(const AlignLaunch& L, hipStream_t st);
hipError_t launch_record(const AlignLaunch& L, AlignLaunch* rec, hipStream_t st);
hipError_t launch_sort(const uint32_t* qlens, const uint32_t* tlens, int n, uint32_t* hist, uint32_t nbuckets,
                       uint32_t* order, float* totals, hipStream_t st);
hipError_t launch_seq_ops(const uint8_t* unpacked, uint32_t* packed, const uint32_t* lens, const uint32_t* offsets,
                          const uint8_t* ops, uint32_t n, hipStream_t st);
hipError_t launch_pack(const uint8_t* unpacked, uint32_t nbytes, uint32_t* packed, hipStream_t st);

}  // namespace agatha
