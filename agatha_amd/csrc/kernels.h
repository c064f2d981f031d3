// kernels.h -- private interface between the C-ABI (capi.cpp) and the HIP kernels (align_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace agatha {

// field order of the reference's gasal_subst_scores (AGAThA/src/gasal.h:165-173)
struct AlignParams { int32_t match, mismatch, gap_open, gap_extend, slice_width, z_threshold, band_width; };

struct AlignLaunch {
    const uint32_t *packed_q, *packed_t, *qlens, *tlens, *qoffs, *toffs, *order;
    int n;
    unsigned int* queue;
    uint8_t* exotic;               // per pair kind: 0 = plain, 1 = holds letters outside ACGTN (compare kernel),
                                   // 2 = abandoned by the packed-int16 kernel (int32 profile kernel takes it)
    int use16;                     // 1 = the packed-int16 kernel runs first and takes the kind-0 pairs
    int force_cmp;                 // 1 = scores do not fit the byte profile: compare path for every pair
    int32_t *score, *qend, *tend;
    AlignParams p;
    int num_cus;
    int max_blocks_override;       // > 0: cap of the persistent grid (tuning knob, AGATHA_AMD_MAX_BLOCKS)
    const AlignLaunch* self_dev;   // device copy of this record (lives in the workspace)
};

// window_blocks = blocks that can be live on one block-anti-diagonal; picks the smallest (G, S) covering it
hipError_t launch_align(const AlignLaunch& L, int window_blocks, int* G_out, int* S_out, hipStream_t st);
int max_window_blocks();
// packed-int16 kernel (align16_kernel.hip): usable for these scores / this window?  launch (kind-0 pairs only)
bool align16_available(const AlignParams& p, int window_blocks);
int align16_group_capacity(const AlignParams& p, int window_blocks, int num_cus);
bool launch_align16(const AlignLaunch& L, int window_blocks, int* G_out, int* S_out, hipStream_t st, hipError_t* err);
int key_bits_for_window(int window_blocks);
// writes L into *rec on the device (by-value kernel argument: no host-memory lifetime to care about) and zeroes *L.queue
hipError_t launch_exotic(const AlignLaunch& L, hipStream_t st);
hipError_t launch_record(const AlignLaunch& L, AlignLaunch* rec, hipStream_t st);
hipError_t launch_sort(const uint32_t* qlens, const uint32_t* tlens, int n, uint32_t* hist, uint32_t nbuckets,
                       uint32_t* order, hipStream_t st);
hipError_t launch_seq_ops(const uint8_t* unpacked, uint32_t* packed, const uint32_t* lens, const uint32_t* offsets,
                          const uint8_t* ops, uint32_t n, hipStream_t st);
hipError_t launch_pack(const uint8_t* unpacked, uint32_t nbytes, uint32_t* packed, hipStream_t st);

}  // namespace agatha
