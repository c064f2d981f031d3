// device_common.h -- small device helpers shared by the alignment kernels (align_kernel.hip, align16_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

namespace agatha {

#define NEG_INF2 (-16384)   // SHRT_MIN/2: the reference's -infinity (gasal_kernels.h:39)
#define N_VALUE 14u         // 'N' & 0xF (AGAThA/Makefile:4)

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax3(int a, int b, int c) { return imax(imax(a, b), c); }

// value of lane `src` (absolute lane id) -- ds_bpermute_b32, no LDS storage involved
__device__ __forceinline__ int lane_read(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }

// max over the G lanes of a group of eight values at once, result in every lane: DPP row rotations inside a 16-lane
// row, then all cross-row ds_bpermutes are issued before the first one is waited for
template <int G>
__device__ __forceinline__ void group_max8(int (&v)[8], int lane)
{
#pragma unroll
    for (int x = 0; x < 8; x++) {
        v[x] = imax(v[x], __builtin_amdgcn_update_dpp(INT_MIN, v[x], 0x121, 0xf, 0xf, true));
        v[x] = imax(v[x], __builtin_amdgcn_update_dpp(INT_MIN, v[x], 0x122, 0xf, 0xf, true));
        v[x] = imax(v[x], __builtin_amdgcn_update_dpp(INT_MIN, v[x], 0x124, 0xf, 0xf, true));
        v[x] = imax(v[x], __builtin_amdgcn_update_dpp(INT_MIN, v[x], 0x128, 0xf, 0xf, true));
    }
    if (G >= 32) {
        int o[8];
#pragma unroll
        for (int x = 0; x < 8; x++) o[x] = lane_read(v[x], lane ^ 16);
#pragma unroll
        for (int x = 0; x < 8; x++) v[x] = imax(v[x], o[x]);
    }
    if (G >= 64) {
        int o[8];
#pragma unroll
        for (int x = 0; x < 8; x++) o[x] = lane_read(v[x], lane ^ 32);
#pragma unroll
        for (int x = 0; x < 8; x++) v[x] = imax(v[x], o[x]);
    }
}

template <int GS> struct KeyBits {          // smallest K with 2^K >= 8 * (GS + 2)
    static constexpr int value = (8 * (GS + 2) <= 128) ? 7 : (8 * (GS + 2) <= 256) ? 8 : (8 * (GS + 2) <= 512) ? 9
                               : (8 * (GS + 2) <= 1024) ? 10 : (8 * (GS + 2) <= 2048) ? 11 : (8 * (GS + 2) <= 4096) ? 12 : 13;
};


// 0xFF in every byte of R that equals the 4-bit code replicated in code4 (bytes hold codes <= 15)
__device__ __forceinline__ uint32_t eq_bytes(uint32_t R, uint32_t code4)
{
    const uint32_t X = R ^ code4;
    const uint32_t ne = ((X + 0x7F7F7F7Fu) & 0x80808080u) >> 7;
    return (ne ^ 0x01010101u) * 0xFFu;
}

}  // namespace agatha
