// cuda_shim.h -- just enough CUDA vocabulary to run the reference's agatha_kernel.h on a CPU.
//
// TEST INFRASTRUCTURE, build-container only.  This is an EMULATION of the CUDA execution model
// (32 cooperative fibres per warp, see shim_driver.cpp), not a CUDA toolchain: the reference is
// CUDA-only and cannot be built natively in this image.  It exists to (1) cross-check the plain-C
// oracle (oracle/agatha_oracle.c) against the reference kernel's own source text and (2) generate
// the golden vectors committed under tests/golden/.  Nothing here ships or runs on the GPU box.
#pragma once
#include <stdint.h>
#include <limits.h>
#include <algorithm>

struct short2 { short x, y; };
static inline short2 make_short2(int x, int y) { short2 r; r.x = (short)x; r.y = (short)y; return r; }
struct uint4 { unsigned x, y, z, w; };

// result struct: 7 pointers, as the reference's gasal.h:85-94
struct gasal_res { int32_t *aln_score, *query_batch_end, *target_batch_end, *query_batch_start, *target_batch_start;
                   uint8_t *cigar; uint32_t *n_cigar_ops; };
typedef struct gasal_res gasal_res_t;

#define __global__
#define __shared__
#define __constant__ static

struct shim_dim { unsigned x; };
extern shim_dim blockIdx, blockDim, gridDim;
shim_dim shim_thread_idx();
#define threadIdx (shim_thread_idx())

static inline int max(int a, int b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }

void shim_syncwarp();
unsigned shim_activemask();
unsigned shim_match_any(unsigned mask, int v);
int shim_reduce_max(unsigned mask, int v);
void shim_yield();
#define __syncwarp() shim_syncwarp()
#define __activemask() shim_activemask()
#define __match_any_sync(m, v) shim_match_any((m), (v))
#define __reduce_max_sync(m, v) shim_reduce_max((m), (v))
#define SHIM_YIELD() shim_yield()

// device "constants" (reference gasal_kernels.h:29-36) and scoring macros (ibid. :38-50, N_PENALTY=1 build)
static int32_t _cudaGapO, _cudaGapOE, _cudaGapExtend, _cudaMatchScore, _cudaMismatchScore,
               _cudaSliceWidth, _cudaZThreshold, _cudaBandWidth;
#define MINUS_INF2 (SHRT_MIN / 2)
#define N_VALUE (0x4E & 0xF)
#define DEV_GET_SUB_SCORE_GLOBAL(score, rbase, gbase) \
    score = ((rbase) == (gbase)) ? _cudaMatchScore : -_cudaMismatchScore; \
    score = (((rbase) == N_VALUE) || ((gbase) == N_VALUE)) ? -1 : score;
