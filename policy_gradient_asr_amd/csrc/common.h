// Shared helpers for the gfx950 kernels of libpgasr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "pgasr_hip.h"

#define PGASR_WAVE 64

// hipGetLastError() is sticky per thread: clear whatever an earlier, unrelated runtime call left
// behind (e.g. PyTorch's start-up probes) before a launch, so the check after it reports THIS launch.
#define PGASR_LAUNCH_KERNEL(...)                               \
    do {                                                       \
        (void)hipGetLastError();                               \
        hipLaunchKernelGGL(__VA_ARGS__);                       \
    } while (0)

#define PGASR_CHECK_LAUNCH()                                                                  \
    do {                                                                                      \
        hipError_t e__ = hipGetLastError();                                                   \
        if (e__ != hipSuccess) {                                                              \
            if (getenv("PGASR_DEBUG"))                                                        \
                fprintf(stderr, "[pgasr] %s:%d launch failed: %s\n", __FILE__, __LINE__,      \
                        hipGetErrorString(e__));                                              \
            return PGASR_ERR_LAUNCH;                                                          \
        }                                                                                     \
    } while (0)

static inline size_t pgasr_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#define PGASR_NEG_INF (-INFINITY)

// ---- wave-level reductions over all 64 lanes (deterministic butterfly order) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- Philox4x32-10 (Salmon et al. 2011); same constants as oracle/decode_ref.py ----
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---- internal (not part of the C ABI): the 256 x 256-tile TN product of gemm_dma.hip, reached through pgasr_gemm_f32 ----
// partial[z][M][N] = alpha * sum_{k in slab(z)} A_b[k][m] * B_b[k][n],  z = b * splitk + s, slab(z) = [s*kper, min(K, (s+1)*kper))
// bf16x3 arithmetic.  Requirements (checked by pgasr_internal_tn256_ok): M % 256 == 0, N % 256 == 0, K % 32 == 0,
// kper % 32 == 0, lda % 4 == 0, ldb % 4 == 0, batch strides % 4 == 0, 16-byte aligned A / B.
struct PgasrTn256Args {
    const float* A; const float* B; float* partial;
    int M, N, K, lda, ldb;
    long long sA, sB;
    int batch, splitk, kper;
    float alpha;
    unsigned* queue;            // zeroed word: work items are drawn from it (queue mode), or nullptr (item = blockIdx.x)
    const unsigned* xcc_busy;   // queue mode: workgroups on an XCD whose word is non-zero take no item
    int diag;                   // diagnostic variants (PGASR_TN_DIAG, results invalid): bit 0 no loads / conversion in the loop, bit 1 no MFMA phase
};
bool pgasr_internal_tn256_ok(const PgasrTn256Args& a);
int pgasr_internal_tn256_launch(PgasrTn256Args a, int masked_then_unmasked, hipStream_t st);

// ---- internal: the 8-wave 256 x 256 x3w kernel of gemm_c256.hip (cooperative A split), reached through pgasr_gemm_x3w_f32 / _feed_f32 ----
struct PgasrX3cArgs {
    const float* A; const unsigned short* Whi; const unsigned short* Wlo; float* C;
    int M, N, K, lda, ldc;
    const float* bias; const float* dact_y; float slope;
    int feed;                   // 0: plain launch (one workgroup per tile); 1: persistent feed-ahead launch (two passes)
    unsigned* queue; const unsigned* xcc_busy; unsigned* tiles_done;
    int mt_count, nt_count, order, quarters, split_tiles;
    float* slabs; unsigned* arrive;
};
size_t pgasr_internal_x3c_slab_bytes();
int pgasr_internal_x3c_launch(const PgasrX3cArgs& a, hipStream_t st);
