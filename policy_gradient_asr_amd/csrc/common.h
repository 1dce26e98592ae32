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

// ---- the K-slabs (time ranges) of a BLSTM layer's weight-gradient products (pgasr_lstm_wgrads_streamed) ----
// Frames 0 = h_0 < h_1 < .. < h_n = T with slab sizes 16, 24, 32, 40, 48, 64, 80, 104, 128, 128, .. (x 5/4, rounded to 8, capped at 128;
// the last slab takes what is left while that is <= 1.5 sizes).  Direction 0's backward sweep walks DOWN in time, so its tiles use the
// slabs [h_j, h_j+1) as they are -- the big ones are complete early, the small ones last -- and direction 1's tiles use the mirror
// image [T - h_j+1, T - h_j).  Either way the slab that is complete after sweep step T - h_j - 1 is the (n - j)-th, so ONE list of
// publication points P_k = T - h_(n-k), k = 1..n, serves both directions.  Why shrinking slabs: a 256 x 256 tile costs a CU c us per
// 32 rows (few items in flight: latency bound) while the sweep produces 32 rows in ~1.3 us, so the products run (c s - 1.3 R) us
// behind the sweep's end for a slab of s steps with R steps after it.  Rounds 3-4 sized them for the bf16x3 kernel (c ~ 3: x 11/8,
// cap 168, <= ~50 us behind; x 13/8: 112 us; sixteen equal slabs: 185 us).  The six-product kernel of the "f32" mode costs c ~ 5.5, and
// with x 11/8 / 168 its products ran ~400 us behind the sweep's end -- holding the CUs the NEXT layer's feed wants at the start of its
// sweep, and lengthening the tail of the step.  Round 5 (tools/dev/r5_slab_sched.sh, f32 step, one box): 11/8 cap 168 9.95-9.99 ms,
// 5/4 cap 176 9.90, 5/4 cap 128 9.80, 5/4 cap 96 9.82, 11/8 cap 64 9.90 (tail 0.61 -> 0.54 ms); 5/4 cap 128 it is (13 slabs at T = 1000).
// A function of T only -- it DEFINES the summation order of these products in every mode.  (PGASR_WSLAB_*: A/B builds only, `make variant`.)
#ifndef PGASR_WSLAB_NUM
#define PGASR_WSLAB_NUM 5
#define PGASR_WSLAB_DEN 4
#define PGASR_WSLAB_CAP 128
#endif
__host__ __device__ inline int pgasr_wslab_next(int s) { const int n = ((s * PGASR_WSLAB_NUM) / PGASR_WSLAB_DEN + 4) & ~7; return n > PGASR_WSLAB_CAP ? PGASR_WSLAB_CAP : n; }
__host__ __device__ inline int pgasr_wslab_count(int T) {
    int h = 0, s = 16, n = 1;
    while (T - h > s + s / 2) { h += s; s = pgasr_wslab_next(s); ++n; }
    return n;
}
__host__ __device__ inline int pgasr_wslab_edge(int T, int i) {     // h_i, i = 0 .. n
    int h = 0, s = 16;
    for (int j = 0; j < i; ++j) {
        if (T - h <= s + s / 2) return T;
        h += s; s = pgasr_wslab_next(s);
    }
    return h;
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
    // gated mode (queue mode only; gemm_c256.hip): A lies inside the d(pre-activation) tensor of a backward sweep that is still running
    const unsigned* gate;       // the sweep's slab_done words [2 * gate_nbg] (cluster = 2 * group + direction), or nullptr
    const float* gate_base;     // row 0, column 0 of that tensor: A's row offset gives the time, the half of the row its columns lie in the direction
    int gate_T, gate_B, gate_nbg;     // frames, utterances (rows per frame), 16-utterance groups
    int tslabs;                 // != 0: the K-slabs are the TIME slabs of pgasr_wslab_edge(gate_T, .) (splitk = pgasr_wslab_count(gate_T)), taken in
                                // the order a backward sweep completes them; A's rows are rows of the tensor at gate_base (gate may be nullptr)
    int* gate_err;              // set to 1 when a wait gives up (3 s)
};
bool pgasr_internal_tn256_ok(const PgasrTn256Args& a, int tk = 32);     // tk: rows per k-step of the kernel that will run (32: t256, 16: t6)
int pgasr_internal_tn256_launch(PgasrTn256Args a, int masked_then_unmasked, hipStream_t st, const PgasrTn256Args* second = nullptr);
// the same contract in the six-product arithmetic (three bf16 planes per operand, 16-deep steps): gemm_x6.hip
int pgasr_internal_tn6_launch(PgasrTn256Args a, int masked_then_unmasked, hipStream_t st, const PgasrTn256Args* second = nullptr);

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
