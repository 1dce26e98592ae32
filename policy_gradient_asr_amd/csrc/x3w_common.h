// Shared by the LDS-DMA GEMM kernels (gemm_dma.hip, gemm_c256.hip): argument block, DMA / split helpers, tile order.
#pragma once
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct DmaGemmArgs {
    const float* A; const unsigned short* Whi; const unsigned short* Wlo; float* C;
    int M, N, K, lda, ldc;
    const float* bias; const float* dact_y; float slope;
    // feed-ahead mode (FEED kernels only): the consumer of C is a persistent LSTM sweep that is ALREADY RUNNING
    unsigned* queue;            // tile counter (zeroed by the host)
    const unsigned* xcc_busy;   // [8] per-XCD count of sweep clusters, or nullptr (sweeper pass: any XCD)
    unsigned* tiles_done;       // [2][mt_count]: finished column tiles per (direction half of N, row tile)
    int mt_count, nt_count, order;   // order 0: forward-sweep consumption order, 1: backward-sweep order (mirrored)
    // K in quarters (both kernels): the result is DEFINED as ((q0 + q1) + q2) + q3 with every quarter accumulated from
    // zero, so that a tile whose quarters are computed by four workgroups in parallel (the first split_tiles tiles of a
    // feed: a sweep is waiting for them, and one workgroup needs K/32 x 1.7 us for a tile) gives the same bits as a
    // tile computed by one workgroup
    int quarters;               // 1 or 4
    int split_tiles;            // FEED: tiles (in queue order) whose quarters are separate work items
    float* slabs;               // FEED: [split_tiles][4][64][512] partial accumulators
    unsigned* arrive;           // FEED: [split_tiles] quarters finished (zeroed by the host)
    int single;                 // FEED: every workgroup takes ONE work item and leaves (the head launch in front of a sweep)
    const unsigned short* Wmid; // six-product kernels (gemm_x6.hip): the middle plane of the 3-plane split (hi, mid, lo)
    int head_help;              // six-product FEED: workgroups on the sweep's own XCDs take K-split head items before they leave (0: they leave at once)
    int split8_tiles, split2_tiles;   // six-product FEED: tiles in K-eighths in front of the split_tiles quarter tiles, tiles in K-halves behind them
    int slab_count;             // six-product FEED: parked accumulator sets in `slabs` (8 split8 + 4 split + 2 split2)
};

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_base, 16, 0, 0);
}

__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi_pk, unsigned& lo_pk) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    const bf2 h = __builtin_convertvector((f2){x0, x1}, bf2);
    hi_pk = __builtin_bit_cast(unsigned, h);
    const float h0 = __uint_as_float(hi_pk << 16), h1 = __uint_as_float(hi_pk & 0xFFFF0000u);
    const bf2 l = __builtin_convertvector((f2){x0 - h0, x1 - h1}, bf2);
    lo_pk = __builtin_bit_cast(unsigned, l);
}

// 3-plane split of a pair: plane p = bf16 of what the planes before it left over (every subtraction is exact in fp32), so
// x = hi + mid + lo to 2^-24 relative -- the operand format of the fp32-faithful six-product kernels (gemm_x6.hip)
__device__ __forceinline__ void split3(float x0, float x1, unsigned& hi_pk, unsigned& mid_pk, unsigned& lo_pk) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    hi_pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){x0, x1}, bf2));
    float r0 = x0 - __uint_as_float(hi_pk << 16), r1 = x1 - __uint_as_float(hi_pk & 0xFFFF0000u);
    mid_pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){r0, r1}, bf2));
    r0 -= __uint_as_float(mid_pk << 16); r1 -= __uint_as_float(mid_pk & 0xFFFF0000u);
    lo_pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){r0, r1}, bf2));
}

// XCD-aware tile order (same remap as gemm.hip): each XCD walks a contiguous run of the
// n-fastest tile order, so the column tiles that share an A panel share an L2.
__device__ __forceinline__ void swizzled_tile(int& bx, int& by) {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned nwg = gx * gy;
    const unsigned L = blockIdx.x + gx * blockIdx.y;
    const unsigned q = nwg / 8, r = nwg % 8;
    const unsigned xcd = L % 8, i = L / 8;
    const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    bx = (int)(t % gx); by = (int)(t / gx);
}


}  // namespace
