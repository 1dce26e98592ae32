// Activation x weight GEMM of the path's two big shapes -- the LSTM input projections
// X(T*B, in) W_ih^T (model.py:39-44) and their input gradients dG(T*B, 8H) W_ih -- as
//     C[M,N] = A[M,K] * W[N,K]^T  (+ bias[n]) (* leaky'(dact_y[m,n]))
// with fp32-grade accuracy at bf16 MFMA rate (3-term split hi*hi + hi*lo + lo*hi, fp32 accumulate,
// like gemm.hip's bf16x3 kernel) and a load path built around LDS-DMA:
//
//   * W arrives PRE-SPLIT into bf16 hi / lo planes [N][K] (pgasr_split_bf16_planes, once per step
//     per weight: 4 MB), so its tiles go global -> LDS untouched;
//   * A stays fp32 in HBM (it is an activation: its producer writes it once) and its raw fp32
//     tile goes global -> LDS untouched as well; the hi/lo split happens on the MFMA fragment a lane
//     has just read (v_cvt_pk_bf16_f32), in the shadow of the previous MFMAs;
//   * no staging registers at all: every tile is moved by global_load_lds_dwordx4, three stages
//     deep (two tiles in flight across the single raw s_barrier per k-tile, counted s_waitcnt
//     vmcnt(6); cdna_hip_programming.md "Pipelining across barriers"), which the register-staged
//     kernel could not afford (it spent 42 % of its wave cycles in s_waitcnt);
//   * 256x128x32 tile, 8 waves as 4(M) x 2(N), 64x64 per wave = 2x2 v_mfma_f32_32x32x16_bf16
//     tiles x 3 terms x 2 k-steps = 24 MFMAs per wave per k-tile; 144 KB of LDS, one workgroup
//     per CU, two waves per SIMD;
//   * LDS images are lane-linear per wave-instruction (a DMA constraint), so bank conflicts are
//     removed by swizzling the per-lane SOURCE address and the fragment read address alike:
//     A rows are 128 B (8 chunks of 16 B, chunk ^= (row>>1)&7), plane rows 64 B (4 chunks,
//     chunk ^= (row>>2)&3): any 16 consecutive rows at one logical chunk cover all 16 slots of a
//     256-byte bank row.
//
// Preconditions (else PGASR_ERR_UNSUPPORTED; callers fall back to pgasr_gemm_f32):
// K % 32 == 0, N % 128 == 0, lda % 4 == 0, 16-byte aligned A / planes.  M is arbitrary (rows are
// clamped on load and masked on store).
#include "common.h"
#include "x3w_common.h"
#include <type_traits>

namespace {

constexpr int TM = 256, TN = 128, TK = 32, NST = 3, DMA_THREADS = 512;
constexpr int A_BYTES = TM * TK * 4;                 // 32 KB raw fp32
constexpr int P_BYTES = TN * TK * 2;                 // 8 KB per bf16 plane
constexpr int STAGE_BYTES = A_BYTES + 2 * P_BYTES;   // 48 KB

// Fragments of one 16-deep k-step straight out of LDS: 2 row tiles x 2 chunks of fp32 A, 2 column tiles x (hi, lo)
// of W.  Inline asm because hipcc puts s_waitcnt vmcnt(0) in front of any ds_read it can see after an LDS-DMA
// (checked in the .s, with run-time and with compile-time stage indices), draining the tile meant to stay in flight.
struct RawFrag { u32x4_t a[2][2]; u32x4_t bh[2], bl[2]; };
__device__ __forceinline__ void frag_read(RawFrag& r, unsigned pa0, unsigned pa1, unsigned pb0, unsigned pb1) {
    static_assert(P_BYTES == 8192, "plane offset is spelled in the asm below");
    asm volatile("ds_read_b128 %0, %8\n\t"
                 "ds_read_b128 %1, %9\n\t"
                 "ds_read_b128 %2, %10\n\t"
                 "ds_read_b128 %3, %11\n\t"
                 "ds_read_b128 %4, %12\n\t"
                 "ds_read_b128 %5, %12 offset:8192\n\t"
                 "ds_read_b128 %6, %13\n\t"
                 "ds_read_b128 %7, %13 offset:8192"
                 : "=&v"(r.a[0][0]), "=&v"(r.a[0][1]), "=&v"(r.a[1][0]), "=&v"(r.a[1][1]),
                   "=&v"(r.bh[0]), "=&v"(r.bl[0]), "=&v"(r.bh[1]), "=&v"(r.bl[1])
                 : "v"(pa0), "v"(pa0 ^ 16u), "v"(pa1), "v"(pa1 ^ 16u), "v"(pb0), "v"(pb1)
                 : "memory");
}
__device__ __forceinline__ void frag_wait(RawFrag& r) {     // claims the registers the reads above are filling
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(r.a[0][0]), "+v"(r.a[0][1]), "+v"(r.a[1][0]), "+v"(r.a[1][1]),
                   "+v"(r.bh[0]), "+v"(r.bl[0]), "+v"(r.bh[1]), "+v"(r.bl[1])
                 :: "memory");
}
__device__ __forceinline__ void split8(const u32x4_t& r0, const u32x4_t& r1, bf16x8_t& hi, bf16x8_t& lo) {
    u32x4_t h, l;
    unsigned a, b;
    split2(__uint_as_float(r0.x), __uint_as_float(r0.y), a, b); h.x = a; l.x = b;
    split2(__uint_as_float(r0.z), __uint_as_float(r0.w), a, b); h.y = a; l.y = b;
    split2(__uint_as_float(r1.x), __uint_as_float(r1.y), a, b); h.z = a; l.z = b;
    split2(__uint_as_float(r1.z), __uint_as_float(r1.w), a, b); h.w = a; l.w = b;
    hi = __builtin_bit_cast(bf16x8_t, h); lo = __builtin_bit_cast(bf16x8_t, l);
}

// FEED = the input projection runs BESIDE the forward sweep that consumes it (pgasr_gemm_x3w_feed_f32): persistent
// workgroups on the XCDs the sweep leaves free draw tiles from a counter in the order the sweep needs them -- row
// tile i of the forward direction's column half together with row tile (last - i) of the reverse direction's -- store
// C write-through (sc1: the reader sits on another XCD, behind another L2) and count finished tiles per (direction,
// row tile); the sweep's helper workgroups wait for that count before they stage a step's rows (lstm.hip).
template <bool FEED>
__global__ __launch_bounds__(DMA_THREADS) void gemm_x3w_kernel(DmaGemmArgs g) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];   // the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int nk = g.K / TK;
    if (FEED && g.xcc_busy) {       // a workgroup on one of the sweep's XCDs leaves at once (placement is read, not assumed)
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;
        if (__hip_atomic_load(g.xcc_busy + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    }
  for (;;) {
    int tbx, tby;
    int kt0 = 0, kt1 = nk, qpart = -1;      // k-tile range of this work item; qpart >= 0: one quarter of a split tile
    unsigned tile = 0;
    if (FEED) {
        unsigned* mailbox = reinterpret_cast<unsigned*>(smem + NST * STAGE_BYTES);   // 16 bytes past the stages
        if (tid == 0) *mailbox = __hip_atomic_fetch_add(g.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned t = *mailbox;
        __syncthreads();
        const unsigned S = (unsigned)g.split_tiles, ntot = (unsigned)g.mt_count * (unsigned)g.nt_count;
        if (t >= ntot + 3u * S) return;             // 4 S quarter items, then the remaining ntot - S whole tiles
        if (t < 4u * S) { tile = t >> 2; qpart = (int)(t & 3u); kt0 = qpart * (nk >> 2); kt1 = kt0 + (nk >> 2); }
        else tile = t - 3u * S;
        const int half = g.nt_count >> 1, grp = (int)(tile / (unsigned)g.nt_count), j = (int)(tile % (unsigned)g.nt_count);
        tbx = j;
        tby = ((j < half) != (g.order != 0)) ? grp : g.mt_count - 1 - grp;
    } else {
        swizzled_tile(tbx, tby);
    }
    const int m0 = tby * TM, n0 = tbx * TN;

    // ---- per-lane DMA sources (k offset added per tile) ----
    const float* pa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = j * DMA_THREADS + tid, row = i >> 3, cp = i & 7, c = cp ^ ((row >> 1) & 7);
        int gm = m0 + row; gm = gm < g.M ? gm : g.M - 1;
        pa[j] = g.A + (size_t)gm * g.lda + c * 4;
    }
    const unsigned short *ph, *pl;
    {
        const int row = tid >> 2, cp = tid & 3, c = cp ^ ((row >> 2) & 3);
        const size_t o = (size_t)(n0 + row) * g.K + c * 8;
        ph = g.Whi + o; pl = g.Wlo + o;
    }
    auto issue = [&](int kt, int stage) {      // ALWAYS 6 wave-instructions (k clamped), so the counted wait is exact
        const int k0 = (kt < nk ? kt : nk - 1) * TK;
        unsigned char* sa = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16(pa[j] + k0, sa + (j * DMA_THREADS + w * 64) * 16);
        dma16(ph + k0, sa + A_BYTES + w * 64 * 16);
        dma16(pl + k0, sa + A_BYTES + P_BYTES + w * 64 * 16);
    };

    // ---- per-lane fragment read offsets inside a stage ----
    const int fr = lane & 31, fh = lane >> 5;
    unsigned offA[2][2], offB[2][2];     // [k-step][row tile / column tile]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + fr, sw = (row >> 1) & 7;
            const int n = wn * 64 + i * 32 + fr, swb = (n >> 2) & 3;
            offA[ks][i] = (unsigned)(row * 128 + (((ks * 4 + fh * 2) ^ sw) * 16));   // second chunk: ^ 16
            offB[ks][i] = (unsigned)(A_BYTES + n * 64 + (((ks * 2 + fh) ^ swb) * 16));
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // One barrier per k-tile: "my pieces of tile kt have landed" (counted vmcnt: tile kt+1 may still fly) -> barrier
    // (everyone's pieces visible; nobody reads stage (kt+2)%3 = (kt-1)%3 any more) -> issue tile kt+2 -> read, split,
    // multiply tile kt.  Measured on the path's shapes (NT M=32000,N=2048,K=512 / NN M=32000,N=512,K=2048, same
    // process; the register-staged bf16x3 kernel of gemm.hip: 327 / 317 us): this form 288 / 229 us.  Variants that
    // were built, passed the same tests and were SLOWER: all 16 fragment reads of a k-tile issued up front
    // (315 / 266); two wave groups half a k-tile apart, two barriers per tile, MFMA phase against load phase
    // (339 / 285); 4 waves of 128x64 with the split hand-placed in the MFMA shadows (382 / 270); 4 dedicated loader
    // waves beside the 8 compute waves (308 / 265).  rocprof (profiles/r01_gemm_pmc.txt): no LDS bank conflicts, MFMA
    // pipe 35 % busy, VALU 28 %, per k-tile the CU moves 48 KB by DMA and 128 KB of fragment reads through LDS -- MFMA,
    // LDS, VALU and the load path are all within 2x of each other, so no single reordering wins.  Round 2 measured the
    // "bf16-plane activation format" idea in isolation: with A arriving as hi / lo planes too (same tile, same stages,
    // fragments straight from LDS into the MFMA, no v_cvt on the path; bit-identical results) the two shapes took
    // 276 / 223 us against 286 / 243 us in the same process -- 4-8 %, not enough to pay for producers (sweep storer
    // waves, dropout) writing two planes instead of one fp32 row.  The split was not what the MFMA pipe waits for; the
    // per-k-tile barrier and the 128 KB of fragment reads per k-tile are.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    f32x16 total[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) total[i][j][r] = 0.f;
    const int kq = g.quarters == 4 ? (nk >> 2) : nk;       // k-tiles per quarter
    issue(kt0, 0);
    issue(kt0 + 1, 1);
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        if (kt != kt0 && kt % kq == 0) {                    // quarter boundary inside one workgroup's walk over K
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { total[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
        }
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(kt + 2, stage >= 1 ? stage - 1 : NST - 1);       // (stage + 2) % 3
        const unsigned sbase = lds0 + (unsigned)stage * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            RawFrag r;
            frag_read(r, sbase + offA[ks][0], sbase + offA[ks][1], sbase + offB[ks][0], sbase + offB[ks][1]);
            frag_wait(r);
            bf16x8_t ah[2], al[2];
            split8(r.a[0][0], r.a[0][1], ah[0], al[0]);
            split8(r.a[1][0], r.a[1][1], ah[1], al[1]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, r.bh[j]), bl = __builtin_bit_cast(bf16x8_t, r.bl[j]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
                }
        }
        stage = stage + 1 < NST ? stage + 1 : 0;
    }

    if (FEED && qpart >= 0) {
        // one quarter of a split tile: park the accumulators (thread-major: a wave instruction stores 256 contiguous
        // bytes) write-through, count the arrival; the LAST of the four sums the quarters in index order and goes on to
        // the epilogue, the others take their next work item
        __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(g.slabs, 0, (int)((size_t)g.split_tiles * 4 * 64 * 512 * 4), 0x00020000);
        const unsigned sb = (tile * 4u + (unsigned)qpart) * 64u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), srs, ((sb + (unsigned)((i * 2 + j) * 16 + r)) * 512u + (unsigned)tid) * 4u, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* mailbox = reinterpret_cast<unsigned*>(smem + NST * STAGE_BYTES);
        if (tid == 0) *mailbox = __hip_atomic_fetch_add(g.arrive + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned before = *mailbox;
        __syncthreads();
        if (before != 3u) { if (g.single) return; continue; }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        total[i][j][r] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            srs, (((tile * 4u + (unsigned)qq) * 64u + (unsigned)((i * 2 + j) * 16 + r)) * 512u + (unsigned)tid) * 4u, 0, 16));
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) total[i][j][r] += acc[i][j][r];
    }

    // epilogue: 32x32 accumulator layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Branch-free: rows past M
    // fall outside the buffer resource's range and are dropped by the hardware, and the optional leaky' factors of a
    // 32 x 32 tile are loaded as ONE batch in front of its 16 stores.  (Round 1-2 had `if (m >= M) continue` plus a
    // conditional dact load per element: hipcc then puts s_waitcnt vmcnt(0) in front of EVERY store -- each waits for the
    // previous one's acknowledgement.  Measured round 3 on the 256 x 256 tile with DMA and MFMA switched off: the
    // epilogue of the input projection took 160 of the kernel's 358 us.)
    const int cl = lane & 31, rq = lane >> 5;
    __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(unsigned)((size_t)g.M * g.ldc * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dact_y ? g.dact_y : g.C), 0, (int)(unsigned)((size_t)g.M * g.ldc * 4), 0x00020000);
    float bsum[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bsum[j] = g.bias ? g.bias[n0 + wn * 64 + j * 32 + cl] : 0.f;
    if (g.dact_y) {            // two straight-line loops: a merge point inside one would bring the per-tile s_waitcnt vmcnt(0) back
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned o0 = (unsigned)(((size_t)(m0 + wm * 64 + i * 32 + 4 * rq) * g.ldc + n0 + wn * 64 + j * 32 + cl) * 4);
                float f[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    f[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, o0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldc * 4), 0, 0)) > 0.f ? 1.f : g.slope;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((total[i][j][r] + bsum[j]) * f[r]), crs, o0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldc * 4), 0, FEED ? 16 : 0);
            }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned o0 = (unsigned)(((size_t)(m0 + wm * 64 + i * 32 + 4 * rq) * g.ldc + n0 + wn * 64 + j * 32 + cl) * 4);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(total[i][j][r] + bsum[j]), crs, o0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldc * 4), 0, FEED ? 16 : 0);
            }
    }
    if (!FEED) return;
    // the tile's stores have reached memory (vmcnt(0) in every wave, then the barrier) before it is counted; the
    // barrier also retires every DMA of this tile before the next one reuses the stages
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        __hip_atomic_fetch_add(g.tiles_done + (tbx < (g.nt_count >> 1) ? 0 : g.mt_count) + tby, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g.single) return;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same product on a 256 x 256 tile with 128 x 128 per wave (round 3).  What the 256 x 128 kernel above saturates first
// is not the MFMA pipe (35 % busy) but instruction ISSUE on its two waves per SIMD: per 12 MFMAs a wave reads 8 KB of
// fragments and splits two 32 x 8 fp32 A fragments into bf16 hi / lo (~56 VALU instructions), and an MFMA itself holds
// the SIMD's vector issue for 8 of its 32 cycles.  A 128 x 128 wave tile (4 waves, one per SIMD, the 256 accumulator
// registers in the unified VGPR/AGPR file) does 48 MFMAs per 16-deep k-step on 16 KB of fragments and four A splits:
// half the LDS bytes and half the VALU per MFMA, and no second wave competing for the SIMD's issue slots.
//   * stage = ONE 16-deep k-step: A 256 x 16 fp32 (16 KB) + two W planes 256 x 16 bf16 (8 KB each) = 32 KB, four stages
//     (128 KB), every wave issues exactly 8 LDS-DMA instructions per k-step, counted s_waitcnt vmcnt(16): up to three
//     k-steps (96 KB) in flight per CU;
//   * A rows are 64 B (4 chunks, chunk ^= (row >> 2) & 3), plane rows 32 B (2 chunks, chunk ^= (row >> 3) & 1): any lane
//     group of a ds_read_b128 covers the 16 slots of a 256-byte bank row once;
//   * ONE barrier per k-step, placed BEFORE the last quarter of the step's MFMAs: the wave waits for its DMA pieces of
//     the next k-step, meets the others, refills the stage everybody has finished reading, issues the next k-step's
//     fragment reads (W fragments double-buffered in registers) -- and only then issues row tile 3's twelve MFMAs, which
//     cover the LDS latency and the first split of the next step;
//   * accumulation over K is ONE chain per tile (the 256 x 128 kernel's fixed-order K-quarters would need a second set
//     of 256 accumulators); only the feed's first tiles are computed as four parallel K-quarters, parked in the slab
//     area and summed in index order, and the sequential order of the same product runs the SAME kernel with the same
//     decomposition, so fed and sequential results stay bit-identical (functional.py).
// Preconditions: K % 32 == 0 (k-steps are taken in pairs), N % 256 == 0, lda % 4 == 0, 16-byte aligned operands.
// ------------------------------------------------------------------------------------------------------------------
namespace w256 {
constexpr int TM = 256, TN = 256, TK = 16, NST = 4, THREADS = 256;
constexpr int A_BYTES = TM * TK * 4;                 // 16 KB raw fp32
constexpr int P_BYTES = TN * TK * 2;                 // 8 KB per bf16 plane
constexpr int STAGE_BYTES = A_BYTES + 2 * P_BYTES;   // 32 KB
constexpr int SLAB_FLOATS = 256 * THREADS;           // one parked accumulator set (256 registers x 256 threads) = 256 KB

struct RawA { u32x4_t a[4][2]; };          // row tile x chunk (8 fp32 of one row)
struct RawB { u32x4_t h[4], l[4]; };       // column tile: hi / lo plane (8 bf16 of one row)

__device__ __forceinline__ void read_a(RawA& r, unsigned p0, unsigned p1, unsigned p2, unsigned p3) {
    asm volatile("ds_read_b128 %0, %8\n\t"
                 "ds_read_b128 %1, %9\n\t"
                 "ds_read_b128 %2, %10\n\t"
                 "ds_read_b128 %3, %11\n\t"
                 "ds_read_b128 %4, %12\n\t"
                 "ds_read_b128 %5, %13\n\t"
                 "ds_read_b128 %6, %14\n\t"
                 "ds_read_b128 %7, %15"
                 : "=&v"(r.a[0][0]), "=&v"(r.a[0][1]), "=&v"(r.a[1][0]), "=&v"(r.a[1][1]),
                   "=&v"(r.a[2][0]), "=&v"(r.a[2][1]), "=&v"(r.a[3][0]), "=&v"(r.a[3][1])
                 : "v"(p0), "v"(p0 ^ 16u), "v"(p1), "v"(p1 ^ 16u), "v"(p2), "v"(p2 ^ 16u), "v"(p3), "v"(p3 ^ 16u)
                 : "memory");
}
__device__ __forceinline__ void read_b(RawB& r, unsigned p0, unsigned p1, unsigned p2, unsigned p3) {
    static_assert(P_BYTES == 8192, "plane offset is spelled in the asm below");
    asm volatile("ds_read_b128 %0, %8\n\t"
                 "ds_read_b128 %1, %8 offset:8192\n\t"
                 "ds_read_b128 %2, %9\n\t"
                 "ds_read_b128 %3, %9 offset:8192\n\t"
                 "ds_read_b128 %4, %10\n\t"
                 "ds_read_b128 %5, %10 offset:8192\n\t"
                 "ds_read_b128 %6, %11\n\t"
                 "ds_read_b128 %7, %11 offset:8192"
                 : "=&v"(r.h[0]), "=&v"(r.l[0]), "=&v"(r.h[1]), "=&v"(r.l[1]),
                   "=&v"(r.h[2]), "=&v"(r.l[2]), "=&v"(r.h[3]), "=&v"(r.l[3])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
}
__device__ __forceinline__ void wait_frags(RawA& a, RawB& b) {     // claims the registers the reads above are filling
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a.a[0][0]), "+v"(a.a[0][1]), "+v"(a.a[1][0]), "+v"(a.a[1][1]),
                   "+v"(a.a[2][0]), "+v"(a.a[2][1]), "+v"(a.a[3][0]), "+v"(a.a[3][1]),
                   "+v"(b.h[0]), "+v"(b.l[0]), "+v"(b.h[1]), "+v"(b.l[1]),
                   "+v"(b.h[2]), "+v"(b.l[2]), "+v"(b.h[3]), "+v"(b.l[3])
                 :: "memory");
}

template <bool FEED>
__global__ __launch_bounds__(THREADS) void gemm_x3w256_kernel(DmaGemmArgs g) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];   // the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int nk = g.K / TK;
    if (FEED && g.xcc_busy) {       // a workgroup on one of the sweep's XCDs leaves at once (placement is read, not assumed)
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;
        if (__hip_atomic_load(g.xcc_busy + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    }
  for (;;) {
    int tbx, tby;
    int kt0 = 0, kt1 = nk, qpart = -1;      // k-step range of this work item; qpart >= 0: one quarter of a split tile
    unsigned tile = 0;
    if (FEED) {
        unsigned* mailbox = reinterpret_cast<unsigned*>(smem + NST * STAGE_BYTES);   // 16 bytes past the stages
        if (tid == 0) *mailbox = __hip_atomic_fetch_add(g.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned t = *mailbox;
        __syncthreads();
        const unsigned S = (unsigned)g.split_tiles, ntot = (unsigned)g.mt_count * (unsigned)g.nt_count;
        if (t >= ntot + 3u * S) return;             // 4 S quarter items, then the remaining ntot - S whole tiles
        if (t < 4u * S) { tile = t >> 2; qpart = (int)(t & 3u); kt0 = qpart * (nk >> 2); kt1 = kt0 + (nk >> 2); }
        else tile = t - 3u * S;
        const int half = g.nt_count >> 1, grp = (int)(tile / (unsigned)g.nt_count), j = (int)(tile % (unsigned)g.nt_count);
        tbx = j;
        tby = ((j < half) != (g.order != 0)) ? grp : g.mt_count - 1 - grp;
    } else {
        swizzled_tile(tbx, tby);
    }
    const int m0 = tby * TM, n0 = tbx * TN;

    // ---- per-lane DMA sources (k offset added per k-step): wave w moves A pieces 4j + w (16 rows each) and plane pieces 4j + w (32 rows each)
    const float* pa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 16 * (4 * j + w) + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
        int gm = m0 + row; gm = gm < g.M ? gm : g.M - 1;
        pa[j] = g.A + (size_t)gm * g.lda + c * 4;
    }
    const unsigned short *ph[2], *pl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 32 * (4 * j + w) + (lane >> 1), cp = lane & 1, c = cp ^ ((row >> 3) & 1);
        const size_t o = (size_t)(n0 + row) * g.K + c * 8;
        ph[j] = g.Whi + o; pl[j] = g.Wlo + o;
    }
    auto issue = [&](int kt, int stage) {      // ALWAYS 8 wave-instructions (k clamped), so the counted waits are exact
        const int k0 = (kt < nk ? kt : nk - 1) * TK;
        unsigned char* sa = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16(pa[j] + k0, sa + (4 * j + w) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16(ph[j] + k0, sa + A_BYTES + (4 * j + w) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16(pl[j] + k0, sa + A_BYTES + P_BYTES + (4 * j + w) * 1024);
    };

    // ---- per-lane fragment read offsets inside a stage ----
    const int fr = lane & 31, fh = lane >> 5;
    unsigned offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wm * 128 + i * 32 + fr, n = wn * 128 + i * 32 + fr;
        offA[i] = (unsigned)(row * 64 + (((fh * 2) ^ ((row >> 2) & 3)) * 16));        // second chunk: ^ 16
        offB[i] = (unsigned)(A_BYTES + n * 32 + ((fh ^ ((n >> 3) & 1)) * 16));        // lo plane: + P_BYTES
    }

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int diag = FEED ? 0 : g.quarters;      // plain launches carry a diagnostic code there (x3w_diag)
    RawA ra;
    RawB rb0, rb1;
#pragma unroll
    for (int s = 0; s < NST; ++s) issue(kt0 + s, s);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      // my pieces of the first k-step have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_a(ra, lds0 + offA[0], lds0 + offA[1], lds0 + offA[2], lds0 + offA[3]);
    read_b(rb0, lds0 + offB[0], lds0 + offB[1], lds0 + offB[2], lds0 + offB[3]);

    // one k-step: fragments of step kt are (being) read into ra / BCUR; BNXT receives the next step's W fragments.
    // sched_barrier(0) pins the three regions (hipcc otherwise hoists the barrier to the top of the step and sinks the
    // next step's fragment reads behind the last MFMA, exposing their latency); inside a region the scheduler is free
    // to interleave the A splits with the MFMAs.
    auto kstep = [&](const int kt, RawB& bcur, RawB& bnxt) {
        wait_frags(ra, bcur);
        bf16x8_t ah[4], al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split8(ra.a[i][0], ra.a[i][1], ah[i], al[i]);
        bf16x8_t bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { bh[j] = __builtin_bit_cast(bf16x8_t, bcur.h[j]); bl[j] = __builtin_bit_cast(bf16x8_t, bcur.l[j]); }
        if (FEED || !(diag & 2))
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        // the next k-step: its DMA pieces (mine) have landed, everybody's are visible behind the barrier, and the stage of
        // step kt -- read into registers by every wave before it got here -- is free for step kt + 4
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int rel = kt - kt0;
        const unsigned sb = lds0 + (unsigned)((rel + 1) & 3) * STAGE_BYTES;
        read_a(ra, sb + offA[0], sb + offA[1], sb + offA[2], sb + offA[3]);
        read_b(bnxt, sb + offB[0], sb + offB[1], sb + offB[2], sb + offB[3]);
        if (FEED || !(diag & 1)) issue(kt + NST, rel & 3);
        __builtin_amdgcn_sched_barrier(0);
        if (FEED || !(diag & 2))
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[3][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[3], bh[j], acc[3][j], 0, 0, 0);
            acc[3][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[3], bl[j], acc[3][j], 0, 0, 0);
            acc[3][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[3], bh[j], acc[3][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = kt0; kt < kt1; kt += 2) {       // k-steps come in pairs (K % 32 == 0; a quarter of K % 128 == 0 is even too)
        kstep(kt, rb0, rb1);
        kstep(kt + 1, rb1, rb0);
    }
    wait_frags(ra, rb0);                           // the reads issued for the step behind the last one (discarded)

    if (FEED && qpart >= 0) {
        // one quarter of a split tile: park the accumulators (thread-major: a wave instruction stores 256 contiguous
        // bytes) write-through, count the arrival; the LAST of the four sums the quarters in index order and goes on to
        // the epilogue, the others take their next work item
        __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(g.slabs, 0, (int)((size_t)g.split_tiles * 4 * SLAB_FLOATS * 4), 0x00020000);
        const unsigned sbq = (tile * 4u + (unsigned)qpart) * 256u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), srs, ((sbq + (unsigned)((i * 4 + j) * 16 + r)) * 256u + (unsigned)tid) * 4u, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* mailbox = reinterpret_cast<unsigned*>(smem + NST * STAGE_BYTES);
        if (tid == 0) *mailbox = __hip_atomic_fetch_add(g.arrive + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned before = *mailbox;
        __syncthreads();
        if (before != 3u) continue;
        // total = ((q0 + q1) + q2) + q3, whoever arrives last
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float t = 0.f;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            srs, (((tile * 4u + (unsigned)qq) * 256u + (unsigned)((i * 4 + j) * 16 + r)) * 256u + (unsigned)tid) * 4u, 0, 16));
                        t = qq == 0 ? v : t + v;
                    }
                    acc[i][j][r] = t;
                }
    }

    // epilogue (branch-free, see the 256 x 128 kernel): rows past M are dropped by the buffer resource's range check
    const int cl = lane & 31, rq = lane >> 5;
    __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(unsigned)((size_t)g.M * g.ldc * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dact_y ? g.dact_y : g.C), 0, (int)(unsigned)((size_t)g.M * g.ldc * 4), 0x00020000);
    float bsum[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bsum[j] = g.bias ? g.bias[n0 + wn * 128 + j * 32 + cl] : 0.f;
    if (g.dact_y) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned o0 = (unsigned)(((size_t)(m0 + wm * 128 + i * 32 + 4 * rq) * g.ldc + n0 + wn * 128 + j * 32 + cl) * 4);
                float f[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    f[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, o0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldc * 4), 0, 0)) > 0.f ? 1.f : g.slope;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((acc[i][j][r] + bsum[j]) * f[r]), crs, o0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldc * 4), 0, FEED ? 16 : 0);
            }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned o0 = (unsigned)(((size_t)(m0 + wm * 128 + i * 32 + 4 * rq) * g.ldc + n0 + wn * 128 + j * 32 + cl) * 4);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r] + bsum[j]), crs, o0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldc * 4), 0, FEED ? 16 : 0);
            }
    }
    if (!FEED) return;
    // the tile's stores have reached memory (vmcnt(0) in every wave, then the barrier) before it is counted; the
    // barrier also retires every DMA of this tile before the next one reuses the stages
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        __hip_atomic_fetch_add(g.tiles_done + (tbx < (g.nt_count >> 1) ? 0 : g.mt_count) + tby, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
}  // namespace w256

// fp32 (rows x cols, leading dim ld) -> dense bf16 hi / lo planes; transpose: planes are (cols x rows)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int rows, int cols, int ld,
                                                           int transpose, unsigned short* __restrict__ hi,
                                                           unsigned short* __restrict__ lo) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)rows * cols) return;
    int r, c;
    if (transpose) { c = (int)(idx / rows); r = (int)(idx % rows); }     // output index = c*rows + r
    else { r = (int)(idx / cols); c = (int)(idx % cols); }
    unsigned h, l;
    split2(src[(size_t)r * ld + c], 0.f, h, l);
    hi[idx] = (unsigned short)h;
    lo[idx] = (unsigned short)l;
}

}  // namespace

extern "C" int pgasr_split_bf16_planes(const float* src, int rows, int cols, int ld, int transpose,
                                       unsigned short* hi, unsigned short* lo, void* stream) {
    if (!src || !hi || !lo || rows <= 0 || cols <= 0 || ld < cols) return PGASR_ERR_INVALID_ARG;
    const size_t total = (size_t)rows * cols;
    PGASR_LAUNCH_KERNEL(split_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                        src, rows, cols, ld, transpose, hi, lo);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

// The 256 x 256 kernel is opt-in (PGASR_X3W_TILE=256) until it beats the 256 x 128 one on the path's shapes: first
// measurement (round 3, one box): input projection 336 against 283 us, input gradient 218 against 223 us.
// PGASR_X3W_TILE selects the x3w kernel (read at every call: tests and A/B scripts switch it inside one process):
// "128": 256 x 128 (8 waves x 64 x 64), "256": 256 x 256 / 4 waves (w256 below), "c": 256 x 256 / 8 waves with the
// cooperative A split (gemm_c256.hip).  Unset: plain launches take "c" (round 3, one box, the path's two shapes: 227 / 225 us
// against 260 / 269 for "128" and 237 / 229 for "256"); FEED launches keep "128": what a fed sweep waits for is its FIRST
// row tiles, and the smaller tile delivers them sooner (train step 8.55 ms with "128" feeds against 8.69 with "c" feeds,
// backward sweeps 1.25 / 1.27 against 1.32 / 1.33 ms).
static int x3w_tile_mode(bool feed) {
    const char* e = getenv("PGASR_X3W_TILE");
    if (!e || !e[0]) return feed ? 0 : 2;
    if (e[0] == 'c') return 2;
    if (e[0] == '2' && e[1] == '5' && e[2] == '6') return 1;
    return 0;
}
static int x3w_diag() {      // diagnostic variants of the 256 x 256 / 4-wave kernel (results invalid): bit 0 = no DMA in the k-loop, bit 1 = no MFMA
    const char* e = getenv("PGASR_X3W_DIAG");
    return e ? atoi(e) : 0;
}
static int x3w_quarters(int K) { return (K >= 1024 && K % (4 * TK) == 0) ? 4 : 1; }   // a quarter of >= 8 k-tiles (K = 512 in quarters: step +0.05 ms)
constexpr int FEED_SPLIT_MAX = 64;       // split tiles per feed: 4 x 64 slabs of 128 KB = 32 MB of workspace

// Column tiles per direction half that a feed of an N-column product counts in tiles_done (the consumer's `fed_need`):
// N / 2 / 256 on the 256 x 256 tile, N / 2 / 128 on the 256 x 128 one; 0: not a feedable width.
extern "C" int pgasr_gemm_x3w_feed_col_tiles(int N) {
    if (N <= 0 || N % (2 * TN)) return 0;
    if (N % (2 * w256::TN) == 0 && x3w_tile_mode(true) != 0) return N / (2 * w256::TN);
    return N / (2 * TN);
}

// Work items in the first `groups` tile groups of a feed (a group = the column tiles of row tile i of one direction half and
// of row tile last - i of the other): what a HEAD launch (phase 1) computes.  0: no head for this shape / tile structure
// (the 256-wide feeds have none), or more items than one wave of workgroups.
extern "C" int pgasr_gemm_x3w_feed_head_items(int N, int K, int groups) {
    if (groups <= 0 || N <= 0 || K <= 0 || (N % (2 * TN)) || (K % TK) || x3w_tile_mode(true) != 0) return 0;
    const int nt = N / TN;
    long long tiles = (long long)groups * nt;
    int items;
    if (x3w_quarters(K) == 4) {
        long long split = 16LL * nt;
        if (split > FEED_SPLIT_MAX) split = FEED_SPLIT_MAX;
        if (tiles > split) return 0;              // a head stays inside the quarter-split tiles (the host sizes the workspace for them)
        items = (int)(4 * tiles);
    } else {
        items = (int)tiles;
    }
    return items <= 256 ? items : 0;
}

extern "C" size_t pgasr_gemm_x3w_feed_workspace_bytes(void) { return 1024 + (size_t)FEED_SPLIT_MAX * 4 * 64 * 512 * 4; }

extern "C" int pgasr_gemm_x3w_f32(int M, int N, int K, const float* A, int lda, const unsigned short* Whi,
                                  const unsigned short* Wlo, float* C, int ldc, const float* bias,
                                  const float* dact_y, float slope, void* stream) {
    if (!A || !Whi || !Wlo || !C || M <= 0 || N <= 0 || K <= 0 || lda < K || ldc < N) return PGASR_ERR_INVALID_ARG;
    if ((K % TK) || (N % TN) || (lda & 3) || (((size_t)A) & 15) || (((size_t)Whi) & 15) || (((size_t)Wlo) & 15))
        return PGASR_ERR_UNSUPPORTED;
    const unsigned gy = (unsigned)((M + TM - 1) / TM);
    if (gy > 65535u) return PGASR_ERR_UNSUPPORTED;
    if ((size_t)M * ldc * 4 >= ((size_t)1 << 32)) return PGASR_ERR_UNSUPPORTED;      // buffer-addressed epilogue (callers fall back to pgasr_gemm_f32)
    if (N % 256 == 0 && x3w_tile_mode(false) == 2 && (size_t)M * lda * 4 < ((size_t)1 << 32)) {
        PgasrX3cArgs a{A, Whi, Wlo, C, M, N, K, lda, ldc, bias, dact_y, slope, 0, nullptr, nullptr, nullptr, 0, 0, 0, 1, 0, nullptr, nullptr};
        return pgasr_internal_x3c_launch(a, (hipStream_t)stream);
    }
    if (N % w256::TN == 0 && x3w_tile_mode(false) == 1) {     // the 256 x 256 / 4-wave tile (one accumulation chain over K per tile)
        const size_t lds2 = (size_t)w256::NST * w256::STAGE_BYTES;
        if (hipFuncSetAttribute((const void*)w256::gemm_x3w256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
            return PGASR_ERR_LAUNCH;
        DmaGemmArgs g2{A, Whi, Wlo, C, M, N, K, lda, ldc, bias, dact_y, slope, nullptr, nullptr, nullptr, 0, 0, 0, x3w_diag(), 0, nullptr, nullptr};
        PGASR_LAUNCH_KERNEL(w256::gemm_x3w256_kernel<false>, dim3((unsigned)(N / w256::TN), gy), dim3(w256::THREADS), lds2, (hipStream_t)stream, g2);
        PGASR_CHECK_LAUNCH();
        return PGASR_OK;
    }
    const size_t lds = (size_t)NST * STAGE_BYTES;   // 144 KB of the CU's 160 KB: opt in per call (idempotent, no state kept)
    if (hipFuncSetAttribute((const void*)gemm_x3w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    DmaGemmArgs g{A, Whi, Wlo, C, M, N, K, lda, ldc, bias, dact_y, slope, nullptr, nullptr, nullptr, 0, 0, 0, x3w_quarters(K), 0, nullptr, nullptr};
    PGASR_LAUNCH_KERNEL(gemm_x3w_kernel<false>, dim3((unsigned)(N / TN), gy), dim3(DMA_THREADS), lds, (hipStream_t)stream, g);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

// The same product in feed-ahead mode (see the FEED kernel): C rows are produced in the order a forward LSTM sweep
// that is already running consumes them (N = the two directions' column halves, rows = (t, b) time-major).
//   tiles_done : [2][ceil(M/256)] words, zeroed by the caller BEFORE the sweep is launched; word [d][i] reaches N/256
//                when rows [256 i, 256 i + 256) of direction d's columns are in memory
//   xcc_busy   : the sweep's per-XCD busy counters (pgasr_lstm_busy_offset); workgroups that find themselves on a
//                busy XCD take no tile; a second, unmasked launch picks up whatever is left (normally nothing)
//   order      : 0 = rows in the order a FORWARD sweep consumes them (direction 0 ascending in t, direction 1
//                descending), 1 = the order of a BACKWARD sweep (mirrored)
//   workspace  : >= 1024 bytes (tile and arrival counters); with pgasr_gemm_x3w_feed_workspace_bytes() the first tiles
//                are computed as four parallel K-quarters (same bits, a quarter of the latency)
extern "C" int pgasr_gemm_x3w_feed_f32(int M, int N, int K, const float* A, int lda, const unsigned short* Whi,
                                       const unsigned short* Wlo, float* C, int ldc, const float* bias,
                                       const unsigned* xcc_busy, unsigned* tiles_done, int order, int phase, int head_groups,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !Whi || !Wlo || !C || !tiles_done || M <= 0 || N <= 0 || K <= 0 || lda < K || ldc < N) return PGASR_ERR_INVALID_ARG;
    if (order < 0 || order > 1 || phase < 0 || phase > 2 || head_groups < 0) return PGASR_ERR_INVALID_ARG;
    if (phase != 0 && pgasr_gemm_x3w_feed_head_items(N, K, head_groups) <= 0) return PGASR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < 1024) return PGASR_ERR_WORKSPACE;
    if ((K % TK) || (N % (2 * TN)) || (lda & 3) || (((size_t)A) & 15) || (((size_t)Whi) & 15) || (((size_t)Wlo) & 15))
        return PGASR_ERR_UNSUPPORTED;
    if ((size_t)M * ldc * 4 >= ((size_t)1 << 31)) return PGASR_ERR_UNSUPPORTED;      // buffer-addressed stores
    hipStream_t st = (hipStream_t)stream;
    if (pgasr_gemm_x3w_feed_col_tiles(N) == N / (2 * w256::TN) && x3w_tile_mode(true) == 2 && (size_t)M * lda * 4 < ((size_t)1 << 32)) {
        const int mt2 = (M + 255) / 256, nt2 = N / 256;
        if (hipMemsetAsync(workspace, 0, 1024, st) != hipSuccess) return PGASR_ERR_LAUNCH;
        const int quarters2 = x3w_quarters(K);
        int split2 = 0;
        if (quarters2 == 4) {
            const size_t room = (workspace_bytes - 1024) / ((size_t)4 * pgasr_internal_x3c_slab_bytes());
            split2 = 16 * nt2;
            if (split2 > FEED_SPLIT_MAX) split2 = FEED_SPLIT_MAX;
            if ((size_t)split2 > room) split2 = (int)room;
            if (split2 > mt2 * nt2) split2 = mt2 * nt2;
        }
        PgasrX3cArgs a{A, Whi, Wlo, C, M, N, K, lda, ldc, bias, nullptr, 0.f, 1, (unsigned*)workspace, xcc_busy, tiles_done, mt2, nt2, order,
                       quarters2, split2, (float*)((char*)workspace + 1024), (unsigned*)workspace + 64};
        return pgasr_internal_x3c_launch(a, st);
    }
    if (pgasr_gemm_x3w_feed_col_tiles(N) == N / (2 * w256::TN)) {
        // 256 x 256 tiles: the same queue, counters and quarter protocol, slabs of 256 KB
        const int mt2 = (M + w256::TM - 1) / w256::TM, nt2 = N / w256::TN;
        const size_t lds2 = (size_t)w256::NST * w256::STAGE_BYTES + 16;
        if (hipFuncSetAttribute((const void*)w256::gemm_x3w256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
            return PGASR_ERR_LAUNCH;
        if (hipMemsetAsync(workspace, 0, 1024, st) != hipSuccess) return PGASR_ERR_LAUNCH;
        const int quarters2 = x3w_quarters(K);
        int split2 = 0;
        if (quarters2 == 4) {
            const size_t room = (workspace_bytes - 1024) / ((size_t)4 * w256::SLAB_FLOATS * 4);
            split2 = 16 * nt2;
            if (split2 > FEED_SPLIT_MAX) split2 = FEED_SPLIT_MAX;
            if ((size_t)split2 > room) split2 = (int)room;
            if (split2 > mt2 * nt2) split2 = mt2 * nt2;
        }
        DmaGemmArgs g2{A, Whi, Wlo, C, M, N, K, lda, ldc, bias, nullptr, 0.f, (unsigned*)workspace, xcc_busy, tiles_done, mt2, nt2, order,
                       quarters2, split2, (float*)((char*)workspace + 1024), (unsigned*)workspace + 64};
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) g2.xcc_busy = nullptr;
            PGASR_LAUNCH_KERNEL(w256::gemm_x3w256_kernel<true>, dim3(256), dim3(w256::THREADS), lds2, st, g2);
            PGASR_CHECK_LAUNCH();
        }
        return PGASR_OK;
    }
    const int mt = (M + TM - 1) / TM, nt = N / TN;
    const size_t lds = (size_t)NST * STAGE_BYTES + 16;
    if (hipFuncSetAttribute((const void*)gemm_x3w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    if (phase != 2 && hipMemsetAsync(workspace, 0, 1024, st) != hipSuccess) return PGASR_ERR_LAUNCH;   // tile counter + arrival counters
    // the first tile groups (16 time-ordered groups, at most FEED_SPLIT_MAX tiles and what the workspace holds) are split
    // into K-quarters: the sweep is waiting for exactly these
    const int quarters = x3w_quarters(K);
    int split = 0;
    if (quarters == 4) {
        const size_t room = (workspace_bytes - 1024) / ((size_t)4 * 64 * 512 * 4);
        split = 16 * nt;
        if (split > FEED_SPLIT_MAX) split = FEED_SPLIT_MAX;
        if ((size_t)split > room) split = (int)room;
        if (split > mt * nt) split = mt * nt;
    }
    DmaGemmArgs g{A, Whi, Wlo, C, M, N, K, lda, ldc, bias, nullptr, 0.f, (unsigned*)workspace, xcc_busy, tiles_done, mt, nt, order,
                  quarters, split, (float*)((char*)workspace + 1024), (unsigned*)workspace + 64, 0};
    if (phase == 1) {
        // HEAD: the first `head_groups` tile groups, one work item per workgroup, no XCD mask -- launched on the sweep's own
        // stream IN FRONT of the sweep, whose first rows are then in memory when it starts; phase 2 continues the queue
        g.single = 1; g.xcc_busy = nullptr;
        PGASR_LAUNCH_KERNEL(gemm_x3w_kernel<true>, dim3((unsigned)pgasr_gemm_x3w_feed_head_items(N, K, head_groups)), dim3(DMA_THREADS), lds, st, g);
        PGASR_CHECK_LAUNCH();
        return PGASR_OK;
    }
    // one persistent workgroup per CU (144 KB of LDS each); pass 1 ignores the busy counters
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) g.xcc_busy = nullptr;
        PGASR_LAUNCH_KERNEL(gemm_x3w_kernel<true>, dim3(256), dim3(DMA_THREADS), lds, st, g);
        PGASR_CHECK_LAUNCH();
    }
    return PGASR_OK;
}
