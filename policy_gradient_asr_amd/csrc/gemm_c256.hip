// C[M,N] = A[M,K] * W[N,K]^T (+ bias[n]) (* leaky'(dact_y[m,n])) -- the LSTM input projections X W_ih^T and their
// input gradients dG W_ih (model.py:39-44) -- third structure (round 3), built from what the first two measured:
//
//   * gemm_dma.hip's 256 x 128 kernel (8 waves x 64 x 64): every wave re-splits the fp32 A fragments it shares with the
//     other column wave (VALU ~28 % busy), MFMA pipe 35 %;
//   * its 256 x 256 / 4-wave variant (128 x 128 per wave, one wave per SIMD): half the split work per MFMA, but ONE wave
//     per SIMD issues everything in order -- LDS-DMA (~60 cycles each), splits, fragment reads and MFMAs add instead of
//     overlapping: 1.6-1.9 us per 16-deep k-step against 0.65 us of MFMA time (PMC: 4.4 VALU instructions per MFMA,
//     35 % of wave cycles in issue stalls).  With DMA and MFMA switched off in turn: DMA path alone 0.9 us per step,
//     MFMA + VALU alone 1.3 us.
//
// Here: 256 x 256 tile, EIGHT waves (two per SIMD, 128 x 64 each, 128 accumulator registers), 32-deep steps.
//   * fp32 A is split into bf16 hi / lo ONCE per workgroup: every thread loads 4 x 16 B of the NEXT step's A rows into
//     registers (fully coalesced: 8 lanes per 128-byte row segment), converts them and writes the two planes into an LDS
//     buffer ([row][32 k] bf16, 64-byte rows, chunk ^= (row >> 2) & 3) -- 48 VALU instructions per thread and step instead
//     of 112 per wave and 16-deep step; A fragments are then plain ds_read_b128 like the W fragments;
//   * W planes (pre-split, L2-resident) come by LDS-DMA, one 32-deep stage ahead (2 stages);
//   * ONE raw s_barrier per 32-deep step (48 MFMAs per wave between barriers); LDS: 2 x 32 KB A planes + 2 x 32 KB W = 128 KB;
//   * the two waves of a SIMD run the step's two halves in OPPOSITE order (waves 0-3: convert, then multiply; waves 4-7:
//     multiply, then convert), so that one wave's VALU / LDS-write phase sits beside its partner's MFMA phase;
//   * branch-free epilogue (buffer stores, rows past M dropped by the resource's range check).
// One accumulation chain over K per tile; only the feed's first tiles are fixed-order sums of four K-quarters (slabs), and
// the sequential order of the same product runs the same kernel (functional.py), so both orders give the same bits.
// Preconditions: K % 32 == 0 (quarters: K % 128 == 0), N % 256 == 0, lda % 4 == 0, 16-byte aligned A / planes,
// M * ldc * 4 < 2^32 and M * lda * 4 < 2^32.
#include "common.h"
#include "x3w_common.h"
#include <type_traits>

namespace {
namespace c256 {
constexpr int TM = 256, TN = 256, TK = 32, THREADS = 512;
constexpr int AP_BYTES = TM * TK * 2;            // one A plane of one buffer: 16 KB
constexpr int ABUF_BYTES = 2 * AP_BYTES;         // hi | lo
constexpr int WP_BYTES = TN * TK * 2;            // one W plane of one stage: 16 KB
constexpr int WSTAGE_BYTES = 2 * WP_BYTES;       // hi | lo
constexpr int LDS_W = 2 * ABUF_BYTES;            // [A buffer 0][A buffer 1][W stages ...][mailbox (FEED)]
constexpr int LDS_BYTES = LDS_W + 2 * WSTAGE_BYTES;      // 128 KB: the FEED kernel (two W stages, loads one step ahead)
constexpr int LDS_BYTES_DEEP = LDS_W + 3 * WSTAGE_BYTES; // 160 KB: the plain kernel (three W stages, A loads two steps ahead)
constexpr int SLAB_FLOATS = 128 * THREADS;       // one parked accumulator set: 256 KB

struct Frag { u32x4_t ah[4], al[4], bh[2], bl[2]; };

__device__ __forceinline__ void read_frags(Frag& f, unsigned a0, unsigned a1, unsigned a2, unsigned a3, unsigned b0, unsigned b1) {
    static_assert(AP_BYTES == 16384 && WP_BYTES == 16384, "plane offsets are spelled in the asm below");
    asm volatile("ds_read_b128 %0, %12\n\t"
                 "ds_read_b128 %1, %12 offset:16384\n\t"
                 "ds_read_b128 %8, %16\n\t"
                 "ds_read_b128 %9, %16 offset:16384\n\t"
                 "ds_read_b128 %10, %17\n\t"
                 "ds_read_b128 %11, %17 offset:16384\n\t"
                 "ds_read_b128 %2, %13\n\t"
                 "ds_read_b128 %3, %13 offset:16384\n\t"
                 "ds_read_b128 %4, %14\n\t"
                 "ds_read_b128 %5, %14 offset:16384\n\t"
                 "ds_read_b128 %6, %15\n\t"
                 "ds_read_b128 %7, %15 offset:16384"
                 : "=&v"(f.ah[0]), "=&v"(f.al[0]), "=&v"(f.ah[1]), "=&v"(f.al[1]), "=&v"(f.ah[2]), "=&v"(f.al[2]),
                   "=&v"(f.ah[3]), "=&v"(f.al[3]), "=&v"(f.bh[0]), "=&v"(f.bl[0]), "=&v"(f.bh[1]), "=&v"(f.bl[1])
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1)
                 : "memory");
}
__device__ __forceinline__ void wait_frags(Frag& f) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.ah[0]), "+v"(f.al[0]), "+v"(f.ah[1]), "+v"(f.al[1]), "+v"(f.ah[2]), "+v"(f.al[2]),
                   "+v"(f.ah[3]), "+v"(f.al[3]), "+v"(f.bh[0]), "+v"(f.bl[0]), "+v"(f.bh[1]), "+v"(f.bl[1])
                 :: "memory");
}
// two 8-byte LDS stores (hi plane, lo plane = + 16 KB) hidden from hipcc, which would drain the LDS-DMA in flight
// (s_waitcnt vmcnt(0)) in front of any LDS access it can see
__device__ __forceinline__ void write_planes(unsigned addr, unsigned h0, unsigned h1, unsigned l0, unsigned l1) {
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    const u2 h = {h0, h1}, l = {l0, l1};
    asm volatile("ds_write_b64 %0, %1\n\t"
                 "ds_write_b64 %0, %2 offset:16384" :: "v"(addr), "v"(h), "v"(l) : "memory");
}

// DEEP (= !FEED, the plain launches): A loads run TWO steps ahead (two register sets) and W has three LDS stages.  What a
// CU pulls through its memory pipeline is in-flight bytes / latency, and one step of lead left 64 KB in flight per CU
// (measured: ~18 GB/s per CU with every CU pulling, the same as the 128-wide kernels); the FEED kernel keeps one step of
// lead (its 256 registers are full, and feeds default to the 256 x 128 kernel anyway).
// (free functions: clang rejects inline-asm operands that name captured variables inside a generic lambda)
__device__ __forceinline__ void load_a_regs(u32x4_t (&r)[4], const unsigned (&aoff)[4], unsigned kb, const float* A) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r[j]) : "v"(aoff[j] + kb), "s"(A) : "memory");
}
template <int HALF>
__device__ __forceinline__ void load_a_pair(u32x4_t (&r)[4], const unsigned (&aoff)[4], unsigned kb, const float* A) {
#pragma unroll
    for (int j = 2 * HALF; j < 2 * HALF + 2; ++j)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r[j]) : "v"(aoff[j] + kb), "s"(A) : "memory");
}
template <int CNT>
__device__ __forceinline__ void wait_a_regs(u32x4_t (&r)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(CNT) : "memory");
}

template <bool FEED>
__global__ __launch_bounds__(THREADS) void gemm_x3c_kernel(DmaGemmArgs g) {
    constexpr bool DEEP = !FEED;
    constexpr int NW = DEEP ? 3 : 2, NA = DEEP ? 2 : 1, VMW = DEEP ? 12 : 4, VMX = DEEP ? 8 : 0;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];   // the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int nk = g.K / TK;
    if (FEED && g.xcc_busy) {       // a workgroup on one of the sweep's XCDs leaves at once (placement is read, not assumed)
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;
        if (__hip_atomic_load(g.xcc_busy + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  for (;;) {
    int tbx, tby;
    int kt0 = 0, kt1 = nk, qpart = -1;      // step range of this work item; qpart >= 0: one quarter of a split tile
    unsigned tile = 0;
    if (FEED) {
        unsigned* mailbox = reinterpret_cast<unsigned*>(smem + LDS_BYTES);
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

    // ---- A: lane -> (row within an 8-row group, 16-byte chunk of the step's 128-byte row segment); wave w owns rows 32 w .. 32 w + 31
    const int rsub = lane >> 3, ac = lane & 7;
    unsigned aoff[4];           // byte offsets into A of (row, chunk) at k = 0 (rows past M clamped: their products are never stored)
    unsigned apw[4];            // byte offsets into an A-plane buffer (hi plane) of the 8 bytes this lane writes per row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = w * 32 + j * 8 + rsub;
        int gm = m0 + r; gm = gm < g.M ? gm : g.M - 1;
        aoff[j] = (unsigned)(((size_t)gm * g.lda + ac * 4) * 4);
        apw[j] = (unsigned)(r * 64 + (((ac >> 1) ^ ((r >> 2) & 3)) * 16) + (ac & 1) * 8);
    }
    // ---- W planes by LDS-DMA: piece p = 16 rows x 64 B; wave w moves pieces w and w + 8 of either plane
    const unsigned short *ph[2], *pl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 16 * (w + 8 * j) + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
        const size_t o = (size_t)(n0 + row) * g.K + c * 8;
        ph[j] = g.Whi + o; pl[j] = g.Wlo + o;
    }
    auto issue_w2 = [&](int kt, int stage, int plane) {     // two of a step's four W pieces: the hi (0) or the lo (1) plane
        const int k0 = (kt < nk ? kt : nk - 1) * TK;
        unsigned char* ws = smem + LDS_W + stage * WSTAGE_BYTES + plane * WP_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16((plane ? pl[j] : ph[j]) + k0, ws + (w + 8 * j) * 1024);
    };
    auto issue_w = [&](int kt, int stage) {      // ALWAYS 4 wave-instructions per step (k clamped), so the counted waits are exact
        issue_w2(kt, stage, 0); issue_w2(kt, stage, 1);
    };
    u32x4_t araw[NA][4];
    // ALWAYS 4 loads, in inline asm: a load hipcc can see is waited for with s_waitcnt vmcnt(0) where its registers are
    // handed to the asm wait below -- which would also drain the W pieces issued just before (seen in the .s)
    auto load_a = [&](int kt, auto setc) {        // the register set is a compile-time index
        const unsigned kb = (unsigned)((kt < nk ? kt : nk - 1) * TK * 4);
        load_a_regs(araw[decltype(setc)::value], aoff, kb, g.A);
    };
    auto load_a2 = [&](int kt, auto setc, auto halfc) {     // two of a step's four A loads
        const unsigned kb = (unsigned)((kt < nk ? kt : nk - 1) * TK * 4);
        load_a_pair<decltype(halfc)::value>(araw[decltype(setc)::value], aoff, kb, g.A);
    };
    auto wait_a = [&](auto setc, auto cntc) {     // s_waitcnt vmcnt(CNT) that claims the set's registers
        wait_a_regs<decltype(cntc)::value>(araw[decltype(setc)::value]);
    };
    auto convert_a = [&](int buf, auto setc) {    // register set -> hi / lo planes of buffer `buf`
        constexpr int S = decltype(setc)::value;
        const unsigned base = lds0 + (unsigned)buf * ABUF_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned h0, l0, h1, l1;
            split2(__uint_as_float(araw[S][j].x), __uint_as_float(araw[S][j].y), h0, l0);
            split2(__uint_as_float(araw[S][j].z), __uint_as_float(araw[S][j].w), h1, l1);
            write_planes(base + apw[j], h0, h1, l0, l1);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, NA - 1> I1;      // the second set (the first one again without DEEP)
    typedef std::integral_constant<int, VMW> IW;
    typedef std::integral_constant<int, 4> I4;

    // ---- fragment read offsets (k-step 0 of a stage; k-step 1 = ^ 32) ----
    const int fr = lane & 31, fh = lane >> 5;
    unsigned offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wm * 128 + i * 32 + fr;
        offA[i] = (unsigned)(row * 64 + ((fh ^ ((row >> 2) & 3)) * 16));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = wn * 64 + j * 32 + fr;
        offB[j] = (unsigned)(LDS_W + n * 64 + ((fh ^ ((n >> 2) & 3)) * 16));
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one 16-deep half of a step: 24 MFMAs on A buffer `cur`, W stage `wst`; `mem(0)` runs behind the first twelve, `mem(1)`
    // behind the last twelve.  The step's eight memory instructions are issued there, two at a time: a wave that issues
    // them in one burst at the step's start sits in the issue queue of a saturated memory pipeline (the no-MFMA variant
    // of this kernel needs 1.5 us per step for its 64 KB) with NO MFMA of its own in flight -- measured: MFMA time and
    // memory time added up exactly (2.9 us per step); behind twelve MFMAs (384 pipe cycles) a blocked issue costs nothing.
    auto multiply_half = [&](int cur, int wst, int ks, auto&& mem) {
        const unsigned ab = lds0 + (unsigned)cur * ABUF_BYTES, wb = lds0 + (unsigned)wst * WSTAGE_BYTES;
        Frag f;
        const unsigned x = ks ? 32u : 0u;
        read_frags(f, (ab + offA[0]) ^ x, (ab + offA[1]) ^ x, (ab + offA[2]) ^ x, (ab + offA[3]) ^ x, (wb + offB[0]) ^ x, (wb + offB[1]) ^ x);
        wait_frags(f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, f.ah[i]), al = __builtin_bit_cast(bf16x8_t, f.al[i]);
                const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, f.bh[j]), bl = __builtin_bit_cast(bf16x8_t, f.bl[j]);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
            }
            if (i & 1) {
                __builtin_amdgcn_sched_barrier(0);
                mem(i >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- prologue: A planes of the first step, W stage(s) ahead, A registers of the following step(s) ----
    load_a(kt0, I0{});
    issue_w(kt0, 0);
    wait_a(I0{}, I4{});                           // the A loads (older than the 4 W pieces) are in
    convert_a(0, I0{});
    if (DEEP) {
        load_a(kt0 + 1, I1{});
        issue_w(kt0 + 1, 1);
        load_a(kt0 + 2, I0{});
        asm volatile("s_waitcnt vmcnt(12)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");    // W stage 0 landed (my pieces), my plane stores done
    } else {
        load_a(kt0 + 1, I0{});
        asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // one 32-deep step; SET = the register set that holds the A rows of step kt + 1.  In flight when the step starts,
    // oldest first: [A(kt+1)] [W(kt+1)] [A(kt+2)] (DEEP) / [A(kt+1)] (else); every step issues its W pieces FIRST, then its A loads
    typedef std::integral_constant<int, VMX> IX;
    typedef std::integral_constant<int, 1> H1;
    auto step = [&](const int kt, auto setc) {
        const int rel = kt - kt0, cur = rel & 1, wst = rel % NW;
        const int kw = kt + NW - 1, sw = (rel + NW - 1) % NW;      // the W step issued now, into the stage everybody finished reading before the barrier just passed
        auto mem_w = [&](int half) { issue_w2(kw, sw, half); };                                   // W pieces: hi plane, then lo plane
        auto mem_a = [&](int half) { if (half == 0) load_a2(kt + NA + 1, setc, I0{}); else load_a2(kt + NA + 1, setc, H1{}); };
        __builtin_amdgcn_sched_barrier(0);
        if (w < 4) {
            // in flight, oldest first: [A(kt+1)] [W(kt+1)] [A(kt+2)] (DEEP) / [A(kt+1)]
            wait_a(setc, IX{});
            convert_a(cur ^ 1, setc);
            __builtin_amdgcn_sched_barrier(0);
            multiply_half(cur, wst, 0, mem_w);
            multiply_half(cur, wst, 1, mem_a);
        } else {
            multiply_half(cur, wst, 0, mem_w);
            __builtin_amdgcn_sched_barrier(0);
            wait_a(setc, IW{});          // .. plus the four W pieces just issued
            convert_a(cur ^ 1, setc);
            __builtin_amdgcn_sched_barrier(0);
            multiply_half(cur, wst, 1, mem_a);
        }
        __builtin_amdgcn_sched_barrier(0);
        // my W pieces of step kt + 1 have landed (what was issued behind them may still fly), my plane stores are done
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VMW) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int kt = kt0; kt < kt1; kt += 2) {
        step(kt, I1{});                           // A(kt + 1) sits in the second set on even steps (the only set without DEEP)
        if (kt + 1 < kt1) step(kt + 1, I0{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the loads of the steps behind the last one (discarded)
    asm volatile("" : "+v"(araw[0][0]), "+v"(araw[0][1]), "+v"(araw[0][2]), "+v"(araw[0][3]));
    asm volatile("" : "+v"(araw[NA - 1][0]), "+v"(araw[NA - 1][1]), "+v"(araw[NA - 1][2]), "+v"(araw[NA - 1][3]));

    if (FEED && qpart >= 0) {
        // one quarter of a split tile: park the accumulators write-through, count the arrival; the LAST of the four sums the quarters in
        // index order and goes on to the epilogue, the others take their next work item.  Slab layout [32 vectors][512 threads][4 floats]: a
        // wave instruction moves 1 KB of consecutive bytes (round 5, measured on the six-product kernel: with 4-byte accesses the last arriver
        // of a feed's FIRST tile needed 56 us for its sum and epilogue -- parked at 72 us, counted at 128; gemm_x6.hip)
        __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(g.slabs, 0, (int)(unsigned)((size_t)g.split_tiles * 4 * SLAB_FLOATS * 4), 0x00020000);
        const unsigned sbq = (tile * 4u + (unsigned)qpart) * 32u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const u32x4_t v4 = {__float_as_uint(acc[i][j][4 * r4]), __float_as_uint(acc[i][j][4 * r4 + 1]),
                                        __float_as_uint(acc[i][j][4 * r4 + 2]), __float_as_uint(acc[i][j][4 * r4 + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v4, srs, ((sbq + (unsigned)((i * 2 + j) * 4 + r4)) * 512u + (unsigned)tid) * 16u, 0, 16);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* mailbox = reinterpret_cast<unsigned*>(smem + LDS_BYTES);
        if (tid == 0) *mailbox = __hip_atomic_fetch_add(g.arrive + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned before = *mailbox;
        __syncthreads();
        if (before != 3u) continue;
        // total = ((q0 + q1) + q2) + q3, whoever arrives last
        for (unsigned qq = 0; qq < 4u; ++qq) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const u32x4_t v4 = __builtin_amdgcn_raw_buffer_load_b128(srs, (((tile * 4u + qq) * 32u + (unsigned)((i * 2 + j) * 4 + r4)) * 512u + (unsigned)tid) * 16u, 0, 16);
                        acc[i][j][4 * r4]     = qq == 0 ? __uint_as_float(v4.x) : acc[i][j][4 * r4]     + __uint_as_float(v4.x);
                        acc[i][j][4 * r4 + 1] = qq == 0 ? __uint_as_float(v4.y) : acc[i][j][4 * r4 + 1] + __uint_as_float(v4.y);
                        acc[i][j][4 * r4 + 2] = qq == 0 ? __uint_as_float(v4.z) : acc[i][j][4 * r4 + 2] + __uint_as_float(v4.z);
                        acc[i][j][4 * r4 + 3] = qq == 0 ? __uint_as_float(v4.w) : acc[i][j][4 * r4 + 3] + __uint_as_float(v4.w);
                    }
        }
    }

    // epilogue (branch-free): 32x32 accumulator layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int cl = lane & 31, rq = lane >> 5;
    __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(unsigned)((size_t)g.M * g.ldc * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dact_y ? g.dact_y : g.C), 0, (int)(unsigned)((size_t)g.M * g.ldc * 4), 0x00020000);
    float bsum[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bsum[j] = g.bias ? g.bias[n0 + wn * 64 + j * 32 + cl] : 0.f;
    if (g.dact_y) {            // two straight-line loops: a merge point inside one brings a per-tile s_waitcnt vmcnt(0) back
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned o0 = (unsigned)(((size_t)(m0 + wm * 128 + i * 32 + 4 * rq) * g.ldc + n0 + wn * 64 + j * 32 + cl) * 4);
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
            for (int j = 0; j < 2; ++j) {
                const unsigned o0 = (unsigned)(((size_t)(m0 + wm * 128 + i * 32 + 4 * rq) * g.ldc + n0 + wn * 64 + j * 32 + cl) * 4);
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
}  // namespace c256
}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Weight gradients dW = dY^T X (model.py:39-44 backward: dW_ih = dgates^T x, dW_hh = dgates^T h_prev) on the same
// 256 x 256 / 8-wave structure.  BOTH operands are fp32 activations stored k-major (rows = (t, b); the M / N index is
// contiguous), K = T*B is long and the output small, so the product is split over K into slabs that gemm.hip's
// gemm_reduce_kernel sums in index order (deterministic) -- this kernel is reached through pgasr_gemm_f32.
//   * no LDS-DMA: a thread loads 8 x 16 B per 32-deep step (a wave instruction = one 1-KB k-row, fully coalesced),
//     splits the 32 values into bf16 hi / lo and writes both planes with 8-byte stores into a k-major LDS image
//     [k][256 + 32 pad] (pitch 576 B = 64 mod 256: the four k-rows of a transposing read's two 4 x 16 blocks cover the
//     64 banks once) -- the split happens once per workgroup, not once per wave as in gemm.hip's 128 x 128 kernel;
//   * MFMA fragments (8 k of one m) come out of ds_read_b64_tr_b16, as in gemm.hip's TN kernel;
//   * two LDS buffers of 72 KB, one raw s_barrier per step; the loads of step t + 2 are issued as soon as the registers
//     of step t + 1 have been converted, i.e. a full step ahead of their use, and are NOT drained by the barrier
//     (a plain __syncthreads() would wait for them);
//   * the two waves of a SIMD are half a step out of phase (waves 4-7 multiply the first 16-deep half before they
//     convert), so one wave's VALU / LDS-write phase sits beside its partner's MFMA phase;
//   * 256 x 256 tiles halve the operand bytes per flop of the 128 x 128 kernel (16 x 1000 steps x 64 KB = 1 GB against
//     2 GB on the dW_ih shape).
// Requirements: pgasr_internal_tn256_ok.
// ------------------------------------------------------------------------------------------------------------------
namespace t256 {
constexpr int TM = 256, TN = 256, TK = 32, THREADS = 512;
constexpr int PITCH = 288;                        // halfs per k-row of a plane image (576 B)
constexpr int PLANE_HALFS = TK * PITCH;           // 9216 halfs = 18 KB
constexpr int BUF_HALFS = 4 * PLANE_HALFS;        // A hi | A lo | B hi | B lo = 72 KB
constexpr int LDS_BYTES = 2 * BUF_HALFS * 2;      // 144 KB
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ bf16x8_t tr_frag(const unsigned short* p) {
    typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_ptr;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * PITCH));
    return __builtin_bit_cast(bf16x8_t, (s16x8_t){a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w});
}

// GATED: operand A is the d(pre-activation) tensor of a backward LSTM sweep that is STILL RUNNING on other XCDs
// (pgasr_lstm_layer_bwd_streamed).  Queue mode only.  The K-slabs (time ranges) are visited in the order in which the sweep
// completes them -- direction 1 walks time upwards, direction 0 downwards; an item's direction is the half of the tensor's row its
// A columns lie in -- and before its first load an item waits until the sweep's slab_done words cover its rows.  A is then read
// with agent-scope (sc1) loads: the rows were written back to memory by the producer's L2, this XCD's L2 may still hold lines of
// an earlier tenant of the same addresses.  Every item computes exactly what the plain kernel computes for the same slab, so the
// result has the same bits; the wait is bounded (gate_err).
// Two products in one queue-mode launch (g1.M > 0; same splitk): a layer's dW_ih and dW_hh share the sweep they wait for, and as
// two launches on one stream the second would start when the first has seen the sweep's LAST slab.  Items are dealt slab by slab:
// the tiles of product 0's slab j, then those of product 1's slab j.
template <bool GATED>
__global__ __launch_bounds__(THREADS) void gemm_t256_kernel(PgasrTn256Args g0, PgasrTn256Args g1) {
    extern __shared__ __attribute__((aligned(128))) unsigned short S[];      // the ONLY LDS object: [buffer][A hi, A lo, B hi, B lo][k][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    if (g0.queue && g0.xcc_busy) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;
        if (__hip_atomic_load(g0.xcc_busy + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    }
    const unsigned per0 = (unsigned)((g0.N / TN) * (g0.M / TM) * g0.batch), per1 = g1.M > 0 ? (unsigned)((g1.N / TN) * (g1.M / TM) * g1.batch) : 0u;
    const unsigned nitems_all = (per0 + per1) * (unsigned)g0.splitk;
    // lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3 of the group's 4 x 16 block; groups 0,1 take
    // columns 0-15 / 16-31 of k 0-7, groups 2,3 the same columns of k 8-15 (= the 32x32x16 operand map)
    const int tro = (8 * (lane >> 5) + ((lane & 15) >> 2)) * PITCH + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    bool gate_dead = false;      // GATED: the sweep's error word has been seen set (or a wait gave up): later items do not wait
  for (;;) {
    unsigned item;
    bool second = false;
    if (g0.queue) {
        unsigned* mailbox = reinterpret_cast<unsigned*>(S);       // the buffers are idle between two items
        if (tid == 0) *mailbox = __hip_atomic_fetch_add(g0.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        item = (unsigned)__builtin_amdgcn_readfirstlane((int)*mailbox);      // wave-uniform for the compiler too (scalar address arithmetic)
        __syncthreads();
        if (item >= nitems_all) return;
        if (per1) {
            const unsigned jj = item / (per0 + per1), r = item % (per0 + per1);
            second = r >= per0;
            item = second ? jj * per1 + (r - per0) : jj * per0 + r;
        }
    }
    const PgasrTn256Args g = second ? g1 : g0;
    const int tx = g.N / TN, ty = g.M / TM;
    const unsigned nitems = (unsigned)(tx * ty) * (unsigned)(g.batch * g.splitk);
    if (!g0.queue) {
        // plain launch: workgroups are dealt round-robin over the 8 XCDs (observed, speed only), so workgroup b shares an L2
        // with b + 8, b + 16, ..: give every XCD WHOLE K-slabs -- the tiles of a slab read the same k-rows (an A byte is
        // read by N/256 of them, a B byte by M/256), and side by side behind one L2 they fetch them from HBM once
        const unsigned per = (unsigned)(tx * ty), nslab = (unsigned)(g.batch * g.splitk);
        const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const unsigned zz = xcd + 8u * (j / per);
        item = (nitems % (8u * per) == 0u && nslab % 8u == 0u) ? zz * per + j % per : blockIdx.x;
    }
    if (item >= nitems) return;
    // items of one K-slab are adjacent: they run at the same time and share the slab's rows in L2 / MALL
    int z = (int)(item / (unsigned)(tx * ty)), t2 = (int)(item % (unsigned)(tx * ty));
    int bidx = z / g.splitk, sidx = z % g.splitk;
    int k_beg = sidx * g.kper, k_end = (k_beg + g.kper < g.K) ? k_beg + g.kper : g.K;
    if (g.tslabs) {
        // item = (time slab in the order a backward sweep completes them, batch, tile); see pgasr_wslab_edge (common.h)
        const unsigned per = (unsigned)(tx * ty * g.batch);
        const int jj = (int)(item / per), rem = (int)(item % per);
        bidx = rem / (tx * ty); t2 = rem % (tx * ty);
        const long long off = (long long)(g.A - g.gate_base) + (long long)bidx * g.sA + (long long)(t2 / tx) * TM;
        const int row_off = (int)(off / g.lda), dir = (off % g.lda) >= g.lda / 2 ? 1 : 0;
        const int n = g.splitk, T = g.gate_T;
        const int h_lo = pgasr_wslab_edge(T, n - jj - 1), h_hi = pgasr_wslab_edge(T, n - jj);
        const long long ra = (long long)(dir ? T - h_hi : h_lo) * g.gate_B - row_off, rb = (long long)(dir ? T - h_lo : h_hi) * g.gate_B - row_off;
        k_beg = ra < 0 ? 0 : (int)ra; k_end = rb > g.K ? g.K : (int)rb;
        sidx = jj; z = bidx * g.splitk + sidx;
        if constexpr (GATED) {
            if (tid < 64 && !gate_dead) {
                const int t_lo = (row_off + k_beg) / g.gate_B, t_hi = (row_off + k_end - 1) / g.gate_B;
                const int s_last = dir ? t_hi : T - 1 - t_lo;          // the last sweep step that writes one of these rows
                unsigned need = 1;                                      // publication k covers sweep steps < T - h_(n-k)
                while ((int)need < n && T - pgasr_wslab_edge(T, n - (int)need) <= s_last) ++need;
                unsigned spins = 0; long long t0 = 0;
                while (true) {
                    asm volatile("" ::: "memory");
                    unsigned v = need;
                    if (lane < g.gate_nbg) v = __hip_atomic_load(g.gate + 2 * lane + dir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!__any(v < need)) break;
                    // a sweep that gave up (its sticky error word) publishes nothing more: do not sit out 3 s per work item
                    if (g.gate_err && __hip_atomic_load(g.gate_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { gate_dead = true; break; }
                    __builtin_amdgcn_s_sleep(32);
                    if (((++spins) & 255u) == 0) {
                        const long long now = wall_clock64();
                        if (spins == 256u) t0 = now;
                        else if (now - t0 > 300000000LL) { if (g.gate_err) *g.gate_err = 1; gate_dead = true; break; }     // 3 s of the 100 MHz clock
                    }
                }
            }
            __syncthreads();
        }
    }
    const int tbx = t2 % tx, tby = t2 / tx;
    const int nk = (k_end - k_beg) / TK;               // >= 1 (pgasr_internal_tn256_ok: no empty slab)
    const int m0 = tby * TM, n0 = tbx * TN;
    // wave w loads k-rows 4 w + j (j = 0..3) of both operands: one 1-KB row per wave instruction
    const float* Ab = g.A + (size_t)bidx * g.sA + (size_t)(k_beg + 4 * w) * g.lda + m0 + 4 * lane;
    const float* Bb = g.B + (size_t)bidx * g.sB + (size_t)(k_beg + 4 * w) * g.ldb + n0 + 4 * lane;
    f32x4_t ra[4], rb[4];
    // GATED: this item's A rows as a buffer (kper * lda * 4 < 2^31, checked by the launcher): sc1 loads
    __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + (size_t)bidx * g.sA + (size_t)k_beg * g.lda + m0), 0,
                                                                   GATED ? (int)((size_t)nk * TK * g.lda * 4) : 0, 0x00020000);
    auto load = [&](int kt) {
        const int kc = kt < nk ? kt : nk - 1;          // past the slab: reload the last step (never converted)
        const float* pa = Ab + (size_t)kc * TK * g.lda;
        const float* pb = Bb + (size_t)kc * TK * g.ldb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (GATED) {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                const u32x4_t u = __builtin_amdgcn_raw_buffer_load_b128(ars, (unsigned)(((size_t)(kc * TK + 4 * w + j) * g.lda + 4 * lane) * 4), 0, 16);
                ra[j] = __builtin_bit_cast(f32x4_t, u);
            } else {
                ra[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(pa + (size_t)j * g.lda));
            }
            rb[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(pb + (size_t)j * g.ldb));
        }
    };
    auto convert = [&](int buf) {
        unsigned short* base = S + buf * BUF_HALFS + (4 * w) * PITCH + 4 * lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned h0, l0, h1, l1;
            split2(ra[j].x, ra[j].y, h0, l0); split2(ra[j].z, ra[j].w, h1, l1);
            *reinterpret_cast<u32x2_t*>(base + j * PITCH) = (u32x2_t){h0, h1};
            *reinterpret_cast<u32x2_t*>(base + PLANE_HALFS + j * PITCH) = (u32x2_t){l0, l1};
            split2(rb[j].x, rb[j].y, h0, l0); split2(rb[j].z, rb[j].w, h1, l1);
            *reinterpret_cast<u32x2_t*>(base + 2 * PLANE_HALFS + j * PITCH) = (u32x2_t){h0, h1};
            *reinterpret_cast<u32x2_t*>(base + 3 * PLANE_HALFS + j * PITCH) = (u32x2_t){l0, l1};
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto multiply = [&](int buf, int ks) {             // one 16-deep half of a step: 24 MFMAs
        const unsigned short* img = S + buf * BUF_HALFS + ks * 16 * PITCH + tro;
        bf16x8_t ah[4], al[4], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = tr_frag(img + wm * 128 + i * 32);
            al[i] = tr_frag(img + PLANE_HALFS + wm * 128 + i * 32);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bh[j] = tr_frag(img + 2 * PLANE_HALFS + wn * 64 + j * 32);
            bl[j] = tr_frag(img + 3 * PLANE_HALFS + wn * 64 + j * 32);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            }
    };
    // a raw barrier: __syncthreads() would also wait (vmcnt) for the loads of the step after next that are in flight
#define T256_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

    load(0);
    convert(0);
    load(1);
    T256_BARRIER();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool mem = !(g.diag & 1), mul = !(g.diag & 2);
        if (w < 4) {
            if (mem) { if (kt + 1 < nk) convert(cur ^ 1); load(kt + 2); }
            if (mul) { multiply(cur, 0); multiply(cur, 1); }
        } else {
            if (mul) multiply(cur, 0);
            if (mem) { if (kt + 1 < nk) convert(cur ^ 1); load(kt + 2); }
            if (mul) multiply(cur, 1);
        }
        T256_BARRIER();
    }
#undef T256_BARRIER

    // raw alpha * acc into this item's slab (branch-free buffer stores)
    float* slab = g.partial + (size_t)z * g.M * g.N;
    __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)(unsigned)((size_t)g.M * g.N * 4), 0x00020000);
    const int cl = lane & 31, rq = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned o0 = (unsigned)(((size_t)(m0 + wm * 128 + i * 32 + 4 * rq) * g.N + n0 + wn * 64 + j * 32 + cl) * 4);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(g.alpha * acc[i][j][r]), prs, o0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.N * 4), 0, 0);
        }
    if (!g0.queue) return;
    __syncthreads();           // every read of this item's last buffer is done before the mailbox / the next item's images are written
  }
}
}  // namespace t256

// ---- internal entry points used by gemm_dma.hip's C ABI functions ----
size_t pgasr_internal_x3c_slab_bytes() { return (size_t)c256::SLAB_FLOATS * 4; }

int pgasr_internal_x3c_launch(const PgasrX3cArgs& a, hipStream_t st) {
    DmaGemmArgs g{a.A, a.Whi, a.Wlo, a.C, a.M, a.N, a.K, a.lda, a.ldc, a.bias, a.dact_y, a.slope,
                  a.queue, a.xcc_busy, a.tiles_done, a.mt_count, a.nt_count, a.order, a.quarters, a.split_tiles, a.slabs, a.arrive};
    if (!a.feed) {
        const size_t lds = (size_t)c256::LDS_BYTES_DEEP;
        if (hipFuncSetAttribute((const void*)c256::gemm_x3c_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PGASR_ERR_LAUNCH;
        PGASR_LAUNCH_KERNEL(c256::gemm_x3c_kernel<false>, dim3((unsigned)(a.N / c256::TN), (unsigned)((a.M + c256::TM - 1) / c256::TM)), dim3(c256::THREADS), lds, st, g);
        PGASR_CHECK_LAUNCH();
        return PGASR_OK;
    }
    const size_t lds = (size_t)c256::LDS_BYTES + 16;
    if (hipFuncSetAttribute((const void*)c256::gemm_x3c_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    for (int pass = 0; pass < 2; ++pass) {     // one persistent workgroup per CU; pass 1 ignores the busy counters
        if (pass == 1) g.xcc_busy = nullptr;
        PGASR_LAUNCH_KERNEL(c256::gemm_x3c_kernel<true>, dim3(256), dim3(c256::THREADS), lds, st, g);
        PGASR_CHECK_LAUNCH();
    }
    return PGASR_OK;
}

bool pgasr_internal_tn256_ok(const PgasrTn256Args& a, int tk) {
    if (tk != 16 && tk != 32) return false;
    if (a.gate && (!a.queue || !a.tslabs || a.gate_nbg <= 0 || a.gate_nbg > 32)) return false;
    if (a.tslabs) {
        if (!a.gate_base || a.gate_T <= 0 || a.gate_B <= 0 || (a.gate_B % tk) || a.A < a.gate_base || (a.lda & 1)) return false;
        if (a.splitk != pgasr_wslab_count(a.gate_T) || (size_t)a.gate_T * a.gate_B * a.lda * 4 >= ((size_t)1 << 31)) return false;
        // no empty slab: every batch's rows must reach into the first and the last 16 frames
        const long long first = (long long)((a.A - a.gate_base) / a.lda), last = first + a.K;
        if (first >= 16LL * a.gate_B || last <= (long long)(a.gate_T - 16) * a.gate_B) return false;
    }
    // PGASR_TN_TILE=128 (read at every call) keeps gemm.hip's 128 x 128 kernel (A/B measurements)
    const char* e = getenv("PGASR_TN_TILE");
    if (e && e[0] == '1' && e[1] == '2' && e[2] == '8') return false;
    if (!a.A || !a.B || !a.partial || a.M <= 0 || a.N <= 0 || a.K < 32 || a.batch <= 0 || a.splitk <= 0) return false;
    if ((a.M % t256::TM) || (a.N % t256::TN) || (a.K % tk) || (!a.tslabs && (a.kper % 32)) || (a.lda & 3) || (a.ldb & 3)) return false;
    if ((a.sA & 3) || (a.sB & 3) || (((size_t)a.A) & 15) || (((size_t)a.B) & 15)) return false;
    if (!a.tslabs && (long long)(a.splitk - 1) * a.kper >= a.K) return false;        // no empty slab
    if ((size_t)a.M * a.N * 4 >= ((size_t)1 << 32)) return false;
    return true;
}

int pgasr_internal_tn256_launch(PgasrTn256Args a, int masked_then_unmasked, hipStream_t st, const PgasrTn256Args* second) {
    PgasrTn256Args b{};
    if (second) {
        if (!a.queue || second->splitk != a.splitk || !pgasr_internal_tn256_ok(*second)) return PGASR_ERR_UNSUPPORTED;
        b = *second;
    }
    const size_t lds = (size_t)t256::LDS_BYTES;
    if (hipFuncSetAttribute((const void*)t256::gemm_t256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    const unsigned nitems = (unsigned)((a.N / t256::TN) * (a.M / t256::TM)) * (unsigned)(a.batch * a.splitk);
    if (!a.queue) {
        PGASR_LAUNCH_KERNEL(t256::gemm_t256_kernel<false>, dim3(nitems), dim3(t256::THREADS), lds, st, a, b);
        PGASR_CHECK_LAUNCH();
        return PGASR_OK;
    }
    if (a.gate && hipFuncSetAttribute((const void*)t256::gemm_t256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    const unsigned* busy = a.xcc_busy;
    for (int pass = 0; pass < ((masked_then_unmasked && busy) ? 2 : 1); ++pass) {
        a.xcc_busy = (pass == 0) ? busy : nullptr;
        b.xcc_busy = a.xcc_busy;
        if (a.gate) PGASR_LAUNCH_KERNEL(t256::gemm_t256_kernel<true>, dim3(256), dim3(t256::THREADS), lds, st, a, b);
        else        PGASR_LAUNCH_KERNEL(t256::gemm_t256_kernel<false>, dim3(256), dim3(t256::THREADS), lds, st, a, b);
        PGASR_CHECK_LAUNCH();
    }
    return PGASR_OK;
}
