// The path's big products in the REFERENCE'S OWN arithmetic (torch fp32: nn.LSTM's hoisted W_ih products, model.py:38-44) at
// bf16-MFMA rate: every fp32 operand is split into THREE bf16 planes (hi, mid, lo: 8 + 8 + 8 mantissa bits, x = hi + mid + lo to
// 2^-24) and a product is the SIX terms hh + hm + mh + hl + lh + mm -- every term down to 2^-24 of the product, fp32 accumulate;
// the same split the recurrent sweeps' NP = 3 instantiation uses (lstm.hip).  This file is the "f32" precision mode's counterpart
// of gemm_c256.hip (x3c: C = A W^T with pre-split W planes; t256: dW = dY^T X), so that the fp32-faithful step runs the same
// feed-ahead and streamed orders as the bf16x3 one instead of 114-TF fp32-MFMA GEMMs in front of its sweeps.
//
// Both kernels keep gemm_c256.hip's 256 x 256 tile / 8 waves x (128 x 64) / one raw barrier per step, but a step is 16 deep:
// six products on a 16-deep step are the 48 MFMAs per wave that three products on a 32-deep step were, the LDS image of a step
// (three planes) is 3/4 of the old one, and the bytes a CU pulls per MFMA HALVE (40 KB per 48 MFMAs against 64 KB) -- the x3c
// kernel sat at its CU's memory-path floor with the MFMA pipe 46 % busy, so the second three products are nearly free.
//   x6c: A (fp32) loaded to registers THREE steps ahead (3 sets of 2 x 16 B per thread), split once per workgroup into the three
//        LDS planes ([row][16 k] bf16, 32-byte rows, chunk ^= (row >> 3) & 1); W planes (pre-split, pgasr_split_bf16_planes3) by
//        LDS-DMA into FOUR stages (three steps ahead): ~120 KB in flight per CU; hand-counted s_waitcnt vmcnt (5 memory
//        instructions per wave and step: 3 W pieces, then 2 A loads).
//   t6:  both operands fp32, k-major; registers two steps ahead (2 sets), planes in a k-major LDS image read with
//        ds_read_b64_tr_b16; queue / time-slab / gated modes exactly as t256.
// Results: one fp32 accumulation chain over K per tile (x6c; the feed's first tiles with K >= 1024: fixed-order K-quarter sums,
// and the sequential order runs the same kernel -- same bits), slab sums in index order (t6).
#include "common.h"
#include "x3w_common.h"
#include <type_traits>

// Diagnostic build only (make diag: -DPGASR_X6_DIAG; results invalid): PGASR_X6_DIAG=<bits> switches parts of the plain x6c kernel's
// k-loop off -- 1 no W DMA, 2 no MFMA, 4 no A loads, 8 no conversion, 16 no fragment reads, 32 no per-step barrier
#ifdef PGASR_X6_DIAG
#define X6D(bit) (!FEED && (g.single & (bit)))
// in-kernel stamps (diagnostic build): every wave of workgroup 0 accumulates the cycles between consecutive stamps per segment
__device__ long long x6_stamp_out[8][16];
#define X6STAMP_DECL long long st_last_ = clock64(); long long st_acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const long long st_c0_ = clock64(), st_r0_ = wall_clock64()
#define X6STAMP(slot) do { const long long now_ = clock64(); st_acc_[slot] += now_ - st_last_; st_last_ = now_; } while (0)
#define X6STAMP_FLUSH() do { if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) { for (int i_ = 0; i_ < 12; ++i_) x6_stamp_out[w][i_] = st_acc_[i_]; \
                             x6_stamp_out[w][12] = clock64() - st_c0_; x6_stamp_out[w][13] = wall_clock64() - st_r0_; } } while (0)
#else
#define X6D(bit) false
#define X6STAMP_DECL
#define X6STAMP(slot) do { } while (0)
#define X6STAMP_FLUSH() do { } while (0)
#endif

namespace {
#ifdef PGASR_LSTM_DIAG
__device__ unsigned x6_feed_diag[4096];      // [0] wall clock at workgroup 0's start, [16 + dir * mt + row tile] when that row tile was counted (the LAST feed that ran)
#endif
namespace x6c {
constexpr int TM = 256, TN = 256, TK = 16, THREADS = 512;
constexpr int NW = 4;                            // W stages in LDS: the DMA runs three steps ahead
constexpr int NA = 3;                            // A register sets: the loads run three steps ahead of their conversion
constexpr int AP_BYTES = TM * TK * 2;            // one A plane of one buffer: 8 KB
constexpr int ABUF_BYTES = 3 * AP_BYTES;         // hi | mid | lo
constexpr int WP_BYTES = TN * TK * 2;            // one W plane of one stage: 8 KB
constexpr int WSTAGE_BYTES = 3 * WP_BYTES;       // hi | mid | lo
constexpr int LDS_W = 2 * ABUF_BYTES;            // [A buffer 0][A buffer 1][W stage 0 .. 3][mailbox (FEED)]
constexpr int LDS_BYTES = LDS_W + NW * WSTAGE_BYTES;     // 144 KB
constexpr int SLAB_FLOATS = 128 * THREADS;       // one parked accumulator set: 256 KB

typedef u32x4_t Planes2[2][3];                   // two 32-row tiles x (hi, mid, lo): 8 bf16 of one row each

// fragment reads in inline asm: hipcc drains the LDS-DMA in flight (s_waitcnt vmcnt(0)) in front of any LDS access it can see
__device__ __forceinline__ void read_pair(Planes2& o, unsigned p0, unsigned p1) {
    static_assert(AP_BYTES == 8192 && WP_BYTES == 8192, "plane offsets are spelled in the asm below");
    asm volatile("ds_read_b128 %0, %6\n\t"
                 "ds_read_b128 %1, %6 offset:8192\n\t"
                 "ds_read_b128 %2, %6 offset:16384\n\t"
                 "ds_read_b128 %3, %7\n\t"
                 "ds_read_b128 %4, %7 offset:8192\n\t"
                 "ds_read_b128 %5, %7 offset:16384"
                 : "=&v"(o[0][0]), "=&v"(o[0][1]), "=&v"(o[0][2]), "=&v"(o[1][0]), "=&v"(o[1][1]), "=&v"(o[1][2])
                 : "v"(p0), "v"(p1)
                 : "memory");
}
__device__ __forceinline__ void wait_pair(Planes2& o) {      // claims the registers the reads above are filling
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(o[0][0]), "+v"(o[0][1]), "+v"(o[0][2]), "+v"(o[1][0]), "+v"(o[1][1]), "+v"(o[1][2])
                 :: "memory");
}
__device__ __forceinline__ void wait_pairs(Planes2& a, Planes2& b) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]),
                   "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2])
                 :: "memory");
}
__device__ __forceinline__ void read_row(u32x4_t (&o)[3], unsigned p) {       // one 32-row tile: hi, mid, lo
    asm volatile("ds_read_b128 %0, %3\n\t"
                 "ds_read_b128 %1, %3 offset:8192\n\t"
                 "ds_read_b128 %2, %3 offset:16384"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]) : "v"(p) : "memory");
}
__device__ __forceinline__ void wait_row(u32x4_t (&o)[3]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]) :: "memory");
}
__device__ __forceinline__ void wait_row_pair(u32x4_t (&a)[3], Planes2& b) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2])
                 :: "memory");
}
// three 8-byte LDS stores (hi plane, mid = + 8 KB, lo = + 16 KB), hidden from hipcc like the reads
__device__ __forceinline__ void write_planes3(unsigned addr, unsigned h0, unsigned h1, unsigned m0, unsigned m1, unsigned l0, unsigned l1) {
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    const u2 h = {h0, h1}, m = {m0, m1}, l = {l0, l1};
    asm volatile("ds_write_b64 %0, %1\n\t"
                 "ds_write_b64 %0, %2 offset:8192\n\t"
                 "ds_write_b64 %0, %3 offset:16384" :: "v"(addr), "v"(h), "v"(m), "v"(l) : "memory");
}
// A loads in inline asm: a load hipcc can see is waited for with s_waitcnt vmcnt(0) where its registers are handed to the asm
// wait below -- which would also drain the W pieces in flight
template <int J>
__device__ __forceinline__ void load_a_one(u32x4_t (&r)[2], const unsigned (&aoff)[2], unsigned kb, const float* A) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r[J]) : "v"(aoff[J] + kb), "s"(A) : "memory");
}
template <int CNT>
__device__ __forceinline__ void wait_a_regs(u32x4_t (&r)[2]) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"(CNT) : "memory");
}

// acc[i][j] += A_i x B_j as hh + hm + mh + hl + lh + mm, the two column tiles interleaved (no MFMA waits for its predecessor's result)
__device__ __forceinline__ void mfma6(f32x16 (&acc)[2], const u32x4_t (&a)[3], const Planes2& b) {
    constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[PA[p]]), __builtin_bit_cast(bf16x8_t, b[j][PB[p]]), acc[j], 0, 0, 0);
}

// Round 5 note -- why the A operand is still converted in the k-loop.  The review asked for activations delivered as three bf16 planes by
// their producers (the sweeps' storer waves, the affine / dropout epilogues), so that both operands become LDS-DMA operands.  The upper
// bounds were already measured with this kernel's diagnostic build (round 4, NOTES.md 1.3, 1.5): the input-gradient shape runs 410 us, 384
// WITHOUT any conversion (-6 %), and the weight-gradient kernel 373 -> 374 -> 376 us with its B operand left unsplit, unstored and then
// unloaded (0 %): what these kernels pay for is MFMA issue (1.7-1.9 PF is what the chip sustains on bare MFMAs), fragment reads and the step
// barrier.  Against at most -6 % on one of the two kernels stand +50 % bytes written by every sweep (393 instead of 262 MB per layer, by
// sweeps that slow down 24 % beside 2.9 TB/s of foreign traffic) and a second 393 MB buffer per layer.  Not built.
// VAR (PGASR_X6_VAR, read at every call; bit-identical results): 7 (default) = SELF-INTERLEAVED, every wave carries its own non-MFMA
// work as fillers behind its own MFMAs; 0 = gemm_c256.hip's structure (two row tiles' fragments read, waited for, multiplied; the
// conversion as a block, the two waves of a SIMD half a step out of phase).  Round 4 measured five more structures on one box --
// fragment reads pipelined one row tile ahead (1), that with the conversion in two halves (2), PING-PONG roles with a barrier
// between the phases (3), without it (4), with s_setprio 3 on the memory phase (5), VAR 0 with s_setprio 3 around the conversion
// (6) -- all within +-3 % of VAR 0 (410-440 us on the two shapes).  In-kernel stamps said why: while one wave of a SIMD issues
// back-to-back MFMAs its partner gets about ONE instruction per MFMA (a 44-VALU conversion beside the partner's 48 MFMAs took 1,880
// cycles, whatever the priorities), so a "memory phase beside a multiply phase" does not overlap -- with the k-loop's pieces switched
// off in turn, MFMA + fragment reads alone took 322 us, everything but the MFMAs 180 us, together 410: they ADD.  And a whole
// chip of bare v_mfma_f32_32x32x16_bf16 on random operands sustains 1.7-1.9 PF at 1.7-1.86 GHz (tools/mfma_peak.hip), not 2.5.
// A ONE-wave-per-SIMD form of VAR 7 (four waves x 128 x 128, 256 accumulators in AGPRs, ~2 fillers per MFMA gap; NOTES.md 1.3, not in the
// tree) was built too: correct, bit-identical, 403 / 376 us against 420 / 382 -- 3 %, and its FEED form spilled; not kept.  What a
// filler costs beside back-to-back MFMAs with nobody else on the SIMD is in profiles/r04_mfma_filler.txt (tools/mfma_filler.hip):
// VALU up to ~4 per gap nearly free (32.7 -> 35 cycles per MFMA), but one ds_read_b128 per gap +9 cycles and LDS writes far more --
// the LDS pipe, not the issue port, is what the fragment reads and plane stores of these kernels pay for.
template <bool FEED, int VAR>
__global__ __launch_bounds__(THREADS) void gemm_x6c_kernel(DmaGemmArgs g) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];   // the ONLY LDS object
#ifdef PGASR_LSTM_DIAG
    if (FEED && threadIdx.x == 0 && blockIdx.x == 0) x6_feed_diag[0] = (unsigned)wall_clock64();     // workgroup 0's start (of the LAST feed that ran)
#endif
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;      // (w made wave-uniform for the compiler -- readfirstlane -- spills: measured, not kept)
    const int wm = w >> 2, wn = w & 3;
    const int nk = g.K / TK;
    bool head_only = false;         // a workgroup on one of the sweep's XCDs (placement is read, not assumed) ..
    if (FEED && g.xcc_busy) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;
        if (__hip_atomic_load(g.xcc_busy + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            if (!g.head_help) return;       // .. leaves at once,
            head_only = true;               // .. or (round 5, PGASR_X6_HEAD_HELP) helps with the K-split head items first: the eleven CUs per XCD that a
        }                                   // sweep leaves idle are free the moment it starts, while the free XCDs' CUs may still belong to the layer above's
    }                                       // weight-gradient workgroups -- and a sweep that waits for its first rows has no L2 traffic to disturb
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  for (;;) {
    int tbx, tby;
    int kt0 = 0, kt1 = nk, qpart = -1;      // step range of this work item; qpart >= 0: part qpart of a tile split into `parts` K ranges
    unsigned tile = 0, parts = 1u, slab0 = 0u;     // slab0: the tile's first parked accumulator set
#ifdef PGASR_LSTM_DIAG
    unsigned diag_t_ = 0xFFFFFFFFu;
#endif
    if (FEED) {
        unsigned* mailbox = reinterpret_cast<unsigned*>(smem + LDS_BYTES);
        // queue order: S8 tiles in eighths, S tiles in quarters, S2 tiles in halves, the rest whole (round 5: the graded head -- see the host side)
        const unsigned S8 = (unsigned)g.split8_tiles, S = (unsigned)g.split_tiles, S2 = (unsigned)g.split2_tiles, ntot = (unsigned)g.mt_count * (unsigned)g.nt_count;
        const unsigned n8 = 8u * S8, n4 = 4u * S, n2 = 2u * S2;
        if (tid == 0) {
            // a head-only workgroup looks before it draws: past the head it leaves without taking an item (a late look may still draw one
            // whole tile: harmless)
            if (head_only && __hip_atomic_load(g.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n8 + n4) *mailbox = 0xFFFFFFFFu;
            else *mailbox = __hip_atomic_fetch_add(g.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const unsigned t = *mailbox;
        __syncthreads();
        if (t >= ntot + 7u * S8 + 3u * S + S2) return;             // 8 S8 + 4 S + 2 S2 part items, then the remaining whole tiles
#ifdef PGASR_LSTM_DIAG
        if (tid == 0 && t < 4u) x6_feed_diag[4 + t] = (unsigned)wall_clock64();          // the first four items: drawn at ..
        if (tid == 0 && t >= n8 + n4 + n2 && t < n8 + n4 + n2 + 4u) x6_feed_diag[1024 + (t - n8 - n4 - n2)] = (unsigned)wall_clock64();      // .. and the first four whole tiles
        diag_t_ = t;
#endif
        if (t < n8) { tile = t >> 3; qpart = (int)(t & 7u); parts = 8u; slab0 = 8u * tile; }
        else if (t < n8 + n4) { const unsigned u = t - n8; tile = S8 + (u >> 2); qpart = (int)(u & 3u); parts = 4u; slab0 = n8 + 4u * (u >> 2); }
        else if (t < n8 + n4 + n2) { const unsigned u = t - n8 - n4; tile = S8 + S + (u >> 1); qpart = (int)(u & 1u); parts = 2u; slab0 = n8 + n4 + 2u * (u >> 1); }
        else tile = t - 7u * S8 - 3u * S - S2;
        if (qpart >= 0) { kt0 = qpart * (nk / (int)parts); kt1 = kt0 + nk / (int)parts; }
        const int half = g.nt_count >> 1, grp = (int)(tile / (unsigned)g.nt_count), j = (int)(tile % (unsigned)g.nt_count);
        tbx = j;
        tby = ((j < half) != (g.order != 0)) ? grp : g.mt_count - 1 - grp;
    } else {
        swizzled_tile(tbx, tby);
    }
    const int m0 = tby * TM, n0 = tbx * TN;

    // ---- A: lane -> (row within a 16-row group, 16-byte chunk of the step's 64-byte row segment); wave w owns rows 32 w .. 32 w + 31
    const int rsub = lane >> 2, ac = lane & 3;
    unsigned aoff[2];           // byte offsets into A of (row, chunk) at k = 0 (rows past M clamped: their products are never stored)
    unsigned apw[2];            // byte offsets into an A buffer (hi plane) of the 8 bytes this lane writes per row
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = w * 32 + j * 16 + rsub;
        int gm = m0 + r; gm = gm < g.M ? gm : g.M - 1;
        aoff[j] = (unsigned)(((size_t)gm * g.lda + ac * 4) * 4);
        apw[j] = (unsigned)(r * 32 + (((ac >> 1) ^ ((r >> 3) & 1)) * 16) + (ac & 1) * 8);
    }
    // ---- W planes by LDS-DMA: piece = 32 rows x 32 B; wave w moves piece w of each plane
    // PACKED operand (g.Wmid == nullptr; pgasr_pack_x6w_planes): W is stored tile by tile in the LDS image's own order,
    // [column tile][16-deep step][plane][8 KB image], so a W piece is 1 KB of consecutive bytes -- 8 full lines per wave instruction
    // instead of 32 quarter lines of a row-major plane (1,024 -> 448 TCP requests per step)
    const bool packed = g.Wmid == nullptr;
    const unsigned short* pw[3];
    {
        const int row = 32 * w + (lane >> 1), cp = lane & 1, c = cp ^ ((row >> 3) & 1);
        const size_t o = (size_t)(n0 + row) * g.K + c * 8;
        pw[0] = g.Whi + o; pw[1] = packed ? g.Whi : g.Wmid + o; pw[2] = g.Wlo + o;
    }
    const unsigned char* wpk = reinterpret_cast<const unsigned char*>(g.Whi) + (size_t)tbx * nk * WSTAGE_BYTES + w * 1024 + lane * 16;
    auto issue_w1 = [&](int kt, int stage, int plane) {     // one of a step's three W pieces
        if (X6D(1) && kt >= kt0 + NW - 1) return;
        const int kc = kt < nk ? kt : nk - 1;
        unsigned char* dst = smem + LDS_W + stage * WSTAGE_BYTES + plane * WP_BYTES + w * 1024;
        if (packed) dma16(wpk + ((size_t)kc * 3 + plane) * WP_BYTES, dst);
        else dma16(pw[plane] + kc * TK, dst);
    };
    auto issue_w = [&](int kt, int stage) { issue_w1(kt, stage, 0); issue_w1(kt, stage, 1); issue_w1(kt, stage, 2); };
    u32x4_t araw[NA][2];
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    auto load_a1 = [&](int kt, auto setc, int j) {      // ONE of a step's two A loads (k clamped: always issued, the counted waits are exact)
        const unsigned kb = (unsigned)((kt < nk ? kt : nk - 1) * TK * 4);
        if (X6D(4) && kt >= kt0 + NA + 1) return;
        if (j == 0) load_a_one<0>(araw[decltype(setc)::value], aoff, kb, g.A);
        else load_a_one<1>(araw[decltype(setc)::value], aoff, kb, g.A);
    };
    auto load_a = [&](int kt, auto setc) { load_a1(kt, setc, 0); load_a1(kt, setc, 1); };
    auto convert_a1 = [&](int buf, auto setc, int j) {    // one of a register set's two rows -> the three planes of buffer `buf`
        constexpr int S = decltype(setc)::value;
        const unsigned base = lds0 + (unsigned)buf * ABUF_BYTES;
        if (X6D(8)) return;
        unsigned h0, m0_, l0, h1, m1_, l1;
        if (X6D(128)) { h0 = araw[S][j].x; h1 = araw[S][j].y; m0_ = araw[S][j].z; m1_ = araw[S][j].w; l0 = h0; l1 = h1; }      // 128: no split arithmetic
        else {
            split3(__uint_as_float(araw[S][j].x), __uint_as_float(araw[S][j].y), h0, m0_, l0);
            split3(__uint_as_float(araw[S][j].z), __uint_as_float(araw[S][j].w), h1, m1_, l1);
        }
        if (X6D(64)) { asm volatile("" :: "v"(h0), "v"(h1), "v"(m0_), "v"(m1_), "v"(l0), "v"(l1)); return; }                     // 64: no LDS stores
        write_planes3(base + apw[j], h0, h1, m0_, m1_, l0, l1);
    };
    auto convert_a = [&](int buf, auto setc) { convert_a1(buf, setc, 0); convert_a1(buf, setc, 1); };

    // ---- fragment read offsets (hi plane; mid / lo = + 8 KB / + 16 KB) ----
    const int fr = lane & 31, fh = lane >> 5;
    unsigned offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wm * 128 + i * 32 + fr;
        offA[i] = (unsigned)(row * 32 + ((fh ^ ((row >> 3) & 1)) * 16));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = wn * 64 + j * 32 + fr;
        offB[j] = (unsigned)(LDS_W + n * 32 + ((fh ^ ((n >> 3) & 1)) * 16));
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    X6STAMP_DECL;

    // one half of a step: row tiles 2 HALF, 2 HALF + 1 against both column tiles = 24 MFMAs; `mem(0)` runs behind the first
    // twelve, `mem(1)` behind the last twelve (the step's five memory instructions are spread over its four twelve-MFMA
    // blocks: a blocked issue behind 384 cycles of queued MFMA costs nothing, see gemm_c256.hip)
    auto half_step = [&](int cur, int wst, auto halfc, Planes2& fb, auto&& mem) {
        constexpr int HALF = decltype(halfc)::value;
        const unsigned ab = lds0 + (unsigned)cur * ABUF_BYTES, wb = lds0 + (unsigned)wst * WSTAGE_BYTES;
        Planes2 fa;
        if (!X6D(16)) {
            if (HALF == 0) read_pair(fb, wb + offB[0], wb + offB[1]);
            read_pair(fa, ab + offA[2 * HALF], ab + offA[2 * HALF + 1]);
        }
        if (HALF == 0) wait_pairs(fa, fb); else wait_pair(fa);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!X6D(2)) mfma6(acc[2 * HALF + i], fa[i], fb);
            __builtin_amdgcn_sched_barrier(0);
            mem(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: A planes of the first step; then, in the steady state's issue order, W(1) A(2) W(2) A(3) behind W(0) A(1) ----
    load_a(kt0, I0{});
    issue_w(kt0, 0);
    wait_a_regs<3>(araw[0]);                      // the A loads (older than the 3 W pieces) are in
    convert_a(0, I0{});
    load_a(kt0 + 1, I1{});
    issue_w(kt0 + 1, 1);
    load_a(kt0 + 2, I2{});
    issue_w(kt0 + 2, 2);
    load_a(kt0 + 3, I0{});
    asm volatile("s_waitcnt vmcnt(12)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");    // W stage 0 landed (my pieces), my plane stores done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // one 16-deep step; SET = the register set that holds the A rows of step kt + 1.  In flight when the step starts, oldest first:
    //   [A(kt+1) 2] [W(kt+1) 3] [A(kt+2) 2] [W(kt+2) 3] [A(kt+3) 2];   the step issues [W(kt+3) 3] in its first half, [A(kt+4) 2] in its second
    typedef std::integral_constant<int, 1> H1;
    auto step = [&](const int kt, auto setc) {
        const int rel = kt - kt0, cur = rel & 1, wst = rel & (NW - 1);
        const int kw = kt + NW - 1, sw = (rel + NW - 1) & (NW - 1);    // the W step issued now, into the stage everybody finished reading before the barrier just passed
        auto mem_w = [&](int s) { if (s == 0) { issue_w1(kw, sw, 0); issue_w1(kw, sw, 1); } else issue_w1(kw, sw, 2); };
        auto mem_a = [&](int s) { load_a1(kt + 1 + NA, setc, s); };
        Planes2 fb;
        __builtin_amdgcn_sched_barrier(0);
        X6STAMP(0);
        if constexpr (VAR == 7) {
            // SELF-INTERLEAVED (round 4, after the stamps of the other structures: while one wave of a SIMD issues back-to-back MFMAs its
            // partner gets about ONE instruction per MFMA -- a 44-VALU conversion beside the partner's 48 MFMAs took 1,880 cycles, with
            // or without s_setprio -- so "one wave converts while the other multiplies" does not overlap, the phases ADD).  Here every
            // wave carries its own non-MFMA work as fillers in its own MFMA gaps, a few instructions behind each MFMA: the fragment
            // reads of the next row tile, the conversion in eight half-splits, the plane stores, the three W pieces, the two A loads.
            // Both waves of a SIMD run the same stream in step; their MFMAs alternate on the pipe.
            const unsigned ab = lds0 + (unsigned)cur * ABUF_BYTES, wb = lds0 + (unsigned)wst * WSTAGE_BYTES;
            const unsigned cb = lds0 + (unsigned)(cur ^ 1) * ABUF_BYTES;
            constexpr int S = decltype(setc)::value;
            u32x4_t f0[3], f1[3];
            unsigned hh[2], mm[2], ll[2];      // the packed planes of the row being converted: [pair]
            float r0[2], r1[2];
            auto split_a = [&](int j, int q) {      // stage A of a pair's 3-way split: hi, residual
                const float x0 = __uint_as_float(q ? araw[S][j].z : araw[S][j].x), x1 = __uint_as_float(q ? araw[S][j].w : araw[S][j].y);
                typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
                typedef __attribute__((ext_vector_type(2))) float f2;
                hh[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){x0, x1}, bf2));
                r0[q] = x0 - __uint_as_float(hh[q] << 16); r1[q] = x1 - __uint_as_float(hh[q] & 0xFFFF0000u);
            };
            auto split_b = [&](int q) {             // stage B: mid, residual, lo
                typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
                typedef __attribute__((ext_vector_type(2))) float f2;
                mm[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){r0[q], r1[q]}, bf2));
                const float s0 = r0[q] - __uint_as_float(mm[q] << 16), s1 = r1[q] - __uint_as_float(mm[q] & 0xFFFF0000u);
                ll[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){s0, s1}, bf2));
            };
            // Row tiles are taken in the order 0 1 2 3 by waves 0-3 and 2 3 0 1 by waves 4-7, and the fillers ride on the FIRST two row
            // slots of waves 0-3 and on the LAST two of waves 4-7: one wave's filler-laden MFMAs run beside its SIMD partner's bare ones.
            auto filler = [&](int kind, int gp, int nextrow, u32x4_t (&fnext)[3]) {    // kind 0 / 1: first / second filler row, 2: bare row
                if (gp == 0 && nextrow >= 0) read_row(fnext, ab + offA[nextrow]);
                if (kind == 0) {
                    if (gp == 1) wait_a_regs<10>(araw[S]);
                    else if (gp == 2) split_a(0, 0);
                    else if (gp == 3) split_b(0);
                    else if (gp == 4) split_a(0, 1);
                    else if (gp == 5) split_b(1);
                    else if (gp == 6) { if (!X6D(8)) write_planes3(cb + apw[0], hh[0], hh[1], mm[0], mm[1], ll[0], ll[1]); }
                    else if (gp == 7) issue_w1(kw, sw, 0);
                    else if (gp == 8) issue_w1(kw, sw, 1);
                    else if (gp == 9) issue_w1(kw, sw, 2);
                } else if (kind == 1) {
                    if (gp == 1) split_a(1, 0);
                    else if (gp == 2) split_b(0);
                    else if (gp == 3) split_a(1, 1);
                    else if (gp == 4) split_b(1);
                    else if (gp == 5) { if (!X6D(8)) write_planes3(cb + apw[1], hh[0], hh[1], mm[0], mm[1], ll[0], ll[1]); }
                    else if (gp == 6) load_a1(kt + 1 + NA, setc, 0);
                    else if (gp == 7) load_a1(kt + 1 + NA, setc, 1);
                }
            };
            auto row_mfma = [&](auto rowc, auto kindc, auto nextc, u32x4_t (&fa)[3], u32x4_t (&fnext)[3]) {
                constexpr int ROW = decltype(rowc)::value, KIND = decltype(kindc)::value, NEXT = decltype(nextc)::value;
                constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    const int pp = m >> 1, j = m & 1;
                    acc[ROW][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[PA[pp]]), __builtin_bit_cast(bf16x8_t, fb[j][PB[pp]]), acc[ROW][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    filler(KIND, m, NEXT, fnext);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            typedef std::integral_constant<int, 2> R2;
            typedef std::integral_constant<int, 3> R3;
            typedef std::integral_constant<int, -1> RN;
            read_pair(fb, wb + offB[0], wb + offB[1]);
            if (w < 4) {
                read_row(f0, ab + offA[0]);
                wait_row_pair(f0, fb);
                X6STAMP(3);
                row_mfma(I0{}, I0{}, I1{}, f0, f1);
                wait_row(f1);
                X6STAMP(1);
                row_mfma(I1{}, I1{}, R2{}, f1, f0);
                wait_row(f0);
                X6STAMP(4);
                row_mfma(R2{}, R2{}, R3{}, f0, f1);
                wait_row(f1);
                X6STAMP(5);
                row_mfma(R3{}, R2{}, RN{}, f1, f0);
                X6STAMP(6);
            } else {
                read_row(f0, ab + offA[2]);
                wait_row_pair(f0, fb);
                X6STAMP(3);
                row_mfma(R2{}, R2{}, R3{}, f0, f1);
                wait_row(f1);
                X6STAMP(1);
                row_mfma(R3{}, R2{}, I0{}, f1, f0);
                wait_row(f0);
                X6STAMP(4);
                row_mfma(I0{}, I0{}, I1{}, f0, f1);
                wait_row(f1);
                X6STAMP(5);
                row_mfma(I1{}, I1{}, RN{}, f1, f0);
                X6STAMP(6);
            }
        } else if (w < 4) {
            wait_a_regs<10>(araw[decltype(setc)::value]);
            X6STAMP(3);
            convert_a(cur ^ 1, setc);
            __builtin_amdgcn_sched_barrier(0);
            X6STAMP(4);
            half_step(cur, wst, I0{}, fb, mem_w);
            X6STAMP(1);
            half_step(cur, wst, H1{}, fb, mem_a);
            X6STAMP(10);
        } else {
            half_step(cur, wst, I0{}, fb, mem_w);
            __builtin_amdgcn_sched_barrier(0);
            X6STAMP(1);
            wait_a_regs<13>(araw[decltype(setc)::value]);     // .. plus the three W pieces just issued
            X6STAMP(3);
            convert_a(cur ^ 1, setc);
            __builtin_amdgcn_sched_barrier(0);
            X6STAMP(4);
            half_step(cur, wst, H1{}, fb, mem_a);
            X6STAMP(10);
        }
        __builtin_amdgcn_sched_barrier(0);
        // my W pieces of step kt + 1 have landed (behind them: A(kt+2) W(kt+2) A(kt+3) W(kt+3) A(kt+4) = 12), my plane stores are done
        X6STAMP(7);
        if (X6D(1 | 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        X6STAMP(8);
        if (!X6D(32)) __builtin_amdgcn_s_barrier();
        X6STAMP(9);
        asm volatile("" ::: "memory");
    };
    for (int kt = kt0; kt < kt1; kt += 3) {        // A(kt + 1) sits in set (kt - kt0 + 1) % 3
        step(kt, I1{});
        if (kt + 1 < kt1) step(kt + 1, I2{});
        if (kt + 2 < kt1) step(kt + 2, I0{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the loads of the steps behind the last one (discarded)
#pragma unroll
    for (int s = 0; s < NA; ++s) asm volatile("" : "+v"(araw[s][0]), "+v"(araw[s][1]));
    X6STAMP(11);
    X6STAMP_FLUSH();

    if (FEED && qpart >= 0) {
        // one part of a split tile: park the accumulators write-through, count the arrival; the LAST of the tile's parts sums them in index
        // order and goes on to the epilogue, the others take their next work item.  Slab layout [32 vectors][512 threads][4 floats]: a wave
        // instruction moves 1 KB of consecutive bytes (round 5: with 4-byte accesses -- 256 B per instruction -- the last arriver of the FIRST
        // tile needed 56 us for its sum and epilogue, tools/dev/r5_feed_timeline.py: parked at 72 us, counted at 128)
        __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(g.slabs, 0, (int)(unsigned)((size_t)g.slab_count * SLAB_FLOATS * 4), 0x00020000);
        const unsigned sbq = (slab0 + (unsigned)qpart) * 32u;
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
#ifdef PGASR_LSTM_DIAG
        if (tid == 0 && diag_t_ < 4u) x6_feed_diag[8 + diag_t_] = (unsigned)wall_clock64();      // .. parked at
#endif
        __syncthreads();
        const unsigned before = *mailbox;
        __syncthreads();
        if (before != parts - 1u) { if (FEED && g.single) return; continue; }      // single: the head launch -- one work item per workgroup
        // total = ((p0 + p1) + p2) + .., whoever arrives last
        for (unsigned qq = 0; qq < parts; ++qq) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const u32x4_t v4 = __builtin_amdgcn_raw_buffer_load_b128(srs, (((slab0 + qq) * 32u + (unsigned)((i * 2 + j) * 4 + r4)) * 512u + (unsigned)tid) * 16u, 0, 16);
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
    if (g.single) return;
#ifdef PGASR_LSTM_DIAG
    {
        const unsigned nh_ = 8u * (unsigned)g.split8_tiles + 4u * (unsigned)g.split_tiles + 2u * (unsigned)g.split2_tiles;
        if (tid == 0 && diag_t_ >= nh_ && diag_t_ < nh_ + 4u) x6_feed_diag[1028 + (diag_t_ - nh_)] = (unsigned)wall_clock64();
    }
    if (tid == 0) {      // when was which row tile counted (100 MHz wall clock; the LAST feed that ran): tools/dev/r5_feed_timeline.py
        const unsigned slot_ = 16u + (unsigned)((tbx < (g.nt_count >> 1) ? 0 : g.mt_count) + tby);
        if (slot_ < 4096u) x6_feed_diag[slot_] = (unsigned)wall_clock64();
    }
#endif
  }
}
}  // namespace x6c


// ------------------------------------------------------------------------------------------------------------------
// Weight gradients dW = dY^T X in the same arithmetic (model.py:39-44 backward: dW_ih = dgates^T x, dW_hh = dgates^T h_prev):
// gemm_c256.hip's t256 kernel on 16-deep steps and three planes per operand.  Per step a thread loads 2 x 16 B of either
// operand (a wave instruction = one 1-KB k-row), two steps ahead (two register sets), splits the 16 values into hi / mid / lo
// and writes the six planes of the k-major image [k][256 + 32 pad] (pitch 576 B); fragments by ds_read_b64_tr_b16; two LDS
// buffers of 54 KB; the two waves of a SIMD half a step out of phase.  Queue mode, time slabs, the gate on a running sweep's
// slab_done words and the sc1 loads of its dgates: as in t256 (same PgasrTn256Args, same requirements).
// ------------------------------------------------------------------------------------------------------------------
namespace t6 {
constexpr int TM = 256, TN = 256, TK = 16, THREADS = 512;
constexpr int PITCH = 288;                        // halfs per k-row of a plane image (576 B)
constexpr int PLANE_HALFS = TK * PITCH;           // 4608 halfs = 9 KB
constexpr int BUF_HALFS = 6 * PLANE_HALFS;        // A hi | mid | lo | B hi | mid | lo = 54 KB
constexpr int LDS_BYTES = 2 * BUF_HALFS * 2;      // 108 KB
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

// PP (PGASR_T6_VAR): 0 = the two waves of a SIMD half a step out of phase (t256's arrangement); 1 = ping-pong -- waves 0-3 multiply
// while waves 4-7 convert and load, a barrier, then the other way round; 2 = the same without the barrier in the middle
template <bool GATED, int PP>
__global__ __launch_bounds__(THREADS) void gemm_t6_kernel(PgasrTn256Args g0, PgasrTn256Args g1) {
    extern __shared__ __attribute__((aligned(128))) unsigned short S[];      // the ONLY LDS object: [buffer][A hi, mid, lo, B hi, mid, lo][k][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    if (g0.queue && g0.xcc_busy) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;
        if (__hip_atomic_load(g0.xcc_busy + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    }
    const unsigned per0 = (unsigned)((g0.N / TN) * (g0.M / TM) * g0.batch), per1 = g1.M > 0 ? (unsigned)((g1.N / TN) * (g1.M / TM) * g1.batch) : 0u;
    const unsigned nitems_all = (per0 + per1) * (unsigned)g0.splitk;
    // the 32x32x16 operand map of a transposing read (see t256)
    const int tro = (8 * (lane >> 5) + ((lane & 15) >> 2)) * PITCH + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    bool gate_dead = false;      // GATED: the sweep's error word has been seen set: it publishes nothing more, later items do not wait
  for (;;) {
    unsigned item;
    bool second = false;
    if (g0.queue) {
        unsigned* mailbox = reinterpret_cast<unsigned*>(S);       // the buffers are idle between two items
        if (tid == 0) *mailbox = __hip_atomic_fetch_add(g0.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        item = (unsigned)__builtin_amdgcn_readfirstlane((int)*mailbox);
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
        // plain launch: whole K-slabs per XCD (see t256)
        const unsigned per = (unsigned)(tx * ty), nslab = (unsigned)(g.batch * g.splitk);
        const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const unsigned zz = xcd + 8u * (j / per);
        item = (nitems % (8u * per) == 0u && nslab % 8u == 0u) ? zz * per + j % per : blockIdx.x;
    }
    if (item >= nitems) return;
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
        const long long ra_ = (long long)(dir ? T - h_hi : h_lo) * g.gate_B - row_off, rb_ = (long long)(dir ? T - h_lo : h_hi) * g.gate_B - row_off;
        k_beg = ra_ < 0 ? 0 : (int)ra_; k_end = rb_ > g.K ? g.K : (int)rb_;
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
                    // a sweep that gave up (its sticky error word) publishes nothing more: do not sit out 3 s per item
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
    const int nk = (k_end - k_beg) / TK;               // >= 1 (pgasr_internal_tn256_ok: no empty slab, slabs are whole steps)
    const int m0 = tby * TM, n0 = tbx * TN;
    // wave w loads k-rows 2 w + j (j = 0, 1) of both operands: one 1-KB row per wave instruction
    const float* Ab = g.A + (size_t)bidx * g.sA + (size_t)(k_beg + 2 * w) * g.lda + m0 + 4 * lane;
    const float* Bb = g.B + (size_t)bidx * g.sB + (size_t)(k_beg + 2 * w) * g.ldb + n0 + 4 * lane;
    f32x4_t ra[2][2], rb[2][2];                         // [register set][k-row]
    __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + (size_t)bidx * g.sA + (size_t)k_beg * g.lda + m0), 0,
                                                                   GATED ? (int)((size_t)nk * TK * g.lda * 4) : 0, 0x00020000);
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    auto load = [&](int kt, auto setc) {
        constexpr int s = decltype(setc)::value;
        const int kc = kt < nk ? kt : nk - 1;          // past the slab: reload the last step (never converted)
        const float* pa = Ab + (size_t)kc * TK * g.lda;
        const float* pb = Bb + (size_t)kc * TK * g.ldb;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (GATED) {
                const u32x4_t u = __builtin_amdgcn_raw_buffer_load_b128(ars, (unsigned)(((size_t)(kc * TK + 2 * w + j) * g.lda + 4 * lane) * 4), 0, 16);
                ra[s][j] = __builtin_bit_cast(f32x4_t, u);
            } else {
                ra[s][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(pa + (size_t)j * g.lda));
            }
            rb[s][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(pb + (size_t)j * g.ldb));
        }
    };
    auto convert = [&](int buf, auto setc) {
        constexpr int s = decltype(setc)::value;
        unsigned short* base = S + buf * BUF_HALFS + (2 * w) * PITCH + 4 * lane;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned h0, m0_, l0, h1, m1_, l1;
            split3(ra[s][j].x, ra[s][j].y, h0, m0_, l0); split3(ra[s][j].z, ra[s][j].w, h1, m1_, l1);
            *reinterpret_cast<u32x2_t*>(base + j * PITCH) = (u32x2_t){h0, h1};
            *reinterpret_cast<u32x2_t*>(base + PLANE_HALFS + j * PITCH) = (u32x2_t){m0_, m1_};
            *reinterpret_cast<u32x2_t*>(base + 2 * PLANE_HALFS + j * PITCH) = (u32x2_t){l0, l1};
            split3(rb[s][j].x, rb[s][j].y, h0, m0_, l0); split3(rb[s][j].z, rb[s][j].w, h1, m1_, l1);
            *reinterpret_cast<u32x2_t*>(base + 3 * PLANE_HALFS + j * PITCH) = (u32x2_t){h0, h1};
            *reinterpret_cast<u32x2_t*>(base + 4 * PLANE_HALFS + j * PITCH) = (u32x2_t){m0_, m1_};
            *reinterpret_cast<u32x2_t*>(base + 5 * PLANE_HALFS + j * PITCH) = (u32x2_t){l0, l1};
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one half of a step: row tiles 2 HALF, 2 HALF + 1 against both column tiles = 24 MFMAs (hh + hm + mh + hl + lh + mm)
    auto multiply = [&](int buf, auto halfc, bf16x8_t (&bf)[2][3]) {
        constexpr int HALF = decltype(halfc)::value;
        const unsigned short* img = S + buf * BUF_HALFS + tro;
        if (HALF == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[j][p] = tr_frag(img + (3 + p) * PLANE_HALFS + wn * 64 + j * 32);
        }
        bf16x8_t af[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[i][p] = tr_frag(img + p * PLANE_HALFS + wm * 128 + (2 * HALF + i) * 32);
        constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[2 * HALF + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[p]], bf[j][PB[p]], acc[2 * HALF + i][j], 0, 0, 0);
    };
    // a raw barrier: __syncthreads() would also wait (vmcnt) for the loads of the steps ahead that are in flight
#define T6_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

    load(0, I0{});
    convert(0, I0{});
    load(1, I1{});
    load(2, I0{});
    T6_BARRIER();
    const bool mem = !(g.diag & 1), mul = !(g.diag & 2);
    auto step = [&](int kt, auto setc) {               // setc: the register set that holds step kt + 1
        const int cur = kt & 1;
        bf16x8_t bf[2][3];
        if constexpr (PP >= 1) {
            if (w < 4) { if (mul) { multiply(cur, I0{}, bf); multiply(cur, I1{}, bf); } }
            else if (mem) { if (kt + 1 < nk) convert(cur ^ 1, setc); load(kt + 3, setc); }
            if (PP == 1) T6_BARRIER();
            if (w >= 4) { if (mul) { multiply(cur, I0{}, bf); multiply(cur, I1{}, bf); } }
            else if (mem) { if (kt + 1 < nk) convert(cur ^ 1, setc); load(kt + 3, setc); }
        } else if (w < 4) {
            if (mem) { if (kt + 1 < nk) convert(cur ^ 1, setc); load(kt + 3, setc); }
            if (mul) { multiply(cur, I0{}, bf); multiply(cur, I1{}, bf); }
        } else {
            if (mul) multiply(cur, I0{}, bf);
            if (mem) { if (kt + 1 < nk) convert(cur ^ 1, setc); load(kt + 3, setc); }
            if (mul) multiply(cur, I1{}, bf);
        }
        T6_BARRIER();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, I1{});
        if (kt + 1 < nk) step(kt + 1, I0{});
    }
#undef T6_BARRIER

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
}  // namespace t6

// fp32 (rows x cols, leading dim ld) -> dense bf16 hi / mid / lo planes; transpose: planes are (cols x rows)
__global__ __launch_bounds__(256) void split_planes3_kernel(const float* __restrict__ src, int rows, int cols, int ld,
                                                            int transpose, unsigned short* __restrict__ hi,
                                                            unsigned short* __restrict__ mid, unsigned short* __restrict__ lo) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)rows * cols) return;
    int r, c;
    if (transpose) { c = (int)(idx / rows); r = (int)(idx % rows); }     // output index = c*rows + r
    else { r = (int)(idx / cols); c = (int)(idx % cols); }
    unsigned h, m, l;
    split3(src[(size_t)r * ld + c], 0.f, h, m, l);
    hi[idx] = (unsigned short)h;
    mid[idx] = (unsigned short)m;
    lo[idx] = (unsigned short)l;
}

static int x6_var() {       // PGASR_X6_VAR (read at every call: A/B inside one process); default: see the kernel's header
    const char* e = getenv("PGASR_X6_VAR");
    return (e && e[0] == '0') ? 0 : 7;
}
// fp32 W (N x K; or stored K x N with transpose) -> the packed three-plane operand of x6c: [N/256][K/16][plane 3][row 256][2 chunks of 8 bf16],
// the chunk at position cp of row r holding k-chunk cp ^ ((r >> 3) & 1) -- the LDS image itself.  One thread per 16-byte chunk position.
__global__ __launch_bounds__(256) void pack_x6w_kernel(const float* __restrict__ src, int N, int K, int ld, int transpose, u32x4_t* __restrict__ dst) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over N * K / 8 chunk positions
    if (idx >= (size_t)N * K / 8) return;
    const int nk = K / 16;
    const int pos = (int)(idx % 512), kt = (int)((idx / 512) % nk), tn = (int)(idx / 512 / nk);
    const int r = pos >> 1, cp = pos & 1, c = cp ^ ((r >> 3) & 1);
    const int n = tn * 256 + r, k0 = kt * 16 + c * 8;
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = transpose ? src[(size_t)(k0 + 2 * e) * ld + n] : src[(size_t)n * ld + k0 + 2 * e];
        const float x1 = transpose ? src[(size_t)(k0 + 2 * e + 1) * ld + n] : src[(size_t)n * ld + k0 + 2 * e + 1];
        split3(x0, x1, h[e], m[e], l[e]);
    }
    u32x4_t* blk = dst + ((size_t)tn * nk + kt) * 3 * 512 + pos;
    blk[0] = (u32x4_t){h[0], h[1], h[2], h[3]};
    blk[512] = (u32x4_t){m[0], m[1], m[2], m[3]};
    blk[1024] = (u32x4_t){l[0], l[1], l[2], l[3]};
}

// Workgroups of a feed that land on the sweep's own XCDs take K-split head items before they leave (PGASR_X6_HEAD_HELP=0: they leave at once).
// Round 5, tools/dev/r5_head_help.sh, A/B/A/B on one box: fed backward sweeps 1.64-1.66 -> 1.61-1.63 ms, f32 step 10.05-10.10 -> 10.02 ms.  Small,
// because the stall only moves: the tiles behind the head still wait for the free XCDs' CUs (NOTES 0.2).
constexpr int X6_HEAD_HELP_DEFAULT = 1;
// Tile groups (of nt tiles, in the order the sweep takes them) in K-eighths, -quarters and -halves at the head of a feed.  Round 5, one box,
// f32 step (tools/dev/r5_graded_head.sh: eighths, quarters, halves -> ms): 0,16,0 9.78 / 9.79; 0,16,16 9.745; 0,16,32 9.78; 2,14,16 9.82; 2,14,32 9.76;
// 4,12,32 9.78.  The halves take the sweep's second stall away (group 16 used to be ready ~150 us after the sweep wanted it) and the 16-byte slab
// accesses 43 us of its first -- and the fed sweeps gain only 0.02 ms of the 0.09, because what sets their pace inside the step is the CLOCK: 2.40 GHz
// with no GEMM beside them, 2.2-2.3 with the weight-gradient and feed GEMMs on the other XCDs, 2.14-2.2 when the head keeps more of them busy
// (in-kernel shader clock against the 100 MHz wall clock, tools/dev/r5_feed_timeline.py; 3,220 cycles per step in every case).
constexpr int X6_SPLIT8_GROUPS_DEFAULT = 0, X6_SPLIT4_GROUPS_DEFAULT = 16, X6_SPLIT2_GROUPS_DEFAULT = 16;
constexpr int X6_FWD_SPLIT4_GROUPS_DEFAULT = 2;      // r5_fwd_quarters2.sh, one box: 0 groups 9.72-9.74 ms, 2: 9.67, 4: 9.68-9.70, 8: 9.73, 16: 9.71
constexpr int X6_FEED_SPLIT_MAX = 128;     // split tiles per feed at most: 4 x 128 slabs of 256 KB = 128 MB of workspace (arrival counters: words 64..191 of the head)
// Time-ordered tile groups (of nt tiles) at the head of a feed whose tiles are split into K-quarters (PGASR_X6_SPLIT_GROUPS, read at every
// call; the fed and the sequential order read the same value, so they keep giving the same bits).
int x6_split_groups() {
    const char* e = getenv("PGASR_X6_SPLIT_GROUPS");
    const int v = e ? atoi(e) : X6_SPLIT4_GROUPS_DEFAULT;
    return v < 0 ? 0 : v;
}
int x6_split8_groups() {      // tile groups in K-eighths in front of the quarters
    const char* e = getenv("PGASR_X6_SPLIT8_GROUPS");
    const int v = e ? atoi(e) : X6_SPLIT8_GROUPS_DEFAULT;
    return v < 0 ? 0 : v;
}
int x6_fwd_split_groups() {   // K < 1024 (the forward projections): tile groups in quarters; no eighths, no halves
    const char* e = getenv("PGASR_X6_FWD_SPLIT_GROUPS");
    const int v = e ? atoi(e) : X6_FWD_SPLIT4_GROUPS_DEFAULT;
    return v < 0 ? 0 : v;
}
int x6_split2_groups() {      // tile groups in K-halves behind the quarters
    const char* e = getenv("PGASR_X6_SPLIT2_GROUPS");
    const int v = e ? atoi(e) : X6_SPLIT2_GROUPS_DEFAULT;
    return v < 0 ? 0 : v;
}
// K in quarters for the first tiles of a feed.  K >= 1024 (the input-gradient feeds): since round 4.  K = 512 (the forward projections): measured
// SLOWER in round 4 (forward sweeps 1.54 against 1.47 ms) and again early in round 5 -- with 4-byte accesses to the parked accumulators, which
// cost the last arriver of a tile 56 us.  With 16-byte accesses (tools/dev/r5_fwd_quarters2.sh, one box): forward sweeps 1.45-1.49 / 1.43-1.44 /
// 1.41-1.42 -> 1.40-1.42 / 1.38-1.40 / 1.36-1.38 ms with two or four groups in quarters.  A forward projection's first groups are therefore sums
// of four K-quarters whenever the FEED kernel computes them, and the sequential order calls the feed kernel too (functional.py), so that both
// orders keep giving the same bits.  PGASR_X6_QUARTER_K: A/B only.
int x6_quarters(int K) {
    const char* e = getenv("PGASR_X6_QUARTER_K");
    const int kmin = e ? atoi(e) : 512;
    return (K >= kmin && K >= 16 * x6c::TK && K % (4 * x6c::TK) == 0) ? 4 : 1;
}

}  // namespace

#ifdef PGASR_X6_DIAG
extern "C" int pgasr_diag_x6_stamps(long long* out) {      // host copy of workgroup 0's stamps: [wave 8][16]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(x6_stamp_out), sizeof(long long) * 8 * 16) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int pgasr_pack_x6w_planes(const float* src, int rows, int cols, int ld, int transpose, void* pack, void* stream) {
    if (!src || !pack || rows <= 0 || cols <= 0 || ld < cols) return PGASR_ERR_INVALID_ARG;
    const int N = transpose ? cols : rows, K = transpose ? rows : cols;
    if ((N % x6c::TN) || (K % x6c::TK) || (((size_t)pack) & 15)) return PGASR_ERR_UNSUPPORTED;
    const size_t chunks = (size_t)N * K / 8;
    PGASR_LAUNCH_KERNEL(pack_x6w_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, N, K, ld, transpose, (u32x4_t*)pack);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_split_bf16_planes3(const float* src, int rows, int cols, int ld, int transpose,
                                        unsigned short* hi, unsigned short* mid, unsigned short* lo, void* stream) {
    if (!src || !hi || !mid || !lo || rows <= 0 || cols <= 0 || ld < cols) return PGASR_ERR_INVALID_ARG;
    const size_t total = (size_t)rows * cols;
    PGASR_LAUNCH_KERNEL(split_planes3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                        src, rows, cols, ld, transpose, hi, mid, lo);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

static bool x6w_shape_ok(int M, int N, int K, const float* A, int lda, int ldc, const void* p0, const void* p1, const void* p2) {
    if ((K % x6c::TK) || K < 4 * x6c::TK || (N % x6c::TN) || (lda & 3)) return false;
    if ((((size_t)A) & 15) || (((size_t)p0) & 15) || (((size_t)p1) & 15) || (((size_t)p2) & 15)) return false;
    // buffer / 32-bit-offset addressing.  The bound is on the PADDED row count: the epilogue (and the dact_y loads) address the rows of
    // whole 256-row tiles with 32-bit offsets and rely on the buffer range to drop rows >= M -- an offset that wraps past 2^32 would
    // land back inside the range and overwrite (read) the first rows instead of being dropped
    const size_t Mp = ((size_t)M + x6c::TM - 1) / x6c::TM * x6c::TM;
    if (Mp * ldc * 4 >= ((size_t)1 << 32) || Mp * lda * 4 >= ((size_t)1 << 32)) return false;
    return (M + x6c::TM - 1) / x6c::TM <= 65535;
}

extern "C" int pgasr_gemm_x6w_f32(int M, int N, int K, const float* A, int lda, const unsigned short* Whi,
                                  const unsigned short* Wmid, const unsigned short* Wlo, float* C, int ldc,
                                  const float* bias, const float* dact_y, float slope, void* stream) {
    if (!A || !Whi || (!Wmid != !Wlo) || !C || M <= 0 || N <= 0 || K <= 0 || lda < K || ldc < N) return PGASR_ERR_INVALID_ARG;     // Wmid == Wlo == NULL: Whi is a pack
    if (!x6w_shape_ok(M, N, K, A, lda, ldc, Whi, Wmid, Wlo)) return PGASR_ERR_UNSUPPORTED;
    const size_t lds = (size_t)x6c::LDS_BYTES;
    const int var = x6_var();
    auto kern = var == 7 ? x6c::gemm_x6c_kernel<false, 7> : x6c::gemm_x6c_kernel<false, 0>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    DmaGemmArgs g{A, Whi, Wlo, C, M, N, K, lda, ldc, bias, dact_y, slope, nullptr, nullptr, nullptr, 0, 0, 0, 1, 0, nullptr, nullptr, 0, Wmid};
#ifdef PGASR_X6_DIAG
    if (const char* e = getenv("PGASR_X6_DIAG")) g.single = atoi(e);
#endif
    PGASR_LAUNCH_KERNEL(kern, dim3((unsigned)(N / x6c::TN), (unsigned)((M + x6c::TM - 1) / x6c::TM)), dim3(x6c::THREADS), lds,
                        (hipStream_t)stream, g);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" size_t pgasr_gemm_x6w_feed_workspace_bytes(void) { return 1024 + (size_t)X6_FEED_SPLIT_MAX * 4 * x6c::SLAB_FLOATS * 4; }

// Column tiles per direction half that a six-product feed of an N-column product counts in tiles_done (the consumer's fed_need)
extern "C" int pgasr_gemm_x6w_feed_col_tiles(int N) { return (N > 0 && N % (2 * x6c::TN) == 0) ? N / (2 * x6c::TN) : 0; }

namespace {
// tiles of a feed's graded head (in queue order): eighths, quarters, halves -- a function of the product's shape, the knobs and the workspace only
void x6_feed_splits(int mt, int nt, int K, size_t workspace_bytes, int& split8, int& split, int& split2) {
    split8 = split = split2 = 0;
    if (x6_quarters(K) != 4) return;
    // THE GRADED HEAD (round 5, tools/dev/r5_feed_timeline.py): a whole tile of the input-gradient product (K = 2048) is 260-275 us on one CU and
    // the sweep takes a tile group every ~11 us, so what the sweep waits for is not only its first rows: with sixteen groups in quarters and whole
    // tiles behind them, group 16 was ready at ~340 us where the sweep wanted it at ~190 -- its second stall (20-40 us per cluster after the
    // 120 us at step 0).  Groups in eighths first (PGASR_X6_SPLIT8_GROUPS), then quarters (PGASR_X6_SPLIT_GROUPS), then halves
    // (PGASR_X6_SPLIT2_GROUPS), then whole tiles; every tile is the fixed-order sum of its parts, its position decides how many: the fed and the
    // sequential order run the same decomposition and give the same bits.
    const size_t room = (workspace_bytes - 1024) / ((size_t)x6c::SLAB_FLOATS * 4);          // parked accumulator sets
    const int total = mt * nt;
    if (K >= 1024) { split8 = x6_split8_groups() * nt; split = x6_split_groups() * nt; split2 = x6_split2_groups() * nt; }
    else split = x6_fwd_split_groups() * nt;
    if (K % (8 * x6c::TK) || K < 32 * x6c::TK) { split += split8; split8 = 0; }
    if (split8 > total) split8 = total;
    if (split8 + split > total) split = total - split8;
    if (split8 + split + split2 > total) split2 = total - split8 - split;
    while (split8 + split + split2 > X6_FEED_SPLIT_MAX || (size_t)(8 * split8 + 4 * split + 2 * split2) > room) {     // arrival words; slabs
        if (split2 > 0) --split2; else if (split > 0) --split; else --split8;
    }
}
}  // namespace

// Work items of a HEAD launch (phase 1 of pgasr_gemm_x6w_feed_phase_f32): the eighth and quarter items of the feed's graded head; 0: none
// (the shape has no split head, or more items than one workgroup per CU)
extern "C" int pgasr_gemm_x6w_feed_head_items(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || pgasr_gemm_x6w_feed_col_tiles(N) == 0 || (K % x6c::TK)) return 0;
    int s8, s4, s2;
    x6_feed_splits((M + x6c::TM - 1) / x6c::TM, N / x6c::TN, K, pgasr_gemm_x6w_feed_workspace_bytes(), s8, s4, s2);
    const int items = 8 * s8 + 4 * s4;
    return items <= 256 ? items : 0;
}

// phase 0: the whole feed (memset of the queue head, masked pass, unmasked pass).  phase 1: the HEAD -- queue head zeroed, one workgroup per
// eighth / quarter item, no XCD mask, every workgroup leaves after its item.  phase 2: the rest -- no memset, both passes continue the queue.
// The head needs no registration of the consuming sweep's XCDs, so it can be issued right behind the PREVIOUS sweep (the single launch starts
// ~40 us after that sweep's end: event -> prepare -> sweep -> registration -> gate -> memset -> launch).  Round 5 wired it into the step twice --
// on the feeding stream, and on a stream of its own with caller-zeroed queue words -- and measured nothing to gain (hipops.py, NOTES 0.46);
// the step uses phase 0.
extern "C" int pgasr_gemm_x6w_feed_phase_f32(int M, int N, int K, const float* A, int lda, const unsigned short* Whi,
                                             const unsigned short* Wmid, const unsigned short* Wlo, float* C, int ldc, const float* bias,
                                             const unsigned* xcc_busy, unsigned* tiles_done, int order, int phase, unsigned* ctrl,
                                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !Whi || (!Wmid != !Wlo) || !C || !tiles_done || M <= 0 || N <= 0 || K <= 0 || lda < K || ldc < N) return PGASR_ERR_INVALID_ARG;
    if (order < 0 || order > 1 || phase < 0 || phase > 2) return PGASR_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < 1024) return PGASR_ERR_WORKSPACE;
    if (!x6w_shape_ok(M, N, K, A, lda, ldc, Whi, Wmid, Wlo) || pgasr_gemm_x6w_feed_col_tiles(N) == 0) return PGASR_ERR_UNSUPPORTED;
    if ((size_t)M * ldc * 4 >= ((size_t)1 << 31)) return PGASR_ERR_UNSUPPORTED;
    // a phased feed sizes its head for the FULL workspace (pgasr_gemm_x6w_feed_head_items knows no other)
    if (phase != 0 && workspace_bytes < pgasr_gemm_x6w_feed_workspace_bytes()) return PGASR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int mt = (M + x6c::TM - 1) / x6c::TM, nt = N / x6c::TN;
    const size_t lds = (size_t)x6c::LDS_BYTES + 16;
    const int var = x6_var();
    auto kern = var == 7 ? x6c::gemm_x6c_kernel<true, 7> : x6c::gemm_x6c_kernel<true, 0>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    const int quarters = x6_quarters(K);
    int split8, split, split2;
    x6_feed_splits(mt, nt, K, phase != 0 ? pgasr_gemm_x6w_feed_workspace_bytes() : workspace_bytes, split8, split, split2);
    const int head_items = 8 * split8 + 4 * split;
    if (phase != 0 && (head_items <= 0 || head_items > 256)) return PGASR_ERR_UNSUPPORTED;
    // ctrl != NULL: 256 words the CALLER has zeroed for this feed (tile counter at 0, arrival counters from word 64) -- no memset in any phase,
    // so the head and the rest may be issued on two streams and draw from the one queue side by side
    if (!ctrl && phase != 2 && hipMemsetAsync(workspace, 0, 1024, st) != hipSuccess) return PGASR_ERR_LAUNCH;     // tile counter + arrival counters
    unsigned* cw = ctrl ? ctrl : (unsigned*)workspace;
    DmaGemmArgs g{A, Whi, Wlo, C, M, N, K, lda, ldc, bias, nullptr, 0.f, cw, xcc_busy, tiles_done, mt, nt, order,
                  quarters, split, (float*)((char*)workspace + 1024), cw + 64, 0, Wmid};
    g.split8_tiles = split8; g.split2_tiles = split2; g.slab_count = 8 * split8 + 4 * split + 2 * split2;
    if (phase == 1) {
        g.single = 1; g.xcc_busy = nullptr; g.head_help = 0;
        PGASR_LAUNCH_KERNEL(kern, dim3((unsigned)head_items), dim3(x6c::THREADS), lds, st, g);
        PGASR_CHECK_LAUNCH();
        return PGASR_OK;
    }
    {   // PGASR_X6_HEAD_HELP (read at every call; a speed hint, the same bits either way): workgroups on the sweep's XCDs take K-split head items
        const char* eh = getenv("PGASR_X6_HEAD_HELP");
        g.head_help = (eh ? atoi(eh) : X6_HEAD_HELP_DEFAULT) != 0 && split8 + split > 0 && (phase == 0 || ctrl != nullptr);
    }
    // PGASR_X6_FWD_FEED_GRID (A/B only): persistent workgroups of the masked pass of a FORWARD feed (K < 1024).  The forward phase has CUs to
    // spare, and a sweep runs at the clock the GEMMs beside it leave (NOTES 0.46): does a thinner, longer feed cost the sweep less?  No
    // (tools/dev/r5_fwd_feed_grid.sh, one box, forward sweeps of the f32 step): 256 workgroups 4.20-4.23 ms, 192: 4.23, 128: 4.20, 96: 4.18,
    // 64: 5.94 (the sweep waits for its rows) -- the same GEMM energy beside the sweep costs it the same time, spread or not.
    int grid0 = 256;
    if (K < 1024 && xcc_busy) { const char* eg = getenv("PGASR_X6_FWD_FEED_GRID"); if (eg && atoi(eg) >= 16 && atoi(eg) <= 256) grid0 = atoi(eg); }
    for (int pass = 0; pass < 2; ++pass) {     // one persistent workgroup per CU; pass 1 ignores the busy counters
        if (pass == 1) g.xcc_busy = nullptr;
        PGASR_LAUNCH_KERNEL(kern, dim3(pass == 0 ? grid0 : 256), dim3(x6c::THREADS), lds, st, g);
        PGASR_CHECK_LAUNCH();
    }
    return PGASR_OK;
}

extern "C" int pgasr_gemm_x6w_feed_f32(int M, int N, int K, const float* A, int lda, const unsigned short* Whi,
                                       const unsigned short* Wmid, const unsigned short* Wlo, float* C, int ldc, const float* bias,
                                       const unsigned* xcc_busy, unsigned* tiles_done, int order,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    return pgasr_gemm_x6w_feed_phase_f32(M, N, K, A, lda, Whi, Wmid, Wlo, C, ldc, bias, xcc_busy, tiles_done, order, 0, nullptr, workspace, workspace_bytes, stream);
}

// ---- internal: the six-product TN kernel behind pgasr_lstm_wgrads_streamed(planes = 3) (same contract as pgasr_internal_tn256_launch) ----
int pgasr_internal_tn6_launch(PgasrTn256Args a, int masked_then_unmasked, hipStream_t st, const PgasrTn256Args* second) {
    PgasrTn256Args b{};
    // time slabs of a sweep over B % 16 == 0 utterances are whole 16-row steps of this kernel (dW_hh's shortened slab included);
    // every other K partition keeps the 32-row rule of the shared dispatch
    const int tk = a.tslabs ? t6::TK : 32;
    if (!pgasr_internal_tn256_ok(a, tk)) return PGASR_ERR_UNSUPPORTED;
    if (second) {
        if (!a.queue || second->splitk != a.splitk || !pgasr_internal_tn256_ok(*second, tk)) return PGASR_ERR_UNSUPPORTED;
        b = *second;
    }
    const size_t lds = (size_t)t6::LDS_BYTES;
    const char* ev = getenv("PGASR_T6_VAR");
    const int pp = (ev && ev[0] >= '0' && ev[0] <= '2') ? ev[0] - '0' : 0;
    auto kplain = pp == 2 ? t6::gemm_t6_kernel<false, 2> : pp == 1 ? t6::gemm_t6_kernel<false, 1> : t6::gemm_t6_kernel<false, 0>;
    auto kgated = pp == 2 ? t6::gemm_t6_kernel<true, 2> : pp == 1 ? t6::gemm_t6_kernel<true, 1> : t6::gemm_t6_kernel<true, 0>;
    if (hipFuncSetAttribute((const void*)kplain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    const unsigned nitems = (unsigned)((a.N / t6::TN) * (a.M / t6::TM)) * (unsigned)(a.batch * a.splitk);
    if (!a.queue) {
        PGASR_LAUNCH_KERNEL(kplain, dim3(nitems), dim3(t6::THREADS), lds, st, a, b);
        PGASR_CHECK_LAUNCH();
        return PGASR_OK;
    }
    if (a.gate && hipFuncSetAttribute((const void*)kgated, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    const unsigned* busy = a.xcc_busy;
    // persistent workgroups of the MASKED pass (PGASR_T6_GRID, A/B only; default one per CU = 256, of which those on the sweep's XCDs leave at once)
    const char* eg = getenv("PGASR_T6_GRID");
    const int grid0 = (eg && atoi(eg) >= 8 && atoi(eg) <= 256) ? atoi(eg) : 256;
    for (int pass = 0; pass < ((masked_then_unmasked && busy) ? 2 : 1); ++pass) {
        a.xcc_busy = (pass == 0) ? busy : nullptr;
        b.xcc_busy = a.xcc_busy;
        const int grid = (pass == 0 && busy) ? grid0 : 256;
        if (a.gate) PGASR_LAUNCH_KERNEL(kgated, dim3(grid), dim3(t6::THREADS), lds, st, a, b);
        else        PGASR_LAUNCH_KERNEL(kplain, dim3(grid), dim3(t6::THREADS), lds, st, a, b);
        PGASR_CHECK_LAUNCH();
    }
    return PGASR_OK;
}

#ifdef PGASR_LSTM_DIAG
extern "C" int pgasr_diag_x6_feed(unsigned* out, int reset) {      // host copy of x6_feed_diag
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(x6_feed_diag), sizeof(unsigned) * 4096) != hipSuccess) return 1;
    if (reset) { static unsigned z[4096]; if (hipMemcpyToSymbol(HIP_SYMBOL(x6_feed_diag), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
