// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, bit-equal to an
// fmaf chain), used by every dense contraction of the path: framewise affine (model.py:38,50),
// LSTM input projections (model.py:39-44), vocabulary head and all their weight/activation
// gradients.
//
// Tile 128x128x16 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 MFMA tiles of
// 32x32, 64 accumulator registers), operands staged global -> registers -> LDS k-major
// ([k][m] / [k][n], row pad 4 floats) so a wave's MFMA operand read is two conflict-free rows of
// 32 consecutive floats; next tile's global loads are issued before the current tile's MFMAs and
// written to the other LDS buffer after them (one barrier per k-tile).
//
// Beyond C = alpha*op(A)*op(B): strided batches; split-K or batch-summed partial slabs reduced
// by a second deterministic kernel; bias / leaky-ReLU epilogue; d(leaky-ReLU) epilogue mask;
// per-batch (x - shift)*scale transform on operand load (the per-utterance instance norm of
// model.py:48 fused into the affine's tile load).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, PADF = 4;
constexpr int GEMM_THREADS = 256;

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    long long sA, sB, sC;      // batch strides (elements)
    int batch, splitk, kper;   // z = batch_index * splitk + split_index ; kper multiple of BK
    float alpha;
    const float* bias; const float* bias2;
    int act; float slope;      // act 1: leaky relu(slope)
    int accumulate;            // C += result
    const float* dact_y;       // if set: result *= (dact_y[m*ldc+n] > 0 ? 1 : slope)
    int norm_operand;          // 0 none, 1: A = (A - shift[b])*scale[b], 2: same on B
    const float* shift; const float* scale;
    float* partial;            // if set: raw alpha*acc slabs [z][M][N]; epilogue done by reduce
    unsigned* queue;           // queue mode (else NULL): global tile counter, zeroed before the launch
    const unsigned* xcc_busy;  // queue mode: [8] words; a workgroup running on XCD i (HW_REG_XCC_ID) with busy[i] != 0 takes no tile
    int tx, ty, tz;            // queue mode: tile counts
    int wide;                  // some byte offset of the output needs more than 32 bits: the epilogue keeps 64-bit addressing
};

// Epilogue of both tile kernels (32x32 accumulator layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).
// Branch-free per element: every store and load goes through a buffer resource whose range check drops what lies past
// the output (rows >= M; columns >= N and dead workgroups get the out-of-range offset 0xFFFFFFFF), and the optional
// operands of a 32 x 32 tile (leaky' mask, accumulate target) are loaded as one batch in front of its 16 stores.
// Rounds 1-2 had `if (m >= M) continue` and conditional loads per element; hipcc then puts s_waitcnt vmcnt(0) in front
// of EVERY store, i.e. each store waits for the previous one's acknowledgement (found in round 3 with the k-loop of an
// LDS-DMA kernel switched off: the epilogue of the input projection took 160 of 358 us).  Needs every byte offset
// below 2^32 (checked on the host: GemmArgs::wide == 0); otherwise the old per-element form runs.
__device__ __forceinline__ void store_tiles_fast(const GemmArgs& g, const f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn,
                                                 int lane, int z, int bidx, bool live) {
    const int cl = lane & 31, rq = lane >> 5;
    if (g.partial) {
        float* slab = g.partial + (size_t)z * g.M * g.N;
        __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)(unsigned)((size_t)g.M * g.N * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + cl;
                const bool ok = live && n < g.N;
                const unsigned o0 = (unsigned)(((size_t)(m0 + wm * 64 + i * 32 + 4 * rq) * g.N + n) * 4);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(g.alpha * acc[i][j][r]), prs,
                                                          ok ? o0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.N * 4) : 0xFFFFFFFFu, 0, 0);
            }
        return;
    }
    float* Cb = g.C + (size_t)bidx * g.sC;
    const unsigned cbytes = (unsigned)(((size_t)(g.M - 1) * g.ldc + g.N) * 4);
    __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(Cb, 0, (int)cbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dact_y ? g.dact_y + (size_t)bidx * g.sC : Cb), 0, (int)cbytes, 0x00020000);
    float bsum[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + cl;
        bsum[j] = 0.f;
        if (n < g.N) {
            if (g.bias) bsum[j] += g.bias[n];
            if (g.bias2) bsum[j] += g.bias2[n];
        }
    }
    const bool lk = g.act == 1;
    if (!g.dact_y && !g.accumulate) {             // straight-line stores
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + cl;
                const bool ok = live && n < g.N;
                const unsigned o0 = (unsigned)(((size_t)(m0 + wm * 64 + i * 32 + 4 * rq) * g.ldc + n) * 4);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = g.alpha * acc[i][j][r] + bsum[j];
                    v = (lk && !(v > 0.f)) ? v * g.slope : v;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), crs, ok ? o0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldc * 4) : 0xFFFFFFFFu, 0, 0);
                }
            }
        return;
    }
    const bool hd = g.dact_y != nullptr, ha = g.accumulate != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + cl;
            const bool ok = live && n < g.N;
            const unsigned o0 = (unsigned)(((size_t)(m0 + wm * 64 + i * 32 + 4 * rq) * g.ldc + n) * 4);
            float f[16], c0[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {       // out-of-range loads return 0 (harmless: their stores are dropped as well)
                const unsigned o = ok ? o0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldc * 4) : 0xFFFFFFFFu;
                f[r] = hd ? (__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, o, 0, 0)) > 0.f ? 1.f : g.slope) : 1.f;
                c0[r] = ha ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crs, o, 0, 0)) : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned o = ok ? o0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldc * 4) : 0xFFFFFFFFu;
                float v = g.alpha * acc[i][j][r] + bsum[j];
                v = (lk && !(v > 0.f)) ? v * g.slope : v;
                v = v * f[r] + c0[r];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), crs, o, 0, 0);
            }
        }
}

// Load an 8-float strip of a tile operand.  KCONTIG: the operand is stored with k contiguous
// (row-major MxK A, or NxK "B^T"); otherwise the M/N index is contiguous.
//   KCONTIG : thread -> (mn = tid>>1, k0 = (tid&1)*8), strip runs along k
//   !KCONTIG: thread -> (k = tid>>4, mn0 = (tid&15)*8), strip runs along mn
template <bool KCONTIG>
__device__ __forceinline__ void load_strip(const float* __restrict__ P, int ld, int mn_base, int mn_lim,
                                           int k_base, int k_lim, int tid, bool vec_ok, float v[8]) {
    if (KCONTIG) {
        const int mn = mn_base + (tid >> 1);
        const int k0 = k_base + (tid & 1) * 8;
        if (mn < mn_lim && k0 + 8 <= k_lim && vec_ok) {
            const float4* p = reinterpret_cast<const float4*>(P + (size_t)mn * ld + k0);
            const float4 a = p[0], b = p[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = (mn < mn_lim && k0 + i < k_lim) ? P[(size_t)mn * ld + k0 + i] : 0.f;
        }
    } else {
        const int k = k_base + (tid >> 4);
        const int mn0 = mn_base + (tid & 15) * 8;
        if (k < k_lim && mn0 + 8 <= mn_lim && vec_ok) {
            const float4* p = reinterpret_cast<const float4*>(P + (size_t)k * ld + mn0);
            const float4 a = p[0], b = p[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = (k < k_lim && mn0 + i < mn_lim) ? P[(size_t)k * ld + mn0 + i] : 0.f;
        }
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_strip(float (*S)[BM + PADF], int tid, const float v[8]) {
    if (KCONTIG) {
        const int mn = tid >> 1, k0 = (tid & 1) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) S[k0 + i][mn] = v[i];
    } else {
        const int k = tid >> 4, mn0 = (tid & 15) * 8;
        *reinterpret_cast<float4*>(&S[k][mn0]) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(&S[k][mn0 + 4]) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// XCD-aware tile order (cdna_hip_programming.md T1): workgroups are dealt round-robin over the 8
// XCDs, so consecutive linear ids land on different L2s.  Remap so that each XCD walks one
// CONTIGUOUS chunk of the (x fastest) tile order: tiles that share an operand panel then share
// an L2.  Bijective for any grid size; placement affects speed only.
__device__ __forceinline__ void swizzled_tile(int& bx, int& by, int& bz) {
    const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const unsigned nwg = gx * gy * gz;
    const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned q = nwg / 8, r = nwg % 8;
    const unsigned xcd = L % 8, i = L / 8;
    const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    bx = (int)(t % gx); by = (int)((t / gx) % gy); bz = (int)(t / (gx * gy));
}

template <bool TA, bool TB>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + PADF];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + PADF];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int tbx, tby, z;
    swizzled_tile(tbx, tby, z);
    const int bidx = z / g.splitk, sidx = z % g.splitk;
    const int m0 = tby * BM, n0 = tbx * BN;
    const float* A = g.A + (size_t)bidx * g.sA;
    const float* B = g.B + (size_t)bidx * g.sB;
    const int kbeg = sidx * g.kper;
    const int kend = min(g.K, kbeg + g.kper);
    const bool vecA = ((g.lda & 3) == 0) && ((((size_t)A) & 15) == 0);
    const bool vecB = ((g.ldb & 3) == 0) && ((((size_t)B) & 15) == 0);
    float nsh = 0.f, nsc = 1.f;
    if (g.norm_operand) { nsh = g.shift[bidx]; nsc = g.scale[bidx]; }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[8], rb[8];
    auto gload = [&](int kb) {
        // A stored MxK row-major (k contiguous) unless TA; B stored KxN (n contiguous) unless TB
        load_strip<!TA>(A, g.lda, m0, g.M, kb, kend, tid, vecA, ra);
        load_strip<TB>(B, g.ldb, n0, g.N, kb, kend, tid, vecB, rb);
        if (g.norm_operand == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ra[i] = (ra[i] - nsh) * nsc;
        } else if (g.norm_operand == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rb[i] = (rb[i] - nsh) * nsc;
        }
    };
    // out-of-range cells of a normalised operand must stay 0: handled by zeroing after the
    // transform for edge tiles (k tail or mn tail)
    auto fix_edges = [&](int kb) {
        if (g.norm_operand == 1) {
            if (!TA) { const int mn = m0 + (tid >> 1), k0 = kb + (tid & 1) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) if (!(mn < g.M && k0 + i < kend)) ra[i] = 0.f; }
            else { const int k = kb + (tid >> 4), mn0 = m0 + (tid & 15) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) if (!(k < kend && mn0 + i < g.M)) ra[i] = 0.f; }
        } else if (g.norm_operand == 2) {
            if (TB) { const int mn = n0 + (tid >> 1), k0 = kb + (tid & 1) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) if (!(mn < g.N && k0 + i < kend)) rb[i] = 0.f; }
            else { const int k = kb + (tid >> 4), mn0 = n0 + (tid & 15) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) if (!(k < kend && mn0 + i < g.N)) rb[i] = 0.f; }
        }
    };

    if (kbeg < kend) {
        gload(kbeg); fix_edges(kbeg);
        store_strip<!TA>(As[0], tid, ra);
        store_strip<TB>(Bs[0], tid, rb);
    }
    __syncthreads();
    int cur = 0;
    for (int kb = kbeg; kb < kend; kb += BK) {
        const bool more = kb + BK < kend;
        if (more) { gload(kb + BK); fix_edges(kb + BK); }
        const int r = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = As[cur][kk + kh][wm * 64 + r];
            const float a1 = As[cur][kk + kh][wm * 64 + 32 + r];
            const float b0 = Bs[cur][kk + kh][wn * 64 + r];
            const float b1 = Bs[cur][kk + kh][wn * 64 + 32 + r];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            store_strip<!TA>(As[cur ^ 1], tid, ra);
            store_strip<TB>(Bs[cur ^ 1], tid, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue
    if (!g.wide) { store_tiles_fast(g, acc, m0, n0, wm, wn, lane, z, bidx, true); return; }
    const int cl = lane & 31, rq = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + cl;
            if (n >= g.N) continue;
            float bsum = 0.f;
            if (!g.partial) {
                if (g.bias) bsum += g.bias[n];
                if (g.bias2) bsum += g.bias2[n];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                if (m >= g.M) continue;
                float v = g.alpha * acc[i][j][r];
                if (g.partial) {
                    g.partial[((size_t)z * g.M + m) * g.N + n] = v;
                } else {
                    float* c = g.C + (size_t)bidx * g.sC + (size_t)m * g.ldc + n;
                    v += bsum;
                    if (g.act == 1) v = v > 0.f ? v : v * g.slope;
                    if (g.dact_y) v *= (g.dact_y[(size_t)bidx * g.sC + (size_t)m * g.ldc + n] > 0.f ? 1.f : g.slope);
                    if (g.accumulate) v += *c;
                    *c = v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------
// bf16x3 variant: same tiling and epilogue, but each fp32 operand element is split ONCE per tile
// into bf16 hi + lo while it is staged into LDS (v_cvt_pk_bf16_f32), and every 16-deep k-step
// issues hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate, so ~5x after
// the 3 terms).  fp32 accumulate; dropping lo*lo leaves ~2^-16 relative error per product
// (measured ~1e-6 relative on the path's GEMMs).  LDS image per operand: [row][32 k + 8 pad] bf16,
// k fastest, so a fragment is one aligned 16-byte read and 16 consecutive rows fall on distinct
// banks; operands whose M/N index is contiguous in memory use the [k][m] image further down.
// ------------------------------------------------------------------------------------------
constexpr int XBK = 32, XPITCH = 40;   // halfs per LDS row

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi_pk, unsigned& lo_pk) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    const bf2 h = __builtin_convertvector((f2){x0, x1}, bf2);
    hi_pk = __builtin_bit_cast(unsigned, h);
    const float h0 = __uint_as_float(hi_pk << 16), h1 = __uint_as_float(hi_pk & 0xFFFF0000u);
    const bf2 l = __builtin_convertvector((f2){x0 - h0, x1 - h1}, bf2);
    lo_pk = __builtin_bit_cast(unsigned, l);
}

// 16-float strip of a 128 x 32 tile of a k-contiguous operand: thread -> (row = tid>>1, k0 = (tid&1)*16)
__device__ __forceinline__ void xload_strip(const float* __restrict__ P, int ld, int mn_base, int mn_lim,
                                            int k_base, int k_lim, int tid, bool vec_ok, float v[16]) {
    const int mn = mn_base + (tid >> 1);
    const int k0 = k_base + (tid & 1) * 16;
    if (mn < mn_lim && k0 + 16 <= k_lim && vec_ok) {
        const float4* p = reinterpret_cast<const float4*>(P + (size_t)mn * ld + k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float4 a = p[i]; v[4 * i] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w; }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (mn < mn_lim && k0 + i < k_lim) ? P[(size_t)mn * ld + k0 + i] : 0.f;
    }
}

__device__ __forceinline__ void xstore_strip(unsigned short* hi, unsigned short* lo, int tid, const float v[16]) {
    const int row = tid >> 1, k0 = (tid & 1) * 16;
    unsigned h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) split2(v[2 * i], v[2 * i + 1], h[i], l[i]);
    u32x4_t* ph = reinterpret_cast<u32x4_t*>(hi + row * XPITCH + k0);
    u32x4_t* pl = reinterpret_cast<u32x4_t*>(lo + row * XPITCH + k0);
    ph[0] = (u32x4_t){h[0], h[1], h[2], h[3]}; ph[1] = (u32x4_t){h[4], h[5], h[6], h[7]};
    pl[0] = (u32x4_t){l[0], l[1], l[2], l[3]}; pl[1] = (u32x4_t){l[4], l[5], l[6], l[7]};
}

// ---- operands stored K x M/N with the M/N index contiguous (A of a TN/TT product, B of a TN/NN product; both
// operands of the weight gradients dW = dY^T X) ----
// The LDS image keeps the global orientation, [k][128 m/n + 32 pad] bf16, and the MFMA fragments (8 consecutive k of
// one row) come out of gfx950's transposing read ds_read_b64_tr_b16: the global loads are coalesced along m/n
// (a half-wave reads 512 contiguous bytes of one k row), the staging writes are 8-byte (the [row][k] image above
// needs 32 two-byte writes per strip and reads 16 bytes from each of 64 different lines per load instruction).
// Pitch 160 halfs = 320 B = 64 mod 256: the four k rows of a half-wave's two 4 x 16 blocks cover all 64 banks once.
constexpr int TPITCH = 160;
static_assert(XBK * TPITCH == BM * XPITCH, "both LDS images fill the same array");
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

// thread -> (k = (tid>>5) + 8 i, 4 consecutive m/n from (tid&31)*4), i = 0..3
__device__ __forceinline__ void tload_strip(const float* __restrict__ P, int ld, int mn_base, int mn_lim,
                                            int k_base, int k_lim, int tid, bool vec_ok, float v[16]) {
    const int mn0 = mn_base + (tid & 31) * 4;
    // interior tiles take a WORKGROUP-uniform branch (s_cbranch_scc, four back-to-back 16-byte loads); a per-lane
    // condition here turns every load into its own exec-masked region
    if (vec_ok && k_base + XBK <= k_lim && mn_base + BM <= mn_lim) {
        const float* p = P + (size_t)(k_base + (tid >> 5)) * ld + mn0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(p + (size_t)(8 * i) * ld);
            v[4 * i] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k_base + (tid >> 5) + 8 * i;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * i + c] = (k < k_lim && mn0 + c < mn_lim) ? P[(size_t)k * ld + mn0 + c] : 0.f;
    }
}

__device__ __forceinline__ void tstore_strip(unsigned short* hi, unsigned short* lo, int tid, const float v[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h0, l0, h1, l1;
        split2(v[4 * i], v[4 * i + 1], h0, l0);
        split2(v[4 * i + 2], v[4 * i + 3], h1, l1);
        const int o = ((tid >> 5) + 8 * i) * TPITCH + (tid & 31) * 4;
        *reinterpret_cast<u32x2_t*>(hi + o) = (u32x2_t){h0, h1};
        *reinterpret_cast<u32x2_t*>(lo + o) = (u32x2_t){l0, l1};
    }
}

// 8 consecutive k (two 4 x 16 blocks) of this lane's row: ``p`` points at the lane's Mechanism address of the first
// block (row k0 + q, columns 4p..4p+3 of the lane group's 16), the second block is 4 k rows further down
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned short* p) {
    typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_ptr;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * TPITCH));
    return __builtin_bit_cast(bf16x8_t, (s16x8_t){a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w});
}

template <bool TA, bool TB, bool QUEUE>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16x3_kernel(GemmArgs g) {
    // [buffer][operand A/B][plane hi/lo][128 rows][XPITCH]
    __shared__ __attribute__((aligned(16))) unsigned short S[2][2][2][BM * XPITCH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // Queue mode (used for GEMMs that run beside a persistent LSTM sweep): a workgroup that finds itself on
    // an XCD the sweep occupies leaves at once; every other one draws ONE tile number from a global counter.
    // Which XCD a workgroup runs on is READ (HW_REG_XCC_ID), never assumed: the masked launch is followed by
    // an unmasked sweeper launch on the same counter that picks up whatever tiles are left (normally none),
    // so the result is complete for any placement.
    int tbx = 0, tby = 0, z = 0;
    bool live = true;      // (no early return: a dead workgroup just runs an empty k range and stores nothing)
    if (QUEUE) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xFu;
        const bool allowed = g.xcc_busy == nullptr ||
                             __hip_atomic_load(g.xcc_busy + (xcc & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
        // the tile number travels through the first word of the staging array (a second __shared__ object
        // next to it made hipcc 7.2 fall back to a 1-wave/SIMD register allocation)
        unsigned* mailbox = reinterpret_cast<unsigned*>(&S[0][0][0][0]);
        if (tid == 0) *mailbox = allowed ? __hip_atomic_fetch_add(g.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
        __syncthreads();
        const unsigned t = *mailbox;
        __syncthreads();
        live = t < (unsigned)g.tx * g.ty * g.tz;
        if (live) { tbx = (int)(t % g.tx); tby = (int)((t / g.tx) % g.ty); z = (int)(t / ((unsigned)g.tx * g.ty)); }
    } else {
        swizzled_tile(tbx, tby, z);
    }
    const int bidx = z / g.splitk, sidx = z % g.splitk;
    const int m0 = tby * BM, n0 = tbx * BN;
    const float* A = g.A + (size_t)bidx * g.sA;
    const float* B = g.B + (size_t)bidx * g.sB;
    const int kbeg = sidx * g.kper;
    const int kend = live ? min(g.K, kbeg + g.kper) : kbeg;
    const bool vecA = ((g.lda & 3) == 0) && ((((size_t)A) & 15) == 0);
    const bool vecB = ((g.ldb & 3) == 0) && ((((size_t)B) & 15) == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Two register sets: the loads of tile t+2 are issued while tile t is multiplied and tile t+1
    // (loaded one iteration earlier) is split and written to the other LDS buffer, so a load has
    // almost two iterations to land (the one-deep version spent 55 % of its wave cycles in
    // s_waitcnt: rocprofv3 SQ_WAIT_ANY, profiles/r01_gemm_pmc.txt).
    float ra0[16], rb0[16], ra1[16], rb1[16];
    const int fr = lane & 31, fk = (lane >> 5) * 8;
    // an operand whose M/N index is contiguous in memory (A stored K x M, B stored K x N) keeps that orientation in LDS
    // and is read through the transposing instruction (see tload_strip); a k-contiguous operand uses the [row][k] image
    constexpr bool TRA = TA, TRB = !TB;
    // lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3 of the group's 4 x 16 block; groups 0,1 take
    // columns 0-15 / 16-31 of k 0-7, groups 2,3 the same columns of k 8-15 (= the 32x32x16 operand map)
    const int tro = (8 * (lane >> 5) + ((lane & 15) >> 2)) * TPITCH + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto compute = [&](int cur) {
#pragma unroll
        for (int kk = 0; kk < XBK; kk += 16) {
            bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (TRA) {
                    const int ro = kk * TPITCH + wm * 64 + i * 32 + tro;
                    ah[i] = tr_frag(&S[cur][0][0][ro]);
                    al[i] = tr_frag(&S[cur][0][1][ro]);
                } else {
                    const int ro = (wm * 64 + i * 32 + fr) * XPITCH + kk + fk;
                    ah[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(&S[cur][0][0][ro]));
                    al[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(&S[cur][0][1][ro]));
                }
                if constexpr (TRB) {
                    const int co = kk * TPITCH + wn * 64 + i * 32 + tro;
                    bh[i] = tr_frag(&S[cur][1][0][co]);
                    bl[i] = tr_frag(&S[cur][1][1][co]);
                } else {
                    const int co = (wn * 64 + i * 32 + fr) * XPITCH + kk + fk;
                    bh[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(&S[cur][1][0][co]));
                    bl[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(&S[cur][1][1][co]));
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    // per-utterance instance norm fused into the load of operand A (norm_operand 1) or B (2): applied after
    // the plain load under a wave-uniform branch (unconditional sub+mul cost every GEMM ~10 %), in-range only
    const int nrm = g.norm_operand;
    const float nsh = nrm ? g.shift[bidx] : 0.f, nsc = nrm ? g.scale[bidx] : 1.f;
    auto norm_strip = [&](float (&v)[16], bool tr, int mn_base, int mn_lim, int kb) {     // tr: the operand's strip mapping
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int mn = tr ? mn_base + (tid & 31) * 4 + (i & 3) : mn_base + (tid >> 1);
            const int k = tr ? kb + (tid >> 5) + 8 * (i >> 2) : kb + (tid & 1) * 16 + i;
            v[i] = (mn < mn_lim && k < kend) ? (v[i] - nsh) * nsc : 0.f;
        }
    };
#define XLOAD(RA, RB, KB) do { if constexpr (TRA) tload_strip(A, g.lda, m0, g.M, (KB), kend, tid, vecA, RA); \
                               else xload_strip(A, g.lda, m0, g.M, (KB), kend, tid, vecA, RA); \
                               if constexpr (TRB) tload_strip(B, g.ldb, n0, g.N, (KB), kend, tid, vecB, RB); \
                               else xload_strip(B, g.ldb, n0, g.N, (KB), kend, tid, vecB, RB); \
                               if (nrm == 1) norm_strip(RA, TRA, m0, g.M, (KB)); \
                               else if (nrm == 2) norm_strip(RB, TRB, n0, g.N, (KB)); } while (0)
#define XSTORE(RA, RB, BUF) do { if constexpr (TRA) tstore_strip(S[BUF][0][0], S[BUF][0][1], tid, RA); \
                                 else xstore_strip(S[BUF][0][0], S[BUF][0][1], tid, RA); \
                                 if constexpr (TRB) tstore_strip(S[BUF][1][0], S[BUF][1][1], tid, RB); \
                                 else xstore_strip(S[BUF][1][0], S[BUF][1][1], tid, RB); } while (0)
    if (kbeg < kend) {
        XLOAD(ra0, rb0, kbeg);
        XSTORE(ra0, rb0, 0);
        if (kbeg + XBK < kend) XLOAD(ra0, rb0, kbeg + XBK);
    }
    __syncthreads();
    for (int kb = kbeg; kb < kend; kb += 2 * XBK) {
        // even tile: pending set 0 (tile kb+XBK), free set 1
        if (kb + 2 * XBK < kend) XLOAD(ra1, rb1, kb + 2 * XBK);
        compute(0);
        if (kb + XBK < kend) XSTORE(ra0, rb0, 1);
        __syncthreads();
        if (kb + XBK >= kend) break;
        // odd tile: pending set 1 (tile kb+2*XBK), free set 0
        if (kb + 3 * XBK < kend) XLOAD(ra0, rb0, kb + 3 * XBK);
        compute(1);
        if (kb + 2 * XBK < kend) XSTORE(ra1, rb1, 0);
        __syncthreads();
    }
#undef XLOAD
#undef XSTORE

    // epilogue (identical to the fp32 kernel: the 32x32 accumulator layout is dtype independent)
    if (!g.wide) { store_tiles_fast(g, acc, m0, n0, wm, wn, lane, z, bidx, live); return; }
    const int cl = lane & 31, rq = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + cl;
            if (n >= g.N || !live) continue;
            float bsum = 0.f;
            if (!g.partial) {
                if (g.bias) bsum += g.bias[n];
                if (g.bias2) bsum += g.bias2[n];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                if (m >= g.M) continue;
                float v = g.alpha * acc[i][j][r];
                if (g.partial) {
                    g.partial[((size_t)z * g.M + m) * g.N + n] = v;
                } else {
                    float* c = g.C + (size_t)bidx * g.sC + (size_t)m * g.ldc + n;
                    v += bsum;
                    if (g.act == 1) v = v > 0.f ? v : v * g.slope;
                    if (g.dact_y) v *= (g.dact_y[(size_t)bidx * g.sC + (size_t)m * g.ldc + n] > 0.f ? 1.f : g.slope);
                    if (g.accumulate) v += *c;
                    *c = v;
                }
            }
        }
}

// Sum the Z partial slabs in index order (deterministic) and apply the epilogue.
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float* __restrict__ partial, int Z, int M, int N,
                                                          float* __restrict__ C, int ldc, long long sC,
                                                          const float* bias,
                                                          const float* bias2, int act, float slope, int accumulate) {
    // blockIdx.y = output batch; its Z slabs are contiguous
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    partial += (size_t)blockIdx.y * Z * M * N;
    C += (size_t)blockIdx.y * sC;
    float v = 0.f;
    for (int zz = 0; zz < Z; ++zz) v += partial[(size_t)zz * M * N + idx];
    if (bias) v += bias[n];
    if (bias2) v += bias2[n];
    if (act == 1) v = v > 0.f ? v : v * slope;
    float* c = C + (size_t)m * ldc + n;
    if (accumulate) v += *c;
    *c = v;
}

// column sums of X (rows x cols, leading dim ld): bias gradients
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int rows, int cols, int ld,
                                                             int rows_per_block, float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + rl; r < r1; r += 4) s += X[(size_t)r * ld + c];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < cols)
        partial[(size_t)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nparts, int cols,
                                                           float* __restrict__ out, float* __restrict__ out2, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * cols + c];
    if (accumulate) { out[c] += s; if (out2) out2[c] += s; }
    else { out[c] = s; if (out2) out2[c] = s; }
}

// per-utterance mean and 1/sqrt(var+eps) over all F*T values (model.py:37,48), fp64 sums
__global__ __launch_bounds__(1024) void instnorm_stats_kernel(const float* __restrict__ x, long long n_per, float eps,
                                                              float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ double s1[16], s2[16];
    const float* p = x + (size_t)blockIdx.x * n_per;
    double a = 0.0, b = 0.0;
    long long done = 0;
    if (((n_per & 3) == 0) && ((((size_t)x) & 15) == 0)) {     // 16-byte loads, four in flight per thread
        const float4* p4 = reinterpret_cast<const float4*>(p);
        const long long n4 = n_per >> 2;
        double a1 = 0.0, b1 = 0.0;
        long long i = threadIdx.x;
        for (; i + 3 * 1024 < n4; i += 4 * 1024) {
            const float4 u0 = p4[i], u1 = p4[i + 1024], u2 = p4[i + 2 * 1024], u3 = p4[i + 3 * 1024];
            a += ((double)u0.x + (double)u0.y) + ((double)u0.z + (double)u0.w);
            b += ((double)u0.x * u0.x + (double)u0.y * u0.y) + ((double)u0.z * u0.z + (double)u0.w * u0.w);
            a1 += ((double)u1.x + (double)u1.y) + ((double)u1.z + (double)u1.w);
            b1 += ((double)u1.x * u1.x + (double)u1.y * u1.y) + ((double)u1.z * u1.z + (double)u1.w * u1.w);
            a += ((double)u2.x + (double)u2.y) + ((double)u2.z + (double)u2.w);
            b += ((double)u2.x * u2.x + (double)u2.y * u2.y) + ((double)u2.z * u2.z + (double)u2.w * u2.w);
            a1 += ((double)u3.x + (double)u3.y) + ((double)u3.z + (double)u3.w);
            b1 += ((double)u3.x * u3.x + (double)u3.y * u3.y) + ((double)u3.z * u3.z + (double)u3.w * u3.w);
        }
        for (; i < n4; i += 1024) {
            const float4 u = p4[i];
            a += ((double)u.x + (double)u.y) + ((double)u.z + (double)u.w);
            b += ((double)u.x * u.x + (double)u.y * u.y) + ((double)u.z * u.z + (double)u.w * u.w);
        }
        a += a1; b += b1;
        done = n_per;
    }
    for (long long i = done + threadIdx.x; i < n_per; i += 1024) { const double v = p[i]; a += v; b += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = a; s2[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int i = 0; i < 16; ++i) { ta += s1[i]; tb += s2[i]; }
        const double mu = ta / (double)n_per;
        double var = tb / (double)n_per - mu * mu;
        if (var < 0.0) var = 0.0;
        mean[blockIdx.x] = (float)mu;
        rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

}  // namespace

static int gemm_launch(GemmArgs& g, int transA, int transB, int sum_batches, int precision, const unsigned* xcc_busy, void* workspace,
                       size_t workspace_bytes, hipStream_t st) {
    const int Z = g.batch * g.splitk;
    const bool use_partial = (g.splitk > 1) || sum_batches;
    const bool queue_mode = precision >= 1 && xcc_busy != nullptr;
    const size_t head = 256;   // the first 256 workspace bytes hold the tile counter of queue mode
    if (use_partial || queue_mode) {
        const size_t need = head + (use_partial ? (size_t)Z * g.M * g.N * sizeof(float) : 0);
        if (!workspace || workspace_bytes < need) return PGASR_ERR_WORKSPACE;
    }
    g.partial = use_partial ? (float*)((char*)workspace + head) : nullptr;
    g.queue = nullptr; g.xcc_busy = nullptr;
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, Z);
    if (grid.y > 65535 || grid.z > 65535) return PGASR_ERR_UNSUPPORTED;
    g.tx = (int)grid.x; g.ty = (int)grid.y; g.tz = (int)grid.z;
    if (queue_mode) {
        g.queue = (unsigned*)workspace; g.xcc_busy = xcc_busy;
        if (hipMemsetAsync(workspace, 0, head, st) != hipSuccess) return PGASR_ERR_LAUNCH;
    }
    const unsigned ntiles_q = grid.x * grid.y * grid.z;

    // weight-gradient shaped TN products (both operands k-major fp32, K long, split over K): the 256 x 256 LDS-DMA tile
    // of gemm_dma.hip writes the same raw partial slabs; the reduce below is shared
    bool done_tn256 = false;
    if (precision >= 1 && transA && !transB && use_partial && g.norm_operand == 0) {
        PgasrTn256Args t{g.A, g.B, g.partial, g.M, g.N, g.K, g.lda, g.ldb, g.sA, g.sB, g.batch, g.splitk, g.kper, g.alpha,
                         queue_mode ? g.queue : nullptr, queue_mode ? g.xcc_busy : nullptr, getenv("PGASR_TN_DIAG") ? atoi(getenv("PGASR_TN_DIAG")) : 0,
                         nullptr, nullptr, 0, 0, 0, 0, nullptr};     // no gate, no time slabs
        if (pgasr_internal_tn256_ok(t)) {
            const int st_ = precision == 2 ? pgasr_internal_tn6_launch(t, queue_mode ? 1 : 0, st) : pgasr_internal_tn256_launch(t, queue_mode ? 1 : 0, st);
            if (st_ != PGASR_OK) return st_;
            done_tn256 = true;
        }
    }
    if (done_tn256) {
    }
    else if (precision == 2) return PGASR_ERR_UNSUPPORTED;      // the six-product arithmetic exists for the 256 x 256 TN shapes only
    else if (precision == 1) {
        if (queue_mode) {
            // pass 0: masked, enough workgroups that the allowed XCDs alone can cover every tile under
            // round-robin dealing; pass 1: unmasked sweeper for any tiles left over (normally all exit at once)
            for (int pass = 0; pass < 2; ++pass) {
                // pass 0: twice the tiles, so that half the XCDs alone can cover them; pass 1 ignores the hint
                dim3 qgrid(pass == 0 ? 2u * ntiles_q + 8u : ntiles_q);
                if (pass == 1) g.xcc_busy = nullptr;
                if (!transA && !transB) PGASR_LAUNCH_KERNEL((gemm_bf16x3_kernel<false, false, true>), qgrid, dim3(GEMM_THREADS), 0, st, g);
                else if (!transA && transB) PGASR_LAUNCH_KERNEL((gemm_bf16x3_kernel<false, true, true>), qgrid, dim3(GEMM_THREADS), 0, st, g);
                else if (transA && !transB) PGASR_LAUNCH_KERNEL((gemm_bf16x3_kernel<true, false, true>), qgrid, dim3(GEMM_THREADS), 0, st, g);
                else PGASR_LAUNCH_KERNEL((gemm_bf16x3_kernel<true, true, true>), qgrid, dim3(GEMM_THREADS), 0, st, g);
                PGASR_CHECK_LAUNCH();
            }
        }
        else if (!transA && !transB) PGASR_LAUNCH_KERNEL((gemm_bf16x3_kernel<false, false, false>), grid, dim3(GEMM_THREADS), 0, st, g);
        else if (!transA && transB) PGASR_LAUNCH_KERNEL((gemm_bf16x3_kernel<false, true, false>), grid, dim3(GEMM_THREADS), 0, st, g);
        else if (transA && !transB) PGASR_LAUNCH_KERNEL((gemm_bf16x3_kernel<true, false, false>), grid, dim3(GEMM_THREADS), 0, st, g);
        else PGASR_LAUNCH_KERNEL((gemm_bf16x3_kernel<true, true, false>), grid, dim3(GEMM_THREADS), 0, st, g);
    }
    else if (!transA && !transB) PGASR_LAUNCH_KERNEL((gemm_f32_kernel<false, false>), grid, dim3(GEMM_THREADS), 0, st, g);
    else if (!transA && transB) PGASR_LAUNCH_KERNEL((gemm_f32_kernel<false, true>), grid, dim3(GEMM_THREADS), 0, st, g);
    else if (transA && !transB) PGASR_LAUNCH_KERNEL((gemm_f32_kernel<true, false>), grid, dim3(GEMM_THREADS), 0, st, g);
    else PGASR_LAUNCH_KERNEL((gemm_f32_kernel<true, true>), grid, dim3(GEMM_THREADS), 0, st, g);
    PGASR_CHECK_LAUNCH();
    if (use_partial) {
        const size_t total = (size_t)g.M * g.N;
        const int nout = sum_batches ? 1 : g.batch;
        PGASR_LAUNCH_KERNEL(gemm_reduce_kernel, dim3((unsigned)((total + 255) / 256), nout), dim3(256), 0, st,
                           g.partial, Z / nout, g.M, g.N, g.C, g.ldc, g.sC, g.bias, g.bias2, g.act, g.slope,
                           g.accumulate);
        PGASR_CHECK_LAUNCH();
    }
    return PGASR_OK;
}

extern "C" size_t pgasr_gemm_workspace_bytes(int M, int N, int batch, int splitk, int sum_batches) {
    if (M <= 0 || N <= 0 || batch <= 0 || splitk <= 0) return 0;
    if (splitk <= 1 && !sum_batches) return 256;
    return 256 + (size_t)batch * splitk * M * N * sizeof(float);
}

extern "C" int pgasr_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                              const float* A, int lda, long long strideA,
                              const float* B, int ldb, long long strideB,
                              float* C, int ldc, long long strideC,
                              int batch, int sum_batches, int splitk,
                              const float* bias, const float* bias2, int act, float slope, int accumulate,
                              const float* dact_y, int norm_operand, const float* shift, const float* scale,
                              int precision, const unsigned* xcc_busy, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || splitk <= 0) return PGASR_ERR_INVALID_ARG;
    if (norm_operand < 0 || norm_operand > 2 || (norm_operand && (!shift || !scale))) return PGASR_ERR_INVALID_ARG;
    if (act < 0 || act > 1 || precision < 0 || precision > 2) return PGASR_ERR_INVALID_ARG;
    if ((splitk > 1 || sum_batches) && dact_y) return PGASR_ERR_INVALID_ARG;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sA = strideA; g.sB = strideB; g.sC = strideC; g.batch = batch; g.splitk = splitk;
    int kper = (K + splitk - 1) / splitk;
    const int kq = precision >= 1 ? XBK : BK;
    kper = (kper + kq - 1) / kq * kq;
    g.kper = kper;
    g.alpha = alpha; g.bias = bias; g.bias2 = bias2; g.act = act; g.slope = slope; g.accumulate = accumulate;
    g.dact_y = dact_y; g.norm_operand = norm_operand; g.shift = shift; g.scale = scale; g.partial = nullptr;
    g.wide = (((size_t)(M - 1) * ldc + N) * 4 >= ((size_t)1 << 32) || (size_t)M * N * 4 >= ((size_t)1 << 32)) ? 1 : 0;
    return gemm_launch(g, transA, transB, sum_batches, precision, xcc_busy, workspace, workspace_bytes, (hipStream_t)stream);
}

// The two weight-gradient products of one BLSTM layer in ONE queue-mode launch of the 256 x 256 TN kernel (gemm_c256.hip):
//   dwih_perm (2*4H x in_dim)   = dgates^T x                      (K = T*B rows)
//   dwhh_perm (2 x 4H x H)      = dgates[d]^T h_prev(d)           (K = (T-1)*B rows; prev = t-1 for d = 0, t+1 for d = 1)
// bf16x3 arithmetic; each 256 x 256 tile is the sum, in the order a backward sweep completes them, of its partial products over the
// TIME slabs of pgasr_wslab_edge(T, .) (common.h: a function of T only, so every mode gives the same bits).
// slab_done != NULL: dgates belongs to a backward sweep that is STILL RUNNING (pgasr_lstm_layer_bwd_streamed on another stream,
// launched before this call); the work items then wait for the slab_done words covering their rows and read dgates with agent-scope loads.
extern "C" int pgasr_lstm_wgrad_slabs(int T, int* edges, int max_edges) {
    if (T <= 0) return 0;
    const int n = pgasr_wslab_count(T);
    for (int i = 0; edges && i <= n && i < max_edges; ++i) edges[i] = pgasr_wslab_edge(T, i);
    return n;
}
extern "C" size_t pgasr_lstm_wgrads_workspace_bytes(int T, int in_dim) {
    if (in_dim <= 0 || T <= 0) return 0;
    return 256 + (size_t)pgasr_wslab_count(T) * ((size_t)2048 * in_dim + (size_t)2 * 1024 * 256) * sizeof(float);
}
extern "C" int pgasr_lstm_wgrads_streamed(const float* dgates, const float* x, const float* out, int T, int B, int in_dim,
                                          float* dwih_perm, float* dwhh_perm, const unsigned* xcc_busy,
                                          const unsigned* slab_done, int* err_word, int planes,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    constexpr int H = 256, G = 2 * 4 * H;
    if (!dgates || !x || !out || !dwih_perm || !dwhh_perm || T <= 1 || B <= 0 || in_dim <= 0) return PGASR_ERR_INVALID_ARG;
    if (planes != 2 && planes != 3) return PGASR_ERR_INVALID_ARG;
    const size_t need = pgasr_lstm_wgrads_workspace_bytes(T, in_dim);
    if (!workspace || workspace_bytes < need) return PGASR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nslab = pgasr_wslab_count(T);
    float* part_ih = (float*)((char*)workspace + 256);
    float* part_hh = part_ih + (size_t)nslab * G * in_dim;
    const long long K_ih = (long long)T * B, K_hh = (long long)(T - 1) * B;
    if (K_ih >= ((long long)1 << 31)) return PGASR_ERR_UNSUPPORTED;
    PgasrTn256Args ih{dgates, x, part_ih, G, in_dim, (int)K_ih, G, in_dim, 0, 0, 1, nslab, 0, 1.f,
                      (unsigned*)workspace, xcc_busy, 0, slab_done, dgates, T, B, (B + 15) / 16, 1, err_word};
    PgasrTn256Args hh{dgates + (size_t)B * G, out, part_hh, 4 * H, H, (int)K_hh, G, 2 * H, (long long)4 * H - (long long)B * G,
                      (long long)B * 2 * H + H, 2, nslab, 0, 1.f,
                      (unsigned*)workspace, xcc_busy, 0, slab_done, dgates, T, B, (B + 15) / 16, 1, err_word};
    const int tk = planes == 3 ? 16 : 32;     // rows per k-step: the six-product kernel takes B % 16 == 0, the bf16x3 one B % 32 == 0
    if (!pgasr_internal_tn256_ok(ih, tk) || !pgasr_internal_tn256_ok(hh, tk)) return PGASR_ERR_UNSUPPORTED;
    if (hipMemsetAsync(workspace, 0, 256, st) != hipSuccess) return PGASR_ERR_LAUNCH;
    const int rc = planes == 3 ? pgasr_internal_tn6_launch(ih, 1, st, &hh) : pgasr_internal_tn256_launch(ih, 1, st, &hh);
    if (rc != PGASR_OK) return rc;
    // Why the time-slab partials are summed by two small launches and not by the last-arriving workgroup of each tile (asked twice in
    // review): (1) the sum over the slabs IN SLAB ORDER is what makes every order of a step give the same bits, and a fixed-order sum by
    // "whoever arrives last" means that workgroup reads 11 partial tiles of 256 KB one after the other on ONE CU (~2.9 MB at the ~0.15 TB/s
    // a single CU streams: ~20 us per tile, 24 tiles) while it holds a persistent slot of the queue the still-running sweep feeds; the two
    // launches move the same 34 MB with the whole chip in 46-54 + 13-16 us (kernel timeline of a step, profiles/r04_step_timeline_f32.txt).
    // (2) They are not on the step's chain: they run on the weight-gradient stream behind the products, beside the NEXT layer's sweep; the
    // first layer's (the tail) run beside the main stream's dropout -> input-layer gradient -> Adam chain and end before it does.
    PGASR_LAUNCH_KERNEL(gemm_reduce_kernel, dim3((unsigned)(((size_t)G * in_dim + 255) / 256), 1), dim3(256), 0, st,
                       part_ih, nslab, G, in_dim, dwih_perm, in_dim, (long long)0, (const float*)nullptr, (const float*)nullptr, 0, 0.f, 0);
    PGASR_CHECK_LAUNCH();
    PGASR_LAUNCH_KERNEL(gemm_reduce_kernel, dim3((unsigned)(((size_t)4 * H * H + 255) / 256), 2), dim3(256), 0, st,
                       part_hh, nslab, 4 * H, H, dwhh_perm, H, (long long)4 * H * H, (const float*)nullptr, (const float*)nullptr, 0, 0.f, 0);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" size_t pgasr_colsum_workspace_bytes(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    const int rpb = 512;
    return (size_t)((rows + rpb - 1) / rpb) * cols * sizeof(float);
}

extern "C" int pgasr_colsum_f32(const float* X, int rows, int cols, int ld, float* out, float* out2, int accumulate,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (!X || !out || rows <= 0 || cols <= 0 || ld < cols) return PGASR_ERR_INVALID_ARG;
    const int rpb = 512;
    const int nparts = (rows + rpb - 1) / rpb;
    if (!workspace || workspace_bytes < (size_t)nparts * cols * sizeof(float)) return PGASR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    PGASR_LAUNCH_KERNEL(colsum_partial_kernel, dim3((cols + 63) / 64, nparts), dim3(256), 0, st, X, rows, cols, ld, rpb,
                       (float*)workspace);
    PGASR_CHECK_LAUNCH();
    PGASR_LAUNCH_KERNEL(colsum_final_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, (const float*)workspace, nparts,
                       cols, out, out2, accumulate);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_instnorm_stats(const float* x, int B, int F, int T, float eps, float* mean, float* rstd,
                                    void* stream) {
    if (!x || !mean || !rstd || B <= 0 || F <= 0 || T <= 0) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(instnorm_stats_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, (long long)F * T, eps,
                       mean, rstd);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}
