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
};

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

template <bool TA, bool TB>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + PADF];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + PADF];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int z = blockIdx.z;
    const int bidx = z / g.splitk, sidx = z % g.splitk;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
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
    for (long long i = threadIdx.x; i < n_per; i += 1024) { const double v = p[i]; a += v; b += v * v; }
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

static int gemm_launch(GemmArgs& g, int transA, int transB, int sum_batches, void* workspace, size_t workspace_bytes,
                       hipStream_t st) {
    const int Z = g.batch * g.splitk;
    const bool use_partial = (g.splitk > 1) || sum_batches;
    if (use_partial) {
        const size_t need = (size_t)Z * g.M * g.N * sizeof(float);
        if (!workspace || workspace_bytes < need) return PGASR_ERR_WORKSPACE;
        g.partial = (float*)workspace;
    } else {
        g.partial = nullptr;
    }
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, Z);
    if (grid.y > 65535 || grid.z > 65535) return PGASR_ERR_UNSUPPORTED;
    if (!transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(GEMM_THREADS), 0, st, g);
    else if (!transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(GEMM_THREADS), 0, st, g);
    else if (transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(GEMM_THREADS), 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(GEMM_THREADS), 0, st, g);
    PGASR_CHECK_LAUNCH();
    if (use_partial) {
        const size_t total = (size_t)g.M * g.N;
        const int nout = sum_batches ? 1 : g.batch;
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((total + 255) / 256), nout), dim3(256), 0, st,
                           g.partial, Z / nout, g.M, g.N, g.C, g.ldc, g.sC, g.bias, g.bias2, g.act, g.slope,
                           g.accumulate);
        PGASR_CHECK_LAUNCH();
    }
    return PGASR_OK;
}

extern "C" size_t pgasr_gemm_workspace_bytes(int M, int N, int batch, int splitk, int sum_batches) {
    if (M <= 0 || N <= 0 || batch <= 0 || splitk <= 0) return 0;
    if (splitk <= 1 && !sum_batches) return 0;
    return (size_t)batch * splitk * M * N * sizeof(float);
}

extern "C" int pgasr_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                              const float* A, int lda, long long strideA,
                              const float* B, int ldb, long long strideB,
                              float* C, int ldc, long long strideC,
                              int batch, int sum_batches, int splitk,
                              const float* bias, const float* bias2, int act, float slope, int accumulate,
                              const float* dact_y, int norm_operand, const float* shift, const float* scale,
                              void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || splitk <= 0) return PGASR_ERR_INVALID_ARG;
    if (norm_operand < 0 || norm_operand > 2 || (norm_operand && (!shift || !scale))) return PGASR_ERR_INVALID_ARG;
    if (act < 0 || act > 1) return PGASR_ERR_INVALID_ARG;
    if ((splitk > 1 || sum_batches) && dact_y) return PGASR_ERR_INVALID_ARG;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sA = strideA; g.sB = strideB; g.sC = strideC; g.batch = batch; g.splitk = splitk;
    int kper = (K + splitk - 1) / splitk;
    kper = (kper + BK - 1) / BK * BK;
    g.kper = kper;
    g.alpha = alpha; g.bias = bias; g.bias2 = bias2; g.act = act; g.slope = slope; g.accumulate = accumulate;
    g.dact_y = dact_y; g.norm_operand = norm_operand; g.shift = shift; g.scale = scale; g.partial = nullptr;
    return gemm_launch(g, transA, transB, sum_batches, workspace, workspace_bytes, (hipStream_t)stream);
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
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 63) / 64, nparts), dim3(256), 0, st, X, rows, cols, ld, rpb,
                       (float*)workspace);
    PGASR_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, (const float*)workspace, nparts,
                       cols, out, out2, accumulate);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_instnorm_stats(const float* x, int B, int F, int T, float eps, float* mean, float* rstd,
                                    void* stream) {
    if (!x || !mean || !rstd || B <= 0 || F <= 0 || T <= 0) return PGASR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(instnorm_stats_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, (long long)F * T, eps,
                       mean, rstd);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}
