// CTC head (model.py:52-55 as the build's Seq2Seq uses it: Linear(2H -> V) + log_softmax over V) as ONE kernel.
//
// The product is skinny: M = T*B rows (32,000), K = 512, V = 29 outputs.  On the general 128 x 128 fp32 kernel three quarters of the MFMA
// work multiply padding and the launch took 78 us, followed by a separate log-softmax pass (8 us + two dispatch gaps) -- all of it on the
// step's critical chain between the last forward sweep and the CTC lattice.  Here a wave owns 32 rows: v_mfma_f32_32x32x2_f32 with the
// rows as M and the (zero padded) 32 outputs as N, i.e. exact fp32 arithmetic in both precision modes; x streams from HBM straight into
// the A operand registers (one pass over the 65 MB, 8 k-steps in flight per wave), W^T sits in LDS in B-operand order, and the
// log-softmax is taken on the accumulators (a row's 32 outputs are the 32 lanes of a half wave).  Bound: HBM read of x -- reached to
// 1.3-1.4 TB/s only (49 us; MFMA time is 7 us): one wave per SIMD walks its 64 KB in 8 dependent load rounds, and a load instruction
// touches 32 lines for 32 bytes each (the A-operand layout wants a row per lane).  Staging x through LDS with full-line LDS-DMA would
// fix both; not built -- the kernel is 0.5 % of the step.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HEAD_KMAX = 1024;           // LDS: K * 32 outputs * 4 B <= 128 KB
constexpr int HEAD_UNIT = 4;              // K must be a multiple of 16 * HEAD_UNIT

// grid.x workgroups of 4 waves; wave w of workgroup g owns row blocks g*4 + w, + 4*grid.x, ...
// HEAD_CHUNK: 16-deep k-steps loaded together (2 register sets of HEAD_CHUNK * 2 float4); NT: non-temporal loads of x
template <int HEAD_CHUNK, bool NT>
__global__ __launch_bounds__(256) void head_logsoftmax_kernel(const float* __restrict__ x, long long M, int K, int ldx,
                                                              const float* __restrict__ W, const float* __restrict__ bias, int V,
                                                              float* __restrict__ logits, float* __restrict__ lp) {
    extern __shared__ __attribute__((aligned(16))) float Wt[];      // [K/16][2][32][8]: lane (v, kk) of step s reads 8 consecutive floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < K * 8; i += 256) {                        // float4 pieces: (s, kk, v, half)
        const int half = i & 1, v = (i >> 1) & 31, kk = (i >> 6) & 1, s = i >> 7;
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        if (v < V) w = *reinterpret_cast<const f32x4*>(W + (size_t)v * K + s * 16 + kk * 8 + half * 4);
        *reinterpret_cast<f32x4*>(Wt + (size_t)i * 4) = w;
    }
    __syncthreads();
    const int r = lane & 31, kk = lane >> 5;
    const float bv = (bias != nullptr && r < V) ? bias[r] : 0.f;
    const long long nblk = (M + 31) / 32;
    const int nchunk = K / (16 * HEAD_CHUNK);                       // K % 64 == 0 (checked by the host)
    for (long long blk = (long long)blockIdx.x * 4 + wave; blk < nblk; blk += (long long)gridDim.x * 4) {
        long long row = blk * 32 + r;
        if (row >= M) row = M - 1;                                  // the tail block reads a valid row, its stores are masked
        const float* xr = x + row * ldx + kk * 8;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        f32x4 a[2][HEAD_CHUNK][2];
        auto load = [&](int set, int c) {
#pragma unroll
            for (int s = 0; s < HEAD_CHUNK; ++s) {
                const float* p = xr + (c * HEAD_CHUNK + s) * 16;
                if (NT) {
                    a[set][s][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
                    a[set][s][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
                } else {
                    a[set][s][0] = *reinterpret_cast<const f32x4*>(p);
                    a[set][s][1] = *reinterpret_cast<const f32x4*>(p + 4);
                }
            }
        };
        auto mul = [&](int set, int c) {
#pragma unroll
            for (int s = 0; s < HEAD_CHUNK; ++s) {
                const float* wp = Wt + ((size_t)((c * HEAD_CHUNK + s) * 2 + kk) * 32 + r) * 8;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][s][0][e], w0[e], acc, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][s][1][e], w1[e], acc, 0, 0, 0);
            }
        };
        load(0, 0);
        for (int c = 0; c < nchunk; c += 2) {
            if (c + 1 < nchunk) load(1, c + 1);
            mul(0, c);
            if (c + 1 < nchunk) {
                if (c + 2 < nchunk) load(0, c + 2);
                mul(1, c + 1);
            }
        }
        // acc[i]: row 8*(i/4) + 4*kk + i%4 of the block, output v = r
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long long orow = blk * 32 + 8 * (i >> 2) + 4 * kk + (i & 3);
            const float z = acc[i] + bv;
            float mx = r < V ? z : -INFINITY;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float e = r < V ? __expf(z - mx) : 0.f;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
            const float lse = mx + __logf(e);
            if (r < V && orow < M) {
                if (logits) logits[orow * V + r] = z;
                if (lp) lp[orow * V + r] = z - lse;
            }
        }
    }
}

}  // namespace

extern "C" int pgasr_head_logsoftmax(const float* x, long long rows, int K, int ldx, const float* W, const float* bias, int V,
                                     float* logits, float* log_probs, void* stream) {
    if (!x || !W || (!logits && !log_probs) || rows <= 0 || K <= 0 || V <= 0 || ldx < K) return PGASR_ERR_INVALID_ARG;
    if (V > 32 || K > HEAD_KMAX || (K % (16 * HEAD_UNIT)) || (ldx & 3) || (((size_t)x) & 15) || (((size_t)W) & 15)) return PGASR_ERR_UNSUPPORTED;
    const size_t lds = (size_t)K * 32 * sizeof(float);
    // measured at (32000, 512, 29), x cold: <4, nt> 48.8 us, <8, nt> 50.9, <8, plain> 54.2, <4, plain> 53.4 (general fp32 GEMM + log-softmax pass: 110)
    auto kern = head_logsoftmax_kernel<4, true>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    const long long nblk = (rows + 31) / 32;
    const long long groups = (nblk + 3) / 4;
    const unsigned grid = (unsigned)(groups < 1024 ? groups : 1024);
    PGASR_LAUNCH_KERNEL(kern, dim3(grid), dim3(256), lds, (hipStream_t)stream, x, rows, K, ldx, W, bias, V, logits, log_probs);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}
