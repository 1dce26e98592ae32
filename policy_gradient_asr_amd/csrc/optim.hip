// Elementwise pieces of the train step for gfx950: inverted dropout with a counter-based mask
// (nn.Dropout of model.py:45,51 and the inter-layer LSTM dropout of model.py:42) and Adam on the
// flat parameter buffer (optim.Adam(lr=5e-4), model.py:207).  HBM-bound streaming kernels:
// 16 B per lane, grid-stride.
#include "common.h"

namespace {

// keep-mask for 4 consecutive elements from one Philox4x32-10 call:
// counter = (quad index lo, quad index hi, offset, 0), key = seed
__device__ __forceinline__ void keep4(unsigned long long quad, uint32_t offset, uint32_t k0, uint32_t k1,
                                      uint32_t thresh, bool keep[4]) {
    uint32_t r[4];
    philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), offset, 0u, k0, k1, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) keep[i] = r[i] >= thresh;   // P(drop) = thresh / 2^32
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      unsigned long long n, uint32_t thresh, float scale,
                                                      uint32_t k0, uint32_t k1, uint32_t offset,
                                                      const float* __restrict__ dact_y, float slope) {
    const unsigned long long nq = (n + 3) / 4;
    for (unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x; q < nq;
         q += (unsigned long long)gridDim.x * 256) {
        bool keep[4];
        keep4(q, offset, k0, k1, thresh, keep);
        const unsigned long long i = q * 4;
        if (i + 4 <= n) {
            float4 v = *reinterpret_cast<const float4*>(x + i);
            if (dact_y) {       // backward through leaky_relu as well: times leaky'(pre) = (y > 0 ? 1 : slope)
                const float4 a = *reinterpret_cast<const float4*>(dact_y + i);
                v.x *= a.x > 0.f ? 1.f : slope; v.y *= a.y > 0.f ? 1.f : slope;
                v.z *= a.z > 0.f ? 1.f : slope; v.w *= a.w > 0.f ? 1.f : slope;
            }
            float4 o;
            o.x = keep[0] ? v.x * scale : 0.f; o.y = keep[1] ? v.y * scale : 0.f;
            o.z = keep[2] ? v.z * scale : 0.f; o.w = keep[3] ? v.w * scale : 0.f;
            *reinterpret_cast<float4*>(y + i) = o;
        } else {
            for (int k = 0; k < 4 && i + k < n; ++k)
                y[i + k] = keep[k] ? x[i + k] * scale * (dact_y ? (dact_y[i + k] > 0.f ? 1.f : slope) : 1.f) : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, unsigned long long n,
                                                   float lr, float beta1, float beta2, float eps, int call,
                                                   float weight_decay, const int32_t* __restrict__ guard0,
                                                   const int32_t* __restrict__ guard1, int32_t* __restrict__ applied) {
    // guard words (the sticky error words of the step's sweep workspaces, or the error flag that travelled through the
    // gradient all-reduce): a sweep that gave up on a bounded wait -- on ANY rank -- left invalid gradients behind: the
    // update is skipped, parameters and moments stay as they were (uniform branch)
    const bool skip = (guard0 && *guard0 != 0) || (guard1 && *guard1 != 0);
    // bias correction counts the updates that were APPLIED, not the calls: applied[(call-1)&1] = updates before this
    // call, applied[call&1] := updates after it (ping-pong words: no block reads the word another block writes)
    int eff = call;
    if (applied) {
        const int before = applied[(call - 1) & 1];
        eff = before + 1;
        if (blockIdx.x == 0 && threadIdx.x == 0) applied[call & 1] = skip ? before : eff;
    }
    if (skip) return;
    const float bc1 = 1.f - powf(beta1, (float)eff);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)eff));
    // torch.optim.Adam semantics: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
    // p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
    const float step = lr / bc1;
    for (unsigned long long i = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n;
         i += (unsigned long long)gridDim.x * 256 * 4) {
        if (i + 4 <= n) {
            float4 pp = *reinterpret_cast<float4*>(p + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<float4*>(m + i);
            float4 vv = *reinterpret_cast<float4*>(v + i);
            float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = G[k] + weight_decay * P[k];
                M[k] = beta1 * M[k] + (1.f - beta1) * gk;
                V[k] = beta2 * V[k] + (1.f - beta2) * gk * gk;
                P[k] -= step * M[k] / (sqrtf(V[k]) / bc2_sqrt + eps);
            }
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (unsigned long long k = i; k < n; ++k) {
                const float gk = g[k] + weight_decay * p[k];
                m[k] = beta1 * m[k] + (1.f - beta1) * gk;
                v[k] = beta2 * v[k] + (1.f - beta2) * gk * gk;
                p[k] -= step * m[k] / (sqrtf(v[k]) / bc2_sqrt + eps);
            }
        }
    }
}

// Batch hand-over (model.py:227-230, `.to(device)`): a few workgroups stream a pinned, device-mapped HOST buffer into
// HBM with 16-byte loads, 8 in flight per lane.  Unlike hipMemcpyAsync (measured here: the call returns only when the
// DMA has run, so a copy queued behind an event costs the host its lead over the GPU and every launch latency of the
// next step is exposed: 9.8 -> 13.9 ms per step) the launch is asynchronous, and on a stream of its own it runs beside
// the step on CUs the sweeps leave idle.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(const v4u* __restrict__ src, v4u* __restrict__ dst,
                                                          unsigned long long n16, const unsigned char* __restrict__ src_tail,
                                                          unsigned char* __restrict__ dst_tail, unsigned tail) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        v4u v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[i + k * stride] = v[k];
    }
    for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
    if (blockIdx.x == 0 && threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

}  // namespace

extern "C" int pgasr_stream_copy(const void* src, void* dst, unsigned long long bytes, int workgroups, void* stream) {
    if (!src || !dst || bytes == 0 || workgroups <= 0 || workgroups > 1024) return PGASR_ERR_INVALID_ARG;
    if ((((size_t)src) | ((size_t)dst)) & 15) return PGASR_ERR_INVALID_ARG;
    const unsigned long long n16 = bytes / 16;
    const unsigned tail = (unsigned)(bytes % 16);
    // The copy's workgroups must not share a CU with anybody's dependent chain (a sweep member, a lattice workgroup): a CU that
    // also serves eight 16-byte host reads per lane slows its other tenant, and a sweep is as slow as its slowest member.  An LDS
    // reservation nobody else leaves room for makes the dispatcher pick an otherwise idle CU (PGASR_COPY_LDS=0: no reservation).
    static const int copy_lds = [] {
        const char* e = getenv("PGASR_COPY_LDS");
        int want = e ? atoi(e) : 156 * 1024, dev = 0, cap = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && cap > 0 && want > cap)
            want = cap;          // never more than the device gives one workgroup
        return want > 0 ? want : 0;
    }();
    if (copy_lds > 0 && hipFuncSetAttribute((const void*)stream_copy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, copy_lds) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    PGASR_LAUNCH_KERNEL(stream_copy_kernel, dim3(workgroups), dim3(256), (size_t)(copy_lds > 0 ? copy_lds : 0), (hipStream_t)stream, (const v4u*)src, (v4u*)dst, n16,
                       (const unsigned char*)src + n16 * 16, (unsigned char*)dst + n16 * 16, tail);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_dropout(const float* x, float* y, unsigned long long n, float p, uint64_t seed, uint32_t offset,
                             const float* dact_y, float slope, void* stream) {
    if (!x || !y || n == 0 || !(p >= 0.f) || !(p < 1.f)) return PGASR_ERR_INVALID_ARG;
    if ((((size_t)x) | ((size_t)y) | ((size_t)dact_y)) & 15) return PGASR_ERR_INVALID_ARG;
    const uint32_t thresh = (uint32_t)fmin(4294967295.0, (double)p * 4294967296.0);
    const unsigned long long nq = (n + 3) / 4;
    unsigned blocks = (unsigned)((nq + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    PGASR_LAUNCH_KERNEL(dropout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n, thresh,
                       1.f / (1.f - p), (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), offset, dact_y, slope);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

// out[0] = 1.0f if one of the (up to two) sweep error words is set, else 0.0f: the flag a rank puts into word 0 of its gradient buffer
// before the last bucket's all-reduce (train_step.DataParallelStep) -- one launch instead of four tiny tensor ops on the critical stream
namespace { __global__ void error_flag_kernel(const int* w0, const int* w1, float* out) {
    if (threadIdx.x == 0) out[0] = ((w0 && *w0 != 0) || (w1 && *w1 != 0)) ? 1.f : 0.f;
} }
extern "C" int pgasr_error_flag(const int32_t* word0, const int32_t* word1, float* out, void* stream) {
    if (!out) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(error_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)word0, (const int*)word1, out);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, unsigned long long n,
                               int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                               const int32_t* guard0, const int32_t* guard1, int32_t* applied, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n == 0 || step < 1) return PGASR_ERR_INVALID_ARG;
    if ((((size_t)param) | ((size_t)grad) | ((size_t)exp_avg) | ((size_t)exp_avg_sq)) & 15) return PGASR_ERR_INVALID_ARG;
    if ((((size_t)guard0) | ((size_t)guard1) | ((size_t)applied)) & 3) return PGASR_ERR_INVALID_ARG;
    unsigned blocks = (unsigned)((n / 4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    PGASR_LAUNCH_KERNEL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       lr, beta1, beta2, eps, step, weight_decay, guard0, guard1, applied);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}
