// Feature front end of the reference (data.py:44-79): per utterance
//   MFCC(40) of torchaudio's defaults (16 kHz, n_fft 400, hop 200, periodic Hann, reflect-centred frames, power
//   spectrum, 128 HTK mel bands, 10*log10 floored at max-80 dB, orthonormal DCT-II) + delta + delta-delta
//   (5-tap regression, replicate padding), stacked to (120, T) and zero padded to the batch's longest utterance.
// The three contractions (DFT as a 400x402 real matrix, mel bank 201x128, DCT 128x40) are GEMMs and go through
// pgasr_gemm_f32 (exact fp32 MFMA) from the host; this file holds what is not a GEMM: framing + window, |X|^2, the
// dB map with its per-utterance floor, and the delta filters with the final (B, 120, Tmax) layout + mask.
#include "common.h"

namespace {

constexpr int N_FFT = 400, HOP = 200, N_BINS = N_FFT / 2 + 1;

// frames[(b*Tmax + t)][n] = x_b[reflect(t*HOP - N_FFT/2 + n)] * hann[n]   (0 for t >= n_frames[b])
__global__ __launch_bounds__(256) void feat_frames_kernel(const float* __restrict__ wave, const int32_t* __restrict__ n_samples,
                                                          const int32_t* __restrict__ n_frames, long long wave_stride,
                                                          int Tmax, float* __restrict__ frames) {
    const long long row = blockIdx.x;
    const int b = (int)(row / Tmax), t = (int)(row % Tmax);
    const int N = n_samples[b];
    const bool live = t < n_frames[b];
    const float* x = wave + (size_t)b * wave_stride;
    for (int n = threadIdx.x; n < N_FFT; n += 256) {
        float v = 0.f;
        if (live) {
            int i = t * HOP - N_FFT / 2 + n;
            if (i < 0) i = -i;                       // reflect without repeating the edge sample
            if (i >= N) i = 2 * (N - 1) - i;
            const float w = 0.5f - 0.5f * cosf(6.283185307179586f * (float)n / (float)N_FFT);
            v = x[i] * w;
        }
        frames[row * N_FFT + n] = v;
    }
}

// spec (rows, 2*N_BINS) = [re | im]  ->  power (rows, N_BINS)
__global__ __launch_bounds__(256) void feat_power_kernel(const float* __restrict__ spec, long long rows, float* __restrict__ power) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * N_BINS) return;
    const long long r = i / N_BINS; const int k = (int)(i % N_BINS);
    const float re = spec[r * (2 * N_BINS) + k], im = spec[r * (2 * N_BINS) + N_BINS + k];
    power[i] = re * re + im * im;
}

// in place: x = 10*log10(max(x, 1e-10)); then x = max(x, max_over_utterance(x) - top_db).  One workgroup per utterance.
__global__ __launch_bounds__(1024) void feat_db_kernel(float* __restrict__ mel, const int32_t* __restrict__ n_frames, int Tmax,
                                                       int n_mels, float top_db) {
    __shared__ float red[1024];
    const int b = blockIdx.x;
    float* x = mel + (size_t)b * Tmax * n_mels;
    const long long n = (long long)n_frames[b] * n_mels;
    float mx = -INFINITY;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        const float v = 10.f * log10f(fmaxf(x[i], 1e-10f));
        x[i] = v;
        mx = fmaxf(mx, v);
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    const float floor_db = red[0] - top_db;
    for (long long i = threadIdx.x; i < n; i += 1024) x[i] = fmaxf(x[i], floor_db);
}

// mfcc (B, Tmax, C) -> feat (B, 3C, Tmax) = [mfcc | delta | delta-delta] (time contiguous, 0 past the length), mask (B,1,Tmax)
__global__ __launch_bounds__(256) void feat_deltas_kernel(const float* __restrict__ mfcc, const int32_t* __restrict__ n_frames,
                                                          int Tmax, int C, float* __restrict__ feat, float* __restrict__ fmask) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (t >= Tmax) return;
    const int T = n_frames[b];
    float* out = feat + ((size_t)b * 3 * C) * Tmax;
    if (c == 0 && fmask) fmask[(size_t)b * Tmax + t] = t < T ? 1.f : 0.f;
    if (t >= T) {
        out[(size_t)c * Tmax + t] = 0.f; out[(size_t)(C + c) * Tmax + t] = 0.f; out[(size_t)(2 * C + c) * Tmax + t] = 0.f;
        return;
    }
    const float* x = mfcc + (size_t)b * Tmax * C + c;
    auto X = [&](int u) { u = u < 0 ? 0 : (u >= T ? T - 1 : u); return x[(size_t)u * C]; };
    auto D1 = [&](int u) {       // delta at the (clamped) frame u
        u = u < 0 ? 0 : (u >= T ? T - 1 : u);
        return (-2.f * X(u - 2) - X(u - 1) + X(u + 1) + 2.f * X(u + 2)) / 10.f;
    };
    out[(size_t)c * Tmax + t] = x[(size_t)t * C];
    out[(size_t)(C + c) * Tmax + t] = D1(t);
    out[(size_t)(2 * C + c) * Tmax + t] = (-2.f * D1(t - 2) - D1(t - 1) + D1(t + 1) + 2.f * D1(t + 2)) / 10.f;
}

// x (B, Tmax, C) -> feat (B, C, Tmax) (time contiguous, 0 past the length), mask (B,1,Tmax): the layout step of a front end WITHOUT
// deltas (the 80-band log-mel features of the benchmark's F = 80)
__global__ __launch_bounds__(256) void feat_stack_kernel(const float* __restrict__ x, const int32_t* __restrict__ n_frames,
                                                         int Tmax, int C, float* __restrict__ feat, float* __restrict__ fmask) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (t >= Tmax) return;
    const int T = n_frames[b];
    if (c == 0 && fmask) fmask[(size_t)b * Tmax + t] = t < T ? 1.f : 0.f;
    feat[((size_t)b * C + c) * Tmax + t] = t < T ? x[((size_t)b * Tmax + t) * C + c] : 0.f;
}

}  // namespace

extern "C" int pgasr_feat_stack(const float* x, const int32_t* n_frames, int B, int Tmax, int C, float* feat, float* fmask, void* stream) {
    if (!x || !n_frames || !feat || B <= 0 || Tmax <= 0 || C <= 0 || C > 65535 || B > 65535) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(feat_stack_kernel, dim3((Tmax + 255) / 256, C, B), dim3(256), 0, (hipStream_t)stream, x, n_frames, Tmax, C, feat, fmask);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_feat_frames(const float* wave, const int32_t* n_samples, const int32_t* n_frames, int B,
                                 long long wave_stride, int Tmax, float* frames, void* stream) {
    if (!wave || !n_samples || !n_frames || !frames || B <= 0 || Tmax <= 0 || wave_stride <= 0) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(feat_frames_kernel, dim3((unsigned)((long long)B * Tmax)), dim3(256), 0, (hipStream_t)stream,
                       wave, n_samples, n_frames, wave_stride, Tmax, frames);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_feat_power(const float* spec, long long rows, float* power, void* stream) {
    if (!spec || !power || rows <= 0) return PGASR_ERR_INVALID_ARG;
    const long long n = rows * N_BINS;
    PGASR_LAUNCH_KERNEL(feat_power_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, spec, rows, power);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_feat_db(float* mel, const int32_t* n_frames, int B, int Tmax, int n_mels, float top_db, void* stream) {
    if (!mel || !n_frames || B <= 0 || Tmax <= 0 || n_mels <= 0 || !(top_db > 0.f)) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(feat_db_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, mel, n_frames, Tmax, n_mels, top_db);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_feat_deltas_stack(const float* mfcc, const int32_t* n_frames, int B, int Tmax, int n_mfcc,
                                       float* feat, float* fmask, void* stream) {
    if (!mfcc || !n_frames || !feat || B <= 0 || Tmax <= 0 || n_mfcc <= 0 || n_mfcc > 65535 || B > 65535) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(feat_deltas_kernel, dim3((Tmax + 255) / 256, n_mfcc, B), dim3(256), 0, (hipStream_t)stream,
                       mfcc, n_frames, Tmax, n_mfcc, feat, fmask);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}
