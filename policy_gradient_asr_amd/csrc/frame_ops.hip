// Per-frame label ops on (T,B,V) score tensors for gfx950: argmax / inverse-CDF sample
// (SURVEY.md §8a A9, A12), CTC collapse of frame paths (A9) and the REINFORCE gradient (A12).
// All HBM-bound streaming kernels: one 32-lane half-wave per (t,b) row of V <= 64 scores, rows
// are contiguous so a wave's loads cover one contiguous span.
#include "common.h"

namespace {


// one wave per (t,b) row
__global__ __launch_bounds__(256) void frame_argmax_sample_kernel(
    const float* __restrict__ scores, long long rows, int B, int V, uint32_t k0, uint32_t k1,
    uint32_t offset, int ctr_stride, int ctr_base, int32_t* __restrict__ greedy, int32_t* __restrict__ sample) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float x = (lane < V) ? scores[r * V + lane] : -INFINITY;
    // argmax, first max wins: reduce (value, index) pairs
    float bv = x; int bi = (lane < V) ? lane : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (greedy && lane == 0) greedy[r] = bi;
    if (sample) {
        const float e = (lane < V) ? __expf(x - bv) : 0.f;
        // inclusive prefix sum over lanes in label order
        float c = e;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(c, o, 64);
            if (lane >= o) c += up;
        }
        const float total = __shfl(c, 63, 64);
        uint32_t rnd[4];
        // the draw of frame t of utterance b is addressed by t * ctr_stride + ctr_base + b: with stride = the GLOBAL batch and
        // base = this rank's first utterance, N ranks draw exactly what one process holding the whole batch draws
        // utterances BEYOND the global batch (ctr_base + b >= ctr_stride: the empty utterances a ragged batch is padded with) draw from
        // a disjoint counter domain (third counter word 1), so their addresses never coincide with a real utterance's
        const int bg_ = ctr_base + (int)(r % B);
        const bool outside = bg_ >= ctr_stride;
        const long long ctr = outside ? (long long)r : (r / B) * (long long)ctr_stride + bg_;
        philox4x32_10((uint32_t)ctr, offset, outside ? 1u : 0u, 0u, k0, k1, rnd);
        const float u = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f);
        const float thr = u * total;
        // k = number of labels whose inclusive cdf <= u  (oracle: (cdf <= u).sum())
        const unsigned long long m = __ballot(lane < V && c <= thr);
        int k = __popcll(m);
        if (k > V - 1) k = V - 1;
        if (lane == 0) sample[r] = k;
    }
}

// one workgroup per (path set p, utterance b); ballot-compaction over frames
__global__ __launch_bounds__(256) void ctc_collapse_kernel(
    const int32_t* __restrict__ paths, const int32_t* __restrict__ lengths, int T, int B, int blank,
    int32_t* __restrict__ tokens, int32_t* __restrict__ token_lengths) {
    const int b = blockIdx.x, p = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int32_t* path = paths + (size_t)p * T * B;
    int32_t* out = tokens + ((size_t)p * B + b) * T;
    int Tb = lengths ? lengths[b] : T; Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    __shared__ int wcount[4];
    __shared__ int base;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int t0 = 0; t0 < Tb; t0 += 256) {
        const int t = t0 + tid;
        int k = blank, prev = -1;
        if (t < Tb) {
            k = path[(size_t)t * B + b];
            prev = (t > 0) ? path[(size_t)(t - 1) * B + b] : -1;
        }
        const bool keep = (t < Tb) && (k != blank) && (k != prev);
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[wid] = __popcll(m);
        __syncthreads();
        int woff = base;
        for (int w = 0; w < wid; ++w) woff += wcount[w];
        if (keep) out[woff + before] = k;
        __syncthreads();
        if (tid == 0) base += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    if (tid == 0) token_lengths[(size_t)p * B + b] = base;
}

// one wave per (t,b) row
__global__ __launch_bounds__(256) void reinforce_grad_kernel(
    const float* __restrict__ scores, const int32_t* __restrict__ path, const float* __restrict__ coef,
    const int32_t* __restrict__ lengths, long long rows, int B, int V, int accumulate,
    float* __restrict__ grad) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int t = (int)(r / B), b = (int)(r % B);
    const int Tb = lengths ? lengths[b] : 0x7fffffff;
    if (t >= Tb) { if (!accumulate && lane < V) grad[r * V + lane] = 0.f; return; }
    const float x = (lane < V) ? scores[r * V + lane] : -INFINITY;
    const float mx = wave_max(x);
    const float e = (lane < V) ? __expf(x - mx) : 0.f;
    const float tot = wave_sum(e);
    const int k = path[r];
    const float g = coef[b] * (e / tot - (lane == k ? 1.f : 0.f));
    if (lane < V) {
        if (accumulate) grad[r * V + lane] += g; else grad[r * V + lane] = g;
    }
}

// one wave per row: log_softmax over V <= 64 labels
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x, long long rows, int V,
                                                               float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float v = (lane < V) ? x[r * V + lane] : -INFINITY;
    const float mx = wave_max(v);
    const float e = (lane < V) ? __expf(v - mx) : 0.f;
    const float lse = mx + __logf(wave_sum(e));
    if (lane < V) y[r * V + lane] = v - lse;
}

// Rewards of the greedy (baseline) and sampled paths from their edit distances, and the per-utterance
// coefficients the fused CTC + REINFORCE gradient takes (policy_grad.py:4-16 intent):
//   R = -ED / max(L,1);  pg_coef = lam/Bg * (R_s - R_g);  utt_scale = 1 / (Bg * max(L,1))
__global__ __launch_bounds__(256) void pg_rewards_kernel(const int32_t* __restrict__ dist, const int32_t* __restrict__ tg_len,
                                                         int B, float lam, float inv_bg, float* __restrict__ R_g,
                                                         float* __restrict__ R_s, float* __restrict__ coef,
                                                         float* __restrict__ utt_scale) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int L = tg_len[b];
    const float Lf = (float)(L > 1 ? L : 1);
    const float rg = -(float)dist[b] / Lf, rs = -(float)dist[B + b] / Lf;
    R_g[b] = rg; R_s[b] = rs;
    coef[b] = (lam * inv_bg) * (rs - rg);
    utt_scale[b] = inv_bg / Lf;
}

// Value of the objective per utterance: nll_b * utt_scale_b - coef_b * sum_{t < T_b} log p(path[t,b]).
// One workgroup per utterance, fixed-order reduction (deterministic).
__global__ __launch_bounds__(256) void pg_loss_value_kernel(const float* __restrict__ lp, const int32_t* __restrict__ path,
                                                            const int32_t* __restrict__ in_len, const float* __restrict__ nll,
                                                            const float* __restrict__ utt_scale, const float* __restrict__ coef,
                                                            int T, int B, int V, int coef_per_frame,
                                                            float* __restrict__ terms) {
    __shared__ float red[256];
    const int b = blockIdx.x;
    const int Tb = min(in_len[b], T);
    float s = 0.f;
    if (path && coef)
        for (int t = threadIdx.x; t < Tb; t += 256) {
            const int k = path[(size_t)t * B + b];
            const float v = lp[((size_t)t * B + b) * V + k];
            s += coef_per_frame ? coef[(size_t)t * B + b] * v : v;
        }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) terms[b] = nll[b] * utt_scale[b] - (coef ? (coef_per_frame ? red[0] : coef[b] * red[0]) : 0.f);
}

// Per-frame REINFORCE coefficients from the per-step rewards of policy_grad.py:10-15 (reward_mode "per_step").
// rho_j = ED(y, yhat[:j-1]) - ED(y, yhat[:j]) is what character j of the collapsed path earns (the reference's r_t in these terms:
// r_1 = rho_1 + rho_2, r_t = rho_{t+1} for t >= 2; they telescope to |y| - ED(y, yhat)).  The reward-to-go of frame t is the sum over
// the characters that START at frames >= t:  G(t) = ED(y, yhat[:c(t)]) - ED(y, yhat),  c(t) = characters started in frames < t.
//   coef[t,b] = lam/Bg * (G_sample(t) - G_greedy(t)) / max(|y|,1)      (0 for t >= T_b)
// the greedy path's reward-to-go at the same frame is the baseline (it does not depend on the sampled action); frame 0 carries the
// utterance-level coefficient lam/Bg (R_s - R_g) of pg_rewards_kernel.  One wave per utterance, the character count by ballots.
__global__ __launch_bounds__(64) void pg_step_coef_kernel(const int32_t* __restrict__ paths, const int32_t* __restrict__ in_len,
                                                          const int32_t* __restrict__ prefix, int pstride,
                                                          const int32_t* __restrict__ tok_len, const int32_t* __restrict__ tg_len,
                                                          int T, int B, int blank, float lam, float inv_bg, float* __restrict__ coef) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int Tb = in_len[b]; Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    const int L = tg_len[b];
    const float k = lam * inv_bg / (float)(L > 1 ? L : 1);
    const int32_t* pd[2] = {prefix + (size_t)b * pstride, prefix + (size_t)(B + b) * pstride};
    int n[2] = {tok_len[b], tok_len[B + b]};
    int carry[2] = {0, 0};
    for (int w = 0; w < 2; ++w) n[w] = n[w] < 0 ? 0 : (n[w] > pstride - 1 ? pstride - 1 : n[w]);
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        float G[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int32_t* p = paths + (size_t)w * T * B;
            const int cur = t < Tb ? p[(size_t)t * B + b] : blank;
            const int prev = (t > 0 && t < Tb) ? p[(size_t)(t - 1) * B + b] : blank;
            const bool start = t < Tb && cur != blank && (t == 0 || cur != prev);
            const unsigned long long m = __ballot(start);
            int c = carry[w] + __popcll(m & ((1ull << lane) - 1ull));
            carry[w] += __popcll(m);
            c = c > n[w] ? n[w] : c;
            G[w] = (float)(pd[w][c] - pd[w][n[w]]);
        }
        if (t < T) coef[(size_t)t * B + b] = t < Tb ? k * (G[1] - G[0]) : 0.f;
    }
}

}  // namespace

extern "C" int pgasr_log_softmax_rows(const float* logits, long long rows, int V, float* log_probs, void* stream) {
    if (!logits || !log_probs || rows <= 0 || V <= 0) return PGASR_ERR_INVALID_ARG;
    if (V > 64) return PGASR_ERR_UNSUPPORTED;
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    PGASR_LAUNCH_KERNEL(log_softmax_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, rows, V, log_probs);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_frame_argmax_sample(const float* scores, int T, int B, int V,
                                         uint64_t seed, uint32_t offset, int ctr_stride, int ctr_base,
                                         int32_t* greedy_path, int32_t* sample_path, void* stream) {
    if (!scores || T <= 0 || B <= 0 || V <= 0) return PGASR_ERR_INVALID_ARG;
    if (ctr_stride == 0) ctr_stride = B;              // the single-process layout: counter = t * B + b
    if (ctr_stride <= 0 || ctr_base < 0 || ctr_base >= ctr_stride) return PGASR_ERR_INVALID_ARG;
    if (V > 64) return PGASR_ERR_UNSUPPORTED;
    if (!greedy_path && !sample_path) return PGASR_OK;
    const long long rows = (long long)T * B;
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    PGASR_LAUNCH_KERNEL(frame_argmax_sample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       scores, rows, B, V, (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), offset,
                       ctr_stride, ctr_base, greedy_path, sample_path);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

namespace {
// model.py:227-230's batch, made ready for the kernels in ONE launch: frame counts from the feature mask (model.py:52), label
// counts from the target mask (data.py:101), targets as int32.  (As five torch ops these were ~30 us of launches in front of
// every step's first kernel.)
__global__ __launch_bounds__(256) void batch_prep_kernel(const float* __restrict__ fmask, int T, const long long* __restrict__ tmask,
                                                         const long long* __restrict__ targets, int L, int32_t* __restrict__ in_len,
                                                         int32_t* __restrict__ tg_len, int32_t* __restrict__ targets32) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float fs = 0.f; int ts = 0;
    for (int t = tid; t < T; t += 256) fs += fmask[(size_t)b * T + t];
    for (int i = tid; i < L; i += 256) {
        ts += (int)tmask[(size_t)b * L + i];
        targets32[(size_t)b * L + i] = (int32_t)targets[(size_t)b * L + i];
    }
    fs = wave_sum(fs);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ts += __shfl_xor(ts, o, 64);
    __shared__ float sf[4]; __shared__ int si[4];
    if (lane == 0) { sf[wid] = fs; si[wid] = ts; }
    __syncthreads();
    if (tid == 0) {
        in_len[b] = (int32_t)((sf[0] + sf[1]) + (sf[2] + sf[3]));       // mask entries are 0/1: exact in fp32 up to 2^24 frames
        tg_len[b] = (si[0] + si[1]) + (si[2] + si[3]);
    }
}
}  // namespace

extern "C" int pgasr_batch_prep(const float* fmask, int B, int T, const long long* tmask, const long long* targets, int L,
                                int32_t* in_len, int32_t* tg_len, int32_t* targets32, void* stream) {
    if (!fmask || !tmask || !targets || !in_len || !tg_len || !targets32 || B <= 0 || T <= 0 || L <= 0) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(batch_prep_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, fmask, T, tmask, targets, L, in_len, tg_len, targets32);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_ctc_collapse(const int32_t* paths, const int32_t* lengths, int P, int T, int B,
                                  int blank, int32_t* tokens, int32_t* token_lengths, void* stream) {
    if (!paths || !tokens || !token_lengths || P <= 0 || T <= 0 || B <= 0) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(ctc_collapse_kernel, dim3(B, P), dim3(256), 0, (hipStream_t)stream,
                       paths, lengths, T, B, blank, tokens, token_lengths);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_reinforce_grad(const float* scores, const int32_t* path, const float* coef,
                                    const int32_t* lengths, int T, int B, int V, int accumulate,
                                    float* grad, void* stream) {
    if (!scores || !path || !coef || !grad || T <= 0 || B <= 0 || V <= 0) return PGASR_ERR_INVALID_ARG;
    if (V > 64) return PGASR_ERR_UNSUPPORTED;
    const long long rows = (long long)T * B;
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    PGASR_LAUNCH_KERNEL(reinforce_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       scores, path, coef, lengths, rows, B, V, accumulate, grad);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_pg_rewards(const int32_t* dist, const int32_t* target_lengths, int B, float lam,
                                float inv_global_batch, float* R_greedy, float* R_sample, float* pg_coef,
                                float* utt_scale, void* stream) {
    if (!dist || !target_lengths || !R_greedy || !R_sample || !pg_coef || !utt_scale || B <= 0) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(pg_rewards_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       dist, target_lengths, B, lam, inv_global_batch, R_greedy, R_sample, pg_coef, utt_scale);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_pg_loss_value(const float* log_probs, const int32_t* path, const int32_t* input_lengths,
                                   const float* nll, const float* utt_scale, const float* pg_coef,
                                   int T, int B, int V, int pg_coef_per_frame, float* terms, void* stream) {
    if (!log_probs || !input_lengths || !nll || !utt_scale || !terms || T <= 0 || B <= 0 || V <= 0) return PGASR_ERR_INVALID_ARG;
    if ((pg_coef == nullptr) != (path == nullptr)) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(pg_loss_value_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream,
                       log_probs, path, input_lengths, nll, utt_scale, pg_coef, T, B, V, pg_coef_per_frame ? 1 : 0, terms);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_pg_step_coefs(const int32_t* paths, const int32_t* input_lengths, const int32_t* prefix_dist, int prefix_stride,
                                   const int32_t* token_lengths, const int32_t* target_lengths, int T, int B, int blank,
                                   float lam, float inv_global_batch, float* coef, void* stream) {
    if (!paths || !input_lengths || !prefix_dist || !token_lengths || !target_lengths || !coef) return PGASR_ERR_INVALID_ARG;
    if (T <= 0 || B <= 0 || prefix_stride < 1 || blank < 0) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(pg_step_coef_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream,
                       paths, input_lengths, prefix_dist, prefix_stride, token_lengths, target_lengths, T, B, blank, lam,
                       inv_global_batch, coef);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}
