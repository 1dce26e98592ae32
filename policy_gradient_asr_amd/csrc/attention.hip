// The reference's attention context (model.py:58-94, Attention.forward) AS EXECUTED -- SURVEY.md section 8f N4.  Per query q (a decoder
// state d of utterance b = q % B) and encoder frame i the reference forms the (H,H) outer product exp(d[r] * e_i[k]), divides it by
// its row sums lined up with the LAST axis (model.py:73's broadcast: entry [r,k] / rowsum[k], a recorded defect) and accumulates
// a * e_i over the frames, then sums over r:
//     ctx[q,k] = sum_i e[b,i,k] * S1 / S2,    S1 = sum_r exp(d[r] e[b,i,k]),    S2 = sum_k' exp(d[k] e[b,i,k'])
// One workgroup per (query, 256-column tile); thread = column k; d and the current frame live in LDS (broadcast reads), both sums
// run over H terms with the exact maximum of their exponents taken out (for S1 the maximum over r of d[r] e_k is e_k * dmax or
// e_k * dmin, for S2 d_k * emax_i or d_k * emin_i), so nothing overflows where the reference's own exp stays finite and the
// result is the same quotient.  2 H exponentials per thread and frame: a defined function for parity, not a hot kernel.
#include "common.h"

namespace {
constexpr int ATT_THREADS = 256;

__global__ __launch_bounds__(ATT_THREADS) void attention_ctx_kernel(const float* __restrict__ dec, const float* __restrict__ enc,
                                                                    int B, int T, int H, float* __restrict__ ctx) {
    extern __shared__ float sm[];            // d[H] | e[H] | red[2 * 4]
    float* sd = sm; float* se = sm + H; float* red = sm + 2 * H;
    const int q = blockIdx.x, b = q % B, k = blockIdx.y * ATT_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* d = dec + (size_t)q * H;
    float mx = -INFINITY, mn = INFINITY;
    for (int r = threadIdx.x; r < H; r += ATT_THREADS) { const float v = d[r]; sd[r] = v; mx = fmaxf(mx, v); mn = fminf(mn, v); }
    mx = wave_max(mx); mn = -wave_max(-mn);
    if (lane == 0) { red[w] = mx; red[4 + w] = mn; }
    __syncthreads();
    const float dmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), dmin = fminf(fminf(red[4], red[5]), fminf(red[6], red[7]));
    const float dk = k < H ? sd[k] : 0.f;
    float acc = 0.f;
    for (int i = 0; i < T; ++i) {
        __syncthreads();                      // everybody is done with the previous frame (and with red)
        const float* e = enc + ((size_t)b * T + i) * H;
        float ex = -INFINITY, en = INFINITY;
        for (int r = threadIdx.x; r < H; r += ATT_THREADS) { const float v = e[r]; se[r] = v; ex = fmaxf(ex, v); en = fminf(en, v); }
        ex = wave_max(ex); en = -wave_max(-en);
        if (lane == 0) { red[w] = ex; red[4 + w] = en; }
        __syncthreads();
        const float emax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), emin = fminf(fminf(red[4], red[5]), fminf(red[6], red[7]));
        if (k < H) {
            const float ek = se[k];
            const float m1 = ek > 0.f ? ek * dmax : ek * dmin;       // max over r of d[r] * ek
            const float m2 = dk > 0.f ? dk * emax : dk * emin;       // max over k' of dk * e[k']
            float s1 = 0.f, s2 = 0.f;
            for (int r = 0; r < H; ++r) {
                s1 += __expf(sd[r] * ek - m1);
                s2 += __expf(dk * se[r] - m2);
            }
            acc += ek * __expf(m1 - m2) * s1 / s2;
        }
    }
    if (k < H) ctx[(size_t)q * H + k] = acc;
}

// One step of the decoder's LSTM cell (model.py:104,111: nn.LSTM(128 -> hidden), gate order i, f, g, o): the recurrent product
// gh = h W_hh^T comes from the exact fp32 GEMM, xp = x_t W_ih^T + b_ih + b_hh was made for all steps at once.  Thread = (b, j).
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ gh, const float* __restrict__ xp,
                                                        float* __restrict__ c, float* __restrict__ h, float* __restrict__ h_out,
                                                        int B, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * H) return;
    const int b = idx / H, j = idx - b * H;
    const size_t g0 = (size_t)b * 4 * H + j;
    const float gi = gh[g0] + xp[g0], gf = gh[g0 + H] + xp[g0 + H], gg = gh[g0 + 2 * H] + xp[g0 + 2 * H], go = gh[g0 + 3 * H] + xp[g0 + 3 * H];
    const float i = 1.f / (1.f + expf(-gi)), f = 1.f / (1.f + expf(-gf)), o = 1.f / (1.f + expf(-go));
    const float cn = f * c[idx] + i * tanhf(gg);
    const float hn = o * tanhf(cn);
    c[idx] = cn; h[idx] = hn;
    if (h_out) h_out[idx] = hn;
}
}  // namespace

extern "C" int pgasr_lstm_cell_f32(const float* gh, const float* xp, float* c, float* h, float* h_out, int B, int H, void* stream) {
    if (!gh || !xp || !c || !h || B <= 0 || H <= 0) return PGASR_ERR_INVALID_ARG;
    if ((long long)B * H > (1ll << 30)) return PGASR_ERR_UNSUPPORTED;
    PGASR_LAUNCH_KERNEL(lstm_cell_kernel, dim3((unsigned)((B * H + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gh, xp, c, h, h_out, B, H);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_attention_ctx(const float* dec, const float* enc, int NQ, int B, int T, int H, float* ctx, void* stream) {
    if (!dec || !enc || !ctx || NQ <= 0 || B <= 0 || T <= 0 || H <= 0 || (NQ % B)) return PGASR_ERR_INVALID_ARG;
    if (H > 8192) return PGASR_ERR_UNSUPPORTED;
    const size_t lds = (size_t)(2 * H + 8) * sizeof(float);
    PGASR_LAUNCH_KERNEL(attention_ctx_kernel, dim3((unsigned)NQ, (unsigned)((H + ATT_THREADS - 1) / ATT_THREADS)), dim3(ATT_THREADS), lds,
                        (hipStream_t)stream, dec, enc, B, T, H, ctx);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}
