// CTC loss and gradient for gfx950.
//
// Two launches:
//   ctc_lattice_kernel : grid (B,3).  y=0 sweeps alpha forward in time, y=1 sweeps beta
//       backward, y=2 builds the label->states lists.  One workgroup per utterance and
//       direction, one lattice state per thread (strided when S > 256), one barrier per
//       frame, the previous row double-buffered in LDS.  The recursion is a serial chain of
//       T dependent steps, so this kernel is latency bound, not HBM bound.
//   ctc_grad_kernel    : one wave per (t,b): posterior occupancy per label from alpha+beta,
//       grad = scale_b * (softmax - occupancy) [+ REINFORCE term], fully parallel over T*B.
//
// Measured and NOT kept (round 3): a barrier-free lattice for S <= 256 -- one state per lane, neighbours by DPP wave_shr:1 /
// wave_shl:1, the four compute waves as a pipeline that hands only its two edge states per frame through an LDS ring with
// frame counters (prefetched a frame ahead), the storer following the counters four frames at a time.  Bit-identical, and
// no faster: 268-298 us against 272.  Switching its parts off one by one (diagnostic flags) showed why: without the
// neighbour exchange 278 us, without the storer 280, with the emissions prefetched 16 frames ahead 268, with cached
// emissions 283 -- and with ALL of these AND lse3 removed still 239 us, 0.24 us = ~570 cycles per frame for ~60 dependent
// instructions of one wave per SIMD.  The frame is bound by the dependent-issue latency of a single wave's instruction
// chain, not by the barrier, the LDS round trip, the emission loads or the stores; more waves do not shorten a chain.
//
// Measured and NOT kept (round 5, commit cdcbccd): a LINEAR-domain lattice for S <= 256 -- one compute wave per utterance and direction,
// four adjacent states per lane, alpha_t(s) = p_t(s) (alpha_{t-1}(s) + alpha_{t-1}(s-1) + [skip] alpha_{t-1}(s-2)) as two additions and a
// multiplication per state instead of an lse3, neighbours by DPP wave_shr / wave_shl, block floating point with one binary exponent per
// lane (re-based on the neighbour's when that is far above), a feeder wave turning log-probs into probabilities 32 frames at a time, three
// storer waves writing the same workspace format, a redo in log space when fp32's range empties the lattice.  Bit-for-bit the same nll to
// eight digits and all CTC tests green -- and 0.31 us per frame in fp64 (fp64 vector instructions issue at half rate), 0.31 in fp32 with
// per-frame renormalisation, 0.36 with exponents frozen for four frames, against 0.27 for the kernel below.  In-kernel clocks: 709 cycles per
// frame at 2.4 GHz for ~45 vector instructions, 490 with every LDS access removed: once more (see the round-3 note above) a frame of ONE
// wave costs the dependent-issue latency of its chain, ~13-20 cycles per dependent instruction, plus an LDS round trip -- shortening the
// arithmetic does not shorten that.  The lattice stays at 0.27 us per frame.
//
// Numerics: alpha/beta are kept in fp64 (adds/max are native fp64 VALU ops) while exp/log
// run in fp32 on the *differences* to the row maximum, which are O(1..50): absolute error
// per step ~1e-7 instead of the ~2e-4 ulp an fp32 log-space value of magnitude 3000 has at
// T=1000.  Sums over states are taken in a fixed order (wave butterfly, then list order), so
// results are run-to-run reproducible.
#include "common.h"

namespace {

constexpr int CTC_THREADS = 256;
constexpr int CTC_SPT = 8;                        // states per thread -> S <= 2048
constexpr int CTC_SMAX = CTC_THREADS * CTC_SPT;   // 2048
constexpr int CTC_VMAX = 64;

struct CtcWs {
    // The lattice leaves the chip as fp32 OFFSETS from the row maximum plus one fp64 maximum per row (round 2): the rows
    // live in fp64 in LDS while the recursion runs (values reach -3000 at T = 1000, where fp32 resolves 2.4e-4), but
    // within a row only states within ~100 of the maximum carry any posterior mass, and there an fp32 offset resolves
    // < 1e-5.  Halves the lattice traffic of the round-1 fp64 spill (229 -> ~105 MB per step at B=32, T=1000, S=201).
    float* alpha;      // [B][T][SP]     alpha_t(s) - amax[t]   (-inf stays -inf); SP = Smax rounded up to 64: the storer
    float* beta;       // [B][T][SP]     beta_t(s)  - bmax[t]      writes whole 64-state groups, no bounds test per state
    int SP;
    double* amax;      // [B][T]         row maxima (0 for a row that is -inf everywhere)
    double* bmax;      // [B][T]
    double* nll64;     // [B]
    int32_t* lab_off;  // [B][V+1]   offsets into lab_states, per label
    int32_t* lab_states;  // [B][Smax] odd (non-blank) states grouped by label, ascending s
};

__host__ __device__ inline size_t ctc_ws_layout(int T, int B, int V, int Smax, CtcWs* ws, char* base) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const int SP = (Smax + 63) / 64 * 64;
    size_t a = take((size_t)B * T * SP * sizeof(float));
    size_t b = take((size_t)B * T * SP * sizeof(float));
    size_t am = take((size_t)B * T * sizeof(double));
    size_t bm = take((size_t)B * T * sizeof(double));
    size_t n = take((size_t)B * sizeof(double));
    size_t lo = take((size_t)B * (V + 1) * sizeof(int32_t));
    size_t ls = take((size_t)B * Smax * sizeof(int32_t));
    if (ws) {
        ws->alpha = (float*)(base + a); ws->beta = (float*)(base + b);
        ws->amax = (double*)(base + am); ws->bmax = (double*)(base + bm); ws->SP = SP;
        ws->nll64 = (double*)(base + n); ws->lab_off = (int32_t*)(base + lo);
        ws->lab_states = (int32_t*)(base + ls);
    }
    return off;
}

// max over the 64 lanes of a wave, in all lanes: DPP row rotations + four readlanes (a butterfly of ds_bpermute pairs costs
// six LDS-crossbar round trips, more than a lattice frame lasts)
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_max_f32_all(float m) {
    m = fmaxf(m, dpp_f32<0x128>(m)); m = fmaxf(m, dpp_f32<0x124>(m));     // row_ror:8, :4
    m = fmaxf(m, dpp_f32<0x122>(m)); m = fmaxf(m, dpp_f32<0x121>(m));     // row_ror:2, :1 -> every lane holds its row's maximum
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// log(exp(a0)+exp(a1)+exp(a2)) with fp64 carries and fp32 transcendentals.
__device__ __forceinline__ double lse3(double a0, double a1, double a2) {
    const double m = fmax(fmax(a0, a1), a2);
    // no branch on the all -inf row (a select at the end instead): every taken branch on the T-step chain refills the
    // instruction buffer.  mz keeps the differences finite-or--inf when m = -inf.
    const double mz = (m == -INFINITY) ? 0.0 : m;
    const float s = __builtin_amdgcn_exp2f(1.4426950408889634f * (float)(a0 - mz)) + __builtin_amdgcn_exp2f(1.4426950408889634f * (float)(a1 - mz)) +
                    __builtin_amdgcn_exp2f(1.4426950408889634f * (float)(a2 - mz));
    // s is in [1, 3]: the bare v_log_f32 (log2) needs none of __logf's denormal / range fix-ups, which sat on the chain
    const double r = m + (double)(0.6931471805599453f * __builtin_amdgcn_logf(s));
    return (m == -INFINITY) ? -INFINITY : r;
}

// label -> states lists for the gradient pass (grid row y = 2 of the lattice launch)
__device__ __forceinline__ void ctc_labels_body(const int32_t* __restrict__ targets, const int32_t* __restrict__ tg_len,
                                                int V, int Lmax, int Smax, int blank, CtcWs ws) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    int Lb = tg_len[b]; Lb = Lb < 0 ? 0 : (Lb > Lmax ? Lmax : Lb);
    const int32_t* tgt = targets + (size_t)b * Lmax;
        // label -> list of odd states carrying it (ascending s).  One thread per label.
        __shared__ int cnt[CTC_VMAX + 1];
        if (tid <= V) cnt[tid] = 0;
        __syncthreads();
        if (tid < V) {
            int c = 0;
            if (tid != blank)
                for (int i = 0; i < Lb; ++i) c += (tgt[i] == tid);
            cnt[tid + 1] = c;
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int v = 0; v <= V; ++v) { run += cnt[v]; cnt[v] = run; }  // cnt[v] = start of label v
            // after the loop cnt[v] holds the inclusive sum up to v; shift handled below
        }
        __syncthreads();
        // cnt[v] now = sum_{u<=v} count_u where count stored at u+1 => cnt[v] = start of label v
        if (tid <= V) ws.lab_off[(size_t)b * (V + 1) + tid] = cnt[tid];
        if (tid < V && tid != blank) {
            int w = cnt[tid];
            for (int i = 0; i < Lb; ++i)
                if (tgt[i] == tid) ws.lab_states[(size_t)b * Smax + (w++)] = 2 * i + 1;
        }
}

// NSPT = states per thread, a compile-time constant: the common case S <= 256 (L <= 127) runs with no per-state
// loop or bound checks on the T-step chain (measured 492 -> see DESIGN.md at T=1000, S=201).
template <int NSPT, int ROLE>
__device__ __forceinline__ void ctc_lattice_body(
    const float* __restrict__ lp, const int32_t* __restrict__ targets,
    const int32_t* __restrict__ in_len, const int32_t* __restrict__ tg_len,
    int T, int B, int V, int Lmax, int Smax, int blank, CtcWs ws, float* __restrict__ nll_out) {
    const int b = blockIdx.x;
    constexpr int role = ROLE;          // compile-time: the frame loop carries no direction test
    const int tid = threadIdx.x;
    int Tb = in_len[b]; Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    int Lb = tg_len[b]; Lb = Lb < 0 ? 0 : (Lb > Lmax ? Lmax : Lb);
    const int S = 2 * Lb + 1;
    const int32_t* tgt = targets + (size_t)b * Lmax;

    // row buffers: position p = s + 2, two guard cells of -inf on each side
    constexpr int ROW = NSPT * CTC_THREADS + 4;
    __shared__ double row[2][ROW];
    // Wave 4 is the STORER: it copies each finished row from LDS to the alpha/beta array.  With the store in the
    // compute threads the chain waited every step for the store's write acknowledgement: hipcc must use vmcnt(0)
    // for the emission prefetch as soon as a store is also outstanding (loads and stores retire out of order
    // with respect to each other) -- 391 us against 492 us before the NSPT template, see DESIGN.md.
    const bool storer = tid >= CTC_THREADS;
    for (int i = tid; i < 2 * ROW; i += CTC_THREADS + 64) (&row[0][0])[i] = -INFINITY;

    // per-thread state descriptors
    int lab[NSPT];
    bool skip[NSPT];   // alpha: may come from s-2 ; beta: may go to s+2
#pragma unroll
    for (int j = 0; j < NSPT; ++j) {
        const int s = storer ? S : tid + j * CTC_THREADS;       // the storer owns no state
        lab[j] = blank; skip[j] = false;
        if (s < S && (s & 1)) lab[j] = tgt[s >> 1];
        if (role == 0) {
            if (s < S && (s & 1) && s >= 3) skip[j] = (tgt[s >> 1] != tgt[(s >> 1) - 1]);
        } else {
            if (s + 2 < S && (s & 1)) skip[j] = (tgt[(s >> 1) + 1] != tgt[s >> 1]);
        }
        if (lab[j] < 0 || lab[j] >= V) lab[j] = blank;  // defensive: never index outside the row
    }
    __syncthreads();

    float* out = (role == 0 ? ws.alpha : ws.beta) + (size_t)b * T * ws.SP;
    double* outmax = (role == 0 ? ws.amax : ws.bmax) + (size_t)b * T;
    // the storer wave's row write: maximum over the row (fixed butterfly order), then fp32 offsets
    double m_ref = 0.0;          // the storer's current reference (refreshed every 4th row)
    auto store_row = [&](const double* rc, int t_row, bool refresh) {
        // Branch-free: the storer lane keeps 64-strided states in registers -- cells past S hold the -inf the rows were
        // initialised with, and the output rows are padded to whole 64-state groups -- so the only waits are the LDS reads.
        // (It sits at the same per-frame barrier as the compute waves: a slower storer would set the frame time.)
        constexpr int NG = NSPT * 4;
        const int ls = tid - CTC_THREADS;
        const int ng = (S + 63) >> 6;           // 64-state groups that hold states (wave-uniform)
        double v[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) v[i] = rc[ls + 64 * i + 2];
        // The reference of a row need not be its maximum, only NEAR it: offsets are formed in fp64 against whatever
        // reference is stored with the row.  It is refreshed every 4th row (a row maximum moves by one frame's log-prob
        // per frame, a few units; even 4 x 88 keeps the fp32 offset's resolution at 3e-5), and the wave reduction runs on
        // fp32 DPP maxima -- so three rows out of four cost the storer four reads, four subtractions and four stores.
        if (refresh) {
            double ml = -INFINITY;
#pragma unroll
            for (int i = 0; i < NG; ++i) ml = fmax(ml, v[i]);
            const double mw = (double)wave_max_f32_all((float)ml);
            m_ref = (mw == -INFINITY) ? 0.0 : mw;
        }
        const double m = m_ref;
        float* o = out + (size_t)t_row * ws.SP + ls;
#pragma unroll
        for (int i = 0; i < NG; ++i)
            if (i < ng) o[64 * i] = (float)(v[i] - m);
        if (ls == 0) outmax[t_row] = m;
    };
    if (Tb == 0) {
        if (role == 0 && tid == 0) {
            const double v = (Lb == 0) ? 0.0 : INFINITY;
            ws.nll64[b] = v; nll_out[b] = (float)v;
        }
        return;
    }

    const int t0 = (role == 0) ? 0 : Tb - 1;
    const int dt = (role == 0) ? 1 : -1;

    // frame t0
    {
        const float* lpt = lp + ((size_t)t0 * B + b) * V;
#pragma unroll
        for (int j = 0; j < NSPT; ++j) {
            const int s = storer ? S : tid + j * CTC_THREADS;
            if (s < S) {
                double v = -INFINITY;
                if (role == 0) { if (s <= 1) v = (double)lpt[lab[j]]; }
                else           { if (s >= S - 2) v = (double)lpt[lab[j]]; }
                row[0][s + 2] = v;
            }
        }
    }
    int cur = 0;
    // LDS-only barrier: __syncthreads() would also drain vmcnt (the storer's stores, the emission prefetch)
#define ROW_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); \
                           asm volatile("" ::: "memory"); } while (0)
    if (storer) {
        // its own loop (same number of barriers): nothing in this path ever waits for a store
        for (int k = 1; k < Tb; ++k) {
            ROW_BARRIER();
            store_row(row[cur], t0 + (k - 1) * dt, ((k - 1) & 3) == 0);      // row[cur] = frame t0 + (k-1)*dt, stable until the next barrier
            cur ^= 1;
        }
    } else {
        float lpn[NSPT];
#pragma unroll
        for (int j = 0; j < NSPT; ++j) lpn[j] = 0.f;
        if (Tb > 1) {
            const float* lpt = lp + ((size_t)(t0 + dt) * B + b) * V;
#pragma unroll
            for (int j = 0; j < NSPT; ++j) lpn[j] = lpt[lab[j]];
        }
        for (int k = 1; k < Tb; ++k) {
            const int t = t0 + k * dt;
            float lpc[NSPT];
#pragma unroll
            for (int j = 0; j < NSPT; ++j) lpc[j] = lpn[j];
            if (k + 1 < Tb) {  // prefetch the next frame's emissions: off the dependent chain
                const float* lpt = lp + ((size_t)(t + dt) * B + b) * V;
#pragma unroll
                for (int j = 0; j < NSPT; ++j) lpn[j] = lpt[lab[j]];
            }
            ROW_BARRIER();
            const double* rc = row[cur];
            double* rn = row[cur ^ 1];
#pragma unroll
            for (int j = 0; j < NSPT; ++j) {
                // no test on the chain: all three neighbours are read (guard cells on both sides), a state that may not
                // skip and a thread past S select -inf
                const int s = tid + j * CTC_THREADS;
                const int p = s + 2;
                const double a0 = rc[p];
                const double a1 = rc[role == 0 ? p - 1 : p + 1];
                const double a2r = rc[role == 0 ? p - 2 : p + 2];
                const double a2 = skip[j] ? a2r : -INFINITY;
                const double v = lse3(a0, a1, a2) + (double)lpc[j];
                rn[p] = (s < S) ? v : -INFINITY;
            }
            cur ^= 1;
        }
    }
#undef ROW_BARRIER
    {                    // the last frame's row (the only one when Tb == 1)
        __syncthreads();
        if (storer) store_row(row[cur], t0 + (Tb - 1) * dt, ((Tb - 1) & 3) == 0);
    }
    if (role == 0) {
        __syncthreads();
        if (tid == 0) {
            const double* rc = row[cur];
            const double ll = lse3(rc[S - 1 + 2], (S > 1) ? rc[S - 2 + 2] : -INFINITY, -INFINITY);
            ws.nll64[b] = -ll;
            nll_out[b] = (float)(-ll);
        }
    }
}

template <int NSPT>
__global__ __launch_bounds__(CTC_THREADS + 64) void ctc_lattice_kernel(
    const float* __restrict__ lp, const int32_t* __restrict__ targets,
    const int32_t* __restrict__ in_len, const int32_t* __restrict__ tg_len,
    int T, int B, int V, int Lmax, int Smax, int blank, CtcWs ws, float* __restrict__ nll_out, int role_base) {
    const int role = blockIdx.y + role_base;      // one uniform branch per workgroup, none per frame
    if (role == 0) ctc_lattice_body<NSPT, 0>(lp, targets, in_len, tg_len, T, B, V, Lmax, Smax, blank, ws, nll_out);
    else if (role == 1) ctc_lattice_body<NSPT, 1>(lp, targets, in_len, tg_len, T, B, V, Lmax, Smax, blank, ws, nll_out);
    else ctc_labels_body(targets, tg_len, V, Lmax, Smax, blank, ws);
}


// one wave per (t,b)
__global__ __launch_bounds__(256) void ctc_grad_kernel(
    const float* __restrict__ lp, const int32_t* __restrict__ in_len, const int32_t* __restrict__ tg_len,
    int T, int B, int V, int Lmax, int Smax, int blank, CtcWs ws,
    const float* __restrict__ utt_scale, const float* __restrict__ pg_coef,
    const int32_t* __restrict__ pg_path, int coef_per_frame, float* __restrict__ grad) {
    const int lane = threadIdx.x & 63;
    const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= (long long)T * B) return;
    const int t = (int)(w / B), b = (int)(w % B);
    int Tb = in_len[b]; Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    int Lb = tg_len[b]; Lb = Lb < 0 ? 0 : (Lb > Lmax ? Lmax : Lb);
    const size_t o = ((size_t)t * B + b) * V;
    if (t >= Tb) { if (lane < V) grad[o + lane] = 0.f; return; }

    const float lpv = (lane < V) ? lp[o + lane] : 0.f;
    const float sm = (lane < V) ? __expf(lpv) : 0.f;
    float g = 0.f;
    const double nll = ws.nll64[b];
    if (nll != INFINITY) {
        const int S = 2 * Lb + 1;
        const float* al = ws.alpha + ((size_t)b * T + t) * ws.SP;
        const float* be = ws.beta + ((size_t)b * T + t) * ws.SP;
        // log occupancy of state s = (alpha offset + beta offset) + [row maxima + nll - log p]: the bracket is O(10)
        const double cst = ws.amax[(size_t)b * T + t] + ws.bmax[(size_t)b * T + t] + nll;
        // blank occupancy: even states, all lanes, fixed butterfly order
        const float lpb = __shfl(lpv, blank, 64);
        const float cb = (float)(cst - (double)lpb);
        float accb = 0.f;
        for (int s = 2 * lane; s < S; s += 128)
            accb += __expf((al[s] + be[s]) + cb);
        accb = wave_sum(accb);
        float occ = accb;
        if (lane < V && lane != blank) {
            const int32_t* lo = ws.lab_off + (size_t)b * (V + 1);
            const int32_t* ls = ws.lab_states + (size_t)b * Smax;
            float acc = 0.f;
            const float cl = (float)(cst - (double)lpv);
            for (int i = lo[lane]; i < lo[lane + 1]; ++i) {
                const int s = ls[i];
                acc += __expf((al[s] + be[s]) + cl);
            }
            occ = acc;
        }
        const float sc = utt_scale ? utt_scale[b] : 1.f;
        g = sc * (sm - occ);
    }
    if (pg_coef != nullptr && pg_path != nullptr) {
        const int k = pg_path[(size_t)t * B + b];
        g += pg_coef[coef_per_frame ? (size_t)t * B + b : (size_t)b] * (sm - (lane == k ? 1.f : 0.f));
    }
    if (lane < V) grad[o + lane] = g;
}

}  // namespace

extern "C" size_t pgasr_ctc_workspace_bytes(int T, int B, int V, int Lmax) {
    if (T <= 0 || B <= 0 || V <= 0 || Lmax < 0) return 0;
    return ctc_ws_layout(T, B, V, 2 * Lmax + 1, nullptr, nullptr);
}

extern "C" int pgasr_ctc_loss_grad(const float* log_probs, const int32_t* targets,
                                   const int32_t* input_lengths, const int32_t* target_lengths,
                                   int T, int B, int V, int Lmax, int blank,
                                   const float* utt_scale, const float* pg_coef, const int32_t* pg_path,
                                   float* nll, float* grad_logits,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!log_probs || !targets || !input_lengths || !target_lengths || !nll) return PGASR_ERR_INVALID_ARG;
    if (T <= 0 || B <= 0 || V <= 0 || Lmax < 0 || blank < 0 || blank >= V) return PGASR_ERR_INVALID_ARG;
    if ((pg_coef == nullptr) != (pg_path == nullptr)) return PGASR_ERR_INVALID_ARG;
    const int Smax = 2 * Lmax + 1;
    if (Smax > CTC_SMAX || V > CTC_VMAX) return PGASR_ERR_UNSUPPORTED;
    CtcWs ws;
    const size_t need = ctc_ws_layout(T, B, V, Smax, &ws, (char*)workspace);
    if (!workspace || workspace_bytes < need) return PGASR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    // Lmax == 0 still needs a valid targets row pointer; Lmax>=1 is the caller's job.
    // (a single-wave register-resident variant was measured SLOWER: 850 us vs 492 us at S=201 --
    // one wave's fp64 VALU issue rate, not the barrier, is then the limit)
#define PGASR_LATTICE(NSPT) PGASR_LAUNCH_KERNEL(ctc_lattice_kernel<NSPT>, dim3(B, 3), dim3(CTC_THREADS + 64), 0, st, \
                        log_probs, targets, input_lengths, target_lengths, T, B, V, Lmax > 0 ? Lmax : 1, Smax, blank, ws, nll, 0)
    if (Smax <= CTC_THREADS) PGASR_LATTICE(1);
    else if (Smax <= 2 * CTC_THREADS) PGASR_LATTICE(2);
    else if (Smax <= 4 * CTC_THREADS) PGASR_LATTICE(4);
    else PGASR_LATTICE(8);
#undef PGASR_LATTICE
    PGASR_CHECK_LAUNCH();
    if (grad_logits) {
        const long long waves = (long long)T * B;
        const int wpb = 4;
        const unsigned blocks = (unsigned)((waves + wpb - 1) / wpb);
        PGASR_LAUNCH_KERNEL(ctc_grad_kernel, dim3(blocks), dim3(64 * wpb), 0, st,
                           log_probs, input_lengths, target_lengths, T, B, V,
                           Lmax > 0 ? Lmax : 1, Smax, blank, ws, utt_scale, pg_coef, pg_path, 0, grad_logits);
        PGASR_CHECK_LAUNCH();
    }
    return PGASR_OK;
}

// Second half of pgasr_ctc_loss_grad on its own: the gradient pass over a lattice that an earlier
// pgasr_ctc_loss_grad(..., grad_logits = NULL, ...) call with the SAME shapes left in `workspace`.
// Lets the host run the lattice (a 64-workgroup serial chain) on one stream beside the sampling /
// decode / edit-distance kernels whose rewards the gradient needs.
extern "C" int pgasr_ctc_grad_from_lattice(const float* log_probs, const int32_t* input_lengths,
                                           const int32_t* target_lengths, int T, int B, int V, int Lmax, int blank,
                                           const float* utt_scale, const float* pg_coef, const int32_t* pg_path,
                                           int pg_coef_per_frame, float* grad_logits, void* workspace, size_t workspace_bytes,
                                           void* stream) {
    if (!log_probs || !input_lengths || !target_lengths || !grad_logits) return PGASR_ERR_INVALID_ARG;
    if (T <= 0 || B <= 0 || V <= 0 || Lmax < 0 || blank < 0 || blank >= V) return PGASR_ERR_INVALID_ARG;
    if ((pg_coef == nullptr) != (pg_path == nullptr)) return PGASR_ERR_INVALID_ARG;
    const int Smax = 2 * Lmax + 1;
    if (Smax > CTC_SMAX || V > CTC_VMAX) return PGASR_ERR_UNSUPPORTED;
    CtcWs ws;
    const size_t need = ctc_ws_layout(T, B, V, Smax, &ws, (char*)workspace);
    if (!workspace || workspace_bytes < need) return PGASR_ERR_WORKSPACE;
    const long long waves = (long long)T * B;
    const int wpb = 4;
    const unsigned blocks = (unsigned)((waves + wpb - 1) / wpb);
    PGASR_LAUNCH_KERNEL(ctc_grad_kernel, dim3(blocks), dim3(64 * wpb), 0, (hipStream_t)stream,
                       log_probs, input_lengths, target_lengths, T, B, V,
                       Lmax > 0 ? Lmax : 1, Smax, blank, ws, utt_scale, pg_coef, pg_path, pg_coef_per_frame ? 1 : 0, grad_logits);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}
