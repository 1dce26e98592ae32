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


// =====================================================================================================================
// Linear-domain lattice for S <= 256 (round 5): ONE compute wave per utterance and direction walks the T frames, four
// adjacent states per lane, in SCALED PROBABILITIES instead of log space:
//     alpha_t(s) = p_t(s) * (alpha_{t-1}(s) + alpha_{t-1}(s-1) + [skip] alpha_{t-1}(s-2))        (beta: mirrored)
// is two additions and one multiplication per state and frame -- against max, three differences, three exp2, a log2 and
// their conversions for a log-space lse3.  Rounds 1-3 found the log-space frame bound by the dependent-issue latency of one
// wave's ~60-instruction chain (0.27 us per frame whatever the barrier, the LDS round trip or the stores cost, see the
// header); the chain of a frame here is neighbour hand-off (DPP wave_shr / wave_shl) -> rescale -> add -> fma -> mul.
// Arithmetic: fp32 values with a binary exponent PER LANE (true value = a * 2^e: block floating point over the lane's four
// adjacent states), renormalised every frame (v_frexp_exp / v_ldexp); the neighbour's exponent travels with its values, and
// a lane whose neighbour is more than 2^32 above it re-bases itself on the neighbour's exponent (what it held is below the
// resolution of what flows in).  So states far from the mass keep their true magnitude across the row -- a whole-row scale
// factor (the textbook scaled forward pass) would flush states 87 nats below the row maximum, which is where the path of a
// badly fitting transcript lies.  (The first version of this kernel kept fp64 values: fp64 vector instructions issue at half
// rate and v_ldexp_f64 slower still -- 0.31 us per frame, slower than the log-space kernel it was to replace.)  Rounding:
// ~1e-7 relative per frame, a random walk of ~5e-6 over 1000 frames in alpha -- the log-space kernel's fp32 exp2 / log2 on
// differences are of the same order.  What fp32 cannot hold is a frame probability below 2^-126 (log-prob < -87): it becomes
// 0.  If that empties the whole lattice (likelihood 0 although the alignment is feasible) the workgroup REDOES its direction
// with the log-space body below; short of that, a path through such a frame weighs e^-87 of the total and is dropped.
// Other waves of the workgroup, none on the chain: a FEEDER turns log-probs into probabilities 32 frames at a time (exp2 of
// the fractional part, the integer part by ldexp), three STORERS turn finished rows into the fp32 log offsets + fp64 row
// reference the gradient pass reads (the same workspace format as the log-space kernel: ctc_grad_kernel does not know which
// lattice kernel ran).  Hand-offs through LDS rings with monotonic counters; every wait is bounded (-> redo in log space).
// =====================================================================================================================
typedef __attribute__((ext_vector_type(4))) float pgasr_f4;
namespace lin {
constexpr int RING = 32;              // finished rows between the compute wave and the storers
constexpr int NSTORE = 3;
constexpr int CH = 32;                // frames per emission chunk
constexpr int WAVES = 2 + NSTORE;     // compute, feeder, storers
constexpr int THREADS = 64 * WAVES;
constexpr int SPIN_MAX = 1 << 22;
constexpr int REBASE = 16;            // a neighbour more than 2^REBASE above this lane: take its exponent
struct Lds {
    float ring_a[RING][64][4];        // 32 KB
    int ring_e[RING][64];             //  8 KB
    float pbuf[2][CH][64];            // 16 KB: probabilities of the frame's symbols, two chunks
    int done, st[NSTORE], fed, cchunk, bad, redo;
};
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
#define LIN_GET(var) __hip_atomic_load(&(var), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LIN_SET(var, val) __hip_atomic_store(&(var), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
// wave-uniform bounded wait for  counter >= want
__device__ __forceinline__ bool wait_ge(int& counter, int want, int& bad) {
    for (int spin = 0; ; ++spin) {
        asm volatile("" ::: "memory");
        if (LIN_GET(counter) >= want) return true;
        if (LIN_GET(bad) || spin > SPIN_MAX) { LIN_SET(bad, 1); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
}
}  // namespace lin

// returns (to every thread of the workgroup) whether the direction has to be redone in log space
template <int ROLE>
__device__ __forceinline__ bool ctc_lattice_lin_body(
    const float* __restrict__ lp, const int32_t* __restrict__ targets, const int32_t* __restrict__ in_len,
    const int32_t* __restrict__ tg_len, int T, int B, int V, int Lmax, int blank, CtcWs ws, float* __restrict__ nll_out, lin::Lds& L, int diag) {
    using namespace lin;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int Tb = in_len[b]; Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    int Lb = tg_len[b]; Lb = Lb < 0 ? 0 : (Lb > Lmax ? Lmax : Lb);
    const int S = 2 * Lb + 1;
    const int32_t* tgt = targets + (size_t)b * Lmax;
    if (tid == 0) {
        LIN_SET(L.done, 0); LIN_SET(L.fed, 0); LIN_SET(L.cchunk, 0); LIN_SET(L.bad, 0); LIN_SET(L.redo, 0);
        for (int i = 0; i < NSTORE; ++i) LIN_SET(L.st[i], 0);
    }
    __syncthreads();
    if (Tb == 0) {
        if (ROLE == 0 && tid == 0) { const double v = (Lb == 0) ? 0.0 : INFINITY; ws.nll64[b] = v; nll_out[b] = (float)v; }
        return false;
    }
    const int t0 = (ROLE == 0) ? 0 : Tb - 1, dt = (ROLE == 0) ? 1 : -1;
    const int nchunk = (Tb + CH - 1) / CH;

    if (w == 1) {
        // ---- feeder: chunk c = sweep steps [32 c, 32 c + 32); lane = symbol ----
        float pre[CH];
        auto issue = [&](int c) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int k = c * CH + i;
                pre[i] = (lane < V && k < Tb) ? lp[((size_t)(t0 + k * dt) * B + b) * V + lane] : -INFINITY;
            }
        };
        issue(0);
        for (int c = 0; c < nchunk; ++c) {
            if (c >= 2 && !wait_ge(L.cchunk, c - 1, L.bad)) break;       // buffer c & 1 held chunk c - 2: the compute wave has moved on to c - 1
            float* dst = &L.pbuf[c & 1][0][0];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                // p = 2^x, x = lp * log2(e) formed in fp64 (an fp32 product would lose 2e-6 at lp = -30): fractional part through
                // v_exp_f32, integer part by ldexp; below 2^-126 the result is 0 (see the header)
                const double x = (double)pre[i] * 1.4426950408889634;
                const double xi = floor(x);
                const float fr = __builtin_amdgcn_exp2f((float)(x - xi));
                dst[i * 64 + lane] = (pre[i] > -200.f) ? ldexpf(fr, (int)xi) : 0.f;
            }
            if (c + 1 < nchunk) issue(c + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) LIN_SET(L.fed, c + 1);
        }
    } else if (w >= 2) {
        // ---- storers: frame f = sw, sw + 3, ..: (a, e) of the lane's four states -> log offsets from the row reference ----
        const int sw = w - 2;
        float* out = (ROLE == 0 ? ws.alpha : ws.beta) + (size_t)b * T * ws.SP;
        double* outmax = (ROLE == 0 ? ws.amax : ws.bmax) + (size_t)b * T;
        for (int f = sw; f < Tb; f += NSTORE) {
            if (!wait_ge(L.done, f + 1, L.bad)) break;
            const int slot = f % RING;
            const pgasr_f4 av = *reinterpret_cast<const pgasr_f4*>(&L.ring_a[slot][lane][0]);
            const int e = L.ring_e[slot][lane];
            double lg[4];
            double ml = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = av[j];
                const int kx = __builtin_amdgcn_frexp_expf(v);
                const float fm = __builtin_amdgcn_frexp_mantf(v);                          // [0.5, 1)
                const double l2 = (double)(kx + e) + (double)__builtin_amdgcn_logf(fm);    // v_log_f32 = log2
                lg[j] = (v > 0.f) ? l2 * 0.6931471805599453 : -INFINITY;
                ml = fmax(ml, lg[j]);
            }
            const double mw = (double)wave_max_f32_all((float)ml);
            const double m = (mw == -INFINITY) ? 0.0 : mw;
            const int t_row = t0 + f * dt;
            if (4 * lane < ws.SP) {
                pgasr_f4 o = {(float)(lg[0] - m), (float)(lg[1] - m), (float)(lg[2] - m), (float)(lg[3] - m)};
                *reinterpret_cast<pgasr_f4*>(out + (size_t)t_row * ws.SP + 4 * lane) = o;
            }
            if (lane == 0) { outmax[t_row] = m; LIN_SET(L.st[sw], f + 1); }
        }
    } else {
        // ---- compute wave ----
        const int s0 = 4 * lane;
        int lab[4]; float skd[4]; bool val[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = s0 + j;
            val[j] = s < S; lab[j] = blank; skd[j] = 0.f;
            if (val[j] && (s & 1)) {
                const int l = tgt[s >> 1];
                lab[j] = (l < 0 || l >= V) ? blank : l;
                if (ROLE == 0) { if (s >= 3 && tgt[s >> 1] != tgt[(s >> 1) - 1]) skd[j] = 1.f; }
                else           { if (s + 2 < S && tgt[(s >> 1) + 1] != tgt[s >> 1]) skd[j] = 1.f; }
            }
        }
        auto pget = [&](int k, float (&p)[4]) {          // probabilities of this lane's states at sweep step k (chunk k / 32 must be fed)
            const float* src = &L.pbuf[(k / CH) & 1][k % CH][0];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = val[j] ? src[lab[j]] : 0.f;
        };
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        int e = 0;
        float scale = 0.f;            // 2^(upstream neighbour's exponent - this lane's): what its two hand-over states are multiplied by
        float pn[4];
        // ALIGNMENT (every fourth frame, never on a frame's critical path otherwise): each lane brings its largest value back to [0.5, 1);
        // empty lanes copy the exponent of the mass side (two hops: mass advances at most two lanes in four frames), so that mass arriving
        // later lands in a sensible exponent; a lane more than 2^REBASE below its upstream neighbour re-bases itself on that neighbour's
        // exponent (what it held is below the resolution of what will flow in) -- repeated until no lane is, because a re-base changes the
        // exponent the NEXT lane sees; finally every lane fixes the factor for its neighbour's hand-over values.  Between alignments the
        // exponents are frozen: values may grow by 3 * 2^REBASE per frame and lane (< 2^127 over four frames), or shrink (a frame
        // probability below ~2^-30 four frames in a row flushes a lane: see the header).
        auto dpp_i = [&](int v) { return ROLE == 0 ? __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); };
        auto align = [&]() {
            const float m = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
            const int kx = __builtin_amdgcn_frexp_expf(m);            // 0 for m == 0
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = ldexpf(a[j], -kx);
            e += kx;
            int ev = m > 0.f ? 1 : 0;                                 // is this lane's exponent meaningful?
#pragma unroll
            for (int hop = 0; hop < 2; ++hop) {
                const int enb = dpp_i(e), evn = dpp_i(ev);
                if (!ev && evn) { e = enb; ev = 1; }
            }
            for (int it = 0; it < 64; ++it) {
                const int enb = dpp_i(e), evn = dpp_i(ev);
                const int d = (evn && enb - e > REBASE) ? enb - e : 0;
                if (!__any(d != 0)) break;
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = ldexpf(a[j], -d);
                e += d; ev |= (d != 0);
            }
            const int enb = dpp_i(e), evn = dpp_i(ev);
            int sh = evn ? enb - e : 0;
            sh = sh < -140 ? -140 : sh;
            scale = (diag & 4) ? 1.f : ldexpf(1.f, sh);               // diag bit 2, timing only
        };
        bool ok = wait_ge(L.fed, 1, L.bad);
        if (ok) {
            pget(0, pn);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = s0 + j;
                if (ROLE == 0) { if (s <= 1) a[j] = pn[j]; }
                else           { if (s >= S - 2) a[j] = pn[j]; }       // pn is 0 past S
            }
            align();
        }
        auto put_row = [&](int k) {
            if (diag & 1) { if (lane == 0 && k == 0) LIN_SET(L.done, Tb); return; }      // timing only: no rows (the storers run through stale LDS)
            const int slot = k % RING;
            *reinterpret_cast<pgasr_f4*>(&L.ring_a[slot][lane][0]) = (pgasr_f4){a[0], a[1], a[2], a[3]};
            L.ring_e[slot][lane] = e;
            if (lane == 0) LIN_SET(L.done, k + 1);          // LDS operations of one wave execute in order: the row is written first
        };
        if (ok) { put_row(0); if (Tb > 1) pget(1, pn); }
        // one frame: hand-over (DPP) -> scale -> add -> fma -> mul.  The probabilities of the NEXT frame are fetched behind this frame's
        // arithmetic and row write, so nothing in a frame waits for an LDS read it has just issued.
        auto frame = [&](const int k, const bool fetch_next) {
            float nb1, nb2;
            if (diag & 8) { nb1 = a[1]; nb2 = a[2]; }                // timing only: no cross-lane hand-off
            else if (ROLE == 0) { nb1 = dpp_f<0x138>(a[3]); nb2 = dpp_f<0x138>(a[2]); }
            else                { nb1 = dpp_f<0x130>(a[0]); nb2 = dpp_f<0x130>(a[1]); }
            nb1 *= scale; nb2 *= scale;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this frame's probabilities (fetched a frame ago) and the last row's writes
            float n[4];
            if (ROLE == 0) {
                n[0] = (a[0] + nb1) * pn[0];                              // even states (blanks) never skip
                n[1] = fmaf(skd[1], nb1, a[1] + a[0]) * pn[1];
                n[2] = (a[2] + a[1]) * pn[2];
                n[3] = fmaf(skd[3], a[1], a[3] + a[2]) * pn[3];
            } else {
                n[3] = fmaf(skd[3], nb2, a[3] + nb1) * pn[3];
                n[2] = (a[2] + a[3]) * pn[2];
                n[1] = fmaf(skd[1], a[3], a[1] + a[2]) * pn[1];
                n[0] = (a[0] + a[1]) * pn[0];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = n[j];
            if ((k & 3) == 0) align();
            put_row(k);
            if (fetch_next && !(diag & 2)) pget(k + 1, pn);          // diag bit 1, timing only: the first frame's probabilities for every frame
        };
        const long long dg_c0 = (diag & 96) ? clock64() : 0, dg_r0 = (diag & 96) ? wall_clock64() : 0;
        // frames in blocks of eight: ring space and emission chunks are checked once per block, a frame itself carries no test
        for (int k0 = 0; k0 < Tb && ok; k0 += 8) {
            const int kend = k0 + 8 < Tb ? k0 + 8 : Tb;               // frames [max(k0, 1), kend); the last one fetches frame kend
            if (kend < Tb && !wait_ge(L.fed, kend / CH + 1, L.bad)) { ok = false; break; }     // the chunk of the frame fetched last in this block
            if (k0 % CH == 0 && lane == 0) LIN_SET(L.cchunk, k0 / CH);   // every read of the chunk before was issued (LDS in order): the feeder may refill it
            if (kend > RING) {                                         // rows up to kend - 1 reuse the slots of frames < kend - RING: those must be stored
                const int need = kend - RING - 2;                      // st + 2 >= X  <=>  every frame of that storer below X is stored
                for (int i = 0; i < NSTORE; ++i) ok = ok && wait_ge(L.st[i], need, L.bad);
                if (!ok) break;
            }
            for (int k = k0 > 0 ? k0 : 1; k < kend; ++k) frame(k, k + 1 < Tb);
        }
        // total likelihood from the last row (this wave's own LDS writes): alpha_{T-1}(S-1) + alpha_{T-1}(S-2) = beta_0(0) + beta_0(1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) {
            double ll = -INFINITY;
            bool redo = !ok || LIN_GET(L.bad) != 0;
            if (!redo && !(diag & 1)) {
                const int slot = (Tb - 1) % RING;
                double x[2] = {-INFINITY, -INFINITY};
                for (int i = 0; i < 2; ++i) {
                    const int s = (ROLE == 0) ? S - 1 - i : i;
                    if (s < 0 || s >= S) continue;
                    const float v = L.ring_a[slot][s >> 2][s & 3];
                    if (v > 0.f) x[i] = log((double)v) + (double)L.ring_e[slot][s >> 2] * 0.6931471805599453;
                }
                const double mx = fmax(x[0], x[1]);
                ll = (mx == -INFINITY) ? -INFINITY : mx + log(exp(x[0] - mx) + exp(x[1] - mx));
                // an empty lattice although the alignment is feasible (T >= L + repeats): a frame probability fell below fp32's range
                if (ll == -INFINITY) {
                    int rep = 0;
                    for (int i = 1; i < Lb; ++i) rep += (tgt[i] == tgt[i - 1]);
                    redo = Tb >= Lb + rep;
                }
            }
            if (redo) LIN_SET(L.redo, 1);
            else if (ROLE == 0) { ws.nll64[b] = -ll; nll_out[b] = (float)(-ll); }
            // timing only: shader-clock cycles (bit 5) / 100 MHz ticks x 100 (bit 6) per frame of this wave instead of the nll
            if (ROLE == 0 && (diag & 32)) nll_out[b] = (float)(clock64() - dg_c0) / (float)Tb;
            if (ROLE == 0 && (diag & 64)) nll_out[b] = 100.f * (float)(wall_clock64() - dg_r0) / (float)Tb;
        }
    }
    __syncthreads();
    return LIN_GET(L.redo) != 0;
}

__global__ __launch_bounds__(lin::THREADS) void ctc_lattice_lin_kernel(
    const float* __restrict__ lp, const int32_t* __restrict__ targets,
    const int32_t* __restrict__ in_len, const int32_t* __restrict__ tg_len,
    int T, int B, int V, int Lmax, int Smax, int blank, CtcWs ws, float* __restrict__ nll_out, int diag) {
    extern __shared__ __attribute__((aligned(32))) unsigned char lin_smem[];
    lin::Lds& L = *reinterpret_cast<lin::Lds*>(lin_smem);
    const int role = blockIdx.y;
    if (role == 0) {
        if (ctc_lattice_lin_body<0>(lp, targets, in_len, tg_len, T, B, V, Lmax, blank, ws, nll_out, L, diag) || (diag & 16))
            ctc_lattice_body<1, 0>(lp, targets, in_len, tg_len, T, B, V, Lmax, Smax, blank, ws, nll_out);
    } else if (role == 1) {
        if (ctc_lattice_lin_body<1>(lp, targets, in_len, tg_len, T, B, V, Lmax, blank, ws, nll_out, L, diag) || (diag & 16))
            ctc_lattice_body<1, 1>(lp, targets, in_len, tg_len, T, B, V, Lmax, Smax, blank, ws, nll_out);
    } else ctc_labels_body(targets, tg_len, V, Lmax, Smax, blank, ws);
}
static_assert(lin::THREADS == CTC_THREADS + 64, "the log-space body and ctc_labels_body run with the launch geometry of either lattice kernel");

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
    // S <= 256: the linear-domain kernel (PGASR_CTC_LIN=0 keeps the log-space kernel: A/B and fall-back)
    const char* elin = getenv("PGASR_CTC_LIN");
    if (Smax <= 4 * 64 && !(elin && elin[0] == '0')) {
        if (hipFuncSetAttribute((const void*)ctc_lattice_lin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(lin::Lds)) != hipSuccess)
            return PGASR_ERR_LAUNCH;
        PGASR_LAUNCH_KERNEL(ctc_lattice_lin_kernel, dim3(B, 3), dim3(lin::THREADS), sizeof(lin::Lds), st,
                            log_probs, targets, input_lengths, target_lengths, T, B, V, Lmax > 0 ? Lmax : 1, Smax, blank, ws, nll,
                            getenv("PGASR_CTC_LIN_DIAG") ? atoi(getenv("PGASR_CTC_LIN_DIAG")) : 0);
    } else if (Smax <= CTC_THREADS) PGASR_LATTICE(1);
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
