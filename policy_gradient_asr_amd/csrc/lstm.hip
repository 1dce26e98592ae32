// Bidirectional LSTM recurrence for gfx950 (model.py:39-44,52-55: 3 layers x 2 directions,
// H = 256, packed-sequence semantics).  The input projections X*W_ih^T are hoisted into one big
// MFMA GEMM per layer (gemm.hip); this file is the serial part:
//     gates_t = xproj_t + h_{t-1} W_hh^T ; i,f,o = sigmoid, g = tanh ; c_t = f c + i g ; h_t = o tanh c_t
// and its reverse-time gradient.
//
// Design (DESIGN.md "LSTM recurrence"): W_hh (1 MB fp32) does not fit one CU, so each
// (direction, 16-utterance batch group) is served by a CLUSTER of 16 persistent workgroups, one
// per CU, each owning 16 hidden units.  The weight slice stays in registers for the whole sweep
// (64 VGPRs of bf16 hi/lo pairs per lane); every step a workgroup
//   1. waits until the 16 slices of h_{t-1} are published (one monotonic counter per cluster),
//   2. loads h_{t-1} (16 utt x 256, as bf16 hi+lo = 16 KB) straight into MFMA B-operand registers,
//   3. 24 x v_mfma_f32_16x16x32_bf16 per wave: hi*hi + hi*lo + lo*hi with fp32 accumulation
//      (the 3-term split keeps ~16 mantissa bits per operand: error ~1e-5 of a gate pre-activation,
//      against 2e-3 for plain bf16),
//   4. gate math in fp32, state c/h in registers,
//   5. publishes its 16 x 16 slice of h_t: bf16 hi/lo re-laid out through LDS into the consumers'
//      operand order and written with ONE 1-KiB write-through (sc1) store, drained, then one
//      agent-scope atomic add on the cluster counter.
// Hand-off protocol = MI355X_MICROARCH.md "Valid forms", table row 1: every payload store sc1 and
// drained (s_waitcnt vmcnt(0)) by the storing wave before one lane signals; consumers poll the
// counter with sc1 loads, join a workgroup barrier, and read the payload ONLY with sc1 loads.
// No result depends on dispatch order or XCD placement; every spin is bounded (err flag + exit).
//
// The backward sweep has the same skeleton with K = 1024 (all gate gradients of the cluster) split
// over the 4 waves and reduced through LDS; it overwrites the saved gate activations with dgates
// in place, which the weight-gradient GEMMs then consume.
#include "common.h"

namespace {

constexpr int HID = 256;           // hidden size per direction (model.py:40)
constexpr int G_CLUSTER = 16;      // workgroups per (direction, batch group): 16 units each
constexpr int LSTM_THREADS = 256;
constexpr long long SPIN_TIMEOUT_TICKS = 300000000LL;  // 3 s of the 100 MHz realtime counter

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned short f2bf(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ void split_bf16(float x, unsigned short& hi, unsigned short& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}
__device__ __forceinline__ float sigmoidf_fast(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_fast(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }

struct LstmArgs {
    float* gates;            // [T][B][2][H][4]: in = permuted xproj (+biases); out = activations i,f,g,o;
                             // after the backward sweep: d(pre-activations)
    float* out;              // [T][B][2H]   h_t (zeros past each length)
    float* cbuf;             // [T][B][2][H] c_t
    const float* dout;       // [T][B][2H]   (backward) gradient w.r.t. out
    const u32x4* wpack;      // packed bf16 hi/lo W_hh in MFMA A-operand order (see pack kernel)
    unsigned char* xbuf;     // exchange buffers
    unsigned* ctr;           // [2][NBG] counters, 32 words apart
    int* err;                // set to 1 when a bounded wait gives up
    const int* lengths;      // [B]
    int T, B, NBG;
};

// bounded wait of wave 0 on the cluster counter; returns false on timeout
__device__ __forceinline__ bool wait_counter(unsigned* ctr, unsigned target) {
    unsigned v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= target) return true;
    const long long t0 = wall_clock64();
    unsigned spins = 0;
    while (true) {
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target) return true;
        if (((++spins) & 1023u) == 0 && wall_clock64() - t0 > SPIN_TIMEOUT_TICKS) return false;
    }
}

// ------------------------------------------------------------------------------------------
// forward sweep.  grid (16, NBG, 2); wave w of workgroup g owns units 4*(4g+w) .. +3
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_fwd_kernel(LstmArgs a) {
    const int g = blockIdx.x, bg = blockIdx.y, dir = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int tau = 4 * g + w;
    const int unit = 4 * tau + q;
    const int T = a.T, B = a.B;

    __shared__ __attribute__((aligned(16))) unsigned short hs[NT * 16 * 2 * 16];  // [nt][n][hl][16 units]
    __shared__ volatile int s_abort;
    if (tid == 0) s_abort = 0;

    // weight slice -> registers (A operand: row = 4*uu+gate, k = 32ks + 8(l>>4) + j)
    bf16x8 Whi[8], Wlo[8];
    {
        const u32x4* wp = a.wpack + ((size_t)(dir * 64 + tau) * 8) * 2 * 64;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            Whi[ks] = __builtin_bit_cast(bf16x8, wp[(ks * 2 + 0) * 64 + lane]);
            Wlo[ks] = __builtin_bit_cast(bf16x8, wp[(ks * 2 + 1) * 64 + lane]);
        }
    }
    const size_t xregion = (size_t)2 * NT * 32 * 16 * 2 * 16;  // bytes per (dir,bg): [parity][nt][kc][n][hl][16B]
    unsigned char* xb = a.xbuf + (size_t)(dir * a.NBG + bg) * xregion;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)xregion, 0x00020000);
    unsigned* ctr = a.ctr + (size_t)(dir * a.NBG + bg) * 32;

    int bidx[NT], len[NT];
    float c[NT], h[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bidx[nt] = bg * 16 * NT + nt * 16 + n;
        len[nt] = (bidx[nt] < B) ? a.lengths[bidx[nt]] : 0;
        c[nt] = 0.f; h[nt] = 0.f;
    }
    auto gate_ptr = [&](int t, int nt) {
        return reinterpret_cast<float4*>(a.gates + ((((size_t)t * B + bidx[nt]) * 2 + dir) * HID + unit) * 4);
    };
    float4 xg[NT];
    {
        const int t = dir ? T - 1 : 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xg[nt] = (bidx[nt] < B) ? *gate_ptr(t, nt) : make_float4(0, 0, 0, 0);
    }
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        float4 xn[NT];
        if (step + 1 < T) {
            const int tn = dir ? t - 1 : t + 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) xn[nt] = (bidx[nt] < B) ? *gate_ptr(tn, nt) : make_float4(0, 0, 0, 0);
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        if (step > 0) {
            if (w == 0) {
                if (!wait_counter(ctr, (unsigned)(G_CLUSTER * step))) { s_abort = 1; *a.err = 1; }
            }
            __syncthreads();
            if (s_abort) break;
            const int p = (step - 1) & 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bf16x8 Hhi[8], Hlo[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const unsigned off = (unsigned)(((((p * NT + nt) * 32 + (4 * ks + q)) * 16 + n) * 2) * 16);
                    Hhi[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16));
                    Hlo[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 16));
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Whi[ks], Hhi[ks], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Whi[ks], Hlo[ks], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wlo[ks], Hhi[ks], acc[nt], 0, 0, 0);
                }
            }
        }
        float4 gsave[NT];
        float hout[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float gi = sigmoidf_fast(acc[nt][0] + xg[nt].x);
            const float gf = sigmoidf_fast(acc[nt][1] + xg[nt].y);
            const float gg = tanhf_fast(acc[nt][2] + xg[nt].z);
            const float go = sigmoidf_fast(acc[nt][3] + xg[nt].w);
            const bool active = t < len[nt];
            const float cn = gf * c[nt] + gi * gg;
            const float hn = go * tanhf_fast(cn);
            if (active) { c[nt] = cn; h[nt] = hn; }
            gsave[nt] = active ? make_float4(gi, gf, gg, go) : make_float4(0, 0, 0, 0);
            hout[nt] = active ? hn : 0.f;
            unsigned short hi, lo;
            split_bf16(h[nt], hi, lo);
            hs[((nt * 16 + n) * 2 + 0) * 16 + 4 * w + q] = hi;
            hs[((nt * 16 + n) * 2 + 1) * 16 + 4 * w + q] = lo;
        }
        __syncthreads();
        if (w == 0 && step + 1 < T) {
            const int p = step & 1;
            const int ln = lane >> 2, lhl = (lane >> 1) & 1, lc2 = lane & 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(&hs[(nt * 64 + lane) * 8]);
                const unsigned off = (unsigned)((((((p * NT + nt) * 32 + (2 * g + lc2)) * 16 + ln) * 2) + lhl) * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // bulk stores for the backward pass / next layer (off the dependent chain)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (bidx[nt] < B) {
                *gate_ptr(t, nt) = gsave[nt];
                a.cbuf[(((size_t)t * B + bidx[nt]) * 2 + dir) * HID + unit] = c[nt];
                a.out[((size_t)t * B + bidx[nt]) * (2 * HID) + dir * HID + unit] = hout[nt];
            }
            xg[nt] = xn[nt];
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward sweep.  grid (16, NBG, 2); workgroup g owns OUTPUT units 16g..16g+15 of
// dh_{prev} = dgates * W_hh; wave w reduces over gate rows r' in [256w, 256w+256).
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_bwd_kernel(LstmArgs a) {
    const int g = blockIdx.x, bg = blockIdx.y, dir = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4, n = lane & 15;   // MFMA operand coordinates
    const int ul = tid >> 4, pn = tid & 15;   // pointwise coordinates: local unit, batch column
    const int unit = 16 * g + ul;
    const int T = a.T, B = a.B;

    __shared__ __attribute__((aligned(16))) float part[NT * 4 * 16 * 16];              // [nt][w][m][n]
    __shared__ __attribute__((aligned(16))) unsigned short dgs[NT * 16 * 2 * 64];      // [nt][n][hl][64 r']
    __shared__ volatile int s_abort;
    if (tid == 0) s_abort = 0;

    bf16x8 Whi[8], Wlo[8];
    {
        const u32x4* wp = a.wpack + ((size_t)(dir * 16 + g) * 32) * 2 * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ks = 8 * w + i;
            Whi[i] = __builtin_bit_cast(bf16x8, wp[(ks * 2 + 0) * 64 + lane]);
            Wlo[i] = __builtin_bit_cast(bf16x8, wp[(ks * 2 + 1) * 64 + lane]);
        }
    }
    const size_t xregion = (size_t)2 * NT * 128 * 16 * 2 * 16;  // [parity][nt][kc 128][n][hl][16B]
    unsigned char* xb = a.xbuf + (size_t)(dir * a.NBG + bg) * xregion;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)xregion, 0x00020000);
    unsigned* ctr = a.ctr + (size_t)(dir * a.NBG + bg) * 32;

    int bidx[NT], len[NT];
    float dc[NT], carry[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bidx[nt] = bg * 16 * NT + nt * 16 + pn;
        len[nt] = (bidx[nt] < B) ? a.lengths[bidx[nt]] : 0;
        dc[nt] = 0.f; carry[nt] = 0.f;
    }
    struct Saved { float4 gt; float ct, cp, dy; };
    auto load_saved = [&](int t, int nt) {
        Saved s; s.gt = make_float4(0, 0, 0, 0); s.ct = 0.f; s.cp = 0.f; s.dy = 0.f;
        if (bidx[nt] < B) {
            const size_t gi = (((size_t)t * B + bidx[nt]) * 2 + dir) * HID + unit;
            s.gt = *reinterpret_cast<const float4*>(a.gates + gi * 4);
            s.ct = a.cbuf[gi];
            const int tp = dir ? t + 1 : t - 1;   // the step the forward sweep ran just before t
            if (tp >= 0 && tp < T) s.cp = a.cbuf[(((size_t)tp * B + bidx[nt]) * 2 + dir) * HID + unit];
            s.dy = a.dout[((size_t)t * B + bidx[nt]) * (2 * HID) + dir * HID + unit];
        }
        return s;
    };
    Saved sv[NT];
    {
        const int t = dir ? 0 : T - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sv[nt] = load_saved(t, nt);
    }
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : T - 1 - step;
        Saved sn[NT];
        if (step + 1 < T) {
            const int tn = dir ? t + 1 : t - 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sn[nt] = load_saved(tn, nt);
        }
        float dh_rec[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dh_rec[nt] = carry[nt];

        if (step > 0) {
            if (w == 0) {
                if (!wait_counter(ctr, (unsigned)(G_CLUSTER * step))) { s_abort = 1; *a.err = 1; }
            }
            __syncthreads();
            if (s_abort) break;
            const int p = (step - 1) & 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bf16x8 Dhi[8], Dlo[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kc = 4 * (8 * w + i) + q;
                    const unsigned off = (unsigned)(((((p * NT + nt) * 128 + kc) * 16 + n) * 2) * 16);
                    Dhi[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16));
                    Dlo[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 16));
                }
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Whi[i], Dhi[i], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Whi[i], Dlo[i], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wlo[i], Dhi[i], acc, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) part[((nt * 4 + w) * 16 + (4 * q + j)) * 16 + n] = acc[j];
            }
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float* pp = &part[(nt * 4 * 16 + ul) * 16 + pn];
                dh_rec[nt] += (pp[0] + pp[256]) + (pp[512] + pp[768]);
            }
        }
        float4 dgt[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bool active = t < len[nt];
            const float gi = sv[nt].gt.x, gf = sv[nt].gt.y, gg = sv[nt].gt.z, go = sv[nt].gt.w;
            const float dh = sv[nt].dy + dh_rec[nt];
            const float tc = tanhf_fast(sv[nt].ct);
            const float dct = dh * go * (1.f - tc * tc) + dc[nt];
            float4 d;
            d.x = dct * gg * gi * (1.f - gi);
            d.y = dct * sv[nt].cp * gf * (1.f - gf);
            d.z = dct * gi * (1.f - gg * gg);
            d.w = dh * tc * go * (1.f - go);
            if (active) { dc[nt] = dct * gf; carry[nt] = 0.f; }
            else { d = make_float4(0, 0, 0, 0); carry[nt] = dh_rec[nt]; }
            dgt[nt] = d;
            unsigned short hi[4], lo[4];
            split_bf16(d.x, hi[0], lo[0]); split_bf16(d.y, hi[1], lo[1]);
            split_bf16(d.z, hi[2], lo[2]); split_bf16(d.w, hi[3], lo[3]);
            // [nt][n][hl][r' local = ul*4 + gate]
            unsigned short* dst = &dgs[((nt * 16 + pn) * 2) * 64 + ul * 4];
            *reinterpret_cast<uint2*>(dst) = make_uint2(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16));
            *reinterpret_cast<uint2*>(dst + 64) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
        }
        __syncthreads();
        if (step + 1 < T) {
            const int p = step & 1;
            const int ln = tid >> 4, lhl = (tid >> 3) & 1, lc8 = tid & 7;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(&dgs[(nt * 256 + tid) * 8]);
                const unsigned off = (unsigned)((((((p * NT + nt) * 128 + (8 * g + lc8)) * 16 + ln) * 2) + lhl) * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (bidx[nt] < B) {
                const size_t gi = (((size_t)t * B + bidx[nt]) * 2 + dir) * HID + unit;
                *reinterpret_cast<float4*>(a.gates + gi * 4) = dgt[nt];
            }
            sv[nt] = sn[nt];
        }
    }
}

// ------------------------------------------------------------------------------------------
// weight packing / gradient unpacking (layout glue between torch's (4H, in) gate-major
// parameters -- rows i|f|g|o, SURVEY Appendix A -- and the unit-major, gate-minor column order
// the recurrent kernels use: column = dir*4H + unit*4 + gate)
// ------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w_ih[2]; const float* w_hh[2]; const float* b_ih[2]; const float* b_hh[2];
    int in_dim;
    float* wih_perm;        // [2*4H][in_dim]
    float* bias_perm;       // [2*4H]
    unsigned short* wpf;    // forward A-operand pack  [2][64 tiles][8 ks][2 hl][64 lanes][8]
    unsigned short* wpb;    // backward A-operand pack [2][16 tiles][32 ks][2 hl][64 lanes][8]
};

__global__ __launch_bounds__(256) void lstm_pack_kernel(PackArgs p) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n_wih = (size_t)2 * 4 * HID * p.in_dim;
    const size_t n_pack = (size_t)2 * 64 * 8 * 64 * 8;  // elements per hl plane set (one of hi/lo)
    if (i < n_wih) {
        const int col = (int)(i % p.in_dim);
        const int row = (int)(i / p.in_dim);          // dir*4H + unit*4 + gate
        const int dir = row / (4 * HID), r = row % (4 * HID);
        const int unit = r >> 2, gate = r & 3;
        p.wih_perm[i] = p.w_ih[dir][(size_t)(gate * HID + unit) * p.in_dim + col];
        if (col == 0) p.bias_perm[row] = p.b_ih[dir][gate * HID + unit] + p.b_hh[dir][gate * HID + unit];
    }
    if (i < n_pack) {
        // forward pack element: [dir][tau][ks][lane][j]
        int j = (int)(i & 7); size_t r = i >> 3;
        int lane = (int)(r & 63); r >>= 6;
        int ks = (int)(r & 7); r >>= 3;
        int tau = (int)(r & 63); int dir = (int)(r >> 6);
        {
            const int row = lane & 15, uu = row >> 2, gate = row & 3;
            const int k = 32 * ks + 8 * (lane >> 4) + j;
            const float v = p.w_hh[dir][(size_t)(gate * HID + 4 * tau + uu) * HID + k];
            unsigned short hi, lo; split_bf16(v, hi, lo);
            const size_t base = ((((size_t)(dir * 64 + tau) * 8 + ks) * 2) * 64 + lane) * 8 + j;
            p.wpf[base] = hi; p.wpf[base + 64 * 8] = lo;
        }
        // backward pack element: reinterpret the same flat index as [dir][mu][ks32][lane][j]
        j = (int)(i & 7); r = i >> 3;
        lane = (int)(r & 63); r >>= 6;
        int ks32 = (int)(r & 31); r >>= 5;
        int mu = (int)(r & 15); dir = (int)(r >> 4);
        {
            const int ko = 16 * mu + (lane & 15);
            const int rp = 32 * ks32 + 8 * (lane >> 4) + j;   // r' = unit*4 + gate
            const int unit = rp >> 2, gate = rp & 3;
            const float v = p.w_hh[dir][(size_t)(gate * HID + unit) * HID + ko];
            unsigned short hi, lo; split_bf16(v, hi, lo);
            const size_t base = ((((size_t)(dir * 16 + mu) * 32 + ks32) * 2) * 64 + lane) * 8 + j;
            p.wpb[base] = hi; p.wpb[base + 64 * 8] = lo;
        }
    }
}

struct UnpackArgs {
    const float* dwih_perm;   // [2*4H][in_dim]
    const float* dbias_perm;  // [2*4H]
    const float* dwhh_perm;   // [2][4H perm rows][H]
    float* dw_ih[2]; float* dw_hh[2]; float* db_ih[2]; float* db_hh[2];
    int in_dim; int accumulate;
};

__global__ __launch_bounds__(256) void lstm_unpack_kernel(UnpackArgs u) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n_wih = (size_t)2 * 4 * HID * u.in_dim;
    const size_t n_whh = (size_t)2 * 4 * HID * HID;
    if (i < n_wih) {
        const int col = (int)(i % u.in_dim);
        const int row = (int)(i / u.in_dim);
        const int dir = row / (4 * HID), r = row % (4 * HID);
        const int unit = r >> 2, gate = r & 3;
        float* d = &u.dw_ih[dir][(size_t)(gate * HID + unit) * u.in_dim + col];
        *d = u.accumulate ? *d + u.dwih_perm[i] : u.dwih_perm[i];
        if (col == 0) {
            const float b = u.dbias_perm[row];
            float* d1 = &u.db_ih[dir][gate * HID + unit];
            float* d2 = &u.db_hh[dir][gate * HID + unit];
            *d1 = u.accumulate ? *d1 + b : b;
            *d2 = u.accumulate ? *d2 + b : b;
        }
    }
    if (i < n_whh) {
        const int col = (int)(i % HID);
        const int row = (int)(i / HID);
        const int dir = row / (4 * HID), r = row % (4 * HID);
        const int unit = r >> 2, gate = r & 3;
        float* d = &u.dw_hh[dir][(size_t)(gate * HID + unit) * HID + col];
        *d = u.accumulate ? *d + u.dwhh_perm[i] : u.dwhh_perm[i];
    }
}

struct WsLayout { size_t xbuf, ctr, err, total; };
WsLayout lstm_ws_layout(int B, bool backward) {
    const int NT = 1;
    const int NBG = (B + 16 * NT - 1) / (16 * NT);
    WsLayout l;
    size_t off = 0;
    const size_t xregion = (size_t)2 * NT * (backward ? 128 : 32) * 16 * 2 * 16;
    l.ctr = off; off += (size_t)2 * NBG * 32 * sizeof(unsigned);   // zeroed every call
    l.err = off; off += 256;
    l.xbuf = off; off += (size_t)2 * NBG * xregion;
    l.total = off;
    return l;
}

}  // namespace

extern "C" size_t pgasr_lstm_pack_bytes(int which) {
    // which: 0 = forward W_hh pack, 1 = backward W_hh pack (both 2 MB: bf16 hi+lo of 2 x 1024 x 256)
    (void)which;
    return (size_t)2 * 64 * 8 * 2 * 64 * 8 * sizeof(unsigned short);
}

extern "C" int pgasr_lstm_pack_weights(const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                                       const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                                       int in_dim, float* wih_perm, float* bias_perm,
                                       void* whh_pack_fwd, void* whh_pack_bwd, void* stream) {
    if (!w_ih_f || !w_hh_f || !b_ih_f || !b_hh_f || !w_ih_r || !w_hh_r || !b_ih_r || !b_hh_r) return PGASR_ERR_INVALID_ARG;
    if (!wih_perm || !bias_perm || !whh_pack_fwd || !whh_pack_bwd || in_dim <= 0) return PGASR_ERR_INVALID_ARG;
    PackArgs p;
    p.w_ih[0] = w_ih_f; p.w_hh[0] = w_hh_f; p.b_ih[0] = b_ih_f; p.b_hh[0] = b_hh_f;
    p.w_ih[1] = w_ih_r; p.w_hh[1] = w_hh_r; p.b_ih[1] = b_ih_r; p.b_hh[1] = b_hh_r;
    p.in_dim = in_dim; p.wih_perm = wih_perm; p.bias_perm = bias_perm;
    p.wpf = (unsigned short*)whh_pack_fwd; p.wpb = (unsigned short*)whh_pack_bwd;
    const size_t n_wih = (size_t)2 * 4 * HID * in_dim, n_pack = (size_t)2 * 64 * 8 * 64 * 8;
    const size_t n = n_wih > n_pack ? n_wih : n_pack;
    hipLaunchKernelGGL(lstm_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_lstm_unpack_grads(const float* dwih_perm, const float* dbias_perm, const float* dwhh_perm, int in_dim,
                                       float* dw_ih_f, float* dw_hh_f, float* db_ih_f, float* db_hh_f,
                                       float* dw_ih_r, float* dw_hh_r, float* db_ih_r, float* db_hh_r,
                                       int accumulate, void* stream) {
    if (!dwih_perm || !dbias_perm || !dwhh_perm || in_dim <= 0) return PGASR_ERR_INVALID_ARG;
    if (!dw_ih_f || !dw_hh_f || !db_ih_f || !db_hh_f || !dw_ih_r || !dw_hh_r || !db_ih_r || !db_hh_r) return PGASR_ERR_INVALID_ARG;
    UnpackArgs u;
    u.dwih_perm = dwih_perm; u.dbias_perm = dbias_perm; u.dwhh_perm = dwhh_perm;
    u.dw_ih[0] = dw_ih_f; u.dw_hh[0] = dw_hh_f; u.db_ih[0] = db_ih_f; u.db_hh[0] = db_hh_f;
    u.dw_ih[1] = dw_ih_r; u.dw_hh[1] = dw_hh_r; u.db_ih[1] = db_ih_r; u.db_hh[1] = db_hh_r;
    u.in_dim = in_dim; u.accumulate = accumulate;
    const size_t n_wih = (size_t)2 * 4 * HID * in_dim, n_whh = (size_t)2 * 4 * HID * HID;
    const size_t n = n_wih > n_whh ? n_wih : n_whh;
    hipLaunchKernelGGL(lstm_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" size_t pgasr_lstm_workspace_bytes(int T, int B, int backward) {
    if (T <= 0 || B <= 0) return 0;
    return lstm_ws_layout(B, backward != 0).total;
}

static int lstm_launch(bool backward, float* gates, float* out, float* cbuf, const float* dout, const void* wpack,
                       const int* lengths, int T, int B, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (!gates || !out || !cbuf || !wpack || !lengths || T <= 0 || B <= 0) return PGASR_ERR_INVALID_ARG;
    if (backward && !dout) return PGASR_ERR_INVALID_ARG;
    const WsLayout l = lstm_ws_layout(B, backward);
    if (!workspace || workspace_bytes < l.total) return PGASR_ERR_WORKSPACE;
    const int NBG = (B + 15) / 16;
    // every cluster must be co-resident: 16 * NBG * 2 workgroups, one per CU
    if (G_CLUSTER * NBG * 2 > 256) return PGASR_ERR_UNSUPPORTED;
    char* ws = (char*)workspace;
    if (hipMemsetAsync(ws + l.ctr, 0, l.err + 256 - l.ctr, st) != hipSuccess) return PGASR_ERR_LAUNCH;
    LstmArgs a;
    a.gates = gates; a.out = out; a.cbuf = cbuf; a.dout = dout; a.wpack = (const u32x4*)wpack;
    a.xbuf = (unsigned char*)(ws + l.xbuf); a.ctr = (unsigned*)(ws + l.ctr); a.err = (int*)(ws + l.err);
    a.lengths = lengths; a.T = T; a.B = B; a.NBG = NBG;
    dim3 grid(G_CLUSTER, NBG, 2);
    if (backward) hipLaunchKernelGGL(lstm_bwd_kernel<1>, grid, dim3(LSTM_THREADS), 0, st, a);
    else hipLaunchKernelGGL(lstm_fwd_kernel<1>, grid, dim3(LSTM_THREADS), 0, st, a);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_lstm_layer_fwd(float* gates, float* out, float* cbuf, const void* whh_pack_fwd,
                                    const int32_t* lengths, int T, int B,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return lstm_launch(false, gates, out, cbuf, nullptr, whh_pack_fwd, lengths, T, B, workspace, workspace_bytes,
                       (hipStream_t)stream);
}

extern "C" int pgasr_lstm_layer_bwd(float* gates, const float* out, const float* cbuf, const float* dout,
                                    const void* whh_pack_bwd, const int32_t* lengths, int T, int B,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return lstm_launch(true, gates, const_cast<float*>(out), const_cast<float*>(cbuf), dout, whh_pack_bwd, lengths, T, B,
                       workspace, workspace_bytes, (hipStream_t)stream);
}

// reads the error word written by a timed-out wait (host-side check after a sync)
extern "C" int pgasr_lstm_error_offset(int B, int backward, size_t* offset) {
    if (!offset || B <= 0) return PGASR_ERR_INVALID_ARG;
    *offset = lstm_ws_layout(B, backward != 0).err;
    return PGASR_OK;
}
