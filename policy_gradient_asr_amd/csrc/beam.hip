// CTC prefix beam search on gfx950 (CTCdecoder.py:41-116, Hannun's algorithm in log space).
//
// One workgroup per utterance; frames are a serial chain, the work inside a frame is parallel:
//   * beam entries are nodes of a per-utterance trie (parent id, symbol), hash-consed through a
//     (parent,symbol) -> id table in global memory so that one prefix has ONE id for the whole
//     utterance; "extension of entry i by s equals existing entry e" is then the exact integer test
//     parent(e) == id(i) && last(e) == s  (the reference merges by tuple equality, :88,:100);
//   * every (entry j, symbol s) pair is one candidate: s == blank keeps prefix j ("stay"), any other
//     s extends it unless that extension already sits in the beam, in which case its mass is added to
//     that entry's stay candidate in the reference's update order (:90-96, :103-106);
//   * candidates are ranked by a bitonic sort in LDS on (score descending, first-touch time
//     ascending) -- the reference's stable sort over dict insertion order (:110-113), whose loop nest
//     is symbol-major / rank-minor (:68,:74).
// Scores are fp64 (log-sum-exp exactly as CTCdecoder.py:31-39: max, sum of exps in argument order,
// log); this is integer/latency work, not MFMA work.
#include "common.h"

namespace {

constexpr int BEAM_KMAX = 128;
constexpr int BEAM_VMAX = 64;
constexpr int BEAM_THREADS = 256;

struct BeamWs {
    unsigned long long* table;   // [B][H] open-addressing (key+1)<<32 | node id ; 0 = empty
    unsigned* nodes;             // [B][NN] packed (parent << 8 | sym); node 0 = root
    int H, NN;
};

__host__ __device__ inline size_t beam_ws_layout(int T, int B, int beam, BeamWs* ws, char* base) {
    const long long nn = (long long)T * beam + 1;
    int H = 1024;
    while ((long long)H < 2 * nn) H <<= 1;
    size_t off = 0;
    const size_t t_off = off; off += (size_t)B * H * sizeof(unsigned long long);
    off = (off + 255) / 256 * 256;
    const size_t n_off = off; off += (size_t)B * nn * sizeof(unsigned);
    off = (off + 255) / 256 * 256;
    if (ws) {
        ws->table = (unsigned long long*)(base + t_off);
        ws->nodes = (unsigned*)(base + n_off);
        ws->H = H; ws->NN = (int)nn;
    }
    return off;
}

// CTCdecoder.py:31-39 with 2 or 3 arguments (argument order preserved)
__device__ __forceinline__ double lse2(double a, double b) {
    const double m = fmax(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + log(exp(a - m) + exp(b - m));
}
__device__ __forceinline__ double lse3(double a, double b, double c) {
    const double m = fmax(fmax(a, b), c);
    if (m == -INFINITY) return -INFINITY;
    return m + log((exp(a - m) + exp(b - m)) + exp(c - m));
}

// Fast variants (fp32 device log-probs, training-time rewards): fp64 carries, fp32 exp/log on the
// differences to the maximum -- the software fp64 transcendentals dominate the exact path (28 ms
// vs ~1/4 of that for T=1000, beam 16, 32 utterances).  Scores differ from the exact path by
// ~1e-7 relative, so only exact score ties could rank differently.
template <bool FAST> __device__ __forceinline__ double lse2x(double a, double b) {
    if (!FAST) return lse2(a, b);
    const double m = fmax(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + (double)__logf(__expf((float)(a - m)) + __expf((float)(b - m)));
}
template <bool FAST> __device__ __forceinline__ double lse3x(double a, double b, double c) {
    if (!FAST) return lse3(a, b, c);
    const double m = fmax(fmax(a, b), c);
    if (m == -INFINITY) return -INFINITY;
    return m + (double)__logf((__expf((float)(a - m)) + __expf((float)(b - m))) + __expf((float)(c - m)));
}

// order-preserving map double -> u64 (ascending)
__device__ __forceinline__ unsigned long long ordered_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

struct SortItem { unsigned long long key; unsigned time; unsigned cand; };   // key = ~ordered(score): ascending sort = score descending

__device__ __forceinline__ bool item_less(const SortItem& a, const SortItem& b) {
    return a.key < b.key || (a.key == b.key && a.time < b.time);
}

template <typename TIn, bool FAST>
__global__ __launch_bounds__(BEAM_THREADS) void beam_search_kernel(
    const TIn* __restrict__ lp, long long stride_t, long long stride_b, const int32_t* __restrict__ lengths,
    int T, int V, int K, int blank, BeamWs ws, int32_t* __restrict__ out_tokens, int32_t* __restrict__ out_len,
    double* __restrict__ out_score) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    int Tb = lengths ? lengths[b] : T; Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);

    int P = 1; while (P < K * V) P <<= 1;                 // sort size
    // LDS carve-up (all offsets multiples of 16)
    SortItem* items = reinterpret_cast<SortItem*>(smem);                                   // P
    double* c_pb = reinterpret_cast<double*>(smem + (size_t)P * sizeof(SortItem));          // K*V
    double* c_pnb = c_pb + (size_t)K * V;                                                   // K*V
    double* pb = c_pnb + (size_t)K * V;        // [2][K]
    double* pnb = pb + 2 * K;                  // [2][K]
    double* frame = pnb + 2 * K;               // V log-probs of the current frame
    int* id = reinterpret_cast<int*>(frame + BEAM_VMAX);   // [2][K]
    int* last = id + 2 * K;                    // [2][K]
    int* par = last + 2 * K;                   // [2][K]
    int* pidx = par + 2 * K;                   // [K]
    int* s_ctl = pidx + K;                     // [4]: nb, node counter
    short* cb = reinterpret_cast<short*>(s_ctl + 4);       // [K][V] child-in-beam table
#define s_nb s_ctl[0]
#define s_nodes s_ctl[1]

    unsigned long long* table = ws.table + (size_t)b * ws.H;
    unsigned* nodes = ws.nodes + (size_t)b * ws.NN;
    const unsigned hmask = (unsigned)ws.H - 1u;

    if (tid == 0) {
        s_nb = 1; s_nodes = 1;
        id[0] = 0; last[0] = -1; par[0] = -1; pb[0] = 0.0; pnb[0] = -INFINITY;
        nodes[0] = 0xFFFFFFFFu;
    }
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < Tb; ++t) {
        const int nb = s_nb;
        const int* cid = id + cur * K; const int* clast = last + cur * K; const int* cpar = par + cur * K;
        const double* cpb = pb + cur * K; const double* cpnb = pnb + cur * K;
        if (tid < V) frame[tid] = (double)lp[(long long)t * stride_t + (long long)b * stride_b + tid];
        for (int i = tid; i < nb * V; i += BEAM_THREADS) cb[i] = -1;
        if (tid < nb) {
            int f = -1;
            for (int i = 0; i < nb; ++i) if (cid[i] == cpar[tid]) f = i;
            pidx[tid] = f;
        }
        __syncthreads();
        if (tid < nb && pidx[tid] >= 0) cb[pidx[tid] * V + clast[tid]] = (short)tid;
        __syncthreads();
        // ---- candidates: c = j*V + s ----
        const int C = nb * V;
        for (int c = tid; c < P; c += BEAM_THREADS) {
            SortItem it; it.key = ~0ull; it.time = 0xFFFFFFFFu; it.cand = 0xFFFFFFFFu;   // padding sorts last
            if (c < C) {
                const int j = c / V, s = c % V;
                double npb = -INFINITY, npnb = -INFINITY;
                unsigned time = 0xFFFFFFFFu;
                bool alive = true;
                if (s == blank) {
                    // prefix j unchanged.  blank update (:78-82)
                    npb = lse3x<FAST>(-INFINITY, cpb[j] + frame[blank], cpnb[j] + frame[blank]);
                    time = (unsigned)(blank * nb + j);
                    const int lj = clast[j];
                    if (lj >= 0 && lj != blank) {
                        const double pl = frame[lj];
                        const int i = pidx[j];
                        // two possible non-blank updates, applied in loop order (symbol lj, rank i vs j)
                        const double own = cpnb[j] + pl;                              // repeat branch (:103-106)
                        if (i >= 0) {
                            double e1 = cpb[i] + pl, e2 = cpnb[i] + pl;               // extension of parent i by lj (:90-96)
                            const bool rep = (clast[i] == lj);
                            if (i < j) {
                                npnb = rep ? lse2x<FAST>(npnb, e1) : lse3x<FAST>(npnb, e1, e2);
                                npnb = lse2x<FAST>(npnb, own);
                            } else {
                                npnb = lse2x<FAST>(npnb, own);
                                npnb = rep ? lse2x<FAST>(npnb, e1) : lse3x<FAST>(npnb, e1, e2);
                            }
                            const unsigned tt = (unsigned)(lj * nb + (i < j ? i : j));
                            time = tt < time ? tt : time;
                        } else {
                            npnb = lse2x<FAST>(npnb, own);
                            const unsigned tt = (unsigned)(lj * nb + j);
                            time = tt < time ? tt : time;
                        }
                    }
                } else if (cb[j * V + s] >= 0) {
                    alive = false;          // merged into an existing entry's stay candidate
                } else {
                    const double ps = frame[s];
                    npnb = (s != clast[j]) ? lse3x<FAST>(-INFINITY, cpb[j] + ps, cpnb[j] + ps) : lse2x<FAST>(-INFINITY, cpb[j] + ps);
                    time = (unsigned)(s * nb + j);
                }
                if (alive) {
                    c_pb[c] = npb; c_pnb[c] = npnb;
                    it.key = ~ordered_key(lse2x<FAST>(npb, npnb));
                    it.time = time; it.cand = (unsigned)c;
                }
            }
            items[c] = it;
        }
        __syncthreads();
        // ---- bitonic sort of P items ----
        // Element i is handled by thread i % 256, so all elements of a 64-aligned group belong to one
        // wave: a pass with partner distance j2 < 64 exchanges only inside a wave (LDS accesses of one
        // wave are ordered) and needs a workgroup barrier only next to a cross-wave pass.
        {
            bool prev_cross = false;   // the candidate loop above ended with __syncthreads()
            for (int k2 = 2; k2 <= P; k2 <<= 1) {
                for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                    const bool cross = j2 >= 64;
                    if (cross || prev_cross) __syncthreads();
                    for (int i = tid; i < P; i += BEAM_THREADS) {
                        const int l = i ^ j2;
                        if (l > i) {
                            const SortItem x = items[i], y = items[l];
                            const bool up = ((i & k2) == 0);
                            if (item_less(y, x) == up) { items[i] = y; items[l] = x; }
                        }
                    }
                    prev_cross = cross;
                }
            }
            __syncthreads();
        }
        // ---- new beam ----
        const int nxt = cur ^ 1;
        int* nid = id + nxt * K; int* nlast = last + nxt * K; int* npar = par + nxt * K;
        double* nbpb = pb + nxt * K; double* nbpnb = pnb + nxt * K;
        if (tid < K) {
            const SortItem it = items[tid];
            if (it.cand != 0xFFFFFFFFu) {
                const int c = (int)it.cand, j = c / V, s = c % V;
                nbpb[tid] = c_pb[c]; nbpnb[tid] = c_pnb[c];
                if (s == blank) { nid[tid] = cid[j]; nlast[tid] = clast[j]; npar[tid] = cpar[j]; }
                else {
                    // canonical node for (id[j], s): look up, else create
                    const unsigned keyv = ((unsigned)cid[j] << 8) | (unsigned)s;
                    unsigned h = (keyv * 2654435761u) & hmask;
                    int node = -1;
                    while (true) {
                        const unsigned long long slot = table[h];
                        if (slot == 0ull) {
                            if (node < 0) { node = atomicAdd(&s_nodes, 1); nodes[node] = keyv; }
                            const unsigned long long want = ((unsigned long long)(keyv + 1u) << 32) | (unsigned)node;
                            const unsigned long long old = atomicCAS(&table[h], 0ull, want);
                            if (old == 0ull) break;
                            if ((unsigned)(old >> 32) == keyv + 1u) { node = (int)(unsigned)old; break; }   // cannot happen within a frame
                        } else if ((unsigned)(slot >> 32) == keyv + 1u) { node = (int)(unsigned)slot; break; }
                        h = (h + 1u) & hmask;
                    }
                    nid[tid] = node; nlast[tid] = s; npar[tid] = cid[j];
                }
            }
        }
        if (tid == 0) {
            int cnt = 0;
            for (int r = 0; r < K && r < P; ++r) if (items[r].cand != 0xFFFFFFFFu) ++cnt;
            s_nb = cnt;
        }
        __syncthreads();
        cur = nxt;
    }
    // ---- result: walk the best entry's ancestors ----
    if (tid == 0) {
        const int best = id[cur * K];
        int n = 0;
        for (int v = best; v > 0; v = (int)(nodes[v] >> 8)) ++n;
        out_len[b] = n;
        int32_t* o = out_tokens + (size_t)b * T;
        int k = n - 1;
        for (int v = best; v > 0; v = (int)(nodes[v] >> 8)) o[k--] = (int32_t)(nodes[v] & 0xFFu);
        out_score[b] = -lse2x<FAST>(pb[cur * K], pnb[cur * K]);
    }
}

#undef s_nb
#undef s_nodes

inline size_t beam_lds_bytes(int K, int V) {
    int P = 1; while (P < K * V) P <<= 1;
    size_t n = (size_t)P * sizeof(SortItem);
    n += (size_t)2 * K * V * sizeof(double);          // c_pb, c_pnb
    n += (size_t)4 * K * sizeof(double);              // pb, pnb double-buffered
    n += (size_t)BEAM_VMAX * sizeof(double);          // frame
    n += (size_t)(6 * K + K + 4) * sizeof(int);       // id,last,par (x2), pidx, control words
    n += (size_t)K * V * sizeof(short);               // cb
    return (n + 15) / 16 * 16;
}

}  // namespace

extern "C" size_t pgasr_beam_workspace_bytes(int T, int B, int V, int beam) {
    if (T <= 0 || B <= 0 || V <= 0 || beam <= 0) return 0;
    return beam_ws_layout(T, B, beam, nullptr, nullptr);
}

extern "C" int pgasr_ctc_beam_search(const void* log_probs, int is_f64, long long stride_t, long long stride_b,
                                     const int32_t* lengths, int T, int B, int V, int beam, int blank,
                                     int32_t* out_tokens, int32_t* out_len, double* out_score,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    if (!log_probs || !out_tokens || !out_len || !out_score) return PGASR_ERR_INVALID_ARG;
    if (T <= 0 || B <= 0 || V <= 0 || beam <= 0 || blank < 0 || blank >= V) return PGASR_ERR_INVALID_ARG;
    if (V > BEAM_VMAX || beam > BEAM_KMAX) return PGASR_ERR_UNSUPPORTED;
    BeamWs ws;
    const size_t need = beam_ws_layout(T, B, beam, &ws, (char*)workspace);
    if (!workspace || workspace_bytes < need) return PGASR_ERR_WORKSPACE;
    if ((long long)T * beam + 1 >= (1ll << 24)) return PGASR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws.table, 0, (size_t)B * ws.H * sizeof(unsigned long long), st) != hipSuccess) return PGASR_ERR_LAUNCH;
    const size_t lds = beam_lds_bytes(beam, V);
    if (lds > 160 * 1024) return PGASR_ERR_UNSUPPORTED;
    if (is_f64) {   // exact path: fp64 transcendentals (drop-in CTCDecoder.decode)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&beam_search_kernel<double, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        PGASR_LAUNCH_KERNEL((beam_search_kernel<double, false>), dim3(B), dim3(BEAM_THREADS), lds, st, (const double*)log_probs,
                           stride_t, stride_b, lengths, T, V, beam, blank, ws, out_tokens, out_len, out_score);
    } else {        // fp32 device log-probs: fast transcendentals
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&beam_search_kernel<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        PGASR_LAUNCH_KERNEL((beam_search_kernel<float, true>), dim3(B), dim3(BEAM_THREADS), lds, st, (const float*)log_probs,
                           stride_t, stride_b, lengths, T, V, beam, blank, ws, out_tokens, out_len, out_score);
    }
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}
