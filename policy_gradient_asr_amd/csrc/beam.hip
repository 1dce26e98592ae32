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
#include <type_traits>

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
    int T, int V, int K, int blank, int collapse, BeamWs ws, int32_t* __restrict__ out_tokens, int32_t* __restrict__ out_len,
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
        if (collapse) {      // collapse_fn (CTCdecoder.py:119-131): adjacent duplicates removed
            int w = 0;
            for (int i = 0; i < n; ++i) if (i == 0 || o[i] != o[i - 1]) { const int32_t x = o[i]; o[w++] = x; }
            for (int i = w; i < n; ++i) o[i] = 0;
            out_len[b] = w;
        }
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


// =====================================================================================================================
// Small-beam search for the training path (reward hypothesis of policy_grad.py:6-8 inside the train step):
// fp32 device log-probs, beam <= 16, V <= 64, T * beam <= 24576 and T <= 4096.  ONE WAVE per utterance, no workgroup barrier, no
// LDS sort: the generic kernel above spends 24 us per frame in 45 LDS bitonic passes with barriers (24 ms for T = 1000);
// a frame here is ~800 wave instructions.
//
//   lane = (q = lane / 16, j = lane % 16): entry j's state (p_b, p_nb, total, node id, last symbol, parent id) is
//     replicated in the four 16-lane rows; lane (q, j) owns the candidates (entry j, symbol SPL q + i), i < SPL = 8 (V <= 32; 16 for
//     V <= 64, round 5: same kernel, twice the candidates per lane, a 63-exchange network; 4.5-4.9 ms for 32 x T = 1000 at beam 16 against
//     3.2-4.1 at V <= 32 and 45 ms on the generic kernel, tools/dev/r5_beam_v64.py) -- the slot
//     of the blank symbol holds entry j's "stay" candidate (prefix unchanged).
//   scores: the same algebra as CTCdecoder.py:74-106 with the per-entry total lse(p_b, p_nb) factored out (an
//     extension is total_j + log p(s), or p_b_j + log p(s) for a repeat of the last symbol; a stay collects blank,
//     repeat and -- when entry j's parent i is in the beam -- the merged extension of i by last(j)); fp64 carries,
//     fp32 exp/log on differences (like the generic fp32 path: ~1e-7 relative, so exact ties aside both rank alike).
//   prefix identity: hash-consed trie in LDS.  A node's id IS its slot in one 32768-word open-addressing table whose
//     word holds (parent id << 8 | symbol) + 1 -- table and node store in one array -- and, in bits 25..29, the beam
//     position + 1 of the entry that currently carries that prefix: "is my parent in the beam, where?" is one LDS
//     read, "which of my children are in the beam" one LDS atomic-or per entry.
//   top-K: every candidate becomes a unique 64-bit key = order-preserving image of (score - best total) with the low
//     17 mantissa bits replaced by [first-touch order of the reference's loop nest (:68,:74), slot, entry]: descending
//     key order IS the reference's stable sort (:110-113).  Each lane sorts its 8 keys with a 19-exchange network in
//     registers; K rounds of (row all-reduce max by DPP rotations on the high words, 4 readlanes, rarely a second pass
//     on the low words) pop the winners in rank order (a pop shifts the winner lane's register list).
//   frames are staged 32 at a time through LDS, two chunks ahead in registers, so no load is ever waited for.
// =====================================================================================================================
#ifdef PGASR_BEAM_DIAG
__device__ unsigned long long beam_diag_counters[4];      // frames, frames redone by the exact rounds, hash-table probes, longest probe chain
#endif
#ifndef PGASR_BEAM_UNROLL
#define PGASR_BEAM_UNROLL 1
#endif
namespace sb {
constexpr int K_MAX = 16, V_MAX = 64, H = 32768, CH = 32;     // template parameter SPL = symbols per lane: 8 (V <= 32) or 16 (V <= 64)
constexpr long long MAX_NODES = 24576;               // T * beam: load factor of the table <= 0.75
constexpr int MAX_TOKENS = 4096;                     // a hypothesis has at most T tokens and is staged in the 8 KB `frames` region (beam < 6 would admit longer T)
constexpr unsigned ROOT = 0x8000u, NONE = 0xFFFFu, BAD_ID = 0x0FFFFFFFu;
constexpr unsigned KEYMASK = 0x01FFFFFFu;
constexpr size_t LDS_TABLE = (size_t)H * 4;          // 128 KB
constexpr size_t lds_frames(int spl) { return (size_t)2 * CH * (4 * spl) * 4; }      // two chunks of CH frames, 4 * SPL symbols each
constexpr size_t LDS_MM = 128;
constexpr size_t lds_bytes(int spl) { return LDS_TABLE + lds_frames(spl) + LDS_MM; }

__device__ __forceinline__ double lse2f(double a, double b) {
    // 1 + e^x with x <= 0 is in [1, 2]: the bare v_exp_f32 / v_log_f32 (base 2) need none of the library forms' range fix-ups
    const double m = fmax(a, b), n = fmin(a, b);
    const float e = __builtin_amdgcn_exp2f(1.4426950408889634f * (float)(n - m));
    const double r = m + (double)(0.6931471805599453f * __builtin_amdgcn_logf(1.0f + e));
    return (m == -INFINITY) ? -INFINITY : r;
}

template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// max over the 16 lanes of every row, left in all of them (row_ror:8,4,2,1)
__device__ __forceinline__ unsigned row_allmax(unsigned v) {
    unsigned x = dpp_u32<0x128>(v); v = v > x ? v : x;
    x = dpp_u32<0x124>(v); v = v > x ? v : x;
    x = dpp_u32<0x122>(v); v = v > x ? v : x;
    x = dpp_u32<0x121>(v); v = v > x ? v : x;
    return v;
}
// the same over all 64 lanes with VALU operations only (gfx950's row swaps): nothing goes through the scalar unit, so the
// result can feed the next vector instruction of a round's chain directly
__device__ __forceinline__ unsigned wave_allmax_valu(unsigned v) {
    v = row_allmax(v);
    auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);      // rows (0,1) and (2,3) meet
    v = a[0] > a[1] ? a[0] : a[1];
    auto c = __builtin_amdgcn_permlane32_swap(v, v, false, false);      // the two halves meet
    return c[0] > c[1] ? c[0] : c[1];
}
// the same as ONE scalar: row maxima by DPP rotations, rows 1 / 3 take row 0's / 2's lane 15 (row_bcast:15), rows 2, 3 take lane 31 (row_bcast:31),
// lane 63 holds the maximum and v_readlane hands it to the scalar unit -- seven dependent instructions instead of ten, the compare that follows
// takes a scalar operand.  Measured in the pop rounds and NOT used there (round 5): flat 3.19 -> 3.37 ms, peaked 3.68 -> 3.83 at beam 16 -- the trip
// through the scalar unit costs the chain more than three vector instructions save.  Kept for the diagnostic record only.
template <int CTRL, int RM> __device__ __forceinline__ unsigned dpp_rows(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, RM, 0xF, false);
}
__device__ __forceinline__ unsigned wave_allmax_scalar(unsigned v) {
    v = row_allmax(v);
    // the two cross-row steps as fused v_max_u32_dpp (the builtin gives v_mov_b32_dpp + v_max_u32: a row mask other than 0xf is not folded);
    // rows outside the mask keep their value; the wait states a DPP read behind a vector write needs are written out
    asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// row 0's value of v in all four rows
__device__ __forceinline__ unsigned row0_to_all(unsigned v) {
    auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    auto c = __builtin_amdgcn_permlane32_swap(a[0], a[0], false, false);
    return c[0];
}
__device__ __forceinline__ unsigned wave_allmax(unsigned v) {
    v = row_allmax(v);
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// one 64-bit compare and four 32-bit selects per exchange (written as selects of 64-bit values the compiler emits a second compare, v_cmp_gt_u64
// beside v_cmp_lt_u64, for the other output: 19 more half-rate instructions per frame)
#define SB_CE(a, b) { const unsigned long long x_ = k[a], y_ = k[b]; const bool sw_ = x_ < y_;                                  \
                      const unsigned xl_ = (unsigned)x_, xh_ = (unsigned)(x_ >> 32), yl_ = (unsigned)y_, yh_ = (unsigned)(y_ >> 32);   \
                      k[a] = ((unsigned long long)(sw_ ? yh_ : xh_) << 32) | (sw_ ? yl_ : xl_);                                    \
                      k[b] = ((unsigned long long)(sw_ ? xh_ : yh_) << 32) | (sw_ ? xl_ : yl_); }

template <int SPL>
__global__ __launch_bounds__(64) void beam_small_kernel(
    const float* __restrict__ lp, long long stride_t, long long stride_b, const int32_t* __restrict__ lengths,
    int T, int V, int K, int blank, int collapse, int32_t* __restrict__ out_tokens, int32_t* __restrict__ out_len,
    double* __restrict__ out_score) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* table = reinterpret_cast<unsigned*>(smem);
    static_assert(SPL == 8 || SPL == 16, "8 or 16 symbols per lane");
    constexpr int VM = 4 * SPL, SB = (SPL == 8) ? 3 : 4;                  // symbols the four rows cover; bits of a slot number
    constexpr unsigned TKM = (unsigned)(VM << 5) - 1u;                     // largest first-touch rank (symbol << 5 | entry << 1 | bit)
    constexpr int LOWBITS = 4 + SB + (SPL == 8 ? 10 : 11);                 // [rank | slot | entry] at the bottom of a key: 17 / 19 bits
    using mask_t = typename std::conditional<SPL == 8, unsigned, unsigned long long>::type;      // one bit per symbol
    float* frames = reinterpret_cast<float*>(smem + LDS_TABLE);
    mask_t* mm = reinterpret_cast<mask_t*>(smem + LDS_TABLE + lds_frames(SPL));
    const int b = blockIdx.x, lane = threadIdx.x, q = lane >> 4, j = lane & 15;
    int Tb = lengths ? lengths[b] : T; Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);

    for (int i = lane; i < H / 4; i += 64) reinterpret_cast<uint4*>(table)[i] = make_uint4(0u, 0u, 0u, 0u);
#ifdef PGASR_BEAM_DIAG
    const long long dg_c0_ = clock64();
    const bool getenv_cycles_ = (collapse & 2) != 0;      // diagnostic: flags bit 2 of the entry point -> out_score = cycles per frame
#endif

    // entry state, replicated per row
    double pb = (j == 0) ? 0.0 : -INFINITY, pnb = -INFINITY, tot = (j == 0) ? 0.0 : -INFINITY;
    unsigned id = (j == 0) ? ROOT : BAD_ID, par = NONE;
    int last = -1;
    int nb = 1, root_pos = 0;

    const float* base = lp + (long long)b * stride_b;
    constexpr int FPI = 64 / VM, NPRE = CH / FPI;          // frames one wave instruction covers (2 / 1), instructions per chunk
    const int lsym = lane & (VM - 1), lhalf = lane / VM;
    float pre[NPRE];
    auto issue = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int t = t0 + FPI * i + lhalf;
            pre[i] = (lsym < V && t < Tb) ? base[(long long)t * stride_t + lsym] : -INFINITY;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) frames[buf * (CH * VM) + (FPI * i + lhalf) * VM + lsym] = pre[i];
    };
    issue(0); commit(0); issue(CH);
    __syncthreads();
    bool tied_last = false;      // the previous frame needed the exact rounds: two beam entries with bit-identical totals stay tied for as long as both
                                 // live (every extension of one meets the same extension of the other), so the speculative rounds are skipped

    for (int t = 0; t < Tb; ++t) {
        if ((t & (CH - 1)) == 0 && t > 0) {          // chunk boundary: the next chunk's rows have long arrived
            commit((t >> 5) & 1);
            issue(t + CH);
            __syncthreads();
        }
        const float* fr = frames + ((t >> 5) & 1) * (CH * VM) + (t & (CH - 1)) * VM;
        float lpv[SPL];
#pragma unroll
        for (int c = 0; c < SPL / 4; ++c) {
            const float4 f4 = *reinterpret_cast<const float4*>(fr + SPL * q + 4 * c);
            lpv[4 * c] = f4.x; lpv[4 * c + 1] = f4.y; lpv[4 * c + 2] = f4.z; lpv[4 * c + 3] = f4.w;
        }
        const float lp_bl = fr[blank];
        const float lp_la = fr[last >= 0 ? last : 0];

        // ---- where is my parent, which of my children are in the beam ----
        int pidx = -1;
        if (j < nb) {
            if (par == ROOT) pidx = root_pos;
            else if (par != NONE) pidx = (int)((table[par] >> 25) & 31u) - 1;
        }
        // DS operations of one wave execute in issue order: zero, atomic-or and read need no wait between them
        if (lane < 16) mm[lane] = (mask_t)0;
        if (lane < nb && pidx >= 0) atomicOr(&mm[pidx], (mask_t)1 << last);
        const mask_t mmask = __hip_atomic_load(&mm[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

        const int src = pidx >= 0 ? pidx : 0;
        const double pb_i = __shfl(pb, src, 16), tot_i = __shfl(tot, src, 16);
        const int last_i = __shfl(last, src, 16);

        // ---- stay candidate of entry j (:78-82, :90-96 merged, :103-106) ----
        const double npb = tot + (double)lp_bl;
        const bool has_last = last >= 0, has_par = pidx >= 0;
        const double dla = (double)lp_la;
        const double own = has_last ? pnb + dla : -INFINITY;                                    // repeat of the last symbol
        const double ext = (has_last && has_par) ? ((last_i == last) ? pb_i : tot_i) + dla : -INFINITY;   // parent i extended by it
        const double npnb = lse2f(ext, own);
        const unsigned lbits = (unsigned)(has_last ? last : 0) << 5;
        const unsigned t_bl = ((unsigned)blank << 5) | ((unsigned)j << 1);
        const unsigned t_own = has_last ? (lbits | ((unsigned)j << 1) | 1u) : 0xFFFFu;
        const unsigned t_e = (has_last && has_par) ? (lbits | ((unsigned)pidx << 1)) : 0xFFFFu;
        unsigned tie = t_bl < t_own ? t_bl : t_own;
        tie = tie < t_e ? tie : t_e;
        const double sstay = lse2f(npb, npnb);

        // ---- the SPL candidates of this lane as keys ----
        const long long tb0 = __double_as_longlong(tot);
        const double tot0 = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(tb0 >> 32), 0) << 32) |
                                                 (unsigned)__builtin_amdgcn_readlane((int)(unsigned)tb0, 0));
        const double ref = (tot0 == -INFINITY) ? 0.0 : tot0;
        unsigned long long k[SPL];
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
            const int s = SPL * q + i;
            const bool is_bl = (s == blank);
            const bool alive = (j < nb) && (s < V) && (is_bl || !((mmask >> s) & (mask_t)1));
            const double sx = ((s == last) ? pb : tot) + (double)lpv[i];
            const double sc = is_bl ? sstay : sx;
            const unsigned tk = is_bl ? tie : (((unsigned)s << 5) | ((unsigned)j << 1));
            const long long bits = __double_as_longlong(sc - ref);
            unsigned long long key = (unsigned long long)bits ^ ((unsigned long long)(bits >> 63) | 0x8000000000000000ull);
            key = (key & ~((1ull << LOWBITS) - 1ull)) | ((unsigned long long)(TKM - tk) << (4 + SB)) | ((unsigned long long)i << 4) | (unsigned long long)j;
            k[i] = alive ? key : 0ull;
        }
        if constexpr (SPL == 8) {
        // descending 8-input sorting network (19 exchanges)
        SB_CE(0, 1) SB_CE(2, 3) SB_CE(4, 5) SB_CE(6, 7)
        SB_CE(0, 2) SB_CE(1, 3) SB_CE(4, 6) SB_CE(5, 7)
        SB_CE(1, 2) SB_CE(5, 6) SB_CE(0, 4) SB_CE(3, 7)
        SB_CE(1, 5) SB_CE(2, 6)
        SB_CE(1, 4) SB_CE(3, 6)
        SB_CE(2, 4) SB_CE(3, 5)
        SB_CE(3, 4)
        } else {
        // descending 16-input network: Batcher's odd-even merge sort, 63 exchanges in 10 layers (checked on all 2^16 0/1 inputs)
        SB_CE(0, 1) SB_CE(2, 3) SB_CE(0, 2) SB_CE(1, 3) SB_CE(1, 2) SB_CE(4, 5) SB_CE(6, 7) SB_CE(4, 6) SB_CE(5, 7) SB_CE(5, 6)
        SB_CE(0, 4) SB_CE(2, 6) SB_CE(2, 4) SB_CE(1, 5) SB_CE(3, 7) SB_CE(3, 5) SB_CE(1, 2) SB_CE(3, 4) SB_CE(5, 6)
        SB_CE(8, 9) SB_CE(10, 11) SB_CE(8, 10) SB_CE(9, 11) SB_CE(9, 10) SB_CE(12, 13) SB_CE(14, 15) SB_CE(12, 14) SB_CE(13, 15) SB_CE(13, 14)
        SB_CE(8, 12) SB_CE(10, 14) SB_CE(10, 12) SB_CE(9, 13) SB_CE(11, 15) SB_CE(11, 13) SB_CE(9, 10) SB_CE(11, 12) SB_CE(13, 14)
        SB_CE(0, 8) SB_CE(4, 12) SB_CE(4, 8) SB_CE(2, 10) SB_CE(6, 14) SB_CE(6, 10) SB_CE(2, 4) SB_CE(6, 8) SB_CE(10, 12)
        SB_CE(1, 9) SB_CE(5, 13) SB_CE(5, 9) SB_CE(3, 11) SB_CE(7, 15) SB_CE(7, 11) SB_CE(3, 5) SB_CE(7, 9) SB_CE(11, 13)
        SB_CE(1, 2) SB_CE(3, 4) SB_CE(5, 6) SB_CE(7, 8) SB_CE(9, 10) SB_CE(11, 12) SB_CE(13, 14)
        }
        // the lane's sorted list stays in registers; a pop shifts it (14 v_mov under a one-lane exec mask: no LDS round trip,
        // no wait on the K-round chain)

        // ---- K rounds: pop the winners in rank order ----
        // Round 5: a round used to be all-reduce (4 DPP) -> 4 readlanes -> 3 scalar max -> compare -> ballot -> ffs -> readlane of the
        // winner's low word -> 16 moves under a one-lane exec mask, ~420 cycles of mostly scalar-unit round trips, sixteen times per
        // frame (60 % of it).  Now the chain of a round is vector-only: all-reduce of the lists' HIGH words (4 DPP + 2 row swaps),
        // compare, and a shift of the eight high words of the lanes that won (v_cndmask on the compare's mask).  What a round
        // leaves behind -- which lane won (s_ff1 of the mask -> v_writelane into lane r) and a won-bit per lane -- is scalar work off
        // the chain; the winners' identities are put together AFTER the rounds: lane r fetches its winner lane's won-bits and sort
        // permutation with two ds_bpermute and finds the slot as the (number of earlier wins of that lane)-th of its sorted list.
        // A round in which two lanes share the maximal high word (equal to 2^-20 relative) makes both pop: the scalar tallies disagree and
        // the frame is redone by the exact loop below -- 0.1-0.8 % of the frames on random and model-like log-probs (make beamdiag,
        // tools/dev/r5_beam_ties.py).  Except: two beam entries with BIT-IDENTICAL totals stay tied for as long as both live (every extension of
        // one meets the same extension of the other); one such utterance of 32 fell back in 871 of its 1000 frames and set the kernel's time
        // (5.47 ms against 4.07 for the same distribution with another seed).  So a frame that follows a tied frame goes straight to the exact
        // rounds (tied_last), and those use the vector-only all-reduce as well: 4.71 ms on that data (tools/dev/r5_beam_var.py).
        // (Measured and not kept in round 4: TWO winners per round from one all-reduce over (first, second) pairs of high words.)
        unsigned packed = 0u;
        int nnew = 0;
        {
            unsigned hq[SPL];
            unsigned perm[2] = {0u, 0u};           // slot of the d-th key of the sorted list, SB bits each, eight per word
#pragma unroll
            for (int d = 0; d < SPL; ++d) { hq[d] = (unsigned)(k[d] >> 32); perm[d >> 3] |= (((unsigned)k[d] >> 4) & (unsigned)(SPL - 1)) << (SB * (d & 7)); }
            unsigned won = 0u, wl_of = 0u;
            int pops = 0, rounds = 0;
            bool need_exact = tied_last;                        // straight to the exact rounds
            if (!need_exact) {
            // sixteen rounds written out (default) or a rolled loop (-DPGASR_BEAM_UNROLL=0): written out, the compiler keeps a 64-bit "K > r" mask
            // per round alive across the frame loop (106 SGPRs) -- but the rolled loop is no faster: same box, A/B/A/B by library
            // (tools/dev/r5_beam_unroll.sh), beam 16 flat 3.37 written out / 3.55 rolled, peaked 3.91 / 4.09, beam 5 2.34 / 2.29
#if PGASR_BEAM_UNROLL
#define SB_ROUND_BODY(r, RC)                                                                                             \
            {                                                                                                            \
                const unsigned hh = hq[0];                                                                               \
                const unsigned smax = wave_allmax_valu(hh);                                                              \
                const bool win = (hh == smax) && (smax != 0u);                                                           \
                const unsigned long long m = __ballot(hh == smax) & __ballot(smax != 0u);   /* two compare masks: no 0/1 round trip */ \
                pops += __popcll(m); rounds += (m != 0ull) ? 1 : 0;                                                      \
                {   /* lane r of wl_of := the winner lane (a scalar); v_writelane has no builtin in this compiler */     \
                    const int wls = __ffsll((long long)m) - 1;                                                           \
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(wl_of) : "s"(wls), RC(r));                          \
                }                                                                                                        \
                won |= win ? (1u << (r)) : 0u;                                                                           \
                _Pragma("unroll") for (int d = 0; d < SPL - 1; ++d) hq[d] = win ? hq[d + 1] : hq[d];                     \
                hq[SPL - 1] = win ? 0u : hq[SPL - 1];                                                                    \
            }
#define SB_ROUND(r) if (K > (r)) SB_ROUND_BODY(r, "n")
            SB_ROUND(0) SB_ROUND(1) SB_ROUND(2) SB_ROUND(3) SB_ROUND(4) SB_ROUND(5) SB_ROUND(6) SB_ROUND(7)
            SB_ROUND(8) SB_ROUND(9) SB_ROUND(10) SB_ROUND(11) SB_ROUND(12) SB_ROUND(13) SB_ROUND(14) SB_ROUND(15)
#undef SB_ROUND
            static_assert(K_MAX == 16, "sixteen rounds are written out");
#else
#pragma unroll 1
            for (int r = 0; r < K; ++r) {
                const unsigned hh = hq[0];
                const unsigned smax = wave_allmax_valu(hh);
                const bool win = (hh == smax) && (smax != 0u);
                const unsigned long long m = __ballot(win);
                pops += __popcll(m); rounds += (m != 0ull) ? 1 : 0;
                {   // lane r of wl_of := the winner lane (a scalar); v_writelane has no builtin in this compiler
                    const int wls = __ffsll((long long)m) - 1;
                    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(wl_of) : "s"(wls), "s"(r) : "m0");      // two scalar operands: the lane select goes through M0
                }
                won |= win ? (1u << r) : 0u;
#pragma unroll
                for (int d = 0; d < SPL - 1; ++d) hq[d] = win ? hq[d + 1] : hq[d];
                hq[SPL - 1] = win ? 0u : hq[SPL - 1];
            }
#endif
                need_exact = pops != rounds;
            }
            tied_last = false;
            if (!need_exact) {
                nnew = rounds;
                const unsigned wl = row0_to_all(wl_of);                       // lane (q, j): the winner lane of round j
                const unsigned wwon = (unsigned)__shfl((int)won, (int)(wl & 63u), 64);
                unsigned wperm = (unsigned)__shfl((int)perm[0], (int)(wl & 63u), 64);
                const int idx = __popc(wwon & ((1u << j) - 1u));               // how many rounds before round j that lane had won
                if constexpr (SPL == 16) {
                    const unsigned wperm1 = (unsigned)__shfl((int)perm[1], (int)(wl & 63u), 64);
                    wperm = (idx & 8) ? wperm1 : wperm;
                }
                const unsigned slot = (wperm >> (SB * (idx & 7))) & (unsigned)(SPL - 1);
                packed = ((wl >> 4) << (4 + SB)) | (slot << 4) | (wl & 15u);
            } else {
#ifdef PGASR_BEAM_DIAG
                if (lane == 0) atomicAdd(&beam_diag_counters[1], 1ull);
#endif
                // exact rounds on the full keys (ties between high words; rare)
                for (int r = 0; r < K; ++r) {
                    const unsigned long long head = k[0];
                    const unsigned hh = (unsigned)(head >> 32);
                    const unsigned smax = wave_allmax_valu(hh);
                    if (smax == 0u) break;
                    unsigned long long m = __ballot(hh == smax);
                    if (__popcll(m) != 1) {
                        tied_last = true;
                        const unsigned ll = (hh == smax) ? (unsigned)head : 0u;
                        const unsigned smaxlo = wave_allmax_valu(ll);
                        m = __ballot(hh == smax && (unsigned)head == smaxlo);
                    }
                    const int wl = __ffsll((long long)m) - 1;
                    const unsigned wlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)head, wl);
                    const unsigned pk = ((unsigned)(wl >> 4) << (4 + SB)) | (wlo & ((1u << (4 + SB)) - 1u));
                    if (j == r) packed = pk;
                    if (lane == wl) {
#pragma unroll
                        for (int d = 0; d < SPL - 1; ++d) k[d] = k[d + 1];
                        k[SPL - 1] = 0ull;
                    }
                    ++nnew;
                }
            }
        }

#ifdef PGASR_BEAM_DIAG
        if (lane == 0) atomicAdd(&beam_diag_counters[0], 1ull);
#endif
        // ---- the new beam: entry r <- winner r ----
        const int pj = (int)(packed & 15u), slot = (int)((packed >> 4) & (unsigned)(SPL - 1)), pq = (int)(packed >> (4 + SB));
        const int s = SPL * pq + slot;
        const bool valid = j < nnew;
        const bool stay = (s == blank);
        const double g_tot = __shfl(tot, pj, 16), g_pb = __shfl(pb, pj, 16);
        const double g_npb = __shfl(npb, pj, 16), g_npnb = __shfl(npnb, pj, 16), g_sstay = __shfl(sstay, pj, 16);
        const unsigned g_id = (unsigned)__shfl((int)id, pj, 16), g_par = (unsigned)__shfl((int)par, pj, 16);
        const int g_last = __shfl(last, pj, 16);
        const float lps = fr[valid ? s : 0];

        // beam-position bits of the outgoing entries are cleared before the incoming ones are set
        if (lane < nb && id != ROOT) atomicAnd(&table[id], KEYMASK);
        unsigned n_id = BAD_ID;
        if (lane < 16 && valid) {
            if (stay) n_id = g_id;
            else {
                // canonical node of (parent id, symbol): find, else insert
                const unsigned key = ((g_id << 8) | (unsigned)s) + 1u;
                // Triangular probing (h + 1, + 2, + 3, ..: visits every slot of a power-of-two table) behind a two-round mix.  Round 5: the table
                // holds every prefix an utterance ever had in its beam (up to T * beam of 32768 slots), and with linear probing behind one
                // multiplicative round the chains clustered -- up to 24-30 probes, each a dependent LDS round trip the whole wave waits for,
                // and the kernel's time followed the unluckiest utterance (same distribution, other seed: 4.07 against 5.47 ms at beam 16).
                unsigned h = key * 2654435761u;
                h ^= h >> 15; h *= 0x2C1B3C6Du;
                h >>= 17;
                for (int guard = 0; guard < H; ++guard) {
#ifdef PGASR_BEAM_DIAG
                    atomicAdd(&beam_diag_counters[2], 1ull);
                    atomicMax(&beam_diag_counters[3], (unsigned long long)(guard + 1));
#endif
                    const unsigned v = table[h];
                    if ((v & KEYMASK) == key) { n_id = h; break; }
                    if (v == 0u) {
                        const unsigned old = atomicCAS(&table[h], 0u, key);
                        if (old == 0u || (old & KEYMASK) == key) { n_id = h; break; }
                    }
                    h = (h + (unsigned)guard + 1u) & (unsigned)(H - 1);
                }
            }
        }
        n_id = (unsigned)__shfl((int)n_id, j, 64);
        if (lane < 16 && valid && n_id != ROOT && n_id != BAD_ID) atomicOr(&table[n_id], (unsigned)(lane + 1) << 25);
        const unsigned long long rootm = __ballot(lane < 16 && valid && n_id == ROOT);
        root_pos = rootm ? (__ffsll((long long)rootm) - 1) : -1;

        if (valid) {
            if (stay) { pb = g_npb; pnb = g_npnb; tot = g_sstay; last = g_last; par = g_par; }
            else {
                pb = -INFINITY;
                pnb = ((s == g_last) ? g_pb : g_tot) + (double)lps;
                tot = pnb; last = s; par = g_id;
            }
            id = n_id;
        } else { pb = pnb = tot = -INFINITY; last = -1; par = NONE; id = BAD_ID; }
        nb = nnew;
    }

    // ---- result: ancestors of the best entry, in order, optionally through collapse_fn ----
    __syncthreads();
    unsigned short* tmp = reinterpret_cast<unsigned short*>(frames);      // 8 KB = 4096 tokens: the dispatch admits T <= MAX_TOKENS only
    unsigned cur = (unsigned)__builtin_amdgcn_readlane((int)id, 0);
    int n = 0;
    if (nb > 0) {
        while (cur != ROOT && cur < (unsigned)H && n < T) {
            const unsigned v = (table[cur] & KEYMASK) - 1u;
            if (lane == 0) tmp[n] = (unsigned short)(v & 255u);
            ++n;
            cur = v >> 8;
        }
    }
    __syncthreads();
    int32_t* o = out_tokens + (size_t)b * T;
    int outn = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        int tok = 0; bool keep = false;
        if (i < n) {
            tok = tmp[n - 1 - i];
            keep = !(collapse & 1) || i == 0 || tok != (int)tmp[n - i];
        }
        const unsigned long long m = __ballot(keep);
        if (keep) o[outn + __popcll(m & ((1ull << lane) - 1ull))] = tok;
        outn += __popcll(m);
    }
    if (lane == 0) {
        out_len[b] = outn;
        const double t0 = __shfl(tot, 0, 64);
        out_score[b] = (Tb > 0) ? -t0 : -0.0;
#ifdef PGASR_BEAM_DIAG
        if (getenv_cycles_) out_score[b] = (double)(clock64() - dg_c0_) / (double)(Tb > 0 ? Tb : 1);      // cycles per frame of this utterance
#endif
    }
}
#undef SB_CE
}  // namespace sb

}  // namespace

#ifdef PGASR_BEAM_DIAG
extern "C" int pgasr_diag_beam_counters(unsigned long long* out, int reset) {      // host copy of (frames, frames redone by the exact rounds)
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(beam_diag_counters), sizeof(unsigned long long) * 4) != hipSuccess) return 1;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(beam_diag_counters), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" size_t pgasr_beam_workspace_bytes(int T, int B, int V, int beam) {
    if (T <= 0 || B <= 0 || V <= 0 || beam <= 0) return 0;
    return beam_ws_layout(T, B, beam, nullptr, nullptr);
}

extern "C" int pgasr_ctc_beam_search(const void* log_probs, int is_f64, long long stride_t, long long stride_b,
                                     const int32_t* lengths, int T, int B, int V, int beam, int blank, int flags,
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
#ifdef PGASR_BEAM_DIAG
    const int collapse = (flags & 1) | ((flags & 4) ? 2 : 0);
#else
    const int collapse = flags & 1;
#endif
    if (!is_f64 && !(flags & 2) && beam <= sb::K_MAX && V <= sb::V_MAX && (long long)T * beam <= sb::MAX_NODES && T <= sb::MAX_TOKENS) {
        // training path: one wave per utterance, trie and candidate lists in LDS, no workspace traffic; 8 symbols per lane up to V = 32
        // (the English alphabet of the headline), 16 up to V = 64 (round 5: CommonVoice's larger alphabets stay on this kernel)
        auto kern = V <= 32 ? &sb::beam_small_kernel<8> : &sb::beam_small_kernel<16>;
        const size_t lds_small = sb::lds_bytes(V <= 32 ? 8 : 16);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small);
        PGASR_LAUNCH_KERNEL(kern, dim3(B), dim3(64), lds_small, st, (const float*)log_probs, stride_t, stride_b,
                           lengths, T, V, beam, blank, collapse, out_tokens, out_len, out_score);
        PGASR_CHECK_LAUNCH();
        return PGASR_OK;
    }
    if (hipMemsetAsync(ws.table, 0, (size_t)B * ws.H * sizeof(unsigned long long), st) != hipSuccess) return PGASR_ERR_LAUNCH;
    const size_t lds = beam_lds_bytes(beam, V);
    if (lds > 160 * 1024) return PGASR_ERR_UNSUPPORTED;
    if (is_f64) {   // exact path: fp64 transcendentals (drop-in CTCDecoder.decode)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&beam_search_kernel<double, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        PGASR_LAUNCH_KERNEL((beam_search_kernel<double, false>), dim3(B), dim3(BEAM_THREADS), lds, st, (const double*)log_probs,
                           stride_t, stride_b, lengths, T, V, beam, blank, collapse, ws, out_tokens, out_len, out_score);
    } else {        // fp32 device log-probs: fast transcendentals
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&beam_search_kernel<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        PGASR_LAUNCH_KERNEL((beam_search_kernel<float, true>), dim3(B), dim3(BEAM_THREADS), lds, st, (const float*)log_probs,
                           stride_t, stride_b, lengths, T, V, beam, blank, collapse, ws, out_tokens, out_len, out_score);
    }
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}
