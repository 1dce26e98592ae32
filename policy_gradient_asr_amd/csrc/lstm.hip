// Bidirectional LSTM recurrence for gfx950 (model.py:39-44,52-55: 3 layers x 2 directions,
// H = 256, packed-sequence semantics).  The input projections X*W_ih^T are hoisted into one big
// MFMA GEMM per layer (gemm.hip); this file is the serial part:
//     gates_t = xproj_t + h_{t-1} W_hh^T ; i,f,o = sigmoid, g = tanh ; c_t = f c + i g ; h_t = o tanh c_t
// and its reverse-time gradient.
//
// Design (DESIGN.md "LSTM recurrence"): W_hh (1 MB fp32) does not fit one CU, so each
// (direction, 16-utterance batch group) is served by a CLUSTER of 16 persistent workgroups, one
// per CU, each owning 16 hidden units, with its weight slice resident in registers (64 VGPRs of
// bf16 hi/lo pairs per lane) for the whole sweep.  Matrix products run on
// v_mfma_f32_16x16x32_bf16 as hi*hi + hi*lo + lo*hi with fp32 accumulation (~15-16 mantissa bits
// per operand; measured 1e-5 relative on layer outputs against fp64).  The kernels are templates on the
// number of bf16 PLANES an fp32 operand is split into: NP = 2 is that default; NP = 3 (flags bit 1, the
// "f32" precision mode of the host layer) is the fp32-faithful variant the reference's arithmetic asks
// for (model.py:38-44: nn.LSTM in torch's default fp32): hi/mid/lo planes and the six products
// hh + hm + mh + hl + lh + mm, i.e. every term down to 2^-24 of the product -- 96 weight VGPRs
// per lane instead of 64, 48 MFMAs per step instead of 24 (+~400 cycles), 24 KiB instead of 16 KiB polled
// per forward step.  Per step the members
// exchange h_t (forward, 16 KiB) or partial dh sums (backward, 16 KiB read per member) through
// global memory using self-validating words (below): no flag, counter, fence or drain sits on
// the dependent chain, results cannot depend on dispatch order, XCD placement or the order in
// which stores become visible, and every spin is bounded (err word + exit).
// Placement is used for SPEED only: the grid is laid out so that a cluster's members share
// blockIdx % 8 (one XCD under the observed round-robin dispatch); the members check at run time
// (HW_REG_XCC_ID) whether they really share an XCD and then publish with plain stores that stay
// in that XCD's L2, otherwise with write-through (sc1) stores.  Loads always bypass L1 (sc1).
// The rows a step needs from HBM (xproj / saved activations, c_t, dout) never enter a compute
// workgroup's memory pipeline cold: N_HELPERS helper workgroups per cluster copy them, steps
// ahead, into an L2-resident ring inside the workspace (helper_loop below); the compute
// workgroups' loader waves DMA from that ring.
#include "common.h"

namespace {

constexpr int HID = 256;           // hidden size per direction (model.py:40)
constexpr int G_CLUSTER = 16;      // compute workgroups per (direction, batch group): 16 units each
#ifndef PGASR_N_HELPERS
#define PGASR_N_HELPERS 4
#endif
constexpr int N_HELPERS = PGASR_N_HELPERS;       // + helper workgroups per cluster that stage the sweep's HBM rows into a hot ring
#ifndef PGASR_FWD_RING_STEPS
#define PGASR_FWD_RING_STEPS 16
#endif
#ifndef PGASR_BWD_RING_STEPS
#define PGASR_BWD_RING_STEPS 12
#endif
// ring depth per sweep (steps).  Stand-alone sweeps (round 1): forward 8 -> 1.15 us/step, 16 -> 1.23, 32 -> 1.33 (a smaller ring
// stays in L2); backward 8 -> 1.71, 16 -> 1.62 (its loader runs 4 steps ahead and needs the slack).
// Round 4: the BACKWARD ring is 12 steps.  Its slots are 96 KB (gates + c + dout), and beside the 512 KB exchange slots a 16-step ring
// (1.5 MB per cluster) did not survive in the XCD's 4 MB L2 between its uses: every staged line was written back to HBM once -- the
// backward sweep's WRITE_SIZE was 538-675 MB per launch against 262 MB of dgates (round 1-3 verdicts).  Same box, A/B by library
// (tools/dev/r4_ring.sh): 16 steps 538 MB, 12 steps 287 MB, 10 steps 269 MB; stand-alone backward sweep 1.130 / 1.130 / 1.160 ms
// (bf16x3), 1.298 / 1.287 / 1.297 (f32); f32 step 10.22 / 10.18 / 10.18 ms.  (Round 2's 8-step ring cut the traffic too but cost the
// sweep 30 %: the helpers lose the slack that hides a slow DRAM burst; 12 keeps it.)
constexpr int FWD_RING_STEPS = PGASR_FWD_RING_STEPS, BWD_RING_STEPS = PGASR_BWD_RING_STEPS;
constexpr int RING_STEPS_MAX = 32;     // depth of that ring (steps)
constexpr int FWD_STEP_FLOATS = 16 * 1024;                           // 16 utterances x (256 units x 4 gates)
constexpr int BWD_STEP_FLOATS = 16 * 1024 + 16 * 256 + 16 * 256;     // + c_t + dout
constexpr int LSTM_THREADS = 384;     // 4 compute waves + loader wave + storer wave
constexpr int HELLO_STRIDE = 64;      // words per cluster in the start-up block
constexpr int STAMP_MAX_T = 4096;
constexpr long long SPIN_TIMEOUT_TICKS = 300000000LL;  // 3 s of the 100 MHz realtime counter

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned short f2bf(float x) {   // round to nearest even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(unsigned short, (__bf16)x);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ void split_bf16(float x, unsigned short& hi, unsigned short& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}
// v_exp_f32 / v_rcp_f32 (1 ulp each): a correctly rounded divide costs ~10 instructions per gate
__device__ __forceinline__ float sigmoidf_fast(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_fast(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.885390081777927f * x));
}

struct LstmArgs {
    float* gates;            // [T][B][2][H][4]: in = permuted xproj (+biases); out = activations i,f,g,o;
                             // after the backward sweep: d(pre-activations)
    float* out;              // [T][B][2H]   h_t (zeros past each length)
    float* out_drop;         // (forward, optional) [T][B][2H] = dropout(out): the next layer's input (model.py:42), written by the storer wave
    float* cbuf;             // [T][B][2][H] c_t
    const float* dout;       // [T][B][2H]   (backward) gradient w.r.t. out
    float* dbias_part;       // (backward, optional) [ceil(B/16)][2][H][4]: per 16-utterance group sums over t of dgates
    const u32x4* wpack;      // packed bf16 hi/lo W_hh in MFMA A-operand order (see pack kernel)
    unsigned char* xbuf;     // exchange buffers, pre-filled with the "stale" pattern
    unsigned* hello;         // [clusters][HELLO_STRIDE] words, zeroed per call: 0..15 start-up words (XCC id of each member), 16 the
                             // flusher's, 32..47 "stored" (slabs whose dgates stores a member's storer wave has seen acknowledged)
    unsigned* progress;      // [clusters] current step of member 0 (paces the helpers), zeroed per call
    float* ring;             // [clusters][RING_STEPS][step floats]: the rows of the next steps, staged by the helper workgroups
    unsigned* ready;         // [clusters][32]: word i = (step + 1) held by ring slot i (0 = nothing yet), zeroed per call
    int n_helpers;           // helper workgroups per cluster: N_HELPERS, or 0 when two clusters must share an XCD (B > 64)
    const unsigned* fed;     // (forward, optional) [2][fed_mt] finished-tile counts of an input projection that runs BESIDE this sweep
    int fed_mt, fed_need;    // row tiles of 256 (t, b) rows; a row tile is complete at fed_need (gemm_dma.hip, FEED kernel)
                             // backward: the fed tensor is dout = the input gradient of the layer above, optionally still
                             // WITHOUT the inter-layer dropout mask, which the helpers then apply while they stage its rows
    unsigned drop_thresh, drop_k0, drop_k1, drop_off; float drop_scale; int drop_on;
    unsigned* slab_done;     // (backward, optional) [clusters]: publications so far -- publication k says that the dgates rows of sweep steps
                             // < T - pgasr_wslab_edge(T, n - k) are IN MEMORY, readable by agent-scope loads from any XCD while the sweep
                             // runs (common.h: the time slabs of the weight-gradient products; published by the cluster's flusher workgroup)
    int* err;                // set to 1 when a bounded wait gives up
    unsigned* busy;          // [8] per-XCD count of clusters currently sweeping there (read by queue-mode GEMMs)
    const int* lengths;      // [B]
    int T, B, NBG, NCL8;
    int force_mode;          // 0 auto, 1 force write-through (cross-XCD safe), for tests
    int diag;                // diagnostic timing switches (results invalid): bit0 skip bulk stores, bit1 skip xproj prefetch,
                             // bit2 take the first poll as it comes (no tag check), bit3 no exchange loads at all, bit4 no publish,
                             // bit5 the loader waits for the staging ring but issues no LDS-DMA, bit6 LDS-DMA without waiting for the ring
    long long* stamps;       // diagnostic build only (-DPGASR_LSTM_STAMPS)
};

#ifndef PGASR_PUBLISH_BRANCHFREE
#define PGASR_PUBLISH_BRANCHFREE 1
#endif
// (__builtin_expect on the publish protocol's if / else -- one taken branch per store instead of two -- was measured, library A/B: the
// sweeps alone gain up to 1.5 %, the f32 train step LOSES 0.2 ms, twice; code placement matters at this granularity.  Not kept.)
#ifdef PGASR_LSTM_STAMPS
// Diagnostic build: wave 0 of every workgroup accumulates, in registers, the cycles between consecutive stamps (segment
// ending at `slot`); one store per workgroup at the end (member 5 of cluster 0 is read by tools/dev/tools_stamps.py).  No memory
// traffic in the loop: per-step stores of the stamps sat in front of the wave's next s_waitcnt vmcnt(0) and distorted the picture.
#define STAMP_DECL long long st_last_ = clock64(); long long st_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
                   const long long st_c0_ = clock64(), st_r0_ = wall_clock64()
#define STAMP(slot) do { if (w == 0) { const long long now_ = clock64(); st_acc_[slot] += now_ - st_last_; st_last_ = now_; } } while (0)
#define STAMP_FLUSH() do { if (g == 5 && bg == 0 && dir == 0 && tid == 0) { for (int i_ = 0; i_ < 8; ++i_) a.stamps[i_] = st_acc_[i_]; \
                           a.stamps[8] = clock64() - st_c0_; a.stamps[9] = wall_clock64() - st_r0_; } } while (0)
#else
#define STAMP_DECL
#define STAMP_FLUSH() do { } while (0)
#define STAMP(slot) do { } while (0)
#endif

// Diagnostic build (-DPGASR_LSTM_DIAG): what the loader wave's wait for the staging ring costs.  Per cluster, hello words 48..51 (member 5)
// and 52..55 (member 0): waits whose first poll found the slot not ready, retries in total, cycles spent in ring_wait, steps.
#ifdef PGASR_LSTM_DIAG
// .. and WHERE in the sweep member 0's loader waited (round 5, tools/dev/r5_feed_timeline.py): per cluster the first 16 late steps as
// (step, cycles / 16, 100 MHz wall clock at the end of the wait), word 60 = wall clock when the loader first asked for a step
__device__ unsigned lstm_diag_late[8][64];
#define RW_DECL unsigned rw_late_ = 0, rw_retry_ = 0, rw_steps_ = 0; long long rw_cyc_ = 0, rw_t0_ = 0; bool rw_first_ = true
#define RW_BEGIN() do { rw_t0_ = clock64(); rw_first_ = true; if (rw_steps_ == 0 && g == 0) lstm_diag_late[cl & 7][60] = (unsigned)wall_clock64();   \
                        if (g == 0 && (s & 127) == 0 && (s >> 7) < 8) { lstm_diag_late[cl & 7][48 + (s >> 7)] = (unsigned)wall_clock64();    /* the loader's request for steps 0, 128, .., 896 */ \
                                                                        lstm_diag_late[cl & 7][56 - 16 + (s >> 7)] = (unsigned)clock64(); }    /* .. and the shader clock's count there (words 40..47) */ \
                        ++rw_steps_; } while (0)
#define RW_RETRY() do { if (rw_first_) { ++rw_late_; rw_first_ = false; } ++rw_retry_; } while (0)
#define RW_END() do { const long long d_ = clock64() - rw_t0_; rw_cyc_ += d_;                                                   \
                      if (!rw_first_ && g == 0 && rw_late_ <= 16) { unsigned* o_ = &lstm_diag_late[cl & 7][3 * (rw_late_ - 1)];  \
                          o_[0] = (unsigned)s; o_[1] = (unsigned)(d_ >> 4); o_[2] = (unsigned)wall_clock64(); } } while (0)
#define RW_FLUSH() do { if (w == LOADER_WAVE && lane == 0 && (g == 5 || g == 0)) { unsigned* o_ = a.hello + (size_t)cl * HELLO_STRIDE + (g == 5 ? 48 : 52); \
                        o_[0] = rw_late_; o_[1] = rw_retry_; o_[2] = (unsigned)(rw_cyc_ >> 4); o_[3] = rw_steps_; } } while (0)
#else
#define RW_DECL
#define RW_BEGIN() do { } while (0)
#define RW_RETRY() do { } while (0)
#define RW_END() do { } while (0)
#define RW_FLUSH() do { } while (0)
#endif

// ---- self-validating exchange words -------------------------------------------------------
// Every 32-bit word of an exchange buffer carries the epoch of the step that produced it in
// bits that are numerically harmless: e = (step>>1)&1 and
//   * bf16-pair words (forward h):  (bit0, bit16) = (e, 1-e)   -- the mantissa LSBs of the pair;
//   * fp32 words (backward partial sums): (bit0, bit1) = (e, 1-e).
// A slot (step&1) is rewritten every second step, so a word still holding the previous round
// shows the complementary pattern: the bit that must be 0 in fresh data is 1.  A consumer ORs
// the words it loaded and tests that one bit -- no flag, counter, fence or drain sits on the
// dependent chain and no ordering between stores is assumed: each aligned dword is written
// atomically and validates itself (the guide's R2 "the data IS the flag", at word granularity).
// For bf16 pairs the hi part is rounded with its LSB forced and the lo part is the residual to
// THAT value, so hi+lo still carries ~15 mantissa bits and the tag never needs masking.
__device__ __forceinline__ void split_tagged(float x, unsigned tb, unsigned short& hi, unsigned short& lo) {
    const unsigned short h = (unsigned short)((f2bf(x) & 0xFFFEu) | tb);
    const unsigned short l = (unsigned short)((f2bf(x - bf2f(h)) & 0xFFFEu) | tb);
    hi = h; lo = l;
}
__device__ __forceinline__ void split_plain(float x, unsigned short& hi, unsigned short& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}
// NP planes: plane p = bf16 of what the planes before it left over (each subtraction is exact in fp32)
template <int NP>
__device__ __forceinline__ void split_planes_tagged(float x, unsigned tb, unsigned short (&pl)[NP]) {
    float r = x;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        pl[p] = (unsigned short)((f2bf(r) & 0xFFFEu) | tb);
        r -= bf2f(pl[p]);
    }
}
template <int NP>
__device__ __forceinline__ void split_planes_plain(float x, unsigned short (&pl)[NP]) {
    float r = x;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        pl[p] = f2bf(r);
        r -= bf2f(pl[p]);
    }
}
// acc += A x B for operands given as NP bf16 planes: every product term of weight >= 2^-24 relative.
// NP = 2: hh + hl + lh (the order the sweeps have always used); NP = 3: hh + hm + mh + hl + lh + mm.
template <int NP>
__device__ __forceinline__ void mfma_planes_pair(const bf16x8 (&A0)[NP], const bf16x8 (&A1)[NP], const bf16x8 (&Bp)[NP], f32x4 (&acc)[2]) {
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0[0], Bp[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[0], Bp[0], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0[0], Bp[1], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[0], Bp[1], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0[1], Bp[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[1], Bp[0], acc[1], 0, 0, 0);
    if constexpr (NP == 3) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0[0], Bp[2], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[0], Bp[2], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0[2], Bp[0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[2], Bp[0], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0[1], Bp[1], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[1], Bp[1], acc[1], 0, 0, 0);
    }
}
__device__ __forceinline__ unsigned or4(u32x4 v) { return (v.x | v.y) | (v.z | v.w); }
// non-zero iff some word's (bit0, bit16) differs from the wanted pattern
__device__ __forceinline__ unsigned bad4(u32x4 v, unsigned want) {
    return (((v.x ^ want) | (v.y ^ want)) | ((v.z ^ want) | (v.w ^ want))) & 0x00010001u;
}

// Every polling attempt starts with this: the exchange buffers are written by OTHER workgroups, which the
// compiler cannot see.  Without it the (side-effect free) buffer loads may legally be hoisted out of the
// retry loop, or the loop deleted outright (observed on a microbenchmark with hipcc 7.2).
#define POLL_FENCE() asm volatile("" ::: "memory")

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also drains vmcnt (it is a
// workgroup-scope fence), i.e. every wave would wait at every step for all of its outstanding global
// loads and stores -- for the I/O wave that is a DRAM round trip per step.  Global data here is ordered
// by the exchange protocol itself, never by this barrier.
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ void release_xcd(const LstmArgs& a, int g, int tid) {
    if (g == 0 && tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xFu;
        __hip_atomic_fetch_add(a.busy + xcc, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // -1
    }
}

// Workgroup flags in LDS.  They were `volatile int` reached through generic pointers: hipcc then emits FLAT loads with
// sc0 sc1 followed by s_waitcnt vmcnt(0) -- in the step loop that was one flat round trip per step on the chain AND a
// drain of every outstanding global store / LDS-DMA of the loader and storer waves (whose whole point is never to
// wait for memory).  Relaxed workgroup-scope atomics on plain __shared__ ints compile to ds_read_b32 / ds_write_b32.
#define LDS_FLAG_GET(var) __hip_atomic_load(&(var), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LDS_FLAG_SET(var, val) __hip_atomic_store(&(var), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

struct SpinGuard {
    unsigned spins = 0; long long t0 = 0;
    // returns false when the wait has lasted longer than SPIN_TIMEOUT_TICKS
    __device__ __forceinline__ bool keep_waiting() {
        // the realtime clock is read only on long waits (never in the common path)
        if (((++spins) & 1023u) == 0) {
            const long long now = wall_clock64();
            if (spins == 1024u) t0 = now;
            else if (now - t0 > SPIN_TIMEOUT_TICKS) return false;
        }
        return true;
    }
};

// Cluster start-up: every member publishes the XCD it runs on (HW_REG_XCC_ID); all members read
// the same 16 words, so they reach the same verdict.  true = the whole cluster shares one XCD
// (one L2): payload stores may stay in that L2 (plain stores), which consumers reach with
// L1-bypassing loads at L2-hit latency.  Otherwise stores are write-through (sc1) so that they
// reach memory where any XCD sees them.  Placement only selects the faster legal protocol.
__device__ __forceinline__ bool cluster_same_xcd(const LstmArgs& a, int cl, int g, int tid, int& s_flag, int& s_abort) {
    if (tid < 64) {
        unsigned* hw = a.hello + (size_t)cl * HELLO_STRIDE;
        const int members = a.slab_done ? 17 : 16;     // a streamed sweep's flusher workgroup must share the XCD as well
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xFu;
        // bits 16..: HW_REG_HW_ID (cu / sh / se of this workgroup) -- diagnostic only (tools/dev/tools_stamps.py)
        const unsigned hwid = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4) & 0xFFFFu;
        if (tid == 0) __hip_atomic_store(hw + g, 0x100u | xcc | (hwid << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SpinGuard sg;
        unsigned v = 0x100u | xcc;
        while (true) {
            POLL_FENCE();
            if (tid < members) v = __hip_atomic_load(hw + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__any(v == 0u)) break;
            if (!sg.keep_waiting()) { LDS_FLAG_SET(s_abort, 1); *a.err = 1; break; }
        }
        const bool same = !__any((v & 0xFu) != xcc);
        if (tid == 0) LDS_FLAG_SET(s_flag, (same && a.force_mode == 0) ? 1 : 0);
        // tell overlapped GEMMs which XCD this cluster occupies (a speed hint, released at kernel end)
        if (tid == 0 && g == 0) __hip_atomic_fetch_add(a.busy + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return LDS_FLAG_GET(s_flag) != 0;
}


// ---- the flusher workgroup of a STREAMED backward sweep (member index G_CLUSTER + N_HELPERS) -------------------
// A backward sweep can hand its dgates rows to consumers on OTHER XCDs while it runs (the weight-gradient GEMMs of the
// same layer, gemm_c256.hip's gated TN kernel): the rows leave the storer waves as plain stores that stay dirty in this
// XCD's L2 (write-through stores cost the sweep 10 %), so somebody has to write them back.  Every storer wave
// publishes, per time slab of the weight-gradient products (pgasr_wslab_edge, common.h: n shrinking slabs, the sweep steps
// < T - h_(n-k) make up the first k), that its stores of the slab are acknowledged (stored[g] = slabs so far);
// ONE wave of this workgroup waits for the sixteen words, executes the agent-scope release (buffer_wbl2 sc1: the dirty
// lines of this XCD's L2 go to memory -- the L2 is one per XCD, so the write-back covers every member's lines) and only
// then publishes slab_done[cluster] = k with an agent-scope store.  Nothing of this sits on the sweep's chain.  Placement is
// verified as everywhere: this workgroup is the 17th start-up word, and a cluster that does NOT share one XCD with it
// runs the placement-independent protocol -- the storer waves then release their own stores before they publish, and this
// wave only aggregates.
__device__ __forceinline__ void flusher_loop(const LstmArgs& a, int cl) {
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    unsigned* hw = a.hello + (size_t)cl * HELLO_STRIDE;
    const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xFu;
    if (lane == 0) __hip_atomic_store(hw + 16, 0x100u | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SpinGuard sg;
    unsigned v = 0x100u | xcc;
    while (true) {
        POLL_FENCE();
        if (lane < 16) v = __hip_atomic_load(hw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!__any(v == 0u)) break;
        __builtin_amdgcn_s_sleep(8);
        if (!sg.keep_waiting()) { *a.err = 1; return; }
    }
    const bool same = !__any((v & 0xFu) != xcc) && a.force_mode == 0;
    const unsigned* stored = hw + 32;
    const int nslab = pgasr_wslab_count(a.T);
    sg.spins = 0;
    for (int k = 1; k <= nslab; ++k) {
        while (true) {
            POLL_FENCE();
            unsigned sv = (unsigned)k;
            if (lane < 16) sv = __hip_atomic_load(stored + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__any(sv < (unsigned)k)) break;
            // a sweep that gave up publishes nothing more: leave with it (the consumers' waits are bounded too)
            if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            if (!sg.keep_waiting()) { *a.err = 1; return; }
            __builtin_amdgcn_s_sleep(32);
        }
        sg.spins = 0;
        if (same) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // buffer_wbl2 sc1 + s_waitcnt vmcnt(0)
        if (lane == 0) __hip_atomic_store(a.slab_done + cl, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the cluster's helper workgroups (member indices G_CLUSTER .. G_CLUSTER + N_HELPERS - 1) -------------
// The compute workgroups must never touch DRAM: measured with the loader wave reading the real (HBM-cold) rows
// the forward step took 1.51 us, reading L2-resident rows 1.30 us, reading nothing 1.26 us -- and an L2 "touch"
// prefetcher running ahead on another CU did not change the first figure.  So the rows are MOVED: helper h copies
// the rows of steps h, h + N_HELPERS, .. (16 utterances x 4 KiB of gates, plus c_t and dout for the backward
// sweep) from HBM into a small ring that lives in the XCD's L2, and the compute workgroups' loader waves DMA
// from the ring.  DRAM latency and traffic stay on the helpers' CUs.
//   ready[slot] = step + 1 is published after the slot's stores are complete (vmcnt(0) in every wave, workgroup
//   barrier, then one store); a slot is rewritten only once member 0 has passed the step it held by two
//   (members are never more than one step apart), i.e. helpers run at most RING_STEPS - 2 steps ahead.
//   Placement is verified, not assumed: a helper that finds itself on another XCD than member 0 writes the ring
//   with write-through stores, and the loaders always read it with L1-bypassing loads.
__device__ __forceinline__ void helper_loop(const LstmArgs& a, int cl, int dir, int bg, int h, bool backward,
                                            int* s_flag) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int T = a.T, B = a.B;
    if (w >= 4) return;      // four waves move the rows (the barriers below count only live waves)
    bool same = false;
    {   // does the whole cluster sit on THIS XCD?  (the members' start-up words)
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xFu;
        SpinGuard sg0;
        unsigned v = 0x100u | xcc;
        while (true) {
            POLL_FENCE();
            if (lane < 16) v = __hip_atomic_load(a.hello + (size_t)cl * HELLO_STRIDE + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__any(v == 0u)) break;
            __builtin_amdgcn_s_sleep(8);
            if (!sg0.keep_waiting()) { *a.err = 1; return; }
        }
        same = !__any((v & 0xFu) != xcc) && a.force_mode == 0;
    }
    const int step_floats = backward ? BWD_STEP_FLOATS : FWD_STEP_FLOATS;
    const int RING_STEPS = backward ? BWD_RING_STEPS : FWD_RING_STEPS;
    float* ring = a.ring + (size_t)cl * RING_STEPS * step_floats;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ring, 0, (int)(RING_STEPS * step_floats * 4), 0x00020000);
    unsigned* ready = a.ready + (size_t)cl * 32;
    SpinGuard sg;
#ifdef PGASR_LSTM_DIAG
    long long hp_pace_ = 0, hp_fed_ = 0, hp_move_ = 0, hp_t_ = 0; unsigned hp_n_ = 0, hp_lead_ = 0;     // helper 0: cycles waiting for its turn / for the
                                                                                                         // feed's tiles / moving a step's rows; lead left when done
#endif
    for (int s = h; s < T; s += N_HELPERS) {
#ifdef PGASR_LSTM_DIAG
        hp_t_ = clock64();
#endif
        // back-pressure: slot s % RING_STEPS held step s - RING_STEPS
        while (true) {
            POLL_FENCE();
            const unsigned cur = __hip_atomic_load(a.progress + cl * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)s <= cur + (unsigned)(RING_STEPS - 2)) break;
            __builtin_amdgcn_s_sleep(2);
            if (!sg.keep_waiting()) { *a.err = 1; return; }
        }
        sg.spins = 0;
#ifdef PGASR_LSTM_DIAG
        { const long long n_ = clock64(); hp_pace_ += n_ - hp_t_; hp_t_ = n_; }
#endif
        const int t = backward ? (dir ? s : T - 1 - s) : (dir ? T - 1 - s : s);
        const unsigned slot_off = (unsigned)((s % RING_STEPS) * step_floats);
        u32x4 v[16];
        if (a.fed) {
            // the GEMM that produces this sweep's rows (forward: the input projection into gates; backward: the input
            // gradient of the layer above into dout) is still running on other XCDs: wait until the row tile(s) holding
            // this step's 16 rows are counted complete; the rows are then read with agent-scope loads (they were
            // written through to memory)
            const int blast = bg * 16 + 15 < B ? bg * 16 + 15 : B - 1;
            const int mt0 = (t * B + bg * 16) >> 8, mt1 = (t * B + blast) >> 8;
            while (true) {
                POLL_FENCE();
                const unsigned c0 = __hip_atomic_load(a.fed + dir * a.fed_mt + mt0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned c1 = __hip_atomic_load(a.fed + dir * a.fed_mt + mt1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c0 >= (unsigned)a.fed_need && c1 >= (unsigned)a.fed_need) break;
                __builtin_amdgcn_s_sleep(4);
                if (!sg.keep_waiting()) { *a.err = 1; return; }
            }
            sg.spins = 0;
            POLL_FENCE();
        }
#ifdef PGASR_LSTM_DIAG
        { const long long n_ = clock64(); hp_fed_ += n_ - hp_t_; hp_t_ = n_; }
#endif
        if (!backward && a.fed) {
            __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(a.gates, 0, (int)((size_t)T * B * 2 * HID * 4 * 4), 0x00020000);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                int b = bg * 16 + n; b = b < B ? b : B - 1;
                v[n] = __builtin_amdgcn_raw_buffer_load_b128(grs, (unsigned)((((((size_t)t * B + b) * 2 + dir) * HID) * 4 + w * 256 + lane * 4) * 4), 0, 16);
            }
        } else {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                int b = bg * 16 + n; b = b < B ? b : B - 1;
                v[n] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.gates + ((((size_t)t * B + b) * 2 + dir) * HID) * 4 + w * 256 + lane * 4));   // streamed once: should not displace the ring in L2
            }
        }
        u32x4 vc[4], vd[4];
        if (backward) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = j * 256 + tid;                  // 16-byte chunk of the 16 x 256 floats
                int b = bg * 16 + (i >> 6); b = b < B ? b : B - 1;
                vc[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.cbuf + (((size_t)t * B + b) * 2 + dir) * HID + (i & 63) * 4));
            }
            if (a.fed) {
                __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dout), 0, (int)((size_t)T * B * 2 * HID * 4), 0x00020000);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = j * 256 + tid;
                    int b = bg * 16 + (i >> 6); b = b < B ? b : B - 1;
                    vd[j] = __builtin_amdgcn_raw_buffer_load_b128(drs, (unsigned)((((size_t)t * B + b) * (2 * HID) + dir * HID + (i & 63) * 4) * 4), 0, 16);
                }
                if (a.drop_on) {      // nn.LSTM's inter-layer dropout (model.py:42), same mask as pgasr_dropout: one Philox call per 4 elements
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = j * 256 + tid;
                        int b = bg * 16 + (i >> 6); b = b < B ? b : B - 1;
                        const unsigned long long quad = (((unsigned long long)t * B + b) * (2 * HID) + dir * HID + (i & 63) * 4) >> 2;
                        uint32_t r[4];
                        philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), a.drop_off, 0u, a.drop_k0, a.drop_k1, r);
                        vd[j].x = r[0] >= a.drop_thresh ? __float_as_uint(__uint_as_float(vd[j].x) * a.drop_scale) : 0u;
                        vd[j].y = r[1] >= a.drop_thresh ? __float_as_uint(__uint_as_float(vd[j].y) * a.drop_scale) : 0u;
                        vd[j].z = r[2] >= a.drop_thresh ? __float_as_uint(__uint_as_float(vd[j].z) * a.drop_scale) : 0u;
                        vd[j].w = r[3] >= a.drop_thresh ? __float_as_uint(__uint_as_float(vd[j].w) * a.drop_scale) : 0u;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = j * 256 + tid;
                    int b = bg * 16 + (i >> 6); b = b < B ? b : B - 1;
                    vd[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.dout + ((size_t)t * B + b) * (2 * HID) + dir * HID + (i & 63) * 4));
                }
            }
        }
        const int aux = same ? 0 : 16;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const unsigned off = (slot_off + (unsigned)(n * 1024 + w * 256 + lane * 4)) * 4u;
            if (same) __builtin_amdgcn_raw_buffer_store_b128(v[n], rs, off, 0, 0);
            else      __builtin_amdgcn_raw_buffer_store_b128(v[n], rs, off, 0, 16);
        }
        if (backward) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned i = (unsigned)(j * 256 + tid);
                const unsigned oc = (slot_off + 16 * 1024 + i * 4) * 4u, od = (slot_off + 16 * 1024 + 16 * 256 + i * 4) * 4u;
                if (same) { __builtin_amdgcn_raw_buffer_store_b128(vc[j], rs, oc, 0, 0); __builtin_amdgcn_raw_buffer_store_b128(vd[j], rs, od, 0, 0); }
                else      { __builtin_amdgcn_raw_buffer_store_b128(vc[j], rs, oc, 0, 16); __builtin_amdgcn_raw_buffer_store_b128(vd[j], rs, od, 0, 16); }
            }
        }
        (void)aux;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores are complete
        __syncthreads();                                      // .. and so are the other three waves'
        if (tid == 0) __hip_atomic_store(ready + (s % RING_STEPS), (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef PGASR_LSTM_DIAG
        {
            const long long n_ = clock64(); hp_move_ += n_ - hp_t_; ++hp_n_;
            const unsigned cur_ = __hip_atomic_load(a.progress + cl * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hp_lead_ += (unsigned)s > cur_ ? (unsigned)s - cur_ : 0u;      // steps this slot is ahead of member 0 when it becomes ready
        }
#endif
    }
#ifdef PGASR_LSTM_DIAG
    if (h == 0 && tid == 0) {
        unsigned* o_ = a.hello + (size_t)cl * HELLO_STRIDE + 56;
        o_[0] = (unsigned)(hp_pace_ >> 4); o_[1] = (unsigned)(hp_fed_ >> 4); o_[2] = (unsigned)(hp_move_ >> 4); o_[3] = hp_n_; o_[4] = hp_lead_;
    }
#endif
    (void)s_flag;
}

// ------------------------------------------------------------------------------------------
// Workgroup anatomy of both sweeps: 6 waves.  Waves 0-3 compute (MFMA + cell math) and touch
// global memory ONLY for the exchange.  Wave 4 is the LOADER: it streams the rows the cells need
// (xproj / saved activations) a few steps ahead straight into an LDS ring with LDS-DMA
// (global_load_lds: no register result) behind a hand-counted s_waitcnt vmcnt(N).  Wave 5 is the
// STORER: it copies the cells' results (staged in LDS) to HBM and never waits on memory.
// Reason (measured, tools/dev/tools_lstm_diag.py): vmcnt retires in order and hipcc emits only vmcnt(0) in
// this kernel, so any DRAM-latency load or scattered store issued by a compute wave sits in
// front of its polling loads -- the sweep ran 1.63 ms with that traffic in the compute waves and
// 1.05 ms without it, while a cached load in the same place cost nothing.
// ------------------------------------------------------------------------------------------
#ifndef PGASR_FWD_XLAYOUT
#define PGASR_FWD_XLAYOUT 1
#endif
// The loader waves' wait for the staging ring (ring_wait) -- what round 5 measured (tools/dev/r5_instep_diag.py, r5_loader_ab.sh, r5_ring_wait.py
// with the -DPGASR_LSTM_DIAG counters below; f32 train step, one box each):
//   * with the wait REMOVED (diag bit 6: LDS-DMA from slots that may not be ready) the six sweeps of the step take 1.26-1.32 ms instead of
//     1.38-1.42 (forward) / 1.31 / 1.60 / 1.61 (backward): the step 8.9 instead of 10.0 ms.  But that is not a per-step cost: the helpers are
//     late in 2-6 of 1000 steps, and those few waits are ~400 (forward) / ~1,100-1,250 (fed backward) retries long -- the START of a fed
//     sweep, ~0.1 ms (forward) and ~0.25-0.3 ms (backward) of waiting for the first row tiles of the GEMM that feeds it (backward: that GEMM
//     gets its CUs only once the previous layer's weight-gradient workgroups have left them).  The backward phase is GEMM-bound (NOTES.md),
//     so handing the feed those CUs earlier moves work, it does not remove it: more split head tiles (PGASR_X6_SPLIT_GROUPS 16 -> 64) and
//     fewer persistent weight-gradient workgroups (PGASR_T6_GRID 256 -> 64: fed sweeps 1.61 -> 1.52 ms, tail 0.69 -> 2.9 ms) gained nothing.
//   * cheaper polls gain little or lose: the first look through the scalar memory path (s_load_dword glc: no vmcnt(0) on the loader's
//     LDS-DMAs) forward sweeps 1.42 -> 1.385, backward +0.01-0.02, step -0.05 ms; the look taken one iteration early (poll-ahead, LDS-DMAs
//     right behind the barrier) step +0.1 ms; deeper loader leads (FWD_LEAD 6-8, BWD_LEAD 7-9) backward sweeps +0.03-0.08 ms.  Not kept.
// The storer waves' result stores (gates / c / h forward, dgates backward) are NON-TEMPORAL: result lines are written once and read
// by nobody on this XCD, and as ordinary dirty lines they push the helpers' staging ring out of the L2 (whose write-back is the
// sweeps' excess HBM write traffic).  Round 4, same box, A/B by library: WRITE_SIZE per sweep launch (forward + backward averaged,
// f32) 545 -> 479 MB; sweep and step times unchanged within noise (f32 step 10.33 / 10.32 ms, bf16x3 7.84 / 7.80).  -DPGASR_NT_RESULTS=0
// builds the plain stores.
#ifndef PGASR_NT_RESULTS
#define PGASR_NT_RESULTS 1
#endif
typedef __attribute__((ext_vector_type(4))) float pgasr_f4v;
__device__ __forceinline__ void store_result4(float4* p, const float4& v) {
#if PGASR_NT_RESULTS
    __builtin_nontemporal_store((pgasr_f4v){v.x, v.y, v.z, v.w}, reinterpret_cast<pgasr_f4v*>(p));
#else
    *p = v;
#endif
}
constexpr int IO_WAVE = 4;        // first non-compute wave
constexpr int LOADER_WAVE = 4, STORER_WAVE = 5;
#ifndef PGASR_FWD_LEAD
#define PGASR_FWD_LEAD 3
#endif
constexpr int FWD_LEAD = PGASR_FWD_LEAD, FWD_RING = FWD_LEAD + 2;   // loader runs FWD_LEAD steps ahead; ring slot reuse distance > lead + 1
#ifndef PGASR_BWD_LEAD
#define PGASR_BWD_LEAD 4
#endif
constexpr int BWD_LEAD = PGASR_BWD_LEAD, BWD_RING = BWD_LEAD + 2;
static_assert(BWD_LEAD >= 3 && 6 * (BWD_LEAD - 2) <= 63 && 4 * FWD_LEAD <= 63, "counted s_waitcnt vmcnt immediates");
#ifndef PGASR_BWD_POLL_DELAY
#define PGASR_BWD_POLL_DELAY 2
#endif

// the same with the agent-scope (L1-bypassing) policy: for data another workgroup has just written
__device__ __forceinline__ void dma16_sc1(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_base, 16, 0, 16);
}
// one LDS-DMA wave-instruction: 64 lanes x 16 B, per-lane global source, LDS destination = base + lane*16
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_base, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------
// forward sweep.  1-D grid of (16 + N_HELPERS)*NCL8 workgroups (16 compute members + the helpers of each cluster):
// cluster cl = b % NCL8 (members share b % 8, i.e. an XCD under the observed round-robin
// placement), member g = b / NCL8 owns hidden units 16g..16g+15.  Wave w < 4 multiplies the
// k-quarter [64w, 64w+64) of h_{t-1} into all 64 gate rows of the workgroup (4 MFMA tiles x 2
// k-steps x 3 split terms); the four partial tiles are summed through LDS and each compute thread
// finishes ONE (unit, utterance) cell.
// exchange slot per cluster: [parity 2][kc 32][plane NP][n 16][8 bf16]; member g owns kc = 2g, 2g+1.  (Rounds 1-3 had the plane
// inside the utterance, [kc][n][plane][8]: a poll instruction -- 4 kc x 16 n x 16 B of ONE plane -- then touched every line of
// 4 x 16 x NP x 16 B and used 1/NP of each, 64 line requests per wave and step with two planes, 144 with three; the exchange is
// request-bound, and stamps put the three-plane poll at 1259 cycles against 760.  Plane-major, an instruction reads 8
// full lines: 32 / 48 requests.  PGASR_FWD_XLAYOUT=0 builds the old layout for A/B.)
// Measured and NOT kept (round 3, commit "second forward-sweep structure" in the history): the polled h_{t-1} itself through
// LDS instead of the partial tiles -- every wave copies its validated k-quarter into an LDS image, one barrier, every wave
// reads the whole 16 KB and multiplies all 256 k into its OWN 16 gate rows, so that the 16 x 16 MFMA output hands each lane
// the four gates of one cell and the 4-way partial sum (16 KB of ds_write_b128, 4 reads + adds per cell) disappears.  Correct
// (all BLSTM tests, both precisions), but 1.22 ms per sweep against 1.05, with one accumulator chain and with four: the
// K-split's MFMAs run BEFORE the barrier, i.e. under the spread of the four waves' poll completions; behind the barrier all
// 24 of them (and 16 fragment reads) are serial time on the chain.
// ------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_fwd_kernel(LstmArgs a) {
    const int cl = blockIdx.x % a.NCL8, g = blockIdx.x / a.NCL8;
    if (cl >= 2 * a.NBG) return;
    const int dir = cl & 1, bg = cl >> 1;
    if (g >= G_CLUSTER) { helper_loop(a, cl, dir, bg, g - G_CLUSTER, false, nullptr); return; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4, n = lane & 15;      // MFMA coordinates
    const int pu = tid & 15, pn = (tid >> 4) & 15;   // cell coordinates of a compute thread (unit fastest)
    const int T = a.T, B = a.B;

    constexpr int PROW = 16 * 4 + 4;             // padded row of 16 units x 4 gates (conflict-free both ways)
    __shared__ __attribute__((aligned(16))) float part[2][4 * 16 * PROW];       // [step parity][w][n][u][gate]
    __shared__ __attribute__((aligned(16))) float4 xin[FWD_RING][256];          // xproj ring (filled by LDS-DMA), by cell id
    __shared__ __attribute__((aligned(16))) float4 rg[2][256];                  // results: gate activations
    __shared__ __attribute__((aligned(16))) float rc[2][256];                   //          c_t
    __shared__ __attribute__((aligned(16))) float rh[2][256];                   //          h_t (0 past the length)
    __shared__ int s_abort;
    __shared__ int s_same;
    if (tid == 0) { LDS_FLAG_SET(s_abort, 0); LDS_FLAG_SET(s_same, 0); }
    __syncthreads();

    // weight slices -> registers: tile m (units 4(4g+m)..+3, row = 4*uu+gate), k-steps 2w, 2w+1, NP planes each
    bf16x8 W[4][2][NP];
    if (w < IO_WAVE) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const u32x4* wp = a.wpack + ((size_t)(dir * 64 + 4 * g + m) * 8) * NP * 64;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) W[m][i][pl] = __builtin_bit_cast(bf16x8, wp[((2 * w + i) * NP + pl) * 64 + lane]);
            }
        }
    }
    constexpr unsigned SLOT = 32 * 16 * NP * 16;      // bytes per parity slot (16 KiB; 24 KiB with three planes)
    unsigned char* xb = a.xbuf + (size_t)cl * (2 * (32 * 16 * 3 * 16));      // cluster blocks are laid out for three planes (lstm_ws_layout)
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)(2 * SLOT), 0x00020000);
    const bool same_xcd = cluster_same_xcd(a, cl, g, tid, s_same, s_abort);

    const int bidx = bg * 16 + pn;
    const int len = (bidx < B) ? a.lengths[bidx] : 0;
    float c = 0.f, h = 0.f;
    STAMP_DECL;

    // ---- loader / storer: lane L serves cells 64k+L (k = 0..3) for 16-byte rows, cells 4L..4L+3 for 4-byte rows
    auto step_t = [&](int s) { return dir ? T - 1 - s : s; };
    auto gate_addr = [&](int t, int cell) {
        const int b = bg * 16 + (cell >> 4);
        return reinterpret_cast<float4*>(a.gates + ((((size_t)t * B + b) * 2 + dir) * HID + 16 * g + (cell & 15)) * 4);
    };
    // 4 LDS-DMA instructions per step, ALWAYS issued (step and utterance clamped into range) so that the
    // counted wait below is exact: vmcnt(4*FWD_LEAD) leaves only the loads of the next FWD_LEAD steps in flight
    constexpr int RING_STEPS = FWD_RING_STEPS;
    const float* ring = a.ring + (size_t)cl * RING_STEPS * FWD_STEP_FLOATS;
    const unsigned* ready = a.ready + (size_t)cl * 32;
    RW_DECL;
    auto ring_wait = [&](int s) {      // until the helpers have staged step s (wave-uniform; only the loader wave waits here)
        SpinGuard sgr;
        RW_BEGIN();
        while (true) {
            POLL_FENCE();
            if (__hip_atomic_load(ready + (s % RING_STEPS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(s + 1)) break;
            RW_RETRY();
            if (LDS_FLAG_GET(s_abort) || !sgr.keep_waiting()) { LDS_FLAG_SET(s_abort, 1); *a.err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        RW_END();
    };
    auto loader_issue = [&](int s) {
        if (a.diag & 2) return;
        const int sc = s < T ? s : T - 1;
        if (a.n_helpers == 0) {      // no room for helpers (two clusters per XCD): straight from HBM, the slower way
            const int t = step_t(sc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cell = 64 * k + lane;
                int b = bg * 16 + (cell >> 4); b = b < B ? b : B - 1;
                dma16(a.gates + ((((size_t)t * B + b) * 2 + dir) * HID + 16 * g + (cell & 15)) * 4, &xin[s % FWD_RING][64 * k]);
            }
            return;
        }
        if (!(a.diag & 64)) ring_wait(sc);
        if (a.diag & 32) return;
        const float* slot = ring + (size_t)(sc % RING_STEPS) * FWD_STEP_FLOATS;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cell = 64 * k + lane;
            dma16_sc1(slot + (cell >> 4) * 1024 + (16 * g + (cell & 15)) * 4, &xin[s % FWD_RING][64 * k]);
        }
    };
    auto io_store_results = [&](int s) {      // results of step s: 4 + 1 + 1 sixteen-byte stores per lane
        if (a.diag & 1) return;
        const int t = step_t(s), par = s & 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cell = 64 * k + lane;
            if (bg * 16 + (cell >> 4) < B) store_result4(gate_addr(t, cell), rg[par][cell]);
        }
        const int cell0 = 4 * lane, b = bg * 16 + (cell0 >> 4), u0 = 16 * g + (cell0 & 15);
        if (b < B) {
            store_result4(reinterpret_cast<float4*>(a.cbuf + (((size_t)t * B + b) * 2 + dir) * HID + u0), *reinterpret_cast<const float4*>(&rc[par][cell0]));
            const float4 hv = *reinterpret_cast<const float4*>(&rh[par][cell0]);
            const size_t oi = ((size_t)t * B + b) * (2 * HID) + dir * HID + u0;
            store_result4(reinterpret_cast<float4*>(a.out + oi), hv);
            if (a.out_drop) {
                // nn.LSTM's inter-layer dropout (model.py:42) on the way out: the mask pgasr_dropout would give this tensor
                // (one Philox call per 4 consecutive elements), so the separate 2 x 65 MB pass and its launch disappear
                uint32_t r[4];
                philox4x32_10((uint32_t)(oi >> 2), (uint32_t)((oi >> 2) >> 32), a.drop_off, 0u, a.drop_k0, a.drop_k1, r);
                float4 dv;
                dv.x = r[0] >= a.drop_thresh ? hv.x * a.drop_scale : 0.f;
                dv.y = r[1] >= a.drop_thresh ? hv.y * a.drop_scale : 0.f;
                dv.z = r[2] >= a.drop_thresh ? hv.z * a.drop_scale : 0.f;
                dv.w = r[3] >= a.drop_thresh ? hv.w * a.drop_scale : 0.f;
                store_result4(reinterpret_cast<float4*>(a.out_drop + oi), dv);
            }
        }
    };
    // Three role-specialised loops with ONE LDS barrier per step each (the counts must match).  One shared loop with
    // `if (w == ...)` regions cost the compute waves a dozen exec-mask branches per step: a taken branch refills the
    // instruction buffer, and at ~2700 cycles per step those bubbles were measurable.  Abort protocol: every wave reads
    // the flag right behind the barrier (a wave can only set it after seconds of spinning, so all see the same value)
    // and leaves at the end of that step.
    if (w == LOADER_WAVE) {
        for (int s = 0; s < FWD_LEAD; ++s) loader_issue(s);
        for (int step = 0; step < T; ++step) {
            loader_issue(step + FWD_LEAD);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * FWD_LEAD) : "memory");   // this step's rows have landed
            if (g == 0 && lane == 0)    // paces the helpers (which may sit on another XCD: agent-scope store)
                __hip_atomic_store(a.progress + cl * 32, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            LDS_BARRIER();
            if (LDS_FLAG_GET(s_abort)) break;
        }
    } else if (w == STORER_WAVE) {
        for (int step = 0; step < T; ++step) {
            LDS_BARRIER();        // the previous step's results are visible
            const int aborted = LDS_FLAG_GET(s_abort);
            if (step > 0) io_store_results(step - 1);
            if (aborted) break;
        }
    } else {
#ifdef PGASR_SWEEP_PRIO
        // A/B only: the compute waves of SIMD 0 / 1 share their instruction arbiter with the loader / storer wave.  Round 5 (tools/dev/r5_sweep_prio.sh,
        // f32 step, A/B/C/A/B by library): priority 3 forward sweeps 4.17-4.21 ms against 4.20-4.23, backward 4.42-4.45 against 4.38-4.41; priority 1
        // 4.18 / 4.44 -- nothing either way, not built into the product
        __builtin_amdgcn_s_setprio(PGASR_SWEEP_PRIO);
#endif
        // one step of a compute wave; `first` is a literal at both call sites, so the step-0 special cases fold away
        auto compute_step = [&](const int step, const bool first) -> int {
            const int t = step_t(step);
            STAMP(0);
            if (!first) {
                const unsigned pbase = (unsigned)((step - 1) & 1) * SLOT;
                // fresh word of epoch e: (bit0, bit16) = (e, 1-e); the two halves of a word are written by
                // different lanes (2-byte stores), so BOTH bits are checked
                const unsigned want = (((step - 1) >> 1) & 1) ? 0x00000001u : 0x00010000u;
                u32x4 vp[2][NP];
                auto issue_loads = [&]() {
                    POLL_FENCE();
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
#if PGASR_FWD_XLAYOUT
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
                            vp[i][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pbase + (unsigned)(((((4 * (2 * w + i) + q) * NP + pl) * 16 + n)) * 16), 0, 16);
#else
                        const unsigned off = pbase + (unsigned)((((4 * (2 * w + i) + q) * 16 + n) * NP) * 16);
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) vp[i][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16 * pl, 0, 16);
#endif
                    }
                };
#ifdef PGASR_LSTM_DIAG
                if (a.diag & 8) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) vp[i][pl] = (u32x4){0u, 0u, 0u, 0u};
                } else
#endif
                issue_loads();            // the operand loads ARE the poll
                SpinGuard sg;
                while (true) {
                    unsigned bad = (bad4(vp[0][0], want) | bad4(vp[0][1], want)) | (bad4(vp[1][0], want) | bad4(vp[1][1], want));
                    if constexpr (NP == 3) bad |= bad4(vp[0][2], want) | bad4(vp[1][2], want);
                    if (!__any(bad != 0)) break;
#ifdef PGASR_LSTM_DIAG
                    if (a.diag & 12) break;
#endif
                    if (LDS_FLAG_GET(s_abort) || !sg.keep_waiting()) { LDS_FLAG_SET(s_abort, 1); *a.err = 1; break; }
                    issue_loads();
                }
                STAMP(1);
                bf16x8 Hp[2][NP];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) Hp[i][pl] = __builtin_bit_cast(bf16x8, vp[i][pl]);
                }
                // Two tile pairs, each pair's 12 MFMAs interleaved, and the first pair's partial tiles written to LDS while the
                // second pair is multiplied: ds_write_b128 moves only ~80 B/clk per CU, i.e. the 16 KiB of partials are ~200
                // cycles of LDS store time -- half of it sits under the MFMAs instead of behind them.
                // tile m, lane (q,n): gates of local unit 4m+q for utterance n
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    f32x4 acc[2];
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) acc[m2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 2; ++i) mfma_planes_pair<NP>(W[2 * half][i], W[2 * half + 1][i], Hp[i], acc);
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2)
                        *reinterpret_cast<float4*>(&part[step & 1][(w * 16 + n) * PROW + (4 * (2 * half + m2) + q) * 4]) =
                            make_float4(acc[m2][0], acc[m2][1], acc[m2][2], acc[m2][3]);
                    __builtin_amdgcn_sched_barrier(0);      // keep the first pair's stores in front of the second pair's MFMAs
                }
                STAMP(2);
            }
            LDS_BARRIER();        // partial tiles + this step's xproj visible; previous step's results visible to the storer wave
            STAMP(3);
            const int aborted = LDS_FLAG_GET(s_abort);
            float4 pre = xin[step % FWD_RING][tid];
            if (!first) {
                const float* pp = &part[step & 1][pn * PROW + pu * 4];
                const float4 p0 = *reinterpret_cast<const float4*>(pp);
                const float4 p1 = *reinterpret_cast<const float4*>(pp + 16 * PROW);
                const float4 p2 = *reinterpret_cast<const float4*>(pp + 32 * PROW);
                const float4 p3 = *reinterpret_cast<const float4*>(pp + 48 * PROW);
                pre.x += (p0.x + p1.x) + (p2.x + p3.x);
                pre.y += (p0.y + p1.y) + (p2.y + p3.y);
                pre.z += (p0.z + p1.z) + (p2.z + p3.z);
                pre.w += (p0.w + p1.w) + (p2.w + p3.w);
            }
            STAMP(4);
            const float gi = sigmoidf_fast(pre.x);
            const float gf = sigmoidf_fast(pre.y);
            const float gg = tanhf_fast(pre.z);
            const float go = sigmoidf_fast(pre.w);
            const bool active = t < len;
            const float cn = gf * c + gi * gg;
            const float hn = go * tanhf_fast(cn);
            if (active) { c = cn; h = hn; }
#ifdef PGASR_LSTM_DIAG
            if (!(a.diag & 16))
#endif
            {
                // publish h_t (also after the last step: nobody reads that slot, and lstm_prepare_kernel resets it): each cell
                // thread writes its own bf16 planes (2-byte stores); layout [kc = unit/8][plane][n][unit%8] of this member's block
                const unsigned e = (unsigned)(step >> 1) & 1u;
                const unsigned tb = (pu & 1) ? (1u - e) : e;
                unsigned short hp[NP];
                split_planes_tagged<NP>(h, tb, hp);
#if PGASR_FWD_XLAYOUT
                const unsigned off = (unsigned)(step & 1) * SLOT + (unsigned)g * (512u * NP) +
                                     (unsigned)(((pu >> 3) * NP * 16 + pn) * 16 + (pu & 7) * 2);
                constexpr unsigned PSTRIDE = 256;      // plane stride inside a kc block: 16 n x 16 B
#else
                const unsigned off = (unsigned)(step & 1) * SLOT + (unsigned)g * (512u * NP) +
                                     (unsigned)((((pu >> 3) * 16 + pn) * NP) * 16 + (pu & 7) * 2);
                constexpr unsigned PSTRIDE = 16;
#endif
                if (same_xcd) {
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) __builtin_amdgcn_raw_buffer_store_b16(hp[pl], rsrc, off + PSTRIDE * pl, 0, 0);
                } else {
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) __builtin_amdgcn_raw_buffer_store_b16(hp[pl], rsrc, off + PSTRIDE * pl, 0, 16);
                }
            }
            STAMP(5);
            // results for the backward pass / next layer: staged in LDS, written out by the storer wave
            rg[step & 1][tid] = active ? make_float4(gi, gf, gg, go) : make_float4(0, 0, 0, 0);
            rc[step & 1][tid] = c;
            rh[step & 1][tid] = active ? hn : 0.f;
            STAMP(6);
            return aborted;
        };
        if (T > 0 && !compute_step(0, true)) {
            for (int step = 1; step < T; ++step)
                if (compute_step(step, false)) break;
        }
    }
    STAMP_FLUSH();
    __syncthreads();
    if (w == STORER_WAVE && !LDS_FLAG_GET(s_abort) && T > 0) io_store_results(T - 1);
    RW_FLUSH();
    release_xcd(a, g, tid);
}

// ------------------------------------------------------------------------------------------
// backward sweep.  Same grid, ownership and wave roles.  The recurrent gradient
// dh_{prev}[k] = sum_r dgates[r] W_hh[r,k] is formed as a sum of 16 PARTIAL products, one per
// member, each over the member's own 64 gate rows (K = 64): every step a workgroup
//   1. loads the 16 partial sums addressed to its units (16 fp32 words per compute thread, tagged),
//   2. finishes one (unit, utterance) cell: dgates from the saved activations (from the I/O wave),
//   3. multiplies its dgates (bf16 hi/lo through LDS) into all 256 outputs
//      (16 MFMA tiles over the 4 compute waves, 2 k-steps x 3 split terms) and
//   4. publishes that partial: [src g][n 16][unit 256] fp32 = 16 KiB, 4 x 16-B stores per lane.
// Exchange read per workgroup and step: 16 KiB (a dgates all-gather would be 64 KiB).
// ------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_bwd_kernel(LstmArgs a) {
    const int cl = blockIdx.x % a.NCL8, g = blockIdx.x / a.NCL8;
    if (cl >= 2 * a.NBG) return;
    const int dir = cl & 1, bg = cl >> 1;
    if (g >= G_CLUSTER + a.n_helpers) { flusher_loop(a, cl); return; }      // launched only with slab_done
    if (g >= G_CLUSTER) { helper_loop(a, cl, dir, bg, g - G_CLUSTER, true, nullptr); return; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4, n = lane & 15;      // MFMA coordinates
    const int pu = tid & 15, pn = (tid >> 4) & 15;   // cell coordinates (unit fastest: coalesced partial reads)
    const int unit = 16 * g + pu;
    const int T = a.T, B = a.B;

    __shared__ __attribute__((aligned(16))) unsigned short dgl[2 * 16 * NP * 64];  // [buf][n][plane][64 r' local]
    __shared__ __attribute__((aligned(16))) float4 sg_[BWD_RING][256];             // saved gate activations ring, by cell id
    __shared__ __attribute__((aligned(16))) float sct[BWD_RING][256];              // c_t ring (c_prev of step s = c_t of step s+1)
    __shared__ __attribute__((aligned(16))) float sdy[BWD_RING][256];              // dout ring
    __shared__ __attribute__((aligned(16))) float4 rdg[2][256];                    // results: d(pre-activation gates)
    __shared__ int s_abort;
    __shared__ int s_same;
    if (tid == 0) { LDS_FLAG_SET(s_abort, 0); LDS_FLAG_SET(s_same, 0); }
    __syncthreads();

    // A operand tiles: output units 16*(4w+mt)..+15 (rows), k = own gate rows r' = 64g + 32i + ..
    bf16x8 W[4][2][NP];
    if (w < IO_WAVE) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const u32x4* wp = a.wpack + ((size_t)(dir * 16 + 4 * w + mt) * 32) * NP * 64;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) W[mt][i][pl] = __builtin_bit_cast(bf16x8, wp[((2 * g + i) * NP + pl) * 64 + lane]);
            }
        }
    }
    constexpr unsigned SLOT = 16 * 16 * 256 * 4;      // [src 16][dst 16][n 16][unit 16] fp32 = 256 KiB
    unsigned char* xb = a.xbuf + (size_t)cl * (2 * SLOT);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)(2 * SLOT), 0x00020000);
    const bool same_xcd = cluster_same_xcd(a, cl, g, tid, s_same, s_abort);
    // publish descriptors: the plain form lands in this XCD's L2 (cluster verified to share it), the write-through form is for
    // a cluster spread over XCDs; the one that does not apply has an empty range, so stores through it are dropped
    __amdgpu_buffer_rsrc_t rs_same = __builtin_amdgcn_make_buffer_rsrc(xb, 0, same_xcd ? (int)(2 * SLOT) : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_far = __builtin_amdgcn_make_buffer_rsrc(xb, 0, same_xcd ? 0 : (int)(2 * SLOT), 0x00020000);

    const int bidx = bg * 16 + pn;
    const int len = (bidx < B) ? a.lengths[bidx] : 0;
    float dc = 0.f, carry = 0.f;
    float4 dbs = make_float4(0, 0, 0, 0);      // running sum over the sweep of this cell's dgates (bias gradient)

    // ---- loader / storer: lane L serves cells 64k+L for the 16-byte gate rows and cells 4L..4L+3 for 4-byte rows
    auto step_t = [&](int s) { return dir ? s : T - 1 - s; };
    // 6 LDS-DMA instructions per step, ALWAYS issued (step and utterance clamped) so the counted wait is exact
    constexpr int RING_STEPS = BWD_RING_STEPS;
    const float* ring = a.ring + (size_t)cl * RING_STEPS * BWD_STEP_FLOATS;
    const unsigned* ready = a.ready + (size_t)cl * 32;
    RW_DECL;
    auto ring_wait = [&](int s) {      // until the helpers have staged step s (wave-uniform; only the loader wave waits here)
        SpinGuard sgr;
        RW_BEGIN();
        while (true) {
            POLL_FENCE();
            if (__hip_atomic_load(ready + (s % RING_STEPS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(s + 1)) break;
            RW_RETRY();
            if (LDS_FLAG_GET(s_abort) || !sgr.keep_waiting()) { LDS_FLAG_SET(s_abort, 1); *a.err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        RW_END();
    };
    auto loader_issue = [&](int s) {
        if (a.diag & 2) return;
        const int sc = s < T ? s : T - 1;
        const int slot = s % BWD_RING;
        if (a.n_helpers == 0) {      // no room for helpers (two clusters per XCD): straight from HBM, the slower way
            const int t = step_t(sc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cell = 64 * k + lane;
                int b = bg * 16 + (cell >> 4); b = b < B ? b : B - 1;
                dma16(a.gates + ((((size_t)t * B + b) * 2 + dir) * HID + 16 * g + (cell & 15)) * 4, &sg_[slot][64 * k]);
            }
            const int cell0 = 4 * lane, u0 = 16 * g + (cell0 & 15);
            int b = bg * 16 + (cell0 >> 4); b = b < B ? b : B - 1;
            dma16(a.cbuf + (((size_t)t * B + b) * 2 + dir) * HID + u0, &sct[slot][0]);
            dma16(a.dout + ((size_t)t * B + b) * (2 * HID) + dir * HID + u0, &sdy[slot][0]);
            return;
        }
        if (!(a.diag & 64)) ring_wait(sc);
        if (a.diag & 32) return;
        const float* rs_ = ring + (size_t)(sc % RING_STEPS) * BWD_STEP_FLOATS;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cell = 64 * k + lane;
            dma16_sc1(rs_ + (cell >> 4) * 1024 + (16 * g + (cell & 15)) * 4, &sg_[slot][64 * k]);
        }
        const int cell0 = 4 * lane, u0 = 16 * g + (cell0 & 15);
        dma16_sc1(rs_ + 16 * 1024 + (cell0 >> 4) * 256 + u0, &sct[slot][0]);
        dma16_sc1(rs_ + 16 * 1024 + 16 * 256 + (cell0 >> 4) * 256 + u0, &sdy[slot][0]);
    };
    auto io_store_results = [&](int s) {
        if (a.diag & 1) return;
        const int t = step_t(s), par = s & 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cell = 64 * k + lane, b = bg * 16 + (cell >> 4);
            if (b < B) store_result4(reinterpret_cast<float4*>(a.gates + ((((size_t)t * B + b) * 2 + dir) * HID + 16 * g + (cell & 15)) * 4), rdg[par][cell]);
        }
    };
    if (w == LOADER_WAVE) {
        for (int s = 0; s < BWD_LEAD; ++s) loader_issue(s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // rows of steps 0..BWD_LEAD-1 are in LDS
    }
    LDS_BARRIER();

    // Role-specialised loops, one LDS barrier per step each (see lstm_fwd_kernel).
    if (w == LOADER_WAVE) {
        for (int step = 0; step < T; ++step) {
            if (g == 0 && lane == 0)     // paces the helpers
                __hip_atomic_store(a.progress + cl * 32, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // cells of step+1 read rows(step+1) and c_t(step+2) before the NEXT barrier: keep only the two
            // youngest steps (12 instructions) in flight
            loader_issue(step + BWD_LEAD);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 * (BWD_LEAD - 2)) : "memory");
            LDS_BARRIER();
            if (LDS_FLAG_GET(s_abort)) break;
        }
    } else if (w == STORER_WAVE) {
        // streamed sweep: stored[g] = k once the dgates stores of sweep steps < T - h_(n-k) are acknowledged (see
        // flusher_loop).  The wait sits in FRONT of a step's stores, a whole step after the youngest store it covers was
        // issued, so it normally finds nothing outstanding.
        unsigned* stored = a.hello + (size_t)cl * HELLO_STRIDE + 32 + g;
        const int nslab = a.slab_done ? pgasr_wslab_count(T) : 0;
        int next_pub = nslab > 1 ? T - pgasr_wslab_edge(T, nslab - 1) : T + 1;
        unsigned k_pub = 1;
        auto publish = [&]() {
            if (same_xcd) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // members on several XCDs: each releases its own stores
            if (lane == 0) __hip_atomic_store(stored, k_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++k_pub;
        };
        int aborted = 0;
        for (int step = 0; step < T; ++step) {
            LDS_BARRIER();
            aborted = LDS_FLAG_GET(s_abort);
            if (step == next_pub) { publish(); next_pub = (int)k_pub < nslab ? T - pgasr_wslab_edge(T, nslab - (int)k_pub) : T + 1; }
            io_store_results(step);
            if (aborted) break;
        }
        if (nslab && !aborted) publish();       // the last slab (k_pub == number of slabs here)
    } else {
#ifdef PGASR_SWEEP_PRIO
        __builtin_amdgcn_s_setprio(PGASR_SWEEP_PRIO);     // A/B only (see lstm_fwd_kernel)
#endif
        STAMP_DECL;        // diagnostic build: 0 loop top + pre-poll cell work, 1 poll, 2 cell gradient + plane stores, 3 LDS barrier, 4 MFMA + publish
        auto compute_step = [&](const int step, const bool first) -> int {
            const int t = step_t(step);
            float4 d = make_float4(0, 0, 0, 0);
            unsigned short* dbuf = &dgl[(step & 1) * (16 * NP * 64)];
            // everything that does not need the incoming dh is done BEFORE the poll (the rows of this step have
            // been in the LDS ring since the previous barrier): after the hand-off only five multiply-adds remain
            const bool active = t < len;
            const float4 gt = sg_[step % BWD_RING][tid];
            const float ct = sct[step % BWD_RING][tid], dy = sdy[step % BWD_RING][tid];
            const float cp = (step + 1 < T) ? sct[(step + 1) % BWD_RING][tid] : 0.f;
            const float gi = gt.x, gf = gt.y, gg = gt.z, go = gt.w;
            const float tc = tanhf_fast(ct);
            const float k_c = go * (1.f - tc * tc);          // d(dh) -> d(c)
            const float k_i = gg * gi * (1.f - gi), k_f = cp * gf * (1.f - gf), k_g = gi * (1.f - gg * gg), k_o = tc * go * (1.f - go);
            float dh_rec = carry;
            STAMP(0);
            if (!first) {
                // slot layout [src 16][dst 16][n 16][16 units]: what one member reads from one source is ONE contiguous KiB
                // (a wave's load = 256 contiguous bytes), and what a wave publishes for one destination as well
                const unsigned pbase = (unsigned)((step - 1) & 1) * SLOT + (unsigned)((g * 256 + pn * 16 + pu) * 4);
                const unsigned stale_bit = (((step - 1) >> 1) & 1) ? 0x2u : 0x1u;
                float v[16];
                SpinGuard sg;
                // The partial sums cannot be there earlier than one hand-off after this member's own publish, and a
                // poll is 16 KiB per workgroup through the L2 that the other members' 16-KiB publishes are still
                // entering: hold the first poll back (x64 cycles).  Measured stand-alone 0: 1.61, 6: 1.59, 8: 1.55,
                // 10: 1.52-1.57, 14: 1.61 us per step; train step 11.82 -> 11.64 ms with 10.  Re-measured after the dh-independent
                // cell work moved in front of the poll: 6: 1.42, 8: 1.41, 10: 1.45, 12: 1.50, 14: 1.55; and again with the
                // [src][dst][n][unit] slot layout (publishes land sooner): 0-4: 1.26-1.30, 6: 1.32, 8: 1.37, 10: 1.43 -> 2.  Round 4, both
                // plane counts (ms per sweep, NP = 3 / NP = 2): 0: 1.29 / 1.155, 2: 1.26 / 1.13, 4: 1.28 / 1.16, 6: 1.34 / 1.21.  (The forward sweep
                // publishes 1 KiB per member and only loses from a delay: 1.15 -> 1.19 / 1.30 / 1.37 for 4 / 8 / 12.)
                __builtin_amdgcn_s_sleep(PGASR_BWD_POLL_DELAY);
                while (true) {
                    POLL_FENCE();
                    unsigned orr = 0;
#ifdef PGASR_LSTM_DIAG
                    if (a.diag & 8) {
#pragma unroll
                        for (int s = 0; s < 16; ++s) v[s] = 0.f;
                    } else
#endif
                    {
#pragma unroll
                        for (int s = 0; s < 16; ++s) {
                            const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(rsrc, pbase + (unsigned)s * (16 * 256 * 4), 0, 16);
                            orr |= u;
                            v[s] = __uint_as_float(u);
                        }
                    }
                    if (!__any((orr & stale_bit) != 0)) break;
#ifdef PGASR_LSTM_DIAG
                    if (a.diag & 12) break;
#endif
                    if (LDS_FLAG_GET(s_abort) || !sg.keep_waiting()) { LDS_FLAG_SET(s_abort, 1); *a.err = 1; break; }
                }
                float sum = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) sum += v[s];     // fixed order: reproducible
                dh_rec += sum;
            }
            STAMP(1);
            const float dh = dy + dh_rec;
            const float dct = dh * k_c + dc;
            d.x = dct * k_i;
            d.y = dct * k_f;
            d.z = dct * k_g;
            d.w = dh * k_o;
            if (active) { dc = dct * gf; carry = 0.f; }
            else { d = make_float4(0, 0, 0, 0); carry = dh_rec; }
            dbs.x += d.x; dbs.y += d.y; dbs.z += d.z; dbs.w += d.w;
            {
                unsigned short px[NP], py[NP], pz[NP], pw[NP];
                split_planes_plain<NP>(d.x, px); split_planes_plain<NP>(d.y, py);
                split_planes_plain<NP>(d.z, pz); split_planes_plain<NP>(d.w, pw);
                unsigned short* dst = &dbuf[(pn * NP) * 64 + pu * 4];      // [n][plane][r' local = 4*pu + gate]
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    *reinterpret_cast<uint2*>(dst + 64 * pl) = make_uint2(px[pl] | ((unsigned)py[pl] << 16), pz[pl] | ((unsigned)pw[pl] << 16));
            }
            rdg[step & 1][tid] = d;      // written to HBM by the storer wave after the barrier
            STAMP(2);
            LDS_BARRIER();
            STAMP(3);
            const int aborted = LDS_FLAG_GET(s_abort);       // read behind the barrier, used at the end of the step
            {
                // (also after the last step: nobody reads that slot, and lstm_prepare_kernel resets it)
                bf16x8 Dp[2][NP];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        Dp[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&dbuf[(n * NP + pl) * 64 + 32 * i + 8 * q]));
                }
                const unsigned e = (unsigned)(step >> 1) & 1u;
                const unsigned tag = e ? 0x1u : 0x2u;       // (bit0, bit1) = (e, 1-e)
                const unsigned obase = (unsigned)(step & 1) * SLOT + (unsigned)g * (16 * 256 * 4);
#ifdef PGASR_LSTM_DIAG
                const bool no_publish = (a.diag & 16) != 0;
#endif
                // two tile pairs: the first pair is on its way to L2 while the second is still being multiplied (the
                // 16 KiB this workgroup publishes per step is the start of everybody else's hand-off)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    f32x4 acc[2];
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) acc[m2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 2; ++i) mfma_planes_pair<NP>(W[2 * half][i], W[2 * half + 1][i], Dp[i], acc);
                    if (half == 0) STAMP(5); else STAMP(7);
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) {
                        const int mt = 2 * half + m2;
                        u32x4 o;
                        o.x = (__float_as_uint(acc[m2][0]) & ~3u) | tag;
                        o.y = (__float_as_uint(acc[m2][1]) & ~3u) | tag;
                        o.z = (__float_as_uint(acc[m2][2]) & ~3u) | tag;
                        o.w = (__float_as_uint(acc[m2][3]) & ~3u) | tag;
                        // lane (q,n), tile 4w+mt: output units 16(4w+mt) + 4q .. +3 for utterance n
                        const unsigned off = obase + (unsigned)(((4 * w + mt) * 256 + n * 16 + 4 * q) * 4);
#ifdef PGASR_LSTM_DIAG
                        if (no_publish) continue;
#endif
                        if constexpr (PGASR_PUBLISH_BRANCHFREE && NP == 3) {
                            // both forms are issued, one of them against a descriptor with an EMPTY range (the hardware drops it)
                            // instead of a wave-uniform if / else around each store (two taken branches per store, eight per step).
                            // Measured (tools/dev/r4_branchfree.sh, library A/B, two rounds): three-plane backward sweep 1.279 ->
                            // 1.259 ms alone, 4.74 -> 4.60 ms for the three of a step; the two-plane backward sweep LOSES (1.131 ->
                            // 1.26 ms) and so do both forward sweeps (1.00 -> 1.10 / 1.33 -> 1.39 ms: 2 x NP two-byte stores per thread,
                            // the dropped ones are not free) -- they keep the branches.
                            __builtin_amdgcn_raw_buffer_store_b128(o, rs_same, off, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(o, rs_far, off, 0, 16);
                        } else {
                            if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, off, 0, 0);
                            else          __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, off, 0, 16);
                        }
                    }
                    if (half == 0) STAMP(6);
                }
            }
            STAMP(4);
            return aborted;
        };
        if (T > 0 && !compute_step(0, true)) {
            for (int step = 1; step < T; ++step)
                if (compute_step(step, false)) break;
        }
        STAMP_FLUSH();
    }
    if (a.dbias_part) {
        // bias gradient for free: sum the per-cell running sums over the group's 16 utterances in a fixed order
        __syncthreads();                 // the storer wave has read rdg for the last step
        if (w < IO_WAVE) rdg[0][tid] = dbs;
        __syncthreads();
        if (tid < 16) {
            float4 sum = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float4 v = rdg[0][u * 16 + tid];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            *reinterpret_cast<float4*>(a.dbias_part + (((size_t)bg * 2 + dir) * HID + 16 * g + tid) * 4) = sum;
        }
    }
    RW_FLUSH();
    release_xcd(a, g, tid);
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
    unsigned short* wpf;    // forward A-operand pack  [2][64 tiles][8 ks][planes][64 lanes][8]
    unsigned short* wpb;    // backward A-operand pack [2][16 tiles][32 ks][planes][64 lanes][8]
    int planes;             // 2 (hi, lo) or 3 (hi, mid, lo)
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
            float v = p.w_hh[dir][(size_t)(gate * HID + 4 * tau + uu) * HID + k];
            const size_t base = ((((size_t)(dir * 64 + tau) * 8 + ks) * p.planes) * 64 + lane) * 8 + j;
            for (int pl = 0; pl < p.planes; ++pl) {
                const unsigned short b = f2bf(v);
                p.wpf[base + (size_t)pl * 64 * 8] = b;
                v -= bf2f(b);
            }
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
            float v = p.w_hh[dir][(size_t)(gate * HID + unit) * HID + ko];
            const size_t base = ((((size_t)(dir * 16 + mu) * 32 + ks32) * p.planes) * 64 + lane) * 8 + j;
            for (int pl = 0; pl < p.planes; ++pl) {
                const unsigned short b = f2bf(v);
                p.wpb[base + (size_t)pl * 64 * 8] = b;
                v -= bf2f(b);
            }
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

__global__ __launch_bounds__(256) void lstm_prepare_kernel(u32x4* head, unsigned nhead, u32x4* xbuf, unsigned nx) {
    const unsigned i0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (unsigned i = i0; i < nhead; i += stride) head[i] = (u32x4){0u, 0u, 0u, 0u};
    for (unsigned i = i0; i < nx; i += stride) xbuf[i] = (u32x4){1u, 1u, 1u, 1u};
}

struct WsLayout { size_t err, hello, progress, ready, xbuf, xbytes, ring, stamps, total; int NBG, NCL8; };
WsLayout lstm_ws_layout(int B, bool backward) {
    WsLayout l;
    l.NBG = (B + 15) / 16;
    const int ncl = 2 * l.NBG;
    l.NCL8 = (ncl + 7) / 8 * 8;
    // forward: sized for three planes (24 KiB per parity); the two-plane kernel uses the first 2 x 16 KiB of a cluster's block
    const size_t slot = backward ? (size_t)16 * 16 * 256 * 4 : (size_t)32 * 16 * 3 * 16;
    l.err = 0;                                   // 256 bytes
    l.hello = 256;                               // [clusters][HELLO_STRIDE] words; err+hello zeroed every call
    l.progress = l.hello + pgasr_align_up((size_t)ncl * HELLO_STRIDE * sizeof(unsigned), 256);   // one 128-B line per cluster
    l.ready = l.progress + pgasr_align_up((size_t)ncl * 128, 256);                    // [cluster][32 words], zeroed every call
    l.xbuf = l.ready + pgasr_align_up((size_t)ncl * 128, 256);
    l.xbytes = (size_t)ncl * 2 * slot;           // [cluster][parity][slot], filled with 0x00000001 every call
    l.ring = l.xbuf + l.xbytes;                  // [cluster][RING_STEPS][step floats]: staging ring of the helper workgroups
    l.stamps = l.ring + (size_t)ncl * (backward ? BWD_RING_STEPS * BWD_STEP_FLOATS : FWD_RING_STEPS * FWD_STEP_FLOATS) * sizeof(float);
    l.total = l.stamps;
#ifdef PGASR_LSTM_STAMPS
    l.total += (size_t)STAMP_MAX_T * 8 * sizeof(long long);
#endif
    return l;
}

}  // namespace

extern "C" size_t pgasr_lstm_pack_bytes(int which, int planes) {
    // which: 0 = forward W_hh pack, 1 = backward W_hh pack (both 1 MB per bf16 plane of 2 x 1024 x 256)
    (void)which;
    if (planes != 2 && planes != 3) return 0;
    return (size_t)2 * 64 * 8 * planes * 64 * 8 * sizeof(unsigned short);
}

extern "C" int pgasr_lstm_pack_weights(const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                                       const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                                       int in_dim, float* wih_perm, float* bias_perm,
                                       void* whh_pack_fwd, void* whh_pack_bwd, int planes, void* stream) {
    if (planes != 2 && planes != 3) return PGASR_ERR_INVALID_ARG;
    if (!w_ih_f || !w_hh_f || !b_ih_f || !b_hh_f || !w_ih_r || !w_hh_r || !b_ih_r || !b_hh_r) return PGASR_ERR_INVALID_ARG;
    if (!wih_perm || !bias_perm || !whh_pack_fwd || !whh_pack_bwd || in_dim <= 0) return PGASR_ERR_INVALID_ARG;
    PackArgs p;
    p.w_ih[0] = w_ih_f; p.w_hh[0] = w_hh_f; p.b_ih[0] = b_ih_f; p.b_hh[0] = b_hh_f;
    p.w_ih[1] = w_ih_r; p.w_hh[1] = w_hh_r; p.b_ih[1] = b_ih_r; p.b_hh[1] = b_hh_r;
    p.in_dim = in_dim; p.wih_perm = wih_perm; p.bias_perm = bias_perm;
    p.wpf = (unsigned short*)whh_pack_fwd; p.wpb = (unsigned short*)whh_pack_bwd; p.planes = planes;
    const size_t n_wih = (size_t)2 * 4 * HID * in_dim, n_pack = (size_t)2 * 64 * 8 * 64 * 8;
    const size_t n = n_wih > n_pack ? n_wih : n_pack;
    PGASR_LAUNCH_KERNEL(lstm_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
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
    PGASR_LAUNCH_KERNEL(lstm_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" size_t pgasr_lstm_workspace_bytes(int T, int B, int backward) {
    if (T <= 0 || B <= 0) return 0;
    return lstm_ws_layout(B, backward != 0).total;
}

static int lstm_launch(bool backward, float* gates, float* out, float* cbuf, const float* dout, float* dbias_part, const void* wpack,
                       const int* lengths, int T, int B, int flags, void* workspace, size_t workspace_bytes, hipStream_t st,
                       const unsigned* fed = nullptr, int fed_need = 0, float drop_p = 0.f, uint64_t drop_seed = 0,
                       uint32_t drop_offset = 0, float* out_drop = nullptr, unsigned* slab_done = nullptr) {
    if (!gates || !out || !cbuf || !wpack || !lengths || T <= 0 || B <= 0) return PGASR_ERR_INVALID_ARG;
    if (slab_done && !backward) return PGASR_ERR_INVALID_ARG;
    if (backward && !dout) return PGASR_ERR_INVALID_ARG;
    const WsLayout l = lstm_ws_layout(B, backward);
    if (!workspace || workspace_bytes < l.total) return PGASR_ERR_WORKSPACE;
    // every cluster must be co-resident, one workgroup per CU: at most 2 clusters per XCD
    if (2 * l.NBG > 16) return PGASR_ERR_UNSUPPORTED;
    char* ws = (char*)workspace;
    // one launch: zero the head block (busy counters, hello and progress words) and set every exchange word to "stale
    // for epoch 0" (bit0 = 1).  The first 16 bytes -- the error word -- are NOT touched: the flag is sticky, the owner of
    // the workspace zeroes it once and a time-out in any later launch stays visible (pgasr_lstm_error_offset)
    PGASR_LAUNCH_KERNEL(lstm_prepare_kernel, dim3(64), dim3(256), 0, st, (u32x4*)(ws + l.err + 16), (unsigned)((l.xbuf - l.err - 16) / 16),
                       (u32x4*)(ws + l.xbuf), (unsigned)(l.xbytes / 16));
    PGASR_CHECK_LAUNCH();
    LstmArgs a;
    a.gates = gates; a.out = out; a.out_drop = out_drop; a.cbuf = cbuf; a.dout = dout; a.dbias_part = dbias_part; a.wpack = (const u32x4*)wpack;
    a.xbuf = (unsigned char*)(ws + l.xbuf); a.hello = (unsigned*)(ws + l.hello); a.err = (int*)(ws + l.err);
    a.busy = (unsigned*)(ws + l.err + 64);   // 8 words inside the zeroed 256-byte head block
    a.progress = (unsigned*)(ws + l.progress);
    a.ring = (float*)(ws + l.ring); a.ready = (unsigned*)(ws + l.ready);
    a.lengths = lengths; a.T = T; a.B = B; a.NBG = l.NBG; a.NCL8 = l.NCL8;
    a.force_mode = (flags & 1) ? 1 : 0;
    a.diag = (flags >> 8) & 0xFFFF;
    a.stamps = (long long*)(ws + l.stamps);
#ifdef PGASR_LSTM_STAMPS
    if (T > STAMP_MAX_T) return PGASR_ERR_UNSUPPORTED;
#endif
    // Helpers are not optional once the loaders wait for them, and this kernel gets at most one cluster's worth of
    // extra workgroups onto an XCD that already holds a cluster (measured: B = 80 deadlocked until the time-outs
    // with 2 x 20 workgroups on one XCD): with more than 8 clusters the loaders read HBM themselves, as before.
    a.n_helpers = (2 * l.NBG <= 8 && !(flags & 4)) ? N_HELPERS : 0;      // flags bit 2: no helpers (two processes sharing one GPU)
    a.fed = fed; a.fed_mt = (int)(((size_t)T * B + 255) / 256); a.fed_need = fed_need;
    a.slab_done = slab_done;
    // a streamed sweep needs its flusher workgroup next to the cluster and XCDs left over for the consumers
    if (slab_done && (a.n_helpers == 0 || 2 * l.NBG > 4)) return PGASR_ERR_UNSUPPORTED;
    if (fed) {
        // a fed sweep waits for a GEMM that must find XCDs of its own: helpers on, at most 4 clusters (half the chip)
        if (fed_need <= 0 || a.n_helpers == 0 || 2 * l.NBG > 4) return PGASR_ERR_UNSUPPORTED;
        if ((size_t)T * B * 2 * HID * 4 * 4 >= ((size_t)1 << 31)) return PGASR_ERR_UNSUPPORTED;     // buffer-addressed loads
    }
    a.drop_on = 0; a.drop_thresh = 0; a.drop_scale = 1.f; a.drop_k0 = a.drop_k1 = a.drop_off = 0;
    if (out_drop && (backward || drop_p == 0.f || (((size_t)out_drop) & 15))) return PGASR_ERR_INVALID_ARG;
    if (drop_p != 0.f) {
        // backward: the mask is applied to the fed dout rows by the helpers; forward: to out on its way to out_drop
        if ((backward && !fed) || (!backward && !out_drop) || !(drop_p > 0.f) || !(drop_p < 1.f)) return PGASR_ERR_INVALID_ARG;
        a.drop_on = 1;
        a.drop_thresh = (uint32_t)fmin(4294967295.0, (double)drop_p * 4294967296.0);      // as pgasr_dropout
        a.drop_scale = 1.f / (1.f - drop_p);
        a.drop_k0 = (uint32_t)(drop_seed & 0xffffffffu); a.drop_k1 = (uint32_t)(drop_seed >> 32); a.drop_off = drop_offset;
    }
    dim3 grid((G_CLUSTER + a.n_helpers + (slab_done ? 1 : 0)) * l.NCL8);   // + the helper workgroups (+ the flusher) of each cluster
    // flags bit 1: three bf16 planes per fp32 operand (the fp32-faithful sweeps); `wpack` must have been packed with planes = 3
    const bool three = (flags & 2) != 0;
    if (backward) {
        if (three) PGASR_LAUNCH_KERNEL(lstm_bwd_kernel<3>, grid, dim3(LSTM_THREADS), 0, st, a);
        else       PGASR_LAUNCH_KERNEL(lstm_bwd_kernel<2>, grid, dim3(LSTM_THREADS), 0, st, a);
    } else {
        if (three) PGASR_LAUNCH_KERNEL(lstm_fwd_kernel<3>, grid, dim3(LSTM_THREADS), 0, st, a);
        else       PGASR_LAUNCH_KERNEL(lstm_fwd_kernel<2>, grid, dim3(LSTM_THREADS), 0, st, a);
    }
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_lstm_layer_fwd(float* gates, float* out, float* cbuf, const void* whh_pack_fwd,
                                    const int32_t* lengths, int T, int B, int flags,
                                    float* out_drop, float drop_p, uint64_t drop_seed, uint32_t drop_offset,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return lstm_launch(false, gates, out, cbuf, nullptr, nullptr, whh_pack_fwd, lengths, T, B, flags, workspace, workspace_bytes,
                       (hipStream_t)stream, nullptr, 0, drop_p, drop_seed, drop_offset, out_drop);
}

// Forward sweep whose input projection is produced WHILE it runs (pgasr_gemm_x3w_feed_f32 on another stream, launched
// after this call): ``fed`` = that call's tiles_done words (zeroed by the caller before this launch), ``fed_need`` =
// column tiles per direction (8H / 256).  PGASR_ERR_UNSUPPORTED when the sweep cannot be fed (more than 32
// utterances: no XCD would be left for the GEMM; helpers disabled): run the projection first and call
// pgasr_lstm_layer_fwd instead.  pgasr_lstm_fed_ok answers the same question without launching.
extern "C" int pgasr_lstm_layer_fwd_fed(float* gates, float* out, float* cbuf, const void* whh_pack_fwd,
                                        const int32_t* lengths, int T, int B, int flags, const unsigned* fed, int fed_need,
                                        float* out_drop, float drop_p, uint64_t drop_seed, uint32_t drop_offset,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!fed) return PGASR_ERR_INVALID_ARG;
    return lstm_launch(false, gates, out, cbuf, nullptr, nullptr, whh_pack_fwd, lengths, T, B, flags, workspace, workspace_bytes,
                       (hipStream_t)stream, fed, fed_need, drop_p, drop_seed, drop_offset, out_drop);
}

// Backward sweep whose dout rows are produced WHILE it runs: dout = the input gradient of the layer above, written by
// pgasr_gemm_x3w_feed_f32(order = 1) after this call; fed_need = 2H/256 column tiles per direction.  drop_p != 0: dout
// arrives WITHOUT the inter-layer dropout mask (model.py:42) and the helpers apply pgasr_dropout(p, seed, offset)'s
// mask to the rows they stage (dout itself keeps the unmasked values).
extern "C" int pgasr_lstm_layer_bwd_fed(float* gates, const float* out, const float* cbuf, const float* dout,
                                        const void* whh_pack_bwd, const int32_t* lengths, int T, int B, int flags,
                                        float* dbias_part, const unsigned* fed, int fed_need, float drop_p, uint64_t drop_seed,
                                        uint32_t drop_offset, void* workspace, size_t workspace_bytes, void* stream) {
    if (!fed) return PGASR_ERR_INVALID_ARG;
    if (dbias_part && (((size_t)dbias_part) & 15)) return PGASR_ERR_INVALID_ARG;
    return lstm_launch(true, gates, const_cast<float*>(out), const_cast<float*>(cbuf), dout, dbias_part, whh_pack_bwd, lengths, T, B,
                       flags, workspace, workspace_bytes, (hipStream_t)stream, fed, fed_need, drop_p, drop_seed, drop_offset);
}

// Backward sweep that publishes its progress for consumers running BESIDE it (the same layer's weight-gradient products,
// pgasr_lstm_wgrads_streamed): slab_done[c] (c = 2 * (16-utterance group) + direction; zeroed by the caller before this launch)
// counts publications; publication k (k = 1 .. n = pgasr_lstm_wgrad_slabs(T)) says that the d(pre-activation) rows of sweep
// steps < T - h_(n-k) are in memory and readable with agent-scope loads from any XCD.  Sweep step s is frame T-1-s for
// direction 0 and frame s for direction 1.  fed == NULL: dout is complete (pgasr_lstm_layer_bwd); else as
// pgasr_lstm_layer_bwd_fed.  Same conditions as the fed sweeps (pgasr_lstm_fed_ok).
extern "C" int pgasr_lstm_layer_bwd_streamed(float* gates, const float* out, const float* cbuf, const float* dout,
                                             const void* whh_pack_bwd, const int32_t* lengths, int T, int B, int flags,
                                             float* dbias_part, const unsigned* fed, int fed_need, float drop_p, uint64_t drop_seed,
                                             uint32_t drop_offset, unsigned* slab_done,
                                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!slab_done) return PGASR_ERR_INVALID_ARG;
    if (dbias_part && (((size_t)dbias_part) & 15)) return PGASR_ERR_INVALID_ARG;
    return lstm_launch(true, gates, const_cast<float*>(out), const_cast<float*>(cbuf), dout, dbias_part, whh_pack_bwd, lengths, T, B,
                       flags, workspace, workspace_bytes, (hipStream_t)stream, fed, fed_need, drop_p, drop_seed, drop_offset,
                       nullptr, slab_done);
}

extern "C" int pgasr_lstm_fed_ok(int T, int B, int flags) {
    if (T <= 0 || B <= 0) return 0;
    const WsLayout l = lstm_ws_layout(B, false);
    if (2 * l.NBG > 4 || (flags & 4)) return 0;
    if ((size_t)T * B * 2 * HID * 4 * 4 >= ((size_t)1 << 31)) return 0;
    return 1;
}

extern "C" int pgasr_lstm_layer_bwd(float* gates, const float* out, const float* cbuf, const float* dout,
                                    const void* whh_pack_bwd, const int32_t* lengths, int T, int B, int flags,
                                    float* dbias_part, void* workspace, size_t workspace_bytes, void* stream) {
    if (dbias_part && (((size_t)dbias_part) & 15)) return PGASR_ERR_INVALID_ARG;
    return lstm_launch(true, gates, const_cast<float*>(out), const_cast<float*>(cbuf), dout, dbias_part, whh_pack_bwd, lengths, T, B,
                       flags, workspace, workspace_bytes, (hipStream_t)stream);
}

// reads the error word written by a timed-out wait (host-side check after a sync)
extern "C" int pgasr_lstm_error_offset(int B, int backward, size_t* offset) {
    if (!offset || B <= 0) return PGASR_ERR_INVALID_ARG;
    *offset = lstm_ws_layout(B, backward != 0).err;
    return PGASR_OK;
}

// Synchronous check: waits for `stream`, reads the workspace's sticky error word and turns it into a status.
extern "C" int pgasr_lstm_status(const void* workspace, size_t workspace_bytes, int B, int backward, void* stream) {
    if (!workspace || B <= 0) return PGASR_ERR_INVALID_ARG;
    const WsLayout l = lstm_ws_layout(B, backward != 0);
    if (workspace_bytes < l.total) return PGASR_ERR_WORKSPACE;
    int word = 0;
    if (hipMemcpyAsync(&word, (const char*)workspace + l.err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
        return PGASR_ERR_LAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return PGASR_ERR_LAUNCH;
    return word != 0 ? PGASR_ERR_TIMEOUT : PGASR_OK;
}

extern "C" int pgasr_lstm_busy_offset(int B, int backward, size_t* offset) {
    if (!offset || B <= 0) return PGASR_ERR_INVALID_ARG;
    *offset = lstm_ws_layout(B, backward != 0).err + 64;
    return PGASR_OK;
}

namespace {
// One wave that holds a stream back until a sweep has registered itself (any busy counter != 0) or the
// time-out passes: gives "sweep first, GEMMs second" dispatch order across two streams.
// report (optional, 2 words): how the gate left -- [0] = 1 opened on the busy counters, 2 opened on a publication (`running`),
// 3 timed out; [1] = microseconds it held the stream.  Device evidence for tests (a host clock cannot tell a slow host from a
// timed-out gate).
__global__ void __launch_bounds__(64) stream_gate_kernel(const unsigned* words, int count, unsigned need, const unsigned* running,
                                                         long long timeout_ticks, unsigned* report) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    unsigned how = 0;
    for (;;) {
        unsigned sum = 0;
        for (int i = 0; i < count; ++i) sum += __hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool published = running && __hip_atomic_load(running, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        POLL_FENCE();
        if (sum >= need) { how = 1; break; }
        if (published) { how = 2; break; }
        if (wall_clock64() - t0 > timeout_ticks) { how = 3; break; }
        __builtin_amdgcn_s_sleep(16);
    }
    if (report) {
        __hip_atomic_store(report, how, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(report + 1, (unsigned)((wall_clock64() - t0) / 100), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
}  // namespace

namespace {
// words[1] = 1 iff words[0] became non-zero within the time-out: did a kernel on ANOTHER stream run while this one was
// resident?  (Kernel serialisation -- counter-collecting profilers, launch-blocking debug modes, a single hardware
// queue -- would make a fed sweep wait for a GEMM that cannot start; callers probe once and fall back.)
__global__ void __launch_bounds__(64) stream_probe_kernel(unsigned* words, long long timeout_ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    unsigned seen = 0;
    for (;;) {
        seen = __hip_atomic_load(words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        POLL_FENCE();
        if (seen != 0u || wall_clock64() - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
    __hip_atomic_store(words + 1, seen != 0u ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

extern "C" int pgasr_stream_probe(unsigned* words, int timeout_us, void* stream) {
    if (!words || timeout_us <= 0 || timeout_us > 1000000) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(stream_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, words, (long long)timeout_us * 100);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

extern "C" int pgasr_stream_gate(const unsigned* words, int count, int timeout_us, void* stream) {
    if (!words || count <= 0 || count > 64 || timeout_us < 0 || timeout_us > 100000) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(stream_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, words, count, 1u, (const unsigned*)nullptr, (long long)timeout_us * 100, (unsigned*)nullptr);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

// The same gate for consumers that WAIT for the sweep (pgasr_lstm_wgrads_streamed): holds `stream` until the busy counters add up to
// `need` -- every cluster of the sweep has registered, i.e. all of its workgroups are resident -- since workgroups that poll for the
// sweep's publications must not take CUs the sweep still needs.  Bounded like the other one (timeout_us <= 100000).
extern "C" int pgasr_stream_gate_sum(const unsigned* words, int count, int need, const unsigned* running, int timeout_us, void* stream) {
    return pgasr_stream_gate_report(words, count, need, running, timeout_us, nullptr, stream);
}

extern "C" int pgasr_stream_gate_report(const unsigned* words, int count, int need, const unsigned* running, int timeout_us,
                                        unsigned* report, void* stream) {
    if (!words || count <= 0 || count > 64 || need <= 0 || timeout_us < 0 || timeout_us > 100000) return PGASR_ERR_INVALID_ARG;
    PGASR_LAUNCH_KERNEL(stream_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, words, count, (unsigned)need, running, (long long)timeout_us * 100, report);
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}

#ifdef PGASR_LSTM_DIAG
extern "C" int pgasr_diag_lstm_late(unsigned* out, int reset) {      // host copy of lstm_diag_late [8 clusters][64 words] (the LAST sweep that ran)
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lstm_diag_late), sizeof(unsigned) * 8 * 64) != hipSuccess) return 1;
    if (reset) { static unsigned z[8 * 64]; if (hipMemcpyToSymbol(HIP_SYMBOL(lstm_diag_late), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
