// Batched Levenshtein distance for gfx950 (metrics.py:4-21).
//
// One wave per (reference, hypothesis) pair.  The DP row over the reference positions
// j = 0..n lives in registers, J consecutive cells per lane.  A row update is
//     x[j]   = min(prev[j] + 1, prev[j-1] + (ref[j-1] != h))      (up, diagonal)
//     new[j] = min_{k<=j} (x[k] + j - k)                           (left chain)
// and the left chain is a min-plus prefix scan: new[j] - j = prefixmin(x[k] - k), done as a
// serial pass inside each lane plus a 6-step cross-lane scan.  Rows are a serial chain of
// |hyp| steps, so this is latency bound (integer work, no LDS, no barriers).
#include "common.h"

namespace {

constexpr int ED_BIG = 0x3fffffff;

template <int J>
__global__ __launch_bounds__(64) void edit_distance_kernel(
    const int32_t* __restrict__ ref, const int32_t* __restrict__ ref_len, int ref_stride,
    const int32_t* __restrict__ hyp, const int32_t* __restrict__ hyp_len, int hyp_stride,
    int32_t* __restrict__ dist, int32_t* __restrict__ prefix_dist) {
    const int n_pair = blockIdx.x;
    const int lane = threadIdx.x;
    int n = ref_len[n_pair]; n = n < 0 ? 0 : (n > ref_stride ? ref_stride : n);
    int m = hyp_len[n_pair]; m = m < 0 ? 0 : (m > hyp_stride ? hyp_stride : m);
    const int32_t* r = ref + (size_t)n_pair * ref_stride;
    const int32_t* h = hyp + (size_t)n_pair * hyp_stride;
    int32_t* pd = prefix_dist ? prefix_dist + (size_t)n_pair * (hyp_stride + 1) : nullptr;
    // The distance is symmetric, and the rows are the serial chain: unless per-prefix distances
    // of the hypothesis are requested, iterate over the SHORTER sequence and keep the longer one
    // across the lanes (an untrained model emits ~T tokens against ~T/10 reference labels).
    if (!pd && m > n) { const int32_t* tp = r; r = h; h = tp; const int tn = n; n = m; m = tn; }

    const int j0 = lane * J;
    int rt[J];    // ref token left of cell j (ref[j-1]); unused for j == 0 or j > n
    int prev[J];  // dp[i-1][j]
#pragma unroll
    for (int q = 0; q < J; ++q) {
        const int j = j0 + q;
        rt[q] = (j >= 1 && j <= n) ? r[j - 1] : -1;
        prev[q] = (j <= n) ? j : ED_BIG;
    }
    const int own_lane = n / J, own_q = n % J;
    auto cell_n = [&](const int (&row)[J]) {
        int v = 0;
#pragma unroll
        for (int q = 0; q < J; ++q) v = (q == own_q) ? row[q] : v;
        return __shfl(v, own_lane, 64);
    };
    if (pd && lane == 0) pd[0] = n;

    int hn = (m > 0) ? h[0] : 0;
    for (int i = 1; i <= m; ++i) {
        const int tok = hn;
        if (i < m) hn = h[i];
        const int left_prev = __shfl_up(prev[J - 1], 1, 64);  // dp[i-1][j0-1]
        int y[J];
        int diag = left_prev;
#pragma unroll
        for (int q = 0; q < J; ++q) {
            const int j = j0 + q;
            int x;
            if (j == 0) x = i;
            else if (j > n) x = ED_BIG;
            else x = min(prev[q] + 1, diag + (rt[q] != tok ? 1 : 0));
            diag = prev[q];
            y[q] = (q == 0) ? x : min(x, y[q - 1] + 1);
        }
        // exclusive cross-lane prefix-min of z = y[J-1] - (j0 + J - 1)
        int z = y[J - 1] - (j0 + J - 1);
        int inc = z;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o, 64);
            if (lane >= o) inc = min(inc, up);
        }
        int exc = __shfl_up(inc, 1, 64);
        if (lane == 0) exc = ED_BIG;
#pragma unroll
        for (int q = 0; q < J; ++q) {
            const int j = j0 + q;
            int v = y[q];
            if (exc < ED_BIG / 2) v = min(v, exc + j);
            prev[q] = (j <= n) ? v : ED_BIG;
        }
        if (pd) {
            const int d = cell_n(prev);
            if (lane == 0) pd[i] = d;
        }
    }
    const int d = cell_n(prev);
    if (lane == 0) dist[n_pair] = d;
}

}  // namespace

extern "C" int pgasr_edit_distance(const int32_t* ref, const int32_t* ref_len, int ref_stride,
                                   const int32_t* hyp, const int32_t* hyp_len, int hyp_stride,
                                   int N, int32_t* dist, int32_t* prefix_dist, void* stream) {
    if (!ref_len || !hyp_len || !dist || N <= 0 || ref_stride < 0 || hyp_stride < 0) return PGASR_ERR_INVALID_ARG;
    if ((ref_stride > 0 && !ref) || (hyp_stride > 0 && !hyp)) return PGASR_ERR_INVALID_ARG;
    const int col_max = prefix_dist ? ref_stride : (ref_stride > hyp_stride ? ref_stride : hyp_stride);
    if (col_max > 4095) return PGASR_ERR_UNSUPPORTED;
    const int need = (col_max + 1 + 63) / 64;  // cells per lane
    hipStream_t st = (hipStream_t)stream;
    // A pair is one wave's serial chain, and inside the train step it runs BESIDE the CTC lattice, whose workgroups are chains
    // too: two chains on one SIMD slow each other, and both kernels are as slow as their slowest workgroup.  Up to 128 pairs (the
    // step's 2 B) each get an LDS reservation nobody else leaves room for, i.e. an idle CU of their own (loss section 0.48 ->
    // 0.45 ms, step -0.04 ms; PGASR_ED_LDS=0 switches it off); bulk calls keep many waves per CU.
    static const int ed_env = [] {
        const char* e = getenv("PGASR_ED_LDS");
        int want = e ? atoi(e) : 156 * 1024, dev = 0, cap = 0;
        // never ask for more than the device gives one workgroup (round-3 advice: a smaller-LDS part would fail every small call)
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && cap > 0 && want > cap)
            want = cap;
        return want > 0 ? want : 0;
    }();
    int ed_lds = N <= 128 ? ed_env : 0;
#define ED_LAUNCH(JJ)                                                                             \
    do { if (ed_lds > 0 && hipFuncSetAttribute((const void*)edit_distance_kernel<JJ>, hipFuncAttributeMaxDynamicSharedMemorySize, ed_lds) != hipSuccess) \
             ed_lds = 0;                  /* no reservation rather than a failed launch: the pairs then share CUs (slower beside a lattice, never wrong) */ \
    PGASR_LAUNCH_KERNEL(edit_distance_kernel<JJ>, dim3(N), dim3(64), (size_t)ed_lds, st, ref, ref_len,          \
                       ref_stride, hyp, hyp_len, hyp_stride, dist, prefix_dist); } while (0)
    if (need <= 1) ED_LAUNCH(1);
    else if (need <= 2) ED_LAUNCH(2);
    else if (need <= 4) ED_LAUNCH(4);
    else if (need <= 8) ED_LAUNCH(8);
    else if (need <= 16) ED_LAUNCH(16);
    else if (need <= 32) ED_LAUNCH(32);
    else ED_LAUNCH(64);
#undef ED_LAUNCH
    PGASR_CHECK_LAUNCH();
    return PGASR_OK;
}
