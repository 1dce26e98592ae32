// Microbenchmark (development aid): one-way latency of the LSTM hand-off primitive between two
// workgroups -- 1 KiB of tagged words written by one wave (plain or sc1 stores), polled by the
// other side with sc1 loads.  build: hipcc --offload-arch=gfx950 -O3 tools/pingpong.hip -o /tmp/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ __launch_bounds__(64) void pingpong(unsigned char* buf, int iters, int other_slot, int sc1_store, unsigned* xcc, long long* cycles) {
    const int me = (blockIdx.x == 0) ? 0 : (blockIdx.x == other_slot ? 1 : -1);
    if (me < 0) return;
    const int lane = threadIdx.x;
    if (lane == 0) xcc[me] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 0xF;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 4096, 0x00020000);
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        // ping: block 0 writes on even phases, block 1 on odd phases
        for (int ph = 0; ph < 2; ++ph) {
            const unsigned tag = (unsigned)(it * 2 + ph + 1);
            if (ph == me) {
                u32x4 v = {tag, tag, tag, tag};
                if (sc1_store) __builtin_amdgcn_raw_buffer_store_b128(v, rs, ph * 1024 + lane * 16, 0, 16);
                else __builtin_amdgcn_raw_buffer_store_b128(v, rs, ph * 1024 + lane * 16, 0, 0);
            } else {
                unsigned* w = reinterpret_cast<unsigned*>(buf + ph * 1024 + lane * 16);
                long long guard = 0;
                while (true) {   // atomic (sc1) loads: never hoisted, every lane checks first and last word of its 16 B
                    const unsigned a = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned b = __hip_atomic_load(w + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!__any(a != tag || b != tag)) break;
                    if (++guard > 50000000) { if (lane == 0) cycles[2] = it; return; }
                }
            }
        }
    }
    if (lane == 0) cycles[me] = clock64() - t0;
}
int main() {
    unsigned char* buf; unsigned* xcc; long long* cyc;
    hipMalloc(&buf, 4096); hipMalloc(&xcc, 8); hipMalloc(&cyc, 32); hipMemset(cyc, 0, 32);
    for (int other : {8, 1}) for (int sc1 : {0, 1}) {
        if (other == 1 && sc1 == 0) continue;   // plain stores never become visible across XCDs
        hipMemset(buf, 0, 4096);
        const int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(pingpong, dim3(16), dim3(64), 0, 0, buf, iters, other, sc1, xcc, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned hx[2]; hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost);
        long long hc[3]; hipMemcpy(hc, cyc, 24, hipMemcpyDeviceToHost);
        printf("partner block %d (xcc %u vs %u) %s stores: %.3f us per one-way hand-off (%lld cycles each, stuck_at=%lld, err=%s)\n", other, hx[0], hx[1],
               sc1 ? "sc1" : "plain", ms * 1e3 / (iters * 2), hc[0] / (iters * 2), hc[2], hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
