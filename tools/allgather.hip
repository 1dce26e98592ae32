// Microbenchmark (development aid): the LSTM forward exchange pattern without the math.
// NCL clusters of G workgroups (members share blockIdx % 8); per step every member publishes a
// tagged 1-KiB block (G=16) and each of its 4 waves loads a 4-KiB quarter of the cluster's 16 KiB,
// retrying until fresh.  Reports microseconds per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int MODE>   // 0: wave-0 1KiB b128 store after barrier (old), 1: every thread 2x b16 stores (current)
__global__ __launch_bounds__(256) void allgather(unsigned char* xbuf, int steps, int ncl, int work_cycles, long long* cycles) {
    const int cl = blockIdx.x % 8, g = blockIdx.x / 8;
    if (cl >= ncl) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    constexpr unsigned SLOT = 16384;
    unsigned char* xb = xbuf + (size_t)cl * 2 * SLOT;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 2 * SLOT, 0x00020000);
    __shared__ unsigned short hs[512];
    long long t0 = clock64();
    unsigned acc = 0;
    for (int step = 0; step < steps; ++step) {
        const unsigned e = (step >> 1) & 1;
        if (step > 0) {
            const unsigned pbase = ((step - 1) & 1) * SLOT;
            const unsigned want = (((step - 1) >> 1) & 1) ? 0x00000001u : 0x00010000u;
            while (true) {
                asm volatile("" ::: "memory");
                unsigned bad = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned off = pbase + (unsigned)((((4 * (2 * w + i) + q) * 16 + n) * 2) * 16);
                    u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
                    u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 16);
                    bad |= ((a.x ^ want) | (a.y ^ want) | (a.z ^ want) | (a.w ^ want) | (b.x ^ want) | (b.y ^ want) | (b.z ^ want) | (b.w ^ want)) & 0x00010001u;
                    acc += a.x + b.w;
                }
                if (!__any(bad != 0)) break;
            }
        }
        // stand-in for the MFMA + cell math
        if (work_cycles > 0) { const long long t1 = clock64(); while (clock64() - t1 < work_cycles) {} }
        __syncthreads();   // barrier B1 of the real kernel
        const unsigned short hv = (unsigned short)(((acc + step) & 0xFFFEu) | (((tid & 1) ? (1u - e) : e)));
        const unsigned pu = tid & 15, pn = tid >> 4;
        if (MODE == 1) {
            const unsigned off = (step & 1) * SLOT + g * 1024u + (((pu >> 3) * 16 + pn) * 2) * 16 + (pu & 7) * 2;
            __builtin_amdgcn_raw_buffer_store_b16(hv, rs, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16(hv, rs, off + 16, 0, 0);
        } else {
            hs[(((pu >> 3) * 16 + pn) * 2 + 0) * 8 + (pu & 7)] = hv;
            hs[(((pu >> 3) * 16 + pn) * 2 + 1) * 8 + (pu & 7)] = hv;
            __syncthreads();
            if (w == 0) {
                u32x4 v = *reinterpret_cast<const u32x4*>(&hs[lane * 8]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (step & 1) * SLOT + g * 1024u + lane * 16, 0, 0);
            }
        }
    }
    if (tid == 0 && g == 0) cycles[cl] = clock64() - t0 + (acc & 1);
}
int main() {
    unsigned char* buf; long long* cyc;
    hipMalloc(&buf, 8 * 2 * 16384); hipMalloc(&cyc, 64);
    const int steps = 3000;
    for (int mode : {1}) for (int ncl : {4}) for (int work : {0, 250, 500, 1000, 1500, 2000, 3000}) {
        hipMemsetD32((hipDeviceptr_t)buf, 1, 8 * 2 * 16384 / 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(allgather<0>, dim3(128), dim3(256), 0, 0, buf, steps, ncl, work, cyc);
        else hipLaunchKernelGGL(allgather<1>, dim3(128), dim3(256), 0, 0, buf, steps, ncl, work, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long hc[8]; hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
        printf("mode %d (%s) clusters %d work %4d cycles: %.3f us/step (%lld cycles/step) err=%s\n", mode,
               mode ? "2B stores, 1 barrier" : "wave-0 1KiB store, 2 barriers", ncl, work, ms * 1e3 / steps, hc[0] / steps,
               hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
