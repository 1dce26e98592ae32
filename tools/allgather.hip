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

// mode 2: publish like mode 1 (2-byte stores), poll with TWO loads in flight, issued `gap` sleep units apart
__global__ __launch_bounds__(256) void allgather2(unsigned char* xbuf, int steps, int ncl, int work_cycles, int gap, long long* cycles) {
    const int cl = blockIdx.x % 8, g = blockIdx.x / 8;
    if (cl >= ncl) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    constexpr unsigned SLOT = 16384;
    unsigned char* xb = xbuf + (size_t)cl * 2 * SLOT;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 2 * SLOT, 0x00020000);
    long long t0 = clock64();
    unsigned acc = 0;
    for (int step = 0; step < steps; ++step) {
        const unsigned e = (step >> 1) & 1;
        if (step > 0) {
            const unsigned pbase = ((step - 1) & 1) * SLOT;
            const unsigned want = (((step - 1) >> 1) & 1) ? 0x00000001u : 0x00010000u;
            const unsigned off0 = pbase + (unsigned)((((4 * (2 * w + 0) + q) * 16 + n) * 2) * 16);
            const unsigned off1 = pbase + (unsigned)((((4 * (2 * w + 1) + q) * 16 + n) * 2) * 16);
            u32x4 a[4], b[4];
#define LOAD4(r) do { asm volatile("" ::: "memory"); r[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, off0, 0, 16); r[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, off0 + 16, 0, 16); \
                      r[2] = __builtin_amdgcn_raw_buffer_load_b128(rs, off1, 0, 16); r[3] = __builtin_amdgcn_raw_buffer_load_b128(rs, off1 + 16, 0, 16); } while (0)
#define BAD4(r) ((((r[0].x ^ want) | (r[0].y ^ want) | (r[0].z ^ want) | (r[0].w ^ want) | (r[1].x ^ want) | (r[1].y ^ want) | (r[1].z ^ want) | (r[1].w ^ want) | \
                   (r[2].x ^ want) | (r[2].y ^ want) | (r[2].z ^ want) | (r[2].w ^ want) | (r[3].x ^ want) | (r[3].y ^ want) | (r[3].z ^ want) | (r[3].w ^ want)) & 0x00010001u))
            LOAD4(a);
            for (int k = 0; k < gap; ++k) __builtin_amdgcn_s_sleep(1);
            LOAD4(b);
            while (true) {
                if (!__any(BAD4(a) != 0)) { acc += a[0].x + a[3].w; break; }
                LOAD4(a);
                if (!__any(BAD4(b) != 0)) { acc += b[0].x + b[3].w; break; }
                LOAD4(b);
            }
        }
        if (work_cycles > 0) { const long long t1 = clock64(); while (clock64() - t1 < work_cycles) {} }
        __syncthreads();
        const unsigned short hv = (unsigned short)(((acc + step) & 0xFFFEu) | (((tid & 1) ? (1u - e) : e)));
        const unsigned pu = tid & 15, pn = tid >> 4;
        const unsigned off = (step & 1) * SLOT + g * 1024u + (((pu >> 3) * 16 + pn) * 2) * 16 + (pu & 7) * 2;
        __builtin_amdgcn_raw_buffer_store_b16(hv, rs, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b16(hv, rs, off + 16, 0, 0);
    }
    if (tid == 0 && g == 0) cycles[cl] = clock64() - t0 + (acc & 1);
}
int main() {
    unsigned char* buf; long long* cyc;
    hipMalloc(&buf, 8 * 2 * 16384); hipMalloc(&cyc, 64);
    const int steps = 3000;
    for (int mode : {1}) for (int ncl : {4}) for (int work : {1000, 2000}) {
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
    for (int gap : {0, 2, 4, 6, 8}) for (int work : {1000, 2000}) {
        hipMemsetD32((hipDeviceptr_t)buf, 1, 8 * 2 * 16384 / 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(allgather2, dim3(128), dim3(256), 0, 0, buf, steps, 4, work, gap, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode 2 (two polls in flight, gap %d) clusters 4 work %4d cycles: %.3f us/step err=%s\n", gap, work, ms * 1e3 / steps,
               hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
