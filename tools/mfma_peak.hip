// Microbenchmark (round 4): what does the WHOLE chip sustain on bare v_mfma_f32_32x32x16_bf16 with random operands -- every CU,
// one or two waves per SIMD, operands in registers -- and at which clock?  The price list for the six-product GEMMs (gemm_x6.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak.bin && tools/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void k(const unsigned* seed, float* sink, long long* clk, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 A[6], B[6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 8; ++j) {
            unsigned s = seed[(threadIdx.x * 131 + i * 17 + j * 7 + blockIdx.x) & 4095];
            A[i][j] = (__bf16)((float)(int)(s & 0xFFFF) * (1.f / 65536.f) - 0.5f);
            B[i][j] = (__bf16)((float)(int)(s >> 16) * (1.f / 65536.f) - 0.5f);
        }
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[p], B[(p + m) % 6], acc[m], 0, 0, 0);
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    float t = 0.f;
    for (int m = 0; m < 8; ++m) t += acc[m][lane & 15];
    if (t == 1234.5f) sink[0] = t;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

int main() {
    unsigned* seed; float* sink; long long* clk;
    hipMalloc(&seed, 4096 * 4); hipMalloc(&sink, 64); hipMalloc(&clk, 64);
    unsigned h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned)rand() * 2654435761u;
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) {
        for (int iters : {2000, 20000}) {
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, seed, sink, clk, 200);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, seed, sink, clk, iters);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            const double mfmas = (double)iters * 48 * (threads / 64) * 256;
            const double tf = mfmas * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
            printf("%d waves/CU, %d iterations: %.3f ms  %.0f TF = %.1f %% of 2500; in-kernel clock %.0f MHz; cycles per MFMA per SIMD %.1f\n",
                   threads / 64, iters, ms, tf, tf / 25.0, (double)c[0] / (double)c[1] * 100.0, (double)c[0] / ((double)iters * 48 * (threads / 256)));
        }
    }
    return 0;
}
