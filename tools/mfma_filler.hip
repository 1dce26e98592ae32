// Microbenchmark (round 4): what does a filler instruction COST beside back-to-back v_mfma_f32_32x32x16_bf16 -- one wave per SIMD,
// every CU busy, N fillers of one kind behind every MFMA?  Prints cycles per MFMA for N = 0..8 and several filler kinds.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_filler.hip -o tools/mfma_filler.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int KIND, int N>
__global__ __launch_bounds__(256) void k(const unsigned* seed, float* sink, long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[16384];
    const int lane = threadIdx.x & 63;
    bf16x8 A[4], B[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            unsigned s = seed[(threadIdx.x * 131 + i * 17 + j * 7 + blockIdx.x) & 4095];
            A[i][j] = (__bf16)((float)(int)(s & 0xFFFF) * (1.f / 65536.f) - 0.5f);
            B[i][j] = (__bf16)((float)(int)(s >> 16) * (1.f / 65536.f) - 0.5f);
        }
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = seed[i & 4095];
    __syncthreads();
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float v[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { v[i] = 0.001f * (lane + i); u[i] = seed[(lane + i) & 4095]; }
    const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds + (unsigned)threadIdx.x * 16u;
    u32x4 rd[4];
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m & 3], B[(m >> 1) & 3], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < N; ++f) {
                const int x = (m * N + f) & 7;
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[x]) : "v"(v[(x + 3) & 7]));
                else if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[x]) : "v"(v[(x + 1) & 7]), "v"(v[(x + 2) & 7]));
                else if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(rd[x & 3]) : "v"(laddr) : "memory");
                else if (KIND == 3) asm volatile("ds_write_b64 %0, %1" :: "v"(laddr), "v"((u32x2){u[x], u[(x + 1) & 7]}) : "memory");
                else if (KIND == 4) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[x]) : "v"(u[(x + 1) & 7]));
                else if (KIND == 5) asm volatile("s_nop 0");
                else if (KIND == 6) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[x]) : "v"(v[x]));      // dependent chain on one register
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 2 || KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long c1 = clock64();
    float t = 0.f;
    for (int m = 0; m < 8; ++m) t += acc[m][lane & 15];
    for (int i = 0; i < 8; ++i) t += v[i] + (float)u[i];
    if (KIND == 2) t += (float)(rd[0].x + rd[1].y + rd[2].z + rd[3].w);
    if (t == 1234.5f) sink[0] = t;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

template <int KIND, int N>
static void run(const unsigned* seed, float* sink, long long* clk, const char* name) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 0, 0, seed, sink, clk, 100);
    hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 0, 0, seed, sink, clk, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("  %-22s N=%d: %.1f cycles per MFMA\n", name, N, (double)c / (iters * 8.0));
}
#define ROW(KIND, NAME) run<KIND, 0>(seed, sink, clk, NAME); run<KIND, 1>(seed, sink, clk, NAME); run<KIND, 2>(seed, sink, clk, NAME); \
    run<KIND, 3>(seed, sink, clk, NAME); run<KIND, 4>(seed, sink, clk, NAME); run<KIND, 6>(seed, sink, clk, NAME); run<KIND, 8>(seed, sink, clk, NAME);

int main() {
    unsigned* seed; float* sink; long long* clk;
    (void)hipMalloc(&seed, 4096 * 4); (void)hipMalloc(&sink, 64); (void)hipMalloc(&clk, 64);
    unsigned h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned)rand() * 2654435761u;
    (void)hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    printf("one wave per SIMD, every CU, N fillers behind each v_mfma_f32_32x32x16_bf16:\n");
    ROW(0, "v_add_f32")
    ROW(1, "v_cvt_pk_bf16_f32")
    ROW(4, "v_lshlrev_b32")
    ROW(6, "v_sub_f32 (dependent)")
    ROW(2, "ds_read_b128")
    ROW(3, "ds_write_b64")
    ROW(5, "s_nop 0")
    return 0;
}
