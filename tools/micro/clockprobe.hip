// Micro-benchmark: what one wave per SIMD really pays, in wall-clock ns (s_memrealtime, 100 MHz), for a dependent FMA chain,
// MFMA streams, an LDS round trip and a 4-wave barrier, with only a few workgroups on the chip (the decode situation).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* t, int n) {
    __shared__ float lds[1024];
    const int tid = threadIdx.x;
    float x = out[tid];
    unsigned long long t0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    unsigned long long t1 = wall_clock64(), c1 = clock64();
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(x * 1e-9f + j); b[j] = (_Float16)(0.001f * j); }
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t2 = wall_clock64();
    for (int i = 0; i < n / 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
    }
    unsigned long long t3 = wall_clock64();
    for (int i = 0; i < n; ++i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[0], 0, 0, 0);
    unsigned long long t4 = wall_clock64();
    float y = x;
    for (int i = 0; i < n / 8; ++i) {      // LDS write -> barrier -> read
        lds[tid] = y;
        __syncthreads();
        y += lds[(tid + 64) & 255];
        __syncthreads();
    }
    unsigned long long t5 = wall_clock64();
    for (int i = 0; i < n / 8; ++i) { __builtin_amdgcn_s_barrier(); }
    unsigned long long t6 = wall_clock64();
    float z = y;
    for (int i = 0; i < n / 8; ++i) z = 1.0f / (1.0f + __expf(-z));     // dependent exp + rcp
    unsigned long long t7 = wall_clock64();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1];
    out[tid] = s + z;
    if (tid == 0) {
        t[0] = t1 - t0; t[1] = c1 - c0; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4; t[5] = t6 - t5; t[6] = t7 - t6;
    }
}
int main() {
    float* out; unsigned long long* t;
    hipMalloc(&out, 4096); hipMalloc(&t, 64); hipMemset(out, 0, 4096);
    const int n = 80000;
    for (int grid : {1, 17, 256, 1024}) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, out, t, n);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        printf("grid %4d: dependent fma %.2f ns (%.2f clk64 cycles) | mfma 16x16x32 f16: 8 streams %.1f ns, dependent %.1f ns | lds+2 barriers %.0f ns | s_barrier %.0f ns | exp+rcp %.0f ns\n",
               grid, h[0] * 10.0 / n, (double)h[1] / n, h[2] * 10.0 / n, h[3] * 10.0 / n, h[4] * 10.0 / (n / 8), h[5] * 10.0 / (n / 8), h[6] * 10.0 / (n / 8));
    }
    return 0;
}
