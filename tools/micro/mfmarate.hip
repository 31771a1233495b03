// Micro-benchmark: sustained rate of the f16 / bf16 MFMA shapes on gfx950, per SIMD, with 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* t, int n) {
    const int tid = threadIdx.x;
    f16x8 a, b; bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (tid + j)); b[j] = (_Float16)(0.002f * j); ab[j] = (__bf16)(0.001f * (tid + j)); bb[j] = (__bf16)(0.002f * j); }
    float s = 0.f;
    unsigned long long t0 = wall_clock64();
    if (SHAPE == 0) {          // 16x16x32 f16, 8 accumulators
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = f32x4{(float)(j + tid), (float)(2 * j + tid), (float)(3 * j + tid), (float)(5 * j + tid)};   // distinct in every component: no folding, no accumulator shuffles in the loop
        for (int i = 0; i < n / 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));   // (the builtin form made hipcc shuffle accumulators inside the loop)
        }
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else if (SHAPE == 1) {   // 32x32x16 f16, 4 accumulators
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = (float)(j + e + tid);
        for (int i = 0; i < n / 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    } else if (SHAPE == 2) {   // 16x16x32 bf16
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = f32x4{(float)(j + tid), (float)(2 * j + tid), (float)(3 * j + tid), (float)(5 * j + tid)};   // distinct in every component: no folding, no accumulator shuffles in the loop
        for (int i = 0; i < n / 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(ab), "v"(bb));
        }
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {                   // 32x32x16 bf16
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = (float)(j + e + tid);
        for (int i = 0; i < n / 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    }
    unsigned long long t1 = wall_clock64();
    out[tid] = s;
    if (tid == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* t;
    hipMalloc(&out, 4096); hipMalloc(&t, 64);
    const int n = 160000;
    const char* names[] = {"16x16x32 f16", "32x32x16 f16", "16x16x32 bf16", "32x32x16 bf16"};
    const double flops[] = {16384, 32768, 16384, 32768};
    for (int shape = 0; shape < 4; ++shape)
        for (int grid : {256, 512, 1024}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(rate<0>, dim3(grid), dim3(256), 0, 0, out, t, n);
            if (shape == 1) hipLaunchKernelGGL(rate<1>, dim3(grid), dim3(256), 0, 0, out, t, n);
            if (shape == 2) hipLaunchKernelGGL(rate<2>, dim3(grid), dim3(256), 0, 0, out, t, n);
            if (shape == 3) hipLaunchKernelGGL(rate<3>, dim3(grid), dim3(256), 0, 0, out, t, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double agg = (double)grid * 4 * n * flops[shape] / (ms * 1e-3) / 1e15;      // whole kernel, by the event clock
            hipDeviceSynchronize();
            unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
            const double ns = h * 10.0, per = ns / n;                 // per MFMA of ONE wave
            const int wps = grid / 256;                                 // waves per SIMD
            const double simd_ns = per / wps;                           // per MFMA of a SIMD
            printf("%-14s %d workgroup(s)/CU: %.1f ns per MFMA in wave 0 (if co-resident: %.1f ns per SIMD, %.0f flop/ns/SIMD) | whole kernel by events: %.2f PFLOP/s\n",
                   names[shape], wps, per, simd_ns, flops[shape] / simd_ns, agg);
        }
    return 0;
}
