// probe: how v_mfma_f32_16x16x32_{f16,bf16} rounds its 32-term dot product + accumulator.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfmaround.hip -o tools/micro/mfmaround && tools/micro/mfmaround
// Each case sets C = cval and the 32 products of result element (0,0) to chosen values (A row 0 = a[k], B column 0 = b[k]),
// and prints the fp32 result next to the exactly rounded (RNE) one computed in double on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool BF>
__global__ void k(const float* a, const float* b, float cval, float* out) {   // a[32], b[32]
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    f32x4 acc = {cval, cval, cval, cval};
    if (BF) {
        bf16x8 av, bv;
        for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(c == 0 ? a[8 * q + j] : 0.f); bv[j] = (__bf16)(c == 0 ? b[8 * q + j] : 0.f); }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
    } else {
        f16x8 av, bv;
        for (int j = 0; j < 8; ++j) { av[j] = (_Float16)(c == 0 ? a[8 * q + j] : 0.f); bv[j] = (_Float16)(c == 0 ? b[8 * q + j] : 0.f); }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    }
    if (lane == 0) out[0] = acc[0];
}
// chained accumulation: n MFMAs into the same accumulator, each adding the same 32 products
template <bool BF>
__global__ void kchain(const float* a, const float* b, float cval, int n, float* out) {
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    f32x4 acc = {cval, cval, cval, cval};
    bf16x8 av, bv; f16x8 ah, bh;
    for (int j = 0; j < 8; ++j) {
        av[j] = (__bf16)(c == 0 ? a[8 * q + j] : 0.f); bv[j] = (__bf16)(c == 0 ? b[8 * q + j] : 0.f);
        ah[j] = (_Float16)(c == 0 ? a[8 * q + j] : 0.f); bh[j] = (_Float16)(c == 0 ? b[8 * q + j] : 0.f);
    }
    for (int i = 0; i < n; ++i) {
        if (BF) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    }
    if (lane == 0) out[0] = acc[0];
}
static float *da, *db, *dout;
static void run(const char* tag, const float* a, const float* b, float cval) {
    hipMemcpy(da, a, 128, hipMemcpyHostToDevice); hipMemcpy(db, b, 128, hipMemcpyHostToDevice);
    double ex = cval;
    for (int i = 0; i < 32; ++i) ex += (double)a[i] * (double)b[i];
    float r[2];
    hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, da, db, cval, dout); hipMemcpy(&r[0], dout, 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, da, db, cval, dout); hipMemcpy(&r[1], dout, 4, hipMemcpyDeviceToHost);
    const float rne = (float)ex;
    printf("%-44s exact %.10e  rne %.10e | f16 %.10e (%+.2f ulp)  bf16 %.10e (%+.2f ulp)\n", tag, ex, rne, r[0],
           (r[0] - ex) / (double)(nextafterf(fabsf(rne), INFINITY) - fabsf(rne)), r[1],
           (r[1] - ex) / (double)(nextafterf(fabsf(rne), INFINITY) - fabsf(rne)));
}
int main() {
    hipMalloc(&da, 128); hipMalloc(&db, 128); hipMalloc(&dout, 4);
    float a[32], b[32];
    auto clear = [&]() { for (int i = 0; i < 32; ++i) { a[i] = 0.f; b[i] = 1.f; } };
    const float u = ldexpf(1.f, -23);           // ulp of 1.0
    clear(); a[0] = 0.5f * u;   run("C=1, one product = 0.5 ulp (tie)", a, b, 1.f);
    clear(); a[0] = 0.75f * u;  run("C=1, one product = 0.75 ulp", a, b, 1.f);
    clear(); a[0] = 0.25f * u;  run("C=1, one product = 0.25 ulp", a, b, 1.f);
    clear(); a[0] = -0.25f * u; run("C=1, one product = -0.25 ulp", a, b, 1.f);
    clear(); a[0] = -0.75f * u; run("C=1, one product = -0.75 ulp", a, b, 1.f);
    clear(); for (int i = 0; i < 32; ++i) a[i] = 0.125f * u;  run("C=1, 32 products of 1/8 ulp (sum 4 ulp)", a, b, 1.f);
    clear(); for (int i = 0; i < 32; ++i) a[i] = 0.375f * u;  run("C=1, 32 products of 3/8 ulp (sum 12 ulp)", a, b, 1.f);
    clear(); for (int i = 0; i < 32; ++i) a[i] = ldexpf(1.f, -30);  run("C=1, 32 products of 2^-30 (sum 2^-25)", a, b, 1.f);
    clear(); for (int i = 0; i < 32; ++i) a[i] = (i & 1) ? -0.375f * u : 0.375f * u; a[0] = 0.75f * u; run("C=1, alternating +-3/8 ulp, net +3/8", a, b, 1.f);
    clear(); a[0] = 1.f; a[1] = 0.75f * u; run("C=0, products 1 and 0.75 ulp", a, b, 0.f);
    clear(); a[0] = 1.f; for (int i = 1; i < 32; ++i) a[i] = 0.125f * u; run("C=0, products 1 and 31 x 1/8 ulp", a, b, 0.f);
    clear(); a[0] = 1024.f; a[1] = -1024.f; a[2] = 0.3f * u; run("C=0, 1024 - 1024 + 0.3 ulp(1)", a, b, 0.f);
    clear(); a[0] = 1024.f; a[2] = 0.3f * u * 1024.f * 0.001f; run("C=-1024, +1024 + tiny", a, b, -1024.f);
    clear(); for (int i = 0; i < 32; ++i) { a[i] = 1.0009765625f; b[i] = 1.0009765625f; } run("32 products (1+2^-10)^2, C = 0", a, b, 0.f);
    // chained: bias of many accumulations
    for (int bf = 0; bf < 2; ++bf) {
        for (int i = 0; i < 32; ++i) { a[i] = 0.011f * (1 + (i % 5)); b[i] = 0.013f * (1 + (i % 3)); }
        hipMemcpy(da, a, 128, hipMemcpyHostToDevice); hipMemcpy(db, b, 128, hipMemcpyHostToDevice);
        for (int n : {64, 512}) {
            double ex = 0, step = 0;
            for (int i = 0; i < 32; ++i) {
                float ar = bf ? (float)(__bf16)a[i] : (float)(_Float16)a[i], br = bf ? (float)(__bf16)b[i] : (float)(_Float16)b[i];
                step += (double)ar * br;
            }
            ex = step * n;
            float acc32 = 0.f; for (int i = 0; i < n; ++i) acc32 += (float)step;        // fp32 RNE chain of exactly-summed steps
            float r;
            if (bf) hipLaunchKernelGGL(kchain<true>, dim3(1), dim3(64), 0, 0, da, db, 0.f, n, dout);
            else hipLaunchKernelGGL(kchain<false>, dim3(1), dim3(64), 0, 0, da, db, 0.f, n, dout);
            hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
            printf("chain %s n=%d: exact %.9e  fp32-RNE chain %.9e (rel %+.2e)  mfma %.9e (rel %+.2e)\n", bf ? "bf16" : "f16", n, ex, acc32,
                   (acc32 - ex) / ex, r, (r - ex) / ex);
        }
    }
    return 0;
}
