// check: two-column MFMA product + DPP pair sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 pairsum(const f32x4& v) {
    // (written as asm: through __builtin_amdgcn_update_dpp in an unrolled loop over the vector's elements this compiler
    // folded the four DPP reads into the first one - checked with tools/micro/twocol.hip; s_nop covers the
    // VALU-write -> DPP-read hazard the assembler does not see inside an asm block)
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float o;
        const float x = v[i];
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(x));
        r[i] = o;
    }
    return r;
}
__global__ void k(const float* W, const float* x, float* out, float* ps) {   // W [16][32], x [32]
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    f16x8 ahi, alo, b;
    for (int j = 0; j < 8; ++j) {
        float w = W[c * 32 + 8 * q + j];
        _Float16 h = (_Float16)w; ahi[j] = h; alo[j] = (_Float16)(w - (float)h);
        float xv = x[8 * q + j];
        _Float16 xh = (_Float16)xv, xl = (_Float16)(xv - (float)xh);
        b[j] = (c & 1) ? xl : xh;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, b, acc, 0, 0, 0);
    if (x[0] > 100.f) acc[0] += 1.f;
    for (int i = 0; i < 4; ++i) ps[64 + lane * 4 + i] = acc[i];
    f32x4 r = pairsum(acc);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = r[i];
    f32x4 t = {(float)lane, 0.f, 0.f, 0.f};
    ps[lane] = pairsum(t)[0];
}
int main() {
    float hW[512], hx[32], *W, *x, *out, *ps, ho[256], hp[64];
    for (int i = 0; i < 512; ++i) hW[i] = sinf(i * 0.37f) * 1.3f;
    for (int i = 0; i < 32; ++i) hx[i] = cosf(i * 0.91f) * 2.1f;
    hipMalloc(&W, 2048); hipMalloc(&x, 128); hipMalloc(&out, 1024); hipMalloc(&ps, 256 + 1024);
    hipMemcpy(W, hW, 2048, hipMemcpyHostToDevice); hipMemcpy(x, hx, 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, W, x, out, ps);
    hipMemcpy(ho, out, 1024, hipMemcpyDeviceToHost); hipMemcpy(hp, ps, 256, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 4; ++i) {
        int row = 4 * (lane >> 4) + i; double ref = 0; for (int kk = 0; kk < 32; ++kk) ref += (double)hW[row * 32 + kk] * hx[kk];
        double e = fabs(ref - ho[lane * 4 + i]); if (e > worst) worst = e;
    }
    printf("two-column product: worst |err| over all lanes %.3g\n", worst);
    float raw[256]; hipMemcpy(raw, ps + 64, 1024, hipMemcpyDeviceToHost);
    for (int row = 0; row < 2; ++row) {
        double rh = 0, rl = 0, rf = 0;
        for (int kk = 0; kk < 32; ++kk) { float xv = hx[kk]; _Float16 xh = (_Float16)xv; float xl = (float)(_Float16)(xv - (float)xh); rh += (double)hW[row * 32 + kk] * (float)xh; rl += (double)hW[row * 32 + kk] * xl; rf += (double)hW[row * 32 + kk] * xv; }
        printf("row %d: ref W.xh %.6f  W.xl %.6g  full %.6f | lanes c=0..3 (q=0, i=%d): %.6f %.6g %.6f %.6g | summed: %.6f\n", row, rh, rl, rf, row, raw[0 * 4 + row], raw[1 * 4 + row], raw[2 * 4 + row], raw[3 * 4 + row], ho[0 * 4 + row]);
    }
    printf("pairsum(lane): "); for (int i = 0; i < 8; ++i) printf("%g ", hp[i]); printf("\n");
    return 0;
}
