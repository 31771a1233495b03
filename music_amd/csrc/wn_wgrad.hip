// Weight-gradient products: C[m][n] += sum_{b,t} A[m][t + a_shift] * B_tap[n][t + b_shift_tap]
// (dWd = sum dy z^T, dWf0 = sum df x(t-d)^T, dWs = sum du z^T, ...).
//
// Both operands are channels-first rows with time contiguous, so TIME is the MFMA k index and
// every fragment is 8 consecutive samples of one row (two float4 loads, no transposition):
//   A fragment: lane (c,q) holds A[16m + c][tb + 8q + j]
//   B fragment: lane (c,q) holds B[16n + c][tb + 8q + j]        (C = A * B^T)
// One wave owns a 64x64 block of C and walks a chunk of the time axis; partial sums are added
// into the fp32 result with global float atomics (one 64-B segment per 16 lanes).
// Gradients are split in bf16 (fp32 exponent range); see wn_common.h for the x3 scheme.
#include "wn_common.h"
#include "wn_kernels.h"

template <class T, int NS>
__device__ __forceinline__ void wg_frag(Frag<T>& f, const float* row, int col, int ncols, int t,
                                        int t_lo, int t_hi, bool masked, bool relu) {
    f32x4 u0 = ld4g(row + col, col, 0, ncols);
    f32x4 u1 = ld4g(row + col + 4, col + 4, 0, ncols);
    float v[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
    if (masked) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (t + j < t_lo || t + j >= t_hi) v[j] = 0.f;
    }
    if (relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    split8<T, NS>(f, v);
}

template <class T, int NS>
__global__ __launch_bounds__(256) void wgrad_k(WnWgradArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int b = blockIdx.z;
    const int nt_total = a.nt_per_tap * (a.b1 ? 2 : 1);
    const int nblk_n = (nt_total + 3) / 4, nblk_m = (a.mt + 3) / 4;
    const int blk = blockIdx.y * 4 + wave;
    if (blk >= nblk_n * nblk_m) return;
    const int mb = blk / nblk_n, nb = blk % nblk_n;
    const int tc0 = a.t_base + blockIdx.x * a.chunk;
    int tc1 = tc0 + a.chunk;
    if (tc1 > a.t_hi) tc1 = a.t_hi;
    if (tc0 >= a.t_hi) return;

    const float* A = a.a + (size_t)b * a.a_bstride;
    const float* arow[4];
    const float* brow[4];
    int bshift[4];
    bool mval[4], nval[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        int mt = mb * 4 + m;
        mval[m] = mt < a.mt;
        arow[m] = A + (size_t)((mval[m] ? mt : 0) * 16 + c) * a.a_pitch;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        int nt = nb * 4 + n;
        nval[n] = nt < nt_total;
        if (!nval[n]) nt = 0;
        int tap = nt / a.nt_per_tap, r = (nt % a.nt_per_tap) * 16 + c;
        const float* base = (tap == 0 ? a.b0 : a.b1) + (size_t)b * a.b_bstride;
        brow[n] = base + (size_t)r * a.b_pitch;
        bshift[n] = tap == 0 ? a.b_shift0 : a.b_shift1;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tb = tc0; tb < tc1; tb += 32) {
        const bool masked = (tb < a.t_lo) || (tb + 32 > tc1);
        const int t = tb + 8 * q;
        Frag<T> af[4], bf[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            wg_frag<T, NS>(af[m], arow[m], t + a.a_shift, a.a_cols, t, a.t_lo, tc1, masked, false);
#pragma unroll
        for (int n = 0; n < 4; ++n)
            wg_frag<T, NS>(bf[n], brow[n], t + bshift[n], a.b_cols, t, a.t_lo, tc1, masked, a.relu_b != 0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) mma<T, NS>(acc[m][n], af[m], bf[n]);
    }

#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (!mval[m]) continue;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (!nval[n]) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = (mb * 4 + m) * 16 + 4 * q + i;
                int col = (nb * 4 + n) * 16 + c;
                atomicAdd(a.c + (size_t)row * a.ldc + col, acc[m][n][i]);
            }
        }
    }
}

int wn_launch_wgrad(const WnWgradArgs& a, int batch, int mode, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    WnWgradArgs k = a;
    k.t_base = a.t_lo & ~31;
    if (k.chunk < 32) k.chunk = 32;
    k.chunk = (k.chunk + 31) & ~31;
    const int nt_total = a.nt_per_tap * (a.b1 ? 2 : 1);
    const int nblk = ((nt_total + 3) / 4) * ((a.mt + 3) / 4);
    dim3 g((a.t_hi - k.t_base + k.chunk - 1) / k.chunk, (nblk + 3) / 4, batch), b(256);
    switch (mode) {
        case WN_MODE_BF16X3: hipLaunchKernelGGL((wgrad_k<BF16, 3>), g, b, 0, st, k); break;
        case WN_MODE_BF16X1: hipLaunchKernelGGL((wgrad_k<BF16, 1>), g, b, 0, st, k); break;
        case WN_MODE_F16X3: hipLaunchKernelGGL((wgrad_k<F16, 3>), g, b, 0, st, k); break;
        case WN_MODE_F16X1: hipLaunchKernelGGL((wgrad_k<F16, 1>), g, b, 0, st, k); break;
        default: return wn_set_error_msg(-2, "wgrad: bad mode");
    }
    WN_CHECK_LAUNCH();
    return 0;
}
