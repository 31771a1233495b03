// Generic channel-mixing GEMM over the time axis (time on the MFMA lanes) and the weight packer.
//
//   out[m][t + out_shift] = epi( bias[m] + sum_tap sum_k W[m][tap,k] * pre(in_tap[k][t + shift_tap]) )
//
// Used for: causal conv (wavenet/model.py:104, two taps of K=Q), the skip product over the
// concatenated z-crops and post_process_1/2 (model.py:128-138), and every "weights transposed"
// data-gradient product of the backward pass.  The fused residual-block kernels live in
// wn_resblock.hip; this kernel is the unfused workhorse around them.
#include "wn_common.h"
#include "wn_kernels.h"

// ---------------------------------------------------------------------------------------------
// Weight packing: flat fp32 parameters -> fragment-ordered 16-bit hi/lo pairs.
// idx[p] (p in logical order [frag][lane][j]) is the offset of the source weight in the flat
// parameter buffer, or -1 for a structural zero (padding).  Built once on the host.
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void pack_weights_k(const float* __restrict__ flat, const int32_t* __restrict__ idx,
                               uint16_t* __restrict__ out, int n, int ns) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int src = idx[p];
    float v = src >= 0 ? flat[src] : 0.0f;
    typename T::elem h = T::cvt(v);
    int frag = p >> 9, r = p & 511;
    size_t o = (size_t)frag * (ns == 3 ? 1024 : 512) + r;
    out[o] = __builtin_bit_cast(uint16_t, h);
    if (ns == 3) {
        typename T::elem l = T::cvt(v - T::back(h));
        out[o + 512] = __builtin_bit_cast(uint16_t, l);
    }
}

int wn_launch_pack(const float* flat, const int32_t* idx, uint16_t* out, int n, int is_bf16, int ns,
                   hipStream_t st) {
    if (n <= 0) return 0;
    dim3 g((n + 255) / 256), b(256);
    if (is_bf16) hipLaunchKernelGGL(pack_weights_k<BF16>, g, b, 0, st, flat, idx, out, n, ns);
    else hipLaunchKernelGGL(pack_weights_k<F16>, g, b, 0, st, flat, idx, out, n, ns);
    WN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// chan_gemm: WG = 4 waves; wave w owns columns [tile*256 + 64w, +64) and 4 M-tiles (64 rows) of
// M-block blockIdx.y; blockIdx.z = clip.
// ---------------------------------------------------------------------------------------------
template <class T, int NS>
__global__ __launch_bounds__(256) void chan_gemm_k(WnGemmArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int b = blockIdx.z;
    const int t0 = a.t_base + blockIdx.x * 256 + wave * 64;      // first column of this wave
    const int tl = t0 + 4 * c;                                    // this lane's first column
    if (t0 >= a.t_hi) return;
    const int m0 = blockIdx.y * 4;                                // first M-tile
    const int KS = a.ks0 + a.ks1;

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 init = {0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = (m0 + m) * 16 + 4 * q + i;
                init[i] = row < a.m_valid ? a.bias[row] : 0.f;
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = init;
    }

    const float* in0 = a.in0 + (size_t)b * a.in_bstride;
    const float* in1 = a.in1 ? a.in1 + (size_t)b * a.in_bstride : nullptr;
    const int col0 = tl + a.shift0, col1 = tl + a.shift1;

    f32x4 raw[8];
    auto issue = [&](int s) {
        const float* base; int col; int ch;
        if (s < a.ks0) { base = in0; col = col0; ch = s * 32; }
        else { base = in1; col = col1; ch = (s - a.ks0) * 32; }
        const float* p = base + (size_t)(ch + 8 * q) * a.in_pitch + col;
        // input columns outside [in_lo, in_hi) read as 0 and are never dereferenced
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = ld4g(p + (size_t)j * a.in_pitch, col, a.in_lo, a.in_hi);
    };
    issue(0);
    for (int s = 0; s < KS; ++s) {
        Frag<T> bf[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = raw[j][n];
                v[j] = a.relu_in ? fmaxf(x, 0.f) : x;
            }
            split8<T, NS>(bf[n], v);
        }
        if (s + 1 < KS) issue(s + 1);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m0 + m < a.mt) {
                Frag<T> af;
                load_a<T, NS>(af, a.wpack, (m0 + m) * KS + s, lane);
#pragma unroll
                for (int n = 0; n < 4; ++n) mma<T, NS>(acc[m][n], af, bf[n]);
            }
        }
    }

    float* out = a.out + (size_t)b * a.out_bstride;
    const float* resid = a.resid ? a.resid + (size_t)b * a.resid_bstride : nullptr;
    const float* mask = a.mask ? a.mask + (size_t)b * a.mask_bstride : nullptr;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m0 + m >= a.mt) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = (m0 + m) * 16 + 4 * q + i;
            if (row >= a.m_valid) continue;
            f32x4 v = {acc[m][0][i], acc[m][1][i], acc[m][2][i], acc[m][3][i]};
            if (resid) {
                const float* rp = resid + (size_t)row * a.resid_pitch + tl;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (tl + e >= a.resid_lo && tl + e >= a.t_lo && tl + e < a.t_hi) v[e] += rp[e];
            }
            if (mask) {
                const float* mp = mask + (size_t)row * a.mask_pitch + tl;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (tl + e >= a.t_lo && tl + e < a.t_hi) v[e] = mp[e] > 0.f ? v[e] : 0.f;
            }
            float* op = out + (size_t)row * a.out_pitch + tl + a.out_shift;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (tl + e >= a.t_lo && tl + e < a.t_hi) op[e] = v[e];
        }
    }
}

int wn_launch_gemm(const WnGemmArgs& a, int batch, int mode, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    WnGemmArgs k = a;
    k.t_base = a.t_lo & ~3;                 // lanes own 4 consecutive, 4-aligned columns
    int ncol = a.t_hi - k.t_base;
    dim3 g((ncol + 255) / 256, (a.mt + 3) / 4, batch), b(256);
    switch (mode) {
        case WN_MODE_F16X3: hipLaunchKernelGGL((chan_gemm_k<F16, 3>), g, b, 0, st, k); break;
        case WN_MODE_F16X1: hipLaunchKernelGGL((chan_gemm_k<F16, 1>), g, b, 0, st, k); break;
        case WN_MODE_BF16X3: hipLaunchKernelGGL((chan_gemm_k<BF16, 3>), g, b, 0, st, k); break;
        case WN_MODE_BF16X1: hipLaunchKernelGGL((chan_gemm_k<BF16, 1>), g, b, 0, st, k); break;
        default: return wn_set_error_msg(-2, "wn_launch_gemm: bad mode");
    }
    WN_CHECK_LAUNCH();
    return 0;
}
