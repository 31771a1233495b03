// Fused gated residual block of the WaveNet stack (wavenet/model.py:108-129), forward and the
// recompute half of the backward.
//
// forward, per clip and per output sample t (absolute time), CH = padded channel count:
//   [f;g] = Wfg[:, 0:CH] x[:, t-d] + Wfg[:, CH:2CH] x[:, t]   (+bias)     one (2CH x 2CH) product
//   z     = tanh(f) * sigmoid(g)                                           in registers
//   x'    = Wd z + x[:, t]                                   (+bias)       chained product (z never
//                                                                          leaves the accumulators)
//   z is stored only on the crop t >= z_lo that the skip product needs.
// The packed weights (hi/lo fragments) of the block are staged once per workgroup in LDS; the
// activations stream HBM -> registers as float4 (time on the lanes, see wn_common.h).
#include <stdlib.h>
#include "wn_common.h"
#include "wn_kernels.h"

// One workgroup = 8 waves = 512 time columns.  The packed weights of a block (80 KB in the x3
// modes) are staged once per workgroup; with 4-wave groups only ONE group fits a CU (2 x 80 KB is
// exactly the 160 KB of LDS) and the 472 groups of a config-2 layer ran as two rounds.
#define WN_RES_THREADS 512
#define WN_RES_COLS 512

int wn_launch_resblock_fwd(const WnResArgs& a, int ch, int batch, int mode, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    return wn_launch_resblock_fwd_nt(a, ch, batch, mode, st);      // wn_resblock2.hip
}

// ---------------------------------------------------------------------------------------------
// Backward, recompute half:
//   recompute f,g from x_i (same arithmetic as the forward), th = tanh f, sg = sigmoid g, z = th*sg
//   dz = Wd^T dy  (+ dz_crop for t >= z_lo)
//   df = dz * sg * (1 - th^2)        dg = dz * th * sg * (1 - sg)
//   stores [df; dg] (2CH rows) and z (CH rows); the data gradient dx_i (a two-tap product over
//   [df;dg]) and all weight gradients are separate launches (chan_gemm / wgrad).
// TF/NSF = forward operand type (recompute must match the forward bit for bit),
// TB/NSB = gradient operand type (bf16: gradients need fp32's exponent range).
// ---------------------------------------------------------------------------------------------
template <class TF, int NSF, class TB, int NSB, int CH>
__global__ __launch_bounds__(WN_RES_THREADS) void resblock_bwd_k(WnResBwdArgs a) {
    constexpr int MT = 2 * CH / 16, KS = 2 * CH / 32, KT = CH / 32, MT2 = CH / 16, KS2 = CH / 32;
    constexpr int FRF = (NSF == 3 ? 1024 : 512), FRB = (NSB == 3 ? 1024 : 512);
    constexpr int NFG = MT * KS, ND = MT2 * KS2;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* l_fg = lds;
    uint16_t* l_dt = lds + (size_t)NFG * FRF;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const WnBlock blk = wn_block(a.swz);
    const int b = blk.y;
    const int t0 = a.t_base + blk.x * WN_RES_COLS + wave * 64;
    const int tl = t0 + 4 * c;
    const float* xin = a.x_in + (size_t)b * a.x_bstride;
    const int colm = tl - a.d;          // see resblock_fwd_k
    f32x4 raw[8];
    auto issue = [&](int s) {
        const int tap = s / KT, ch = (s % KT) * 32 + 8 * q;
        const float* p = xin + (size_t)ch * a.pitch + (tap == 0 ? colm : tl);
        if (tap == 0) {        // always the alignment-free form for the shifted tap (no run-time merge of loaded registers)
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = ld4u(p + (size_t)j * a.pitch);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = ld4(p + (size_t)j * a.pitch);
        }
    };
    issue(0);
    {
        const u32x4* s0 = reinterpret_cast<const u32x4*>(a.wfg);
        u32x4* d0 = reinterpret_cast<u32x4*>(l_fg);
        for (int i = threadIdx.x; i < NFG * FRF / 8; i += WN_RES_THREADS) d0[i] = s0[i];
        const u32x4* s1 = reinterpret_cast<const u32x4*>(a.wdT);
        u32x4* d1 = reinterpret_cast<u32x4*>(l_dt);
        for (int i = threadIdx.x; i < ND * FRB / 8; i += WN_RES_THREADS) d1[i] = s1[i];
    }
    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x4 init = {0.f, 0.f, 0.f, 0.f};
        const float* bp = m < MT2 ? a.bias_f : a.bias_g;
        if (bp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = (m % MT2) * 16 + 4 * q + i;
                init[i] = row < a.n_f ? bp[row] : 0.f;
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = init;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        Frag<TF> bf[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = raw[j][n];
            split8<TF, NSF>(bf[n], v);
        }
        if (s + 1 < KS) issue(s + 1);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            Frag<TF> af;
            load_a<TF, NSF>(af, l_fg, m * KS + s, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n) mma<TF, NSF>(acc[m][n], af, bf[n]);
        }
    }

    if (a.cond) {       // same conditioning bias as the forward (wavenet_autoencoder/model1.py:183)
        const float* cb = a.cond + (size_t)b * a.cond_bstride;
        int idx[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            int tr = tl + n - a.t_lo;
            tr = tr < 0 ? 0 : tr;
            int ix = a.cond_mode == 1 ? tr / a.cond_q : tr % a.cond_le;
            idx[n] = ix < a.cond_le ? ix : a.cond_le - 1;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* cr = cb + (size_t)(16 * m + 4 * q + i) * a.cond_pitch;
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n][i] += cr[idx[n]];
            }
    }
    // dz = Wd^T dy : M = CH (dilation channels), K = CH (residual channels), B = dy rows (natural k)
    f32x4 dz[MT2][4];
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) dz[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.dy) {
        const float* dy = a.dy + (size_t)b * a.x_bstride;
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
            const float* p = dy + (size_t)(32 * s + 8 * q) * a.pitch + tl;
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = ld4(p + (size_t)j * a.pitch);
            Frag<TB> bf[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // columns outside [t_lo, t_hi) hold no gradient
                    int t = tl + n;
                    v[j] = (t >= a.t_lo && t < a.t_hi) ? raw[j][n] : 0.f;
                }
                split8<TB, NSB>(bf[n], v);
            }
#pragma unroll
            for (int m = 0; m < MT2; ++m) {
                Frag<TB> af;
                load_a<TB, NSB>(af, l_dt, m * KS2 + s, lane);
#pragma unroll
                for (int n = 0; n < 4; ++n) mma<TB, NSB>(dz[m][n], af, bf[n]);
            }
        }
    }
    const float* dzc = a.dz + (size_t)b * a.dz_bstride;
    float* dfg = a.dfg + (size_t)b * a.dfg_bstride;
    float* zo = a.z ? a.z + (size_t)b * a.z_bstride : nullptr;   // null: the caller keeps the forward's z
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * m + 4 * q + i;
            f32x4 g = {dz[m][0][i], dz[m][1][i], dz[m][2][i], dz[m][3][i]};
            if (tl + 3 >= a.z_lo) {
                f32x4 cr = ld4(dzc + (size_t)row * a.pitch + tl);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (tl + e >= a.z_lo && tl + e < a.t_hi) g[e] += cr[e];
            }
            f32x4 df, dg, zz;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const WnGateD gd = wn_gate_d(acc[m][n][i], acc[m + MT2][n][i]);
                zz[n] = gd.z;
                df[n] = g[n] * gd.dzdf;
                dg[n] = g[n] * gd.dzdg;
            }
            st4m(dfg + (size_t)row * a.pitch + tl, df, tl, a.t_lo, a.t_hi);
            st4m(dfg + (size_t)(CH + row) * a.pitch + tl, dg, tl, a.t_lo, a.t_hi);
            if (a.z) st4m(zo + (size_t)row * a.pitch + tl, zz, tl, a.t_lo, a.t_hi);
        }
}

template <class TF, int NSF, class TB, int NSB>
static int launch_bwd(const WnResBwdArgs& a, int ch, int batch, hipStream_t st) {
    WnResBwdArgs k = a;
    k.swz = wn_xcd_swizzle_enabled();
    k.t_base = wn_tile_origin(a.t_lo);
    int ncol = a.t_hi - k.t_base;
    dim3 g((ncol + WN_RES_COLS - 1) / WN_RES_COLS, batch), b(WN_RES_THREADS);
    const size_t frf = (NSF == 3 ? 1024 : 512) * 2, frb = (NSB == 3 ? 1024 : 512) * 2;
    if (ch == 32) {
        size_t sh = 8 * frf + 2 * frb;
        hipLaunchKernelGGL((resblock_bwd_k<TF, NSF, TB, NSB, 32>), g, b, sh, st, k);
    } else if (ch == 64) {
        size_t sh = 32 * frf + 8 * frb;
        static WnDevOnce done;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (done.need(dev)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_bwd_k<TF, NSF, TB, NSB, 64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            done.done(dev);
        }
        hipLaunchKernelGGL((resblock_bwd_k<TF, NSF, TB, NSB, 64>), g, b, sh, st, k);
    } else {
        return wn_set_error_msg(-3, "resblock: padded channel count must be 32 or 64");
    }
    WN_CHECK_LAUNCH();
    return 0;
}

int wn_launch_resblock_bwd(const WnResBwdArgs& a, int ch, int batch, int mode_fwd, int mode_bwd,
                           hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    // supported pairs: (F16X3,BF16X3) parity grade, (F16X1,BF16X1) fast, (BF16X3,BF16X3), (BF16X1,BF16X1)
    if (mode_fwd == WN_MODE_F16X3 && mode_bwd == WN_MODE_BF16X3) return launch_bwd<F16, 3, BF16, 3>(a, ch, batch, st);
    if (mode_fwd == WN_MODE_F16X1 && mode_bwd == WN_MODE_BF16X1) return launch_bwd<F16, 1, BF16, 1>(a, ch, batch, st);
    if (mode_fwd == WN_MODE_BF16X3 && mode_bwd == WN_MODE_BF16X3) return launch_bwd<BF16, 3, BF16, 3>(a, ch, batch, st);
    if (mode_fwd == WN_MODE_BF16X1 && mode_bwd == WN_MODE_BF16X1) return launch_bwd<BF16, 1, BF16, 1>(a, ch, batch, st);
    return wn_set_error_msg(-2, "resblock_bwd: unsupported mode pair");
}
