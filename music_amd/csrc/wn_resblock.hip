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

template <class T, int NS, int CH>
__global__ __launch_bounds__(WN_RES_THREADS) void resblock_fwd_k(WnResArgs a) {
    constexpr int MT = 2 * CH / 16;        // fg row tiles (f rows then g rows)
    constexpr int KS = 2 * CH / 32;        // fg k-steps (tap 0 channels then tap 1 channels)
    constexpr int KT = CH / 32;            // k-steps per tap
    constexpr int MT2 = CH / 16;           // dense row tiles
    constexpr int KS2 = CH / 32;           // dense k-steps
    constexpr int FR = (NS == 3 ? 1024 : 512);          // halfs per packed fragment
    constexpr int NFG = MT * KS, ND = MT2 * KS2;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* l_fg = lds;
    uint16_t* l_d = lds + (size_t)NFG * FR;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const WnBlock blk = wn_block(a.swz);
    const int b = blk.y;
    const int t0 = a.t_base + blk.x * WN_RES_COLS + wave * 64;
    const int tl = t0 + 4 * c;

    const float* xin = a.x_in + (size_t)b * a.x_bstride;
    // tap-0 column.  Lanes that own at least one valid output have tl - d >= -2 (t_lo >= d + 1);
    // every activation buffer is allocated with >= 64 floats of slack in front and >= 256 behind,
    // so the (masked-out) garbage columns are still addressable.
    const int colm = tl - a.d;

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
    issue(0);      // first activation loads are in flight while the weights are staged

    {   // stage the packed weights (contiguous copies, 16 B per thread per step)
        const u32x4* s0 = reinterpret_cast<const u32x4*>(a.wfg);
        u32x4* d0 = reinterpret_cast<u32x4*>(l_fg);
        for (int i = threadIdx.x; i < NFG * FR / 8; i += WN_RES_THREADS) d0[i] = s0[i];
        const u32x4* s1 = reinterpret_cast<const u32x4*>(a.wd);
        u32x4* d1 = reinterpret_cast<u32x4*>(l_d);
        for (int i = threadIdx.x; i < ND * FR / 8; i += WN_RES_THREADS) d1[i] = s1[i];
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
        Frag<T> bf[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = raw[j][n];
            split8<T, NS>(bf[n], v);
        }
        if (s + 1 < KS) issue(s + 1);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            Frag<T> af;
            load_a<T, NS>(af, l_fg, m * KS + s, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n) mma<T, NS>(acc[m][n], af, bf[n]);
        }
    }

    if (a.cond) {       // per-(channel, time-bucket) conditioning bias, gathered from a tiny table
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

    // residual rows in C layout (row 16m+4q+i, columns tl..tl+3): issue early, used at the end
    f32x4 res[MT2][4];
    if (a.write_x) {
#pragma unroll
        for (int m = 0; m < MT2; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                res[m][i] = ld4(xin + (size_t)(16 * m + 4 * q + i) * a.pitch + tl);
    }

    // gate: z tile m = tanh(f tile m) * sigmoid(g tile m)
    f32x4 z[MT2][4];
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                z[m][n][i] = wn_tanh(acc[m][n][i]) * wn_sigmoid(acc[m + MT2][n][i]);

    // z-crop store (rows 16m+4q+i; the lane's 4 N-tiles are 4 consecutive samples)
    {
        float* zo = a.z_out + (size_t)b * a.z_bstride;
#pragma unroll
        for (int m = 0; m < MT2; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = {z[m][0][i], z[m][1][i], z[m][2][i], z[m][3][i]};
                st4m(zo + (size_t)(16 * m + 4 * q + i) * a.pitch + tl, v, tl, a.z_lo, a.t_hi);
            }
    }
    if (!a.write_x) return;

    // dense: x' = Wd z + x   (B fragments straight from the z accumulators, chained k order)
    f32x4 acc2[MT2][4];
#pragma unroll
    for (int m = 0; m < MT2; ++m) {
        f32x4 init = {0.f, 0.f, 0.f, 0.f};
        if (a.bias_d) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = m * 16 + 4 * q + i;
                init[i] = row < a.n_d ? a.bias_d[row] : 0.f;
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc2[m][n] = init;
    }
#pragma unroll
    for (int s = 0; s < KS2; ++s) {
        Frag<T> bf[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = z[2 * s][n][i]; v[4 + i] = z[2 * s + 1][n][i]; }
            split8<T, NS>(bf[n], v);
        }
#pragma unroll
        for (int m = 0; m < MT2; ++m) {
            Frag<T> af;
            load_a<T, NS>(af, l_d, m * KS2 + s, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n) mma<T, NS>(acc2[m][n], af, bf[n]);
        }
    }
    float* xo = a.x_out + (size_t)b * a.x_bstride;
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = {acc2[m][0][i] + res[m][i][0], acc2[m][1][i] + res[m][i][1],
                       acc2[m][2][i] + res[m][i][2], acc2[m][3][i] + res[m][i][3]};
            st4m(xo + (size_t)(16 * m + 4 * q + i) * a.pitch + tl, v, tl, a.t_lo, a.t_hi);
        }
}

template <class T, int NS>
static int launch_fwd(const WnResArgs& a, int ch, int batch, hipStream_t st) {
    WnResArgs k = a;
    k.swz = wn_xcd_swizzle_enabled();
    k.t_base = wn_tile_origin(a.t_lo);
    int ncol = a.t_hi - k.t_base;
    dim3 g((ncol + WN_RES_COLS - 1) / WN_RES_COLS, batch), b(WN_RES_THREADS);
    const size_t fr = (NS == 3 ? 1024 : 512) * sizeof(uint16_t);
    if (ch == 32) {
        size_t sh = (size_t)(4 * 2 + 2 * 1) * fr;
        hipLaunchKernelGGL((resblock_fwd_k<T, NS, 32>), g, b, sh, st, k);
    } else if (ch == 64) {
        size_t sh = (size_t)(8 * 4 + 4 * 2) * fr;
        static unsigned long long done = 0;       // per-device bit
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!((done >> dev) & 1ull)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_fwd_k<T, NS, 64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            done |= 1ull << dev;
        }
        hipLaunchKernelGGL((resblock_fwd_k<T, NS, 64>), g, b, sh, st, k);
    } else {
        return wn_set_error_msg(-3, "resblock: padded channel count must be 32 or 64");
    }
    WN_CHECK_LAUNCH();
    return 0;
}

int wn_launch_resblock_fwd(const WnResArgs& a, int ch, int batch, int mode, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    // default: the NT-templated kernel of wn_resblock2.hip with 4 N-tiles per wave (26.6 us per
    // config-2 layer vs 29.2 for resblock_fwd_k below; NT = 2 measured 33 us).  WN_FWD_NT=0|2|4 overrides.
    // WN_FWD_CS=1: the channel-split form (wn_resblock3.hip; 64 padded channels at f16x3).  Correct (the
    // whole GPU suite passes with it) but 8-10 % SLOWER than the kernel below at config 2 (28.8 vs
    // 26.2 us per block): the forward is bound by its HBM share (390 KB per CU and launch), not by the
    // length of a wave's dependency chain, so shortening the chain buys nothing.  Off by default.
    // WN_FWD_RW (default 1): the two-role persistent kernel of wn_resfwd_rw.hip takes 64-channel x3 launches
    static int cs = -1;
    if (cs < 0) { const char* e = getenv("WN_FWD_CS"); cs = e ? atoi(e) : 0; }
    if (!cs && wn_launch_resblock_fwd_rw(a, ch, batch, mode, st)) {
        WN_CHECK_LAUNCH();
        return 0;
    }
    if (cs && ch == 64 && mode == WN_MODE_F16X3) return wn_launch_resblock_fwd_cs(a, batch, st);
    static int nt = -1;
    if (nt < 0) { const char* e = getenv("WN_FWD_NT"); nt = e ? atoi(e) : 4; }
    if (nt == 2 || nt == 4) return wn_launch_resblock_fwd_nt(a, ch, batch, mode, nt, st);
    switch (mode) {
        case WN_MODE_F16X3: return launch_fwd<F16, 3>(a, ch, batch, st);
        case WN_MODE_F16X1: return launch_fwd<F16, 1>(a, ch, batch, st);
        case WN_MODE_BF16X3: return launch_fwd<BF16, 3>(a, ch, batch, st);
        case WN_MODE_BF16X1: return launch_fwd<BF16, 1>(a, ch, batch, st);
    }
    return wn_set_error_msg(-2, "resblock_fwd: bad mode");
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
                float th = wn_tanh(acc[m][n][i]);
                float sg = wn_sigmoid(acc[m + MT2][n][i]);
                zz[n] = th * sg;
                df[n] = g[n] * sg * (1.0f - th * th);
                dg[n] = g[n] * th * sg * (1.0f - sg);
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
        static unsigned long long done = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!((done >> dev) & 1ull)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_bwd_k<TF, NSF, TB, NSB, 64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            done |= 1ull << dev;
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
