// Fully fused backward of one gated residual block: one launch per layer does
//   1. recompute f, g from x_i (identical arithmetic to the forward), th = tanh f, sg = sigmoid g
//   2. dy = dx_{i+1} (given as the pair P_{i+1}[t] + Q_{i+1}[t + d_{i+1}], see below),
//      dz = Wd^T dy + dz_crop,  df = dz sg (1 - th^2),  dg = dz th sg (1 - sg),  z = th sg
//   3. data gradient as two UNSHIFTED products  P[t] = W1^T [df;dg][t] + dy[t],  Q[t] = W0^T [df;dg][t]
//      so that dx_i[t] = P[t] + Q[t + d]: the shift-add is left to the consumer (the next launch
//      reads dy that way), which removes the halo / second pass a shifted product would need
//   4. both weight-gradient products of the block over the workgroup's 512 columns,
//      dWfg = sum [df;dg] [x(t-d) | x(t)]^T and dWd = sum dy z^T, written as one slab per workgroup.
// [df;dg], z and dy never go to HBM as tensors: each workgroup parks its 512-column tile in a
// private scratch tile that it re-reads at once with L1-bypassing loads (the tile is still in
// L2), first in "channel on k" fragment order for step 3, then in "time on k" order for step 4.
// One workgroup = 8 waves = 512 columns; LDS holds the three packed weight sets (144 KB).
#include "wn_common.h"
#include "wn_kernels.h"

#define RF_THREADS 512
#define RF_COLS 512

__device__ __forceinline__ f32x4 ld4nt(const float* p) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    return __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
}

// dy[t..t+3] of one channel row = P[t] (t >= p_lo) + Q[t + dn] (t + dn < t_hi)
__device__ __forceinline__ f32x4 load_dy(const float* prow, const float* qrow, int t, int dn, int p_lo, int t_hi) {
    f32x4 p = ld4(prow + t);
    f32x4 qv = ld4u(qrow + t + dn);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        r[e] = ((t + e >= p_lo) ? p[e] : 0.f) + ((t + e + dn < t_hi) ? qv[e] : 0.f);
    return r;
}

template <class TF, int NSF, class TB, int NSB, int CH>
__global__ __launch_bounds__(RF_THREADS) void resblock_bwd_fused_k(WnResFusedArgs a) {
    constexpr int MT = 2 * CH / 16, KS = 2 * CH / 32, KT = CH / 32, MT2 = CH / 16, KS2 = CH / 32;
    constexpr int FRF = (NSF == 3 ? 1024 : 512), FRB = (NSB == 3 ? 1024 : 512);
    constexpr int NFG = MT * KS, ND = MT2 * KS2;
    constexpr int PQ_MT = 2 * CH / 16;            // rows: P (CH) then Q (CH)
    constexpr int PQ_KS = 2 * CH / 32;            // k = [df | dg] channels
    constexpr int NPQ = PQ_MT * PQ_KS;
    constexpr int SROWS = 4 * CH;                 // scratch rows: dfg (2CH) | z (CH) | dy (CH)
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* l_fg = lds;
    uint16_t* l_dt = l_fg + (size_t)NFG * FRF;
    uint16_t* l_pq = l_dt + (size_t)ND * FRB;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int b = blockIdx.y;
    const int c0 = a.t_base + blockIdx.x * RF_COLS;           // first column of the workgroup
    const int t0 = c0 + wave * 64;
    const int tl = t0 + 4 * c;
    const int lc = wave * 64 + 4 * c;                          // local column inside the scratch tile
    float* scr = a.scratch + ((size_t)b * gridDim.x + blockIdx.x) * (size_t)SROWS * RF_COLS;

    const float* xin = a.x_in + (size_t)b * a.x_bstride;
    const bool aligned_d = (a.d & 3) == 0;
    const int colm = tl - a.d;
    f32x4 raw[8];
    auto issue = [&](int s) {
        const int tap = s / KT, ch = (s % KT) * 32 + 8 * q;
        const float* p = xin + (size_t)ch * a.pitch + (tap == 0 ? colm : tl);
        if (tap == 0 && !aligned_d) {
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
        for (int i = threadIdx.x; i < NFG * FRF / 8; i += RF_THREADS) d0[i] = s0[i];
        const u32x4* s1 = reinterpret_cast<const u32x4*>(a.wdT);
        u32x4* d1 = reinterpret_cast<u32x4*>(l_dt);
        for (int i = threadIdx.x; i < ND * FRB / 8; i += RF_THREADS) d1[i] = s1[i];
        const u32x4* s2 = reinterpret_cast<const u32x4*>(a.wpq);
        u32x4* d2 = reinterpret_cast<u32x4*>(l_pq);
        for (int i = threadIdx.x; i < NPQ * FRB / 8; i += RF_THREADS) d2[i] = s2[i];
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
    // ---------------- 1a. dz = Wd^T dy first (its 64 accumulators would not fit beside the f/g
    // accumulators): parked in the z rows of the scratch tile, re-read row by row in step 2
    const float* dP = a.dP_in ? a.dP_in + (size_t)b * a.x_bstride : nullptr;
    const float* dQ = a.dQ_in ? a.dQ_in + (size_t)b * a.x_bstride : nullptr;
    if (dP) {
        f32x4 dz[MT2][4];
#pragma unroll
        for (int m = 0; m < MT2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) dz[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
            f32x4 dyr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const size_t ro = (size_t)(32 * s + 8 * q + j) * a.pitch;
                dyr[j] = load_dy(dP + ro, dQ + ro, tl, a.dn, a.p_lo, a.t_hi);
            }
            Frag<TB> bf[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (tl + n >= a.t_lo && tl + n < a.t_hi) ? dyr[j][n] : 0.f;
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
#pragma unroll
        for (int m = 0; m < MT2; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = {dz[m][0][i], dz[m][1][i], dz[m][2][i], dz[m][3][i]};
                *reinterpret_cast<f32x4*>(scr + (size_t)(2 * CH + 16 * m + 4 * q + i) * RF_COLS + lc) = v;
            }
    }
    // ---------------- 1b. recompute f, g
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
    // ---------------- 2. gates; park [df;dg], z, dy in the scratch tile
    const float* dzc = a.dz + (size_t)b * a.dz_bstride;
#pragma unroll
    for (int m = 0; m < MT2; ++m) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * m + 4 * q + i;
            // dz of this row was parked in the (not yet written) z rows of the scratch tile
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (dP) g = ld4nt(scr + (size_t)(2 * CH + row) * RF_COLS + lc);
            if (tl + 3 >= a.z_lo) {
                f32x4 cr = ld4(dzc + (size_t)row * a.pitch + tl);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (tl + e >= a.z_lo && tl + e < a.t_hi) g[e] += cr[e];
            }
            f32x4 df, dg, zz, dyv = {0.f, 0.f, 0.f, 0.f};
            if (dP) dyv = load_dy(dP + (size_t)row * a.pitch, dQ + (size_t)row * a.pitch, tl, a.dn, a.p_lo, a.t_hi);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const bool ok = tl + n >= a.t_lo && tl + n < a.t_hi;
                float th = wn_tanh(acc[m][n][i]);
                float sg = wn_sigmoid(acc[m + MT2][n][i]);
                zz[n] = ok ? th * sg : 0.f;
                df[n] = ok ? g[n] * sg * (1.0f - th * th) : 0.f;
                dg[n] = ok ? g[n] * th * sg * (1.0f - sg) : 0.f;
                dyv[n] = ok ? dyv[n] : 0.f;
            }
            *reinterpret_cast<f32x4*>(scr + (size_t)row * RF_COLS + lc) = df;
            *reinterpret_cast<f32x4*>(scr + (size_t)(CH + row) * RF_COLS + lc) = dg;
            *reinterpret_cast<f32x4*>(scr + (size_t)(2 * CH + row) * RF_COLS + lc) = zz;
            *reinterpret_cast<f32x4*>(scr + (size_t)(3 * CH + row) * RF_COLS + lc) = dyv;
        }
    }
    __syncthreads();          // (also drains this wave's stores: the tile is complete in L2)

    // ---------------- 3. P = W1^T [df;dg] + dy,  Q = W0^T [df;dg]   (B fragments from the scratch tile)
    if (!(a.dbg_skip & 1)) {
        f32x4 pq[PQ_MT][4];
#pragma unroll
        for (int m = 0; m < PQ_MT; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) pq[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto issue3 = [&](int s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = ld4nt(scr + (size_t)(32 * s + 8 * q + j) * RF_COLS + lc);
        };
        issue3(0);
#pragma unroll
        for (int s = 0; s < PQ_KS; ++s) {
            Frag<TB> bf[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = raw[j][n];
                split8<TB, NSB>(bf[n], v);
            }
            if (s + 1 < PQ_KS) issue3(s + 1);
#pragma unroll
            for (int m = 0; m < PQ_MT; ++m) {
                Frag<TB> af;
                load_a<TB, NSB>(af, l_pq, m * PQ_KS + s, lane);
#pragma unroll
                for (int n = 0; n < 4; ++n) mma<TB, NSB>(pq[m][n], af, bf[n]);
            }
        }
        float* po = a.dP_out + (size_t)b * a.x_bstride;
        float* qo = a.dQ_out + (size_t)b * a.x_bstride;
#pragma unroll
        for (int m = 0; m < MT2; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * m + 4 * q + i;
                f32x4 dyv = ld4nt(scr + (size_t)(3 * CH + row) * RF_COLS + lc);
                f32x4 pv = {pq[m][0][i] + dyv[0], pq[m][1][i] + dyv[1], pq[m][2][i] + dyv[2], pq[m][3][i] + dyv[3]};
                f32x4 qv = {pq[m + MT2][0][i], pq[m + MT2][1][i], pq[m + MT2][2][i], pq[m + MT2][3][i]};
                st4m(po + (size_t)row * a.pitch + tl, pv, tl, a.t_lo, a.t_hi);
                st4m(qo + (size_t)row * a.pitch + tl, qv, tl, a.t_lo, a.t_hi);
            }
    }

    // ---------------- 4. weight gradients over the 512 columns of the tile (time on k)
    if (!(a.dbg_skip & 2)) {
        constexpr int MW = (2 * CH / 16) / 2;          // dWfg: M-tiles per wave (wave grid 2 x 4)
        constexpr int NW = (2 * CH / 16) / 4;          //       N-tiles per wave
        const int wm = wave >> 2, wn = wave & 3;
        f32x4 wacc[MW][NW];
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NW; ++n) wacc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // dWd: CH/16 x CH/16 tiles; CH 64: 2 M-tiles (wm) x 1 N-tile (wn); CH 32: waves 0..3 own one tile
        constexpr int DMW = CH == 64 ? 2 : 1;
        const bool d_on = a.has_d && (CH == 64 || wm == 0);
        const int d_m0 = CH == 64 ? 2 * wm : (wn >> 1), d_n = CH == 64 ? wn : (wn & 1);
        f32x4 dacc[DMW];
#pragma unroll
        for (int m = 0; m < DMW; ++m) dacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

        // raw operands of one k-step (32 samples); the next step's loads fly behind the MFMAs
        f32x4 ra[MW][2], rb[NW][2], rz[2], ry[DMW][2];
        auto issue4 = [&](int ks) {
            const int lt = ks * 32 + 8 * q;            // local time of this lane's 8 samples
            const int gt = c0 + lt;                    // absolute time
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const float* r = scr + (size_t)((wm * MW + m) * 16 + c) * RF_COLS + lt;
                ra[m][0] = ld4nt(r); ra[m][1] = ld4nt(r + 4);
            }
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const int nt = wn * NW + n;                        // column tile of [x(t-d) | x(t)]
                const int tap = nt / (CH / 16), r0 = (nt % (CH / 16)) * 16 + c;
                const float* r = xin + (size_t)r0 * a.pitch + gt + (tap == 0 ? -a.d : 0);
                rb[n][0] = ld4u(r); rb[n][1] = ld4u(r + 4);
            }
            if (d_on) {
                const float* r = scr + (size_t)(2 * CH + d_n * 16 + c) * RF_COLS + lt;
                rz[0] = ld4nt(r); rz[1] = ld4nt(r + 4);
#pragma unroll
                for (int m = 0; m < DMW; ++m) {
                    const float* ry_ = scr + (size_t)(3 * CH + (d_m0 + m) * 16 + c) * RF_COLS + lt;
                    ry[m][0] = ld4nt(ry_); ry[m][1] = ld4nt(ry_ + 4);
                }
            }
        };
        auto frag = [&](Frag<TB>& f, const f32x4* u) {
            float v[8] = {u[0][0], u[0][1], u[0][2], u[0][3], u[1][0], u[1][1], u[1][2], u[1][3]};
            split8<TB, NSB>(f, v);
        };
        issue4(0);
        for (int ks = 0; ks < RF_COLS / 32; ++ks) {
            Frag<TB> af[MW], bfr[NW], zf, yf[DMW];
#pragma unroll
            for (int m = 0; m < MW; ++m) frag(af[m], ra[m]);
#pragma unroll
            for (int n = 0; n < NW; ++n) frag(bfr[n], rb[n]);
            if (d_on) {
                frag(zf, rz);
#pragma unroll
                for (int m = 0; m < DMW; ++m) frag(yf[m], ry[m]);
            }
            if (ks + 1 < RF_COLS / 32) issue4(ks + 1);
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n) mma<TB, NSB>(wacc[m][n], af[m], bfr[n]);
            if (d_on) {
#pragma unroll
                for (int m = 0; m < DMW; ++m) mma<TB, NSB>(dacc[m], yf[m], zf);
            }
        }
        const size_t slab = (size_t)b * gridDim.x + blockIdx.x;
        float* sf = a.slab_fg + slab * (size_t)(4 * CH * CH);
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NW; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (wm * MW + m) * 16 + 4 * q + i, col = (wn * NW + n) * 16 + c;
                    sf[(size_t)row * (2 * CH) + col] = wacc[m][n][i];
                }
        if (d_on) {
            float* sd = a.slab_d + slab * (size_t)(CH * CH);
#pragma unroll
            for (int m = 0; m < DMW; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (d_m0 + m) * 16 + 4 * q + i, col = d_n * 16 + c;
                    sd[(size_t)row * CH + col] = dacc[m][i];
                }
        }
    }
}

template <class TF, int NSF, class TB, int NSB>
static int launch_fused(const WnResFusedArgs& a, int ch, int batch, hipStream_t st) {
    WnResFusedArgs k = a;
    k.t_base = a.t_lo & ~3;
    const int ncol = a.t_hi - k.t_base;
    dim3 g((ncol + RF_COLS - 1) / RF_COLS, batch), b(RF_THREADS);
    const size_t frf = (NSF == 3 ? 1024 : 512) * 2, frb = (NSB == 3 ? 1024 : 512) * 2;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (ch == 32) {
        size_t sh = 8 * frf + 2 * frb + 8 * frb;
        hipLaunchKernelGGL((resblock_bwd_fused_k<TF, NSF, TB, NSB, 32>), g, b, sh, st, k);
    } else if (ch == 64) {
        size_t sh = 32 * frf + 8 * frb + 32 * frb;
        static unsigned long long done = 0;
        if (!((done >> dev) & 1ull)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_bwd_fused_k<TF, NSF, TB, NSB, 64>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            done |= 1ull << dev;
        }
        hipLaunchKernelGGL((resblock_bwd_fused_k<TF, NSF, TB, NSB, 64>), g, b, sh, st, k);
    } else {
        return wn_set_error_msg(-3, "resblock: padded channel count must be 32 or 64");
    }
    WN_CHECK_LAUNCH();
    return 0;
}

int wn_resfused_tiles(int t_lo, int t_hi) {
    if (t_hi <= t_lo) return 0;
    return (t_hi - (t_lo & ~3) + RF_COLS - 1) / RF_COLS;
}

int wn_launch_resblock_bwd_fused(const WnResFusedArgs& a, int ch, int batch, int mode_fwd, int mode_bwd, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    if (mode_fwd == WN_MODE_F16X3 && mode_bwd == WN_MODE_BF16X3) return launch_fused<F16, 3, BF16, 3>(a, ch, batch, st);
    if (mode_fwd == WN_MODE_F16X1 && mode_bwd == WN_MODE_BF16X1) return launch_fused<F16, 1, BF16, 1>(a, ch, batch, st);
    if (mode_fwd == WN_MODE_BF16X3 && mode_bwd == WN_MODE_BF16X3) return launch_fused<BF16, 3, BF16, 3>(a, ch, batch, st);
    if (mode_fwd == WN_MODE_BF16X1 && mode_bwd == WN_MODE_BF16X1) return launch_fused<BF16, 1, BF16, 1>(a, ch, batch, st);
    return wn_set_error_msg(-2, "resblock_bwd_fused: unsupported mode pair");
}
