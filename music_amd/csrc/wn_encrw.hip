// Backward of one ENCODER block of the autoencoder (wavenet_autoencoder/model1.py:137-152) with both weight
// gradients in the launch, two-role form (see wn_resrw.hip; CH = 64 padded channels, gradient products in BF16x3).
//   forward:  h = Wdil [relu x(t-d); relu x(t)] ; x' = Wd relu(h) + x(t)        (h = the stored pre-activation)
//   backward: dr = Wd^T dy ; dh = dr * [h > 0]                                  -> dh to HBM (the dx product reads it)
//             dWd  = sum_t dy relu(h)^T ; dWdil = sum_t dh [relu x(t-d); relu x(t)]^T   -> one slab per workgroup
// Nothing is recomputed (h comes from the forward), so compared with resblock_bwd_rw_k there is no x-fragment stage:
//   * R waves (0..3; wave g owns h channels 16g..16g+15): Wd^T fragments in registers, fill the dy fragments of the
//     next item, dr out of LDS, mask with h, store dh, leave dh and relu(h) in LDS as [channel][time] hi/lo tiles;
//   * W waves (4..7): relu x(t-d), relu x(t), dy as [row][time] tiles, products into 12 accumulator tiles.
// 32-column items, two LDS stages of 48 KB, one barrier per item, items interleaved within an XCD.
// The data gradient dx = [x > 0] (Wdil1^T dh[t] + Wdil0^T dh[t+d]) + dy stays a wn_chan_gemm launch.
#include <stdlib.h>
#include "wn_common.h"
#include "wn_kernels.h"

#define ER_THREADS 512
#define ER_CH 64
#define ER_COLS 32
#define ER_DYF 0
#define ER_WO 4096           // halfs: 4 dy fragments | 12 operand tiles | 8 result tiles
#define ER_T 16384
#define ER_STAGE 24576

typedef float er_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 er_bf16x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) ErF2U { float v[2]; };
__device__ __forceinline__ er_f32x2 er_ld2u(const float* p) {
    ErF2U u = *reinterpret_cast<const ErF2U*>(p);
    er_f32x2 r = {u.v[0], u.v[1]};
    return r;
}
__device__ __forceinline__ void er_split2(float a, float b, uint32_t& hi, uint32_t& lo) { split2<BF16>(a, b, hi, lo); }

// WnResMsArgs fields as used here: x_in = x_i, dy, dz = h (pre-activation, dz_bstride), dfg = dh out (CH rows,
// dfg_bstride), wdT, slab_fg (CH x 2CH per workgroup), slab_d (CH x CH), d, t_lo, t_hi, z_lo = first column on
// which dy exists (the top block's gradient only exists on the pooled crop), t_base, steps_per_clip, items_per_wg, batch.
__global__ __launch_bounds__(ER_THREADS) void enc_bwd_rw_k(WnResMsArgs a) {
    constexpr int CH = ER_CH;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wv & 3;
    const int c = lane & 15, q = lane >> 4;
    const int tile_rd = (16 * q + (c ^ q)) * 8;                     // see resblock_bwd_rw_k

    int first, cnt, j;
    if (a.swz) {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int qn = nwg >> 3, rn = nwg & 7, xcd = id & 7;
        first = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
        cnt = xcd < rn ? qn + 1 : qn;
        j = id >> 3;
    } else {
        first = 0; cnt = gridDim.x; j = blockIdx.x;
    }
    const int wgid = first + j;
    const int total = a.steps_per_clip * a.batch;
    const int i_lo = first * a.items_per_wg + j;
    int i_hi = (first + cnt) * a.items_per_wg;
    if (i_hi > total) i_hi = total;
    const int n_items = i_lo < i_hi ? (i_hi - i_lo + cnt - 1) / cnt : 0;

    struct Pos { int b, t0; bool live; };
    auto pos_k = [&](int k) {
        const bool live = k < n_items;
        k = k < n_items ? k : n_items - 1;
        int it = i_lo + (k < 0 ? 0 : k) * cnt;
        it = it < total ? it : total - 1;
        Pos p;
        p.b = it / a.steps_per_clip;
        p.t0 = a.t_base + ER_COLS * (it - p.b * a.steps_per_clip);
        p.live = live;
        return p;
    };

    {   // stage 1 of the [row][time] arrays is multiplied once before anything was written to it: zeros
        u32x4* z = reinterpret_cast<u32x4*>(lds + ER_STAGE + ER_WO);
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 5; ++k) z[k * ER_THREADS + threadIdx.x] = zero;       // 20 tiles x 2 KB = 40 KB
    }

    if (wv < 4) {
        // =========================== R waves ===========================
        Frag<BF16> wd[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) load_a<BF16, 3>(wd[s], a.wdT, g * 2 + s, lane);
        int t_wr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t_wr[i] = (16 * (c >> 2) + ((4 * q + i) ^ (c >> 2))) * 8 + (c & 3) * 2;
        struct RawD { er_f32x2 dy[4]; };
        // dy fragments ("time on lanes"): wave g converts rows 4(g&1).. of k-step g>>1, both N-tiles
        auto load_dy = [&](RawD& r, Pos ps) {
            const int tl = ps.t0 + 2 * c;
            const float* pd = ps.live ? a.dy + (size_t)ps.b * a.x_bstride + (size_t)(32 * (g >> 1) + 8 * q + 4 * (g & 1)) * a.pitch + tl : a.dy;
            const size_t rp = ps.live ? (size_t)a.pitch : 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) r.dy[jj] = er_ld2u(pd + jj * rp);
        };
        auto fill_dy = [&](const RawD& r, Pos ps, int stage) {
            const int tl = ps.t0 + 2 * c;
            uint16_t* dyf = lds + (size_t)stage * ER_STAGE + ER_DYF;
            const int ks = g >> 1, h = g & 1;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const bool ok = tl + n >= a.z_lo && tl + n < a.t_hi;
                uint32_t hh[2], ll[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    er_split2(ok ? r.dy[2 * jj][n] : 0.f, ok ? r.dy[2 * jj + 1][n] : 0.f, hh[jj], ll[jj]);
                uint16_t* fb = dyf + (size_t)(ks * 2 + n) * 1024 + lane * 8 + h * 4;
                *reinterpret_cast<uint2*>(fb) = uint2{hh[0], hh[1]};
                *reinterpret_cast<uint2*>(fb + 512) = uint2{ll[0], ll[1]};
            }
        };
        auto load_h = [&](er_f32x2* hr, Pos ps) {
            const float* hp = ps.live ? a.dz + (size_t)ps.b * a.dz_bstride + (size_t)(16 * g + 4 * q) * a.pitch + ps.t0 + 2 * c : a.dz;
            const size_t rp = ps.live ? (size_t)a.pitch : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) hr[i] = er_ld2u(hp + i * rp);
        };
        er_f32x2 hA[4], hB[4];
        RawD d0, d1;                                        // d1 / d0: raw dy rows of items it+1 / it+2
        load_h(hA, pos_k(0));
        load_h(hB, pos_k(1));
        load_dy(d0, pos_k(0));
        load_dy(d1, pos_k(1));
        fill_dy(d0, pos_k(0), 0);
        load_dy(d0, pos_k(2));
        __syncthreads();
        auto r_body = [&](const int it, er_f32x2* hr, RawD& rd) {
            const Pos ps = pos_k(it);
            const bool live = it < n_items;
            fill_dy(rd, pos_k(it + 1), (it + 1) & 1);
            load_dy(rd, pos_k(it + 3));
            const int tl = ps.t0 + 2 * c;
            uint16_t* st = lds + (size_t)(it & 1) * ER_STAGE;
            const uint16_t* dyf = st + ER_DYF;
            f32x4 dr[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int idx = 0; idx < 4; ++idx) {
                Frag<BF16> by;
                load_a<BF16, 3>(by, dyf, idx, lane);
                mma<BF16, 3>(dr[idx & 1], wd[idx >> 1], by);
            }
            float* dh = a.dfg + (size_t)ps.b * a.dfg_bstride;
            uint16_t* tt = st + ER_T;
            const bool ok0 = live && tl >= a.t_lo && tl < a.t_hi, ok1 = live && tl + 1 >= a.t_lo && tl + 1 < a.t_hi;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * g + 4 * q + i;
                float vh[2], vz[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const bool ok = n ? ok1 : ok0;
                    const float hv = hr[i][n];
                    vh[n] = (ok && hv > 0.f) ? dr[n][i] : 0.f;
                    vz[n] = ok ? fmaxf(hv, 0.f) : 0.f;
                }
                float* pf = dh + (size_t)row * a.pitch + tl;
                if (ok0 && ok1) {
                    *reinterpret_cast<ErF2U*>(pf) = ErF2U{{vh[0], vh[1]}};
                } else {
                    if (ok0) pf[0] = vh[0];
                    if (ok1) pf[1] = vh[1];
                }
                auto put = [&](int kind, const float* v) {
                    uint32_t hi, lo;
                    er_split2(v[0], v[1], hi, lo);
                    uint16_t* p = tt + (kind * 4 + g) * 1024 + t_wr[i];
                    *reinterpret_cast<uint32_t*>(p) = hi;
                    *reinterpret_cast<uint32_t*>(p + 512) = lo;
                };
                put(0, vh);
                put(1, vz);
            }
            load_h(hr, pos_k(it + 2));
            __syncthreads();
        };
        for (int it = 0; it < n_items; it += 2) {
            r_body(it, hA, d1);
            r_body(it + 1, hB, d0);
        }
        return;
    }

    // =========================== W waves ===========================
    f32x4 cx[8], cd[4];
#pragma unroll
    for (int n = 0; n < 8; ++n) cx[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 4; ++n) cd[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    struct RawWO { f32x4 v[3][2]; };
    // [row][time] operands: row tile g of relu x(t-d), relu x(t) and dy; lane (row c, q) owns samples t0 + 8q .. + 7
    auto load_wo = [&](RawWO& r, Pos ps) {
#pragma unroll
        for (int kind = 0; kind < 3; ++kind) {
            const float* base = (kind == 2 ? a.dy : a.x_in) + (size_t)ps.b * a.x_bstride;
            const float* p = ps.live ? base + (size_t)(16 * g + c) * a.pitch + ps.t0 + 8 * q + (kind == 0 ? -a.d : 0) : a.x_in;
            r.v[kind][0] = ld4u(p);
            r.v[kind][1] = ld4u(p + 4);
        }
    };
    auto fill_wo = [&](const RawWO& r, Pos ps, int stage) {
        uint16_t* wo = lds + (size_t)stage * ER_STAGE + ER_WO;
#pragma unroll
        for (int kind = 0; kind < 3; ++kind) {
            float w[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                float x = r.v[kind][jj >> 2][jj & 3];
                if (kind == 2) {
                    const int t = ps.t0 + 8 * q + jj;
                    if (t < a.z_lo || t >= a.t_hi) x = 0.f;
                } else {
                    x = fmaxf(x, 0.f);
                }
                w[jj] = x;
            }
            u32x4 fh, fl;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                uint32_t hi, lo;
                er_split2(w[2 * jj], w[2 * jj + 1], hi, lo);
                fh[jj] = hi;
                fl[jj] = lo;
            }
            u32x4* p = reinterpret_cast<u32x4*>(wo + (kind * 4 + g) * 1024 + tile_rd);
            p[0] = fh;
            p[64] = fl;
        }
    };
    auto load_tile = [&](Frag<BF16>& f, const uint16_t* base, int tile) {
        const u32x4* p = reinterpret_cast<const u32x4*>(base + tile * 1024 + tile_rd);
        f.hi = __builtin_bit_cast(bf16x8, p[0]);
        f.lo = __builtin_bit_cast(bf16x8, p[64]);
    };
    auto wgrad = [&](int stage) {
        const uint16_t* wo = lds + (size_t)stage * ER_STAGE + ER_WO;
        const uint16_t* tt = lds + (size_t)stage * ER_STAGE + ER_T;
        Frag<BF16> adh, azr;
        load_tile(adh, tt, g);
        load_tile(azr, tt, 4 + g);
        Frag<BF16> bo[2];
        load_tile(bo[0], wo, 0);
#pragma unroll
        for (int nt = 0; nt < 12; ++nt) {
            if (nt + 1 < 12) load_tile(bo[(nt + 1) & 1], wo, nt + 1);
            if (nt < 8) mma<BF16, 3>(cx[nt], adh, bo[nt & 1]);
            else mma<BF16, 3>(cd[nt - 8], azr, bo[nt & 1]);
        }
    };
    {
        RawWO rw0, rw1;
        load_wo(rw0, pos_k(0));
        load_wo(rw1, pos_k(1));
        __syncthreads();
        auto w_body = [&](const int it, RawWO& rw) {
            fill_wo(rw, pos_k(it), it & 1);
            load_wo(rw, pos_k(it + 2));
            wgrad((it + 1) & 1);
            __syncthreads();
        };
        for (int it = 0; it < n_items; it += 2) {
            w_body(it, rw0);
            w_body(it + 1, rw1);
        }
        wgrad(1);
    }
    float* sfg = a.slab_fg + (size_t)wgid * (2 * CH * CH);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            sfg[(size_t)(16 * g + 4 * q + i) * (2 * CH) + nt * 16 + c] = cx[nt][i];
    float* sd = a.slab_d + (size_t)wgid * (CH * CH);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            sd[(size_t)(r * 16 + c) * CH + 16 * g + 4 * q + i] = cd[r][i];
}

int wn_enc_bwd_slabs(int t_lo, int t_hi, int batch) {
    if (t_hi <= t_lo || batch <= 0) return 0;
    int tb, steps, ipw, nwg;
    wn_resrw_plan(t_lo, t_hi, batch, tb, steps, ipw, nwg);
    return nwg;
}

int wn_launch_enc_bwd_rw(const WnResMsArgs& a, int ch, int batch, int mode_bwd, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    if (ch != ER_CH) return wn_set_error_msg(-3, "enc_resblock_bwd: 64 padded channels only");
    if (mode_bwd != WN_MODE_BF16X3) return wn_set_error_msg(-2, "enc_resblock_bwd: bf16x3 only");
    WnResMsArgs k = a;
    int nwg;
    wn_resrw_plan(a.t_lo, a.t_hi, batch, k.t_base, k.steps_per_clip, k.items_per_wg, nwg);
    k.batch = batch;
    k.swz = wn_xcd_swizzle_enabled();
    const size_t sh = (size_t)2 * ER_STAGE * sizeof(uint16_t);
    static WnDevOnce done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done.need(dev)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_bwd_rw_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        done.done(dev);
    }
    hipLaunchKernelGGL(enc_bwd_rw_k, dim3(nwg), dim3(ER_THREADS), sh, st, k);
    WN_CHECK_LAUNCH();
    return 0;
}
