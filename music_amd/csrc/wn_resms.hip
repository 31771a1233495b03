// Backward of one gated residual block with BOTH weight gradients in the same launch
// ("channel-split" form; CH = 64, recompute in F16x3, gradient products in BF16x3).
//
// Why a second form of the block.  The weight gradients contract over TIME, so their MFMA operands
// need consecutive samples of one channel per lane, while the block computes with time on the lanes.
// In the time-split kernel (wn_resblock.hip: one wave = 64 columns x all channels) a wave would need
// the whole 128x128 + 64x64 result (320 accumulator registers) next to its block state, so dfg and z
// went to HBM and two more launches (wgrad_k) read them back together with x and dy.  Here the 4 waves
// of a workgroup split the CHANNELS of the same 64 columns:
//   * wave g owns dilation channels 16g..16g+15: the f and g row tiles g of the (2CH x 2CH) product and
//     row tile g of Wd^T.  Their packed weights (10 fragments, 80 registers) stay in registers for
//     the whole launch - no weight traffic at all;
//   * the operands every wave needs - x(t-d), x(t) as f16 hi/lo fragments, dy as bf16 fragments - are
//     fetched and split ONCE per workgroup and shared through LDS;
//   * df, dg, z of the wave's 16 channels are turned from "time on lanes" into "time on k" by the
//     matrix core itself: D = A' S with the data as the A operand and a 0/1 selection matrix S as B
//     lands element (channel, time) on lane = channel.  The data are split into their bf16 hi / lo
//     halves first, so every product is (value x 1.0) + zeros: exact.  No LDS round trip, no shuffles;
//   * the matching "time on k" operands x(t-d), x(t), dy (16 consecutive samples of one row per lane,
//     re-ordered by register selection to the k order the transposition produces) are again built
//     once per workgroup and shared through LDS;
//   * wave g accumulates rows [df_g; dg_g] of dWfg (16 tiles) and rows z_g of dWd^T (4 tiles) over all
//     the columns the workgroup walks (persistent workgroups, one slab each, reduced by reduce_slabs_k).
// HBM traffic per block: x, dy, dz-crop in, [df;dg] out (the dx product still reads it): ~4.8
// activation-sized tensors against ~10.8 for resblock_bwd_k + 2 x wgrad_k.  z never leaves the CU.
#include <stdlib.h>
#include <type_traits>
#include "wn_common.h"
#include "wn_kernels.h"

#define MS_THREADS 256
#define MS_CH 64

template <class T>
__device__ __forceinline__ void ms_store_frag(uint16_t* base, int idx, int lane, const Frag<T>& f) {
    u32x4* p = reinterpret_cast<u32x4*>(base) + (size_t)idx * 128 + lane;
    p[0] = __builtin_bit_cast(u32x4, f.hi);
    p[64] = __builtin_bit_cast(u32x4, f.lo);
}

template <bool HAS_DY>
__global__ __launch_bounds__(MS_THREADS) void resblock_bwd_ms_k(WnResMsArgs a) {
    constexpr int CH = MS_CH;
    constexpr int FR = 1024;                                   // halfs per x3 fragment
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    // 2 stages x (16 x fragments [k-step s = (tap, channel half)][N-tile n] + 8 dy fragments [k-step][N-tile]),
    // then 24 "time on k" fragments [(x(t-d) 4 | x(t) 4 | dy 4 row tiles)][k-step]: 72 x 2 KB = 144 KB
    uint16_t* l_xf = lds;
    uint16_t* l_wo = lds + 48 * FR;

    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, q = lane >> 4;
    constexpr bool has_dy = HAS_DY;                            // false: last block (no dz product, no dWd)
    constexpr bool has_d = HAS_DY;

    // ---- stationary weights of this wave's 16 dilation channels
    Frag<F16> wf[4], wg[4];
    Frag<BF16> wd[2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        load_a<F16, 3>(wf[s], a.wfg, g * 4 + s, lane);
        load_a<F16, 3>(wg[s], a.wfg, (4 + g) * 4 + s, lane);
    }
    if (has_dy) {
#pragma unroll
        for (int s = 0; s < 2; ++s) load_a<BF16, 3>(wd[s], a.wdT, g * 2 + s, lane);
    }
    // selection matrices of the in-register transposition (B operands): S0 picks k-slot (q = n>>2, j = n&3),
    // S1 picks (q = n>>2, j = 4 + (n&3)), n = lane & 15
    BF16::vec8 s0, s1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s0[j] = BF16::cvt((q == (c >> 2) && j == (c & 3)) ? 1.0f : 0.0f);
        s1[j] = BF16::cvt((q == (c >> 2) && j == 4 + (c & 3)) ? 1.0f : 0.0f);
    }
    float bias_f[4], bias_g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 16 * g + 4 * q + i;
        bias_f[i] = (a.bias_f && row < a.n_f) ? a.bias_f[row] : 0.f;
        bias_g[i] = (a.bias_g && row < a.n_f) ? a.bias_g[row] : 0.f;
    }

    f32x4 cfg[2][8], cd[4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < 8; ++n) cfg[h][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 4; ++n) cd[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- this workgroup's run of (clip, 64-column step) items (XCD-contiguous)
    const WnBlock blk = wn_block(a.swz);
    const int wgid = blk.x;
    const int total = a.steps_per_clip * a.batch;
    const int item0 = wgid * a.items_per_wg;
    int item_end = item0 + a.items_per_wg;
    if (item_end > total) item_end = total;

    // Software pipeline over the items (all loads unconditional on a clamped item index, so that no
    // loaded value is merged with an old register across a branch):
    //   registers hold the raw x / dy rows of item k+1 (recompute layout) and of item k ("time on k" layout);
    //   item k:  split raw(k+1) -> LDS stage (k+1)&1, re-arm the registers with item k+2
    //            recompute + gate of item k from LDS stage k&1                          | barrier A
    //            split the "time on k" rows of item k -> LDS, re-arm with item k+1      | barrier B
    //            transpose df, dg, z on the matrix core; weight-gradient products
    const float* dy_or_x = has_dy ? a.dy : a.x_in;      // loads stay unconditional (see above)
    struct RawXD { f32x4 x[8]; f32x4 dy[4]; };
    struct RawWO { f32x4 v[3][4]; };
    struct Pos { int b, t0; };
    auto pos_of = [&](int it) {
        it = it < item_end ? it : item_end - 1;
        Pos p;
        p.b = it / a.steps_per_clip;
        p.t0 = a.t_base + 64 * (it - p.b * a.steps_per_clip);
        return p;
    };
    auto next_pos = [&](Pos p, int it_next) {                 // position of item it_next = (item of p) + 1, clamped
        if (it_next >= item_end) return p;
        p.t0 += 64;
        if (p.t0 >= a.t_base + 64 * a.steps_per_clip) { p.t0 = a.t_base; p.b += 1; }
        return p;
    };
    auto load_xd = [&](RawXD& r, Pos ps) {
        const int b = ps.b, t0 = ps.t0;
        const int tl = t0 + 4 * c;
        const float* xin = a.x_in + (size_t)b * a.x_bstride;
        const int tap = g >> 1;                 // wave g converts k-step g = (tap g>>1, channel half g&1)
        const float* p = xin + (size_t)(32 * (g & 1) + 8 * q) * a.pitch + (tap == 0 ? tl - a.d : tl);
#pragma unroll
        for (int j = 0; j < 8; ++j) r.x[j] = ld4u(p + (size_t)j * a.pitch);      // alignment-free form, branch-free (see above)
        {                                       // dy rows 4(g&1).. of k-step g>>1 (last block: dummy rows of x, unused)
            const float* pd = dy_or_x + (size_t)b * a.x_bstride + (size_t)(32 * (g >> 1) + 8 * q + 4 * (g & 1)) * a.pitch + tl;
#pragma unroll
            for (int j = 0; j < 4; ++j) r.dy[j] = ld4(pd + (size_t)j * a.pitch);
        }
    };
    auto fill_xd = [&](const RawXD& r, Pos ps, int stage) {
        const int t0 = ps.t0;
        const int tl = t0 + 4 * c;
        uint16_t* xf = l_xf + (size_t)stage * 24 * FR;
        uint16_t* dyf = xf + 16 * FR;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = r.x[j][n];
            Frag<F16> f;
            split8<F16, 3>(f, v);
            ms_store_frag<F16>(xf, g * 4 + n, lane, f);
        }
        if (has_dy) {
            const int ks = g >> 1, h = g & 1;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const bool ok = tl + n >= a.t_lo && tl + n < a.t_hi;      // columns outside hold no gradient
                uint16_t hh[4], ll[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = ok ? r.dy[j][n] : 0.f;
                    const __bf16 hv = BF16::cvt(x);
                    hh[j] = __builtin_bit_cast(uint16_t, hv);
                    ll[j] = __builtin_bit_cast(uint16_t, BF16::cvt(x - BF16::back(hv)));
                }
                uint16_t* fb = dyf + (size_t)(ks * 4 + n) * FR + lane * 8 + h * 4;
                *reinterpret_cast<uint2*>(fb) = uint2{(uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16)};
                *reinterpret_cast<uint2*>(fb + 512) = uint2{(uint32_t)ll[0] | ((uint32_t)ll[1] << 16), (uint32_t)ll[2] | ((uint32_t)ll[3] << 16)};
            }
        }
    };
    // "time on k" operands: row tile g of x(t-d), x(t) and dy; lane (row c, group q) owns samples t0+16q..+15
    auto load_wo = [&](RawWO& r, Pos ps) {
        const int b = ps.b, t0 = ps.t0;
#pragma unroll
        for (int kind = 0; kind < (has_d ? 3 : 2); ++kind) {
            const float* base = (kind == 2 ? dy_or_x : a.x_in) + (size_t)b * a.x_bstride;
            const float* p = base + (size_t)(16 * g + c) * a.pitch + t0 + 16 * q + (kind == 0 ? -a.d : 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) r.v[kind][e] = ld4u(p + 4 * e);
        }
    };
    auto fill_wo = [&](const RawWO& r, Pos ps) {
        const int t0 = ps.t0;
#pragma unroll
        for (int kind = 0; kind < (has_d ? 3 : 2); ++kind) {
            // k order of the transposed operands: slot j of k-step ks <-> sample 16q + 4(j&3) + 2ks + (j>>2)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = r.v[kind][j & 3][2 * ks + (j >> 2)];
                    if (kind == 2) {
                        const int t = t0 + 16 * q + 4 * (j & 3) + 2 * ks + (j >> 2);
                        if (t < a.t_lo || t >= a.t_hi) x = 0.f;
                    }
                    w[j] = x;
                }
                Frag<BF16> f;
                split8<BF16, 3>(f, w);
                ms_store_frag<BF16>(l_wo, (kind * 4 + g) * 2 + ks, lane, f);
            }
        }
    };

    if (item0 < item_end) {
        RawXD rx;
        RawWO rw;
        Pos p_cur = pos_of(item0);
        Pos p_n1 = next_pos(p_cur, item0 + 1);
        load_xd(rx, p_cur);
        load_wo(rw, p_cur);
        fill_xd(rx, p_cur, item0 & 1);
        load_xd(rx, p_n1);
        __syncthreads();
        for (int item = item0; item < item_end; ++item) {
            const Pos p_n2 = next_pos(p_n1, item + 2);
            const int b = p_cur.b, t0 = p_cur.t0;
            const int tl = t0 + 4 * c;
            // d z-crop rows of this item (used after the recompute MFMAs)
            const float* dzc = a.dz + (size_t)b * a.dz_bstride;
            f32x4 cr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) cr[i] = ld4(dzc + (size_t)(16 * g + 4 * q + i) * a.pitch + tl);
            // stage of the NEXT item; then re-arm the registers two items ahead
            fill_xd(rx, p_n1, (item + 1) & 1);
            load_xd(rx, p_n2);

            // ================= recompute f, g of channels 16g.. ; dz ; gate =================
            const uint16_t* xf = l_xf + (size_t)(item & 1) * 24 * FR;
            const uint16_t* dyf = xf + 16 * FR;
            f32x4 af[4], ag[4], dz[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                af[n] = f32x4{bias_f[0], bias_f[1], bias_f[2], bias_f[3]};
                ag[n] = f32x4{bias_g[0], bias_g[1], bias_g[2], bias_g[3]};
                dz[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            {   // the next fragment is on its way from LDS while the matrix core works on this one
                Frag<F16> bx[2];
                load_a<F16, 3>(bx[0], xf, 0, lane);
#pragma unroll
                for (int idx = 0; idx < 16; ++idx) {
                    if (idx + 1 < 16) load_a<F16, 3>(bx[(idx + 1) & 1], xf, idx + 1, lane);
                    mma<F16, 3>(af[idx & 3], wf[idx >> 2], bx[idx & 1]);
                    mma<F16, 3>(ag[idx & 3], wg[idx >> 2], bx[idx & 1]);
                }
#ifndef MS_NO_SGB
                // pin the order "read fragment i+1 (2 x ds_read_b128), then the 6 MFMAs of fragment i": left to
                // itself the scheduler puts each read right in front of its MFMAs and the LDS latency shows
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int idx = 0; idx < 15; ++idx) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#endif
            }
            if (has_dy) {
                Frag<BF16> by[2];
                load_a<BF16, 3>(by[0], dyf, 0, lane);
#pragma unroll
                for (int idx = 0; idx < 8; ++idx) {
                    if (idx + 1 < 8) load_a<BF16, 3>(by[(idx + 1) & 1], dyf, idx + 1, lane);
                    mma<BF16, 3>(dz[idx & 3], wd[idx >> 2], by[idx & 1]);
                }
            }
            if (a.cond) {       // same conditioning bias as the forward (wavenet_autoencoder/model1.py:183)
                const float* cb = a.cond + (size_t)b * a.cond_bstride;
                int idx[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    int tr = tl + n - a.t_lo;
                    tr = tr < 0 ? 0 : tr;
                    const int ix = a.cond_mode == 1 ? tr / a.cond_q : tr % a.cond_le;
                    idx[n] = ix < a.cond_le ? ix : a.cond_le - 1;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* rf = cb + (size_t)(16 * g + 4 * q + i) * a.cond_pitch;
                    const float* rg = cb + (size_t)(CH + 16 * g + 4 * q + i) * a.cond_pitch;
#pragma unroll
                    for (int n = 0; n < 4; ++n) { af[n][i] += rf[idx[n]]; ag[n][i] += rg[idx[n]]; }
                }
            }
            float* dfg = a.dfg + (size_t)b * a.dfg_bstride;
            float df[4][4], dg[4][4], zz[4][4];                      // [n][i]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * g + 4 * q + i;
                f32x4 gz = {dz[0][i], dz[1][i], dz[2][i], dz[3][i]};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (tl + e >= a.z_lo && tl + e < a.t_hi) gz[e] += cr[i][e];
                f32x4 sdf, sdg;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const bool ok = tl + n >= a.t_lo && tl + n < a.t_hi;
                    const float th = wn_tanh(af[n][i]);
                    const float sg = wn_sigmoid(ag[n][i]);
                    const float vz = ok ? th * sg : 0.f;
                    const float vf = ok ? gz[n] * sg * (1.0f - th * th) : 0.f;
                    const float vg = ok ? gz[n] * th * sg * (1.0f - sg) : 0.f;
                    zz[n][i] = vz; df[n][i] = vf; dg[n][i] = vg;
                    sdf[n] = vf; sdg[n] = vg;
                }
                st4m(dfg + (size_t)row * a.pitch + tl, sdf, tl, a.t_lo, a.t_hi);
                st4m(dfg + (size_t)(CH + row) * a.pitch + tl, sdg, tl, a.t_lo, a.t_hi);
            }
            __syncthreads();                    // A: every wave is past the products of the previous item
            fill_wo(rw, p_cur);
            load_wo(rw, p_n1);
            __syncthreads();                    // B: "time on k" operands of this item (and the next recompute stage) are in LDS

            // ================= transpose df, dg, z on the matrix core; weight-gradient products =================
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 tdf[2][2], tdg[2][2], tz[2][2];               // [which N-tile of the pair][hi/lo]
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n = 2 * ks + u;
                    float v[8] = {df[n][0], df[n][1], df[n][2], df[n][3], dg[n][0], dg[n][1], dg[n][2], dg[n][3]};
                    Frag<BF16> a2;
                    split8<BF16, 3>(a2, v);
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    tdf[u][0] = BF16::mfma(a2.hi, s0, zero);
                    tdg[u][0] = BF16::mfma(a2.hi, s1, zero);
                    tdf[u][1] = BF16::mfma(a2.lo, s0, zero);
                    tdg[u][1] = BF16::mfma(a2.lo, s1, zero);
                    if (has_d) {
                        float vz[8] = {zz[n][0], zz[n][1], zz[n][2], zz[n][3], 0.f, 0.f, 0.f, 0.f};
                        Frag<BF16> a3;
                        split8<BF16, 3>(a3, vz);
                        tz[u][0] = BF16::mfma(a3.hi, s0, zero);
                        tz[u][1] = BF16::mfma(a3.lo, s0, zero);
                    }
                }
                Frag<BF16> adf, adg, az;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    adf.hi[j] = BF16::cvt(tdf[j >> 2][0][j & 3]);
                    adf.lo[j] = BF16::cvt(tdf[j >> 2][1][j & 3]);
                    adg.hi[j] = BF16::cvt(tdg[j >> 2][0][j & 3]);
                    adg.lo[j] = BF16::cvt(tdg[j >> 2][1][j & 3]);
                    az.hi[j] = BF16::cvt(tz[j >> 2][0][j & 3]);
                    az.lo[j] = BF16::cvt(tz[j >> 2][1][j & 3]);
                }
                {
                    constexpr int NTILES = has_d ? 12 : 8;
                    Frag<BF16> bo[2];
                    load_a<BF16, 3>(bo[0], l_wo, ks, lane);
#pragma unroll
                    for (int nt = 0; nt < NTILES; ++nt) {
                        if (nt + 1 < NTILES) load_a<BF16, 3>(bo[(nt + 1) & 1], l_wo, (nt + 1) * 2 + ks, lane);
                        if (nt < 8) {
                            mma<BF16, 3>(cfg[0][nt], adf, bo[nt & 1]);
                            mma<BF16, 3>(cfg[1][nt], adg, bo[nt & 1]);
                        } else {
                            mma<BF16, 3>(cd[nt - 8], az, bo[nt & 1]);
                        }
                    }
                }
            }
            p_cur = p_n1;
            p_n1 = p_n2;
        }
    }

    // ---- slab of this workgroup (every workgroup writes one, also an idle one: zeros)
    float* sfg = a.slab_fg + (size_t)wgid * (4 * CH * CH);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                sfg[(size_t)(h * CH + 16 * g + 4 * q + i) * (2 * CH) + nt * 16 + c] = cfg[h][nt][i];
    if (has_d && a.slab_d) {
        float* sd = a.slab_d + (size_t)wgid * (CH * CH);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                sd[(size_t)(r * 16 + c) * CH + 16 * g + 4 * q + i] = cd[r][i];
    }
}

int wn_ms_two_role() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("WN_MS_RW");
        v = e ? (atoi(e) != 0) : 1;
    }
    return v;
}

static void ms_plan(int t_lo, int t_hi, int batch, int& t_base, int& steps, int& ipw, int& nwg) {
    t_base = t_lo & ~3;
    steps = (t_hi - t_base + 63) / 64;
    const int total = steps * batch;
    ipw = (total + 255) / 256;
    if (ipw < 1) ipw = 1;
    nwg = (total + ipw - 1) / ipw;
    if (nwg < 1) nwg = 1;
}

int wn_resms_slabs(int t_lo, int t_hi, int batch) {
    if (t_hi <= t_lo || batch <= 0) return 0;
    int tb, steps, ipw, nwg;
    if (wn_ms_two_role()) wn_resrw_plan(t_lo, t_hi, batch, tb, steps, ipw, nwg);
    else ms_plan(t_lo, t_hi, batch, tb, steps, ipw, nwg);
    return nwg;
}

int wn_launch_resblock_bwd_ms(const WnResMsArgs& a, int ch, int batch, int mode_fwd, int mode_bwd, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    if (ch != MS_CH) return wn_set_error_msg(-3, "resblock_bwd_ms: 64 padded channels only");
    if (mode_fwd != WN_MODE_F16X3 || mode_bwd != WN_MODE_BF16X3)
        return wn_set_error_msg(-2, "resblock_bwd_ms: (f16x3, bf16x3) only");
    if (wn_ms_two_role()) return wn_launch_resblock_bwd_rw(a, batch, st);
    WnResMsArgs k = a;
    int nwg;
    ms_plan(a.t_lo, a.t_hi, batch, k.t_base, k.steps_per_clip, k.items_per_wg, nwg);
    k.batch = batch;
    k.swz = wn_xcd_swizzle_enabled();
    const size_t sh = (size_t)(2 * (16 + 8) + 24) * 1024 * sizeof(uint16_t);
    static unsigned long long done = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_bwd_ms_k<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_bwd_ms_k<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        done |= 1ull << dev;
    }
    if (k.dy && k.slab_d) hipLaunchKernelGGL(resblock_bwd_ms_k<true>, dim3(nwg), dim3(MS_THREADS), sh, st, k);
    else hipLaunchKernelGGL(resblock_bwd_ms_k<false>, dim3(nwg), dim3(MS_THREADS), sh, st, k);
    WN_CHECK_LAUNCH();
    return 0;
}
