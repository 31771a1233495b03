// Backward of one gated residual block with both weight gradients, "two-role" form
// (CH = 64, recompute in F16x3, gradient products in BF16x3).  Same contract, arguments and slab
// format as resblock_bwd_ms_k (wn_resms.hip); what changes is who does what inside the workgroup.
//
// resblock_bwd_ms_k needs ~500 registers per wave (stationary weights + this wave's weight-gradient
// tiles + the block state), i.e. ONE wave per SIMD, and its matrix-core work, its VALU work (operand
// splits, gate) and its memory waits simply add up.  Here the workgroup has 8 waves of <= 256
// registers, two per SIMD, with complementary jobs:
//   * R waves (0..3; wave g owns dilation channels 16g..16g+15) keep the packed f/g/Wd^T weights in
//     registers (80), split the raw x rows of the NEXT item into its recompute fragments (k-step g;
//     they had the slack: -DRW_XF_W gives that job back to the W waves), recompute f, g, form dz, gate,
//     write [df;dg] to HBM, and leave df, dg, z in LDS as 16-bit hi/lo arrays [channel][time];
//   * W waves (4..7) keep the weight-gradient tiles in registers (80), turn dy into the recompute
//     fragments of the next item and x(t-d), x(t), dy of the current one into [row][time] arrays, and
//     multiply [df;dg;z] of the previous item by them.  The [channel][time] arrays ARE the "time on k"
//     operands (a lane reads 8 consecutive samples of one row with one ds_read_b128), so the
//     transposition the weight gradients need is the LDS round trip itself - no matrix-core
//     transposition, no accumulator read-back.
// Items are 32 columns, every LDS buffer has two stages (2 x 72 KB), there is ONE barrier per item, and
// every prefetch register set is re-armed two items ahead (the loops are unrolled by two: no copies):
// in iteration i the R waves fill item i+1's x fragments and work on item i (stage i&1) while the W waves
// fill item i+1's dy fragments and item i's [row][time] operands and multiply item i-1 (stage (i-1)&1).
// While one wave of a SIMD is in its MFMA phase the other is mostly in its VALU phase.  The workgroups of
// an XCD walk their item range interleaved (every dilated tap an L2 hit); tile slots in LDS are XOR-placed
// (bank-conflict free for the three access patterns); the slabs leave with streaming stores.
//
// RW_T_* macros are timing-build switches (remove one ingredient; results are then wrong) used to see
// what a launch is made of: -DRW_T_NOREC / NOWG / NOFILL / NOGATE / NOCR via `make EXTRA=...`.
#include <stdlib.h>
#include "wn_common.h"
#include "wn_kernels.h"

#define RW_THREADS 512
#define RW_CH 64
#define RW_COLS 32

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) F2U { float v[2]; };
__device__ __forceinline__ f32x2 ld2u(const float* p) {
    F2U u = *reinterpret_cast<const F2U*>(p);
    f32x2 r = {u.v[0], u.v[1]};
    return r;
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// (a, b) -> packed bf16 pairs hi = (bf16(a), bf16(b)) and lo = (bf16(a - hi_a), bf16(b - hi_b)): 6 VALU per pair
__device__ __forceinline__ void split2_bf16(float a, float b, uint32_t& hi, uint32_t& lo) { split2<BF16>(a, b, hi, lo); }

// LDS map of one stage, in halfs (uint16): 8 x fragments | 4 dy fragments | 12 operand tiles | 12 result tiles
#define RW_XF 0
#define RW_DYF 8192
#define RW_WO 12288
#define RW_T 24576
#define RW_STAGE 36864


template <class T>
__device__ __forceinline__ void rw_store_frag(uint16_t* base, int idx, int lane, const Frag<T>& f) {
    u32x4* p = reinterpret_cast<u32x4*>(base) + (size_t)idx * 128 + lane;
    p[0] = __builtin_bit_cast(u32x4, f.hi);
    p[64] = __builtin_bit_cast(u32x4, f.lo);
}

template <bool HAS_DY>
__global__ __launch_bounds__(RW_THREADS) void resblock_bwd_rw_k(WnResMsArgs a) {
    constexpr int CH = RW_CH;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wv & 3;
    const int c = lane & 15, q = lane >> 4;
    // 16-byte chunk (row, q) of a [16 rows][32 samples] tile plane sits at slot 16q + (row ^ q): conflict free for
    // the ds_read_b128 lane groups and the 8-lane ds_write_b128 groups of lanes (row = c, q) AND for the 32-lane
    // ds_write_b32 groups of lanes (row = 4q + i, samples 2c, 2c+1) (checked against the LDS banking table of
    // MI355X_MICROARCH.md; SQ_LDS_BANK_CONFLICT 28 % -> see profiles)
    const int tile_rd = (16 * q + (c ^ q)) * 8;                     // halfs; this lane's chunk as an MFMA operand

    // Items of this workgroup.  The workgroups that share an XCD (one contiguous run of logical ids, see
    // wn_block) own one contiguous range of items and walk it INTERLEAVED: workgroup j takes items
    // j, j + cnt, j + 2 cnt, ...  At any moment the XCD works on ~cnt neighbouring items, so the rows the
    // dilated tap needs (x(t-d), d <= 512 = 16 items back) were fetched by a neighbour moments ago and
    // are still in that XCD's L2, for every dilation.
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
    auto pos_k = [&](int k) {                                  // position of this workgroup's k-th item, clamped
        const bool live = k < n_items;
        k = k < n_items ? k : n_items - 1;
        int it = i_lo + (k < 0 ? 0 : k) * cnt;
        it = it < total ? it : total - 1;
        Pos p;
        p.b = it / a.steps_per_clip;
        p.t0 = a.t_base + RW_COLS * (it - p.b * a.steps_per_clip);
        p.live = live;
        return p;
    };

    // stage 1 of the [row][time] arrays is multiplied once before anything was written to it: zeros
    {
        u32x4* z = reinterpret_cast<u32x4*>(lds + RW_STAGE + RW_WO);
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 6; ++k) z[k * RW_THREADS + threadIdx.x] = zero;       // 24 tiles x 2 KB = 48 KB
    }

    const float* dy_or_x = HAS_DY ? a.dy : a.x_in;          // loads stay unconditional
    struct RawX { f32x2 x[8]; };
    struct RawD { f32x2 dy[4]; };
    // recompute operands ("time on lanes"; lane (c, q) of N-tile n holds sample t0 + 2c + n): wave g of a role
    // converts k-step g = (tap g>>1, channel half g&1) of x / rows 4(g&1).. of k-step g>>1 of dy, both N-tiles.
    // The R waves do x (they have the slack: RW_XF_W moves it back to the W waves), the W waves do dy.
    auto load_x = [&](RawX& r, Pos ps) {
        const int tl = ps.t0 + 2 * c;
        const float* xin = a.x_in + (size_t)ps.b * a.x_bstride;
        // a position past the end (prefetch beyond the last item) reads ONE address in every lane: no traffic,
        // and the loads stay unconditional
        const float* p = ps.live ? xin + (size_t)(32 * (g & 1) + 8 * q) * a.pitch + ((g >> 1) == 0 ? tl - a.d : tl) : a.x_in;
        const size_t rp = ps.live ? (size_t)a.pitch : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) r.x[j] = ld2u(p + j * rp);
    };
    auto load_dy = [&](RawD& r, Pos ps) {
        const int tl = ps.t0 + 2 * c;
        const float* pd = ps.live ? dy_or_x + (size_t)ps.b * a.x_bstride + (size_t)(32 * (g >> 1) + 8 * q + 4 * (g & 1)) * a.pitch + tl : dy_or_x;
        const size_t rp = ps.live ? (size_t)a.pitch : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.dy[j] = ld2u(pd + j * rp);
    };
    auto fill_x = [&](const RawX& r, int stage) {
        uint16_t* xf = lds + (size_t)stage * RW_STAGE + RW_XF;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = r.x[j][n];
            Frag<F16> f;
            split8<F16, 3>(f, v);
            rw_store_frag<F16>(xf, g * 2 + n, lane, f);
        }
    };
    auto fill_dy = [&](const RawD& r, Pos ps, int stage) {
        if (HAS_DY) {
            const int tl = ps.t0 + 2 * c;
            uint16_t* dyf = lds + (size_t)stage * RW_STAGE + RW_DYF;
            const int ks = g >> 1, h = g & 1;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const bool ok = tl + n >= a.t_lo && tl + n < a.t_hi;      // columns outside hold no gradient
                uint32_t hh[2], ll[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    split2_bf16(ok ? r.dy[2 * j][n] : 0.f, ok ? r.dy[2 * j + 1][n] : 0.f, hh[j], ll[j]);
                uint16_t* fb = dyf + (size_t)(ks * 2 + n) * 1024 + lane * 8 + h * 4;
                *reinterpret_cast<uint2*>(fb) = uint2{hh[0], hh[1]};
                *reinterpret_cast<uint2*>(fb + 512) = uint2{ll[0], ll[1]};
            }
        }
    };

    if (wv < 4) {
        // =========================== R waves: recompute, dz, gate ===========================
        Frag<F16> wf[4], wg[4];
        Frag<BF16> wd[2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            load_a<F16, 3>(wf[s], a.wfg, g * 4 + s, lane);
            load_a<F16, 3>(wg[s], a.wfg, (4 + g) * 4 + s, lane);
        }
        if (HAS_DY) {
#pragma unroll
            for (int s = 0; s < 2; ++s) load_a<BF16, 3>(wd[s], a.wdT, g * 2 + s, lane);
        }
        float bias_f[4], bias_g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * g + 4 * q + i;
            bias_f[i] = (a.bias_f && row < a.n_f) ? a.bias_f[row] : 0.f;
            bias_g[i] = (a.bias_g && row < a.n_f) ? a.bias_g[row] : 0.f;
        }
        // this lane's dwords in the result tiles: row 4q + i, samples 2c, 2c+1 -> chunk c>>2, dword c&3
        int t_wr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t_wr[i] = (16 * (c >> 2) + ((4 * q + i) ^ (c >> 2))) * 8 + (c & 3) * 2;

        auto load_cr = [&](f32x2* cr, Pos ps) {
            const float* dzc = ps.live ? a.dz + (size_t)ps.b * a.dz_bstride + (size_t)(16 * g + 4 * q) * a.pitch + ps.t0 + 2 * c : a.dz;
            const size_t rp = ps.live ? (size_t)a.pitch : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) cr[i] = ld2u(dzc + i * rp);
        };

        // The loop is unrolled by two so that every prefetch register set is re-armed TWO items ahead without
        // copies (a loaded HBM latency is about one item long); an odd item count is padded with a void item
        // (clamped position, everything masked: zeros in LDS, no stores).
        f32x2 crA[4], crB[4];
        load_cr(crA, pos_k(0));
        load_cr(crB, pos_k(1));
        RawX x0, x1;                                        // x1 / x0 hold the raw rows of items it+1 / it+2
        load_x(x0, pos_k(0));
        load_x(x1, pos_k(1));
        fill_x(x0, 0);
        load_x(x0, pos_k(2));
        __syncthreads();                                    // stage 0 operands of the first item are in LDS
        auto r_body = [&](const int it, f32x2* cr, RawX& rx) {
            fill_x(rx, (it + 1) & 1);                        // recompute operands of the next item
            load_x(rx, pos_k(it + 3));
            const Pos p_cur = pos_k(it);
            const bool live = it < n_items;
            const int b = p_cur.b, t0 = p_cur.t0;
            const int tl = t0 + 2 * c;
            uint16_t* st = lds + (size_t)(it & 1) * RW_STAGE;
            const uint16_t* xf = st + RW_XF;
            const uint16_t* dyf = st + RW_DYF;

            f32x4 af[2], ag[2], dz[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                af[n] = f32x4{bias_f[0], bias_f[1], bias_f[2], bias_f[3]};
                ag[n] = f32x4{bias_g[0], bias_g[1], bias_g[2], bias_g[3]};
                dz[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            {
                // One accumulator is touched by every 4th (5th) MFMA only: the three products of an x3 term and
                // the f / g / N-tile accumulators are walked in rotation, with the dz products woven in, so no
                // MFMA waits for the result of the one in front of it.
                auto term = [](auto tr, f32x4& acc, const auto& wa, const auto& xb, int t) {
                    typedef decltype(tr) TT;
                    acc = t == 0 ? TT::mfma(wa.lo, xb.hi, acc) : t == 1 ? TT::mfma(wa.hi, xb.lo, acc) : TT::mfma(wa.hi, xb.hi, acc);
                };
                Frag<F16> bx[2][2];
                Frag<BF16> by[2];
                load_a<F16, 3>(bx[0][0], xf, 0, lane);
                load_a<F16, 3>(bx[0][1], xf, 1, lane);
                if (HAS_DY) load_a<BF16, 3>(by[0], dyf, 0, lane);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks + 1 < 4) {
                        load_a<F16, 3>(bx[(ks + 1) & 1][0], xf, 2 * ks + 2, lane);
                        load_a<F16, 3>(bx[(ks + 1) & 1][1], xf, 2 * ks + 3, lane);
                        if (HAS_DY) load_a<BF16, 3>(by[(ks + 1) & 1], dyf, ks + 1, lane);      // dy fragment (k-step (ks+1)>>1, N-tile (ks+1)&1)
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        term(F16(), af[0], wf[ks], bx[ks & 1][0], t);
                        term(F16(), ag[0], wg[ks], bx[ks & 1][0], t);
                        term(F16(), af[1], wf[ks], bx[ks & 1][1], t);
                        term(F16(), ag[1], wg[ks], bx[ks & 1][1], t);
                        if (HAS_DY) term(BF16(), dz[ks & 1], wd[ks >> 1], by[ks & 1], t);
                    }
                }
            }
            if (a.cond) {       // same conditioning bias as the forward (wavenet_autoencoder/model1.py:183)
                const float* cb = a.cond + (size_t)b * a.cond_bstride;
                int idx[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
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
                    for (int n = 0; n < 2; ++n) { af[n][i] += rf[idx[n]]; ag[n][i] += rg[idx[n]]; }
                }
            }
            float* dfg = a.dfg + (size_t)b * a.dfg_bstride;
            uint16_t* tt = st + RW_T;
            const bool ok0 = live && tl >= a.t_lo && tl < a.t_hi, ok1 = live && tl + 1 >= a.t_lo && tl + 1 < a.t_hi;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * g + 4 * q + i;
                float vz[2], vf[2], vg[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const bool ok = n ? ok1 : ok0;
                    float gz = dz[n][i];
                    if (tl + n >= a.z_lo && tl + n < a.t_hi) gz += cr[i][n];
                    const WnGateD gd = wn_gate_d(af[n][i], ag[n][i]);
                    vz[n] = ok ? gd.z : 0.f;
                    vf[n] = ok ? gz * gd.dzdf : 0.f;
                    vg[n] = ok ? gz * gd.dzdg : 0.f;
                }
                float* pf = dfg + (size_t)row * a.pitch + tl;
                float* pg = dfg + (size_t)(CH + row) * a.pitch + tl;
                if (ok0 && ok1) {
                    *reinterpret_cast<F2U*>(pf) = F2U{{vf[0], vf[1]}};
                    *reinterpret_cast<F2U*>(pg) = F2U{{vg[0], vg[1]}};
                } else {
                    if (ok0) { pf[0] = vf[0]; pg[0] = vg[0]; }
                    if (ok1) { pf[1] = vf[1]; pg[1] = vg[1]; }
                }
                // 16-bit hi/lo pairs of (sample 2c, sample 2c+1) -> one dword each in the [channel][time] tiles
                auto put = [&](int kind, const float* v) {
                    uint32_t hi, lo;
                    split2_bf16(v[0], v[1], hi, lo);
                    uint16_t* p = tt + (kind * 4 + g) * 1024 + t_wr[i];
                    *reinterpret_cast<uint32_t*>(p) = hi;
                    *reinterpret_cast<uint32_t*>(p + 512) = lo;
                };
                put(0, vf);
                put(1, vg);
                if (HAS_DY) put(2, vz);
            }
            load_cr(cr, pos_k(it + 2));
            __syncthreads();
        };
        for (int it = 0; it < n_items; it += 2) {
            r_body(it, crA, x1);
            r_body(it + 1, crB, x0);
        }
        return;
    }

    // =========================== W waves: operand fills, weight-gradient products ===========================
    f32x4 cfg[2][8], cd[4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < 8; ++n) cfg[h][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 4; ++n) cd[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct RawWO { f32x4 v[3][2]; };
    // [row][time] operands: row tile g of x(t-d), x(t) and dy; lane (row c, q) owns samples t0 + 8q .. + 7
    auto load_wo = [&](RawWO& r, Pos ps) {
#pragma unroll
        for (int kind = 0; kind < (HAS_DY ? 3 : 2); ++kind) {
            const float* base = (kind == 2 ? dy_or_x : a.x_in) + (size_t)ps.b * a.x_bstride;
            const float* p = ps.live ? base + (size_t)(16 * g + c) * a.pitch + ps.t0 + 8 * q + (kind == 0 ? -a.d : 0) : a.x_in;
            r.v[kind][0] = ld4u(p);
            r.v[kind][1] = ld4u(p + 4);
        }
    };
    auto fill_wo = [&](const RawWO& r, Pos ps, int stage) {
        uint16_t* wo = lds + (size_t)stage * RW_STAGE + RW_WO;
#pragma unroll
        for (int kind = 0; kind < (HAS_DY ? 3 : 2); ++kind) {
            float w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = r.v[kind][j >> 2][j & 3];
                if (kind == 2) {
                    const int t = ps.t0 + 8 * q + j;
                    if (t < a.t_lo || t >= a.t_hi) x = 0.f;
                }
                w[j] = x;
            }
            u32x4 fh, fl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t hi, lo;
                split2_bf16(w[2 * j], w[2 * j + 1], hi, lo);
                fh[j] = hi;
                fl[j] = lo;
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
        const uint16_t* wo = lds + (size_t)stage * RW_STAGE + RW_WO;
        const uint16_t* tt = lds + (size_t)stage * RW_STAGE + RW_T;
        Frag<BF16> adf, adg, az;
        load_tile(adf, tt, g);
        load_tile(adg, tt, 4 + g);
        if (HAS_DY) load_tile(az, tt, 8 + g);
        constexpr int NTILES = HAS_DY ? 12 : 8;
        Frag<BF16> bo[2];
        load_tile(bo[0], wo, 0);
#pragma unroll
        for (int nt = 0; nt < NTILES; ++nt) {
            if (nt + 1 < NTILES) load_tile(bo[(nt + 1) & 1], wo, nt + 1);
            if (nt < 8) {
                mma<BF16, 3>(cfg[0][nt], adf, bo[nt & 1]);
                mma<BF16, 3>(cfg[1][nt], adg, bo[nt & 1]);
            } else {
                mma<BF16, 3>(cd[nt - 8], az, bo[nt & 1]);
            }
        }
    };

    {
        // rx1 / rx0 hold the raw recompute rows of items it+1 / it+2, rw0 / rw1 the raw [row][time] rows of items
        // it / it+1; each set is re-armed two items ahead right after it was converted (see the R loop)
        RawD d0, d1;
        RawWO rw0, rw1;
        load_dy(d0, pos_k(0));
        load_wo(rw0, pos_k(0));
        load_dy(d1, pos_k(1));
        load_wo(rw1, pos_k(1));
        fill_dy(d0, pos_k(0), 0);
        load_dy(d0, pos_k(2));
        RawX x0, x1;
        __syncthreads();
        auto w_body = [&](const int it, RawD& rd, RawX& rx, RawWO& rw) {
            fill_dy(rd, pos_k(it + 1), (it + 1) & 1);        // recompute operands of the next item
            load_dy(rd, pos_k(it + 3));
            fill_wo(rw, pos_k(it), it & 1);                  // [row][time] operands of this item
            load_wo(rw, pos_k(it + 2));
            wgrad((it + 1) & 1);                             // products of the previous item
            __syncthreads();
        };
        for (int it = 0; it < n_items; it += 2) {
            w_body(it, d1, x1, rw0);
            w_body(it + 1, d0, x0, rw1);
        }
        wgrad(1);                                           // the last item of the (even) padded count
    }

    // ---- slab of this workgroup (every workgroup writes one, also an idle one: zeros)
    float* sfg = a.slab_fg + (size_t)wgid * (4 * CH * CH);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_nontemporal_store(cfg[h][nt][i], &sfg[(size_t)(h * CH + 16 * g + 4 * q + i) * (2 * CH) + nt * 16 + c]);
    if (HAS_DY && a.slab_d) {
        float* sd = a.slab_d + (size_t)wgid * (CH * CH);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_nontemporal_store(cd[r][i], &sd[(size_t)(r * 16 + c) * CH + 16 * g + 4 * q + i]);
    }
}

void wn_resrw_plan(int t_lo, int t_hi, int batch, int& t_base, int& steps, int& ipw, int& nwg) {
    // items start on multiples of 32 samples of absolute time = 128-byte lines of every row (pitch and bases
    // are multiples of 128 B): each 32-sample row segment an item reads or writes is exactly ONE cache line
    t_base = t_lo & ~(RW_COLS - 1);
    steps = (t_hi - t_base + RW_COLS - 1) / RW_COLS;
    const int total = steps * batch;
    ipw = (total + 255) / 256;
    if (ipw < 1) ipw = 1;
    nwg = (total + ipw - 1) / ipw;
    if (nwg < 1) nwg = 1;
}

int wn_launch_resblock_bwd_rw(const WnResMsArgs& a, int batch, hipStream_t st) {
    WnResMsArgs k = a;
    int nwg;
    wn_resrw_plan(a.t_lo, a.t_hi, batch, k.t_base, k.steps_per_clip, k.items_per_wg, nwg);
    k.batch = batch;
    k.swz = wn_xcd_swizzle_enabled();
    const size_t sh = (size_t)2 * RW_STAGE * sizeof(uint16_t);
    static WnDevOnce done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done.need(dev)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_bwd_rw_k<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_bwd_rw_k<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        done.done(dev);
    }
    if (k.dy && k.slab_d) hipLaunchKernelGGL(resblock_bwd_rw_k<true>, dim3(nwg), dim3(RW_THREADS), sh, st, k);
    else hipLaunchKernelGGL(resblock_bwd_rw_k<false>, dim3(nwg), dim3(RW_THREADS), sh, st, k);
    WN_CHECK_LAUNCH();
    return 0;
}

int wn_resms_slabs(int t_lo, int t_hi, int batch) {
    if (t_hi <= t_lo || batch <= 0) return 0;
    int tb, steps, ipw, nwg;
    wn_resrw_plan(t_lo, t_hi, batch, tb, steps, ipw, nwg);
    return nwg;
}

int wn_launch_resblock_bwd_ms(const WnResMsArgs& a, int ch, int batch, int mode_fwd, int mode_bwd, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    if (ch != RW_CH) return wn_set_error_msg(-3, "resblock_bwd_ms: 64 padded channels only");
    if (mode_fwd != WN_MODE_F16X3 || mode_bwd != WN_MODE_BF16X3)
        return wn_set_error_msg(-2, "resblock_bwd_ms: (f16x3, bf16x3) only");
    return wn_launch_resblock_bwd_rw(a, batch, st);
}
