// Forward of one gated residual block, "two-role" persistent form (CH = 64 padded channels, x3 modes).
// Same contract as resblock_fwd_nt_k (wn_resblock2.hip): [f;g] = Wfg [x(t-d); x(t)] (+ bias, + cond),
// z = tanh f * sigmoid g (stored on [z_lo, t_hi)), x_out = Wd z (+ bias) + x(t) on [t_lo, t_hi).
//
// resblock_fwd_nt_k gives a wave 64 columns x all channels and ONE pass "load - split - multiply - gate -
// multiply - store"; all waves of the launch are in the same phase at the same time, so the load ramp and
// the store burst of a launch are not covered by anything (timing builds come apart additively).  Here a
// workgroup is persistent over 32-column items and its waves have two jobs (as in resblock_bwd_rw_k):
//   * G waves (0..3; wave g owns dilation channels 16g..16g+15): packed f/g weights in registers (64),
//     both N-tiles of f and g out of the LDS x fragments, gate, z to HBM (fp32) and to LDS (16-bit hi/lo
//     fragments in the chained k order of the dense product);
//   * D waves (4..7; wave g owns residual rows 16g..16g+15): stream the raw x rows two items ahead, split
//     them into the x fragments of the NEXT item, and do the dense product + residual + x_out store of the
//     PREVIOUS item out of the z fragments (Wd row tile in registers, 16).
// One barrier per item, LDS 2 x (16 KB x + 8 KB z), <= 128 registers: two workgroups (16 waves) per CU.
// MEASURED SLOWER than resblock_fwd_nt_k at config 2 (32.5 vs 29 us per block on the same box; deeper
// prefetch or one workgroup per CU do not change that): the G waves carry the MFMAs AND the gate while the
// D waves have little to do, so the role split buys no overlap here, and the LDS round trips and the
// per-item barrier are extra.  Opt-in (WN_FWD_RW=1), kept correct by the switch tests.
#include <stdlib.h>
#include "wn_common.h"
#include "wn_kernels.h"

#define FR_THREADS 512
#define FR_COLS 32
#define FR_XF 0
#define FR_Z 8192            // halfs: 8 x fragments, then 4 z fragments
#define FR_STAGE 12288
#ifndef FR_DEPTH
#define FR_DEPTH 2         // even (4 and 8 measured slower: 1.02-1.10 vs 0.97 ms per 30 blocks)
#endif
#ifndef FR_WGS_PER_CU
#define FR_WGS_PER_CU 2
#endif
#if FR_WGS_PER_CU == 2
#define FR_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define FR_OCC
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) FrF2U { float v[2]; };
__device__ __forceinline__ f32x2 fr_ld2u(const float* p) {
    FrF2U u = *reinterpret_cast<const FrF2U*>(p);
    f32x2 r = {u.v[0], u.v[1]};
    return r;
}
// z = tanh(f) * sigmoid(g) with one reciprocal (same formula as resblock_fwd_nt_k)
__device__ __forceinline__ float fr_gate(float f, float g) {
    f = fminf(fmaxf(f, -15.f), 15.f);
    const float e1 = __expf(-2.0f * f), e2 = __expf(-g);
    return (1.0f - e1) * __builtin_amdgcn_rcpf((1.0f + e1) * (1.0f + e2));
}

struct FrPlan { int steps_per_clip, items_per_wg, batch; };

template <class T>
__global__ __launch_bounds__(FR_THREADS) FR_OCC void resblock_fwd_rw_k(WnResArgs a, FrPlan pl) {
    constexpr int CH = 64;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wv & 3;
    const int c = lane & 15, q = lane >> 4;

    // items of this workgroup: interleaved within the XCD's contiguous range (see resblock_bwd_rw_k)
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
    const int total = pl.steps_per_clip * pl.batch;
    const int i_lo = first * pl.items_per_wg + j;
    int i_hi = (first + cnt) * pl.items_per_wg;
    if (i_hi > total) i_hi = total;
    const int n_items = i_lo < i_hi ? (i_hi - i_lo + cnt - 1) / cnt : 0;

    struct Pos { int b, t0; };
    auto pos_k = [&](int k) {
        k = k < n_items ? k : n_items - 1;
        int it = i_lo + (k < 0 ? 0 : k) * cnt;
        it = it < total ? it : total - 1;
        Pos p;
        p.b = it / pl.steps_per_clip;
        p.t0 = a.t_base + FR_COLS * (it - p.b * pl.steps_per_clip);
        return p;
    };

    if (wv < 4) {
        // =========================== G waves: f, g, gate ===========================
        Frag<T> wf[4], wg[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            load_a<T, 3>(wf[s], a.wfg, g * 4 + s, lane);
            load_a<T, 3>(wg[s], a.wfg, (4 + g) * 4 + s, lane);
        }
        float bias_f[4], bias_g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * g + 4 * q + i;
            bias_f[i] = (a.bias_f && row < a.n_f) ? a.bias_f[row] : 0.f;
            bias_g[i] = (a.bias_g && row < a.n_f) ? a.bias_g[row] : 0.f;
        }
        __syncthreads();                                    // stage 0 fragments of the first item are in LDS
        for (int it = 0; it < n_items; ++it) {
            const Pos ps = pos_k(it);
            const int tl = ps.t0 + 2 * c;
            uint16_t* st = lds + (size_t)(it & 1) * FR_STAGE;
            const uint16_t* xf = st + FR_XF;
            f32x4 af[2], ag[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                af[n] = f32x4{bias_f[0], bias_f[1], bias_f[2], bias_f[3]};
                ag[n] = f32x4{bias_g[0], bias_g[1], bias_g[2], bias_g[3]};
            }
            {
                Frag<T> bx[2];
                load_a<T, 3>(bx[0], xf, 0, lane);
#pragma unroll
                for (int idx = 0; idx < 8; ++idx) {
                    if (idx + 1 < 8) load_a<T, 3>(bx[(idx + 1) & 1], xf, idx + 1, lane);
                    mma<T, 3>(af[idx & 1], wf[idx >> 1], bx[idx & 1]);
                    mma<T, 3>(ag[idx & 1], wg[idx >> 1], bx[idx & 1]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int idx = 0; idx < 7; ++idx) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
            if (a.cond) {       // per-(channel, time-bucket) conditioning bias (wavenet_autoencoder/model1.py:183,227-247)
                const float* cb = a.cond + (size_t)ps.b * a.cond_bstride;
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
            float z[2][4];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) z[n][i] = fr_gate(af[n][i], ag[n][i]);
            // z-crop store (rows 16g+4q+i, samples tl, tl+1)
            {
                float* zo = a.z_out + (size_t)ps.b * a.z_bstride + (size_t)(16 * g + 4 * q) * a.pitch + tl;
                const bool ok0 = tl >= a.z_lo && tl < a.t_hi, ok1 = tl + 1 >= a.z_lo && tl + 1 < a.t_hi;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float* o = zo + (size_t)i * a.pitch;
                    if (ok0 && ok1) {
                        *reinterpret_cast<FrF2U*>(o) = FrF2U{{z[0][i], z[1][i]}};
                    } else {
                        if (ok0) o[0] = z[0][i];
                        if (ok1) o[1] = z[1][i];
                    }
                }
            }
            // z as B fragments of the dense product, chained k order: this wave supplies slots 4(g&1).. of k-step g>>1
            if (a.write_x) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    typename T::elem h[4], l[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        h[i] = T::cvt(z[n][i]);
                        l[i] = T::cvt(z[n][i] - T::back(h[i]));
                    }
                    uint16_t* fb = st + FR_Z + (size_t)((g >> 1) * 2 + n) * 1024 + lane * 8 + (g & 1) * 4;
                    auto pk = [](typename T::elem x, typename T::elem y) {
                        return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
                    };
                    *reinterpret_cast<uint2*>(fb) = uint2{pk(h[0], h[1]), pk(h[2], h[3])};
                    *reinterpret_cast<uint2*>(fb + 512) = uint2{pk(l[0], l[1]), pk(l[2], l[3])};
                }
            }
            __syncthreads();
        }
        return;
    }

    // =========================== D waves: x fragments of the next item, dense product of the previous one ===========================
    Frag<T> wd[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) load_a<T, 3>(wd[s], a.wd, g * 2 + s, lane);
    float bias_d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 16 * g + 4 * q + i;
        bias_d[i] = (a.bias_d && row < a.n_d) ? a.bias_d[row] : 0.f;
    }
    struct RawX { f32x2 x[8]; };
    struct RawR { f32x2 r[4]; };
    // wave g converts k-step g = (tap g>>1, channel half g&1); lane (c, q) of N-tile n holds sample t0 + 2c + n
    auto load_x = [&](RawX& r, Pos ps) {
        const int tl = ps.t0 + 2 * c;
        const float* p = a.x_in + (size_t)ps.b * a.x_bstride + (size_t)(32 * (g & 1) + 8 * q) * a.pitch +
                         ((g >> 1) == 0 ? tl - a.d : tl);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) r.x[jj] = fr_ld2u(p + (size_t)jj * a.pitch);
    };
    auto fill_x = [&](const RawX& r, int stage) {
        uint16_t* xf = lds + (size_t)stage * FR_STAGE + FR_XF;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float v[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[jj] = r.x[jj][n];
            Frag<T> f;
            split8<T, 3>(f, v);
            u32x4* p = reinterpret_cast<u32x4*>(xf) + (size_t)(g * 2 + n) * 128 + lane;
            p[0] = __builtin_bit_cast(u32x4, f.hi);
            p[64] = __builtin_bit_cast(u32x4, f.lo);
        }
    };
    auto load_res = [&](RawR& r, Pos ps) {           // residual rows 16g+4q+i of x(t), samples tl, tl+1
        const float* p = a.x_in + (size_t)ps.b * a.x_bstride + (size_t)(16 * g + 4 * q) * a.pitch + ps.t0 + 2 * c;
#pragma unroll
        for (int i = 0; i < 4; ++i) r.r[i] = fr_ld2u(p + (size_t)i * a.pitch);
    };
    auto dense = [&](const RawR& rr, Pos ps, int stage, bool live) {
        const uint16_t* zf = lds + (size_t)stage * FR_STAGE + FR_Z;
        f32x4 ad[2];
        ad[0] = f32x4{bias_d[0], bias_d[1], bias_d[2], bias_d[3]};
        ad[1] = ad[0];
#pragma unroll
        for (int idx = 0; idx < 4; ++idx) {
            Frag<T> bz;
            load_a<T, 3>(bz, zf, idx, lane);
            mma<T, 3>(ad[idx & 1], wd[idx >> 1], bz);
        }
        const int tl = ps.t0 + 2 * c;
        float* xo = a.x_out + (size_t)ps.b * a.x_bstride + (size_t)(16 * g + 4 * q) * a.pitch + tl;
        const bool ok0 = live && tl >= a.t_lo && tl < a.t_hi, ok1 = live && tl + 1 >= a.t_lo && tl + 1 < a.t_hi;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v0 = ad[0][i] + rr.r[i][0], v1 = ad[1][i] + rr.r[i][1];
            float* o = xo + (size_t)i * a.pitch;
            if (ok0 && ok1) {
                *reinterpret_cast<FrF2U*>(o) = FrF2U{{v0, v1}};
            } else {
                if (ok0) o[0] = v0;
                if (ok1) o[1] = v1;
            }
        }
    };

    // FR_DEPTH register sets of raw x rows: set (k mod FR_DEPTH) holds item k, re-armed FR_DEPTH items ahead right
    // after its conversion (the loop is unrolled by FR_DEPTH so the set index is static: no copies).  A persistent
    // workgroup only has as many bytes in flight as its prefetch registers hold, and an HBM miss under load
    // is several items long.  rr1 / rr0: residual rows of items it-1 / it.
    RawX xs[FR_DEPTH];
    RawR rr0, rr1;
    load_x(xs[0], pos_k(0));
#pragma unroll
    for (int k = 1; k < FR_DEPTH; ++k) load_x(xs[k], pos_k(k));
    load_res(rr0, pos_k(0));
    load_res(rr1, pos_k(1));
    fill_x(xs[0], 0);
    load_x(xs[0], pos_k(FR_DEPTH));
    __syncthreads();
    auto d_body = [&](const int it, RawX& rx, RawR& rr) {       // rx: item it+1, rr: item it-1
        fill_x(rx, (it + 1) & 1);
        load_x(rx, pos_k(it + 1 + FR_DEPTH));
        if (a.write_x) {
            dense(rr, pos_k(it - 1), (it + 1) & 1, it >= 1 && it <= n_items);
            load_res(rr, pos_k(it + 1));
        }
        if (it < n_items) __syncthreads();
    };
    for (int it = 0; it <= n_items; it += FR_DEPTH) {
#pragma unroll
        for (int u = 0; u < FR_DEPTH; ++u) {
            if (u & 1) d_body(it + u, xs[(u + 1) % FR_DEPTH], rr0);
            else d_body(it + u, xs[(u + 1) % FR_DEPTH], rr1);
        }
    }
}

static int fr_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("WN_FWD_RW"); v = e ? (atoi(e) != 0) : 0; }
    return v;
}

// returns 1 if the launch was taken, 0 if the arguments are outside this kernel's preconditions
int wn_launch_resblock_fwd_rw(const WnResArgs& a, int ch, int batch, int mode, hipStream_t st) {
    if (!fr_enabled()) return 0;
    if (ch != 64 || (mode != WN_MODE_F16X3 && mode != WN_MODE_BF16X3)) return 0;
    WnResArgs k = a;
    k.t_base = wn_tile_origin(a.t_lo);
    if (k.t_base & (FR_COLS - 1)) return 0;
    k.swz = wn_xcd_swizzle_enabled();
    FrPlan pl;
    pl.batch = batch;
    pl.steps_per_clip = (a.t_hi - k.t_base + FR_COLS - 1) / FR_COLS;
    const int total = pl.steps_per_clip * batch;
    pl.items_per_wg = (total + 256 * FR_WGS_PER_CU - 1) / (256 * FR_WGS_PER_CU);
    if (pl.items_per_wg < 1) pl.items_per_wg = 1;
    const int nwg = (total + pl.items_per_wg - 1) / pl.items_per_wg;
    const size_t sh = (size_t)2 * FR_STAGE * sizeof(uint16_t);
    if (mode == WN_MODE_F16X3) hipLaunchKernelGGL(resblock_fwd_rw_k<F16>, dim3(nwg), dim3(FR_THREADS), sh, st, k, pl);
    else hipLaunchKernelGGL(resblock_fwd_rw_k<BF16>, dim3(nwg), dim3(FR_THREADS), sh, st, k, pl);
    return 1;
}
