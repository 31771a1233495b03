// Forward of one gated residual block, channel-split form (CH = 64, F16x3): the counterpart of
// wn_resms.hip for wavenet/model.py:111-129.
//
// In resblock_fwd_nt_k a wave owns 64 columns x all channels and a CU runs one 8-wave workgroup, so a
// launch lasts as long as ONE wave's dependency chain (loads -> 384 MFMAs -> 64 gates per lane -> 96
// MFMAs -> stores: 27 us per block at config 2, whatever is removed from it comes off one for one).
// Here the 4 waves of a workgroup split the CHANNELS of the same 64 columns:
//   * wave g owns dilation channels 16g..16g+15 (row tile g of f and of g) and residual channels
//     16g..16g+15 of the dense 1x1; the 10 weight fragments it needs (80 registers) never move;
//   * x(t-d), x(t) are fetched and split into f16 hi/lo fragments once per workgroup (wave g does
//     k-step g) and shared through LDS, two stages deep, the next item's rows in flight in registers;
//   * z goes from the gate straight into the B-fragment order of the dense product (each wave writes
//     its 8-byte pieces of the two z k-steps into the LDS stage the recompute has just finished with);
//   * a wave's chain per 64-column item is a quarter of the old one, a workgroup walks its items back
//     to back (persistent), and with 64 KB of LDS and <= 256 registers TWO workgroups share a CU.
#include "wn_common.h"
#include "wn_kernels.h"

#define CS_THREADS 256
#define CS_CH 64

template <class T>
__device__ __forceinline__ void cs_store_frag(uint16_t* base, int idx, int lane, const Frag<T>& f) {
    u32x4* p = reinterpret_cast<u32x4*>(base) + (size_t)idx * 128 + lane;
    p[0] = __builtin_bit_cast(u32x4, f.hi);
    p[64] = __builtin_bit_cast(u32x4, f.lo);
}

__device__ __forceinline__ float cs_gate(float f, float g) {        // same arithmetic as wn_gate (wn_resblock2.hip)
    f = fminf(fmaxf(f, -15.f), 15.f);
    const float e1 = __expf(-2.0f * f), e2 = __expf(-g);
    return (1.0f - e1) * __builtin_amdgcn_rcpf((1.0f + e1) * (1.0f + e2));
}

template <bool WRITE_X>
__global__ __launch_bounds__(CS_THREADS, 2) void resblock_fwd_cs_k(WnResArgs a, int steps_per_clip, int items_per_wg, int batch) {
    constexpr int CH = CS_CH;
    constexpr int FR = 1024;                                   // halfs per x3 fragment
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];       // 2 stages x 16 fragments (32 KB each)

    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, q = lane >> 4;

    Frag<F16> wf[4], wg[4], wd[2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        load_a<F16, 3>(wf[s], a.wfg, g * 4 + s, lane);
        load_a<F16, 3>(wg[s], a.wfg, (4 + g) * 4 + s, lane);
    }
    if (WRITE_X) {
#pragma unroll
        for (int s = 0; s < 2; ++s) load_a<F16, 3>(wd[s], a.wd, g * 2 + s, lane);
    }
    float bias_f[4], bias_g[4], bias_d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 16 * g + 4 * q + i;
        bias_f[i] = (a.bias_f && row < a.n_f) ? a.bias_f[row] : 0.f;
        bias_g[i] = (a.bias_g && row < a.n_f) ? a.bias_g[row] : 0.f;
        bias_d[i] = (a.bias_d && row < a.n_d) ? a.bias_d[row] : 0.f;
    }

    const WnBlock blk = wn_block(a.swz);
    const int wgid = blk.x;
    const int total = steps_per_clip * batch;
    const int item0 = wgid * items_per_wg;
    int item_end = item0 + items_per_wg;
    if (item_end > total) item_end = total;
    if (item0 >= item_end) return;

    struct Pos { int b, t0; };
    auto pos_of = [&](int it) {
        it = it < item_end ? it : item_end - 1;
        Pos p;
        p.b = it / steps_per_clip;
        p.t0 = a.t_base + 64 * (it - p.b * steps_per_clip);
        return p;
    };
    auto next_pos = [&](Pos p, int it_next) {                 // position of the next item, clamped to the last one
        if (it_next >= item_end) return p;
        p.t0 += 64;
        if (p.t0 >= a.t_base + 64 * steps_per_clip) { p.t0 = a.t_base; p.b += 1; }
        return p;
    };
    // raw rows of k-step g = (tap g>>1, channel half g&1); unconditional, alignment-free loads
    auto load_x = [&](f32x4* r, Pos ps) {
        const int tl = ps.t0 + 4 * c;
        const float* p = a.x_in + (size_t)ps.b * a.x_bstride + (size_t)(32 * (g & 1) + 8 * q) * a.pitch + ((g >> 1) == 0 ? tl - a.d : tl);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = ld4u(p + (size_t)j * a.pitch);
    };
    auto fill_x = [&](const f32x4* r, int stage) {
        uint16_t* xf = lds + (size_t)stage * 16 * FR;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = r[j][n];
            Frag<F16> f;
            split8<F16, 3>(f, v);
            cs_store_frag<F16>(xf, g * 4 + n, lane, f);
        }
    };

    f32x4 rx[8];
    Pos p_cur = pos_of(item0);
    Pos p_n1 = next_pos(p_cur, item0 + 1);
    load_x(rx, p_cur);
    fill_x(rx, item0 & 1);
    load_x(rx, p_n1);
    for (int item = item0; item < item_end; ++item) {
        const Pos p_n2 = next_pos(p_n1, item + 2);
        const int b = p_cur.b, tl = p_cur.t0 + 4 * c;
        __syncthreads();                    // A: stage item&1 is filled; the other stage is free again
        // residual rows of this item (used at the very end), then the next item's stage and loads
        f32x4 res[4];
        if (WRITE_X) {
            const float* xr = a.x_in + (size_t)b * a.x_bstride + (size_t)(16 * g + 4 * q) * a.pitch + tl;
#pragma unroll
            for (int i = 0; i < 4; ++i) res[i] = ld4(xr + (size_t)i * a.pitch);
        }
        fill_x(rx, (item + 1) & 1);
        load_x(rx, p_n2);

        // ---- f, g of channels 16g.. (weight fragments in registers, x fragments from LDS)
        uint16_t* xf = lds + (size_t)(item & 1) * 16 * FR;
        f32x4 af[4], ag[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            af[n] = f32x4{bias_f[0], bias_f[1], bias_f[2], bias_f[3]};
            ag[n] = f32x4{bias_g[0], bias_g[1], bias_g[2], bias_g[3]};
        }
        {
            Frag<F16> bx[2];
            load_a<F16, 3>(bx[0], xf, 0, lane);
#pragma unroll
            for (int idx = 0; idx < 16; ++idx) {
                if (idx + 1 < 16) load_a<F16, 3>(bx[(idx + 1) & 1], xf, idx + 1, lane);
                mma<F16, 3>(af[idx & 3], wf[idx >> 2], bx[idx & 1]);
                mma<F16, 3>(ag[idx & 3], wg[idx >> 2], bx[idx & 1]);
            }
            // fragment i+1 is read from LDS while the matrix core works on fragment i (pinned order)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int idx = 0; idx < 15; ++idx) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
        if (a.cond) {       // per-(channel, time-bucket) conditioning bias (wavenet_autoencoder/model1.py:183)
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
        // ---- gate, z store
        f32x4 z[4];                                            // [n][i]
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) z[n][i] = cs_gate(af[n][i], ag[n][i]);
        {
            float* zo = a.z_out + (size_t)b * a.z_bstride + (size_t)(16 * g + 4 * q) * a.pitch + tl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = {z[0][i], z[1][i], z[2][i], z[3][i]};
                st4m(zo + (size_t)i * a.pitch, v, tl, a.z_lo, a.t_hi);
            }
        }
        if (!WRITE_X) {                     // last block: its residual output is unused
            p_cur = p_n1;
            p_n1 = p_n2;
            continue;
        }

        // ---- z as B fragments of the dense product (chained k order: k-step g>>1, slots 4(g&1)..+3),
        //      into the stage every wave has finished reading
        __syncthreads();                    // B
        {
            uint16_t* zf = xf + (size_t)((g >> 1) * 4) * FR + lane * 8 + (g & 1) * 4;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                uint16_t hh[4], ll[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const _Float16 hv = F16::cvt(z[n][i]);
                    hh[i] = __builtin_bit_cast(uint16_t, hv);
                    ll[i] = __builtin_bit_cast(uint16_t, F16::cvt(z[n][i] - F16::back(hv)));
                }
                *reinterpret_cast<uint2*>(zf + (size_t)n * FR) = uint2{(uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16)};
                *reinterpret_cast<uint2*>(zf + (size_t)n * FR + 512) = uint2{(uint32_t)ll[0] | ((uint32_t)ll[1] << 16), (uint32_t)ll[2] | ((uint32_t)ll[3] << 16)};
            }
        }
        __syncthreads();                    // C
        // ---- dense 1x1: residual channels 16g.. ; x' = Wd z + x
        f32x4 acc2[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc2[n] = f32x4{bias_d[0], bias_d[1], bias_d[2], bias_d[3]};
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                Frag<F16> bz;
                load_a<F16, 3>(bz, xf, s * 4 + n, lane);
                mma<F16, 3>(acc2[n], wd[s], bz);
            }
        {
            float* xo = a.x_out + (size_t)b * a.x_bstride + (size_t)(16 * g + 4 * q) * a.pitch + tl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = {acc2[0][i] + res[i][0], acc2[1][i] + res[i][1], acc2[2][i] + res[i][2], acc2[3][i] + res[i][3]};
                st4m(xo + (size_t)i * a.pitch, v, tl, a.t_lo, a.t_hi);
            }
        }
        p_cur = p_n1;
        p_n1 = p_n2;
    }
}

int wn_launch_resblock_fwd_cs(const WnResArgs& a, int batch, hipStream_t st) {
    WnResArgs k = a;
    k.swz = wn_xcd_swizzle_enabled();
    k.t_base = a.t_lo & ~3;
    const int steps = (a.t_hi - k.t_base + 63) / 64;
    const int total = steps * batch;
    static int wgs = -1;
    if (wgs < 0) { const char* e = getenv("WN_FWD_CS_WGS"); wgs = e ? atoi(e) : 512; }
    int ipw = (total + wgs - 1) / wgs;      // 512: two workgroups per CU
    if (ipw < 1) ipw = 1;
    const int nwg = (total + ipw - 1) / ipw;
    const size_t sh = (size_t)2 * 16 * 1024 * sizeof(uint16_t);
    static unsigned long long done = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_fwd_cs_k<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_fwd_cs_k<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        done |= 1ull << dev;
    }
    if (a.write_x) hipLaunchKernelGGL(resblock_fwd_cs_k<true>, dim3(nwg), dim3(CS_THREADS), sh, st, k, steps, ipw, batch);
    else hipLaunchKernelGGL(resblock_fwd_cs_k<false>, dim3(nwg), dim3(CS_THREADS), sh, st, k, steps, ipw, batch);
    WN_CHECK_LAUNCH();
    return 0;
}
