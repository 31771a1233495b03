// Narrow channel-mixing product, "two-role" persistent form:  out[64 rows][t] = W [in0(t + shift0); in1(t + shift1)] (+ bias)
// (+ resid) for 8 k-steps of 32 input rows (the per-layer data gradient dx = W1^T dfg[t] + W0^T dfg[t+d] + dy).
//
// chan_gemm_k gives every wave 64 columns x all rows and one pass through "load - split - multiply - store":
// all waves of the launch are in the same phase at the same time (one generation of workgroups), so the
// load ramp and the final store burst are not covered by anything.  Here a workgroup is persistent and
// has two kinds of waves (same idea as resblock_bwd_rw_k):
//   * L waves (4..7) stream the fp32 input rows of the NEXT 32-column item from HBM (two items ahead in
//     registers), split them into 16-bit hi/lo MFMA fragments and leave them in LDS (2 stages x 32 KB);
//   * M waves (0..3; wave g owns output rows 16g..16g+15) keep their packed weights in registers (64),
//     multiply the current item out of LDS, add the residual rows and store.
// One barrier per item; two workgroups (16 waves, <= 128 registers) per CU; the workgroups of an XCD walk
// their range of items interleaved, so the shifted tap finds its rows in that XCD's L2.
// Preconditions (checked by the launcher, otherwise chan_gemm_k runs): x3 mode, 4 row tiles, 8 or 4 k-steps,
// no relu_in, out_shift 0 (bias, residual and mask are supported).  Columns outside [in_lo, in_hi) are never dereferenced (load addresses are
// clamped into the range, the values replaced by zeros).
#include <stdlib.h>
#include <type_traits>
#include "wn_common.h"
#include "wn_kernels.h"

#define GR_THREADS 512
#define GR_COLS 32

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) GrF2U { float v[2]; };
__device__ __forceinline__ f32x2 gr_ld2u(const float* p) {
    GrF2U u = *reinterpret_cast<const GrF2U*>(p);
    f32x2 r = {u.v[0], u.v[1]};
    return r;
}

struct GrPlan { int steps_per_clip, items_per_wg, batch; };

template <class T, int GR_KS>        // GR_KS = 8 (two taps of 128 rows: decoder / WaveNet blocks) or 4 (two taps of 64 rows: encoder blocks)
__global__ __launch_bounds__(GR_THREADS, 2) void chan_gemm_rw_k(WnGemmArgs a, GrPlan pl) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];        // 2 stages x 2 GR_KS fragments x 2 KB
    constexpr int STAGE = 2 * GR_KS * 1024;                                // halfs
    constexpr int PER = GR_KS / 4;                                         // k-steps per L wave

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

    struct Pos { int b, t0; bool live; };
    auto pos_k = [&](int k) {
        const bool live = k < n_items;
        k = k < n_items ? k : n_items - 1;
        int it = i_lo + (k < 0 ? 0 : k) * cnt;
        it = it < total ? it : total - 1;
        Pos p;
        p.b = it / pl.steps_per_clip;
        p.t0 = a.t_base + GR_COLS * (it - p.b * pl.steps_per_clip);
        p.live = live;
        return p;
    };

    if (wv < 4) {
        // =========================== M waves ===========================
        Frag<T> wa[GR_KS];
#pragma unroll
        for (int s = 0; s < GR_KS; ++s) load_a<T, 3>(wa[s], a.wpack, g * GR_KS + s, lane);
        float bias[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * g + 4 * q + i;
            bias[i] = (a.bias && row < a.m_valid) ? a.bias[row] : 0.f;
        }
        __syncthreads();
        for (int it = 0; it < n_items; ++it) {
            const Pos ps = pos_k(it);
            const int tl = ps.t0 + 2 * c;
            // residual rows of this item (consumed after the products)
            f32x2 rr[4];
            const float* rp = (a.resid ? a.resid : a.in0) + (size_t)ps.b * (a.resid ? a.resid_bstride : a.in_bstride) +
                              (size_t)(16 * g + 4 * q) * (a.resid ? a.resid_pitch : a.in_pitch) + tl;
#pragma unroll
            for (int i = 0; i < 4; ++i) rr[i] = gr_ld2u(rp + (size_t)i * (a.resid ? a.resid_pitch : a.in_pitch));
            // mask rows (keep where > 0; applied before the residual, as chan_gemm_k does)
            f32x2 mk[4];
            const float* mp = (a.mask ? a.mask : a.in0) + (size_t)ps.b * (a.mask ? a.mask_bstride : a.in_bstride) +
                              (size_t)(16 * g + 4 * q) * (a.mask ? a.mask_pitch : a.in_pitch) + tl;
#pragma unroll
            for (int i = 0; i < 4; ++i) mk[i] = gr_ld2u(mp + (size_t)i * (a.mask ? a.mask_pitch : a.in_pitch));

            const uint16_t* st = lds + (size_t)(it & 1) * STAGE;
            f32x4 acc[2];
            acc[0] = f32x4{bias[0], bias[1], bias[2], bias[3]};
            acc[1] = acc[0];
            Frag<T> bx[2];
            load_a<T, 3>(bx[0], st, 0, lane);
#pragma unroll
            for (int idx = 0; idx < 2 * GR_KS; ++idx) {
                if (idx + 1 < 2 * GR_KS) load_a<T, 3>(bx[(idx + 1) & 1], st, idx + 1, lane);
                mma<T, 3>(acc[idx & 1], wa[idx >> 1], bx[idx & 1]);
            }
            float* op = a.out + (size_t)ps.b * a.out_bstride + tl;
            const bool ok0 = tl >= a.t_lo && tl < a.t_hi, ok1 = tl + 1 >= a.t_lo && tl + 1 < a.t_hi;
            const bool r0 = a.resid && tl >= a.resid_lo, r1 = a.resid && tl + 1 >= a.resid_lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * g + 4 * q + i;
                float v0 = acc[0][i], v1 = acc[1][i];
                if (a.mask) {
                    v0 = mk[i][0] > 0.f ? v0 : 0.f;
                    v1 = mk[i][1] > 0.f ? v1 : 0.f;
                }
                if (r0) v0 += rr[i][0];
                if (r1) v1 += rr[i][1];
                float* o = op + (size_t)row * a.out_pitch;
                if (row < a.m_valid) {
                    if (ok0 && ok1) {
                        *reinterpret_cast<GrF2U*>(o) = GrF2U{{v0, v1}};
                    } else {
                        if (ok0) o[0] = v0;
                        if (ok1) o[1] = v1;
                    }
                }
            }
            __syncthreads();
        }
        return;
    }

    // =========================== L waves: k-steps PER*g .. PER*g + PER-1 ===========================
    struct Raw { f32x2 v[PER][8]; };
    auto load_raw = [&](Raw& r, Pos ps) {
        const int tl = ps.t0 + 2 * c;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int s = PER * g + u;
            const bool tap1 = s >= a.ks0;
            const float* base = (tap1 ? a.in1 : a.in0) + (size_t)ps.b * a.in_bstride;
            const int blk = tap1 ? s - a.ks0 : s;
            // Load address: the lane's column pair clamped INTO [in_lo, in_hi - 2] (columns outside the range are never
            // dereferenced - wn_chan_gemm's contract; fill() picks the right element / zero); a position past the end
            // (prefetch beyond the last item) reads ONE address in every lane: no traffic.  The loads stay
            // unconditional (a load under a run-time condition loses its prefetch)
            int cc = tl + (tap1 ? a.shift1 : a.shift0);
            cc = cc < a.in_lo ? a.in_lo : cc;
            cc = cc > a.in_hi - 2 ? a.in_hi - 2 : cc;
            const float* p = ps.live ? base + (size_t)(32 * blk + 8 * q) * a.in_pitch + cc : a.in0 + a.in_lo;
            const size_t rp = ps.live ? (size_t)a.in_pitch : 0;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) r.v[u][jj] = gr_ld2u(p + jj * rp);
        }
    };
    auto fill = [&](const Raw& r, Pos ps, int stage) {
        uint16_t* st = lds + (size_t)stage * STAGE;
        const int tl = ps.t0 + 2 * c;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int s = PER * g + u;
            const int sh = s >= a.ks0 ? a.shift1 : a.shift0;
            const int t0s = __builtin_amdgcn_readfirstlane(ps.t0) + sh;
            const bool inner = t0s >= a.in_lo && t0s + GR_COLS <= a.in_hi;       // wave-uniform
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = r.v[u][jj][n];
                if (!inner) {                                   // edge item: undo the address clamp of load_raw
                    const int col = tl + n + sh;
                    int cc = tl + sh;
                    cc = cc < a.in_lo ? a.in_lo : cc;
                    cc = cc > a.in_hi - 2 ? a.in_hi - 2 : cc;
                    const bool ok = col >= a.in_lo && col < a.in_hi;
                    const bool second = col - cc == 1;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) v[jj] = ok ? (second ? r.v[u][jj][1] : r.v[u][jj][0]) : 0.f;
                }
                u32x4 fh, fl;
                if (std::is_same<T, BF16>::value) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const f32x2 pv = {v[2 * jj], v[2 * jj + 1]};
                        const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(pv, bf16x2));
                        const f32x2 rv = {pv[0] - __builtin_bit_cast(float, hi << 16),
                                          pv[1] - __builtin_bit_cast(float, hi & 0xffff0000u)};
                        fh[jj] = hi;
                        fl[jj] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rv, bf16x2));
                    }
                } else {
                    Frag<T> f;
                    split8<T, 3>(f, v);
                    fh = __builtin_bit_cast(u32x4, f.hi);
                    fl = __builtin_bit_cast(u32x4, f.lo);
                }
                u32x4* p = reinterpret_cast<u32x4*>(st) + (size_t)(s * 2 + n) * 128 + lane;
                p[0] = fh;
                p[64] = fl;
            }
        }
    };

    // r1 / r0 hold the raw rows of items it+1 / it+2 (loop unrolled by two: each set is re-armed two items
    // ahead right after its conversion, no register copies; an odd count is padded with a clamped duplicate
    // whose fragments nobody multiplies)
    Raw r0, r1;
    load_raw(r0, pos_k(0));
    load_raw(r1, pos_k(1));
    fill(r0, pos_k(0), 0);
    load_raw(r0, pos_k(2));
    __syncthreads();
    for (int it = 0; it < n_items; it += 2) {
        fill(r1, pos_k(it + 1), 1);
        load_raw(r1, pos_k(it + 3));
        __syncthreads();
        if (it + 1 < n_items) {
            fill(r0, pos_k(it + 2), 0);
            load_raw(r0, pos_k(it + 4));
            __syncthreads();
        }
    }
}

// returns 1 if the launch was taken, 0 if the arguments are outside this kernel's preconditions
int wn_launch_gemm_rw(const WnGemmArgs& k, int batch, int mode, hipStream_t st) {
    if (mode != WN_MODE_BF16X3 && mode != WN_MODE_F16X3) return 0;
    const int ks = k.ks0 + k.ks1;
    if (k.mt != 4 || (ks != 8 && ks != 4) || (k.ks1 > 0 && !k.in1) || k.relu_in || k.out_shift != 0) return 0;
    if (k.t_base & (GR_COLS - 1)) return 0;
    if (k.in_hi - k.in_lo < 2) return 0;
    GrPlan pl;
    pl.batch = batch;
    pl.steps_per_clip = (k.t_hi - k.t_base + GR_COLS - 1) / GR_COLS;
    const int total = pl.steps_per_clip * batch;
    pl.items_per_wg = (total + 511) / 512;
    if (pl.items_per_wg < 1) pl.items_per_wg = 1;
    const int nwg = (total + pl.items_per_wg - 1) / pl.items_per_wg;
    const size_t sh = (size_t)2 * 2 * ks * 1024 * sizeof(uint16_t);
    static WnDevOnce done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done.need(dev)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chan_gemm_rw_k<BF16, 8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chan_gemm_rw_k<F16, 8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        done.done(dev);
    }
    const dim3 gr(nwg), bl(GR_THREADS);
    if (mode == WN_MODE_BF16X3) {
        if (ks == 8) hipLaunchKernelGGL((chan_gemm_rw_k<BF16, 8>), gr, bl, sh, st, k, pl);
        else hipLaunchKernelGGL((chan_gemm_rw_k<BF16, 4>), gr, bl, sh, st, k, pl);
    } else {
        if (ks == 8) hipLaunchKernelGGL((chan_gemm_rw_k<F16, 8>), gr, bl, sh, st, k, pl);
        else hipLaunchKernelGGL((chan_gemm_rw_k<F16, 4>), gr, bl, sh, st, k, pl);
    }
    return 1;
}
