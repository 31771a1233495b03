// Fused gated residual block, forward (see wn_resblock.hip for the algorithm), templated on the number NT of
// 16-column N-tiles a wave owns.  NT = 4 is what runs: 8 waves x 64 columns (float4 per lane), 512 columns per
// workgroup, one copy of the packed weights in LDS.  (NT = 2, 16 waves x 32 columns, measured 33 vs 26.6 us per
// config-2 block and is not instantiated.)
#include <stdlib.h>
#include <type_traits>
#include "wn_common.h"
#include "wn_kernels.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) F2U { float v[2]; };
template <int NT> struct VecN;
template <> struct VecN<4> {
    typedef f32x4 t;
    static __device__ __forceinline__ t ld(const float* p) { return ld4(p); }
    static __device__ __forceinline__ t ldu(const float* p) { return ld4u(p); }
    static __device__ __forceinline__ void stm(float* p, t v, int tt, int lo, int hi) { st4m(p, v, tt, lo, hi); }
};
template <> struct VecN<2> {
    typedef f32x2 t;
    static __device__ __forceinline__ t ld(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
    static __device__ __forceinline__ t ldu(const float* p) {
        F2U u = *reinterpret_cast<const F2U*>(p);
        t r = {u.v[0], u.v[1]};
        return r;
    }
    static __device__ __forceinline__ void stm(float* p, t v, int tt, int lo, int hi) {
        if (tt >= lo && tt + 1 < hi) *reinterpret_cast<f32x2*>(p) = v;
        else {
            if (tt >= lo && tt < hi) p[0] = v[0];
            if (tt + 1 >= lo && tt + 1 < hi) p[1] = v[1];
        }
    }
};

// z = tanh(f) * sigmoid(g) with ONE reciprocal: (1 - e1) / ((1 + e1)(1 + e2)), e1 = exp(-2f), e2 = exp(-g).
// Absolute error ~1e-7 (the cancellation in 1 - e1 only costs RELATIVE accuracy near f = 0).
__device__ __forceinline__ float wn_gate(float f, float g) {
    f = fminf(fmaxf(f, -15.f), 15.f);
    const float e1 = __expf(-2.0f * f), e2 = __expf(-g);
    return (1.0f - e1) * __builtin_amdgcn_rcpf((1.0f + e1) * (1.0f + e2));
}

template <int NT> struct NtCfg { static constexpr int WAVES = NT == 4 ? 8 : 16; };     // 512 columns per workgroup either way

// ENC = the autoencoder's ENCODER block (wavenet_autoencoder/model1.py:137-152) on the same skeleton:
//   h = Wdil [relu x(t-d); relu x(t)] (+ bias) ; x_out = Wd relu(h) (+ bias) + x(t) ; h (pre-activation) is stored where
//   the decoder block stores z.  One row group (no gate), ReLU on load and in front of the dense product.
template <class T, int NS, int CH, int NT, bool ENC = false, int WV = NtCfg<NT>::WAVES, bool CND = false>
__global__ __launch_bounds__(64 * WV) void resblock_fwd_nt_k(WnResArgs a) {
    static_assert(!CND || (NT == 4 && NS == 3 && !ENC), "conditioning k-step: 4 samples per lane, x3 packs");
    typedef typename VecN<NT>::t fvec;
    constexpr int THREADS = 64 * WV;
    constexpr int COLS = WV * 16 * NT;
    constexpr int MT = (ENC ? 1 : 2) * CH / 16;        // fg row tiles (f rows then g rows); encoder: h rows only
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
    const int t0 = a.t_base + blk.x * COLS + wave * (16 * NT);
    const int tl = t0 + NT * c;

    const float* xin = a.x_in + (size_t)b * a.x_bstride;
    // tap-0 column.  Lanes that own at least one valid output have tl - d >= -2 (t_lo >= d + 1);
    // every activation buffer is allocated with >= 64 floats of slack in front and >= 256 behind,
    // so the (masked-out) garbage columns are still addressable.
    const int colm = tl - a.d;

    fvec raw[8];
    auto issue = [&](int s) {
        const int tap = s / KT, ch = (s % KT) * 32 + 8 * q;
        const float* p = xin + (size_t)ch * a.pitch + (tap == 0 ? colm : tl);
        // the shifted tap is always loaded with the alignment-free form: choosing between the two forms
        // at run time would merge the loaded registers across a branch, which the compiler implements as
        // load -> wait -> copy, i.e. no prefetch at all
        if (tap == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = VecN<NT>::ldu(p + (size_t)j * a.pitch);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = VecN<NT>::ld(p + (size_t)j * a.pitch);
        }
    };
    issue(0);      // first activation loads are in flight while the weights are staged
    // CND: buckets of the lane's 4 samples (columns beyond t_hi: any valid bytes, their results are masked)
    uint32_t cbytes = 0;
    if (CND) {
        struct U32U { uint32_t v; } __attribute__((packed, aligned(1)));
        int off = tl - a.t_lo;
        off = off < a.t_hi - a.t_lo ? off : a.t_hi - a.t_lo;
        cbytes = reinterpret_cast<const U32U*>(a.cond_idx + (WN_PQ_IDX_PAD + off))->v;
    }

    // The packed weights go to LDS one k-step at a time: only the MT fragments of k-step 0 are waited for before the first
    // MFMA; the fragments of k-step s + 1 (after the last one: the dense product's) are fetched from L2 while k-step s is
    // multiplied and written behind it, one workgroup barrier per k-step.  (All 80 KB up front held every wave at the first
    // barrier for 4 of the launch's 24 us: in-kernel stamps, tools/fw_spans.py on a -DFW_DBG build.)
    constexpr int FRV = FR / 8;                               // u32x4 per fragment
    constexpr int WPT = (MT * FRV + THREADS - 1) / THREADS;   // u32x4 per thread for one k-step's fg fragments
    constexpr int DPT = (ND * FRV + THREADS - 1) / THREADS;   // ... for the dense product's fragments
    constexpr int SPT = WPT > DPT ? WPT : DPT;
    u32x4 wst[SPT];
    // CND: s == KS are the MT fragments of this clip's conditioning table (they take the place of k-step 0's in LDS), the
    // dense fragments follow as s == KS + 1
    auto stage_ld = [&](int s) {          // s < KS: fg fragments (m, s), m = 0..MT-1 ; s == KS: the dense fragments
        if (CND && s == KS) {
            const u32x4* src = reinterpret_cast<const u32x4*>(a.cond_pack + (size_t)b * a.cond_pack_bstride);
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int v = threadIdx.x + i * THREADS;
                if (MT * FRV % THREADS == 0 || v < MT * FRV) wst[i] = src[v];
            }
        } else if (s < KS) {
            const u32x4* src = reinterpret_cast<const u32x4*>(a.wfg);
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int v = threadIdx.x + i * THREADS;
                if (MT * FRV % THREADS == 0 || v < MT * FRV) wst[i] = src[(size_t)((v / FRV) * KS + s) * FRV + v % FRV];
            }
        } else {
            const u32x4* src = reinterpret_cast<const u32x4*>(a.wd);
#pragma unroll
            for (int i = 0; i < DPT; ++i) {
                const int v = threadIdx.x + i * THREADS;
                if (ND * FRV % THREADS == 0 || v < ND * FRV) wst[i] = src[v];
            }
        }
    };
    auto stage_st = [&](int s) {
        if (CND && s == KS) {
            u32x4* dst = reinterpret_cast<u32x4*>(l_fg);
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int v = threadIdx.x + i * THREADS;
                if (MT * FRV % THREADS == 0 || v < MT * FRV) dst[(size_t)((v / FRV) * KS) * FRV + v % FRV] = wst[i];
            }
        } else if (s < KS) {
            u32x4* dst = reinterpret_cast<u32x4*>(l_fg);
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int v = threadIdx.x + i * THREADS;
                if (MT * FRV % THREADS == 0 || v < MT * FRV) dst[(size_t)((v / FRV) * KS + s) * FRV + v % FRV] = wst[i];
            }
        } else {
            u32x4* dst = reinterpret_cast<u32x4*>(l_d);
#pragma unroll
            for (int i = 0; i < DPT; ++i) {
                const int v = threadIdx.x + i * THREADS;
                if (ND * FRV % THREADS == 0 || v < ND * FRV) dst[v] = wst[i];
            }
        }
    };
    stage_ld(0);
    stage_st(0);

    f32x4 acc[MT][NT];
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
        for (int n = 0; n < NT; ++n) acc[m][n] = init;
    }
    __syncthreads();

#pragma unroll
    for (int s = 0; s < KS; ++s) {
        Frag<T> bf[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ENC ? fmaxf(raw[j][n], 0.f) : raw[j][n];
            split8<T, NS>(bf[n], v);
        }
        if (s + 1 < KS) issue(s + 1);
        stage_ld(s + 1);
        {   // weight fragment m+1 is on its way from LDS while the matrix core works on fragment m
            Frag<T> af[2];
            load_a<T, NS>(af[0], l_fg, s, lane);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m + 1 < MT) load_a<T, NS>(af[(m + 1) & 1], l_fg, (m + 1) * KS + s, lane);
#pragma unroll
                for (int n = 0; n < NT; ++n) mma<T, NS>(acc[m][n], af[m & 1], bf[n]);
            }
            // pin that order (the scheduler would put each LDS read right in front of its MFMAs)
            constexpr int RD = NS == 3 ? 2 : 1, MM = NT * NS;
            __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
#pragma unroll
            for (int m = 0; m < MT - 1; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MM, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MM, 0);
        }
        stage_st(s + 1);
        __syncthreads();
    }

    if (CND) {
        // conditioning bias as one more k-step: table fragments (hi + lo) times E[bucket][t] = (bucket(t) == bucket)
        typename T::vec8 e[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int id = (cbytes >> (8 * n)) & 0xFF;
            const bool mine = (id >> 3) == q;
            const uint32_t one = T::one16 << (16 * (id & 1));
            u32x4 ev;
#pragma unroll
            for (int w = 0; w < 4; ++w) ev[w] = (mine && ((id & 7) >> 1) == w) ? one : 0u;
            e[n] = __builtin_bit_cast(typename T::vec8, ev);
        }
        stage_ld(KS + 1);
        Frag<T> af[2];
        load_a<T, NS>(af[0], l_fg, 0, lane);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m + 1 < MT) load_a<T, NS>(af[(m + 1) & 1], l_fg, (m + 1) * KS, lane);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[m][n] = T::mfma(af[m & 1].lo, e[n], acc[m][n]);
                acc[m][n] = T::mfma(af[m & 1].hi, e[n], acc[m][n]);
            }
        }
        stage_st(KS + 1);
        __syncthreads();
    }
    if (!ENC && !CND && a.cond) {       // per-(channel, time-bucket) conditioning bias, gathered from a tiny table
        const float* cb = a.cond + (size_t)b * a.cond_bstride;
        int idx[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
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
                for (int n = 0; n < NT; ++n) acc[m][n][i] += cr[idx[n]];
            }
    }

    // residual rows in C layout (row 16m+4q+i, columns tl..tl+3): issue early, used at the end
    fvec res[MT2][4];
    if (a.write_x && NT == 4) {
#pragma unroll
        for (int m = 0; m < MT2; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                res[m][i] = VecN<NT>::ld(xin + (size_t)(16 * m + 4 * q + i) * a.pitch + tl);
    }

    // gate: z tile m = tanh(f tile m) * sigmoid(g tile m)
    f32x4 z[MT2][NT];
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                z[m][n][i] = ENC ? acc[m][n][i] : wn_gate(acc[m][n][i], acc[ENC ? m : m + MT2][n][i]);

    // z-crop store (rows 16m+4q+i; the lane's 4 N-tiles are 4 consecutive samples)
    {
        float* zo0 = a.z_out + (size_t)b * a.z_bstride;
        const long zx = a.z_half ? a.z_half - (long)(CH / 2) * a.pitch : 0;        // second clip of a pair (rows CH/2..)
#pragma unroll
        for (int m = 0; m < MT2; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* zo = zo0 + (CH == 64 && m >= MT2 / 2 ? zx : 0);
                fvec v;
#pragma unroll
                for (int n = 0; n < NT; ++n) v[n] = z[m][n][i];
                // z is not read again before the whole stack is done (skip product, backward): a streaming store keeps it
                // from pushing x_out - which the NEXT launch reads - out of L2 (about 1 % on the stack and on the skip product)
                if (NT == 4 && tl >= a.z_lo && tl + 3 < a.t_hi)
                    __builtin_nontemporal_store(v, reinterpret_cast<fvec*>(zo + (size_t)(16 * m + 4 * q + i) * a.pitch + tl));
                else
                VecN<NT>::stm(zo + (size_t)(16 * m + 4 * q + i) * a.pitch + tl, v, tl, a.z_lo, a.t_hi);
            }
    }
    if (!a.write_x) return;

    // dense: x' = Wd z + x   (B fragments straight from the z accumulators, chained k order)
    f32x4 acc2[MT2][NT];
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
        for (int n = 0; n < NT; ++n) acc2[m][n] = init;
    }
#pragma unroll
    for (int s = 0; s < KS2; ++s) {
        Frag<T> bf[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = ENC ? fmaxf(z[2 * s][n][i], 0.f) : z[2 * s][n][i];
                v[4 + i] = ENC ? fmaxf(z[2 * s + 1][n][i], 0.f) : z[2 * s + 1][n][i];
            }
            split8<T, NS>(bf[n], v);
        }
#pragma unroll
        for (int m = 0; m < MT2; ++m) {
            Frag<T> af;
            load_a<T, NS>(af, l_d, m * KS2 + s, lane);
#pragma unroll
            for (int n = 0; n < NT; ++n) mma<T, NS>(acc2[m][n], af, bf[n]);
        }
    }
    if (NT != 4) {                        // 128-VGPR budget: fetch the residual rows only now
#pragma unroll
        for (int m = 0; m < MT2; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                res[m][i] = VecN<NT>::ld(xin + (size_t)(16 * m + 4 * q + i) * a.pitch + tl);
    }
    float* xo = a.x_out + (size_t)b * a.x_bstride;
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fvec v;
#pragma unroll
            for (int n = 0; n < NT; ++n) v[n] = acc2[m][n][i] + res[m][i][n];
            // streaming store here too, although the next launch reads these rows: a launch's stores all come in its last
            // third, and what they leave dirty in the L2s is written back at the kernel boundary with nothing running
            // (0.838 vs 0.88-0.90 ms for the stack; write-through `sc1` stores: 0.88)
            if (NT == 4 && tl >= a.t_lo && tl + 3 < a.t_hi)
                __builtin_nontemporal_store(v, reinterpret_cast<fvec*>(xo + (size_t)(16 * m + 4 * q + i) * a.pitch + tl));
            else
            VecN<NT>::stm(xo + (size_t)(16 * m + 4 * q + i) * a.pitch + tl, v, tl, a.t_lo, a.t_hi);
        }
}


template <class T, int NS, int NT>
static int launch_fwd_nt(const WnResArgs& a, int ch, int batch, hipStream_t st) {
    WnResArgs k = a;
    k.swz = wn_xcd_swizzle_enabled();
    k.t_base = wn_tile_origin(a.t_lo);
    const int ncol = a.t_hi - k.t_base;
    constexpr int COLS = NtCfg<NT>::WAVES * 16 * NT;
    dim3 g((ncol + COLS - 1) / COLS, batch), b(64 * NtCfg<NT>::WAVES);
    const size_t fr = (NS == 3 ? 1024 : 512) * sizeof(uint16_t);
    if (ch == 32) {
        hipLaunchKernelGGL((resblock_fwd_nt_k<T, NS, 32, NT>), g, b, (size_t)(4 * 2 + 2 * 1) * fr, st, k);
    } else if (ch == 64) {
        const size_t sh = (size_t)(8 * 4 + 4 * 2) * fr;
        static WnDevOnce done;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (done.need(dev)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_fwd_nt_k<T, NS, 64, NT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            done.done(dev);
        }
        if constexpr (NS == 3 && NT == 4 && std::is_same<T, F16>::value) {
            if (k.cond && k.cond_pack && k.cond_idx && k.cond_le <= 32) {       // conditioning bias on the matrix cores
                if (a.t_lo - k.t_base > WN_PQ_IDX_PAD) return wn_set_error_msg(-4, "resblock_fwd: tile origin beyond the cond_idx pad");
                static WnDevOnce done_c;
                if (done_c.need(dev)) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_fwd_nt_k<T, NS, 64, NT, false, NtCfg<NT>::WAVES, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
                    done_c.done(dev);
                }
                hipLaunchKernelGGL((resblock_fwd_nt_k<T, NS, 64, NT, false, NtCfg<NT>::WAVES, true>), g, b, sh, st, k);
                WN_CHECK_LAUNCH();
                return 0;
            }
        }
        hipLaunchKernelGGL((resblock_fwd_nt_k<T, NS, 64, NT>), g, b, sh, st, k);
    } else {
        return wn_set_error_msg(-3, "resblock: padded channel count must be 32 or 64");
    }
    WN_CHECK_LAUNCH();
    return 0;
}

template <class T, int NS>
static int launch_enc(const WnResArgs& a, int ch, int batch, hipStream_t st) {
    WnResArgs k = a;
    k.swz = wn_xcd_swizzle_enabled();
    k.t_base = wn_tile_origin(a.t_lo);
    const int ncol = a.t_hi - k.t_base;
    constexpr int COLS = NtCfg<4>::WAVES * 64;
    dim3 g((ncol + COLS - 1) / COLS, batch), b(64 * NtCfg<4>::WAVES);
    const size_t fr = (NS == 3 ? 1024 : 512) * sizeof(uint16_t);
    if (ch == 32) {
        hipLaunchKernelGGL((resblock_fwd_nt_k<T, NS, 32, 4, true>), g, b, (size_t)(2 * 2 + 2 * 1) * fr, st, k);
    } else if (ch == 64) {
        hipLaunchKernelGGL((resblock_fwd_nt_k<T, NS, 64, 4, true>), g, b, (size_t)(4 * 4 + 4 * 2) * fr, st, k);
    } else {
        return wn_set_error_msg(-3, "enc_resblock: padded channel count must be 32 or 64");
    }
    WN_CHECK_LAUNCH();
    return 0;
}

int wn_launch_enc_resblock_fwd(const WnResArgs& a, int ch, int batch, int mode, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    switch (mode) {
        case WN_MODE_F16X3: return launch_enc<F16, 3>(a, ch, batch, st);
        case WN_MODE_F16X1: return launch_enc<F16, 1>(a, ch, batch, st);
        case WN_MODE_BF16X3: return launch_enc<BF16, 3>(a, ch, batch, st);
        case WN_MODE_BF16X1: return launch_enc<BF16, 1>(a, ch, batch, st);
    }
    return wn_set_error_msg(-2, "enc_resblock_fwd: bad mode");
}

int wn_launch_resblock_fwd_nt(const WnResArgs& a, int ch, int batch, int mode, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    switch (mode) {
        case WN_MODE_F16X3: return launch_fwd_nt<F16, 3, 4>(a, ch, batch, st);
        case WN_MODE_F16X1: return launch_fwd_nt<F16, 1, 4>(a, ch, batch, st);
        case WN_MODE_BF16X3: return launch_fwd_nt<BF16, 3, 4>(a, ch, batch, st);
        case WN_MODE_BF16X1: return launch_fwd_nt<BF16, 1, 4>(a, ch, batch, st);
    }
    return wn_set_error_msg(-2, "resblock_fwd: bad mode");
}
