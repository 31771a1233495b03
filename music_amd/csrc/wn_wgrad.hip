// Weight-gradient products: C[m][n] = sum_{b,t} A[m][t + a_shift] * B_tap[n][t + b_shift_tap]
// (dWd = sum dy z^T, dWf0 = sum df x(t-d)^T, dWs = sum du z^T, ...).
//
// Both operands are channels-first rows with time contiguous, so TIME is the MFMA k index and
// every fragment is 8 consecutive samples of one row (two float4 loads, no transposition):
//   A fragment: lane (c,q) holds A[16m + c][tb + 8q + j]
//   B fragment: lane (c,q) holds B[16n + c][tb + 8q + j]        (C = A * B^T)
// One wave owns a 64x64 block of C and walks a chunk of the time axis with the next step's loads
// in flight behind the current step's MFMAs.  A 64x64 problem (one block) is instead split in
// time over the 4 waves of the workgroup and combined through LDS.
// Each workgroup writes its partial C into its own SLAB with plain stores (no float atomics: the
// chip-wide atomic rate, ~1.3 TB/s, would dominate, and the result would depend on arrival
// order); wn_reduce_slabs sums the slabs in a fixed order, so weight gradients are bit-reproducible.
// Gradients are split in bf16 (fp32 exponent range); see wn_common.h for the x3 scheme.
#include <type_traits>
#include "wn_common.h"
#include "wn_kernels.h"

struct WgRaw { f32x4 u0, u1; };

__device__ __forceinline__ WgRaw wg_load(const float* row, int col, int ncols) {
    WgRaw r;
    r.u0 = ld4g(row + col, col, 0, ncols);
    r.u1 = ld4g(row + col + 4, col + 4, 0, ncols);
    return r;
}

template <class T, int NS>
__device__ __forceinline__ void wg_frag(Frag<T>& f, const WgRaw& r, int t, int t_lo, int t_hi, bool masked, bool relu) {
    float v[8] = {r.u0[0], r.u0[1], r.u0[2], r.u0[3], r.u1[0], r.u1[1], r.u1[2], r.u1[3]};
    if (masked) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (t + j < t_lo || t + j >= t_hi) v[j] = 0.f;
    }
    if (relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    split8<T, NS>(f, v);
}

// Two independent problems can share one launch (the per-layer dWfg and dWd products): the
// workgroups with blockIdx.y >= p.y_split work on p.p[1].
struct WnWgradPair { WnWgradArgs p[2]; int y_split; };

template <class T, int NS>
__global__ __launch_bounds__(512) void wgrad_k(WnWgradPair pr) {
    // 8 waves: waves w and w + 4 own the same 64x64 block and split the chunk in time (a single-block
    // problem is split 8 ways); the partial blocks are combined through LDS at the end.  A layer's
    // launch has <= 256 workgroups, so the second set of waves is what fills each SIMD's issue slots
    // while the first waits on its loads.
    __shared__ float red[4][64 * 64];
    const WnBlock wb = wn_block(pr.p[0].swz);
    const bool second = wb.y >= pr.y_split;
    const WnWgradArgs& a = pr.p[second ? 1 : 0];
    const int by = second ? wb.y - pr.y_split : wb.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int b = wb.z;
    const int nt_total = a.nt_per_tap * (a.b1 ? 2 : 1);
    const int nblk_n = (nt_total + 3) / 4, nblk_m = (a.mt + 3) / 4;
    const int nblk = nblk_n * nblk_m;
    const bool split_time = nblk == 1;
    const int blk = split_time ? 0 : by * 4 + (wave & 3);
    const bool active = blk < nblk;
    const int mb = active ? blk / nblk_n : 0, nb = active ? blk % nblk_n : 0;
    int tc0 = a.t_base + wb.x * a.chunk;
    int tc1 = tc0 + a.chunk;
    if (tc1 > a.t_hi) tc1 = a.t_hi;
    const int n_chunks = (a.t_hi - a.t_base + a.chunk - 1) / a.chunk;     // grid.x may be larger (paired launch)
    {                                                   // this wave's share of the chunk (multiple of 32)
        const int ts = split_time ? 8 : 2, part = split_time ? wave : (wave >> 2);
        const int sub = ((a.chunk / ts) + 31) & ~31;
        tc0 += part * sub;
        int e = tc0 + sub;
        if (e < tc1) tc1 = e;
    }

    const float* A = a.a + (size_t)b * a.a_bstride;
    const float* arow[4];
    const float* brow[4];
    int bshift[4];
    bool mval[4], nval[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        int mt = mb * 4 + m;
        mval[m] = mt < a.mt;
        arow[m] = A + (size_t)((mval[m] ? mt : 0) * 16 + c) * a.a_pitch;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        int nt = nb * 4 + n;
        nval[n] = nt < nt_total;
        if (!nval[n]) nt = 0;
        int tap = nt / a.nt_per_tap, r = (nt % a.nt_per_tap) * 16 + c;
        const float* base = (tap == 0 ? a.b0 : a.b1) + (size_t)b * a.b_bstride;
        brow[n] = base + (size_t)r * a.b_pitch;
        bshift[n] = tap == 0 ? a.b_shift0 : a.b_shift1;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (active && tc0 < tc1) {
        WgRaw ra[4], rb[4];
        // [tb_lo, tb_hi]: k-steps whose 32 samples are addressable in every operand row
        const int bs1 = a.b1 ? a.b_shift1 : a.b_shift0;
        const int tb_lo = -min(a.a_shift, min(a.b_shift0, bs1));
        const int tb_hi = min(a.a_cols - a.a_shift, a.b_cols - max(a.b_shift0, bs1)) - 32;
        auto issue = [&](int tb) {
            const int t = tb + 8 * q;
            const int tbu = __builtin_amdgcn_readfirstlane(tb);
            if (tbu >= tb_lo && tbu <= tb_hi) {         // wave-uniform: plain 16-B loads
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float* p = arow[m] + t + a.a_shift;
                    ra[m].u0 = ld4u(p);
                    ra[m].u1 = ld4u(p + 4);
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const float* p = brow[n] + t + bshift[n];
                    rb[n].u0 = ld4u(p);
                    rb[n].u1 = ld4u(p + 4);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) ra[m] = wg_load(arow[m], t + a.a_shift, a.a_cols);
#pragma unroll
                for (int n = 0; n < 4; ++n) rb[n] = wg_load(brow[n], t + bshift[n], a.b_cols);
            }
        };
        issue(tc0);
        for (int tb = tc0; tb < tc1; tb += 32) {
            const bool masked = (tb < a.t_lo) || (tb + 32 > tc1);
            const int t = tb + 8 * q;
            Frag<T> af[4], bf[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) wg_frag<T, NS>(af[m], ra[m], t, a.t_lo, tc1, masked, false);
#pragma unroll
            for (int n = 0; n < 4; ++n) wg_frag<T, NS>(bf[n], rb[n], t, a.t_lo, tc1, masked, a.relu_b != 0);
            if (tb + 32 < tc1) issue(tb + 32);          // next step's loads fly behind the MFMAs
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) mma<T, NS>(acc[m][n], af[m], bf[n]);
        }
    }

    // combine the time shares through LDS: waves 4..7 -> waves 0..3, then (single block) 1..3 -> 0
    auto park = [&](int slot) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                *reinterpret_cast<f32x4*>(&red[slot][((m * 4 + n) * 64 + lane) * 4]) = acc[m][n];
    };
    auto take = [&](int slot) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[m][n] += *reinterpret_cast<const f32x4*>(&red[slot][((m * 4 + n) * 64 + lane) * 4]);
    };
    if (wave >= 4) park(wave - 4);
    __syncthreads();
    if (wave < 4) take(wave);
    if (split_time) {                  // (uniform over the workgroup) shares 1..3 -> wave 0
        if (wave >= 1 && wave < 4) park(wave);          // slot w was read by wave w only: no hazard
        __syncthreads();
        if (wave == 0) { take(1); take(2); take(3); }
        if (wave != 0) return;
    }
    if (wave >= 4) return;
    if (!active) return;
    // slab of this workgroup: plain stores, every element of the block is written
    if (wb.x >= n_chunks) return;
    float* cs = a.c + ((size_t)b * n_chunks + wb.x) * a.c_slab_stride;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (!mval[m]) continue;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (!nval[n]) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = (mb * 4 + m) * 16 + 4 * q + i;
                int col = (nb * 4 + n) * 16 + c;
                cs[(size_t)row * a.ldc + col] = acc[m][n][i];
            }
        }
    }
}

int wn_wgrad_num_slabs(int t_lo, int t_hi, int chunk, int batch) {
    if (t_hi <= t_lo || batch <= 0) return 0;
    if (chunk < 128) chunk = 128;
    chunk = (chunk + 127) & ~127;
    const int t_base = t_lo & ~31;
    return ((t_hi - t_base + chunk - 1) / chunk) * batch;
}

// ---------------------------------------------------------------------------------------------
// Large outputs (>= 256 x 256: dP1, dP2, dWskip): one workgroup of 8 waves owns a 256 x 256 block
// of C.  Per 32-sample k-step every wave loads + splits only 4 of the 32 operand tiles and parks
// the 16-bit fragments in LDS (double-buffered, 2 x 64 KB in the x3 modes); all waves then read
// the 8 A and 4 B fragments of their 128 x 64 sub-block with ds_read_b128.  This removes the
// 8-30x redundant fp32 -> hi/lo splitting that bounds wgrad_k on these shapes.
// ---------------------------------------------------------------------------------------------
template <class T, int NS>
__global__ __launch_bounds__(512) void wgrad_big_k(WnWgradArgs a) {
    constexpr int FR = (NS == 3 ? 1024 : 512);
    extern __shared__ __attribute__((aligned(16))) uint16_t l_f[];      // [2][32][FR]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const WnBlock wb = wn_block<true>(a.swz);       // output blocks over the same time chunk share an XCD
    const int b = wb.z;
    const int nt_total = a.nt_per_tap * (a.b1 ? 2 : 1);
    const int n_ng = (nt_total + 15) / 16;
    const int mg = wb.y / n_ng, ng = wb.y % n_ng;
    const int tc0 = a.t_base + wb.x * a.chunk;
    int tc1 = tc0 + a.chunk;
    if (tc1 > a.t_hi) tc1 = a.t_hi;
    const int wm = wave >> 2, wn = wave & 3;

    // the 4 tiles this wave loads: A tiles 2w, 2w+1 and B tiles 2w, 2w+1 of the block
    const float* lrow[4];
    int lshift[4], lcols[4];
    bool lval[4], lrelu[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int tile = 2 * wave + (k & 1);
        if (k < 2) {
            const int mt = mg * 16 + tile;
            lval[k] = mt < a.mt;
            lrow[k] = a.a + (size_t)b * a.a_bstride + (size_t)((lval[k] ? mt : 0) * 16 + c) * a.a_pitch;
            lshift[k] = a.a_shift; lcols[k] = a.a_cols; lrelu[k] = false;
        } else {
            int nt = ng * 16 + tile;
            lval[k] = nt < nt_total;
            if (!lval[k]) nt = 0;
            const int tap = nt / a.nt_per_tap, r = (nt % a.nt_per_tap) * 16 + c;
            lrow[k] = (tap == 0 ? a.b0 : a.b1) + (size_t)b * a.b_bstride + (size_t)r * a.b_pitch;
            lshift[k] = tap == 0 ? a.b_shift0 : a.b_shift1; lcols[k] = a.b_cols; lrelu[k] = a.relu_b != 0;
        }
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    WgRaw raw[4];
    // [tb_lo, tb_hi]: k-steps whose 32 samples are addressable in every operand row; the time loop
    // exists twice (plain / guarded loads), chosen per workgroup chunk (see wgrad_k)
    const int bs1 = a.b1 ? a.b_shift1 : a.b_shift0;
    const int tb_lo = -min(a.a_shift, min(a.b_shift0, bs1));
    const int tb_hi = min(a.a_cols - a.a_shift, a.b_cols - max(a.b_shift0, bs1)) - 32;
    auto park = [&](int buf, int tb) {
        const bool masked = (tb < a.t_lo) || (tb + 32 > tc1);
        const int t = tb + 8 * q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            Frag<T> f;
            wg_frag<T, NS>(f, raw[k], t, a.t_lo, tc1, masked || !lval[k], lrelu[k]);
            if (!lval[k]) {                                  // tile outside the matrix: zeros
#pragma unroll
                for (int j = 0; j < 8; ++j) { f.hi[j] = T::cvt(0.f); f.lo[j] = T::cvt(0.f); }
            }
            const int slot = (k < 2 ? 0 : 16) + 2 * wave + (k & 1);
            u32x4* dst = reinterpret_cast<u32x4*>(l_f + ((size_t)buf * 32 + slot) * FR);
            dst[lane] = __builtin_bit_cast(u32x4, f.hi);
            if (NS == 3) dst[64 + lane] = __builtin_bit_cast(u32x4, f.lo);
        }
    };
    auto run = [&](auto guarded) {
        auto issue = [&](int tb) {
            const int t = tb + 8 * q;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (decltype(guarded)::value) {
                    raw[k] = wg_load(lrow[k], t + lshift[k], lcols[k]);
                } else {
                    const float* p = lrow[k] + t + lshift[k];
                    raw[k].u0 = ld4u(p);
                    raw[k].u1 = ld4u(p + 4);
                }
            }
        };
        // one request of the 8 per k-step (tile k, first / second 16 bytes of the lane's 8 samples)
        auto issue_piece = [&](int tb, int k, int half) {
            const int t = tb + 8 * q;
            if (decltype(guarded)::value) {
                if (half == 0) raw[k] = wg_load(lrow[k], t + lshift[k], lcols[k]);
            } else {
                const float* p = lrow[k] + t + lshift[k];
                if (half == 0) raw[k].u0 = ld4u(p);
                else raw[k].u1 = ld4u(p + 4);
            }
        };
        const int tb_last = tc0 + ((tc1 - 1 - tc0) & ~31);
        if (tc0 < tc1) {
            issue(tc0);
            park(0, tc0);
            issue(tc0 + 32 < tc1 ? tc0 + 32 : tb_last);
        }
        __syncthreads();
        int it = 0;
        // k-step it: MFMAs out of stage it & 1.  Between its row tiles 3 and 4 the rows of k-step it + 1 (requested during the second
        // half of k-step it - 1) are split into the other stage, and behind that their registers are re-armed for k-step it + 2 one
        // tile (two requests) per row tile: eight waves that request 8 KB each right behind the barrier queue for the CU's one address path while
        // the matrix pipe waits (round 6, profiles/r06_gemm_tiles.md).  A fragments one row tile ahead of their MFMAs (two register
        // sets, fenced: hipcc folds them back into read - wait - 12 MFMAs otherwise).
        for (int tb = tc0; tb < tc1; tb += 32, ++it) {
            const int tb1 = tb + 32 < tc1 ? tb + 32 : tb_last, tb2 = tb + 64 < tc1 ? tb + 64 : tb_last;
            const uint16_t* base = l_f + (size_t)(it & 1) * 32 * FR;
            Frag<T> bf[4], af[2];
#pragma unroll
            for (int n = 0; n < 4; ++n) load_a<T, NS>(bf[n], base, 16 + wn * 4 + n, lane);
            load_a<T, NS>(af[0], base, wm * 8, lane);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m + 1 < 8) load_a<T, NS>(af[(m + 1) & 1], base, wm * 8 + m + 1, lane);
                if (m == 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    park((it + 1) & 1, tb1);                   // (the last fill is never read)
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (m >= 4) {
                    issue_piece(tb2, m - 4, 0);                // tile m - 4: both halves of the lane's 8 samples
                    issue_piece(tb2, m - 4, 1);
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) mma<T, NS>(acc[m][n], af[m & 1], bf[n]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    };
    {
        const int last_tb = tc0 + ((tc1 - 1 - tc0) & ~31);
        if (tc0 >= tb_lo && last_tb <= tb_hi) run(std::false_type{});
        else run(std::true_type{});
    }
    const int n_chunks = (a.t_hi - a.t_base + a.chunk - 1) / a.chunk;
    float* cs = a.c + ((size_t)b * n_chunks + wb.x) * a.c_slab_stride;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int mt = mg * 16 + wm * 8 + m;
        if (mt >= a.mt) continue;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int nt = ng * 16 + wn * 4 + n;
            if (nt >= nt_total) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                cs[(size_t)(mt * 16 + 4 * q + i) * a.ldc + nt * 16 + c] = acc[m][n][i];
        }
    }
}

static void wg_prepare(WnWgradArgs& k, int& nchunks, int& ygroups) {
    k.t_base = k.t_lo & ~31;
    k.swz = wn_xcd_swizzle_enabled();
    if (k.chunk < 128) k.chunk = 128;
    k.chunk = (k.chunk + 127) & ~127;
    const int nt_total = k.nt_per_tap * (k.b1 ? 2 : 1);
    const int nblk = ((nt_total + 3) / 4) * ((k.mt + 3) / 4);
    nchunks = (k.t_hi - k.t_base + k.chunk - 1) / k.chunk;
    ygroups = (nblk + 3) / 4;
}

// a2 may be null.  Both problems share the clip count; each writes its own slabs.
int wn_launch_wgrad2(const WnWgradArgs* a1, const WnWgradArgs* a2, int batch, int mode, hipStream_t st) {
    if (batch <= 0) return 0;
    WnWgradPair pr;
    int n = 0, nch[2] = {0, 0}, yg[2] = {0, 0};
    const WnWgradArgs* src[2] = {a1, a2};
    for (int i = 0; i < 2; ++i) {
        if (!src[i] || src[i]->t_hi <= src[i]->t_lo) continue;
        pr.p[n] = *src[i];
        wg_prepare(pr.p[n], nch[n], yg[n]);
        ++n;
    }
    if (n == 0) return 0;
    if (n == 1) {
        const WnWgradArgs& k = pr.p[0];
        const int nt_total = k.nt_per_tap * (k.b1 ? 2 : 1);
        if (k.mt >= 16 && nt_total >= 16) {
            const int ns = wn_mode_ns(mode);
            const size_t sh = (size_t)2 * 32 * (ns == 3 ? 1024 : 512) * sizeof(uint16_t);
            dim3 g(nch[0], ((k.mt + 15) / 16) * ((nt_total + 15) / 16), batch), b(512);
            int dev = 0;
            (void)hipGetDevice(&dev);
            static WnDevOnce done[4];
#define WN_BIG(TT, NN, slot) do { \
                if (done[slot].need(dev)) { \
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_big_k<TT, NN>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
                    done[slot].done(dev); } \
                hipLaunchKernelGGL((wgrad_big_k<TT, NN>), g, b, sh, st, k); } while (0)
            switch (mode) {
                case WN_MODE_BF16X3: WN_BIG(BF16, 3, 0); break;
                case WN_MODE_BF16X1: WN_BIG(BF16, 1, 1); break;
                case WN_MODE_F16X3: WN_BIG(F16, 3, 2); break;
                case WN_MODE_F16X1: WN_BIG(F16, 1, 3); break;
                default: return wn_set_error_msg(-2, "wgrad: bad mode");
            }
#undef WN_BIG
            WN_CHECK_LAUNCH();
            return 0;
        }
        pr.p[1] = pr.p[0]; yg[1] = 0; nch[1] = 0;
    }
    pr.y_split = yg[0];
    dim3 g(nch[0] > nch[1] ? nch[0] : nch[1], yg[0] + yg[1], batch), b(512);
    switch (mode) {
        case WN_MODE_BF16X3: hipLaunchKernelGGL((wgrad_k<BF16, 3>), g, b, 0, st, pr); break;
        case WN_MODE_BF16X1: hipLaunchKernelGGL((wgrad_k<BF16, 1>), g, b, 0, st, pr); break;
        case WN_MODE_F16X3: hipLaunchKernelGGL((wgrad_k<F16, 3>), g, b, 0, st, pr); break;
        case WN_MODE_F16X1: hipLaunchKernelGGL((wgrad_k<F16, 1>), g, b, 0, st, pr); break;
        default: return wn_set_error_msg(-2, "wgrad: bad mode");
    }
    WN_CHECK_LAUNCH();
    return 0;
}

int wn_launch_wgrad(const WnWgradArgs& a, int batch, int mode, hipStream_t st) {
    return wn_launch_wgrad2(&a, nullptr, batch, mode, st);
}

// Batched deterministic slab reduction.  desc[op] = {vec_start, slab_off, n_slabs, stride, out_off, n}
// (int64 each): out[out_off + e] = sum_{s < n_slabs} slab[slab_off + s*stride + e], e < n, summed in
// slab order.  Work item v (4 floats) belongs to the op with vec_start <= v < next vec_start.
__global__ __launch_bounds__(256) void reduce_slabs_k(const long* __restrict__ desc, int n_ops, long total_vec,
                                                      const float* __restrict__ slab, float* __restrict__ out) {
    long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= total_vec) return;
    int lo = 0, hi = n_ops - 1;
    while (lo < hi) {                                   // last op with vec_start <= v
        int mid = (lo + hi + 1) >> 1;
        if (desc[mid * 6] <= v) lo = mid; else hi = mid - 1;
    }
    const long* d = desc + lo * 6;
    const long e = (v - d[0]) * 4, n = d[5], stride = d[3];
    const int ns = (int)d[2];
    const float* sp = slab + d[1] + e;
    float* op = out + d[4] + e;
    if (e + 3 < n) {
        // 8 slabs in flight per thread (8 partial sums, combined in a fixed order: still deterministic); with one
        // running sum the loop waits out a memory latency per couple of slabs
        f32x4 s8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s8[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        int i = 0;
        for (; i + 8 <= ns; i += 8) {
            f32x4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = ld4u(sp + (size_t)(i + k) * stride);
#pragma unroll
            for (int k = 0; k < 8; ++k) s8[k] += t[k];
        }
        for (; i < ns; ++i) s8[0] += ld4u(sp + (size_t)i * stride);
        const f32x4 s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        op[0] = s[0]; op[1] = s[1]; op[2] = s[2]; op[3] = s[3];
    } else {
        for (long k = 0; e + k < n; ++k) {
            float s = 0.f;
            for (int i = 0; i < ns; ++i) s += sp[(size_t)i * stride + k];
            op[k] = s;
        }
    }
}

int wn_launch_reduce_slabs(const long* desc, int n_ops, long total_vec, const float* slab, float* out, hipStream_t st) {
    if (total_vec <= 0 || n_ops <= 0) return 0;
    hipLaunchKernelGGL(reduce_slabs_k, dim3((unsigned)((total_vec + 255) / 256)), dim3(256), 0, st, desc, n_ops,
                       total_vec, slab, out);
    WN_CHECK_LAUNCH();
    return 0;
}
