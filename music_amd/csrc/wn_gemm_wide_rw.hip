// Wide channel-mixing product (>= 256 output rows: skip product, post-processing convs and their data gradients),
// "two-role" persistent form.  out[rows][t] = W[rows][K] in[K][t + shift0] (+ bias) (mask / relu-on-load as
// chan_gemm_wide2_k), one tap, x3 modes.
//
// chan_gemm_wide2_k: 256 x 256 tiles, all 8 waves of the workgroup convert, fill LDS and multiply in lock-step
// around one barrier per k-step (in-tile MFMA utilisation ~47 %), one workgroup per CU and per tile, so a tile's
// first loads and its 256 KB store are not covered by anything either (24 us of a 2-round launch).  Here
//   * M waves (0..3; wave i owns rows 64i..64i+63 of a 256-row group x 128 columns = 32 accumulator tiles) take
//     their packed weight fragments STRAIGHT from L2 into registers (they are private to the wave - no LDS),
//     one k-step ahead, read the activation fragments from LDS and do nothing but MFMAs (+ the tile's epilogue);
//   * C waves (4..7) stream the fp32 activation rows two k-steps ahead, split them and fill LDS (2 x 16 KB);
//   * a workgroup is persistent: it walks (tile, k-step) as one sequence, so the C waves are already converting
//     the next tile's first k-step while the M waves store the current tile.  The row groups of one column tile
//     run on the same XCD at the same time (its activations come from HBM once).
#include <stdlib.h>
#include <type_traits>
#include "wn_common.h"
#include "wn_kernels.h"

#define WR_THREADS 512
#define WR_COLS 128

typedef __bf16 wr_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wr_f32x2 __attribute__((ext_vector_type(2)));

struct WrPlan { int nct, rg, batch, n_col_total; };

template <class T>
__global__ __launch_bounds__(WR_THREADS) void chan_gemm_wide_rw_k(WnGemmArgs a, WrPlan pl) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];          // 2 stages x 8 fragments x 2 KB
    constexpr int STAGE = 8 * 1024;                                          // halfs
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wv & 3;
    const int c = lane & 15, q = lane >> 4;
    const int KS = a.ks0;

    // ---- this workgroup's sequence of (column tile, row group) items: the XCD x = id & 7 owns the column tiles
    // C = x, x + 8, ...; its workgroups walk (C, row group) interleaved, row group fastest
    int x, xs, j, wx;
    if (a.swz) {
        const int nwg = gridDim.x, id = blockIdx.x;
        x = id & 7; xs = 8; j = id >> 3;
        wx = (nwg - x + 7) >> 3;                       // workgroups on this XCD
    } else {
        x = 0; xs = 1; j = blockIdx.x; wx = gridDim.x;
    }
    const int ncx = pl.n_col_total > x ? (pl.n_col_total - x + xs - 1) / xs : 0;      // column tiles of this XCD
    const int items_x = ncx * pl.rg;
    const int n_items = j < items_x ? (items_x - j + wx - 1) / wx : 0;
    const int n_steps = n_items * KS;

    struct Pos { int b, t0, rgi, s; bool live; };
    auto pos_step = [&](int n) {                       // position of step n of this workgroup (clamped)
        Pos p;
        p.live = n < n_steps;
        n = n < n_steps ? n : n_steps - 1;
        n = n < 0 ? 0 : n;
        const int k = n / KS;
        p.s = n - k * KS;
        const int I = j + k * wx;
        const int cl = I / pl.rg;
        p.rgi = I - cl * pl.rg;
        int C = x + xs * cl;
        C = C < pl.n_col_total ? C : pl.n_col_total - 1;
        p.b = C / pl.nct;
        p.t0 = a.t_base + WR_COLS * (C - p.b * pl.nct);
        return p;
    };

    if (wv < 4) {
        // =========================== M waves ===========================
        f32x4 acc[4][8];
        Frag<T> wa0[4], wa1[4];                      // weight fragments of the even / odd steps (static indexing)
        auto load_w = [&](Frag<T>* w, Pos ps) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                int mt = ps.rgi * 16 + 4 * g + m;
                mt = mt < a.mt ? mt : a.mt - 1;                      // rows past the matrix are never stored
                load_a<T, 3>(w[m], a.wpack, mt * KS + ps.s, lane);
            }
        };
        load_w(wa0, pos_step(0));
        __syncthreads();
        auto m_body = [&](const int n, const Frag<T>* w, Frag<T>* wnext) {
            const Pos ps = pos_step(n);
            load_w(wnext, pos_step(n + 1));
            if (ps.s == 0) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x4 init = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias != nullptr) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = (ps.rgi * 16 + 4 * g + m) * 16 + 4 * q + i;
                            init[i] = row < a.m_valid ? a.bias[row] : 0.f;
                        }
                    }
#pragma unroll
                    for (int nn = 0; nn < 8; ++nn) acc[m][nn] = init;
                }
            }
            const uint16_t* st = lds + (size_t)(n & 1) * STAGE;
            Frag<T> bx[2];
            load_a<T, 3>(bx[0], st, 0, lane);
#pragma unroll
            for (int nn = 0; nn < 8; ++nn) {
                if (nn + 1 < 8) load_a<T, 3>(bx[(nn + 1) & 1], st, nn + 1, lane);
#pragma unroll
                for (int m = 0; m < 4; ++m) mma<T, 3>(acc[m][nn], w[m], bx[nn & 1]);
            }
            if (ps.s == KS - 1) {
                // ---- epilogue of this tile: rows of this wave, two 64-column groups
                float* out = a.out + (size_t)ps.b * a.out_bstride;
                const float* mask = a.mask ? a.mask + (size_t)ps.b * a.mask_bstride : nullptr;
#pragma unroll
                for (int cg = 0; cg < 2; ++cg) {
                    const int tl = ps.t0 + 64 * cg + 4 * c;
                    const bool full = tl >= a.t_lo && tl + 3 < a.t_hi;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = (ps.rgi * 16 + 4 * g + m) * 16 + 4 * q + i;
                            if (row >= a.m_valid) continue;
                            f32x4 v = {acc[m][4 * cg + 0][i], acc[m][4 * cg + 1][i], acc[m][4 * cg + 2][i], acc[m][4 * cg + 3][i]};
                            if (mask) {
                                const float* mp = mask + (size_t)row * a.mask_pitch + tl;
                                if (full) {
                                    const f32x4 mv = ld4u(mp);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (tl + e >= a.t_lo && tl + e < a.t_hi) v[e] = mp[e] > 0.f ? v[e] : 0.f;
                                }
                            }
                            float* op = out + (size_t)row * a.out_pitch + tl + a.out_shift;
                            if (full) {
                                F4U u = {{v[0], v[1], v[2], v[3]}};
                                *reinterpret_cast<F4U*>(op) = u;
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (tl + e >= a.t_lo && tl + e < a.t_hi) op[e] = v[e];
                            }
                        }
                    }
                }
            }
            __syncthreads();
        };
        for (int n = 0; n < n_steps; n += 2) {
            m_body(n, wa0, wa1);
            if (n + 1 < n_steps) m_body(n + 1, wa1, wa0);
        }
        return;
    }

    // =========================== C waves: column group g>>1, row half g&1 of every k-step ===========================
    const int lg = g >> 1, lh = g & 1;
    struct Raw { f32x4 v[4]; };
    auto load_b = [&](Raw& r, Pos ps) {
        const int col = ps.t0 + 64 * lg + 4 * c + a.shift0;
        const float* p = a.in0 + (size_t)ps.b * a.in_bstride + (size_t)(ps.s * 32 + 8 * q + 4 * lh) * a.in_pitch + col;
        const int tg0 = __builtin_amdgcn_readfirstlane(ps.t0) + 64 * lg + a.shift0;
        const bool inner = ps.live && tg0 >= a.in_lo && tg0 + 64 <= a.in_hi;
        if (inner) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) r.v[jj] = ld4u(p + (size_t)jj * a.in_pitch);
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) r.v[jj] = ps.live ? ld4g(p + (size_t)jj * a.in_pitch, col, a.in_lo, a.in_hi) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // split 4 rows x 4 columns and write the 8-byte hi / lo pieces of the 4 fragments of column group lg
    auto fill = [&](const Raw& r, int stage) {
        uint16_t* bb = lds + (size_t)stage * STAGE + (size_t)(lg * 4) * 1024 + lane * 8 + lh * 4;
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
            float x4[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float xv = r.v[jj][nn];
                if (a.relu_in) xv = fmaxf(xv, 0.f);
                x4[jj] = xv;
            }
            uint2 hv, lv;
            if (std::is_same<T, BF16>::value) {
                uint32_t h[2], l[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const wr_f32x2 pv = {x4[2 * u], x4[2 * u + 1]};
                    h[u] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pv, wr_bf16x2));
                    const wr_f32x2 rv = {pv[0] - __builtin_bit_cast(float, h[u] << 16), pv[1] - __builtin_bit_cast(float, h[u] & 0xffff0000u)};
                    l[u] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rv, wr_bf16x2));
                }
                hv = uint2{h[0], h[1]};
                lv = uint2{l[0], l[1]};
            } else {
                typename T::elem h[4], l[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    h[jj] = T::cvt(x4[jj]);
                    l[jj] = T::cvt(x4[jj] - T::back(h[jj]));
                }
                auto pk = [](typename T::elem u0, typename T::elem u1) {
                    return (uint32_t)__builtin_bit_cast(uint16_t, u0) | ((uint32_t)__builtin_bit_cast(uint16_t, u1) << 16);
                };
                hv = uint2{pk(h[0], h[1]), pk(h[2], h[3])};
                lv = uint2{pk(l[0], l[1]), pk(l[2], l[3])};
            }
            *reinterpret_cast<uint2*>(bb + (size_t)nn * 1024) = hv;
            *reinterpret_cast<uint2*>(bb + (size_t)nn * 1024 + 512) = lv;
        }
    };
    // r1 / r0 hold the raw rows of steps n+1 / n+2 (loop unrolled by two; each set is re-armed two steps ahead)
    Raw r0, r1;
    load_b(r0, pos_step(0));
    load_b(r1, pos_step(1));
    fill(r0, 0);
    load_b(r0, pos_step(2));
    __syncthreads();
    for (int n = 0; n < n_steps; n += 2) {
        fill(r1, 1);
        load_b(r1, pos_step(n + 3));
        __syncthreads();
        if (n + 1 < n_steps) {
            fill(r0, 0);
            load_b(r0, pos_step(n + 4));
            __syncthreads();
        }
    }
}

static int wr_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("WN_GEMM_WIDE_RW"); v = e ? (atoi(e) != 0) : 0; }
    return v;
}

// returns 1 if the launch was taken, 0 if the arguments are outside this kernel's preconditions
int wn_launch_gemm_wide_rw(const WnGemmArgs& k, int batch, int mode, hipStream_t st) {
    if (!wr_enabled()) return 0;
    if (mode != WN_MODE_BF16X3 && mode != WN_MODE_F16X3) return 0;
    if (k.mt < 16 || k.ks1 != 0 || k.in1 || k.resid || k.ks0 < 2) return 0;
    // OPT-IN (WN_GEMM_WIDE_RW=1; WN_GEMM_WIDE_RW_MINRG = least number of 256-row groups, default 1): measured SLOWER
    // than chan_gemm_wide2_k at config 2, each kernel alone: skip product 0.42 vs 0.36 ms, post-processing 82 vs 67 us,
    // Ws^T dU (1920 rows) 0.425 vs 0.388 ms - with ONE MFMA wave per SIMD nothing fills the matrix core while that
    // wave waits for LDS or L2, which costs more than the lock-step of wide2's two.  (Beside the side-stream
    // weight gradients it looks faster, 0.72 -> 0.46 ms, only because a persistent kernel takes the CUs from them:
    // the stack that follows then pays 0.3 ms.)
    static int minrg = -1;
    if (minrg < 0) { const char* e = getenv("WN_GEMM_WIDE_RW_MINRG"); minrg = e ? atoi(e) : 1; }
    if ((k.mt + 15) / 16 < minrg) return 0;
    WrPlan pl;
    pl.batch = batch;
    pl.nct = (k.t_hi - k.t_base + WR_COLS - 1) / WR_COLS;
    pl.rg = (k.mt + 15) / 16;
    pl.n_col_total = pl.nct * batch;
    const long items = (long)pl.n_col_total * pl.rg;
    const int nwg = (int)(items < 256 ? items : 256);
    const size_t sh = (size_t)2 * 8 * 1024 * sizeof(uint16_t);
    if (mode == WN_MODE_BF16X3) hipLaunchKernelGGL(chan_gemm_wide_rw_k<BF16>, dim3(nwg), dim3(WR_THREADS), sh, st, k, pl);
    else hipLaunchKernelGGL(chan_gemm_wide_rw_k<F16>, dim3(nwg), dim3(WR_THREADS), sh, st, k, pl);
    return 1;
}
