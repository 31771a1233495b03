// B-stationary form of the wide channel-mixing product (time on the MFMA lanes) for SHORT reductions and MANY rows:
//
//   out[m][t + out_shift] = mask( bias[m] + sum_k W[m][k] * pre(in[k][t + shift0]) ),   K = 32 ks0 <= 256, one tap
//
// i.e. every "weights transposed" data-gradient product of the epilogue whose K is a skip-channel count - above all
// dZ = Ws^T dU (wavenet/model.py:127-134 backward: 30 x 64 = 1920 rows against K = 256, 795 MB of output at config 2).
// chan_gemm_wide2_k pays for such a product per 256 x 256 tile: eight row groups each fetch and split the same activation
// tile, every k-step is an LDS fill + barrier of both operands, and 42 % of a tile's time is its cold start and its store.
// Here the ACTIVATION tile is the stationary operand: a workgroup splits K x 128 columns into ready hi/lo B fragments ONCE
// (128 KB of LDS in the x3 modes), then walks the row tiles of the whole product over it.  In that walk nothing is written
// to LDS and there is no barrier: a wave owns MT row tiles x all 8 column tiles (MT x 8 accumulator tiles), takes its packed
// weight fragments straight from L2 (they are read by one wave only: no sharing to organise; the 2 MB pack of Ws^T stays
// L2-resident) one k-step ahead, reads the B fragments with ds_read_b128 (22 % of the LDS read rate), and stores a pass's rows
// (streaming stores where the rows are 16-byte aligned: 333 -> 293 us alone)
// while the next pass's first weights are already on their way (vector memory completes in order: the requests are issued in
// front of the stores).  The eight waves run free of each other; the two on a SIMD fill each other's stalls.
//
// Persistent grid (one workgroup per CU): whole tiles round-robin, the leftover tiles of the last round split by PASSES so
// that every CU multiplies the same number of row tiles to within one pass (816 tiles on 256 CUs: 3 whole tiles each, then
// 48 x 5 passes over 240 CUs).
#include <stdlib.h>
#include <type_traits>
#include "wn_common.h"
#include "wn_kernels.h"

#define BST_COLS 128

template <class T, int NS, int MT, int KS>
__global__ __launch_bounds__(512) void chan_gemm_bst_k(WnGemmArgs a, int ntx, int ntiles, int npass) {
    static_assert(KS % 2 == 0, "the weight slots alternate by k-step parity");
    constexpr int FR = (NS == 3 ? 1024 : 512);               // halfs per fragment
    extern __shared__ __attribute__((aligned(16))) uint16_t l_s[];   // [KS][8 column tiles] B fragments
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int G = gridDim.x, wg = blockIdx.x;
    const int nfull = ntiles / G;
    const int first_left = nfull * G;
    const int U = (ntiles - first_left) * npass;             // leftover (tile, pass) units, dealt in contiguous runs
    const int u0 = (int)((long)wg * U / G), u1 = (int)((long)(wg + 1) * U / G);

    // loader role of this wave in a fill: column group lg, row half lh, k-steps sp, sp + 2, ...
    const int lg = wave & 1, lh = (wave >> 1) & 1, sp = wave >> 2;
    const bool nt_ok = ((a.out_shift | a.out_pitch) & 3) == 0 && (a.out_bstride & 3) == 0 && ((size_t)a.out & 15) == 0;

    auto fill = [&](int b, int tile0) {
        const float* in = a.in0 + (size_t)b * a.in_bstride;
        const int tg0 = tile0 + lg * 64 + a.shift0;
        const bool inner = tg0 >= a.in_lo && tg0 + 64 <= a.in_hi;
        const int col = tg0 + 4 * c;
        f32x4 raw[(KS + 1) / 2][4];
#pragma unroll
        for (int i = 0; i < (KS + 1) / 2; ++i) {
            const int s = sp + 2 * i;
            if (s < KS) {
                const float* p = in + (size_t)(s * 32 + 8 * q + 4 * lh) * a.in_pitch + col;
                if (inner) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) raw[i][j] = ld4u(p + (size_t)j * a.in_pitch);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) raw[i][j] = ld4g(p + (size_t)j * a.in_pitch, col, a.in_lo, a.in_hi);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < (KS + 1) / 2; ++i) {
            const int s = sp + 2 * i;
            if (s < KS) {
                uint16_t* bb = l_s + (size_t)(s * 8 + lg * 4) * FR + lane * 8 + lh * 4;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    float x[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = a.relu_in ? fmaxf(raw[i][j][n], 0.f) : raw[i][j][n];
                    uint2 hv, lv;
                    if (NS == 3) {
                        split2<T>(x[0], x[1], hv.x, lv.x);
                        split2<T>(x[2], x[3], hv.y, lv.y);
                    } else {
                        hv.x = cvt2<T>(x[0], x[1]);
                        hv.y = cvt2<T>(x[2], x[3]);
                    }
                    *reinterpret_cast<uint2*>(bb + (size_t)n * FR) = hv;
                    if (NS == 3) *reinterpret_cast<uint2*>(bb + (size_t)n * FR + 512) = lv;
                }
            }
        }
    };

    // weights of row tiles m0 .. m0 + MT - 1, k-step s
    auto load_w = [&](Frag<T>* af, int m0, int s) {
#pragma unroll
        for (int i = 0; i < MT; ++i) load_a<T, NS>(af[i], a.wpack, (m0 + i) * KS + s, lane);
    };

    auto run = [&](int tile, int p0, int p1, bool first) {
        const int b = tile / ntx, tile0 = a.t_base + (tile % ntx) * BST_COLS;
        if (!first) __syncthreads();                          // every wave is done with the previous tile's fragments
        fill(b, tile0);
        __syncthreads();
        float* out = a.out + (size_t)b * a.out_bstride;
        const bool tile_in = tile0 >= a.t_lo && tile0 + BST_COLS <= a.t_hi;
        const float* mask = a.mask ? a.mask + (size_t)b * a.mask_bstride : nullptr;
        Frag<T> af[2][MT];
        int m0 = (p0 * 8 + wave) * MT;
        if (m0 < a.mt) load_w(af[0], m0, 0);
        for (int p = p0; p < p1; ++p, m0 += 8 * MT) {
            if (m0 >= a.mt) break;                            // wave-uniform; later passes only have higher rows
            f32x4 acc[MT][8];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f32x4 init = {0.f, 0.f, 0.f, 0.f};
                if (a.bias != nullptr) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = (m0 + i) * 16 + 4 * q + r;
                        init[r] = row < a.m_valid ? a.bias[row] : 0.f;
                    }
                }
#pragma unroll
                for (int n = 0; n < 8; ++n) acc[i][n] = init;
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) load_w(af[(s + 1) & 1], m0, s + 1);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    Frag<T> bf;
                    load_a<T, NS>(bf, l_s, s * 8 + n, lane);
#pragma unroll
                    for (int i = 0; i < MT; ++i) mma<T, NS>(acc[i][n], af[s & 1][i], bf);
                }
            }
            // the next pass's first weights are requested in front of this pass's stores (in-order completion)
            const int mn = m0 + 8 * MT;
            // (unconditionally - a branch around loads makes hipcc drain the queue - so the last pass re-requests its own;
            // KS is even: slot 0 was last read at k-step KS - 2)
            load_w(af[0], mn < a.mt ? mn : m0, 0);
            // interior tile, every row real, no mask: 8 MT plain 16-byte stores back to back (wave-uniform test)
            if (tile_in && mask == nullptr && (m0 + MT) * 16 <= a.m_valid) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* op = out + (size_t)((m0 + i) * 16 + 4 * q + r) * a.out_pitch + tile0 + 4 * c + a.out_shift;
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            if (nt_ok) {        // 16-byte aligned rows (dZ): streaming stores, the output is read by a later launch
                                const f32x4 v = {acc[i][4 * g][r], acc[i][4 * g + 1][r], acc[i][4 * g + 2][r], acc[i][4 * g + 3][r]};
                                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(op + 64 * g));
                            } else {
                                F4U u = {{acc[i][4 * g][r], acc[i][4 * g + 1][r], acc[i][4 * g + 2][r], acc[i][4 * g + 3][r]}};
                                *reinterpret_cast<F4U*>(op + 64 * g) = u;
                            }
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (m0 + i) * 16 + 4 * q + r;
                    if (row >= a.m_valid) continue;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int tl = tile0 + 64 * g + 4 * c;
                        const bool full = tl >= a.t_lo && tl + 3 < a.t_hi;
                        f32x4 v = {acc[i][4 * g][r], acc[i][4 * g + 1][r], acc[i][4 * g + 2][r], acc[i][4 * g + 3][r]};
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
    };

    bool first = true;
    for (int j = 0; j < nfull; ++j) {
        run(j * G + wg, 0, npass, first);
        first = false;
    }
    for (int u = u0; u < u1;) {
        const int p0 = u % npass;
        int p1 = p0 + (u1 - u);
        if (p1 > npass) p1 = npass;
        run(first_left + u / npass, p0, p1, first);
        first = false;
        u += p1 - p0;
    }
}

template <class T, int NS, int MT, int KS>
static void bst_launch(const WnGemmArgs& k, int ntx, int ntiles, int npass, int grid, hipStream_t st) {
    const size_t sh = (size_t)KS * 8 * (NS == 3 ? 1024 : 512) * sizeof(uint16_t);
    static WnDevOnce done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done.need(dev)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chan_gemm_bst_k<T, NS, MT, KS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        done.done(dev);
    }
    hipLaunchKernelGGL((chan_gemm_bst_k<T, NS, MT, KS>), dim3(grid), dim3(512), sh, st, k, ntx, ntiles, npass);
}

// returns 1 if the launch was taken, 0 if the arguments are outside this kernel's preconditions
int wn_launch_gemm_bst(const WnGemmArgs& k, int batch, int mode, hipStream_t st) {
    if (mode != WN_MODE_BF16X3 && mode != WN_MODE_F16X3) return 0;
    if (k.ks1 != 0 || k.in1 != nullptr || k.resid != nullptr) return 0;
    if (k.ks0 != 8 || k.mt < 32 || k.mt % 3 != 0) return 0;            // K = 256, >= 512 rows in whole groups of 3 row tiles
    if (k.t_base & 63) return 0;
    // WN_GEMM_BST: 0 = chan_gemm_wide2_k; 1 = one workgroup per CU (the balanced persistent walk: fastest ALONE, 293 against 321 us);
    // default = one workgroup per tile: this product runs beside the epilogue's weight gradients on the side stream, and a grid that
    // holds every CU for its whole duration pushes them out to run beside the backward stack (same-box bench: 4.48 against 4.51 ms)
    const char* e = getenv("WN_GEMM_BST");
    if (e && e[0] == '0') return 0;
    const int ntx = (k.t_hi - k.t_base + BST_COLS - 1) / BST_COLS;
    const int ntiles = ntx * batch;
    const int npass = (k.mt / 3 + 7) / 8;
    int grid = wn_num_cus();
    if (grid > ntiles || !(e && e[0] == '1')) grid = ntiles;
    if (mode == WN_MODE_BF16X3) bst_launch<BF16, 3, 3, 8>(k, ntx, ntiles, npass, grid, st);
    else bst_launch<F16, 3, 3, 8>(k, ntx, ntiles, npass, grid, st);
    return 1;
}
