// Wide channel-mixing product (>= 256 rows, one tap, x3 modes) with ONE wave per SIMD: 256 rows x 256 columns per workgroup,
// four waves of 512 registers, wave w = all 256 rows x columns 64w .. 64w + 63 (accumulators 16 x 4 tiles = 256 registers).
//
// Why (round 5, profiles/r05_gemm.md): chan_gemm_wide2_k's k-step takes ~5000 cycles at 2.3 GHz against 3072 of matrix pipe,
// and without its MFMAs still 75 % of that - its eight waves (two per SIMD, one barrier per k-step) all read fragments,
// multiply, convert and fill at the same time, the younger half loses the arbitration for the matrix pipe and then runs its
// fill with nothing beside it.  Two re-arrangements of the SAME two-waves-per-SIMD tile were built and measured no faster
// (LDS-DMA operands, one barrier; LDS-DMA + ping-pong phases half a k-step apart: tools/exp/gemm_dma.patch).  Here a SIMD has
// ONE instruction stream, so nothing is arbitrated: per k-step a wave issues 192 MFMAs and, in their shadows, everything else -
//   * B (fp32 activations): the wave's OWN 32 x 64 slab straight into registers (8 x 16-byte loads, three k-steps in flight
//     with two register sets) and split there - each element fetched and split exactly once per workgroup, NO LDS round trip
//     (wide2: 8 ds_write_b64 + 8 ds_read_b128 per thread and k-step);
//   * A (packed 16-bit hi / lo weight fragments, 32 KB per k-step, shared by the four waves): staged L2 -> registers -> LDS one
//     k-step ahead by all four waves (8 x 16 bytes per thread), read back as 16 fragments two tiles ahead of their MFMAs;
//     ring of three stages, one barrier per k-step (four waves meet instead of eight).
// LDS reads per CU and k-step: 4 x 32 KB (wide2: 8 x 24 KB).
#include <stdlib.h>
#include <type_traits>
#include "wn_common.h"
#include "wn_kernels.h"

#define W1_NST 3
#define W1_STAGE_HALFS (16 * 1024)
#define W1_LDS_BYTES (W1_NST * W1_STAGE_HALFS * 2)

template <class T, bool RELU>
__global__ __launch_bounds__(256) void chan_gemm_w1_k(WnGemmArgs a) {
    constexpr int MT = 16;
    constexpr int FRV = 128;                                   // u32x4 per fragment (hi block + lo block)
    constexpr int PER_A = MT * FRV / 256;                      // u32x4 of a k-step's A image per thread
    extern __shared__ __attribute__((aligned(16))) uint16_t l_a[];       // [W1_NST][16 fragments][1024 halfs]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, q = lane >> 4;
    const WnBlock blk = wn_block<true>(a.swz);
    const int b = blk.z;
    const int tile0 = a.t_base + blk.x * 256;
    const int t0 = tile0 + wave * 64;
    const int tl = t0 + 4 * c;
    const int mg0 = blk.y * MT;
    const int KS = a.ks0;

    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x4 init = {0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = (mg0 + m) * 16 + 4 * q + i;
                init[i] = row < a.m_valid ? a.bias[row] : 0.f;
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = init;
    }
    const float* in0 = a.in0 + (size_t)b * a.in_bstride;
    const int col = tl + a.shift0;
    const int t0u = __builtin_amdgcn_readfirstlane(t0);
    const bool inner = t0u + a.shift0 >= a.in_lo && t0u + 64 + a.shift0 <= a.in_hi;      // wave-uniform
    auto load_b = [&](f32x4* raw, int s) {
        const float* p = in0 + (size_t)(s * 32 + 8 * q) * a.in_pitch + col;
        if (inner) {
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = ld4u(p + (size_t)j * a.in_pitch);
        } else {        // input columns outside [in_lo, in_hi) read as 0 and are never dereferenced
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = ld4g(p + (size_t)j * a.in_pitch, col, a.in_lo, a.in_hi);
        }
    };
    auto split_b = [&](Frag<T>* bf, const f32x4* raw) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = RELU ? fmaxf(raw[j][n], 0.f) : raw[j][n];
            split8<T, 3>(bf[n], v);
        }
    };
    // A image of a k-step: 16 fragments x 128 u32x4; thread t moves u32x4 t + 256 i.  Row tiles beyond the matrix (the last row
    // group of a 1920-row product) read the last real tile: their results are never stored
    u32x4 wreg[PER_A];
    const u32x4* wsrc[PER_A];
#pragma unroll
    for (int i = 0; i < PER_A; ++i) {
        const int v = threadIdx.x + i * 256;
        int mt = mg0 + v / FRV;
        mt = mt < a.mt ? mt : a.mt - 1;
        wsrc[i] = reinterpret_cast<const u32x4*>(a.wpack) + (size_t)mt * KS * FRV + (v % FRV);
    }
    auto load_w = [&](int s) {
#pragma unroll
        for (int i = 0; i < PER_A; ++i) wreg[i] = wsrc[i][(size_t)s * FRV];
    };
    auto store_w = [&](int st) {
        u32x4* d = reinterpret_cast<u32x4*>(l_a + (size_t)st * W1_STAGE_HALFS);
#pragma unroll
        for (int i = 0; i < PER_A; ++i) d[threadIdx.x + i * 256] = wreg[i];
    };

    f32x4 raw0[8], raw1[8];
    Frag<T> bf0[4], bf1[4];
    // prologue: A(0) staged, A(1) requested; B(0) split, B(1) and B(2) in flight
    load_w(0);
    load_b(raw0, 0);
    if (KS > 1) load_b(raw1, 1);
    store_w(0);
    if (KS > 1) load_w(1);
    split_b(bf0, raw0);
    if (KS > 2) load_b(raw0, 2);
    __syncthreads();

    // one k-step: 192 MFMAs on (stage s % 3, bf_cur); in their shadows A(s + 1) goes registers -> LDS and A(s + 2) is requested,
    // B(s + 1) is split into bf_nxt and its registers are re-armed with B(s + 3).  raw_nxt = the registers that hold B(s + 1).
    auto step = [&](int s, Frag<T>* bf_cur, Frag<T>* bf_nxt, f32x4* raw_nxt) __attribute__((always_inline)) {
        const uint16_t* la = l_a + (size_t)(s % W1_NST) * W1_STAGE_HALFS;
        Frag<T> af[3];
        load_a<T, 3>(af[0], la, 0, lane);
        load_a<T, 3>(af[1], la, 1, lane);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m + 2 < MT) load_a<T, 3>(af[(m + 2) % 3], la, m + 2, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n) mma<T, 3>(acc[m][n], af[m % 3], bf_cur[n]);
            if (m == 2 && s + 1 < KS) {
                store_w((s + 1) % W1_NST);
                if (s + 2 < KS) load_w(s + 2);
            }
            if (m >= 4 && m < 8 && s + 1 < KS) {              // one N-tile of the next k-step per row tile
                const int n = m - 4;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = RELU ? fmaxf(raw_nxt[j][n], 0.f) : raw_nxt[j][n];
                split8<T, 3>(bf_nxt[n], v);
            }
            if (m == 8 && s + 3 < KS) load_b(raw_nxt, s + 3);
        }
        __syncthreads();
    };
    for (int s = 0; s < KS; s += 2) {
        step(s, bf0, bf1, raw1);
        if (s + 1 < KS) step(s + 1, bf1, bf0, raw0);
    }
    if (t0 >= a.t_hi) return;

    float* out = a.out + (size_t)b * a.out_bstride;
    const float* resid = a.resid ? a.resid + (size_t)b * a.resid_bstride : nullptr;
    const float* mask = a.mask ? a.mask + (size_t)b * a.mask_bstride : nullptr;
    const bool full = tl >= a.t_lo && tl + 3 < a.t_hi;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (mg0 + m >= a.mt) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = (mg0 + m) * 16 + 4 * q + i;
            if (row >= a.m_valid) continue;
            f32x4 v = {acc[m][0][i], acc[m][1][i], acc[m][2][i], acc[m][3][i]};
            if (mask) {
                const float* mp = mask + (size_t)row * a.mask_pitch + tl;
                if (full) {
                    f32x4 mv = ld4u(mp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (tl + e >= a.t_lo && tl + e < a.t_hi) v[e] = mp[e] > 0.f ? v[e] : 0.f;
                }
            }
            if (resid) {
                const float* rp = resid + (size_t)row * a.resid_pitch + tl;
                if (full && tl >= a.resid_lo) {
                    v += ld4u(rp);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (tl + e >= a.resid_lo && tl + e >= a.t_lo && tl + e < a.t_hi) v[e] += rp[e];
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

template <class T, bool RELU>
static void w1_launch(const WnGemmArgs& k, dim3 g, hipStream_t st) {
    static WnDevOnce done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done.need(dev)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chan_gemm_w1_k<T, RELU>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  W1_LDS_BYTES);
        done.done(dev);
    }
    hipLaunchKernelGGL((chan_gemm_w1_k<T, RELU>), g, dim3(256), W1_LDS_BYTES, st, k);
}

// 1 = launched, 0 = arguments not covered (the caller falls back to chan_gemm_wide2_k).  WN_GEMM_W1=1 turns it on.
int wn_launch_gemm_w1(const WnGemmArgs& k, int batch, int mode, hipStream_t st) {
    const char* e = getenv("WN_GEMM_W1");                  // (read per launch: a same-process A/B can flip it, tools/gemm_bench.py)
    const int on = (e && e[0] == '1') ? 1 : 0;             // OFF unless asked for: measured slower than chan_gemm_wide2_k (profiles/r05_gemm.md)
    if (!on || k.mt <= 4 || k.in1 != nullptr || k.ks1 != 0 || k.ks0 < 1) return 0;
    if (mode != WN_MODE_F16X3 && mode != WN_MODE_BF16X3) return 0;
    const int ncol = k.t_hi - k.t_base;
    const dim3 g((ncol + 255) / 256, (k.mt + 15) / 16, batch);
    if (mode == WN_MODE_F16X3) {
        if (k.relu_in) w1_launch<F16, true>(k, g, st); else w1_launch<F16, false>(k, g, st);
    } else {
        if (k.relu_in) w1_launch<BF16, true>(k, g, st); else w1_launch<BF16, false>(k, g, st);
    }
    return 1;
}
