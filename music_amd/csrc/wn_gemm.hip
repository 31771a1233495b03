// Generic channel-mixing GEMM over the time axis (time on the MFMA lanes) and the weight packer.
//
//   out[m][t + out_shift] = resid[m][t] + mask( bias[m] + sum_tap sum_k W[m][tap,k] * pre(in_tap[k][t + shift_tap]) )
//
// Used for: causal conv (wavenet/model.py:104, two taps of K=Q), the skip product over the
// concatenated z-crops and post_process_1/2 (model.py:128-138), and every "weights transposed"
// data-gradient product of the backward pass.  The fused residual-block kernels live in
// wn_resblock.hip; this kernel is the unfused workhorse around them.
#include <type_traits>
#include "wn_common.h"
#include "wn_kernels.h"

// ---------------------------------------------------------------------------------------------
// Weight packing: flat fp32 parameters -> fragment-ordered 16-bit hi/lo pairs.
// idx[p] (p in logical order [frag][lane][j]) is the offset of the source weight in the flat
// parameter buffer, or -1 for a structural zero (padding).  Built once on the host.
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void pack_weights_k(const float* __restrict__ flat, const int32_t* __restrict__ idx,
                               uint16_t* __restrict__ out, int n, int ns) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int src = idx[p];
    float v = src >= 0 ? flat[src] : 0.0f;
    typename T::elem h = T::cvt(v);
    int frag = p >> 9, r = p & 511;
    size_t o = (size_t)frag * (ns == 3 ? 1024 : 512) + r;
    out[o] = __builtin_bit_cast(uint16_t, h);
    if (ns == 3) {
        typename T::elem l = T::cvt(v - T::back(h));
        out[o + 512] = __builtin_bit_cast(uint16_t, l);
    }
}

int wn_launch_pack(const float* flat, const int32_t* idx, uint16_t* out, int n, int is_bf16, int ns,
                   hipStream_t st) {
    if (n <= 0) return 0;
    dim3 g((n + 255) / 256), b(256);
    if (is_bf16) hipLaunchKernelGGL(pack_weights_k<BF16>, g, b, 0, st, flat, idx, out, n, ns);
    else hipLaunchKernelGGL(pack_weights_k<F16>, g, b, 0, st, flat, idx, out, n, ns);
    WN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// chan_gemm: WG = 8 waves arranged WM (row groups) x 8/WM (column groups of 64).  A wave owns
// MTW 16-row tiles x 64 columns (accumulators MTW x 4 x f32x4):
//   wide   (MTW 8, WM 2): 256 rows x 256 columns per workgroup - every byte of the activation
//          tile and of the packed weights is fetched once per workgroup (L1 serves the sharing
//          waves), which is what bounds the K = 1920 skip product and its transposes;
//   narrow (MTW 4, WM 1): 64 rows x 512 columns per workgroup (causal conv, per-layer dx).
// blockIdx.y = row group, blockIdx.z = clip.  Activations stream HBM -> registers two k-steps
// ahead; weight fragments come from L2 one tile ahead of the MFMAs that use them.
// ---------------------------------------------------------------------------------------------
template <class T, int NS, int MTW, int WM>
__global__ __launch_bounds__(512) void chan_gemm_k(WnGemmArgs a) {
    constexpr int WN = 8 / WM;
    constexpr int PF = MTW > 4 ? 1 : 2;        // k-steps of activations in flight (register budget)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const WnBlock blk = wn_block(a.swz);
    const int b = blk.z;
    const int wm = wave / WN, wn = wave % WN;
    const int t0 = a.t_base + (blk.x * WN + wn) * 64;            // first column of this wave
    const int tl = t0 + 4 * c;                                    // this lane's first column
    const int m0 = (blk.y * WM + wm) * MTW;                       // first M-tile of this wave
    if (t0 >= a.t_hi || m0 >= a.mt) return;
    const int KS = a.ks0 + a.ks1;

    f32x4 acc[MTW][4];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        f32x4 init = {0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = (m0 + m) * 16 + 4 * q + i;
                init[i] = row < a.m_valid ? a.bias[row] : 0.f;
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = init;
    }

    const float* in0 = a.in0 + (size_t)b * a.in_bstride;
    const float* in1 = a.in1 ? a.in1 + (size_t)b * a.in_bstride : nullptr;
    const int col0 = tl + a.shift0, col1 = tl + a.shift1;

    // wave-uniform: all 64 columns of a tap lie inside [in_lo, in_hi) -> plain 16-B loads
    const int t0u = __builtin_amdgcn_readfirstlane(t0);
    const bool inner0 = t0u + a.shift0 >= a.in_lo && t0u + 64 + a.shift0 <= a.in_hi;
    const bool inner1 = t0u + a.shift1 >= a.in_lo && t0u + 64 + a.shift1 <= a.in_hi;
    // Two taps of equal length are walked INTERLEAVED (tap 0 block j, tap 1 block j, ...): the two
    // reads of one channel block (columns t + shift0 and t + shift1) are then back to back, so for
    // small shift differences the second one finds its lines in L1/L2 instead of HBM.
    const bool weave = a.ks1 == a.ks0;
    auto kmap = [&](int s) { return weave ? ((s & 1) ? a.ks0 + (s >> 1) : (s >> 1)) : s; };
    auto issue = [&](f32x4* raw, int s0) {
        const int s = kmap(s0);
        const float* base; int col; int ch; bool inner;
        if (s < a.ks0) { base = in0; col = col0; ch = s * 32; inner = inner0; }
        else { base = in1; col = col1; ch = (s - a.ks0) * 32; inner = inner1; }
        const float* p = base + (size_t)(ch + 8 * q) * a.in_pitch + col;
        if (inner) {
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = ld4u(p + (size_t)j * a.in_pitch);
        } else {     // input columns outside [in_lo, in_hi) read as 0 and are never dereferenced
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = ld4g(p + (size_t)j * a.in_pitch, col, a.in_lo, a.in_hi);
        }
    };
    auto step = [&](f32x4* raw, int s0) {
        const int s = kmap(s0);
        Frag<T> bf[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = raw[j][n];
                v[j] = a.relu_in ? fmaxf(x, 0.f) : x;
            }
            split8<T, NS>(bf[n], v);
        }
        if (s0 + PF < KS) issue(raw, s0 + PF);
        Frag<T> af[2];
        load_a<T, NS>(af[0], a.wpack, m0 * KS + s, lane);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            if (m + 1 < MTW && m0 + m + 1 < a.mt) load_a<T, NS>(af[(m + 1) & 1], a.wpack, (m0 + m + 1) * KS + s, lane);
            if (m0 + m < a.mt) {
#pragma unroll
                for (int n = 0; n < 4; ++n) mma<T, NS>(acc[m][n], af[m & 1], bf[n]);
            }
        }
    };
    // An unmasked residual is added into the accumulators UP FRONT: its loads then share the start-up
    // latency of the first k-steps' loads instead of being waited for alone after the last MFMA
    // (9 us of a 40 us per-layer dx product).
    const bool early_resid = a.resid != nullptr && a.mask == nullptr;
    auto add_resid = [&]() {
        const float* resid = a.resid + (size_t)b * a.resid_bstride;
        const int r_lo = a.resid_lo > a.t_lo ? a.resid_lo : a.t_lo;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            if (m0 + m >= a.mt) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (m0 + m) * 16 + 4 * q + i;
                if (row >= a.m_valid) continue;
                const float* rp = resid + (size_t)row * a.resid_pitch + tl;
                const f32x4 r = ld4g(rp, tl, r_lo, a.t_hi);
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n][i] += r[n];
            }
        }
    };
    if (PF == 2) {
        f32x4 raw0[8], raw1[8];
        issue(raw0, 0);
        if (KS > 1) issue(raw1, 1);
        if (early_resid) add_resid();
        for (int s = 0; s < KS; s += 2) {
            step(raw0, s);
            if (s + 1 < KS) step(raw1, s + 1);
        }
    } else {
        f32x4 raw0[8];
        issue(raw0, 0);
        if (early_resid) add_resid();
        for (int s = 0; s < KS; ++s) step(raw0, s);
    }

    float* out = a.out + (size_t)b * a.out_bstride;
    const float* resid = (a.resid && !early_resid) ? a.resid + (size_t)b * a.resid_bstride : nullptr;
    const float* mask = a.mask ? a.mask + (size_t)b * a.mask_bstride : nullptr;
    const bool full = tl >= a.t_lo && tl + 3 < a.t_hi;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        if (m0 + m >= a.mt) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = (m0 + m) * 16 + 4 * q + i;
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

// ---------------------------------------------------------------------------------------------
// Wide variant 2: BOTH operands of a k-step are shared through LDS as ready-to-use 16-bit hi/lo
// fragments.  Each activation element is fetched from HBM and split ONCE per workgroup (every wave
// converts 1/8 of the 32 x 256 k-step slab: 4 float4 per lane) instead of once per row-half wave,
// which leaves registers for a second k-step of activations in flight per wave (the kernel is a
// streaming one: 0.8 GB of z-crops against 0.3 TFLOP for the skip product, so bytes in flight are
// what sets its speed).  LDS: 2 stages x (16 A + 16 B fragments) = 128 KB in the x3 modes.
// 256 rows x 256 columns per workgroup, 8 waves = 2 (rows) x 4 (column groups of 64).
// ---------------------------------------------------------------------------------------------
template <class T, int NS>
__global__ __launch_bounds__(512) void chan_gemm_wide2_k(WnGemmArgs a) {
    constexpr int MTW = 8, WN = 4;
    constexpr int FR = (NS == 3 ? 1024 : 512);               // halfs per fragment
    constexpr int FRV = FR / 8;                               // u32x4 per fragment
    constexpr int STAGE = 32 * FR;                            // halfs per stage: 16 A then 16 B fragments
    constexpr int PER_A = 16 * FRV / 512;                     // u32x4 of the A image per thread
    extern __shared__ __attribute__((aligned(16))) uint16_t l_s[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const WnBlock blk = wn_block<true>(a.swz);
    const int b = blk.z;
    const int wm = wave / WN, wn = wave % WN;
    const int tile0 = a.t_base + blk.x * 256;
    const int t0 = tile0 + wn * 64;
    const int tl = t0 + 4 * c;
    const int mg0 = blk.y * 16;
    const int m0 = mg0 + wm * MTW;
    const int KS = a.ks0 + a.ks1;

    f32x4 acc[MTW][4];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        f32x4 init = {0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = (m0 + m) * 16 + 4 * q + i;
                init[i] = row < a.m_valid ? a.bias[row] : 0.f;
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = init;
    }
    // loader role of this wave for the activations: column group lg, row half lh of every k-step
    const int lg = wave & 3, lh = wave >> 2;
    const int ltl = tile0 + lg * 64 + 4 * c;
    const float* in0 = a.in0 + (size_t)b * a.in_bstride;
    const float* in1 = a.in1 ? a.in1 + (size_t)b * a.in_bstride : nullptr;
    const int tg0 = tile0 + lg * 64;
    const bool inner0 = tg0 + a.shift0 >= a.in_lo && tg0 + 64 + a.shift0 <= a.in_hi;
    const bool inner1 = tg0 + a.shift1 >= a.in_lo && tg0 + 64 + a.shift1 <= a.in_hi;
    auto load_b = [&](f32x4* raw, int s) {
        const float* base; int col; int ch; bool inner;
        if (s < a.ks0) { base = in0; col = ltl + a.shift0; ch = s * 32; inner = inner0; }
        else { base = in1; col = ltl + a.shift1; ch = (s - a.ks0) * 32; inner = inner1; }
        const float* p = base + (size_t)(ch + 8 * q + 4 * lh) * a.in_pitch + col;
        if (inner) {
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[j] = ld4u(p + (size_t)j * a.in_pitch);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[j] = ld4g(p + (size_t)j * a.in_pitch, col, a.in_lo, a.in_hi);
        }
    };
    // split 4 rows x 4 columns and write the 8-byte hi / lo pieces of the 4 B fragments of group lg
    auto store_b_as = [&](const f32x4* raw, int st, auto relu) {
        uint16_t* bb = l_s + (size_t)st * STAGE + (size_t)(16 + lg * 4) * FR + lane * 8 + lh * 4;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = decltype(relu)::value ? fmaxf(raw[j][n], 0.f) : raw[j][n];
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
    };
    auto store_b = [&](const f32x4* raw, int st) {
        if (a.relu_in) store_b_as(raw, st, std::true_type());
        else store_b_as(raw, st, std::false_type());
    };
    u32x4 wreg[PER_A];
    auto load_w = [&](int s) {
#pragma unroll
        for (int i = 0; i < PER_A; ++i) {
            int v = threadIdx.x + i * 512;
            int mt = v / FRV, r = v % FRV;
            u32x4 z = {0u, 0u, 0u, 0u};
            if (mg0 + mt < a.mt) z = reinterpret_cast<const u32x4*>(a.wpack)[((size_t)(mg0 + mt) * KS + s) * FRV + r];
            wreg[i] = z;
        }
    };
    auto store_w = [&](int st) {
        u32x4* d = reinterpret_cast<u32x4*>(l_s + (size_t)st * STAGE);
#pragma unroll
        for (int i = 0; i < PER_A; ++i) d[threadIdx.x + i * 512] = wreg[i];
    };
    f32x4 raw0[4], raw1[4];
    load_w(0);
    load_b(raw0, 0);
    if (KS > 1) load_b(raw1, 1);
    store_w(0);
    store_b(raw0, 0);
    if (KS > 1) load_w(1);
    if (KS > 2) load_b(raw0, 2);
    __syncthreads();
    // one k-step: MFMAs on stage s & 1, meanwhile the next stage is filled from the registers whose
    // loads were issued two k-steps ago, and those registers are re-armed two (A: one) k-steps ahead.
    // (Tidier forms of this loop - branch-free k-steps, the fill spread over the row tiles, a
    // duplicated loop for edge waves - all measured SLOWER: 405-418 us against 355 us for the skip
    // product)
    auto step = [&](int s, f32x4* rnext) {
        const uint16_t* la = l_s + (size_t)(s & 1) * STAGE;
        const uint16_t* lb = la + (size_t)(16 + wn * 4) * FR;
        Frag<T> bf[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) load_a<T, NS>(bf[n], lb, n, lane);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            Frag<T> af;
            load_a<T, NS>(af, la, wm * MTW + m, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n) mma<T, NS>(acc[m][n], af, bf[n]);
            if (m == 3 && s + 1 < KS) {
                store_b(rnext, (s + 1) & 1);
                store_w((s + 1) & 1);
                if (s + 3 < KS) load_b(rnext, s + 3);
                if (s + 2 < KS) load_w(s + 2);
            }
        }
        __syncthreads();
    };
    for (int s = 0; s < KS; s += 2) {
        step(s, raw1);
        if (s + 1 < KS) step(s + 1, raw0);
    }
    if (t0 >= a.t_hi || m0 >= a.mt) return;

    float* out = a.out + (size_t)b * a.out_bstride;
    const float* resid = a.resid ? a.resid + (size_t)b * a.resid_bstride : nullptr;
    const float* mask = a.mask ? a.mask + (size_t)b * a.mask_bstride : nullptr;
    const bool full = tl >= a.t_lo && tl + 3 < a.t_hi;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        if (m0 + m >= a.mt) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = (m0 + m) * 16 + 4 * q + i;
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

// ---------------------------------------------------------------------------------------------
// Wide variant 3 = variant 2 with v_mfma_f32_32x32x16 (x3 modes): the same LDS fragments, read as 32-row / 32-column
// operands (tools/micro/mfmarate.hip: one wave per SIMD issues either shape every ~14 ns and this one does twice the
// arithmetic; the 16x16x32 form of this kernel ran AT that shape's ceiling).  A wave's 128 x 64 tile = 4 x 2 tiles of
// 32 x 32, 48 MFMAs per k-step instead of 96.
// ---------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) G2U { float v[2]; };   // 8-B load / store, 4-B aligned

template <class T, int NS>
static int launch_gemm(const WnGemmArgs& k, int batch, hipStream_t st) {
    const int ncol = k.t_hi - k.t_base;
    if (k.mt <= 4) {                     // narrow: 64 rows x 512 columns per workgroup
        dim3 g((ncol + 511) / 512, 1, batch), b(512);
        hipLaunchKernelGGL((chan_gemm_k<T, NS, 4, 1>), g, b, 0, st, k);
    } else {                             // wide: 256 rows x 256 columns per workgroup
        dim3 g((ncol + 255) / 256, (k.mt + 15) / 16, batch), b(512);
        const size_t sh = (size_t)2 * 32 * (NS == 3 ? 1024 : 512) * sizeof(uint16_t);
        static WnDevOnce done;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (done.need(dev)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chan_gemm_wide2_k<T, NS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            done.done(dev);
        }
        hipLaunchKernelGGL((chan_gemm_wide2_k<T, NS>), g, b, sh, st, k);
    }
    return 0;
}

int wn_launch_gemm(const WnGemmArgs& a, int batch, int mode, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    WnGemmArgs k = a;
    k.t_base = wn_tile_origin(a.t_lo);
    k.swz = wn_xcd_swizzle_enabled();
    if (wn_launch_gemm_rw(k, batch, mode, st) || wn_launch_gemm_bst(k, batch, mode, st)) {
        WN_CHECK_LAUNCH();
        return 0;
    }
    switch (mode) {
        case WN_MODE_F16X3: launch_gemm<F16, 3>(k, batch, st); break;
        case WN_MODE_F16X1: launch_gemm<F16, 1>(k, batch, st); break;
        case WN_MODE_BF16X3: launch_gemm<BF16, 3>(k, batch, st); break;
        case WN_MODE_BF16X1: launch_gemm<BF16, 1>(k, batch, st); break;
        default: return wn_set_error_msg(-2, "wn_launch_gemm: bad mode");
    }
    WN_CHECK_LAUNCH();
    return 0;
}
