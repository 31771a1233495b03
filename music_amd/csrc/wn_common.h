// Shared device helpers for the WaveNet HIP kernels (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this directory
// --------------------------------------------------
// * Activations are fp32, channels-first, time contiguous, in ABSOLUTE time: element (b, c, t) of
//   a layer buffer lives at  base + b*bstride + c*pitch + t  where t is the index of the newest
//   input sample the value depends on.  All layers therefore share one pitch and the "shifted"
//   read of a dilated tap is just  t - d  in the same coordinate system.  pitch % 4 == 0 and all
//   bases are 16-byte aligned, so a lane that owns 4 consecutive t (t % 4 == 0) moves float4s.
// * Every channel-mixing product runs on v_mfma_f32_16x16x32_{f16,bf16} with TIME ON THE LANES:
//   C[channel][time] = W[channel][k] * X[k][time].  One wave owns 64 time columns as four 16-wide
//   N-tiles; lane (c = lane&15, q = lane>>4) of N-tile n holds time  t0 + 4c + n, so the four
//   N-tiles of a lane are 4 consecutive samples (one float4 per channel row).
//     A fragment: lane holds W[16m + c][kmap(s,q,j)], j = 0..7   (8 x 16-bit = 16 B)
//     B fragment: lane holds X[kmap(s,q,j)][col c]
//     C/D       : lane holds C[16m + 4q + i][col c], i = 0..3
//   kmap is either natural (32s + 8q + j: B built from global rows) or "chained"
//   (32s + 16(j>>2) + 4q + (j&3): B built from the accumulators of a previous product, no LDS
//   round trip).  Weights are pre-packed into fragment order by wn_pack_weights.
// * fp32-grade accuracy comes from a 2-term split of both operands into 16-bit pieces
//   (x = hi + lo) and three MFMAs per product (hi*hi + lo*hi + hi*lo) accumulated in fp32
//   ("x3" modes).  The x1 modes use hi*hi only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct __attribute__((packed, aligned(4))) F4U { float v[4]; };   // 16-B load, 4-B aligned

#define WN_WAVE 64
#define WN_FRAG_HALFS 512          // 64 lanes x 8 halfs: one packed A fragment (1 KB)

// ---- 16-bit operand traits -----------------------------------------------------------------
struct F16 {
    typedef f16x8 vec8;
    typedef _Float16 elem;
    static constexpr uint32_t one16 = 0x3C00u;             // 1.0
    static __device__ __forceinline__ elem cvt(float x) { return (_Float16)x; }
    static __device__ __forceinline__ float back(elem h) { return (float)h; }
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {      // 32x32x16: twice the arithmetic per issue slot
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
struct BF16 {
    typedef bf16x8 vec8;
    typedef __bf16 elem;
    static constexpr uint32_t one16 = 0x3F80u;
    static __device__ __forceinline__ elem cvt(float x) { return (__bf16)x; }
    static __device__ __forceinline__ float back(elem h) { return (float)h; }
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

template <class T>
struct Frag {            // one operand fragment, split into hi + lo
    typename T::vec8 hi, lo;
};

// acc += A*B with NS products (1: hi*hi ; 3: hi*hi + lo*hi + hi*lo)
template <class T, int NS>
__device__ __forceinline__ void mma(f32x4& acc, const Frag<T>& a, const Frag<T>& b) {
    if (NS == 3) {
        acc = T::mfma(a.lo, b.hi, acc);
        acc = T::mfma(a.hi, b.lo, acc);
    }
    acc = T::mfma(a.hi, b.hi, acc);
}

// (a, b) -> packed 16-bit pairs hi = (cvt(a), cvt(b)) and lo = (cvt(a - hi_a), cvt(b - hi_b)), a in the low half.
// f16 is written as instructions: hipcc's own code for the same arithmetic converts every hi twice (once packed for the
// fragment, once alone for the subtraction), packs the subtractions into v_pk_add_f32 and shuffles registers into pairs
// for it - 4-5 VALU per element.  The mixed-precision FMA reads the f16 hi as an f32 operand, subtracts it from the
// f32 input and rounds the difference - which is exact in f32 - to f16 in one instruction per element: 1.5 per
// element, the same bits as cvt(x - float(hi)) (tests/test_gpu_kernels.py, wn_split16).  Worth 1 % of the step
// (-DWN_SPLIT_PLAIN is the compiler's form, for timing): the kernels are not bound by vector issue.
template <class T> __device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo);
typedef _Float16 wn_f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wn_f32x2 __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ void split2<F16>(float a, float b, uint32_t& hi, uint32_t& lo) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
}
// bf16 has no mixed form: hi back to f32 by shift / mask, six instructions per pair, and here the compiler's own code is
// the faster one (the same six written as instructions: backward stack 1.998 vs 1.963 ms on one box)
template <> __device__ __forceinline__ void split2<BF16>(float a, float b, uint32_t& hi, uint32_t& lo) {
    const wn_f32x2 v = {a, b};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, wn_bf16x2));
    const wn_f32x2 r = {a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, wn_bf16x2));
}
template <class T> __device__ __forceinline__ uint32_t cvt2(float a, float b);       // hi pair only (x1 modes)
template <> __device__ __forceinline__ uint32_t cvt2<F16>(float a, float b) {
    uint32_t h;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    return h;
}
template <> __device__ __forceinline__ uint32_t cvt2<BF16>(float a, float b) {
    uint32_t h;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    return h;
}

template <class T, int NS>
__device__ __forceinline__ void split8(Frag<T>& f, const float* v) {
    u32x4 h, l = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t hh, ll = 0u;
        if (NS == 3) split2<T>(v[2 * j], v[2 * j + 1], hh, ll);
        else hh = cvt2<T>(v[2 * j], v[2 * j + 1]);
        h[j] = hh;
        l[j] = ll;
    }
    f.hi = __builtin_bit_cast(typename T::vec8, h);
    if (NS == 3) f.lo = __builtin_bit_cast(typename T::vec8, l);
}

// Packed A fragment (hi block then lo block, 1 KB each) from global or LDS.
// layout: [(m*KS + s)][NS==3 ? 2 : 1][64 lanes][8 halfs]
template <class T, int NS>
__device__ __forceinline__ void load_a(Frag<T>& f, const uint16_t* pack, int frag_index, int lane) {
    const u32x4* p = reinterpret_cast<const u32x4*>(pack) +
                     (size_t)frag_index * (NS == 3 ? 128 : 64) + lane;
    u32x4 h = p[0];
    f.hi = __builtin_bit_cast(typename T::vec8, h);
    if (NS == 3) {
        u32x4 l = p[64];
        f.lo = __builtin_bit_cast(typename T::vec8, l);
    }
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4u(const float* p) {
    F4U u = *reinterpret_cast<const F4U*>(p);
    f32x4 r = {u.v[0], u.v[1], u.v[2], u.v[3]};
    return r;
}
// guarded 4-column load: columns outside [lo, hi) read as 0 and are never dereferenced
__device__ __forceinline__ f32x4 ld4g(const float* p, int col, int lo, int hi) {
    if (col >= lo && col + 3 < hi) return ld4u(p);
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (col + e >= lo && col + e < hi) r[e] = p[e];
    return r;
}
// store the 4 consecutive samples t..t+3 of one channel row, only those inside [lo, hi)
__device__ __forceinline__ void st4m(float* p, f32x4 v, int t, int lo, int hi) {
    if (t >= lo && t + 3 < hi) {
        *reinterpret_cast<f32x4*>(p) = v;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t + e >= lo && t + e < hi) p[e] = v[e];
    }
}

__device__ __forceinline__ float wn_sigmoid(float g) { return __builtin_amdgcn_rcpf(1.0f + __expf(-g)); }
// tanh(f) = (1 - e) / (1 + e), e = exp(-2f) (f clamped to +-15, where tanh is +-1 in fp32).
// ABSOLUTE error ~1e-7; the cancellation in 1 - e near f = 0 only costs relative accuracy, which
// nothing downstream needs (the parity bar is absolute), and it halves the VALU work of the gate.
__device__ __forceinline__ float wn_tanh(float f) {
    f = fminf(fmaxf(f, -15.f), 15.f);
    const float e = __expf(-2.0f * f);
    return (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
}

// The gate and its two derivatives for the backward kernels (round 6): z = tanh(f) sigmoid(g), dz/df = sigmoid(g) (1 - tanh^2 f),
// dz/dg = tanh(f) sigmoid(g) (1 - sigmoid(g)), from e1 = exp(-2|f|), e2 = exp(-|g|) - both <= 1 - and ONE reciprocal:
//   1 / (1 + e1) = u, 1 / (1 + e2) = w;  |tanh f| = (1 - e1) u;  1 - tanh^2 f = 4 e1 u^2;  sigmoid(g) = w (g >= 0) or e2 w;
//   sigmoid (1 - sigmoid) = e2 w^2.
// Nothing overflows (every intermediate lies in [0, 4]: no clamps), and neither `1 - th * th` nor `1 - sg` is formed by subtraction: where the
// gate saturates the derivatives go to 0 as fast as they do in exact arithmetic.  The forms they replace - exp(-2f) up to e^30 with th = (1 - e1) /
// (1 + e1) one ulp off -1, then 1 - th^2 = +-1.2e-7 in place of 4e-13, and a product (1 + e1)(1 + e2) that overflowed to inf for f <= -15, g <= -59
// (th = -inf * 0 = NaN) - cost a model whose filter / gate convs are 40 x / 150 x the usual size 1.6e-3 of its gradients (the float32 CPU path:
// 2e-6) or all of them (tests/test_gpu_parity.py: test_saturated_gates_give_finite_gradients_vs_oracle; found by tools/soak_determinism.py).
struct WnGateD { float z, dzdf, dzdg; };
__device__ __forceinline__ WnGateD wn_gate_d(float f, float g) {
    const float e1 = __builtin_amdgcn_exp2f(__builtin_fabsf(f) * -2.88539008177792681472f);
    const float e2 = __builtin_amdgcn_exp2f(__builtin_fabsf(g) * -1.44269504088896340736f);
    const float a1 = 1.0f + e1, b1 = 1.0f + e2;
    const float rr = __builtin_amdgcn_rcpf(a1 * b1);
    const float u = b1 * rr, w = a1 * rr;
    const float th = __builtin_copysignf((1.0f - e1) * u, f);
    const float sm = e2 * w;                              // sigmoid(-|g|) = 1 - sigmoid(|g|)
    const float sg = g >= 0.f ? w : sm;
    WnGateD r;
    r.z = th * sg;
    r.dzdf = sg * (4.0f * e1) * (u * u);
    r.dzdg = th * (sm * w);
    return r;
}

// XCD-aware work mapping.  Workgroups are observed to be dealt round-robin over the 8 XCDs in
// dispatch order (x fastest), so blocks with equal (linear id % 8) share one 4 MB L2.  The remap gives
// each XCD one CONTIGUOUS run of the (x, y, z) work items: neighbouring time tiles of one clip then sit
// behind the same L2, where the dilated tap (t - d), the shifted gradient tap (t + d) and a second
// launch over the same columns find the rows their neighbour fetched.  Bijective for any grid size;
// speed only, nothing depends on the placement.
struct WnBlock { int x, y, z; };
template <bool YFAST = false>      // YFAST: y is the fastest-varying work index (row groups that share columns)
__device__ __forceinline__ WnBlock wn_block(int swz) {
    WnBlock o;
    if (!swz) { o.x = blockIdx.x; o.y = blockIdx.y; o.z = blockIdx.z; return o; }
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
    const int id = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
    const int qn = nwg >> 3, rn = nwg & 7, xcd = id & 7;
    const int w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (id >> 3);
    if (YFAST) {
        o.y = w % gy;
        const int t = w / gy;
        o.x = t % gx;
        o.z = t / gx;
    } else {
        o.x = w % gx;
        const int t = w / gx;
        o.y = t % gy;
        o.z = t / gy;
    }
    return o;
}
int wn_xcd_swizzle_enabled();      // host: 1 (a switch until round 4)
int wn_num_cus();                  // host: compute units of the device (cached)
// host: first column of a launch's tiling.  A multiple of 64 samples of absolute time, so that
// the 256-byte row segment of a 64-column wave tile is exactly two 128-byte lines (pitch and bases are
// multiples of 128 B); columns in front of t_lo are masked.  Must stay a multiple of 4 (float4 lanes).
int wn_tile_origin(int t_lo);

// "this kernel instantiation has its dynamic-LDS attribute set on device d": one object per instantiation.  The C ABI promises
// re-entrancy across host threads (include/wavenet_hip.h), so the mask is atomic; two threads that race on a first launch both
// set the attribute (idempotent).  Device ordinals >= 64 have no bit: the attribute is set on every launch there.
struct WnDevOnce {
    std::atomic<unsigned long long> mask{0};
    bool need(int dev) const { return dev < 0 || dev >= 64 || !((mask.load(std::memory_order_acquire) >> dev) & 1ull); }
    void done(int dev) { if (dev >= 0 && dev < 64) mask.fetch_or(1ull << dev, std::memory_order_release); }
};

#define WN_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return wn_set_error(e_, __FILE__, __LINE__); } while (0)
int wn_set_error(hipError_t e, const char* file, int line);
int wn_set_error_msg(int code, const char* msg);
