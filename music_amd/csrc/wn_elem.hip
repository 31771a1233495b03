// Memory-bound helpers of the WaveNet path: the reference's 256-wide CHUNK softmax
// (wavenet/model.py:142-144), the CrossEntropyLoss the training loop applies to those
// probabilities (wavenet/train.py:146,179), flat Adam, the loader's one-hot layouts
// (wavenet/faster_audio_data.py:62-83), and mu-law (wavenet/audio_func.py:5-39).
#include "wn_common.h"
#include "wn_kernels.h"

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wave per 256-float row: lane holds 4 consecutive floats (one 1 KB coalesced access per wave).
__global__ __launch_bounds__(256) void softmax256_fwd_k(const float* __restrict__ x, float* __restrict__ y, long nrows) {
    const int lane = threadIdx.x & 63;
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * 4;
    for (; row < nrows; row += stride) {
        f32x4 v = ld4(x + row * 256 + lane * 4);
        float m = wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        f32x4 e = {expf(v[0] - m), expf(v[1] - m), expf(v[2] - m), expf(v[3] - m)};
        float s = wave_sum((e[0] + e[1]) + (e[2] + e[3]));
        float inv = 1.0f / s;
        f32x4 r = {e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv};
        *reinterpret_cast<f32x4*>(y + row * 256 + lane * 4) = r;
    }
}

// dx = y * (dy - <dy, y>)
__global__ __launch_bounds__(256) void softmax256_bwd_k(const float* __restrict__ y, const float* __restrict__ dy,
                                                        float* __restrict__ dx, long nrows) {
    const int lane = threadIdx.x & 63;
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * 4;
    for (; row < nrows; row += stride) {
        f32x4 p = ld4(y + row * 256 + lane * 4);
        f32x4 g = ld4(dy + row * 256 + lane * 4);
        float dot = wave_sum((p[0] * g[0] + p[1] * g[1]) + (p[2] * g[2] + p[3] * g[3]));
        f32x4 r = {p[0] * (g[0] - dot), p[1] * (g[1] - dot), p[2] * (g[2] - dot), p[3] * (g[3] - dot)};
        *reinterpret_cast<f32x4*>(dx + row * 256 + lane * 4) = r;
    }
}

// Fused: p = softmax(x_row); loss_row = logsumexp(p) - p[y]; dp = (softmax(p) - e_y) * inv_n;
// dx = p * (dp - <dp,p>).  probs / dx may be null.  loss_part[blockIdx.x] = this block's share of
// mean_r loss_r (the caller sums the WN_CE_PARTIALS partials; no contended atomics).
#define CE_WAVES 8
__global__ __launch_bounds__(64 * CE_WAVES) void softmax256_ce_k(const float* __restrict__ x, const int64_t* __restrict__ target,
                                                                 float* __restrict__ probs, float* __restrict__ dx,
                                                                 float* __restrict__ loss_part, long nrows, float inv_n) {
    __shared__ float red[CE_WAVES];
    const int lane = threadIdx.x & 63;
    long row = (long)blockIdx.x * CE_WAVES + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * CE_WAVES;
    float lacc = 0.f;
    // the next row is on its way while this one goes through its five wave reductions (unconditional load of a
    // clamped row index: a load under a run-time condition would not be a prefetch)
    long rc = row < nrows ? row : nrows - 1;
    f32x4 v = ld4(x + rc * 256 + lane * 4);
    long y = target[rc];
    for (; row < nrows; row += stride) {
        const long rn = row + stride < nrows ? row + stride : nrows - 1;
        const f32x4 vn = ld4(x + rn * 256 + lane * 4);
        const long yn = target[rn];
        float m = wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        f32x4 e = {expf(v[0] - m), expf(v[1] - m), expf(v[2] - m), expf(v[3] - m)};
        float inv = 1.0f / wave_sum((e[0] + e[1]) + (e[2] + e[3]));
        f32x4 p = {e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv};
        if (probs) *reinterpret_cast<f32x4*>(probs + row * 256 + lane * 4) = p;
        // second (log-)softmax over the probabilities: p in [0,1] so no max shift is needed
        f32x4 e2 = {expf(p[0]), expf(p[1]), expf(p[2]), expf(p[3])};
        float s2 = wave_sum((e2[0] + e2[1]) + (e2[2] + e2[3]));
        // a target outside [0, 256) (nn.CrossEntropyLoss raises "Target out of bounds" for it) poisons the loss and this
        // row's gradient with NaN instead of silently reading another lane's probability: corrupt data stays visible
        // in the loss log without a device -> host check per step
        const bool bad = (unsigned long)y > 255ul;
        const int yl = (int)(y >> 2) & 63, ye = (int)y & 3;
        float py = __shfl(ye == 0 ? p[0] : ye == 1 ? p[1] : ye == 2 ? p[2] : p[3], yl, 64);
        lacc += bad ? __builtin_nanf("") : logf(s2) - py;
        if (dx) {
            float is2 = inv_n / s2;
            f32x4 dp = {e2[0] * is2, e2[1] * is2, e2[2] * is2, e2[3] * is2};
            if (lane == yl) dp[ye] -= inv_n;
            float dot = wave_sum((p[0] * dp[0] + p[1] * dp[1]) + (p[2] * dp[2] + p[3] * dp[3]));
            f32x4 r = {p[0] * (dp[0] - dot), p[1] * (dp[1] - dot), p[2] * (dp[2] - dot), p[3] * (dp[3] - dot)};
            if (bad) r = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
            *reinterpret_cast<f32x4*>(dx + row * 256 + lane * 4) = r;
        }
        v = vn;
        y = yn;
    }
    if (lane == 0) red[threadIdx.x >> 6] = lacc * inv_n;
    __syncthreads();
    if (threadIdx.x == 0 && loss_part) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < CE_WAVES; ++w) t += red[w];
        loss_part[blockIdx.x] = t;
    }
}

static inline int sm_grid(long nrows) {
    long g = (nrows + 3) / 4;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
int wn_launch_softmax_fwd(const float* x, float* y, long nrows, hipStream_t st) {
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(softmax256_fwd_k, dim3(sm_grid(nrows)), dim3(256), 0, st, x, y, nrows);
    WN_CHECK_LAUNCH();
    return 0;
}
int wn_launch_softmax_bwd(const float* y, const float* dy, float* dx, long nrows, hipStream_t st) {
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(softmax256_bwd_k, dim3(sm_grid(nrows)), dim3(256), 0, st, y, dy, dx, nrows);
    WN_CHECK_LAUNCH();
    return 0;
}
int wn_launch_softmax_ce(const float* x, const int64_t* target, float* probs, float* dx, float* loss_part,
                         long nrows, float inv_n, hipStream_t st) {
    if (nrows <= 0) return 0;
    // always WN_CE_PARTIALS blocks so that every partial is (re)written each call
    hipLaunchKernelGGL(softmax256_ce_k, dim3(WN_CE_PARTIALS), dim3(64 * CE_WAVES), 0, st, x, target, probs, dx,
                       loss_part, nrows, inv_n);
    WN_CHECK_LAUNCH();
    return 0;
}

// out[b][r][t] = p[b][r][t] (t >= p_lo) + q[b][r][t + dn] (t + dn < t_hi), t in [t_lo, t_hi): materialises a data
// gradient that the one-launch backward block (wn_respq.hip) hands on as an unshifted (P, Q) pair.
__global__ void shift_add_k(const float* __restrict__ p, const float* __restrict__ q, float* __restrict__ out,
                            long bstride, int pitch, int dn, int p_lo, int t_lo, int t_hi) {
    const int t = t_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= t_hi) return;
    const size_t o = (size_t)blockIdx.z * bstride + (size_t)blockIdx.y * pitch + t;
    const float a = t >= p_lo ? p[o] : 0.f;
    const float b = t + dn < t_hi ? q[o + dn] : 0.f;
    out[o] = a + b;
}
int wn_launch_shift_add(const float* p, const float* q, float* out, long bstride, int pitch, int rows, int dn,
                        int p_lo, int t_lo, int t_hi, int batch, hipStream_t st) {
    if (t_hi <= t_lo || rows <= 0 || batch <= 0) return 0;
    hipLaunchKernelGGL(shift_add_k, dim3((t_hi - t_lo + 255) / 256, rows, batch), dim3(256), 0, st, p, q, out, bstride,
                       pitch, dn, p_lo, t_lo, t_hi);
    WN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Flat Adam (torch.optim.Adam semantics, wavenet/train.py:39-42): g is pre-scaled by gscale
// (1/world_size after the all-reduce); bc1 = 1 - b1^t, bc2 = 1 - b2^t are supplied by the host.
// ---------------------------------------------------------------------------------------------
__global__ void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float bc1,
                       float bc2, float gscale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    const float step = lr / bc1, rs = 1.0f / sqrtf(bc2);
    for (; i < n; i += stride) {
        float gi = g[i] * gscale;
        float mi = b1 * m[i] + (1.0f - b1) * gi;
        float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step * mi / (sqrtf(vi) * rs + eps);
    }
}
int wn_launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                   float eps, float bc1, float bc2, float gscale, hipStream_t st) {
    if (n <= 0) return 0;
    long grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(adam_k, dim3((int)grid), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2, gscale);
    WN_CHECK_LAUNCH();
    return 0;
}

// Flat SGD with momentum and flat RMSprop (torch.optim.SGD / torch.optim.RMSprop semantics as wavenet/train.py:28-38 constructs them:
// lr + momentum, every other argument at its default - no dampening, no Nesterov, no weight decay, RMSprop alpha / eps given, not
// centered).  g is pre-scaled by gscale (1 / world_size after the all-reduce).
//   SGD      buf = first ? g : momentum * buf + g ;  p -= lr * buf            (momentum == 0: p -= lr * g, buf untouched)
//   RMSprop  sq = alpha * sq + (1 - alpha) * g * g ; avg = sqrt(sq) + eps ;
//            momentum > 0: buf = momentum * buf + g / avg ; p -= lr * buf     else  p -= lr * g / avg
__global__ void sgd_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n, float lr, float momentum,
                      float gscale, int first) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float gi = g[i] * gscale;
        if (momentum != 0.f) {
            gi = first ? gi : buf[i] * momentum + gi;
            buf[i] = gi;
        }
        p[i] += -lr * gi;
    }
}
int wn_launch_sgd(float* p, const float* g, float* buf, long n, float lr, float momentum, float gscale, int first, hipStream_t st) {
    if (n <= 0) return 0;
    long grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(sgd_k, dim3((int)grid), dim3(256), 0, st, p, g, buf, n, lr, momentum, gscale, first);
    WN_CHECK_LAUNCH();
    return 0;
}
__global__ void rmsprop_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, float* __restrict__ buf, long n,
                          float lr, float alpha, float eps, float momentum, float gscale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float s = sq[i] * alpha + ((1.0f - alpha) * gi) * gi;
        sq[i] = s;
        const float avg = sqrtf(s) + eps;
        if (momentum > 0.f) {
            const float b = buf[i] * momentum + gi / avg;
            buf[i] = b;
            p[i] += -lr * b;
        } else {
            p[i] += -lr * (gi / avg);
        }
    }
}
int wn_launch_rmsprop(float* p, const float* g, float* sq, float* buf, long n, float lr, float alpha, float eps, float momentum,
                      float gscale, hipStream_t st) {
    if (n <= 0) return 0;
    long grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(rmsprop_k, dim3((int)grid), dim3(256), 0, st, p, g, sq, buf, n, lr, alpha, eps, momentum, gscale);
    WN_CHECK_LAUNCH();
    return 0;
}

// flat_grad[i] = packed[idx[i]]  (idx < 0: structural zero).  Maps the dense C matrices written
// by wgrad into the reference's (out, in, k) state_dict layout.
__global__ void gather_grads_k(const float* __restrict__ packed, const int32_t* __restrict__ idx,
                               float* __restrict__ flat, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flat[i] = idx[i] >= 0 ? packed[idx[i]] : 0.f;
}
// ... with a second source per element: flat[i] = packed[idx[i]] + packed[idx2[i]] (either < 0: nothing).  A weight that sits
// in two places of a block-diagonal effective matrix (two clips of a 32-channel model side by side on the 64-channel block
// kernels, music_amd/engine.py) has its gradient in both.
__global__ void gather_grads2_k(const float* __restrict__ packed, const int32_t* __restrict__ idx, const int32_t* __restrict__ idx2,
                                float* __restrict__ flat, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flat[i] = (idx[i] >= 0 ? packed[idx[i]] : 0.f) + (idx2[i] >= 0 ? packed[idx2[i]] : 0.f);
}
int wn_launch_gather_grads2(const float* packed, const int32_t* idx, const int32_t* idx2, float* flat_grad, int n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gather_grads2_k, dim3((n + 255) / 256), dim3(256), 0, st, packed, idx, idx2, flat_grad, n);
    WN_CHECK_LAUNCH();
    return 0;
}
int wn_launch_gather_grads(const float* packed, const int32_t* idx, float* flat_grad, int n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gather_grads_k, dim3((n + 255) / 256), dim3(256), 0, st, packed, idx, flat_grad, n);
    WN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// One-hot construction on device from int32 codes (B, T) -> float32 (B, Q, T).
//   scrambled = 1: faster_audio_data.py:77-81 layout, one for sample s at flat offset s*Q + code
//   scrambled = 0: textbook layout, out[code][s] = 1  (fast_generate.py:159-160)
// The output must have been zero-filled (the launcher does it with hipMemsetAsync).
// ---------------------------------------------------------------------------------------------
__global__ void onehot_k(const int32_t* __restrict__ idx, float* __restrict__ out, int q, int t, int scrambled, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    long b = i / t, s = i % t;
    int code = idx[i];
    if (code < 0 || code >= q) return;
    long pos = scrambled ? (s * q + code) : ((long)code * t + s);
    out[b * (long)q * t + pos] = 1.0f;
}
int wn_launch_onehot(const int32_t* idx, float* out, int batch, int q, int t, int scrambled, hipStream_t st) {
    long total = (long)batch * t;
    if (total <= 0) return 0;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)total * q * sizeof(float), st);
    if (e != hipSuccess) return wn_set_error(e, __FILE__, __LINE__);
    hipLaunchKernelGGL(onehot_k, dim3((int)((total + 255) / 256)), dim3(256), 0, st, idx, out, q, t, scrambled, total);
    WN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// mu-law.  encode: code = #{k : thr[k] <= a} over the 255 float32 decision thresholds of
// audio_func.mu_law_encode (bit-exact by construction); decode: 256-entry table.
// ---------------------------------------------------------------------------------------------
__global__ void mulaw_encode_k(const float* __restrict__ audio, const float* __restrict__ thr,
                               uint8_t* __restrict__ codes, long n) {
    __shared__ float s_thr[256];
    if (threadIdx.x < 255) s_thr[threadIdx.x] = thr[threadIdx.x];
    __syncthreads();
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float a = audio[i];
        int lo = 0, hi = 255;                 // count of thresholds <= a, by bisection
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (s_thr[mid] <= a) lo = mid + 1; else hi = mid;
        }
        codes[i] = (uint8_t)lo;
    }
}
__global__ void mulaw_decode_k(const uint8_t* __restrict__ codes, const float* __restrict__ table,
                               float* __restrict__ audio, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) audio[i] = table[codes[i]];
}
// any number of quantisation channels (wavenet/audio_func.py takes it as an argument): n_thr = q - 1 thresholds in global memory
// (L1 / L2 resident), int32 codes
__global__ void mulaw_encode_q_k(const float* __restrict__ audio, const float* __restrict__ thr, int n_thr,
                                 int32_t* __restrict__ codes, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float a = audio[i];
        int lo = 0, hi = n_thr;               // count of thresholds <= a
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (thr[mid] <= a) lo = mid + 1; else hi = mid;
        }
        codes[i] = lo;
    }
}
__global__ void mulaw_decode_q_k(const int32_t* __restrict__ codes, const float* __restrict__ table, int q,
                                 float* __restrict__ audio, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int c = codes[i];
        audio[i] = table[c < 0 ? 0 : (c >= q ? q - 1 : c)];
    }
}
int wn_launch_mulaw_encode_q(const float* audio, const float* thr, int n_thr, int32_t* codes, long n, hipStream_t st) {
    if (n <= 0) return 0;
    long grid = (n + 255) / 256; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(mulaw_encode_q_k, dim3((int)grid), dim3(256), 0, st, audio, thr, n_thr, codes, n);
    WN_CHECK_LAUNCH();
    return 0;
}
int wn_launch_mulaw_decode_q(const int32_t* codes, const float* table, int q, float* audio, long n, hipStream_t st) {
    if (n <= 0) return 0;
    long grid = (n + 255) / 256; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(mulaw_decode_q_k, dim3((int)grid), dim3(256), 0, st, codes, table, q, audio, n);
    WN_CHECK_LAUNCH();
    return 0;
}
int wn_launch_mulaw_encode(const float* audio, const float* thr, uint8_t* codes, long n, hipStream_t st) {
    if (n <= 0) return 0;
    long grid = (n + 255) / 256; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(mulaw_encode_k, dim3((int)grid), dim3(256), 0, st, audio, thr, codes, n);
    WN_CHECK_LAUNCH();
    return 0;
}
int wn_launch_mulaw_decode(const uint8_t* codes, const float* table, float* audio, long n, hipStream_t st) {
    if (n <= 0) return 0;
    long grid = (n + 255) / 256; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(mulaw_decode_k, dim3((int)grid), dim3(256), 0, st, codes, table, audio, n);
    WN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// bias gradient: out[row] = sum_{b, t in [t_lo,t_hi)} a[b][row][t + a_shift]; one WG per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_grad_k(const float* __restrict__ a, long a_bstride, int a_pitch, int a_shift,
                                                   int t_lo, int t_hi, int batch, float* __restrict__ out) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < batch; ++b) {
        const float* p = a + (size_t)b * a_bstride + (size_t)row * a_pitch + a_shift;
        for (int t = t_lo + threadIdx.x; t < t_hi; t += 256) s += p[t];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[row] = (red[0] + red[1]) + (red[2] + red[3]);
}
int wn_launch_bias_grad(const float* a, long a_bstride, int a_pitch, int a_shift, int rows, int t_lo,
                        int t_hi, int batch, float* out, hipStream_t st) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(bias_grad_k, dim3(rows), dim3(256), 0, st, a, a_bstride, a_pitch, a_shift, t_lo, t_hi, batch, out);
    WN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// AvgPool1d(pool) over the time axis (wavenet_autoencoder/model1.py:154-155): one wave per output.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_k(const float* __restrict__ in, long in_bstride, int in_pitch, int t0,
                                                 int pool, int n_out, int rows, float* __restrict__ out,
                                                 long out_bstride, int out_pitch) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (item >= (long)rows * n_out) return;
    const int row = (int)(item / n_out), j = (int)(item % n_out);
    const float* p = in + (size_t)b * in_bstride + (size_t)row * in_pitch + t0 + (size_t)j * pool;
    float s = 0.f;
    for (int k = lane; k < pool; k += 64) s += p[k];
    s = wave_sum(s);
    if (lane == 0) out[(size_t)b * out_bstride + (size_t)row * out_pitch + j] = s / (float)pool;
}
int wn_launch_avgpool(const float* in, long in_bstride, int in_pitch, int t0, int pool, int n_out, int rows,
                      float* out, long out_bstride, int out_pitch, int batch, hipStream_t st) {
    if (rows <= 0 || n_out <= 0 || batch <= 0) return 0;
    long items = (long)rows * n_out;
    hipLaunchKernelGGL(avgpool_k, dim3((unsigned)((items + 3) / 4), batch), dim3(256), 0, st, in, in_bstride, in_pitch, t0,
                       pool, n_out, rows, out, out_bstride, out_pitch);
    WN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Gradient w.r.t. the autoencoder's conditioning table (model1.py:227-247): bucket sums of a row
// over time.  One workgroup (4 waves) per (row, clip).  No float atomics: every partial sum has a
// fixed owner and the partials are combined in a fixed order, so the result is bit-reproducible.
//   mode 1 (stretch, bucket = (t - t_lo) / q, clamped to le-1): bucket j is the contiguous segment
//          [j q, (j+1) q) (the last one runs to the end); wave w sums buckets w, w+4, ... lane-strided.
//   mode 2 (tile, bucket = (t - t_lo) % le), le <= 64: a wave walks its quarter of the row in chunks of
//          floor(64/le)*le samples, so lane l always meets bucket l % le and keeps a private sum;
//          the 4 x floor(64/le) partials of a bucket are added in a fixed order through LDS.
//   mode 2 with le > 64 (not used by the shipped configs): one bucket per thread, strided walk.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cond_grad_k(const float* __restrict__ in, long in_bstride, int in_pitch, int t_lo,
                                                   int t_hi, int mode, int le, int q, float* __restrict__ out,
                                                   long out_bstride, int out_pitch) {
    __shared__ float part[4][64];
    const int row = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* p = in + (size_t)b * in_bstride + (size_t)row * in_pitch + t_lo;
    float* o = out + (size_t)b * out_bstride + (size_t)row * out_pitch;
    const int L = t_hi - t_lo;
    if (mode == 1) {
        for (int j = wave; j < le; j += 4) {
            const int s0 = j * q;
            const int s1 = (j == le - 1) ? L : (s0 + q < L ? s0 + q : L);
            // four loads in flight per lane, four partial sums combined in a fixed order
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int t = s0 + lane;
            for (; t + 192 < s1; t += 256) { a0 += p[t]; a1 += p[t + 64]; a2 += p[t + 128]; a3 += p[t + 192]; }
            for (; t < s1; t += 64) a0 += p[t];
            float acc = wave_sum((a0 + a1) + (a2 + a3));
            if (lane == 0) o[j] = acc;
        }
        return;
    }
    if (le <= 64) {
        const int rep = 64 / le, chunk = rep * le;
        const int nchunks = (L + chunk - 1) / chunk;
        const int per = (nchunks + 3) / 4;                     // chunks per wave
        const int c0 = wave * per, c1 = (c0 + per < nchunks) ? c0 + per : nchunks;
        float acc = 0.f;
        if (lane < chunk) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int ch = c0;
            for (; ch + 3 < c1 && (ch + 3) * chunk + lane < L; ch += 4) {
                const int t = ch * chunk + lane;
                a0 += p[t]; a1 += p[t + chunk]; a2 += p[t + 2 * chunk]; a3 += p[t + 3 * chunk];
            }
            for (; ch < c1; ++ch) {
                const int t = ch * chunk + lane;
                if (t < L) a0 += p[t];
            }
            acc = (a0 + a1) + (a2 + a3);
        }
        part[wave][lane] = lane < chunk ? acc : 0.f;
        __syncthreads();
        if (threadIdx.x < le) {
            float sacc = 0.f;
            for (int w = 0; w < 4; ++w)
                for (int r = 0; r < rep; ++r) sacc += part[w][r * le + threadIdx.x];
            o[threadIdx.x] = sacc;
        }
        return;
    }
    for (int j = threadIdx.x; j < le; j += 256) {
        float acc = 0.f;
        for (int t = j; t < L; t += le) acc += p[t];
        o[j] = acc;
    }
}
int wn_launch_cond_grad(const float* in, long in_bstride, int in_pitch, int rows, int t_lo, int t_hi, int mode, int le,
                        int q, float* out, long out_bstride, int out_pitch, int batch, hipStream_t st) {
    if (rows <= 0 || batch <= 0 || le <= 0) return 0;
    if (le > 1024) return wn_set_error_msg(-4, "cond_grad: more than 1024 buckets");
    hipLaunchKernelGGL(cond_grad_k, dim3(rows, batch), dim3(256), 0, st, in, in_bstride, in_pitch, t_lo, t_hi, mode, le,
                       q, out, out_bstride, out_pitch);
    WN_CHECK_LAUNCH();
    return 0;
}

// The conditioning term itself expanded over time (wavenet_autoencoder/model1.py:227-247, `_conditon`): out[b][row][t] =
// tab[b][row][bucket(t)] for t in [t_lo, t_hi), bucket as above.  One thread per four samples of one row.
__global__ __launch_bounds__(256) void cond_expand_k(const float* __restrict__ tab, long tab_bstride, int tab_pitch, int t_lo, int t_hi,
                                                     int mode, int le, int q, float* __restrict__ out, long out_bstride, int out_pitch) {
    const int t0 = t_lo + 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (t0 >= t_hi) return;
    const int row = blockIdx.y, b = blockIdx.z;
    const float* tr = tab + (size_t)b * tab_bstride + (size_t)row * tab_pitch;
    float* o = out + (size_t)b * out_bstride + (size_t)row * out_pitch;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int t = t0 + e;
        if (t >= t_hi) break;
        const int r = t - t_lo;
        int ix = mode == 1 ? r / q : r % le;
        ix = ix < le ? ix : le - 1;
        o[t] = tr[ix];
    }
}
int wn_launch_cond_expand(const float* tab, long tab_bstride, int tab_pitch, int rows, int t_lo, int t_hi, int mode, int le, int q,
                          float* out, long out_bstride, int out_pitch, int batch, hipStream_t st) {
    if (rows <= 0 || batch <= 0 || le <= 0 || t_hi <= t_lo) return 0;
    hipLaunchKernelGGL(cond_expand_k, dim3((t_hi - t_lo + 1023) / 1024, rows, batch), dim3(256), 0, st, tab, tab_bstride, tab_pitch,
                       t_lo, t_hi, mode, le, q, out, out_bstride, out_pitch);
    WN_CHECK_LAUNCH();
    return 0;
}

__global__ void avgpool_bwd_k(const float* __restrict__ denc, long denc_bstride, int denc_pitch, int t0, int pool,
                              int n_out, float* __restrict__ out, long out_bstride, int out_pitch, int t_hi) {
    const int t = t0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y, b = blockIdx.z;
    if (t >= t_hi) return;
    const int j = (t - t0) / pool;
    float v = 0.f;
    if (j < n_out) v = denc[(size_t)b * denc_bstride + (size_t)row * denc_pitch + j] / (float)pool;
    out[(size_t)b * out_bstride + (size_t)row * out_pitch + t] = v;
}
int wn_launch_avgpool_bwd(const float* denc, long denc_bstride, int denc_pitch, int t0, int pool, int n_out, int rows,
                          float* out, long out_bstride, int out_pitch, int t_hi, int batch, hipStream_t st) {
    if (rows <= 0 || batch <= 0 || t_hi <= t0) return 0;
    hipLaunchKernelGGL(avgpool_bwd_k, dim3((t_hi - t0 + 255) / 256, rows, batch), dim3(256), 0, st, denc, denc_bstride,
                       denc_pitch, t0, pool, n_out, out, out_bstride, out_pitch, t_hi);
    WN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The operand format of the x3 products, as a kernel of its own: hi = cvt16(x), lo = cvt16(x - hi) for every element
// (split2 of wn_common.h - the function every MFMA operand of the library goes through - on pairs).  A utility for
// hosts that want to pre-split a tensor, and the handle by which tests/test_gpu_kernels.py checks split2 bit for bit.
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void split16_k(const float* __restrict__ x, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, long n) {
    const long i = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const float a = x[i], b = i + 1 < n ? x[i + 1] : 0.f;
    uint32_t h, l;
    split2<T>(a, b, h, l);
    hi[i] = (uint16_t)h;
    lo[i] = (uint16_t)l;
    if (i + 1 < n) {
        hi[i + 1] = (uint16_t)(h >> 16);
        lo[i + 1] = (uint16_t)(l >> 16);
    }
}
int wn_launch_split16(const float* x, uint16_t* hi, uint16_t* lo, long n, int is_bf16, hipStream_t st) {
    if (n <= 0) return 0;
    const long pairs = (n + 1) / 2;
    dim3 g((unsigned)((pairs + 255) / 256)), b(256);
    if (is_bf16) hipLaunchKernelGGL(split16_k<BF16>, g, b, 0, st, x, hi, lo, n);
    else hipLaunchKernelGGL(split16_k<F16>, g, b, 0, st, x, hi, lo, n);
    WN_CHECK_LAUNCH();
    return 0;
}
