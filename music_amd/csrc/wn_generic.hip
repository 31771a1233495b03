// Kernels of the GENERAL path (music_amd/engine_generic.py): constructor arguments the specialised kernels do not
// cover - filter_width != 2, quantization_channels != 256, more than 64 residual / dilation channels
// (wavenet/model.py:8-15 takes any).  The channel-mixing products of that path are wn_chan_gemm / wn_wgrad launches (any
// row count, any K, two taps per launch, accumulation through `resid`); what is left is elementwise:
//   * the gate z = tanh f * sigmoid g and its derivative (wavenet/model.py:120),
//   * the CHUNK softmax over rows of Q consecutive floats for any Q (model.py:142-144), its backward, and the
//     fused softmax + CrossEntropyLoss-on-probabilities step (wavenet/train.py:146,179).
// One wave per softmax row (lane j takes elements j, j + 64, ...); no float atomics anywhere.
#include "wn_common.h"
#include "wn_kernels.h"

__device__ __forceinline__ float gq_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float gq_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// fg: [B][2*dp][pitch] (f rows [0, dp), g rows [dp, 2dp)); z: [B][...][pitch] rows [0, n_d)
__global__ __launch_bounds__(256) void gate_fwd_k(const float* __restrict__ fg, long fg_bstride, int dp, float* __restrict__ z,
                                                  long z_bstride, int pitch, int t_lo, int t_hi) {
    const int t = t_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= t_hi) return;
    const int r = blockIdx.y, b = blockIdx.z;
    const float* p = fg + (size_t)b * fg_bstride + (size_t)r * pitch + t;
    const float f = p[0], g = p[(size_t)dp * pitch];
    z[(size_t)b * z_bstride + (size_t)r * pitch + t] = wn_tanh(f) * wn_sigmoid(g);
}
// dfg[f row] = dz sigma(g) (1 - tanh^2 f) ; dfg[g row] = dz tanh(f) sigma(g) (1 - sigma(g))
__global__ __launch_bounds__(256) void gate_bwd_k(const float* __restrict__ fg, long fg_bstride, int dp, const float* __restrict__ dz,
                                                  long dz_bstride, float* __restrict__ dfg, long dfg_bstride, int pitch, int t_lo,
                                                  int t_hi) {
    const int t = t_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= t_hi) return;
    const int r = blockIdx.y, b = blockIdx.z;
    const float* p = fg + (size_t)b * fg_bstride + (size_t)r * pitch + t;
    const WnGateD gd = wn_gate_d(p[0], p[(size_t)dp * pitch]);
    const float g = dz[(size_t)b * dz_bstride + (size_t)r * pitch + t];
    float* o = dfg + (size_t)b * dfg_bstride + (size_t)r * pitch + t;
    o[0] = g * gd.dzdf;
    o[(size_t)dp * pitch] = g * gd.dzdg;
}
int wn_launch_gate_fwd(const float* fg, long fg_bstride, int dp, int rows, float* z, long z_bstride, int pitch, int t_lo, int t_hi,
                       int batch, hipStream_t st) {
    if (t_hi <= t_lo || rows <= 0 || batch <= 0) return 0;
    hipLaunchKernelGGL(gate_fwd_k, dim3((t_hi - t_lo + 255) / 256, rows, batch), dim3(256), 0, st, fg, fg_bstride, dp, z,
                       z_bstride, pitch, t_lo, t_hi);
    WN_CHECK_LAUNCH();
    return 0;
}
int wn_launch_gate_bwd(const float* fg, long fg_bstride, int dp, int rows, const float* dz, long dz_bstride, float* dfg,
                       long dfg_bstride, int pitch, int t_lo, int t_hi, int batch, hipStream_t st) {
    if (t_hi <= t_lo || rows <= 0 || batch <= 0) return 0;
    hipLaunchKernelGGL(gate_bwd_k, dim3((t_hi - t_lo + 255) / 256, rows, batch), dim3(256), 0, st, fg, fg_bstride, dp, dz,
                       dz_bstride, dfg, dfg_bstride, pitch, t_lo, t_hi);
    WN_CHECK_LAUNCH();
    return 0;
}

// ---- chunk softmax over rows of q consecutive floats, any q >= 1
__global__ __launch_bounds__(256) void softmaxq_fwd_k(const float* __restrict__ x, float* __restrict__ y, long nrows, int q) {
    const int lane = threadIdx.x & 63;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += (long)gridDim.x * 4) {
        const float* xr = x + row * q;
        float m = -INFINITY;
        for (int j = lane; j < q; j += 64) m = fmaxf(m, xr[j]);
        m = gq_wave_max(m);
        float s = 0.f;
        for (int j = lane; j < q; j += 64) s += expf(xr[j] - m);
        const float inv = 1.0f / gq_wave_sum(s);
        for (int j = lane; j < q; j += 64) y[row * q + j] = expf(xr[j] - m) * inv;
    }
}
__global__ __launch_bounds__(256) void softmaxq_bwd_k(const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ dx, long nrows, int q) {
    const int lane = threadIdx.x & 63;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += (long)gridDim.x * 4) {
        float dot = 0.f;
        for (int j = lane; j < q; j += 64) dot += y[row * q + j] * dy[row * q + j];
        dot = gq_wave_sum(dot);
        for (int j = lane; j < q; j += 64) dx[row * q + j] = y[row * q + j] * (dy[row * q + j] - dot);
    }
}
// as softmax256_ce_k (wn_elem.hip) for any q: loss_part[block] = this block's share of mean_r [logsumexp(p_r) - p_r[y_r]];
// dx = p (dp - <dp, p>), dp = (softmax(p) - e_y) inv_n.  A target outside [0, q) gives NaN (loss and that row's gradient).
__global__ __launch_bounds__(256) void softmaxq_ce_k(const float* __restrict__ x, const int64_t* __restrict__ target,
                                                     float* __restrict__ probs, float* __restrict__ dx,
                                                     float* __restrict__ loss_part, long nrows, int q, float inv_n) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63;
    float lacc = 0.f;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += (long)gridDim.x * 4) {
        const float* xr = x + row * q;
        const long y = target[row];
        const bool bad = (unsigned long)y >= (unsigned long)q;
        float m = -INFINITY;
        for (int j = lane; j < q; j += 64) m = fmaxf(m, xr[j]);
        m = gq_wave_max(m);
        float s = 0.f;
        for (int j = lane; j < q; j += 64) s += expf(xr[j] - m);
        const float inv = 1.0f / gq_wave_sum(s);
        float s2 = 0.f, py = 0.f;
        for (int j = lane; j < q; j += 64) {
            const float p = expf(xr[j] - m) * inv;
            if (probs) probs[row * q + j] = p;
            s2 += expf(p);                              // second softmax over probabilities in [0, 1]: no shift needed
            if (j == y) py = p;
        }
        s2 = gq_wave_sum(s2);
        py = gq_wave_sum(py);
        lacc += bad ? __builtin_nanf("") : logf(s2) - py;
        if (dx) {
            const float is2 = inv_n / s2;
            float dot = 0.f;
            for (int j = lane; j < q; j += 64) {
                const float p = expf(xr[j] - m) * inv;
                dot += p * (expf(p) * is2 - (j == y ? inv_n : 0.f));
            }
            dot = gq_wave_sum(dot);
            for (int j = lane; j < q; j += 64) {
                const float p = expf(xr[j] - m) * inv;
                const float r = p * (expf(p) * is2 - (j == y ? inv_n : 0.f) - dot);
                dx[row * q + j] = bad ? __builtin_nanf("") : r;
            }
        }
    }
    if (lane == 0) red[threadIdx.x >> 6] = lacc * inv_n;
    __syncthreads();
    if (threadIdx.x == 0 && loss_part) loss_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
static inline int gq_grid(long nrows) {
    long g = (nrows + 3) / 4;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
int wn_launch_softmaxq_fwd(const float* x, float* y, long nrows, int q, hipStream_t st) {
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(softmaxq_fwd_k, dim3(gq_grid(nrows)), dim3(256), 0, st, x, y, nrows, q);
    WN_CHECK_LAUNCH();
    return 0;
}
int wn_launch_softmaxq_bwd(const float* y, const float* dy, float* dx, long nrows, int q, hipStream_t st) {
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(softmaxq_bwd_k, dim3(gq_grid(nrows)), dim3(256), 0, st, y, dy, dx, nrows, q);
    WN_CHECK_LAUNCH();
    return 0;
}
int wn_launch_softmaxq_ce(const float* x, const int64_t* target, float* probs, float* dx, float* loss_part, long nrows, int q,
                          float inv_n, hipStream_t st) {
    if (nrows <= 0) return 0;
    // always WN_CE_PARTIALS blocks so that every partial is (re)written each call
    hipLaunchKernelGGL(softmaxq_ce_k, dim3(WN_CE_PARTIALS), dim3(256), 0, st, x, target, probs, dx, loss_part, nrows, q, inv_n);
    WN_CHECK_LAUNCH();
    return 0;
}
