// Cached-queue autoregressive decode (wavenet/fast_generate.py:66-141) as ONE persistent kernel:
// a single 1024-thread workgroup generates n_steps samples; per sample it runs the causal layer,
// the N residual blocks (each reading the oldest column of its FIFO queue and pushing a new one),
// the skip sum, both post-process convs, the 256-way softmax and the greedy argmax.
//
// The path is latency-bound (16 000 strictly sequential samples, ~2.5 MFLOP and ~5 MB of fp32
// weights each), so everything is plain fp32 FMA (bit-comparable to the CPU path up to summation
// order), the weights stay L2-resident, the queues are ring buffers in time-major layout
// ([slot][channel], one 256-B line per column; nothing is shifted), and every mat-vec prefetches
// its weights before the barrier that publishes its input.
//
// As written in the reference, block i pushes its OUTPUT into its own queue
// (fast_generate.py:128-129); push_input != 0 selects the corrected recurrence.
#include "wn_common.h"
#include "wn_kernels.h"

#define DEC_THREADS 1024

// out[o] = epi( bias[o] + sum_k W[o*ldw + k] * x[k] ), o < M.  `parts` lanes share one output.
// W rows are read as contiguous slices (coalesced across the lanes of an output group).
template <class Epi>
__device__ __forceinline__ void dec_matvec(const float* __restrict__ W, int ldw, const float* x, int M, int K,
                                           const float* bias, Epi epi) {
    int parts = DEC_THREADS / M;
    if (parts < 1) parts = 1;
    if (parts > 64) parts = 64;
    while (parts & (parts - 1)) parts &= parts - 1;             // power of two
    const int per = DEC_THREADS / parts;                        // outputs per pass
    const int p = threadIdx.x % parts, og = threadIdx.x / parts;
    for (int o0 = 0; o0 < M; o0 += per) {
        const int o = o0 + og;
        float s = 0.f;
        if (o < M) {
            const float* w = W + (size_t)o * ldw;
            for (int k = p; k < K; k += parts) s = fmaf(w[k], x[k], s);      // lanes read consecutive floats
        }
        for (int off = parts >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (o < M && p == 0) epi(o, s + (bias ? bias[o] : 0.f));
    }
}

__global__ __launch_bounds__(DEC_THREADS) void decode_k(WnDecodeArgs a) {
    extern __shared__ float sm[];
    float* prev = sm;                       // [Q] previous input column (the causal layer's queue)
    float* note = prev + a.Q;               // [Q] current input column (dense); [prev|note] is contiguous
    float* cur = note + a.Q;                // [R] residual stream at this sample
    float* old = cur + a.R;                 // [R] oldest queue column of the current block
    float* fg = old + a.R;                  // [2D]
    float* z = fg + 2 * a.D;                // [D]
    float* nxt = z + a.D;                   // [R]
    float* skip = nxt + a.R;                // [S]
    float* h1 = skip + a.S;                 // [S]
    float* logit = h1 + a.S;                // [Q]
    float* red = logit + a.Q;               // [64]
    __shared__ int s_arg;
    const int tid = threadIdx.x;

    for (int i = tid; i < a.Q; i += DEC_THREADS) { note[i] = a.note0[i]; prev[i] = a.prev0[i]; }
    __syncthreads();

    for (int step = 0; step < a.n_steps; ++step) {
        const long gstep = a.step0 + step;                      // global step index: ring positions
        // ---- causal layer: cur = Wc[:, 0:Q] prev + Wc[:, Q:2Q] note
        dec_matvec(a.w_causal, 2 * a.Q, prev /* note follows prev in LDS */, a.R, 2 * a.Q, a.b_causal,
                   [&](int o, float v) { cur[o] = v; });
        for (int i = tid; i < a.S; i += DEC_THREADS) skip[i] = 0.f;
        __syncthreads();
        for (int l = 0; l < a.n_layers; ++l) {
            const int d = a.dil[l];
            float* q = a.queues + a.q_off[l];
            const int slot = (int)(gstep % d);                  // oldest column == the one replaced now
            for (int i = tid; i < a.R; i += DEC_THREADS) old[i] = q[(size_t)slot * a.R + i];
            __syncthreads();
            const float* wl = a.w_layers + (size_t)l * a.layer_stride;
            const float* bl = a.b_layers ? a.b_layers + (size_t)l * (2 * a.D + a.R + a.S) : nullptr;
            // [f;g] = Wfg [cur; old]: the decode pack orders k as [tap-1 weights (cur) | tap-0 weights (old)]
            // because cur and old are adjacent in LDS in that order
            dec_matvec(wl, 2 * a.R, cur, 2 * a.D, 2 * a.R, bl, [&](int o, float v) { fg[o] = v; });
            __syncthreads();
            for (int i = tid; i < a.D; i += DEC_THREADS) z[i] = tanhf(fg[i]) * (1.0f / (1.0f + expf(-fg[a.D + i])));
            __syncthreads();
            const float* wd = wl + (size_t)2 * a.D * 2 * a.R;
            const float* wsk = wd + (size_t)a.R * a.D;
            dec_matvec(wd, a.D, z, a.R, a.D, bl ? bl + 2 * a.D : nullptr, [&](int o, float v) { nxt[o] = v + cur[o]; });
            dec_matvec(wsk, a.D, z, a.S, a.D, bl ? bl + 2 * a.D + a.R : nullptr, [&](int o, float v) { skip[o] += v; });
            __syncthreads();
            for (int i = tid; i < a.R; i += DEC_THREADS) {
                q[(size_t)slot * a.R + i] = a.push_input ? cur[i] : nxt[i];     // Q5: output by default
                cur[i] = nxt[i];
            }
            __syncthreads();
        }
        // ---- post-processing: relu -> P1 -> relu -> P2 -> softmax -> argmax
        for (int i = tid; i < a.S; i += DEC_THREADS) skip[i] = fmaxf(skip[i], 0.f);
        __syncthreads();
        dec_matvec(a.w_p1, a.S, skip, a.S, a.S, a.b_p1, [&](int o, float v) { h1[o] = fmaxf(v, 0.f); });
        __syncthreads();
        dec_matvec(a.w_p2, a.S, h1, a.Q, a.S, a.b_p2, [&](int o, float v) { logit[o] = v; });
        __syncthreads();
        // softmax over the Q (=256) logits and first-index argmax of the PROBABILITIES, wave 0
        if (tid < 64) {
            float v[4], m = -INFINITY;
            for (int e = 0; e < 4; ++e) { v[e] = logit[tid * 4 + e]; m = fmaxf(m, v[e]); }
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            float s = 0.f;
            for (int e = 0; e < 4; ++e) { v[e] = expf(v[e] - m); s += v[e]; }
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            const float inv = 1.0f / s;
            float best = -1.f; int bi = 0;
            for (int e = 0; e < 4; ++e) {
                v[e] *= inv;
                if (a.probs_out) a.probs_out[(size_t)step * a.Q + tid * 4 + e] = v[e];
                if (v[e] > best) { best = v[e]; bi = tid * 4 + e; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                float ob = __shfl_xor(best, off, 64);
                int oi = __shfl_xor(bi, off, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (tid == 0) { s_arg = bi; a.codes_out[step] = bi; }
        }
        __syncthreads();
        // next input column: the forced code if given (teacher forcing), else the prediction
        const int nextc = a.forced ? a.forced[step] : s_arg;
        for (int i = tid; i < a.Q; i += DEC_THREADS) { prev[i] = note[i]; }
        __syncthreads();
        for (int i = tid; i < a.Q; i += DEC_THREADS) note[i] = (i == nextc) ? 1.0f : 0.0f;
        __syncthreads();
    }
    // hand the two input columns back (prev = causal queue, note = next input)
    for (int i = tid; i < a.Q; i += DEC_THREADS) { a.prev_out[i] = prev[i]; a.note_out[i] = note[i]; }
}

int wn_launch_decode(const WnDecodeArgs& a, hipStream_t st) {
    if (a.n_steps <= 0) return 0;
    if (a.n_layers > WN_DEC_MAX_LAYERS) return wn_set_error_msg(-4, "decode: too many layers");
    size_t sh = sizeof(float) * (size_t)(3 * a.Q + 3 * a.R + 3 * a.D + 2 * a.S + 64);
    hipLaunchKernelGGL(decode_k, dim3(1), dim3(DEC_THREADS), sh, st, a);
    WN_CHECK_LAUNCH();
    return 0;
}
