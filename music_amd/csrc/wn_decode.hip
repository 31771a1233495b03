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


// Next code from the 256 pre-softmax logits (one wave, lane l owns entries 4l..4l+3): probabilities =
// softmax(logit * inv_temp) (written to probs_dst if given), then either the first-index argmax (the
// reference's greedy topk(1), fast_generate.py:139) or - SURVEY 8f2 - a draw from that distribution by
// inverse CDF with the uniform number u in [0,1).  The result is valid in every lane.
__device__ __forceinline__ int dec_choose(const float* logit, int lane, float* probs_dst, float inv_temp, bool sample, float u) {
    float v[4], m = -INFINITY;
    for (int e = 0; e < 4; ++e) { v[e] = logit[lane * 4 + e]; m = fmaxf(m, v[e]); }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float s = 0.f;
    for (int e = 0; e < 4; ++e) { v[e] = expf((v[e] - m) * inv_temp); s += v[e]; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float inv = 1.0f / s;
    float best = -1.f; int bi = 0;
    for (int e = 0; e < 4; ++e) {
        v[e] *= inv;
        if (probs_dst) probs_dst[lane * 4 + e] = v[e];
        if (v[e] > best) { best = v[e]; bi = lane * 4 + e; }
    }
    if (!sample) {
        for (int off = 32; off > 0; off >>= 1) {
            float ob = __shfl_xor(best, off, 64);
            int oi = __shfl_xor(bi, off, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        return bi;
    }
    // inclusive scan of the lane sums, then the first entry whose cumulative probability exceeds u
    const float mine = (v[0] + v[1]) + (v[2] + v[3]);
    float incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    float c = incl - mine;
    int pick = 1 << 20;
    for (int e = 0; e < 4; ++e) {
        c += v[e];
        if (pick == (1 << 20) && c > u) pick = lane * 4 + e;
    }
    for (int off = 32; off > 0; off >>= 1) pick = min(pick, __shfl_xor(pick, off, 64));
    return pick < 256 ? pick : 255;                 // (u above the rounded total: last entry)
}

// Uniform number in [0,1) for (seed, global step, utterance): splitmix64 finaliser, 24 random bits.
__device__ __forceinline__ float dec_uniform(unsigned long long seed, unsigned long long step, unsigned long long utt) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (step + 1) + 0xD1B54A32D192ED03ull * (utt + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(DEC_THREADS) void decode_k(WnDecodeArgs a) {
    // utterance of a batched launch: per-utterance pointers as LOCALS (the argument struct itself must stay
    // untouched: a modified copy would be moved to scratch and every dil[] / q_off[] lookup with it)
    const size_t utt = blockIdx.x;
    float* const u_queues = a.queues + utt * a.queues_ustride;
    const float* const u_note0 = a.note0 + utt * a.Q;
    const float* const u_prev0 = a.prev0 + utt * a.Q;
    float* const u_note_out = a.note_out + utt * a.Q;
    float* const u_prev_out = a.prev_out + utt * a.Q;
    const int32_t* const u_forced = a.forced ? a.forced + utt * a.n_steps : nullptr;
    int32_t* const u_codes_out = a.codes_out + utt * a.n_steps;
    float* const u_probs_out = a.probs_out ? a.probs_out + utt * (size_t)a.n_steps * a.Q : nullptr;
    unsigned long long* const u_sync = a.sync ? a.sync + utt * ((size_t)a.n_layers * a.D + 2) : nullptr;
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

    for (int i = tid; i < a.Q; i += DEC_THREADS) { note[i] = u_note0[i]; prev[i] = u_prev0[i]; }
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
            float* q = u_queues + a.q_off[l];
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
            const float ur = a.sample ? dec_uniform(a.seed, (unsigned long long)(a.step0 + step), utt) : 0.f;
            const int bi = dec_choose(logit, tid, u_probs_out ? u_probs_out + (size_t)step * a.Q : nullptr, a.inv_temp, a.sample != 0, ur);
            if (tid == 0) { s_arg = bi; u_codes_out[step] = bi; }
        }
        __syncthreads();
        // next input column: the forced code if given (teacher forcing), else the prediction
        const int nextc = u_forced ? u_forced[step] : s_arg;
        for (int i = tid; i < a.Q; i += DEC_THREADS) { prev[i] = note[i]; }
        __syncthreads();
        for (int i = tid; i < a.Q; i += DEC_THREADS) note[i] = (i == nextc) ? 1.0f : 0.0f;
        __syncthreads();
    }
    // hand the two input columns back (prev = causal queue, note = next input)
    for (int i = tid; i < a.Q; i += DEC_THREADS) { u_prev_out[i] = prev[i]; u_note_out[i] = note[i]; }
}

// ---------------------------------------------------------------------------------------------
// decode_v4_k: same recurrence, built for the real bottleneck of a one-CU sequential kernel, the
// INSTRUCTION count per multiply-add: every thread owns a CONTIGUOUS k-slice of one output row,
// reads its weights as float4 (prefetched one block ahead, they do not depend on data) and its
// slice of the input vector as float4 from LDS, and `parts` adjacent lanes combine with shuffles.
// The residual stream ping-pongs between two LDS buffers, queue columns are written by the lanes
// that own the outputs, ring positions live in LDS (no 64-bit modulo per block).
// Requires (else decode_k): no biases, one pass per product, weights per thread = 16 / 4 / 16 for
// the f/g, dense and skip products and multiples of 4 for causal / post-process.
// ---------------------------------------------------------------------------------------------
struct DecMap { int parts, p, o, nw; };
__device__ __forceinline__ DecMap dec_map(int M, int K) {
    int parts = DEC_THREADS / M;
    if (parts < 1) parts = 1;
    if (parts > 64) parts = 64;
    while (parts & (parts - 1)) parts &= parts - 1;
    DecMap m;
    m.parts = parts; m.p = threadIdx.x % parts; m.o = threadIdx.x / parts; m.nw = K / parts;
    return m;
}
template <int NV>      // NV float4 of weights: W[o][p*nw .. p*nw + 4*NV)
__device__ __forceinline__ void dec_loadw4(f32x4 (&w)[NV], const DecMap& m, const float* __restrict__ W, int ldw, int M) {
    const float* r = W + (size_t)(m.o < M ? m.o : 0) * ldw + m.p * m.nw;
#pragma unroll
    for (int j = 0; j < NV; ++j) w[j] = ld4(r + 4 * j);
}
template <int NV>
__device__ __forceinline__ float dec_dot4(const f32x4 (&w)[NV], const DecMap& m, const float* x) {
    const float* xs = x + m.p * m.nw;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + 4 * j);
        s = fmaf(w[j][0], xv[0], s); s = fmaf(w[j][1], xv[1], s);
        s = fmaf(w[j][2], xv[2], s); s = fmaf(w[j][3], xv[3], s);
    }
    for (int off = m.parts >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}
// long rows (causal, post-process): weights streamed at use, 4 float4 in flight
__device__ __forceinline__ float dec_dot_stream(const DecMap& m, const float* __restrict__ W, int ldw, int M, const float* x) {
    const float* r = W + (size_t)(m.o < M ? m.o : 0) * ldw + m.p * m.nw;
    const float* xs = x + m.p * m.nw;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int k = 0; k < m.nw; k += 16) {
        f32x4 w0 = ld4(r + k), w1 = ld4(r + k + 4), w2 = ld4(r + k + 8), w3 = ld4(r + k + 12);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + k), x1 = *reinterpret_cast<const f32x4*>(xs + k + 4);
        const f32x4 x2 = *reinterpret_cast<const f32x4*>(xs + k + 8), x3 = *reinterpret_cast<const f32x4*>(xs + k + 12);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s0 = fmaf(w0[e], x0[e], s0); s1 = fmaf(w1[e], x1[e], s1);
            s2 = fmaf(w2[e], x2[e], s2); s3 = fmaf(w3[e], x3[e], s3);
        }
    }
    float s = (s0 + s1) + (s2 + s3);
    for (int off = m.parts >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// Workgroup barrier that only drains LDS traffic.  __syncthreads() also waits for every outstanding
// global access (vmcnt(0), because of the queue-column stores), which would expose the L2 latency
// of the weight prefetches at each of the ~90 barriers of a sample.
__device__ __forceinline__ void dec_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(DEC_THREADS) void decode_v4_k(WnDecodeArgs a) {
    // utterance of a batched launch: per-utterance pointers as LOCALS (the argument struct itself must stay
    // untouched: a modified copy would be moved to scratch and every dil[] / q_off[] lookup with it)
    const size_t utt = blockIdx.x;
    float* const u_queues = a.queues + utt * a.queues_ustride;
    const float* const u_note0 = a.note0 + utt * a.Q;
    const float* const u_prev0 = a.prev0 + utt * a.Q;
    float* const u_note_out = a.note_out + utt * a.Q;
    float* const u_prev_out = a.prev_out + utt * a.Q;
    const int32_t* const u_forced = a.forced ? a.forced + utt * a.n_steps : nullptr;
    int32_t* const u_codes_out = a.codes_out + utt * a.n_steps;
    float* const u_probs_out = a.probs_out ? a.probs_out + utt * (size_t)a.n_steps * a.Q : nullptr;
    unsigned long long* const u_sync = a.sync ? a.sync + utt * ((size_t)a.n_layers * a.D + 2) : nullptr;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* prev = sm;                       // [Q]
    float* note = prev + a.Q;               // [Q]   ([prev|note] contiguous)
    float* cur0 = note + a.Q;               // [2R]  cur (R) followed by old (R), buffer 0
    float* cur1 = cur0 + 2 * a.R;           // [2R]  buffer 1
    float* fg = cur1 + 2 * a.R;             // [2D]
    float* zz = fg + 2 * a.D;               // [D]
    float* skip = zz + a.D;                 // [S]
    float* h1 = skip + a.S;                 // [S]
    float* logit = h1 + a.S;                // [Q]
    __shared__ int s_arg;
    __shared__ int slots[WN_DEC_MAX_LAYERS];          // ring position of every block
    const int tid = threadIdx.x;
    const int R = a.R, D = a.D, S = a.S, Q = a.Q;
    const DecMap mc = dec_map(R, 2 * Q), mfg = dec_map(2 * D, 2 * R), md = dec_map(R, D), ms = dec_map(S, D);
    const DecMap mp1 = dec_map(S, S), mp2 = dec_map(Q, S);
    const size_t lstride = (size_t)a.layer_stride;
    const size_t o_d = (size_t)2 * D * 2 * R, o_s = o_d + (size_t)R * D;

    for (int i = tid; i < Q; i += DEC_THREADS) { note[i] = u_note0[i]; prev[i] = u_prev0[i]; }
    if (tid < a.n_layers) slots[tid] = (int)(a.step0 % a.dil[tid]);
    dec_sync();

    f32x4 wfg[4], wd[1], ws[4];
    for (int step = 0; step < a.n_steps; ++step) {
        // block-0 weights and queue column are fetched while the causal layer runs
        dec_loadw4(wfg, mfg, a.w_layers, 2 * R, 2 * D);
        dec_loadw4(wd, md, a.w_layers + o_d, D, R);
        dec_loadw4(ws, ms, a.w_layers + o_s, D, S);
        float oldv = 0.f;
        if (tid < R) oldv = u_queues[a.q_off[0] + (size_t)slots[0] * R + tid];
        if (!(a.dbg & 8)) {
            const float s = dec_dot_stream(mc, a.w_causal, 2 * Q, R, prev);
            if (mc.o < R && mc.p == 0) cur0[mc.o] = s;
        }
        for (int i = tid; i < S; i += DEC_THREADS) skip[i] = 0.f;
        if (tid < R) cur0[R + tid] = oldv;
        dec_sync();
        float* cur = cur0;
        float* nxt = cur1;
        for (int l = 0; l < ((a.dbg & 16) ? 1 : a.n_layers); ++l) {
            // ---- [f;g] = Wfg [cur | old]
            const float s = dec_dot4(wfg, mfg, cur);
            if (mfg.o < 2 * D && mfg.p == 0) fg[mfg.o] = s;
            const int ln = l + 1;
            const float* wn = a.w_layers + (size_t)ln * lstride;
            float oldn = 0.f;
            if (ln < a.n_layers) {
                dec_loadw4(wfg, mfg, wn, 2 * R, 2 * D);
                if (tid < R && !(a.dbg & 2)) oldn = u_queues[a.q_off[ln] + (size_t)slots[ln] * R + tid];
            }
            dec_sync();
            if (tid < D) zz[tid] = (a.dbg & 1) ? wn_tanh(fg[tid]) * wn_sigmoid(fg[D + tid])
                                               : tanhf(fg[tid]) * (1.0f / (1.0f + expf(-fg[D + tid])));
            dec_sync();
            // ---- dense (+ residual) and skip products
            const float sd = dec_dot4(wd, md, zz);
            const float ss = dec_dot4(ws, ms, zz);
            if (md.o < R && md.p == 0) {
                const float v = sd + cur[md.o];
                nxt[md.o] = v;
                if (!(a.dbg & 2)) u_queues[a.q_off[l] + (size_t)slots[l] * R + md.o] = a.push_input ? cur[md.o] : v;   // Q5: output by default
            }
            if (ms.o < S && ms.p == 0) skip[ms.o] += ss;
            if (ln < a.n_layers) {
                dec_loadw4(wd, md, wn + o_d, D, R);
                dec_loadw4(ws, ms, wn + o_s, D, S);
                if (tid < R) nxt[R + tid] = oldn;
            }
            dec_sync();
            float* t = cur; cur = nxt; nxt = t;
        }
        // ---- post-processing: relu -> P1 -> relu -> P2 -> softmax -> argmax
        for (int i = tid; i < S; i += DEC_THREADS) skip[i] = fmaxf(skip[i], 0.f);
        dec_sync();
        if (!(a.dbg & 4)) {
            const float s = dec_dot_stream(mp1, a.w_p1, S, S, skip);
            if (mp1.o < S && mp1.p == 0) h1[mp1.o] = fmaxf(s, 0.f);
        }
        dec_sync();
        if (!(a.dbg & 4)) {
            const float s = dec_dot_stream(mp2, a.w_p2, S, Q, h1);
            if (mp2.o < Q && mp2.p == 0) logit[mp2.o] = s;
        }
        dec_sync();
        if (tid < 64) {
            const float ur = a.sample ? dec_uniform(a.seed, (unsigned long long)(a.step0 + step), utt) : 0.f;
            const int bi = dec_choose(logit, tid, u_probs_out ? u_probs_out + (size_t)step * a.Q : nullptr, a.inv_temp, a.sample != 0, ur);
            if (tid == 0) { s_arg = bi; u_codes_out[step] = bi; }
        }
        dec_sync();
        const int nextc = u_forced ? u_forced[step] : s_arg;
        for (int i = tid; i < Q; i += DEC_THREADS) prev[i] = note[i];
        if (tid < a.n_layers) { int sl = slots[tid] + 1; slots[tid] = sl == a.dil[tid] ? 0 : sl; }
        dec_sync();
        for (int i = tid; i < Q; i += DEC_THREADS) note[i] = (i == nextc) ? 1.0f : 0.0f;
        dec_sync();
    }
    for (int i = tid; i < Q; i += DEC_THREADS) { u_prev_out[i] = prev[i]; u_note_out[i] = note[i]; }
}

// ---------------------------------------------------------------------------------------------
// decode_duo_k: the recurrence split over TWO workgroups on (normally) two XCDs, because one CU can
// only stream ~35 GB/s from beyond its XCD's 4 MB L2 and the 5 MB of fp32 weights do not fit it:
//   block 0 "chain": causal layer, per block f/g product, gate, dense product, queue update
//                    (2.6 MB of weights -> resident in ITS L2);
//   block 1 "skip" : per block the skip product Ws z_l (1.97 MB), then relu/P1/relu/P2/softmax/argmax
//                    (0.5 MB) -> resident in ITS L2.
// z_l travels as 8-byte {value, tag} granules written with one agent-scope relaxed 64-bit store
// and polled with agent-scope relaxed loads (tag = sample number; no fences, no flags: the
// placement-independent hand-off of MI355X_MICROARCH.md "R2 granule"); the predicted code travels
// back the same way.  Every spin is bounded; a timeout sets sync[err] and both blocks run out.
// Measured (config 5): 8.1-8.4 k samples/s vs 5.1 k for one workgroup.  What bounds the chain now is
// the ~64 KB of f/g weights a block needs per sample through ONE CU's memory pipe (~1.7 us per
// block even from L2; bisected with the WN_DEC_DBG switches: without the f/g product the chain
// runs 17.7 k samples/s); the next step is a block-pipelined chain over ~15 CUs with the weights
// resident in LDS.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long dec_pack(float v, unsigned tag) {
    return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}
__device__ __forceinline__ bool dec_poll(const unsigned long long* p, unsigned tag, float& v, unsigned long long* err) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(g >> 32) == tag) { v = __uint_as_float((unsigned)g); return true; }
        if ((spin & 255) == 255 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = 0.f;
    return false;
}

__global__ __launch_bounds__(DEC_THREADS) void decode_duo_k(WnDecodeArgs a) {
    // workgroups 2u (chain) and 2u+1 (skip + post) serve utterance u of a batched launch: per-utterance pointers as LOCALS (the argument struct itself must stay
    // untouched: a modified copy would be moved to scratch and every dil[] / q_off[] lookup with it)
    const size_t utt = blockIdx.x >> 1;
    float* const u_queues = a.queues + utt * a.queues_ustride;
    const float* const u_note0 = a.note0 + utt * a.Q;
    const float* const u_prev0 = a.prev0 + utt * a.Q;
    float* const u_note_out = a.note_out + utt * a.Q;
    float* const u_prev_out = a.prev_out + utt * a.Q;
    const int32_t* const u_forced = a.forced ? a.forced + utt * a.n_steps : nullptr;
    int32_t* const u_codes_out = a.codes_out + utt * a.n_steps;
    float* const u_probs_out = a.probs_out ? a.probs_out + utt * (size_t)a.n_steps * a.Q : nullptr;
    unsigned long long* const u_sync = a.sync ? a.sync + utt * ((size_t)a.n_layers * a.D + 2) : nullptr;
    const int role = blockIdx.x & 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int s_arg;
    __shared__ int slots[WN_DEC_MAX_LAYERS];
    const int tid = threadIdx.x;
    const int R = a.R, D = a.D, S = a.S, Q = a.Q;
    unsigned long long* zg = u_sync;                         // [n_layers][D] z granules
    unsigned long long* cg = u_sync + (size_t)a.n_layers * D;    // code granule
    unsigned long long* err = cg + 1;
    const size_t lstride = (size_t)a.layer_stride;
    const size_t o_d = (size_t)2 * D * 2 * R, o_s = o_d + (size_t)R * D;

    if (role == 0) {
        // ------------------------------------------------------------------ chain
        float* prev = sm;
        float* note = prev + Q;
        float* cur0 = note + Q;
        float* cur1 = cur0 + 2 * R;
        float* fg = cur1 + 2 * R;
        float* zz = fg + 2 * D;
        float* oldb = zz + D;                    // [n_layers][R] oldest queue columns of this sample
        float* pushb = oldb + a.n_layers * R;    // [n_layers][R] columns pushed by this sample
        const DecMap mc = dec_map(R, 2 * Q), mfg = dec_map(2 * D, 2 * R), md = dec_map(R, D);
        for (int i = tid; i < Q; i += DEC_THREADS) { note[i] = u_note0[i]; prev[i] = u_prev0[i]; }
        if (tid < a.n_layers) slots[tid] = (int)(a.step0 % a.dil[tid]);
        dec_sync();
        f32x4 wfg[4], wd[1];
        for (int step = 0; step < a.n_steps; ++step) {
            const unsigned tag = (unsigned)step + 1u;
            // all queue traffic of a sample happens here (oldest columns in, L1-bypassing loads) and after
            // the last block (pushed columns out): global stores inside the block loop would sit in the
            // in-order vmcnt queue (~2 us each) in front of every wait for prefetched weights
            for (int i = tid; i < a.n_layers * R; i += DEC_THREADS) {
                const int l = i / R, r = i - l * R;
                oldb[i] = __hip_atomic_load(u_queues + a.q_off[l] + (size_t)slots[l] * R + r, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
            }
            dec_loadw4(wfg, mfg, a.w_layers, 2 * R, 2 * D);
            dec_loadw4(wd, md, a.w_layers + o_d, D, R);
            {
                const float s = dec_dot_stream(mc, a.w_causal, 2 * Q, R, prev);
                if (mc.o < R && mc.p == 0) cur0[mc.o] = s;
            }
            dec_sync();
            if (tid < R) cur0[R + tid] = oldb[tid];
            dec_sync();
            float* cur = cur0;
            float* nxt = cur1;
            for (int l = 0; l < a.n_layers; ++l) {
                const float s = (a.dbg & 1024) ? cur[tid & 63] : dec_dot4(wfg, mfg, cur);
                if (mfg.o < 2 * D && mfg.p == 0) fg[mfg.o] = s;
                const int ln = l + 1;
                const float* wn = a.w_layers + (size_t)ln * lstride;
                if (ln < a.n_layers && !(a.dbg & 256)) dec_loadw4(wfg, mfg, wn, 2 * R, 2 * D);
                dec_sync();
                if (tid < D) {
                    const float z = (a.dbg & 64) ? wn_tanh(fg[tid]) * wn_sigmoid(fg[D + tid])
                                                 : tanhf(fg[tid]) * (1.0f / (1.0f + expf(-fg[D + tid])));
                    zz[tid] = z;
                    if (!(a.dbg & 128))
                        __hip_atomic_store(zg + (size_t)l * D + tid, dec_pack(z, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!(a.dbg & 2048)) dec_sync();
                const float sd = (a.dbg & 4096) ? zz[tid & 63] : dec_dot4(wd, md, zz);
                if (md.o < R && md.p == 0) {
                    const float v = sd + cur[md.o];
                    nxt[md.o] = v;
                    pushb[l * R + md.o] = a.push_input ? cur[md.o] : v;          // Q5: output by default
                }
                if (ln < a.n_layers) {
                    if (!(a.dbg & 256)) dec_loadw4(wd, md, wn + o_d, D, R);
                    if (tid < R) nxt[R + tid] = oldb[ln * R + tid];
                }
                dec_sync();
                float* t = cur; cur = nxt; nxt = t;
            }
            for (int i = tid; i < a.n_layers * R; i += DEC_THREADS) {            // queue columns out
                const int l = i / R, r = i - l * R;
                u_queues[a.q_off[l] + (size_t)slots[l] * R + r] = pushb[i];
            }
            // the prediction comes back from the skip block
            if (tid == 0) {
                float cv = 0.f;
                if (!(a.dbg & 32)) dec_poll(cg, tag, cv, err);
                s_arg = (int)cv;
            }
            __syncthreads();                       // full fence: the queue stores are complete before the next sample reads
            const int nextc = u_forced ? u_forced[step] : s_arg;
            for (int i = tid; i < Q; i += DEC_THREADS) prev[i] = note[i];
            if (tid < a.n_layers) { int sl = slots[tid] + 1; slots[tid] = sl == a.dil[tid] ? 0 : sl; }
            dec_sync();
            for (int i = tid; i < Q; i += DEC_THREADS) note[i] = (i == nextc) ? 1.0f : 0.0f;
            dec_sync();
        }
        for (int i = tid; i < Q; i += DEC_THREADS) { u_prev_out[i] = prev[i]; u_note_out[i] = note[i]; }
    } else {
        // ------------------------------------------------------------------ skip + post-processing
        float* zz0 = sm;                       // [2][D]
        float* skip = zz0 + 2 * D;             // [S]
        float* h1 = skip + S;                  // [S]
        float* logit = h1 + S;                 // [Q]
        const DecMap ms = dec_map(S, D), mp1 = dec_map(S, S), mp2 = dec_map(Q, S);
        f32x4 ws[4];
        for (int step = 0; step < ((a.dbg & 32) ? 0 : a.n_steps); ++step) {
            const unsigned tag = (unsigned)step + 1u;
            float part = 0.f;                                    // this thread's slice of its skip row, all blocks
            dec_loadw4(ws, ms, a.w_layers + o_s, D, S);
            for (int l = 0; l < a.n_layers; ++l) {
                float* zz = zz0 + (l & 1) * D;
                if (tid < D) {
                    float z;
                    dec_poll(zg + (size_t)l * D + tid, tag, z, err);
                    zz[tid] = z;
                }
                dec_sync();
                f32x4 wsn[4];
                if (l + 1 < a.n_layers) dec_loadw4(wsn, ms, a.w_layers + (size_t)(l + 1) * lstride + o_s, D, S);
                const float* xs = zz + ms.p * ms.nw;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + 4 * j);
                    part = fmaf(ws[j][0], xv[0], part); part = fmaf(ws[j][1], xv[1], part);
                    part = fmaf(ws[j][2], xv[2], part); part = fmaf(ws[j][3], xv[3], part);
                }
                if (l + 1 < a.n_layers) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ws[j] = wsn[j];
                }
            }
            for (int off = ms.parts >> 1; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (ms.o < S && ms.p == 0) skip[ms.o] = fmaxf(part, 0.f);
            dec_sync();
            {
                const float s = dec_dot_stream(mp1, a.w_p1, S, S, skip);
                if (mp1.o < S && mp1.p == 0) h1[mp1.o] = fmaxf(s, 0.f);
            }
            dec_sync();
            {
                const float s = dec_dot_stream(mp2, a.w_p2, S, Q, h1);
                if (mp2.o < Q && mp2.p == 0) logit[mp2.o] = s;
            }
            dec_sync();
            if (tid < 64) {
                const float ur = a.sample ? dec_uniform(a.seed, (unsigned long long)(a.step0 + step), utt) : 0.f;
                const int bi = dec_choose(logit, tid, u_probs_out ? u_probs_out + (size_t)step * Q : nullptr, a.inv_temp, a.sample != 0, ur);
                if (tid == 0) {
                    u_codes_out[step] = bi;
                    __hip_atomic_store(cg, dec_pack((float)bi, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            dec_sync();
        }
    }
}

int wn_launch_decode(const WnDecodeArgs& a, hipStream_t st) {
    if (a.n_steps <= 0) return 0;
    if (a.n_layers > WN_DEC_MAX_LAYERS) return wn_set_error_msg(-4, "decode: too many layers");
    const int nu = a.n_utt > 0 ? a.n_utt : 1;
    // the two workgroups of an utterance spin on each other's hand-offs, so every pair must be resident
    // at once: one workgroup per CU -> at most 128 utterances per launch on this part
    if (nu > 128) return wn_set_error_msg(-4, "decode: at most 128 utterances per launch");
    // the float4 kernel needs: no biases, one pass per product, exactly 16 / 4 / 16 weights per thread for
    // the per-block products and a multiple of 16 for the streamed ones
    auto nw = [](int M, int K) {
        int parts = DEC_THREADS / M; if (parts < 1) parts = 1; if (parts > 64) parts = 64;
        while (parts & (parts - 1)) parts &= parts - 1;
        return (K % parts) ? -1 : K / parts;
    };
    const bool v4 = !a.b_layers && !a.b_causal && !a.b_p1 && !a.b_p2 &&
                    2 * a.D <= DEC_THREADS && a.S <= DEC_THREADS && a.Q <= DEC_THREADS && a.R <= DEC_THREADS &&
                    nw(2 * a.D, 2 * a.R) == 16 && nw(a.R, a.D) == 4 && nw(a.S, a.D) == 16 &&
                    nw(a.R, 2 * a.Q) > 0 && nw(a.R, 2 * a.Q) % 16 == 0 && nw(a.S, a.S) > 0 && nw(a.S, a.S) % 16 == 0 &&
                    nw(a.Q, a.S) > 0 && nw(a.Q, a.S) % 16 == 0 && (a.layer_stride % 4) == 0;
    if (v4 && a.sync && a.n_steps >= 4 && !(a.dbg & 31)) {
        const size_t nsync = ((size_t)a.n_layers * a.D + 2) * sizeof(unsigned long long) * (size_t)nu;
        hipError_t e = hipMemsetAsync(a.sync, 0, nsync, st);              // tags start at 1
        if (e != hipSuccess) return wn_set_error(e, __FILE__, __LINE__);
        size_t sh0 = sizeof(float) * (size_t)(2 * a.Q + 4 * a.R + 3 * a.D + 2 * a.n_layers * a.R);
        size_t sh1 = sizeof(float) * (size_t)(2 * a.D + 2 * a.S + a.Q);
        hipLaunchKernelGGL(decode_duo_k, dim3(2 * nu), dim3(DEC_THREADS), sh0 > sh1 ? sh0 : sh1, st, a);
    } else if (v4) {
        size_t sh = sizeof(float) * (size_t)(3 * a.Q + 4 * a.R + 3 * a.D + 2 * a.S);
        hipLaunchKernelGGL(decode_v4_k, dim3(nu), dim3(DEC_THREADS), sh, st, a);
    } else {
        size_t sh = sizeof(float) * (size_t)(3 * a.Q + 3 * a.R + 3 * a.D + 2 * a.S + 64);
        hipLaunchKernelGGL(decode_k, dim3(nu), dim3(DEC_THREADS), sh, st, a);
    }
    WN_CHECK_LAUNCH();
    return 0;
}
