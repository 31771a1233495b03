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
#include <type_traits>
#include "wn_common.h"
#include "wn_kernels.h"

#define DEC_THREADS 1024
#define DCLK(v)
#define DACC(i, d)
#define DINIT
#define DFLUSH

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
    unsigned long long* const u_sync = a.sync ? a.sync + utt * (size_t)a.sync_ustride : nullptr;
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

// Workgroup barrier that only drains LDS traffic.  __syncthreads() also waits for every outstanding
// global access (vmcnt(0), because of the queue-column stores), which would expose the L2 latency
// of the weight prefetches at each of the ~90 barriers of a sample.
__device__ __forceinline__ void dec_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// Hand-off protocol of the two-workgroup matrix-core decoder below: the recurrence is split over TWO workgroups
// (normally on two XCDs: each one's weights then fit a 4 MB L2) -
//   block 0 "chain": causal layer, per block f/g product, gate, dense product, queue update;
//   block 1 "skip" : per block the skip product Ws z_l, then relu/P1/relu/P2/softmax/argmax.
// z_l travels as {value, tag} granules written with one agent-scope relaxed 64-bit store and polled with agent-scope
// relaxed loads (tag = sample number; no fences, no flags: the placement-independent hand-off of MI355X_MICROARCH.md
// "R2 granule"); the predicted code travels back the same way.  Every spin is bounded; a timeout sets sync[err] and
// both blocks run out.
// ---------------------------------------------------------------------------------------------
// one of the three MFMAs of an x3 product (t = 0: lo*hi, 1: hi*lo, 2: hi*hi): issued term by term ACROSS independent
// accumulators, no product waits for the one in front of it
__device__ __forceinline__ void dec_term(f32x4& acc, const Frag<F16>& wa, const Frag<F16>& xb, int t) {
    acc = t == 0 ? F16::mfma(wa.lo, xb.hi, acc) : t == 1 ? F16::mfma(wa.hi, xb.lo, acc) : F16::mfma(wa.hi, xb.hi, acc);
}
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

// Matrix-core form of decode_duo_k (R = D = 64, S = Q = 256, no biases): the same two roles and hand-off protocol, 256
// threads per workgroup (4 waves: up to 512 registers each, which is what lets the weight fragments of TWO blocks sit in
// registers), every product on v_mfma_f32_16x16x32_f16 with the training engine's packed hi/lo fragments.
#define DEC_MT 256
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// The decode vectors live in LDS already split into f16 hi + lo halfs (vector v: hi halfs at v[0..n), lo halfs at
// v[n..2n)): the producer of an element splits it ONCE; the sixteen lanes that need it as a B operand read the halfs
// (at one wave per SIMD a wave64 VALU instruction costs 4 cycles: splitting 8 values x 4 fragments in every lane was
// a quarter of the chain's block time).
__device__ __forceinline__ void dec_put(uint16_t* v, int n, int i, float x) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    v[i] = __builtin_bit_cast(uint16_t, h);
    v[n + i] = __builtin_bit_cast(uint16_t, l);
}
__device__ __forceinline__ void dec_put4(uint16_t* v, int n, int i, const f32x4& x) {      // i % 4 == 0
    f16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = (_Float16)x[j]; l[j] = (_Float16)(x[j] - (float)h[j]); }
    *reinterpret_cast<f16x4*>(v + i) = h;
    *reinterpret_cast<f16x4*>(v + n + i) = l;
}
// Two-column form of the fp32-grade product: lanes of EVEN column c feed the hi halfs of the vector as their B operand,
// lanes of odd c the lo halfs, so  A.hi x B  and  A.lo x B  (2 MFMAs instead of 3, one LDS read instead of two) leave
// W x_hi in the even and W x_lo in the odd columns; dec_pairsum adds each column to its neighbour (one DPP add per register)
// and every lane holds W x again.  (Of 16 result columns a matrix-vector product needs one: two are used.)
__device__ __forceinline__ float dec_pick4(const f32x4& r, int i) { return i == 0 ? r[0] : i == 1 ? r[1] : i == 2 ? r[2] : r[3]; }
__device__ __forceinline__ f16x8 dec_get8c(const uint16_t* v, int n, int i, int c) {        // i % 8 == 0
    return *reinterpret_cast<const f16x8*>(v + (c & 1) * n + i);
}
__device__ __forceinline__ f32x4 dec_pairsum(const f32x4& v) {
    // (written as asm: through __builtin_amdgcn_update_dpp in an unrolled loop over the vector's elements this compiler
    // folded the four DPP reads into the first one - checked with tools/micro/twocol.hip; s_nop covers the
    // VALU-write -> DPP-read hazard the assembler does not see inside an asm block)
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float o;
        const float x = v[i];
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(x));
        r[i] = o;
    }
    return r;
}
__device__ __forceinline__ void dec_get8(const uint16_t* v, int n, int i, Frag<F16>& f) {   // i % 8 == 0
    f.hi = *reinterpret_cast<const f16x8*>(v + i);
    f.lo = *reinterpret_cast<const f16x8*>(v + n + i);
}

// ---------------------------------------------------------------------------------------------------------------------
// EIGHT utterances per workgroup pair (batched decoding).  A matrix-vector product uses 2 of the MFMA's 16
// result columns; here column pair (2u, 2u+1) carries utterance u of the pair's eight (x_hi | x_lo of ITS vector), so
// the same MFMAs and the same weight stream serve eight utterances and only the vector work (queue columns, gates,
// epilogues, hand-offs, softmax) is done eight times - by lanes that were idle before: lane (c, q) works for
// utterance c >> 1 on rows 4q + 2(c & 1) and + 1 of its wave's 16.  Per column the arithmetic is exactly that of
// decode_duo_mfma_k, so a row of a batch equals the single-utterance launch bit for bit.
__device__ __forceinline__ void dec_put2(uint16_t* v, int n, int i, float x0, float x1) {       // i even
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<f16x2*>(v + i) = f16x2{h0, h1};
    *reinterpret_cast<f16x2*>(v + n + i) = f16x2{l0, l1};
}
// Two adjacent granules (16-byte aligned) in one store / one load: a granule is still only trusted when ITS tag matches.
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void dec_send2(unsigned long long* p, float v0, float v1, unsigned tag) {
    u64x2 g = {dec_pack(v0, tag), dec_pack(v1, tag)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ bool dec_poll2(const unsigned long long* p, unsigned tag, float& v0, float& v1, unsigned long long* err) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        u64x2 g;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(p) : "memory");
        if ((unsigned)(g[0] >> 32) == tag && (unsigned)(g[1] >> 32) == tag) { v0 = __uint_as_float((unsigned)g[0]); v1 = __uint_as_float((unsigned)g[1]); return true; }
        if ((spin & 255) == 255 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v0 = v1 = 0.f;
    return false;
}
// T0 ("tap 0 ahead"): the tap-0 half of every block's f / g product, W_tap0 x(t - d), depends only on the queues, and all of
// them are final once the previous sample's chain pass has stored its columns.  The chain workgroup therefore forms these
// partial sums for ALL blocks of the next sample in the window in which it otherwise only waits for the other workgroup's
// code (skip sum + post-processing of the current sample, ~5 us), and a block's critical path keeps the tap-1 half only:
// 8 instead of 16 f / g MFMAs, half the LDS operand reads, half the weight re-arm loads.  The partial sums (4 floats per
// thread and block) take the place of the split queue columns in LDS (n_layers x 4 KB; chosen when that fits: <= 31 blocks).
#define DEC_T0_CHUNK 4
// S = 256 or 512 skip channels (the reference's shipped parameters have 512): MS = S / 64 row tiles of an S-vector per wave.
//
// T0 = 2: the tap-0 partial sums of more than 31 blocks do not fit LDS (4 KB per block) - they live in the pair's hand-off area
// in global memory instead (L2 resident; written a sample ahead, read back one block ahead of their use).
//
// KS > 1 (round 4): the skip sum and the post-processing are the work of KS = S / 64 workgroups per pair instead of one.  With 512
// skip channels ONE workgroup streams 128 KB of skip weights per block and 1.5 MB of post-processing weights per sample from
// L2 - 6.6 MB per sample at the ~65 GB/s a single CU gets from L2 is 100 us: the skip workgroup, not the chain, set the pace
// (9.5 k samples/s for the reference's shipped 40-block 32 / 32 / 512 model).  Part j takes rows 64 j .. 64 j + 63 of every
// S-vector - ONE 16-row tile per wave: 16 KB of skip weights per block, and its tiles of post_process_1 / post_process_2 stay in
// REGISTERS for the whole launch (2 x 128) - so the tail behind the last block has no weight traffic at all.  The S-vectors
// (skip sum, post_process_1 output) are exchanged between the parts as tagged granules like z (all-gather: every part
// publishes its 64 rows and polls the others'), the logits go to part 0, which chooses the codes.
template <bool BIAS, int T0, int S = 256, int KS = 1>
__global__ __launch_bounds__(DEC_MT) void decode_duo_mfma8_k(WnDecodeArgs a) {
    constexpr int NU = 8, R = 64, D = 64, Q = 256, MS = S / 64;
    static_assert(KS == 1 || KS == S / 64, "one row tile per wave in the split form");
    // LDS strides between utterances (halfs): the eight utterances of a 16-lane group read 16-byte pieces at the same offset of
    // their own vectors - with the natural strides (256 B, 1 KB) all of them in the same banks.  +16 B per utterance spreads the
    // group over all 64 banks (hi | lo halves are 128 B apart for the 64-vectors; 640 B for the 256-vectors)
    constexpr int VS = 2 * R + 8, WLO = S + 64, WS = 2 * WLO + 8;       // (S = 256: 320 and 648)
    const int pair = blockIdx.x / (1 + KS), role = blockIdx.x - pair * (1 + KS);
    const size_t ubase = (size_t)pair * NU;
    // a pair with fewer than eight utterances left (n_utt not a multiple of 8; a single utterance): the spare columns MIRROR the
    // last real one - same inputs, same arithmetic, same addresses, so their stores only repeat identical values
    const size_t ulast = (size_t)(a.n_utt > 0 ? a.n_utt : 1) - 1;
    auto ux = [&](int uu) { const size_t g = ubase + uu; return g < ulast ? g : ulast; };
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, q = lane >> 4;
    const int u = c >> 1, h = c & 1;                       // this lane's utterance of the eight and its half (hi / lo columns)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int s_code[NU], s_pc[NU], s_nc[NU];
    __shared__ int slots[WN_DEC_MAX_LAYERS];
    // hand-off area of an utterance (8-byte granules): z [n_layers][D] | skip vector [S] | post_process_1 output [S] | logits [Q] |
    // (first utterance of a pair, T0 = 2: tap-0 table [n_layers][256] x 16 bytes) | code | error flag
    auto zg_of = [&](int uu) { return a.sync + ux(uu) * (size_t)a.sync_ustride; };
    auto sv_of = [&](int uu) { return zg_of(uu) + (size_t)a.n_layers * D; };
    auto hv_of = [&](int uu) { return sv_of(uu) + S; };
    auto lg_of = [&](int uu) { return hv_of(uu) + S; };
    auto cg_of = [&](int uu) { return a.sync + (ux(uu) + 1) * (size_t)a.sync_ustride - 2; };
    unsigned long long* const zg = zg_of(u);                 // [n_layers][D] z granules of this lane's utterance

    if (role == 0) {
        // ------------------------------------------------------------------ chain
        constexpr int BL = 2 * D + R;
        float* prev = sm;                                   // [NU][Q]
        float* note = prev + NU * Q;                        // [NU][Q]
        float* xc0 = note + NU * Q;                         // [NU][R] fp32 residual stream (two buffers)
        float* xc1 = xc0 + NU * R;
        float* bias = xc1 + NU * R;                         // [n_layers][bf D | bg D | bd R] (BIAS)
        // (T0: no bias table in LDS - the f / g biases ride in the tap-0 partial sums, a block's dense bias is fetched at its top)
        uint16_t* xh0 = reinterpret_cast<uint16_t*>(bias + (BIAS && a.b_layers && !T0 ? a.n_layers * BL : 0));   // [NU][hi R | lo R]
        uint16_t* xh1 = xh0 + NU * VS;
        uint16_t* zh = xh1 + NU * VS;                    // [NU][hi D | lo D], chained k order
        uint16_t* oldh = zh + NU * VS;                   // [n_layers][NU][hi R | lo R] queue columns of this sample (T0: of DEC_T0_CHUNK blocks)
        // T0: [n_layers][256] tap-0 partial sums (f0, f1, g0, g1): in LDS (T0 = 1) or in the pair's hand-off area (T0 = 2)
        f32x4* part = T0 == 2 ? reinterpret_cast<f32x4*>(lg_of(0) + Q) : reinterpret_cast<f32x4*>(oldh + (size_t)DEC_T0_CHUNK * NU * VS);
        for (int i = tid; i < NU * Q; i += 256) { const int uu = i / Q, e = i - uu * Q; note[i] = a.note0[ux(uu) * Q + e]; prev[i] = a.prev0[ux(uu) * Q + e]; }
        if (BIAS && a.b_layers && !T0) for (int i = tid; i < a.n_layers * BL; i += 256) { const int l = i / BL, e = i - l * BL; bias[i] = a.b_layers[(size_t)l * (BL + S) + e]; }
        if (tid < a.n_layers) slots[tid] = (int)(a.step0 % a.dil[tid]);
        if (tid < NU) { s_pc[tid] = -1; s_nc[tid] = -1; }
        dec_sync();
        const uint16_t* fgb = a.pk + a.pk_fg0;
        const uint16_t* db = a.pk + a.pk_d0;
        float* const uq = a.queues + ux(u) * (size_t)a.queues_ustride;       // this lane's utterance
        const int ra = 16 * w + 4 * q + 2 * h;                                       // its two rows: ra, ra + 1
        // the queue columns x(t - d) of ALL blocks for one sample, split into halfs: a batch of 32 loads per thread in flight
        // (hand-off-scope loads go out to memory: ~1 us alone, 2-3 us on a busy chip), then the splits.  Called for sample
        // t + 1 while the chain waits for the codes of sample t (the columns do not depend on them).
        auto load_queues = [&]() {
            // four consecutive channels per load (a queue column is 64 contiguous floats): 15 loads of 16 bytes per thread for
            // 30 blocks x 8 utterances, all in flight, then the splits.  Non-temporal loads: the columns were stored by THIS
            // workgroup (write-through L1, completed by the barrier before this call) or by an earlier launch, and are read once
            const int n_q4 = a.n_layers * NU * (R / 4);
            for (int i0 = 0; i0 < n_q4; i0 += 16 * 256) {
                f32x4 qv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int e = i0 + j * 256 + tid, ec = e < n_q4 ? e : 0;
                    const int l = ec / (NU * (R / 4)), uu = (ec >> 4) & (NU - 1), r4 = ec & 15;
                    qv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(
                        a.queues + ux(uu) * (size_t)a.queues_ustride + a.q_off[l] + (size_t)slots[l] * R + 4 * r4));
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int e = i0 + j * 256 + tid;
                    if (e < n_q4) dec_put4(oldh + (size_t)(e >> 4) * VS, R, 4 * (e & 15), qv[j]);       // (e >> 4) = l * NU + uu
                }
            }
        };
        // T0: queue columns of DEC_T0_CHUNK blocks at a time -> halfs in LDS -> this wave's 16 rows of W_tap0 x for f and g of each
        // block -> part[l][tid].  (Same two-column products and pair sums as the block itself runs for its tap-1 half.)
        auto tap0_ahead = [&]() {
            const uint16_t* fgw = a.pk + a.pk_fg0;
            Frag<F16> tf[2][2], tg[2][2];                   // [set][k-step 0, 1] of block l (set l & 1), re-armed one block ahead
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                load_a<F16, 3>(tf[0][ks], fgw, w * 4 + ks, lane);
                load_a<F16, 3>(tg[0][ks], fgw, (4 + w) * 4 + ks, lane);
            }
            // the f / g biases of this thread's two rows, fetched one block ahead like the weights (a load issued where it is used
            // would put an L2 round trip per block into this window)
            auto bias4 = [&](int l) {
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (BIAS && a.b_layers) {
                    const float* bl = a.b_layers + (size_t)l * (BL + S);
                    b = f32x4{bl[ra], bl[ra + 1], bl[D + ra], bl[D + ra + 1]};
                }
                return b;
            };
            f32x4 bcur = bias4(0);
            for (int l0 = 0; l0 < a.n_layers; l0 += DEC_T0_CHUNK) {
                // DEC_T0_CHUNK blocks x 8 utterances x 16 float4 = 512 loads of 16 bytes: two per thread
                f32x4 qv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = j * 256 + tid;
                    int l = l0 + (e >> 7);
                    l = l < a.n_layers ? l : a.n_layers - 1;
                    const int uu = (e >> 4) & (NU - 1), r4 = e & 15;
                    qv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(
                        a.queues + ux(uu) * (size_t)a.queues_ustride + a.q_off[l] + (size_t)slots[l] * R + 4 * r4));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = j * 256 + tid;
                    dec_put4(oldh + (size_t)(e >> 4) * VS, R, 4 * (e & 15), qv[j]);       // (e >> 4) = (l - l0) * NU + uu
                }
                dec_sync();
#pragma unroll
                for (int lc = 0; lc < DEC_T0_CHUNK; ++lc) {
                    const int l = l0 + lc;
                    if (l < a.n_layers) {
                        const int ln = l + 1 < a.n_layers ? l + 1 : l;
                        const uint16_t* fgn = fgw + (size_t)ln * a.pk_lstride;
                        Frag<F16> (&cf)[2] = tf[lc & 1];
                        Frag<F16> (&cg)[2] = tg[lc & 1];
                        Frag<F16> (&nf)[2] = tf[(lc + 1) & 1];
                        Frag<F16> (&ng)[2] = tg[(lc + 1) & 1];
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            load_a<F16, 3>(nf[ks], fgn, w * 4 + ks, lane);
                            load_a<F16, 3>(ng[ks], fgn, (4 + w) * 4 + ks, lane);
                        }
                        const f32x4 bnext = bias4(ln);
                        const uint16_t* ob = oldh + ((size_t)lc * NU + u) * VS + h * R;
                        f16x8 bx[2];
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) bx[ks] = *reinterpret_cast<const f16x8*>(ob + 32 * ks + 8 * q);
                        f32x4 pf[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, pg[2] = {pf[0], pf[0]};
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) { pf[ks] = F16::mfma(cf[ks].hi, bx[ks], pf[ks]); pg[ks] = F16::mfma(cg[ks].hi, bx[ks], pg[ks]); }
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) { pf[ks] = F16::mfma(cf[ks].lo, bx[ks], pf[ks]); pg[ks] = F16::mfma(cg[ks].lo, bx[ks], pg[ks]); }
                        const f32x4 af = dec_pairsum(pf[0] + pf[1]), ag = dec_pairsum(pg[0] + pg[1]);
                        f32x4 pv = {h ? af[2] : af[0], h ? af[3] : af[1], h ? ag[2] : ag[0], h ? ag[3] : ag[1]};
                        pv += bcur;                          // the block's f / g biases ride along
                        bcur = bnext;
                        part[(size_t)l * 256 + tid] = pv;
                    }
                }
                dec_sync();                                  // the chunk's halfs are consumed: the next chunk may overwrite them
            }
            if (T0 == 2) {                                   // the table's stores have left this CU before anybody reads it back
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                dec_sync();
            }
        };
        if (T0) tap0_ahead();
        else load_queues();
        float bdn0 = 0.f, bdn1 = 0.f;                        // T0: the next block's dense bias (two rows of this thread)
        f32x4 t0n = {0.f, 0.f, 0.f, 0.f};                    // T0 = 2: the next block's partial sums, fetched a block ahead (L2 round trip)
        for (int step = 0; step < a.n_steps; ++step) {
            const unsigned tag = (unsigned)step + 1u;
            if (T0 == 2) t0n = __builtin_nontemporal_load(part + tid);
            // weight fragments: two register sets, each re-armed two blocks ahead (as in decode_duo_mfma_k)
            Frag<F16> wfA[4], wgA[4], wdA[2], wfB[4], wgB[4], wdB[2];
            const size_t lb1 = a.n_layers > 1 ? (size_t)a.pk_lstride : 0;
#pragma unroll
            for (int s2 = T0 ? 2 : 0; s2 < 4; ++s2) {             // T0: the tap-0 k-steps (0, 1) were multiplied a sample ahead
                load_a<F16, 3>(wfA[s2], fgb, w * 4 + s2, lane);
                load_a<F16, 3>(wgA[s2], fgb, (4 + w) * 4 + s2, lane);
                load_a<F16, 3>(wfB[s2], fgb + lb1, w * 4 + s2, lane);
                load_a<F16, 3>(wgB[s2], fgb + lb1, (4 + w) * 4 + s2, lane);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                load_a<F16, 3>(wdA[s2], db, w * 2 + s2, lane);
                load_a<F16, 3>(wdB[s2], db + lb1, w * 2 + s2, lane);
            }
            {   // causal layer, 2 rows of one utterance per thread: x0 = Wc[:, tap0] prev + Wc[:, tap1] note (one-hot: a gather)
                const int uu = tid >> 5, o0 = (tid & 31) * 2;
                const int pc = s_pc[uu], nc = s_nc[uu];
                float sv[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float* wrow = a.w_causal + (size_t)(o0 + e) * 2 * Q;
                    float t = 0.f;
                    if (pc >= 0) t += wrow[pc];
                    else for (int kk = 0; kk < Q; ++kk) t = fmaf(wrow[kk], prev[uu * Q + kk], t);
                    if (nc >= 0) t += wrow[Q + nc];
                    else for (int kk = 0; kk < Q; ++kk) t = fmaf(wrow[Q + kk], note[uu * Q + kk], t);
                    if (BIAS && a.b_causal) t += a.b_causal[o0 + e];
                    sv[e] = t;
                }
                xc0[uu * R + o0] = sv[0]; xc0[uu * R + o0 + 1] = sv[1];
                dec_put2(xh0 + uu * VS, R, o0, sv[0], sv[1]);
            }
            dec_sync();
            float* cur = xc0;
            float* nxt = xc1;
            uint16_t* curh = xh0;
            uint16_t* nxth = xh1;
            if (BIAS && a.b_layers && T0 && step == 0) { bdn0 = a.b_layers[2 * D + ra]; bdn1 = a.b_layers[2 * D + ra + 1]; }
            auto blk = [&](const int l, Frag<F16> (&wf)[4], Frag<F16> (&wg)[4], Frag<F16> (&wd2)[2]) {
                const int l2 = l + 2 < a.n_layers ? l + 2 : a.n_layers - 1;      // the set's next use (clamped: harmless reload)
                const uint16_t* fgn = fgb + (size_t)l2 * a.pk_lstride;
                const uint16_t* dn = db + (size_t)l2 * a.pk_lstride;
                const uint16_t* ob = oldh + ((size_t)l * NU + u) * VS + h * R;
                const uint16_t* xb = curh + u * VS + h * R;
                f16x8 bx[4];
#pragma unroll
                for (int ks = T0 ? 2 : 0; ks < 4; ++ks) bx[ks] = *reinterpret_cast<const f16x8*>(ks < 2 ? ob + 32 * ks + 8 * q : xb + 32 * (ks - 2) + 8 * q);
                f32x4 pf[4], pg[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { pf[ks] = f32x4{0.f, 0.f, 0.f, 0.f}; pg[ks] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int ks = T0 ? 2 : 0; ks < 4; ++ks) { pf[ks] = F16::mfma(wf[ks].hi, bx[ks], pf[ks]); pg[ks] = F16::mfma(wg[ks].hi, bx[ks], pg[ks]); }
#pragma unroll
                for (int ks = T0 ? 2 : 0; ks < 4; ++ks) { pf[ks] = F16::mfma(wf[ks].lo, bx[ks], pf[ks]); pg[ks] = F16::mfma(wg[ks].lo, bx[ks], pg[ks]); }
#pragma unroll
                for (int ks = T0 ? 2 : 0; ks < 4; ++ks) {
                    load_a<F16, 3>(wf[ks], fgn, w * 4 + ks, lane);
                    load_a<F16, 3>(wg[ks], fgn, (4 + w) * 4 + ks, lane);
                }
                float f0, f1, g0, g1;
                if (T0) {
                    f32x4 t0;                                                       // W_tap0 x(t - d), formed a sample ahead
                    if (T0 == 2) {
                        t0 = t0n;
                        t0n = __builtin_nontemporal_load(part + (size_t)(l + 1 < a.n_layers ? l + 1 : l) * 256 + tid);
                    } else {
                        t0 = part[(size_t)l * 256 + tid];
                    }
                    const f32x4 af = dec_pairsum(pf[2] + pf[3]), ag = dec_pairsum(pg[2] + pg[3]);
                    f0 = (h ? af[2] : af[0]) + t0[0]; f1 = (h ? af[3] : af[1]) + t0[1];
                    g0 = (h ? ag[2] : ag[0]) + t0[2]; g1 = (h ? ag[3] : ag[1]) + t0[3];
                } else {
                    const f32x4 af = dec_pairsum((pf[0] + pf[1]) + (pf[2] + pf[3])), ag = dec_pairsum((pg[0] + pg[1]) + (pg[2] + pg[3]));
                    f0 = h ? af[2] : af[0]; f1 = h ? af[3] : af[1]; g0 = h ? ag[2] : ag[0]; g1 = h ? ag[3] : ag[1];
                }
                if (BIAS && a.b_layers && !T0) {
                    f0 += bias[l * BL + ra]; f1 += bias[l * BL + ra + 1];
                    g0 += bias[l * BL + D + ra]; g1 += bias[l * BL + D + ra + 1];
                }
                const float bd0 = bdn0, bd1 = bdn1;          // T0: this block's dense bias was fetched during the block before
                if (BIAS && a.b_layers && T0) {
                    const int ln1 = l + 1 < a.n_layers ? l + 1 : 0;              // (block 0's for the next sample)
                    const float* bl = a.b_layers + (size_t)ln1 * (BL + S) + 2 * D;
                    bdn0 = bl[ra]; bdn1 = bl[ra + 1];
                }
                const float z0 = wn_tanh(f0) * wn_sigmoid(g0), z1 = wn_tanh(f1) * wn_sigmoid(g1);
                dec_put2(zh + u * VS, D, 32 * (w >> 1) + 8 * q + 4 * (w & 1) + 2 * h, z0, z1);     // chained k order of the dense weights
                // both granules in ONE 16-byte store (ra is even): each 8-byte granule still validates itself by its tag
                dec_send2(zg + (size_t)l * D + ra, z0, z1, tag);
                dec_sync();
                f16x8 bz[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) bz[s2] = *reinterpret_cast<const f16x8*>(zh + u * VS + h * D + 32 * s2 + 8 * q);
                f32x4 pd[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) pd[s2] = F16::mfma(wd2[s2].hi, bz[s2], pd[s2]);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) pd[s2] = F16::mfma(wd2[s2].lo, bz[s2], pd[s2]);
                const f32x4 ad = dec_pairsum(pd[0] + pd[1]);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) load_a<F16, 3>(wd2[s2], dn, w * 2 + s2, lane);
                const float xa = cur[u * R + ra], xb1 = cur[u * R + ra + 1];
                float v0 = (h ? ad[2] : ad[0]) + xa, v1 = (h ? ad[3] : ad[1]) + xb1;
                if (BIAS && a.b_layers) {
                    if (T0) { v0 += bd0; v1 += bd1; }
                    else { v0 += bias[l * BL + 2 * D + ra]; v1 += bias[l * BL + 2 * D + ra + 1]; }
                }
                nxt[u * R + ra] = v0; nxt[u * R + ra + 1] = v1;
                dec_put2(nxth + u * VS, R, ra, v0, v1);
                float* qd = uq + a.q_off[l] + (size_t)slots[l] * R + ra;          // the slot read at the top of this sample
                qd[0] = a.push_input ? xa : v0;                                     // Q5: output by default
                qd[1] = a.push_input ? xb1 : v1;
                dec_sync();
                float* t2 = cur; cur = nxt; nxt = t2;
                uint16_t* t3 = curh; curh = nxth; nxth = t3;
            };
            for (int l = 0; l < a.n_layers; l += 2) {
                blk(l, wfA, wgA, wdA);
                if (l + 1 < a.n_layers) blk(l + 1, wfB, wgB, wdB);
            }
            __syncthreads();                       // all queue stores of this sample are complete (dilation 1 reads them back next)
            if (tid < a.n_layers) { int sl = slots[tid] + 1; slots[tid] = sl == a.dil[tid] ? 0 : sl; }
            dec_sync();
            if (step + 1 < a.n_steps) { if (T0) tap0_ahead(); else load_queues(); }
            if (tid < NU) {
                float cv = 0.f;
                dec_poll(cg_of(tid), tag, cv, cg_of(tid) + 1);
                s_code[tid] = a.forced ? a.forced[ux(tid) * a.n_steps + step] : (int)cv;
            }
            dec_sync();
            for (int i = tid; i < NU * Q; i += 256) prev[i] = note[i];
            if (tid < NU) { s_pc[tid] = s_nc[tid]; s_nc[tid] = s_code[tid]; }
            dec_sync();
            for (int i = tid; i < NU * Q; i += 256) note[i] = ((i & (Q - 1)) == s_code[i / Q]) ? 1.0f : 0.0f;
            dec_sync();
        }
        for (int i = tid; i < NU * Q; i += 256) { const int uu = i / Q, e = i - uu * Q; a.prev_out[ux(uu) * Q + e] = prev[i]; a.note_out[ux(uu) * Q + e] = note[i]; }
    } else if (KS > 1) {
        // ------------------------------------------------------------------ skip sum + post-processing, part j of KS: one row tile per wave
        constexpr int KP = S / 32;                           // k-steps of a product over an S-vector
        const int jpart = role - 1;
        const int tile = 4 * jpart + w;                      // this wave's 16 rows of the skip sum and of post_process_1
        const bool has_p2 = tile < Q / 16;                   // ... and of post_process_2 (Q / 16 tiles: the first waves)
        uint16_t* zz0 = reinterpret_cast<uint16_t*>(sm);     // [2][NU][hi D | lo D]
        uint16_t* skip = zz0 + 2 * NU * VS;                  // [NU][hi S | lo S]: the WHOLE vector (own rows + the other parts')
        uint16_t* h1 = skip + NU * WS;                       // [NU][hi S | lo S]
        float* logit = reinterpret_cast<float*>(h1 + NU * WS);   // [NU][Q] (part 0)
        float* bsk = logit + NU * Q;                         // [S] summed skip biases, [S] post_process_1 bias, [Q] post_process_2 bias
        if (BIAS && a.b_layers) {
            for (int r = tid; r < S; r += 256) {
                float t = 0.f;
                for (int l = 0; l < a.n_layers; ++l) t += a.b_layers[(size_t)l * (2 * D + R + S) + 2 * D + R + r];
                bsk[r] = t;
            }
        }
        if (BIAS && a.b_p1) for (int r = tid; r < S; r += 256) bsk[S + r] = a.b_p1[r];
        if (BIAS && a.b_p2) bsk[2 * S + tid] = a.b_p2[tid];
        dec_sync();
        const int KSS = a.n_layers * D / 32;
        const uint16_t* skb = a.pk + a.pk_skip;
        // this wave's tiles of post_process_1 and post_process_2: 2 x KP fragments, in registers for the whole launch
        Frag<F16> wp1[KP], wp2[KP];
#pragma unroll
        for (int ks = 0; ks < KP; ++ks) {
            load_a<F16, 3>(wp1[ks], a.pk + a.pk_p1, tile * KP + ks, lane);
            load_a<F16, 3>(wp2[ks], a.pk + a.pk_p2, (has_p2 ? tile : 0) * KP + ks, lane);
        }
        const int row = 16 * tile + 4 * q + 2 * h;           // this lane's two rows (of utterance u): row, row + 1
        // all-gather of an S-vector (or of the logits): thread (utterance uu = tid >> 5, i = tid & 31) polls the granule pairs
        // i, i + 32, .. of that utterance - own rows included (they come back from L2 like everybody else's) - until every tag
        // is this sample's, and leaves the values split in LDS (vector) or as floats (logits)
        auto gather_rows = [&](const unsigned long long* base, const int n_rows, uint16_t* vec, float* flt, const unsigned tag, unsigned long long* err) __attribute__((always_inline)) {
            constexpr int NLMAX = S / 64;                    // pairs per thread of an S-vector (logits: Q / 64)
            const int nl = n_rows / 64;
            const int uu = tid >> 5, i = tid & 31;
            unsigned long long g[2 * NLMAX];
            for (int spin = 0; spin < (1 << 22); ++spin) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < NLMAX; ++k) {
                    const unsigned long long* p = base + 2 * (size_t)(i + 32 * (k < nl ? k : 0));
                    g[2 * k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g[2 * k + 1] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < NLMAX; ++k)
                    if (k < nl) ok = ok && (unsigned)(g[2 * k] >> 32) == tag && (unsigned)(g[2 * k + 1] >> 32) == tag;
                if (ok) break;
                if ((spin & 255) == 255 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (spin == (1 << 22) - 1) __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < NLMAX; ++k) {
                if (k < nl) {
                    const int r2 = 2 * (i + 32 * k);
                    const float v0 = __uint_as_float((unsigned)g[2 * k]), v1 = __uint_as_float((unsigned)g[2 * k + 1]);
                    if (vec) dec_put2(vec + uu * WS, WLO, r2, v0, v1);
                    else { flt[uu * Q + r2] = v0; flt[uu * Q + r2 + 1] = v1; }
                }
            }
        };
        const unsigned long long* const zmine = zg_of(tid >> 5) + (tid & 31) * 2;       // this thread's two granules of block 0
        unsigned long long* const errp = cg_of(tid >> 5) + 1;
        unsigned long long pa = 0, pb = 0;                                               // prefetched pair (tag 0 = nothing yet)
        for (int step = 0; step < a.n_steps; ++step) {
            const unsigned tag = (unsigned)step + 1u;
            f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            Frag<F16> ws[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) load_a<F16, 3>(ws[s2], skb, tile * KSS + s2, lane);
            for (int l = 0; l < a.n_layers; ++l) {
                uint16_t* zz = zz0 + (l & 1) * NU * VS;
                {   // z of block l (asked for one block earlier), as in the one-workgroup form
                    float za, zb;
                    if ((unsigned)(pa >> 32) == tag && (unsigned)(pb >> 32) == tag) { za = __uint_as_float((unsigned)pa); zb = __uint_as_float((unsigned)pb); }
                    else dec_poll2(zmine + (size_t)l * D, tag, za, zb, errp);
                    const unsigned long long* nx = zmine + (size_t)(l + 1 < a.n_layers ? l + 1 : 0) * D;      // (block 0: the next sample's)
                    pa = __hip_atomic_load(nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pb = __hip_atomic_load(nx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dec_put2(zz + (tid >> 5) * VS, D, (tid & 31) * 2, za, zb);
                }
                dec_sync();
                const int ln = l + 1 < a.n_layers ? l + 1 : l;
                const f16x8 bz[2] = {*reinterpret_cast<const f16x8*>(zz + u * VS + h * D + 8 * q),
                                     *reinterpret_cast<const f16x8*>(zz + u * VS + h * D + 32 + 8 * q)};
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) acc2[s2] = F16::mfma(ws[s2].hi, bz[s2], acc2[s2]);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) acc2[s2] = F16::mfma(ws[s2].lo, bz[s2], acc2[s2]);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) load_a<F16, 3>(ws[s2], skb, tile * KSS + 2 * ln + s2, lane);      // next block (unconditional)
            }
            {   // this wave's rows of relu(skip sum): published, then the whole vector gathered
                const f32x4 t = dec_pairsum(acc2[0] + acc2[1]);
                float v0 = h ? t[2] : t[0], v1 = h ? t[3] : t[1];
                if (BIAS && a.b_layers) { v0 += bsk[row]; v1 += bsk[row + 1]; }
                dec_send2(sv_of(u) + row, fmaxf(v0, 0.f), fmaxf(v1, 0.f), tag);
            }
            gather_rows(sv_of(tid >> 5), S, skip, nullptr, tag, errp);
            dec_sync();
            {   // post_process_1: this wave's tile
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < KP; ++ks) {
                    const f16x8 bx = *reinterpret_cast<const f16x8*>(skip + u * WS + h * WLO + 32 * ks + 8 * q);
                    acc[ks & 1] = F16::mfma(wp1[ks].hi, bx, acc[ks & 1]);
                    acc[ks & 1] = F16::mfma(wp1[ks].lo, bx, acc[ks & 1]);
                }
                const f32x4 t = dec_pairsum(acc[0] + acc[1]);
                float v0 = h ? t[2] : t[0], v1 = h ? t[3] : t[1];
                if (BIAS && a.b_p1) { v0 += bsk[S + row]; v1 += bsk[S + row + 1]; }
                dec_send2(hv_of(u) + row, fmaxf(v0, 0.f), fmaxf(v1, 0.f), tag);
            }
            gather_rows(hv_of(tid >> 5), S, h1, nullptr, tag, errp);
            dec_sync();
            if (has_p2) {   // post_process_2: the first Q / 16 waves of the pair's parts hold a tile each
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < KP; ++ks) {
                    const f16x8 bx = *reinterpret_cast<const f16x8*>(h1 + u * WS + h * WLO + 32 * ks + 8 * q);
                    acc[ks & 1] = F16::mfma(wp2[ks].hi, bx, acc[ks & 1]);
                    acc[ks & 1] = F16::mfma(wp2[ks].lo, bx, acc[ks & 1]);
                }
                const f32x4 t = dec_pairsum(acc[0] + acc[1]);
                float v0 = h ? t[2] : t[0], v1 = h ? t[3] : t[1];
                if (BIAS && a.b_p2) { v0 += bsk[2 * S + row]; v1 += bsk[2 * S + row + 1]; }
                dec_send2(lg_of(u) + row, v0, v1, tag);
            }
            if (jpart == 0) {   // part 0 chooses: the logits of all parts, then as the one-workgroup form
                gather_rows(lg_of(tid >> 5), Q, nullptr, logit, tag, errp);
                dec_sync();
#pragma unroll
                for (int e = 0; e < 2; ++e) {          // wave w chooses for utterances 2w and 2w + 1
                    const int uu = 2 * w + e;
                    const size_t ug = ux(uu);
                    const float ur = a.sample ? dec_uniform(a.seed, (unsigned long long)(a.step0 + step), ug) : 0.f;
                    float* pdst = a.probs_out ? a.probs_out + (ug * (size_t)a.n_steps + step) * Q : nullptr;
                    const int bi = dec_choose(logit + uu * Q, lane, pdst, a.inv_temp, a.sample != 0, ur);
                    if (lane == 0) {
                        a.codes_out[ug * a.n_steps + step] = bi;
                        __hip_atomic_store(cg_of(uu), dec_pack((float)bi, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            dec_sync();
        }
    } else {
        // ------------------------------------------------------------------ skip sum + post-processing
        uint16_t* zz0 = reinterpret_cast<uint16_t*>(sm);        // [2][NU][hi D | lo D]
        uint16_t* skip = zz0 + 2 * NU * VS;                  // [NU][hi S | lo S]
        uint16_t* h1 = skip + NU * WS;                       // [NU][hi S | lo S]
        float* logit = reinterpret_cast<float*>(h1 + NU * WS);   // [NU][Q]
        float* bsk = logit + NU * Q;                            // [S] summed skip biases, [S] post_process_1 bias, [Q] post_process_2 bias
        if (BIAS && a.b_layers) {
            for (int r = tid; r < S; r += 256) {
                float t = 0.f;
                for (int l = 0; l < a.n_layers; ++l) t += a.b_layers[(size_t)l * (2 * D + R + S) + 2 * D + R + r];
                bsk[r] = t;
            }
        }
        if (BIAS && a.b_p1) for (int r = tid; r < S; r += 256) bsk[S + r] = a.b_p1[r];
        if (BIAS && a.b_p2) bsk[2 * S + tid] = a.b_p2[tid];
        dec_sync();
        const int KSS = a.n_layers * D / 32;
        const uint16_t* skb = a.pk + a.pk_skip;
        const uint16_t* p1b = a.pk + a.pk_p1;
        const uint16_t* p2b = a.pk + a.pk_p2;
        // NG groups of 256 output rows, four row tiles per wave and group (rows 256 g + 64 w + 16 m + 4q + 2h, + 1 of this lane's
        // utterance), K = S.  (All 8 tiles of a 512-row product at once: the unrolled k loop's fragment loads are hoisted and
        // 480 registers spill.)
        auto post = [&](const int ng, const uint16_t* wb, const uint16_t* in, uint16_t* outh, float* outf, const float* bvec) {
            constexpr int KP = S / 32;
            for (int gr = 0; gr < ng; ++gr) {
                f32x4 acc[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int rt0 = 16 * gr + 4 * w;
                Frag<F16> wa[2][4];
#pragma unroll
                for (int m = 0; m < 4; ++m) load_a<F16, 3>(wa[0][m], wb, (rt0 + m) * KP, lane);
#pragma unroll
                for (int ks = 0; ks < KP; ++ks) {
                    if (ks + 1 < KP) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) load_a<F16, 3>(wa[(ks + 1) & 1][m], wb, (rt0 + m) * KP + ks + 1, lane);
                    }
                    const f16x8 bx = *reinterpret_cast<const f16x8*>(in + u * WS + h * WLO + 32 * ks + 8 * q);
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = F16::mfma(wa[ks & 1][m].hi, bx, acc[m]);
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = F16::mfma(wa[ks & 1][m].lo, bx, acc[m]);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x4 t = dec_pairsum(acc[m]);
                    const int row = 16 * (rt0 + m) + 4 * q + 2 * h;
                    float v0 = h ? t[2] : t[0], v1 = h ? t[3] : t[1];
                    if (bvec) { v0 += bvec[row]; v1 += bvec[row + 1]; }
                    if (outh) dec_put2(outh + u * WS, WLO, row, fmaxf(v0, 0.f), fmaxf(v1, 0.f));
                    else { outf[u * Q + row] = v0; outf[u * Q + row + 1] = v1; }
                }
            }
        };
        const unsigned long long* const zmine = zg_of(tid >> 5) + (tid & 31) * 2;       // this thread's two granules of block 0
        unsigned long long pa = 0, pb = 0;                                               // prefetched pair (tag 0 = nothing yet)
        for (int step = 0; step < a.n_steps; ++step) {
            const unsigned tag = (unsigned)step + 1u;
            f32x4 acc2[2][MS];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int m = 0; m < MS; ++m) acc2[s2][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            Frag<F16> ws[2][MS];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int m = 0; m < MS; ++m) load_a<F16, 3>(ws[s2][m], skb, (16 * (m >> 2) + 4 * w + (m & 3)) * KSS + s2, lane);
            for (int l = 0; l < a.n_layers; ++l) {
                uint16_t* zz = zz0 + (l & 1) * NU * VS;
                {   // two adjacent granules per thread.  They were asked for one block EARLIER (a hand-off-scope load is a
                    // round trip to memory, 2-3 us on a busy chip): usually they are there; else poll, one 16-byte load per try
                    float za, zb;
                    if ((unsigned)(pa >> 32) == tag && (unsigned)(pb >> 32) == tag) { za = __uint_as_float((unsigned)pa); zb = __uint_as_float((unsigned)pb); }
                    else dec_poll2(zmine + (size_t)l * D, tag, za, zb, cg_of(tid >> 5) + 1);
                    const unsigned long long* nx = zmine + (size_t)(l + 1 < a.n_layers ? l + 1 : 0) * D;      // (block 0: the next sample's)
                    pa = __hip_atomic_load(nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pb = __hip_atomic_load(nx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dec_put2(zz + (tid >> 5) * VS, D, (tid & 31) * 2, za, zb);
                }
                dec_sync();
                const int ln = l + 1 < a.n_layers ? l + 1 : l;
                const f16x8 bz[2] = {*reinterpret_cast<const f16x8*>(zz + u * VS + h * D + 8 * q),
                                     *reinterpret_cast<const f16x8*>(zz + u * VS + h * D + 32 + 8 * q)};
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int m = 0; m < MS; ++m) acc2[s2][m] = F16::mfma(ws[s2][m].hi, bz[s2], acc2[s2][m]);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int m = 0; m < MS; ++m) acc2[s2][m] = F16::mfma(ws[s2][m].lo, bz[s2], acc2[s2][m]);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int m = 0; m < MS; ++m) load_a<F16, 3>(ws[s2][m], skb, (16 * (m >> 2) + 4 * w + (m & 3)) * KSS + 2 * ln + s2, lane);      // next block (unconditional)
            }
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                const f32x4 t = dec_pairsum(acc2[0][m] + acc2[1][m]);
                const int row = 16 * (16 * (m >> 2) + 4 * w + (m & 3)) + 4 * q + 2 * h;
                float v0 = h ? t[2] : t[0], v1 = h ? t[3] : t[1];
                if (BIAS && a.b_layers) { v0 += bsk[row]; v1 += bsk[row + 1]; }
                dec_put2(skip + u * WS, WLO, row, fmaxf(v0, 0.f), fmaxf(v1, 0.f));
            }
            dec_sync();
            post(S / 256, p1b, skip, h1, nullptr, BIAS && a.b_p1 ? bsk + S : nullptr);
            dec_sync();
            post(Q / 256, p2b, h1, nullptr, logit, BIAS && a.b_p2 ? bsk + 2 * S : nullptr);
            dec_sync();
#pragma unroll
            for (int e = 0; e < 2; ++e) {          // wave w chooses for utterances 2w and 2w + 1
                const int uu = 2 * w + e;
                const size_t ug = ux(uu);
                const float ur = a.sample ? dec_uniform(a.seed, (unsigned long long)(a.step0 + step), ug) : 0.f;
                float* pdst = a.probs_out ? a.probs_out + (ug * (size_t)a.n_steps + step) * Q : nullptr;
                const int bi = dec_choose(logit + uu * Q, lane, pdst, a.inv_temp, a.sample != 0, ur);
                if (lane == 0) {
                    a.codes_out[ug * a.n_steps + step] = bi;
                    __hip_atomic_store(cg_of(uu), dec_pack((float)bi, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            dec_sync();
        }
    }
}

// granules of one utterance's hand-off area: z of every block | skip vector | post_process_1 output | logits | the pair's tap-0
// table (deeper than 31 blocks: it does not fit LDS) | code | error flag
// LDS bytes of the chain workgroup with the tap-0 table in LDS (T0 = 1; the matrix-core path has Q = 256, R = 64 inside)
static size_t dec_t0_lds_bytes(int n_layers) {
    return sizeof(float) * (size_t)(16 * 256 + 16 * 64) + sizeof(uint16_t) * (size_t)(8 * 136 * (3 + DEC_T0_CHUNK)) + (size_t)n_layers * 256 * sizeof(f32x4);
}
static bool dec_t0_fits_lds(int n_layers) { return dec_t0_lds_bytes(n_layers) + 1024 <= 160 * 1024; }      // up to 31 blocks
long wn_decode_granules(int n_layers, int D, int S) {
    return (long)n_layers * D + 2 * (long)S + 256 + (dec_t0_fits_lds(n_layers) ? 0 : (long)n_layers * 512) + 2;
}

int wn_launch_decode(const WnDecodeArgs& a, hipStream_t st) {
    if (a.n_steps <= 0) return 0;
    if (a.n_layers > WN_DEC_MAX_LAYERS) return wn_set_error_msg(-4, "decode: too many layers");
    const int nu = a.n_utt > 0 ? a.n_utt : 1;
    // Matrix-core pair of workgroups (64 / 64 / 256 / 256 channels, with or without biases; the API layer only hands `pk`
    // over at those shapes): eight utterances per pair, a launch with fewer mirrors the last one into the spare columns.
    // The two workgroups of a pair spin on each other's hand-offs, so every pair must be resident at once: 128 pairs =
    // 1024 utterances per launch.  Everything else (other channel counts, fewer than 4 steps) runs on decode_k: one
    // workgroup per utterance, fp32 FMA.
    const bool any_bias = a.b_layers || a.b_causal || a.b_p1 || a.b_p2;
    const bool mf = a.pk && a.pk_skip >= 0 && a.sync && a.n_steps >= 4 && !(a.dbg & 31) && (a.S == 256 || a.S == 512);
    if (nu > (mf ? 1024 : 128)) return wn_set_error_msg(-4, "decode: at most 128 utterances per launch (1024 on the matrix-core path)");
    if (mf) {
        const size_t nsync = (size_t)a.sync_ustride * sizeof(unsigned long long) * (size_t)nu;
        hipError_t e = hipMemsetAsync(a.sync, 0, nsync, st);              // tags start at 1
        if (e != hipSuccess) return wn_set_error(e, __FILE__, __LINE__);
        const size_t s80_f = sizeof(float) * (size_t)(16 * a.Q + 16 * a.R + (a.b_layers ? a.n_layers * (2 * a.D + a.R) : 0));
        const size_t s80 = s80_f + sizeof(uint16_t) * (size_t)(8 * 136 * (3 + (size_t)a.n_layers));
        // tap 0 ahead (decode_duo_mfma8_k<.., 1 | 2>): split queue columns of DEC_T0_CHUNK blocks (+ 4 KB of partial sums per block in LDS: T0 = 1)
        const size_t s80_t2 = sizeof(float) * (size_t)(16 * a.Q + 16 * a.R) + sizeof(uint16_t) * (size_t)(8 * 136 * (3 + DEC_T0_CHUNK));
        const size_t s80_t0 = dec_t0_lds_bytes(a.n_layers);
        const size_t ws_h = 2 * ((size_t)a.S + 64) + 8;               // halfs of one utterance's split S-vector (the kernel's WS)
        const size_t s81 = sizeof(uint16_t) * (size_t)(2 * 8 * 136 + 2 * 8 * ws_h) + sizeof(float) * (size_t)(8 * a.Q + 2 * a.S + a.Q);
        static int t0_env = -1, ks_env = -2;
        if (t0_env < 0) { const char* e = getenv("WN_DEC_T0"); t0_env = e ? atoi(e) : 1; }
        if (ks_env < -1) { const char* e = getenv("WN_DEC_KS"); ks_env = e ? atoi(e) : -1; }
        // tap-0 table: in LDS when it fits (<= 31 blocks), else in the pair's hand-off area (wn_decode_sync_granules leaves room)
        const long need_tab = wn_decode_granules(a.n_layers, a.D, a.S);
        const int t0 = !t0_env ? 0 : (dec_t0_fits_lds(a.n_layers) ? 1 : (a.sync_ustride >= need_tab ? 2 : 0));
        // split skip / post-processing (KS = S / 64 workgroups + the chain per pair): the default while every workgroup of every
        // pair is resident at once (they spin on each other: at most 224 workgroups per launch); WN_DEC_KS=1 forces the
        // one-workgroup form (what larger batches run), any other value the split form wherever it fits.  Measured (round 4): the
        // reference's shipped 40-block 32 / 32 / 512 model 9.4 -> 17.3 k samples/s single stream and 0.96 -> 1.40 M at 128
        // utterances; config 5 (30 blocks, 256 skip channels) 27.0 -> 27.6 k single stream, but 1.59 -> 1.41 M at 64 utterances and
        // 3.14 -> 2.82 M at 128 (one workgroup keeps pace there, the extra pollers only cost): at 256 skip channels the split
        // form is the default for ONE pair (up to eight utterances) only
        const int pairs = (nu + 7) / 8, ks_full = a.S / 64;
        const bool fits = (long)pairs * (1 + ks_full) <= 224 && a.sync_ustride >= need_tab;
        const bool want = ks_env > 1 || (ks_env < 0 && (a.S == 512 || pairs <= 1));
        const int ks = (fits && want) ? ks_full : 1;
        const size_t s_chain = t0 == 1 ? s80_t0 : (t0 == 2 ? s80_t2 : s80);
        const size_t sh = s_chain > s81 ? s_chain : s81;
        if (sh + 1024 > 160 * 1024) return wn_set_error_msg(-4, "decode: this many blocks do not fit the matrix-core kernel's LDS");
        if (getenv("WN_DEC_VERBOSE")) fprintf(stderr, "[wn_decode] matrix-core, 8 utterances per pair: %d utterances, %d steps, %d skip channels, biases %d, tap-0 ahead %d, skip parts %d\n", nu, a.n_steps, a.S, any_bias ? 1 : 0, t0, ks);
        int dev = 0;
        (void)hipGetDevice(&dev);
        const int mx = 160 * 1024 - 1024;           // (the kernel also has ~350 bytes of static LDS)
        const dim3 gr((1 + ks) * pairs), bl(DEC_MT);
#define DEC_GO3(BB, TT, SS, KK) do { \
        static WnDevOnce done_; \
        if (done_.need(dev)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_duo_mfma8_k<BB, TT, SS, KK>), hipFuncAttributeMaxDynamicSharedMemorySize, mx); done_.done(dev); } \
        hipLaunchKernelGGL((decode_duo_mfma8_k<BB, TT, SS, KK>), gr, bl, sh, st, a); } while (0)
#define DEC_GO2(BB, SS, KK) do { if (t0 == 2) DEC_GO3(BB, 2, SS, KK); else if (t0 == 1) DEC_GO3(BB, 1, SS, KK); else DEC_GO3(BB, 0, SS, KK); } while (0)
#define DEC_GO(SS, KK) do { if (any_bias) DEC_GO2(true, SS, KK); else DEC_GO2(false, SS, KK); } while (0)
        if (a.S == 512 && ks > 1) DEC_GO(512, 8);
        else if (a.S == 512) DEC_GO(512, 1);
        else if (ks > 1) DEC_GO(256, 4);
        else DEC_GO(256, 1);
#undef DEC_GO
#undef DEC_GO2
#undef DEC_GO3
    } else {
        if (getenv("WN_DEC_VERBOSE")) fprintf(stderr, "[wn_decode] generic fp32 kernel: %d utterances, %d steps\n", nu, a.n_steps);
        size_t sh = sizeof(float) * (size_t)(3 * a.Q + 3 * a.R + 3 * a.D + 2 * a.S + 64);
        hipLaunchKernelGGL(decode_k, dim3(nu), dim3(DEC_THREADS), sh, st, a);
    }
    WN_CHECK_LAUNCH();
    return 0;
}
