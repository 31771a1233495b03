// Fused forward epilogue of the WaveNet (wavenet/model.py:127-138): for one tile of 128 output columns
//
//   U = bias_s + Ws  Z          (the skip convs of all N blocks + the Python `sum`: ONE product over the stacked z-crops, K = N*CH)
//   H = bias_1 + P1 relu(U)     (post_process_1 on the ReLU of the skip sum)
//   O = bias_2 + P2 relu(H)     (post_process_2; compact [Q][W] pre-softmax in the reference's memory order)
//
// in ONE launch, the U and H tiles handed from product to product ON CHIP.  U and H are still stored (the backward masks
// with them and multiplies them into the weight gradients), but never read back here: the three wide GEMMs this replaces
// each paid a cold start, a half-empty last round of workgroups and a re-read of the previous product.
//
// Tile economy (what differs from chan_gemm_wide2_k): 256 rows x 128 columns per workgroup, a wave owns 2 row tiles x all 8
// column tiles (64 accumulators, not 128), and the WEIGHTS never touch LDS: a wave's 2 row tiles are its own, so it takes
// their packed fragments straight from L2, one iteration ahead (no sharing to organise, no barrier on that path).  Only the
// activations go through LDS: per iteration (two 32-row k-steps) the eight waves fetch 32 KB of z rows two iterations
// ahead into registers, split them once into ready f16 hi/lo B fragments (2 x 32 KB stages), one barrier per iteration.
// The hand-over U -> P1 needs no transposition: with the weights packed in the "chained" k order a wave's accumulators ARE
// the B fragments of k-step (wave index) - ReLU, split, 16 ds_write_b128 per wave - and the 256 x 128 operand (128 KB) takes
// the place of the stages; the P1 / P2 products then read it like the B-stationary product does (wn_gemm_bst.hip): no
// barrier, no LDS write inside them.
#include <stdlib.h>
#include <type_traits>
#include "wn_common.h"
#include "wn_kernels.h"

#define EPI_COLS 128

template <class T>
__global__ __launch_bounds__(512) void skip_epilogue_fwd_k(WnEpiFwdArgs a) {
    constexpr int NS = 3, FR = 1024;                          // halfs per fragment (hi + lo)
    constexpr int STAGE = 16 * FR;                            // halfs per stage: 2 k-steps x 8 column tiles
    extern __shared__ __attribute__((aligned(16))) uint16_t l_s[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: row tiles, roles and row pointers stay in SGPRs)
    const int c = lane & 15, q = lane >> 4;
    const int b = blockIdx.x / a.ntx, tile0 = a.t_base + (blockIdx.x % a.ntx) * EPI_COLS;
    const int KS = a.ks_skip, NI = KS >> 1;
    // Workgroups of one round run in step: every CU streams z (reads only) for the same 70 us, then every CU stores its U, H and O
    // tiles in the same 20 us - a burst of pure writes that HBM takes at its write rate while nothing else moves.  The first round is
    // therefore STAGGERED: workgroup g of the first `stagger_n` waits g / stagger_n of `stagger_cycles` (most of a tile's time) before it
    // starts, later rounds inherit the offsets, and one CU's stores land beside another's reads.  Costs nothing at a fill like 3.19
    // rounds: the CUs that start first take the fourth-round tiles
    if (blockIdx.x < (unsigned)a.stagger_n && a.stagger_cycles > 0) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        const unsigned long long wait = (unsigned long long)a.stagger_cycles * blockIdx.x / a.stagger_n;
        while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
    const int m0 = 2 * wave;                                  // this wave's two row tiles (of 16)

    // loader role: k-step sp of an iteration's pair, column group lg, row half lh
    const int lg = wave & 1, lh = (wave >> 1) & 1, sp = wave >> 2;
    const float* zin = a.z + (size_t)b * a.z_bstride;
    const int col = tile0 + lg * 64 + 4 * c;
    const float* zrow = zin + (size_t)(sp * 32 + 8 * q + 4 * lh) * a.pitch + col;
    // The loop body is ONE basic block (no branch around a load: hipcc cannot schedule across one, and drains the load queue at the
    // join): every lane loads its 4 columns of every row unguarded - they lie inside the row's pitch, which the caller keeps readable -
    // and columns outside [t_lo, t_hi) are zeroed by a select (stale values there may be anything, NaN included); iterations past
    // the end re-request the last one.
    bool keep[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) keep[e] = col + e >= a.t_lo && col + e < a.t_hi;
    auto load_raw = [&](f32x4* raw, int it) {                 // the 4 rows x 4 columns of this lane in iteration it
        const float* p = zrow + (size_t)(it < NI ? it : NI - 1) * 64 * a.pitch;
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)j * a.pitch));
    };
    auto store_raw = [&](const f32x4* raw, int st) {          // split, 8-byte hi / lo pieces of the 4 fragments of group lg
        uint16_t* bb = l_s + (size_t)st * STAGE + (size_t)(sp * 8 + lg * 4) * FR + lane * 8 + lh * 4;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            uint2 hv, lv;
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = keep[n] ? raw[j][n] : 0.f;
            split2<T>(x[0], x[1], hv.x, lv.x);
            split2<T>(x[2], x[3], hv.y, lv.y);
            *reinterpret_cast<uint2*>(bb + (size_t)n * FR) = hv;
            *reinterpret_cast<uint2*>(bb + (size_t)n * FR + 512) = lv;
        }
    };
    // weights of this wave's two row tiles for the two k-steps of iteration it (pack: [16 row tiles][ks] fragments)
    auto load_w2 = [&](Frag<T> (*af)[2], const uint16_t* pack, int ks, int it) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) load_a<T, NS>(af[kk][i], pack, (m0 + i) * ks + 2 * (it < NI ? it : NI - 1) + kk, lane);
    };

    f32x4 acc[2][8], acc2[2][8];
    auto init_acc = [&](f32x4 (*acc)[8], const float* bias, int valid) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (m0 + i) * 16 + 4 * q + r;
                    v[r] = row < valid ? bias[row] : 0.f;
                }
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[i][n] = v;
        }
    };
    // one k-step: 8 column tiles x 2 row tiles.  `between(n)` is issued behind column tile n's reads: the loop hangs its global
    // requests there one at a time - eight waves that request 12 KB each right behind a barrier queue for the CU's one address
    // path (1 KB per 16 cycles) while the matrix pipe waits
    auto mma_half = [&](f32x4 (*acc)[8], const uint16_t* frags, Frag<T>* af, auto between) {
        // B fragments two column tiles ahead of their MFMAs (three register sets: a read's latency is behind 12 MFMAs, not in
        // front of 6)
        Frag<T> bf[3];
        load_a<T, NS>(bf[0], frags, 0, lane);
        load_a<T, NS>(bf[1], frags, 1, lane);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (n + 2 < 8) load_a<T, NS>(bf[(n + 2) % 3], frags, n + 2, lane);
            between(n);
#pragma unroll
            for (int i = 0; i < 2; ++i) mma<T, NS>(acc[i][n], af[i], bf[n % 3]);
            __builtin_amdgcn_sched_barrier(0);                // (hipcc otherwise folds the three sets back into one: read, wait, 6 MFMAs)
        }
    };
    auto nothing = [](int) {};

    // ---------------------------------------------------------------- U = Ws Z over the stage ring
    init_acc(acc, a.bias_s, a.s_valid);
    f32x4 raw0[4], raw1[4];
    Frag<T> af0[2][2], af1[2][2], af2[2][2];
    // NI mod 6 iterations one at a time (request, wait, fill, multiply: nothing overlaps; short products and odd depths only),
    // so that the pipelined loop below runs whole groups of six
    const int it0 = NI % 6;
    for (int it = 0; it < it0; ++it) {
        load_raw(raw0, it);
        load_w2(af0, a.w_skip, KS, it);
        store_raw(raw0, 0);
        __syncthreads();
        mma_half(acc, l_s, af0[0], nothing);
        mma_half(acc, l_s + 8 * FR, af0[1], nothing);
        __syncthreads();
    }
    // iteration it: MFMAs of its two k-steps out of stage it & 1; between them the rows of iteration it + 1 (requested two
    // iterations ago) are split into the other stage and their registers re-armed two iterations ahead; the weights of
    // iteration it + 2 are requested at its top (three register sets).  The loop body is six iterations of straight-line code and
    // the prologue issues its requests in the loop's own order (weights, rows, weights, rows): hipcc's wait counts are then exact
    // (24 / 20 requests may stay in flight); any other arrangement made it wait for requests it had only just issued
    auto iter = [&](int it, f32x4* rnext, Frag<T> (*afc)[2], Frag<T> (*afn)[2]) {
        const uint16_t* st = l_s + (size_t)(it & 1) * STAGE;
        // (the scheduling fences pin the requests where they are written: left alone, hipcc sinks both kinds of load to the end of
        // the iteration and hoists the selects of the rows to its top)
        const int itw = it + 2 < NI ? it + 2 : NI - 1, itr = it + 3 < NI ? it + 3 : NI - 1;
        mma_half(acc, st, afc[0], [&](int n) {                // the 8 weight requests of iteration it + 2, one per column tile
            const int kk = n >> 2, i = (n >> 1) & 1;
            const u32x4* p = reinterpret_cast<const u32x4*>(a.w_skip) + (size_t)((m0 + i) * KS + 2 * itw + kk) * 128 + lane + 64 * (n & 1);
            if (n & 1) afn[kk][i].lo = __builtin_bit_cast(typename T::vec8, *p);
            else afn[kk][i].hi = __builtin_bit_cast(typename T::vec8, *p);
        });
        __builtin_amdgcn_sched_barrier(0);
        store_raw(rnext, (it + 1) & 1);                       // (the last iteration fills a stage nobody reads)
        __builtin_amdgcn_sched_barrier(0);
        mma_half(acc, st + 8 * FR, afc[1], [&](int n) {       // the 4 row requests of iteration it + 3
            if (n < 4) rnext[n] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(zrow + (size_t)itr * 64 * a.pitch + (size_t)n * a.pitch));
        });
        __syncthreads();
    };
    if (it0 < NI) {
        load_raw(raw0, it0);
        load_w2(af0, a.w_skip, KS, it0);
        load_raw(raw1, it0 + 1);
        load_w2(af1, a.w_skip, KS, it0 + 1);
        store_raw(raw0, it0 & 1);
        load_raw(raw0, it0 + 2);
        __syncthreads();
        for (int it = it0; it < NI; it += 6) {
            iter(it, raw1, af0, af2);
            iter(it + 1, raw0, af1, af0);
            iter(it + 2, raw1, af2, af1);
            iter(it + 3, raw0, af0, af2);
            iter(it + 4, raw1, af1, af0);
            iter(it + 5, raw0, af2, af1);
        }
    }

    // ---------------------------------------------------------------- store a tile, hand it on as the next product's B operand
    const bool tile_in = tile0 >= a.t_lo && tile0 + EPI_COLS <= a.t_hi;
    // row-stores e0 .. e1 - 1 of a tile's 16 per lane (e = 8 i + 2 r + g: row tile i, register r, column group g)
    auto store_rows = [&](f32x4 (*acc)[8], int e0, int e1, float* base, long bstride, int pitch, int shift, int valid) {
        float* out = base + (size_t)b * bstride + (size_t)(m0 * 16 + 4 * q) * pitch + tile0 + 4 * c + shift;
        if (tile_in && (m0 + 2) * 16 <= valid) {              // wave-uniform: an interior tile, both row tiles real
#pragma unroll
            for (int e = e0; e < e1; ++e) {
                const int i = e >> 3, r = (e >> 1) & 3, g = e & 1;
                F4U u = {{acc[i][4 * g][r], acc[i][4 * g + 1][r], acc[i][4 * g + 2][r], acc[i][4 * g + 3][r]}};
                *reinterpret_cast<F4U*>(out + (size_t)(16 * i + r) * pitch + 64 * g) = u;
            }
            return;
        }
#pragma unroll
        for (int e = e0; e < e1; ++e) {
            const int i = e >> 3, r = (e >> 1) & 3, g = e & 1;
            if ((m0 + i) * 16 + 4 * q + r >= valid) continue;
            const int tl = tile0 + 64 * g + 4 * c;
            float* op = out + (size_t)(16 * i + r) * pitch + 64 * g;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tl + k >= a.t_lo && tl + k < a.t_hi) op[k] = acc[i][4 * g + k][r];
        }
    };
    // relu(acc) as the B fragments of k-step `wave` in the chained k order (k = 32 s + 16 (j >> 2) + 4 q + (j & 3)): element j of
    // lane (c, q) in column tile n is row tile j >> 2, register j & 3 of this lane's own accumulators
    auto hand_over = [&](f32x4 (*acc)[8]) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(acc[j >> 2][n][j & 3], 0.f);
            Frag<T> f;
            split8<T, NS>(f, v);
            u32x4* p = reinterpret_cast<u32x4*>(l_s) + (size_t)(wave * 8 + n) * 128 + lane;
            p[0] = __builtin_bit_cast(u32x4, f.hi);
            p[64] = __builtin_bit_cast(u32x4, f.lo);
        }
    };
    // H = bias_1 + P1 relu(U), O = bias_2 + P2 relu(H): 8 k-steps each over the 128 KB operand.  ONE ring of two weight slots runs
    // through both products (chained packs [16][8]): a slot is re-armed two k-steps ahead once its MFMAs are issued (three slots: 9 registers spilled).  U stays in its
    // accumulators while H is formed in the second set (and H while O is formed in the first), and a tile's 16 row-stores per lane
    // go out two per k-step BEHIND that k-step's weight request: vector memory completes in order, so a request waits for every
    // store in front of it - spread like this a store has two k-steps to drain before anything waits for it
    Frag<T> wf[2][2];
    auto request = [&](Frag<T>* slot, const uint16_t* pack, int s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) load_a<T, NS>(slot[i], pack, (m0 + i) * 8 + s, lane);
    };
#pragma unroll
    for (int s = 0; s < 2; ++s) request(wf[s], a.w_p1c, s);
    hand_over(acc);                                           // (the loop's last barrier: every wave is done with the stages)
    __syncthreads();
    init_acc(acc2, a.bias_1, a.s_valid);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        mma_half(acc2, l_s + (size_t)s * 8 * FR, wf[s & 1], nothing);
        if (s + 2 < 8) request(wf[s & 1], a.w_p1c, s + 2);
        else request(wf[s & 1], a.w_p2c, s + 2 - 8);
        store_rows(acc, 2 * s, 2 * s + 2, a.u, a.s_bstride, a.pitch, 0, a.s_valid);
    }
    __syncthreads();                                          // every wave has read relu(U)
    hand_over(acc2);
    __syncthreads();
    init_acc(acc, a.bias_2, a.q_valid);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        mma_half(acc, l_s + (size_t)s * 8 * FR, wf[s & 1], nothing);
        if (s + 2 < 8) request(wf[s & 1], a.w_p2c, s + 2);
        store_rows(acc2, 2 * s, 2 * s + 2, a.h, a.s_bstride, a.pitch, 0, a.s_valid);
    }
    store_rows(acc, 0, 16, a.o, a.o_bstride, a.o_pitch, -a.t_lo, a.q_valid);
}

// ---------------------------------------------------------------------------------------------
// Backward of the same epilogue, data gradients (autograd of wavenet/model.py:127-138), one launch per 128-column tile:
//
//   dH = (P2^T dO) * [H > 0] ;  dU = (P1^T dH) * [U > 0] ;  dZ = Ws^T dU          (dH, dU stored: the weight gradients read them)
//
// The compact dO tile (256 rows x 128 columns) is split ONCE into the 128 KB B operand; dH and dU are handed from product to
// product out of the accumulators like U and H in the forward (chained packs of P1^T and Ws^T); the dZ product - 30 x 64 rows
// against K = 256, 795 MB of output - walks its row tiles over the resident dU like chan_gemm_bst_k: a wave owns 3 row tiles x 8
// column tiles per pass, weights straight from L2 one k-step ahead, streaming stores.  Replaces three wn_chan_gemm launches that
// each paid a cold start and a half-empty last round; dH / dU are never read back here.
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(512) void skip_epilogue_bwd_k(WnEpiBwdArgs a) {
    constexpr int NS = 3, FR = 1024;
    extern __shared__ __attribute__((aligned(16))) uint16_t l_s[];      // [8 k-steps][8 column tiles] B fragments
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;      // (as a scalar - readfirstlane - the forward launch gains 1.5 %, this one nothing)
    const int c = lane & 15, q = lane >> 4;
    // whole tiles first; the tiles of a last, partly filled round are dealt out by dZ PASSES (one workgroup per pass re-forms dH and dU -
    // a quarter of a tile's work - and walks one pass of row tiles; only pass 0 stores dH / dU): 816 tiles on 256 CUs are then 3 rounds
    // + 240 short workgroups instead of 4 rounds
    const int npass = (a.mt_z / 3 + 7) / 8;
    int tile = blockIdx.x, p0 = 0, p1 = npass;
    if ((int)blockIdx.x >= a.n_whole) {
        const int u = blockIdx.x - a.n_whole;
        tile = a.n_whole + u / npass;
        p0 = u % npass;
        p1 = p0 + 1;
    }
    const bool store_sd = p0 == 0;
    const int b = tile / a.ntx, tile0 = a.t_base + (tile % a.ntx) * EPI_COLS;
    const int m0 = 2 * wave;
    const bool tile_in = tile0 >= a.t_lo && tile0 + EPI_COLS <= a.t_hi;
    const unsigned lane_off = (unsigned)(4 * q) * (unsigned)a.pitch + 4u * c;      // this lane inside a 16-row x 64-column piece

    // ---- fill: dO rows (k-steps sp, sp + 2, ..), column group lg, row half lh; compact layout: column t - t_lo, 4-byte aligned rows
    {
        const int lg = wave & 1, lh = (wave >> 1) & 1, sp = wave >> 2;
        const float* in = a.d_o + (size_t)b * a.o_bstride;
        const int tg0 = tile0 + lg * 64;
        const bool inner = tg0 >= a.t_lo && tg0 + 64 <= a.t_hi;
        const int col = tg0 + 4 * c;
        f32x4 raw[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* p = in + (size_t)((sp + 2 * i) * 32 + 8 * q + 4 * lh) * a.o_pitch + (col - a.t_lo);
            if (inner) {
#pragma unroll
                for (int j = 0; j < 4; ++j) raw[i][j] = ld4u(p + (size_t)j * a.o_pitch);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) raw[i][j] = ld4g(p + (size_t)j * a.o_pitch, col, a.t_lo, a.t_hi);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint16_t* bb = l_s + (size_t)((sp + 2 * i) * 8 + lg * 4) * FR + lane * 8 + lh * 4;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                uint2 hv, lv;
                split2<T>(raw[i][0][n], raw[i][1][n], hv.x, lv.x);
                split2<T>(raw[i][2][n], raw[i][3][n], hv.y, lv.y);
                *reinterpret_cast<uint2*>(bb + (size_t)n * FR) = hv;
                *reinterpret_cast<uint2*>(bb + (size_t)n * FR + 512) = lv;
            }
        }
    }

    auto mma_step = [&](auto& acc, const uint16_t* frags, Frag<T>* af, auto mt) {      // one k-step: 8 column tiles x MT row tiles
        constexpr int MT = decltype(mt)::value;
        Frag<T> bf[3];
        load_a<T, NS>(bf[0], frags, 0, lane);
        load_a<T, NS>(bf[1], frags, 1, lane);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (n + 2 < 8) load_a<T, NS>(bf[(n + 2) % 3], frags, n + 2, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i) mma<T, NS>(acc[i][n], af[i], bf[n % 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // masks of this lane's 2 x 8 accumulator tiles: mask[row][col] > 0 for e = 8 i + 2 r + g (row tile i, register r, group g).  They are
    // requested BEHIND k-step 5 of the product they gate - the weight ring is draining by then, so their 64 registers are free - and land
    // during its last two k-steps (requested in front of the product they lay beside two accumulator sets' worth of live values: 22
    // registers spilled; packed into sign bits as they landed: 31)
    auto load_mask = [&](f32x4* mk, const float* base) {
        // (addresses as a wave-uniform row pointer + ONE 32-bit lane offset)
        const float* mp = base + (size_t)b * a.s_bstride + (size_t)(m0 * 16) * a.pitch + tile0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = e >> 3, r = (e >> 1) & 3, g = e & 1;
            // (unguarded: the mask rows are workspace rows with the activation layout's slack; values outside the tile's valid
            // columns only gate results that are never stored)
            mk[e] = ld4u(mp + (size_t)(16 * i + r) * a.pitch + 64 * g + lane_off);
        }
    };
    auto apply_mask = [&](f32x4 (*acc)[8], const f32x4* mk) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = e >> 3, r = (e >> 1) & 3, g = e & 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[i][4 * g + k][r] = mk[e][k] > 0.f ? acc[i][4 * g + k][r] : 0.f;
        }
    };
    auto store_rows = [&](f32x4 (*acc)[8], float* base, int valid) {
        float* out = base + (size_t)b * a.s_bstride + (size_t)(m0 * 16) * a.pitch + tile0;
        if (tile_in && (m0 + 2) * 16 <= valid) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = e >> 3, r = (e >> 1) & 3, g = e & 1;
                F4U u = {{acc[i][4 * g][r], acc[i][4 * g + 1][r], acc[i][4 * g + 2][r], acc[i][4 * g + 3][r]}};
                *reinterpret_cast<F4U*>(out + (size_t)(16 * i + r) * a.pitch + 64 * g + lane_off) = u;
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = e >> 3, r = (e >> 1) & 3, g = e & 1;
            if ((m0 + i) * 16 + 4 * q + r >= valid) continue;
            const int tl = tile0 + 64 * g + 4 * c;
            float* op = out + (size_t)(16 * i + r) * a.pitch + 64 * g + lane_off;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tl + k >= a.t_lo && tl + k < a.t_hi) op[k] = acc[i][4 * g + k][r];
        }
    };
    auto hand_over = [&](f32x4 (*acc)[8]) {                   // acc as the B fragments of k-step `wave`, chained k order
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = acc[j >> 2][n][j & 3];
            Frag<T> f;
            split8<T, NS>(f, v);
            u32x4* p = reinterpret_cast<u32x4*>(l_s) + (size_t)(wave * 8 + n) * 128 + lane;
            p[0] = __builtin_bit_cast(u32x4, f.hi);
            p[64] = __builtin_bit_cast(u32x4, f.lo);
        }
    };
    // acc = W x (the operand in LDS), 16 row tiles in all (this wave: 2), weights three k-steps ahead through a 3-slot ring; `late` runs
    // behind k-step 5 (the masks are requested there)
    auto product16 = [&](f32x4 (*acc)[8], const uint16_t* pack, auto late) {
        Frag<T> wf[3][2];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) load_a<T, NS>(wf[s][i], pack, (m0 + i) * 8 + s, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            mma_step(acc, l_s + (size_t)s * 8 * FR, wf[s % 3], std::integral_constant<int, 2>());
            if (s + 3 < 8) {
#pragma unroll
                for (int i = 0; i < 2; ++i) load_a<T, NS>(wf[s % 3][i], pack, (m0 + i) * 8 + s + 3, lane);
            }
            if (s == 5) late();
        }
    };

    f32x4 acc[2][8], mk[16];
    __syncthreads();                                          // the dO fragments are in place
    product16(acc, a.w_p2T, [&]() { load_mask(mk, a.h); });
    apply_mask(acc, mk);
    if (store_sd) store_rows(acc, a.d_h, a.s_valid);
    __syncthreads();                                          // every wave has read dO
    hand_over(acc);
    __syncthreads();
    product16(acc, a.w_p1Tc, [&]() { load_mask(mk, a.u); });
    apply_mask(acc, mk);
    if (store_sd) store_rows(acc, a.d_u, a.s_valid);
    __syncthreads();                                          // every wave has read dH
    hand_over(acc);
    __syncthreads();

    // ---- dZ = Ws^T dU: passes of 3 row tiles per wave over the resident operand (wn_gemm_bst.hip), streaming stores
    {
        constexpr int MT = 3;
        float* out = a.d_z + (size_t)b * a.z_bstride;
        Frag<T> af[2][MT];
        int mz = (p0 * 8 + wave) * MT;
        auto load_w = [&](Frag<T>* f, int m, int s) {
#pragma unroll
            for (int i = 0; i < MT; ++i) load_a<T, NS>(f[i], a.w_skipTc, (m + i) * 8 + s, lane);
        };
        if (mz < a.mt_z) load_w(af[0], mz, 0);
        for (int p = p0; p < p1; ++p, mz += 8 * MT) {
            if (mz >= a.mt_z) break;
            f32x4 az[MT][8];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int n = 0; n < 8; ++n) az[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s + 1 < 8) load_w(af[(s + 1) & 1], mz, s + 1);
                mma_step(az, l_s + (size_t)s * 8 * FR, af[s & 1], std::integral_constant<int, MT>());
            }
            const int mn = mz + 8 * MT;
            load_w(af[0], mn < a.mt_z ? mn : mz, 0);          // the next pass's first weights in front of this pass's stores
            if (tile_in && (mz + MT) * 16 <= a.z_valid) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* op = out + (size_t)((mz + i) * 16 + r) * a.pitch + tile0;
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const f32x4 v = {az[i][4 * g][r], az[i][4 * g + 1][r], az[i][4 * g + 2][r], az[i][4 * g + 3][r]};
                            if (a.nt_dz) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(op + 64 * g + lane_off));
                            else *reinterpret_cast<f32x4*>(op + 64 * g + lane_off) = v;
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (mz + i) * 16 + 4 * q + r;
                    if (row >= a.z_valid) continue;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int tl = tile0 + 64 * g + 4 * c;
                        float* op = out + (size_t)row * a.pitch + tl;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (tl + k >= a.t_lo && tl + k < a.t_hi) op[k] = az[i][4 * g + k][r];
                    }
                }
            }
        }
    }
}

int wn_launch_skip_epilogue_bwd(const WnEpiBwdArgs& a0, int batch, int mode, hipStream_t st) {
    if (a0.t_hi <= a0.t_lo || batch <= 0) return 0;
    if (mode != WN_MODE_F16X3 && mode != WN_MODE_BF16X3) return wn_set_error_msg(-2, "wn_skip_epilogue_bwd: x3 modes only");
    if (a0.mt_z <= 0 || a0.mt_z % 3 != 0) return wn_set_error_msg(-4, "wn_skip_epilogue_bwd: the z rows in groups of 48");
    WnEpiBwdArgs a = a0;
    a.t_base = wn_tile_origin(a.t_lo);
    a.ntx = (a.t_hi - a.t_base + EPI_COLS - 1) / EPI_COLS;
    if (a.t_base + a.ntx * EPI_COLS > a.pitch)
        return wn_set_error_msg(-4, "wn_skip_epilogue_bwd: the 128-column tiles over [t_lo & ~63, t_hi) must lie inside the row pitch (mask rows are read over whole tiles)");
    const size_t sh = (size_t)64 * 1024 * sizeof(uint16_t);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static WnDevOnce done;
    if (done.need(dev)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skip_epilogue_bwd_k<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skip_epilogue_bwd_k<BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        done.done(dev);
    }
    const int cus = wn_num_cus();
    const int ntiles = a.ntx * batch, npass = (a.mt_z / 3 + 7) / 8, rest = ntiles % cus;
    // dZ by plain stores: streaming stores are 12 % faster for this product ALONE (wn_gemm_bst.hip) but inside the step nothing
    // (4.176 against 4.192 ms; the stack's first blocks read the rows written last out of the caches); WN_EPI_BWD_NT=1 = streaming
    { const char* en = getenv("WN_EPI_BWD_NT"); a.nt_dz = en ? atoi(en) : 0; }
    const char* es = getenv("WN_EPI_BWD_SPLIT");              // 0: every tile whole
    const bool split = ntiles > cus && rest > 0 && rest * npass <= cus && !(es && es[0] == '0');
    a.n_whole = split ? ntiles - rest : ntiles;
    const dim3 g(a.n_whole + (split ? rest * npass : 0)), bl(512);
    if (mode == WN_MODE_F16X3) hipLaunchKernelGGL(skip_epilogue_bwd_k<F16>, g, bl, sh, st, a);
    else hipLaunchKernelGGL(skip_epilogue_bwd_k<BF16>, g, bl, sh, st, a);
    WN_CHECK_LAUNCH();
    return 0;
}

int wn_launch_skip_epilogue_fwd(const WnEpiFwdArgs& a0, int batch, int mode, hipStream_t st) {
    if (a0.t_hi <= a0.t_lo || batch <= 0) return 0;
    if (mode != WN_MODE_F16X3 && mode != WN_MODE_BF16X3)
        return wn_set_error_msg(-2, "wn_skip_epilogue_fwd: x3 modes only");
    if (a0.ks_skip < 2 || (a0.ks_skip & 1)) return wn_set_error_msg(-4, "wn_skip_epilogue_fwd: an even number of 32-row k-steps");
    WnEpiFwdArgs a = a0;
    a.t_base = wn_tile_origin(a.t_lo);
    a.ntx = (a.t_hi - a.t_base + EPI_COLS - 1) / EPI_COLS;
    if (a.t_base + a.ntx * EPI_COLS > a.pitch)
        return wn_set_error_msg(-4, "wn_skip_epilogue_fwd: the 128-column tiles over [t_lo & ~63, t_hi) must lie inside the row pitch (rows are read over whole tiles)");
    const size_t sh = (size_t)64 * 1024 * sizeof(uint16_t);  // 128 KB: the hand-over operand (the two 32 KB stages lie inside it)
    int dev = 0;
    (void)hipGetDevice(&dev);
    static WnDevOnce done;
    if (done.need(dev)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skip_epilogue_fwd_k<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skip_epilogue_fwd_k<BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        done.done(dev);
    }
    // stagger of the first round (kernel comment): on when the launch has more than one round; WN_EPI_STAGGER = cycles per k-step pair
    // (0 = off; default 3600, i.e. about 0.75 of a tile's loop time)
    const int cus = wn_num_cus();
    const char* es = getenv("WN_EPI_STAGGER");
    const int per_it = es ? atoi(es) : 3600;
    a.stagger_n = a.ntx * batch > cus ? cus : 0;
    a.stagger_cycles = per_it * (a.ks_skip / 2);
    const dim3 g(a.ntx * batch), bl(512);
    if (mode == WN_MODE_F16X3) hipLaunchKernelGGL(skip_epilogue_fwd_k<F16>, g, bl, sh, st, a);
    else hipLaunchKernelGGL(skip_epilogue_fwd_k<BF16>, g, bl, sh, st, a);
    WN_CHECK_LAUNCH();
    return 0;
}
