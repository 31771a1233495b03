// Weight gradient of the causal layer (wavenet/model.py:104, nn.Conv1d(Q -> R, k = 2)) when its input is a ONE-HOT
// tensor the caller built from integer codes (faster_audio_data.py:62-83 / fast_generate.py:159-160):
//
//     dW[r][q][tap] = sum_{b,t} dx0[b][r][t] * in[b][q][t - 1 + tap],   t in [1, T)
//
// `in` holds T ones per clip, so the sum is a scatter of columns of dx0 - ~33 MB of dx0 read once instead of the
// 131 MB dense one-hot streamed through the generic weight-gradient product.  (The dense path stays for arbitrary
// float inputs: the module's forward takes any tensor)
//
// Both layouts of the reference are handled from the codes alone:
//   proper    in[code[s]][s] = 1                                          (fast_generate.py:159-160)
//   scrambled the (T, Q) one-hot REINTERPRETED as (Q, T) (faster_audio_data.py:77-81): the one of sample s
//             sits at flat index s*Q + code[s] = row (flat / T), column (flat % T).  Row q holds the flat range
//             [qT, (q+1)T), i.e. a run of ~T/Q consecutive samples, so the ones of row q inside a tile of <= 256
//             columns come from at most TWO samples - they are found without any search.
//
// Deterministic by construction (no float atomics between waves): a workgroup walks its tiles of 64 input columns in a
// fixed order; per tile the ones are listed in LDS in a fixed slot order; the one at row q is owned by wave q % 8, which
// adds the two dx0 columns it selects (tap 1: t = column, tap 0: t = column + 1) into the workgroup's LDS accumulator
// [2Q][R] in slot order; each workgroup leaves one slab, summed in slab order by reduce_slabs_k.
#include "wn_common.h"
#include "wn_kernels.h"

#define CW_THREADS 512
#define CW_TC 64              // input columns per tile
#define CW_Q 256
#define CW_ACC_LD 65          // accumulator [2Q][R <= 64] with an odd leading dimension (transposed read-out conflict free)

template <int CH>
__global__ __launch_bounds__(CW_THREADS) void causal_wgrad_codes_k(const int32_t* __restrict__ codes, int scrambled,
                                                                   const float* __restrict__ dx, const float* __restrict__ dxq,
                                                                   int dn, int p_lo, long dx_bstride, int pitch,
                                                                   int T, int batch, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) float cw_lds[];
    float* acc = cw_lds;                                   // [2Q][CW_ACC_LD]
    float* dxs = acc + 2 * CW_Q * CW_ACC_LD;               // [CH][CW_TC + 1]: dx0 columns t0 .. t0 + TC of the tile's clip
    int* list = reinterpret_cast<int*>(dxs + CH * (CW_TC + 1));      // [2Q] entries (q << 8 | j) or -1
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * CW_Q * CW_ACC_LD; i += CW_THREADS) acc[i] = 0.f;
    const int tiles_per_clip = (T + CW_TC - 1) / CW_TC;
    const int n_tiles = tiles_per_clip * batch;
    // dx0 columns t0 .. t0 + TC of a tile, (CH * (TC + 1) + 511) / 512 values per thread; the layer's output exists on
    // [1, T): everything else counts as 0.  The NEXT tile's values are fetched while this one is accumulated.
    constexpr int NV = (CH * (CW_TC + 1) + CW_THREADS - 1) / CW_THREADS;
    float nxt[NV];
    auto fetch = [&](int tile) {
        const bool live = tile < n_tiles;
        const int tl = live ? tile : 0;
        const int b = tl / tiles_per_clip, t0 = (tl - b * tiles_per_clip) * CW_TC;
        const float* src = dx + (size_t)b * dx_bstride;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + k * CW_THREADS;
            const int r = i / (CW_TC + 1), j = i - r * (CW_TC + 1), t = t0 + j;
            const bool ok = live && i < CH * (CW_TC + 1) && t >= 1 && t < T;
            if (dxq == nullptr) {
                nxt[k] = ok ? src[(size_t)r * pitch + t] : 0.f;
            } else {
                // the first block handed its data gradient on as the unshifted pair: dx0[t] = P[t] (t >= p_lo) + Q[t + dn]
                const float pv = (ok && t >= p_lo) ? src[(size_t)r * pitch + t] : 0.f;
                const float qv = (ok && t + dn < T) ? dxq[(size_t)b * dx_bstride + (size_t)r * pitch + t + dn] : 0.f;
                nxt[k] = pv + qv;
            }
        }
    };
    // the ones of a tile, one list slot per possible position; also computed one tile ahead (the code look-ups are
    // dependent global loads)
    int ne0 = -1, ne1 = -1;
    auto entries = [&](int tile) {
        ne0 = ne1 = -1;
        if (tile >= n_tiles) return;
        const int b = tile / tiles_per_clip, t0 = (tile - b * tiles_per_clip) * CW_TC;
        const int32_t* cb = codes + (size_t)b * T;
        if (scrambled) {
            if (tid < CW_Q) {
                const long rowbase = (long)tid * T;         // flat index of (row q, column 0)
                const long F = rowbase + t0;
                long hi = F + CW_TC;
                if (hi > rowbase + T) hi = rowbase + T;     // columns beyond T do not exist
                if (hi > F) {
                    const long s0 = F >> 8, s1 = (hi - 1) >> 8;            // the only samples whose flat index can fall in [F, hi)
                    const int c0 = cb[s0], c1 = cb[s1];
                    const long f0 = (s0 << 8) + c0, f1 = (s1 << 8) + c1;
                    if (c0 >= 0 && c0 < CW_Q && f0 >= F && f0 < hi) ne0 = (tid << 8) | (int)(f0 - F);
                    if (s1 != s0 && c1 >= 0 && c1 < CW_Q && f1 >= F && f1 < hi) ne1 = (tid << 8) | (int)(f1 - F);
                }
            }
        } else if (tid < CW_TC && t0 + tid < T) {
            const int c = cb[t0 + tid];
            if (c >= 0 && c < CW_Q) ne0 = (c << 8) | tid;
        }
    };
    fetch(blockIdx.x);
    entries(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                   // the previous tile's readers are done (and acc is zeroed)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + k * CW_THREADS;
            if (i < CH * (CW_TC + 1)) dxs[i] = nxt[k];
        }
        if (scrambled) {
            if (tid < CW_Q) { list[2 * tid] = ne0; list[2 * tid + 1] = ne1; }
        } else {
            list[tid] = ne0;                                // slots TC .. 2Q-1 stay -1
        }
        fetch(tile + gridDim.x);
        entries(tile + gridDim.x);
        __syncthreads();
        // wave w owns the rows q with q % 8 == w; slot order = summation order
        for (int base = 0; base < 2 * CW_Q; base += 64) {
            const int e = list[base + lane];
            unsigned long long m = __ballot(e >= 0 && ((e >> 8) & 7) == wv);
            while (m) {
                const int src_lane = __builtin_ctzll(m);
                m &= m - 1;
                const int ee = __builtin_amdgcn_readlane(e, src_lane);
                const int q = ee >> 8, j = ee & 255;
                if (lane < CH) {
                    const float v1 = dxs[lane * (CW_TC + 1) + j];          // tap 1: output t = input column
                    const float v0 = dxs[lane * (CW_TC + 1) + j + 1];      // tap 0: output t = input column + 1
                    // plain read-modify-write: this wave is the only one that touches row q's accumulators and the LDS
                    // serves one wave's accesses in order (ds_add_f32 measured 3.6x slower: 226 vs 62 us per launch)
                    acc[(CW_Q + q) * CW_ACC_LD + lane] += v1;
                    acc[q * CW_ACC_LD + lane] += v0;
                }
            }
        }
    }
    __syncthreads();
    // slab [CH][2Q] (leading dimension 2Q: columns = tap 0 rows q | tap 1 rows q), as the generic product writes it
    float* out = slab + (size_t)blockIdx.x * CH * 2 * CW_Q;
    for (int r = 0; r < CH; ++r) out[(size_t)r * 2 * CW_Q + tid] = acc[tid * CW_ACC_LD + r];
}

int wn_causal_codes_slabs(int T, int batch) {
    const int tiles = ((T + CW_TC - 1) / CW_TC) * batch;
    if (tiles <= 0) return 0;
    return tiles < 256 ? tiles : 256;
}

int wn_launch_causal_wgrad_codes(const int32_t* codes, int scrambled, const float* dx, const float* dxq, int dn, int p_lo,
                                 long dx_bstride, int pitch, int ch, int T, int batch, float* slab, hipStream_t st) {
    const int nwg = wn_causal_codes_slabs(T, batch);
    if (nwg <= 0) return 0;
    if (ch != 32 && ch != 64) return wn_set_error_msg(-3, "causal_wgrad_codes: padded channel count must be 32 or 64");
    static_assert(CW_THREADS == 2 * CW_Q, "one thread per slab column in the read-out");
    const size_t sh = sizeof(float) * ((size_t)2 * CW_Q * CW_ACC_LD + (size_t)ch * (CW_TC + 1)) + sizeof(int) * 2 * CW_Q;
    const size_t sh_max = sizeof(float) * ((size_t)2 * CW_Q * CW_ACC_LD + (size_t)64 * (CW_TC + 1)) + sizeof(int) * 2 * CW_Q;
    static WnDevOnce done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done.need(dev)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&causal_wgrad_codes_k<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_max);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&causal_wgrad_codes_k<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_max);
        done.done(dev);
    }
    if (ch == 32) hipLaunchKernelGGL(causal_wgrad_codes_k<32>, dim3(nwg), dim3(CW_THREADS), sh, st, codes, scrambled, dx, dxq, dn, p_lo, dx_bstride, pitch, T, batch, slab);
    else hipLaunchKernelGGL(causal_wgrad_codes_k<64>, dim3(nwg), dim3(CW_THREADS), sh, st, codes, scrambled, dx, dxq, dn, p_lo, dx_bstride, pitch, T, batch, slab);
    WN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward of the causal layer from the same codes:  x0[r][t] = bias[r] + sum_{ones (q, t-1)} W[r][q][0] + sum_{ones (q, t)}
// W[r][q][1],  t in [1, T) - a GATHER of weight columns instead of a (R x 2Q) x (2Q x T) product over a 131 MB tensor of
// zeros and ones (which then need not exist at all).  wt = the weight as [tap][q][ch] (ch contiguous, padded to CH).
// One workgroup per tile of 64 output columns; the ones of input columns t0-1 .. t0+63 are listed as in the backward;
// output column j of the tile is owned by wave j % 8, which adds its contributions in slot order, tap 1 before tap 0 of
// the same slot: deterministic.  Sums of exact fp32 weights: closer to the reference's fp32 conv than the split product.
// ---------------------------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(CW_THREADS) void causal_fwd_codes_k(const int32_t* __restrict__ codes, int scrambled,
                                                                 const float* __restrict__ wt, const float* __restrict__ bias,
                                                                 int n_rows, float* __restrict__ x0, long x_bstride, int pitch, int T) {
    __shared__ float tile[CH * (CW_TC + 1)];
    __shared__ int list[2 * CW_Q];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, t0 = blockIdx.x * CW_TC;              // output columns t0 .. t0 + 63
    const int32_t* cb = codes + (size_t)b * T;
    const int c_lo = t0 > 0 ? t0 - 1 : 0;                           // first input column of interest
    int c_hi = t0 + CW_TC;                                          // one past the last
    if (c_hi > T) c_hi = T;
    for (int i = tid; i < CH * (CW_TC + 1); i += CW_THREADS) {
        const int r = i / (CW_TC + 1);
        tile[i] = (bias != nullptr && r < n_rows) ? bias[r] : 0.f;
    }
    // entries (q << 8 | j), j = input column - (t0 - 1) in [0, 64]
    if (scrambled) {
        if (tid < CW_Q) {
            const long rowbase = (long)tid * T;
            const long F = rowbase + c_lo, hi = rowbase + c_hi;
            int e0 = -1, e1 = -1;
            if (hi > F) {
                const long s0 = F >> 8, s1 = (hi - 1) >> 8;
                const int k0 = cb[s0], k1 = cb[s1];
                const long f0 = (s0 << 8) + k0, f1 = (s1 << 8) + k1;
                if (k0 >= 0 && k0 < CW_Q && f0 >= F && f0 < hi) e0 = (tid << 8) | (int)(f0 - rowbase - (t0 - 1));
                if (s1 != s0 && k1 >= 0 && k1 < CW_Q && f1 >= F && f1 < hi) e1 = (tid << 8) | (int)(f1 - rowbase - (t0 - 1));
            }
            list[2 * tid] = e0;
            list[2 * tid + 1] = e1;
        }
    } else {
        int e = -1;
        const int col = t0 - 1 + tid;
        if (tid <= CW_TC && col >= 0 && col < T) {
            const int k = cb[col];
            if (k >= 0 && k < CW_Q) e = (k << 8) | tid;
        }
        list[tid] = e;
    }
    __syncthreads();
    for (int base = 0; base < 2 * CW_Q; base += 64) {
        const int e = list[base + lane];
        const int j = e & 255;
        // tap 1 lands on tile column j - 1, tap 0 on tile column j
        unsigned long long m = __ballot(e >= 0 && ((j >= 1 && ((j - 1) & 7) == wv) || (j < CW_TC && (j & 7) == wv)));
        while (m) {
            const int src_lane = __builtin_ctzll(m);
            m &= m - 1;
            const int ee = __builtin_amdgcn_readlane(e, src_lane);
            const int q = ee >> 8, jj = ee & 255;
            if (lane < CH) {
                if (jj >= 1 && ((jj - 1) & 7) == wv) tile[lane * (CW_TC + 1) + jj - 1] += wt[(size_t)(CW_Q + q) * CH + lane];
                if (jj < CW_TC && (jj & 7) == wv) tile[lane * (CW_TC + 1) + jj] += wt[(size_t)q * CH + lane];
            }
        }
    }
    __syncthreads();
    float* out = x0 + (size_t)b * x_bstride;
    for (int i = tid; i < CH * CW_TC; i += CW_THREADS) {
        const int r = i >> 6, jo = i & 63, t = t0 + jo;
        if (t >= 1 && t < T && r < n_rows) out[(size_t)r * pitch + t] = tile[r * (CW_TC + 1) + jo];
    }
}

int wn_launch_causal_fwd_codes(const int32_t* codes, int scrambled, const float* wt, const float* bias, int n_rows, float* x0,
                               long x_bstride, int pitch, int ch, int T, int batch, hipStream_t st) {
    if (T <= 1 || batch <= 0) return 0;
    if (ch != 32 && ch != 64) return wn_set_error_msg(-3, "causal_fwd_codes: padded channel count must be 32 or 64");
    dim3 g((T + CW_TC - 1) / CW_TC, batch);
    if (ch == 32) hipLaunchKernelGGL(causal_fwd_codes_k<32>, g, dim3(CW_THREADS), 0, st, codes, scrambled, wt, bias, n_rows, x0, x_bstride, pitch, T);
    else hipLaunchKernelGGL(causal_fwd_codes_k<64>, g, dim3(CW_THREADS), 0, st, codes, scrambled, wt, bias, n_rows, x0, x_bstride, pitch, T);
    WN_CHECK_LAUNCH();
    return 0;
}
