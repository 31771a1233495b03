// Backward of one ENCODER block of the autoencoder (wavenet_autoencoder/model1.py:137-152) in ONE launch, data gradient
// included: the (P, Q) form of wn_respq.hip for the block that has no gate and recomputes nothing (CH = 64 padded
// channels, gradient products in BF16x3, no biases; biased blocks keep enc_bwd_rw_k + chan_gemm_rw_k).
//   forward:  h = W0 relu x(t-d) + W1 relu x(t) ; x' = Wd relu(h) + x(t)        (h = the stored pre-activation)
//   backward: dh = (Wd^T dy) [h > 0]
//             dWd = sum_t dy relu(h)^T ;  dWdil = sum_t dh [relu x(t-d); relu x(t)]^T      -> one slab per workgroup
//             dx[s] = dy[s] + [x(s) > 0] (W1^T dh[s] + W0^T dh[s + d])
// What enc_bwd_rw_k leaves to a second launch is the last line, which made dh travel to HBM and back twice and x and dy be
// read again.  Here the block hands dx on as the UNSHIFTED pair
//     P[t] = dy[t] + [x(t) > 0] W1^T dh[t]          Q[t] = [x(t-d) > 0] W0^T dh[t]         (dx[s] = P[s] + Q[s + d])
// - both masks are the block's OWN input at the columns a lane already holds - and takes dy from the block above in the
// same form (P_in[t] for t >= p_lo, + Q_in[t + dn]; the top block: the plain tensor as P_in, no Q_in).  Per block: x, h,
// P_in, Q_in in, P, Q out = 6 activation tensors against 10.
//
// Division of labour (8 waves, two per SIMD, 32-column items, two LDS stages, one barrier per item), tiles, swizzle and the
// transposed (P, Q) product exactly as in wn_respq.hip:
//   * R waves (0..3; wave g = h channels 16g..): dy fragments of the next item, dr = Wd^T dy, mask with h, leave dh and
//     relu(h) in LDS as 16-bit hi/lo tiles [channel][time], and the Q half of the previous item (mask from x(t-d) rows
//     loaded in the output layout);
//   * W waves (4..7; wave g = ROW tile g of relu x(t-d), relu x(t), dy): the weight-gradient columns of their rows (all 64
//     dh rows x their 32 x columns, all 64 relu(h) rows x their 16 dy columns) and the P half, with the lane's own fp32 dy
//     rows as the residual term and the sign bits of its x(t) rows as the mask.
#include <stdlib.h>
#include <string.h>
#include "wn_common.h"
#include "wn_kernels.h"
#include "wn_pqchain.h"

#define EP_THREADS 512
#define EP_CH 64
#define EP_COLS 32
// LDS map, in halfs (uint16): per stage 4 dy fragments | 8 result tiles (dh 0..3, relu h 4..7); then the packed (P, Q) weights
#define EP_DYF 0
#define EP_T 4096
#define EP_STAGE 12288
#define EP_W (2 * EP_STAGE)
// CHAIN form: what the W waves hand the R waves per item (two parities): their fp32 dy rows of samples 0..15 and the ReLU signs of x(t) / x(t-d)
#define EP_HD (EP_W + 16384)                               // [parity][wave g][lane] f32x4
#define EP_HK (EP_HD + 4096)                               // [parity][wave g][lane] uint32: keep | keepb << 4
// LCH form: the unmasked Q rows of the last three items, fp32 [slot][channel][32 samples + 4] (9 KB each)
#define EP_QB (EP_HK + 1024)
#define EP_QROW 36
#define EP_LDS_HALFS_CHAIN (EP_HK + 1024)
#define EP_LDS_HALFS (EP_QB + 3 * EP_CH * EP_QROW * 2)

typedef float ep_f32x2 __attribute__((ext_vector_type(2)));
typedef short ep_s16x4 __attribute__((ext_vector_type(4)));
typedef short ep_s16x8 __attribute__((ext_vector_type(8)));
struct __attribute__((packed, aligned(4))) EpF2U { float v[2]; };
__device__ __forceinline__ ep_f32x2 ep_ld2u(const float* p) {
    EpF2U u = *reinterpret_cast<const EpF2U*>(p);
    ep_f32x2 r = {u.v[0], u.v[1]};
    return r;
}
__device__ __forceinline__ void ep_split2(float a, float b, uint32_t& hi, uint32_t& lo) { split2<BF16>(a, b, hi, lo); }
// chunk swizzle of the result tiles (wn_respq.hip): 16-byte chunk `ch` (8 positions) of row `r` sits at slot 16*ch + (r ^ K[ch])
__device__ __forceinline__ int ep_k(int ch) { return ch == 0 ? 0 : ch == 1 ? 13 : ch == 2 ? 6 : 11; }

// FORM 0: the (P, Q) pair; 1: CHAIN (d a multiple of 32, Q rows carried in registers); 2: LCH (d < 32: a workgroup walks ADJACENT items
// downwards - the chain plan of d = 32 - and the Q rows of an item reach the dx rows of the same and the next item through LDS)
template <bool HAS_Q, int FORM>
__global__ __launch_bounds__(EP_THREADS) void enc_bwd_pq_k(WnEncPqArgs a) {
    constexpr bool CHAIN = FORM != 0, LCH = FORM == 2;
    constexpr int CH = EP_CH;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wv & 3;
    const int c = lane & 15, q = lane >> 4;
    const int tile_rd = (16 * q + (c ^ ep_k(q))) * 8;            // halfs; chunk q of row c (a 16-byte row read)

    // items of this workgroup.  (P, Q) form: the workgroups of an XCD walk one contiguous item range interleaved (wn_resrw.hip).
    // CHAIN form (wn_respq.hip, wn_pqchain.h): a run of `n_items` items in CHAIN ORDER (clip, residue of the item index mod d / 32, then
    // downwards in time), the first of them possibly a halo item; a window of five positions (items it-1 .. it+3) lives in scalars.
    struct Pos { int b, t0; bool live, halo, top, bot; };
    int first = 0, cnt = 1, j = 0, wgid, total = 0, i_lo = 0, n_items;
    const PqChain chp = {a.ch_s, a.ch_qn, a.ch_rm, a.ch_g, a.ch_nchain, a.steps_per_clip};
    PqCS cs_front = {0, 0, 0, 1};
    int k_front = 0;
    bool has_halo = false;
    int win_b[5], win_t[5];                                 // clip | t0 (a multiple of 32) + live, halo, top, bot bits
    auto cs_pos = [&](PqCS cc, int k) __attribute__((always_inline)) {
        const int t0 = a.t_base + EP_COLS * (cc.r + (cc.m - 1 - cc.pos) * a.ch_s);
        return t0 | ((k >= 0 && k < n_items) ? 1 : 0) | ((has_halo && k == 0) ? 2 : 0) | (cc.pos == 0 ? 4 : 0) | (cc.pos == cc.m - 1 ? 8 : 0);
    };
    if (CHAIN) {
        wgid = blockIdx.x;
        int n_real;
        PqCS cc;
        pq_chain_start(chp, wgid, gridDim.x, cc, n_real, has_halo);
        n_items = n_real + (has_halo ? 1 : 0);
        win_b[0] = cc.b; win_t[0] = cs_pos(cc, -1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            win_b[k + 1] = cc.b; win_t[k + 1] = cs_pos(cc, k);
            if (k < 3) cc = pq_cs_next(cc, chp);
        }
        cs_front = cc;
        k_front = 3;
    } else {
        if (a.swz) {
            const int nwg = gridDim.x, id = blockIdx.x;
            const int qn = nwg >> 3, rn = nwg & 7, xcd = id & 7;
            first = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
            cnt = xcd < rn ? qn + 1 : qn;
            j = id >> 3;
        } else {
            first = 0; cnt = gridDim.x; j = blockIdx.x;
        }
        wgid = first + j;
        total = a.steps_per_clip * a.batch;
        i_lo = first * a.items_per_wg + j;
        int i_hi = (first + cnt) * a.items_per_wg;
        if (i_hi > total) i_hi = total;
        n_items = i_lo < i_hi ? (i_hi - i_lo + cnt - 1) / cnt : 0;
    }
    auto win_advance = [&]() __attribute__((always_inline)) {
        if (CHAIN) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { win_b[k] = win_b[k + 1]; win_t[k] = win_t[k + 1]; }
            cs_front = pq_cs_next(cs_front, chp);
            k_front += 1;
            win_b[4] = cs_front.b; win_t[4] = cs_pos(cs_front, k_front);
        }
    };
    auto pos_k = [&](int k) {                                  // (P, Q) form: position of this workgroup's k-th item, clamped
        const bool live = k >= 0 && k < n_items;
        k = k < n_items ? k : n_items - 1;
        int it = i_lo + (k < 0 ? 0 : k) * cnt;
        it = it < total ? it : total - 1;
        Pos p;
        p.b = it / a.steps_per_clip;
        p.t0 = a.t_base + EP_COLS * (it - p.b * a.steps_per_clip);
        p.live = live;
        p.halo = p.top = p.bot = false;
        return p;
    };
    // position of item `it + rel`, `it` being the iteration the window stands at (rel = -1 .. 3, a constant at every call site)
    auto pos_r = [&](int it, int rel) __attribute__((always_inline)) {
        if (!CHAIN) return pos_k(it + rel);
        Pos p;
        const int v = win_t[rel + 1];
        p.b = win_b[rel + 1];
        p.t0 = v & ~31;
        p.live = v & 1; p.halo = v & 2; p.top = v & 4; p.bot = v & 8;
        return p;
    };

    // the packed [W1^T; W0^T] weights (16 fragments x 2 KB) and zeros in the result tiles of stage 1 (multiplied once before
    // anything was written to them)
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.wpq);
        u32x4* dst = reinterpret_cast<u32x4*>(lds + EP_W);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k * EP_THREADS + threadIdx.x] = src[k * EP_THREADS + threadIdx.x];
        u32x4* z = reinterpret_cast<u32x4*>(lds + EP_STAGE + EP_T);
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 2; ++k) z[k * EP_THREADS + threadIdx.x] = zero;       // 8 tiles x 2 KB = 16 KB
    }

    const float* q_or_p = HAS_Q ? a.q_in : a.p_in;             // loads stay unconditional
    // dy rows for the R waves' dr product, as fragments: wave g converts rows 4(g&1).. of k-step g>>1
    struct RawD { ep_f32x2 p[4], qq[4]; };
    auto load_dy = [&](RawD& r, Pos ps) {
        const size_t ro = ps.live ? (size_t)ps.b * a.x_bstride + (size_t)(32 * (g >> 1) + 8 * q + 4 * (g & 1)) * a.pitch + ps.t0 + 2 * c : 0;
        const size_t rp = ps.live ? (size_t)a.pitch : 0;
        const int dn = ps.live ? a.dn : 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            r.p[jj] = ep_ld2u(a.p_in + ro + jj * rp);
            r.qq[jj] = HAS_Q ? ep_ld2u(q_or_p + ro + dn + jj * rp) : ep_f32x2{0.f, 0.f};
        }
    };
    // dy[t] = P_in[t] (t >= p_lo) + Q_in[t + dn], on [t_lo, t_hi) only (Q_in is never written beyond t_hi)
    auto dyv = [&](float p, float qv, int t) {
        const float pv = t >= a.p_lo ? p : 0.f;
        return (t >= a.t_lo && t < a.t_hi) ? pv + qv : 0.f;
    };
    // an item whose 32 columns all lie inside [max(p_lo, t_lo), t_hi) needs none of these masks (wave-uniform)
    auto interior = [&](Pos ps) { return ps.live && ps.t0 >= a.p_lo && ps.t0 >= a.t_lo && ps.t0 + EP_COLS <= a.t_hi; };
    auto fill_dy = [&](const RawD& r, Pos ps, int stage) {
        const int tl = ps.t0 + 2 * c;
        uint16_t* dyf = lds + (size_t)stage * EP_STAGE + EP_DYF;
        const int ks = g >> 1, h = g & 1;
        const bool fast = interior(ps);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            uint32_t hh[2], ll[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const float v0 = fast ? r.p[2 * jj][n] + r.qq[2 * jj][n] : dyv(r.p[2 * jj][n], r.qq[2 * jj][n], tl + n);
                const float v1 = fast ? r.p[2 * jj + 1][n] + r.qq[2 * jj + 1][n] : dyv(r.p[2 * jj + 1][n], r.qq[2 * jj + 1][n], tl + n);
                ep_split2(v0, v1, hh[jj], ll[jj]);
            }
            uint16_t* fb = dyf + (size_t)(ks * 2 + n) * 1024 + lane * 8 + h * 4;
            *reinterpret_cast<uint2*>(fb) = uint2{hh[0], hh[1]};
            *reinterpret_cast<uint2*>(fb + 512) = uint2{ll[0], ll[1]};
        }
    };
    // ---- one half of (P, Q)^T = dh^T [W1 | W0] for the item whose result tiles sit in `stage`: rows = time (two 16-sample
    // tiles), columns = the 16 P (sel 0) or Q (sel 1) rows of wave g; `keep` bit 4m+i = the ReLU mask of sample
    // t0 + 16m + 4q + i of that row (x(t) for P, x(t-d) for Q), dy32 = the residual term of P.
    // transposed-read addresses (halfs) of this lane inside a tile plane (wn_respq.hip)
    int tr_off[2];
    {
        const int qp = c >> 2, p = c & 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) tr_off[h] = (16 * p + ((8 * (q & 1) + 4 * h + qp) ^ ep_k(p))) * 8;
    }
    auto pq_half = [&](int stage, int sel, Pos ps, const float* dy32, uint32_t keep) __attribute__((always_inline)) {
        const uint16_t* tt = lds + (size_t)stage * EP_STAGE + EP_T;
        const uint16_t* pw = lds + EP_W;
        f32x4 acc[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {                           // two k-steps, two time tiles: four chains in flight
            acc[s][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[s][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            Frag<BF16> w;
            load_a<BF16, 3>(w, pw, (sel * 4 + g) * 2 + s, lane);
            const uint16_t* tb = tt + (2 * s + (q >> 1)) * 1024;
            Frag<BF16> ad[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                typedef __attribute__((address_space(3))) ep_s16x4 lds_s16x4;
                ep_s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + tr_off[0] + 4 * m));
                ep_s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + tr_off[1] + 4 * m));
                ep_s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 512 + tr_off[0] + 4 * m));
                ep_s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 512 + tr_off[1] + 4 * m));
                ep_s16x8 hh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                ep_s16x8 ll = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                ad[m].hi = __builtin_bit_cast(bf16x8, hh);
                ad[m].lo = __builtin_bit_cast(bf16x8, ll);
            }
            acc[s][0] = BF16::mfma(ad[0].lo, w.hi, acc[s][0]);
            acc[s][1] = BF16::mfma(ad[1].lo, w.hi, acc[s][1]);
            acc[s][0] = BF16::mfma(ad[0].hi, w.lo, acc[s][0]);
            acc[s][1] = BF16::mfma(ad[1].hi, w.lo, acc[s][1]);
            acc[s][0] = BF16::mfma(ad[0].hi, w.hi, acc[s][0]);
            acc[s][1] = BF16::mfma(ad[1].hi, w.hi, acc[s][1]);
        }
        if (ps.live) {
            float* out = (sel ? a.q_out : a.p_out) + (size_t)ps.b * a.x_bstride + (size_t)(16 * g + c) * a.pitch + ps.t0 + 4 * q;
            const bool whole = ps.t0 >= a.t_lo && ps.t0 + EP_COLS <= a.t_hi;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f32x4 v = acc[0][m] + acc[1][m];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = ((keep >> (4 * m + i)) & 1u) ? v[i] : 0.f;
                    if (dy32 != nullptr) v[i] += dy32[4 * m + i];
                }
                if (whole) *reinterpret_cast<f32x4*>(out + 16 * m) = v;      // plain: the next launch finds P and Q in L2
                else st4m(out + 16 * m, v, ps.t0 + 16 * m + 4 * q, a.t_lo, a.t_hi);
            }
        }
    };

    // ---- CHAIN form: the 16-sample half `mt` of the item's 32 columns, BOTH weight halves: dx rows = dy + [x > 0] (P rows + the Q
    // rows carried from the item above), then this item's Q rows become the carry; the last item of a chain leaves its carry d columns
    // further down (dx on [t_lo - d, t_base): nothing but the masked Q).  A halo item only makes the carry.  keep4 / keepb4: the ReLU
    // masks of the lane's four samples at x(t) / x(t - d).
    // the P and Q sums of the 16-sample tile `mt` (both weight halves), nothing added, nothing masked
    auto pq_mt_acc = [&](int stage, int mt, f32x4& sP, f32x4& sQ) __attribute__((always_inline)) {
        const uint16_t* tt = lds + (size_t)stage * EP_STAGE + EP_T;
        const uint16_t* pw = lds + EP_W;
        f32x4 aP[2], aQ[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            aP[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            aQ[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            Frag<BF16> w1, w0;
            load_a<BF16, 3>(w1, pw, g * 2 + s, lane);
            load_a<BF16, 3>(w0, pw, (4 + g) * 2 + s, lane);
            const uint16_t* tb = tt + (2 * s + (q >> 1)) * 1024;
            typedef __attribute__((address_space(3))) ep_s16x4 lds_s16x4;
            ep_s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + tr_off[0] + 4 * mt));
            ep_s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + tr_off[1] + 4 * mt));
            ep_s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 512 + tr_off[0] + 4 * mt));
            ep_s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 512 + tr_off[1] + 4 * mt));
            ep_s16x8 hh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            ep_s16x8 ll = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            Frag<BF16> ad;
            ad.hi = __builtin_bit_cast(bf16x8, hh);
            ad.lo = __builtin_bit_cast(bf16x8, ll);
            aP[s] = BF16::mfma(ad.lo, w1.hi, aP[s]);
            aQ[s] = BF16::mfma(ad.lo, w0.hi, aQ[s]);
            aP[s] = BF16::mfma(ad.hi, w1.lo, aP[s]);
            aQ[s] = BF16::mfma(ad.hi, w0.lo, aQ[s]);
            aP[s] = BF16::mfma(ad.hi, w1.hi, aP[s]);
            aQ[s] = BF16::mfma(ad.hi, w0.hi, aQ[s]);
        }
        sP = aP[0] + aP[1];
        sQ = aQ[0] + aQ[1];
    };
    // ---- LCH (d < 32, adjacent items walked downwards): each role takes ONE 16-sample tile of an item (R waves tile 0, W waves tile 1) -
    // its P and Q sums, the Q sums to LDS (slot = item index mod 3), and an iteration later, when both tiles' Q rows of that item and
    // of the item above are there, its tile of dx[s] = dy[s] + [x(s) > 0] (P[s] + Q[s + d]).  A chain's last item also owns the columns
    // below it: dx[s] = [x(s) > 0] Q[s + d] on [t_lo - d, t_base).
    struct LchHold { f32x4 p; float dy4[4]; uint32_t keep4; Pos ps; };
    auto lch_put = [&](int k, int mt, const f32x4& sQ) __attribute__((always_inline)) {
        float* qb = reinterpret_cast<float*>(lds + EP_QB) + ((size_t)(((k % 3) + 3) % 3) * CH + 16 * g + c) * EP_QROW + 16 * mt + 4 * q;
        *reinterpret_cast<f32x4*>(qb) = sQ;
    };
    auto lch_dx = [&](int k, int mt, const LchHold& h) __attribute__((always_inline)) {      // tile mt of dx of item k (held in h)
        const Pos ps = h.ps;
        if (!ps.live || ps.halo) return;
        const float* qbase = reinterpret_cast<const float*>(lds + EP_QB);
        const float* q_own = qbase + ((size_t)(((k % 3) + 3) % 3) * CH + 16 * g + c) * EP_QROW;
        const float* q_abv = qbase + ((size_t)((((k - 1) % 3) + 3) % 3) * CH + 16 * g + c) * EP_QROW;
        float* out = a.p_out + (size_t)ps.b * a.x_bstride + (size_t)(16 * g + c) * a.pitch + ps.t0 + 16 * mt + 4 * q;
        const int tq = ps.t0 + 16 * mt + 4 * q;
        // Q[s + d] of the lane's four samples: position jd0 .. jd0 + 3 counted from this item's first column, out of the item's own slot or,
        // beyond 32, the slot of the item above (a chain's first item has none).  One 16-byte read when d is a multiple of 4 (the four sit
        // on one side of the boundary), two 8-byte reads for an even d, four 4-byte reads otherwise (those hit 8 banks: 8-way conflicts)
        const int jd0 = 16 * mt + 4 * q + a.d;
        f32x4 qv4;
        if ((a.d & 3) == 0) {
            const float* src = jd0 < EP_COLS ? q_own + jd0 : q_abv + (jd0 - EP_COLS);
            qv4 = *reinterpret_cast<const f32x4*>(src);
            if (jd0 >= EP_COLS && ps.top) qv4 = f32x4{0.f, 0.f, 0.f, 0.f};
        } else if ((a.d & 1) == 0) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int jd = jd0 + 2 * hh;
                const float* src = jd < EP_COLS ? q_own + jd : q_abv + (jd - EP_COLS);
                ep_f32x2 t2 = *reinterpret_cast<const ep_f32x2*>(src);
                if (jd >= EP_COLS && ps.top) t2 = ep_f32x2{0.f, 0.f};
                qv4[2 * hh] = t2[0];
                qv4[2 * hh + 1] = t2[1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int jd = jd0 + i;
                qv4[i] = jd < EP_COLS ? q_own[jd] : (ps.top ? 0.f : q_abv[jd - EP_COLS]);
            }
        }
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (((h.keep4 >> i) & 1u) ? h.p[i] + qv4[i] : 0.f) + h.dy4[i];
        if (ps.t0 + EP_COLS <= a.t_hi && ps.t0 >= a.t_lo - a.d) *reinterpret_cast<f32x4*>(out) = v;
        else st4m(out, v, tq, a.t_lo - a.d, a.t_hi);
        if (ps.bot) {                                           // the columns below the chain: nothing but the masked Q of this item
            const f32x4 xv = ld4u(a.x_in + (size_t)ps.b * a.x_bstride + (size_t)(16 * g + c) * a.pitch + tq - EP_COLS);
            f32x4 vb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int jd = 16 * mt + 4 * q + i + a.d - EP_COLS;     // Q[s + d] for s = t0 - 32 + (16 mt + 4q + i)
                vb[i] = (jd >= 0 && xv[i] > 0.f) ? q_own[jd] : 0.f;
            }
            st4m(out - EP_COLS, vb, tq - EP_COLS, a.t_lo - a.d, a.t_lo);
        }
    };
    auto pq_mt = [&](int stage, int mt, Pos ps, const float* dy4, uint32_t keep4, uint32_t keepb4, f32x4& carry) __attribute__((always_inline)) {
        if (!ps.live) return;
        const uint16_t* tt = lds + (size_t)stage * EP_STAGE + EP_T;
        const uint16_t* pw = lds + EP_W;
        f32x4 aP[2], aQ[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            aP[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            aQ[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            Frag<BF16> w1, w0;
            load_a<BF16, 3>(w1, pw, g * 2 + s, lane);
            load_a<BF16, 3>(w0, pw, (4 + g) * 2 + s, lane);
            const uint16_t* tb = tt + (2 * s + (q >> 1)) * 1024;
            typedef __attribute__((address_space(3))) ep_s16x4 lds_s16x4;
            ep_s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + tr_off[0] + 4 * mt));
            ep_s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + tr_off[1] + 4 * mt));
            ep_s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 512 + tr_off[0] + 4 * mt));
            ep_s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 512 + tr_off[1] + 4 * mt));
            ep_s16x8 hh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            ep_s16x8 ll = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            Frag<BF16> ad;
            ad.hi = __builtin_bit_cast(bf16x8, hh);
            ad.lo = __builtin_bit_cast(bf16x8, ll);
            aP[s] = BF16::mfma(ad.lo, w1.hi, aP[s]);
            aQ[s] = BF16::mfma(ad.lo, w0.hi, aQ[s]);
            aP[s] = BF16::mfma(ad.hi, w1.lo, aP[s]);
            aQ[s] = BF16::mfma(ad.hi, w0.lo, aQ[s]);
            aP[s] = BF16::mfma(ad.hi, w1.hi, aP[s]);
            aQ[s] = BF16::mfma(ad.hi, w0.hi, aQ[s]);
        }
        if (ps.top) carry = f32x4{0.f, 0.f, 0.f, 0.f};
        float* out = a.p_out + (size_t)ps.b * a.x_bstride + (size_t)(16 * g + c) * a.pitch + ps.t0 + 16 * mt + 4 * q;
        const int tq = ps.t0 + 16 * mt + 4 * q;
        if (!ps.halo) {
            f32x4 v = (aP[0] + aP[1]) + carry;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (((keep4 >> i) & 1u) ? v[i] : 0.f) + dy4[i];
            if (ps.t0 + EP_COLS <= a.t_hi) *reinterpret_cast<f32x4*>(out) = v;      // (t0 >= t_base > t_lo - d always)
            else st4m(out, v, tq, a.t_lo - a.d, a.t_hi);
        }
        carry = aQ[0] + aQ[1];
        if (ps.bot) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ((keepb4 >> i) & 1u) ? carry[i] : 0.f;
            st4m(out - a.d, v, tq - a.d, a.t_lo - a.d, a.t_lo);
        }
    };

    if (wv < 4) {
        // =========================== R waves: dr = Wd^T dy, mask, tiles; Q half (CHAIN: samples 0..15 of dx) ===========================
        Frag<BF16> wd[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) load_a<BF16, 3>(wd[s], a.wdT, g * 2 + s, lane);
        // this lane's dwords in the result tiles: row 4q + i, samples 2c, 2c+1 (wn_respq.hip)
        int t_wr[4];
        {
            const int chk = (c & 7) >> 1, dw = 2 * (c >> 3) + (c & 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) t_wr[i] = (16 * chk + ((4 * q + i) ^ ep_k(chk))) * 8 + dw * 2;
        }
        // h in the layout of the dr accumulators: rows 16g + 4q + i, samples t0 + 2c, + 1
        auto load_h = [&](ep_f32x2* hr, Pos ps) {
            const float* hp = ps.live ? a.h + (size_t)ps.b * a.h_bstride + (size_t)(16 * g + 4 * q) * a.pitch + ps.t0 + 2 * c : a.h;
            const size_t rp = ps.live ? (size_t)a.pitch : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) hr[i] = ep_ld2u(hp + i * rp);
        };
        // CHAIN: what the wave's half of dx needs besides the product - the fp32 dy rows (the residual term) and the ReLU signs of x(t) /
        // x(t - d) at row 16g + c, samples t0 + 4q .. + 3 - is what W wave g holds for the same lane: it leaves them in LDS when it converts
        // the item's rows (loaded here in the output layout instead - three 16-byte loads per lane that touch 16 cache lines each - the chain
        // form's launches took 4.8 us longer: encoder stack backward 1.523 against 1.451 ms in a timing build without them)
        f32x4 carry = {0.f, 0.f, 0.f, 0.f};
        auto mt_r = [&](int stage, Pos ps) __attribute__((always_inline)) {      // CHAIN: samples 0..15 of dx of item ps
            if (!ps.live) return;
            const f32x4 dv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds + EP_HD) + ((stage * 4 + g) * 64 + lane) * 16);
            const uint32_t kk = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds + EP_HK) + ((stage * 4 + g) * 64 + lane) * 4);
            float d4[4] = {dv[0], dv[1], dv[2], dv[3]};
            pq_mt(stage, 0, ps, d4, kk & 15u, (kk >> 8) & 15u, carry);
        };
        // LCH: tile 0 of an item - products, Q to LDS, and the item before it gets its dx tile; the residual rows and masks of that tile
        // come from W wave g (EP_HD / EP_HK of the item's parity), read one iteration before they are used (the parity is rewritten then)
        LchHold hold_r;
        hold_r.p = f32x4{0.f, 0.f, 0.f, 0.f};
        hold_r.keep4 = 0;
        hold_r.ps = pos_r(0, -1);
#pragma unroll
        for (int i = 0; i < 4; ++i) hold_r.dy4[i] = 0.f;
        auto lch_r = [&](int stage, Pos ps, int k) __attribute__((always_inline)) {      // k = index of the item in `stage`
            f32x4 sP = {0.f, 0.f, 0.f, 0.f}, sQ;
            if (ps.live) {
                pq_mt_acc(stage, 0, sP, sQ);
                lch_put(k, 0, sQ);
            }
            lch_dx(k - 1, 0, hold_r);
            hold_r.p = sP;
            hold_r.ps = ps;
            const f32x4 dv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds + EP_HD) + ((stage * 4 + g) * 64 + lane) * 16);
            hold_r.keep4 = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds + EP_HK) + ((stage * 4 + g) * 64 + lane) * 4) & 15u;
#pragma unroll
            for (int i = 0; i < 4; ++i) hold_r.dy4[i] = dv[i];
        };
        ep_f32x2 hA[4], hB[4];
        load_h(hA, pos_r(0, 0));
        load_h(hB, pos_r(0, 1));
        RawD rd;                                            // dy rows (as the pair) of item it+1
        load_dy(rd, pos_r(0, 0));
        fill_dy(rd, pos_r(0, 0), 0);
        load_dy(rd, pos_r(0, 1));
        // the Q mask of an item = the signs of x(t - d) at this lane's row and samples: W wave g holds those rows for the weight gradients and
        // leaves the bits in LDS (a load of the rows in the output layout here costs the launch 3 us)
        auto q_keep = [&](int par) __attribute__((always_inline)) {
            return (*reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds + EP_HK) + ((par * 4 + g) * 64 + lane) * 4) >> 8) & 255u;
        };
        __syncthreads();                                    // stage 0 operands of the first item, the weights, the zeros
        auto r_body = [&](const int it, ep_f32x2* hr) {
            if (it >= n_items) {
                // the void item that pads an odd count: only the Q rows (CHAIN: the first half of dx) of the last real item
                if (LCH) lch_r((it + 1) & 1, pos_r(it, -1), it - 1);
                else if (CHAIN) mt_r((it + 1) & 1, pos_r(it, -1));
                else pq_half((it + 1) & 1, 1, pos_r(it, -1), nullptr, q_keep((it + 1) & 1));
                win_advance();
                __syncthreads();
                return;
            }
            fill_dy(rd, pos_r(it, 1), (it + 1) & 1);         // dy fragments of the next item
            load_dy(rd, pos_r(it, 2));
            const Pos p_cur = pos_r(it, 0);
            if (LCH) {
                lch_r((it + 1) & 1, pos_r(it, -1), it - 1);  // tile 0 of the previous item; dx tile 0 of the one before it
            } else if (CHAIN) {
                mt_r((it + 1) & 1, pos_r(it, -1));           // the first half of dx of the previous item
            } else {
                pq_half((it + 1) & 1, 1, pos_r(it, -1), nullptr, q_keep((it + 1) & 1));   // Q rows of the previous item
            }
            const int tl = p_cur.t0 + 2 * c;
            uint16_t* st = lds + (size_t)(it & 1) * EP_STAGE;
            const uint16_t* dyf = st + EP_DYF;
            f32x4 dr[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            {
                Frag<BF16> by[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) load_a<BF16, 3>(by[f], dyf, f, lane);      // fragment (k-step f >> 1, N-tile f & 1)
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        dr[f & 1] = t == 0 ? BF16::mfma(wd[f >> 1].lo, by[f].hi, dr[f & 1])
                                  : t == 1 ? BF16::mfma(wd[f >> 1].hi, by[f].lo, dr[f & 1])
                                           : BF16::mfma(wd[f >> 1].hi, by[f].hi, dr[f & 1]);
            }
            uint16_t* tt = st + EP_T;
            const bool ok0 = tl >= a.t_lo && tl < a.t_hi, ok1 = tl + 1 >= a.t_lo && tl + 1 < a.t_hi;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float vh[2], vr[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const bool ok = n ? ok1 : ok0;
                    const float hv = hr[i][n];
                    vh[n] = (ok && hv > 0.f) ? dr[n][i] : 0.f;
                    vr[n] = ok ? fmaxf(hv, 0.f) : 0.f;
                }
                auto put = [&](int kind, const float* v) {
                    uint32_t hi, lo;
                    ep_split2(v[0], v[1], hi, lo);
                    uint16_t* p = tt + (kind * 4 + g) * 1024 + t_wr[i];
                    *reinterpret_cast<uint32_t*>(p) = hi;
                    *reinterpret_cast<uint32_t*>(p + 512) = lo;
                };
                put(0, vh);
                put(1, vr);
            }
            load_h(hr, pos_r(it, 2));
            win_advance();
            __syncthreads();
        };
        for (int it = 0; it < n_items; it += 2) {
            r_body(it, hA);
            r_body(it + 1, hB);
        }
        {
            const int n_even = (n_items + 1) & ~1;
            const Pos pl = pos_r(n_even, -1);
            if (LCH) {
                lch_r((n_even - 1) & 1, pl, n_even - 1);        // (a void last item: only the dx tile of the item before it)
            } else if (pl.live) {                               // Q rows (CHAIN: the first half of dx) of the last item
                if (CHAIN) mt_r((n_even - 1) & 1, pl);
                else pq_half((n_even - 1) & 1, 1, pl, nullptr, q_keep((n_even - 1) & 1));
            }
        }
        __syncthreads();                                    // the W waves' extra round (products of the last item)
        if (LCH) lch_dx(((n_items + 1) & ~1) - 1, 0, hold_r);     // ... whose dx tile needs the W waves' Q rows of it
        return;
    }

    // =========================== W waves: weight gradients and the P half ===========================
    f32x4 cfg[4][2], cd[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { cfg[m][0] = f32x4{0.f, 0.f, 0.f, 0.f}; cfg[m][1] = f32x4{0.f, 0.f, 0.f, 0.f}; cd[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // raw rows of this wave's row tile: lane (row c, q) holds samples t0 + 4q .. + 3 and t0 + 16 + 4q .. + 3
    struct RawRows { f32x4 x0[2], x1[2], p[2], qq[2]; };
    // CHAIN: the x(t) rows of an item that continues its chain ARE the x(t - d) rows of the item before it (the chain steps d columns down):
    // only a chain's first item and the workgroup's first item request them (`fresh`); the others take the previous item's x0 rows
    auto load_rows = [&](RawRows& r, Pos ps, bool fresh) {
        const size_t ro = ps.live ? (size_t)ps.b * a.x_bstride + (size_t)(16 * g + c) * a.pitch + ps.t0 + 4 * q : 0;
        const int dd = ps.live ? a.d : 0, dn = ps.live ? a.dn : 0, h = ps.live ? 16 : 0;
        const float* xr = a.x_in + ro;
        r.x0[0] = ld4u(xr - dd); r.x0[1] = ld4u(xr - dd + h);
        if (!CHAIN || LCH || fresh) { r.x1[0] = ld4u(xr); r.x1[1] = ld4u(xr + h); }
        r.p[0] = ld4u(a.p_in + ro); r.p[1] = ld4u(a.p_in + ro + h);
        if (HAS_Q) { r.qq[0] = ld4u(q_or_p + ro + dn); r.qq[1] = ld4u(q_or_p + ro + dn + h); }
        else { r.qq[0] = f32x4{0.f, 0.f, 0.f, 0.f}; r.qq[1] = r.qq[0]; }
    };
    // this wave's B operands of the weight-gradient products (k = the 8 positions of the lane's chunk), its dy rows in fp32
    // (the residual term of P) and the ReLU mask of its x(t) rows
    struct Ops { Frag<BF16> x0, x1, dy; float dy32[8]; uint32_t keep, keepb; };      // keepb (CHAIN): the signs of the x(t - d) rows
    auto to_frag = [&](Frag<BF16>& f, const float* w) {
        u32x4 fh, fl;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            uint32_t hi, lo;
            ep_split2(w[2 * jj], w[2 * jj + 1], hi, lo);
            fh[jj] = hi;
            fl[jj] = lo;
        }
        f.hi = __builtin_bit_cast(bf16x8, fh);
        f.lo = __builtin_bit_cast(bf16x8, fl);
    };
    f32x4 prev_x0[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto convert = [&](Ops& o, const RawRows& r, Pos ps, int par, bool fresh) {
        float w[8];
        uint32_t keepb = 0;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float v = r.x0[jj >> 2][jj & 3];
            w[jj] = ps.live ? fmaxf(v, 0.f) : 0.f;
            keepb |= v > 0.f ? 1u << jj : 0u;
        }
        to_frag(o.x0, w);
        o.keepb = keepb;
        uint32_t keep = 0;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float v = (!CHAIN || LCH || fresh) ? r.x1[jj >> 2][jj & 3] : prev_x0[jj >> 2][jj & 3];
            w[jj] = ps.live ? fmaxf(v, 0.f) : 0.f;
            keep |= v > 0.f ? 1u << jj : 0u;
        }
        to_frag(o.x1, w);
        if (CHAIN && !LCH) { prev_x0[0] = r.x0[0]; prev_x0[1] = r.x0[1]; }
        o.keep = keep;
        if (interior(ps)) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o.dy32[jj] = r.p[jj >> 2][jj & 3] + r.qq[jj >> 2][jj & 3];
        } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int t = ps.t0 + 16 * (jj >> 2) + 4 * q + (jj & 3);
                o.dy32[jj] = ps.live ? dyv(r.p[jj >> 2][jj & 3], r.qq[jj >> 2][jj & 3], t) : 0.f;
            }
        }
        to_frag(o.dy, o.dy32);
        // what R wave g needs of these rows for the same lane, parity of the item: the ReLU signs of x(t) / x(t - d) (its Q mask; CHAIN:
        // the masks of samples 0..15 of dx) and, CHAIN, the fp32 dy rows of samples 0..15 (the residual term)
        *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds + EP_HK) + ((par * 4 + g) * 64 + lane) * 4) = o.keep | (o.keepb << 8);
        if (CHAIN)
            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(lds + EP_HD) + ((par * 4 + g) * 64 + lane) * 16) =
                f32x4{o.dy32[0], o.dy32[1], o.dy32[2], o.dy32[3]};
    };
    auto load_tile = [&](Frag<BF16>& f, const uint16_t* base, int tile) {
        const u32x4* p = reinterpret_cast<const u32x4*>(base + tile * 1024 + tile_rd);
        f.hi = __builtin_bit_cast(bf16x8, p[0]);
        f.lo = __builtin_bit_cast(bf16x8, p[64]);
    };
    f32x4 carry_w = {0.f, 0.f, 0.f, 0.f};                  // CHAIN: Q rows of samples 16..31 of the item above
    f32x4 p_new = {0.f, 0.f, 0.f, 0.f};                     // LCH: the P sums (tile 1) of the item `products` just saw
    auto products = [&](int stage, const Ops& o, Pos ps, int k_item) __attribute__((always_inline)) {
        const uint16_t* tt = lds + (size_t)stage * EP_STAGE + EP_T;
        if (!(CHAIN && ps.halo)) {                              // (a halo item's weight gradients belong to the workgroup above)
            // weight gradients: rows = all dh / relu(h) tiles, columns = this wave's x / dy rows; the three products of an x3
            // term are walked across the accumulators of a tile pair, the next pair is read meanwhile
            auto term = [](f32x4& acc, const Frag<BF16>& wa, const Frag<BF16>& xb, int t) {
                acc = t == 0 ? BF16::mfma(wa.lo, xb.hi, acc) : t == 1 ? BF16::mfma(wa.hi, xb.lo, acc) : BF16::mfma(wa.hi, xb.hi, acc);
            };
            Frag<BF16> am[2][2];
            load_tile(am[0][0], tt, 0);
            load_tile(am[0][1], tt, 1);
#pragma unroll
            for (int mm = 0; mm < 8; mm += 2) {
                const int cur = (mm >> 1) & 1;
                if (mm + 2 < 8) {
                    load_tile(am[cur ^ 1][0], tt, mm + 2);
                    load_tile(am[cur ^ 1][1], tt, mm + 3);
                }
                if (mm < 4) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        term(cfg[mm][0], am[cur][0], o.x0, t);
                        term(cfg[mm][1], am[cur][0], o.x1, t);
                        term(cfg[mm + 1][0], am[cur][1], o.x0, t);
                        term(cfg[mm + 1][1], am[cur][1], o.x1, t);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        term(cd[mm - 4], am[cur][0], o.dy, t);
                        term(cd[mm - 3], am[cur][1], o.dy, t);
                    }
                }
            }
        }
        if (LCH) {                                              // tile 1: products, Q to LDS; its dx tile an iteration later (w_body)
            p_new = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ps.live) {
                f32x4 sQ;
                pq_mt_acc(stage, 1, p_new, sQ);
                lch_put(k_item, 1, sQ);
            }
        } else if (CHAIN) pq_mt(stage, 1, ps, o.dy32 + 4, (o.keep >> 4) & 15u, (o.keepb >> 4) & 15u, carry_w);      // samples 16..31 of dx
        else pq_half(stage, 0, ps, o.dy32, o.keep);
    };
    LchHold hold_w;
    hold_w.p = f32x4{0.f, 0.f, 0.f, 0.f};
    hold_w.keep4 = 0;
    hold_w.ps = pos_r(0, -1);
#pragma unroll
    for (int i = 0; i < 4; ++i) hold_w.dy4[i] = 0.f;
    auto lch_w_shift = [&](const Ops& o, Pos ps) __attribute__((always_inline)) {      // the item `products` just saw becomes the held one
        hold_w.p = p_new;
        hold_w.keep4 = (o.keep >> 4) & 15u;
        hold_w.ps = ps;
#pragma unroll
        for (int i = 0; i < 4; ++i) hold_w.dy4[i] = o.dy32[4 + i];
    };

    {
        RawRows rr, rr2;                                    // raw rows of items it / it+1: requested two items ahead
        Ops ops;
        load_rows(rr, pos_r(0, 0), true);
        load_rows(rr2, pos_r(0, 1), pos_r(0, 1).top);
        convert(ops, rr, pos_r(0, -1), 1, true);            // "item -1": zeros (its products meet the zeroed tiles of stage 1)
        __syncthreads();
        // iteration it: products of item it-1 (result tiles of stage (it-1)&1, operands in `ops`); then the raw rows of
        // item it become `ops` and the rows of item it+2 are requested (loop unrolled by two: no register copies)
        // (the row conversion between the weight gradients and the dx product - what pays in wn_respq.hip, whose R waves have a long
        // vector phase - is SLOWER here, round 4: encoder stack backward 1.64 against 1.57 ms at config 4, three alternations)
        auto w_body = [&](const int it, RawRows& r) {
            products((it + 1) & 1, ops, pos_r(it, -1), it - 1);
            if (LCH) {
                lch_dx(it - 2, 1, hold_w);                      // the item before the one `products` just saw
                lch_w_shift(ops, pos_r(it, -1));
            }
            convert(ops, r, pos_r(it, 0), it & 1, it == 0 || pos_r(it, 0).top);
            load_rows(r, pos_r(it, 2), pos_r(it, 2).top);
            win_advance();
            __syncthreads();
        };
        const int n_even = (n_items + 1) & ~1;
        for (int it = 0; it < n_even; it += 2) { w_body(it, rr); w_body(it + 1, rr2); }
        const Pos p_last = pos_r(n_even, -1);
        if (p_last.live) products((n_even - 1) & 1, ops, p_last, n_even - 1);    // the last item, unless it is the void one
        if (LCH) lch_dx(n_even - 2, 1, hold_w);
        __syncthreads();
        if (LCH && p_last.live) {                               // ... and its dx tile, once the R waves have left their Q rows of it
            lch_w_shift(ops, p_last);
            lch_dx(n_even - 1, 1, hold_w);
        }
    }

    // ---- slabs of this workgroup (every workgroup writes them, also an idle one: zeros); layouts of enc_bwd_rw_k
    float* sfg = a.slab_dil + (size_t)wgid * (2 * CH * CH);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int tap = 0; tap < 2; ++tap)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_nontemporal_store(cfg[m][tap][i], &sfg[(size_t)(16 * m + 4 * q + i) * (2 * CH) + tap * CH + 16 * g + c]);
    float* sd = a.slab_d + (size_t)wgid * (CH * CH);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_nontemporal_store(cd[m][i], &sd[(size_t)(16 * g + c) * CH + 16 * m + 4 * q + i]);
}

int wn_launch_enc_bwd_pq(const WnEncPqArgs& a, int ch, int batch, int mode_bwd, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    if (ch != EP_CH) return wn_set_error_msg(-3, "enc_resblock_bwd_pq: 64 padded channels only");
    if (mode_bwd != WN_MODE_BF16X3) return wn_set_error_msg(-2, "enc_resblock_bwd_pq: bf16x3 only");
    WnEncPqArgs k = a;
    int nwg;
    if (k.chain == 2) {
        // LCH: d < 32; the workgroups walk adjacent items downwards = the chain plan of d = 32 (one chain per clip)
        if (a.d >= EP_COLS || a.d < 1) return wn_set_error_msg(-4, "enc_resblock_bwd_pq: chain form 2 is for 1 <= d < 32");
        wn_pq_chain_plan(a.t_lo, a.t_hi, batch, EP_COLS, k.t_base, k.steps_per_clip, k.ch_s, k.ch_qn, k.ch_rm, k.ch_g, k.ch_nchain, nwg);
        k.items_per_wg = 0;
    } else if (k.chain) {
        if (!wn_pq_chain_ok(a.t_lo, a.t_hi, batch, a.d)) return wn_set_error_msg(-4, "enc_resblock_bwd_pq: chain form needs d % 32 == 0 and an item per chain");
        wn_pq_chain_plan(a.t_lo, a.t_hi, batch, a.d, k.t_base, k.steps_per_clip, k.ch_s, k.ch_qn, k.ch_rm, k.ch_g, k.ch_nchain, nwg);
        k.items_per_wg = 0;
    } else {
        wn_resrw_plan(a.t_lo, a.t_hi, batch, k.t_base, k.steps_per_clip, k.items_per_wg, nwg);      // same items and slabs as enc_bwd_rw_k
    }
    k.batch = batch;
    k.swz = wn_xcd_swizzle_enabled();
    const size_t sh = (size_t)EP_LDS_HALFS * sizeof(uint16_t);
    static WnDevOnce done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done.need(dev)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_bwd_pq_k<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_bwd_pq_k<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_bwd_pq_k<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_bwd_pq_k<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_bwd_pq_k<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_bwd_pq_k<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        done.done(dev);
    }
    if (k.chain == 2) {
        if (k.q_in) hipLaunchKernelGGL((enc_bwd_pq_k<true, 2>), dim3(nwg), dim3(EP_THREADS), sh, st, k);
        else hipLaunchKernelGGL((enc_bwd_pq_k<false, 2>), dim3(nwg), dim3(EP_THREADS), sh, st, k);
    } else if (k.chain) {
        if (k.q_in) hipLaunchKernelGGL((enc_bwd_pq_k<true, 1>), dim3(nwg), dim3(EP_THREADS), sh, st, k);
        else hipLaunchKernelGGL((enc_bwd_pq_k<false, 1>), dim3(nwg), dim3(EP_THREADS), sh, st, k);
    } else {
        if (k.q_in) hipLaunchKernelGGL((enc_bwd_pq_k<true, 0>), dim3(nwg), dim3(EP_THREADS), sh, st, k);
        else hipLaunchKernelGGL((enc_bwd_pq_k<false, 0>), dim3(nwg), dim3(EP_THREADS), sh, st, k);
    }
    WN_CHECK_LAUNCH();
    return 0;
}
