// Backward of one gated residual block in ONE launch, data gradient included (CH = 64, recompute in F16x3,
// gradient products in BF16x3; no biases: biased blocks keep resblock_bwd_rw_k + chan_gemm_rw_k).  The autoencoder's
// conditioned decoder blocks run it too (COND, at most 32 buckets): the conditioning bias T_b[row][bucket(t)] is one more
// k-step of the recompute (T_b times a 0/1 matrix), and its gradient - a bucket sum over [df;dg] - a 0/1 selection
// product in the W waves (cslab), so that [df;dg] stays on the CU there as well.
//
// What resblock_bwd_rw_k (wn_resrw.hip) leaves to a second launch is the data gradient of the dilated convs,
//     dx_i[t] = W1^T [df;dg][t] + W0^T [df;dg][t + d] + dx_{i+1}[t],
// which made [df;dg] (2 activation tensors) travel to HBM and back twice.  Here the block hands its data gradient on as
// the UNSHIFTED pair
//     P[t] = W1^T [df;dg][t] + dx_{i+1}[t]        Q[t] = W0^T [df;dg][t]            (dx_i[t] = P[t] + Q[t + d])
// and takes dx_{i+1} from the block above in the same form (P_in[t] for t >= p_lo, + Q_in[t + dn]); [df;dg] never
// leaves the CU.  Per block: x, P_in, Q_in, dz-crop in; P, Q out = 5.9 activation tensors against 9.8.
//
// Division of labour (8 waves, two per SIMD, 32-column items, two LDS stages, one barrier per item, as wn_resrw.hip):
//   * R waves (0..3; wave g = dilation channels 16g..): recompute f, g from x fragments, dz = Wd^T dy, gate, and leave
//     df, dg, z in LDS as 16-bit hi/lo tiles [channel][time].  They no longer store anything to HBM.
//   * W waves (4..7; wave g = ROW tile g of x(t-d), x(t), dy): weight gradients and the (P, Q) product of the PREVIOUS
//     item.  A W wave now owns the COLUMNS of the weight-gradient results that belong to its 16 x rows
//     (all 128 [df;dg] rows x its 32 columns, all 64 z rows x its 16 dy columns), so its own raw rows - 8 samples of one
//     row per lane - ARE the MFMA B operands: no [row][time] operand tiles in LDS (48 KB freed), and the packed
//     [W1^T; W0^T] weights (64 KB) take their place.  The (P, Q) product runs TRANSPOSED, out[time][row] =
//     [df;dg]^T W^T: its A operand ([time][channel]) comes out of the result tiles with ds_read_b64_tr_b16, its result
//     lands as 4 consecutive samples of one row per lane - the layout of the raw dy rows the lane already holds (exact
//     fp32 residual add) and a 16-byte store.
// The time axis inside a tile is permuted (position 8q+j holds sample 4q+j for j < 4, 16+4q+j-4 otherwise) so that one
// 16-byte row read gives the samples of the lane's two 4-sample groups; the R waves' pair writes, the row reads and the
// transposed reads are all conflict free / 2-way with the chunk swizzle K[] below (MI355X_MICROARCH.md LDS table).
//
// CHAIN form (round 4; d a multiple of 32): the pair costs a tensor written and a tensor read per block.  dx_i[t] needs Q of the
// item d columns ABOVE it, so a workgroup that walks the items of one residue class (item index mod d/32) DOWNWARDS in time
// has that Q in registers: the (P, Q) product is split by 16-sample halves instead of by P | Q (R waves: samples 0..15, W
// waves: 16..31, each BOTH weight halves - the same 24 MFMAs per wave), the Q rows of an item stay in the wave that made
// them (4 registers) and are added to the P rows of the next item: dx_i leaves the launch WHOLE (p_out, valid on
// [t_lo - d, t_hi); below t_base it is the last carry of each chain), and x(t - d) of an item is x(t) of the next one (an
// L1 / L2 hit).  A chain is cut into segments for the 256 CUs; a segment that does not start at the top of its chain
// first recomputes the item above it for its Q rows (HALO item: no stores, no weight gradients).  QIN = false: the block
// above handed dx on whole (p_in alone).
// Timing / phase-clock / span builds of this kernel (one ingredient removed, phases reordered, stamps): tools/exp/dev_switches.patch,
// applied to a scratch copy by tools/mkvar.sh - the shipped source builds correct code only.
#include <stdlib.h>
#include <string.h>
#include "wn_common.h"
#include "wn_kernels.h"
#include "wn_pqchain.h"

#define PQ_THREADS 512
#define PQ_CH 64
#define PQ_COLS 32

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
struct __attribute__((packed, aligned(4))) PqF2U { float v[2]; };
__device__ __forceinline__ f32x2 pq_ld2u(const float* p) {
    PqF2U u = *reinterpret_cast<const PqF2U*>(p);
    f32x2 r = {u.v[0], u.v[1]};
    return r;
}
#define PQ_LD2(base, off) pq_ld2u((base) + (off))
#define PQ_LD4(base, off) ld4u((base) + (off))
typedef BF16 PQG;                                          // the 16-bit type of the GRADIENT operands (float32's exponent range)
#define PQ_ONE16 0x3F80u                                   // 1.0 in it
typedef PQG::vec8 pqg8;
// (a, b) -> packed 16-bit pairs hi = (cvt(a), cvt(b)) and lo = (cvt(a - hi_a), cvt(b - hi_b))
__device__ __forceinline__ void pq_split2(float a, float b, uint32_t& hi, uint32_t& lo) { split2<PQG>(a, b, hi, lo); }

// LDS map, in halfs (uint16): per stage 8 x fragments | 4 dy fragments | 12 result tiles; then the packed (P, Q) weights
#define PQ_XF 0
#define PQ_DYF 8192
#define PQ_T 12288
#define PQ_STAGE 24576
#define PQ_W (2 * PQ_STAGE)
#define PQ_LDS_HALFS (PQ_W + 32768)

// chunk swizzle of the result tiles: 16-byte chunk `ch` (8 positions) of row `r` sits at slot 16*ch + (r ^ K[ch])
__device__ __forceinline__ int pq_k(int ch) { return ch == 0 ? 0 : ch == 1 ? 13 : ch == 2 ? 6 : 11; }

template <class T>
__device__ __forceinline__ void pq_store_frag(uint16_t* base, int idx, int lane, const Frag<T>& f) {
    u32x4* p = reinterpret_cast<u32x4*>(base) + (size_t)idx * 128 + lane;
    p[0] = __builtin_bit_cast(u32x4, f.hi);
    p[64] = __builtin_bit_cast(u32x4, f.lo);
}

// COND: the conditioned form (conditioning table in the recompute, bucket sums of [df;dg]); compiled apart so that the
// plain form keeps its register budget (242, no spills)
template <bool HAS_DY, bool COND, bool QIN, bool CHAIN>
__global__ __launch_bounds__(PQ_THREADS) void resblock_bwd_pq_k(WnResPqArgs a) {
    static_assert(!(COND && CHAIN), "the conditioned block keeps the (P, Q) form");
    static_assert(HAS_DY || QIN, "no dy: one form");
    constexpr int CH = PQ_CH;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wv & 3;
    const int c = lane & 15, q = lane >> 4;
    const int tile_rd = (16 * q + (c ^ pq_k(q))) * 8;            // halfs; chunk q of row c (a 16-byte row read)
    // items of this workgroup.  (P, Q) form: the workgroups of an XCD walk one contiguous item range interleaved (wn_resrw.hip).
    // CHAIN form: a run of `n_items` items in CHAIN ORDER (clip, residue r of the item index mod s, then downwards in time),
    // the first of them possibly a halo item; positions are stepped, a window of five (items it-1 .. it+3) lives in scalars.
    struct Pos { int b, t0; bool live, halo, top, bot; };
    int first = 0, cnt = 1, j = 0, wgid, total = 0, i_lo = 0, n_items;
    typedef PqCS CS;
    const PqChain chp = {a.ch_s, a.ch_qn, a.ch_rm, a.ch_g, a.ch_nchain, a.steps_per_clip};
    CS cs_front = {0, 0, 0, 1};
    int k_front = 0;
    bool has_halo = false;
    int win_b[5], win_t[5];                                 // the window, packed: clip | t0 (a multiple of 32) + live, halo, top, bot bits
    auto cs_next = [&](CS c) __attribute__((always_inline)) { return pq_cs_next(c, chp); };
    auto cs_pos = [&](CS c, int k) __attribute__((always_inline)) {      // -> packed t0 + flags
        const int t0 = a.t_base + PQ_COLS * (c.r + (c.m - 1 - c.pos) * a.ch_s);
        return t0 | ((k >= 0 && k < n_items) ? 1 : 0) | ((has_halo && k == 0) ? 2 : 0) | (c.pos == 0 ? 4 : 0) | (c.pos == c.m - 1 ? 8 : 0);
    };
    auto win_get = [&](int i) __attribute__((always_inline)) {
        Pos p;
        const int v = win_t[i];
        p.b = win_b[i];
        p.t0 = v & ~31;
        p.live = v & 1; p.halo = v & 2; p.top = v & 4; p.bot = v & 8;
        return p;
    };
    if (CHAIN) {
        wgid = blockIdx.x;
        int n_real;
        CS c;
        pq_chain_start(chp, wgid, gridDim.x, c, n_real, has_halo);
        n_items = n_real + (has_halo ? 1 : 0);
        win_b[0] = c.b; win_t[0] = cs_pos(c, -1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            win_b[k + 1] = c.b; win_t[k + 1] = cs_pos(c, k);
            if (k < 3) c = cs_next(c);
        }
        cs_front = c;
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
            cs_front = cs_next(cs_front);
            k_front += 1;
            win_b[4] = cs_front.b; win_t[4] = cs_pos(cs_front, k_front);
        }
    };
    auto pos_abs = [&](int k) {                                // (P, Q) form: position of this workgroup's k-th item, clamped
        const bool live = k >= 0 && k < n_items;
        k = k < n_items ? k : n_items - 1;
        int it = i_lo + (k < 0 ? 0 : k) * cnt;
        it = it < total ? it : total - 1;
        Pos p;
        p.b = it / a.steps_per_clip;
        p.t0 = a.t_base + PQ_COLS * (it - p.b * a.steps_per_clip);
        p.live = live;
        p.halo = p.top = p.bot = false;
        return p;
    };
    // position of item `it + rel` where `it` is the iteration the window stands at (rel = -1 .. 3, a constant at every call site)
    auto pos_r = [&](int it, int rel) __attribute__((always_inline)) { return CHAIN ? win_get(rel + 1) : pos_abs(it + rel); };

    // the packed [W1^T; W0^T] weights (32 fragments x 2 KB) and zeros in the result tiles of stage 1 (multiplied once
    // before anything was written to them).  (Requested by LDS-DMA from the W waves instead and first used in iteration
    // 1: the first barrier stays 6 us after the workgroup's start - it waits for the first item's rows, not for these.)
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.wpq);
        u32x4* dst = reinterpret_cast<u32x4*>(lds + PQ_W);
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k * PQ_THREADS + threadIdx.x] = src[k * PQ_THREADS + threadIdx.x];
        u32x4* z = reinterpret_cast<u32x4*>(lds + PQ_STAGE + PQ_T);
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 3; ++k) z[k * PQ_THREADS + threadIdx.x] = zero;       // 12 tiles x 2 KB = 24 KB
    }

    const float* p_or_x = HAS_DY ? a.p_in : a.x_in;         // loads stay unconditional
    const float* q_or_x = (HAS_DY && QIN) ? a.q_in : a.x_in;
    // dy rows for the R waves' dz product, as recompute-style fragments: wave g converts rows 4(g&1).. of k-step g>>1
    struct RawD { f32x2 p[4], qq[4]; };
    auto load_dy = [&](RawD& r, Pos ps) {
        if (HAS_DY) {
            const size_t ro = ps.live ? (size_t)ps.b * a.x_bstride + (size_t)(32 * (g >> 1) + 8 * q + 4 * (g & 1)) * a.pitch + ps.t0 + 2 * c : 0;
            const size_t rp = ps.live ? (size_t)a.pitch : 0;
            const int dn = ps.live ? a.dn : 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                r.p[jj] = PQ_LD2(p_or_x, ro + jj * rp);
                if (QIN) r.qq[jj] = PQ_LD2(q_or_x, ro + dn + jj * rp);
                else r.qq[jj] = f32x2{0.f, 0.f};
            }
        }
    };
    // dx_{i+1}[t] = P_in[t] (t >= p_lo) + Q_in[t + dn], on [t_lo, t_hi) only (Q_in is never written beyond t_hi)
    auto dyv = [&](float p, float qv, int t) {
        const float pv = t >= a.p_lo ? p : 0.f;
        return (t >= a.t_lo && t < a.t_hi) ? pv + qv : 0.f;
    };
    // an item whose 32 columns all lie inside [max(p_lo, t_lo), t_hi) needs none of these masks (wave-uniform)
    auto interior = [&](Pos ps) { return ps.live && ps.t0 >= a.p_lo && ps.t0 >= a.t_lo && ps.t0 + PQ_COLS <= a.t_hi; };
    auto fill_dy = [&](const RawD& r, Pos ps, int stage) {
        if (HAS_DY) {
            const int tl = ps.t0 + 2 * c;
            uint16_t* dyf = lds + (size_t)stage * PQ_STAGE + PQ_DYF;
            const int ks = g >> 1, h = g & 1;
            const bool fast = interior(ps);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                uint32_t hh[2], ll[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float v0 = fast ? r.p[2 * jj][n] + r.qq[2 * jj][n] : dyv(r.p[2 * jj][n], r.qq[2 * jj][n], tl + n);
                    const float v1 = fast ? r.p[2 * jj + 1][n] + r.qq[2 * jj + 1][n] : dyv(r.p[2 * jj + 1][n], r.qq[2 * jj + 1][n], tl + n);
                    pq_split2(v0, v1, hh[jj], ll[jj]);
                }
                uint16_t* fb = dyf + (size_t)(ks * 2 + n) * 1024 + lane * 8 + h * 4;
                *reinterpret_cast<uint2*>(fb) = uint2{hh[0], hh[1]};
                *reinterpret_cast<uint2*>(fb + 512) = uint2{ll[0], ll[1]};
            }
        }
    };
    // ---- one half of (P, Q)^T = [df;dg]^T [W1 | W0] for the item whose result tiles sit in `stage`: rows = time (two
    // 16-sample tiles), columns = the 16 P (sel 0) or Q (sel 1) rows of wave g.  The W waves take P (+ their fp32 dy rows),
    // the R waves Q, so that both roles issue 84 MFMAs per item.
    // transposed-read addresses (halfs) of this lane inside a tile plane: lane c = 4q' + p of its 16-lane group supplies
    // row (r0 + q'), piece p (positions 8p + 4m' .. + 3 = samples 16m' + 4p .. + 3); r0 = 8(q&1) + 4h
    int tr_off[2];
    {
        const int qp = c >> 2, p = c & 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) tr_off[h] = (16 * p + ((8 * (q & 1) + 4 * h + qp) ^ pq_k(p))) * 8;
    }
    struct PqAcc { f32x4 a[2], b[2]; };                       // pair form: even / odd k-steps of the two 16-sample tiles; chain form: P | Q rows
    auto pq_zero = [&](PqAcc& r) __attribute__((always_inline)) {
        r.a[0] = r.a[1] = r.b[0] = r.b[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto pq_half_step = [&](int stage, int sel, int s, PqAcc& r) __attribute__((always_inline)) {
        const uint16_t* tt = lds + (size_t)stage * PQ_STAGE + PQ_T;
        const uint16_t* pw = lds + PQ_W;
        Frag<PQG> w;
        load_a<PQG, 3>(w, pw, (sel * 4 + g) * 4 + s, lane);
        const uint16_t* tb = tt + ((s >> 1) * 4 + 2 * (s & 1) + (q >> 1)) * 1024;
        Frag<PQG> ad[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + tr_off[0] + 4 * m));
            s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + tr_off[1] + 4 * m));
            s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 512 + tr_off[0] + 4 * m));
            s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 512 + tr_off[1] + 4 * m));
            s16x8 hh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            s16x8 ll = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            ad[m].hi = __builtin_bit_cast(pqg8, hh);
            ad[m].lo = __builtin_bit_cast(pqg8, ll);
        }
        f32x4* ac = (s & 1) ? r.b : r.a;                      // odd k-steps: four chains in flight
        ac[0] = PQG::mfma(ad[0].lo, w.hi, ac[0]);
        ac[1] = PQG::mfma(ad[1].lo, w.hi, ac[1]);
        ac[0] = PQG::mfma(ad[0].hi, w.lo, ac[0]);
        ac[1] = PQG::mfma(ad[1].hi, w.lo, ac[1]);
        ac[0] = PQG::mfma(ad[0].hi, w.hi, ac[0]);
        ac[1] = PQG::mfma(ad[1].hi, w.hi, ac[1]);
    };
    auto pq_half_tail = [&](int sel, Pos ps, const float* dy32, PqAcc& r) __attribute__((always_inline)) {
        f32x4 acc[2] = {r.a[0] + r.b[0], r.a[1] + r.b[1]};
        if (ps.live) {
            float* out = (sel ? a.q_out : a.p_out) + (size_t)ps.b * a.x_bstride + (size_t)(16 * g + c) * a.pitch + ps.t0 + 4 * q;
            const bool whole = ps.t0 >= a.t_lo && ps.t0 + PQ_COLS <= a.t_hi;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f32x4 v = acc[m];
                if (dy32 != nullptr) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += dy32[4 * m + i];
                }
                if (whole) *reinterpret_cast<f32x4*>(out + 16 * m) = v;      // plain: the next launch finds P and Q in L2 (streaming stores: 2.21 vs 2.00 ms for the stack)
                else st4m(out + 16 * m, v, ps.t0 + 16 * m + 4 * q, a.t_lo, a.t_hi);
            }
        }
    };
    auto pq_half = [&](int stage, int sel, Pos ps, const float* dy32) __attribute__((always_inline)) {
        PqAcc r;
        pq_zero(r);
#pragma unroll
        for (int s = 0; s < 4; ++s) pq_half_step(stage, sel, s, r);
        pq_half_tail(sel, ps, dy32, r);
    };
    // ---- CHAIN form: the 16-sample half `mt` of the item's 32 columns, BOTH weight halves: dx rows = P rows + the Q rows carried
    // from the item above (+ the lane's fp32 dy rows, dy4), then this item's Q rows become the carry; the last item of a chain
    // leaves its carry d columns further down (dx on [t_lo - d, t_base): nothing but Q).  A halo item only makes the carry.
    auto pq_mt_step = [&](int stage, int mt, int s, PqAcc& r) __attribute__((always_inline)) {      // r.a = P rows, r.b = Q rows (even / odd k-steps)
        const uint16_t* tt = lds + (size_t)stage * PQ_STAGE + PQ_T;
        const uint16_t* pw = lds + PQ_W;
        Frag<PQG> w1, w0;
        load_a<PQG, 3>(w1, pw, g * 4 + s, lane);
        load_a<PQG, 3>(w0, pw, (4 + g) * 4 + s, lane);
        const uint16_t* tb = tt + ((s >> 1) * 4 + 2 * (s & 1) + (q >> 1)) * 1024;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + tr_off[0] + 4 * mt));
        s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + tr_off[1] + 4 * mt));
        s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 512 + tr_off[0] + 4 * mt));
        s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 512 + tr_off[1] + 4 * mt));
        s16x8 hh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        s16x8 ll = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        Frag<PQG> ad;
        ad.hi = __builtin_bit_cast(pqg8, hh);
        ad.lo = __builtin_bit_cast(pqg8, ll);
        r.a[s & 1] = PQG::mfma(ad.lo, w1.hi, r.a[s & 1]);
        r.b[s & 1] = PQG::mfma(ad.lo, w0.hi, r.b[s & 1]);
        r.a[s & 1] = PQG::mfma(ad.hi, w1.lo, r.a[s & 1]);
        r.b[s & 1] = PQG::mfma(ad.hi, w0.lo, r.b[s & 1]);
        r.a[s & 1] = PQG::mfma(ad.hi, w1.hi, r.a[s & 1]);
        r.b[s & 1] = PQG::mfma(ad.hi, w0.hi, r.b[s & 1]);
    };
    auto pq_mt_tail = [&](int mt, Pos ps, const float* dy4, f32x4& carry, PqAcc& r) __attribute__((always_inline)) {
        if (!ps.live) return;
        if (ps.top) carry = f32x4{0.f, 0.f, 0.f, 0.f};
        float* out = a.p_out + (size_t)ps.b * a.x_bstride + (size_t)(16 * g + c) * a.pitch + ps.t0 + 16 * mt + 4 * q;
        const int tq = ps.t0 + 16 * mt + 4 * q;
        if (!ps.halo) {
            f32x4 v = (r.a[0] + r.a[1]) + carry;
            if (dy4 != nullptr) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += dy4[i];
            }
            if (ps.t0 + PQ_COLS <= a.t_hi) *reinterpret_cast<f32x4*>(out) = v;      // (t0 >= t_base > t_lo - d always)
            else st4m(out, v, tq, a.t_lo - a.d, a.t_hi);
        }
        carry = r.b[0] + r.b[1];
        if (ps.bot) st4m(out - a.d, carry, tq - a.d, a.t_lo - a.d, a.t_lo);
    };
    auto pq_mt = [&](int stage, int mt, Pos ps, const float* dy4, f32x4& carry) __attribute__((always_inline)) {
        if (!ps.live) return;
        PqAcc r;
        pq_zero(r);
#pragma unroll
        for (int s = 0; s < 4; ++s) pq_mt_step(stage, mt, s, r);
        pq_mt_tail(mt, ps, dy4, carry, r);
    };

    if (wv < 4) {
        // =========================== R waves: recompute, dz, gate ===========================
        struct RawX { f32x2 x[8]; };
        // recompute operands ("time on lanes"; lane (c, q) of N-tile n holds sample t0 + 2c + n): wave g converts
        // k-step g = (tap g>>1, channel half g&1) of x, both N-tiles
        auto load_x = [&](RawX& r, Pos ps) {
            const int tl = ps.t0 + 2 * c;
            const float* xin = a.x_in + (size_t)ps.b * a.x_bstride;
            const float* p = ps.live ? xin + (size_t)(32 * (g & 1) + 8 * q) * a.pitch + ((g >> 1) == 0 ? tl - a.d : tl) : a.x_in;
            const size_t rp = ps.live ? (size_t)a.pitch : 0;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) r.x[jj] = pq_ld2u(p + jj * rp);
        };
        auto fill_x = [&](const RawX& r, int stage) {
            uint16_t* xf = lds + (size_t)stage * PQ_STAGE + PQ_XF;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                Frag<F16> f;
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = r.x[jj][n];
                split8<F16, 3>(f, v);
                pq_store_frag<F16>(xf, g * 2 + n, lane, f);
            }
        };
        Frag<F16> wf[4], wg[4];
        Frag<PQG> wd[2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            load_a<F16, 3>(wf[s], a.wfg, g * 4 + s, lane);
            load_a<F16, 3>(wg[s], a.wfg, (4 + g) * 4 + s, lane);
        }
        if (HAS_DY) {
#pragma unroll
            for (int s = 0; s < 2; ++s) load_a<PQG, 3>(wd[s], a.wdT, g * 2 + s, lane);
        }
        // this lane's dwords in the result tiles: row 4q + i, samples 2c, 2c+1 -> position 8*((2c & 15) >> 2) + 4*(c >> 3) +
        // (2c & 3): chunk (c & 7) >> 1, dword 2*(c >> 3) + (c & 1)
        int t_wr[4];
        {
            const int chk = (c & 7) >> 1, dw = 2 * (c >> 3) + (c & 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) t_wr[i] = (16 * chk + ((4 * q + i) ^ pq_k(chk))) * 8 + dw * 2;
        }
        const long dzx = a.dz_half ? a.dz_half - (long)(CH / 2) * a.pitch : 0;     // second clip of a pair
        auto load_cr = [&](f32x2* cr, Pos ps) {
            const float* dzc = ps.live ? a.dz + (size_t)ps.b * a.dz_bstride + (size_t)(16 * g + 4 * q) * a.pitch + ps.t0 + 2 * c + (g >= 2 ? dzx : 0) : a.dz;
            const size_t rp = ps.live ? (size_t)a.pitch : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) cr[i] = pq_ld2u(dzc + i * rp);
        };

        // CHAIN: this wave's fp32 dy rows of samples 0..15 of an item (the residual term of its half of dx), one item ahead
        f32x4 dyrP = {0.f, 0.f, 0.f, 0.f}, dyrQ = {0.f, 0.f, 0.f, 0.f}, carry = {0.f, 0.f, 0.f, 0.f};
        auto load_dyr = [&](Pos ps) __attribute__((always_inline)) {
            if (CHAIN && HAS_DY) {
                const size_t ro = ps.live ? (size_t)ps.b * a.x_bstride + (size_t)(16 * g + c) * a.pitch + ps.t0 + 4 * q : 0;
                dyrP = PQ_LD4(p_or_x, ro);
                if (QIN) dyrQ = PQ_LD4(q_or_x, ro + (ps.live ? a.dn : 0));
            }
        };
        auto pq_r = [&](int stage, Pos ps) __attribute__((always_inline)) {      // the R waves' share of the data gradient of item ps
            if (CHAIN) {
                float d4[4];
                if (HAS_DY) {
                    const bool fast = interior(ps);
#pragma unroll
                    for (int i = 0; i < 4; ++i) d4[i] = fast ? dyrP[i] + dyrQ[i] : dyv(dyrP[i], dyrQ[i], ps.t0 + 4 * q + i);
                }
                pq_mt(stage, 0, ps, HAS_DY ? d4 : nullptr, carry);
            } else {
                pq_half(stage, 1, ps, nullptr);                 // Q rows
            }
        };

        auto pq_r_step = [&](int stage, int s, PqAcc& r) __attribute__((always_inline)) {
            if (CHAIN) pq_mt_step(stage, 0, s, r);
            else pq_half_step(stage, 1, s, r);
        };
        auto pq_r_tail = [&](Pos ps, PqAcc& r) __attribute__((always_inline)) {
            if (CHAIN) {
                float d4[4];
                if (HAS_DY) {
                    const bool fast = interior(ps);
#pragma unroll
                    for (int i = 0; i < 4; ++i) d4[i] = fast ? dyrP[i] + dyrQ[i] : dyv(dyrP[i], dyrQ[i], ps.t0 + 4 * q + i);
                }
                pq_mt_tail(0, ps, HAS_DY ? d4 : nullptr, carry, r);
            } else {
                pq_half_tail(1, ps, nullptr, r);
            }
        };
        f32x2 crA[4], crB[4];
        load_cr(crA, pos_r(0, 0));
        load_cr(crB, pos_r(0, 1));
        RawX x0, x1;                                        // x1 / x0 hold the raw rows of items it+1 / it+2
        RawD rd;                                            // dy rows (as the pair) of item it+1
        // XD: items the raw x rows are requested ahead.  The chain form that is fed by a pair (the first chain block under a
        // (P, Q) block: 2 launches of 30) holds the dy rows twice more (dyrP, dyrQ) and keeps ONE set of x rows: no spills
        constexpr int XD = (CHAIN && QIN && HAS_DY) ? 1 : 2;
        constexpr bool RMIX = !CHAIN && !COND;
        load_x(x0, pos_r(0, 0));
        load_dy(rd, pos_r(0, 0));
        if (XD == 2) load_x(x1, pos_r(0, 1));
        fill_x(x0, 0);
        fill_dy(rd, pos_r(0, 0), 0);
        load_x(x0, pos_r(0, XD));
        load_dy(rd, pos_r(0, 1));
        __syncthreads();                                    // stage 0 operands of the first item, the weights, the zeros
        // conditioned form with the bucket bytes and <= 32 buckets: the conditioning bias T_b[row][bucket(t)] is one more
        // k-step of the recompute, T_b (this wave's f and g rows, k = bucket, f16 hi + lo) times the 0/1 matrix
        // E[bucket][t] = (bucket(t) == bucket) - four MFMAs per N-tile instead of 16 gathered loads per lane
        Frag<F16> tcf, tcg;
        int tc_b = -1;
        auto load_tab = [&](int b) __attribute__((always_inline)) {
            const float* rf = a.cond + (size_t)b * a.cond_bstride + (size_t)(16 * g + c) * a.cond_pitch;
            const float* rg = rf + (size_t)CH * a.cond_pitch;
            float vf[8], vg[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int bk = 8 * q + jj;
                const int bc = bk < a.cond_le ? bk : 0;
                vf[jj] = bk < a.cond_le ? rf[bc] : 0.f;
                vg[jj] = bk < a.cond_le ? rg[bc] : 0.f;
            }
            split8<F16, 3>(tcf, vf);
            split8<F16, 3>(tcg, vg);
            tc_b = b;
        };
        auto r_body = [&](const int it, f32x2* cr, RawX& rx) {
            if (it >= n_items) {
                // the void item that pads an odd count: nothing of its own to do (its result tiles are never multiplied:
                // the W waves skip a void item's products), only the Q rows of the last real item
                pq_r((it + 1) & 1, pos_r(it, -1));
                win_advance();
                __syncthreads();
                return;
            }
            int cidx[2] = {0, 0};                            // buckets of this lane's two samples (used after the Q rows below)
            if (COND) {
                const uint8_t* ip = a.cond_idx + (WN_PQ_IDX_PAD + pos_r(it, 0).t0 + 2 * c - a.t_lo);
                cidx[0] = ip[0]; cidx[1] = ip[1];
            }
            // (x operands written by the W waves instead - they hold the same rows for the weight gradients; f16 hi/lo [channel][time]
            // tiles read here with transposed reads, no x loads or conversions in the R waves - is correct and SLOWER, round 4: stack
            // 1.89-1.92 vs 1.85-1.87 ms same box; the R waves then wait 14-16 % at the barrier and the W waves set the pace)
            // RMIX (pair form): the k-steps of the R waves' share of the dx product woven into the operand fills (1.81 vs 1.83 ms for
            // the stack, round 4; the chain form, at its register limit, gains nothing: 1.823 vs 1.829 with one set of x rows)
            PqAcc pa;
            if (RMIX) {
                pq_zero(pa);
                pq_r_step((it + 1) & 1, 0, pa);
            }
            fill_x(rx, (it + 1) & 1);                        // recompute operands of the next item
            load_x(rx, pos_r(it, XD + 1));                  // (one item ahead instead of two: no change, 1.970 vs 1.976 ms)
            if (RMIX) pq_r_step((it + 1) & 1, 1, pa);
            // (on the W waves instead - they wait 15 % of their time at the barrier in the chain form, the R waves 3.5 % - the stack
            // got SLOWER, 1.99-2.02 vs 1.93-1.97 ms same box, round 4: their loads of the same rows then queue behind each other)
            fill_dy(rd, pos_r(it, 1), (it + 1) & 1);
            load_dy(rd, pos_r(it, 2));
            if (RMIX) {
                pq_r_step((it + 1) & 1, 2, pa);
                pq_r_step((it + 1) & 1, 3, pa);
                pq_r_tail(pos_r(it, -1), pa);
                load_dyr(pos_r(it, 0));
            }
            if (!RMIX) pq_r((it + 1) & 1, pos_r(it, -1));                 // Q rows (CHAIN: the first half of dx) of the previous item
            if (!RMIX) load_dyr(pos_r(it, 0));
            const Pos p_cur = pos_r(it, 0);
            const bool live = it < n_items;
            const int tl = p_cur.t0 + 2 * c;
            uint16_t* st = lds + (size_t)(it & 1) * PQ_STAGE;
            const uint16_t* xf = st + PQ_XF;
            const uint16_t* dyf = st + PQ_DYF;

            f32x4 af[2], ag[2], dz[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                af[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                ag[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                dz[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            {
                // the three products of an x3 term and the f / g / N-tile accumulators are walked in rotation, the dz
                // products woven in: no MFMA waits for the result of the one in front of it
                auto term = [](auto tr, f32x4& acc, const auto& wa, const auto& xb, int t) {
                    typedef decltype(tr) TT;
                    acc = t == 0 ? TT::mfma(wa.lo, xb.hi, acc) : t == 1 ? TT::mfma(wa.hi, xb.lo, acc) : TT::mfma(wa.hi, xb.hi, acc);
                };
                Frag<F16> bx[2][2];
                Frag<PQG> by[2];
                load_a<F16, 3>(bx[0][0], xf, 0, lane);
                load_a<F16, 3>(bx[0][1], xf, 1, lane);
                if (HAS_DY) load_a<PQG, 3>(by[0], dyf, 0, lane);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks + 1 < 4) {
                        load_a<F16, 3>(bx[(ks + 1) & 1][0], xf, 2 * ks + 2, lane);
                        load_a<F16, 3>(bx[(ks + 1) & 1][1], xf, 2 * ks + 3, lane);
                        if (HAS_DY) load_a<PQG, 3>(by[(ks + 1) & 1], dyf, ks + 1, lane);      // dy fragment (k-step (ks+1)>>1, N-tile (ks+1)&1)
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        term(F16(), af[0], wf[ks], bx[ks & 1][0], t);
                        term(F16(), ag[0], wg[ks], bx[ks & 1][0], t);
                        term(F16(), af[1], wf[ks], bx[ks & 1][1], t);
                        term(F16(), ag[1], wg[ks], bx[ks & 1][1], t);
                        if (HAS_DY) term(PQG(), dz[ks & 1], wd[ks >> 1], by[ks & 1], t);
                    }
                }
            }
            if (COND) {
                if (p_cur.b != tc_b) load_tab(p_cur.b);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int id = cidx[n];
                    const bool mine = (id >> 3) == q;
                    const uint32_t one = 0x3C00u << (16 * (id & 1));
                    u32x4 ev;
#pragma unroll
                    for (int w = 0; w < 4; ++w) ev[w] = (mine && ((id & 7) >> 1) == w) ? one : 0u;
                    const F16::vec8 e = __builtin_bit_cast(F16::vec8, ev);
                    af[n] = F16::mfma(tcf.lo, e, af[n]);
                    ag[n] = F16::mfma(tcg.lo, e, ag[n]);
                    af[n] = F16::mfma(tcf.hi, e, af[n]);
                    ag[n] = F16::mfma(tcg.hi, e, ag[n]);
                }
            }
            uint16_t* tt = st + PQ_T;
            // (mask-free copies of this phase for items inside [t_lo, t_hi) and on one side of z_lo - a fifth of the R waves' vector
            // instructions are compares and selects - change nothing, round 4: 1.858 vs 1.858 ms chain form, 1.897 vs 1.910 pair form)
            const bool ok0 = live && tl >= a.t_lo && tl < a.t_hi, ok1 = live && tl + 1 >= a.t_lo && tl + 1 < a.t_hi;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float vz[2], vf[2], vg[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const bool ok = n ? ok1 : ok0;
                    float gz = dz[n][i];
                    if (tl + n >= a.z_lo && tl + n < a.t_hi) gz += cr[i][n];
                    const WnGateD gd = wn_gate_d(af[n][i], ag[n][i]);          // (wn_common.h: one reciprocal, no cancellation, no overflow)
                    vz[n] = ok ? gd.z : 0.f;
                    vf[n] = ok ? gz * gd.dzdf : 0.f;
                    vg[n] = ok ? gz * gd.dzdg : 0.f;
                }
                // 16-bit hi/lo pairs of (sample 2c, sample 2c+1) -> one dword each in the [channel][time] tiles
                auto put = [&](int kind, const float* v) {
                    uint32_t hi, lo;
                    pq_split2(v[0], v[1], hi, lo);
                    uint16_t* p = tt + (kind * 4 + g) * 1024 + t_wr[i];
                    *reinterpret_cast<uint32_t*>(p) = hi;
                    *reinterpret_cast<uint32_t*>(p + 512) = lo;
                };
                put(0, vf);
                put(1, vg);
                if (HAS_DY) put(2, vz);
            }
            load_cr(cr, pos_r(it, 2));
            win_advance();
            __syncthreads();
        };
        for (int it = 0; it < n_items; it += 2) {
            r_body(it, crA, XD == 2 ? x1 : x0);
            r_body(it + 1, crB, x0);
        }
        {
            const int n_even = (n_items + 1) & ~1;
            if (pos_r(n_even, -1).live) pq_r((n_even - 1) & 1, pos_r(n_even, -1));      // Q rows of the last item
        }
        __syncthreads();                                    // the W waves' extra round (products of the last item)
        return;
    }

    // =========================== W waves: weight gradients and the (P, Q) product ===========================
    f32x4 cfg[8][2], cd[4];
#pragma unroll
    for (int m = 0; m < 8; ++m) { cfg[m][0] = f32x4{0.f, 0.f, 0.f, 0.f}; cfg[m][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int m = 0; m < 4; ++m) cd[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // raw rows of this wave's row tile: lane (row c, q) holds samples t0 + 4q .. + 3 and t0 + 16 + 4q .. + 3
    struct PqU32U { uint32_t v; } __attribute__((packed, aligned(1)));
    struct RawRows { f32x4 x0[2], x1[2], p[2], qq[2]; uint32_t bk[2]; };
    const bool do_c = COND && a.cslab != nullptr;
    // CHAIN: the x(t) rows of an item that continues its chain ARE the x(t - d) rows of the item before it (the chain steps d columns down): only
    // a chain's first item and the workgroup's first item request them (`fresh`); for the others w_body copies the previous item's x0 rows
    // (16-byte loads in this layout touch 16 cache lines per wave: wn_encpq.hip)
    auto load_rows = [&](RawRows& r, Pos ps, bool fresh) {
        const size_t ro = ps.live ? (size_t)ps.b * a.x_bstride + (size_t)(16 * g + c) * a.pitch + ps.t0 + 4 * q : 0;
        const int dd = ps.live ? a.d : 0, dn = ps.live ? a.dn : 0, h = ps.live ? 16 : 0;
        const float* xr = a.x_in + ro;
        r.x0[0] = ld4u(xr - dd); r.x0[1] = ld4u(xr - dd + h);
        if (!CHAIN || fresh) { r.x1[0] = ld4u(xr); r.x1[1] = ld4u(xr + h); }
        if (HAS_DY) {
            r.p[0] = PQ_LD4(p_or_x, ro); r.p[1] = PQ_LD4(p_or_x, ro + h);
            if (QIN) { r.qq[0] = PQ_LD4(q_or_x, ro + dn); r.qq[1] = PQ_LD4(q_or_x, ro + dn + h); }
            else { r.qq[0] = f32x4{0.f, 0.f, 0.f, 0.f}; r.qq[1] = r.qq[0]; }
        }
        if (COND && do_c) {                                 // buckets of the lane's 4 + 4 samples
            const uint8_t* ip = a.cond_idx + (WN_PQ_IDX_PAD + ps.t0 + 4 * q - a.t_lo);
            r.bk[0] = reinterpret_cast<const PqU32U*>(ip)->v;
            r.bk[1] = reinterpret_cast<const PqU32U*>(ip + 16)->v;
        }
    };
    // this wave's B operands of the weight-gradient products (k = the 8 positions of the lane's chunk) and its dy rows
    // in fp32 (the residual term of P)
    struct Ops { Frag<PQG> x0, x1, dy; float dy32[8]; uint32_t bk[2]; };
    auto to_frag = [&](Frag<PQG>& f, const float* w) {
        u32x4 fh, fl;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            uint32_t hi, lo;
            pq_split2(w[2 * jj], w[2 * jj + 1], hi, lo);
            fh[jj] = hi;
            fl[jj] = lo;
        }
        f.hi = __builtin_bit_cast(pqg8, fh);
        f.lo = __builtin_bit_cast(pqg8, fl);
    };
    auto convert = [&](Ops& o, const RawRows& r, Pos ps) {
        float w[8];
        if (COND) { o.bk[0] = r.bk[0]; o.bk[1] = r.bk[1]; }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) w[jj] = ps.live ? r.x0[jj >> 2][jj & 3] : 0.f;
        to_frag(o.x0, w);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) w[jj] = ps.live ? r.x1[jj >> 2][jj & 3] : 0.f;
        to_frag(o.x1, w);
        if (HAS_DY) {
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
        } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o.dy32[jj] = 0.f;
        }
    };
    auto load_tile = [&](Frag<PQG>& f, const uint16_t* base, int tile) {
        const u32x4* p = reinterpret_cast<const u32x4*>(base + tile * 1024 + tile_rd);
        f.hi = __builtin_bit_cast(pqg8, p[0]);
        f.lo = __builtin_bit_cast(pqg8, p[64]);
    };
    // ---- conditioning gradient in the launch (COND, a.cslab): d cond[b][row][bucket] = sum_t [df;dg][row][t] over
    // bucket(t) == bucket is a product with a 0/1 selection matrix S[time][bucket] (exact in 16 bits: two MFMAs per tile,
    // hi S + lo S).  W wave g takes row tiles 2g, 2g+1 of [df;dg]; S is built per item from the buckets of the lane's own 8
    // samples (the k order of the tiles); the sums run over all items of one clip, then go to the workgroup's slot of that
    // clip (plain stores; wn_launch_pq_cond_reduce adds the workgroups in a fixed order: no float atomics, bit-reproducible)
    f32x4 carry_w = {0.f, 0.f, 0.f, 0.f};                  // CHAIN: Q rows of samples 16..31 of the item above
    f32x4 cacc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { cacc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f}; cacc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    int c_b = -1, c_next = 0;
    const int c_first = pos_r(0, 0).b;
    float* const cs_base = do_c ? a.cslab + (size_t)wgid * a.cslab_slots * (2 * CH * 32) : nullptr;
    auto c_store = [&](int slot, bool zero) __attribute__((always_inline)) {
        float* sp = cs_base + (size_t)slot * (2 * CH * 32);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    sp[(size_t)(16 * (2 * g + m) + 4 * q + i) * 32 + 16 * n + c] = zero ? 0.f : cacc[m][n][i];
    };
    auto c_flush = [&](int newb) __attribute__((always_inline)) {      // the sums of clip c_b are complete: write its slot (zeros into skipped ones)
        if (c_b >= 0) {
            const int slot = c_b - c_first;
            for (int s_ = c_next; s_ < slot && s_ < a.cslab_slots; ++s_) c_store(s_, true);
            if (slot < a.cslab_slots) c_store(slot, false);
            c_next = slot + 1;
#pragma unroll
            for (int m = 0; m < 2; ++m) { cacc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f}; cacc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        c_b = newb;
    };
    auto products = [&](int stage, const Ops& o, Pos ps) __attribute__((always_inline)) {
        const uint16_t* tt = lds + (size_t)stage * PQ_STAGE + PQ_T;
        const bool c_item = COND && do_c && ps.live;
        if (COND && c_item && ps.b != c_b) c_flush(ps.b);
        // ---- weight gradients: rows = all [df;dg] / z tiles, columns = this wave's x / dy rows (not for a halo item: the
        // workgroup above counts it)
        if (!(CHAIN && ps.halo)) {
            // the three products of an x3 term are walked across FOUR accumulators (two tiles x two column blocks), so
            // no MFMA waits for the one in front of it; the next pair of tiles is read meanwhile
            auto term = [](f32x4& acc, const Frag<PQG>& wa, const Frag<PQG>& xb, int t) {
                acc = t == 0 ? PQG::mfma(wa.lo, xb.hi, acc) : t == 1 ? PQG::mfma(wa.hi, xb.lo, acc) : PQG::mfma(wa.hi, xb.hi, acc);
            };
            Frag<PQG> am[2][2];
            load_tile(am[0][0], tt, 0);
            load_tile(am[0][1], tt, 1);
#pragma unroll
            for (int mm = 0; mm < (HAS_DY ? 12 : 8); mm += 2) {
                const int cur = (mm >> 1) & 1;
                if (mm + 2 < (HAS_DY ? 12 : 8)) {
                    load_tile(am[cur ^ 1][0], tt, mm + 2);
                    load_tile(am[cur ^ 1][1], tt, mm + 3);
                }
                if (mm < 8) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        term(cfg[mm][0], am[cur][0], o.x0, t);
                        term(cfg[mm][1], am[cur][0], o.x1, t);
                        term(cfg[mm + 1][0], am[cur][1], o.x0, t);
                        term(cfg[mm + 1][1], am[cur][1], o.x1, t);
                    }
                    if (COND && c_item && mm == 2 * g) {          // this wave's two row tiles of the bucket sums
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            const uint32_t key = (uint32_t)(16 * n + c) * 0x01010101u;
                            const uint32_t x0 = o.bk[0] ^ key, x1 = o.bk[1] ^ key;
                            u32x4 sv;
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const uint32_t xx = jj < 2 ? x0 : x1;
                                const uint32_t e0 = (xx >> (16 * (jj & 1))) & 0xFFu, e1 = (xx >> (16 * (jj & 1) + 8)) & 0xFFu;
                                sv[jj] = (e0 == 0 ? (uint32_t)PQ_ONE16 : 0u) | (e1 == 0 ? (uint32_t)PQ_ONE16 << 16 : 0u);
                            }
                            const pqg8 sel = __builtin_bit_cast(pqg8, sv);
                            cacc[0][n] = PQG::mfma(am[cur][0].lo, sel, cacc[0][n]);
                            cacc[1][n] = PQG::mfma(am[cur][1].lo, sel, cacc[1][n]);
                            cacc[0][n] = PQG::mfma(am[cur][0].hi, sel, cacc[0][n]);
                            cacc[1][n] = PQG::mfma(am[cur][1].hi, sel, cacc[1][n]);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        term(cd[mm - 8], am[cur][0], o.dy, t);
                        term(cd[mm - 7], am[cur][1], o.dy, t);
                    }
                }
            }
        }
    };
    // the W waves' share of the data gradient of item ps (result tiles in `stage`); dyk = the lane's fp32 dy rows of that item
    auto pq_w = [&](int stage, const float* dyk, Pos ps) __attribute__((always_inline)) {
        if (CHAIN) pq_mt(stage, 1, ps, HAS_DY ? dyk + 4 : nullptr, carry_w);
        else pq_half(stage, 0, ps, dyk);
    };

    {
        RawRows rr, rr2;                                    // raw rows of items it / it+1: requested two items ahead
        Ops ops;                                            // (one item ahead: 1.968 vs 1.941 ms for the stack; the conditioned
        constexpr int WD = COND ? 1 : 2;                    // form does that and spends the 32 registers on its bucket sums)
        load_rows(rr, pos_r(0, 0), true);
        convert(ops, rr, pos_r(0, -1));                        // "item -1": zeros (its products meet the zeroed tiles of stage 1)
        __syncthreads();
        // the SECOND item's rows only now (round 6): every workgroup of a launch starts at once, and what they all request before their first barrier
        // - 160 KB each, 41 MB per launch - is what that barrier waits 6 us for (the rows it needs arrive with everybody else's prefetch).  These
        // 32 KB are not needed before iteration 1: stack backward 1.910 -> 1.894, 1.921 -> 1.904 ms (profiles/r06_ab_first_requests.json).  Moving the R
        // waves' skip-gradient rows behind their first x / dy rows as well, or delaying this role's first requests, gives the gain back.
        if (WD == 2) load_rows(rr2, pos_r(0, 1), pos_r(0, 1).top);
        // iteration it: products of item it-1 (result tiles of stage (it-1)&1, operands in `ops`); then the raw rows of
        // item it become `ops` and the rows of item it+2 are requested (loop unrolled by two: no register copies)
        // (the conditioned form and the pair-fed chain form have no registers for it; forced on the conditioned form, 12 spilled registers
        // instead of 9: decoder stack backward 2.12 against 2.14 ms at config 4, inside the spread)
        constexpr bool WMID = !COND && !(CHAIN && QIN && HAS_DY);
        auto w_body = [&](const int it, RawRows& rr, RawRows& rn) {
            // (rn: the rows of item it + 1; when that item continues the chain its x(t) rows are this item's x(t - d) rows)
            auto hand_x = [&]() __attribute__((always_inline)) {
                if (CHAIN && WD == 2 && !pos_r(it, 1).top) { rn.x1[0] = rr.x0[0]; rn.x1[1] = rr.x0[1]; }
            };
            products((it + 1) & 1, ops, pos_r(it, -1));
            if (!WMID) {
                pq_w((it + 1) & 1, ops.dy32, pos_r(it, -1));      // rounds 2-4's order: weight gradients, dx product, conversion
                convert(ops, rr, pos_r(it, 0));
                hand_x();
                load_rows(rr, pos_r(it, WD), WD == 1 || pos_r(it, WD).top);
            } else {
                // the row conversion BETWEEN the weight gradients and the dx product: vector work beside the R waves' recompute MFMAs,
                // the product's MFMAs beside their gate phase (the other order pairs matrix with matrix and vector with vector)
                float dyk[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) dyk[jj] = (jj >= (CHAIN ? 4 : 0)) ? ops.dy32[jj] : 0.f;
                convert(ops, rr, pos_r(it, 0));
                hand_x();
                load_rows(rr, pos_r(it, WD), WD == 1 || pos_r(it, WD).top);
                pq_w((it + 1) & 1, dyk, pos_r(it, -1));
            }

            win_advance();
            __syncthreads();
        };
        const int n_even = (n_items + 1) & ~1;
        for (int it = 0; it < n_even; it += 2) { w_body(it, rr, WD == 1 ? rr : rr2); w_body(it + 1, WD == 1 ? rr : rr2, rr); }
        if (pos_r(n_even, -1).live) {                        // the last item, unless it is the void one
            products((n_even - 1) & 1, ops, pos_r(n_even, -1));
            pq_w((n_even - 1) & 1, ops.dy32, pos_r(n_even, -1));
        }
        __syncthreads();
    }
    if (COND && do_c) c_flush(-1);                      // the last clip's sums

    // ---- slab of this workgroup (every workgroup writes one, also an idle one: zeros)
    float* sfg = a.slab_fg + (size_t)wgid * (4 * CH * CH);
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int tap = 0; tap < 2; ++tap)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_nontemporal_store(cfg[m][tap][i], &sfg[(size_t)(16 * m + 4 * q + i) * (2 * CH) + tap * CH + 16 * g + c]);
    if (HAS_DY && a.slab_d) {
        float* sd = a.slab_d + (size_t)wgid * (CH * CH);
#pragma unroll
        for (int m = 0; m < 4; ++m)            // 4 consecutive floats of a row per lane: one 16-byte store
            __builtin_nontemporal_store(cd[m], reinterpret_cast<f32x4*>(&sd[(size_t)(16 * g + c) * CH + 16 * m + 4 * q]));
    }
}

// slots per workgroup of the conditioning-gradient slabs: a workgroup's items are at most (items_per_wg - 1) x (workgroups of its
// XCD) items apart, i.e. span that many / steps_per_clip + 2 clips at most (never more than the batch)
int wn_pq_cond_slots(int t_lo, int t_hi, int batch) {
    if (t_hi <= t_lo || batch <= 0) return 0;
    int t_base, steps, ipw, nwg;
    wn_resrw_plan(t_lo, t_hi, batch, t_base, steps, ipw, nwg);
    const int cnt_max = wn_xcd_swizzle_enabled() ? (nwg >> 3) + 1 : nwg;
    const int s = ((ipw - 1) * cnt_max) / steps + 2;
    return s < batch ? s : batch;
}
int wn_pq_cond_slab_floats(int t_lo, int t_hi, int batch) {
    if (t_hi <= t_lo || batch <= 0) return 0;
    int t_base, steps, ipw, nwg;
    wn_resrw_plan(t_lo, t_hi, batch, t_base, steps, ipw, nwg);
    return nwg * wn_pq_cond_slots(t_lo, t_hi, batch) * (2 * PQ_CH * 32);
}

// out[l][b][row][bucket] = sum over the workgroups w of block launch l of cslab_l[w][b - first_clip(w)][row][bucket], over the
// workgroups whose items reach clip b; first / last clip of w from the same plan as the block kernel's item walk
// (wave-uniform integer work).  A workgroup takes 256 consecutive (row, bucket) elements of one clip of one launch (16 bytes
// per lane); its four waves take every fourth workgroup of an XCD's range (eight loads in flight each) and their partial
// sums are added in a fixed order: bit-reproducible.  Up to PQ_RED_LAYERS launches' slabs per reduce (blockIdx.z), one plan each.
#define PQ_RED_LAYERS 64
struct PqRedPlan { long off[PQ_RED_LAYERS]; int nwg[PQ_RED_LAYERS], ipw[PQ_RED_LAYERS], steps[PQ_RED_LAYERS], slots[PQ_RED_LAYERS]; };
__global__ __launch_bounds__(256) void pq_cond_reduce_k(const float* __restrict__ cslab_all, PqRedPlan pl, int swz, int batch, int le,
                                                        float* __restrict__ out, long out_lstride, long out_bstride, int out_pitch) {
    __shared__ f32x4 part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 256 + lane * 4;                 // four consecutive buckets of one row
    const int b = blockIdx.y, l = blockIdx.z;
    const float* cslab = cslab_all + pl.off[l];
    const int nwg = pl.nwg[l], ipw = pl.ipw[l], steps = pl.steps[l], slots = pl.slots[l];
    const int total = steps * batch;
    const int qn = nwg >> 3, rn = nwg & 7;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    for (int x = 0; x < (swz ? 8 : 1); ++x) {
        const int first = swz ? (x < rn ? x * (qn + 1) : rn * (qn + 1) + (x - rn) * qn) : 0;
        const int cnt = swz ? (x < rn ? qn + 1 : qn) : nwg;
        int i_hi = (first + cnt) * ipw;
        i_hi = i_hi > total ? total : i_hi;
        // clips of this XCD's item range: skip the whole range when clip b is outside
        if (cnt == 0 || first * ipw >= i_hi || b < (first * ipw) / steps || b > (i_hi - 1) / steps) continue;
        // unconditional loads (an invalid term reads the region's first slab and is dropped): eight 16-byte loads in flight
        auto term = [&](int j) {
            const int i_lo = first * ipw + j;
            const bool in = j < cnt && i_lo < i_hi;
            const int i_lc = in ? i_lo : first * ipw;
            const int n_items = (i_hi - i_lc + cnt - 1) / cnt;
            const int b_first = i_lc / steps, b_last = (i_lc + (n_items - 1) * cnt) / steps;
            const int slot = b - b_first;
            const bool ok = in && slot >= 0 && b <= b_last && slot < slots;
            const f32x4 v = *reinterpret_cast<const f32x4*>(cslab + (ok ? ((size_t)(first + j) * slots + slot) * (2 * PQ_CH * 32) : 0) + e);
            return ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        for (int j0 = wave; j0 < cnt; j0 += 32) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = term(j0 + 4 * u);
            acc0 += (v[0] + v[2]) + (v[4] + v[6]);
            acc1 += (v[1] + v[3]) + (v[5] + v[7]);
        }
    }
    part[wave][lane] = acc0 + acc1;
    __syncthreads();
    if (wave == 0) {
        const f32x4 r = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        float* o = out + (size_t)l * out_lstride + (size_t)b * out_bstride + (size_t)(e >> 5) * out_pitch + (e & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((e & 31) + i < le) o[i] = r[i];
    }
}
// n launches' slabs at once: launch l ran with t_lo[l] (host array) and wrote its slabs at cslab + off[l] floats
int wn_launch_pq_cond_reduce(const float* cslab, const long* off, const int* t_lo, int n, int t_hi, int batch, int le, float* out,
                             long out_lstride, long out_bstride, int out_pitch, hipStream_t st) {
    if (batch <= 0) return 0;
    for (int l0 = 0; l0 < n; l0 += PQ_RED_LAYERS) {
        const int nl = n - l0 < PQ_RED_LAYERS ? n - l0 : PQ_RED_LAYERS;
        PqRedPlan pl;
        memset(&pl, 0, sizeof(pl));
        for (int l = 0; l < nl; ++l) {
            if (t_hi <= t_lo[l0 + l]) return wn_set_error_msg(-4, "pq_cond_reduce: empty launch");
            int t_base;
            wn_resrw_plan(t_lo[l0 + l], t_hi, batch, t_base, pl.steps[l], pl.ipw[l], pl.nwg[l]);
            pl.slots[l] = wn_pq_cond_slots(t_lo[l0 + l], t_hi, batch);
            pl.off[l] = off[l0 + l];
        }
        hipLaunchKernelGGL(pq_cond_reduce_k, dim3(2 * PQ_CH * 32 / 256, batch, nl), dim3(256), 0, st, cslab, pl,
                           wn_xcd_swizzle_enabled(), batch, le, out + (size_t)l0 * out_lstride, out_lstride, out_bstride, out_pitch);
        WN_CHECK_LAUNCH();
    }
    return 0;
}

// ---- CHAIN plan.  Items are the 32-column tiles j = 0 .. steps-1 of a clip (from t_base); chain (clip b, residue r) holds the
// items j = r (mod s), s = d / 32, walked from the highest j down: qn + 1 items for r < rm, qn otherwise.  Chain order = clips,
// then residues, then positions.  With fewer than 256 chains every chain is cut into g segments (the first `m mod g` one item
// longer), one workgroup each; otherwise a workgroup takes whole chains.  Gradient bits depend on this plan only.
int wn_pq_chain_ok(int t_lo, int t_hi, int batch, int d) {
    if (d < PQ_COLS || d % PQ_COLS != 0 || t_hi <= t_lo || batch <= 0) return 0;
    const int t_base = t_lo & ~(PQ_COLS - 1);
    const int steps = (t_hi - t_base + PQ_COLS - 1) / PQ_COLS;
    return steps >= d / PQ_COLS;                           // every chain has an item (its tail is what writes dx below t_base)
}
void wn_pq_chain_plan(int t_lo, int t_hi, int batch, int d, int& t_base, int& steps, int& s, int& qn, int& rm, int& g, int& nchain, int& nwg) {
    t_base = t_lo & ~(PQ_COLS - 1);
    steps = (t_hi - t_base + PQ_COLS - 1) / PQ_COLS;
    s = d / PQ_COLS;
    qn = steps / s;
    rm = steps % s;
    nchain = s * batch;
    if (nchain >= 256) { g = 0; nwg = 256; return; }
    g = 256 / nchain;
    const int gmax = qn / 4 > 1 ? qn / 4 : 1;              // segments of at least ~4 items: a halo item costs most of an item
    if (g > gmax) g = gmax;
    nwg = nchain * g;
}
// test hook: the items workgroup `wg` walks, as (clip, t0, flags: 1 halo | 2 top | 4 bottom) triples; returns their number or -1
int wn_pq_chain_items(int t_lo, int t_hi, int batch, int d, int wg, int* out, int cap) {
    if (!wn_pq_chain_ok(t_lo, t_hi, batch, d)) return -1;
    PqChain p;
    int t_base, nwg;
    wn_pq_chain_plan(t_lo, t_hi, batch, d, t_base, p.steps, p.s, p.qn, p.rm, p.g, p.nchain, nwg);
    if (wg < 0 || wg >= nwg) return -1;
    PqCS c;
    int n_real;
    bool halo;
    pq_chain_start(p, wg, nwg, c, n_real, halo);
    const int n = n_real + (halo ? 1 : 0);
    for (int k = 0; k < n && k < cap; ++k) {
        out[3 * k] = c.b;
        out[3 * k + 1] = t_base + PQ_COLS * (c.r + (c.m - 1 - c.pos) * p.s);
        out[3 * k + 2] = ((halo && k == 0) ? 1 : 0) | (c.pos == 0 ? 2 : 0) | (c.pos == c.m - 1 ? 4 : 0);
        c = pq_cs_next(c, p);
    }
    return n;
}
int wn_pq_slabs(int t_lo, int t_hi, int batch, int d, int chain) {
    if (t_hi <= t_lo || batch <= 0) return 0;
    int t_base, steps, ipw, nwg;
    if (chain && wn_pq_chain_ok(t_lo, t_hi, batch, d)) {
        int s, qn, rm, g, nchain;
        wn_pq_chain_plan(t_lo, t_hi, batch, d, t_base, steps, s, qn, rm, g, nchain, nwg);
        return nwg;
    }
    wn_resrw_plan(t_lo, t_hi, batch, t_base, steps, ipw, nwg);
    return nwg;
}

template <bool HAS_DY, bool COND, bool QIN, bool CHAIN>
static void pq_launch(const WnResPqArgs& k, int nwg, size_t sh, hipStream_t st) {
    static WnDevOnce done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done.need(dev)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_bwd_pq_k<HAS_DY, COND, QIN, CHAIN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        done.done(dev);
    }
    hipLaunchKernelGGL((resblock_bwd_pq_k<HAS_DY, COND, QIN, CHAIN>), dim3(nwg), dim3(PQ_THREADS), sh, st, k);
}

int wn_launch_resblock_bwd_pq(const WnResPqArgs& a, int batch, hipStream_t st) {
    if (a.t_hi <= a.t_lo || batch <= 0) return 0;
    WnResPqArgs k = a;
    int nwg;
    k.batch = batch;
    k.swz = wn_xcd_swizzle_enabled();
    if (k.chain) {
        if (k.cond) return wn_set_error_msg(-4, "resblock_bwd_pq: the conditioned block has no chain form");
        if (!wn_pq_chain_ok(a.t_lo, a.t_hi, batch, a.d)) return wn_set_error_msg(-4, "resblock_bwd_pq: chain form needs d % 32 == 0 and an item per chain");
        wn_pq_chain_plan(a.t_lo, a.t_hi, batch, a.d, k.t_base, k.steps_per_clip, k.ch_s, k.ch_qn, k.ch_rm, k.ch_g, k.ch_nchain, nwg);
        k.items_per_wg = 0;
    } else {
        wn_resrw_plan(a.t_lo, a.t_hi, batch, k.t_base, k.steps_per_clip, k.items_per_wg, nwg);      // same items and slabs as wn_resrw.hip
    }
    if (k.cond) {
        if (!k.cond_idx || k.cond_le > 32 || k.cond_le < 1)
            return wn_set_error_msg(-4, "resblock_bwd_pq: a conditioned block needs cond_idx and 1..32 buckets");
        if (a.t_lo - k.t_base > WN_PQ_IDX_PAD) return wn_set_error_msg(-4, "resblock_bwd_pq: item alignment beyond the cond_idx pad");
        k.cslab_slots = k.cslab ? wn_pq_cond_slots(a.t_lo, a.t_hi, batch) : 0;
    } else if (k.cslab) return wn_set_error_msg(-4, "resblock_bwd_pq: cslab without cond");
    const size_t sh = (size_t)PQ_LDS_HALFS * sizeof(uint16_t);
    const bool has_dy = k.p_in != nullptr, qin = k.q_in != nullptr, cnd = k.cond != nullptr, chn = k.chain != 0;
    if (cnd) {
        if (has_dy && !qin) return wn_set_error_msg(-4, "resblock_bwd_pq: a conditioned block takes the (P, Q) pair");
        if (has_dy) pq_launch<true, true, true, false>(k, nwg, sh, st);
        else pq_launch<false, true, true, false>(k, nwg, sh, st);
    } else if (chn) {
        if (!has_dy) pq_launch<false, false, true, true>(k, nwg, sh, st);
        else if (qin) pq_launch<true, false, true, true>(k, nwg, sh, st);
        else pq_launch<true, false, false, true>(k, nwg, sh, st);
    } else {
        if (!has_dy) pq_launch<false, false, true, false>(k, nwg, sh, st);
        else if (qin) pq_launch<true, false, true, false>(k, nwg, sh, st);
        else pq_launch<true, false, false, false>(k, nwg, sh, st);
    }
    WN_CHECK_LAUNCH();
    return 0;
}
