// Wide channel-mixing product (>= 256 rows, one tap, x3 modes) with BOTH operands brought into LDS by LDS-DMA
// (global_load_lds_dwordx4): the skip product over the concatenated z-crops, post_process_1 / 2 (wavenet/model.py:127-138)
// and their "weights transposed" data-gradient products.  Same tile as chan_gemm_wide2_k (wn_gemm.hip): 256 rows x 256
// columns per workgroup, 8 waves = 2 (row halves) x 4 (column groups of 64), time on the MFMA lanes.
//
// What round 5's timing builds said about chan_gemm_wide2_k (profiles/r05_noconv_builds.md): without its MFMAs the forward
// epilogue still takes 77 % of its time, the backward one 86 % - the k-step is bound by its SKELETON (two operand paths
// global -> VGPR -> convert / copy -> ds_write -> barrier -> ds_read, every wave in the same phase at the same time), not by
// the matrix pipe.  Here the skeleton is the one MI355X_MICROARCH / cdna_hip_programming.md section 5 prescribe:
//   * A (packed 16-bit hi / lo weight fragments, L2-resident): 32 x 1 KB LDS-DMA pieces per k-step, straight into the
//     fragment image (the pack IS lane-linear) - no VGPRs, no ds_write; ring of 2 stages, requested one k-step ahead.
//   * B (fp32 activations, HBM): 32 x 1 KB LDS-DMA pieces per k-step of RAW rows - lane (c, q) of piece (column group, j)
//     fetches row 8q + j, columns 4c .. 4c + 3, so the consuming lane later reads ITS OWN 16 bytes back (ds_read_b128,
//     conflict free, no transposition) and splits them in registers; ring of 3 stages, requested two k-steps ahead.  The
//     two row-half waves of a column group both split that group's 32 x 64 slab (2x the vector work of wide2, which
//     shared the split through LDS - and paid 8 ds_write_b64 + 4 ds_write_b128 + 8 VGPR-staged loads per thread and
//     k-step for it); the split runs beside the partner wave's MFMAs.
//   * raw s_barriers behind COUNTED s_waitcnt vmcnt: the newest requests stay in flight across a barrier (a __syncthreads()
//     would drain them).
//   * PING-PONG: the two waves of a SIMD run half a k-step apart (one in its MFMA phase, the other reading / splitting /
//     requesting), two barriers per k-step - see the loop.
// Column groups that touch the input's edge (columns outside [in_lo, in_hi) read as 0) take guarded register loads +
// ds_write into the same pieces; those waves wait vmcnt(0).  LDS: 2 x 32 KB + 3 x 32 KB = all 160 KB, one workgroup per CU
// (as wide2 with its 128 KB).
#include <stdlib.h>
#include <type_traits>
#include "wn_common.h"
#include "wn_kernels.h"

#define GD_A_STAGE (32 * 1024)
#define GD_B_STAGE (32 * 1024)
#define GD_LDS_BYTES (2 * GD_A_STAGE + 3 * GD_B_STAGE)

typedef __attribute__((address_space(3))) void gd_lds_void;
typedef __attribute__((address_space(1))) const void gd_glb_void;
__device__ __forceinline__ void gd_dma16(const void* src_lane, void* lds_wave_base) {
    // 16 bytes per lane: LDS destination = wave-uniform base + lane * 16 (M0), global source per lane
    __builtin_amdgcn_global_load_lds((gd_glb_void*)src_lane, (gd_lds_void*)lds_wave_base, 16, 0, 0);
}

template <class T, bool RELU>
__global__ __launch_bounds__(512) void chan_gemm_dma_k(WnGemmArgs a) {
    constexpr int MTW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char l_s[];      // [2 A stages][3 B stages]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, q = lane >> 4;
    const WnBlock blk = wn_block<true>(a.swz);
    const int b = blk.z;
    const int wm = wave >> 2, wn = wave & 3;
    const int tile0 = a.t_base + blk.x * 256;
    const int t0 = tile0 + wn * 64;
    const int tl = t0 + 4 * c;
    const int mg0 = blk.y * 16;
    const int m0 = mg0 + wm * MTW;
    const int KS = a.ks0;

    f32x4 acc[MTW][4];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        f32x4 init = {0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = (m0 + m) * 16 + 4 * q + i;
                init[i] = row < a.m_valid ? a.bias[row] : 0.f;
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = init;
    }
    // ---- loader role of this wave: column group lg, rows 4 lh .. 4 lh + 3 of every octet (B); pieces 4 wave .. + 3 (A)
    const int lg = wave & 3, lh = wave >> 2;
    const float* in0 = a.in0 + (size_t)b * a.in_bstride;
    const int lcol = tile0 + lg * 64 + 4 * c + a.shift0;
    const int tg0 = tile0 + lg * 64 + a.shift0;
    const bool inner = tg0 >= a.in_lo && tg0 + 64 <= a.in_hi;                 // wave-uniform
    unsigned char* const lb_base = l_s + 2 * GD_A_STAGE;
    auto issue_b = [&](int s, int stage) {
        const float* p = in0 + (size_t)(s * 32 + 8 * q + 4 * lh) * a.in_pitch + lcol;
        unsigned char* dst = lb_base + stage * GD_B_STAGE + (lg * 8 + 4 * lh) * 1024;
        if (inner) {
#pragma unroll
            for (int j = 0; j < 4; ++j) gd_dma16(p + (size_t)j * a.in_pitch, dst + j * 1024);
        } else {        // columns outside [in_lo, in_hi) read as 0 and are never dereferenced
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = ld4g(p + (size_t)j * a.in_pitch, lcol, a.in_lo, a.in_hi);
                *reinterpret_cast<f32x4*>(dst + j * 1024 + lane * 16) = v;
            }
        }
    };
    // row tiles beyond the matrix (the last row group of a 1920-row product) fetch the last real tile: their results are never stored
    int a_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = 4 * wave + i;
        int mm = mg0 + (p >> 1);
        mm = mm < a.mt ? mm : a.mt - 1;
        a_off[i] = (mm * KS) * 2048 + (p & 1) * 1024 + lane * 16;
    }
    const unsigned char* const wp = reinterpret_cast<const unsigned char*>(a.wpack);
    auto issue_a = [&](int s, int stage) {
        unsigned char* dst = l_s + stage * GD_A_STAGE + (4 * wave) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) gd_dma16(wp + (size_t)a_off[i] + (size_t)s * 2048, dst + i * 1024);
    };

#ifndef GD_T
#define GD_T 0
#endif
    // timing builds (-DGD_T=n, wrong results): 1 no MFMAs, 2 no split (raw bits as fragments)
    // ---- phases.  M(s): the 96 MFMAs of k-step s on the fragments in `bf` (A fragments read one tile ahead).  L(k): this wave's
    // 32 x 64 slab of raw activations of k-step k - its own 16 bytes of each of the 8 pieces of column group wn - split into `bf`.
    Frag<T> bf[4];
    auto phase_l = [&](int k) __attribute__((always_inline)) {
        const unsigned char* lb = lb_base + (k % 3) * GD_B_STAGE + (wn * 8) * 1024 + lane * 16;
        f32x4 raw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = *reinterpret_cast<const f32x4*>(lb + j * 1024);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (GD_T == 2) {
                const u32x4 h_ = {__builtin_bit_cast(uint32_t, raw[0][n]), __builtin_bit_cast(uint32_t, raw[1][n]), __builtin_bit_cast(uint32_t, raw[2][n]), __builtin_bit_cast(uint32_t, raw[3][n])};
                const u32x4 l_ = {__builtin_bit_cast(uint32_t, raw[4][n]), __builtin_bit_cast(uint32_t, raw[5][n]), __builtin_bit_cast(uint32_t, raw[6][n]), __builtin_bit_cast(uint32_t, raw[7][n])};
                bf[n].hi = __builtin_bit_cast(typename T::vec8, h_ & 0x3BFF3BFFu);
                bf[n].lo = __builtin_bit_cast(typename T::vec8, l_ & 0x13FF13FFu);
                continue;
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = RELU ? fmaxf(raw[j][n], 0.f) : raw[j][n];
            split8<T, 3>(bf[n], v);
        }
    };
    auto phase_m = [&](int k) __attribute__((always_inline)) {
        const uint16_t* la = reinterpret_cast<const uint16_t*>(l_s + (k & 1) * GD_A_STAGE);
        Frag<T> af[2];
        load_a<T, 3>(af[0], la, wm * MTW, lane);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            if (m + 1 < MTW) load_a<T, 3>(af[(m + 1) & 1], la, wm * MTW + m + 1, lane);
            if (GD_T == 1) {
                asm volatile("" :: "v"(af[m & 1].hi), "v"(af[m & 1].lo), "v"(bf[m & 3].hi), "v"(bf[m & 3].lo));
                continue;
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) mma<T, 3>(acc[m][n], af[m & 1], bf[n]);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // end of a phase: this wave's requests that the NEXT phase's readers need have landed (counted: the newest `keep` groups of
    // 4 pieces stay in flight across the barrier), its LDS reads are done, everybody meets
    auto phase_end = [&](int keep) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        if (!inner || keep == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (keep == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- PING-PONG: the two waves of a SIMD (wave w of row half 0, wave w + 4 of row half 1) run half a k-step apart - one
    // multiplies (phase M) while the other reads, splits and requests (phase L) - so the matrix pipe always has a wave in its MFMA
    // phase and the vector / LDS / request work runs beside it (MI355X_MICROARCH.md "Two waves per SIMD", items 5 and 9; the 8-phase
    // template of cdna_hip_programming.md section 5).  Phases p = -1, 0, 1, ...; EVERY wave ends every phase with phase_end():
    //     p = -1      both halves: L(0)
    //     p = 2s      half 0: M(s)         half 1: L(s)   (s > 0)
    //     p = 2s + 1  half 0: L(s + 1)     half 1: M(s)
    // Requests (each wave its own 4 + 4 pieces): prologue A(0), B(0), A(1), B(1), B(2); at the start of p = 2s (s >= 1): A(s + 1), into
    // the stage half 1 finished reading in p = 2s - 1; at the start of p = 2s + 1: B(s + 3), into the stage half 1 finished reading
    // in p = 2s.  Deadlines: B(k) is first read in p = 2k - 1, A(k) in p = 2k.  In issue order a wave's requests end
    // ... B(s+2) [p = 2s-1], A(s+1) [p = 2s], B(s+3) [p = 2s+1]: the end of an EVEN phase 2s needs B(s+1) and keeps the newest two
    // groups in flight, the end of an ODD phase 2s+1 needs A(s+1) and keeps the newest one.  Once a request is skipped near the
    // end of K the counts no longer hold: vmcnt(0) from there (3 k-steps), and always for edge waves.
    issue_a(0, 0);
    issue_b(0, 0);
    if (KS > 1) issue_a(1, 1);
    if (KS > 1) issue_b(1, 1);
    if (KS > 2) issue_b(2, 2);
    // B(0) and A(0) landed; with KS > 2 the three newest groups (A(1), B(1), B(2)) may stay in flight - but the end of p = -1 must
    // also deliver B(1) (read in p = 1 by half 0 ... deadline p = 0): wait for A(0), B(0) here, the rest at the phase ends
    if (inner && KS > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    phase_l(0);                                                   // p = -1
    phase_end(KS > 3 ? 2 : 0);                                    // keeps B(1)?  no: order is A1, B1, B2 - B(1) must land (p = 1 reads it): keep B(2) only
    if (wm == 0) {
        for (int s = 0; s < KS; ++s) {
            const bool steady = s + 3 < KS;                       // every request of this k-step is issued: the counts hold
            if (s >= 1 && s + 1 < KS) issue_a(s + 1, (s + 1) & 1);    // ---- p = 2s
            phase_m(s);
            phase_end(steady && s >= 1 ? 2 : 0);
            if (s + 3 < KS) issue_b(s + 3, (s + 3) % 3);              // ---- p = 2s + 1
            if (s + 1 < KS) phase_l(s + 1);
            phase_end(steady ? 1 : 0);
        }
    } else {
        for (int s = 0; s < KS; ++s) {
            const bool steady = s + 3 < KS;
            if (s >= 1 && s + 1 < KS) issue_a(s + 1, (s + 1) & 1);    // ---- p = 2s
            if (s > 0) phase_l(s);
            phase_end(steady && s >= 1 ? 2 : 0);
            if (s + 3 < KS) issue_b(s + 3, (s + 3) % 3);              // ---- p = 2s + 1
            phase_m(s);
            phase_end(steady ? 1 : 0);
        }
    }
    if (t0 >= a.t_hi || m0 >= a.mt) return;

    float* out = a.out + (size_t)b * a.out_bstride;
    const float* resid = a.resid ? a.resid + (size_t)b * a.resid_bstride : nullptr;
    const float* mask = a.mask ? a.mask + (size_t)b * a.mask_bstride : nullptr;
    const bool full = tl >= a.t_lo && tl + 3 < a.t_hi;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        if (m0 + m >= a.mt) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = (m0 + m) * 16 + 4 * q + i;
            if (row >= a.m_valid) continue;
            f32x4 v = {acc[m][0][i], acc[m][1][i], acc[m][2][i], acc[m][3][i]};
            if (mask) {
                const float* mp = mask + (size_t)row * a.mask_pitch + tl;
                if (full) {
                    f32x4 mv = ld4u(mp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (tl + e >= a.t_lo && tl + e < a.t_hi) v[e] = mp[e] > 0.f ? v[e] : 0.f;
                }
            }
            if (resid) {
                const float* rp = resid + (size_t)row * a.resid_pitch + tl;
                if (full && tl >= a.resid_lo) {
                    v += ld4u(rp);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (tl + e >= a.resid_lo && tl + e >= a.t_lo && tl + e < a.t_hi) v[e] += rp[e];
                }
            }
            float* op = out + (size_t)row * a.out_pitch + tl + a.out_shift;
            if (full) {
                F4U u = {{v[0], v[1], v[2], v[3]}};
                *reinterpret_cast<F4U*>(op) = u;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (tl + e >= a.t_lo && tl + e < a.t_hi) op[e] = v[e];
            }
        }
    }
}

template <class T, bool RELU>
static void gd_launch(const WnGemmArgs& k, dim3 g, hipStream_t st) {
    static WnDevOnce done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done.need(dev)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chan_gemm_dma_k<T, RELU>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  GD_LDS_BYTES);
        done.done(dev);
    }
    hipLaunchKernelGGL((chan_gemm_dma_k<T, RELU>), g, dim3(512), GD_LDS_BYTES, st, k);
}

// 1 = launched, 0 = arguments not covered (the caller falls back to chan_gemm_wide2_k).  WN_GEMM_DMA=1 turns it on.
int wn_launch_gemm_dma(const WnGemmArgs& k, int batch, int mode, hipStream_t st) {
    const char* e = getenv("WN_GEMM_DMA");                 // (read per launch: a same-process A/B can flip it, tools/gemm_bench.py)
    const int on = (e && e[0] == '1') ? 1 : 0;             // OFF by default: measured no faster alone and slower inside the step (profiles/r05_gemm_dma.md)
    if (!on || k.mt <= 4 || k.in1 != nullptr || k.ks1 != 0 || k.ks0 < 1) return 0;
    if (mode != WN_MODE_F16X3 && mode != WN_MODE_BF16X3) return 0;
    if ((size_t)k.mt * k.ks0 * 2048 + 2048 > 0x7fffffffull) return 0;            // 32-bit fragment offsets
    const int ncol = k.t_hi - k.t_base;
    const dim3 g((ncol + 255) / 256, (k.mt + 15) / 16, batch);
    if (mode == WN_MODE_F16X3) {
        if (k.relu_in) gd_launch<F16, true>(k, g, st); else gd_launch<F16, false>(k, g, st);
    } else {
        if (k.relu_in) gd_launch<BF16, true>(k, g, st); else gd_launch<BF16, false>(k, g, st);
    }
    return 1;
}
