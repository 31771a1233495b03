// Internal host-side launch interface between the C-ABI layer (wn_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// arithmetic modes of the channel-mixing products (see wn_common.h)
enum { WN_MODE_F16X3 = 0, WN_MODE_F16X1 = 1, WN_MODE_BF16X3 = 2, WN_MODE_BF16X1 = 3 };
static inline int wn_mode_is_bf16(int m) { return m >= 2; }
static inline int wn_mode_ns(int m) { return (m & 1) ? 1 : 3; }
static inline size_t wn_frag_halfs(int mode) { return wn_mode_ns(mode) == 3 ? 1024 : 512; }

struct WnGemmArgs {
    const float* in0; const float* in1;   // tap inputs ([K rows][in_pitch]); in1 may be null
    long in_bstride; int in_pitch; int in_lo, in_hi;   // input columns outside [in_lo,in_hi) read as 0
    int shift0, shift1;                    // input column = t + shift
    int ks0, ks1;                          // 32-channel k-steps per tap
    const uint16_t* wpack;                 // [mt][ks0+ks1] packed A fragments
    int mt, m_valid;                       // 16-row tiles / number of real rows
    float* out; long out_bstride; int out_pitch; int out_shift;   // out column = t + out_shift
    const float* bias;                     // [m_valid] or null
    const float* resid; long resid_bstride; int resid_pitch; int resid_lo;   // += resid[row][t] for t >= resid_lo
    const float* mask; long mask_bstride; int mask_pitch;         // keep where mask[row][t] > 0
    int t_lo, t_hi, t_base;                // valid output columns [t_lo, t_hi); t_base set by launcher
    int relu_in;
    int swz;                               // XCD-aware block remap (set by the launcher)
};
int wn_launch_gemm(const WnGemmArgs& a, int batch, int mode, hipStream_t st);
// two-role persistent form of the narrow product (wn_gemm_rw.hip); 1 = launched, 0 = arguments not covered
int wn_launch_gemm_rw(const WnGemmArgs& a, int batch, int mode, hipStream_t st);
// B-stationary persistent form of the wide product for K = 256 and >= 512 rows (wn_gemm_bst.hip); 1 = launched, 0 = not covered
int wn_launch_gemm_bst(const WnGemmArgs& a, int batch, int mode, hipStream_t st);
// fused forward epilogue: skip product over the stacked z-crops -> post_process_1 -> post_process_2 per 128-column tile (wn_epilogue.hip)
struct WnEpiFwdArgs {
    const float* z; long z_bstride; int pitch;             // [B][32 ks_skip rows][pitch] stacked z-crops; u / h share the pitch
    int ks_skip;                                           // 32-row k-steps of the skip product (even)
    const uint16_t* w_skip; const float* bias_s;           // packed [16][ks_skip] (natural k), summed skip bias or null
    float* u; float* h; long s_bstride;                    // out: pre-ReLU skip sum and post_process_1 output, [B][256][pitch]
    const uint16_t* w_p1c; const float* bias_1;            // packed [16][8] in the CHAINED k order
    const uint16_t* w_p2c; const float* bias_2;
    float* o; long o_bstride; int o_pitch;                 // out: compact pre-softmax [B][256][o_pitch], column t - t_lo
    int s_valid, q_valid;                                  // real skip / quantisation rows (<= 256)
    int t_lo, t_hi, t_base, ntx;                           // valid columns; t_base / tiles per clip set by the launcher
    int stagger_n, stagger_cycles;                         // first-round stagger (set by the launcher)
};
int wn_launch_skip_epilogue_fwd(const WnEpiFwdArgs& a, int batch, int mode, hipStream_t st);
// ... and its backward, data gradients: dH = (P2^T dO) [H > 0], dU = (P1^T dH) [U > 0], dZ = Ws^T dU per 128-column tile
struct WnEpiBwdArgs {
    const float* d_o; long o_bstride; int o_pitch;         // compact d loss / d pre-softmax [B][256][o_pitch], column t - t_lo
    const float* h; const float* u; long s_bstride; int pitch;   // the forward's H and U (masks), [B][256][pitch]
    float* d_h; float* d_u;                                // out, same layout (the weight gradients read them)
    float* d_z; long z_bstride;                            // out: [B][16 mt_z][pitch]
    const uint16_t* w_p2T; const uint16_t* w_p1Tc; const uint16_t* w_skipTc;   // packed [16][8] natural, [16][8] chained, [mt_z][8] chained
    int mt_z, z_valid, s_valid;                            // 16-row tiles of dZ (a multiple of 3), real z rows, real skip rows
    int t_lo, t_hi, t_base, ntx;
    int nt_dz;                                             // dZ by streaming (non-temporal) stores
    int n_whole;                                           // tiles done whole; the rest are dealt out by dZ passes (set by the launcher)
};
int wn_launch_skip_epilogue_bwd(const WnEpiBwdArgs& a, int batch, int mode, hipStream_t st);
struct WnResArgs;
int wn_launch_enc_resblock_fwd(const WnResArgs& a, int ch, int batch, int mode, hipStream_t st);   // wn_resblock2.hip (ENC)
int wn_launch_pack(const float* flat, const int32_t* idx, uint16_t* out, int n, int is_bf16, int ns,
                   hipStream_t st);

struct WnResArgs {
    const float* x_in; float* x_out; float* z_out;      // [B][CH][pitch] each (z_out: layer slice)
    long x_bstride, z_bstride; int pitch;
    const uint16_t* wfg; const uint16_t* wd;            // packed [2CH/16][2CH/32] and [CH/16][CH/32] (chained k)
    const float* bias_f; const float* bias_g; const float* bias_d;   // [<=CH] or null
    int n_f, n_d;                                       // real dilation / residual channel counts
    int d, t_lo, t_hi, z_lo, t_base;                    // outputs valid on [t_lo,t_hi); z stored for t >= z_lo
    int write_x;                                        // 0 for the last block (its x is unused)
    // optional conditioning (wavenet_autoencoder/model1.py:183,227-247): [f;g] += cond[b][row][idx(t)]
    const float* cond; long cond_bstride; int cond_pitch;   // [B][2CH][cond_pitch]
    int cond_mode, cond_le, cond_q;                     // 1: idx = (t-t_lo)/cond_q (stretch); 2: idx = (t-t_lo) % cond_le (tile)
    // the same bias on the matrix cores (64 channels, f16x3, cond_le <= 32): cond_pack[b] = the table as 8 packed A fragments
    // ([2CH rows][K = 32 buckets], wn_pack_weights order), cond_idx = bucket bytes as in WnResPqArgs; one more k-step of
    // the fg product against a 0/1 matrix instead of 128 gathered loads per lane
    const uint16_t* cond_pack; long cond_pack_bstride; const uint8_t* cond_idx;
    // 64 channels as TWO 32-channel clips side by side (block-diagonal packs): the z rows of the second clip (32..63) are
    // stored z_half floats behind the first clip's instead of 32 rows below them (0: one 64-row tensor)
    long z_half;
    int swz;
};
int wn_launch_resblock_fwd(const WnResArgs& a, int ch, int batch, int mode, hipStream_t st);
int wn_launch_resblock_fwd_nt(const WnResArgs& a, int ch, int batch, int mode, hipStream_t st);

struct WnResBwdArgs {
    const float* x_in;          // x_i          [B][CH][pitch]
    const float* dy;            // d x_{i+1}    [B][CH][pitch]   (may be null for the last block)
    const float* dz;            // d z-crop     [B][CH][pitch] layer slice (valid t >= z_lo)
    float* dfg;                 // out: [B][2CH][pitch]  rows [0,CH) = d f, [CH,2CH) = d g
    float* z;                   // out: [B][CH][pitch]   recomputed gated activation
    long x_bstride, dz_bstride, dfg_bstride, z_bstride; int pitch;
    const uint16_t* wfg;        // forward fg pack
    const uint16_t* wdT;        // packed Wd^T [CH/16][CH/32] (natural k), grad mode
    const float* bias_f; const float* bias_g; int n_f;
    int d, t_lo, t_hi, z_lo, t_base;
    const float* cond; long cond_bstride; int cond_pitch;   // as in WnResArgs (the recompute adds it too)
    int cond_mode, cond_le, cond_q;
    int swz;
};
int wn_launch_resblock_bwd(const WnResBwdArgs& a, int ch, int batch, int mode_fwd, int mode_bwd,
                           hipStream_t st);

// channel-split backward of one block with both weight gradients in the launch (wn_resms.hip)
struct WnResMsArgs {
    const float* x_in; const float* dy; const float* dz;   // x_i, d x_{i+1} (null for the last block), d z-crop slice
    float* dfg;                                            // out: [B][2CH][pitch]
    long x_bstride, dz_bstride, dfg_bstride; int pitch;
    const uint16_t* wfg; const uint16_t* wdT;              // forward fg pack (f16x3), Wd^T pack (bf16x3)
    const float* bias_f; const float* bias_g; int n_f;
    float* slab_fg; float* slab_d;                         // one slab per workgroup: [2CH][2CH] and [CH][CH]
    int d, t_lo, t_hi, z_lo, t_base;
    int steps_per_clip, items_per_wg, batch;               // set by the launcher
    int swz;
    const float* cond; long cond_bstride; int cond_pitch;  // conditioning table as in WnResArgs (null: none)
    int cond_mode, cond_le, cond_q;
};
int wn_launch_resblock_bwd_ms(const WnResMsArgs& a, int ch, int batch, int mode_fwd, int mode_bwd, hipStream_t st);
int wn_resms_slabs(int t_lo, int t_hi, int batch);
// two-role form of the same block (wn_resrw.hip): 8 waves, 32-column items; same arguments and slab format
int wn_launch_resblock_bwd_rw(const WnResMsArgs& a, int batch, hipStream_t st);
void wn_resrw_plan(int t_lo, int t_hi, int batch, int& t_base, int& steps, int& ipw, int& nwg);
// backward of one block with the data gradient inside, handed on as the unshifted (P, Q) pair (wn_respq.hip)
struct WnResPqArgs {
    const float* x_in;                                     // x_i
    const float* p_in; const float* q_in; int dn, p_lo;    // dx_{i+1}[t] = p_in[t] (t >= p_lo) + q_in[t + dn]; null for the last block
    const float* dz;                                       // d z-crop slice (valid t >= z_lo)
    float* p_out; float* q_out;                            // dx_i[t] = p_out[t] + q_out[t + d], both written on [t_lo, t_hi)
    long x_bstride, dz_bstride; int pitch;                 // x / P / Q share x_bstride and pitch
    const uint16_t* wfg; const uint16_t* wdT; const uint16_t* wpq;   // fg pack (f16x3), Wd^T and [W1^T; W0^T] packs (bf16x3)
    float* slab_fg; float* slab_d;                         // one slab per workgroup: [2CH][2CH] and [CH][CH]
    int d, t_lo, t_hi, z_lo, t_base;
    int steps_per_clip, items_per_wg, batch;               // set by the launcher
    int swz;
    // conditioned form (null: plain): cond[b][2CH rows][cond_le <= 32 buckets] is added to [f;g][row][t] at bucket(t), read as
    // bytes from cond_idx[WN_PQ_IDX_PAD + (t - t_lo)] (zeros in front and 64 behind).  With cslab the conditioning
    // GRADIENT's bucket sums are formed in the launch: workgroup w writes cslab[w][slot][2CH][32] = sum over its items of
    // clip (first clip of w) + slot of [df;dg][row][t] by bucket; wn_launch_pq_cond_reduce adds the workgroups
    const float* cond; long cond_bstride; int cond_pitch; int cond_le;
    const uint8_t* cond_idx; float* cslab; int cslab_slots;
    long dz_half;                                          // two 32-channel clips side by side: dz rows 32..63 sit dz_half floats behind rows 0..31 (0: 64 rows)
    // CHAIN form (d a multiple of 32, every chain non-empty: wn_pq_chain_ok): a workgroup walks the 32-column items of ONE residue
    // class of (item index mod d/32) downwards in time, so the Q half of an item is the carry of the next one and dx_i leaves
    // the launch WHOLE in p_out (valid on [t_lo - d, t_hi)); q_out is not touched.  Set by the launcher (wn_pq_chain_plan):
    int chain, ch_s, ch_qn, ch_rm, ch_g, ch_nchain;        // d/32, items per chain (qn, +1 for the first rm chains), segments per chain (0: whole chains per workgroup), chains
};
#define WN_PQ_IDX_PAD 64
int wn_launch_resblock_bwd_pq(const WnResPqArgs& a, int batch, hipStream_t st);
// chain form: can this launch walk chains (d % 32 == 0, at least one item per chain)?  and its plan / slab count
int wn_pq_chain_ok(int t_lo, int t_hi, int batch, int d);
void wn_pq_chain_plan(int t_lo, int t_hi, int batch, int d, int& t_base, int& steps, int& s, int& qn, int& rm, int& g, int& nchain, int& nwg);
int wn_pq_slabs(int t_lo, int t_hi, int batch, int d, int chain);      // slabs (= workgroups) of one launch
int wn_pq_chain_items(int t_lo, int t_hi, int batch, int d, int wg, int* out, int cap);
int wn_pq_cond_slots(int t_lo, int t_hi, int batch);       // slots per workgroup of WnResPqArgs::cslab
int wn_pq_cond_slab_floats(int t_lo, int t_hi, int batch); // floats of the whole cslab of one launch
int wn_launch_pq_cond_reduce(const float* cslab, const long* off, const int* t_lo, int n, int t_hi, int batch, int le, float* out,
                             long out_lstride, long out_bstride, int out_pitch, hipStream_t st);
int wn_launch_split16(const float* x, uint16_t* hi, uint16_t* lo, long n, int is_bf16, hipStream_t st);
int wn_launch_shift_add(const float* p, const float* q, float* out, long bstride, int pitch, int rows, int dn,
                        int p_lo, int t_lo, int t_hi, int batch, hipStream_t st);
// backward of an autoencoder ENCODER block with both weight gradients (wn_encrw.hip); WnResMsArgs fields as documented there
int wn_launch_enc_bwd_rw(const WnResMsArgs& a, int ch, int batch, int mode_bwd, hipStream_t st);
int wn_enc_bwd_slabs(int t_lo, int t_hi, int batch);
// ... and with the data gradient inside, as the unshifted (P, Q) pair (wn_encpq.hip)
struct WnEncPqArgs {
    const float* x_in;                                     // x_i (the block's input)
    const float* p_in; const float* q_in; int dn, p_lo;    // dy[t] = p_in[t] (t >= p_lo) + q_in[t + dn]; q_in null: a plain tensor
    const float* h; long h_bstride;                        // stored pre-activation of the block
    float* p_out; float* q_out;                            // dx_i[t] = p_out[t] + q_out[t + d], both written on [t_lo, t_hi)
    long x_bstride; int pitch;                             // x / P / Q share x_bstride; every tensor shares pitch
    const uint16_t* wdT; const uint16_t* wpq;              // Wd^T [CH/16][CH/32] and [W1^T; W0^T] [2CH/16][CH/32] packs (bf16x3)
    float* slab_dil; float* slab_d;                        // one slab per workgroup: [CH][2CH] and [CH][CH] (enc_bwd_rw_k's)
    int d, t_lo, t_hi, t_base;
    int steps_per_clip, items_per_wg, batch, swz;          // set by the launcher
    // CHAIN form (WnResPqArgs: d a multiple of 32, every chain non-empty): dx leaves the launch WHOLE in p_out (valid on [t_lo - d, t_hi)),
    // q_out is not touched; the plan fields are set by the launcher (wn_pq_chain_plan)
    int chain, ch_s, ch_qn, ch_rm, ch_g, ch_nchain;
};
int wn_launch_enc_bwd_pq(const WnEncPqArgs& a, int ch, int batch, int mode_bwd, hipStream_t st);

struct WnWgradArgs {
    const float* a; long a_bstride; int a_pitch; int a_shift; int a_cols;   // A: [M rows][time]
    const float* b0; const float* b1; long b_bstride; int b_pitch; int b_shift0, b_shift1; int b_cols;
    int nt_per_tap;             // 16-row tiles of B per tap (C columns = taps * nt_per_tap * 16)
    int mt;                     // 16-row tiles of A
    int relu_b;
    float* c; int ldc;          // slab base: workgroup (b, chunk) writes C[mt*16][ldc] at c + slab*c_slab_stride
    long c_slab_stride;
    int t_lo, t_hi, t_base; int chunk;      // time range and per-WG chunk (multiple of 32)
    int swz;
};
int wn_launch_wgrad(const WnWgradArgs& a, int batch, int mode, hipStream_t st);
int wn_launch_wgrad2(const WnWgradArgs* a1, const WnWgradArgs* a2, int batch, int mode, hipStream_t st);
int wn_wgrad_num_slabs(int t_lo, int t_hi, int chunk, int batch);
int wn_launch_reduce_slabs(const long* desc, int n_ops, long total_vec, const float* slab, float* out, hipStream_t st);

// causal-layer weight gradient from integer codes (wn_causal.hip); one slab [ch][2*256] per workgroup
int wn_launch_causal_wgrad_codes(const int32_t* codes, int scrambled, const float* dx, const float* dxq, int dn, int p_lo,
                                 long dx_bstride, int pitch, int ch, int T, int batch, float* slab, hipStream_t st);
int wn_causal_codes_slabs(int T, int batch);
int wn_launch_causal_fwd_codes(const int32_t* codes, int scrambled, const float* wt, const float* bias, int n_rows, float* x0,
                               long x_bstride, int pitch, int ch, int T, int batch, hipStream_t st);

int wn_launch_softmax_fwd(const float* x, float* y, long nrows, hipStream_t st);
int wn_launch_softmax_bwd(const float* y, const float* dy, float* dx, long nrows, hipStream_t st);
#define WN_CE_PARTIALS 1024
int wn_launch_softmax_ce(const float* x, const int64_t* target, float* probs, float* dx,
                         float* loss_part, long nrows, float inv_n, hipStream_t st);
int wn_launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                   float eps, float bc1, float bc2, float gscale, hipStream_t st);
int wn_launch_sgd(float* p, const float* g, float* buf, long n, float lr, float momentum, float gscale, int first, hipStream_t st);
int wn_launch_rmsprop(float* p, const float* g, float* sq, float* buf, long n, float lr, float alpha, float eps, float momentum,
                      float gscale, hipStream_t st);
int wn_launch_gather_grads(const float* packed, const int32_t* idx, float* flat_grad, int n, hipStream_t st);
int wn_launch_gather_grads2(const float* packed, const int32_t* idx, const int32_t* idx2, float* flat_grad, int n, hipStream_t st);
int wn_launch_onehot(const int32_t* idx, float* out, int batch, int q, int t, int scrambled, hipStream_t st);
int wn_launch_mulaw_encode(const float* audio, const float* thr, uint8_t* codes, long n, hipStream_t st);
int wn_launch_mulaw_decode(const uint8_t* codes, const float* table, float* audio, long n, hipStream_t st);
int wn_launch_mulaw_encode_q(const float* audio, const float* thr, int n_thr, int32_t* codes, long n, hipStream_t st);
int wn_launch_mulaw_decode_q(const int32_t* codes, const float* table, int q, float* audio, long n, hipStream_t st);
int wn_launch_bias_grad(const float* a, long a_bstride, int a_pitch, int a_shift, int rows, int t_lo,
                        int t_hi, int batch, float* out, hipStream_t st);

int wn_launch_cond_expand(const float* tab, long tab_bstride, int tab_pitch, int rows, int t_lo, int t_hi, int mode, int le, int q,
                          float* out, long out_bstride, int out_pitch, int batch, hipStream_t st);
int wn_launch_cond_grad(const float* in, long in_bstride, int in_pitch, int rows, int t_lo, int t_hi, int mode, int le,
                        int q, float* out, long out_bstride, int out_pitch, int batch, hipStream_t st);
int wn_launch_avgpool_bwd(const float* denc, long denc_bstride, int denc_pitch, int t0, int pool, int n_out, int rows,
                          float* out, long out_bstride, int out_pitch, int t_hi, int batch, hipStream_t st);
int wn_launch_avgpool(const float* in, long in_bstride, int in_pitch, int t0, int pool, int n_out, int rows,
                      float* out, long out_bstride, int out_pitch, int batch, hipStream_t st);

// general path (wn_generic.hip): gate and chunk softmax for any shape
int wn_launch_gate_fwd(const float* fg, long fg_bstride, int dp, int rows, float* z, long z_bstride, int pitch, int t_lo, int t_hi,
                       int batch, hipStream_t st);
int wn_launch_gate_bwd(const float* fg, long fg_bstride, int dp, int rows, const float* dz, long dz_bstride, float* dfg,
                       long dfg_bstride, int pitch, int t_lo, int t_hi, int batch, hipStream_t st);
int wn_launch_softmaxq_fwd(const float* x, float* y, long nrows, int q, hipStream_t st);
int wn_launch_softmaxq_bwd(const float* y, const float* dy, float* dx, long nrows, int q, hipStream_t st);
int wn_launch_softmaxq_ce(const float* x, const int64_t* target, float* probs, float* dx, float* loss_part, long nrows, int q,
                          float inv_n, hipStream_t st);

#define WN_DEC_MAX_LAYERS 64
struct WnDecodeArgs {
    int n_layers, R, D, S, Q;
    int dil[WN_DEC_MAX_LAYERS];
    long q_off[WN_DEC_MAX_LAYERS];       // float offset of each block's ring buffer [d][R] inside `queues`
    float* queues;
    const float* w_causal; const float* b_causal;        // [R][2Q] (k = tap0 q | tap1 q), [R]
    const float* w_layers; long layer_stride;            // per block: Wfg [2D][2R] (k = tap1 r | tap0 r), Wd [R][D], Ws [S][D]
    const float* b_layers;                               // per block: [bf D | bg D | bd R | bs S] or null
    const float* b_skip_sum;                             // unused (reserved)
    const float* w_p1; const float* b_p1; const float* w_p2; const float* b_p2;   // [S][S], [Q][S]
    const float* note0; const float* prev0;              // [Q] dense: first input column and the one before it
    float* note_out; float* prev_out;                    // [Q] state handed back
    const int32_t* forced;                               // [n_steps] teacher-forced next codes, or null (greedy)
    int32_t* codes_out; float* probs_out;                // [n_steps], [n_steps][Q] or null
    long step0; int n_steps; int push_input;
    int dbg;                                             // WN_DEC_DBG timing diagnostics (wrong results)
    unsigned long long* sync;                            // hand-off area of the multi-workgroup kernels (sync_ustride granules of 8 B per utterance), or null
    long sync_ustride;                                   // code granule at [stride-2], error flag at [stride-1] of an utterance's region
    // independent utterances decoded side by side (one workgroup, or one pair, each): element strides
    // between utterances of queues / [Q] state vectors / per-step outputs; weights are shared
    int n_utt; long queues_ustride;
    // sampling: sample != 0 draws from softmax(logit * inv_temp) instead of taking the argmax
    int sample; float inv_temp; unsigned long long seed;
    // optional MFMA form of the chain (R = D = 64, no biases): the training engine's packed f16 hi/lo weight fragments
    // ("fg<l>" natural k order, "d<l>" chained k order), offsets in halfs: fragment base of block l = pk + pk_*0 + l * pk_lstride
    const uint16_t* pk; long pk_fg0, pk_d0, pk_lstride;
    long pk_skip, pk_p1, pk_p2;          // "skip" ([S/16][n_layers*D/32]), "p1" ([S/16][S/32]), "p2" ([Q/16][S/32]) fragment bases, natural k order (S = Q = 256)
};
int wn_launch_decode(const WnDecodeArgs& a, hipStream_t st);
long wn_decode_granules(int n_layers, int D, int S);      // 8-byte granules of one utterance's hand-off area (matrix-core kernels)

// wn_coll.hip
int wn_coll_loaded();
int wn_coll_unique_id(char* id128);
int wn_coll_create(int nranks, int rank, const char* id128, void** comm);
int wn_coll_destroy(void* comm);
int wn_coll_allreduce_flat(void* comm, float* buf, int64_t n, hipStream_t st);
