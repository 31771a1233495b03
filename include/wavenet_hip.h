/* libwavenet_hip.so — C ABI of the MI355X (gfx950) WaveNet hot path.
 *
 * The reference (deep-art-project/Music) is pure Python/PyTorch: it has no FFI, plugin table or
 * operator registry for this path.  The drop-in boundary is therefore the
 * torch.nn.Module surface of wavenet/model.py, which music_amd/model.py mirrors; that module calls
 * ONLY the entry points below (through ctypes, music_amd/_lib.py).  Each entry point names the
 * reference code it replaces.  INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Rules of the ABI
 *   - plain C types only: device pointers, sizes, a stream handle (hipStream_t passed as void*);
 *   - returns 0 on success, a negative code on error (wn_last_error() gives the text); no C++
 *     exceptions cross the boundary;
 *   - a call that has work to do checks its REQUIRED pointers first: a NULL one returns -4 and wn_last_error() names the
 *     function and the argument, nothing is launched (pointers documented as optional may be NULL; a call with no work -
 *     zero clips / rows / columns / elements - returns 0 whatever its pointers are);
 *   - no allocation, no synchronisation, no retained pointers: all workspace is the caller's,
 *     every call only enqueues work on `stream` (safe under hipGraph stream capture);
 *   - re-entrant across devices/streams and threads: no global mutable state; the last-error text is
 *     thread-local (wn_last_error() returns the calling thread's);
 *   - the one collective of the path (the data-parallel gradient sum) is `wn_allreduce_flat` on a communicator
 *     the CALLER owns; RCCL is resolved at first use from the process image (or librccl.so), not linked.  The Python host keeps
 *     `torch.distributed.all_reduce(flat_grad)` (music_amd/dist.py): ProcessGroupNCCL owns its communicator and does not hand
 *     the ncclComm_t out, and a second communicator behind its back would duplicate the xGMI rings and the bootstrap for one
 *     5 MB all-reduce per step; a host without torch uses the entry points below.
 *
 * Data layout: activations are float32 [clip][channel][time] with time
 * contiguous, in ABSOLUTE time (column t = index of the newest input sample the value depends
 * on), all with one row pitch (multiple of 4 floats) and 16-byte aligned bases, allocated with
 * >= 64 floats of slack in front of and >= 256 behind the addressed range.  Channel counts are
 * padded to a multiple of 32 with zero weights.  "mode" selects the arithmetic of the
 * channel-mixing products on the v_mfma_f32_16x16x32 matrix cores:
 *   WN_F16X3 / WN_BF16X3 : operands split x = hi + lo in f16 / bf16, 3 MFMAs per product,
 *                          fp32 accumulate (fp32-grade, the default: fwd F16X3, bwd BF16X3)
 *   WN_F16X1 / WN_BF16X1 : plain 16-bit operands, fp32 accumulate.
 */
#ifndef WAVENET_HIP_H
#define WAVENET_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* wn_stream_t;                 /* hipStream_t */
enum { WN_F16X3 = 0, WN_F16X1 = 1, WN_BF16X3 = 2, WN_BF16X1 = 3 };
#define WN_ABI_VERSION 5
#define WN_CE_NUM_PARTIALS 1024

int wn_version(void);
const char* wn_last_error(void);

/* Weight packing: flat fp32 parameter buffer -> MFMA A-fragment order, 16-bit hi(/lo) pieces.
 * idx[p] = offset of the source weight in `flat` (or -1 = structural zero) for logical position
 * p = (fragment, lane, j); n = number of logical positions (multiple of 512).
 * Replaces nothing in the reference (cuDNN consumes nn.Conv1d weights directly). */
int wn_pack_weights(const float* flat, const int32_t* idx, uint16_t* out, int n, int mode,
                    wn_stream_t stream);

/* Generic channel-mixing product over time:
 *   out[b][m][t+out_shift] = resid[b][m][t] (t >= resid_lo) + mask( bias[m]
 *        + sum_k W[m][k]      * pre(in0[b][k][t+shift0])            (k <  32*ks0)
 *        + sum_k W[m][K0 + k] * pre(in1[b][k][t+shift1]) )          (k <  32*ks1)
 * for t in [t_lo, t_hi); pre = relu if relu_in; mask keeps values where mask[b][m][t] > 0.
 * input columns (t+shift) outside [in_lo,in_hi) read as 0 and are never dereferenced.  Replaces: causal nn.Conv1d (wavenet/model.py:104),
 * the skip 1x1 convs + Python sum (model.py:127-134), post_process_1/2 (model.py:136-138) and
 * their autograd backward (data gradients). */
int wn_chan_gemm(const float* in0, const float* in1, int64_t in_bstride, int in_pitch, int in_lo, int in_hi,
                 int shift0, int shift1, int ks0, int ks1, const uint16_t* wpack, int mt, int m_valid,
                 float* out, int64_t out_bstride, int out_pitch, int out_shift, const float* bias,
                 const float* resid, int64_t resid_bstride, int resid_pitch, int resid_lo,
                 const float* mask, int64_t mask_bstride, int mask_pitch,
                 int t_lo, int t_hi, int relu_in, int batch, int mode, wn_stream_t stream);

/* Fused gated residual block, forward (wavenet/model.py:111-129 for one dilation d):
 *   [f;g] = Wfg [x(t-d); x(t)] ; z = tanh f * sigmoid g ; x_out = Wd z + x(t) on [t_lo,t_hi);
 *   z is stored on [z_lo, t_hi) (the crop the skip product needs).  ch = padded channels (32|64).
 * Optional conditioning of the autoencoder's decoder (wavenet_autoencoder/model1.py:175-192,
 * 227-247): [f;g][row][t] += cond[b][row][idx(t)], cond = [B][2*ch][cond_pitch] (rows f then g),
 * idx = (t - t_lo) / cond_q when cond_mode == 1 ("stretch"), (t - t_lo) % cond_le when 2 ("tile");
 * cond == NULL disables it.  Optional, for ch = 64, mode f16x3 and cond_le <= 32: the same table once more as packed
 * A fragments, cond_pack[b] = wn_pack_weights order of the [2*ch rows][K = 32 buckets, zero beyond cond_le] matrix
 * (8 fragments = cond_pack_bstride 8192 halfs per clip), and the buckets as bytes, cond_idx (layout under
 * wn_resblock_bwd_pq) - the bias is then one more k-step of the fg product (table x 0/1 matrix on the matrix cores,
 * hi + lo: the table to ~2^-22) instead of 128 gathered loads per lane; NULL: the gather.
 * z_half_stride != 0 (ch = 64): TWO clips of a model with <= 32 channels side by side on the 64-channel block - x / x_out
 * are the two clips' 32-row tensors back to back (x_bstride = 64 rows), the packs block-diagonal ([W 0; 0 W] per product:
 * the zero blocks cost matrix time, no traffic), and the z rows of the second clip go z_half_stride floats behind the
 * first clip's (its own z slice) instead of 32 rows below them.  Same results as two launches of the 32-channel form. */
int wn_resblock_fwd(const float* x_in, float* x_out, float* z_out, int64_t x_bstride, int64_t z_bstride,
                    int pitch, const uint16_t* wfg, const uint16_t* wd, const float* bias_f,
                    const float* bias_g, const float* bias_d, int n_f, int n_d, int ch, int d,
                    int t_lo, int t_hi, int z_lo, int write_x, const float* cond, int64_t cond_bstride,
                    int cond_pitch, int cond_mode, int cond_le, int cond_q, const uint16_t* cond_pack,
                    int64_t cond_pack_bstride, const uint8_t* cond_idx, int64_t z_half_stride, int batch, int mode,
                    wn_stream_t stream);

/* Fused gated residual block, backward recompute half (autograd of model.py:118-124, SURVEY
 * Appendix B): recomputes f,g,z from x_in; dz = Wd^T dy (+ dz_crop on t >= z_lo);
 * writes dfg = [df; dg] (2*ch rows) and z (ch rows) on [t_lo,t_hi).  dy may be NULL; z may be NULL
 * (the caller kept the forward's z, stored with z_lo = t_lo).
 * cond*: the conditioning table of wn_resblock_fwd (the recompute adds it as well); NULL = none. */
int wn_resblock_bwd(const float* x_in, const float* dy, const float* dz, float* dfg, float* z,
                    int64_t x_bstride, int64_t dz_bstride, int64_t dfg_bstride, int64_t z_bstride,
                    int pitch, const uint16_t* wfg, const uint16_t* wdT, const float* bias_f,
                    const float* bias_g, int n_f, int ch, int d, int t_lo, int t_hi, int z_lo,
                    const float* cond, int64_t cond_bstride, int cond_pitch, int cond_mode, int cond_le, int cond_q,
                    int batch, int mode_fwd, int mode_bwd, wn_stream_t stream);

/* The forward epilogue in ONE launch (ABI v5; replaces the skip 1x1 convs + Python sum, F.relu, post_process_1, F.relu,
 * post_process_2 of wavenet/model.py:127-138 - three wn_chan_gemm launches):  per tile of 128 columns
 *   u = bias_skip + Ws z ;  h = bias_p1 + P1 relu(u) ;  o = bias_p2 + P2 relu(h)        on [t_lo, t_hi)
 * z: the z-crops of all blocks stacked on the channel axis, [B][32 ks_skip][pitch] (ks_skip even), valid on [t_lo, t_hi).  The launch
 * tiles the columns from t_lo & ~63 in steps of 128 and READS every row over whole tiles (values outside [t_lo, t_hi) are ignored, NaN
 * included): the tiles must lie inside the row pitch (-4 otherwise), i.e. the rows need the activation layout's slack.  u, h: [B][256][pitch] (s_bstride floats per clip; rows >= s_valid are not written) -
 * the backward masks with them.  o: compact [B][256][o_pitch], column t - t_lo (the reference's pre-softmax memory order).
 * w_skip: packed [16][ks_skip], natural k.  w_p1c / w_p2c: packed [16][8] in the CHAINED k order (the u / h tile is handed from
 * product to product out of the accumulators, like wn_resblock_fwd's dense weights).  At most 256 skip and 256 quantisation
 * channels; x3 modes.  Biases may be NULL. */
int wn_skip_epilogue_fwd(const float* z, int64_t z_bstride, int pitch, int ks_skip, const uint16_t* w_skip, const float* bias_skip,
                         float* u, float* h, int64_t s_bstride, const uint16_t* w_p1c, const float* bias_p1,
                         const uint16_t* w_p2c, const float* bias_p2, float* o, int64_t o_bstride, int o_pitch,
                         int s_valid, int q_valid, int t_lo, int t_hi, int batch, int mode, wn_stream_t stream);

/* The backward of that epilogue, data gradients, in ONE launch (ABI v5; autograd of wavenet/model.py:127-138 - three
 * wn_chan_gemm launches):  per tile of 128 columns
 *   dh = (P2^T d_o) * [h > 0] ;  du = (P1^T dh) * [u > 0] ;  dz = Ws^T du        on [t_lo, t_hi)
 * d_o: compact [B][256][o_pitch], column t - t_lo (d loss / d pre-softmax).  h, u: the forward's tensors (masks); dh, du: same
 * layout, stored for the weight gradients (wn_wgrad).  dz: [B][16 mt_z][pitch], all blocks' z-crop gradients stacked on the channel
 * axis, mt_z a multiple of 3.  w_p2T: packed P2^T [16][8], natural k; w_p1Tc, w_skipTc: P1^T [16][8] and Ws^T [mt_z][8] in the
 * CHAINED k order.  The mask rows are read unguarded over the tile's 128 columns (workspace rows of the activation layout). */
int wn_skip_epilogue_bwd(const float* d_o, int64_t o_bstride, int o_pitch, const float* h, const float* u, int64_t s_bstride, int pitch,
                         float* d_h, float* d_u, float* d_z, int64_t z_bstride, const uint16_t* w_p2T, const uint16_t* w_p1Tc,
                         const uint16_t* w_skipTc, int mt_z, int z_valid, int s_valid, int t_lo, int t_hi, int batch, int mode,
                         wn_stream_t stream);

/* Encoder block of the autoencoder, forward (wavenet_autoencoder/model1.py:137-152 for one dilation d), one launch:
 *   h = Wdil [relu x(t-d); relu x(t)] (+ bias_dil) ; x_out = Wd relu(h) (+ bias_d) + x(t)   on [t_lo, t_hi);
 *   h (the pre-activation the backward masks with) is stored on the same range.  wdil: packed [ch/16][2ch/32] (natural
 *   k: tap 0 channels then tap 1 channels), wd: packed [ch/16][ch/32] in the chained k order (as wn_resblock_fwd's
 *   dense weights).  ch = padded channels (32|64); n_h / n_d = real dilation / residual channel counts. */
int wn_enc_resblock_fwd(const float* x_in, float* x_out, float* h_out, int64_t x_bstride, int64_t h_bstride, int pitch,
                        const uint16_t* wdil, const uint16_t* wd, const float* bias_dil, const float* bias_d, int n_h,
                        int n_d, int ch, int d, int t_lo, int t_hi, int batch, int mode, wn_stream_t stream);

/* Encoder block of the autoencoder, backward except the data gradient (64 padded channels, bf16x3), one launch:
 *   dh = (Wd^T dy) * [h > 0]  written on [t_lo, t_hi) (dy counts as 0 below y_lo: the top block's gradient only exists on
 *   the pooled crop);  one slab per workgroup: slab_dil[w] = partial dWdil (ch x 2ch, columns = tap0 ch | tap1 ch, the
 *   x operand is relu x), slab_d[w] = partial dWd (ch x ch, rows = dy rows, the other operand is relu h);
 *   wn_enc_resblock_bwd_slabs gives the number of slabs (sum them with wn_reduce_slabs).  The data gradient
 *   dx = [x > 0] (Wdil1^T dh[t] + Wdil0^T dh[t+d]) + dy is a wn_chan_gemm launch on dh.  wdT: packed Wd^T. */
int wn_enc_resblock_bwd(const float* x_in, const float* dy, const float* h, float* dh, int64_t x_bstride,
                        int64_t h_bstride, int64_t dh_bstride, int pitch, const uint16_t* wdT, int ch, int d, int t_lo,
                        int t_hi, int y_lo, float* slab_dil, float* slab_d, int batch, int mode_bwd, wn_stream_t stream);
int wn_enc_resblock_bwd_slabs(int t_lo, int t_hi, int batch);
/* The same backward with the data gradient INSIDE the launch (no biases), handed on as the unshifted pair of
 * wn_resblock_bwd_pq:  in : dy[t] = p_in[t] (t >= p_lo) + q_in[t + dn]  (the pair the block above wrote, dn = ITS dilation,
 * p_lo = ITS t_lo; q_in = NULL: p_in is a plain tensor valid from p_lo - the top block);
 * out: p_out[t] = dy[t] + [x(t) > 0] W1^T dh[t],  q_out[t] = [x(t-d) > 0] W0^T dh[t]  on [t_lo, t_hi), so that
 * dx[s] = p_out[s] + q_out[s + d] (wn_shift_add makes it whole); dh never reaches HBM.  x / P / Q share x_bstride and pitch,
 * q buffers must read as zero beyond t_hi.  wpq: packed [W1^T; W0^T] ([2ch rows][K = ch dh channels], bf16x3); slabs as
 * wn_enc_resblock_bwd.  Autograd of wavenet_autoencoder/model1.py:143-152 for one layer.
 * chain != 0 (d a multiple of 32, wn_resblock_bwd_pq_chain_ok; slabs = wn_resblock_bwd_pq_slabs(.., chain)): the CHAIN form of
 * wn_resblock_bwd_pq - a workgroup walks the items of one residue class downwards in time and carries the Q rows in registers, dx
 * leaves the launch WHOLE in p_out (valid on [t_lo - d, t_hi)), q_out is not touched (may be NULL); the block below takes it as a
 * plain tensor (q_in = NULL, p_lo = this launch's t_lo - d).  4 activation tensors per block instead of 6: the launch is bound by
 * its bytes (profiles/r05_ab_enc_noq.json).
 * chain == 2 (1 <= d < 32; slabs = wn_resblock_bwd_pq_slabs(t_lo, t_hi, batch, 32, 1)): the same hand-over for the small dilations - the
 * workgroups walk ADJACENT items downwards (the chain plan of d = 32) and an item's Q rows reach the dx rows of the same and of the next
 * item through LDS; dx whole in p_out on [t_lo - d, t_hi) as above. */
int wn_enc_resblock_bwd_pq(const float* x_in, const float* p_in, const float* q_in, int dn, int p_lo, const float* h,
                           float* p_out, float* q_out, int64_t x_bstride, int64_t h_bstride, int pitch, const uint16_t* wdT,
                           const uint16_t* wpq, int ch, int d, int t_lo, int t_hi, float* slab_dil, float* slab_d, int chain,
                           int batch, int mode_bwd, wn_stream_t stream);

/* Backward of one residual block with BOTH weight gradients in the launch (channel-split form,
 * 64 padded channels, modes (f16x3, bf16x3)): what wn_resblock_bwd + the two per-layer wn_wgrad calls
 * compute (autograd of wavenet/model.py:111-129 for one layer except the data gradient of the
 * dilated convs, which wn_chan_gemm forms from dfg).  Writes dfg = [df; dg] on [t_lo,t_hi) and one
 * slab per workgroup: slab_fg[w] = partial dWfg (2ch x 2ch, columns = tap0 ch | tap1 ch), slab_d[w] =
 * partial dWd (ch x ch); wn_resblock_bwd_ms_slabs gives the number of slabs (sum them with
 * wn_reduce_slabs).  dy NULL (last block): no dz product, no dWd.  z is not written at all.
 * cond*: the conditioning table of wn_resblock_fwd (the recompute adds it as well); NULL = none.
 * Kernel: the two-role persistent block of wn_resrw.hip (8 waves).  The buffers need the usual slack: rows are
 * read up to 63 columns outside [t_lo - d, t_hi). */
int wn_resblock_bwd_ms(const float* x_in, const float* dy, const float* dz, float* dfg, int64_t x_bstride,
                       int64_t dz_bstride, int64_t dfg_bstride, int pitch, const uint16_t* wfg, const uint16_t* wdT,
                       const float* bias_f, const float* bias_g, int n_f, int ch, int d, int t_lo, int t_hi, int z_lo,
                       float* slab_fg, float* slab_d, const float* cond, int64_t cond_bstride, int cond_pitch,
                       int cond_mode, int cond_le, int cond_q, int batch, int mode_fwd, int mode_bwd,
                       wn_stream_t stream);
int wn_resblock_bwd_ms_slabs(int t_lo, int t_hi, int batch);

/* The whole backward of one residual block in ONE launch, data gradient of the dilated convs included (64 padded
 * channels, modes (f16x3, bf16x3), no biases): what wn_resblock_bwd_ms + the wn_chan_gemm launch on its dfg compute,
 * without [df;dg] ever reaching HBM.  The data gradient travels between blocks as an UNSHIFTED pair:
 *   in : dx_{i+1}[t] = p_in[t] (t >= p_lo) + q_in[t + dn]   (the pair the block above wrote; dn = ITS dilation, p_lo =
 *        ITS t_lo; both NULL for the last block: no dz product, no dWd);
 *   out: p_out[t] = W1^T [df;dg][t] + dx_{i+1}[t],  q_out[t] = W0^T [df;dg][t]  on [t_lo, t_hi), so that
 *        dx_i[t] = p_out[t] + q_out[t + d] (wn_shift_add makes it whole where a plain tensor is needed).
 * x / P / Q share x_bstride and pitch; q buffers must read as zero beyond t_hi (never written there).  wpq: packed
 * [W1^T; W0^T] ([2ch rows][K = df | dg], bf16x3).  Slabs as wn_resblock_bwd_ms (same count: wn_resblock_bwd_ms_slabs).
 * Autograd of wavenet/model.py:111-129 for one layer.
 * The autoencoder's conditioned decoder blocks (wavenet_autoencoder/model1.py:183,227-247): cond != NULL is the table
 * cond[b][2ch rows][cond_le <= 32] whose column bucket(t) is added to [f;g][.][t]; the buckets come as bytes,
 * cond_idx[WN_COND_IDX_PAD + (t - t_lo)] for t in [t_lo, t_hi) with WN_COND_IDX_PAD zero bytes in front and 64 behind
 * (stretch or tile rule: whatever the caller wrote there), and the bias is formed on the matrix cores as one more k-step
 * of the recompute (table x 0/1 matrix).  cslab != NULL (wn_resblock_bwd_pq_cond_floats() floats): the conditioning
 * GRADIENT d cond[b][row][j] = sum of [df;dg][b][row][t] over bucket j is formed inside the launch as well (a 0/1
 * selection product, exact): every workgroup leaves its sums per clip in cslab and wn_resblock_bwd_pq_cond_reduce adds
 * them in a fixed order (no float atomics; the same sums as wn_cond_grad on the [df;dg] wn_resblock_bwd_ms writes, up to
 * summation order; dz_half_stride != 0: two 32-channel clips side by side as under wn_resblock_fwd - the dz-crop rows of the
 * second clip sit that many floats behind the first clip's, the slabs hold the block-diagonal gradients, wn_gather_grads2
 * adds the two copies) - for n_launches block launches in ONE reduce: launch l ran with t_lo[l] (HOST array, as slab_off) and
 * the same t_hi / batch and wrote its slabs at cslab + slab_off[l] floats; out[l][b][2ch rows][cond_le] with the strides
 * given.  More than 32 buckets: wn_resblock_bwd_ms + wn_cond_grad.
 * WHOLE forms (ABI 2).  q_in == NULL with p_in != NULL: the block above handed dx_{i+1} on as ONE tensor, p_in, valid on
 * [p_lo, t_hi).  chain != 0 (unconditioned blocks with d % 32 == 0 and at least d / 32 items of 32 columns per clip:
 * wn_resblock_bwd_pq_chain_ok): the launch walks its 32-column items in chains of stride d downwards in time, so that the Q
 * rows of an item are added to the P rows of the next one in registers - dx_i is written WHOLE to p_out on
 * [t_lo - d, t_hi), q_out is not touched, nothing else changes (same sums per item; the weight-gradient slabs follow the
 * chain plan: wn_resblock_bwd_pq_slabs(…, d, chain) slabs). */
#define WN_COND_IDX_PAD 64
int wn_resblock_bwd_pq(const float* x_in, const float* p_in, const float* q_in, int dn, int p_lo, const float* dz,
                       float* p_out, float* q_out, int64_t x_bstride, int64_t dz_bstride, int pitch, const uint16_t* wfg,
                       const uint16_t* wdT, const uint16_t* wpq, int ch, int d, int t_lo, int t_hi, int z_lo,
                       float* slab_fg, float* slab_d, const float* cond, int64_t cond_bstride, int cond_pitch, int cond_le,
                       const uint8_t* cond_idx, float* cslab, int64_t dz_half_stride, int chain, int batch, int mode_fwd,
                       int mode_bwd, wn_stream_t stream);
int wn_resblock_bwd_pq_chain_ok(int t_lo, int t_hi, int batch, int d);
int wn_resblock_bwd_pq_slabs(int t_lo, int t_hi, int batch, int d, int chain);
/* Host-only view of the chain plan (what tests check it with; nothing is launched): the items workgroup wg of a chain-form launch
 * walks, in order, as triples (clip, first column t0, flags: 1 = halo item - recomputed for its Q rows only, 2 = top of its chain,
 * 4 = bottom) written to out[3 * k ..]; returns the number of items (at most cap are written), -1 when there is no chain form. */
int wn_resblock_bwd_pq_chain_items(int t_lo, int t_hi, int batch, int d, int wg, int* out, int cap);
int wn_resblock_bwd_pq_cond_floats(int t_lo, int t_hi, int batch);
int wn_resblock_bwd_pq_cond_reduce(const float* cslab, const int64_t* slab_off, const int* t_lo, int n_launches, int t_hi, int batch,
                                   int cond_le, float* out, int64_t out_lstride, int64_t out_bstride, int out_pitch,
                                   wn_stream_t stream);
/* ---- the GENERAL path: shapes the specialised kernels do not cover (filter_width != 2, quantization_channels != 256, more
 * than 64 residual / dilation channels; wavenet/model.py:8-15 takes any).  Its channel-mixing products are wn_chan_gemm /
 * wn_wgrad launches (any row count and K, two taps per launch, more taps accumulate through `resid`); these are the
 * elementwise pieces (music_amd/engine_generic.py sequences them).
 * Gate (model.py:120): z[b][r][t] = tanh(fg[b][r][t]) * sigmoid(fg[b][dp + r][t]) for r < rows, t in [t_lo, t_hi);
 * its derivative: dfg[b][r] = dz sigma(g)(1 - tanh^2 f), dfg[b][dp + r] = dz tanh(f) sigma(g)(1 - sigma(g)). */
int wn_gate_fwd(const float* fg, int64_t fg_bstride, int dp, int rows, float* z, int64_t z_bstride, int pitch, int t_lo, int t_hi,
                int batch, wn_stream_t stream);
int wn_gate_bwd(const float* fg, int64_t fg_bstride, int dp, int rows, const float* dz, int64_t dz_bstride, float* dfg,
                int64_t dfg_bstride, int pitch, int t_lo, int t_hi, int batch, wn_stream_t stream);
/* CHUNK softmax for any row length q (model.py:142-144: the contiguous (B, Q, W) buffer viewed (-1, Q)), its
 * backward dx = y (dy - <dy, y>), and the fused softmax + nn.CrossEntropyLoss-on-the-probabilities step of
 * wn_chunk_softmax256_ce (train.py:146,179) for any q; loss_part has WN_CE_NUM_PARTIALS entries, probs / dx may be NULL. */
int wn_chunk_softmax_fwd(const float* x, float* y, int64_t nrows, int q, wn_stream_t stream);
int wn_chunk_softmax_bwd(const float* y, const float* dy, float* dx, int64_t nrows, int q, wn_stream_t stream);
int wn_chunk_softmax_ce(const float* x, const int64_t* target, float* probs, float* dx, float* loss_part, int64_t nrows, int q,
                        float inv_n, wn_stream_t stream);
/* The operand format of the "x3" products: hi[i] = round16(x[i]), lo[i] = round16(x[i] - hi[i]),
 * 16-bit = IEEE half (is_bf16 = 0) or bfloat16 (1), round to nearest even.  The same device function every MFMA operand
 * of the library is split with; no counterpart in the reference (its products are fp32). */
int wn_split16(const float* x, uint16_t* hi, uint16_t* lo, int64_t n, int is_bf16, wn_stream_t stream);
/* out[b][r][t] = p[b][r][t] (t >= p_lo) + q[b][r][t+dn] (t+dn < t_hi), t in [t_lo,t_hi) */
int wn_shift_add(const float* p, const float* q, float* out, int64_t bstride, int pitch, int rows, int dn, int p_lo,
                 int t_lo, int t_hi, int batch, wn_stream_t stream);

/* Weight gradient: C[m][n] = sum_{b, t in [t_lo,t_hi)} A[b][m][t+a_shift] * B_tap[b][n][t+b_shift_tap]
 * C columns [0, 16*nt_per_tap) come from b0, the next 16*nt_per_tap from b1 (if not NULL).
 * The time axis is cut into chunks; workgroup (clip b, chunk j) writes its partial C (leading
 * dimension ldc) with plain stores into slab number b*nchunks + j at c + slab*c_slab_stride.
 * wn_wgrad_slabs() returns the number of slabs a call writes; wn_reduce_slabs() sums them in slab
 * order (bit-reproducible, no float atomics).  Replaces the weight half of autograd's conv
 * backward (wavenet/train.py:181). */
int wn_wgrad(const float* a, int64_t a_bstride, int a_pitch, int a_shift, int a_cols,
             const float* b0, const float* b1, int64_t b_bstride, int b_pitch, int b_shift0,
             int b_shift1, int b_cols, int nt_per_tap, int mt, int relu_b, float* c, int ldc,
             int64_t c_slab_stride, int t_lo, int t_hi, int chunk, int batch, int mode, wn_stream_t stream);
int wn_wgrad_slabs(int t_lo, int t_hi, int chunk, int batch);
/* Weight gradient of the causal layer (autograd of wavenet/model.py:104) when the layer's input is the ONE-HOT tensor
 * wn_onehot built from `codes` (int32 [batch][t]; scrambled as there): dW[r][q][tap] = sum_{b,s} dx[b][r][s] *
 * in[b][q][s-1+tap] becomes a scatter of dx columns - dx (ch rows) is read once, the 4*q*t bytes per clip of dense
 * one-hot are not read at all.  Writes wn_causal_wgrad_codes_slabs(t, batch) slabs [ch][2q] (columns = tap 0 rows |
 * tap 1 rows, the layout wn_wgrad gives the same product); sum them with wn_reduce_slabs.  Bit-reproducible.
 * dx_q != NULL: the data gradient comes as the unshifted pair wn_resblock_bwd_pq writes, dx[s] (s >= p_lo) + dx_q[s + dn].
 * Arbitrary float inputs (faster_audio_data.py hands the model a dense tensor) keep using wn_wgrad. */
int wn_causal_wgrad_codes(const int32_t* codes, int scrambled, const float* dx, const float* dx_q, int dn, int p_lo,
                          int64_t dx_bstride, int pitch, int ch, int q, int t, int batch, float* slab, wn_stream_t stream);
int wn_causal_wgrad_codes_slabs(int t, int batch);
/* Forward of the causal layer (wavenet/model.py:104) from the same codes, without the one-hot tensor:
 * x0[b][r][s] = bias[r] + sum over the ones (q, s-1) of W[r][q][0] + sum over the ones (q, s) of W[r][q][1], s in [1, t).
 * wt = the layer's weight re-laid as [tap][q][ch] floats (ch contiguous, zero-padded to ch); n_rows = real channel count;
 * bias may be NULL.  Sums of exact fp32 weights in a fixed order (bit-reproducible). */
int wn_causal_fwd_codes(const int32_t* codes, int scrambled, const float* wt, const float* bias, int n_rows, float* x0,
                        int64_t x_bstride, int pitch, int ch, int q, int t, int batch, wn_stream_t stream);
/* desc[op] = {vec_start, slab_off, n_slabs, stride, out_off, n} (int64, device memory):
 * out[out_off+e] = sum_s slab[slab_off + s*stride + e] for e < n; work item v covers 4 floats and
 * belongs to the op with vec_start <= v. */
int wn_reduce_slabs(const int64_t* desc, int n_ops, int64_t total_vec, const float* slab, float* out,
                    wn_stream_t stream);

/* out[row] = sum_{b,t} a[b][row][t+a_shift]  (bias gradients, use_bias=true) */
int wn_bias_grad(const float* a, int64_t a_bstride, int a_pitch, int a_shift, int rows, int t_lo,
                 int t_hi, int batch, float* out, wn_stream_t stream);

/* The reference's chunk softmax: rows of 256 CONSECUTIVE floats of the (B,256,W) buffer
 * (wavenet/model.py:142-144). */
int wn_chunk_softmax256_fwd(const float* x, float* y, int64_t nrows, wn_stream_t stream);
int wn_chunk_softmax256_bwd(const float* y, const float* dy, float* dx, int64_t nrows, wn_stream_t stream);
/* Fused chunk softmax + nn.CrossEntropyLoss applied to the PROBABILITIES (wavenet/train.py:146,179;
 * SURVEY Q1) + both backward steps.  probs/dx may be NULL.  loss_part: WN_CE_NUM_PARTIALS floats,
 * their sum is the mean loss.  A target outside [0, 256) - nn.CrossEntropyLoss raises for it - turns the loss and
 * that row's dx into NaN (no silent wrong value, no host round trip). */
int wn_chunk_softmax256_ce(const float* x, const int64_t* target, float* probs, float* dx,
                           float* loss_part, int64_t nrows, float inv_n, wn_stream_t stream);

/* torch.optim.Adam step on a flat buffer (wavenet/train.py:39-42,182); g is multiplied by gscale. */
int wn_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, float bias_corr1, float bias_corr2, float gscale,
                 wn_stream_t stream);
/* torch.optim.SGD(params, lr, momentum) / torch.optim.RMSprop(params, lr, momentum) steps on a flat buffer - the other two
 * optimizers wavenet/train.py:28-38 constructs (every further argument at torch's default: no dampening / Nesterov / weight
 * decay; RMSprop not centered, alpha and eps passed in).  g is multiplied by gscale.  SGD: first_step != 0 sets the momentum
 * buffer to the gradient (torch's first step); momentum == 0 needs no buffer (NULL).  RMSprop: momentum_buf may be NULL when
 * momentum == 0. */
int wn_sgd_flat(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum, float gscale,
                int first_step, wn_stream_t stream);
int wn_rmsprop_flat(float* p, const float* g, float* square_avg, float* momentum_buf, int64_t n, float lr, float alpha,
                    float eps, float momentum, float gscale, wn_stream_t stream);
/* The reference's nn.DataParallel gradient reduction (wavenet/train.py:116-122) as ONE in-place sum over the ranks of the flat
 * fp32 gradient buffer: ncclAllReduce(buf, buf, n, ncclFloat32, ncclSum, comm, stream) on the caller's RCCL communicator
 * (`comm` = an ncclComm_t).  The 1 / world_size of the mean goes into wn_adam_flat's gscale.  Returns -5 when RCCL is neither
 * loaded in the process nor on the loader path (wn_coll_available() == 0), -6 with RCCL's error text when RCCL fails.
 * wn_comm_unique_id / wn_comm_create / wn_comm_destroy are ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy for a host that
 * does not link RCCL itself: rank 0 makes the 128-byte id, every rank gets it by the host's own means and creates its
 * communicator on its current device (collective: returns when all ranks have called). */
int wn_coll_available(void);
int wn_comm_unique_id(char* id128);
int wn_comm_create(int nranks, int rank, const char* id128, void** comm);
int wn_comm_destroy(void* comm);
int wn_allreduce_flat(void* comm, float* buf, int64_t n, wn_stream_t stream);
/* flat_grad[i] = packed[idx[i]] (idx<0 -> 0): dense wgrad results -> state_dict (out,in,k) layout. */
int wn_gather_grads(const float* packed, const int32_t* idx, float* flat_grad, int n, wn_stream_t stream);
/* ... with two sources per element, flat_grad[i] = packed[idx[i]] + packed[idx2[i]] (< 0: nothing): the gradient of a weight
 * that occupies two places of a block-diagonal effective matrix (see z_half_stride of wn_resblock_fwd). */
int wn_gather_grads2(const float* packed, const int32_t* idx, const int32_t* idx2, float* flat_grad, int n, wn_stream_t stream);

/* The conditioning term expanded over time (model1.py:227-247 `_conditon`, both branches): out[b][row][t] =
 * tab[b][row][idx(t)] for t in [t_lo, t_hi), idx as in wn_resblock_fwd (mode 1: (t - t_lo) / q clamped to le - 1, "stretch";
 * mode 2: (t - t_lo) % le, "tile").  What the reference's repeat / index expressions build as a tensor. */
int wn_cond_expand(const float* tab, int64_t tab_bstride, int tab_pitch, int rows, int t_lo, int t_hi, int mode, int le, int q,
                   float* out, int64_t out_bstride, int out_pitch, int batch, wn_stream_t stream);
/* Gradient of the conditioning table: out[b][row][j] = sum over t in [t_lo,t_hi) with idx(t) == j of
 * in[b][row][t]; idx as in wn_resblock_fwd (mode 1 stretch by q, mode 2 tile modulo le)
 * (autograd of wavenet_autoencoder/model1.py:227-247). */
int wn_cond_grad(const float* in, int64_t in_bstride, int in_pitch, int rows, int t_lo, int t_hi, int mode, int le,
                 int q, float* out, int64_t out_bstride, int out_pitch, int batch, wn_stream_t stream);
/* Backward of AvgPool1d: out[b][c][t0 + j*pool + k] = denc[b][c][j] / pool (j < n_out), 0 up to t_hi. */
int wn_avgpool_bwd(const float* denc, int64_t denc_bstride, int denc_pitch, int t0, int pool, int n_out, int rows,
                   float* out, int64_t out_bstride, int out_pitch, int t_hi, int batch, wn_stream_t stream);

/* out[b][c][j] = mean_{k < pool} in[b][c][t0 + j*pool + k], j < n_out  (nn.AvgPool1d,
 * wavenet_autoencoder/model1.py:154-155). */
int wn_avgpool(const float* in, int64_t in_bstride, int in_pitch, int t0, int pool, int n_out, int rows,
               float* out, int64_t out_bstride, int out_pitch, int batch, wn_stream_t stream);

/* One-hot input on device from int32 codes (B,T) -> float32 (B,Q,T).  scrambled=1 reproduces
 * faster_audio_data.one_hot_encode's reshape (wavenet/faster_audio_data.py:77-81);
 * scrambled=0 is the textbook layout fast_generate.py:159-160 builds. */
int wn_onehot(const int32_t* codes, float* out, int batch, int q, int t, int scrambled, wn_stream_t stream);

/* mu-law (wavenet/audio_func.py:5-39): encode through the 255-entry float32 threshold table of the
 * canonical encoder (bit-exact); decode through the 256-entry table. */
int wn_mulaw_encode_tbl(const float* audio, const float* thresholds, uint8_t* codes, int64_t n, wn_stream_t stream);
int wn_mulaw_decode_lut(const uint8_t* codes, const float* table, float* audio, int64_t n, wn_stream_t stream);
/* The same for any number q >= 2 of quantisation channels (the argument of audio_func.py:5,24): q - 1 ascending float32
 * thresholds (code = number of thresholds <= the sample), a q-entry decode table (codes clamped to [0, q)), int32 codes. */
int wn_mulaw_encode_q(const float* audio, const float* thresholds, int q, int32_t* codes, int64_t n, wn_stream_t stream);
int wn_mulaw_decode_q(const int32_t* codes, const float* table, int q, float* audio, int64_t n, wn_stream_t stream);

/* Cached-queue greedy decode, n_steps samples in one persistent launch
 * (wavenet/fast_generate.py:66-141 per sample; :166-172 loop).  All weights fp32 in "decode
 * layout" (music_amd/fast_generate.py builds it): w_causal [R][2Q] (k = tap0 q | tap1 q);
 * per block at w_layers + i*layer_stride: Wfg [2D][2R] (rows f then g; k = tap1 r | tap0 r),
 * Wd [R][D], Ws [S][D]; b_layers per block [bf D | bg D | bd R | bs S] or NULL; w_p1 [S][S],
 * w_p2 [Q][S].  queues: block i's FIFO as a ring [d_i][R] (time-major) at float offset q_off[i];
 * at global step g the column at slot g % d_i is the oldest, is consumed, then overwritten with
 * the block OUTPUT (as written in the reference) or its INPUT (push_input != 0).
 * note0 / prev0: dense [Q] current and previous input columns; forced: teacher-forced next codes
 * (NULL = feed back the argmax).  codes_out[n_steps] = argmax of the probabilities (first index on
 * ties); probs_out optional.  dilations_host / q_off_host are HOST arrays.
 * sync: optional device scratch of (n_layers*D + 2) uint64, used by the two-workgroup matrix-core form
 * (wn_decode_batch_pk; hand-offs through tagged 8-byte granules, the last word is an error flag: non-zero = a
 * bounded spin timed out).  wn_decode / wn_decode_batch run one workgroup of fp32 FMAs per utterance
 * (any channel counts, with or without biases) and ignore it. */
int wn_decode(int n_layers, int R, int D, int S, int Q, const int32_t* dilations_host, const int64_t* q_off_host,
              float* queues, const float* w_causal, const float* b_causal, const float* w_layers,
              int64_t layer_stride, const float* b_layers, const float* w_p1, const float* b_p1,
              const float* w_p2, const float* b_p2, const float* note0, const float* prev0, float* note_out,
              float* prev_out, const int32_t* forced, int32_t* codes_out, float* probs_out, int64_t step0,
              int n_steps, int push_input, uint64_t* sync, wn_stream_t stream);

/* The same loop for n_utt (<= 128) INDEPENDENT utterances side by side in one launch (batched
 * utterances; the reference generates one at a time): weights shared, everything else per utterance with
 * utterance u at queues + u*queues_ustride, note0/prev0/note_out/prev_out + u*Q, forced/codes_out +
 * u*n_steps, probs_out + u*n_steps*Q, sync + u*(n_layers*D + 2).  All utterances share step0 / n_steps.
 * temperature > 0: SAMPLE each code from softmax(logits / temperature) (inverse CDF, uniform numbers from
 * a counter-based generator keyed by (seed, step0 + step, u): reproducible); <= 0: greedy argmax. */
int wn_decode_batch(int n_layers, int R, int D, int S, int Q, const int32_t* dilations_host, const int64_t* q_off_host,
                    float* queues, const float* w_causal, const float* b_causal, const float* w_layers,
                    int64_t layer_stride, const float* b_layers, const float* w_p1, const float* b_p1,
                    const float* w_p2, const float* b_p2, const float* note0, const float* prev0, float* note_out,
                    float* prev_out, const int32_t* forced, int32_t* codes_out, float* probs_out, int64_t step0,
                    int n_steps, int push_input, uint64_t* sync, int n_utt, int64_t queues_ustride, float temperature,
                    uint64_t seed, wn_stream_t stream);
/* wn_decode_batch with the chain's products on the matrix cores: pk = the packed f16 hi/lo weight fragments of the
 * forward blocks (wn_pack_weights, mode WN_F16X3: per block l "fg" at pk + pk_fg0 + l*pk_lstride halfs in natural k
 * order, "d" at pk + pk_d0 + l*pk_lstride in chained k order, as wn_resblock_fwd takes them; pk_skip / pk_p1 / pk_p2 >= 0:
 * the skip product and the two post-processing products as well (S = 256 or 512, Q = 256), else -1).  Used when R = D = 64,
 * S = 256 or 512 and Q = 256 with all of pk given (biases allowed); NULL or other shapes = wn_decode_batch.  A model with
 * FEWER residual / dilation channels (the reference's shipped 32 / 32 / 512) runs here as the 64 / 64 model it is with zero
 * rows and columns: the decoder is bound by latency, not by traffic, so the padding is free (music_amd/fast_generate.py
 * hands over padded weights, packs and 64-wide queue columns).
 * sync here holds wn_decode_sync_granules(n_layers, D, S) uint64 PER UTTERANCE (error flag = the last word of an
 * utterance's region).  Eight utterances share a workgroup pair, one pair of MFMA result columns each (n_utt <= 1024
 * on this path; when fewer than eight are left for a pair the spare columns mirror the last utterance; same arithmetic
 * per utterance, so rows are bit-identical whatever the batch).
 * With 512 skip channels (and room: at most 24 pairs per launch, else the form above) the skip sum and the post-processing of a
 * pair are split over S / 64 workgroups, one 16-row tile per wave, their post-processing tiles register-resident, the
 * S-vectors exchanged as tagged granules through the hand-off area (decode): same sums per row, same codes.
 * Models deeper than 31 blocks keep the tap-0 partial sums of a sample in the pair's hand-off area instead of LDS. */
int64_t wn_decode_sync_granules(int n_layers, int D, int S);
int wn_decode_batch_pk(int n_layers, int R, int D, int S, int Q, const int32_t* dilations_host, const int64_t* q_off_host,
                       float* queues, const float* w_causal, const float* b_causal, const float* w_layers,
                       int64_t layer_stride, const float* b_layers, const float* w_p1, const float* b_p1,
                       const float* w_p2, const float* b_p2, const float* note0, const float* prev0, float* note_out,
                       float* prev_out, const int32_t* forced, int32_t* codes_out, float* probs_out, int64_t step0,
                       int n_steps, int push_input, uint64_t* sync, int n_utt, int64_t queues_ustride, float temperature,
                       uint64_t seed, const uint16_t* pk, int64_t pk_fg0, int64_t pk_d0, int64_t pk_lstride, int64_t pk_skip,
                       int64_t pk_p1, int64_t pk_p2, wn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
