// C ABI of libwavenet_hip.so (declared in include/wavenet_hip.h): argument marshalling only.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/wavenet_hip.h"
#include "wn_common.h"
#include "wn_kernels.h"

static_assert(WN_COND_IDX_PAD == WN_PQ_IDX_PAD, "public and kernel-side pad of the bucket bytes differ");
static thread_local char g_err[512] = "";

int wn_set_error(hipError_t e, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
    return -1;
}
int wn_tile_origin(int t_lo) { return t_lo & ~63; }       // 64 samples = two 128-byte lines of a row (29.7 -> 27.3 us per forward block)

// compute units of the current device (all devices of a node are the same part: asked once)
int wn_num_cus() {
    static std::atomic<int> cus{0};
    int v = cus.load(std::memory_order_relaxed);
    if (v > 0) return v;
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess || pr.multiProcessorCount <= 0) return 256;
    cus.store(pr.multiProcessorCount, std::memory_order_relaxed);
    return pr.multiProcessorCount;
}

int wn_xcd_swizzle_enabled() { return 1; }                 // the XCD-aware block remap is always on (speed only: wn_common.h)

int wn_set_error_msg(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

// Required pointers of a call that has work to do: a NULL one is reported (-4, the argument's name in wn_last_error) before anything is
// launched - the alternative is a memory fault on the device, which takes the caller's process with it.
static int wn_null_error(const char* fn, const char* names, int which) {
    const char* b = names;
    for (int i = 0; i < which && *b; ++b)
        if (*b == ',') ++i;
    while (*b == ' ') ++b;
    int len = 0;
    while (b[len] && b[len] != ',') ++len;
    snprintf(g_err, sizeof(g_err), "%s: argument '%.*s' must not be NULL", fn, len, b);
    return -4;
}
#define WN_REQUIRE(fn, ...)                                                          \
    do {                                                                             \
        const void* wn_req_[] = {__VA_ARGS__};                                       \
        for (int wn_i_ = 0; wn_i_ < (int)(sizeof(wn_req_) / sizeof(wn_req_[0])); ++wn_i_) \
            if (!wn_req_[wn_i_]) return wn_null_error(fn, #__VA_ARGS__, wn_i_);      \
    } while (0)

static_assert(WN_F16X3 == WN_MODE_F16X3 && WN_F16X1 == WN_MODE_F16X1 && WN_BF16X3 == WN_MODE_BF16X3 &&
              WN_BF16X1 == WN_MODE_BF16X1, "mode enums out of sync");
static_assert(WN_CE_NUM_PARTIALS == WN_CE_PARTIALS, "partials out of sync");

extern "C" {

int wn_version(void) { return WN_ABI_VERSION; }
const char* wn_last_error(void) { return g_err; }

int wn_pack_weights(const float* flat, const int32_t* idx, uint16_t* out, int n, int mode, wn_stream_t stream) {
    if (n % 512 != 0) return wn_set_error_msg(-4, "wn_pack_weights: n must be a multiple of 512");
    if (n > 0) WN_REQUIRE("wn_pack_weights", flat, idx, out);
    return wn_launch_pack(flat, idx, out, n, wn_mode_is_bf16(mode), wn_mode_ns(mode), (hipStream_t)stream);
}

int wn_chan_gemm(const float* in0, const float* in1, int64_t in_bstride, int in_pitch, int in_lo, int in_hi,
                 int shift0, int shift1, int ks0, int ks1, const uint16_t* wpack, int mt, int m_valid,
                 float* out, int64_t out_bstride, int out_pitch, int out_shift, const float* bias,
                 const float* resid, int64_t resid_bstride, int resid_pitch, int resid_lo,
                 const float* mask, int64_t mask_bstride, int mask_pitch,
                 int t_lo, int t_hi, int relu_in, int batch, int mode, wn_stream_t stream) {
    if (ks0 <= 0 || (ks1 > 0 && !in1) || mt <= 0) return wn_set_error_msg(-4, "wn_chan_gemm: bad shape");
    if (batch > 0 && t_hi > t_lo) WN_REQUIRE("wn_chan_gemm", in0, wpack, out);
    WnGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.in0 = in0; a.in1 = ks1 > 0 ? in1 : nullptr; a.in_bstride = in_bstride; a.in_pitch = in_pitch; a.in_lo = in_lo; a.in_hi = in_hi;
    a.shift0 = shift0; a.shift1 = shift1; a.ks0 = ks0; a.ks1 = ks1 > 0 ? ks1 : 0; a.wpack = wpack; a.mt = mt; a.m_valid = m_valid;
    a.out = out; a.out_bstride = out_bstride; a.out_pitch = out_pitch; a.out_shift = out_shift; a.bias = bias;
    a.resid = resid; a.resid_bstride = resid_bstride; a.resid_pitch = resid_pitch; a.resid_lo = resid_lo;
    a.mask = mask; a.mask_bstride = mask_bstride; a.mask_pitch = mask_pitch;
    a.t_lo = t_lo; a.t_hi = t_hi; a.relu_in = relu_in;
    return wn_launch_gemm(a, batch, mode, (hipStream_t)stream);
}

int wn_skip_epilogue_fwd(const float* z, int64_t z_bstride, int pitch, int ks_skip, const uint16_t* w_skip, const float* bias_skip,
                         float* u, float* h, int64_t s_bstride, const uint16_t* w_p1c, const float* bias_p1,
                         const uint16_t* w_p2c, const float* bias_p2, float* o, int64_t o_bstride, int o_pitch,
                         int s_valid, int q_valid, int t_lo, int t_hi, int batch, int mode, wn_stream_t stream) {
    if (pitch % 4 != 0) return wn_set_error_msg(-4, "wn_skip_epilogue_fwd: pitch must be a multiple of 4");
    if (s_valid <= 0 || s_valid > 256 || q_valid <= 0 || q_valid > 256)
        return wn_set_error_msg(-4, "wn_skip_epilogue_fwd: at most 256 skip and 256 quantisation channels (wn_chan_gemm covers the rest)");
    if (batch > 0 && t_hi > t_lo) WN_REQUIRE("wn_skip_epilogue_fwd", z, w_skip, u, h, w_p1c, w_p2c, o);
    WnEpiFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.z = z; a.z_bstride = z_bstride; a.pitch = pitch; a.ks_skip = ks_skip; a.w_skip = w_skip; a.bias_s = bias_skip;
    a.u = u; a.h = h; a.s_bstride = s_bstride; a.w_p1c = w_p1c; a.bias_1 = bias_p1; a.w_p2c = w_p2c; a.bias_2 = bias_p2;
    a.o = o; a.o_bstride = o_bstride; a.o_pitch = o_pitch; a.s_valid = s_valid; a.q_valid = q_valid; a.t_lo = t_lo; a.t_hi = t_hi;
    return wn_launch_skip_epilogue_fwd(a, batch, mode, (hipStream_t)stream);
}

int wn_skip_epilogue_bwd(const float* d_o, int64_t o_bstride, int o_pitch, const float* h, const float* u, int64_t s_bstride, int pitch,
                         float* d_h, float* d_u, float* d_z, int64_t z_bstride, const uint16_t* w_p2T, const uint16_t* w_p1Tc,
                         const uint16_t* w_skipTc, int mt_z, int z_valid, int s_valid, int t_lo, int t_hi, int batch, int mode,
                         wn_stream_t stream) {
    if (pitch % 4 != 0) return wn_set_error_msg(-4, "wn_skip_epilogue_bwd: pitch must be a multiple of 4");
    if (s_valid <= 0 || s_valid > 256) return wn_set_error_msg(-4, "wn_skip_epilogue_bwd: at most 256 skip channels");
    if (batch > 0 && t_hi > t_lo) WN_REQUIRE("wn_skip_epilogue_bwd", d_o, h, u, d_h, d_u, d_z, w_p2T, w_p1Tc, w_skipTc);
    WnEpiBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.d_o = d_o; a.o_bstride = o_bstride; a.o_pitch = o_pitch; a.h = h; a.u = u; a.s_bstride = s_bstride; a.pitch = pitch;
    a.d_h = d_h; a.d_u = d_u; a.d_z = d_z; a.z_bstride = z_bstride; a.w_p2T = w_p2T; a.w_p1Tc = w_p1Tc; a.w_skipTc = w_skipTc;
    a.mt_z = mt_z; a.z_valid = z_valid; a.s_valid = s_valid; a.t_lo = t_lo; a.t_hi = t_hi;
    return wn_launch_skip_epilogue_bwd(a, batch, mode, (hipStream_t)stream);
}

int wn_enc_resblock_fwd(const float* x_in, float* x_out, float* h_out, int64_t x_bstride, int64_t h_bstride, int pitch,
                        const uint16_t* wdil, const uint16_t* wd, const float* bias_dil, const float* bias_d, int n_h,
                        int n_d, int ch, int d, int t_lo, int t_hi, int batch, int mode, wn_stream_t stream) {
    if (pitch % 4 != 0) return wn_set_error_msg(-4, "wn_enc_resblock_fwd: pitch must be a multiple of 4");
    if (t_lo < d + 1) return wn_set_error_msg(-4, "wn_enc_resblock_fwd: t_lo must be >= d + 1");
    if (batch > 0 && t_hi > t_lo) WN_REQUIRE("wn_enc_resblock_fwd", x_in, x_out, h_out, wdil, wd);
    WnResArgs a;
    memset(&a, 0, sizeof(a));
    a.x_in = x_in; a.x_out = x_out; a.z_out = h_out; a.x_bstride = x_bstride; a.z_bstride = h_bstride; a.pitch = pitch;
    a.wfg = wdil; a.wd = wd; a.bias_f = bias_dil; a.bias_g = nullptr; a.bias_d = bias_d; a.n_f = n_h; a.n_d = n_d;
    a.d = d; a.t_lo = t_lo; a.t_hi = t_hi; a.z_lo = t_lo; a.write_x = 1;
    return wn_launch_enc_resblock_fwd(a, ch, batch, mode, (hipStream_t)stream);
}

int wn_resblock_fwd(const float* x_in, float* x_out, float* z_out, int64_t x_bstride, int64_t z_bstride,
                    int pitch, const uint16_t* wfg, const uint16_t* wd, const float* bias_f,
                    const float* bias_g, const float* bias_d, int n_f, int n_d, int ch, int d,
                    int t_lo, int t_hi, int z_lo, int write_x, const float* cond, int64_t cond_bstride,
                    int cond_pitch, int cond_mode, int cond_le, int cond_q, const uint16_t* cond_pack,
                    int64_t cond_pack_bstride, const uint8_t* cond_idx, int64_t z_half_stride, int batch, int mode,
                    wn_stream_t stream) {
    if (pitch % 4 != 0) return wn_set_error_msg(-4, "wn_resblock_fwd: pitch must be a multiple of 4");
    if (cond && (cond_le <= 0 || (cond_mode == 1 && cond_q <= 0) || (cond_mode != 1 && cond_mode != 2)))
        return wn_set_error_msg(-4, "wn_resblock_fwd: bad conditioning arguments");
    if (t_lo < d + 1) return wn_set_error_msg(-4, "wn_resblock_fwd: t_lo must be >= d + 1");
    if (batch > 0 && t_hi > t_lo) {
        WN_REQUIRE("wn_resblock_fwd", x_in, wfg, wd, z_out);
        if (write_x) WN_REQUIRE("wn_resblock_fwd", x_out);
    }
    WnResArgs a;
    memset(&a, 0, sizeof(a));
    a.x_in = x_in; a.x_out = x_out; a.z_out = z_out; a.x_bstride = x_bstride; a.z_bstride = z_bstride; a.pitch = pitch;
    a.wfg = wfg; a.wd = wd; a.bias_f = bias_f; a.bias_g = bias_g; a.bias_d = bias_d; a.n_f = n_f; a.n_d = n_d;
    a.d = d; a.t_lo = t_lo; a.t_hi = t_hi; a.z_lo = z_lo; a.write_x = write_x;
    a.cond = cond; a.cond_bstride = cond_bstride; a.cond_pitch = cond_pitch; a.cond_mode = cond_mode;
    a.cond_le = cond_le; a.cond_q = cond_q;
    if (cond && cond_pack && cond_idx) { a.cond_pack = cond_pack; a.cond_pack_bstride = cond_pack_bstride; a.cond_idx = cond_idx; }
    if (z_half_stride && ch != 64) return wn_set_error_msg(-4, "wn_resblock_fwd: z_half_stride is for two 32-channel clips on the 64-channel block");
    a.z_half = z_half_stride;
    return wn_launch_resblock_fwd(a, ch, batch, mode, (hipStream_t)stream);
}

int wn_resblock_bwd(const float* x_in, const float* dy, const float* dz, float* dfg, float* z,
                    int64_t x_bstride, int64_t dz_bstride, int64_t dfg_bstride, int64_t z_bstride,
                    int pitch, const uint16_t* wfg, const uint16_t* wdT, const float* bias_f,
                    const float* bias_g, int n_f, int ch, int d, int t_lo, int t_hi, int z_lo,
                    const float* cond, int64_t cond_bstride, int cond_pitch, int cond_mode, int cond_le, int cond_q,
                    int batch, int mode_fwd, int mode_bwd, wn_stream_t stream) {
    if (pitch % 4 != 0) return wn_set_error_msg(-4, "wn_resblock_bwd: pitch must be a multiple of 4");
    if (batch > 0 && t_hi > t_lo) {
        WN_REQUIRE("wn_resblock_bwd", x_in, dz, dfg, wfg);
        if (dy) WN_REQUIRE("wn_resblock_bwd", wdT);
    }
    WnResBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x_in = x_in; a.dy = dy; a.dz = dz; a.dfg = dfg; a.z = z;
    a.x_bstride = x_bstride; a.dz_bstride = dz_bstride; a.dfg_bstride = dfg_bstride; a.z_bstride = z_bstride; a.pitch = pitch;
    a.wfg = wfg; a.wdT = wdT; a.bias_f = bias_f; a.bias_g = bias_g; a.n_f = n_f;
    a.d = d; a.t_lo = t_lo; a.t_hi = t_hi; a.z_lo = z_lo;
    a.cond = cond; a.cond_bstride = cond_bstride; a.cond_pitch = cond_pitch; a.cond_mode = cond_mode;
    a.cond_le = cond_le; a.cond_q = cond_q;
    return wn_launch_resblock_bwd(a, ch, batch, mode_fwd, mode_bwd, (hipStream_t)stream);
}

int wn_wgrad(const float* a_, int64_t a_bstride, int a_pitch, int a_shift, int a_cols,
             const float* b0, const float* b1, int64_t b_bstride, int b_pitch, int b_shift0,
             int b_shift1, int b_cols, int nt_per_tap, int mt, int relu_b, float* c, int ldc,
             int64_t c_slab_stride, int t_lo, int t_hi, int chunk, int batch, int mode, wn_stream_t stream) {
    if (c_slab_stride < (int64_t)mt * 16 * ldc) return wn_set_error_msg(-4, "wn_wgrad: slab stride smaller than C");
    if (batch > 0 && t_hi > t_lo) WN_REQUIRE("wn_wgrad", a_, b0, c);
    WnWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.a = a_; a.a_bstride = a_bstride; a.a_pitch = a_pitch; a.a_shift = a_shift; a.a_cols = a_cols;
    a.b0 = b0; a.b1 = b1; a.b_bstride = b_bstride; a.b_pitch = b_pitch; a.b_shift0 = b_shift0; a.b_shift1 = b_shift1; a.b_cols = b_cols;
    a.nt_per_tap = nt_per_tap; a.mt = mt; a.relu_b = relu_b; a.c = c; a.ldc = ldc; a.c_slab_stride = c_slab_stride;
    a.t_lo = t_lo; a.t_hi = t_hi; a.chunk = chunk;
    return wn_launch_wgrad(a, batch, mode, (hipStream_t)stream);
}

int wn_wgrad_slabs(int t_lo, int t_hi, int chunk, int batch) { return wn_wgrad_num_slabs(t_lo, t_hi, chunk, batch); }

int wn_reduce_slabs(const int64_t* desc, int n_ops, int64_t total_vec, const float* slab, float* out, wn_stream_t stream) {
    if (n_ops > 0 && total_vec > 0) WN_REQUIRE("wn_reduce_slabs", desc, slab, out);
    return wn_launch_reduce_slabs(reinterpret_cast<const long*>(desc), n_ops, total_vec, slab, out, (hipStream_t)stream);
}

int wn_bias_grad(const float* a, int64_t a_bstride, int a_pitch, int a_shift, int rows, int t_lo,
                 int t_hi, int batch, float* out, wn_stream_t stream) {
    if (batch > 0 && t_hi > t_lo && rows > 0) WN_REQUIRE("wn_bias_grad", a, out);
    return wn_launch_bias_grad(a, a_bstride, a_pitch, a_shift, rows, t_lo, t_hi, batch, out, (hipStream_t)stream);
}

int wn_chunk_softmax256_fwd(const float* x, float* y, int64_t nrows, wn_stream_t stream) {
    if (nrows > 0) WN_REQUIRE("wn_chunk_softmax256_fwd", x, y);
    return wn_launch_softmax_fwd(x, y, nrows, (hipStream_t)stream);
}
int wn_chunk_softmax256_bwd(const float* y, const float* dy, float* dx, int64_t nrows, wn_stream_t stream) {
    if (nrows > 0) WN_REQUIRE("wn_chunk_softmax256_bwd", y, dy, dx);
    return wn_launch_softmax_bwd(y, dy, dx, nrows, (hipStream_t)stream);
}
int wn_chunk_softmax256_ce(const float* x, const int64_t* target, float* probs, float* dx,
                           float* loss_part, int64_t nrows, float inv_n, wn_stream_t stream) {
    if (nrows > 0) WN_REQUIRE("wn_chunk_softmax256_ce", x, target);
    return wn_launch_softmax_ce(x, target, probs, dx, loss_part, nrows, inv_n, (hipStream_t)stream);
}
int wn_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, float bias_corr1, float bias_corr2, float gscale, wn_stream_t stream) {
    if (n > 0) WN_REQUIRE("wn_adam_flat", p, g, m, v);
    return wn_launch_adam(p, g, m, v, n, lr, beta1, beta2, eps, bias_corr1, bias_corr2, gscale, (hipStream_t)stream);
}
int wn_sgd_flat(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum, float gscale, int first_step,
                wn_stream_t stream) {
    if (momentum != 0.f && !momentum_buf) return wn_set_error_msg(-4, "wn_sgd_flat: momentum needs its buffer");
    if (n > 0) WN_REQUIRE("wn_sgd_flat", p, g);
    return wn_launch_sgd(p, g, momentum_buf, n, lr, momentum, gscale, first_step, (hipStream_t)stream);
}
int wn_rmsprop_flat(float* p, const float* g, float* square_avg, float* momentum_buf, int64_t n, float lr, float alpha, float eps,
                    float momentum, float gscale, wn_stream_t stream) {
    if (!square_avg || (momentum > 0.f && !momentum_buf)) return wn_set_error_msg(-4, "wn_rmsprop_flat: missing state buffer");
    if (n > 0) WN_REQUIRE("wn_rmsprop_flat", p, g);
    return wn_launch_rmsprop(p, g, square_avg, momentum_buf, n, lr, alpha, eps, momentum, gscale, (hipStream_t)stream);
}
int wn_coll_available(void) { return wn_coll_loaded(); }
int wn_comm_unique_id(char* id128) { return wn_coll_unique_id(id128); }
int wn_comm_create(int nranks, int rank, const char* id128, void** comm) { return wn_coll_create(nranks, rank, id128, comm); }
int wn_comm_destroy(void* comm) { return wn_coll_destroy(comm); }
int wn_allreduce_flat(void* comm, float* buf, int64_t n, wn_stream_t stream) {
    return wn_coll_allreduce_flat(comm, buf, n, (hipStream_t)stream);
}
int wn_gather_grads(const float* packed, const int32_t* idx, float* flat_grad, int n, wn_stream_t stream) {
    if (n > 0) WN_REQUIRE("wn_gather_grads", packed, idx, flat_grad);
    return wn_launch_gather_grads(packed, idx, flat_grad, n, (hipStream_t)stream);
}
int wn_gather_grads2(const float* packed, const int32_t* idx, const int32_t* idx2, float* flat_grad, int n, wn_stream_t stream) {
    if (!packed || !idx || !idx2 || !flat_grad) return wn_set_error_msg(-4, "wn_gather_grads2: null argument");
    return wn_launch_gather_grads2(packed, idx, idx2, flat_grad, n, (hipStream_t)stream);
}
int wn_onehot(const int32_t* codes, float* out, int batch, int q, int t, int scrambled, wn_stream_t stream) {
    if (batch > 0 && q > 0 && t > 0) WN_REQUIRE("wn_onehot", codes, out);
    return wn_launch_onehot(codes, out, batch, q, t, scrambled, (hipStream_t)stream);
}
int wn_mulaw_encode_tbl(const float* audio, const float* thresholds, uint8_t* codes, int64_t n, wn_stream_t stream) {
    if (n > 0) WN_REQUIRE("wn_mulaw_encode_tbl", audio, thresholds, codes);
    return wn_launch_mulaw_encode(audio, thresholds, codes, n, (hipStream_t)stream);
}
int wn_mulaw_decode_lut(const uint8_t* codes, const float* table, float* audio, int64_t n, wn_stream_t stream) {
    if (n > 0) WN_REQUIRE("wn_mulaw_decode_lut", codes, table, audio);
    return wn_launch_mulaw_decode(codes, table, audio, n, (hipStream_t)stream);
}
int wn_mulaw_encode_q(const float* audio, const float* thresholds, int q, int32_t* codes, int64_t n, wn_stream_t stream) {
    if (q < 2 || (n > 0 && (!audio || !thresholds || !codes))) return wn_set_error_msg(-4, "wn_mulaw_encode_q: bad argument");
    return wn_launch_mulaw_encode_q(audio, thresholds, q - 1, codes, n, (hipStream_t)stream);
}
int wn_mulaw_decode_q(const int32_t* codes, const float* table, int q, float* audio, int64_t n, wn_stream_t stream) {
    if (q < 2 || (n > 0 && (!audio || !table || !codes))) return wn_set_error_msg(-4, "wn_mulaw_decode_q: bad argument");
    return wn_launch_mulaw_decode_q(codes, table, q, audio, n, (hipStream_t)stream);
}

int wn_resblock_bwd_ms(const float* x_in, const float* dy, const float* dz, float* dfg, int64_t x_bstride,
                       int64_t dz_bstride, int64_t dfg_bstride, int pitch, const uint16_t* wfg, const uint16_t* wdT,
                       const float* bias_f, const float* bias_g, int n_f, int ch, int d, int t_lo, int t_hi, int z_lo,
                       float* slab_fg, float* slab_d, const float* cond, int64_t cond_bstride, int cond_pitch,
                       int cond_mode, int cond_le, int cond_q, int batch, int mode_fwd, int mode_bwd,
                       wn_stream_t stream) {
    if (pitch % 4 != 0) return wn_set_error_msg(-4, "wn_resblock_bwd_ms: pitch must be a multiple of 4");
    if (cond && (cond_le <= 0 || (cond_mode == 1 && cond_q <= 0) || (cond_mode != 1 && cond_mode != 2)))
        return wn_set_error_msg(-4, "wn_resblock_bwd_ms: bad conditioning arguments");
    if (!slab_fg) return wn_set_error_msg(-4, "wn_resblock_bwd_ms: slab_fg is required");
    if (batch > 0 && t_hi > t_lo) {
        WN_REQUIRE("wn_resblock_bwd_ms", x_in, dz, dfg, wfg);
        if (dy) WN_REQUIRE("wn_resblock_bwd_ms", wdT, slab_d);
    }
    WnResMsArgs a;
    memset(&a, 0, sizeof(a));
    a.x_in = x_in; a.dy = dy; a.dz = dz; a.dfg = dfg; a.x_bstride = x_bstride; a.dz_bstride = dz_bstride;
    a.dfg_bstride = dfg_bstride; a.pitch = pitch; a.wfg = wfg; a.wdT = wdT; a.bias_f = bias_f; a.bias_g = bias_g; a.n_f = n_f;
    a.slab_fg = slab_fg; a.slab_d = dy ? slab_d : nullptr; a.d = d; a.t_lo = t_lo; a.t_hi = t_hi; a.z_lo = z_lo;
    a.cond = cond; a.cond_bstride = cond_bstride; a.cond_pitch = cond_pitch; a.cond_mode = cond_mode;
    a.cond_le = cond_le; a.cond_q = cond_q;
    return wn_launch_resblock_bwd_ms(a, ch, batch, mode_fwd, mode_bwd, (hipStream_t)stream);
}
int wn_enc_resblock_bwd(const float* x_in, const float* dy, const float* h, float* dh, int64_t x_bstride,
                        int64_t h_bstride, int64_t dh_bstride, int pitch, const uint16_t* wdT, int ch, int d, int t_lo,
                        int t_hi, int y_lo, float* slab_dil, float* slab_d, int batch, int mode_bwd, wn_stream_t stream) {
    if (pitch % 4 != 0) return wn_set_error_msg(-4, "wn_enc_resblock_bwd: pitch must be a multiple of 4");
    if (!x_in || !dy || !h || !dh || !wdT || !slab_dil || !slab_d) return wn_set_error_msg(-4, "wn_enc_resblock_bwd: null argument");
    WnResMsArgs a;
    memset(&a, 0, sizeof(a));
    a.x_in = x_in; a.dy = dy; a.dz = h; a.dfg = dh; a.x_bstride = x_bstride; a.dz_bstride = h_bstride;
    a.dfg_bstride = dh_bstride; a.pitch = pitch; a.wdT = wdT; a.slab_fg = slab_dil; a.slab_d = slab_d;
    a.d = d; a.t_lo = t_lo; a.t_hi = t_hi; a.z_lo = y_lo < t_lo ? t_lo : y_lo;
    return wn_launch_enc_bwd_rw(a, ch, batch, mode_bwd, (hipStream_t)stream);
}
int wn_enc_resblock_bwd_slabs(int t_lo, int t_hi, int batch) { return wn_enc_bwd_slabs(t_lo, t_hi, batch); }
int wn_enc_resblock_bwd_pq(const float* x_in, const float* p_in, const float* q_in, int dn, int p_lo, const float* h,
                           float* p_out, float* q_out, int64_t x_bstride, int64_t h_bstride, int pitch, const uint16_t* wdT,
                           const uint16_t* wpq, int ch, int d, int t_lo, int t_hi, float* slab_dil, float* slab_d, int chain, int batch,
                           int mode_bwd, wn_stream_t stream) {
    if (pitch % 4 != 0) return wn_set_error_msg(-4, "wn_enc_resblock_bwd_pq: pitch must be a multiple of 4");
    if (!x_in || !p_in || !h || !p_out || (!q_out && !chain) || !wdT || !wpq || !slab_dil || !slab_d)
        return wn_set_error_msg(-4, "wn_enc_resblock_bwd_pq: null argument");
    WnEncPqArgs a;
    memset(&a, 0, sizeof(a));
    a.x_in = x_in; a.p_in = p_in; a.q_in = q_in; a.dn = q_in ? dn : 0; a.p_lo = p_lo; a.h = h; a.h_bstride = h_bstride;
    a.p_out = p_out; a.q_out = q_out; a.x_bstride = x_bstride; a.pitch = pitch; a.wdT = wdT; a.wpq = wpq;
    a.slab_dil = slab_dil; a.slab_d = slab_d; a.d = d; a.t_lo = t_lo; a.t_hi = t_hi; a.chain = chain == 2 ? 2 : chain ? 1 : 0;
    return wn_launch_enc_bwd_pq(a, ch, batch, mode_bwd, (hipStream_t)stream);
}
int wn_resblock_bwd_ms_slabs(int t_lo, int t_hi, int batch) { return wn_resms_slabs(t_lo, t_hi, batch); }

int wn_resblock_bwd_pq(const float* x_in, const float* p_in, const float* q_in, int dn, int p_lo, const float* dz,
                       float* p_out, float* q_out, int64_t x_bstride, int64_t dz_bstride, int pitch, const uint16_t* wfg,
                       const uint16_t* wdT, const uint16_t* wpq, int ch, int d, int t_lo, int t_hi, int z_lo,
                       float* slab_fg, float* slab_d, const float* cond, int64_t cond_bstride, int cond_pitch, int cond_le,
                       const uint8_t* cond_idx, float* cslab, int64_t dz_half_stride, int chain, int batch, int mode_fwd,
                       int mode_bwd, wn_stream_t stream) {
    if (pitch % 4 != 0) return wn_set_error_msg(-4, "wn_resblock_bwd_pq: pitch must be a multiple of 4");
    if (cond && (!cond_idx || cond_le < 1 || cond_le > 32))
        return wn_set_error_msg(-4, "wn_resblock_bwd_pq: a conditioned block needs cond_idx and 1..32 buckets");
    if (cslab && !cond) return wn_set_error_msg(-4, "wn_resblock_bwd_pq: cslab without cond");
    if (ch != 64) return wn_set_error_msg(-3, "wn_resblock_bwd_pq: 64 padded channels only");
    if (mode_fwd != WN_MODE_F16X3 || mode_bwd != WN_MODE_BF16X3) return wn_set_error_msg(-2, "wn_resblock_bwd_pq: (f16x3, bf16x3) only");
    if (!x_in || !dz || !p_out || (!q_out && !chain) || !wfg || !wpq || !slab_fg) return wn_set_error_msg(-4, "wn_resblock_bwd_pq: null argument");
    if ((p_in && !wdT) || (!p_in && q_in)) return wn_set_error_msg(-4, "wn_resblock_bwd_pq: p_in and wdT go together, q_in needs p_in");
    if (chain && (cond || dz_half_stride < 0)) return wn_set_error_msg(-4, "wn_resblock_bwd_pq: no chain form of a conditioned block");
    WnResPqArgs a;
    memset(&a, 0, sizeof(a));
    a.x_in = x_in; a.p_in = p_in; a.q_in = q_in; a.dn = dn; a.p_lo = p_lo; a.dz = dz; a.p_out = p_out; a.q_out = q_out;
    a.x_bstride = x_bstride; a.dz_bstride = dz_bstride; a.pitch = pitch; a.wfg = wfg; a.wdT = wdT; a.wpq = wpq;
    a.slab_fg = slab_fg; a.slab_d = p_in ? slab_d : nullptr; a.d = d; a.t_lo = t_lo; a.t_hi = t_hi; a.z_lo = z_lo;
    a.cond = cond; a.cond_bstride = cond_bstride; a.cond_pitch = cond_pitch; a.cond_le = cond_le;
    a.cond_idx = cond ? cond_idx : nullptr; a.cslab = cslab; a.dz_half = dz_half_stride; a.chain = chain ? 1 : 0;
    return wn_launch_resblock_bwd_pq(a, batch, (hipStream_t)stream);
}
int wn_resblock_bwd_pq_chain_ok(int t_lo, int t_hi, int batch, int d) { return wn_pq_chain_ok(t_lo, t_hi, batch, d); }
int wn_resblock_bwd_pq_slabs(int t_lo, int t_hi, int batch, int d, int chain) { return wn_pq_slabs(t_lo, t_hi, batch, d, chain); }
int wn_resblock_bwd_pq_chain_items(int t_lo, int t_hi, int batch, int d, int wg, int* out, int cap) {
    return wn_pq_chain_items(t_lo, t_hi, batch, d, wg, out, cap);
}
int wn_resblock_bwd_pq_cond_floats(int t_lo, int t_hi, int batch) { return wn_pq_cond_slab_floats(t_lo, t_hi, batch); }
int wn_resblock_bwd_pq_cond_reduce(const float* cslab, const int64_t* slab_off, const int* t_lo, int n_launches, int t_hi, int batch,
                                   int cond_le, float* out, int64_t out_lstride, int64_t out_bstride, int out_pitch,
                                   wn_stream_t stream) {
    if (!cslab || !out || !slab_off || !t_lo || n_launches < 1 || cond_le < 1 || cond_le > 32)
        return wn_set_error_msg(-4, "wn_resblock_bwd_pq_cond_reduce: bad argument");
    static_assert(sizeof(long) == sizeof(int64_t), "LP64");
    return wn_launch_pq_cond_reduce(cslab, reinterpret_cast<const long*>(slab_off), t_lo, n_launches, t_hi, batch, cond_le, out,
                                    (long)out_lstride, (long)out_bstride, out_pitch, (hipStream_t)stream);
}
int wn_gate_fwd(const float* fg, int64_t fg_bstride, int dp, int rows, float* z, int64_t z_bstride, int pitch, int t_lo, int t_hi,
                int batch, wn_stream_t stream) {
    if (!fg || !z || rows > dp) return wn_set_error_msg(-4, "wn_gate_fwd: bad argument");
    return wn_launch_gate_fwd(fg, (long)fg_bstride, dp, rows, z, (long)z_bstride, pitch, t_lo, t_hi, batch, (hipStream_t)stream);
}
int wn_gate_bwd(const float* fg, int64_t fg_bstride, int dp, int rows, const float* dz, int64_t dz_bstride, float* dfg,
                int64_t dfg_bstride, int pitch, int t_lo, int t_hi, int batch, wn_stream_t stream) {
    if (!fg || !dz || !dfg || rows > dp) return wn_set_error_msg(-4, "wn_gate_bwd: bad argument");
    return wn_launch_gate_bwd(fg, (long)fg_bstride, dp, rows, dz, (long)dz_bstride, dfg, (long)dfg_bstride, pitch, t_lo, t_hi, batch,
                              (hipStream_t)stream);
}
int wn_chunk_softmax_fwd(const float* x, float* y, int64_t nrows, int q, wn_stream_t stream) {
    if (q < 1 || (nrows > 0 && (!x || !y))) return wn_set_error_msg(-4, "wn_chunk_softmax_fwd: bad argument");
    return wn_launch_softmaxq_fwd(x, y, (long)nrows, q, (hipStream_t)stream);
}
int wn_chunk_softmax_bwd(const float* y, const float* dy, float* dx, int64_t nrows, int q, wn_stream_t stream) {
    if (q < 1 || (nrows > 0 && (!y || !dy || !dx))) return wn_set_error_msg(-4, "wn_chunk_softmax_bwd: bad argument");
    return wn_launch_softmaxq_bwd(y, dy, dx, (long)nrows, q, (hipStream_t)stream);
}
int wn_chunk_softmax_ce(const float* x, const int64_t* target, float* probs, float* dx, float* loss_part, int64_t nrows, int q,
                        float inv_n, wn_stream_t stream) {
    if (q < 1 || (nrows > 0 && (!x || !target))) return wn_set_error_msg(-4, "wn_chunk_softmax_ce: bad argument");
    return wn_launch_softmaxq_ce(x, target, probs, dx, loss_part, (long)nrows, q, inv_n, (hipStream_t)stream);
}
int wn_split16(const float* x, uint16_t* hi, uint16_t* lo, int64_t n, int is_bf16, wn_stream_t stream) {
    if (n > 0 && (!x || !hi || !lo)) return wn_set_error_msg(-4, "wn_split16: null argument");
    return wn_launch_split16(x, hi, lo, (long)n, is_bf16, (hipStream_t)stream);
}
int wn_shift_add(const float* p, const float* q, float* out, int64_t bstride, int pitch, int rows, int dn, int p_lo,
                 int t_lo, int t_hi, int batch, wn_stream_t stream) {
    if (batch > 0 && rows > 0 && t_hi > t_lo) WN_REQUIRE("wn_shift_add", p, q, out);
    return wn_launch_shift_add(p, q, out, bstride, pitch, rows, dn, p_lo, t_lo, t_hi, batch, (hipStream_t)stream);
}

int wn_causal_wgrad_codes(const int32_t* codes, int scrambled, const float* dx, const float* dx_q, int dn, int p_lo,
                          int64_t dx_bstride, int pitch, int ch, int q, int t, int batch, float* slab, wn_stream_t stream) {
    if (q != 256) return wn_set_error_msg(-4, "wn_causal_wgrad_codes: 256 quantisation channels only");
    if (!codes || !dx || !slab) return wn_set_error_msg(-4, "wn_causal_wgrad_codes: null argument");
    return wn_launch_causal_wgrad_codes(codes, scrambled, dx, dx_q, dn, p_lo, dx_bstride, pitch, ch, t, batch, slab, (hipStream_t)stream);
}
int wn_causal_wgrad_codes_slabs(int t, int batch) { return wn_causal_codes_slabs(t, batch); }
int wn_causal_fwd_codes(const int32_t* codes, int scrambled, const float* wt, const float* bias, int n_rows, float* x0,
                        int64_t x_bstride, int pitch, int ch, int q, int t, int batch, wn_stream_t stream) {
    if (q != 256) return wn_set_error_msg(-4, "wn_causal_fwd_codes: 256 quantisation channels only");
    if (!codes || !wt || !x0) return wn_set_error_msg(-4, "wn_causal_fwd_codes: null argument");
    return wn_launch_causal_fwd_codes(codes, scrambled, wt, bias, n_rows, x0, x_bstride, pitch, ch, t, batch, (hipStream_t)stream);
}

int wn_cond_grad(const float* in, int64_t in_bstride, int in_pitch, int rows, int t_lo, int t_hi, int mode, int le,
                 int q, float* out, int64_t out_bstride, int out_pitch, int batch, wn_stream_t stream) {
    if (batch > 0 && rows > 0 && t_hi > t_lo) WN_REQUIRE("wn_cond_grad", in, out);
    return wn_launch_cond_grad(in, in_bstride, in_pitch, rows, t_lo, t_hi, mode, le, q, out, out_bstride, out_pitch, batch,
                               (hipStream_t)stream);
}
int wn_cond_expand(const float* tab, int64_t tab_bstride, int tab_pitch, int rows, int t_lo, int t_hi, int mode, int le, int q,
                   float* out, int64_t out_bstride, int out_pitch, int batch, wn_stream_t stream) {
    if (!tab || !out || le <= 0 || (mode == 1 && q <= 0) || (mode != 1 && mode != 2))
        return wn_set_error_msg(-4, "wn_cond_expand: bad argument");
    return wn_launch_cond_expand(tab, (long)tab_bstride, tab_pitch, rows, t_lo, t_hi, mode, le, q, out, (long)out_bstride, out_pitch,
                                 batch, (hipStream_t)stream);
}
int wn_avgpool_bwd(const float* denc, int64_t denc_bstride, int denc_pitch, int t0, int pool, int n_out, int rows,
                   float* out, int64_t out_bstride, int out_pitch, int t_hi, int batch, wn_stream_t stream) {
    if (batch > 0 && rows > 0) WN_REQUIRE("wn_avgpool_bwd", denc, out);
    return wn_launch_avgpool_bwd(denc, denc_bstride, denc_pitch, t0, pool, n_out, rows, out, out_bstride, out_pitch, t_hi,
                                 batch, (hipStream_t)stream);
}

int wn_avgpool(const float* in, int64_t in_bstride, int in_pitch, int t0, int pool, int n_out, int rows,
               float* out, int64_t out_bstride, int out_pitch, int batch, wn_stream_t stream) {
    if (batch > 0 && rows > 0 && n_out > 0) WN_REQUIRE("wn_avgpool", in, out);
    return wn_launch_avgpool(in, in_bstride, in_pitch, t0, pool, n_out, rows, out, out_bstride, out_pitch, batch,
                             (hipStream_t)stream);
}

}  // extern "C"
static int decode_impl(int n_layers, int R, int D, int S, int Q, const int32_t* dilations_host, const int64_t* q_off_host,
                       float* queues, const float* w_causal, const float* b_causal, const float* w_layers,
                       int64_t layer_stride, const float* b_layers, const float* w_p1, const float* b_p1,
                       const float* w_p2, const float* b_p2, const float* note0, const float* prev0, float* note_out,
                       float* prev_out, const int32_t* forced, int32_t* codes_out, float* probs_out, int64_t step0,
                       int n_steps, int push_input, uint64_t* sync, int64_t sync_ustride, int n_utt, int64_t queues_ustride,
                       float temperature, uint64_t seed, const uint16_t* pk, int64_t pk_fg0, int64_t pk_d0, int64_t pk_lstride,
                       int64_t pk_skip, int64_t pk_p1, int64_t pk_p2, wn_stream_t stream);
extern "C" {
int64_t wn_decode_sync_granules(int n_layers, int D, int S);
int wn_decode_batch(int n_layers, int R, int D, int S, int Q, const int32_t* dilations_host, const int64_t* q_off_host,
                    float* queues, const float* w_causal, const float* b_causal, const float* w_layers,
                    int64_t layer_stride, const float* b_layers, const float* w_p1, const float* b_p1,
                    const float* w_p2, const float* b_p2, const float* note0, const float* prev0, float* note_out,
                    float* prev_out, const int32_t* forced, int32_t* codes_out, float* probs_out, int64_t step0,
                    int n_steps, int push_input, uint64_t* sync, int n_utt, int64_t queues_ustride, float temperature,
                    uint64_t seed, wn_stream_t stream);
int wn_decode_batch_pk(int n_layers, int R, int D, int S, int Q, const int32_t* dilations_host, const int64_t* q_off_host,
                       float* queues, const float* w_causal, const float* b_causal, const float* w_layers,
                       int64_t layer_stride, const float* b_layers, const float* w_p1, const float* b_p1,
                       const float* w_p2, const float* b_p2, const float* note0, const float* prev0, float* note_out,
                       float* prev_out, const int32_t* forced, int32_t* codes_out, float* probs_out, int64_t step0,
                       int n_steps, int push_input, uint64_t* sync, int n_utt, int64_t queues_ustride, float temperature,
                       uint64_t seed, const uint16_t* pk, int64_t pk_fg0, int64_t pk_d0, int64_t pk_lstride, int64_t pk_skip,
                       int64_t pk_p1, int64_t pk_p2, wn_stream_t stream);

int wn_decode(int n_layers, int R, int D, int S, int Q, const int32_t* dilations_host, const int64_t* q_off_host,
              float* queues, const float* w_causal, const float* b_causal, const float* w_layers,
              int64_t layer_stride, const float* b_layers, const float* w_p1, const float* b_p1,
              const float* w_p2, const float* b_p2, const float* note0, const float* prev0, float* note_out,
              float* prev_out, const int32_t* forced, int32_t* codes_out, float* probs_out, int64_t step0,
              int n_steps, int push_input, uint64_t* sync, wn_stream_t stream) {
    return wn_decode_batch(n_layers, R, D, S, Q, dilations_host, q_off_host, queues, w_causal, b_causal, w_layers, layer_stride,
                           b_layers, w_p1, b_p1, w_p2, b_p2, note0, prev0, note_out, prev_out, forced, codes_out, probs_out,
                           step0, n_steps, push_input, sync, 1, 0, 0.0f, 0, stream);
}

int wn_decode_batch(int n_layers, int R, int D, int S, int Q, const int32_t* dilations_host, const int64_t* q_off_host,
                    float* queues, const float* w_causal, const float* b_causal, const float* w_layers,
                    int64_t layer_stride, const float* b_layers, const float* w_p1, const float* b_p1,
                    const float* w_p2, const float* b_p2, const float* note0, const float* prev0, float* note_out,
                    float* prev_out, const int32_t* forced, int32_t* codes_out, float* probs_out, int64_t step0,
                    int n_steps, int push_input, uint64_t* sync, int n_utt, int64_t queues_ustride, float temperature,
                    uint64_t seed, wn_stream_t stream) {
    return decode_impl(n_layers, R, D, S, Q, dilations_host, q_off_host, queues, w_causal, b_causal, w_layers, layer_stride,
                       b_layers, w_p1, b_p1, w_p2, b_p2, note0, prev0, note_out, prev_out, forced, codes_out, probs_out,
                       step0, n_steps, push_input, sync, (int64_t)n_layers * D + 2, n_utt, queues_ustride, temperature, seed,
                       nullptr, 0, 0, 0, -1, -1, -1, stream);
}

int64_t wn_decode_sync_granules(int n_layers, int D, int S) {
    return (int64_t)wn_decode_granules(n_layers, D, S);      // z of every block, the split form's vectors, the tap-0 table, code, error flag
}

int wn_decode_batch_pk(int n_layers, int R, int D, int S, int Q, const int32_t* dilations_host, const int64_t* q_off_host,
                       float* queues, const float* w_causal, const float* b_causal, const float* w_layers,
                       int64_t layer_stride, const float* b_layers, const float* w_p1, const float* b_p1,
                       const float* w_p2, const float* b_p2, const float* note0, const float* prev0, float* note_out,
                       float* prev_out, const int32_t* forced, int32_t* codes_out, float* probs_out, int64_t step0,
                       int n_steps, int push_input, uint64_t* sync, int n_utt, int64_t queues_ustride, float temperature,
                       uint64_t seed, const uint16_t* pk, int64_t pk_fg0, int64_t pk_d0, int64_t pk_lstride, int64_t pk_skip,
                       int64_t pk_p1, int64_t pk_p2, wn_stream_t stream) {
    return decode_impl(n_layers, R, D, S, Q, dilations_host, q_off_host, queues, w_causal, b_causal, w_layers, layer_stride,
                       b_layers, w_p1, b_p1, w_p2, b_p2, note0, prev0, note_out, prev_out, forced, codes_out, probs_out,
                       step0, n_steps, push_input, sync, wn_decode_sync_granules(n_layers, D, S), n_utt, queues_ustride,
                       temperature, seed, pk, pk_fg0, pk_d0, pk_lstride, pk_skip, pk_p1, pk_p2, stream);
}

}  // extern "C"

static int decode_impl(int n_layers, int R, int D, int S, int Q, const int32_t* dilations_host, const int64_t* q_off_host,
                       float* queues, const float* w_causal, const float* b_causal, const float* w_layers,
                       int64_t layer_stride, const float* b_layers, const float* w_p1, const float* b_p1,
                       const float* w_p2, const float* b_p2, const float* note0, const float* prev0, float* note_out,
                       float* prev_out, const int32_t* forced, int32_t* codes_out, float* probs_out, int64_t step0,
                       int n_steps, int push_input, uint64_t* sync, int64_t sync_ustride, int n_utt, int64_t queues_ustride,
                       float temperature, uint64_t seed, const uint16_t* pk, int64_t pk_fg0, int64_t pk_d0, int64_t pk_lstride,
                       int64_t pk_skip, int64_t pk_p1, int64_t pk_p2, wn_stream_t stream) {
    if (n_utt <= 0) return 0;
    if (n_layers > WN_DEC_MAX_LAYERS || n_layers <= 0) return wn_set_error_msg(-4, "wn_decode: 1..64 layers supported");
    WN_REQUIRE("wn_decode", dilations_host, q_off_host);              // (host arrays, read right here)
    if (n_steps > 0) WN_REQUIRE("wn_decode", queues, w_causal, w_layers, w_p1, w_p2, sync);
    WnDecodeArgs a;
    memset(&a, 0, sizeof(a));
    a.n_layers = n_layers; a.R = R; a.D = D; a.S = S; a.Q = Q;
    for (int i = 0; i < n_layers; ++i) { a.dil[i] = dilations_host[i]; a.q_off[i] = q_off_host[i]; }
    a.queues = queues; a.w_causal = w_causal; a.b_causal = b_causal; a.w_layers = w_layers; a.layer_stride = layer_stride;
    a.b_layers = b_layers; a.w_p1 = w_p1; a.b_p1 = b_p1; a.w_p2 = w_p2; a.b_p2 = b_p2;
    a.note0 = note0; a.prev0 = prev0; a.note_out = note_out; a.prev_out = prev_out; a.forced = forced;
    a.codes_out = codes_out; a.probs_out = probs_out; a.step0 = step0; a.n_steps = n_steps; a.push_input = push_input;
    a.dbg = 0;
    a.sync = reinterpret_cast<unsigned long long*>(sync);
    a.sync_ustride = sync_ustride;
    a.n_utt = n_utt; a.queues_ustride = queues_ustride;
    a.sample = temperature > 0.0f ? 1 : 0; a.inv_temp = temperature > 0.0f ? 1.0f / temperature : 1.0f; a.seed = seed;
    a.pk_skip = -1;
    // the matrix-core kernel needs all of pk (chain, skip, post-processing: 64 / 64 / 256 / 256 channels); biases are fine
    const bool any_bias = b_layers || b_causal || b_p1 || b_p2;
    const bool post_pk = (S == 256 || S == 512) && Q == 256 && pk_skip >= 0 && pk_p1 >= 0 && pk_p2 >= 0;
    if (pk && R == 64 && D == 64 && (post_pk || !any_bias)) {
        a.pk = pk; a.pk_fg0 = pk_fg0; a.pk_d0 = pk_d0; a.pk_lstride = pk_lstride;
        if (post_pk) { a.pk_skip = pk_skip; a.pk_p1 = pk_p1; a.pk_p2 = pk_p2; }
    }
    return wn_launch_decode(a, (hipStream_t)stream);
}
