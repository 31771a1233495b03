"""The GENERAL execution plan: every constructor argument the reference accepts (wavenet/model.py:8-15).

`engine.WaveNetEngine` drives the specialised kernels (fused residual block forward, one-launch backward block, code-aware
causal layer) and covers filter_width == 2, quantization_channels == 256 and up to 64 residual / dilation channels - every
configuration the reference ships or BASELINE.json names.  Anything else used to raise.  This engine runs the same
arithmetic for ANY filter width, quantisation channel count and channel widths out of the library's
general kernels, one product per launch:

    conv with k taps        wn_chan_gemm, two taps per launch (input shifted by -(k-1-j) d for tap j), further pairs
                            accumulate through `resid`                      (model.py:104,118-119)
    gate                    wn_gate_fwd / wn_gate_bwd (f and g are kept for the backward)   (model.py:120)
    dense + residual, skip, post-processing      wn_chan_gemm               (model.py:121-138)
    chunk softmax (+ CE)    wn_chunk_softmax256_* when Q == 256, wn_chunk_softmax_* otherwise   (model.py:142-144)
    weight gradients        wn_wgrad slabs + wn_reduce_slabs (bit-reproducible), data gradients wn_chan_gemm on W^T

Same x3 arithmetic (f16 split forward, bf16 split backward), same HBM layout (absolute time, one pitch, channels padded to
32 with zero weights), same flat parameter / gradient buffers, same workspace pool as the fast engine; 2x slower per step
at config-2 shapes (8.6 vs 4.3 ms: it is round 1's first correct structure), which is why it is only selected where the fast engine does
not apply.  PyTorch is used for device memory and streams only.  Nothing here imports oracle/.
"""
import os

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr
from .engine import SLACK, PAD_BACK, WorkspacePool, _Spec, _pad, pack_index


class GenericWaveNetEngine:
    def __init__(self, dilations, residual_channels, dilation_channels, skip_channels, quantization_channels=256,
                 filter_width=2, use_bias=False, mode_fwd="f16x3", mode_bwd="bf16x3", device=None):
        if filter_width < 1:
            raise ValueError("filter_width must be >= 1")
        self.dil = [int(d) for d in dilations]
        self.N = len(self.dil)
        self.k = int(filter_width)
        self.R, self.D, self.S, self.Q = residual_channels, dilation_channels, skip_channels, quantization_channels
        self.use_bias = bool(use_bias)
        self.RP, self.DP, self.SP, self.QP = (_pad(v, 32) for v in (self.R, self.D, self.S, self.Q))
        self.CH = self.RP                                  # rows of a residual-stream layer (fast_generate reads ws["X"])
        self.rf = (self.k - 1) * (sum(self.dil) + 1) + 1
        self.off = [self.k - 1]                            # first valid absolute time of x_i
        for d in self.dil:
            self.off.append(self.off[-1] + (self.k - 1) * d)
        assert self.off[-1] == self.rf - 1
        self.pairs = [(j, j + 1 if j + 1 < self.k else None) for j in range(0, self.k, 2)]
        self.mode_fwd = _lib.MODE_NAMES[mode_fwd] if isinstance(mode_fwd, str) else mode_fwd
        self.mode_bwd = _lib.MODE_NAMES[mode_bwd] if isinstance(mode_bwd, str) else mode_bwd
        self.device = torch.device(device if device is not None else "cuda")
        _lib.load()
        self._build_spec()
        self._build_packs()
        self._ws = WorkspacePool(self._make_workspace)
        self._gen = 0
        self.adam_state = None
        self.marks = None
        self.mark_only = None

    def mark(self, name):
        if self.marks is not None and (self.mark_only is None or name in self.mark_only):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev))

    # ------------------------------------------------------------------ parameters (the fast engine's layout)
    def _build_spec(self):
        k = self.k
        names = [("causal_layer.weight", (self.R, self.Q, k))]
        if self.use_bias:
            names.append(("causal_layer.bias", (self.R,)))
        for i in range(self.N):
            for j, shp in enumerate([(self.D, self.R, k), (self.D, self.R, k), (self.R, self.D, 1), (self.S, self.D, 1)]):
                names.append(("dilation_layer_stack.%d.weight" % (4 * i + j), shp))
                if self.use_bias:
                    names.append(("dilation_layer_stack.%d.bias" % (4 * i + j), (shp[0],)))
        names.append(("post_process_1.weight", (self.S, self.S, 1)))
        if self.use_bias:
            names.append(("post_process_1.bias", (self.S,)))
        names.append(("post_process_2.weight", (self.Q, self.S, 1)))
        if self.use_bias:
            names.append(("post_process_2.bias", (self.Q,)))
        self.spec = _Spec(names)
        self.param_names = [n for n, _ in names]
        self.flat = torch.zeros(self.spec.total, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros(self.spec.total, dtype=torch.float32, device=self.device)

    def param_view(self, name, grad=False):
        o, shp = self.spec.off[name], self.spec.shape[name]
        n = int(np.prod(shp))
        return (self.flat_grad if grad else self.flat)[o:o + n].view(shp)

    def load_state_dict_tensors(self, sd):
        with torch.no_grad():
            for n in self.param_names:
                self.param_view(n).copy_(sd[n])

    def _bias_ptr(self, name):
        return ptr(self.flat, self.spec.off[name]) if self.use_bias else None

    # ------------------------------------------------------------------ packs and gradient maps
    def _build_packs(self):
        sp, N, R, D, S, Q = self.spec, self.N, self.R, self.D, self.S, self.Q
        RP, DP, SP, QP = self.RP, self.DP, self.SP, self.QP
        fwd, bwd, gp = [], [], []
        gidx = np.full(sp.total, -1, dtype=np.int64)
        gp_off, go = {}, [0]

        def full(m, kk):
            return np.full((m, kk), -1, dtype=np.int64)

        def add_gp(name, rows, cols):
            gp_off[name] = (go[0], rows, cols)
            go[0] += rows * cols
            return gp_off[name][0]

        def put(pname, mat_off):
            po = sp.off[pname]
            gidx[po:po + mat_off.size] = mat_off.reshape(-1)

        def taps_of(pair):
            return [j for j in pair if j is not None]
        wc = sp.conv("causal_layer.weight")                                  # [R][Q][k]
        gmap_c = np.zeros((R, Q, self.k), dtype=np.int64)
        for p, pair in enumerate(self.pairs):
            tp = taps_of(pair)
            w = full(RP, len(tp) * QP)
            o0 = add_gp("causal_%d" % p, RP, len(tp) * QP)
            for tl, j in enumerate(tp):
                w[:R, tl * QP:tl * QP + Q] = wc[:, :, j]
                gmap_c[:, :, j] = o0 + np.arange(R)[:, None] * (len(tp) * QP) + tl * QP + np.arange(Q)[None, :]
            fwd.append(("causal_%d" % p, pack_index(w)))
            wt = full(QP, len(tp) * RP)                      # W_j^T per tap: the gradient w.r.t. the input (input_grad)
            for tl, j in enumerate(tp):
                wt[:Q, tl * RP:tl * RP + R] = wc[:, :, j].T
            bwd.append(("causalT_%d" % p, pack_index(wt)))
        put("causal_layer.weight", gmap_c)
        for i in range(N):
            wf = sp.conv("dilation_layer_stack.%d.weight" % (4 * i))        # [D][R][k]
            wg = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 1))
            wd = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 2))[:, :, 0]   # [R][D]
            gm = [np.zeros((D, R, self.k), dtype=np.int64) for _ in range(2)]
            for p, pair in enumerate(self.pairs):
                tp = taps_of(pair)
                w = full(2 * DP, len(tp) * RP)                               # forward: rows [f | g], cols per tap
                wt = full(RP, len(tp) * 2 * DP)                              # backward: W_j^T, cols per tap = [f | g]
                o0 = add_gp("fg%d_%d" % (i, p), 2 * DP, len(tp) * RP)
                for tl, j in enumerate(tp):
                    for h, src in enumerate((wf, wg)):
                        w[h * DP:h * DP + D, tl * RP:tl * RP + R] = src[:, :, j]
                        wt[:R, tl * 2 * DP + h * DP:tl * 2 * DP + h * DP + D] = src[:, :, j].T
                        gm[h][:, :, j] = o0 + (h * DP + np.arange(D)[:, None]) * (len(tp) * RP) + tl * RP + np.arange(R)[None, :]
                fwd.append(("fg%d_%d" % (i, p), pack_index(w)))
                bwd.append(("fgT%d_%d" % (i, p), pack_index(wt)))
            put("dilation_layer_stack.%d.weight" % (4 * i), gm[0])
            put("dilation_layer_stack.%d.weight" % (4 * i + 1), gm[1])
            w = full(RP, DP)
            w[:R, :D] = wd
            fwd.append(("d%d" % i, pack_index(w)))
            bwd.append(("dT%d" % i, pack_index(np.ascontiguousarray(w.T))))
            o0 = add_gp("d%d" % i, RP, DP)
            put("dilation_layer_stack.%d.weight" % (4 * i + 2), o0 + np.arange(R)[:, None] * DP + np.arange(D)[None, :])
        w = full(SP, N * DP)
        for i in range(N):
            w[:S, i * DP:i * DP + D] = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 3))[:, :, 0]
        fwd.append(("skip", pack_index(w)))
        bwd.append(("skipT", pack_index(np.ascontiguousarray(w.T))))
        o0 = add_gp("skip", SP, N * DP)
        for i in range(N):
            put("dilation_layer_stack.%d.weight" % (4 * i + 3), o0 + np.arange(S)[:, None] * (N * DP) + i * DP + np.arange(D)[None, :])
        w = full(SP, SP)
        w[:S, :S] = sp.conv("post_process_1.weight")[:, :, 0]
        fwd.append(("p1", pack_index(w)))
        bwd.append(("p1T", pack_index(np.ascontiguousarray(w.T))))
        o0 = add_gp("p1", SP, SP)
        put("post_process_1.weight", o0 + np.arange(S)[:, None] * SP + np.arange(S)[None, :])
        w = full(QP, SP)
        w[:Q, :S] = sp.conv("post_process_2.weight")[:, :, 0]
        fwd.append(("p2", pack_index(w)))
        bwd.append(("p2T", pack_index(np.ascontiguousarray(w.T))))
        o0 = add_gp("p2", QP, SP)
        put("post_process_2.weight", o0 + np.arange(Q)[:, None] * SP + np.arange(S)[None, :])
        self.gp_bias_off = {}
        if self.use_bias:
            for name in self.param_names:
                if name.endswith(".bias"):
                    n = sp.shape[name][0]
                    self.gp_bias_off[name] = go[0]
                    put(name, go[0] + np.arange(n))
                    go[0] += _pad(n, 4)
        assert (gidx >= 0).all()
        self.gp_off = gp_off
        dev = self.device
        self.gpack = torch.zeros(go[0], dtype=torch.float32, device=dev)
        self.gidx = torch.from_numpy(gidx.astype(np.int32)).to(dev)

        def finish(lst, mode):
            hpf = 1024 if mode in (_lib.F16X3, _lib.BF16X3) else 512
            offs, o = {}, 0
            for name, idx in lst:
                offs[name] = o * hpf // 512
                o += len(idx)
            idx_all = torch.from_numpy(np.concatenate([i for _, i in lst]).astype(np.int32)).to(dev)
            return offs, idx_all, torch.zeros(o * hpf // 512, dtype=torch.int16, device=dev)
        self.pk_f_off, self.pk_f_idx, self.pk_f = finish(fwd, self.mode_fwd)
        self.pk_b_off, self.pk_b_idx, self.pk_b = finish(bwd, self.mode_bwd)
        if self.use_bias:
            # [f | g] bias rows of every layer in the padded row order of the fg product, gathered from the flat buffer
            bi = np.full((N, 2 * DP), -1, dtype=np.int64)
            for i in range(N):
                for h in range(2):
                    o = sp.off["dilation_layer_stack.%d.bias" % (4 * i + h)]
                    bi[i, h * DP:h * DP + D] = o + np.arange(D)
            self.bfg_idx = torch.from_numpy(bi.reshape(-1).astype(np.int32)).to(dev)
            self.bfg = torch.zeros(N * 2 * DP, dtype=torch.float32, device=dev)

    def pack_weights(self):
        st = _lib.stream()
        call("wn_pack_weights", ptr(self.flat), ptr(self.pk_f_idx), ptr(self.pk_f), self.pk_f_idx.numel(), self.mode_fwd, st)
        call("wn_pack_weights", ptr(self.flat), ptr(self.pk_b_idx), ptr(self.pk_b), self.pk_b_idx.numel(), self.mode_bwd, st)
        if self.use_bias:
            call("wn_gather_grads", ptr(self.flat), ptr(self.bfg_idx), ptr(self.bfg), self.bfg.numel(), st)

    # ------------------------------------------------------------------ workspace
    def workspace(self, B, T):
        return self._ws.peek(B, T)

    def _make_workspace(self, B, T):
        dev = self.device
        pitch = _pad(T, 256) + 512
        W = T - self.rf + 1
        N, RP, DP, SP, Q = self.N, self.RP, self.DP, self.SP, self.Q

        def buf(rows):
            return torch.zeros(SLACK + B * rows * pitch + PAD_BACK, dtype=torch.float32, device=dev)
        ws = dict(B=B, T=T, W=W, pitch=pitch, bwd=None)
        ws["X"] = torch.zeros(SLACK + (N + 1) * B * RP * pitch + PAD_BACK, dtype=torch.float32, device=dev)
        ws["FG"] = torch.zeros(SLACK + N * B * 2 * DP * pitch + PAD_BACK, dtype=torch.float32, device=dev)
        ws["Z"], ws["U"], ws["H"] = buf(N * DP), buf(SP), buf(SP)
        # the compact (B, Q, W) logits; rows Q .. QP-1 of the LAST clip are read (against zero weights) by the products that
        # take this tensor as an operand, so QP - Q rows of finite slack follow it
        ws["O"] = torch.zeros(B * Q * W + 32 * W + PAD_BACK, dtype=torch.float32, device=dev)
        if self.QP != Q:
            ws["Xin"] = torch.zeros(B * self.QP * T + PAD_BACK, dtype=torch.float32, device=dev)
        return ws

    def _x(self, ws, i):
        return ptr(ws["X"], SLACK + i * ws["B"] * self.RP * ws["pitch"])

    def _fg(self, ws, i):
        return ptr(ws["FG"], SLACK + i * ws["B"] * 2 * self.DP * ws["pitch"])

    def _gemm(self, st, B, mode, wpack, in0, in1, in_bs, in_pitch, in_lo, in_hi, s0, s1, ks0, ks1, mt, m_valid, out, out_bs,
              out_pitch, out_shift, bias, resid, mask, t_lo, t_hi, relu_in):
        call("wn_chan_gemm", in0, in1, in_bs, in_pitch, in_lo, in_hi, s0, s1, ks0, ks1, wpack, mt, m_valid,
             out, out_bs, out_pitch, out_shift, bias, resid[0], resid[1], resid[2], resid[3],
             mask[0], mask[1], mask[2], t_lo, t_hi, relu_in, B, mode, st)

    def _taps(self, pair, d):
        """(shift of tap j0, shift of tap j1 or 0, 1 if there is a second tap): input column = t - (k-1-j) d"""
        j0, j1 = pair
        return -(self.k - 1 - j0) * d, (-(self.k - 1 - j1) * d if j1 is not None else 0), (1 if j1 is not None else 0)

    # ------------------------------------------------------------------ forward
    def forward_logits(self, x, ws=None):
        B, Q, T = x.shape
        assert Q == self.Q and x.is_contiguous() and x.dtype == torch.float32 and x.is_cuda
        W = T - self.rf + 1
        if W <= 0:
            raise ValueError("wave sample not long enough")          # wavenet/model.py:100-101
        ws = ws or self._ws.get(B, T)
        st = _lib.stream()
        N, RP, DP, SP, QP, pitch, mf = self.N, self.RP, self.DP, self.SP, self.QP, ws["pitch"], self.mode_fwd
        fr = lambda name: ptr(self.pk_f, self.pk_f_off[name])
        NONE4, NONE3 = (None, 0, 0, 0), (None, 0, 0)
        self._gen += 1
        ws["gen"], ws["x_in"], ws["x_ver"] = self._gen, x, x._version
        xb, fb, zb, sb = RP * pitch, 2 * DP * pitch, N * DP * pitch, SP * pitch
        if QP != Q:                                          # K runs over whole 32-row steps: a zero-padded copy of the input
            xin = ws["Xin"][:B * QP * T].view(B, QP, T)
            xin[:, :Q].copy_(x)
            xin_p, xin_bs = ptr(ws["Xin"]), QP * T
        else:
            xin_p, xin_bs = ptr(x), Q * T
        ws["xin"] = (xin_p, xin_bs)
        k0 = self.k - 1
        for p, pair in enumerate(self.pairs):                # causal conv (model.py:104): x0[t] = sum_j Wc_j in[t - (k-1-j)]
            s0, s1, two = self._taps(pair, 1)
            self._gemm(st, B, mf, fr("causal_%d" % p), xin_p, xin_p if two else None, xin_bs, T, 0, T, s0, s1, QP // 32,
                       QP // 32 if two else 0, RP // 16, self.R, self._x(ws, 0), xb, pitch, 0,
                       self._bias_ptr("causal_layer.bias") if p == 0 else None,
                       (self._x(ws, 0), xb, pitch, k0) if p else NONE4, NONE3, k0, T, 0)
        self.mark("causal_fwd")
        bn = "dilation_layer_stack.%d.bias"
        for i, d in enumerate(self.dil):
            t_in, t_lo = self.off[i], self.off[i + 1]
            for p, pair in enumerate(self.pairs):            # [f; g] = sum_j [Wf_j; Wg_j] x_i[t - (k-1-j) d]   (model.py:118-119)
                s0, s1, two = self._taps(pair, d)
                self._gemm(st, B, mf, fr("fg%d_%d" % (i, p)), self._x(ws, i), self._x(ws, i) if two else None, xb, pitch, t_in, T,
                           s0, s1, RP // 32, RP // 32 if two else 0, 2 * DP // 16, 2 * DP, self._fg(ws, i), fb, pitch, 0,
                           ptr(self.bfg, i * 2 * DP) if (self.use_bias and p == 0) else None,
                           (self._fg(ws, i), fb, pitch, t_lo) if p else NONE4, NONE3, t_lo, T, 0)
            z_i = ptr(ws["Z"], SLACK + i * DP * pitch)
            call("wn_gate_fwd", self._fg(ws, i), fb, DP, DP, z_i, zb, pitch, t_lo, T, B, st)          # model.py:120
            if i < N - 1:                                    # x_{i+1} = Wd z + x_i[t]                            (model.py:121-124)
                self._gemm(st, B, mf, fr("d%d" % i), z_i, None, zb, pitch, t_lo, T, 0, 0, DP // 32, 0, RP // 16, self.R,
                           self._x(ws, i + 1), xb, pitch, 0, self._bias_ptr(bn % (4 * i + 2)),
                           (self._x(ws, i), xb, pitch, t_lo), NONE3, t_lo, T, 0)
        self.mark("stack_fwd")
        lo = self.rf - 1
        bias_s = None
        if self.use_bias:
            ws["bias_skip"] = sum(self.param_view(bn % (4 * i + 3)) for i in range(N)).contiguous()
            bias_s = ptr(ws["bias_skip"])
        U, H = ptr(ws["U"], SLACK), ptr(ws["H"], SLACK)
        self._gemm(st, B, mf, fr("skip"), ptr(ws["Z"], SLACK), None, zb, pitch, lo, T, 0, 0, N * DP // 32, 0, SP // 16, self.S,
                   U, sb, pitch, 0, bias_s, NONE4, NONE3, lo, T, 0)                                    # model.py:127-134
        self._gemm(st, B, mf, fr("p1"), U, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, SP // 16, self.S, H, sb, pitch, 0,
                   self._bias_ptr("post_process_1.bias"), NONE4, NONE3, lo, T, 1)
        self._gemm(st, B, mf, fr("p2"), H, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, QP // 16, Q, ptr(ws["O"]), Q * W, W, -lo,
                   self._bias_ptr("post_process_2.bias"), NONE4, NONE3, lo, T, 1)
        self.mark("epilogue_fwd")
        return ws

    def _softmax_fwd(self, logits, probs, n):
        if self.Q == 256:
            call("wn_chunk_softmax256_fwd", logits, probs, n, _lib.stream())
        else:
            call("wn_chunk_softmax_fwd", logits, probs, n, self.Q, _lib.stream())

    def forward(self, x):
        """wavenet/model.py:86-145 -> probabilities (B*W, Q) (fresh tensor)."""
        self.pack_weights()
        ws = self.forward_logits(x)
        B, W = ws["B"], ws["W"]
        probs = torch.empty(B * W, self.Q, dtype=torch.float32, device=self.device)
        self._softmax_fwd(ptr(ws["O"]), ptr(probs), B * W)
        ws["probs"] = probs
        return probs, ws

    # ------------------------------------------------------------------ backward
    def _bwd_workspace(self, ws):
        if ws["bwd"] is not None:
            return ws["bwd"]
        B, T, W, pitch, dev = ws["B"], ws["T"], ws["W"], ws["pitch"], self.device
        N, RP, DP, SP, QP, Q = self.N, self.RP, self.DP, self.SP, self.QP, self.Q

        def buf(rows):
            return torch.zeros(SLACK + B * rows * pitch + PAD_BACK, dtype=torch.float32, device=dev)
        bw = dict(dO=torch.zeros(B * Q * W + 32 * W + PAD_BACK, dtype=torch.float32, device=dev), dH=buf(SP), dU=buf(SP),
                  dZ=buf(N * DP), dX=[buf(RP), buf(RP)], dz=buf(DP), dfg=buf(2 * DP))
        lo = self.rf - 1
        ops = [("p2", lo, 1024), ("p1", lo, 1024), ("skip", lo, 2048)]
        for i in range(N):
            ops += [("fg%d_%d" % (i, p), self.off[i + 1], 512) for p in range(len(self.pairs))]
            if i < N - 1:
                ops.append(("d%d" % i, self.off[i + 1], 512))
        ops += [("causal_%d" % p, self.k - 1, 512) for p in range(len(self.pairs))]
        plan, desc, so, vs = {}, [], 0, 0
        for name, t_lo, chunk in ops:
            go, r, c = self.gp_off[name]
            n = r * c
            ns = _lib.wgrad_slabs(t_lo, T, chunk, B)
            plan[name] = (so, n, chunk)
            desc.append([vs, so, ns, n, go, n])
            vs += (n + 3) // 4
            so += ns * n
        bw["slab"] = torch.empty(so, dtype=torch.float32, device=dev)
        bw["plan"], bw["vec"], bw["nops"] = plan, vs, len(desc)
        bw["desc"] = torch.tensor(desc, dtype=torch.int64, device=dev)
        ws["bwd"] = bw
        return bw

    def backward_from_dlogits(self, ws):
        """ws['bwd']['dO'] holds d loss / d pre-softmax (B, Q, W).  Fills self.flat_grad."""
        bw = self._bwd_workspace(ws)
        st = _lib.stream()
        B, T, W, pitch = ws["B"], ws["T"], ws["W"], ws["pitch"]
        N, RP, DP, SP, QP, Q, mb = self.N, self.RP, self.DP, self.SP, self.QP, self.Q, self.mode_bwd
        br = lambda name: ptr(self.pk_b, self.pk_b_off[name])
        NONE4, NONE3 = (None, 0, 0, 0), (None, 0, 0)
        lo = self.rf - 1
        xb, fb, zb, sb = RP * pitch, 2 * DP * pitch, N * DP * pitch, SP * pitch
        plan = bw["plan"]
        if ws.get("x_ver") is not None and ws["x_in"]._version != ws["x_ver"]:
            raise RuntimeError("music_amd: the input of this forward was modified in place before backward()")

        def wgrad(name, *args):
            """args = wn_wgrad's arguments up to relu_b, then ldc, t_lo, t_hi"""
            so, n, chunk = plan[name]
            head, (ldc, t_lo, t_hi) = args[:-3], args[-3:]
            call("wn_wgrad", *head, ptr(bw["slab"], so), ldc, n, t_lo, t_hi, chunk, B, mb, st)

        def bias_grad(name, a, a_bs, a_pitch, a_shift, rows, t_lo, dst=0):
            if self.use_bias:
                call("wn_bias_grad", a, a_bs, a_pitch, a_shift, rows, t_lo, T, B, ptr(self.gpack, self.gp_bias_off[name] + dst), st)
        dO, dH, dU, dZ = ptr(bw["dO"]), ptr(bw["dH"], SLACK), ptr(bw["dU"], SLACK), ptr(bw["dZ"], SLACK)
        U, H, Z = ptr(ws["U"], SLACK), ptr(ws["H"], SLACK), ptr(ws["Z"], SLACK)
        # ---- epilogue: o = P2 relu(h), h = P1 relu(u), u = sum_i Ws_i z_i
        wgrad("p2", dO, Q * W, W, -lo, W, H, None, sb, pitch, 0, 0, pitch, SP // 16, QP // 16, 1, SP, lo, T)
        self._gemm(st, B, mb, br("p2T"), dO, None, Q * W, W, 0, W, -lo, 0, QP // 32, 0, SP // 16, self.S, dH, sb, pitch, 0, None,
                   NONE4, (H, sb, pitch), lo, T, 0)
        wgrad("p1", dH, sb, pitch, 0, pitch, U, None, sb, pitch, 0, 0, pitch, SP // 16, SP // 16, 1, SP, lo, T)
        self._gemm(st, B, mb, br("p1T"), dH, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, SP // 16, self.S, dU, sb, pitch, 0, None,
                   NONE4, (U, sb, pitch), lo, T, 0)
        wgrad("skip", dU, sb, pitch, 0, pitch, Z, None, zb, pitch, 0, 0, pitch, N * DP // 16, SP // 16, 0, N * DP, lo, T)
        self._gemm(st, B, mb, br("skipT"), dU, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, N * DP // 16, N * DP, dZ, zb, pitch, 0,
                   None, NONE4, NONE3, lo, T, 0)
        bn = "dilation_layer_stack.%d.bias"
        bias_grad("post_process_2.bias", dO, Q * W, W, -lo, Q, lo)
        bias_grad("post_process_1.bias", dH, sb, pitch, 0, self.S, lo)
        for i in range(N if self.use_bias else 0):
            bias_grad(bn % (4 * i + 3), dU, sb, pitch, 0, self.S, lo)
        self.mark("epilogue_bwd")
        dfg, dzb = ptr(bw["dfg"], SLACK), ptr(bw["dz"], SLACK)
        for i in range(N - 1, -1, -1):
            d, t_in, t_lo = self.dil[i], self.off[i], self.off[i + 1]
            z_i, dz_i = ptr(ws["Z"], SLACK + i * DP * pitch), ptr(bw["dZ"], SLACK + i * DP * pitch)
            dy = ptr(bw["dX"][(i + 1) % 2], SLACK) if i < N - 1 else None
            if dy is not None:                               # dz = Wd^T dy + dz_crop                       (Appendix B)
                wgrad("d%d" % i, dy, xb, pitch, 0, pitch, z_i, None, zb, pitch, 0, 0, pitch, DP // 16, RP // 16, 0, DP, t_lo, T)
                self._gemm(st, B, mb, br("dT%d" % i), dy, None, xb, pitch, t_lo, T, 0, 0, RP // 32, 0, DP // 16, DP, dzb,
                           DP * pitch, pitch, 0, None, (dz_i, zb, pitch, lo), NONE3, t_lo, T, 0)
                dz_p, dz_bs = dzb, DP * pitch
                bias_grad(bn % (4 * i + 2), dy, xb, pitch, 0, self.R, t_lo)
            else:                                            # the last block's x_N is unused: only the skip path reaches z
                dz_p, dz_bs = dz_i, zb
            call("wn_gate_bwd", self._fg(ws, i), fb, DP, DP, dz_p, dz_bs, dfg, fb, pitch, t_lo, T, B, st)
            bias_grad(bn % (4 * i), dfg, fb, pitch, 0, self.D, t_lo)
            bias_grad(bn % (4 * i + 1), ptr(bw["dfg"], SLACK + DP * pitch), fb, pitch, 0, self.D, t_lo)
            out = ptr(bw["dX"][i % 2], SLACK)
            for p, pair in enumerate(self.pairs):
                s0, s1, two = self._taps(pair, d)
                ntap = 2 if two else 1
                # dW[f;g]_j = sum_t [df;dg][t] x_i[t - (k-1-j) d]^T
                wgrad("fg%d_%d" % (i, p), dfg, fb, pitch, 0, pitch, self._x(ws, i), self._x(ws, i) if two else None, xb, pitch,
                      s0, s1, pitch, RP // 16, 2 * DP // 16, 0, ntap * RP, t_lo, T)
                # dx_i[t] = sum_j W_j^T [df;dg][t + (k-1-j) d] + dx_{i+1}[t]
                resid = (out, xb, pitch, t_in) if p else ((dy, xb, pitch, t_lo) if dy is not None else NONE4)
                self._gemm(st, B, mb, br("fgT%d_%d" % (i, p)), dfg, dfg if two else None, fb, pitch, t_lo, T, -s0, -s1,
                           2 * DP // 32, 2 * DP // 32 if two else 0, RP // 16, self.R, out, xb, pitch, 0, None, resid, NONE3,
                           t_in, T, 0)
        self.mark("stack_bwd")
        dx0 = ptr(bw["dX"][0], SLACK)
        xin_p, xin_bs = ws["xin"]
        k0 = self.k - 1
        for p, pair in enumerate(self.pairs):
            s0, s1, two = self._taps(pair, 1)
            wgrad("causal_%d" % p, dx0, xb, pitch, 0, pitch, xin_p, xin_p if two else None, xin_bs, T, s0, s1, T, QP // 16,
                  RP // 16, 0, (2 if two else 1) * QP, k0, T)
        bias_grad("causal_layer.bias", dx0, xb, pitch, 0, self.R, k0)
        call("wn_reduce_slabs", ptr(bw["desc"]), bw["nops"], bw["vec"], ptr(bw["slab"]), ptr(self.gpack), st)
        call("wn_gather_grads", ptr(self.gpack), ptr(self.gidx), ptr(self.flat_grad), self.spec.total, st)
        self.mark("slab_reduce")

    def input_grad(self, ws):
        """Gradient of the last backward w.r.t. the module's INPUT (the causal nn.Conv1d's data gradient, wavenet/model.py:104):
        din[q][s] = sum_j sum_r Wc_j[r][q] dx0[r][s + (k-1-j)], dx0 living on [k-1, T).  One channel product per tap pair."""
        bw = ws["bwd"]
        if bw is None:
            raise RuntimeError("music_amd: input_grad() needs the backward of this forward to have run")
        B, T, pitch, RP, QP, Q = ws["B"], ws["T"], ws["pitch"], self.RP, self.QP, self.Q
        dx0 = ptr(bw["dX"][0], SLACK)
        din = None
        for p, pair in enumerate(self.pairs):
            s0, s1, two = self._taps(pair, 1)
            out = torch.empty(B, Q, T, dtype=torch.float32, device=self.device)
            self._gemm(_lib.stream(), B, self.mode_bwd, ptr(self.pk_b, self.pk_b_off["causalT_%d" % p]), dx0, dx0 if two else None, RP * pitch,
                       pitch, self.k - 1, T, -s0, -s1, RP // 32, RP // 32 if two else 0, QP // 16, Q, ptr(out), Q * T, T, 0, None,
                       (None, 0, 0, 0), (None, 0, 0), 0, T, 0)
            din = out if din is None else din.add_(out)
        return din

    def backward(self, ws, dprobs):
        """dprobs: (B*W, Q) gradient w.r.t. the probabilities returned by forward()."""
        bw = self._bwd_workspace(ws)
        dprobs = dprobs.contiguous()
        n = ws["B"] * ws["W"]
        if self.Q == 256:
            call("wn_chunk_softmax256_bwd", ptr(ws["probs"]), ptr(dprobs), ptr(bw["dO"]), n, _lib.stream())
        else:
            call("wn_chunk_softmax_bwd", ptr(ws["probs"]), ptr(dprobs), ptr(bw["dO"]), n, self.Q, _lib.stream())
        self.backward_from_dlogits(ws)

    # ------------------------------------------------------------------ fused training step (wavenet/train.py:178-181)
    def loss_and_grad(self, x, target, want_probs=False):
        if getattr(self, "_throttle", None) is None:
            self._throttle = _lib.StepThrottle()   # at most WN_MAX_STEPS_IN_FLIGHT fused steps in flight (music_amd/_lib.py)
        self._throttle.enter()
        self.mark("begin")
        self.pack_weights()
        ws = self.forward_logits(x)
        bw = self._bwd_workspace(ws)
        n = ws["B"] * ws["W"]
        target = target.reshape(-1)
        assert target.numel() == n and target.dtype == torch.int64 and target.is_cuda
        if "loss_part" not in ws:
            ws["loss_part"] = torch.zeros(_lib.CE_NUM_PARTIALS, dtype=torch.float32, device=self.device)
        probs = None
        if want_probs:
            probs = torch.empty(n, self.Q, dtype=torch.float32, device=self.device)
            ws["probs"] = probs
        if self.Q == 256:
            call("wn_chunk_softmax256_ce", ptr(ws["O"]), ptr(target), ptr(probs), ptr(bw["dO"]), ptr(ws["loss_part"]), n, 1.0 / n,
                 _lib.stream())
        else:
            call("wn_chunk_softmax_ce", ptr(ws["O"]), ptr(target), ptr(probs), ptr(bw["dO"]), ptr(ws["loss_part"]), n, self.Q,
                 1.0 / n, _lib.stream())
        self.backward_from_dlogits(ws)
        loss = ws["loss_part"].sum()
        self._throttle.leave()
        return loss

    def loss_and_grad_codes(self, codes, target, scrambled=True, want_probs=False):
        """the fast engine's entry point on integer codes; here the one-hot is built (wn_onehot) and the dense path runs"""
        return self.loss_and_grad(self.onehot(codes, scrambled), target, want_probs)

    def adam_init(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.adam_state = dict(m=torch.zeros_like(self.flat), v=torch.zeros_like(self.flat), t=0,
                               lr=lr, b1=betas[0], b2=betas[1], eps=eps)

    def adam_step(self, gscale=1.0):
        s = self.adam_state
        s["t"] += 1
        call("wn_adam_flat", ptr(self.flat), ptr(self.flat_grad), ptr(s["m"]), ptr(s["v"]), self.spec.total,
             s["lr"], s["b1"], s["b2"], s["eps"], 1.0 - s["b1"] ** s["t"], 1.0 - s["b2"] ** s["t"], gscale, _lib.stream())

    def onehot(self, codes, scrambled=True):
        """int32 (B,T) codes on the device -> float32 (B,Q,T) (faster_audio_data.py:62-83)."""
        B, T = codes.shape
        out = torch.empty(B, self.Q, T, dtype=torch.float32, device=self.device)
        call("wn_onehot", ptr(codes), ptr(out), B, self.Q, T, 1 if scrambled else 0, _lib.stream())
        return out
