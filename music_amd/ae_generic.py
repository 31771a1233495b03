"""The GENERAL execution plan of the autoencoder: every constructor argument the reference accepts
(wavenet_autoencoder/model1.py:14-31).

`model1._AutoencoderEngine` drives the specialised kernels (fused encoder / decoder blocks, one-launch backward blocks,
conditioning on the matrix cores) and covers filter_width == 2, quantization_channel == 256 and up to 64 residual /
dilation channels on both sides - what the reference ships and BASELINE.json names.  This engine runs the same arithmetic
(model1.py:137-268 forward, its autograd backward: SURVEY Appendix B) for ANY filter width, quantisation width and
channel counts out of the library's general kernels, one product per launch - the autoencoder counterpart of
music_amd/engine_generic.py:

    conv with k taps          wn_chan_gemm, two taps per launch (input shifted by -(k-1-j) d), further pairs accumulate
                              through `resid`; ReLU on the input / mask on the output where the encoder needs them
    conditioning (_conditon)  wn_cond_expand of the per-clip table into the conv's output buffer (the products then
                              accumulate onto it), wn_cond_grad for its gradient (bucket sums)
    gate                      wn_gate_fwd / wn_gate_bwd (rows [filter | gate]: the pack puts model1.py:188-190's halves there)
    pooled bottleneck         wn_avgpool / wn_avgpool_bwd
    chunk softmax (+ CE)      wn_chunk_softmax256_* when Q == 256, wn_chunk_softmax_* otherwise
    weight gradients          wn_wgrad slabs + wn_reduce_slabs (bit-reproducible), biases wn_bias_grad

Same x3 arithmetic, HBM layout (absolute time, one pitch, channels padded to 32 with zero weights), flat parameter /
gradient buffers and workspace pool as the fast engine.  The only arithmetic left to torch is what the fast engine leaves
there too: the N + 1 conditioning projections of the pooled encoding ((C x Bw) . (Bw x Le) per clip) and their transposes
in the backward.  PyTorch is otherwise used for device memory and streams only.  Nothing here imports oracle/.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from ._lib import call, ptr
from .engine import SLACK, PAD_BACK, WorkspacePool, _Spec, _pad, pack_index


class GenericAutoencoderEngine:
    def __init__(self, net, device, mode="f16x3", mode_bwd="bf16x3"):
        self.net, self.device = net, device
        self.mode = _lib.MODE_NAMES[mode]
        self.mode_b = _lib.MODE_NAMES[mode_bwd]
        self.dil = [int(d) for d in net.dilations]
        self.N = len(self.dil)
        self.k = int(net.filter_width)
        if self.k < 1:
            raise ValueError("filter_width must be >= 1")
        self.Q = int(net.quantization_channel)
        self.Re, self.De, self.Bw, self.pool = (net.en_residual_channel, net.en_dilation_channel, net.en_bottleneck_width,
                                                net.en_pool_kernel_size)
        self.Rd, self.Dd, self.Sd = net.de_residual_channel, net.de_dilation_channel, net.de_skip_channel
        self.ReP, self.DeP, self.BwP, self.RdP, self.DdP, self.SP, self.QP = (
            _pad(v, 32) for v in (self.Re, self.De, self.Bw, self.Rd, self.Dd, self.Sd, self.Q))
        self.rf = (self.k - 1) * (sum(self.dil) + 1) + 1
        self.off = [self.k - 1]
        for d in self.dil:
            self.off.append(self.off[-1] + (self.k - 1) * d)
        assert self.off[-1] == self.rf - 1
        self.pairs = [(j, j + 1 if j + 1 < self.k else None) for j in range(0, self.k, 2)]
        self.use_bias = bool(net.use_bias)
        named = list(net.named_parameters())
        self.param_names = [n for n, _ in named]
        self.spec = _Spec([(n, tuple(p.shape)) for n, p in named])
        self.flat = torch.zeros(self.spec.total, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(self.spec.total, dtype=torch.float32, device=device)
        with torch.no_grad():
            for n, p in named:
                o = self.spec.off[n]
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view                                # the module's parameters now ARE the flat buffer
        self._build_packs()
        self._ws = WorkspacePool(self._make_workspace)
        self._gen = 0
        self.adam_state = None
        self.marks = None

    def mark(self, name):
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev))

    def _bias(self, name):
        return ptr(self.flat, self.spec.off[name + ".bias"]) if self.use_bias else None

    def _taps(self, pair, d):
        """(shift of tap j0, shift of tap j1 or 0, 1 if there is a second tap): input column = t - (k-1-j) d"""
        j0, j1 = pair
        return -(self.k - 1 - j0) * d, (-(self.k - 1 - j1) * d if j1 is not None else 0), (1 if j1 is not None else 0)

    # ------------------------------------------------------------------ packs and gradient maps
    def _build_packs(self):
        sp, N, k = self.spec, self.N, self.k
        Re, De, Bw, Rd, Dd, Sd, Q = self.Re, self.De, self.Bw, self.Rd, self.Dd, self.Sd, self.Q
        ReP, DeP, BwP, RdP, DdP, SP, QP = self.ReP, self.DeP, self.BwP, self.RdP, self.DdP, self.SP, self.QP
        fwd, bwd = [], []
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

        def conv_k(pname, tag, rows, rows_p, cols, cols_p, row_map=None, with_T=True):
            """A k-tap conv weight [rows][cols][k] as per-pair forward packs "<tag>_<p>" (rows_p x taps * cols_p), their transposes
            "<tag>T_<p>" (cols_p x taps * rows_p) and the gradient map of every tap.  row_map: pack row of weight row r."""
            w3 = sp.conv(pname)
            rm = np.arange(rows) if row_map is None else row_map
            gm = np.zeros((rows, cols, k), dtype=np.int64)
            for p, pair in enumerate(self.pairs):
                tp = taps_of(pair)
                w = full(rows_p, len(tp) * cols_p)
                wt = full(cols_p, len(tp) * rows_p)
                o0 = add_gp("%s_%d" % (tag, p), rows_p, len(tp) * cols_p)
                for tl, j in enumerate(tp):
                    w[rm, tl * cols_p:tl * cols_p + cols] = w3[:, :, j]
                    wt[:cols, tl * rows_p + rm] = w3[:, :, j].T
                    gm[:, :, j] = o0 + rm[:, None] * (len(tp) * cols_p) + tl * cols_p + np.arange(cols)[None, :]
                fwd.append(("%s_%d" % (tag, p), pack_index(w)))
                if with_T:
                    bwd.append(("%sT_%d" % (tag, p), pack_index(wt)))
            put(pname, gm)

        def conv_1(pname, tag, rows, rows_p, cols, cols_p, col0=0, ncols_p=None, shared=None):
            """A 1x1 conv weight [rows][cols][1] as a forward pack "<tag>" and its transpose "<tag>T"; `shared` = (name, total
            padded columns, column offset): a slice of a wider matrix (the skip convs stacked on the K axis)."""
            w2 = sp.conv(pname)[:, :, 0]
            if shared is None:
                w = full(rows_p, cols_p)
                w[:rows, :cols] = w2
                fwd.append((tag, pack_index(w)))
                bwd.append((tag + "T", pack_index(np.ascontiguousarray(w.T))))
                o0 = add_gp(tag, rows_p, cols_p)
                put(pname, o0 + np.arange(rows)[:, None] * cols_p + np.arange(cols)[None, :])
            return w2

        # encoder (model1.py:137-156)
        conv_k("en_causal_layer.weight", "en_causal", Re, ReP, Q, QP)       # (its transpose: input_grad)
        for i in range(N):
            conv_k("en_dilation_layer_stack.%d.weight" % i, "en_dil%d" % i, De, DeP, Re, ReP)
            conv_1("en_dense_layer_stack.%d.weight" % i, "en_dense%d" % i, Re, ReP, De, DeP)
        conv_1("bottleneck_layer.weight", "bottleneck", Bw, BwP, Re, ReP)
        # decoder (model1.py:158-225): filter_gate rows are [gate (Dd) | filter (Dd)] in the reference (:188-190); the pack puts
        # filter first, gate second - the [f | g] order of wn_gate_fwd / wn_gate_bwd
        conv_k("de_causal_layer.weight", "de_causal", Rd, RdP, Q, QP)
        self.fg_rows = np.concatenate([DdP + np.arange(Dd), np.arange(Dd)])          # reference row r -> pack row
        for i in range(N):
            conv_k("de_dilation_layer_stack.%d.weight" % (3 * i), "de_fg%d" % i, 2 * Dd, 2 * DdP, Rd, RdP, row_map=self.fg_rows)
            conv_1("de_dilation_layer_stack.%d.weight" % (3 * i + 1), "de_d%d" % i, Rd, RdP, Dd, DdP)
        w = full(SP, N * DdP)
        o0 = add_gp("skip", SP, N * DdP)
        for i in range(N):
            pname = "de_dilation_layer_stack.%d.weight" % (3 * i + 2)
            w[:Sd, i * DdP:i * DdP + Dd] = sp.conv(pname)[:, :, 0]
            put(pname, o0 + np.arange(Sd)[:, None] * (N * DdP) + i * DdP + np.arange(Dd)[None, :])
        fwd.append(("skip", pack_index(w)))
        bwd.append(("skipT", pack_index(np.ascontiguousarray(w.T))))
        conv_1("connection_1.weight", "c1", Sd, SP, Sd, SP)
        conv_1("connection_2.weight", "c2", Q, QP, Sd, SP)
        self.gp_bias_off = {}
        if self.use_bias:
            for name in self.param_names:
                if name.endswith(".bias"):
                    n = sp.shape[name][0]
                    self.gp_bias_off[name[:-5]] = go[0]
                    if name.startswith("de_dilation_layer_stack.") and int(name.split(".")[1]) % 3 == 0:
                        # filter_gate bias: its gradient rows come in the pack's [f | g] order, each half apart
                        put(name, go[0] + np.concatenate([DdP + np.arange(Dd), np.arange(Dd)]))
                        go[0] += 2 * DdP
                    else:
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
        self.pk_f_off, self.pk_f_idx, self.pk_f = finish(fwd, self.mode)
        self.pk_b_off, self.pk_b_idx, self.pk_b = finish(bwd, self.mode_b)
        if self.use_bias:
            # the filter_gate biases of every block in the pack's padded [f | g] row order, gathered from the flat buffer
            bi = np.full((N, 2 * DdP), -1, dtype=np.int64)
            for i in range(N):
                o = sp.off["de_dilation_layer_stack.%d.bias" % (3 * i)]
                bi[i, :Dd] = o + Dd + np.arange(Dd)                          # filter = second half of the reference's rows
                bi[i, DdP:DdP + Dd] = o + np.arange(Dd)
            self.bfg_idx = torch.from_numpy(bi.reshape(-1).astype(np.int32)).to(dev)
            self.bfg = torch.zeros(N * 2 * DdP, dtype=torch.float32, device=dev)

    def pack_weights(self):
        st = _lib.stream()
        call("wn_pack_weights", ptr(self.flat), ptr(self.pk_f_idx), ptr(self.pk_f), self.pk_f_idx.numel(), self.mode, st)
        call("wn_pack_weights", ptr(self.flat), ptr(self.pk_b_idx), ptr(self.pk_b), self.pk_b_idx.numel(), self.mode_b, st)
        if self.use_bias:
            call("wn_gather_grads", ptr(self.flat), ptr(self.bfg_idx), ptr(self.bfg), self.bfg.numel(), st)

    # ------------------------------------------------------------------ workspace
    def workspace(self, B, T):
        return self._ws.peek(B, T)

    def _make_workspace(self, B, T):
        dev = self.device
        pitch = _pad(T, 256) + 512
        W = T - self.rf + 1
        N = self.N

        def buf(rows):
            return torch.zeros(SLACK + B * rows * pitch + PAD_BACK, dtype=torch.float32, device=dev)
        ws = dict(B=B, T=T, W=W, pitch=pitch, bwd=None)
        ws["Xe"], ws["He"] = buf((N + 1) * self.ReP), buf(N * self.DeP)
        ws["E"] = buf(self.BwP)
        ws["Xd"], ws["FG"], ws["Z"] = buf((N + 1) * self.RdP), buf(N * 2 * self.DdP), buf(N * self.DdP)
        ws["U"], ws["R1"] = buf(self.SP), buf(self.SP)
        ws["O"] = torch.zeros(B * self.Q * W + 32 * W + PAD_BACK, dtype=torch.float32, device=dev)
        if self.QP != self.Q:
            ws["Xin"] = torch.zeros(B * self.QP * T + PAD_BACK, dtype=torch.float32, device=dev)
        return ws

    # layer i of a per-layer stacked buffer laid out [layer][clip][rows][pitch]
    def _lay(self, ws, key, i, rows):
        return ptr(ws[key], SLACK + i * ws["B"] * rows * ws["pitch"])

    def _gemm(self, st, B, mode, wpack, in0, in1, in_bs, in_pitch, in_lo, in_hi, s0, s1, ks0, ks1, mt, m_valid, out, out_bs,
              out_pitch, out_shift, bias, resid, mask, t_lo, t_hi, relu_in):
        call("wn_chan_gemm", in0, in1, in_bs, in_pitch, in_lo, in_hi, s0, s1, ks0, ks1, wpack, mt, m_valid,
             out, out_bs, out_pitch, out_shift, bias, resid[0], resid[1], resid[2], resid[3],
             mask[0], mask[1], mask[2], t_lo, t_hi, relu_in, B, mode, st)

    def _cond_modes(self, T, Le):
        """(mode, q) of _conditon (model1.py:227-247) on every decoder block's output and on the epilogue: a length that the
        number of pooled frames divides takes the stretch branch (1, L / Le), any other the tile branch (2, -)."""
        out = []
        for i in range(self.N + 1):
            L = T - (self.off[i + 1] if i < self.N else self.rf - 1)
            out.append((1, L // Le) if L % Le == 0 else (2, 1))
        return out

    # ------------------------------------------------------------------ forward (model1.py:256-268)
    def forward(self, x, cond, want_probs=True):
        B, Q, T = x.shape
        assert Q == self.Q and x.is_contiguous() and x.dtype == torch.float32 and x.is_cuda
        W = T - self.rf + 1
        if W <= 0:
            raise ValueError("wave sample not long enough")
        Le = W // self.pool
        if Le < 1:
            raise RuntimeError("Output size is too small: %d samples of encoding cannot be pooled by %d" % (W, self.pool))
        ws = self._ws.get(B, T)
        self._gen += 1
        ws["gen"], ws["x_in"], ws["x_ver"], ws["Le"] = self._gen, x, x._version, Le
        st = _lib.stream()
        N, k, pitch, m = self.N, self.k, ws["pitch"], self.mode
        ReP, DeP, BwP, RdP, DdP, SP, QP = self.ReP, self.DeP, self.BwP, self.RdP, self.DdP, self.SP, self.QP
        dev = self.device
        cw = torch.stack([c[0][:, :, 0] for c in cond[:N]]).to(dev)          # (N, 2Dd, Bw) reference row order [gate | filter]
        cb = torch.stack([c[1] for c in cond[:N]]).to(dev)
        cfw, cfb = cond[N][0].to(dev), cond[N][1].to(dev)
        self.pack_weights()
        fr = lambda name: ptr(self.pk_f, self.pk_f_off[name])
        NONE4, NONE3 = (None, 0, 0, 0), (None, 0, 0)
        lo, k0 = self.rf - 1, k - 1
        gemm = lambda pack, *a: self._gemm(st, B, m, fr(pack), *a)
        if QP != Q:                                          # K runs over whole 32-row steps: a zero-padded copy of the input
            xin = ws["Xin"][:B * QP * T].view(B, QP, T)
            xin[:, :Q].copy_(x)
            xin_p, xin_bs = ptr(ws["Xin"]), QP * T
        else:
            xin_p, xin_bs = ptr(x), Q * T
        ws["xin"] = (xin_p, xin_bs)

        def causal(tag, rows, rows_p, out, out_bs, bias):
            for p, pair in enumerate(self.pairs):
                s0, s1, two = self._taps(pair, 1)
                gemm("%s_%d" % (tag, p), xin_p, xin_p if two else None, xin_bs, T, 0, T, s0, s1, QP // 32, QP // 32 if two else 0,
                     rows_p // 16, rows, out, out_bs, pitch, 0, bias if p == 0 else None,
                     (out, out_bs, pitch, k0) if p else NONE4, NONE3, k0, T, 0)

        # ---------------- encoder: relu -> dilated conv -> relu -> 1x1, residual; every x_i and h_i is kept for the backward
        eb, hb = ReP * pitch, DeP * pitch
        xe = lambda i: self._lay(ws, "Xe", i, ReP)
        he = lambda i: self._lay(ws, "He", i, DeP)
        causal("en_causal", self.Re, ReP, xe(0), eb, self._bias("en_causal_layer"))
        for i, d in enumerate(self.dil):
            t_in, t_lo = self.off[i], self.off[i + 1]
            for p, pair in enumerate(self.pairs):
                s0, s1, two = self._taps(pair, d)
                gemm("en_dil%d_%d" % (i, p), xe(i), xe(i) if two else None, eb, pitch, t_in, T, s0, s1, ReP // 32,
                     ReP // 32 if two else 0, DeP // 16, self.De, he(i), hb, pitch, 0,
                     self._bias("en_dilation_layer_stack.%d" % i) if p == 0 else None,
                     (he(i), hb, pitch, t_lo) if p else NONE4, NONE3, t_lo, T, 1)
            gemm("en_dense%d" % i, he(i), None, hb, pitch, t_lo, T, 0, 0, DeP // 32, 0, ReP // 16, self.Re, xe(i + 1), eb, pitch, 0,
                 self._bias("en_dense_layer_stack.%d" % i), (xe(i), eb, pitch, t_lo), NONE3, t_lo, T, 1)
        self.mark("enc_stack_fwd")
        E = ptr(ws["E"], SLACK)
        gemm("bottleneck", xe(N), None, eb, pitch, lo, T, 0, 0, ReP // 32, 0, BwP // 16, self.Bw, E, BwP * pitch, pitch, 0,
             self._bias("bottleneck_layer"), NONE4, NONE3, lo, T, 0)
        enc = torch.empty(B, self.Bw, Le, dtype=torch.float32, device=dev)
        call("wn_avgpool", E, BwP * pitch, pitch, lo, self.pool, Le, self.Bw, ptr(enc), self.Bw * Le, Le, B, st)

        # ---------------- conditioning tables en_i = Conv1d_rand(enc) (model1.py:178-179, 216-217), rows in the pack's [f | g] order
        Dd, Sd = self.Dd, self.Sd
        en = torch.einsum("nck,bkl->nbcl", cw, enc) + cb[:, None, :, None]             # (N, B, 2Dd, Le)
        tab = torch.zeros(N, B, 2 * DdP, Le, dtype=torch.float32, device=dev)
        tab[:, :, :Dd] = en[:, :, Dd:]
        tab[:, :, DdP:DdP + Dd] = en[:, :, :Dd]
        enf = F.conv1d(enc, cfw, cfb).contiguous()                                     # (B, Sd, Le)
        cmodes = self._cond_modes(T, Le)
        ws.update(enc=enc, cw=cw, cfw=cfw, cmodes=cmodes)

        # ---------------- decoder
        db, fb, zb, sb = RdP * pitch, 2 * DdP * pitch, N * DdP * pitch, SP * pitch
        xd = lambda i: self._lay(ws, "Xd", i, RdP)
        fg = lambda i: self._lay(ws, "FG", i, 2 * DdP)
        causal("de_causal", self.Rd, RdP, xd(0), db, self._bias("de_causal_layer"))
        bn = "de_dilation_layer_stack.%d"
        for i, d in enumerate(self.dil):
            t_in, t_lo = self.off[i], self.off[i + 1]
            mode_c, q = cmodes[i]
            # [f; g] = conditioning (expanded over time) + sum_j W_j x_i[t - (k-1-j) d] (+ bias)
            call("wn_cond_expand", ptr(tab[i]), 2 * DdP * Le, Le, 2 * DdP, t_lo, T, mode_c, Le, q, fg(i), fb, pitch, B, st)
            for p, pair in enumerate(self.pairs):
                s0, s1, two = self._taps(pair, d)
                gemm("de_fg%d_%d" % (i, p), xd(i), xd(i) if two else None, db, pitch, t_in, T, s0, s1, RdP // 32,
                     RdP // 32 if two else 0, 2 * DdP // 16, 2 * DdP, fg(i), fb, pitch, 0,
                     ptr(self.bfg, i * 2 * DdP) if (self.use_bias and p == 0) else None, (fg(i), fb, pitch, t_lo), NONE3, t_lo, T, 0)
            z_i = ptr(ws["Z"], SLACK + i * DdP * pitch)
            call("wn_gate_fwd", fg(i), fb, DdP, DdP, z_i, zb, pitch, t_lo, T, B, st)
            if i < N - 1:
                gemm("de_d%d" % i, z_i, None, zb, pitch, t_lo, T, 0, 0, DdP // 32, 0, RdP // 16, self.Rd, xd(i + 1), db, pitch, 0,
                     self._bias(bn % (3 * i + 1)), (xd(i), db, pitch, t_lo), NONE3, t_lo, T, 0)
        self.mark("dec_stack_fwd")
        U, R1 = ptr(ws["U"], SLACK), ptr(ws["R1"], SLACK)
        bias_s = None
        if self.use_bias:
            o = self.spec.off
            ws["bias_skip"] = sum(self.flat[o[bn % (3 * i + 2) + ".bias"]:o[bn % (3 * i + 2) + ".bias"] + Sd] for i in range(N)).contiguous()
            bias_s = ptr(ws["bias_skip"])
        gemm("skip", ptr(ws["Z"], SLACK), None, zb, pitch, lo, T, 0, 0, N * DdP // 32, 0, SP // 16, Sd, U, sb, pitch, 0, bias_s,
             NONE4, NONE3, lo, T, 0)
        mode_c, q = cmodes[N]
        call("wn_cond_expand", ptr(enf), Sd * Le, Le, Sd, lo, T, mode_c, Le, q, R1, sb, pitch, B, st)
        gemm("c1", U, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, SP // 16, Sd, R1, sb, pitch, 0, self._bias("connection_1"),
             (R1, sb, pitch, lo), NONE3, lo, T, 1)
        gemm("c2", R1, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, QP // 16, Q, ptr(ws["O"]), Q * W, W, -lo,
             self._bias("connection_2"), NONE4, NONE3, lo, T, 1)
        probs = None
        if want_probs:
            probs = torch.empty(B * W, Q, dtype=torch.float32, device=dev)
            if Q == 256:
                call("wn_chunk_softmax256_fwd", ptr(ws["O"]), ptr(probs), B * W, st)
            else:
                call("wn_chunk_softmax_fwd", ptr(ws["O"]), ptr(probs), B * W, Q, st)
        ws["probs"] = probs
        self.mark("epilogue_fwd")
        return probs, enc, ws

    # ------------------------------------------------------------------ backward
    def _bwd_workspace(self, ws):
        if ws["bwd"] is not None:
            return ws["bwd"]
        B, T, W, pitch, dev = ws["B"], ws["T"], ws["W"], ws["pitch"], self.device
        N, Q = self.N, self.Q

        def buf(rows):
            return torch.zeros(SLACK + B * rows * pitch + PAD_BACK, dtype=torch.float32, device=dev)
        bw = dict(dO=torch.zeros(B * Q * W + 32 * W + PAD_BACK, dtype=torch.float32, device=dev), dR1=buf(self.SP), dU=buf(self.SP),
                  dZ=buf(N * self.DdP), dXd=[buf(self.RdP), buf(self.RdP)], dz=buf(self.DdP), dfg=buf(2 * self.DdP),
                  dE=buf(self.BwP), dXe=[buf(self.ReP), buf(self.ReP)], dHe=buf(self.DeP))
        lo, np_ = self.rf - 1, len(self.pairs)
        ops = [("c2", lo, 1024), ("c1", lo, 1024), ("skip", lo, 2048)]
        for i in range(N):
            ops += [("de_fg%d_%d" % (i, p), self.off[i + 1], 512) for p in range(np_)]
            if i < N - 1:
                ops.append(("de_d%d" % i, self.off[i + 1], 512))
        ops += [("de_causal_%d" % p, self.k - 1, 512) for p in range(np_)]
        ops.append(("bottleneck", lo, 512))
        for i in range(N):
            ops += [("en_dil%d_%d" % (i, p), self.off[i + 1], 512) for p in range(np_)]
            ops.append(("en_dense%d" % i, self.off[i + 1], 512))
        ops += [("en_causal_%d" % p, self.k - 1, 512) for p in range(np_)]
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

    def input_grad(self, ws):
        """Gradient of the last backward w.r.t. the module's INPUT (the encoder's and the decoder's causal conv, model1.py:137,158):
        din[q][s] = sum over both layers, taps j and rows r of  Wc_j[r][q] dx0[r][s + (k-1-j)],  dx0 living on [k-1, T)."""
        bw = ws["bwd"]
        if bw is None:
            raise RuntimeError("music_amd: input_grad() needs the backward of this forward to have run")
        B, T, pitch, QP, Q = ws["B"], ws["T"], ws["pitch"], self.QP, self.Q
        din = None
        for tag, rp, key in (("de_causalT", self.RdP, "dXd"), ("en_causalT", self.ReP, "dXe")):
            dx0 = ptr(bw[key][0], SLACK)
            for p, pair in enumerate(self.pairs):
                s0, s1, two = self._taps(pair, 1)
                out = torch.empty(B, Q, T, dtype=torch.float32, device=self.device)
                self._gemm(_lib.stream(), B, self.mode_b, ptr(self.pk_b, self.pk_b_off["%s_%d" % (tag, p)]), dx0, dx0 if two else None,
                           rp * pitch, pitch, self.k - 1, T, -s0, -s1, rp // 32, rp // 32 if two else 0, QP // 16, Q, ptr(out), Q * T, T, 0,
                           None, (None, 0, 0, 0), (None, 0, 0), 0, T, 0)
                din = out if din is None else din.add_(out)
        return din

    def backward(self, ws, dprobs):
        """Fills self.flat_grad from d loss / d probabilities (B*W, Q); dprobs None = bw["dO"] already holds d loss / d logits."""
        bw = self._bwd_workspace(ws)
        st = _lib.stream()
        B, T, W, pitch, Le = ws["B"], ws["T"], ws["W"], ws["pitch"], ws["Le"]
        N, k, Q, mb = self.N, self.k, self.Q, self.mode_b
        ReP, DeP, BwP, RdP, DdP, SP, QP = self.ReP, self.DeP, self.BwP, self.RdP, self.DdP, self.SP, self.QP
        Dd, Sd, Bw = self.Dd, self.Sd, self.Bw
        dev = self.device
        if ws.get("x_ver") is not None and ws["x_in"]._version != ws["x_ver"]:
            raise RuntimeError("music_amd: the input of this forward was modified in place before backward()")
        if dprobs is not None:
            dprobs = dprobs.contiguous()
            if Q == 256:
                call("wn_chunk_softmax256_bwd", ptr(ws["probs"]), ptr(dprobs), ptr(bw["dO"]), B * W, st)
            else:
                call("wn_chunk_softmax_bwd", ptr(ws["probs"]), ptr(dprobs), ptr(bw["dO"]), B * W, Q, st)
        br = lambda name: ptr(self.pk_b, self.pk_b_off[name])
        NONE4, NONE3 = (None, 0, 0, 0), (None, 0, 0)
        lo, k0 = self.rf - 1, k - 1
        plan = bw["plan"]
        gemm = lambda pack, *a: self._gemm(st, B, mb, br(pack), *a)

        def wgrad(name, *args):
            """args = wn_wgrad's arguments up to relu_b, then ldc, t_lo, t_hi"""
            so, n, chunk = plan[name]
            head, (ldc, t_lo, t_hi) = args[:-3], args[-3:]
            call("wn_wgrad", *head, ptr(bw["slab"], so), ldc, n, t_lo, t_hi, chunk, B, mb, st)

        def bias_grad(name, a, a_bs, a_pitch, a_shift, rows, t_lo, dst=0):
            if self.use_bias:
                call("wn_bias_grad", a, a_bs, a_pitch, a_shift, rows, t_lo, T, B, ptr(self.gpack, self.gp_bias_off[name] + dst), st)
        db, fb, zb, sb, eb, hb = RdP * pitch, 2 * DdP * pitch, N * DdP * pitch, SP * pitch, ReP * pitch, DeP * pitch
        dO, dR1, dU, dZ = ptr(bw["dO"]), ptr(bw["dR1"], SLACK), ptr(bw["dU"], SLACK), ptr(bw["dZ"], SLACK)
        U, R1, Z = ptr(ws["U"], SLACK), ptr(ws["R1"], SLACK), ptr(ws["Z"], SLACK)
        cmodes = ws["cmodes"]
        # ---- epilogue: o = C2 relu(r1), r1 = C1 relu(u) + cond, u = sum_i Ws_i z_i
        wgrad("c2", dO, Q * W, W, -lo, W, R1, None, sb, pitch, 0, 0, pitch, SP // 16, QP // 16, 1, SP, lo, T)
        gemm("c2T", dO, None, Q * W, W, 0, W, -lo, 0, QP // 32, 0, SP // 16, Sd, dR1, sb, pitch, 0, None, NONE4, (R1, sb, pitch), lo, T, 0)
        mode_c, q = cmodes[N]
        d_enf = torch.zeros(B, Sd, Le, dtype=torch.float32, device=dev)
        call("wn_cond_grad", dR1, sb, pitch, Sd, lo, T, mode_c, Le, q, ptr(d_enf), Sd * Le, Le, B, st)
        wgrad("c1", dR1, sb, pitch, 0, pitch, U, None, sb, pitch, 0, 0, pitch, SP // 16, SP // 16, 1, SP, lo, T)
        gemm("c1T", dR1, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, SP // 16, Sd, dU, sb, pitch, 0, None, NONE4, (U, sb, pitch), lo, T, 0)
        wgrad("skip", dU, sb, pitch, 0, pitch, Z, None, zb, pitch, 0, 0, pitch, N * DdP // 16, SP // 16, 0, N * DdP, lo, T)
        gemm("skipT", dU, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, N * DdP // 16, N * DdP, dZ, zb, pitch, 0, None, NONE4, NONE3, lo, T, 0)
        bn = "de_dilation_layer_stack.%d"
        bias_grad("connection_2", dO, Q * W, W, -lo, Q, lo)
        bias_grad("connection_1", dR1, sb, pitch, 0, Sd, lo)
        for i in range(N if self.use_bias else 0):
            bias_grad(bn % (3 * i + 2), dU, sb, pitch, 0, Sd, lo)
        self.mark("epilogue_bwd")
        # ---- decoder blocks
        xd = lambda i: self._lay(ws, "Xd", i, RdP)
        fg = lambda i: self._lay(ws, "FG", i, 2 * DdP)
        dfg, dzb = ptr(bw["dfg"], SLACK), ptr(bw["dz"], SLACK)
        d_tab = torch.zeros(N, B, 2 * DdP, Le, dtype=torch.float32, device=dev)
        for i in range(N - 1, -1, -1):
            d, t_in, t_lo = self.dil[i], self.off[i], self.off[i + 1]
            z_i, dz_i = ptr(ws["Z"], SLACK + i * DdP * pitch), ptr(bw["dZ"], SLACK + i * DdP * pitch)
            dy = ptr(bw["dXd"][(i + 1) % 2], SLACK) if i < N - 1 else None
            if dy is not None:                               # dz = Wd^T dy + dz_crop
                wgrad("de_d%d" % i, dy, db, pitch, 0, pitch, z_i, None, zb, pitch, 0, 0, pitch, DdP // 16, RdP // 16, 0, DdP, t_lo, T)
                gemm("de_d%dT" % i, dy, None, db, pitch, t_lo, T, 0, 0, RdP // 32, 0, DdP // 16, DdP, dzb, DdP * pitch, pitch, 0, None,
                     (dz_i, zb, pitch, lo), NONE3, t_lo, T, 0)
                dz_p, dz_bs = dzb, DdP * pitch
                bias_grad(bn % (3 * i + 1), dy, db, pitch, 0, self.Rd, t_lo)
            else:                                            # the last block's x_N is unused: only the skip path reaches z
                dz_p, dz_bs = dz_i, zb
            call("wn_gate_bwd", fg(i), fb, DdP, DdP, dz_p, dz_bs, dfg, fb, pitch, t_lo, T, B, st)
            bias_grad(bn % (3 * i), dfg, fb, pitch, 0, 2 * DdP, t_lo)          # rows [f | g] of the pack (the gather map un-permutes)
            mode_c, q = cmodes[i]
            call("wn_cond_grad", dfg, fb, pitch, 2 * DdP, t_lo, T, mode_c, Le, q, ptr(d_tab[i]), 2 * DdP * Le, Le, B, st)
            out = ptr(bw["dXd"][i % 2], SLACK)
            for p, pair in enumerate(self.pairs):
                s0, s1, two = self._taps(pair, d)
                ntap = 2 if two else 1
                wgrad("de_fg%d_%d" % (i, p), dfg, fb, pitch, 0, pitch, xd(i), xd(i) if two else None, db, pitch, s0, s1, pitch,
                      RdP // 16, 2 * DdP // 16, 0, ntap * RdP, t_lo, T)
                resid = (out, db, pitch, t_in) if p else ((dy, db, pitch, t_lo) if dy is not None else NONE4)
                gemm("de_fg%dT_%d" % (i, p), dfg, dfg if two else None, fb, pitch, t_lo, T, -s0, -s1, 2 * DdP // 32,
                     2 * DdP // 32 if two else 0, RdP // 16, self.Rd, out, db, pitch, 0, None, resid, NONE3, t_in, T, 0)
        self.mark("dec_stack_bwd")
        xin_p, xin_bs = ws["xin"]

        def causal_wgrad(tag, dx0, bs, rows_p):
            for p, pair in enumerate(self.pairs):
                s0, s1, two = self._taps(pair, 1)
                wgrad("%s_%d" % (tag, p), dx0, bs, pitch, 0, pitch, xin_p, xin_p if two else None, xin_bs, T, s0, s1, T, QP // 16,
                      rows_p // 16, 0, (2 if two else 1) * QP, k0, T)
        dxd0 = ptr(bw["dXd"][0], SLACK)
        causal_wgrad("de_causal", dxd0, db, RdP)
        bias_grad("de_causal_layer", dxd0, db, pitch, 0, self.Rd, k0)
        # ---- conditioning projections (unregistered, no gradient of their own): d enc = sum_i cw_i^T d en_i + cfw^T d enf
        d_en = torch.cat([d_tab[:, :, DdP:DdP + Dd], d_tab[:, :, :Dd]], 2)             # reference row order [gate | filter]
        d_enc = (torch.einsum("nck,nbcl->bkl", ws["cw"], d_en) + torch.einsum("ck,bcl->bkl", ws["cfw"][:, :, 0], d_enf)).contiguous()
        # ---- encoder: avgpool -> bottleneck -> N blocks -> causal
        dE = ptr(bw["dE"], SLACK)
        call("wn_avgpool_bwd", ptr(d_enc), Bw * Le, Le, lo, self.pool, Le, Bw, dE, BwP * pitch, pitch, T, B, st)
        xe = lambda i: self._lay(ws, "Xe", i, ReP)
        he = lambda i: self._lay(ws, "He", i, DeP)
        wgrad("bottleneck", dE, BwP * pitch, pitch, 0, pitch, xe(N), None, eb, pitch, 0, 0, pitch, ReP // 16, BwP // 16, 0, ReP, lo, T)
        bias_grad("bottleneck_layer", dE, BwP * pitch, pitch, 0, Bw, lo)
        dxe = [ptr(t, SLACK) for t in bw["dXe"]]
        dHe = ptr(bw["dHe"], SLACK)
        gemm("bottleneckT", dE, None, BwP * pitch, pitch, lo, T, 0, 0, BwP // 32, 0, ReP // 16, self.Re, dxe[N % 2], eb, pitch, 0, None,
             NONE4, NONE3, lo, T, 0)
        for i in range(N - 1, -1, -1):
            d, t_in, t_lo = self.dil[i], self.off[i], self.off[i + 1]
            y_lo = lo if i == N - 1 else t_lo                     # the top gradient only exists on the crop [rf - 1, T)
            dy = dxe[(i + 1) % 2]
            # x_{i+1} = Wdense relu(h) + x_i[t]:  dWdense = sum dy relu(h)^T,  dh = (Wdense^T dy) * [h > 0]
            wgrad("en_dense%d" % i, dy, eb, pitch, 0, pitch, he(i), None, hb, pitch, 0, 0, pitch, DeP // 16, ReP // 16, 1, DeP, y_lo, T)
            bias_grad("en_dense_layer_stack.%d" % i, dy, eb, pitch, 0, self.Re, y_lo)
            gemm("en_dense%dT" % i, dy, None, eb, pitch, y_lo, T, 0, 0, ReP // 32, 0, DeP // 16, self.De, dHe, hb, pitch, 0, None, NONE4,
                 (he(i), hb, pitch), t_lo, T, 0)
            bias_grad("en_dilation_layer_stack.%d" % i, dHe, hb, pitch, 0, self.De, t_lo)
            out = dxe[i % 2]
            for p, pair in enumerate(self.pairs):
                s0, s1, two = self._taps(pair, d)
                ntap = 2 if two else 1
                # h = sum_j Wdil_j relu(x_i)[t - (k-1-j) d]:  dWdil_j = sum dh relu(x_i)[t - (k-1-j) d]^T
                wgrad("en_dil%d_%d" % (i, p), dHe, hb, pitch, 0, pitch, xe(i), xe(i) if two else None, eb, pitch, s0, s1, pitch,
                      ReP // 16, DeP // 16, 1, ntap * ReP, t_lo, T)
                # dx_i[t] = [x_i > 0] sum_j Wdil_j^T dh[t + (k-1-j) d] + dy[t]
                resid = (out, eb, pitch, t_in) if p else (dy, eb, pitch, y_lo)
                gemm("en_dil%dT_%d" % (i, p), dHe, dHe if two else None, hb, pitch, t_lo, T, -s0, -s1, DeP // 32,
                     DeP // 32 if two else 0, ReP // 16, self.Re, out, eb, pitch, 0, None, resid, (xe(i), eb, pitch), t_in, T, 0)
        self.mark("enc_stack_bwd")
        causal_wgrad("en_causal", dxe[0], eb, ReP)
        bias_grad("en_causal_layer", dxe[0], eb, pitch, 0, self.Re, k0)
        call("wn_reduce_slabs", ptr(bw["desc"]), bw["nops"], bw["vec"], ptr(bw["slab"]), ptr(self.gpack), st)
        call("wn_gather_grads", ptr(self.gpack), ptr(self.gidx), ptr(self.flat_grad), self.spec.total, st)
        self.mark("slab_reduce")

    # ------------------------------------------------------------------ fused training step (wavenet_autoencoder/train.py:146-160)
    def loss_and_grad(self, x, target, cond):
        if getattr(self, "_throttle", None) is None:
            self._throttle = _lib.StepThrottle()   # at most WN_MAX_STEPS_IN_FLIGHT fused steps in flight (music_amd/_lib.py)
        self._throttle.enter()
        _, enc, ws = self.forward(x, cond, want_probs=False)
        bw = self._bwd_workspace(ws)
        n = ws["B"] * ws["W"]
        target = target.reshape(-1)
        assert target.numel() == n and target.dtype == torch.int64 and target.is_cuda
        if "loss_part" not in ws:
            ws["loss_part"] = torch.zeros(_lib.CE_NUM_PARTIALS, dtype=torch.float32, device=self.device)
        if self.Q == 256:
            call("wn_chunk_softmax256_ce", ptr(ws["O"]), ptr(target), None, ptr(bw["dO"]), ptr(ws["loss_part"]), n, 1.0 / n, _lib.stream())
        else:
            call("wn_chunk_softmax_ce", ptr(ws["O"]), ptr(target), None, ptr(bw["dO"]), ptr(ws["loss_part"]), n, self.Q, 1.0 / n,
                 _lib.stream())
        self.backward(ws, None)
        loss = ws["loss_part"].sum()
        self._throttle.leave()
        return loss

    def adam_init(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.adam_state = dict(m=torch.zeros_like(self.flat), v=torch.zeros_like(self.flat), t=0, lr=lr, b1=betas[0], b2=betas[1], eps=eps)

    def adam_step(self, gscale=1.0):
        s = self.adam_state
        s["t"] += 1
        call("wn_adam_flat", ptr(self.flat), ptr(self.flat_grad), ptr(s["m"]), ptr(s["v"]), self.spec.total,
             s["lr"], s["b1"], s["b2"], s["eps"], 1.0 - s["b1"] ** s["t"], 1.0 - s["b2"] ** s["t"], gscale, _lib.stream())
