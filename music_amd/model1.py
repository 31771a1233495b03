"""Counterpart of the reference's ``wavenet_autoencoder/model1.py`` on the MI355X (forward path).

Same constructor kwargs, attributes, registered sub-modules and ``state_dict`` keys
(``en_dilation_layer_stack.i``, ``en_dense_layer_stack.i``, ``de_dilation_layer_stack.{3i+0..2}`` =
filter_gate / dense / skip, ``en_causal_layer``, ``bottleneck_layer``, ``de_causal_layer``,
``connection_1/2``), same ``forward`` contract (model1.py:256-268): ``(B, Q, T)`` float ->
probabilities ``(B*(T-rf+1), Q)`` with the chunk softmax.

Reference behaviours that are reproduced on purpose:
  * the conditioning projections are 31 FRESH ``nn.Conv1d(bottleneck, C, 1)`` drawn from the global
    torch CPU RNG inside every forward, with bias, never registered or trained
    (model1.py:178-179,216-217) — seed ``torch.manual_seed(s)`` right before ``forward``
    to reproduce a reference run;
  * ``_conditon`` (sic) stretches the encoding when ``len(x) % len(enc) == 0`` and tiles it
    otherwise (model1.py:227-247);
  * gate = FIRST half of the filter_gate channels, filter = second half (model1.py:188-190).
The arithmetic runs through libwavenet_hip.so only (no CPU path).  The backward pass of this model
is not implemented yet (the reference trains the encoder only through those random projections).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

try:
    from . import _lib
    from ._lib import call, ptr
    from .engine import SLACK, PAD_BACK, _Spec, _pad, pack_index
except ImportError:
    from music_amd import _lib
    from music_amd._lib import call, ptr
    from music_amd.engine import SLACK, PAD_BACK, _Spec, _pad, pack_index


class _AutoencoderEngine:
    def __init__(self, net, device, mode="f16x3"):
        self.net, self.device = net, device
        self.mode = _lib.MODE_NAMES[mode]
        self.dil = [int(d) for d in net.dilations]
        self.N = len(self.dil)
        self.Q = net.quantization_channel
        if net.filter_width != 2 or self.Q != 256:
            raise NotImplementedError("HIP path implements filter_width == 2 and quantization_channel == 256")
        self.Re, self.De, self.Bw, self.pool = net.en_residual_channel, net.en_dilation_channel, net.en_bottleneck_width, net.en_pool_kernel_size
        self.Rd, self.Dd, self.Sd = net.de_residual_channel, net.de_dilation_channel, net.de_skip_channel
        self.CHe = _pad(max(self.Re, self.De), 32)
        self.CHd = _pad(max(self.Rd, self.Dd), 32)
        if self.CHd not in (32, 64):
            raise NotImplementedError("HIP path supports decoder residual/dilation channels <= 64")
        self.SP, self.BwP = _pad(self.Sd, 32), _pad(self.Bw, 32)
        self.rf = sum(self.dil) + 2
        self.off = [1]
        for d in self.dil:
            self.off.append(self.off[-1] + d)
        self.use_bias = bool(net.use_bias)
        named = list(net.named_parameters())
        self.spec = _Spec([(n, tuple(p.shape)) for n, p in named])
        self.flat = torch.zeros(self.spec.total, dtype=torch.float32, device=device)
        with torch.no_grad():
            for n, p in named:
                o = self.spec.off[n]
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self._build_packs()
        self._ws = {}

    def _bias(self, name):
        return ptr(self.flat, self.spec.off[name + ".bias"]) if self.use_bias else None

    def _build_packs(self):
        sp, Q, N = self.spec, self.Q, self.N
        CHe, CHd, SP, BwP = self.CHe, self.CHd, self.SP, self.BwP
        Re, De, Rd, Dd, Sd, Bw = self.Re, self.De, self.Rd, self.Dd, self.Sd, self.Bw
        packs = []

        def full(m, k):
            return np.full((m, k), -1, dtype=np.int64)

        for name, ch, r in (("en_causal", CHe, Re), ("de_causal", CHd, Rd)):
            wc = sp.conv(name + "_layer.weight")
            w = full(ch, 2 * Q)
            w[:r, :Q], w[:r, Q:] = wc[:, :, 0], wc[:, :, 1]
            packs.append((name, pack_index(w)))
        for i in range(N):
            wd = sp.conv("en_dilation_layer_stack.%d.weight" % i)              # [De,Re,2]
            w = full(CHe, 2 * CHe)
            w[:De, :Re], w[:De, CHe:CHe + Re] = wd[:, :, 0], wd[:, :, 1]
            packs.append(("en_dil%d" % i, pack_index(w)))
            w = full(CHe, CHe)
            w[:Re, :De] = sp.conv("en_dense_layer_stack.%d.weight" % i)[:, :, 0]
            packs.append(("en_dense%d" % i, pack_index(w)))
            wfg = sp.conv("de_dilation_layer_stack.%d.weight" % (3 * i))       # [2Dd,Rd,2], gate rows first
            w = full(2 * CHd, 2 * CHd)
            for h, rows in enumerate((slice(Dd, 2 * Dd), slice(0, Dd))):        # my rows: filter then gate
                w[h * CHd:h * CHd + Dd, :Rd] = wfg[rows, :, 0]
                w[h * CHd:h * CHd + Dd, CHd:CHd + Rd] = wfg[rows, :, 1]
            packs.append(("de_fg%d" % i, pack_index(w)))
            w = full(CHd, CHd)
            w[:Rd, :Dd] = sp.conv("de_dilation_layer_stack.%d.weight" % (3 * i + 1))[:, :, 0]
            packs.append(("de_d%d" % i, pack_index(w, chained=True)))
        w = full(BwP, CHe)
        w[:Bw, :Re] = sp.conv("bottleneck_layer.weight")[:, :, 0]
        packs.append(("bottleneck", pack_index(w)))
        w = full(SP, N * CHd)
        for i in range(N):
            w[:Sd, i * CHd:i * CHd + Dd] = sp.conv("de_dilation_layer_stack.%d.weight" % (3 * i + 2))[:, :, 0]
        packs.append(("skip", pack_index(w)))
        w = full(SP, SP)
        w[:Sd, :Sd] = sp.conv("connection_1.weight")[:, :, 0]
        packs.append(("c1", pack_index(w)))
        w = full(Q, SP)
        w[:, :Sd] = sp.conv("connection_2.weight")[:, :, 0]
        packs.append(("c2", pack_index(w)))
        hp = 1024 if self.mode in (_lib.F16X3, _lib.BF16X3) else 512
        self.pk_off, o = {}, 0
        for name, idx in packs:
            self.pk_off[name] = o * hp // 512
            o += len(idx)
        self.pk_idx = torch.from_numpy(np.concatenate([i for _, i in packs]).astype(np.int32)).to(self.device)
        self.pk = torch.zeros(o * hp // 512, dtype=torch.int16, device=self.device)

    def workspace(self, B, T):
        ws = self._ws.get((B, T))
        if ws is not None:
            return ws
        self._ws.clear()
        dev = self.device
        pitch = _pad(T, 256) + 512
        W = T - self.rf + 1

        def buf(rows):
            return torch.zeros(SLACK + B * rows * pitch + PAD_BACK, dtype=torch.float32, device=dev)

        ws = dict(B=B, T=T, W=W, pitch=pitch, Xe=[buf(self.CHe), buf(self.CHe)], He=buf(self.CHe), E=buf(self.BwP),
                  Xd=[buf(self.CHd), buf(self.CHd)], Z=buf(self.N * self.CHd), U=buf(self.SP), R1=buf(self.SP),
                  C1=buf(self.SP), O=torch.zeros(B * self.Q * W + PAD_BACK, dtype=torch.float32, device=dev))
        self._ws[(B, T)] = ws
        return ws

    def forward(self, x, cond):
        """cond: list of N+1 (weight (C,Bw,1), bias (C,)) CPU tensors (see wavenet_autoencoder.forward)."""
        B, Q, T = x.shape
        W = T - self.rf + 1
        Le = W // self.pool
        if Le < 1:
            raise RuntimeError("Output size is too small: %d samples of encoding cannot be pooled by %d" % (W, self.pool))
        ws = self.workspace(B, T)
        st = _lib.stream()
        m, pitch, N, CHe, CHd, SP, BwP = self.mode, ws["pitch"], self.N, self.CHe, self.CHd, self.SP, self.BwP
        call("wn_pack_weights", ptr(self.flat), ptr(self.pk_idx), ptr(self.pk), self.pk_idx.numel(), m, st)
        fr = lambda name: ptr(self.pk, self.pk_off[name])
        lo = self.rf - 1
        NONE3 = (None, 0, 0)

        def gemm(in0, in1, in_bs, in_pitch, in_lo, in_hi, s0, s1, ks0, ks1, pack, mt, mvalid, out, out_bs, out_pitch, out_shift,
                 bias, resid, mask, t_lo, t_hi, relu_in):
            call("wn_chan_gemm", in0, in1, in_bs, in_pitch, in_lo, in_hi, s0, s1, ks0, ks1, fr(pack), mt, mvalid,
                 out, out_bs, out_pitch, out_shift, bias, resid[0], resid[1], resid[2], resid[3] if len(resid) > 3 else 0,
                 mask[0], mask[1], mask[2], t_lo, t_hi, relu_in, B, m, st)

        # ---------------- encoder (model1.py:137-156)
        xe = [ptr(b_, SLACK) for b_ in ws["Xe"]]
        He, E = ptr(ws["He"], SLACK), ptr(ws["E"], SLACK)
        eb = CHe * pitch
        gemm(ptr(x), ptr(x), Q * T, T, 0, T, -1, 0, Q // 32, Q // 32, "en_causal", CHe // 16, self.Re,
             xe[0], eb, pitch, 0, self._bias("en_causal_layer"), NONE3, NONE3, 1, T, 0)
        for i, d in enumerate(self.dil):
            src, dst = xe[i % 2], xe[(i + 1) % 2]
            t_lo = self.off[i + 1]
            # h = dilated_conv(relu(x));   x' = dense(relu(h)) + x[tail]
            gemm(src, src, eb, pitch, self.off[i], T, -d, 0, CHe // 32, CHe // 32, "en_dil%d" % i, CHe // 16, self.De,
                 He, eb, pitch, 0, self._bias("en_dilation_layer_stack.%d" % i), NONE3, NONE3, t_lo, T, 1)
            gemm(He, None, eb, pitch, t_lo, T, 0, 0, CHe // 32, 0, "en_dense%d" % i, CHe // 16, self.Re,
                 dst, eb, pitch, 0, self._bias("en_dense_layer_stack.%d" % i), (src, eb, pitch, t_lo), NONE3, t_lo, T, 1)
        gemm(xe[N % 2], None, eb, pitch, lo, T, 0, 0, CHe // 32, 0, "bottleneck", BwP // 16, self.Bw,
             E, BwP * pitch, pitch, 0, self._bias("bottleneck_layer"), NONE3, NONE3, lo, T, 0)
        enc = torch.empty(B, self.Bw, Le, dtype=torch.float32, device=self.device)
        call("wn_avgpool", E, BwP * pitch, pitch, lo, self.pool, Le, self.Bw, ptr(enc), self.Bw * Le, Le, B, st)

        # ---------------- conditioning tables: en = Conv1d_rand(enc)  (model1.py:178-179, 216-217)
        Dd, Sd = self.Dd, self.Sd
        cw = torch.stack([c[0][:, :, 0] for c in cond[:N]]).to(self.device)            # (N, 2Dd, Bw), gate rows first
        cb = torch.stack([c[1] for c in cond[:N]]).to(self.device)                     # (N, 2Dd)
        en = torch.einsum("nck,bkl->nbcl", cw, enc) + cb[:, None, :, None]             # (N, B, 2Dd, Le)
        tab = torch.zeros(N, B, 2 * CHd, Le, dtype=torch.float32, device=self.device)
        tab[:, :, :Dd] = en[:, :, Dd:]                                                 # my rows: filter first
        tab[:, :, CHd:CHd + Dd] = en[:, :, :Dd]
        enf = F.conv1d(enc, cond[N][0].to(self.device), cond[N][1].to(self.device))    # (B, Sd, Le)

        # ---------------- decoder (model1.py:158-225)
        xd = [ptr(b_, SLACK) for b_ in ws["Xd"]]
        db, zb = CHd * pitch, N * CHd * pitch
        gemm(ptr(x), ptr(x), Q * T, T, 0, T, -1, 0, Q // 32, Q // 32, "de_causal", CHd // 16, self.Rd,
             xd[0], db, pitch, 0, self._bias("de_causal_layer"), NONE3, NONE3, 1, T, 0)
        for i, d in enumerate(self.dil):
            t_lo = self.off[i + 1]
            L = T - t_lo
            mode_c, q = (1, L // Le) if L % Le == 0 else (2, 0)
            bn = "de_dilation_layer_stack.%d"
            bias_fg = self._bias(bn % (3 * i))
            # filter_gate bias: gate rows first in the reference tensor
            bf = bias_fg + 4 * Dd if bias_fg is not None else None
            call("wn_resblock_fwd", xd[i % 2], xd[(i + 1) % 2], ptr(ws["Z"], SLACK + i * CHd * pitch), db, zb, pitch,
                 fr("de_fg%d" % i), fr("de_d%d" % i), bf, bias_fg, self._bias(bn % (3 * i + 1)), Dd, self.Rd, CHd, d,
                 t_lo, T, lo, 1 if i < N - 1 else 0, ptr(tab[i]), 2 * CHd * Le, Le, mode_c, Le, q, B, m, st)
        U, R1, C1 = ptr(ws["U"], SLACK), ptr(ws["R1"], SLACK), ptr(ws["C1"], SLACK)
        sb = SP * pitch
        bias_s = None
        if self.use_bias:
            o = self.spec.off
            bsum = sum(self.flat[o[bn % (3 * i + 2) + ".bias"]:o[bn % (3 * i + 2) + ".bias"] + Sd] for i in range(N)).contiguous()
            ws["bias_skip"] = bsum
            bias_s = ptr(bsum)
        gemm(ptr(ws["Z"], SLACK), None, zb, pitch, lo, T, 0, 0, N * CHd // 32, 0, "skip", SP // 16, Sd,
             U, sb, pitch, 0, bias_s, NONE3, NONE3, lo, T, 0)
        # final conditioning expanded over time (stretch / tile rule on the length-W sequence)
        tr = torch.arange(W, device=self.device)
        idx = tr // (W // Le) if W % Le == 0 else tr % Le
        ws["C1"][SLACK:SLACK + B * SP * pitch].view(B, SP, pitch)[:, :Sd, lo:T] = enf[:, :, idx]
        gemm(U, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, "c1", SP // 16, Sd,
             R1, sb, pitch, 0, self._bias("connection_1"), (C1, sb, pitch, lo), NONE3, lo, T, 1)
        gemm(R1, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, "c2", Q // 16, Q,
             ptr(ws["O"]), Q * W, W, -lo, self._bias("connection_2"), NONE3, NONE3, lo, T, 1)
        probs = torch.empty(B * W, Q, dtype=torch.float32, device=self.device)
        call("wn_chunk_softmax256_fwd", ptr(ws["O"]), ptr(probs), B * W, st)
        return probs, enc


class wavenet_autoencoder(nn.Module):

    def __init__(self, filter_width, quantization_channel, dilations, en_residual_channel, en_dilation_channel,
                 en_bottleneck_width, en_pool_kernel_size, de_residual_channel, de_dilation_channel,
                 de_skip_channel, use_bias):
        super(wavenet_autoencoder, self).__init__()
        self.filter_width = filter_width
        self.quantization_channel = quantization_channel
        self.dilations = dilations
        self.en_residual_channel = en_residual_channel
        self.en_dilation_channel = en_dilation_channel
        self.en_bottleneck_width = en_bottleneck_width
        self.en_pool_kernel_size = en_pool_kernel_size
        self.de_residual_channel = de_residual_channel
        self.de_dilation_channel = de_dilation_channel
        self.de_skip_channel = de_skip_channel
        self.use_bias = use_bias
        self.receptive_field = self._calc_receptive_field()
        self.softmax = nn.Softmax(dim=1)
        # construction (= RNG draw) order of model1.py:55-58: encoder pairs, decoder triples, the
        # three input/bottleneck convs, the two output convs
        self.en_dilation_layer_stack = nn.ModuleList()
        self.en_dense_layer_stack = nn.ModuleList()
        for d in dilations:
            self.en_dilation_layer_stack.append(nn.Conv1d(en_residual_channel, en_dilation_channel, filter_width,
                                                          dilation=d, bias=use_bias))
            self.en_dense_layer_stack.append(nn.Conv1d(en_dilation_channel, en_residual_channel, 1, bias=use_bias))
        self.de_dilation_layer_stack = nn.ModuleList()
        for d in dilations:
            self.de_dilation_layer_stack.extend([
                nn.Conv1d(de_residual_channel, 2 * de_dilation_channel, filter_width, dilation=d, bias=use_bias),
                nn.Conv1d(de_dilation_channel, de_residual_channel, kernel_size=1, dilation=d, bias=use_bias),
                nn.Conv1d(de_dilation_channel, de_skip_channel, dilation=d, kernel_size=1, bias=use_bias)])
        self.en_causal_layer = nn.Conv1d(quantization_channel, en_residual_channel, filter_width, bias=use_bias)
        self.bottleneck_layer = nn.Conv1d(en_residual_channel, en_bottleneck_width, 1, bias=use_bias)
        self.de_causal_layer = nn.Conv1d(quantization_channel, de_residual_channel, filter_width, bias=use_bias)
        self.connection_1 = nn.Conv1d(de_skip_channel, de_skip_channel, 1, bias=use_bias)
        self.connection_2 = nn.Conv1d(de_skip_channel, quantization_channel, 1, bias=use_bias)
        self._engine = None
        self.last_encoding = None

    def _calc_receptive_field(self):
        return (self.filter_width - 1) * (sum(self.dilations) + 1) + 1

    def _draw_conditioning(self):
        """The 31 per-forward conditioning convs, drawn on the CPU from the global RNG in the
        reference's order (model1.py:178 per layer, :216 final)."""
        n = len(self.dilations)
        cond = []
        for i in range(n + 1):
            c = nn.Conv1d(self.en_bottleneck_width,
                          2 * self.de_dilation_channel if i < n else self.de_skip_channel, 1)
            cond.append((c.weight.detach(), c.bias.detach()))
        return cond

    def _engine_for(self, device):
        if device.type != "cuda":
            raise RuntimeError("music_amd.wavenet_autoencoder runs on an MI355X (ROCm) device only; there is no CPU path")
        eng = self._engine
        p0 = next(self.parameters())
        if eng is None or eng.device != device or p0.data_ptr() != eng.flat.data_ptr():
            if any(p.device != device for p in self.parameters()):
                raise RuntimeError("music_amd.wavenet_autoencoder: parameters and input are on different devices")
            eng = self._engine = _AutoencoderEngine(self, device)
        return eng

    def forward(self, wave_sample):
        batch_size, original_channels, seq_len = wave_sample.size()
        output_width = seq_len - self.receptive_field + 1
        if output_width <= 0:
            raise ValueError("wave sample not long enough")
        eng = self._engine_for(wave_sample.device)
        cond = self._draw_conditioning()
        with torch.no_grad():
            probs, enc = eng.forward(wave_sample.detach().float().contiguous(), cond)
        self.last_encoding = enc
        return probs
