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
The arithmetic runs through libwavenet_hip.so only (no CPU path).  ``forward`` is differentiable
(``loss.backward()`` fills every parameter's gradient, the encoder's through the random projections
exactly as in the reference).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

try:
    from . import _lib, _losshook
    from ._lib import call, ptr
    from .engine import SLACK, PAD_BACK, _Spec, _pad, pack_index, pack_positions, WorkspacePool, WorkspaceHold
except ImportError:
    from music_amd import _lib, _losshook
    from music_amd._lib import call, ptr
    from music_amd.engine import SLACK, PAD_BACK, _Spec, _pad, pack_index, pack_positions, WorkspacePool, WorkspaceHold


class _AutoencoderEngine:
    """Host-side plan of the autoencoder on one MI355X: flat parameters, packed-weight index maps,
    workspaces, forward (model1.py:256-268) and backward (autograd of it)."""

    def __init__(self, net, device, mode="f16x3", mode_bwd="bf16x3"):
        self.net, self.device = net, device
        self.mode = _lib.MODE_NAMES[mode]
        self.mode_b = _lib.MODE_NAMES[mode_bwd]
        self.dil = [int(d) for d in net.dilations]
        self.N = len(self.dil)
        self.Q = net.quantization_channel
        self.fused_loss_ok = True           # _losshook.py: nn.CrossEntropyLoss on the module's output may run as wn_chunk_softmax256_ce
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
        # one launch per encoder block (wn_enc_resblock_fwd) instead of two channel GEMMs; WN_AE_FUSED_ENC=0 = the GEMMs
        self.fused_encoder = os.environ.get("WN_AE_FUSED_ENC", "1") == "1"
        # 32 / 32 channels on both sides (the reference's shipped model_params.json): even batches run BOTH stacks on the
        # 64-channel one-launch blocks, two clips per 64-row tensor with block-diagonal packs (music_amd/engine.py "pair")
        self.pair_ok = (self.CHe == 32 and self.CHd == 32 and not self.use_bias and self.mode == _lib.F16X3 and
                        self.mode_b == _lib.BF16X3 and os.environ.get("WN_PAIR32", "1") == "1")
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
                p.data = view
        self._build_packs()
        self._ws = WorkspacePool(self._make_workspace)
        self._gen = 0
        self.marks = None            # list of (name, torch.cuda.Event) when phase timing is on (bench.py)
        self._side = None            # second HIP stream for the epilogue's weight gradients (as music_amd/engine.py)
        self.overlap_wgrad = True

    def mark(self, name):
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev))

    def _stage_cond(self, cond):
        """cond (N+1 CPU (weight, bias) pairs) -> device tensors cw (N, 2Dd, Bw), cb (N, 2Dd), cfw (Sd, Bw, 1), cfb (Sd) through one
        of three rotating pinned staging buffers (an event per buffer says when its last copy has left)."""
        N, Dd, Sd, Bw = self.N, self.Dd, self.Sd, self.Bw
        n_cw, n_cb, n_fw = N * 2 * Dd * Bw, N * 2 * Dd, Sd * Bw
        total = n_cw + n_cb + n_fw + Sd
        if getattr(self, "_cpin", None) is None or self._cpin[0][0].numel() != total:
            self._cpin = [(torch.empty(total, dtype=torch.float32).pin_memory(), None) for _ in range(3)]
            self._cpin_i = 0
        k = self._cpin_i
        self._cpin_i = (k + 1) % 3
        pin, ev = self._cpin[k]
        if ev is not None:
            ev.synchronize()
        # plain memcpys (numpy): a torch CPU op on more than 32768 elements opens an OpenMP region that wakes EVERY intra-op
        # thread (128 on an MI355X host), and in a container with a CPU quota those spinning threads get the whole process
        # throttled - 80 ms out of every 100 with the reference's shipped parameters (Bw = 512)
        pn = pin.numpy()
        pn[:n_cw].reshape(N, 2 * Dd * Bw)[...] = np.stack([c[0].numpy().reshape(-1) for c in cond[:N]])
        pn[n_cw:n_cw + n_cb].reshape(N, 2 * Dd)[...] = np.stack([c[1].numpy() for c in cond[:N]])
        pn[n_cw + n_cb:n_cw + n_cb + n_fw] = cond[N][0].numpy().reshape(-1)
        pn[n_cw + n_cb + n_fw:] = cond[N][1].numpy()
        dev = torch.empty(total, dtype=torch.float32, device=self.device)
        dev.copy_(pin, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._cpin[k] = (pin, ev)
        return (dev[:n_cw].view(N, 2 * Dd, Bw), dev[n_cw:n_cw + n_cb].view(N, 2 * Dd),
                dev[n_cw + n_cb:n_cw + n_cb + n_fw].view(Sd, Bw, 1), dev[n_cw + n_cb + n_fw:])

    def _bias(self, name):
        return ptr(self.flat, self.spec.off[name + ".bias"]) if self.use_bias else None

    def _build_packs(self):
        sp, Q, N = self.spec, self.Q, self.N
        CHe, CHd, SP, BwP = self.CHe, self.CHd, self.SP, self.BwP
        Re, De, Rd, Dd, Sd, Bw = self.Re, self.De, self.Rd, self.Dd, self.Sd, self.Bw
        fwd, bwd, gp = [], [], []
        gidx = np.full(self.spec.total, -1, dtype=np.int64)
        gsize = [0]

        def full(m, k):
            return np.full((m, k), -1, dtype=np.int64)

        def add(name, w, chained=False, grad=True):
            """forward pack of the effective matrix w (entries = flat parameter offsets) + its
            gradient matrix (same shape) + the gather map back to the parameters"""
            fwd.append((name, pack_index(w, chained)))
            if grad:
                o = gsize[0]
                gp.append((name, o, w.shape[0], w.shape[1]))
                r, c = np.nonzero(w >= 0)
                gidx[w[r, c]] = o + r * w.shape[1] + c
                gsize[0] += w.size

        pa, pb = {}, {}                                      # pair mode: parameter offset -> its two places in a block-diagonal gradient

        def diag(m32, rb, cb):
            """[rb*32][cb*32] blocks of 32 x 32 -> [rb*64][cb*64], every block doubled on the diagonal (clip A, clip B)"""
            out = full(rb * 64, cb * 64)
            for a_ in range(rb):
                for b_ in range(cb):
                    blk = m32[a_ * 32:(a_ + 1) * 32, b_ * 32:(b_ + 1) * 32]
                    for c_ in range(2):
                        out[a_ * 64 + c_ * 32:a_ * 64 + (c_ + 1) * 32, b_ * 64 + c_ * 32:b_ * 64 + (c_ + 1) * 32] = blk
            return out

        def add2(name, w32, rb, cb, chained=False):
            """pair-mode forward pack + gradient matrix of the block-diagonal form of w32, and where each parameter's two
            gradient copies sit in it"""
            fwd.append((name, pack_index(diag(w32, rb, cb), chained)))
            o, cols = gsize[0], cb * 64
            gp.append((name, o, rb * 64, cols))
            r, c = np.nonzero(w32 >= 0)
            base = o + ((r // 32) * 64 + r % 32) * cols + (c // 32) * 64 + c % 32
            for par, pos in zip(w32[r, c], base):
                pa[int(par)], pb[int(par)] = int(pos), int(pos) + 32 * cols + 32
            gsize[0] += rb * 64 * cols

        self.wt_idx, self.wt = {}, {}
        for name, ch, r in (("en_causal", CHe, Re), ("de_causal", CHd, Rd)):
            wc = sp.conv(name + "_layer.weight")
            w = full(ch, 2 * Q)
            w[:r, :Q], w[:r, Q:] = wc[:, :, 0], wc[:, :, 1]
            add(name, w)
            wT = full(Q, 2 * ch)                          # its transpose for the gradient w.r.t. the input (input_grad): K = [tap1^T | tap0^T]
            wT[:, :r], wT[:, ch:ch + r] = wc[:, :, 1].T, wc[:, :, 0].T
            bwd.append((name + "T", pack_index(wT)))
            # the same weight as [tap][q][ch] for the forward from codes (wn_causal_fwd_codes), a gather map
            wt = np.full((2, Q, ch), -1, dtype=np.int64)
            wt[:, :, :r] = wc.transpose(2, 1, 0)
            self.wt_idx[name] = torch.from_numpy(wt.reshape(-1).astype(np.int32)).to(self.device)
            self.wt[name] = torch.zeros(2 * Q * ch, dtype=torch.float32, device=self.device)
        for i in range(N):
            wd = sp.conv("en_dilation_layer_stack.%d.weight" % i)              # [De,Re,2]
            w = full(CHe, 2 * CHe)
            w[:De, :Re], w[:De, CHe:CHe + Re] = wd[:, :, 0], wd[:, :, 1]
            add("en_dil%d" % i, w)
            wt = full(CHe, 2 * CHe)                                            # dx: rows Re, K = [tap1^T | tap0^T] over De
            wt[:Re, :De], wt[:Re, CHe:CHe + De] = wd[:, :, 1].T, wd[:, :, 0].T
            bwd.append(("en_dilT%d" % i, pack_index(wt)))
            wq = full(2 * CHe, CHe)                     # [W1^T; W0^T] over dh: the one-launch backward block of the encoder
            wq[:CHe], wq[CHe:] = wt[:, :CHe], wt[:, CHe:]
            bwd.append(("en_pq%d" % i, pack_index(wq)))
            if self.pair_ok:
                wdil32 = full(CHe, 2 * CHe)
                wdil32[:De, :Re], wdil32[:De, CHe:CHe + Re] = wd[:, :, 0], wd[:, :, 1]
                add2("en_dil2_%d" % i, wdil32, 1, 2)
                bwd.append(("en_pq2_%d" % i, pack_index(diag(wq, 2, 1))))
            w = full(CHe, CHe)
            w[:Re, :De] = sp.conv("en_dense_layer_stack.%d.weight" % i)[:, :, 0]
            add("en_dense%d" % i, w)
            fwd.append(("en_dense_c%d" % i, pack_index(w, True)))               # chained k order: the fused encoder block
            bwd.append(("en_denseT%d" % i, pack_index(np.ascontiguousarray(w.T))))
            if self.pair_ok:
                add2("en_dense2_%d" % i, w, 1, 1, chained=True)                 # (its pack doubles as "en_dense_c2")
                bwd.append(("en_denseT2_%d" % i, pack_index(diag(np.ascontiguousarray(w.T), 1, 1))))
            wfg = sp.conv("de_dilation_layer_stack.%d.weight" % (3 * i))       # [2Dd,Rd,2], gate rows first
            w = full(2 * CHd, 2 * CHd)
            wx = full(CHd, 4 * CHd)
            for h, rows in enumerate((slice(Dd, 2 * Dd), slice(0, Dd))):        # my rows: filter then gate
                w[h * CHd:h * CHd + Dd, :Rd] = wfg[rows, :, 0]
                w[h * CHd:h * CHd + Dd, CHd:CHd + Rd] = wfg[rows, :, 1]
                wx[:Rd, h * CHd:h * CHd + Dd] = wfg[rows, :, 1].T
                wx[:Rd, 2 * CHd + h * CHd:2 * CHd + h * CHd + Dd] = wfg[rows, :, 0].T
            add("de_fg%d" % i, w)
            bwd.append(("de_fgT%d" % i, pack_index(wx)))
            wq = full(2 * CHd, 2 * CHd)                 # [W1^T; W0^T] over (df | dg): the one-launch backward block
            wq[:CHd], wq[CHd:] = wx[:, :2 * CHd], wx[:, 2 * CHd:]
            bwd.append(("de_pq%d" % i, pack_index(wq)))
            if self.pair_ok:
                wfg32 = full(2 * CHd, 2 * CHd)
                for h, rows in enumerate((slice(Dd, 2 * Dd), slice(0, Dd))):
                    wfg32[h * CHd:h * CHd + Dd, :Rd] = wfg[rows, :, 0]
                    wfg32[h * CHd:h * CHd + Dd, CHd:CHd + Rd] = wfg[rows, :, 1]
                add2("de_fg2_%d" % i, wfg32, 2, 2)
                bwd.append(("de_pq2_%d" % i, pack_index(diag(wq, 2, 2))))
            w = full(CHd, CHd)
            w[:Rd, :Dd] = sp.conv("de_dilation_layer_stack.%d.weight" % (3 * i + 1))[:, :, 0]
            add("de_d%d" % i, w, chained=True)
            bwd.append(("de_dT%d" % i, pack_index(np.ascontiguousarray(w.T))))
            if self.pair_ok:
                add2("de_d2_%d" % i, w, 1, 1, chained=True)
                bwd.append(("de_dT2_%d" % i, pack_index(diag(np.ascontiguousarray(w.T), 1, 1))))
        w = full(BwP, CHe)
        w[:Bw, :Re] = sp.conv("bottleneck_layer.weight")[:, :, 0]
        add("bottleneck", w)
        bwd.append(("bottleneckT", pack_index(np.ascontiguousarray(w.T))))
        w = full(SP, N * CHd)
        for i in range(N):
            w[:Sd, i * CHd:i * CHd + Dd] = sp.conv("de_dilation_layer_stack.%d.weight" % (3 * i + 2))[:, :, 0]
        add("skip", w)
        bwd.append(("skipT", pack_index(np.ascontiguousarray(w.T))))
        bwd.append(("skipTc", pack_index(np.ascontiguousarray(w.T), chained=True)))     # chained k order: wn_skip_epilogue_bwd
        w = full(SP, SP)
        w[:Sd, :Sd] = sp.conv("connection_1.weight")[:, :, 0]
        add("c1", w)
        bwd.append(("c1T", pack_index(np.ascontiguousarray(w.T))))
        bwd.append(("c1Tc", pack_index(np.ascontiguousarray(w.T), chained=True)))
        w = full(Q, SP)
        w[:, :Sd] = sp.conv("connection_2.weight")[:, :, 0]
        add("c2", w)
        bwd.append(("c2T", pack_index(np.ascontiguousarray(w.T))))

        def finish(lst, mode):
            hp = 1024 if mode in (_lib.F16X3, _lib.BF16X3) else 512
            offs, o = {}, 0
            for name, idx in lst:
                offs[name] = o * hp // 512
                o += len(idx)
            idx_all = torch.from_numpy(np.concatenate([i for _, i in lst]).astype(np.int32)).to(self.device)
            return offs, idx_all, torch.zeros(o * hp // 512, dtype=torch.int16, device=self.device)

        self.pk_off, self.pk_idx, self.pk = finish(fwd, self.mode)
        self.pkb_off, self.pkb_idx, self.pkb = finish(bwd, self.mode_b)
        self.gp_off = {name: (o, r, c) for name, o, r, c in gp}
        self.gpack = torch.zeros(gsize[0], dtype=torch.float32, device=self.device)
        self.gidx = torch.from_numpy(gidx.astype(np.int32)).to(self.device)   # -1 (biases) -> zero gradient
        if self.pair_ok:                                  # pair mode: a stack weight's gradient = the sum of its two copies
            ga, gb = gidx.copy(), np.full(self.spec.total, -1, dtype=np.int64)
            for par, pos in pa.items():
                ga[par], gb[par] = pos, pb[par]
            self.gidx_pa = torch.from_numpy(ga.astype(np.int32)).to(self.device)
            self.gidx_pb = torch.from_numpy(gb.astype(np.int32)).to(self.device)

    def workspace(self, B, T):
        return self._ws.peek(B, T)

    def _make_workspace(self, B, T):
        dev = self.device
        pitch = _pad(T, 256) + 512
        W = T - self.rf + 1
        N = self.N

        def buf(rows):
            return torch.zeros(SLACK + B * rows * pitch + PAD_BACK, dtype=torch.float32, device=dev)

        ws = dict(B=B, T=T, W=W, pitch=pitch, Xe=buf((N + 1) * self.CHe), He=buf(N * self.CHe), E=buf(self.BwP),
                  Xd=buf((N + 1) * self.CHd), Z=buf(N * self.CHd), U=buf(self.SP), R1=buf(self.SP),
                  C1=buf(self.SP), O=torch.zeros(B * self.Q * W + PAD_BACK, dtype=torch.float32, device=dev), bwd=None)
        return ws

    # layer i of a stacked [(N+1) or N][B][CH][pitch] buffer
    def _lay(self, t, i, ch, ws):
        return ptr(t, SLACK + i * ws["B"] * ch * ws["pitch"])

    def _gemm(self, st, B, mode, pack_ptr, in0, in1, in_bs, in_pitch, in_lo, in_hi, s0, s1, ks0, ks1, mt, mvalid, out, out_bs,
              out_pitch, out_shift, bias, resid, mask, t_lo, t_hi, relu_in):
        call("wn_chan_gemm", in0, in1, in_bs, in_pitch, in_lo, in_hi, s0, s1, ks0, ks1, pack_ptr, mt, mvalid,
             out, out_bs, out_pitch, out_shift, bias, resid[0], resid[1], resid[2], resid[3] if len(resid) > 3 else 0,
             mask[0], mask[1], mask[2], t_lo, t_hi, relu_in, B, mode, st)

    def forward(self, x, cond, want_probs=True):
        """cond: list of N+1 (weight (C,Bw,1), bias (C,)) CPU tensors (see wavenet_autoencoder.forward).
        want_probs=False stops at the pre-softmax logits in ws["O"] (the fused training step)."""
        B, Q, T = x.shape
        W = T - self.rf + 1
        Le = W // self.pool
        if Le < 1:
            raise RuntimeError("Output size is too small: %d samples of encoding cannot be pooled by %d" % (W, self.pool))
        ws = self._ws.get(B, T)
        self._gen += 1
        ws["gen"], ws["x_in"], ws["Le"] = self._gen, x, Le
        pair = ws["pair"] = self.pair_ok and B % 2 == 0 and Le <= 32      # decided per workspace shape (B, T fix Le)
        # the forward blocks pair only when that fills the chip (music_amd/engine.py); the backward blocks always
        pair_f = pair and (B // 2) * ((T + 511) // 512) >= 200
        # a one-hot built from integer codes (engine.onehot / the loader) carries them: both causal layers then run on the
        # codes (gather forward, scatter backward), as in music_amd/engine.py
        ws["x_codes"] = None
        tag = getattr(x, "_wn_codes", None)
        if tag is not None:
            codes, scrambled, version, cversion = tag
            if (x._version == version and codes._version == cversion and codes.is_cuda and codes.dtype == torch.int32 and
                    codes.is_contiguous() and tuple(codes.shape) == (B, T)):
                ws["x_codes"] = (codes, scrambled)
        st = _lib.stream()
        m, pitch, N, CHe, CHd, SP, BwP = self.mode, ws["pitch"], self.N, self.CHe, self.CHd, self.SP, self.BwP
        # the 31 conditioning projections (drawn on the CPU, model1.py:178,216) go to the device FIRST, in one asynchronous copy
        # from pinned memory: as four pageable .to(device) copies behind the encoder they made the host wait for the encoder
        # stack and the decoder start from an empty queue (0.35 ms for this phase at config 4)
        cw, cb, cfw, cfb = self._stage_cond(cond)
        call("wn_pack_weights", ptr(self.flat), ptr(self.pk_idx), ptr(self.pk), self.pk_idx.numel(), m, st)
        fr = lambda name: ptr(self.pk, self.pk_off[name])
        lo = self.rf - 1
        NONE3 = (None, 0, 0)
        gemm = lambda pack, *a: self._gemm(st, B, m, fr(pack), *a)

        # ---------------- encoder (model1.py:137-156); every x_i and h_i is kept for the backward
        xe = lambda i: self._lay(ws["Xe"], i, CHe, ws)
        he = lambda i: self._lay(ws["He"], i, CHe, ws)
        E = ptr(ws["E"], SLACK)
        eb = CHe * pitch
        def causal(name, ch, rows, out, obs, bias):
            if ws["x_codes"] is None:
                gemm(name, ptr(x), ptr(x), Q * T, T, 0, T, -1, 0, Q // 32, Q // 32, ch // 16, rows, out, obs, pitch, 0, bias,
                     NONE3, NONE3, 1, T, 0)
                return
            codes, scrambled = ws["x_codes"]
            call("wn_gather_grads", ptr(self.flat), ptr(self.wt_idx[name]), ptr(self.wt[name]), self.wt[name].numel(), st)
            call("wn_causal_fwd_codes", ptr(codes), 1 if scrambled else 0, ptr(self.wt[name]), bias, rows, out, obs, pitch, ch, Q, T,
                 B, st)
        self.mark("begin")
        causal("en_causal", CHe, self.Re, xe(0), eb, self._bias("en_causal_layer"))
        self.mark("en_causal_fwd")
        for i, d in enumerate(self.dil):
            t_lo = self.off[i + 1]
            # h = dilated_conv(relu(x));   x' = dense(relu(h)) + x[tail]
            if pair_f:                       # two clips per 64-row tensor, block-diagonal packs
                call("wn_enc_resblock_fwd", xe(i), xe(i + 1), he(i), 2 * eb, 2 * eb, pitch, fr("en_dil2_%d" % i), fr("en_dense2_%d" % i),
                     None, None, 64, 64, 64, d, t_lo, T, B // 2, m, st)
                continue
            if self.fused_encoder:
                call("wn_enc_resblock_fwd", xe(i), xe(i + 1), he(i), eb, eb, pitch, fr("en_dil%d" % i), fr("en_dense_c%d" % i),
                     self._bias("en_dilation_layer_stack.%d" % i), self._bias("en_dense_layer_stack.%d" % i), self.De, self.Re,
                     CHe, d, t_lo, T, B, m, st)
                continue
            gemm("en_dil%d" % i, xe(i), xe(i), eb, pitch, self.off[i], T, -d, 0, CHe // 32, CHe // 32, CHe // 16, self.De,
                 he(i), eb, pitch, 0, self._bias("en_dilation_layer_stack.%d" % i), NONE3, NONE3, t_lo, T, 1)
            gemm("en_dense%d" % i, he(i), None, eb, pitch, t_lo, T, 0, 0, CHe // 32, 0, CHe // 16, self.Re,
                 xe(i + 1), eb, pitch, 0, self._bias("en_dense_layer_stack.%d" % i), (xe(i), eb, pitch, t_lo), NONE3, t_lo, T, 1)
        self.mark("enc_stack_fwd")
        gemm("bottleneck", xe(N), None, eb, pitch, lo, T, 0, 0, CHe // 32, 0, BwP // 16, self.Bw,
             E, BwP * pitch, pitch, 0, self._bias("bottleneck_layer"), NONE3, NONE3, lo, T, 0)
        enc = torch.empty(B, self.Bw, Le, dtype=torch.float32, device=self.device)
        call("wn_avgpool", E, BwP * pitch, pitch, lo, self.pool, Le, self.Bw, ptr(enc), self.Bw * Le, Le, B, st)

        # ---------------- conditioning tables: en = Conv1d_rand(enc)  (model1.py:178-179, 216-217)
        Dd, Sd = self.Dd, self.Sd
        en = torch.einsum("nck,bkl->nbcl", cw, enc) + cb[:, None, :, None]             # (N, B, 2Dd, Le)
        tab = torch.zeros(N, B, 2 * CHd, Le, dtype=torch.float32, device=self.device)
        tab[:, :, :Dd] = en[:, :, Dd:]                                                 # my rows: filter first
        tab[:, :, CHd:CHd + Dd] = en[:, :, :Dd]
        enf = F.conv1d(enc, cfw, cfb)                                                  # (B, Sd, Le)
        ws.update(enc=enc, tab=tab, cw=cw, cfw=cfw)

        # ---------------- decoder (model1.py:158-225)
        xd = lambda i: self._lay(ws["Xd"], i, CHd, ws)
        db, zb = CHd * pitch, N * CHd * pitch
        causal("de_causal", CHd, self.Rd, xd(0), db, self._bias("de_causal_layer"))
        self.mark("bottleneck_cond_de_causal")
        bn = "de_dilation_layer_stack.%d"
        cmodes = []
        for i in range(N):
            L = T - self.off[i + 1]
            cmodes.append((1, L // Le) if L % Le == 0 else (2, 0))
        ws["cmodes"] = cmodes
        cpk = cix = None
        CHp, Bp = (64, B // 2) if pair else (CHd, B)         # pair mode: 64-row tensors of two clips, rows [f: A B | g: A B]
        tab_c = tab                                         # per clip, rows [f | g]: what the 32-channel forward block gathers from
        if pair:
            tab = tab.view(N, Bp, 2, 2, CHd, Le).permute(0, 1, 3, 2, 4, 5).reshape(N, Bp, 4 * CHd, Le).contiguous()
            ws["tab"] = tab
        if Le <= 32 and CHp == 64 and m == _lib.F16X3 and (pair or os.environ.get("WN_AE_COND_MFMA", "1") == "1"):
            # the conditioning bias on the matrix cores: bucket of every sample of every block as bytes (built once per
            # workspace: row i = PAD zeros, bucket(t - t_lo) for t in [t_lo, T), zeros) and, per forward, the tables as
            # packed A fragments ([2CH rows][32 buckets] per block and clip)
            if "cidx" not in ws:
                PADI = _lib.COND_IDX_PAD
                cidx = torch.zeros(N, PADI + T + 64, dtype=torch.uint8, device=self.device)
                for i in range(N):
                    L = T - self.off[i + 1]
                    trr = torch.arange(L, device=self.device)
                    mode_c, q = cmodes[i]
                    cidx[i, PADI:PADI + L] = (torch.clamp(trr // q, max=Le - 1) if mode_c == 1 else trr % Le).to(torch.uint8)
                ws["cidx"] = cidx
                row, k = pack_positions(2 * CHp // 16, 1, False)
                one = np.where(k < Le, row * Le + k, -1).astype(np.int64)                  # one [2CH][Le] table
                base = np.arange(N * Bp, dtype=np.int64)[:, None] * (2 * CHp * Le)
                ws["ctab_idx"] = torch.from_numpy(np.where(one[None, :] >= 0, base + one[None, :], -1).astype(np.int32)
                                                  .reshape(-1)).to(self.device)
                ws["ctab_pk"] = torch.empty(N * Bp * 2 * CHp * 32 * 2, dtype=torch.int16, device=self.device)
            if pair_f or not pair:                        # (the backward blocks read the fp32 table; only a 64-channel forward the pack)
                call("wn_pack_weights", ptr(tab), ptr(ws["ctab_idx"]), ptr(ws["ctab_pk"]), ws["ctab_idx"].numel(), m, st)
                cpk, cix = ws["ctab_pk"], ws["cidx"]
        cpb = 2 * CHp * 32 * 2                          # halfs of one clip's packed table (hi + lo planes)
        for i, d in enumerate(self.dil):
            t_lo = self.off[i + 1]
            mode_c, q = cmodes[i]
            if pair_f:
                call("wn_resblock_fwd", xd(i), xd(i + 1), ptr(ws["Z"], SLACK + i * CHd * pitch), 2 * db, 2 * zb, pitch,
                     fr("de_fg2_%d" % i), fr("de_d2_%d" % i), None, None, None, 64, 64, 64, d,
                     t_lo, T, t_lo, 1 if i < N - 1 else 0, ptr(tab[i]), 4 * CHd * Le, Le, mode_c, Le, q,
                     ptr(cpk, i * Bp * cpb), cpb, ptr(cix[i]), zb, Bp, m, st)
                continue
            bias_fg = self._bias(bn % (3 * i))
            bf = bias_fg + 4 * Dd if bias_fg is not None else None      # filter_gate bias: gate rows first
            call("wn_resblock_fwd", xd(i), xd(i + 1), ptr(ws["Z"], SLACK + i * CHd * pitch), db, zb, pitch,
                 fr("de_fg%d" % i), fr("de_d%d" % i), bf, bias_fg, self._bias(bn % (3 * i + 1)), Dd, self.Rd, CHd, d,
                 t_lo, T, t_lo, 1 if i < N - 1 else 0, ptr(tab_c[i]), 2 * CHd * Le, Le, mode_c, Le, q,
                 ptr(cpk, i * B * cpb) if cpk is not None else None, cpb, ptr(cix[i]) if cix is not None else None,
                 0, B, m, st)   # z on the whole valid range: the backward's dWd reads it
        self.mark("dec_stack_fwd")
        U, R1, C1 = ptr(ws["U"], SLACK), ptr(ws["R1"], SLACK), ptr(ws["C1"], SLACK)
        sb = SP * pitch
        bias_s = None
        if self.use_bias:
            o = self.spec.off
            bsum = sum(self.flat[o[bn % (3 * i + 2) + ".bias"]:o[bn % (3 * i + 2) + ".bias"] + Sd] for i in range(N)).contiguous()
            ws["bias_skip"] = bsum
            bias_s = ptr(bsum)
        # final conditioning expanded over time (stretch / tile rule on the length-W sequence)
        ws["cf_mode"] = (1, W // Le) if W % Le == 0 else (2, 0)
        enf = enf.contiguous()
        call("wn_cond_expand", ptr(enf), Sd * Le, Le, Sd, lo, T, ws["cf_mode"][0], Le, max(ws["cf_mode"][1], 1), C1, sb, pitch, B, st)

        def chain(b0, nb, s_):
            """skip product -> connection_1 (+ conditioning) -> connection_2 for clips b0 .. b0 + nb - 1 on stream s_"""
            g_ = lambda pack, *a: self._gemm(s_, nb, m, fr(pack), *a)
            o = b0 * sb
            g_("skip", ptr(ws["Z"], SLACK + b0 * zb), None, zb, pitch, lo, T, 0, 0, N * CHd // 32, 0, SP // 16, Sd,
               ptr(ws["U"], SLACK + o), sb, pitch, 0, bias_s, NONE3, NONE3, lo, T, 0)
            g_("c1", ptr(ws["U"], SLACK + o), None, sb, pitch, lo, T, 0, 0, SP // 32, 0, SP // 16, Sd,
               ptr(ws["R1"], SLACK + o), sb, pitch, 0, self._bias("connection_1"), (ptr(ws["C1"], SLACK + o), sb, pitch, lo), NONE3, lo, T, 1)
            g_("c2", ptr(ws["R1"], SLACK + o), None, sb, pitch, lo, T, 0, 0, SP // 32, 0, Q // 16, Q,
               ptr(ws["O"], b0 * Q * W), Q * W, W, -lo, self._bias("connection_2"), NONE3, NONE3, lo, T, 1)
        # two per-clip-group chains, the second on the side stream: a product's half-empty last round of workgroups packs into
        # the other chain's launches (music_amd/engine.py epi_chains; bit-identical results)
        nsplit = min(2, B)
        if nsplit >= 2 and self.overlap_wgrad:
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = _lib.side_stream(self.device)
            side = self._side
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            half = B // 2
            chain(0, half, st)
            with torch.cuda.stream(side):
                chain(half, B - half, _lib.stream())
            ev2 = torch.cuda.Event()
            ev2.record(side)
            main.wait_event(ev2)
        else:
            chain(0, B, st)
        probs = None
        if want_probs:
            probs = torch.empty(B * W, Q, dtype=torch.float32, device=self.device)
            call("wn_chunk_softmax256_fwd", ptr(ws["O"]), ptr(probs), B * W, st)
        ws["probs"] = probs
        self.mark("epilogue_fwd")
        return probs, enc, ws

    # ------------------------------------------------------------------ backward
    def _bwd_workspace(self, ws):
        if ws["bwd"] is not None:
            return ws["bwd"]
        B, T, W, pitch, dev, N = ws["B"], ws["T"], ws["W"], ws["pitch"], self.device, self.N

        def buf(rows):
            return torch.zeros(SLACK + B * rows * pitch + PAD_BACK, dtype=torch.float32, device=dev)

        bw = dict(dO=torch.zeros(B * self.Q * W + PAD_BACK, dtype=torch.float32, device=dev),
                  dR1=buf(self.SP), dU=buf(self.SP), dZ=buf(N * self.CHd), dXd=[buf(self.CHd), buf(self.CHd)],
                  dE=buf(self.BwP), dXe=[buf(self.CHe), buf(self.CHe)],
                  dHe=buf(self.CHe))
        lo = self.rf - 1
        ops = [("c2", lo, T, 1024), ("c1", lo, T, 1024), ("skip", lo, T, 2048), ("bottleneck", lo, T, 512),
               ("de_causal", 1, T, 512), ("en_causal", 1, T, 512)]
        # decoder blocks: the channel-split block kernel (both weight gradients inside the block launch)
        # where it applies, else resblock_bwd + two wgrad launches
        pair = bw["pair"] = ws.get("pair", False)            # both stacks as clip pairs on the 64-channel one-launch blocks
        ms = pair or (self.CHd == 64 and self.mode == _lib.F16X3 and self.mode_b == _lib.BF16X3
                      and os.environ.get("WN_MS_BWD", "1") == "1")
        bw["ms"] = ms
        # ... and the data gradient inside the same launch, as the (P, Q) pair (wn_resblock_bwd_pq), without biases
        # (the conditioning gradient too, as bucket sums on the matrix cores: at most 32 pooled frames - the forward built the
        # bucket bytes then; longer encodings, and WN_AE_COND_MFMA=0, keep wn_resblock_bwd_ms + wn_cond_grad)
        bw["pq"] = pair or (ms and not self.use_bias and os.environ.get("WN_PQ_BWD", "1") == "1" and "cidx" in ws)
        if bw["pq"]:
            bw["PQ"] = [(buf(self.CHd), buf(self.CHd)), (buf(self.CHd), buf(self.CHd))]
        else:
            bw["dfg"] = buf(2 * self.CHd)               # [df;dg] in HBM: only the other block kernels write it
        # encoder blocks: wn_enc_resblock_bwd (dh + both weight gradients in one launch) where it applies
        enc_fused = pair or (self.CHe == 64 and self.mode_b == _lib.BF16X3 and self.fused_encoder)
        bw["enc_fused"] = enc_fused
        # ... and the data gradient inside the same launch, as the (P, Q) pair (wn_enc_resblock_bwd_pq), without biases
        bw["enc_pq"] = pair or (enc_fused and not self.use_bias and os.environ.get("WN_AE_ENC_PQ", "1") == "1")
        if bw["enc_pq"]:
            bw["PQe"] = (bw["PQ"] if bw["pq"] and self.CHe == self.CHd else     # the decoder's pairs are free again by then
                         [(buf(self.CHe), buf(self.CHe)), (buf(self.CHe), buf(self.CHe))])
        sfx = "2_" if pair else ""                            # pair mode: the block-diagonal gradient matrices, B / 2 "clips"
        # one-launch encoder blocks whose dilation is a multiple of 32 hand dx on WHOLE (chain form of wn_enc_resblock_bwd_pq, as
        # wn_resblock_bwd_pq's in music_amd/engine.py): 4 activation tensors per block instead of 6 - the launch is bound by its bytes
        # ... and those with d < 32 too (form 2: adjacent items walked downwards, the Q rows cross from item to item through LDS)
        want_chain = bw["enc_pq"] and os.environ.get("WN_PQ_CHAIN", "1") == "1"
        want_lch = want_chain and os.environ.get("WN_ENC_LCH", "1") == "1"
        bw["enc_chain"] = [(1 if want_chain and _lib.pq_chain_ok(self.off[i + 1], T, B // 2 if pair else B, self.dil[i]) else
                            2 if want_lch and self.dil[i] < 32 else 0) for i in range(N)]
        for i in range(N):
            ench = (-3 - i if bw["enc_chain"][i] else -2) if enc_fused else 512      # -3 - i: layer i in chain form (its own slab count)
            ops += [("de_fg%s%d" % (sfx, i), self.off[i + 1], T, -1 if ms else 512), ("en_dil%s%d" % (sfx, i), self.off[i + 1], T, ench),
                    ("en_dense%s%d" % (sfx, i), self.off[i + 1], T, ench)]
            if i < N - 1:
                ops.append(("de_d%s%d" % (sfx, i), self.off[i + 1], T, -1 if ms else 512))
        plan, desc, so, vs = {}, [], 0, 0
        row_of = {}
        for name, t_lo, t_hi, chunk in ops:
            go, r, c = self.gp_off[name]
            n = r * c
            Bs = B // 2 if pair and chunk < 0 else B
            ns = (_lib.wgrad_slabs(t_lo, t_hi, chunk, B) if chunk > 0 else
                  _lib.ms_slabs(t_lo, t_hi, Bs) if chunk == -1 else _lib.enc_slabs(t_lo, t_hi, Bs) if chunk == -2 else
                  _lib.pq_slabs(t_lo, t_hi, Bs, self.dil[-3 - chunk] if bw["enc_chain"][-3 - chunk] == 1 else 32, True))
            plan[name] = (so, n, chunk)
            row_of[name] = len(desc)
            desc.append([vs, so, ns, n, go, n])
            so += ns * n
            vs += (n + 3) // 4
        # the causal layers' weight gradients from codes (wn_causal_wgrad_codes): their own slab regions, and a second
        # reduction table in which only those two rows differ
        desc_codes = [list(r) for r in desc]
        for name in ("de_causal", "en_causal"):
            go, r, c = self.gp_off[name]
            ns = _lib.causal_codes_slabs(T, B)
            plan[name + "_codes"] = (so, r * c, None)
            desc_codes[row_of[name]][1:3] = [so, ns]
            so += ns * r * c
        bw["slab"] = torch.empty(so, dtype=torch.float32, device=dev)
        bw["plan"], bw["vec"], bw["nops"] = plan, vs, len(desc)
        bw["desc"] = torch.tensor(desc, dtype=torch.int64, device=dev)
        bw["desc_codes"] = torch.tensor(desc_codes, dtype=torch.int64, device=dev)
        ws["bwd"] = bw
        return bw

    def loss_and_grad(self, x, target, cond):
        """Fused training step body (the autoencoder counterpart of engine.loss_and_grad): forward to the logits, ONE
        kernel for chunk softmax + CrossEntropyLoss on the probabilities (wavenet_autoencoder/train.py:146-160) + both
        backward steps, then the backward.  Returns the loss (0-d device tensor); gradients land in self.flat_grad."""
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
        call("wn_chunk_softmax256_ce", ptr(ws["O"]), ptr(target), None, ptr(bw["dO"]), ptr(ws["loss_part"]), n, 1.0 / n,
             _lib.stream())
        self.backward(ws, None)
        loss = ws["loss_part"].sum()
        self._throttle.leave()
        return loss

    def adam_init(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.adam_state = dict(m=torch.zeros_like(self.flat), v=torch.zeros_like(self.flat), t=0,
                               lr=lr, b1=betas[0], b2=betas[1], eps=eps)

    def adam_step(self, gscale=1.0):
        """torch.optim.Adam semantics on the flat parameter buffer (the nn.Parameters are views of it)."""
        s = self.adam_state
        s["t"] += 1
        call("wn_adam_flat", ptr(self.flat), ptr(self.flat_grad), ptr(s["m"]), ptr(s["v"]), self.spec.total,
             s["lr"], s["b1"], s["b2"], s["eps"], 1.0 - s["b1"] ** s["t"], 1.0 - s["b2"] ** s["t"], gscale, _lib.stream())

    def backward_from_dlogits(self, ws):
        self.backward(ws, None)

    def input_grad(self, ws):
        """Gradient of the last backward w.r.t. the module's INPUT (model1.py:137,158: the input feeds the encoder's and the decoder's causal
        conv): din[q][s] = sum over both layers of  W[r][q][1] dx0[r][s] + W[r][q][0] dx0[r][s + 1],  dx0 living on [1, T)."""
        bw = ws["bwd"]
        if bw is None:
            raise RuntimeError("music_amd: input_grad() needs the backward of this forward to have run")
        B, T, pitch, Q = ws["B"], ws["T"], ws["pitch"], self.Q
        parts = []
        for name, ch, key in (("de_causalT", self.CHd, "dXd"), ("en_causalT", self.CHe, "dXe")):
            out = torch.empty(B, Q, T, dtype=torch.float32, device=self.device)
            dx0 = ptr(bw[key][0], SLACK)
            call("wn_chan_gemm", dx0, dx0, ch * pitch, pitch, 1, T, 0, 1, ch // 32, ch // 32, ptr(self.pkb, self.pkb_off[name]), Q // 16, Q,
                 ptr(out), Q * T, T, 0, None, None, 0, 0, 0, None, 0, 0, 0, T, 0, B, self.mode_b, _lib.stream())
            parts.append(out)
        return parts[0].add_(parts[1])

    def backward(self, ws, dprobs):
        """Fills self.flat_grad from d loss / d probabilities (B*W, Q); dprobs None = bw["dO"] already holds
        d loss / d logits (loss_and_grad)."""
        bw = self._bwd_workspace(ws)
        st = _lib.stream()
        # bias gradients (use_bias=True): row sums of the matching output gradient, collected in one small buffer
        # and copied to their flat-parameter positions after the weight gradients were gathered
        if self.use_bias and getattr(self, "_bias_plan", None) is None:
            names = [n for n in self.param_names if n.endswith(".bias")]
            off, o = {}, 0
            for n in names:
                off[n[:-5]] = o
                o += int(np.prod(self.spec.shape[n]))
            idx = np.concatenate([np.arange(self.spec.off[n], self.spec.off[n] + int(np.prod(self.spec.shape[n]))) for n in names])
            self._bias_plan = (off, torch.from_numpy(idx.astype(np.int64)).to(self.device),
                               torch.zeros(o, dtype=torch.float32, device=self.device))
        if self.use_bias:
            b_off, b_idx, b_grad = self._bias_plan

            def bias_grad(name, a, a_bstride, a_pitch, a_shift, rows, t_lo, t_hi, dst=0):
                call("wn_bias_grad", a, a_bstride, a_pitch, a_shift, rows, t_lo, t_hi, ws["B"], ptr(b_grad, b_off[name] + dst), st)
        else:
            def bias_grad(*a, **k):
                pass
        B, T, W, pitch, Le = ws["B"], ws["T"], ws["W"], ws["pitch"], ws["Le"]
        N, CHe, CHd, SP, BwP, Q = self.N, self.CHe, self.CHd, self.SP, self.BwP, self.Q
        Dd, Sd, Rd, Re, De, Bw = self.Dd, self.Sd, self.Rd, self.Re, self.De, self.Bw
        mf, mb = self.mode, self.mode_b
        lo = self.rf - 1
        call("wn_pack_weights", ptr(self.flat), ptr(self.pkb_idx), ptr(self.pkb), self.pkb_idx.numel(), mb, st)
        br = lambda name: ptr(self.pkb, self.pkb_off[name])
        fr = lambda name: ptr(self.pk, self.pk_off[name])
        NONE3 = (None, 0, 0)
        gemm = lambda pack, *a: self._gemm(st, B, mb, br(pack), *a)
        plan = bw["plan"]

        def wgrad(name, *args):
            so, n, chunk = plan[name]
            head, (ldc, t_lo, t_hi) = args[:-3], args[-3:]
            call("wn_wgrad", *head, ptr(bw["slab"], so), ldc, n, t_lo, t_hi, chunk, B, mb, st)

        # the decoder epilogue's three weight gradients only feed the slab reduction at the very end: on a second
        # (high-priority = own hardware queue) stream their half-empty last rounds of workgroups pack into the
        # data-gradient GEMMs beside them, as in music_amd/engine.py (config 4: 1.30 -> ~1.0 ms for this phase)
        main = torch.cuda.current_stream()
        overlap = self.overlap_wgrad
        if overlap and self._side is None:
            self._side = _lib.side_stream(self.device)
        side = self._side if overlap else main

        def wgrad_s(name, *args):
            if not overlap:
                return wgrad(name, *args)
            so, n, chunk = plan[name]
            head, (ldc, t_lo, t_hi) = args[:-3], args[-3:]
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                call("wn_wgrad", *head, ptr(bw["slab"], so), ldc, n, t_lo, t_hi, chunk, B, mb, _lib.stream())

        if dprobs is not None:
            dprobs = dprobs.contiguous()
            call("wn_chunk_softmax256_bwd", ptr(ws["probs"]), ptr(dprobs), ptr(bw["dO"]), B * W, st)
        dO, dR1, dU, dZ = ptr(bw["dO"]), ptr(bw["dR1"], SLACK), ptr(bw["dU"], SLACK), ptr(bw["dZ"], SLACK)
        U, R1, Z = ptr(ws["U"], SLACK), ptr(ws["R1"], SLACK), ptr(ws["Z"], SLACK)
        sb, db, zb, eb = SP * pitch, CHd * pitch, N * CHd * pitch, CHe * pitch
        # ---- decoder epilogue: o = c2(relu(r)), r = c1(relu(u)) + cond_f, u = skip(z)
        wgrad_s("c2", dO, Q * W, W, -lo, W, R1, None, sb, pitch, 0, 0, pitch, SP // 16, Q // 16, 1, SP, lo, T)
        cmode, cq = ws["cf_mode"]
        d_enf = torch.zeros(B, Sd, Le, dtype=torch.float32, device=self.device)
        fused = (os.environ.get("WN_EPI_FUSED_BWD", "1") == "1" and SP == 256 and Q == 256 and (N * CHd // 16) % 3 == 0
                 and mb in (_lib.F16X3, _lib.BF16X3))
        if fused:
            # dR1, dU and dZ in ONE launch per 128-column tile (wn_skip_epilogue_bwd, music_amd/engine.py); the weight gradients that read
            # dR1 / dU follow - connection_1's on the side stream, the skip convs' on the main stream - and the stack starts behind both
            call("wn_skip_epilogue_bwd", dO, Q * W, W, R1, U, sb, pitch, dR1, dU, dZ, zb, br("c2T"), br("c1Tc"), br("skipTc"),
                 N * CHd // 16, N * CHd, Sd, lo, T, B, mb, st)
            call("wn_cond_grad", dR1, sb, pitch, Sd, lo, T, cmode, Le, max(cq, 1), ptr(d_enf), Sd * Le, Le, B, st)
            wgrad_s("c1", dR1, sb, pitch, 0, pitch, U, None, sb, pitch, 0, 0, pitch, SP // 16, SP // 16, 1, SP, lo, T)
            wgrad("skip", dU, sb, pitch, 0, pitch, Z, None, zb, pitch, 0, 0, pitch, N * CHd // 16, SP // 16, 0, N * CHd, lo, T)
            if overlap:
                ev = torch.cuda.Event()
                ev.record(side)
                main.wait_event(ev)
        else:
            gemm("c2T", dO, None, Q * W, W, 0, W, -lo, 0, Q // 32, 0, SP // 16, Sd, dR1, sb, pitch, 0, None, NONE3,
                 (R1, sb, pitch), lo, T, 0)
            call("wn_cond_grad", dR1, sb, pitch, Sd, lo, T, cmode, Le, max(cq, 1), ptr(d_enf), Sd * Le, Le, B, st)
            wgrad_s("c1", dR1, sb, pitch, 0, pitch, U, None, sb, pitch, 0, 0, pitch, SP // 16, SP // 16, 1, SP, lo, T)
            gemm("c1T", dR1, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, SP // 16, Sd, dU, sb, pitch, 0, None, NONE3,
                 (U, sb, pitch), lo, T, 0)
        bias_grad("connection_2", dO, Q * W, W, -lo, Q, lo, T)
        bias_grad("connection_1", dR1, sb, pitch, 0, Sd, lo, T)
        for i in range(N if self.use_bias else 0):
            bias_grad("de_dilation_layer_stack.%d" % (3 * i + 2), dU, sb, pitch, 0, Sd, lo, T)
        if not fused:
            wgrad_s("skip", dU, sb, pitch, 0, pitch, Z, None, zb, pitch, 0, 0, pitch, N * CHd // 16, SP // 16, 0, N * CHd, lo, T)
            gemm("skipT", dU, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, N * CHd // 16, N * CHd, dZ, zb, pitch, 0, None, NONE3,
                 NONE3, lo, T, 0)
        self.mark("ce_epilogue_bwd")
        # ---- decoder stack
        xd = lambda i: self._lay(ws["Xd"], i, CHd, ws)
        dfg = ptr(bw["dfg"], SLACK) if "dfg" in bw else None
        d_tab = torch.zeros(N, B, 2 * CHd, Le, dtype=torch.float32, device=self.device)
        pair = bw["pair"]
        Bp, sfx = (B // 2, "2_") if pair else (B, "")
        if pair:
            d_tab = torch.zeros(N, Bp, 4 * CHd, Le, dtype=torch.float32, device=self.device)     # rows [f: A B | g: A B]
        if bw["pq"] and "cslab" not in bw:
            # per-workgroup bucket sums of every block launch (one region each), added by ONE reduce behind the stack
            import ctypes
            fl = [_lib.load().wn_resblock_bwd_pq_cond_floats(self.off[i + 1], T, Bp) for i in range(N)]
            bw["cs_off"] = (ctypes.c_int64 * N)(*np.concatenate([[0], np.cumsum(fl)[:-1]]).tolist())
            bw["cs_tlo"] = (ctypes.c_int * N)(*[self.off[i + 1] for i in range(N)])
            bw["cslab"] = torch.empty(sum(fl), dtype=torch.float32, device=self.device)
        for i in range(N - 1, -1, -1):
            d, t_lo = self.dil[i], self.off[i + 1]
            dy = ptr(bw["dXd"][(i + 1) % 2], SLACK) if i < N - 1 else None
            mode_c, q = ws["cmodes"][i]
            bias_fg = self._bias("de_dilation_layer_stack.%d" % (3 * i))
            bf = bias_fg + 4 * Dd if bias_fg is not None else None      # filter_gate bias: gate rows first

            def block_bias_grads():
                if not self.use_bias:
                    return
                nm = "de_dilation_layer_stack.%d" % (3 * i)
                bias_grad(nm, dfg + 4 * CHd * pitch, 2 * CHd * pitch, pitch, 0, Dd, t_lo, T)          # gate rows = dg
                bias_grad(nm, dfg, 2 * CHd * pitch, pitch, 0, Dd, t_lo, T, dst=Dd)                    # filter rows = df
                if dy is not None:
                    bias_grad("de_dilation_layer_stack.%d" % (3 * i + 1), dy, db, pitch, 0, Rd, t_lo, T)
            if bw["pq"]:
                p_out, q_out = (ptr(t, SLACK) for t in bw["PQ"][i % 2])
                if i < N - 1:
                    p_in, q_in = (ptr(t, SLACK) for t in bw["PQ"][(i + 1) % 2])
                    dn, p_lo = self.dil[i + 1], self.off[i + 2]
                else:
                    p_in = q_in = None
                    dn = p_lo = 0
                if pair:
                    call("wn_resblock_bwd_pq", xd(i), p_in, q_in, dn, p_lo, ptr(bw["dZ"], SLACK + i * CHd * pitch), p_out, q_out,
                         2 * db, 2 * zb, pitch, fr("de_fg2_%d" % i), br("de_dT2_%d" % i), br("de_pq2_%d" % i), 64, d, t_lo, T, lo,
                         ptr(bw["slab"], plan["de_fg2_%d" % i][0]), ptr(bw["slab"], plan["de_d2_%d" % i][0]) if i < N - 1 else None,
                         ptr(ws["tab"][i]), 4 * CHd * Le, Le, Le, ptr(ws["cidx"][i]), ptr(bw["cslab"], bw["cs_off"][i]), zb, 0, Bp, mf, mb, st)
                else:
                    call("wn_resblock_bwd_pq", xd(i), p_in, q_in, dn, p_lo, ptr(bw["dZ"], SLACK + i * CHd * pitch), p_out, q_out,
                         db, zb, pitch, fr("de_fg%d" % i), br("de_dT%d" % i), br("de_pq%d" % i), CHd, d, t_lo, T, lo,
                         ptr(bw["slab"], plan["de_fg%d" % i][0]), ptr(bw["slab"], plan["de_d%d" % i][0]) if i < N - 1 else None,
                         ptr(ws["tab"][i]), 2 * CHd * Le, Le, Le, ptr(ws["cidx"][i]), ptr(bw["cslab"], bw["cs_off"][i]), 0, 0, B, mf, mb, st)
                if i == 0:
                    call("wn_shift_add", p_out, q_out, ptr(bw["dXd"][0], SLACK), db, pitch, CHd, d, t_lo, self.off[0], T, B, st)
                continue
            if bw["ms"]:
                call("wn_resblock_bwd_ms", xd(i), dy, ptr(bw["dZ"], SLACK + i * CHd * pitch), dfg, db, zb, 2 * CHd * pitch, pitch,
                     fr("de_fg%d" % i), br("de_dT%d" % i), bf, bias_fg, Dd, CHd, d, t_lo, T, lo,
                     ptr(bw["slab"], plan["de_fg%d" % i][0]), ptr(bw["slab"], plan["de_d%d" % i][0]) if i < N - 1 else None,
                     ptr(ws["tab"][i]), 2 * CHd * Le, Le, mode_c, Le, max(q, 1), B, mf, mb, st)
                call("wn_cond_grad", dfg, 2 * CHd * pitch, pitch, 2 * CHd, t_lo, T, mode_c, Le, max(q, 1),
                     ptr(d_tab[i]), 2 * CHd * Le, Le, B, st)
                block_bias_grads()
                gemm("de_fgT%d" % i, dfg, dfg, 2 * CHd * pitch, pitch, t_lo, T, 0, d, 2 * CHd // 32, 2 * CHd // 32, CHd // 16, Rd,
                     ptr(bw["dXd"][i % 2], SLACK), db, pitch, 0, None, (dy, db, pitch, t_lo) if dy else NONE3, NONE3, self.off[i], T, 0)
                continue
            call("wn_resblock_bwd", xd(i), dy, ptr(bw["dZ"], SLACK + i * CHd * pitch), dfg, None,
                 db, zb, 2 * CHd * pitch, db, pitch, fr("de_fg%d" % i), br("de_dT%d" % i), bf, bias_fg, Dd, CHd, d, t_lo, T, lo,
                 ptr(ws["tab"][i]), 2 * CHd * Le, Le, mode_c, Le, max(q, 1), B, mf, mb, st)
            call("wn_cond_grad", dfg, 2 * CHd * pitch, pitch, 2 * CHd, t_lo, T, mode_c, Le, max(q, 1),
                 ptr(d_tab[i]), 2 * CHd * Le, Le, B, st)
            block_bias_grads()
            wgrad("de_fg%d" % i, dfg, 2 * CHd * pitch, pitch, 0, pitch, xd(i), xd(i), db, pitch, -d, 0, pitch,
                  CHd // 16, 2 * CHd // 16, 0, 2 * CHd, t_lo, T)
            if i < N - 1:
                wgrad("de_d%d" % i, dy, db, pitch, 0, pitch, ptr(ws["Z"], SLACK + i * CHd * pitch), None, zb, pitch, 0, 0, pitch,
                      CHd // 16, CHd // 16, 0, CHd, t_lo, T)
            gemm("de_fgT%d" % i, dfg, dfg, 2 * CHd * pitch, pitch, t_lo, T, 0, d, 2 * CHd // 32, 2 * CHd // 32, CHd // 16, Rd,
                 ptr(bw["dXd"][i % 2], SLACK), db, pitch, 0, None, (dy, db, pitch, t_lo) if dy else NONE3, NONE3, self.off[i], T, 0)
        if bw["pq"]:
            rows = 4 * CHd if pair else 2 * CHd
            call("wn_resblock_bwd_pq_cond_reduce", ptr(bw["cslab"]), bw["cs_off"], bw["cs_tlo"], N, T, Bp, Le, ptr(d_tab),
                 Bp * rows * Le, rows * Le, Le, st)
            if pair:                                    # back to per-clip tables, rows [f | g]
                d_tab = d_tab.view(N, Bp, 2, 2, CHd, Le).permute(0, 1, 3, 2, 4, 5).reshape(N, B, 2 * CHd, Le)
        self.mark("dec_stack_bwd")
        x = ws["x_in"]
        codes_path = ws.get("x_codes") is not None

        def causal_wgrad(name, dx0, bs, ch):
            if codes_path:
                codes, scrambled = ws["x_codes"]
                call("wn_causal_wgrad_codes", ptr(codes), 1 if scrambled else 0, dx0, None, 0, 0, bs, pitch, ch, Q, T, B,
                     ptr(bw["slab"], plan[name + "_codes"][0]), st)
            else:
                wgrad(name, dx0, bs, pitch, 0, pitch, ptr(x), ptr(x), Q * T, T, -1, 0, T, Q // 16, ch // 16, 0, 2 * Q, 1, T)
        causal_wgrad("de_causal", ptr(bw["dXd"][0], SLACK), db, CHd)
        bias_grad("de_causal_layer", ptr(bw["dXd"][0], SLACK), db, pitch, 0, Rd, 1, T)
        # ---- conditioning: en_i = cw_i enc + b (rows in the reference order: gate first), enf = cfw enc + b
        d_en = torch.cat([d_tab[:, :, CHd:CHd + Dd], d_tab[:, :, :Dd]], 2)            # (N,B,2Dd,Le) reference row order
        d_enc = torch.einsum("nck,nbcl->bkl", ws["cw"], d_en) + torch.einsum("ck,bcl->bkl", ws["cfw"][:, :, 0], d_enf)
        d_enc = d_enc.contiguous()
        # ---- encoder: avgpool -> bottleneck -> N blocks -> causal
        dE = ptr(bw["dE"], SLACK)
        call("wn_avgpool_bwd", ptr(d_enc), Bw * Le, Le, lo, self.pool, Le, Bw, dE, BwP * pitch, pitch, T, B, st)
        xe = lambda i: self._lay(ws["Xe"], i, CHe, ws)
        he = lambda i: self._lay(ws["He"], i, CHe, ws)
        wgrad("bottleneck", dE, BwP * pitch, pitch, 0, pitch, xe(N), None, eb, pitch, 0, 0, pitch, CHe // 16, BwP // 16, 0, CHe, lo, T)
        bias_grad("bottleneck_layer", dE, BwP * pitch, pitch, 0, Bw, lo, T)
        dxe = [ptr(t, SLACK) for t in bw["dXe"]]
        dHe = ptr(bw["dHe"], SLACK)
        gemm("bottleneckT", dE, None, BwP * pitch, pitch, lo, T, 0, 0, BwP // 32, 0, CHe // 16, Re, dxe[N % 2], eb, pitch, 0, None,
             NONE3, NONE3, lo, T, 0)
        self.mark("de_causal_cond_bottleneck_bwd")
        for i in range(N - 1, -1, -1):
            d, t_lo = self.dil[i], self.off[i + 1]
            y_lo = lo if i == N - 1 else t_lo                     # the top gradient only exists on the crop
            dy = dxe[(i + 1) % 2]
            if bw["enc_pq"]:
                # the whole backward of the block in one launch; dx travels as the unshifted pair (P, Q)
                p_out, q_out = (ptr(t, SLACK) for t in bw["PQe"][i % 2])
                chain = bw["enc_chain"][i]
                if i == 0 and chain:
                    p_out = dxe[0]                                # a first block in chain form hands dx_0 on whole: straight to the causal layer's buffer
                if i < N - 1:
                    p_in, q_in = (ptr(t, SLACK) for t in bw["PQe"][(i + 1) % 2])
                    dn, p_lo = self.dil[i + 1], self.off[i + 2]
                    if bw["enc_chain"][i + 1]:                    # the block above handed dx on whole (valid from ITS t_lo - d = this t_lo)
                        q_in, dn, p_lo = None, 0, t_lo
                else:
                    p_in, q_in, dn, p_lo = dy, None, 0, y_lo
                if pair:
                    call("wn_enc_resblock_bwd_pq", xe(i), p_in, q_in, dn, p_lo, he(i), p_out, q_out, 2 * eb, 2 * eb, pitch,
                         br("en_denseT2_%d" % i), br("en_pq2_%d" % i), 64, d, t_lo, T, ptr(bw["slab"], plan["en_dil2_%d" % i][0]),
                         ptr(bw["slab"], plan["en_dense2_%d" % i][0]), chain, Bp, mb, st)
                else:
                    call("wn_enc_resblock_bwd_pq", xe(i), p_in, q_in, dn, p_lo, he(i), p_out, q_out, eb, eb, pitch,
                         br("en_denseT%d" % i), br("en_pq%d" % i), CHe, d, t_lo, T, ptr(bw["slab"], plan["en_dil%d" % i][0]),
                         ptr(bw["slab"], plan["en_dense%d" % i][0]), chain, B, mb, st)
                if i == 0 and not chain:
                    call("wn_shift_add", p_out, q_out, dxe[0], eb, pitch, CHe, d, t_lo, self.off[0], T, B, st)
                continue
            if bw["enc_fused"]:
                # dh, dW1 = sum dy relu(h)^T and dWdil = sum dh [relu x(t-d) | relu x(t)]^T in one launch
                call("wn_enc_resblock_bwd", xe(i), dy, he(i), dHe, eb, eb, eb, pitch, br("en_denseT%d" % i), CHe, d, t_lo, T, y_lo,
                     ptr(bw["slab"], plan["en_dil%d" % i][0]), ptr(bw["slab"], plan["en_dense%d" % i][0]), B, mb, st)
                bias_grad("en_dense_layer_stack.%d" % i, dy, eb, pitch, 0, Re, y_lo, T)
                bias_grad("en_dilation_layer_stack.%d" % i, dHe, eb, pitch, 0, De, t_lo, T)
                gemm("en_dilT%d" % i, dHe, dHe, eb, pitch, t_lo, T, 0, d, CHe // 32, CHe // 32, CHe // 16, Re, dxe[i % 2], eb, pitch, 0,
                     None, (dy, eb, pitch, y_lo), (xe(i), eb, pitch), self.off[i], T, 0)
                continue
            # dh = (W1^T dy) * [h > 0];  dW1 = sum dy relu(h)^T
            wgrad("en_dense%d" % i, dy, eb, pitch, 0, pitch, he(i), None, eb, pitch, 0, 0, pitch, CHe // 16, CHe // 16, 1, CHe, y_lo, T)
            bias_grad("en_dense_layer_stack.%d" % i, dy, eb, pitch, 0, Re, y_lo, T)
            gemm("en_denseT%d" % i, dy, None, eb, pitch, y_lo, T, 0, 0, CHe // 32, 0, CHe // 16, De, dHe, eb, pitch, 0, None, NONE3,
                 (he(i), eb, pitch), t_lo, T, 0)
            # dWdil = sum dh [relu(x)(t-d) | relu(x)(t)]^T
            wgrad("en_dil%d" % i, dHe, eb, pitch, 0, pitch, xe(i), xe(i), eb, pitch, -d, 0, pitch, CHe // 16, CHe // 16, 1, 2 * CHe, t_lo, T)
            bias_grad("en_dilation_layer_stack.%d" % i, dHe, eb, pitch, 0, De, t_lo, T)
            # dx_i[t] = [x_i > 0] (Wdil1^T dh[t] + Wdil0^T dh[t+d]) + dy[t]
            gemm("en_dilT%d" % i, dHe, dHe, eb, pitch, t_lo, T, 0, d, CHe // 32, CHe // 32, CHe // 16, Re, dxe[i % 2], eb, pitch, 0,
                 None, (dy, eb, pitch, y_lo), (xe(i), eb, pitch), self.off[i], T, 0)
        self.mark("enc_stack_bwd")
        causal_wgrad("en_causal", dxe[0], eb, CHe)
        bias_grad("en_causal_layer", dxe[0], eb, pitch, 0, Re, 1, T)
        if overlap:
            ev_join = torch.cuda.Event()
            ev_join.record(side)
            main.wait_event(ev_join)
        call("wn_reduce_slabs", ptr(bw["desc_codes"] if codes_path else bw["desc"]), bw["nops"], bw["vec"], ptr(bw["slab"]),
             ptr(self.gpack), st)
        if pair:
            call("wn_gather_grads2", ptr(self.gpack), ptr(self.gidx_pa), ptr(self.gidx_pb), ptr(self.flat_grad), self.spec.total, st)
        else:
            call("wn_gather_grads", ptr(self.gpack), ptr(self.gidx), ptr(self.flat_grad), self.spec.total, st)
        if self.use_bias:
            self.flat_grad.index_copy_(0, b_idx, b_grad)
        self.mark("en_causal_slab_reduce")


class _AutoencoderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, grad_on, wave_sample, cond, *params):
        eng = net._engine_for(wave_sample.device)
        x = wave_sample.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        else:
            tag = getattr(wave_sample, "_wn_codes", None)        # one-hot built from codes (see _AutoencoderEngine.forward)
            if tag is not None and wave_sample._version == tag[2]:
                x._wn_codes = (tag[0], tag[1], x._version, tag[3])
        probs, enc, ws = eng.forward(x, cond)
        net.last_encoding = enc
        ctx.eng, ctx.ws, ctx.gen = eng, ws, ws["gen"]
        ctx.loss_hook = net._last_hook = _losshook.make(eng, ws, grad_on)
        ctx.hold = WorkspaceHold(ws) if (grad_on and any(ctx.needs_input_grad)) else None      # see music_amd/model.py
        return probs.detach()            # (an alias: the workspace's own reference must not carry the autograd node)

    @staticmethod
    def backward(ctx, dprobs):
        eng, ws = ctx.eng, ctx.ws
        if ws.get("gen") != ctx.gen:
            raise RuntimeError("music_amd.wavenet_autoencoder: activations were overwritten by a later forward")
        if not _losshook.backward(ctx.loss_hook, eng, ws, dprobs):          # (the loss ran fused: see _losshook.py)
            eng.backward(ws, dprobs)
        if ctx.hold is not None:
            ctx.hold.release()
        g = eng.flat_grad.clone()
        grads = []
        for name in eng.param_names:
            o, shp = eng.spec.off[name], eng.spec.shape[name]
            grads.append(g[o:o + int(np.prod(shp))].view(shp))
        din = eng.input_grad(ws) if ctx.needs_input_grad[2] else None       # (the reference's two causal nn.Conv1d give it, model1.py:137,158)
        return (None, None, din, None) + tuple(grads)


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
        # nn.CrossEntropyLoss()(net(x), target) with its default arguments runs fused (music_amd/_losshook.py); False: torch's own
        self.fuse_loss = True
        self._last_hook = None
        # (forward, backward) arithmetic of the matrix-core products, as on `wavenet`; ("bf16x3", "bf16x3") gives the forward float32's exponent range
        # (an un-normalised ReLU encoder can leave f16's: DESIGN section 5) at 2^-17 instead of 2^-22 per product
        self.precision = ("f16x3", "bf16x3")

    def __getstate__(self):
        # copy.deepcopy / pickle / torch.save(module): the engine (HIP streams, workspaces, ctypes plans) stays behind and is rebuilt
        # on the copy's first forward; the parameters travel as tensors
        state = self.__dict__.copy()
        state["_engine"] = None
        state["_last_hook"] = None
        return state

    def _calc_receptive_field(self):
        return (self.filter_width - 1) * (sum(self.dilations) + 1) + 1

    def _draw_conditioning(self):
        """The 31 per-forward conditioning convs, drawn on the CPU from the global RNG in the
        reference's order (model1.py:178 per layer, :216 final)."""
        # nn.Conv1d(Bw, C, 1).reset_parameters() without the module around it: the same two uniform_ draws per conv, in the same
        # order, with the bounds computed as torch.nn.init does (kaiming_uniform_(a = sqrt(5)) on the weight, then
        # U(-1/sqrt(fan_in), 1/sqrt(fan_in)) on the bias) - bit-identical tensors (tests/test_host_logic.py) at a fraction of the
        # Python objects (41 modules per forward otherwise)
        n = len(self.dilations)
        fan_in = self.en_bottleneck_width                      # kernel size 1
        gain = math.sqrt(2.0 / (1 + math.sqrt(5) ** 2))
        bound_w = math.sqrt(3.0) * (gain / math.sqrt(fan_in))
        bound_b = 1 / math.sqrt(fan_in)
        cond = []
        for i in range(n + 1):
            c_out = 2 * self.de_dilation_channel if i < n else self.de_skip_channel
            w = torch.empty(c_out, fan_in, 1).uniform_(-bound_w, bound_w)
            b = torch.empty(c_out).uniform_(-bound_b, bound_b)
            cond.append((w, b))
        return cond

    def _engine_for(self, device):
        if device.type != "cuda":
            raise RuntimeError("music_amd.wavenet_autoencoder runs on an MI355X (ROCm) device only; there is no CPU path")
        if not hasattr(self, "precision"):                   # (a module pickled before the attribute existed)
            self.precision = ("f16x3", "bf16x3")
        eng = self._engine
        p0 = next(self.parameters())
        if eng is None or eng.device != device or p0.data_ptr() != eng.flat.data_ptr() or getattr(eng, "mode_names", None) != tuple(self.precision):
            if any(p.device != device for p in self.parameters()):
                raise RuntimeError("music_amd.wavenet_autoencoder: parameters and input are on different devices")
            if any(p.dtype != torch.float32 for p in self.parameters()):
                raise TypeError("music_amd.wavenet_autoencoder: parameters must be float32 (got %s)"
                                % next(p.dtype for p in self.parameters() if p.dtype != torch.float32))
            # the specialised kernels cover filter_width 2, 256 quantisation channels and up to 64 residual / dilation channels on
            # both sides (what the reference ships and BASELINE.json names); any other constructor argument takes the general plan
            fast = (self.filter_width == 2 and self.quantization_channel == 256 and
                    max(self.en_residual_channel, self.en_dilation_channel, self.de_residual_channel, self.de_dilation_channel) <= 64)
            if fast:
                eng = _AutoencoderEngine(self, device, mode=self.precision[0], mode_bwd=self.precision[1])
            else:
                try:
                    from .ae_generic import GenericAutoencoderEngine
                except ImportError:
                    from music_amd.ae_generic import GenericAutoencoderEngine
                eng = GenericAutoencoderEngine(self, device, mode=self.precision[0], mode_bwd=self.precision[1])
            eng.mode_names = tuple(self.precision)
            self._engine = eng
        return eng

    def forward(self, wave_sample):
        batch_size, original_channels, seq_len = wave_sample.size()
        output_width = seq_len - self.receptive_field + 1
        if output_width <= 0:
            raise ValueError("wave sample not long enough")
        self._engine_for(wave_sample.device)
        cond = self._draw_conditioning()
        self._last_hook = None
        out = _AutoencoderFunction.apply(self, torch.is_grad_enabled(), wave_sample, cond, *list(self.parameters()))
        hook, self._last_hook = self._last_hook, None
        return _losshook.wrap(out, hook) if self.fuse_loss else out
