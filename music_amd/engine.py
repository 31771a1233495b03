"""Host-side execution plan of the WaveNet hot path on one MI355X.

`WaveNetEngine` owns the flat parameter / gradient buffers, the fragment-packing index maps, the
per-(batch, length) workspaces, and sequences the C-ABI kernels of libwavenet_hip.so for

    forward   wavenet/model.py:86-145          (probabilities, chunk-softmax semantics)
    backward  autograd of the above             (SURVEY Appendix B formulas)
    train_step  wavenet/train.py:171-182        (forward + CE-on-probs + backward [+ all-reduce] + Adam)

PyTorch is used for device memory and streams only.  Nothing here imports oracle/.
"""
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr

SLACK = 64          # floats in front of every activation buffer
PAD_BACK = 2048     # floats behind


def _pad(n, m):
    return (n + m - 1) // m * m


def pack_positions(mt, ks, chained):
    """Logical A-fragment order -> (row, k) of the effective weight matrix.
    p = ((m*ks + s)*64 + lane)*8 + j ; row = 16m + (lane&15) ; k = kmap(s, lane>>4, j)."""
    m, s, lane, j = np.indices((mt, ks, 64, 8))
    row = 16 * m + (lane & 15)
    q = lane >> 4
    if chained:
        k = 32 * s + 16 * (j >> 2) + 4 * q + (j & 3)
    else:
        k = 32 * s + 8 * q + j
    return row.reshape(-1), k.reshape(-1)


def pack_index(weff, chained=False):
    """weff: int32 [Mp, Kp] of flat-parameter offsets (-1 = zero).  Returns idx[p] (int32)."""
    mp, kp = weff.shape
    assert mp % 16 == 0 and kp % 32 == 0
    row, k = pack_positions(mp // 16, kp // 32, chained)
    return weff[row, k].astype(np.int32)


class WorkspacePool:
    """(B, T) -> workspaces of that shape.  A workspace whose activations a pending autograd node still needs is HELD
    (hold() in the Function's forward, release() after its backward or when the graph dies): the next forward of the
    same shape then gets ANOTHER workspace instead of overwriting it, so a module can have several forwards in flight
    (gradient accumulation over micro-batches, the reference's autograd semantics, wavenet/model.py:86-145).  Shapes are
    evicted one at a time, least recently used first, never one that is held."""
    MAX_SHAPES = 4
    MAX_PER_SHAPE = 4

    def __init__(self, make):
        from collections import OrderedDict
        self._make = make
        self._d = OrderedDict()
        self._last = {}

    def peek(self, B, T):
        """The workspace the LAST forward of this shape used (held or not) - what a caller inspects after a step; a
        fresh one if there has been none."""
        ws = self._last.get((B, T))
        return ws if ws is not None else self.get(B, T)

    def get(self, B, T):
        """A workspace for the NEXT forward of this shape: the first one no pending backward holds, else a new one."""
        key = (B, T)
        lst = self._d.get(key)
        if lst is None:
            if len(self._d) >= self.MAX_SHAPES:
                for k, wl in self._d.items():
                    if not any(w.get("held") for w in wl):
                        del self._d[k]
                        self._last.pop(k, None)
                        break
            lst = self._d[key] = []
        self._d.move_to_end(key)
        for ws in lst:
            if not ws.get("held"):
                self._last[key] = ws
                return ws
        if len(lst) >= self.MAX_PER_SHAPE:
            # the reference never runs out (autograd just keeps allocating): take the OLDEST waiting forward's workspace over, say
            # so once, and let a backward that still arrives for it fail loudly (its generation no longer matches)
            import warnings
            warnings.warn("music_amd: %d forwards of shape %s are waiting for their backward; the oldest one's activations are "
                          "reused (run backward(), drop the outputs, or use torch.no_grad() for inference)" % (len(lst), key))
            ws = min(lst, key=lambda w: w.get("gen", 0))
            ws["held"] = False
            self._last[key] = ws
            return ws
        ws = self._make(B, T)
        lst.append(ws)
        self._last[key] = ws
        return ws

    def clear(self):
        self._d.clear()
        self._last.clear()

    def __len__(self):
        return sum(len(v) for v in self._d.values())


class WorkspaceHold:
    """Keeps a workspace out of the pool's hands while an autograd node needs it; released explicitly after backward or
    by garbage collection of the node (an output that was dropped without a backward)."""

    def __init__(self, ws):
        self.ws, self.gen = ws, ws["gen"]
        ws["held"] = True

    def release(self):
        if self.ws is not None and self.ws.get("gen") == self.gen:
            self.ws["held"] = False
        self.ws = None

    def __del__(self):
        self.release()


class _Spec:
    """Offsets of every reference parameter inside the flat buffer (state_dict order)."""

    def __init__(self, named_shapes):
        self.off, self.shape = {}, {}
        o = 0
        for name, shape in named_shapes:
            self.off[name] = o
            self.shape[name] = tuple(shape)
            o += int(np.prod(shape))
        self.total = o

    def conv(self, name):
        """int64 array [O, I, k] of flat offsets of a Conv1d weight."""
        shp = self.shape[name]
        return self.off[name] + np.arange(int(np.prod(shp)), dtype=np.int64).reshape(shp)


class WaveNetEngine:
    def __init__(self, dilations, residual_channels, dilation_channels, skip_channels,
                 quantization_channels=256, filter_width=2, use_bias=False,
                 mode_fwd="f16x3", mode_bwd="bf16x3", device=None):
        if filter_width != 2:
            raise NotImplementedError("HIP path implements filter_width == 2 (the reference's only configuration)")
        if quantization_channels != 256:
            raise NotImplementedError("HIP path implements quantization_channels == 256 (chunk softmax width)")
        self.dil = [int(d) for d in dilations]
        self.N = len(self.dil)
        self.R, self.D, self.S, self.Q = residual_channels, dilation_channels, skip_channels, quantization_channels
        self.fused_loss_ok = True           # model.py: nn.CrossEntropyLoss on the module's output may run as wn_chunk_softmax256_ce (Q = 256 here)
        self.use_bias = bool(use_bias)
        self.CH = _pad(max(self.R, self.D), 32)
        if self.CH not in (32, 64):
            raise NotImplementedError("HIP path supports residual/dilation channels <= 64")
        self.SP = _pad(self.S, 32)
        self.rf = sum(self.dil) + 2
        self.off = [1]                                   # first valid absolute time of x_i
        for d in self.dil:
            self.off.append(self.off[-1] + d)
        assert self.off[-1] == self.rf - 1
        self.mode_fwd = _lib.MODE_NAMES[mode_fwd] if isinstance(mode_fwd, str) else mode_fwd
        self.mode_bwd = _lib.MODE_NAMES[mode_bwd] if isinstance(mode_bwd, str) else mode_bwd
        # <= 32 channels: TWO clips side by side are one 64-row tensor, and with block-diagonal packs the stack runs on the
        # 64-channel block kernels (one-launch backward included) on exactly the bytes of the 32-channel tensors - the
        # zero blocks cost matrix time only.  Even batches, no biases, (f16x3, bf16x3); WN_PAIR32=0: the 32-channel kernels.
        self.pair_ok = (self.CH == 32 and not self.use_bias and self.mode_fwd == _lib.F16X3 and self.mode_bwd == _lib.BF16X3
                        and os.environ.get("WN_PAIR32", "1") == "1" and os.environ.get("WN_PQ_BWD", "1") == "1")
        self.device = torch.device(device if device is not None else "cuda")
        _lib.load()
        self._build_spec()
        self._build_packs()
        self._ws = WorkspacePool(self._make_workspace)
        self._gen = 0
        self.adam_state = None
        self.marks = None            # list of (name, torch.cuda.Event) when profiling is on
        self.mark_only = None        # optional set of mark names to keep
        # weight-gradient launches that only feed the final slab reduction run on a second HIP stream: the
        # epilogue's three (2 rounds of workgroups at 80 % fill each) then pack into the data-gradient GEMMs'
        # idle CUs (epilogue backward 1.25 -> 1.00 ms)
        self.overlap_wgrad = True
        # the forward epilogue's three products as this many per-clip-group chains, every second one on the side stream (1 = one
        # chain on the main stream; bit-identical results: tests/test_gpu_switches.py)
        self.epi_chains = 2
        # ... or all three in ONE launch per 128-column tile (wn_skip_epilogue_fwd, round 6; 256 skip / 256 quantisation channels, x3
        # modes; WN_EPI_FUSED=0 = the three launches above)
        self.epi_fused = os.environ.get("WN_EPI_FUSED", "1") == "1"
        self._throttle = _lib.StepThrottle()
        self.epi_fused_bwd = os.environ.get("WN_EPI_FUSED_BWD", "1") == "1"
        # Channel-split backward block with both weight gradients in the launch (wn_resblock_bwd_ms):
        # 64 padded channels, (f16x3, bf16x3) only; None = whenever it applies (WN_MS_BWD=0 turns it off)
        self.ms_bwd = None
        self._side = None
        self.fine_marks = False
        # the unfused backward reads the forward's z (stored on the full valid range) for dWd
        self.z_from_fwd = True

    def mark(self, name):
        """Record a timing event on the current stream (only when self.marks is a list; self.mark_only, if set, limits
        the events to those names - every event costs a marker packet between two kernels)."""
        if self.marks is not None and (self.mark_only is None or name in self.mark_only):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev))

    def fmark(self, name):
        """Per-kernel timing marks of the epilogue (tools/kbench.py epi); off unless self.fine_marks."""
        if self.fine_marks:
            self.mark(name)

    # ------------------------------------------------------------------ parameters
    def _build_spec(self):
        names = [("causal_layer.weight", (self.R, self.Q, 2))]
        if self.use_bias:
            names.append(("causal_layer.bias", (self.R,)))
        for i in range(self.N):
            for k, shp in enumerate([(self.D, self.R, 2), (self.D, self.R, 2), (self.R, self.D, 1), (self.S, self.D, 1)]):
                names.append(("dilation_layer_stack.%d.weight" % (4 * i + k), shp))
                if self.use_bias:
                    names.append(("dilation_layer_stack.%d.bias" % (4 * i + k), (shp[0],)))
        names.append(("post_process_1.weight", (self.S, self.S, 1)))
        if self.use_bias:
            names.append(("post_process_1.bias", (self.S,)))
        names.append(("post_process_2.weight", (self.Q, self.S, 1)))
        if self.use_bias:
            names.append(("post_process_2.bias", (self.Q,)))
        self.spec = _Spec(names)
        self.param_names = [n for n, _ in names]
        dev = self.device
        self.flat = torch.zeros(self.spec.total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(self.spec.total, dtype=torch.float32, device=dev)

    def param_view(self, name, grad=False):
        o, shp = self.spec.off[name], self.spec.shape[name]
        n = int(np.prod(shp))
        return (self.flat_grad if grad else self.flat)[o:o + n].view(shp)

    def load_state_dict_tensors(self, sd):
        with torch.no_grad():
            for n in self.param_names:
                self.param_view(n).copy_(sd[n])

    def _bias_ptr(self, name):
        if not self.use_bias:
            return None
        return ptr(self.flat, self.spec.off[name])

    # ------------------------------------------------------------------ packs
    def _build_packs(self):
        sp, CH, N, R, D, S, Q, SP = self.spec, self.CH, self.N, self.R, self.D, self.S, self.Q, self.SP
        fwd, bwd = [], []            # lists of (name, idx array)
        gp = []                      # gradient C matrices: (name, rows, cols)

        def full(m, k):
            return np.full((m, k), -1, dtype=np.int64)

        # 1. causal: rows R, K = [tap0 Q | tap1 Q]
        wc = sp.conv("causal_layer.weight")                 # [R,Q,2]
        w = full(CH, 2 * Q)
        w[:R, :Q] = wc[:, :, 0]
        w[:R, Q:] = wc[:, :, 1]
        fwd.append(("causal", pack_index(w)))
        gp.append(("causal", CH, 2 * Q))
        # 1b. its transpose for the gradient w.r.t. the INPUT (input_grad): rows Q, K = [tap1^T over dx0[t] | tap0^T over dx0[t+1]]
        w = full(Q, 2 * CH)
        w[:, :R] = wc[:, :, 1].T
        w[:, CH:CH + R] = wc[:, :, 0].T
        bwd.append(("causalT", pack_index(w)))
        for i in range(N):
            wf = sp.conv("dilation_layer_stack.%d.weight" % (4 * i))       # [D,R,2]
            wg = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 1))
            wd = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 2))   # [R,D,1]
            # 2. fg: rows [f(D) pad CH | g(D) pad CH], K = [tap0 CH | tap1 CH]
            w = full(2 * CH, 2 * CH)
            for h, src in enumerate((wf, wg)):
                w[h * CH:h * CH + D, 0:R] = src[:, :, 0]
                w[h * CH:h * CH + D, CH:CH + R] = src[:, :, 1]
            fwd.append(("fg%d" % i, pack_index(w)))
            gp.append(("fg%d" % i, 2 * CH, 2 * CH))
            # 3. dense (chained k order: B fragments come from the z accumulators)
            w = full(CH, CH)
            w[:R, :D] = wd[:, :, 0]
            fwd.append(("d%d" % i, pack_index(w, chained=True)))
            gp.append(("d%d" % i, CH, CH))
            # 10. Wd^T (rows D, K = R)
            bwd.append(("dT%d" % i, pack_index(np.ascontiguousarray(w.T))))
            # 11. data gradient of fg: rows R, K = [W1^T over (df|dg) | W0^T over (df|dg)]
            w = full(CH, 4 * CH)
            for h, src in enumerate((wf, wg)):
                w[:R, h * CH:h * CH + D] = src[:, :, 1].T
                w[:R, 2 * CH + h * CH:2 * CH + h * CH + D] = src[:, :, 0].T
            bwd.append(("fgT%d" % i, pack_index(w)))
            # 11b. the same weights as two UNSHIFTED row blocks for the one-launch backward block (wn_resblock_bwd_pq):
            #      rows [0,CH) = W1^T (-> P), rows [CH,2CH) = W0^T (-> Q), K = (df | dg)
            w = full(2 * CH, 2 * CH)
            for h, src in enumerate((wf, wg)):
                w[:R, h * CH:h * CH + D] = src[:, :, 1].T
                w[CH:CH + R, h * CH:h * CH + D] = src[:, :, 0].T
            bwd.append(("pq%d" % i, pack_index(w)))
            if self.pair_ok:
                # 12. the same four matrices block-diagonal for two clips side by side (rows / columns of a 32-block: clip A
                #     then clip B) - what the 64-channel block kernels multiply in pair mode
                def diag(m32, rb, cb):
                    """m32: [rb*32][cb*32] blocks of 32 x 32 -> [rb*64][cb*64] with every block doubled on the diagonal"""
                    out = full(rb * 64, cb * 64)
                    for a_ in range(rb):
                        for b_ in range(cb):
                            blk = m32[a_ * 32:(a_ + 1) * 32, b_ * 32:(b_ + 1) * 32]
                            for c_ in range(2):
                                out[a_ * 64 + c_ * 32:a_ * 64 + (c_ + 1) * 32, b_ * 64 + c_ * 32:b_ * 64 + (c_ + 1) * 32] = blk
                    return out
                wfg32 = full(2 * CH, 2 * CH)
                for h, src in enumerate((wf, wg)):
                    wfg32[h * CH:h * CH + D, 0:R] = src[:, :, 0]
                    wfg32[h * CH:h * CH + D, CH:CH + R] = src[:, :, 1]
                wd32 = full(CH, CH)
                wd32[:R, :D] = wd[:, :, 0]
                fwd.append(("fg2_%d" % i, pack_index(diag(wfg32, 2, 2))))
                fwd.append(("d2_%d" % i, pack_index(diag(wd32, 1, 1), chained=True)))
                bwd.append(("dT2_%d" % i, pack_index(diag(np.ascontiguousarray(wd32.T), 1, 1))))
                bwd.append(("pq2_%d" % i, pack_index(diag(w, 2, 2))))
                gp.append(("fg2_%d" % i, 4 * CH, 4 * CH))
                gp.append(("d2_%d" % i, 2 * CH, 2 * CH))
        # 4. skip over the concatenated z-crops: rows S, K = N*CH
        w = full(SP, N * CH)
        for i in range(N):
            ws = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 3))   # [S,D,1]
            w[:S, i * CH:i * CH + D] = ws[:, :, 0]
        fwd.append(("skip", pack_index(w)))
        gp.append(("skip", SP, N * CH))
        bwd.append(("skipT", pack_index(np.ascontiguousarray(w.T))))      # rows N*CH, K = SP
        bwd.append(("skipTc", pack_index(np.ascontiguousarray(w.T), chained=True)))     # chained k order: the fused backward epilogue
        # 5/6. post-process
        p1 = sp.conv("post_process_1.weight")[:, :, 0]
        p2 = sp.conv("post_process_2.weight")[:, :, 0]
        w = full(SP, SP)
        w[:S, :S] = p1
        fwd.append(("p1", pack_index(w)))
        fwd.append(("p1c", pack_index(w, chained=True)))     # chained k order: the fused forward epilogue takes relu(U) out of the accumulators
        gp.append(("p1", SP, SP))
        bwd.append(("p1T", pack_index(np.ascontiguousarray(w.T))))
        bwd.append(("p1Tc", pack_index(np.ascontiguousarray(w.T), chained=True)))
        w = full(Q, SP)
        w[:, :S] = p2
        fwd.append(("p2", pack_index(w)))
        fwd.append(("p2c", pack_index(w, chained=True)))
        gp.append(("p2", Q, SP))
        bwd.append(("p2T", pack_index(np.ascontiguousarray(w.T))))        # rows SP, K = Q

        dev = self.device

        def finish(lst, mode):
            halfs_per_frag = 1024 if mode in (_lib.F16X3, _lib.BF16X3) else 512
            offs, o = {}, 0
            for name, idx in lst:
                offs[name] = o * halfs_per_frag // 512          # offset in halfs of the packed buffer
                o += len(idx)
            idx_all = torch.from_numpy(np.concatenate([i for _, i in lst]).astype(np.int32)).to(dev)
            buf = torch.zeros(o * halfs_per_frag // 512, dtype=torch.int16, device=dev)
            return offs, idx_all, buf

        self.pk_f_off, self.pk_f_idx, self.pk_f = finish(fwd, self.mode_fwd)
        self.pk_b_off, self.pk_b_idx, self.pk_b = finish(bwd, self.mode_bwd)

        # gradient matrices + gather map (flat parameter element -> offset in gpack)
        self.gp_off, o = {}, 0
        for name, r, c in gp:
            self.gp_off[name] = (o, r, c)
            o += r * c
        bias_rows = {}
        if self.use_bias:
            for name in self.param_names:
                if name.endswith(".bias"):
                    bias_rows[name] = o
                    o += _pad(self.spec.shape[name][0], 4)
        self.gp_bias_off = bias_rows
        self.gpack = torch.zeros(o, dtype=torch.float32, device=dev)
        gidx = np.full(self.spec.total, -1, dtype=np.int64)

        def put(pname, mat_off):
            """mat_off: int array with the same shape as the parameter holding gpack offsets."""
            po = self.spec.off[pname]
            gidx[po:po + mat_off.size] = mat_off.reshape(-1)

        o0, r, c = self.gp_off["causal"]
        rows = np.arange(R)[:, None, None]
        put("causal_layer.weight", o0 + rows * c + (np.arange(2)[None, None, :] * Q + np.arange(Q)[None, :, None]))
        for i in range(N):
            o0, r, c = self.gp_off["fg%d" % i]
            for h in range(2):
                put("dilation_layer_stack.%d.weight" % (4 * i + h),
                    o0 + (h * CH + np.arange(D)[:, None, None]) * c +
                    (np.arange(2)[None, None, :] * CH + np.arange(R)[None, :, None]))
            o0, r, c = self.gp_off["d%d" % i]
            put("dilation_layer_stack.%d.weight" % (4 * i + 2),
                o0 + np.arange(R)[:, None, None] * c + np.arange(D)[None, :, None])
            o0, r, c = self.gp_off["skip"]
            put("dilation_layer_stack.%d.weight" % (4 * i + 3),
                o0 + np.arange(S)[:, None, None] * c + (i * CH + np.arange(D)[None, :, None]))
        o0, r, c = self.gp_off["p1"]
        put("post_process_1.weight", o0 + np.arange(S)[:, None, None] * c + np.arange(S)[None, :, None])
        o0, r, c = self.gp_off["p2"]
        put("post_process_2.weight", o0 + np.arange(Q)[:, None, None] * c + np.arange(S)[None, :, None])
        for name, bo in bias_rows.items():
            put(name, bo + np.arange(self.spec.shape[name][0]))
        assert (gidx >= 0).all()
        self.gidx = torch.from_numpy(gidx.astype(np.int32)).to(dev)
        if self.pair_ok:
            # pair mode: the stack's weight gradients come out of the 64-channel block kernels as block-diagonal matrices -
            # a weight's gradient is the sum of its two copies (wn_gather_grads2); everything else as above
            ga, gb = gidx.copy(), np.full(self.spec.total, -1, dtype=np.int64)

            def put2(pname, off_a, off_b):
                po = self.spec.off[pname]
                ga[po:po + off_a.size] = off_a.reshape(-1)
                gb[po:po + off_b.size] = off_b.reshape(-1)
            for i in range(N):
                o0, r, c = self.gp_off["fg2_%d" % i]
                for h in range(2):
                    rows = h * 64 + np.arange(D)[:, None, None]
                    cols = np.arange(2)[None, None, :] * 64 + np.arange(R)[None, :, None]
                    put2("dilation_layer_stack.%d.weight" % (4 * i + h), o0 + rows * c + cols, o0 + (rows + 32) * c + cols + 32)
                o0, r, c = self.gp_off["d2_%d" % i]
                rows, cols = np.arange(R)[:, None, None], np.arange(D)[None, :, None]
                put2("dilation_layer_stack.%d.weight" % (4 * i + 2), o0 + rows * c + cols, o0 + (rows + 32) * c + cols + 32)
            self.gidx_pa = torch.from_numpy(ga.astype(np.int32)).to(dev)
            self.gidx_pb = torch.from_numpy(gb.astype(np.int32)).to(dev)
        # causal weight re-laid as [tap][q][ch] for the forward from codes (wn_causal_fwd_codes): a gather map over the
        # flat parameter buffer (-1 = padded channel, reads as 0)
        wt = np.full((2, Q, CH), -1, dtype=np.int64)
        wt[:, :, :R] = wc.transpose(2, 1, 0)
        self.wt_idx = torch.from_numpy(wt.reshape(-1).astype(np.int32)).to(dev)
        self.wt = torch.zeros(2 * Q * CH, dtype=torch.float32, device=dev)

    def pack_weights(self):
        st = _lib.stream()
        call("wn_pack_weights", ptr(self.flat), ptr(self.pk_f_idx), ptr(self.pk_f), self.pk_f_idx.numel(), self.mode_fwd, st)
        call("wn_pack_weights", ptr(self.flat), ptr(self.pk_b_idx), ptr(self.pk_b), self.pk_b_idx.numel(), self.mode_bwd, st)

    # ------------------------------------------------------------------ workspace
    def workspace(self, B, T):
        """The workspace the last forward of this shape used (what callers inspect after a step; WorkspacePool.peek).  A
        forward takes its own through WorkspacePool.get: the first one no pending backward holds."""
        return self._ws.peek(B, T)

    def _make_workspace(self, B, T):
        dev = self.device
        pitch = _pad(T, 256) + 512        # tiles of 512 columns may overhang T by < 512
        W = T - self.rf + 1

        def buf(rows):
            t = torch.zeros(SLACK + B * rows * pitch + PAD_BACK, dtype=torch.float32, device=dev)
            return t

        ws = dict(B=B, T=T, W=W, pitch=pitch)
        ws["X"] = torch.zeros(SLACK + (self.N + 1) * B * self.CH * pitch + PAD_BACK, dtype=torch.float32, device=dev)
        ws["Z"] = buf(self.N * self.CH)
        ws["U"] = buf(self.SP)
        ws["H"] = buf(self.SP)
        ws["O"] = torch.zeros(B * self.Q * W + PAD_BACK, dtype=torch.float32, device=dev)
        ws["bwd"] = None
        # which backward blocks this workspace is planned for: decided ONCE here (slab layout, (P, Q) buffers, and whether
        # the forward has to store z on each block's whole range for the fallback backward) - a switch flipped later
        # takes effect with the next workspace, never half-way between a forward and its backward
        ws["ms"], ws["pq"] = self._use_ms(), self._use_pq()
        ws["pair"] = self.pair_ok and B % 2 == 0
        # the FORWARD block of the 64-channel form takes 512 columns per workgroup: below ~200 workgroups (B / 2 pairs x T / 512)
        # the 32-channel forward fills the chip better (same X / Z layout either way; the backward pairs regardless)
        ws["pair_fwd"] = ws["pair"] and (B // 2) * ((T + 511) // 512) >= 200
        return ws

    def _bwd_workspace(self, ws):
        if ws["bwd"] is not None:
            return ws["bwd"]
        B, pitch, W, dev = ws["B"], ws["pitch"], ws["W"], self.device

        def buf(rows):
            return torch.zeros(SLACK + B * rows * pitch + PAD_BACK, dtype=torch.float32, device=dev)

        bw = dict(dO=torch.zeros(B * self.Q * W + PAD_BACK, dtype=torch.float32, device=dev),
                  dH=buf(self.SP), dU=buf(self.SP), dZ=buf(self.N * self.CH),
                  dX=[buf(self.CH), buf(self.CH)], dfg=[buf(2 * self.CH), buf(2 * self.CH)],
                  zs=[buf(self.CH), buf(self.CH)])
        # weight-gradient slabs: every wgrad workgroup writes its partial C with plain stores,
        # one batched kernel then sums the slabs of all ops in a fixed order (deterministic)
        T, lo = ws["T"], self.rf - 1
        ms = ws["ms"]
        bw["ms"] = ms
        bw["pq"] = ws["pq"]
        pair = bw["pair"] = ws["pair"]
        if bw["pq"] or pair:
            bw["PQ"] = [(buf(self.CH), buf(self.CH)), (buf(self.CH), buf(self.CH))]
        # (time chunk per workgroup of the epilogue's weight gradients; 512 / 256 / 2048-4096 measured on one box in round 5: step 4.36 /
        # 4.46 / 4.35 against 4.34 ms - shorter chunks pay in slabs to reduce, longer ones in the blocks that run beside them)
        # round 6 (the fused epilogue: the weight gradients of post_process_1 and of the skip convs now run on their own, beside each
        # other, behind the fused launch): 2048 for all three - step 4.141 against 4.19 - 4.23 at (1024, 1024, 2048), 4.22 at (512, 512, 1024),
        # 4.25 at 3264, 4.38 at 4096 (profiles/r06_ab_wgrad_chunks.json)
        ck = [int(v) for v in os.environ.get("WN_EPI_WGRAD_CHUNKS", "2048,2048,2048").split(",")]
        if "WN_EPI_WGRAD_CHUNKS" not in os.environ:
            # ... cut into EQUAL chunks of at most that: 12960 columns are 6 x 2048 + 672, and the launch takes as long as its full chunks
            # (7 x 1856: epilogue backward 0.861 -> 0.845 ms)
            span = T - (lo & ~31)

            def even(c):
                n = -(-span // c)                      # chunks per clip
                return -(-(-(-span // n)) // 32) * 32    # their common length, a multiple of the 32-sample k-step
            ck = [even(c) for c in ck]
        ops = [("p2", lo, T, ck[0]), ("p1", lo, T, ck[1]), ("skip", lo, T, ck[2])]
        # one-launch blocks whose dilation is a multiple of 32 hand dx on WHOLE (chain form of wn_resblock_bwd_pq: the Q rows
        # of an item are the carry of the next item of its chain); decided here, once per workspace, with the slab counts
        Bp = B // 2 if pair else B
        want = os.environ.get("WN_PQ_CHAIN", "1") == "1"       # 0: every block hands the (P, Q) pair on (round 3's form)
        bw["chain"] = [want and bool(bw["pq"] or pair) and _lib.pq_chain_ok(self.off[i + 1], T, Bp, self.dil[i]) for i in range(self.N)]
        for i in range(self.N):
            if pair:                                          # the 64-channel one-launch block on B / 2 clip pairs
                ops.append(("fg2_%d" % i, self.off[i + 1], T, (-3, i)))
                if i < self.N - 1:
                    ops.append(("d2_%d" % i, self.off[i + 1], T, (-3, i)))
                continue
            kind = (-3, i) if bw["pq"] else (-1 if ms else 512)
            ops.append(("fg%d" % i, self.off[i + 1], T, kind))
            if i < self.N - 1:
                ops.append(("d%d" % i, self.off[i + 1], T, kind))
        ops.append(("causal", 1, T, 512))
        # the same gradient from integer codes (wn_causal_wgrad_codes) when the input is a one-hot this engine / the
        # loader built: its own slab region and its own reduction table (only the last row differs)
        ops.append(("causal_codes", 1, T, None))
        plan, desc, so, vs = {}, [], 0, 0
        nslab = {}
        for name, t_lo, t_hi, chunk in ops:
            if isinstance(chunk, tuple):
                nslab[name] = _lib.pq_slabs(t_lo, t_hi, Bp, self.dil[chunk[1]], bw["chain"][chunk[1]])
        for name, t_lo, t_hi, chunk in ops:
            go, r, c = self.gp_off["causal" if name == "causal_codes" else name]
            n = r * c
            if chunk is None:
                ns = _lib.causal_codes_slabs(T, B)
            elif isinstance(chunk, tuple):                    # one-launch block (on clips or clip pairs): one slab per workgroup of ITS plan
                ns = nslab[name]
            elif chunk > 0:
                ns = _lib.wgrad_slabs(t_lo, t_hi, chunk, B)
            else:                                             # channel-split block: one slab per workgroup
                ns = _lib.ms_slabs(t_lo, t_hi, B)
            plan[name] = (so, n, chunk, ns, go)
            if name == "causal_codes":
                desc_codes = desc[:-1] + [[desc[-1][0], so, ns, n, go, n]]
            else:
                desc.append([vs, so, ns, n, go, n])
                vs += (n + 3) // 4
            so += ns * n
        bw["slab"] = torch.empty(so, dtype=torch.float32, device=dev)
        bw["slab_plan"], bw["slab_vec"], bw["slab_nops"] = plan, vs, len(desc)
        bw["slab_desc"] = torch.tensor(desc, dtype=torch.int64, device=dev)
        bw["slab_desc_codes"] = torch.tensor(desc_codes, dtype=torch.int64, device=dev)
        ws["bwd"] = bw
        return bw

    def _use_ms(self):
        ok = self.CH == 64 and self.mode_fwd == _lib.F16X3 and self.mode_bwd == _lib.BF16X3
        if self.ms_bwd is None:
            return ok and os.environ.get("WN_MS_BWD", "1") == "1"
        if self.ms_bwd and not ok:
            raise NotImplementedError("ms_bwd needs 64 padded channels and precision (f16x3, bf16x3)")
        return bool(self.ms_bwd)

    def _use_pq(self):
        """The whole per-block backward, data gradient included, in one launch (wn_resblock_bwd_pq): wherever the
        two-role block applies and there are no biases (their gradients are sums over [df;dg], which that kernel
        never writes out).  WN_PQ_BWD=0: resblock_bwd_rw_k + chan_gemm_rw_k."""
        return self._use_ms() and not self.use_bias and os.environ.get("WN_PQ_BWD", "1") == "1"

    def _x(self, ws, i):
        return ptr(ws["X"], SLACK + i * ws["B"] * self.CH * ws["pitch"])

    # ------------------------------------------------------------------ forward
    def forward_logits(self, x, ws=None, codes=None):
        """x: (B,Q,T) float32 contiguous on the device.  Runs causal conv, the residual stack, the
        skip product and both post-process convs; leaves the pre-softmax (B,Q,W) in ws['O'].
        codes = (int32 (B,T) device tensor, scrambled) instead of x: the one-hot is never built."""
        if x is None:
            B, T = codes[0].shape
            Q = self.Q
            assert codes[0].is_cuda and codes[0].dtype == torch.int32 and codes[0].is_contiguous()
        else:
            B, Q, T = x.shape
            assert Q == self.Q and x.is_contiguous() and x.dtype == torch.float32 and x.is_cuda
        W = T - self.rf + 1
        if W <= 0:
            raise ValueError("wave sample not long enough")          # wavenet/model.py:100-101
        ws = ws or self._ws.get(B, T)
        st = _lib.stream()
        CH, N, SP, pitch, mf = self.CH, self.N, self.SP, ws["pitch"], self.mode_fwd
        fr = lambda name: ptr(self.pk_f, self.pk_f_off[name])
        xb = CH * pitch
        self._gen += 1
        ws["gen"] = self._gen
        ws["x_in"] = x
        # a one-hot built from integer codes by onehot() / the loader carries them along: the backward then forms the
        # causal layer's weight gradient by scatter instead of streaming the dense tensor (still valid only while the
        # tensor has not been written to since)
        tag = getattr(x, "_wn_codes", None) if x is not None else None
        ws["x_codes"] = (codes[0], bool(codes[1])) if x is None else None
        if tag is not None:
            codes, scrambled, version, cversion = tag
            if (x._version == version and codes._version == cversion and codes.is_cuda and codes.dtype == torch.int32 and codes.is_contiguous() and
                    tuple(codes.shape) == (B, T)):
                ws["x_codes"] = (codes, scrambled)
        # what the backward re-checks: the causal layer's weight gradient is formed later from these same tensors
        ws["x_ver"] = (None if x is None else x._version, None if ws["x_codes"] is None else ws["x_codes"][0]._version)
        # causal conv (wavenet/model.py:104): x0[t] = W0 in[t-1] + W1 in[t], t in [1,T)
        if ws["x_codes"] is not None:
            # the input is the one-hot of known codes: a gather of weight columns (the dense tensor is not read)
            codes, scrambled = ws["x_codes"]
            call("wn_gather_grads", ptr(self.flat), ptr(self.wt_idx), ptr(self.wt), self.wt.numel(), st)
            call("wn_causal_fwd_codes", ptr(codes), 1 if scrambled else 0, ptr(self.wt), self._bias_ptr("causal_layer.bias"),
                 self.R, self._x(ws, 0), xb, pitch, CH, Q, T, B, st)
        else:
            call("wn_chan_gemm", ptr(x), ptr(x), Q * T, T, 0, T, -1, 0, Q // 32, Q // 32, fr("causal"), CH // 16, self.R,
                 self._x(ws, 0), xb, pitch, 0, self._bias_ptr("causal_layer.bias"),
                 None, 0, 0, 0, None, 0, 0, 1, T, 0, B, mf, st)
        zb = N * CH * pitch
        self.mark("causal_fwd")
        # z: the skip product needs it on the crop [rf-1, T) only, and the two-role / one-launch backward blocks recompute
        # it on the CU.  Only the fallback backward (resblock_bwd_k + wgrad_k: 32 padded channels, x1 modes) reads the
        # forward's z for dWd on the block's whole valid range [off_{i+1}, T) (19 % more z; it saves that path a second
        # copy written by its recompute kernel).
        z_whole = self.z_from_fwd and not ws["ms"] and not ws["pair"]      # (pair: the one-launch backward recomputes z)
        for i, d in enumerate(self.dil):
            bn = "dilation_layer_stack.%d.bias"
            if ws["pair_fwd"]:
                # two clips per 64-row tensor, block-diagonal packs; the second clip's z rows go to its own slice (z_half = zb)
                call("wn_resblock_fwd", self._x(ws, i), self._x(ws, i + 1), ptr(ws["Z"], SLACK + i * CH * pitch), 2 * xb, 2 * zb,
                     pitch, fr("fg2_%d" % i), fr("d2_%d" % i), None, None, None, 64, 64, 64, d, self.off[i + 1], T, self.rf - 1,
                     1 if i < N - 1 else 0, None, 0, 0, 0, 0, 0, None, 0, None, zb, B // 2, mf, st)
                continue
            call("wn_resblock_fwd", self._x(ws, i), self._x(ws, i + 1), ptr(ws["Z"], SLACK + i * CH * pitch), xb, zb, pitch,
                 fr("fg%d" % i), fr("d%d" % i), self._bias_ptr(bn % (4 * i)), self._bias_ptr(bn % (4 * i + 1)),
                 self._bias_ptr(bn % (4 * i + 2)), self.D, self.R, CH, d, self.off[i + 1], T,
                 self.off[i + 1] if z_whole else self.rf - 1,
                 1 if i < N - 1 else 0, None, 0, 0, 0, 0, 0, None, 0, None, 0, B, mf, st)
        self.mark("stack_fwd")
        lo = self.rf - 1
        bias_s = None
        if self.use_bias:
            bs = sum(self.param_view("dilation_layer_stack.%d.bias" % (4 * i + 3)) for i in range(N))
            ws["bias_skip"] = bs.contiguous()
            bias_s = ptr(ws["bias_skip"])
        def chain(b0, nb, s_):
            """skip product -> post-processing 1 -> 2 for clips b0 .. b0 + nb - 1 on stream s_"""
            call("wn_chan_gemm", ptr(ws["Z"], SLACK + b0 * zb), None, zb, pitch, lo, T, 0, 0, N * CH // 32, 0, fr("skip"), SP // 16, self.S,
                 ptr(ws["U"], SLACK + b0 * SP * pitch), SP * pitch, pitch, 0, bias_s, None, 0, 0, 0, None, 0, 0, lo, T, 0, nb, mf, s_)
            self.fmark("f_skip")
            call("wn_chan_gemm", ptr(ws["U"], SLACK + b0 * SP * pitch), None, SP * pitch, pitch, lo, T, 0, 0, SP // 32, 0, fr("p1"), SP // 16, self.S,
                 ptr(ws["H"], SLACK + b0 * SP * pitch), SP * pitch, pitch, 0, self._bias_ptr("post_process_1.bias"),
                 None, 0, 0, 0, None, 0, 0, lo, T, 1, nb, mf, s_)
            self.fmark("f_p1")
            call("wn_chan_gemm", ptr(ws["H"], SLACK + b0 * SP * pitch), None, SP * pitch, pitch, lo, T, 0, 0, SP // 32, 0, fr("p2"), Q // 16, Q,
                 ptr(ws["O"], b0 * Q * W), Q * W, W, -lo, self._bias_ptr("post_process_2.bias"),
                 None, 0, 0, 0, None, 0, 0, lo, T, 1, nb, mf, s_)
        nsplit = min(int(self.epi_chains), B)
        if self.epi_fused and SP == 256 and Q == 256 and (N * CH // 32) % 2 == 0 and mf in (_lib.F16X3, _lib.BF16X3):
            # the three products in ONE launch per 128-column tile, U and H handed on chip (wn_skip_epilogue_fwd, ABI v5)
            call("wn_skip_epilogue_fwd", ptr(ws["Z"], SLACK), zb, pitch, N * CH // 32, fr("skip"), bias_s,
                 ptr(ws["U"], SLACK), ptr(ws["H"], SLACK), SP * pitch, fr("p1c"), self._bias_ptr("post_process_1.bias"),
                 fr("p2c"), self._bias_ptr("post_process_2.bias"), ptr(ws["O"]), Q * W, W, self.S, Q, lo, T, B, mf, st)
        elif nsplit >= 2:
            # the three products of each part of the clips as a chain of its own, every second chain on the side stream: a
            # product's half-empty last round of workgroups (408 tiles of 256 columns on 256 CUs) then packs into the other
            # chain's launches (0.435-0.445 vs 0.466-0.469 ms with two chains)
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = _lib.side_stream(self.device)
            side = self._side
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            bounds = [B * k // nsplit for k in range(nsplit + 1)]
            for k in range(nsplit):
                b0, nb = bounds[k], bounds[k + 1] - bounds[k]
                if k % 2 == 1:
                    with torch.cuda.stream(side):
                        chain(b0, nb, _lib.stream())
                else:
                    chain(b0, nb, st)
            ev2 = torch.cuda.Event()
            ev2.record(side)
            main.wait_event(ev2)
        else:
            chain(0, B, st)
        self.mark("epilogue_fwd")
        return ws

    def forward(self, x):
        """wavenet/model.py:86-145 -> probabilities (B*W, Q) (fresh tensor)."""
        self.pack_weights()
        ws = self.forward_logits(x)
        B, W = ws["B"], ws["W"]
        probs = torch.empty(B * W, self.Q, dtype=torch.float32, device=self.device)
        call("wn_chunk_softmax256_fwd", ptr(ws["O"]), ptr(probs), B * W, _lib.stream())
        ws["probs"] = probs
        return probs, ws

    # ------------------------------------------------------------------ backward
    def backward_from_dlogits(self, ws):
        """ws['bwd']['dO'] holds d loss / d pre-softmax (B,Q,W).  Fills self.flat_grad."""
        bw = self._bwd_workspace(ws)
        st = _lib.stream()
        B, T, W, pitch = ws["B"], ws["T"], ws["W"], ws["pitch"]
        CH, N, SP, Q, mb, mf = self.CH, self.N, self.SP, self.Q, self.mode_bwd, self.mode_fwd
        br = lambda name: ptr(self.pk_b, self.pk_b_off[name])
        fr = lambda name: ptr(self.pk_f, self.pk_f_off[name])
        lo = self.rf - 1
        xb, zb, sb = CH * pitch, N * CH * pitch, SP * pitch
        plan = bw["slab_plan"]

        def wgrad(name, *args):
            """args = everything of wn_wgrad up to and including relu_b, then ldc, t_lo, t_hi"""
            so, n, chunk = plan[name][:3]
            head, (ldc, t_lo, t_hi) = args[:-3], args[-3:]
            call("wn_wgrad", *head, ptr(bw["slab"], so), ldc, n, t_lo, t_hi, chunk, B, mb, st)
        dO, dH, dU, dZ = ptr(bw["dO"]), ptr(bw["dH"], SLACK), ptr(bw["dU"], SLACK), ptr(bw["dZ"], SLACK)
        U, H, Z = ptr(ws["U"], SLACK), ptr(ws["H"], SLACK), ptr(ws["Z"], SLACK)
        main = torch.cuda.current_stream()
        overlap = self.overlap_wgrad
        if overlap and self._side is None:
            # HIGH priority = its own hardware queue.  A default-priority stream is dealt one of a few hardware
            # queues round-robin; once RCCL has created its streams (torchrun) the side stream landed on the
            # MAIN stream's queue and the overlap silently disappeared (epilogue_bwd 0.99 -> 1.26 ms).
            self._side = _lib.side_stream(self.device)
        side = self._side if overlap else main

        def on_side(fn):
            """Run fn on the side stream after everything enqueued so far on the main stream."""
            if overlap:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
            with torch.cuda.stream(side):
                fn(_lib.stream())

        def wgrad_s(name, *args):
            so, n, chunk = plan[name][:3]
            head, (ldc, t_lo, t_hi) = args[:-3], args[-3:]
            on_side(lambda s2: call("wn_wgrad", *head, ptr(bw["slab"], so), ldc, n, t_lo, t_hi, chunk, B, mb, s2))

        # weight gradients of the epilogue run on the side stream as soon as their operands exist
        wgrad_s("p2", dO, Q * W, W, -lo, W, H, None, sb, pitch, 0, 0, pitch, SP // 16, Q // 16, 1, SP, lo, T)
        self.fmark("b_wgrad_p2")
        if (self.epi_fused_bwd and SP == 256 and Q == 256 and (N * CH // 16) % 3 == 0 and mb in (_lib.F16X3, _lib.BF16X3)
                and not self.use_bias):
            # dH, dU and dZ in ONE launch per 128-column tile (wn_skip_epilogue_bwd, ABI v5); the two weight gradients that read
            # dH / dU follow on the side stream
            call("wn_skip_epilogue_bwd", dO, Q * W, W, H, U, sb, pitch, dH, dU, dZ, zb, br("p2T"), br("p1Tc"), br("skipTc"),
                 N * CH // 16, N * CH, self.S, lo, T, B, mb, st)
            self.fmark("b_fused")
            wgrad_s("p1", dH, sb, pitch, 0, pitch, U, None, sb, pitch, 0, 0, pitch, SP // 16, SP // 16, 1, SP, lo, T)
            # the skip weight gradient on the MAIN stream, beside post_process_1's on the side stream, and the stack starts when both are
            # done: left to run beside the stack (WN_EPI_BWD_ORDER=0) they stretch every backward-block launch - a block launch wants all
            # 256 CUs at once - for the same total (bench A/B on one box: 0.95 + 2.0 against 0.52 + 2.6 ms), and the stack's own time
            # (what `roofline` is computed from) would read 40 % high
            order = os.environ.get("WN_EPI_BWD_ORDER", "1")
            if order == "1":
                wgrad("skip", dU, sb, pitch, 0, pitch, Z, None, zb, pitch, 0, 0, pitch, N * CH // 16, SP // 16, 0, N * CH, lo, T)
            else:
                wgrad_s("skip", dU, sb, pitch, 0, pitch, Z, None, zb, pitch, 0, 0, pitch, N * CH // 16, SP // 16, 0, N * CH, lo, T)
            if order == "1" and overlap:
                ev = torch.cuda.Event()
                ev.record(side)
                main.wait_event(ev)
            self.mark("epilogue_bwd")
            return self._backward_stack(ws, bw, st, main, side, overlap, plan, wgrad)
        # dH = (P2^T dO) * [H > 0]
        call("wn_chan_gemm", dO, None, Q * W, W, 0, W, -lo, 0, Q // 32, 0, br("p2T"), SP // 16, self.S,
             dH, sb, pitch, 0, None, None, 0, 0, 0, H, sb, pitch, lo, T, 0, B, mb, st)
        self.fmark("b_p2T")
        wgrad_s("p1", dH, sb, pitch, 0, pitch, U, None, sb, pitch, 0, 0, pitch, SP // 16, SP // 16, 1, SP, lo, T)
        self.fmark("b_wgrad_p1")
        # dU = (P1^T dH) * [U > 0]
        call("wn_chan_gemm", dH, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, br("p1T"), SP // 16, self.S,
             dU, sb, pitch, 0, None, None, 0, 0, 0, U, sb, pitch, lo, T, 0, B, mb, st)
        self.fmark("b_p1T")
        wgrad_s("skip", dU, sb, pitch, 0, pitch, Z, None, zb, pitch, 0, 0, pitch, N * CH // 16, SP // 16, 0, N * CH, lo, T)
        self.fmark("b_wgrad_skip")
        # dZ = Ws^T dU   (all N crops at once)
        call("wn_chan_gemm", dU, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, br("skipT"), N * CH // 16, N * CH,
             dZ, zb, pitch, 0, None, None, 0, 0, 0, None, 0, 0, lo, T, 0, B, mb, st)
        if self.use_bias:
            bo = self.gp_bias_off
            call("wn_bias_grad", dO, Q * W, W, -lo, Q, lo, T, B, ptr(self.gpack, bo["post_process_2.bias"]), st)
            call("wn_bias_grad", dH, sb, pitch, 0, self.S, lo, T, B, ptr(self.gpack, bo["post_process_1.bias"]), st)
            for i in range(N):
                call("wn_bias_grad", dU, sb, pitch, 0, self.S, lo, T, B,
                     ptr(self.gpack, bo["dilation_layer_stack.%d.bias" % (4 * i + 3)]), st)
        self.mark("epilogue_bwd")
        return self._backward_stack(ws, bw, st, main, side, overlap, plan, wgrad)

    def _backward_stack(self, ws, bw, st, main, side, overlap, plan, wgrad):
        """The residual stack's backward, the causal layer's weight gradient and the slab reduction (second half of backward_from_dlogits)."""
        B, T, W, pitch = ws["B"], ws["T"], ws["W"], ws["pitch"]
        CH, N, SP, Q, mb, mf = self.CH, self.N, self.SP, self.Q, self.mode_bwd, self.mode_fwd
        br = lambda name: ptr(self.pk_b, self.pk_b_off[name])
        fr = lambda name: ptr(self.pk_f, self.pk_f_off[name])
        lo = self.rf - 1
        xb, zb, sb = CH * pitch, N * CH * pitch, SP * pitch
        # The per-layer weight-gradient products only feed the slab reduction at the very end, so
        # they run on a second HIP stream next to the data-gradient chain
        # (resblock_bwd -> dx product -> next block); dfg / z scratch is double-buffered for that.
        ev_w = [None, None]          # side-stream completion of the wgrads that read scratch buffer k
        ev_prev = None               # ... of the previous layer's wgrads (they read dX[(i+1)%2])
        for i in range(N - 1, -1, -1):
            d = self.dil[i]
            t_lo = self.off[i + 1]
            k = i % 2
            dfg, zs = ptr(bw["dfg"][k], SLACK), ptr(bw["zs"][k], SLACK)
            dy = ptr(bw["dX"][(i + 1) % 2], SLACK) if i < N - 1 else None
            bn = "dilation_layer_stack.%d.bias"
            if overlap and ev_w[k] is not None:
                main.wait_event(ev_w[k])
            if bw["pq"] or bw["pair"]:
                p_out, q_out = (ptr(t, SLACK) for t in bw["PQ"][i % 2])
                chain = 1 if bw["chain"][i] else 0
                if i == 0 and chain:
                    # a first block in chain form (d_0 a multiple of 32) hands dx_0 on WHOLE: straight into the buffer the causal
                    # layer's weight gradient reads (same layout and stride as the (P, Q) buffers)
                    p_out = ptr(bw["dX"][0], SLACK)
                if i < N - 1:
                    p_in, q_in = (ptr(t, SLACK) for t in bw["PQ"][(i + 1) % 2])
                    dn, p_lo = self.dil[i + 1], self.off[i + 2]
                    if bw["chain"][i + 1]:                    # the block above handed dx on whole (valid from ITS t_lo - d = this t_lo)
                        q_in, dn, p_lo = None, 0, t_lo
                else:
                    p_in = q_in = None
                    dn = p_lo = 0
                if bw["pair"]:
                    # clip pairs on the 64-channel block: block-diagonal packs, the second clip's dz rows in its own slice
                    call("wn_resblock_bwd_pq", self._x(ws, i), p_in, q_in, dn, p_lo, ptr(bw["dZ"], SLACK + i * CH * pitch),
                         p_out, q_out, 2 * xb, 2 * zb, pitch, fr("fg2_%d" % i), br("dT2_%d" % i), br("pq2_%d" % i), 64, d, t_lo, T, lo,
                         ptr(bw["slab"], plan["fg2_%d" % i][0]), ptr(bw["slab"], plan["d2_%d" % i][0]) if i < N - 1 else None,
                         None, 0, 0, 0, None, None, zb, chain, B // 2, mf, mb, st)
                else:
                    call("wn_resblock_bwd_pq", self._x(ws, i), p_in, q_in, dn, p_lo, ptr(bw["dZ"], SLACK + i * CH * pitch),
                         p_out, q_out, xb, zb, pitch, fr("fg%d" % i), br("dT%d" % i), br("pq%d" % i), CH, d, t_lo, T, lo,
                         ptr(bw["slab"], plan["fg%d" % i][0]), ptr(bw["slab"], plan["d%d" % i][0]) if i < N - 1 else None,
                         None, 0, 0, 0, None, None, 0, chain, B, mf, mb, st)
                self.fmark("b_block")
                if i == 0 and not chain:
                    # dx_0 for the causal layer: the pair made whole once (19 us; the scatter from codes can also take the
                    # pair as it is - wn_causal_wgrad_codes(dx_q) - but its doubled, masked tile loads cost the same 20 us)
                    call("wn_shift_add", p_out, q_out, ptr(bw["dX"][0], SLACK), xb, pitch, CH, d, t_lo, self.off[0], T, B, st)
                continue
            if bw["ms"]:
                call("wn_resblock_bwd_ms", self._x(ws, i), dy, ptr(bw["dZ"], SLACK + i * CH * pitch), dfg, xb, zb, 2 * CH * pitch,
                     pitch, fr("fg%d" % i), br("dT%d" % i), self._bias_ptr(bn % (4 * i)), self._bias_ptr(bn % (4 * i + 1)),
                     self.D, CH, d, t_lo, T, lo, ptr(bw["slab"], plan["fg%d" % i][0]),
                     ptr(bw["slab"], plan["d%d" % i][0]) if i < N - 1 else None, None, 0, 0, 0, 0, 0, B, mf, mb, st)
                self.fmark("b_block")
                if self.use_bias:
                    bo = self.gp_bias_off
                    call("wn_bias_grad", dfg, 2 * CH * pitch, pitch, 0, self.D, t_lo, T, B, ptr(self.gpack, bo[bn % (4 * i)]), st)
                    call("wn_bias_grad", ptr(bw["dfg"][k], SLACK + CH * pitch), 2 * CH * pitch, pitch, 0, self.D, t_lo, T, B,
                         ptr(self.gpack, bo[bn % (4 * i + 1)]), st)
                    if i < N - 1:
                        call("wn_bias_grad", dy, xb, pitch, 0, self.R, t_lo, T, B, ptr(self.gpack, bo[bn % (4 * i + 2)]), st)
                # input range [t_lo, pitch): the [df;dg] scratch holds zeros beyond T (never written: every store is
                # masked to t < T, and pitch >= T + 512 >= T + d), so the waves at the end of a clip need not take
                # the guarded-load path for the shifted tap.  (Below t_lo the guard IS needed: dx exists on
                # [t_lo - d, T) and the unshifted tap must read zeros there, not another layer's stale rows.)
                call("wn_chan_gemm", dfg, dfg, 2 * CH * pitch, pitch, t_lo, pitch, 0, d, 2 * CH // 32, 2 * CH // 32, br("fgT%d" % i),
                     CH // 16, self.R, ptr(bw["dX"][i % 2], SLACK), xb, pitch, 0, None,
                     dy, xb, pitch, t_lo, None, 0, 0, self.off[i], T, 0, B, mb, st)
                self.fmark("b_dx")
                continue
            zs = None if self.z_from_fwd else zs
            call("wn_resblock_bwd", self._x(ws, i), dy, ptr(bw["dZ"], SLACK + i * CH * pitch), dfg, zs,
                 xb, zb, 2 * CH * pitch, xb, pitch, fr("fg%d" % i), br("dT%d" % i),
                 self._bias_ptr(bn % (4 * i)), self._bias_ptr(bn % (4 * i + 1)), self.D, CH, d, t_lo, T, lo,
                 None, 0, 0, 0, 0, 0, B, mf, mb, st)
            if overlap:
                ev_r = torch.cuda.Event()
                ev_r.record(main)
                side.wait_event(ev_r)
            with torch.cuda.stream(side):
                st2 = _lib.stream()
                so, n, chunk = plan["fg%d" % i][:3]
                call("wn_wgrad", dfg, 2 * CH * pitch, pitch, 0, pitch, self._x(ws, i), self._x(ws, i), xb, pitch, -d, 0, pitch,
                     CH // 16, 2 * CH // 16, 0, ptr(bw["slab"], so), 2 * CH, n, t_lo, T, chunk, B, mb, st2)
                if i < N - 1:
                    so, n, chunk = plan["d%d" % i][:3]
                    zsrc, zstr = (ptr(ws["Z"], SLACK + i * CH * pitch), zb) if self.z_from_fwd else (zs, xb)
                    call("wn_wgrad", dy, xb, pitch, 0, pitch, zsrc, None, zstr, pitch, 0, 0, pitch, CH // 16, CH // 16, 0,
                         ptr(bw["slab"], so), CH, n, t_lo, T, chunk, B, mb, st2)
                if self.use_bias:
                    bo = self.gp_bias_off
                    call("wn_bias_grad", dfg, 2 * CH * pitch, pitch, 0, self.D, t_lo, T, B, ptr(self.gpack, bo[bn % (4 * i)]), st2)
                    call("wn_bias_grad", ptr(bw["dfg"][k], SLACK + CH * pitch), 2 * CH * pitch, pitch, 0, self.D, t_lo, T, B,
                         ptr(self.gpack, bo[bn % (4 * i + 1)]), st2)
                    if i < N - 1:
                        call("wn_bias_grad", dy, xb, pitch, 0, self.R, t_lo, T, B, ptr(self.gpack, bo[bn % (4 * i + 2)]), st2)
                if overlap:
                    ev_w[k] = torch.cuda.Event()
                    ev_w[k].record(side)
            # dx_i[t] = W1^T dfg[t] + W0^T dfg[t+d] + dy[t]        on [off_i, T)
            # (writes dX[i%2], which the PREVIOUS layer's weight gradients may still be reading)
            if overlap and ev_prev is not None:
                main.wait_event(ev_prev)
            call("wn_chan_gemm", dfg, dfg, 2 * CH * pitch, pitch, t_lo, pitch, 0, d, 2 * CH // 32, 2 * CH // 32, br("fgT%d" % i),
                 CH // 16, self.R, ptr(bw["dX"][i % 2], SLACK), xb, pitch, 0, None,
                 dy, xb, pitch, t_lo, None, 0, 0, self.off[i], T, 0, B, mb, st)
            ev_prev = ev_w[k]
        if overlap:
            for e in ev_w:
                if e is not None:
                    main.wait_event(e)
            ev_join = torch.cuda.Event()          # everything on the side stream (epilogue weight gradients)
            ev_join.record(side)
            main.wait_event(ev_join)
        self.mark("stack_bwd")
        # causal weight gradient: dWc[r][q][tap] = sum dx0[r][t] in[q][t-1+tap]
        x = ws["x_in"]
        dx0 = ptr(bw["dX"][0], SLACK)
        desc = bw["slab_desc"]
        # the input (and the codes it was built from) must still be what the forward saw: in-place writes that bump the
        # version counter are caught here (autograd's own rule for saved tensors); writes that do not (x.data.zero_(), a
        # raw-pointer kernel) cannot be - onehot() documents the tensor as immutable while tagged
        xv, cv = ws.get("x_ver", (None, None))
        if x is not None and xv is not None and x._version != xv:
            raise RuntimeError("music_amd: the input of this forward was modified in place before backward()")
        x_codes = ws.get("x_codes")
        if x_codes is not None and cv is not None and x_codes[0]._version != cv:
            if x is None:
                raise RuntimeError("music_amd: the integer codes of this forward were modified in place before backward()")
            x_codes = None                  # the dense tensor is intact: the dense weight-gradient product
        if x_codes is not None:
            codes, scrambled = x_codes
            call("wn_causal_wgrad_codes", ptr(codes), 1 if scrambled else 0, dx0, None, 0, 0, xb, pitch, CH, Q, T, B,
                 ptr(bw["slab"], plan["causal_codes"][0]), st)
            desc = bw["slab_desc_codes"]
        else:
            wgrad("causal", dx0, xb, pitch, 0, pitch, ptr(x), ptr(x), Q * T, T, -1, 0, T, Q // 16, CH // 16, 0, 2 * Q, 1, T)
        if self.use_bias:
            call("wn_bias_grad", dx0, xb, pitch, 0, self.R, 1, T, B, ptr(self.gpack, self.gp_bias_off["causal_layer.bias"]), st)
        self.mark("causal_bwd")
        call("wn_reduce_slabs", ptr(desc), bw["slab_nops"], bw["slab_vec"], ptr(bw["slab"]), ptr(self.gpack), st)
        if bw["pair"]:
            call("wn_gather_grads2", ptr(self.gpack), ptr(self.gidx_pa), ptr(self.gidx_pb), ptr(self.flat_grad), self.spec.total, st)
        else:
            call("wn_gather_grads", ptr(self.gpack), ptr(self.gidx), ptr(self.flat_grad), self.spec.total, st)
        self.mark("slab_reduce")

    def input_grad(self, ws):
        """Gradient of the last backward w.r.t. the module's INPUT (what autograd gives the reference when the input requires grad:
        the causal nn.Conv1d's data gradient, wavenet/model.py:104):  din[q][s] = sum_r W[r][q][1] dx0[r][s] + W[r][q][0] dx0[r][s + 1],
        dx0 living on [1, T).  One channel product on the first block's data gradient, which the backward leaves in its workspace."""
        bw = ws["bwd"]
        if bw is None:
            raise RuntimeError("music_amd: input_grad() needs the backward of this forward to have run")
        B, T, pitch, CH, Q = ws["B"], ws["T"], ws["pitch"], self.CH, self.Q
        din = torch.empty(B, Q, T, dtype=torch.float32, device=self.device)
        dx0 = ptr(bw["dX"][0], SLACK)
        call("wn_chan_gemm", dx0, dx0, CH * pitch, pitch, 1, T, 0, 1, CH // 32, CH // 32, ptr(self.pk_b, self.pk_b_off["causalT"]), Q // 16, Q,
             ptr(din), Q * T, T, 0, None, None, 0, 0, 0, None, 0, 0, 0, T, 0, B, self.mode_bwd, _lib.stream())
        return din

    def backward(self, ws, dprobs):
        """dprobs: (B*W, Q) gradient w.r.t. the probabilities returned by forward()."""
        bw = self._bwd_workspace(ws)
        dprobs = dprobs.contiguous()
        call("wn_chunk_softmax256_bwd", ptr(ws["probs"]), ptr(dprobs), ptr(bw["dO"]), ws["B"] * ws["W"], _lib.stream())
        self.backward_from_dlogits(ws)

    # ------------------------------------------------------------------ fused training step
    def loss_and_grad_codes(self, codes, target, scrambled=True, want_probs=False):
        """loss_and_grad on the integer codes themselves (int32 (B,T) on the device; `scrambled` = the loader's one-hot
        layout, faster_audio_data.py:77-81): same result as loss_and_grad(self.onehot(codes, scrambled), target), but the
        (B,256,T) float tensor is never built, written or read - the causal layer is a gather forward and
        a scatter backward."""
        return self.loss_and_grad(None, target, want_probs, codes=(codes, scrambled))

    def loss_and_grad(self, x, target, want_probs=False, codes=None):
        """forward + CrossEntropyLoss(probs, target) + backward (wavenet/train.py:178-181).
        Returns the loss as a 0-d device tensor; gradients land in self.flat_grad."""
        self._throttle.enter()                 # at most WN_MAX_STEPS_IN_FLIGHT fused steps in flight (_lib.StepThrottle)
        self.mark("begin")
        self.pack_weights()
        self.mark("pack")
        ws = self.forward_logits(x, codes=codes)
        bw = self._bwd_workspace(ws)
        B, W = ws["B"], ws["W"]
        n = B * W
        target = target.reshape(-1)
        assert target.numel() == n and target.dtype == torch.int64 and target.is_cuda
        if "loss_part" not in ws:
            ws["loss_part"] = torch.zeros(_lib.CE_NUM_PARTIALS, dtype=torch.float32, device=self.device)
        probs = None
        if want_probs:
            probs = torch.empty(n, self.Q, dtype=torch.float32, device=self.device)
            ws["probs"] = probs
        call("wn_chunk_softmax256_ce", ptr(ws["O"]), ptr(target), ptr(probs), ptr(bw["dO"]), ptr(ws["loss_part"]),
             n, 1.0 / n, _lib.stream())
        self.mark("softmax_ce")
        self.backward_from_dlogits(ws)
        loss = ws["loss_part"].sum()
        self._throttle.leave()
        return loss

    def adam_init(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.adam_state = dict(m=torch.zeros_like(self.flat), v=torch.zeros_like(self.flat), t=0,
                               lr=lr, b1=betas[0], b2=betas[1], eps=eps)

    def adam_step(self, gscale=1.0):
        s = self.adam_state
        s["t"] += 1
        bc1 = 1.0 - s["b1"] ** s["t"]
        bc2 = 1.0 - s["b2"] ** s["t"]
        call("wn_adam_flat", ptr(self.flat), ptr(self.flat_grad), ptr(s["m"]), ptr(s["v"]), self.spec.total,
             s["lr"], s["b1"], s["b2"], s["eps"], bc1, bc2, gscale, _lib.stream())
        self.mark("adam")

    def onehot(self, codes, scrambled=True):
        """int32 (B,T) codes on the device -> float32 (B,Q,T) (faster_audio_data.py:62-83).
        The result carries its codes (`_wn_codes`), and a forward on it runs the causal layer from the codes instead of
        streaming the 131 MB dense tensor.  CONTRACT: while tagged, the tensor and the codes are immutable - an in-place
        op that bumps the version counter drops the fast path (forward) or is reported (backward); a write that does
        not (`.data`, raw pointers, `out=` on the storage) is undetectable and would make the causal layer compute on
        the original codes.  A copy (`.to()`, `.contiguous()` of a non-contiguous view, `.clone()`) loses the tag and
        silently takes the dense path (+0.08 ms per step at config 2)."""
        B, T = codes.shape
        out = torch.empty(B, self.Q, T, dtype=torch.float32, device=self.device)
        call("wn_onehot", ptr(codes), ptr(out), B, self.Q, T, 1 if scrambled else 0, _lib.stream())
        self.mark("onehot")
        if codes.is_contiguous() and codes.dtype == torch.int32:
            out._wn_codes = (codes, bool(scrambled), out._version, codes._version)      # see forward_logits
        return out
