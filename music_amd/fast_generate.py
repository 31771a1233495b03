"""Counterpart of the reference's ``wavenet/fast_generate.py`` (cached-queue autoregressive decode).

``predict_next(net, note, state_queue=None)`` keeps the reference's contract
(wavenet/fast_generate.py:13-141): the first call takes a ``(1, Q, receptive_field)`` one-hot piece,
runs the full forward (HIP path) and snapshots the per-layer queues; later calls take one
``(1, Q, 1)`` column and advance every queue by one sample.  It returns ``(LongTensor(1,), queue)``
where ``queue`` behaves like the reference's ``OrderedDict`` ('causal_layer' -> (1,Q,1),
'block_i' -> (1,R,d_i), oldest column first).

As WRITTEN in the reference each block pushes its OUTPUT into its own queue instead of its input
(fast_generate.py:128-129), so fast generation differs from naive generation; that
recurrence is reproduced by default.  ``correct_queue=True`` selects the fast-wavenet recurrence.

``generate()`` (fast_generate.py:144-179) runs the whole greedy loop as ONE persistent kernel launch
(``wn_decode``) instead of one Python iteration per sample, and writes the wav with scipy (librosa,
which the reference uses for that, is not a dependency here).  Unlike the reference module this one
has no import-time side effect (the reference calls ``generate(...)`` at import, :182-186).
"""
from collections import OrderedDict
import ctypes
import json
import os

import numpy as np
import torch

try:
    from . import _lib
    from ._lib import call, ptr
    from .audio_func import mu_law_decode
    from .engine import SLACK, pack_index
    from .model import wavenet
    from .train import load_model
except ImportError:
    from music_amd import _lib
    from music_amd._lib import call, ptr
    from music_amd.audio_func import mu_law_decode
    from music_amd.engine import SLACK, pack_index
    from music_amd.model import wavenet
    from music_amd.train import load_model


def _mfma_decode(eng):
    """True when the matrix-core decode kernels serve this model (wn_decode_batch_pk): up to 64 residual / dilation channels -
    fewer are PADDED to 64 with zero rows and columns (the decoder is bound by latency, not traffic: the padding is free) -,
    256 or 512 skip channels, 256 quantisation channels, f16x3 forward mode; biases are fine.  WN_DEC_MFMA=0: never.
    Decided ONCE per engine (the queue column width of every DecodeState and the layout of the weight pack follow from it:
    a switch flipped inside a process must not make them disagree)."""
    v = getattr(eng, "_dec_mfma", None)
    if v is None:
        v = eng._dec_mfma = bool(os.environ.get("WN_DEC_MFMA", "1") == "1" and getattr(eng, "k", 2) == 2 and eng.R <= 64 and
                                 eng.D <= 64 and eng.S in (256, 512) and eng.Q == 256 and eng.mode_fwd == _lib.F16X3 and
                                 "fg0" in getattr(eng, "pk_f_off", {}))
    return v


def _ring_width(eng):
    """Floats per queue column: the (padded) residual channel count the decode kernels run with."""
    return 64 if _mfma_decode(eng) else eng.R


class _DecodePack:
    """fp32 weights in the layout wn_decode reads (rebuilt from the engine's flat buffer), and for the matrix-core kernels
    the same weights as packed f16 hi/lo fragments.  With fewer than 64 residual / dilation channels on that path every
    matrix is laid out for Rp = Dp = 64 channels with structural zeros (index -1)."""

    def __init__(self, eng):
        sp, R, D, S, Q, N = eng.spec, eng.R, eng.D, eng.S, eng.Q, eng.N
        if getattr(eng, "k", 2) != 2:
            # the reference's own cached-queue decoder keeps d_i columns per block and convolves [queue | note] once
            # (wavenet/fast_generate.py:73-90): it only exists for filter_width 2
            raise NotImplementedError("fast_generate implements the reference's cached-queue recurrence, which is defined for "
                                      "filter_width == 2 only (wavenet/fast_generate.py:73-90)")
        self.mfma = _mfma_decode(eng)
        Rp = Dp = self.Rp = self.Dp = 64 if self.mfma else None
        if not self.mfma:
            Rp, Dp = self.Rp, self.Dp = R, D

        def pad(m, rows, cols):
            out = np.full((rows, cols), -1, dtype=np.int64)
            out[:m.shape[0], :m.shape[1]] = m
            return out

        def padv(v, n):
            out = np.full(n, -1, dtype=np.int64)
            out[:len(v)] = v
            return out
        parts, mats = [], {}
        wc = sp.conv("causal_layer.weight")                                 # [R,Q,2]
        self.o_causal = 0
        parts.append(pad(np.concatenate([wc[:, :, 0], wc[:, :, 1]], 1), Rp, 2 * Q).reshape(-1))
        self.layer_stride = 2 * Dp * 2 * Rp + Rp * Dp + S * Dp
        self.o_layers = sum(len(p) for p in parts)
        for i in range(N):
            wf = sp.conv("dilation_layer_stack.%d.weight" % (4 * i))        # [D,R,2]
            wg = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 1))
            wd = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 2))[:, :, 0]
            ws = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 3))[:, :, 0]
            blk = lambda a, b: np.concatenate([pad(a, Dp, Rp), pad(b, Dp, Rp)], 1)
            fg = np.concatenate([blk(wf[:, :, 1], wf[:, :, 0]),                      # k = [tap1 (cur) | tap0 (old)]
                                 blk(wg[:, :, 1], wg[:, :, 0])], 0)
            parts += [fg.reshape(-1), pad(wd, Rp, Dp).reshape(-1), pad(ws, S, Dp).reshape(-1)]
            # the packed forms: [f; g] rows, K = [tap0 | tap1] in natural order; dense in chained order (as wn_resblock_fwd)
            mats["fg%d" % i] = (np.concatenate([blk(wf[:, :, 0], wf[:, :, 1]), blk(wg[:, :, 0], wg[:, :, 1])], 0), False)
            mats["d%d" % i] = (pad(wd, Rp, Dp), True)
            mats.setdefault("skip_cols", []).append(pad(ws, S, Dp))
        self.o_p1 = sum(len(p) for p in parts)
        parts.append(sp.conv("post_process_1.weight")[:, :, 0].reshape(-1))
        self.o_p2 = sum(len(p) for p in parts)
        parts.append(sp.conv("post_process_2.weight")[:, :, 0].reshape(-1))
        self.o_bias = None
        if eng.use_bias:
            self.o_bias = sum(len(p) for p in parts)
            b = lambda n: sp.off[n] + np.arange(sp.shape[n][0])
            self.ob_causal = self.o_bias
            parts.append(padv(b("causal_layer.bias"), Rp))
            self.ob_layers = sum(len(p) for p in parts)
            for i in range(N):
                parts += [padv(b("dilation_layer_stack.%d.bias" % (4 * i + k)), n) for k, n in enumerate((Dp, Dp, Rp, S))]
            self.ob_p1 = sum(len(p) for p in parts)
            parts.append(b("post_process_1.bias"))
            self.ob_p2 = sum(len(p) for p in parts)
            parts.append(b("post_process_2.bias"))
        idx = np.concatenate(parts).astype(np.int32)
        self.idx = torch.from_numpy(idx).to(eng.device)
        self.buf = torch.empty(len(idx), dtype=torch.float32, device=eng.device)
        self.eng = eng
        self.pk = None
        if self.mfma:
            # packed fragments: per block "fg" then "d" at a fixed stride, then skip, p1, p2 (offsets in halfs, hi + lo planes)
            lst = []
            for i in range(N):
                lst += [("fg%d" % i,) + mats["fg%d" % i], ("d%d" % i,) + mats["d%d" % i]]
            lst += [("skip", np.concatenate(mats["skip_cols"], 1), False),
                    ("p1", sp.conv("post_process_1.weight")[:, :, 0], False), ("p2", sp.conv("post_process_2.weight")[:, :, 0], False)]
            offs, o, pidx = {}, 0, []
            for name, m, chained in lst:
                offs[name] = 2 * o
                pi = pack_index(m.astype(np.int64), chained)
                pidx.append(pi)
                o += len(pi)
            self.pk_off = offs
            self.pk_stride = offs["fg1"] - offs["fg0"] if N > 1 else 0
            self.pk_idx = torch.from_numpy(np.concatenate(pidx).astype(np.int32)).to(eng.device)
            self.pk = torch.zeros(2 * o, dtype=torch.int16, device=eng.device)

    def refresh(self):
        call("wn_gather_grads", ptr(self.eng.flat), ptr(self.idx), ptr(self.buf), self.idx.numel(), _lib.stream())
        if self.pk is not None:
            call("wn_pack_weights", ptr(self.eng.flat), ptr(self.pk_idx), ptr(self.pk), self.pk_idx.numel(), _lib.F16X3, _lib.stream())

    def p(self, off):
        return None if off is None else ptr(self.buf, off)

    def chain(self):
        """(pk pointer, fg offset, d offset, per-block stride, skip, p1, p2 offsets) for wn_decode_batch_pk, or Nones / -1."""
        if self.pk is None:
            return None, 0, 0, 0, -1, -1, -1
        o = self.pk_off
        return ptr(self.pk), o["fg0"], o["d0"], self.pk_stride, o["skip"], o["p1"], o["p2"]


class DecodeState(OrderedDict):
    """The per-layer FIFO queues of the decoder.  Internally every block's queue is a ring buffer in
    time-major layout on the device; indexing by the reference's keys materialises the
    time-ordered ``(1, C, d)`` tensor the reference would hold."""

    def __init__(self, eng, rings, prev, steps=0):
        super().__init__()
        self.eng, self.rings, self.prev, self.steps = eng, rings, prev, steps
        self.rw = _ring_width(eng)                # floats per queue column (64 on the matrix-core kernels, zeros beyond eng.R)
        self.q_off = np.cumsum([0] + [d * self.rw for d in eng.dil[:-1]]).astype(np.int64)
        for k in ["causal_layer"] + ["block_%d" % (i + 1) for i in range(eng.N)]:
            OrderedDict.__setitem__(self, k, None)

    def _ring(self, i):
        d = self.eng.dil[i]
        return self.rings[self.q_off[i]:self.q_off[i] + d * self.rw].view(d, self.rw)[:, :self.eng.R]

    def __getitem__(self, key):
        if key == "causal_layer":
            return self.prev.view(1, -1, 1).clone()
        i = int(key.split("_")[1]) - 1
        d = self.eng.dil[i]
        ring = torch.roll(self._ring(i), -(self.steps % d), 0)        # oldest column first
        return ring.t().contiguous().view(1, self.eng.R, d)

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    @staticmethod
    def from_tensors(eng, queue):
        """Build the ring form from reference-style tensors (time-ordered, oldest first)."""
        rw = _ring_width(eng)
        rings = torch.cat([torch.nn.functional.pad(queue["block_%d" % (i + 1)].to(eng.device).float().reshape(eng.R, eng.dil[i]).t(),
                                                   (0, rw - eng.R)).reshape(-1) for i in range(eng.N)])
        prev = queue["causal_layer"].to(eng.device).float().reshape(-1).clone()
        return DecodeState(eng, rings.contiguous(), prev, 0)


def _decode(net, state, note0, n_steps, forced=None, want_probs=False, correct_queue=False, temperature=None, seed=0):
    eng = state.eng
    pack = getattr(net, "_decode_pack", None)
    if pack is None or pack.eng is not eng:
        pack = net._decode_pack = _DecodePack(eng)
    if state.rw != pack.Rp:
        raise RuntimeError("music_amd.fast_generate: this DecodeState holds %d floats per queue column, the weight pack is laid "
                           "out for %d (built for another engine or decode kernel form)" % (state.rw, pack.Rp))
    pack.refresh()
    dev = eng.device
    codes = torch.empty(n_steps, dtype=torch.int32, device=dev)
    probs = torch.empty(n_steps, eng.Q, dtype=torch.float32, device=dev) if want_probs else None
    note_out = torch.empty(eng.Q, dtype=torch.float32, device=dev)
    prev_out = torch.empty(eng.Q, dtype=torch.float32, device=dev)
    dil = (ctypes.c_int32 * eng.N)(*eng.dil)
    qoff = (ctypes.c_int64 * eng.N)(*[int(v) for v in state.q_off])
    forced_t = forced.to(device=dev, dtype=torch.int32).contiguous() if forced is not None else None
    sync = getattr(net, "_decode_sync", None)
    n_sync = _lib.decode_sync_granules(eng.N, pack.Dp, eng.S)
    if sync is None or sync.device != dev or sync.numel() != n_sync:
        sync = net._decode_sync = torch.zeros(n_sync, dtype=torch.int64, device=dev)
    bias = pack.o_bias is not None
    pk = pack.chain()
    call("wn_decode_batch_pk", eng.N, pack.Rp, pack.Dp, eng.S, eng.Q, ctypes.cast(dil, ctypes.c_void_p), ctypes.cast(qoff, ctypes.c_void_p),
         ptr(state.rings), pack.p(pack.o_causal), pack.p(pack.ob_causal) if bias else None,
         pack.p(pack.o_layers), pack.layer_stride, pack.p(pack.ob_layers) if bias else None,
         pack.p(pack.o_p1), pack.p(pack.ob_p1) if bias else None, pack.p(pack.o_p2), pack.p(pack.ob_p2) if bias else None,
         ptr(note0), ptr(state.prev), ptr(note_out), ptr(prev_out), ptr(forced_t), ptr(codes), ptr(probs),
         state.steps, n_steps, 1 if correct_queue else 0, ptr(sync), 1, 0,
         float(temperature) if temperature else 0.0, int(seed), pk[0], pk[1], pk[2], pk[3], pk[4], pk[5], pk[6], _lib.stream())
    if n_steps >= 4 and int(sync[-1].item()) != 0:
        raise _lib.WavenetHipError("wn_decode: a hand-off between the two decode workgroups timed out")
    state.prev = prev_out
    state.steps += n_steps
    return codes, probs, note_out


def predict_next(net, note, state_queue=None, correct_queue=False):
    """wavenet/fast_generate.py:13-141."""
    if state_queue is None:
        assert note.size()[2] == net.receptive_field
        dev = note.device if note.is_cuda else torch.device("cuda")
        x = note.detach().to(dev).float().contiguous()
        with torch.no_grad():
            probs = net(x)                                        # (1, Q): W == 1
        eng = net._engine
        ws = eng.workspace(1, x.size(2))
        T, pitch, CH, R = x.size(2), ws["pitch"], eng.CH, eng.R
        X = ws["X"][SLACK:SLACK + (eng.N + 1) * CH * pitch].view(eng.N + 1, CH, pitch)
        # queue of block i = the last d_i columns of that block's INPUT (fast_generate.py:42-47)
        rw = _ring_width(eng)
        rings = torch.cat([torch.nn.functional.pad(X[i, :R, T - d:T].t(), (0, rw - R)).reshape(-1)
                           for i, d in enumerate(eng.dil)]).contiguous()
        state = DecodeState(eng, rings, x[0, :, -1].clone(), 0)
        _, predict = torch.topk(probs.view(-1), 1)
        return predict.to(note.device), state
    assert note.size()[2] == 1
    eng = net._engine_for(note.device if note.is_cuda else torch.device("cuda"))
    if not isinstance(state_queue, DecodeState):
        state_queue = DecodeState.from_tensors(eng, state_queue)
    note0 = note.detach().to(eng.device).float().reshape(-1).contiguous()
    codes, _, _ = _decode(net, state_queue, note0, 1, correct_queue=correct_queue)
    return codes.to(torch.int64).to(note.device), state_queue


def generate_codes(net, start_piece, note_num, correct_queue=False, temperature=None, seed=0):
    """The greedy loop of fast_generate.py:162-172 as one init forward + ONE persistent launch.
    Returns the note_num generated codes (int64, on the device).
    ``temperature`` (the reference is greedy only): if given and > 0, every code after the
    first is SAMPLED from softmax(logits / temperature), reproducibly for a given ``seed``."""
    with torch.no_grad():
        first, state = predict_next(net, start_piece, None)
    if note_num <= 1:
        return first.to(state.eng.device)[:note_num]
    note0 = torch.zeros(net.quantization_channels, dtype=torch.float32, device=state.eng.device)
    note0[int(first[0])] = 1.0
    codes, _, _ = _decode(net, state, note0, note_num - 1, correct_queue=correct_queue, temperature=temperature, seed=seed)
    return torch.cat([first.to(codes.device).to(torch.int64), codes.to(torch.int64)])


def generate_codes_batch(net, start_pieces, note_num, correct_queue=False, temperature=None, seed=0):
    """SURVEY 8f2 (batched utterances; the reference generates one at a time): greedy generation of
    ``note_num`` codes for U independent start pieces ``(U, Q, receptive_field)`` in ONE persistent
    launch - on the matrix-core path eight utterances to a workgroup pair, one pair of MFMA result columns each (else one
    pair per utterance), the weights are shared.  Returns int64
    ``(U, note_num)`` on the device; row u equals ``generate_codes(net, start_pieces[u:u+1], note_num)``."""
    assert start_pieces.dim() == 3 and start_pieces.size(2) == net.receptive_field
    U = start_pieces.size(0)
    if U > 1024:
        raise ValueError("at most 1024 utterances per launch (128 off the matrix-core path)")
    dev = start_pieces.device if start_pieces.is_cuda else torch.device("cuda")
    x = start_pieces.detach().to(dev).float().contiguous()
    with torch.no_grad():
        probs = net(x)                                            # (U, Q): W == 1
    eng = net._engine
    ws = eng.workspace(U, x.size(2))
    T, pitch, CH, R, N, Q = x.size(2), ws["pitch"], eng.CH, eng.R, eng.N, eng.Q
    X = ws["X"][SLACK:SLACK + (N + 1) * U * CH * pitch].view(N + 1, U, CH, pitch)
    # ring of block i of utterance u = the last d_i columns of that block's input, time-major
    rw = _ring_width(eng)
    rings = torch.cat([torch.nn.functional.pad(X[i, :, :R, T - d:T].transpose(1, 2), (0, rw - R)).reshape(U, d * rw)
                       for i, d in enumerate(eng.dil)], 1).contiguous()
    first = probs.view(U, Q).argmax(1)
    if note_num <= 1:
        return first.view(U, 1)[:, :note_num]
    pack = getattr(net, "_decode_pack", None)
    if pack is None or pack.eng is not eng:
        pack = net._decode_pack = _DecodePack(eng)
    pack.refresh()
    n_steps = note_num - 1
    note0 = torch.zeros(U, Q, dtype=torch.float32, device=dev)
    note0[torch.arange(U, device=dev), first] = 1.0
    prev0 = x[:, :, -1].contiguous()
    codes = torch.empty(U, n_steps, dtype=torch.int32, device=dev)
    note_out = torch.empty(U, Q, dtype=torch.float32, device=dev)
    prev_out = torch.empty(U, Q, dtype=torch.float32, device=dev)
    sync = torch.zeros(U * _lib.decode_sync_granules(N, pack.Dp, eng.S), dtype=torch.int64, device=dev)
    dil = (ctypes.c_int32 * N)(*eng.dil)
    q_off = np.cumsum([0] + [d * rw for d in eng.dil[:-1]]).astype(np.int64)
    qoff = (ctypes.c_int64 * N)(*[int(v) for v in q_off])
    bias = pack.o_bias is not None
    pk = pack.chain()
    call("wn_decode_batch_pk", N, pack.Rp, pack.Dp, eng.S, Q, ctypes.cast(dil, ctypes.c_void_p), ctypes.cast(qoff, ctypes.c_void_p),
         ptr(rings), pack.p(pack.o_causal), pack.p(pack.ob_causal) if bias else None,
         pack.p(pack.o_layers), pack.layer_stride, pack.p(pack.ob_layers) if bias else None,
         pack.p(pack.o_p1), pack.p(pack.ob_p1) if bias else None, pack.p(pack.o_p2), pack.p(pack.ob_p2) if bias else None,
         ptr(note0), ptr(prev0), ptr(note_out), ptr(prev_out), None, ptr(codes), None,
         0, n_steps, 1 if correct_queue else 0, ptr(sync), U, rings.size(1),
         float(temperature) if temperature else 0.0, int(seed), pk[0], pk[1], pk[2], pk[3], pk[4], pk[5], pk[6], _lib.stream())
    if n_steps >= 4:
        flags = sync.view(U, -1)[:, -1]
        if int(flags.abs().max().item()) != 0:
            raise _lib.WavenetHipError("wn_decode_batch: a hand-off between two decode workgroups timed out")
    return torch.cat([first.view(U, 1).to(torch.int64), codes.to(torch.int64)], 1)


def generate(model_path, model_name, generate_path, generate_name, start_piece=None, sr=16000, duration=10):
    """wavenet/fast_generate.py:144-179."""
    if os.path.exists(generate_path) is False:
        os.makedirs(generate_path)
    with open('./params/wavenet_params.json', 'r') as f:
        params = json.load(f)
    net = wavenet(**params)
    net = load_model(net, model_path, model_name)
    if net is None:
        raise FileNotFoundError(model_path + model_name)
    net = net.cuda()
    if start_piece is None:
        start_piece = torch.zeros(1, 256, net.receptive_field)
        start_piece[:, 128, :] = 1.0
    generated_piece = generate_codes(net, start_piece.cuda(), duration * sr)
    print(generated_piece.tolist()[:32], "...")
    audio = mu_law_decode(generated_piece, net.quantization_channels).cpu().numpy()
    from scipy.io import wavfile
    wavfile.write(generate_path + generate_name, sr, audio.astype(np.float32))
    return generated_piece
