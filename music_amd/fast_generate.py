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
    from .engine import SLACK
    from .model import wavenet
    from .train import load_model
except ImportError:
    from music_amd import _lib
    from music_amd._lib import call, ptr
    from music_amd.audio_func import mu_law_decode
    from music_amd.engine import SLACK
    from music_amd.model import wavenet
    from music_amd.train import load_model


class _DecodePack:
    """fp32 weights in the layout wn_decode reads (rebuilt from the engine's flat buffer)."""

    def __init__(self, eng):
        sp, R, D, S, Q, N = eng.spec, eng.R, eng.D, eng.S, eng.Q, eng.N
        if getattr(eng, "k", 2) != 2:
            # the reference's own cached-queue decoder keeps d_i columns per block and convolves [queue | note] once
            # (wavenet/fast_generate.py:73-90): it only exists for filter_width 2
            raise NotImplementedError("fast_generate implements the reference's cached-queue recurrence, which is defined for "
                                      "filter_width == 2 only (wavenet/fast_generate.py:73-90)")
        parts = []
        wc = sp.conv("causal_layer.weight")                                 # [R,Q,2]
        self.o_causal = 0
        parts.append(np.concatenate([wc[:, :, 0], wc[:, :, 1]], 1).reshape(-1))
        self.layer_stride = 2 * D * 2 * R + R * D + S * D
        self.o_layers = sum(len(p) for p in parts)
        for i in range(N):
            wf = sp.conv("dilation_layer_stack.%d.weight" % (4 * i))        # [D,R,2]
            wg = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 1))
            wd = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 2))[:, :, 0]
            ws = sp.conv("dilation_layer_stack.%d.weight" % (4 * i + 3))[:, :, 0]
            fg = np.concatenate([np.concatenate([wf[:, :, 1], wf[:, :, 0]], 1),      # k = [tap1 (cur) | tap0 (old)]
                                 np.concatenate([wg[:, :, 1], wg[:, :, 0]], 1)], 0)
            parts += [fg.reshape(-1), wd.reshape(-1), ws.reshape(-1)]
        self.o_p1 = sum(len(p) for p in parts)
        parts.append(sp.conv("post_process_1.weight")[:, :, 0].reshape(-1))
        self.o_p2 = sum(len(p) for p in parts)
        parts.append(sp.conv("post_process_2.weight")[:, :, 0].reshape(-1))
        self.o_bias = None
        if eng.use_bias:
            self.o_bias = sum(len(p) for p in parts)
            b = lambda n: sp.off[n] + np.arange(sp.shape[n][0])
            self.ob_causal = self.o_bias
            parts.append(b("causal_layer.bias"))
            self.ob_layers = sum(len(p) for p in parts)
            for i in range(N):
                parts += [b("dilation_layer_stack.%d.bias" % (4 * i + k)) for k in range(4)]
            self.ob_p1 = sum(len(p) for p in parts)
            parts.append(b("post_process_1.bias"))
            self.ob_p2 = sum(len(p) for p in parts)
            parts.append(b("post_process_2.bias"))
        idx = np.concatenate(parts).astype(np.int32)
        self.idx = torch.from_numpy(idx).to(eng.device)
        self.buf = torch.empty(len(idx), dtype=torch.float32, device=eng.device)
        self.eng = eng

    def refresh(self):
        call("wn_gather_grads", ptr(self.eng.flat), ptr(self.idx), ptr(self.buf), self.idx.numel(), _lib.stream())

    def p(self, off):
        return None if off is None else ptr(self.buf, off)


class DecodeState(OrderedDict):
    """The per-layer FIFO queues of the decoder.  Internally every block's queue is a ring buffer in
    time-major layout on the device; indexing by the reference's keys materialises the
    time-ordered ``(1, C, d)`` tensor the reference would hold."""

    def __init__(self, eng, rings, prev, steps=0):
        super().__init__()
        self.eng, self.rings, self.prev, self.steps = eng, rings, prev, steps
        self.q_off = np.cumsum([0] + [d * eng.R for d in eng.dil[:-1]]).astype(np.int64)
        for k in ["causal_layer"] + ["block_%d" % (i + 1) for i in range(eng.N)]:
            OrderedDict.__setitem__(self, k, None)

    def _ring(self, i):
        d, R = self.eng.dil[i], self.eng.R
        return self.rings[self.q_off[i]:self.q_off[i] + d * R].view(d, R)

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
        rings = torch.cat([queue["block_%d" % (i + 1)].to(eng.device).float().reshape(eng.R, eng.dil[i]).t().reshape(-1)
                           for i in range(eng.N)])
        prev = queue["causal_layer"].to(eng.device).float().reshape(-1).clone()
        return DecodeState(eng, rings.contiguous(), prev, 0)


def _packed_chain(eng):
    """(pk pointer, fg offset, d offset, per-block stride) of the training engine's packed f16x3 forward weights for the
    matrix-core form of the decode chain (wn_decode_batch_pk), or (None, 0, 0, 0) when it does not apply (64 residual /
    dilation channels, f16x3 forward mode; biases are fine) or WN_DEC_MFMA=0; plus the fragment bases of the skip and
    post-processing products (256 skip / quantisation channels), or -1."""
    import os
    if (os.environ.get("WN_DEC_MFMA", "1") != "1" or eng.R != 64 or eng.D != 64 or eng.mode_fwd != _lib.F16X3 or
            "fg0" not in eng.pk_f_off):        # (the general plan's packs have another fragment order: fp32 decode kernel)
        return None, 0, 0, 0, -1, -1, -1
    off = eng.pk_f_off
    fg0, d0 = off["fg0"], off["d0"]
    stride = off["fg1"] - fg0 if eng.N > 1 else 0
    for i in range(eng.N):
        if off["fg%d" % i] != fg0 + i * stride or off["d%d" % i] != d0 + i * stride:
            return None, 0, 0, 0, -1, -1, -1
    eng.pack_weights()
    post = (off["skip"], off["p1"], off["p2"]) if (eng.S == 256 and eng.Q == 256) else (-1, -1, -1)
    return (ptr(eng.pk_f), fg0, d0, stride) + post


def _decode(net, state, note0, n_steps, forced=None, want_probs=False, correct_queue=False, temperature=None, seed=0):
    eng = state.eng
    pack = getattr(net, "_decode_pack", None)
    if pack is None or pack.eng is not eng:
        pack = net._decode_pack = _DecodePack(eng)
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
    n_sync = _lib.decode_sync_granules(eng.N, eng.D, eng.S)
    if sync is None or sync.device != dev or sync.numel() != n_sync:
        sync = net._decode_sync = torch.zeros(n_sync, dtype=torch.int64, device=dev)
    bias = pack.o_bias is not None
    pk = _packed_chain(eng)
    call("wn_decode_batch_pk", eng.N, eng.R, eng.D, eng.S, eng.Q, ctypes.cast(dil, ctypes.c_void_p), ctypes.cast(qoff, ctypes.c_void_p),
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
        rings = torch.cat([X[i, :R, T - d:T].t().reshape(-1) for i, d in enumerate(eng.dil)]).contiguous()
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
    rings = torch.cat([X[i, :, :R, T - d:T].transpose(1, 2).reshape(U, d * R) for i, d in enumerate(eng.dil)], 1).contiguous()
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
    sync = torch.zeros(U * _lib.decode_sync_granules(N, eng.D, eng.S), dtype=torch.int64, device=dev)
    dil = (ctypes.c_int32 * N)(*eng.dil)
    q_off = np.cumsum([0] + [d * R for d in eng.dil[:-1]]).astype(np.int64)
    qoff = (ctypes.c_int64 * N)(*[int(v) for v in q_off])
    bias = pack.o_bias is not None
    pk = _packed_chain(eng)
    call("wn_decode_batch_pk", N, R, eng.D, eng.S, Q, ctypes.cast(dil, ctypes.c_void_p), ctypes.cast(qoff, ctypes.c_void_p),
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
