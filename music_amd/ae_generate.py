"""Counterpart of the reference's ``wavenet_autoencoder/generate.py``: naive generation by a full
forward per generated sample (generate.py:13-65).  The reference version cannot run as shipped (``librosa`` used
without import :65, the start tensor instead of the predicted int appended :48, ``.cuda()`` hard-wired :53); the
algorithm is kept, the output is written with scipy.  It is a harness, not a kernel.

As written, the window update ``input_wav[:,-net.receptive_field-511:]`` (generate.py:55) slices dimension 1 - the
256 CHANNELS - so it is a no-op and the window GROWS by one sample per step (the pooled encoding gains a frame
every ``pool`` steps and the ``_conditon`` branches change with it; cost O(T^2)).  ``generate`` reproduces that by
default, like ``fast_generate`` reproduces the as-written queue recurrence; ``sliding_window=True`` is the evident
intent (the last ``receptive_field + 511`` samples along TIME plus the new one, O(T * rf)) and NOT the reference's
behaviour.  Both are pinned by tests/golden/g9_ae_harness.json (outputs of the reference's own ``predict_next``).

``generate_cached`` / ``cached_decoder`` are the SURVEY 8f3 replacement: encoder ONCE on the start window,
conditioning projections drawn ONCE, then the persistent cached-queue decode kernel (wn_decode).  That is a
different algorithm from the reference's, not a faster form of it: the reference re-encodes the sliding window
and re-draws its 31 random projections for every generated sample (generate.py:13-19, model1.py:178-217).
With a time-constant encoding (one pooled frame: the reference's window of receptive_field + 512 samples and
pool 512 give exactly one) the conditioning is a constant per-channel bias, so the conditioned decoder IS a
plain WaveNet with effective biases, and the existing decode kernel runs it unchanged."""
import json
import os

import numpy as np
import torch

try:
    from .audio_func import mu_law_decode
    from .model1 import wavenet_autoencoder
    from .train import load_model
except ImportError:
    from music_amd.audio_func import mu_law_decode
    from music_amd.model1 import wavenet_autoencoder
    from music_amd.train import load_model


def predict_next(net, input_wav, quantization_channel=256):
    """generate.py:13-19 — argmax over the last chunk-row of the forward output."""
    with torch.no_grad():
        out = net(input_wav).view(-1, quantization_channel)
    return int(torch.topk(out[-1, :].view(-1), 1)[1])


def cached_decoder(net, encoding, cond):
    """The autoencoder's conditioned decoder (model1.py:158-225) for ONE pooled frame of encoding
    ``(1, bottleneck, 1)`` and fixed conditioning projections ``cond`` (N+1 (weight (C, bottleneck, 1), bias (C,))
    pairs, gate rows first, the last one for the post-processing stage) as a ``music_amd.model.wavenet``:
    filter / gate = second / first half of ``filter_gate`` (model1.py:188-190), conditioning folded into the
    biases.  Its forward and its cached-queue decoder (``fast_generate``) then reproduce the decoder exactly."""
    try:
        from .model import wavenet
    except ImportError:
        from music_amd.model import wavenet
    if encoding.dim() != 3 or encoding.size(0) != 1 or encoding.size(2) != 1:
        raise ValueError("cached_decoder needs a single pooled frame of encoding, got %s" % (tuple(encoding.shape),))
    N, Dd = len(net.dilations), net.de_dilation_channel
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    enc = encoding.detach().float().cpu()[0, :, 0]
    proj = [w.detach().float().cpu()[:, :, 0] @ enc + b.detach().float().cpu() for w, b in cond]
    zeros = lambda n: torch.zeros(n)
    bias = lambda name, n: sd[name + ".bias"] if net.use_bias else zeros(n)
    out = {"causal_layer.weight": sd["de_causal_layer.weight"],
           "causal_layer.bias": bias("de_causal_layer", net.de_residual_channel)}
    for i in range(N):
        fg, dn, sk = ("de_dilation_layer_stack.%d" % (3 * i + k) for k in range(3))
        w, b = sd[fg + ".weight"], bias(fg, 2 * Dd) + proj[i]
        out["dilation_layer_stack.%d.weight" % (4 * i)], out["dilation_layer_stack.%d.bias" % (4 * i)] = w[Dd:], b[Dd:]
        out["dilation_layer_stack.%d.weight" % (4 * i + 1)], out["dilation_layer_stack.%d.bias" % (4 * i + 1)] = w[:Dd], b[:Dd]
        out["dilation_layer_stack.%d.weight" % (4 * i + 2)] = sd[dn + ".weight"]
        out["dilation_layer_stack.%d.bias" % (4 * i + 2)] = bias(dn, net.de_residual_channel)
        out["dilation_layer_stack.%d.weight" % (4 * i + 3)] = sd[sk + ".weight"]
        out["dilation_layer_stack.%d.bias" % (4 * i + 3)] = bias(sk, net.de_skip_channel)
    out["post_process_1.weight"] = sd["connection_1.weight"]
    out["post_process_1.bias"] = bias("connection_1", net.de_skip_channel) + proj[N]
    out["post_process_2.weight"] = sd["connection_2.weight"]
    out["post_process_2.bias"] = bias("connection_2", net.quantization_channel)
    wnet = wavenet(filter_width=net.filter_width, dilations=list(net.dilations), dilation_channels=Dd,
                   residual_channels=net.de_residual_channel, skip_channels=net.de_skip_channel,
                   quantization_channels=net.quantization_channel, use_bias=True)
    wnet.load_state_dict({k: v.contiguous() for k, v in out.items()})
    return wnet.cuda()


def generate_cached(net, start_piece, note_num, cond=None, temperature=None, seed=0):
    """encode ``start_piece`` (1, Q, >= receptive_field) once, draw the conditioning projections once
    (or take ``cond``), then generate ``note_num`` codes with the persistent cached-queue decoder (corrected queue
    recurrence).  Returns (codes int64 on the device, the wavenet-form decoder, the encoding)."""
    try:
        from . import fast_generate as fg
    except ImportError:
        from music_amd import fast_generate as fg
    with torch.no_grad():
        net(start_piece.cuda())                              # sets net.last_encoding (model1.py:256-268)
    enc = net.last_encoding
    if enc.size(2) != 1:
        raise ValueError("the start piece pools to %d frames; the cached decoder needs exactly one "
                         "(receptive_field + pool .. receptive_field + 2*pool - 1 samples)" % enc.size(2))
    if cond is None:
        cond = net._draw_conditioning()
    wnet = cached_decoder(net, enc, cond)
    codes = fg.generate_codes(wnet, start_piece[:, :, -wnet.receptive_field:].cuda(), note_num, correct_queue=True,
                              temperature=temperature, seed=seed)
    return codes, wnet, enc


def generate_codes_naive(net, start_piece, note_num, sliding_window=False, window=None, seed=None):
    """The loop of generate.py:44-55: one full forward (encoder + conditioned decoder, fresh random conditioning
    projections) per generated code.  ``sliding_window`` False = as written (growing window), True = the last
    ``window - 1`` samples + the new one (``window`` defaults to receptive_field + 512).  ``seed``: torch.manual_seed(seed + i)
    before step i (the projections are drawn from the global RNG), for reproducible runs and the golden tests."""
    win = window if window is not None else net.receptive_field + 512
    input_wav = start_piece.cuda()
    generated = []
    for i in range(note_num):
        if seed is not None:
            torch.manual_seed(int(seed) + i)
        code = predict_next(net, input_wav, net.quantization_channel)
        generated.append(code)
        note = torch.zeros(1, net.quantization_channel, 1, device=input_wav.device)
        note[0, code, 0] = 1.0
        keep = input_wav[:, :, -(win - 1):] if sliding_window else input_wav
        input_wav = torch.cat((keep, note), 2)
    return generated


def generate(model_path, model_name, generate_path, generate_name, start_piece=None, sr=16000, duration=10,
             sliding_window=False, seed=None):
    if os.path.exists(generate_path) is False:
        os.makedirs(generate_path)
    with open('./params/model_params.json') as f:
        model_params = json.load(f)
    net = wavenet_autoencoder(**model_params)
    net = load_model(net, model_path, model_name)
    if net is None:
        raise FileNotFoundError(model_path + model_name)
    net = net.cuda()
    if start_piece is None:
        start_piece = torch.zeros(1, 256, net.receptive_field + 512)
        start_piece[:, 128, :] = 1.0
    generated = generate_codes_naive(net, start_piece, int(duration * sr), sliding_window=sliding_window,
                                     window=start_piece.size(2), seed=seed)
    audio = mu_law_decode(torch.tensor(generated, dtype=torch.int64), net.quantization_channel).cpu().numpy()
    from scipy.io import wavfile
    wavfile.write(generate_path + generate_name, sr, audio.astype(np.float32))
    return generated
