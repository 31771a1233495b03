"""Counterpart of the reference's ``wavenet/audio_func.py`` (mu-law companding) on the MI355X.

``mu_law_encode`` is the CANONICAL encoder of the path: bit-exact against the
reference's float32 torch formula through its 255 float32 decision thresholds
(music_amd/mulaw_tables.npz, derived from the reference by bisection in tools/make_golden.py),
evaluated by the HIP kernel ``wn_mulaw_encode_tbl``.  ``mu_law_decode`` is the 256-entry table of
the reference's decode formula (``wn_mulaw_decode_lut``).  Inputs may live on any device; the work
always runs on the GPU (no CPU implementation) and the result is returned on the input's device.
``trim_silence`` (librosa, offline data prep) is out of scope.
"""
import os

import numpy as np
import torch

try:
    from . import _lib
except ImportError:
    from music_amd import _lib

_TABLES = {}


def _tables(device):
    key = str(device)
    if key not in _TABLES:
        d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mulaw_tables.npz"))
        _TABLES[key] = (torch.from_numpy(d["thresholds"]).to(device), torch.from_numpy(d["decode_table"]).to(device))
    return _TABLES[key]


def _gpu(t):
    if not torch.cuda.is_available():
        raise RuntimeError("music_amd.audio_func runs on an MI355X (ROCm) device only; there is no CPU path")
    return t if t.is_cuda else t.cuda()


def mu_law_encode(audio, quantization_channels=256):
    """wavenet/audio_func.py:5-22 — float audio (any shape) -> int64 codes in [0, 255]."""
    if quantization_channels != 256:
        raise NotImplementedError("the threshold table is for quantization_channels == 256")
    a = _gpu(audio.detach().to(torch.float32)).contiguous()
    thr, _ = _tables(a.device)
    codes = torch.empty(a.shape, dtype=torch.uint8, device=a.device)
    _lib.call("wn_mulaw_encode_tbl", _lib.ptr(a), _lib.ptr(thr), _lib.ptr(codes), a.numel(), _lib.stream())
    return codes.to(torch.int64).to(audio.device)


def mu_law_decode(output, quantization_channels=256):
    """wavenet/audio_func.py:24-39 — int codes -> float32 audio in (-1, 1)."""
    if quantization_channels != 256:
        raise NotImplementedError("the decode table is for quantization_channels == 256")
    c = _gpu(output.detach()).to(torch.uint8).contiguous()
    _, tab = _tables(c.device)
    out = torch.empty(c.shape, dtype=torch.float32, device=c.device)
    _lib.call("wn_mulaw_decode_lut", _lib.ptr(c), _lib.ptr(tab), _lib.ptr(out), c.numel(), _lib.stream())
    return out.to(output.device)
