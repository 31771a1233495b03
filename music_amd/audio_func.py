"""Counterpart of the reference's ``wavenet/audio_func.py`` (mu-law companding) on the MI355X.

``mu_law_encode`` is the CANONICAL encoder of the path: bit-exact against the
reference's float32 torch formula through its float32 decision thresholds, evaluated by the HIP
kernels ``wn_mulaw_encode_tbl`` (256 channels: the 255 thresholds of music_amd/mulaw_tables.npz,
derived from the reference by bisection in tools/make_golden.py) and ``wn_mulaw_encode_q`` (any
other ``quantization_channels``: the q - 1 thresholds are found on first use by the same bisection
over the float32 ordering, on the formula of audio_func.py:16-22 evaluated with the ATen float32
ops the reference itself calls - a table of constants, like a packed weight; every sample is
encoded on the device).  ``mu_law_decode`` is the q-entry table of the reference's decode formula
(audio_func.py:35-39).  Inputs may live on any device; the work always runs on the GPU (no CPU
implementation) and the result is returned on the input's device.
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


def _formula_encode(a, q):
    """audio_func.py:16-22 on a float32 CPU tensor, op for op (used for the threshold table only)."""
    mu = torch.Tensor([q - 1]).float()
    safe_audio_abs = torch.abs(torch.clamp(a, -1.0, 1.0))
    magnitude = torch.log1p(mu * safe_audio_abs) / torch.log1p(mu)
    signal = torch.sign(a) * magnitude
    return ((signal + 1) / 2 * mu + 0.5).long()


def build_tables(q):
    """(thresholds float32 (q-1,), decode table float32 (q,)) of ``quantization_channels`` = q.  thresholds[k] = the smallest
    float32 the reference encoder maps to a code > k (it is monotone); 33 bisection rounds over the float32 ordering for
    all codes at once."""
    q = int(q)
    if q < 2:
        raise ValueError("quantization_channels must be at least 2")
    f2o = lambda v: int(np.array([v], dtype=np.float32).view(np.int32)[0])        # 1.0 / -1.0 -> bit pattern (both orderable)
    to_float = lambda o: torch.where(o >= 0, o, -(2 ** 31) - o).to(torch.int32).view(torch.float32)
    k = torch.arange(1, q, dtype=torch.int64)
    neg1 = -(2 ** 31) - f2o(-1.0)                           # ordered key of -1.0 (negative floats: -2^31 - bits, as signed)
    lo = torch.full_like(k, neg1)                           # encode(lo) < k   (encode(-1) = 0)
    hi = torch.full_like(k, f2o(1.0))                       # encode(hi) >= k  (encode(1) = q - 1)
    while bool(((hi - lo) > 1).any()):
        mid = lo + (hi - lo) // 2
        ge = _formula_encode(to_float(mid), q) >= k
        hi = torch.where(ge, mid, hi)
        lo = torch.where(ge, lo, mid)
    thr = to_float(hi).clone()
    assert bool((thr[1:] > thr[:-1]).all()) if q > 2 else True
    mu = torch.Tensor([q - 1]).float()
    signal = 2.0 * (torch.arange(q).float() / mu) - 1.0     # audio_func.py:35-39
    magnitude = (1.0 / mu) * ((1.0 + mu) ** torch.abs(signal) - 1.0)
    return thr, (torch.sign(signal) * magnitude).clone()


def _tables(device, q=256):
    key = (str(device), int(q))
    if key not in _TABLES:
        if q == 256:
            d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mulaw_tables.npz"))
            thr, tab = torch.from_numpy(d["thresholds"]), torch.from_numpy(d["decode_table"])
        else:
            thr, tab = build_tables(q)
        _TABLES[key] = (thr.to(device), tab.to(device))
    return _TABLES[key]


def _gpu(t):
    if not torch.cuda.is_available():
        raise RuntimeError("music_amd.audio_func runs on an MI355X (ROCm) device only; there is no CPU path")
    return t if t.is_cuda else t.cuda()


def mu_law_encode(audio, quantization_channels=256):
    """wavenet/audio_func.py:5-22 — float audio (any shape) -> int64 codes in [0, quantization_channels)."""
    q = int(quantization_channels)
    a = _gpu(audio.detach().to(torch.float32)).contiguous()
    thr, _ = _tables(a.device, q)
    if q == 256:
        codes = torch.empty(a.shape, dtype=torch.uint8, device=a.device)
        _lib.call("wn_mulaw_encode_tbl", _lib.ptr(a), _lib.ptr(thr), _lib.ptr(codes), a.numel(), _lib.stream())
    else:
        codes = torch.empty(a.shape, dtype=torch.int32, device=a.device)
        _lib.call("wn_mulaw_encode_q", _lib.ptr(a), _lib.ptr(thr), q, _lib.ptr(codes), a.numel(), _lib.stream())
    return codes.to(torch.int64).to(audio.device)


def mu_law_decode(output, quantization_channels=256):
    """wavenet/audio_func.py:24-39 — int codes -> float32 audio in (-1, 1)."""
    q = int(quantization_channels)
    c = _gpu(output.detach())
    _, tab = _tables(c.device, q)
    out = torch.empty(c.shape, dtype=torch.float32, device=c.device)
    if q == 256:
        c = c.to(torch.uint8).contiguous()
        _lib.call("wn_mulaw_decode_lut", _lib.ptr(c), _lib.ptr(tab), _lib.ptr(out), c.numel(), _lib.stream())
    else:
        c = c.to(torch.int32).contiguous()
        _lib.call("wn_mulaw_decode_q", _lib.ptr(c), _lib.ptr(tab), q, _lib.ptr(out), c.numel(), _lib.stream())
    return out.to(output.device)
