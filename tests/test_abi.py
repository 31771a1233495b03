"""CPU checks of the boundary: the C-ABI library builds/loads and exports exactly the entry points
include/wavenet_hip.h declares (no compute calls without a GPU), and the host-side packing maps."""
import ctypes
import os
import re

import numpy as np

from tests.helpers import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "wavenet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from music_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert "wn_resblock_fwd" in names and "wn_chunk_softmax256_fwd" in names and len(names) >= 15
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libwavenet_hip.so does not export %s" % n
    # the ctypes table binds exactly the declared set
    assert sorted(list(_lib.SIGNATURES) + ["wn_last_error"]) == names
    assert lib.wn_version() == _lib.ABI_VERSION


def test_missing_library_fails_loudly(monkeypatch):
    from music_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libwavenet_hip.so")
    try:
        _lib.load()
    except _lib.WavenetHipError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the library is missing")


def test_pack_positions_cover_matrix_once():
    from music_amd.engine import pack_index
    for chained in (False, True):
        m, k = 48, 96
        weff = np.arange(m * k, dtype=np.int64).reshape(m, k)
        idx = pack_index(weff, chained)
        assert idx.shape == (m * k,) and sorted(idx.tolist()) == list(range(m * k))
    # natural order: lane (c,q), element j of fragment (m,s) is W[16m+c][32s+8q+j]
    idx = pack_index(weff, False).reshape(3, 3, 64, 8)
    assert idx[1, 2, 5 + 16 * 3, 6] == (16 + 5) * k + 64 + 24 + 6
    # chained order: k = 32s + 16(j>>2) + 4q + (j&3)
    idx = pack_index(weff, True).reshape(3, 3, 64, 8)
    assert idx[0, 1, 2 + 16 * 1, 5] == 2 * k + 32 + 16 + 4 + 1


def test_module_without_gpu_raises_not_falls_back():
    import torch
    from music_amd.model import wavenet
    net = wavenet(2, [1, 2], 16, 16, 16, 256, False)
    assert net.receptive_field == 5
    assert list(net.state_dict().keys())[:3] == ["causal_layer.weight", "dilation_layer_stack.0.weight",
                                                 "dilation_layer_stack.1.weight"]
    try:
        net(torch.zeros(1, 256, 4))
    except ValueError as e:
        assert "wave sample not long enough" in str(e)
    else:
        raise AssertionError
    try:
        net(torch.zeros(1, 256, 8))
    except RuntimeError as e:
        assert "no CPU path" in str(e)
    else:
        raise AssertionError("CPU input must raise")
