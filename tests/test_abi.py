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


def _chain_items(lib, t_lo, t_hi, batch, d, wg):
    import ctypes
    out = (ctypes.c_int * (3 * 4096))()
    n = lib.wn_resblock_bwd_pq_chain_items(t_lo, t_hi, batch, d, wg, out, 4096)
    return n, [(out[3 * k], out[3 * k + 1], out[3 * k + 2]) for k in range(max(n, 0))]


def test_chain_plan_covers_every_item_once():
    """The chain form of wn_resblock_bwd_pq (host view of its plan, nothing launched): over the workgroups of a launch every
    32-column item of every clip is owned exactly once, items of a workgroup follow their chain downwards in steps of d,
    a halo item is the item d columns above the first owned one, the top / bottom flags mark the ends of every chain, and
    the tails (t0 - d of the bottoms) tile [t_base - d, t_base) once per clip."""
    from music_amd import _lib
    lib = _lib.load()
    cases = [(3071, 16000, 8, 512), (2047, 16000, 8, 256), (1100, 16000, 8, 32), (3071, 16000, 1, 512), (3071, 16000, 64, 512),
             (40, 16000, 64, 32), (1024, 4000, 1, 512), (1024, 1024 + 16 * 32, 2, 512), (100, 100 + 33, 3, 32), (512, 9000, 5, 64),
             (77, 3000, 2, 96)]
    for t_lo, t_hi, batch, d in cases:
        assert lib.wn_resblock_bwd_pq_chain_ok(t_lo, t_hi, batch, d) == 1, (t_lo, t_hi, batch, d)
        nwg = lib.wn_resblock_bwd_pq_slabs(t_lo, t_hi, batch, d, 1)
        assert 1 <= nwg <= 256
        t_base = t_lo & ~31
        steps = (t_hi - t_base + 31) // 32
        owned, tails = {}, {}
        for wg in range(nwg):
            n, items = _chain_items(lib, t_lo, t_hi, batch, d, wg)
            assert n >= 0
            prev = None
            for k, (b, t0, fl) in enumerate(items):
                assert 0 <= b < batch and (t0 - t_base) % 32 == 0 and t_base <= t0 < t_base + 32 * steps
                halo, top, bot = fl & 1, fl & 2, fl & 4
                assert bool(halo) == (k == 0 and not top and n > 0 and items[0][2] & 1 == 1)
                assert bool(top) == (t0 + d >= t_base + 32 * steps)       # nothing above it
                assert bool(bot) == (t0 - d < t_base)
                if prev is not None and not top:
                    assert (b, t0 + d) == prev[:2]                         # the item above it came just before
                if k == 0 and not top:
                    assert halo                                            # a segment that starts inside a chain brings the item above it
                if halo:
                    assert k == 0 and n > 1
                else:
                    assert (b, t0) not in owned
                    owned[(b, t0)] = wg
                    if bot:
                        tails[(b, t0 - d)] = 1
                prev = (b, t0, fl)
        assert len(owned) == batch * steps, (t_lo, t_hi, batch, d, len(owned), batch * steps)
        for b in range(batch):
            assert sorted(t for (bb, t) in tails if bb == b) == list(range(t_base - d, t_base, 32))
    # no chain form: d not a multiple of 32, fewer items than chains
    assert lib.wn_resblock_bwd_pq_chain_ok(100, 16000, 8, 16) == 0
    assert lib.wn_resblock_bwd_pq_chain_ok(1024, 1024 + 15 * 32, 2, 512) == 0
    assert _chain_items(lib, 100, 16000, 8, 16, 0)[0] == -1


def test_collective_entry_points_check_their_arguments_without_a_device():
    """wn_allreduce_flat / wn_comm_* (the thin ncclAllReduce wrapper on the caller's communicator): RCCL is resolved
    at first use from the process image, not linked; bad arguments come back as -4 before RCCL is touched, and the error text
    names the function.  (The 1-rank collective itself runs in tests/test_gpu_dist.py.)"""
    import ctypes
    from music_amd import _lib
    lib = _lib.load()
    if not lib.wn_coll_available():
        assert lib.wn_allreduce_flat(None, None, 0, None) == -5 and b"RCCL" in lib.wn_last_error()
        return
    assert lib.wn_allreduce_flat(None, None, 16, None) == -4
    assert b"wn_allreduce_flat" in lib.wn_last_error()
    comm = ctypes.c_void_p()
    assert lib.wn_comm_create(0, 0, b"\0" * 128, ctypes.byref(comm)) == -4
    assert lib.wn_comm_create(2, 2, b"\0" * 128, ctypes.byref(comm)) == -4
    assert lib.wn_comm_destroy(None) == 0



def test_null_required_pointers_are_reported_not_dereferenced():
    """Every compute entry point checks its REQUIRED pointers before anything is launched (a NULL that reaches a kernel is a memory
    fault on the device and takes the caller's process with it): status -4, wn_last_error names the function and the argument.  A call
    with no work (zero rows / columns / clips) may carry NULLs and returns 0.  Runs without a device: the checks come first."""
    import ctypes
    from music_amd import _lib
    lib = _lib.load()
    P = 1 << 20            # "some non-NULL address": never dereferenced, every case below fails (or is empty) before a launch

    def bad(name, arg, *args):
        rc = getattr(lib, name)(*args)
        msg = lib.wn_last_error().decode()
        assert rc == -4 and name.replace("_batch_pk", "").replace("_batch", "") in msg and ("'%s'" % arg) in msg, (name, rc, msg)

    bad("wn_pack_weights", "idx", P, None, P, 512, 0, None)
    assert lib.wn_pack_weights(None, None, None, 0, 0, None) == 0
    #            in0 in1 bs pitch lo hi s0 s1 ks0 ks1 wpack mt mv out obs op osh bias resid rbs rp rlo mask mbs mp t_lo t_hi relu batch mode
    gemm = [P, None, 64, 64, 0, 64, 0, 0, 1, 0, P, 1, 16, P, 64, 64, 0, None, None, 0, 0, 0, None, 0, 0, 0, 64, 0, 1, 0, None]
    for i, arg in ((0, "in0"), (10, "wpack"), (13, "out")):
        a = list(gemm)
        a[i] = None
        bad("wn_chan_gemm", arg, *a)
    a = list(gemm)
    a[0], a[28] = None, 0                                    # batch 0: nothing to do
    assert lib.wn_chan_gemm(*a) == 0
    #      x_in x_out z_out xbs zbs pitch wfg wd bf bg bd n_f n_d ch d t_lo t_hi z_lo write_x cond cbs cp cm cle cq cpk cpbs cidx zhalf batch mode
    fwd = [P, P, P, 64, 64, 64, P, P, None, None, None, 32, 32, 32, 1, 2, 66, 2, 1, None, 0, 0, 0, 0, 0, None, 0, None, 0, 1, 0, None]
    for i, arg in ((0, "x_in"), (1, "x_out"), (2, "z_out"), (6, "wfg"), (7, "wd")):
        a = list(fwd)
        a[i] = None
        bad("wn_resblock_fwd", arg, *a)
    #         z zbs pitch ks w_skip bias u h sbs w_p1c b1 w_p2c b2 o obs op s_valid q_valid t_lo t_hi batch mode
    epf = [P, 1 << 20, 512, 4, P, None, P, P, 1 << 20, P, None, P, None, P, 1 << 20, 200, 256, 256, 100, 300, 1, 0, None]
    for i, arg in ((0, "z"), (4, "w_skip"), (6, "u"), (7, "h"), (9, "w_p1c"), (11, "w_p2c"), (13, "o")):
        a = list(epf)
        a[i] = None
        bad("wn_skip_epilogue_fwd", arg, *a)
    a = list(epf)
    a[2] = 256                                               # the 128-column tiles over [64, 300) do not fit a pitch of 256: refused, not read
    assert lib.wn_skip_epilogue_fwd(*a) == -4 and "pitch" in lib.wn_last_error().decode()
    a = list(epf)
    a[3] = 3                                                 # an odd number of k-steps
    assert lib.wn_skip_epilogue_fwd(*a) == -4
    a = list(epf)
    a[0], a[20] = None, 0                                    # batch 0: nothing to do
    assert lib.wn_skip_epilogue_fwd(*a) == 0
    #         d_o obs op h u sbs pitch d_h d_u d_z zbs w_p2T w_p1Tc w_skipTc mt_z z_valid s_valid t_lo t_hi batch mode
    epb = [P, 1 << 20, 200, P, P, 1 << 20, 512, P, P, P, 1 << 20, P, P, P, 6, 96, 256, 100, 300, 1, 2, None]
    for i, arg in ((0, "d_o"), (3, "h"), (4, "u"), (7, "d_h"), (8, "d_u"), (9, "d_z"), (11, "w_p2T"), (12, "w_p1Tc"), (13, "w_skipTc")):
        a = list(epb)
        a[i] = None
        bad("wn_skip_epilogue_bwd", arg, *a)
    a = list(epb)
    a[14] = 7                                                # z row tiles not in groups of 3
    assert lib.wn_skip_epilogue_bwd(*a) == -4
    a = list(epb)
    a[6] = 256
    assert lib.wn_skip_epilogue_bwd(*a) == -4 and "pitch" in lib.wn_last_error().decode()
    bad("wn_wgrad", "c", P, 64, 64, 0, 64, P, None, 64, 64, 0, 0, 64, 2, 2, 0, None, 32, 1 << 20, 0, 64, 64, 1, 2, None)
    bad("wn_reduce_slabs", "slab", P, 1, 4, None, P, None)
    assert lib.wn_reduce_slabs(None, 0, 0, None, None, None) == 0
    bad("wn_bias_grad", "out", P, 64, 64, 0, 4, 0, 64, 1, None, None)
    bad("wn_chunk_softmax256_fwd", "y", P, None, 4, None)
    bad("wn_chunk_softmax256_bwd", "dy", P, None, P, 4, None)
    bad("wn_chunk_softmax256_ce", "target", P, None, None, None, None, 4, 0.25, None)
    assert lib.wn_chunk_softmax256_ce(None, None, None, None, None, 0, 1.0, None) == 0
    bad("wn_adam_flat", "v", P, P, P, None, 8, 1e-3, 0.9, 0.999, 1e-8, 0.1, 0.001, 1.0, None)
    bad("wn_sgd_flat", "g", P, None, None, 8, 0.1, 0.0, 1.0, 1, None)
    bad("wn_rmsprop_flat", "p", None, P, P, None, 8, 0.1, 0.99, 1e-8, 0.0, 1.0, None)
    bad("wn_gather_grads", "flat_grad", P, P, None, 8, None)
    bad("wn_onehot", "codes", None, P, 1, 256, 8, 0, None)
    bad("wn_mulaw_encode_tbl", "thresholds", P, None, P, 8, None)
    bad("wn_mulaw_decode_lut", "audio", P, P, None, 8, None)
    bad("wn_shift_add", "q", P, None, P, 64, 64, 4, 1, 2, 1, 64, 1, None)
    dil = (ctypes.c_int32 * 2)(1, 2)
    qoff = (ctypes.c_int64 * 2)(0, 64)
    dec = [2, 32, 32, 64, 256, ctypes.cast(dil, ctypes.c_void_p), ctypes.cast(qoff, ctypes.c_void_p), P, P, None, P, 0, None, P, None, P, None,
           None, None, None, None, None, P, None, 0, 4, 1, P, None]
    for i, arg in ((5, "dilations_host"), (7, "queues"), (10, "w_layers"), (27, "sync")):
        a = list(dec)
        a[i] = None
        bad("wn_decode", arg, *a)
