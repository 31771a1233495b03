"""CPU tests of the host side: dataset chopping vs the golden vectors, the train() loop's files and
formats vs the text the reference's own train() wrote (G7), checkpoint compatibility."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, load_npz


def test_dataset_pieces_match_reference(tmp_path):
    from music_amd.faster_audio_data import audio_dataset
    d = load_npz("g4_data.npz")
    n = 0
    while "mp%d_lens" % n in d:
        rf, win = (int(v) for v in d["mp%d_rf_win" % n])
        flat, data, o = d["mp%d_data" % n], [], 0
        for l in d["mp%d_lens" % n]:
            data.append(flat[o:o + l].astype(np.int32))
            o += l
        path = tmp_path / ("a%d.pkl" % n)
        pickle.dump(data, open(path, "wb"))
        ds = audio_dataset(str(path), rf, win)
        assert len(ds) == int(d["mp%d_n" % n])
        assert [len(p["audio_piece"]) for p in ds.data] == list(d["mp%d_piece_len" % n])
        assert [int(p["audio_piece"].long().sum()) for p in ds.data] == list(d["mp%d_piece_sum" % n])
        assert [int(p["audio_piece"][0]) for p in ds.data] == list(d["mp%d_piece_first" % n])
        assert [int(p["audio_target"].sum()) for p in ds.data] == list(d["mp%d_target_sum" % n])
        assert all(p["audio_target"].dtype == torch.int64 for p in ds.data)
        n += 1
    assert n == 3
    pickle.dump([np.arange(30, dtype=np.int32)], open(tmp_path / "short.pkl", "wb"))
    with pytest.raises(NameError):
        audio_dataset(str(tmp_path / "short.pkl"), 20, 30)


def _setup_run(tmp_path, g7, monkeypatch, gain):
    from music_amd import train as T
    from music_amd import faster_audio_data as fad
    from tests.cpu_model import OracleWavenet, onehot_oracle
    os.makedirs(tmp_path / "params")
    rng = np.random.default_rng(g7["data_seed"])
    data = [rng.integers(0, 256, size=(l,)).astype(np.int32) for l in g7["data_lens"]]
    pickle.dump(data, open(tmp_path / "np_audio.pkl", "wb"))
    dp = dict(g7["dataset_params"], audio_path=str(tmp_path / "np_audio.pkl"))
    for n, p in (("wavenet", g7["wavenet_params"]), ("dataset", dp), ("train", g7["train_params"])):
        json.dump(p, open(tmp_path / "params" / (n + "_params.json"), "w"))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(fad, "onehot_device", onehot_oracle)

    def ctor(**kw):
        net = OracleWavenet(**kw)
        if gain is not None:
            with torch.no_grad():
                for p in net.parameters():
                    p.mul_(gain)
        return net
    monkeypatch.setattr(T, "wavenet", ctor)
    return T


@pytest.mark.parametrize("tag", ["plain", "gain"])
def test_train_loop_files_match_reference(tmp_path, monkeypatch, tag):
    g7 = json.load(open(os.path.join(GOLDEN, "g7_train.json")))
    T = _setup_run(tmp_path, g7, monkeypatch, None if tag == "plain" else g7["gain"])
    torch.manual_seed(0)
    T.train()
    got = open(tmp_path / "log" / "loss_log.log").read()
    want = g7[tag + "_loss_log"]
    gl, wl = got.strip().split("\n"), want.strip().split("\n")
    assert len(gl) == len(wl)
    for a, b in zip(gl, wl):
        assert a.split("Average")[0] == b.split("Average")[0]          # "Trained over N pieces,"
        assert abs(float(a.split(' ')[-1]) - float(b.split(' ')[-1])) < 1e-6
    assert open(tmp_path / "log" / "store_log.log").read() == g7[tag + "_store_log"]
    assert sorted(os.listdir(tmp_path / "restore")) == g7[tag + "_files"]
    ck = torch.load(tmp_path / "restore" / "wavenet2.model")
    assert list(ck.keys()) == g7[tag + "_ckpt_keys"]
    assert [list(v.shape) for v in ck.values()] == g7[tag + "_ckpt_shapes"]
    for v, s in zip(ck.values(), g7[tag + "_ckpt_abs_sum"]):
        assert abs(float(v.double().abs().sum()) - s) <= 1e-4 * max(1.0, s)
    # resume: counter continues from the last log line, epoch number from the file name
    tp = dict(g7["train_params"], restore_model="wavenet2.model", num_epochs=1)
    json.dump(tp, open(tmp_path / "params" / "train_params.json", "w"))
    T.train()
    lines = open(tmp_path / "log" / "loss_log.log").read().strip().split("\n")
    assert int(lines[-1].split(' ')[2]) > int(wl[-1].split(' ')[2])
    assert open(tmp_path / "log" / "store_log.log").read().endswith("Epoch 3, model saved!\n")
    assert "wavenet3.model" in os.listdir(tmp_path / "restore")


def test_checkpoint_rotation_and_module_prefix(tmp_path):
    from music_amd import train as T
    from music_amd.model import wavenet
    rd = str(tmp_path) + "/"
    net = wavenet(2, [1, 2], 16, 16, 16, 256, False)
    for n in (3, 10, 4):
        T.save_model(net, n, rd)
    T._rotate_checkpoints(rd, 3)                      # numeric, not lexicographic: deletes 3
    assert sorted(os.listdir(rd)) == ["wavenet10.model", "wavenet4.model"]
    sd = torch.load(rd + "wavenet4.model")
    torch.save({"module." + k: v for k, v in sd.items()}, rd + "wavenet5.model")
    net2 = wavenet(2, [1, 2], 16, 16, 16, 256, False)
    assert T.load_model(net2, rd, "wavenet5.model") is net2
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))
    assert T.load_model(net2, rd, "missing.model") is None
    assert T.get_optimizer(net, "sgd", 0.1, 0.9).defaults["momentum"] == 0.9
    assert isinstance(T.get_optimizer(net, "rmsprop", 0.1, 0.9), torch.optim.RMSprop)
    assert isinstance(T.get_optimizer(net, "adam", 0.1, 0.9), torch.optim.Adam)
    assert T.get_optimizer(net, "lbfgs", 0.1, 0.9) is None


def test_autoencoder_harness_surface(tmp_path):
    """ae_train: optimizer names, checkpoint prefix / epoch parsing (the reference's own [7:] slice
    raises on its file names), module import without side effects."""
    from music_amd import ae_train as A
    from music_amd.model1 import wavenet_autoencoder
    cfg = json.load(open(os.path.join(GOLDEN, "g8_cfg.json")))
    net = wavenet_autoencoder(**cfg)
    d = load_npz("g8_autoencoder.npz")
    assert list(net.state_dict().keys()) == [k[2:] for k in d if k.startswith("w:")]
    rd = str(tmp_path) + "/"
    A.save_model(net, 12, rd)
    assert os.listdir(rd) == ["wavenet_autoencoder12.model"]
    assert A._epoch_of(rd + "wavenet_autoencoder12.model") == 12
    net2 = wavenet_autoencoder(**cfg)
    assert A.load_model(net2, rd, "wavenet_autoencoder12.model") is net2
    assert isinstance(A.get_optimizer(net, "Adam", 1e-3), torch.optim.Adam)
    assert isinstance(A.get_optimizer(net, "RMSprop", 1e-3, 0.5), torch.optim.RMSprop)
    assert isinstance(A.get_optimizer(net, "sgd", 1e-3, 0.5), torch.optim.SGD)
    assert isinstance(A.get_optimizer(net, "lbfgs", 1e-3), torch.optim.LBFGS)
    import music_amd.ae_generate  # noqa: F401  (no import-time generation)


def test_optimizer_state_survives_a_restart(tmp_path, monkeypatch):
    """with "save_optimizer_state" a run that is stopped after epoch 2 and resumed from
    wavenet2.model (+ wavenet2.opt) for one more epoch ends with the same weights as an uninterrupted
    3-epoch run (Adam: the moments and the step count come back); without the .opt file it does not."""
    import shutil
    g7 = json.load(open(os.path.join(GOLDEN, "g7_train.json")))
    finals = {}
    for mode in ("straight", "resumed", "resumed_cold"):
        root = tmp_path / mode
        os.makedirs(root)
        T = _setup_run(root, g7, monkeypatch, g7["gain"])
        base = dict(g7["train_params"], optimizer="adam", learning_rate=1e-3, save_optimizer_state=True,
                    check_point_every=1, max_check_points=10)
        if mode == "straight":
            json.dump(dict(base, num_epochs=3), open(root / "params" / "train_params.json", "w"))
            torch.manual_seed(0)
            T.train()
        else:
            json.dump(dict(base, num_epochs=2), open(root / "params" / "train_params.json", "w"))
            torch.manual_seed(0)
            T.train()
            assert os.path.exists(root / "restore" / "wavenet2.opt")
            if mode == "resumed_cold":
                os.remove(root / "restore" / "wavenet2.opt")
            json.dump(dict(base, num_epochs=1, restore_model="wavenet2.model"), open(root / "params" / "train_params.json", "w"))
            T.train()
        finals[mode] = torch.load(root / "restore" / "wavenet3.model")
        monkeypatch.undo()
    same = all(torch.equal(a, b) for a, b in zip(finals["straight"].values(), finals["resumed"].values()))
    cold_same = all(torch.equal(a, b) for a, b in zip(finals["straight"].values(), finals["resumed_cold"].values()))
    assert same and not cold_same


def test_generic_engine_index_maps_cover_every_parameter_once():
    """music_amd/engine_generic.py (any filter_width / channel counts): every weight appears exactly once in the forward
    packs and once in the backward packs (the causal layer's transpose is there for the gradient w.r.t. the input), at the
    (row, k) its product expects; the gradient gather map is a bijection into the gradient matrices."""
    import numpy as np
    import torch
    from music_amd.engine import pack_positions
    from music_amd.engine_generic import GenericWaveNetEngine
    eng = GenericWaveNetEngine([1, 2, 4], 20, 24, 40, quantization_channels=48, filter_width=3, use_bias=True, device="cpu")
    assert eng.rf == 2 * (7 + 1) + 1 and eng.off == [2, 4, 8, 16] and eng.pairs == [(0, 1), (2, None)]
    n_w = sum(int(np.prod(eng.spec.shape[n])) for n in eng.param_names if n.endswith(".weight"))
    fidx = eng.pk_f_idx.numpy()
    seen = fidx[fidx >= 0]
    assert len(seen) == n_w and len(np.unique(seen)) == n_w
    bidx = eng.pk_b_idx.numpy()
    seen_b = bidx[bidx >= 0]
    assert len(seen_b) == n_w and len(np.unique(seen_b)) == n_w
    # the causal layer's transposed pack, element by element: rows = input channel q, K = [tap 0 rows r | tap 1 rows r] of pair 0
    o = eng.pk_b_off["causalT_0"] // 2
    mt, ks = eng.QP // 16, 2 * eng.RP // 32
    row, k = pack_positions(mt, ks, False)
    idx = bidx[o:o + mt * ks * 512]
    wc = eng.spec.conv("causal_layer.weight")            # [R][Q][k] of flat offsets
    for r, kk, v in zip(row[::53], k[::53], idx[::53]):
        tap, ch = divmod(int(kk), eng.RP)
        want = wc[ch, int(r), tap] if (int(r) < eng.Q and ch < eng.R) else -1
        assert v == want, (r, kk, v, want)
    g = eng.gidx.numpy()
    assert len(np.unique(g)) == eng.spec.total and g.min() >= 0 and g.max() < eng.gpack.numel()
    # one pack checked element by element: fg of layer 1, tap pair 0 = taps (0, 1): rows [f | g] x K = [tap 0 ch | tap 1 ch]
    o = eng.pk_f_off["fg1_0"] // 2                     # fragments are 1024 halfs (x3): offset in index entries
    mt, ks = 2 * eng.DP // 16, 2 * eng.RP // 32
    row, k = pack_positions(mt, ks, False)
    idx = fidx[o:o + mt * ks * 512]
    wf = eng.spec.conv("dilation_layer_stack.4.weight")
    wg = eng.spec.conv("dilation_layer_stack.5.weight")
    for r, kk, v in zip(row[::97], k[::97], idx[::97]):
        h, c = divmod(int(r), eng.DP)
        tap, ch = divmod(int(kk), eng.RP)
        want = (wf, wg)[h][c, ch, tap] if (c < eng.D and ch < eng.R) else -1
        assert v == want, (r, kk, v, want)
    # and the gradient of that weight lands where wn_wgrad writes it: C[h*DP + c][tap*RP + ch] of matrix fg1_0
    go, rows, cols = eng.gp_off["fg1_0"]
    assert g[wg[3, 5, 1]] == go + (eng.DP + 3) * cols + eng.RP + 5
    go2, _, cols2 = eng.gp_off["fg1_1"]
    assert g[wf[2, 7, 2]] == go2 + 2 * cols2 + 7


def test_workspace_pool_semantics_without_a_gpu():
    """WorkspacePool / WorkspaceHold (music_amd/engine.py): get() hands out the first workspace no pending backward holds,
    a held one is never reused or evicted, peek() is what the last forward used, shapes are evicted one at a time (LRU),
    a dropped hold releases on garbage collection, a fifth in-flight forward of one shape takes the OLDEST waiting forward's
    workspace over with a warning (the reference's autograd never runs out; a backward that still arrives for the evicted
    forward fails on its generation check)."""
    import pytest
    from music_amd.engine import WorkspaceHold, WorkspacePool
    made = []

    def make(B, T):
        ws = {"id": len(made), "gen": 0}
        made.append(ws)
        return ws
    pool = WorkspacePool(make)
    a = pool.get(1, 100)
    assert pool.get(1, 100) is a and pool.peek(1, 100) is a and len(pool) == 1
    a["gen"] = 1
    h = WorkspaceHold(a)
    b = pool.get(1, 100)                       # the first one is held: a second workspace of that shape
    assert b is not a and pool.peek(1, 100) is b and len(pool) == 2
    h.release()
    assert pool.get(1, 100) is a               # free again: reused, no third allocation
    # a hold whose workspace has been reused by a later forward (generation moved on) must not release it
    a["gen"] = 2
    stale = WorkspaceHold(a)
    a["gen"] = 3
    a["held"] = True
    stale.release()
    assert a["held"] is True
    a["held"] = False
    # garbage collection of the hold releases
    a["gen"] = 4
    h2 = WorkspaceHold(a)
    assert a["held"]
    del h2
    assert not a["held"]
    # LRU eviction of whole shapes, never a held one
    a["gen"] = 5
    keep = WorkspaceHold(a)
    for t in (101, 102, 103):
        pool.get(1, t)
    pool.get(1, 104)                           # fifth shape: (1, 100) is the oldest but held -> (1, 101) goes
    assert sorted(pool._d.keys()) == [(1, 100), (1, 102), (1, 103), (1, 104)]
    keep.release()
    # at most MAX_PER_SHAPE forwards in flight
    holds = []
    for g in range(WorkspacePool.MAX_PER_SHAPE):
        w = pool.get(2, 50)
        w["gen"] = g
        holds.append(WorkspaceHold(w))
    with pytest.warns(UserWarning, match="waiting for their backward"):
        w5 = pool.get(2, 50)
    assert w5 is holds[0].ws and w5["gen"] == 0 and not w5["held"]      # the oldest forward's workspace, free for the new forward
    w5["gen"] = 99                                                       # (the new forward stamps its generation)
    holds[0].release()                                                   # the evicted forward's hold no longer owns it
    assert pool.peek(2, 50) is w5
    pool.clear()
    assert len(pool) == 0


def test_conditioning_draw_equals_the_reference_modules_bit_for_bit():
    """wavenet_autoencoder._draw_conditioning draws the 31 per-forward conditioning convs WITHOUT building nn.Conv1d modules
    (wavenet_autoencoder/model1.py:178,216 builds them: SURVEY Q8): the same values from the same global RNG state, and the
    generator is left where the reference would leave it."""
    import torch.nn as nn
    from music_amd.model1 import wavenet_autoencoder
    for bw, dd, sk in ((512, 32, 512), (64, 64, 256), (10, 60, 72), (3, 7, 5)):
        cfg = dict(filter_width=2, quantization_channel=256, dilations=[1, 2, 4], en_residual_channel=8, en_dilation_channel=8,
                   en_bottleneck_width=bw, en_pool_kernel_size=4, de_residual_channel=8, de_dilation_channel=dd,
                   de_skip_channel=sk, use_bias=False)
        ae = wavenet_autoencoder(**cfg)
        torch.manual_seed(11)
        got = ae._draw_conditioning()
        after = torch.rand(1).item()
        torch.manual_seed(11)
        for i, (w, b) in enumerate(got):
            c = nn.Conv1d(bw, 2 * dd if i < 3 else sk, 1)
            assert torch.equal(w, c.weight.detach()) and torch.equal(b, c.bias.detach())
        assert torch.rand(1).item() == after


def test_cpu_quota_is_read_and_respected(monkeypatch):
    """_lib.cpu_quota(): None or the CPUs of the cgroup's bandwidth limit; respect_cpu_quota() never raises the thread count and
    cuts it to the quota (torch ignores container quotas; spinning OpenMP threads get the process throttled)."""
    from music_amd import _lib
    q = _lib.cpu_quota()
    assert q is None or q >= 1
    n0 = torch.get_num_threads()
    try:
        monkeypatch.setattr(_lib, "cpu_quota", lambda: 1)
        _lib.respect_cpu_quota()
        assert torch.get_num_threads() == 1
        monkeypatch.setattr(_lib, "cpu_quota", lambda: 4096)
        _lib.respect_cpu_quota()
        assert torch.get_num_threads() == 1
        torch.set_num_threads(n0)
        monkeypatch.setenv("WN_KEEP_TORCH_THREADS", "1")
        monkeypatch.setattr(_lib, "cpu_quota", lambda: 1)
        _lib.respect_cpu_quota()
        assert torch.get_num_threads() == n0
    finally:
        torch.set_num_threads(n0)


def test_loss_hook_degrades_when_torch_lacks_its_private_hooks(monkeypatch):
    """music_amd/_losshook.py intercepts nn.CrossEntropyLoss through three torch internals; on a torch that lacks one of them the module must
    not fuse (make() -> None: the module then returns a plain tensor and torch's own loss runs) instead of failing inside a backward."""
    from music_amd import _losshook

    class Eng:
        fused_loss_ok = True
    assert _losshook.AVAILABLE is True and _losshook._private_api_present()          # this torch has them
    assert _losshook.make(Eng(), {"gen": 1}, True) is not None
    monkeypatch.setattr(_losshook, "AVAILABLE", False)
    assert _losshook.make(Eng(), {"gen": 1}, True) is None
    out = torch.ones(2, 3, requires_grad=True)
    assert _losshook.wrap(out, None) is out
    monkeypatch.delattr(torch._C, "DisableTorchFunctionSubclass")
    assert _losshook._private_api_present() is False
