"""Row a14 on the CPU: the autoencoder training harness (music_amd/ae_train.py) and the naive generation loop
(music_amd/ae_generate.py) reproduce what the REFERENCE's own train() / predict_next produced for the same seeds
(tests/golden/g9_ae_harness.json, written by tools/make_golden.py from /root/reference) when their model is the CPU
oracle - i.e. host logic (file formats, step order, RNG consumption order, window update) and oracle are both pinned.
The GPU versions of the same comparisons are tests/test_gpu_harness.py."""
import json
import os
import pickle

import numpy as np
import torch

from tests.helpers import GOLDEN, load_npz, params_from


def g9():
    return json.load(open(os.path.join(GOLDEN, "g9_ae_harness.json")))


def write_g9_run(tmp_path, g, extra=None):
    os.makedirs(tmp_path / "params", exist_ok=True)
    rng = np.random.default_rng(g["data_seed"])
    data = [rng.integers(0, 256, size=(l,)).astype(np.int32) for l in g["data_lens"]]
    pickle.dump(data, open(tmp_path / "np_audio.pkl", "wb"))
    dp = dict(g["dataset_params"], audio_path=str(tmp_path / "np_audio.pkl"))
    tp = dict(g["train_params"], **(extra or {}))
    for n, p in (("model", g["model_params"]), ("dataset", dp), ("train", tp)):
        json.dump(p, open(tmp_path / "params" / (n + "_params.json"), "w"))


def check_g9_logs(tmp_path, g, tol):
    got = open(tmp_path / "log" / "loss_log.log").read()
    want = g["loss_log"]
    gl, wl = got.strip().split("\n"), want.strip().split("\n")
    assert len(gl) == len(wl) and got.endswith("\n")
    for a, b in zip(gl, wl):
        assert a.startswith("Average loss is ") and abs(float(a.split(' ')[-1]) - float(b.split(' ')[-1])) < tol, (a, b)
    assert open(tmp_path / "log" / "store_log.log").read() == g["store_log"]
    assert sorted(os.listdir(tmp_path / "restore")) == g["files"]
    ck = torch.load(tmp_path / "restore" / g["files"][-1])
    assert list(ck.keys()) == g["ckpt_keys"] and [list(v.shape) for v in ck.values()] == g["ckpt_shapes"]
    for v, s in zip(ck.values(), g["ckpt_abs_sum"]):
        assert abs(float(v.double().abs().sum()) - s) <= 2e-3 * max(1.0, s)


def test_g9_train_harness_reproduces_reference_logs(tmp_path, monkeypatch):
    from music_amd import ae_train as A
    from music_amd import faster_audio_data as fad
    from tests.cpu_model import OracleAutoencoder, onehot_oracle
    g = g9()
    write_g9_run(tmp_path, g)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(fad, "onehot_device", onehot_oracle)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)

    def ctor(**kw):
        net = OracleAutoencoder(**kw)
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(g["gain"])
        return net
    monkeypatch.setattr(A, "wavenet_autoencoder", ctor)
    torch.set_num_threads(1)
    torch.manual_seed(0)
    A.train()
    check_g9_logs(tmp_path, g, 1e-6)
    # a second run on the same directories resumes the counter from the log (the reference dies on int("is"))
    assert A._resume_counter("./log/", g["train_params"]["print_every"]) == 14


def test_g9_naive_generation_reproduces_reference_codes():
    from music_amd import ae_generate as G
    from tests.cpu_model import OracleAutoencoder
    g = g9()
    net = OracleAutoencoder(**g["model_params"])
    net.load_state_dict(params_from(load_npz("g9_gen_weights.npz")))
    gen = g["gen"]
    start = torch.zeros(1, 256, len(gen["start"]))
    start[0, torch.tensor(gen["start"]), torch.arange(len(gen["start"]))] = 1.0
    torch.set_num_threads(1)
    orig = torch.Tensor.cuda
    try:
        torch.Tensor.cuda = lambda self, *a, **k: self
        n = len(gen["codes_as_written"])
        assert G.generate_codes_naive(net, start, n, seed=gen["seed0"]) == gen["codes_as_written"]
        assert G.generate_codes_naive(net, start, n, sliding_window=True, window=start.size(2), seed=gen["seed0"]) == gen["codes_sliding"]
    finally:
        torch.Tensor.cuda = orig
    assert gen["window_lens"] == list(range(start.size(2), start.size(2) + n))          # the as-written window grows
