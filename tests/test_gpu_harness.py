"""Row a14 and the generation harnesses on the MI355X, against outputs of the reference itself
(tests/golden/g9_ae_harness.json) and end to end (json -> checkpoint -> codes -> .wav).  Run with -m gpu."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import load_npz, params_from
from tests.test_ae_harness_cpu import check_g9_logs, g9, write_g9_run


@pytest.mark.parametrize("fused", [False, True], ids=["autograd", "fused_step"])
def test_g9_autoencoder_train_reproduces_reference_logs(tmp_path, monkeypatch, fused):
    """music_amd/ae_train.py: train() end to end on the device (loader -> HIP one-hot -> HIP autoencoder with fresh
    conditioning projections per forward -> CrossEntropyLoss -> backward -> Adam) writes the loss_log / store_log /
    checkpoints the reference's own wavenet_autoencoder/train.py wrote for the same seed, data and gain-3 weights."""
    from music_amd import ae_train as A
    from music_amd.model1 import wavenet_autoencoder
    g = g9()
    write_g9_run(tmp_path, g, {"fused_step": fused})
    monkeypatch.chdir(tmp_path)

    def ctor(**kw):
        net = wavenet_autoencoder(**kw)
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(g["gain"])
        return net
    monkeypatch.setattr(A, "wavenet_autoencoder", ctor)
    torch.manual_seed(0)
    A.train()
    check_g9_logs(tmp_path, g, 1e-4)


@pytest.mark.parametrize("sliding", [False, True], ids=["as_written_growing_window", "sliding_window"])
def test_g9_naive_autoencoder_generation_reproduces_reference_codes(sliding):
    """music_amd/ae_generate.generate_codes_naive vs the reference's predict_next (generate.py:13-19) driven by the
    window update of generate.py:55 (as written: the window grows) and by the sliding window it evidently meant."""
    from music_amd import ae_generate as G
    from music_amd.model1 import wavenet_autoencoder
    g = g9()
    gen = g["gen"]
    net = wavenet_autoencoder(**g["model_params"])
    net.load_state_dict(params_from(load_npz("g9_gen_weights.npz")))
    net = net.cuda()
    start = torch.zeros(1, 256, len(gen["start"]))
    start[0, torch.tensor(gen["start"]), torch.arange(len(gen["start"]))] = 1.0
    want = gen["codes_sliding"] if sliding else gen["codes_as_written"]
    got = G.generate_codes_naive(net, start, len(want), sliding_window=sliding, window=start.size(2), seed=gen["seed0"])
    if got != want:
        # an argmax may only flip where the reference's own top two were within fp32 noise of each other
        k = next(i for i, (a, b) in enumerate(zip(got, want)) if a != b)
        assert not sliding and gen["margins_as_written"][k] < 1e-4, (k, got, want)
    else:
        assert got == want


def test_ae_generate_end_to_end_writes_wav(tmp_path, monkeypatch):
    """ae_generate.generate(): ./params/model_params.json -> checkpoint -> naive codes -> mu-law decode -> .wav."""
    from music_amd import ae_generate as G
    from music_amd import ae_train as A
    from music_amd.audio_func import mu_law_decode
    from music_amd.model1 import wavenet_autoencoder
    from scipy.io import wavfile
    g = g9()
    os.makedirs(tmp_path / "params")
    json.dump(g["model_params"], open(tmp_path / "params" / "model_params.json", "w"))
    monkeypatch.chdir(tmp_path)
    net = wavenet_autoencoder(**g["model_params"])
    net.load_state_dict(params_from(load_npz("g9_gen_weights.npz")))
    os.makedirs("restore")
    A.save_model(net, 1, "./restore/")
    gen = g["gen"]
    start = torch.zeros(1, 256, len(gen["start"]))
    start[0, torch.tensor(gen["start"]), torch.arange(len(gen["start"]))] = 1.0
    codes = G.generate("./restore/", "wavenet_autoencoder1.model", "./generate/", "g.wav", start_piece=start, sr=13, duration=2,
                       seed=gen["seed0"])
    assert codes == gen["codes_as_written"][:26]
    sr, audio = wavfile.read(tmp_path / "generate" / "g.wav")
    assert sr == 13 and audio.dtype == np.float32 and audio.shape == (26,)
    want = mu_law_decode(torch.tensor(codes, dtype=torch.int64), 256).cpu().numpy().astype(np.float32)
    np.testing.assert_array_equal(audio, want)
    with pytest.raises(FileNotFoundError):
        G.generate("./restore/", "missing.model", "./generate/", "x.wav", sr=4, duration=1)


def test_fast_generate_end_to_end_writes_wav(tmp_path, monkeypatch):
    """fast_generate.generate() (wavenet/fast_generate.py:144-179): ./params/wavenet_params.json -> checkpoint written by
    train.save_model -> cached-queue decode of duration*sr codes from the class-128 start piece -> .wav; the codes are
    the ones generate_codes gives and the samples are their mu-law decode table entries."""
    from music_amd import fast_generate as fg
    from music_amd import train as T
    from music_amd.audio_func import mu_law_decode
    from music_amd.model import wavenet
    from scipy.io import wavfile
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 8, 16, 32, 1, 2, 4, 8], dilation_channels=64, residual_channels=64,
               skip_channels=256, quantization_channels=256, use_bias=False)
    os.makedirs(tmp_path / "params")
    json.dump(cfg, open(tmp_path / "params" / "wavenet_params.json", "w"))
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(17)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.5)
    os.makedirs("restore")
    T.save_model(net, 3, "./restore/")
    codes = fg.generate("./restore/", "wavenet3.model", "./generate/", "f.wav", sr=400, duration=2)
    assert codes.numel() == 800 and codes.dtype == torch.int64 and len(torch.unique(codes)) > 4
    start = torch.zeros(1, 256, net.receptive_field)
    start[:, 128, :] = 1.0
    assert torch.equal(codes.cpu().view(-1), fg.generate_codes(net.cuda(), start.cuda(), 800).cpu().view(-1))
    sr, audio = wavfile.read(tmp_path / "generate" / "f.wav")
    assert sr == 400 and audio.dtype == np.float32 and audio.shape == (800,)
    tab = load_npz("g5_mulaw.npz")["decode_table"] if "decode_table" in load_npz("g5_mulaw.npz") else None
    want = mu_law_decode(codes, 256).cpu().numpy().astype(np.float32)
    np.testing.assert_array_equal(audio, want)
    if tab is not None:
        np.testing.assert_array_equal(audio, tab[codes.cpu().numpy().reshape(-1)].astype(np.float32))


def test_reference_smoke_script_runs_on_the_module():
    """music_amd/test.py = the reference's wavenet/test.py (:11-70): the SHIPPED model on dense random features of 256 x 32000 with
    27907 targets per item, batch 2, Adam 1e-3, three "consumption" lines per epoch."""
    import math
    from music_amd import test as T
    ds = T.simple_dataset()
    assert len(ds) == 100
    s0 = ds[3]
    assert tuple(s0["feature"].shape) == (256, 32000) and s0["feature"].dtype == torch.float32
    assert tuple(s0["target"].shape) == (27907,) and s0["target"].dtype == torch.int64 and 0 <= int(s0["target"].min()) and int(s0["target"].max()) < 256
    lines = []
    fr = T.test(epochs=2, items=5, num_workers=0, out=lines.append)          # 5 items: the last batch of an epoch holds one
    assert len(lines) == 6
    for k, name in enumerate(["Forward", "Backward", "Optimize"] * 2):
        assert lines[k].startswith(name + " consumption is "), lines
        assert 0.0 < float(lines[k].split(" ")[-1]) < 1.0
    assert abs(sum(fr) - 1.0) < 1e-9 and fr[1] > fr[0] > fr[2]
    net = T.test.last_net
    assert net.receptive_field == 4094 and len(net.dilations) == 40
    # random targets on an untrained model: the loss sits at ln 256, and every parameter has moved and is finite
    assert abs(float(T.test.last_loss) - math.log(256.0)) < 0.05
    torch.manual_seed(0)
    from music_amd.model import wavenet
    fresh = wavenet(**json.load(open(T._PARAMS)))
    dead = "dilation_layer_stack.%d.weight" % (4 * 39 + 2)     # the last block's dense conv feeds nothing (model.py:124-129): zero gradient
    for (n, a), (_, b) in zip(net.state_dict().items(), fresh.state_dict().items()):
        assert torch.isfinite(a).all() and torch.equal(a.cpu(), b) == (n == dead), n
