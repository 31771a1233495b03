"""The fallback kernels that real configurations reach (32 padded channels, x1 modes, other decode shapes) must stay
CORRECT at the config-2 / config-5 shapes too, where a developer switch selects them: the 64-channel ragged-shape parity
case and the fused-step parity test are re-run in a fresh interpreter per switch (the switches are read once per
process).  Run with -m gpu."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    {"WN_PQ_BWD": "0"},            # resblock_bwd_rw_k + chan_gemm_rw_k instead of the one-launch block (what biased / conditioned blocks run)
    {"WN_MS_BWD": "0"},            # resblock_bwd_k + 2 x wgrad_k instead of the two-role block (what 32 padded channels / x1 modes run)
    {"WN_PQ_CHAIN": "0"},          # every one-launch block hands the (P, Q) pair on (no chain walk: what d < 32 and short clips run)
]


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_alternative_paths_stay_correct(env):
    e = dict(os.environ, **env)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_sweep.py"), "-k", "d10_R64 or d2_R64 or d4_R48",
           os.path.join(ROOT, "tests", "test_gpu_parity.py") + "::test_fused_train_step_matches_autograd_path_and_oracle"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("env", [{"WN_AE_FUSED_ENC": "0"}, {"WN_PQ_BWD": "0"}, {"WN_AE_COND_MFMA": "0"}, {"WN_AE_ENC_PQ": "0"}], ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_autoencoder_alternative_paths_stay_correct(env):
    """The autoencoder with its encoder blocks as two channel GEMMs forward and GEMM + weight-gradient launches backward
    (WN_AE_FUSED_ENC=0: the paths 32-channel encoders run) / its decoder blocks on wn_resblock_bwd_ms + the data-gradient
    GEMM (what biased decoders run) / the conditioning bias gathered in the forward block instead of multiplied and its
    gradient as bucket sums of a written [df;dg] (WN_AE_COND_MFMA=0, wn_cond_grad: what more than 32 pooled frames run) /
    the encoder blocks' data gradient as a launch of its own on a written dh (what biased encoders run): the G8 forward
    fixture and the 64-channel backward parity test."""
    e = dict(os.environ, **env)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-k", "g8_autoencoder_forward or autoencoder_backward_64"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


def test_32_channel_kernels_stay_correct_without_clip_pairs():
    """Models with <= 32 channels run even batches as clip pairs on the 64-channel block kernels; WN_PAIR32=0 keeps every
    batch on the 32-channel kernels (what odd batches, biased models and x1 modes run): the ragged-shape cases of both
    kinds against the oracle."""
    e = dict(os.environ, WN_PAIR32="0")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_sweep.py"), "-k", "d5_R32 or d4_R20 or d3_R16 or d7_R32"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


def test_autoencoder_32_channel_kernels_stay_correct_without_clip_pairs():
    """The autoencoder with 32 / 32 padded channels runs even batches as clip pairs on the 64-channel one-launch blocks
    (the G8 configuration does); WN_PAIR32=0 keeps the 32-channel block kernels (what odd batches and biased models run):
    the G8 forward fixture and the backward against the oracle on them."""
    e = dict(os.environ, WN_PAIR32="0")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-k", "g8_autoencoder_forward or autoencoder_backward_vs_oracle"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


def test_decode_generic_kernel_stays_correct():
    """The cached-queue decoder on the generic fp32 kernel (decode_k: what every shape other than 64/64/256/256 runs),
    forced at the config-5 shapes with WN_DEC_MFMA=0: the config-5 oracle test (both queue recurrences), the one-launch
    generation test and the batched / sampling tests."""
    e = dict(os.environ, WN_DEC_MFMA="0")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-k", "(decode or generat) and not eight_per_pair and not full_size"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


def test_decode_without_tap0_ahead_stays_correct():
    """The matrix-core decoder forms the tap-0 half of every block's f / g product a sample ahead (WN_DEC_T0, default 1, when
    its partial sums fit LDS: <= 31 blocks); WN_DEC_T0=0 is the kernel deeper models run: the config-5 oracle tests, the
    batched / sampling tests and the autoencoder's cached generation on it."""
    e = dict(os.environ, WN_DEC_T0="0")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-k", "decode or generat"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("env,expr", [({"WN_DEC_KS": "1"}, "decode or generat"), ({"WN_DEC_KS": "8"}, "decode or generat")],
                         ids=["one_skip_workgroup", "split_skip_parts"])
def test_decode_skip_forms_stay_correct(env, expr):
    """The skip sum + post-processing of a pair of eight utterances runs as S / 64 workgroups (one row tile per wave, the
    S-vectors exchanged as tagged granules) at 512 skip channels and for a single pair at 256, and as ONE workgroup otherwise
    (batches at 256 skip channels; more than 24 pairs per launch at 512).  WN_DEC_KS=1 forces the one-workgroup form, any
    larger value the split form wherever all workgroups of the launch are resident: every decode and generation test against
    the oracle on both."""
    e = dict(os.environ, **env)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-k", expr]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("blocks,skip,bias", [(31, 256, False), (32, 256, False), (32, 512, True), (36, 256, False), (41, 512, True)])
def test_decode_deeper_than_the_tap0_table_matches_the_oracle(blocks, skip, bias):
    """The tap-0 partial sums (4 KB per block) fit beside the rest of the chain's LDS up to 31 blocks; from 32 on they live in
    the pair's hand-off area in global memory (read back a block ahead; `wn_decode_sync_granules` leaves room for them from
    exactly that depth on - round 4's first cut reserved it from 33 and a 32-block model timed out, found by
    tools/fuzz_decode.py --shapes); ids and probabilities against the oracle on both sides of the boundary."""
    import numpy as np
    import torch
    from music_amd import fast_generate as fg
    from music_amd.model import wavenet
    from oracle import intops
    from oracle import wavenet_oracle as wo
    dil = ([1, 2, 4, 8] * 11)[:blocks]
    cfg = dict(filter_width=2, dilations=dil, dilation_channels=64, residual_channels=64, skip_channels=skip,
               quantization_channels=256, use_bias=bias)
    torch.manual_seed(41)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rng = np.random.default_rng(42)
    oh = lambda ix: torch.from_numpy(intops.one_hot_proper(np.atleast_1d(ix)))[None]
    start = rng.integers(0, 256, size=(net.receptive_field,))
    forced = rng.integers(0, 256, size=(12,))
    torch.set_num_threads(8)
    pred_o, q_o = wo.fast_predict_next(params, dil, oh(start), None)
    want = [int(pred_o[0])]
    want_p = []
    for s_ in forced:
        pred_o, q_o, pr = wo.fast_predict_next(params, dil, oh(s_), q_o, return_probs=True)
        want.append(int(pred_o[0]))
        want_p.append(pr.numpy())
    pred, st = fg.predict_next(net, oh(start).cuda(), None)
    nxt = torch.from_numpy(np.concatenate([forced[1:], [0]]).astype(np.int32))
    codes, probs, _ = fg._decode(net, st, oh(forced[0]).reshape(-1).cuda(), len(forced), forced=nxt, want_probs=True)
    assert [int(pred[0])] + codes.cpu().tolist() == want
    assert np.abs(probs.cpu().numpy() - np.stack(want_p)).max() < 1e-4


@pytest.mark.parametrize("bias", [False, True])
def test_forward_epilogue_chains_give_the_same_bits(bias, monkeypatch):
    """The skip product and the two post-processing products run as per-clip-group chains on two streams (engine.epi_chains,
    default 2; read at every call): every clip goes through the same tiles whatever the grouping, so probabilities, loss
    and gradients must be bit-identical for 1 chain, 2, and 3 (ragged groups of a batch of 3)."""
    import numpy as np
    import torch
    from music_amd.model import wavenet
    from tests.helpers import scrambled_input
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 8, 16, 32], dilation_channels=64, residual_channels=64,
               skip_channels=96, quantization_channels=256, use_bias=bias)
    torch.manual_seed(21)
    net = wavenet(**cfg).cuda()
    rng = np.random.default_rng(22)
    T = net.receptive_field + 700
    x = scrambled_input(rng.integers(0, 256, size=(3, T))).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(3 * 701,)).astype(np.int64)).cuda()
    got = {}
    net(x)
    for split in ("1", "2", "3"):
        net._engine.epi_chains = int(split)
        probs = net(x).detach().clone()
        eng = net._engine
        loss = eng.loss_and_grad(x, target)
        torch.cuda.synchronize()
        got[split] = (probs, loss.clone(), eng.flat_grad.clone())
    for split in ("2", "3"):
        for a, b in zip(got["1"], got[split]):
            assert torch.equal(a, b), split


def test_fused_epilogue_equals_the_three_launches(monkeypatch):
    """256 skip / 256 quantisation channels: the skip product, the post-processing and their data gradients run as ONE launch per
    direction (wn_skip_epilogue_fwd / _bwd, engine.epi_fused / epi_fused_bwd; round 6); WN_EPI_FUSED=0 / WN_EPI_FUSED_BWD=0 are the three
    wn_chan_gemm launches each (dZ then on the B-stationary product, and with WN_GEMM_BST=0 on chan_gemm_wide2_k).  Same products, the
    intermediate tiles in the chained k order: probabilities within 2e-6, loss within 1e-6, every gradient within 2e-5 of its max-abs;
    a second run of the fused form reproduces its bits.  Ragged batch of 3 (a partly filled last round: tiles dealt out by dZ passes
    is exercised at the bench geometry by tests/test_gpu_fullsize.py)."""
    import numpy as np
    import torch
    from music_amd.model import wavenet
    from tests.helpers import scrambled_input
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 8, 16, 32], dilation_channels=64, residual_channels=64,
               skip_channels=256, quantization_channels=256, use_bias=False)
    rng = np.random.default_rng(31)
    got = {}
    for tag, env in (("fused", {}), ("fused2", {}), ("three", {"WN_EPI_FUSED": "0", "WN_EPI_FUSED_BWD": "0"}),
                     ("three_wide2", {"WN_EPI_FUSED": "0", "WN_EPI_FUSED_BWD": "0", "WN_GEMM_BST": "0"})):
        for k in ("WN_EPI_FUSED", "WN_EPI_FUSED_BWD", "WN_GEMM_BST"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        torch.manual_seed(33)
        net = wavenet(**cfg)
        with torch.no_grad():
            for p_ in net.parameters():
                p_.mul_(2.5)
        net = net.cuda()
        if tag == "fused":
            T = net.receptive_field + 900
            x = scrambled_input(rng.integers(0, 256, size=(3, T))).cuda()
            target = torch.from_numpy(rng.integers(0, 256, size=(3 * 901,)).astype(np.int64)).cuda()
        probs = net(x).detach().clone()
        eng = net._engine
        assert eng.epi_fused == (tag.startswith("fused")) and eng.epi_fused_bwd == (tag.startswith("fused"))
        loss = eng.loss_and_grad(x, target)
        torch.cuda.synchronize()
        got[tag] = (probs, loss.clone(), eng.flat_grad.clone())
    for a, b in zip(got["fused"], got["fused2"]):
        assert torch.equal(a, b)
    assert torch.equal(got["three"][2], got["three_wide2"][2])          # the B-stationary product is bit-identical to wide2
    for other in ("three",):
        assert (got["fused"][0] - got[other][0]).abs().max().item() <= 2e-6
        assert abs(got["fused"][1].item() - got[other][1].item()) <= 1e-6
        g0, g1 = got["fused"][2], got[other][2]
        eng = net._engine
        worst = 0.0
        for name in eng.param_names:
            o, n = eng.spec.off[name], int(np.prod(eng.spec.shape[name]))
            scale = g1[o:o + n].abs().max().item()
            if scale > 0:
                worst = max(worst, (g0[o:o + n] - g1[o:o + n]).abs().max().item() / scale)
        print("fused vs three launches: probs %.2e, worst gradient %.2e of its max-abs" % ((got["fused"][0] - got[other][0]).abs().max().item(), worst))
        assert worst <= 2e-5


def test_chain_form_equals_pair_form(monkeypatch):
    """Blocks with d % 32 == 0 hand dx on whole (chain walk, Q rows carried in registers); WN_PQ_CHAIN=0 makes every block hand
    the (P, Q) pair on.  Same products per item, another summation order of the weight-gradient slabs and of P + Q + dy:
    loss bit-identical (the forward is the same), every gradient within 2e-5 of its max-abs; a second run of either
    form reproduces its bits.  Shapes: a ragged one with short chains (segments + halo items, chains of 1-3 items at d = 512)
    and a batch of 3 with whole chains per workgroup."""
    import numpy as np
    import torch
    from music_amd.model import wavenet
    for dil, B, extra, seed in (([1, 2, 4, 8, 16, 32, 64, 128, 256, 512], 2, 1700, 5), ([32, 64, 1, 2, 32, 128], 3, 2100, 6),
                                ([1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 2, 16, 600, 7)):
        cfg = dict(filter_width=2, dilations=dil, dilation_channels=64, residual_channels=64, skip_channels=64,
                   quantization_channels=256, use_bias=False)
        torch.manual_seed(seed)
        net = wavenet(**cfg)
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(2.5)
        net = net.cuda()
        eng = net._engine_for(torch.device("cuda", 0))
        rng = np.random.default_rng(seed)
        T = net.receptive_field + extra
        codes = torch.from_numpy(rng.integers(0, 256, size=(B, T)).astype(np.int32)).cuda()
        target = torch.from_numpy(rng.integers(0, 256, size=(B * (extra + 1),)).astype(np.int64)).cuda()
        got = {}
        for form in ("1", "0", "1b", "0b"):                       # b: the same again
            monkeypatch.setenv("WN_PQ_CHAIN", form[0])
            eng._ws.clear()
            loss = eng.loss_and_grad_codes(codes, target)
            torch.cuda.synchronize()
            ws = eng.workspace(B, T)
            assert any(ws["bwd"]["chain"]) == (form[0] == "1")
            got[form] = (loss.clone(), eng.flat_grad.clone())
        assert torch.equal(got["1"][0], got["0"][0])
        assert torch.equal(got["1"][1], got["1b"][1]) and torch.equal(got["0"][1], got["0b"][1])
        worst = 0.0
        for name in eng.param_names:
            o, shp = eng.spec.off[name], eng.spec.shape[name]
            n = int(np.prod(shp))
            a, b = got["1"][1][o:o + n], got["0"][1][o:o + n]
            worst = max(worst, ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item())
        print("chain vs pair form: worst gradient difference %.2e of max-abs (dilations %s, batch %d)" % (worst, dil[:10], B))
        assert worst < 2e-5, worst
    monkeypatch.delenv("WN_PQ_CHAIN")
    eng._ws.clear()


def test_encoder_block_forms_agree(monkeypatch):
    """The autoencoder's encoder block hands dx on as the (P, Q) pair (form 0), whole by the chain walk (form 1: d % 32 == 0, Q rows in
    registers) or whole by the walk over adjacent items (form 2: d < 32, Q rows through LDS).  WN_PQ_CHAIN=0 forces form 0 everywhere,
    WN_ENC_LCH=0 form 0 for d < 32 only.  Same products per item, other summation orders: loss bit-identical, every gradient (the input's
    included) within 2e-5 of its max-abs; each form reproduces its bits.  Shapes: ragged lengths with short chains and halo segments,
    a batch of many short clips (workgroups spanning clips), every small dilation incl. d = 3 and 5 (unaligned LDS reads)."""
    import numpy as np
    import torch
    from music_amd.model1 import wavenet_autoencoder
    for dil, B, extra, seed in (([1, 2, 4, 8, 16, 32, 64, 128], 2, 1500, 3), ([3, 5, 32, 1, 64, 2], 3, 777, 4), ([16, 8, 4, 2, 1, 32], 24, 90, 5)):
        cfg = dict(filter_width=2, quantization_channel=256, dilations=dil, en_residual_channel=64, en_dilation_channel=48,
                   en_bottleneck_width=8, en_pool_kernel_size=25, de_residual_channel=32, de_dilation_channel=32, de_skip_channel=64,
                   use_bias=False)
        torch.manual_seed(seed)
        net = wavenet_autoencoder(**cfg)
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(2.0)
        net = net.cuda()
        g = torch.Generator().manual_seed(seed)
        T = net.receptive_field + extra
        x = (torch.randn(B, 256, T, generator=g) * 0.5).cuda()
        W = T - net.receptive_field + 1
        target = torch.randint(0, 256, (B * W,), generator=g).cuda()
        got = {}
        for tag, env in (("all", {}), ("nolch", {"WN_ENC_LCH": "0"}), ("pair", {"WN_PQ_CHAIN": "0"}), ("all_b", {})):
            for k in ("WN_ENC_LCH", "WN_PQ_CHAIN"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            if net._engine is not None:
                net._engine._ws.clear()
            net.zero_grad()
            xi = x.clone().requires_grad_(True)
            torch.manual_seed(100 + seed)
            loss = torch.nn.CrossEntropyLoss()(net(xi), target)
            loss.backward()
            torch.cuda.synchronize()
            forms = net._engine.workspace(B, T)["bwd"]["enc_chain"]
            want = {"all": [1 if d % 32 == 0 else 2 for d in dil], "nolch": [1 if d % 32 == 0 else 0 for d in dil], "pair": [0] * len(dil)}
            assert [int(f) for f in forms] == want[tag.split("_")[0]], (tag, forms)
            got[tag] = (loss.detach().clone(), [p.grad.clone() for p in net.parameters()] + [xi.grad.clone()])
        assert torch.equal(got["all"][0], got["pair"][0]) and torch.equal(got["all"][0], got["nolch"][0])
        assert all(torch.equal(a, b) for a, b in zip(got["all"][1], got["all_b"][1]))
        for other in ("nolch", "pair"):
            worst = max(((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() for a, b in zip(got["all"][1], got[other][1]))
            print("encoder block, shipped forms vs %s: worst gradient difference %.2e of max-abs (dilations %s, batch %d)" % (other, worst, dil, B))
            assert worst < 2e-5, (other, worst)
    for k in ("WN_ENC_LCH", "WN_PQ_CHAIN"):
        monkeypatch.delenv(k, raising=False)
