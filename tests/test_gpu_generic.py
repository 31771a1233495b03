"""GPU parity of the GENERAL plan (music_amd/engine_generic.py): constructor arguments the specialised kernels do not
cover - filter_width 1, 3, 4, quantization_channels 64 / 100 / 512, up to 160 residual / dilation channels, with and
without biases - against the CPU oracle on gain-scaled weights.  Probabilities and pre-softmax within 1e-3, loss 1e-4,
every gradient within 3e-4 of its tensor's max-abs, through the nn.Module surface (autograd) and the fused step.
Run with -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import wavenet_oracle as wo

LOGIT_TOL = 1e-3
GRAD_RTOL = 3e-4

CASES = [
    # (name, filter_width, dilations, D, R, S, Q, bias, B, extra window)
    ("fw3_small", 3, [1, 2, 4, 1, 2], 24, 20, 40, 256, False, 2, 300),
    ("fw4_bias", 4, [1, 3, 2], 32, 32, 64, 256, True, 1, 517),
    ("fw1", 1, [1, 2], 16, 16, 32, 256, True, 2, 130),
    ("q100", 2, [1, 2, 4, 8], 32, 32, 64, 100, True, 2, 401),
    ("q64_fw3", 3, [2, 1], 48, 40, 72, 64, False, 3, 257),
    ("q512", 2, [1, 2, 4], 32, 32, 96, 512, False, 1, 600),
    ("ch128", 2, [1, 2, 4, 8, 16], 128, 128, 256, 256, False, 2, 700),
    ("ch160_96", 2, [1, 4, 16], 96, 160, 288, 256, True, 1, 333),
]


def _net(case, gain=2.5):
    from music_amd.model import wavenet
    name, fw, dil, D, R, S, Q, bias, B, win = case
    cfg = dict(filter_width=fw, dilations=dil, dilation_channels=D, residual_channels=R, skip_channels=S,
               quantization_channels=Q, use_bias=bias)
    torch.manual_seed(100 + [c[0] for c in CASES + [case]].index(name))
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(gain)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    return net.cuda(), cfg, params


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_general_plan_forward_and_gradients_vs_oracle(case):
    from music_amd.engine_generic import GenericWaveNetEngine
    name, fw, dil, D, R, S, Q, bias, B, win = case
    net, cfg, params = _net(case)
    rf = net.receptive_field
    assert rf == wo.receptive_field(fw, dil)
    T = rf + win - 1
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.standard_normal((B, Q, T)).astype(np.float32) * 0.5)        # arbitrary floats, as the reference accepts
    x[:, :, ::3] = 0
    target = torch.from_numpy(rng.integers(0, Q, size=(B * win,)).astype(np.int64))
    probs = net(x.cuda())
    assert isinstance(net._engine, GenericWaveNetEngine)
    loss = torch.nn.functional.cross_entropy(probs, target.cuda())
    loss.backward()
    eng = net._engine
    W = win
    # the reference gradient takes the device's sign at ReLU pre-activations within 2e-4 of zero (tests/test_gpu_fullsize.py
    # explains why: one flipped element moves a whole gradient tensor by ~1e-3) and insists on equal signs elsewhere
    from music_amd.engine import SLACK
    from tests.test_gpu_fullsize import _device_relu
    wsd = eng.workspace(B, T)
    pitch, lo = wsd["pitch"], rf - 1
    v = lambda buf: buf[SLACK:SLACK + B * eng.SP * pitch].view(B, eng.SP, pitch)[:, :S, lo:T].cpu()
    relu, stats = _device_relu({"skip_sum": v(wsd["U"]), "post_process_1": v(wsd["H"])})
    inter = {}
    l_ref, p_ref, g_ref = wo.loss_and_grads(params, dil, x, target, filter_width=fw, quantization_channels=Q, intermediates=inter,
                                            relu=relu)
    e_p = (probs.detach().cpu() - p_ref).abs().max().item()
    o_dev = eng.workspace(B, T)["O"][:B * Q * W].view(B, Q, W).cpu()
    e_o = (o_dev - inter["pre_softmax"].detach()).abs().max().item()
    assert probs.shape == (B * W, Q) and e_p <= LOGIT_TOL and e_o <= LOGIT_TOL, (e_p, e_o)
    from tests.helpers import nonvacuous
    nonvacuous(p_ref, "general plan " + name, 6.0 / Q)         # at least six times the uniform probability (Q = 64 ... 512)
    assert abs(loss.item() - l_ref.item()) < 1e-4
    worst = 0.0
    gmax = max(g.abs().max().item() for g in g_ref.values())
    for n, p in net.named_parameters():
        want = g_ref[n]
        got = torch.zeros_like(want) if p.grad is None else p.grad.cpu()
        scale = max(want.abs().max().item(), 1e-3 * gmax)
        err = (got - want).abs().max().item() / scale
        worst = max(worst, err)
        assert err <= GRAD_RTOL, (n, err)
    # the fused step: same loss, same gradients (bit-identical forward, the CE kernel instead of torch's)
    l2 = eng.loss_and_grad(x.cuda(), target.cuda(), want_probs=True)
    assert abs(l2.item() - l_ref.item()) < 1e-4
    for n, p in net.named_parameters():
        want = g_ref[n]
        scale = max(want.abs().max().item(), 1e-3 * gmax)
        assert (eng.param_view(n, grad=True).cpu() - want).abs().max().item() / scale <= GRAD_RTOL, n
    g1 = eng.flat_grad.clone()
    eng.loss_and_grad(x.cuda(), target.cuda())
    assert torch.equal(g1, eng.flat_grad)                       # bit-reproducible
    print("%s: fw %d Q %d R/D/S %d/%d/%d bias %d: pre-softmax err %.1e (|max| %.1f), probs err %.1e, worst grad err %.1e" %
          (name, fw, Q, R, D, S, bias, e_o, inter["pre_softmax"].abs().max().item(), e_p, worst))


def test_general_plan_trains_and_generates():
    """The drop-in loops on a shape only the general plan covers (128 channels, Q = 100 is not a mu-law width the decoder
    needs, so 256): three Adam steps reduce the loss; fast_generate's cached-queue decode agrees with the naive forward."""
    from music_amd import fast_generate as fg
    from music_amd.model import predict_next
    case = ("gen128", 2, [1, 2, 4, 8], 96, 128, 160, 256, True, 1, 64)
    net, cfg, params = _net(case, gain=2.0)
    rf = net.receptive_field
    rng = np.random.default_rng(3)
    codes = rng.integers(0, 256, size=(rf + 40,))
    onehot = torch.zeros(1, 256, len(codes))
    onehot[0, torch.from_numpy(codes), torch.arange(len(codes))] = 1.0
    x = onehot.cuda()
    # the cached-queue decoder on this shape (the generic fp32 decode kernel): its first prediction is the naive forward's,
    # and 20 teacher-forced steps follow the oracle's cached recurrence
    pred, st = fg.predict_next(net, x[:, :, :rf].contiguous(), None)
    want = predict_next(net, x[:, :, :rf].contiguous())
    assert int(pred[0]) == int(want[0])
    pred_o, q_o = wo.fast_predict_next(params, cfg["dilations"], onehot[:, :, :rf], None)
    for t in range(rf, rf + 20):
        note = onehot[:, :, t:t + 1]
        pred, st = fg.predict_next(net, note.cuda(), st)
        pred_o, q_o = wo.fast_predict_next(params, cfg["dilations"], note, q_o)
        assert int(pred[0]) == int(pred_o[0]), t
    # training
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    target = torch.from_numpy(rng.integers(0, 256, size=(x.size(2) - rf + 1,)).astype(np.int64)).cuda()
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(net(x), target)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]


def test_cached_queue_decode_on_the_general_plan_with_64_channels():
    """64 / 64 channels but 128 quantisation channels: the model runs the general plan, whose weight packs the matrix-core
    decoder cannot use - fast_generate must fall back to the generic decode kernel, and agree with the oracle."""
    from music_amd import fast_generate as fg
    case = ("gen64_q128", 2, [1, 2, 4], 64, 64, 96, 128, False, 1, 8)
    net, cfg, params = _net(case, gain=2.0)
    rf = net.receptive_field
    rng = np.random.default_rng(5)
    codes = rng.integers(0, 128, size=(rf + 12,))
    onehot = torch.zeros(1, 128, len(codes))
    onehot[0, torch.from_numpy(codes), torch.arange(len(codes))] = 1.0
    pred, st = fg.predict_next(net, onehot[:, :, :rf].cuda(), None)
    pred_o, q_o = wo.fast_predict_next(params, cfg["dilations"], onehot[:, :, :rf], None, quantization_channels=128)
    assert int(pred[0]) == int(pred_o[0])
    for t in range(rf, rf + 12):
        pred, st = fg.predict_next(net, onehot[:, :, t:t + 1].cuda(), st)
        pred_o, q_o = wo.fast_predict_next(params, cfg["dilations"], onehot[:, :, t:t + 1], q_o, quantization_channels=128)
        assert int(pred[0]) == int(pred_o[0]), t


AE_CASES = [
    # (name, filter_width, dilations, en R / D, bottleneck, pool, de R / D / S, Q, bias, B, W)
    ("ae_fw3", 3, [1, 2, 4], 24, 20, 10, 40, 32, 28, 48, 256, False, 2, 320),
    ("ae_q64_bias", 2, [1, 2, 4, 3], 16, 24, 6, 25, 24, 16, 40, 64, True, 2, 150),
    ("ae_ch96_80", 2, [1, 2, 4, 8], 72, 96, 12, 50, 96, 80, 112, 256, True, 1, 400),
    ("ae_fw4_q100", 4, [2, 1], 33, 17, 5, 9, 20, 36, 33, 100, False, 3, 137),
]


@pytest.mark.parametrize("case", AE_CASES, ids=[c[0] for c in AE_CASES])
def test_general_plan_autoencoder_vs_oracle(case):
    """wavenet_autoencoder with constructor arguments the specialised kernels do not cover (model1.py:14-31 takes any): other
    filter widths, quantisation widths, more than 64 channels - through music_amd/ae_generic.py.  Encoding, probabilities,
    loss and EVERY gradient (bias gradients included) against autograd on the CPU oracle with the same per-forward conditioning
    projections; both branches of _conditon occur; the fused step gives the same gradients and is bit-reproducible."""
    from music_amd.model1 import wavenet_autoencoder
    from music_amd.ae_generic import GenericAutoencoderEngine
    from oracle import intops
    name, fw, dil, eR, eD, bw_, pool, dR, dD, dS, Q, bias, B, W = case
    cfg = dict(filter_width=fw, quantization_channel=Q, dilations=dil, en_residual_channel=eR, en_dilation_channel=eD,
               en_bottleneck_width=bw_, en_pool_kernel_size=pool, de_residual_channel=dR, de_dilation_channel=dD,
               de_skip_channel=dS, use_bias=bias)
    torch.manual_seed(200 + len(name))
    net = wavenet_autoencoder(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
        net.connection_2.weight.mul_(6.0)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rf = net.receptive_field
    assert rf == wo.receptive_field(fw, dil)
    rng = np.random.default_rng(9)
    idx = rng.integers(0, Q, size=(B, rf + W - 1))
    x = torch.from_numpy(np.stack([intops.one_hot_proper(r, Q) for r in idx]))
    target = torch.from_numpy(rng.integers(0, Q, size=(B * W,)).astype(np.int64))
    Le = W // pool
    T = rf + W - 1
    L, stretch = T - (fw - 1), []
    for d in dil:
        L -= (fw - 1) * d
        stretch.append(L % Le == 0)
    torch.manual_seed(91)
    net.zero_grad()
    probs = net(x.cuda())
    assert isinstance(net._engine, GenericAutoencoderEngine)
    loss = torch.nn.CrossEntropyLoss()(probs, target.cuda())
    loss.backward()
    torch.manual_seed(91)
    cond = wo.draw_conditioning(len(dil), bw_, dD, dS)
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    p_ref, enc_ref = wo.autoencoder_forward(leaf, dil, x, pool, cond, filter_width=fw, q=Q)
    l_ref = torch.nn.functional.cross_entropy(p_ref, target)
    g_ref = torch.autograd.grad(l_ref, list(leaf.values()), allow_unused=True)
    e_enc = (net.last_encoding.cpu() - enc_ref.detach()).abs().max().item()
    e_p = (probs.detach().cpu() - p_ref.detach()).abs().max().item()
    assert probs.shape == (B * W, Q) and e_enc < 1e-4 and e_p <= LOGIT_TOL, (e_enc, e_p)
    from tests.helpers import nonvacuous
    nonvacuous(p_ref.detach(), "general plan " + name, 6.0 / Q)
    assert abs(loss.item() - l_ref.item()) < 1e-4
    gs = [torch.zeros_like(leaf[n]) if g is None else g for (n, _), g in zip(net.named_parameters(), g_ref)]
    gmax = max(g.abs().max().item() for g in gs)
    worst = 0.0
    for (n, p), want in zip(net.named_parameters(), gs):
        got = torch.zeros_like(want) if p.grad is None else p.grad.cpu()
        err = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-3 * gmax)
        worst = max(worst, err)
        assert err <= GRAD_RTOL, (n, err)
    # the fused step with the same projections: same loss and gradients, bit-reproducible
    eng = net._engine
    l2 = eng.loss_and_grad(x.cuda(), target.cuda(), cond)
    assert abs(l2.item() - l_ref.item()) < 1e-4
    g1 = eng.flat_grad.clone()
    for (n, p), want in zip(net.named_parameters(), gs):
        o = eng.spec.off[n]
        err = (g1[o:o + p.numel()].view(p.shape).cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-3 * gmax)
        assert err <= GRAD_RTOL, (n, err)
    eng.loss_and_grad(x.cuda(), target.cuda(), cond)
    assert torch.equal(g1, eng.flat_grad)
    print("%s: fw %d Q %d encoder %d/%d decoder %d/%d/%d bias %d, %d stretch / %d tile layers: enc err %.1e probs err %.1e worst grad err %.1e" %
          (name, fw, Q, eR, eD, dR, dD, dS, bias, sum(stretch), len(stretch) - sum(stretch), e_enc, e_p, worst))
