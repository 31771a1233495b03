"""What a user of the reference does to the nn.Module besides forward / backward, on the MI355X: copies and pickles of a module that has
already run (its engine holds HIP streams, workspaces and ctypes plans), device round trips, dtype casts, and the gradient with respect
to the INPUT - which autograd gives the reference through its causal nn.Conv1d (wavenet/model.py:104; wavenet_autoencoder/model1.py:137,158)
- for all four host plans (specialised / general, WaveNet / autoencoder), against autograd on the CPU oracle.  Run with -m gpu."""
import copy
import io

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import wavenet_oracle as wo
from tests.helpers import nonvacuous

GRAD_RTOL = 3e-4
WN = dict(filter_width=2, dilations=[1, 2, 4, 8, 32], dilation_channels=32, residual_channels=32, skip_channels=64,
          quantization_channels=256, use_bias=False)
AE = dict(filter_width=2, quantization_channel=256, dilations=[1, 2, 4, 8], en_residual_channel=32, en_dilation_channel=32,
          en_bottleneck_width=16, en_pool_kernel_size=100, de_residual_channel=32, de_dilation_channel=32, de_skip_channel=64, use_bias=False)


def _dense(B, T, seed, q=256):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, q, T, generator=g) * 0.5


def _wavenet(cfg=WN, gain=3.0, seed=0):
    from music_amd.model import wavenet
    torch.manual_seed(seed)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(gain)
    return net


def _autoencoder(cfg=AE, gain=2.0, seed=0):
    from music_amd.model1 import wavenet_autoencoder
    torch.manual_seed(seed)
    net = wavenet_autoencoder(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(gain)
    return net


def test_copies_and_pickles_of_a_module_that_has_run():
    net = _wavenet().cuda()
    x = _dense(2, net.receptive_field + 200, 1).cuda()
    ref = net(x).detach().clone()                            # the engine exists now
    tgt = torch.randint(0, 256, (ref.shape[0],), device="cuda")
    nn.CrossEntropyLoss()(net(x), tgt).backward()
    g_ref = [p.grad.clone() for p in net.parameters()]

    def check(n2, what):
        assert n2._engine is None, what                      # the copy builds its own on first use
        out = n2(x)
        assert torch.equal(out.detach(), ref), what
        n2.zero_grad()
        nn.CrossEntropyLoss()(out, tgt).backward()
        for a, b in zip(n2.parameters(), g_ref):
            assert torch.equal(a.grad, b), what
        assert all(a.data_ptr() != b.data_ptr() for a, b in zip(n2.parameters(), net.parameters())), what   # independent storage
    check(copy.deepcopy(net), "deepcopy")
    buf = io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    check(torch.load(buf, weights_only=False), "torch.save(module)")
    # a device round trip rebuilds the engine on the parameters that came back
    n3 = copy.deepcopy(net).cpu().cuda()
    assert torch.equal(n3(x).detach(), ref)
    # the original is untouched by all of this
    assert torch.equal(net(x).detach(), ref)
    # errors a caller can act on, not a crash inside a kernel
    with pytest.raises(RuntimeError, match="no CPU path"):
        copy.deepcopy(net).cpu()(x.cpu())
    with pytest.raises(RuntimeError, match="different devices"):
        copy.deepcopy(net).cpu()(x)
    with pytest.raises(ValueError, match="not long enough"):
        net(x[:, :, :net.receptive_field - 1])
    for cast in ("half", "double"):
        with pytest.raises(TypeError, match="must be float32"):
            getattr(copy.deepcopy(net), cast)()(x)
    # inputs of another dtype are converted, as the kernels read float32
    assert torch.equal(net(x.double()).detach(), ref)


def test_autoencoder_copies_and_pickles():
    net = _autoencoder().cuda()
    x = _dense(2, net.receptive_field + 300, 2).cuda()
    torch.manual_seed(5)
    ref = net(x).detach().clone()
    for what, n2 in (("deepcopy", copy.deepcopy(net)), ("pickle", None)):
        if n2 is None:
            buf = io.BytesIO()
            torch.save(net, buf)
            buf.seek(0)
            n2 = torch.load(buf, weights_only=False)
        assert n2._engine is None
        torch.manual_seed(5)                                 # the same per-forward conditioning draws
        assert torch.equal(n2(x).detach(), ref), what


@pytest.mark.parametrize("plan", ["specialised", "general_k3", "general_q100"])
def test_input_gradient_of_the_wavenet_vs_oracle(plan):
    cfg = dict(WN)
    if plan == "general_k3":
        cfg.update(filter_width=3, dilations=[1, 2, 4])
    if plan == "general_q100":
        cfg.update(quantization_channels=100)
    q = cfg["quantization_channels"]
    net = _wavenet(cfg)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    T = net.receptive_field + 150
    x = _dense(2, T, 3, q)
    W = T - net.receptive_field + 1
    tgt = torch.randint(0, q, (2 * W,), generator=torch.Generator().manual_seed(4))
    xi = x.clone().cuda().requires_grad_(True)
    out = net(xi)
    loss = nn.CrossEntropyLoss()(out, tgt.cuda())
    loss.backward()
    assert type(net._engine).__name__ == ("WaveNetEngine" if plan == "specialised" else "GenericWaveNetEngine")
    xr = x.clone().requires_grad_(True)
    p_ref = wo.wavenet_forward(params, cfg["dilations"], xr, filter_width=cfg["filter_width"], quantization_channels=q)
    nonvacuous(p_ref.detach(), plan, 0.3)
    l_ref = nn.functional.cross_entropy(p_ref, tgt)
    (g_ref,) = torch.autograd.grad(l_ref, [xr])
    assert abs(loss.item() - l_ref.item()) < 1e-4
    assert xi.grad is not None and xi.grad.shape == x.shape
    err = (xi.grad.cpu() - g_ref).abs().max().item() / g_ref.abs().max().item()
    print("input gradient (%s): relative error %.2e" % (plan, err))
    assert err <= GRAD_RTOL
    # and the parameters' gradients are what they are without it
    net.zero_grad()
    nn.CrossEntropyLoss()(net(x.cuda()), tgt.cuda()).backward()
    g_plain = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    nn.CrossEntropyLoss()(net(x.clone().cuda().requires_grad_(True)), tgt.cuda()).backward()
    assert all(torch.equal(a.grad, b) for a, b in zip(net.parameters(), g_plain))


@pytest.mark.parametrize("plan", ["specialised", "general_k3"])
def test_input_gradient_of_the_autoencoder_vs_oracle(plan):
    cfg = dict(AE)
    if plan == "general_k3":
        cfg.update(filter_width=3, dilations=[1, 2, 4])
    net = _autoencoder(cfg)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    T = net.receptive_field + 299
    x = _dense(2, T, 6)
    W = T - net.receptive_field + 1
    tgt = torch.randint(0, 256, (2 * W,), generator=torch.Generator().manual_seed(7))
    xi = x.clone().cuda().requires_grad_(True)
    torch.manual_seed(9)
    loss = nn.CrossEntropyLoss()(net(xi), tgt.cuda())
    loss.backward()
    assert type(net._engine).__name__ == ("_AutoencoderEngine" if plan == "specialised" else "GenericAutoencoderEngine")
    torch.manual_seed(9)
    cond = wo.draw_conditioning(len(cfg["dilations"]), cfg["en_bottleneck_width"], cfg["de_dilation_channel"], cfg["de_skip_channel"])
    xr = x.clone().requires_grad_(True)
    p_ref, _ = wo.autoencoder_forward(params, cfg["dilations"], xr, cfg["en_pool_kernel_size"], cond, filter_width=cfg["filter_width"])
    l_ref = nn.functional.cross_entropy(p_ref, tgt)
    (g_ref,) = torch.autograd.grad(l_ref, [xr])
    assert abs(loss.item() - l_ref.item()) < 1e-4
    err = (xi.grad.cpu() - g_ref).abs().max().item() / g_ref.abs().max().item()
    print("autoencoder input gradient (%s): relative error %.2e" % (plan, err))
    assert err <= GRAD_RTOL


def test_autograd_and_module_usage_patterns():
    """Accumulating backward passes, retain_graph, torch.autograd.grad, scaled losses, parameter / forward hooks, an optimizer that was
    built before the module moved to the device, re-initialisation and load_state_dict after the engine exists, a side stream, frozen
    models - each against the module's own plain gradient (tools/surface_probe2.py prints the same as a table)."""
    from music_amd.model import wavenet
    cfg = dict(WN, use_bias=True)
    net = _wavenet(cfg)
    opt_before = torch.optim.SGD(net.parameters(), lr=0.1)
    net = net.cuda()
    T = net.receptive_field + 300
    x = _dense(2, T, 1).cuda()
    W = T - net.receptive_field + 1
    tgt = torch.randint(0, 256, (2 * W,), generator=torch.Generator().manual_seed(2)).cuda()
    ce = nn.CrossEntropyLoss()

    def grads():
        return torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    net.zero_grad()
    ce(net(x), tgt).backward()
    g1 = grads()
    net.zero_grad()
    ce(net(x), tgt).backward()
    ce(net(x), tgt).backward()
    assert torch.equal(grads(), 2 * g1)
    net.zero_grad()
    loss = ce(net(x), tgt)
    loss.backward(retain_graph=True)
    loss.backward()
    assert torch.equal(grads(), 2 * g1)
    gs = torch.autograd.grad(ce(net(x), tgt), list(net.parameters()))
    assert torch.equal(torch.cat([a.reshape(-1) for a in gs]), g1)
    net.zero_grad()
    (ce(net(x), tgt) * 0.5).backward()
    assert torch.equal(grads(), 0.5 * g1)
    seen = []
    h = net.causal_layer.weight.register_hook(lambda gr: seen.append(gr.clone()) or gr * 2)
    net.zero_grad()
    ce(net(x), tgt).backward()
    h.remove()
    assert torch.equal(net.causal_layer.weight.grad, 2 * seen[0])
    shapes = []
    h = net.register_forward_hook(lambda m, i, o: shapes.append(tuple(o.shape)))
    ref = net(x).detach().clone()
    h.remove()
    assert shapes == [(2 * W, 256)]
    # the optimizer holds the Parameter objects; the engine re-points their storage, not the objects
    net.zero_grad()
    ce(net(x), tgt).backward()
    w0 = net.post_process_2.weight.detach().clone()
    opt_before.step()
    assert not torch.equal(net.post_process_2.weight.detach(), w0) and not torch.equal(net(x).detach(), ref)
    # in-place re-initialisation and load_state_dict land in the flat buffer the kernels read
    with torch.no_grad():
        net.apply(lambda m: nn.init.normal_(m.weight, std=0.05) if isinstance(m, nn.Conv1d) else None)
    n2 = wavenet(**cfg)
    n2.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
    assert torch.equal(n2.cuda()(x).detach(), net(x).detach())
    sd = {k: torch.randn_like(v) * 0.05 for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    n2 = wavenet(**cfg)
    n2.load_state_dict(sd)
    ref = net(x).detach().clone()
    assert torch.equal(n2.cuda()(x).detach(), ref)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        o = net(x)
        ce(o, tgt).backward()
    torch.cuda.current_stream().wait_stream(s)
    assert torch.equal(o.detach(), ref)
    for p in net.parameters():
        p.requires_grad_(False)
    assert not net(x).requires_grad
