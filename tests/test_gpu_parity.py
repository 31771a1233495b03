"""GPU parity tests of the whole hot path: HIP (through the reference-shaped Python surface and the
C ABI) vs the golden vectors from the real reference and vs the CPU oracle.  Run with -m gpu.

Tolerances (BASELINE.json north_star): integers bit-exact; float logits / probabilities within
1e-3 absolute — checked here on GAIN-SCALED weights (at default init a constant 1/256
passes).  Gradients: 3e-4 of the tensor's max-abs."""
from collections import OrderedDict

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import wavenet_oracle as wo
from tests.helpers import g1_input, g1_meta, grads_from, load_npz, nonvacuous, params_from, scrambled_input
from tests.tools_cfg import TINY

LOGIT_TOL = 1e-3
GRAD_RTOL = 3e-4


def build(cfg, params, precision=None):
    from music_amd.model import wavenet
    net = wavenet(**cfg)
    net.load_state_dict(params)
    if precision:
        net.precision = precision
    return net.cuda()


@pytest.mark.parametrize("meta", g1_meta(), ids=lambda m: m["name"])
def test_g1_forward_and_grads(meta):
    d = load_npz("g1_%s.npz" % meta["name"])
    params = params_from(d)
    net = build(meta["cfg"], params)
    x = g1_input(d, meta).cuda()
    target = torch.from_numpy(d["target"]).cuda()
    probs = net(x)
    eng = net._engine
    ws = eng.workspace(x.size(0), x.size(2))
    B, W = ws["B"], ws["W"]
    pre = ws["O"][:B * 256 * W].view(B, 256, W).cpu().numpy()
    assert probs.shape == (B * W, 256)
    if "pre_softmax" in d:
        e_pre = np.abs(pre - d["pre_softmax"]).max()
        e_p = np.abs(probs.detach().cpu().numpy() - d["probs"]).max()
        scale = np.abs(d["pre_softmax"]).max()
    else:
        e_pre = np.abs(pre[:, :, ::53] - d["pre_softmax_cols"]).max()
        e_p = np.abs(probs.detach().cpu().numpy()[d["rows"]] - d["probs_rows"]).max()
        scale = np.abs(d["pre_softmax_cols"]).max()
    print(meta["name"], "pre-softmax err %.3e (|max| %.3f)  probs err %.3e" % (e_pre, scale, e_p))
    assert e_pre <= LOGIT_TOL and e_p <= LOGIT_TOL
    if meta["gain"] > 1:            # (the one default-init fixture, gain 1, is the reference's own flat output: kept for the plumbing)
        nonvacuous(d["probs"] if "probs" in d else d["probs_rows"], "G1 " + meta["name"], 0.5)
    loss = torch.nn.CrossEntropyLoss()(probs, target)
    assert abs(loss.item() - float(d["loss"])) < 1e-4
    loss.backward()
    worst = 0.0
    for (name, p) in net.named_parameters():
        g = d["g:" + name]
        scale = max(np.abs(g).max(), 1e-12)
        err = np.abs(p.grad.cpu().numpy() - g).max() / scale
        worst = max(worst, err)
        assert err <= GRAD_RTOL, (name, err)
    print(meta["name"], "worst relative grad err %.3e" % worst)


def test_g2_per_layer_activations():
    d = load_npz("g2_layers.npz")
    net = build(TINY, params_from(d))
    x = scrambled_input(d["idx"]).cuda()
    probs = net(x)
    eng = net._engine
    ws = eng.workspace(1, x.size(2))
    from music_amd.engine import SLACK
    pitch, CH, T = ws["pitch"], eng.CH, x.size(2)
    X = ws["X"][SLACK:SLACK + (eng.N + 1) * CH * pitch].view(eng.N + 1, CH, pitch).cpu().numpy()
    Z = ws["Z"][SLACK:SLACK + eng.N * CH * pitch].view(eng.N, CH, pitch).cpu().numpy()
    e0 = np.abs(X[0, :16, 1:T] - d["x0"][0]).max()
    assert e0 < 1e-5, e0
    W = T - eng.rf + 1
    for i in range(eng.N):
        zi = d["z%d" % i][0]                                  # (16, L_{i+1})
        ez = np.abs(Z[i, :16, T - W:T] - zi[:, -W:]).max()
        xi = d["dense%d" % i][0]
        assert ez < 2e-5, (i, ez)
        if i < eng.N - 1:
            lo = eng.off[i + 1]
            dense = X[i + 1, :16, lo:T] - X[i, :16, lo:T]
            ex = np.abs(dense - xi).max()
            assert ex < 5e-5, (i, ex)
    assert np.abs(probs.detach().cpu().numpy() - d["probs"]).max() < 1e-5


def test_precision_modes_fast_path():
    """The plain 16-bit modes are available but are NOT the parity-grade default: they must still
    be close on a well-conditioned case."""
    meta = [m for m in g1_meta() if m["name"] == "tiny_s0_g1_w1"][0]
    d = load_npz("g1_%s.npz" % meta["name"])
    for prec in (("f16x1", "bf16x1"), ("bf16x3", "bf16x3")):
        net = build(meta["cfg"], params_from(d), prec)
        probs = net(g1_input(d, meta).cuda())
        assert np.abs(probs.detach().cpu().numpy() - d["probs"]).max() < 1e-4


def test_module_surface():
    from music_amd.model import wavenet, predict_next
    meta = g1_meta()[1]
    d = load_npz("g1_%s.npz" % meta["name"])
    params = params_from(d)
    net = wavenet(**meta["cfg"])
    assert list(net.state_dict().keys()) == list(params.keys())
    assert all(tuple(net.state_dict()[k].shape) == tuple(v.shape) for k, v in params.items())
    net.load_state_dict(params)
    net = net.cuda()
    assert net.receptive_field == int(d["rf"]) == net.calc_receptive_field()
    with pytest.raises(ValueError, match="wave sample not long enough"):
        net(torch.zeros(1, 256, net.receptive_field - 1, device="cuda"))
    with pytest.raises(RuntimeError, match="MI355X"):
        net(torch.zeros(1, 256, net.receptive_field))           # CPU input: no CPU path
    x = g1_input(d, meta)
    with torch.no_grad():
        pred = predict_next(net, x[:1].cuda())
    want = wo.predict_next_naive(params, meta["cfg"]["dilations"], x[:1])
    assert pred.dtype == torch.int64 and pred.shape == (1,) and int(pred[0]) == int(want[0])
    # state_dict round trip after the parameters were aliased onto the flat buffer
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    for k, v in params.items():
        assert torch.equal(sd[k], v)


def test_fused_train_step_matches_autograd_path_and_oracle():
    meta = [m for m in g1_meta() if m["name"] == "tiny_s0_g3_w130"][0]
    d = load_npz("g1_%s.npz" % meta["name"])
    params = params_from(d)
    net = build(meta["cfg"], params)
    x = g1_input(d, meta).cuda()
    target = torch.from_numpy(d["target"]).cuda()
    net(x)                                                      # creates the engine
    eng = net._engine
    loss = eng.loss_and_grad(x, target)
    assert abs(loss.item() - float(d["loss"])) < 1e-4
    for name in eng.param_names:
        g = d["g:" + name]
        err = np.abs(eng.param_view(name, grad=True).cpu().numpy() - g).max() / max(np.abs(g).max(), 1e-12)
        assert err <= GRAD_RTOL, (name, err)
    # three Adam steps vs torch.optim.Adam on the oracle
    ref = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam(list(ref.values()), lr=1e-3)
    eng.adam_init(lr=1e-3)
    xc, tc = x.cpu(), target.cpu()
    for step in range(3):
        lg = eng.loss_and_grad(x, target)
        eng.adam_step()
        opt.zero_grad()
        lr = wo.ce_on_probs(wo.wavenet_forward(ref, meta["cfg"]["dilations"], xc), tc)
        lr.backward()
        opt.step()
        assert abs(lg.item() - lr.item()) < 2e-4, (step, lg.item(), lr.item())
    for name in eng.param_names:
        if name in ("dilation_layer_stack.%d.weight" % (4 * (eng.N - 1) + 2),):
            continue
        a, b = eng.param_view(name).cpu(), ref[name].detach()
        assert (a - b).abs().max().item() < 5e-3 * max(1e-3, b.abs().max().item()), name


def test_flat_adam_is_torch_adam_on_the_drop_in_surface():
    """train.get_optimizer(net, 'adam', lr, momentum) returns a torch.optim.Adam whose step() is one wn_adam_flat launch on
    the flat parameter / gradient buffers (wavenet/train.py:39-42 builds optim.Adam(model.parameters(), lr)): the reference's
    loop (zero_grad, net(x), CrossEntropyLoss, backward, step) gives the same parameters as torch.optim.Adam stepping the
    same module, the fast path is the one that runs, state_dict() loads into a torch.optim.Adam and back, a step with
    foreign gradients (not one flat tensor) and a step with a closure fall through to torch's own path on the same state."""
    from music_amd import train as T
    from music_amd.model import wavenet
    meta = [m for m in g1_meta() if m["name"] == "tiny_s0_g3_w130"][0]
    d = load_npz("g1_%s.npz" % meta["name"])
    x = g1_input(d, meta).cuda()
    target = torch.from_numpy(d["target"]).cuda()
    ce = torch.nn.CrossEntropyLoss()
    nets, opts = [], []
    for kind in ("flat", "torch"):
        net = build(meta["cfg"], params_from(d))
        opt = T.get_optimizer(net, "adam", 1e-3, 0.9) if kind == "flat" else torch.optim.Adam(net.parameters(), lr=1e-3)
        nets.append(net)
        opts.append(opt)
    assert isinstance(opts[0], torch.optim.Adam) and type(opts[0]).__name__ == "FlatAdam"
    calls = []
    from music_amd import _lib
    real_call = _lib.call

    def spy(name, *a):
        calls.append(name)
        return real_call(name, *a)
    _lib.call = spy
    try:
        for step in range(4):
            for net, opt in zip(nets, opts):
                opt.zero_grad()
                loss = ce(net(x), target)
                loss.backward()
                if step == 2 and opt is opts[0]:
                    for p in net.parameters():              # foreign gradient tensors: torch's per-tensor path, same state
                        p.grad = p.grad.clone()
                opt.step()
    finally:
        _lib.call = real_call
    assert calls.count("wn_adam_flat") == 3                    # steps 0, 1, 3 of the flat optimizer
    for (n, a), (_, b) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        # (four float32 Adam updates of lr = 1e-3: two correct implementations differ by a few ulp of a step)
        assert (a - b).abs().max().item() <= 4e-6 * max(1.0, b.abs().max().item()), n
    sd = opts[0].state_dict()
    assert float(sd["state"][0]["step"]) == 4.0
    plain = torch.optim.Adam(nets[1].parameters(), lr=1e-3)
    plain.load_state_dict(sd)                                   # torch's Adam takes it ...
    opts[0].load_state_dict(opts[1].state_dict())               # ... and the flat one takes torch's
    opts[0].zero_grad()
    ce(nets[0](x), target).backward()
    opts[0].step()
    assert float(opts[0].state_dict()["state"][0]["step"]) == 5.0
    m0 = opts[0].state_dict()["state"][0]["exp_avg"]
    assert m0.shape == next(nets[0].parameters()).shape and torch.isfinite(m0).all()


@pytest.mark.parametrize("kind,momentum", [("sgd", 0.9), ("sgd", 0.0), ("rmsprop", 0.9), ("rmsprop", 0.0)])
def test_flat_sgd_and_rmsprop_are_torchs_on_the_drop_in_surface(kind, momentum):
    """train.get_optimizer(net, 'sgd' | 'rmsprop', lr, momentum) (wavenet/train.py:28-38) returns torch's own optimizer class whose
    step() is ONE wn_sgd_flat / wn_rmsprop_flat launch on the flat buffers: over four steps of the reference's loop the parameters
    equal those of torch.optim.SGD / RMSprop stepping the same module (a few ulp of a step), the fast path is the one that runs,
    a step with foreign gradient tensors falls through to torch's path on the SAME state, and state_dict() interchanges."""
    from music_amd import train as T
    from music_amd import _lib
    meta = [m for m in g1_meta() if m["name"] == "tiny_s0_g3_w130"][0]
    d = load_npz("g1_%s.npz" % meta["name"])
    x = g1_input(d, meta).cuda()
    target = torch.from_numpy(d["target"]).cuda()
    ce = torch.nn.CrossEntropyLoss()
    cls = torch.optim.SGD if kind == "sgd" else torch.optim.RMSprop
    lr = 1e-2 if kind == "sgd" else 1e-3
    nets, opts = [], []
    for flat in (True, False):
        net = build(meta["cfg"], params_from(d))
        nets.append(net)
        opts.append(T.get_optimizer(net, kind, lr, momentum) if flat else cls(net.parameters(), lr=lr, momentum=momentum))
    assert isinstance(opts[0], cls) and type(opts[0]).__name__ == ("FlatSGD" if kind == "sgd" else "FlatRMSprop")
    calls, real_call = [], _lib.call

    def spy(name, *a):
        calls.append(name)
        return real_call(name, *a)
    _lib.call = spy
    def both_step(step):
        # the SAME gradients go to both optimizers (the flat module's own; the torch-stepped module gets copies): RMSprop divides
        # by sqrt(E[g^2]), so an element whose gradient is rounding noise moves by a full lr either way and two separately trained
        # replicas drift apart by construction - what is compared is the UPDATE arithmetic on equal inputs
        opts[0].zero_grad()
        ce(nets[0](x), target).backward()
        for p0, p1 in zip(nets[0].parameters(), nets[1].parameters()):
            p1.grad = None if p0.grad is None else p0.grad.clone()
        if step == 2:
            for p in nets[0].parameters():                  # foreign gradient tensors: torch's per-tensor path, same state
                p.grad = p.grad.clone()
        for opt in opts:
            opt.step()
    try:
        for step in range(4):
            both_step(step)
    finally:
        _lib.call = real_call
    assert calls.count("wn_sgd_flat" if kind == "sgd" else "wn_rmsprop_flat") == 3       # steps 0, 1, 3
    for (n, a), (_, b) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        assert (a - b).abs().max().item() <= 4e-6 * max(1.0, b.abs().max().item()), n
    # the state is torch's: it loads into the plain optimizer and back, and the next flat step continues from it
    sd = opts[0].state_dict()
    plain = cls(nets[1].parameters(), lr=lr, momentum=momentum)
    plain.load_state_dict(sd)
    opts[0].load_state_dict(opts[1].state_dict())
    both_step(4)
    for (n, a), (_, b) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        assert (a - b).abs().max().item() <= 6e-6 * max(1.0, b.abs().max().item()), n
    if kind == "rmsprop":
        assert float(opts[0].state_dict()["state"][0]["step"]) == 5.0


def test_full_size_c2_properties():
    """BASELINE config 2 (30 layers, 64/64/256, batch 8 x 16000): size-independent properties."""
    from music_amd.model import wavenet
    cfg = dict(filter_width=2, dilations=[2 ** i for i in range(10)] * 3, dilation_channels=64,
               residual_channels=64, skip_channels=256, quantization_channels=256, use_bias=False)
    torch.manual_seed(0)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)                                           # non-vacuous logits
    net = net.cuda()
    assert net.receptive_field == 3071
    rng = np.random.default_rng(1234)
    codes = torch.from_numpy(rng.integers(0, 256, size=(8, 16000)).astype(np.int32)).cuda()
    net(torch.zeros(1, 256, 3071, device="cuda"))
    eng = net._engine
    x = eng.onehot(codes, scrambled=True)
    assert x.sum().item() == 8 * 16000
    with torch.no_grad():
        p1 = net(x)
        assert p1.shape == (8 * 12930, 256)
        assert torch.isfinite(p1).all()
        assert (p1.sum(1) - 1).abs().max().item() < 1e-5 and p1.min().item() >= 0
        p2 = net(x)
        assert torch.equal(p1, p2)                               # forward is bit-deterministic
        # clips are independent (rows never cross clips, Q2): a sub-batch gives the same rows
        p3 = net(eng.onehot(codes[2:5].contiguous(), scrambled=True))     # (same code-aware path as the full batch)
        assert torch.equal(p3, p1[2 * 12930:5 * 12930])
        assert (net(x[2:5].contiguous()) - p3).abs().max().item() < 1e-5  # the dense path on the same values
        # causality / receptive field: the first output row block depends only on the first rf samples
        x4 = x[:1, :, :3071 + 255].contiguous()
        p4 = net(x4)
    # compare a window against the CPU oracle (one clip, 600 samples of output)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    xs = x[3:4, :, 5000:5000 + 3071 + 599].contiguous()
    with torch.no_grad():
        got = net(xs).cpu()
        want = wo.wavenet_forward(sd, cfg["dilations"], xs.cpu())
    err = (got - want).abs().max().item()
    print("c2 window probs err %.3e" % err)
    assert err < LOGIT_TOL
    nonvacuous(want, "c2 window at 8 x 16000")
    assert p4.shape == (256, 256)


@pytest.mark.parametrize("tag,correct", [("asis", False), ("fixed", True)])
def test_g6_fast_generate(tag, correct):
    """Cached-queue decode (persistent kernel) vs the reference's fast_generate.predict_next:
    argmax ids bit-exact, queues within float tolerance; as-written (Q5) and corrected recurrences."""
    from music_amd import fast_generate as fg
    from oracle import intops
    d = load_npz("g6_fastgen.npz")
    dil = [int(v) for v in d["dilations"]]
    cfg = dict(TINY, dilations=dil)
    net = build(cfg, params_from(d))

    def onehot(ix):
        return torch.from_numpy(intops.one_hot_proper(np.atleast_1d(ix)))[None].cuda()

    forced = d["forced"]
    # (a) one persistent launch, teacher forced
    pred, st = fg.predict_next(net, onehot(d["start"]), None)
    assert pred.dtype == torch.int64 and pred.shape == (1,)
    preds = [int(pred[0])]
    assert list(st.keys())[:2] == ["causal_layer", "block_1"] and len(st) == len(dil) + 1
    np.testing.assert_array_equal(st["causal_layer"].cpu().numpy(), d["%s_init_causal" % tag])
    for i in range(len(dil)):
        q = st["block_%d" % (i + 1)]
        assert tuple(q.shape) == (1, 16, dil[i])
        np.testing.assert_allclose(q.cpu().numpy(), d["%s_init_block%d" % (tag, i + 1)], atol=2e-5, rtol=0)
    note0 = onehot(forced[0]).reshape(-1)
    nxt = torch.from_numpy(np.concatenate([forced[1:], [0]]).astype(np.int32))
    codes, probs, _ = fg._decode(net, st, note0, len(forced), forced=nxt, want_probs=True, correct_queue=correct)
    preds += codes.cpu().tolist()
    assert preds == list(d["%s_preds" % tag])
    assert abs(probs.sum(1).cpu().numpy() - 1).max() < 1e-5
    for i in range(len(dil)):
        np.testing.assert_allclose(st["block_%d" % (i + 1)].cpu().numpy(), d["%s_final_block%d" % (tag, i + 1)],
                                   atol=1e-4, rtol=0)
    np.testing.assert_array_equal(st["causal_layer"].cpu().numpy(), d["%s_final_causal" % tag])
    if correct:
        assert preds == list(d["naive_preds"])
    # (b) the per-sample API, 12 steps
    pred, st = fg.predict_next(net, onehot(d["start"]), None)
    preds = [int(pred[0])]
    for s in forced[:12]:
        pred, st = fg.predict_next(net, onehot(s), st, correct_queue=correct)
        preds.append(int(pred[0]))
    assert preds == list(d["%s_preds" % tag][:13])
    # (c) free-running greedy generation: init + ONE launch
    free = fg.generate_codes(net, onehot(d["start"]), 49, correct_queue=correct)
    assert free.cpu().tolist() == list(d["%s_free" % tag])
    # (d) a reference-style queue dict (plain tensors) is accepted too
    pred, st = fg.predict_next(net, onehot(d["start"]), None)
    plain = OrderedDict((k, v.cpu()) for k, v in st.items())
    p1, _ = fg.predict_next(net, onehot(forced[0]), plain, correct_queue=correct)
    assert int(p1[0]) == int(d["%s_preds" % tag][1])


def test_g8_autoencoder_forward():
    """wavenet_autoencoder/model1.py forward (encoder, pooled encoding, conditioned decoder with the
    per-forward random projections, both _conditon branches) vs the reference's own outputs."""
    import json
    import os
    from music_amd.model1 import wavenet_autoencoder
    from oracle import intops
    from tests.helpers import GOLDEN
    d = load_npz("g8_autoencoder.npz")
    cfg = json.load(open(os.path.join(GOLDEN, "g8_cfg.json")))
    params = params_from(d)
    net = wavenet_autoencoder(**cfg)
    assert list(net.state_dict().keys()) == list(params.keys())
    net.load_state_dict(params)
    net = net.cuda()
    assert net.receptive_field == int(d["rf"])
    for tag in ("a", "b"):
        idx = d[tag + "_idx"]
        x = torch.from_numpy(np.stack([intops.one_hot_proper(r) for r in idx])).cuda()
        torch.manual_seed(int(d[tag + "_fwd_seed"]))
        probs = net(x)
        e_enc = np.abs(net.last_encoding.cpu().numpy() - d[tag + "_enc"]).max()
        probs = probs.detach()
        e_p = np.abs(probs.cpu().numpy() - d[tag + "_probs"]).max()
        print("autoencoder", tag, "enc err %.2e probs err %.2e" % (e_enc, e_p))
        assert e_enc < 1e-4 and e_p < LOGIT_TOL
        nonvacuous(d[tag + "_probs"], "G8 " + tag)
        assert probs.shape == d[tag + "_probs"].shape
        torch.manual_seed(int(d[tag + "_fwd_seed"]) + 1)                  # other projections -> other output
        assert np.abs(net(x).detach().cpu().numpy() - d[tag + "_probs"]).max() > 1e-6
    with pytest.raises(ValueError):
        net(torch.zeros(1, 256, net.receptive_field - 1, device="cuda"))


def test_grads_64_channels_vs_oracle():
    """The 64-channel kernel instantiations (BASELINE config-2 width) on a 7-block stack with
    tiles that straddle the 512-column workgroup boundary: loss and every gradient vs the oracle."""
    from music_amd.model import wavenet
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 8, 16, 512, 3], dilation_channels=64, residual_channels=64,
               skip_channels=96, quantization_channels=256, use_bias=False)
    torch.manual_seed(11)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.5)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rng = np.random.default_rng(12)
    T = net.receptive_field + 1100
    x = scrambled_input(rng.integers(0, 256, size=(2, T)))
    target = torch.from_numpy(rng.integers(0, 256, size=(2 * 1101,)).astype(np.int64))
    net(x[:, :, :net.receptive_field].cuda())
    eng = net._engine
    loss = eng.loss_and_grad(x.cuda(), target.cuda())
    l_ref, p_ref, g_ref = wo.loss_and_grads(params, cfg["dilations"], x, target)
    assert abs(loss.item() - l_ref.item()) < 1e-4
    worst = 0.0
    for name in eng.param_names:
        g = g_ref[name]
        err = (eng.param_view(name, grad=True).cpu() - g).abs().max().item() / max(g.abs().max().item(), 1e-12)
        worst = max(worst, err)
        assert err <= GRAD_RTOL, (name, err)
    print("64-channel grads: worst relative err %.2e" % worst)
    # weight gradients are bit-reproducible (slab reduction, no float atomics)
    g1 = eng.flat_grad.clone()
    eng.loss_and_grad(x.cuda(), target.cuda())
    assert torch.equal(g1, eng.flat_grad)


def test_saturated_gates_give_finite_gradients_vs_oracle():
    """One block whose filter and gate convs are 40 x / 150 x larger than the rest: a few percent of its pre-activations have f <= -15
    AND g <= -59, where tanh is -1 and the sigmoid 0 to every bit.  The one-launch backward block forms both from one reciprocal of
    (1 + e^-2f)(1 + e^-g); that product must not overflow (round 6: it did - NaN gradients in the whole model, found at config 4's
    geometry with every weight x 2.5).  Loss and every gradient against the oracle, all finite."""
    from music_amd.model import wavenet
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 32, 3], dilation_channels=64, residual_channels=64,
               skip_channels=256, quantization_channels=256, use_bias=False)
    torch.manual_seed(21)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.5)
        for k, gain in ((4, 40.0), (5, 150.0)):                # filter / gate conv of block 1
            dict(net.named_parameters())["dilation_layer_stack.%d.weight" % k].mul_(gain)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rng = np.random.default_rng(22)
    T = net.receptive_field + 700
    x = scrambled_input(rng.integers(0, 256, size=(2, T)))
    target = torch.from_numpy(rng.integers(0, 256, size=(2 * 701,)).astype(np.int64))
    eng = net._engine_for(torch.device("cuda", 0))
    loss = eng.loss_and_grad(x.cuda(), target.cuda())
    assert eng.workspace(2, T)["pq"], "this test is about the one-launch backward block"
    assert torch.isfinite(eng.flat_grad).all(), "non-finite gradients: %d" % int((~torch.isfinite(eng.flat_grad)).sum())
    # judged like the full-size cases (tests/test_gpu_fullsize.py): float64 oracle with the device's sign at post-processing ReLU pre-activations
    # within 2e-4 of zero - the gradients of this model are tiny (most of block 1's paths are shut), and ONE such sign moves them by percent
    from tests.test_gpu_fullsize import _oracle_grads_f32_f64, _c2_dev_pre, _check_grads
    l32, p32, e32, l64, g64 = _oracle_grads_f32_f64(params, cfg["dilations"], x, target, _c2_dev_pre(eng, eng.workspace(2, T)))
    assert abs(loss.item() - l64.item()) < 1e-4
    worst = _check_grads({n: eng.param_view(n, grad=True) for n in eng.param_names}, g64, e32, rtol=GRAD_RTOL)
    print("saturated gates: worst relative gradient err %.2e (%s; the float32 CPU path there: %.2e)" % worst)


@pytest.mark.parametrize("bias", [False, True])
def test_grads_64_channels_channel_split_block(bias):
    """The channel-split backward block (wn_resblock_bwd_ms: both weight gradients inside the block
    launch, df/dg/z transposed on the matrix core): loss and every gradient vs the oracle, on tiles
    that straddle the 64-column steps and the workgroups' item runs, with a dilation that is not a
    multiple of 4 (unaligned shifted tap) and ragged real channel counts inside the 64 padded ones."""
    from music_amd.model import wavenet
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 8, 16, 512, 3], dilation_channels=56 if bias else 64,
               residual_channels=64 if bias else 60, skip_channels=96, quantization_channels=256, use_bias=bias)
    torch.manual_seed(21)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.5)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rng = np.random.default_rng(22)
    T = net.receptive_field + 1100
    x = scrambled_input(rng.integers(0, 256, size=(3, T)))
    target = torch.from_numpy(rng.integers(0, 256, size=(3 * 1101,)).astype(np.int64))
    net(x[:, :, :net.receptive_field].cuda())
    eng = net._engine
    eng.ms_bwd = True
    eng._ws.clear()                      # the backward workspace plan depends on the flag
    loss = eng.loss_and_grad(x.cuda(), target.cuda())
    l_ref, p_ref, g_ref = wo.loss_and_grads(params, cfg["dilations"], x, target)
    assert abs(loss.item() - l_ref.item()) < 1e-4
    floor = 1e-3 * max(g.abs().max().item() for g in g_ref.values())
    worst = 0.0
    for name in eng.param_names:
        g = g_ref[name]
        err = (eng.param_view(name, grad=True).cpu() - g).abs().max().item() / max(g.abs().max().item(), floor)
        worst = max(worst, err)
        assert err <= GRAD_RTOL, (name, err)
    print("channel-split block (bias=%s): worst relative grad err %.2e" % (bias, worst))
    g1 = eng.flat_grad.clone()
    eng.loss_and_grad(x.cuda(), target.cuda())
    assert torch.equal(g1, eng.flat_grad)            # still bit-reproducible


@pytest.mark.parametrize("bias", [False, True], ids=["nobias", "bias"])
def test_decode_config5_vs_oracle(bias):
    """BASELINE config 5 shape (30 blocks, 64/64/256): the persistent decode kernel (matrix-core pair of workgroups;
    with biases too) vs the oracle's cached-queue recurrence - argmax ids exact, probabilities within 1e-4."""
    from music_amd import fast_generate as fg
    from music_amd.model import wavenet
    from oracle import intops
    cfg = dict(filter_width=2, dilations=[2 ** i for i in range(10)] * 3, dilation_channels=64,
               residual_channels=64, skip_channels=256, quantization_channels=256, use_bias=bias)
    torch.manual_seed(21)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.2)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rng = np.random.default_rng(22)
    start = rng.integers(0, 256, size=(net.receptive_field,))
    forced = rng.integers(0, 256, size=(20,))

    def onehot(ix):
        return torch.from_numpy(intops.one_hot_proper(np.atleast_1d(ix)))[None]

    torch.set_num_threads(8)
    for correct in (False, True):
        pred_o, q_o, p0 = wo.fast_predict_next(params, cfg["dilations"], onehot(start), None, return_probs=True)
        want, want_p = [int(pred_o[0])], []
        for s in forced:
            pred_o, q_o, pr = wo.fast_predict_next(params, cfg["dilations"], onehot(s), q_o, correct_queue=correct,
                                                  return_probs=True)
            want.append(int(pred_o[0]))
            want_p.append(pr.numpy())
        pred, st = fg.predict_next(net, onehot(start).cuda(), None)
        got = [int(pred[0])]
        nxt = torch.from_numpy(np.concatenate([forced[1:], [0]]).astype(np.int32))
        codes, probs, _ = fg._decode(net, st, onehot(forced[0]).reshape(-1).cuda(), len(forced), forced=nxt,
                                     want_probs=True, correct_queue=correct)
        got += codes.cpu().tolist()
        err = np.abs(probs.cpu().numpy() - np.stack(want_p)).max()
        print("config-5 decode (correct_queue=%s): probs err %.2e" % (correct, err))
        assert got == want and err < 1e-4
        np.testing.assert_allclose(st["block_30"].cpu().numpy(), q_o["block_30"].numpy(), atol=1e-4, rtol=0)


@pytest.mark.parametrize("shape", [(32, 32, 512, False, 12), (24, 20, 512, True, 12), (48, 64, 512, False, 6), (17, 33, 256, True, 36)],
                         ids=lambda s: "R%d_D%d_S%d_%s_N%d" % (s[0], s[1], s[2], "bias" if s[3] else "nobias", s[4]))
def test_decode_padded_and_512_skip_channels_vs_oracle(shape):
    """The matrix-core decode kernels at the reference's SHIPPED widths (32 / 32 / 512, wavenet_params.json) and other models
    with at most 64 residual / dilation channels: fewer than 64 are padded to 64 with zero rows and columns (64-wide queue
    columns inside; the state a caller indexes keeps the reference's shapes), 512 skip channels run the S = 512 form of the
    skip / post-processing workgroup; 36 blocks: without the tap-0 partial sums.  Both queue recurrences against the oracle:
    argmax ids exact, probabilities within 1e-4, the last block's queue as the reference would hold it."""
    from music_amd import fast_generate as fg
    from music_amd.model import wavenet
    from oracle import intops
    R, D, S, bias, N = shape
    dil = ([1, 2, 4, 8, 16, 32] * 6)[:N]
    cfg = dict(filter_width=2, dilations=dil, dilation_channels=D, residual_channels=R, skip_channels=S,
               quantization_channels=256, use_bias=bias)
    torch.manual_seed(61)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.2)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rng = np.random.default_rng(62)
    start = rng.integers(0, 256, size=(net.receptive_field,))
    forced = rng.integers(0, 256, size=(12,))

    def onehot(ix):
        return torch.from_numpy(intops.one_hot_proper(np.atleast_1d(ix)))[None]
    torch.set_num_threads(8)
    for correct in (False, True):
        pred_o, q_o, p0 = wo.fast_predict_next(params, dil, onehot(start), None, return_probs=True)
        want, want_p = [int(pred_o[0])], []
        for s_ in forced:
            pred_o, q_o, pr = wo.fast_predict_next(params, dil, onehot(s_), q_o, correct_queue=correct, return_probs=True)
            want.append(int(pred_o[0]))
            want_p.append(pr.numpy())
        pred, st = fg.predict_next(net, onehot(start).cuda(), None)
        if os.environ.get("WN_DEC_MFMA", "1") == "1":        # (the switch test re-runs this on the fp32 kernel)
            assert fg._mfma_decode(st.eng) and st.rw == 64
        got = [int(pred[0])]
        nxt = torch.from_numpy(np.concatenate([forced[1:], [0]]).astype(np.int32))
        codes, probs, _ = fg._decode(net, st, onehot(forced[0]).reshape(-1).cuda(), len(forced), forced=nxt,
                                     want_probs=True, correct_queue=correct)
        got += codes.cpu().tolist()
        err = np.abs(probs.cpu().numpy() - np.stack(want_p)).max()
        print("decode R%d D%d S%d (correct_queue=%s): probs err %.2e" % (R, D, S, correct, err))
        assert got == want and err < 1e-4
        last = "block_%d" % N
        assert tuple(st[last].shape) == tuple(q_o[last].shape)
        np.testing.assert_allclose(st[last].cpu().numpy(), q_o[last].numpy(), atol=1e-4, rtol=0)
    # batched generation: rows equal the single-stream result
    starts = torch.cat([onehot(np.roll(start, k)) for k in range(3)]).cuda()
    rows = fg.generate_codes_batch(net, starts, 24)
    for k in range(3):
        assert torch.equal(rows[k], fg.generate_codes(net, starts[k:k + 1], 24))


def test_autoencoder_backward_vs_oracle():
    """loss.backward() through the autoencoder (decoder with conditioning, epilogue conditioning,
    pooled encoding, encoder blocks) vs autograd on the CPU oracle, same per-forward projections."""
    import json
    import os
    from music_amd.model1 import wavenet_autoencoder
    from oracle import intops
    from tests.helpers import GOLDEN
    d = load_npz("g8_autoencoder.npz")
    cfg = json.load(open(os.path.join(GOLDEN, "g8_cfg.json")))
    params = params_from(d)
    net = wavenet_autoencoder(**cfg)
    net.load_state_dict(params)
    net = net.cuda()
    rng = np.random.default_rng(31)
    for tag in ("a", "b"):                       # "a": some layers stretch, "b": all tile
        idx = d[tag + "_idx"]
        x = torch.from_numpy(np.stack([intops.one_hot_proper(r) for r in idx]))
        W = idx.shape[1] - net.receptive_field + 1
        target = torch.from_numpy(rng.integers(0, 256, size=(idx.shape[0] * W,)).astype(np.int64))
        torch.manual_seed(77)
        net.zero_grad()
        probs = net(x.cuda())
        loss = torch.nn.CrossEntropyLoss()(probs, target.cuda())
        loss.backward()
        # oracle with the same projections
        torch.manual_seed(77)
        cond = wo.draw_conditioning(len(cfg["dilations"]), cfg["en_bottleneck_width"], cfg["de_dilation_channel"],
                                    cfg["de_skip_channel"])
        leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        p_ref, _ = wo.autoencoder_forward(leaf, cfg["dilations"], x, cfg["en_pool_kernel_size"], cond)
        l_ref = torch.nn.functional.cross_entropy(p_ref, target)
        g_ref = torch.autograd.grad(l_ref, list(leaf.values()), allow_unused=True)
        assert abs(loss.item() - l_ref.item()) < 1e-4
        worst = 0.0
        for (name, p), g in zip(net.named_parameters(), g_ref):
            g = torch.zeros_like(leaf[name]) if g is None else g
            err = (p.grad.cpu() - g).abs().max().item() / max(g.abs().max().item(), 1e-12)
            worst = max(worst, err)
            assert err <= GRAD_RTOL, (tag, name, err)
        print("autoencoder", tag, "worst relative grad err %.2e" % worst)


def test_autoencoder_bf16_forward_survives_activations_beyond_f16():
    """The autoencoder's encoder is an un-normalised ReLU residual net: with its block weights x 12 the residual stream reaches 8e4 - 6e5, beyond
    f16's 65504.  The default forward arithmetic (f16 hi / lo split) then returns garbage or NaN - a stated limit, DESIGN section 5 - and
    `net.precision = ("bf16x3", "bf16x3")` (float32's exponent range, 2^-17 per product) must give the oracle's probabilities and finite, close
    gradients on the same model."""
    from music_amd.model1 import wavenet_autoencoder
    from oracle import intops
    cfg = dict(filter_width=2, quantization_channel=256, dilations=[1, 2, 4, 8, 16, 32, 3], en_residual_channel=64, en_dilation_channel=64,
               en_bottleneck_width=16, en_pool_kernel_size=50, de_residual_channel=64, de_dilation_channel=64, de_skip_channel=256, use_bias=False)
    torch.manual_seed(11)
    net = wavenet_autoencoder(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
        net.connection_2.weight.mul_(6.0)
        for n, p in net.named_parameters():
            if n.startswith("en_dilation_layer_stack"):
                p.mul_(12.0)
        net.bottleneck_layer.weight.mul_(12.0 ** -4)                # (keeps the encoding, and with it the decoder, in its usual range)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rng = np.random.default_rng(12)
    B, W = 2, 400
    idx = rng.integers(0, 256, size=(B, net.receptive_field + W - 1))
    x = torch.from_numpy(np.stack([intops.one_hot_proper(r) for r in idx]))
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))
    torch.manual_seed(77)
    cond = wo.draw_conditioning(len(cfg["dilations"]), cfg["en_bottleneck_width"], cfg["de_dilation_channel"], cfg["de_skip_channel"])
    leaf = {k: v.double().clone().requires_grad_(True) for k, v in params.items()}
    p_ref, _ = wo.autoencoder_forward(leaf, cfg["dilations"], x.double(), cfg["en_pool_kernel_size"], [(w.double(), b.double()) for (w, b) in cond])
    l_ref = torch.nn.functional.cross_entropy(p_ref, target)
    g_ref = torch.autograd.grad(l_ref, list(leaf.values()), allow_unused=True)
    gmax = max(g.abs().max().item() for g in g_ref if g is not None)
    out = {}
    for prec in (("f16x3", "bf16x3"), ("bf16x3", "bf16x3")):
        net.precision = prec
        torch.manual_seed(77)
        net.zero_grad()
        probs = net(x.cuda())
        loss = torch.nn.CrossEntropyLoss()(probs, target.cuda())
        loss.backward()
        assert net._engine.mode_names == prec
        xe = net._engine.workspace(B, idx.shape[1])["Xe"]
        e_p = (probs.detach().cpu().double() - p_ref.detach()).abs().max().item()
        finite = all(torch.isfinite(p.grad).all().item() for p in net.parameters())
        worst = 0.0
        if finite:
            for (name, p), g in zip(net.named_parameters(), g_ref):
                g = torch.zeros_like(leaf[name]) if g is None else g
                worst = max(worst, (p.grad.cpu().double() - g).abs().max().item() / max(g.abs().max().item(), 1e-3 * gmax))
        out[prec[0]] = (e_p, finite, worst, float(xe[torch.isfinite(xe)].abs().max()))
        print("  %s forward: largest encoder activation %.2e, probabilities err %.2e, gradients finite %s, worst relative err %.2e" %
              (prec[0], out[prec[0]][3], e_p, finite, worst))
    assert out["bf16x3"][3] > 65504.0, "the model of this test must leave f16's range"
    assert (not out["f16x3"][1]) or out["f16x3"][0] > 1e-2          # the stated limit is real ...
    e_p, finite, worst, _ = out["bf16x3"]
    assert finite and e_p <= LOGIT_TOL and worst <= 5e-3             # ... and the bf16 split carries the model through it


def test_batched_decode_equals_single_utterances():
    """wn_decode_batch: U utterances side by side in one launch give, row by row, exactly
    the codes of U single-utterance launches (same arithmetic, independent state), with the as-written
    and with the corrected queue recurrence."""
    from music_amd.model import wavenet
    from music_amd import fast_generate as fg
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 8, 16, 32, 1, 2, 4, 8], dilation_channels=64, residual_channels=64,
               skip_channels=256, quantization_channels=256, use_bias=False)
    torch.manual_seed(3)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(3.0)
    net = net.cuda()
    rng = np.random.default_rng(4)
    U, n = 5, 200
    starts = scrambled_input(rng.integers(0, 256, size=(U, net.receptive_field))).cuda()
    for correct in (False, True):
        batch = fg.generate_codes_batch(net, starts, n, correct_queue=correct)
        assert batch.shape == (U, n)
        for u in range(U):
            single = fg.generate_codes(net, starts[u:u + 1], n, correct_queue=correct)
            assert torch.equal(batch[u], single.view(-1)), (correct, u)
        assert len(torch.unique(batch)) > 8           # not a degenerate constant stream
    print("batched decode: %d utterances x %d codes identical to single-utterance launches" % (U, n))


@pytest.mark.parametrize("bias", [False, True], ids=["nobias", "bias"])
def test_batched_decode_eight_per_pair_equals_single_utterances(bias):
    """Utterances run eight to a workgroup pair, one pair of MFMA result columns per utterance (decode_duo_mfma8_k):
    every row must equal the single-utterance launch (the utterance mirrored into all eight column pairs) exactly, for
    both queue recurrences, with and without biases, greedy and sampled."""
    from music_amd.model import wavenet
    from music_amd import fast_generate as fg
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 8, 16, 32, 1, 2, 4, 8, 3], dilation_channels=64, residual_channels=64,
               skip_channels=256, quantization_channels=256, use_bias=bias)
    torch.manual_seed(13)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(3.0)
    net = net.cuda()
    rng = np.random.default_rng(14)
    U, n = 24, 150
    starts = scrambled_input(rng.integers(0, 256, size=(U, net.receptive_field))).cuda()
    for correct in (False, True):
        batch = fg.generate_codes_batch(net, starts, n, correct_queue=correct)
        assert batch.shape == (U, n)
        for u in (0, 1, 7, 8, 15, 23):
            single = fg.generate_codes(net, starts[u:u + 1], n, correct_queue=correct)
            assert torch.equal(batch[u], single.view(-1)), (correct, u)
        assert len(torch.unique(batch)) > 8
    sampled = fg.generate_codes_batch(net, starts, n, temperature=0.9, seed=5)
    again = fg.generate_codes_batch(net, starts, n, temperature=0.9, seed=5)
    assert torch.equal(sampled, again) and not torch.equal(sampled, batch)
    print("eight-per-pair batched decode (bias=%s): %d utterances x %d codes identical to single launches" % (bias, U, n))


def test_sampled_decode_follows_the_distribution():
    """Sampling (an extension: the reference is greedy): with teacher forcing the kernel
    returns, per step, the distribution it drew from and the drawn code.  Checks: (1) every drawn code is
    what inverse-CDF sampling of THAT distribution gives for the kernel's own uniform number (restated on
    the CPU: splitmix64 of (seed, step, utterance)); (2) same seed -> same codes, other seed -> other codes;
    (3) temperature -> 0 reproduces the greedy codes; (4) the distribution at temperature T is the
    renormalised T = 1 distribution to the power 1/T."""
    from music_amd.model import wavenet
    from music_amd import fast_generate as fg
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 8, 16], dilation_channels=64, residual_channels=64,
               skip_channels=256, quantization_channels=256, use_bias=False)
    torch.manual_seed(9)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
    net = net.cuda()
    rng = np.random.default_rng(10)
    start = scrambled_input(rng.integers(0, 256, size=(1, net.receptive_field))).cuda()
    n = 300
    forced = torch.from_numpy(rng.integers(0, 256, size=(n,)).astype(np.int32)).cuda()

    def run(temperature, seed):
        with torch.no_grad():
            first, state = fg.predict_next(net, start, None)
        note0 = torch.zeros(256, device="cuda")
        note0[int(first[0])] = 1.0
        codes, probs, _ = fg._decode(net, state, note0, n, forced=forced, want_probs=True, temperature=temperature, seed=seed)
        return codes.cpu().numpy(), probs.cpu().numpy().astype(np.float64)

    def uniform(seed, step, utt=0):
        M = (1 << 64) - 1
        z = (seed + 0x9E3779B97F4A7C15 * (step + 1) + 0xD1B54A32D192ED03 * (utt + 1)) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        return (z >> 40) / 16777216.0

    c1, p1 = run(1.0, 7)
    assert np.allclose(p1.sum(1), 1.0, atol=1e-5)
    for t in range(n):                               # (1) the draw is the inverse CDF of the reported distribution
        u = uniform(7, t)
        cdf = np.cumsum(p1[t])
        k = int(c1[t])
        lo = cdf[k - 1] if k > 0 else 0.0
        assert lo - 1e-5 <= u < cdf[k] + 1e-5, (t, k, u, lo, cdf[k])
    c1b, _ = run(1.0, 7)
    c2, _ = run(1.0, 8)
    assert (c1 == c1b).all() and (c1 != c2).any()    # (2)
    cg, pg = run(None, 0)                            # greedy
    c0, _ = run(1e-9, 5)
    assert (c0 == cg).mean() >= 0.99                 # (3) (an exact fp32 tie of the two largest logits may go either way)
    assert (cg == pg.argmax(1)).all()
    _, ph = run(0.5, 3)                              # (4) T = 0.5: p^2 renormalised
    ref = pg ** 2
    ref /= ref.sum(1, keepdims=True)
    assert np.abs(ph - ref).max() < 1e-5
    assert len(np.unique(c1)) > 20
    print("sampled decode: %d draws consistent with their distributions; %d distinct codes" % (n, len(np.unique(c1))))


def test_wav_to_numpy_prep_on_device(tmp_path):
    """.wav files -> np_audio.pkl with the companding on the device: the codes are bit-exact
    against the oracle's restatement of the reference encoder (audio_func.py:5-22), the pickle has the
    reference's format (list of int32 arrays) and feeds audio_dataset unchanged."""
    import pickle
    from scipy.io import wavfile
    from music_amd import wav_to_numpy as w2n
    from music_amd.faster_audio_data import audio_dataset
    from oracle import intops
    thr = load_npz("g5_mulaw.npz")["thresholds"]             # the reference encoder's 255 float32 decision thresholds (G5)
    rng = np.random.default_rng(13)
    waves = [np.clip(0.3 * rng.standard_normal(n), -1.2, 1.2).astype(np.float32) for n in (5000, 12345)]
    waves[0][:6] = [0.0, 1.0, -1.0, 1.5, -1.5, 1e-9]              # clipping and the centre code
    codes = w2n.encode_waveforms(waves)
    for w, c in zip(waves, codes):
        assert c.dtype == np.int32 and c.shape == w.shape
        assert np.array_equal(c, intops.mu_law_encode_table(w, thr).astype(np.int32))
    d = str(tmp_path) + "/"
    pcm = (np.clip(waves[1], -1, 1) * 32767).astype(np.int16)
    wavfile.write(d + "a.wav", 16000, pcm)
    wavfile.write(d + "b.wav", 32000, np.repeat(pcm, 2))          # resampled to 16 kHz on load
    out = w2n.main(d)
    stored = pickle.load(open(d + "np_audio.pkl", "rb"))
    assert len(stored) == 2 and all(a.dtype == np.int32 for a in stored)
    assert np.array_equal(stored[0], intops.mu_law_encode_table(pcm.astype(np.float32) / 32768.0, thr).astype(np.int32))
    assert abs(len(stored[1]) - len(pcm)) <= 1
    ds = audio_dataset(d + "np_audio.pkl", 1025, 2000)
    assert len(ds) > 0 and ds[0]["audio_piece"].dtype == torch.int32


@pytest.mark.parametrize("pool", [50, 9], ids=["pool50", "pool9"])
def test_autoencoder_backward_64_channels_vs_oracle(pool):
    """The autoencoder at 64 decoder channels (BASELINE config-4 width): its decoder blocks run the one-launch backward
    block WITH the conditioning table (stretch and tile layers; bias and gradient of the conditioning on the matrix
    cores): loss and every gradient vs autograd on the CPU oracle with the same per-forward projections.  pool 9: 44 and
    81 pooled frames - more than the 32 buckets of that form, the blocks gather the bias and sum [df;dg] by bucket in a
    launch of its own (wn_resblock_bwd_ms + wn_cond_grad)."""
    from music_amd.model1 import wavenet_autoencoder
    from oracle import intops
    cfg = dict(filter_width=2, quantization_channel=256, dilations=[1, 2, 4, 8, 3, 16], en_residual_channel=48,
               en_dilation_channel=40, en_bottleneck_width=12, en_pool_kernel_size=pool, de_residual_channel=64,
               de_dilation_channel=64, de_skip_channel=80, use_bias=False)
    torch.manual_seed(41)
    net = wavenet_autoencoder(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
        net.connection_2.weight.mul_(10.0)          # peaked outputs (max p 0.4): the 1e-3 bar on the probabilities means something
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rng = np.random.default_rng(42)
    rf = net.receptive_field
    # 400 = 8 pooled frames (some layers stretch), 733: ragged; 70 short clips: the block launch's workgroups take items of
    # several clips, some clips apart (their conditioning-gradient sums go to one slot per clip, skipped clips get zeros)
    for B, W in ((2, 400), (1, 733), (70, 130))[:3 if pool == 50 else 2]:
        idx = rng.integers(0, 256, size=(B, rf + W - 1))
        x = torch.from_numpy(np.stack([intops.one_hot_proper(r) for r in idx]))
        target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))
        torch.manual_seed(78)
        net.zero_grad()
        probs = net(x.cuda())
        loss = torch.nn.CrossEntropyLoss()(probs, target.cuda())
        loss.backward()
        torch.manual_seed(78)
        cond = wo.draw_conditioning(len(cfg["dilations"]), cfg["en_bottleneck_width"], cfg["de_dilation_channel"],
                                    cfg["de_skip_channel"])
        leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        p_ref, _ = wo.autoencoder_forward(leaf, cfg["dilations"], x, cfg["en_pool_kernel_size"], cond)
        l_ref = torch.nn.functional.cross_entropy(p_ref, target)
        g_ref = torch.autograd.grad(l_ref, list(leaf.values()), allow_unused=True)
        assert (probs.detach().cpu() - p_ref.detach()).abs().max().item() <= LOGIT_TOL
        nonvacuous(p_ref.detach(), "autoencoder backward, %d x %d" % (B, W))
        assert abs(loss.item() - l_ref.item()) < 1e-4
        gs = [torch.zeros_like(leaf[n]) if g is None else g for (n, _), g in zip(net.named_parameters(), g_ref)]
        floor = 1e-3 * max(g.abs().max().item() for g in gs)
        worst = 0.0
        for (name, p), g in zip(net.named_parameters(), gs):
            err = (p.grad.cpu() - g).abs().max().item() / max(g.abs().max().item(), floor)
            worst = max(worst, err)
            assert err <= GRAD_RTOL, (B, W, name, err)
        print("autoencoder 64 ch (B=%d, W=%d): worst relative grad err %.2e" % (B, W, worst))


@pytest.mark.parametrize("width", [16, 64], ids=["16ch", "64ch"])
def test_autoencoder_backward_with_bias_vs_oracle(width):
    """use_bias=True through the whole autoencoder (every conv of model1.py:33-134 has a bias then): loss and
    every gradient, bias gradients included, vs autograd on the CPU oracle.  64 decoder channels take the
    channel-split backward block (bias added in its recompute), 16 the generic path."""
    from music_amd.model1 import wavenet_autoencoder
    from oracle import intops
    cfg = dict(filter_width=2, quantization_channel=256, dilations=[1, 2, 4, 3, 8], en_residual_channel=width - 3,
               en_dilation_channel=width - 5, en_bottleneck_width=10, en_pool_kernel_size=40, de_residual_channel=width,
               de_dilation_channel=width - 2, de_skip_channel=48, use_bias=True)
    torch.manual_seed(51 + width)
    net = wavenet_autoencoder(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
        net.connection_2.weight.mul_(10.0)          # peaked outputs (max p 0.24 / 0.30)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    assert any(k.endswith(".bias") for k in params)
    net = net.cuda()
    rng = np.random.default_rng(43)
    rf = net.receptive_field
    for B, W in ((2, 320), (1, 517)):            # 320 = 8 pooled frames (some layers stretch), 517: ragged (tile)
        idx = rng.integers(0, 256, size=(B, rf + W - 1))
        x = torch.from_numpy(np.stack([intops.one_hot_proper(r) for r in idx]))
        target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))
        torch.manual_seed(79)
        net.zero_grad()
        probs = net(x.cuda())
        loss = torch.nn.CrossEntropyLoss()(probs, target.cuda())
        loss.backward()
        torch.manual_seed(79)
        cond = wo.draw_conditioning(len(cfg["dilations"]), cfg["en_bottleneck_width"], cfg["de_dilation_channel"],
                                    cfg["de_skip_channel"])
        leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        p_ref, _ = wo.autoencoder_forward(leaf, cfg["dilations"], x, cfg["en_pool_kernel_size"], cond)
        l_ref = torch.nn.functional.cross_entropy(p_ref, target)
        g_ref = torch.autograd.grad(l_ref, list(leaf.values()), allow_unused=True)
        assert (probs.detach().cpu() - p_ref.detach()).abs().max().item() <= LOGIT_TOL
        nonvacuous(p_ref.detach(), "autoencoder backward, %d x %d" % (B, W))
        assert abs(loss.item() - l_ref.item()) < 1e-4
        gs = [torch.zeros_like(leaf[n]) if g is None else g for (n, _), g in zip(net.named_parameters(), g_ref)]
        floor = 1e-3 * max(g.abs().max().item() for g in gs)
        worst = 0.0
        for (name, p), g in zip(net.named_parameters(), gs):
            err = (p.grad.cpu() - g).abs().max().item() / max(g.abs().max().item(), floor)
            worst = max(worst, err)
            assert err <= GRAD_RTOL, (B, W, name, err)
        print("autoencoder with bias, %d ch (B=%d, W=%d): worst relative grad err %.2e" % (width, B, W, worst))


@pytest.mark.parametrize("use_bias", [False, True], ids=["nobias", "bias"])
def test_autoencoder_cached_generation_matches_decoder(use_bias):
    """encoder once + fixed conditioning projections + the persistent cached-queue decode kernel.
    (a) teacher forced, the decode kernel's probabilities equal the CPU oracle's conditioned decoder
    (model1.py:158-225 restated, same encoding and projections) on the sliding receptive-field window, step by
    step; (b) the greedy roll-out of ``generate_cached`` equals the oracle's greedy roll-out."""
    from music_amd.model1 import wavenet_autoencoder
    from music_amd import ae_generate as ag
    from music_amd import fast_generate as fg
    from oracle import intops
    cfg = dict(filter_width=2, quantization_channel=256, dilations=[1, 2, 4, 8, 3, 1, 2], en_residual_channel=24,
               en_dilation_channel=20, en_bottleneck_width=6, en_pool_kernel_size=32, de_residual_channel=40,
               de_dilation_channel=36, de_skip_channel=72, use_bias=use_bias)
    torch.manual_seed(61)
    net = wavenet_autoencoder(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.5)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rf, dil = net.receptive_field, cfg["dilations"]
    rng = np.random.default_rng(62)
    start = rng.integers(0, 256, size=(rf + 32 + 5,))            # pools to exactly one frame
    forced = rng.integers(0, 256, size=(24,))

    def onehot(ix):
        return torch.from_numpy(intops.one_hot_proper(np.atleast_1d(ix)))[None]

    x0 = onehot(start)
    with torch.no_grad():
        net(x0.cuda())
    enc = net.last_encoding.cpu()
    assert tuple(enc.shape) == (1, 6, 1)
    enc_ref = wo.autoencoder_encode(params, dil, x0, cfg["en_pool_kernel_size"])
    assert (enc - enc_ref).abs().max().item() < 1e-4
    torch.manual_seed(63)
    cond = net._draw_conditioning()
    wnet = ag.cached_decoder(net, enc, cond)

    def ref_probs(seq):
        return wo.autoencoder_decode(params, dil, onehot(np.asarray(seq[-rf:])), enc_ref, 1, cond).reshape(-1)

    # (a) teacher forced
    seq = list(start)
    want = [ref_probs(seq)]
    for c in forced:
        seq.append(int(c))
        want.append(ref_probs(seq))
    pred, st = fg.predict_next(wnet, onehot(start[-rf:]).cuda(), None)
    nxt = torch.from_numpy(np.concatenate([forced[1:], [0]]).astype(np.int32))
    codes, probs, _ = fg._decode(wnet, st, onehot(forced[0]).reshape(-1).cuda(), len(forced), forced=nxt, want_probs=True,
                                 correct_queue=True)
    assert int(pred[0]) == int(want[0].argmax())
    err = (probs.cpu() - torch.stack(want[1:])).abs().max().item()
    assert err <= LOGIT_TOL, err
    nonvacuous(torch.stack(want[1:]), "autoencoder cached generation")
    assert codes.cpu().tolist() == [int(w.argmax()) for w in want[1:]]
    # (b) greedy roll-out
    n = 16
    got, _, _ = ag.generate_cached(net, x0, n, cond=cond)
    seq, ref = list(start), []
    for _ in range(n):
        c = int(ref_probs(seq).argmax())
        ref.append(c)
        seq.append(c)
    assert got.cpu().tolist() == ref
    print("autoencoder cached generation (bias=%s): teacher-forced probs err %.2e, %d greedy codes equal" % (use_bias, err, n))


def test_autoencoder_fused_step_equals_autograd_path():
    """The autoencoder engine's fused training step (forward to logits, ONE softmax + CE + backward kernel, backward,
    flat Adam) against the drop-in path (forward -> nn.CrossEntropyLoss -> autograd -> torch.optim.Adam) with the same
    conditioning projections: loss, every gradient and the parameters after one Adam step."""
    from music_amd.model1 import wavenet_autoencoder
    from oracle import intops
    cfg = dict(filter_width=2, quantization_channel=256, dilations=[1, 2, 4, 8, 3], en_residual_channel=60,
               en_dilation_channel=52, en_bottleneck_width=10, en_pool_kernel_size=40, de_residual_channel=64,
               de_dilation_channel=60, de_skip_channel=72, use_bias=True)
    rng = np.random.default_rng(71)
    nets = []
    for _ in range(2):
        torch.manual_seed(72)
        net = wavenet_autoencoder(**cfg)
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(2.0)
        nets.append(net.cuda())
    a, b = nets
    rf, B, W = a.receptive_field, 2, 333
    idx = rng.integers(0, 256, size=(B, rf + W - 1))
    x = torch.from_numpy(np.stack([intops.one_hot_proper(r) for r in idx])).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).cuda()
    # (a) drop-in path
    opt = torch.optim.Adam(a.parameters(), lr=1e-3)
    torch.manual_seed(73)
    opt.zero_grad()
    loss_a = torch.nn.CrossEntropyLoss()(a(x), target)
    loss_a.backward()
    grads_a = {n: p.grad.detach().clone() for n, p in a.named_parameters()}
    opt.step()
    # (b) fused path, same projections
    eng = b._engine_for(x.device)
    eng.adam_init(lr=1e-3)
    torch.manual_seed(73)
    cond = b._draw_conditioning()
    loss_b = eng.loss_and_grad(x, target, cond)
    assert abs(loss_a.item() - loss_b.item()) < 1e-6
    gmax = max(g.abs().max().item() for g in grads_a.values())
    for n, p in b.named_parameters():
        o = eng.spec.off[n]
        g = eng.flat_grad[o:o + p.numel()].view(p.shape)
        assert (g - grads_a[n]).abs().max().item() <= 2e-4 * max(grads_a[n].abs().max().item(), 1e-3 * gmax), n      # two CE formulations
    g_b = eng.flat_grad.clone()
    before = {n: p.detach().clone() for n, p in b.named_parameters()}
    eng.adam_step()
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        # the flat Adam against the formula of the first step (m^ = g, v^ = g^2) on ITS OWN gradient: tight ...
        o = eng.spec.off[n]
        g = g_b[o:o + pb.numel()].view(pb.shape)
        want = before[n] - 1e-3 * g / (g.abs() + 1e-8)
        assert (pb.detach() - want).abs().max().item() <= 1e-7, n
        # ... and against torch.optim.Adam on the drop-in path wherever the step is not a division of two rounding errors
        # (|g| >> Adam's eps = 1e-8; below that a 1e-10 difference between the two CE formulations moves the update by O(lr))
        big = (g.abs() > 1e-6) & (grads_a[n].abs() > 1e-6)
        if big.any():
            assert (pa - pb).detach()[big].abs().max().item() <= 3e-6, n


def test_clip_pairs_on_the_64_channel_blocks_equal_the_32_channel_kernels(monkeypatch):
    """A model with <= 32 channels and an even batch runs its stack on the 64-channel block kernels, two clips per 64-row
    tensor with block-diagonal packs (ws["pair"]); WN_PAIR32=0 and odd batches keep the 32-channel kernels.  Same function:
    probabilities, loss and every gradient of the two paths agree to rounding, and an odd batch takes the old path."""
    from music_amd.engine import WaveNetEngine
    cfg = dict(dilations=[1, 2, 4, 8, 3, 16], residual_channels=24, dilation_channels=32, skip_channels=72)
    rng = np.random.default_rng(5)
    engs = []
    for env in ("1", "0"):
        monkeypatch.setenv("WN_PAIR32", env)
        torch.manual_seed(9)
        e = WaveNetEngine(**cfg, device="cuda")
        torch.nn.init.uniform_(e.flat, -0.3, 0.3)
        engs.append(e)
    engs[1].flat.copy_(engs[0].flat)
    assert engs[0].pair_ok and not engs[1].pair_ok
    rf = engs[0].rf
    for B, W in ((4, 333), (3, 120)):
        T = rf + W - 1
        codes = torch.from_numpy(rng.integers(0, 256, size=(B, T)).astype(np.int32)).cuda()
        target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).cuda()
        outs = []
        for e in engs:
            loss = e.loss_and_grad_codes(codes, target, scrambled=True, want_probs=True)
            ws = e.workspace(B, T)
            outs.append((loss.item(), ws["probs"].clone(), e.flat_grad.clone(), ws["pair"]))
        assert outs[0][3] == (B % 2 == 0) and not outs[1][3]
        assert abs(outs[0][0] - outs[1][0]) < 1e-6
        assert (outs[0][1] - outs[1][1]).abs().max().item() < 1e-6
        gmax = outs[1][2].abs().max().item()
        assert (outs[0][2] - outs[1][2]).abs().max().item() <= 2e-5 * gmax


@pytest.mark.parametrize("ch", [64, 32], ids=["64ch", "32ch_clip_pairs"])
def test_training_reduces_the_loss_64_channels(ch):
    """End-to-end sanity of the production kernels (one-launch backward block, fused CE, flat
    Adam): 60 fused steps on one fixed small batch drive the loss from ln(256) towards its floor.
    The reference applies CrossEntropyLoss to PROBABILITIES, so the loss lives in
    [log(e + 255) - 1, ...] = [4.552, ...]: memorising the batch means approaching 4.55.  32 channels: the same on clip pairs."""
    from music_amd.model import wavenet
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 8, 16, 32], dilation_channels=ch, residual_channels=ch,
               skip_channels=128, quantization_channels=256, use_bias=False)
    torch.manual_seed(5)
    net = wavenet(**cfg).cuda()
    rng = np.random.default_rng(6)
    T = net.receptive_field + 95
    x = scrambled_input(rng.integers(0, 256, size=(2, T))).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(2 * 96,)).astype(np.int64)).cuda()
    net(x[:, :, :net.receptive_field])
    eng = net._engine
    eng.adam_init(lr=3e-3)
    losses = []
    for _ in range(60):
        losses.append(eng.loss_and_grad(x, target).item())
        eng.adam_step()
    print("loss %.4f -> %.4f (floor 4.552)" % (losses[0], losses[-1]))
    assert abs(losses[0] - np.log(256.0)) < 5e-3
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0] - 0.3          # (the double softmax makes early gradients tiny: 5.545 -> ~5.05 in 60 steps)


def _g7_run(tmp_path, monkeypatch, tag, extra=None):
    import json
    import os
    import pickle
    from tests.helpers import GOLDEN
    g7 = json.load(open(os.path.join(GOLDEN, "g7_train.json")))
    os.makedirs(tmp_path / "params")
    rng = np.random.default_rng(g7["data_seed"])
    data = [rng.integers(0, 256, size=(l,)).astype(np.int32) for l in g7["data_lens"]]
    pickle.dump(data, open(tmp_path / "np_audio.pkl", "wb"))
    dp = dict(g7["dataset_params"], audio_path=str(tmp_path / "np_audio.pkl"))
    tp = dict(g7["train_params"], **(extra or {}))
    for n, p in (("wavenet", g7["wavenet_params"]), ("dataset", dp), ("train", tp)):
        json.dump(p, open(tmp_path / "params" / (n + "_params.json"), "w"))
    monkeypatch.chdir(tmp_path)
    return g7


@pytest.mark.parametrize("fused", [False, True])
def test_g7_train_loop_on_gpu(tmp_path, monkeypatch, fused):
    """music_amd/train.py: train() end to end on the MI355X (loader -> HIP one-hot -> HIP model ->
    CrossEntropyLoss -> backward -> Adam) reproduces the loss_log / store_log / checkpoints the
    reference's own train() wrote for the same seed, data and (gain-3) weights."""
    from music_amd import train as T
    from music_amd.model import wavenet
    g7 = _g7_run(tmp_path, monkeypatch, "gain", {"fused_step": fused})

    def ctor(**kw):
        net = wavenet(**kw)
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(g7["gain"])
        return net
    monkeypatch.setattr(T, "wavenet", ctor)
    torch.manual_seed(0)
    T.train()
    got = open(tmp_path / "log" / "loss_log.log").read().strip().split("\n")
    want = g7["gain_loss_log"].strip().split("\n")
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a.split("Average")[0] == b.split("Average")[0]
        assert abs(float(a.split(' ')[-1]) - float(b.split(' ')[-1])) < 1e-4, (a, b)
    assert open(tmp_path / "log" / "store_log.log").read() == g7["gain_store_log"]
    ck = torch.load(tmp_path / "restore" / "wavenet2.model")
    assert list(ck.keys()) == g7["gain_ckpt_keys"]
    for v, s in zip(ck.values(), g7["gain_ckpt_abs_sum"]):
        assert abs(float(v.double().abs().sum()) - s) <= 2e-3 * max(1.0, s)


@pytest.mark.parametrize("fused", [False, True])
def test_optimizer_state_survives_a_restart_on_gpu(tmp_path, monkeypatch, fused):
    """SURVEY 8f4 on the device: stop after epoch 2, resume from wavenet2.model + wavenet2.opt for one
    epoch == an uninterrupted 3-epoch run, bit for bit (the HIP path is deterministic), for the torch
    optimizer and for the flat Adam buffers of the fused step."""
    import json
    import os
    from music_amd import train as T
    from music_amd.model import wavenet
    finals = {}
    for mode in ("straight", "resumed"):
        root = tmp_path / mode
        os.makedirs(root)
        extra = {"fused_step": fused, "optimizer": "adam", "learning_rate": 1e-3, "save_optimizer_state": True,
                 "check_point_every": 1, "max_check_points": 10, "num_epochs": 3 if mode == "straight" else 2}
        g7 = _g7_run(root, monkeypatch, "gain", extra)

        def ctor(**kw):
            net = wavenet(**kw)
            with torch.no_grad():
                for p in net.parameters():
                    p.mul_(g7["gain"])
            return net
        monkeypatch.setattr(T, "wavenet", ctor)
        torch.manual_seed(0)
        T.train()
        if mode == "resumed":
            assert os.path.exists(root / "restore" / "wavenet2.opt")
            tp = dict(g7["train_params"], **extra)
            tp.update(num_epochs=1, restore_model="wavenet2.model")
            json.dump(tp, open(root / "params" / "train_params.json", "w"))
            T.train()
        finals[mode] = torch.load(root / "restore" / "wavenet3.model")
        monkeypatch.undo()
    for (k, a), b in zip(finals["straight"].items(), finals["resumed"].values()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("fused", [False, True], ids=["autograd", "fused_step"])
def test_autoencoder_train_harness_on_gpu(tmp_path, monkeypatch, fused):
    import json
    import os
    import pickle
    from music_amd import ae_train as A
    from tests.helpers import GOLDEN
    cfg = json.load(open(os.path.join(GOLDEN, "g8_cfg.json")))
    os.makedirs(tmp_path / "params")
    rng = np.random.default_rng(3)
    pickle.dump([rng.integers(0, 256, size=(l,)).astype(np.int32) for l in (700, 500)], open(tmp_path / "a.pkl", "wb"))
    dp = dict(batch_size=2, shuffle=True, num_workers=0, pin_memory=False, audio_path=str(tmp_path / "a.pkl"),
              receptive_field=32, window_length=100, cuda_available=True, quantization_channels=256)
    tp = dict(log_dir="./log/", restore_dir="./restore/", restore_model="", check_point_every=1, print_every=1,
              num_epochs=2, optimizer_type="Adam", max_check_points=2, learning_rate=1e-3, momentum=0.9,
              device_ids=None, seed=5, fused_step=fused)
    for n, p in (("model", cfg), ("dataset", dp), ("train", tp)):
        json.dump(p, open(tmp_path / "params" / (n + "_params.json"), "w"))
    monkeypatch.chdir(tmp_path)
    A.train()
    lines = open(tmp_path / "log" / "loss_log.log").read().strip().split("\n")
    losses = [float(l.split(' ')[-1]) for l in lines]
    assert len(losses) >= 4 and all(np.isfinite(losses)) and all(5.0 < v < 6.0 for v in losses)
    assert sorted(os.listdir(tmp_path / "restore")) == ["wavenet_autoencoder1.model", "wavenet_autoencoder2.model"]


@pytest.mark.parametrize("scrambled", [True, False], ids=["scrambled", "proper"])
def test_code_aware_causal_gradient_equals_dense_path(scrambled, monkeypatch):
    """A one-hot built by eng.onehot / the loader carries its codes: the causal layer's weight gradient is then a
    scatter (wn_causal_wgrad_codes).  Same gradients as the dense product on the same tensor; a tensor that was
    modified after it was built, or any other float input, takes the dense path."""
    from music_amd.model import wavenet
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 8, 16, 32, 64], dilation_channels=64, residual_channels=64,
               skip_channels=64, quantization_channels=256, use_bias=True)
    torch.manual_seed(5)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.5)
    net = net.cuda()
    eng = net._engine_for(torch.device("cuda", 0))
    rng = np.random.default_rng(6)
    B, T = 3, net.receptive_field + 900
    W = T - net.receptive_field + 1
    codes = torch.from_numpy(rng.integers(0, 256, size=(B, T)).astype(np.int32)).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).cuda()
    x = eng.onehot(codes, scrambled=scrambled)
    assert x._wn_codes[0] is codes
    eng.loss_and_grad(x, target)
    assert eng.workspace(B, T)["x_codes"] is not None
    g_codes = eng.flat_grad.clone()
    eng.loss_and_grad(x.clone(), target)                  # a copy carries no codes: the dense path
    assert eng.workspace(B, T)["x_codes"] is None
    g_dense = eng.flat_grad.clone()
    # both the forward (gather of exact fp32 weights instead of the f16 split product: x0 moves by ~2e-7) and the causal
    # weight gradient (scatter) differ in rounding only
    o = eng.spec.off["causal_layer.weight"]
    n = 64 * 256 * 2
    a, b = g_codes[o:o + n], g_dense[o:o + n]
    for name in eng.param_names:
        po, k = eng.spec.off[name], int(np.prod(eng.spec.shape[name]))
        u, v = g_codes[po:po + k], g_dense[po:po + k]
        assert (u - v).abs().max().item() <= 1e-2 * max(v.abs().max().item(), 1e-30), name     # two valid roundings of x0
    # through the nn.Module + autograd surface the tag survives detach()
    net.zero_grad()
    torch.nn.CrossEntropyLoss()(net(x), target).backward()
    assert eng.workspace(B, T)["x_codes"] is not None
    assert (net.causal_layer.weight.grad.reshape(-1) - a).abs().max().item() <= GRAD_RTOL * a.abs().max().item()
    # a modified tensor is no longer the one-hot of its codes: dense path, and the gradient follows the data
    x[0, :, 100] = 0.5
    eng.loss_and_grad(x, target)
    assert eng.workspace(B, T)["x_codes"] is None
    # ... and so is one whose CODES were overwritten after it was built (a reused staging buffer)
    x2 = eng.onehot(codes, scrambled=scrambled)
    saved = codes.clone()
    codes.add_(1).remainder_(256)
    eng.loss_and_grad(x2, target)
    assert eng.workspace(B, T)["x_codes"] is None
    codes.copy_(saved)
    # plain float input (no codes at all)
    eng.loss_and_grad(torch.rand(B, 256, T, device="cuda"), target)
    assert eng.workspace(B, T)["x_codes"] is None
    # the codes alone (no one-hot tensor at all): same loss, probabilities and gradients
    x = eng.onehot(codes, scrambled=scrambled)
    l_dense = eng.loss_and_grad(x.clone(), target, want_probs=True).item()      # (a copy: the dense path)
    p_dense = eng.workspace(B, T)["probs"].clone()
    g_dense = eng.flat_grad.clone()
    l_codes = eng.loss_and_grad_codes(codes, target, scrambled=scrambled, want_probs=True).item()
    assert abs(l_codes - l_dense) < 1e-6
    assert (eng.workspace(B, T)["probs"] - p_dense).abs().max().item() < 1e-4
    assert (eng.flat_grad - g_dense).abs().max().item() <= 1e-2 * g_dense.abs().max().item()
    # ... and it is the code-aware path of the tagged tensor, bit for bit
    g_c = eng.flat_grad.clone()
    eng.loss_and_grad(eng.onehot(codes, scrambled=scrambled), target)
    assert torch.equal(g_c, eng.flat_grad)


@pytest.mark.parametrize("scrambled", [True, False], ids=["scrambled", "proper"])
def test_autoencoder_causal_layers_on_codes_equal_dense_path(scrambled, monkeypatch):
    """The autoencoder's two causal layers (encoder and decoder) on the integer codes of a loader-built one-hot (gather
    forward, scatter backward) against the dense products on the same tensor, same conditioning projections."""
    from music_amd.faster_audio_data import onehot_device
    from music_amd.model1 import wavenet_autoencoder
    cfg = dict(filter_width=2, quantization_channel=256, dilations=[1, 2, 4, 8, 16, 3], en_residual_channel=64,
               en_dilation_channel=64, en_bottleneck_width=12, en_pool_kernel_size=50, de_residual_channel=64,
               de_dilation_channel=64, de_skip_channel=80, use_bias=False)
    torch.manual_seed(43)
    net = wavenet_autoencoder(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
    net = net.cuda()
    rng = np.random.default_rng(44)
    B, W = 2, 500
    T = net.receptive_field + W - 1
    codes = torch.from_numpy(rng.integers(0, 256, size=(B, T)).astype(np.int32)).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).cuda()
    eng = net._engine_for(torch.device("cuda", 0))
    x = onehot_device(codes, 256, scrambled=scrambled)
    torch.manual_seed(9)
    cond = net._draw_conditioning()
    l1 = eng.loss_and_grad(x, target, cond).item()
    assert eng.workspace(B, T)["x_codes"] is not None
    g1 = eng.flat_grad.clone()
    l0 = eng.loss_and_grad(x.clone(), target, cond).item()      # a copy carries no codes: both causal layers on the dense tensor
    assert eng.workspace(B, T)["x_codes"] is None
    g0 = eng.flat_grad.clone()
    assert abs(l1 - l0) < 1e-6
    for name, p in net.named_parameters():
        o = eng.spec.off[name]
        u, v = g1[o:o + p.numel()], g0[o:o + p.numel()]
        assert (u - v).abs().max().item() <= 1e-2 * max(v.abs().max().item(), 1e-30), name
    # bit-reproducible, and through autograd the tag survives detach()
    eng.loss_and_grad(x, target, cond)
    assert torch.equal(g1, eng.flat_grad)
    torch.manual_seed(9)
    net.zero_grad()
    torch.nn.CrossEntropyLoss()(net(x), target).backward()
    assert eng.workspace(B, T)["x_codes"] is not None


def test_two_forwards_in_flight_accumulate_like_the_reference():
    """Gradient accumulation over two micro-batches of the SAME shape with both forwards run before the first backward
    (the reference's autograd allows it, wavenet/model.py:86-145): each forward keeps its own
    workspace until its backward has run.  .grad must equal the sum of the oracle's two gradients."""
    meta = [m for m in g1_meta() if m["name"] == "tiny_s0_g3_w130"][0]
    d = load_npz("g1_%s.npz" % meta["name"])
    params = params_from(d)
    net = build(meta["cfg"], params)
    x1 = g1_input(d, meta)
    t1 = torch.from_numpy(d["target"])
    x2 = x1.flip(2).contiguous()
    t2 = t1.flip(0).contiguous()
    ce = torch.nn.CrossEntropyLoss()
    p1 = net(x1.cuda())
    p2 = net(x2.cuda())                                         # same shape, first forward still waiting for its backward
    eng = net._engine
    assert len(eng._ws) == 2
    (ce(p1, t1.cuda().view(-1)) + ce(p2, t2.cuda().view(-1))).backward()
    _, _, g1 = wo.loss_and_grads(params, meta["cfg"]["dilations"], x1, t1)
    _, _, g2 = wo.loss_and_grads(params, meta["cfg"]["dilations"], x2, t2)
    for name, p in net.named_parameters():
        want = g1[name] + g2[name]
        got = torch.zeros_like(want) if p.grad is None else p.grad.cpu()
        err = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-12)
        assert err <= GRAD_RTOL, (name, err)
    # both workspaces are free again: a third forward reuses the first one, no third allocation
    with torch.no_grad():
        net(x1.cuda())
    assert len(eng._ws) == 2
    # an output dropped without a backward frees its workspace when the graph dies
    p3 = net(x1.cuda())
    del p3
    p4 = net(x1.cuda())
    p5 = net(x2.cuda())
    assert len(eng._ws) == 2
    ce(p4, t1.cuda().view(-1)).backward()
    ce(p5, t2.cuda().view(-1)).backward()


def test_workspace_pool_evicts_one_shape_at_a_time():
    """A fifth input shape evicts the least recently used shape only (round 2 dropped every cached workspace), never one
    that a pending backward holds."""
    meta = [m for m in g1_meta() if m["name"] == "tiny_s0_g3_w130"][0]
    d = load_npz("g1_%s.npz" % meta["name"])
    net = build(meta["cfg"], params_from(d))
    rf = net.receptive_field
    x = lambda T: torch.zeros(1, 256, T, device="cuda")
    held = net(x(rf + 10))                                     # grad mode: shape A, held until its backward
    eng = net._engine
    with torch.no_grad():
        for extra in (1, 2, 3):
            net(x(rf + 10 + extra))                             # shapes B, C, D: four shapes cached
    assert sorted(eng._ws._d.keys()) == [(1, rf + 10 + e) for e in (0, 1, 2, 3)]
    a_ws = eng.workspace(1, rf + 10)
    with torch.no_grad():
        net(x(rf + 50))                                        # a fifth shape evicts the least recently used one that is free: B
    shapes = sorted(eng._ws._d.keys())
    assert shapes == [(1, rf + 10), (1, rf + 12), (1, rf + 13), (1, rf + 50)], shapes
    assert eng.workspace(1, rf + 10) is a_ws and a_ws.get("held")
    held.sum().backward()
    assert not a_ws.get("held")


def test_in_place_write_to_a_tagged_input_is_reported_at_backward():
    """the code-aware causal layer re-checks its input at backward time - an in-place op on the one-hot (or on
    the codes it was built from) between forward and backward is an error (dense input) or a fall-back to the dense
    weight-gradient product (codes changed, dense tensor intact), never a silently wrong gradient."""
    meta = [m for m in g1_meta() if m["name"] == "tiny_s0_g3_w130"][0]
    d = load_npz("g1_%s.npz" % meta["name"])
    params = params_from(d)
    net = build(meta["cfg"], params)
    idx = torch.from_numpy(d["idx"].astype(np.int32)).cuda()
    target = torch.from_numpy(d["target"]).cuda().view(-1)
    net(torch.zeros(1, 256, net.receptive_field, device="cuda"))
    eng = net._engine
    ce = torch.nn.CrossEntropyLoss()
    x = eng.onehot(idx, scrambled=True)
    ce(net(x), target).backward()
    want = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    # codes modified after the forward: the dense tensor still holds what the forward saw -> dense product, same gradient
    net.zero_grad()
    x = eng.onehot(idx, scrambled=True)
    out = net(x)
    idx.add_(0)                                                # bumps the version counter, values unchanged
    ce(out, target).backward()
    for n, g in want.items():
        got = dict(net.named_parameters())[n].grad
        assert (got - g).abs().max().item() <= 1e-5 * max(g.abs().max().item(), 1e-12), n
    # the dense tensor itself modified: reported
    x = eng.onehot(idx, scrambled=True)
    out = net(x)
    x.mul_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        ce(out, target).backward()


def _small_net(bias=False, seed=5):
    from music_amd.model import wavenet
    torch.manual_seed(seed)
    net = wavenet(filter_width=2, dilations=[1, 2, 4, 8, 16, 32, 64], dilation_channels=64, residual_channels=64, skip_channels=256,
                  quantization_channels=256, use_bias=bias)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.5)
    return net.cuda()


def test_cross_entropy_on_the_module_output_runs_fused_and_equals_torchs():
    """`nn.CrossEntropyLoss()(net(x), target)` - the reference's own loss call (wavenet/train.py:146,179) - is intercepted on the
    module's output (a Tensor subclass) and runs as the engine's one-pass softmax + CE + backward; loss and EVERY parameter gradient
    must equal what torch's own CrossEntropyLoss gives on the same module (`fuse_loss = False`), including: a scaled loss, a loss
    that uses the probabilities a second time, non-default arguments (torch's path), an in-place change of the output (torch's
    path), and inference under no_grad."""
    import numpy as np
    from music_amd import model as mm
    from tests.helpers import scrambled_input
    net = _small_net()
    rng = np.random.default_rng(3)
    B, W = 3, 700
    T = net.receptive_field + W - 1
    x = scrambled_input(rng.integers(0, 256, size=(B, T))).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).cuda()
    crit = torch.nn.CrossEntropyLoss()

    def run(fuse, loss_fn):
        net.fuse_loss = fuse
        net.zero_grad()
        out = net(x)
        loss = loss_fn(out)
        loss.backward()
        return float(loss.detach()), [p.grad.clone() for p in net.parameters()], out

    cases = {
        "plain": lambda o: crit(o, target),
        "scaled": lambda o: 3.0 * crit(o, target),
        "second use": lambda o: crit(o, target) + 0.25 * (o * o).sum() / o.numel(),
        "functional": lambda o: torch.nn.functional.cross_entropy(o, target),
        "sum reduction (torch's path)": lambda o: torch.nn.CrossEntropyLoss(reduction="sum")(o, target) / target.numel(),
        "smoothing (torch's path)": lambda o: torch.nn.CrossEntropyLoss(label_smoothing=0.1)(o, target),
    }
    for name, fn in cases.items():
        l0, g0, _ = run(False, fn)
        l1, g1, out = run(True, fn)
        assert type(out) is mm._Probs and out.shape == (B * W, 256)
        assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0)), (name, l0, l1)
        gmax = max(g.abs().max().item() for g in g0)
        for a, b in zip(g0, g1):
            # (the two paths apply a loss scale on different sides of the backward's 16-bit operand split, 2^-17 per product: observed
            # up to 2.05e-5 of a tensor's max - the bar was 2e-5 until the gate derivatives changed their rounding in round 6)
            assert (a - b).abs().max().item() <= 5e-5 * max(a.abs().max().item(), 1e-3 * gmax), name
    # a retained graph: a second backward through the fused loss gives the same gradients again (accumulated: twice)
    net.fuse_loss = True
    net.zero_grad()
    out = net(x)
    loss = crit(out, target) + 0.25 * (out * out).sum() / out.numel()
    loss.backward(retain_graph=True)
    first = [p.grad.clone() for p in net.parameters()]
    loss.backward()
    for a, p in zip(first, net.parameters()):
        assert (p.grad - 2.0 * a).abs().max().item() <= 1e-6 * max(a.abs().max().item(), 1e-30) + 1e-12
    # the fused pass really ran in the plain case (and only there among the torch-path cases)
    net.fuse_loss = True
    net.zero_grad()
    out = net(x)
    hook = out._wn_hook
    loss = crit(out, target)
    assert hook.fused and loss.grad_fn is not None and type(loss) is torch.Tensor
    assert crit(out, target).grad_fn is not None and hook.fused          # a second loss on the same output: torch's path
    loss.backward()
    out = net(x)
    out.mul_(1.0)                                                       # modified in place: torch's path
    assert not (crit(out, target), out._wn_hook.fused)[1]
    with torch.no_grad():
        o2 = net(x)
        assert type(o2) is torch.Tensor and abs(float(crit(o2, target)) - float(loss)) < 1e-5
    # everything else sees an ordinary tensor
    assert type(out + 1) is torch.Tensor and type(out.view(-1)) is torch.Tensor and type(out.detach()) is torch.Tensor


def test_fused_loss_survives_other_backward_passes_on_the_same_output():
    """Backward passes in any order over one forward: (a) ANOTHER loss on the output is back-propagated on its own
    (retain_graph) before the fused loss - that ordinary backward overwrites the workspace's d loss / d pre-softmax, which the fused
    loss's backward must re-form; (b) `torch.autograd.grad(loss, out)` runs the fused node's backward without the module's - its
    upstream gradient must not leak into the next, unrelated backward.  Reference: the same sequence with `fuse_loss = False`."""
    import numpy as np
    from tests.helpers import scrambled_input
    net = _small_net()
    rng = np.random.default_rng(4)
    B, W = 2, 500
    T = net.receptive_field + W - 1
    x = scrambled_input(rng.integers(0, 256, size=(B, T))).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).cuda()
    crit = torch.nn.CrossEntropyLoss()

    def seq_a(fuse):
        net.fuse_loss = fuse
        net.zero_grad()
        out = net(x)
        loss = 1.5 * crit(out, target)
        aux = (out * out).sum() / out.numel()
        aux.backward(retain_graph=True)
        g_aux = [p.grad.clone() for p in net.parameters()]
        loss.backward()
        return g_aux, [p.grad.clone() for p in net.parameters()]

    def seq_b(fuse):
        net.fuse_loss = fuse
        net.zero_grad()
        out = net(x)
        loss = 1.5 * crit(out, target)
        torch.autograd.grad(loss, out, retain_graph=True)        # the fused node's backward alone (its result: the zero token)
        aux = (out * out).sum() / out.numel()
        aux.backward(retain_graph=True)                          # must be aux's gradient only
        g_aux = [p.grad.clone() for p in net.parameters()]
        loss.backward()
        return g_aux, [p.grad.clone() for p in net.parameters()]

    for seq in (seq_a, seq_b):
        ref_aux, ref_all = seq(False)
        got_aux, got_all = seq(True)
        for ref, got in ((ref_aux, got_aux), (ref_all, got_all)):
            gmax = max(g.abs().max().item() for g in ref)
            for a, b in zip(ref, got):
                assert (a - b).abs().max().item() <= 5e-5 * max(a.abs().max().item(), 1e-3 * gmax), seq.__name__


def test_module_without_the_private_torch_hooks_runs_unfused(monkeypatch):
    """_losshook.AVAILABLE False (a torch without the internals the interception rests on): plain tensor out, torch's own loss, same numbers."""
    import numpy as np
    from music_amd import _losshook
    from tests.helpers import scrambled_input
    net = _small_net()
    rng = np.random.default_rng(5)
    B, W = 2, 300
    x = scrambled_input(rng.integers(0, 256, size=(B, net.receptive_field + W - 1))).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).cuda()
    crit = torch.nn.CrossEntropyLoss()
    res = []
    for avail in (True, False):
        monkeypatch.setattr(_losshook, "AVAILABLE", avail)
        net.zero_grad()
        out = net(x)
        assert (type(out) is _losshook.Probs) == avail
        loss = crit(out, target)
        loss.backward()
        res.append((float(loss.detach()), [p.grad.clone() for p in net.parameters()]))
    (l0, g0), (l1, g1) = res
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0))
    gmax = max(g.abs().max().item() for g in g0)
    for a, b in zip(g0, g1):
        assert (a - b).abs().max().item() <= 5e-5 * max(a.abs().max().item(), 1e-3 * gmax)


def test_cross_entropy_on_the_autoencoder_output_runs_fused_and_equals_torchs():
    """The same interception on `wavenet_autoencoder` (its train loop applies nn.CrossEntropyLoss to the output as well): loss and
    every gradient against the unfused module, with the same per-forward conditioning projections (same torch seed)."""
    import numpy as np
    from music_amd import _losshook
    from music_amd.model1 import wavenet_autoencoder
    from oracle import intops
    torch.manual_seed(11)
    net = wavenet_autoencoder(filter_width=2, quantization_channel=256, dilations=[1, 2, 4, 8, 16], en_residual_channel=64,
                              en_dilation_channel=64, en_bottleneck_width=16, en_pool_kernel_size=50, de_residual_channel=64,
                              de_dilation_channel=64, de_skip_channel=256, use_bias=False)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
        net.connection_2.weight.mul_(6.0)
    net = net.cuda()
    rng = np.random.default_rng(12)
    B, W = 2, 400
    idx = rng.integers(0, 256, size=(B, net.receptive_field + W - 1))
    x = torch.from_numpy(np.stack([intops.one_hot_proper(r) for r in idx])).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).cuda()
    crit = torch.nn.CrossEntropyLoss()
    res = []
    for fuse in (False, True):
        net.fuse_loss = fuse
        net.zero_grad()
        torch.manual_seed(77)
        out = net(x)
        assert (type(out) is _losshook.Probs) == fuse
        loss = 2.0 * crit(out, target)
        if fuse:
            assert out._wn_hook.fused
        loss.backward()
        res.append((float(loss.detach()), [p.grad.clone() for p in net.parameters()]))
    (l0, g0), (l1, g1) = res
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0)), (l0, l1)
    gmax = max(g.abs().max().item() for g in g0)
    for a, b in zip(g0, g1):
        assert (a - b).abs().max().item() <= 5e-5 * max(a.abs().max().item(), 1e-3 * gmax)

