"""Full-size parity (VERDICT r1 "next" #2): the BASELINE configurations at the sizes the benchmark runs them, not
at the few-thousand-column sizes of tests/test_gpu_parity.py.  The persistent backward kernels walk item ranges,
per-XCD interleaved orders and slab counts that depend on the size, so gradients are compared at T = 16000 too.

  (a) config 2   30 blocks, 64/64/256, B = 2, T = 16000        loss + every gradient vs the oracle's autograd
  (b) config 4   autoencoder 30 + 30 blocks, 64 ch, pool 512   probabilities + every gradient vs the oracle
                 (Le = 25: the stretch / tile mix of SURVEY Q9)
  (c) config 5   30-block decoder                               1100+ teacher-forced steps (the d = 512 rings wrap
                 twice) vs the oracle's cached-queue recurrence, both recurrences; a 16 000-step free run whose
                 first 1100 codes are what the oracle predicts from the same history
  (d) the reference's shipped wavenet_params.json (40 blocks, 32/32/512, rf 4094): forward + gradients at
      window 4000, and the shipped batch (4 x 44093, window 40000) against an oracle window

Run with -m gpu.  Probabilities: 1e-3 absolute on gain-scaled weights, integers exact, as tests/test_gpu_parity.py.

Gradients at full size, and what the comparison has to hold fixed (round 3, tools/diag_stage.py + tools/emu_bwd.py).
Round 2 compared with the float64 oracle at 3e-2 of a tensor's max-abs, because this path sat 2.5e-3 off float64 on a
well-conditioned ("aligned") batch where the float32 CPU path sits 8e-6 off, and blamed the bf16 hi/lo operands of the
backward.  That was the wrong culprit.  Measured on the device, every backward kernel reproduces float64 evaluated ON
ITS OWN INPUTS to 5e-6 ... 1.5e-5 of its output's max-abs, and the operand-rounding emulation of the whole backward in
float64 puts the bf16 split at 3e-5 (the float32 CPU path: 2.7e-5).  The 2.5e-3 comes from 14 + 16 of the 13 M
pre-activations of the two post-processing ReLUs (model.py:135,137) that lie within 8e-6 of zero: this path's forward
and the float64 forward disagree on their SIGN (both are within 1e-5 of each other there - any two correct float32
forwards do this; the float32 CPU path happened to flip none on that batch and flips plenty on others), and one
flipped element of dU moves every skip-weight gradient by ~1e-3 of its max-abs.  So the reference gradient is
evaluated with the subgradient of ReLU at |pre-activation| < RELU_EPS * max-abs chosen the way the device chose it
(oracle `relu=` hook; a mask that differs anywhere ELSE fails the test), and the bar drops from 3e-2 to 2e-4 of a
tensor's max-abs (or 3x the float32 CPU path's error on that tensor against float64 with ITS near-zero signs, which
only the saturated shipped configuration needs: measured 3.4e-5 / 4.0e-5 / 6.8e-5 on aligned config 2 / random config 2 /
config 4, where the float32 CPU path makes 1.6e-5).  What that bar cannot see - a
dropped tile at a clip or workgroup boundary moves a sum by ~1e-3 - is also caught by a
property with no conditioning in it: the batch gradient must equal the mean of the single-clip gradients, which the
persistent kernels compute with a different partition of items, slabs and XCD walks (1e-4)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import intops
from oracle import wavenet_oracle as wo
from tests.helpers import nonvacuous, scrambled_input

import os

LOGIT_TOL = 1e-3
FULL_GRAD_RTOL = 2e-4           # random-target batches (or 3x the float32 CPU path's own error on that tensor)
ALIGNED_GRAD_RTOL = 2e-4        # structured clip + constant target: the terms of every gradient sum line up
RELU_EPS = 2e-4                 # |pre-activation| below this fraction of the tensor's max-abs: the device's sign is taken
ORACLE_THREADS = min(32, os.cpu_count() or 1)      # ATen's CPU convs stop scaling (then collapse) beyond that
C2 = dict(filter_width=2, dilations=[2 ** i for i in range(10)] * 3, dilation_channels=64, residual_channels=64,
          skip_channels=256, quantization_channels=256, use_bias=False)
# /root/reference/wavenet/params/wavenet_params.json (values restated here: the file does not travel to the GPU box)
SHIPPED = dict(filter_width=2, dilations=[2 ** i for i in range(10)] * 4, dilation_channels=32, residual_channels=32,
               skip_channels=512, quantization_channels=256, use_bias=False)


def _scaled(net, gain):
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(gain)
    return {k: v.clone() for k, v in net.state_dict().items()}


def _check_grads(got, g64, e32, rtol=FULL_GRAD_RTOL):
    """got / g64: name -> gradient (this path, float64 oracle with the device's near-zero ReLU signs); e32: name -> the
    float32 CPU path's error on that tensor against float64 with ITS OWN near-zero signs (the like-for-like yardstick).
    A tensor passes within rtol of its max-abs or within 3x the float32 path's error.  Returns the worst (error, name,
    the float32 path's error on that tensor)."""
    worst = (0.0, None, 0.0)
    for name, ref in g64.items():
        if ref is None:
            assert got[name] is None or got[name].abs().max().item() == 0.0, name
            continue
        scale = max(ref.abs().max().item(), 1e-30)
        e_gpu = (got[name].detach().cpu().double() - ref).abs().max().item() / scale
        e_cpu = e32.get(name, 0.0)
        if e_gpu > worst[0]:
            worst = (e_gpu, name, e_cpu)
        assert e_gpu <= max(rtol, 3.0 * e_cpu), (name, e_gpu, e_cpu)
    return worst


def _device_relu(dev_pre):
    """Oracle `relu=` hook that follows the DEVICE's sign where the reference pre-activation is within RELU_EPS of zero
    (relative to the tensor's max-abs) and insists on equal signs everywhere else.  dev_pre: name -> the device's own
    pre-activation (CPU tensor, the reference's shape).  Returns (hook, stats)."""
    stats = dict(near=0, flips=0)

    def relu(name, t):
        d = dev_pre[name]
        assert d.shape == t.shape, (name, d.shape, t.shape)
        eps = RELU_EPS * t.detach().abs().max().item()
        near = t.detach().abs() < eps
        ref_m, dev_m = t.detach() > 0, d > 0
        assert not ((ref_m != dev_m) & ~near).any(), "ReLU mask of %s differs outside the tolerance band" % name
        stats["near"] += int(near.sum())
        stats["flips"] += int(((ref_m != dev_m) & near).sum())
        return t * torch.where(near, dev_m, ref_m).to(t.dtype)
    return relu, stats


def _c2_dev_pre(eng, ws):
    """the device's pre-ReLU skip sum and post_process_1 output, (B, S, W) each"""
    from music_amd.engine import SLACK
    B, pitch, T, lo = ws["B"], ws["pitch"], ws["T"], eng.rf - 1
    v = lambda buf: buf[SLACK:SLACK + B * eng.SP * pitch].view(B, eng.SP, pitch)[:, :eng.S, lo:T].cpu()
    return {"skip_sum": v(ws["U"]), "post_process_1": v(ws["H"])}


def _c4_dev_pre(aeng, B, T):
    """the autoencoder's pre-ReLU tensors on the device, in the oracle's shapes and names (autoencoder_encode / _decode)"""
    from music_amd.engine import SLACK
    ws = aeng.workspace(B, T)
    pitch, lo, N = ws["pitch"], aeng.rf - 1, aeng.N
    xe = ws["Xe"][SLACK:SLACK + (N + 1) * B * aeng.CHe * pitch].view(N + 1, B, aeng.CHe, pitch)
    he = ws["He"][SLACK:SLACK + N * B * aeng.CHe * pitch].view(N, B, aeng.CHe, pitch)
    pre = {}
    for i in range(N):
        pre["en_x%d" % i] = xe[i][:, :aeng.Re, aeng.off[i]:T].cpu()
        pre["en_h%d" % i] = he[i][:, :aeng.De, aeng.off[i + 1]:T].cpu()
    v = lambda buf: buf[SLACK:SLACK + B * aeng.SP * pitch].view(B, aeng.SP, pitch)[:, :aeng.Sd, lo:T].cpu()
    pre["de_skip"], pre["de_conn"] = v(ws["U"]), v(ws["R1"])
    return pre


def _oracle_grads_f32_f64(params, dilations, x, target, dev_pre):
    """-> (float32 loss, float32 probs, e32, float64 loss, g64): g64 = float64 gradients with the DEVICE's signs at the
    near-zero ReLU pre-activations; e32[name] = error of the float32 CPU path against float64 with the float32 path's
    own signs there (two float64 passes)."""
    torch.set_num_threads(ORACLE_THREADS)
    inter = {}
    l32, p32, g32 = wo.loss_and_grads(params, dilations, x, target, intermediates=inter)
    cpu_pre = {k: inter[k].detach() for k in ("skip_sum", "post_process_1")}
    del inter
    p64 = {k: v.double() for k, v in params.items()}
    relu, stats = _device_relu(dev_pre)
    l64, _, g64 = wo.loss_and_grads(p64, dilations, x.double(), target, relu=relu)
    relu_c, stats_c = _device_relu(cpu_pre)
    _, _, g64c = wo.loss_and_grads(p64, dilations, x.double(), target, relu=relu_c)
    e32 = {n: (g32[n].double() - g64c[n]).abs().max().item() / max(g64c[n].abs().max().item(), 1e-30) for n in g64c}
    print("  ReLU pre-activations inside the tolerance band: %d; sign differs from float64's on the device at %d, in the float32 CPU path at %d" %
          (stats["near"], stats["flips"], stats_c["flips"]))
    return l32, p32, e32, l64, g64


def test_c2_full_length_loss_and_gradients_vs_oracle():
    """(a) config 2 at B = 2, T = 16000 through eng.loss_and_grad on the default kernels."""
    from music_amd.model import wavenet
    torch.manual_seed(3)
    net = wavenet(**C2)
    params = _scaled(net, 2.5)
    net = net.cuda()
    rng = np.random.default_rng(31)
    B, T = 2, 16000
    rf = net.receptive_field
    W = T - rf + 1
    codes = rng.integers(0, 256, size=(B, T))
    x = scrambled_input(codes)
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))
    eng = net._engine_for(torch.device("cuda", 0))
    xd = eng.onehot(torch.from_numpy(codes.astype(np.int32)).cuda(), scrambled=True)
    assert torch.equal(xd.cpu(), x)
    loss = eng.loss_and_grad(xd, target.cuda(), want_probs=True)
    probs = eng.workspace(B, T)["probs"].cpu()
    got = {n: eng.param_view(n, grad=True).clone() for n in eng.param_names}
    l_ref, p_ref, g32, l64, g64 = _oracle_grads_f32_f64(params, C2["dilations"], x, target, _c2_dev_pre(eng, eng.workspace(B, T)))
    e_p = (probs - p_ref).abs().max().item()
    assert probs.shape == (B * W, 256) and e_p <= LOGIT_TOL, e_p
    nonvacuous(p_ref, "c2 at 2 x 16000", 0.5)            # 
    assert abs(loss.item() - l_ref.item()) < 1e-4 and abs(loss.item() - l64.item()) < 1e-4
    worst, name, cpu = _check_grads(got, g64, g32)
    print("c2 full length (2 x 16000): probs err %.2e, loss %.7f (oracle f32 %.7f, f64 %.7f), worst grad err vs f64 %.2e "
          "(%s; the float32 CPU path: %.2e)" % (e_p, loss.item(), l_ref.item(), l64.item(), worst, name, cpu))
    # the same step again: weight gradients are bit-reproducible (slab sums in a fixed order, no float atomics)
    g1 = eng.flat_grad.clone()
    eng.loss_and_grad(xd, target.cuda())
    assert torch.equal(g1, eng.flat_grad)
    # partition independence: batch gradient == mean of the single-clip gradients (other item / slab / XCD partitions)
    acc = torch.zeros_like(g1)
    cdev = torch.from_numpy(codes.astype(np.int32)).cuda()
    for b in range(B):             # (through the codes, as the batch above: the tagged one-hot takes the code-aware causal layer)
        eng.loss_and_grad_codes(cdev[b:b + 1].contiguous(), target.view(B, W)[b].contiguous().cuda(), scrambled=True)
        acc += eng.flat_grad
    acc /= B
    for n in eng.param_names:
        o, shp = eng.spec.off[n], eng.spec.shape[n]
        k = int(np.prod(shp))
        a1, a2 = g1[o:o + k], acc[o:o + k]
        assert (a1 - a2).abs().max().item() <= 1e-4 * max(a1.abs().max().item(), 1e-30), n


def test_c2_bench_geometry_gradients_vs_oracle():
    """Config 2 at the BENCHMARK's geometry, 8 x 16000 through eng.loss_and_grad_codes (what bench.py times): loss,
    probabilities and EVERY gradient against the oracle.  The persistent backward's item ranges, chain segments, slab
    counts and XCD walks depend on the batch; the 2-clip tests above do not reach this partition.  The oracle runs clip by
    clip (chunk-softmax rows never cross clips, SURVEY Q2, and all clips have W rows, so the batch loss is the mean of the
    clip losses and the batch gradient the mean of the clip gradients): float32 for loss / probabilities, float64 with the
    device's sign at near-zero ReLU pre-activations for the gradients (module docstring), 3 GB of host memory at a time."""
    from music_amd.model import wavenet
    torch.manual_seed(3)
    net = wavenet(**C2)
    params = _scaled(net, 2.5)
    net = net.cuda()
    rng = np.random.default_rng(33)
    B, T = 8, 16000
    W = T - net.receptive_field + 1
    codes = rng.integers(0, 256, size=(B, T))
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))
    eng = net._engine_for(torch.device("cuda", 0))
    loss = eng.loss_and_grad_codes(torch.from_numpy(codes.astype(np.int32)).cuda(), target.cuda(), scrambled=True, want_probs=True)
    ws = eng.workspace(B, T)
    assert any(ws["bwd"]["chain"]) and ws["bwd"]["pq"]          # the default kernels: one-launch blocks, chain form for d >= 32
    probs = ws["probs"].cpu()
    got = {n: eng.param_view(n, grad=True).clone() for n in eng.param_names}
    dev_pre = _c2_dev_pre(eng, ws)
    torch.set_num_threads(ORACLE_THREADS)
    p64 = {k: v.double() for k, v in params.items()}
    g64 = {k: torch.zeros_like(v) for k, v in p64.items()}
    l32 = l64 = 0.0
    e_p = p_max = 0.0
    near = flips = 0
    for b in range(B):
        xb = scrambled_input(codes[b:b + 1])
        tb = target[b * W:(b + 1) * W]
        with torch.no_grad():
            pb = wo.wavenet_forward(params, C2["dilations"], xb)
        l32 += torch.nn.functional.cross_entropy(pb, tb).item() / B
        e_p = max(e_p, (probs[b * W:(b + 1) * W] - pb).abs().max().item())
        p_max = max(p_max, pb.max().item())
        relu, stats = _device_relu({k: v[b:b + 1] for k, v in dev_pre.items()})
        lb, _, gb = wo.loss_and_grads(p64, C2["dilations"], xb.double(), tb, relu=relu)
        l64 += lb.item() / B
        near, flips = near + stats["near"], flips + stats["flips"]
        for k in g64:
            g64[k] += gb[k] / B
    assert e_p <= LOGIT_TOL, e_p
    nonvacuous(torch.tensor(p_max), "c2 at 8 x 16000", 0.5)
    assert abs(loss.item() - l32) < 1e-4 and abs(loss.item() - l64) < 1e-4, (loss.item(), l32, l64)
    worst, name, _ = _check_grads(got, g64, {})
    print("c2 at the bench geometry (8 x 16000, codes path): probs err %.2e, loss %.7f (oracle f32 %.7f, f64 %.7f), worst grad err vs f64 "
          "%.2e (%s); %d ReLU pre-activations inside the tolerance band, device sign differs at %d" % (e_p, loss.item(), l32, l64, worst, name, near, flips))


def _aligned_batch(B, T, W):
    """structured clip (period-7 pattern, a different phase per clip) + constant target: the terms of every gradient
    sum line up instead of cancelling, so the comparison resolves 2^-17 from 2^-22 arithmetic (tools/diag_fullsize.py)"""
    codes = (np.arange(T)[None, :] * 37 % 7 * 31 + 11 + np.arange(B)[:, None]) % 256
    return codes, torch.from_numpy(np.full((B * W,), 7, dtype=np.int64))


@pytest.mark.parametrize("case", ["c2", "shipped"])
def test_aligned_full_length_gradients_vs_oracle(case):
    """The discriminating gradient test: a well-conditioned full-length batch, every gradient
    within ALIGNED_GRAD_RTOL = 2e-4 of its tensor's max-abs against the float64 oracle."""
    from music_amd.model import wavenet
    cfg, gain, seed, T = (C2, 2.5, 3, 16000) if case == "c2" else (SHIPPED, 3.0, 7, 4094 + 3999)
    torch.manual_seed(seed)
    net = wavenet(**cfg)
    params = _scaled(net, gain)
    net = net.cuda()
    B = 2
    W = T - net.receptive_field + 1
    codes, target = _aligned_batch(B, T, W)
    x = scrambled_input(codes)
    eng = net._engine_for(torch.device("cuda", 0))
    loss = eng.loss_and_grad(x.cuda(), target.cuda(), want_probs=True)
    got = {n: eng.param_view(n, grad=True).clone() for n in eng.param_names}
    l_ref, p_ref, g32, l64, g64 = _oracle_grads_f32_f64(params, cfg["dilations"], x, target, _c2_dev_pre(eng, eng.workspace(B, T)))
    assert abs(loss.item() - l64.item()) < 1e-5
    worst, name, cpu = _check_grads(got, g64, g32, ALIGNED_GRAD_RTOL)
    print("aligned %s (2 x %d): loss %.7f (f64 %.7f), worst grad err vs f64 %.2e (%s; the float32 CPU path on that tensor: %.2e)" %
          (case, T, loss.item(), l64.item(), worst, name, cpu))


def test_gradient_operands_have_no_range_limit():
    """The gradient operands of the backward are split in bf16, which has float32's exponent range: a loss scaled by
    2^-32 ... 2^+32 must give the SAME gradient bits times that power of two (an f16 split would underflow / overflow;
    VERDICT r2 next #1c asks for exactly this guard), on gain-scaled weights whose |d logits| span more than 2^20."""
    from music_amd.model import wavenet
    torch.manual_seed(11)
    net = wavenet(**C2)
    _scaled(net, 3.0)
    net = net.cuda()
    B, T = 1, 4000
    rng = np.random.default_rng(12)
    eng = net._engine_for(torch.device("cuda", 0))
    x = eng.onehot(torch.from_numpy(rng.integers(0, 256, size=(B, T)).astype(np.int32)).cuda(), scrambled=True)
    W = T - net.receptive_field + 1
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).cuda()
    probs, ws = eng.forward(x)
    p = probs.detach().clone().requires_grad_(True)
    torch.nn.functional.cross_entropy(p, target).backward()
    dprobs = p.grad
    eng.backward(ws, dprobs)
    g0 = eng.flat_grad.clone()
    dO = ws["bwd"]["dO"][:B * 256 * W].abs()
    span = (dO.max() / dO[dO > 0].min()).log2().item()
    assert span >= 20, span
    for k in (-32, -16, 16, 32):
        eng.backward(ws, dprobs * 2.0 ** k)
        assert torch.equal(eng.flat_grad * 2.0 ** (-k), g0), k
    print("gradient bits invariant under loss scales 2^-32 .. 2^32; |d logits| span 2^%.1f" % span)


def test_c4_full_size_autoencoder_vs_oracle():
    """(b) config 4: 30 + 30 blocks, 64 channels, bottleneck 64, pool 512 => Le = 25 pooled frames at T = 16000:
    decoder layers whose length is a multiple of 25 take the stretch branch of _conditon, the others the tile branch."""
    from music_amd.model1 import wavenet_autoencoder
    cfg = dict(filter_width=2, quantization_channel=256, dilations=C2["dilations"], en_residual_channel=64,
               en_dilation_channel=64, en_bottleneck_width=64, en_pool_kernel_size=512, de_residual_channel=64,
               de_dilation_channel=64, de_skip_channel=256, use_bias=False)
    torch.manual_seed(5)
    net = wavenet_autoencoder(**cfg)
    _scaled(net, 1.6)
    # 60 gated blocks in a row: a uniform gain makes the float32 reference itself irreproducible (at gain 3 float32 and float64
    # differ by 8e-3 on the probabilities, with max p still 0.06) long before the output becomes confident, so the last 1x1
    # conv alone takes the extra gain that makes the 1e-3 bar on the probabilities mean something (max p > 0.5, asserted below)
    with torch.no_grad():
        net.connection_2.weight.mul_(12.0)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rng = np.random.default_rng(51)
    B, T = 2, 16000
    rf = net.receptive_field
    W = T - rf + 1
    assert rf == 3071 and W // 512 == 25
    # which branch each decoder layer's conditioning takes: both must occur
    L, stretch = T - 1, []
    for d in cfg["dilations"]:
        L -= d
        stretch.append(L % 25 == 0)
    assert any(stretch) and not all(stretch), stretch
    idx = rng.integers(0, 256, size=(B, T))
    x = torch.from_numpy(np.stack([intops.one_hot_proper(r) for r in idx]))
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))
    torch.manual_seed(77)
    net.zero_grad()
    probs = net(x.cuda())
    loss = torch.nn.CrossEntropyLoss()(probs, target.cuda())
    loss.backward()
    dev_pre = _c4_dev_pre(net._engine_for(torch.device("cuda", 0)), B, T)
    torch.set_num_threads(ORACLE_THREADS)
    torch.manual_seed(77)
    cond = wo.draw_conditioning(len(cfg["dilations"]), cfg["en_bottleneck_width"], cfg["de_dilation_channel"],
                                cfg["de_skip_channel"])
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    p_ref, enc_ref = wo.autoencoder_forward(leaf, cfg["dilations"], x, cfg["en_pool_kernel_size"], cond)
    assert enc_ref.shape == (B, 64, 25)
    l_ref = torch.nn.functional.cross_entropy(p_ref, target)
    g32 = dict(zip(leaf.keys(), torch.autograd.grad(l_ref, list(leaf.values()), allow_unused=True)))
    leaf64 = {k: v.double().requires_grad_(True) for k, v in params.items()}
    relu, stats = _device_relu(dev_pre)
    p64, _ = wo.autoencoder_forward(leaf64, cfg["dilations"], x.double(), cfg["en_pool_kernel_size"],
                                    [(w.double(), b.double()) for w, b in cond], relu=relu)
    print("  ReLU pre-activations inside the tolerance band: %d, of which the device's sign differs: %d" % (stats["near"], stats["flips"]))
    l64 = torch.nn.functional.cross_entropy(p64, target)
    g64 = dict(zip(leaf64.keys(), torch.autograd.grad(l64, list(leaf64.values()), allow_unused=True)))
    e_enc = (net.last_encoding.cpu() - enc_ref.detach()).abs().max().item()
    e_p = (probs.detach().cpu() - p_ref.detach()).abs().max().item()
    assert e_enc < 1e-4 and e_p <= LOGIT_TOL, (e_enc, e_p)
    nonvacuous(p_ref.detach(), "c4 at 2 x 16000", 0.5)
    assert abs(loss.item() - l_ref.item()) < 1e-4
    got = {name: p.grad for name, p in net.named_parameters()}
    assert list(got.keys()) == list(g64.keys())
    worst, worst_name, cpu = _check_grads(got, g64, {})            # fixed bar (the float32 path is not consulted here)
    print("c4 full size (2 x 16000, Le 25, %d stretch / %d tile layers): enc err %.2e probs err %.2e (max p %.3f) "
          "worst grad err vs f64 %.2e (%s; the float32 CPU path: %.2e)" %
          (sum(stretch), len(stretch) - sum(stretch), e_enc, e_p, p_ref.max().item(), worst, worst_name, cpu))
    # partition independence with the SAME conditioning projections: batch gradient == mean of single-clip gradients
    aeng = net._engine_for(torch.device("cuda", 0))
    xd, td = x.cuda(), target.cuda()
    aeng.loss_and_grad(xd, td, cond)
    g_batch = aeng.flat_grad.clone()
    acc = torch.zeros_like(g_batch)
    for b in range(B):
        aeng.loss_and_grad(xd[b:b + 1].contiguous(), td.view(B, W)[b].contiguous(), cond)
        acc += aeng.flat_grad
    acc /= B
    for name, p in net.named_parameters():
        o = aeng.spec.off[name]
        a1, a2 = g_batch[o:o + p.numel()], acc[o:o + p.numel()]
        assert (a1 - a2).abs().max().item() <= 1e-4 * max(a1.abs().max().item(), 1e-30), name


def test_c4_bench_geometry_vs_oracle():
    """Config 4 at the BENCHMARK's geometry: autoencoder 30 + 30 blocks, 8 x 16000 - what bench.py's
    extra.c4_autoencoder times - loss, encoding, probabilities and EVERY gradient against the oracle.  The conditioned backward's
    per-clip bucket-sum slabs, pq_cond_reduce_k's slot order and the encoder's item ranges all depend on the batch; the 2-clip
    test above does not reach this partition.  The oracle runs clip by clip with the batch's ONE set of 31 conditioning draws
    (drawn once per forward, applied to every clip's encoding): chunk-softmax rows and the pooled encoding never cross
    clips and every clip has W rows, so the batch loss / gradient is the mean of the clip losses / gradients.  float32 for the
    probabilities and the encoding, float64 with the device's sign at near-zero ReLU pre-activations for the gradients."""
    from music_amd.model1 import wavenet_autoencoder
    cfg = dict(filter_width=2, quantization_channel=256, dilations=C2["dilations"], en_residual_channel=64,
               en_dilation_channel=64, en_bottleneck_width=64, en_pool_kernel_size=512, de_residual_channel=64,
               de_dilation_channel=64, de_skip_channel=256, use_bias=False)
    torch.manual_seed(6)
    net = wavenet_autoencoder(**cfg)
    _scaled(net, 1.6)
    with torch.no_grad():
        net.connection_2.weight.mul_(12.0)                    # (see test_c4_full_size_autoencoder_vs_oracle)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rng = np.random.default_rng(61)
    B, T = 8, 16000
    W = T - net.receptive_field + 1
    idx = rng.integers(0, 256, size=(B, T))
    x = torch.from_numpy(np.stack([intops.one_hot_proper(r) for r in idx]))
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))
    torch.manual_seed(78)
    net.zero_grad()
    probs = net(x.cuda())
    loss = torch.nn.CrossEntropyLoss()(probs, target.cuda())
    loss.backward()
    aeng = net._engine_for(torch.device("cuda", 0))
    dev_pre = _c4_dev_pre(aeng, B, T)
    probs = probs.detach().cpu()
    enc_dev = net.last_encoding.cpu()
    got = {name: p.grad for name, p in net.named_parameters()}
    torch.set_num_threads(ORACLE_THREADS)
    torch.manual_seed(78)
    cond = wo.draw_conditioning(len(cfg["dilations"]), cfg["en_bottleneck_width"], cfg["de_dilation_channel"], cfg["de_skip_channel"])
    cond64 = [(w.double(), b.double()) for w, b in cond]
    p64 = {k: v.double() for k, v in params.items()}
    g64 = {k: None for k in p64}
    l32 = l64 = e_p = e_enc = p_max = 0.0
    near = flips = 0
    for b in range(B):
        xb, tb = x[b:b + 1], target[b * W:(b + 1) * W]
        with torch.no_grad():
            pb, eb = wo.autoencoder_forward(params, cfg["dilations"], xb, cfg["en_pool_kernel_size"], cond)
        l32 += torch.nn.functional.cross_entropy(pb, tb).item() / B
        e_p = max(e_p, (probs[b * W:(b + 1) * W] - pb).abs().max().item())
        e_enc = max(e_enc, (enc_dev[b:b + 1] - eb).abs().max().item())
        p_max = max(p_max, pb.max().item())
        leaf = {k: v.clone().requires_grad_(True) for k, v in p64.items()}
        relu, stats = _device_relu({k: v[b:b + 1] for k, v in dev_pre.items()})
        pq, _ = wo.autoencoder_forward(leaf, cfg["dilations"], xb.double(), cfg["en_pool_kernel_size"], cond64, relu=relu)
        lb = torch.nn.functional.cross_entropy(pq, tb)
        gb = dict(zip(leaf.keys(), torch.autograd.grad(lb, list(leaf.values()), allow_unused=True)))
        l64 += lb.item() / B
        near, flips = near + stats["near"], flips + stats["flips"]
        for k, g in gb.items():
            if g is not None:
                g64[k] = g / B if g64[k] is None else g64[k] + g / B
        del leaf, pq, lb, gb
    assert e_enc < 1e-4 and e_p <= LOGIT_TOL, (e_enc, e_p)
    nonvacuous(torch.tensor(p_max), "c4 at 8 x 16000", 0.5)
    assert abs(loss.item() - l32) < 1e-4 and abs(loss.item() - l64) < 1e-4, (loss.item(), l32, l64)
    assert list(got.keys()) == list(g64.keys())
    worst, name, _ = _check_grads(got, g64, {})
    print("c4 at the bench geometry (8 x 16000): enc err %.2e, probs err %.2e (max p %.3f), loss %.7f (oracle f32 %.7f, f64 %.7f), worst grad "
          "err vs f64 %.2e (%s); %d ReLU pre-activations inside the tolerance band, device sign differs at %d" %
          (e_enc, e_p, p_max, loss.item(), l32, l64, worst, name, near, flips))


def _onehot(ix):
    return torch.from_numpy(intops.one_hot_proper(np.atleast_1d(ix)))[None]


@pytest.mark.parametrize("correct", [False, True], ids=["as_written", "corrected"])
def test_c5_full_size_decode_rings_wrap_vs_oracle(correct):
    """(c) config 5: 1100 teacher-forced steps on the matrix-core decoder vs the oracle's cached-queue recurrence
    (every ring, the three d = 512 ones included, wraps at least twice): argmax ids exact, probabilities 1e-4."""
    from music_amd import fast_generate as fg
    from music_amd.model import wavenet
    torch.manual_seed(21)
    net = wavenet(**C2)
    params = _scaled(net, 2.2)
    net = net.cuda()
    rng = np.random.default_rng(22)
    n = 1100
    start = rng.integers(0, 256, size=(net.receptive_field,))
    forced = rng.integers(0, 256, size=(n,))
    torch.set_num_threads(4)
    pred_o, q_o = wo.fast_predict_next(params, C2["dilations"], _onehot(start), None)
    want, want_p = [int(pred_o[0])], []
    for s in forced:
        pred_o, q_o, pr = wo.fast_predict_next(params, C2["dilations"], _onehot(s), q_o, correct_queue=correct, return_probs=True)
        want.append(int(pred_o[0]))
        want_p.append(pr.numpy())
    want_p = np.stack(want_p)
    pred, st = fg.predict_next(net, _onehot(start).cuda(), None)
    got = [int(pred[0])]
    nxt = torch.from_numpy(np.concatenate([forced[1:], [0]]).astype(np.int32))
    codes, probs, _ = fg._decode(net, st, _onehot(forced[0]).reshape(-1).cuda(), n, forced=nxt, want_probs=True,
                                 correct_queue=correct)
    got += codes.cpu().tolist()
    probs = probs.cpu().numpy()
    err = np.abs(probs - want_p).max()
    # an argmax may only differ where the oracle's own top two are closer than the tolerance
    for k, (a, b) in enumerate(zip(got[1:], want[1:])):
        if a != b:
            assert abs(want_p[k][a] - want_p[k][b]) < 1e-5, (k, a, b)
    print("config-5 decode, %d teacher-forced steps (correct_queue=%s): probs err %.2e, %d/%d ids equal" %
          (n, correct, err, sum(a == b for a, b in zip(got, want)), len(want)))
    assert err < 1e-4
    for i in (9, 19, 29):                                  # the d = 512 rings after two wraps
        np.testing.assert_allclose(st["block_%d" % (i + 1)].cpu().numpy(), q_o["block_%d" % (i + 1)].numpy(), atol=2e-4, rtol=0)


def test_c5_16000_step_free_run_first_codes_vs_oracle():
    """(c) one 16 000-sample greedy free run (1 s of audio, BASELINE config 5) in one launch: deterministic, and its
    first 1100 codes are what the oracle's recurrence predicts when it is fed the same history."""
    from music_amd import fast_generate as fg
    from music_amd.model import wavenet
    torch.manual_seed(23)
    net = wavenet(**C2)
    params = _scaled(net, 2.2)
    net = net.cuda()
    start = np.full((net.receptive_field,), 128)           # fast_generate.py:159-160: class-128 start piece
    codes = fg.generate_codes(net, _onehot(start).cuda(), 16000).cpu().view(-1)
    again = fg.generate_codes(net, _onehot(start).cuda(), 16000).cpu().view(-1)
    assert codes.shape == (16000,) and torch.equal(codes, again)
    assert int(codes.min()) >= 0 and int(codes.max()) < 256 and len(torch.unique(codes)) > 4
    torch.set_num_threads(4)
    n = 1100
    pred_o, q_o = wo.fast_predict_next(params, C2["dilations"], _onehot(start), None)
    assert int(pred_o[0]) == int(codes[0])
    mism = 0
    for k in range(n - 1):
        pred_o, q_o, pr = wo.fast_predict_next(params, C2["dilations"], _onehot(int(codes[k])), q_o, return_probs=True)
        if int(pred_o[0]) != int(codes[k + 1]):
            assert abs(pr[int(pred_o[0])] - pr[int(codes[k + 1])]).item() < 1e-5, k
            mism += 1
    print("config-5 free run: 16000 codes, %d distinct; first %d agree with the oracle (%d near-ties)" %
          (len(torch.unique(codes)), n, mism))
    assert mism <= 2


def test_shipped_config_forward_and_gradients_vs_oracle():
    """(d) the reference's own wavenet_params.json: 40 blocks, 32 residual / dilation channels, 512 skip channels
    (rf 4094) at window 4000, B = 2: probabilities, loss and every gradient."""
    from music_amd.model import wavenet
    torch.manual_seed(7)
    net = wavenet(**SHIPPED)
    params = _scaled(net, 3.0)
    net = net.cuda()
    assert net.receptive_field == 4094
    rng = np.random.default_rng(71)
    B, W = 2, 4000
    T = net.receptive_field + W - 1
    codes = rng.integers(0, 256, size=(B, T))
    x = scrambled_input(codes)
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))
    eng = net._engine_for(torch.device("cuda", 0))
    loss = eng.loss_and_grad(x.cuda(), target.cuda(), want_probs=True)
    probs = eng.workspace(B, T)["probs"].cpu()
    got = {n: eng.param_view(n, grad=True).clone() for n in eng.param_names}
    l_ref, p_ref, g32, l64, g64 = _oracle_grads_f32_f64(params, SHIPPED["dilations"], x, target, _c2_dev_pre(eng, eng.workspace(B, T)))
    e_p = (probs - p_ref).abs().max().item()
    assert e_p <= LOGIT_TOL, e_p
    nonvacuous(p_ref, "shipped 40-block model", 0.3)
    assert abs(loss.item() - l_ref.item()) < 1e-4
    worst, name, cpu = _check_grads(got, g64, g32)
    print("shipped config (40 blocks, 32/32/512, 2 x %d): probs err %.2e (max p %.3f), worst grad err vs f64 %.2e "
          "(%s; the float32 CPU path: %.2e)" % (T, e_p, p_ref.max().item(), worst, name, cpu))


def test_shipped_config_shipped_batch_window_vs_oracle():
    """(d) the shipped dataset_params.json shape: batch 4, window 40000 (T = 44093).  Size-independent properties of
    the full batch plus one oracle window of 700 outputs cut from the middle of a clip."""
    from music_amd.model import wavenet
    torch.manual_seed(8)
    net = wavenet(**SHIPPED)
    _scaled(net, 3.0)
    net = net.cuda()
    rf, B, W = net.receptive_field, 4, 40000
    T = rf + W - 1
    rng = np.random.default_rng(81)
    codes = torch.from_numpy(rng.integers(0, 256, size=(B, T)).astype(np.int32)).cuda()
    eng = net._engine_for(torch.device("cuda", 0))
    x = eng.onehot(codes, scrambled=True)
    with torch.no_grad():
        p = net(x)
        assert p.shape == (B * W, 256) and torch.isfinite(p).all()
        assert (p.sum(1) - 1).abs().max().item() < 1e-5 and p.min().item() >= 0
        assert torch.equal(p, net(x))
        assert torch.equal(net(eng.onehot(codes[1:3].contiguous(), scrambled=True)), p[W:3 * W])   # same code-aware path
        assert (net(x[1:3].contiguous()) - p[W:3 * W]).abs().max().item() < 2 * LOGIT_TOL          # the dense path (each within 1e-3 of the oracle)
        xs = x[2:3, :, 20000:20000 + rf + 699].contiguous()
        got = net(xs).cpu()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    want = wo.wavenet_forward(sd, SHIPPED["dilations"], xs.cpu())
    err = (got - want).abs().max().item()
    print("shipped config at 4 x %d: window probs err %.2e" % (T, err))
    assert err <= LOGIT_TOL
    nonvacuous(want, "shipped model, window of the shipped batch", 0.3)
    # a full training step at this shape runs and is bit-reproducible
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).cuda()
    l1 = eng.loss_and_grad(x, target).item()
    g1 = eng.flat_grad.clone()
    l2 = eng.loss_and_grad(x, target).item()
    assert l1 == l2 and torch.equal(g1, eng.flat_grad) and np.isfinite(l1)


def test_c2_addressing_beyond_4gb():
    """Config 2's model on FOUR IDENTICAL clips of 300 000 samples: the stacked z / dz tensors are 9.2 GB each and the residual streams 9.5 GB, so
    clips 2 and 3 live wholly beyond byte offset 2^32 in them (tools/big_offsets.py).  Every clip's probabilities must equal clip 0's AND the
    one-clip run's bit for bit, and the gradient of the mean loss the one-clip run's to summation-order rounding (four equal clips scale every
    16-bit operand split by an exact power of two).  A 32-bit offset in any kernel or launcher of the step shows up here and in no 8 x 16000 case."""
    from music_amd.model import wavenet
    torch.manual_seed(0)
    net = wavenet(**C2)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.5)
    net = net.cuda()
    eng = net._engine_for(torch.device("cuda", 0))
    rng = np.random.default_rng(5)
    T, B = 300000, 4
    W = T - net.receptive_field + 1
    one = rng.integers(0, 256, size=(1, T)).astype(np.int32)
    tgt = rng.integers(0, 256, size=(W,)).astype(np.int64)
    res = {}
    for b in (1, B):
        codes = torch.from_numpy(np.repeat(one, b, axis=0)).cuda()
        target = torch.from_numpy(np.tile(tgt, b)).cuda()
        loss = eng.loss_and_grad_codes(codes, target, scrambled=True, want_probs=True)
        res[b] = (float(loss), eng.flat_grad.clone(), eng.workspace(b, T)["probs"].reshape(b, -1).clone())
        eng._ws.clear()
        torch.cuda.empty_cache()
    (l1, g1, p1), (lb, gb, pb) = res[1], res[B]
    assert torch.isfinite(gb).all() and torch.isfinite(pb).all()
    for k in range(B):
        assert torch.equal(pb[k], p1[0]), "probabilities of clip %d differ" % k
    assert abs(l1 - lb) < 1e-5
    rel = float((gb - g1).abs().max() / g1.abs().max())
    print("  4 x 300000: gradient of the four-clip run against the one-clip run: %.1e of its max" % rel)
    assert rel < 2e-6
