#!/usr/bin/env python3
"""Randomised parity fuzz (GPU): random WaveNet shapes vs the CPU oracle.

    python tools/fuzz_parity.py [--cases N] [--seed S]

Every case draws dilations, channel counts (ragged: not multiples of 16), skip width, batch, clip
length and bias at random, runs forward + CE + backward through the HIP path twice (nn.Module autograd
surface and the fused training-step entry) and checks pre-softmax logits / probabilities (1e-3), loss
(1e-4), every gradient (3e-4 of its tensor's max) and the gradient w.r.t. a dense input against oracle/wavenet_oracle.py.  A case whose forward
agrees but whose float32 gradients differ is judged again against the float64 oracle with the device's sign at the
post-processing ReLUs' near-zero pre-activations (the derivative jumps there; tests/test_gpu_fullsize.py) - no case is
skipped.  Test infrastructure (it imports oracle/); not part of the product path."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import wavenet_oracle as wo  # noqa: E402
from tests.helpers import scrambled_input  # noqa: E402

RELU_EPS = 2e-4          # tests/test_gpu_fullsize.py: the band around zero inside which a ReLU's subgradient follows the device


GAIN = 2.5               # --gain: every weight times this (default init leaves every probability at 1 / 256)
EPI = False              # --epi: every case has 256 skip channels (the fused epilogue launches), most of them no biases and 3 or 6 blocks


def one_case(rng, k, only=None):
    from music_amd.model import wavenet
    n = int(rng.integers(1, 7))
    dil = [int(rng.choice([1, 2, 3, 4, 5, 8, 16, 31, 64, 100, 256, 512])) for _ in range(n)]
    wide = rng.random() < 0.5
    R = int(rng.integers(33, 65)) if wide else int(rng.integers(4, 33))
    D = int(rng.integers(33, 65)) if wide else int(rng.integers(4, 33))
    S = int(rng.choice([8, 24, 33, 64, 100, 256]))
    B = int(rng.integers(1, 4))
    extra = int(rng.choice([0, 1, 3, 17, 63, 64, 65, 255, 511, 513, 1025]))
    bias = bool(rng.random() < 0.4)
    if EPI:
        S = 256
        if rng.random() < 0.7:              # the fused BACKWARD launch wants the stacked z rows in groups of 48 and no biases
            bias = False
            dil = (dil * 6)[:3 if rng.random() < 0.5 else 6]
    cfg = dict(filter_width=2, dilations=dil, dilation_channels=D, residual_channels=R, skip_channels=S,
               quantization_channels=256, use_bias=bias)
    torch.manual_seed(1000 + k)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(GAIN)
    params = {kk: v.clone() for kk, v in net.state_dict().items()}
    net = net.cuda()
    T = net.receptive_field + extra
    W = extra + 1
    x = scrambled_input(rng.integers(0, 256, size=(B, T)))
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))
    if only is not None and k != only:          # same random stream, no compute
        return True
    l_ref, p_ref, g_ref = wo.loss_and_grads(params, dil, x, target)
    inter = {}
    with torch.no_grad():
        wo.wavenet_forward(params, dil, x, intermediates=inter)
    floor = max(1e-3 * max(g.abs().max().item() for g in g_ref.values()), 1e-30)
    probs = net(x.cuda())
    eng = net._engine
    pre = eng.workspace(B, T)["O"][:B * 256 * W].view(B, 256, W).cpu()
    e_pre = (pre - inter["pre_softmax"].reshape(B, 256, W)).abs().max().item()
    e_p = (probs.detach().cpu() - p_ref).abs().max().item()
    loss = torch.nn.CrossEntropyLoss()(probs, target.cuda())
    loss.backward()
    worst = 0.0
    for name, p in net.named_parameters():
        g = g_ref[name]
        worst = max(worst, (p.grad.cpu() - g).abs().max().item() / max(g.abs().max().item(), floor))
    eng._ws.clear()
    for rep in range(2):                       # twice: stale-workspace hazards only show on the second call
        loss2 = eng.loss_and_grad(x.cuda(), target.cuda())
    worst2 = 0.0
    for name in eng.param_names:
        g = g_ref[name]
        e = (eng.param_view(name, grad=True).cpu() - g).abs().max().item() / max(g.abs().max().item(), floor)
        if only is not None and e > 3e-4:
            print("   %-40s rel err %.2e" % (name, e))
        worst2 = max(worst2, e)
    nonfinite = int((~torch.isfinite(eng.flat_grad)).sum()) + int((~torch.isfinite(probs)).sum())
    if nonfinite and all(torch.isfinite(g).all() for g in g_ref.values()):
        print("NONFINITE case %3d: %d non-finite device values where the oracle is finite" % (k, nonfinite), flush=True)
    ok = (e_pre <= 1e-3 and e_p <= 1e-3 and abs(loss.item() - l_ref.item()) < 1e-4 and abs(loss2.item() - l_ref.item()) < 1e-4
          and worst <= 3e-4 and worst2 <= 3e-4)
    tie_note = ""
    if not ok and e_pre <= 1e-3 and e_p <= 1e-3:
        # A post-processing ReLU on a pre-activation within rounding of 0 with opposite signs on the two sides makes the float32 gradients
        # differ legitimately (the derivative jumps).  Such a case is JUDGED, not skipped: the gradient is evaluated again
        # in float64 with the DEVICE's sign wherever the reference pre-activation lies within RELU_EPS of zero (relative to the tensor's
        # max-abs) - a mask that differs anywhere else fails the case - and both entry points are held to the same 3e-4 bar.
        from music_amd.engine import SLACK
        ws = eng.workspace(B, T)
        pitch, lo = ws["pitch"], eng.rf - 1
        v = lambda buf: buf[SLACK:SLACK + B * eng.SP * pitch].view(B, eng.SP, pitch)[:, :S, lo:T].cpu()
        dev_pre = {"skip_sum": v(ws["U"]), "post_process_1": v(ws["H"])}
        stats = dict(near=0, flips=0)

        def relu(name, t):
            d = dev_pre[name]
            assert d.shape == t.shape, (name, d.shape, t.shape)
            near = t.detach().abs() < RELU_EPS * t.detach().abs().max().item()
            ref_m, dev_m = t.detach() > 0, d > 0
            assert not ((ref_m != dev_m) & ~near).any(), "ReLU mask of %s differs outside the tolerance band" % name
            stats["near"] += int(near.sum())
            stats["flips"] += int(((ref_m != dev_m) & near).sum())
            return t * torch.where(near, dev_m, ref_m).to(t.dtype)
        try:
            l64, _, g64 = wo.loss_and_grads({kk: vv.double() for kk, vv in params.items()}, dil, x.double(), target, relu=relu)
        except AssertionError as e:
            print("FAIL case %3d  %s" % (k, e), flush=True)
            return False
        floor = 1e-3 * max(g.abs().max().item() for g in g64.values() if g is not None)
        worst = worst2 = 0.0
        for name, p in net.named_parameters():
            g = g64[name] if g64[name] is not None else torch.zeros_like(p, dtype=torch.float64, device="cpu")
            den = max(g.abs().max().item(), floor)
            worst = max(worst, (p.grad.cpu().double() - g).abs().max().item() / den)
            worst2 = max(worst2, (eng.param_view(name, grad=True).cpu().double() - g).abs().max().item() / den)
        ok = abs(loss.item() - l64.item()) < 1e-4 and abs(loss2.item() - l64.item()) < 1e-4 and worst <= 3e-4 and worst2 <= 3e-4
        tie_note = "  [float64 oracle with the device's sign at %d of %d near-zero ReLU pre-activations]" % (stats["flips"], stats["near"])
    # the gradient w.r.t. the INPUT (a dense float tensor that requires grad: the causal nn.Conv1d's data gradient, model.py:104), when the
    # float32 comparison stands on its own (no ReLU tie in this case)
    e_in = float("nan")
    if ok and not tie_note:
        xi = x.cuda().clone().requires_grad_(True)
        net.zero_grad()
        torch.nn.CrossEntropyLoss()(net(xi), target.cuda()).backward()
        xr = x.clone().requires_grad_(True)
        (g_in,) = torch.autograd.grad(torch.nn.functional.cross_entropy(wo.wavenet_forward(params, dil, xr), target), [xr])
        e_in = (xi.grad.cpu() - g_in).abs().max().item() / max(g_in.abs().max().item(), 1e-30)
        if e_in > 3e-4:
            # the same ReLU-tie rule for the input gradient (round 6, the 3e-4 bar): one flipped post-processing ReLU that the weight
            # gradients (sums over all rows) absorb under the bar can still move a column of d loss / d input by more - judged against
            # the float64 oracle with the device's sign inside the band, like the weight gradients above
            from music_amd.engine import SLACK
            ws = eng.workspace(B, T)
            pitch, lo = ws["pitch"], eng.rf - 1
            v = lambda buf: buf[SLACK:SLACK + B * eng.SP * pitch].view(B, eng.SP, pitch)[:, :S, lo:T].cpu()
            dev_pre = {"skip_sum": v(ws["U"]), "post_process_1": v(ws["H"])}
            stats = dict(near=0, flips=0)

            def relu_in(name, t):
                d = dev_pre[name]
                near = t.detach().abs() < RELU_EPS * t.detach().abs().max().item()
                ref_m, dev_m = t.detach() > 0, d > 0
                assert not ((ref_m != dev_m) & ~near).any(), "ReLU mask of %s differs outside the tolerance band" % name
                stats["near"] += int(near.sum())
                stats["flips"] += int(((ref_m != dev_m) & near).sum())
                return t * torch.where(near, dev_m, ref_m).to(t.dtype)
            try:
                _, _, g64 = wo.loss_and_grads({kk: vv.double() for kk, vv in params.items()}, dil, x.double(), target, input_grad=True, relu=relu_in)
                g_in64 = g64["(input)"]
                e_in = (xi.grad.cpu().double() - g_in64).abs().max().item() / max(g_in64.abs().max().item(), 1e-30)
                tie_note = "  [input gradient: float64 oracle with the device's sign at %d of %d near-zero ReLU pre-activations]" % (stats["flips"], stats["near"])
            except AssertionError as e:
                print("FAIL case %3d  %s" % (k, e), flush=True)
                return False
        ok = ok and e_in <= 3e-4
    print("%s case %3d  dil=%s R=%d D=%d S=%d B=%d W=%d bias=%d  pre %.1e p %.1e grad %.1e / %.1e  d input %.1e"
          % ("ok  " if ok else "FAIL", k, dil, R, D, S, B, W, bias, e_pre, e_p, worst, worst2, e_in) + tie_note, flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--epi", action="store_true", help="256 skip channels in every case: wn_skip_epilogue_fwd / _bwd on random shapes")
    ap.add_argument("--gain", type=float, default=2.5, help="weight gain (2.5: the fixtures' conditioning; above ~4 the float32 reference itself is no longer reproducible to 1e-3 - use it to look for non-finite results)")
    ap.add_argument("--only", type=int, default=None, help="run just this case of the stream (prints the failing tensors)")
    args = ap.parse_args()
    global EPI, GAIN
    EPI, GAIN = args.epi, args.gain
    rng = np.random.default_rng(args.seed)
    bad = sum(0 if one_case(rng, k, args.only) else 1 for k in range(args.cases))
    print("%d / %d cases failed" % (bad, args.cases))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
