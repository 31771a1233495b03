#!/usr/bin/env python3
"""Randomised GPU parity fuzz of the GENERAL plan (music_amd/engine_generic.py): random filter width 1..5, quantisation
width 8..600, channel counts 4..200 / skip 8..300, depth 1..7 with random dilations, bias on / off, batch 1..3, 1..600 output
columns; pre-softmax, probabilities, loss and every gradient (the input's included) against the oracle (ReLU signs near zero taken from the device,
tests/test_gpu_fullsize.py).  Test infrastructure (imports oracle/); not part of the product path.

    python tools/fuzz_generic.py [--cases N] [--seed S]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import wavenet_oracle as wo  # noqa: E402
from tests.test_gpu_fullsize import _device_relu  # noqa: E402


def one_case(rng, k):
    from music_amd.engine import SLACK
    from music_amd.engine_generic import GenericWaveNetEngine
    from music_amd.model import wavenet
    fw = int(rng.integers(1, 6))
    n = int(rng.integers(1, 8))
    dil = [int(rng.choice([1, 2, 3, 4, 5, 8, 16, 31])) for _ in range(n)]
    R, D, S, Q = int(rng.integers(4, 201)), int(rng.integers(4, 201)), int(rng.integers(8, 301)), int(rng.integers(8, 601))
    if fw == 2 and Q == 256 and max(R, D) <= 64:
        Q = 255
    bias = bool(rng.random() < 0.5)
    B, win = int(rng.integers(1, 4)), int(rng.choice([1, 2, 63, 64, 65, 257, int(rng.integers(1, 601))]))
    cfg = dict(filter_width=fw, dilations=dil, dilation_channels=D, residual_channels=R, skip_channels=S,
               quantization_channels=Q, use_bias=bias)
    torch.manual_seed(7000 + k)
    net = wavenet(**cfg)
    gain = float(rng.uniform(1.5, 3.0))
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(gain)
    params = {kk: v.clone() for kk, v in net.state_dict().items()}
    net = net.cuda()
    rf = net.receptive_field
    T = rf + win - 1
    x = torch.from_numpy(rng.standard_normal((B, Q, T)).astype(np.float32) * 0.5)
    target = torch.from_numpy(rng.integers(0, Q, size=(B * win,)).astype(np.int64))
    xi = x.cuda().requires_grad_(True)          # the input's own gradient is checked with the parameters'
    probs = net(xi)
    eng = net._engine
    assert isinstance(eng, GenericWaveNetEngine)
    loss = torch.nn.functional.cross_entropy(probs, target.cuda())
    loss.backward()
    ws = eng.workspace(B, T)
    pitch, lo = ws["pitch"], rf - 1
    v = lambda buf: buf[SLACK:SLACK + B * eng.SP * pitch].view(B, eng.SP, pitch)[:, :S, lo:T].cpu()
    relu, _ = _device_relu({"skip_sum": v(ws["U"]), "post_process_1": v(ws["H"])})
    inter = {}
    l_ref, p_ref, g_ref = wo.loss_and_grads(params, dil, x, target, filter_width=fw, quantization_channels=Q, intermediates=inter,
                                            relu=relu, input_grad=True)
    e_p = (probs.detach().cpu() - p_ref).abs().max().item()
    e_o = (ws["O"][:B * Q * win].view(B, Q, win).cpu() - inter["pre_softmax"].detach()).abs().max().item()
    gmax = max(g.abs().max().item() for g in g_ref.values())
    worst = 0.0
    gmax = max(g.abs().max().item() for nme, g in g_ref.items() if nme != "(input)")
    for nme, p in net.named_parameters():
        want = g_ref[nme]
        got = torch.zeros_like(want) if p.grad is None else p.grad.cpu()
        worst = max(worst, (got - want).abs().max().item() / max(want.abs().max().item(), 1e-3 * gmax))
    g_in = g_ref["(input)"]
    e_in = (xi.grad.cpu() - g_in).abs().max().item() / max(g_in.abs().max().item(), 1e-30)
    ok = e_p <= 1e-3 and e_o <= 1e-3 and abs(loss.item() - l_ref.item()) < 1e-4 and worst <= 3e-4 and e_in <= 3e-4
    print("%s case %3d  fw=%d dil=%s R=%d D=%d S=%d Q=%d bias=%d B=%d W=%d  pre %.1e p %.1e grad %.1e d input %.1e" % (
        "ok  " if ok else "FAIL", k, fw, dil, R, D, S, Q, bias, B, win, e_o, e_p, worst, e_in))
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    torch.set_num_threads(8)
    bad = sum(0 if one_case(rng, k) else 1 for k in range(a.cases))
    print("%d / %d cases failed" % (bad, a.cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
