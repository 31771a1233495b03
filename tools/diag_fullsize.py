#!/usr/bin/env python3
"""Developer diagnostic (GPU): per-parameter gradient error of the HIP path and of the fp32 CPU oracle, both against the
oracle evaluated in float64, at the full-size configurations of tests/test_gpu_fullsize.py - each against float64 with
ITS OWN signs at the near-zero ReLU pre-activations (tests/test_gpu_fullsize.py explains why), and, for the HIP path,
also against plain float64 (the round-2 figure, dominated by a few dozen sign flips)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import wavenet_oracle as wo  # noqa: E402
from tests.helpers import scrambled_input  # noqa: E402
from tests.test_gpu_fullsize import C2, SHIPPED, _scaled, _oracle_grads_f32_f64, _c2_dev_pre  # noqa: E402


def run(cfg, gain, B, T, seed, tag, aligned=False):
    from music_amd.model import wavenet
    torch.manual_seed(seed)
    net = wavenet(**cfg)
    params = _scaled(net, gain)
    net = net.cuda()
    rng = np.random.default_rng(seed * 10 + 1)
    rf = net.receptive_field
    W = T - rf + 1
    codes = rng.integers(0, 256, size=(B, T))
    x = scrambled_input(codes)
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))
    if aligned:          # targets = the clip's own next samples for a STRUCTURED clip (period-7 pattern): the terms of every gradient sum line up
        codes = (np.arange(T)[None, :] * 37 % 7 * 31 + 11 + np.arange(B)[:, None]) % 256
        x = scrambled_input(codes)
        target = torch.from_numpy(np.full((B * W,), 7, dtype=np.int64))
    eng = net._engine_for(torch.device("cuda", 0))
    loss = eng.loss_and_grad(x.cuda(), target.cuda(), want_probs=True)
    l32, p32, e32, l64, g64 = _oracle_grads_f32_f64(params, cfg["dilations"], x, target, _c2_dev_pre(eng, eng.workspace(B, T)))
    _, _, g64p = wo.loss_and_grads({k: v.double() for k, v in params.items()}, cfg["dilations"], x.double(), target)
    print("==", tag, "loss gpu %.7f cpu32 %.7f f64 %.7f" % (loss.item(), l32.item(), l64.item()))
    rows = []
    for name in eng.param_names:
        ref = g64[name]
        s = max(ref.abs().max().item(), 1e-30)
        got = eng.param_view(name, grad=True).cpu().double()
        rows.append((name, (got - ref).abs().max().item() / s, e32[name], (got - g64p[name]).abs().max().item() / s, s))
    rows.sort(key=lambda r: -r[1])
    for name, eg, ec, egp, s in rows[:12]:
        print("  %-34s gpu-vs-f64 %.2e   cpu32-vs-f64 %.2e   (gpu vs PLAIN f64 %.2e)   |g|max %.2e" % (name, eg, ec, egp, s))
    print("  worst gpu %.2e, worst cpu32 %.2e, worst gpu vs plain f64 %.2e" % (max(r[1] for r in rows), max(r[2] for r in rows), max(r[3] for r in rows)))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "aligned":
        run(C2, 2.5, 2, 16000, 3, "c2 2x16000 gain 2.5 ALIGNED", aligned=True)
        run(SHIPPED, 3.0, 2, 4094 + 3999, 7, "shipped 2x8093 gain 3 ALIGNED", aligned=True)
    if what in ("c2", "all"):
        run(C2, 2.5, 2, 16000, 3, "c2 2x16000 gain 2.5")
        run(C2, 2.5, 2, 6000, 3, "c2 2x6000 gain 2.5")
        run(C2, 1.5, 2, 16000, 3, "c2 2x16000 gain 1.5")
    if what in ("shipped", "all"):
        run(SHIPPED, 3.0, 2, 4094 + 3999, 7, "shipped 2x8093 gain 3")
        run(SHIPPED, 2.0, 2, 4094 + 3999, 7, "shipped 2x8093 gain 2")
