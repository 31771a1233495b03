#!/usr/bin/env python3
"""Randomised parity fuzz of the autoencoder (GPU): random widths (decoder / encoder at 64 padded channels half of the time
each, so that the one-launch backward blocks run - the decoder's WITH the conditioning on the matrix cores), dilations,
pooling (stretch and tile conditioning, more than 32 pooled frames included), batch (a quarter of the cases: 20-60 short
clips), clip length and bias; loss and every gradient against autograd on oracle/wavenet_oracle.py with the same per-forward
projections.  Test infrastructure (imports oracle/); not part of the product path.

    python tools/fuzz_ae.py [--cases N] [--seed S]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import wavenet_oracle as wo  # noqa: E402
from oracle import intops  # noqa: E402


def one_case(rng, k, general=False):
    from music_amd.model1 import wavenet_autoencoder
    n = int(rng.integers(2, 7))
    dil = [int(rng.choice([1, 2, 3, 4, 8, 16, 5])) for _ in range(n)]
    wide = rng.random() < 0.5
    ewide = rng.random() < 0.5                  # 64 padded encoder channels: the one-launch encoder backward block
    cfg = dict(filter_width=2, quantization_channel=256, dilations=dil,
               en_residual_channel=int(rng.integers(33, 65) if ewide else rng.integers(8, 33)),
               en_dilation_channel=int(rng.integers(33, 65) if ewide else rng.integers(8, 33)),
               en_bottleneck_width=int(rng.integers(2, 17)), en_pool_kernel_size=int(rng.choice([7, 16, 25, 50, 64])),
               de_residual_channel=int(rng.integers(33, 65)) if wide else int(rng.integers(8, 33)),
               de_dilation_channel=int(rng.integers(33, 65)) if wide else int(rng.integers(8, 33)),
               de_skip_channel=int(rng.choice([16, 40, 64, 100])), use_bias=bool(rng.random() < 0.35))
    fw, Q = 2, 256
    if general:                                 # constructor arguments of the general plan (music_amd/ae_generic.py)
        fw = int(rng.choice([1, 2, 3, 4]))
        Q = int(rng.choice([64, 100, 256, 256]))
        big = rng.random() < 0.4
        cfg.update(filter_width=fw, quantization_channel=Q)
        if big or (fw == 2 and Q == 256):       # (filter width 2 at 256 channels needs > 64 channels to leave the fast engine)
            cfg.update(de_residual_channel=int(rng.integers(65, 100)), de_dilation_channel=int(rng.integers(40, 100)),
                       en_dilation_channel=int(rng.integers(20, 90)))
        dil = dil[:4]
        cfg["dilations"] = dil
        n = len(dil)
    torch.manual_seed(500 + k)
    net = wavenet_autoencoder(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
    params = {kk: v.clone() for kk, v in net.state_dict().items()}
    net = net.cuda()
    rf = net.receptive_field
    many = rng.random() < 0.25                  # many short clips: a workgroup's items then span several clips
    B = int(rng.integers(20, 60)) if many else int(rng.integers(1, 4))
    W = int(cfg["en_pool_kernel_size"] * (rng.integers(1, 3) if many else rng.integers(1, 9)) + rng.choice([0, 0, 1, 3, 17]))
    idx = rng.integers(0, Q, size=(B, rf + W - 1))
    x = torch.from_numpy(np.stack([intops.one_hot_proper(r, Q) for r in idx]))
    target = torch.from_numpy(rng.integers(0, Q, size=(B * W,)).astype(np.int64))
    torch.manual_seed(900 + k)
    net.zero_grad()
    probs = net(x.cuda())
    loss = torch.nn.CrossEntropyLoss()(probs, target.cuda())
    loss.backward()
    torch.manual_seed(900 + k)
    cond = wo.draw_conditioning(n, cfg["en_bottleneck_width"], cfg["de_dilation_channel"], cfg["de_skip_channel"])
    leaf = {kk: v.clone().requires_grad_(True) for kk, v in params.items()}
    p_ref, _ = wo.autoencoder_forward(leaf, dil, x, cfg["en_pool_kernel_size"], cond, filter_width=fw, q=Q)
    l_ref = torch.nn.functional.cross_entropy(p_ref, target)
    g_ref = torch.autograd.grad(l_ref, list(leaf.values()), allow_unused=True)
    e_p = (probs.detach().cpu() - p_ref.detach()).abs().max().item()
    gs = [torch.zeros_like(leaf[nm]) if g is None else g for (nm, _), g in zip(net.named_parameters(), g_ref)]
    floor = 1e-3 * max(g.abs().max().item() for g in gs)
    worst, wname = 0.0, ""
    for (name, p), g in zip(net.named_parameters(), gs):
        err = (p.grad.cpu() - g).abs().max().item() / max(g.abs().max().item(), floor)
        if err > worst:
            worst, wname = err, name
    ok = e_p <= 1e-3 and abs(loss.item() - l_ref.item()) < 1e-4 and worst <= 2e-3
    if not ok and e_p <= 1e-3 and general:
        # general plan: every ReLU's pre-activation (encoder x_i / h_i, skip sum, connection_1 output) as the device holds it
        # against the oracle's: opposite signs within rounding of 0 make the gradients differ legitimately
        from music_amd.engine import SLACK
        eng = net._engine
        T = idx.shape[1]
        ws = eng.workspace(B, T)
        pitch = ws["pitch"]
        inter = {}
        with torch.no_grad():
            wo.autoencoder_forward(params, dil, x, cfg["en_pool_kernel_size"], cond, filter_width=fw, q=Q, intermediates=inter)
        ties, near = 0, 0.0

        def cmp(key, i, rows_p, rows, ref):
            buf = ws[key][SLACK + i * B * rows_p * pitch:SLACK + (i + 1) * B * rows_p * pitch].view(B, rows_p, pitch)
            gpu = buf[:, :rows, T - ref.size(2):T].cpu()
            diff = (gpu > 0) != (ref > 0)
            return int(diff.sum()), (float(ref[diff].abs().max() / ref.abs().max()) if diff.any() else 0.0)
        for i in range(n):
            for key, nm, rp, r in (("Xe", "en_x%d" % i, eng.ReP, eng.Re), ("He", "en_h%d" % i, eng.DeP, eng.De)):
                t_, m_ = cmp(key, i, rp, r, inter[nm])
                ties, near = ties + t_, max(near, m_)
        for key, nm in (("U", "de_skip"), ("R1", "de_conn")):
            t_, m_ = cmp(key, 0, eng.SP, eng.Sd, inter[nm])
            ties, near = ties + t_, max(near, m_)
        if ties and near < 1e-5:
            print("tie  case %3d  %d ReLU pre-activation(s) within %.1e of 0 (relative) with opposite signs; grad %.1e (%s) not judged"
                  % (k, ties, near, worst, wname), flush=True)
            return True
    if not ok and e_p <= 1e-3 and not general:
        # ReLU ties in the encoder (its ReLUs sit on x_i and h_i): a pre-activation within rounding of 0 with opposite
        # signs on the two sides makes the gradients differ legitimately
        import torch.nn.functional as F
        from music_amd.engine import SLACK
        eng = net._engine
        ws = eng.workspace(B, idx.shape[1])
        pitch, T = ws["pitch"], idx.shape[1]
        with torch.no_grad():
            xr = F.conv1d(x, params["en_causal_layer.weight"], params.get("en_causal_layer.bias"))
            ties, off = 0, 1
            for i, d in enumerate(dil):
                hr = F.conv1d(F.relu(xr), params["en_dilation_layer_stack.%d.weight" % i], params.get("en_dilation_layer_stack.%d.bias" % i), dilation=d)
                for nm, ref, rows, o in (("Xe", xr, cfg["en_residual_channel"], off), ("He", hr, cfg["en_dilation_channel"], off + d)):
                    buf = ws[nm][SLACK + i * B * eng.CHe * pitch:SLACK + (i + 1) * B * eng.CHe * pitch].view(B, eng.CHe, pitch)
                    gpu = buf[:, :rows, o:T].cpu()
                    ties += int(((gpu > 0) != (ref > 0)).sum())
                xn = F.conv1d(F.relu(hr), params["en_dense_layer_stack.%d.weight" % i], params.get("en_dense_layer_stack.%d.bias" % i))
                xr = xn + xr[:, :, -xn.size(2):]
                off += d
        if ties:
            print("tie  case %3d  %d encoder ReLU pre-activation(s) within rounding of 0 with opposite signs; grad %.1e (%s) not judged"
                  % (k, ties, worst, wname), flush=True)
            return True
    print("%s case %3d fw=%d Q=%d dil=%s en=%d/%d bw=%d pool=%d de=%d/%d S=%d bias=%d B=%d W=%d  p %.1e grad %.1e %s"
          % ("ok  " if ok else "FAIL", k, fw, Q, dil, cfg["en_residual_channel"], cfg["en_dilation_channel"], cfg["en_bottleneck_width"],
             cfg["en_pool_kernel_size"], cfg["de_residual_channel"], cfg["de_dilation_channel"], cfg["de_skip_channel"],
             cfg["use_bias"], B, W, e_p, worst, "" if ok else wname), flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--general", action="store_true", help="draw constructor arguments of the general plan (filter widths 1..4, Q 64 / 100 / 256, > 64 channels)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = sum(0 if one_case(rng, k, args.general) else 1 for k in range(args.cases))
    print("%d / %d cases failed" % (bad, args.cases))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
