#!/usr/bin/env python3
"""GPU box: the fused training step must give the SAME BITS every time it runs on the same inputs (no float atomics, every
reduction in a fixed order - DESIGN sections 3 / 4).  Runs forward + CE + backward of config 2 (WaveNet, 8 x 16000) and of
config 4 (autoencoder, same geometry) N times WITHOUT an optimizer step and compares the loss and the whole flat gradient of
every run with the first run's, bit for bit.  A data hazard or a race in any kernel of the step shows up as a mismatch here
long before it moves a tolerance test.

    python tools/soak_determinism.py [--reps 400] [--what c2,c4]

Prints one JSON object; exit code 1 on any mismatch."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def soak(step, grad, reps):
    first_loss = first = None
    bad = []
    for k in range(reps):
        loss = step()
        g = grad()
        if first is None:
            first, first_loss = g.clone(), loss.clone()
            continue
        same = torch.equal(g, first) and torch.equal(loss, first_loss)
        if not same:
            bad.append((k, int((g != first).sum().item()), float((g - first).abs().max().item())))
    torch.cuda.synchronize()
    return dict(runs=reps, mismatching_runs=len(bad), first_mismatches=bad[:5], loss=float(first_loss.item()),
                grad_l1=float(first.double().abs().sum().item()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=400)
    ap.add_argument("--what", default="c2,c4")
    a = ap.parse_args()
    from bench import CFG, B_LOCAL, T
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    out = {}
    if "c2" in a.what:
        from music_amd.model import wavenet
        torch.manual_seed(0)
        net = wavenet(**CFG)
        with torch.no_grad():
            for p_ in net.parameters():
                p_.mul_(2.5)                               # (default init leaves every probability at 1 / 256: the fuzzers' gain)
        net = net.cuda()
        eng = net._engine_for(dev)
        codes = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL, T)).astype(np.int32)).cuda()
        W = T - net.receptive_field + 1
        target = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL * W,)).astype(np.int64)).cuda()
        out["c2"] = soak(lambda: eng.loss_and_grad_codes(codes, target, scrambled=True), lambda: eng.flat_grad, a.reps)
        del net, eng
        torch.cuda.empty_cache()
    if "c4" in a.what:
        from music_amd.model1 import wavenet_autoencoder
        torch.manual_seed(0)
        ae = wavenet_autoencoder(filter_width=2, quantization_channel=256, dilations=CFG["dilations"], en_residual_channel=64,
                                 en_dilation_channel=64, en_bottleneck_width=64, en_pool_kernel_size=512,
                                 de_residual_channel=64, de_dilation_channel=64, de_skip_channel=256, use_bias=False)
        with torch.no_grad():
            for p_ in ae.parameters():
                p_.mul_(2.5)
        ae = ae.cuda()
        aeng = ae._engine_for(dev)
        idx = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL, T)).astype(np.int64)).cuda()
        x = torch.nn.functional.one_hot(idx, 256).permute(0, 2, 1).contiguous().float()
        W = T - ae.receptive_field + 1
        target = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL * W,)).astype(np.int64)).cuda()
        torch.manual_seed(3)
        cond = ae._draw_conditioning()               # ONE draw of the conditioning projections for every run
        out["c4"] = soak(lambda: aeng.loss_and_grad(x, target, cond), lambda: aeng.flat_grad, a.reps)
    if "alt" in a.what:
        # the other forms of the step on a smaller geometry (3 clips x 9000 samples, 12 blocks): engine switches are read when an engine / a
        # workspace is built, so every setting gets a fresh module
        from music_amd.model import wavenet
        alts = [("default", {}, {}), ("pair_form_blocks", {"WN_PQ_CHAIN": "0"}, {}), ("two_role_blocks", {"WN_PQ_BWD": "0"}, {}),
                ("three_launch_epilogue", {"WN_EPI_FUSED": "0", "WN_EPI_FUSED_BWD": "0"}, {}), ("biases", {}, {"use_bias": True}),
                ("32_channels", {}, {"dilation_channels": 32, "residual_channels": 32}),
                ("ragged_channels", {}, {"dilation_channels": 40, "residual_channels": 48, "skip_channels": 100})]
        for name, env, kw in alts:
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                cfg = dict(CFG, dilations=[1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 3, 96], **kw)
                torch.manual_seed(1)
                net = wavenet(**cfg)
                with torch.no_grad():
                    for p_ in net.parameters():
                        p_.mul_(2.5)
                net = net.cuda()
                eng = net._engine_for(dev)
                B2 = 4 if name == "32_channels" else 3
                T2 = 9000
                codes = torch.from_numpy(rng.integers(0, 256, size=(B2, T2)).astype(np.int32)).cuda()
                W = T2 - net.receptive_field + 1
                target = torch.from_numpy(rng.integers(0, 256, size=(B2 * W,)).astype(np.int64)).cuda()
                out["alt:" + name] = soak(lambda: eng.loss_and_grad_codes(codes, target, scrambled=True), lambda: eng.flat_grad, max(a.reps // 4, 50))
                del net, eng
                torch.cuda.empty_cache()
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
    print(json.dumps(out))
    sys.exit(1 if any(v["mismatching_runs"] for v in out.values()) else 0)


if __name__ == "__main__":
    main()
