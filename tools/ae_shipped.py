#!/usr/bin/env python3
"""Developer tool (GPU): phase table of the fused training step of the autoencoder with the reference's SHIPPED parameters
(wavenet_autoencoder/params/model_params.json: 40 + 40 blocks, 32 / 32 channels, bottleneck 512, pool 512, skip 512) at
4 clips of 16384 predicted samples."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHIPPED_AE = dict(filter_width=2, quantization_channel=256, dilations=[2 ** i for i in range(10)] * 4, en_residual_channel=32,
                  en_dilation_channel=32, en_bottleneck_width=512, en_pool_kernel_size=512, de_residual_channel=32,
                  de_dilation_channel=32, de_skip_channel=512, use_bias=False)


def main():
    from music_amd.model1 import wavenet_autoencoder
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    ae = wavenet_autoencoder(**SHIPPED_AE).cuda()
    eng = ae._engine_for(dev)
    eng.adam_init(lr=1e-4)
    B, W = 4, 16384
    T = ae.receptive_field + W - 1
    rng = np.random.default_rng(0)
    codes = torch.from_numpy(rng.integers(0, 256, size=(B, T)).astype(np.int32)).to(dev)
    from music_amd.model import wavenet
    x = torch.zeros(B, 256, T, device=dev)
    x.scatter_(1, codes.long().unsqueeze(1), 1.0)
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).to(dev)

    def step():
        loss = eng.loss_and_grad(x, target, ae._draw_conditioning())
        eng.adam_step()
        return loss
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    eng.marks = []
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ph = {}
    m = eng.marks
    for (n0, e0), (n1, e1) in zip(m[:-1], m[1:]):
        if n1 != "begin":
            ph[n1] = ph.get(n1, 0.0) + e0.elapsed_time(e1) / 3
    print("%.3f ms/step (%.2f M samples/s), loss %.4f; " % (dt * 1e3, B * T / dt / 1e6, loss.item()) +
          ", ".join("%s %.3f" % kv for kv in ph.items()))


if __name__ == "__main__":
    main()
