#!/usr/bin/env python3
"""Developer tool (GPU): the fused training step (fwd + CE + bwd + Adam, resident batch) across shapes - BASELINE config 1
(10 blocks, 32 channels, 1 x 4000), config 2 at several batch sizes and clip lengths, the reference's shipped
wavenet_params.json (40 blocks, 32 / 32 / 512) at its shipped batch (4 x 44093)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

C1 = dict(filter_width=2, dilations=[2 ** i for i in range(10)], dilation_channels=32, residual_channels=32, skip_channels=32,
          quantization_channels=256, use_bias=False)
SHIPPED = dict(filter_width=2, dilations=[2 ** i for i in range(10)] * 4, dilation_channels=32, residual_channels=32,
               skip_channels=512, quantization_channels=256, use_bias=False)


def run(tag, cfg, B, T, steps=20):
    from music_amd.model import wavenet
    torch.manual_seed(0)
    net = wavenet(**cfg).cuda()
    eng = net._engine_for(torch.device("cuda", 0))
    eng.adam_init(lr=1e-4)
    rng = np.random.default_rng(0)
    rf = net.receptive_field
    W = T - rf + 1
    codes = torch.from_numpy(rng.integers(0, 256, size=(B, T)).astype(np.int32)).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).cuda()

    def step():
        eng.loss_and_grad_codes(codes, target, scrambled=True)
        eng.adam_step()
    # settle: after a light load (or idle) the chip needs 0.15-0.2 s of full load before its clocks are back up - a config-2 run
    # timed right behind the 0.8 ms steps of config 1 read 7.6 instead of 4.2 ms per step
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < 0.4:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    print("%-44s %8.3f ms/step  %7.2f M samples/s  (%.1f GiB)" % (tag, dt * 1e3, B * T / dt / 1e6, mem))
    del net, eng
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    run("config 1: 10 blocks, 32 ch, 1 x 4000", C1, 1, 4000, 50)
    run("config 1 model, 8 x 16000", C1, 8, 16000, 30)
    for B, T in ((1, 16000), (4, 16000), (8, 16000), (16, 16000), (32, 16000), (8, 32000), (8, 64000)):
        run("config 2 model, %d x %d" % (B, T), bench.CFG, B, T)
    run("shipped 40 blocks 32/32/512, 4 x 44093", SHIPPED, 4, 44093, 10)
