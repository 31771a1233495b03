#!/usr/bin/env python3
"""Developer tool (GPU): phase table of the config-4 (autoencoder) fused training step, as bench.py's extra.c4_autoencoder
prints it (HIP events between the phases, mean of 5 steps)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from music_amd.model1 import wavenet_autoencoder
    from music_amd.model import wavenet
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    ae = wavenet_autoencoder(filter_width=2, quantization_channel=256, dilations=bench.CFG["dilations"], en_residual_channel=64,
                             en_dilation_channel=64, en_bottleneck_width=64, en_pool_kernel_size=512, de_residual_channel=64,
                             de_dilation_channel=64, de_skip_channel=256, use_bias=False).cuda()
    aeng = ae._engine_for(dev)
    aeng.adam_init(lr=1e-4)
    net = wavenet(**bench.CFG).cuda()
    eng = net._engine_for(dev)
    codes = bench.synth_codes(0, bench.B_LOCAL, bench.T)
    rf = net.receptive_field
    W = bench.T - rf + 1
    x = eng.onehot(codes[:, :bench.T].contiguous(), scrambled=True)
    target = codes[:, rf:rf + W].to(torch.int64).contiguous().view(-1)

    def step():
        loss = aeng.loss_and_grad(x, target, ae._draw_conditioning())
        aeng.adam_step()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    aeng.marks = []
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    m, aeng.marks = aeng.marks, None
    ph = {}
    for (n0, e0), (n1, e1) in zip(m[:-1], m[1:]):
        if n1 != "begin":
            ph[n1] = ph.get(n1, 0.0) + e0.elapsed_time(e1) / 5
    print("%.3f ms/step; " % (dt * 1e3) + ", ".join("%s %.3f" % kv for kv in ph.items()))


if __name__ == "__main__":
    main()
