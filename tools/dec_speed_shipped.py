#!/usr/bin/env python3
"""Developer tool (GPU): cached-queue decode speed of the reference's SHIPPED WaveNet parameters (40 blocks, 32 / 32 / 512)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from music_amd import fast_generate as fg
    from music_amd.model import wavenet
    torch.manual_seed(0)
    net = wavenet(filter_width=2, dilations=[2 ** i for i in range(10)] * 4, dilation_channels=32, residual_channels=32,
                  skip_channels=512, quantization_channels=256, use_bias=False).cuda()
    dev = torch.device("cuda", 0)
    start = torch.zeros(1, 256, net.receptive_field, device=dev)
    start[0, 128, :] = 1.0
    n = 8000
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seq = fg.generate_codes(net, start, n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("one stream: %d samples in %.3f s = %.2f k samples/s (%d distinct codes)" % (n, dt, n / dt / 1e3, int(torch.unique(seq).numel())))
    U = 128
    st = torch.zeros(U, 256, net.receptive_field, device=dev)
    for uu in range(U):
        st[uu, (128 + uu) % 256, :] = 1.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fg.generate_codes_batch(net, st, 1001)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d utterances x 1000 samples: %.3f s = %.3f M samples/s" % (U, dt, U * 1000 / dt / 1e6))


if __name__ == "__main__":
    main()
