#!/usr/bin/env python3
"""Developer tool (GPU): cached-queue decode speed of the config-5 model (30 blocks, 64 / 64 / 256 / 256), one stream and
batches, as bench.py's extra.c5_decode measures it; honours the WN_DEC_* switches.  `--bias`: a biased model (the
autoencoder's cached decoder is one)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from music_amd import fast_generate as fg
    from music_amd.model import wavenet
    torch.manual_seed(0)
    cfg = dict(bench.CFG, use_bias="--bias" in sys.argv)
    net = wavenet(**cfg).cuda()
    dev = torch.device("cuda", 0)
    start = torch.zeros(1, 256, net.receptive_field, device=dev)
    start[0, 128, :] = 1.0
    n = 16000
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seq = fg.generate_codes(net, start, n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("one stream: %d samples in %.3f s = %.1f k samples/s (%d distinct codes)" % (n, dt, n / dt / 1e3, int(torch.unique(seq).numel())))
    for U in (128, 1024):
        st = torch.zeros(U, 256, net.receptive_field, device=dev)
        for uu in range(U):
            st[uu, (128 + uu) % 256, :] = 1.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fg.generate_codes_batch(net, st, 2001)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%d utterances x 2000 samples: %.3f s = %.2f M samples/s" % (U, dt, U * 2000 / dt / 1e6))


if __name__ == "__main__":
    main()
