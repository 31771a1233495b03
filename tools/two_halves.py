#!/usr/bin/env python3
"""Developer experiment (GPU): would two half-batches in flight on two streams - one in its dilated-conv stack while the
other is in its epilogue GEMMs - beat one batch of 8?  Two engines of 4 clips each stepped concurrently on two streams
against one engine of 8 clips; prints samples/s for both."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from music_amd.model import wavenet
    dev = torch.device("cuda", 0)
    T = bench.T
    codes = bench.synth_codes(0, 8, T)
    nets = [wavenet(**bench.CFG).cuda() for _ in range(3)]
    engs = [n._engine_for(dev) for n in nets]
    for e in engs:
        e.adam_init(lr=1e-4)
    rf = nets[0].receptive_field
    W = T - rf + 1
    piece = codes[:, :T].contiguous()
    target = codes[:, rf:rf + W].to(torch.int64).contiguous()

    def step(e, p, t):
        e.loss_and_grad_codes(p, t.view(-1), scrambled=True)
        e.adam_step()
    n = 40
    for _ in range(5):
        step(engs[0], piece, target)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(engs[0], piece, target)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / n
    pa, ta, pb, tb = piece[:4].contiguous(), target[:4].contiguous(), piece[4:].contiguous(), target[4:].contiguous()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for off in (0, 1):                     # off = 1: stream B starts half a step late (A in its epilogue while B is in its stack)
        for _ in range(3):
            with torch.cuda.stream(sa):
                step(engs[1], pa, ta)
            with torch.cuda.stream(sb):
                step(engs[2], pb, tb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if off:
            with torch.cuda.stream(sb):
                engs[2].forward_logits(None, codes=(pb, True)) if False else None
        for k in range(n):
            with torch.cuda.stream(sa):
                step(engs[1], pa, ta)
            with torch.cuda.stream(sb):
                step(engs[2], pb, tb)
        torch.cuda.synchronize()
        two = (time.perf_counter() - t0) / n
        print("one batch of 8: %.3f ms/step = %.2f M samples/s; two half-batches on two streams: %.3f ms per pair = %.2f M samples/s" %
              (one * 1e3, 8 * T / one / 1e6, two * 1e3, 8 * T / two / 1e6))
    # the half batch alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(engs[1], pa, ta)
    torch.cuda.synchronize()
    half = (time.perf_counter() - t0) / n
    print("one half-batch alone: %.3f ms/step = %.2f M samples/s" % (half * 1e3, 4 * T / half / 1e6))


if __name__ == "__main__":
    main()
