#!/usr/bin/env python3
"""Developer tool (GPU): step time of the config-2 step in windows of 25 steps from the first step of the first GPU
process on a fresh box (how long until the figure is steady)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from music_amd.model import wavenet
    t_start = time.perf_counter()
    torch.manual_seed(0)
    net = wavenet(**bench.CFG).cuda()
    eng = net._engine_for(torch.device("cuda", 0))
    eng.adam_init(lr=1e-4)
    B, T = bench.B_LOCAL, bench.T
    codes = bench.synth_codes(0, B, T)
    rf = net.receptive_field
    W = T - rf + 1
    piece = codes[:, :T].contiguous()
    target = codes[:, rf:rf + W].to(torch.int64).contiguous().view(-1)
    torch.cuda.synchronize()
    out = []
    pre = len(sys.argv) > 2 and sys.argv[2] == "prefetch"
    if pre:          # the bench loop's loader: pinned host buffers, copies one step ahead on a copy stream
        piece_h, target_h = piece.cpu().pin_memory(), target.cpu().pin_memory()
        main_s = torch.cuda.current_stream()
        copy_stream = torch.cuda.Stream()
        bufs = [(torch.empty_like(piece), torch.empty_like(target), torch.cuda.Event(), torch.cuda.Event()) for _ in range(2)]
        for b in bufs:
            b[3].record(main_s)

        def prefetch(k):
            p, t, ready, free = bufs[k & 1]
            copy_stream.wait_event(free)
            with torch.cuda.stream(copy_stream):
                p.copy_(piece_h, non_blocking=True)
                t.copy_(target_h, non_blocking=True)
                ready.record(copy_stream)
        prefetch(0)
    k = 0
    for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
        t0 = time.perf_counter()
        for _ in range(25):
            if pre:
                p, t, ready, free = bufs[k & 1]
                prefetch(k + 1)
                main_s.wait_event(ready)
                eng.loss_and_grad_codes(p, t, scrambled=True)
                free.record(main_s)
                k += 1
            else:
                eng.loss_and_grad_codes(piece, target, scrambled=True)
            eng.adam_step()
        torch.cuda.synchronize()
        out.append("%.3f@%.2fs" % ((time.perf_counter() - t0) / 25 * 1e3, time.perf_counter() - t_start))
    print("ms/step per 25-step window @ seconds since start:", " ".join(out))


if __name__ == "__main__":
    main()
