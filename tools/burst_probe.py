#!/usr/bin/env python3
"""Developer tool (GPU): per-step GPU time of the config-2 fused step over a few hundred steps, resident batch vs the bench loop's
prefetching loader, with the steps that take more than 1.1 x the median listed (index: ms) - do long steps come in bursts, and does
the H2D path have a part in them?      python tools/burst_probe.py [steps]"""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from music_amd.model import wavenet
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
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

    state = {"k": 0}

    def step_resident():
        eng.loss_and_grad_codes(piece, target, scrambled=True)
        eng.adam_step()

    def step_prefetch():
        k = state["k"]
        state["k"] = k + 1
        p, t, ready, free = bufs[k & 1]
        prefetch(k + 1)
        main_s.wait_event(ready)
        eng.loss_and_grad_codes(p, t, scrambled=True)
        free.record(main_s)
        eng.adam_step()

    prefetch(0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    gc.collect()
    gc.disable()
    for name, step in (("resident", step_resident), ("prefetch", step_prefetch), ("resident", step_resident), ("prefetch", step_prefetch)):
        for _ in range(80):
            step()
        ev[0].record()
        for i in range(n):
            step()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        med = sorted(ts)[n // 2]
        long_ = ["%d: %.2f" % (i, t) for i, t in enumerate(ts) if t > 1.1 * med]
        print("%-9s mean %.3f median %.3f max %.3f  long steps (> 1.1 x median): %s" % (name, sum(ts) / n, med, max(ts), ", ".join(long_) or "none"), flush=True)


if __name__ == "__main__":
    main()
