#!/usr/bin/env python3
"""Developer tool (GPU): the config-2 training step eager vs captured in ONE hipGraph (torch.cuda.CUDAGraph over the
ctypes launches, both side streams included).  Prints ms per step for both and checks that the replayed step produces
the eager step's gradient bits."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from music_amd.model import wavenet
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    torch.manual_seed(0)
    net = wavenet(**bench.CFG).cuda()
    dev = torch.device("cuda", 0)
    eng = net._engine_for(dev)
    eng.adam_init(lr=1e-4)
    B, T = bench.B_LOCAL, bench.T
    codes = bench.synth_codes(0, B, T)
    rf = net.receptive_field
    W = T - rf + 1
    piece = codes[:, :T].contiguous()
    target = codes[:, rf:rf + W].to(torch.int64).contiguous().view(-1)

    def step():
        loss = eng.loss_and_grad_codes(piece, target, scrambled=True)
        eng.adam_step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps * 1e3
    # the bench loop's extras, one at a time: H2D prefetch on a copy stream (+ ready / free events)
    piece_h, target_h = piece.cpu().pin_memory(), target.cpu().pin_memory()
    main = torch.cuda.current_stream()
    copy_stream = torch.cuda.Stream(device=dev)
    bufs = [(torch.empty_like(piece), torch.empty_like(target), torch.cuda.Event(), torch.cuda.Event()) for _ in range(2)]

    def prefetch(k):
        p, t, ready, free = bufs[k & 1]
        copy_stream.wait_event(free)
        with torch.cuda.stream(copy_stream):
            p.copy_(piece_h, non_blocking=True)
            t.copy_(target_h, non_blocking=True)
            ready.record(copy_stream)

    def run_prefetch(n, marks, do_copy=True, do_sync=True, same_stream=False):
        for b in bufs:
            b[3].record(main)
        prefetch(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            p, t, ready, free = bufs[k & 1]
            if do_copy and do_sync and not same_stream:
                prefetch(k + 1)
            elif do_copy and same_stream:
                p2, t2, _, _ = bufs[(k + 1) & 1]
                p2.copy_(piece_h, non_blocking=True)
                t2.copy_(target_h, non_blocking=True)
            elif do_copy:
                p2, t2, _, _ = bufs[(k + 1) & 1]
                with torch.cuda.stream(copy_stream):
                    p2.copy_(piece_h, non_blocking=True)
                    t2.copy_(target_h, non_blocking=True)
            elif do_sync:
                copy_stream.wait_event(free)
                ready.record(copy_stream)
            if marks and k % 4 == 0:
                eng.marks, eng.mark_only = [], {"step_begin", "causal_fwd", "stack_fwd", "epilogue_bwd", "stack_bwd"}
            else:
                eng.marks = None
            if do_sync and not same_stream:
                main.wait_event(ready)
            eng.loss_and_grad_codes(p, t, scrambled=True)
            if do_sync and not same_stream:
                free.record(main)
            eng.adam_step()
        torch.cuda.synchronize()
        eng.marks, eng.mark_only = None, None
        return (time.perf_counter() - t0) / n * 1e3
    run_prefetch(5, False)
    res = {}
    for rep in range(2):
        for name, kw in (("bare", None), ("copy+sync", dict()), ("copy only (copy stream, no events)", dict(do_sync=False)),
                         ("events only (no copies)", dict(do_copy=False)), ("copies on the main stream", dict(same_stream=True)),
                         ("copy+sync+marks", dict(marks=True))):
            if kw is None:
                t0 = time.perf_counter()
                for _ in range(steps):
                    step()
                torch.cuda.synchronize()
                v = (time.perf_counter() - t0) / steps * 1e3
            else:
                kw = dict(kw)
                v = run_prefetch(steps, kw.pop("marks", False), **kw)
            res.setdefault(name, []).append(v)
    for k, v in res.items():
        print("  %-40s %s ms/step" % (k, " ".join("%.3f" % x for x in v)))
    # gradient of one more eager step from a known parameter state
    flat0 = eng.flat.clone()
    m0, v0, t_0 = eng.adam_state["m"].clone(), eng.adam_state["v"].clone(), eng.adam_state["t"]
    eng.loss_and_grad_codes(piece, target, scrambled=True)
    g_eager = eng.flat_grad.clone()
    torch.cuda.synchronize()

    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()                                   # warm-up on the capture stream (side streams, attributes)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            loss = step()
    torch.cuda.synchronize()
    eng.flat.copy_(flat0)
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(eng.flat_grad, g_eager)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / steps * 1e3
    print("eager %.3f ms/step, hipGraph replay %.3f ms/step (%d steps each); replayed gradient bits == eager: %s; loss %.6f" %
          (eager, graphed, steps, same, loss.item()))


if __name__ == "__main__":
    main()
