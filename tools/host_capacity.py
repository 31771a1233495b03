#!/usr/bin/env python3
"""GPU box: what the HOST side of N ranks costs when they enqueue the config-2 step at the same time.

    python tools/host_capacity.py [--ranks 1,2,4,8] [--steps 12] [--rounds 3]

A 1-GPU box cannot measure a scaling curve, but it can measure the one thing about an 8-rank run that does not need eight
GPUs: eight Python processes, each with the per-rank thread budget of an 8-rank launch (music_amd/_lib.py: thread_budget),
each issuing the ~110 ctypes launches of a step, on the box's CPU quota.  Every process builds the bench's model on the ONE
visible GPU (the kernels of different processes time-slice; their speed is not what is measured), waits at a barrier, then
times `steps` enqueues of the fused step - wall clock and CPU time of the enqueuing thread - with the device queue empty at
the start and nothing waited for inside (the device runs N times slower than the hosts enqueue; `steps` stays below the depth
at which HIP throttles the host).  Prints one JSON object: per rank count, every rank's enqueue ms per step (median of the
rounds), the slowest rank, and the process CPU time per step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(rank, n, steps, rounds, bar, q):
    from music_amd import _lib as wl
    budget = wl.thread_budget(wl.cpu_quota(), n)
    os.environ["OMP_NUM_THREADS"] = str(min(8, budget))
    os.environ["LOCAL_WORLD_SIZE"] = str(n)
    import numpy as np
    import torch
    from bench import CFG, B_LOCAL, T
    from music_amd.model import wavenet
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    net = wavenet(**CFG).cuda()
    eng = net._engine_for(torch.device("cuda", 0))
    eng.adam_init(lr=1e-4)
    rng = np.random.default_rng(rank)
    codes = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL, T)).astype(np.int32)).cuda()
    W = T - net.receptive_field + 1
    target = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL * W,)).astype(np.int64)).cuda()

    def step():
        eng.loss_and_grad_codes(codes, target, scrambled=True)
        eng.adam_step(gscale=1.0 / n)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.disable()
    wall, cpu, proc = [], [], []
    for _ in range(rounds):
        torch.cuda.synchronize()
        bar.wait()
        t0, c0, p0 = time.perf_counter(), time.thread_time(), time.process_time()
        for _ in range(steps):
            step()
        t1, c1, p1 = time.perf_counter(), time.thread_time(), time.process_time()
        torch.cuda.synchronize()
        wall.append((t1 - t0) / steps * 1e3)
        cpu.append((c1 - c0) / steps * 1e3)
        proc.append((p1 - p0) / steps * 1e3)
        bar.wait()
    wall.sort(); cpu.sort(); proc.sort()
    q.put((rank, wall[len(wall) // 2], cpu[len(cpu) // 2], proc[len(proc) // 2], torch.get_num_threads(), wall[0], wall[-1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    from music_amd import _lib as wl
    out = {"cpu_quota": wl.cpu_quota(), "host_cpus": os.cpu_count(), "steps": a.steps, "rounds": a.rounds, "ranks": {}}
    for n in [int(v) for v in a.ranks.split(",")]:
        bar, q = ctx.Barrier(n), ctx.Queue()
        ps = [ctx.Process(target=child, args=(r, n, a.steps, a.rounds, bar, q)) for r in range(n)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=900) for _ in range(n))
        for p in ps:
            p.join(timeout=120)
        out["ranks"][str(n)] = {
            "thread_budget_per_rank": wl.thread_budget(wl.cpu_quota(), n), "torch_threads": res[0][4],
            "enqueue_ms_per_step": [round(r[1], 3) for r in res], "slowest": round(max(r[1] for r in res), 3),
            "slowest_of_any_round": round(max(r[6] for r in res), 3), "fastest_of_any_round": round(min(r[5] for r in res), 3),
            "enqueue_thread_cpu_ms_per_step": [round(r[2], 3) for r in res],
            "process_cpu_ms_per_step": [round(r[3], 3) for r in res],
        }
        print("ranks %d: slowest %.3f ms per step (all: %s)" % (n, out["ranks"][str(n)]["slowest"], out["ranks"][str(n)]["enqueue_ms_per_step"]),
              file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
