#!/usr/bin/env python3
"""Developer aid (GPU box): the reference's own training loop (wavenet/train.py:171-182) on the drop-in module at 8 x 16000 -
what `extra.reference_surface_step` of the bench line times - for a rocprofv3 kernel trace, plus its wall time per step.
    python tools/surface_prof.py [--opt flat|torch] [--steps N]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CFG, B_LOCAL, T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opt", default="flat")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    from music_amd.model import wavenet
    from music_amd import train as wtrain
    torch.manual_seed(0)
    net = wavenet(**CFG).cuda()
    eng = net._engine_for(torch.device("cuda", 0))
    rng = np.random.default_rng(0)
    codes = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL, T)).astype(np.int32)).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL * (T - 3070),)).astype(np.int64)).cuda()
    x = eng.onehot(codes, scrambled=True)
    ce = torch.nn.CrossEntropyLoss()
    opt = wtrain.get_optimizer(net, "adam", 1e-4, 0.9) if args.opt == "flat" else torch.optim.Adam(net.parameters(), lr=1e-4)

    def step():
        opt.zero_grad()
        loss = ce(net(x), target)
        loss.backward()
        opt.step()
        return loss
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print("surface step (%s optimizer): %.3f ms per step, loss %.5f" % (args.opt, dt * 1e3, loss.item()))


if __name__ == "__main__":
    main()
