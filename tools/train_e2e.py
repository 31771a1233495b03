#!/usr/bin/env python3
"""Developer tool (GPU): END-TO-END speed of the drop-in training harness (music_amd/train.py: its data loader, H2D, the model,
the optimizer, logging) at the config-2 shape - 8 clips x 16000 samples per step - on a synthetic np_audio.pkl, next to
bench.py's engine-only step.  `--fused` sets "fused_step" in train_params.json (the engine's fused step inside the harness);
`--workers N`: DataLoader workers.

    python tools/train_e2e.py [--fused] [--workers N] [--steps 60]"""
import argparse
import json
import os
import pickle
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--steps", type=int, default=60)
    args = ap.parse_args()
    from music_amd import train as T
    rf = sum(bench.CFG["dilations"]) + 2
    W = bench.T - rf + 1
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "params"))
    rng = np.random.default_rng(0)
    n_pieces = 8 * args.steps
    # one long recording chopped into pieces of rf + W - 1 samples by the loader (faster_audio_data.py): give it exactly enough
    data = [rng.integers(0, 256, size=(rf + W * n_pieces,)).astype(np.int32)]
    pickle.dump(data, open(os.path.join(tmp, "np_audio.pkl"), "wb"))
    dp = dict(batch_size=8, shuffle=False, num_workers=args.workers, pin_memory=True, audio_path=os.path.join(tmp, "np_audio.pkl"),
              receptive_field=rf, window_length=W, cuda_available=True, quantization_channels=256)
    tp = dict(log_dir="./log/", restore_dir="./restore/", restore_model="", check_point_every=1000, print_every=20, num_epochs=1,
              wavenet_params="./wavenet_params.json", optimizer="adam", max_check_points=2, learning_rate=1e-4, momentum=0.9,
              device_ids=None, fused_step=bool(args.fused))
    for n, p in (("wavenet", bench.CFG), ("dataset", dp), ("train", tp)):
        json.dump(p, open(os.path.join(tmp, "params", n + "_params.json"), "w"))
    os.chdir(tmp)
    torch.manual_seed(0)
    t0 = time.perf_counter()
    T.train()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lines = open(os.path.join(tmp, "log", "loss_log.log")).read().strip().split("\n")
    print("train(): %d steps of 8 x %d in %.2f s = %.2f ms per step incl. start-up (%s, %d workers); last log line: %s"
          % (args.steps, bench.T, dt, dt / args.steps * 1e3, "fused_step" if args.fused else "autograd + torch Adam", args.workers, lines[-1]))


if __name__ == "__main__":
    main()
