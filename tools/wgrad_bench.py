#!/usr/bin/env python3
"""GPU box: the epilogue's three weight-gradient launches (`wgrad_big_k`) ALONE at config 2's geometry, random operands.

    python tools/wgrad_bench.py [--reps 20] [--chunk 1856]

post_process_2 (dO x relu(H)^T, 256 x 256), post_process_1 (dH x relu(U)^T, 256 x 256), skip convs (dU x Z^T, 256 x 1920), K = 8 x 12930 samples.
WAVENET_HIP_LIB selects a variant library (timing builds give wrong numbers on purpose).  Prints one JSON object (us per launch, median)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from music_amd import _lib  # noqa: E402
from music_amd._lib import call, ptr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--chunk", type=int, default=1856)
    a = ap.parse_args()
    B, T, rf, SP, Q, NCH = 8, 16000, 3070, 256, 256, 1920
    lo, W, pitch, SLACK = rf - 1, T - rf + 1, 16128, 256
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda rows, p: torch.randn(SLACK + B * rows * p + 1024, device=dev, generator=g)
    dO, H, U, dH, dU, Z = mk(Q, W), mk(SP, pitch), mk(SP, pitch), mk(SP, pitch), mk(SP, pitch), mk(NCH, pitch)
    mb = _lib.BF16X3
    ns = _lib.wgrad_slabs(lo, T, a.chunk, B)
    slab = torch.empty(ns * 256 * NCH, device=dev)
    st = _lib.stream()
    sb, zb = SP * pitch, NCH * pitch
    ops = {
        "p2": (ptr(dO), Q * W, W, -lo, W, ptr(H, SLACK), None, sb, pitch, 0, 0, pitch, SP // 16, Q // 16, 1, ptr(slab), SP, 256 * 256),
        "p1": (ptr(dH, SLACK), sb, pitch, 0, pitch, ptr(U, SLACK), None, sb, pitch, 0, 0, pitch, SP // 16, SP // 16, 1, ptr(slab), SP, 256 * 256),
        "skip": (ptr(dU, SLACK), sb, pitch, 0, pitch, ptr(Z, SLACK), None, zb, pitch, 0, 0, pitch, NCH // 16, SP // 16, 0, ptr(slab), NCH, 256 * NCH),
    }
    out = {"chunk": a.chunk, "slabs": ns, "lib": os.environ.get("WAVENET_HIP_LIB", "shipped")}
    for name, args in ops.items():
        run = lambda: call("wn_wgrad", *args, lo, T, a.chunk, B, mb, st)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        out[name] = round(ts[len(ts) // 2], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
