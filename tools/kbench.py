#!/usr/bin/env python3
"""Micro-benchmark of single kernels of the hot path at BASELINE config-2 shapes (GPU only).

    python tools/kbench.py [fwd|bwd|all] [--reps N] [--precision f16x3,bf16x3]

Prints per-phase / per-layer kernel times measured with HIP events on the launch stream.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CFG, B_LOCAL, T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--precision", default="f16x3,bf16x3")
    ap.add_argument("--overlap", type=int, default=1)
    args = ap.parse_args()
    from music_amd.model import wavenet
    from music_amd import _lib
    from music_amd._lib import call, ptr
    from music_amd.engine import SLACK
    torch.manual_seed(0)
    net = wavenet(**CFG)
    net.precision = tuple(args.precision.split(","))
    net = net.cuda()
    eng = net._engine_for(torch.device("cuda", 0))
    eng.overlap_wgrad = bool(args.overlap)
    rng = np.random.default_rng(0)
    codes = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL, T)).astype(np.int32)).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL * (T - 3070),)).astype(np.int64)).cuda()
    x = eng.onehot(codes)
    for _ in range(2):
        eng.loss_and_grad(x, target)
    torch.cuda.synchronize()
    ws = eng.workspace(B_LOCAL, T)
    st = _lib.stream()
    CH, N, pitch = eng.CH, eng.N, ws["pitch"]
    xb, zb = CH * pitch, N * CH * pitch
    fr = lambda name: ptr(eng.pk_f, eng.pk_f_off[name])
    res = {}
    if args.what in ("fwd", "all"):
        per_layer = []
        for i, d in enumerate(eng.dil):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(args.reps):
                call("wn_resblock_fwd", eng._x(ws, i), eng._x(ws, i + 1), ptr(ws["Z"], SLACK + i * CH * pitch), xb, zb, pitch,
                     fr("fg%d" % i), fr("d%d" % i), None, None, None, eng.D, eng.R, CH, d, eng.off[i + 1], T, eng.rf - 1,
                     1, None, 0, 0, 0, 0, 0, None, 0, None, 0, B_LOCAL, eng.mode_fwd, st)
            ev[1].record()
            torch.cuda.synchronize()
            per_layer.append(ev[0].elapsed_time(ev[1]) / args.reps * 1e3)
        res["resblock_fwd_us_by_layer"] = [round(v, 1) for v in per_layer]
        res["resblock_fwd_us_total"] = round(sum(per_layer), 1)
    if args.what == "dx":
        # the per-layer data-gradient product in isolation, and cut-down forms of it (timing only)
        eng.loss_and_grad(x, target)
        bw = ws["bwd"]
        i = 12
        d, t_lo = eng.dil[i], eng.off[i + 1]
        dfg = ptr(bw["dfg"][0], SLACK)
        dy = ptr(bw["dX"][1], SLACK)
        out = ptr(bw["dX"][0], SLACK)
        br = lambda name: ptr(eng.pk_b, eng.pk_b_off[name])

        def run(ks0, ks1, resid, label, shift=d, in_hi=pitch):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(args.reps):
                call("wn_chan_gemm", dfg, dfg if ks1 else None, 2 * CH * pitch, pitch, t_lo, in_hi, 0, shift, ks0, ks1, br("fgT%d" % i),
                     CH // 16, eng.R, out, xb, pitch, 0, None, dy if resid else None, xb, pitch, t_lo, None, 0, 0, eng.off[i], T, 0,
                     B_LOCAL, eng.mode_bwd, st)
            ev[1].record()
            torch.cuda.synchronize()
            res[label] = round(ev[0].elapsed_time(ev[1]) / args.reps * 1e3, 1)
        run(4, 4, True, "dx_full_us")
        run(4, 4, False, "dx_noresid_us")
        run(4, 0, True, "dx_one_tap_us")
        run(2, 0, True, "dx_half_tap_us")
        run(1, 0, False, "dx_one_kstep_noresid_us")
        run(4, 4, True, "dx_shift0_us", shift=0)
        res["layer"] = dict(i=i, d=d, t_lo=t_lo)
    if args.what == "skip":
        # the skip product (wide GEMM, K = 1920) in isolation with shortened K (timing only)
        eng.loss_and_grad(x, target)
        lo, SP = eng.rf - 1, eng.SP

        def run(ks, label):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(args.reps):
                call("wn_chan_gemm", ptr(ws["Z"], SLACK), None, zb, pitch, lo, T, 0, 0, ks, 0, fr("skip"), SP // 16, eng.S,
                     ptr(ws["U"], SLACK), SP * pitch, pitch, 0, None, None, 0, 0, 0, None, 0, 0, lo, T, 0, B_LOCAL, eng.mode_fwd, st)
            ev[1].record()
            torch.cuda.synchronize()
            res[label] = round(ev[0].elapsed_time(ev[1]) / args.reps * 1e3, 1)
        for ks in (60, 30, 15, 8, 2):
            run(ks, "skip_ks%d_us" % ks)
    if args.what in ("bwd", "all", "epi"):
        eng.fine_marks = args.what == "epi"
        eng.marks = []
        for _ in range(args.reps):
            eng.loss_and_grad(x, target)
        torch.cuda.synchronize()
        marks, eng.marks = eng.marks, None
        ph = {}
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            ph[n1] = ph.get(n1, 0.0) + e0.elapsed_time(e1)
        res["phase_ms"] = {k: round(v / args.reps, 3) for k, v in ph.items() if k != "begin"}
    if args.what in ("decode", "all"):
        # BASELINE config 5: 30-layer model, class-128 start piece, 1 s of 16 kHz audio, greedy
        import time
        from music_amd import fast_generate as fg
        start = torch.zeros(1, 256, net.receptive_field, device="cuda")
        start[:, 128, :] = 1.0
        fg.generate_codes(net, start, 200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        codes = fg.generate_codes(net, start, 16000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["decode_16000_samples_s"] = round(dt, 4)
        res["decode_samples_per_s"] = round(16000 / dt, 1)
        res["decode_distinct_codes"] = int(torch.unique(codes).numel())
        # batched utterances: U independent streams in one launch
        for U in (16, 64, 128, 512, 1024):
            starts = start.repeat(U, 1, 1).clone()
            for u in range(U):                      # different start classes so the streams differ
                starts[u].zero_()
                starts[u, (128 + u) % 256, :] = 1.0
            fg.generate_codes_batch(net, starts, 50)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cb = fg.generate_codes_batch(net, starts, 4000)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res["decode_batch%d_samples_per_s" % U] = round(U * 4000 / dt, 1)
        # the same model with biases (what the autoencoder's cached decoder looks like to the kernel)
        from music_amd.model import wavenet as _wn
        torch.manual_seed(1)
        netb = _wn(**dict(CFG, use_bias=True)).cuda()
        fg.generate_codes(netb, start, 200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fg.generate_codes(netb, start, 8000)
        torch.cuda.synchronize()
        res["decode_bias_samples_per_s"] = round(8000 / (time.perf_counter() - t0), 1)
    if args.what in ("ae", "all"):
        # BASELINE config 4: autoencoder, 30+30 blocks, 64 ch, skip 256, bottleneck 64, pool 512, batch 8 x 16000:
        # forward + CE + backward (fresh conditioning projections every forward, as in the reference)
        import time
        from music_amd.model1 import wavenet_autoencoder
        torch.manual_seed(0)
        ae = wavenet_autoencoder(filter_width=2, quantization_channel=256, dilations=CFG["dilations"], en_residual_channel=64,
                                 en_dilation_channel=64, en_bottleneck_width=64, en_pool_kernel_size=512,
                                 de_residual_channel=64, de_dilation_channel=64, de_skip_channel=256, use_bias=False).cuda()
        opt = torch.optim.Adam(ae.parameters(), lr=1e-4)
        lossf = torch.nn.CrossEntropyLoss()

        def ae_step():
            opt.zero_grad()
            loss = lossf(ae(x), target)
            loss.backward()
            opt.step()
            return loss
        for _ in range(2):
            ae_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            loss = ae_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        res["autoencoder_step_ms"] = round(dt * 1e3, 2)
        res["autoencoder_samples_per_s"] = round(B_LOCAL * T / dt, 1)
        res["autoencoder_loss"] = round(float(loss.item()), 5)
        # the fused step of the engine (one softmax + CE + backward kernel, flat Adam; no autograd, no per-tensor optimizer)
        aeng = ae._engine_for(x.device)
        aeng.adam_init(lr=1e-4)

        def ae_fused():
            loss = aeng.loss_and_grad(x, target, ae._draw_conditioning())
            aeng.adam_step()
            return loss
        for _ in range(2):
            ae_fused()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            loss = ae_fused()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        res["autoencoder_fused_step_ms"] = round(dt * 1e3, 2)
        res["autoencoder_fused_samples_per_s"] = round(B_LOCAL * T / dt, 1)
        # cached-queue generation from the autoencoder (one pooled frame of encoding, conditioning folded into biases)
        from music_amd import ae_generate as ag
        piece = torch.zeros(1, 256, ae.receptive_field + 512, device="cuda")
        piece[:, 128, :] = 1.0
        try:
            ag.generate_cached(ae, piece, 200)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ag.generate_cached(ae, piece, 8000)
            torch.cuda.synchronize()
            res["autoencoder_cached_generation_samples_per_s"] = round(8000 / (time.perf_counter() - t0), 1)
        except ValueError as e:           # (the piece must pool to exactly one frame)
            res["autoencoder_cached_generation"] = str(e)[:120]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
