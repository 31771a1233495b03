#!/usr/bin/env python3
"""GPU box: the six wide channel-mixing products of config 2's epilogue, each ALONE on the stream (no second stream, whole batch),
HIP-event time per launch; with --ab NAME an experimental form (tools/exp/wn_gemm_<name>.hip, built in by tools/mkvar.sh with
-DWN_EXP_GEMM and selected by WN_GEMM_<NAME>=1) and chan_gemm_wide2_k alternate in ONE process.

    WAVENET_HIP_LIB=tools/_var_expg.so python tools/gemm_bench.py [--reps 20] [--ab w1|dma] [--rounds 3]
"""
import argparse, json, os, statistics, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CFG, B_LOCAL, T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--ab", default="")
    args = ap.parse_args()
    from music_amd.model import wavenet
    from music_amd import _lib
    from music_amd._lib import call, ptr
    from music_amd.engine import SLACK
    torch.manual_seed(0)
    net = wavenet(**CFG).cuda()
    eng = net._engine_for(torch.device("cuda", 0))
    rng = np.random.default_rng(0)
    codes = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL, T)).astype(np.int32)).cuda()
    target = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL * (T - 3070),)).astype(np.int64)).cuda()
    for _ in range(3):
        eng.loss_and_grad_codes(codes, target)
    torch.cuda.synchronize()
    ws = eng.workspace(B_LOCAL, T)
    bw = ws["bwd"]
    st = _lib.stream()
    B, CH, N, SP, Q, pitch, W = B_LOCAL, eng.CH, eng.N, eng.SP, eng.Q, ws["pitch"], ws["W"]
    lo = eng.rf - 1
    zb, sb = N * CH * pitch, SP * pitch
    fr = lambda n: ptr(eng.pk_f, eng.pk_f_off[n])
    br = lambda n: ptr(eng.pk_b, eng.pk_b_off[n])
    mf, mb = eng.mode_fwd, eng.mode_bwd
    U, H, Z, O = ptr(ws["U"], SLACK), ptr(ws["H"], SLACK), ptr(ws["Z"], SLACK), ptr(ws["O"])
    dO, dH, dU, dZ = ptr(bw["dO"]), ptr(bw["dH"], SLACK), ptr(bw["dU"], SLACK), ptr(bw["dZ"], SLACK)
    prods = {
        "f_skip (256 x 1920)": lambda: call("wn_chan_gemm", Z, None, zb, pitch, lo, T, 0, 0, N * CH // 32, 0, fr("skip"), SP // 16, eng.S, U, sb, pitch, 0, None, None, 0, 0, 0, None, 0, 0, lo, T, 0, B, mf, st),
        "f_p1 (256 x 256, relu in)": lambda: call("wn_chan_gemm", U, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, fr("p1"), SP // 16, eng.S, H, sb, pitch, 0, None, None, 0, 0, 0, None, 0, 0, lo, T, 1, B, mf, st),
        "f_p2 (256 x 256, relu in, compact out)": lambda: call("wn_chan_gemm", H, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, fr("p2"), Q // 16, Q, O, Q * W, W, -lo, None, None, 0, 0, 0, None, 0, 0, lo, T, 1, B, mf, st),
        "b_p2T (256 x 256, compact in, mask)": lambda: call("wn_chan_gemm", dO, None, Q * W, W, 0, W, -lo, 0, Q // 32, 0, br("p2T"), SP // 16, eng.S, dH, sb, pitch, 0, None, None, 0, 0, 0, H, sb, pitch, lo, T, 0, B, mb, st),
        "b_p1T (256 x 256, mask)": lambda: call("wn_chan_gemm", dH, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, br("p1T"), SP // 16, eng.S, dU, sb, pitch, 0, None, None, 0, 0, 0, U, sb, pitch, lo, T, 0, B, mb, st),
        "b_skipT (1920 x 256)": lambda: call("wn_chan_gemm", dU, None, sb, pitch, lo, T, 0, 0, SP // 32, 0, br("skipT"), N * CH // 16, N * CH, dZ, zb, pitch, 0, None, None, 0, 0, 0, None, 0, 0, lo, T, 0, B, mb, st),
    }

    if hasattr(_lib.load(), "wn_skip_epilogue_fwd"):
        prods["fused forward epilogue (skip -> p1 -> p2, one launch)"] = lambda: call(
            "wn_skip_epilogue_fwd", Z, zb, pitch, N * CH // 32, fr("skip"), None, U, H, sb, fr("p1c"), None, fr("p2c"), None, O, Q * W, W,
            eng.S, Q, lo, T, B, mf, st)
        prods["fused backward epilogue (dH, dU, dZ, one launch)"] = lambda: call(
            "wn_skip_epilogue_bwd", dO, Q * W, W, H, U, sb, pitch, dH, dU, dZ, zb, br("p2T"), br("p1Tc"), br("skipTc"), N * CH // 16, N * CH,
            eng.S, lo, T, B, mb, st)

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps * 1e3

    key = "WN_GEMM_%s" % args.ab.upper()
    modes = [(args.ab, "1"), ("wide2", "0")] if args.ab else [("wide2", "0")]
    res = {m: {k: [] for k in prods} for m, _ in modes}
    for _ in range(args.rounds):
        for m, v in modes:
            os.environ[key] = v
            for k, fn in prods.items():
                res[m][k].append(timeit(fn))
    out = {m: {k: round(statistics.median(v), 1) for k, v in r.items()} for m, r in res.items()}
    for m in out:
        out[m]["sum_us"] = round(sum(out[m].values()), 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
