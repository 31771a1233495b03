#!/usr/bin/env python3
"""GPU box, developer build -DWN_CLK (tools/mkvar.sh clk "-DWN_CLK" wn_gemm.hip wn_resblock2.hip wn_respq.hip): the shader clock the chip
HOLDS inside the hot loops of config 2's step = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over the last launch's workgroups
(MI355X_MICROARCH.md, DVFS give-back item 6), after a few hundred back-to-back steps on the bench's synthetic batch.

    WAVENET_HIP_LIB=tools/_var_clk.so python tools/clk_probe.py [--steps 300]
"""
import argparse, ctypes, json, os, statistics, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CFG, B_LOCAL, T, synth_codes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    from music_amd.model import wavenet
    from music_amd import _lib
    torch.manual_seed(0)
    net = wavenet(**CFG).cuda()
    eng = net._engine_for(torch.device("cuda", 0))
    eng.adam_init(lr=1e-4)
    codes = synth_codes(0, B_LOCAL, T)
    rf = net.receptive_field
    W = T - rf + 1
    piece = codes[:, :T].contiguous()
    target = codes[:, rf:rf + W].to(torch.int64).contiguous().view(-1)
    for _ in range(a.steps):
        eng.loss_and_grad_codes(piece, target, scrambled=True)
        eng.adam_step()
    torch.cuda.synchronize()
    lib = _lib.load()
    out = {}
    for name, fn in (("chan_gemm_wide2_k k loop (last launch: skip^T data gradient... whichever ran last)", "wn_gw_clk_read"),
                     ("resblock_fwd_nt_k start .. end of the f/g product", "wn_rf_clk_read"),
                     ("resblock_bwd_pq_k R waves' item loop", "wn_pq_clk_read")):
        buf = (ctypes.c_ulonglong * (256 * 4))()
        f = getattr(lib, fn)
        f.argtypes = [ctypes.c_void_p]
        assert f(buf) == 0
        v = np.array(buf, dtype=np.uint64).reshape(256, 4).astype(np.float64)
        ok = (v[:, 3] > v[:, 1]) & (v[:, 2] > v[:, 0])
        ghz = (v[ok, 2] - v[ok, 0]) / (v[ok, 3] - v[ok, 1]) * 0.1
        us = (v[ok, 3] - v[ok, 1]) / 100.0
        out[name] = dict(workgroups=int(ok.sum()), clock_GHz_median=round(float(np.median(ghz)), 3), clock_GHz_min=round(float(ghz.min()), 3),
                         clock_GHz_max=round(float(ghz.max()), 3), span_us_median=round(float(np.median(us)), 2))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
