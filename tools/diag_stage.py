#!/usr/bin/env python3
"""Developer diagnostic (GPU): which KERNEL of the backward epilogue loses accuracy?  Runs the aligned config-2 case of
tools/diag_fullsize.py, then recomputes every epilogue product in float64 on the host FROM THE DEVICE'S OWN INPUTS of
that product (so only that kernel's arithmetic is measured, not what it inherited):

    dH  = (P2^T dO) * [H > 0]       dWp2 = sum dO  relu(H)^T
    dU  = (P1^T dH) * [U > 0]       dWp1 = sum dH  relu(U)^T
    dZ  = Ws^T dU                   dWs  = sum dU  Z^T
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import scrambled_input  # noqa: E402
from tests.test_gpu_fullsize import C2, _scaled  # noqa: E402
from music_amd.engine import SLACK  # noqa: E402


def view(buf, B, rows, pitch):
    return buf[SLACK:SLACK + B * rows * pitch].view(B, rows, pitch)


def main():
    from music_amd.model import wavenet
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
    torch.manual_seed(3)
    net = wavenet(**C2)
    params = _scaled(net, 2.5)
    net = net.cuda()
    B = 2
    rf = net.receptive_field
    W = T - rf + 1
    codes = (np.arange(T)[None, :] * 37 % 7 * 31 + 11 + np.arange(B)[:, None]) % 256
    x = scrambled_input(codes)
    target = torch.from_numpy(np.full((B * W,), 7, dtype=np.int64))
    eng = net._engine_for(torch.device("cuda", 0))
    eng.overlap_wgrad = os.environ.get("DIAG_OVERLAP", "1") == "1"
    loss = eng.loss_and_grad(x.cuda(), target.cuda(), want_probs=True)
    torch.cuda.synchronize()
    ws = eng.workspace(B, T)
    bw = ws["bwd"]
    pitch, lo = ws["pitch"], rf - 1
    SP, N, CH, Q = eng.SP, eng.N, eng.CH, eng.Q
    dO = bw["dO"][:B * Q * W].view(B, Q, W).cpu().double()
    H = view(ws["H"], B, SP, pitch)[:, :, lo:T].cpu().double()
    U = view(ws["U"], B, SP, pitch)[:, :, lo:T].cpu().double()
    dH = view(bw["dH"], B, SP, pitch)[:, :, lo:T].cpu().double()
    dU = view(bw["dU"], B, SP, pitch)[:, :, lo:T].cpu().double()
    Z = view(ws["Z"], B, N * CH, pitch)[:, :, lo:T].cpu().double()
    dZ = view(bw["dZ"], B, N * CH, pitch)[:, :, lo:T].cpu().double()
    P2 = params["post_process_2.weight"][:, :, 0].double()
    P1 = params["post_process_1.weight"][:, :, 0].double()
    Ws = torch.cat([params["dilation_layer_stack.%d.weight" % (4 * i + 3)][:, :, 0] for i in range(N)], 1).double()   # (S, N*D)
    torch.set_num_threads(min(32, os.cpu_count()))

    def rel(a, b):
        return ((a - b).abs().max() / b.abs().max()).item()

    print("loss %.7f" % loss.item())
    print("max|dO| %.3e  |dH| %.3e  |dU| %.3e  |dZ| %.3e" % (dO.abs().max(), dH.abs().max(), dU.abs().max(), dZ.abs().max()))
    dH_ref = torch.einsum("qs,bqw->bsw", P2, dO) * (H > 0)
    print("dH  kernel error (vs f64 from its own inputs): %.2e" % rel(dH, dH_ref))
    dU_ref = torch.einsum("hs,bhw->bsw", P1, dH) * (U > 0)
    print("dU  kernel error: %.2e" % rel(dU, dU_ref))
    dZ_ref = torch.einsum("sk,bsw->bkw", Ws, dU)
    print("dZ  kernel error: %.2e" % rel(dZ, dZ_ref))
    g = lambda n: eng.param_view(n, grad=True).cpu().double()
    gp2 = torch.einsum("bqw,bsw->qs", dO, H.clamp(min=0))
    print("dWp2 kernel error: %.2e   (|g|max %.2e)" % (rel(g("post_process_2.weight")[:, :, 0], gp2), gp2.abs().max()))
    gp1 = torch.einsum("bhw,bsw->hs", dH, U.clamp(min=0))
    print("dWp1 kernel error: %.2e   (|g|max %.2e)" % (rel(g("post_process_1.weight")[:, :, 0], gp1), gp1.abs().max()))
    gs = torch.einsum("bsw,bkw->sk", dU, Z)
    worst = 0.0
    for i in range(N):
        got = g("dilation_layer_stack.%d.weight" % (4 * i + 3))[:, :, 0]
        e = rel(got, gs[:, i * CH:i * CH + eng.D])
        worst = max(worst, e)
        if i in (0, 4, 14, 29):
            print("dWs[%d] kernel error: %.2e  (|g|max %.2e; sum|terms| %.2e)" % (i, e, gs[:, i * CH:(i + 1) * CH].abs().max(),
                                                                               torch.einsum("bsw,bkw->sk", dU.abs(), Z[:, i * CH:(i + 1) * CH].abs()).max()))
    print("dWs worst kernel error over layers: %.2e" % worst)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "oracle"):
    main()


def main2():
    """Second pass: GPU intermediates against the float64 ORACLE's (inherited error, ReLU mask flips)."""
    import torch.nn.functional as F
    from oracle import wavenet_oracle as wo
    from music_amd.model import wavenet
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
    torch.manual_seed(3)
    net = wavenet(**C2)
    params = _scaled(net, 2.5)
    net = net.cuda()
    B = 2
    rf = net.receptive_field
    W = T - rf + 1
    codes = (np.arange(T)[None, :] * 37 % 7 * 31 + 11 + np.arange(B)[:, None]) % 256
    x = scrambled_input(codes)
    target = torch.from_numpy(np.full((B * W,), 7, dtype=np.int64))
    eng = net._engine_for(torch.device("cuda", 0))
    eng.loss_and_grad(x.cuda(), target.cuda(), want_probs=True)
    torch.cuda.synchronize()
    ws = eng.workspace(B, T)
    bw = ws["bwd"]
    pitch, lo = ws["pitch"], rf - 1
    SP, N, CH, Q = eng.SP, eng.N, eng.CH, eng.Q
    torch.set_num_threads(min(32, os.cpu_count()))
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        p = {k: v.to(dt) for k, v in params.items()}
        inter = {}
        probs = wo.wavenet_forward(p, C2["dilations"], x.to(dt), intermediates=inter)
        u = inter["skip_sum"]
        a1 = F.relu(u)
        h = F.conv1d(a1, p["post_process_1.weight"])
        n = B * W
        sm = F.softmax(probs, dim=1)
        dp = sm.clone()
        dp[torch.arange(n), target] -= 1.0
        dp /= n
        dO = (probs * (dp - (dp * probs).sum(1, keepdim=True))).view(B, Q, W)
        dH = torch.einsum("qs,bqw->bsw", p["post_process_2.weight"][:, :, 0], dO) * (h > 0)
        dU = torch.einsum("hs,bhw->bsw", p["post_process_1.weight"][:, :, 0], dH) * (u > 0)
        if tag == "f64":
            ref = dict(u=u, h=h, dO=dO, dH=dH, dU=dU, z=inter["z"])
        else:
            c32 = dict(u=u.double(), h=h.double(), dO=dO.double(), dH=dH.double(), dU=dU.double())
    g = dict(u=view(ws["U"], B, SP, pitch)[:, :, lo:T].cpu().double(), h=view(ws["H"], B, SP, pitch)[:, :, lo:T].cpu().double(),
             dO=bw["dO"][:B * Q * W].view(B, Q, W).cpu().double(), dH=view(bw["dH"], B, SP, pitch)[:, :, lo:T].cpu().double(),
             dU=view(bw["dU"], B, SP, pitch)[:, :, lo:T].cpu().double())
    for k in ("u", "h", "dO", "dH", "dU"):
        r = ref[k]
        print("%-3s max|ref| %.3e   gpu: max err %.2e (rel to max), mean|err|/mean|ref| %.2e   cpu32: %.2e, %.2e" % (
            k, r.abs().max(), (g[k] - r).abs().max() / r.abs().max(), (g[k] - r).abs().mean() / r.abs().mean(),
            (c32[k] - r).abs().max() / r.abs().max(), (c32[k] - r).abs().mean() / r.abs().mean()))
    for k in ("u", "h"):
        fl_g = ((g[k] > 0) != (ref[k] > 0))
        fl_c = ((c32[k] > 0) != (ref[k] > 0))
        print("mask flips in %s: gpu %d (max |ref| there %.2e), cpu32 %d" % (k, int(fl_g.sum()), float(ref[k][fl_g].abs().max()) if fl_g.any() else 0.0, int(fl_c.sum())))
    # contribution of the dU difference to dWs of layer 4
    z4 = ref["z"][4][:, :, -W:]
    for nm, src in (("gpu", g), ("cpu32", c32)):
        dd = src["dU"] - ref["dU"]
        e = torch.einsum("bsw,bkw->sk", dd, z4)
        full = torch.einsum("bsw,bkw->sk", ref["dU"], z4)
        print("dWs[4] error inherited through dU (%s): %.2e of |g|max; signed mean err of dU / mean|dU| %.2e" % (
            nm, e.abs().max() / full.abs().max(), dd.mean() / ref["dU"].abs().mean()))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "oracle":
    main2()
