#!/usr/bin/env python3
"""Developer tool (CPU): operand-rounding emulation of the BACKWARD arithmetic of the HIP path on the float64 oracle.

Every product of the backward is evaluated in float64 from operands that were first rounded the way
the kernels round them - a two-term 16-bit split x = hi + lo, product = hi*hi + lo*hi + hi*lo - for each candidate
arithmetic:

    bf16x3      both operands split in bf16 (round 2's backward: 2^-17 per element, any range)
    f16x3@k     both operands split in f16, dO pre-multiplied by 2^k (2^-22 per element while the operand's magnitude
                stays inside f16's range; subnormal / overflow behaviour is torch's IEEE half conversion)
    *x2w        the weight operand's lo term dropped (hi*hi + hi_w*lo_g), i.e. two MFMAs instead of three

and compared with the exact float64 gradients.  Also prints max|.| of every gradient tensor (range budget of a static
scale).  Test infrastructure: imports oracle/, never imported by music_amd/.

    python tools/emu_bwd.py [c2|tiny] [T] [gain] [aligned|random]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import intops  # noqa: E402
from oracle import wavenet_oracle as wo  # noqa: E402

C2 = dict(dilations=[2 ** i for i in range(10)] * 3, D=64, R=64, S=256)
TINY = dict(dilations=[1, 2, 4, 8] * 2, D=16, R=16, S=32)


def split(x, kind):
    """-> (hi, lo) as float64 tensors; kind in {'exact','bf16','f16'}"""
    if kind == "exact":
        return x, torch.zeros_like(x)
    dt = torch.bfloat16 if kind == "bf16" else torch.float16
    x32 = x.float()
    hi = x32.to(dt).float()
    lo = (x32 - hi).to(dt).float()
    return hi.double(), lo.double()


class Arith:
    def __init__(self, kind, terms=3, wterms=None):
        self.kind, self.terms = kind, terms
        self.wterms = wterms if wterms is not None else terms        # 2: weight lo dropped (only for weight x gradient products)
        self.amax = {}

    def mm(self, a, b, a_is_weight=False, name=None):
        """a: (M,K), b: (K,N) float64 -> emulated a @ b"""
        if name:
            self.amax[name] = max(self.amax.get(name, 0.0), float(b.abs().max()) if a_is_weight else float(a.abs().max()))
        if self.kind == "exact":
            return a @ b
        ah, al = split(a, self.kind)
        bh, bl = split(b, self.kind)
        out = ah @ bh + ah @ bl
        if not (a_is_weight and self.wterms == 2):
            out = out + al @ bh
        return out


def backward_emulated(params, dil, x, target, ar, scale=1.0):
    """Manual backward (Appendix B) with every product through ar.mm.  params float64.  Returns name -> grad (unscaled)."""
    N = len(dil)
    inter = {}
    probs = wo.wavenet_forward(params, dil, x, intermediates=inter)
    B, Q, T = x.shape
    rf = wo.receptive_field(2, dil)
    W = T - rf + 1
    n = B * W
    xs, zs = inter["x"], inter["z"]
    u = inter["skip_sum"]
    a1 = F.relu(u)
    h = F.conv1d(a1, params["post_process_1.weight"])
    a2 = F.relu(h)
    # CE on probabilities (train.py:146,179)
    sm = F.softmax(probs, dim=1)
    dp = sm.clone()
    dp[torch.arange(n), target] -= 1.0
    dp /= n
    dchunk = probs * (dp - (dp * probs).sum(1, keepdim=True))
    dO = dchunk.view(B, Q, W) * scale
    g = {}

    def cols(t):          # (B,C,L) -> (C, B*L)
        return t.permute(1, 0, 2).reshape(t.shape[1], -1)

    def uncols(m, Bn):
        return m.view(m.shape[0], Bn, -1).permute(1, 0, 2)

    P2 = params["post_process_2.weight"][:, :, 0]
    P1 = params["post_process_1.weight"][:, :, 0]
    g["post_process_2.weight"] = ar.mm(cols(dO), cols(a2).t(), name="dO")[:, :, None]
    dH = uncols(ar.mm(P2.t(), cols(dO), a_is_weight=True), B) * (h > 0)
    g["post_process_1.weight"] = ar.mm(cols(dH), cols(a1).t(), name="dH")[:, :, None]
    dU = uncols(ar.mm(P1.t(), cols(dH), a_is_weight=True), B) * (u > 0)
    dx = None
    for i in range(N - 1, -1, -1):
        d = dil[i]
        xi = xs[i]
        L = xi.shape[2]
        Lo = L - d
        xm, xp = xi[:, :, :Lo], xi[:, :, d:]
        Wf = params["dilation_layer_stack.%d.weight" % (4 * i)]
        Wg = params["dilation_layer_stack.%d.weight" % (4 * i + 1)]
        Wd = params["dilation_layer_stack.%d.weight" % (4 * i + 2)][:, :, 0]
        Ws = params["dilation_layer_stack.%d.weight" % (4 * i + 3)][:, :, 0]
        z = zs[i]
        g["dilation_layer_stack.%d.weight" % (4 * i + 3)] = ar.mm(cols(dU), cols(z[:, :, -W:]).t(), name="dU")[:, :, None]
        dz = torch.zeros_like(z)
        dz[:, :, -W:] = uncols(ar.mm(Ws.t(), cols(dU), a_is_weight=True), B)
        if dx is not None:
            g["dilation_layer_stack.%d.weight" % (4 * i + 2)] = ar.mm(cols(dx), cols(z).t(), name="dy%d" % i)[:, :, None]
            dz = dz + uncols(ar.mm(Wd.t(), cols(dx), a_is_weight=True), B)
        else:
            g["dilation_layer_stack.%d.weight" % (4 * i + 2)] = torch.zeros_like(params["dilation_layer_stack.%d.weight" % (4 * i + 2)])
        f = F.conv1d(xi, Wf, dilation=d)
        gg = F.conv1d(xi, Wg, dilation=d)
        th, sg = torch.tanh(f), torch.sigmoid(gg)
        df = dz * sg * (1 - th * th)
        dg = dz * th * sg * (1 - sg)
        dfg = torch.cat([df, dg], 1)
        xx = torch.cat([xm, xp], 1)
        gw = ar.mm(cols(dfg), cols(xx).t(), name="dfg%d" % i)          # (2D, 2R)
        D, R = Wf.shape[0], Wf.shape[1]
        g["dilation_layer_stack.%d.weight" % (4 * i)] = torch.stack([gw[:D, :R], gw[:D, R:]], 2)
        g["dilation_layer_stack.%d.weight" % (4 * i + 1)] = torch.stack([gw[D:, :R], gw[D:, R:]], 2)
        W0 = torch.cat([Wf[:, :, 0], Wg[:, :, 0]], 0)           # (2D, R)
        W1 = torch.cat([Wf[:, :, 1], Wg[:, :, 1]], 0)
        Pm = uncols(ar.mm(W1.t(), cols(dfg), a_is_weight=True), B)
        Qm = uncols(ar.mm(W0.t(), cols(dfg), a_is_weight=True), B)
        ndx = torch.zeros_like(xi)
        ndx[:, :, d:] += Pm
        ndx[:, :, :Lo] += Qm
        if dx is not None:
            ndx[:, :, d:] += dx
        dx = ndx
    # causal layer: scatter (exact fp32 sums in the kernels)
    ar.amax["dx0"] = float(dx.abs().max())
    xin = torch.cat([x[:, :, :-1], x[:, :, 1:]], 1)
    gw = cols(dx) @ cols(xin).t()
    Qn = x.shape[1]
    g["causal_layer.weight"] = torch.stack([gw[:, :Qn], gw[:, Qn:]], 2)
    return {k: v / scale for k, v in g.items()}


def make_params(dil, R, D, S, Q=256):
    """default nn.Conv1d init in the reference's construction order (wavenet/model.py:46-84), no biases"""
    mk = lambda o, i, k: torch.nn.Conv1d(i, o, k, bias=False).weight.detach()
    p = {"causal_layer.weight": mk(R, Q, 2)}
    for i in range(len(dil)):
        for k, (o, ii, kk) in enumerate([(D, R, 2), (D, R, 2), (R, D, 1), (S, D, 1)]):
            p["dilation_layer_stack.%d.weight" % (4 * i + k)] = mk(o, ii, kk)
    p["post_process_1.weight"] = mk(S, S, 1)
    p["post_process_2.weight"] = mk(Q, S, 1)
    return p


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    gain = float(sys.argv[3]) if len(sys.argv) > 3 else 2.5
    mode = sys.argv[4] if len(sys.argv) > 4 else "aligned"
    cfg = C2 if which == "c2" else TINY
    dil = cfg["dilations"]
    torch.manual_seed(3)
    params = make_params(dil, cfg["R"], cfg["D"], cfg["S"])
    params = {k: (v * gain).double() for k, v in params.items()}
    B = 2
    rf = wo.receptive_field(2, dil)
    W = T - rf + 1
    rng = np.random.default_rng(31)
    if mode == "aligned":
        codes = (np.arange(T)[None, :] * 37 % 7 * 31 + 11 + np.arange(B)[:, None]) % 256
        target = torch.from_numpy(np.full((B * W,), 7, dtype=np.int64))
    else:
        codes = rng.integers(0, 256, size=(B, T))
        target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))
    x = torch.from_numpy(np.stack([intops.one_hot_scrambled(r, 256) for r in codes])).double()
    torch.set_num_threads(8)
    exact = Arith("exact")
    g_ex = backward_emulated(params, dil, x, target, exact)
    # sanity: the manual backward equals autograd
    l64, _, g_auto = wo.loss_and_grads(params, dil, x, target)
    worst = max(((g_ex[k] - g_auto[k]).abs().max() / max(g_auto[k].abs().max(), 1e-300)).item() for k in g_ex)
    print("manual-vs-autograd (f64): %.2e   loss %.6f   N=%d (2^%.1f)" % (worst, l64.item(), B * W, np.log2(B * W)))
    am = exact.amax
    keys = ["dO", "dH", "dU"] + ["dy%d" % i for i in range(len(dil) - 2, -1, -1)] + ["dfg%d" % i for i in range(len(dil) - 1, -1, -1)] + ["dx0"]
    vals = {k: am[k] for k in keys if k in am}
    lg = {k: np.log2(v) for k, v in vals.items() if v > 0}
    print("max|.| (log2): dO %.1f dH %.1f dU %.1f | dy min %.1f max %.1f | dfg min %.1f max %.1f | dx0 %.1f" % (
        lg["dO"], lg["dH"], lg["dU"],
        min(v for k, v in lg.items() if k.startswith("dy")), max(v for k, v in lg.items() if k.startswith("dy")),
        min(v for k, v in lg.items() if k.startswith("dfg")), max(v for k, v in lg.items() if k.startswith("dfg")), lg["dx0"]))
    # the float32 path for comparison
    p32 = {k: v.float() for k, v in params.items()}
    _, _, g32 = wo.loss_and_grads(p32, dil, x.float(), target)

    def report(tag, gq):
        errs = {k: ((gq[k].double() - g_ex[k]).abs().max() / max(g_ex[k].abs().max(), 1e-300)).item() for k in g_ex}
        w = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print("%-14s worst %.2e  (%s)" % (tag, w[0][1], ", ".join("%s %.1e" % (k.replace("dilation_layer_stack", "dls").replace(".weight", ""), v) for k, v in w)))

    report("cpu float32", g32)
    report("bf16x3", backward_emulated(params, dil, x, target, Arith("bf16")))
    k0 = int(np.floor(-lg["dO"]))            # 2^k0 * max|dO| in [1, 2)
    for dk in (-8, -4, 0, 4, 8, 12):
        report("f16x3@2^%d%+d" % (k0, dk), backward_emulated(params, dil, x, target, Arith("f16"), scale=2.0 ** (k0 + dk)))
    report("f16x2w@2^%d+4" % k0, backward_emulated(params, dil, x, target, Arith("f16", 3, 2), scale=2.0 ** (k0 + 4)))
    report("bf16x2w", backward_emulated(params, dil, x, target, Arith("bf16", 3, 2)))


if __name__ == "__main__":
    main()
