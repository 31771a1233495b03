#!/usr/bin/env python3
"""Developer tool (CPU): which products of the FORWARD could run on two MFMAs instead of three?

Operand-rounding emulation on the float64 oracle, as tools/emu_bwd.py does for the backward: every channel-mixing
product is evaluated in float64 from f16 hi/lo operands, with one product class at a time reduced to

    x2w   weight lo dropped      W_hi x_hi + W_hi x_lo
    x2a   activation lo dropped  W_hi x_hi + W_lo x_hi
    x1    both dropped           W_hi x_hi

while every other product keeps the three-term form; prints max |d pre-softmax| and max |d probability| against exact
float64 (bar: 1e-3 on the probabilities AND on gain-scaled pre-softmax values, BASELINE.json north_star / SURVEY Q11).

    python tools/emu_fwd.py [T] [gain]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import intops  # noqa: E402
from oracle import wavenet_oracle as wo  # noqa: E402
from emu_bwd import C2, make_params, split  # noqa: E402


def mm(w, x, form):
    """w (M,K), x (B,K,L) float64 -> emulated w @ x"""
    if form == "exact":
        return torch.einsum("mk,bkl->bml", w, x)
    wh, wl = split(w, "f16")
    xh, xl = split(x, "f16")
    out = torch.einsum("mk,bkl->bml", wh, xh)
    if form in ("x3", "x2a"):
        out = out + torch.einsum("mk,bkl->bml", wl, xh)
    if form in ("x3", "x2w"):
        out = out + torch.einsum("mk,bkl->bml", wh, xl)
    return out


def forward(params, dil, x, forms):
    """forms: dict product class ('fg','dense','skip','p1','p2') -> form"""
    rf = wo.receptive_field(2, dil)
    W = x.shape[2] - rf + 1
    h = F.conv1d(x, params["causal_layer.weight"])          # a gather of weight columns in the kernels: exact
    u = None
    for i, d in enumerate(dil):
        wf = params["dilation_layer_stack.%d.weight" % (4 * i)]
        wg = params["dilation_layer_stack.%d.weight" % (4 * i + 1)]
        wd = params["dilation_layer_stack.%d.weight" % (4 * i + 2)][:, :, 0]
        ws = params["dilation_layer_stack.%d.weight" % (4 * i + 3)][:, :, 0]
        L = h.shape[2]
        xx = torch.cat([h[:, :, :L - d], h[:, :, d:]], 1)
        wfg = torch.cat([torch.cat([wf[:, :, 0], wf[:, :, 1]], 1), torch.cat([wg[:, :, 0], wg[:, :, 1]], 1)], 0)
        fg = mm(wfg, xx, forms["fg"])
        D = wf.shape[0]
        z = torch.tanh(fg[:, :D]) * torch.sigmoid(fg[:, D:])
        h = mm(wd, z, forms["dense"]) + h[:, :, d:]
        s = mm(ws, z[:, :, -W:], forms["skip"])
        u = s if u is None else u + s
    a1 = F.relu(u)
    hh = mm(params["post_process_1.weight"][:, :, 0], a1, forms["p1"])
    o = mm(params["post_process_2.weight"][:, :, 0], F.relu(hh), forms["p2"])
    return o, wo.chunk_softmax(o, 256)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    gain = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
    dil = C2["dilations"]
    torch.manual_seed(3)
    params = {k: (v * gain).double() for k, v in make_params(dil, 64, 64, 256).items()}
    rng = np.random.default_rng(31)
    codes = rng.integers(0, 256, size=(2, T))
    x = torch.from_numpy(np.stack([intops.one_hot_scrambled(r, 256) for r in codes])).double()
    torch.set_num_threads(8)
    classes = ("fg", "dense", "skip", "p1", "p2")
    o_ex, p_ex = forward(params, dil, x, {c: "exact" for c in classes})
    print("c2, 2 x %d, gain %.1f: |pre-softmax| max %.1f, max probability %.3f" % (T, gain, o_ex.abs().max(), p_ex.max()))
    print("%-22s %14s %14s" % ("products on < 3 MFMAs", "d pre-softmax", "d probability"))

    def row(tag, forms):
        o, p = forward(params, dil, x, forms)
        print("%-22s %14.2e %14.2e" % (tag, (o - o_ex).abs().max(), (p - p_ex).abs().max()))
    base = {c: "x3" for c in classes}
    row("none (all x3)", base)
    for c in classes:
        for form in ("x2w", "x2a", "x1"):
            row("%s %s" % (c, form), dict(base, **{c: form}))
    row("epilogue (skip,p1,p2) x2w", dict(base, skip="x2w", p1="x2w", p2="x2w"))
    row("epilogue (skip,p1,p2) x2a", dict(base, skip="x2a", p1="x2a", p2="x2a"))


if __name__ == "__main__":
    main()
