#!/usr/bin/env python3
"""Randomised parity fuzz of the cached-queue decoder at the matrix-core shapes (up to 64 residual / dilation channels -
narrower models run padded -, 256 or 512 skip channels): random depth (1..44 blocks, odd depths too; more than 32 blocks keep
the tap-0 table in the hand-off area) and dilations, with and without biases, both queue recurrences, teacher-forced
codes; argmax ids and probabilities of every step against the oracle's cached recurrence (oracle/wavenet_oracle.py),
and a multiple-of-8 batch (eight utterances per workgroup pair) against single-utterance launches.  Test
infrastructure (imports oracle/); not part of the product path.

    python tools/fuzz_decode.py [--cases N] [--seed S]          (WN_DEC_PIPE=1 / WN_DEC_MFMA=0 select the other kernels)"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import wavenet_oracle as wo  # noqa: E402
from oracle import intops  # noqa: E402


def onehot(ix):
    return torch.from_numpy(intops.one_hot_proper(np.atleast_1d(ix)))[None]


def one_case(rng, k, wide_shapes=False, only=None, verbose=False):
    from music_amd import fast_generate as fg
    from music_amd.model import wavenet
    n = int(rng.integers(1, 45 if wide_shapes else 31))
    dil = [int(rng.choice([1, 2, 3, 4, 8, 16, 5, 32])) for _ in range(n)]
    bias = bool(rng.random() < 0.5)
    correct = bool(rng.random() < 0.5)
    R = D = 64
    S = 256
    if wide_shapes:
        R = int(rng.choice([64, 64, 32, int(rng.integers(8, 65))]))
        D = int(rng.choice([R, 64, 32, int(rng.integers(8, 65))]))
        S = int(rng.choice([256, 512]))
    cfg = dict(filter_width=2, dilations=dil, dilation_channels=D, residual_channels=R, skip_channels=S,
               quantization_channels=256, use_bias=bias)
    torch.manual_seed(900 + k)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(float(rng.uniform(1.5, 3.0)))
    params = {kk: v.clone() for kk, v in net.state_dict().items()}
    steps = int(rng.integers(6, 16))
    start = rng.integers(0, 256, size=(net.receptive_field,))
    forced = rng.integers(0, 256, size=(steps,))
    U, m = int(rng.choice([3, 8, 13, 16])), 40
    idx = torch.from_numpy(rng.integers(0, 256, size=(U, net.receptive_field)))
    if only is not None and k != only:          # same random stream, no compute
        return True
    if verbose:
        print("     case %3d  R/D/S=%d/%d/%d blocks=%2d dil=%s bias=%d correct_queue=%d steps=%d U=%d" % (k, R, D, S, n, dil, bias, correct, steps, U), flush=True)
    net = net.cuda()
    pred_o, q_o, _ = wo.fast_predict_next(params, dil, onehot(start), None, return_probs=True)
    want, want_p = [int(pred_o[0])], []
    for s in forced:
        pred_o, q_o, pr = wo.fast_predict_next(params, dil, onehot(s), q_o, correct_queue=correct, return_probs=True)
        want.append(int(pred_o[0]))
        want_p.append(pr.numpy())
    pred, st = fg.predict_next(net, onehot(start).cuda(), None)
    got = [int(pred[0])]
    nxt = torch.from_numpy(np.concatenate([forced[1:], [0]]).astype(np.int32))
    codes, probs, _ = fg._decode(net, st, onehot(forced[0]).reshape(-1).cuda(), steps, forced=nxt, want_probs=True,
                                 correct_queue=correct)
    got += codes.cpu().tolist()
    err = float(np.abs(probs.cpu().numpy() - np.stack(want_p)).max())
    ok = got == want and err < 1e-4
    # eight utterances per pair against single launches (greedy, free-running)
    starts = torch.zeros(U, 256, net.receptive_field)
    starts.scatter_(1, idx[:, None, :], 1.0)
    batch = fg.generate_codes_batch(net, starts.cuda(), m, correct_queue=correct)
    same = all(torch.equal(batch[u], fg.generate_codes(net, starts[u:u + 1].cuda(), m, correct_queue=correct).view(-1)) for u in sorted({0, U // 2, U - 1}))
    print("%s case %3d  R/D/S=%d/%d/%d blocks=%2d dil=%s bias=%d correct_queue=%d steps=%d  probs err %.1e  codes %s  batch-of-8 rows %s" % (
        "ok  " if ok and same else "FAIL", k, R, D, S, n, dil[:6] + (["..."] if n > 6 else []), bias, correct, steps, err,
        "equal" if got == want else "DIFFER", "equal" if same else "DIFFER"))
    return ok and same


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--shapes", action="store_true", help="also draw narrower models, 512 skip channels and up to 44 blocks")
    ap.add_argument("--only", type=int, default=None, help="run just this case of the stream")
    ap.add_argument("--verbose", action="store_true", help="print a case's shape before it runs")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    torch.set_num_threads(8)
    bad = sum(0 if one_case(rng, k, a.shapes, a.only, a.verbose) else 1 for k in range(a.cases))
    print("%d / %d cases failed" % (bad, a.cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
