"""Shared helpers for the tests (fixture loading, config shorthands)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def g1_meta():
    with open(os.path.join(GOLDEN, "g1_meta.json")) as f:
        return json.load(f)


def params_from(d, prefix="w:"):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in d.items() if k.startswith(prefix)}


def grads_from(d):
    return params_from(d, "g:")


def scrambled_input(idx, q=256):
    """(B,T) int -> (B,q,T) float32 loader-faithful one-hot (oracle.intops)."""
    from oracle import intops
    return torch.from_numpy(np.stack([intops.one_hot_scrambled(r, q) for r in idx]))


def g1_input(d, meta):
    if meta["kind"] == "randn":
        return torch.from_numpy(d["x"])
    return scrambled_input(d["idx"])
