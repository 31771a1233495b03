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


def nonvacuous(p_ref, what, floor=0.05):
    """A probability comparison at an absolute 1e-3 only says something when the reference distribution is not flat
    (default-initialised models sit within 1.1e-4 of 1/256 everywhere: SURVEY Q11): the largest reference probability must
    exceed `floor` (1/256 = 0.0039).  Prints it, so the log shows how peaked each fixture is."""
    m = float(torch.as_tensor(p_ref).max())
    print("  non-vacuity (%s): largest reference probability %.4f (floor %.3f)" % (what, m, floor))
    assert m > floor, (what, m, floor)
    return m


def g1_input(d, meta):
    if meta["kind"] == "randn":
        return torch.from_numpy(d["x"])
    return scrambled_input(d["idx"])
