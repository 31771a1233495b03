"""Configuration constants shared between tools/make_golden.py and the tests (kept identical)."""
import numpy as np

TINY = dict(filter_width=2, dilations=[1, 2, 4, 8, 1, 2, 4, 8], dilation_channels=16,
            residual_channels=16, skip_channels=32, quantization_channels=256, use_bias=False)


def g3_inputs():
    """Inputs of the chunk-softmax fixture, regenerated from a fixed numpy seed (not stored)."""
    rng = np.random.default_rng(3)
    return {w: (4.0 * rng.standard_normal((1, 256, w))).astype(np.float32) for w in (1, 130, 255, 256)}
