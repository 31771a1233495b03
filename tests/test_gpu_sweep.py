"""GPU parity sweep over RAGGED shapes: channel counts that are not multiples of 16 (padded rows in
every packed matrix), odd clip lengths (column tiles that end mid-float4, unaligned (B,Q,W) rows),
dilations that are not multiples of 4 (unaligned shifted taps), batch 1..3, bias on and off, one
output column (W = 1) up to several 512-column tiles.  Every case: pre-softmax logits and
probabilities within 1e-3 of the CPU oracle, loss within 1e-4, every gradient within 3e-4 of the
tensor's max-abs, through both the nn.Module surface and the fused training step.  Run with -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import wavenet_oracle as wo
from tests.helpers import nonvacuous, scrambled_input

LOGIT_TOL = 1e-3
GRAD_RTOL = 3e-4
# a gradient that is analytically zero (post_process_2.bias: the chunk softmax removes any per-row
# constant) is rounding noise of order 1e-12 in the oracle and on the GPU alike
GRAD_FLOOR_REL = 1e-3        # ... so a tensor's scale is at least this fraction of the largest gradient of the model

CASES = [
    # (dilations, R, D, S, bias, B, extra columns W-1, gain)
    ([1, 2, 4], 16, 16, 32, False, 1, 0, 3.0),
    ([1, 2, 4, 8, 1, 3], 24, 20, 40, True, 2, 37, 3.0),
    ([5, 1, 7], 33, 17, 65, False, 3, 129, 2.5),
    ([1, 2, 4, 8, 16, 32, 64], 32, 32, 256, True, 2, 700, 2.5),
    ([3, 9, 27, 81], 48, 64, 100, False, 1, 1030, 2.5),
    ([1, 2, 4, 8, 16, 32, 64, 128, 256, 512], 64, 64, 128, False, 2, 515, 2.0),
    ([2, 6], 64, 40, 24, True, 3, 511, 3.0),
    ([1, 1, 1, 1], 8, 8, 8, True, 2, 3, 4.0),
    # <= 32 channels, no bias, even batch: clip PAIRS on the 64-channel block kernels (block-diagonal packs, music_amd/engine.py)
    ([1, 2, 4, 8, 16], 32, 32, 64, False, 2, 300, 2.5),
    ([3, 1, 9, 2], 20, 28, 48, False, 4, 77, 3.0),
]


def _ids(c):
    return "d%d_R%d_D%d_S%d_%s_B%d_W%d" % (len(c[0]), c[1], c[2], c[3], "bias" if c[4] else "nobias", c[5], c[6] + 1)


@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_ragged_shapes_vs_oracle(case):
    from music_amd.model import wavenet
    dil, R, D, S, bias, B, extra, gain = case
    cfg = dict(filter_width=2, dilations=dil, dilation_channels=D, residual_channels=R, skip_channels=S,
               quantization_channels=256, use_bias=bias)
    torch.manual_seed(1000 + len(dil) * 7 + R)
    net = wavenet(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(gain)
    params = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    rng = np.random.default_rng(R * 131 + extra)
    T = net.receptive_field + extra
    W = extra + 1
    x = scrambled_input(rng.integers(0, 256, size=(B, T)))
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64))

    l_ref, p_ref, g_ref = wo.loss_and_grads(params, dil, x, target)
    GRAD_FLOOR = GRAD_FLOOR_REL * max(g.abs().max().item() for g in g_ref.values())
    inter = {}
    with torch.no_grad():
        wo.wavenet_forward(params, dil, x, intermediates=inter)
    pre_ref = inter["pre_softmax"]

    # 1. nn.Module surface: forward + autograd backward
    probs = net(x.cuda())
    assert probs.shape == (B * W, 256)
    e_p = (probs.detach().cpu() - p_ref).abs().max().item()
    eng = net._engine
    ws = eng.workspace(B, T)
    pre = ws["O"][:B * 256 * W].view(B, 256, W).cpu()
    e_pre = (pre - pre_ref.reshape(B, 256, W)).abs().max().item()
    assert e_pre <= LOGIT_TOL, e_pre
    assert e_p <= LOGIT_TOL, e_p
    # shallow ragged models: what the absolute 1e-3 bar bites on here is the PRE-SOFTMAX (|max| 0.8 ... 34 over the cases), the
    # probabilities only have to be off the uniform distribution (2 / 256)
    assert pre_ref.abs().max().item() > 0.5
    nonvacuous(p_ref, "sweep, |pre-softmax| up to %.2f" % pre_ref.abs().max().item(), 2.0 / 256)
    loss = torch.nn.CrossEntropyLoss()(probs, target.cuda())
    assert abs(loss.item() - l_ref.item()) < 1e-4
    loss.backward()
    worst = 0.0
    for name, p in net.named_parameters():
        g = g_ref[name]
        err = (p.grad.cpu() - g).abs().max().item() / max(g.abs().max().item(), GRAD_FLOOR)
        worst = max(worst, err)
        assert err <= GRAD_RTOL, (name, err)

    # 2. fused training-step entry (forward + CE + backward in one call), fresh workspace
    eng._ws.clear()
    loss2 = eng.loss_and_grad(x.cuda(), target.cuda())
    assert abs(loss2.item() - l_ref.item()) < 1e-4
    for name in eng.param_names:
        g = g_ref[name]
        err = (eng.param_view(name, grad=True).cpu() - g).abs().max().item() / max(g.abs().max().item(), GRAD_FLOOR)
        assert err <= GRAD_RTOL, (name, err)
    print(_ids(case), "probs err %.2e  worst grad err %.2e" % (e_p, worst))


def test_too_short_input_raises():
    """wavenet/model.py:100-101: fewer samples than the receptive field is an error, not a result."""
    from music_amd.model import wavenet
    net = wavenet(filter_width=2, dilations=[1, 2, 4], dilation_channels=16, residual_channels=16,
                  skip_channels=16, quantization_channels=256, use_bias=False).cuda()
    x = torch.zeros(1, 256, net.receptive_field - 1, device="cuda")
    with pytest.raises(Exception):
        net(x)


def test_decode_fuzz_seeded():
    """tools/fuzz_decode.py as a seeded test: six random decoders (depth 1..30, odd depths, mixed
    dilations, with / without biases, both queue recurrences) - teacher-forced ids and probabilities against the oracle's
    cached recurrence, and ragged batches of the eight-per-pair kernel against single-utterance launches."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_decode", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                              "tools", "fuzz_decode.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(2026)
    torch.set_num_threads(8)
    assert all([fz.one_case(rng, k) for k in range(6)])
