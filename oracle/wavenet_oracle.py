"""CPU oracle for the WaveNet hot path (TEST INFRASTRUCTURE — never imported by music_amd/).

This is a from-scratch restatement, in functional torch-CPU float32 ops, of the arithmetic of
deep-art-project/Music's WaveNet path.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  Each function cites the reference file:line it follows
(paths relative to the reference checkout).

Parity pinning: the reference holds NO golden vectors or known-answer tests for this path
(SURVEY.md §4).  The oracle is pinned instead against outputs of the reference itself, produced
by tools/make_golden.py (which imports the reference in the build container) and committed as
tests/golden/*.npz; tests/test_oracle_golden.py checks every function here against them.

Parameters are passed as a plain dict keyed by the reference's state_dict names
(``causal_layer.weight``, ``dilation_layer_stack.{4i+k}.weight`` k=filter,gate,dense,skip,
``post_process_{1,2}.weight`` and the matching ``.bias`` when use_bias) so a checkpoint written by
either implementation can be fed straight in.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# wavenet/model.py
# --------------------------------------------------------------------------------------------
def receptive_field(filter_width, dilations):
    """wavenet/model.py:43-44 — (k-1)(sum(d)+1)+1."""
    return (filter_width - 1) * (sum(dilations) + 1) + 1


def _b(params, name):
    return params.get(name + ".bias", None)


def chunk_softmax(total, q):
    """wavenet/model.py:142-144 — ``total`` (B,Q,W) contiguous is viewed (-1,Q) WITHOUT a
    transpose and soft-maxed along dim 1 (nn.Softmax() on a 2-D input picks dim 1), so each row
    is Q consecutive floats of the channels-first buffer."""
    return F.softmax(total.contiguous().view(-1, q), dim=1)


def _relu(relu, name, t):
    """F.relu, or the test's override ``relu(name, pre_activation)``.  The override exists for ONE purpose: a
    full-size gradient comparison has to take the subgradient of ReLU at |pre-activation| < (forward tolerance)
    the way the implementation under test did - at T = 16000 a handful of the 6.6 M pre-activations of
    model.py:135,137 lie within 1e-5 of zero, two correct float32 forwards disagree on their sign, and each such
    element moves whole gradient tensors by ~1e-3 of their max-abs (tools/diag_stage.py)."""
    return F.relu(t) if relu is None else relu(name, t)


def wavenet_forward(params, dilations, wave_sample, filter_width=2, quantization_channels=256,
                    intermediates=None, relu=None):
    """wavenet/model.py:86-145.  Returns probabilities of shape (B*W, Q).

    If ``intermediates`` is a dict it receives 'x' (list of N+1 residual-stream tensors),
    'z' (list of N gated activations), 'skip_sum', 'post_process_1' (both pre-ReLU), 'pre_softmax'.
    ``relu``: see _relu (names 'skip_sum', 'post_process_1').
    """
    rf = receptive_field(filter_width, dilations)
    out_w = wave_sample.size(2) - rf + 1
    if out_w <= 0:                                            # model.py:100-101
        raise ValueError("wave sample not long enough")
    x = F.conv1d(wave_sample, params["causal_layer.weight"], _b(params, "causal_layer"))  # :104
    xs, zs = [x], []
    skip_sum = None
    for i, d in enumerate(dilations):                         # :108
        p = "dilation_layer_stack.%d" % (4 * i)
        pg = "dilation_layer_stack.%d" % (4 * i + 1)
        pd = "dilation_layer_stack.%d" % (4 * i + 2)
        ps = "dilation_layer_stack.%d" % (4 * i + 3)
        f = F.conv1d(x, params[p + ".weight"], _b(params, p), dilation=d)     # :118
        g = F.conv1d(x, params[pg + ".weight"], _b(params, pg), dilation=d)   # :119
        z = torch.sigmoid(g) * torch.tanh(f)                                    # :120
        dense = F.conv1d(z, params[pd + ".weight"], _b(params, pd))             # :121
        x = dense + x[:, :, -dense.size(2):]                                    # :122-124
        s = F.conv1d(z[:, :, -out_w:], params[ps + ".weight"], _b(params, ps))  # :127-128
        skip_sum = s if skip_sum is None else skip_sum + s                      # :134 (sum)
        xs.append(x)
        zs.append(z)
    total = _relu(relu, "skip_sum", skip_sum)                                   # :135
    h1 = F.conv1d(total, params["post_process_1.weight"], _b(params, "post_process_1"))
    total = _relu(relu, "post_process_1", h1)                                   # :137
    total = F.conv1d(total, params["post_process_2.weight"], _b(params, "post_process_2"))
    if intermediates is not None:
        intermediates.update(x=xs, z=zs, skip_sum=skip_sum, post_process_1=h1, pre_softmax=total)
    return chunk_softmax(total, quantization_channels)                          # :142-145


def ce_on_probs(probs, target):
    """wavenet/train.py:146,179 — nn.CrossEntropyLoss applied to the model's *probabilities*
    (SURVEY Q1): mean_r[logsumexp(p_r) - p_r[y_r]]."""
    return F.cross_entropy(probs, target.view(-1))


def loss_and_grads(params, dilations, wave_sample, target, input_grad=False, **kw):
    """One forward + CE + backward (wavenet/train.py:178-181) on detached copies of ``params``.
    Returns (loss float tensor, probs, OrderedDict name -> grad); with ``input_grad`` the dict also holds the gradient with
    respect to ``wave_sample`` (what autograd gives through the causal nn.Conv1d, model.py:104) under the key "(input)"."""
    leaf = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in params.items())
    if input_grad:
        wave_sample = wave_sample.detach().clone().requires_grad_(True)
    probs = wavenet_forward(leaf, dilations, wave_sample, **kw)
    loss = ce_on_probs(probs, target)
    if input_grad:
        leaf["(input)"] = wave_sample
    grads = torch.autograd.grad(loss, list(leaf.values()), allow_unused=True)
    # the last block's dense conv never reaches the output (its x_N is unused): the reference
    # leaves that .grad None; reported here as exact zeros
    grads = [torch.zeros_like(v) if g is None else g for g, v in zip(grads, leaf.values())]
    return loss.detach(), probs.detach(), OrderedDict(zip(leaf.keys(), grads))


def predict_next_naive(params, dilations, wave_sample, **kw):
    """wavenet/model.py:148-165 — argmax over the LAST chunk-row of forward()."""
    probs = wavenet_forward(params, dilations, wave_sample, **kw)
    return torch.topk(probs[-1, :].view(-1), 1)[1]


# --------------------------------------------------------------------------------------------
# wavenet/fast_generate.py  (cached-queue incremental inference)
# --------------------------------------------------------------------------------------------
def _block_weights(params, i):
    return [(params["dilation_layer_stack.%d.weight" % (4 * i + k)],
             _b(params, "dilation_layer_stack.%d" % (4 * i + k))) for k in range(4)]


def _post(params, skip_sum, q):
    total = F.relu(skip_sum)
    total = F.conv1d(total, params["post_process_1.weight"], _b(params, "post_process_1"))
    total = F.relu(total)
    total = F.conv1d(total, params["post_process_2.weight"], _b(params, "post_process_2"))
    return chunk_softmax(total, q).view(-1)


def fast_predict_next(params, dilations, note, state_queue=None, filter_width=2,
                      quantization_channels=256, correct_queue=False, return_probs=False):
    """wavenet/fast_generate.py:13-141.

    First call (state_queue None, note (1,Q,rf)): full forward, snapshot queues (:29-65).
    Later calls (note (1,Q,1)): one column per layer from queue+note (:66-129).
    As written, each block's queue receives that block's OUTPUT (:128-129);
    ``correct_queue=True`` pushes the block INPUT instead (the fast-wavenet recurrence).
    Returns (LongTensor(1,), queue) [+ probs if return_probs].
    """
    q = quantization_channels
    res_ch = params["causal_layer.weight"].size(0)
    skip_sum = None
    if state_queue is None:
        rf = receptive_field(filter_width, dilations)
        assert note.size(2) == rf                                               # :30
        state_queue = OrderedDict()
        x = F.conv1d(note, params["causal_layer.weight"], _b(params, "causal_layer"))
        state_queue["causal_layer"] = note[:, :, -1].contiguous().view(1, q, 1)   # :33-38
        for i, d in enumerate(dilations):
            state_queue["block_%d" % (i + 1)] = x[:, :, -d:].contiguous().view(1, res_ch, d)
            (wf, bf), (wg, bg), (wd, bd), (ws, bs) = _block_weights(params, i)
            z = torch.sigmoid(F.conv1d(x, wg, bg, dilation=d)) * \
                torch.tanh(F.conv1d(x, wf, bf, dilation=d))
            dense = F.conv1d(z, wd, bd)
            x = dense + x[:, :, -dense.size(2):]
            s = F.conv1d(z[:, :, -1:], ws, bs)                                  # :63-64
            skip_sum = s if skip_sum is None else skip_sum + s
    else:
        assert note.size(2) == 1                                                # :67
        new_queue = OrderedDict()
        cstate = state_queue["causal_layer"]
        layer_in = torch.cat([cstate, note], 2)                                 # :73-75
        cur = F.conv1d(layer_in, params["causal_layer.weight"], _b(params, "causal_layer"))
        new_queue["causal_layer"] = torch.cat([cstate[:, :, 1:], note], 2)      # :99-104,116
        for i, d in enumerate(dilations):
            name = "block_%d" % (i + 1)
            st = state_queue[name]
            note_in = cur
            layer_in = torch.cat([st, note_in], 2)                              # (1,R,d+1)
            (wf, bf), (wg, bg), (wd, bd), (ws, bs) = _block_weights(params, i)
            z = torch.sigmoid(F.conv1d(layer_in, wg, bg, dilation=d)) * \
                torch.tanh(F.conv1d(layer_in, wf, bf, dilation=d))              # :84-86
            dense = F.conv1d(z, wd, bd)
            cur = dense + layer_in[:, :, -dense.size(2):]                       # :87-90
            s = F.conv1d(z[:, :, -1:], ws, bs)
            skip_sum = s if skip_sum is None else skip_sum + s
            pushed = note_in if correct_queue else cur                          # :128-129 (Q5)
            new_queue[name] = torch.cat([st[:, :, 1:], pushed], 2)
        state_queue = new_queue
    probs = _post(params, skip_sum, q)                                          # :130-139
    pred = torch.topk(probs, 1)[1]                                              # :140
    if return_probs:
        return pred, state_queue, probs
    return pred, state_queue


# --------------------------------------------------------------------------------------------
# wavenet_autoencoder/model1.py
# --------------------------------------------------------------------------------------------
def condition(x, enc):
    """wavenet_autoencoder/model1.py:227-247 (``_conditon``).
    len(x) % len(enc) == 0 -> nearest-neighbour stretch  enc[t // (Lx/Le)];
    otherwise           -> periodic tile               enc[t %  Le]."""
    mb, ch, le = enc.shape
    lx = x.size(2)
    if lx % le == 0:
        return (x.reshape(mb, ch, le, -1) + enc.reshape(mb, ch, le, 1)).reshape(mb, ch, lx)
    idx = torch.arange(lx) % le
    return x + enc[:, :, idx]


def autoencoder_encode(params, dilations, wave_sample, pool, relu=None, intermediates=None):
    """model1.py:137-156 — relu -> dilated conv -> relu -> 1x1, residual on the tail slice;
    then bottleneck 1x1 and AvgPool1d(pool).  ``relu``: see _relu (names 'en_x<i>', 'en_h<i>');
    ``intermediates`` receives those pre-activations under the same names."""
    x = F.conv1d(wave_sample, params["en_causal_layer.weight"], _b(params, "en_causal_layer"))
    for i, d in enumerate(dilations):
        h = F.conv1d(_relu(relu, "en_x%d" % i, x), params["en_dilation_layer_stack.%d.weight" % i],
                     _b(params, "en_dilation_layer_stack.%d" % i), dilation=d)
        if intermediates is not None:
            intermediates["en_x%d" % i] = x
            intermediates["en_h%d" % i] = h
        h = F.conv1d(_relu(relu, "en_h%d" % i, h), params["en_dense_layer_stack.%d.weight" % i],
                     _b(params, "en_dense_layer_stack.%d" % i))
        x = h + x[:, :, -h.size(2):]
    x = F.conv1d(x, params["bottleneck_layer.weight"], _b(params, "bottleneck_layer"))
    return F.avg_pool1d(x, pool)


def autoencoder_decode(params, dilations, wave_sample, enc, out_w, cond, q=256, relu=None, intermediates=None):
    """model1.py:158-225.  ``cond`` is the list of N+1 (weight (C,Bw,1), bias (C,)) pairs the
    reference draws afresh inside every forward (unregistered nn.Conv1d): N per-layer
    (2*Dd channels) + 1 final (Sd channels).  gate = first half of filter_gate channels,
    filter = second half (:188-190)."""
    x = F.conv1d(wave_sample, params["de_causal_layer.weight"], _b(params, "de_causal_layer"))
    skip_sum = None
    for i, d in enumerate(dilations):
        p = "de_dilation_layer_stack.%d" % (3 * i)
        pd = "de_dilation_layer_stack.%d" % (3 * i + 1)
        ps = "de_dilation_layer_stack.%d" % (3 * i + 2)
        h = F.conv1d(x, params[p + ".weight"], _b(params, p), dilation=d)       # :175
        en = F.conv1d(enc, cond[i][0], cond[i][1])                              # :178-179
        h = condition(h, en)                                                    # :183
        half = h.size(1) // 2
        xg, xf = h[:, :-half], h[:, -half:]                                     # :188-190
        z = torch.tanh(xf) * torch.sigmoid(xg)                                  # :192
        r = F.conv1d(z, params[pd + ".weight"], _b(params, pd))
        x = x[:, :, -r.size(2):] + r                                            # :196-201
        s = F.conv1d(z[:, :, -out_w:], params[ps + ".weight"], _b(params, ps))  # :203-205
        skip_sum = s if skip_sum is None else skip_sum + s
    r = F.conv1d(_relu(relu, "de_skip", skip_sum), params["connection_1.weight"], _b(params, "connection_1"))
    en = F.conv1d(enc, cond[-1][0], cond[-1][1])                                # :216-217
    c1 = condition(r, en)
    if intermediates is not None:
        intermediates.update(de_skip=skip_sum, de_conn=c1)
    r = _relu(relu, "de_conn", c1)                                              # :219-220
    r = F.conv1d(r, params["connection_2.weight"], _b(params, "connection_2"))
    return chunk_softmax(r, q)                                                  # :222-224


def draw_conditioning(n_layers, bottleneck, de_dilation_channel, de_skip_channel):
    """Draw the N+1 per-forward conditioning convs from the GLOBAL torch CPU RNG in the same
    order and with the same initialiser as ``nn.Conv1d(...)`` at model1.py:178,216 does."""
    cond = []
    for i in range(n_layers + 1):
        c = torch.nn.Conv1d(bottleneck,
                            2 * de_dilation_channel if i < n_layers else de_skip_channel, 1)
        cond.append((c.weight.detach(), c.bias.detach()))
    return cond


def autoencoder_forward(params, dilations, wave_sample, pool, cond, filter_width=2, q=256, relu=None,
                        intermediates=None):
    """model1.py:256-268."""
    rf = receptive_field(filter_width, dilations)
    out_w = wave_sample.size(2) - rf + 1
    enc = autoencoder_encode(params, dilations, wave_sample, pool, relu, intermediates)
    return autoencoder_decode(params, dilations, wave_sample, enc, out_w, cond, q, relu, intermediates), enc
