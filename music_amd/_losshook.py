"""`nn.CrossEntropyLoss` on a module's output as ONE pass over the pre-softmax buffer.

The reference's training loops apply `nn.CrossEntropyLoss()` to the model's PROBABILITIES (wavenet/train.py:146,179,
wavenet_autoencoder/train.py).  On this path torch's criterion is five kernels over 106 MB of probabilities (0.42 ms of a
4.7 ms step at 8 x 16000) behind a chunk softmax that then also runs as separate forward and backward kernels, where the engines have
the whole thing as one kernel (wn_chunk_softmax256_ce: loss, and d loss / d pre-softmax).  The caller's code cannot change, but the
module's OUTPUT can carry the knowledge: `forward` returns a `torch.Tensor` subclass whose `__torch_function__` lets every operation
through to torch except `F.cross_entropy` with default arguments on the unmodified output of the latest forward.

Autograd stays exact.  The fused node hands `probs` a gradient of zeros that occupies four bytes (expanded, stride 0); its own
contribution - d loss / d pre-softmax for an upstream gradient of 1, already in the workspace - is picked up by the module's backward
through the hook and scaled by the upstream gradient (the backward is linear in it).  If the user's loss uses the probabilities
elsewhere too, autograd adds that dense gradient to the zeros; the module's backward then sees a tensor that is not the token, runs
its softmax backward on it and adds the fused part.  One documented difference from torch: a target equal to `ignore_index` (-100)
makes the fused loss NaN instead of being skipped."""
import torch
import torch.nn.functional as F

try:
    from . import _lib
except ImportError:
    from music_amd import _lib


class LossHook(object):
    """What `nn.CrossEntropyLoss` on this forward's output needs to run as ONE pass over the pre-softmax buffer (the engine's
    fused chunk softmax + cross entropy + both backward steps, wn_chunk_softmax256_ce) instead of torch's five kernels over the
    probabilities (0.42 ms of a 4.7 ms step at 8 x 16000)."""
    __slots__ = ("eng", "ws", "gen", "version", "fused", "dloss")

    def __init__(self, eng, ws, gen):
        self.eng, self.ws, self.gen = eng, ws, gen
        self.version, self.fused, self.dloss = None, False, None


_ZERO = {}


def _zero_token(like):
    """A gradient of zeros for `probs` that occupies four bytes (expanded with stride 0): what the fused loss hands autograd.  Its own
    contribution travels through the hook; any OTHER use of the probabilities in the user's loss adds its dense gradient to these
    zeros, and the result is then no longer this buffer."""
    key = (like.device, like.dtype)
    if key not in _ZERO:
        _ZERO[key] = torch.zeros(1, dtype=like.dtype, device=like.device)
    return _ZERO[key]


class _FusedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, target, hook):
        eng, ws = hook.eng, hook.ws
        n = ws["B"] * ws["W"]
        bw = eng._bwd_workspace(ws) if ctx.needs_input_grad[0] else None
        if "loss_part" not in ws:
            ws["loss_part"] = torch.zeros(_lib.CE_NUM_PARTIALS, dtype=torch.float32, device=eng.device)
        _lib.call("wn_chunk_softmax256_ce", _lib.ptr(ws["O"]), _lib.ptr(target), None, _lib.ptr(bw["dO"]) if bw else None,
                  _lib.ptr(ws["loss_part"]), n, 1.0 / n, _lib.stream())
        hook.fused = bw is not None
        ctx.hook, ctx.shape = hook, probs.shape
        return ws["loss_part"].sum()

    @staticmethod
    def backward(ctx, dloss):
        ctx.hook.dloss = dloss
        return _zero_token(dloss).expand(ctx.shape), None, None


class Probs(torch.Tensor):
    """The module's output while a backward may follow: an ordinary tensor for every operation but one - the reference's own
    `nn.CrossEntropyLoss()(net(x), target)` (wavenet/train.py:146,179: mean reduction, no weights, no smoothing) runs fused."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is F.cross_entropy:
            out = _fused_cross_entropy(*args, **kwargs)
            if out is not NotImplemented:
                return out
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def _fused_cross_entropy(input, target, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction="mean",
                         label_smoothing=0.0):
    hook = getattr(input, "_wn_hook", None)
    if (hook is None or hook.fused or weight is not None or size_average is not None or reduce is not None or
            reduction != "mean" or label_smoothing != 0.0 or ignore_index != -100 or type(target) is not torch.Tensor):
        return NotImplemented
    ws = hook.ws
    if (ws.get("gen") != hook.gen or input._version != hook.version or target.dtype != torch.int64 or target.device != input.device or
            target.dim() != 1 or target.numel() != ws["B"] * ws["W"]):
        return NotImplemented
    with torch._C.DisableTorchFunctionSubclass():
        return _FusedCE.apply(input, target.contiguous(), hook)


def make(eng, ws, grad_on):
    """Called in the module's autograd Function forward: a hook for this forward, or None (inference, or an engine without the fused
    kernel: the general plans)."""
    return LossHook(eng, ws, ws["gen"]) if (grad_on and getattr(eng, "fused_loss_ok", False)) else None


def wrap(out, hook):
    """The module's output as the intercepting subclass (an alias of `out` under autograd)."""
    if hook is None or not out.requires_grad:
        return out
    out = out.as_subclass(Probs)
    hook.version = out._version
    out._wn_hook = hook
    return out


def backward(hook, eng, ws, dprobs):
    """Called first thing in the module's autograd Function backward.  True: the gradients are in eng.flat_grad (the loss ran fused);
    False: nothing was done, run the ordinary backward from `dprobs`."""
    if hook is None or not hook.fused or hook.dloss is None:
        return False
    tok = _zero_token(dprobs)
    if dprobs.data_ptr() == tok.data_ptr() and not any(dprobs.stride()):
        eng.backward_from_dlogits(ws)                       # d loss / d pre-softmax is in the workspace (for an upstream gradient of 1)
        eng.flat_grad.mul_(hook.dloss)
    else:                                                   # the probabilities are used elsewhere in the loss too
        n = ws["B"] * ws["W"]
        d_o = eng._bwd_workspace(ws)["dO"][:n * eng.Q]
        keep = d_o.clone()                                  # (a second backward through a retained graph finds it unchanged)
        extra = torch.empty(n * eng.Q, dtype=torch.float32, device=eng.device)
        _lib.call("wn_chunk_softmax256_bwd", _lib.ptr(ws["probs"]), _lib.ptr(dprobs.contiguous()), _lib.ptr(extra), n, _lib.stream())
        d_o.mul_(hook.dloss).add_(extra)
        eng.backward_from_dlogits(ws)
        d_o.copy_(keep)
    hook.dloss = None
    return True

