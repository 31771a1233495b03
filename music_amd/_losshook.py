"""`nn.CrossEntropyLoss` on a module's output as ONE pass over the pre-softmax buffer.

The reference's training loops apply `nn.CrossEntropyLoss()` to the model's PROBABILITIES (wavenet/train.py:146,179,
wavenet_autoencoder/train.py).  On this path torch's criterion is five kernels over 106 MB of probabilities (0.42 ms of a
4.7 ms step at 8 x 16000) behind a chunk softmax that then also runs as separate forward and backward kernels, where the engines have
the whole thing as one kernel (wn_chunk_softmax256_ce: loss, and d loss / d pre-softmax).  The caller's code cannot change, but the
module's OUTPUT can carry the knowledge: `forward` returns a `torch.Tensor` subclass whose `__torch_function__` lets every operation
through to torch except `F.cross_entropy` with default arguments on the unmodified output of the latest forward.

The PARAMETER gradients autograd delivers are exact.  The fused node hands `probs` a gradient of zeros that occupies four bytes
(expanded, stride 0); its own contribution - d loss / d pre-softmax for an upstream gradient of 1, already in the workspace - is picked
up by the module's backward through the hook and scaled by the upstream gradient (the backward is linear in it).  If the user's loss
uses the probabilities elsewhere too, autograd adds that dense gradient to the zeros; the module's backward then sees a tensor that is
not the token, runs its softmax backward on it and adds the fused part.  Any order of backward passes over the same forward is
handled: a backward of ANOTHER loss on the output that runs before the fused loss's own (retain_graph) overwrites the workspace's
d loss / d pre-softmax, which the hook notices (`do_valid`) and re-forms from the kept target before it is used; an upstream gradient
that no module backward consumed (`torch.autograd.grad(loss, out)`) is dropped when that backward pass ends, not carried into the next.

Limits next to torch's own criterion (`net.fuse_loss = False` gives torch's behaviour in every one of them):
  * what autograd REPORTS for the probabilities themselves is the zero token: `out.retain_grad()`, tensor hooks on `out` and
    `torch.autograd.grad(loss, out)` see zeros, not d loss / d probs (parameter gradients are unaffected);
  * a target equal to `ignore_index` (-100) makes the fused loss NaN instead of being skipped, and a target outside [0, 256) makes the
    loss and that row's gradient NaN where torch raises a device assert (the kernel never indexes with it: wn_elem.hip);
  * `create_graph=True` (double backward) is not supported through the fused node."""
import torch
import torch.nn.functional as F

try:
    from . import _lib
except ImportError:
    from music_amd import _lib


def _private_api_present():
    """The interception rests on three torch internals (a Tensor-subclass escape hatch and the autograd engine's end-of-pass callback).  If a
    torch build lacks any of them the module simply does not fuse its loss: `nn.CrossEntropyLoss` then runs as torch's own kernels over the
    probabilities - slower (0.4 ms of a step), same numbers - instead of failing inside a backward."""
    eng = getattr(getattr(torch.autograd, "Variable", None), "_execution_engine", None)
    return (hasattr(torch._C, "DisableTorchFunctionSubclass") and callable(getattr(eng, "queue_callback", None)) and
            hasattr(torch.Tensor, "as_subclass"))


AVAILABLE = _private_api_present()


class LossHook(object):
    """What `nn.CrossEntropyLoss` on this forward's output needs to run as ONE pass over the pre-softmax buffer (the engine's
    fused chunk softmax + cross entropy + both backward steps, wn_chunk_softmax256_ce) instead of torch's five kernels over the
    probabilities (0.42 ms of a 4.7 ms step at 8 x 16000)."""
    __slots__ = ("eng", "ws", "gen", "version", "fused", "dloss", "target", "do_valid")

    def __init__(self, eng, ws, gen):
        self.eng, self.ws, self.gen = eng, ws, gen
        self.version, self.fused, self.dloss = None, False, None
        # target: what the fused loss was taken against (kept so that d loss / d pre-softmax can be formed again);
        # do_valid: the workspace's dO still holds it (an ordinary backward on the same forward overwrites dO)
        self.target, self.do_valid = None, False


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
        hook.fused = hook.do_valid = bw is not None
        hook.target = target if bw is not None else None
        ctx.hook, ctx.shape = hook, probs.shape
        return ws["loss_part"].sum()

    @staticmethod
    def backward(ctx, dloss):
        hook = ctx.hook
        hook.dloss = dloss
        # the module's backward of THIS pass consumes it; if none runs (torch.autograd.grad(loss, out)) it must not survive
        # into a later, unrelated backward: dropped when the pass ends
        def _drop(h=hook, d=dloss):
            if h.dloss is d:
                h.dloss = None
        torch.autograd.Variable._execution_engine.queue_callback(_drop)
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
    return LossHook(eng, ws, ws["gen"]) if (AVAILABLE and grad_on and getattr(eng, "fused_loss_ok", False)) else None


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
    if hook is None or not hook.fused:
        return False
    if hook.dloss is None:
        # an ordinary backward (another loss on the same output) while the fused loss is pending: eng.backward() is about to
        # overwrite the workspace's dO, which the fused loss's own backward will need
        hook.do_valid = False
        return False
    if not hook.do_valid:
        if ws.get("gen") != hook.gen:
            raise RuntimeError("music_amd: the activations of this forward were overwritten before the fused loss's backward")
        n = ws["B"] * ws["W"]
        _lib.call("wn_chunk_softmax256_ce", _lib.ptr(ws["O"]), _lib.ptr(hook.target), None, _lib.ptr(eng._bwd_workspace(ws)["dO"]),
                  None, n, 1.0 / n, _lib.stream())
        hook.do_valid = True
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

