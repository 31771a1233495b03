"""Drop-in counterpart of the reference's ``wavenet/model.py`` backed by the MI355X HIP kernels.

Same constructor kwargs, attributes, registered ``nn.Conv1d`` sub-modules (hence the same
``state_dict`` keys / shapes and the attribute access ``fast_generate.py`` relies on), same
``forward`` contract (wavenet/model.py:86-145): input ``(B, Q, T >= receptive_field)`` float ->
probabilities ``(B*(T-rf+1), Q)`` with the reference's CHUNK softmax semantics, and the same
``ValueError("wave sample not long enough")``.

The arithmetic runs only through libwavenet_hip.so (music_amd/engine.py).  There is no CPU
implementation: calling ``forward`` without a ROCm device raises.
"""
import torch
import torch.nn as nn

try:
    from . import _losshook
    from .engine import WaveNetEngine, WorkspaceHold
    from .engine_generic import GenericWaveNetEngine
except ImportError:                      # imported as a bare module (`from model import wavenet`)
    from music_amd import _losshook
    from music_amd.engine import WaveNetEngine, WorkspaceHold
    from music_amd.engine_generic import GenericWaveNetEngine

_Probs = _losshook.Probs                 # the type of the module's output while a backward may follow


class _WaveNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, grad_on, wave_sample, *params):
        eng = net._engine_for(wave_sample.device)
        x = wave_sample.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        else:
            tag = getattr(wave_sample, "_wn_codes", None)        # one-hot built from codes (engine.forward_logits)
            if tag is not None and wave_sample._version == tag[2]:
                x._wn_codes = (tag[0], tag[1], x._version, tag[3])
        probs, ws = eng.forward(x)
        ctx.eng, ctx.ws, ctx.gen = eng, ws, ws["gen"]
        ctx.loss_hook = net._last_hook = _losshook.make(eng, ws, grad_on)
        # a forward that some backward may follow keeps its workspace: the next forward of this shape gets another one (several
        # micro-batches in flight, as the reference's autograd allows).  `grad_on` is torch.is_grad_enabled() as the caller saw
        # it (inside Function.forward it is always off, and needs_input_grad is True for parameters even under no_grad): an
        # inference forward holds nothing
        ctx.hold = WorkspaceHold(ws) if (grad_on and any(ctx.needs_input_grad)) else None
        # a fresh alias goes out: the workspace keeps `probs` for the backward, and the tensor autograd hangs this node on
        # must not be the one the workspace holds (workspace -> output -> node -> hold -> workspace would never die)
        return probs.detach()

    @staticmethod
    def backward(ctx, dprobs):
        eng, ws = ctx.eng, ctx.ws
        if ws.get("gen") != ctx.gen:
            raise RuntimeError("music_amd.wavenet: the activations of this forward were overwritten by a later "
                               "forward of the same module before backward() ran")
        if not _losshook.backward(ctx.loss_hook, eng, ws, dprobs):          # (the loss ran fused: see _losshook.py)
            eng.backward(ws, dprobs)
        if ctx.hold is not None:
            ctx.hold.release()           # (retain_graph + a second backward still works until the next forward reuses it)
        g = eng.flat_grad.clone()
        grads = []
        for name in eng.param_names:
            o, shp = eng.spec.off[name], eng.spec.shape[name]
            n = 1
            for s in shp:
                n *= s
            grads.append(g[o:o + n].view(shp))
        # the input's own gradient, when autograd asks for it (the reference's causal nn.Conv1d provides it, wavenet/model.py:104)
        din = eng.input_grad(ws) if ctx.needs_input_grad[2] else None
        return (None, None, din) + tuple(grads)


class wavenet(nn.Module):

    def __init__(self, filter_width, dilations, dilation_channels, residual_channels, skip_channels,
                 quantization_channels, use_bias):
        super(wavenet, self).__init__()
        self.filter_width = filter_width
        self.dilations = dilations
        self.dilation_channels = dilation_channels
        self.residual_channels = residual_channels
        self.skip_channels = skip_channels
        self.quantization_channels = quantization_channels
        self.use_bias = use_bias
        self.receptive_field = self.calc_receptive_field()
        # construction order (and therefore the default-init RNG stream) follows
        # wavenet/model.py:38-41: causal, then per dilation filter/gate/dense/skip, then post-process
        self.causal_layer = nn.Conv1d(quantization_channels, residual_channels, filter_width, bias=use_bias)
        self.dilation_layer_stack = nn.ModuleList()
        for d in dilations:
            self.dilation_layer_stack.extend([
                nn.Conv1d(residual_channels, dilation_channels, filter_width, dilation=d, bias=use_bias),
                nn.Conv1d(residual_channels, dilation_channels, filter_width, dilation=d, bias=use_bias),
                nn.Conv1d(dilation_channels, residual_channels, 1, bias=use_bias),
                nn.Conv1d(dilation_channels, skip_channels, 1, bias=use_bias)])
        self.post_process_1 = nn.Conv1d(skip_channels, skip_channels, 1, bias=use_bias)
        self.post_process_2 = nn.Conv1d(skip_channels, quantization_channels, 1, bias=use_bias)
        self.softmax = nn.Softmax(dim=1)     # the reference's nn.Softmax() resolves to dim 1 on 2-D input
        self._engine = None
        self.precision = ("f16x3", "bf16x3")     # (forward, backward) arithmetic of the MFMA products
        # nn.CrossEntropyLoss()(net(x), target) with its default arguments runs as the engine's fused softmax + CE pass (the output
        # is a Tensor subclass that intercepts exactly that call; everything else sees an ordinary tensor).  False: torch's own
        self.fuse_loss = True
        self._last_hook = None

    def __getstate__(self):
        # copy.deepcopy / pickle / torch.save(module): the engine (HIP streams, workspaces, ctypes plans) stays behind and is rebuilt
        # on the copy's first forward; the parameters travel as tensors
        state = self.__dict__.copy()
        state["_engine"] = None
        state["_last_hook"] = None
        return state

    def calc_receptive_field(self):
        return (self.filter_width - 1) * (sum(self.dilations) + 1) + 1

    # ---- engine plumbing ------------------------------------------------------------------
    def _named_ref_params(self):
        """Parameters in the reference's state_dict order."""
        return list(self.named_parameters())

    def _engine_for(self, device):
        if device.type != "cuda":
            raise RuntimeError("music_amd.wavenet runs on an MI355X (ROCm) device only; there is no CPU path. "
                               "Move the module and its input to 'cuda'.")
        eng = self._engine
        named = self._named_ref_params()
        if eng is not None and eng.device == device and eng.mode_names == tuple(self.precision):
            # parameters must still alias the flat buffer (e.g. net.cpu().cuda() breaks that)
            p0 = named[0][1]
            if p0.data_ptr() == eng.flat.data_ptr() + 4 * eng.spec.off[named[0][0]]:
                return eng
        if any(p.device != device for _, p in named):
            raise RuntimeError("music_amd.wavenet: parameters and input are on different devices")
        if any(p.dtype != torch.float32 for _, p in named):
            # (.half() / .double() modules: the kernels keep float32 parameters and form fp32-grade products from 16-bit pieces,
            # DESIGN.md section 5; re-typing the module's parameters behind the caller's back would be worse than saying so)
            raise TypeError("music_amd.wavenet: parameters must be float32 (got %s)" % next(p.dtype for _, p in named if p.dtype != torch.float32))
        # the specialised kernels cover filter_width 2, 256 quantisation channels and up to 64 residual / dilation channels
        # (everything the reference ships and BASELINE.json names); any other constructor argument takes the general plan
        fast = (self.filter_width == 2 and self.quantization_channels == 256 and
                max(self.residual_channels, self.dilation_channels) <= 64)
        eng = (WaveNetEngine if fast else GenericWaveNetEngine)(
            self.dilations, self.residual_channels, self.dilation_channels, self.skip_channels,
            self.quantization_channels, self.filter_width, self.use_bias,
            mode_fwd=self.precision[0], mode_bwd=self.precision[1], device=device)
        eng.mode_names = tuple(self.precision)
        assert [n for n, _ in named] == eng.param_names, "parameter order differs from the reference layout"
        with torch.no_grad():
            for name, p in named:
                view = eng.param_view(name)
                view.copy_(p.data)
                p.data = view                     # the module's parameters now ARE the flat buffer
        self._engine = eng
        return eng

    def forward(self, wave_sample):
        batch_size, original_channels, seq_len = wave_sample.size()
        output_width = seq_len - self.receptive_field + 1
        if output_width <= 0:
            raise ValueError("wave sample not long enough")
        params = [p for _, p in self._named_ref_params()]
        self._last_hook = None
        out = _WaveNetFunction.apply(self, torch.is_grad_enabled(), wave_sample, *params)
        hook, self._last_hook = self._last_hook, None
        return _losshook.wrap(out, hook) if self.fuse_loss else out


def predict_next(model, wave_var, quantization_channels=256):
    """wavenet/model.py:148-165 — argmax over the last chunk-row of the model output."""
    raw_out = model(wave_var)
    out = raw_out.view(-1, quantization_channels)
    last = out[-1, :].view(-1)
    _, predict = torch.topk(last, 1)
    return predict
