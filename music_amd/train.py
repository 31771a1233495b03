"""Counterpart of the reference's ``wavenet/train.py`` for the MI355X path.

Same function names, JSON schema (./params/{train,wavenet,dataset}_params.json read relative to the
CWD), step order ``zero_grad -> forward -> CrossEntropyLoss(probs, target.view(-1)) -> backward ->
step`` (wavenet/train.py:171-182), batch counter and its recovery from the last log line
(:160-166,186), log-line formats (:189-191,217-218), checkpoint naming / rotation / ``module.``
stripping (:45-73,198-216).  The reference file itself cannot run on Python >= 3.7
(``.cuda(async=True)``), so this module is the drop-in.

Differences, all additive:
  * ``nn.DataParallel`` (one process, GPU-0 bottleneck) is replaced by one process per GPU
    (``torchrun --nproc-per-node N train.py``) with ONE flat RCCL all-reduce per step
    (music_amd/dist.py); each rank builds only its own shard of every global batch.
  * optional keys in train_params.json: ``"seed"`` (int), ``"fused_step"`` (bool: Adam only — the
    whole step runs as forward+CE+backward+flat-Adam kernels without autograd), ``"save_optimizer_state"``
    (bool, SURVEY 8f4: the reference drops the optimizer state at every checkpoint, so a resumed Adam /
    momentum run restarts cold; with this key ``wavenet{N}.opt`` is written next to ``wavenet{N}.model``
    and read back when that model is restored — the .model file itself stays the reference's format).
"""
from collections import OrderedDict
from functools import cmp_to_key
import glob
import json
import os

import torch
import torch.nn as nn
import torch.optim as optim

try:
    from . import dist as wdist
    from .faster_audio_data import audio_data_loader
    from .model import wavenet
except ImportError:                                  # run as a script / bare modules from the CWD
    from music_amd import dist as wdist
    from music_amd.faster_audio_data import audio_data_loader
    from music_amd.model import wavenet


def get_params(json_dir):
    with open(json_dir, 'r') as f:
        return json.load(f)


def get_arguments():
    return (get_params('./params/train_params.json'),
            get_params('./params/wavenet_params.json'),
            get_params('./params/dataset_params.json'))


class FlatAdam(optim.Adam):
    """``torch.optim.Adam`` whose ``step()`` is ONE ``wn_adam_flat`` launch when it can be.

    The module's parameters are views of one flat buffer (music_amd/engine.py) and the gradients autograd hands to them
    are views of one flat buffer too (``_WaveNetFunction.backward``), so the 123 per-tensor updates of the reference's
    ``optim.Adam(model.parameters(), lr)`` (wavenet/train.py:39-42) are one elementwise pass.  The optimizer state is
    torch's own - per parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` - with the two moment tensors being VIEWS of two flat
    buffers: ``state_dict()`` / ``load_state_dict()`` interchange with ``torch.optim.Adam``, and any step the fast path
    does not cover (parameters not on the flat buffer yet, gradients that are not one flat tensor, weight decay /
    amsgrad / maximize / closures, several parameter groups) falls through to ``torch.optim.Adam.step`` on the same state."""

    def __init__(self, model, lr):
        super().__init__(model.parameters(), lr=lr)
        self._model = model
        self._flat = None            # (engine, m, v, steps): flat moments of that engine's flat parameter buffer

    def _fast_state(self):
        eng = getattr(self._model, "_engine", None)
        if eng is None or getattr(eng, "flat", None) is None or len(self.param_groups) != 1:
            return None
        g = self.param_groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            return None
        params = g["params"]
        named = [p for _, p in self._model.named_parameters()]
        if len(params) != len(named) or any(a is not b for a, b in zip(params, named)):
            return None
        base = eng.flat.data_ptr()
        if any(p.data_ptr() != base + 4 * eng.spec.off[n] for n, p in zip(eng.param_names, params)):
            return None                                   # (the module was moved: its parameters left the flat buffer)
        if self._flat is not None and self._flat[0] is eng:
            return self._flat
        # adopt whatever per-parameter state exists (a loaded state_dict, earlier torch steps) into flat moment buffers
        m, v = torch.zeros_like(eng.flat), torch.zeros_like(eng.flat)
        steps = []
        for n, p in zip(eng.param_names, params):
            o, k = eng.spec.off[n], p.numel()
            st = self.state[p]
            if "exp_avg" in st:
                m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            step = st.get("step")
            step = torch.zeros((), dtype=torch.float32) if step is None else torch.as_tensor(step, dtype=torch.float32).detach().cpu().reshape(())
            st["step"], st["exp_avg"], st["exp_avg_sq"] = step, m[o:o + k].view(p.shape), v[o:o + k].view(p.shape)
            steps.append(step)
        if any(float(s) != float(steps[0]) for s in steps):
            return None                                   # parameters at different step counts: torch's per-tensor path
        self._flat = (eng, m, v, steps)
        return self._flat

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat = None                                 # re-adopt the loaded tensors at the next step

    @torch.no_grad()
    def step(self, closure=None):
        fast = self._fast_state() if closure is None else None
        if fast is not None:
            eng, m, v, steps = fast
            params = self.param_groups[0]["params"]
            g0 = params[0].grad
            ok = g0 is not None and g0.dtype == torch.float32
            if ok:
                gbase = g0.data_ptr() - 4 * eng.spec.off[eng.param_names[0]]
                ok = all(p.grad is not None and p.grad.is_contiguous() and p.grad.data_ptr() == gbase + 4 * eng.spec.off[n]
                         for n, p in zip(eng.param_names, params))
            if ok:
                from music_amd import _lib
                grp = self.param_groups[0]
                torch._foreach_add_(steps, 1.0)
                t = float(steps[0])
                b1, b2 = grp["betas"]
                _lib.call("wn_adam_flat", eng.flat.data_ptr(), gbase, m.data_ptr(), v.data_ptr(), eng.spec.total,
                          float(grp["lr"]), b1, b2, grp["eps"], 1.0 - b1 ** t, 1.0 - b2 ** t, 1.0, _lib.stream())
                return None
        return super().step(closure)


def _flat_layout(opt, model, plain_group):
    """(engine, params, flat gradient base pointer) when `opt` can take its step as ONE launch on the engine's flat buffers: one
    parameter group holding exactly the module's parameters in order, every parameter a view of the flat parameter buffer, every
    gradient a contiguous fp32 view of ONE flat gradient buffer at the same offsets (what the module's backward hands autograd),
    and `plain_group(group)` true (no option the kernel does not implement).  None otherwise: torch's own step runs."""
    eng = getattr(model, "_engine", None)
    if eng is None or getattr(eng, "flat", None) is None or len(opt.param_groups) != 1:
        return None
    grp = opt.param_groups[0]
    if not plain_group(grp):
        return None
    params = grp["params"]
    named = [p for _, p in model.named_parameters()]
    if len(params) != len(named) or any(a is not b for a, b in zip(params, named)):
        return None
    base = eng.flat.data_ptr()
    if any(p.data_ptr() != base + 4 * eng.spec.off[n] for n, p in zip(eng.param_names, params)):
        return None                                       # (the module was moved: its parameters left the flat buffer)
    g0 = params[0].grad
    if g0 is None or g0.dtype != torch.float32:
        return None
    gbase = g0.data_ptr() - 4 * eng.spec.off[eng.param_names[0]]
    if not all(p.grad is not None and p.grad.is_contiguous() and p.grad.data_ptr() == gbase + 4 * eng.spec.off[n]
               for n, p in zip(eng.param_names, params)):
        return None
    return eng, params, gbase


def _flat_views(opt, eng, params, key, cache):
    """One flat buffer for the per-parameter state tensors `key`, the state entries being VIEWS of it (so `state_dict()` /
    `load_state_dict()` interchange with torch's optimizer).  Existing per-parameter tensors (a loaded state_dict, earlier torch
    steps) are adopted.  Returns (flat buffer, how many parameters had the entry before)."""
    hit = cache.get(key)
    if hit is not None and hit[0] is eng and all(opt.state[p].get(key) is v for p, v in zip(params, hit[2])):
        return hit[1], len(params)
    flat = torch.zeros_like(eng.flat)
    had, views = 0, []
    for n, p in zip(eng.param_names, params):
        o, k = eng.spec.off[n], p.numel()
        st = opt.state[p]
        old = st.get(key)
        view = flat[o:o + k].view(p.shape)
        if old is not None:
            view.copy_(old)
            had += 1
        st[key] = view
        views.append(view)
    cache[key] = (eng, flat, views)
    return flat, had


class FlatSGD(optim.SGD):
    """``torch.optim.SGD(model.parameters(), lr, momentum)`` (wavenet/train.py:28-31) whose ``step()`` is ONE ``wn_sgd_flat``
    launch when it can be (see FlatAdam): same state (``momentum_buffer`` per parameter, views of one flat buffer), same
    ``state_dict``; anything the kernel does not cover (dampening, Nesterov, weight decay, maximize, closures, parameters or
    gradients off the flat buffers, parameters with and without a momentum buffer mixed) falls through to torch's own step."""

    def __init__(self, model, lr, momentum):
        super().__init__(model.parameters(), lr=lr, momentum=momentum)
        self._model, self._cache = model, {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._cache = {}

    @torch.no_grad()
    def step(self, closure=None):
        lay = None
        if closure is None:
            lay = _flat_layout(self, self._model, lambda g: (g.get("dampening", 0) == 0 and not g.get("nesterov") and g.get("weight_decay", 0) == 0
                                                           and not g.get("maximize") and not g.get("differentiable")))
        if lay is not None:
            eng, params, gbase = lay
            grp = self.param_groups[0]
            mom = float(grp["momentum"])
            buf, first = None, 0
            if mom != 0.0:
                have = sum(1 for p in params if self.state[p].get("momentum_buffer") is not None)
                if have not in (0, len(params)):
                    return super().step(closure)
                flat, _ = _flat_views(self, eng, params, "momentum_buffer", self._cache)
                buf, first = flat.data_ptr(), 1 if have == 0 else 0
            from music_amd import _lib
            _lib.call("wn_sgd_flat", eng.flat.data_ptr(), gbase, buf, eng.spec.total, float(grp["lr"]), mom, 1.0, first, _lib.stream())
            return None
        return super().step(closure)


class FlatRMSprop(optim.RMSprop):
    """``torch.optim.RMSprop(model.parameters(), lr, momentum=momentum)`` (wavenet/train.py:32-37) whose ``step()`` is ONE
    ``wn_rmsprop_flat`` launch when it can be (see FlatAdam): torch's own state (``step``, ``square_avg``, ``momentum_buffer``; the
    tensors are views of flat buffers), same ``state_dict``; centered / weight decay / maximize / closures fall through to torch."""

    def __init__(self, model, lr, momentum):
        super().__init__(model.parameters(), lr=lr, momentum=momentum)
        self._model, self._cache = model, {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._cache = {}

    @torch.no_grad()
    def step(self, closure=None):
        lay = None
        if closure is None:
            lay = _flat_layout(self, self._model, lambda g: (not g.get("centered") and g.get("weight_decay", 0) == 0 and not g.get("maximize")
                                                           and not g.get("capturable") and not g.get("differentiable")))
        if lay is not None:
            eng, params, gbase = lay
            grp = self.param_groups[0]
            mom = float(grp["momentum"])
            sq, _ = _flat_views(self, eng, params, "square_avg", self._cache)
            buf = None
            if mom > 0.0:
                buf = _flat_views(self, eng, params, "momentum_buffer", self._cache)[0].data_ptr()
            steps = self._cache.get("step")                   # torch's step counters (a 0-d tensor per parameter), bumped in one call
            if steps is None or any(self.state[p].get("step") is not t for p, t in zip(params, steps)):
                steps = []
                for p in params:
                    st = self.state[p]
                    step = st.get("step")
                    st["step"] = (torch.zeros((), dtype=torch.float32) if step is None else
                                  torch.as_tensor(step, dtype=torch.float32).detach().cpu().reshape(()).clone())
                    steps.append(st["step"])
                self._cache["step"] = steps
            torch._foreach_add_(steps, 1.0)
            from music_amd import _lib
            _lib.call("wn_rmsprop_flat", eng.flat.data_ptr(), gbase, sq.data_ptr(), buf, eng.spec.total, float(grp["lr"]), float(grp["alpha"]),
                      float(grp["eps"]), mom, 1.0, _lib.stream())
            return None
        return super().step(closure)


def get_optimizer(model, optimizer_type, learning_rate, momentum):
    """wavenet/train.py:28-42 — 'sgd' / 'rmsprop' (with momentum) / 'adam'; anything else -> None.  Each is torch's own optimizer class
    (same hyper-parameters, same state_dict) whose step() is one launch on the module's flat buffers when it can be."""
    if optimizer_type == 'sgd':
        return FlatSGD(model, lr=learning_rate, momentum=momentum)
    if optimizer_type == 'rmsprop':
        return FlatRMSprop(model, lr=learning_rate, momentum=momentum)
    if optimizer_type == 'adam':
        return FlatAdam(model, lr=learning_rate)          # an optim.Adam (same state_dict); one launch per step on the flat buffers


def save_model(model, num_iter, path):
    """``restore_dir + "wavenet{N}.model"`` = bare state_dict pickle (wavenet/train.py:45-50)."""
    checkpoint_path = path + "wavenet" + str(num_iter) + ".model"
    print("Storing checkpoint to {} ...".format(path))
    state = OrderedDict((k, v.detach().cpu().clone()) for k, v in model.state_dict().items())
    torch.save(state, checkpoint_path)
    print("Done!")


def load_model(model, path, model_name):
    """Returns the model, or None when the file does not exist (wavenet/train.py:53-73).  Keys
    saved from a DataParallel wrapper ("module." prefix) are accepted."""
    checkpoint_path = path + model_name
    print("Trying to restore saved checkpoint from ", "{}".format(checkpoint_path))
    if not os.path.exists(checkpoint_path):
        print("No checkpoint found!")
        return None
    print("Checkpoint found, restoring!")
    state_dict = torch.load(checkpoint_path, map_location="cpu")
    if list(state_dict.keys())[0][:6] == 'module':
        state_dict = OrderedDict((k[7:], v) for k, v in state_dict.items())
    model.load_state_dict(state_dict)
    return model


def _rotate_checkpoints(restore_dir, max_check_points):
    """Delete the numerically oldest wavenet{N}.model once max_check_points exist (:198-213)."""
    stored = glob.glob(restore_dir + "*.model")
    if len(stored) == max_check_points:
        def number(p):
            return int(p.split('/')[-1].split('.')[0][7:])
        stored = sorted(stored, key=cmp_to_key(lambda a, b: number(a) - number(b)))
        os.remove(stored[0])
        if os.path.exists(stored[0][:-len(".model")] + ".opt"):       # its optimizer state, if one was written
            os.remove(stored[0][:-len(".model")] + ".opt")


def _optimizer_state(optimizer, engine):
    """What wavenet{N}.opt holds: the torch optimizer's state_dict, or the flat Adam buffers of the fused step."""
    if engine is not None and engine.adam_state is not None:
        st = engine.adam_state
        return {"kind": "fused_adam", "m": st["m"].detach().cpu().clone(), "v": st["v"].detach().cpu().clone(), "t": int(st["t"])}
    return {"kind": "torch", "state": optimizer.state_dict()}


def _restore_optimizer_state(path, optimizer, engine_factory):
    """Load wavenet{N}.opt if it exists.  Returns the engine if the fused Adam state was restored."""
    if not os.path.exists(path):
        return None
    blob = torch.load(path, map_location="cpu")
    if blob.get("kind") == "fused_adam":
        engine = engine_factory()
        if engine is None:
            return None
        engine.adam_state["m"].copy_(blob["m"])
        engine.adam_state["v"].copy_(blob["v"])
        engine.adam_state["t"] = int(blob["t"])
        return engine
    if blob.get("kind") == "torch" and optimizer is not None:
        optimizer.load_state_dict(blob["state"])
    return None


def _resume_counter(log_dir):
    """Batches trained so far = third word of the last loss_log line (:160-166)."""
    with open(log_dir + 'loss_log.log', 'r') as f:
        lines = f.readlines()
    return int(lines[-1].split(' ')[2]) if lines else 0


def train():
    cuda_available = torch.cuda.is_available()
    train_params, wavenet_params, dataset_params = get_arguments()
    rank, world, local_rank = wdist.init_from_env()
    # every rank must draw the same shuffling permutation (each slices its chunk of the global batch): the JSON
    # "seed" if there is one, else one drawn on rank 0 and broadcast
    if world > 1:
        wdist.shared_seed(train_params.get("seed"))
    elif train_params.get("seed") is not None:
        torch.manual_seed(int(train_params["seed"]))

    net = wavenet(**wavenet_params)
    epoch_trained = 0
    restored_from = None
    if train_params["restore_model"]:
        net = load_model(net, train_params["restore_dir"], train_params["restore_model"])
        if net is None:
            print("Initialize network and train from scratch.")
            net = wavenet(**wavenet_params)
        else:
            epoch_trained = int(train_params["restore_model"].split('.')[0][7:])
            restored_from = train_params["restore_dir"] + train_params["restore_model"]

    if cuda_available is False and train_params["device_ids"] is not None:
        raise ValueError("Cuda is not avalable,", " can not train model using multi-gpu.")
    if world > 1:
        # DataParallel semantics: the JSON batch_size is the GLOBAL batch, split evenly
        assert dataset_params["batch_size"] % world == 0
        dataset_params = dict(dataset_params, shard=(rank, world))
    dataloader = audio_data_loader(**dataset_params)
    if cuda_available:
        net = net.cuda()
    wdist.broadcast_parameters(list(net.parameters()))

    print("Start training.")
    print("Writing logging information to ", "{}".format(train_params["log_dir"]))
    print("Models are saved in {}".format(train_params["restore_dir"]))

    optimizer = get_optimizer(net, train_params["optimizer"], train_params["learning_rate"],
                              train_params["momentum"])
    loss_func = nn.CrossEntropyLoss()
    if cuda_available:
        loss_func = loss_func.cuda()
    fused = bool(train_params.get("fused_step")) and train_params["optimizer"] == 'adam' and cuda_available
    is_writer = rank == 0
    loss_log_file = store_log_file = None
    if is_writer:
        os.makedirs(train_params["log_dir"], exist_ok=True)
        os.makedirs(train_params["restore_dir"], exist_ok=True)
        loss_log_file = open(train_params["log_dir"] + 'loss_log.log', 'a')
        store_log_file = open(train_params["log_dir"] + 'store_log.log', 'a')
    if world > 1:
        torch.distributed.barrier()
    num_trained = _resume_counter(train_params["log_dir"])

    device = next(net.parameters()).device
    total_loss = torch.zeros((), dtype=torch.float64, device=device)     # summed without host syncs
    engine = None
    keep_opt = bool(train_params.get("save_optimizer_state"))
    if keep_opt and restored_from is not None:

        def _fused_engine():
            if not fused:
                return None
            e = net._engine_for(device)
            e.adam_init(lr=train_params["learning_rate"])
            return e
        engine = _restore_optimizer_state(restored_from[:-len(".model")] + ".opt", optimizer, _fused_engine)
    for epoch in range(train_params["num_epochs"]):
        for i_batch, sampled_batch in enumerate(dataloader):
            piece = sampled_batch["audio_piece"]
            target = sampled_batch["audio_target"]
            # ragged last batch under data parallelism: weight of this rank's shard (0 and piece None = no items)
            dp_scale = float(sampled_batch.get("dp_scale", 1.0))
            if piece is not None:
                if cuda_available:
                    piece = piece.cuda(non_blocking=True)
                    target = target.cuda(non_blocking=True)
                target = target.view(-1)
            loss = torch.zeros((), device=device)
            if fused:
                if engine is None:
                    engine = net._engine_for(device)
                    engine.adam_init(lr=train_params["learning_rate"])
                if piece is not None:
                    loss = engine.loss_and_grad(piece.contiguous(), target)
                else:
                    engine.flat_grad.zero_()
                wdist.allreduce_flat_(engine.flat_grad, average=True, scale=dp_scale)
                engine.adam_step()
            else:
                optimizer.zero_grad()
                if piece is not None:
                    logits = net(piece)            # probabilities, named as in the reference (Q1)
                    loss = loss_func(logits, target)
                    loss.backward()
                wdist.allreduce_gradients(net.parameters(), average=True, scale=dp_scale)
                optimizer.step()
            loss = loss.detach() * dp_scale
            total_loss += loss.detach().double()
            num_trained += 1
            if num_trained % train_params["print_every"] == 0:
                if world > 1:
                    torch.distributed.all_reduce(total_loss)
                    total_loss /= world
                avg_loss = total_loss.item() / train_params["print_every"]
                if is_writer:
                    loss_log_file.writelines("Trained over " + str(num_trained) + " pieces," +
                                             "Average loss is " + str(avg_loss) + "\n")
                    loss_log_file.flush()
                total_loss.zero_()

        if (epoch + 1) % train_params["check_point_every"] == 0 and is_writer:
            _rotate_checkpoints(train_params["restore_dir"], train_params["max_check_points"])
            save_model(net, epoch_trained + epoch + 1, train_params["restore_dir"])
            if keep_opt:
                torch.save(_optimizer_state(optimizer, engine),
                           train_params["restore_dir"] + "wavenet" + str(epoch_trained + epoch + 1) + ".opt")
            store_log_file.writelines("Epoch " + str(epoch_trained + epoch + 1) + ", model saved!\n")
            store_log_file.flush()
    if is_writer:
        loss_log_file.close()
        store_log_file.close()


if __name__ == '__main__':
    train()
