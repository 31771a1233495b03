"""Data parallelism for the WaveNet path: one process per GPU, RCCL all-reduce over xGMI.

Replaces ``nn.DataParallel`` (wavenet/train.py:116-122): instead of scatter / replicate / gather /
reduce-to-GPU-0 every step, each rank keeps a persistent replica and its own clips, and the only
exchange is ONE sum all-reduce of the flat fp32 gradient buffer per step (5.08 MB at the 30-layer
config), followed by the 1/world scale.  Equal per-rank batches make that the gradient of the mean
loss over the global batch, which is what DataParallel computes.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* if WORLD_SIZE > 1 (``force``: also for one rank
    under a launcher, so that a 1-GPU box exercises the RCCL path).  Returns (rank, world, local_rank)."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    r = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if (w > 1 or (force and "MASTER_ADDR" in os.environ)) and not dist.is_initialized():
        if backend is None:
            # WN_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses duplicate devices) - used by the GPU tests
            # to run the N-rank code path on a 1-GPU box; production is nccl (= RCCL), one rank per GPU
            backend = os.environ.get("WN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            n_dev = max(1, torch.cuda.device_count())
            if backend == "gloo":
                lr = lr % n_dev              # several ranks may share one GPU under gloo (the 1-GPU-box tests)
            elif lr >= n_dev:
                raise RuntimeError("music_amd.dist: LOCAL_RANK %d but only %d GPU(s) visible - one rank per GPU under "
                                   "RCCL (nproc-per-node too large?)" % (lr, n_dev))
            torch.cuda.set_device(lr)
        if backend == "nccl":
            dist.init_process_group(backend, device_id=torch.device("cuda", lr))
        else:
            dist.init_process_group(backend)
    return r, w, lr


def broadcast_parameters(params, src=0):
    """Make every replica identical to rank `src` (DataParallel replicates from device 0)."""
    if world() == 1:
        return
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        dist.broadcast(flat, src)
        o = 0
        for p in params:
            n = p.numel()
            p.copy_(flat[o:o + n].view_as(p))
            o += n


def shared_seed(seed=None):
    """One torch seed for every rank: `seed` if given, else one drawn on rank 0 and broadcast.  All ranks then call
    torch.manual_seed with it, so that shuffling DataLoaders built afterwards draw the SAME permutation (each rank
    slices its own chunk of every global batch) and per-forward random draws agree across replicas."""
    if seed is None:
        seed = int(torch.empty((), dtype=torch.int64).random_(0, 2 ** 31 - 1).item())
    if world() > 1:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([int(seed)], dtype=torch.int64, device=dev)
        dist.broadcast(t, 0)
        seed = int(t.item())
    torch.manual_seed(int(seed))
    return int(seed)


def allreduce_flat_(flat_grad, average=True, scale=1.0):
    """In-place sum all-reduce of one flat gradient buffer (+ 1/world).  `scale` multiplies the LOCAL buffer first
    (a ragged shard's dp_scale, see faster_audio_data._Collate)."""
    w = world()
    if w == 1:
        return flat_grad
    if scale != 1.0:
        flat_grad.mul_(scale)
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    if average:
        flat_grad.mul_(1.0 / w)
    return flat_grad


def allreduce_gradients(params, average=True, scale=1.0):
    """All-reduce the .grad of `params` as ONE flat bucket.  A parameter without a grad on THIS rank contributes zeros (a
    rank whose shard of a ragged batch is empty ran no backward) and receives the reduced gradient IF some rank had
    one, so every rank issues the same collective and takes the same optimizer step.  A parameter that has no gradient
    on ANY rank (frozen, unused: `requires_grad=False` or never reached) keeps `.grad = None`, as under the reference's
    DataParallel - no optimizer state is created for it.  The "has a gradient" flags ride in the same bucket."""
    w = world()
    if w == 1:
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    ref = params[0]
    has = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=ref.dtype, device=ref.device)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params] + [has])
    allreduce_flat_(flat, average, scale)
    # (flags are scaled like the gradients: > 0 wherever a rank with a non-empty shard had one; read back - a device
    # sync - only on a rank that is missing some gradient itself)
    any_has = (flat[-len(params):] > 0).tolist() if any(p.grad is None for p in params) else [True] * len(params)
    o = 0
    for p, h in zip(params, any_has):
        n = p.numel()
        if p.grad is not None:
            p.grad.copy_(flat[o:o + n].view_as(p))
        elif h:
            p.grad = flat[o:o + n].view_as(p).clone()
        o += n
