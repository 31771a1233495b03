"""Worker for tests/test_gpu_dist.py: one rank per GPU under torch.distributed.run, backend nccl (= RCCL)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    workdir = sys.argv[1]
    shards = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # logical shards of the global batch (2 clips each)
    from music_amd import dist as wdist
    from music_amd.model import wavenet
    rank, world, local = wdist.init_from_env(force=True)
    assert torch.distributed.is_initialized() or world == 1
    backend = torch.distributed.get_backend() if torch.distributed.is_initialized() else "none"
    cfg = dict(filter_width=2, dilations=[1, 2, 4, 8, 16, 32], dilation_channels=64, residual_channels=64,
               skip_channels=64, quantization_channels=256, use_bias=False)
    torch.manual_seed(0)
    net = wavenet(**cfg).cuda()
    wdist.broadcast_parameters(net.parameters())
    eng = net._engine_for(torch.device("cuda", local))
    eng.adam_init(lr=1e-3)
    rng = np.random.default_rng(100)
    assert shards % world == 0
    B, T = 2 * shards // world, net.receptive_field + 300
    # the SAME global batch whatever the world size; rank r trains on its contiguous chunk of it
    codes = torch.from_numpy(rng.integers(0, 256, size=(2 * shards, T + 1)).astype(np.int32))[rank * B:(rank + 1) * B].cuda()
    rf = net.receptive_field
    W = T - rf + 1
    losses, gsum = [], None
    for step in range(3):
        x = eng.onehot(codes[:, :T].contiguous(), scrambled=True)
        target = codes[:, rf:rf + W].to(torch.int64).contiguous().view(-1)
        loss = eng.loss_and_grad(x, target)
        before = eng.flat_grad.clone()
        if torch.distributed.is_initialized():
            torch.distributed.all_reduce(eng.flat_grad)             # ONE flat bucket, RCCL (also with a single rank)
        if world == 1:                                              # a 1-rank sum must not change anything
            assert torch.equal(before, eng.flat_grad)
        eng.adam_step(gscale=1.0 / world)
        if torch.distributed.is_initialized():
            torch.distributed.all_reduce(loss)
            loss = loss / world
        losses.append(float(loss.item()))
        gsum = float(eng.flat_grad.abs().sum().item()) / world
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    if rank == 0:
        json.dump({"backend": backend, "world": world, "losses": losses, "gsum": gsum,
                   "psum": float(eng.flat.double().abs().sum().item())}, open(os.path.join(workdir, "dist_gpu.json"), "w"))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
