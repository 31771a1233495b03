"""Worker for tests/test_dist_cpu.py (launched by torch.distributed.run with the gloo backend)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    mode, workdir = sys.argv[1], sys.argv[2]
    os.chdir(workdir)
    from music_amd import train as T
    from music_amd import faster_audio_data as fad
    from tests.cpu_model import OracleWavenet, onehot_oracle
    fad.onehot_device = onehot_oracle
    T.wavenet = OracleWavenet
    if mode == "train":
        T.train()
    elif mode == "grads":
        # gradient averaging == DataParallel's global-batch-mean gradient
        from music_amd import dist as wdist
        rank, world, _ = wdist.init_from_env("gloo")
        torch.manual_seed(0)
        cfg = json.load(open("cfg.json"))
        net = OracleWavenet(**cfg)
        data = torch.load("batch.pt")
        x, y = data["x"], data["y"]
        n = x.size(0) // world
        xs, ys = x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]
        loss = torch.nn.CrossEntropyLoss()(net(xs), ys.reshape(-1))
        loss.backward()
        wdist.allreduce_gradients(net.parameters(), average=True)
        # a parameter with no gradient on ANY rank keeps .grad None, as under DataParallel (the last block's dense conv
        # never reaches the output, wavenet/model.py:122-129); a frozen one is not touched
        last_dense = net.dilation_layer_stack[4 * (len(cfg["dilations"]) - 1) + 2].weight
        assert last_dense.grad is None
        frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)
        unused = torch.nn.Parameter(torch.ones(2))
        only_here = torch.nn.Parameter(torch.ones(2))
        if rank == 0:
            only_here.grad = torch.full((2,), 4.0)
        wdist.allreduce_gradients([frozen, unused, only_here], average=True)
        assert frozen.grad is None and unused.grad is None
        assert torch.equal(only_here.grad, torch.full((2,), 4.0 / world))       # a rank without it receives the mean
        if rank == 0:
            torch.save({k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in net.named_parameters()},
                       "grads_dp.pt")
        torch.distributed.barrier()


if __name__ == "__main__":
    main()
