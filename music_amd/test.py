"""Counterpart of the reference's ``wavenet/test.py`` (a smoke / timing script, not a unit test): the SHIPPED
``wavenet_params.json`` model trained on random data - dense float features of 256 x 32000 per item, 27907 targets
(= 32000 - 4094 + 1), batch 2, Adam 1e-3 - printing per epoch which share of the step the forward, the backward and the
optimizer take (wavenet/test.py:11-70).  ``params/wavenet_params_shipped.json`` holds those hyper-parameters (40 blocks, 32 / 32
channels, 512 skip channels, receptive field 4094; ``params/wavenet_params.json`` is BASELINE config 2).  Same names
(``simple_dataset``, ``test``), same step order, same three print lines.

The reference file itself cannot run on Python >= 3.7 (``.cuda(async=True)``) and needs two CUDA devices; this
module is the drop-in.  Differences, all stated:
  * ``nn.DataParallel(net, device_ids=[0, 1])`` (:29) is one process per GPU here (``torchrun --nproc-per-node 2 test.py``,
    music_amd/dist.py): every rank takes its share of each batch of 2 and the gradients are averaged by one flat all-reduce,
    counted as backward time.  A single process runs the whole batch on its one device.
  * the reference's clocks are ``time.time()`` deltas around asynchronous launches (:54-64), i.e. they time the ENQUEUE; here the
    device is synchronised at every boundary, so the three fractions are device time.
  * the reference draws ``randn(256, 3 200 000)`` once as float64 (6.5 GB of host memory, :13-14) and slices it; here item ``idx``
    is drawn when it is asked for from a generator seeded with (seed, idx) - the same distribution, 33 MB at a time.
  * ``test()`` takes the epoch / item counts as arguments (defaults: the reference's 100 / 100) and returns the last epoch's three
    fractions; run as a script it reads WN_TEST_EPOCHS / WN_TEST_ITEMS from the environment.
"""
import json
import os
import time

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import Dataset, DataLoader

try:
    from . import dist as wdist
    from .model import wavenet
    from .train import FlatAdam
except ImportError:                                  # run as a script / bare modules from the CWD
    from music_amd import dist as wdist
    from music_amd.model import wavenet
    from music_amd.train import FlatAdam

_PARAMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "params", "wavenet_params_shipped.json")


class simple_dataset(Dataset):
    """Random features / targets in the reference's shapes (wavenet/test.py:11-21): ``feature`` float32 (channels, length),
    ``target`` int64 (targets,), 100 items."""

    def __init__(self, items=100, length=32000, targets=27907, channels=256, seed=0):
        self.items, self.length, self.targets, self.channels, self.seed = items, length, targets, channels, seed

    def __len__(self):
        return self.items

    def __getitem__(self, idx):
        if not 0 <= idx < self.items:
            raise IndexError(idx)
        rng = np.random.default_rng([self.seed, idx])
        sample = {}
        sample["feature"] = torch.from_numpy(rng.standard_normal((self.channels, self.length), dtype=np.float32))
        sample["target"] = torch.from_numpy(rng.integers(0, 256, size=self.targets, dtype=np.int64))
        return sample


def test(epochs=100, items=100, batch_size=2, params_path=_PARAMS, num_workers=4, seed=0, out=print):
    if not torch.cuda.is_available():
        raise RuntimeError("music_amd.test needs a GPU: there is no CPU path (DESIGN.md section 1)")
    with open(params_path, 'r') as f:
        params = json.load(f)
    rank, world, _ = wdist.init_from_env()
    if batch_size % world != 0:
        raise ValueError("batch_size %d does not split over %d ranks" % (batch_size, world))
    torch.manual_seed(seed)                              # every rank: the same shuffling permutation, the same initial weights
    net = wavenet(**params).cuda()
    wdist.broadcast_parameters(list(net.parameters()))
    rf = net.receptive_field
    length = 32000
    dataset = simple_dataset(items=items, length=length, targets=length - rf + 1, channels=params.get("quantization_channels", 256), seed=seed)
    dataloader = DataLoader(dataset, shuffle=True, num_workers=num_workers, batch_size=batch_size, pin_memory=True)
    loss_func = nn.CrossEntropyLoss().cuda()
    optimizer = FlatAdam(net, lr=1e-3)                  # optim.Adam(net.parameters(), lr=1e-3), one launch per step
    per = batch_size // world
    sync = torch.cuda.synchronize
    fractions = None
    test.last_net, test.last_loss = net, None           # for whoever wants to look at what was trained
    for epoch in range(epochs):
        forward_time = backward_time = optimize_time = 0.0
        for i, sample in enumerate(dataloader):
            optimizer.zero_grad()
            feature, label = sample["feature"], sample["target"]
            n = feature.shape[0]                          # a last batch may be short: shard it as DataParallel's scatter does
            lo, hi = min(rank * per, n), min((rank + 1) * per, n)
            scale = float(hi - lo) * world / n if world > 1 else 1.0
            feature, label = feature[lo:hi].cuda(non_blocking=True), label[lo:hi].cuda(non_blocking=True)
            sync()
            now = time.time()
            logits = net(feature).view(-1, 256) if hi > lo else None
            sync()
            forward_time += time.time() - now

            now = time.time()
            if logits is not None:
                loss = loss_func(logits, label.view(-1))
                loss.backward()
                test.last_loss = loss.detach()
            wdist.allreduce_gradients(net.parameters(), average=True, scale=scale)
            sync()
            backward_time += time.time() - now

            now = time.time()
            optimizer.step()
            sync()
            optimize_time += time.time() - now
        total_time = forward_time + backward_time + optimize_time
        if total_time > 0:
            fractions = (forward_time / total_time, backward_time / total_time, optimize_time / total_time)
            if rank == 0:
                out("Forward consumption is {}".format(fractions[0]))
                out("Backward consumption is {}".format(fractions[1]))
                out("Optimize consumption is {}".format(fractions[2]))
    return fractions


if __name__ == '__main__':
    test(epochs=int(os.environ.get("WN_TEST_EPOCHS", "100")), items=int(os.environ.get("WN_TEST_ITEMS", "100")))
