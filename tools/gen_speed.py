#!/usr/bin/env python3
"""Developer tool (GPU): step time of the GENERAL plan (music_amd/engine_generic.py) at the config-2 shape (where the fast
engine runs 4.3 ms) and on a 128-channel model only it covers."""
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
import bench
from music_amd.engine_generic import GenericWaveNetEngine
from music_amd.model import wavenet
torch.manual_seed(0)
net = wavenet(**bench.CFG).cuda()
sd = {k: v for k, v in net.state_dict().items()}
eng = GenericWaveNetEngine(bench.CFG["dilations"], 64, 64, 256, device="cuda")
eng.load_state_dict_tensors(sd)
eng.adam_init()
codes = bench.synth_codes(0, 8, bench.T)
rf = net.receptive_field; W = bench.T - rf + 1
piece = codes[:, :bench.T].contiguous(); target = codes[:, rf:rf + W].to(torch.int64).contiguous().view(-1)
x = eng.onehot(piece, True)
for _ in range(3):
    l = eng.loss_and_grad(x, target); eng.adam_step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    l = eng.loss_and_grad(x, target); eng.adam_step()
torch.cuda.synchronize()
print("general plan at config 2 (8 x 16000): %.2f ms/step, loss %.5f" % ((time.perf_counter() - t0) / 10 * 1e3, l.item()))
# 128-channel model, same depth
eng2 = GenericWaveNetEngine(bench.CFG["dilations"], 128, 128, 256, device="cuda")
torch.nn.init.uniform_(eng2.flat, -0.05, 0.05)
for _ in range(2):
    l = eng2.loss_and_grad(x, target)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    l = eng2.loss_and_grad(x, target)
torch.cuda.synchronize()
print("general plan, 128 / 128 / 256 channels, 30 blocks, 8 x 16000: %.2f ms/step" % ((time.perf_counter() - t0) / 5 * 1e3))
