"""Developer aid: per-phase clock sums of the fused forward epilogue's skip loop (needs a -DEPI_DBG build of the library, passed as
WAVENET_HIP_LIB: bash tools/mkvar.sh epidbg "-DEPI_DBG" wn_epilogue.hip).  Prints each phase's share of the loop for waves 0-3 / 4-7,
and cycles per iteration / per tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import CFG, B_LOCAL, T
from music_amd import _lib
from music_amd.model import wavenet
lib = _lib.load()
out = (ctypes.c_ulonglong * 32)()
torch.manual_seed(0)
net = wavenet(**CFG).cuda()
eng = net._engine_for(torch.device("cuda", 0))
rng = np.random.default_rng(0)
codes = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL, T)).astype(np.int32)).cuda()
target = torch.from_numpy(rng.integers(0, 256, size=(B_LOCAL * (T - 3070),)).astype(np.int64)).cuda()
for _ in range(30):
    eng.loss_and_grad_codes(codes, target)
torch.cuda.synchronize()
lib.wn_epi_dbg_read(out, 1)
n = 20
for _ in range(n):
    eng.loss_and_grad_codes(codes, target)
torch.cuda.synchronize()
lib.wn_epi_dbg_read(out, 0)
v = list(out)
tiles = v[16] or 1
names = ["weight requests", "k-step 0 (reads + MFMAs)", "split + LDS fill + z requests", "k-step 1", "barrier"]
for base in (0, 8):
    tot = sum(v[base:base + 5]) or 1
    print("waves %d-%d: " % (base // 2, base // 2 + 3) + ", ".join("%s %.1f%%" % (names[i], 100.0 * v[base + i] / tot) for i in range(5)))
    print("   per wave: loop %.0f cycles per tile = %.0f per iteration (30); post-processing part %.0f; whole kernel %.0f"
          % (v[base + 5] / tiles / 4, v[base + 5] / tiles / 4 / 30, v[base + 6] / tiles / 4, v[base + 7] / tiles / 4))
print("tiles", tiles // n, "per launch")
