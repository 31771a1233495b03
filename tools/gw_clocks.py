"""Developer aid: per-phase clock sums of the wide GEMM's k-step on the skip product (needs a -DGW_DBG build of the library,
passed as WAVENET_HIP_LIB).  Prints each phase's share of the loop time for waves 0-3 and 4-7."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music_amd import _lib
lib = _lib.load()
out = (ctypes.c_ulonglong * 16)()
import tools.kbench as kb
sys.argv = ["kbench", "skip", "--reps", "2"]
kb.main()
torch.cuda.synchronize()
lib.wn_gw_dbg_read(out, 1)
sys.argv = ["kbench", "skip", "--reps", "3"]
kb.main()
torch.cuda.synchronize()
lib.wn_gw_dbg_read(out, 0)
v = list(out)
names = ["reads + MFMAs of 4 row tiles", "wait for global loads", "convert + LDS fill", "issue next loads", "MFMAs of 4 row tiles", "barrier"]
for base in (0, 8):
    tot = sum(v[base:base + 6]) or 1
    print("waves %d-%d: " % (base // 2, base // 2 + 3) + ", ".join("%s %.1f%%" % (names[i], 100.0 * v[base + i] / tot) for i in range(6)), " total clocks", tot)
