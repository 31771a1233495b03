"""Developer aid: per-phase clock sums of the two-role backward block (needs a -DRW_DBG build of the library,
passed as WAVENET_HIP_LIB).  Prints the share of each phase in the R and W waves' loop time."""
import ctypes, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music_amd import _lib
lib = _lib.load()
out = (ctypes.c_ulonglong * 8)()
lib.wn_rw_dbg_read(out, 1)
import tools.kbench as kb
sys.argv = ["kbench", "bwd", "--reps", "3"]
kb.main()
torch.cuda.synchronize()
lib.wn_rw_dbg_read(out, 0)
v = list(out)
names = ["R mfma", "R gate+store", "R barrier", "-", "W fill", "W wgrad", "W barrier", "-"]
for base in (0, 4):
    tot = sum(v[base:base + 3]) or 1
    print(", ".join("%s %.1f%%" % (names[base + i], 100.0 * v[base + i] / tot) for i in range(3)), " total clocks", tot)
