"""Developer aid: per-phase clock sums of the one-launch backward block (needs a -DPQ_DBG build of the library, passed as
WAVENET_HIP_LIB).  Prints the share of each phase in the R and W waves' loop time.  `ae`: the conditioned form, on the
config-4 autoencoder step (tools/ae_phases.py) instead of the config-2 backward."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music_amd import _lib
lib = _lib.load()
out = (ctypes.c_ulonglong * 16)()
AE = sys.argv[1:2] == ["ae"]
if AE:
    import tools.ae_phases as kb
else:
    import tools.kbench as kb
sys.argv = ["kbench", "bwd", "--reps", "2"]
kb.main()
torch.cuda.synchronize()
lib.wn_pq_dbg_read(out, 1)
sys.argv = ["kbench", "bwd", "--reps", "3"]
kb.main()
torch.cuda.synchronize()
lib.wn_pq_dbg_read(out, 0)
v = list(out)
names = ["R fill_x", "R recompute", "R gate+put", "R barrier", "W fill_dy", "W wgrad", "W pq", "W store", "W convert+loads", "W barrier"]
for lo, hi in ((0, 4), (4, 10)):
    tot = sum(v[lo:hi]) or 1
    print(", ".join("%s %.1f%%" % (names[i], 100.0 * v[i] / tot) for i in range(lo, hi)), " total clocks", tot)
tot = sum(v[0:4]) or 1
print("R first phase apart (share of the R loop): " + ", ".join("%s %.1f%%" % (n, 100.0 * v[10 + i] / tot) for i, n in enumerate(
    ["x fill + loads", "dy fill + loads", "dx product half", "dy row loads"])))
