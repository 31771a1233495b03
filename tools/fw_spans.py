"""Developer aid: when do the workgroups of the forward block's launches start and end?  Needs a -DFW_DBG build of the
library (WAVENET_HIP_LIB).  Runs forward passes of config 2 and prints, per launch of the last pass: first / last
workgroup start, first / last end (after the last store was ISSUED), and the gap to the next launch's first start -
all in microseconds on the 100 MHz realtime clock."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from music_amd import _lib
lib = _lib.load()
import tools.kbench as kb
sys.argv = ["kbench", "fwd", "--reps", "2"]
kb.main()          # per-layer loops: warm-up only
sys.argv = ["kbench", "bwd", "--reps", "2"]
kb.main()          # whole steps: the last 30 forward launches are one forward stack back to back
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * (64 * 256 * 8))()
lib.wn_fw_dbg_read(out)
v = np.array(list(out), dtype=np.float64).reshape(64, 256, 8)
first = np.where(v[:, :, 0] > 0, v[:, :, 0], np.inf).min(axis=1)
order = np.argsort(first)
rows = []
for s in order:
    st = v[s, :, 0]
    live = (st > 0) & (v[s, :, 7] - st < 1e4) & (v[s, :, 7] > st)        # same pass
    if live.sum() < 200:
        continue
    rows.append((s, live))
rows = rows[-6:]
names = ["weights staged", "f/g product done", "gate done", "z stores issued", "dense product done", "x stores issued (wave 0)", "all waves' stores issued"]
print("per launch (last 6 whole launches): mean over workgroups of the time since the workgroup's start, us")
for s, live in rows:
    d = (v[s][live][:, 1:] - v[s][live][:, :1]) / 100.0
    st = v[s][live][:, 0]
    print("slot %2d  wgs %3d  start spread %.2f us | " % (s, live.sum(), (st.max() - st.min()) / 100.0) +
          ", ".join("%s %.2f" % (n, m) for n, m in zip(names, d.mean(axis=0))) + " | slowest workgroup %.2f" % d[:, -1].max())
clk = (ctypes.c_ulonglong * (64 * 256 * 2))()
lib.wn_fw_clk_read(clk)
c = np.array(list(clk), dtype=np.float64).reshape(64, 256, 2)
for s_, live in rows[-2:]:
    dt_rt = (v[s_][live][:, 2] - v[s_][live][:, 1]) / 100.0          # us
    dt_ck = c[s_][live][:, 1] - c[s_][live][:, 0]
    print("slot %2d  shader clock during the f/g phase: %.0f MHz (mean of %d workgroups; %.0f clocks in %.2f us)" %
          (s_, (dt_ck / dt_rt).mean(), live.sum(), dt_ck.mean(), dt_rt.mean()))
