"""Developer aid: when do the workgroups of the backward block's launches start, reach their first barrier, leave their item
loop and end?  Needs a -DPQ_SPAN build of the library (WAVENET_HIP_LIB).  Microseconds on the 100 MHz realtime clock."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from music_amd import _lib
lib = _lib.load()
import tools.kbench as kb
sys.argv = ["kbench", "bwd", "--reps", "2"]
kb.main()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * (64 * 256 * 8))()
lib.wn_pq_span_read(out)
v = np.array(list(out), dtype=np.float64).reshape(64, 256, 8)
first = np.where(v[:, :, 0] > 0, v[:, :, 0], np.inf).min(axis=1)
order = [s for s in np.argsort(first) if np.isfinite(first[s])]
rows = []
for s in order:
    live = (v[s, :, 0] > 0) & (v[s, :, 5] > v[s, :, 0]) & (v[s, :, 5] - v[s, :, 0] < 2e4)
    if live.sum() >= 200:
        rows.append((s, live))
rows = rows[-29:]
names = ["R first barrier", "R loop end", "W first barrier", "W loop end", "end"]
print("launch  wgs items | start spread | " + " | ".join(names) + " | slowest end | next launch's first start after this one's last end")
for i, (s, live) in enumerate(rows):
    st = v[s][live][:, 0]
    d = (v[s][live][:, 1:6] - v[s][live][:, :1]) / 100.0
    t0 = st.min()
    last_end = v[s][live][:, 5].max()
    gap = (v[rows[i + 1][0]][rows[i + 1][1]][:, 0].min() - last_end) / 100.0 if i + 1 < len(rows) else float("nan")
    print("%4d   %4d %5.1f | %5.2f | " % (i, live.sum(), v[s][live][:, 6].mean(), (st.max() - st.min()) / 100.0) +
          " | ".join("%6.2f" % x for x in d.mean(axis=0)) + " | %6.2f | %5.2f" % ((last_end - t0) / 100.0, gap))
# which workgroups are the slow ones? (last whole launch of the list)
s, live = rows[-2]
ids = np.nonzero(live)[0]
dur = (v[s][live][:, 5] - v[s][live][:, 0]) / 100.0
loop = (v[s][live][:, 4] - v[s][live][:, 3]) / 100.0
startd = (v[s][live][:, 0] - v[s][live][:, 0].min()) / 100.0
print("by XCD (block id mod 8): mean duration / mean loop / mean start delay, us")
for x in range(8):
    m = (ids % 8) == x
    print("  xcd-group %d: n %3d  dur %6.2f  loop %6.2f  start +%4.2f  max dur %6.2f" % (x, m.sum(), dur[m].mean(), loop[m].mean(), startd[m].mean(), dur[m].max()))
o = np.argsort(-dur)[:12]
print("slowest workgroups (block id, items, duration, loop, start delay):")
for k in o:
    print("  %3d  items %2d  %6.2f  %6.2f  +%4.2f" % (ids[k], int(v[s][live][k, 6]), dur[k], loop[k], startd[k]))
print("fastest: %.2f  median %.2f  p90 %.2f  max %.2f" % (dur.min(), np.median(dur), np.percentile(dur, 90), dur.max()))
