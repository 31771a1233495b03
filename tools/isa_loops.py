#!/usr/bin/env python3
"""Developer aid: instruction mix of the loops of one kernel in a hipcc -S listing (no GPU needed).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only music_amd/csrc/wn_respq.hip -o /tmp/respq.s
    python tools/isa_loops.py /tmp/respq.s _Z17resblock_bwd_pq_kILb1ELb0ELb0ELb1EEv11WnResPqArgs

Prints, for every backward branch (loop) of the kernel that spans more than `min` instructions, the counts of MFMA, other
VALU, SALU, LDS, vector-memory loads / stores, waits and barriers between the branch target and the branch."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")): return "vmem_ld"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")): return "vmem_st"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body, labels = [], {}
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            if re.match(r"^\.LBB\d+_\d+:", t):
                labels[t.split(":")[0]] = len(body)
            continue
        if re.match(r"^\.?LBB\d+_\d+:", t):
            labels[t.split(":")[0]] = len(body)
            continue
        body.append(t.split()[0:2])
    print("%s: %d instructions" % (kern, len(body)))
    for i, ins in enumerate(body):
        if ins[0].startswith(("s_cbranch", "s_branch")) and len(ins) > 1 and ins[1] in labels and labels[ins[1]] <= i:
            a = labels[ins[1]]
            if i - a < lo:
                continue
            c = Counter(classify(x[0]) for x in body[a:i + 1])
            ops = Counter(x[0] for x in body[a:i + 1])
            print("loop [%d, %d] %d instr: " % (a, i, i - a + 1) + ", ".join("%s %d" % kv for kv in sorted(c.items())))
            print("    top: " + ", ".join("%s %d" % kv for kv in ops.most_common(14)))


if __name__ == "__main__":
    main()
