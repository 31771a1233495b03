#!/usr/bin/env python3
"""Headline benchmark: audio samples/sec trained, 30-layer WaveNet @16 kHz (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed, backend nccl = RCCL); weak scaling, 8 clips x 16000 samples
per GPU.  One "step" = on-device one-hot build + forward + CrossEntropy-on-probabilities +
backward + gradient all-reduce (N > 1) + Adam, i.e. wavenet/train.py:171-182 of the reference, with
the int32 sample codes already resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves as a CHILD
    `python -m torch.distributed.run ... bench.py <same args>` and relay its output and exit code.  This runs before
    torch is imported - the parent never touches the GPU, and nothing that has is ever re-exec'ed."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    known, _ = ap.parse_known_args()
    if known.gpus <= 1:
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(known.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes on this host driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.call(cmd, env=env))


if __name__ == "__main__":
    _self_launch()

import numpy as np  # noqa: E402
import torch  # noqa: E402

CFG = dict(filter_width=2, dilations=[2 ** i for i in range(10)] * 3, dilation_channels=64,
           residual_channels=64, skip_channels=256, quantization_channels=256, use_bias=False)
B_LOCAL, T = 8, 16000
HBM_PEAK = 8.0e12          # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec); 6.29e12 measured copy


def synth_codes(rank, b, t):
    """rng(1234+rank), 0.3*N(0,1) clipped to [-1,1] -> canonical mu-law codes."""
    rng = np.random.default_rng(1234 + rank)
    a = np.clip(0.3 * rng.standard_normal((b, t + 1)), -1.0, 1.0).astype(np.float32)
    tabs = np.load(os.path.join(ROOT, "music_amd", "mulaw_tables.npz"))
    from music_amd import _lib
    from music_amd._lib import call, ptr
    ad = torch.from_numpy(a).cuda()
    thr = torch.from_numpy(tabs["thresholds"]).cuda()
    codes = torch.empty(a.size, dtype=torch.uint8, device="cuda")
    call("wn_mulaw_encode_tbl", ptr(ad), ptr(thr), ptr(codes), a.size, _lib.stream())
    return codes.view(b, t + 1).to(torch.int32)


def stack_bytes(dil, r, d, b, t):
    """SURVEY §8(d) algorithmic bytes of the dilated-conv stack, fp32, per step (B clips)."""
    L = [t - 1]
    for x in dil:
        L.append(L[-1] - x)
    w = L[-1]
    fwd = 4 * sum(r * L[i] + r * L[i + 1] + d * w for i in range(len(dil)))
    bwd = 4 * sum(r * L[i + 1] + r * L[i] + d * w + r * L[i] for i in range(len(dil)))
    return fwd * b, bwd * b


def cpu_baseline(seconds_budget=25.0):
    """The CPU oracle (== the reference's ATen-CPU op sequence, pinned by tests/golden) timed on
    this host: full training step (forward + CE + backward + Adam) on ONE clip of 16000 samples."""
    from oracle import wavenet_oracle as wo
    from oracle import intops
    torch.manual_seed(0)
    from music_amd.model import wavenet
    net = wavenet(**CFG)
    params = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-4)
    rng = np.random.default_rng(1234)
    codes = rng.integers(0, 256, size=(T + 1,))
    x = torch.from_numpy(intops.one_hot_scrambled(codes[:T]))[None]
    rf = 3071
    target = torch.from_numpy(codes[rf:rf + T - rf + 1].astype(np.int64))
    # ATen's CPU conv kernels stop scaling (and then collapse) long before a 256-core host is
    # full: use at most 32 threads and say so in `cores`
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    times = []
    t_start = time.time()
    for it in range(4):
        t0 = time.time()
        opt.zero_grad()
        loss = wo.ce_on_probs(wo.wavenet_forward(params, CFG["dilations"], x), target)
        loss.backward()
        opt.step()
        times.append(time.time() - t0)
        if time.time() - t_start > seconds_budget:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": T / best, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%d full training steps (fwd+CE+bwd+Adam) of the 30-layer config on 1 clip x 16000 samples, "
                      "torch CPU threads=%d, best of %d after 1 warm-up" % (len(times), cores, max(1, len(times) - 1))}


def measured_copy_gbs():
    """Device-to-device copy rate (read + write bytes / time) of a 1 GiB buffer, best of 5."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.empty(n, dtype=torch.float32, device="cuda")
    a.fill_(1.0)
    b.copy_(a)
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    del a, b
    return 2.0 * n * 4 / (best * 1e-3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="f16x3,bf16x3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phases", action="store_true", help="print the per-phase GPU time table to stderr")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch plumbing only (CPU, gloo): every rank joins the group, rank 0 prints the world size")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, which launches the "
                         "N ranks itself, or under `python -m torch.distributed.run --nproc-per-node N`)" % (args.gpus, world))
    if args.dry_run:
        import torch.distributed as dist
        if "MASTER_ADDR" in os.environ:
            dist.init_process_group("gloo")
            t = torch.tensor([rank + 1.0])
            dist.all_reduce(t)
            dist.destroy_process_group()
        else:
            t = torch.tensor([1.0])
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "rank_sum": t.item()}))
        return
    torch.cuda.set_device(local)
    dist = None
    # under torchrun (RANK/MASTER_ADDR set) the process group is always created, also for 1 rank,
    # so that the RCCL path of an N-GPU run is the path a 1-rank launch exercises
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from music_amd.model import wavenet
    torch.manual_seed(0)                               # identical replicas on every rank
    net = wavenet(**CFG)
    net.precision = tuple(args.precision.split(","))
    net = net.cuda()
    eng = net._engine_for(torch.device("cuda", local))
    eng.adam_init(lr=1e-4)
    codes = synth_codes(rank, B_LOCAL, T)
    rf = net.receptive_field
    W = T - rf + 1
    piece = codes[:, :T].contiguous()
    target = codes[:, rf:rf + W].to(torch.int64).contiguous().view(-1)

    def step():
        x = eng.onehot(piece, scrambled=True)
        loss = eng.loss_and_grad(x, target)
        if use_dist:
            dist.all_reduce(eng.flat_grad)          # ONE flat fp32 bucket (5.08 MB), RCCL over xGMI
            eng.mark("allreduce")
        eng.adam_step(gscale=1.0 / world)
        return loss

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    eng.marks = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    marks, eng.marks = eng.marks, None
    if use_dist:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()

    # per-phase GPU time from the HIP events recorded on the launch stream inside the timed region
    phase = {}
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        phase[n1] = phase.get(n1, 0.0) + e0.elapsed_time(e1)
    phase = {k: v / args.steps for k, v in phase.items()}       # ms per step
    if args.phases and rank == 0:
        print(json.dumps({"phase_ms_per_step": phase}), file=sys.stderr)

    fwd_b, bwd_b = stack_bytes(CFG["dilations"], 64, 64, B_LOCAL, T)
    n_layers = len(CFG["dilations"])
    nan = float("nan")
    fwd_ms, bwd_ms = phase.get("stack_fwd", nan), phase.get("stack_bwd", nan)

    def gbs(nbytes, ms):
        return nbytes / (ms * 1e-3) / 1e9 if ms == ms and ms > 0 else None

    # HBM traffic of the dominant kernel from the separate rocprofv3 --pmc passes (tools/gpu_check.sh
    # prof; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), if a summary is committed
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_resblock_fwd.json")
    if os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    ach = gbs(fwd_b, fwd_ms)
    copy_gbs = measured_copy_gbs() if rank == 0 else None
    # Per-kernel split of the backward stack, from 3 extra (untimed) steps with one HIP event per launch
    # (the events would cost ~1 % inside the timed region): the block kernel and the data-gradient product.
    bwd_kernels = None
    if rank == 0 and world == 1 and getattr(eng, "_use_ms", lambda: False)():
        eng.fine_marks, eng.marks = True, []
        for _ in range(3):
            x = eng.onehot(piece, scrambled=True)
            eng.loss_and_grad(x, target)
        torch.cuda.synchronize()
        m2, eng.marks, eng.fine_marks = eng.marks, None, False
        tot, cnt = {}, {}
        for (n0, e0), (n1, e1) in zip(m2[:-1], m2[1:]):
            if n1 in ("b_block", "b_dx"):
                tot[n1] = tot.get(n1, 0.0) + e0.elapsed_time(e1)
                cnt[n1] = cnt.get(n1, 0) + 1
        dil, off = CFG["dilations"], [1]
        for d in dil:
            off.append(off[-1] + d)
        Wc = T - off[-1]                                          # columns of the skip crop
        ch = 64
        blk_b = sum(4 * B_LOCAL * ch * ((T - off[i]) + (T - off[i + 1]) + Wc + 2 * (T - off[i + 1])) for i in range(len(dil))) / len(dil)
        dx_b = sum(4 * B_LOCAL * ch * (2 * (T - off[i + 1]) + (T - off[i + 1]) + (T - off[i])) for i in range(len(dil))) / len(dil)
        bwd_kernels = []
        for key, name, nbytes, what in (("b_block", "resblock_bwd_rw_k", blk_b, "x, dy, dz-crop in; [df;dg] out (+ 20 MB of weight-gradient slabs, not counted)"),
                                        ("b_dx", "chan_gemm_rw_k", dx_b, "[df;dg], dy in; dx out")):
            if cnt.get(key):
                ms = tot[key] / cnt[key]
                bwd_kernels.append({"kernel": name, "avg_launch_ms": ms, "bytes_per_launch": nbytes, "bytes": what,
                                    "achieved": gbs(nbytes, ms), "unit": "GB/s",
                                    "frac": gbs(nbytes, ms) * 1e9 / HBM_PEAK,
                                    "frac_of_measured_copy": (gbs(nbytes, ms) / copy_gbs) if copy_gbs else None})
    out = {
        "metric": "audio samples/sec trained (whole node), 30-layer WaveNet @16kHz",
        "value": world * B_LOCAL * T * args.steps / dt,
        "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (f16/bf16 2-term split operands, 3 MFMA per product, f32 accumulate)"
                 if args.precision == "f16x3,bf16x3" else args.precision,
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 30-layer (3x dilations 1..512) WaveNet, 64 res/dil, 256 skip, "
                               "batch 8x16000 per GPU, full train step (one-hot + fwd + CE + bwd + all-reduce + Adam)",
                   "global_batch": world * B_LOCAL, "seq_len": T, "parallelism": "dp%d" % world,
                   "precision": args.precision, "final_loss": float(loss.item())},
        # dominant kernel family = the dilated-conv stack: forward kernel, one launch per block
        "roofline": {"bound": "hbm", "kernel": "resblock_fwd_nt_k (dilated-conv stack forward, %d launches/step)" % n_layers,
                     "achieved": ach, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": (ach * 1e9 / HBM_PEAK) if ach else None, "traffic": traffic,
                     "algorithmic_bytes_per_launch": fwd_b / n_layers,
                     "avg_launch_ms": fwd_ms / n_layers,
                     # the nominal peak next to what a plain device copy reaches on this box
                     # (read + write bytes of a 1 GiB float4 copy / its time)
                     "measured_copy_GBs": copy_gbs,
                     "frac_of_measured_copy": (ach / copy_gbs) if (ach and copy_gbs) else None},
        # the same stack, backward (resblock_bwd_rw_k + chan_gemm_rw_k per block) and forward+backward
        "roofline_stack_bwd": {"bound": "hbm", "achieved": gbs(bwd_b, bwd_ms), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": (gbs(bwd_b, bwd_ms) * 1e9 / HBM_PEAK) if gbs(bwd_b, bwd_ms) else None,
                               "algorithmic_bytes_per_step": bwd_b},
        "roofline_stack_fwd_bwd": {"bound": "hbm", "achieved": gbs(fwd_b + bwd_b, fwd_ms + bwd_ms), "peak": HBM_PEAK / 1e9,
                                   "unit": "GB/s",
                                   "frac": (gbs(fwd_b + bwd_b, fwd_ms + bwd_ms) * 1e9 / HBM_PEAK)
                                   if gbs(fwd_b + bwd_b, fwd_ms + bwd_ms) else None,
                                   "algorithmic_bytes_per_step": fwd_b + bwd_b},
        "phase_ms_per_step": {k: round(v, 4) for k, v in phase.items()},
    }
    if bwd_kernels:
        # per launch, kernel-level bytes ([df;dg] goes through HBM); avg_launch_ms is the time between two HIP events
        # around the lone launch (3 untimed steps), ~10 % above its back-to-back time inside the timed region
        out["roofline_bwd_kernels"] = bwd_kernels
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
