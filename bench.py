#!/usr/bin/env python3
"""Headline benchmark: audio samples/sec trained, 30-layer WaveNet @16 kHz (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed, backend nccl = RCCL); weak scaling, 8 clips x 16000 samples
per GPU.  One "step" = H2D of the int32 codes + int64 targets (prefetched) + forward on the loader-layout one-hot of
those codes (not materialised: gather / scatter in the causal layer) + CrossEntropy-on-probabilities + backward +
gradient all-reduce (N > 1) + Adam, i.e. wavenet/train.py:171-182 of the reference.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
    # --standalone: torchrun picks AND holds the rendezvous port itself (a bind / close / hand-over of a "free" port
    # can lose it to another process in between)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(known.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)                                  # (inherited as is: the pool exports what RCCL needs across processes)
    # host threads per rank = this host's CPU quota shared by the ranks (music_amd/_lib.py: thread_budget; torchrun's own default
    # is 1).  _lib imports neither torch nor the library at module level: the parent stays off the GPU
    from music_amd import _lib as _wl
    env.setdefault("OMP_NUM_THREADS", str(min(8, _wl.thread_budget(_wl.cpu_quota(), known.gpus))))
    sys.exit(subprocess.call(cmd, env=env))


if __name__ == "__main__":
    _self_launch()

import numpy as np  # noqa: E402
import torch  # noqa: E402

CFG = dict(filter_width=2, dilations=[2 ** i for i in range(10)] * 3, dilation_channels=64,
           residual_channels=64, skip_channels=256, quantization_channels=256, use_bias=False)
B_LOCAL, T = 8, 16000
MARK_EVERY = 4             # timed region: HIP events of the stack kernels on every 4th step (an event costs a marker packet)
HBM_PEAK = 8.0e12          # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec); the line reports the box's own copy rate beside it (5.0 - 5.3e12: roofline.measured_copy_GBs)
MFMA_PEAK = 2.5e15         # FLOP/s, dense bf16 / f16 MFMA (same guide)
BWD_KERNELS = "resblock_bwd_pq_k"
BWD_PMC = "resblock_bwd_pq_k<"                       # every form of the backward block (pair / chain, with / without dy)


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


def enc_stack_bytes(dil, r, d, b, t):
    """Algorithmic bytes of the autoencoder's ENCODER stack (wavenet_autoencoder/model1.py:137-152), fp32, layer at a
    time, every tensor touched once - the same accounting as SURVEY 8(d) uses for the decoder stack.  Forward per
    block: read x_i, write x_{i+1}, write the pre-activation h_i (the backward needs it); backward: read dx_{i+1}, x_i
    (ReLU mask and weight-gradient operand) and h_i, write dx_i."""
    L = [t - 1]
    for x in dil:
        L.append(L[-1] - x)
    fwd = 4 * sum(r * L[i] + r * L[i + 1] + d * L[i + 1] for i in range(len(dil)))
    bwd = 4 * sum(r * L[i + 1] + r * L[i] + d * L[i + 1] + r * L[i] for i in range(len(dil)))
    return fwd * b, bwd * b


def c4_traffic(df, db_, ef, eb_):
    """PMC bytes per launch of config 4's four stack kernels (profiles/pmc_kernels.json, "c4:" entries) beside the algorithmic
    bytes per launch; None when there is no summary for these kernel sources."""
    path = os.path.join(ROOT, "profiles", "pmc_kernels.json")
    try:
        pmc = json.load(open(path))
    except Exception:
        return None
    if pmc.get("_csrc_sha16") not in (None, csrc_sha()):
        return None
    n = len(CFG["dilations"])

    def pick(*must):
        hit = [(v["hbm_bytes_per_launch"], v.get("launches", 0)) for k, v in pmc.items()
               if k.startswith("c4:") and isinstance(v, dict) and all(m in k for m in must)]
        tot = sum(c for _, c in hit)
        return (sum(b * c for b, c in hit) / tot) if tot else None
    out = {"decoder_fwd": {"kernel": "resblock_fwd_nt_k<.., CND>", "hbm_bytes": pick("resblock_fwd_nt_k", "false, 8, true"), "algorithmic_bytes": df / n},
           "decoder_bwd": {"kernel": "resblock_bwd_pq_k<.., COND, ..>", "hbm_bytes": pick("resblock_bwd_pq_k", ", true, true, false>"), "algorithmic_bytes": db_ / n},
           "encoder_fwd": {"kernel": "resblock_fwd_nt_k<.., ENC>", "hbm_bytes": pick("resblock_fwd_nt_k", "true, 8, false"), "algorithmic_bytes": ef / n},
           "encoder_bwd": {"kernel": "enc_bwd_pq_k", "hbm_bytes": pick("enc_bwd_pq_k"), "algorithmic_bytes": eb_ / n}}
    return out if any(v["hbm_bytes"] for v in out.values()) else None


def host_cores():
    """(logical CPUs, physical cores) of this host, from /proc/cpuinfo where it says."""
    logical = os.cpu_count() or 1
    phys = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return logical, (len(phys) if phys else logical)


def flops_per_clip(dil, r, d, s, q, t):
    """SURVEY 8(d) algorithmic FLOPs per clip: (stack forward: f/g products, dense; skip; post-processing; causal)."""
    L = [t - 1]
    for x in dil:
        L.append(L[-1] - x)
    w, lout = L[-1], sum(L[1:])
    fg = 2 * (2 * 2 * r * d) * lout
    dense = 2 * d * r * lout
    skip = 2 * d * s * w * len(dil)
    post = 2 * s * s * w + 2 * s * q * w
    causal = 2 * (2 * q * r) * L[0]
    return dict(fg=fg, dense=dense, skip=skip, post=post, causal=causal)


def cpu_baseline():
    """the CPU oracle (== the reference's ATen-CPU op sequence, pinned by tests/golden) timed on this
    host on the SAME workload shape - full training step (forward + CE + backward + Adam) on 8 clips x 16000 samples,
    2 warm-up + 5 timed steps - plus a one-thread figure on one clip.  Bounded: about 30 s of CPU work."""
    from oracle import wavenet_oracle as wo
    from oracle import intops
    torch.manual_seed(0)
    from music_amd.model import wavenet
    net = wavenet(**CFG)
    params = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-4)
    rng = np.random.default_rng(1234)
    codes = rng.integers(0, 256, size=(B_LOCAL, T + 1))
    x = torch.from_numpy(np.stack([intops.one_hot_scrambled(c[:T]) for c in codes]))
    rf = 3071
    W = T - rf + 1
    target = torch.from_numpy(codes[:, rf:rf + W].astype(np.int64)).reshape(-1)
    logical, physical = host_cores()

    def step(xb, tb):
        t0 = time.time()
        opt.zero_grad()
        loss = wo.ce_on_probs(wo.wavenet_forward(params, CFG["dilations"], xb), tb)
        loss.backward()
        opt.step()
        return time.time() - t0

    # ATen's CPU conv kernels stop scaling (and then collapse) long before a 256-core host is full: one probe step
    # on ONE clip each at all physical cores and at 32 threads (the second of two, the first warms the allocator up),
    # the timed steps run at the faster setting
    # ... and at the container's CPU quota when there is one (cgroup cpu.max: more threads than that only spin and get the
    # process throttled)
    from music_amd import _lib as _wn_lib
    quota = _wn_lib.cpu_quota()
    cand = sorted({min(physical, 256), min(32, physical)} | ({min(quota, physical)} if quota else set()), reverse=True)
    probe = {}
    for n in cand:
        torch.set_num_threads(n)
        step(x[:1], target[:W])
        probe[n] = step(x[:1], target[:W])
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    for _ in range(2):
        step(x, target)                               # 2 warm-up steps at the workload's batch
    times = [step(x, target) for _ in range(5)]
    best = min(times)
    torch.set_num_threads(1)
    t1 = step(x[:1], target[:W])
    # (value = the MEAN of the timed steps, like the GPU figure beside it; the best step is kept as best_value)
    return {"value": B_LOCAL * T * len(times) / sum(times), "unit": "samples/s", "cores": cores, "kind": "port",
            "host_logical_cpus": logical, "host_physical_cores": physical, "host_cpu_quota": quota,
            "probe_s_per_step": {str(k): round(v, 3) for k, v in probe.items()},
            "best_value": B_LOCAL * T / best,
            "one_thread": {"value": T / t1, "unit": "samples/s", "sample": "1 step on 1 clip x 16000, 1 thread"},
            "sample": "full training steps (fwd+CE+bwd+Adam) of the 30-layer config on %d clips x %d samples: thread setting picked "
                      "by a one-clip probe step at each of %s, then 2 warm-up + 5 timed steps at torch CPU threads=%d (mean; the best step in best_value)" %
                      (B_LOCAL, T, cand, cores)}


def sub_benchmarks(net, x, target):
    """Driver-timed figures for BASELINE configs[3] and [4] (extra keys of the JSON line): the autoencoder's fused
    training step at 8 x 16000 and the cached-queue decoder on 16 000 greedy samples, one stream and 128 utterances."""
    out = {}
    dev = x.device
    from music_amd.model1 import wavenet_autoencoder
    torch.manual_seed(0)
    ae = wavenet_autoencoder(filter_width=2, quantization_channel=256, dilations=CFG["dilations"], en_residual_channel=64,
                             en_dilation_channel=64, en_bottleneck_width=64, en_pool_kernel_size=512,
                             de_residual_channel=64, de_dilation_channel=64, de_skip_channel=256, use_bias=False).cuda()
    aeng = ae._engine_for(dev)
    aeng.adam_init(lr=1e-4)

    def ae_step():
        loss = aeng.loss_and_grad(x, target, ae._draw_conditioning())      # fresh conditioning projections per forward
        aeng.adam_step()
        return loss
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < 0.3:                # warm-up: the device's clock is back up after ~7 steps of load (see the timed loop)
        ae_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        loss = ae_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    # phase table + roofline of its two stacks: 3 more steps with HIP events between the phases
    aeng.marks = []
    for _ in range(3):
        ae_step()
    torch.cuda.synchronize()
    m, aeng.marks = aeng.marks, None
    ph = {}
    for (n0, e0), (n1, e1) in zip(m[:-1], m[1:]):
        if n1 != "begin":
            ph[n1] = ph.get(n1, 0.0) + e0.elapsed_time(e1) / 3
    df, db_ = stack_bytes(CFG["dilations"], 64, 64, B_LOCAL, T)
    ef, eb_ = enc_stack_bytes(CFG["dilations"], 64, 64, B_LOCAL, T)
    st_ms = sum(ph.get(k, float("nan")) for k in ("enc_stack_fwd", "dec_stack_fwd", "dec_stack_bwd", "enc_stack_bwd"))
    frac = lambda nbytes, ms: nbytes / (ms * 1e-3) / 8e12 if ms == ms and ms > 0 else None
    out["c4_autoencoder"] = {"workload": "BASELINE configs[3]: autoencoder 30+30 blocks, 64 ch, 256 skip, bottleneck 64, pool 512, "
                                         "batch 8x16000, fused step (fwd + CE + bwd + Adam), 10 steps after 0.3 s of warm-up steps",
                             "ms_per_step": dt * 1e3, "samples_per_s": B_LOCAL * T / dt, "final_loss": float(loss.item()),
                             "phase_ms_per_step": {k: round(v, 4) for k, v in ph.items()},
                             "roofline_stacks": {
                                 "bound": "hbm", "peak": 8000.0, "unit": "GB/s",
                                 "algorithmic_bytes_per_step": {"decoder_fwd": df, "decoder_bwd": db_, "encoder_fwd": ef, "encoder_bwd": eb_},
                                 "frac_decoder_fwd": frac(df, ph.get("dec_stack_fwd", float("nan"))),
                                 "frac_decoder_bwd": frac(db_, ph.get("dec_stack_bwd", float("nan"))),
                                 "frac_encoder_fwd": frac(ef, ph.get("enc_stack_fwd", float("nan"))),
                                 "frac_encoder_bwd": frac(eb_, ph.get("enc_stack_bwd", float("nan"))),
                                 "frac": frac(df + db_ + ef + eb_, st_ms), "stacks_ms": st_ms,
                                 # HBM bytes per launch of the four stack kernels from the config-4 PMC passes (tools/gpu_check.sh prof ->
                                 # profiles/pmc_kernels.json, entries "c4:<kernel>"; FETCH doubled per MI355X_MICROARCH.md), next to the
                                 # algorithmic bytes per launch; null when no committed summary belongs to these kernel sources
                                 "traffic_per_launch": c4_traffic(df, db_, ef, eb_),
                                 "note": "decoder stack: SURVEY 8(d) A_f / A_b; encoder stack: the same accounting (enc_stack_bytes); the "
                                         "conditioning (packed tables, bucket bytes, per-workgroup bucket sums) is not counted"}}
    del ae, aeng
    from music_amd import fast_generate as fg
    start = torch.zeros(1, 256, net.receptive_field, device=dev)
    start[:, 128, :] = 1.0
    fg.generate_codes(net, start, 200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    codes = fg.generate_codes(net, start, 16000)
    torch.cuda.synchronize()
    dt1 = time.perf_counter() - t0
    U = 128
    starts = torch.zeros(U, 256, net.receptive_field, device=dev)
    for u in range(U):
        starts[u, (128 + u) % 256, :] = 1.0
    fg.generate_codes_batch(net, starts, 50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fg.generate_codes_batch(net, starts, 16000)
    torch.cuda.synchronize()
    dtu = time.perf_counter() - t0
    out["c5_decode"] = {"workload": "BASELINE configs[4]: cached-queue greedy decode, 30-layer model, 16000 samples (1 s @16 kHz) "
                                    "from the class-128 start piece, one persistent launch (queue fill included)",
                        "single_stream_samples_per_s": 16000 / dt1, "single_stream_s": dt1,
                        "real_time_factor": 16000 / dt1 / 16000.0,
                        "batch128_samples_per_s": U * 16000 / dtu, "batch128_s": dtu,
                        "distinct_codes": int(torch.unique(codes).numel())}
    return out


def shipped_benchmarks(dev):
    """The reference's SHIPPED parameter files (wavenet/params/wavenet_params.json, wavenet_autoencoder/params/model_params.json:
    40 blocks, 32 / 32 channels, 512 skip channels; bottleneck 512, pool 512), fused training step: the WaveNet at its shipped
    batch 4 x 44093 (wavenet/params/train_params.json + dataset_params.json), the autoencoder at 4 clips of 16384 predicted
    samples.  Not a BASELINE config; these are the shapes a user of the reference starts from (32-channel models run as
    clip pairs on the 64-channel block kernels)."""
    import numpy as np
    from music_amd.model import wavenet
    from music_amd.model1 import wavenet_autoencoder
    out = {}
    dil = [2 ** i for i in range(10)] * 4
    rng = np.random.default_rng(0)

    def timed(step, n):
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < 0.3:            # clocks back up after the lighter work before
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    torch.manual_seed(0)
    net = wavenet(filter_width=2, dilations=dil, dilation_channels=32, residual_channels=32, skip_channels=512,
                  quantization_channels=256, use_bias=False).cuda()
    eng = net._engine_for(dev)
    eng.adam_init(lr=1e-4)
    B, T = 4, 44093
    W = T - net.receptive_field + 1
    codes = torch.from_numpy(rng.integers(0, 256, size=(B, T)).astype(np.int32)).to(dev)
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).to(dev)

    def wn_step():
        eng.loss_and_grad_codes(codes, target, scrambled=True)
        eng.adam_step()
    dt = timed(wn_step, 10)
    out["wavenet"] = {"workload": "wavenet_params.json (40 blocks, 32 / 32 / 512) at 4 x 44093, fused step from codes",
                      "ms_per_step": dt * 1e3, "samples_per_s": B * T / dt, "clip_pairs": bool(eng.workspace(B, T)["pair"])}
    # ... and its cached-queue generation (fast_generate.py): 8000 greedy samples from the class-128 start piece
    from music_amd import fast_generate as fg
    start = torch.zeros(1, 256, net.receptive_field, device=dev)
    start[:, 128, :] = 1.0
    fg.generate_codes(net, start, 200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gen = fg.generate_codes(net, start, 8000)
    torch.cuda.synchronize()
    dtg = time.perf_counter() - t0
    out["wavenet"]["decode_single_stream_samples_per_s"] = 8000 / dtg
    out["wavenet"]["decode_matrix_core_kernels"] = bool(fg._mfma_decode(net._engine))
    out["wavenet"]["decode_distinct_codes"] = int(torch.unique(gen).numel())
    del net, eng
    torch.cuda.empty_cache()
    torch.manual_seed(0)
    ae = wavenet_autoencoder(filter_width=2, quantization_channel=256, dilations=dil, en_residual_channel=32, en_dilation_channel=32,
                             en_bottleneck_width=512, en_pool_kernel_size=512, de_residual_channel=32, de_dilation_channel=32,
                             de_skip_channel=512, use_bias=False).cuda()
    aeng = ae._engine_for(dev)
    aeng.adam_init(lr=1e-4)
    B, W = 4, 16384
    T = ae.receptive_field + W - 1
    codes = torch.from_numpy(rng.integers(0, 256, size=(B, T)).astype(np.int32)).to(dev)
    x = torch.zeros(B, 256, T, device=dev)
    x.scatter_(1, codes.long().unsqueeze(1), 1.0)
    target = torch.from_numpy(rng.integers(0, 256, size=(B * W,)).astype(np.int64)).to(dev)

    def ae_step():
        aeng.loss_and_grad(x, target, ae._draw_conditioning())
        aeng.adam_step()
    dt = timed(ae_step, 10)
    out["autoencoder"] = {"workload": "model_params.json (40 + 40 blocks, 32 / 32, bottleneck 512, pool 512, skip 512) at 4 x %d "
                                      "(16384 predicted samples per clip), fused step, dense one-hot input" % T,
                          "ms_per_step": dt * 1e3, "samples_per_s": B * T / dt, "clip_pairs": bool(aeng.workspace(B, T)["pair"])}
    del ae, aeng
    torch.cuda.empty_cache()
    return out


def surface_benchmarks(net, eng, piece, target):
    """What a caller of the REFERENCE SURFACE gets: the loop of wavenet/train.py:171-182 as written -
    optimizer.zero_grad(), net(x), nn.CrossEntropyLoss on the probabilities, backward(), torch.optim.Adam.step() - at
    8 x 16000, on (a) the loader's one-hot (carries its codes: the causal layer runs on them) and (b) a plain dense
    (B, 256, T) float tensor; and the fused engine step captured in ONE hipGraph next to its eager time."""
    out = {}
    B, W = piece.shape[0], target.numel() // piece.shape[0]
    ce = torch.nn.CrossEntropyLoss()
    # the reference builds its optimizer through train.get_optimizer (wavenet/train.py:28-42,138-139); so does this loop, through
    # the drop-in module's: a torch.optim.Adam whose step() is one launch on the flat buffers (music_amd/train.py FlatAdam)
    from music_amd import train as wtrain
    opt = wtrain.get_optimizer(net, "adam", 1e-4, 0.9)
    x_tag = eng.onehot(piece, scrambled=True)
    x_plain = x_tag.clone()                                    # a copy loses the tag: the dense path

    def ref_step(x):
        opt.zero_grad()
        loss = ce(net(x), target)
        loss.backward()
        opt.step()
        return loss
    for name, x in (("loader_onehot", x_tag), ("dense_tensor", x_plain)):
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < 0.3:
            ref_step(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            loss = ref_step(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        out[name] = {"ms_per_step": dt * 1e3, "samples_per_s": B * piece.shape[1] / dt, "loss": float(loss.item())}
    out["workload"] = ("the reference's own training loop (wavenet/train.py:171-182: zero_grad, net(x), CrossEntropyLoss on the "
                       "probabilities, backward, step of train.get_optimizer(net, 'adam', ..)) on this nn.Module at 8 x 16000, input "
                       "resident, 10 steps after 0.3 s of warm-up steps; the CrossEntropyLoss call is intercepted on the module's output and runs as the "
                       "engine's fused softmax + CE pass (net.fuse_loss; torch's own kernels: +0.45 ms per step, DESIGN.md)")
    del x_tag, x_plain, opt

    def fused():
        loss = eng.loss_and_grad_codes(piece, target, scrambled=True)
        eng.adam_step()
        return loss
    for _ in range(3):
        fused()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fused()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20
    g = torch.cuda.CUDAGraph()
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s_):
            fused()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s_):
                fused()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        graphed = (time.perf_counter() - t0) / 20
        out["hipgraph_step"] = {"eager_ms": eager * 1e3, "graph_replay_ms": graphed * 1e3,
                                "note": "the fused step (resident batch, no H2D) eager vs captured in one hipGraph (main + side streams); the "
                                        "replay serialises what the two streams overlap - the step is GPU-bound, not launch-bound; Adam's "
                                        "bias-correction scalars are frozen in the capture (timing only)"}
    except Exception as e:                                     # capture is a diagnostic: never fail the bench on it
        out["hipgraph_step"] = {"eager_ms": eager * 1e3, "error": repr(e)[:200]}
    return out


def csrc_sha():
    """sha256[:16] over the kernel sources (music_amd/csrc/*.hip, *.h and the public header), in name order."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "music_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))) + [os.path.join(ROOT, "include", "wavenet_hip.h")]:
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


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
    ap.add_argument("--steps", type=int, default=50)          # discard 10 warm-up steps, time >= 50
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle", type=int, default=60, help="discarded steps before the warm-up (one-time costs of a fresh box)")
    ap.add_argument("--dump-steps", action="store_true", help="every timed step's GPU time in the line (ms_per_step_stats.all_in_order)")
    ap.add_argument("--precision", default="f16x3,bf16x3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-4 / config-5 sub-benchmarks")
    ap.add_argument("--phases", action="store_true", help="print the per-phase GPU time table to stderr")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch plumbing only (CPU, gloo): every rank joins the group, rank 0 prints the world size")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to fd 1 when the process group comes
    # up (and libraries may print more), so fd 1 points at stderr until the line is written
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
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
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "rank_sum": t.item(), "omp_num_threads": os.environ.get("OMP_NUM_THREADS")}),
                  flush=True)
        return
    backend = os.environ.get("WN_DIST_BACKEND", "nccl")       # gloo: several ranks on one GPU (tests on a 1-GPU box)
    n_dev = max(1, torch.cuda.device_count())
    if backend == "gloo":
        local = local % n_dev
    elif local >= n_dev:
        raise SystemExit("bench.py: LOCAL_RANK %d but %d GPU(s) visible: one rank per GPU under RCCL" % (local, n_dev))
    torch.cuda.set_device(local)
    dist = None
    # under torchrun (RANK/MASTER_ADDR set) the process group is always created, also for 1 rank,
    # so that the RCCL path of an N-GPU run is the path a 1-rank launch exercises
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from music_amd.model import wavenet
    torch.manual_seed(0)                               # identical replicas on every rank
    net = wavenet(**CFG)
    net.precision = tuple(args.precision.split(","))
    net = net.cuda()
    dev = torch.device("cuda", local)
    eng = net._engine_for(dev)
    eng.adam_init(lr=1e-4)
    codes = synth_codes(rank, B_LOCAL, T)
    rf = net.receptive_field
    W = T - rf + 1
    # what a loader hands over per step: int32 sample codes and int64 targets in pinned host memory (1.3 MB); the
    # H2D copies are part of the step, the one-hot is built in HBM
    piece_h = codes[:, :T].contiguous().cpu().pin_memory()
    target_h = codes[:, rf:rf + W].to(torch.int64).contiguous().view(-1).cpu().pin_memory()
    # double-buffered device copies, filled one step ahead on a copy stream (what a prefetching loader does): the
    # copies of step k+1 run beside the kernels of step k, the step waits for its own batch's event
    main = torch.cuda.current_stream()
    copy_stream = torch.cuda.Stream(device=dev)
    bufs = [(torch.empty(B_LOCAL, T, dtype=torch.int32, device=dev), torch.empty(B_LOCAL * W, dtype=torch.int64, device=dev),
             torch.cuda.Event(), torch.cuda.Event()) for _ in range(2)]
    state = {"k": 0}

    def prefetch(k):
        p, t, ready, free = bufs[k & 1]
        copy_stream.wait_event(free)                   # the step that last used this pair is done with it
        with torch.cuda.stream(copy_stream):
            p.copy_(piece_h, non_blocking=True)
            t.copy_(target_h, non_blocking=True)
            ready.record(copy_stream)

    for b in bufs:
        b[3].record(main)
    prefetch(0)

    def step():
        k = state["k"]
        state["k"] = k + 1
        piece, target, ready, free = bufs[k & 1]
        prefetch(k + 1)
        if state.get("sampled") is not None:            # timed region: events on every MARK_EVERY-th step only
            eng.marks = state["sampled"] if (k - state["k0"]) % MARK_EVERY == 0 else None
        eng.mark("step_begin")
        main.wait_event(ready)
        eng.mark("h2d_wait")
        # the loader-faithful (scrambled) one-hot of these codes is what the model sees; it is never
        # materialised: the causal layer runs as a gather / scatter on the codes (engine.loss_and_grad_codes)
        loss = eng.loss_and_grad_codes(piece, target, scrambled=True)
        free.record(main)
        if use_dist:
            dist.all_reduce(eng.flat_grad)          # ONE flat fp32 bucket (5.08 MB), RCCL over xGMI
            eng.mark("allreduce")
        eng.adam_step(gscale=1.0 / world)
        return loss

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Settle phase, before the W warm-up steps: the first GPU process on a fresh box pays one-time costs well past its
    # tenth step (code-object loads, workspace allocation, the first timing events of the sampled marks, the copy stream's
    # first transfers: 25-step windows read 5.3-5.7, then 4.24 ms from the second window on, tools/settle_probe.py), and
    # the metric is STEADY-STATE throughput.  The settle steps run the timed loop's own code path, sampled
    # marks included, and are discarded; the contract's W untimed + K timed steps follow unchanged.
    state["sampled"], state["k0"] = [], state["k"]
    eng.mark_only = {"step_begin", "causal_fwd", "stack_fwd", "epilogue_bwd", "stack_bwd"}
    # ... but the one-time cost stays visible: the first window of the process (min(25, settle) steps, wall clock) is reported
    first_n = min(25, args.settle)
    first_ms = None
    import gc
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    gc.collect()
    gc.disable()
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    for i_ in range(args.settle):
        loss = step()
        if i_ + 1 == first_n:
            torch.cuda.synchronize()
            first_ms = (time.perf_counter() - t_first) / first_n * 1e3
    barrier()
    state["sampled"], eng.marks, eng.mark_only = None, None, None
    # everything the host has to do besides launching went IN FRONT of the settle steps (above): a device left idle for milliseconds between
    # the warm-up and the timed steps lowers its clock and takes ~7 steps to raise it again (5.19, 4.80, 4.63, 4.56, 4.49, 4.39, 4.36, then
    # 4.25 ms in one run whose collector pass sat between the two).  The interpreter's collector runs here, not inside the timed region
    # (the host issues ~110 launches per step, 0.74 ms of its time per 4.3 ms step: a collector pause shows up as one long step)
    for _ in range(args.warmup):
        loss = step()
    barrier()
    # inside the timed region only the events the roofline figures need (the two ends of the forward and of the backward
    # stack), and only on every MARK_EVERY-th step: a timing event is a marker packet between two dependent kernels, and
    # five of them per step cost 0.12-0.17 ms of a 4.6 ms step (same-box A/B).  The full phase table comes from 3 untimed
    # steps afterwards.
    state["sampled"], state["k0"] = [], state["k"]
    eng.mark_only = {"step_begin", "causal_fwd", "stack_fwd", "epilogue_bwd", "stack_bwd"}
    # one event per step end (enable_timing): median / min / max of the per-step GPU time
    t0 = time.perf_counter()
    step_ev[0].record()
    host_t = [0.0] * args.steps                              # when the host had a step enqueued (ms after t0): one float store per step
    for i_ in range(args.steps):
        loss = step()
        step_ev[i_ + 1].record()
        host_t[i_] = time.perf_counter()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    per_step_order = [round(step_ev[i_].elapsed_time(step_ev[i_ + 1]), 3) for i_ in range(args.steps)]
    per_step = sorted(per_step_order)
    marks, eng.marks, eng.mark_only = state["sampled"], None, None
    state["sampled"] = None
    n_sampled = (args.steps + MARK_EVERY - 1) // MARK_EVERY
    dt_ranks = [dt]
    if use_dist:
        # every rank's own wall time over the K steps (what a straggler looks like in the line), then the contract's MAX over ranks
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        tg = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(tg, tt)
        dt_ranks = [float(t.item()) for t in tg]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()

    # per-phase GPU time from the HIP events recorded on the launch stream inside the timed region
    phase, phase_all = {}, {}
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        if n1 != "step_begin":
            dt_ = e0.elapsed_time(e1)
            phase[n1] = phase.get(n1, 0.0) + dt_
            phase_all.setdefault(n1, []).append(dt_)
    phase = {k: v / n_sampled for k, v in phase.items()}       # ms per step (mean of the sampled steps)
    # ... and the MEDIAN over the sampled steps beside it: the device inserts steps of 6 ms, and one of them among
    # the 13 sampled steps of a 50-step run moves a phase's mean by 5 - 10 %
    phase_median = {k: sorted(v)[len(v) // 2] for k, v in phase_all.items()}
    # with the coarse marks: "causal_fwd" = everything from the step's begin to the causal layer's end, "epilogue_bwd" =
    # epilogue forward + softmax/CE + epilogue backward; stack_fwd / stack_bwd are exact
    timed = {"stack_fwd": phase.get("stack_fwd"), "stack_bwd": phase.get("stack_bwd"),
             "begin_to_stack": phase.get("causal_fwd"), "between_stacks": phase.get("epilogue_bwd")}
    # how long the HOST takes to enqueue a step (device idle at the start, nothing waited for inside): the margin by which the
    # interpreter runs ahead of the device.  Timed steps that take longer than the median are host stalls when this is close to it
    barrier()
    depth, eng._throttle.depth = eng._throttle.depth, 0      # (the engine keeps at most WN_MAX_STEPS_IN_FLIGHT steps in flight - _lib.StepThrottle -; this measures the enqueue alone)
    t_h = time.perf_counter()
    for _ in range(8):
        step()
    host_enqueue_ms = (time.perf_counter() - t_h) / 8 * 1e3
    eng._throttle.depth = depth
    del eng._throttle._ev[:]
    barrier()
    # full phase table: 3 untimed steps with every mark
    eng.marks = []
    for _ in range(3):
        step()
    barrier()
    m3, eng.marks = eng.marks, None
    full = {}
    for (n0, e0), (n1, e1) in zip(m3[:-1], m3[1:]):
        if n1 != "step_begin":
            full[n1] = full.get(n1, 0.0) + e0.elapsed_time(e1)
    full = {k: v / 3 for k, v in full.items()}
    full["stack_fwd"], full["stack_bwd"] = phase.get("stack_fwd", float("nan")), phase.get("stack_bwd", float("nan"))
    phase = full
    if args.phases and rank == 0:
        print(json.dumps({"phase_ms_per_step": phase, "timed_region": timed}), file=sys.stderr)

    dil = CFG["dilations"]
    fwd_b, bwd_b = stack_bytes(dil, 64, 64, B_LOCAL, T)
    n_layers = len(dil)
    nan = float("nan")
    fwd_ms, bwd_ms = phase.get("stack_fwd", nan), phase.get("stack_bwd", nan)

    def gbs(nbytes, ms):
        return nbytes / (ms * 1e-3) / 1e9 if ms == ms and ms > 0 else None

    def roof(nbytes, ms, **extra):
        g = gbs(nbytes, ms)
        r = {"bound": "hbm", "achieved": g, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": (g * 1e9 / HBM_PEAK) if g else None}
        r.update(extra)
        return r

    # HBM traffic per launch from the separate rocprofv3 --pmc passes of the same command (tools/gpu_check.sh prof ->
    # profiles/pmc_kernels.json; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), when a summary is committed
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_kernels.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
        except Exception:
            pmc = {}

    # the committed PMC summary belongs to the kernel sources it was measured on: a hash of music_amd/csrc travels with it,
    # and a figure measured on other sources is dropped (traffic: null) instead of being carried along
    src_sha = csrc_sha()
    pmc_stale = bool(pmc) and pmc.get("_csrc_sha16") not in (None, src_sha)
    if pmc_stale:
        pmc = {}

    def traffic_of(*prefixes):
        tot = 0.0
        for pre in prefixes:
            # (entries with a "source" were measured on another workload - the config-4 step - and are not this run's kernels)
            hit = [v["hbm_bytes_per_launch"] for k, v in pmc.items() if k.startswith(pre) and isinstance(v, dict) and "source" not in v]
            if not hit:
                return None
            tot += max(hit)
        return tot

    def traffic_mean(prefix):
        """launch-weighted mean over every instantiation of a kernel (the backward block runs in several forms per step)"""
        hit = [(v["hbm_bytes_per_launch"], v.get("launches_per_step", 1)) for k, v in pmc.items()
               if k.startswith(prefix) and isinstance(v, dict) and "source" not in v]
        n = sum(c for _, c in hit)
        return sum(b * c for b, c in hit) / n if n else None

    copy_gbs = measured_copy_gbs() if rank == 0 else None
    # Per-kernel split of the backward stack, from 3 extra (untimed) steps with one HIP event per launch
    # (the events would cost ~1 % inside the timed region)
    kern = None
    if rank == 0 and world == 1:
        eng.fine_marks, eng.marks = True, []
        torch.cuda.synchronize()
        piece, target = bufs[0][0], bufs[0][1]
        for _ in range(3):
            eng.loss_and_grad_codes(piece, target, scrambled=True)
        torch.cuda.synchronize()
        m2, eng.marks, eng.fine_marks = eng.marks, None, False
        tot, cnt = {}, {}
        for (n0, e0), (n1, e1) in zip(m2[:-1], m2[1:]):
            tot[n1] = tot.get(n1, 0.0) + e0.elapsed_time(e1)
            cnt[n1] = cnt.get(n1, 0) + 1
        off = [1]
        for d in dil:
            off.append(off[-1] + d)
        Wc, ch = T - off[-1], 64
        per = {k: tot[k] / cnt[k] for k in tot}
        kern = {"fine_ms_per_launch": {k: round(v, 5) for k, v in per.items() if k.startswith(("b_", "f_"))}}

    fl = flops_per_clip(dil, 64, 64, 256, 256, T)
    B = B_LOCAL
    issued = {      # x3: every product is hi*hi + lo*hi + hi*lo on the 16-bit matrix cores
        "stack_fwd": 3 * B * (fl["fg"] + fl["dense"]),
        "stack_bwd": 3 * B * (3 * fl["fg"] + 2 * fl["dense"]),          # recompute, dz, dx, dWfg, dWd
        "epilogue_fwd": 3 * B * (fl["skip"] + fl["post"]),
        "epilogue_bwd": 3 * B * 2 * (fl["skip"] + fl["post"]),
        "causal_fwd": 3 * B * fl["causal"],
    }
    x3 = args.precision == "f16x3,bf16x3"
    mfma = {"peak_TFLOPs": MFMA_PEAK / 1e12, "unit": "fraction of the dense 16-bit MFMA peak",
            "note": "issued MFMA FLOPs = 3 x algorithmic (2-term operand split) / phase time / peak" if x3 else
                    "issued MFMA FLOPs = algorithmic / phase time / peak"}
    tot_fl = 0.0
    for k, v in issued.items():
        v = v if x3 else v / 3
        tot_fl += v
        if phase.get(k):
            mfma[k] = v / (phase[k] * 1e-3) / MFMA_PEAK
    mfma["step"] = tot_fl / (dt / args.steps) / MFMA_PEAK
    mfma["issued_TFLOP_per_step"] = tot_fl / 1e12

    bwd_launch_ms = bwd_ms / n_layers
    out = {
        "metric": "audio samples/sec trained (whole node), 30-layer WaveNet @16kHz",
        "value": world * B_LOCAL * T * args.steps / dt,
        "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": args.settle,
        "ms_per_step": dt / args.steps * 1e3,
        "ms_per_step_stats": {"median": per_step[len(per_step) // 2], "min": per_step[0], "max": per_step[-1],
                              "p90": per_step[min(len(per_step) - 1, (len(per_step) * 9) // 10)], "first_10_in_order": per_step_order[:10], **({"all_in_order": per_step_order, "host_enqueued_at_ms": [round((t - t0) * 1e3, 2) for t in host_t]} if args.dump_steps else {}),
                              "note": "GPU time between the end-of-step HIP events of consecutive timed steps (rank 0); the interpreter's garbage collector runs before the timed region, not inside it"} if per_step else None,
        "ms_per_step_ranks": {"min": min(dt_ranks) / args.steps * 1e3, "max": max(dt_ranks) / args.steps * 1e3,
                              "all": [round(v / args.steps * 1e3, 4) for v in dt_ranks],
                              "note": "each rank's wall time over the K timed steps / K; ms_per_step is the max (the contract)"},
        "host_enqueue_ms_per_step": host_enqueue_ms,
        "host_threads": {"torch_intra_op": torch.get_num_threads(), "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"),
                         "local_world_size": int(os.environ.get("LOCAL_WORLD_SIZE", "1"))},
        "first_window": {"steps": first_n, "ms_per_step": first_ms,
                         "note": "the first steps of this process (one-time costs included), before the settle / warm-up steps are discarded"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (forward + recompute: f16 2-term split operands, 2^-22 per element; backward products: bf16 2-term split, 2^-17 per element, f32 range; 3 MFMA per product, f32 accumulate)" if x3 else args.precision,
        "data": "synthetic",
        "rccl_ranks": world if (use_dist and dist.get_backend() == "nccl") else 0, "backend": (dist.get_backend() if use_dist else "none"),
        "config": {"workload": "BASELINE configs[1]: 30-layer (3x dilations 1..512) WaveNet, 64 res/dil, 256 skip, "
                               "batch 8x16000 per GPU, full train step (H2D of codes and targets, prefetched one step ahead on a copy stream, + fwd on the loader-layout one-hot of the codes (never materialised) + CE + bwd + all-reduce + Adam)",
                   "global_batch": world * B_LOCAL, "seq_len": T, "parallelism": "dp%d" % world,
                   "precision": args.precision, "final_loss": float(loss.item()),
                   # (the driver's record keeps `config`: the median of the K timed steps beside the mean in ms_per_step - the device has
                   # bursts of 6 ms steps on some boxes, profiles/r06_gemm_tiles.md section 3)
                   "ms_per_step_median": per_step[len(per_step) // 2] if per_step else None,
                   "ms_per_step_p90": per_step[min(len(per_step) - 1, (len(per_step) * 9) // 10)] if per_step else None,
                   "slow_steps_over_1p15_median": sum(1 for v in per_step if v > 1.15 * per_step[len(per_step) // 2]) if per_step else None},
        # the time-dominant kernels: one residual block's backward (SURVEY 8d A_b per block; duration = the HIP-event
        # time of the backward stack inside the timed region / its 30 blocks)
        "roofline": roof(bwd_b / n_layers, bwd_launch_ms,
                         kernel="%s: backward of one residual block, %d per step" % (BWD_KERNELS, n_layers),
                         traffic=traffic_mean(BWD_PMC),
                         traffic_source=("profiles/pmc_kernels.json was measured on other kernel sources (hash mismatch): dropped" if pmc_stale else
                                         "profiles/pmc_kernels.json: the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an EARLIER run of "
                                         "this command on these kernel sources (committed summary, source hash checked), corrected per "
                                         "MI355X_MICROARCH.md; not measured in this run; mean over the step's %d launches" % n_layers),
                         algorithmic_bytes_per_launch=bwd_b / n_layers,
                         avg_launch_ms=bwd_launch_ms, measured_copy_GBs=copy_gbs,
                         median_step={"avg_launch_ms": phase_median.get("stack_bwd", nan) / n_layers,
                                      "frac": (gbs(bwd_b / n_layers, phase_median.get("stack_bwd", nan) / n_layers) or nan) * 1e9 / HBM_PEAK,
                                      "note": "the same figure from the MEDIAN of the sampled steps' backward-stack times (a burst step among "
                                              "the sampled ones moves the mean, which `frac` is computed from as the contract asks)"},
                         frac_of_measured_copy=(gbs(bwd_b / n_layers, bwd_launch_ms) / copy_gbs) if copy_gbs and bwd_ms == bwd_ms else None),
        # the dilated-conv stack of the north star, forward / backward / both (SURVEY 8d A_f, A_b)
        "roofline_stack_fwd": roof(fwd_b, fwd_ms, kernel="resblock_fwd_nt_k x %d" % n_layers, traffic=traffic_of("resblock_fwd_nt_k"),
                                   algorithmic_bytes_per_launch=fwd_b / n_layers, avg_launch_ms=fwd_ms / n_layers,
                                   algorithmic_bytes_per_step=fwd_b),
        "roofline_stack_bwd": roof(bwd_b, bwd_ms, algorithmic_bytes_per_step=bwd_b),
        "roofline_stack_fwd_bwd": roof(fwd_b + bwd_b, fwd_ms + bwd_ms, algorithmic_bytes_per_step=fwd_b + bwd_b),
        "mfma_util": mfma,
        # the same run as predicted samples/s (x W/T), and the dilated-conv stack alone (forward, forward + backward)
        "predicted_samples_per_s": world * B_LOCAL * W * args.steps / dt,
        "stack_only": {"fwd_ms": fwd_ms, "fwd_bwd_ms": fwd_ms + bwd_ms,
                       "fwd_bwd_input_samples_per_s_per_gpu": B_LOCAL * T / ((fwd_ms + bwd_ms) * 1e-3) if fwd_ms == fwd_ms else None,
                       "target_fwd_bwd_ms_at_40pct_of_hbm": (fwd_b + bwd_b) / (0.4 * HBM_PEAK) * 1e3},
        # stack_fwd / stack_bwd: HIP events inside the timed region; the other phases: 3 untimed steps with every mark
        "phase_ms_per_step": {k: round(v, 4) for k, v in phase.items()},
        "timed_region_ms_per_step": {k: (round(v, 4) if v is not None else None) for k, v in timed.items()},
        "timed_region_events": "HIP events on the launch stream around the forward and the backward stack, on every %d-th of the "
                               "timed steps (%d of %d); means over those steps" % (MARK_EVERY, n_sampled, args.steps),
    }
    if kern:
        out["kernels"] = kern
    # the side measurements must never cost the headline line: a failure is recorded in place of the figure
    if rank == 0 and world == 1 and not args.no_extras:
        out["extra"] = {}
        try:
            x = eng.onehot(bufs[0][0], scrambled=True)
            out["extra"] = sub_benchmarks(net, x, bufs[0][1])
            del x
        except Exception as e:
            out["extra"]["error"] = "%s: %s" % (type(e).__name__, e)
        try:
            out["extra"]["reference_surface_step"] = surface_benchmarks(net, eng, bufs[0][0], bufs[0][1])
        except Exception as e:
            out["extra"]["reference_surface_step"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            out["extra"]["shipped_params"] = shipped_benchmarks(torch.device("cuda", 0))
        except Exception as e:
            out["extra"]["shipped_params"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline()
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": 0, "kind": "port", "sample": "failed",
                                   "error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        # LAST key, short: the driver keeps the contract keys and the last 2 KB of this line - the side configurations' figures at a glance
        # (each also sits, with its workload text, under `extra`)
        def dig(d, *ks):
            for k in ks:
                d = d.get(k) if isinstance(d, dict) else None
            return round(d, 4) if isinstance(d, float) else d
        ex = out.get("extra", {})
        out["summary"] = {
            "c2_ms_per_step": round(out["ms_per_step"], 4), "c2_ms_per_step_median": dig(out, "config", "ms_per_step_median"),
            "c2_stack_fwd_bwd_frac_of_8TBs": dig(out, "roofline_stack_fwd_bwd", "frac"), "c2_stack_fwd_frac": dig(out, "roofline_stack_fwd", "frac"),
            "c2_bwd_block_frac": dig(out, "roofline", "frac"),
            "c4_autoencoder_ms_per_step": dig(ex, "c4_autoencoder", "ms_per_step"), "c4_samples_per_s": dig(ex, "c4_autoencoder", "samples_per_s"),
            "c4_frac_decoder_fwd": dig(ex, "c4_autoencoder", "roofline_stacks", "frac_decoder_fwd"),
            "c4_frac_decoder_bwd": dig(ex, "c4_autoencoder", "roofline_stacks", "frac_decoder_bwd"),
            "c4_frac_encoder_fwd": dig(ex, "c4_autoencoder", "roofline_stacks", "frac_encoder_fwd"),
            "c4_frac_encoder_bwd": dig(ex, "c4_autoencoder", "roofline_stacks", "frac_encoder_bwd"),
            "c5_decode_samples_per_s": dig(ex, "c5_decode", "single_stream_samples_per_s"), "c5_batch128_samples_per_s": dig(ex, "c5_decode", "batch128_samples_per_s"),
            "reference_loop_ms_per_step": dig(ex, "reference_surface_step", "loader_onehot", "ms_per_step"),
            "shipped_wavenet_ms_per_step": dig(ex, "shipped_params", "wavenet", "ms_per_step"),
            "shipped_autoencoder_ms_per_step": dig(ex, "shipped_params", "autoencoder", "ms_per_step"),
            "cpu_baseline_samples_per_s": dig(out, "cpu_baseline", "value"),
        }
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    os.close(stdout_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
