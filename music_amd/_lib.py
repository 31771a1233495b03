"""ctypes binding of libwavenet_hip.so (the C ABI declared in include/wavenet_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WAVENET_HIP_LIB: developer override (kernel experiments built next to the product library)
LIB_PATH = os.environ.get("WAVENET_HIP_LIB") or os.path.join(_HERE, "libwavenet_hip.so")

F16X3, F16X1, BF16X3, BF16X1 = 0, 1, 2, 3
MODE_NAMES = {"f16x3": F16X3, "f16x1": F16X1, "bf16x3": BF16X3, "bf16x1": BF16X1}
CE_NUM_PARTIALS = 1024
ABI_VERSION = 5

_p = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_int64
_f = ctypes.c_float

# name -> argtypes  (every function returns int status except wn_last_error)
SIGNATURES = {
    "wn_version": [],
    "wn_pack_weights": [_p, _p, _p, _i, _i, _p],
    "wn_chan_gemm": [_p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p, _i, _i,
                     _p, _l, _i, _i, _p,
                     _p, _l, _i, _i,
                     _p, _l, _i,
                     _i, _i, _i, _i, _i, _p],
    "wn_skip_epilogue_fwd": [_p, _l, _i, _i, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p],
    "wn_skip_epilogue_bwd": [_p, _l, _i, _p, _p, _l, _i, _p, _p, _p, _l, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "wn_resblock_fwd": [_p, _p, _p, _l, _l, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i,
                        _i, _i, _i, _i, _p, _l, _i, _i, _i, _i, _p, _l, _p, _l, _i, _i, _p],
    "wn_enc_resblock_fwd": [_p, _p, _p, _l, _l, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "wn_enc_resblock_bwd": [_p, _p, _p, _p, _l, _l, _l, _i, _p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _p],
    "wn_enc_resblock_bwd_slabs": [_i, _i, _i],
    "wn_enc_resblock_bwd_pq": [_p, _p, _p, _i, _i, _p, _p, _p, _l, _l, _i, _p, _p, _i, _i, _i, _i, _p, _p, _i, _i, _i, _p],
    "wn_avgpool": [_p, _l, _i, _i, _i, _i, _i, _p, _l, _i, _i, _p],
    "wn_resblock_bwd": [_p, _p, _p, _p, _p, _l, _l, _l, _l, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i,
                        _p, _l, _i, _i, _i, _i, _i, _i, _i, _p],
    "wn_cond_expand": [_p, _l, _i, _i, _i, _i, _i, _i, _i, _p, _l, _i, _i, _p],
    "wn_cond_grad": [_p, _l, _i, _i, _i, _i, _i, _i, _i, _p, _l, _i, _i, _p],
    "wn_avgpool_bwd": [_p, _l, _i, _i, _i, _i, _i, _p, _l, _i, _i, _i, _p],
    "wn_wgrad": [_p, _l, _i, _i, _i, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p, _i,
                 _l, _i, _i, _i, _i, _i, _p],
    "wn_wgrad_slabs": [_i, _i, _i, _i],
    "wn_resblock_bwd_ms": [_p, _p, _p, _p, _l, _l, _l, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p,
                           _p, _l, _i, _i, _i, _i, _i, _i, _i, _p],
    "wn_resblock_bwd_ms_slabs": [_i, _i, _i],
    "wn_resblock_bwd_pq": [_p, _p, _p, _i, _i, _p, _p, _p, _l, _l, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p,
                           _p, _l, _i, _i, _p, _p, _l, _i, _i, _i, _i, _p],
    "wn_resblock_bwd_pq_chain_ok": [_i, _i, _i, _i],
    "wn_resblock_bwd_pq_slabs": [_i, _i, _i, _i, _i],
    "wn_resblock_bwd_pq_chain_items": [_i, _i, _i, _i, _i, _p, _i],
    "wn_resblock_bwd_pq_cond_floats": [_i, _i, _i],
    "wn_resblock_bwd_pq_cond_reduce": [_p, _p, _p, _i, _i, _i, _i, _p, _l, _l, _i, _p],
    "wn_split16": [_p, _p, _p, _l, _i, _p],
    "wn_shift_add": [_p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p],
    "wn_causal_wgrad_codes": [_p, _i, _p, _p, _i, _i, _l, _i, _i, _i, _i, _i, _p, _p],
    "wn_causal_wgrad_codes_slabs": [_i, _i],
    "wn_causal_fwd_codes": [_p, _i, _p, _p, _i, _p, _l, _i, _i, _i, _i, _i, _p],
    "wn_reduce_slabs": [_p, _i, _l, _p, _p, _p],
    "wn_bias_grad": [_p, _l, _i, _i, _i, _i, _i, _i, _p, _p],
    "wn_chunk_softmax256_fwd": [_p, _p, _l, _p],
    "wn_chunk_softmax256_bwd": [_p, _p, _p, _l, _p],
    "wn_chunk_softmax256_ce": [_p, _p, _p, _p, _p, _l, _f, _p],
    "wn_gate_fwd": [_p, _l, _i, _i, _p, _l, _i, _i, _i, _i, _p],
    "wn_gate_bwd": [_p, _l, _i, _i, _p, _l, _p, _l, _i, _i, _i, _i, _p],
    "wn_chunk_softmax_fwd": [_p, _p, _l, _i, _p],
    "wn_chunk_softmax_bwd": [_p, _p, _p, _l, _i, _p],
    "wn_chunk_softmax_ce": [_p, _p, _p, _p, _p, _l, _i, _f, _p],
    "wn_adam_flat": [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _f, _p],
    "wn_sgd_flat": [_p, _p, _p, _l, _f, _f, _f, _i, _p],
    "wn_rmsprop_flat": [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p],
    "wn_coll_available": [],
    "wn_comm_unique_id": [_p],
    "wn_comm_create": [_i, _i, _p, _p],
    "wn_comm_destroy": [_p],
    "wn_allreduce_flat": [_p, _p, _l, _p],
    "wn_gather_grads": [_p, _p, _p, _i, _p],
    "wn_gather_grads2": [_p, _p, _p, _p, _i, _p],
    "wn_onehot": [_p, _p, _i, _i, _i, _i, _p],
    "wn_mulaw_encode_tbl": [_p, _p, _p, _l, _p],
    "wn_mulaw_decode_lut": [_p, _p, _p, _l, _p],
    "wn_mulaw_encode_q": [_p, _p, _i, _p, _l, _p],
    "wn_mulaw_decode_q": [_p, _p, _i, _p, _l, _p],
    "wn_decode": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l,
                  _i, _i, _p, _p],
    "wn_decode_batch": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l,
                        _i, _i, _p, _i, _l, _f, _l, _p],
    "wn_decode_sync_granules": [_i, _i, _i],
    "wn_decode_batch_pk": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l,
                           _i, _i, _p, _i, _l, _f, _l, _p, _l, _l, _l, _l, _l, _l, _p],
}

_lib = None


class WavenetHipError(RuntimeError):
    pass


def cpu_quota():
    """CPUs this process may use per its cgroup's bandwidth limit (cpu.max / cfs_quota_us), or None without one."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, -(-int(q) // int(per)))        # ceil: a quota of 1.5 CPUs keeps two threads
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, -(-q // per))
    except (OSError, ValueError):
        return None


class StepThrottle:
    """At most `depth` fused training steps of one engine in flight.  The host enqueues a step in 0.7 ms and the device takes 4: a loop that
    never reads a result back (the reference reads loss.data[0] every step, wavenet/train.py:182; a benchmark loop does not) is 40 steps
    ahead ten steps after a synchronisation, the HIP runtime's command and kernel-argument pools run full there, and while they do the
    device is handed its work in fits: 8 - 11 of the next 25 steps take 6 ms instead of 4 (round 6).  enter() waits
    for the END of the step issued `depth` steps ago, leave() marks the end of this one; WN_MAX_STEPS_IN_FLIGHT (default 4; 0 = no limit)."""

    def __init__(self):
        self.depth = int(os.environ.get("WN_MAX_STEPS_IN_FLIGHT", "4"))
        self._ev = []

    def enter(self):
        if self.depth > 0 and len(self._ev) >= self.depth:
            self._ev.pop(0).synchronize()

    def leave(self):
        if self.depth > 0:
            import torch
            ev = torch.cuda.Event()
            ev.record()
            self._ev.append(ev)


def local_world_size(env=None):
    """Ranks the launcher started on THIS host (torchrun exports LOCAL_WORLD_SIZE; 1 without a launcher)."""
    env = os.environ if env is None else env
    try:
        return max(1, int(env.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        return 1


def thread_budget(quota, local_world, cpus=None):
    """Host threads ONE rank may run: the container's CPU quota (or, without one, the host's CPUs) shared by the ranks of
    this host, rounded up, at least 1.  8 ranks on a 16-CPU quota -> 2 each (not 16 each: 128 runnable threads on 16 CPUs
    is the throttling stall of DESIGN_HISTORY.md, 68 instead of 10 ms per step)."""
    total = quota if quota is not None else (cpus if cpus is not None else (os.cpu_count() or 1))
    return max(1, -(-int(total) // max(1, int(local_world))))


def respect_cpu_quota():
    """torch sizes its intra-op thread pool by the machine's cores (128 on an MI355X host) and does not look at the
    container's CPU quota; an OpenMP region then wakes all of them, they spin, and the scheduler throttles the WHOLE process
    for the rest of the period - host code that feeds a GPU loses 80 ms out of every 100 (measured: the autoencoder with the
    reference's shipped parameters, 68 instead of 10 ms per step).  Called once when the library is loaded: the pool is
    cut to this rank's SHARE of the quota - quota / LOCAL_WORLD_SIZE, rounded up (one process per GPU: eight ranks share the
    host) - and the change is REPORTED once on stderr, because it changes the host application's intra-op thread count too.
    Without a quota nothing is cut for a single process; under a launcher the share of the host's CPUs applies.
    WN_KEEP_TORCH_THREADS=1 leaves the pool alone."""
    import sys
    import torch
    if os.environ.get("WN_KEEP_TORCH_THREADS", "0") == "1":
        return
    q, lw = cpu_quota(), local_world_size()
    if q is None and lw == 1:
        return
    n = thread_budget(q, lw)
    if torch.get_num_threads() > n:
        print("music_amd: torch intra-op threads %d -> %d (%s shared by %d rank(s) of this host; WN_KEEP_TORCH_THREADS=1 keeps torch's "
              "own setting)" % (torch.get_num_threads(), n, "the container's CPU quota of %d" % q if q is not None else
                                "%d CPUs" % (os.cpu_count() or 1), lw), file=sys.stderr)
        torch.set_num_threads(n)


def load():
    """Load the shared library (once).  Raises WavenetHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WavenetHipError(
            "libwavenet_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C music_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    # torch first: libwavenet_hip.so must bind to the SAME libamdhip64 (HIP runtime, device
    # context, streams) that PyTorch-ROCm has loaded, not to a second copy from /opt/rocm
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError here = header/library out of sync
        fn.argtypes = args
        fn.restype = ctypes.c_int
    lib.wn_decode_sync_granules.restype = ctypes.c_int64
    lib.wn_last_error.argtypes = []
    lib.wn_last_error.restype = ctypes.c_char_p
    if lib.wn_version() != ABI_VERSION:
        raise WavenetHipError("libwavenet_hip.so ABI version %d != expected %d" % (lib.wn_version(), ABI_VERSION))
    _lib = lib
    respect_cpu_quota()
    return lib


def wgrad_slabs(t_lo, t_hi, chunk, batch):
    """Number of slabs one wn_wgrad call writes (plain int return, not a status)."""
    return load().wn_wgrad_slabs(t_lo, t_hi, chunk, batch)


def causal_codes_slabs(t, batch):
    """Number of slabs one wn_causal_wgrad_codes call writes (plain int return, not a status)."""
    return load().wn_causal_wgrad_codes_slabs(t, batch)


COND_IDX_PAD = 64       # include/wavenet_hip.h WN_COND_IDX_PAD


def ms_slabs(t_lo, t_hi, batch):
    """Number of slabs one wn_resblock_bwd_ms call writes (plain int return, not a status)."""
    return load().wn_resblock_bwd_ms_slabs(t_lo, t_hi, batch)


def pq_chain_ok(t_lo, t_hi, batch, d):
    """Can wn_resblock_bwd_pq run this block in its chain form (dx handed on whole)?  (plain int return, not a status)"""
    return bool(load().wn_resblock_bwd_pq_chain_ok(t_lo, t_hi, batch, d))


def pq_slabs(t_lo, t_hi, batch, d, chain):
    """Number of slabs one wn_resblock_bwd_pq call writes (plain int return, not a status)."""
    return load().wn_resblock_bwd_pq_slabs(t_lo, t_hi, batch, d, 1 if chain else 0)


def enc_slabs(t_lo, t_hi, batch):
    """Number of slabs one wn_enc_resblock_bwd call writes (plain int return, not a status)."""
    return load().wn_enc_resblock_bwd_slabs(t_lo, t_hi, batch)


def decode_sync_granules(n_layers, D, S):
    """uint64 words of hand-off scratch wn_decode_batch_pk wants per utterance (plain int return, not a status)."""
    return int(load().wn_decode_sync_granules(n_layers, D, S))


def call(name, *args):
    """Call an entry point; raise on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise WavenetHipError("%s failed (%d): %s" % (name, rc, lib.wn_last_error().decode()))


def ptr(t, offset=0):
    """Device pointer of a torch tensor (+ element offset), or None."""
    if t is None:
        return None
    return t.data_ptr() + offset * t.element_size()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


_side_streams = {}


def side_stream(device):
    """THE second HIP stream of `device` for every engine of this process (epilogue weight gradients, the second forward
    epilogue chain).  HIGH priority = a hardware queue of its own: a default-priority stream is dealt one of a few hardware
    queues round-robin and can land on the main stream's (the overlap then silently disappears).  One per device, created
    once: streams created later map onto the queues round-robin again - the 4th and 5th engine of a process, each with a
    side stream of its own, ran their config-2 step in 7.6 instead of 4.2 ms (tools/shape_sweep.py)."""
    import os
    import torch
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=dev, priority=-1)
    return st
