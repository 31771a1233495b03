"""The data-parallel path on the real device: `torch.distributed.run` with ONE rank per visible GPU,
backend nccl (= RCCL): process-group init bound to the device, parameter broadcast, the flat-bucket all-reduce on the
engine's gradient buffer between the fused backward and the flat Adam, barrier, teardown.  With one GPU this is the
1-rank RCCL path (the collective must be an identity); the multi-rank arithmetic is covered on CPU by
tests/test_dist_cpu.py (gloo, world_size 2).  Run with -m gpu."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(nproc, tmp_path, with_launcher=True, shards=1):
    env = dict(os.environ, PYTHONPATH=ROOT)
    worker = os.path.join(ROOT, "tests", "dist_worker_gpu.py")
    if with_launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, str(tmp_path), str(shards)]
    else:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, worker, str(tmp_path), str(shards)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.load(open(tmp_path / "dist_gpu.json"))


def test_rccl_path_one_rank_per_gpu(tmp_path):
    n = torch.cuda.device_count()
    assert n >= 1
    plain = _run(1, tmp_path, with_launcher=False)            # no process group at all
    assert plain["backend"] == "none" and len(plain["losses"]) == 3
    if n == 1:
        # one rank under the launcher: the worker forces the process group, so init (device-bound), broadcast,
        # all_reduce on RCCL, barrier and teardown all run; a 1-rank sum is the identity, so the numbers are the plain run's
        one = _run(1, tmp_path)
        assert one["backend"] == "nccl" and one["world"] == 1
        assert one["losses"] == plain["losses"] and one["gsum"] == plain["gsum"]
    else:
        # N ranks with 2 clips each == 1 rank with the same 2N clips (DataParallel's global-batch-mean gradient):
        # losses, the reduced gradient and the parameters after 3 Adam steps agree to fp32 reduction-order noise
        whole = _run(1, tmp_path, with_launcher=False, shards=n)
        many = _run(n, tmp_path, shards=n)
        assert many["backend"] == "nccl" and many["world"] == n
        for a, b in zip(many["losses"], whole["losses"]):
            assert abs(a - b) < 2e-5, (many["losses"], whole["losses"])
        assert abs(many["gsum"] - whole["gsum"]) < 1e-3 * whole["gsum"]
        assert abs(many["psum"] - whole["psum"]) < 1e-5 * whole["psum"]
