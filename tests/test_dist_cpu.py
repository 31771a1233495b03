"""world_size-2 data-parallel tests on CPU (gloo): the flat-bucket all-reduce reproduces
DataParallel's global-batch-mean gradient, and train() under 2 ranks writes the same logs /
checkpoints as 1 rank with the same global batch."""
import json
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import torch

from tests.helpers import ROOT

CFG = dict(filter_width=2, dilations=[1, 2, 4, 8], dilation_channels=16, residual_channels=16,
           skip_channels=16, quantization_channels=256, use_bias=False)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(nproc, mode, workdir):
    env = dict(os.environ, OMP_NUM_THREADS="2" if nproc <= 2 else "1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(workdir)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_allreduce_equals_global_batch_gradient(tmp_path):
    from tests.cpu_model import OracleWavenet, onehot_oracle
    json.dump(CFG, open(tmp_path / "cfg.json", "w"))
    rng = np.random.default_rng(0)
    rf = 17
    codes = torch.from_numpy(rng.integers(0, 256, size=(4, rf + 39)).astype(np.int64))
    x = onehot_oracle(codes)
    y = torch.from_numpy(rng.integers(0, 256, size=(4, 40)).astype(np.int64))
    torch.save({"x": x, "y": y}, tmp_path / "batch.pt")
    _launch(2, "grads", tmp_path)
    torch.manual_seed(0)
    net = OracleWavenet(**CFG)
    with torch.no_grad():
        pass
    loss = torch.nn.CrossEntropyLoss()(net(x), y.reshape(-1))
    loss.backward()
    got = torch.load(tmp_path / "grads_dp.pt")
    for k, p in net.named_parameters():
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        assert (got[k] - want).abs().max().item() <= 1e-6 * max(1e-3, want.abs().max().item()) + 1e-9, k


def _write_run(tmp, batch_size, lens=(900, 700)):
    os.makedirs(tmp / "params", exist_ok=True)
    rng = np.random.default_rng(5)
    data = [rng.integers(0, 256, size=(l,)).astype(np.int32) for l in lens]
    pickle.dump(data, open(tmp / "np_audio.pkl", "wb"))
    dp = dict(batch_size=batch_size, shuffle=True, num_workers=0, pin_memory=False, audio_path=str(tmp / "np_audio.pkl"),
              receptive_field=17, window_length=100, cuda_available=False, quantization_channels=256)
    tp = dict(log_dir="./log/", restore_dir="./restore/", restore_model="", check_point_every=1, print_every=1,
              num_epochs=2, wavenet_params="", optimizer="adam", max_check_points=10, learning_rate=1e-3,
              momentum=0.9, device_ids=None, seed=3)
    for n, p in (("wavenet", CFG), ("dataset", dp), ("train", tp)):
        json.dump(p, open(tmp / "params" / (n + "_params.json"), "w"))


import pytest


# (900, 700): 24 pieces = whole batches of 4; (917, 617): 15 pieces, the last batch has 3 items (shards of 2 and 1);
# (917, 734): 17 pieces, the last batch has ONE item (rank 1 gets an empty shard and must still join the all-reduce)
@pytest.mark.parametrize("lens", [(900, 700), (917, 617), (917, 734)])
def test_train_two_ranks_equals_one_rank(tmp_path, lens):
    a, b = tmp_path / "one", tmp_path / "two"
    os.makedirs(a), os.makedirs(b)
    _write_run(a, 4, lens)
    _write_run(b, 4, lens)
    _launch(1, "train", a)
    _launch(2, "train", b)
    la = open(a / "log" / "loss_log.log").read().strip().split("\n")
    lb = open(b / "log" / "loss_log.log").read().strip().split("\n")
    assert len(la) == len(lb) and len(la) >= 4
    for x, y in zip(la, lb):
        assert x.split("Average")[0] == y.split("Average")[0]
        assert abs(float(x.split(' ')[-1]) - float(y.split(' ')[-1])) < 2e-6
    assert open(a / "log" / "store_log.log").read() == open(b / "log" / "store_log.log").read()
    ca, cb = torch.load(a / "restore" / "wavenet2.model"), torch.load(b / "restore" / "wavenet2.model")
    for k in ca:
        assert (ca[k] - cb[k]).abs().max().item() < 1e-5, k


def test_train_eight_ranks_equals_one_rank(tmp_path):
    """The 8-GPU job's host logic (BASELINE configs[2]: DP over 8 ranks) on CPU: 8 gloo ranks, global batch 8, 15 pieces - a whole
    batch (one item per rank) and a ragged one of 7 (rank 7 gets an EMPTY shard and must still join the all-reduce) - write the
    same loss log, store log and checkpoint as ONE rank with the same global batch (wavenet/train.py:116-122: DataParallel's
    global-batch mean)."""
    a, b = tmp_path / "one", tmp_path / "eight"
    os.makedirs(a), os.makedirs(b)
    _write_run(a, 8, (917, 617))
    _write_run(b, 8, (917, 617))
    _launch(1, "train", a)
    _launch(8, "train", b)
    la = open(a / "log" / "loss_log.log").read().strip().split("\n")
    lb = open(b / "log" / "loss_log.log").read().strip().split("\n")
    assert len(la) == len(lb) and len(la) >= 4
    for x, y in zip(la, lb):
        assert x.split("Average")[0] == y.split("Average")[0]
        assert abs(float(x.split(' ')[-1]) - float(y.split(' ')[-1])) < 2e-6
    assert open(a / "log" / "store_log.log").read() == open(b / "log" / "store_log.log").read()
    ca, cb = torch.load(a / "restore" / "wavenet2.model"), torch.load(b / "restore" / "wavenet2.model")
    for k in ca:
        assert (ca[k] - cb[k]).abs().max().item() < 1e-5, k


def test_thread_budget_per_rank(monkeypatch):
    """Host threads per rank = ceil(CPU quota / ranks of this host) (VERDICT r4 #4a): 8 ranks on the GPU box's 16-CPU quota get 2
    each - 8 x 16 runnable threads on 16 CPUs was the 68-ms-per-step throttling stall of DESIGN_HISTORY.md."""
    from music_amd import _lib
    assert _lib.thread_budget(16, 8) == 2 and _lib.thread_budget(16, 1) == 16 and _lib.thread_budget(16, 3) == 6
    assert _lib.thread_budget(4, 8) == 1 and _lib.thread_budget(None, 8, cpus=256) == 32 and _lib.thread_budget(None, 1, cpus=8) == 8
    assert _lib.local_world_size({"LOCAL_WORLD_SIZE": "8"}) == 8 and _lib.local_world_size({}) == 1
    assert _lib.local_world_size({"LOCAL_WORLD_SIZE": "x"}) == 1
    before = torch.get_num_threads()
    try:
        monkeypatch.setattr(_lib, "cpu_quota", lambda: 16)
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
        monkeypatch.delenv("WN_KEEP_TORCH_THREADS", raising=False)
        torch.set_num_threads(8)
        _lib.respect_cpu_quota()
        assert torch.get_num_threads() == 2
        torch.set_num_threads(8)
        monkeypatch.setenv("WN_KEEP_TORCH_THREADS", "1")
        _lib.respect_cpu_quota()
        assert torch.get_num_threads() == 8
        # no quota, one process: torch's own setting stays
        monkeypatch.delenv("WN_KEEP_TORCH_THREADS", raising=False)
        monkeypatch.setattr(_lib, "cpu_quota", lambda: None)
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
        _lib.respect_cpu_quota()
        assert torch.get_num_threads() == 8
    finally:
        torch.set_num_threads(before)


def test_collate_shards_like_dataparallel_scatter():
    """_Collate's shard=(r, w) slices are torch.chunk's (DataParallel scatter) for every batch length, and the
    dp_scale weights sum to world (so the weighted mean over ranks is the global-batch mean)."""
    from music_amd.faster_audio_data import shard_bounds
    for w in (2, 3, 4, 8):
        for n in range(1, 2 * w + 2):
            want = [c.tolist() for c in torch.arange(n).chunk(w)]
            got = [list(range(*shard_bounds(n, r, w))) for r in range(w)]
            assert [g for g in got if g] == want, (n, w, got, want)
            assert all(not g for g in got[len(want):])
            assert abs(sum(len(g) * w / n for g in got) - w) < 1e-12


def test_collate_ragged_batches_no_gpu(monkeypatch):
    from music_amd import faster_audio_data as fad
    from tests.cpu_model import onehot_oracle
    monkeypatch.setattr(fad, "onehot_device", onehot_oracle)
    items = [{"audio_piece": torch.full((20,), i, dtype=torch.int32), "audio_target": torch.full((4,), i)} for i in range(3)]
    b0 = fad._Collate(256, (0, 2))(items)
    b1 = fad._Collate(256, (1, 2))(items)
    assert b0["audio_target"][:, 0].tolist() == [0, 1] and b1["audio_target"][:, 0].tolist() == [2]
    assert abs(b0["dp_scale"] - 4 / 3) < 1e-12 and abs(b1["dp_scale"] - 2 / 3) < 1e-12
    e = fad._Collate(256, (1, 2))(items[:1])
    assert e["audio_piece"] is None and e["dp_scale"] == 0.0
    assert "dp_scale" not in fad._Collate(256)(items)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus N` (no launcher around it) must start N ranks itself: the parent spawns
    a torch.distributed.run child before anything touches the GPU and relays rank 0's JSON line and the exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    env.pop("OMP_NUM_THREADS")          # the parent sets the per-rank budget itself (quota / ranks, at most 8)
    for n in (1, 2, 3, 8):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out["n_gpus"] == n and out["rank_sum"] == n * (n + 1) / 2
        if n > 1:
            from music_amd import _lib
            assert out["omp_num_threads"] == str(min(8, _lib.thread_budget(_lib.cpu_quota(), n))), out
