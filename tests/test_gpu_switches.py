"""Every documented developer switch (README.md) selects an alternative kernel path that must stay
CORRECT: the 64-channel ragged-shape parity case and the channel-split parity test are re-run in a
fresh interpreter per switch (the switches are read once per process).  Run with -m gpu."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    {"WN_MS_BWD": "0"},            # resblock_bwd_k + 2 x wgrad_k instead of the channel-split block
    {"WN_MS_RW": "0"},             # one-role channel-split block (4 waves) instead of the two-role one (8 waves)
    {"WN_GEMM_RW": "0"},           # one-pass narrow product (chan_gemm_k) for the per-layer data gradient
    {"WN_TALIGN": "4"},            # tile origins at t_lo & ~3 instead of 64-sample lines (also disables the two-role narrow product)
    {"WN_XCD": "0"},               # no XCD-aware block remap
    {"WN_GEMM_WIDE": "1"},         # first wide-GEMM version
    {"WN_GEMM_WIDE": "3"},         # wide GEMM on 32x32x16 MFMAs
    {"WN_FWD_NT": "0"},            # first forward block kernel
    {"WN_FWD_NT": "2"},            # 16 waves x 2 N-tiles
    {"WN_FWD_CS": "1"},            # channel-split forward block
    {"WN_FWD_RW": "1"},            # two-role persistent forward block
    {"WN_GEMM_WIDE_RW": "1"},      # two-role persistent wide GEMM (skip / post-processing products and their data gradients)
]


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_alternative_paths_stay_correct(env):
    e = dict(os.environ, **env)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_sweep.py"), "-k", "d10_R64 or d2_R64 or d4_R48",
           os.path.join(ROOT, "tests", "test_gpu_parity.py") + "::test_fused_train_step_matches_autograd_path_and_oracle"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("env", [{"WN_AE_FUSED_ENC": "0"}, {"WN_AE_FUSED_ENC_BWD": "0"}, {"WN_MS_RW": "0"}], ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_autoencoder_alternative_paths_stay_correct(env):
    """The autoencoder with its encoder blocks as two channel GEMMs (instead of wn_enc_resblock_fwd) / its decoder
    blocks on the one-role backward kernel / its encoder blocks' backward as GEMM + weight-gradient launches: the G8 forward fixture and the 64-channel backward parity test."""
    e = dict(os.environ, **env)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-k", "g8_autoencoder_forward or autoencoder_backward_64"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("env", [{"WN_DEC_MFMA": "0"}, {"WN_DEC_MFMA_POST": "0"}, {"WN_DEC_PIPE": "1"}, {"WN_DEC_U8": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_decode_alternative_paths_stay_correct(env):
    """The cached-queue decoder on its other kernels: the fp32 FMA pair of workgroups (WN_DEC_MFMA=0), the matrix-core
    chain with FMA skip / post-processing (WN_DEC_MFMA_POST=0), the pipeline of register-resident stages
    (WN_DEC_PIPE=1) and the one-utterance-per-pair matrix-core kernel (WN_DEC_U8=0) - the config-5 oracle test (both queue recurrences), the one-launch generation test and the
    batched / sampling tests."""
    e = dict(os.environ, **env)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-k", "decode or generat"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
