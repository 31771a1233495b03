"""GPU unit tests of the individual C-ABI kernels against plain fp64/fp32 math (run with -m gpu)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from music_amd import _lib
from music_amd._lib import call, ptr
from music_amd.engine import pack_index, SLACK

DEV = "cuda"


def _packed(wmat, mode, chained=False):
    """wmat: numpy [M,K] float32 (M%16==0, K%32==0) -> (packed int16 tensor, flat tensor)."""
    m, k = wmat.shape
    flat = torch.from_numpy(np.ascontiguousarray(wmat, dtype=np.float32).reshape(-1)).to(DEV)
    weff = np.arange(m * k, dtype=np.int64).reshape(m, k)
    idx = torch.from_numpy(pack_index(weff, chained)).to(DEV)
    hp = 1024 if mode in (_lib.F16X3, _lib.BF16X3) else 512
    out = torch.zeros(idx.numel() // 512 * hp, dtype=torch.int16, device=DEV)
    call("wn_pack_weights", ptr(flat), ptr(idx), ptr(out), idx.numel(), mode, _lib.stream())
    return out


def _buf(b, rows, pitch, fill=None, seed=0):
    t = torch.zeros(SLACK + b * rows * pitch + 512, dtype=torch.float32, device=DEV)
    if fill is not None:
        g = torch.Generator(device="cpu").manual_seed(seed)
        v = torch.randn(b, rows, pitch, generator=g) * fill
        t[SLACK:SLACK + b * rows * pitch] = v.reshape(-1).to(DEV)
    return t


def _view(t, b, rows, pitch):
    return t[SLACK:SLACK + b * rows * pitch].view(b, rows, pitch)


# relative to the reference tensor's max-abs; observed 2.6e-7 / 5.7e-6 / 3.6e-4 / 2.6e-3 (round 6: halved - thirded, they were 40 - 80 x loose)
TOL = {_lib.F16X3: 1e-5, _lib.BF16X3: 1e-4, _lib.F16X1: 1e-2, _lib.BF16X1: 5e-2}


@pytest.mark.parametrize("mode", [_lib.F16X3, _lib.BF16X3, _lib.F16X1, _lib.BF16X1])
def test_chan_gemm_single_tap(mode):
    rng = np.random.default_rng(1)
    B, M, K, pitch, T = 2, 80, 96, 1024, 700          # M = 5 tiles (partial M-block), K = 3 k-steps
    w = rng.standard_normal((M, K)).astype(np.float32) * 0.2
    pk = _packed(w, mode)
    xin = _buf(B, K, pitch, 1.0, 2)
    out = _buf(B, M, pitch)
    t_lo, t_hi = 37, T
    call("wn_chan_gemm", ptr(xin, SLACK), None, K * pitch, pitch, 0, pitch, 0, 0, K // 32, 0, ptr(pk), M // 16, M - 3,
         ptr(out, SLACK), M * pitch, pitch, 0, None, None, 0, 0, 0, None, 0, 0, t_lo, t_hi, 0, B, mode, _lib.stream())
    torch.cuda.synchronize()
    x = _view(xin, B, K, pitch).cpu().double()
    ref = torch.einsum("mk,bkt->bmt", torch.from_numpy(w).double(), x)
    got = _view(out, B, M, pitch).cpu().double()
    err = (got[:, :M - 3, t_lo:t_hi] - ref[:, :M - 3, t_lo:t_hi]).abs().max().item()
    scale = ref.abs().max().item()
    print("mode", mode, "err", err, "scale", scale)
    assert err <= TOL[mode] * scale
    # nothing outside the valid window / valid rows was written
    assert got[:, :, :t_lo].abs().max().item() == 0 and got[:, :, t_hi:].abs().max().item() == 0
    assert got[:, M - 3:].abs().sum().item() == 0


@pytest.mark.parametrize("mode", [_lib.BF16X3, _lib.F16X3])
@pytest.mark.parametrize("B,T,t_lo,M,m_valid,epi", [
    (2, 1500, 37, 576, 576, "plain"),         # 24 tiles < 256 workgroups: every tile is dealt out by passes
    (3, 12700, 3069, 576, 570, "plain"),      # 300+ tiles: whole tiles round-robin + a leftover round split by passes; partial last row tile
    (2, 2000, 130, 768, 768, "mask+bias+relu+compact"),
])
def test_chan_gemm_b_stationary(mode, B, T, t_lo, M, m_valid, epi, monkeypatch):
    """K = 256, >= 512 rows in groups of 3 row tiles: chan_gemm_bst_k (wn_gemm_bst.hip; the dZ = Ws^T dU product of
    wavenet/model.py:127-134's backward).  Against float64, and BIT for bit against chan_gemm_wide2_k (same order of terms)."""
    rng = np.random.default_rng(7)
    K = 256
    pitch = ((T + 255) // 256) * 256 + 512
    w = rng.standard_normal((M, K)).astype(np.float32) * 0.1
    pk = _packed(w, mode)
    xin = _buf(B, K, pitch, 1.0, 11)
    fancy = epi != "plain"
    t_hi = T
    in_lo, in_hi = (t_lo + 3, T - 5) if fancy else (t_lo, T)
    W_out = t_hi - t_lo
    bias = torch.from_numpy(rng.standard_normal(M).astype(np.float32)).to(DEV) if fancy else None
    msk = _buf(B, M, pitch, 1.0, 5) if fancy else None

    def run():
        if fancy:       # compact output (column t - t_lo), mask, bias, ReLU on the input
            out = torch.zeros(B * M * W_out + 512, device=DEV)
            call("wn_chan_gemm", ptr(xin, SLACK), None, K * pitch, pitch, in_lo, in_hi, 0, 0, K // 32, 0, ptr(pk), M // 16, m_valid,
                 ptr(out), M * W_out, W_out, -t_lo, ptr(bias), None, 0, 0, 0, ptr(msk, SLACK), M * pitch, pitch,
                 t_lo, t_hi, 1, B, mode, _lib.stream())
            torch.cuda.synchronize()
            return out[:B * M * W_out].view(B, M, W_out).clone()
        out = _buf(B, M, pitch)
        call("wn_chan_gemm", ptr(xin, SLACK), None, K * pitch, pitch, in_lo, in_hi, 0, 0, K // 32, 0, ptr(pk), M // 16, m_valid,
             ptr(out, SLACK), M * pitch, pitch, 0, None, None, 0, 0, 0, None, 0, 0, t_lo, t_hi, 0, B, mode, _lib.stream())
        torch.cuda.synchronize()
        return _view(out, B, M, pitch).clone()

    monkeypatch.setenv("WN_GEMM_BST", "1")
    got = run()
    monkeypatch.setenv("WN_GEMM_BST", "0")
    wide = run()
    assert torch.equal(got, wide), "B-stationary and 256 x 256-tile forms differ: %g" % (got - wide).abs().max().item()
    x = _view(xin, B, K, pitch).cpu().double()
    if fancy:
        x = x.clamp(min=0)
    win = torch.zeros_like(x)
    win[:, :, in_lo:in_hi] = x[:, :, in_lo:in_hi]
    ref = torch.einsum("mk,bkt->bmt", torch.from_numpy(w).double(), win[:, :, t_lo:t_hi])
    g = got.cpu().double()
    if fancy:
        ref = ref + bias.cpu().double()[None, :, None]
        ref = torch.where(_view(msk, B, M, pitch).cpu().double()[:, :, t_lo:t_hi] > 0, ref, torch.zeros_like(ref))
        gv = g
    else:
        gv = g[:, :, t_lo:t_hi]
        assert g[:, :, :t_lo].abs().max().item() == 0 and g[:, :, t_hi:].abs().max().item() == 0
    err = (gv[:, :m_valid] - ref[:, :m_valid]).abs().max().item()
    scale = ref.abs().max().item()
    print("mode", mode, epi, "err", err, "scale", scale)
    assert err <= TOL[mode] * scale
    if m_valid < M:
        assert gv[:, m_valid:].abs().sum().item() == 0


@pytest.mark.parametrize("mode", [_lib.F16X3, _lib.BF16X3])
@pytest.mark.parametrize("B,T,t_lo,KZ,S,Qv,bias", [(2, 1500, 37, 128, 256, 256, False), (3, 2300, 1023, 192, 250, 256, True)])
def test_skip_epilogue_fwd_fused(mode, B, T, t_lo, KZ, S, Qv, bias):
    """wn_skip_epilogue_fwd (wavenet/model.py:127-138 in one launch): u, h and the compact pre-softmax against float64 and against
    the three wn_chan_gemm launches it replaces (chained vs natural k order inside an MFMA: same terms, other order)."""
    rng = np.random.default_rng(11)
    pitch = ((T + 255) // 256) * 256 + 512
    ws_ = np.zeros((256, KZ), np.float32); ws_[:S] = rng.standard_normal((S, KZ)).astype(np.float32) * 0.08
    p1 = np.zeros((256, 256), np.float32); p1[:S, :S] = rng.standard_normal((S, S)).astype(np.float32) * 0.08
    p2 = np.zeros((256, 256), np.float32); p2[:Qv, :S] = rng.standard_normal((Qv, S)).astype(np.float32) * 0.08
    pk_s, pk_1, pk_2 = _packed(ws_, mode), _packed(p1, mode), _packed(p2, mode)
    pk_1c, pk_2c = _packed(p1, mode, chained=True), _packed(p2, mode, chained=True)
    z = _buf(B, KZ, pitch, 1.0, 3)
    bs = [torch.from_numpy(rng.standard_normal(256).astype(np.float32)).to(DEV) if bias else None for _ in range(3)]
    W = T - t_lo
    st = _lib.stream()

    def fused():
        u, h = _buf(B, 256, pitch), _buf(B, 256, pitch)
        o = torch.zeros(B * 256 * W + 512, device=DEV)
        call("wn_skip_epilogue_fwd", ptr(z, SLACK), KZ * pitch, pitch, KZ // 32, ptr(pk_s), ptr(bs[0]) if bias else None,
             ptr(u, SLACK), ptr(h, SLACK), 256 * pitch, ptr(pk_1c), ptr(bs[1]) if bias else None, ptr(pk_2c),
             ptr(bs[2]) if bias else None, ptr(o), 256 * W, W, S, Qv, t_lo, T, B, mode, st)
        torch.cuda.synchronize()
        return _view(u, B, 256, pitch).clone(), _view(h, B, 256, pitch).clone(), o[:B * 256 * W].view(B, 256, W).clone()

    def three():
        u, h = _buf(B, 256, pitch), _buf(B, 256, pitch)
        o = torch.zeros(B * 256 * W + 512, device=DEV)
        call("wn_chan_gemm", ptr(z, SLACK), None, KZ * pitch, pitch, t_lo, T, 0, 0, KZ // 32, 0, ptr(pk_s), 16, S, ptr(u, SLACK), 256 * pitch,
             pitch, 0, ptr(bs[0]) if bias else None, None, 0, 0, 0, None, 0, 0, t_lo, T, 0, B, mode, st)
        call("wn_chan_gemm", ptr(u, SLACK), None, 256 * pitch, pitch, t_lo, T, 0, 0, 8, 0, ptr(pk_1), 16, S, ptr(h, SLACK), 256 * pitch,
             pitch, 0, ptr(bs[1]) if bias else None, None, 0, 0, 0, None, 0, 0, t_lo, T, 1, B, mode, st)
        call("wn_chan_gemm", ptr(h, SLACK), None, 256 * pitch, pitch, t_lo, T, 0, 0, 8, 0, ptr(pk_2), 16, Qv, ptr(o), 256 * W, W, -t_lo,
             ptr(bs[2]) if bias else None, None, 0, 0, 0, None, 0, 0, t_lo, T, 1, B, mode, st)
        torch.cuda.synchronize()
        return _view(u, B, 256, pitch).clone(), _view(h, B, 256, pitch).clone(), o[:B * 256 * W].view(B, 256, W).clone()

    fu, fh, fo = fused()
    tu, th, to = three()
    zz = _view(z, B, KZ, pitch).cpu().double()[:, :, t_lo:T]
    bb = [b_.cpu().double()[None, :, None] if bias else 0.0 for b_ in bs]
    ru = torch.einsum("mk,bkt->bmt", torch.from_numpy(ws_).double(), zz) + bb[0]
    rh = torch.einsum("mk,bkt->bmt", torch.from_numpy(p1).double(), ru.clamp(min=0)) + bb[1]
    ro = torch.einsum("mk,bkt->bmt", torch.from_numpy(p2).double(), rh.clamp(min=0)) + bb[2]
    tol = TOL[mode] * 3
    for name, got, other, ref, rows in (("u", fu, tu, ru, S), ("h", fh, th, rh, S)):
        g = got.cpu().double()
        assert g[:, :, :t_lo].abs().max().item() == 0 and g[:, :, T:].abs().max().item() == 0, name
        assert g[:, rows:].abs().sum().item() == 0, name
        err = (g[:, :rows, t_lo:T] - ref[:, :rows]).abs().max().item()
        dev = (g - other.cpu().double()).abs().max().item()
        print(name, "err", err, "vs three launches", dev, "scale", ref.abs().max().item())
        assert err <= tol * ref.abs().max().item() and dev <= tol * ref.abs().max().item()
    g = fo.cpu().double()
    err = (g[:, :Qv] - ro[:, :Qv]).abs().max().item()
    dev = (g - to.cpu().double()).abs().max().item()
    print("o err", err, "vs three launches", dev, "scale", ro.abs().max().item())
    assert err <= tol * ro.abs().max().item() and dev <= tol * ro.abs().max().item()
    assert g[:, Qv:].abs().sum().item() == 0


@pytest.mark.parametrize("mode", [_lib.BF16X3, _lib.F16X3])
@pytest.mark.parametrize("B,T,t_lo,MZ,S", [(2, 1500, 37, 192, 256), (3, 2300, 1023, 480, 250)])
def test_skip_epilogue_bwd_fused(mode, B, T, t_lo, MZ, S):
    """wn_skip_epilogue_bwd (autograd of wavenet/model.py:127-138, data gradients, in one launch): dh, du, dz against float64 and
    against the three wn_chan_gemm launches it replaces."""
    rng = np.random.default_rng(13)
    pitch = ((T + 255) // 256) * 256 + 512
    W = T - t_lo
    p2 = np.zeros((256, 256), np.float32); p2[:, :S] = rng.standard_normal((256, S)).astype(np.float32) * 0.08       # [Q][S]
    p1 = np.zeros((256, 256), np.float32); p1[:S, :S] = rng.standard_normal((S, S)).astype(np.float32) * 0.08       # [S][S]
    ws_ = np.zeros((256, MZ), np.float32); ws_[:S] = rng.standard_normal((S, MZ)).astype(np.float32) * 0.08         # [S][N*CH]
    pk_p2T, pk_p1T, pk_sT = _packed(p2.T.copy(), mode), _packed(p1.T.copy(), mode), _packed(ws_.T.copy(), mode)
    pk_p1Tc, pk_sTc = _packed(p1.T.copy(), mode, chained=True), _packed(ws_.T.copy(), mode, chained=True)
    g = torch.Generator(device="cpu").manual_seed(5)
    dO = (torch.randn(B, 256, W, generator=g) * 1e-3).to(DEV).contiguous()
    h, u = _buf(B, 256, pitch, 1.0, 6), _buf(B, 256, pitch, 1.0, 7)
    st = _lib.stream()

    def fused():
        dh, du, dz = _buf(B, 256, pitch), _buf(B, 256, pitch), _buf(B, MZ, pitch)
        call("wn_skip_epilogue_bwd", ptr(dO), 256 * W, W, ptr(h, SLACK), ptr(u, SLACK), 256 * pitch, pitch, ptr(dh, SLACK), ptr(du, SLACK),
             ptr(dz, SLACK), MZ * pitch, ptr(pk_p2T), ptr(pk_p1Tc), ptr(pk_sTc), MZ // 16, MZ, S, t_lo, T, B, mode, st)
        torch.cuda.synchronize()
        return _view(dh, B, 256, pitch).clone(), _view(du, B, 256, pitch).clone(), _view(dz, B, MZ, pitch).clone()

    def three():
        dh, du, dz = _buf(B, 256, pitch), _buf(B, 256, pitch), _buf(B, MZ, pitch)
        call("wn_chan_gemm", ptr(dO), None, 256 * W, W, 0, W, -t_lo, 0, 8, 0, ptr(pk_p2T), 16, S, ptr(dh, SLACK), 256 * pitch, pitch, 0, None,
             None, 0, 0, 0, ptr(h, SLACK), 256 * pitch, pitch, t_lo, T, 0, B, mode, st)
        call("wn_chan_gemm", ptr(dh, SLACK), None, 256 * pitch, pitch, t_lo, T, 0, 0, 8, 0, ptr(pk_p1T), 16, S, ptr(du, SLACK), 256 * pitch,
             pitch, 0, None, None, 0, 0, 0, ptr(u, SLACK), 256 * pitch, pitch, t_lo, T, 0, B, mode, st)
        call("wn_chan_gemm", ptr(du, SLACK), None, 256 * pitch, pitch, t_lo, T, 0, 0, 8, 0, ptr(pk_sT), MZ // 16, MZ, ptr(dz, SLACK), MZ * pitch,
             pitch, 0, None, None, 0, 0, 0, None, 0, 0, t_lo, T, 0, B, mode, st)
        torch.cuda.synchronize()
        return _view(dh, B, 256, pitch).clone(), _view(du, B, 256, pitch).clone(), _view(dz, B, MZ, pitch).clone()

    fh, fu, fz = fused()
    th, tu, tz = three()
    dod = dO.cpu().double()
    hm = (_view(h, B, 256, pitch).cpu().double()[:, :, t_lo:T] > 0)
    um = (_view(u, B, 256, pitch).cpu().double()[:, :, t_lo:T] > 0)
    rh = torch.einsum("qs,bqt->bst", torch.from_numpy(p2).double(), dod) * hm
    ru = torch.einsum("rs,brt->bst", torch.from_numpy(p1).double(), rh) * um
    rz = torch.einsum("sm,bst->bmt", torch.from_numpy(ws_).double(), ru)
    tol = TOL[mode] * 3
    for name, got, other, ref, rows in (("dh", fh, th, rh, S), ("du", fu, tu, ru, S), ("dz", fz, tz, rz, MZ)):
        gd = got.cpu().double()
        assert gd[:, :, :t_lo].abs().max().item() == 0 and gd[:, :, T:].abs().max().item() == 0, name
        assert gd[:, rows:].abs().sum().item() == 0, name
        err = (gd[:, :rows, t_lo:T] - ref[:, :rows]).abs().max().item()
        dev = (gd - other.cpu().double()).abs().max().item()
        print(name, "err", err, "vs three launches", dev, "scale", ref.abs().max().item())
        assert err <= tol * ref.abs().max().item() and dev <= tol * ref.abs().max().item()


def test_chan_gemm_two_taps_epilogues():
    mode = _lib.F16X3
    rng = np.random.default_rng(2)
    B, M, K, pitch, T = 2, 64, 64, 768, 600
    w = rng.standard_normal((M, 2 * K)).astype(np.float32) * 0.2
    pk = _packed(w, mode)
    xin = _buf(B, K, pitch, 1.0, 3)
    res = _buf(B, M, pitch, 1.0, 4)
    msk = _buf(B, M, pitch, 1.0, 5)
    bias = torch.from_numpy(rng.standard_normal(M).astype(np.float32)).to(DEV)
    W_out = 500
    out = torch.zeros(B * M * W_out + 512, device=DEV)
    d, t_lo, t_hi = 7, 50, 550                                  # compact output: col = t - 50
    in_lo, in_hi = 45, 548                                      # input window narrower than needed
    call("wn_chan_gemm", ptr(xin, SLACK), ptr(xin, SLACK), K * pitch, pitch, in_lo, in_hi, -d, 0, K // 32, K // 32, ptr(pk),
         M // 16, M, ptr(out), M * W_out, W_out, -t_lo, ptr(bias), ptr(res, SLACK), M * pitch, pitch, 60,
         ptr(msk, SLACK), M * pitch, pitch, t_lo, t_hi, 1, B, mode, _lib.stream())
    torch.cuda.synchronize()
    x = _view(xin, B, K, pitch).cpu().double().clamp(min=0)
    win = torch.zeros_like(x)
    win[:, :, in_lo:in_hi] = x[:, :, in_lo:in_hi]
    wt = torch.from_numpy(w).double()
    ts = torch.arange(t_lo, t_hi)
    ref = torch.einsum("mk,bkt->bmt", wt[:, :K], win[:, :, ts - d]) + torch.einsum("mk,bkt->bmt", wt[:, K:], win[:, :, ts])
    ref = ref + bias.cpu().double()[None, :, None]
    ref = torch.where(_view(msk, B, M, pitch).cpu().double()[:, :, ts] > 0, ref, torch.zeros_like(ref))
    r = _view(res, B, M, pitch).cpu().double()[:, :, ts]
    r[:, :, ts < 60] = 0
    ref = ref + r                                              # the residual is added AFTER the mask
    got = out[:B * M * W_out].view(B, M, W_out).cpu().double()
    err = (got - ref).abs().max().item()
    print("err", err, "scale", ref.abs().max().item())
    assert err <= 3e-5 * ref.abs().max().item()


@pytest.mark.parametrize("mode", [_lib.BF16X3, _lib.F16X3])
@pytest.mark.parametrize("d,in_lo,in_hi,t_lo,t_hi", [(5, 41, 930, 36, 925), (64, 100, 1024, 36, 1000), (1, 7, 300, 6, 299)])
def test_chan_gemm_two_role_window_and_canaries(mode, d, in_lo, in_hi, t_lo, t_hi):
    """The per-layer data-gradient shape (64 rows, two taps of 128 rows = 8 k-steps, residual) runs on the two-role
    persistent kernel (chan_gemm_rw_k).  Input columns outside [in_lo, in_hi) must read as 0 AND must never be
    dereferenced: they hold NaN here, as does everything around the output window."""
    rng = np.random.default_rng(6)
    B, M, K, pitch = 3, 64, 128, 1024
    w = rng.standard_normal((M, 2 * K)).astype(np.float32) * 0.1
    pk = _packed(w, mode)
    xin = _buf(B, K, pitch, 1.0, 7)
    _view(xin, B, K, pitch)[:, :, :in_lo] = float("nan")
    _view(xin, B, K, pitch)[:, :, in_hi:] = float("nan")
    res = _buf(B, M, pitch, 1.0, 8)
    out = _buf(B, M, pitch)
    out.fill_(float("nan"))
    resid_lo = t_lo + d
    call("wn_chan_gemm", ptr(xin, SLACK), ptr(xin, SLACK), K * pitch, pitch, in_lo, in_hi, 0, d, K // 32, K // 32, ptr(pk),
         M // 16, M - 5, ptr(out, SLACK), M * pitch, pitch, 0, None, ptr(res, SLACK), M * pitch, pitch, resid_lo,
         None, 0, 0, t_lo, t_hi, 0, B, mode, _lib.stream())
    torch.cuda.synchronize()
    x = torch.nan_to_num(_view(xin, B, K, pitch).cpu().double(), nan=0.0)
    xpad = torch.cat([x, torch.zeros(B, K, d + 8, dtype=torch.float64)], 2)
    wt = torch.from_numpy(w).double()
    ts = torch.arange(t_lo, t_hi)
    ref = torch.einsum("mk,bkt->bmt", wt[:, :K], xpad[:, :, ts]) + torch.einsum("mk,bkt->bmt", wt[:, K:], xpad[:, :, ts + d])
    r = _view(res, B, M, pitch).cpu().double()[:, :, ts]
    r[:, :, ts < resid_lo] = 0
    ref = ref + r
    got = _view(out, B, M, pitch).cpu().double()
    err = (got[:, :M - 5, t_lo:t_hi] - ref[:, :M - 5]).abs().max().item()
    assert err <= TOL[mode] * ref.abs().max().item(), err          # also fails on any NaN that leaked in
    assert torch.isnan(got[:, :, :t_lo]).all() and torch.isnan(got[:, :, t_hi:]).all() and torch.isnan(got[:, M - 5:]).all()


def test_chan_gemm_two_role_four_ksteps_with_mask():
    """The encoder's data-gradient shape: two taps of 64 rows (4 k-steps), ReLU mask, residual from resid_lo on - also
    on the two-role persistent kernel."""
    mode = _lib.BF16X3
    rng = np.random.default_rng(9)
    B, M, K, pitch = 2, 64, 64, 1280
    d, in_lo, in_hi, t_lo, t_hi, resid_lo = 6, 50, 1100, 44, 1094, 70
    w = rng.standard_normal((M, 2 * K)).astype(np.float32) * 0.1
    pk = _packed(w, mode)
    xin = _buf(B, K, pitch, 1.0, 10)
    _view(xin, B, K, pitch)[:, :, :in_lo] = float("nan")
    _view(xin, B, K, pitch)[:, :, in_hi:] = float("nan")
    res = _buf(B, M, pitch, 1.0, 11)
    msk = _buf(B, M, pitch, 1.0, 12)
    out = _buf(B, M, pitch)
    call("wn_chan_gemm", ptr(xin, SLACK), ptr(xin, SLACK), K * pitch, pitch, in_lo, in_hi, 0, d, K // 32, K // 32, ptr(pk),
         M // 16, M, ptr(out, SLACK), M * pitch, pitch, 0, None, ptr(res, SLACK), M * pitch, pitch, resid_lo,
         ptr(msk, SLACK), M * pitch, pitch, t_lo, t_hi, 0, B, mode, _lib.stream())
    torch.cuda.synchronize()
    x = torch.nan_to_num(_view(xin, B, K, pitch).cpu().double(), nan=0.0)
    xpad = torch.cat([x, torch.zeros(B, K, d + 8, dtype=torch.float64)], 2)
    wt = torch.from_numpy(w).double()
    ts = torch.arange(t_lo, t_hi)
    ref = torch.einsum("mk,bkt->bmt", wt[:, :K], xpad[:, :, ts]) + torch.einsum("mk,bkt->bmt", wt[:, K:], xpad[:, :, ts + d])
    ref = torch.where(_view(msk, B, M, pitch).cpu().double()[:, :, ts] > 0, ref, torch.zeros_like(ref))
    r = _view(res, B, M, pitch).cpu().double()[:, :, ts]
    r[:, :, ts < resid_lo] = 0
    ref = ref + r
    got = _view(out, B, M, pitch).cpu().double()
    err = (got[:, :, t_lo:t_hi] - ref).abs().max().item()
    assert err <= TOL[mode] * ref.abs().max().item(), err
    assert got[:, :, :t_lo].abs().max().item() == 0 and got[:, :, t_hi:].abs().max().item() == 0


def test_chan_gemm_user_tensor_unaligned_pitch():
    """Causal-conv use: input is a plain contiguous (B,Q,T) tensor with T % 4 != 0."""
    mode = _lib.F16X3
    rng = np.random.default_rng(3)
    B, Q, T, M, pitch = 2, 256, 333, 32, 768
    w = rng.standard_normal((M, 2 * Q)).astype(np.float32) * 0.1
    pk = _packed(w, mode)
    x = torch.from_numpy(rng.standard_normal((B, Q, T)).astype(np.float32)).to(DEV)
    out = _buf(B, M, pitch)
    call("wn_chan_gemm", ptr(x), ptr(x), Q * T, T, 0, T, -1, 0, Q // 32, Q // 32, ptr(pk), M // 16, M,
         ptr(out, SLACK), M * pitch, pitch, 0, None, None, 0, 0, 0, None, 0, 0, 1, T, 0, B, mode, _lib.stream())
    torch.cuda.synchronize()
    wt = torch.from_numpy(w).view(M, 2, Q).permute(0, 2, 1).contiguous()       # (M,Q,2) conv weight
    ref = F.conv1d(x.cpu().double(), wt.double())                              # (B,M,T-1): index t-1
    got = _view(out, B, M, pitch).cpu().double()[:, :, 1:T]
    err = (got - ref).abs().max().item()
    print("err", err, "scale", ref.abs().max().item())
    assert err <= 3e-5 * ref.abs().max().item()


def _wgrad(rows, cols, t_lo, t_hi, chunk, B, mode, *head):
    """wn_wgrad into slabs + wn_reduce_slabs -> dense [rows, cols] tensor.  head = the wn_wgrad
    arguments up to and including relu_b."""
    ns = _lib.wgrad_slabs(t_lo, t_hi, chunk, B)
    n = rows * cols
    slab = torch.full((ns * n,), float("nan"), device=DEV)          # every element must be overwritten
    out = torch.full((n + 8,), float("nan"), device=DEV)
    call("wn_wgrad", *head, ptr(slab), cols, n, t_lo, t_hi, chunk, B, mode, _lib.stream())
    desc = torch.tensor([[0, 0, ns, n, 4, n]], dtype=torch.int64, device=DEV)
    call("wn_reduce_slabs", ptr(desc), 1, (n + 3) // 4, ptr(slab), ptr(out), _lib.stream())
    return out[4:4 + n].view(rows, cols)


def _res_ref(x, wf, wg, wd, d):
    f = F.conv1d(x, wf, dilation=d)
    g = F.conv1d(x, wg, dilation=d)
    z = torch.tanh(f) * torch.sigmoid(g)
    y = F.conv1d(z, wd) + x[:, :, d:]
    return f, g, z, y


def _fg_pack(wf, wg, CH, mode):
    D, R = wf.shape[0], wf.shape[1]
    w = np.zeros((2 * CH, 2 * CH), np.float32)
    for h, src in enumerate((wf, wg)):
        w[h * CH:h * CH + D, :R] = src[:, :, 0]
        w[h * CH:h * CH + D, CH:CH + R] = src[:, :, 1]
    return _packed(w, mode), w


@pytest.mark.parametrize("CH,R,D,d", [(32, 32, 32, 1), (32, 16, 16, 2), (64, 64, 64, 4), (64, 64, 64, 512), (64, 48, 40, 3)])
@pytest.mark.parametrize("mode", [_lib.F16X3, _lib.BF16X3])
def test_resblock_fwd(CH, R, D, d, mode):
    rng = np.random.default_rng(10 + d)
    B, T = 2, 1400
    pitch = 2048
    wf = (rng.standard_normal((D, R, 2)) * 0.3).astype(np.float32)
    wg = (rng.standard_normal((D, R, 2)) * 0.3).astype(np.float32)
    wd = (rng.standard_normal((R, D, 1)) * 0.3).astype(np.float32)
    pfg, _ = _fg_pack(wf, wg, CH, mode)
    wdp = np.zeros((CH, CH), np.float32)
    wdp[:R, :D] = wd[:, :, 0]
    pd = _packed(wdp, mode, chained=True)
    xin = _buf(B, CH, pitch, 1.0, 7)
    _view(xin, B, CH, pitch)[:, R:] = 0                       # padded channels are zero by contract
    xout, zout = _buf(B, CH, pitch), _buf(B, CH, pitch)
    off_in = 5
    t_lo, z_lo = off_in + d, off_in + d + 300
    call("wn_resblock_fwd", ptr(xin, SLACK), ptr(xout, SLACK), ptr(zout, SLACK), CH * pitch, CH * pitch, pitch,
         ptr(pfg), ptr(pd), None, None, None, D, R, CH, d, t_lo, T, z_lo, 1, None, 0, 0, 0, 0, 0, None, 0, None, 0, B, mode, _lib.stream())
    torch.cuda.synchronize()
    x = _view(xin, B, CH, pitch).cpu()[:, :R, off_in:T].double()
    f, g, z, y = _res_ref(x, torch.from_numpy(wf).double(), torch.from_numpy(wg).double(), torch.from_numpy(wd).double(), d)
    goty = _view(xout, B, CH, pitch).cpu().double()
    gotz = _view(zout, B, CH, pitch).cpu().double()
    ey = (goty[:, :R, t_lo:T] - y).abs().max().item()
    ez = (gotz[:, :D, z_lo:T] - z[:, :, z_lo - t_lo:]).abs().max().item()
    print("CH", CH, "d", d, "mode", mode, "err y", ey, "err z", ez, "scale", y.abs().max().item())
    tol = 1e-5 if mode == _lib.F16X3 else 2e-4               # observed 1.8e-6 / 8.3e-5
    assert ey <= tol * max(1.0, y.abs().max().item()) and ez <= tol
    assert goty[:, :, :t_lo].abs().max().item() == 0 and gotz[:, :, :z_lo].abs().max().item() == 0
    assert goty[:, R:].abs().sum().item() == 0 and gotz[:, D:].abs().sum().item() == 0


@pytest.mark.parametrize("CH,R,D,d", [(32, 32, 32, 1), (64, 64, 64, 8), (64, 48, 40, 2)])
def test_resblock_bwd_and_dx(CH, R, D, d):
    mf, mb = _lib.F16X3, _lib.BF16X3
    rng = np.random.default_rng(20 + d)
    B, T, pitch = 2, 900, 1536
    wf = (rng.standard_normal((D, R, 2)) * 0.3).astype(np.float32)
    wg = (rng.standard_normal((D, R, 2)) * 0.3).astype(np.float32)
    wd = (rng.standard_normal((R, D, 1)) * 0.3).astype(np.float32)
    pfg, _ = _fg_pack(wf, wg, CH, mf)
    wdT = np.zeros((CH, CH), np.float32)
    wdT[:D, :R] = wd[:, :, 0].T
    pdT = _packed(wdT, mb)
    wx = np.zeros((CH, 4 * CH), np.float32)                    # data-gradient pack [W1^T | W0^T]
    for h, src in enumerate((wf, wg)):
        wx[:R, h * CH:h * CH + D] = src[:, :, 1].T
        wx[:R, 2 * CH + h * CH:2 * CH + h * CH + D] = src[:, :, 0].T
    pX = _packed(wx, mb)
    xin = _buf(B, CH, pitch, 1.0, 8)
    _view(xin, B, CH, pitch)[:, R:] = 0
    dy = _buf(B, CH, pitch, 1e-3, 9)
    _view(dy, B, CH, pitch)[:, R:] = 0
    dz = _buf(B, CH, pitch, 1e-3, 10)
    _view(dz, B, CH, pitch)[:, D:] = 0
    dfg, zs, dx = _buf(B, 2 * CH, pitch, 1.0, 11), _buf(B, CH, pitch), _buf(B, CH, pitch)   # dfg pre-filled with junk
    off_in = 3
    t_lo, z_lo = off_in + d, off_in + d + 200
    st = _lib.stream()
    call("wn_resblock_bwd", ptr(xin, SLACK), ptr(dy, SLACK), ptr(dz, SLACK), ptr(dfg, SLACK), ptr(zs, SLACK),
         CH * pitch, CH * pitch, 2 * CH * pitch, CH * pitch, pitch, ptr(pfg), ptr(pdT), None, None, D, CH, d,
         t_lo, T, z_lo, None, 0, 0, 0, 0, 0, B, mf, mb, st)
    call("wn_chan_gemm", ptr(dfg, SLACK), ptr(dfg, SLACK), 2 * CH * pitch, pitch, t_lo, T, 0, d, 2 * CH // 32, 2 * CH // 32,
         ptr(pX), CH // 16, R, ptr(dx, SLACK), CH * pitch, pitch, 0, None, ptr(dy, SLACK), CH * pitch, pitch, t_lo,
         None, 0, 0, off_in, T, 0, B, mb, st)
    gW = _wgrad(2 * CH, 2 * CH, t_lo, T, 256, B, mb, ptr(dfg, SLACK), 2 * CH * pitch, pitch, 0, pitch, ptr(xin, SLACK),
                ptr(xin, SLACK), CH * pitch, pitch, -d, 0, pitch, CH // 16, 2 * CH // 16, 0)
    gD = _wgrad(CH, CH, t_lo, T, 256, B, mb, ptr(dy, SLACK), CH * pitch, pitch, 0, pitch, ptr(zs, SLACK), None,
                CH * pitch, pitch, 0, 0, pitch, CH // 16, CH // 16, 0)
    torch.cuda.synchronize()
    # reference through autograd in fp64
    x = _view(xin, B, CH, pitch).cpu()[:, :R, off_in:T].double().requires_grad_(True)
    twf, twg, twd = (torch.from_numpy(a).double().requires_grad_(True) for a in (wf, wg, wd))
    f, g, z, y = _res_ref(x, twf, twg, twd, d)
    gy = _view(dy, B, CH, pitch).cpu()[:, :R, t_lo:T].double()
    gz = torch.zeros_like(z)
    gz[:, :, z_lo - t_lo:] = _view(dz, B, CH, pitch).cpu()[:, :D, z_lo:T].double()
    f.retain_grad(); g.retain_grad()
    (y * gy).sum().backward(retain_graph=True, inputs=[x, twf, twg, twd, f, g])
    gx1, gwf1, gwg1, gwd1, gf1, gg1 = x.grad.clone(), twf.grad.clone(), twg.grad.clone(), twd.grad.clone(), f.grad.clone(), g.grad.clone()
    for t in (x, twf, twg, twd, f, g):
        t.grad = None
    (z * gz).sum().backward(inputs=[x, twf, twg, f, g])
    gx, gwf, gwg, gf, gg = gx1 + x.grad, gwf1 + twf.grad, gwg1 + twg.grad, gf1 + f.grad, gg1 + g.grad
    got_dfg = _view(dfg, B, 2 * CH, pitch).cpu().double()
    sc = gf.abs().max().item()
    e1 = (got_dfg[:, :D, t_lo:T] - gf).abs().max().item() / sc
    e2 = (got_dfg[:, CH:CH + D, t_lo:T] - gg).abs().max().item() / sc
    ez = (_view(zs, B, CH, pitch).cpu().double()[:, :D, t_lo:T] - z.detach()).abs().max().item()
    got_dx = _view(dx, B, CH, pitch).cpu().double()[:, :R, off_in:T]
    e3 = (got_dx - gx).abs().max().item() / gx.abs().max().item()
    gWc = gW.cpu().double()
    ref_wf = torch.cat([gwf[:, :, 0], gwf[:, :, 1]], 1)         # [D, 2R] laid out as tap0|tap1
    got_wf = torch.cat([gWc[:D, :R], gWc[:D, CH:CH + R]], 1)
    got_wg = torch.cat([gWc[CH:CH + D, :R], gWc[CH:CH + D, CH:CH + R]], 1)
    ref_wg = torch.cat([gwg[:, :, 0], gwg[:, :, 1]], 1)
    e4 = (got_wf - ref_wf).abs().max().item() / ref_wf.abs().max().item()
    e5 = (got_wg - ref_wg).abs().max().item() / ref_wg.abs().max().item()
    e6 = (gD.cpu().double()[:R, :D] - gwd1[:, :, 0]).abs().max().item() / gwd1.abs().max().item()
    print("df", e1, "dg", e2, "z", ez, "dx", e3, "dWf", e4, "dWg", e5, "dWd", e6)
    assert max(e1, e2, e3, e4, e5, e6) < 1e-4 and ez < 1e-5      # observed 8.1e-6 / 2.1e-6


def test_wgrad_compact_relu():
    mode = _lib.BF16X3
    rng = np.random.default_rng(5)
    B, M, N, pitch, T, W = 2, 48, 80, 1024, 900, 640
    lo = T - W
    a = torch.from_numpy(rng.standard_normal((B, M, W)).astype(np.float32) * 1e-4).to(DEV)   # compact A
    a_pad = torch.cat([a.reshape(-1), torch.zeros(512, device=DEV)])
    b = _buf(B, N, pitch, 1.0, 6)
    c = _wgrad(M, N, lo, T, 128, B, mode, ptr(a_pad), M * W, W, -lo, W, ptr(b, SLACK), None, N * pitch, pitch, 0, 0, pitch,
               N // 16, M // 16, 1)
    torch.cuda.synchronize()
    bb = _view(b, B, N, pitch).cpu().double()[:, :, lo:T].clamp(min=0)
    ref = torch.einsum("bmt,bnt->mn", a.cpu().double(), bb)
    err = (c.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    print("wgrad rel err", err)
    assert err < 5e-5                                            # observed 3.4e-6


def test_chunk_softmax_fwd_bwd_ce():
    from tests.helpers import load_npz
    from tests.tools_cfg import g3_inputs
    d = load_npz("g3_softmax.npz")
    for w, x in g3_inputs().items():
        xt = torch.from_numpy(x).to(DEV)
        y = torch.empty(w, 256, device=DEV)
        call("wn_chunk_softmax256_fwd", ptr(xt), ptr(y), w, _lib.stream())
        np.testing.assert_allclose(y.cpu().numpy(), d["y_w%d" % w], atol=2e-7, rtol=1e-5)
    rng = np.random.default_rng(9)
    n = 1037
    x = torch.from_numpy((3 * rng.standard_normal((n, 256))).astype(np.float32))
    tgt = torch.from_numpy(rng.integers(0, 256, n).astype(np.int64))
    xr = x.clone().double().requires_grad_(True)
    p = torch.softmax(xr, 1)
    loss = F.cross_entropy(p, tgt)
    loss.backward()
    xd, td = x.to(DEV), tgt.to(DEV)
    probs, dx = torch.empty(n, 256, device=DEV), torch.empty(n, 256, device=DEV)
    part = torch.zeros(_lib.CE_NUM_PARTIALS, device=DEV)
    call("wn_chunk_softmax256_ce", ptr(xd), ptr(td), ptr(probs), ptr(dx), ptr(part), n, 1.0 / n, _lib.stream())
    assert abs(part.sum().item() - loss.item()) < 1e-5
    assert (probs.cpu().double() - p.detach()).abs().max().item() < 1e-6
    assert (dx.cpu().double() - xr.grad).abs().max().item() < 1e-9 + 1e-5 * xr.grad.abs().max().item()
    # separate backward kernel: dx = y*(dy - <dy,y>)
    dy = torch.from_numpy(rng.standard_normal((n, 256)).astype(np.float32)).to(DEV)
    dx2 = torch.empty(n, 256, device=DEV)
    call("wn_chunk_softmax256_bwd", ptr(probs), ptr(dy), ptr(dx2), n, _lib.stream())
    pr = p.detach()
    ref = pr * (dy.cpu().double() - (dy.cpu().double() * pr).sum(1, keepdim=True))
    assert (dx2.cpu().double() - ref).abs().max().item() < 1e-5
    # a target outside [0, 256) (nn.CrossEntropyLoss raises for it) must not pass silently: NaN loss, NaN row gradient,
    # the other rows untouched
    for bad in (256, -1, 1 << 40):
        tb = tgt.clone()
        tb[17] = bad
        dxb = torch.empty(n, 256, device=DEV)
        call("wn_chunk_softmax256_ce", ptr(xd), ptr(tb.to(DEV)), None, ptr(dxb), ptr(part), n, 1.0 / n, _lib.stream())
        assert torch.isnan(part.sum()).item() and torch.isnan(dxb[17]).all().item()
        keep = torch.ones(n, dtype=torch.bool)
        keep[17] = False
        assert torch.equal(dxb.cpu()[keep], dx.cpu()[keep])


def test_onehot_and_mulaw():
    from oracle import intops
    from tests.helpers import load_npz
    rng = np.random.default_rng(11)
    codes = rng.integers(0, 256, size=(3, 1025)).astype(np.int32)
    cd = torch.from_numpy(codes).to(DEV)
    for scr in (1, 0):
        out = torch.empty(3, 256, 1025, device=DEV)
        call("wn_onehot", ptr(cd), ptr(out), 3, 256, 1025, scr, _lib.stream())
        ref = np.stack([(intops.one_hot_scrambled if scr else intops.one_hot_proper)(r) for r in codes])
        assert np.array_equal(out.cpu().numpy(), ref)              # bit-exact
    g4 = load_npz("g4_data.npz")
    piece = g4["oh_piece3"]
    out = torch.empty(1, 256, len(piece), device=DEV)
    call("wn_onehot", ptr(torch.from_numpy(piece.astype(np.int32)).to(DEV)), ptr(out), 1, 256, len(piece), 1, _lib.stream())
    assert np.array_equal(np.flatnonzero(out.cpu().numpy().reshape(-1)), g4["oh_flatpos3"])
    g5 = load_npz("g5_mulaw.npz")
    x = torch.from_numpy(g5["x"]).to(DEV)
    thr = torch.from_numpy(g5["thresholds"]).to(DEV)
    c = torch.empty(x.numel(), dtype=torch.uint8, device=DEV)
    call("wn_mulaw_encode_tbl", ptr(x), ptr(thr), ptr(c), x.numel(), _lib.stream())
    assert np.array_equal(c.cpu().numpy(), g5["codes"])             # bit-exact vs the reference encoder
    tab = torch.from_numpy(g5["decode_table"]).to(DEV)
    a = torch.empty(x.numel(), device=DEV)
    call("wn_mulaw_decode_lut", ptr(c), ptr(tab), ptr(a), x.numel(), _lib.stream())
    assert np.array_equal(a.cpu().numpy(), g5["decode_table"][g5["codes"]])
    # encode(decode(k)) == k round trip on device
    ks = torch.arange(256, dtype=torch.uint8, device=DEV)
    a2 = torch.empty(256, device=DEV)
    c2 = torch.empty(256, dtype=torch.uint8, device=DEV)
    call("wn_mulaw_decode_lut", ptr(ks), ptr(tab), ptr(a2), 256, _lib.stream())
    call("wn_mulaw_encode_tbl", ptr(a2), ptr(thr), ptr(c2), 256, _lib.stream())
    assert torch.equal(c2, ks)


def test_mulaw_other_channel_counts_on_device():
    """audio_func.mu_law_encode / mu_law_decode with quantization_channels = 64 / 100 / 512 (the reference takes any,
    audio_func.py:5,24): bit-exact against the reference's own known answers (tests/golden/g5q_mulaw.npz), code boundaries
    included; encode(decode(k)) == k; tensors of any shape and device come back on their device."""
    from music_amd import audio_func as af
    from tests.helpers import load_npz
    d = load_npz("g5q_mulaw.npz")
    for q in (64, 100, 512):
        x = torch.from_numpy(d["x%d" % q])
        c = af.mu_law_encode(x.to(DEV), q)
        assert c.dtype == torch.int64 and c.is_cuda and np.array_equal(c.cpu().numpy(), d["codes%d" % q])
        a = af.mu_law_decode(torch.arange(q), q)
        assert not a.is_cuda and np.array_equal(a.numpy(), d["decode%d" % q])
        assert torch.equal(af.mu_law_encode(a, q), torch.arange(q))
        c2 = af.mu_law_encode(x.view(-1, 1)[:100].expand(100, 3), q)            # non-contiguous, 2-D
        assert c2.shape == (100, 3) and torch.equal(c2[:, 0], c[:100].cpu())
    with pytest.raises(ValueError):
        af.mu_law_encode(torch.zeros(4), 1)


def test_adam_flat_matches_torch():
    n = 100003
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(n, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for t in range(1, 6):
        gr = torch.randn(n, generator=g) * 0.01
        p_ref.grad = gr.clone()
        opt.step()
        call("wn_adam_flat", ptr(p), ptr((gr * 4).to(DEV)), ptr(m), ptr(v), n, 1e-3, 0.9, 0.999, 1e-8,
             1 - 0.9 ** t, 1 - 0.999 ** t, 0.25, _lib.stream())
    assert (p.cpu() - p_ref.detach()).abs().max().item() < 2e-6


def test_wgrad_big_lds_path():
    """Outputs >= 256 x 256 take the LDS-shared workgroup kernel (wgrad_big_k): 256 x 304 with a
    ragged last column group, two taps, relu on B, masked chunk ends."""
    mode = _lib.BF16X3
    rng = np.random.default_rng(15)
    B, M, NB, pitch, T = 2, 256, 160, 2304, 1900          # C = [256][2 taps * 160 = 320]... use 152 -> 19 tiles
    NB = 152
    a = _buf(B, M, pitch, 1e-3, 21)
    bsrc = _buf(B, NB + 8, pitch, 1.0, 22)
    t_lo, d = 700, 5
    c = _wgrad(M, 2 * NB + 16, t_lo, T, 512, B, mode, ptr(a, SLACK), M * pitch, pitch, 0, pitch, ptr(bsrc, SLACK),
               ptr(bsrc, SLACK), (NB + 8) * pitch, pitch, -d, 0, pitch, (NB + 8) // 16, M // 16, 1)
    torch.cuda.synchronize()
    aa = _view(a, B, M, pitch).cpu().double()[:, :, t_lo:T]
    bb = _view(bsrc, B, NB + 8, pitch).cpu().double().clamp(min=0)
    ref0 = torch.einsum("bmt,bnt->mn", aa, bb[:, :, t_lo - d:T - d])
    ref1 = torch.einsum("bmt,bnt->mn", aa, bb[:, :, t_lo:T])
    ref = torch.cat([ref0, ref1], 1)
    err = (c.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    print("wgrad big rel err", err)
    assert err < 5e-5                                            # observed 4.6e-6


@pytest.mark.parametrize("mode,le,q", [(1, 7, 41), (1, 31, 16), (2, 31, 0), (2, 5, 0), (2, 64, 0), (2, 100, 0)])
def test_cond_grad_bucket_sums(mode, le, q):
    """wn_cond_grad (conditioning-table gradient, model1.py:227-247 backward): bucket sums of every row
    over time, stretch (bucket = (t-t_lo)//q, clamped) and tile (bucket = (t-t_lo) % le) layouts,
    including the le > 64 path; deterministic (two runs bit-identical)."""
    B, rows, pitch, t_lo, t_hi = 2, 24, 1200, 37, 37 + 1103
    if mode == 1:
        t_hi = t_lo + le * q + 13            # a ragged tail that the last bucket absorbs
    src = _buf(B, rows, pitch, fill=1.0, seed=5)
    x = _view(src, B, rows, pitch).cpu().double().numpy()
    out = torch.full((B, rows, le), 7.0, dtype=torch.float32, device=DEV)
    for _ in range(2):
        call("wn_cond_grad", ptr(src, SLACK), rows * pitch, pitch, rows, t_lo, t_hi, mode, le, max(q, 1),
             ptr(out), rows * le, le, B, _lib.stream())
        if _ == 0:
            first = out.clone()
    assert torch.equal(first, out)
    L = t_hi - t_lo
    tr = np.arange(L)
    ix = np.minimum(tr // q, le - 1) if mode == 1 else tr % le
    ref = np.zeros((B, rows, le))
    for j in range(le):
        ref[:, :, j] = x[:, :, t_lo:t_hi][:, :, ix == j].sum(-1)
    err = np.abs(out.cpu().numpy() - ref).max()
    print("cond_grad mode %d le %d: max err %.2e" % (mode, le, err))
    assert err < 5e-5                                            # observed <= 5.9e-6


def test_conditioned_block_entry_points_refuse_bad_arguments():
    """The conditioned form of wn_resblock_bwd_pq needs the bucket bytes and at most 32 buckets, its bucket-sum slabs need a
    table, the reduce and wn_cond_expand check their shapes: status -4 and a message, nothing launched."""
    buf = torch.zeros(1 << 16, device=DEV)
    i8 = torch.zeros(4096, dtype=torch.uint8, device=DEV)
    pk = torch.zeros(1 << 16, dtype=torch.int16, device=DEV)
    pq = lambda cond, le, idx, cslab: call(
        "wn_resblock_bwd_pq", ptr(buf), None, None, 0, 0, ptr(buf), ptr(buf), ptr(buf), 64 * 256, 64 * 256, 256, ptr(pk), ptr(pk), ptr(pk),
        64, 1, 8, 200, 8, ptr(buf), ptr(buf), cond, 128 * 8, 8, le, idx, cslab, 0, 0, 1, _lib.F16X3, _lib.BF16X3, _lib.stream())
    for args, what in (((ptr(buf), 8, None, None), "cond_idx"), ((ptr(buf), 33, ptr(i8), None), "buckets"),
                       ((None, 8, ptr(i8), ptr(buf)), "cslab without cond")):
        with pytest.raises(_lib.WavenetHipError, match=what):
            pq(*args)
    import ctypes
    off, tlo = (ctypes.c_int64 * 1)(0), (ctypes.c_int * 1)(8)
    with pytest.raises(_lib.WavenetHipError, match="bad argument"):
        call("wn_resblock_bwd_pq_cond_reduce", ptr(buf), off, tlo, 1, 200, 1, 40, ptr(buf), 0, 128 * 40, 40, _lib.stream())
    with pytest.raises(_lib.WavenetHipError, match="bad argument"):
        call("wn_cond_expand", ptr(buf), 8 * 8, 8, 8, 8, 200, 3, 8, 1, ptr(buf), 8 * 256, 256, 1, _lib.stream())


@pytest.mark.parametrize("mode,le,q", [(1, 7, 41), (1, 31, 16), (2, 31, 0), (2, 5, 0), (2, 100, 0)])
def test_cond_expand_matches_the_index_expression(mode, le, q):
    """wn_cond_expand (the conditioning term over time, model1.py:227-247 `_conditon`): out[b][row][t] = tab[b][row][idx(t)],
    stretch (idx = (t - t_lo) // q, clamped) and tile (idx = (t - t_lo) % le) rules, a length that is no multiple of four,
    nothing written outside [t_lo, t_hi)."""
    B, rows, pitch, t_lo = 2, 10, 1300, 37
    t_hi = t_lo + (le * q + 13 if mode == 1 else 1103)
    tab = torch.randn(B, rows, le, device=DEV)
    out = torch.full((B, rows, pitch), 7.0, dtype=torch.float32, device=DEV)
    call("wn_cond_expand", ptr(tab), rows * le, le, rows, t_lo, t_hi, mode, le, max(q, 1), ptr(out), rows * pitch, pitch, B,
         _lib.stream())
    tr = torch.arange(t_hi - t_lo, device=DEV)
    ix = torch.clamp(tr // q, max=le - 1) if mode == 1 else tr % le
    assert torch.equal(out[:, :, t_lo:t_hi], tab[:, :, ix])
    assert (out[:, :, :t_lo] == 7.0).all() and (out[:, :, t_hi:] == 7.0).all()


@pytest.mark.parametrize("scrambled", [True, False], ids=["scrambled", "proper"])
@pytest.mark.parametrize("ch,T,B", [(64, 1000, 3), (32, 517, 2), (64, 16000, 2)])
def test_causal_wgrad_from_codes_equals_dense_product(scrambled, ch, T, B):
    """wn_causal_wgrad_codes (scatter of dx columns selected by the integer codes) against the definition
    dW[r][q][tap] = sum_{b,t in [1,T)} dx[b][r][t] * in[b][q][t-1+tap] evaluated in float64 on the dense one-hot the
    same codes give, both layouts; repeated codes (silence), the first / last columns and stale data outside [1, T)."""
    from oracle import intops
    rng = np.random.default_rng(T + ch)
    codes = rng.integers(0, 256, size=(B, T)).astype(np.int32)
    codes[0, : T // 3] = 128                                  # a run of one class: many ones in one row
    codes[-1, -5:] = [0, 255, 255, 0, 7]
    fn = intops.one_hot_scrambled if scrambled else intops.one_hot_proper
    dense = torch.from_numpy(np.stack([fn(r) for r in codes])).double()          # (B, 256, T)
    pitch = ((T + 255) // 256) * 256 + 512
    dx = torch.from_numpy(rng.standard_normal((B, ch, pitch)).astype(np.float32))     # garbage outside [1, T) on purpose
    want = torch.zeros(ch, 512, dtype=torch.float64)
    d = dx.double()
    for b in range(B):
        want[:, :256] += d[b, :, 1:T] @ dense[b, :, 0:T - 1].t()                 # tap 0: in[t-1]
        want[:, 256:] += d[b, :, 1:T] @ dense[b, :, 1:T].t()                     # tap 1: in[t]
    ns = _lib.causal_codes_slabs(T, B)
    slab = torch.full((ns, ch, 512), float("nan"), device=DEV)
    cd, dxd = torch.from_numpy(codes).to(DEV), dx.to(DEV)
    call("wn_causal_wgrad_codes", ptr(cd), 1 if scrambled else 0, ptr(dxd), None, 0, 0, ch * pitch, pitch, ch, 256, T, B, ptr(slab), _lib.stream())
    got = slab.double().sum(0).cpu()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-5, err
    slab2 = torch.empty_like(slab)
    call("wn_causal_wgrad_codes", ptr(cd), 1 if scrambled else 0, ptr(dxd), None, 0, 0, ch * pitch, pitch, ch, 256, T, B, ptr(slab2), _lib.stream())
    assert torch.equal(slab, slab2)                           # bit-reproducible
    # the data gradient as an unshifted pair (what wn_resblock_bwd_pq hands on): dx[s] = P[s] (s >= p_lo) + Q[s + dn]
    dn, p_lo = 3, 5
    P = torch.from_numpy(rng.standard_normal((B, ch, pitch)).astype(np.float32))
    Qv = torch.from_numpy(rng.standard_normal((B, ch, pitch)).astype(np.float32))
    whole = torch.zeros(B, ch, pitch)
    whole[:, :, p_lo:T] = P[:, :, p_lo:T]
    whole[:, :, 1:T - dn] += Qv[:, :, 1 + dn:T]
    Pd, Qd, wd = P.to(DEV), Qv.to(DEV), whole.to(DEV)
    call("wn_causal_wgrad_codes", ptr(cd), 1 if scrambled else 0, ptr(wd), None, 0, 0, ch * pitch, pitch, ch, 256, T, B, ptr(slab), _lib.stream())
    call("wn_causal_wgrad_codes", ptr(cd), 1 if scrambled else 0, ptr(Pd), ptr(Qd), dn, p_lo, ch * pitch, pitch, ch, 256, T, B, ptr(slab2), _lib.stream())
    assert torch.equal(slab, slab2)


@pytest.mark.parametrize("scrambled", [True, False], ids=["scrambled", "proper"])
@pytest.mark.parametrize("ch,R,T,B,bias", [(64, 64, 1000, 3, False), (32, 20, 517, 2, True), (64, 48, 16000, 2, True)])
def test_causal_forward_from_codes_equals_conv(scrambled, ch, R, T, B, bias):
    """wn_causal_fwd_codes (gather of weight columns selected by the codes) against F.conv1d in float64 on the dense
    one-hot the same codes give: x0 on [1, T), both layouts, runs of one class, padded channels, bias."""
    from oracle import intops
    rng = np.random.default_rng(T + R)
    codes = rng.integers(0, 256, size=(B, T)).astype(np.int32)
    codes[0, : T // 3] = 128
    codes[-1, -5:] = [0, 255, 255, 0, 7]
    fn = intops.one_hot_scrambled if scrambled else intops.one_hot_proper
    dense = torch.from_numpy(np.stack([fn(r) for r in codes])).double()
    w = torch.from_numpy(rng.standard_normal((R, 256, 2)).astype(np.float32))
    bvec = torch.from_numpy(rng.standard_normal(R).astype(np.float32)) if bias else None
    want = F.conv1d(dense, w.double(), bvec.double() if bias else None)              # (B, R, T-1): column j = time j+1
    wt = torch.zeros(2, 256, ch)
    wt[:, :, :R] = w.permute(2, 1, 0)
    pitch = ((T + 255) // 256) * 256 + 512
    x0 = torch.full((B, ch, pitch), 7.0, device=DEV)                                 # canary outside [1, T) x real rows
    cd, wtd, bd = torch.from_numpy(codes).to(DEV), wt.to(DEV), (bvec.to(DEV) if bias else None)     # (kept alive past the call)
    call("wn_causal_fwd_codes", ptr(cd), 1 if scrambled else 0, ptr(wtd), ptr(bd), R, ptr(x0), ch * pitch, pitch, ch, 256, T, B,
         _lib.stream())
    got = x0.cpu()
    err = (got[:, :R, 1:T].double() - want).abs().max().item()
    assert err < 1e-5, err
    assert (got[:, :, 0] == 7.0).all() and (got[:, :, T:] == 7.0).all() and (got[:, R:] == 7.0).all()


@pytest.mark.parametrize("bf16", [0, 1])
def test_split16_is_round_to_nearest_hi_and_lo(bf16):
    """The operand split every MFMA product goes through (wn_common.h split2, hand-written instructions): hi = round16(x),
    lo = round16(x - hi), bit for bit what torch's conversions give, over normal, tiny (f16-subnormal), huge and special
    values, odd and even lengths."""
    g = torch.Generator().manual_seed(5)
    parts = [torch.randn(4001, generator=g) * s for s in (1.0, 1e-3, 1e-6, 3e-8, 1e3, 6e4)]
    parts.append(torch.tensor([0.0, -0.0, 1.0, -1.0, 65504.0, -65504.0, 6.1e-5, 5.96e-8, 2.0 ** -25, 1.0 + 2.0 ** -11,
                               1.0 + 2.0 ** -12, 1.0 + 3 * 2.0 ** -12, float("inf"), -float("inf"), 1e38, -1e38, 7e4]))
    x = torch.cat(parts).to(DEV)
    n = x.numel()
    assert n % 2 == 1
    t16 = torch.bfloat16 if bf16 else torch.float16
    hi = torch.zeros(n, dtype=torch.int16, device=DEV)
    lo = torch.zeros(n, dtype=torch.int16, device=DEV)
    call("wn_split16", ptr(x), ptr(hi), ptr(lo), n, bf16, _lib.stream())
    torch.cuda.synchronize()
    xr = x.cpu()
    h_ref = xr.to(t16)
    l_ref = (xr - h_ref.float()).to(t16)
    h_got, l_got = hi.cpu().view(t16), lo.cpu().view(t16)
    assert torch.equal(h_got.view(torch.int16), h_ref.view(torch.int16))
    fin = torch.isfinite(h_ref.float())
    assert torch.equal(l_got.view(torch.int16)[fin], l_ref.view(torch.int16)[fin])
    # x beyond the 16-bit range: hi = +-inf and lo = x - hi = -+inf, or NaN where x itself is infinite - as the reference
    nan_ref = torch.isnan(l_ref.float())
    assert torch.equal(torch.isnan(l_got.float()), nan_ref)
    assert torch.equal(l_got.view(torch.int16)[~nan_ref], l_ref.view(torch.int16)[~nan_ref])
    # and the pair carries the value to 2^-22 (f16) / 2^-17 (bf16) of its magnitude where nothing underflows
    ok = fin & (xr.abs() > 0.5)                    # (an f16 lo below 6e-5 is subnormal: absolute, not relative, precision)
    rel = ((h_got.float() + l_got.float() - xr).abs() / xr.abs())[ok].max().item()
    assert rel < (2.0 ** -16 if bf16 else 2.0 ** -21), rel


@pytest.mark.parametrize("q", [1, 7, 64, 100, 256, 300, 1000])
def test_chunk_softmax_any_width_fwd_bwd_ce(q):
    """wn_chunk_softmax_fwd / _bwd / _ce (the general plan's softmax, any row length) against torch in float64; for q = 256
    they must agree with the specialised 256-wide kernels to the last few ulps."""
    n = 1237
    g = torch.Generator().manual_seed(q)
    x = (torch.randn(n, q, generator=g) * 3).to(DEV)
    dy = torch.randn(n, q, generator=g).to(DEV) * 1e-3
    tgt = torch.randint(0, q, (n,), generator=g).to(DEV)
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    st = _lib.stream()
    call("wn_chunk_softmax_fwd", ptr(x), ptr(y), n, q, st)
    y64 = torch.softmax(x.double(), 1)
    assert (y.double() - y64).abs().max().item() < 5e-7
    call("wn_chunk_softmax_bwd", ptr(y), ptr(dy), ptr(dx), n, q, st)
    want = y64 * (dy.double() - (dy.double() * y64).sum(1, keepdim=True))
    assert (dx.double() - want).abs().max().item() < 1e-9
    # fused softmax + CrossEntropyLoss on the probabilities (wavenet/train.py:146,179) + both backward steps
    probs = torch.empty_like(x)
    part = torch.zeros(_lib.CE_NUM_PARTIALS, dtype=torch.float32, device=DEV)
    call("wn_chunk_softmax_ce", ptr(x), ptr(tgt), ptr(probs), ptr(dx), ptr(part), n, q, 1.0 / n, st)
    xr = x.double().clone().requires_grad_(True)
    loss = F.cross_entropy(torch.softmax(xr, 1), tgt)
    loss.backward()
    assert abs(part.sum().item() - loss.item()) < 1e-5
    assert (probs.double() - y64).abs().max().item() < 5e-7
    assert (dx.double() - xr.grad).abs().max().item() <= 1e-4 * xr.grad.abs().max().item() + 1e-12
    if q == 256:
        y2 = torch.empty_like(x)
        call("wn_chunk_softmax256_fwd", ptr(x), ptr(y2), n, st)
        assert (y - y2).abs().max().item() < 1e-6
    # a target outside [0, q) poisons the loss and that row's gradient (nn.CrossEntropyLoss raises for it)
    bad = tgt.clone()
    bad[5] = q
    call("wn_chunk_softmax_ce", ptr(x), ptr(bad), None, ptr(dx), ptr(part), n, q, 1.0 / n, st)
    assert torch.isnan(part.sum()) and torch.isnan(dx[5]).all() and torch.isfinite(dx[6]).all()


def test_gate_kernels_vs_torch():
    """wn_gate_fwd / wn_gate_bwd (general plan): z = tanh f * sigmoid g on [t_lo, t_hi) of the first `rows` rows, untouched
    elsewhere; the derivative of SURVEY Appendix B."""
    b, dp, rows, pitch, t_lo, t_hi = 2, 64, 50, 1024, 37, 901
    fg = _buf(b, 2 * dp, pitch, fill=2.0, seed=1)
    dz = _buf(b, dp, pitch, fill=1.0, seed=2)
    z = _buf(b, dp, pitch)
    dfg = _buf(b, 2 * dp, pitch)
    z.fill_(7.0)
    dfg.fill_(7.0)
    st = _lib.stream()
    call("wn_gate_fwd", ptr(fg, SLACK), 2 * dp * pitch, dp, rows, ptr(z, SLACK), dp * pitch, pitch, t_lo, t_hi, b, st)
    call("wn_gate_bwd", ptr(fg, SLACK), 2 * dp * pitch, dp, rows, ptr(dz, SLACK), dp * pitch, ptr(dfg, SLACK), 2 * dp * pitch, pitch,
         t_lo, t_hi, b, st)
    f64 = _view(fg, b, 2 * dp, pitch).double()
    f, g = f64[:, :dp], f64[:, dp:]
    th, sg = torch.tanh(f), torch.sigmoid(g)
    zv, dv, gz = _view(z, b, dp, pitch), _view(dfg, b, 2 * dp, pitch), _view(dz, b, dp, pitch).double()
    sl = (slice(None), slice(0, rows), slice(t_lo, t_hi))
    assert (zv[sl].double() - (th * sg)[sl]).abs().max().item() < 5e-7
    assert (dv[:, :dp][sl].double() - (gz * sg * (1 - th * th))[sl]).abs().max().item() < 2e-6
    assert (dv[:, dp:][sl].double() - (gz * th * sg * (1 - sg))[sl]).abs().max().item() < 2e-6
    # nothing outside the range / rows is written
    assert (zv[:, rows:] == 7.0).all() and (zv[:, :, :t_lo] == 7.0).all() and (zv[:, :, t_hi:] == 7.0).all()
    assert (dv[:, rows:dp] == 7.0).all() and (dv[:, dp + rows:] == 7.0).all() and (dv[:, :, t_hi:] == 7.0).all()
