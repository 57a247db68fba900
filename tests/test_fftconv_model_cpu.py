"""CPU model of the frequency-domain convolution of csrc/conv_fft.hip (overlap-save forward, overlap-add data
gradient, spectral weight gradient on 32x32 tiles with 17 kept kx bins), checked against torch's conv2d autograd --
the arithmetic the reference's ResidualBlock / ConvBlock layers execute (AE_model_unet.py:45-77).  This pins the
tiling / conjugation / Hermitian-weight algebra the HIP kernels implement, without a GPU, and records the accuracy
argument of DESIGN.md §2.4: in fp32 the tiled FFT result is at least as close to an fp64 convolution as a direct fp32 sum.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

N = 32


def _tiles(H, W, k):
    T = N - k + 1
    return T, -(-H // T), -(-W // T)


def fft_conv_forward(x, w, reflect=False, dtype=np.float64):
    """x [C,H,W], w [Co,C,k,k] -> y [Co,H,W] (cross-correlation, pad k//2)."""
    C, H, W = x.shape
    Co, _, k, _ = w.shape
    p = k // 2
    T, ty_n, tx_n = _tiles(H, W, k)
    xp = np.pad(x, ((0, 0), (p, p + N), (p, p + N)), mode="constant")
    if reflect:
        xr = np.pad(x, ((0, 0), (p, p), (p, p)), mode="reflect")
        xp = np.pad(xr, ((0, 0), (0, N), (0, N)), mode="constant")
    wf = np.conj(np.fft.rfft2(w.astype(dtype), s=(N, N))).astype(np.complex64 if dtype == np.float32 else np.complex128)
    y = np.zeros((Co, ty_n * T, tx_n * T), dtype)
    for ty in range(ty_n):
        for tx in range(tx_n):
            patch = xp[:, ty * T:ty * T + N, tx * T:tx * T + N].astype(dtype)
            xf = np.fft.rfft2(patch).astype(wf.dtype)
            yf = np.einsum("cyx,ncyx->nyx", xf, wf)
            y[:, ty * T:(ty + 1) * T, tx * T:(tx + 1) * T] = np.fft.irfft2(yf, s=(N, N))[:, :T, :T]
    return y[:, :H, :W]


def fft_conv_backward(x, w, gy):
    """zero-padded layer: returns (dx, dw) from ONE transform of the dy tiles (no halo)."""
    C, H, W = x.shape
    Co, _, k, _ = w.shape
    p = k // 2
    T, ty_n, tx_n = _tiles(H, W, k)
    xp = np.pad(x, ((0, 0), (p, p + N), (p, p + N)))
    gyp = np.pad(gy, ((0, 0), (0, N), (0, N)))
    wf = np.fft.rfft2(w, s=(N, N))                       # un-conjugated: convolution
    dxp = np.zeros((C, ty_n * T + N, tx_n * T + N))
    dwf = np.zeros((Co, C, N, N // 2 + 1), np.complex128)
    for ty in range(ty_n):
        for tx in range(tx_n):
            d = np.zeros((Co, N, N))
            d[:, :T, :T] = gyp[:, ty * T:(ty + 1) * T, tx * T:(tx + 1) * T]
            df = np.fft.rfft2(d)
            ef = np.einsum("nyx,ncyx->cyx", df, wf)
            dxp[:, ty * T:ty * T + N, tx * T:tx * T + N] += np.fft.irfft2(ef, s=(N, N))     # overlap-add at offset -p
            xf = np.fft.rfft2(xp[:, ty * T:ty * T + N, tx * T:tx * T + N])
            dwf += np.conj(df)[:, None] * xf[None]
    dw = np.fft.irfft2(dwf, s=(N, N))[:, :, :k, :k]
    return dxp[:, p:p + H, p:p + W], dw


@pytest.mark.parametrize("C,Co,k,H,W", [(3, 4, 9, 40, 70), (4, 3, 7, 26, 52), (2, 2, 5, 33, 31), (2, 3, 3, 9, 12)])
def test_tiled_fft_conv_matches_conv2d(C, Co, k, H, W):
    rng = np.random.default_rng(k * 100 + H)
    x, w, gy = rng.standard_normal((C, H, W)), rng.standard_normal((Co, C, k, k)), rng.standard_normal((Co, H, W))
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    yt = F.conv2d(xt[None], wt, padding=k // 2)[0]
    yt.backward(torch.tensor(gy))
    np.testing.assert_allclose(fft_conv_forward(x, w), yt.detach().numpy(), atol=1e-10)
    dx, dw = fft_conv_backward(x, w, gy)
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(dw, wt.grad.numpy(), atol=1e-9)
    yr = F.conv2d(F.pad(torch.tensor(x)[None], (k // 2,) * 4, mode="reflect"), torch.tensor(w))[0]
    np.testing.assert_allclose(fft_conv_forward(x, w, reflect=True), yr.numpy(), atol=1e-10)


def test_fp32_tiled_fft_is_as_accurate_as_a_direct_fp32_sum():
    """9x9 on 64 channels (the level-0 residual layer): error against fp64 of the fp32 tiled FFT vs torch's fp32 conv."""
    rng = np.random.default_rng(0)
    C, k, H, W = 64, 9, 48, 72
    x = rng.standard_normal((C, H, W)).astype(np.float32)
    w = (rng.standard_normal((C, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    ref = F.conv2d(torch.tensor(x, dtype=torch.float64)[None], torch.tensor(w, dtype=torch.float64), padding=4)[0].numpy()
    direct = F.conv2d(torch.tensor(x)[None], torch.tensor(w), padding=4)[0].numpy()
    tiled = fft_conv_forward(x, w, dtype=np.float32)
    scale = np.abs(ref).max()
    e_direct, e_fft = np.abs(direct - ref).max() / scale, np.abs(tiled - ref).max() / scale
    assert e_fft < 2e-6 and e_fft < 4 * e_direct, (e_fft, e_direct)
