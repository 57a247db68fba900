"""CPU model of the Winograd F(2x2,3x3) convolution of csrc/conv_wino.hip (forward, data gradient as the same pipeline on
dy with flipped / role-swapped taps, weight gradient dW = G^T [sum (B^T d B) (.) (A dy A^T)] G), checked against torch's
conv2d autograd -- the arithmetic of the reference's 512-channel ResidualBlocks (AE_model_unet.py:45-57).  Pins the
transform matrices and the tile bookkeeping the HIP kernels implement, and the accuracy argument of DESIGN.md §2.5.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def _tiles(x, dtype):
    C, H, W = x.shape
    ty, tx = -(-H // 2), -(-W // 2)
    xp = np.zeros((C, 2 * ty + 2, 2 * tx + 2), dtype)
    xp[:, 1:H + 1, 1:W + 1] = x
    d = np.stack([xp[:, 2 * a:2 * a + 4, 2 * b:2 * b + 4] for a in range(ty) for b in range(tx)])     # [T,C,4,4]
    return d, ty, tx


def wino_forward(x, w, dtype=np.float64):
    C, H, W = x.shape
    d, ty, tx = _tiles(x.astype(dtype), dtype)
    V = np.einsum("ij,tcjk,lk->iltc", BT.astype(dtype), d, BT.astype(dtype))
    U = np.einsum("ij,ncjk,lk->ilnc", G.astype(dtype), w.astype(dtype), G.astype(dtype))
    M = np.einsum("iltc,ilnc->iltn", V, U).astype(dtype)
    y4 = np.einsum("ij,jktn,lk->tnil", AT.astype(dtype), M, AT.astype(dtype))                      # [T,N,2,2]
    y = y4.reshape(ty, tx, -1, 2, 2).transpose(2, 0, 3, 1, 4).reshape(-1, 2 * ty, 2 * tx)
    return y[:, :H, :W], V


def wino_wgrad(V, gy):
    N, H, W = gy.shape
    ty, tx = -(-H // 2), -(-W // 2)
    gp = np.zeros((N, 2 * ty, 2 * tx))
    gp[:, :H, :W] = gy
    g4 = gp.reshape(N, ty, 2, tx, 2).transpose(1, 3, 0, 2, 4).reshape(ty * tx, N, 2, 2)
    Dv = np.einsum("ji,tnjk,kl->iltn", AT, g4, AT)            # A dy A^T with A = AT^T
    P = np.einsum("iltn,iltc->ilnc", Dv, V)
    return np.einsum("ji,jknc,kl->ncil", G, P, G)             # G^T P G


@pytest.mark.parametrize("C,N,H,W", [(3, 4, 8, 26), (4, 3, 9, 13), (2, 2, 1, 7), (2, 3, 2, 2)])
def test_winograd_matches_conv2d(C, N, H, W):
    rng = np.random.default_rng(H * 10 + W)
    x, w, gy = rng.standard_normal((C, H, W)), rng.standard_normal((N, C, 3, 3)), rng.standard_normal((N, H, W))
    xt, wt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True)
    yt = F.conv2d(xt[None], wt, padding=1)[0]
    yt.backward(torch.tensor(gy))
    y, V = wino_forward(x, w)
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-11)
    # data gradient = the same pipeline on dy with flipped, role-swapped taps
    wflip = np.ascontiguousarray(w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3))
    dx, _ = wino_forward(gy, wflip)
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(wino_wgrad(V, gy), wt.grad.numpy(), atol=1e-10)


def test_fp32_winograd_error_stays_near_the_direct_sum():
    rng = np.random.default_rng(0)
    C, H, W = 512, 8, 26
    x = rng.standard_normal((C, H, W)).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    ref = F.conv2d(torch.tensor(x, dtype=torch.float64)[None], torch.tensor(w, dtype=torch.float64), padding=1)[0].numpy()
    direct = F.conv2d(torch.tensor(x)[None], torch.tensor(w), padding=1)[0].numpy()
    y, _ = wino_forward(x, w, dtype=np.float32)
    scale = np.abs(ref).max()
    e_w = np.sqrt(((y - ref) ** 2).mean()) / scale
    e_d = np.sqrt(((direct - ref) ** 2).mean()) / scale
    assert e_w < 3e-7 and e_w < 4 * e_d, (e_w, e_d)
