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


# ---- F(4x4,3x3) (round 4: the zero-padded 512-channel layers whose GEMMs run as bf16 x 3 products).  Interpolation points
# 0, +-a, +-b, inf with a = 5/8, b = 3/2 (csrc/conv_wino.hip: W4_A, W4_B; the textbook choice is a = 1, b = 2); the weight
# gradient is F(3x3,4x4) on the same points, so the saved B^T d B is shared.  With M(x) = x (x^2 - a^2)(x^2 - b^2):
#   B^T rows = coefficients of M(x) / (x - p) (row inf: M),  G rows = (1, p, p^2) / N_p,  N_p = prod_{q != p} (p - q),
#   A^T columns = (1, p, p^2, p^3) (column inf: e_3);  G_w rows = (1, p, p^2, p^3) / N_p,  A_w^T columns = (1, p, p^2) ----
def f4_matrices(a, b):
    a2, b2 = a * a, b * b
    n0, na, nb = a2 * b2, 2 * a2 * (a2 - b2), 2 * b2 * (b2 - a2)
    BT = np.array([[a2 * b2, 0, -(a2 + b2), 0, 1, 0],
                   [0, -a * b2, -b2, a, 1, 0], [0, a * b2, -b2, -a, 1, 0],
                   [0, -a2 * b, -a2, b, 1, 0], [0, a2 * b, -a2, -b, 1, 0],
                   [0, a2 * b2, 0, -(a2 + b2), 0, 1]], np.float64)
    pts, N = [0.0, a, -a, b, -b], [n0, na, na, nb, nb]
    G = np.array([[p ** j / n for j in range(3)] for p, n in zip(pts, N)] + [[0, 0, 1]], np.float64)
    GW = np.array([[p ** j / n for j in range(4)] for p, n in zip(pts, N)] + [[0, 0, 0, 1]], np.float64)
    AT = np.array([[p ** k for p in pts] + [1.0 if k == 3 else 0.0] for k in range(4)], np.float64)
    AWT = np.array([[p ** k for p in pts] + [1.0 if k == 2 else 0.0] for k in range(3)], np.float64)
    return BT, G, AT, GW, AWT


BT4, G4, AT4, GW4, AWT4 = f4_matrices(0.625, 1.5)


def test_f4_textbook_points_reproduce_lavin_gray():
    """The closed forms at a = 1, b = 2 are the published F(4x4,3x3) matrices."""
    BT, G, AT, _, _ = f4_matrices(1.0, 2.0)
    np.testing.assert_array_equal(BT, [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                                       [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]])
    np.testing.assert_allclose(G, [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                                   [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], rtol=1e-15)
    np.testing.assert_array_equal(AT, [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]])


def _tiles4(x, dtype):
    C, H, W = x.shape
    ty, tx = -(-H // 4), -(-W // 4)
    xp = np.zeros((C, 4 * ty + 2, 4 * tx + 2), dtype)
    xp[:, 1:H + 1, 1:W + 1] = x
    d = np.stack([xp[:, 4 * a:4 * a + 6, 4 * b:4 * b + 6] for a in range(ty) for b in range(tx)])     # [T,C,6,6]
    return d, ty, tx


def wino4_forward(x, w, dtype=np.float64):
    C, H, W = x.shape
    d, ty, tx = _tiles4(x.astype(dtype), dtype)
    V = np.einsum("ij,tcjk,lk->iltc", BT4.astype(dtype), d, BT4.astype(dtype))
    U = np.einsum("ij,ncjk,lk->ilnc", G4.astype(dtype), w.astype(dtype), G4.astype(dtype))
    M = np.einsum("iltc,ilnc->iltn", V, U).astype(dtype)
    y4 = np.einsum("ij,jktn,lk->tnil", AT4.astype(dtype), M, AT4.astype(dtype))                    # [T,N,4,4]
    y = y4.reshape(ty, tx, -1, 4, 4).transpose(2, 0, 3, 1, 4).reshape(-1, 4 * ty, 4 * tx)
    return y[:, :H, :W], V


def wino4_wgrad(V, gy, dtype=np.float64):
    N, H, W = gy.shape
    ty, tx = -(-H // 4), -(-W // 4)
    gp = np.zeros((N, 4 * ty, 4 * tx), dtype)
    gp[:, :H, :W] = gy
    g4 = gp.reshape(N, ty, 4, tx, 4).transpose(1, 3, 0, 2, 4).reshape(ty * tx, N, 4, 4)
    Dv = np.einsum("ij,tnjk,lk->iltn", GW4.astype(dtype), g4, GW4.astype(dtype))
    P = np.einsum("iltn,iltc->ilnc", Dv, V).astype(dtype)
    return np.einsum("ij,jknc,lk->ncil", AWT4.astype(dtype), P, AWT4.astype(dtype))


@pytest.mark.parametrize("C,N,H,W", [(3, 4, 8, 26), (4, 3, 9, 13), (2, 2, 4, 4), (2, 3, 7, 5)])
def test_winograd_f4_matches_conv2d(C, N, H, W):
    rng = np.random.default_rng(H * 10 + W)
    x, w, gy = rng.standard_normal((C, H, W)), rng.standard_normal((N, C, 3, 3)), rng.standard_normal((N, H, W))
    xt, wt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True)
    yt = F.conv2d(xt[None], wt, padding=1)[0]
    yt.backward(torch.tensor(gy))
    y, V = wino4_forward(x, w)
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-10)
    wflip = np.ascontiguousarray(w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3))
    dx, _ = wino4_forward(gy, wflip)
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(wino4_wgrad(V, gy), wt.grad.numpy(), atol=1e-9)


def test_winograd_f4_fp32_error_budget():
    """What F(4x4,3x3) costs in fp32 at 512 channels (transforms and products rounded to fp32, exact sums -- the bf16 x 3 GEMMs
    deliver fp32-exact products): rms error within 4x the direct fp32 sum's (measured 3.1x; the textbook points 0, +-1, +-2: 6.7x), the
    worst output below 1e-4 of the output rms; the weight gradient's F(3x3,4x4) stays below 1e-4 of the largest tap gradient."""
    rng = np.random.default_rng(0)
    C = N = 256
    H, W = 8, 12
    x = rng.standard_normal((C, H, W)); w = rng.standard_normal((N, C, 3, 3)) / np.sqrt(9 * C); gy = rng.standard_normal((N, H, W))
    y64, V64 = wino4_forward(x, w)
    y32, V32 = wino4_forward(x, w, np.float32)
    xp = np.zeros((C, H + 2, W + 2), np.float32); xp[:, 1:-1, 1:-1] = x
    yd = np.zeros((N, H, W), np.float32)
    for a in range(3):
        for b in range(3):
            yd += np.einsum("nc,chw->nhw", w[:, :, a, b].astype(np.float32), xp[:, a:a + H, b:b + W])
    rms = np.sqrt((y64 ** 2).mean())
    e4, ed = y32.astype(np.float64) - y64, yd.astype(np.float64) - y64
    assert np.sqrt((e4 ** 2).mean()) < 10 * np.sqrt((ed ** 2).mean()) and np.abs(e4).max() < 1e-4 * rms
    dw64 = wino4_wgrad(V64, gy)
    dw32 = wino4_wgrad(V32, gy, np.float32)
    assert np.abs(dw32 - dw64).max() < 1e-4 * np.abs(dw64).max()
    # the points: a = 5/8, b = 3/2 against the textbook a = 1, b = 2 in the same arithmetic
    global BT4, G4, AT4, GW4, AWT4
    keep = (BT4, G4, AT4, GW4, AWT4)
    try:
        BT4, G4, AT4, GW4, AWT4 = f4_matrices(1.0, 2.0)
        yt32, _ = wino4_forward(x, w, np.float32)
    finally:
        BT4, G4, AT4, GW4, AWT4 = keep
    et = yt32.astype(np.float64) - y64
    assert np.sqrt((e4 ** 2).mean()) < 0.7 * np.sqrt((et ** 2).mean()) and np.sqrt((e4 ** 2).mean()) < 4 * np.sqrt((ed ** 2).mean())
