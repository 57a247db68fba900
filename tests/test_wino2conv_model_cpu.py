"""CPU model of the Winograd F(3x3,2x2) scheme of csrc/conv_wino2.hip for the 4x4 stride-2 pad-1 layers (ConvBlock k4 s2
and ConvTBlock of AutoEncoder_DtoD, AE_model_unet.py:497-520): the polyphase decomposition, both tilings (form A: 3x3
tiles of the small image; form B: 6x6 blocks of the large image at offset -1 from one 4x4 patch of the small one), the
role swaps of the two backward passes and the weight gradient, written with the SAME index arithmetic as the kernels and
checked against torch's conv2d / conv_transpose2d autograd.  Pins the transform matrices (F(3,2): points 0, +-1, inf)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, -1, 0, 1]], np.float64)
G = np.array([[1, 0], [.5, .5], [.5, -.5], [0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, 0], [0, 1, 1, 1]], np.float64)


def test_f32_matrices_1d():
    rng = np.random.default_rng(0)
    d, g, dy = rng.standard_normal(4), rng.standard_normal(2), rng.standard_normal(3)
    np.testing.assert_allclose(AT @ ((G @ g) * (BT @ d)), [d[i] * g[0] + d[i + 1] * g[1] for i in range(3)], atol=1e-12)
    np.testing.assert_allclose(G.T @ ((BT @ d) * (AT.T @ dy)), [sum(dy[i] * d[i + u] for i in range(3)) for u in range(2)], atol=1e-12)


def _pad1(x, reflect, dtype):
    C, H, W = x.shape
    if reflect:
        return np.pad(x, ((0, 0), (1, 1), (1, 1)), mode="reflect").astype(dtype)
    xp = np.zeros((C, H + 2, W + 2), dtype)
    xp[:, 1:-1, 1:-1] = x
    return xp


def form_a_input(x, reflect, dtype=np.float64):
    """V[16][tile][(a*2+b)*C + c] for the LARGE image x [C,H,W]; tiles of 3x3 outputs = 8x8 footprints at (6t - 1)."""
    C, H, W = x.shape
    Hy, Wy = H // 2, W // 2
    ty, tx = -(-Hy // 3), -(-Wy // 3)
    xp = _pad1(x, reflect, dtype)                           # xp[r] = x[r - 1]
    big = np.zeros((C, 6 * ty + 2, 6 * tx + 2), dtype)       # rows beyond the padded image read as zero
    big[:, :min(H + 2, big.shape[1]), :min(W + 2, big.shape[2])] = xp[:, :big.shape[1], :big.shape[2]]
    V = np.zeros((4, 4, ty * tx, 4 * C), dtype)
    for t in range(ty * tx):
        a0, b0 = 6 * (t // tx), 6 * (t % tx)
        P = big[:, a0:a0 + 8, b0:b0 + 8]
        for a in range(2):
            for b in range(2):
                X = P[:, a::2, b::2]                          # X_ab[p][q] = P[2p + a][2q + b]
                V[:, :, t, (a * 2 + b) * C:(a * 2 + b + 1) * C] = np.einsum("ij,cjk,lk->ilc", BT.astype(dtype), X, BT.astype(dtype))
    return V, ty, tx


def form_a_weights(w, dtype=np.float64):
    """U[16][n][(a*2+b)*C + c] from w[n][c][4][4] (conv form)."""
    N, C = w.shape[:2]
    U = np.zeros((4, 4, N, 4 * C), dtype)
    for a in range(2):
        for b in range(2):
            g = w[:, :, a::2, b::2].astype(dtype)            # g_ab[u][v] = w[2u + a][2v + b]
            U[:, :, :, (a * 2 + b) * C:(a * 2 + b + 1) * C] = np.einsum("ij,ncjk,lk->ilnc", G.astype(dtype), g, G.astype(dtype))
    return U


def form_a_output(M, ty, tx, Hy, Wy):
    """M[16][tile][n] -> y[n][Hy][Wy]."""
    y9 = np.einsum("ij,jktn,lk->tnil", AT.astype(M.dtype), M, AT.astype(M.dtype))       # [T,N,3,3]
    y = y9.reshape(ty, tx, -1, 3, 3).transpose(2, 0, 3, 1, 4).reshape(-1, 3 * ty, 3 * tx)
    return y[:, :Hy, :Wy]


def strided_conv(x, w, reflect, dtype=np.float64):
    """y = conv2d(pad1(x), w, stride 2) through form A.  Returns y and V."""
    V, ty, tx = form_a_input(x, reflect, dtype)
    U = form_a_weights(w, dtype)
    M = np.einsum("iltk,ilnk->iltn", V, U).astype(dtype)
    return form_a_output(M, ty, tx, x.shape[1] // 2, x.shape[2] // 2), V


def transposed_conv(d, wt, Hout, Wout, off, dtype=np.float64):
    """z[r][s] = sum_{i,j} d[i][j] * wt[:, :, r + 1 - 2i, s + 1 - 2j] through form B; wt[cd][n][4][4].
    off = 0: rows r in [0, Hout);  off = 1: the padded domain, out row = r + 1 for r in [-1, Hout - 2]."""
    Cd, Hy, Wy = d.shape
    N = wt.shape[1]
    ty, tx = -(-(Hout + 1 - off) // 6), -(-(Wout + 1 - off) // 6)   # = cdiv(Hx + ext, 6): Hx = Hout - 2*off, ext = 1 + off
    dp = np.zeros((Cd, 3 * ty + 3, 3 * tx + 3), dtype)               # dp[s] = d[s - 1]
    dp[:, 1:Hy + 1, 1:Wy + 1] = d
    out = np.zeros((N, Hout, Wout), dtype)
    for t in range(ty * tx):
        a0, b0 = 3 * (t // tx), 3 * (t % tx)
        Vd = np.einsum("ij,cjk,lk->ilc", BT.astype(dtype), dp[:, a0:a0 + 4, b0:b0 + 4], BT.astype(dtype))
        for a in range(2):
            for b in range(2):
                g = np.stack([[wt[:, :, 3 - a - 2 * u, 3 - b - 2 * v] for v in range(2)] for u in range(2)])   # [2,2,cd,n]
                Ug = np.einsum("ij,jkcn,lk->ilcn", G.astype(dtype), g.astype(dtype), G.astype(dtype))
                m = np.einsum("ilc,ilcn->iln", Vd, Ug)
                o = np.einsum("ij,jkn,lk->nil", AT.astype(dtype), m, AT.astype(dtype))                       # [n,3,3]
                for i in range(3):
                    for j in range(3):
                        oy, ox = 6 * (t // tx) + 2 * i - a + off, 6 * (t % tx) + 2 * j - b + off
                        if 0 <= oy < Hout and 0 <= ox < Wout:
                            out[:, oy, ox] = o[:, i, j]
    return out


def lift(sm, ty, tx):
    """Dv[16][tile][n] = A s A^T of the 3x3 tiles of the small image."""
    N, Hy, Wy = sm.shape
    gp = np.zeros((N, 3 * ty, 3 * tx))
    gp[:, :Hy, :Wy] = sm
    g9 = gp.reshape(N, ty, 3, tx, 3).transpose(1, 3, 0, 2, 4).reshape(ty * tx, N, 3, 3)
    return np.einsum("ji,tnjk,kl->iltn", AT, g9, AT)


def wgrad(V, sm, C):
    """dw[n][c][4][4] from V (form A of the large image) and the small image sm [N,Hy,Wy]."""
    Hy, Wy = sm.shape[1:]
    ty, tx = -(-Hy // 3), -(-Wy // 3)
    P = np.einsum("iltn,iltk->ilnk", lift(sm, ty, tx), V)
    dw = np.zeros((sm.shape[0], C, 4, 4))
    for a in range(2):
        for b in range(2):
            dw[:, :, a::2, b::2] = np.einsum("ji,jknc,kl->ncil", G, P[:, :, :, (a * 2 + b) * C:(a * 2 + b + 1) * C], G)
    return dw


def fold_reflect1(dxp):
    """padded-domain gradient [C,H+2,W+2] -> [C,H,W] (rows -1 / H mirror rows 1 / H-2)."""
    g = dxp[:, 1:-1, 1:-1].copy()
    g[:, 1, :] += dxp[:, 0, 1:-1]; g[:, -2, :] += dxp[:, -1, 1:-1]
    g[:, :, 1] += dxp[:, 1:-1, 0]; g[:, :, -2] += dxp[:, 1:-1, -1]
    g[:, 1, 1] += dxp[:, 0, 0]; g[:, 1, -2] += dxp[:, 0, -1]; g[:, -2, 1] += dxp[:, -1, 0]; g[:, -2, -2] += dxp[:, -1, -1]
    return g


@pytest.mark.parametrize("C,N,H,W,reflect", [(3, 2, 12, 18, True), (2, 3, 8, 26, False), (2, 2, 4, 6, True), (1, 2, 16, 10, True),
                                             (2, 1, 14, 4, False)])
def test_strided_conv_all_three_passes(C, N, H, W, reflect):
    rng = np.random.default_rng(H * 100 + W)
    x, w = rng.standard_normal((C, H, W)), rng.standard_normal((N, C, 4, 4))
    gy = rng.standard_normal((N, H // 2, W // 2))
    xt, wt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True)
    xp = F.pad(xt[None], (1, 1, 1, 1), mode="reflect") if reflect else F.pad(xt[None], (1, 1, 1, 1))
    yt = F.conv2d(xp, wt, stride=2)[0]
    yt.backward(torch.tensor(gy))
    y, V = strided_conv(x, w, reflect)
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-11)
    np.testing.assert_allclose(wgrad(V, gy, C), wt.grad.numpy(), atol=1e-10)
    # data gradient: form B on dy with the role-swapped weights (reduction over N); a reflection layer goes through the
    # padded domain and the border fold, a zero-padded one crops
    wsw = w                                                  # wt[cd = n][nout = c]
    if reflect:
        dxp = transposed_conv(gy, wsw, H + 2, W + 2, 1)
        dx = fold_reflect1(dxp)
    else:
        dx = transposed_conv(gy, wsw, H, W, 0)
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-10)


@pytest.mark.parametrize("Ci,Co,H,W", [(2, 3, 4, 13), (3, 2, 8, 3), (1, 1, 1, 2), (2, 2, 5, 5)])
def test_conv_transpose_all_three_passes(Ci, Co, H, W):
    rng = np.random.default_rng(H * 100 + W)
    d, w = rng.standard_normal((Ci, H, W)), rng.standard_normal((Ci, Co, 4, 4))        # torch ConvTranspose2d weight [Cin,Cout,4,4]
    gz = rng.standard_normal((Co, 2 * H, 2 * W))
    dt, wt = torch.tensor(d, requires_grad=True), torch.tensor(w, requires_grad=True)
    zt = F.conv_transpose2d(dt[None], wt, stride=2, padding=1)[0]
    zt.backward(torch.tensor(gz))
    z = transposed_conv(d, w, 2 * H, 2 * W, 0)
    np.testing.assert_allclose(z, zt.detach().numpy(), atol=1e-11)
    # data gradient = a zero-padded strided conv of dz with the role-swapped weights (conv-form n = Cin_T, c = Cout_T);
    # the weight gradient reduces the SAME transform of dz against the lifted tiles of d
    dd, V = strided_conv(gz, w, False)
    np.testing.assert_allclose(dd, dt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(wgrad(V, d, Co), wt.grad.numpy(), atol=1e-10)


def test_fp32_error_stays_near_the_direct_sum():
    rng = np.random.default_rng(0)
    C, H, W = 256, 12, 24
    x = rng.standard_normal((C, H, W)).astype(np.float32)
    w = (rng.standard_normal((C, C, 4, 4)) / np.sqrt(C * 16)).astype(np.float32)
    pad = lambda t: F.pad(t[None], (1, 1, 1, 1), mode="reflect")
    ref = F.conv2d(pad(torch.tensor(x, dtype=torch.float64)), torch.tensor(w, dtype=torch.float64), stride=2)[0].numpy()
    direct = F.conv2d(pad(torch.tensor(x)), torch.tensor(w), stride=2)[0].numpy()
    y, _ = strided_conv(x, w, True, dtype=np.float32)
    scale = np.abs(ref).max()
    e_w = np.sqrt(((y - ref) ** 2).mean()) / scale
    e_d = np.sqrt(((direct - ref) ** 2).mean()) / scale
    print("rms error / max: winograd F(3x3,2x2) %.3e, direct fp32 %.3e" % (e_w, e_d))
    assert e_w < 3e-7 and e_w < 4 * e_d, (e_w, e_d)
