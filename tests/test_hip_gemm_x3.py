"""fp32 per-bin GEMMs on the bf16 matrix pipe (csrc/gemm_x3.hip: bf16 x 3 split products) against fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bins,M,N,K", [(1, 128, 128, 32), (3, 200, 256, 96), (16, 1040, 512, 512), (2, 77, 128, 2048)])
def test_gemm_x3_nt_matches_fp64(gpu, bins, M, N, K):
    """C = A @ B^T per bin.  The six bf16 products reproduce every fp32 product to ~2^-23, so the result must sit at the
    fp32-rounding distance from the fp64 product -- not at bf16's: the bar is 4e-6 of sum |a||b| (an fp32 dot product of K
    terms is allowed ~K^0.5 * 6e-8), and within 4x of torch's (blocked, pairwise-summed) fp32 CPU matmul of the same operands."""
    from gdn_amd import ops
    g = torch.Generator().manual_seed(bins * 1000 + K)
    A = torch.randn(bins, M, K, generator=g)
    B = torch.randn(bins, N, K, generator=g) * torch.logspace(-2, 2, N).view(1, N, 1)      # rows of very different scale
    ref = torch.bmm(A.double(), B.double().transpose(1, 2))
    scale = torch.bmm(A.double().abs(), B.double().abs().transpose(1, 2))
    f32 = torch.bmm(A, B.transpose(1, 2)).double()
    Bp = ops.gemm_x3_pack(B.to(gpu))
    C = ops.gemm_x3_nt(A.to(gpu), Bp, N).cpu().double()
    err = float(((C - ref).abs() / scale).max())
    err32 = float(((f32 - ref).abs() / scale).max())
    print("gemm_x3 %dx%dx%dx%d: max err / sum|a||b| = %.3e (fp32 matmul %.3e)" % (bins, M, N, K, err, err32))
    assert err < 4e-6 and err < max(4 * err32, 4e-7)


def test_gemm_x3_exact_on_integers(gpu):
    """Small-integer operands: every product and every partial sum is exact in fp32, so C must equal the integer result
    bit for bit -- catches any fragment / swizzle / plane mix-up that a tolerance would blur (asymmetric B)."""
    from gdn_amd import ops
    g = torch.Generator().manual_seed(5)
    bins, M, N, K = 2, 160, 256, 64
    A = torch.randint(-8, 9, (bins, M, K), generator=g).float()
    B = torch.randint(-8, 9, (bins, N, K), generator=g).float() + torch.arange(N).view(1, N, 1).float()      # row-dependent
    ref = torch.bmm(A.double(), B.double().transpose(1, 2))
    C = ops.gemm_x3_nt(A.to(gpu), ops.gemm_x3_pack(B.to(gpu)), N).cpu().double()
    assert torch.equal(C, ref)


def test_gemm_x3_rejects_unsupported(gpu):
    from gdn_amd import GdnError, ops
    A = torch.zeros(1, 64, 64, device=gpu)
    Bp = ops.gemm_x3_pack(torch.zeros(1, 128, 64, device=gpu))
    with pytest.raises(GdnError):
        ops.gemm_x3_nt(A, Bp, 96)                      # N not a multiple of 128


@pytest.mark.parametrize("bins,T,NI,NJ,ns", [(1, 32, 128, 128, 1), (2, 1000, 256, 128, 3), (16, 1040, 512, 512, 2), (3, 77, 128, 384, 1)])
def test_gemm_x3_tn_matches_fp64(gpu, bins, T, NI, NJ, ns):
    """The weight gradients' reduction GEMM P = A^T B over rows (both operands split on the fly, transposed in registers),
    with the reduction cut into `ns` partial sets: their sum against fp64, same bar as the NT kernel."""
    from gdn_amd import ops
    g = torch.Generator().manual_seed(bins * 100 + T)
    A = torch.randn(bins, T, NI, generator=g) * torch.logspace(-1, 1, NI).view(1, 1, NI)
    B = torch.randn(bins, T, NJ, generator=g)
    ref = torch.bmm(A.double().transpose(1, 2), B.double())
    scale = torch.bmm(A.double().abs().transpose(1, 2), B.double().abs())
    P = ops.gemm_x3_tn(A.to(gpu), B.to(gpu), ns).cpu().double()
    assert P.shape == (ns, bins, NI, NJ)
    err = float(((P.sum(0) - ref).abs() / scale).max())
    print("gemm_x3_tn %dx%dx%dx%d/%d: max err / sum|a||b| = %.3e" % (bins, T, NI, NJ, ns, err))
    assert err < 4e-6


def test_gemm_x3_tn_exact_on_integers(gpu):
    from gdn_amd import ops
    g = torch.Generator().manual_seed(6)
    bins, T, NI, NJ = 2, 100, 128, 256
    A = torch.randint(-8, 9, (bins, T, NI), generator=g).float() + torch.arange(NI).view(1, 1, NI).float()
    B = torch.randint(-8, 9, (bins, T, NJ), generator=g).float()
    ref = torch.bmm(A.double().transpose(1, 2), B.double())
    P = ops.gemm_x3_tn(A.to(gpu), B.to(gpu), 2).cpu().double().sum(0)
    assert torch.equal(P, ref)


@pytest.mark.parametrize("bins,M,N,K,cus", [(1, 128, 128, 32, 0), (3, 300, 384, 96, 0), (16, 1040, 512, 512, 0), (9, 700, 256, 128, 16),
                                            (5, 129, 128, 64, 8), (2, 1000, 640, 256, 32)])
def test_gemm_x3_ring_matches_nt_kernel(gpu, bins, M, N, K, cus, monkeypatch):
    """The LDS-DMA ring kernel (csrc/gemm_x3_ring.h: both operands packed, persistent workgroups) against gemm_x3_nt: the same
    products in the same order per accumulator, so whole units are BIT-identical; units of the last round that are cut along K
    (fp32 partial slabs summed in order) differ by fp32 rounding of the partial sums only -- the fp64 bar of the test above.
    cus: plan for a chip of that many CUs (GDN_RING_CUS) so small shapes reach several rounds, odd panel counts (128 x 256
    units, a duplicated last B panel) and the K-split tail; rows past M and the padding rows of the last panel are never stored."""
    from gdn_amd import ops
    if cus:
        monkeypatch.setenv("GDN_RING_CUS", str(cus))
    g = torch.Generator().manual_seed(bins * 77 + M)
    A = torch.randn(bins, M, K, generator=g)
    B = torch.randn(bins, N, K, generator=g) * torch.logspace(-1, 1, N).view(1, N, 1)
    Ap, Bp = ops.gemm_x3_pack(A.to(gpu)), ops.gemm_x3_pack(B.to(gpu))
    C0 = ops.gemm_x3_nt(A.to(gpu), Bp, N)
    guard = torch.full((bins * M * N + 4096,), 12345.0, device=gpu)
    C1 = guard[:bins * M * N].view(bins, M, N)
    ops.gemm_x3_nt_packed(Ap, Bp, bins, M, N, K, out=C1)
    torch.cuda.synchronize()
    assert bool((guard[bins * M * N:] == 12345.0).all()), "stored past the result"
    ref = torch.bmm(A.double(), B.double().transpose(1, 2))
    scale = torch.bmm(A.double().abs(), B.double().abs().transpose(1, 2))
    err = float(((C1.cpu().double() - ref).abs() / scale).max())
    same = float((C0 == C1).float().mean())
    print("ring %dx%dx%dx%d cus %d: identical to nt on %.1f %% of the outputs, max err / sum|a||b| = %.3e" % (bins, M, N, K, cus, 100 * same, err))
    assert err < 4e-6
    monkeypatch.setenv("GDN_X3_RING_TAIL", "0")                    # no K split: every unit whole -> bit-identical
    C2 = torch.empty_like(C0)
    ops.gemm_x3_nt_packed(Ap, Bp, bins, M, N, K, out=C2)
    assert torch.equal(C0, C2)


def test_gemm_x3_ring_exact_on_integers(gpu, monkeypatch):
    """Integer operands through the ring kernel with a K-split tail: partial sums are exact, so the result is the integer product."""
    from gdn_amd import ops
    monkeypatch.setenv("GDN_RING_CUS", "16")
    g = torch.Generator().manual_seed(5)
    bins, M, N, K = 3, 520, 256, 128
    A = torch.randint(-8, 9, (bins, M, K), generator=g).float()
    B = torch.randint(-8, 9, (bins, N, K), generator=g).float()
    C = ops.gemm_x3_nt_packed(ops.gemm_x3_pack(A.to(gpu)), ops.gemm_x3_pack(B.to(gpu)), bins, M, N, K).cpu()
    assert torch.equal(C, torch.bmm(A, B.transpose(1, 2)))
