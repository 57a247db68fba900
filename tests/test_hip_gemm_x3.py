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
