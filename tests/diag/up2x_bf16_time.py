"""GPU box: R's four decoder upsampling sites at B = 20 in bf16 -- BatchNorm-apply (+ residual) then upsample2x, against the one-pass
gdn_bn_apply_up2x; reflection-layer data gradient + fold then upsample2x_bwd, against gdn_conv_dgrad(dx_up2x).  usage: up2x_bf16_time.py"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
# (name, low H, low W, C of the block output, consumer Cout, k)
SITES = [("x6 -> upconv0", 8, 26, 512, 512, 3), ("x8 -> upconv1", 16, 52, 512, 256, 3), ("x10 -> upconv2", 32, 104, 256, 128, 5), ("x12 -> upconv3", 64, 208, 128, 64, 7)]


def timeit(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator(device=dev).manual_seed(0)
tot = [0.0] * 4
for name, H, W, C, co, k in SITES:
    y = torch.randn(B, H, W, C, device=dev, generator=g).bfloat16()
    res = torch.randn(B, H, W, C, device=dev, generator=g).bfloat16()
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    t_a = timeit(lambda: ops.bn_apply(y, sc, sh, False, res, out_dtype=torch.bfloat16))
    low = ops.bn_apply(y, sc, sh, False, res, out_dtype=torch.bfloat16)
    t_u = timeit(lambda: ops.upsample2x(low, False))
    t_f = timeit(lambda: ops.bn_apply_up2x(y, sc, sh, False, res, out_dtype=torch.bfloat16))
    op = ops.Conv(C, co, k, 1, k // 2, reflect=True)
    gy = torch.randn(B, 2 * H, 2 * W, co, device=dev, generator=g).bfloat16()
    wt = (torch.randn(k * k, C, co, device=dev, generator=g) * 0.02).bfloat16()
    t_d = timeit(lambda: op.dgrad(gy, wt, (2 * H, 2 * W)))
    dxh = op.dgrad(gy, wt, (2 * H, 2 * W))
    t_b = timeit(lambda: ops.upsample2x_bwd(dxh, False))
    t_df = timeit(lambda: op.dgrad(gy, wt, (2 * H, 2 * W), up2x=1))
    print("%-16s fwd: bn_apply %6.1f + upsample %6.1f = %6.1f us | one pass %6.1f us || bwd: dgrad+fold %7.1f + adjoint %6.1f = %7.1f us | fused fold %7.1f us"
          % (name, t_a, t_u, t_a + t_u, t_f, t_d, t_b, t_d + t_b, t_df), flush=True)
    for i, v in enumerate((t_a + t_u, t_f, t_d + t_b, t_df)):
        tot[i] += v
print("sum: fwd two-pass %.1f us, one pass %.1f us; bwd two-pass %.1f us, fused %.1f us" % tuple(tot))
