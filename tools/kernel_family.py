"""One classifier of this library's kernel names for every trace / counter summary under tools/ (prof_step.sh, pmc_step.sh,
check_no_mfma16_beside_fft.py), so the family tables cannot drift apart again (`gemm_x3*` landed in "other")."""
import re

# kernels whose inner loop alternates bursts of 16-bit matrix instructions with workgroup barriers: the
# bf16 x 3 GEMMs of the Winograd paths and every bf16 convolution kernel
# (conv_head_mfma_kernel: the 64 -> 1 heads on bf16 MFMAs, in every fp32 step too -- VERDICT r4; wgrad_ring_bf16: round 5)
MFMA16 = re.compile(r"gemm_x3|conv_\w*bf16|cgemm_x3|_bf16_kernel|conv_head_mfma|wgrad_ring_bf16")
# the frequency-domain chain (transforms, complex per-bin GEMMs, overlap-add inverses, tap transform, reflection folds)
FFT = re.compile(r"fft|cgemm")


def family(n):
    if re.search(r"gemm_x3|x3_pack|x3r_combine|wino4?_weights_x3", n):
        return "winograd+x3 (bf16 x 3 GEMMs)"
    if FFT.search(n):
        return "fft chain"
    if "wino" in n or re.search(r"\bw2_", n):
        return "winograd (transforms, fp32 GEMMs)"
    if "wgrad" in n:
        return "direct wgrad"
    if re.search(r"conv_igemm|conv_rowpatch|conv_ring|conv_head|conv_c1|splitk", n):
        return "direct conv (igemm/rowpatch/ring/c1/head/splitk)"
    if re.search(r"bn_", n):
        return "batchnorm"
    if re.search(r"upsample|reflect_fold|add_|cast_|transpose|scale_dev|tanh_bwd|fill_|nchw|nhwc|aug_", n):
        return "pointwise (upsample/fold/add/cast/layout)"
    if re.search(r"berhu|sobel|smooth|sqdiff|finalize_sum|absdiff|zero_u32|depth_metrics", n):
        return "losses/metrics"
    if "adam" in n:
        return "adam"
    if "at::native" in n or n.startswith("void at::"):
        return "torch (at::native)"
    if "rocclr" in n or "copyBuffer" in n or "fillBuffer" in n:
        return "runtime copies/fills (set-up, outside the steps)"
    if re.search(r"nccl|rccl|AllReduce|Broadcast", n):
        return "rccl"
    return "other gdn"


OURS = ("winograd+x3", "fft", "winograd", "direct", "batchnorm", "pointwise", "losses/metrics", "adam", "other")


def is_ours(n):
    return family(n).split()[0] in OURS
