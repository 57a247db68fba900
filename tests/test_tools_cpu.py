"""CPU tests of the trace tooling under tools/ (no GPU, no rocprofv3: synthetic kernel traces)."""
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
HDR = ('"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id",'
       '"Start_Timestamp","End_Timestamp"\n')


def _trace(tmp_path, rows):
    f = tmp_path / "step_kernel_trace.csv"
    f.write_text(HDR + "".join('"KERNEL_DISPATCH","Agent 2",1,%d,1,%d,1,"%s",%d,%d,%d\n' % (st, i, n, i, s, e)
                               for i, (n, st, s, e) in enumerate(rows)))
    return f


def _run(f):
    return subprocess.run([sys.executable, str(ROOT / "tools" / "check_no_mfma16_beside_fft.py"), str(f), "--label", "t"],
                          capture_output=True, text=True)


def test_checker_passes_a_serial_trace_and_a_two_stream_fft_backward(tmp_path):
    """gemm_x3 / bf16 convolutions strictly before or after frequency-domain kernels; two frequency-domain kernels overlapping
    EACH OTHER (the backward's two streams) is allowed."""
    rows = [("void gemm_x3_nt_kernel(float const*)", 0, 0, 100), ("void fft2d_fwd_kernel<40>(float const*)", 0, 100, 200),
            ("void cgemm_bins_kernel<true>(float const*)", 0, 200, 400), ("void cgemm_tn_bins_kernel(float const*)", 1, 210, 390),
            ("void conv_rowpatch_bf16_kernel<64,4,1>(void const*)", 0, 400, 500), ("void bn_apply_kernel(float*)", 0, 150, 450)]
    r = _run(_trace(tmp_path, rows))
    assert r.returncode == 0 and "0 overlapping pairs  OK" in r.stdout, r.stdout + r.stderr


def test_checker_fails_on_an_overlap(tmp_path):
    rows = [("void fft2d_fwd_kernel<40>(float const*)", 0, 100, 200), ("void gemm_x3_tn_kernel(float const*)", 1, 150, 260),
            ("void ifft_rows_overlap_kernel<32>(float*)", 0, 250, 300), ("void conv_wgrad_bf16_kernel<9,8,7>(void const*)", 1, 299, 310)]
    r = _run(_trace(tmp_path, rows))
    assert r.returncode == 1 and "3 overlapping pairs" in r.stdout and "OVERLAP" in r.stdout, r.stdout + r.stderr


def test_family_classifier_names_the_x3_gemms():
    sys.path.insert(0, str(ROOT / "tools"))
    from kernel_family import family, is_ours
    assert family("void gemm_x3_nt8_kernel<64>(float const*)").startswith("winograd+x3")
    assert family("wino_weights_x3_kernel").startswith("winograd+x3") and family("x3_pack_rows_kernel").startswith("winograd+x3")
    assert family("void wino_input_kernel(float const*)").startswith("winograd (")
    assert family("void cgemm_bins_kernel<false>").startswith("fft") and family("void conv_wgrad_f32_kernel<7,64>") == "direct wgrad"
    assert family("void bn_bwd_apply8_kernel").startswith("batchnorm") and family("adam_kernel") == "adam"
    assert is_ours("void upsample2x_fwd_kernel") and not is_ours("void at::native::vectorized_elementwise_kernel")
