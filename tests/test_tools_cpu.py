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


def test_summarize_trace_drops_the_first_step_outlier(tmp_path):
    """tools/summarize_trace.py: per-(kernel, grid) means are over steady-state steps: a 20 ms first-step
    launch of a 0.2 ms kernel must not show up in its mean."""
    rows, t = [], 0
    for step in range(4):
        d = 20_000_000 if step == 0 else 200_000
        rows.append(("void conv_igemm_bf16(IgemmParams)", 0, t, t + d)); t += d
        rows.append(("void adam_kernel(float*)", 0, t, t + 1000)); t += 1000
    f = _trace(tmp_path, rows)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "summarize_trace.py"), str(f)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "steady-state steps only: 3 of 4 steps" in r.stdout
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("conv_igemm_bf16") or " conv_igemm_bf16" in ln][0]
    assert " 3 " in line and "200.0" in line, line


def test_bench_kernel_family_table():
    """bench.py's family table for step_kernel_breakdown: every symbol counts (not only the top rows), the two GEMM
    families are separate from the transforms that share their prefix."""
    sys.path.insert(0, str(ROOT))
    import bench
    f = bench.kernel_family
    assert f("cgemm_bins_kernel") == "cgemm" and f("cgemm_tn_bins_kernel") == "cgemm" and f("gemm_x3_nt8_kernel") == "gemm_x3"
    assert f("fft2d_fwd_kernel") == f("ifft_rows_overlap_kernel") == f("fft_wgrad_taps_kernel") == "fft_transforms"
    assert f("wino_gemm_kernel") == "wino_gemm_f32" and f("wino4_input_kernel") == "winograd_transforms"
    assert f("bn_bwd_apply_kernel") == "batchnorm" and f("conv_ring2_bf16") == "bf16_ring" and f("berhu_kernel") == "other"
