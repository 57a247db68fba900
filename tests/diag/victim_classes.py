"""GPU box: which KIND of kernel does a concurrent barrier-paced bf16 matrix loop disturb?  tests/diag/victims.hip on one
stream (each variant repeated and compared with its first result), tests/diag/mfma_neighbour.hip variant 5 on a second."""
import ctypes, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
import torch
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mf = ctypes.CDLL(str(ROOT / "tests/diag/_build/libmfma_neighbour.so"))
mf.mfma_loop.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
vc = ctypes.CDLL(str(ROOT / "tests/diag/_build/libvictims.so"))
vc.victim.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
g = torch.Generator(device=dev).manual_seed(5)
inp = torch.rand(1 << 20, device=dev, generator=g) + 0.5
side = torch.cuda.Stream()
sink = torch.zeros(4, device=dev)
names = ["scalar fp32 FMA chain", "packed fp32 FMA chain", "LDS transpose + barriers", "192-register working set (~200 VGPRs)",
         "scratch array, dynamic index", "global gather + multiply", "LDS transpose + barriers + packed math",
         "torch.fft.rfft2 (rocFFT)", "table look-ups, wave-uniform (scalar cache)", "table look-ups, per-lane (L1 hits)", "register butterflies + twiddles (v_pk_add/mul)"]
img = torch.randn(8, 64, 64, 64, device=dev, generator=g)
main = torch.cuda.current_stream()
for neighbour in (False, True):
    for v in range(11):
        out = torch.empty(1 << 20, device=dev)
        ref, bad = None, 0
        for it in range(n):
            if neighbour:
                with torch.cuda.stream(side):
                    for _ in range(4):
                        mf.mfma_loop(5, 1024, 200, sink.data_ptr(), side.cuda_stream)
            if v != 7:
                vc.victim(v, inp.data_ptr(), out.data_ptr(), 4096, 400, main.cuda_stream)
                res = out
            else:
                res = torch.view_as_real(torch.fft.rfft2(img))
            torch.cuda.synchronize()
            if ref is None:
                ref = res.clone()
            elif not torch.equal(res, ref):
                bad += 1
        print("%-28s %-42s %3d of %d repeats differ" % ("bf16 loop on a second stream:" if neighbour else "alone:", names[v], bad, n), flush=True)
