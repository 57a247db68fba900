#!/usr/bin/env python3
"""CPU-side tool: AddressSanitizer build of the HOST side of libgdn_hip (python gdn-pytorch_amd/build.py --asan).

Kept out of build.py and listed in .gpurunignore: sanitizer builds are neither needed nor allowed on the GPU boxes, and the
code this build checks (geometry / plan / workspace-size queries, argument checks) runs without a GPU
(tests/test_abi_cpu.py::test_host_planning_code_under_asan)."""
import concurrent.futures as cf
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OUT = ROOT / "lib"
HIPCC = __import__("os").environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def build_asan(verbose=False):
    """AddressSanitizer build of the HOST side of the library (geometry / plan / workspace-size code, argument checks):
    host objects instrumented (-Xarch_host -fsanitize=address), device code compiled as usual at -O1.  GPU ASan is not
    available on this pool, and the kernels are covered by the parity tests; what a sanitizer can find here is the host
    planning code, which runs without a GPU.  -> lib/asan/libgdn_hip_asan.so (use with LD_PRELOAD=<libclang_rt.asan>)."""
    out = OUT / "asan"
    out.mkdir(parents=True, exist_ok=True)
    lib = out / "libgdn_hip_asan.so"
    srcs = sorted(CSRC.glob("*.hip"))
    hdrs = list(CSRC.glob("*.h")) + [ROOT.parent / "include" / "gdn_hip.h"]
    newest = max(f.stat().st_mtime for f in srcs + hdrs)
    if lib.exists() and lib.stat().st_mtime > newest:
        return lib
    flags = ["-O1", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-Xarch_host", "-fsanitize=address", "-Xarch_host",
             "-fno-omit-frame-pointer", "-Wno-pass-failed", "-I" + str(ROOT.parent / "include")]

    def cc(s):
        r = subprocess.run([HIPCC, *flags, "-c", str(s), "-o", str(out / (s.stem + ".o"))], capture_output=True, text=True)
        return s, r
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        for s, r in ex.map(cc, srcs):
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError("hipcc (asan) failed on %s" % s)
            if verbose and r.stderr:
                sys.stderr.write(r.stderr)
    r = subprocess.run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", "-o", str(lib),
                        *[str(out / (s.stem + ".o")) for s in srcs]], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("asan link failed")
    return lib


def asan_runtime():
    """Path of clang's shared ASan runtime (to LD_PRELOAD into the python that loads the instrumented library)."""
    clang = pathlib.Path(HIPCC).resolve().parent.parent / "lib" / "llvm" / "bin" / "clang"
    if not clang.exists():
        clang = pathlib.Path("/opt/rocm/lib/llvm/bin/clang")
    r = subprocess.run([str(clang), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    return r.stdout.strip()
