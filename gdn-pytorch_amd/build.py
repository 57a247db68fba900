#!/usr/bin/env python3
"""Build libgdn_hip.so (gfx950) in-tree with hipcc.  No torch headers involved.

    python gdn-pytorch_amd/build.py [--force] [--verbose]
"""
import concurrent.futures as cf
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OUT = ROOT / "lib"
LIB = OUT / "libgdn_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I" + str(ROOT.parent / "include")]


def _newer(src, dst):
    return (not dst.exists()) or src.stat().st_mtime > dst.stat().st_mtime


def build(force=False, verbose=False):
    OUT.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    hdrs = list(CSRC.glob("*.h")) + [ROOT.parent / "include" / "gdn_hip.h"]
    hdr_m = max(h.stat().st_mtime for h in hdrs)
    jobs = []
    for s in srcs:
        o = OUT / (s.stem + ".o")
        if force or _newer(s, o) or o.stat().st_mtime < hdr_m:
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC, *FLAGS, "-c", str(s), "-o", str(o)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r
    with cf.ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for s, r in ex.map(cc, jobs):
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError("hipcc failed on %s" % s)
            if verbose and r.stderr:
                sys.stderr.write(r.stderr)
    objs = [OUT / (s.stem + ".o") for s in srcs]
    if force or jobs or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return LIB


def _asan_module():
    """The AddressSanitizer build recipe lives in build_asan.py, a CPU-side tool that is NOT shipped to the GPU boxes
    (.gpurunignore: sanitizer builds are not run there; the host planning code it checks needs no GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gdn_build_asan", ROOT / "build_asan.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(_asan_module().build_asan(verbose="--verbose" in sys.argv))
    else:
        lib = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
        print(lib)
