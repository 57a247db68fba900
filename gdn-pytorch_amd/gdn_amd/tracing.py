"""roctx ranges around the phases of a training step (forward / losses / backward / all-reduce / Adam), so a
``rocprofv3 --marker-trace --kernel-trace`` summary splits by phase (tracing).

``librocprofiler-sdk-roctx.so`` (else ``libroctx64.so``) is loaded with ctypes on first use; without it (or with GDN_ROCTX=0) every call is a no-op.  Markers
cost ~100 ns each when no profiler is attached.
"""
import contextlib
import ctypes
import os

_lib = None
_tried = False


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("GDN_ROCTX", "1") == "0":
        return None
    # rocprofv3 (rocprofiler-sdk) intercepts its own roctx library; the legacy libroctx64 serves rocprof v1/v2
    for name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so",
                 "/opt/rocm/lib/libroctx64.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.restype = ctypes.c_int
            _lib = lib
            break
        except OSError:
            continue
    return _lib


def push(name):
    lib = _load()
    if lib is not None:
        lib.roctxRangePushA(name.encode())


def pop():
    lib = _load()
    if lib is not None:
        lib.roctxRangePop()


@contextlib.contextmanager
def span(name):
    """with tracing.span("gdn.backward"): ..."""
    push(name)
    try:
        yield
    finally:
        pop()
