"""ctypes binding of the C-ABI library (include/dfold_hip.h).  The library is loaded lazily on first
device op; a missing library is a hard error -- the device path has no CPU / eager fallback."""
import ctypes
import os
import re
from ctypes import POINTER, Structure, byref, c_float, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFOLD_LIB: diagnostic builds of the same ABI (kernel experiments); the product loads the in-tree library
LIB_PATH = os.environ.get("DFOLD_LIB") or os.path.join(_HERE, "csrc", "libdfold_hip.so")
HEADER = os.path.join(_HERE, "..", "include", "dfold_hip.h")

_lib = None


class RowMap(Structure):
    _fields_ = [("base", c_int64), ("ld", c_int64), ("mode", c_int32), ("n", c_int32), ("f", c_int32),
                ("fp", c_int32), ("wp", c_int32)]


class GemmDesc(Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("C2", c_void_p), ("bias", c_void_p),
                ("R", c_void_p), ("R2", c_void_p), ("zeros", c_void_p), ("a_seg0", c_int64), ("a_seg_s0", c_int64),
                ("a_seg_s1", c_int64), ("a_seg_s2", c_int64), ("b_seg0", c_int64), ("b_seg_s0", c_int64),
                ("b_seg_s1", c_int64), ("b_seg_s2", c_int64), ("seg_div", c_int32), ("seg_div_mid", c_int32),
                ("a_rows", RowMap), ("c_rows", RowMap), ("ldb", c_int64),
                ("sa0", c_int64), ("sa1", c_int64), ("sb0", c_int64), ("sb1", c_int64), ("sc0", c_int64),
                ("sc1", c_int64), ("M", c_int32), ("N", c_int32), ("nseg", c_int32), ("seglen", c_int32),
                ("nbatch", c_int32), ("nb1", c_int32), ("flags", c_int32), ("alpha", c_float),
                ("splitk", c_int32), ("conv_frames", c_int32), ("splitk_ws", c_void_p), ("splitk_cnt", c_void_p),
                ("nz_ps", c_void_p), ("nz_radius", c_int32), ("nz_f0", c_int32)]


GEMM_BIAS, GEMM_RELU, GEMM_RESID, GEMM_RELUMASK, GEMM_OUT_BF16, GEMM_ACCUM, GEMM_ATOMIC = 1, 2, 4, 8, 16, 32, 64


def header_symbols():
    """Every entry point declared in include/dfold_hip.h."""
    with open(HEADER) as fh:
        txt = fh.read()
    return sorted(set(re.findall(r"^\s*int\s+(dfold_\w+)\s*\(", txt, flags=re.M)))


def header_abi_version():
    with open(HEADER) as fh:
        return int(re.search(r"^#define\s+DFOLD_ABI_VERSION\s+(\d+)", fh.read(), flags=re.M).group(1))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m dynamicpdb_amd.build_ext` "
                "(the MI355X device path has no fallback)")
        # torch bundles its own libamdhip64.so.7; it must be resident BEFORE our library is dlopen'ed so that both bind
        # to the same HIP runtime instance (loading ours first pulls /opt/rocm's copy in and torch's device pointers
        # then belong to a different runtime: every launch fails).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name in header_symbols():
            fn = getattr(L, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype = c_int32
        want = header_abi_version()
        if L.dfold_abi_version() != want:
            raise RuntimeError(f"{LIB_PATH}: ABI version {L.dfold_abi_version()}, include/dfold_hip.h declares {want} "
                               "(stale or variant library: rebuild with `python -m dynamicpdb_amd.build_ext --force`)")
        _lib = L
    return _lib


def check(rc, what):
    if rc == 0:
        return
    if rc == -1:
        raise ValueError(f"{what}: invalid argument (DFOLD_EINVAL)")
    raise RuntimeError(f"{what}: kernel launch failed (rc={rc})")


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
