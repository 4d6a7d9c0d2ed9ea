"""ctypes binding of libgpslc_hip.so (include/gpslc_hip.h).

This is the Python stand-in for the Julia ``ccall`` shim of INTEGRATION.md: it binds exactly the
symbols a Julia maintainer would bind, with the same argument conventions (column-major float64,
host pointers for the plain entry points, device pointers for the ``_dev`` ones).

There is NO CPU fallback: if the shared library is missing or cannot be loaded, importing the
package's compute entry points raises ``GPSLCLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgpslc_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gpslc_hip.h")


class GPSLCLibraryError(RuntimeError):
    pass


class GPSLCError(RuntimeError):
    """A negative status from the library (bad argument, HIP failure, ...)."""

    def __init__(self, status, message):
        super().__init__(f"gpslc status {status}: {message}")
        self.status = status


class PosDefException(ArithmeticError):
    """Mirrors Julia's LinearAlgebra.PosDefException(info) raised by PDMats inside Gen.mvnormal:
    ``info`` is the 1-based pivot at which the Cholesky factorisation broke down (values > n refer
    to the CovITE factorisation, pivot = info - n)."""

    def __init__(self, info):
        super().__init__(f"matrix is not positive definite; Cholesky factorization failed (info = {info})")
        self.info = info


c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)

# name -> (restype, argtypes); must cover every function include/gpslc_hip.h declares
_D = C.c_void_p  # double* passed as raw address (host numpy pointer or device pointer)
SIGNATURES = {
    "gpslc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int32, C.c_int32, C.c_uint32]),
    "gpslc_destroy": (C.c_int, [C.c_void_p]),
    "gpslc_set_data": (C.c_int, [C.c_void_p, _D, _D, _D]),
    "gpslc_set_data_dev": (C.c_int, [C.c_void_p, _D, _D, _D]),
    "gpslc_set_tuning": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "gpslc_set_task_schedule": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "gpslc_set_ensemble": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    "gpslc_last_error": (C.c_char_p, [C.c_void_p]),
    "gpslc_rbf_log": (C.c_int, [C.c_void_p, _D, _D, C.c_int64, C.c_int32, _D, C.c_int32, _D]),
    "gpslc_rbf_log_dev": (C.c_int, [C.c_void_p, _D, _D, C.c_int64, C.c_int32, _D, C.c_int32, _D]),
    "gpslc_process_cov": (C.c_int, [C.c_void_p, _D, C.c_int64, C.c_double, C.c_double, _D]),
    "gpslc_process_cov_dev": (C.c_int, [C.c_void_p, _D, C.c_int64, C.c_double, C.c_double, _D]),
    "gpslc_y_logpdf": (C.c_int, [C.c_void_p, C.c_int64, _D, _D, _D, _D, _D, _D, _D, _D, _D]),
    "gpslc_gp_logpdf": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, _D, C.c_int32, _D, _D, _D, _D, C.c_int32, _D]),
    "gpslc_nodes_logpdf": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, _D]),
    "gpslc_nodes_draw": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, _D, _D]),
    "gpslc_mvn_logpdf": (C.c_int, [C.c_void_p, C.c_int64, _D, _D, _D, _D]),
    "gpslc_mvn_draw": (C.c_int, [C.c_void_p, C.c_int64, _D, _D, _D, _D]),
    "gpslc_predict": (C.c_int, [C.c_void_p, C.c_int64, _D, _D, _D, _D, _D, _D, C.c_int32, _D, C.c_double,
                                C.c_int32, C.c_uint64, _D, _D, _D, _D, _D]),
    "gpslc_predict_dev": (C.c_int, [C.c_void_p, C.c_int64, _D, _D, _D, _D, _D, _D, C.c_int32, _D, C.c_double,
                                    C.c_int32, C.c_uint64, _D, _D, _D, _D, _D]),
    "gpslc_predict_multi": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.c_int64, _D, _D, _D, _D, _D, _D, C.c_int32, _D,
                                      C.c_double, C.c_int32, C.c_uint64, _D, _D, _D, _D, _D, c_int32_p]),
    "gpslc_shard_range": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, c_int64_p, c_int64_p]),
    "gpslc_ite_distributions": (C.c_int, [C.c_void_p, C.c_int64, _D, _D, _D, _D, _D, _D, C.c_double,
                                          C.c_double, _D, _D]),
    "gpslc_likelihood_distribution": (C.c_int, [C.c_void_p, _D, _D, _D, C.c_double, C.c_double, C.c_double,
                                                C.c_double, _D, _D, _D, _D, _D, _D, _D]),
    "gpslc_summarize": (C.c_int, [C.c_void_p, _D, C.c_int64, C.c_int64, C.c_double, _D, _D, _D]),
    "gpslc_summarize_dev": (C.c_int, [C.c_void_p, _D, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double, _D, _D, _D]),
    "gpslc_sate_samples": (C.c_int, [_D, _D, C.c_int64, C.c_int32, C.c_uint64, _D, _D]),
    "gpslc_last_info": (C.c_int, [C.c_void_p, c_int32_p, C.c_int64]),
    "gpslc_profile_reset": (C.c_int, [C.c_void_p]),
    "gpslc_profile_get": (C.c_int, [C.c_void_p, c_int64_p, c_double_p, c_double_p]),
    "gpslc_profile_get_class": (C.c_int, [C.c_void_p, C.c_int32, c_int64_p, c_double_p, c_double_p]),
    "gpslc_version": (C.c_char_p, []),
    "gpslc_pack_save": (C.c_int, [C.c_char_p, C.c_void_p, _D, _D, _D, _D, _D, _D, _D, _D, _D]),
    "gpslc_pack_read_header": (C.c_int, [C.c_char_p, C.c_void_p]),
    "gpslc_pack_load": (C.c_int, [C.c_char_p, C.c_int64, C.c_int64, _D, _D, _D, _D, _D, _D, _D, _D, _D]),
}


class Node(C.Structure):
    """gpslc_node (include/gpslc_hip.h)."""
    _fields_ = [("nF", C.c_int32), ("reserved", C.c_int32), ("F", C.c_void_p), ("ls", C.c_void_p),
                ("scale", C.c_double), ("noise", C.c_double), ("target", C.c_void_p)]


class PackHeader(C.Structure):
    """gpslc_pack_header (include/gpslc_hip.h)."""
    _fields_ = [("n", C.c_int64), ("nX", C.c_int64), ("nU", C.c_int64), ("S", C.c_int64),
                ("binary_t", C.c_int64), ("reserved", C.c_int64), ("hyper", C.c_double * 7)]

_lib = None


def header_symbols(path: str = HEADER_PATH):
    """Every function name declared in include/gpslc_hip.h."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gpslc_[a-z_0-9]+)\s*\(", txt)))


def _one_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same SONAME as
    /opt/rocm's); the dynamic loader binds every later user of that SONAME to whichever copy was loaded first,
    and a process that initialises the system copy first and torch's HSA stack second sees "No HIP GPUs".  When
    torch is installed (bench.py and sharded.py use it for device memory and the process group) its copy is
    loaded before libgpslc_hip.so, so the library and torch share a runtime whatever the import order.  Without
    torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load the library (once).  Raises GPSLCLibraryError when it is absent — by design there is
    no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GPSLCLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C causalgpslc.jl_amd/csrc`; there is no CPU fallback")
    _one_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise GPSLCLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
