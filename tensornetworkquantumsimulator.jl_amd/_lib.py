"""ctypes binding of libtnqs_hip.so (include/tnqs.h).  The HIP library is the product: there is no CPU
fallback, and importing this module fails loudly when the shared object is missing."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtnqs_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or tensornetworkquantumsimulator.jl_amd/csrc/build.sh). There is no CPU fallback.")



def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own HIP / HSA runtimes (same SONAMEs as /opt/rocm's).  Whoever loads first wins: if this
    library pulled in /opt/rocm's copies and `import torch` came later, torch would bring up a second runtime stack and find no GPU.
    Loading torch's copies first (only the shared objects -- torch itself is not imported) makes both orders share one runtime, which
    is the configuration the tests and bench.py run in."""
    import sys
    if "torch" in sys.modules or os.environ.get("TNQS_NO_TORCH_RUNTIME") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:                 # never fatal: without torch's copies the system runtime is used
        pass


_share_torch_hip_runtime()
lib = C.CDLL(LIB_PATH)

TNQS_C64, TNQS_C128, TNQS_F32, TNQS_F64 = 0, 1, 2, 3
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NUMERIC, ERR_COMM = 0, -1, -2, -3, -4, -5


class BpOpts(C.Structure):
    _fields_ = [("maxiter", C.c_int), ("tolerance", C.c_double), ("normalize", C.c_int), ("n_sequence", C.c_int),
                ("seq_src", C.POINTER(C.c_int32)), ("seq_dst", C.POINTER(C.c_int32))]


class ApplyOpts(C.Structure):
    _fields_ = [("maxdim", C.c_int), ("cutoff", C.c_double), ("normalize_tensors", C.c_int),
                ("sqrt_cutoff", C.c_double), ("update_cache", C.c_int)]


class ApplyStats(C.Structure):
    _fields_ = [("n_bp_updates", C.c_int), ("n_bp_sweeps", C.c_int), ("n_batches", C.c_int), ("n_two_site", C.c_int),
                ("bp_not_converged", C.c_int), ("last_bp_diff", C.c_double),
                ("n_chol_fallbacks", C.c_int), ("n_qr2_sites", C.c_int), ("n_lowrank_svd", C.c_int), ("n_tall_svd", C.c_int), ("n_svd_sweeps", C.c_int), ("n_svd_sweeps_max", C.c_int), ("n_deferred_1site", C.c_int), ("n_bp_products_reused", C.c_int), ("n_bp_products_evicted", C.c_int), ("n_lowrank_fallbacks", C.c_int), ("n_spec_batches", C.c_int), ("n_spec_redone", C.c_int)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)

H = C.c_void_p
_I32P = C.POINTER(C.c_int32)
_I64P = C.POINTER(C.c_int64)
_DP = C.POINTER(C.c_double)

_SIGS = {
    "tnqs_version": ([], C.c_int),
    "tnqs_last_error": ([], C.c_char_p),
    "tnqs_device_count": ([C.POINTER(C.c_int)], C.c_int),
    "tnqs_create": ([C.c_int, C.c_int, _I32P, _I32P, _I32P, C.c_int, C.c_int, C.POINTER(H)], C.c_int),
    "tnqs_destroy": ([H], C.c_int),
    "tnqs_copy": ([H, C.POINTER(H)], C.c_int),
    "tnqs_scalartype": ([H, C.POINTER(C.c_int)], C.c_int),
    "tnqs_set_stream": ([H, C.c_void_p], C.c_int),
    "tnqs_set_site_tensor": ([H, C.c_int, C.c_void_p, C.c_int, _I64P, _I32P], C.c_int),
    "tnqs_set_site_random": ([H, C.c_int, C.c_int, _I64P, C.c_uint64, C.c_double], C.c_int),
    "tnqs_get_site_tensor": ([H, C.c_int, C.c_void_p, C.c_int, _I32P], C.c_int),
    "tnqs_site_tensor_size": ([H, C.c_int, _I64P], C.c_int),
    "tnqs_set_message": ([H, C.c_int, C.c_int, C.c_void_p, C.c_int], C.c_int),
    "tnqs_get_message": ([H, C.c_int, C.c_int, C.c_void_p, C.c_int], C.c_int),
    "tnqs_bond_dim": ([H, C.c_int, C.c_int, C.POINTER(C.c_int)], C.c_int),
    "tnqs_maxvirtualdim": ([H, C.POINTER(C.c_int)], C.c_int),
    "tnqs_bp_update": ([H, C.POINTER(BpOpts), C.POINTER(C.c_int), _DP], C.c_int),
    "tnqs_apply_gates": ([H, C.c_int, _I32P, _I32P, _DP, C.POINTER(ApplyOpts), C.POINTER(BpOpts), _DP,
                          C.POINTER(ApplyStats)], C.c_int),
    "tnqs_truncate": ([H, C.c_int, C.c_double, C.c_int, C.c_int, _I32P, _I32P, _I32P, C.POINTER(BpOpts),
                       C.POINTER(ApplyStats)], C.c_int),
    "tnqs_rdm_1site": ([H, C.c_int, _DP], C.c_int),
    "tnqs_expect_1site": ([H, C.c_int, _DP, _DP], C.c_int),
    "tnqs_expect_all": ([H, _DP, _DP], C.c_int),
    "tnqs_expect_region": ([H, C.c_int, _I32P, _I32P, _DP, _DP], C.c_int),
    "tnqs_vertex_scalars": ([H, _DP], C.c_int),
    "tnqs_edge_scalars": ([H, _DP], C.c_int),
    "tnqs_rescale": ([H], C.c_int),
    "tnqs_rescale_messages": ([H, C.c_int, _I32P, _I32P], C.c_int),
    "tnqs_rescale_vertices": ([H, C.c_int, _I32P], C.c_int),
    "tnqs_symmetric_gauge": ([H, C.c_double], C.c_int),
    "tnqs_set_sharding": ([H, C.c_int, C.c_int, _I32P, ALLGATHER_FN, C.c_void_p, C.c_void_p, C.c_int64], C.c_int),
    "tnqs_rccl_unique_id": ([C.c_void_p], C.c_int),
    "tnqs_set_sharding_rccl": ([H, C.c_int, C.c_int, _I32P, C.c_void_p, C.c_int64], C.c_int),
    "tnqs_sharding_stats": ([H, _I64P, _I64P], C.c_int),
    "tnqs_rccl_selftest": ([C.c_int, C.c_int64], C.c_int),
    "tnqs_rccl_preflight": ([], C.c_int),
    "tnqs_profile_enable": ([H, C.c_int], C.c_int),
    "tnqs_profile_get": ([H, C.c_int, _I64P, _DP, _DP, _DP], C.c_int),
    "tnqs_profile_reset": ([H], C.c_int),
}
EXPORTS = tuple(_SIGS)
for _name, (_args, _res) in _SIGS.items():
    _f = getattr(lib, _name)          # AttributeError here = header and library out of sync
    _f.argtypes = _args
    _f.restype = _res


class TnqsError(RuntimeError):
    """mirrors `error(...)` in the reference (ErrorException)"""


class TnqsArgumentError(ValueError):
    """mirrors Julia ArgumentError"""


class TnqsDomainError(ArithmeticError):
    """mirrors Julia DomainError (sqrt of a negative message eigenvalue, src/utils.jl:21)"""


def check(status: int):
    if status == OK:
        return
    msg = lib.tnqs_last_error().decode("utf-8", "replace")
    if status == ERR_NUMERIC:
        raise TnqsDomainError(msg)
    raise TnqsError(msg)


def i32(arr):
    import numpy as np
    a = np.ascontiguousarray(arr, dtype=np.int32)
    return a, a.ctypes.data_as(_I32P)
