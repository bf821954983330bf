"""ctypes binding of libwtp.so (include/wtp.h).  Loading fails loudly: there is no CPU path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# WTP_LIB selects another build of the same sources (A/B experiments); default: csrc/libwtp.so
SO_PATH = os.environ.get("WTP_LIB") or os.path.join(CSRC, "libwtp.so")

WTP_OK, WTP_ERR_ARG, WTP_ERR_OOM, WTP_ERR_HIP, WTP_ERR_STATE, WTP_ERR_NO_DEVICE = range(6)
WTP_F32, WTP_F64 = 0, 1
WTP_SPACING_CONSTANT, WTP_SPACING_PER_POINT, WTP_SPACING_LOGLIKE, WTP_SPACING_BOUNDARY_LAYER = range(4)


class WtpError(RuntimeError):
    """Non-argument failure reported by libwtp (the Julia shim raises ErrorException)."""

    def __init__(self, code, msg):
        super().__init__(f"libwtp status {code}: {msg}")
        self.code = code


class WtpArgumentError(ValueError):
    """WTP_ERR_ARG: the reference throws ArgumentError for the same input (src/repel.jl:74)."""


class ForceDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("beta", C.c_double), ("u0", C.c_double), ("gamma", C.c_double)]


class SpacingDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("constant", C.c_double), ("per_point", C.c_void_p),
                ("p0", C.c_double), ("p1", C.c_double), ("p2", C.c_double),
                ("boundary_xyz", C.c_void_p), ("n_boundary", C.c_int64)]


class StepStats(C.Structure):
    _fields_ = [
        ("max_force", C.c_double), ("sum_u", C.c_double), ("sum_u2", C.c_double),
        ("n_move", C.c_int64), ("argmin_i", C.c_int64), ("argmin_j", C.c_int64),
        ("argmin_r", C.c_double), ("n_fallback", C.c_int64), ("n_uncovered", C.c_int64),
        ("n_escaped", C.c_int64),
    ]


class BlockDesc(C.Structure):
    _fields_ = [("rank", C.c_int32), ("nranks", C.c_int32), ("boxes", C.POINTER(C.c_double)),
                ("ghost_width", C.c_double), ("margin", C.c_double)]


class BlockInfo(C.Structure):
    _fields_ = [("n_owned", C.c_int64), ("n_ghost", C.c_int64), ("n_sent_rows", C.c_int64), ("n_recv_rows", C.c_int64),
                ("n_emigrated", C.c_int64), ("n_immigrated", C.c_int64), ("n_peers", C.c_int32), ("widened", C.c_int32),
                ("host_syncs", C.c_int32), ("redone", C.c_int32), ("ghost_width", C.c_double), ("overlapped", C.c_int32),
                ("reserved", C.c_int32)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_int64))


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allgather", ALLGATHER_FN), ("exchange", EXCHANGE_FN)]


# every symbol include/wtp.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
SIGNATURES = {
    "wtp_create": (_i, [C.POINTER(_i), _i, C.POINTER(_vp)]),
    "wtp_destroy": (_i, [_vp]),
    "wtp_last_error": (C.c_char_p, [_vp]),
    "wtp_version": (C.c_char_p, []),
    "wtp_knn": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "wtp_knn_dev": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "wtp_radius_count": (_i, [_vp, _vp, _i64, _i, _i, _d, _vp]),
    "wtp_radius_fill": (_i, [_vp, _vp, _vp]),
    "wtp_radius_offsets": (_i, [_vp, _vp, _i64, _i, _i, _d, _vp]),
    "wtp_relax_init": (_i, [_vp, _vp, _i64, _i64, _i, _i, C.POINTER(SpacingDesc), C.POINTER(ForceDesc), _i, _d, _d]),
    "wtp_relax_init_dev": (_i, [_vp, _vp, _i64, _i64, _i, _i, C.POINTER(SpacingDesc), C.POINTER(ForceDesc), _i, _d, _d]),
    "wtp_relax_step": (_i, [_vp, _i, C.POINTER(StepStats)]),
    "wtp_relax_run": (_i, [_vp, _i, _i, _vp, C.POINTER(StepStats)]),
    "wtp_relax_run_until": (_i, [_vp, _i, _i, _d, _i, _d, _vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(StepStats)]),
    "wtp_relax_get": (_i, [_vp, _vp]),
    "wtp_relax_get_dev": (_i, [_vp, _vp]),
    "wtp_relax_get_point_data": (_i, [_vp, _vp, _vp, _vp]),
    "wtp_relax_set": (_i, [_vp, _i64, _vp]),
    "wtp_relax_revert": (_i, [_vp]),
    "wtp_relax_set_batch": (_i, [_vp, _vp, _vp, _i64]),
    "wtp_relax_set_spacing": (_i, [_vp, _vp]),
    "wtp_relax_end": (_i, [_vp]),
    "wtp_relax_get_spacing": (_i, [_vp, _vp]),
    "wtp_spacing_eval": (_i, [_vp, C.POINTER(SpacingDesc), _vp, _i64, _i, _i, _vp]),
    "wtp_pca_normals": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp]),
    "wtp_gradient_limit": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp, _d, _d, _i, _vp, C.POINTER(_i)]),
    "wtp_mesh_set": (_i, [_vp, _vp, _i64, _vp, _i64, _i]),
    "wtp_mesh_clear": (_i, [_vp]),
    "wtp_mesh_face_normals": (_i, [_vp, _vp]),
    "wtp_mesh_bounds": (_i, [_vp, _vp]),
    "wtp_mesh_query": (_i, [_vp, _vp, _i64, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "wtp_relax_set_wall": (_i, [_vp, _i64, _d]),
    "wtp_relax_get_wall": (_i, [_vp, _vp, _vp, _vp, _i]),
    "wtp_relax_set_wall_flags": (_i, [_vp, _vp, _vp]),
    "wtp_relax_query_knn": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "wtp_isinside_greens": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "wtp_isinside_winding": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _vp, _vp]),
    "wtp_set_stream": (_i, [_vp, _vp, _i]),
    "wtp_relax_layers_dev": (_i, [_vp, _i, _d, _d, _d, _d, _vp, _vp, _i64, C.POINTER(_i64)]),
    "wtp_relax_set_fixed_dev": (_i, [_vp, _vp, _i64]),
    "wtp_relax_step_layers": (_i, [_vp, _i, C.POINTER(StepStats), _i, _d, _d, _d, _d, _vp, _vp, _i64, C.POINTER(_i64)]),
    "wtp_relax_step_layers3": (_i, [_vp, _i, C.POINTER(StepStats), _i, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_vp), C.POINTER(_vp), _i64,
                                    C.POINTER(_i64)]),
    "wtp_relax_set_coverage": (_i, [_vp, _i, _d, _d]),
    "wtp_relax_set_coverage_box": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "wtp_comm_unique_id": (_i, [_vp, _vp]),
    "wtp_comm_init": (_i, [_vp, _vp, _i, _i]),
    "wtp_comm_finalize": (_i, [_vp]),
    "wtp_comm_exchange_rows": (_i, [_vp, _i, _i, _vp, _i64, _vp, _i64, _vp, _vp, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "wtp_comm_allreduce_stats": (_i, [_vp, C.POINTER(StepStats)]),
    "wtp_block_set_transport": (_i, [_vp, C.POINTER(Transport)]),
    "wtp_block_open": (_i, [_vp, C.POINTER(BlockDesc), _vp, _vp, _i64, C.POINTER(SpacingDesc), C.POINTER(ForceDesc), _i, _d, _d]),
    "wtp_block_step": (_i, [_vp, C.POINTER(StepStats), C.POINTER(BlockInfo)]),
    "wtp_block_run": (_i, [_vp, _i, _vp, C.POINTER(StepStats), C.POINTER(BlockInfo)]),
    "wtp_block_run_until": (_i, [_vp, _i, _d, _i, _d, _vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(StepStats)]),
    "wtp_block_get": (_i, [_vp, _vp, _vp, _i64, C.POINTER(_i64)]),
    "wtp_block_close": (_i, [_vp]),
    "wtp_block_grid": (_i, [_i, C.POINTER(_i)]),
    "wtp_block_morton_rank": (_i, [_i, _i, _i, C.POINTER(_i)]),
    "wtp_comm_exchange_peers": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_vp), C.POINTER(_i64)]),
    "wtp_comm_allgather_dev": (_i, [_vp, _vp, _vp, _i64]),
    "wtp_timers_get": (_i, [_vp, C.POINTER(_d)]),
    "wtp_timers_reset": (_i, [_vp]),
    "wtp_debug_diag": (_i, [_vp, C.POINTER(C.c_ulonglong)]),
    "wtp_gen_uniform_dev": (_i, [_vp, C.c_uint64, _i64, _i64, _i, _i, _vp]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile libwtp.so for gfx950 with hipcc (csrc/Makefile).  Works without a GPU."""
    res = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libwtp.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    return SO_PATH


def load():
    """dlopen libwtp.so and bind every symbol of include/wtp.h (no compute is called)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 and a process
    # that maps the system copy first (through libwtp) leaves torch without devices ("No HIP GPUs
    # are available", observed).  Mapping torch's copy first makes both sides share it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx_handle, rc: int):
    if rc == WTP_OK:
        return
    msg = load().wtp_last_error(ctx_handle)
    msg = msg.decode("utf-8", "replace") if msg else ""
    if rc == WTP_ERR_ARG:
        raise WtpArgumentError(msg)
    raise WtpError(rc, msg)
