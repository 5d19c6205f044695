"""Loader of the native library ``outfit_amd/csrc/_build/liboutfit_hip.so``.

The product path has no CPU fallback: if the HIP library is missing (or cannot
be loaded) every operation fails loudly with the build instruction.
"""
import ctypes as C
import os
import subprocess

from ._abi import (ABI_VERSION, CallOptionsC, DeHeaderC, DiffCorConfigC, EphemC, GaussResultC, IODParamsC, LsqResultC, ObsBatchC,
                   ObsFitC, SolverTypeC)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "_build", "liboutfit_hip.so")

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def build_native(verbose=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.run(cmd, check=True)
    return LIB_PATH


def lib():
    """The loaded C ABI; raises NativeLibraryError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch ships its own libamdhip64 and cannot enumerate the GPU once the
    # system copy has been initialised by this library (observed: "No HIP GPUs are available"). When torch is
    # already imported, let it load its runtime first; the loader then resolves ours to the same SONAME.
    import sys
    if "torch" in sys.modules:
        try:
            sys.modules["torch"].cuda.is_available()
        except Exception:  # torch without GPU support: nothing to order
            pass
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 not found
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    vp = C.c_void_p
    L.outfit_hip_abi_version.restype = C.c_int
    if L.outfit_hip_abi_version() != ABI_VERSION:
        raise NativeLibraryError("liboutfit_hip.so ABI version mismatch: rebuild it")
    L.outfit_hip_last_error.restype = C.c_char_p
    L.outfit_hip_device_count.restype = C.c_int
    L.outfit_hip_iod_params_default.argtypes = [C.POINTER(IODParamsC)]
    L.outfit_hip_iod_params_default.restype = None
    L.outfit_hip_iod_params_validate.argtypes = [C.POINTER(IODParamsC)]
    L.outfit_hip_fit_full_iod.argtypes = [C.POINTER(IODParamsC), C.POINTER(ObsBatchC), C.POINTER(EphemC),
                                          C.c_int, C.POINTER(GaussResultC)]
    L.outfit_hip_session_create.argtypes = [C.POINTER(IODParamsC), C.POINTER(ObsBatchC), C.POINTER(EphemC),
                                            C.c_int, C.POINTER(vp)]
    L.outfit_hip_call_options_default.argtypes = [C.POINTER(CallOptionsC)]
    L.outfit_hip_call_options_default.restype = None
    L.outfit_hip_fit_full_iod_ex.argtypes = [C.POINTER(IODParamsC), C.POINTER(ObsBatchC), C.POINTER(EphemC),
                                             C.c_int, C.POINTER(CallOptionsC), C.POINTER(GaussResultC)]
    L.outfit_hip_session_create_ex.argtypes = [C.POINTER(IODParamsC), C.POINTER(ObsBatchC), C.POINTER(EphemC),
                                               C.c_int, C.POINTER(CallOptionsC), C.POINTER(vp)]
    L.outfit_hip_session_run.argtypes = [vp, vp, vp]
    L.outfit_hip_session_results.argtypes = [vp, C.POINTER(GaussResultC)]
    L.outfit_hip_session_check.argtypes = [vp]
    L.outfit_hip_session_device_results.argtypes = [vp]
    L.outfit_hip_session_device_results.restype = vp
    L.outfit_hip_session_candidate_scores.argtypes = [vp, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_uint32),
                                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.outfit_hip_session_kernel_times.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float),
                                                  C.POINTER(C.c_int)]
    L.outfit_hip_session_set_profiling.argtypes = [vp, C.c_int]
    L.outfit_hip_session_destroy.argtypes = [vp]
    L.outfit_hip_session_destroy.restype = None
    L.outfit_hip_propagate_universal.argtypes = [
        C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
        C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.outfit_hip_solver_type_default.argtypes = [C.POINTER(SolverTypeC)]
    L.outfit_hip_solver_type_default.restype = None
    L.outfit_hip_propagate_universal_ex.argtypes = [
        C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(SolverTypeC),
        C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.outfit_hip_selftest_division.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.outfit_hip_crmath_eval.argtypes = [C.c_int, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.outfit_hip_selftest_crmath.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.outfit_hip_selftest_aberth.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.outfit_hip_observer_positions.argtypes = [C.POINTER(EphemC), C.c_uint64] + [C.POINTER(C.c_double)] * 5
    L.outfit_hip_device_normals.argtypes = [C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.c_uint32,
                                            C.c_uint64, C.POINTER(C.c_double)]
    L.outfit_hip_diffcor_config_default.argtypes = [C.POINTER(DiffCorConfigC)]
    L.outfit_hip_diffcor_config_default.restype = None
    L.outfit_hip_diffcor_config_validate.argtypes = [C.POINTER(DiffCorConfigC)]
    L.outfit_hip_fit_lsq.argtypes = [C.POINTER(IODParamsC), C.POINTER(DiffCorConfigC), C.POINTER(ObsBatchC),
                                     C.POINTER(EphemC), C.POINTER(GaussResultC), C.c_int,
                                     C.POINTER(LsqResultC), C.POINTER(ObsFitC)]
    L.outfit_hip_fit_lsq_ex.argtypes = [C.POINTER(IODParamsC), C.POINTER(DiffCorConfigC), C.POINTER(ObsBatchC),
                                        C.POINTER(EphemC), C.POINTER(GaussResultC), C.c_int, C.POINTER(CallOptionsC),
                                        C.POINTER(LsqResultC), C.POINTER(ObsFitC)]
    L.outfit_hip_session_run_lsq.argtypes = [vp, C.POINTER(DiffCorConfigC), C.POINTER(GaussResultC), vp]
    L.outfit_hip_session_lsq_results.argtypes = [vp, C.POINTER(LsqResultC), C.POINTER(ObsFitC)]
    L.outfit_hip_ephem_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.outfit_hip_ephem_header.argtypes = [vp, C.POINTER(DeHeaderC)]
    L.outfit_hip_ephem_view.argtypes = [vp, C.POINTER(EphemC)]
    L.outfit_hip_ephem_body_record.argtypes = [vp, C.c_uint32, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                               C.c_uint64]
    L.outfit_hip_ephem_close.argtypes = [vp]
    L.outfit_hip_ephem_close.restype = None
    L.outfit_hip_glibc_eval.argtypes = [C.c_int, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double)]
    L.outfit_hip_calibration_stream.argtypes = [C.c_uint64, C.c_int]
    L.outfit_hip_prelim_kepuni.argtypes = [C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.outfit_hip_set_math_mode.argtypes = [C.c_int]
    L.outfit_hip_get_math_mode.restype = C.c_int
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "outfit_hip_iod_params_default", "outfit_hip_iod_params_validate", "outfit_hip_fit_full_iod",
    "outfit_hip_session_create", "outfit_hip_session_run", "outfit_hip_session_results",
    "outfit_hip_session_device_results", "outfit_hip_session_candidate_scores", "outfit_hip_session_check", "outfit_hip_session_kernel_times",
    "outfit_hip_session_set_profiling", "outfit_hip_session_destroy", "outfit_hip_propagate_universal",
    "outfit_hip_last_error", "outfit_hip_abi_version", "outfit_hip_device_count",
    "outfit_hip_selftest_division", "outfit_hip_selftest_crmath", "outfit_hip_crmath_eval", "outfit_hip_selftest_aberth", "outfit_hip_device_normals", "outfit_hip_observer_positions", "outfit_hip_diffcor_config_default", "outfit_hip_diffcor_config_validate",
    "outfit_hip_fit_lsq", "outfit_hip_session_run_lsq", "outfit_hip_session_lsq_results",
    "outfit_hip_solver_type_default", "outfit_hip_propagate_universal_ex", "outfit_hip_trim_cache", "outfit_hip_calibration_stream", "outfit_hip_glibc_eval", "outfit_hip_prelim_kepuni", "outfit_hip_set_math_mode", "outfit_hip_get_math_mode",
    "outfit_hip_call_options_default", "outfit_hip_fit_full_iod_ex", "outfit_hip_session_create_ex", "outfit_hip_fit_lsq_ex",
    "outfit_hip_selftest_abi_guard", "outfit_hip_ephem_open", "outfit_hip_ephem_header", "outfit_hip_ephem_view", "outfit_hip_ephem_close", "outfit_hip_ephem_body_record",
]


def last_error():
    return lib().outfit_hip_last_error().decode()
