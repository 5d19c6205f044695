"""ctypes mirror of ``include/outfit_hip.h`` (the C-ABI drop-in boundary).

Field order and types follow the header exactly; the header cites the
reference interface each struct replaces (``IODParams``:
src/initial_orbit_determination/mod.rs:225-266, ``FitIOD`` inputs:
src/initial_orbit_determination/obs_dataset_api.rs:145-296).
"""
import ctypes as C

ABI_VERSION = 7

OUTFIT_OK = 0
OUTFIT_ERR_INVALID_PARAM = -1
OUTFIT_ERR_INVALID_BATCH = -2
OUTFIT_ERR_NO_DEVICE = -3
OUTFIT_ERR_HIP = -4
OUTFIT_ERR_EPHEM_RANGE = -5
OUTFIT_ERR_BAD_HANDLE = -6
OUTFIT_ERR_EPHEM_FILE = -7
OUTFIT_ERR_NO_MEMORY = -8
OUTFIT_ERR_INTERNAL = -9

TRAJ_OK = 0
TRAJ_NO_FEASIBLE_TRIPLETS = 1
TRAJ_NO_VIABLE_ORBIT = 2
TRAJ_INVALID_CONVERSION = 3
TRAJ_INVALID_ORBIT = 4

CAUSE_NONE = 0
CAUSE_GAUSS_NO_ROOTS_FOUND = 1
CAUSE_SINGULAR_DIRECTION_MATRIX = 2
CAUSE_POLYNOMIAL_ROOT_FINDING_FAILED = 3
CAUSE_NON_FINITE_SCORE = 4

STAGE_PRELIM = 0
STAGE_CORRECTED = 1
KIND_KEPLERIAN = 0
KIND_COMETARY = 1

_dp = C.POINTER(C.c_double)


class IODParamsC(C.Structure):
    _fields_ = [
        ("n_noise_realizations", C.c_uint64),
        ("noise_scale", C.c_double),
        ("extf", C.c_double),
        ("dtmax", C.c_double),
        ("dt_min", C.c_double),
        ("dt_max_triplet", C.c_double),
        ("optimal_interval_time", C.c_double),
        ("max_obs_for_triplets", C.c_uint64),
        ("max_triplets", C.c_uint32),
        ("_pad0", C.c_uint32),
        ("gap_max", C.c_double),
        ("max_ecc", C.c_double),
        ("max_perihelion_au", C.c_double),
        ("min_rho2_au", C.c_double),
        ("aberth_max_iter", C.c_uint32),
        ("_pad1", C.c_uint32),
        ("aberth_eps", C.c_double),
        ("kepler_eps", C.c_double),
        ("max_tested_solutions", C.c_uint64),
        ("r2_min_au", C.c_double),
        ("r2_max_au", C.c_double),
        ("newton_eps", C.c_double),
        ("newton_max_it", C.c_uint64),
        ("root_imag_eps", C.c_double),
    ]


class ObsBatchC(C.Structure):
    _fields_ = [
        ("n_traj", C.c_uint64),
        ("traj_offset", C.POINTER(C.c_uint64)),
        ("mjd_tt", _dp),
        ("ra", _dp),
        ("dec", _dp),
        ("sigma_ra", _dp),
        ("sigma_dec", _dp),
        ("obs_helio_equ", _dp),
        ("obs_geo_ecl", _dp),
        ("noise_z", _dp),
        ("noise_source", C.c_uint32),
        ("_pad0", C.c_uint32),
        ("noise_seed", C.c_uint64),
        ("traj_key", C.POINTER(C.c_uint64)),
        ("traj_key_base", C.c_uint64),
        ("obs_body_fixed", _dp),
        ("mjd_ut1", _dp),
    ]


class EphemC(C.Structure):
    _fields_ = [
        ("jd_start", C.c_double),
        ("jd_end", C.c_double),
        ("block_days", C.c_double),
        ("emrat", C.c_double),
        ("ncoef", C.c_uint32 * 3),
        ("nsub", C.c_uint32 * 3),
        ("n_blocks", C.c_uint64),
        ("coef", _dp),
    ]


class GaussResultC(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("cause", C.c_int32),
        ("stage", C.c_int32),
        ("kind", C.c_int32),
        ("attempts", C.c_uint32),
        ("best_triplet", C.c_uint32),
        ("best_realization", C.c_uint32),
        ("n_triplets", C.c_uint32),
        ("epoch", C.c_double),
        ("elem", C.c_double * 6),
        ("rms", C.c_double),
    ]


assert C.sizeof(GaussResultC) == 96
assert C.sizeof(IODParamsC) == 176

# numpy view of outfit_gauss_result_t (same 96-byte layout)
RESULT_DTYPE = [
    ("status", "<i4"), ("cause", "<i4"), ("stage", "<i4"), ("kind", "<i4"),
    ("attempts", "<u4"), ("best_triplet", "<u4"), ("best_realization", "<u4"),
    ("n_triplets", "<u4"), ("epoch", "<f8"), ("elem", "<f8", (6,)), ("rms", "<f8"),
]


# ---- differential correction (FitLSQ::fit_lsq, differential_orbit_correction/) --------------
PROPAGATOR_TWOBODY = 0
LSQ_DIFFCOR = 0
LSQ_IOD_KEPT = 1
LSQ_IOD_ERROR = 2
LSQ_CONVERSION_ERROR = 3
DC_NONE = 0
DC_FAILED = 1
DC_BIZARRE = 2
DC_DIVERGED = 3
OBS_ACTIVE = 0
OBS_REJECTED = 1
OBS_FORCED_OUT = 2
NOISE_HOST = 0
NOISE_DEVICE = 1


class CallOptionsC(C.Structure):
    """outfit_hip_call_options_t (ABI v7): per-call settings; math_mode -1 = the process-wide default; the two
    capacities (0 = not stated) let the library refuse an offset table that runs past the caller's arrays."""
    _fields_ = [("struct_size", C.c_uint32), ("math_mode", C.c_int32), ("n_obs_capacity", C.c_uint64),
                ("noise_z_capacity", C.c_uint64), ("reserved", C.c_uint64 * 1)]


class SolverTypeC(C.Structure):
    """outfit_solver_type_t = SolverType { kind, params: SolverParams } (kepler/params.rs:24-73)"""
    _fields_ = [("kind", C.c_int32), ("parabolic_solving_method", C.c_int32), ("convergency", C.c_double),
                ("max_iter_prelim_kepuni", C.c_uint64), ("has_psi_guess", C.c_int32), ("_pad", C.c_int32),
                ("psi_guess", C.c_double)]


SOLVER_NEWTON_RAPHSON, SOLVER_BRENT_DECKER, SOLVER_AUTO = 0, 1, 2
PARABOLIC_CARDANO, PARABOLIC_NEWTON = 0, 1


class DiffCorConfigC(C.Structure):
    _fields_ = [
        ("max_newton_iterations", C.c_uint64),
        ("max_outlier_rejection_passes", C.c_uint64),
        ("convergence_threshold", C.c_double),
        ("convergence_before_rejection_threshold", C.c_double),
        ("rms_stagnation_ratio", C.c_double),
        ("rms_divergence_ratio", C.c_double),
        ("max_stagnation_iterations", C.c_uint64),
        ("enable_outlier_rejection", C.c_uint32),
        ("propagator", C.c_uint32),
        ("chi_squared_rejection_threshold", C.c_double),
        ("chi_squared_recovery_threshold", C.c_double),
        ("eccentricity_limit", C.c_double),
        ("min_semi_major_axis", C.c_double),
        ("max_semi_major_axis", C.c_double),
        ("min_periapsis_distance", C.c_double),
        ("max_apoapsis_distance", C.c_double),
        ("free_elements", C.c_uint8 * 6),
        ("_pad", C.c_uint8 * 2),
    ]


class LsqResultC(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("dc_error", C.c_int32),
        ("conv_error", C.c_int32),
        ("total_newton_iterations", C.c_uint32),
        ("num_measurements", C.c_uint32),
        ("n_rejected", C.c_uint32),
        ("outer_passes", C.c_uint32),
        ("_pad", C.c_uint32),
        ("epoch", C.c_double),
        ("elem", C.c_double * 6),
        ("normalised_rms", C.c_double),
        ("covariance", C.c_double * 36),
        ("iod", GaussResultC),
    ]


class ObsFitC(C.Structure):
    _fields_ = [
        ("residual_ra", C.c_double),
        ("residual_dec", C.c_double),
        ("chi", C.c_double),
        ("selection", C.c_int32),
        ("_pad", C.c_int32),
    ]


assert C.sizeof(DiffCorConfigC) == 128
assert C.sizeof(LsqResultC) == 480
assert C.sizeof(ObsFitC) == 32

LSQ_DTYPE = [
    ("status", "<i4"), ("dc_error", "<i4"), ("conv_error", "<i4"), ("total_newton_iterations", "<u4"),
    ("num_measurements", "<u4"), ("n_rejected", "<u4"), ("outer_passes", "<u4"), ("_pad", "<u4"),
    ("epoch", "<f8"), ("elem", "<f8", (6,)), ("normalised_rms", "<f8"), ("covariance", "<f8", (6, 6)),
    ("iod", RESULT_DTYPE),
]
OBS_FIT_DTYPE = [("residual_ra", "<f8"), ("residual_dec", "<f8"), ("chi", "<f8"),
                 ("selection", "<i4"), ("_pad", "<i4")]


class DeHeaderC(C.Structure):
    """`outfit_de_header_t` (HorizonHeader, jpl_ephem/horizon/horizon_data.rs:130-160)."""
    _fields_ = [
        ("start_period", C.c_double), ("end_period", C.c_double), ("period_length", C.c_double),
        ("earth_moon_mass_ratio", C.c_double), ("au_km", C.c_double),
        ("ipt", (C.c_uint32 * 3) * 15), ("numde", C.c_uint32), ("ncon", C.c_int32),
        ("recsize", C.c_uint64), ("n_blocks", C.c_uint64),
    ]


assert C.sizeof(DeHeaderC) == 40 + 180 + 8 + 16 + 4  # 4 bytes of padding after the 15x3 u32 + numde + ncon
