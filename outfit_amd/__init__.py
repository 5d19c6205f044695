"""outfit_amd — MI355X-native batched Gauss initial orbit determination.

One hot path: the reference's ``FitIOD::fit_full_iod`` (FusRoman/Outfit,
src/initial_orbit_determination/obs_dataset_api.rs:145-207), re-implemented as
hand-written HIP kernels for gfx950 behind a C ABI (include/outfit_hip.h).
"""
from .api import (  # noqa: F401
    DifferentialCorrection, DifferentialCorrectionConfig, EquinoctialElements, fit_lsq, fit_lsq_records,
    device_normals, observer_positions, lsq_record_to_result, Ephemeris, EphemerisFileError, GaussResult, IODGauss, IODParams, IODParamsBuilder, IODSession, InvalidBatch,
    InvalidConversion, InvalidIODParameter, InvalidOrbit, NoFeasibleTriplets, NoViableOrbit, ObsBatch,
    OrbitalElements, OutfitError, fit_full_iod, fit_full_iod_records, fit_iod, propagate_universal,
    SolverType, MATH_DEFAULT, MATH_CORRECTLY_ROUNDED, MATH_GLIBC, set_math_mode, get_math_mode, math_mode,
)
from ._lib import NativeLibraryError, build_native  # noqa: F401
