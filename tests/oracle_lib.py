"""ctypes bindings of the CPU oracle (oracle/_build/liboutfit_oracle.so).

Test infrastructure only: the oracle is the checker, never the product path.
The library is built on demand with ``make -C oracle`` (gcc only).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from outfit_amd._abi import (DiffCorConfigC, EphemC, GaussResultC, IODParamsC, LSQ_DTYPE, LsqResultC,
                             OBS_FIT_DTYPE, ObsBatchC, ObsFitC, RESULT_DTYPE)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "liboutfit_oracle.so")

_dp = C.POINTER(C.c_double)
D3 = C.c_double * 3
D4 = C.c_double * 4
D8 = C.c_double * 8
D9 = C.c_double * 9
D11 = C.c_double * 11
M33 = (C.c_double * 3) * 3


class KeplerParams(C.Structure):
    _fields_ = [
        ("dt", C.c_double), ("r0", C.c_double), ("sig0", C.c_double), ("mu", C.c_double),
        ("alpha", C.c_double), ("e0", C.c_double), ("convergency", C.c_double),
        ("has_guess", C.c_int), ("psi_guess", C.c_double),
        ("max_iter_prelim_kepuni", C.c_int), ("kind", C.c_int), ("parabolic_method", C.c_int),
    ]


class KeplerSolution(C.Structure):
    _fields_ = [("psi", C.c_double), ("s0", C.c_double), ("s1", C.c_double),
                ("s2", C.c_double), ("s3", C.c_double)]


class Elements(C.Structure):
    _fields_ = [("kind", C.c_int), ("epoch", C.c_double), ("e", C.c_double * 6)]


class Equinoctial(C.Structure):
    _fields_ = [("epoch", C.c_double), ("a", C.c_double), ("h", C.c_double), ("k", C.c_double),
                ("p", C.c_double), ("q", C.c_double), ("lam", C.c_double)]


class GaussObs(C.Structure):
    _fields_ = [("idx", C.c_uint32 * 3), ("ra", D3), ("dec", D3), ("t", D3), ("obs", M33)]


class GaussPrelim(C.Structure):
    _fields_ = [("tau1", C.c_double), ("tau3", C.c_double), ("S", M33), ("Sinv", M33),
                ("a", D3), ("b", D3)]


class GaussResult(C.Structure):
    _fields_ = [("stage", C.c_int), ("el", Elements)]


D6 = C.c_double * 6
M63 = (C.c_double * 3) * 6
M66 = (C.c_double * 6) * 6


class RefSystem(C.Structure):
    _fields_ = [("kind", C.c_int), ("j2000", C.c_int), ("date", C.c_double)]


class ObsEquation(C.Structure):
    _fields_ = [("d_ra", D6), ("d_dec", D6), ("residual_ra", C.c_double), ("residual_dec", C.c_double),
                ("weight_ra", C.c_double), ("weight_dec", C.c_double), ("weight_cross", C.c_double),
                ("active", C.c_int)]


class ObsFit(C.Structure):
    _fields_ = [("sigma_ra", C.c_double), ("sigma_dec", C.c_double), ("bias_ra", C.c_double),
                ("bias_dec", C.c_double), ("residual_ra", C.c_double), ("residual_dec", C.c_double),
                ("chi", C.c_double), ("selection", C.c_int)]


class LsqSolution(C.Structure):
    _fields_ = [("dx", D6), ("normal", M66), ("covariance", M66), ("inversion_succeeded", C.c_int),
                ("used_qr", C.c_int), ("normalised_rms", C.c_double), ("num_measurements", C.c_uint32)]


class DiffCorOutput(C.Structure):
    _fields_ = [("elements", Equinoctial), ("covariance", M66), ("normal", M66),
                ("normalised_rms", C.c_double), ("total_newton_iterations", C.c_uint32),
                ("num_measurements", C.c_uint32), ("outer_passes", C.c_uint32)]


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_SO)
        L.orc_principal_angle.restype = C.c_double
        L.orc_principal_angle.argtypes = [C.c_double]
        L.orc_s_funct.argtypes = [C.c_double, C.c_double, D4]
        L.orc_prelim_kepuni.restype = C.c_double
        L.orc_prelim_kepuni.argtypes = [C.POINTER(KeplerParams)]
        L.orc_kepler_solve.argtypes = [C.POINTER(KeplerParams), C.POINTER(KeplerSolution)]
        L.orc_velocity_correction_with_guess.argtypes = [
            D3, D3, D3, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
            D3, _dp, _dp, _dp]
        L.orc_propagate_universal.argtypes = [D3, D3, C.c_double, C.c_double, C.c_double, C.c_int, D11]
        from outfit_amd._abi import SolverTypeC
        L.orc_propagate_universal_ex.argtypes = [D3, D3, C.c_double, C.c_double, C.POINTER(SolverTypeC), D11]
        L.orc_eccentricity_control.argtypes = [D3, D3, C.c_double, C.c_double,
                                               C.POINTER(C.c_int), _dp, _dp, _dp]
        L.orc_ccek1.argtypes = [D3, D3, C.c_double, C.POINTER(Elements)]
        L.orc_to_equinoctial.argtypes = [C.POINTER(Elements), C.POINTER(Equinoctial)]
        L.orc_solve_kepler_equation.argtypes = [C.POINTER(Equinoctial), C.c_double, C.c_double, _dp]
        L.orc_propagate_twobody.argtypes = [C.POINTER(Equinoctial), C.c_double, C.c_double, D3, D3]
        L.orc_aberth9.argtypes = [D9, C.c_uint32, C.c_double, D8, D8, C.POINTER(C.c_uint32)]
        L.orc_gauss_prelim_fn.argtypes = [C.POINTER(GaussObs), C.POINTER(GaussPrelim)]
        L.orc_coeff_eight_poly.argtypes = [C.POINTER(GaussObs), C.POINTER(GaussPrelim), _dp, _dp, _dp]
        L.orc_descartes_upper_bound.restype = C.c_uint32
        L.orc_descartes_upper_bound.argtypes = [C.c_double] * 4
        L.orc_solve_8poly.argtypes = [D9, C.c_uint32, C.c_double, C.c_double, D8, C.POINTER(C.c_int)]
        L.orc_position_vector_and_reference_epoch.argtypes = [
            C.POINTER(GaussObs), C.POINTER(IODParamsC), M33, M33, D3, M33, _dp]
        L.orc_gibbs_correction.argtypes = [M33, C.c_double, C.c_double, D3]
        L.orc_pos_and_vel_correction.argtypes = [
            C.POINTER(GaussObs), C.POINTER(IODParamsC), M33, D3, M33, M33, C.c_double, C.c_double,
            C.c_double, C.c_uint64, M33, D3, _dp]
        L.orc_prelim_orbit.argtypes = [C.POINTER(GaussObs), C.POINTER(IODParamsC), C.POINTER(GaussResult)]
        L.orc_generate_triplet_indices.restype = C.c_size_t
        L.orc_generate_triplet_indices.argtypes = [_dp, C.c_size_t, C.POINTER(IODParamsC),
                                                   C.POINTER(C.c_uint32), _dp]
        L.orc_downsample_uniform_with_edges.restype = C.c_size_t
        L.orc_downsample_uniform_with_edges.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_earth_ephemeris.argtypes = [C.POINTER(EphemC), C.c_double, D3]
        L.orc_topocentric_radec.argtypes = [D3, D3, D3, _dp, _dp]
        L.orc_ephemeris_error.argtypes = [C.POINTER(Equinoctial), C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_double, D3, _dp]
        L.orc_select_rms_interval.argtypes = [_dp, C.c_size_t, C.c_uint32 * 3, C.POINTER(IODParamsC),
                                              C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_fit_full_iod.argtypes = [C.POINTER(IODParamsC), C.POINTER(ObsBatchC), C.POINTER(EphemC),
                                       C.c_int, C.c_int, C.POINTER(GaussResultC)]
        L.orc_iod_params_default.argtypes = [C.POINTER(IODParamsC)]
        L.orc_iod_params_validate.argtypes = [C.POINTER(IODParamsC)]
        L.orc_diffcor_config_default.argtypes = [C.POINTER(DiffCorConfigC)]
        L.orc_is_bizarre.argtypes = [C.POINTER(Equinoctial), C.POINTER(DiffCorConfigC)]
        L.orc_compute_derivative.argtypes = [C.POINTER(Equinoctial)] + [C.c_double] * 13 + [D3, D3, D3, D3, M63, M63]
        L.orc_propagate_twobody_partials.argtypes = [C.POINTER(Equinoctial), C.c_double, C.c_double, D3, D3,
                                                     M63, M63]
        L.orc_topocentric_radec_and_partials.argtypes = [D3, D3, D3, _dp, _dp, D3, D3]
        L.orc_obs_and_partials_2body.argtypes = [C.POINTER(Equinoctial), C.c_double, D3, _dp, _dp, D6, D6]
        L.orc_angular_diff.restype = C.c_double
        L.orc_angular_diff.argtypes = [C.c_double, C.c_double]
        L.orc_solve_weighted_least_squares.argtypes = [C.POINTER(ObsEquation), C.c_size_t, C.c_uint8 * 6,
                                                       C.POINTER(LsqSolution)]
        L.orc_rescale_covariance.argtypes = [M66, M66, C.c_size_t, C.c_size_t, C.c_double]
        L.orc_update_observation_selection.restype = C.c_size_t
        L.orc_update_observation_selection.argtypes = [C.POINTER(ObsFit), C.POINTER(ObsEquation), C.c_size_t,
                                                       M66, C.c_double, C.c_double]
        L.orc_run_differential_correction.argtypes = [C.POINTER(Equinoctial), _dp, _dp, _dp, _dp, C.c_size_t,
                                                      C.POINTER(ObsFit), C.POINTER(DiffCorConfigC),
                                                      C.POINTER(DiffCorOutput)]
        L.orc_obleq.restype = C.c_double
        L.orc_obleq.argtypes = [C.c_double]
        L.orc_nutn80.argtypes = [C.c_double, _dp, _dp]
        L.orc_rnut80.argtypes = [C.c_double, M33]
        L.orc_equequ.restype = C.c_double
        L.orc_equequ.argtypes = [C.c_double]
        L.orc_prec.argtypes = [C.c_double, M33]
        L.orc_rotmt.argtypes = [C.c_double, C.c_int, M33]
        L.orc_rotpn.argtypes = [C.POINTER(RefSystem), C.POINTER(RefSystem), M33]
        L.orc_gmst.restype = C.c_double
        L.orc_gmst.argtypes = [C.c_double]
        L.orc_pvobs.argtypes = [C.c_double, C.c_double, D3, D3]
        L.orc_helio_position.argtypes = [D3, D3, D3]
        L.orc_observer_prepass.argtypes = [C.POINTER(EphemC), C.c_double, C.c_double, D3, D3, D3]
        L.orc_observer_positions.argtypes = [C.POINTER(EphemC), C.c_uint64] + [_dp] * 5
        L.orc_fit_lsq.argtypes = [C.POINTER(IODParamsC), C.POINTER(DiffCorConfigC), C.POINTER(ObsBatchC),
                                  C.POINTER(EphemC), C.POINTER(GaussResultC), C.c_int,
                                  C.POINTER(LsqResultC), C.POINTER(ObsFitC)]
        _lib = L
    return _lib


class math_flags:
    """Context manager for the oracle's elementary-function switches (oracle.h): `libm` = orc_use_libm (platform libm
    everywhere: the reference-faithful mode on a glibc host), `glibc` = orc_math_glibc (sin/cos/sinh/cosh restated
    from glibc 2.35, what the device's OUTFIT_MATH_GLIBC computes). None leaves a switch as it is; both are restored."""

    def __init__(self, libm=None, glibc=None):
        self.want = {"orc_use_libm": libm, "orc_math_glibc": glibc}

    def __enter__(self):
        self.prev = {}
        for name, v in self.want.items():
            f = C.c_int.in_dll(lib(), name)
            self.prev[name] = f.value
            if v is not None:
                f.value = int(v)
        return self

    def __exit__(self, *exc):
        for name, v in self.prev.items():
            C.c_int.in_dll(lib(), name).value = v


def default_params():
    p = IODParamsC()
    lib().orc_iod_params_default(C.byref(p))
    return p


def mat33(rows):
    """rows[r][c] -> C double[3][3]"""
    m = M33()
    for r in range(3):
        for c in range(3):
            m[r][c] = rows[r][c]
    return m


def mat33_from_colmajor(flat9):
    """nalgebra as_slice() order (column-major) -> C double[3][3] (m[row][col])."""
    m = M33()
    for c in range(3):
        for r in range(3):
            m[r][c] = flat9[3 * c + r]
    return m


def mat33_to_colmajor(m):
    return [m[r][c] for c in range(3) for r in range(3)]


def gauss_obs(idx, ra, dec, t, obs_cols):
    """obs_cols: three [x,y,z] columns (one per epoch)."""
    g = GaussObs()
    for k in range(3):
        g.idx[k] = idx[k]
        g.ra[k] = ra[k]
        g.dec[k] = dec[k]
        g.t[k] = t[k]
        for r in range(3):
            g.obs[r][k] = obs_cols[k][r]
    return g


def fit_full_iod(params, batch_c, ephem_c=None, n_threads=1, hoist=0):
    n = batch_c.n_traj
    out = (GaussResultC * n)()
    rc = lib().orc_fit_full_iod(C.byref(params), C.byref(batch_c),
                                C.byref(ephem_c) if ephem_c is not None else None,
                                n_threads, hoist, out)
    arr = np.frombuffer(out, dtype=RESULT_DTYPE).copy()
    return rc, arr


def observer_positions(ephem_c, mjd_tt, mjd_ut1, body_fixed):
    """(rc, geo_ecl [n,3], helio_equ [n,3]) — the observer pre-pass of OutfitCache::build over a batch."""
    tt = np.ascontiguousarray(mjd_tt, dtype=np.float64)
    ut = np.ascontiguousarray(mjd_ut1, dtype=np.float64)
    fx = np.ascontiguousarray(body_fixed, dtype=np.float64).reshape(-1)
    n = tt.shape[0]
    geo, hel = np.empty(3 * n), np.empty(3 * n)
    rc = lib().orc_observer_positions(C.byref(ephem_c), n, tt.ctypes.data_as(_dp), ut.ctypes.data_as(_dp),
                                      fx.ctypes.data_as(_dp), geo.ctypes.data_as(_dp), hel.ctypes.data_as(_dp))
    return rc, geo.reshape(n, 3), hel.reshape(n, 3)


def default_diffcor_config():
    c = DiffCorConfigC()
    lib().orc_diffcor_config_default(C.byref(c))
    return c


def fit_lsq(params, config, batch_c, ephem_c=None, initial=None, n_threads=1):
    """initial: numpy RESULT_DTYPE array or None. Returns (rc, LSQ_DTYPE array, OBS_FIT_DTYPE array)."""
    n = batch_c.n_traj
    n_obs = batch_c.traj_offset[n] if n else 0
    out = (LsqResultC * max(n, 1))()
    fit = (ObsFitC * max(n_obs, 1))()
    init_c = None
    if initial is not None:
        init_c = (GaussResultC * max(n, 1)).from_buffer_copy(np.ascontiguousarray(initial).tobytes())
    rc = lib().orc_fit_lsq(C.byref(params), C.byref(config), C.byref(batch_c),
                           C.byref(ephem_c) if ephem_c is not None else None, init_c, n_threads, out, fit)
    res = np.frombuffer(out, dtype=LSQ_DTYPE)[:n].copy()
    obs = np.frombuffer(fit, dtype=OBS_FIT_DTYPE)[:n_obs].copy()
    return rc, res, obs
