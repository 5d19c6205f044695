"""Host-side mirror of the reference's `FitIOD` surface, on top of the C ABI.

The names follow the reference so that tests read like its own:
``IODParams`` / ``IODParams.builder()`` (src/initial_orbit_determination/mod.rs:225-630),
``fit_full_iod`` (obs_dataset_api.rs:145-207), ``GaussResult`` (gauss_result.rs:99-102),
``FitOrbitResult.IODGauss`` / ``FullOrbitResult`` (constants.rs:134-195) and the
``OutfitError`` variants the path can return (outfit_errors.rs:195-250).

What stays on the caller's side of the boundary (as in the reference, where it is
`prepare_iod`, obs_dataset_api.rs:254-275): the astrometric error model and the
observer-position cache; their outputs are the columns of :class:`ObsBatch`.
All computation happens in liboutfit_hip.so on the GPU; there is no CPU fallback.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _abi
from ._abi import (CallOptionsC, DeHeaderC, DiffCorConfigC, EphemC, GaussResultC, IODParamsC, LSQ_DTYPE, LsqResultC, OBS_FIT_DTYPE,
                   ObsBatchC, ObsFitC, RESULT_DTYPE)
from ._lib import NativeLibraryError, last_error, lib

_dp = C.POINTER(C.c_double)


# ----------------------------------------------------------------------------- errors
class OutfitError(Exception):
    """Base of the error variants of outfit_errors.rs that this path can produce."""


class InvalidIODParameter(OutfitError):
    pass


class InvalidBatch(OutfitError, ValueError):
    """OUTFIT_ERR_INVALID_BATCH, and what ObsBatch.validate() raises before a malformed batch crosses the C ABI."""


class EphemerisRange(OutfitError):
    """The reference panics here (horizon_data.rs:722-724); the ABI reports it."""


class HipRuntimeError(OutfitError):
    pass


class EphemerisFileError(OutfitError):
    """The DE file cannot be opened or parsed (the reference panics: horizon_data.rs:622-629)."""


@dataclass
class NoFeasibleTriplets(OutfitError):
    span: float
    n_obs: int
    dt_min: float
    dt_max: float


@dataclass
class NoViableOrbit(OutfitError):
    cause: str
    attempts: int


class InvalidConversion(OutfitError):
    pass


class InvalidOrbit(OutfitError):
    pass


_CAUSES = {
    _abi.CAUSE_GAUSS_NO_ROOTS_FOUND: "GaussNoRootsFound",
    _abi.CAUSE_SINGULAR_DIRECTION_MATRIX: "SingularDirectionMatrix",
    _abi.CAUSE_POLYNOMIAL_ROOT_FINDING_FAILED: "PolynomialRootFindingFailed",
    _abi.CAUSE_NON_FINITE_SCORE: "NonFiniteScore",
}


def _raise_rc(rc):
    msg = last_error()
    if rc == _abi.OUTFIT_ERR_INVALID_PARAM:
        raise InvalidIODParameter(msg)
    if rc == _abi.OUTFIT_ERR_INVALID_BATCH:
        raise InvalidBatch(msg)
    if rc == _abi.OUTFIT_ERR_EPHEM_RANGE:
        raise EphemerisRange(msg)
    if rc == _abi.OUTFIT_ERR_NO_DEVICE:
        raise NativeLibraryError(msg)
    if rc == _abi.OUTFIT_ERR_EPHEM_FILE:
        raise EphemerisFileError(msg)
    if rc == _abi.OUTFIT_ERR_NO_MEMORY:   # a host allocation inside the library failed (std::bad_alloc)
        raise MemoryError(msg)
    if rc == _abi.OUTFIT_ERR_INTERNAL:    # any other C++ exception inside the library: a bug, not a resource limit
        raise NativeLibraryError(msg)
    raise HipRuntimeError(f"rc={rc}: {msg}")


# ----------------------------------------------------------------------------- params
_PARAM_FIELDS = [n for n, _ in IODParamsC._fields_ if not n.startswith("_")]


class IODParams:
    """`IODParams` (mod.rs:225-266). Defaults = `IODParams::default()` (mod.rs:308-344)."""

    def __init__(self, **kw):
        c = IODParamsC()
        lib().outfit_hip_iod_params_default(C.byref(c))
        for n in _PARAM_FIELDS:
            setattr(self, n, getattr(c, n))
        for k, v in kw.items():
            if k not in _PARAM_FIELDS:
                raise TypeError(f"unknown IODParams field {k!r}")
            setattr(self, k, v)

    @staticmethod
    def builder():
        return IODParamsBuilder()

    def as_c(self):
        c = IODParamsC()
        for n in _PARAM_FIELDS:
            setattr(c, n, getattr(self, n))
        return c

    def __repr__(self):
        return "IODParams(" + ", ".join(f"{n}={getattr(self, n)!r}" for n in _PARAM_FIELDS) + ")"


class IODParamsBuilder:
    """`IODParamsBuilder` (mod.rs:348-630): fluent setters + validated `build()`."""

    def __init__(self):
        self._p = IODParams()

    def __getattr__(self, name):
        if name in _PARAM_FIELDS:
            def setter(v):
                setattr(self._p, name, v)
                return self
            return setter
        raise AttributeError(name)

    def build(self):
        c = self._p.as_c()
        rc = lib().outfit_hip_iod_params_validate(C.byref(c))
        if rc != 0:
            raise InvalidIODParameter(last_error())
        return self._p


# ----------------------------------------------------------------------------- inputs
def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Ephemeris:
    """Legacy-DE Chebyshev subset (EMB, Moon, Sun) = `outfit_ephem_t`."""

    def __init__(self, jd_start, jd_end, block_days, emrat, ncoef, nsub, n_blocks, coef):
        self.jd_start, self.jd_end, self.block_days, self.emrat = jd_start, jd_end, block_days, emrat
        self.ncoef, self.nsub, self.n_blocks = tuple(ncoef), tuple(nsub), int(n_blocks)
        self.coef = _f64(coef)

    @classmethod
    def from_dict(cls, d):
        return cls(d["jd_start"], d["jd_end"], d["block_days"], d["emrat"], d["ncoef"], d["nsub"],
                   d["n_blocks"], d["coef"])

    @classmethod
    def from_file(cls, path):
        """`HorizonData::read_horizon_file` (horizon_data.rs:598-684) through `outfit_hip_ephem_open`: the
        EMB / Moon / Sun segments of a little-endian legacy-DE binary. Sets `.header` (dict)."""
        L = lib()
        h = C.c_void_p()
        rc = L.outfit_hip_ephem_open(str(path).encode(), C.byref(h))
        if rc != 0:
            _raise_rc(rc)
        try:
            view, hd = EphemC(), DeHeaderC()
            L.outfit_hip_ephem_view(h, C.byref(view))
            L.outfit_hip_ephem_header(h, C.byref(hd))
            per = sum(3 * view.ncoef[b] * view.nsub[b] for b in range(3))
            coef = np.ctypeslib.as_array(view.coef, shape=(view.n_blocks * per,)).copy()
            e = cls(view.jd_start, view.jd_end, view.block_days, view.emrat, tuple(view.ncoef), tuple(view.nsub),
                    view.n_blocks, coef)
            e.header = {"jpl_version": f"DE{hd.numde}", "ipt": [list(r) for r in hd.ipt],
                        "start_period": hd.start_period, "end_period": hd.end_period,
                        "period_lenght": hd.period_length, "recsize": hd.recsize,
                        "earth_moon_mass_ratio": hd.earth_moon_mass_ratio, "ncon": hd.ncon, "n_blocks": hd.n_blocks}
            return e
        finally:
            L.outfit_hip_ephem_close(h)

    @staticmethod
    def body_records(path, ipt_row, block):
        """`HorizonData.records[block][ipt_row]` through `outfit_hip_ephem_body_record`: (array [n_sub, axes, n_coeffs],
        (start_jd, end_jd)) — any body of the file, not only the three the path consumes."""
        L = lib()
        h = C.c_void_p()
        rc = L.outfit_hip_ephem_open(str(path).encode(), C.byref(h))
        if rc != 0:
            _raise_rc(rc)
        try:
            hd = DeHeaderC()
            L.outfit_hip_ephem_header(h, C.byref(hd))
            nc, ns = int(hd.ipt[ipt_row][1]), int(hd.ipt[ipt_row][2])
            buf = np.empty(max(1, 3 * nc * ns))
            jd = (C.c_double * 2)()
            n = L.outfit_hip_ephem_body_record(h, ipt_row, block, jd, buf.ctypes.data_as(_dp), buf.size)
            if n < 0:
                _raise_rc(n)
            return buf[:n].reshape(ns, n // (ns * nc), nc).copy(), (jd[0], jd[1])
        finally:
            L.outfit_hip_ephem_close(h)

    def as_c(self):
        e = EphemC()
        e.jd_start, e.jd_end, e.block_days, e.emrat = self.jd_start, self.jd_end, self.block_days, self.emrat
        for i in range(3):
            e.ncoef[i] = self.ncoef[i]
            e.nsub[i] = self.nsub[i]
        e.n_blocks = self.n_blocks
        e.coef = self.coef.ctypes.data_as(_dp)
        e._owner = self  # the struct holds bare pointers: keep the arrays alive as long as it lives
        return e


class ObsBatch:
    """All trajectories of a prepared dataset = `outfit_obs_batch_t` (SoA, time-sorted per trajectory).

    `traj_ids` are the caller's trajectory identifiers (the keys of `FullOrbitResult`)."""

    def __init__(self, traj_offset, mjd_tt, ra, dec, sigma_ra, sigma_dec, obs_helio_equ,
                 obs_geo_ecl=None, noise_z=None, traj_ids=None, noise_seed=None, traj_key=None, traj_key_base=0,
                 obs_body_fixed=None, mjd_ut1=None):
        self.traj_offset = np.ascontiguousarray(traj_offset, dtype=np.uint64)
        self.mjd_tt, self.ra, self.dec = _f64(mjd_tt), _f64(ra), _f64(dec)
        self.sigma_ra, self.sigma_dec = _f64(sigma_ra), _f64(sigma_dec)
        self.obs_helio_equ = None if obs_helio_equ is None else _f64(obs_helio_equ)
        # device-side observer positions (row n2): Earth-fixed site vectors + UT1 epochs instead of the two cached vectors
        self.obs_body_fixed = None if obs_body_fixed is None else _f64(obs_body_fixed)
        self.mjd_ut1 = None if mjd_ut1 is None else _f64(mjd_ut1)
        self.obs_geo_ecl = None if obs_geo_ecl is None else _f64(obs_geo_ecl)
        self.noise_z = None if noise_z is None else _f64(noise_z)
        # device-side draws (device/noise.hpp): noise_seed set => OUTFIT_NOISE_DEVICE, keyed per trajectory
        self.noise_seed = noise_seed
        self.traj_key = None if traj_key is None else np.ascontiguousarray(traj_key, dtype=np.uint64)
        self.traj_key_base = int(traj_key_base)
        self.n_traj = len(self.traj_offset) - 1
        self.traj_ids = list(range(self.n_traj)) if traj_ids is None else list(traj_ids)
        self.validate()

    def validate(self, params=None):
        """Every array against the offset table — the library indexes the columns with the offsets themselves, so a
        table that is monotone but runs past them would be an out-of-bounds read on the other side of the C ABI (the
        reference materialises each trajectory from data it owns, obs_dataset_api.rs:210-226, and cannot overrun).
        Raises InvalidBatch (a ValueError). With `params`, also the length of the host-drawn normals."""
        if self.traj_offset.ndim != 1 or self.n_traj < 0:
            raise InvalidBatch("traj_offset needs n_traj + 1 entries (at least one)")
        off = self.traj_offset
        if self.n_traj and bool((off[1:] < off[:-1]).any()):
            raise InvalidBatch("traj_offset must be non-decreasing")
        end = int(off[-1])
        for k in ("mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec", "mjd_ut1"):
            a = getattr(self, k)
            if a is not None and a.size < end:
                raise InvalidBatch(f"traj_offset[-1] = {end} runs past `{k}` ({a.size} rows)")
            if a is None and k != "mjd_ut1" and end > int(off[0]):
                raise InvalidBatch(f"`{k}` is missing")
        for k in ("obs_helio_equ", "obs_geo_ecl", "obs_body_fixed"):
            a = getattr(self, k)
            if a is not None and a.size < 3 * end:
                raise InvalidBatch(f"traj_offset[-1] = {end} runs past `{k}` ({a.size} doubles, 3 per observation are needed)")
        if self.traj_key is not None and self.traj_key.size < self.n_traj:
            raise InvalidBatch(f"`traj_key` has {self.traj_key.size} entries for {self.n_traj} trajectories")
        if len(self.traj_ids) != self.n_traj:
            raise InvalidBatch(f"`traj_ids` has {len(self.traj_ids)} entries for {self.n_traj} trajectories")
        if params is not None and self.noise_z is not None and self.noise_seed is None and params.n_noise_realizations > 0:
            need = self.n_traj * int(params.max_triplets) * int(params.n_noise_realizations) * 6
            if self.noise_z.size < need:
                raise InvalidBatch(f"`noise_z` has {self.noise_z.size} normals; n_traj x max_triplets x "
                                   f"n_noise_realizations x 6 = {need} are needed")

    def capacities(self):
        """(n_obs_capacity, noise_z_capacity) for outfit_hip_call_options_t: what the arrays handed over really hold."""
        cols = [getattr(self, k).size for k in ("mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec", "mjd_ut1")
                if getattr(self, k) is not None]
        cols += [getattr(self, k).size // 3 for k in ("obs_helio_equ", "obs_geo_ecl", "obs_body_fixed")
                 if getattr(self, k) is not None]
        return (min(cols) if cols else 0), (0 if self.noise_z is None else int(self.noise_z.size))

    @classmethod
    def from_dict(cls, d, with_geo=True, device_geometry=False):
        """`device_geometry=True` hands over the sites (obs_body_fixed + mjd_ut1) instead of observer positions:
        the library then runs the observer pre-pass itself (row n2)."""
        if device_geometry:
            return cls(d["traj_offset"], d["mjd_tt"], d["ra"], d["dec"], d["sigma_ra"], d["sigma_dec"], None, None,
                       d.get("noise_z"), obs_body_fixed=d["obs_body_fixed"], mjd_ut1=d["mjd_ut1"])
        return cls(d["traj_offset"], d["mjd_tt"], d["ra"], d["dec"], d["sigma_ra"], d["sigma_dec"],
                   d["obs_helio_equ"], d.get("obs_geo_ecl") if with_geo else None, d.get("noise_z"))

    def slice(self, t0, t1):
        """Trajectories [t0, t1) as an independent batch (used to shard over ranks)."""
        o0, o1 = int(self.traj_offset[t0]), int(self.traj_offset[t1])
        geo = None if self.obs_geo_ecl is None else self.obs_geo_ecl[3 * o0:3 * o1]
        return ObsBatch(self.traj_offset[t0:t1 + 1] - np.uint64(o0), self.mjd_tt[o0:o1], self.ra[o0:o1],
                        self.dec[o0:o1], self.sigma_ra[o0:o1], self.sigma_dec[o0:o1],
                        None if self.obs_helio_equ is None else self.obs_helio_equ[3 * o0:3 * o1], geo, None,
                        self.traj_ids[t0:t1],
                        obs_body_fixed=None if self.obs_body_fixed is None else self.obs_body_fixed[3 * o0:3 * o1],
                        mjd_ut1=None if self.mjd_ut1 is None else self.mjd_ut1[o0:o1],
                        noise_seed=self.noise_seed,
                        traj_key=None if self.traj_key is None else self.traj_key[t0:t1],
                        traj_key_base=self.traj_key_base + (0 if self.traj_key is not None else t0))

    def as_c(self):
        b = ObsBatchC()
        b.n_traj = self.n_traj
        b.traj_offset = self.traj_offset.ctypes.data_as(C.POINTER(C.c_uint64))
        for k in ("mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec", "obs_helio_equ", "obs_body_fixed", "mjd_ut1"):
            if getattr(self, k) is not None:
                setattr(b, k, getattr(self, k).ctypes.data_as(_dp))
        if self.obs_geo_ecl is not None:
            b.obs_geo_ecl = self.obs_geo_ecl.ctypes.data_as(_dp)
        if self.noise_z is not None:
            b.noise_z = self.noise_z.ctypes.data_as(_dp)
        if self.noise_seed is not None:
            b.noise_source = _abi.NOISE_DEVICE
            b.noise_seed = int(self.noise_seed) & 0xFFFFFFFFFFFFFFFF
        if self.traj_key is not None:
            b.traj_key = self.traj_key.ctypes.data_as(C.POINTER(C.c_uint64))
        b.traj_key_base = self.traj_key_base
        b._owner = self  # as above (a temporary `batch.slice(..).as_c()` must not dangle)
        return b


# ----------------------------------------------------------------------------- results
@dataclass
class OrbitalElements:
    """`OrbitalElements::{Keplerian, Cometary}` (orbit_type/mod.rs:221-237)."""
    kind: str            # "Keplerian" (a,e,i,Omega,omega,M) | "Cometary" (q,e,i,Omega,omega,nu)
    reference_epoch: float
    elements: tuple


@dataclass
class GaussResult:
    """`GaussResult::{PrelimOrbit, CorrectedOrbit}` (gauss_result.rs:99-102)."""
    stage: str           # "PrelimOrbit" | "CorrectedOrbit"
    orbit: OrbitalElements

    def get_orbit(self):
        return self.orbit


@dataclass
class IODGauss:
    """`FitOrbitResult::IODGauss((GaussResult, IODRMS))` (constants.rs:134-137)."""
    gauss_result: GaussResult
    rms: float
    best_triplet: int = 0
    best_realization: int = 0
    n_triplets: int = 0
    attempts: int = 0

    def orbital_elements(self):
        return self.gauss_result.orbit

    def orbit_quality(self):
        return self.rms


def record_to_result(rec, batch=None, traj_index=None, params=None):
    """One `outfit_gauss_result_t` -> FitOrbitResult or an OutfitError instance (not raised)."""
    st = int(rec["status"])
    if st == _abi.TRAJ_OK:
        kind = "Keplerian" if int(rec["kind"]) == _abi.KIND_KEPLERIAN else "Cometary"
        stage = "CorrectedOrbit" if int(rec["stage"]) == _abi.STAGE_CORRECTED else "PrelimOrbit"
        orb = OrbitalElements(kind, float(rec["epoch"]), tuple(float(x) for x in rec["elem"]))
        return IODGauss(GaussResult(stage, orb), float(rec["rms"]), int(rec["best_triplet"]),
                        int(rec["best_realization"]), int(rec["n_triplets"]), int(rec["attempts"]))
    if st == _abi.TRAJ_NO_FEASIBLE_TRIPLETS:
        span, n_obs = 0.0, 0
        if batch is not None and traj_index is not None:
            o0, o1 = int(batch.traj_offset[traj_index]), int(batch.traj_offset[traj_index + 1])
            n_obs = o1 - o0
            span = float(batch.mjd_tt[o1 - 1] - batch.mjd_tt[o0]) if n_obs else 0.0
        return NoFeasibleTriplets(span, n_obs, params.dt_min if params else float("nan"),
                                  params.dt_max_triplet if params else float("nan"))
    if st == _abi.TRAJ_NO_VIABLE_ORBIT:
        return NoViableOrbit(_CAUSES.get(int(rec["cause"]), "Unknown"), int(rec["attempts"]))
    if st == _abi.TRAJ_INVALID_CONVERSION:
        return InvalidConversion("Parabolic orbit cannot be represented with finite a")
    if st == _abi.TRAJ_INVALID_ORBIT:
        return InvalidOrbit("eccentricity must be > 1 for hyperbolic orbits")
    return OutfitError(f"unknown status {st}")


# ----------------------------------------------------------------------------- math mode
MATH_DEFAULT, MATH_CORRECTLY_ROUNDED, MATH_GLIBC = -1, 0, 1


def _options(math_mode, batch=None, params=None):
    """outfit_hip_call_options_t for one call: the math mode (None = the process-wide default) and, with `batch`, the
    lengths of its arrays — checked here (InvalidBatch) and stated to the library, which checks them again."""
    if math_mode is None and batch is None:
        return None
    o = CallOptionsC()
    lib().outfit_hip_call_options_default(C.byref(o))
    if math_mode is not None:
        o.math_mode = int(math_mode)
    if batch is not None:
        batch.validate(params)          # attributes may have been replaced since __init__ (e.g. noise_z of a slice)
        o.n_obs_capacity, o.noise_z_capacity = batch.capacities()
    return C.byref(o)


def set_math_mode(mode):
    """The process-wide DEFAULT of OUTFIT_MATH_* (include/outfit_hip.h): which bits sin/cos/sinh/cosh return in the
    functions that feed the correction loop. Calls and sessions that name their mode (`math_mode=`) never read it;
    sessions keep the mode they were created with."""
    rc = lib().outfit_hip_set_math_mode(int(mode))
    if rc != 0:
        _raise_rc(rc)


def get_math_mode():
    return lib().outfit_hip_get_math_mode()


class math_mode:
    """Context manager: `with math_mode(MATH_GLIBC): ...`"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_math_mode()
        set_math_mode(self.mode)

    def __exit__(self, *exc):
        set_math_mode(self.prev)


# ----------------------------------------------------------------------------- entry points
def fit_full_iod_records(batch, ephem, params, n_devices=0, math_mode=None):
    """The blocking C-ABI call; returns the raw records (numpy structured array). `math_mode`: MATH_* for this
    call (outfit_hip_call_options_t), None = the process-wide default."""
    L = lib()
    pc, bc = params.as_c(), batch.as_c()
    ec = ephem.as_c() if ephem is not None else None
    out = np.empty(max(batch.n_traj, 1), dtype=RESULT_DTYPE)   # the records land in the array that is returned
    rc = L.outfit_hip_fit_full_iod_ex(C.byref(pc), C.byref(bc), C.byref(ec) if ec is not None else None,
                                      n_devices, _options(math_mode, batch, params),
                                      out.ctypes.data_as(C.POINTER(GaussResultC)))
    if rc != 0:
        _raise_rc(rc)
    return out[:batch.n_traj]


def fit_full_iod(batch, ephem, params, n_devices=0, math_mode=None):
    """`FitIOD::fit_full_iod` after `prepare_iod`: {traj_id: FitOrbitResult | OutfitError}
    (`FullOrbitResult`, constants.rs:195). Per-trajectory failures are values, not raises."""
    recs = fit_full_iod_records(batch, ephem, params, n_devices, math_mode)
    return {tid: record_to_result(recs[i], batch, i, params) for i, tid in enumerate(batch.traj_ids)}


def fit_iod(batch, traj_id, ephem, params):
    """`FitIOD::fit_iod` for a single trajectory (obs_dataset_api.rs:130-143); raises on error."""
    i = batch.traj_ids.index(traj_id)
    sub = batch.slice(i, i + 1)
    if batch.noise_z is not None:
        stride = params.max_triplets * params.n_noise_realizations * 6
        sub.noise_z = batch.noise_z[i * stride:(i + 1) * stride]
    res = record_to_result(fit_full_iod_records(sub, ephem, params, 1)[0], sub, 0, params)
    if isinstance(res, OutfitError):
        raise res
    return res


class IODSession:
    """Inputs resident in HBM; `run()` enqueues the kernels of the whole path on a stream."""

    def __init__(self, batch, ephem, params, device=0, math_mode=None):
        self._L = lib()
        self.n_traj = batch.n_traj
        self.n_obs = int(batch.traj_offset[-1]) if batch.n_traj else 0
        self._keep = (batch, ephem)
        pc, bc = params.as_c(), batch.as_c()
        ec = ephem.as_c() if ephem is not None else None
        h = C.c_void_p()
        rc = self._L.outfit_hip_session_create_ex(C.byref(pc), C.byref(bc),
                                                  C.byref(ec) if ec is not None else None, device,
                                                  _options(math_mode, batch, params), C.byref(h))
        if rc != 0:
            _raise_rc(rc)
        self._h = h

    def run(self, stream=None, d_out=None):
        """Enqueue one pass. `stream`: a HIP stream handle, e.g. `torch.cuda.Stream().cuda_stream`; None or 0 (which is
        what torch's DEFAULT stream reports) means the session's own non-blocking stream — work queued on torch's
        default stream is then not ordered after the run (use `results()` or a device-wide synchronize)."""
        rc = self._L.outfit_hip_session_run(self._h, C.c_void_p(stream or 0), C.c_void_p(d_out or 0))
        if rc != 0:
            _raise_rc(rc)

    def results(self):
        out = (GaussResultC * max(self.n_traj, 1))()
        rc = self._L.outfit_hip_session_results(self._h, out)
        if rc != 0:
            _raise_rc(rc)
        return np.frombuffer(out, dtype=RESULT_DTYPE)[:self.n_traj].copy()

    def candidate_scores(self):
        """Diagnostic (outfit_hip_session_candidate_scores): (sum, n_win, flags) K2c left for the last chunk of the
        last run, one entry per candidate slot."""
        n = C.c_uint64(0)
        self._L.outfit_hip_session_candidate_scores(self._h, 0, None, None, None, C.byref(n))
        total = int(n.value)
        total_sum, n_win, flags = np.empty(total), np.empty(total, np.uint32), np.empty(total, np.uint32)
        if total:
            rc = self._L.outfit_hip_session_candidate_scores(
                self._h, total, total_sum.ctypes.data_as(C.POINTER(C.c_double)),
                n_win.ctypes.data_as(C.POINTER(C.c_uint32)), flags.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n))
            if rc != 0:
                _raise_rc(rc)
        return total_sum, n_win, flags

    def check(self):
        """Synchronise and raise what `results()` would (EphemerisRange) without fetching anything."""
        rc = self._L.outfit_hip_session_check(self._h)
        if rc != 0:
            _raise_rc(rc)

    def set_profiling(self, on):
        self._L.outfit_hip_session_set_profiling(self._h, 1 if on else 0)

    def kernel_times(self):
        """{kernel: (total ms, launches)} since set_profiling(True)."""
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        launches = (C.c_int * 16)()
        n = self._L.outfit_hip_session_kernel_times(self._h, 16, names, ms, launches)
        if n < 0:
            _raise_rc(n)
        return {names[i].decode(): (float(ms[i]), int(launches[i])) for i in range(n)}

    def close(self):
        if self._h:
            self._L.outfit_hip_session_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------- differential correction
_DC_FIELDS = [n for n, _ in DiffCorConfigC._fields_ if not n.startswith("_")]
_DC_ERRORS = {_abi.DC_FAILED: "DifferentialCorrectionFailed", _abi.DC_BIZARRE: "BizarreOrbit",
              _abi.DC_DIVERGED: "DifferentialCorrectionDiverged"}


class DifferentialCorrectionConfig:
    """`DifferentialCorrectionConfig` (differential_orbit_correction/diff_cor.rs:39-178) with
    `OutlierRejectionConfig` and `EquinoctialLimits` flattened; defaults = `::default()`."""

    def __init__(self, **kw):
        c = DiffCorConfigC()
        lib().outfit_hip_diffcor_config_default(C.byref(c))
        for n in _DC_FIELDS:
            setattr(self, n, list(getattr(c, n)) if n == "free_elements" else getattr(c, n))
        for k, v in kw.items():
            if k not in _DC_FIELDS:
                raise TypeError(f"unknown DifferentialCorrectionConfig field {k!r}")
            setattr(self, k, v)

    def as_c(self):
        c = DiffCorConfigC()
        for n in _DC_FIELDS:
            if n == "free_elements":
                for j in range(6):
                    c.free_elements[j] = 1 if self.free_elements[j] else 0
            else:
                setattr(c, n, int(getattr(self, n)) if n == "enable_outlier_rejection" else getattr(self, n))
        return c


@dataclass
class EquinoctialElements:
    """`OrbitalElements::Equinoctial { elements, uncertainty, covariance }` (orbit_type/mod.rs)."""
    reference_epoch: float
    elements: tuple          # (a, h, k, p, q, lambda)
    covariance: np.ndarray   # 6x6, rescaled (least_square.rs:341-362)

    @property
    def uncertainty(self):
        """EquinoctialUncertainty::from_covariance (orbit_type/uncertainty.rs:261-271)."""
        return tuple(float(np.sqrt(self.covariance[i, i])) for i in range(6))

    def to_keplerian(self):
        """`KeplerianElements::from(EquinoctialElements)` (orbit_type/keplerian_element.rs:185-233; known answer
        equinoctial_element.rs:1240-1264): (a, e, i, Omega, omega, M), angles by `principal_angle`."""
        import math
        a, h, k, p, q, lam = (float(x) for x in self.elements)
        two_pi = 6.283185307179586
        pa = lambda x: (lambda r: r + two_pi if r < 0.0 else r)(math.fmod(x, two_pi))  # noqa: E731  f64::rem_euclid
        ecc = math.sqrt(h * h + k * k)
        dig = 0.0 if ecc < 1.0e-12 else math.atan2(h, k)
        tgi2 = math.sqrt(p * p + q * q)
        node = 0.0 if tgi2 < 1.0e-12 else math.atan2(p, q)
        return (a, ecc, 2.0 * math.atan(tgi2), node, pa(dig - node), pa(lam - dig))


@dataclass
class DifferentialCorrection:
    """`FitOrbitResult::DifferentialCorrection((OrbitalElements, Chi2))` (constants.rs:136)."""
    orbit: EquinoctialElements
    normalised_rms: float
    total_newton_iterations: int = 0
    num_measurements: int = 0
    n_rejected: int = 0

    def orbital_elements(self):
        return self.orbit

    def orbit_quality(self):
        return self.normalised_rms


def lsq_record_to_result(rec, batch=None, traj_index=None, params=None):
    """One `outfit_lsq_result_t` -> FitOrbitResult (DifferentialCorrection, or the IOD orbit the reference
    falls back to, mod.rs:109) or an OutfitError instance (not raised)."""
    st = int(rec["status"])
    if st == _abi.LSQ_DIFFCOR:
        orb = EquinoctialElements(float(rec["epoch"]), tuple(float(x) for x in rec["elem"]),
                                  np.array(rec["covariance"], dtype=np.float64).reshape(6, 6))
        return DifferentialCorrection(orb, float(rec["normalised_rms"]), int(rec["total_newton_iterations"]),
                                      int(rec["num_measurements"]), int(rec["n_rejected"]))
    if st == _abi.LSQ_IOD_KEPT or st == _abi.LSQ_IOD_ERROR:
        return record_to_result(rec["iod"], batch, traj_index, params)
    if st == _abi.LSQ_CONVERSION_ERROR:
        if int(rec["conv_error"]) == _abi.TRAJ_INVALID_CONVERSION:
            return InvalidConversion("Parabolic orbit cannot be represented with finite a")
        return InvalidOrbit("eccentricity must be > 1 for hyperbolic orbits")
    return OutfitError(f"unknown status {st}")


def _initial_c(initial, n):
    if initial is None:
        return None
    a = np.ascontiguousarray(initial, dtype=RESULT_DTYPE)
    if a.shape != (n,):
        raise InvalidBatch("initial_orbits must hold one record per trajectory")
    return (GaussResultC * max(n, 1)).from_buffer_copy(a.tobytes() if n else bytes(96))


def fit_lsq_records(batch, ephem, iod_params, diff_cor_config, initial_orbits=None, n_devices=0, math_mode=None):
    """The blocking C-ABI call `outfit_hip_fit_lsq_ex`; returns (records, per-observation fit data)."""
    L = lib()
    pc, cc, bc = iod_params.as_c(), diff_cor_config.as_c(), batch.as_c()
    ec = ephem.as_c() if ephem is not None else None
    n, n_obs = batch.n_traj, int(batch.traj_offset[-1]) if batch.n_traj else 0
    out = (LsqResultC * max(n, 1))()
    fit = (ObsFitC * max(n_obs, 1))()
    rc = L.outfit_hip_fit_lsq_ex(C.byref(pc), C.byref(cc), C.byref(bc), C.byref(ec) if ec is not None else None,
                                 _initial_c(initial_orbits, n), n_devices, _options(math_mode, batch, iod_params), out, fit)
    if rc != 0:
        _raise_rc(rc)
    return (np.frombuffer(out, dtype=LSQ_DTYPE)[:n].copy(), np.frombuffer(fit, dtype=OBS_FIT_DTYPE)[:n_obs].copy())


def fit_lsq(batch, ephem, iod_params, diff_cor_config, initial_orbits=None, n_devices=0):
    """`FitLSQ::fit_lsq` after the dataset preparation (obs_dataset_api.rs:57-99):
    {traj_id: FitOrbitResult | OutfitError}. `initial_orbits`: records of a previous `fit_full_iod_records`."""
    recs, _ = fit_lsq_records(batch, ephem, iod_params, diff_cor_config, initial_orbits, n_devices)
    return {tid: lsq_record_to_result(recs[i], batch, i, iod_params) for i, tid in enumerate(batch.traj_ids)}


def _session_run_lsq(self, diff_cor_config, initial_orbits=None, stream=None):
    """Enqueue the corrector on the session's resident observations (seeds: the session's IOD results,
    overridden by `initial_orbits` records with status 0)."""
    cc = diff_cor_config.as_c()
    rc = self._L.outfit_hip_session_run_lsq(self._h, C.byref(cc), _initial_c(initial_orbits, self.n_traj),
                                            C.c_void_p(stream or 0))
    if rc != 0:
        _raise_rc(rc)


def _session_lsq_results(self):
    out = (LsqResultC * max(self.n_traj, 1))()
    fit = (ObsFitC * max(self.n_obs, 1))()
    rc = self._L.outfit_hip_session_lsq_results(self._h, out, fit)
    if rc != 0:
        _raise_rc(rc)
    return (np.frombuffer(out, dtype=LSQ_DTYPE)[:self.n_traj].copy(),
            np.frombuffer(fit, dtype=OBS_FIT_DTYPE)[:self.n_obs].copy())


IODSession.run_lsq = _session_run_lsq
IODSession.lsq_results = _session_lsq_results


def observer_positions(ephem, mjd_tt, mjd_ut1, obs_body_fixed):
    """`outfit_hip_observer_positions`: what `OutfitCache::build` stores per observation (Observer::pvobs and
    helio_position, observer_extension.rs:180-237): (geocentric ecliptic J2000 [n,3], heliocentric equatorial J2000 [n,3])."""
    tt, ut1, fx = _f64(mjd_tt), _f64(mjd_ut1), _f64(obs_body_fixed).reshape(-1)
    n = tt.shape[0]
    geo, hel = np.empty(3 * n), np.empty(3 * n)
    ec = ephem.as_c()
    rc = lib().outfit_hip_observer_positions(C.byref(ec), n, tt.ctypes.data_as(_dp), ut1.ctypes.data_as(_dp),
                                             fx.ctypes.data_as(_dp), geo.ctypes.data_as(_dp), hel.ctypes.data_as(_dp))
    if rc != 0:
        _raise_rc(rc)
    return geo.reshape(n, 3), hel.reshape(n, 3)


def device_normals(noise_seed, n_traj, max_triplets, n_noise, traj_key=None, traj_key_base=0):
    """`outfit_hip_device_normals`: the standard normals the device generator (Philox4x32-10 + Box-Muller,
    device/noise.hpp) yields, in the layout of `ObsBatch.noise_z`."""
    out = np.empty(int(n_traj) * int(max_triplets) * int(n_noise) * 6, dtype=np.float64)
    keys = None if traj_key is None else np.ascontiguousarray(traj_key, dtype=np.uint64)
    rc = lib().outfit_hip_device_normals(int(noise_seed) & 0xFFFFFFFFFFFFFFFF,
                                         None if keys is None else keys.ctypes.data_as(C.POINTER(C.c_uint64)),
                                         int(traj_key_base), int(n_traj), int(max_triplets), int(n_noise),
                                         out.ctypes.data_as(_dp))
    if rc != 0:
        _raise_rc(rc)
    return out


class SolverType:
    """`SolverType { kind, params: SolverParams }` (kepler/params.rs:24-73): kind 'NewtonRaphson' | 'BrentDecker' |
    'Auto'; convergency; psi_guess (None = cold start); max_iter_prelim_kepuni; parabolic_solving_method 'Cardano' | 'Newton'."""
    KINDS = {"NewtonRaphson": 0, "BrentDecker": 1, "Auto": 2}
    PARABOLIC = {"Cardano": 0, "Newton": 1}

    def __init__(self, kind="NewtonRaphson", convergency=100 * 2.220446049250313e-16, psi_guess=None,
                 max_iter_prelim_kepuni=20, parabolic_solving_method="Cardano"):
        self.kind, self.convergency, self.psi_guess = kind, convergency, psi_guess
        self.max_iter_prelim_kepuni, self.parabolic_solving_method = max_iter_prelim_kepuni, parabolic_solving_method

    def as_c(self):
        c = _abi.SolverTypeC()
        c.kind = self.KINDS[self.kind] if isinstance(self.kind, str) else int(self.kind)
        pm = self.parabolic_solving_method
        c.parabolic_solving_method = self.PARABOLIC[pm] if isinstance(pm, str) else int(pm)
        c.convergency = self.convergency
        c.max_iter_prelim_kepuni = self.max_iter_prelim_kepuni
        c.has_psi_guess = 0 if self.psi_guess is None else 1
        c.psi_guess = 0.0 if self.psi_guess is None else self.psi_guess
        return c


def propagate_universal(r, v, t0, t1, convergency=100 * 2.220446049250313e-16, solver=None):
    """Batched `kepler::propagate_universal` (kepler/propagation.rs:114-174). `solver`: a SolverType (default:
    NewtonRaphson with `convergency`). Returns (out[n,11] = r1,v1,f,g,fdot,gdot,psi ; status[n]); status 0 ok,
    1 / 3 DegenerateState, 2 NewtonRaphsonKeplerConvergence, 4 BrentDekkerKeplerConvergence."""
    if solver is not None:
        r, v = _f64(r).reshape(-1, 3), _f64(v).reshape(-1, 3)
        n = r.shape[0]
        rv = np.ascontiguousarray(np.concatenate([r, v], axis=1))
        t0a, t1a = _f64(np.broadcast_to(t0, (n,))), _f64(np.broadcast_to(t1, (n,)))
        out = np.empty((n, 11))
        st = np.empty(n, dtype=np.int32)
        sc = solver.as_c()
        rc = lib().outfit_hip_propagate_universal_ex(n, rv.ctypes.data_as(_dp), t0a.ctypes.data_as(_dp),
                                                     t1a.ctypes.data_as(_dp), C.byref(sc), out.ctypes.data_as(_dp),
                                                     st.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc != 0:
            _raise_rc(rc)
        return out, st
    r, v = _f64(r).reshape(-1, 3), _f64(v).reshape(-1, 3)
    n = r.shape[0]
    rv = np.ascontiguousarray(np.concatenate([r, v], axis=1))
    t0a, t1a = _f64(np.broadcast_to(t0, (n,))), _f64(np.broadcast_to(t1, (n,)))
    out = np.empty((n, 11))
    st = np.empty(n, dtype=np.int32)
    rc = lib().outfit_hip_propagate_universal(n, rv.ctypes.data_as(_dp), t0a.ctypes.data_as(_dp),
                                              t1a.ctypes.data_as(_dp), convergency, out.ctypes.data_as(_dp),
                                              st.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        _raise_rc(rc)
    return out, st
