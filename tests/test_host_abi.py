"""The C-ABI shared library without a GPU: it builds, loads, exports every symbol that
include/outfit_hip.h declares, validates parameters like the reference's builder, and
fails LOUDLY (no CPU fallback) when no HIP device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
import outfit_amd
from outfit_amd import _abi, _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    outfit_amd.build_native()
    return _lib.lib()


def test_library_exports_every_declared_symbol(L):
    header = open(os.path.join(ROOT, "include", "outfit_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(outfit_hip_\w+)\s*\(", header))
    assert declared, "no declarations found in the header"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in outfit_hip.h but not exported"
    assert L.outfit_hip_abi_version() == _abi.ABI_VERSION


def test_struct_layouts_match_the_header():
    assert C.sizeof(_abi.GaussResultC) == 96
    assert C.sizeof(_abi.IODParamsC) == 176
    assert _abi.GaussResultC.epoch.offset == 32 and _abi.GaussResultC.rms.offset == 88
    assert np.dtype(_abi.RESULT_DTYPE).itemsize == 96


def test_ctypes_mirror_matches_the_compiled_header(tmp_path):
    """sizeof / offsetof of every struct of include/outfit_hip.h as the C compiler lays them out, against the
    ctypes mirror the tests and bench.py go through."""
    import subprocess
    src = tmp_path / "layout.c"
    probes = [("outfit_iod_params_t", "IODParamsC", ["max_triplets", "gap_max", "aberth_eps", "root_imag_eps"]),
              ("outfit_obs_batch_t", "ObsBatchC", ["noise_z", "noise_source", "noise_seed", "traj_key", "traj_key_base"]),
              ("outfit_ephem_t", "EphemC", ["ncoef", "nsub", "n_blocks", "coef"]),
              ("outfit_gauss_result_t", "GaussResultC", ["attempts", "epoch", "elem", "rms"]),
              ("outfit_diffcor_config_t", "DiffCorConfigC", ["max_stagnation_iterations", "propagator",
                                                             "eccentricity_limit", "free_elements"]),
              ("outfit_lsq_result_t", "LsqResultC", ["total_newton_iterations", "epoch", "covariance", "iod"]),
              ("outfit_obs_fit_t", "ObsFitC", ["chi", "selection"]),
              ("outfit_de_header_t", "DeHeaderC", ["ipt", "numde", "ncon", "recsize", "n_blocks"])]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/outfit_hip.h"', "int main(void) {"]
    for cname, _, fields in probes:
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for f in fields:
            lines.append(f'  printf(" %zu", offsetof({cname}, {f}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for line, (cname, pyname, fields) in zip(out, probes):
        got = [int(x) for x in line.split()[1:]]
        cls = getattr(_abi, pyname)
        want = [C.sizeof(cls)] + [getattr(cls, f).offset for f in fields]
        assert got == want, (cname, got, want)


def test_params_default_matches_reference_defaults(L):
    """IODParams::default() (mod.rs:308-344), and product == oracle restatement."""
    p = outfit_amd.IODParams()
    q = O.default_params()
    for name, _ in _abi.IODParamsC._fields_:
        if not name.startswith("_"):
            assert getattr(p, name) == getattr(q, name), name
    assert (p.n_noise_realizations, p.noise_scale, p.extf, p.dtmax) == (20, 1.0, -1.0, 30.0)
    assert (p.dt_min, p.dt_max_triplet, p.optimal_interval_time) == (0.03, 150.0, 20.0)
    assert (p.max_obs_for_triplets, p.max_triplets, p.gap_max) == (100, 10, 8.0 / 24.0)
    assert (p.max_ecc, p.max_perihelion_au, p.min_rho2_au) == (5.0, 1e3, 0.01)
    assert (p.aberth_max_iter, p.aberth_eps, p.kepler_eps) == (50, 1e-6, 1e3 * 2.220446049250313e-16)
    assert (p.max_tested_solutions, p.r2_min_au, p.r2_max_au) == (3, 0.05, 200.0)
    assert (p.newton_eps, p.newton_max_it, p.root_imag_eps) == (1e-10, 50, 1e-6)


@pytest.mark.parametrize("field,bad", [
    ("noise_scale", -1.0), ("dt_min", -0.1), ("dt_max_triplet", float("nan")), ("dtmax", -1.0),
    ("max_ecc", -0.5), ("root_imag_eps", -1e-9), ("max_perihelion_au", 0.0), ("min_rho2_au", 0.0),
    ("aberth_eps", 0.0), ("kepler_eps", float("nan")), ("newton_eps", 0.0), ("newton_max_it", 0),
    ("aberth_max_iter", 0), ("max_tested_solutions", 0), ("r2_min_au", 0.0), ("r2_max_au", 0.01),
])
def test_builder_rejects_what_the_reference_rejects(field, bad):
    """IODParamsBuilder::build (mod.rs:544-624)."""
    b = outfit_amd.IODParams.builder()
    getattr(b, field)(bad)
    with pytest.raises(outfit_amd.InvalidIODParameter):
        b.build()


def test_builder_accepts_valid_and_infinite_bounds():
    p = (outfit_amd.IODParams.builder().max_triplets(100).n_noise_realizations(0)
         .max_perihelion_au(float("inf")).r2_min_au(0.5).r2_max_au(0.5).build())
    assert p.max_triplets == 100 and p.r2_min_au == p.r2_max_au


def test_malformed_batches_are_rejected_before_touching_the_gpu(L):
    ds = synth.make_dataset(3)
    params = synth.bench_params(outfit_amd.IODParams())
    good = outfit_amd.ObsBatch.from_dict(ds)
    bad = outfit_amd.ObsBatch.from_dict(ds)
    bad.traj_offset = np.array([0, 40, 20, 60], dtype=np.uint64)
    with pytest.raises(outfit_amd.InvalidBatch):
        outfit_amd.fit_full_iod_records(bad, None, params)
    with pytest.raises(outfit_amd.InvalidBatch):  # noise requested, no normals supplied
        outfit_amd.fit_full_iod_records(good, None, outfit_amd.IODParams())
    with pytest.raises(outfit_amd.InvalidIODParameter):
        outfit_amd.fit_full_iod_records(good, None, outfit_amd.IODParams(n_noise_realizations=0, newton_max_it=0))


def test_no_cpu_fallback(L):
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    if L.outfit_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    ds = synth.make_dataset(2)
    with pytest.raises(outfit_amd.NativeLibraryError, match="no CPU fallback"):
        outfit_amd.fit_full_iod_records(outfit_amd.ObsBatch.from_dict(ds), None,
                                        synth.bench_params(outfit_amd.IODParams()))
    with pytest.raises(outfit_amd.NativeLibraryError):
        outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), None, synth.bench_params(outfit_amd.IODParams()))
    with pytest.raises(outfit_amd.NativeLibraryError):
        outfit_amd.propagate_universal([[1.0, 0, 0]], [[0, 0.017, 0]], 0.0, 1.0)


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under outfit_amd/ may import, link or call it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "outfit_amd")):
        if "_build" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in text and "liboutfit_oracle" not in text \
                    and not re.search(r"\borc_\w+\(", text), \
                    f"{os.path.join(dirpath, f)} refers to the oracle"
                # ... nor keep a hook through which a caller could put something else in the kernels' place
                # (outfit_amd/dist.py had `compute=` for the CPU tests until round 5: that stand-in lives in tests/ now)
                if f.endswith(".py"):
                    assert "compute=" not in text and not re.search(r"def \w+\([^)]*\bcompute\b", text), \
                        f"{os.path.join(dirpath, f)} has a compute-injection hook"


def test_record_to_result_mirrors_reference_variants():
    rec = np.zeros(1, dtype=_abi.RESULT_DTYPE)[0]
    rec["status"], rec["stage"], rec["kind"], rec["rms"] = 0, 1, 0, 0.5
    r = outfit_amd.api.record_to_result(rec)
    assert isinstance(r, outfit_amd.IODGauss) and r.gauss_result.stage == "CorrectedOrbit"
    assert r.orbital_elements().kind == "Keplerian" and r.orbit_quality() == 0.5
    rec["status"], rec["cause"], rec["attempts"] = 2, 1, 100
    e = outfit_amd.api.record_to_result(rec)
    assert isinstance(e, outfit_amd.NoViableOrbit) and e.cause == "GaussNoRootsFound" and e.attempts == 100
    rec["status"] = 1
    assert isinstance(outfit_amd.api.record_to_result(rec), outfit_amd.NoFeasibleTriplets)
    rec["status"] = 3
    assert isinstance(outfit_amd.api.record_to_result(rec), outfit_amd.InvalidConversion)


def test_no_kernel_needs_a_dynamically_sized_stack():
    """The private segment of every kernel must be sized at compile time: hipcc's resource report (kept by the
    Makefile next to the objects) has to say `Dynamic Stack: False` for each of them. A kernel that calls through a
    function pointer or recurses gets `Dynamic Stack: True`, and the HIP runtime then sizes its scratch from
    hipLimitStackSize (1 KiB per lane by default) whatever the callee needs — an overrun is a
    HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION ("scratch")."""
    import glob
    import re
    build = os.path.join(ROOT, "outfit_amd", "csrc", "_build")
    reports = glob.glob(os.path.join(build, "*.resource.txt"))
    if not reports:
        pytest.skip("library not built by the Makefile of this tree")
    text = "".join(open(r).read() for r in reports)
    names = re.findall(r"Function Name: (\S+)", text)
    dyn = re.findall(r"Dynamic Stack: (\w+)", text)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", text)]
    assert len(names) >= 20 and len(names) == len(dyn) == len(scratch)
    assert all(d == "False" for d in dyn), [n for n, d in zip(names, dyn) if d != "False"]
    assert max(scratch) <= 1024, dict(zip(names, scratch))   # today: 448 B (k_gauss_correct), spills only


def test_equinoctial_to_keplerian_known_answer():
    """orbit_type/equinoctial_element.rs:1240-1264 (`test_equinoctial_conversion`, assert_eq!): the host-side conversion
    of a `fit_lsq` result to Keplerian elements reproduces the reference's values bit for bit (on a glibc host; the
    functions involved — sqrt, atan2, atan — are evaluated by the platform libm on both sides)."""
    import numpy as np
    equ = outfit_amd.EquinoctialElements(0.0, (1.8017360713, 0.2693736809404963, 0.08856415260522467,
                                               0.0008089970142830734, 0.10168201110394352, 1.693697008), np.eye(6))
    want = (1.8017360713, 0.2835591457, 0.20267383289999996, 0.007955979, 1.2451951388, 0.4405458902000001)
    got = equ.to_keplerian()
    assert all(abs(g - w) <= 1e-15 * max(1.0, abs(w)) for g, w in zip(got, want)), got
    import platform
    if platform.libc_ver()[0] == "glibc":
        assert got == want


def test_round2_host_entry_points_without_a_gpu(L):
    """Host-side behaviour of the ABI v5 additions (no compute): math mode, SolverType defaults and validation, cache."""
    from outfit_amd import _abi
    assert L.outfit_hip_abi_version() == 7
    prev = L.outfit_hip_get_math_mode()
    assert prev == _abi_const("OUTFIT_MATH_GLIBC")        # the drop-in default (unless OUTFIT_MATH_MODE=cr)
    try:
        assert L.outfit_hip_set_math_mode(_abi_const("OUTFIT_MATH_GLIBC")) == 0 and L.outfit_hip_get_math_mode() == 1
        assert L.outfit_hip_set_math_mode(7) == -1 and b"math mode" in L.outfit_hip_last_error()
        assert L.outfit_hip_get_math_mode() == 1
    finally:
        L.outfit_hip_set_math_mode(prev)
    st = _abi.SolverTypeC()
    L.outfit_hip_solver_type_default(C.byref(st))        # SolverType::default(), kepler/params.rs:35-44,66-73
    assert (st.kind, st.parabolic_solving_method, st.max_iter_prelim_kepuni, st.has_psi_guess) == (0, 0, 20, 0)
    assert st.convergency == 100 * 2.220446049250313e-16
    dp = C.POINTER(C.c_double)
    st.kind = 3
    assert L.outfit_hip_propagate_universal_ex(0, None, None, None, C.byref(st), None, None) == -1   # INVALID_PARAM
    st.kind, st.parabolic_solving_method = 2, 5
    assert L.outfit_hip_propagate_universal_ex(0, None, None, None, C.byref(st), None, None) == -1
    st.parabolic_solving_method = 1
    assert L.outfit_hip_propagate_universal_ex(0, None, None, None, C.byref(st), None, None) == 0    # nothing to do
    assert L.outfit_hip_propagate_universal_ex(1, None, None, None, C.byref(st), None, None) == -2   # INVALID_BATCH
    assert L.outfit_hip_trim_cache() == 0
    assert L.outfit_hip_session_check(None) == -6                                                    # BAD_HANDLE


def _abi_const(name):
    import re
    src = open(os.path.join(ROOT, "include", "outfit_hip.h")).read()
    return int(re.search(rf"#define {name}\s+\(?(-?\d+)\)?", src).group(1))


def test_call_options_abi_v6_without_a_gpu(L):
    """outfit_hip_call_options_t: defaults, and rejection of a bad mode / a short struct before any device work."""
    from outfit_amd import _abi
    import outfit_amd
    from outfit_amd import synth
    o = _abi.CallOptionsC()
    L.outfit_hip_call_options_default(C.byref(o))
    assert o.struct_size == C.sizeof(_abi.CallOptionsC) == 32 and o.math_mode == _abi_const("OUTFIT_MATH_DEFAULT") == -1
    ds = synth.make_dataset(4)
    batch, ephem = outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ds["ephem"])
    params = synth.bench_params(outfit_amd.IODParams())
    pc, bc, ec = params.as_c(), batch.as_c(), ephem.as_c()
    out = (_abi.GaussResultC * 4)()
    o.math_mode = 7
    assert L.outfit_hip_fit_full_iod_ex(C.byref(pc), C.byref(bc), C.byref(ec), 1, C.byref(o), out) == -1
    assert b"math_mode" in L.outfit_hip_last_error()
    # struct_size is checked FIRST and only the bytes it covers are read: 4 bytes do not reach math_mode -> refused,
    # whatever the mode field would say (OUTFIT_MATH_DEFAULT included)
    for mode in (1, -1):
        o.math_mode, o.struct_size = mode, 4
        assert L.outfit_hip_fit_full_iod_ex(C.byref(pc), C.byref(bc), C.byref(ec), 1, C.byref(o), out) == -1
        assert b"struct_size" in L.outfit_hip_last_error()
    # reserved words must be zero (so that a later ABI can give them a meaning) ...
    L.outfit_hip_call_options_default(C.byref(o))
    assert (o.n_obs_capacity, o.noise_z_capacity) == (0, 0)     # ABI v7: two of v6's reserved words; 0 = not stated
    o.reserved[0] = 5
    assert L.outfit_hip_fit_full_iod_ex(C.byref(pc), C.byref(bc), C.byref(ec), 1, C.byref(o), out) == -1
    assert b"reserved" in L.outfit_hip_last_error()
    # ... but are not read beyond struct_size: a caller struct that ends after math_mode is a valid shorter ABI
    o.math_mode, o.struct_size = 1, 8
    rc = L.outfit_hip_fit_full_iod_ex(C.byref(pc), C.byref(bc), C.byref(ec), 1, C.byref(o), out)
    assert b"reserved" not in L.outfit_hip_last_error() and b"struct_size" not in L.outfit_hip_last_error()
    assert rc in (0, _abi_const("OUTFIT_ERR_NO_DEVICE"))    # options accepted: the call gets as far as the device
    o.reserved[0] = 0
    o.math_mode, o.struct_size = 1, 4
    h = C.c_void_p()
    assert L.outfit_hip_session_create_ex(C.byref(pc), C.byref(bc), C.byref(ec), 0, C.byref(o), C.byref(h)) == -1


def test_issue_cost_microbenchmark_compiles(tmp_path):
    """tools/issue_cost.hip (the measured cycles per instruction behind bench.py's cycle-weighted issue fractions,
    profiles/round04_issue_cost_gfx950.json) is a stand-alone gfx950 program: it has to keep compiling."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "issue_cost.hip")
    r = subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-c", src, "-o", str(tmp_path / "issue_cost.o")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_no_exception_crosses_the_c_abi():
    """Every entry point that can allocate is a function-try-block: a C++ exception inside the library comes back as
    OUTFIT_ERR_NO_MEMORY with a message, never as an unwinding through the caller's (Rust / C / ctypes) frames."""
    L = _lib.lib()
    L.outfit_hip_selftest_abi_guard.argtypes = [C.c_int]
    L.outfit_hip_selftest_abi_guard.restype = C.c_int
    L.outfit_hip_last_error.restype = C.c_char_p
    assert L.outfit_hip_selftest_abi_guard(3) == 0
    # std::bad_alloc is a resource limit (-8, MemoryError in the Python mirror); anything else is a bug (-9, v7)
    for kind, text, code in ((0, b"allocation", _abi.OUTFIT_ERR_NO_MEMORY), (1, b"selftest", _abi.OUTFIT_ERR_INTERNAL),
                             (2, b"unknown", _abi.OUTFIT_ERR_INTERNAL)):
        assert L.outfit_hip_selftest_abi_guard(kind) == code
        assert text in L.outfit_hip_last_error()


def _overrunning_batch():
    """The offset table of the round-4 crash (gpurun_out/r4c, fixed in the TEST by e84fdce): blocks appended with
    `off.extend(off[-1] + n * (i + 1) ...)` evaluated lazily, so that the offsets grow as b + n, b + 3n, b + 6n, ... —
    non-decreasing, every trajectory shorter than 2^21 rows, and far past the end of the observation columns."""
    import outfit_amd
    from outfit_amd import synth
    ds = synth.make_dataset(40)
    n = 20
    off = [0]
    for _ in range(2):
        off.extend(off[-1] + n * (i + 1) for i in range(20))     # the generator reads off[-1] while extend() grows it
    off = np.array(off, dtype=np.uint64)
    assert (np.diff(off.astype(np.int64)) > 0).all() and int(off[-1]) > 40 * n
    cols = [ds[k] for k in ("mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec")]
    return outfit_amd, ds, off, cols


def test_offset_table_past_the_columns_is_an_error_not_a_crash(L):
    """the library indexes the columns with the offsets themselves. The Python mirror knows every
    length and refuses the batch (InvalidBatch, a ValueError); a C caller states the lengths in the call options
    (n_obs_capacity, ABI v7) and gets OUTFIT_ERR_INVALID_BATCH — before any device work, so this runs without a GPU."""
    from outfit_amd import _abi
    outfit_amd, ds, off, cols = _overrunning_batch()
    with pytest.raises(ValueError, match="runs past"):
        outfit_amd.ObsBatch(off, *cols, ds["obs_helio_equ"], ds["obs_geo_ecl"])
    good = outfit_amd.ObsBatch.from_dict(ds)
    # every column is checked, the 3-vectors against 3 x rows
    for k in ("ra", "sigma_dec"):
        d = dict(ds); d[k] = ds[k][:-1]
        with pytest.raises(outfit_amd.InvalidBatch, match=k):
            outfit_amd.ObsBatch.from_dict(d)
    for k in ("obs_helio_equ", "obs_geo_ecl"):
        d = dict(ds); d[k] = ds[k][:-3]
        with pytest.raises(outfit_amd.InvalidBatch, match=k):
            outfit_amd.ObsBatch.from_dict(d)
    with pytest.raises(outfit_amd.InvalidBatch, match="non-decreasing"):
        outfit_amd.ObsBatch(np.array([0, 40, 20], dtype=np.uint64), *cols, ds["obs_helio_equ"])
    with pytest.raises(outfit_amd.InvalidBatch, match="traj_key"):
        outfit_amd.ObsBatch(ds["traj_offset"], *cols, ds["obs_helio_equ"], traj_key=np.arange(3))
    # host-drawn normals against the parameters, at call time (attributes can be replaced after construction)
    params = outfit_amd.IODParams(n_noise_realizations=2, max_triplets=5)
    good.noise_z = np.zeros(40 * 5 * 2 * 6 - 1)
    with pytest.raises(outfit_amd.InvalidBatch, match="noise_z"):
        outfit_amd.fit_full_iod_records(good, outfit_amd.Ephemeris.from_dict(ds["ephem"]), params, n_devices=1)
    good.noise_z = None
    good.traj_offset = off                                     # the crash's table, put in behind the constructor's back
    with pytest.raises(ValueError, match="runs past"):
        outfit_amd.IODSession(good, outfit_amd.Ephemeris.from_dict(ds["ephem"]), outfit_amd.IODParams())
    # the C side: the same table with the lengths stated in the options
    good.traj_offset = ds["traj_offset"]
    pc, bc, ec = outfit_amd.IODParams().as_c(), good.as_c(), outfit_amd.Ephemeris.from_dict(ds["ephem"]).as_c()
    bc.traj_offset = off.ctypes.data_as(C.POINTER(C.c_uint64))
    o = _abi.CallOptionsC()
    L.outfit_hip_call_options_default(C.byref(o))
    o.n_obs_capacity = 40 * 20
    out = (_abi.GaussResultC * 40)()
    h = C.c_void_p()
    assert L.outfit_hip_fit_full_iod_ex(C.byref(pc), C.byref(bc), C.byref(ec), 1, C.byref(o), out) == _abi.OUTFIT_ERR_INVALID_BATCH
    assert b"n_obs_capacity" in L.outfit_hip_last_error()
    assert L.outfit_hip_session_create_ex(C.byref(pc), C.byref(bc), C.byref(ec), 0, C.byref(o), C.byref(h)) == _abi.OUTFIT_ERR_INVALID_BATCH
    assert not h.value
    # noise_z_capacity: one normal short
    bc.traj_offset = good.traj_offset.ctypes.data_as(C.POINTER(C.c_uint64))
    z = np.zeros(40 * 5 * 2 * 6 - 1)
    bc.noise_z = z.ctypes.data_as(C.POINTER(C.c_double))
    o.noise_z_capacity = z.size
    pn = params.as_c()
    assert L.outfit_hip_fit_full_iod_ex(C.byref(pn), C.byref(bc), C.byref(ec), 1, C.byref(o), out) == _abi.OUTFIT_ERR_INVALID_BATCH
    assert b"noise_z_capacity" in L.outfit_hip_last_error()
