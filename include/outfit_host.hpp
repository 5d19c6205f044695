// outfit_host.hpp — C++17 host-side mirror of the reference's FitIOD surface on top of
// the C ABI (include/outfit_hip.h). Header-only; link with -loutfit_hip.
//
// The reference is Rust; its toolchain is not available in this build, so the layer a
// maintainer would write in Rust is provided here in C++ with the
// reference's names and error behaviour:
//   IODParams / IODParams::builder()      src/initial_orbit_determination/mod.rs:225-630
//   ObsDataset (prepared)                 what prepare_iod leaves (obs_dataset_api.rs:254-275)
//   FitIOD::fit_full_iod / fit_iod        obs_dataset_api.rs:130-207
//   GaussResult, FitOrbitResult           gauss_result.rs:99-102, constants.rs:134-137
//   FullOrbitResult                       constants.rs:195 (map TrajId -> Result<..>)
//   OutfitError variants                  outfit_errors.rs:195-250
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "outfit_hip.h"

namespace outfit {

using TrajId = std::string;

// ---- errors (outfit_errors.rs) ------------------------------------------------------
struct OutfitError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct InvalidIODParameter : OutfitError { using OutfitError::OutfitError; };
struct InvalidBatch : OutfitError { using OutfitError::OutfitError; };
struct DeviceError : OutfitError { using OutfitError::OutfitError; };
struct TrajectoryIdNotFound : OutfitError { using OutfitError::OutfitError; };

// per-trajectory error values (never thrown by fit_full_iod: they are map entries)
struct NoFeasibleTriplets { double span; std::size_t n_obs; double dt_min, dt_max; };
struct NoViableOrbit { std::string cause; std::size_t attempts; };
struct InvalidConversion { std::string what; };
struct InvalidOrbit { std::string what; };
using TrajError = std::variant<NoFeasibleTriplets, NoViableOrbit, InvalidConversion, InvalidOrbit>;

// ---- math mode (include/outfit_hip.h OUTFIT_MATH_*): which bits sin/cos/sinh/cosh return where the reference calls
// the platform libm. Glibc (the library's default) reproduces a glibc 2.35 / x86-64 host — the platform the reference's
// known answers were recorded on; CorrectlyRounded is platform independent. `Default` = the process-wide default.
enum class MathMode : int { Default = OUTFIT_MATH_DEFAULT, CorrectlyRounded = OUTFIT_MATH_CORRECTLY_ROUNDED, Glibc = OUTFIT_MATH_GLIBC };
// Per-call options (outfit_hip_call_options_t): what a call depends on besides its arguments. Calls with different
// options may run concurrently from several host threads.
struct CallOptions : outfit_hip_call_options_t {
  CallOptions() { outfit_hip_call_options_default(this); }
  explicit CallOptions(MathMode m) { outfit_hip_call_options_default(this); math_mode = static_cast<int>(m); }
};
inline void set_math_mode(MathMode m) {  // the process-wide default only
  if (outfit_hip_set_math_mode(static_cast<int>(m)) != OUTFIT_OK) throw InvalidIODParameter(outfit_hip_last_error());
}
inline MathMode math_mode() { return static_cast<MathMode>(outfit_hip_get_math_mode()); }

// ---- SolverType of kepler::propagate_universal (kepler/params.rs:24-73) -----------------------------
struct SolverType : outfit_solver_type_t {
  SolverType() { outfit_hip_solver_type_default(this); }
};
struct UniversalPropagResult { double r1[3], v1[3], f_lag, g_lag, f_dot, g_dot, psy; };  // propagation.rs:66-82
// kepler::propagate_universal for one state; throws DeviceError with the reference's error kind on failure
inline UniversalPropagResult propagate_universal(const double position[3], const double velocity[3], double t0, double t1,
                                                 const SolverType& solver = SolverType()) {
  const double rv[6] = {position[0], position[1], position[2], velocity[0], velocity[1], velocity[2]};
  double out[11];
  int32_t st = 0;
  if (outfit_hip_propagate_universal_ex(1, rv, &t0, &t1, &solver, out, &st) != OUTFIT_OK)
    throw DeviceError(outfit_hip_last_error());
  if (st == 1 || st == 3) throw DeviceError("DegenerateState");
  if (st == 2) throw DeviceError("NewtonRaphsonKeplerConvergence");
  if (st == 4) throw DeviceError("BrentDekkerKeplerConvergence");
  UniversalPropagResult r;
  for (int i = 0; i < 3; ++i) { r.r1[i] = out[i]; r.v1[i] = out[3 + i]; }
  r.f_lag = out[6]; r.g_lag = out[7]; r.f_dot = out[8]; r.g_dot = out[9]; r.psy = out[10];
  return r;
}

// ---- parameters -----------------------------------------------------------------------
class IODParamsBuilder;
struct IODParams : outfit_iod_params_t {
  IODParams() { outfit_hip_iod_params_default(this); }  // IODParams::default(), mod.rs:308-344
  static IODParamsBuilder builder();
};
class IODParamsBuilder {
  IODParams p_;

 public:
#define OUTFIT_SETTER(name, type) \
  IODParamsBuilder& name(type v) { p_.name = v; return *this; }
  OUTFIT_SETTER(n_noise_realizations, std::uint64_t)
  OUTFIT_SETTER(noise_scale, double)
  OUTFIT_SETTER(extf, double)
  OUTFIT_SETTER(dtmax, double)
  OUTFIT_SETTER(dt_min, double)
  OUTFIT_SETTER(dt_max_triplet, double)
  OUTFIT_SETTER(optimal_interval_time, double)
  OUTFIT_SETTER(max_obs_for_triplets, std::uint64_t)
  OUTFIT_SETTER(max_triplets, std::uint32_t)
  OUTFIT_SETTER(gap_max, double)
  OUTFIT_SETTER(max_ecc, double)
  OUTFIT_SETTER(max_perihelion_au, double)
  OUTFIT_SETTER(min_rho2_au, double)
  OUTFIT_SETTER(r2_min_au, double)
  OUTFIT_SETTER(r2_max_au, double)
  OUTFIT_SETTER(aberth_max_iter, std::uint32_t)
  OUTFIT_SETTER(aberth_eps, double)
  OUTFIT_SETTER(kepler_eps, double)
  OUTFIT_SETTER(max_tested_solutions, std::uint64_t)
  OUTFIT_SETTER(newton_eps, double)
  OUTFIT_SETTER(newton_max_it, std::uint64_t)
  OUTFIT_SETTER(root_imag_eps, double)
#undef OUTFIT_SETTER
  // IODParamsBuilder::build (mod.rs:544-624)
  IODParams build() const {
    if (outfit_hip_iod_params_validate(&p_) != OUTFIT_OK) throw InvalidIODParameter(outfit_hip_last_error());
    return p_;
  }
};
inline IODParamsBuilder IODParams::builder() { return IODParamsBuilder(); }

// ---- prepared dataset -----------------------------------------------------------------
// One observation after prepare_iod: TT epoch, (ra, dec) with their model errors, and the two cached
// observer vectors (cache/mod.rs:144-174).
struct Observation {
  double mjd_tt, ra, dec, ra_error, dec_error;
  double helio_equ[3];  // cache.get_helio_position
  double geo_ecl[3];    // cache.get_observer_geocentric_position
  // ABI v4: with ObsDataset::positions_on_device the two vectors above are ignored and the library runs the
  // observer pre-pass itself (Observer::pvobs / helio_position, src/observer_extension.rs:180-237) from these:
  double body_fixed[3] = {0, 0, 0};  // Observer::earth_fixed_position, AU
  double mjd_ut1 = 0;                // the observation epoch on the UT1 scale
};

class ObsDataset {
  std::map<TrajId, std::vector<Observation>> traj_;

 public:
  bool positions_on_device = false;  // hand over (body_fixed, mjd_ut1) instead of cached observer positions
  void push(const TrajId& id, const Observation& o) { traj_[id].push_back(o); }
  std::size_t n_trajectories() const { return traj_.size(); }
  const std::map<TrajId, std::vector<Observation>>& trajectories() const { return traj_; }
};

struct JPLEphem {  // legacy-DE subset, layout of outfit_ephem_t; empty = use the cached helio positions
  double jd_start = 0, jd_end = 0, block_days = 32, emrat = 81.30056822149722;
  std::uint32_t ncoef[3] = {13, 13, 11}, nsub[3] = {2, 8, 2};
  std::vector<double> coef;
  bool empty() const { return coef.empty(); }
  // HorizonData::read_horizon_file (jpl_ephem/horizon/horizon_data.rs:598-684): a little-endian legacy-DE binary
  static JPLEphem from_file(const std::string& path) {
    outfit_ephem_file_t* h = nullptr;
    if (outfit_hip_ephem_open(path.c_str(), &h) != OUTFIT_OK) throw OutfitError(outfit_hip_last_error());
    outfit_ephem_t v{};
    outfit_hip_ephem_view(h, &v);
    JPLEphem e;
    e.jd_start = v.jd_start; e.jd_end = v.jd_end; e.block_days = v.block_days; e.emrat = v.emrat;
    std::size_t per = 0;
    for (int k = 0; k < 3; ++k) { e.ncoef[k] = v.ncoef[k]; e.nsub[k] = v.nsub[k]; per += 3u * v.ncoef[k] * v.nsub[k]; }
    e.coef.assign(v.coef, v.coef + v.n_blocks * per);
    outfit_hip_ephem_close(h);
    return e;
  }
};

// ---- results ----------------------------------------------------------------------------
struct KeplerianElements { double reference_epoch, semi_major_axis, eccentricity, inclination,
                                  ascending_node_longitude, periapsis_argument, mean_anomaly; };
struct CometaryElements { double reference_epoch, perihelion_distance, eccentricity, inclination,
                                 ascending_node_longitude, periapsis_argument, true_anomaly; };
using OrbitalElements = std::variant<KeplerianElements, CometaryElements>;
struct GaussResult {
  enum Stage { PrelimOrbit, CorrectedOrbit } stage;
  OrbitalElements orbit;
  const OrbitalElements& get_orbit() const { return orbit; }
};
struct FitOrbitResult {  // FitOrbitResult::IODGauss((GaussResult, IODRMS))
  GaussResult gauss;
  double rms;
  double orbit_quality() const { return rms; }
  const OrbitalElements& orbital_elements() const { return gauss.orbit; }
};
using TrajResult = std::variant<FitOrbitResult, TrajError>;  // Result<FitOrbitResult, OutfitError>
using FullOrbitResult = std::map<TrajId, TrajResult>;

// ---- FitIOD ---------------------------------------------------------------------------------
class FitIOD {
 public:
  // fit_full_iod (obs_dataset_api.rs:145-207) after prepare_iod. `noise_z` (optional): per trajectory, in
  // map order, max_triplets * n_noise_realizations * 6 standard normals.
  static FullOrbitResult fit_full_iod(const ObsDataset& ds, const JPLEphem& jpl, const IODParams& params,
                                      const std::vector<double>* noise_z = nullptr, int n_devices = 0,
                                      const CallOptions& options = CallOptions()) {
    if (outfit_hip_iod_params_validate(&params) != OUTFIT_OK) throw InvalidIODParameter(outfit_hip_last_error());
    Flat f = flatten(ds);
    outfit_obs_batch_t b = f.batch();
    if (noise_z) b.noise_z = noise_z->data();
    outfit_ephem_t e{};
    if (!jpl.empty()) e = ephem(jpl);
    std::vector<outfit_gauss_result_t> out(f.ids.size());
    // the lengths of what is handed over (ABI v7): the columns are this function's own; the normals are the caller's —
    // a vector shorter than n_traj * max_triplets * n_noise_realizations * 6 comes back as InvalidBatch, not as a read
    // past its end
    CallOptions opt = options;
    opt.n_obs_capacity = f.t.size();
    if (noise_z) opt.noise_z_capacity = noise_z->size();
    const int rc = outfit_hip_fit_full_iod_ex(&params, &b, jpl.empty() ? nullptr : &e, n_devices, &opt, out.data());
    if (rc == OUTFIT_ERR_INVALID_PARAM) throw InvalidIODParameter(outfit_hip_last_error());
    if (rc == OUTFIT_ERR_INVALID_BATCH) throw InvalidBatch(outfit_hip_last_error());
    if (rc != OUTFIT_OK) throw DeviceError(outfit_hip_last_error());
    FullOrbitResult res;
    for (std::size_t i = 0; i < f.ids.size(); ++i) res.emplace(f.ids[i], convert(out[i], f, i, params));
    return res;
  }

  // fit_iod (obs_dataset_api.rs:130-143): one trajectory; errors are thrown like the reference's Err
  static FitOrbitResult fit_iod(const ObsDataset& ds, const TrajId& id, const JPLEphem& jpl, const IODParams& params) {
    auto it = ds.trajectories().find(id);
    if (it == ds.trajectories().end()) throw TrajectoryIdNotFound(id);
    ObsDataset one;
    for (const auto& o : it->second) one.push(id, o);
    FullOrbitResult r = fit_full_iod(one, jpl, params, nullptr, 1);
    const TrajResult& tr = r.at(id);
    if (auto ok = std::get_if<FitOrbitResult>(&tr)) return *ok;
    throw OutfitError("fit_iod failed for " + id + ": " + describe(std::get<TrajError>(tr)));
  }

  static std::string describe(const TrajError& e) {
    if (auto p = std::get_if<NoFeasibleTriplets>(&e))
      return "NoFeasibleTriplets{span=" + std::to_string(p->span) + ", n_obs=" + std::to_string(p->n_obs) + "}";
    if (auto p = std::get_if<NoViableOrbit>(&e))
      return "NoViableOrbit{cause=" + p->cause + ", attempts=" + std::to_string(p->attempts) + "}";
    if (auto p = std::get_if<InvalidConversion>(&e)) return "InvalidConversion(" + p->what + ")";
    return "InvalidOrbit(" + std::get<InvalidOrbit>(e).what + ")";
  }

 private:
  friend class FitLSQ;
  struct Flat {
    std::vector<TrajId> ids;
    std::vector<std::uint64_t> off;
    std::vector<double> t, ra, dec, sra, sdec, helio, geo, fixed, ut1;
    bool on_device = false;
    outfit_obs_batch_t batch() const {
      outfit_obs_batch_t b{};
      b.n_traj = ids.size(); b.traj_offset = off.data();
      b.mjd_tt = t.data(); b.ra = ra.data(); b.dec = dec.data(); b.sigma_ra = sra.data(); b.sigma_dec = sdec.data();
      if (on_device) { b.obs_body_fixed = fixed.data(); b.mjd_ut1 = ut1.data(); }
      else { b.obs_helio_equ = helio.data(); b.obs_geo_ecl = geo.data(); }
      return b;
    }
  };
  // materialize_trajectory + sort by mjd_tt (fit_single_traj, obs_dataset_api.rs:218-223; stable like sort_by)
  static Flat flatten(const ObsDataset& ds) {
    Flat f;
    f.on_device = ds.positions_on_device;
    for (const auto& [id, obs_in] : ds.trajectories()) {
      std::vector<Observation> obs = obs_in;
      std::stable_sort(obs.begin(), obs.end(), [](const Observation& a, const Observation& b) { return a.mjd_tt < b.mjd_tt; });
      f.ids.push_back(id);
      f.off.push_back(f.t.size());
      for (const auto& o : obs) {
        f.t.push_back(o.mjd_tt); f.ra.push_back(o.ra); f.dec.push_back(o.dec);
        f.sra.push_back(o.ra_error); f.sdec.push_back(o.dec_error);
        f.helio.insert(f.helio.end(), o.helio_equ, o.helio_equ + 3);
        f.geo.insert(f.geo.end(), o.geo_ecl, o.geo_ecl + 3);
        f.fixed.insert(f.fixed.end(), o.body_fixed, o.body_fixed + 3);
        f.ut1.push_back(o.mjd_ut1);
      }
    }
    f.off.push_back(f.t.size());
    return f;
  }
  static outfit_ephem_t ephem(const JPLEphem& j) {
    outfit_ephem_t e{};
    e.jd_start = j.jd_start; e.jd_end = j.jd_end; e.block_days = j.block_days; e.emrat = j.emrat;
    std::size_t per = 0;
    for (int k = 0; k < 3; ++k) { e.ncoef[k] = j.ncoef[k]; e.nsub[k] = j.nsub[k]; per += 3u * j.ncoef[k] * j.nsub[k]; }
    e.n_blocks = per ? j.coef.size() / per : 0;
    e.coef = j.coef.data();
    return e;
  }
  static TrajResult convert(const outfit_gauss_result_t& r, const Flat& f, std::size_t i, const IODParams& p) {
    static const char* causes[] = {"None", "GaussNoRootsFound", "SingularDirectionMatrix",
                                   "PolynomialRootFindingFailed", "NonFiniteScore"};
    switch (r.status) {
      case OUTFIT_TRAJ_OK: {
        GaussResult g;
        g.stage = r.stage == OUTFIT_STAGE_CORRECTED ? GaussResult::CorrectedOrbit : GaussResult::PrelimOrbit;
        if (r.kind == OUTFIT_KIND_KEPLERIAN)
          g.orbit = KeplerianElements{r.epoch, r.elem[0], r.elem[1], r.elem[2], r.elem[3], r.elem[4], r.elem[5]};
        else
          g.orbit = CometaryElements{r.epoch, r.elem[0], r.elem[1], r.elem[2], r.elem[3], r.elem[4], r.elem[5]};
        return FitOrbitResult{g, r.rms};
      }
      case OUTFIT_TRAJ_NO_FEASIBLE_TRIPLETS: {
        const std::size_t n = f.off[i + 1] - f.off[i];
        const double span = n ? f.t[f.off[i + 1] - 1] - f.t[f.off[i]] : 0.0;
        return TrajError{NoFeasibleTriplets{span, n, p.dt_min, p.dt_max_triplet}};
      }
      case OUTFIT_TRAJ_NO_VIABLE_ORBIT:
        return TrajError{NoViableOrbit{causes[r.cause >= 0 && r.cause <= 4 ? r.cause : 0], r.attempts}};
      case OUTFIT_TRAJ_INVALID_CONVERSION:
        return TrajError{InvalidConversion{"Parabolic orbit cannot be represented with finite a"}};
      default:
        return TrajError{InvalidOrbit{"eccentricity must be > 1 for hyperbolic orbits"}};
    }
  }
};

// ---- FitLSQ (differential_orbit_correction/) ---------------------------------------------------
// DifferentialCorrectionConfig (diff_cor.rs:39-178); defaults = DifferentialCorrectionConfig::default()
struct DifferentialCorrectionConfig : outfit_diffcor_config_t {
  DifferentialCorrectionConfig() { outfit_hip_diffcor_config_default(this); }
};
struct EquinoctialElements { double reference_epoch, semi_major_axis, eccentricity_sin_lon, eccentricity_cos_lon,
                                    tan_half_incl_sin_node, tan_half_incl_cos_node, mean_longitude; };
// FitOrbitResult::DifferentialCorrection((OrbitalElements::Equinoctial{elements, uncertainty, covariance}, Chi2))
struct DifferentialCorrectionFit {
  EquinoctialElements elements;
  std::array<double, 36> covariance;  // row-major, rescaled (least_square.rs:341-362)
  double normalised_rms;
  unsigned total_newton_iterations, num_measurements, n_rejected;
  // EquinoctialUncertainty::from_covariance (orbit_type/uncertainty.rs:261-271)
  std::array<double, 6> uncertainty() const {
    std::array<double, 6> u{};
    for (int j = 0; j < 6; ++j) u[j] = std::sqrt(covariance[7 * j]);
    return u;
  }
  double orbit_quality() const { return normalised_rms; }
};
// Ok(DifferentialCorrection) | Ok(the IOD orbit, when the corrector erred: mod.rs:109) | Err
using LsqTrajResult = std::variant<DifferentialCorrectionFit, FitOrbitResult, TrajError>;
using FullLsqResult = std::map<TrajId, LsqTrajResult>;

class FitLSQ {
 public:
  // fit_lsq (obs_dataset_api.rs:57-99) after the dataset preparation. `initial_orbits`: a previous
  // FullOrbitResult; trajectories without an Ok entry get the IOD first (obs_dataset_api.rs:118-120).
  static FullLsqResult fit_lsq(const ObsDataset& ds, const JPLEphem& jpl, const IODParams& iod_params,
                               const DifferentialCorrectionConfig& config,
                               const FullOrbitResult* initial_orbits = nullptr,
                               const std::vector<double>* noise_z = nullptr, int n_devices = 0) {
    FitIOD::Flat f = FitIOD::flatten(ds);
    outfit_obs_batch_t b = f.batch();
    if (noise_z) b.noise_z = noise_z->data();
    outfit_ephem_t e{};
    if (!jpl.empty()) e = FitIOD::ephem(jpl);
    std::vector<outfit_gauss_result_t> init;
    if (initial_orbits) {
      init.assign(f.ids.size(), outfit_gauss_result_t{});
      for (std::size_t i = 0; i < f.ids.size(); ++i) {
        init[i].status = OUTFIT_TRAJ_NO_VIABLE_ORBIT;  // "no usable entry"
        auto it = initial_orbits->find(f.ids[i]);
        if (it == initial_orbits->end()) continue;
        if (auto ok = std::get_if<FitOrbitResult>(&it->second)) init[i] = to_record(*ok);
      }
    }
    std::vector<outfit_lsq_result_t> out(f.ids.size());
    CallOptions opt;
    opt.n_obs_capacity = f.t.size();
    if (noise_z) opt.noise_z_capacity = noise_z->size();
    const int rc = outfit_hip_fit_lsq_ex(&iod_params, &config, &b, jpl.empty() ? nullptr : &e,
                                         initial_orbits ? init.data() : nullptr, n_devices, &opt, out.data(), nullptr);
    if (rc == OUTFIT_ERR_INVALID_PARAM) throw InvalidIODParameter(outfit_hip_last_error());
    if (rc == OUTFIT_ERR_INVALID_BATCH) throw InvalidBatch(outfit_hip_last_error());
    if (rc != OUTFIT_OK) throw DeviceError(outfit_hip_last_error());
    FullLsqResult res;
    for (std::size_t i = 0; i < f.ids.size(); ++i) {
      const outfit_lsq_result_t& r = out[i];
      if (r.status == OUTFIT_LSQ_DIFFCOR) {
        DifferentialCorrectionFit d{};
        d.elements = {r.epoch, r.elem[0], r.elem[1], r.elem[2], r.elem[3], r.elem[4], r.elem[5]};
        std::copy(r.covariance, r.covariance + 36, d.covariance.begin());
        d.normalised_rms = r.normalised_rms;
        d.total_newton_iterations = r.total_newton_iterations;
        d.num_measurements = r.num_measurements;
        d.n_rejected = r.n_rejected;
        res.emplace(f.ids[i], d);
      } else if (r.status == OUTFIT_LSQ_CONVERSION_ERROR) {
        if (r.conv_error == OUTFIT_TRAJ_INVALID_CONVERSION)
          res.emplace(f.ids[i], TrajError{InvalidConversion{"Parabolic orbit cannot be represented with finite a"}});
        else
          res.emplace(f.ids[i], TrajError{InvalidOrbit{"eccentricity must be > 1 for hyperbolic orbits"}});
      } else {  // IOD_KEPT (Ok(initial_orbit)) or IOD_ERROR: both are what the IOD record says
        TrajResult t = FitIOD::convert(r.iod, f, i, iod_params);
        if (auto ok = std::get_if<FitOrbitResult>(&t)) res.emplace(f.ids[i], *ok);
        else res.emplace(f.ids[i], std::get<TrajError>(t));
      }
    }
    return res;
  }

 private:
  static outfit_gauss_result_t to_record(const FitOrbitResult& fit) {
    outfit_gauss_result_t r{};
    r.status = OUTFIT_TRAJ_OK;
    r.stage = fit.gauss.stage == GaussResult::CorrectedOrbit ? OUTFIT_STAGE_CORRECTED : OUTFIT_STAGE_PRELIM;
    r.rms = fit.rms;
    if (auto k = std::get_if<KeplerianElements>(&fit.gauss.orbit)) {
      r.kind = OUTFIT_KIND_KEPLERIAN; r.epoch = k->reference_epoch;
      const double e[6] = {k->semi_major_axis, k->eccentricity, k->inclination, k->ascending_node_longitude,
                           k->periapsis_argument, k->mean_anomaly};
      std::copy(e, e + 6, r.elem);
    } else {
      const auto& c = std::get<CometaryElements>(fit.gauss.orbit);
      r.kind = OUTFIT_KIND_COMETARY; r.epoch = c.reference_epoch;
      const double e[6] = {c.perihelion_distance, c.eccentricity, c.inclination, c.ascending_node_longitude,
                           c.periapsis_argument, c.true_anomaly};
      std::copy(e, e + 6, r.elem);
    }
    return r;
  }
};

}  // namespace outfit
