// earthrot.hpp — device-side observer-position pre-pass.
//
// Replaces Observer::pvobs + helio_position (src/observer_extension.rs:180-237), i.e. what
// OutfitCache::build evaluates per observation (cache/observer_centric_cache.rs:80-110), with gmst
// (time.rs:326-365), obleq / nutn80 / rnut80 / equequ / prec (earth_orientation.rs:119-592), rotmt and the
// one rotpn path pvobs takes, Equt(of date) -> Equm(of date) -> Equm(J2000) -> Eclm(J2000)
// (ref_system.rs:230-421, pinned by its test at :783-810).
//
// Same operation order as oracle/earthrot.c (so that both give the same bits: the observer positions feed
// the chaotic correction loop) with the correctly-rounded sin/cos of crmath.hpp. The nutation is the IAU 1980
// series evaluated from a table (nutation_table.hpp, derived by tools/derive_nutation_table.py), not the
// reference's hand-expanded form: a few 1e-15 arcsec apart.
#pragma once
#include "ephem.hpp"
#include "nutation_table.hpp"

namespace outfit {
namespace erot {

constexpr double kT2000 = 51544.5;
constexpr double kRadeg = kPi / 180.0;
constexpr double kRadsec = kPi / 648000.0;

__device__ __forceinline__ double obleq(double tjm) {
  const double ob0 = ((23.0 * 3600.0 + 26.0 * 60.0) + 21.448) * kRadsec;
  const double ob1 = -46.815 * kRadsec, ob2 = -0.0006 * kRadsec, ob3 = 0.00181 * kRadsec;
  const double t = (tjm - kT2000) / 36525.0;
  return ((ob3 * t + ob2) * t + ob1) * t + ob0;
}

__device__ __forceinline__ double as_rad(double a0, double a1, double a2, double a3, double t, double t2, double t3) {
  return fma(a3, t3, fma(a2, t2, fma(a1, t, a0))) * kRadsec;
}

// arcseconds
__device__ __forceinline__ void nutn80(double tjm, double& dpsi, double& deps) {
  const double t = (tjm - kT2000) / 36525.0, t2 = t * t, t3 = t2 * t;
  const double a0 = as_rad(485866.733, 1717915922.633, 31.310, 0.064, t, t2, t3);
  const double a1 = as_rad(1287099.804, 129596581.224, -0.577, -0.012, t, t2, t3);
  const double a2 = as_rad(335778.877, 1739527263.137, -13.257, 0.011, t, t2, t3);
  const double a3 = as_rad(1072261.307, 1602961601.328, -6.891, 0.019, t, t2, t3);
  const double a4 = as_rad(450160.280, -6962890.539, 7.455, 0.008, t, t2, t3);
  double sp = 0.0, se = 0.0;
#pragma unroll 1
  for (int i = 0; i < NUT80_N_DPSI; ++i) {
    double a = 0.0;
    a += (double)NUT80_DPSI_K[i][0] * a0;
    a += (double)NUT80_DPSI_K[i][1] * a1;
    a += (double)NUT80_DPSI_K[i][2] * a2;
    a += (double)NUT80_DPSI_K[i][3] * a3;
    a += (double)NUT80_DPSI_K[i][4] * a4;
    double s, c;
    crm::sincos(a, s, c);
    sp += (NUT80_DPSI_A[i][0] + NUT80_DPSI_A[i][1] * t) * s;
  }
#pragma unroll 1
  for (int i = 0; i < NUT80_N_DEPS; ++i) {
    double a = 0.0;
    a += (double)NUT80_DEPS_K[i][0] * a0;
    a += (double)NUT80_DEPS_K[i][1] * a1;
    a += (double)NUT80_DEPS_K[i][2] * a2;
    a += (double)NUT80_DEPS_K[i][3] * a3;
    a += (double)NUT80_DEPS_K[i][4] * a4;
    double s, c;
    crm::sincos(a, s, c);
    se += (NUT80_DEPS_A[i][0] + NUT80_DEPS_A[i][1] * t) * c;
  }
  dpsi = sp * 1e-4;
  deps = se * 1e-4;
}

// Rotation3::from_axis_angle about coordinate axis K (compile-time)
template <int K>
__device__ __forceinline__ Mat3 rotmt(double alpha) {
  double s, c;
  crm::sincos(alpha, s, c);
  Mat3 m;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) m.m[i][j] = 0.0;
  constexpr int a = (K + 1) % 3, b = (K + 2) % 3;
  m.m[K][K] = 1.0;
  m.m[a][a] = c; m.m[b][b] = c;
  m.m[b][a] = s; m.m[a][b] = -s;
  return m;
}
__device__ __forceinline__ Mat3 matmul(const Mat3& A, const Mat3& B) {
  Mat3 R;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) R.m[i][j] = (A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j]) + A.m[i][2] * B.m[2][j];
  return R;
}
__device__ __forceinline__ Mat3 transpose(const Mat3& A) {
  Mat3 R;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R.m[i][j] = A.m[j][i];
  return R;
}

__device__ __forceinline__ double gmst(double tjm) {
  constexpr double C0 = 24110.54841, C1 = 8640184.812866, C2 = 9.3104e-2, C3 = -6.2e-6, RAP = 1.00273790934;
  const double itjm = floor(tjm);
  const double t = (itjm - kT2000) / 36525.0;
  double gmst0 = ((C3 * t + C2) * t + C1) * t + C0;
  gmst0 *= kTwoPi / 86400.0;
  const double h = (tjm - trunc(tjm)) * kTwoPi;
  double g = gmst0 + h * RAP;
  long long i = (long long)floor(g / kTwoPi);
  if (g < 0.0) i -= 1;
  g -= (double)i * kTwoPi;
  return g;
}

// pvobs (position): geocentric observer position, ecliptic J2000, from the Earth-fixed vector
__device__ __forceinline__ Vec3 pvobs(double tt, double tut, const Vec3& dxbf) {
  // one nutation evaluation serves equequ and rnut80 (same arguments, same values)
  double dpsi_as, deps_as;
  nutn80(tt, dpsi_as, deps_as);
  const double epsm = obleq(tt);
  double se, ce;
  crm::sincos(epsm, se, ce);
  const double eqeq = kRadsec * dpsi_as * ce;  // equequ (earth_orientation.rs:508-519)
  const double gast = gmst(tut) + eqeq;
  const Mat3 rot = rotmt<2>(-gast);
  // rnut80(tt) = (R1(eps) R3(-dpsi)) R1(-(eps + deps))   (:459-480)
  const double dpsi = dpsi_as * kRadsec;
  const double epst = epsm + deps_as * kRadsec;
  const Mat3 rnut = matmul(matmul(rotmt<0>(epsm), rotmt<2>(-dpsi)), rotmt<0>(-epst));
  // prec(tt) (:561-592)
  const double zed = 0.6406161 * kRadeg, zd = 0.6406161 * kRadeg, thd = 0.5567530 * kRadeg;
  const double zedd = 0.0000839 * kRadeg, zdd = 0.0003041 * kRadeg, thdd = -0.0001185 * kRadeg;
  const double zeddd = 0.0000050 * kRadeg, zddd = 0.0000051 * kRadeg, thddd = -0.0000116 * kRadeg;
  const double t = (tt - kT2000) / 36525.0;
  const double zeta = ((zeddd * t + zedd) * t + zed) * t;
  const double z = ((zddd * t + zdd) * t + zd) * t;
  const double theta = ((thddd * t + thdd) * t + thd) * t;
  const Mat3 prec = matmul(matmul(rotmt<2>(-zeta), rotmt<1>(theta)), rotmt<2>(-z));
  // rotpn(Equt(tt) -> Eclm(J2000)): I * rnut^T, * prec^T, * R1(obleq(J2000))
  Mat3 R;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R.m[i][j] = i == j ? 1.0 : 0.0;
  R = matmul(R, transpose(rnut));
  R = matmul(R, transpose(prec));
  R = matmul(R, rotmt<0>(obleq(kT2000)));
  const Mat3 rotmat = matmul(transpose(R), transpose(rot));
  return gemv(rotmat, dxbf);
}

// helio_position: earth (equatorial) + ROT_ECLMJ2000_TO_EQUMJ2000 * geo
__device__ __forceinline__ Vec3 helio_position(const Vec3& earth_equ, const Vec3& geo_ecl) {
  const Vec3 r = ecl_to_equ(geo_ecl);
  return {earth_equ.x + r.x, earth_equ.y + r.y, earth_equ.z + r.z};
}

}  // namespace erot
}  // namespace outfit
