/*
 * earthrot.c — oracle restatement of the observer-position pre-pass:
 * Observer::pvobs / helio_position (src/observer_extension.rs:180-237) with gmst (src/time.rs:326-365),
 * obleq / nutn80 / rnut80 / equequ / prec (src/earth_orientation.rs:119-592) and rotpn / rotmt
 * (src/ref_system.rs:230-462). TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * One deliberate difference: nutn80 evaluates the IAU 1980 series from a TABLE (nutation_table.h, derived
 * from the reference's behaviour by tools/derive_nutation_table.py) — sum (A + A't) sin / cos (k . args) —
 * instead of the reference's hand-expanded angle-addition algebra. Both are the same 106-term series; the
 * results agree to a few 1e-15 arcsec (reference KAT at J2000: 5e-15), so the reference's bit-exact KATs of
 * nutn80 / rnut80 / rotpn are met to ~1e-15 rather than to the bit. sin / cos are the correctly-rounded ones
 * (crmath.c) so that the device (device/earthrot.hpp) reproduces these functions bit for bit: the observer
 * positions feed the chaotic correction loop.
 * nalgebra restated: Rotation3::from_axis_angle for a coordinate axis, 3x3 gemm (column axpy) and transpose.
 */
#include <math.h>
#include <string.h>
#include "oracle.h"
#include "nutation_table.h"

/* correctly rounded in both math modes (crmath.c, orc_sincos_prepass) */
static double er_sin(double x) { double s, c; orc_sincos_prepass(x, &s, &c); return s; }
static double er_cos(double x) { double s, c; orc_sincos_prepass(x, &s, &c); return c; }

#define T2000 51544.5
#define RADEG (ORC_PI / 180.0)
#define RADSEC (ORC_PI / 648000.0)

/* earth_orientation.rs:119-130 */
double orc_obleq(double tjm) {
  double ob0 = ((23.0 * 3600.0 + 26.0 * 60.0) + 21.448) * RADSEC;
  double ob1 = -46.815 * RADSEC, ob2 = -0.0006 * RADSEC, ob3 = 0.00181 * RADSEC;
  double t = (tjm - T2000) / 36525.0;
  return ((ob3 * t + ob2) * t + ob1) * t + ob0;
}

static double as_rad(double a0, double a1, double a2, double a3, double t, double t2, double t3) {
  return fma(a3, t3, fma(a2, t2, fma(a1, t, a0))) * RADSEC; /* f64::mul_add chain, :178-181 */
}

/* earth_orientation.rs:170-456 (series from the table); arcseconds */
void orc_nutn80(double tjm, double *dpsi, double *deps) {
  double t = (tjm - T2000) / 36525.0, t2 = t * t, t3 = t2 * t;
  double arg[5];
  arg[0] = as_rad(485866.733, 1717915922.633, 31.310, 0.064, t, t2, t3);  /* l  */
  arg[1] = as_rad(1287099.804, 129596581.224, -0.577, -0.012, t, t2, t3); /* l' */
  arg[2] = as_rad(335778.877, 1739527263.137, -13.257, 0.011, t, t2, t3); /* F  */
  arg[3] = as_rad(1072261.307, 1602961601.328, -6.891, 0.019, t, t2, t3); /* D  */
  arg[4] = as_rad(450160.280, -6962890.539, 7.455, 0.008, t, t2, t3);     /* Om */
  double sp = 0.0, se = 0.0;
  for (int i = 0; i < NUT80_N_DPSI; ++i) {
    double a = 0.0;
    for (int j = 0; j < 5; ++j) a += (double)NUT80_DPSI_K[i][j] * arg[j];
    sp += (NUT80_DPSI_A[i][0] + NUT80_DPSI_A[i][1] * t) * er_sin(a);
  }
  for (int i = 0; i < NUT80_N_DEPS; ++i) {
    double a = 0.0;
    for (int j = 0; j < 5; ++j) a += (double)NUT80_DEPS_K[i][j] * arg[j];
    se += (NUT80_DEPS_A[i][0] + NUT80_DEPS_A[i][1] * t) * er_cos(a);
  }
  *dpsi = sp * 1e-4;
  *deps = se * 1e-4;
}

/* ref_system.rs:453-462: Rotation3::from_axis_angle about coordinate axis k (m[row][col]) */
void orc_rotmt(double alpha, int k, double m[3][3]) {
  double s, c;
  orc_sincos_prepass(alpha, &s, &c);
  memset(m, 0, 9 * sizeof(double));
  int a = (k + 1) % 3, b = (k + 2) % 3;
  m[k][k] = 1.0;
  m[a][a] = c; m[b][b] = c;
  m[b][a] = s; m[a][b] = -s;
}
static void matmul3(const double A[3][3], const double B[3][3], double C[3][3]) {
  double R[3][3];
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) R[i][j] = (A[i][0] * B[0][j] + A[i][1] * B[1][j]) + A[i][2] * B[2][j];
  memcpy(C, R, sizeof R);
}
static void transpose3(const double A[3][3], double T[3][3]) {
  double R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = A[j][i];
  memcpy(T, R, sizeof R);
}

/* earth_orientation.rs:459-480 */
void orc_rnut80(double tjm, double m[3][3]) {
  double epsm = orc_obleq(tjm), dpsi, deps;
  orc_nutn80(tjm, &dpsi, &deps);
  dpsi *= RADSEC;
  double epst = epsm + deps * RADSEC;
  double r1[3][3], r2[3][3], r3[3][3], t[3][3];
  orc_rotmt(epsm, 0, r1);
  orc_rotmt(-dpsi, 2, r2);
  orc_rotmt(-epst, 0, r3);
  matmul3(r1, r2, t);
  matmul3(t, r3, m);
}

/* earth_orientation.rs:508-519 */
double orc_equequ(double tjm) {
  double oblm = orc_obleq(tjm), dpsi, deps;
  orc_nutn80(tjm, &dpsi, &deps);
  return RADSEC * dpsi * er_cos(oblm);
}

/* earth_orientation.rs:561-592 */
void orc_prec(double tjm, double m[3][3]) {
  double zed = 0.6406161 * RADEG, zd = 0.6406161 * RADEG, thd = 0.5567530 * RADEG;
  double zedd = 0.0000839 * RADEG, zdd = 0.0003041 * RADEG, thdd = -0.0001185 * RADEG;
  double zeddd = 0.0000050 * RADEG, zddd = 0.0000051 * RADEG, thddd = -0.0000116 * RADEG;
  double t = (tjm - T2000) / 36525.0;
  double zeta = ((zeddd * t + zedd) * t + zed) * t;
  double z = ((zddd * t + zdd) * t + zd) * t;
  double theta = ((thddd * t + thdd) * t + thd) * t;
  double r1[3][3], r2[3][3], r3[3][3], u[3][3];
  orc_rotmt(-zeta, 2, r1);
  orc_rotmt(theta, 1, r2);
  orc_rotmt(-z, 2, r3);
  matmul3(r1, r2, u);
  matmul3(u, r3, m);
}

/* ref_system.rs:230-300,379-421 ; kind: 0 Equm, 1 Equt, 2 Eclm ; returns 0 ok, 1 Err(InvalidRefSystem) */
static double epoch_date(const orc_ref_system *s) { return s->j2000 ? T2000 : s->date; }
int orc_rotpn(const orc_ref_system *src, const orc_ref_system *dst, double rot[3][3]) {
  orc_ref_system cur = *src;
  double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int it = 0; it < 20; ++it) {
    int equal = (cur.j2000 && dst->j2000) ? 1 : fabs(epoch_date(&cur) - epoch_date(dst)) <= 1e-6;
    double step[3][3];
    if (!equal) { /* transform_to_equm_date(dst.epoch()) */
      double d = epoch_date(&cur);
      if (cur.kind == 2) { orc_rotmt(-orc_obleq(d), 1, step); cur.kind = 0; }          /* axis 1, as the reference */
      else if (cur.kind == 1) { orc_rnut80(d, step); transpose3(step, step); cur.kind = 0; }
      else if (cur.j2000) {
        if (dst->j2000) return 1;
        orc_prec(dst->date, step);
        cur.j2000 = 0; cur.date = dst->date;
      } else {
        orc_prec(d, step); transpose3(step, step);
        cur.j2000 = 1; cur.date = 0.0;
      }
      matmul3(R, step, R);
      continue;
    }
    if (cur.kind == dst->kind) { memcpy(rot, R, sizeof R); return 0; }
    double d = epoch_date(&cur); /* transform_to_target_system(dst) */
    if (cur.kind == 1) { orc_rnut80(d, step); transpose3(step, step); cur.kind = 0; }
    else if (cur.kind == 2) { orc_rotmt(-orc_obleq(d), 0, step); cur.kind = 0; }
    else if (dst->kind == 1) { orc_rnut80(d, step); cur.kind = 1; }
    else if (dst->kind == 2) { orc_rotmt(orc_obleq(d), 0, step); cur.kind = 2; }
    else return 1;
    matmul3(R, step, R);
  }
  return 1;
}

/* time.rs:326-365 */
double orc_gmst(double tjm) {
  const double C0 = 24110.54841, C1 = 8640184.812866, C2 = 9.3104e-2, C3 = -6.2e-6, RAP = 1.00273790934;
  double itjm = floor(tjm);
  double t = (itjm - T2000) / 36525.0;
  double gmst0 = ((C3 * t + C2) * t + C1) * t + C0;
  gmst0 *= ORC_DPI / 86400.0;
  double fr = tjm - trunc(tjm); /* f64::fract */
  double h = fr * ORC_DPI;
  double g = gmst0 + h * RAP;
  long long i = (long long)floor(g / ORC_DPI);
  if (g < 0.0) i -= 1;
  g -= (double)i * ORC_DPI;
  return g;
}

/* observer_extension.rs:180-222 (position only): tt = epoch MJD TT, tut = the UT1 epoch as the reference's
 * `mjd_ut1.to_mjd_tai_days()`, dxbf = Earth-fixed observer position (AU). out = geocentric, ecliptic J2000 */
void orc_pvobs(double tt, double tut, const double dxbf[3], double out[3]) {
  double gast = orc_gmst(tut) + orc_equequ(tt);
  double rot[3][3], rot1[3][3], rotmat[3][3];
  orc_rotmt(-gast, 2, rot);
  orc_ref_system s1 = {1, 0, tt}, s2 = {2, 1, 0.0};
  orc_rotpn(&s1, &s2, rot1);
  transpose3(rot1, rot1);
  transpose3(rot, rot);
  matmul3(rot1, rot, rotmat);
  for (int i = 0; i < 3; ++i) out[i] = (rotmat[i][0] * dxbf[0] + rotmat[i][1] * dxbf[1]) + rotmat[i][2] * dxbf[2];
}

/* observer_extension.rs:224-237: earth (equatorial, AU) + ROT_ECLMJ2000_TO_EQUMJ2000 * geo */
void orc_helio_position(const double earth_equ[3], const double geo_ecl[3], double out[3]) {
  const double c = 9.174820620691818e-1, s = 3.977771559319137e-1;
  double r[3] = {(1.0 * geo_ecl[0] + 0.0 * geo_ecl[1]) + 0.0 * geo_ecl[2],
                 (0.0 * geo_ecl[0] + c * geo_ecl[1]) + (-s) * geo_ecl[2],
                 (0.0 * geo_ecl[0] + s * geo_ecl[1]) + c * geo_ecl[2]};
  for (int i = 0; i < 3; ++i) out[i] = earth_equ[i] + r[i];
}

/* the whole pre-pass for one observation: 0 ok / OUTFIT_ERR_EPHEM_RANGE */
int orc_observer_prepass(const outfit_ephem_t *eph, double tt, double tut, const double dxbf[3], double geo_ecl[3],
                         double helio_equ[3]) {
  double earth[3];
  orc_pvobs(tt, tut, dxbf, geo_ecl);
  int rc = orc_earth_ephemeris(eph, tt, earth);
  if (rc) return rc;
  orc_helio_position(earth, geo_ecl, helio_equ);
  return 0;
}

/* the pre-pass over a batch, in the layout of outfit_hip_observer_positions (the checker of that entry point) */
int orc_observer_positions(const outfit_ephem_t *eph, uint64_t n, const double *tt, const double *tut,
                           const double *dxbf, double *geo_ecl, double *helio_equ) {
  int rc = 0;
  for (uint64_t o = 0; o < n; ++o) {
    int r = orc_observer_prepass(eph, tt[o], tut[o], dxbf + 3 * o, geo_ecl + 3 * o, helio_equ + 3 * o);
    if (r) {
      rc = r;
      for (int k = 0; k < 3; ++k) helio_equ[3 * o + k] = NAN;
    }
  }
  return rc;
}
