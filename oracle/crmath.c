/*
 * crmath.c — correctly-rounded sin / cos / acos for the oracle.
 *
 * The reference calls f64::sin/cos/acos, i.e. the platform libm (glibc on the
 * machines its KATs were recorded on), whose results are correctly rounded for
 * all but ~0.1 % of arguments. The Gauss correction loop is chaotically
 * sensitive to a 1-ulp change of the direction cosines or of the Kepler start
 * value, so a libm-dependent oracle could not be compared
 * bit for bit with a GPU that has a different libm (ROCm ocml). The oracle
 * therefore restates these three functions as CORRECTLY ROUNDED — the value an
 * ideal libm returns — using double-double arithmetic; the device code does
 * the same (outfit_amd/csrc/device/crmath.hpp, written independently against
 * the same constants from tools/gen_crmath_constants.py). The reference's
 * bit-exact KATs still pass with them (tests/test_oracle_kats.py).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include <math.h>
#include "oracle.h"

int orc_use_libm = 0; /* 1 = plain libm sin/cos/acos (platform-dependent), 0 = correctly rounded */
/* study aid: per-function platform-libm selection when orc_use_libm == 0
 * (1 sin/cos, 2 acos, 4 sinh/cosh, 8 log, 16 atan2, 32 Aberth start directions) */
int orc_libm_mask = 0;
#define LIBM(bit) (orc_use_libm || (orc_libm_mask & (bit)))
/* 1 = sin, cos, sinh, cosh as glibc 2.35 evaluates them (glibcmath.c: restated, no libm call), everything else
 * as in the default mode — what the device's OUTFIT_MATH_GLIBC mode computes. orc_use_libm = 1 overrides it. */
int orc_math_glibc = 0;

#define CRM_PIO2_P1 1.5707963267341256     /* 0x1.921fb54400000p+0  */
#define CRM_PIO2_P2 6.077100506303966e-11  /* 0x1.0b4611a600000p-34 */
#define CRM_PIO2_P3 2.0222662487111665e-21 /* 0x1.3198a2e000000p-69 */
#define CRM_PIO2_P4 8.4784276603689e-32    /* 0x1.b839a252049c1p-104 */
#define CRM_TWO_OVER_PI 0.6366197723675814
#define CRM_PI_HI 3.141592653589793

static const double CRM_INV_FACT[32][2] = {
  {1.0, 0.0},
  {1.0, 0.0},
  {0.5, 0.0},
  {0.16666666666666666, 9.25185853854297e-18},
  {0.041666666666666664, 2.3129646346357427e-18},
  {0.008333333333333333, 1.1564823173178714e-19},
  {0.001388888888888889, -5.300543954373577e-20},
  {0.0001984126984126984, 1.7209558293420705e-22},
  {2.48015873015873e-05, 2.1511947866775882e-23},
  {2.7557319223985893e-06, -1.858393274046472e-22},
  {2.755731922398589e-07, 2.3767714622250297e-23},
  {2.505210838544172e-08, -1.448814070935912e-24},
  {2.08767569878681e-09, -1.20734505911326e-25},
  {1.6059043836821613e-10, 1.2585294588752098e-26},
  {1.1470745597729725e-11, 2.0655512752830745e-28},
  {7.647163731819816e-13, 7.03872877733453e-30},
  {4.779477332387385e-14, 4.399205485834081e-31},
  {2.8114572543455206e-15, 1.6508842730861433e-31},
  {1.5619206968586225e-16, 1.1910679660273754e-32},
  {8.22063524662433e-18, 2.2141894119604265e-34},
  {4.110317623312165e-19, 1.4412973378659527e-36},
  {1.9572941063391263e-20, -1.3643503830087908e-36},
  {8.896791392450574e-22, -7.911402614872376e-38},
  {3.868170170630684e-23, -8.843177655482344e-40},
  {1.6117375710961184e-24, -3.6846573564509766e-41},
  {6.446950284384474e-26, -1.9330404233703465e-42},
  {2.4795962632247976e-27, -1.2953730964765229e-43},
  {9.183689863795546e-29, 1.4303150396787322e-45},
  {3.279889237069838e-30, 1.5117542744029879e-46},
  {1.1309962886447716e-31, 1.0498015412959506e-47},
  {3.7699876288159054e-33, 2.5870347832750324e-49},
  {1.216125041553518e-34, 5.586290567888806e-51},
};

/* theta_k = fma(2pi/8, k, (pi/2)/8): (cos, sin), correctly rounded */
const double orc_aberth_cos_sin[8][2] = {
  {0.9807852804032304, 0.19509032201612825},
  {0.5555702330196023, 0.8314696123025452},
  {-0.1950903220161282, 0.9807852804032304},
  {-0.8314696123025453, 0.5555702330196022},
  {-0.9807852804032304, -0.19509032201612836},
  {-0.5555702330196022, -0.8314696123025452},
  {0.1950903220161283, -0.9807852804032304},
  {0.8314696123025452, -0.5555702330196022},
};

typedef struct { double hi, lo; } dd;

static inline dd two_sum(double a, double b) {
  double s = a + b, bb = s - a;
  return (dd){s, (a - (s - bb)) + (b - bb)};
}
static inline dd fast_two_sum(double a, double b) { /* |a| >= |b| */
  double s = a + b;
  return (dd){s, b - (s - a)};
}
static inline dd two_prod(double a, double b) {
  double p = a * b;
  return (dd){p, fma(a, b, -p)};
}
static inline dd dd_add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi), t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = fast_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return fast_two_sum(s.hi, s.lo);
}
static inline dd dd_add_d(dd a, double b) {
  dd s = two_sum(a.hi, b);
  s.lo += a.lo;
  return fast_two_sum(s.hi, s.lo);
}
static inline dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return fast_two_sum(p.hi, p.lo);
}
static inline dd dd_neg(dd a) { return (dd){-a.hi, -a.lo}; }
static inline dd dd_div(dd a, dd b) {
  double q1 = a.hi / b.hi;
  dd r = dd_add(a, dd_neg(dd_mul(b, (dd){q1, 0.0})));
  double q2 = r.hi / b.hi;
  r = dd_add(r, dd_neg(dd_mul(b, (dd){q2, 0.0})));
  double q3 = r.hi / b.hi;
  dd q = fast_two_sum(q1, q2);
  return dd_add_d(q, q3);
}

/* r = x - k*pi/2 as a double-double, |r| <= pi/4 (+ tiny); valid for |x| < 2^20 */
static dd reduce_pio2(double x, int *quadrant) {
  double k = nearbyint(x * CRM_TWO_OVER_PI);
  *quadrant = (int)((long long)k & 3);
  if (k == 0.0) return (dd){x, 0.0};
  double t = x - k * CRM_PIO2_P1;          /* exact: k*P1 is exact (33+20 bits) and cancels */
  dd a = two_sum(t, -(k * CRM_PIO2_P2));   /* k*P2 exact */
  dd b = two_sum(a.hi, -(k * CRM_PIO2_P3)); /* k*P3 exact */
  double lo = (a.lo + b.lo) - k * CRM_PIO2_P4;
  return two_sum(b.hi, lo);
}

/* sin and cos of a reduced double-double argument, |r| <= pi/4: Taylor series, the
 * tail (terms below 2^-53 of the sum) in double, the head in double-double */
static void sincos_kernel(dd r, dd *s, dd *c) {
  dd z = dd_mul(r, r);
  double zh = z.hi;
  /* sin: r * sum_{j=0}^{14} (-1)^j z^j / (2j+1)! */
  double ts = 0.0;
  for (int j = 14; j >= 8; --j) {
    double cj = CRM_INV_FACT[2 * j + 1][0];
    ts = ts * zh + ((j & 1) ? -cj : cj);
  }
  dd p = {ts, 0.0};
  for (int j = 7; j >= 0; --j) {
    dd cj = {CRM_INV_FACT[2 * j + 1][0], CRM_INV_FACT[2 * j + 1][1]};
    if (j & 1) cj = dd_neg(cj);
    p = dd_add(dd_mul(p, z), cj);
  }
  *s = dd_mul(r, p);
  /* cos: sum_{j=0}^{15} (-1)^j z^j / (2j)! */
  double tc = 0.0;
  for (int j = 15; j >= 9; --j) {
    double cj = CRM_INV_FACT[2 * j][0];
    tc = tc * zh + ((j & 1) ? -cj : cj);
  }
  dd q = {tc, 0.0};
  for (int j = 8; j >= 0; --j) {
    dd cj = {CRM_INV_FACT[2 * j][0], CRM_INV_FACT[2 * j][1]};
    if (j & 1) cj = dd_neg(cj);
    q = dd_add(dd_mul(q, z), cj);
  }
  *c = q;
}

static void sincos_dd(double x, dd *s, dd *c) {
  int q;
  dd r = reduce_pio2(x, &q);
  dd sr, cr;
  sincos_kernel(r, &sr, &cr);
  switch (q) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = dd_neg(sr); break;
    case 2: *s = dd_neg(sr); *c = dd_neg(cr); break;
    default: *s = dd_neg(cr); *c = sr; break;
  }
}

/* the correctly-rounded pair, whatever the math mode (the differential corrector uses it in every mode but libm) */
static void sincos_cr(double x, double *s, double *c) {
  if (!(fabs(x) < 1048576.0)) { *s = sin(x); *c = cos(x); return; }
  if (x == 0.0) { *s = x; *c = 1.0; return; }
  dd sd, cd;
  sincos_dd(x, &sd, &cd);
  *s = sd.hi;
  *c = cd.hi;
}
/* PAIRED sites — x.sin() and x.cos() of the same operand in one basic block of the reference: rustc/LLVM emits ONE
 * call of sincos() there on *-linux-gnu (glibcmath.c explains; glibc's sincos and its standalone sin / cos return
 * different last bits for 0.07 % of arguments). The platform-libm mode therefore calls sincos() explicitly (it used
 * to write sin(x); cos(x) and relied on gcc merging them), the restated mode the sincos-equivalent restatement. */
static void (*volatile libm_sincos)(double, double *, double *) = sincos; /* volatile: gcc turns a sincos() whose
                                          other result is unused back into a standalone sin() / cos() call */
void orc_sincos(double x, double *s, double *c) {
  if (LIBM(1)) { libm_sincos(x, s, c); return; }
  if (orc_math_glibc && fabs(x) < 1048576.0) { *s = orc_glibc_sin(x); *c = orc_glibc_cos(x); return; }
  sincos_cr(x, s, c);
}
/* ISOLATED sites — a sin or a cos whose partner is not evaluated on the same operand (prelim_elliptic.rs:97): the
 * standalone libm entry point, which on FMA hardware is glibc's contracted variant (orc_glibc_sin1 / cos1). The
 * calls go through volatile pointers so that this file's own compiler cannot pair them. */
static double (*volatile libm_sin)(double) = sin;
static double (*volatile libm_cos)(double) = cos;
double orc_sin1(double x) {
  if (LIBM(1)) return libm_sin(x);
  if (orc_math_glibc && fabs(x) < 1048576.0) return orc_glibc_sin1(x);
  double s, c; sincos_cr(x, &s, &c); return s;
}
double orc_cos1(double x) {
  if (LIBM(1)) return libm_cos(x);
  if (orc_math_glibc && fabs(x) < 1048576.0) return orc_glibc_cos1(x);
  double s, c; sincos_cr(x, &s, &c); return c;
}
/* The Earth-rotation pre-pass (earthrot.c, row n2) evaluates its ~170 sin/cos per observation correctly rounded in
 * BOTH math modes, like the corrector: its nutation series is a table summed in another order than the reference's
 * hand-expanded text (agreement 1e-15, not bits), so glibc's last places would pin nothing there; the device does the
 * same (device/earthrot.hpp). orc_use_libm = 1 still selects the platform libm. */
void orc_sincos_prepass(double x, double *s, double *c) {
  if (LIBM(1)) { *s = sin(x); *c = cos(x); return; }
  sincos_cr(x, s, c);
}
double orc_sin(double x) { double s, c; orc_sincos(x, &s, &c); return s; }
double orc_cos(double x) { double s, c; orc_sincos(x, &s, &c); return c; }

/* acos by one Newton step in double-double from the libm seed: y1 = y0 + (cos y0 - x)/sin y0 */
static int scoped_cr(void);
double orc_acos(double x) {
  if (LIBM(2) || !(fabs(x) <= 1.0)) return acos(x);
  if (orc_math_glibc && !scoped_cr()) return orc_glibc_acos(x);  /* glibc 2.35's acos, restated (glibcmath3.c) */
  if (x == 1.0) return 0.0;
  if (x == -1.0) return CRM_PI_HI;
  double y0 = acos(x);
  dd sd, cd;
  sincos_dd(y0, &sd, &cd);
  dd num = dd_add_d(cd, -x);
  dd q = dd_div(num, sd);
  dd y1 = dd_add_d(q, y0);
  return y1.hi;
}

/* ---- sinh / cosh / log of the hyperbolic cold start (prelim_hyperbolic.rs:45-141), correctly rounded ----
 * Its value seeds the same chaotically sensitive correction loop as the elliptic start (measured: glibc's
 * 1-ulp sinh/cosh against correctly-rounded ones moved a = 14 AU by 1.6e-9 on 1 object in 30 000), so it gets
 * the same treatment as sin/cos: the correctly rounded value. Here through binary128 (libquadmath, error
 * < 1 ulp of 113 bits, then ONE rounding to double: a wrong last bit needs the exact value within 2^-113
 * relative of a rounding boundary) — an implementation independent of the device's double-double one
 * (outfit_amd/csrc/device/crmath.hpp), both pinned by tests/golden/crmath_hyp_vectors.json (mpmath). */
#include <quadmath.h>
double orc_sinh(double x) { return LIBM(4) ? sinh(x) : orc_math_glibc ? orc_glibc_sinh(x) : (double)sinhq((__float128)x); }
double orc_cosh(double x) { return LIBM(4) ? cosh(x) : orc_math_glibc ? orc_glibc_cosh(x) : (double)coshq((__float128)x); }
double orc_log(double x) { return LIBM(8) ? log(x) : (orc_math_glibc && !scoped_cr()) ? orc_glibc_log(x) : (double)logq((__float128)x); }
/* atan2 of the state -> elements conversion (orb_elem.rs:58-226): a last-place difference of i or Omega is
 * amplified ~a/r in a near-parabolic 1/a and ~1/e in omega, M of a near-circular orbit. */
double orc_atan2(double y, double x) {
  return LIBM(16) ? atan2(y, x) : orc_math_glibc ? orc_glibc_atan2(y, x) : (double)atan2q((__float128)y, (__float128)x);
}

/* ---- scoped selection: the differential corrector evaluates ALL its elementary functions correctly rounded
 * (its ill-conditioned normal equations amplify a last place up to 1e13 times; the device does the same and is
 * then bit-identical), the per-observation residuals of the IOD's RMS stage keep the platform libm (2x10^6
 * evaluations per 1000 objects; they move the RMS by <= 1e-9 relative, never the elements). orc_cr_scope is
 * raised around the corrector (diffcor.c); orc_use_libm = 1 overrides everything to the platform libm. */
__thread int orc_cr_scope = 0;
static int scoped_cr(void) { return !orc_use_libm && orc_cr_scope > 0; }
/* Outside the corrector's scope — the RMS stage of the IOD (ccek1's caller chain, to_equinoctial, propagate_twobody,
 * ephemeris_error) — the reference's platform libm is what counts: in the restated-glibc mode these are glibc 2.35's
 * functions restated (glibcmath.c, glibcmath2.c: the `rms` field and the selection then follow a glibc run bit for
 * bit without calling glibc); in the default mode and with orc_use_libm the platform's. */
static int restated_glibc(void) { return orc_math_glibc && !orc_use_libm; }
/* paired sites of the RMS stage (to_equinoctial, the Kepler-equation closures, the position formulas): sincos */
static void sincos_rms(double x, double *s, double *c) {
  if (restated_glibc() && fabs(x) < 1048576.0) { *s = orc_glibc_sin(x); *c = orc_glibc_cos(x); return; }
  libm_sincos(x, s, c);
}
double orc_sin_s(double x) { double s, c; if (scoped_cr()) sincos_cr(x, &s, &c); else sincos_rms(x, &s, &c); return s; }
double orc_cos_s(double x) { double s, c; if (scoped_cr()) sincos_cr(x, &s, &c); else sincos_rms(x, &s, &c); return c; }
/* the one isolated cosine of the RMS stage: cos(dec_obs) in ephemeris_error (observation_ephemeris.rs:412) */
double orc_cos1_s(double x) {
  double s, c;
  if (scoped_cr()) { sincos_cr(x, &s, &c); return c; }
  return restated_glibc() && fabs(x) < 1048576.0 ? orc_glibc_cos1(x) : libm_cos(x);
}
double orc_tan_s(double x) { return scoped_cr() ? (double)tanq((__float128)x) : restated_glibc() ? orc_glibc_tan(x) : tan(x); }
double orc_atan2_s(double y, double x) {
  return scoped_cr() ? (double)atan2q((__float128)y, (__float128)x) : restated_glibc() ? orc_glibc_atan2(y, x) : atan2(y, x);
}
double orc_hypot_s(double x, double y) {
  return scoped_cr() ? (double)hypotq((__float128)x, (__float128)y) : restated_glibc() ? orc_glibc_hypot(x, y) : hypot(x, y);
}
