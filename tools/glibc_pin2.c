/* Pin of oracle/glibcmath2.c against the platform libm (glibc 2.35, x86-64 with FMA): bit-for-bit comparison of
 * atan2 / tan / hypot on pseudo-random arguments. Build and run (CPU only, test infrastructure):
 *   gcc -O2 -mfma -ffp-contract=off -I include -I oracle tools/glibc_pin2.c oracle/glibcmath2.c -lm -lpthread -o /tmp/pin2
 *   /tmp/pin2 <millions of arguments per function> [threads]
 * Prints the number of differing results per function (must be 0) and the first few offenders. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
double orc_glibc_atan2(double, double);
double orc_glibc_tan(double);
double orc_glibc_hypot(double, double);
static uint64_t sm(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double u01(uint64_t *s) { return (double)(sm(s) >> 11) * 0x1p-53; }
static double rnd(uint64_t *s, int er) { uint64_t b = sm(s); int e = (int)(sm(s) % (uint64_t)(2 * er + 1)) - er; uint64_t v = (b & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 + e) << 52); double d; memcpy(&d, &v, 8); return d; }
static int same(double a, double b) { uint64_t x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8); return x == y || (a != a && b != b); }
typedef struct { uint64_t seed, n; uint64_t bad[3]; } job_t;
static void *work(void *arg) {
  job_t *j = (job_t *)arg; uint64_t s = j->seed;
  for (uint64_t i = 0; i < j->n; ++i) {
    double y, x; int m = (int)(i & 7);
    if (m < 3) { double r = exp((u01(&s) - 0.5) * 20.0), t = (u01(&s) - 0.5) * 6.283185307179586 * 1.000001; y = r * sin(t); x = r * cos(t); }
    else if (m == 3) { x = rnd(&s, 40); y = x * rnd(&s, 6); }                  /* all ratios around the table range */
    else if (m == 4) { x = rnd(&s, 3); y = rnd(&s, 3); }
    else if (m == 5) { x = rnd(&s, 300); y = rnd(&s, 300); }
    else if (m == 6) { x = rnd(&s, 60); y = x * rnd(&s, 62); }
    else { x = (u01(&s) - 0.5) * 4.0; y = (u01(&s) - 0.5) * 4.0; }
    double a = atan2(y, x), b = orc_glibc_atan2(y, x);
    if (!same(a, b)) { if (j->bad[0]++ < 3) fprintf(stderr, "atan2(%a, %a) = %a, restated %a\n", y, x, a, b); }
    double t;
    if (m < 3) t = (u01(&s) - 0.5) * 3.2; else if (m == 3) t = (u01(&s) - 0.5) * 50.0; else if (m == 4) t = rnd(&s, 30) * 0x1p-26;
    else if (m == 5) t = 1.5707963267948966 + rnd(&s, 10) * 0x1p-30; else if (m == 6) t = (u01(&s) - 0.5) * 0.2; else t = (double)((int)(u01(&s) * 32) - 16) * 1.5707963267948966 + rnd(&s, 20) * 0x1p-40;
    if (fabs(t) <= 25.0) { a = tan(t); b = orc_glibc_tan(t); if (!same(a, b)) { if (j->bad[1]++ < 3) fprintf(stderr, "tan(%a) = %a, restated %a\n", t, a, b); } }
    double p, q;
    if (m < 4) { p = rnd(&s, 4); q = rnd(&s, 4); } else if (m == 4) { p = rnd(&s, 1000); q = rnd(&s, 1000); } else if (m == 5) { p = rnd(&s, 30); q = p * rnd(&s, 56); }
    else if (m == 6) { p = rnd(&s, 520) ; q = p * rnd(&s, 3); } else { p = (u01(&s) - 0.5) * 10; q = (u01(&s) - 0.5) * 10; }
    a = hypot(p, q); b = orc_glibc_hypot(p, q);
    if (!same(a, b)) { if (j->bad[2]++ < 3) fprintf(stderr, "hypot(%a, %a) = %a, restated %a\n", p, q, a, b); }
  }
  return 0;
}
int main(int argc, char **argv) {
  uint64_t n = (argc > 1 ? strtoull(argv[1], 0, 10) : 10) * 1000000ull; int nt = argc > 2 ? atoi(argv[2]) : 8;
  pthread_t th[64]; job_t jobs[64];
  for (int t = 0; t < nt; ++t) { jobs[t] = (job_t){0x1234567ull * (t + 1) + 20263, n / nt, {0, 0, 0}}; pthread_create(&th[t], 0, work, &jobs[t]); }
  uint64_t bad[3] = {0, 0, 0};
  for (int t = 0; t < nt; ++t) { pthread_join(th[t], 0); for (int k = 0; k < 3; ++k) bad[k] += jobs[t].bad[k]; }
  printf("arguments per function %llu: differing atan2 %llu, tan %llu, hypot %llu\n", (unsigned long long)(n / nt * nt), (unsigned long long)bad[0], (unsigned long long)bad[1], (unsigned long long)bad[2]);
  return (bad[0] | bad[1] | bad[2]) != 0;
}
