/* Pin of oracle/glibcmath3.c against the platform libm (glibc 2.35, x86-64 with FMA): bit-for-bit comparison of acos and log.
 *   gcc -O2 -mfma -ffp-contract=off -I include -I oracle tools/glibc_pin3.c oracle/glibcmath3.c -lm -lpthread -o /tmp/pin3
 *   /tmp/pin3 <millions of arguments> [threads]
 * acos arguments: uniform in [-1, 1], dense near +-1 (1 - 2^-e), near every interval boundary, tiny, and out of range;
 * log arguments: 1 .. 31, e^+-700, around 1 (the polynomial branch and its two ends), any positive bit pattern, anything. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
double orc_glibc_acos(double);
double orc_glibc_log(double);
static uint64_t sm(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double u01(uint64_t *s) { return (double)(sm(s) >> 11) * 0x1p-53; }
static int same(double a, double b) { uint64_t x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8); return x == y || (a != a && b != b); }
typedef struct { uint64_t seed, n, bad, bad_log; } job_t;
static const double edge[] = {0.125, 0.25, 0.5, 0.75, 0.921875, 0.953125, 0.96875, 1.0, 0x1p-55, 0.0};
static void *work(void *arg) {
  job_t *j = (job_t *)arg; uint64_t s = j->seed;
  for (uint64_t i = 0; i < j->n; ++i) {
    double x; int m = (int)(i & 7);
    if (m < 4) x = 2.0 * u01(&s) - 1.0;
    else if (m == 4) { x = 1.0 - ldexp(u01(&s), -(int)(sm(&s) % 54)); if (sm(&s) & 1) x = -x; }
    else if (m == 5) { uint64_t b; double e = edge[sm(&s) % 10]; memcpy(&b, &e, 8); b += (int64_t)(sm(&s) % 2001) - 1000; memcpy(&x, &b, 8); if (sm(&s) & 1) x = -x; }
    else if (m == 6) { x = ldexp(u01(&s) - 0.5, -(int)(sm(&s) % 70)); }
    else { uint64_t b = sm(&s); memcpy(&x, &b, 8); }   /* anything: out of range, NaN, infinities */
    double a = acos(x), b = orc_glibc_acos(x);
    if (!same(a, b)) { if (j->bad++ < 4) fprintf(stderr, "acos(%a) = %a, restated %a\n", x, a, b); }
    if (m < 2) x = 1.0 + u01(&s) * 30.0;
    else if (m == 2) x = exp((u01(&s) - 0.5) * 1400.0);
    else if (m == 3) x = 1.0 + (u01(&s) - 0.5) * 0.14;
    else if (m == 4) x = 1.0 + ldexp(u01(&s) - 0.5, -(int)(sm(&s) % 54));
    else if (m == 5) { uint64_t v = sm(&s) & 0x7fffffffffffffffull; memcpy(&x, &v, 8); }
    else if (m == 6) { uint64_t v; double e = (sm(&s) & 1) ? 0x1.ep-1 : 0x1.109p+0; memcpy(&v, &e, 8); v += (int64_t)(sm(&s) % 2001) - 1000; memcpy(&x, &v, 8); }
    else { uint64_t v = sm(&s); memcpy(&x, &v, 8); }
    a = log(x); b = orc_glibc_log(x);
    if (!same(a, b)) { if (j->bad_log++ < 4) fprintf(stderr, "log(%a) = %a, restated %a\n", x, a, b); }
  }
  return 0;
}
int main(int argc, char **argv) {
  uint64_t n = (argc > 1 ? strtoull(argv[1], 0, 10) : 10) * 1000000ull; int nt = argc > 2 ? atoi(argv[2]) : 8;
  pthread_t th[64]; job_t jobs[64];
  for (int t = 0; t < nt; ++t) { jobs[t] = (job_t){0x7654321ull * (t + 1) + 20263, n / nt, 0, 0}; pthread_create(&th[t], 0, work, &jobs[t]); }
  uint64_t bad = 0, bad_log = 0;
  for (int t = 0; t < nt; ++t) { pthread_join(th[t], 0); bad += jobs[t].bad; bad_log += jobs[t].bad_log; }
  printf("arguments per function %llu: differing acos %llu, log %llu\n", (unsigned long long)(n / nt * nt), (unsigned long long)bad, (unsigned long long)bad_log);
  return (bad | bad_log) != 0;
}
