/* Check of the 3-instruction constant division used by the step kernel (dc-rl_amd/csrc/sdc_device.hpp SDC_DIV_CONST):
 * q = RN(x * RN(1/C)); r = fma(-q, C, x); result = fma(r, RN(1/C), q)  must equal the IEEE quotient x / C.
 * usage: div_const_check <samples per constant>; prints the total number of mismatches. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double mk(double x, double c, double rc) { double q = x * rc; double r = fma(-q, c, x); return fma(r, rc, q); }
int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 1000000L;
  /* every constant divisor of sdc_dynamics.hip / sdc_device.hpp */
  const double cs[] = {100.0, 24.0, 20.0, 1e3, 60.0, 2.778, 3.0, 1e4, 1e8, 6.0, 14.0, 17.0, 0.05,
                       /* run-time divisors of the shipped configs: history / queue capacity, rack counts */
                       1e4, 1e3, 16.0, 20.0, 25.0, 257.0,
                       /* sdc_features.hip slope_of: the sums of squared abscissae of the 4-, 6-, 14- and 17-point least-squares fits */
                       5.0, 17.5, 227.5, 408.0};
  long bad = 0;
  for (unsigned k = 0; k < sizeof(cs) / sizeof(cs[0]); k++) {
    const double c = cs[k], rc = 1.0 / c;
    for (long i = 0; i < n; i++) {
      uint64_t bits = (rnd() & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 40 + (rnd() % 80)) << 52);
      if ((i & 7) == 0) bits |= 0x000FFFFFFFFFF000ull;  /* significands near all-ones */
      if ((i & 7) == 1) bits &= ~0x000FFFFFFFFFF000ull; /* near a power of two */
      double x;
      memcpy(&x, &bits, 8);
      if (mk(x, c, rc) != x / c) bad++;
    }
  }
  printf("%ld\n", bad);
  return bad != 0;
}
