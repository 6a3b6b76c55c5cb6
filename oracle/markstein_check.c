/* markstein_check.c -- TEST INFRASTRUCTURE (oracle side).
 * The HIP kernels replace the per-cell IEEE division c = j/i of the reference
 * (src/visibilityBasedSolver.cpp:590-591, 596-597 and the three other nests) by
 *     y = RN(1/i)   (host-computed table, one entry per step)
 *     q = RN(j*y);  r = fma(-i, q, j);  c = fma(r, y, q)
 * (Markstein's correction step).  This program proves, exhaustively over the
 * index range the kernels accept, that c is bit-identical to RN(j/i).
 * usage: markstein_check N   -> exit 0 iff all 1 <= j < i <= N match. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 4096;
  unsigned long long bad = 0, total = 0;
  for (long i = 1; i <= n; ++i) {
    volatile double di = (double)i;
    double y = 1.0 / di;
    for (long j = 0; j < i; ++j) {
      double dj = (double)j;
      double q = dj * y;
      double r = fma(-di, q, dj);
      double c = fma(r, y, q);
      double ref = dj / di;
      if (memcmp(&c, &ref, 8) != 0) {
        if (bad < 10) fprintf(stderr, "mismatch j=%ld i=%ld\n", j, i);
        ++bad;
      }
      ++total;
    }
  }
  printf("checked %llu pairs up to %ld: %llu mismatches\n", total, n, bad);
  return bad ? 1 : 0;
}
