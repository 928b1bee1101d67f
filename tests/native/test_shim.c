/* Links libmodsx like an application that used to link libdegensac and calls exp_ransacHcustom / exp_ransacFcustom with
 * the argument list of LORANSACFiltering (matching/matching.cpp:883, 891).  Plain C, no device needed.
 * Prints "OK <inliers H> <inliers F>"; the pytest wrapper compares the flags with modsx_ransac_h / modsx_ransac_f. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "modsx.h"
#include "modsx_degensac.h"

static unsigned lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static double uni(unsigned *s, double a, double b) { return a + (b - a) * (lcg(s) & 0xffff) / 65535.0; }

int main(void) {
  const int T = 400;
  const double Hgt[9] = {1.1, 0.05, 20, -0.03, 0.95, -10, 1e-5, 2e-5, 1};
  double *u = (double *)malloc(sizeof(double) * 6 * T);
  unsigned s = 7;
  for (int i = 0; i < T; i++) {
    const double x = uni(&s, 20, 1000), y = uni(&s, 20, 700);
    const double w = Hgt[6] * x + Hgt[7] * y + Hgt[8];
    double x2 = (Hgt[0] * x + Hgt[1] * y + Hgt[2]) / w + uni(&s, -0.7, 0.7), y2 = (Hgt[3] * x + Hgt[4] * y + Hgt[5]) / w + uni(&s, -0.7, 0.7);
    if (i % 3 == 0) { x2 = uni(&s, 20, 1000); y2 = uni(&s, 20, 700); }   /* outliers */
    u[6 * i] = x; u[6 * i + 1] = y; u[6 * i + 2] = 1; u[6 * i + 3] = x2; u[6 * i + 4] = y2; u[6 * i + 5] = 1;
  }
  /* --- H, exactly as matching.cpp:889-893 --- */
  int *data_out = (int *)malloc(T * 18 * sizeof(int));
  double *resids = NULL, H[9];
  unsigned char *inl = (unsigned char *)malloc(T), *inl2 = (unsigned char *)malloc(T);
  modsx_ransac_set_seed(5, 1);
  Score S = exp_ransacHcustom(u, T, 3.0 * 3.0, 0.99, 100000, H, inl, 4, data_out, 1, 0, &resids, &HDs, &HDsi, &HDsidx, 1);
  if (!resids) { printf("FAIL resids\n"); return 1; }
  free(resids);
  int d3[3]; double H2[9], J = 0;
  const int n2 = modsx_ransac_h(u, T, 9.0, 0.99, 100000, H2, inl2, d3, 1, 1, 5, &J);
  if ((int)S.I != n2 || memcmp(inl, inl2, T) || memcmp(H, H2, sizeof H) || S.J != J) { printf("FAIL H %u %d\n", S.I, n2); return 1; }
  /* a foreign error function must be refused, not silently replaced */
  Score Sbad = exp_ransacHcustom(u, T, 9.0, 0.99, 1000, H, inl2, 4, data_out, 1, 0, &resids, (HDsPtr)&FDs, &HDsi, &HDsidx, 1);
  free(resids);
  if (Sbad.I != 0) { printf("FAIL foreign pointer accepted\n"); return 1; }
  /* --- the errorType switch of LORANSACFiltering (matching.cpp:821-846) takes the address of every one of these; the
   * SYMM_SUM triple is the RANSACPars default (matching.hpp) and SYMM_MAX the other symmetric choice --- */
  {
    HDsPtr hds[3] = {&HDs, &HDsSymMax, &HDsSym};
    HDsiPtr hdsi[3] = {&HDsi, &HDsiSymMax, &HDsiSym};
    HDsidxPtr hdsidx[3] = {&HDsidx, &HDsSymidxMax, &HDsSymidx};
    for (int et = 1; et < 3; et++) {
      Score Se = exp_ransacHcustom(u, T, 9.0, 0.99, 100000, H, inl, 4, data_out, 1, 0, &resids, hds[et], hdsi[et], hdsidx[et], 1);
      free(resids);
      const int ne = modsx_ransac_h_errtype(u, T, 9.0, 0.99, 100000, H2, inl2, d3, 1, 1, et, 5, &J);
      if ((int)Se.I != ne || ne < T / 2 || memcmp(inl, inl2, T) || memcmp(H, H2, sizeof H)) { printf("FAIL H errtype %d: %u %d\n", et, Se.I, ne); return 1; }
      /* the subset forms agree with the full form */
      double pall[400], psub[3];
      int idx[3] = {1, 7, 399};
      hds[et](NULL, u, H, pall, T);
      hdsi[et](NULL, u, H, psub, T, idx, 3);
      if (psub[0] != pall[1] || psub[1] != pall[7] || psub[2] != pall[399]) { printf("FAIL HDsi errtype %d\n", et); return 1; }
      hdsidx[et](NULL, u, H, psub, T, idx, 3);
      if (psub[0] != pall[1] || psub[1] != pall[7] || psub[2] != pall[399]) { printf("FAIL HDsidx errtype %d\n", et); return 1; }
    }
    /* a mixed triple is refused */
    Score Smix = exp_ransacHcustom(u, T, 9.0, 0.99, 1000, H, inl2, 4, data_out, 1, 0, &resids, &HDsSym, &HDsi, &HDsidx, 1);
    free(resids);
    if (Smix.I != 0) { printf("FAIL mixed triple accepted\n"); return 1; }
  }
  /* --- F, exactly as matching.cpp:876-885 --- */
  double F[9], F2[9], HinF[9];
  int I_H = 0;
  const int nF = exp_ransacFcustom(u, T, 4.0 * 4.0, 0.99, 100000, F, inl, data_out, 1, 0, &resids, HinF, &I_H, &exFDs, &FDs, 1);
  free(resids);
  const int nF2 = modsx_ransac_f(u, T, 16.0, 0.99, 100000, 1, 0, 0, 1, 5, F2, inl2, d3);
  if (nF != nF2 || memcmp(inl, inl2, T) || memcmp(F, F2, sizeof F)) { printf("FAIL F %d %d\n", nF, nF2); return 1; }
  /* the exported Sampson error agrees with a direct evaluation for a perfect correspondence */
  double p1[1];
  const double uu[6] = {100, 200, 1, (Hgt[0] * 100 + Hgt[1] * 200 + Hgt[2]) / (Hgt[6] * 100 + Hgt[7] * 200 + 1),
                        (Hgt[3] * 100 + Hgt[4] * 200 + Hgt[5]) / (Hgt[6] * 100 + Hgt[7] * 200 + 1), 1};
  FDs(uu, F, p1, 1);
  printf("OK %u %d\n", S.I, nF);
  free(u); free(data_out); free(inl); free(inl2);
  return 0;
}
