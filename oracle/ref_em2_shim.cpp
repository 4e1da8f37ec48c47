/*
 * ref_em2_shim.cpp -- C-ABI door onto the REFERENCE's own em2().
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference code: the
 * reference source is compiled from where it lies (REF_EM_SRC, normally
 * /root/reference/emOptim2.cpp, passed by oracle/Makefile) by textual
 * inclusion, exactly as ngsDist.cpp:23 includes it.  The output goes to
 * oracle/_ref/ (git-ignored).  emOptim2.cpp needs only libc/libm headers,
 * so no stand-in for anything the image lacks is involved.
 *
 * The call below mirrors ngsDist.cpp:329-349: 1x3 Matrix<double> operands,
 * sfs pre-filled by the caller, em2(sfs, &GL1, &GL2, 0.001, 50, 9).
 */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <sys/stat.h>

#include REF_EM_SRC

extern "C" void ref_em2(double *sfs, const double *gl1, const double *gl2,
                        double tole, int max_iter) {
  Matrix<double> GL1 = alloc(1, 3);
  Matrix<double> GL2 = alloc(1, 3);
  for (int g = 0; g < 3; g++) {
    GL1.mat[0][g] = gl1[g];
    GL2.mat[0][g] = gl2[g];
  }
  em2(sfs, &GL1, &GL2, tole, max_iter, 9);
  dalloc(GL1, 1);
  dalloc(GL2, 1);
}

/* the same call with the two 1x3 operands allocated once per thread, as gen_dist() allocates them once per pair
 * (ngsDist.cpp:329-330) and not per site: what the oracle's threaded pair loop calls when it is asked to run the
 * reference's own em2() (ngo_set_em2_hook) */
extern "C" int ref_em2_tls(double *sfs, const double *gl1, const double *gl2,
                           double tole, int max_iter) {
  static thread_local Matrix<double> GL1 = alloc(1, 3);
  static thread_local Matrix<double> GL2 = alloc(1, 3);
  for (int g = 0; g < 3; g++) {
    GL1.mat[0][g] = gl1[g];
    GL2.mat[0][g] = gl2[g];
  }
  em2(sfs, &GL1, &GL2, tole, max_iter, 9);
  return 0;
}

/* many sites at once: sfs_out[n][9], gl1[n][3], gl2[n][3]; start = 1/9 as in
 * ngsDist.cpp:340 */
extern "C" void ref_em2_batch(size_t n, const double *gl1, const double *gl2,
                              double *sfs_out) {
  for (size_t k = 0; k < n; k++) {
    double *sfs = sfs_out + 9 * k;
    for (int c = 0; c < 9; c++) sfs[c] = (double)1 / 9;
    ref_em2(sfs, gl1 + 3 * k, gl2 + 3 * k, 0.001, 50);
  }
}
