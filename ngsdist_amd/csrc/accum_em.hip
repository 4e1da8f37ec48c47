// accum_em.hip -- gen_dist() without --indep_geno: per (pair, site) the 9-cell
// joint-genotype EM of the reference's emOptim2.cpp (em2 :112-135, emStep2
// :91-109, lik2 :77-89, normalize :69-75; called from ngsDist.cpp:340-353 with
// one site, start 1/9, tole 0.001, maxIter 50), then the score-weighted sum.
//
// Workgroup = 256 threads = one 16x16 tile of pairs x one slice of sites; each
// thread owns one pair and walks the slice's sites in order, so the per-slice
// sum is a plain sequential accumulation like the reference's.  The operand
// reads are the fragment-major image's 128-byte runs (16 consecutive
// individuals of one (site, genotype)); the path is FP64-VALU bound.
//
// Two device forms of the same EM:
//  * faithful: every multiply/add/divide of emStep2/normalize/lik2 in the
//    reference's order (IEEE double, no contraction), so the iterates are
//    bit-identical to the CPU's; only log() is the device's (<= 1 ulp), which
//    enters the stopping test alone.
//  * fast: the single-site EM has the closed-form iterate sfs_t = a^t / S_t with
//    a_k = GL1[x]*GL2[y] and S_t = SUM_k a_k^t, and lik_t = log(S_{t+1}/S_t); the
//    stopping rule |lik_t - lik_{t-1}| < tole becomes S_{t+1}*S_{t-1} < e^tole * S_t^2
//    (the likelihood is non-decreasing).  No divide and no log per iteration;
//    one divide per site.  Agrees with the faithful form to ~1e-14 relative.
#include "ngd_internal.h"

namespace {

constexpr double TOLE = 0.001;  // ngsDist.cpp:349
constexpr int MAX_ITER = 50;    // ngsDist.cpp:349

__device__ __forceinline__ double lik2(const double *sfs, const double *g1, const double *g2) {
  double tmp = 0;
#pragma unroll
  for (int x = 0; x < 3; x++)
#pragma unroll
    for (int y = 0; y < 3; y++) tmp += sfs[3 * x + y] * g1[x] * g2[y];
  return 0 + log(tmp);
}

__device__ __forceinline__ void normalize9(double *t) {
  double s = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) s += t[i];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] /= s;
}

// returns SUM_k score_k * sfs_k accumulated INTO acc in the reference's order
__device__ __forceinline__ double site_faithful(const double *g1, const double *g2, const ngd_score &sc,
                                                double acc, double w, bool weighted) {
  double sfs[9];
#pragma unroll
  for (int k = 0; k < 9; k++) sfs[k] = (double)1 / 9;
  double oldLik = lik2(sfs, g1, g2);
  for (int it = 0; it < MAX_ITER; it++) {
    double inner[9];
#pragma unroll
    for (int x = 0; x < 3; x++)
#pragma unroll
      for (int y = 0; y < 3; y++) inner[3 * x + y] = sfs[3 * x + y] * g1[x] * g2[y];
    normalize9(inner);
#pragma unroll
    for (int k = 0; k < 9; k++) sfs[k] = 0.0 + inner[k];
    normalize9(sfs);
    double lik = lik2(sfs, g1, g2);
    bool stop = fabs(lik - oldLik) < TOLE;
    oldLik = lik;
    if (stop) break;
  }
  if (weighted) {
    double c = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) c += sc.v[k] * sfs[k];
    return acc + c * w;
  }
#pragma unroll
  for (int k = 0; k < 9; k++) acc += sc.v[k] * sfs[k];
  return acc;
}

// fast form.  a_k = g1[x]*g2[y] is rank one, so a_k^t = g1[x]^t * g2[y]^t and
//   S_t = SUM_k a_k^t = (SUM_x g1[x]^t) * (SUM_y g2[y]^t) = A_t * B_t :
// six running powers instead of nine, and the score-weighted sum at the end is
// u1' * score * u2 / (A_T * B_T) with u = g^T.
__device__ __forceinline__ double site_fast(const double *g1, const double *g2, const ngd_score &sc,
                                            double acc, double w) {
  const double E = 0x1.0041919b7ee34p+0;  // exp(0.001), the tolerance of ngsDist.cpp:349
  double u1[3] = {g1[0], g1[1], g1[2]}, u2[3] = {g2[0], g2[1], g2[2]};  // g^t      (t odd)
  double n1[3], n2[3];                                                  // g^(t+1)
  double Sm = 9.0;                                                     // S_{t-1}; S_0 = 9 (uniform start)
  double Sc = ((u1[0] + u1[1]) + u1[2]) * ((u2[0] + u2[1]) + u2[2]);   // S_t;     S_1
  double Sn = 0;                                                       // S_{t+1}
  // |lik_t - lik_{t-1}| < tole  <=>  S_{t+1} S_{t-1} < e^tole S_t^2   (lik is non-decreasing).
  // Two steps per trip, the powers alternating between u and n, so that nothing is copied per step; a
  // lane that stops is masked off and keeps its registers, and `even` says which of the two holds g^T.
  int even = 0;
  for (int t = 1;; t += 2) {
#pragma unroll
    for (int x = 0; x < 3; x++) { n1[x] = u1[x] * g1[x]; n2[x] = u2[x] * g2[x]; }
    Sn = ((n1[0] + n1[1]) + n1[2]) * ((n2[0] + n2[1]) + n2[2]);
    if (Sn * Sm < E * (Sc * Sc)) break;  // T = t (odd, so never MAX_ITER): g^T = u, S_T = Sc
#pragma unroll
    for (int x = 0; x < 3; x++) { u1[x] = n1[x] * g1[x]; u2[x] = n2[x] * g2[x]; }
    const double Su = ((u1[0] + u1[1]) + u1[2]) * ((u2[0] + u2[1]) + u2[2]);  // S_{t+2}
    if (Su * Sc < E * (Sn * Sn) || t + 1 == MAX_ITER) { even = 1; break; }  // T = t+1: g^T = n, S_T = Sn
    Sm = Sn;
    Sc = Su;
  }
  // g^T and S_T are selected AFTER the loop, through a flag the compiler cannot trace back to the loop exits
  // (the empty asm): left to itself it turns the selection into register copies at both exits, i.e. into
  // every trip of the loop.  One scoring and one divide per site instead of two.
  asm volatile("" : "+v"(even));
  double a[3], b[3];
#pragma unroll
  for (int x = 0; x < 3; x++) { a[x] = even ? n1[x] : u1[x]; b[x] = even ? n2[x] : u2[x]; }
  const double S = even ? Sn : Sc;
  // score-weighted sum of the 9 cells, fused multiply-adds (this form promises 1e-9, not the reference's bits)
  double q[3];
#pragma unroll
  for (int x = 0; x < 3; x++)
    q[x] = __builtin_fma(sc.v[3 * x + 2], b[2], __builtin_fma(sc.v[3 * x + 1], b[1], sc.v[3 * x] * b[0]));
  const double c = __builtin_fma(a[2], q[2], __builtin_fma(a[1], q[1], a[0] * q[0]));
  return acc + (c / S) * w;
}

template <bool FAST, bool WEIGHTED, bool PDEL>
__global__ __launch_bounds__(256) void k_accum_em(const double *__restrict__ PA,
                                                   const uint32_t *__restrict__ ws, ngd_score sc,
                                                   const ngd_tile *__restrict__ tiles, uint32_t n_tiles,
                                                   uint32_t n_ig, uint32_t n_pad, uint64_t n_ind,
                                                   uint64_t n_sites_eff, uint64_t sites_per_slice,
                                                   double *__restrict__ slab) {
  const uint32_t tile = blockIdx.x % n_tiles;
  const uint32_t ks = blockIdx.x / n_tiles;
  const uint32_t ig = tiles[tile].ti, jg = tiles[tile].tj;
  const uint32_t i = ig * 16 + (threadIdx.x >> 4);
  const uint32_t j = jg * 16 + (threadIdx.x & 15);
  const bool valid = i < j && j < n_ind;
  const uint64_t s0 = (uint64_t)ks * sites_per_slice;
  uint64_t s1 = s0 + sites_per_slice;
  if (s1 > n_sites_eff) s1 = n_sites_eff;

  const uint64_t kstride = (uint64_t)n_ig * 64;  // doubles between consecutive k-groups
  double acc = 0;
  if (valid) {
    const double *pi = PA + (uint64_t)ig * 64 + (i & 15);
    const double *pj = PA + (uint64_t)jg * 64 + (j & 15);
    for (uint64_t s = s0; s < s1; s++) {
      double w = 1.0;
      if (WEIGHTED) {
        const uint32_t m = ws[s];
        if (m == 0) continue;  // site not drawn in this replicate (uniform across the workgroup)
        w = (double)m;
      }
      double g1[3], g2[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const uint64_t k = 3 * s + c;
        const uint64_t off = (k >> 2) * kstride + (k & 3) * 16;
        g1[c] = pi[off];
        g2[c] = pj[off];
      }
      if (PDEL && (ngd_miss(g1[0], g1[1], g1[2]) || ngd_miss(g2[0], g2[1], g2[2]))) continue;
      if (FAST) acc = site_fast(g1, g2, sc, acc, w);
      else acc = site_faithful(g1, g2, sc, acc, w, WEIGHTED);
    }
  }
  slab[((uint64_t)ks * n_pad + i) * n_pad + j] = acc;
}

// RB bootstrap replicates in ONE pass: the EM of a (pair, site) does not depend on the replicate, only
// its weight does, so the site's contribution c is computed once and added RB times,
//   acc[r] = acc[r] + c * W[s][r]          (the arithmetic of the one-replicate kernel: same bits),
// W site-major so that the RB weights of a site are one scalar load.  The slab holds RB planes per slice.
template <bool FAST, bool PDEL, int RB>
__global__ __launch_bounds__(256) void k_accum_em_batch(const double *__restrict__ PA, const double *__restrict__ Wb,
                                                         ngd_score sc, const ngd_tile *__restrict__ tiles,
                                                         uint32_t n_tiles, uint32_t n_ig, uint32_t n_pad,
                                                         uint64_t n_ind, uint64_t n_sites_eff,
                                                         uint64_t sites_per_slice, double *__restrict__ slab) {
  const uint32_t tile = blockIdx.x % n_tiles;
  const uint32_t ks = blockIdx.x / n_tiles;
  const uint32_t ig = tiles[tile].ti, jg = tiles[tile].tj;
  const uint32_t i = ig * 16 + (threadIdx.x >> 4);
  const uint32_t j = jg * 16 + (threadIdx.x & 15);
  const bool valid = i < j && j < n_ind;
  const uint64_t s0 = (uint64_t)ks * sites_per_slice;
  uint64_t s1 = s0 + sites_per_slice;
  if (s1 > n_sites_eff) s1 = n_sites_eff;
  const uint64_t kstride = (uint64_t)n_ig * 64;
  double acc[RB];
#pragma unroll
  for (int r = 0; r < RB; r++) acc[r] = 0;
  if (valid) {
    const double *pi = PA + (uint64_t)ig * 64 + (i & 15);
    const double *pj = PA + (uint64_t)jg * 64 + (j & 15);
    for (uint64_t s = s0; s < s1; s++) {
      const double *w = Wb + s * RB;  // uniform across the workgroup
      bool any = false;
#pragma unroll
      for (int r = 0; r < RB; r++) any |= w[r] != 0.0;
      if (!any) continue;  // drawn by none of these replicates
      double g1[3], g2[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const uint64_t k = 3 * s + c;
        const uint64_t off = (k >> 2) * kstride + (k & 3) * 16;
        g1[c] = pi[off];
        g2[c] = pj[off];
      }
      if (PDEL && (ngd_miss(g1[0], g1[1], g1[2]) || ngd_miss(g2[0], g2[1], g2[2]))) continue;
      // the site's score-weighted sum, exactly as the one-replicate kernel forms it (0 + c * 1 == c)
      const double c = FAST ? site_fast(g1, g2, sc, 0.0, 1.0) : site_faithful(g1, g2, sc, 0.0, 1.0, true);
      if (__builtin_fabs(c) <= 1.7976931348623157e308) {
#pragma unroll
        for (int r = 0; r < RB; r++) acc[r] = acc[r] + c * w[r];
      } else {  // NaN (an all-zero site): only the replicates that draw the site take it -- 0 x NaN must add nothing
#pragma unroll
        for (int r = 0; r < RB; r++) acc[r] = w[r] != 0.0 ? acc[r] + c * w[r] : acc[r];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; r++) slab[(((uint64_t)ks * RB + r) * n_pad + i) * n_pad + j] = acc[r];
}

}  // namespace

void ngd_launch_accum_em_batch(hipStream_t st, const ngd_geom &g, const double *PA, const double *d_Wb, int rb,
                               uint64_t n_sites_eff, const ngd_score &score, int pairwise_del, int fast,
                               const ngd_tile *d_tiles16, uint32_t n_tiles16, uint32_t n_ks,
                               uint64_t sites_per_slice, double *slab) {
  if (!n_tiles16) return;
  dim3 grid(n_tiles16 * n_ks), block(256);
#define NGD_EMB(F, P, R)                                                                                   \
  hipLaunchKernelGGL((k_accum_em_batch<F, P, R>), grid, block, 0, st, PA, d_Wb, score, d_tiles16, n_tiles16, \
                     g.n_ig, g.n_pad, g.n_ind, n_sites_eff, sites_per_slice, slab)
#define NGD_EMB_R(F, P)                                    \
  do {                                                     \
    if (rb == 4) NGD_EMB(F, P, 4);                         \
    else if (rb == 8) NGD_EMB(F, P, 8);                    \
    else NGD_EMB(F, P, 16);                                \
  } while (0)
  const bool p = pairwise_del != 0;
  if (fast) { if (p) NGD_EMB_R(true, true); else NGD_EMB_R(true, false); }
  else      { if (p) NGD_EMB_R(false, true); else NGD_EMB_R(false, false); }
#undef NGD_EMB_R
#undef NGD_EMB
}

void ngd_launch_accum_em(hipStream_t st, const ngd_geom &g, const double *PA, const uint32_t *d_ws,
                         uint64_t n_sites_eff, const ngd_score &score, int pairwise_del, int fast,
                         const ngd_tile *d_tiles16, uint32_t n_tiles16, uint32_t n_ks,
                         uint64_t sites_per_slice, double *slab) {
  if (!n_tiles16) return;
  dim3 grid(n_tiles16 * n_ks), block(256);
#define NGD_EM(F, W, P)                                                                              \
  hipLaunchKernelGGL((k_accum_em<F, W, P>), grid, block, 0, st, PA, d_ws, score, d_tiles16, n_tiles16, \
                     g.n_ig, g.n_pad, g.n_ind, n_sites_eff, sites_per_slice, slab)
  const bool w = d_ws != nullptr, p = pairwise_del != 0;
  if (fast) {
    if (w) { if (p) NGD_EM(true, true, true); else NGD_EM(true, true, false); }
    else   { if (p) NGD_EM(true, false, true); else NGD_EM(true, false, false); }
  } else {
    if (w) { if (p) NGD_EM(false, true, true); else NGD_EM(false, true, false); }
    else   { if (p) NGD_EM(false, false, true); else NGD_EM(false, false, false); }
  }
#undef NGD_EM
}
