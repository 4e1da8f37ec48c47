// accum_stream.hip -- the per-pair site accumulation of gen_dist()
// (reference ngsDist.cpp:333-364, --indep_geno branch of :353) in its literal
// data-parallel form: ONE WAVEFRONT PER PAIR (i1, i2).
//
// Per pair-site the wavefront moves 48 B (two 3-double GL vectors); a tile of
// 128 sites of each individual is fetched with fully coalesced 16-B loads,
// staged in LDS, re-read per lane as one site's 3 doubles, and accumulated in
// the reference's term order (g1 outer, g2 inner, all nine terms).  Lanes own
// sites s = lane (mod 64); the 64 partial sums are combined by a fixed
// shuffle tree, so the result is deterministic.  No MFMA: in this form the
// path is a bandwidth-bound streaming reduction.  (accum_mfma.hip is the
// form that re-uses operands; DESIGN.md compares the two.)
#include "ngd_internal.h"

namespace {

constexpr int TS = 128;              // sites per staged tile
constexpr int TILE_DBL = TS * 3;     // doubles per individual per tile (3 KiB)
constexpr int WAVES = 4;

__device__ __forceinline__ void pair_from_index(uint64_t n, uint64_t k, uint32_t &i, uint32_t &j) {
  // invert k = i*(2n-i-1)/2 + (j-i-1); float estimate then exact fix-up
  double b = 2.0 * (double)n - 1.0;
  double r = (b - sqrt(b * b - 8.0 * (double)k)) * 0.5;
  uint64_t ii = (uint64_t)r;
  if (ii + 2 > n) ii = n - 2;
  while (ii > 0 && ngd_pair_idx(n, ii, ii + 1) > k) ii--;
  while (ii + 2 < n && ngd_pair_idx(n, ii + 1, ii + 2) <= k) ii++;
  i = (uint32_t)ii;
  j = (uint32_t)(k - ngd_pair_idx(n, ii, ii + 1) + ii + 1);
}

template <bool WEIGHTED, bool PDEL>
__global__ __launch_bounds__(256) void k_accum_stream(ngd_geom g, const double *__restrict__ PI,
                                                       const uint32_t *__restrict__ ws,
                                                       uint64_t n_sites_eff, ngd_score sc,
                                                       const uint64_t *__restrict__ pairs,
                                                       uint64_t n_owned, double *__restrict__ d_sum) {
  __shared__ __attribute__((aligned(16))) double lds[WAVES][2][TILE_DBL];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint64_t slot = (uint64_t)blockIdx.x * WAVES + wave;
  const uint64_t stride = (uint64_t)gridDim.x * WAVES;
  double(*L)[TILE_DBL] = lds[wave];

  for (; slot < n_owned; slot += stride) {
    const uint64_t pk = pairs ? pairs[slot] : slot;
    uint32_t i1, i2;
    pair_from_index(g.n_ind, pk, i1, i2);
    const double *b1 = PI + (uint64_t)i1 * g.n_sites_pad * 3;
    const double *b2 = PI + (uint64_t)i2 * g.n_sites_pad * 3;
    double acc = 0;
    for (uint64_t s0 = 0; s0 < n_sites_eff; s0 += TS) {
      // stage 128 sites x 3 doubles of both individuals: 3 x (64 lanes x 16 B) each.
      // n_sites_pad is a multiple of 4 but not of 128: clamp the tail loads.
      const uint64_t lim = (g.n_sites_pad - s0) * 3;  // doubles available from s0
#pragma unroll
      for (int c = 0; c < 3; c++) {
        uint32_t d = (c * 64 + lane) * 2;
        double2 v1 = make_double2(0, 0), v2 = make_double2(0, 0);
        if (d + 1 < lim) {
          v1 = *reinterpret_cast<const double2 *>(b1 + s0 * 3 + d);
          v2 = *reinterpret_cast<const double2 *>(b2 + s0 * 3 + d);
        }
        *reinterpret_cast<double2 *>(&L[0][d]) = v1;
        *reinterpret_cast<double2 *>(&L[1][d]) = v2;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int h = 0; h < TS / 64; h++) {
        const int sl = h * 64 + lane;
        const uint64_t s = s0 + sl;
        const double p0 = L[0][3 * sl], p1 = L[0][3 * sl + 1], p2 = L[0][3 * sl + 2];
        const double q0 = L[1][3 * sl], q1 = L[1][3 * sl + 1], q2 = L[1][3 * sl + 2];
        bool use = s < n_sites_eff;
        if (PDEL) use = use && !(ngd_miss(p0, p1, p2) || ngd_miss(q0, q1, q2));
        // ngsDist.cpp:351-353: dist += score[g1][g2] * (p1[g1]*p2[g2]), nine terms in order
        double c = 0;
        c += sc.v[0] * (p0 * q0); c += sc.v[1] * (p0 * q1); c += sc.v[2] * (p0 * q2);
        c += sc.v[3] * (p1 * q0); c += sc.v[4] * (p1 * q1); c += sc.v[5] * (p1 * q2);
        c += sc.v[6] * (p2 * q0); c += sc.v[7] * (p2 * q1); c += sc.v[8] * (p2 * q2);
        if (WEIGHTED) c *= (double)(use ? ws[s] : 0u);
        acc += use ? c : 0.0;
      }
      __builtin_amdgcn_wave_barrier();
    }
    // fixed shuffle tree over the 64 lanes
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) d_sum[pk] = acc;
  }
}

}  // namespace

void ngd_launch_accum_stream(hipStream_t st, const ngd_geom &g, const double *PI, const uint32_t *d_ws,
                             uint64_t n_sites_eff, const ngd_score &score, int pairwise_del,
                             const uint64_t *d_pairs, uint64_t n_owned, double *d_sum) {
  if (!n_owned) return;
  uint64_t blocks = (n_owned + WAVES - 1) / WAVES;
  if (blocks > 256u * 64u) blocks = 256u * 64u;  // grid-stride beyond 64 workgroups per CU
  dim3 grid((unsigned)blocks), block(64 * WAVES);
#define NGD_LAUNCH(W, P)                                                                        \
  hipLaunchKernelGGL((k_accum_stream<W, P>), grid, block, 0, st, g, PI, d_ws, n_sites_eff, score, \
                     d_pairs, n_owned, d_sum)
  if (d_ws) {
    if (pairwise_del) NGD_LAUNCH(true, true); else NGD_LAUNCH(true, false);
  } else {
    if (pairwise_del) NGD_LAUNCH(false, true); else NGD_LAUNCH(false, false);
  }
#undef NGD_LAUNCH
}
