// fixup.hip -- the pairs a single-image engine in congruent coordinates (ngd_config.single_image = 2, accum_mfma.hip on
// ONE operand image) cannot hold to 1e-9 relative, recomputed the reference's way.
//
// In congruent coordinates gen_dist()'s sum (ngsDist.cpp:351-353) is a DIFFERENCE of squares per site,
//   p1' S p2 = 1/2 (s1 s2 - delta1 delta2 - m1 m2),   s = p0 + p1 + p2, delta = p2 - p0, m = p1,
// so a pair's sum carries an absolute error of <= 4e-17 per site whatever its size: fine wherever the mean per-site term
// is above ~1e-7, not for nearly identical individuals (clones, technical replicates: confident, equal genotypes
// everywhere), whose sums are tiny.  The reduction kernels note every pair whose sum is below NGD_FIX_MEAN (1e-6) x the
// sites its matrix visits -- relative error <= 4e-11 for all the others -- and this file recomputes the noted ones as
// gen_dist() does: nine products score[g1][g2] * (p1[g1] * p2[g2]) per site, all of one sign, added in the reference's
// order (ngsDist.cpp:351-353), from p itself.
//
// p is not resident (that is the point of the one image), but it comes back from the image to the last bit or so given ONE
// more double per (individual, site): the image holds m = p1 exactly and delta = fl(p2 - p0); what it cannot hold is the
// SMALLER of p0 and p2 next to the larger (1e-20 beside 1 is gone from s and delta alike), so emit() (layout.hip) keeps
// sm = min(p0, p2) beside the image (SM[site][individual], 8 of the 24 bytes a second image would take) and
//   p1 = m;  delta >= 0: p0 = sm, p2 = sm + delta;  delta < 0: p2 = sm, p0 = sm - delta
// (the larger one to two roundings, relative).  Missing sites under --pairwise_del are zero in the image and in SM: they
// add exactly nothing, as in the MFMA pass.
//
// Two kernels.  k_fixup: one wavefront per (noted pair, slice of sites): lanes take consecutive sites, every value is its own
// 64-byte sector of the fragment-major image (8 x the bytes of a streaming read: ~400 B per pair-site) -- for pairs that
// are alone in their neighbourhood.  k_fixup_tile: nearly identical individuals come in clusters, so a 16 x 16 tile of
// individuals that holds several noted pairs is recomputed whole, coalesced (below).  The engine groups the noted pairs by
// tile on the host and recomputes every one of them in launches of bounded size (engine.hip fixup_pass; a caller-set
// budget, NGD_OPT_FIXUP_WORK, is the only way to have pairs left as the MFMA pass computed them).
#include "ngd_internal.h"

namespace {

template <bool WEIGHTED>
__global__ __launch_bounds__(64) void k_fixup(ngd_geom g, ngd_score sc, const double *__restrict__ T,
                                               const double *__restrict__ SM, const uint32_t *__restrict__ ws,
                                               const unsigned long long *__restrict__ list, uint64_t s_lo, uint64_t s_hi,
                                               uint64_t sites_per_slice, uint32_t n_slices, int out_mode,
                                               double *__restrict__ out) {
  const uint32_t q = blockIdx.x / n_slices, sl = blockIdx.x % n_slices;
  const uint32_t lane = threadIdx.x;
  const unsigned long long ij = list[q];
  const uint32_t i = (uint32_t)(ij >> 32), j = (uint32_t)ij;
  const uint64_t a = s_lo + (uint64_t)sl * sites_per_slice;
  uint64_t b = a + sites_per_slice;
  if (b > s_hi) b = s_hi;
  double acc = 0;
  for (uint64_t s = a + lane; s < b; s += 64) {
    double p[2][3];
#pragma unroll
    for (int w = 0; w < 2; w++) {
      const uint32_t x = w ? j : i;
      const uint64_t k = 3 * s;
      const double d = sc.fix_sign * T[ngd_frag_off(k + 1, x, g.n_ig)];  // p2 - p0
      const double m = T[ngd_frag_off(k + 2, x, g.n_ig)];
      const double sm = SM[s * g.n_ind + x];
      p[w][1] = m;
      p[w][0] = d >= 0 ? sm : sm - d;
      p[w][2] = d >= 0 ? sm + d : sm;
    }
    // ngsDist.cpp:351-353: dist += score[g1][g2] * (p1[g1]*p2[g2]), nine terms in order (accum_stream.hip's arithmetic)
    double c = 0;
    c += sc.v[0] * (p[0][0] * p[1][0]); c += sc.v[1] * (p[0][0] * p[1][1]); c += sc.v[2] * (p[0][0] * p[1][2]);
    c += sc.v[3] * (p[0][1] * p[1][0]); c += sc.v[4] * (p[0][1] * p[1][1]); c += sc.v[5] * (p[0][1] * p[1][2]);
    c += sc.v[6] * (p[0][2] * p[1][0]); c += sc.v[7] * (p[0][2] * p[1][1]); c += sc.v[8] * (p[0][2] * p[1][2]);
    if (WEIGHTED) c *= (double)ws[s];
    acc += c;
  }
  // fixed shuffle tree over the 64 lanes
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) {
    if (out_mode == 0) out[(uint64_t)q * n_slices + sl] = acc;
    else out[((uint64_t)sl * g.n_pad + i) * g.n_pad + j] = acc;
  }
}

// The same recomputation for a whole 16 x 16 tile of pairs (row group ig, column group jg of 16 individuals) at once:
// nearly identical individuals come in CLUSTERS (k copies are k (k - 1) / 2 noted pairs that share their individuals), and
// pair by pair every value above is its own 64-byte sector.  A workgroup of 256 threads (thread = pair) takes the tile
// through a slice of sites four at a time: 128 threads recover p of the 16 + 16 individuals at the 4 sites -- 16
// consecutive individuals of one (site, coordinate) are 128 contiguous bytes of the image, 16 consecutive entries of SM --
// into LDS, then every thread adds its pair's nine products, sites in ascending order.  4 bytes of traffic per
// pair-site instead of ~400; only the noted pairs of the tile (mask) are written.  [measured] tools/fixup_cost.py.
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void k_fixup_tile(ngd_geom g, ngd_score sc, const double *__restrict__ T,
                                                     const double *__restrict__ SM, const uint32_t *__restrict__ ws,
                                                     const ngd_fix_tile *__restrict__ tiles, uint64_t s_lo, uint64_t s_hi,
                                                     uint64_t sites_per_slice, uint32_t n_slices, int out_mode,
                                                     double *__restrict__ out) {
  __shared__ double P[2][4][16][3];  // [row / column side][site of the four][individual of the group][genotype]
  const uint32_t q = blockIdx.x / n_slices, sl = blockIdx.x % n_slices;
  const uint32_t ig = tiles[q].ig, jg = tiles[q].jg;
  const uint32_t tid = threadIdx.x, r = tid >> 4, c = tid & 15;
  const uint64_t a = s_lo + (uint64_t)sl * sites_per_slice;
  uint64_t b = a + sites_per_slice;
  if (b > s_hi) b = s_hi;
  // the recovering role (threads 0..127): side, site of the four, individual
  const uint32_t side = tid >> 6, su = (tid >> 4) & 3, x = (side ? jg : ig) * 16 + (tid & 15);
  double acc = 0;
  for (uint64_t s4 = a; s4 < b; s4 += 4) {
    if (tid < 128) {
      double p0 = 0, p1 = 0, p2 = 0;
      const uint64_t s = s4 + su;
      if (s < b && x < g.n_ind) {
        const uint64_t k = 3 * s;
        const double d = sc.fix_sign * T[ngd_frag_off(k + 1, x, g.n_ig)];  // p2 - p0
        const double m = T[ngd_frag_off(k + 2, x, g.n_ig)];
        const double sm = SM[s * g.n_ind + x];
        p1 = m;
        p0 = d >= 0 ? sm : sm - d;
        p2 = d >= 0 ? sm + d : sm;
      }
      P[side][su][tid & 15][0] = p0; P[side][su][tid & 15][1] = p1; P[side][su][tid & 15][2] = p2;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {  // (sites past the slice were stored as zeros: they add +0)
      const double *pa = P[0][u][r], *pb = P[1][u][c];
      // ngsDist.cpp:351-353: dist += score[g1][g2] * (p1[g1]*p2[g2]), nine terms in order
      double t = 0;
      t += sc.v[0] * (pa[0] * pb[0]); t += sc.v[1] * (pa[0] * pb[1]); t += sc.v[2] * (pa[0] * pb[2]);
      t += sc.v[3] * (pa[1] * pb[0]); t += sc.v[4] * (pa[1] * pb[1]); t += sc.v[5] * (pa[1] * pb[2]);
      t += sc.v[6] * (pa[2] * pb[0]); t += sc.v[7] * (pa[2] * pb[1]); t += sc.v[8] * (pa[2] * pb[2]);
      if (WEIGHTED) t *= (s4 + u < b) ? (double)ws[s4 + u] : 0.0;
      acc += t;
    }
    __syncthreads();
  }
  if (!((tiles[q].mask[tid >> 6] >> (tid & 63)) & 1ull)) return;  // not a noted pair
  if (out_mode == 0) out[((uint64_t)q * n_slices + sl) * 256 + tid] = acc;
  else out[((uint64_t)sl * g.n_pad + (ig * 16 + r)) * g.n_pad + (jg * 16 + c)] = acc;
}

__global__ __launch_bounds__(256) void k_fixup_tile_finish(const ngd_fix_tile *__restrict__ tiles, const double *__restrict__ parts,
                                                            uint32_t n_slices, uint64_t n_ind, double *__restrict__ d_sum) {
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  if (!((tiles[q].mask[tid >> 6] >> (tid & 63)) & 1ull)) return;
  double s = 0;
  for (uint32_t sl = 0; sl < n_slices; sl++) s += parts[((uint64_t)q * n_slices + sl) * 256 + tid];
  d_sum[ngd_pair_idx(n_ind, tiles[q].ig * 16 + (tid >> 4), tiles[q].jg * 16 + (tid & 15))] = s;
}

__global__ __launch_bounds__(64) void k_fixup_finish(const unsigned long long *__restrict__ list, uint32_t n_list,
                                                      const double *__restrict__ parts, uint32_t n_slices, uint64_t n_ind,
                                                      double *__restrict__ d_sum) {
  const uint32_t q = blockIdx.x * 64 + threadIdx.x;
  if (q >= n_list) return;
  const unsigned long long ij = list[q];
  double s = 0;
  for (uint32_t sl = 0; sl < n_slices; sl++) s += parts[(uint64_t)q * n_slices + sl];
  d_sum[ngd_pair_idx(n_ind, (uint32_t)(ij >> 32), (uint32_t)ij)] = s;
}

}  // namespace

void ngd_launch_fixup(hipStream_t st, const ngd_geom &g, const ngd_score &score, const double *T, const double *SM,
                      const uint32_t *d_ws, const unsigned long long *d_list, uint32_t n_list, uint64_t s_lo, uint64_t s_hi,
                      uint64_t sites_per_slice, uint32_t n_slices, int out_mode, double *out) {
  if (!n_list || !n_slices) return;
  const dim3 grid(n_list * n_slices), block(64);
  if (d_ws)
    hipLaunchKernelGGL(k_fixup<true>, grid, block, 0, st, g, score, T, SM, d_ws, d_list, s_lo, s_hi, sites_per_slice,
                       n_slices, out_mode, out);
  else
    hipLaunchKernelGGL(k_fixup<false>, grid, block, 0, st, g, score, T, SM, d_ws, d_list, s_lo, s_hi, sites_per_slice,
                       n_slices, out_mode, out);
}

void ngd_launch_fixup_finish(hipStream_t st, const ngd_geom &g, const unsigned long long *d_list, uint32_t n_list,
                             const double *parts, uint32_t n_slices, double *d_sum) {
  if (!n_list) return;
  hipLaunchKernelGGL(k_fixup_finish, dim3((n_list + 63) / 64), dim3(64), 0, st, d_list, n_list, parts, n_slices, g.n_ind,
                     d_sum);
}

void ngd_launch_fixup_tiles(hipStream_t st, const ngd_geom &g, const ngd_score &score, const double *T, const double *SM,
                            const uint32_t *d_ws, const ngd_fix_tile *d_tiles, uint32_t n_tiles, uint64_t s_lo, uint64_t s_hi,
                            uint64_t sites_per_slice, uint32_t n_slices, int out_mode, double *out) {
  if (!n_tiles || !n_slices) return;
  const dim3 grid(n_tiles * n_slices), block(256);
  if (d_ws)
    hipLaunchKernelGGL(k_fixup_tile<true>, grid, block, 0, st, g, score, T, SM, d_ws, d_tiles, s_lo, s_hi, sites_per_slice,
                       n_slices, out_mode, out);
  else
    hipLaunchKernelGGL(k_fixup_tile<false>, grid, block, 0, st, g, score, T, SM, d_ws, d_tiles, s_lo, s_hi, sites_per_slice,
                       n_slices, out_mode, out);
}

void ngd_launch_fixup_tiles_finish(hipStream_t st, const ngd_geom &g, const ngd_fix_tile *d_tiles, uint32_t n_tiles,
                                   const double *parts, uint32_t n_slices, double *d_sum) {
  if (!n_tiles) return;
  hipLaunchKernelGGL(k_fixup_tile_finish, dim3(n_tiles), dim3(256), 0, st, d_tiles, parts, n_slices, g.n_ind, d_sum);
}
