// layout.hip -- turns the caller's prepared GL array into the device-resident
// operand images (see ngd_internal.h for the fragment-major layout), derives
// the missing-site masks (reference gen_func.cpp:862-868) and, for bootstrap,
// expands a block map (reference ngsDist.cpp:416-437) into per-site
// multiplicities instead of moving any data.
#include "ngd_internal.h"

namespace {

struct ngd_d3 {
  double v[3];
};
inline ngd_d3 d3_of(const double *d3) { return d3 ? ngd_d3{{d3[0], d3[1], d3[2]}} : ngd_d3{{1.0, 1.0, 1.0}}; }

// One (individual, site): write every image that is allocated.
__device__ __forceinline__ void emit(const ngd_geom &g, const ngd_score &sc, int pairwise_del,
                                     uint64_t s, uint32_t i, double p0, double p1, double p2,
                                     double *PA, double *QB, double *PI, unsigned long long *mask) {
  if (PI && !sc.congruent) {  // individual-major copy for the streaming kernel (unmasked, as gen_dist reads it)
    double *d = PI + (i * g.n_sites_pad + s) * 3;
    d[0] = p0; d[1] = p1; d[2] = p2;
  }
  bool miss = ngd_miss(p0, p1, p2);
  if (mask && !miss) atomicOr(&mask[(uint64_t)i * g.n_words + (s >> 6)], 1ull << (s & 63));
  if (pairwise_del && miss) { p0 = 0; p1 = 0; p2 = 0; }  // a skipped site contributes nothing
  // congruent image of the reference's matrices, t = (p0 + p1 + p2, +-(p2 - p0), p1): the smaller of p0 and p2 is what
  // the image cannot hold to the last bit -- kept beside it (site-major: coalesced here), p is then recoverable (fixup.hip)
  if (PI && sc.congruent) PI[s * g.n_ind + i] = p0 < p2 ? p0 : p2;
  uint64_t k = 3 * s;
  if (PA && sc.congruent) {  // t_r = c_r . p (products and sums rounded one by one; exact for called genotypes)
    for (int r = 0; r < 3; r++) {
      double t = sc.c[3 * r] * p0;
      t = t + sc.c[3 * r + 1] * p1;
      t = t + sc.c[3 * r + 2] * p2;
      PA[ngd_frag_off(k + r, i, g.n_ig)] = t;
    }
  } else if (PA) {
    PA[ngd_frag_off(k, i, g.n_ig)] = p0;
    PA[ngd_frag_off(k + 1, i, g.n_ig)] = p1;
    PA[ngd_frag_off(k + 2, i, g.n_ig)] = p2;
  }
  if (QB) {  // q[g1] = sum_g2 score[g1][g2] * p[g2]
    for (int a = 0; a < 3; a++) {
      double q = sc.v[3 * a] * p0;
      q = q + sc.v[3 * a + 1] * p1;
      q = q + sc.v[3 * a + 2] * p2;
      QB[ngd_frag_off(k + a, i, g.n_ig)] = q;
    }
  }
}

// Three k-groups hold 12 contraction indices = 4 WHOLE sites: a wavefront takes one group of 16 individuals through
// QB_PER such periods -- whole 512-byte fragments in, whole fragments out, a site's three values exchanged through LDS
// (lane (k % 4, i % 16) of fragment f holds index 4 f + k % 4 of the period).  k-groups outside [kg_lo, kg_hi) belong
// to other ranges and are not written; those past the image's tail read as zero.
constexpr int QB_PER = 4;
__global__ __launch_bounds__(256) void k_qb_range(ngd_geom g, ngd_score sc, const double *__restrict__ PA, uint64_t u_lo,
                                                  uint64_t u_hi, uint64_t kg_lo, uint64_t kg_hi, double *__restrict__ QBs) {
  __shared__ double t[4][QB_PER][192];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kk = lane >> 4, ii = lane & 15;
  const uint32_t n_q = g.n_ig >> 2;  // (n_pad is a multiple of 128: n_ig of 8)
  const uint64_t kg_end = g.n_kg + NGD_KG_TAIL;
  // this lane's row of the score matrix in each of a period's three fragments: index 4 f + kk is genotype (f + kk) % 3
  double c[3][3];
#pragma unroll
  for (int f = 0; f < 3; f++) {
    const uint32_t a = (f + kk) % 3;
#pragma unroll
    for (int b = 0; b < 3; b++) c[f][b] = a == 0 ? sc.v[b] : a == 1 ? sc.v[3 + b] : sc.v[6 + b];
  }
  const uint32_t ig = (blockIdx.x % n_q) * 4 + wave;
  const uint64_t u0 = u_lo + (uint64_t)(blockIdx.x / n_q) * QB_PER;
  double p[QB_PER][3];
#pragma unroll
  for (int u = 0; u < QB_PER; u++)
#pragma unroll
    for (int f = 0; f < 3; f++) {
      const uint64_t kg = 3 * (u0 + u) + f;
      p[u][f] = (u0 + u < u_hi && kg < kg_end) ? PA[(kg * g.n_ig + ig) * 64 + lane] : 0.0;
    }
#pragma unroll
  for (int u = 0; u < QB_PER; u++)
#pragma unroll
    for (int f = 0; f < 3; f++) t[wave][u][f * 64 + lane] = p[u][f];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < QB_PER; u++)
#pragma unroll
    for (int f = 0; f < 3; f++) {
      const uint64_t kg = 3 * (u0 + u) + f;
      const uint32_t s3 = (4 * f + kk) / 3 * 3;  // first index of this lane's site within the period
      double q = c[f][0] * t[wave][u][s3 * 16 + ii];
      q = q + c[f][1] * t[wave][u][(s3 + 1) * 16 + ii];
      q = q + c[f][2] * t[wave][u][(s3 + 2) * 16 + ii];
      if (u0 + u < u_hi && kg >= kg_lo && kg < kg_hi) QBs[((kg - kg_lo) * g.n_ig + ig) * 64 + lane] = q;
    }
}

// Both operand images of the two-image arithmetic for a RANGE of k-groups, out of a congruent one-image engine's image T and
// its side array SM = min(p0, p2) (fixup.hip has the recovery: p1 = t2, delta = +-t1, the smaller of p0 / p2 beside the
// larger): Ps = p, Qs = score . p with emit()'s arithmetic, per-site weight folded into p (a bootstrap multiplicity: a small
// integer).  One thread per (site, padded individual) of the sites that reach into [kg_lo, kg_end); scratch entries of
// padding individuals, of sites past the data set and of the tail groups are written as zeros.
__global__ void k_pq_range(ngd_geom g, ngd_score sc, const double *__restrict__ T, const double *__restrict__ SM,
                           const uint32_t *__restrict__ ws, uint64_t s_first, uint64_t n_s, uint64_t kg_lo, uint64_t kg_end,
                           double *__restrict__ Ps, double *__restrict__ Qs) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_s * g.n_pad) return;
  const uint64_t s = s_first + e / g.n_pad;
  const uint32_t i = (uint32_t)(e % g.n_pad);
  double p0 = 0, p1 = 0, p2 = 0;
  if (s < g.n_sites && i < g.n_ind) {
    const uint64_t k = 3 * s;
    const double d = sc.fix_sign * T[ngd_frag_off(k + 1, i, g.n_ig)];  // p2 - p0
    const double m = T[ngd_frag_off(k + 2, i, g.n_ig)];
    const double sm = SM[s * g.n_ind + i];
    p1 = m;
    p0 = d >= 0 ? sm : sm - d;
    p2 = d >= 0 ? sm + d : sm;
  }
  const double w = (ws && s < g.n_sites) ? (double)ws[s] : 1.0;  // (on ONE operand: the product carries it once)
  const double p[3] = {p0 * w, p1 * w, p2 * w};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const uint64_t k = 3 * s + a, kg = k >> 2;
    if (kg < kg_lo || kg >= kg_end) continue;
    double q = sc.v[3 * a] * p0;
    q = q + sc.v[3 * a + 1] * p1;
    q = q + sc.v[3 * a + 2] * p2;
    const uint64_t off = ((kg - kg_lo) * g.n_ig + (i >> 4)) * 64 + (k & 3) * 16 + (i & 15);
    Ps[off] = p[a];
    Qs[off] = q;
  }
}

__global__ void k_layout(ngd_geom g, const double *__restrict__ raw, int raw_ind_major, uint64_t s0,
                         uint64_t n_chunk, ngd_score sc, int pairwise_del, double *PA, double *QB,
                         double *PI, unsigned long long *mask) {
  uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_chunk * g.n_ind) return;
  uint64_t sl = e / g.n_ind;
  uint32_t i = (uint32_t)(e - sl * g.n_ind);
  const double *src = raw_ind_major ? raw + ((uint64_t)i * n_chunk + sl) * 3 : raw + e * 3;
  emit(g, sc, pairwise_del, s0 + sl, i, src[0], src[1], src[2], PA, QB, PI, mask);
}

// K0: the kernel-input construction of the reference for ONE (individual, site) of
// a binary GL file, on the device: read_data.cpp:37-45 (log unless --log_scale, -inf
// clamp of conv_space gen_func.cpp:123-130, post_prob :920-932 over logsum :135-151,
// NaN check) then ngsDist.cpp:165-174 (call_geno gen_func.cpp:886-914, exp).
// Same operation order as the host; log/exp are the device's (<= 1 ulp of glibc's).
__global__ void k_prep_layout(ngd_geom g, const double *__restrict__ raw, uint64_t s0, uint64_t n_chunk,
                              int in_logscale, int call_geno, double N_thresh, double call_thresh,
                              ngd_score sc, int pairwise_del, double *PA, double *QB, double *PI,
                              unsigned long long *mask, int *nan_flag) {
  uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_chunk * g.n_ind) return;
  uint64_t sl = e / g.n_ind;
  uint32_t i = (uint32_t)(e - sl * g.n_ind);
  double l[3] = {raw[3 * e], raw[3 * e + 1], raw[3 * e + 2]};
  if (!in_logscale) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      l[c] = log(l[c]);
      if (l[c] == -INFINITY) l[c] = -1e15;  // INF, gen_func.hpp:15
    }
  }
  {  // post_prob / logsum
    double M = l[0];
    M = (l[1] >= M ? l[1] : M);
    M = (l[2] >= M ? l[2] : M);
    double norm;
    if (M == -INFINITY) {
      norm = -INFINITY;
    } else {
      double sum = 0;
      sum += exp(l[0] - M); sum += exp(l[1] - M); sum += exp(l[2] - M);
      norm = log(sum) + M;
    }
    l[0] -= norm; l[1] -= norm; l[2] -= norm;
  }
  if (l[0] != l[0] || l[1] != l[1] || l[2] != l[2]) atomicOr(nan_flag, 1);
  if (call_geno) {
    int max_pos = 0, min_pos = 0;
    double mx = -INFINITY, mn = INFINITY;
#pragma unroll
    for (int c = 0; c < 3; c++) if (l[c] > mx) { mx = l[c]; max_pos = c; }
#pragma unroll
    for (int c = 0; c < 3; c++) if (l[c] < mn) { mn = l[c]; min_pos = c; }
    double max_pp = exp(l[max_pos]);
    if (l[min_pos] == l[max_pos]) max_pp = -1;
    if (max_pp < N_thresh) { const double t = log((double)1 / 3); l[0] = l[1] = l[2] = t; }
    if (max_pp >= call_thresh) {
      l[0] = l[1] = l[2] = -1e15;
      l[max_pos] = 0.0;  // log(1)
    }
  }
  emit(g, sc, pairwise_del, s0 + sl, i, exp(l[0]), exp(l[1]), exp(l[2]), PA, QB, PI, mask);
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Counter-based synthetic data, bit-identical to oracle ngo_synth_one().
__global__ void k_synth(ngd_geom g, uint64_t seed, double miss_frac, uint64_t site0, ngd_score sc,
                        int pairwise_del, double *PA, double *QB, double *PI, unsigned long long *mask) {
  uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= g.n_sites * g.n_ind) return;
  uint64_t s = e / g.n_ind;
  uint32_t i = (uint32_t)(e - s * g.n_ind);
  e += site0 * g.n_ind;  // the generator is indexed by the site's position in the WHOLE data set
  uint64_t base = seed * 0x9E3779B97F4A7C15ull;
  double x[3];
  for (int c = 0; c < 3; c++) {
    uint64_t z = mix64(base + (e * 3 + (uint64_t)c));
    double u = ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    x[c] = (u * u) * u;
  }
  double t = (x[0] + x[1]) + x[2];
  double p0 = x[0] / t, p1 = x[1] / t, p2 = x[2] / t;
  if (miss_frac > 0) {
    uint64_t z = mix64(base ^ (0xD1B54A32D192ED03ull + e));
    double u = ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    if (u < miss_frac) p0 = p1 = p2 = (double)1 / 3;
  }
  emit(g, sc, pairwise_del, s, i, p0, p1, p2, PA, QB, PI, mask);
}

// per-site multiplicity ws[s] (EM / streaming kernels, count planes) and, for the MFMA kernel, the
// same thing per contraction index k = 3 s + g as a double, wk[k], so that its operand pipeline fetches the
// weight of a k-group like any other operand (no integer division, no conversion in the hot loop)
__global__ void k_expand(const uint32_t *__restrict__ mult, uint64_t n_eff, uint64_t block_size,
                         uint64_t n_sites, uint32_t *ws, double *wk, ngd_d3 d3) {
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_sites) return;
  const uint32_t m = s < n_eff ? mult[s / block_size] : 0u;
  ws[s] = m;
  if (wk) {  // (d3: 1, 1, 1 -- or the congruence's diagonal, single_image = 2)
    wk[3 * s] = (double)m * d3.v[0];
    wk[3 * s + 1] = (double)m * d3.v[1];
    wk[3 * s + 2] = (double)m * d3.v[2];
  }
}

__global__ void k_index_weights(uint64_t n_k, ngd_d3 d3, double *W) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_k) W[k] = d3.v[k % 3];
}

// W[s][RB] for the EM batch kernel: replicate r's multiplicity of site s (mult is [n_rep][n_blocks]); rows
// r >= n_rep are zero; with lead_full, row 0 is the full data set instead: weight 1 on EVERY site
__global__ void k_expand_batch(const uint32_t *__restrict__ mult, uint32_t n_rep, uint32_t rb, int lead_full,
                               uint64_t n_blocks, uint64_t block_size, uint64_t n_sites, uint64_t n_sites_alloc,
                               double *W) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_sites_alloc) return;
  const uint64_t b = s / block_size;
  const bool in_boot = b < n_blocks;
  for (uint32_t r = 0; r < rb; r++) {
    double w = 0.0;
    if (lead_full && r == 0) w = s < n_sites ? 1.0 : 0.0;
    else {
      const uint32_t q = r - (lead_full ? 1u : 0u);
      if (q < n_rep && in_boot) w = (double)mult[(uint64_t)q * n_blocks + b];
    }
    W[s * rb + r] = w;
  }
}

// 0/1 weights per slice for bootstrap blocks that are not whole k-groups (accum_mfma.hip, k_per_slice)
__global__ void k_slice_weights(uint32_t n_slices, uint32_t stride, uint64_t k_per_slice, uint64_t k_total, double *W,
                                ngd_d3 d3) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)n_slices * stride * 4) return;
  const uint64_t slice = t / ((uint64_t)stride * 4), r = t % ((uint64_t)stride * 4);
  const uint64_t k = (((slice * k_per_slice) >> 2) << 2) + r;
  const uint64_t lo = slice * k_per_slice, hi = lo + k_per_slice < k_total ? lo + k_per_slice : k_total;
  W[t] = (k >= lo && k < hi) ? d3.v[k % 3] : 0.0;
}

// bit-planes of the per-site multiplicity, for weighted valid-site counts
__global__ void k_planes(const uint32_t *__restrict__ ws, uint64_t n_sites, uint32_t n_words,
                         uint32_t n_planes, unsigned long long *planes) {
  uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  for (uint32_t b = 0; b < n_planes; b++) {
    unsigned long long bits = 0;
    for (int t = 0; t < 64; t++) {
      uint64_t s = (uint64_t)w * 64 + t;
      if (s < n_sites && ((ws[s] >> b) & 1u)) bits |= 1ull << t;
    }
    planes[(uint64_t)b * n_words + w] = bits;
  }
}

// ---- list of the k-groups a bootstrap replicate visits at all -------------------------------------
// A k-group (4 consecutive contraction indices = parts of two sites) whose four weights are zero
// contributes nothing: about 1/e of the sites of a replicate are not drawn (1/e^2 of the k-groups at
// block size 1, 1/e of them for blocks of a few sites and more).  The weighted MFMA pass walks this list
// instead of the whole k range.  Order-preserving compaction in three small kernels (count per 1024
// k-groups, scan of the counts, scatter), so the list is ascending and the pass stays deterministic.
constexpr int CPT = 4;               // k-groups per thread
constexpr int CPB = 256 * CPT;       // k-groups per workgroup

__device__ __forceinline__ bool kg_live(const double *__restrict__ wk, uint64_t kg, uint64_t n_kg) {
  if (kg >= n_kg) return false;
  const double *w = wk + 4 * kg;
  return w[0] != 0.0 || w[1] != 0.0 || w[2] != 0.0 || w[3] != 0.0;
}

__global__ __launch_bounds__(256) void k_kg_count(const double *__restrict__ wk, uint64_t n_kg, uint32_t *counts) {
  const uint64_t base = (uint64_t)blockIdx.x * CPB + threadIdx.x * CPT;
  int c = 0;
#pragma unroll
  for (int t = 0; t < CPT; t++) c += kg_live(wk, base + t, n_kg);
  __shared__ uint32_t sh[256];
  sh[threadIdx.x] = c;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = sh[0];
}

// exclusive scan of the per-workgroup counts, in place; the grand total goes to counts[n]
__global__ __launch_bounds__(1024) void k_kg_scan(uint32_t *counts, uint32_t n) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < n; b0 += 1024) {
    const uint32_t i = b0 + threadIdx.x;
    const uint32_t v = i < n ? counts[i] : 0u;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
      const uint32_t add = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < n) counts[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[n] = carry;
}

__global__ __launch_bounds__(256) void k_kg_scatter(const double *__restrict__ wk, uint64_t n_kg,
                                                     const uint32_t *__restrict__ offsets, uint32_t *list) {
  const uint64_t base = (uint64_t)blockIdx.x * CPB + threadIdx.x * CPT;
  bool live[CPT];
  uint32_t c = 0;
#pragma unroll
  for (int t = 0; t < CPT; t++) { live[t] = kg_live(wk, base + t, n_kg); c += live[t]; }
  __shared__ uint32_t sh[256];
  sh[threadIdx.x] = c;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t add = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
    __syncthreads();
    sh[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t o = offsets[blockIdx.x] + sh[threadIdx.x] - c;
#pragma unroll
  for (int t = 0; t < CPT; t++)
    if (live[t]) list[o++] = (uint32_t)(base + t);
}

// the entries past the end of the list point at the zeroed tail k-groups of the images (pipeline run-ahead)
__global__ void k_kg_pad(uint32_t *list, const uint32_t *total, uint32_t n_pad_entries, uint32_t tail_kg) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_pad_entries) list[*total + t] = tail_kg;
}

}  // namespace

uint32_t ngd_kg_count_blocks(uint64_t n_kg) { return (uint32_t)((n_kg + CPB - 1) / CPB); }

void ngd_launch_kg_compact(hipStream_t st, const double *d_wk, uint64_t n_kg, uint32_t tail_kg, uint32_t *d_counts,
                           uint32_t *d_list) {
  const uint32_t nb = ngd_kg_count_blocks(n_kg);
  hipLaunchKernelGGL(k_kg_count, dim3(nb), dim3(256), 0, st, d_wk, n_kg, d_counts);
  hipLaunchKernelGGL(k_kg_scan, dim3(1), dim3(1024), 0, st, d_counts, nb);
  hipLaunchKernelGGL(k_kg_scatter, dim3(nb), dim3(256), 0, st, d_wk, n_kg, d_counts, d_list);
  hipLaunchKernelGGL(k_kg_pad, dim3(1), dim3(64), 0, st, d_list, d_counts + nb, (uint32_t)NGD_KG_LIST_PAD, tail_kg);
}

// Single-image engines (ngd_config.single_image): the score-weighted image QB is not resident; before a launch of the MFMA
// kernel over k-groups [kg_lo, kg_hi) it is formed from PA into a scratch (element (i, k) at ngd_frag_off(k, i) minus the
// range's first k-group) with layout.hip's own arithmetic (emit() above: q[g1] = sum_g2 score[g1][g2] * p[g2], products
// and sums rounded one by one), so the kernel's results carry the same bits as with both images resident.
void ngd_launch_qb_range(hipStream_t st, const ngd_geom &g, const ngd_score &score, const double *PA, uint64_t kg_lo,
                         uint64_t kg_hi, double *QBs) {
  if (kg_hi <= kg_lo) return;
  const uint64_t u_lo = kg_lo / 3, u_hi = (kg_hi + 2) / 3;  // every period of three k-groups with one in the range
  const uint64_t n_blk = (u_hi - u_lo + QB_PER - 1) / QB_PER * (g.n_ig >> 2);
  hipLaunchKernelGGL(k_qb_range, dim3((unsigned)n_blk), dim3(256), 0, st, g, score, PA, u_lo, u_hi, kg_lo, kg_hi, QBs);
}

void ngd_launch_pq_range(hipStream_t st, const ngd_geom &g, const ngd_score &score, const double *T, const double *SM,
                         const uint32_t *d_ws, uint64_t kg_lo, uint64_t kg_end, double *Ps, double *Qs) {
  if (kg_end <= kg_lo) return;
  const uint64_t s_first = 4 * kg_lo / 3, s_last = (4 * kg_end + 2) / 3;  // every site with an index in [4 kg_lo, 4 kg_end)
  const uint64_t n = (s_last - s_first) * g.n_pad;
  hipLaunchKernelGGL(k_pq_range, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, score, T, SM, d_ws, s_first,
                     s_last - s_first, kg_lo, kg_end, Ps, Qs);
}

void ngd_launch_layout(hipStream_t st, const ngd_geom &g, const double *raw, int raw_ind_major,
                       uint64_t s0, uint64_t n_chunk, const ngd_score &score, int pairwise_del,
                       double *PA, double *QB, double *PI, unsigned long long *mask) {
  uint64_t n = n_chunk * g.n_ind;
  if (!n) return;
  hipLaunchKernelGGL(k_layout, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, raw,
                     raw_ind_major, s0, n_chunk, score, pairwise_del, PA, QB, PI, mask);
}

void ngd_launch_prep_layout(hipStream_t st, const ngd_geom &g, const double *raw, uint64_t s0, uint64_t n_chunk,
                            int in_logscale, int call_geno, double N_thresh, double call_thresh,
                            const ngd_score &score, int pairwise_del, double *PA, double *QB, double *PI,
                            unsigned long long *mask, int *nan_flag) {
  uint64_t n = n_chunk * g.n_ind;
  if (!n) return;
  hipLaunchKernelGGL(k_prep_layout, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, raw, s0, n_chunk,
                     in_logscale, call_geno, N_thresh, call_thresh, score, pairwise_del, PA, QB, PI, mask, nan_flag);
}

void ngd_launch_synth(hipStream_t st, const ngd_geom &g, uint64_t seed, double miss_frac, uint64_t site0,
                      const ngd_score &score, int pairwise_del, double *PA, double *QB, double *PI,
                      unsigned long long *mask) {
  uint64_t n = g.n_sites * g.n_ind;
  hipLaunchKernelGGL(k_synth, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, seed,
                     miss_frac, site0, score, pairwise_del, PA, QB, PI, mask);
}

void ngd_launch_weights(hipStream_t st, uint64_t n_blocks, uint64_t block_size, uint64_t n_sites,
                        const uint32_t *d_mult, uint32_t *d_ws, double *d_wk, const double *d3) {
  hipLaunchKernelGGL(k_expand, dim3((unsigned)((n_sites + 255) / 256)), dim3(256), 0, st, d_mult,
                     n_blocks * block_size, block_size, n_sites, d_ws, d_wk, d3_of(d3));
}

void ngd_launch_index_weights(hipStream_t st, uint64_t n_k, const double *d3, double *d_W) {
  hipLaunchKernelGGL(k_index_weights, dim3((unsigned)((n_k + 255) / 256)), dim3(256), 0, st, n_k, d3_of(d3), d_W);
}

void ngd_launch_weights_batch(hipStream_t st, const uint32_t *d_mult, uint32_t n_rep, uint32_t rb, int lead_full,
                              uint64_t n_blocks, uint64_t block_size, uint64_t n_sites, uint64_t n_sites_alloc,
                              double *d_W) {
  hipLaunchKernelGGL(k_expand_batch, dim3((unsigned)((n_sites_alloc + 255) / 256)), dim3(256), 0, st, d_mult, n_rep, rb,
                     lead_full, n_blocks, block_size, n_sites, n_sites_alloc, d_W);
}

void ngd_launch_slice_weights(hipStream_t st, uint32_t n_slices, uint32_t stride, uint64_t k_per_slice, uint64_t k_total,
                              double *d_W, const double *d3) {
  const uint64_t n = (uint64_t)n_slices * stride * 4;
  if (!n) return;
  hipLaunchKernelGGL(k_slice_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n_slices, stride, k_per_slice,
                     k_total, d_W, d3_of(d3));
}

void ngd_launch_planes(hipStream_t st, const uint32_t *d_ws, uint64_t n_sites, uint32_t n_words,
                       uint32_t n_planes, unsigned long long *d_planes) {
  hipLaunchKernelGGL(k_planes, dim3((n_words + 255) / 256), dim3(256), 0, st, d_ws, n_sites, n_words,
                     n_planes, d_planes);
}
