// reduce.hip -- (1) fixed-order reduction of the per-slice slabs into the
// per-pair sums, (2) per-pair valid-site counts for --pairwise_del
// (reference ngsDist.cpp:335-338 + :362: cnt++ only on sites where neither
// individual is missing), from bit masks: cnt = popcount(mask_i & mask_j),
// bootstrap multiplicities entering as bit-planes.
#include "ngd_internal.h"

namespace {

// grid = owned 128-tiles x 128 rows; 128 threads = columns.  Eight interleaved partial sums (slice
// ks goes to partial ks % 8) keep eight loads in flight per thread; they are combined in a fixed tree,
// so the result is a fixed function of the slabs (deterministic), just not the left-to-right sum.
__global__ __launch_bounds__(128) void k_reduce(const double *__restrict__ slab, uint32_t n_ks,
                                                 const ngd_tile *__restrict__ tiles, uint32_t n_pad,
                                                 uint64_t n_ind, double *__restrict__ d_sum) {
  const uint32_t tile = blockIdx.x >> 7, row = blockIdx.x & 127;
  const uint32_t i = tiles[tile].ti * NGD_TILE + row;
  const uint32_t j = tiles[tile].tj * NGD_TILE + threadIdx.x;
  if (!(i < j && j < n_ind)) return;
  const uint64_t plane = (uint64_t)n_pad * n_pad;
  const double *p = slab + (uint64_t)i * n_pad + j;
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t ks = 0;
  for (; ks + 8 <= n_ks; ks += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) s[u] += p[(uint64_t)(ks + u) * plane];
  }
  for (int u = 0; ks < n_ks; ks++, u++) s[u] += p[(uint64_t)ks * plane];
  d_sum[ngd_pair_idx(n_ind, i, j)] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// Bootstrap replicate from per-block partial sums: sum = SUM_slice w[slice] * slab[slice], slices in
// ascending order (deterministic).  w = multiplicity of the slice's block in this replicate
// (reference rnd_map_data, ngsDist.cpp:416-437: a block drawn m times is visited m times).
__global__ __launch_bounds__(128) void k_reduce_w(const double *__restrict__ slab, uint32_t n_ks,
                                                   const double *__restrict__ w,
                                                   const ngd_tile *__restrict__ tiles, uint32_t n_pad,
                                                   uint64_t n_ind, double *__restrict__ d_sum) {
  const uint32_t tile = blockIdx.x >> 7, row = blockIdx.x & 127;
  const uint32_t i = tiles[tile].ti * NGD_TILE + row;
  const uint32_t j = tiles[tile].tj * NGD_TILE + threadIdx.x;
  if (!(i < j && j < n_ind)) return;
  const uint64_t plane = (uint64_t)n_pad * n_pad;
  const double *p = slab + (uint64_t)i * n_pad + j;
  double s = 0;
  for (uint32_t ks = 0; ks < n_ks; ks++) {
    const double wk = w[ks];  // uniform across the workgroup
    if (wk != 0.0) s += wk * p[ks * plane];
  }
  d_sum[ngd_pair_idx(n_ind, i, j)] = s;
}

__global__ __launch_bounds__(128) void k_fill_cnt(const ngd_tile *__restrict__ tiles, uint64_t n_ind,
                                                   unsigned long long value,
                                                   unsigned long long *__restrict__ d_cnt) {
  const uint32_t tile = blockIdx.x >> 7, row = blockIdx.x & 127;
  const uint32_t i = tiles[tile].ti * NGD_TILE + row;
  const uint32_t j = tiles[tile].tj * NGD_TILE + threadIdx.x;
  if (!(i < j && j < n_ind)) return;
  d_cnt[ngd_pair_idx(n_ind, i, j)] = value;
}

constexpr int CW = 64;  // mask words staged per step

// block = 16x16 pairs; LDS holds CW words of the 16+16 individuals (and planes).
__global__ __launch_bounds__(256) void k_count(const unsigned long long *__restrict__ mask,
                                                const unsigned long long *__restrict__ planes,
                                                uint32_t n_planes, uint32_t n_words,
                                                const ngd_tile *__restrict__ tiles, uint64_t n_ind,
                                                unsigned long long *__restrict__ d_cnt) {
  __shared__ unsigned long long mi[16][CW + 1], mj[16][CW + 1], pl[32][CW];
  const uint32_t ig = tiles[blockIdx.x].ti, jg = tiles[blockIdx.x].tj;
  const uint32_t ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const uint32_t i = ig * 16 + ty, j = jg * 16 + tx;
  unsigned long long cnt = 0;
  for (uint32_t w0 = 0; w0 < n_words; w0 += CW) {
    for (uint32_t t = threadIdx.x; t < 16 * CW; t += 256) {
      const uint32_t r = t / CW, c = t % CW;
      const uint32_t w = w0 + c;
      const uint32_t ii = ig * 16 + r, jj = jg * 16 + r;
      mi[r][c] = (w < n_words && ii < n_ind) ? mask[(uint64_t)ii * n_words + w] : 0ull;
      mj[r][c] = (w < n_words && jj < n_ind) ? mask[(uint64_t)jj * n_words + w] : 0ull;
    }
    for (uint32_t t = threadIdx.x; t < n_planes * CW; t += 256) {
      const uint32_t b = t / CW, c = t % CW;
      const uint32_t w = w0 + c;
      pl[b][c] = w < n_words ? planes[(uint64_t)b * n_words + w] : 0ull;
    }
    __syncthreads();
    if (n_planes == 0) {
#pragma unroll 8
      for (int c = 0; c < CW; c++) cnt += __popcll(mi[ty][c] & mj[tx][c]);
    } else {
      for (int c = 0; c < CW; c++) {
        const unsigned long long m = mi[ty][c] & mj[tx][c];
        for (uint32_t b = 0; b < n_planes; b++) cnt += (unsigned long long)__popcll(m & pl[b][c]) << b;
      }
    }
    __syncthreads();
  }
  if (i < j && j < n_ind) d_cnt[ngd_pair_idx(n_ind, i, j)] = cnt;
}

}  // namespace

void ngd_launch_reduce(hipStream_t st, const ngd_geom &g, const double *slab, uint32_t n_ks,
                       const ngd_tile *d_tiles, uint32_t n_tiles, double *d_sum) {
  if (!n_tiles) return;
  hipLaunchKernelGGL(k_reduce, dim3(n_tiles * NGD_TILE), dim3(128), 0, st, slab, n_ks, d_tiles, g.n_pad,
                     g.n_ind, d_sum);
}

void ngd_launch_reduce_w(hipStream_t st, const ngd_geom &g, const double *slab, uint32_t n_ks,
                         const double *d_w, const ngd_tile *d_tiles, uint32_t n_tiles, double *d_sum) {
  if (!n_tiles) return;
  hipLaunchKernelGGL(k_reduce_w, dim3(n_tiles * NGD_TILE), dim3(128), 0, st, slab, n_ks, d_w, d_tiles, g.n_pad,
                     g.n_ind, d_sum);
}

void ngd_launch_fill_cnt(hipStream_t st, const ngd_geom &g, const ngd_tile *d_tiles, uint32_t n_tiles,
                         unsigned long long value, unsigned long long *d_cnt) {
  if (!n_tiles) return;
  hipLaunchKernelGGL(k_fill_cnt, dim3(n_tiles * NGD_TILE), dim3(128), 0, st, d_tiles, g.n_ind, value,
                     d_cnt);
}

void ngd_launch_count(hipStream_t st, const ngd_geom &g, const unsigned long long *mask,
                      const unsigned long long *planes, uint32_t n_planes, const ngd_tile *d_tiles16,
                      uint32_t n_tiles16, unsigned long long *d_cnt) {
  if (!n_tiles16) return;
  hipLaunchKernelGGL(k_count, dim3(n_tiles16), dim3(256), 0, st, mask, planes, n_planes, g.n_words,
                     d_tiles16, g.n_ind, d_cnt);
}
