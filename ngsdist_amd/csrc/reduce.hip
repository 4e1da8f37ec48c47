// reduce.hip -- (1) fixed-order reduction of the per-slice slabs into the
// per-pair sums, (2) per-pair valid-site counts for --pairwise_del
// (reference ngsDist.cpp:335-338 + :362: cnt++ only on sites where neither
// individual is missing), from bit masks: cnt = popcount(mask_i & mask_j),
// bootstrap multiplicities entering as bit-planes, (3) bootstrap replicates
// as weighted reductions of per-block partial (sum, cnt), many replicates per
// pass over the partials.
#include "ngd_internal.h"

namespace {

// A block that is NOT drawn (weight 0) must add nothing even when its partial sum is NaN (the EM of an all-zero site,
// 0/0 as on the CPU): 0 x NaN would poison replicates that never visit the site.  A select per (slice, replicate)
// makes this memory-bound kernel VALU-bound ([measured] cfg 5: 0.43 -> 1.31 ms; a branch per slice 0.89 ms), so the main
// loop only NOTES a non-finite partial and a pair that met one is summed again with the undrawn blocks left out.
__device__ __forceinline__ bool ngd_finite(double v) { return __builtin_fabs(v) <= 1.7976931348623157e308; }

// grid = owned 128-tiles x 128 rows; 128 threads = columns.  Eight interleaved partial sums (slice
// ks goes to partial ks % 8) keep eight loads in flight per thread; they are combined in a fixed tree,
// so the result is a fixed function of the slabs (deterministic), just not the left-to-right sum.
// d_cnt != NULL: the pair's count is written too (no --pairwise_del: the same number of sites for every pair,
// ngsDist.cpp:362) -- one launch less per matrix, which a 0.3 ms job notices.
__global__ __launch_bounds__(128) void k_reduce(const double *__restrict__ slab, uint32_t n_ks,
                                                 uint32_t planes_per_slice, const ngd_tile *__restrict__ tiles,
                                                 uint32_t n_pad, uint64_t n_ind, double *__restrict__ d_sum,
                                                 unsigned long long *__restrict__ d_cnt, unsigned long long cnt_value,
                                                 ngd_fix_flags fix, double fix_thr) {
  const uint32_t tile = blockIdx.x >> 7, row = blockIdx.x & 127;
  const uint32_t i = tiles[tile].ti * NGD_TILE + row;
  const uint32_t j = tiles[tile].tj * NGD_TILE + threadIdx.x;
  if (!(i < j && j < n_ind)) return;
  const uint64_t plane = (uint64_t)n_pad * n_pad * planes_per_slice;  // distance between consecutive slices
  const double *p = slab + (uint64_t)i * n_pad + j;
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t ks = 0;
  for (; ks + 8 <= n_ks; ks += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) s[u] += p[(uint64_t)(ks + u) * plane];
  }
  for (int u = 0; ks < n_ks; ks++, u++) s[u] += p[(uint64_t)ks * plane];
  const uint64_t idx = ngd_pair_idx(n_ind, i, j);
  const double sum = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  d_sum[idx] = sum;
  if (d_cnt) d_cnt[idx] = cnt_value;
  // single_image = 2 engines: a sum this small is not held to 1e-9 relative by the congruent arithmetic -- noted for the
  // fix-up pass (fixup.hip); rare, so one atomic per noted pair
  if (fix.list && sum < fix_thr) {
    const uint32_t slot = atomicAdd(fix.count, 1u);
    if (slot < fix.cap) fix.list[slot] = ((unsigned long long)i << 32) | j;
  }
}

// Bootstrap replicates from per-block partial sums, RB replicates per pass over the slab:
//   sum[r][pair] = SUM_slice W[slice][r] * slab[slice][pair],  slices in ascending order,
// W = multiplicity of the slice's block in replicate r (reference rnd_map_data, ngsDist.cpp:416-437: a
// block drawn m times is visited m times).  W is slice-major (stride w_stride, zero padded to a multiple
// of RB), so the RB weights of a slice are one scalar load; grid.y = replicate chunk.  One order of
// summation for every replicate and every batch size: a replicate's result does not depend on which
// call (single or batched) produced it.
template <int RB>
__global__ __launch_bounds__(128) void k_reduce_wb(const double *__restrict__ slab, uint32_t n_ks,
                                                    const double *__restrict__ W, uint32_t w_stride,
                                                    uint32_t n_rep, const ngd_tile *__restrict__ tiles,
                                                    uint32_t n_pad, uint64_t n_ind, uint64_t n_pairs,
                                                    double *__restrict__ d_sum, ngd_fix_flags fix,
                                                    const double *__restrict__ fix_thr) {
  const uint32_t tile = blockIdx.x >> 7, row = blockIdx.x & 127;
  const uint32_t i = tiles[tile].ti * NGD_TILE + row;
  const uint32_t j = tiles[tile].tj * NGD_TILE + threadIdx.x;
  if (!(i < j && j < n_ind)) return;
  const uint32_t r0 = blockIdx.y * RB;
  const uint64_t plane = (uint64_t)n_pad * n_pad;
  const double *p = slab + (uint64_t)i * n_pad + j;
  const double *w = W + r0;
  double acc[RB];
#pragma unroll
  for (int r = 0; r < RB; r++) acc[r] = 0;
  if (RB == 1) {  // one replicate: about 1/e of the blocks are not drawn at all -> their slices are not read
    for (uint32_t ks = 0; ks < n_ks; ks++) {
      const double wk = w[(uint64_t)ks * w_stride];  // uniform across the workgroup
      if (wk != 0.0) acc[0] = __builtin_fma(wk, p[ks * plane], acc[0]);
    }
  } else {
    constexpr int U = 4;  // slices in flight per thread
    uint32_t ks = 0;
    bool bad = false;  // a non-finite partial among this pair's slices
    for (; ks + U <= n_ks; ks += U) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = p[(uint64_t)(ks + u) * plane];
#pragma unroll
      for (int u = 0; u < U; u++) bad |= !ngd_finite(v[u]);
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int r = 0; r < RB; r++) acc[r] = __builtin_fma(w[(uint64_t)(ks + u) * w_stride + r], v[u], acc[r]);
    }
    for (; ks < n_ks; ks++) {
      const double v = p[(uint64_t)ks * plane];
      bad |= !ngd_finite(v);
#pragma unroll
      for (int r = 0; r < RB; r++) acc[r] = __builtin_fma(w[(uint64_t)ks * w_stride + r], v, acc[r]);
    }
    if (__builtin_expect(bad, 0)) {  // again, slices in the same order, blocks that are not drawn left out
#pragma unroll
      for (int r = 0; r < RB; r++) acc[r] = 0;
      for (ks = 0; ks < n_ks; ks++) {
        const double v = p[(uint64_t)ks * plane];
#pragma unroll
        for (int r = 0; r < RB; r++) {
          const double wr = w[(uint64_t)ks * w_stride + r];
          if (wr != 0.0) acc[r] = __builtin_fma(wr, v, acc[r]);
        }
      }
    }
  }
  const uint64_t idx = ngd_pair_idx(n_ind, i, j);
  bool small = false;
#pragma unroll
  for (int r = 0; r < RB; r++)
    if (r0 + r < n_rep) {
      d_sum[(uint64_t)(r0 + r) * n_pairs + idx] = acc[r];
      if (fix.list) small = small || acc[r] < fix_thr[r0 + r];
    }
  // single_image = 2 engines (see k_reduce): noted once, whichever replicate chunk sees a small sum first
  if (small && !((atomicOr(&fix.seen[idx >> 5], 1u << (idx & 31)) >> (idx & 31)) & 1u)) {
    const uint32_t slot = atomicAdd(fix.count, 1u);
    if (slot < fix.cap) fix.list[slot] = ((unsigned long long)i << 32) | j;
  }
}

// single_image = 2 engines under --pairwise_del (ngsDist.cpp:335-338): which pairs want the fix-up pass is decided HERE,
// once a pair's valid-site counts are known, not in the reductions above -- a pair that shares no valid site with its
// partner in a matrix has the sum 0 there, exactly, and must not be noted (a data set with many missing sites would
// otherwise note thousands of empty pairs, pass the engine's limit and have the fix-up pass skipped for the few pairs that need
// it: tools/fuzz_parity.py cases 202651, 202706, 202716, 207606).  The bound of the congruent arithmetic is per VISITED
// site, so the threshold is per pair too: noted if, in any of the n_rep matrices, 0 < cnt and sum < mean x cnt.
__global__ __launch_bounds__(128) void k_fix_flag(const double *__restrict__ d_sum, const unsigned long long *__restrict__ d_cnt,
                                                   uint32_t n_rep, const ngd_tile *__restrict__ tiles, uint64_t n_ind,
                                                   uint64_t n_pairs, double mean, ngd_fix_flags fix) {
  const uint32_t tile = blockIdx.x >> 7, row = blockIdx.x & 127;
  const uint32_t i = tiles[tile].ti * NGD_TILE + row;
  const uint32_t j = tiles[tile].tj * NGD_TILE + threadIdx.x;
  if (!(i < j && j < n_ind)) return;
  const uint64_t idx = ngd_pair_idx(n_ind, i, j);
  bool small = false;
  for (uint32_t r = 0; r < n_rep; r++) {
    const unsigned long long c = d_cnt[(uint64_t)r * n_pairs + idx];
    small = small || (c != 0 && d_sum[(uint64_t)r * n_pairs + idx] < mean * (double)c);
  }
  if (small) {
    const uint32_t slot = atomicAdd(fix.count, 1u);
    if (slot < fix.cap) fix.list[slot] = ((unsigned long long)i << 32) | j;
  }
}

// The fix-up pass by a whole two-operand pass (engine.hip fixup_by_pass): d_new holds every pair's sum in the two-image
// arithmetic; it replaces the one-image sum of exactly the pairs the noting rule picks (sum below mean x the sites visited;
// under --pairwise_del x the pair's own count, never a pair without a valid site) -- all others keep their bits.
__global__ __launch_bounds__(128) void k_fix_merge(const double *__restrict__ d_new, double *__restrict__ d_sum,
                                                    const unsigned long long *__restrict__ d_cnt, double thr, double mean,
                                                    const ngd_tile *__restrict__ tiles, uint64_t n_ind) {
  const uint32_t tile = blockIdx.x >> 7, row = blockIdx.x & 127;
  const uint32_t i = tiles[tile].ti * NGD_TILE + row;
  const uint32_t j = tiles[tile].tj * NGD_TILE + threadIdx.x;
  if (!(i < j && j < n_ind)) return;
  const uint64_t idx = ngd_pair_idx(n_ind, i, j);
  const double old = d_sum[idx];
  bool small;
  if (d_cnt) {
    const unsigned long long c = d_cnt[idx];
    small = c != 0 && old < mean * (double)c;
  } else {
    small = old < thr;
  }
  if (small) d_sum[idx] = d_new[idx];
}

// The same for the valid-site counts of --pairwise_del: cnt[r][pair] = SUM_b M[b][r] * C[b][pair] with
// C = per-block popcounts (k_count_blocks) and M the block multiplicities (uint32, block-major).
template <int RB>
__global__ __launch_bounds__(128) void k_reduce_cb(const uint32_t *__restrict__ C, uint32_t n_blocks,
                                                    const uint32_t *__restrict__ M, uint32_t m_stride,
                                                    uint32_t n_rep, const ngd_tile *__restrict__ tiles,
                                                    uint32_t n_pad, uint64_t n_ind, uint64_t n_pairs,
                                                    unsigned long long *__restrict__ d_cnt) {
  const uint32_t tile = blockIdx.x >> 7, row = blockIdx.x & 127;
  const uint32_t i = tiles[tile].ti * NGD_TILE + row;
  const uint32_t j = tiles[tile].tj * NGD_TILE + threadIdx.x;
  if (!(i < j && j < n_ind)) return;
  const uint32_t r0 = blockIdx.y * RB;
  const uint64_t plane = (uint64_t)n_pad * n_pad;
  const uint32_t *p = C + (uint64_t)i * n_pad + j;
  const uint32_t *m = M + r0;
  unsigned long long acc[RB];
#pragma unroll
  for (int r = 0; r < RB; r++) acc[r] = 0;
  constexpr int U = 4;
  uint32_t b = 0;
  for (; b + U <= n_blocks; b += U) {
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = p[(uint64_t)(b + u) * plane];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int r = 0; r < RB; r++) acc[r] += (unsigned long long)m[(uint64_t)(b + u) * m_stride + r] * v[u];
  }
  for (; b < n_blocks; b++) {
    const uint32_t v = p[(uint64_t)b * plane];
#pragma unroll
    for (int r = 0; r < RB; r++) acc[r] += (unsigned long long)m[(uint64_t)b * m_stride + r] * v;
  }
  const uint64_t idx = ngd_pair_idx(n_ind, i, j);
#pragma unroll
  for (int r = 0; r < RB; r++)
    if (r0 + r < n_rep) d_cnt[(uint64_t)(r0 + r) * n_pairs + idx] = acc[r];
}

// grid.y = replicate: cnt[r][pair] = values[r] (no --pairwise_del: every visited site counts, ngsDist.cpp:362)
__global__ __launch_bounds__(128) void k_fill_cnt(const ngd_tile *__restrict__ tiles, uint64_t n_ind,
                                                   unsigned long long value,
                                                   const unsigned long long *__restrict__ values,
                                                   uint64_t n_pairs, unsigned long long *__restrict__ d_cnt) {
  const uint32_t tile = blockIdx.x >> 7, row = blockIdx.x & 127;
  const uint32_t i = tiles[tile].ti * NGD_TILE + row;
  const uint32_t j = tiles[tile].tj * NGD_TILE + threadIdx.x;
  if (!(i < j && j < n_ind)) return;
  d_cnt[(uint64_t)blockIdx.y * n_pairs + ngd_pair_idx(n_ind, i, j)] = values ? values[blockIdx.y] : value;
}

constexpr int CW = 64;  // mask words staged per step

// Workgroup = 32 x 32 pairs of one 128-tile (grid = owned 128-tiles x 16 sub-tiles), thread = 2 x 2 pairs: four
// LDS reads feed four popcounts (the one-pair-per-thread form was LDS-bandwidth bound: two reads per popcount).
// LDS holds CW words of the 32 + 32 individuals (and of the planes).
__global__ __launch_bounds__(256) void k_count(const unsigned long long *__restrict__ mask,
                                                const unsigned long long *__restrict__ planes,
                                                uint32_t n_planes, uint32_t n_words,
                                                const ngd_tile *__restrict__ tiles, uint64_t n_ind,
                                                unsigned long long *__restrict__ d_cnt) {
  __shared__ unsigned long long mi[32][CW + 1], mj[32][CW + 1], pl[32][CW];
  const uint32_t tile = blockIdx.x >> 4, sa = (blockIdx.x >> 2) & 3, sb = blockIdx.x & 3;
  const uint32_t ti = tiles[tile].ti, tj = tiles[tile].tj;
  if (ti == tj && sa > sb) return;  // below the diagonal
  const uint32_t i0 = ti * NGD_TILE + sa * 32, j0 = tj * NGD_TILE + sb * 32;
  if (i0 >= n_ind || j0 >= n_ind) return;
  const uint32_t ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  unsigned long long cnt[2][2] = {{0, 0}, {0, 0}};
  // grid.y splits the words (sites): integer atomics make the sum independent of the order
  const uint32_t per_y = ((n_words + gridDim.y - 1) / gridDim.y + CW - 1) / CW * CW;
  const uint32_t w_begin = blockIdx.y * per_y, w_end = w_begin + per_y < n_words ? w_begin + per_y : n_words;
  for (uint32_t w0 = w_begin; w0 < w_end; w0 += CW) {
    for (uint32_t t = threadIdx.x; t < 32 * CW; t += 256) {
      const uint32_t r = t / CW, c = t % CW;
      const uint32_t w = w0 + c;
      const uint32_t ii = i0 + r, jj = j0 + r;
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
      uint32_t q00 = 0, q01 = 0, q10 = 0, q11 = 0;  // at most 64 x CW per chunk: 32 bits are plenty
#pragma unroll 4
      for (int c = 0; c < CW; c++) {
        const unsigned long long a0 = mi[ty][c], a1 = mi[ty + 16][c], b0 = mj[tx][c], b1 = mj[tx + 16][c];
        q00 += __popcll(a0 & b0); q01 += __popcll(a0 & b1);
        q10 += __popcll(a1 & b0); q11 += __popcll(a1 & b1);
      }
      cnt[0][0] += q00; cnt[0][1] += q01; cnt[1][0] += q10; cnt[1][1] += q11;
    } else {
      for (int c = 0; c < CW; c++) {
        const unsigned long long a0 = mi[ty][c], a1 = mi[ty + 16][c], b0 = mj[tx][c], b1 = mj[tx + 16][c];
        const unsigned long long m00 = a0 & b0, m01 = a0 & b1, m10 = a1 & b0, m11 = a1 & b1;
        for (uint32_t b = 0; b < n_planes; b++) {
          const unsigned long long p = pl[b][c];
          cnt[0][0] += (unsigned long long)__popcll(m00 & p) << b; cnt[0][1] += (unsigned long long)__popcll(m01 & p) << b;
          cnt[1][0] += (unsigned long long)__popcll(m10 & p) << b; cnt[1][1] += (unsigned long long)__popcll(m11 & p) << b;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const uint32_t i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
      if (i < j && j < n_ind) atomicAdd(&d_cnt[ngd_pair_idx(n_ind, i, j)], cnt[a][b]);  // d_cnt was zeroed
    }
}

// Per-block valid-site counts C[b][i][j] = popcount(mask_i & mask_j over the sites of block b): the cnt half
// of the per-block partials.  Workgroup = 16x16 pairs; grid.y strides over the blocks.  A block is any
// site range [b*B, (b+1)*B), so its first and last mask words are trimmed.
__global__ __launch_bounds__(256) void k_count_blocks(const unsigned long long *__restrict__ mask,
                                                       uint32_t n_words, uint64_t block_size, uint32_t n_blocks,
                                                       const ngd_tile *__restrict__ tiles, uint32_t n_pad,
                                                       uint64_t n_ind, uint32_t *__restrict__ C) {
  __shared__ unsigned long long mi[16][CW + 1], mj[16][CW + 1];
  const uint32_t ig = tiles[blockIdx.x].ti, jg = tiles[blockIdx.x].tj;
  const uint32_t ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const uint32_t i = ig * 16 + ty, j = jg * 16 + tx;
  for (uint32_t b = blockIdx.y; b < n_blocks; b += gridDim.y) {
    const uint64_t s_lo = (uint64_t)b * block_size, s_hi = s_lo + block_size;  // sites [s_lo, s_hi)
    const uint32_t w_lo = (uint32_t)(s_lo >> 6), w_hi = (uint32_t)((s_hi - 1) >> 6);
    uint32_t cnt = 0;
    for (uint32_t w0 = w_lo; w0 <= w_hi; w0 += CW) {
      for (uint32_t t = threadIdx.x; t < 16 * CW; t += 256) {
        const uint32_t r = t / CW, c = t % CW;
        const uint32_t w = w0 + c;
        unsigned long long keep = 0;
        if (w <= w_hi) {
          keep = ~0ull;
          if (w == w_lo) keep &= ~0ull << (s_lo & 63);
          if (w == w_hi && (s_hi & 63)) keep &= ~0ull >> (64 - (s_hi & 63));
        }
        const uint32_t ii = ig * 16 + r, jj = jg * 16 + r;
        mi[r][c] = (keep && ii < n_ind) ? mask[(uint64_t)ii * n_words + w] & keep : 0ull;
        mj[r][c] = (keep && jj < n_ind) ? mask[(uint64_t)jj * n_words + w] : 0ull;
      }
      __syncthreads();
      const uint32_t left = w_hi - w0 + 1;
      const uint32_t nw = left < (uint32_t)CW ? left : (uint32_t)CW;
      for (uint32_t c = 0; c < nw; c++) cnt += __popcll(mi[ty][c] & mj[tx][c]);
      __syncthreads();
    }
    C[((uint64_t)b * n_pad + i) * n_pad + j] = cnt;
  }
}

template <int RB>
void reduce_wb(hipStream_t st, const ngd_geom &g, const double *slab, uint32_t n_ks, const double *d_W,
               uint32_t w_stride, uint32_t n_rep, const ngd_tile *d_tiles, uint32_t n_tiles, double *d_sum,
               const ngd_fix_flags &fix, const double *d_thr) {
  hipLaunchKernelGGL((k_reduce_wb<RB>), dim3(n_tiles * NGD_TILE, (n_rep + RB - 1) / RB), dim3(128), 0, st, slab,
                     n_ks, d_W, w_stride, n_rep, d_tiles, g.n_pad, g.n_ind, g.n_ind * (g.n_ind - 1) / 2, d_sum, fix, d_thr);
}

template <int RB>
void reduce_cb(hipStream_t st, const ngd_geom &g, const uint32_t *C, uint32_t n_blocks, const uint32_t *d_M,
               uint32_t m_stride, uint32_t n_rep, const ngd_tile *d_tiles, uint32_t n_tiles,
               unsigned long long *d_cnt) {
  hipLaunchKernelGGL((k_reduce_cb<RB>), dim3(n_tiles * NGD_TILE, (n_rep + RB - 1) / RB), dim3(128), 0, st, C,
                     n_blocks, d_M, m_stride, n_rep, d_tiles, g.n_pad, g.n_ind, g.n_ind * (g.n_ind - 1) / 2, d_cnt);
}

}  // namespace

void ngd_launch_reduce(hipStream_t st, const ngd_geom &g, const double *slab, uint32_t n_ks,
                       uint32_t planes_per_slice, const ngd_tile *d_tiles, uint32_t n_tiles, double *d_sum,
                       unsigned long long *d_cnt, unsigned long long cnt_value, const ngd_fix_flags *fix, double fix_thr) {
  if (!n_tiles) return;
  hipLaunchKernelGGL(k_reduce, dim3(n_tiles * NGD_TILE), dim3(128), 0, st, slab, n_ks, planes_per_slice, d_tiles,
                     g.n_pad, g.n_ind, d_sum, d_cnt, cnt_value, fix ? *fix : ngd_fix_flags{nullptr, nullptr, nullptr, 0}, fix_thr);
}

void ngd_launch_fix_flag(hipStream_t st, const ngd_geom &g, const double *d_sum, const unsigned long long *d_cnt,
                         uint32_t n_rep, const ngd_tile *d_tiles, uint32_t n_tiles, const ngd_fix_flags &fix) {
  if (!n_tiles || !n_rep) return;
  hipLaunchKernelGGL(k_fix_flag, dim3(n_tiles * NGD_TILE), dim3(128), 0, st, d_sum, d_cnt, n_rep, d_tiles, g.n_ind,
                     ngd_n_pairs(g.n_ind), NGD_FIX_MEAN, fix);
}

void ngd_launch_fix_merge(hipStream_t st, const ngd_geom &g, const double *d_new, double *d_sum, const unsigned long long *d_cnt,
                          double thr, const ngd_tile *d_tiles, uint32_t n_tiles) {
  if (!n_tiles) return;
  hipLaunchKernelGGL(k_fix_merge, dim3(n_tiles * NGD_TILE), dim3(128), 0, st, d_new, d_sum, d_cnt, thr, NGD_FIX_MEAN, d_tiles,
                     g.n_ind);
}

// replicates per pass over the partials; the weight arrays are padded to a multiple of it
uint32_t ngd_reduce_chunk(uint32_t n_rep) { return n_rep <= 1 ? 1 : n_rep <= 4 ? 4 : n_rep <= 16 ? 16 : 32; }

void ngd_launch_reduce_w(hipStream_t st, const ngd_geom &g, const double *slab, uint32_t n_ks, const double *d_W,
                         uint32_t w_stride, uint32_t n_rep, const ngd_tile *d_tiles, uint32_t n_tiles,
                         double *d_sum, const ngd_fix_flags *fix, const double *d_thr, uint32_t chunk) {
  if (!n_tiles || !n_rep) return;
  const ngd_fix_flags f = fix && d_thr ? *fix : ngd_fix_flags{nullptr, nullptr, nullptr, 0};
  // chunk: one group of a larger job launched on its own (the caller has offset W, d_sum and d_thr to the group's first
  // replicate) keeps the job's template -- a replicate's bits do not depend on it, its speed does
  switch (chunk ? chunk : ngd_reduce_chunk(n_rep)) {
    case 1: reduce_wb<1>(st, g, slab, n_ks, d_W, w_stride, n_rep, d_tiles, n_tiles, d_sum, f, d_thr); break;
    case 4: reduce_wb<4>(st, g, slab, n_ks, d_W, w_stride, n_rep, d_tiles, n_tiles, d_sum, f, d_thr); break;
    case 16: reduce_wb<16>(st, g, slab, n_ks, d_W, w_stride, n_rep, d_tiles, n_tiles, d_sum, f, d_thr); break;
    default: reduce_wb<32>(st, g, slab, n_ks, d_W, w_stride, n_rep, d_tiles, n_tiles, d_sum, f, d_thr);
  }
}

void ngd_launch_reduce_c(hipStream_t st, const ngd_geom &g, const uint32_t *C, uint32_t n_blocks, const uint32_t *d_M,
                         uint32_t m_stride, uint32_t n_rep, const ngd_tile *d_tiles, uint32_t n_tiles,
                         unsigned long long *d_cnt, uint32_t chunk) {
  if (!n_tiles || !n_rep) return;
  switch (chunk ? chunk : ngd_reduce_chunk(n_rep)) {
    case 1: reduce_cb<1>(st, g, C, n_blocks, d_M, m_stride, n_rep, d_tiles, n_tiles, d_cnt); break;
    case 4: reduce_cb<4>(st, g, C, n_blocks, d_M, m_stride, n_rep, d_tiles, n_tiles, d_cnt); break;
    case 16: reduce_cb<16>(st, g, C, n_blocks, d_M, m_stride, n_rep, d_tiles, n_tiles, d_cnt); break;
    default: reduce_cb<32>(st, g, C, n_blocks, d_M, m_stride, n_rep, d_tiles, n_tiles, d_cnt);
  }
}

void ngd_launch_count_blocks(hipStream_t st, const ngd_geom &g, const unsigned long long *mask, uint64_t block_size,
                             uint32_t n_blocks, const ngd_tile *d_tiles16, uint32_t n_tiles16, uint32_t *C) {
  if (!n_tiles16 || !n_blocks) return;
  hipLaunchKernelGGL(k_count_blocks, dim3(n_tiles16, n_blocks < 1024 ? n_blocks : 1024), dim3(256), 0, st, mask,
                     g.n_words, block_size, n_blocks, d_tiles16, g.n_pad, g.n_ind, C);
}

void ngd_launch_fill_cnt(hipStream_t st, const ngd_geom &g, const ngd_tile *d_tiles, uint32_t n_tiles,
                         unsigned long long value, const unsigned long long *d_values, uint32_t n_rep,
                         unsigned long long *d_cnt) {
  if (!n_tiles || !n_rep) return;
  const uint64_t n_pairs = g.n_ind * (g.n_ind - 1) / 2;
  for (uint32_t r0 = 0; r0 < n_rep; r0 += 32768) {  // (grid.y holds 65 535)
    const uint32_t n = n_rep - r0 < 32768 ? n_rep - r0 : 32768;
    hipLaunchKernelGGL(k_fill_cnt, dim3(n_tiles * NGD_TILE, n), dim3(128), 0, st, d_tiles, g.n_ind, value,
                       d_values ? d_values + r0 : nullptr, n_pairs, d_cnt + (uint64_t)r0 * n_pairs);
  }
}

void ngd_launch_count(hipStream_t st, const ngd_geom &g, const unsigned long long *mask,
                      const unsigned long long *planes, uint32_t n_planes, const ngd_tile *d_tiles,
                      uint32_t n_tiles, unsigned long long *d_cnt) {
  if (!n_tiles) return;
  // enough workgroups to fill the chip: split the sites when the pair tiles alone are too few
  uint32_t ny = 1;
  while ((uint64_t)n_tiles * 12 * ny < 8192 && (uint64_t)ny * 2 * CW <= g.n_words) ny *= 2;
  hipLaunchKernelGGL(k_count, dim3(n_tiles * 16, ny), dim3(256), 0, st, mask, planes, n_planes, g.n_words, d_tiles,
                     g.n_ind, d_cnt);
}
