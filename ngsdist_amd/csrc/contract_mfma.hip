// contract_mfma.hip -- bootstrap replicates of the EM path when the blocks are too small for per-block partial
// results (the reference's defaults: no --indep_geno, --boot_block_size 1; parse_args.cpp:29-31, the replicate loop
// ngsDist.cpp:217-289 around rnd_map_data :416-437).
//
// The term c(pair, site) that gen_dist() adds (ngsDist.cpp:351-353 after em2()) does not depend on the replicate; only
// its weight does (the multiplicity of the site's block in the replicate's block map).  So the sums of ALL matrices of
// a job are one contraction over sites,
//     sum[r][pair] = SUM_s W[r][s] * c[pair][s],
// with K = n_sites -- or, since a replicate weights whole blocks of sites, K = the number of UNITS of q consecutive sites
// of one block whose terms are added up before they leave the EM kernel (q = the block size, or a divisor of it: the
// reference's own example of 10-site blocks moves a tenth of the bytes and flops of block size 1).
// accum_em_table.hip (SPILL) writes c for a chunk of units in the fragment-major operand
// layout of ngd_internal.h ("individual" = pair slot, k = unit of the chunk); k_spill_weights writes the chunk's
// weights in the same layout ("individual" = matrix); k_contract_mfma multiplies the two with v_mfma_f64_16x16x4_f64
// (A = weights: M = matrices, B = terms: N = pair slots) and adds the product to the job's running sums D, which a
// wavefront owns for its block of (matrices x pair slots) -- chunks follow each other in stream order, so the order of
// additions is fixed and the result reproducible.  Weights are small integers: exact in FP64.
//
// Registers cap the fused form (one accumulator per matrix in the EM kernel) at 8 matrices per pass; here a pass of the
// EM kernel serves any number, and the contraction costs 2 * R flop per pair-site on the matrix pipe against the ~100
// lane-instructions of the EM itself.
#include <algorithm>

#include "ngd_internal.h"

namespace {

// one operand fragment: 512 B for the wavefront, lane l takes bytes [8l, 8l+8) at base + OFF
template <int OFF>
__device__ __forceinline__ void load_frag(double &dst, uint32_t lane_off, const double *base) {
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(lane_off), "s"(base), "n"(OFF));
}
template <int N, int I = 0>
struct frag_loader {
  static __device__ __forceinline__ void go(double *dst, uint32_t lane_off, const double *base) {
    load_frag<512 * I>(dst[I], lane_off, base);
    if constexpr (I + 1 < N) frag_loader<N, I + 1>::go(dst, lane_off, base);
  }
};

// weight of site s in matrix r of the job: matrix 0 is the full data set when `lead` (1 on every site), the others are
// replicates (multiplicity of the site's block; sites at or beyond n_blocks * block_size are not visited, ngsDist.cpp:236)
__device__ __forceinline__ double job_weight(const uint32_t *__restrict__ mult, uint32_t r, uint32_t n_mat, int lead,
                                             uint64_t s, uint64_t n_sites, uint64_t n_eff, uint64_t n_blocks,
                                             uint64_t block_size) {
  if (r >= n_mat) return 0.0;
  if (lead && r == 0) return s < n_sites ? 1.0 : 0.0;
  if (s >= n_eff) return 0.0;
  return (double)mult[(uint64_t)(r - (lead ? 1u : 0u)) * n_blocks + s / block_size];
}

// Wt[(kg * n_rg + rg) * 64 + (k & 3) * 16 + (r & 15)] = weight of unit k (sites s_lo + k q .. + q - 1, one block) in
// matrix r0 + r, for the chunk's k-groups 0 .. n_kg (one more than the chunk has: the contraction's operand fetch runs
// one k-group ahead)
__global__ __launch_bounds__(256) void k_spill_weights(const uint32_t *__restrict__ mult, uint32_t r0, uint32_t n_mat,
                                                        int lead, uint64_t s_lo, uint64_t s_hi, uint32_t q, uint64_t n_sites,
                                                        uint64_t n_eff, uint64_t n_blocks, uint64_t block_size,
                                                        uint32_t n_rg, uint64_t n_frag, double *__restrict__ Wt) {
  const uint64_t f = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= n_frag) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t kg = f / n_rg;
  const uint32_t rg = (uint32_t)(f % n_rg);
  const uint64_t s = s_lo + (kg * 4 + (lane >> 4)) * q;  // the unit's first site
  const uint32_t r = r0 + rg * 16 + (lane & 15);
  Wt[f * 64 + lane] = s < s_hi ? job_weight(mult, r, n_mat, lead, s, n_sites, n_eff, n_blocks, block_size) : 0.0;
}

// D tile (rg, pg) = 16 matrices x 16 pair slots, stored as the accumulator registers are: [v][lane], matrix
// rg * 16 + (lane >> 4) + 4 v, pair slot pg * 16 + (lane & 15)
__device__ __forceinline__ uint64_t d_tile_off(uint32_t rg, uint64_t pg, uint64_t n_pg) { return ((uint64_t)rg * n_pg + pg) * 256; }

// Wavefront = RT x PT MFMA tiles (RT groups of 16 matrices x PT groups of 16 pair slots), all k-groups of the chunk.
// The operand pipeline is accum_mfma.hip's: hand-issued loads, one k-group in flight per wavefront, three wavefronts
// per SIMD that cover each other's load phases.
template <int RT, int PT>
__global__ __launch_bounds__(256, 3) void k_contract_mfma(const double *__restrict__ Wt, const double *__restrict__ C,
                                                          uint32_t n_rg, uint32_t n_pg, uint32_t n_kg,
                                                          double *__restrict__ D, uint32_t d_rg0) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t pg0 = (blockIdx.x * 4 + wave) * PT;
  if (pg0 >= n_pg) return;
  const uint32_t lane_off = lane * 8;
  // the block's running sums: tile bases are wave-uniform (SGPR base + lane offset, like the operands)
  ngd_d4 acc[RT][PT];
  {
    double t[RT][PT][4];
#pragma unroll
    for (int m = 0; m < RT; m++)
#pragma unroll
      for (int n = 0; n < PT; n++) frag_loader<4>::go(t[m][n], lane_off, D + d_tile_off(d_rg0 + m, pg0 + n, n_pg));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int m = 0; m < RT; m++)
#pragma unroll
      for (int n = 0; n < PT; n++) {
#pragma unroll
        for (int v = 0; v < 4; v++) asm volatile("" : "+v"(t[m][n][v]));  // read only behind the wait
        acc[m][n] = (ngd_d4){t[m][n][0], t[m][n][1], t[m][n][2], t[m][n][3]};
      }
  }
  __builtin_amdgcn_sched_barrier(0);

  const uint64_t sa = (uint64_t)n_rg * 64, sb = (uint64_t)n_pg * 64;  // doubles per k-group
  const double *pa = Wt;
  const double *pb = C + (uint64_t)pg0 * 64;
  double a[RT], b[PT];
  frag_loader<RT>::go(a, lane_off, pa);
  frag_loader<PT>::go(b, lane_off, pb);
  for (uint32_t kg = 0; kg < n_kg; kg++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int m = 0; m < RT; m++) asm volatile("" : "+v"(a[m]));  // pins the MFMAs behind the wait
#pragma unroll
    for (int n = 0; n < PT; n++) asm volatile("" : "+v"(b[n]));
#pragma unroll
    for (int m = 0; m < RT; m++)
#pragma unroll
      for (int n = 0; n < PT; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);  // the refill stays BEHIND the MFMAs that read the registers
    pa += sa;
    pb += sb;
    frag_loader<RT>::go(a, lane_off, pa);  // (the last trip fetches the tail k-group, never consumed)
    frag_loader<PT>::go(b, lane_off, pb);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead: the registers are free again
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < RT; m++)
#pragma unroll
    for (int n = 0; n < PT; n++) {
      double *d = D + d_tile_off(d_rg0 + m, pg0 + n, n_pg);
      asm volatile(
          "global_store_dwordx2 %0, %1, %5\n\t"
          "global_store_dwordx2 %0, %2, %5 offset:512\n\t"
          "global_store_dwordx2 %0, %3, %5 offset:1024\n\t"
          "global_store_dwordx2 %0, %4, %5 offset:1536"
          :
          : "v"(lane_off), "v"((double)acc[m][n][0]), "v"((double)acc[m][n][1]), "v"((double)acc[m][n][2]),
            "v"((double)acc[m][n][3]), "s"(d)
          : "memory");
    }
}

// After a chunk whose spill kernel raised the flag: terms that are not finite become 0 in C and NaN in the running sums
// of exactly the matrices that draw their site (NaN + anything stays NaN through the later chunks).
__global__ __launch_bounds__(256) void k_spill_sanitize(double *__restrict__ C, const unsigned long long *__restrict__ flag,
                                                         uint64_t n_elems, uint32_t n_pg, const uint32_t *__restrict__ mult,
                                                         uint32_t n_mat, int lead, uint64_t s_lo, uint32_t q,
                                                         uint64_t n_sites, uint64_t n_eff, uint64_t n_blocks,
                                                         uint64_t block_size, double *__restrict__ D) {
  if (*flag == 0) return;
  for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n_elems; x += (uint64_t)gridDim.x * blockDim.x) {
    const double c = C[x];
    if (__builtin_fabs(c) <= 1.7976931348623157e308) continue;
    C[x] = 0.0;
    const uint64_t frag = x >> 6;
    const uint32_t l = (uint32_t)(x & 63);
    const uint64_t kg = frag / n_pg, pg = frag % n_pg;
    const uint64_t s = s_lo + (kg * 4 + (l >> 4)) * q;  // the unit's first site: its block is every site's of the unit
    for (uint32_t r = 0; r < n_mat; r++)
      if (job_weight(mult, r, n_mat, lead, s, n_sites, n_eff, n_blocks, block_size) != 0.0)
        D[d_tile_off(r >> 4, pg, n_pg) + ((r & 15) >> 2) * 64 + ((r & 3) << 4) + (l & 15)] = __builtin_nan("");
  }
}

// The job's sums leave D for the caller's [n_mat][n_pairs] arrays in the reference's pair order (ngsDist.cpp:244-245).
// Pair slot of (row, column) of a tile: 16 * (rowpg[tile * 64 + row] + column / 16) + column % 16
__global__ __launch_bounds__(256) void k_spill_scatter(const double *__restrict__ D, uint32_t n_pg,
                                                        const ngd_tile *__restrict__ tiles, const uint32_t *__restrict__ rowpg,
                                                        uint64_t n_ind, uint64_t n_pairs, uint32_t n_mat,
                                                        double *__restrict__ d_sum) {
  const uint32_t tile = blockIdx.x >> 4;
  const uint32_t slot = (blockIdx.x & 15) * 256 + threadIdx.x;  // within the 64 x 64 tile: row * 64 + column
  const uint32_t row = slot >> 6, col = slot & 63;
  const uint64_t i = (uint64_t)tiles[tile].ti * 64 + row, j = (uint64_t)tiles[tile].tj * 64 + col;
  if (i >= j || j >= n_ind) return;
  const uint64_t p = (uint64_t)(uint32_t)((int32_t)rowpg[tile * 64 + row] + (int32_t)(col >> 4)) * 16 + (col & 15);
  const uint64_t out = ngd_pair_idx(n_ind, i, j);
  for (uint32_t r = 0; r < n_mat; r++)
    d_sum[(uint64_t)r * n_pairs + out] =
        D[d_tile_off(r >> 4, p >> 4, n_pg) + ((r & 15) >> 2) * 64 + ((r & 3) << 4) + (uint32_t)(p & 15)];
}

}  // namespace

uint32_t ngd_contract_rep_groups(uint32_t n_mat) { return (n_mat + 15) / 16; }

void ngd_launch_spill_weights(hipStream_t st, const uint32_t *d_mult, uint32_t n_mat, int lead, uint64_t s_lo,
                              uint64_t s_hi, uint32_t q, uint64_t n_sites, uint64_t n_eff, uint64_t n_blocks,
                              uint64_t block_size, double *d_Wt) {
  const uint32_t n_rg = ngd_contract_rep_groups(n_mat);
  const uint64_t n_units = (s_hi - s_lo + q - 1) / q;
  const uint64_t n_kg = (n_units + 3) / 4 + 1;  // + the tail k-group of the operand run-ahead
  const uint64_t n_frag = n_kg * n_rg;
  hipLaunchKernelGGL(k_spill_weights, dim3((unsigned)((n_frag + 3) / 4)), dim3(256), 0, st, d_mult, 0u, n_mat, lead, s_lo,
                     s_hi, q, n_sites, n_eff, n_blocks, block_size, n_rg, n_frag, d_Wt);
}

void ngd_launch_spill_sanitize(hipStream_t st, double *C, const unsigned long long *d_flag, uint64_t n_kg, uint32_t n_pg,
                               const uint32_t *d_mult, uint32_t n_mat, int lead, uint64_t s_lo, uint32_t q, uint64_t n_sites,
                               uint64_t n_eff, uint64_t n_blocks, uint64_t block_size, double *D) {
  hipLaunchKernelGGL(k_spill_sanitize, dim3(4096), dim3(256), 0, st, C, d_flag, n_kg * n_pg * 64, n_pg, d_mult, n_mat, lead,
                     s_lo, q, n_sites, n_eff, n_blocks, block_size, D);
}

// D += Wt x C over the chunk's n_kg k-groups, for all ngd_contract_rep_groups(n_mat) groups of 16 matrices
void ngd_launch_contract(hipStream_t st, const double *d_Wt, const double *C, uint32_t n_mat, uint32_t n_pg, uint32_t n_kg,
                         double *D) {
  const uint32_t n_rg = ngd_contract_rep_groups(n_mat);
  // matrices in batches of up to 8 groups (128): the terms are read once per batch
#define NGD_CT(RT, PT)                                                                                                \
  hipLaunchKernelGGL((k_contract_mfma<RT, PT>), dim3((n_pg / PT + 3) / 4), dim3(256), 0, st, d_Wt + (uint64_t)rg0 * 64, C, \
                     n_rg, n_pg, n_kg, D, rg0)
  for (uint32_t rg0 = 0; rg0 < n_rg; rg0 += 8) {
    switch (std::min(8u, n_rg - rg0)) {
      case 1: NGD_CT(1, 4); break;
      case 2: NGD_CT(2, 4); break;
      case 3: NGD_CT(3, 4); break;
      case 4: NGD_CT(4, 4); break;
      case 5: NGD_CT(5, 2); break;
      case 6: NGD_CT(6, 2); break;
      case 7: NGD_CT(7, 2); break;
      default: NGD_CT(8, 2); break;
    }
  }
#undef NGD_CT
}

void ngd_launch_spill_scatter(hipStream_t st, const double *D, uint32_t n_pg, const ngd_tile *d_tiles64, uint32_t n_tiles64,
                              const uint32_t *d_rowpg, uint64_t n_ind, uint32_t n_mat, double *d_sum) {
  if (!n_tiles64) return;
  hipLaunchKernelGGL(k_spill_scatter, dim3(n_tiles64 * 16), dim3(256), 0, st, D, n_pg, d_tiles64, d_rowpg, n_ind,
                     n_ind * (n_ind - 1) / 2, n_mat, d_sum);
}
