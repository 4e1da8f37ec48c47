// Internal declarations shared by the engine and its kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ngsdist_amd.h"

#define NGD_TILE 128     // pair tile edge owned by one workgroup of the MFMA kernel
#define NGD_IG 16        // individuals per fragment group (one MFMA operand edge)
#define NGD_IG_PER_TILE (NGD_TILE / NGD_IG)
#define NGD_KG_TAIL 8    // zeroed k-groups appended to the operand images (pipeline run-ahead)
#define NGD_KG_LIST_PAD 16  // entries appended to a k-group list, all pointing at the first tail k-group

typedef double ngd_d4 __attribute__((ext_vector_type(4)));

struct ngd_score {
  double v[9];
  // single_image = 2 (host_util.cpp ngd_score_congruence()): score = SUM_r d[r] c_r c_r^T.  The ONE operand image then holds
  // t_r = c_r . p per site instead of p, both operands of the MFMA kernel are read from it, and d[r] rides on the
  // per-index weights.  congruent = 0: the image holds p.
  double c[9], d[3];
  int congruent;
  // congruent images of the reference's two matrices (parse_args.cpp:25-27, :134-137) hold t = (p0 + p1 + p2, +-(p2 - p0),
  // p1): with min(p0, p2) kept beside the image (one double per individual and site), p comes back to the last bit or
  // so -- what the fix-up pass of nearly identical pairs recomputes from (fixup.hip).  fix = 1: the image has that form
  // (fix_sign = the sign of p2 in t_1); 0: any other symmetric matrix, no fix-up.
  int fix;
  double fix_sign;
};

// Geometry of the resident data set.
//
// Fragment-major operand layout ("PA", and "QB" = score-weighted copy):
//   the contraction index is k = 3*s + g  (site s, genotype g);
//   element (i, k) lives at  ((k/4)*n_ig + i/16)*64 + (k%4)*16 + i%16.
// One 64-double group (512 B) is exactly the per-lane operand image of
// v_mfma_f64_16x16x4_f64 (lane l <-> individual l&15, k-offset l>>4), so a
// wavefront fetches an operand with one fully coalesced 512-B load and 16
// consecutive individuals of one (site, genotype) are 128 contiguous bytes.
struct ngd_geom {
  uint64_t n_ind;
  uint64_t n_sites;
  uint64_t n_sites_pad;  // multiple of 16 -> n_kg multiple of 12
  uint64_t n_kg;         // k groups of 4 = 3*n_sites_pad/4
  uint32_t n_ig;         // individual groups of 16, padded to a multiple of 8
  uint32_t n_t;          // 128-individual tiles per edge
  uint32_t n_pad;        // n_t * 128
  uint32_t n_words;      // 64-site mask words per individual
};

__host__ __device__ inline uint64_t ngd_frag_off(uint64_t k, uint32_t i, uint32_t n_ig) {
  return ((k >> 2) * n_ig + (i >> 4)) * 64 + (k & 3) * 16 + (i & 15);
}

// row-major upper-triangle pair index, ngsDist.cpp:244-245
__host__ __device__ inline uint64_t ngd_pair_idx(uint64_t n, uint64_t i, uint64_t j) {
  return i * (2 * n - i - 1) / 2 + (j - i - 1);
}

// miss_data(), reference gen_func.cpp:862-868 (EPSILON = 1e-5, gen_func.hpp:16)
__host__ __device__ inline bool ngd_miss(double p0, double p1, double p2) {
  double a = p0 - p1, b = p1 - p2;
  a = (a >= 0 ? a : -a);
  b = (b >= 0 ? b : -b);
  return a < 1e-5 && b < 1e-5;
}

struct ngd_tile {
  uint16_t ti, tj;  // tile coordinates (units depend on the list: 128 or 16 individuals)
};

// One wavefront's work in the MFMA kernel: the block of pairs whose first row / column groups (of 16
// individuals) are ig0 / jg0, rows x cols MFMA tiles (1..4 each), upper triangle only if tri.
// rows == 0 marks a padding entry of the list.
struct ngd_job {
  uint16_t ig0, jg0;
  uint8_t rows, cols, tri, pad;
};

// The block shapes accum_mfma.hip has a code path for, per block form (ngd_engine::exact_shapes: 0 full 4 x 4 pattern;
// 1 / 3 blocks of up to 4 x 4 tiles, any shape falls back to the full pattern; 2 / 4 / 5 blocks of up to 2 x 4).
// ngd_create() checks the job list it built against this; the kernel poisons (NaN) a block whose shape it does not list.
inline bool ngd_mfma_shape_listed(int form, uint32_t rows, uint32_t cols, uint32_t tri) {
  if (rows == 0) return true;  // padding entry
  if (form == 0 || form == 1 || form == 3) return rows <= 4 && cols <= 4;
  if (tri) return (rows == 2 && cols >= 2 && cols <= 4) || (rows == 1 && cols == 1);
  return rows == 2 && cols >= 1 && cols <= 4;
}

// ---- kernel launchers (each in its own .hip file) -------------------------
// layout.hip
// (PI: the individual-major copy of the streaming kernel -- or, with score.congruent and score.fix, the side array
// SM[site][individual] = min(p0, p2) of the fix-up pass)
void ngd_launch_layout(hipStream_t st, const ngd_geom &g, const double *raw, int raw_ind_major,
                       uint64_t s0, uint64_t n_sites_chunk, const ngd_score &score, int pairwise_del,
                       double *PA, double *QB, double *PI, unsigned long long *mask);
// single-image engines: QB of k-groups [kg_lo, kg_hi) from PA, element (i, k) at ngd_frag_off(k, i) - kg_lo * n_ig * 64
void ngd_launch_qb_range(hipStream_t st, const ngd_geom &g, const ngd_score &score, const double *PA, uint64_t kg_lo,
                         uint64_t kg_hi, double *QBs);
void ngd_launch_prep_layout(hipStream_t st, const ngd_geom &g, const double *raw, uint64_t s0, uint64_t n_chunk,
                            int in_logscale, int call_geno, double N_thresh, double call_thresh,
                            const ngd_score &score, int pairwise_del, double *PA, double *QB, double *PI,
                            unsigned long long *mask, int *nan_flag);
void ngd_launch_synth(hipStream_t st, const ngd_geom &g, uint64_t seed, double miss_frac, uint64_t site0,
                      const ngd_score &score, int pairwise_del, double *PA, double *QB, double *PI,
                      unsigned long long *mask);
void ngd_launch_weights(hipStream_t st, uint64_t n_blocks, uint64_t block_size, uint64_t n_sites,
                        const uint32_t *d_mult, uint32_t *d_ws, double *d_wk, const double *d3 = nullptr);
// list of the k-groups with a non-zero bootstrap weight (ascending), NGD_KG_LIST_PAD entries of padding;
// d_counts: ngd_kg_count_blocks(n_kg) + 1 words of scratch, the last one receives the list length
uint32_t ngd_kg_count_blocks(uint64_t n_kg);
void ngd_launch_kg_compact(hipStream_t st, const double *d_wk, uint64_t n_kg, uint32_t tail_kg, uint32_t *d_counts,
                           uint32_t *d_list);
void ngd_launch_weights_batch(hipStream_t st, const uint32_t *d_mult, uint32_t n_rep, uint32_t rb, int lead_full,
                              uint64_t n_blocks, uint64_t block_size, uint64_t n_sites, uint64_t n_sites_alloc,
                              double *d_W);
// W[slice][j][c] = 1 if contraction index 4 (kg0(slice) + j) + c lies in [slice k_per_slice, (slice+1) k_per_slice)
// and below k_total, else 0; kg0(slice) = slice * k_per_slice / 4; j < stride
void ngd_launch_slice_weights(hipStream_t st, uint32_t n_slices, uint32_t stride, uint64_t k_per_slice, uint64_t k_total,
                              double *d_W, const double *d3 = nullptr /* weight of index k inside a slice: d3[k % 3], not 1 */);
// single_image = 2: the weights of a plain pass, d3[k % 3] for every contraction index of the images (+ tail)
void ngd_launch_index_weights(hipStream_t st, uint64_t n_k, const double *d3, double *d_W);
void ngd_launch_planes(hipStream_t st, const uint32_t *d_ws, uint64_t n_sites, uint32_t n_words,
                       uint32_t n_planes, unsigned long long *d_planes);

// accum_stream.hip : one wavefront per pair
void ngd_launch_accum_stream(hipStream_t st, const ngd_geom &g, const double *PI, const uint32_t *d_ws,
                             uint64_t n_sites_eff, const ngd_score &score, int pairwise_del,
                             const uint64_t *d_pairs, uint64_t n_owned, double *d_sum);

// accum_mfma.hip : FP64 MFMA tiles, split over site slices into slabs
// d_wk / d_kgl: bootstrap weights per contraction index and the list of k-groups to visit; with a list,
// kg_per_slice and n_kg_eff count LIST entries.  k_per_slice != 0: slices are k_per_slice contraction indices
// (not whole k-groups) and d_wk holds w_slice_stride k-groups of 0/1 weights PER SLICE (ngd_launch_slice_weights).
void ngd_launch_accum_mfma(hipStream_t st, const ngd_geom &g, const double *PA, const double *QB,
                           const double *d_wk, const uint32_t *d_kgl, const ngd_job *d_jobs, uint32_t n_wg,
                           int exact_shapes, uint32_t wg_waves, uint32_t n_ks, uint64_t kg_per_slice, uint64_t n_kg_eff,
                           uint64_t k_per_slice, uint32_t w_slice_stride, double *slab,
                           unsigned long long *d_clk /* [2] or NULL: shader-cycle / constant-rate counter deltas of one wavefront */,
                           uint32_t ks0 = 0 /* the launch covers slices ks0 .. ks0 + n_ks - 1 (multiples of 8) */,
                           uint32_t resume = 0 /* 1: every block continues from its plane of the slab (a pass in ranges) */);

// accum_em.hip : per-site EM, one thread per pair of a 16x16 tile
void ngd_launch_accum_em(hipStream_t st, const ngd_geom &g, const double *PA, const uint32_t *d_ws,
                         uint64_t n_sites_eff, const ngd_score &score, int pairwise_del, int fast,
                         const ngd_tile *d_tiles16, uint32_t n_tiles16, uint32_t n_ks,
                         uint64_t sites_per_slice, double *slab);

// accum_em_table.hip : per-site EM with per-individual tables shared by a 64 x 64 tile of pairs
void ngd_launch_accum_em_table(hipStream_t st, const ngd_geom &g, const double *PA, const uint32_t *d_ws,
                               uint64_t n_sites_eff, const ngd_score &score, int pairwise_del, int shape,
                               const ngd_tile *d_tiles64, uint32_t n_tiles64, uint32_t n_ks, uint64_t sites_per_slice,
                               double *slab, unsigned long long *d_counters /* [4]: += (tile, site) visits, table rounds; [2..3] = clock counters */);

// accum_em_table.hip, rb (4 or 8) matrices in one pass: d_Wb is [n_sites][rb] doubles, slab [n_ks][rb][n_pad][n_pad]
void ngd_launch_accum_em_table_batch(hipStream_t st, const ngd_geom &g, const double *PA, const double *d_Wb, int rb,
                                     uint64_t n_sites_eff, const ngd_score &score, int pairwise_del,
                                     const ngd_tile *d_tiles64, uint32_t n_tiles64, uint32_t n_ks,
                                     uint64_t sites_per_slice, double *slab, unsigned long long *d_counters);

// accum_em_table.hip, terms not summed over the slice: sites [s_lo, s_hi) in units of q consecutive sites (one term per
// pair slot and unit) into C (fragment-major: k-groups of 4 units x n_pg groups of 16 pair slots; d_rowpg[tile * 64 + row] + g =
// slot group of the row's group g of 16 columns, for the groups that hold a pair); *d_nanflag = 1 if a term was not finite
void ngd_launch_accum_em_table_spill(hipStream_t st, const ngd_geom &g, const double *PA, uint64_t s_lo, uint64_t s_hi,
                                     const ngd_score &score, int pairwise_del, const ngd_tile *d_tiles64,
                                     uint32_t n_tiles64, uint32_t n_ks, uint64_t sites_per_slice, uint32_t q,
                                     const uint32_t *d_rowpg, uint32_t n_pg, double *C, unsigned long long *d_counters,
                                     unsigned long long *d_nanflag);

// contract_mfma.hip : running sums D[matrix][pair slot] += W[matrix][unit] * C[unit][pair slot] over a chunk of sites
// (a unit = q consecutive sites of one bootstrap block; site s_lo is the first site of the chunk's unit 0)
uint32_t ngd_contract_rep_groups(uint32_t n_mat);  // groups of 16 matrices
void ngd_launch_spill_weights(hipStream_t st, const uint32_t *d_mult, uint32_t n_mat, int lead, uint64_t s_lo,
                              uint64_t s_hi, uint32_t q, uint64_t n_sites, uint64_t n_eff, uint64_t n_blocks,
                              uint64_t block_size, double *d_Wt);
void ngd_launch_spill_sanitize(hipStream_t st, double *C, const unsigned long long *d_flag, uint64_t n_kg, uint32_t n_pg,
                               const uint32_t *d_mult, uint32_t n_mat, int lead, uint64_t s_lo, uint32_t q, uint64_t n_sites,
                               uint64_t n_eff, uint64_t n_blocks, uint64_t block_size, double *D);
// n_pg must be a multiple of 4 (ngd_create pads the slot groups)
void ngd_launch_contract(hipStream_t st, const double *d_Wt, const double *C, uint32_t n_mat, uint32_t n_pg, uint32_t n_kg,
                         double *D);
void ngd_launch_spill_scatter(hipStream_t st, const double *D, uint32_t n_pg, const ngd_tile *d_tiles64, uint32_t n_tiles64,
                              const uint32_t *d_rowpg, uint64_t n_ind, uint32_t n_mat, double *d_sum);

// rb (4, 8 or 16) replicates in one pass; d_Wb is [n_sites][rb] doubles, slab [n_ks][rb][n_pad][n_pad]
void ngd_launch_accum_em_table_slices(hipStream_t st, const ngd_geom &g, const double *PA, const ngd_score &score, int pairwise_del,
                                      int shape, const ngd_tile *d_tiles64, uint32_t n_tiles64, uint32_t ks0, uint32_t n_sub,
                                      uint64_t sites_per_slice, double *slab, unsigned long long *d_counters, uint32_t lds_pad);
void ngd_launch_accum_em_batch(hipStream_t st, const ngd_geom &g, const double *PA, const double *d_Wb, int rb,
                               uint64_t n_sites_eff, const ngd_score &score, int pairwise_del, int fast,
                               const ngd_tile *d_tiles16, uint32_t n_tiles16, uint32_t n_ks,
                               uint64_t sites_per_slice, double *slab);

// fixup.hip : single_image = 2 engines, the pairs whose sums the congruent arithmetic cannot hold to 1e-9 relative
// (mean per-site term below NGD_FIX_MEAN: nearly identical individuals) recomputed with two-operand arithmetic from
// p recovered out of the image T and the side array SM[site][individual] = min(p0, p2).
#define NGD_FIX_MEAN 1e-6  // flag a pair whose sum is below this x the sites its matrix visits (error bound: 4e-17 per site)
#define NGD_FIX_CAP 4096u  // pairs (or tiles) recomputed per LAUNCH of the fix-up kernels (the size of their scratch)
// The reductions note up to NGD_FIX_LIST pairs (ngd_engine::fix_cap: the capacity of the list) and the pass recomputes
// every one of them (more than the list holds: every pair of the engine, tile by tile).  Its cost in pair-sites of work --
// a pair recomputed alone counts its sites once ([measured] 1.25e10 pair-sites/s: ~400 bytes of 64-byte sectors per
// pair-site), a 16 x 16 tile of pairs recomputed whole counts them NGD_FIX_TILE_COST times (2.9e9 tile-sites/s) however
// many of its 256 pairs are noted -- only matters to a caller that sets a budget (NGD_OPT_FIXUP_WORK; rounds 4-5 had a
// built-in one of 4.1e9, ~0.33 s, beyond which NO noted pair was recomputed: removed in round 6).  A small data set may
// have every pair recomputed: identical called genotypes over a handful of sites are thousands of sums of exactly 0, all
// noted (0 may be a cancelled 1e-20).
#define NGD_FIX_LIST (1u << 20)
#define NGD_FIX_TILE_COST_X10 43u  // (4.3)
struct ngd_fix_flags {     // what the reduction kernels need to note the pairs that want the fix-up
  unsigned long long *list;  // [cap] (i << 32) | j
  uint32_t *count;           // pairs noted (may exceed the capacity: then the fix-up is skipped)
  uint32_t *seen;            // [n_pairs / 32 + 1] one bit per pair, for reductions that visit a pair once per replicate chunk
  uint32_t cap;              // entries the list holds; pairs noted beyond it are counted, not listed
};
// a 16 x 16 tile of pairs (row group ig, column group jg of 16 individuals) that holds noted pairs: bit r * 16 + c of mask
struct ngd_fix_tile {
  uint16_t ig, jg;
  uint32_t n;  // noted pairs in the tile
  unsigned long long mask[4];
};
#define NGD_FIX_TILE_MIN 5u  // a tile with at least this many noted pairs is recomputed whole (k_fixup_tile: it costs what 4.3 single pairs do), the rest pair by pair
// out_mode 0: the partial sums of tile q's 256 pairs over slice sl go to out[(q * n_slices + sl) * 256 ..]; 1: the noted pairs'
// slab entries.  ngd_launch_fixup_tiles_finish: d_sum[pair] = the noted pairs' slices added in ascending order
void ngd_launch_fixup_tiles(hipStream_t st, const ngd_geom &g, const ngd_score &score, const double *T, const double *SM,
                            const uint32_t *d_ws, const ngd_fix_tile *d_tiles, uint32_t n_tiles, uint64_t s_lo, uint64_t s_hi,
                            uint64_t sites_per_slice, uint32_t n_slices, int out_mode, double *out);
void ngd_launch_fixup_tiles_finish(hipStream_t st, const ngd_geom &g, const ngd_fix_tile *d_tiles, uint32_t n_tiles,
                                   const double *parts, uint32_t n_slices, double *d_sum);
// out_mode 0: the partial sum of pair slot q over slice sl goes to out[q * n_slices + sl]; 1: to the slab entry
// out[(sl * n_pad + i) * n_pad + j] (per-block partial results).  Slice sl = sites [s_lo + sl * sites_per_slice, ...) below s_hi.
void ngd_launch_fixup(hipStream_t st, const ngd_geom &g, const ngd_score &score, const double *T, const double *SM,
                      const uint32_t *d_ws, const unsigned long long *d_list, uint32_t n_list, uint64_t s_lo, uint64_t s_hi,
                      uint64_t sites_per_slice, uint32_t n_slices, int out_mode, double *out);
// out_mode 0's second step: d_sum[pair] = the pair's slices added in ascending order
void ngd_launch_fixup_finish(hipStream_t st, const ngd_geom &g, const unsigned long long *d_list, uint32_t n_list,
                             const double *parts, uint32_t n_slices, double *d_sum);

// reduce.hip : deterministic slab reduction + valid-site counting
// planes_per_slice: the slab holds that many result planes per slice (EM batch kernel), `slab` points at
// the first slice's plane of the wanted result
// d_cnt != NULL: every pair's count is set to cnt_value in the same launch (no --pairwise_del)
// fix != NULL (single_image = 2 engines): pairs whose sum is below fix_thr are noted for the fix-up pass
void ngd_launch_reduce(hipStream_t st, const ngd_geom &g, const double *slab, uint32_t n_ks,
                       uint32_t planes_per_slice, const ngd_tile *d_tiles, uint32_t n_tiles, double *d_sum,
                       unsigned long long *d_cnt = nullptr, unsigned long long cnt_value = 0,
                       const ngd_fix_flags *fix = nullptr, double fix_thr = 0);
// --pairwise_del: the pairs of d_sum / d_cnt ([n_rep][n_pairs]) that want the fix-up pass, decided with their valid-site
// counts in hand (a pair with no valid site in a matrix is exactly 0 there and is not noted); fix.count zeroed by the caller
// layout.hip / reduce.hip : the fix-up pass as ONE two-operand pass over scratch images formed a range of k-groups at a time
void ngd_launch_pq_range(hipStream_t st, const ngd_geom &g, const ngd_score &score, const double *T, const double *SM,
                         const uint32_t *d_ws, uint64_t kg_lo, uint64_t kg_end, double *Ps, double *Qs);
void ngd_launch_fix_merge(hipStream_t st, const ngd_geom &g, const double *d_new, double *d_sum, const unsigned long long *d_cnt,
                          double thr, const ngd_tile *d_tiles, uint32_t n_tiles);
void ngd_launch_fix_flag(hipStream_t st, const ngd_geom &g, const double *d_sum, const unsigned long long *d_cnt,
                         uint32_t n_rep, const ngd_tile *d_tiles, uint32_t n_tiles, const ngd_fix_flags &fix);
// host_util.cpp: ngd_finish_stream over n_mat matrices, counts per cell (cnt) or one per matrix (cnt_mat)
int ngd_finish_matrices_stream(const double *sum, const uint64_t *cnt, const uint64_t *cnt_mat, uint32_t n_mat, uint64_t n_pairs,
                               uint64_t evol_model, double *dist, const volatile uint64_t *landed);
uint32_t ngd_reduce_chunk(uint32_t n_rep);  // replicates per pass; weight strides are multiples of it
// fix != NULL: a pair is noted if its sum in ANY replicate r is below d_thr[r]
void ngd_launch_reduce_w(hipStream_t st, const ngd_geom &g, const double *slab, uint32_t n_ks, const double *d_W,
                         uint32_t w_stride, uint32_t n_rep, const ngd_tile *d_tiles, uint32_t n_tiles,
                         double *d_sum, const ngd_fix_flags *fix = nullptr, const double *d_thr = nullptr, uint32_t chunk = 0);
void ngd_launch_reduce_c(hipStream_t st, const ngd_geom &g, const uint32_t *C, uint32_t n_blocks, const uint32_t *d_M,
                         uint32_t m_stride, uint32_t n_rep, const ngd_tile *d_tiles, uint32_t n_tiles,
                         unsigned long long *d_cnt, uint32_t chunk = 0);
void ngd_launch_count_blocks(hipStream_t st, const ngd_geom &g, const unsigned long long *mask, uint64_t block_size,
                             uint32_t n_blocks, const ngd_tile *d_tiles16, uint32_t n_tiles16, uint32_t *C);
void ngd_launch_count(hipStream_t st, const ngd_geom &g, const unsigned long long *mask,
                      const unsigned long long *planes, uint32_t n_planes, const ngd_tile *d_tiles,
                      uint32_t n_tiles, unsigned long long *d_cnt);  // owned 128-tiles
void ngd_launch_fill_cnt(hipStream_t st, const ngd_geom &g, const ngd_tile *d_tiles, uint32_t n_tiles,
                         unsigned long long value, const unsigned long long *d_values, uint32_t n_rep,
                         unsigned long long *d_cnt);
