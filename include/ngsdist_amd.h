/*
 * ngsdist_amd.h -- C ABI of the MI355X pairwise-distance engine.
 *
 * This is the drop-in boundary for the ONE hot path of fgvieira/ngsDist: the
 * per-replicate fan-out of gen_dist() over all pairs of individuals.  The
 * reference has no FFI layer; the seam is the task `void gen_dist_slave(void*)`
 * handed to threadpool_add() (reference ngsDist.cpp:251, task type
 * threadpool.h:71-74) around `double gen_dist(params*, uint64_t, uint64_t)`
 * (ngsDist.hpp:55-56).  Each entry point below names the reference lines it
 * stands in for.  INTEGRATION.md shows the patch a maintainer of the
 * reference would apply to call it.
 *
 * Conventions
 *  - plain C, no C++/torch/HIP types in any signature; device pointers and
 *    streams travel as void*;
 *  - every function returns NGD_OK (0) or a negative NGD_E_* code and leaves a
 *    message retrievable with ngd_last_error(); nothing calls exit();
 *  - the caller owns every host buffer; the engine owns every device buffer
 *    and keeps no host pointer after a call returns;
 *  - calls on one engine are not thread-safe (one caller, like main()).
 *  - pair order is the reference's: row-major upper triangle, i1 < i2
 *    (ngsDist.cpp:244-245); n_pairs = n_ind*(n_ind-1)/2.
 *  - results are deterministic run to run (no floating-point atomics): the same calls in the same order on a new engine
 *    give the same bits.  Two calls of the same bootstrap job on ONE engine may take different plans (NGD_OPT_BOOT_PARTIALS
 *    = 1 serves the first calls of a geometry without the per-block partial results while their slab would cost more to
 *    allocate than it has saved) and then agree to rounding, not bit for bit; 0 or 2 pin the plan.
 *  - there is no CPU fallback: without a HIP device ngd_create() fails.
 */
#ifndef NGSDIST_AMD_H
#define NGSDIST_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGD_ABI_VERSION 6 /* 6: ngd_run_job_dist, ngd_run_batch_dist, ngd_run_mult_batch_dist; 5: ngd_finish_stream, ngd_fixup_info.by_pass, NGD_OPT_STAGE_PIECE_MIB / _STAGE_RING / _EAGER_FULL, NGD_OPT_FIXUP_WORK 0 = no budget (every noted pair is recomputed); 4: ngd_last_spill_timing, ngd_last_fixup, ngd_image_mode, ngd_config.single_image 0 = auto / 3 = two images; 3: ngd_config.single_image / second_image_mib (were reserved, zero), ngd_fetch_matrix, ngd_score_congruence */

#define NGD_OK 0
#define NGD_E_INVALID (-1)  /* bad argument / bad state                    */
#define NGD_E_NODEVICE (-2) /* no usable HIP device                        */
#define NGD_E_HIP (-3)      /* a HIP runtime call failed                   */
#define NGD_E_NOMEM (-4)    /* host or device allocation failed            */
#define NGD_E_MODEL (-5)    /* evolutionary model 3..6 (reference: error())*/
#define NGD_E_NAN (-6)      /* "NaN found! Is the file format correct?" (read_data.cpp:42-45) */

/* which kernel serves the per-pair accumulation */
#define NGD_KERNEL_AUTO 0   /* indep: MFMA (the streaming kernel where the MFMA kernel's 8 slab planes of n_pad^2 doubles  */
                            /* cannot fit the device: tens of thousands of individuals); EM: the table kernel above 32     */
                            /* individuals, the fast per-pair kernel up to 32                                              */
#define NGD_KERNEL_STREAM 1 /* indep: one wavefront per pair, streams 48 B per pair-site */
#define NGD_KERNEL_MFMA 2   /* indep: FP64 MFMA tiles over the (pair, site) contraction */
#define NGD_KERNEL_EM_FAITHFUL 3 /* EM: iterates bit-identical to emOptim2.cpp */
#define NGD_KERNEL_EM_FAST 4     /* EM: division-free power iteration, same stopping rule */
#define NGD_KERNEL_EM_TABLE 5    /* EM: the fast form with per-individual tables shared by a 64x64 tile of pairs */

typedef struct ngd_engine ngd_engine;

/* The subset of the reference's `params` (ngsDist.hpp:11-44) that gen_dist()
 * reads, plus placement. */
typedef struct ngd_config {
  uint64_t n_ind;       /* params.n_ind; bounded by device memory (NGD_E_NOMEM: two n_pairs-long result arrays +  */
                        /* n_pad^2 doubles per slice), never above 1 048 448 (16-bit tile indices)          */
  uint64_t n_sites;     /* params.n_sites of the loaded data set              */
  double score[9];      /* params.score[g1][g2] row-major, parse_args.cpp:25-27,134-137 */
  int32_t pairwise_del; /* params.pairwise_del, ngsDist.cpp:335-338           */
  int32_t indep_geno;   /* params.indep_geno, ngsDist.cpp:348-353             */
  int32_t device;       /* HIP device ordinal; -1 = current device            */
  int32_t kernel;       /* NGD_KERNEL_*                                       */
  uint32_t shard_rank;  /* this engine computes the pair tiles that            */
  uint32_t shard_world; /*   ngd_shard_of_pair() gives it; 0/1 = everything   */
  /* Launch geometry; 0 = the measured defaults, which is what a host wants.  They change
   * speed only, never results beyond the order of additions over site slices. */
  uint32_t variant;      /* NGD_KERNEL_EM_TABLE: workgroup shape 0..4 (accum_em_table.hip)            */
  uint32_t n_slices;     /* slices of the site axis per launch (MFMA: rounded up to a multiple of 8);  */
                         /* held to what the data set allows (>= 128 k-groups / >= 1 site per slice)  */
  uint32_t wg_target;    /* workgroups wanted per launch, from which n_slices is derived when it is 0 */
  uint32_t exact_shapes; /* NGD_KERNEL_MFMA, block form: issue only the MFMA tiles a block of pairs needs (narrow edge    */
                         /* blocks, triangular diagonal blocks).  0 = auto: off above 384 padded individuals, else up to  */
                         /* 208 individuals form 6, up to 256 form 4, else form 2.  1 = never (full 4 x 4 pattern);       */
                         /* 7 = the full pattern, four blocks to a workgroup, except that the blocks ON the diagonal leave */
                         /* out the six tiles below it and sit in workgroups of their own (a one-image engine fetches     */
                         /* their four operand fragments once for rows and columns).  Fewer MFMAs, no faster: such blocks */
                         /* run ahead of their slice and its operands then leave HBM more than once; */
                         /* 2 = blocks of up to 4 x 4 tiles of 16 x 16 pairs, one single-wavefront workgroup per block;   */
                         /* 3 = the same with blocks of up to 2 x 4; 4 / 5 = 2 / 3 with a slice's blocks in ONE workgroup */
                         /* that moves through the sites in step (<= 12 / <= 16 blocks; a prefetching wavefront where one */
                         /* fits); 6 = 5 with the operands staged through LDS.  Fallbacks, all silent and result-         */
                         /* neutral: 4 -> 2 and 5 / 6 -> 3 where a slice's blocks do not fit one workgroup; 6 -> 5 for    */
                         /* any pass that carries per-index weights -- bootstrap passes, masked partial-sum slices and    */
                         /* EVERY pass of a single_image = 2 engine (its plain pass is weighted by the congruence)        */
  uint32_t single_image; /* NGD_KERNEL_MFMA: how many operand images the engine holds.  Two (p and q = score . p,           */
                         /* ngsDist.cpp:351-353 regrouped) are 48 bytes per (padded individual, site); ONE is 24, i.e.    */
                         /* up to twice the sites per engine and half the HBM reads per pass:                              */
                         /* 2 = ONE image in coordinates in which the (symmetric) score matrix is diagonal, score = SUM_r  */
                         /*     d_r c_r c_r^T: the image holds t_r = c_r . p, both operands are read from it, d rides on   */
                         /*     the per-index weights.  The speed of two images.  The squares differ in sign: sums are     */
                         /*     exact for called genotypes (c, d dyadic for the reference's matrices) and otherwise carry  */
                         /*     an ABSOLUTE error of <= 4e-17 per site -- 1e-9 relative wherever a pair's mean per-site    */
                         /*     term is above 4e-8.  For the reference's two matrices (parse_args.cpp:25-27, :134-137) the */
                         /*     pairs below that -- nearly identical individuals -- are then RECOMPUTED with the two-operand */
                         /*     arithmetic (the fix-up pass: every pair whose sum is below 1e-6 x the sites visited -- under */
                         /*     --pairwise_del x the pair's own valid sites, and never a pair without one; it needs         */
                         /*     min(p0, p2) beside the image, 8 more bytes per individual and site: 32 in all), so 1e-9    */
                         /*     relative holds at any distance, unconditionally: every qualifying pair is recomputed,      */
                         /*     however many there are (clusters of copies tile by tile, 60 times cheaper than pair by     */
                         /*     pair; where the tiles would cost more than the whole matrix in the two-image arithmetic --  */
                         /*     a data set of clones -- one more pass over scratch images formed a range of sites at a time */
                         /*     does it: ~70 ms at 1000 x 1e6 whatever the data; replicates from per-block partial results  */
                         /*     go tile by tile at any count, up to ~0.8 s at that size).  ngd_last_fixup() reports what a  */
                         /*     run recomputed.  Only a caller that SETS a                                                  */
                         /*     budget (NGD_OPT_FIXUP_WORK) can have pairs left at the absolute bound.                     */
                         /*     Any other symmetric matrix: no fix-up.  NGD_E_INVALID for an asymmetric matrix.            */
                         /* 1 = p resident, q formed for a range of sites at a time before the launch that reads it: the   */
                         /*     arithmetic of two images (sums equal to rounding, per-block partial sums bit for bit), any  */
                         /*     score matrix, a fifth more time (55.6 ms at 1000 x 1e6 instead of 45.6)                    */
                         /* 3 = two images, always                                                                         */
                         /* 0 = auto: 2 where its fix-up pass exists (the reference's matrices) and the kernel runs its     */
                         /*     full 4 x 4 block form (exact_shapes resolves to 0: more than 384 padded individuals, or    */
                         /*     exact_shapes = 1 / 7 asked for -- the block forms of a few hundred individuals take no     */
                         /*     per-index weights in their fastest variant), else two images.  ngd_image_mode() tells what */
                         /*     an engine holds.                                                                           */
  uint32_t second_image_mib; /* single_image = 1: MiB of q kept resident all the same, from the first site on      */
                         /* (what the device has to spare): only the rest is formed range by range, and the extra */
                         /* time shrinks in proportion                                                             */
} ngd_config;

/* Per-run device timings (HIP events on the engine's stream). */
typedef struct ngd_timing {
  double ms_total;      /* first launch of the run -> results ready           */
  double ms_accum;      /* the dominant accumulation kernel alone             */
  double ms_reduce;     /* split-site slab reduction                          */
  double ms_count;      /* valid-site counting (pairwise deletion only)       */
  uint64_t pair_sites;  /* (pairs this engine owns) x (sites visited)         */
  uint64_t launches;    /* accumulation kernel launches in the run            */
} ngd_timing;

const char *ngd_last_error(void);
int ngd_abi_version(void);
/* number of visible HIP devices (0 if none); never fails */
int ngd_device_count(void);

/* Engine lifetime.  Stands in for the thread pool + pth_struct array set up in
 * ngsDist.cpp:197-208 and torn down in :291-306. */
int ngd_create(const ngd_config *cfg, ngd_engine **out);
void ngd_destroy(ngd_engine *e);

/* Input: the prepared, NORMAL-space array gen_dist() reads (params.geno_lkl
 * after ngsDist.cpp:165-174).  Two host layouts are accepted:
 *   ngd_upload_sites     p[(s*n_ind + i)*3 + g], sites s0 .. s0+n-1 -- the
 *                        order of the binary file (read_data.cpp:28-31), so a
 *                        host can stream a file through in chunks;
 *   ngd_upload_ind_major p[(i*n_sites + s)*3 + g] -- the reference's in-memory
 *                        order in_geno_lkl[i][s][g], whole data set at once.
 * ngd_commit() ends the upload (derives the per-individual missing-site masks
 * of gen_func.cpp:862-868 and the score-weighted operand).  After commit the
 * data set is immutable; bootstrap never moves data (see ngd_run). */
int ngd_upload_sites(ngd_engine *e, const double *p, uint64_t s0, uint64_t n);
int ngd_upload_ind_major(ngd_engine *e, const double *p);
int ngd_commit(ngd_engine *e);

/* Raw input: the doubles of a BINARY GL file exactly as stored (read_data.cpp:28-31,
 * [site][individual][3]); the engine performs the reference's kernel-input construction
 * on the device -- log unless in_logscale, normalisation (post_prob), the NaN check of
 * read_data.cpp:37-45, then call_geno and exp of ngsDist.cpp:165-174 -- instead of the
 * host.  Device log/exp agree with glibc's to the last ulp or so: use the host path
 * (ngd_upload_sites) where called genotypes must be decided bit-for-bit as on the CPU.
 *
 * Zero-copy pipeline: ngd_stage_acquire() lends one of the engine's pinned host buffers
 * (*capacity_sites sites of n_ind*3 doubles; a ring of NGD_OPT_STAGE_RING buffers of
 * NGD_OPT_STAGE_PIECE_MIB MiB); the caller reads the file straight into it and calls
 * ngd_stage_submit(), which returns at once: the copy runs on a copy stream of its own, the
 * preparation kernel behind it on the engine's stream, and the next acquire hands out the
 * next buffer of the ring (waiting only for the copy out of THAT buffer, a turn ago).  ngd_upload_raw_sites() is the blocking
 * convenience form.  NGD_E_NAN is reported by ngd_commit() at the latest. */
typedef struct ngd_prep {
  int32_t in_logscale; /* --log_scale                           */
  int32_t call_geno;   /* --call_geno (or -N / -C)              */
  double N_thresh;     /* --N_thresh                            */
  double call_thresh;  /* --call_thresh                         */
} ngd_prep;
int ngd_stage_acquire(ngd_engine *e, double **host_buf, uint64_t *capacity_sites);
int ngd_stage_submit(ngd_engine *e, uint64_t s0, uint64_t n, const ngd_prep *prep);
int ngd_upload_raw_sites(ngd_engine *e, const double *raw, uint64_t s0, uint64_t n, const ngd_prep *prep);

/* Test/bench input: fills the engine with this repository's counter-based
 * synthetic data set (bit-identical to oracle ngo_synth_one) on
 * the device, then commits.  Not part of the reference's surface. */
int ngd_synth_fill(ngd_engine *e, uint64_t seed, double miss_frac);
/* Same generator, but the engine's site 0 is site `site0` of the synthetic data set: an engine
 * that holds one contiguous range of sites (site sharding, below). */
int ngd_synth_fill_range(ngd_engine *e, uint64_t seed, double miss_frac, uint64_t site0);

/* One replicate = everything between rnd_map_data() and the matrix print:
 * the `for i1<i2: threadpool_add(gen_dist_slave)` + threadpool_wait block,
 * ngsDist.cpp:244-269, with gen_dist()'s loop :333-364.
 *   block_map == NULL : the full data set (rep 0), n_blocks/block_size ignored;
 *   else                block_map[b] (b < n_blocks) is the source block that
 *                       rnd_map_data (ngsDist.cpp:416-437) puts at block b,
 *                       i.e. floor(draw_rnd(0, n_blocks)); sites at or beyond
 *                       n_blocks*block_size are not visited (:236).
 * Outputs (host, caller-owned, n_pairs entries each, either may be NULL):
 *   sum[k] : gen_dist()'s `dist` before ngsDist.cpp:376;
 *   cnt[k] : gen_dist()'s `cnt` before the tot_sites override (:372-373).
 * Pairs outside this engine's shard are returned as 0 / 0. */
int ngd_run(ngd_engine *e, const uint64_t *block_map, uint64_t n_blocks,
            uint64_t block_size, double *sum, uint64_t *cnt);

/* Same, but results stay on the device: d_sum (double[n_pairs]) and d_cnt
 * (uint64_t[n_pairs]) are DEVICE pointers owned by the caller (e.g. the
 * storage of a torch tensor used for the RCCL gather).  Returns after the
 * engine's stream has finished. */
int ngd_run_device(ngd_engine *e, const uint64_t *block_map, uint64_t n_blocks,
                   uint64_t block_size, void *d_sum, void *d_cnt);

/* Site sharding.  gen_dist()'s sum and cnt are sums over sites, so a data set can also be split
 * along the SITE axis: every engine (GPU) holds a contiguous range of sites of all individuals,
 * computes all pairs over its range, and the per-engine (sum, cnt) are ADDED (one RCCL
 * reduce; exact for called genotypes, <= 1e-12 relative otherwise -- the addition order differs
 * from one engine's).  For a bootstrap replicate an engine needs the multiplicity of each of ITS
 * blocks, which is not a block map of its own: ngd_run_mult() takes the multiplicities directly
 * (mult[b] = number of draws of the engine's block b; ranges must be whole blocks).  cnt is then
 * block_size * SUM mult[b] (or the multiplicity-weighted valid-site count with --pairwise_del). */
int ngd_run_mult(ngd_engine *e, const uint32_t *mult, uint64_t n_blocks, uint64_t block_size,
                 double *sum, uint64_t *cnt);
int ngd_run_mult_device(ngd_engine *e, const uint32_t *mult, uint64_t n_blocks,
                        uint64_t block_size, void *d_sum, void *d_cnt);

/* n_rep bootstrap replicates in ONE call: n_rep turns of the replicate loop ngsDist.cpp:217-289 (each
 * = rnd_map_data :416-437 + the fan-out/wait :244-269).  block_maps is [n_rep][n_blocks], replicate r's
 * map being what the r-th rnd_map_data() call would draw (the taus stream is consumed by nothing else,
 * so a host may draw all maps up front: ngd_boot_block_map n_rep times); mult is [n_rep][n_blocks]
 * (site sharding, as ngd_run_mult).  Outputs are [n_rep][n_pairs].  The engine computes per-block partial
 * (sum, cnt) once and forms up to 32 replicates per pass over them, so a batch costs little more than
 * one replicate; a replicate's result is the same whether it came from a batch or from ngd_run().
 * When the partials do not apply (streaming kernel, not enough device memory for one partial result per
 * block -- e.g. block size 1 on a large data set) this is n_rep weighted accumulation passes on the
 * --indep_geno path (each walks only the sites its replicate drew).  On the EM path the term of a (pair, site)
 * does not depend on the replicate, so the matrices share the per-site EM: the table-driven kernel (from three
 * matrices on, NGD_OPT_EM_SPILL) writes the terms of a chunk of sites once and ONE FP64 MFMA contraction with
 * every matrix's weights adds the chunk to all of them -- one EM pass whatever the replicate count; matrices then
 * agree with their own ngd_run() pass to rounding (<= 1e-12 relative), counts exactly.  Otherwise (option off, two
 * matrices, the per-pair kernels) one pass serves 8 matrices (table-driven kernel) or 16 (per-pair kernels), each
 * accumulator taking the term with the site's weight in its matrix: same bits as from ngd_run() in the table-driven
 * kernel's default variant and in the per-pair kernels; its other variants (1..4) borrow the per-pair batch kernel
 * and agree with their own ngd_run() to rounding only. */
int ngd_run_batch(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks,
                  uint64_t block_size, double *sum, uint64_t *cnt);
int ngd_run_batch_device(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks,
                         uint64_t block_size, void *d_sum, void *d_cnt);
int ngd_run_mult_batch(ngd_engine *e, const uint32_t *mult, uint32_t n_rep, uint64_t n_blocks,
                       uint64_t block_size, double *sum, uint64_t *cnt);
int ngd_run_mult_batch_device(ngd_engine *e, const uint32_t *mult, uint32_t n_rep, uint64_t n_blocks,
                              uint64_t block_size, void *d_sum, void *d_cnt);

/* The WHOLE replicate loop ngsDist.cpp:217-289 in one call: matrix 0 is the full data set (what
 * ngd_run(e, NULL, ...) returns), matrices 1..n_rep the bootstrap replicates of block_maps
 * ([n_rep][n_blocks], drawn as for ngd_run_batch).  Outputs are [n_rep + 1][n_pairs].  Knowing the whole
 * job lets the engine share work between the matrices: when the blocks cover every site the full-data
 * matrix is the all-ones combination of the same per-block partials as the replicates; on the EM path
 * (no --indep_geno) with blocks too small for partials -- e.g. the default --boot_block_size 1 -- up to 16
 * matrices, the full-data one included, share ONE pass of the per-site EM, which does not depend on the
 * replicate.  Replicates carry the same bits as from ngd_run() (to rounding where ngd_run_batch says so: the
 * spilled-terms plan of the table-driven EM kernel); matrix 0 agrees with ngd_run(e, NULL) to rounding (exactly for
 * called genotypes) because its sum may be formed in a different order.
 * n_rep = 0 is ngd_run(e, NULL, ...). */
int ngd_run_job(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks,
                uint64_t block_size, double *sum, uint64_t *cnt);
int ngd_run_job_device(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks,
                       uint64_t block_size, void *d_sum, void *d_cnt);
/* ngd_run_batch / ngd_run_mult_batch / ngd_run_job with sum = cnt = NULL leave their matrices in the engine; this copies
 * matrix `which` of that last call (0 = the first it computed) to the caller -- a host that prints the matrices one after
 * the other (ngsDist.cpp:282-287) then needs two n_pairs-long buffers, not (n_boot_rep + 1) of them.  Valid until the
 * engine's next run call. */
int ngd_fetch_matrix(ngd_engine *e, uint32_t which, double *sum, uint64_t *cnt);
/* A job AND the tail of gen_dist() (ngsDist.cpp:217-289 with :372-401) in one call: `dist` ([n_rep + 1][n_pairs] for
 * ngd_run_job_dist -- [1][n_pairs] when n_rep = 0 --, [n_rep][n_pairs] for ngd_run_batch_dist / ngd_run_mult_batch_dist; any host memory)
 * receives what ngd_run_job() / ngd_run_batch() / ngd_run_mult_batch() followed by ngd_finish() on every matrix would give, bit
 * for bit.
 * The sums (and, --pairwise_del, the counts) leave the device in chunks of about 8 MiB on a stream of the engine's own into
 * pinned memory the engine keeps, and up to 64 host threads turn each chunk into distances as it lands; in the
 * per-block-partials plan a group of 32 replicates is reduced by a launch of its own and its copy follows at once, beside
 * the reductions of the later groups.  The matrices stay in the engine as after sum = cnt = NULL (ngd_fetch_matrix).
 * tot_sites and evol_model as for ngd_finish() (tot_sites > 0: the count of every cell, --tot_sites, ngsDist.cpp:372-373;
 * NGD_E_INVALID with --pairwise_del, parse_args.cpp:209-210).  NGD_E_MODEL for evol_model > 2 before anything is launched;
 * NGD_E_INVALID on an engine that owns a share of the pairs (ngd_config.shard_world > 1: its matrices are partial). */
int ngd_run_job_dist(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks, uint64_t block_size,
                     uint64_t tot_sites, uint64_t evol_model, double *dist);
int ngd_run_batch_dist(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks, uint64_t block_size,
                       uint64_t tot_sites, uint64_t evol_model, double *dist); /* replicates only, as ngd_run_batch */
int ngd_run_mult_batch_dist(ngd_engine *e, const uint32_t *mult, uint32_t n_rep, uint64_t n_blocks, uint64_t block_size,
                            uint64_t tot_sites, uint64_t evol_model, double *dist);

/* Bootstrap replicates re-use per-block partial (sum, cnt) computed on the first
 * run that carries a block map (valid for that block size / block count;
 * recomputed automatically when either changes).  This forgets them, so that a
 * benchmark can charge the partial-sum pass to every timed step. */
int ngd_drop_caches(ngd_engine *e);

/* Plan selection for the replicate loop, per engine and changeable between runs (defaults in brackets).  The engine
 * picks the cheapest plan that applies by itself; these exist to pin one for a comparison or to bound memory.
 * Every plan returns the same counts and sums equal to rounding ("Plans"). */
#define NGD_OPT_BOOT_PARTIALS 1  /* [1] bootstrap replicates from per-block partial (sum, cnt): 0 never, 1 when they */
                                 /*     fit and pay for their allocation, 2 allocate even a large slab at once       */
#define NGD_OPT_BOOT_MAX_BYTES 2 /* [0 = 85 % of free device memory] budget of those partials (and, when set, of the     */
                                 /*     EM batch pass's result planes: beyond it, one pass per matrix)                */
#define NGD_OPT_BOOT_WG 3        /* [4096] workgroups wanted in the pass that fills them                             */
#define NGD_OPT_BOOT_UNALIGNED 4 /* [1] MFMA path: partials also for block sizes that are not multiples of 4 sites   */
#define NGD_OPT_EM_BATCH 5       /* [1] EM kernels without partials: up to 16 (per-pair) / 8 (table-driven) matrices  */
                                 /*     per accumulation pass                                                         */
#define NGD_OPT_EM_SPILL 6       /* [1] table-driven EM kernel without partials, 3 matrices or more: ONE pass writes   */
                                 /*     the per-(pair, site) terms of a chunk of sites, one FP64 MFMA contraction with  */
                                 /*     the matrices' weights adds the chunk to every matrix (any replicate count);     */
                                 /*     0 never, 1 from 3 matrices on, 2 from 2 on.  Matrices then agree with their own */
                                 /*     ngd_run() pass to rounding (<= 1e-12 relative), not bit for bit                 */
#define NGD_OPT_EM_SPILL_BYTES 7 /* [0 = 6 GB] device scratch for those terms (bounds the sites per chunk)             */
#define NGD_OPT_SINGLE_IMAGE_BYTES 8 /* [0 = 4 GB] ngd_config.single_image engines: bytes of the second operand image */
                                 /*     formed at a time (a pass is so many launches; never less than 64 k-groups per */
                                 /*     slice, or eight bootstrap blocks of a partial-sum pass); set before the first run */
#define NGD_OPT_FIXUP_WORK 9     /* [0 = no budget: every noted pair is recomputed] one-image engines, the fix-up pass of    */
                                 /* nearly identical pairs: a budget in pair-sites of recomputation (a pair alone counts its   */
                                 /* sites once, a 16 x 16 tile of pairs recomputed whole 4.3 times; ~1.2e10 a second).  A run   */
                                 /* whose noted pairs cost more keeps the one-image sums of ALL of them (absolute error <=     */
                                 /* 4e-17 per site instead of 1e-9 relative) and says so in ngd_last_fixup().skipped           */
#define NGD_OPT_STAGE_PIECE_MIB 10 /* [32] ngd_stage_acquire: MiB of raw input per pinned buffer                        */
#define NGD_OPT_STAGE_RING 11     /* [6] ... and how many of them (2 .. 8); both before the first ngd_stage_acquire    */
#define NGD_OPT_EAGER_FULL 12     /* [0] 1: a staged load (ngd_stage_*) of sites in ascending order starts the plain         */
                                 /* full-data pass beside itself: leading slices of the site axis are accumulated, on a       */
                                 /* low-priority stream, as soon as all their sites are prepared; the first ngd_run() with    */
                                 /* no block map launches only what is left.  Same bits as without it.  For a host whose     */
                                 /* first call after ngd_commit IS that pass (no bootstrap): anything else drops the work.    */
                                 /* MFMA kernel above 384 padded individuals and the table-driven EM kernel; else ignored.    */
#define NGD_OPT_DEBUG_FORGE_JOB 100 /* tests only: the first block of the MFMA kernel's job list gets the shape rows | cols << 3 |  */
                                 /*     tri << 6 -- a shape the kernel's block form does not list must fail the run with      */
                                 /*     NGD_E_HIP (its sums poisoned with NaN), never return zeros                            */
                                 /*     Refused (NGD_E_INVALID) unless the environment has NGD_ENABLE_TEST_HOOKS=1: the       */
                                 /*     engine's job list stays forged, every later run of it fails                           */
int ngd_set_option(ngd_engine *e, int option, uint64_t value);

int ngd_last_timing(const ngd_engine *e, ngd_timing *t);
/* What the engine holds (ngd_config.single_image resolved): 3 = two operand images, 1 = one image + the second formed a
 * range at a time, 2 = one image in congruent coordinates; 0 = not an MFMA engine.  *fixup (may be NULL) = 1 if the
 * engine recomputes the pairs its congruent arithmetic cannot hold to 1e-9 relative (the reference's score matrices). */
int ngd_image_mode(const ngd_engine *e, int *fixup);
/* The fix-up pass of the last run call (single_image = 2 engines on the reference's matrices; zeros otherwise): pairs
 * whose sum in a matrix of the job was below 1e-6 x the sites the matrix visits, how many were recomputed with the
 * two-operand arithmetic of ngsDist.cpp:351-353 (all of the engine's pairs when more than 2^20 were noted at once), how
 * many were left as the one-image pass computed them (never, unless the caller set a budget with NGD_OPT_FIXUP_WORK and
 * the work exceeded it: absolute error <= 4e-17 per site), and the device time of the recomputation. */
typedef struct ngd_fixup_info {
  uint64_t flagged, recomputed, skipped;
  double ms;
  uint64_t by_pass; /* recomputations that went the whole-matrix way: so many noted pairs that ONE more pass in the    */
                    /* two-image arithmetic over scratch images (a pass and a half: ~70 ms at 1000 x 1e6) was cheaper  */
                    /* than their tiles -- a data set of clones; the noted pairs take its sums, the others keep theirs */
                    /* (a bootstrap job's per-block partial results, blocks of whole k-groups: EVERY entry of the slab  */
                    /* is formed again that way, all pairs take the two-image engine's bits)                           */
} ngd_fixup_info;
int ngd_last_fixup(const ngd_engine *e, ngd_fixup_info *info);
/* The accumulation phase of the last run that took the spilled-terms plan (EM path, bootstrap blocks too small for
 * per-block partials: NGD_OPT_EM_SPILL), kernel by kernel, HIP events on the engine's stream summed over the job's chunks
 * of sites -- for roofline accounting (bench.py --workload emboot).  All zero if the last run took another plan. */
typedef struct ngd_spill_timing {
  double ms_weights;          /* k_spill_weights + the memsets of a chunk's partial last k-group                     */
  double ms_terms;            /* k_accum_em_table<SPILL>: the per-site EM (emOptim2.cpp:112-135), terms written       */
  double ms_sanitize;         /* k_spill_sanitize (a no-op unless a term was not finite)                             */
  double ms_contract;         /* k_contract_mfma: running sums += weights x terms                                    */
  uint64_t chunks;            /* chunks of sites the job went through (NGD_OPT_EM_SPILL_BYTES)                       */
  uint64_t sites;             /* sites visited                                                                       */
  uint64_t unit_sites;        /* q: consecutive sites of one bootstrap block whose terms leave the EM kernel as one   */
  uint64_t units;             /* K of the contraction = terms written and read per pair slot                         */
  uint64_t slot_groups;       /* groups of 16 pair slots (N of the contraction / 16), padding included               */
  uint64_t slot_groups_live;  /* ... of them written by the EM pass (hold at least one pair)                         */
  uint64_t matrices;          /* matrices of the job                                                                 */
  uint64_t matrix_groups;     /* groups of 16 matrices (M of the contraction / 16)                                   */
  uint64_t contract_launches; /* k_contract_mfma launches (one per chunk and 128 matrices)                           */
} ngd_spill_timing;
int ngd_last_spill_timing(const ngd_engine *e, ngd_spill_timing *t);
/* The shader clock (MHz) the last MFMA / table-driven EM accumulation launch ran at: one wavefront in the middle of
 * the grid reads the shader-cycle counter and the constant-rate counter around its work.  0 = not sampled (other
 * kernels).  For roofline accounting: a kernel's rate against the peak AT THE CLOCK THE CHIP HELD. */
int ngd_last_shader_clock(const ngd_engine *e, double *mhz);
/* Work done by the table-driven EM kernel (NGD_KERNEL_EM_TABLE) in the last run, for roofline accounting: the number
 * of (64 x 64 pair tile, site) visits and of table rounds (16 EM steps of a tile's 128 individuals each) -- the
 * data-dependent part of its operation count (the EM's iteration count, emOptim2.cpp:118-133).  Zero for other kernels. */
int ngd_last_em_work(const ngd_engine *e, uint64_t *tile_sites, uint64_t *table_rounds);

/* The tail of gen_dist(), ngsDist.cpp:372-401, on the HOST with the host's
 * libm so that -0.0 / inf / nan cells print exactly as the reference's do:
 *   cnt = tot_sites if tot_sites > 0; d = sum/cnt;
 *   model 0: d; 1: -log(1-d); 2: -log(1-d*4/3)*3/4; 3..6: NGD_E_MODEL. */
int ngd_finish(const double *sum, const uint64_t *cnt, uint64_t n_pairs,
               uint64_t tot_sites, uint64_t evol_model, double *dist);
/* The same over cells that are still arriving (a job's matrices leaving the device chunk by chunk): cells [0, *landed)
 * of sum and cnt are final; the caller -- another thread -- raises *landed, in any steps, up to n_pairs, and the call
 * returns when every cell is finished.  One wake-up of the host threads for the whole job: the tail runs beside the
 * copies (ngsDist.cpp:372-401 on up to millions of cells per bootstrap job). */
int ngd_finish_stream(const double *sum, const uint64_t *cnt, uint64_t n_pairs, uint64_t tot_sites, uint64_t evol_model,
                      double *dist, const volatile uint64_t *landed);

/* The print block of one matrix, ngsDist.cpp:282-287 with join() gen_func.cpp:479-496:
 * "\n<n_ind>\n", then per individual its label and n_ind cells "\t%.10f", newline.  `dist` is
 * ngd_finish()'s output (pair order); the matrix is its symmetric expansion with a zero diagonal
 * (dist_matrix, ngsDist.cpp:200,:411).  Cells are formatted exactly as printf's "%.10f" does
 * (-0.0000000000, inf, nan, -nan included), rows in parallel on n_threads host threads (0 = auto).
 * Returns the number of bytes of the block (no terminator); they are written to `out` only if
 * cap is large enough (call with out = NULL to size), or a negative NGD_E_* code. */
int64_t ngd_format_matrix(const double *dist, uint64_t n_ind, const char *const *labels, char *out,
                          uint64_t cap, uint32_t n_threads);

/* Bootstrap block map exactly as the reference draws it (gsl_rng_taus seeded
 * with --seed, ngsDist.cpp:179-180; one draw per block, :421-423).  The state
 * is three uint32 carried across replicates by the caller. */
void ngd_taus_seed(uint32_t state[3], uint64_t seed);
uint32_t ngd_taus_get(uint32_t state[3]);
double ngd_taus_uniform(uint32_t state[3]);
void ngd_boot_block_map(uint32_t state[3], uint64_t n_blocks, uint64_t *block_map);

/* geometry helpers */
uint64_t ngd_n_pairs(uint64_t n_ind);
uint64_t ngd_pair_index(uint64_t n_ind, uint64_t i1, uint64_t i2); /* i1 < i2 */
/* the shard (0 .. shard_world-1) that computes pair i1 < i2: 128 x 128 pair tiles of
 * the upper triangle, dealt by cost (off-diagonal tiles first) to the least loaded
 * shard -- the rule ngd_create() uses.  Pure host arithmetic. */
uint32_t ngd_shard_of_pair(uint64_t n_ind, uint64_t i1, uint64_t i2, uint32_t shard_world);
/* the same for every pair, in pair order: owner[ngd_pair_index(n_ind, i1, i2)] (n_pairs entries) */
void ngd_shard_map(uint64_t n_ind, uint32_t shard_world, int32_t *owner);
/* ngd_config.single_image = 2: the symmetric score matrix as a sum of three weighted squares,
 * score = SUM_r d[r] c_r c_r^T with c row-major (c[3 r + g]) -- Lagrange's reduction, dyadic c and d for the reference's
 * two matrices (parse_args.cpp:25-27, :134-137).  NGD_E_INVALID for an asymmetric matrix.  Pure host arithmetic. */
int ngd_score_congruence(const double score[9], double c[9], double d[3]);
/* free / total memory of a device (device < 0: the current one): lets a host decide whether a data set fits one
 * engine or has to go through it a range of sites at a time (site sharding in time instead of across GPUs) */
int ngd_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);
/* bytes of device memory the engine holds for this configuration */
uint64_t ngd_device_bytes(const ngd_engine *e);

#ifdef __cplusplus
}
#endif
#endif /* NGSDIST_AMD_H */
