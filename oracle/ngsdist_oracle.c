/*
 * ngsdist_oracle.c -- CPU restatement of the ngsDist gen_dist() hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke()
 * entry and bench.py's cpu_baseline leg may load it; the product path
 * (ngsdist_amd/) never links, imports or calls anything under oracle/.
 *
 * It restates, in plain C over flat arrays, exactly what the reference
 * (fgvieira/ngsDist v1.0.10, mounted at /root/reference) computes on the
 * path named in SURVEY.md section 8.  Every function cites the reference
 * file:line it follows.  Arithmetic order is kept identical to the reference
 * (no FMA contraction, no re-association): build with -ffp-contract=off and
 * without -ffast-math / -march=native, which is what the reference's own
 * "-O3" x86-64 build amounts to.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - em2/emStep2/lik2/normalize: pinned against oracle/_ref (the reference's
 *     own emOptim2.cpp compiled from /root/reference, no stand-ins).
 *   - gen_dist accumulation, prep, bootstrap: the full reference cannot be
 *     built here (it needs <gsl/gsl_rng.h>, absent in the image), so these are
 *     pinned only by the survey-phase reference outputs kept as data under
 *     tests/golden/survey_probe/ and by GSL's published taus known answer.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NGO_N_GENO 3
/* gen_func.hpp:15-16 */
static const double NGO_INF = 1e15;
static const double NGO_EPSILON = 1e-5;

/* ------------------------------------------------------------------------ */
/* GSL gsl_rng_taus (third-party; README.md:20 pins "gsl v1.15"; the source  */
/* is not under /root/reference).  Restated from the published algorithm of */
/* GSL rng/taus.c (P. L'Ecuyer, "Maximally equidistributed combined          */
/* Tausworthe generators", Math. Comp. 65 (1996)), as summarised in          */
/* SURVEY.md section 8c.  Call sites: ngsDist.cpp:179-180 (alloc + set),     */
/* gen_func.cpp:117-119 (draw_rnd -> gsl_rng_uniform).                       */
/* ------------------------------------------------------------------------ */
typedef struct { uint32_t s1, s2, s3; } ngo_taus;

#define NGO_TAUS(s, a, b, c, d) \
  ((((s) & (c)) << (d)) ^ ((((s) << (a)) ^ (s)) >> (b)))

uint32_t ngo_taus_get(ngo_taus *t) {
  t->s1 = NGO_TAUS(t->s1, 13, 19, 4294967294u, 12);
  t->s2 = NGO_TAUS(t->s2, 2, 25, 4294967288u, 4);
  t->s3 = NGO_TAUS(t->s3, 3, 11, 4294967280u, 17);
  return t->s1 ^ t->s2 ^ t->s3;
}

void ngo_taus_set(ngo_taus *t, uint64_t seed) {
  uint32_t s = (uint32_t)seed;
  if (s == 0) s = 1; /* default seed is 1 */
  t->s1 = 69069u * s;
  t->s2 = 69069u * t->s1;
  t->s3 = 69069u * t->s2;
  for (int i = 0; i < 6; i++) ngo_taus_get(t); /* warm-up */
}

double ngo_taus_uniform(ngo_taus *t) { return ngo_taus_get(t) / 4294967296.0; }

/* rnd_map_data, ngsDist.cpp:416-437: one draw per block, block ascending;
 * rnd_block = floor(draw_rnd(r, 0, n_blocks)) with draw_rnd = min + u*(max-min)
 * (gen_func.cpp:117-119).  The generator state runs on across replicates. */
void ngo_boot_block_map(ngo_taus *t, uint64_t n_blocks, uint64_t *map) {
  for (uint64_t b = 0; b < n_blocks; b++) {
    double r = 0 + ngo_taus_uniform(t) * (double)(n_blocks - 0);
    map[b] = (uint64_t)floor(r);
  }
}

/* ------------------------------------------------------------------------ */
/* Kernel-input construction ("prep")                                        */
/* ------------------------------------------------------------------------ */

/* logsum, gen_func.cpp:135-151 */
static double ngo_logsum(const double *a, int n) {
  double sum = 0;
  double M = a[0];
  for (int i = 1; i < n; i++) M = (a[i] >= M ? a[i] : M); /* max(a[i], M) macro, gen_func.hpp:23 */
  if (M == -INFINITY) return -INFINITY;
  for (int i = 0; i < n; i++) sum += exp(a[i] - M);
  return log(sum) + M;
}

/* post_prob with prior == NULL, gen_func.cpp:920-932 */
void ngo_post_prob(double *l) {
  double norm = ngo_logsum(l, NGO_N_GENO);
  for (int g = 0; g < NGO_N_GENO; g++) l[g] -= norm;
}

/* conv_space(geno, 3, log), gen_func.cpp:123-130 */
static void ngo_conv_log(double *l) {
  for (int g = 0; g < NGO_N_GENO; g++) {
    l[g] = log(l[g]);
    if (l[g] == -INFINITY) l[g] = -NGO_INF;
  }
}

/* call_geno(geno, 3, log_scale=true, N_thresh, call_thresh, miss_data=0),
 * gen_func.cpp:886-914 with array_max_pos/array_min_pos :73-98 (first max /
 * first min wins). */
void ngo_call_geno(double *l, double N_thresh, double call_thresh) {
  int max_pos = 0, min_pos = 0;
  double mx = -INFINITY, mn = +INFINITY;
  for (int g = 0; g < NGO_N_GENO; g++)
    if (l[g] > mx) { max_pos = g; mx = l[g]; }
  for (int g = 0; g < NGO_N_GENO; g++)
    if (l[g] < mn) { min_pos = g; mn = l[g]; }
  double max_pp = exp(l[max_pos]);
  if (l[min_pos] == l[max_pos]) max_pp = -1;
  if (max_pp < N_thresh)
    for (int g = 0; g < NGO_N_GENO; g++) l[g] = log((double)1 / NGO_N_GENO);
  if (max_pp >= call_thresh) {
    for (int g = 0; g < NGO_N_GENO; g++) l[g] = -NGO_INF;
    l[max_pos] = log(1);
  }
}

/* One (individual, site) of the binary path: read_data.cpp:37-45 followed by
 * ngsDist.cpp:165-174.  in: 3 raw doubles as stored in the file; out: the 3
 * normal-space values gen_dist reads.  Returns 1 if the reference would have
 * stopped with "NaN found!" (read_data.cpp:42-45). */
int ngo_prep_binary_one(const double *in, int in_logscale, int call_geno,
                        double N_thresh, double call_thresh, double *out) {
  double l[3] = {in[0], in[1], in[2]};
  if (!in_logscale) ngo_conv_log(l);
  ngo_post_prob(l);
  if (isnan(l[0]) || isnan(l[1]) || isnan(l[2])) return 1;
  if (call_geno) ngo_call_geno(l, N_thresh, call_thresh);
  for (int g = 0; g < 3; g++) {
    out[g] = exp(l[g]); /* conv_space(.., exp): exp never yields -inf */
  }
  return 0;
}

/* One (individual, site) of the gz-text GL/posterior path:
 * read_data.cpp:84-86,98 then ngsDist.cpp:165-174.  Note: no -inf clamp and no
 * NaN check on this path. */
void ngo_prep_text_probs_one(const double *in, int in_logscale, int call_geno,
                             double N_thresh, double call_thresh, double *out) {
  double l[3];
  for (int g = 0; g < 3; g++) l[g] = in_logscale ? in[g] : log(in[g]);
  ngo_post_prob(l);
  if (call_geno) ngo_call_geno(l, N_thresh, call_thresh);
  for (int g = 0; g < 3; g++) out[g] = exp(l[g]);
}

/* One (individual, site) of the gz-text called-genotype path:
 * read_data.cpp:21 (fill -INF), :88-95, :98, then ngsDist.cpp:172-173.
 * Returns 1 for g > 2 (reference: error "Genotypes must be coded as
 * {-1,0,1,2}"). */
int ngo_prep_text_geno_one(double code, double *out) {
  double l[3] = {-NGO_INF, -NGO_INF, -NGO_INF};
  int g = (int)code;
  if (g >= 0) {
    if (g > 2) return 1;
    l[g] = log(1);
  } else {
    l[0] = l[1] = l[2] = log((double)1 / NGO_N_GENO);
  }
  ngo_post_prob(l);
  for (int k = 0; k < 3; k++) out[k] = exp(l[k]);
  return 0;
}

/* A site whose text line was empty keeps its -INF fill (read_data.cpp:21,
 * 58-59) and is exp()ed by ngsDist.cpp:172-173 (call_geno first if set). */
void ngo_prep_empty_line_one(int call_geno, double N_thresh, double call_thresh,
                             double *out) {
  double l[3] = {-NGO_INF, -NGO_INF, -NGO_INF};
  if (call_geno) ngo_call_geno(l, N_thresh, call_thresh);
  for (int k = 0; k < 3; k++) out[k] = exp(l[k]);
}

/* Vectorised binary prep over a site-major block, as the file is laid out
 * (read_data.cpp:28-31: for s { for i { 3 doubles } }).  out may alias in. */
int ngo_prep_binary(const double *in, uint64_t n_elem, int in_logscale,
                    int call_geno, double N_thresh, double call_thresh,
                    double *out) {
  int bad = 0;
  for (uint64_t e = 0; e < n_elem; e++) {
    double tmp[3];
    bad |= ngo_prep_binary_one(in + 3 * e, in_logscale, call_geno, N_thresh,
                               call_thresh, tmp);
    out[3 * e] = tmp[0]; out[3 * e + 1] = tmp[1]; out[3 * e + 2] = tmp[2];
  }
  return bad;
}

/* miss_data, gen_func.cpp:862-868 with the abs() macro of gen_func.hpp:21 */
int ngo_miss_data(const double *p) {
  double a = p[0] - p[1], b = p[1] - p[2];
  a = (a >= 0 ? a : -a);
  b = (b >= 0 ? b : -b);
  return (a < NGO_EPSILON && b < NGO_EPSILON) ? 1 : 0;
}

/* ------------------------------------------------------------------------ */
/* Per-site two-individual EM (emOptim2.cpp), called with GL1.x == 1         */
/* ------------------------------------------------------------------------ */

/* normalize, emOptim2.cpp:69-75 */
static void ngo_normalize(double *tmp, int len) {
  double s = 0;
  for (int i = 0; i < len; i++) s += tmp[i];
  for (int i = 0; i < len; i++) tmp[i] /= s;
}

/* lik2 for one site, emOptim2.cpp:77-89 */
static double ngo_lik2(const double *sfs, const double *g1, const double *g2) {
  double res = 0;
  double tmp = 0;
  int inc = 0;
  for (int x = 0; x < 3; x++)
    for (int y = 0; y < 3; y++) {
      tmp += sfs[inc] * g1[x] * g2[y];
      inc++;
    }
  res += log(tmp);
  return res;
}

/* emStep2 for one site, emOptim2.cpp:91-109 */
static void ngo_emstep2(const double *pre, const double *g1, const double *g2,
                        double *post) {
  double inner[9];
  for (int x = 0; x < 9; x++) post[x] = 0.0;
  int inc = 0;
  for (int x = 0; x < 3; x++)
    for (int y = 0; y < 3; y++) {
      inner[inc] = pre[inc] * g1[x] * g2[y];
      inc++;
    }
  ngo_normalize(inner, 9);
  for (int x = 0; x < 9; x++) post[x] += inner[x];
  ngo_normalize(post, 9);
}

/* em2(sfs, GL1, GL2, tole, maxIter, 9), emOptim2.cpp:112-135.
 * Returns the number of emStep2 calls performed (1..maxIter). */
int ngo_em2(double *sfs, const double *g1, const double *g2, double tole,
            int maxIter) {
  double oldLik, lik;
  double tmp[9];
  int it;
  oldLik = ngo_lik2(sfs, g1, g2);
  for (it = 0; it < maxIter; it++) {
    ngo_emstep2(sfs, g1, g2, tmp);
    for (int i = 0; i < 9; i++) sfs[i] = tmp[i];
    lik = ngo_lik2(sfs, g1, g2);
    if (fabs(lik - oldLik) < tole) {
      oldLik = lik;
      it++;
      break;
    }
    oldLik = lik;
  }
  return it;
}

/* Optional: run the pair loop below on ANOTHER em2 with this signature -- oracle/_ref's ref_em2_tls, i.e. the
 * reference's own emOptim2.cpp compiled from where it lies -- so that the threaded CPU baseline can be timed on the
 * reference's code for the EM (bench.py, "reference-em2").  NULL (default) = the restatement above. */
typedef int (*ngo_em2_fn)(double *, const double *, const double *, double, int);
static ngo_em2_fn ngo_em2_hook = 0;
void ngo_set_em2_hook(ngo_em2_fn fn) { ngo_em2_hook = fn; }

/* ------------------------------------------------------------------------ */
/* gen_dist: ngsDist.cpp:325-404                                             */
/* ------------------------------------------------------------------------ */

/* Raw accumulation (ngsDist.cpp:333-364) for one pair.
 *   p        : prepared normal-space values, individual-major
 *              p[(i*n_sites_total + s)*3 + g]  == in_geno_lkl[i][s][g]
 *   site_src : NULL for the full data set, else site_src[s] = original site
 *              that rnd_map_data put at position s (geno_lkl[i][s] =
 *              in_geno_lkl[i][site_src[s]], ngsDist.cpp:433-434)
 *   n_sites  : number of positions visited (already truncated for rep > 0)
 * Accumulation order: s ascending, g1 outer, g2 inner, all nine terms.
 * em_iters (may be NULL) receives the total number of EM steps. */
void ngo_pair_accum(const double *p, uint64_t n_sites_total,
                    const uint64_t *site_src, uint64_t n_sites, uint64_t i1,
                    uint64_t i2, const double *score /*[9]*/, int pairwise_del,
                    int indep_geno, double *sum_out, uint64_t *cnt_out,
                    uint64_t *em_iters) {
  uint64_t cnt = 0, iters = 0;
  double dist = 0;
  const double *b1 = p + i1 * n_sites_total * 3;
  const double *b2 = p + i2 * n_sites_total * 3;
  for (uint64_t s = 0; s < n_sites; s++) {
    uint64_t src = site_src ? site_src[s] : s;
    const double *g1 = b1 + src * 3;
    const double *g2 = b2 + src * 3;
    if (pairwise_del && (ngo_miss_data(g1) || ngo_miss_data(g2))) continue;
    double sfs[9];
    for (int k = 0; k < 9; k++) sfs[k] = (double)1 / 9; /* ngsDist.cpp:340 */
    if (!indep_geno) iters += (uint64_t)(ngo_em2_hook ? ngo_em2_hook(sfs, g1, g2, 0.001, 50) : ngo_em2(sfs, g1, g2, 0.001, 50));
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        dist += score[3 * a + b] * (indep_geno ? g1[a] * g2[b] : sfs[3 * a + b]);
    cnt++;
  }
  *sum_out = dist;
  *cnt_out = cnt;
  if (em_iters) *em_iters = iters;
}

/* The tail of gen_dist, ngsDist.cpp:372-401.  Models 3..6 and anything else
 * are rejected by the caller (the reference calls error()); here they yield
 * NaN and *err = 1. */
double ngo_finish(double sum, uint64_t cnt, uint64_t tot_sites,
                  uint64_t evol_model, int *err) {
  if (err) *err = 0;
  if (tot_sites > 0) cnt = tot_sites;
  double dist = sum;
  dist /= (double)cnt;
  if (evol_model == 0) {
    dist = dist;
  } else if (evol_model == 1) {
    dist = -log(1 - dist);
  } else if (evol_model == 2) {
    dist = -log(1 - (dist * 4 / 3)) * 3 / 4;
  } else {
    if (err) *err = 1;
    return NAN;
  }
  return dist;
}

/* All pairs, row-major upper triangle (ngsDist.cpp:244-245), threaded the way
 * the reference threads: one task per pair, any order, disjoint outputs. */
typedef struct {
  const double *p; uint64_t n_ind, n_sites_total; const uint64_t *site_src;
  uint64_t n_sites; const double *score; int pairwise_del, indep_geno;
  double *sum; uint64_t *cnt; uint64_t *iters;
  uint64_t next; uint64_t n_pairs; pthread_mutex_t mu;
} ngo_job;

static void ngo_pair_from_index(uint64_t n, uint64_t idx, uint64_t *i1, uint64_t *i2) {
  uint64_t i = 0, row = n - 1;
  while (idx >= row) { idx -= row; row--; i++; }
  *i1 = i; *i2 = i + 1 + idx;
}

static void *ngo_worker(void *arg) {
  ngo_job *j = (ngo_job *)arg;
  for (;;) {
    pthread_mutex_lock(&j->mu);
    uint64_t lo = j->next;
    uint64_t hi = lo + 16;
    if (hi > j->n_pairs) hi = j->n_pairs;
    j->next = hi;
    pthread_mutex_unlock(&j->mu);
    if (lo >= hi) break;
    for (uint64_t k = lo; k < hi; k++) {
      uint64_t i1, i2;
      ngo_pair_from_index(j->n_ind, k, &i1, &i2);
      ngo_pair_accum(j->p, j->n_sites_total, j->site_src, j->n_sites, i1, i2,
                     j->score, j->pairwise_del, j->indep_geno, &j->sum[k],
                     &j->cnt[k], j->iters ? &j->iters[k] : NULL);
    }
  }
  return NULL;
}

int ngo_all_pairs(const double *p, uint64_t n_ind, uint64_t n_sites_total,
                  const uint64_t *site_src, uint64_t n_sites,
                  const double *score, int pairwise_del, int indep_geno,
                  int n_threads, double *sum, uint64_t *cnt, uint64_t *iters) {
  ngo_job j;
  j.p = p; j.n_ind = n_ind; j.n_sites_total = n_sites_total;
  j.site_src = site_src; j.n_sites = n_sites; j.score = score;
  j.pairwise_del = pairwise_del; j.indep_geno = indep_geno;
  j.sum = sum; j.cnt = cnt; j.iters = iters;
  j.next = 0; j.n_pairs = n_ind * (n_ind - 1) / 2;
  pthread_mutex_init(&j.mu, NULL);
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  pthread_t th[256];
  for (int t = 0; t < n_threads; t++)
    if (pthread_create(&th[t], NULL, ngo_worker, &j)) return -1;
  for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  pthread_mutex_destroy(&j.mu);
  return 0;
}

/* "%.10f" of one cell, the format join() uses (gen_func.cpp:479-496). */
int ngo_format_cell(double v, char *buf, int buflen) {
  return snprintf(buf, (size_t)buflen, "%.10f", v);
}

/* ------------------------------------------------------------------------ */
/* Synthetic input (this repo's own recipe): a         */
/* counter-based generator so host, device and tests agree with no files.    */
/* Element (s, i) -> three already-normalised normal-space values.           */
/* ------------------------------------------------------------------------ */
static uint64_t ngo_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void ngo_synth_one(uint64_t seed, uint64_t n_ind, uint64_t s, uint64_t i,
                   double miss_frac, double *out) {
  uint64_t base = seed * 0x9E3779B97F4A7C15ull;
  uint64_t e = (s * n_ind + i);
  double x[3];
  for (int g = 0; g < 3; g++) {
    uint64_t z = ngo_mix64(base + (e * 3 + (uint64_t)g));
    double u = ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    x[g] = (u * u) * u;
  }
  double t = (x[0] + x[1]) + x[2];
  out[0] = x[0] / t; out[1] = x[1] / t; out[2] = x[2] / t;
  if (miss_frac > 0) {
    uint64_t z = ngo_mix64(base ^ (0xD1B54A32D192ED03ull + e));
    double u = ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    if (u < miss_frac) out[0] = out[1] = out[2] = (double)1 / 3;
  }
}

/* individual-major fill: p[(i*n_sites + s)*3 + g], sites s0 .. s0+n_sites-1
 * of individuals i0 .. i0+n_sub-1 of an n_ind-individual data set */
void ngo_synth_fill_indmajor(uint64_t seed, uint64_t n_ind, uint64_t i0,
                             uint64_t n_sub, uint64_t s0, uint64_t n_sites,
                             double miss_frac, double *p) {
  for (uint64_t i = 0; i < n_sub; i++)
    for (uint64_t s = 0; s < n_sites; s++)
      ngo_synth_one(seed, n_ind, s0 + s, i0 + i, miss_frac,
                    p + (i * n_sites + s) * 3);
}
