// tests/host_sanitize/stub_engine.cpp -- TEST INFRASTRUCTURE, never shipped: the device-side entry points of
// include/ngsdist_amd.h with NO compute behind them, so that the C++ host (ngsdist_amd/csrc/host/ngsdist_host.cpp: argument
// parsing, the binary / text / stdin readers, the pinned-buffer pipeline, site ranges, bootstrap bookkeeping, printing)
// can run under AddressSanitizer / UBSan in the CPU container (GPU sanitizers are not available on the pool).  Uploads are
// read in full (so that an over-read is seen), results are fixed numbers.  host_util.cpp is linked for real.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ngsdist_amd.h"

struct ngd_engine {
  ngd_config cfg;
  std::vector<double> stage[3];  // (a ring, as the engine's: a buffer comes back a turn of the ring later)
  uint64_t cap = 0;
  int cur = 0, lent = -1;
  uint64_t opt_piece_mib = 0, opt_ring = 0;
  bool committed = false;
  uint64_t seen_sites = 0;
  double checksum = 0;
  uint32_t kept = 0;  // matrices of the last batch call left "in the engine" (ngd_fetch_matrix)
};

static std::string g_err;
static int fail(int code, const char *msg) { g_err = msg; return code; }

extern "C" {
const char *ngd_last_error(void) { return g_err.c_str(); }
int ngd_abi_version(void) { return NGD_ABI_VERSION; }
int ngd_device_count(void) { return 4; }
int ngd_device_memory(int, uint64_t *free_b, uint64_t *total_b) {
  if (free_b) *free_b = 200ull << 30;
  if (total_b) *total_b = 288ull << 30;
  return NGD_OK;
}
uint64_t ngd_n_pairs(uint64_t n) { return n * (n - 1) / 2; }
uint64_t ngd_pair_index(uint64_t n, uint64_t i, uint64_t j) { return i * (2 * n - i - 1) / 2 + (j - i - 1); }

int ngd_create(const ngd_config *cfg, ngd_engine **out) {
  if (!cfg || !out) return fail(NGD_E_INVALID, "stub: null argument");
  if (cfg->n_ind < 2 || cfg->n_sites < 1) return fail(NGD_E_INVALID, "stub: geometry");
  ngd_engine *e = new ngd_engine();
  e->cfg = *cfg;
  e->cap = 257;  // an odd capacity: the host must not assume anything about it
  for (auto &s : e->stage) s.assign(e->cap * cfg->n_ind * 3, -1.0);
  *out = e;
  return NGD_OK;
}
void ngd_destroy(ngd_engine *e) { delete e; }
static void touch(ngd_engine *e, const double *p, uint64_t n_doubles) {
  double s = 0;
  for (uint64_t k = 0; k < n_doubles; k++) s += p[k];  // every double the host promised is read
  e->checksum += s;
}
int ngd_upload_sites(ngd_engine *e, const double *p, uint64_t s0, uint64_t n) {
  if (!e || !p || e->committed || s0 + n > e->cfg.n_sites) return fail(NGD_E_INVALID, "stub: upload_sites");
  touch(e, p, n * e->cfg.n_ind * 3);
  e->seen_sites += n;
  return NGD_OK;
}
int ngd_stage_acquire(ngd_engine *e, double **buf, uint64_t *cap) {
  if (!e || !buf || !cap || e->committed || e->lent >= 0) return fail(NGD_E_INVALID, "stub: stage_acquire");
  e->lent = e->cur;
  *buf = e->stage[e->cur].data();
  *cap = e->cap;
  return NGD_OK;
}
int ngd_stage_submit(ngd_engine *e, uint64_t s0, uint64_t n, const ngd_prep *prep) {
  if (!e || !prep || e->lent < 0 || n > e->cap || s0 + n > e->cfg.n_sites) return fail(NGD_E_INVALID, "stub: stage_submit");
  touch(e, e->stage[e->lent].data(), n * e->cfg.n_ind * 3);
  std::fill(e->stage[e->lent].begin(), e->stage[e->lent].end(), -1.0);
  e->cur = (e->cur + 1) % 3;
  e->lent = -1;
  e->seen_sites += n;
  return NGD_OK;
}
int ngd_set_option(ngd_engine *e, int option, uint64_t value) {
  if (!e) return fail(NGD_E_INVALID, "stub: set_option");
  if (option == NGD_OPT_STAGE_PIECE_MIB) e->opt_piece_mib = value;
  else if (option == NGD_OPT_STAGE_RING) e->opt_ring = value;
  else if (option == NGD_OPT_EAGER_FULL) (void)value;
  else return fail(NGD_E_INVALID, "stub: unknown option");
  return NGD_OK;
}
int ngd_commit(ngd_engine *e) {
  if (!e) return fail(NGD_E_INVALID, "stub: commit");
  if (e->checksum != e->checksum) return fail(NGD_E_NAN, "NaN found! Is the file format correct?");
  if (e->seen_sites != e->cfg.n_sites) return fail(NGD_E_INVALID, "stub: not every site was uploaded");
  e->committed = true;
  return NGD_OK;
}
static int fill(ngd_engine *e, uint32_t n_mat, uint64_t n_blocks, uint64_t B, bool need_blocks, double *sum, uint64_t *cnt) {
  if (!e || !e->committed || (!sum != !cnt)) return fail(NGD_E_INVALID, "stub: run");
  if (need_blocks && (!B || !n_blocks || n_blocks > e->cfg.n_sites / B)) return fail(NGD_E_INVALID, "stub: block geometry");
  const uint64_t np = ngd_n_pairs(e->cfg.n_ind);
  e->kept = 0;
  if (!sum) {  // the matrices stay "in the engine" for ngd_fetch_matrix
    e->kept = n_mat;
    return NGD_OK;
  }
  for (uint64_t k = 0; k < n_mat * np; k++) { sum[k] = 0.25 * (double)(k % 7); cnt[k] = 1 + k % 3; }
  return NGD_OK;
}
int ngd_fetch_matrix(ngd_engine *e, uint32_t which, double *sum, uint64_t *cnt) {
  if (!e || which >= e->kept || !sum || !cnt) return fail(NGD_E_INVALID, "stub: fetch_matrix");
  const uint64_t np = ngd_n_pairs(e->cfg.n_ind);
  for (uint64_t k = 0; k < np; k++) { sum[k] = 0.25 * (double)((which * np + k) % 7); cnt[k] = 1 + (which * np + k) % 3; }
  return NGD_OK;
}
int ngd_run(ngd_engine *e, const uint64_t *bm, uint64_t nb, uint64_t B, double *sum, uint64_t *cnt) {
  if (bm) for (uint64_t b = 0; b < nb; b++) if (bm[b] >= nb) return fail(NGD_E_INVALID, "stub: map entry");
  return fill(e, 1, nb, B, bm != nullptr, sum, cnt);
}
int ngd_run_batch(ngd_engine *e, const uint64_t *bm, uint32_t n_rep, uint64_t nb, uint64_t B, double *sum, uint64_t *cnt) {
  if (!bm || !n_rep) return fail(NGD_E_INVALID, "stub: run_batch");
  for (uint64_t b = 0; b < n_rep * nb; b++) if (bm[b] >= nb) return fail(NGD_E_INVALID, "stub: map entry");
  return fill(e, n_rep, nb, B, true, sum, cnt);
}
int ngd_run_job(ngd_engine *e, const uint64_t *bm, uint32_t n_rep, uint64_t nb, uint64_t B, double *sum, uint64_t *cnt) {
  if (n_rep && !bm) return fail(NGD_E_INVALID, "stub: run_job");
  for (uint64_t b = 0; b < n_rep * nb; b++) if (bm[b] >= nb) return fail(NGD_E_INVALID, "stub: map entry");
  return fill(e, n_rep + 1, nb, B, n_rep != 0, sum, cnt);
}
int ngd_run_mult_batch(ngd_engine *e, const uint32_t *m, uint32_t n_rep, uint64_t nb, uint64_t B, double *sum, uint64_t *cnt) {
  if (!m || !n_rep) return fail(NGD_E_INVALID, "stub: run_mult_batch");
  uint64_t s = 0;
  for (uint64_t b = 0; b < n_rep * nb; b++) s += m[b];
  (void)s;
  return fill(e, n_rep, nb, B, true, sum, cnt);
}
// (the host reports what the fix-up pass of a one-image engine did; the stub: a data set of 77 individuals has 5 pairs
// left alone in its first call -- the host's warning path -- and 3 recomputed in every call)
int ngd_last_fixup(const ngd_engine *e, ngd_fixup_info *info) {
  if (!e || !info) return fail(NGD_E_INVALID, "stub: last_fixup");
  *info = ngd_fixup_info{};
  if (e->cfg.n_ind == 77) { info->flagged = 8; info->recomputed = 3; info->skipped = 5; info->ms = 0.5; }
  return NGD_OK;
}
}  // extern "C"
