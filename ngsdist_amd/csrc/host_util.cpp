// host_util.cpp -- the host-side pieces of the C ABI that must use the HOST's
// libm / integer arithmetic to reproduce the reference's printed cells exactly:
// the tail of gen_dist() (reference ngsDist.cpp:372-401) and the bootstrap block
// draw (ngsDist.cpp:416-423 over gsl_rng_taus, seeded at :179-180).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

#include "../../include/ngsdist_amd.h"
#include "ngd_shard.h"

namespace {
// gsl_rng_taus: GSL is a third-party dependency of the reference (README.md:20,
// "gsl v1.15") that is not under /root/reference.  Algorithm restated from its
// published description (L'Ecuyer 1996 combined Tausworthe, three components,
// seeding by the 69069 LCG, six warm-up draws); pinned by GSL's own known
// answer in tests/test_host_abi.py (seed 1, 10000th output = 2733957125).
inline uint32_t taus_step(uint32_t s, int a, int b, uint32_t c, int d) {
  return ((s & c) << d) ^ (((s << a) ^ s) >> b);
}
inline uint32_t taus_get(uint32_t st[3]) {
  st[0] = taus_step(st[0], 13, 19, 4294967294u, 12);
  st[1] = taus_step(st[1], 2, 25, 4294967288u, 4);
  st[2] = taus_step(st[2], 3, 11, 4294967280u, 17);
  return st[0] ^ st[1] ^ st[2];
}
}  // namespace

static void finish_range(const double *sum, const uint64_t *cnt, uint64_t lo, uint64_t hi, uint64_t tot_sites,
                         uint64_t evol_model, double *dist) {
  for (uint64_t k = lo; k < hi; k++) {
    uint64_t c = cnt[k];
    if (tot_sites > 0) c = tot_sites;
    double d = sum[k];
    d /= (double)c;
    if (evol_model == 1)
      d = -log(1 - d);
    else if (evol_model == 2)
      d = -log(1 - (d * 4 / 3)) * 3 / 4;
    dist[k] = d;
  }
}

extern "C" {

void ngd_taus_seed(uint32_t st[3], uint64_t seed) {
  uint32_t s = (uint32_t)seed;
  if (s == 0) s = 1;
  st[0] = 69069u * s;
  st[1] = 69069u * st[0];
  st[2] = 69069u * st[1];
  for (int i = 0; i < 6; i++) taus_get(st);
}

double ngd_taus_uniform(uint32_t st[3]) { return taus_get(st) / 4294967296.0; }

uint32_t ngd_taus_get(uint32_t st[3]) { return taus_get(st); }

void ngd_boot_block_map(uint32_t st[3], uint64_t n_blocks, uint64_t *block_map) {
  for (uint64_t b = 0; b < n_blocks; b++) {
    // draw_rnd(r, 0, n_blocks) = min + gsl_rng_uniform(r) * (max - min), gen_func.cpp:117-119
    double r = 0 + ngd_taus_uniform(st) * (double)(n_blocks - 0);
    block_map[b] = (uint64_t)floor(r);
  }
}

// Which shard computes pair (i1 < i2): the owner of its 128 x 128 pair tile (ngd_shard.h).
uint32_t ngd_shard_of_pair(uint64_t n_ind, uint64_t i1, uint64_t i2, uint32_t shard_world) {
  if (shard_world <= 1) return 0;
  static thread_local uint32_t c_nt = 0, c_world = 0;
  static thread_local std::vector<uint32_t> c_owner;
  const uint32_t n_t = (uint32_t)((n_ind + 127) / 128);
  if (c_nt != n_t || c_world != shard_world) {
    c_owner = ngd_tile_owners(n_t, shard_world);
    c_nt = n_t;
    c_world = shard_world;
  }
  return c_owner[ngd_tile_id(n_t, i1 / 128, i2 / 128)];
}

int ngd_finish(const double *sum, const uint64_t *cnt, uint64_t n_pairs, uint64_t tot_sites,
               uint64_t evol_model, double *dist) {
  if (evol_model > 2) return NGD_E_MODEL;  // reference: error("... model not yet supported")
  if (!sum || !cnt || !dist) return NGD_E_INVALID;
  // per-cell and order-free, so threads change nothing but the wall time
  unsigned nt = 1;
  if (n_pairs >= (1u << 16)) nt = std::min(n_pairs >= (1u << 18) ? 16u : 8u, std::max(1u, std::thread::hardware_concurrency()));
  if (nt <= 1) {
    finish_range(sum, cnt, 0, n_pairs, tot_sites, evol_model, dist);
  } else {
    std::vector<std::thread> th;
    const uint64_t per = (n_pairs + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
      const uint64_t lo = t * per, hi = std::min(n_pairs, lo + per);
      if (lo >= hi) break;
      th.emplace_back(finish_range, sum, cnt, lo, hi, tot_sites, evol_model, dist);
    }
    for (auto &t : th) t.join();
  }
  return NGD_OK;
}

}  // extern "C"
