// host_util.cpp -- the host-side pieces of the C ABI that must use the HOST's
// libm / integer arithmetic to reproduce the reference's printed cells exactly:
// the tail of gen_dist() (reference ngsDist.cpp:372-401) and the bootstrap block
// draw (ngsDist.cpp:416-423 over gsl_rng_taus, seeded at :179-180).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ngsdist_amd.h"
#include "ngd_shard.h"

namespace {
// gsl_rng_taus: GSL is a third-party dependency of the reference (README.md:20,
// "gsl v1.15") that is not under /root/reference.  Algorithm restated from its
// published description (L'Ecuyer 1996 combined Tausworthe, three components,
// seeding by the 69069 LCG, six warm-up draws); pinned by GSL's own known
// answer in tests/test_abi.py (seed 1, 10000th output = 2733957125).
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

// d = sum / cnt (ngsDist.cpp:376) for a range of cells, apart from the model's transform: a loop the compiler can
// vectorise (IEEE division is correctly rounded at any width: the same bits as the scalar loop; -ffp-contract=off and no
// fast-math here).  (double)c for counts below 2^52 -- any site count -- is the exponent trick: exact, and free of the
// unsigned 64-bit conversion no AVX2 instruction does.
__attribute__((target_clones("avx2", "default"))) static void divide_range(const double *sum, const uint64_t *cnt, uint64_t lo,
                                                                           uint64_t hi, uint64_t tot_sites, double *dist) {
  const uint64_t magic = 0x4330000000000000ull;  // 2^52 as a bit pattern
  if (tot_sites > 0) {
    const double c = (double)tot_sites;
    for (uint64_t k = lo; k < hi; k++) dist[k] = sum[k] / c;
    return;
  }
  uint64_t big = 0;
  for (uint64_t k = lo; k < hi; k++) big |= cnt[k];
  if (big >> 52) {  // (never a site count; kept exact all the same)
    for (uint64_t k = lo; k < hi; k++) dist[k] = sum[k] / (double)cnt[k];
    return;
  }
  for (uint64_t k = lo; k < hi; k++) {
    const uint64_t bits = cnt[k] | magic;
    double c;
    memcpy(&c, &bits, 8);
    dist[k] = sum[k] / (c - 4503599627370496.0);
  }
}

static void finish_range(const double *sum, const uint64_t *cnt, uint64_t lo, uint64_t hi, uint64_t tot_sites,
                         uint64_t evol_model, double *dist) {
  // in pieces that stay in the cache between the two passes
  for (uint64_t a = lo; a < hi; a += 4096) {
    const uint64_t b = std::min(hi, a + 4096);
    divide_range(sum, cnt, a, b, tot_sites, dist);
    if (evol_model == 1) {
      for (uint64_t k = a; k < b; k++) dist[k] = -log(1 - dist[k]);
    } else if (evol_model == 2) {
      for (uint64_t k = a; k < b; k++) dist[k] = -log(1 - (dist[k] * 4 / 3)) * 3 / 4;
    }
  }
}

// "%.10f" of one cell, byte-for-byte what printf writes (the reference formats every cell with
// snprintf("%.10f"), gen_func.cpp:483-486), without going through printf's arbitrary-precision path:
// x = m * 2^q exactly, so round-half-even(x * 10^10) is an integer computed exactly in 128 bits
// (m < 2^53, 10^10 < 2^34); printf rounds the exact binary value the same way (round-to-nearest mode).
// Values of 2^29 and above, which no distance reaches, and non-finite cells go to snprintf itself.
static inline char *fmt_fixed10(double x, char *o) {
  uint64_t bits;
  memcpy(&bits, &x, 8);
  const int e = (int)((bits >> 52) & 0x7FF);
  uint64_t m = bits & ((1ull << 52) - 1);
  if (e >= 1023 + 29) return o + snprintf(o, 400, "%.10f", x);  // large, inf, nan
  if (bits >> 63) *o++ = '-';
  int q;  // x = m * 2^q
  if (e == 0) q = -1074;
  else { m |= 1ull << 52; q = e - 1075; }
  const unsigned __int128 P = (unsigned __int128)m * 10000000000ull;
  uint64_t N;
  const int sh = -q;  // e < 1052 -> q < 0
  if (sh >= 100) N = 0;  // P < 2^87: below one half of the last digit
  else {
    N = (uint64_t)(P >> sh);
    const unsigned __int128 rem = P & (((unsigned __int128)1 << sh) - 1), half = (unsigned __int128)1 << (sh - 1);
    if (rem > half || (rem == half && (N & 1))) N++;
  }
  const uint64_t ip = N / 10000000000ull;
  uint64_t fp = N - ip * 10000000000ull;
  if (ip < 10) {  // (every distance but a saturated one)
    *o++ = (char)('0' + ip);
  } else {
    char tmp[24];
    int n = 0;
    uint64_t v = ip;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *o++ = tmp[--n];
  }
  *o++ = '.';
  // ten digits, two at a time (32-bit arithmetic: fp < 10^10 = two halves below 10^5)
  static const char kPairs[] =
      "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960"
      "616263646566676869707172737475767778798081828384858687888990919293949596979899";
  const uint32_t hi = (uint32_t)(fp / 100000), lo = (uint32_t)(fp - (uint64_t)hi * 100000);
  const uint32_t h0 = hi / 1000, h1 = hi % 1000;  // hi = h0 (2 digits) h1 (3 digits)
  memcpy(o, kPairs + 2 * h0, 2);
  o[2] = (char)('0' + h1 / 100);
  memcpy(o + 3, kPairs + 2 * (h1 % 100), 2);
  const uint32_t l0 = lo / 1000, l1 = lo % 1000;
  memcpy(o + 5, kPairs + 2 * l0, 2);
  o[7] = (char)('0' + l1 / 100);
  memcpy(o + 8, kPairs + 2 * (l1 % 100), 2);
  return o + 10;
}

// A small persistent pool for the per-cell host work (the tail of gen_dist over up to millions of cells per call).
// Spawning and joining 16 threads costs ~0.3 ms per call ([measured] on the GPU box: 62 437 cells 0.32 ms, 8.1e6 cells in
// one call 2.8 ms but in 8 calls 5.4 ms), which is most of the tail of a multi-GPU step and half of a chunked cfg 5
// tail.  Workers sleep on a condition variable between calls; one job at a time (a second caller runs its job inline
// on its own thread).  Never destroyed: worker threads must not outlive their mutex at process exit.
namespace {
struct HostPool {
  std::mutex m, busy;
  std::condition_variable cv_go, cv_done;
  std::vector<std::thread> workers;
  std::function<void(unsigned)> job;
  unsigned n_parts = 0, next = 0, done = 0;
  uint64_t generation = 0;
  explicit HostPool(unsigned n) {
    for (unsigned t = 0; t < n; t++)
      workers.emplace_back([this]() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
          cv_go.wait(lk, [&] { return generation != seen; });
          seen = generation;
          while (next < n_parts) {
            const unsigned part = next++;
            lk.unlock();
            job(part);
            lk.lock();
            if (++done == n_parts) cv_done.notify_one();
          }
        }
      });
    for (auto &w : workers) w.detach();
  }
  // runs fn(0) .. fn(n - 1), the caller taking parts too; returns when all are done
  void run(unsigned n, const std::function<void(unsigned)> &fn) {
    std::unique_lock<std::mutex> one(busy, std::try_to_lock);
    if (!one.owns_lock() || n <= 1) {  // pool in use by another thread (or nothing to split): do it here
      for (unsigned k = 0; k < n; k++) fn(k);
      return;
    }
    std::unique_lock<std::mutex> lk(m);
    job = fn;
    n_parts = n; next = 0; done = 0;
    generation++;
    cv_go.notify_all();
    while (next < n_parts) {
      const unsigned part = next++;
      lk.unlock();
      fn(part);
      lk.lock();
      ++done;
    }
    cv_done.wait(lk, [&] { return done == n_parts; });
    n_parts = 0;
  }
};
HostPool &host_pool() {
  static HostPool *p = new HostPool(std::min(15u, std::max(1u, std::thread::hardware_concurrency()) - 1));
  return *p;
}
// ... and a wide one for the tail of a whole bootstrap job (millions of cells in one call): a burst of a millisecond on as
// many cores as the box has, up to 64 ([measured, round 6, 256 hardware threads, a CPU quota of 16] 8.1e6 cells of model 1:
// 2.4 ms on 16 threads; a quota counts CPU time per period, and 40 ms of it in a burst is far inside one).  Made on first use.
HostPool &host_pool_wide() {
  static HostPool *p = new HostPool(std::min(63u, std::max(1u, std::thread::hardware_concurrency()) - 1));
  return *p;
}
}  // namespace


extern "C" {

// The print block of one matrix, ngsDist.cpp:282-287: "\n<n_ind>\n", then per individual its label and
// n_ind cells "\t%.10f" (join(), gen_func.cpp:479-496) and a newline.  Rows are formatted by threads.
int64_t ngd_format_matrix(const double *dist, uint64_t n_ind, const char *const *labels, char *out, uint64_t cap,
                          uint32_t n_threads) {
  if (!dist || !labels || n_ind < 2) return NGD_E_INVALID;
  unsigned nt = n_threads ? n_threads : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (n_ind < 64) nt = 1;
  nt = (unsigned)std::min<uint64_t>(nt, n_ind);
  // shares of rows, finer than the threads (rows near the top hold more upper-triangle cells: dist is read row-wise there,
  // column-wise below the diagonal); every share formats into a buffer of its own, sized by the longest a cell can be
  const unsigned n_parts = nt == 1 ? 1 : (unsigned)std::min<uint64_t>(n_ind, 4ull * nt);
  // The shares' buffers are the calling thread's and are kept from call to call (a job prints a hundred matrices of 13 MB: a
  // fresh vector per share and call is an mmap, its zero fill and its page faults every time)
  struct Scratch {
    std::vector<char *> buf;
    std::vector<size_t> cap;
    ~Scratch() { for (char *b : buf) free(b); }
    bool reserve(unsigned t, size_t need) {  // (keeps the contents: a share may grow while it is being written)
      if (t < cap.size() && cap[t] >= need) return true;
      if (t >= buf.size()) return false;
      char *nb = (char *)realloc(buf[t], need);
      if (!nb) return false;
      buf[t] = nb;
      cap[t] = need;
      return true;
    }
  };
  static thread_local Scratch scratch_tl;
  Scratch &scratch = scratch_tl;  // (the pool's threads must see the CALLER's buffers, not thread-local ones of their own)
  if (scratch.buf.size() < n_parts) { scratch.buf.resize(n_parts, nullptr); scratch.cap.resize(n_parts, 0); }
  std::vector<size_t> part_len(n_parts, 0);
  std::atomic<bool> oom{false};
  std::vector<size_t> label_len(n_ind);
  for (uint64_t i = 0; i < n_ind; i++) label_len[i] = strlen(labels[i]);
  // (more threads asked for than the pool of 16 holds: the wide pool's 64 -- [measured, round 6] threads of the call's own,
  // made and joined per matrix, took three times as long as 16 pooled ones)
  const bool wide = n_threads > 16;
  auto run_parts = [&](const std::function<void(unsigned)> &fn) {
    if (nt == 1) {
      for (unsigned t = 0; t < n_parts; t++) fn(t);
    } else {
      (wide ? host_pool_wide() : host_pool()).run(n_parts, fn);
    }
  };
  char head[32];
  const int hn = snprintf(head, sizeof(head), "\n%lu\n", (unsigned long)n_ind);

  // The regular matrix -- every cell prints as "d.dddddddddd", 12 bytes: what distances are -- is formatted ONCE per pair
  // (the printed matrix is symmetric: the general form below formats every cell twice and walks a column of the upper
  // triangle for the cells under the diagonal) into 12-byte slots in pair order; the rows, all of one known length, are
  // then put together from the slots straight into `out`.  Any other cell (a sign, nan, inf, 10 and above) anywhere: the
  // general form.
  if (out && nt > 1) {
    uint64_t total_fixed = (uint64_t)hn;
    std::vector<uint64_t> row_at(n_ind);
    for (uint64_t i = 0; i < n_ind; i++) { row_at[i] = total_fixed; total_fixed += label_len[i] + n_ind * 13 + 1; }
    const uint64_t n_pairs = n_ind * (n_ind - 1) / 2;
    struct Slots { char *p = nullptr; size_t cap = 0; ~Slots() { free(p); } };
    static thread_local Slots slots_tl;
    Slots &slots = slots_tl;
    if (cap >= total_fixed && (slots.cap >= n_pairs * 12 || [&] {
          char *np_ = (char *)realloc(slots.p, n_pairs * 12);
          if (np_) { slots.p = np_; slots.cap = n_pairs * 12; }
          return np_ != nullptr;
        }())) {
      std::atomic<bool> irregular{false};
      char *const sl = slots.p;
      run_parts([&](unsigned t) {
        const uint64_t lo = n_pairs * t / n_parts, hi = n_pairs * (t + 1) / n_parts;
        char tmp[448];
        for (uint64_t k = lo; k < hi; k++) {
          const char *e = fmt_fixed10(dist[k], tmp);
          if (e - tmp != 12) { irregular.store(true, std::memory_order_relaxed); return; }
          memcpy(sl + 12 * k, tmp, 12);
        }
      });
      if (!irregular.load()) {
        memcpy(out, head, (size_t)hn);
        run_parts([&](unsigned t) {
          const uint64_t lo = n_ind * t / n_parts, hi = n_ind * (t + 1) / n_parts;
          for (uint64_t i = lo; i < hi; i++) {
            char *o = out + row_at[i];
            memcpy(o, labels[i], label_len[i]);
            o += label_len[i];
            uint64_t k = i - 1;  // cell (j, i), j < i: pair index j (2n - j - 1) / 2 + (i - j - 1), its step from j to j + 1 is n - j - 2
            for (uint64_t j = 0; j < i; j++) {
              *o++ = '\t';
              memcpy(o, sl + 12 * k, 12);
              o += 12;
              k += n_ind - j - 2;
            }
            memcpy(o, "\t0.0000000000", 13);
            o += 13;
            const char *row = sl + 12 * (i * (2 * n_ind - i - 1) / 2);  // the slots of cells (i, j), j > i, side by side
            for (uint64_t j = i + 1; j < n_ind; j++, row += 12) {
              *o++ = '\t';
              memcpy(o, row, 12);
              o += 12;
            }
            *o = '\n';
          }
        });
        return (int64_t)total_fixed;
      }
    }
  }
  auto rows = [&](unsigned t) {
    const uint64_t lo = n_ind * t / n_parts, hi = n_ind * (t + 1) / n_parts;
    size_t need = 0;
    for (uint64_t i = lo; i < hi; i++) need += label_len[i] + n_ind * 24 + 1;  // ("\t" + sign + 9 digits + "." + 10 = 22 at most below 2^29)
    if (!scratch.reserve(t, need + 512)) { oom = true; return; }
    char *o = scratch.buf[t], *lim = o + scratch.cap[t] - 450;  // (a cell snprintf may write -- 2^29 and above, 1e300 -- needs up to ~330)
    auto grow = [&]() {  // huge cells (never for distances)
      const size_t used = (size_t)(o - scratch.buf[t]);
      if (!scratch.reserve(t, scratch.cap[t] * 2 + 1024)) { oom = true; return false; }
      o = scratch.buf[t] + used;
      lim = scratch.buf[t] + scratch.cap[t] - 450;
      return true;
    };
    for (uint64_t i = lo; i < hi; i++) {
      if (o + label_len[i] > lim && !grow()) return;
      memcpy(o, labels[i], label_len[i]);
      o += label_len[i];
      // dist_matrix is symmetric with a zero diagonal (gen_dist_slave :411, init_ptr :200): cells (j, i), j < i, walk down
      // a column of the upper triangle -- pair index j (2n - j - 1) / 2 + (i - j - 1), its step from j to j + 1 is n - j - 2
      uint64_t k = i - 1;  // (j = 0)
      for (uint64_t j = 0; j < i; j++) {
        if (o > lim && !grow()) return;
        *o++ = '\t';
        o = fmt_fixed10(dist[k], o);
        k += n_ind - j - 2;
      }
      if (o > lim && !grow()) return;
      *o++ = '\t';
      o = fmt_fixed10(0.0, o);
      const double *row = dist + i * (2 * n_ind - i - 1) / 2 - (i + 1);  // row[j] = cell (i, j), j > i
      for (uint64_t j = i + 1; j < n_ind; j++) {
        if (o > lim && !grow()) return;
        *o++ = '\t';
        o = fmt_fixed10(row[j], o);
      }
      *o++ = '\n';
    }
    part_len[t] = (size_t)(o - scratch.buf[t]);
  };
  run_parts(rows);
  if (oom) return NGD_E_NOMEM;
  uint64_t total = (uint64_t)hn;
  std::vector<uint64_t> at(n_parts);
  for (unsigned t = 0; t < n_parts; t++) { at[t] = total; total += part_len[t]; }
  if (out && cap >= total) {
    memcpy(out, head, (size_t)hn);
    run_parts([&](unsigned t) { memcpy(out + at[t], scratch.buf[t], part_len[t]); });  // (13 MB for 1000 individuals: in parallel too)
  }
  return (int64_t)total;
}

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

// The same for every pair at once, in the reference's pair order (ngsDist.cpp:244-245): what a host needs to pack the
// cells it owns for the ONE all-gather of a pair-sharded job.
void ngd_shard_map(uint64_t n_ind, uint32_t shard_world, int32_t *owner) {
  const uint32_t n_t = (uint32_t)((n_ind + 127) / 128);
  const std::vector<uint32_t> own = ngd_tile_owners(n_t, shard_world ? shard_world : 1);
  uint64_t k = 0;
  for (uint64_t i = 0; i < n_ind; i++)
    for (uint64_t j = i + 1; j < n_ind; j++) owner[k++] = (int32_t)own[ngd_tile_id(n_t, i / 128, j / 128)];
}

// ngd_config.single_image = 2 (engine.hip): the symmetric score matrix as a sum of three weighted squares, S = SUM_r d[r] c_r c_r^T
// (Lagrange's reduction; c row-major: c[3 r + g]).  With t_r = c_r . p per site, p1^T S p2 = SUM_r d[r] t_r(p1) t_r(p2):
// ONE image (t) serves both operands of the MFMA kernel and d rides on the per-index weights.  Every step divides by a
// diagonal entry or by twice an off-diagonal one only: for the reference's two matrices (parse_args.cpp:25-27, :134-137:
// entries 0, 0.5, 1) c and d are small dyadic numbers, t is exact for called genotypes and so are the sums.
// NGD_E_INVALID if S is not symmetric or the reconstruction does not give S back to 1e-13 of its largest entry.  Pure host arithmetic.
int ngd_score_congruence(const double *S, double *c, double *d) {
  double A[3][3];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      if (S[3 * a + b] != S[3 * b + a] || !std::isfinite(S[3 * a + b])) return NGD_E_INVALID;
      A[a][b] = S[3 * a + b];
    }
  int n = 0;
  for (int r = 0; r < 9; r++) c[r] = 0;
  for (int r = 0; r < 3; r++) d[r] = 0;
  double scale = 0;
  for (int k = 0; k < 9; k++) scale = std::max(scale, std::fabs(S[k]));
  const double tiny = 1e-14 * scale;  // what an elimination in floating point leaves of an exact zero
  auto deflate = [&](const double *row, double w) {  // A -= w row row^T, the square joins the list
    for (int g = 0; g < 3; g++) c[3 * n + g] = row[g];
    d[n++] = w;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        A[a][b] -= w * row[a] * row[b];
        if (std::fabs(A[a][b]) <= tiny) A[a][b] = 0;
      }
  };
  auto clear = [&](int a) {  // the eliminated variable's row and column are zero by construction
    for (int g = 0; g < 3; g++) A[a][g] = A[g][a] = 0;
  };
  while (n < 3) {
    int pa = -1;
    for (int a = 0; a < 3; a++)
      if (A[a][a] != 0 && (pa < 0 || std::fabs(A[a][a]) > std::fabs(A[pa][pa]))) pa = a;
    if (pa >= 0) {  // a square on the diagonal: A[a][a] (x_a + SUM_b A[a][b] / A[a][a] x_b)^2
      const double piv = A[pa][pa];
      double row[3];
      for (int g = 0; g < 3; g++) row[g] = A[pa][g] / piv;
      deflate(row, piv);
      clear(pa);
      continue;
    }
    int qa = -1, qb = -1;
    for (int a = 0; a < 3; a++)
      for (int b = a + 1; b < 3; b++)
        if (A[a][b] != 0 && (qa < 0 || std::fabs(A[a][b]) > std::fabs(A[qa][qb]))) { qa = a; qb = b; }
    if (qa < 0) break;  // nothing left: rank below 3, the remaining weights stay 0
    if (n > 1) return NGD_E_INVALID;  // (two squares needed)
    // no square, a mixed term: with r_a, r_b the two rows, 2 / beta r_a r_b = 1 / (2 beta) ((r_a + r_b)^2 - (r_a - r_b)^2)
    const double beta = A[qa][qb];
    double plus[3], minus[3];
    for (int g = 0; g < 3; g++) { plus[g] = A[qa][g] + A[qb][g]; minus[g] = A[qa][g] - A[qb][g]; }
    deflate(plus, 1.0 / (2 * beta));
    deflate(minus, -1.0 / (2 * beta));
    clear(qa);
    clear(qb);
  }
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double r = 0;
      for (int k = 0; k < 3; k++) r += d[k] * c[3 * k + a] * c[3 * k + b];
      if (std::fabs(r - S[3 * a + b]) > 1e-13 * scale) return NGD_E_INVALID;
    }
  return NGD_OK;
}

int ngd_finish(const double *sum, const uint64_t *cnt, uint64_t n_pairs, uint64_t tot_sites,
               uint64_t evol_model, double *dist) {
  if (evol_model > 2) return NGD_E_MODEL;  // reference: error("... model not yet supported")
  if (!sum || !cnt || !dist) return NGD_E_INVALID;
  // per-cell and order-free, so threads change nothing but the wall time
  unsigned nt = 1;
  if (n_pairs >= (1u << 15)) nt = std::min(n_pairs >= (1u << 17) ? 16u : 8u, std::max(1u, std::thread::hardware_concurrency()));  // (waking the pool costs ~50 us)
  const bool wide = n_pairs >= (1u << 19) && std::thread::hardware_concurrency() > 16;
  if (wide) nt = std::min(64u, std::thread::hardware_concurrency());
  if (nt <= 1) {
    finish_range(sum, cnt, 0, n_pairs, tot_sites, evol_model, dist);
  } else {
    const unsigned parts = n_pairs >= (1u << 20) ? 4 * nt : nt;  // large jobs: finer than the threads, so that they finish together
    const uint64_t per = (n_pairs + parts - 1) / parts;
    (wide ? host_pool_wide() : host_pool()).run(parts, [&](unsigned k) {
      const uint64_t lo = k * per, hi = std::min(n_pairs, lo + per);
      if (lo < hi) finish_range(sum, cnt, lo, hi, tot_sites, evol_model, dist);
    });
  }
  return NGD_OK;
}

// ngd_finish() over cells that are still ARRIVING (a job's matrices coming off the device chunk by chunk): the pool is
// woken once for the whole job and a share of cells is worked as soon as *landed says it is final, so the tail of
// gen_dist() runs beside the copies instead of one pool wake-up per chunk ([measured, round 6] 8.1e6 cells: 1.3 ms in one
// call, 2.7 ms in 8).  Shares are handed out in ascending order; a thread whose share has not landed yet spins on the counter.
int ngd_finish_stream(const double *sum, const uint64_t *cnt, uint64_t n_pairs, uint64_t tot_sites, uint64_t evol_model,
                      double *dist, const volatile uint64_t *landed) {
  if (evol_model > 2) return NGD_E_MODEL;
  if (!sum || !cnt || !dist || !landed) return NGD_E_INVALID;
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const bool wide = n_pairs >= (1u << 19) && hw > 16;
  const unsigned nt = wide ? std::min(64u, hw) : std::min(16u, hw);
  const unsigned parts = (unsigned)std::min<uint64_t>(std::max<uint64_t>(1, n_pairs / 16384), 8 * nt);
  const uint64_t per = (n_pairs + parts - 1) / parts;
  auto share = [&](unsigned k) {
    const uint64_t lo = k * per, hi = std::min(n_pairs, lo + per);
    if (lo >= hi) return;
    for (unsigned spins = 0; *landed < hi; spins++) {
      if (spins < 4096) __builtin_ia32_pause();
      else std::this_thread::yield();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    finish_range(sum, cnt, lo, hi, tot_sites, evol_model, dist);
  };
  if (nt <= 1 || parts <= 1) {
    for (unsigned k = 0; k < parts; k++) share(k);
  } else {
    (wide ? host_pool_wide() : host_pool()).run(parts, share);
  }
  return NGD_OK;
}

}  // extern "C"

// The same for the engine's own use (engine.hip run_dist: a job's tail inside the call), over n_mat matrices of n_pairs cells:
// the counts per cell (cnt), or -- without --pairwise_del, where every pair of a matrix has the same count, ngsDist.cpp:362 --
// one per matrix (cnt_mat), so that no array of counts has to leave the device.  The division is the reference's either
// way (sum / (double)count).
int ngd_finish_matrices_stream(const double *sum, const uint64_t *cnt, const uint64_t *cnt_mat, uint32_t n_mat, uint64_t n_pairs,
                               uint64_t evol_model, double *dist, const volatile uint64_t *landed) {
  if (evol_model > 2) return NGD_E_MODEL;
  if (!sum || (!cnt && !cnt_mat) || !dist || !landed) return NGD_E_INVALID;
  const uint64_t total = (uint64_t)n_mat * n_pairs;
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const bool wide = total >= (1u << 19) && hw > 16;
  const unsigned nt = wide ? std::min(64u, hw) : std::min(16u, hw);
  // shares of 4096 cells (finish_range's own grain): the cells of the LAST chunk to land are then spread over all the threads
  const unsigned parts = (unsigned)std::min<uint64_t>(std::max<uint64_t>(1, total / 4096), 1u << 20);
  const uint64_t per = (total + parts - 1) / parts;
  auto share = [&](unsigned k) {
    const uint64_t lo = (uint64_t)k * per, hi = std::min(total, lo + per);
    if (lo >= hi) return;
    for (unsigned spins = 0; *landed < hi; spins++) {  // (the engine may be recomputing noted pairs meanwhile: then sleep)
      if (spins < 4096) __builtin_ia32_pause();  // (~0.1 ms; the threads' CPU time counts against the process's quota)
      else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (cnt) {
      finish_range(sum, cnt, lo, hi, 0, evol_model, dist);
      return;
    }
    for (uint64_t a = lo; a < hi;) {  // matrix by matrix: one count each
      const uint64_t m = a / n_pairs, b = std::min(hi, (m + 1) * n_pairs);
      if (cnt_mat[m]) {
        finish_range(sum, nullptr, a, b, cnt_mat[m], evol_model, dist);
      } else {  // (a matrix that visits no site: 0 / 0 as the per-cell form has it)
        const uint64_t zero = 0;
        for (uint64_t c = a; c < b; c++) finish_range(sum + c, &zero, 0, 1, 0, evol_model, dist + c);
      }
      a = b;
    }
  };
  if (nt <= 1 || parts <= 1) {
    for (unsigned k = 0; k < parts; k++) share(k);
  } else {
    // the shares are dealt by a counter of their own, in ascending order (the pool's mutex per part would cost more than a share)
    std::atomic<unsigned> next{0};
    (wide ? host_pool_wide() : host_pool()).run(std::min(nt, parts), [&](unsigned) {
      for (unsigned k; (k = next.fetch_add(1, std::memory_order_relaxed)) < parts;) share(k);
    });
  }
  return NGD_OK;
}
