// engine.hip -- the C ABI of include/ngsdist_amd.h: device-resident data set,
// shard bookkeeping and the per-replicate launch sequence that stands in for
// the reference's `for i1<i2: threadpool_add(gen_dist_slave)` block
// (ngsDist.cpp:244-269).  There is no CPU fallback anywhere in this file: if
// HIP is unusable every entry point fails with an error code.
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ngd_internal.h"
#include "ngd_shard.h"

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

// A failed HIP call also leaves its code behind as the thread's "last error"; it is reported HERE, once, and cleared, so
// that the hipGetLastError() after a later, unrelated kernel launch does not report it a second time (found by
// tests/test_gpu_abi_misuse.py: an engine too large for the device poisoned the next engine's first launch).
#define HIPCHK(call)                                                                       \
  do {                                                                                     \
    hipError_t _e = (call);                                                                \
    if (_e != hipSuccess) {                                                                \
      (void)hipGetLastError();                                                             \
      return fail(_e == hipErrorOutOfMemory ? NGD_E_NOMEM : NGD_E_HIP,                     \
                  std::string(#call) + ": " + hipGetErrorString(_e));                      \
    }                                                                                      \
  } while (0)

struct ngd_engine {
  ngd_config cfg{};
  ngd_geom g{};
  ngd_score sc{};
  int device = 0;
  int kernel = 0;  // resolved NGD_KERNEL_*
  hipStream_t st = nullptr;
  hipEvent_t ev[5] = {};
  // resident data set
  double *PA = nullptr, *QB = nullptr, *PI = nullptr;
  // ngd_config.single_image (MFMA kernel): QB is not resident; a launch forms it for a range of k-groups at a time
  bool single_image = false;    // (ngd_config.single_image = 1: q is formed range by range)
  bool congruent = false;       // ngd_config.single_image = 2: the image holds t (sc.c, sc.d), read for both operands
  double *d_wD = nullptr;       // ... and these are the weights of a plain pass: sc.d[k % 3] per contraction index
  // ... and, for the reference's matrices (sc.fix), the fix-up pass of the pairs its arithmetic cannot hold to 1e-9
  // relative (fixup.hip): SM[site][individual] = min(p0, p2) beside the image, the pairs a reduction noted, scratch
  double *SM = nullptr;
  unsigned long long *d_fixlist = nullptr;
  uint32_t *d_fixcount = nullptr, *d_fixseen = nullptr, *h_fixcount = nullptr;
  double *d_fixparts = nullptr, *d_fixthr = nullptr;
  uint64_t cap_fixthr = 0;
  ngd_fix_tile *d_fixtiles = nullptr;  // 16 x 16 tiles of pairs that hold several noted pairs (fixup_pass)
  double *d_fixtparts = nullptr;       // ... and their per-slice partial sums
  uint64_t cap_fixtiles = 0, cap_fixtparts = 0;
  double *fix_p = nullptr, *fix_q = nullptr, *d_fixnew = nullptr;  // the fix-up pass as a whole two-operand pass (fixup_by_pass)
  uint64_t cap_fix_p = 0, cap_fix_q = 0, cap_fixnew = 0;
  ngd_fixup_info fix_info{};
  std::vector<ngd_tile> h_tiles16;  // host copy of the owned 16 x 16 tiles that hold a pair (the fix-up pass's "every pair")
  uint32_t fix_cap = 0;  // pairs the reductions can note for the fix-up pass (ngd_internal.h NGD_FIX_LIST): the capacity of d_fixlist
  uint64_t opt_fix_work = 0;  // NGD_OPT_FIXUP_WORK: the pass's budget in pair-sites (0 = none: every noted pair is recomputed)
  double *QB_res = nullptr;     // ... except its first qb_res_kg k-groups (ngd_config.second_image_mib), formed at ngd_commit()
  uint64_t qb_res_kg = 0;
  double *qb_chunk = nullptr;
  uint64_t qb_chunk_kg = 0;     // k-groups a range may span (NGD_OPT_SINGLE_IMAGE_BYTES)
  uint64_t qb_chunk_elems = 0;  // capacity of the scratch
  unsigned long long *mask = nullptr, *planes = nullptr;
  // bootstrap
  uint32_t *d_mult = nullptr, *d_ws = nullptr;
  double *d_wk = nullptr;  // multiplicity per contraction index k, as a double (MFMA kernel)
  uint32_t *d_kgl = nullptr, *d_kgcnt = nullptr;  // k-groups a replicate visits (list + compaction scratch)
  uint32_t *h_mult = nullptr;                      // pinned: multiplicities counted from block maps
  uint64_t cap_h_mult = 0;
  uint64_t cap_blocks = 0;
  // shard
  ngd_tile *d_tiles = nullptr, *d_tiles16 = nullptr, *d_tiles64 = nullptr;
  uint32_t n_tiles = 0, n_tiles16 = 0, n_tiles64 = 0;
  int em_shape = 0;  // accum_em_table.hip: workgroup shape
  unsigned long long *d_emcnt = nullptr;  // [4] work counters of the table-driven EM kernel + its clock counters
  unsigned long long em_counts[2] = {0, 0};  // ... of the last run
  // [2] MFMA kernel: shader-cycle / constant-rate counter deltas of one wavefront.  Pinned HOST memory mapped into the
  // device's address space: the wavefront's two stores cross PCIe, and reading them after the stream has been waited for
  // is a plain load (a 16-byte hipMemcpy per pass was 10 us of a 350 us job at cfg 2)
  unsigned long long *d_clk = nullptr, *h_clk = nullptr;  // (the device's and the host's pointer to it)
  double clk_mhz = 0;                   // shader clock of the last accumulation launch (0: not sampled)
  double wall_khz = 100000.0;           // rate of the constant counter (hipDeviceAttributeWallClockRate)
  // MFMA kernel: per-wavefront 64x64 jobs, 4 per workgroup; "tri" = blocks on the diagonal
  ngd_job *d_jobs = nullptr;
  uint32_t n_wg = 0;
  uint32_t wg_waves = 4;  // wavefronts (jobs) per workgroup of the MFMA kernel
  int exact_shapes = 0;  // small n_ind: one code path per block shape (accum_mfma.hip EXACT): 1 = blocks of 4 x 4 tiles, 2 = 2 x 4
  bool tri_diag = false;  // full 4 x 4 blocks (exact_shapes == 0) whose DIAGONAL blocks leave their lower triangle out
  uint64_t *d_pairs = nullptr;
  uint64_t n_owned_pairs = 0;
  // scratch + results
  double *slab = nullptr;
  uint32_t n_ks = 0;
  uint64_t per_slice = 0;
  double *d_sum = nullptr;
  unsigned long long *d_cnt = nullptr;
  // bootstrap by per-block partial sums (valid while boot_B != 0)
  double *slab_boot = nullptr;
  uint64_t boot_B = 0, boot_blocks = 0, boot_per_slice = 0, slab_boot_elems = 0;
  uint32_t boot_nks = 0, boot_sub = 0;
  // a large slab costs ~12 ms per GB to allocate: until the passes it would have saved add up to that, calls are
  // served without it (rent_ms = their estimated cost so far, for the geometry rent_B / rent_blocks)
  double rent_ms = 0;
  uint64_t rent_B = 0, rent_blocks = 0;
  double *d_wslice = nullptr;    // 0/1 weights per slice, bootstrap blocks that are not whole k-groups
  uint64_t cap_wslice = 0;
  uint32_t *cnt_boot = nullptr;  // per-block valid-site counts [n_blocks][n_pad][n_pad] (--pairwise_del)
  uint64_t cnt_B = 0, cnt_blocks = 0, cnt_boot_elems = 0;
  // per-call bootstrap weights (slice-major doubles / block-major uint32) and per-replicate site totals
  double *d_W = nullptr;
  uint32_t *d_M = nullptr;
  unsigned long long *d_drawn = nullptr;
  uint64_t cap_W = 0, cap_M = 0, cap_drawn = 0;
  // batch results for the host-pointer entry points
  double *d_bsum = nullptr;
  unsigned long long *d_bcnt = nullptr;
  uint64_t cap_batch = 0;
  uint32_t n_batch_valid = 0;  // matrices of the last batch / job call, still in d_bsum / d_bcnt (ngd_fetch_matrix)
  // ngd_run_job_dist / ngd_run_mult_batch_dist: the matrices of d_bsum / d_bcnt leave the device in chunks on a stream
  // of their own -- in the per-block-partials plan a group of replicates as soon as its reduction is over, beside the
  // reductions of the later groups -- into pinned memory of the engine's, and the tail of gen_dist() (host_util.cpp) works
  // the cells of a chunk as soon as it has landed
  struct OutStream {
    bool on = false, pdel = false;
    hipStream_t st = nullptr, st2 = nullptr;  // chunks alternate between two copy streams
    uint32_t n_chunk_seq = 0, n_landed = 0;    // chunks queued / declared landed so far in this call
    double *h_sum = nullptr;
    uint64_t *h_cnt = nullptr;
    uint64_t cap_sum = 0, cap_cnt = 0;       // cells
    uint32_t n_mat = 0, queued = 0;          // matrices of this call; matrices [0, queued) have their copies on st
    std::vector<hipEvent_t> pool;            // events, made on demand and kept
    uint32_t n_used = 0;
    std::vector<std::pair<hipEvent_t, uint32_t>> chunks;  // (the copy's event, matrices in host memory once it has happened)
    std::vector<uint64_t> cnt_mat;           // no --pairwise_del: a matrix's count (the sites it visits), ngsDist.cpp:362
    uint64_t evol_model = 0, tot_sites = 0;  // tot_sites > 0: the count of every cell (ngsDist.cpp:372-373)
    double *dist = nullptr;
    volatile uint64_t landed = 0;            // cells of h_sum (h_cnt) that are final: raised by the calling thread
    std::thread finisher;
    int finisher_rc = 0;
    double t0 = 0, t_call = 0;  // NGD_TRACE_OUT
  } out;
  double *staging = nullptr;
  uint64_t staging_sites = 0;
  // raw-input pipeline: a ring of pinned host buffers, each with its device buffer; the copies run on a stream of
  // their own (the copy engine never waits for a preparation kernel), K0 follows each on the engine's stream
  static constexpr int RING = 8;
  uint64_t opt_stage_piece_mib = 32, opt_stage_ring = 6;  // NGD_OPT_STAGE_PIECE_MIB, NGD_OPT_STAGE_RING
  // NGD_OPT_EAGER_FULL: the plain full-data pass starts DURING a staged load -- whenever enough leading slices of the site
  // axis have all their sites prepared, they are accumulated on a low-priority stream of their own beside the copies and
  // preparation kernels of the pieces still arriving; the first ngd_run() then launches what is left and reduces
  bool opt_eager = false;
  hipStream_t st_eager = nullptr;
  hipEvent_t ev_eager = nullptr;
  uint64_t stage_prefix = 0;   // sites [0, stage_prefix) have been submitted, in order
  bool stage_in_order = true;
  uint32_t eager_slices = 0;   // slices [0, eager_slices) of the plain pass have been launched on st_eager
  bool eager_valid = false;
  double *pin[RING] = {}, *draw[RING] = {};
  hipEvent_t pin_free[RING] = {};  // the copy out of pin[b] is done: the caller may fill it again
  hipEvent_t k0_done[RING] = {};   // K0 has read draw[b]: the next copy may overwrite it
  hipStream_t st_copy[2] = {nullptr, nullptr};
  uint64_t n_staged = 0;
  int ring_slots = 0;
  std::thread ring_reaper;  // gives the ring back after ngd_commit, beside whatever the caller does next
  // The ring GROWS: its first buffer is made by the first ngd_stage_acquire, the others by a thread beside the load
  // (hipHostMalloc: 5 ms per 32-MiB buffer); the load turns through the buffers that exist (ring_ready of them)
  std::thread ring_maker;
  std::atomic<int> ring_ready{0};
  std::atomic<bool> ring_stop{false};  // the load is over: no more buffers are needed
  int ring_maker_rc = 0;
  int pin_cur = 0, pin_lent = -1;
  uint64_t pin_sites = 0;
  int *d_nan = nullptr;
  bool committed = false;
  uint64_t dev_bytes = 0;
  // Images and slabs of a GiB and more: an address range reserved at once, its physical memory created, mapped and
  // zeroed 256 MiB at a time by a thread of the engine's own (dev_alloc_pieces, piece_worker) -- the staged load starts at
  // once and waits, piece by piece, only for the part of an image it is about to write (piece_wait_sites).
  enum PieceKind { PIECE_FRAG, PIECE_SITE_MAJOR, PIECE_WHOLE };  // how far into the range a site reaches
  struct PieceRange {
    void *va = nullptr;
    size_t size = 0, ready = 0, n_mapped = 0;  // ready: bytes from the start that are mapped (and zeroed), under piece_mu
    bool zero = false;
    PieceKind kind = PIECE_WHOLE;
    uint64_t bytes_per_site = 0;  // PIECE_SITE_MAJOR
    std::vector<hipMemGenericAllocationHandle_t> hs;
  };
  std::vector<std::unique_ptr<PieceRange>> piece_ranges;
  std::thread piece_thread;
  std::mutex piece_mu;
  std::condition_variable piece_cv;
  bool piece_done = true;  // nothing left to map (or the worker gave up: piece_rc)
  int piece_rc = 0;
  std::string piece_err;
  ngd_timing timing{};
  // plan options (ngd_set_option)
  uint64_t opt_boot_partials = 1, opt_boot_max_bytes = 0, opt_boot_wg = 4096, opt_boot_unaligned = 1, opt_em_batch = 1;
  uint64_t opt_em_spill = 1, opt_em_spill_bytes = 0;
  // the EM batch pass's result planes did not fit this device at this many elements: a request as large is not tried
  // again (0: nothing has failed) -- until ngd_drop_caches(), or until a smaller request (fewer matrices per pass) comes
  uint64_t em_batch_nofit_elems = 0;
  // EM bootstrap by spilled terms + one MFMA contraction (contract_mfma.hip): running sums and per-chunk NaN flags
  double *d_D = nullptr;
  unsigned long long *d_nanflag = nullptr;
  uint64_t cap_D = 0, cap_nanflag = 0;
  // ... its pair slots: groups of 16 consecutive columns of one row of a 64 x 64 tile, dealt to the groups that hold a
  // pair only; d_rowpg[tile * 64 + row] + g = slot group of the row's column group g (a signed 32-bit number: the first live
  // group's slot group minus that group's index), n_pg_spill = their number (+ padding to 4)
  uint32_t *d_rowpg = nullptr;
  uint32_t n_pg_spill = 0, n_pg_live = 0;
  std::vector<hipEvent_t> ev_spill;  // per chunk: before the weights, the EM pass, the sanitiser, the contraction; + one at the end
  ngd_spill_timing spill_timing{};
};

template <typename T>
static int dev_alloc(ngd_engine *e, T **p, uint64_t count, bool zero) {
  *p = nullptr;
  if (!count) return NGD_OK;
  // NGD_TRACE_ALLOC=1: what every allocation of 64 MiB and more costs (the driver clears memory other processes have used
  // as it hands it out: seconds for tens of GB on a device that has just been busy, DESIGN.md section 3 "K0")
  static const bool trace = getenv("NGD_TRACE_ALLOC") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  HIPCHK(hipMalloc((void **)p, count * sizeof(T)));
  if (trace && count * sizeof(T) >= (64u << 20))
    fprintf(stderr, "> alloc: hipMalloc of %.2f GB took %.3f s\n", (double)(count * sizeof(T)) / 1e9,
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  e->dev_bytes += count * sizeof(T);
  if (zero) HIPCHK(hipMemsetAsync(*p, 0, count * sizeof(T), e->st));
  return NGD_OK;
}

// The operand images and slabs (a GiB and more): an address range reserved at once, physical pieces of 256 MiB created,
// mapped and zeroed behind it by a thread of the engine's own, in the order a load needs them.
// [measured, round 6, tools/alloc_cost.hip, rocprofv3 --hip-trace of the C++ host, gpurun_out/r6/e2e_4.jsonl] On a box whose
// device memory has been used before -- every box after its first few jobs -- the driver clears memory as it hands it out:
// hipMalloc of cfg 3's 24.6 GB image takes 0.3 ms on pristine memory and 1.0-1.2 s otherwise (hiptrace_cfg3_3: 984 ms in
// ONE hipMalloc; ~30 GB/s), while the link moves the file at 57 GB/s.  One allocation up front therefore cost more than the
// whole load; piece by piece, beside the load, it costs max(clearing, load).  The pieces read at the same 5.3 TB/s as one
// hipMalloc (tools/alloc_cost.hip) and K1m runs at the same 45.6 ms on them (gpurun_out/r6/bench_cfg3_vmm.json).
// Anything the piecewise calls refuse up front falls back to hipMalloc; a failure later (out of memory) is reported by
// the first call that needs the memory (ngd_stage_submit / ngd_upload_* / ngd_commit: NGD_E_NOMEM).
// (pieces of ONE size per range: hipMemSetAccess refuses a shorter last piece -- [measured] 1 GiB + 512 MiB: invalid
// argument; 15 x 256 MiB: fine -- so a range is rounded up to whole pieces, at most 256 MiB more than asked for)
static const size_t kPiece = (size_t)256 << 20;

static void release_pieces(ngd_engine::PieceRange &r) {
  for (size_t c = 0; c < r.n_mapped; c++) (void)hipMemUnmap((char *)r.va + c * kPiece, std::min(kPiece, r.size - c * kPiece));
  for (auto &h : r.hs) (void)hipMemRelease(h);
  if (r.va) (void)hipMemAddressFree(r.va, r.size);
  r.hs.clear();
  r.va = nullptr;
  r.n_mapped = 0;
}

template <typename T>
static int dev_alloc_pieces(ngd_engine *e, T **p, uint64_t count, bool zero, ngd_engine::PieceKind kind = ngd_engine::PIECE_WHOLE,
                            uint64_t bytes_per_site = 0) {
  *p = nullptr;
  const uint64_t bytes = count * sizeof(T);
  if (bytes < ((uint64_t)512 << 20)) return dev_alloc(e, p, count, zero);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = e->device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) {
    (void)hipGetLastError();
    return dev_alloc(e, p, count, zero);
  }
  if (kPiece % gran) return dev_alloc(e, p, count, zero);
  std::unique_ptr<ngd_engine::PieceRange> r(new ngd_engine::PieceRange());
  r->size = (size_t)((bytes + kPiece - 1) / kPiece * kPiece);
  r->zero = zero;
  r->kind = kind;
  r->bytes_per_site = bytes_per_site;
  if (hipMemAddressReserve(&r->va, r->size, 0, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    return dev_alloc(e, p, count, zero);
  }
  *p = (T *)r->va;
  e->piece_ranges.push_back(std::move(r));
  e->dev_bytes += bytes;
  return NGD_OK;
}

// The worker: always the piece of the range that is furthest behind (relative to its size), so that the images of a data
// set grow together along the site axis; ranges nothing writes during a load (PIECE_WHOLE: slabs) after them.
static void piece_worker(ngd_engine *e) {
  auto give_up = [&](const char *what, hipError_t err) {
    std::lock_guard<std::mutex> lk(e->piece_mu);
    e->piece_rc = err == hipErrorOutOfMemory ? NGD_E_NOMEM : NGD_E_HIP;
    e->piece_err = std::string("device memory, piece by piece: ") + what + ": " + hipGetErrorString(err);
    e->piece_done = true;
    e->piece_cv.notify_all();
  };
  hipError_t err = hipSetDevice(e->device);
  if (err != hipSuccess) return give_up("hipSetDevice", err);
  hipStream_t sa = nullptr;
  if ((err = hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)) != hipSuccess) return give_up("hipStreamCreate", err);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = e->device;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  // test hook (NGD_ENABLE_TEST_HOOKS=1): the NGD_TEST_FAIL_PIECE-th piece "runs out of memory" -- the path a real failure takes
  long fail_at = -1, n_made = 0;
  if (const char *hook = getenv("NGD_ENABLE_TEST_HOOKS"))
    if (!strcmp(hook, "1"))
      if (const char *k = getenv("NGD_TEST_FAIL_PIECE")) fail_at = atol(k);
  for (;;) {
    ngd_engine::PieceRange *r = nullptr;
    for (int whole = 0; whole < 2 && !r; whole++) {
      double best = 2.0;
      for (auto &q : e->piece_ranges) {
        if ((q->kind == ngd_engine::PIECE_WHOLE) != (whole == 1) || q->n_mapped * kPiece >= q->size) continue;
        const double f = (double)(q->n_mapped * kPiece) / (double)q->size;
        if (f < best) { best = f; r = q.get(); }
      }
    }
    if (!r) break;
    const size_t off = r->n_mapped * kPiece, len = std::min(kPiece, r->size - off);
    hipMemGenericAllocationHandle_t h;
    if (fail_at >= 0 && n_made++ == fail_at) { hipStreamDestroy(sa); return give_up("hipMemCreate (test hook)", hipErrorOutOfMemory); }
    if ((err = hipMemCreate(&h, len, &prop, 0)) != hipSuccess) { hipStreamDestroy(sa); return give_up("hipMemCreate", err); }
    r->hs.push_back(h);
    if ((err = hipMemMap((char *)r->va + off, len, 0, h, 0)) != hipSuccess) { hipStreamDestroy(sa); return give_up("hipMemMap", err); }
    r->n_mapped++;
    if ((err = hipMemSetAccess((char *)r->va + off, len, &acc, 1)) != hipSuccess) { hipStreamDestroy(sa); return give_up("hipMemSetAccess", err); }
    if (r->zero) {
      if ((err = hipMemsetAsync((char *)r->va + off, 0, len, sa)) != hipSuccess || (err = hipStreamSynchronize(sa)) != hipSuccess) {
        hipStreamDestroy(sa);
        return give_up("zero fill", err);
      }
    }
    std::lock_guard<std::mutex> lk(e->piece_mu);
    r->ready = off + len;
    e->piece_cv.notify_all();
  }
  hipStreamDestroy(sa);
  std::lock_guard<std::mutex> lk(e->piece_mu);
  e->piece_done = true;
  e->piece_cv.notify_all();
}

// (a reserved address range costs nothing: what the ranges will take is checked against the device's free memory HERE, so
// that a data set that cannot fit is refused by ngd_create -- NGD_E_NOMEM -- and not by the first upload)
static int piece_start(ngd_engine *e) {
  if (e->piece_ranges.empty()) return NGD_OK;
  size_t free_b = 0, total_b = 0, want = 0;
  for (auto &q : e->piece_ranges) want += q->size;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b)
    return fail(NGD_E_NOMEM, "ngd_create: the images and slabs of this data set exceed the device's free memory");
  e->piece_done = false;
  e->piece_thread = std::thread(piece_worker, e);
  return NGD_OK;
}

// every piece of every range is there (or the worker has failed: its error)
static int piece_join(ngd_engine *e) {
  if (e->piece_thread.joinable()) e->piece_thread.join();
  if (e->piece_rc) return fail(e->piece_rc, e->piece_err.c_str());
  return NGD_OK;
}

// ... or only what the sites [0, s_end) of the data set reach in the ranges a load writes
static int piece_wait_sites(ngd_engine *e, uint64_t s_end) {
  if (e->piece_ranges.empty()) return NGD_OK;
  std::unique_lock<std::mutex> lk(e->piece_mu);
  for (auto &q : e->piece_ranges) {
    size_t need = q->size;
    if (s_end < e->g.n_sites) {
      if (q->kind == ngd_engine::PIECE_FRAG) need = std::min<size_t>(q->size, ((3 * s_end + 3) / 4 + 1) * (size_t)e->g.n_ig * 512);
      else if (q->kind == ngd_engine::PIECE_SITE_MAJOR) need = std::min<size_t>(q->size, (size_t)(s_end * q->bytes_per_site));
      else if (q->va == (void *)e->slab) continue;  // (nothing of a load goes there)
    } else if (q->va == (void *)e->slab) {
      continue;
    }
    e->piece_cv.wait(lk, [&] { return q->ready >= need || e->piece_done; });
    if (q->ready < need && e->piece_rc) return fail(e->piece_rc, e->piece_err.c_str());  // (what IS mapped serves its sites)
  }
  return NGD_OK;
}

static bool in_pieces(const ngd_engine *e, const void *p) {
  for (auto &r : e->piece_ranges)
    if (r->va == p) return true;
  return false;
}

// grow-only device scratch
template <typename T>
static int ensure_cap(ngd_engine *e, T **p, uint64_t *cap, uint64_t need) {
  if (need <= *cap) return NGD_OK;
  if (*p) { HIPCHK(hipFree(*p)); e->dev_bytes -= *cap * sizeof(T); }
  *p = nullptr; *cap = 0;
  int rc = dev_alloc(e, p, need, false);
  if (rc) return rc;
  *cap = need;
  return NGD_OK;
}

// Single-image engines: k-groups of the second operand image formed at a time by default (4 GB of them)
static uint64_t single_image_span(const ngd_geom &g) {
  return std::max<uint64_t>(1, std::min<uint64_t>(g.n_kg, (4ull << 30) / ((uint64_t)g.n_ig * 64 * 8)));
}

// Single-image engines, a whole pass in ranges: the piece of a range one slice takes (k-groups: whole pipeline trips, and
// long enough to carry a block's 128 KB of running sums in and out) so that a range is about `span` k-groups.
static uint64_t qb_piece(uint64_t kg_lim, uint32_t n_ks, uint64_t span, uint64_t *n_ranges) {
  uint64_t r = std::max<uint64_t>(1, (kg_lim + span - 1) / span);
  const uint64_t piece = std::max<uint64_t>(64, ((kg_lim + r * n_ks - 1) / (r * n_ks) + 3) / 4 * 4);
  *n_ranges = std::max<uint64_t>(1, (kg_lim + piece * n_ks - 1) / (piece * n_ks));
  return piece;
}

extern "C" {

const char *ngd_last_error(void) { return g_err.c_str(); }
int ngd_abi_version(void) { return NGD_ABI_VERSION; }

int ngd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ngd_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(NGD_E_NODEVICE, "ngd_device_memory: no HIP device");
  int cur = 0;
  HIPCHK(hipGetDevice(&cur));
  if (device < 0) device = cur;
  if (device >= n) return fail(NGD_E_NODEVICE, "ngd_device_memory: device ordinal out of range");
  HIPCHK(hipSetDevice(device));
  size_t f = 0, t = 0;
  HIPCHK(hipMemGetInfo(&f, &t));
  HIPCHK(hipSetDevice(cur));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return NGD_OK;
}

uint64_t ngd_n_pairs(uint64_t n_ind) { return n_ind * (n_ind - 1) / 2; }
uint64_t ngd_pair_index(uint64_t n_ind, uint64_t i1, uint64_t i2) { return ngd_pair_idx(n_ind, i1, i2); }
uint64_t ngd_device_bytes(const ngd_engine *e) { return e ? e->dev_bytes : 0; }

static void stage_reap(ngd_engine *e);
static void pin_release(double *p);

void ngd_destroy(ngd_engine *e) {
  if (!e) return;
  hipSetDevice(e->device);
  if (e->piece_thread.joinable()) e->piece_thread.join();
  if (e->st) hipStreamSynchronize(e->st);
  if (e->st_eager) hipStreamSynchronize(e->st_eager);  // (slices started beside a load and never asked for)
  e->ring_stop = true;
  if (e->ring_maker.joinable()) e->ring_maker.join();
  stage_reap(e);
  void *ptrs[] = {e->PA, e->QB, e->QB_res, e->qb_chunk, e->PI, e->mask, e->planes, e->d_mult, e->d_ws, e->d_wk, e->d_wD, e->d_kgl, e->d_kgcnt,
                  e->d_tiles, e->d_tiles16, e->d_tiles64, e->d_pairs, e->d_jobs, e->slab, e->d_sum, e->d_cnt, e->staging, e->slab_boot,
                  e->cnt_boot, e->d_W, e->d_M, e->d_drawn, e->d_bsum, e->d_bcnt, e->d_wslice, e->d_emcnt, e->d_D, e->d_nanflag,
                  e->d_rowpg, e->SM, e->d_fixlist, e->d_fixcount, e->d_fixseen, e->d_fixparts, e->d_fixthr, e->d_fixtiles, e->d_fixtparts, e->fix_p, e->fix_q, e->d_fixnew};
  for (void *p : ptrs)
    if (p && !in_pieces(e, p)) hipFree(p);
  for (auto &r : e->piece_ranges) release_pieces(*r);
  for (int b = 0; b < ngd_engine::RING; b++) {
    pin_release(e->pin[b]);
    if (e->draw[b]) hipFree(e->draw[b]);
    if (e->pin_free[b]) hipEventDestroy(e->pin_free[b]);
    if (e->k0_done[b]) hipEventDestroy(e->k0_done[b]);
  }
  for (int c = 0; c < 2; c++)
    if (e->st_copy[c]) hipStreamDestroy(e->st_copy[c]);
  if (e->st_eager) hipStreamDestroy(e->st_eager);
  if (e->ev_eager) hipEventDestroy(e->ev_eager);
  if (e->out.st) { hipStreamSynchronize(e->out.st); hipStreamDestroy(e->out.st); }
  if (e->out.st2) { hipStreamSynchronize(e->out.st2); hipStreamDestroy(e->out.st2); }
  for (hipEvent_t v : e->out.pool) hipEventDestroy(v);
  if (e->out.h_sum) hipHostFree(e->out.h_sum);
  if (e->out.h_cnt) hipHostFree(e->out.h_cnt);
  if (e->h_clk) hipHostFree(e->h_clk);
  if (e->h_fixcount) hipHostFree(e->h_fixcount);
  if (e->d_nan) hipFree(e->d_nan);
  if (e->h_mult) hipHostFree(e->h_mult);
  for (auto &v : e->ev)
    if (v) hipEventDestroy(v);
  for (auto &v : e->ev_spill)
    if (v) hipEventDestroy(v);
  if (e->st) hipStreamDestroy(e->st);
  delete e;
}

int ngd_create(const ngd_config *cfg, ngd_engine **out) {
  if (!cfg || !out) return fail(NGD_E_INVALID, "ngd_create: null argument");
  *out = nullptr;
  if (cfg->n_ind < 2) return fail(NGD_E_INVALID, "ngd_create: need at least 2 individuals");
  if (cfg->n_sites < 1) return fail(NGD_E_INVALID, "ngd_create: need at least 1 site");
  // tile lists index groups of 16 individuals with 16 bits; what bounds n_ind in practice is device memory (two
  // n_pairs-long result arrays + one n_pad x n_pad plane per slice), checked below before any list is built
  if ((cfg->n_ind + 127) / 128 * 8 > 65535) return fail(NGD_E_INVALID, "ngd_create: n_ind above 1 048 448 (16-bit tile indices)");
  if (cfg->single_image > 3) return fail(NGD_E_INVALID, "ngd_create: single_image is 0 (auto), 1, 2 or 3 (two images)");
  if (cfg->second_image_mib && cfg->single_image != 1)
    return fail(NGD_E_INVALID, "ngd_create: second_image_mib belongs to single_image = 1 engines");
  if (cfg->exact_shapes > 7)
    return fail(NGD_E_INVALID, "ngd_create: exact_shapes must be 0 (auto), 1 (never), 2 (blocks of 4 x 4 tiles), 3 (2 x 4), 4 "
                               "(4 x 4, a slice's jobs in one workgroup), 5 (2 x 4, one workgroup), 6 (5 with operands "
                               "through LDS) or 7 (full blocks, triangular on the diagonal)");
  if (cfg->variant > 4) return fail(NGD_E_INVALID, "ngd_create: no such kernel variant");
  const uint32_t world = cfg->shard_world ? cfg->shard_world : 1;
  if (cfg->shard_rank >= world) return fail(NGD_E_INVALID, "ngd_create: shard_rank >= shard_world");

  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1)
    return fail(NGD_E_NODEVICE, "ngd_create: no HIP device (this engine has no CPU path)");
  int dev = cfg->device;
  if (dev < 0) HIPCHK(hipGetDevice(&dev));
  if (dev >= n_dev) return fail(NGD_E_NODEVICE, "ngd_create: device ordinal out of range");
  HIPCHK(hipSetDevice(dev));
  int kernel = cfg->kernel;
  if (cfg->indep_geno) {
    if (kernel == NGD_KERNEL_AUTO) kernel = NGD_KERNEL_MFMA;
    if (kernel != NGD_KERNEL_MFMA && kernel != NGD_KERNEL_STREAM)
      return fail(NGD_E_INVALID, "ngd_create: kernel does not serve --indep_geno");
  } else {
    // up to 32 individuals the whole job is three 16 x 16 tiles of the per-pair kernel, against one 64 x 64 tile of the
    // table kernel that is 7-25 % occupied ([measured] 300 000 sites: n_ind = 24: 2.5 ms vs 4.5 ms; 48: 5.6 vs 5.5; 64: 9.3
    // vs 6.0; 200: 86 vs 51; 400: 306 vs 151)
    if (kernel == NGD_KERNEL_AUTO) kernel = cfg->n_ind <= 32 ? NGD_KERNEL_EM_FAST : NGD_KERNEL_EM_TABLE;
    if (kernel != NGD_KERNEL_EM_FAST && kernel != NGD_KERNEL_EM_FAITHFUL && kernel != NGD_KERNEL_EM_TABLE)
      return fail(NGD_E_INVALID, "ngd_create: kernel does not serve the EM path");
  }

  {  // before any list is built: the two result arrays + the fewest slab planes this kernel works with must fit at all
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const uint64_t n_pad = (cfg->n_ind + 127) / 128 * 128;
    // tens of thousands of individuals: the MFMA kernel's 8 planes of n_pad^2 doubles (its XCD deal wants 8 slices) no
    // longer fit beside the results -- `auto` then means the streaming kernel, which writes the results directly
    if (cfg->kernel == NGD_KERNEL_AUTO && kernel == NGD_KERNEL_MFMA &&
        ngd_n_pairs(cfg->n_ind) * 16 + 8 * n_pad * n_pad * 8 > total_b)
      kernel = NGD_KERNEL_STREAM;
    const uint64_t planes = kernel == NGD_KERNEL_MFMA ? 8 : kernel == NGD_KERNEL_STREAM ? 0 : 1;
    if (ngd_n_pairs(cfg->n_ind) * 16 + planes * n_pad * n_pad * 8 > total_b)
      return fail(NGD_E_NOMEM, "ngd_create: the result arrays and slabs of this many individuals exceed the device's memory");
  }
  ngd_engine *e = new (std::nothrow) ngd_engine();
  if (!e) return fail(NGD_E_NOMEM, "ngd_create: host allocation failed");
  e->cfg = *cfg;
  e->cfg.shard_world = world;
  e->device = dev;
  e->kernel = kernel;
  memcpy(e->sc.v, cfg->score, sizeof(e->sc.v));
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) e->wall_khz = khz;
  }

  ngd_geom &g = e->g;
  g.n_ind = cfg->n_ind;
  g.n_sites = cfg->n_sites;
  g.n_sites_pad = (cfg->n_sites + 15) / 16 * 16;  // -> n_kg is a multiple of 12
  g.n_kg = 3 * g.n_sites_pad / 4;
  g.n_t = (uint32_t)((cfg->n_ind + NGD_TILE - 1) / NGD_TILE);
  g.n_pad = g.n_t * NGD_TILE;
  g.n_ig = g.n_pad / NGD_IG;
  g.n_words = (uint32_t)((cfg->n_sites + 63) / 64);

  int rc = NGD_OK;
  auto bail = [&](int code) {
    std::string keep = g_err;
    ngd_destroy(e);
    (void)hipGetLastError();  // reported through `code`; not again by the next launch's check
    g_err = keep;
    return code;
  };
  if (hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(NGD_E_HIP, "ngd_create: hipStreamCreate failed"));
  for (auto &v : e->ev)
    if (hipEventCreate(&v) != hipSuccess) return bail(fail(NGD_E_HIP, "ngd_create: hipEventCreate failed"));

  // ---- shard: upper-triangular 128-tiles dealt by cost over ranks (ngd_shard.h) ----
  std::vector<ngd_tile> tiles, tiles16, tiles64;
  std::vector<uint64_t> pairs;
  const std::vector<uint32_t> owner = ngd_tile_owners(g.n_t, world);
  uint32_t tid = 0;
  for (uint32_t ti = 0; ti < g.n_t; ti++)
    for (uint32_t tj = ti; tj < g.n_t; tj++, tid++) {
      if (owner[tid] != cfg->shard_rank) continue;
      tiles.push_back({(uint16_t)ti, (uint16_t)tj});
      for (uint32_t a = 0; a < 2; a++)  // 64 x 64 tiles of the table-driven EM kernel
        for (uint32_t b = 0; b < 2; b++) {
          const uint32_t i64 = 2 * ti + a, j64 = 2 * tj + b;
          if (i64 > j64 || (uint64_t)i64 * 64 >= g.n_ind || (uint64_t)j64 * 64 >= g.n_ind) continue;
          tiles64.push_back({(uint16_t)i64, (uint16_t)j64});
        }
      for (uint32_t a = 0; a < NGD_IG_PER_TILE; a++)
        for (uint32_t b = 0; b < NGD_IG_PER_TILE; b++) {
          uint32_t ig = ti * NGD_IG_PER_TILE + a, jg = tj * NGD_IG_PER_TILE + b;
          if (ig > jg) continue;                                  // strictly lower: no i<j pair
          if ((uint64_t)ig * 16 >= g.n_ind || (uint64_t)jg * 16 >= g.n_ind) continue;  // all padding
          tiles16.push_back({(uint16_t)ig, (uint16_t)jg});
        }
    }
  e->n_tiles = (uint32_t)tiles.size();
  e->n_tiles16 = (uint32_t)tiles16.size();
  e->n_tiles64 = (uint32_t)tiles64.size();
  // job list of the MFMA kernel (ngd_job, units of 16 individuals).
  std::vector<ngd_job> jobs;
  const uint32_t n_igv = (uint32_t)((g.n_ind + 15) / 16);  // groups that hold at least one individual
  // auto: up to 384 individuals only the tiles a block needs are issued (accum_mfma.hip EXACT); where a slice's jobs fit
  // one workgroup they run in step, each operand fragment leaving HBM once: up to 13 groups of 16 individuals as 16
  // blocks of 2 x 4 tiles with the operands staged through LDS, up to 16 groups as 10 blocks of 4 x 4 ([measured]
  // 100 000 sites, ms per matrix, plain / in step: n_ind = 100: 0.143 / 0.108; 200: 0.356 / 0.262; 250: 0.427 / 0.387;
  // the forms fall back where a slice's jobs do not fit one workgroup)
  e->exact_shapes = cfg->exact_shapes ? (cfg->exact_shapes == 1 || cfg->exact_shapes == 7 ? 0 : (int)cfg->exact_shapes - 1)
                                      : (g.n_pad > 384 ? 0 : n_igv <= 13 ? 5 : n_igv <= 16 ? 3 : 1);
  // Above 384 padded individuals every block runs the full 4 x 4 pattern (form 0).  ngd_config.exact_shapes = 7: the
  // blocks ON the diagonal leave out the 6 tiles below it (10 of 16; on a one-image engine their row fragments are their
  // column fragments: 4 loads per k-group instead of 8) -- measured, not the default (see accum_mfma.hip).
  e->tri_diag = cfg->exact_shapes == 7;
  if (e->exact_shapes == 2 || e->exact_shapes >= 4) {
    // strips of two row groups, cut into blocks of four column groups from the diagonal on (the first block of a strip
    // is triangular: 7 tiles of 8); an odd last row is its diagonal tile.  Under pair-tile sharding blocks must not
    // straddle a 128-tile (8 groups): the first block of a strip then ends at the next multiple of four.
    const bool aligned = world > 1;
    for (uint32_t r = 0; r < n_igv; r += 2) {
      if (n_igv - r == 1) {
        if (owner[ngd_tile_id(g.n_t, r / 8, r / 8)] == cfg->shard_rank) jobs.push_back({(uint16_t)r, (uint16_t)r, 1, 1, 1, 0});
        break;
      }
      for (uint32_t c = r; c < n_igv;) {
        uint32_t w = std::min(4u, n_igv - c);
        if (aligned && c % 4) w = std::min(w, 4 - c % 4);
        if (owner[ngd_tile_id(g.n_t, r / 8, c / 8)] == cfg->shard_rank)
          jobs.push_back({(uint16_t)r, (uint16_t)c, 2, (uint8_t)w, (uint8_t)(c == r), 0});
        c += w;
      }
    }
    auto cost = [](const ngd_job &j) { return j.tri ? j.rows * j.cols - (j.rows > 1 ? 1 : 0) : j.rows * j.cols; };
    std::stable_sort(jobs.begin(), jobs.end(), [&](const ngd_job &a, const ngd_job &b) { return cost(a) > cost(b); });
  } else if (e->exact_shapes) {  // (1, or 3: the same blocks, one workgroup per slice)
    // blocks of up to 4 x 4 groups over the valid groups only; the last block row / column is narrower,
    // blocks on the diagonal are triangular.  Most expensive first, four to a workgroup.
    const uint32_t nb = (n_igv + 3) / 4;
    for (uint32_t bi = 0; bi < nb; bi++)
      for (uint32_t bj = bi; bj < nb; bj++) {
        if (owner[ngd_tile_id(g.n_t, bi / 2, bj / 2)] != cfg->shard_rank) continue;
        const uint8_t r = (uint8_t)std::min(4u, n_igv - 4 * bi), c = (uint8_t)std::min(4u, n_igv - 4 * bj);
        jobs.push_back({(uint16_t)(4 * bi), (uint16_t)(4 * bj), r, c, (uint8_t)(bi == bj), 0});
      }
    auto cost = [](const ngd_job &j) { return j.tri ? j.rows * (j.rows + 1) / 2 : j.rows * j.cols; };
    std::stable_sort(jobs.begin(), jobs.end(), [&](const ngd_job &a, const ngd_job &b) { return cost(a) > cost(b); });
  } else {
    // Off-diagonal 128-tile -> its four 64x64 blocks in one workgroup (they share operands); the blocks of
    // the diagonal tiles (two on the diagonal, one above it) follow, packed four to a workgroup.  Every
    // block runs the full 4x4 pattern, so all workgroups of a slice progress at one rate --
    // tri_diag: the blocks ON the diagonal are triangular (10 tiles of 16) and come last, in workgroups of their
    // own, so that the four jobs of a workgroup still move through the sites together.
    std::vector<ngd_job> diag, ondiag;
    auto live = [&](uint32_t r, uint32_t c) { return r < n_igv && c < n_igv; };
    for (const ngd_tile &t : tiles) {
      const uint16_t r0 = t.ti * NGD_IG_PER_TILE, c0 = t.tj * NGD_IG_PER_TILE;
      if (t.ti != t.tj) {
        for (uint16_t a = 0; a < 2; a++)
          for (uint16_t b = 0; b < 2; b++) {
            ngd_job j = {(uint16_t)(r0 + 4 * a), (uint16_t)(c0 + 4 * b), 4, 4, 0, 0};
            if (!live(j.ig0, j.jg0)) j.rows = 0;  // only padding individuals
            jobs.push_back(j);
          }
      } else {
        const uint8_t tri = e->tri_diag ? 1 : 0;
        const ngd_job d[3] = {{r0, c0, 4, 4, tri, 0}, {r0, (uint16_t)(c0 + 4), 4, 4, 0, 0},
                              {(uint16_t)(r0 + 4), (uint16_t)(c0 + 4), 4, 4, tri, 0}};
        for (const ngd_job &j : d)
          if (live(j.ig0, j.jg0)) (j.tri ? ondiag : diag).push_back(j);
      }
    }
    for (const ngd_job &j : diag) jobs.push_back(j);
    if (!ondiag.empty()) {
      while (jobs.size() % 4) jobs.push_back({0, 0, 0, 0, 0, 0});  // (a workgroup of triangular blocks only)
      for (const ngd_job &j : ondiag) jobs.push_back(j);
    }
  }
  if (e->exact_shapes >= 3) {
    // One workgroup per slice, its wavefronts in step (accum_mfma.hip EXACT = 3 / 4): wavefront w runs on SIMD w % 4, so
    // the jobs are dealt, most expensive first, to the least loaded of four bins and wavefront w takes bin w % 4's
    // next job.  At most 12 wavefronts of 4 x 4 blocks (3 per SIMD at that kernel's register count) or 16 of 2 x 4:
    // else the plain exact form of the same blocks.
    const bool small = e->exact_shapes >= 4;
    auto cost = [&](const ngd_job &j) {
      return small ? (j.tri ? j.rows * j.cols - (j.rows > 1 ? 1 : 0) : j.rows * j.cols)
                   : (j.tri ? j.rows * (j.rows + 1) / 2 : j.rows * j.cols);
    };
    std::vector<ngd_job> bin[4];
    uint32_t load[4] = {0, 0, 0, 0};
    for (const ngd_job &j : jobs) {
      uint32_t b = 0;
      for (uint32_t q = 1; q < 4; q++)
        if (load[q] < load[b]) b = q;
      bin[b].push_back(j);
      load[b] += cost(j);
    }
    std::stable_sort(bin, bin + 4, [](const std::vector<ngd_job> &a, const std::vector<ngd_job> &b) { return a.size() > b.size(); });
    if (jobs.empty() || bin[0].size() > (small ? 4u : 3u)) {
      e->exact_shapes = small ? 2 : 1;
    } else {  // (the fuller bins first: no padding wavefront before the last real one)
      jobs.clear();
      for (size_t d = 0; d < bin[0].size(); d++)
        for (uint32_t b = 0; b < 4; b++)
          if (d < bin[b].size()) jobs.push_back(bin[b][d]);
    }
  }
  for (const ngd_job &j : jobs)  // every block must have a code path in the kernel's form (accum_mfma.hip)
    if (!ngd_mfma_shape_listed(e->exact_shapes, j.rows, j.cols, j.tri))
      return bail(fail(NGD_E_HIP, "ngd_create: internal -- a block shape the MFMA kernel's form does not list"));
  const uint32_t jobs_per_wg = e->exact_shapes >= 3 ? (uint32_t)jobs.size() : e->exact_shapes ? 1 : 4;
  e->wg_waves = jobs_per_wg;
  while (jobs.size() % jobs_per_wg) jobs.push_back({0, 0, 0, 0, 0, 0});
  e->n_wg = (uint32_t)(jobs.size() / jobs_per_wg);
  if (kernel == NGD_KERNEL_STREAM && world > 1) {
    for (const ngd_tile &t : tiles)
      for (uint64_t i = (uint64_t)t.ti * NGD_TILE; i < std::min<uint64_t>(g.n_ind, (t.ti + 1ull) * NGD_TILE); i++)
        for (uint64_t j = std::max<uint64_t>(i + 1, (uint64_t)t.tj * NGD_TILE);
             j < std::min<uint64_t>(g.n_ind, (t.tj + 1ull) * NGD_TILE); j++)
          pairs.push_back(ngd_pair_idx(g.n_ind, i, j));
    std::sort(pairs.begin(), pairs.end());
    e->n_owned_pairs = pairs.size();
  } else {
    e->n_owned_pairs = 0;
    for (const ngd_tile &t : tiles)
      for (uint64_t i = (uint64_t)t.ti * NGD_TILE; i < std::min<uint64_t>(g.n_ind, (t.ti + 1ull) * NGD_TILE); i++) {
        uint64_t jlo = std::max<uint64_t>(i + 1, (uint64_t)t.tj * NGD_TILE);
        uint64_t jhi = std::min<uint64_t>(g.n_ind, (t.tj + 1ull) * NGD_TILE);
        if (jhi > jlo) e->n_owned_pairs += jhi - jlo;
      }
  }

  const uint64_t n_pairs = ngd_n_pairs(g.n_ind);
  // + NGD_KG_TAIL zeroed k-groups: the MFMA kernel's operand pipeline runs ahead of its slice
  const uint64_t frag_elems = (g.n_kg + NGD_KG_TAIL) * (uint64_t)g.n_ig * 64;
#define TRY(x)                       \
  do {                               \
    rc = (x);                        \
    if (rc != NGD_OK) return bail(rc); \
  } while (0)
  TRY(dev_alloc(e, &e->d_tiles, tiles.size(), false));
  TRY(dev_alloc(e, &e->d_tiles16, tiles16.size(), false));
  e->h_tiles16 = tiles16;
  TRY(dev_alloc(e, &e->d_tiles64, tiles64.size(), false));
  TRY(dev_alloc(e, &e->d_pairs, pairs.size(), false));
  TRY(dev_alloc(e, &e->d_jobs, jobs.size(), false));
  if (!jobs.empty())
    if (hipMemcpy(e->d_jobs, jobs.data(), jobs.size() * sizeof(ngd_job), hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(NGD_E_HIP, "ngd_create: job list upload failed"));

  if (!tiles.empty())
    if (hipMemcpy(e->d_tiles, tiles.data(), tiles.size() * sizeof(ngd_tile), hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(NGD_E_HIP, "ngd_create: tile list upload failed"));
  if (!tiles16.empty())
    if (hipMemcpy(e->d_tiles16, tiles16.data(), tiles16.size() * sizeof(ngd_tile), hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(NGD_E_HIP, "ngd_create: tile list upload failed"));
  if (!tiles64.empty())
    if (hipMemcpy(e->d_tiles64, tiles64.data(), tiles64.size() * sizeof(ngd_tile), hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(NGD_E_HIP, "ngd_create: tile list upload failed"));
  if (!pairs.empty())
    if (hipMemcpy(e->d_pairs, pairs.data(), pairs.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(NGD_E_HIP, "ngd_create: pair list upload failed"));

  // ---- resident images (zero-filled: padding individuals/sites contribute nothing) ----
  if (kernel == NGD_KERNEL_STREAM) {
    TRY(dev_alloc_pieces(e, &e->PI, g.n_ind * g.n_sites_pad * 3, true));  // (a row of sites per individual: a load needs all of it)
  } else {
    TRY(dev_alloc_pieces(e, &e->PA, frag_elems, true, ngd_engine::PIECE_FRAG));
    e->single_image = kernel == NGD_KERNEL_MFMA && cfg->single_image == 1;
    if (kernel == NGD_KERNEL_MFMA && (cfg->single_image == 2 || cfg->single_image == 0)) {
      const bool ok = ngd_score_congruence(cfg->score, e->sc.c, e->sc.d) == NGD_OK;
      if (!ok && cfg->single_image == 2)
        return bail(fail(NGD_E_INVALID, "ngd_create: single_image = 2 needs a symmetric score matrix (single_image = 1 takes any)"));
      if (ok) {
        // the reference's two matrices (parse_args.cpp:25-27, :134-137): t = (p0 + p1 + p2, +-(p2 - p0), p1) -- the third
        // square of --avg_nuc_dist has weight 0 and an empty row, which then carries p1 all the same -- is the form the
        // fix-up pass recovers p from (fixup.hip)
        double *c = e->sc.c;
        if (e->sc.d[2] == 0 && c[6] == 0 && c[7] == 0 && c[8] == 0) c[7] = 1.0;
        const bool form = c[0] == 1 && c[1] == 1 && c[2] == 1 && c[4] == 0 && (c[3] == 1 || c[3] == -1) && c[5] == -c[3] &&
                          c[6] == 0 && c[7] == 1 && c[8] == 0;
        e->sc.fix = form ? 1 : 0;
        e->sc.fix_sign = c[5];
      }
      // auto: one image in congruent coordinates where it is safe (the fix-up pass exists for this matrix) and where memory
      // matters -- the block forms of a few hundred individuals take no per-index weights in their fastest variant
      e->congruent = ok && (cfg->single_image == 2 || (e->sc.fix && e->exact_shapes == 0));
      e->sc.congruent = e->congruent ? 1 : 0;
      if (!e->congruent) e->sc.fix = 0;
    }
    if (e->single_image) {  // ... and as much of the second image as the caller has memory to spare for
      e->qb_res_kg = std::min<uint64_t>(g.n_kg, ((uint64_t)cfg->second_image_mib << 20) / ((uint64_t)g.n_ig * 64 * 8));
      if (e->qb_res_kg == g.n_kg) { e->single_image = false; e->qb_res_kg = 0; }  // all of it: the two-image engine
    }
    if (kernel == NGD_KERNEL_MFMA && !e->single_image && !e->congruent) TRY(dev_alloc_pieces(e, &e->QB, frag_elems, true, ngd_engine::PIECE_FRAG));
    if (e->qb_res_kg) TRY(dev_alloc(e, &e->QB_res, (e->qb_res_kg + NGD_KG_TAIL) * (uint64_t)g.n_ig * 64, false));
  }
  if (cfg->pairwise_del) {
    TRY(dev_alloc(e, &e->mask, g.n_ind * (uint64_t)g.n_words, true));
    TRY(dev_alloc(e, &e->planes, 32ull * g.n_words, true));
  }
  TRY(dev_alloc(e, &e->d_ws, g.n_sites_pad + 4 * NGD_KG_TAIL, true));
  if (kernel == NGD_KERNEL_MFMA) TRY(dev_alloc(e, &e->d_wk, 4 * (g.n_kg + NGD_KG_TAIL), true));
  if (e->congruent) {
    TRY(dev_alloc(e, &e->d_wD, 4 * (g.n_kg + NGD_KG_TAIL), false));
    ngd_launch_index_weights(e->st, 4 * (g.n_kg + NGD_KG_TAIL), e->sc.d, e->d_wD);
  }
  if (e->congruent && e->sc.fix) {
    TRY(dev_alloc_pieces(e, &e->SM, g.n_sites * g.n_ind, true, ngd_engine::PIECE_SITE_MAJOR, g.n_ind * 8));
    e->fix_cap = (uint32_t)std::min<uint64_t>(n_pairs, NGD_FIX_LIST);
    TRY(dev_alloc(e, &e->d_fixlist, e->fix_cap, false));
    TRY(dev_alloc(e, &e->d_fixcount, 1, true));
    TRY(dev_alloc(e, &e->d_fixseen, n_pairs / 32 + 1, true));
    TRY(dev_alloc(e, &e->d_fixparts, NGD_FIX_CAP, false));
    if (hipHostMalloc((void **)&e->h_fixcount, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess)
      return bail(fail(NGD_E_NOMEM, "ngd_create: no pinned host memory for the fix-up count"));
    *e->h_fixcount = 0;
  }
  TRY(dev_alloc(e, &e->d_sum, n_pairs, true));
  TRY(dev_alloc(e, &e->d_cnt, n_pairs, true));

  // ---- split over the site axis: slices -> slabs, reduced in fixed order ----
  if (kernel == NGD_KERNEL_MFMA) {
    uint64_t want = cfg->wg_target ? cfg->wg_target : 8192;
    const uint32_t wg_per_slice = std::max(1u, e->n_wg);
    uint64_t ks = (want * (e->exact_shapes && e->exact_shapes < 3 ? 4 : 1) + wg_per_slice - 1) / wg_per_slice;  // EXACT: 1-wave workgroups
    uint64_t max_ks = std::max<uint64_t>(8, g.n_kg / 128);  // at least 128 k-groups per slice
    ks = std::min(ks, max_ks);
    ks = std::max<uint64_t>(8, (ks + 7) / 8 * 8);
    {
      // Workgroups all last the same, so an XCD works through its share (n_wg * ks / 8 workgroups) in rounds of
      // as many as it holds at a time, and a last round that is nearly empty costs as much as a full one
      // ([measured] cfg 3, 34 workgroups per slice: ks = 232 -> 10.27 rounds, 47.5 ms; 240 -> 10.63, 46.1 ms;
      // 248 -> 10.98, 44.65 ms; cfg 2, 10 single-wavefront jobs per slice: 584 -> 1.90 rounds, 448 -> 1.46 rounds,
      // 0.41 ms, 304 -> 0.99 rounds, 0.345 ms and half the slabs to reduce).  Among the slice counts from half the
      // target to 15 % above it take the one whose last round is fullest.
      hipDeviceProp_t prop;
      const uint32_t cus_per_xcd = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8
                                       ? (uint32_t)prop.multiProcessorCount / 8 : 32;
      // (in-step forms: whole workgroups of wg_waves jobs + the prefetching wavefront, 20 / 12 wavefronts to a CU)
      const uint32_t sync_wgs = e->exact_shapes >= 3
                                    ? std::max(1u, (e->exact_shapes >= 4 ? 20u : 12u) / std::min(16u, e->wg_waves + 1)) : 0;
      const double slots = (double)cus_per_xcd * (e->exact_shapes >= 3 ? sync_wgs : e->exact_shapes == 2 ? 24 : e->exact_shapes ? 12 : 3);
      // ... plus what the slice count costs afterwards: the slab reduction reads one plane per slice ([measured] 0.8 us
      // per slice at n_ind = 1000, i.e. ~5 TB/s), against an accumulation pass at ~0.8 of the FP64 peak.  It decides
      // between slice counts that fill their rounds equally well: a 1/8 site shard of cfg 3 takes 112 slices instead
      // of 248 (6.36 instead of 6.54 ms per matrix, the accumulation itself is flat from 88 to 500 slices).
      const double accum_s = 6.0 * (double)e->n_owned_pairs * (double)g.n_sites / (0.8 * 78.6e12);
      const double reduce_s_per_slice = 8.0 * (double)e->n_owned_pairs / 5e12;
      // A single-image engine walks the pass in ranges (launch_accumulate()): every launch has all the slices, and every
      // block adds to its plane of the slab at the end of each ([measured] cfg 3, 248 slices, 12 ranges: +0.53 ms per
      // launch, 2.1 us per slice -- 2.7 reductions' worth).  Fewer slices then: as few as fill their rounds.
      const uint64_t qb_ranges =
          e->single_image ? (g.n_kg - e->qb_res_kg + single_image_span(g) - 1) / single_image_span(g) + (e->qb_res_kg ? 1 : 0) : 0;
      const double per_slice_s = reduce_s_per_slice * (1.0 + 2.7 * (double)qb_ranges);
      double best = 1e30;
      uint64_t best_ks = ks;
      for (uint64_t c = e->single_image ? 8 : std::max<uint64_t>(8, ks / 3 / 8 * 8); c <= std::min(max_ks, ks * 115 / 100); c += 8) {
        const double rounds = (double)wg_per_slice * (double)(c / 8) / slots;
        const double waste = std::ceil(rounds - 1e-9) / rounds + (double)c * per_slice_s / std::max(accum_s, 1e-9);
        if (waste < best) { best = waste; best_ks = c; }
      }
      ks = best_ks;
    }
    if (cfg->n_slices) ks = std::min<uint64_t>(cfg->n_slices, max_ks);  // a caller's count is held to the same bound
    ks = std::max<uint64_t>(8, (ks + 7) / 8 * 8);
    e->n_ks = (uint32_t)ks;
    e->per_slice = ((g.n_kg + ks - 1) / ks + 3) / 4 * 4;  // whole pipeline trips (accum_mfma.hip DEPTH)
    TRY(dev_alloc_pieces(e, &e->slab, ks * (uint64_t)g.n_pad * g.n_pad, false));
    // ([0..1] the clock sample; [2] set by a block whose shape the kernel does not list: mfma_fault())
    if (hipHostMalloc((void **)&e->h_clk, 4 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&e->d_clk, e->h_clk, 0) != hipSuccess)
      return bail(fail(NGD_E_NOMEM, "ngd_create: no pinned host memory for the clock sample"));
    e->h_clk[0] = e->h_clk[1] = e->h_clk[2] = e->h_clk[3] = 0;
    if (e->single_image) {
      // scratch for QB: one range of a whole pass (launch_accumulate(); partial-sum passes grow it if a bootstrap
      // block is longer)
      e->qb_chunk_kg = single_image_span(g);
      uint64_t n_ranges = 0;
      const uint64_t rest_kg = g.n_kg - e->qb_res_kg;  // (what is not resident: ngd_config.second_image_mib)
      const uint64_t range_kg = std::min<uint64_t>(rest_kg, qb_piece(rest_kg, e->n_ks, e->qb_chunk_kg, &n_ranges) * e->n_ks);
      e->qb_chunk_elems = (range_kg + NGD_KG_TAIL) * (uint64_t)g.n_ig * 64;
      TRY(dev_alloc(e, &e->qb_chunk, e->qb_chunk_elems, false));
    }
  } else if (kernel == NGD_KERNEL_EM_TABLE) {
    // 64 x 64 tiles x slices of sites; a workgroup works a site in ~10 us, so slices of a few thousand sites keep
    // the tail of the launch short without making the slab large
    e->em_shape = (int)cfg->variant;
    uint64_t want = cfg->wg_target ? cfg->wg_target : 16384;
    uint64_t ks = e->n_tiles64 ? (want + e->n_tiles64 - 1) / e->n_tiles64 : 1;
    uint64_t max_ks = std::max<uint64_t>(1, g.n_sites / 64);
    ks = std::min(ks, max_ks);
    if (cfg->n_slices) ks = std::min<uint64_t>(cfg->n_slices, g.n_sites);  // never more slices than sites
    e->n_ks = (uint32_t)ks;
    e->per_slice = (g.n_sites + ks - 1) / ks;
    TRY(dev_alloc_pieces(e, &e->slab, ks * (uint64_t)g.n_pad * g.n_pad, true));
    TRY(dev_alloc(e, &e->d_emcnt, 4, true));
    {
      // pair slots of the spilled-terms plan (em_spill_impl): a row of a tile takes one slot group per group of 16 columns
      // that holds a pair -- none for a diagonal tile's lower triangle or for the columns at and beyond n_ind
      std::vector<uint32_t> rowpg((size_t)tiles64.size() * 64, 0xffffffffu);
      uint64_t n_live = 0;
      for (size_t t = 0; t < tiles64.size(); t++)
        for (uint32_t row = 0; row < 64; row++) {
          const uint64_t i = (uint64_t)tiles64[t].ti * 64 + row, j0 = (uint64_t)tiles64[t].tj * 64;
          if (i >= g.n_ind || j0 >= g.n_ind) continue;
          const uint64_t first = tiles64[t].ti == tiles64[t].tj ? row + 1 : 0, last = std::min<uint64_t>(63, g.n_ind - 1 - j0);
          if (first > last) continue;
          rowpg[t * 64 + row] = (uint32_t)n_live - (uint32_t)(first >> 4);  // (+ a column group's index = its slot group)
          n_live += (last >> 4) - (first >> 4) + 1;
        }
      if (n_live + 4 < (1ull << 31)) {  // (else: the plan is not offered, em_spill_impl)
        e->n_pg_live = (uint32_t)n_live;
        e->n_pg_spill = (uint32_t)((n_live + 3) / 4 * 4);  // a wavefront of the contraction takes 2 or 4 slot groups
        TRY(dev_alloc(e, &e->d_rowpg, rowpg.size(), false));
        if (!rowpg.empty() && hipMemcpy(e->d_rowpg, rowpg.data(), rowpg.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
          return bail(fail(NGD_E_HIP, "ngd_create: slot map upload failed"));
      }
    }
  } else if (kernel == NGD_KERNEL_EM_FAST || kernel == NGD_KERNEL_EM_FAITHFUL) {
    uint64_t want = cfg->wg_target ? cfg->wg_target : 4096;
    uint64_t ks = e->n_tiles16 ? (want + e->n_tiles16 - 1) / e->n_tiles16 : 1;
    uint64_t max_ks = std::max<uint64_t>(1, g.n_sites / 256);
    ks = std::min(ks, max_ks);
    if (cfg->n_slices) ks = std::min<uint64_t>(cfg->n_slices, g.n_sites);  // never more slices than sites
    e->n_ks = (uint32_t)ks;
    e->per_slice = (g.n_sites + ks - 1) / ks;
    TRY(dev_alloc_pieces(e, &e->slab, ks * (uint64_t)g.n_pad * g.n_pad, false));
  }
  // upload staging (ngd_upload_sites / _ind_major): at most ~256 MiB of raw doubles, allocated by the first upload that
  // needs it (a staged load -- ngd_stage_* -- never does)
  e->staging_sites = std::max<uint64_t>(1, std::min<uint64_t>(g.n_sites, (256ull << 20) / (g.n_ind * 24)));
#undef TRY
  if (hipStreamSynchronize(e->st) != hipSuccess) return bail(fail(NGD_E_HIP, "ngd_create: sync failed"));
  if (int prc = piece_start(e)) return bail(prc);  // the images' and slabs' memory arrives behind this call (dev_alloc_pieces)
  *out = e;
  return NGD_OK;
}

static int eager_discard(ngd_engine *e);

static int upload_common(ngd_engine *e, const double *p, int ind_major, uint64_t s0, uint64_t n) {
  if (!e || !p) return fail(NGD_E_INVALID, "upload: null argument");
  if (e->committed) return fail(NGD_E_INVALID, "upload: data set already committed");
  if (s0 + n > e->g.n_sites || s0 + n < s0) return fail(NGD_E_INVALID, "upload: site range out of bounds");
  HIPCHK(hipSetDevice(e->device));
  if (int rc = piece_join(e)) return rc;
  if (int rc = eager_discard(e)) return rc;  // (sites may be uploaded again: nothing accumulated beside a staged load is kept)
  e->stage_in_order = false;
  if (!e->staging)
    if (int rc = dev_alloc(e, &e->staging, e->staging_sites * e->g.n_ind * 3, false)) return rc;
  const uint64_t n_ind = e->g.n_ind;
  for (uint64_t done = 0; done < n;) {
    const uint64_t c = std::min(e->staging_sites, n - done);
    if (ind_major) {
      // rows = individuals, each row = c sites x 24 B out of an n_sites-long row
      HIPCHK(hipMemcpy2DAsync(e->staging, c * 24, p + (s0 + done) * 3, e->g.n_sites * 24, c * 24, n_ind,
                              hipMemcpyHostToDevice, e->st));
    } else {
      HIPCHK(hipMemcpyAsync(e->staging, p + done * n_ind * 3, c * n_ind * 24, hipMemcpyHostToDevice, e->st));
    }
    ngd_launch_layout(e->st, e->g, e->staging, ind_major, s0 + done, c, e->sc, e->cfg.pairwise_del, e->PA,
                      e->QB, e->congruent ? e->SM : e->PI, e->mask);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->st));  // staging buffer is reused by the next chunk
    done += c;
  }
  return NGD_OK;
}

int ngd_upload_sites(ngd_engine *e, const double *p, uint64_t s0, uint64_t n) {
  return upload_common(e, p, 0, s0, n);
}

int ngd_upload_ind_major(ngd_engine *e, const double *p) {
  if (!e) return fail(NGD_E_INVALID, "upload: null engine");
  return upload_common(e, p, 1, 0, e->g.n_sites);
}

// [measured, round 6] `tools/host_read_pipeline`: pieces of 32-64 MiB through a ring of 4-8 pinned buffers keep the copy
// engine at the link's rate (56.9 of 57.6 GB/s) while the caller fills the next ones; one piece costs 0.6-1.1 ms of copy,
// far above a launch.  Pinned memory is allocated at ~6.5 GiB/s, so the ring is kept to 192 MiB.
// A slot is allocated when the ring first comes to it: the copy engine is then already busy with the slots before it
// (6 x (hipHostMalloc + hipMalloc) up front were 40-50 ms before the first byte moved).
static void stage_reap(ngd_engine *e) {
  if (e->ring_reaper.joinable()) e->ring_reaper.join();
}

// ---- the full-data pass beside a staged load (NGD_OPT_EAGER_FULL) ----
static bool eager_supported(const ngd_engine *e) {
  if (e->kernel == NGD_KERNEL_EM_TABLE) return e->n_ks > 1;
  return e->kernel == NGD_KERNEL_MFMA && e->exact_shapes == 0 && !e->single_image && e->n_ks >= 16;
}

// slices [ks0, ks0 + n) of the plain pass on `st` (results: their planes of e->slab, as a whole launch leaves them)
static void launch_plain_slices(ngd_engine *e, hipStream_t st, uint32_t ks0, uint32_t n, bool beside_a_load) {
  const ngd_geom &g = e->g;
  if (e->kernel == NGD_KERNEL_EM_TABLE) {
    // beside a load ONE workgroup per CU (12 KB more LDS than its tables need): the chip is not full of workgroups that
    // last tens of milliseconds when the next piece's preparation kernel wants wave slots and registers
    ngd_launch_accum_em_table_slices(st, g, e->PA, e->sc, e->cfg.pairwise_del, e->em_shape, e->d_tiles64, e->n_tiles64, ks0, n,
                                     e->per_slice, e->slab, e->d_emcnt, beside_a_load ? 12u << 10 : 0u);
  } else {
    ngd_launch_accum_mfma(st, g, e->PA, e->congruent ? e->PA : e->QB, e->congruent ? e->d_wD : nullptr, nullptr, e->d_jobs, e->n_wg,
                          e->exact_shapes, e->wg_waves, n, e->per_slice, g.n_kg, 0, 0, e->slab, e->d_clk, ks0);
  }
}

// after the piece of sites [s0, s0 + n) has been submitted (its preparation kernel is on e->st, k0_done[b] recorded)
static int eager_advance(ngd_engine *e, uint64_t s0, uint64_t n, int b) {
  if (!e->opt_eager || !e->stage_in_order) return NGD_OK;
  if (s0 != e->stage_prefix) { e->stage_in_order = false; return NGD_OK; }  // (out of order: what is launched stays valid)
  e->stage_prefix = s0 + n;
  const ngd_geom &g = e->g;
  if (e->stage_prefix >= g.n_sites) return NGD_OK;  // the last piece: ngd_run() launches what is left
  uint32_t done;
  if (e->kernel == NGD_KERNEL_EM_TABLE) {
    done = (uint32_t)std::min<uint64_t>(e->n_ks, e->stage_prefix / e->per_slice);
  } else {
    // a slice's k-groups + the NGD_KG_TAIL groups its operand pipeline runs ahead: index 4 kg + 3 belongs to site (4 kg + 3) / 3
    const uint64_t kg_ready = 3 * e->stage_prefix / 4;  // k-groups whose every index is below 3 * prefix
    const uint64_t full = kg_ready > NGD_KG_TAIL ? (kg_ready - NGD_KG_TAIL) / e->per_slice : 0;
    done = (uint32_t)std::min<uint64_t>(e->n_ks, full) / 8 * 8;  // (launches of whole eights of slices: the XCD deal)
  }
  const uint32_t batch = e->kernel == NGD_KERNEL_EM_TABLE ? std::max(1u, e->n_ks / 32) : std::max(8u, e->n_ks / 8 / 8 * 8);
  if (done < e->eager_slices + batch) return NGD_OK;
  // ONE batch in flight at a time, and a bounded one: what is launched here runs beside the load at a reduced rate (the
  // table-driven EM kernel with one workgroup per CU: 0.56 of its speed) and must not still be running long after it
  // ([measured] every completed slice launched at once: cfg 4's matrix 4.0 s instead of 2.26)
  if (e->eager_valid) {
    const hipError_t q = hipEventQuery(e->ev_eager);
    if (q == hipErrorNotReady) { (void)hipGetLastError(); return NGD_OK; }
    HIPCHK(q);
  }
  done = std::min(done, e->eager_slices + (e->kernel == NGD_KERNEL_EM_TABLE ? batch : 2 * batch));
  {  // the slab's planes of these slices must be mapped (its memory arrives after the images': dev_alloc_pieces)
    std::lock_guard<std::mutex> lk(e->piece_mu);
    for (auto &q : e->piece_ranges)
      if (q->va == (void *)e->slab && q->ready < std::min<size_t>(q->size, (size_t)done * g.n_pad * g.n_pad * 8)) return NGD_OK;  // (next piece)
  }
  if (!e->st_eager) {
    // (a stream confined to a part of the CUs -- hipExtStreamCreateWithCUMask, 7/8 or 3/4 of them -- lets the EM kernel keep
    // two workgroups per CU there and does more beside the load, but the preparation kernels then wait for the few CUs
    // left: [measured] cfg 4 end to end 2.78-2.83 and 2.89-2.92 s against 2.77-2.80 with the plain low-priority stream)
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(hipStreamCreateWithPriority(&e->st_eager, hipStreamNonBlocking, least));
    HIPCHK(hipEventCreateWithFlags(&e->ev_eager, hipEventDisableTiming));
  }
  HIPCHK(hipStreamWaitEvent(e->st_eager, e->k0_done[b], 0));  // this piece's preparation -- and every earlier one's -- is done
  launch_plain_slices(e, e->st_eager, e->eager_slices, done - e->eager_slices, true);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->ev_eager, e->st_eager));
  e->eager_slices = done;
  e->eager_valid = true;
  return NGD_OK;
}

// anything but the plain pass is about to use the slab (or the engine is going away): what was started is waited for and dropped
static int eager_discard(ngd_engine *e) {
  if (e->eager_valid) HIPCHK(hipStreamSynchronize(e->st_eager));
  e->eager_valid = false;
  e->eager_slices = 0;
  return NGD_OK;
}

static int stage_slots(ngd_engine *e);

static int stage_init(ngd_engine *e) {
  if (e->pin_sites) return NGD_OK;
  stage_reap(e);
  e->pin_sites = std::max<uint64_t>(1, std::min<uint64_t>(e->g.n_sites, (e->opt_stage_piece_mib << 20) / (e->g.n_ind * 24)));
  // (a data set of fewer pieces than the ring has slots takes only that many)
  e->ring_slots = (int)std::min<uint64_t>(e->opt_stage_ring, (e->g.n_sites + e->pin_sites - 1) / e->pin_sites);
  // ONE copy stream ([measured] copies alternating two streams load cfg 3 in the same 0.53 s, and a stream costs 7 ms to create)
  if (!e->st_copy[0]) HIPCHK(hipStreamCreateWithFlags(&e->st_copy[0], hipStreamNonBlocking));
  e->n_staged = 0;
  e->pin_cur = 0;
  if (!e->d_nan)
    if (int rc = dev_alloc(e, &e->d_nan, 1, true)) return rc;
  return stage_slots(e);
}

// A pinned buffer of the ring comes from hipHostMalloc (which allocates, zeroes and pins 4-KB pages at ~6.5 GiB/s: 5 ms per
// 32-MiB slot, ~25 ms before the ring has turned once).  Round 6 tried huge-page host memory registered with the runtime
// (posix_memalign + MADV_HUGEPAGE + hipHostRegister: the copies first to last byte 0.468 -> 0.440 s) and took it out again:
// in a process that created and destroyed engine after engine (tools/fuzz_large.py, case ~55 of 80) the GPU faulted on a HOST
// heap address -- registered ranges are handed back to malloc and come round again at the same addresses, and a
// registration that is released late takes the next one's mapping with it.  hipHostMalloc's buffers never share addresses.
static int pin_alloc(ngd_engine *e, int b, uint64_t bytes) {
  HIPCHK(hipHostMalloc((void **)&e->pin[b], bytes, hipHostMallocDefault));
  return NGD_OK;
}

static void pin_release(double *p) {
  if (p) (void)hipHostFree(p);
}

// every slot's device twin and events at once (cheap); pinned buffer 0 at once, the others by ring_maker
static int stage_slots(ngd_engine *e) {
  const uint64_t bytes = e->pin_sites * e->g.n_ind * 24;
  for (int b = 0; b < e->ring_slots; b++) {
    int rc = dev_alloc(e, &e->draw[b], bytes / 8, false);
    if (rc) return rc;
    HIPCHK(hipEventCreateWithFlags(&e->pin_free[b], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->k0_done[b], hipEventDisableTiming));
  }
  if (int rc = pin_alloc(e, 0, bytes)) return rc;
  e->ring_ready = 1;
  e->ring_stop = false;
  e->ring_maker_rc = 0;
  if (e->ring_slots > 1) {
    const int dev = e->device, n = e->ring_slots;
    e->ring_maker = std::thread([e, dev, n, bytes]() {
      if (hipSetDevice(dev) != hipSuccess) { e->ring_maker_rc = NGD_E_HIP; return; }
      for (int b = 1; b < n && !e->ring_stop.load(std::memory_order_relaxed); b++) {
        if (hipHostMalloc((void **)&e->pin[b], bytes, hipHostMallocDefault) != hipSuccess) {  // (the load goes on with the buffers it has)
          e->pin[b] = nullptr;
          (void)hipGetLastError();
          e->ring_maker_rc = NGD_E_NOMEM;
          return;
        }
        e->ring_ready.store(b + 1, std::memory_order_release);
      }
    });
  }
  return NGD_OK;
}

static void ring_maker_join(ngd_engine *e) {
  e->ring_stop = true;
  if (e->ring_maker.joinable()) e->ring_maker.join();
}

int ngd_stage_acquire(ngd_engine *e, double **host_buf, uint64_t *capacity_sites) {
  if (!e || !host_buf || !capacity_sites) return fail(NGD_E_INVALID, "ngd_stage_acquire: null argument");
  if (e->committed) return fail(NGD_E_INVALID, "ngd_stage_acquire: data set already committed");
  HIPCHK(hipSetDevice(e->device));
  int rc = stage_init(e);
  if (rc) return rc;
  const int b = e->pin_cur;
  // the copy out of this buffer, a turn of the ring ago, is done -- and so is the preparation kernel that read its device
  // twin (it follows the copy on the engine's stream, ~30 us): waited for HERE, on the host, so that the copy stream carries
  // no wait of its own ([measured] a stream-side wait on an event costs the copy engine ~50 us of idling per copy)
  HIPCHK(hipEventSynchronize(e->k0_done[b]));
  e->pin_lent = b;
  *host_buf = e->pin[b];
  *capacity_sites = e->pin_sites;
  return NGD_OK;
}

int ngd_stage_submit(ngd_engine *e, uint64_t s0, uint64_t n, const ngd_prep *prep) {
  if (!e || !prep) return fail(NGD_E_INVALID, "ngd_stage_submit: null argument");
  if (e->pin_lent < 0) return fail(NGD_E_INVALID, "ngd_stage_submit: no buffer acquired");
  if (prep->call_geno && prep->N_thresh > prep->call_thresh)  // call_geno(), gen_func.cpp:887-888
    return fail(NGD_E_INVALID, "missing data threshold must be smaller than calling genotype threshold!");
  if (n > e->pin_sites || s0 + n > e->g.n_sites || s0 + n < s0)
    return fail(NGD_E_INVALID, "ngd_stage_submit: site range out of bounds");
  HIPCHK(hipSetDevice(e->device));
  const int b = e->pin_lent;
  hipStream_t cs = e->st_copy[0];
  e->n_staged++;
  HIPCHK(hipMemcpyAsync(e->draw[b], e->pin[b], n * e->g.n_ind * 24, hipMemcpyHostToDevice, cs));
  HIPCHK(hipEventRecord(e->pin_free[b], cs));
  HIPCHK(hipStreamWaitEvent(e->st, e->pin_free[b], 0));
  if (int rc = piece_wait_sites(e, s0 + n)) return rc;  // (the part of the images these sites are written to is mapped)
  ngd_launch_prep_layout(e->st, e->g, e->draw[b], s0, n, prep->in_logscale, prep->call_geno, prep->N_thresh,
                         prep->call_thresh, e->sc, e->cfg.pairwise_del, e->PA, e->QB, e->congruent ? e->SM : e->PI, e->mask,
                         e->d_nan);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->k0_done[b], e->st));
  if (int rc = eager_advance(e, s0, n, b)) return rc;
  e->pin_lent = -1;
  e->pin_cur = (b + 1) % std::max(1, e->ring_ready.load(std::memory_order_acquire));  // (the buffers that exist by now)
  return NGD_OK;
}

int ngd_upload_raw_sites(ngd_engine *e, const double *raw, uint64_t s0, uint64_t n, const ngd_prep *prep) {
  if (!e || !raw || !prep) return fail(NGD_E_INVALID, "ngd_upload_raw_sites: null argument");
  for (uint64_t done = 0; done < n;) {
    double *buf;
    uint64_t cap;
    int rc = ngd_stage_acquire(e, &buf, &cap);
    if (rc) return rc;
    const uint64_t c = std::min(cap, n - done);
    memcpy(buf, raw + done * e->g.n_ind * 3, c * e->g.n_ind * 24);
    rc = ngd_stage_submit(e, s0 + done, c, prep);
    if (rc) return rc;
    done += c;
  }
  return NGD_OK;
}

int ngd_commit(ngd_engine *e) {
  if (!e) return fail(NGD_E_INVALID, "ngd_commit: null engine");
  HIPCHK(hipSetDevice(e->device));
  if (int rc = piece_join(e)) return rc;
  HIPCHK(hipStreamSynchronize(e->st));
  if (e->d_nan) {
    int flag = 0;
    HIPCHK(hipMemcpy(&flag, e->d_nan, sizeof(int), hipMemcpyDeviceToHost));
    ring_maker_join(e);
    {  // the pipeline is over: its buffers go back on a thread of their own (6 x hipHostFree + hipFree are ~30 ms)
      struct Slot { double *pin, *draw; hipEvent_t a, b; };
      std::vector<Slot> slots;
      for (int b = 0; b < ngd_engine::RING; b++) {
        if (e->pin[b] || e->draw[b]) slots.push_back({e->pin[b], e->draw[b], e->pin_free[b], e->k0_done[b]});
        if (e->draw[b]) e->dev_bytes -= e->pin_sites * e->g.n_ind * 24;
        e->pin[b] = nullptr; e->draw[b] = nullptr; e->pin_free[b] = nullptr; e->k0_done[b] = nullptr;
      }
      e->pin_sites = 0;
      e->ring_slots = 0;
      e->ring_ready = 0;
      stage_reap(e);
      const int dev = e->device;
      if (!slots.empty())
        e->ring_reaper = std::thread([slots, dev]() {
          (void)hipSetDevice(dev);
          for (const Slot &s : slots) {
            pin_release(s.pin);
            if (s.draw) (void)hipFree(s.draw);
            if (s.a) (void)hipEventDestroy(s.a);
            if (s.b) (void)hipEventDestroy(s.b);
          }
        });
    }
    e->pin_cur = 0;
    e->pin_lent = -1;
    if (flag) {  // reported once: a caller that uploads again starts from a clean flag and a fresh pipeline
      HIPCHK(hipMemset(e->d_nan, 0, sizeof(int)));
      return fail(NGD_E_NAN, "NaN found! Is the file format correct?");
    }
  }
  if (e->staging) {  // upload is over: give the staging buffer back
    HIPCHK(hipFree(e->staging));
    e->dev_bytes -= e->staging_sites * e->g.n_ind * 24;
    e->staging = nullptr;
  }
  if (e->QB_res)  // single-image engine: the part of the second image it keeps (stream order: before any pass)
    ngd_launch_qb_range(e->st, e->g, e->sc, e->PA, 0, std::min<uint64_t>(e->qb_res_kg + NGD_KG_TAIL, e->g.n_kg + NGD_KG_TAIL),
                        e->QB_res);
  e->committed = true;
  return NGD_OK;
}

int ngd_synth_fill_range(ngd_engine *e, uint64_t seed, double miss_frac, uint64_t site0) {
  if (!e) return fail(NGD_E_INVALID, "ngd_synth_fill: null engine");
  if (e->committed) return fail(NGD_E_INVALID, "ngd_synth_fill: data set already committed");
  HIPCHK(hipSetDevice(e->device));
  if (int rc = piece_join(e)) return rc;
  ngd_launch_synth(e->st, e->g, seed, miss_frac, site0, e->sc, e->cfg.pairwise_del, e->PA, e->QB,
                   e->congruent ? e->SM : e->PI, e->mask);
  HIPCHK(hipGetLastError());
  return ngd_commit(e);
}

int ngd_synth_fill(ngd_engine *e, uint64_t seed, double miss_frac) { return ngd_synth_fill_range(e, seed, miss_frac, 0); }

// w != NULL: one bootstrap replicate; for the MFMA kernel kgl is then the list of k-groups to visit and
// per_slice / kg_lim count list entries
// k_per_slice != 0 (MFMA, bootstrap blocks that are not whole k-groups): slices of k_per_slice contraction indices,
// masked by the per-slice 0/1 weights in e->d_wslice (w_stride k-groups per slice)
static int launch_accumulate(ngd_engine *e, const uint32_t *w, const uint32_t *kgl, uint64_t sites_eff, uint32_t n_ks,
                              uint64_t per_slice, uint64_t kg_lim, double *slab, uint64_t k_per_slice = 0,
                              uint32_t w_stride = 0) {
  const ngd_geom &g = e->g;
  switch (e->kernel) {
    case NGD_KERNEL_MFMA:
      if (!e->single_image) {
        // (single_image = 2: both operands from the one image, the congruence's diagonal on the weights -- of a plain pass too)
        ngd_launch_accum_mfma(e->st, g, e->PA, e->congruent ? e->PA : e->QB,
                              k_per_slice ? e->d_wslice : (w ? e->d_wk : (e->congruent ? e->d_wD : nullptr)),
                                (w && !k_per_slice) ? kgl : nullptr, e->d_jobs, e->n_wg, e->exact_shapes, e->wg_waves, n_ks, per_slice,
                                kg_lim, k_per_slice, w_stride, slab, e->d_clk);
      } else {
        // QB is formed range by range into a scratch (k_qb_range: HBM work, 49 GB a pass at cfg 3) on the accumulation
        // kernel's own stream, each range before the launch that reads it.
        //  * a whole pass (slab == e->slab): EVERY slice takes a piece of every range, so that each launch has the
        //    pass's full grid (a launch over a few whole slices would not fill the chip once: cfg 3 has 34 workgroups
        //    per slice and room for 768); a block adds its sums over the range to its plane of the slab (`resume`).  A
        //    slice is then not one contiguous run of k-groups, as it is with both images resident: the sums of the two
        //    engines agree to rounding (exactly where the arithmetic is exact: called genotypes), not bit for bit.
        //  * per-block partial sums (a slice = a bootstrap block, thousands of them): ranges of whole slices, in eights
        //    (the XCD deal of accum_mfma.hip); the kernel is handed the scratch moved back by the range's first k-group.
        // (single-image engines make no k-group lists: pass_impl() walks every k-group of a weighted pass)
        // [measured, cfg 3] forming a range on a second stream beside the launch over the range before it gains nothing:
        // the accumulation kernel slows by what the overlap hides, however few blocks form the range and with or
        // without non-temporal accesses (profiles/r04_single_image.txt; tools/experiments/single_image_two_streams.patch).
        const uint64_t kstride = (uint64_t)g.n_ig * 64;
        const uint64_t span = std::max<uint64_t>(1, e->qb_chunk_kg);
        const bool whole_pass = slab == e->slab && !k_per_slice;
        auto slice_kg0 = [&](uint64_t ks) { return k_per_slice ? (ks * k_per_slice) >> 2 : ks * per_slice; };
        auto slice_kg1 = [&](uint64_t ks) {
          return std::min<uint64_t>(kg_lim, k_per_slice ? ((ks + 1) * k_per_slice + 3) >> 2 : (ks + 1) * per_slice);
        };
        const double *wsel = k_per_slice ? e->d_wslice : (w ? e->d_wk : nullptr);
        // what the engine keeps of the second image (its first qb_res_kg k-groups, ngd_config.second_image_mib) is read
        // where it lies: one launch over that part of a whole pass, or over the slices that end inside it
        const uint64_t res = std::min<uint64_t>(e->qb_res_kg, kg_lim);
        uint32_t ks_first = 0;
        if (res && whole_pass) {
          const uint64_t piece_r = std::max<uint64_t>(4, ((res + n_ks - 1) / n_ks + 3) / 4 * 4);
          ngd_launch_accum_mfma(e->st, g, e->PA, e->QB_res, wsel, nullptr, e->d_jobs, e->n_wg, e->exact_shapes, e->wg_waves, n_ks,
                                piece_r, res, 0, 0, slab, e->d_clk, 0, 0);
        } else if (res) {
          while (ks_first + 8 <= n_ks && slice_kg1(ks_first + 7) <= res && slice_kg0(ks_first + 7) < kg_lim) ks_first += 8;
          if (ks_first)
            ngd_launch_accum_mfma(e->st, g, e->PA, e->QB_res, wsel, nullptr, e->d_jobs, e->n_wg, e->exact_shapes, e->wg_waves,
                                  ks_first, per_slice, kg_lim, k_per_slice, w_stride, slab, e->d_clk, 0);
        }
        const uint64_t rest0 = whole_pass ? res : 0;  // a whole pass goes on from here
        uint64_t piece = 0, n_ranges = 0;
        if (whole_pass && kg_lim > rest0) piece = qb_piece(kg_lim - rest0, n_ks, span, &n_ranges);
        uint32_t r = 0;
        for (uint32_t ks0 = ks_first; whole_pass ? r < n_ranges : ks0 < n_ks; r++) {
          uint32_t n = 8;
          uint64_t lo, hi;
          if (whole_pass) {
            lo = std::min<uint64_t>(rest0 + (uint64_t)r * piece * n_ks, kg_lim);
            hi = std::min<uint64_t>(lo + piece * n_ks, kg_lim);
          } else {
            while (ks0 + n < n_ks && slice_kg1(ks0 + n + 7) - slice_kg0(ks0) <= span && slice_kg0(ks0 + n) < kg_lim) n += 8;
            n = std::min(n, n_ks - ks0);
            lo = std::min<uint64_t>(slice_kg0(ks0), kg_lim);
            hi = std::max(lo, slice_kg1(ks0 + n - 1));
          }
          const uint64_t need = (hi - lo + NGD_KG_TAIL) * kstride;
          if (need > e->qb_chunk_elems) {
            // a range longer than the scratch was sized for (bootstrap blocks of very many sites: a partial-sum slice
            // is a whole block): the scratch grows to hold it -- the earlier ranges' launches have to be over first
            HIPCHK(hipStreamSynchronize(e->st));
            int rc = ensure_cap(e, &e->qb_chunk, &e->qb_chunk_elems, need);
            if (rc) return rc;
          }
          ngd_launch_qb_range(e->st, g, e->sc, e->PA, lo, std::min<uint64_t>(hi + NGD_KG_TAIL, g.n_kg + NGD_KG_TAIL), e->qb_chunk);
          if (whole_pass)
            ngd_launch_accum_mfma(e->st, g, e->PA + lo * kstride, e->qb_chunk, wsel ? wsel + lo * 4 : nullptr, nullptr, e->d_jobs,
                                  e->n_wg, e->exact_shapes, e->wg_waves, n_ks, piece, hi - lo, 0, 0, slab, e->d_clk, 0,
                                  r > 0 || res > 0);
          else {  // (the kernel indexes the image by absolute k-group: an address below the scratch, formed as an integer)
            // ... so every k-group a launched slice can touch -- its own [kg0, kg1) and the NGD_KG_TAIL k-groups its operand
            // pipeline (the prefetching wavefront included) runs ahead -- must lie inside the scratch as just formed
            const uint64_t first = slice_kg0(ks0), last = std::max(first, slice_kg1(ks0 + n - 1));
            if ((first < lo && first < kg_lim) || last > hi || (hi - lo + NGD_KG_TAIL) * kstride > e->qb_chunk_elems)
              return fail(NGD_E_HIP, "launch_accumulate: internal -- a slice of the range reaches outside the scratch of the second image");
            const double *moved_back = reinterpret_cast<const double *>(reinterpret_cast<uintptr_t>(e->qb_chunk) - lo * kstride * sizeof(double));
            ngd_launch_accum_mfma(e->st, g, e->PA, moved_back, wsel, nullptr, e->d_jobs, e->n_wg, e->exact_shapes, e->wg_waves, n,
                                  per_slice, kg_lim, k_per_slice, w_stride, slab, e->d_clk, ks0);
          }
          ks0 += n;
        }
      }
      break;
    case NGD_KERNEL_EM_TABLE:
      ngd_launch_accum_em_table(e->st, g, e->PA, w, sites_eff, e->sc, e->cfg.pairwise_del, e->em_shape, e->d_tiles64,
                                e->n_tiles64, n_ks, per_slice, slab, e->d_emcnt);
      break;
    default:
      ngd_launch_accum_em(e->st, g, e->PA, w, sites_eff, e->sc, e->cfg.pairwise_del,
                          e->kernel == NGD_KERNEL_EM_FAST, e->d_tiles16, e->n_tiles16, n_ks, per_slice, slab);
  }
  return NGD_OK;
}

// The MFMA kernel met a block whose shape it has no code path for (its sums are NaN): the run fails, loudly.
static int mfma_fault(ngd_engine *e) {
  if (!e->h_clk || !((volatile unsigned long long *)e->h_clk)[2]) return NGD_OK;
  e->h_clk[2] = 0;
  return fail(NGD_E_HIP, "accum_mfma: a block shape the kernel does not list (its sums were set to NaN)");
}

static void read_timing(ngd_engine *e, uint64_t n_eff, uint32_t launches, bool add) {
  if (e->d_emcnt) {  // the stream is idle: counters of the pass(es) since the last read
    unsigned long long c[4] = {0, 0, 0, 0};
    if (hipMemcpy(c, e->d_emcnt, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess) {
      if (!add) e->em_counts[0] = e->em_counts[1] = 0;
      e->em_counts[0] += c[0]; e->em_counts[1] += c[1];
      if (c[3]) e->clk_mhz = (double)c[2] / (double)c[3] * e->wall_khz * 1e-3;
      hipMemsetAsync(e->d_emcnt, 0, sizeof(c), e->st);
    }
  }
  if (e->d_clk && launches) {  // (the stream is idle: the sampling wavefront's stores have landed)
    const unsigned long long c0 = ((volatile unsigned long long *)e->h_clk)[0], c1 = ((volatile unsigned long long *)e->h_clk)[1];
    if (c1) e->clk_mhz = (double)c0 / (double)c1 * e->wall_khz * 1e-3;
  }
  float ms[4] = {0, 0, 0, 0};
  hipEventElapsedTime(&ms[0], e->ev[0], e->ev[4]);
  hipEventElapsedTime(&ms[1], e->ev[1], e->ev[2]);
  hipEventElapsedTime(&ms[2], e->ev[2], e->ev[3]);
  hipEventElapsedTime(&ms[3], e->ev[3], e->ev[4]);
  ngd_timing &t = e->timing;
  if (!add) t = ngd_timing{};
  t.ms_total += ms[0]; t.ms_accum += ms[1]; t.ms_reduce += ms[2]; t.ms_count += ms[3];
  t.pair_sites += e->n_owned_pairs * n_eff;
  t.launches += launches;
}

// single_image = 2 engines on the reference's matrices: the pairs the last reduction noted (sums too small for the
// congruent arithmetic to hold to 1e-9 relative: nearly identical individuals) are recomputed with two-operand arithmetic
// from p recovered out of the image and the side array (fixup.hip).  The stream is idle and *h_fixcount has arrived.
//  * a single matrix (d_sum != NULL): over the sites [0, s_hi) with the per-site weights ws (NULL: none), the sums written
//    over the MFMA pass's;
//  * per-block partial results (d_sum == NULL): the noted pairs' entries of slab_boot, slice by slice -- the caller then
//    forms the replicates again.
// The tolerance is unconditional: EVERY noted pair is recomputed, in launches of bounded size, however many there are
// (round 6; rounds 4-5 gave up on all of them past a budget of ~0.33 s).  More noted pairs than the list holds (fix_cap):
// which ones is then unknown, and every pair of the engine is recomputed.  Where the tiles of a single matrix would cost
// more than the whole matrix in the two-image arithmetic (a data set of clones) it is recomputed that way, in one more
// pass (fixup_by_pass above: 86 ms at cfg 3's size where the tiles take ~0.8 s).  Only a caller who SETS a budget
// (NGD_OPT_FIXUP_WORK != 0) gets the old behaviour: noted work above it is left as the one-image pass computed it and
// ngd_last_fixup() reports the pairs as skipped.
// A pair's slices depend on the number of sites alone (not on how many other pairs were noted), so its recomputed bits do
// not depend on the rest of the data set.
// The whole matrix once more in the two-image arithmetic -- P and Q = score . P formed a range of k-groups at a time from
// the image and SM (layout.hip k_pq_range), K1m over the pair of scratch images range by range, every block adding to its
// plane of the slab (the walk of the single_image = 1 engines, launch_accumulate()) -- then the noting rule once more:
// exactly the pairs it picks take the new sums.  Costs a pass and a half (~70 ms at cfg 3's size) WHATEVER the data, where
// tile by tile a data set of clones costs 0.8 s: fixup_pass() takes this way when its tiles would cost more.
static int fixup_by_pass(ngd_engine *e, const uint32_t *ws, uint64_t s_hi, double *d_sum, const unsigned long long *d_cnt, double thr) {
  const ngd_geom &g = e->g;
  const uint64_t kstride = (uint64_t)g.n_ig * 64;
  const uint64_t kg_lim = std::min<uint64_t>(g.n_kg, (3 * s_hi + 3) / 4);
  // ranges of about 1 GiB per scratch image (two of them), every slice a piece of every range
  const uint64_t span = std::max<uint64_t>(256, ((uint64_t)1 << 30) / (kstride * 8));
  uint64_t n_ranges = 0;
  const uint64_t piece = qb_piece(kg_lim, e->n_ks, span, &n_ranges);
  const uint64_t range_kg = std::min<uint64_t>(kg_lim, piece * e->n_ks);
  const uint64_t need = (range_kg + NGD_KG_TAIL) * kstride;
  int rc = ensure_cap(e, &e->fix_p, &e->cap_fix_p, need);
  if (rc) return rc;
  if ((rc = ensure_cap(e, &e->fix_q, &e->cap_fix_q, need))) return rc;
  if ((rc = ensure_cap(e, &e->d_fixnew, &e->cap_fixnew, ngd_n_pairs(g.n_ind)))) return rc;
  for (uint64_t r = 0; r < n_ranges; r++) {
    const uint64_t lo = std::min<uint64_t>(r * piece * e->n_ks, kg_lim), hi = std::min<uint64_t>(lo + piece * e->n_ks, kg_lim);
    if (hi <= lo) break;
    ngd_launch_pq_range(e->st, g, e->sc, e->PA, e->SM, ws, lo, hi + NGD_KG_TAIL, e->fix_p, e->fix_q);
    ngd_launch_accum_mfma(e->st, g, e->fix_p, e->fix_q, nullptr, nullptr, e->d_jobs, e->n_wg, e->exact_shapes, e->wg_waves, e->n_ks,
                          piece, hi - lo, 0, 0, e->slab, e->d_clk, 0, r > 0);
  }
  ngd_launch_reduce(e->st, g, e->slab, e->n_ks, 1, e->d_tiles, e->n_tiles, e->d_fixnew, nullptr, 0, nullptr, 0.0);
  ngd_launch_fix_merge(e->st, g, e->d_fixnew, d_sum, d_cnt, thr, e->d_tiles, e->n_tiles);
  HIPCHK(hipGetLastError());
  return NGD_OK;
}

// The same for the per-block partial results of a bootstrap job whose blocks are whole k-groups: EVERY entry of the slab is
// formed again by the two-operand arithmetic, the scratch images made for a range of whole slices at a time (in eights: the
// XCD deal of accum_mfma.hip) and handed to the kernel moved back by the range's first k-group, as launch_accumulate() does
// for ngd_config.single_image = 1.  The replicates are then reduced from the slab again (partials_impl).
static int fixup_partials_by_pass(ngd_engine *e, uint64_t s_hi) {
  const ngd_geom &g = e->g;
  const uint64_t kstride = (uint64_t)g.n_ig * 64;
  const uint64_t kg_lim = std::min<uint64_t>(g.n_kg, 3 * s_hi / 4);
  const uint64_t per_slice = e->boot_per_slice;
  const uint32_t n_ks = e->boot_nks;
  if (!per_slice || !n_ks || n_ks % 8) return fail(NGD_E_HIP, "fix-up pass: internal -- the partial results' slices are not in eights");
  const uint64_t span = std::max<uint64_t>(8 * per_slice, ((uint64_t)1 << 30) / (kstride * 8));  // ~1 GiB per scratch image
  auto kg0 = [&](uint64_t ks) { return ks * per_slice; };
  auto kg1 = [&](uint64_t ks) { return std::min<uint64_t>(kg_lim, (ks + 1) * per_slice); };
  for (uint32_t ks0 = 0; ks0 < n_ks;) {
    uint32_t n = 8;
    while (ks0 + n < n_ks && kg1(ks0 + n + 7) - kg0(ks0) <= span && kg0(ks0 + n) < kg_lim) n += 8;
    n = std::min(n, n_ks - ks0);
    const uint64_t lo = std::min<uint64_t>(kg0(ks0), kg_lim), hi = std::max(lo, kg1(ks0 + n - 1));
    const uint64_t need = (hi - lo + NGD_KG_TAIL) * kstride;
    int rc = ensure_cap(e, &e->fix_p, &e->cap_fix_p, need);
    if (rc) return rc;
    if ((rc = ensure_cap(e, &e->fix_q, &e->cap_fix_q, need))) return rc;
    ngd_launch_pq_range(e->st, g, e->sc, e->PA, e->SM, nullptr, lo, std::min<uint64_t>(hi + NGD_KG_TAIL, g.n_kg + NGD_KG_TAIL), e->fix_p, e->fix_q);
    const double *p_back = reinterpret_cast<const double *>(reinterpret_cast<uintptr_t>(e->fix_p) - lo * kstride * sizeof(double));
    const double *q_back = reinterpret_cast<const double *>(reinterpret_cast<uintptr_t>(e->fix_q) - lo * kstride * sizeof(double));
    ngd_launch_accum_mfma(e->st, g, p_back, q_back, nullptr, nullptr, e->d_jobs, e->n_wg, e->exact_shapes, e->wg_waves, n, per_slice,
                          kg_lim, 0, 0, e->slab_boot, e->d_clk, ks0);
    HIPCHK(hipGetLastError());
    ks0 += n;
  }
  return NGD_OK;
}

static int fixup_pass(ngd_engine *e, const uint32_t *ws, uint64_t s_hi, double *d_sum, uint64_t sites_per_slice,
                      uint32_t n_slab_slices, bool *patched, const unsigned long long *d_cnt = nullptr, double thr = 0.0) {
  if (patched) *patched = false;
  const uint32_t n = *(volatile uint32_t *)e->h_fixcount;
  e->fix_info.flagged += n;
  if (!n) return NGD_OK;
  const bool capped = e->opt_fix_work != 0;  // a budget is a caller's explicit leave to skip
  const double budget = (double)e->opt_fix_work;
  const bool all = n > e->fix_cap;  // the list overflowed: which pairs were noted is not known
  const double tile_cost = NGD_FIX_TILE_COST_X10 / 10.0 * (double)s_hi;
  // (the least the pass could cost -- every tile full -- before the list is fetched and sorted)
  if (capped && (all ? (double)e->h_tiles16.size() : (double)((n + 255) / 256)) * tile_cost > budget) {
    e->fix_info.skipped += n;
    return NGD_OK;
  }
  hipEvent_t t0 = e->ev[0], t1 = e->ev[1];  // (the pass's own timings have been read)
  HIPCHK(hipEventRecord(t0, e->st));
  std::vector<ngd_fix_tile> tiles;
  std::vector<unsigned long long> singles;
  if (all) {
    for (const ngd_tile &t16 : e->h_tiles16) {  // (this engine's shard of the pairs)
        const uint32_t ig = t16.ti, jg = t16.tj;
        ngd_fix_tile t{(uint16_t)ig, (uint16_t)jg, 0, {0, 0, 0, 0}};
        for (uint32_t r = 0; r < 16; r++)
          for (uint32_t c = 0; c < 16; c++) {
            const uint64_t i = (uint64_t)ig * 16 + r, j = (uint64_t)jg * 16 + c;
            if (i < j && j < e->g.n_ind) { t.mask[(r * 16 + c) >> 6] |= 1ull << ((r * 16 + c) & 63); t.n++; }
          }
        if (t.n) tiles.push_back(t);
      }
  } else {
    // Nearly identical individuals come in clusters: the noted pairs are grouped by their 16 x 16 tile of individuals on
    // the host (8-byte entries), a tile that holds NGD_FIX_TILE_MIN of them or more is recomputed whole (k_fixup_tile:
    // coalesced, 4 bytes per pair-site), the others pair by pair (k_fixup: ~400)
    std::vector<unsigned long long> list(n);
    HIPCHK(hipMemcpy(list.data(), e->d_fixlist, (size_t)n * 8, hipMemcpyDeviceToHost));  // (the stream is idle: the pass was waited for)
    std::sort(list.begin(), list.end(), [](unsigned long long x, unsigned long long y) {
      const unsigned long long tx = ((x >> 36) << 32) | ((uint32_t)x >> 4), ty = ((y >> 36) << 32) | ((uint32_t)y >> 4);
      return tx != ty ? tx < ty : x < y;
    });
    for (uint32_t k = 0; k < n;) {
      const uint32_t ig = (uint32_t)(list[k] >> 36), jg = (uint32_t)list[k] >> 4;
      uint32_t k1 = k;
      ngd_fix_tile t{(uint16_t)ig, (uint16_t)jg, 0, {0, 0, 0, 0}};
      while (k1 < n && (uint32_t)(list[k1] >> 36) == ig && ((uint32_t)list[k1] >> 4) == jg) {
        const uint32_t bit = ((uint32_t)(list[k1] >> 32) & 15) * 16 + ((uint32_t)list[k1] & 15);
        t.mask[bit >> 6] |= 1ull << (bit & 63);
        k1++;
      }
      t.n = k1 - k;
      if (t.n >= NGD_FIX_TILE_MIN) tiles.push_back(t);
      else singles.insert(singles.end(), list.begin() + k, list.begin() + k1);
      k = k1;
    }
  }
  // what the recomputation costs, in pair-sites (ngd_internal.h)
  if (capped && (double)tiles.size() * tile_cost + (double)singles.size() * (double)s_hi > budget) {
    e->fix_info.skipped += n;
    return NGD_OK;
  }
  // A single matrix whose tiles would cost more than the whole matrix by the two-operand MFMA arithmetic takes that way
  // ([measured] tiles: 6.5e11 pair-sites/s of 256 each; the pass: 6 flop per pair-site at ~55 TF with its ranges' overhead
  // + 80 bytes per (individual, site) to form the scratch images at ~2.4 TB/s)
  if (d_sum && e->kernel == NGD_KERNEL_MFMA && e->exact_shapes == 0 && e->slab) {
    const double t_tiles = ((double)tiles.size() * 256.0 + (double)singles.size() * 60.0) * (double)s_hi / 6.5e11;
    const double t_pass = 6.0 * (double)e->n_owned_pairs * (double)s_hi / 55e12 + 80.0 * (double)e->g.n_pad * (double)s_hi / 2.4e12 + 2e-3;
    if (t_tiles > t_pass) {
      int rc = fixup_by_pass(e, ws, s_hi, d_sum, d_cnt, thr);
      if (rc) return rc;
      HIPCHK(hipEventRecord(t1, e->st));
      HIPCHK(hipStreamSynchronize(e->st));
      if (int rf = mfma_fault(e)) return rf;
      float ms = 0;
      hipEventElapsedTime(&ms, t0, t1);
      e->fix_info.ms += ms;
      e->fix_info.recomputed += all ? e->n_owned_pairs : n;
      e->fix_info.by_pass += 1;
      if (patched) *patched = true;
      return NGD_OK;
    }
  }
  // Per-block partial results (whole k-groups per block): where the noted tiles would cost more than the whole slab again
  // in the two-operand arithmetic, the whole slab it is (round 6; the tiles: 0.8 s for a data set of clones at cfg 3's size)
  if (!d_sum && e->kernel == NGD_KERNEL_MFMA && e->exact_shapes == 0 && e->slab_boot && e->boot_per_slice &&
      e->boot_B % 4 == 0 && (uint64_t)e->boot_per_slice * 4 == sites_per_slice * 3) {
    const double t_tiles = ((double)tiles.size() * 256.0 + (double)singles.size() * 60.0) * (double)s_hi / 6.5e11;
    const double t_pass = 6.0 * (double)e->n_owned_pairs * (double)s_hi / 50e12 + 80.0 * (double)e->g.n_pad * (double)s_hi / 2.4e12 + 2e-3;
    // (tests only, NGD_ENABLE_TEST_HOOKS=1: NGD_TEST_FIX_PARTIALS = "pass" / "tiles" takes the choice away from the estimate)
    const char *forced = (getenv("NGD_ENABLE_TEST_HOOKS") && atoi(getenv("NGD_ENABLE_TEST_HOOKS"))) ? getenv("NGD_TEST_FIX_PARTIALS") : nullptr;
    const bool by_pass = forced ? forced[0] == 'p' : t_tiles > t_pass;
    if (by_pass) {
      int rc = fixup_partials_by_pass(e, s_hi);
      if (rc) return rc;
      HIPCHK(hipEventRecord(t1, e->st));
      HIPCHK(hipStreamSynchronize(e->st));
      if (int rf = mfma_fault(e)) return rf;
      float ms = 0;
      hipEventElapsedTime(&ms, t0, t1);
      e->fix_info.ms += ms;
      e->fix_info.recomputed += all ? e->n_owned_pairs : n;
      e->fix_info.by_pass += 1;
      if (patched) *patched = true;
      return NGD_OK;
    }
  }
  // launches of at most 2^22 workgroups (HIP bounds a launch's threads by 2^32); a pass over per-block partial results has
  // one workgroup per (tile or pair, slab slice)
  const uint64_t max_wg = 1ull << 22;
  if (!d_sum && n_slab_slices > max_wg) return fail(NGD_E_INVALID, "fix-up pass: more slab slices than a launch has workgroups");
  if (!tiles.empty()) {
    int rc = ensure_cap(e, &e->d_fixtiles, &e->cap_fixtiles, tiles.size());
    if (rc) return rc;
    HIPCHK(hipMemcpy(e->d_fixtiles, tiles.data(), tiles.size() * sizeof(ngd_fix_tile), hipMemcpyHostToDevice));
    if (d_sum) {
      // slices of 4096 sites (fewer, longer ones only where NGD_FIX_CAP of them would not cover the sites); as many tiles
      // to a launch as the partial-sum scratch holds (stream order: a launch's scratch is read before the next writes it)
      rc = ensure_cap(e, &e->d_fixtparts, &e->cap_fixtparts, (uint64_t)NGD_FIX_CAP * 256);
      if (rc) return rc;
      const uint64_t sps = std::max<uint64_t>(4096, (s_hi + NGD_FIX_CAP - 1) / NGD_FIX_CAP);
      const uint64_t n_slices = (s_hi + sps - 1) / sps;
      const size_t per = std::max<size_t>(1, NGD_FIX_CAP / n_slices);
      for (size_t off = 0; off < tiles.size(); off += per) {
        const uint32_t m = (uint32_t)std::min<size_t>(per, tiles.size() - off);
        ngd_launch_fixup_tiles(e->st, e->g, e->sc, e->PA, e->SM, ws, e->d_fixtiles + off, m, 0, s_hi, sps, (uint32_t)n_slices, 0,
                               e->d_fixtparts);
        ngd_launch_fixup_tiles_finish(e->st, e->g, e->d_fixtiles + off, m, e->d_fixtparts, (uint32_t)n_slices, d_sum);
      }
    } else {
      const size_t per = (size_t)std::max<uint64_t>(1, max_wg / n_slab_slices);
      for (size_t off = 0; off < tiles.size(); off += per) {
        const uint32_t m = (uint32_t)std::min<size_t>(per, tiles.size() - off);
        ngd_launch_fixup_tiles(e->st, e->g, e->sc, e->PA, e->SM, nullptr, e->d_fixtiles + off, m, 0, s_hi, sites_per_slice,
                               n_slab_slices, 1, e->slab_boot);
      }
    }
  }
  const uint32_t n1 = (uint32_t)singles.size();
  if (n1) HIPCHK(hipMemcpy(e->d_fixlist, singles.data(), (size_t)n1 * 8, hipMemcpyHostToDevice));
  if (n1 && d_sum) {
    const uint64_t sps = std::max<uint64_t>(1024, (s_hi + NGD_FIX_CAP - 1) / NGD_FIX_CAP);
    const uint64_t n_slices = (s_hi + sps - 1) / sps;
    const uint32_t per = (uint32_t)std::max<uint64_t>(1, NGD_FIX_CAP / n_slices);
    for (uint32_t off = 0; off < n1; off += per) {
      const uint32_t m = std::min<uint32_t>(per, n1 - off);
      ngd_launch_fixup(e->st, e->g, e->sc, e->PA, e->SM, ws, e->d_fixlist + off, m, 0, s_hi, sps, (uint32_t)n_slices, 0, e->d_fixparts);
      ngd_launch_fixup_finish(e->st, e->g, e->d_fixlist + off, m, e->d_fixparts, (uint32_t)n_slices, d_sum);
    }
  } else if (n1) {
    const uint32_t per = (uint32_t)std::max<uint64_t>(1, max_wg / n_slab_slices);
    for (uint32_t off = 0; off < n1; off += per) {
      const uint32_t m = std::min<uint32_t>(per, n1 - off);
      ngd_launch_fixup(e->st, e->g, e->sc, e->PA, e->SM, nullptr, e->d_fixlist + off, m, 0, s_hi, sites_per_slice, n_slab_slices, 1,
                       e->slab_boot);
    }
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(t1, e->st));
  HIPCHK(hipStreamSynchronize(e->st));
  float ms = 0;
  hipEventElapsedTime(&ms, t0, t1);
  e->fix_info.ms += ms;
  e->fix_info.recomputed += all ? e->n_owned_pairs : n;
  if (patched) *patched = true;
  return NGD_OK;
}

// One accumulation pass over the resident data set: the full data set (mult == NULL) or one bootstrap
// replicate given as block multiplicities (applied inside the accumulation kernel).
static int pass_impl(ngd_engine *e, const uint32_t *mult, uint32_t mult_max, uint64_t n_blocks, uint64_t block_size,
                     uint64_t n_drawn, double *d_sum, unsigned long long *d_cnt, bool add_timing) {
  const ngd_geom &g = e->g;
  const uint64_t n_pairs = ngd_n_pairs(g.n_ind);
  uint64_t n_eff = g.n_sites;
  const uint32_t *ws = nullptr;
  uint32_t n_planes = 0;
  uint32_t n_list = 0;
  const bool list_pass = mult && e->kernel == NGD_KERNEL_MFMA && !e->single_image;
  HIPCHK(hipEventRecord(e->ev[0], e->st));
  if (mult) {
    n_eff = n_blocks * block_size;
    while (n_planes < 32 && (mult_max >> n_planes)) n_planes++;
    if (!n_planes) n_planes = 1;  // no block drawn (a site range of a larger job): one all-zero plane -- 0 planes means "unweighted"
    if (n_blocks > e->cap_blocks) {
      if (e->d_mult) { hipFree(e->d_mult); e->dev_bytes -= e->cap_blocks * 4; }
      e->d_mult = nullptr; e->cap_blocks = 0;
      int rc = dev_alloc(e, &e->d_mult, n_blocks, false);
      if (rc) return rc;
      e->cap_blocks = n_blocks;
    }
    HIPCHK(hipMemcpyAsync(e->d_mult, mult, n_blocks * 4, hipMemcpyHostToDevice, e->st));
    ngd_launch_weights(e->st, n_blocks, block_size, g.n_sites_pad, e->d_mult, e->d_ws, e->d_wk, e->congruent ? e->sc.d : nullptr);
    if (list_pass) {  // the k-groups this replicate visits at all (about 1/e of the sites are not drawn)
      const uint32_t nb = ngd_kg_count_blocks(g.n_kg);
      if (!e->d_kgl) {
        int rc = dev_alloc(e, &e->d_kgl, g.n_kg + NGD_KG_LIST_PAD, false);
        if (rc) return rc;
        rc = dev_alloc(e, &e->d_kgcnt, (uint64_t)nb + 1, false);
        if (rc) return rc;
      }
      ngd_launch_kg_compact(e->st, e->d_wk, g.n_kg, (uint32_t)g.n_kg, e->d_kgcnt, e->d_kgl);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(&n_list, e->d_kgcnt + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, e->st));
    }
    HIPCHK(hipStreamSynchronize(e->st));  // `mult` is pageable host memory; n_list has arrived
    ws = e->d_ws;
  }
  // pairs outside this engine's shard are returned as 0 / 0; an engine that owns every pair overwrites them all
  // (a device memset moves ~0.15 TB/s: 0.8 ms for the 130 MB of a 65-matrix cfg 5 batch)
  const bool zero_sum = e->cfg.shard_world > 1;
  const bool zero_cnt = zero_sum || e->cfg.pairwise_del;  // k_count adds with integer atomics
  if (zero_sum) HIPCHK(hipMemsetAsync(d_sum, 0, n_pairs * sizeof(double), e->st));
  if (zero_cnt) HIPCHK(hipMemsetAsync(d_cnt, 0, n_pairs * sizeof(unsigned long long), e->st));
  HIPCHK(hipEventRecord(e->ev[1], e->st));
  int rc_acc = NGD_OK;
  if (e->kernel == NGD_KERNEL_STREAM)
    ngd_launch_accum_stream(e->st, g, e->PI, ws, n_eff, e->sc, e->cfg.pairwise_del,
                            e->cfg.shard_world > 1 ? e->d_pairs : nullptr, e->n_owned_pairs, d_sum);
  else if (list_pass)  // slices are equal shares of the list (whole multiples of 4 entries: the deepest operand ring)
    rc_acc = launch_accumulate(e, ws, e->d_kgl, n_eff, e->n_ks, (((uint64_t)n_list + e->n_ks - 1) / e->n_ks + 3) / 4 * 4,
                               n_list, e->slab);
  else if (!mult && e->eager_valid && e->eager_slices) {
    // the leading slices were accumulated beside the load (eager_advance): what is left, behind them
    HIPCHK(hipStreamWaitEvent(e->st, e->ev_eager, 0));
    if (e->eager_slices < e->n_ks) launch_plain_slices(e, e->st, e->eager_slices, e->n_ks - e->eager_slices, false);
    e->eager_valid = false;
    e->eager_slices = 0;
  } else
    rc_acc = launch_accumulate(e, ws, nullptr, n_eff, e->n_ks, e->per_slice, g.n_kg, e->slab);
  if (rc_acc) return rc_acc;
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->ev[2], e->st));
  // (without --pairwise_del the reduction writes the counts too: every pair visits the same number of sites)
  const bool cnt_in_reduce = e->kernel != NGD_KERNEL_STREAM && !e->cfg.pairwise_del;
  const bool fix = e->SM != nullptr;  // (a congruent single-image MFMA engine on one of the reference's matrices)
  const ngd_fix_flags ff{e->d_fixlist, e->d_fixcount, e->d_fixseen, e->fix_cap};
  // (--pairwise_del: the pairs that want the fix-up are noted once their valid-site counts are known, below)
  const bool fix_in_reduce = fix && !e->cfg.pairwise_del;
  if (fix) HIPCHK(hipMemsetAsync(e->d_fixcount, 0, sizeof(uint32_t), e->st));
  if (e->kernel != NGD_KERNEL_STREAM)
    ngd_launch_reduce(e->st, g, e->slab, e->n_ks, 1, e->d_tiles, e->n_tiles, d_sum, cnt_in_reduce ? d_cnt : nullptr,
                      mult ? n_drawn : n_eff, fix_in_reduce ? &ff : nullptr, NGD_FIX_MEAN * (double)(mult ? n_drawn : n_eff));
  if (fix_in_reduce) HIPCHK(hipMemcpyAsync(e->h_fixcount, e->d_fixcount, sizeof(uint32_t), hipMemcpyDeviceToHost, e->st));
  HIPCHK(hipEventRecord(e->ev[3], e->st));
  if (e->cfg.pairwise_del) {
    if (ws) ngd_launch_planes(e->st, ws, g.n_sites, g.n_words, n_planes, e->planes);
    ngd_launch_count(e->st, g, e->mask, e->planes, ws ? n_planes : 0, e->d_tiles, e->n_tiles, d_cnt);
    if (fix) {
      ngd_launch_fix_flag(e->st, g, d_sum, d_cnt, 1, e->d_tiles, e->n_tiles, ff);
      HIPCHK(hipMemcpyAsync(e->h_fixcount, e->d_fixcount, sizeof(uint32_t), hipMemcpyDeviceToHost, e->st));
    }
  } else if (!cnt_in_reduce) {
    ngd_launch_fill_cnt(e->st, g, e->d_tiles, e->n_tiles, mult ? n_drawn : n_eff, nullptr, 1, d_cnt);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->ev[4], e->st));
  HIPCHK(hipStreamSynchronize(e->st));
  read_timing(e, n_eff, 1, add_timing);
  if (int rc = mfma_fault(e)) return rc;
  if (fix)
    return fixup_pass(e, ws, n_eff, d_sum, 0, 0, nullptr, e->cfg.pairwise_del ? d_cnt : nullptr,
                      NGD_FIX_MEAN * (double)(mult ? n_drawn : n_eff));
  return NGD_OK;
}

// Bootstrap by per-block partials: every site's contribution is independent of the
// replicate, so sum_rep = SUM_b multiplicity_rep[b] * S_b with S_b the block's partial sum (and the same
// for the valid-site counts).  One accumulation pass fills S_b; replicates are then weighted reductions of
// the partials, up to 32 per pass over them.  MFMA slices are whole k-groups of 4 contraction indices, so
// blocks must be multiples of 4 sites there.  *feasible = false: the caller falls back to pass_impl().
// ---- a job's matrices leaving the device while later ones are still being reduced (ngd_run_job_dist) ----
static bool out_trace() {
  static const bool on = getenv("NGD_TRACE_OUT") != nullptr;
  return on;
}
static double out_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int out_event(ngd_engine *e, hipEvent_t *ev) {
  auto &o = e->out;
  if (o.n_used == o.pool.size()) {
    hipEvent_t v;
    HIPCHK(hipEventCreateWithFlags(&v, hipEventDisableTiming));
    o.pool.push_back(v);
  }
  *ev = o.pool[o.n_used++];
  return NGD_OK;
}

// The copies of matrices [queued, m_hi) of d_bsum (--pairwise_del: and d_bcnt) are queued behind whatever the engine's stream
// holds NOW (the first event of a call is the first gate: partials_impl waits for it before it wakes the host's threads):
// chunks of about 8 MiB with an event each, alternating between two copy streams (a chunk's set-up and its event then hide
// behind the other stream's transfer: 49 -> 55 GB/s at cfg 5), tapering towards the job's end -- a chunk is at most a quarter
// of what is left -- because what the host's threads still have to do once the last byte has landed is the last chunk's cells.
// ([measured, round 6] a kernel pushing the results into the pinned buffers 64 KiB at a time with a flag in host memory
// behind every piece -- no events, a smooth arrival -- was no faster, 50 GB/s, and slowed the reductions it ran beside.)
static int out_queue(ngd_engine *e, uint32_t m_hi) {
  auto &o = e->out;
  if (!o.on || m_hi <= o.queued) return NGD_OK;
  const uint64_t n_pairs = ngd_n_pairs(e->g.n_ind);
  hipEvent_t gate;
  if (int rc = out_event(e, &gate)) return rc;
  HIPCHK(hipEventRecord(gate, e->st));
  HIPCHK(hipStreamWaitEvent(o.st, gate, 0));
  HIPCHK(hipStreamWaitEvent(o.st2, gate, 0));
  const uint32_t step = (uint32_t)std::min<uint64_t>(1u << 20, std::max<uint64_t>(1, (8ull << 20) / std::max<uint64_t>(1, n_pairs * 8)));
  for (uint32_t a = o.queued, b; a < m_hi; a = b) {
    b = std::min(m_hi, a + std::min(step, std::max(1u, (o.n_mat - a + 3) / 4)));
    hipStream_t st = (o.n_chunk_seq++ & 1) ? o.st2 : o.st;
    HIPCHK(hipMemcpyAsync(o.h_sum + (uint64_t)a * n_pairs, e->d_bsum + (uint64_t)a * n_pairs, (uint64_t)(b - a) * n_pairs * sizeof(double),
                          hipMemcpyDeviceToHost, st));
    if (o.pdel)
      HIPCHK(hipMemcpyAsync(o.h_cnt + (uint64_t)a * n_pairs, e->d_bcnt + (uint64_t)a * n_pairs, (uint64_t)(b - a) * n_pairs * sizeof(uint64_t),
                            hipMemcpyDeviceToHost, st));
    hipEvent_t ev;
    if (int rc = out_event(e, &ev)) return rc;
    HIPCHK(hipEventRecord(ev, st));
    o.chunks.emplace_back(ev, b);
  }
  o.queued = m_hi;
  return NGD_OK;
}

static void out_declare(ngd_engine *e, uint64_t cells) {
  auto &o = e->out;
  std::atomic_thread_fence(std::memory_order_release);
  o.landed = cells;
  if (out_trace()) fprintf(stderr, "[out] %.2f matrices landed +%.3f\n", (double)cells / (double)ngd_n_pairs(e->g.n_ind), out_now() - o.t0);
}

// Declares landed whatever has arrived since the last look (never waits)
static int out_advance(ngd_engine *e) {
  auto &o = e->out;
  while (o.n_landed < o.chunks.size()) {
    const hipError_t q = hipEventQuery(o.chunks[o.n_landed].first);
    if (q == hipErrorNotReady) break;
    HIPCHK(q);
    out_declare(e, (uint64_t)o.chunks[o.n_landed].second * ngd_n_pairs(e->g.n_ind));
    o.n_landed++;
  }
  return NGD_OK;
}

// What has been queued carries sums that a fix-up pass is about to replace
static int out_join(ngd_engine *e, int rc);
static int out_requeue(ngd_engine *e) {
  auto &o = e->out;
  if (!o.on) return NGD_OK;
  HIPCHK(hipStreamSynchronize(o.st));
  HIPCHK(hipStreamSynchronize(o.st2));
  // pieces may have been declared landed already (partials_impl lands what arrives while the last groups are reduced): the
  // host's threads are let through the stale cells and start again from nothing once the matrices have been reduced again
  if (int rc = out_join(e, NGD_OK)) return rc;
  o.on = true;
  o.landed = 0;
  o.n_landed = 0;
  o.queued = 0;
  o.chunks.clear();
  return NGD_OK;
}

static void out_start_finisher(ngd_engine *e) {
  auto &o = e->out;
  if (!o.on || o.finisher.joinable()) return;
  o.finisher_rc = 0;
  if (o.tot_sites) o.cnt_mat.assign(o.n_mat, o.tot_sites);
  const uint64_t n_pairs = ngd_n_pairs(e->g.n_ind);
  o.finisher = std::thread([e, n_pairs]() {
    auto &q = e->out;
    q.finisher_rc = ngd_finish_matrices_stream(q.h_sum, q.pdel ? q.h_cnt : nullptr, q.pdel ? nullptr : q.cnt_mat.data(), q.n_mat, n_pairs,
                                               q.evol_model, q.dist, &q.landed);
  });
}

// The end of a streamed call, good or bad: the host's threads are let through whatever is left (after a failure: over
// cells nobody will read) and joined.
static int out_join(ngd_engine *e, int rc) {
  auto &o = e->out;
  if (o.finisher.joinable()) {
    std::atomic_thread_fence(std::memory_order_release);
    o.landed = (uint64_t)o.n_mat * ngd_n_pairs(e->g.n_ind);
    o.finisher.join();
    if (!rc && o.finisher_rc) rc = fail(o.finisher_rc, "ngd_run_*_dist: the tail of gen_dist() failed");
  }
  o.on = false;
  return rc;
}

static int out_land_all(ngd_engine *e) {
  auto &o = e->out;
  for (; o.n_landed < o.chunks.size(); o.n_landed++) {
    HIPCHK(hipEventSynchronize(o.chunks[o.n_landed].first));
    out_declare(e, (uint64_t)o.chunks[o.n_landed].second * ngd_n_pairs(e->g.n_ind));
  }
  return NGD_OK;
}

static int out_land(ngd_engine *e) {
  auto &o = e->out;
  int rc = out_queue(e, o.n_mat);
  if (!rc) {
    out_start_finisher(e);
    rc = out_land_all(e);
  }
  rc = out_join(e, rc);
  if (out_trace()) fprintf(stderr, "[out] tail joined +%.3f\n", out_now() - o.t0);
  return rc;
}

static int partials_impl(ngd_engine *e, const uint32_t *mult /*[n_rep][n_blocks]*/, const unsigned long long *drawn,
                         uint32_t n_rep, uint64_t n_blocks, uint64_t block_size, double *d_sum,
                         unsigned long long *d_cnt, bool *feasible) {
  const ngd_geom &g = e->g;
  const uint64_t n_pairs = ngd_n_pairs(g.n_ind);
  const uint64_t n_eff = n_blocks * block_size;
  const uint64_t plane = (uint64_t)g.n_pad * g.n_pad;
  const bool mfma = e->kernel == NGD_KERNEL_MFMA;
  const bool pdel = e->cfg.pairwise_del != 0;
  *feasible = false;
  if (e->kernel == NGD_KERNEL_STREAM || !e->opt_boot_partials) return NGD_OK;
  // MFMA slices are whole k-groups of 4 contraction indices; a block of B sites is 3 B of them.  Blocks that are not
  // whole k-groups become slices of every k-group they touch, the shared first / last k-group masked per slice.
  const bool unaligned = mfma && block_size % 4 != 0;
  if (unaligned && !e->opt_boot_unaligned) return NGD_OK;
  if (n_blocks >= (1ull << 31)) return NGD_OK;
  // split large blocks so that there are enough workgroups; slices of one block share its weight
  const uint64_t unit = mfma ? 3 * block_size / 4 : block_size;  // k-groups or sites per block
  const bool cached = e->boot_B == block_size && e->boot_blocks == n_blocks;
  uint64_t sub = 1, nks = 0;
  if (cached) {
    sub = e->boot_sub;
    nks = e->boot_nks;
  } else {
    const uint32_t tiles_n = mfma ? std::max(1u, e->n_wg / (e->exact_shapes && e->exact_shapes < 3 ? 4 : 1))
                                  : e->kernel == NGD_KERNEL_EM_TABLE ? e->n_tiles64 : e->n_tiles16;
    const uint64_t want = e->opt_boot_wg;
    while (!unaligned && tiles_n && (uint64_t)tiles_n * n_blocks * sub < want && unit % (sub * 2) == 0 &&
           unit / (sub * 2) >= 32)
      sub *= 2;
    nks = n_blocks * sub;
    if (mfma) nks = (nks + 7) / 8 * 8;  // the XCD deal of accum_mfma.hip
  }
  if (nks >= (1ull << 31)) return NGD_OK;
  const uint64_t elems = nks * plane, c_elems = pdel ? n_blocks * plane : 0;
  const bool c_cached = !pdel || (e->cnt_B == block_size && e->cnt_blocks == n_blocks);
  if (!cached || !c_cached) {
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const uint64_t need = elems * 8 + c_elems * 4;
    const uint64_t have = e->slab_boot_elems * 8 + e->cnt_boot_elems * 4;
    // default budget: most of what the device has left -- one pass over a slab of tens of GB still beats
    // hundreds of accumulation passes
    const uint64_t budget = e->opt_boot_max_bytes ? e->opt_boot_max_bytes : (uint64_t)((free_b + have) / 100 * 85);
    if (need > budget) return NGD_OK;
    if ((elems > e->slab_boot_elems || c_elems > e->cnt_boot_elems) && need + (1ull << 30) > free_b + have)
      return NGD_OK;
    const double alloc_ms = need > have ? (double)(need - have) * 12e-9 : 0.0;
    if (alloc_ms > 20.0 && e->opt_boot_partials < 2) {
      // what this call costs without the partials: a list-driven pass per replicate (MFMA, ~3/4 of a pass)
      // or a batch pass per 16 replicates (EM); rates are the measured ones of DESIGN.md section 6
      const double ps = (double)e->n_owned_pairs * (double)n_eff;
      const bool table = e->kernel == NGD_KERNEL_EM_TABLE;
      const double pass_ms = mfma ? ps / 1.05e10 : table ? ps / 1.9e8 : e->kernel == NGD_KERNEL_EM_FAST ? ps / 7.5e7 : ps / 3.8e6;
      // (the table-driven kernel's spilled-terms plan, em_spill_impl: ONE EM pass + a contraction whatever the replicate
      // count -- [measured, round 6] without this term the engine bought an 84 GB slab for a single job of 10 000 blocks of
      // 10 sites, 3 ms on a device nobody has used and 5.8 s on one that has just been busy)
      const bool spill = table && e->em_shape == 0 && e->opt_em_spill && n_rep >= (e->opt_em_spill == 2 ? 2u : 3u);
      const double alt_ms = mfma ? 0.75 * pass_ms * n_rep
                            : spill ? 1.1 * pass_ms
                            : table ? std::min(0.65 * pass_ms * n_rep, 2.9 * pass_ms * ((n_rep + 15) / 16))
                                    : 1.1 * pass_ms * ((n_rep + 15) / 16);
      if (e->rent_B != block_size || e->rent_blocks != n_blocks) {
        e->rent_B = block_size; e->rent_blocks = n_blocks; e->rent_ms = 0;
      }
      if (e->rent_ms + alt_ms < alloc_ms) {
        e->rent_ms += alt_ms;
        return NGD_OK;
      }
    }
  }
  *feasible = true;
  if (e->out.on && out_trace()) fprintf(stderr, "[out] plan settled %.3f ms into the call\n", out_now() - e->out.t_call);

  HIPCHK(hipEventRecord(e->ev[0], e->st));
  uint32_t launches = 0;
  HIPCHK(hipEventRecord(e->ev[1], e->st));
  if (!cached) {
    e->boot_B = 0;
    int rc = ensure_cap(e, &e->slab_boot, &e->slab_boot_elems, elems);
    if (rc) return rc;
    e->boot_nks = (uint32_t)nks;
    e->boot_sub = (uint32_t)sub;
    e->boot_per_slice = unit / sub;
    if (unaligned) {
      const uint32_t w_stride = (uint32_t)((3 * block_size + 3) / 4 + 1 + NGD_KG_TAIL);
      rc = ensure_cap(e, &e->d_wslice, &e->cap_wslice, (uint64_t)e->boot_nks * w_stride * 4);
      if (rc) return rc;
      ngd_launch_slice_weights(e->st, e->boot_nks, w_stride, 3 * block_size, 3 * n_eff, e->d_wslice, e->congruent ? e->sc.d : nullptr);
      rc = launch_accumulate(e, nullptr, nullptr, n_eff, e->boot_nks, 0, (3 * n_eff + 3) / 4, e->slab_boot, 3 * block_size,
                             w_stride);
    } else
      rc = launch_accumulate(e, nullptr, nullptr, n_eff, e->boot_nks, e->boot_per_slice, mfma ? 3 * n_eff / 4 : 0, e->slab_boot);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    e->boot_B = block_size;
    e->boot_blocks = n_blocks;
    launches = 1;
  }
  HIPCHK(hipEventRecord(e->ev[2], e->st));

  // W[slice][r]: slice-major, so that the replicates of one pass read their weights of a slice together
  const uint32_t stride = (n_rep + ngd_reduce_chunk(n_rep) - 1) / ngd_reduce_chunk(n_rep) * ngd_reduce_chunk(n_rep);
  const uint64_t n_slices = n_blocks * sub;  // the slab's padding slices (MFMA deal) are never read
  std::vector<double> W(n_slices * stride, 0.0);
  for (uint32_t r = 0; r < n_rep; r++)
    for (uint64_t b = 0; b < n_blocks; b++) {
      const double m = (double)mult[(uint64_t)r * n_blocks + b];
      if (m != 0.0)
        for (uint64_t q = 0; q < sub; q++) W[(b * sub + q) * stride + r] = m;
    }
  int rc = ensure_cap(e, &e->d_W, &e->cap_W, W.size());
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(e->d_W, W.data(), W.size() * 8, hipMemcpyHostToDevice, e->st));
  if (e->cfg.shard_world > 1) {  // (the weighted reductions write every pair this engine owns, sums and counts)
    HIPCHK(hipMemsetAsync(d_sum, 0, (uint64_t)n_rep * n_pairs * sizeof(double), e->st));
    HIPCHK(hipMemsetAsync(d_cnt, 0, (uint64_t)n_rep * n_pairs * sizeof(unsigned long long), e->st));
  }
  const bool fix = e->SM != nullptr && mfma;  // (see pass_impl)
  const ngd_fix_flags ff{e->d_fixlist, e->d_fixcount, e->d_fixseen, e->fix_cap};
  std::vector<double> thr;
  if (fix) {  // a pair is noted if its sum in ANY matrix is below NGD_FIX_MEAN x the sites that matrix visits
    thr.resize(n_rep);
    for (uint32_t r = 0; r < n_rep; r++) thr[r] = NGD_FIX_MEAN * (double)drawn[r];
    rc = ensure_cap(e, &e->d_fixthr, &e->cap_fixthr, (uint64_t)n_rep);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(e->d_fixthr, thr.data(), (uint64_t)n_rep * 8, hipMemcpyHostToDevice, e->st));
    HIPCHK(hipMemsetAsync(e->d_fixcount, 0, sizeof(uint32_t), e->st));
    HIPCHK(hipMemsetAsync(e->d_fixseen, 0, (n_pairs / 32 + 1) * sizeof(uint32_t), e->st));
  }
  const bool fix_in_reduce = fix && !pdel;  // (--pairwise_del: noted once the counts are known, below)
  std::vector<uint32_t> M;
  // ngd_run_*_dist, the job's first matrix at the head of d_bsum: a group of replicates is reduced by a launch of its own
  // and its copy to the host queued behind it, so that the copies run beside the later groups' reductions
  const bool stream_out = e->out.on && d_sum == e->d_bsum && d_cnt == e->d_bcnt && e->out.queued == 0;
  if (stream_out) {
    // the counts' inputs first: they do not depend on the sums
    if (pdel) {
      if (!c_cached) {
        e->cnt_B = 0;
        rc = ensure_cap(e, &e->cnt_boot, &e->cnt_boot_elems, c_elems);
        if (rc) return rc;
        ngd_launch_count_blocks(e->st, g, e->mask, block_size, (uint32_t)n_blocks, e->d_tiles16, e->n_tiles16, e->cnt_boot);
        HIPCHK(hipGetLastError());
        e->cnt_B = block_size;
        e->cnt_blocks = n_blocks;
      }
      M.assign(n_blocks * stride, 0u);
      for (uint32_t r = 0; r < n_rep; r++)
        for (uint64_t b = 0; b < n_blocks; b++) M[b * stride + r] = mult[(uint64_t)r * n_blocks + b];
      rc = ensure_cap(e, &e->d_M, &e->cap_M, M.size());
      if (rc) return rc;
      HIPCHK(hipMemcpyAsync(e->d_M, M.data(), M.size() * 4, hipMemcpyHostToDevice, e->st));
    } else {
      rc = ensure_cap(e, &e->d_drawn, &e->cap_drawn, (uint64_t)n_rep);
      if (rc) return rc;
      HIPCHK(hipMemcpyAsync(e->d_drawn, drawn, (uint64_t)n_rep * 8, hipMemcpyHostToDevice, e->st));
      ngd_launch_fill_cnt(e->st, g, e->d_tiles, e->n_tiles, 0, e->d_drawn, n_rep, d_cnt);
    }
    const uint32_t rb = ngd_reduce_chunk(n_rep);
    for (uint32_t r0 = 0; r0 < n_rep; r0 += rb) {
      const uint32_t n = std::min(rb, n_rep - r0);
      ngd_launch_reduce_w(e->st, g, e->slab_boot, (uint32_t)n_slices, e->d_W + r0, stride, n, e->d_tiles, e->n_tiles,
                          d_sum + (uint64_t)r0 * n_pairs, fix_in_reduce ? &ff : nullptr, e->d_fixthr ? e->d_fixthr + r0 : nullptr, rb);
      if (r0 + rb >= n_rep) HIPCHK(hipEventRecord(e->ev[3], e->st));
      if (pdel)
        ngd_launch_reduce_c(e->st, g, e->cnt_boot, (uint32_t)n_blocks, e->d_M + r0, stride, n, e->d_tiles, e->n_tiles,
                            d_cnt + (uint64_t)r0 * n_pairs, rb);
      HIPCHK(hipGetLastError());
      if ((rc = out_queue(e, r0 + n))) return rc;
    }
    if (fix && pdel) ngd_launch_fix_flag(e->st, g, d_sum, d_cnt, n_rep, e->d_tiles, e->n_tiles, ff);
    if (fix) HIPCHK(hipMemcpyAsync(e->h_fixcount, e->d_fixcount, sizeof(uint32_t), hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[4], e->st));
    // the host's threads are woken once the first group has been reduced: its first chunk is about to land
    const double t_q = out_now();
    HIPCHK(hipEventSynchronize(e->out.pool[0]));
    const double t_g = out_now();
    out_start_finisher(e);
    if (out_trace()) fprintf(stderr, "[out] queued %.3f ms into the call, first group reduced +%.3f ms, finisher started +%.3f\n", t_q - e->out.t_call, t_g - t_q, out_now() - t_q);
    e->out.t0 = t_q;
    // ... and chunks are declared landed as they arrive while the later groups are still being reduced -- before it is known
    // whether the fix-up pass will patch the partial results (it rarely does: then every matrix is copied and finished again)
    for (;;) {
      const hipError_t q = hipEventQuery(e->ev[4]);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) HIPCHK(q);
      if ((rc = out_advance(e))) return rc;
      __builtin_ia32_pause();
    }
  } else {
    ngd_launch_reduce_w(e->st, g, e->slab_boot, (uint32_t)n_slices, e->d_W, stride, n_rep, e->d_tiles, e->n_tiles, d_sum,
                        fix_in_reduce ? &ff : nullptr, e->d_fixthr);
    if (fix_in_reduce) HIPCHK(hipMemcpyAsync(e->h_fixcount, e->d_fixcount, sizeof(uint32_t), hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[3], e->st));

    if (pdel) {
      if (!c_cached) {
        e->cnt_B = 0;
        rc = ensure_cap(e, &e->cnt_boot, &e->cnt_boot_elems, c_elems);
        if (rc) return rc;
        ngd_launch_count_blocks(e->st, g, e->mask, block_size, (uint32_t)n_blocks, e->d_tiles16, e->n_tiles16, e->cnt_boot);
        HIPCHK(hipGetLastError());
        e->cnt_B = block_size;
        e->cnt_blocks = n_blocks;
      }
      M.assign(n_blocks * stride, 0u);
      for (uint32_t r = 0; r < n_rep; r++)
        for (uint64_t b = 0; b < n_blocks; b++) M[b * stride + r] = mult[(uint64_t)r * n_blocks + b];
      rc = ensure_cap(e, &e->d_M, &e->cap_M, M.size());
      if (rc) return rc;
      HIPCHK(hipMemcpyAsync(e->d_M, M.data(), M.size() * 4, hipMemcpyHostToDevice, e->st));
      ngd_launch_reduce_c(e->st, g, e->cnt_boot, (uint32_t)n_blocks, e->d_M, stride, n_rep, e->d_tiles, e->n_tiles, d_cnt);
      if (fix) {
        ngd_launch_fix_flag(e->st, g, d_sum, d_cnt, n_rep, e->d_tiles, e->n_tiles, ff);
        HIPCHK(hipMemcpyAsync(e->h_fixcount, e->d_fixcount, sizeof(uint32_t), hipMemcpyDeviceToHost, e->st));
      }
    } else {
      rc = ensure_cap(e, &e->d_drawn, &e->cap_drawn, (uint64_t)n_rep);
      if (rc) return rc;
      HIPCHK(hipMemcpyAsync(e->d_drawn, drawn, (uint64_t)n_rep * 8, hipMemcpyHostToDevice, e->st));
      ngd_launch_fill_cnt(e->st, g, e->d_tiles, e->n_tiles, 0, e->d_drawn, n_rep, d_cnt);
    }

  }
  HIPCHK(hipGetLastError());
  if (!stream_out) HIPCHK(hipEventRecord(e->ev[4], e->st));
  HIPCHK(hipStreamSynchronize(e->st));  // W, M, drawn are host temporaries
  if (stream_out && out_trace()) fprintf(stderr, "[out] engine stream drained +%.3f\n", out_now() - e->out.t0);
  read_timing(e, n_eff, launches, false);
  if ((rc = mfma_fault(e))) return rc;
  if (fix) {  // the noted pairs' partial results exactly, then the replicates again from the patched slab
    bool patched = false;
    rc = fixup_pass(e, nullptr, n_eff, nullptr, block_size / sub, (uint32_t)n_slices, &patched);
    if (rc) return rc;
    if (patched) {
      if (stream_out && (rc = out_requeue(e))) return rc;  // (what has been copied so far: sums from before the patch)
      ngd_launch_reduce_w(e->st, g, e->slab_boot, (uint32_t)n_slices, e->d_W, stride, n_rep, e->d_tiles, e->n_tiles, d_sum);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(e->st));
    }
  }
  return NGD_OK;
}

// EM path when the blocks are too many for per-block partials (e.g. the reference's default block size 1):
// the EM of a (pair, site) does not depend on the replicate, so up to 16 replicates share ONE accumulation pass
// (accum_em.hip k_accum_em_batch) -- and, with lead_full, so does the full-data matrix, as the row whose weight
// is 1 on every site.  Outputs: [lead_full + n_rep][n_pairs]; a replicate's bits equal the one-replicate pass's.
static int em_batch_impl(ngd_engine *e, const uint32_t *mult, const uint32_t *mult_max, const unsigned long long *drawn,
                         uint32_t n_rep, bool lead_full, uint64_t n_blocks, uint64_t block_size, double *d_sum,
                         unsigned long long *d_cnt, bool add_timing) {
  const ngd_geom &g = e->g;
  const uint64_t n_pairs = ngd_n_pairs(g.n_ind), plane = (uint64_t)g.n_pad * g.n_pad;
  const uint64_t n_eff = n_blocks * block_size;
  const uint32_t n_mat = n_rep + (lead_full ? 1u : 0u);
  const bool fast = e->kernel != NGD_KERNEL_EM_FAITHFUL;
  // the table-driven kernel's own batch form (its default shape): 8 matrices per pass, one workgroup per CU (the 8 x 8
  // accumulators of a wavefront take the registers of a second one)
  const bool table = e->kernel == NGD_KERNEL_EM_TABLE && e->em_shape == 0;
  const uint32_t per_pass = table ? 8 : 16;
  // slices of the per-pair batch kernel: the engine's own when that is its kernel, else (table-driven engine in
  // another shape) what ngd_create() would have picked for it
  uint32_t b_ks = e->n_ks;
  uint64_t b_per = e->per_slice;
  if (table) {
    // the plain pass's own slices: a matrix then adds up in the same order from either (same bits)
  } else if (e->kernel == NGD_KERNEL_EM_TABLE) {
    uint64_t ks = e->n_tiles16 ? (4096 + e->n_tiles16 - 1) / e->n_tiles16 : 1;
    ks = std::max<uint64_t>(1, std::min(ks, std::max<uint64_t>(1, g.n_sites / 256)));
    b_ks = (uint32_t)ks;
    b_per = (g.n_sites + ks - 1) / ks;
  }
  if (e->cfg.shard_world > 1) HIPCHK(hipMemsetAsync(d_sum, 0, (uint64_t)n_mat * n_pairs * sizeof(double), e->st));
  if (e->cfg.shard_world > 1 || e->cfg.pairwise_del)  // k_count adds with integer atomics
    HIPCHK(hipMemsetAsync(d_cnt, 0, (uint64_t)n_mat * n_pairs * sizeof(unsigned long long), e->st));
  e->boot_B = 0;  // the partial-sum slab is re-used as this pass's scratch
  for (uint32_t c0 = 0; c0 < n_mat; c0 += per_pass) {
    const uint32_t nr = std::min(per_pass, n_mat - c0);
    const int rb = nr <= 4 ? 4 : nr <= 8 ? 8 : 16;
    const bool lead = lead_full && c0 == 0;
    const uint32_t q0 = c0 - ((lead_full && c0 > 0) ? 1u : 0u);  // first replicate of the chunk
    const uint32_t nq = nr - (lead ? 1u : 0u);                     // replicates in the chunk
    if (e->opt_boot_max_bytes && (uint64_t)b_ks * rb * plane * 8 > e->opt_boot_max_bytes)  // the caller's scratch budget
      return fail(NGD_E_NOMEM, "EM batch pass: result planes exceed NGD_OPT_BOOT_MAX_BYTES");
    const uint64_t want = (uint64_t)b_ks * rb * plane;
    if (e->em_batch_nofit_elems && want >= e->em_batch_nofit_elems && want > e->slab_boot_elems)
      return fail(NGD_E_NOMEM, "EM batch pass: result planes of this size did not fit the device before");
    int rc = ensure_cap(e, &e->slab_boot, &e->slab_boot_elems, want);
    if (rc == NGD_E_NOMEM && !e->opt_boot_max_bytes) e->em_batch_nofit_elems = want;  // the device's verdict: remembered
    if (rc) return rc;
    rc = ensure_cap(e, &e->d_W, &e->cap_W, g.n_sites * (uint64_t)rb);
    if (rc) return rc;
    rc = ensure_cap(e, &e->d_M, &e->cap_M, std::max<uint64_t>(1, (uint64_t)nq * n_blocks));
    if (rc) return rc;
    HIPCHK(hipEventRecord(e->ev[0], e->st));
    if (nq) HIPCHK(hipMemcpyAsync(e->d_M, mult + (uint64_t)q0 * n_blocks, (uint64_t)nq * n_blocks * 4, hipMemcpyHostToDevice, e->st));
    ngd_launch_weights_batch(e->st, e->d_M, nq, (uint32_t)rb, lead ? 1 : 0, n_blocks, block_size, g.n_sites, g.n_sites,
                             e->d_W);
    HIPCHK(hipEventRecord(e->ev[1], e->st));
    if (table)
      ngd_launch_accum_em_table_batch(e->st, g, e->PA, e->d_W, rb, lead ? g.n_sites : n_eff, e->sc, e->cfg.pairwise_del,
                                      e->d_tiles64, e->n_tiles64, b_ks, b_per, e->slab_boot, e->d_emcnt);
    else
      ngd_launch_accum_em_batch(e->st, g, e->PA, e->d_W, rb, lead ? g.n_sites : n_eff, e->sc, e->cfg.pairwise_del, fast,
                                e->d_tiles16, e->n_tiles16, b_ks, b_per, e->slab_boot);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[2], e->st));
    for (uint32_t r = 0; r < nr; r++)
      ngd_launch_reduce(e->st, g, e->slab_boot + (uint64_t)r * plane, b_ks, (uint32_t)rb, e->d_tiles, e->n_tiles,
                        d_sum + (uint64_t)(c0 + r) * n_pairs);
    HIPCHK(hipEventRecord(e->ev[3], e->st));
    for (uint32_t r = 0; r < nr; r++) {
      unsigned long long *cnt_r = d_cnt + (uint64_t)(c0 + r) * n_pairs;
      const bool is_lead = lead && r == 0;
      const uint32_t q = q0 + r - (lead ? 1u : 0u);
      if (!e->cfg.pairwise_del) {
        ngd_launch_fill_cnt(e->st, g, e->d_tiles, e->n_tiles, is_lead ? g.n_sites : drawn[q], nullptr, 1, cnt_r);
      } else if (is_lead) {
        ngd_launch_count(e->st, g, e->mask, e->planes, 0, e->d_tiles, e->n_tiles, cnt_r);
      } else {
        uint32_t n_planes = 0;
        while (n_planes < 32 && (mult_max[q] >> n_planes)) n_planes++;
        if (!n_planes) n_planes = 1;  // (as in pass_impl: a replicate that drew none of these blocks counts 0 sites)
        ngd_launch_weights(e->st, n_blocks, block_size, g.n_sites_pad, e->d_M + (uint64_t)(q - q0) * n_blocks, e->d_ws,
                           nullptr);
        ngd_launch_planes(e->st, e->d_ws, g.n_sites, g.n_words, n_planes, e->planes);
        ngd_launch_count(e->st, g, e->mask, e->planes, n_planes, e->d_tiles, e->n_tiles, cnt_r);
      }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[4], e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    read_timing(e, lead ? g.n_sites : n_eff, 1, add_timing || c0 > 0);
  }
  return NGD_OK;
}

// EM path, many matrices, blocks too small for per-block partials: ONE pass of the table-driven EM kernel writes the
// per-(pair, unit of sites) terms of a chunk of sites (they do not depend on the replicate), one FP64 MFMA contraction
// adds the chunk to the running sums of every matrix of the job (contract_mfma.hip).  A unit is q consecutive sites of
// one bootstrap block (q = the block size or its largest divisor up to 64): every matrix weights them alike, so their
// terms are added up before they leave the EM kernel -- the bytes written and read, and the flops of the contraction,
// are those of n_sites / q.  The chunk is as many units as the scratch budget holds (NGD_OPT_EM_SPILL_BYTES).
// Outputs: [lead + n_rep][n_pairs]; every matrix agrees with its own ngd_run() pass to rounding (the sums are formed in
// another order).  *done = false: the plan does not apply (no room for a useful chunk) and nothing has been written.
static uint32_t spill_unit(uint64_t block_size) {
  uint32_t q = 1;
  for (uint32_t d = 2; d <= 64 && d <= block_size; d++)
    if (block_size % d == 0) q = d;
  return q;
}

static int em_spill_impl(ngd_engine *e, const uint32_t *mult, const uint32_t *mult_max, const unsigned long long *drawn,
                         uint32_t n_rep, bool lead, uint64_t n_blocks, uint64_t block_size, double *d_sum,
                         unsigned long long *d_cnt, bool *done) {
  const ngd_geom &g = e->g;
  *done = false;
  const uint64_t n_pairs = ngd_n_pairs(g.n_ind);
  const uint64_t n_eff = n_blocks * block_size;
  const uint32_t n_mat = n_rep + (lead ? 1u : 0u);
  const uint32_t n_rg = ngd_contract_rep_groups(n_mat);
  const uint64_t n_pg = e->n_pg_spill;  // groups of 16 pair slots (live groups only, padded to 4)
  const uint64_t s_end = lead ? g.n_sites : n_eff;
  if (!e->n_tiles64 || !n_pg || !e->d_rowpg || !s_end) return NGD_OK;
  const uint32_t q = spill_unit(block_size);
  const uint64_t kg_bytes = n_pg * 64 * 8;  // one k-group (4 units) of terms
  const uint64_t d_elems = (uint64_t)n_rg * n_pg * 256;
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  const uint64_t have = e->slab_boot_elems * 8 + e->cap_D * 8;
  // default scratch: 6 GB of terms (a device allocation costs ~12 ms per GB once; [measured] tools/em_boot_job.py)
  uint64_t budget = e->opt_em_spill_bytes ? e->opt_em_spill_bytes : (6ull << 30);
  if (e->opt_boot_max_bytes) budget = std::min(budget, e->opt_boot_max_bytes);
  const uint64_t room = free_b + have > d_elems * 8 + (2ull << 30) ? free_b + have - d_elems * 8 - (2ull << 30) : 0;
  budget = std::min(budget, room);
  uint64_t chunk_kg = budget / kg_bytes;
  if (chunk_kg < 2) return NGD_OK;
  const uint64_t units_all = (s_end + q - 1) / q;
  chunk_kg = std::min<uint64_t>(chunk_kg - 1, (units_all + 3) / 4);  // (one k-group of tail for the operand run-ahead)
  // chunks of a few sites are launch-bound: the plan is left to the others (unless the caller set the scratch size)
  if (!e->opt_em_spill_bytes && chunk_kg * 4 * q < std::min<uint64_t>(s_end, 64)) return NGD_OK;
  const uint64_t chunk_sites = chunk_kg * 4 * q;
  const uint64_t n_chunks = (s_end + chunk_sites - 1) / chunk_sites;

  e->boot_B = 0;  // the partial-sum slab is this plan's scratch
  int rc = ensure_cap(e, &e->slab_boot, &e->slab_boot_elems, (chunk_kg + 1) * n_pg * 64);
  if (rc) return rc;
  rc = ensure_cap(e, &e->d_W, &e->cap_W, (chunk_kg + 1) * (uint64_t)n_rg * 64);
  if (rc) return rc;
  rc = ensure_cap(e, &e->d_M, &e->cap_M, std::max<uint64_t>(1, (uint64_t)n_rep * n_blocks));
  if (rc) return rc;
  rc = ensure_cap(e, &e->d_D, &e->cap_D, d_elems);
  if (rc) return rc;
  rc = ensure_cap(e, &e->d_nanflag, &e->cap_nanflag, n_chunks);
  if (rc) return rc;
  while (e->ev_spill.size() < 4 * n_chunks + 1) {  // (kept for the engine's lifetime)
    hipEvent_t v = nullptr;
    HIPCHK(hipEventCreate(&v));
    e->ev_spill.push_back(v);
  }
  *done = true;

  HIPCHK(hipEventRecord(e->ev[0], e->st));
  if (n_rep) HIPCHK(hipMemcpyAsync(e->d_M, mult, (uint64_t)n_rep * n_blocks * 4, hipMemcpyHostToDevice, e->st));
  HIPCHK(hipMemsetAsync(e->d_D, 0, d_elems * 8, e->st));
  HIPCHK(hipMemsetAsync(e->d_nanflag, 0, n_chunks * 8, e->st));
  if (e->cfg.shard_world > 1) HIPCHK(hipMemsetAsync(d_sum, 0, (uint64_t)n_mat * n_pairs * sizeof(double), e->st));
  if (e->cfg.shard_world > 1 || e->cfg.pairwise_del)  // k_count adds with integer atomics
    HIPCHK(hipMemsetAsync(d_cnt, 0, (uint64_t)n_mat * n_pairs * sizeof(unsigned long long), e->st));
  HIPCHK(hipEventRecord(e->ev[1], e->st));
  double *C = e->slab_boot;
  uint64_t units_done = 0;
  for (uint64_t c = 0; c < n_chunks; c++) {
    const uint64_t s_lo = c * chunk_sites, s_hi = std::min(s_end, s_lo + chunk_sites);
    const uint64_t len = s_hi - s_lo, n_units = (len + q - 1) / q, n_kg = (n_units + 3) / 4;
    units_done += n_units;
    hipEvent_t *ev = &e->ev_spill[4 * c];
    HIPCHK(hipEventRecord(ev[0], e->st));
    if (n_units & 3) HIPCHK(hipMemsetAsync(C + (n_kg - 1) * n_pg * 64, 0, kg_bytes, e->st));  // the last k-group is partial
    if (n_pg > e->n_pg_live)  // the slot groups of padding, which no wavefront of the EM pass writes
      HIPCHK(hipMemset2DAsync(C + (uint64_t)e->n_pg_live * 64, kg_bytes, 0, (n_pg - e->n_pg_live) * 512, n_kg, e->st));
    ngd_launch_spill_weights(e->st, e->d_M, n_mat, lead ? 1 : 0, s_lo, s_hi, q, g.n_sites, n_eff, n_blocks, block_size, e->d_W);
    HIPCHK(hipEventRecord(ev[1], e->st));
    // slices of the chunk's sites (whole units): enough workgroups to fill the device a few times over, a few sites each
    // at least
    uint64_t ks = std::max<uint64_t>(1, std::min<uint64_t>((8192 + e->n_tiles64 - 1) / e->n_tiles64, len / 8));
    const uint64_t sps = ((len + ks - 1) / ks + q - 1) / q * q;
    ks = (len + sps - 1) / sps;
    ngd_launch_accum_em_table_spill(e->st, g, e->PA, s_lo, s_hi, e->sc, e->cfg.pairwise_del, e->d_tiles64, e->n_tiles64,
                                    (uint32_t)ks, sps, q, e->d_rowpg, (uint32_t)n_pg, C, e->d_emcnt, e->d_nanflag + c);
    HIPCHK(hipEventRecord(ev[2], e->st));
    ngd_launch_spill_sanitize(e->st, C, e->d_nanflag + c, n_kg, (uint32_t)n_pg, e->d_M, n_mat, lead ? 1 : 0, s_lo, q,
                              g.n_sites, n_eff, n_blocks, block_size, e->d_D);
    HIPCHK(hipEventRecord(ev[3], e->st));
    ngd_launch_contract(e->st, e->d_W, C, n_mat, (uint32_t)n_pg, (uint32_t)n_kg, e->d_D);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipEventRecord(e->ev_spill[4 * n_chunks], e->st));
  HIPCHK(hipEventRecord(e->ev[2], e->st));
  ngd_launch_spill_scatter(e->st, e->d_D, (uint32_t)n_pg, e->d_tiles64, e->n_tiles64, e->d_rowpg, g.n_ind, n_mat, d_sum);
  HIPCHK(hipEventRecord(e->ev[3], e->st));
  std::vector<unsigned long long> visited;  // (alive until the synchronisation below)
  if (!e->cfg.pairwise_del) {  // every pair of matrix r counts the sites the matrix visits: one launch for the job
    visited.resize(n_mat);
    for (uint32_t r = 0; r < n_mat; r++) visited[r] = lead && r == 0 ? g.n_sites : drawn[r - (lead ? 1u : 0u)];
    rc = ensure_cap(e, &e->d_drawn, &e->cap_drawn, (uint64_t)n_mat);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(e->d_drawn, visited.data(), (uint64_t)n_mat * 8, hipMemcpyHostToDevice, e->st));
    ngd_launch_fill_cnt(e->st, g, e->d_tiles, e->n_tiles, 0, e->d_drawn, n_mat, d_cnt);
  }
  for (uint32_t r = 0; r < n_mat && e->cfg.pairwise_del; r++) {
    unsigned long long *cnt_r = d_cnt + (uint64_t)r * n_pairs;
    const bool is_lead = lead && r == 0;
    const uint32_t qr = r - (lead ? 1u : 0u);
    if (is_lead) {
      ngd_launch_count(e->st, g, e->mask, e->planes, 0, e->d_tiles, e->n_tiles, cnt_r);
    } else {
      uint32_t n_planes = 0;
      while (n_planes < 32 && (mult_max[qr] >> n_planes)) n_planes++;
      if (!n_planes) n_planes = 1;  // (as in pass_impl: a replicate that drew none of these blocks counts 0 sites)
      ngd_launch_weights(e->st, n_blocks, block_size, g.n_sites_pad, e->d_M + (uint64_t)qr * n_blocks, e->d_ws, nullptr);
      ngd_launch_planes(e->st, e->d_ws, g.n_sites, g.n_words, n_planes, e->planes);
      ngd_launch_count(e->st, g, e->mask, e->planes, n_planes, e->d_tiles, e->n_tiles, cnt_r);
    }
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->ev[4], e->st));
  HIPCHK(hipStreamSynchronize(e->st));  // `mult` is the caller's host memory
  read_timing(e, s_end, 1, false);
  {  // where the accumulation phase went, kernel by kernel (ngd_last_spill_timing)
    ngd_spill_timing &t = e->spill_timing;
    t = ngd_spill_timing{};
    for (uint64_t c = 0; c < n_chunks; c++) {
      float ms[4] = {0, 0, 0, 0};
      for (int k = 0; k < 4; k++) hipEventElapsedTime(&ms[k], e->ev_spill[4 * c + k], e->ev_spill[4 * c + k + 1]);
      t.ms_weights += ms[0]; t.ms_terms += ms[1]; t.ms_sanitize += ms[2]; t.ms_contract += ms[3];
    }
    t.chunks = n_chunks; t.units = units_done; t.unit_sites = q; t.sites = s_end;
    t.slot_groups = n_pg; t.slot_groups_live = e->n_pg_live; t.matrices = n_mat; t.matrix_groups = n_rg;
    t.contract_launches = n_chunks * ((n_rg + 7) / 8);
  }
  return NGD_OK;
}

// The replicate loop: optionally the full data set (matrix 0, lead_full), then n_rep bootstrap replicates given
// as block maps (multiplicities are counted from them) or directly as multiplicities.  Outputs are
// [lead_full + n_rep][n_pairs].  The plan is the cheapest that applies: per-block partials (one pass, then a
// weighted reduction per batch of replicates), the EM batch pass, or one (weighted) pass per matrix.
static int run_impl(ngd_engine *e, const uint64_t *block_maps, const uint32_t *mult_in, uint32_t n_rep, bool lead_full,
                    uint64_t n_blocks, uint64_t block_size, double *d_sum, unsigned long long *d_cnt) {
  if (!e) return fail(NGD_E_INVALID, "ngd_run: null engine");
  if (!e->committed) return fail(NGD_E_INVALID, "ngd_run: call ngd_commit() first");
  HIPCHK(hipSetDevice(e->device));
  const ngd_geom &g = e->g;
  e->spill_timing = ngd_spill_timing{};
  e->fix_info = ngd_fixup_info{};
  e->n_batch_valid = 0;  // (the matrices of an earlier batch are not this call's: set again by copy_out() on success)
  if (!n_rep) return pass_impl(e, nullptr, 0, 0, 0, 0, d_sum, d_cnt, false);
  if (int rc = eager_discard(e)) return rc;  // (a job: its plans share passes between matrices; nothing of a plain pass is reused)

  if (!block_size || !n_blocks) return fail(NGD_E_INVALID, "ngd_run: empty bootstrap geometry");
  if (n_blocks > g.n_sites / block_size) return fail(NGD_E_INVALID, "ngd_run: n_blocks*block_size exceeds n_sites");
  const uint64_t n_pairs = ngd_n_pairs(g.n_ind);
  const uint64_t n_eff = n_blocks * block_size;
  const uint32_t lead = lead_full ? 1u : 0u;
  std::vector<unsigned long long> drawn(n_rep + lead, 0);  // sites visited, with multiplicity = gen_dist's cnt
  std::vector<uint32_t> mult_max(n_rep + lead, 0);
  // multiplicities of matrices lead..: counted into (or copied behind) a leading all-ones row, which stands for
  // the full data set when the blocks cover every site
  const uint64_t need = (uint64_t)(n_rep + lead) * n_blocks;
  if (block_maps || lead) {
    if (need > e->cap_h_mult) {  // pinned and kept: a replicate at block size 1 counts a million draws per call
      if (e->h_mult) { HIPCHK(hipHostFree(e->h_mult)); e->h_mult = nullptr; e->cap_h_mult = 0; }
      HIPCHK(hipHostMalloc((void **)&e->h_mult, need * sizeof(uint32_t), hipHostMallocDefault));
      e->cap_h_mult = need;
    }
  }
  const uint32_t *mult = mult_in;  // [n_rep][n_blocks]
  if (block_maps) {
    uint32_t *base = e->h_mult + (uint64_t)lead * n_blocks;
    memset(base, 0, (uint64_t)n_rep * n_blocks * sizeof(uint32_t));
    for (uint32_t r = 0; r < n_rep; r++) {
      uint32_t *m = base + (uint64_t)r * n_blocks;
      const uint64_t *bm = block_maps + (uint64_t)r * n_blocks;
      for (uint64_t b = 0; b < n_blocks; b++) {
        if (bm[b] >= n_blocks) return fail(NGD_E_INVALID, "ngd_run: block_map entry out of range");
        mult_max[lead + r] = std::max(mult_max[lead + r], ++m[bm[b]]);
      }
      drawn[lead + r] = n_eff;
    }
    mult = base;
  } else {
    for (uint32_t r = 0; r < n_rep; r++)
      for (uint64_t b = 0; b < n_blocks; b++) {
        const uint32_t m = mult[(uint64_t)r * n_blocks + b];
        mult_max[lead + r] = std::max(mult_max[lead + r], m);
        drawn[lead + r] += (unsigned long long)m * block_size;
      }
    if (lead) {
      memcpy(e->h_mult + n_blocks, mult, (uint64_t)n_rep * n_blocks * sizeof(uint32_t));
      mult = e->h_mult + n_blocks;
    }
  }
  if (lead) {
    for (uint64_t b = 0; b < n_blocks; b++) e->h_mult[b] = 1u;
    drawn[0] = g.n_sites;
    mult_max[0] = 1;
  }
  if (e->out.on) e->out.cnt_mat.assign(drawn.begin(), drawn.end());  // (no --pairwise_del: a matrix's count, every pair's)
  double *rep_sum = d_sum + (uint64_t)lead * n_pairs;
  unsigned long long *rep_cnt = d_cnt + (uint64_t)lead * n_pairs;

  // 1. per-block partials; the full data set rides along as the all-ones row when the blocks cover every site
  bool feasible = false;
  int rc;
  if (lead && n_eff == g.n_sites) {
    rc = partials_impl(e, e->h_mult, drawn.data(), n_rep + 1, n_blocks, block_size, d_sum, d_cnt, &feasible);
    if (rc || feasible) return rc;
  } else {
    rc = partials_impl(e, mult, drawn.data() + lead, n_rep, n_blocks, block_size, rep_sum, rep_cnt, &feasible);
    if (rc) return rc;
    if (feasible) return lead ? pass_impl(e, nullptr, 0, 0, 0, 0, d_sum, d_cnt, true) : NGD_OK;
  }
  // 2. EM kernels: many matrices per accumulation pass (the EM of a (pair, site) is computed once and added to up to
  //    16 accumulators per pair -- 8 in the table-driven kernel).  The faithful form keeps matrix 0 on the plain pass,
  //    whose accumulation is the reference's term by term.  The table-driven kernel's batch pass runs one workgroup per
  //    CU and costs ~1.7 plain passes: from three matrices on it beats a plain pass + a weighted pass per replicate
  //    (0.63 of a pass each: they walk only the sites a replicate drew).  Its other shapes borrow the per-pair batch
  //    kernel from three replicates on (those agree with ngd_run()'s to rounding only).
  // 2a. the table-driven kernel, three matrices or more: the terms of a chunk of sites are spilled once and
  //     contracted with every matrix's weights by MFMA -- one EM pass for the whole job, whatever the replicate count
  if (e->kernel == NGD_KERNEL_EM_TABLE && e->em_shape == 0 && e->opt_em_spill &&
      n_rep + lead >= (e->opt_em_spill == 2 ? 2u : 3u)) {
    bool done = false;
    rc = em_spill_impl(e, mult, mult_max.data() + lead, drawn.data() + lead, n_rep, lead != 0, n_blocks, block_size,
                       d_sum, d_cnt, &done);
    if (rc == NGD_E_NOMEM) { (void)hipGetLastError(); g_err.clear(); done = false; rc = NGD_OK; }
    if (rc || done) return rc;
  }
  const bool em_pair = e->kernel == NGD_KERNEL_EM_FAST || e->kernel == NGD_KERNEL_EM_FAITHFUL;
  const bool em_table_batch = e->kernel == NGD_KERNEL_EM_TABLE && e->em_shape == 0 && n_rep + lead >= 3;
  const bool em_borrow = e->kernel == NGD_KERNEL_EM_TABLE && e->em_shape != 0 && n_rep >= 3;
  if ((em_pair || em_borrow || em_table_batch) && n_rep + lead >= 2 && e->opt_em_batch) {
    const bool fold = lead && e->kernel != NGD_KERNEL_EM_FAITHFUL;
    if (lead && !fold) {
      rc = pass_impl(e, nullptr, 0, 0, 0, 0, d_sum, d_cnt, false);
      if (rc) return rc;
    }
    rc = em_batch_impl(e, mult, mult_max.data() + lead, drawn.data() + lead, n_rep, fold, n_blocks, block_size,
                       fold ? d_sum : rep_sum, fold ? d_cnt : rep_cnt, lead && !fold);
    // the batch pass wants RB result planes per slice: if the device cannot hold them (very many individuals), the
    // matrices are computed one pass each instead (the allocation is tried before anything is launched)
    if (rc != NGD_E_NOMEM) return rc;
    (void)hipGetLastError();
    g_err.clear();  // not an error of this call: the matrices are computed one pass each instead
    if (lead && !fold) {  // matrix 0 is done already
      for (uint32_t r = 0; r < n_rep; r++) {
        rc = pass_impl(e, mult + (uint64_t)r * n_blocks, mult_max[lead + r], n_blocks, block_size, drawn[lead + r],
                       rep_sum + (uint64_t)r * n_pairs, rep_cnt + (uint64_t)r * n_pairs, true);
        if (rc) return rc;
      }
      return NGD_OK;
    }
  }
  // 3. one accumulation pass per matrix
  if (lead) {
    rc = pass_impl(e, nullptr, 0, 0, 0, 0, d_sum, d_cnt, false);
    if (rc) return rc;
  }
  for (uint32_t r = 0; r < n_rep; r++) {
    rc = pass_impl(e, mult + (uint64_t)r * n_blocks, mult_max[lead + r], n_blocks, block_size, drawn[lead + r],
                   rep_sum + (uint64_t)r * n_pairs, rep_cnt + (uint64_t)r * n_pairs, lead || r > 0);
    if (rc) return rc;
  }
  return NGD_OK;
}

static int copy_out(ngd_engine *e, uint32_t n_mat, const double *d_sum, const unsigned long long *d_cnt, double *sum,
                    uint64_t *cnt) {
  e->n_batch_valid = d_sum == e->d_bsum ? n_mat : 0;
  const uint64_t n = (uint64_t)n_mat * ngd_n_pairs(e->g.n_ind);
  if (sum) HIPCHK(hipMemcpy(sum, d_sum, n * sizeof(double), hipMemcpyDeviceToHost));
  if (cnt) HIPCHK(hipMemcpy(cnt, d_cnt, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return NGD_OK;
}

static int batch_buffers(ngd_engine *e, uint32_t n_rep) {
  e->n_batch_valid = 0;  // (the buffers may be freed and grown below; a failed call leaves nothing to fetch)
  HIPCHK(hipSetDevice(e->device));
  const uint64_t need = (uint64_t)n_rep * ngd_n_pairs(e->g.n_ind);
  if (need <= e->cap_batch) return NGD_OK;
  uint64_t cap_s = e->cap_batch, cap_c = e->cap_batch;
  int rc = ensure_cap(e, &e->d_bsum, &cap_s, need);
  if (rc) { e->cap_batch = 0; return rc; }
  rc = ensure_cap(e, &e->d_bcnt, &cap_c, need);
  if (rc) { e->cap_batch = 0; return rc; }
  e->cap_batch = need;
  return NGD_OK;
}

int ngd_run_device(ngd_engine *e, const uint64_t *block_map, uint64_t n_blocks, uint64_t block_size,
                   void *d_sum, void *d_cnt) {
  if (!d_sum || !d_cnt) return fail(NGD_E_INVALID, "ngd_run_device: null output");
  return run_impl(e, block_map, nullptr, block_map ? 1 : 0, false, n_blocks, block_size, (double *)d_sum,
                  (unsigned long long *)d_cnt);
}

int ngd_run_mult_device(ngd_engine *e, const uint32_t *mult, uint64_t n_blocks, uint64_t block_size, void *d_sum,
                        void *d_cnt) {
  if (!d_sum || !d_cnt || !mult) return fail(NGD_E_INVALID, "ngd_run_mult_device: null argument");
  return run_impl(e, nullptr, mult, 1, false, n_blocks, block_size, (double *)d_sum, (unsigned long long *)d_cnt);
}

int ngd_run_mult(ngd_engine *e, const uint32_t *mult, uint64_t n_blocks, uint64_t block_size, double *sum,
                 uint64_t *cnt) {
  if (!e || !mult) return fail(NGD_E_INVALID, "ngd_run_mult: null argument");
  int rc = run_impl(e, nullptr, mult, 1, false, n_blocks, block_size, e->d_sum, e->d_cnt);
  if (rc) return rc;
  return copy_out(e, 1, e->d_sum, e->d_cnt, sum, cnt);
}

int ngd_run(ngd_engine *e, const uint64_t *block_map, uint64_t n_blocks, uint64_t block_size, double *sum,
            uint64_t *cnt) {
  if (!e) return fail(NGD_E_INVALID, "ngd_run: null engine");
  int rc = run_impl(e, block_map, nullptr, block_map ? 1 : 0, false, n_blocks, block_size, e->d_sum, e->d_cnt);
  if (rc) return rc;
  return copy_out(e, 1, e->d_sum, e->d_cnt, sum, cnt);
}

int ngd_run_batch_device(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks,
                         uint64_t block_size, void *d_sum, void *d_cnt) {
  if (!block_maps || !n_rep || !d_sum || !d_cnt) return fail(NGD_E_INVALID, "ngd_run_batch_device: null argument");
  return run_impl(e, block_maps, nullptr, n_rep, false, n_blocks, block_size, (double *)d_sum, (unsigned long long *)d_cnt);
}

int ngd_run_mult_batch_device(ngd_engine *e, const uint32_t *mult, uint32_t n_rep, uint64_t n_blocks,
                              uint64_t block_size, void *d_sum, void *d_cnt) {
  if (!mult || !n_rep || !d_sum || !d_cnt) return fail(NGD_E_INVALID, "ngd_run_mult_batch_device: null argument");
  return run_impl(e, nullptr, mult, n_rep, false, n_blocks, block_size, (double *)d_sum, (unsigned long long *)d_cnt);
}

int ngd_run_batch(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks, uint64_t block_size,
                  double *sum, uint64_t *cnt) {
  if (!e || !block_maps || !n_rep) return fail(NGD_E_INVALID, "ngd_run_batch: null argument");
  int rc = batch_buffers(e, n_rep);
  if (rc) return rc;
  rc = run_impl(e, block_maps, nullptr, n_rep, false, n_blocks, block_size, e->d_bsum, e->d_bcnt);
  if (rc) return rc;
  return copy_out(e, n_rep, e->d_bsum, e->d_bcnt, sum, cnt);
}

int ngd_run_mult_batch(ngd_engine *e, const uint32_t *mult, uint32_t n_rep, uint64_t n_blocks, uint64_t block_size,
                       double *sum, uint64_t *cnt) {
  if (!e || !mult || !n_rep) return fail(NGD_E_INVALID, "ngd_run_mult_batch: null argument");
  int rc = batch_buffers(e, n_rep);
  if (rc) return rc;
  rc = run_impl(e, nullptr, mult, n_rep, false, n_blocks, block_size, e->d_bsum, e->d_bcnt);
  if (rc) return rc;
  return copy_out(e, n_rep, e->d_bsum, e->d_bcnt, sum, cnt);
}

int ngd_run_job_device(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks,
                       uint64_t block_size, void *d_sum, void *d_cnt) {
  if (!d_sum || !d_cnt || (n_rep && !block_maps)) return fail(NGD_E_INVALID, "ngd_run_job_device: null argument");
  return run_impl(e, block_maps, nullptr, n_rep, n_rep != 0, n_blocks, block_size, (double *)d_sum,
                  (unsigned long long *)d_cnt);
}

int ngd_run_job(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks, uint64_t block_size,
                double *sum, uint64_t *cnt) {
  if (!e || (n_rep && !block_maps)) return fail(NGD_E_INVALID, "ngd_run_job: null argument");
  int rc = batch_buffers(e, n_rep + 1);
  if (rc) return rc;
  rc = run_impl(e, block_maps, nullptr, n_rep, n_rep != 0, n_blocks, block_size, e->d_bsum, e->d_bcnt);
  if (rc) return rc;
  return copy_out(e, n_rep + 1, e->d_bsum, e->d_bcnt, sum, cnt);
}

// A whole job AND the tail of gen_dist() (ngsDist.cpp:372-401) in one call: the sums (and, --pairwise_del, the counts) leave
// the device chunk by chunk on a stream of their own into pinned memory of the engine's while -- in the per-block-partials
// plan -- later groups of replicates are still being reduced, and the host's threads turn each chunk into distances as it
// lands.  The matrices stay in the engine as after ngd_run_job(..., NULL, NULL) (ngd_fetch_matrix).
static int run_dist(ngd_engine *e, const uint64_t *block_maps, const uint32_t *mult, uint32_t n_rep, bool lead_full,
                    uint64_t n_blocks, uint64_t block_size, uint64_t tot_sites, uint64_t evol_model, double *dist, const char *who) {
  if (!e || !dist) return fail(NGD_E_INVALID, std::string(who) + ": null argument");
  if (tot_sites && e->cfg.pairwise_del)
    return fail(NGD_E_INVALID, std::string(who) + ": a total number of sites cannot go with pairwise deletion (parse_args.cpp:209-210)");
  if (evol_model > 2) return fail(NGD_E_MODEL, std::string(who) + ": evolutionary model not supported (ngsDist.cpp:398-399)");
  if (e->cfg.shard_world > 1)
    return fail(NGD_E_INVALID, std::string(who) + ": an engine that owns a share of the pairs holds part of every matrix -- put the "
                               "shares together first (ngd_run_job_device + the ranks' exchange), then ngd_finish()");
  if (!e->committed) return fail(NGD_E_INVALID, std::string(who) + ": call ngd_commit() first");
  const double t_enter = out_now();
  HIPCHK(hipSetDevice(e->device));
  const uint32_t n_mat = n_rep ? n_rep + (lead_full ? 1u : 0u) : 1u;
  int rc = batch_buffers(e, n_mat);
  if (rc) return rc;
  auto &o = e->out;
  const uint64_t cells = (uint64_t)n_mat * ngd_n_pairs(e->g.n_ind);
  if (!o.st) HIPCHK(hipStreamCreateWithFlags(&o.st, hipStreamNonBlocking));
  if (!o.st2) HIPCHK(hipStreamCreateWithFlags(&o.st2, hipStreamNonBlocking));
  if (cells > o.cap_sum) {
    if (o.h_sum) { HIPCHK(hipHostFree(o.h_sum)); o.h_sum = nullptr; o.cap_sum = 0; }
    HIPCHK(hipHostMalloc((void **)&o.h_sum, std::max<uint64_t>(1, cells) * sizeof(double), hipHostMallocDefault));
    o.cap_sum = cells;
  }
  o.pdel = e->cfg.pairwise_del != 0;
  if (o.pdel && cells > o.cap_cnt) {
    if (o.h_cnt) { HIPCHK(hipHostFree(o.h_cnt)); o.h_cnt = nullptr; o.cap_cnt = 0; }
    HIPCHK(hipHostMalloc((void **)&o.h_cnt, std::max<uint64_t>(1, cells) * sizeof(uint64_t), hipHostMallocDefault));
    o.cap_cnt = cells;
  }
  o.n_mat = n_mat;
  o.queued = 0;
  o.n_used = 0;
  o.n_chunk_seq = 0;
  o.n_landed = 0;
  o.chunks.clear();
  o.landed = 0;
  o.evol_model = evol_model;
  o.tot_sites = tot_sites;
  o.dist = dist;
  o.cnt_mat.assign(n_mat, e->g.n_sites);  // (a plain pass; a job's run_impl writes its matrices' own)
  o.on = true;
  o.t_call = out_now();
  rc = run_impl(e, block_maps, mult, n_rep, n_rep && lead_full, n_blocks, block_size, e->d_bsum, e->d_bcnt);
  rc = rc ? out_join(e, rc) : out_land(e);
  if (!rc) e->n_batch_valid = n_mat;
  if (out_trace()) fprintf(stderr, "[out] call: %.3f ms (setup before it %.3f)\n", out_now() - o.t_call, o.t_call - t_enter);
  return rc;
}

int ngd_run_job_dist(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks, uint64_t block_size,
                     uint64_t tot_sites, uint64_t evol_model, double *dist) {
  if (n_rep && !block_maps) return fail(NGD_E_INVALID, "ngd_run_job_dist: null argument");
  return run_dist(e, block_maps, nullptr, n_rep, true, n_blocks, block_size, tot_sites, evol_model, dist, "ngd_run_job_dist");
}

int ngd_run_batch_dist(ngd_engine *e, const uint64_t *block_maps, uint32_t n_rep, uint64_t n_blocks, uint64_t block_size,
                       uint64_t tot_sites, uint64_t evol_model, double *dist) {
  if (!block_maps || !n_rep) return fail(NGD_E_INVALID, "ngd_run_batch_dist: null argument");
  return run_dist(e, block_maps, nullptr, n_rep, false, n_blocks, block_size, tot_sites, evol_model, dist, "ngd_run_batch_dist");
}

int ngd_run_mult_batch_dist(ngd_engine *e, const uint32_t *mult, uint32_t n_rep, uint64_t n_blocks, uint64_t block_size,
                            uint64_t tot_sites, uint64_t evol_model, double *dist) {
  if (!mult || !n_rep) return fail(NGD_E_INVALID, "ngd_run_mult_batch_dist: null argument");
  return run_dist(e, nullptr, mult, n_rep, false, n_blocks, block_size, tot_sites, evol_model, dist, "ngd_run_mult_batch_dist");
}

int ngd_fetch_matrix(ngd_engine *e, uint32_t which, double *sum, uint64_t *cnt) {
  if (!e) return fail(NGD_E_INVALID, "ngd_fetch_matrix: null engine");
  if (which >= e->n_batch_valid) return fail(NGD_E_INVALID, "ngd_fetch_matrix: no such matrix in the engine's last batch");
  HIPCHK(hipSetDevice(e->device));
  const uint64_t n = ngd_n_pairs(e->g.n_ind);
  if (sum) HIPCHK(hipMemcpy(sum, e->d_bsum + (uint64_t)which * n, n * sizeof(double), hipMemcpyDeviceToHost));
  if (cnt) HIPCHK(hipMemcpy(cnt, e->d_bcnt + (uint64_t)which * n, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return NGD_OK;
}

int ngd_set_option(ngd_engine *e, int option, uint64_t value) {
  if (!e) return fail(NGD_E_INVALID, "ngd_set_option: null engine");
  switch (option) {
    case NGD_OPT_BOOT_PARTIALS:
      if (value > 2) return fail(NGD_E_INVALID, "ngd_set_option: NGD_OPT_BOOT_PARTIALS is 0, 1 or 2");
      e->opt_boot_partials = value;
      break;
    case NGD_OPT_BOOT_MAX_BYTES: e->opt_boot_max_bytes = value; break;
    case NGD_OPT_BOOT_WG:
      if (!value) return fail(NGD_E_INVALID, "ngd_set_option: NGD_OPT_BOOT_WG must be positive");
      e->opt_boot_wg = value;
      break;
    case NGD_OPT_BOOT_UNALIGNED: e->opt_boot_unaligned = value != 0; break;
    case NGD_OPT_EM_BATCH: e->opt_em_batch = value != 0; break;
    case NGD_OPT_EM_SPILL:
      if (value > 2) return fail(NGD_E_INVALID, "ngd_set_option: NGD_OPT_EM_SPILL is 0, 1 or 2");
      e->opt_em_spill = value;
      break;
    case NGD_OPT_EM_SPILL_BYTES: e->opt_em_spill_bytes = value; break;
    case NGD_OPT_SINGLE_IMAGE_BYTES:
      if (!e->cfg.single_image) return fail(NGD_E_INVALID, "ngd_set_option: NGD_OPT_SINGLE_IMAGE_BYTES needs ngd_config.single_image");
      if (!e->single_image) break;  // (another kernel, or second_image_mib holds the whole second image: nothing is formed)
      e->qb_chunk_kg = std::max<uint64_t>(1, (value ? value : 4ull << 30) / ((uint64_t)e->g.n_ig * 64 * 8));
      break;
    case NGD_OPT_FIXUP_WORK: e->opt_fix_work = value; break;
    case NGD_OPT_EAGER_FULL:
      if (e->pin_sites || e->committed) return fail(NGD_E_INVALID, "ngd_set_option: NGD_OPT_EAGER_FULL before the first ngd_stage_acquire");
      e->opt_eager = value != 0 && eager_supported(e);  // (kernels without a slice-range launch: silently off)
      break;
    case NGD_OPT_STAGE_PIECE_MIB:
    case NGD_OPT_STAGE_RING:
      if (e->pin_sites) return fail(NGD_E_INVALID, "ngd_set_option: the staging ring exists already (set before the first ngd_stage_acquire)");
      if (option == NGD_OPT_STAGE_RING) {
        if (value < 2 || value > ngd_engine::RING) return fail(NGD_E_INVALID, "ngd_set_option: NGD_OPT_STAGE_RING is 2 .. 8");
        e->opt_stage_ring = value;
      } else {
        if (value < 1 || value > 1024) return fail(NGD_E_INVALID, "ngd_set_option: NGD_OPT_STAGE_PIECE_MIB is 1 .. 1024");
        e->opt_stage_piece_mib = value;
      }
      break;
    case NGD_OPT_DEBUG_FORGE_JOB: {  // tests only: the first block of the MFMA job list gets another shape
      // (an engine with a forged job list computes nothing right ever after: refused unless the process says it is a test)
      const char *hook = getenv("NGD_ENABLE_TEST_HOOKS");
      if (!hook || strcmp(hook, "1") != 0)
        return fail(NGD_E_INVALID, "ngd_set_option: NGD_OPT_DEBUG_FORGE_JOB is a test hook (set NGD_ENABLE_TEST_HOOKS=1)");
      if (e->kernel != NGD_KERNEL_MFMA || !e->d_jobs) return fail(NGD_E_INVALID, "ngd_set_option: no MFMA job list");
      HIPCHK(hipSetDevice(e->device));
      ngd_job j;
      HIPCHK(hipMemcpy(&j, e->d_jobs, sizeof(j), hipMemcpyDeviceToHost));
      j.rows = (uint8_t)(value & 7); j.cols = (uint8_t)((value >> 3) & 7); j.tri = (uint8_t)((value >> 6) & 1);
      HIPCHK(hipMemcpy(e->d_jobs, &j, sizeof(j), hipMemcpyHostToDevice));
      break;
    }
    default: return fail(NGD_E_INVALID, "ngd_set_option: unknown option");
  }
  return NGD_OK;
}

int ngd_drop_caches(ngd_engine *e) {
  if (!e) return fail(NGD_E_INVALID, "ngd_drop_caches: null engine");
  e->boot_B = 0;
  e->boot_blocks = 0;
  e->cnt_B = 0;
  e->cnt_blocks = 0;
  e->em_batch_nofit_elems = 0;  // (what did not fit may fit now: the partial results' slab is the same scratch)
  return NGD_OK;
}

int ngd_last_em_work(const ngd_engine *e, uint64_t *tile_sites, uint64_t *table_rounds) {
  if (!e) return fail(NGD_E_INVALID, "ngd_last_em_work: null engine");
  if (tile_sites) *tile_sites = e->em_counts[0];
  if (table_rounds) *table_rounds = e->em_counts[1];
  return NGD_OK;
}

int ngd_last_shader_clock(const ngd_engine *e, double *mhz) {
  if (!e || !mhz) return fail(NGD_E_INVALID, "ngd_last_shader_clock: null argument");
  *mhz = e->clk_mhz;
  return NGD_OK;
}

int ngd_last_fixup(const ngd_engine *e, ngd_fixup_info *info) {
  if (!e || !info) return fail(NGD_E_INVALID, "ngd_last_fixup: null argument");
  *info = e->fix_info;
  return NGD_OK;
}

int ngd_image_mode(const ngd_engine *e, int *fixup) {
  if (!e) return fail(NGD_E_INVALID, "ngd_image_mode: null engine");
  if (fixup) *fixup = e->SM != nullptr;
  if (e->kernel != NGD_KERNEL_MFMA) return 0;
  return e->congruent ? 2 : e->single_image ? 1 : 3;
}

int ngd_last_spill_timing(const ngd_engine *e, ngd_spill_timing *t) {
  if (!e || !t) return fail(NGD_E_INVALID, "ngd_last_spill_timing: null argument");
  *t = e->spill_timing;
  return NGD_OK;
}

int ngd_last_timing(const ngd_engine *e, ngd_timing *t) {
  if (!e || !t) return fail(NGD_E_INVALID, "ngd_last_timing: null argument");
  *t = e->timing;
  return NGD_OK;
}

}  // extern "C"
