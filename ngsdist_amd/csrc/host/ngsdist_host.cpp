// ngsdist_host.cpp -- command-line host of the MI355X engine.
//
// Keeps the ngsDist command line (reference parse_args.cpp:54-80, defaults
// :6-37, checks :203-220), its input formats (read_data.cpp:13-116), its
// preparation of the array gen_dist() reads (ngsDist.cpp:165-174) and its
// output (ngsDist.cpp:282-287, "%.10f" cells joined by tabs), and hands the one
// hot block -- `for i1<i2: threadpool_add(gen_dist_slave)` + wait,
// ngsDist.cpp:244-269 -- to the engine through the C ABI of
// include/ngsdist_amd.h.  Written from scratch against that behaviour; it
// shares no code with the reference.
//
// Differences that are deliberate:
//  * binary input is streamed through in site chunks (read -> prepare ->
//    ngd_upload_sites), so the host never holds the n_ind x n_sites x 3 array;
//  * preparation (log / normalise / call / exp, all on the host's libm so cells
//    print identically) runs on --n_threads threads; it is per-element, so the
//    result does not depend on the thread count;
//  * bootstrap moves no data: the block map drawn from the reference's taus
//    stream is handed to ngd_run();
//  * a file name without a '.' is treated as binary instead of dereferencing
//    NULL (ngsDist.cpp:82);
//  * extra options: --n_gpus N (the SITE axis split over N devices of this node: every device reads and holds its own
//    range of sites and computes all pairs over it; the per-range sums are added -- byte-identical output for called
//    genotypes, <= 1e-12 relative otherwise), --device D (first device), --same_device (every range on --device: a
//    rehearsal on one GPU), --max_device_bytes B (device budget; a data set above it goes through in several ranges
//    per device), --kernel auto|stream|mfma|em_table|em_fast|em_faithful, --single_image / --two_images (--indep_geno on
//    the MFMA kernel: ngd_config.single_image = 2 / 3 -- ONE operand image in coordinates in which the score matrix is
//    diagonal + min(p0, p2) beside it: two thirds of the device memory per site, half the HBM reads, the same speed;
//    called genotypes print the same bytes, likelihood sums agree to 4e-17 per site and the pairs that is not enough
//    for, nearly identical individuals, are recomputed the two-operand way -- the engine's own choice above 384
//    individuals; or both images always),
//    --prep auto|host|device (where log/normalise/call/exp of a BINARY input run; auto =
//    device, except host when genotypes are called so that calls are decided by glibc),
//    --eager 0|1|2 (1, the default: on the EM path without bootstrap the full-data pass starts beside the load,
//    NGD_OPT_EAGER_FULL; 2: on the --indep_geno path too; 0: never), --stage piece_MiB,ring[,share_MiB[,drop]] (the load
//    pipeline's geometry, for measurements: tools/r6_stage_sweep.sh);
//  * a binary file is mapped and copied into the engine's ring of pinned buffers by --n_threads (4..16) threads, the
//    copies to the device and the preparation kernel running behind; matrices are written by a thread of their own
//    while the next one is formatted; --verbose 2 ends with a `> phases [s]:` line (where the run's wall time went).
#include <fcntl.h>
#include <getopt.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <ctime>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/ngsdist_amd.h"

static const char *kVersion = "ngsdist_amd 0.1 (ngsDist 1.0.10 command line)";
static const double kInf = 1e15;          // INF, gen_func.hpp:15
static const size_t kLineBuf = 500000;    // BUFF_LEN, gen_func.hpp:17

struct Pars {  // the reference's `params`, ngsDist.hpp:11-44
  const char *in_geno = nullptr;
  bool in_bin = false, in_probs = false, in_logscale = false;
  uint64_t n_ind = 0, n_sites = 0, tot_sites = 0;
  const char *in_labels = nullptr;
  bool in_labels_header = false;
  const char *in_pos = nullptr;
  bool in_pos_header = false;
  bool call_geno = false;
  double N_thresh = 0, call_thresh = 0;
  bool pairwise_del = false;
  double score[9] = {0, 0.5, 1, 0.5, 0, 0.5, 1, 0.5, 0};  // parse_args.cpp:25-27
  uint64_t evol_model = 1;
  bool indep_geno = false;
  uint64_t n_boot_rep = 0, boot_block_size = 1;
  const char *out = nullptr;
  unsigned n_threads = 1, verbose = 1, seed = 0;
  // engine placement (not in the reference)
  int n_gpus = 1, device = 0, kernel = NGD_KERNEL_AUTO;
  bool same_device = false;      // --same_device
  uint64_t max_device_bytes = 0; // --max_device_bytes (0 = 85 % of the device's free memory)
  bool single_image = false;     // --single_image
  bool two_images = false;       // --two_images
  int prep = 0;  // 0 auto (device unless genotypes are called), 1 host, 2 device
  unsigned stage_piece = 0, stage_ring = 0, stage_grain = 2, stage_drop = 1;  // --stage (0: the engine's defaults)
  int eager = 1;  // --eager 0|1|2: the full-data pass beside the load where the first engine call is that pass
};

// --verbose 2: where the wall time of a run goes, as one line of name=seconds pairs at the end of the run (stderr; the
// reference prints nothing comparable).  A mark closes the phase that ran since the previous mark.
struct PhaseLog {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
  std::vector<std::pair<std::string, double>> phases;
  std::mutex mu;  // (--n_gpus: every device's thread loads its own ranges)
  void mark(const char *name) {
    std::lock_guard<std::mutex> lk(mu);
    const auto t = std::chrono::steady_clock::now();
    phases.emplace_back(name, std::chrono::duration<double>(t - last).count());
    last = t;
  }
  void add(const char *name, double secs) {  // a component measured elsewhere (not on the timeline)
    std::lock_guard<std::mutex> lk(mu);
    phases.emplace_back(name, secs);
  }
  void print() const {
    fprintf(stderr, "> phases [s]:");
    for (auto &ph : phases) fprintf(stderr, " %s=%.4f", ph.first.c_str(), ph.second);
    fprintf(stderr, " total_since_main=%.4f\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
};
static PhaseLog g_phases;

// error(), gen_func.cpp:12-18: message, perror, exit(-1)
[[noreturn]] static void die(const char *func, const char *msg) {
  fflush(stdout);
  fprintf(stderr, "\n=====\nERROR: [%s] %s\n=====\n\n", func, msg);
  perror("\t");
  fflush(stderr);
  exit(-1);
}

static void die_engine(const char *func, int rc) {
  std::string m = std::string("engine error ") + std::to_string(rc) + ": " + ngd_last_error();
  die(func, m.c_str());
}

// One-image engines (--single_image, or the engine's own choice: include/ngsdist_amd.h ngd_config.single_image): what the
// fix-up pass of nearly identical pairs did in the last engine call.  Pairs it had to leave alone (more of them than its
// budget covers: a large data set of copies of one individual) keep an ABSOLUTE error bound -- far below what "%.10f"
// prints, but the user should know; said once.  --verbose 2: the pairs recomputed, every call.
static void report_fixup(ngd_engine *h, uint64_t verbose) {
  ngd_fixup_info f;
  if (ngd_last_fixup(h, &f) != NGD_OK) return;
  static bool warned = false;
  if (f.skipped && !warned) {
    warned = true;
    fprintf(stderr, "WARNING: %lu pairs of nearly identical individuals keep the sums of the one-image pass (absolute error <= 4e-17 "
                    "per site, not 1e-9 relative): more of them than the recomputation's budget covers; --two_images holds "
                    "1e-9 relative on every pair.\n", (unsigned long)f.skipped);
  }
  if (verbose >= 2 && f.recomputed)
    fprintf(stderr, "> %lu pairs of nearly identical individuals recomputed with the two-operand arithmetic (%.2f ms)\n",
            (unsigned long)f.recomputed, f.ms);
}

static const char *kModelNames[] = {"Raw p-distance", "Log transf. p-distance", "JC69", "K80", "F81",
                                    "HKY85/F84", "TN93"};

static void parse_cmd_args(Pars &p, int argc, char **argv) {
  static struct option opts[] = {{"geno", required_argument, nullptr, 'g'},
                                 {"probs", no_argument, nullptr, 'p'},
                                 {"log_scale", no_argument, nullptr, 'l'},
                                 {"n_ind", required_argument, nullptr, 'n'},
                                 {"n_sites", required_argument, nullptr, 's'},
                                 {"tot_sites", required_argument, nullptr, 'S'},
                                 {"labels", required_argument, nullptr, 'L'},
                                 {"labelsH", required_argument, nullptr, 'H'},
                                 {"pos", required_argument, nullptr, 'a'},
                                 {"posH", required_argument, nullptr, 'A'},
                                 {"call_geno", no_argument, nullptr, 'c'},
                                 {"N_thresh", required_argument, nullptr, 'N'},
                                 {"call_thresh", required_argument, nullptr, 'C'},
                                 {"pairwise_del", no_argument, nullptr, 'D'},
                                 {"avg_nuc_dist", no_argument, nullptr, 'd'},
                                 {"evol_model", required_argument, nullptr, 'm'},
                                 {"indep_geno", no_argument, nullptr, 'I'},
                                 {"n_boot_rep", required_argument, nullptr, 'b'},
                                 {"boot_block_size", required_argument, nullptr, 'B'},
                                 {"out", required_argument, nullptr, 'o'},
                                 {"n_threads", required_argument, nullptr, 'x'},
                                 {"verbose", required_argument, nullptr, 'V'},
                                 {"seed", required_argument, nullptr, 'r'},
                                 {"n_gpus", required_argument, nullptr, 1001},
                                 {"same_device", no_argument, nullptr, 1005},
                                 {"max_device_bytes", required_argument, nullptr, 1006},
                                 {"single_image", no_argument, nullptr, 1007},
                                 {"two_images", no_argument, nullptr, 1008},
                                 {"device", required_argument, nullptr, 1002},
                                 {"kernel", required_argument, nullptr, 1003},
                                 {"prep", required_argument, nullptr, 1004},
                                 {"stage", required_argument, nullptr, 1009},
                                 {"eager", required_argument, nullptr, 1010},
                                 {nullptr, 0, nullptr, 0}};
  p.seed = (unsigned)time(nullptr);  // parse_args.cpp:35
  int c;
  while ((c = getopt_long_only(argc, argv, "g:pln:s:S:a:A:L:H:cN:C:Ddm:Ib:B:o:x:V:r:", opts, nullptr)) != -1) {
    switch (c) {
      case 'g': p.in_geno = optarg; break;
      case 'p': p.in_probs = true; break;
      case 'l': p.in_logscale = true; p.in_probs = true; break;  // --log_scale implies --probs
      case 'n': p.n_ind = (uint64_t)atol(optarg); break;
      case 's': p.n_sites = (uint64_t)atol(optarg); break;
      case 'S': p.tot_sites = (uint64_t)atol(optarg); break;
      case 'L': p.in_labels = optarg; p.in_labels_header = false; break;
      case 'H': p.in_labels = optarg; p.in_labels_header = true; break;
      case 'a': p.in_pos = optarg; p.in_pos_header = false; break;
      case 'A': p.in_pos = optarg; p.in_pos_header = true; break;
      case 'c': p.call_geno = true; break;
      case 'N': p.N_thresh = atof(optarg); p.call_geno = true; break;    // -N / -C imply --call_geno
      case 'C': p.call_thresh = atof(optarg); p.call_geno = true; break;
      case 'D': p.pairwise_del = true; break;
      case 'd': p.score[4] = 0.5; break;  // --avg_nuc_dist, parse_args.cpp:134-137
      case 'm': p.evol_model = (uint64_t)atol(optarg); break;
      case 'I': p.indep_geno = true; break;
      case 'b': p.n_boot_rep = (uint64_t)atol(optarg); break;
      case 'B': p.boot_block_size = (uint64_t)atol(optarg); break;
      case 'o': p.out = optarg; break;
      case 'x': p.n_threads = (unsigned)atoi(optarg); break;
      case 'V': p.verbose = (unsigned)atoi(optarg); break;
      case 'r': p.seed = (unsigned)atoi(optarg); break;
      case 1001: p.n_gpus = atoi(optarg); break;
      case 1005: p.same_device = true; break;
      case 1006: p.max_device_bytes = strtoull(optarg, nullptr, 10); break;
      case 1010: p.eager = atoi(optarg); break;
      case 1009:  // --stage piece_MiB,ring[,copy share MiB[,drop pages 0|1]]: the load pipeline's geometry (measurement)
        if (sscanf(optarg, "%u,%u,%u,%u", &p.stage_piece, &p.stage_ring, &p.stage_grain, &p.stage_drop) < 2)
          die(__FUNCTION__, "--stage takes piece_MiB,ring[,share_MiB[,drop]]");
        break;
      case 1007: p.single_image = true; break;
      case 1008: p.two_images = true; break;
      case 1002: p.device = atoi(optarg); break;
      case 1003:
        if (!strcmp(optarg, "auto")) p.kernel = NGD_KERNEL_AUTO;
        else if (!strcmp(optarg, "stream")) p.kernel = NGD_KERNEL_STREAM;
        else if (!strcmp(optarg, "mfma")) p.kernel = NGD_KERNEL_MFMA;
        else if (!strcmp(optarg, "em_fast")) p.kernel = NGD_KERNEL_EM_FAST;
        else if (!strcmp(optarg, "em_faithful")) p.kernel = NGD_KERNEL_EM_FAITHFUL;
        else if (!strcmp(optarg, "em_table")) p.kernel = NGD_KERNEL_EM_TABLE;
        else die(__FUNCTION__, "unknown --kernel");
        break;
      case 1004:
        if (!strcmp(optarg, "auto")) p.prep = 0;
        else if (!strcmp(optarg, "host")) p.prep = 1;
        else if (!strcmp(optarg, "device")) p.prep = 2;
        else die(__FUNCTION__, "unknown --prep");
        break;
      default: exit(-1);
    }
  }
  if (p.verbose >= 1) {
    fprintf(stderr, "==> Input Arguments:\n");
    fprintf(stderr,
            "\tgeno: %s\n\tprobs: %s\n\tlog_scale: %s\n\tn_ind: %lu\n\tn_sites: %lu\n\ttot_sites: %lu\n"
            "\tlabels: %s (%s header)\n\tpositions: %s (%s header)\n\tcall_geno: %s\n\tN_thresh: %f\n"
            "\tcall_thresh: %f\n\tpairwise_del: %s\n\tavg_nuc_dist: %s\n\tevol_model: %s\n\tgeno_indep: %s\n"
            "\tn_boot_rep: %lu\n\tboot_block_size: %lu\n\tout: %s\n\tn_threads: %d\n\tverbose: %d\n\tseed: %d\n"
            "\tversion: %s\n\n",
            p.in_geno, p.in_probs ? "true" : "false", p.in_logscale ? "true" : "false", p.n_ind, p.n_sites,
            p.tot_sites, p.in_labels, p.in_labels_header ? "WITH" : "WITHOUT", p.in_pos,
            p.in_pos_header ? "WITH" : "WITHOUT", p.call_geno ? "true" : "false", p.N_thresh, p.call_thresh,
            p.pairwise_del ? "true" : "false", p.score[4] == 0.5 ? "true" : "false",
            p.evol_model <= 6 ? kModelNames[p.evol_model] : "?", p.indep_geno ? "true" : "false", p.n_boot_rep,
            p.boot_block_size, p.out, p.n_threads, p.verbose, p.seed, kVersion);
  }
  if (p.verbose > 4)
    fprintf(stderr, "==> Verbose values greater than 4 for debugging purpose only. Expect large amounts of info on screen\n");
  // parse_args.cpp:203-220
  if (!p.in_geno) die(__FUNCTION__, "genotype input file (--geno) missing!");
  if (p.n_ind == 0) die(__FUNCTION__, "number of individuals (--n_ind) missing!");
  if (p.n_sites == 0) die(__FUNCTION__, "number of sites (--n_sites) missing!");
  if (p.tot_sites > 0 && p.pairwise_del)
    die(__FUNCTION__, "cannot specify total number of sites (--tot_sites) with pairwise deletion (--pairwise_del)!");
  if (p.call_geno && !p.in_probs) die(__FUNCTION__, "can only call genotypes from likelihoods/probabilities!");
  if (p.evol_model > 6) die(__FUNCTION__, "invalid correction method specified!");
  if (p.evol_model > 2 && !p.in_pos)
    die(__FUNCTION__, "use of more complex evolutionary models requires position information!");
  if (!p.out) die(__FUNCTION__, "output prefix (--out) missing!");
  if (p.n_threads < 1) die(__FUNCTION__, "number of threads cannot be less than 1!");
  if (p.n_gpus < 1) die(__FUNCTION__, "number of GPUs cannot be less than 1!");
  if (p.boot_block_size < 1) die(__FUNCTION__, "bootstrap block size cannot be less than 1!");
}

// ---------------------------------------------------------------------------
// text helpers with the reference's semantics
// ---------------------------------------------------------------------------
static gzFile open_gz(const char *name, const char *mode) {  // open_gzfile, gen_func.cpp:208-223
  gzFile fh = strcmp(name, "-") == 0 ? gzdopen(fileno(stdin), mode) : gzopen(name, mode);
  if (fh && gzbuffer(fh, 1 << 20) < 0) return nullptr;
  return fh;
}

static void chomp(char *s) {  // gen_func.cpp:192-199: drops ONE trailing \n or \r
  size_t n = strlen(s);
  if (n && (s[n - 1] == '\n' || s[n - 1] == '\r')) s[n - 1] = '\0';
}

// read_file(), gen_func.cpp:238-283: lines, minus empty ones and ones starting
// with '#', minus `offset` leading ones; a final line without '\n' is lost to
// the gzeof() check exactly as there.
static std::vector<std::string> read_lines(const char *path, uint64_t offset) {
  gzFile fh = open_gz(path, "r");
  if (!fh) die("read_file", "cannot open file!");
  std::vector<std::string> out;
  std::vector<char> buf(kLineBuf);
  for (;;) {
    buf[0] = '\0';
    gzgets(fh, buf.data(), (int)buf.size());
    if (gzeof(fh)) break;
    chomp(buf.data());
    if (buf[0] == '\0' || buf[0] == '#') continue;
    if (offset) { offset--; continue; }
    out.emplace_back(buf.data());
  }
  gzclose(fh);
  return out;
}

// strtod for the common spelling [-+]digits[.digits][e[-+]digits], exactly: up to 19 significant digits are
// gathered in an integer; when it is below 2^53 and the decimal exponent within +-22 the value is one
// correctly rounded multiplication or division by an exact power of ten (Clinger's fast path), which is what a
// correctly rounding strtod returns too.  Anything else (more digits, nan, inf, hex, junk) goes to strtod.
static inline bool fast_strtod(const char *s, size_t len, double *out) {
  static const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char *p = s, *end = s + len;
  bool neg = false;
  if (p < end && (*p == '-' || *p == '+')) neg = *p++ == '-';
  uint64_t m = 0;
  int nd = 0, e10 = 0;
  bool any = false;
  for (; p < end && *p >= '0' && *p <= '9'; p++) {
    any = true;
    if (m || *p != '0') { if (++nd > 19) return false; m = m * 10 + (uint64_t)(*p - '0'); }
  }
  if (p < end && *p == '.') {
    for (p++; p < end && *p >= '0' && *p <= '9'; p++) {
      any = true;
      if (m || *p != '0') { if (++nd > 19) return false; m = m * 10 + (uint64_t)(*p - '0'); }
      e10--;
    }
  }
  if (!any) return false;
  if (p < end && (*p == 'e' || *p == 'E')) {
    p++;
    bool eneg = false;
    if (p < end && (*p == '-' || *p == '+')) eneg = *p++ == '-';
    if (p >= end) return false;
    int ex = 0;
    for (; p < end && *p >= '0' && *p <= '9'; p++) { ex = ex * 10 + (*p - '0'); if (ex > 10000) return false; }
    e10 += eneg ? -ex : ex;
  }
  if (p != end) return false;
  if (m >> 53) return false;
  double v = (double)m;
  if (m == 0) v = 0.0;
  else if (e10 > 0) { if (e10 > 22) return false; v *= p10[e10]; }
  else if (e10 < 0) { if (e10 < -22) return false; v /= p10[-e10]; }
  *out = neg ? -v : v;
  return true;
}

// split(str, " \t", double**), gen_func.cpp:390-417: tokens between separators,
// empty tokens dropped, tokens that strtod does not consume entirely dropped.
static void split_doubles(char *line, const char *sep, std::vector<double> &out) {
  out.clear();
  char *s = line;
  while (s && *s) {
    size_t len = strcspn(s, sep);
    char *next = s[len] ? s + len + 1 : nullptr;
    s[len] = '\0';
    if (len) {
      double v;
      if (fast_strtod(s, len, &v)) out.push_back(v);
      else {
        char *end;
        v = strtod(s, &end);
        if (!*end) out.push_back(v);
      }
    }
    s = next;
  }
}

// ---------------------------------------------------------------------------
// preparation of one (individual, site): read_data.cpp:37-45 / :84-98, then
// ngsDist.cpp:165-174.  Same operation order and libm calls as the reference.
// ---------------------------------------------------------------------------
static inline void post_prob3(double *l) {  // gen_func.cpp:920-932 with logsum :135-151
  double M = l[0];
  for (int i = 1; i < 3; i++) M = (l[i] >= M ? l[i] : M);
  double norm;
  if (M == -INFINITY) {
    norm = -INFINITY;
  } else {
    double sum = 0;
    for (int i = 0; i < 3; i++) sum += exp(l[i] - M);
    norm = log(sum) + M;
  }
  for (int i = 0; i < 3; i++) l[i] -= norm;
}

static inline void call_geno3(double *l, double N_thresh, double call_thresh) {  // gen_func.cpp:886-914
  if (N_thresh > call_thresh) die("call_geno", "missing data threshold must be smaller than calling genotype threshold!");
  int max_pos = 0, min_pos = 0;
  double mx = -INFINITY, mn = INFINITY;
  for (int g = 0; g < 3; g++) if (l[g] > mx) { mx = l[g]; max_pos = g; }
  for (int g = 0; g < 3; g++) if (l[g] < mn) { mn = l[g]; min_pos = g; }
  double max_pp = exp(l[max_pos]);
  if (l[min_pos] == l[max_pos]) max_pp = -1;
  if (max_pp < N_thresh) for (int g = 0; g < 3; g++) l[g] = log((double)1 / 3);
  if (max_pp >= call_thresh) {
    for (int g = 0; g < 3; g++) l[g] = -kInf;
    l[max_pos] = log(1);
  }
}

static inline void finish_prep(const Pars &p, double *l) {  // ngsDist.cpp:165-174
  if (p.call_geno) call_geno3(l, p.N_thresh, p.call_thresh);
  for (int g = 0; g < 3; g++) l[g] = exp(l[g]);
}

// binary element; returns false on NaN (read_data.cpp:42-45)
static inline bool prep_binary(const Pars &p, bool in_logscale, double *l) {
  if (!in_logscale)
    for (int g = 0; g < 3; g++) {
      l[g] = log(l[g]);
      if (l[g] == -INFINITY) l[g] = -kInf;  // conv_space, gen_func.cpp:123-130
    }
  post_prob3(l);
  if (std::isnan(l[0]) || std::isnan(l[1]) || std::isnan(l[2])) return false;
  finish_prep(p, l);
  return true;
}

template <typename F>
static void parallel_for(unsigned n_threads, uint64_t n, uint64_t min_n, F fn) {
  if (n_threads <= 1 || n < min_n || n < 2) { fn(0, n); return; }
  std::vector<std::thread> th;
  uint64_t per = (n + n_threads - 1) / n_threads;
  for (unsigned t = 0; t < n_threads; t++) {
    uint64_t lo = t * per, hi = std::min(n, lo + per);
    if (lo >= hi) break;
    th.emplace_back(fn, lo, hi);
  }
  for (auto &t : th) t.join();
}

// The two parallel phases of a text group (tokenise, then parse + prepare) run a hundred times per load: their threads
// are kept ([measured] 541 MB of GL text, 16 threads: 0.29 s in the two phases with a thread spawned per share and call,
// of which 0.11 s is the work).  One user at a time (the text loader's thread); shares are handed out dynamically.
struct TextPool {
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  std::vector<std::thread> workers;
  std::function<void(uint64_t, uint64_t)> job;
  uint64_t n = 0, per = 1, next = 0, done = 0, generation = 0;
  bool quit = false;
  explicit TextPool(unsigned n_threads) {
    for (unsigned t = 1; t < n_threads; t++)
      workers.emplace_back([this]() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
          cv_go.wait(lk, [&] { return quit || generation != seen; });
          if (quit) return;
          seen = generation;
          work(lk);
        }
      });
  }
  ~TextPool() {
    { std::lock_guard<std::mutex> lk(m); quit = true; }
    cv_go.notify_all();
    for (auto &w : workers) w.join();
  }
  void work(std::unique_lock<std::mutex> &lk) {  // takes shares until none is left; lk held on entry and exit
    while (next < n) {
      const uint64_t lo = next, hi = std::min(n, lo + per);
      next = hi;
      lk.unlock();
      job(lo, hi);
      lk.lock();
      done += hi - lo;
      if (done == n) cv_done.notify_all();
    }
  }
  template <typename F>
  void run(uint64_t count, F fn) {
    if (workers.empty() || count < 2) { fn(0, count); return; }
    std::unique_lock<std::mutex> lk(m);
    job = fn;
    n = count;
    per = std::max<uint64_t>(1, count / (4 * (workers.size() + 1)));
    next = done = 0;
    generation++;
    cv_go.notify_all();
    work(lk);
    cv_done.wait(lk, [&] { return done == n; });
    n = 0;
  }
};

// ---------------------------------------------------------------------------
// BGZF text input (the blocked gzip that bgzip / htslib / ANGSD write: a series of gzip members of at most 64 KB, each
// with its compressed size in a 'BC' extra field).  zlib's gzread -- the reference's reader, gen_func.cpp:208-223 --
// inflates such a file like any other .gz, one member after the other on one thread; the members being independent, here
// they are inflated on --n_threads threads, a few hundred at a time, CRCs checked as gzread checks them.  gets() has
// gzgets' semantics, so everything downstream (line splitting, headers, premature / missing EOF) is the same code.
struct BgzfText {
  int fd = -1;
  uint64_t size = 0, off = 0;  // file size; offset of the first block not yet inflated
  std::vector<unsigned char> comp;
  std::vector<char> text;      // the inflated blocks of the current batch
  size_t cur = 0;
  bool hit_eof = false, bad = false;
  unsigned n_threads = 1;
  static constexpr size_t kBatch = 16u << 20;  // compressed bytes per batch

  // size of the block at q (at least 18 bytes readable), 0 if it is not a BGZF block
  static uint32_t block_size(const unsigned char *q, size_t avail) {
    if (avail < 18 || q[0] != 0x1f || q[1] != 0x8b || q[2] != 8 || !(q[3] & 4)) return 0;
    const uint32_t xlen = q[10] | (uint32_t)q[11] << 8;
    if (12 + (size_t)xlen > avail) return 0;
    for (uint32_t x = 0; x + 4 <= xlen;) {
      const unsigned char *f = q + 12 + x;
      const uint32_t slen = f[2] | (uint32_t)f[3] << 8;
      if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) return (f[4] | (uint32_t)f[5] << 8) + 1u;
      x += 4 + slen;
    }
    return 0;
  }
  static BgzfText *open_if_bgzf(const char *path, unsigned n_threads) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    unsigned char head[64];
    const ssize_t n = pread(fd, head, sizeof(head), 0);
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || n < 18 || block_size(head, (size_t)n) < 26) {
      close(fd);
      return nullptr;
    }
    BgzfText *b = new BgzfText();
    b->fd = fd;
    b->size = (uint64_t)st.st_size;
    b->n_threads = std::max(1u, n_threads);
    return b;
  }
  ~BgzfText() { if (fd >= 0) close(fd); }

  // inflates the next batch of blocks into `text`; false when the file is used up (or damaged: bad)
  bool fill() {
    text.clear();
    cur = 0;
    while (text.empty()) {
      if (off >= size || bad) return false;
      const size_t want = (size_t)std::min<uint64_t>(kBatch, size - off);
      comp.resize(want);
      size_t got = 0;
      while (got < want) {
        const ssize_t r = pread(fd, comp.data() + got, want - got, (off_t)(off + got));
        if (r <= 0) { bad = true; return false; }
        got += (size_t)r;
      }
      struct Blk { size_t in, in_len, out; uint32_t isize, crc; };
      std::vector<Blk> blks;
      size_t q = 0, total = 0;
      while (q + 18 <= want) {
        const uint32_t bs = block_size(comp.data() + q, want - q);
        if (!bs) { bad = true; return false; }  // not a BGZF block where one must start
        if (q + bs > want) break;               // the batch ends inside this block: it opens the next batch
        const uint32_t xlen = comp[q + 10] | (uint32_t)comp[q + 11] << 8;
        if (bs < 20 + xlen) { bad = true; return false; }
        const unsigned char *t = comp.data() + q + bs - 8;
        Blk b;
        b.in = q + 12 + xlen;
        b.in_len = bs - 20 - xlen;
        b.crc = t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        b.isize = t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
        if (b.isize > (1u << 16)) { bad = true; return false; }
        b.out = total;
        total += b.isize;
        blks.push_back(b);
        q += bs;
      }
      if (q == 0) { bad = true; return false; }  // a block larger than what is left of the file
      text.resize(total);
      std::atomic<int> err{0};
      parallel_for(n_threads, blks.size(), 8, [&](uint64_t lo, uint64_t hi) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) { err = 1; return; }
        for (uint64_t k = lo; k < hi && !err; k++) {
          const Blk &b = blks[k];
          if (!b.isize) continue;
          if (inflateReset(&zs) != Z_OK) { err = 1; break; }  // (one state per thread, re-armed for every block)
          zs.next_in = comp.data() + b.in;
          zs.avail_in = (uInt)b.in_len;
          zs.next_out = (Bytef *)text.data() + b.out;
          zs.avail_out = b.isize;
          const int rc = inflate(&zs, Z_FINISH);
          if (rc != Z_STREAM_END || zs.avail_out != 0 ||
              crc32(crc32(0L, Z_NULL, 0), (const Bytef *)text.data() + b.out, b.isize) != b.crc)
            err = 1;
        }
        inflateEnd(&zs);
      });
      if (err) { bad = true; text.clear(); return false; }
      off += q;
    }
    return true;
  }
  char *gets(char *buf, int len) {  // gzgets: up to len-1 characters, through the first newline
    int w = 0;
    while (w < len - 1) {
      if (cur == text.size() && !fill()) { hit_eof = true; break; }
      const char *src = text.data() + cur;
      const size_t want = std::min(text.size() - cur, (size_t)(len - 1 - w));
      const char *nl = (const char *)memchr(src, '\n', want);
      const size_t take = nl ? (size_t)(nl - src) + 1 : want;
      memcpy(buf + w, src, take);
      w += (int)take;
      cur += take;
      if (nl) break;
    }
    if (!w) return nullptr;
    buf[w] = '\0';
    return buf;
  }
  // The next line as gzgets(buf, max_chars + 1) + chomp() would deliver it (a longer line comes in pieces, ONE trailing
  // \n or \r is dropped), copied once, straight out of the inflated batch.
  bool next_line(std::string &out, size_t max_chars) {
    out.clear();
    bool any = false;
    while (out.size() < max_chars) {
      if (cur == text.size() && !fill()) { hit_eof = true; break; }
      const char *src = text.data() + cur;
      const size_t want = std::min(text.size() - cur, max_chars - out.size());
      const char *nl = (const char *)memchr(src, '\n', want);
      const size_t take = nl ? (size_t)(nl - src) + 1 : want;
      out.append(src, take);
      cur += take;
      any = true;
      if (nl) break;
    }
    if (!any) return false;
    const size_t z = out.find('\0');  // what strlen() in chomp() and the std::string made of the C string would see
    if (z != std::string::npos) out.resize(z);
    if (!out.empty() && (out.back() == '\n' || out.back() == '\r')) out.pop_back();
    return true;
  }
  bool eof() const { return hit_eof && !bad; }
};

// ---------------------------------------------------------------------------
// one engine, destroyed with its scope
struct Engine {
  ngd_engine *h = nullptr;
  Engine() = default;
  Engine(const Engine &) = delete;
  Engine &operator=(const Engine &) = delete;
  void upload_sites(const double *p, uint64_t s0, uint64_t n) {
    int rc = ngd_upload_sites(h, p, s0, n);
    if (rc) die_engine("upload", rc);
  }
  void commit() {
    int rc = ngd_commit(h);
    if (rc == NGD_E_NAN) die("read_geno", "NaN found! Is the file format correct?");
    if (rc) die_engine("commit", rc);
  }
  // The run's LAST engine is not taken apart before the process exits: its device memory goes back to the driver with the
  // process ([measured, round 6] cfg 3: unmapping and releasing 34 GB of mapped pieces is 26 ms of a 0.75 s command, of which
  // the exit itself takes back 8).  Builds under AddressSanitizer destroy it all the same (the leak checker's business).
  bool leave_to_exit = false;
  ~Engine() {
#if defined(__SANITIZE_ADDRESS__)
    leave_to_exit = false;
#endif
    if (h && !leave_to_exit) ngd_destroy(h);
  }
};

// read_geno(), read_data.cpp:13-116, fused with the preparation and the upload.  The input is consumed front to
// back in one or more parts (ranges of sites: one per device, or several when the data set is larger than the
// devices): load() puts the next n_part sites into an engine as its sites 0 .. n_part-1 and commits them.  A plain
// binary file can also be entered at any site (first_site), so that every device's thread reads its own ranges.
struct Loader {
  const Pars &p;
  gzFile fh = nullptr;
  std::unique_ptr<BgzfText> bg;  // text input that turned out to be BGZF: inflated on several threads instead of by fh
  char *text_gets(char *buf, int len) { return bg ? bg->gets(buf, len) : gzgets(fh, buf, len); }
  bool text_eof() { return bg ? bg->eof() : gzeof(fh) != 0; }
  int raw_fd = -1;
  uint64_t raw_off = 0, raw_size = 0;
  uint64_t done = 0;  // sites of the whole input consumed by earlier parts
  bool eof = false;
  explicit Loader(const Pars &pars, uint64_t first_site = 0);
  bool seekable() const { return raw_fd >= 0; }
  void load(Engine &eng, uint64_t n_part, bool last_part);
  void finish(bool check_eof = true);
};

Loader::Loader(const Pars &pars, uint64_t first_site) : p(pars) {
  fh = open_gz(p.in_geno, p.in_bin ? "rb" : "r");
  if (!fh) die("read_geno", "cannot open GENO file!");
  if (!p.in_bin && strcmp(p.in_geno, "-") != 0 && p.n_threads > 1) {  // (on one thread gzread does the same job)
    bg.reset(BgzfText::open_if_bgzf(p.in_geno, p.n_threads));
    if (bg && p.verbose >= 2) fprintf(stderr, "> BGZF input: blocks are inflated on %u thread(s)\n", bg->n_threads);
  }
  // A binary GL file that is a plain regular file (the usual case; gzread would only copy it through) is read
  // with pread() on several threads straight into the destination: one thread moves ~12 GB/s out of the page
  // cache, which is less than the copy to the device and the preparation kernel take.
  if (p.in_bin && strcmp(p.in_geno, "-") != 0) {
    struct stat st;
    int fd = open(p.in_geno, O_RDONLY);
    unsigned char magic[2] = {0, 0};
    if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && pread(fd, magic, 2, 0) >= 0 &&
        !(magic[0] == 0x1f && magic[1] == 0x8b)) {
      raw_fd = fd;
      raw_size = (uint64_t)st.st_size;
      raw_off = first_site * p.n_ind * 24;
      done = first_site;
    } else if (fd >= 0) {
      close(fd);
    }
  }
}

void Loader::load(Engine &eng, uint64_t n_part, bool last_part) {
  const uint64_t n_ind = p.n_ind, n_sites = n_part;
  const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(n_sites, (64ull << 20) / (n_ind * 24)));
  bool in_logscale = p.in_logscale;
  const bool device_prep = p.in_bin && (p.prep == 2 || (p.prep == 0 && !p.call_geno));
  std::vector<double> buf(device_prep ? 0 : chunk * n_ind * 3);
  // (12, not 16: with the engine's own threads 16 copying threads overrun a CPU quota of 16 and the whole process is
  // throttled for a few ms every 100 ms -- [measured] gaps of 2-7 ms between copies at 100-ms intervals; 8-20 threads load
  // cfg 3 in the same 0.51-0.53 s, the copy engine being the pace)
  const unsigned n_io = std::min(12u, std::max(4u, p.n_threads));
  auto read_exact = [&](double *dst, uint64_t bytes) {
    if (raw_fd >= 0) {
      if (raw_off + bytes > raw_size) die("read_geno", "GENO file at premature EOF. Check GENO file and number of sites!");
      std::atomic<bool> bad{false};
      parallel_for(n_io, bytes, 8u << 20, [&](uint64_t lo, uint64_t hi) {
        while (lo < hi) {
          ssize_t r = pread(raw_fd, (char *)dst + lo, hi - lo, (off_t)(raw_off + lo));
          if (r <= 0) { bad = true; return; }
          lo += (uint64_t)r;
        }
      });
      if (bad) die("read_geno", "cannot read binary GENO file. Check GENO file and number of sites!");
      raw_off += bytes;
      return;
    }
    uint64_t got = 0;
    while (got < bytes) {
      int r = gzread(fh, (char *)dst + got, (unsigned)std::min<uint64_t>(bytes - got, 1u << 30));
      if (r <= 0) break;
      got += (uint64_t)r;
    }
    if (got != bytes) {
      if (gzeof(fh)) die("read_geno", "GENO file at premature EOF. Check GENO file and number of sites!");
      die("read_geno", "cannot read binary GENO file. Check GENO file and number of sites!");
    }
  };
  if (p.in_bin && device_prep) {
    // Straight into the engine's ring of pinned buffers; the copies to the device and the preparation kernel overlap the
    // filling of the next buffers.  A regular file is MAPPED and its pages copied by a few kept threads: [measured, round
    // 6, tools/host_read_pipeline, 8.6 GB in the page cache / in tmpfs] pread() on 16 threads moves 116 GB/s alone but
    // 27-47 GB/s while the copy engine reads host memory beside it; memcpy() out of a mapping 55-57 GB/s = the link's
    // 57.6 (first touch of the mapping's pages included).  Handing the mapping itself to the copy engine (no host copy at
    // all) works but pins pages at 16-50 GB/s, and is not used.  (A file truncated by someone else during the run is a
    // SIGBUS here, where pread would have reported a premature EOF.)
    // The pages a thread has copied it takes out of the page table again at once (MADV_DONTNEED on a shared file mapping
    // drops the translations, nothing else): left to one munmap() at the end -- or to the kernel at exit -- the 6 million
    // 4-KB translations of 24 GB cost 0.3-0.5 s of ONE core ([measured] cfg 3: load 0.89 s, the copies themselves 0.50 s);
    // a window mapped and unmapped per piece instead makes the copies three times slower (0.72-1.6 s).
    const char *map = nullptr;
    void *map_base = nullptr;
    uint64_t map_len = 0;
    const uint64_t pg = (uint64_t)sysconf(_SC_PAGESIZE);
    if (raw_fd >= 0) {
      if (raw_off + n_sites * n_ind * 24 > raw_size) die("read_geno", "GENO file at premature EOF. Check GENO file and number of sites!");
      const uint64_t lo = raw_off / pg * pg;
      map_len = raw_off + n_sites * n_ind * 24 - lo;
      void *m = mmap(nullptr, map_len, PROT_READ, MAP_SHARED, raw_fd, (off_t)lo);
      if (m != MAP_FAILED) { map_base = m; map = (const char *)m - lo; }  // (map + file offset = that byte; else: pread as before)
    }
    const bool mapped = map != nullptr;
    std::unique_ptr<TextPool> pool;
    if (mapped) pool.reset(new TextPool(n_io));
    auto copy_mapped = [&](double *dst, uint64_t bytes) {
      const uint64_t grain = (uint64_t)std::max(1u, p.stage_grain) << 20, n_g = (bytes + grain - 1) / grain;
      const char *src = map + raw_off;
      pool->run(n_g, [&](uint64_t g0, uint64_t g1) {
        const uint64_t b0 = g0 * grain, b1 = std::min(bytes, g1 * grain);
        memcpy((char *)dst + b0, src + b0, b1 - b0);
        // whole pages inside [b0, b1): a page shared with a neighbouring share stays (its owner may not have read it yet)
        const uint64_t a0 = ((uint64_t)(uintptr_t)(src + b0) + pg - 1) / pg * pg, a1 = (uint64_t)(uintptr_t)(src + b1) / pg * pg;
        if (a1 > a0 && p.stage_drop) madvise((void *)(uintptr_t)a0, a1 - a0, MADV_DONTNEED);
      });
      raw_off += bytes;
    };
    ngd_prep pr;
    pr.in_logscale = in_logscale; pr.call_geno = p.call_geno; pr.N_thresh = p.N_thresh; pr.call_thresh = p.call_thresh;
    double t_acq = 0, t_read = 0, t_sub = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double>(b - a).count();
    };
    for (uint64_t s0 = 0; s0 < n_sites;) {
      double *pin; uint64_t cap;
      auto t0 = now();
      int rc = ngd_stage_acquire(eng.h, &pin, &cap);
      if (rc) die_engine("ngd_stage_acquire", rc);
      const uint64_t n = std::min(cap, n_sites - s0);
      auto t1 = now();
      if (mapped) copy_mapped(pin, n * n_ind * 24);
      else read_exact(pin, n * n_ind * 24);
      auto t2 = now();
      t_acq += secs(t0, t1); t_read += secs(t1, t2);
      auto t3 = now();
      rc = ngd_stage_submit(eng.h, s0, n, &pr);
      if (rc) die_engine("ngd_stage_submit", rc);
      t_sub += secs(t3, now());
      s0 += n;
    }
    if (p.verbose >= 2)
      fprintf(stderr, "> staged load: waiting for a free pinned buffer %.3f s, reading %.3f s (%s), submitting %.3f s\n", t_acq,
              t_read, mapped ? "memcpy out of mapped windows of the file, on several threads" : "gzread", t_sub);
    if (map_base) {
      // (the mapping's pages are gone already; what is left are its page-table pages, 12 000 of them for 24 GB: ~9 ms
      // of one core, beside the distances instead of before them)
      void *mb = map_base;
      const uint64_t ml = map_len;
      std::thread([mb, ml]() { munmap(mb, ml); }).detach();
    }
    g_phases.add("of_load_wait_buffer", t_acq);
    g_phases.add("of_load_read", t_read);
    g_phases.add("of_load_submit", t_sub);
  } else if (p.in_bin) {
    for (uint64_t s0 = 0; s0 < n_sites; s0 += chunk) {
      const uint64_t n = std::min(chunk, n_sites - s0);
      read_exact(buf.data(), n * n_ind * 24);
      std::atomic<bool> bad{false};
      parallel_for(p.n_threads, n * n_ind, 4096, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t e = lo; e < hi; e++)
          if (!prep_binary(p, in_logscale, &buf[3 * e])) bad = true;
      });
      if (bad) die("read_geno", "NaN found! Is the file format correct?");
      eng.upload_sites(buf.data(), s0, n);
    }
  } else {
    // Text: decompression and line splitting are sequential; tokenising, number parsing and the
    // per-(individual, site) preparation of a group of lines run on --n_threads threads.  Which lines are
    // sites (empty line = site at its -INF fill; line without numbers, or a short FIRST line = header) is
    // settled in file order between the two parallel phases, so errors and messages come in the reference's order.
    const uint64_t n_geno = p.in_probs ? 3 : 1;
    const uint64_t need = n_ind * n_geno;
    std::vector<char> line(std::max<size_t>(kLineBuf, need * 32 + 4096));
    const uint64_t group = std::max<uint64_t>(1, std::min<uint64_t>(chunk, 1024));
    std::vector<std::string> lines, ahead;
    std::vector<std::vector<double>> toks(group);
    std::vector<int64_t> slot(group);  // site slot within the group, or -1
    uint64_t s = 0;
    // up to `max_lines` lines of the stream (decompression + line splitting: the sequential part of a text load)
    // (*hit_end: the stream ended before max_lines were read)
    auto read_lines_into = [&](std::vector<std::string> &out, uint64_t max_lines, bool *hit_end) {
      out.clear();
      *hit_end = false;
      if (bg) {
        std::string one;
        while (out.size() < max_lines) {
          if (!bg->next_line(one, line.size() - 1)) { *hit_end = true; break; }
          out.emplace_back(std::move(one));
          one.clear();  // (moved from: valid but unspecified)
        }
        return;
      }
      while (out.size() < max_lines) {
        if (text_gets(line.data(), (int)line.size()) == nullptr) { *hit_end = true; break; }
        chomp(line.data());
        out.emplace_back(line.data());
      }
    };
    TextPool pool(p.n_threads);
    bool have_ahead = false, end_ahead = false;
    double t_first = 0, t_work = 0, t_up = 0, t_join = 0;  // --verbose 2: where a text load spends its time
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
    while (s < n_sites) {
      if (have_ahead) {
        lines.swap(ahead);
        eof = end_ahead;
        have_ahead = false;
      } else {
        // a little read-ahead for header lines -- in the last part only: lines past a part belong to the next one
        const auto t0 = now();
        read_lines_into(lines, std::min<uint64_t>(group, n_sites - s + (last_part ? 64 : 0)), &eof);
        t_first += secs(t0);
      }
      const auto t_w0 = now();
      // The NEXT group is decompressed by a reader thread while this one is tokenised, parsed and uploaded.  It may
      // take only lines that belong to this part whatever this group turns out to hold: at least n_sites - s - (lines
      // of this group) sites are still to come after it.
      std::thread reader;
      {
        const uint64_t sure = n_sites - s > lines.size() ? n_sites - s - lines.size() : 0;
        const uint64_t next_max = std::min<uint64_t>(group, sure + (last_part && sure ? 64 : 0));
        if (next_max && !eof && !lines.empty()) {
          reader = std::thread([&, next_max]() { read_lines_into(ahead, next_max, &end_ahead); });
          have_ahead = true;
        }
      }
      struct Joiner {
        std::thread &t;
        double &acc;
        ~Joiner() {
          const auto t0 = std::chrono::steady_clock::now();
          if (t.joinable()) t.join();
          acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
      } joiner{reader, t_join};
      if (lines.empty()) {
        if (text_eof()) die("read_geno", "GENO file at premature EOF. Check GENO file and number of sites!");
        die("read_geno", "cannot read GZip GENO file. Check GENO file and number of sites!");
      }
      pool.run(lines.size(), [&](uint64_t lo, uint64_t hi) {
        for (uint64_t k = lo; k < hi; k++)
          if (lines[k].empty()) toks[k].clear();
          else split_doubles(&lines[k][0], " \t", toks[k]);
      });
      uint64_t filled = 0;
      for (uint64_t k = 0; k < lines.size(); k++) {
        slot[k] = -1;
        if (s + filled == n_sites)  // anything after the last site: the reference stops reading before it
          die("read_geno", "GENO file not at EOF. Check GENO file and number of sites!");
        if (!lines[k].empty()) {
          if (toks[k].empty() || (done + s + filled == 0 && toks[k].size() < need)) {  // header
            fprintf(stderr, "> Header found! Skipping line...\n");
            continue;
          }
          if (toks[k].size() < need) die("read_geno", "wrong GENO file format. Less fields than expected!");
        }
        slot[k] = (int64_t)filled++;
      }
      std::atomic<int> bad_geno{0};
      pool.run(lines.size(), [&](uint64_t lo, uint64_t hi) {
        for (uint64_t k = lo; k < hi; k++) {
          if (slot[k] < 0) continue;
          double *dst = &buf[(uint64_t)slot[k] * n_ind * 3];
          if (lines[k].empty()) {
            // empty line: the site keeps its -INF fill (read_data.cpp:21,58-59)
            for (uint64_t i = 0; i < n_ind; i++) {
              double *l = dst + 3 * i;
              l[0] = l[1] = l[2] = -kInf;
              finish_prep(p, l);
            }
            continue;
          }
          const double *ptr = toks[k].data() + (toks[k].size() - need);  // last n_ind*n_geno columns
          for (uint64_t i = 0; i < n_ind; i++) {
            double *l = dst + 3 * i;
            if (p.in_probs) {
              for (int g = 0; g < 3; g++) l[g] = in_logscale ? ptr[3 * i + g] : log(ptr[3 * i + g]);
            } else {
              l[0] = l[1] = l[2] = -kInf;
              int g = (int)ptr[i];
              if (g >= 0) {
                if (g > 2) { bad_geno = 1; g = 2; }
                l[g] = log(1);
              } else {
                l[0] = l[1] = l[2] = log((double)1 / 3);
              }
            }
            post_prob3(l);
            finish_prep(p, l);
          }
        }
      });
      if (bad_geno) die("read_geno", "wrong GENO file format. Genotypes must be coded as {-1,0,1,2} !");
      t_work += secs(t_w0);
      const auto t_u0 = now();
      if (filled) eng.upload_sites(buf.data(), s, filled);
      t_up += secs(t_u0);
      s += filled;
      if (eof && s < n_sites) die("read_geno", "GENO file at premature EOF. Check GENO file and number of sites!");
    }
    if (p.verbose >= 2)
      fprintf(stderr, "> text load: first group read %.3f s, tokenise + parse + prepare %.3f s, uploads %.3f s, waiting for the "
              "reader thread (inflate + line split of the next group) %.3f s\n", t_first, t_work, t_up, t_join);
  }
  done += n_part;
  {
    const auto t_c = std::chrono::steady_clock::now();
    eng.commit();
    g_phases.add("of_load_commit", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_c).count());
  }
}

void Loader::finish(bool check_eof) {
  if (raw_fd >= 0) {
    if (check_eof && raw_off != raw_size) die("read_geno", "GENO file not at EOF. Check GENO file and number of sites!");
    close(raw_fd);
  } else {
    char one[2];
    if (bg) {
      if (bg->gets(one, 2) != nullptr || !bg->eof())
        die("read_geno", bg->bad ? "cannot read GZip GENO file. Check GENO file and number of sites!"
                                 : "GENO file not at EOF. Check GENO file and number of sites!");
    } else {
    gzread(fh, one, 1);
    if (!gzeof(fh)) die("read_geno", "GENO file not at EOF. Check GENO file and number of sites!");
    }
  }
  gzclose(fh);
}

int main(int argc, char **argv) {
  Pars pars;
  parse_cmd_args(pars, argc, argv);
  Pars &p = pars;

  const uint64_t n_comb = ngd_n_pairs(p.n_ind);
  if (p.verbose >= 1) fprintf(stderr, "==> Analysis will be run in %lu combinations\n", n_comb);
  // ngsDist.cpp:55-65
  if (!p.in_probs && !p.indep_geno) {
    fprintf(stderr, "==> Using faster algorithm (assuming independence of genotypes) since input are genotypes!\n");
    p.indep_geno = true;
  } else if (p.call_geno && !p.indep_geno) {
    fprintf(stderr, "==> Using faster algorithm (assuming independence of genotypes) since calling genotypes!\n");
    p.indep_geno = true;
  } else if (p.indep_geno && p.verbose >= 1) {
    fprintf(stderr, "==> Using faster algorithm (assuming independence of genotypes)!\n");
  }
  // ngsDist.cpp:73-95
  if (strcmp(p.in_geno, "-") == 0) {
    if (p.verbose >= 1) fprintf(stderr, "==> Reading from STDIN (BINARY)\n");
    p.in_bin = true;
  } else {
    struct stat st;
    if (stat(p.in_geno, &st) != 0) die(__FUNCTION__, "cannot check GENO file size!");
    const char *dot = strrchr(p.in_geno, '.');
    if (dot && strcmp(dot, ".gz") == 0) {
      if (p.verbose >= 1) fprintf(stderr, "==> GZIP input file (never BINARY)\n");
      p.in_bin = false;
    } else {
      if (p.verbose >= 1) fprintf(stderr, "==> BINARY input file\n");
      p.in_bin = true;
      p.in_probs = true;
      if (p.n_sites != (uint64_t)st.st_size / sizeof(double) / p.n_ind / 3)
        die(__FUNCTION__, "invalid/corrupt genotype input file!");
    }
  }
  if (p.evol_model > 2) {  // gen_dist() would error() on the first pair, ngsDist.cpp:387-398
    static const char *msg[] = {"K80 model not yet supported", "F81 model not yet supported",
                                "HKY85 model not yet supported", "TN93 model not yet supported"};
    die("gen_dist", msg[p.evol_model - 3]);
  }

  // labels, ngsDist.cpp:103-125
  std::vector<std::string> labels;
  if (p.in_labels) {
    if (p.verbose >= 1) fprintf(stderr, "==> Reading labels\n");
    labels = read_lines(p.in_labels, p.in_labels_header ? 1 : 0);
    if (labels.size() != p.n_ind) die(__FUNCTION__, "invalid LABELS file!");
    for (auto &l : labels) {
      size_t tab = l.find('\t');
      if (tab != std::string::npos) l.resize(tab);
    }
  } else {
    for (uint64_t i = 0; i < p.n_ind; i++) labels.push_back("Ind_" + std::to_string(i));
  }
  if (p.verbose >= 4) for (auto &l : labels) fprintf(stderr, "%s\n", l.c_str());

  // positions: validated, never used (models 3-6 are unimplemented), ngsDist.cpp:133-149
  if (p.in_pos) {
    if (p.verbose >= 1) fprintf(stderr, "==> Reading positions file\n");
    std::vector<std::string> pos = read_lines(p.in_pos, p.in_pos_header ? 1 : 0);
    uint64_t n_fields = 0;
    for (auto &l : pos) {
      uint64_t n = 1 + (uint64_t)std::count(l.begin(), l.end(), '\t');
      if (!l.empty() && l.back() == '\t') n--;  // no trailing empty field (_strtok, gen_func.cpp:305-322)
      if (n_fields == 0) n_fields = n;
      if (n != n_fields) die("read_split", "invalid number of fields in file!");
    }
    if (pos.size() != p.n_sites || n_fields < 2) die(__FUNCTION__, "invalid POS file!");
  }

  g_phases.mark("args_labels");
  // devices: --n_gpus of them, the site axis split over them
  int n_dev = ngd_device_count();
  if (n_dev < 1) die(__FUNCTION__, "no HIP device found (this program has no CPU path)");
  if (p.device + (p.same_device ? 1 : p.n_gpus) > n_dev) die(__FUNCTION__, "not enough HIP devices for --device/--n_gpus");
  bool eager_ranges = false;  // (set below once it is known whether the job goes through in site ranges)
  auto make_engine = [&](Engine &eng, uint64_t n_sites_part, int dev_index) {
    ngd_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.n_ind = p.n_ind;
    cfg.n_sites = n_sites_part;
    memcpy(cfg.score, p.score, sizeof(cfg.score));
    cfg.pairwise_del = p.pairwise_del;
    cfg.indep_geno = p.indep_geno;
    cfg.device = p.same_device ? p.device : p.device + dev_index;
    cfg.kernel = p.kernel;
    // (the MFMA kernel on one operand image, on two, or -- 0 -- as the engine chooses; nothing to the other kernels)
    cfg.single_image = p.single_image ? 2 : p.two_images ? 3 : 0;
    int rc = ngd_create(&cfg, &eng.h);
    if (rc) die_engine("ngd_create", rc);
    // the first engine call after the load is the plain full-data pass (no bootstrap; or site ranges, whose engines always
    // begin with it): on the EM path, where that pass is several times the load, it starts beside the load
    // (NGD_OPT_EAGER_FULL: [measured] cfg 4 2.96 -> 2.78 s end to end; --eager 0 switches it off, --eager 2 also takes it
    // for --indep_geno, where a 46 ms pass beside a 0.5 s load gains nothing measurable)
    if (p.eager && (!p.indep_geno || p.eager >= 2) && (p.n_boot_rep == 0 || eager_ranges) &&
        (rc = ngd_set_option(eng.h, NGD_OPT_EAGER_FULL, 1)))
      die_engine("ngd_set_option", rc);
    if (p.stage_piece && (rc = ngd_set_option(eng.h, NGD_OPT_STAGE_PIECE_MIB, p.stage_piece))) die_engine("ngd_set_option", rc);
    if (p.stage_ring && (rc = ngd_set_option(eng.h, NGD_OPT_STAGE_RING, p.stage_ring))) die_engine("ngd_set_option", rc);
  };

  // Does the data set fit ONE device?  Resident bytes per site: both operand images (one on the EM path, the
  // individual-major copy for the streaming kernel), masks and bootstrap weights; plus slabs and results.  If it
  // does and one device is asked for, one engine runs the whole job (ngd_run_job).  Otherwise gen_dist()'s sums are
  // split along the SITE axis (include/ngsdist_amd.h "Site sharding"): ranges of sites, each read, held and computed
  // by one device -- side by side on --n_gpus devices, one after the other where the devices are too small -- and
  // the per-range (sum, cnt) are added.
  const uint64_t n_pad = (p.n_ind + 127) / 128 * 128;
  const bool mfma_path = p.indep_geno && p.kernel != NGD_KERNEL_STREAM;
  // (one image + the fix-up pass's side array, 24 n_pad + 8 n_ind: --single_image, or the engine's own choice above 384
  // padded individuals -- ngd_config.single_image = 0; this program's score matrices are the reference's two)
  const bool one_image = mfma_path && !p.two_images && (p.single_image || n_pad > 384);
  const uint64_t per_site = (p.kernel == NGD_KERNEL_STREAM ? 24 * p.n_ind
                             : one_image ? 24 * n_pad + 8 * p.n_ind : (mfma_path ? 48 : 24) * n_pad) +
                            (p.pairwise_del ? p.n_ind / 8 + 1 : 0) + 40;
  const uint64_t n_t = n_pad / 128, n_slabs = std::max<uint64_t>(8, std::min<uint64_t>(256, 8192 / (n_t * (n_t + 1) / 2)));
  const uint64_t fixed = n_slabs * n_pad * n_pad * 8 + n_comb * 64 + (512ull << 20);
  uint64_t dev_free = 0, dev_total = 0;
  if (ngd_device_memory(p.device, &dev_free, &dev_total)) die_engine("ngd_device_memory", -1);
  g_phases.mark("first_hip_calls");
  if (p.same_device) dev_free /= (uint64_t)p.n_gpus;  // the rehearsal's ranges share one device
  const uint64_t budget = p.max_device_bytes ? p.max_device_bytes : dev_free / 100 * 85;
  const bool in_parts = p.n_gpus > 1 || fixed + per_site * p.n_sites > budget;
  eager_ranges = in_parts;

  if (p.verbose >= 2) fprintf(stderr, "==> Setting seed for random number generator\n");
  uint32_t rng[3];
  ngd_taus_seed(rng, p.seed);  // gsl_rng_alloc(gsl_rng_taus) + gsl_rng_set, ngsDist.cpp:179-180

  FILE *out_fh = fopen(p.out, "w");
  if (!out_fh) die(__FUNCTION__, "cannot open output file!");

  double t_compute = 0, t_write = 0;
  std::vector<double> dist(n_comb);
  std::vector<const char *> label_ptr;
  for (auto &l : labels) label_ptr.push_back(l.c_str());
  // A matrix's text is written by a thread of its own while the next matrix is fetched, finished and formatted: two
  // buffers ([measured, round 6] 101 matrices of 1000 individuals, 1.3 GB of text: 8.3 ms a matrix with the write inline).
  std::vector<char> text_buf[2];
  struct Writer {
    FILE *fh = nullptr;
    std::mutex m;
    std::condition_variable cv;
    const char *data = nullptr;
    size_t len = 0;
    bool quit = false, failed = false, busy = false;
    std::thread th;
    void start(FILE *f) {
      fh = f;
      th = std::thread([this]() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
          cv.wait(lk, [&] { return quit || data; });
          if (!data) return;
          const char *d = data;
          const size_t n = len;
          lk.unlock();
          const bool ok = fwrite(d, 1, n, fh) == n;
          lk.lock();
          if (!ok) failed = true;
          data = nullptr;
          busy = false;
          cv.notify_all();
        }
      });
    }
    void wait_idle() {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return !busy; });
    }
    void submit(const char *d, size_t n) {  // (the caller has waited for the writer to be idle)
      std::lock_guard<std::mutex> lk(m);
      data = d; len = n; busy = true;
      cv.notify_all();
    }
    bool finish() {  // everything handed in is written; false if a write failed
      if (!th.joinable()) return true;
      wait_idle();
      { std::lock_guard<std::mutex> lk(m); quit = true; }
      cv.notify_all();
      th.join();
      return !failed;
    }
  } writer;
  writer.start(out_fh);
  uint64_t n_emitted = 0;

  // one matrix of the output: the reference's progress lines in its order (:218-241, :281), the tail of gen_dist
  // and the print block
  auto emit = [&](uint64_t rep, const double *rs, const uint64_t *rc_, const uint64_t *bm, uint64_t n_blocks,
                  const double *ready = nullptr /* the matrix's distances, finished already (beside the matrix before it) */,
                  int ready_rc = 0) {
    if (p.verbose >= 1) {
      if (rep == 0) fprintf(stderr, "==> Analyzing full dataset...\n");
      else fprintf(stderr, "==> Bootstrap replicate # %lu ...\n", rep);
    }
    if (p.verbose >= 2) fprintf(stderr, "> Mapping positions...\n");
    if (rep > 0 && bm && p.verbose >= 5) {
      for (uint64_t b = 0; b < n_blocks; b++)
        for (uint64_t s = 0; s < p.boot_block_size; s++)
          fprintf(stderr, "block: %lu\torig_site: %lu\trand_block:%lu\trand_site: %lu\n", b,
                  b * p.boot_block_size + s, bm[b], bm[b] * p.boot_block_size + s);
    }
    if (p.verbose >= 2) fprintf(stderr, "> Calculating pairwise genetic distances...\n");
    if (p.verbose >= 3) {  // the per-pair line of ngsDist.cpp:366-367
      uint64_t k = 0;
      for (uint64_t i1 = 0; i1 < p.n_ind; i1++)
        for (uint64_t i2 = i1 + 1; i2 < p.n_ind; i2++, k++)
          fprintf(stderr, "\tDistance of %f from %lu valid sites (%f) between %s (ind %lu) and %s (ind %lu)!\n",
                  rs[k], rc_[k], rs[k] / (double)rc_[k], labels[i1].c_str(), i1, labels[i2].c_str(), i2);
    }
    const double *cells = ready;
    if (ready) {
      if (ready_rc) die("gen_dist", "invalid evolutionary model specified!");
    } else {
      const auto t_f0 = std::chrono::steady_clock::now();
      int rc = ngd_finish(rs, rc_, n_comb, p.tot_sites, p.evol_model, dist.data());
      if (rc) die("gen_dist", "invalid evolutionary model specified!");
      t_compute += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_f0).count();
      cells = dist.data();
    }

    if (p.verbose >= 2) fprintf(stderr, "> Printing distance matrix\n");
    // ngsDist.cpp:282-287 (join(), gen_func.cpp:479-496); rows formatted in parallel, same bytes
    const auto t_w0 = std::chrono::steady_clock::now();
    std::vector<char> &text = text_buf[n_emitted++ & 1];  // (the writer may still be on the other one)
    if (text.empty()) text.resize(64 + p.n_ind * (p.n_ind * 16 + 64));
    int64_t need = ngd_format_matrix(cells, p.n_ind, label_ptr.data(), text.data(), text.size(), p.n_threads);
    if (need > (int64_t)text.size()) {
      text.resize((size_t)need);
      need = ngd_format_matrix(cells, p.n_ind, label_ptr.data(), text.data(), text.size(), p.n_threads);
    }
    if (need < 0) die(__FUNCTION__, "cannot format the distance matrix");
    writer.wait_idle();  // the previous matrix is on its way out: matrices are written in order, one at a time
    if (writer.failed) die(__FUNCTION__, "cannot write output file!");
    writer.submit(text.data(), (size_t)need);
    t_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_w0).count();
  };

  // Matrices per engine call.  --indep_geno: 32 (the engine forms 32 replicates per pass over the per-block partials).
  // EM path: the per-site EM does not depend on the replicate and ONE pass of it serves every matrix of a call (the
  // engine spills the per-(pair, site) terms and contracts them with all the weight vectors: contract_mfma.hip), so the
  // whole job goes in one call where its results (16 B per cell) and block maps fit a few GB of host memory.
  uint64_t kBatch = 32;
  if (!p.indep_geno && p.n_boot_rep + 1 > kBatch) {
    const uint64_t by_results = (4ull << 30) / std::max<uint64_t>(1, n_comb * 16);
    const uint64_t blocks = p.boot_block_size ? p.n_sites / p.boot_block_size : 0;
    const uint64_t by_maps = blocks ? (2ull << 30) / (blocks * 8) : ~0ull;
    kBatch = std::max<uint64_t>(kBatch, std::min({by_results, by_maps, (uint64_t)p.n_boot_rep + 1}));
  }
  fflush(stdout);
  if (in_parts) {
    // ---- the site axis in ranges: side by side on the devices, one after the other where they are too small ----
    const uint64_t B = p.boot_block_size;
    const uint64_t n_eff = p.n_boot_rep ? p.n_sites - p.n_sites % B : 0, n_blocks = p.n_boot_rep ? n_eff / B : 0;
    uint64_t unit = 16;  // ranges are whole 16-site groups and whole bootstrap blocks
    if (p.n_boot_rep) { uint64_t a = 16, b = B; while (b) { uint64_t t = a % b; a = b; b = t; } unit = 16 / a * B; }
    // lcm(16, B) can exceed the data set (a large odd block size): then no split into whole units exists, one range
    // is the whole data set, and it only has to fit one device
    const uint64_t whole = (p.n_sites + 15) / 16 * 16;
    if (unit > whole) unit = whole;
    if (budget <= fixed || (budget - fixed) / per_site < unit)
      die(__FUNCTION__, "not even one range of sites (16 sites / one bootstrap block) fits the device");
    const uint64_t cap = (budget - fixed) / per_site / unit * unit;  // sites one device holds
    const uint64_t n_units = (p.n_sites + unit - 1) / unit;
    const uint64_t G = (uint64_t)p.n_gpus;
    const uint64_t part = std::min(cap, (n_units + G - 1) / G * unit);
    const uint64_t n_parts = (p.n_sites + part - 1) / part;
    if (p.verbose >= 1) {
      if (part < (n_units + G - 1) / G * unit)
        fprintf(stderr, "==> Data set larger than the device budget (%.1f GB): %lu ranges of up to %lu sites\n", budget / 1e9,
                n_parts, part);
      if (G > 1) fprintf(stderr, "==> Site axis split over %lu devices: %lu ranges of up to %lu sites\n", G, n_parts, part);
    }
    // every replicate's block multiplicities, drawn in the reference's order before any data is read
    std::vector<uint32_t> mult((uint64_t)p.n_boot_rep * n_blocks, 0);
    std::vector<uint64_t> maps_kept;
    {
      std::vector<uint64_t> bm(n_blocks);
      for (uint64_t r = 0; r < p.n_boot_rep; r++) {
        ngd_boot_block_map(rng, n_blocks, bm.data());
        for (uint64_t b = 0; b < n_blocks; b++) mult[r * n_blocks + bm[b]]++;
        if (p.verbose >= 5) maps_kept.insert(maps_kept.end(), bm.begin(), bm.end());
      }
    }
    const uint64_t n_mat = p.n_boot_rep + 1;
    // per device: the sums of its ranges, added in ascending range order; devices are added in device order at the
    // end -- a fixed order of additions for given --n_gpus and budget
    std::vector<std::vector<double>> dev_sum(G);
    std::vector<std::vector<uint64_t>> dev_cnt(G);
    std::vector<double> dev_secs(G, 0.0);
    // all matrices of one range: the full data set (ngd_run), then the replicates from their block multiplicities
    auto compute_range = [&](Engine &eng, uint64_t c0, uint64_t c1, uint64_t d) {
      const auto t_c0 = std::chrono::steady_clock::now();
      std::vector<double> &ts = dev_sum[d];
      std::vector<uint64_t> &tc = dev_cnt[d];
      if (ts.empty()) { ts.assign(n_mat * n_comb, 0.0); tc.assign(n_mat * n_comb, 0); }
      std::vector<double> ps(std::min<uint64_t>(kBatch, n_mat) * n_comb);
      std::vector<uint64_t> pc(ps.size());
      int rc = ngd_run(eng.h, nullptr, 0, 0, ps.data(), pc.data());
      if (rc) die_engine("ngd_run", rc);
      report_fixup(eng.h, p.verbose);
      for (uint64_t k = 0; k < n_comb; k++) { ts[k] += ps[k]; tc[k] += pc[k]; }
      const uint64_t blk_lo = std::min(c0, n_eff) / (B ? B : 1), blk_hi = std::min(c1, n_eff) / (B ? B : 1);
      const uint64_t nb = blk_hi - blk_lo;  // this range's blocks (none in a range of tail sites only)
      std::vector<uint32_t> mpart;
      for (uint64_t r0 = 0; nb && r0 < p.n_boot_rep; r0 += kBatch) {
        const uint64_t nr = std::min<uint64_t>(kBatch, p.n_boot_rep - r0);
        mpart.resize(nr * nb);
        for (uint64_t r = 0; r < nr; r++)
          memcpy(&mpart[r * nb], &mult[(r0 + r) * n_blocks + blk_lo], nb * sizeof(uint32_t));
        rc = ngd_run_mult_batch(eng.h, mpart.data(), (uint32_t)nr, nb, B, ps.data(), pc.data());
        if (rc) die_engine("ngd_run_mult_batch", rc);
        report_fixup(eng.h, p.verbose);
        for (uint64_t k = 0; k < nr * n_comb; k++) {
          ts[(1 + r0) * n_comb + k] += ps[k];
          tc[(1 + r0) * n_comb + k] += pc[k];
        }
      }
      dev_secs[d] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_c0).count();
    };
    if (p.verbose >= 1) fprintf(stderr, "==> Reading genotype data\n");
    const auto t_l0 = std::chrono::steady_clock::now();
    Loader L(p);
    if (L.seekable() && G > 1) {
      // plain binary file: every device's thread reads its own ranges (nothing is read twice, nothing is replicated)
      // the size check of the whole file, once, with the reference's two messages (read_data.cpp:45-47, :106-109)
      if (L.raw_size < p.n_sites * p.n_ind * 24)
        die("read_geno", "GENO file at premature EOF. Check GENO file and number of sites!");
      L.finish(p.n_sites * p.n_ind * 24 != L.raw_size);
      std::vector<std::thread> th;
      for (uint64_t d = 0; d < G; d++)
        th.emplace_back([&, d]() {
          for (uint64_t k = d; k < n_parts; k += G) {
            const uint64_t c0 = k * part, c1 = std::min(p.n_sites, c0 + part);
            Loader Ld(p, c0);
            Engine eng;
            make_engine(eng, c1 - c0, (int)d);
            Ld.load(eng, c1 - c0, c1 == p.n_sites);
            Ld.finish(false);
            compute_range(eng, c0, c1, d);
          }
        });
      for (auto &t : th) t.join();
    } else {
      // a stream (gz text, gz binary, stdin) is read front to back by this thread; a range is computed by its device's
      // thread while the next range is read into the next device
      std::vector<std::thread> busy(G);
      for (uint64_t k = 0; k < n_parts; k++) {
        const uint64_t d = k % G, c0 = k * part, c1 = std::min(p.n_sites, c0 + part);
        if (busy[d].joinable()) busy[d].join();  // the device's previous range has left it
        Engine *eng = new Engine();
        make_engine(*eng, c1 - c0, (int)d);
        L.load(*eng, c1 - c0, c1 == p.n_sites);
        busy[d] = std::thread([&, eng, c0, c1, d]() {
          compute_range(*eng, c0, c1, d);
          delete eng;
        });
      }
      for (auto &t : busy) if (t.joinable()) t.join();
      L.finish();
    }
    const double t_all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_l0).count();
    double t_dev = 0;
    for (double v : dev_secs) t_dev = std::max(t_dev, v);
    t_compute += t_dev;
    if (p.verbose >= 2)
      fprintf(stderr, "> read + prepare + upload + distances of %lu ranges on %lu device(s): %.3f s (%.2f GB of prepared input "
              "in all; slowest device's kernels and copies %.3f s)\n", n_parts, G, t_all, (double)p.n_ind * p.n_sites * 24 / 1e9,
              t_dev);
    std::vector<double> &tot_sum = dev_sum[0];
    std::vector<uint64_t> &tot_cnt = dev_cnt[0];
    if (tot_sum.empty()) { tot_sum.assign(n_mat * n_comb, 0.0); tot_cnt.assign(n_mat * n_comb, 0); }
    // devices in ascending order for every cell, cells on --n_threads threads
    parallel_for(p.n_threads, n_mat * n_comb, 1u << 16, [&](uint64_t lo, uint64_t hi) {
      for (uint64_t d = 1; d < G; d++)
        if (!dev_sum[d].empty())
          for (uint64_t k = lo; k < hi; k++) { tot_sum[k] += dev_sum[d][k]; tot_cnt[k] += dev_cnt[d][k]; }
    });
    for (uint64_t rep = 0; rep < n_mat; rep++)
      emit(rep, &tot_sum[rep * n_comb], &tot_cnt[rep * n_comb],
           rep && !maps_kept.empty() ? &maps_kept[(rep - 1) * n_blocks] : nullptr, n_blocks);
  } else {
  Engine eng;
  make_engine(eng, p.n_sites, 0);
  g_phases.mark("create");
  if (p.verbose >= 1) fprintf(stderr, "==> Reading genotype data\n");
  const auto t_load0 = std::chrono::steady_clock::now();
  {
    Loader L(p);
    L.load(eng, p.n_sites, true);
    L.finish();
  }
  const double t_load = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_load0).count();
  g_phases.mark("load");
  if (p.verbose >= 2)
    fprintf(stderr, "> read + prepare + upload: %.3f s (%.2f GB of prepared input resident on the device)\n", t_load,
            (double)p.n_ind * p.n_sites * 24 / 1e9);

  // Matrices go to the engine in batches: the first batch is the full-data matrix plus the first replicates
  // (ngd_run_job, which lets the engine share work between them), later ones replicates only (ngd_run_batch).
  // The block maps of a batch are drawn up front, in the order rnd_map_data (ngsDist.cpp:416-437) would draw
  // them -- nothing else consumes the generator.
  std::vector<double> sum;
  std::vector<uint64_t> cnt, block_maps;
  uint64_t n_sites = p.n_sites;

  // A batch's matrices stay in the engine and come to the host one at a time, as they are printed (ngd_fetch_matrix):
  // the host holds two n_pairs-long buffers, not a batch of them ([measured] 1000 individuals, 101 matrices: the 0.8 GB
  // of zero-filled result vectors and their page faults cost 1.5 s, three times the engine's whole job).
  bool in_engine = false;  // this batch's matrices are fetched from the engine
  auto run_all = [&](const uint64_t *maps, uint32_t n_rep, bool with_full, uint64_t n_blocks) {
    in_engine = false;
    if (n_rep && n_blocks == 0) {
      // fewer sites than one bootstrap block: the reference truncates the replicates to 0 sites (ngsDist.cpp:236),
      // visits none and prints 0/0; the full data set is still a plain run
      const uint64_t n_mat = n_rep + (with_full ? 1 : 0);
      sum.assign(n_mat * n_comb, 0.0);
      cnt.assign(n_mat * n_comb, 0);
      if (with_full) {
        int rc = ngd_run(eng.h, nullptr, 0, 0, sum.data(), cnt.data());
        if (rc) die_engine("ngd_run", rc);
        report_fixup(eng.h, p.verbose);
      }
      return;
    }
    sum.resize(n_comb);
    cnt.resize(n_comb);
    if (!n_rep) {  // the full data alone
      int rc = ngd_run(eng.h, nullptr, 0, 0, sum.data(), cnt.data());
      if (rc) die_engine("ngd_run", rc);
      report_fixup(eng.h, p.verbose);
      return;
    }
    int rc = with_full ? ngd_run_job(eng.h, maps, n_rep, n_blocks, p.boot_block_size, nullptr, nullptr)
                       : ngd_run_batch(eng.h, maps, n_rep, n_blocks, p.boot_block_size, nullptr, nullptr);
    if (rc) die_engine(with_full ? "ngd_run_job" : "ngd_run_batch", rc);
    report_fixup(eng.h, p.verbose);
    in_engine = true;
  };

  for (uint64_t rep = 0; rep <= p.n_boot_rep;) {
    const auto t_c0 = std::chrono::steady_clock::now();
    const bool with_full = rep == 0;
    const uint64_t n_in_batch = std::min<uint64_t>(kBatch, p.n_boot_rep - rep + 1);  // matrices of this batch
    const uint64_t n_boot_here = n_in_batch - (with_full ? 1 : 0);
    uint64_t n_blocks = 0;
    if (n_boot_here) {  // ngsDist.cpp:235-238
      n_sites -= n_sites % p.boot_block_size;
      n_blocks = n_sites / p.boot_block_size;
      block_maps.resize(n_boot_here * n_blocks);
      for (uint64_t r = 0; r < n_boot_here; r++) ngd_boot_block_map(rng, n_blocks, &block_maps[r * n_blocks]);
    }
    run_all(block_maps.data(), (uint32_t)n_boot_here, with_full, n_blocks);
    t_compute += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_c0).count();
    if (in_engine) {
      // a batch's matrices come out of the engine one at a time -- and matrix r + 1 is fetched and finished (ngd_fetch_matrix,
      // ngd_finish) by a thread of its own while matrix r is formatted and handed to the writer: two sets of buffers
      struct Pre {
        std::vector<double> sum, dist;
        std::vector<uint64_t> cnt;
        int rc_fetch = 0, rc_finish = 0;
        std::string err;
        std::thread th;
      } pre[2];
      auto start = [&](uint64_t r) {
        Pre &q = pre[r & 1];
        q.sum.resize(n_comb); q.dist.resize(n_comb); q.cnt.resize(n_comb);
        q.th = std::thread([&q, r, &eng, &p, n_comb]() {
          q.rc_fetch = ngd_fetch_matrix(eng.h, (uint32_t)r, q.sum.data(), q.cnt.data());
          if (q.rc_fetch) { q.err = ngd_last_error(); return; }  // (the message is the fetching thread's)
          q.rc_finish = ngd_finish(q.sum.data(), q.cnt.data(), n_comb, p.tot_sites, p.evol_model, q.dist.data());
        });
      };
      start(0);
      for (uint64_t r = 0; r < n_in_batch; r++, rep++) {
        Pre &q = pre[r & 1];
        const auto t_g0 = std::chrono::steady_clock::now();
        q.th.join();
        t_compute += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_g0).count();
        if (q.rc_fetch) die("ngd_fetch_matrix", (std::string("engine error ") + std::to_string(q.rc_fetch) + ": " + q.err).c_str());
        if (r + 1 < n_in_batch) start(r + 1);
        emit(rep, q.sum.data(), q.cnt.data(), rep > 0 ? &block_maps[(r - (with_full ? 1 : 0)) * n_blocks] : nullptr, n_blocks,
             q.dist.data(), q.rc_finish);
      }
    } else {
      for (uint64_t r = 0; r < n_in_batch; r++, rep++)
        emit(rep, &sum[r * n_comb], &cnt[r * n_comb], rep > 0 ? &block_maps[(r - (with_full ? 1 : 0)) * n_blocks] : nullptr, n_blocks);
    }
  }
  g_phases.mark("matrices");
  eng.leave_to_exit = true;
  }
  g_phases.mark("destroy");
  if (!writer.finish()) die(__FUNCTION__, "cannot write output file!");
  g_phases.mark("write_tail");
  fclose(out_fh);
  if (p.verbose >= 2)
    fprintf(stderr, "> distances: %.3f s for %lu matri%s of %lu pairs; formatting + writing them: %.3f s\n", t_compute,
            p.n_boot_rep + 1, p.n_boot_rep ? "ces" : "x", n_comb, t_write);
  if (p.verbose >= 2) {
    g_phases.add("of_matrices_distances", t_compute);
    g_phases.add("of_matrices_format_write", t_write);
    g_phases.print();
  }
  if (p.verbose >= 1) fprintf(stderr, "==> Freeing memory...\n");
  if (p.verbose >= 1) fprintf(stderr, "Done!\n");
  return 0;
}
