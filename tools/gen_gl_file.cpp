// tools/gen_gl_file.cpp -- writes a binary genotype-likelihood file of the layout the reference reads
// (read_data.cpp:28-31: little-endian doubles [site][individual][3]) from the counter-based recipe of SURVEY 8(d), the
// one the engine's k_synth and the oracle use: element (s, i, g) -> splitmix64 finaliser -> u in (0,1) -> u^3, the three
// values of an (individual, site) scaled to sum 1.  So a run of the C++ host on the file can be checked against an
// engine filled on the device with the same seed.  Threads write disjoint site ranges with pwrite.
//   usage: gen_gl_file <path> <n_ind> <n_sites> <seed> [n_threads]
// build: g++ -O2 -ffp-contract=off -pthread -o tools/gen_gl_file tools/gen_gl_file.cpp
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s <path> <n_ind> <n_sites> <seed> [n_threads]\n", argv[0]);
    return 2;
  }
  const char *path = argv[1];
  const uint64_t n_ind = strtoull(argv[2], nullptr, 10), n_sites = strtoull(argv[3], nullptr, 10);
  const uint64_t seed = strtoull(argv[4], nullptr, 10);
  const unsigned n_thr = argc > 5 ? (unsigned)atoi(argv[5]) : std::max(1u, std::thread::hardware_concurrency());
  int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
  if (fd < 0) { perror(path); return 1; }
  const uint64_t base = seed * 0x9E3779B97F4A7C15ull;
  const uint64_t chunk = std::max<uint64_t>(1, (8ull << 20) / (n_ind * 24));  // sites per write
  std::vector<std::thread> th;
  bool bad = false;
  for (unsigned t = 0; t < n_thr; t++)
    th.emplace_back([&, t]() {
      std::vector<double> buf(chunk * n_ind * 3);
      for (uint64_t c = t; c * chunk < n_sites; c += n_thr) {
        const uint64_t s0 = c * chunk, n = std::min(chunk, n_sites - s0);
        for (uint64_t e = 0; e < n * n_ind; e++) {
          const uint64_t ge = s0 * n_ind + e;
          double x[3];
          for (int g = 0; g < 3; g++) {
            const uint64_t z = mix64(base + (ge * 3 + (uint64_t)g));
            const double u = ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
            x[g] = (u * u) * u;
          }
          const double tot = (x[0] + x[1]) + x[2];
          buf[3 * e] = x[0] / tot; buf[3 * e + 1] = x[1] / tot; buf[3 * e + 2] = x[2] / tot;
        }
        const char *src = (const char *)buf.data();
        uint64_t left = n * n_ind * 24, off = s0 * n_ind * 24;
        while (left) {
          const ssize_t w = pwrite(fd, src, left, (off_t)off);
          if (w <= 0) { bad = true; return; }
          src += w; off += (uint64_t)w; left -= (uint64_t)w;
        }
      }
    });
  for (auto &x : th) x.join();
  close(fd);
  if (bad) { fprintf(stderr, "write failed\n"); return 1; }
  return 0;
}
