// tools/host_read_pipeline.cpp -- which way of feeding the copy engine keeps the host link busy?  A binary GL file in host
// memory (page cache / tmpfs) goes to the device in pieces; the candidates differ in how a piece gets from the file's pages
// to something the copy engine can read:
//   pread     worker threads pread() into a ring of pinned buffers                       (round 2..5's loader)
//   memcpy    worker threads memcpy() out of a MAP_SHARED mapping into the ring
//   ntcopy    the same with non-temporal stores (the copy engine then reads DRAM, not dirty lines of the cores' caches)
//   direct    worker threads only fault the mapping's pages in (MADV_POPULATE_READ), hipMemcpy reads the mapping itself
// All with persistent workers, pieces of --piece MiB, a ring of --ring buffers, copies alternating two streams.
// usage: host_read_pipeline <file> [GiB=8] [piece MiB=64] [ring=4]          one JSON line
// build: g++ -O2 -mavx2 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/host_read_pipeline.cpp -L/opt/rocm/lib -lamdhip64
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// persistent workers: run(fn, n, grain) hands [lo, hi) ranges to T threads and returns when all are done
struct Pool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  std::function<void(size_t, size_t)> fn;
  size_t n = 0, grain = 1;
  std::atomic<size_t> next{0};
  unsigned gen = 0, busy = 0;
  bool stop = false;
  explicit Pool(unsigned T) {
    for (unsigned t = 0; t < T; t++)
      th.emplace_back([this]() {
        unsigned seen = 0;
        for (;;) {
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
          }
          for (;;) {
            const size_t lo = next.fetch_add(grain);
            if (lo >= n) break;
            fn(lo, std::min(n, lo + grain));
          }
          std::lock_guard<std::mutex> lk(mu);
          if (--busy == 0) cv_done.notify_all();
        }
      });
  }
  void run(std::function<void(size_t, size_t)> f, size_t n_, size_t grain_) {
    std::unique_lock<std::mutex> lk(mu);
    fn = std::move(f); n = n_; grain = grain_; next = 0; busy = (unsigned)th.size(); gen++;
    cv.notify_all();
    cv_done.wait(lk, [&] { return busy == 0; });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    for (auto &t : th) t.join();
  }
};

static void nt_copy(char *dst, const char *src, size_t n) {  // dst 32-byte aligned, n a multiple of 128
  for (size_t i = 0; i < n; i += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
    _mm256_stream_si256((__m256i *)(dst + i), a);
    _mm256_stream_si256((__m256i *)(dst + i + 32), b);
    _mm256_stream_si256((__m256i *)(dst + i + 64), c);
    _mm256_stream_si256((__m256i *)(dst + i + 96), d);
  }
  _mm_sfence();
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const int fd = open(argv[1], O_RDONLY);
  if (fd < 0) { perror(argv[1]); return 1; }
  struct stat st;
  fstat(fd, &st);
  const size_t cap = (size_t)(argc > 2 ? atoi(argv[2]) : 8) << 30;
  const size_t piece = (size_t)(argc > 3 ? atoi(argv[3]) : 64) << 20;
  const int ring = argc > 4 ? atoi(argv[4]) : 4;
  const size_t n = std::min<size_t>((size_t)st.st_size, cap) / piece * piece;
  if (!n) return 1;
  CK(hipSetDevice(0));
  char *d = nullptr;
  CK(hipMalloc((void **)&d, n));
  std::vector<char *> pin(ring);
  std::vector<hipEvent_t> freed(ring);
  for (int b = 0; b < ring; b++) {
    CK(hipHostMalloc((void **)&pin[b], piece, hipHostMallocDefault));
    memset(pin[b], 0, piece);
    CK(hipEventCreateWithFlags(&freed[b], hipEventDisableTiming));
  }
  hipStream_t s2[2];
  for (int i = 0; i < 2; i++) CK(hipStreamCreateWithFlags(&s2[i], hipStreamNonBlocking));
  printf("{\"file\": \"%s\", \"bytes\": %zu, \"piece_MiB\": %zu, \"ring\": %d", argv[1], n, piece >> 20, ring);
  fflush(stdout);

  enum Mode { PREAD, MEMCPY, NTCOPY, DIRECT };
  const char *names[] = {"pread", "memcpy", "ntcopy", "direct"};
  for (int mode : {PREAD, MEMCPY, NTCOPY, DIRECT})
    for (unsigned T : {8u, 16u}) {
      // a fresh mapping per run: its pages are not yet in this process's page table, as in a real load
      char *map = mode == PREAD ? nullptr : (char *)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
      if (map == MAP_FAILED) { perror("mmap"); return 1; }
      Pool pool(T);
      double t_fill = 0, t_wait = 0;
      CK(hipDeviceSynchronize());
      const double a = now_s();
      int k = 0;
      bool failed = false;
      for (size_t off = 0; off < n && !failed; off += piece, k++) {
        const int b = k % ring;
        char *dst = pin[b];
        double t0 = now_s();
        if (mode != DIRECT) CK(hipEventSynchronize(freed[b]));
        double t1 = now_s();
        t_wait += t1 - t0;
        const size_t grain = std::max<size_t>(1u << 20, piece / (4 * T) / 4096 * 4096);
        switch (mode) {
          case PREAD:
            pool.run([&](size_t lo, size_t hi) {
              while (lo < hi) {
                const ssize_t r = pread(fd, dst + lo, hi - lo, (off_t)(off + lo));
                if (r <= 0) exit(1);
                lo += (size_t)r;
              }
            }, piece, grain);
            break;
          case MEMCPY: pool.run([&](size_t lo, size_t hi) { memcpy(dst + lo, map + off + lo, hi - lo); }, piece, grain); break;
          case NTCOPY: pool.run([&](size_t lo, size_t hi) { nt_copy(dst + lo, map + off + lo, hi - lo); }, piece, grain); break;
          case DIRECT:
            pool.run([&](size_t lo, size_t hi) {
              if (madvise(map + off + lo, hi - lo, MADV_POPULATE_READ) != 0) {  // (old kernels: touch the pages)
                volatile char sink = 0;
                for (size_t p = lo; p < hi; p += 4096) sink += map[off + p];
              }
            }, piece, grain);
            break;
        }
        t_fill += now_s() - t1;
        if (mode == DIRECT) {
          if (hipMemcpyAsync(d + off, map + off, piece, hipMemcpyHostToDevice, s2[k & 1]) != hipSuccess) {
            failed = true;
            (void)hipGetLastError();
          }
        } else {
          CK(hipMemcpyAsync(d + off, dst, piece, hipMemcpyHostToDevice, s2[k & 1]));
          CK(hipEventRecord(freed[b], s2[k & 1]));
        }
      }
      CK(hipDeviceSynchronize());
      const double dt = now_s() - a;
      if (failed) printf(", \"%s_%u_error\": 1", names[mode], T);
      else printf(", \"%s_%u_GBps\": %.2f, \"%s_%u_fill_s\": %.3f, \"%s_%u_wait_s\": %.3f", names[mode], T, n / 1e9 / dt, names[mode], T,
                  t_fill, names[mode], T, t_wait);
      fflush(stdout);
      if (map) munmap(map, n);
    }
  printf("}\n");
  return 0;
}
