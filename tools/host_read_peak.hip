// tools/host_read_peak.hip -- how fast can a binary GL file that sits in host memory (page cache / tmpfs) reach the
// device?  The loader's candidates, side by side on one file:
//   pread_T       T threads pread() the file into a pinned buffer (what Loader::load does), no device copy: the host side alone
//   mmapcpy_T     T threads memcpy() out of a MAP_SHARED mapping into the pinned buffer
//   memcpy_map    hipMemcpy straight out of the mapping (the runtime locks the pages itself -- if the driver takes
//                 file-backed pages at all)
//   register      hipHostRegister of the mapping + hipMemcpyAsync out of it + unregister
//   pipeline_T    pread on T threads into 4 pinned buffers of 128 MiB, hipMemcpyAsync of each as it fills (two streams)
// usage: host_read_peak <file> [max GiB to use = 4]        prints one JSON line
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F>
static void on_threads(unsigned T, size_t bytes, size_t grain, F fn) {
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; t++)
    th.emplace_back([&]() {
      for (;;) {
        const size_t lo = next.fetch_add(grain);
        if (lo >= bytes) return;
        fn(lo, std::min(bytes, lo + grain));
      }
    });
  for (auto &x : th) x.join();
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const int fd = open(argv[1], O_RDONLY);
  if (fd < 0) { perror(argv[1]); return 1; }
  struct stat st;
  fstat(fd, &st);
  const size_t cap = (size_t)(argc > 2 ? atoi(argv[2]) : 4) << 30;
  const size_t n = std::min<size_t>((size_t)st.st_size, cap) / (128u << 20) * (128u << 20);
  if (!n) { fprintf(stderr, "file shorter than 128 MiB\n"); return 1; }
  CK(hipSetDevice(0));
  char *pin = nullptr, *d = nullptr;
  CK(hipHostMalloc((void **)&pin, n, hipHostMallocDefault));
  CK(hipMalloc((void **)&d, n));
  printf("{\"file\": \"%s\", \"bytes\": %zu", argv[1], n);
  auto pread_all = [&](char *dst, size_t lo, size_t hi) {
    while (lo < hi) {
      const ssize_t r = pread(fd, dst + lo, hi - lo, (off_t)lo);
      if (r <= 0) { perror("pread"); exit(1); }
      lo += (size_t)r;
    }
  };
  for (int pass = 0; pass < 2; pass++)
    for (unsigned T : {4u, 8u, 16u, 32u}) {
      const double a = now_s();
      on_threads(T, n, 8u << 20, [&](size_t lo, size_t hi) { pread_all(pin, lo, hi); });
      if (pass) printf(", \"pread_%u_GBps\": %.2f", T, n / 1e9 / (now_s() - a));
    }
  char *map = (char *)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
  if (map == MAP_FAILED) { perror("mmap"); return 1; }
  for (unsigned T : {8u, 16u}) {
    const double a = now_s();
    on_threads(T, n, 8u << 20, [&](size_t lo, size_t hi) { memcpy(pin + lo, map + lo, hi - lo); });
    printf(", \"mmapcpy_%u_GBps\": %.2f", T, n / 1e9 / (now_s() - a));
  }
  {  // straight out of the mapping
    for (int r = 0; r < 2; r++) {
      const double a = now_s();
      const hipError_t e = hipMemcpy(d, map, n, hipMemcpyHostToDevice);
      if (e != hipSuccess) { printf(", \"memcpy_map_error\": \"%s\"", hipGetErrorString(e)); (void)hipGetLastError(); break; }
      printf(", \"memcpy_map_%d_GBps\": %.2f", r, n / 1e9 / (now_s() - a));
    }
  }
  {
    double a = now_s();
    const hipError_t e = hipHostRegister(map, n, hipHostRegisterDefault);
    if (e != hipSuccess) {
      printf(", \"register_error\": \"%s\"", hipGetErrorString(e));
      (void)hipGetLastError();
    } else {
      printf(", \"register_GBps\": %.2f", n / 1e9 / (now_s() - a));
      a = now_s();
      CK(hipMemcpy(d, map, n, hipMemcpyHostToDevice));
      printf(", \"registered_copy_GBps\": %.2f", n / 1e9 / (now_s() - a));
      a = now_s();
      CK(hipHostUnregister(map));
      printf(", \"unregister_GBps\": %.2f", n / 1e9 / (now_s() - a));
    }
  }
  {  // a fresh mapping registered piece by piece on several threads (does the pinning scale with threads?)
    char *map2 = (char *)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
    std::atomic<int> bad{0};
    const double a = now_s();
    on_threads(8, n, 128u << 20, [&](size_t lo, size_t hi) {
      if (hipHostRegister(map2 + lo, hi - lo, hipHostRegisterDefault) != hipSuccess) bad = 1;
    });
    if (bad) {
      printf(", \"register_8threads_error\": 1");
      (void)hipGetLastError();
    } else {
      printf(", \"register_8threads_GBps\": %.2f", n / 1e9 / (now_s() - a));
      for (size_t lo = 0; lo < n; lo += 128u << 20) (void)hipHostUnregister(map2 + lo);
    }
    munmap(map2, n);
  }
  // the pipeline: T reader threads fill 128 MiB pinned pieces, each copied as it fills
  hipStream_t st2[2];
  for (int i = 0; i < 2; i++) CK(hipStreamCreateWithFlags(&st2[i], hipStreamNonBlocking));
  for (unsigned T : {8u, 16u, 24u}) {
    const size_t piece = 128u << 20;
    const int NB = 4;
    hipEvent_t freed[NB];
    for (int b = 0; b < NB; b++) CK(hipEventCreateWithFlags(&freed[b], hipEventDisableTiming));
    const double a = now_s();
    int k = 0;
    for (size_t off = 0; off < n; off += piece, k++) {
      const int b = k % NB;
      CK(hipEventSynchronize(freed[b]));
      char *dst = pin + (size_t)b * piece;
      on_threads(T, piece, 4u << 20, [&](size_t lo, size_t hi) {
        size_t p = lo;
        while (p < hi) {
          const ssize_t r = pread(fd, dst + p, hi - p, (off_t)(off + p));
          if (r <= 0) exit(1);
          p += (size_t)r;
        }
      });
      CK(hipMemcpyAsync(d + off, dst, piece, hipMemcpyHostToDevice, st2[k & 1]));
      CK(hipEventRecord(freed[b], st2[k & 1]));
    }
    CK(hipDeviceSynchronize());
    printf(", \"pipeline_%u_GBps\": %.2f", T, n / 1e9 / (now_s() - a));
    for (int b = 0; b < NB; b++) CK(hipEventDestroy(freed[b]));
  }
  printf("}\n");
  return 0;
}
