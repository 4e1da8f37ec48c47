// tools/alloc_cost.hip -- what does device memory cost to ALLOCATE on this box, and can that cost be taken in pieces
// while a load is running?  (round 6: `hipMalloc` of cfg 3's 24.6 GB image takes 0.3 ms in one run of the host and
// 0.98 s in the next -- rocprofv3 --hip-trace, gpurun_out/r6/hiptrace_cfg3_3.)
//   malloc_N        hipMalloc of `GiB` three times in a row (free in between), seconds each; + a 1 GiB one
//   vmm             the same bytes as 1-GiB physical chunks mapped into one reserved range (hipMemCreate / hipMemMap /
//                   hipMemSetAccess), seconds per chunk (min / mean / max) and in all
//   h2d_during      pinned H2D copy rate while another thread allocates (is the link's rate kept?)
//   read_GBps       a streaming read kernel over 8 GiB of each kind of memory (does a range mapped in pieces read as fast?)
// usage: alloc_cost [GiB = 32]    one JSON line
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
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

__global__ void k_read(const double2 *__restrict__ p, size_t n, double *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0;
  for (; i < n; i += stride) { const double2 v = p[i]; acc += v.x + v.y; }
  if (acc == 12345.678) *out = acc;
}

static double read_rate(const void *p, size_t bytes, double *d_out) {
  double best = 0;
  for (int r = 0; r < 4; r++) {
    CK(hipDeviceSynchronize());
    const double a = now_s();
    hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, (const double2 *)p, bytes / 16, d_out);
    CK(hipDeviceSynchronize());
    const double g = bytes / 1e9 / (now_s() - a);
    if (r && g > best) best = g;
  }
  return best;
}

int main(int argc, char **argv) {
  const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 32, big = gib << 30;
  CK(hipSetDevice(0));
  double *d_out;
  CK(hipMalloc((void **)&d_out, 8));
  printf("{\"GiB\": %zu", gib);
  void *p = nullptr;
  for (int r = 0; r < 3; r++) {
    double a = now_s();
    CK(hipMalloc(&p, big));
    const double t_alloc = now_s() - a;
    a = now_s();
    CK(hipMemset(p, 0, big));
    CK(hipDeviceSynchronize());
    const double t_set = now_s() - a;
    double rr = 0;
    if (r == 2) rr = read_rate(p, std::min<size_t>(big, (size_t)8 << 30), d_out);
    a = now_s();
    CK(hipFree(p));
    printf(", \"malloc_%d_s\": %.4f, \"memset_%d_s\": %.4f, \"free_%d_s\": %.4f", r, t_alloc, r, t_set, r, now_s() - a);
    if (r == 2) printf(", \"read_malloc_GBps\": %.1f", rr);
    fflush(stdout);
  }
  {
    double a = now_s();
    CK(hipMalloc(&p, (size_t)1 << 30));
    printf(", \"malloc_1GiB_s\": %.4f", now_s() - a);
    CK(hipFree(p));
  }
  // the same bytes in pieces
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t chunk = (size_t)1 << 30;
    void *va = nullptr;
    double a = now_s();
    CK(hipMemAddressReserve(&va, big, 0, nullptr, 0));
    const double t_res = now_s() - a;
    std::vector<hipMemGenericAllocationHandle_t> hs(big / chunk);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    double t_min = 1e9, t_max = 0, t_sum = 0, t_create = 0, t_map = 0, t_acc = 0;
    const double a_all = now_s();
    for (size_t c = 0; c < hs.size(); c++) {
      const double t0 = now_s();
      CK(hipMemCreate(&hs[c], chunk, &prop, 0));
      const double t1 = now_s();
      CK(hipMemMap((char *)va + c * chunk, chunk, 0, hs[c], 0));
      const double t2 = now_s();
      CK(hipMemSetAccess((char *)va + c * chunk, chunk, &acc, 1));
      const double t3 = now_s();
      t_create += t1 - t0; t_map += t2 - t1; t_acc += t3 - t2;
      const double t = t3 - t0;
      t_min = std::min(t_min, t); t_max = std::max(t_max, t); t_sum += t;
    }
    const double t_all = now_s() - a_all;
    CK(hipMemset(va, 0, big));
    CK(hipDeviceSynchronize());
    const double rr = read_rate(va, std::min<size_t>(big, (size_t)8 << 30), d_out);
    a = now_s();
    CK(hipMemUnmap(va, big));
    for (auto &h : hs) CK(hipMemRelease(h));
    CK(hipMemAddressFree(va, big));
    printf(", \"vmm_granularity\": %zu, \"vmm_reserve_s\": %.4f, \"vmm_all_s\": %.4f, \"vmm_chunk_s_min_mean_max\": [%.4f, %.4f, %.4f], "
           "\"vmm_create_s\": %.4f, \"vmm_map_s\": %.4f, \"vmm_setaccess_s\": %.4f, \"vmm_free_s\": %.4f, \"read_vmm_GBps\": %.1f",
           gran, t_res, t_all, t_min, t_sum / hs.size(), t_max, t_create, t_map, t_acc, now_s() - a, rr);
    fflush(stdout);
  }
  // the link while memory is being allocated
  {
    const size_t cb = (size_t)1 << 30;
    char *h = nullptr, *d = nullptr;
    CK(hipHostMalloc((void **)&h, cb, hipHostMallocDefault));
    CK(hipMalloc((void **)&d, cb));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto copy_rate = [&](int n) {
      const double a = now_s();
      for (int i = 0; i < n; i++) CK(hipMemcpyAsync(d, h, cb, hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
      return n * (double)cb / 1e9 / (now_s() - a);
    };
    copy_rate(2);
    printf(", \"h2d_alone_GBps\": %.2f", copy_rate(8));
    std::atomic<double> t_alloc{0};
    std::thread th([&]() {
      CK(hipSetDevice(0));
      void *q = nullptr;
      const double a = now_s();
      CK(hipMalloc(&q, big));
      t_alloc = now_s() - a;
      CK(hipFree(q));
    });
    const double r = copy_rate(16);
    th.join();
    printf(", \"h2d_during_malloc_GBps\": %.2f, \"malloc_during_h2d_s\": %.4f", r, t_alloc.load());
    CK(hipFree(d));
    CK(hipHostFree(h));
  }
  printf("}\n");
  return 0;
}
