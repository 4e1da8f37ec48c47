// tools/pcie_peak.hip -- the host link's roof for the load phase: pinned host -> device copy rates
// measured the ways a loader can drive them, so that "file bytes / time" of ngd_stage_submit has a denominator taken
// in the same lease.
//   copy engine: hipMemcpyAsync of 128 MiB and 1 GiB out of pinned memory, on 1, 2 and 4 streams at once
//   kernel:      a kernel reading the pinned buffer through the link (zero copy) and writing HBM
//   D2H:         hipMemcpyAsync device -> pinned, 64 MiB and 1 GiB
//   pageable:    hipMemcpy out of malloc'd memory (what a caller without pinned buffers gets)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/pcie_peak tools/pcie_peak.hip ; prints one JSON line.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
      exit(2);                                                                             \
    }                                                                                      \
  } while (0)

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// every thread moves 16 bytes per trip; enough workgroups to keep the link's request queue full
__global__ void k_pull(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}

int main(int argc, char **argv) {
  const size_t big = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024) << 20;  // MiB
  const int reps = 5;
  CK(hipSetDevice(0));
  char *h = nullptr, *d = nullptr;
  double t0 = now_s();
  CK(hipHostMalloc((void **)&h, big, hipHostMallocDefault));
  const double t_pin = now_s() - t0;
  memset(h, 1, big);
  CK(hipMalloc((void **)&d, big));
  hipStream_t st[4];
  for (int i = 0; i < 4; i++) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  printf("{\"pinned_alloc_gib_per_s\": %.2f", (double)big / (1ull << 30) / t_pin);

  auto rate = [&](size_t bytes, int n_streams, bool d2h) {
    double best = 0;
    for (int r = 0; r < reps + 1; r++) {
      const size_t part = bytes / n_streams / 256 * 256;
      CK(hipDeviceSynchronize());
      const double a = now_s();
      for (int s = 0; s < n_streams; s++) {
        if (d2h) CK(hipMemcpyAsync(h + s * part, d + s * part, part, hipMemcpyDeviceToHost, st[s]));
        else CK(hipMemcpyAsync(d + s * part, h + s * part, part, hipMemcpyHostToDevice, st[s]));
      }
      for (int s = 0; s < n_streams; s++) CK(hipStreamSynchronize(st[s]));
      const double gbs = (double)part * n_streams / 1e9 / (now_s() - a);
      if (r && gbs > best) best = gbs;  // (first pass: warm-up)
    }
    return best;
  };
  for (size_t sz : {(size_t)128 << 20, big})
    for (int ns : {1, 2, 4})
      printf(", \"h2d_%zuMiB_%dstream_GBps\": %.2f", sz >> 20, ns, rate(sz, ns, false));
  for (size_t sz : {(size_t)64 << 20, big}) printf(", \"d2h_%zuMiB_GBps\": %.2f", sz >> 20, rate(sz, 1, true));

  // chunked pipeline as a loader would drive it: 128 MiB pieces of the big buffer, alternating two streams
  {
    double best = 0;
    const size_t piece = (size_t)128 << 20;
    for (int r = 0; r < 3; r++) {
      CK(hipDeviceSynchronize());
      const double a = now_s();
      int k = 0;
      for (size_t off = 0; off + piece <= big; off += piece, k++)
        CK(hipMemcpyAsync(d + off, h + off, piece, hipMemcpyHostToDevice, st[k & 1]));
      CK(hipDeviceSynchronize());
      const double gbs = (double)(big / piece * piece) / 1e9 / (now_s() - a);
      if (gbs > best) best = gbs;
    }
    printf(", \"h2d_128MiB_pieces_2streams_GBps\": %.2f", best);
  }

  // kernel pull (zero copy): grid sizes around the chip's 256 CUs
  for (int blocks : {256, 1024, 4096}) {
    double best = 0;
    for (int r = 0; r < reps + 1; r++) {
      CK(hipDeviceSynchronize());
      const double a = now_s();
      hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, st[0], (const double2 *)h, (double2 *)d, big / 16);
      CK(hipStreamSynchronize(st[0]));
      const double gbs = (double)big / 1e9 / (now_s() - a);
      if (r && gbs > best) best = gbs;
    }
    printf(", \"kernel_pull_%dwg_GBps\": %.2f", blocks, best);
  }
  // copy engine and a pulling kernel at once, half of the buffer each
  {
    double best = 0;
    for (int r = 0; r < reps + 1; r++) {
      CK(hipDeviceSynchronize());
      const double a = now_s();
      CK(hipMemcpyAsync(d, h, big / 2, hipMemcpyHostToDevice, st[1]));
      hipLaunchKernelGGL(k_pull, dim3(1024), dim3(256), 0, st[0], (const double2 *)(h + big / 2), (double2 *)(d + big / 2),
                         big / 32);
      CK(hipDeviceSynchronize());
      const double gbs = (double)big / 1e9 / (now_s() - a);
      if (r && gbs > best) best = gbs;
    }
    printf(", \"copy_engine_plus_kernel_pull_GBps\": %.2f", best);
  }
  {  // pageable memory
    char *m = (char *)malloc(big);
    memset(m, 2, big);
    double best = 0;
    for (int r = 0; r < 3; r++) {
      const double a = now_s();
      CK(hipMemcpy(d, m, big, hipMemcpyHostToDevice));
      const double gbs = (double)big / 1e9 / (now_s() - a);
      if (r && gbs > best) best = gbs;
    }
    printf(", \"h2d_pageable_GBps\": %.2f", best);
    free(m);
  }
  {  // host side: one thread's memcpy into the pinned buffer (what pread costs at least)
    char *m = (char *)malloc(big);
    memset(m, 3, big);
    const double a = now_s();
    memcpy(h, m, big);
    printf(", \"host_memcpy_1thread_GBps\": %.2f", (double)big / 1e9 / (now_s() - a));
    free(m);
  }
  printf(", \"bytes\": %zu}\n", big);
  CK(hipFree(d));
  CK(hipHostFree(h));
  return 0;
}
