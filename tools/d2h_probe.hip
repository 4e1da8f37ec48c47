// tools/d2h_probe.hip -- which engine carries a device-to-host copy into pinned memory: run under
//   rocprofv3 --kernel-trace --memory-copy-trace --stats -- tools/d2h_probe
// and look for __amd_rocclr_copyBuffer (a blit KERNEL on the compute units) against MEMORY_COPY_DEVICE_TO_HOST rows (the DMA engines).
// Prints the rate of 8-MiB and 64-MiB copies, alone and while a memory-bound kernel runs beside them.
// hipcc --offload-arch=gfx950 -O2 -o tools/d2h_probe tools/d2h_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_stream(const double *a, double *b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i] * 1.0000001;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t big = 64u << 20, n_busy = (size_t)1 << 28;  // 2 GiB in, 2 GiB out per busy launch
  double *d, *h, *ba, *bb;
  CK(hipMalloc(&d, big)); CK(hipHostMalloc(&h, big, hipHostMallocDefault));
  CK(hipMalloc(&ba, n_busy * 8)); CK(hipMalloc(&bb, n_busy * 8));
  CK(hipMemset(d, 1, big)); CK(hipMemset(ba, 0, n_busy * 8));
  hipStream_t s1, s2, sk;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
  for (int busy = 0; busy < 2; busy++)
    for (size_t piece : {(size_t)8 << 20, (size_t)64 << 20})
      for (int two = 0; two < 2; two++) {
        double best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
          CK(hipDeviceSynchronize());
          if (busy) for (int q = 0; q < 3; q++) hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, sk, ba, bb, n_busy);
          const double t0 = now();
          int k = 0;
          for (size_t off = 0; off < big; off += piece, k++)
            CK(hipMemcpyAsync((char *)h + off, (char *)d + off, piece, hipMemcpyDeviceToHost, (two && (k & 1)) ? s2 : s1));
          CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
          best = std::min(best, now() - t0);
        }
        printf("{\"d2h_piece_mib\": %zu, \"streams\": %d, \"beside_a_memory_bound_kernel\": %d, \"GBps\": %.2f}\n", piece >> 20, two + 1, busy,
               big / best / 1e9);
      }
  return 0;
}
