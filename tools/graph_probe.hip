// tools/graph_probe.hip -- does a HIP graph of (event, kernel of ~0.3 ms, event, small kernel, event) shorten a step against the
// same calls issued directly, and can events recorded INSIDE a captured graph be read afterwards?  (round 6: no; yes.)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/graph_probe tools/graph_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_spin(double *p, int n) { double a = p[threadIdx.x]; for (int i = 0; i < n; i++) a = a * 1.0000001 + 1e-9; p[threadIdx.x] = a; }
__global__ void k_small(double *p) { p[threadIdx.x] += 1.0; }
int main() {
  CK(hipSetDevice(0));
  double *d; CK(hipMalloc(&d, 4096));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  // direct: two launches + events, one sync
  const int N = 300;
  for (int w = 0; w < 20; w++) { hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, d, 20000); CK(hipStreamSynchronize(st)); }
  double t0 = now_us();
  float acc = 0;
  for (int i = 0; i < N; i++) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, d, 20000);
    CK(hipEventRecord(e1, st));
    hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, d);
    CK(hipEventRecord(e2, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); acc += ms;
  }
  double t_direct = (now_us() - t0) / N;
  printf("direct: %.1f us per step, kernel by events %.1f us\n", t_direct, acc / N * 1e3);
  // graph by stream capture
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(e0, st));
  hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, d, 20000);
  CK(hipEventRecord(e1, st));
  hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, d);
  CK(hipEventRecord(e2, st));
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 5; w++) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
  t0 = now_us(); acc = 0; int ok = 0;
  for (int i = 0; i < N; i++) {
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    float ms; hipError_t er = hipEventElapsedTime(&ms, e0, e1);
    if (er == hipSuccess) { acc += ms; ok++; } else (void)hipGetLastError();
  }
  double t_graph = (now_us() - t0) / N;
  printf("graph: %.1f us per step, events inside readable %d/%d, kernel by events %.1f us\n", t_graph, ok, N, ok ? acc / ok * 1e3 : 0.0);
  // graph without inner events, outer events
  hipGraph_t g2; hipGraphExec_t ge2;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, d, 20000);
  hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, d);
  CK(hipStreamEndCapture(st, &g2));
  CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
  for (int w = 0; w < 5; w++) { CK(hipGraphLaunch(ge2, st)); CK(hipStreamSynchronize(st)); }
  t0 = now_us(); acc = 0;
  for (int i = 0; i < N; i++) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge2, st));
    CK(hipEventRecord(e3, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e3)); acc += ms;
  }
  printf("graph (no inner events): %.1f us per step, both kernels by outer events %.1f us\n", (now_us() - t0) / N, acc / N * 1e3);
  return 0;
}
