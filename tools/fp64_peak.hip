// fp64_peak.hip -- what the FP64 pipes of one MI355X actually sustain, with the
// shader clock read in-kernel (s_memtime cycles / s_memrealtime 100 MHz ticks).
//   mfma<NA,NB>: v_mfma_f64_16x16x4_f64 over an NA x NB register tile (NA*NB accumulators,
//                NA+NB distinct operand registers, as in accum_mfma.hip)
//   fma        : v_fma_f64, 32 independent chains per lane
// build: hipcc --offload-arch=gfx950 -O3 tools/fp64_peak.hip -o tools/fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

struct stamp { unsigned long long cyc, rt; };

template <int NA, int NB, int WPS>
__global__ __launch_bounds__(256, WPS) void k_mfma(double *out, stamp *st, int iters, double a0, double b0) {
  d4 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; i++)
#pragma unroll
    for (int j = 0; j < NB; j++) acc[i][j] = (d4){0, 0, 0, 0};
  double a[NA], b[NB];
#pragma unroll
  for (int i = 0; i < NA; i++) a[i] = a0 * (1 + i) + threadIdx.x * 1e-3;
#pragma unroll
  for (int j = 0; j < NB; j++) b[j] = b0 / (1 + j) - threadIdx.x * 1e-3;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NA; i++)
#pragma unroll
      for (int j = 0; j < NB; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NA; i++)
#pragma unroll
    for (int j = 0; j < NB; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = {c1 - c0, r1 - r0};
}

// 4x4 tile fed from LDS with RANDOM operands that change every step (what a real kernel does):
// constant operands under-state the power an MFMA draws and over-state the clock it can hold.
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_mfma_lds_random(double *out, stamp *st, int iters, double a0, double b0) {
  __shared__ double tab[8][16][64];  // 8 steps x (4 A + 4 B + spare) fragments of 64 lanes
  unsigned long long z = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1) + blockIdx.x;
  for (int k = threadIdx.x; k < 8 * 16 * 64; k += 256) {
    z ^= z << 13; z ^= z >> 7; z ^= z << 17;
    (&tab[0][0][0])[k] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (d4){0, 0, 0, 0};
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
    const int stp = (it + wave) & 7;
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = tab[stp][i][lane]; b[i] = tab[stp][4 + i][lane]; }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = {c1 - c0, r1 - r0};
}

template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_mfma_agpr(double *out, stamp *st, int iters, double a0, double b0) {
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (d4){0, 0, 0, 0};
  double a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { a[i] = a0 * (1 + i) + threadIdx.x * 1e-3; b[i] = b0 / (1 + i) - threadIdx.x * 1e-3; }
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = {c1 - c0, r1 - r0};
}

__global__ __launch_bounds__(256) void k_fma(double *out, stamp *st, int iters, double a0, double b0) {
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; i++) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = __builtin_fma(acc[i], a, b);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = {c1 - c0, r1 - r0};
}

// Mixed: workgroups alternate (per XCD) between an MFMA-only body and a v_fma_f64-only body, to
// see whether the FP64 matrix pipe and the FP64 vector pipe add up or share one datapath.
__global__ __launch_bounds__(512, 4) void k_mixed(double *out, stamp *st, int iters_mfma, int iters_fma, double a0, double b0) {
  const bool is_mfma = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4;  // waves 0-3 MFMA, 4-7 FMA: one of each per SIMD
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  if (is_mfma) {
    d4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = (d4){0, 0, 0, 0};
    double a[2], b[4];
#pragma unroll
    for (int i = 0; i < 2; i++) a[i] = a0 * (1 + i) + threadIdx.x * 1e-3;
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = b0 / (1 + j) - threadIdx.x * 1e-3;
    for (int it = 0; it < iters_mfma; it++) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = i;
    double a = 1.0000001 + threadIdx.x * 1e-9, b = 1e-9;
    for (int it = 0; it < iters_fma; it++) {
#pragma unroll
      for (int i = 0; i < 32; i++) acc[i] = __builtin_fma(acc[i], a, b);
    }
#pragma unroll
    for (int i = 0; i < 32; i++) s += acc[i];
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) st[blockIdx.x * 8 + (threadIdx.x >> 6)] = {c1 - c0, r1 - r0};
}

static double *out; static stamp *dst; static hipEvent_t e0, e1;

template <typename F>
static void run(const char *name, F launch, int grid, double ops_per_wave_iter, int iters, double flop_per_op) {
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0); launch(grid, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<stamp> h(grid * 4);
  hipMemcpy(h.data(), dst, h.size() * sizeof(stamp), hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (auto &s : h) { cyc += s.cyc; rt += s.rt; }
  cyc /= h.size(); rt /= h.size();
  double waves = grid * 4.0;
  double flop = waves * iters * ops_per_wave_iter * flop_per_op;
  printf("%-22s grid %4d: %8.3f ms %7.2f TFLOP/s | %.1f cycles per wave-op, in-kernel clock %.0f MHz\n", name, grid, ms,
         flop / ms / 1e9, cyc / (iters * ops_per_wave_iter), cyc / rt * 100.0);
}

int main() {
  hipMalloc(&out, 2048 * 256 * sizeof(double));
  hipMalloc(&dst, 2048 * 4 * sizeof(stamp));
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("--- operand scale 1 (values grow), then scale 1e-3 (values stay in [0,1]) ---\n");
  run("mfma 4x4 LDS random 2w/SIMD", [](int g, int it) { hipLaunchKernelGGL((k_mfma_lds_random<2>), dim3(g), dim3(256), 0, 0, out, dst, it, 0, 0); }, 512, 16, 40000, 2048);
  run("mfma 4x4 LDS random 3w/SIMD", [](int g, int it) { hipLaunchKernelGGL((k_mfma_lds_random<3>), dim3(g), dim3(256), 0, 0, out, dst, it, 0, 0); }, 768, 16, 40000, 2048);
  run("mfma 4x4 AGPR 1w/SIMD", [](int g, int it) { hipLaunchKernelGGL((k_mfma_agpr<1>), dim3(g), dim3(256), 0, 0, out, dst, it, 1e-3, 2e-3); }, 256, 16, 20000, 2048);
  run("mfma 4x4 AGPR 2w/SIMD", [](int g, int it) { hipLaunchKernelGGL((k_mfma_agpr<2>), dim3(g), dim3(256), 0, 0, out, dst, it, 1e-3, 2e-3); }, 512, 16, 20000, 2048);
  for (double sc : {1e-3}) {
    run("mfma 4x4 1w/SIMD", [sc](int g, int it) { hipLaunchKernelGGL((k_mfma<4, 4, 1>), dim3(g), dim3(256), 0, 0, out, dst, it, sc, 2.0 * sc); }, 256, 16, 20000, 2048);
    run("mfma 4x4 2w/SIMD", [sc](int g, int it) { hipLaunchKernelGGL((k_mfma<4, 4, 2>), dim3(g), dim3(256), 0, 0, out, dst, it, sc, 2.0 * sc); }, 512, 16, 20000, 2048);
    run("mfma 4x4 3w/SIMD", [sc](int g, int it) { hipLaunchKernelGGL((k_mfma<4, 4, 3>), dim3(g), dim3(256), 0, 0, out, dst, it, sc, 2.0 * sc); }, 768, 16, 20000, 2048);
    run("mfma 2x4 4w/SIMD", [sc](int g, int it) { hipLaunchKernelGGL((k_mfma<2, 4, 4>), dim3(g), dim3(256), 0, 0, out, dst, it, sc, 2.0 * sc); }, 1024, 8, 20000, 2048);
    run("mfma 2x2 4w/SIMD", [sc](int g, int it) { hipLaunchKernelGGL((k_mfma<2, 2, 4>), dim3(g), dim3(256), 0, 0, out, dst, it, sc, 2.0 * sc); }, 1024, 4, 40000, 2048);
    run("mfma 2x2 8w/SIMD", [sc](int g, int it) { hipLaunchKernelGGL((k_mfma<2, 2, 8>), dim3(g), dim3(256), 0, 0, out, dst, it, sc, 2.0 * sc); }, 2048, 4, 40000, 2048);
    run("mfma 1x1 8w/SIMD", [sc](int g, int it) { hipLaunchKernelGGL((k_mfma<1, 1, 8>), dim3(g), dim3(256), 0, 0, out, dst, it, sc, 2.0 * sc); }, 2048, 1, 100000, 2048);
  }
  {
    // 512 workgroups of 8 waves = 2 per CU: per SIMD 2 MFMA waves + 2 FMA waves
    const int g = 512, im = 20000, ifm = 80000;
    for (int mode = 0; mode < 3; mode++) {  // 0: both, 1: MFMA waves only, 2: FMA waves only
      float ms = 0;
      const int a = mode == 2 ? 0 : im, b = mode == 1 ? 0 : ifm;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mixed, dim3(g), dim3(512), 0, 0, out, dst, a, b, 1e-3, 2e-3);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      std::vector<stamp> h(256 * 8);
      hipMemcpy(h.data(), dst, h.size() * sizeof(stamp), hipMemcpyDeviceToHost);
      double cm = 0, cf = 0, rt = 0, cy = 0;
      for (size_t k = 0; k < h.size(); k++) { ((k & 7) < 4 ? cm : cf) += h[k].cyc; rt += h[k].rt; cy += h[k].cyc; }
      double f_mfma = g * 4.0 * a * 8 * 2048.0, f_fma = g * 4.0 * b * 32 * 128.0;
      printf("mixed mode %d (0 both,1 mfma,2 fma): %8.3f ms  mfma %.2f + fma %.2f = %.2f TFLOP/s | mfma-wave %.0f kcyc, fma-wave %.0f kcyc, clock %.0f MHz\n",
             mode, ms, f_mfma / ms / 1e9, f_fma / ms / 1e9, (f_mfma + f_fma) / ms / 1e9, cm / 1024 / 1e3, cf / 1024 / 1e3, cy / rt * 100);
    }
  }
  for (int grid : {256, 512, 1024})
    run("v_fma_f64 x32", [](int g, int it) { hipLaunchKernelGGL(k_fma, dim3(g), dim3(256), 0, 0, out, dst, it, 1.0000001, 1e-9); }, grid, 32, 20000, 128);
  return 0;
}
