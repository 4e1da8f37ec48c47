// tools/valu_rates.hip -- issue cost of the instructions the EM table kernel's search is made of, on gfx950:
// cycles per wave-instruction on one SIMD with W wavefronts per SIMD all issuing the same independent stream.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/valu_rates && tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(name, text)                                                                                   \
  __global__ __launch_bounds__(256) void name(unsigned long long *out, int iters) {                      \
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.000001, c = 0.5, d = 2.0;                                  \
    unsigned u = threadIdx.x, v = 12345u, w = 7u;                                                         \
    __shared__ double lds[1024];                                                                          \
    lds[threadIdx.x] = a;                                                                                 \
    __syncthreads();                                                                                      \
    unsigned la = (threadIdx.x & 63) * 8, lb = 0;                                                         \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                 \
    for (int i = 0; i < iters; i++) {                                                                     \
      asm volatile(REP16(text) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(u), "+v"(v), "+v"(w) : "v"(la), "v"(lb) : "vcc", "s20", "s21", "s22", "s23", "memory"); \
    }                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                 \
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;                          \
    if (a + b + c + d + u + v + w == 12345.678) out[0] = 1;                                               \
  }

BODY(k_fma64, "v_fma_f64 %0, %1, %2, %0\n")
BODY(k_mul64, "v_mul_f64 %0, %1, %2\n")
BODY(k_cmp64, "v_cmp_lt_f64 vcc, %1, %2\n")
BODY(k_cmp64s, "v_cmp_lt_f64 s[20:21], %1, %2\n")
BODY(k_cmpx64, "v_cmp_nlt_f64 s[22:23], %1, %2\n")
BODY(k_cmpu64, "v_cmp_lt_u64 vcc, %1, %2\n")
BODY(k_cmpu32, "v_cmp_lt_u32 vcc, %4, %5\n")
BODY(k_add32, "v_add_u32 %4, %5, %4\n")
BODY(k_cnd32, "v_cndmask_b32 %4, %5, %6, vcc\n")
BODY(k_rcp64, "v_rcp_f64 %0, %1\n")
BODY(k_min64, "v_min_f64 %0, %1, %2\n")
BODY(k_add64, "v_add_f64 %0, %1, %2\n")
BODY(k_readlane, "v_readlane_b32 s20, %4, 3\n")
BODY(k_pair, "v_cmp_nlt_f64 vcc, %1, %2\nv_add_u32 %4, %5, %4\n")
BODY(k_pair32, "v_cmp_lt_u32 vcc, %5, %6\nv_add_u32 %4, %5, %4\n")
BODY(k_ldsb128, "ds_read_b128 v[100:103], %8\n")
BODY(k_ldsb64, "ds_read_b64 v[100:101], %7\n")
BODY(k_ldsb64b, "ds_read_b64 v[100:101], %8\n")

int main() {
  unsigned long long *d;
  hipMalloc(&d, 1 << 20);
  struct K { const char *name; void (*f)(unsigned long long *, int); int per; };
  K ks[] = {{"v_fma_f64", k_fma64, 16}, {"v_mul_f64", k_mul64, 16}, {"v_add_f64", k_add64, 16}, {"v_min_f64", k_min64, 16},
            {"v_cmp_lt_f64 vcc", k_cmp64, 16}, {"v_cmp_lt_f64 sgpr", k_cmp64s, 16}, {"v_cmp_nlt_f64 sgpr", k_cmpx64, 16},
            {"v_cmp_lt_u64", k_cmpu64, 16}, {"v_cmp_lt_u32", k_cmpu32, 16}, {"v_add_u32", k_add32, 16},
            {"v_cndmask_b32", k_cnd32, 16}, {"v_rcp_f64", k_rcp64, 16}, {"v_readlane_b32", k_readlane, 16},
            {"cmp_f64 + add_u32 (pair)", k_pair, 16}, {"cmp_u32 + add_u32 (pair)", k_pair32, 16},
            {"ds_read_b128 broadcast", k_ldsb128, 16}, {"ds_read_b64 per-lane", k_ldsb64, 16}, {"ds_read_b64 broadcast", k_ldsb64b, 16}};
  const int iters = 2000;
  for (int wps : {1, 2, 4}) {
    printf("== %d wavefront(s) per SIMD (256-thread workgroups x %d per CU)\n", wps, wps);
    for (auto &k : ks) {
      hipLaunchKernelGGL(k.f, dim3(256 * wps), dim3(256), 0, 0, d, iters);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(k.f, dim3(256 * wps), dim3(256), 0, 0, d, iters);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(256 * wps * 4);
      hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      double sum = 0;
      for (auto x : h) sum += (double)x;
      const double cyc_per_wave_instr = sum / h.size() / (iters * (double)k.per);
      // each SIMD holds wps waves: SIMD cycles per instruction = wave-elapsed cycles per instr / wps
      printf("  %-28s %7.2f cycles per instruction and wavefront -> %6.2f SIMD cycles per wave-instruction\n", k.name,
             cyc_per_wave_instr, cyc_per_wave_instr / wps);
    }
  }
  return 0;
}
