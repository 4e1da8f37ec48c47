// A hazard that exists only ACROSS a loop's back edge (test fixture for tools/check_asm_loads.py; never built into the
// library): the prologue's load is waited for, so the first trip is clean in program order -- but every trip issues
// the next fragment's load at its bottom and the NEXT trip adds the register up before any wait.  A walk that does not
// follow the back edge sees nothing; k_fixed waits at the head of every trip.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <bool FIXED>
__global__ void k_ring(const double *__restrict__ p, double *__restrict__ out, uint32_t n) {
  const uint32_t off = threadIdx.x * 8;
  double a, acc = 0;
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(a) : "v"(off), "s"(p));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (uint32_t i = 0; i < n; i++) {
    if (FIXED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(a));
    acc += a;  // trip 2 on: the load issued at the bottom of the trip before may not have landed
    p += 64;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(a) : "v"(off), "s"(p));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[threadIdx.x] = acc;
}

template __global__ void k_ring<false>(const double *, double *, uint32_t);
template __global__ void k_ring<true>(const double *, double *, uint32_t);
