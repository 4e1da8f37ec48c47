// accum_mfma.hip -- the --indep_geno accumulation of gen_dist() (reference
// ngsDist.cpp:333-364 with the product branch of :353) for ALL pairs at once.
//
// With independent genotypes the per-pair sum is a contraction over k = (site,
// genotype):   sum[i1][i2] = SUM_k P[i1][k] * Q[i2][k],   Q = score . P per site
// (the nine score-weighted products of :351-353 regrouped as three), i.e. the
// upper triangle of P.Q^T with K = 3*n_sites.  Streaming it pair by pair moves
// 48 B per pair-site (accum_stream.hip); tiling it re-uses every operand 128
// times and leaves the FP64 pipe as the only limit.  On gfx950 the FP64 matrix
// rate equals the FP64 vector rate, but v_mfma_f64_16x16x4_f64 takes ONE double
// per lane per operand for 1024 FMAs, so operand delivery (VGPR/LDS/L1 traffic)
// drops 16x versus v_fma_f64 register tiles.
//
// Wavefront  = one JOB = a 64x64 block of pairs = 4x4 MFMA tiles, 16 accumulators of 4 doubles.
// Workgroup  = 4 jobs x one slice of k.  The four jobs of an off-diagonal 128x128 tile share a
// workgroup (their operands overlap); a diagonal tile gives one full job and two jobs ON the
// diagonal, whose MFMA pattern is the upper triangle only (10 of 16 tiles) -- same launch, same
// slice, so their rows are read while the off-diagonal jobs of that slice have them in L2 (as a
// separate launch the diagonal jobs re-streamed the whole data set: 49 GB for 8 % of the flops).
// Job lists are built by the engine (engine.hip).
// Operands come straight from the fragment-major images (ngd_internal.h): one
// coalesced 512-B global load per 16x4 operand, software-pipelined DEPTH k-groups
// ahead in registers; the 4 wavefronts of a tile share operands through L1/L2.
// Block ids are dealt so that all tiles of one k-slice run on one XCD at about
// the same time, which keeps the slice's operand panel in that XCD's L2.
// Slices are written as slabs and summed in fixed order by reduce.hip
// (deterministic; no floating-point atomics).
#include <cstdlib>

#include "ngd_internal.h"

namespace {

constexpr int WM = 4, WN = 4;  // MFMA tiles per wavefront edge

// DEPTH = k-groups of operands in flight per wavefront (register ring, <= NGD_KG_TAIL);
// WPS   = wavefronts per SIMD the register budget is held to (workgroups per CU).
// One wavefront alone issues an f64 MFMA only every ~138 cycles (measured,
// profiles/r01_fp64_peak_microbench.txt); the 64-cycle pipe rate needs two
// wavefronts per SIMD that are BOTH in their MFMA phase, so a third resident
// wavefront is what covers the others' load/wait/epilogue phases.
// TRI_JOBS: build the triangular pattern for blocks on the diagonal (else they run the full one)
template <bool WEIGHTED, int DEPTH, int WPS, bool TRI_JOBS>
__global__ __launch_bounds__(256, WPS) void k_accum_mfma(
    const double *__restrict__ PA, const double *__restrict__ QB, const uint32_t *__restrict__ ws,
    const ngd_tile *__restrict__ jobs, uint32_t n_tiles /* workgroups per slice */, uint32_t n_ig,
    uint32_t n_pad, uint64_t kg_per_slice, uint64_t n_kg, double *__restrict__ slab) {
  // XCD-aware deal: blocks b and b+8 share an XCD (round-robin dispatch; speed only).
  const uint32_t b = blockIdx.x;
  const uint32_t xcd = b & 7u, q = b >> 3;
  const uint32_t tile = q % n_tiles;
  const uint32_t ks = (q / n_tiles) * 8u + xcd;
  // the wavefront index is uniform: say so, so that operand addresses live in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const ngd_tile job = jobs[tile * 4 + wave];  // first row / column group of the 64x64 block
  if (job.ti == 0xFFFFu) return;               // padding entry of the job list
  // blocks ON the diagonal carry a flag: their MFMA tiles below the diagonal hold no pair i1 < i2
  const bool tri = __builtin_amdgcn_readfirstlane(job.tj >> 15) != 0;
  const uint32_t ig0 = job.ti, jg0 = job.tj & 0x7FFFu;

  const uint64_t kg0 = (uint64_t)ks * kg_per_slice;
  uint64_t kg1 = kg0 + kg_per_slice;
  if (kg1 > n_kg) kg1 = n_kg;

  ngd_d4 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; m++)
#pragma unroll
    for (int n = 0; n < WN; n++) acc[m][n] = (ngd_d4){0, 0, 0, 0};

  const uint64_t kstride = (uint64_t)n_ig * 64;
  const double *pa = PA + (uint64_t)ig0 * 64;  // wave-uniform bases (SGPR); the lane adds lane*8 bytes
  const double *pb = QB + (uint64_t)jg0 * 64;
  const uint32_t lane_off = lane * 8;

  double a[DEPTH][WM], bq[DEPTH][WN];
  uint32_t wq[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; d++) wq[d] = 0;

  // The operand pipeline is issued by hand.  hipcc's waitcnt pass puts a full
  // `s_waitcnt vmcnt(0)` at the head of any loop whose loads are consumed one
  // trip later, which would expose the whole memory latency once per trip; so the
  // loads are inline asm (invisible to that pass) and every consume step waits
  // with an exact count: the LPF*(DEPTH-1) youngest loads may stay in flight.
  // Each wait names the registers it guards as "+v" operands, which pins the
  // MFMAs that read them behind it.
  //
  // Loads are unconditional: slices are whole multiples of DEPTH k-groups and the
  // images carry DEPTH zeroed k-groups of tail padding, so the run-ahead past the
  // slice end stays in bounds and is never consumed (it is drained before the
  // epilogue, because the compiler is free to re-use those registers there).
  constexpr int LPF = WM + WN + (WEIGHTED ? 1 : 0);  // loads per fetch
  auto fetch = [&](int d, uint64_t kg) {
    const double *xa = pa + kg * kstride;
    const double *xb = pb + kg * kstride;
    asm volatile(
        "global_load_dwordx2 %0, %4, %5\n\t"
        "global_load_dwordx2 %1, %4, %5 offset:512\n\t"
        "global_load_dwordx2 %2, %4, %5 offset:1024\n\t"
        "global_load_dwordx2 %3, %4, %5 offset:1536"
        : "=&v"(a[d][0]), "=&v"(a[d][1]), "=&v"(a[d][2]), "=&v"(a[d][3])
        : "v"(lane_off), "s"(xa));
    asm volatile(
        "global_load_dwordx2 %0, %4, %5\n\t"
        "global_load_dwordx2 %1, %4, %5 offset:512\n\t"
        "global_load_dwordx2 %2, %4, %5 offset:1024\n\t"
        "global_load_dwordx2 %3, %4, %5 offset:1536"
        : "=&v"(bq[d][0]), "=&v"(bq[d][1]), "=&v"(bq[d][2]), "=&v"(bq[d][3])
        : "v"(lane_off), "s"(xb));
    if (WEIGHTED) {
      const uint32_t *xw = ws + (kg * 4 + (uint64_t)(lane >> 4)) / 3;
      asm volatile("global_load_dword %0, %1, off" : "=&v"(wq[d]) : "v"(xw));
    }
  };
  auto arrive = [&](int d) {  // wait until fetch #d (the oldest outstanding) has landed
    asm volatile("s_waitcnt vmcnt(%9)"
                 : "+v"(a[d][0]), "+v"(a[d][1]), "+v"(a[d][2]), "+v"(a[d][3]), "+v"(bq[d][0]),
                   "+v"(bq[d][1]), "+v"(bq[d][2]), "+v"(bq[d][3]), "+v"(wq[d])
                 : "n"(LPF * (DEPTH - 1)));
  };
  static_assert(WM == 4 && WN == 4, "the asm fetch is written for 4+4 operands");
  static_assert(DEPTH >= 1 && DEPTH <= NGD_KG_TAIL, "tail padding must cover the run-ahead");

  // Prologue, k loop and drain as ONE unit per MFMA pattern (full: 16 tiles, triangular: 10).  The
  // two instantiations must not share a control-flow join while loads are in flight: the asm loads
  // are invisible to the compiler, so a register copy it placed at such a join would read a
  // register the load has not written yet.
  auto run = [&](auto tri_c) {
    constexpr bool TRI = decltype(tri_c)::value;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) fetch(d, kg0 + d);
    for (uint64_t kg = kg0; kg < kg1; kg += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
        arrive(d);
        if (DEPTH == 1 || kg + d < kg1) {  // a slice need not be a whole number of ring trips
          if (WEIGHTED) {  // bootstrap multiplicity of the site (ngsDist.cpp:426-434)
            const double w = (double)wq[d];
#pragma unroll
            for (int m = 0; m < WM; m++) a[d][m] *= w;
          }
#pragma unroll
          for (int m = 0; m < WM; m++)
#pragma unroll
            for (int n = 0; n < WN; n++)
              if (!TRI || m <= n)
                acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d][m], bq[d][n], acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the refill BEHIND the MFMAs that read the buffer
        fetch(d, kg + d + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead
    __builtin_amdgcn_sched_barrier(0);
  };

  if (kg0 < kg1) {
    if (TRI_JOBS && tri) run(std::true_type{}); else run(std::false_type{});
  }

  // D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*r
  double *out = slab + (uint64_t)ks * n_pad * n_pad;
#pragma unroll
  for (int m = 0; m < WM; m++)
#pragma unroll
    for (int n = 0; n < WN; n++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (TRI_JOBS && tri && m > n) continue;
        const uint32_t i = (ig0 + m) * 16 + (lane >> 4) + 4 * r;
        const uint32_t j = (jg0 + n) * 16 + (lane & 15);
        out[(uint64_t)i * n_pad + j] = acc[m][n][r];
      }
}

}  // namespace

void ngd_launch_accum_mfma(hipStream_t st, const ngd_geom &g, const double *PA, const double *QB,
                           const uint32_t *d_ws, const ngd_tile *d_jobs, uint32_t n_wg, uint32_t n_ks,
                           uint64_t kg_per_slice, uint64_t n_kg_eff, double *slab) {
  if (!n_wg) return;
  // n_ks is a multiple of 8 (see the deal in the kernel)
  static const int variant = [] {
    const char *v = getenv("NGD_MFMA_VARIANT");
    return v && *v ? atoi(v) : 0;
  }();
  static const int tri_jobs = [] {
    const char *v = getenv("NGD_MFMA_TRI");
    return v && *v ? atoi(v) : 0;
  }();
#define NGD_MFMA(W, D, P)                                                                                    \
  do {                                                                                                       \
    if (tri_jobs)                                                                                            \
      hipLaunchKernelGGL((k_accum_mfma<W, D, P, true>), dim3(n_wg * n_ks), dim3(256), 0, st, PA, QB, d_ws,    \
                         d_jobs, n_wg, g.n_ig, g.n_pad, kg_per_slice, n_kg_eff, slab);                       \
    else                                                                                                     \
      hipLaunchKernelGGL((k_accum_mfma<W, D, P, false>), dim3(n_wg * n_ks), dim3(256), 0, st, PA, QB, d_ws,   \
                         d_jobs, n_wg, g.n_ig, g.n_pad, kg_per_slice, n_kg_eff, slab);                       \
  } while (0)
  // variant 0 (default): no in-wave run-ahead, 3 wavefronts per SIMD -- measured fastest
  // (profiles/r01_*): the third wavefront covers the others' load phases.
  // variant 1: 4-deep register ring, 2 wavefronts per SIMD.
  // (a 2-deep ring at 3 wavefronts per SIMD needs 168+ VGPRs and spills: not built)
  if (d_ws) {
    if (variant == 1) NGD_MFMA(true, 4, 2); else NGD_MFMA(true, 1, 3);
  } else {
    if (variant == 1) NGD_MFMA(false, 4, 2); else NGD_MFMA(false, 1, 3);
  }
#undef NGD_MFMA
}
