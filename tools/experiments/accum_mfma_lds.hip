// NOT BUILT: an experiment kept for the record (round 1), with its profiles under profiles/r01_cfg3_mfma_lds_*.
// It was variant 2 / 3 of K1m and lost to the register-direct form of ngsdist_amd/csrc/accum_mfma.hip (51.6 / 54.5 ms
// against 45.5 on cfg 3); to try it again add it to ngsdist_amd/csrc/Makefile and call ngd_launch_accum_mfma_lds().
//
// accum_mfma_lds.hip -- K1m, second form: the same FP64-MFMA contraction as
// accum_mfma.hip (reference ngsDist.cpp:333-364, product branch of :353), with the
// operand panels of a 128x128 pair tile staged ONCE per workgroup in LDS by
// LDS-DMA (global_load_lds_dwordx4) instead of once per wavefront in registers.
//
// Why: in the register-direct form every operand fragment is fetched by the two
// wavefronts that need it, and L1 does not merge them (measured: 0.95 of the
// loads go on to L2, 0.78 of those to the fabric, 308 GB per cfg-3 launch against
// 49 GB compulsory -> the kernel sits at the fabric's ~6 TB/s, not at the FP64
// pipe).  Staging through LDS halves the requests at the source and takes the
// operands out of the VGPR budget, which keeps 3 wavefronts per SIMD resident.
//
// Ring of NS stages x KC k-groups; per stage each wavefront issues 4 DMA pieces of
// 1 KiB (2 fragments of 512 B, contiguous in the fragment-major image and in LDS),
// waits for its own pieces of the stage about to be consumed with an exact
// vmcnt, and one raw s_barrier publishes the stage to the workgroup.  The LDS
// image of a fragment is its lane image, so the MFMA operand read is a
// conflict-free ds_read_b64 at lane*8.
#include <cstdlib>

#include "ngd_internal.h"

namespace {

constexpr int WN = 4;  // MFMA tile columns per wavefront (64 pairs); rows WM = 4 (4 waves) or 2 (8 waves)
constexpr int KC = 2;          // k-groups per stage
constexpr int NS = 3;          // stages in the LDS ring
// stage = KC x (A: 8 fragments | B: 8 fragments) x 512 B = 16 KiB; ring = 48 KiB -> 3 workgroups per CU

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

template <bool WEIGHTED, int WM>
__global__ __launch_bounds__(64 * (8 / WM) * 2, WM == 4 ? 3 : 4) void k_accum_mfma_lds(
    const double *__restrict__ PA, const double *__restrict__ QB, const uint32_t *__restrict__ ws,
    const ngd_tile *__restrict__ tiles, uint32_t n_tiles, uint32_t n_ig, uint32_t n_pad,
    uint64_t kg_per_slice, uint64_t n_kg, double *__restrict__ slab) {
  __shared__ __attribute__((aligned(1024))) double ring[NS][KC][2][8][64];

  const uint32_t b = blockIdx.x;
  const uint32_t xcd = b & 7u, q = b >> 3;
  const uint32_t tile = q % n_tiles;
  const uint32_t ks = (q / n_tiles) * 8u + xcd;
  const uint32_t ti = tiles[tile].ti, tj = tiles[tile].tj;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int NW = (8 / WM) * 2;  // wavefronts per workgroup: (8/WM) rows x 2 columns of sub-tiles
  const int wi = wave >> 1, wj = wave & 1;
  // sub-tile rows [wi*WM*16, +WM*16) x columns [wj*64, +64): strictly below the diagonal of a diagonal
  // tile when its first row is past its last column -> loads its share, computes nothing
  const bool idle = (ti == tj && wi * WM >= (wj + 1) * WN);

  const uint64_t kg0 = (uint64_t)ks * kg_per_slice;
  uint64_t kg1 = kg0 + kg_per_slice;
  if (kg1 > n_kg) kg1 = n_kg;
  const uint32_t n_stage = kg0 < kg1 ? (uint32_t)((kg1 - kg0 + KC - 1) / KC) : 0;

  ngd_d4 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; m++)
#pragma unroll
    for (int n = 0; n < WN; n++) acc[m][n] = (ngd_d4){0, 0, 0, 0};

  // DMA duty of this wavefront.  A stage is 16 pieces of 1 KiB: (k-group, panel A|B, fragment pair 0..3);
  // with 4 waves each takes one (k-group, panel) = 4 pieces, with 8 waves half of one = 2 pieces.
  constexpr int PIECES = 16 / NW;
  const int d_unit = wave * PIECES / 4;                 // (k-group, panel) index 0..3
  const int d_kgl = d_unit >> 1, d_half = d_unit & 1;
  const int d_q0 = (wave * PIECES) % 4;                 // first fragment pair
  const uint64_t kstride = (uint64_t)n_ig * 64;
  const double *src0 = (d_half ? QB + (uint64_t)tj * NGD_IG_PER_TILE * 64 : PA + (uint64_t)ti * NGD_IG_PER_TILE * 64) +
                       (kg0 + d_kgl) * kstride + d_q0 * 128 + lane * 2;

  auto dma = [&](uint32_t stage) {  // stage index may run past the slice: tail padding keeps it in bounds
    const double *src = src0 + (uint64_t)stage * KC * kstride;
    double *dst = &ring[stage % NS][d_kgl][d_half][d_q0 * 2][0];
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 1024, 0);
    if (PIECES == 4) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 3072, 0);
    }
  };

  if (n_stage) {
    dma(0);
    dma(1);
    for (uint32_t t = 0; t < n_stage; t++) {
      // my pieces of stage t have landed when at most the PIECES pieces of stage t+1 are still in flight
      if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // everyone's pieces of stage t are in; everyone is done reading stage t-1
      dma(t + 2);                    // refill the slot stage t-1 occupied
      if (!idle) {
        const uint32_t slot = t % NS;
#pragma unroll
        for (int kgl = 0; kgl < KC; kgl++) {
          if (kg0 + (uint64_t)t * KC + kgl >= kg1) break;  // slice not a whole number of stages
          double a[WM], bq[WN];
#pragma unroll
          for (int m = 0; m < WM; m++) a[m] = ring[slot][kgl][0][wi * WM + m][lane];
#pragma unroll
          for (int n = 0; n < WN; n++) bq[n] = ring[slot][kgl][1][wj * WN + n][lane];
          if (WEIGHTED) {  // bootstrap multiplicity of the site of this lane's k (ngsDist.cpp:426-434)
            const uint64_t kbase = (kg0 + (uint64_t)t * KC + kgl) * 4;
            const uint32_t w0 = ws[(kbase + 0) / 3], w1 = ws[(kbase + 1) / 3], w2 = ws[(kbase + 2) / 3],
                           w3 = ws[(kbase + 3) / 3];
            const int kl = lane >> 4;
            const double w = (double)(kl == 0 ? w0 : kl == 1 ? w1 : kl == 2 ? w2 : w3);
#pragma unroll
            for (int m = 0; m < WM; m++) a[m] *= w;
          }
#pragma unroll
          for (int m = 0; m < WM; m++)
#pragma unroll
            for (int n = 0; n < WN; n++)
              acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], bq[n], acc[m][n], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead before LDS is released
  }
  if (idle) return;

  const uint32_t ig0 = ti * NGD_IG_PER_TILE + wi * WM;
  const uint32_t jg0 = tj * NGD_IG_PER_TILE + wj * WN;
  double *out = slab + (uint64_t)ks * n_pad * n_pad;
#pragma unroll
  for (int m = 0; m < WM; m++)
#pragma unroll
    for (int n = 0; n < WN; n++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t i = (ig0 + m) * 16 + (lane >> 4) + 4 * r;
        const uint32_t j = (jg0 + n) * 16 + (lane & 15);
        out[(uint64_t)i * n_pad + j] = acc[m][n][r];
      }
}

}  // namespace

void ngd_launch_accum_mfma_lds(hipStream_t st, const ngd_geom &g, const double *PA, const double *QB,
                               const uint32_t *d_ws, const ngd_tile *d_tiles, uint32_t n_tiles,
                               uint32_t n_ks, uint64_t kg_per_slice, uint64_t n_kg_eff, double *slab) {
  if (!n_tiles) return;
  dim3 grid(n_tiles * n_ks);
  static const int variant = [] {
    const char *v = getenv("NGD_MFMA_VARIANT");
    return v && *v ? atoi(v) : 0;
  }();
#define NGD_LDS(W, M)                                                                                      \
  hipLaunchKernelGGL((k_accum_mfma_lds<W, M>), grid, dim3(64 * (8 / M) * 2), 0, st, PA, QB, d_ws, d_tiles, \
                     n_tiles, g.n_ig, g.n_pad, kg_per_slice, n_kg_eff, slab)
  if (variant == 3) {  // 8 wavefronts of 32 x 64 pairs, 4 per SIMD
    if (d_ws) NGD_LDS(true, 2); else NGD_LDS(false, 2);
  } else {             // 4 wavefronts of 64 x 64 pairs, 3 per SIMD
    if (d_ws) NGD_LDS(true, 4); else NGD_LDS(false, 4);
  }
#undef NGD_LDS
}
