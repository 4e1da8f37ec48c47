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
// workgroup (their operands overlap); the blocks of the diagonal tiles follow, packed four to a
// workgroup -- same launch, same slice, so their rows are read while the off-diagonal jobs of that
// slice have them in L2 (as a separate launch the diagonal jobs re-streamed the whole data set:
// 49 GB for 8 % of the flops).  Job lists are built by the engine (engine.hip).
// Operands come straight from the fragment-major images (ngd_internal.h): one
// coalesced 512-B global load per 16x4 operand, software-pipelined DEPTH k-groups
// ahead in registers; the 4 wavefronts of a tile share operands through L1/L2.
// Block ids are dealt so that all tiles of one k-slice run on one XCD at about
// the same time, which keeps the slice's operand panel in that XCD's L2.
// Slices are written as slabs and summed in fixed order by reduce.hip
// (deterministic; no floating-point atomics).
#include "ngd_internal.h"

namespace {

constexpr int WN = 4;  // MFMA tiles per wavefront edge (columns; rows: 4, or 2 in the small-job form)

// DEPTH = k-groups of operands in flight per wavefront (register ring, <= NGD_KG_TAIL);
// WPS   = wavefronts per SIMD the register budget is held to (workgroups per CU).
// One wavefront alone issues an f64 MFMA only every ~138 cycles (measured,
// profiles/r01_fp64_peak_microbench.txt); the 64-cycle pipe rate needs two
// wavefronts per SIMD that are BOTH in their MFMA phase, so a third resident
// wavefront is what covers the others' load/wait/epilogue phases.
// one operand fragment: 512 B for the wavefront, lane l takes bytes [8l, 8l+8) at base + OFF
template <int OFF>
__device__ __forceinline__ void load_frag(double &dst, uint32_t lane_off, const double *base) {
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(lane_off), "s"(base), "n"(OFF));
}

// EXACT: build one code path per block shape (rows x cols MFMA tiles, triangular on the diagonal) so
// that no MFMA is issued for padding or for the lower triangle; otherwise every block runs the full
// 4x4 pattern.  EXACT is for small n_ind, where up to half of a 64x64 block grid is padding: blocks of
// different shapes progress at different rates through k, which costs the L2 residency that large
// n_ind depends on (measured: 45.5 -> 51.8 ms at n_ind = 1000) but nothing when the whole operand
// panel of a slice is a few tens of KB.
// EXACT = 2: the same with blocks of at most 2 x 4 tiles (8 accumulators: 6 wavefronts per SIMD instead of 3).  At a few
// hundred individuals a slice is ~10 jobs and a launch one dispatch round: with 4 x 4 blocks a SIMD's three jobs are
// what the dispatch order gives it (10 + 16 + 16 tiles here, 4 + 4 + 16 there), and a 16-tile job alone on the pipe
// issues an MFMA only every 138 cycles -- the launch lasts as long as the unluckiest SIMD.  Smaller jobs, more of them
// per SIMD: shorter critical chains, six wavefronts to cover each other's operand fetches.
// EXACT = 3: blocks of 4 x 4 tiles like EXACT = 1, but ALL jobs of a slice in one workgroup (up to 12 wavefronts) that
// meet at a barrier every k-group: the jobs of a slice then read the same k-group at the same time, so each operand
// fragment leaves HBM once and serves its other readers from the CU's own L1 -- single-wavefront jobs of different
// shapes drift apart by more than the L2 holds (one-tile jobs run four times as fast as 16-tile ones), and the launch
// is then bound by the fabric re-delivering fragments, not by the FP64 pipe.
// EXACT = 4: the two together -- blocks of at most 2 x 4 tiles, up to 16 of them in one workgroup in step: a SIMD's
// four wavefronts then have MFMAs to interleave through the whole k-group (a 16-tile job beside 4-tile ones issues its
// second half alone, at the one-wavefront rate of an MFMA every 138 cycles instead of 64).
// EXACT = 5: EXACT = 4's jobs with the slice's operand fragments staged through LDS: the workgroup fetches each of the
// k-group's 2 x n_igv fragments ONCE (a wavefront takes one or two), and every job reads its six from LDS.  [measured]
// cfg 2 (tools/exact_diag.sh): with every job fetching its own operands a k-group took 2380 cycles -- 1810 with the
// loads taken out, 2250 with the MFMAs taken out: the ~90 fragment loads of a k-group (47 KB through one CU's vector L1)
// bound it, not HBM (3.4 TB/s) and not the FP64 pipe (61 % busy).
template <bool WEIGHTED, int DEPTH, int WPS, int EXACT>
__global__ __launch_bounds__(EXACT >= 4 ? 1024 : EXACT == 3 ? 768 : EXACT ? 64 : 256, WPS) void k_accum_mfma(
    const double *__restrict__ PA, const double *__restrict__ QB, const double *__restrict__ wk,
    const uint32_t *__restrict__ kgl, const ngd_job *__restrict__ jobs, uint32_t n_tiles /* workgroups per slice */, uint32_t n_ig,
    uint32_t n_pad, uint64_t kg_per_slice, uint64_t n_kg, uint64_t k_per_slice, uint32_t w_slice_stride,
    double *__restrict__ slab, uint32_t n_igv_touch = 0, unsigned long long *__restrict__ clk = nullptr,
    uint32_t ks0 = 0, uint32_t resume = 0) {
  // XCD-aware deal: blocks b and b+8 share an XCD (round-robin dispatch; speed only).
  const uint32_t b = blockIdx.x;
  const uint32_t xcd = b & 7u, q = b >> 3;
  const uint32_t tile = q % n_tiles;
  const uint32_t ks = ks0 + (q / n_tiles) * 8u + xcd;  // (ks0: a launch over a range of the pass's slices)
  // the wavefront index is uniform: say so, so that operand addresses live in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // EXACT = 3 with n_igv_touch != 0: the workgroup's last wavefront computes nothing -- it reads every operand fragment
  // of the slice two k-groups ahead of the jobs, so that their own fetches find the lines on their way or in the cache
  // (the jobs have no registers left for a deeper operand ring; this wavefront needs none)
  constexpr bool TOUCH = EXACT == 3 || EXACT == 4;
  const uint32_t n_job_waves = (blockDim.x >> 6) - (TOUCH && n_igv_touch ? 1u : 0u);
  const bool toucher = TOUCH && n_igv_touch && (uint32_t)wave == n_job_waves;
  const ngd_job job = toucher ? ngd_job{0, 0, 1, 1, 0, 0} : jobs[tile * n_job_waves + wave];  // 4 jobs per workgroup, or 1 (EXACT)
  if (job.rows == 0) return;  // padding entry of the job list
  // the shader clock this launch runs at: one wavefront in the middle of the grid reads the shader-cycle counter and
  // the constant-rate one around its work (roofline accounting of bench.py; ngd_last_shader_clock())
  const bool clk_wave = clk && b == (gridDim.x >> 1) && wave == 0;
  unsigned long long clk_t0 = 0, clk_r0 = 0;
  if (clk_wave) {
    clk_t0 = __builtin_amdgcn_s_memtime();
    clk_r0 = __builtin_amdgcn_s_memrealtime();
  }
  const uint32_t ig0 = job.ig0, jg0 = job.jg0;
  // shape code, wave-uniform: rows | cols << 3 | tri << 6
  const uint32_t shape = __builtin_amdgcn_readfirstlane((uint32_t)job.rows | ((uint32_t)job.cols << 3) |
                                                        ((uint32_t)job.tri << 6));

  // A slice is kg_per_slice whole k-groups -- or, for bootstrap blocks that are not whole k-groups
  // (k_per_slice != 0: a block of B sites is 3 B contraction indices), every k-group the block touches,
  // with per-slice weights (1 inside the block, 0 outside; w_slice_stride k-groups of them per slice)
  // masking the indices that the first and last k-group share with the neighbouring blocks.
  uint64_t kg0 = (uint64_t)ks * kg_per_slice;
  uint64_t kg1 = kg0 + kg_per_slice;
  if (k_per_slice) {
    kg0 = ((uint64_t)ks * k_per_slice) >> 2;
    kg1 = ((uint64_t)(ks + 1) * k_per_slice + 3) >> 2;
  }
  if (kg1 > n_kg) kg1 = n_kg;
  if (TOUCH && toucher) {  // (wave-uniform)
    const uint64_t ks_d = (uint64_t)n_ig * 64;
    const uint32_t lo = lane * 8;
    // The loads land in `sink` whenever they land: the register must stay this wavefront's own from the first load to
    // the final wait.  Every load therefore READS and writes it ("+v": one unbroken def-use chain) -- as a plain output
    // the compiler counted it dead between two loads and put the loop bound there, which a late-returning fragment then
    // overwrote (found by tools/fuzz_parity.py case 40501: 64 individuals, no zero padding in the last group).
    double sink = 0.0;
    auto touch = [&](uint64_t kg) {
      const double *xa = PA + kg * ks_d, *xb = QB + kg * ks_d;
      for (uint32_t gi = 0; gi < n_igv_touch; gi++) {
        asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(sink) : "v"(lo), "s"(xa + gi * 64));
        asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(sink) : "v"(lo), "s"(xb + gi * 64));
      }
    };
    // (plain k-group ranges only: the engine gives this form no k-group list and no per-slice weights; a slice past the
    // end of the data -- kg0 >= kg1 -- is skipped like the jobs skip it: no barrier, and nothing read out of bounds)
    constexpr int TD = EXACT == 4 ? DEPTH : 1;  // the jobs' own operand ring: as many barriers per trip as they execute
    static_assert(2 * TD + 1 <= NGD_KG_TAIL, "the prefetch runs 2 * TD + 1 k-groups past the slice at most");
    if (kg0 < kg1) touch(kg0 + TD);
    for (uint64_t kg = kg0; kg < kg1; kg += TD) {
#pragma unroll
      for (int d = 0; d < TD; d++) {
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // at most 24 + 2 x 16 loads in flight (the counter holds 63)
        touch(kg + d + TD + 1);  // (past the slice: the next slice's k-groups or the images' zeroed tail)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)::"memory");
    return;
  }
  const double *wk_s = wk;  // weight of real k-group kg: wk_s + (kg - wk_kg0) * 4
  uint64_t wk_kg0 = 0;
  if (WEIGHTED && w_slice_stride) {
    wk_s = wk + (uint64_t)ks * w_slice_stride * 4;
    wk_kg0 = kg0;
  }

  constexpr int WM = (EXACT == 2 || EXACT >= 4) ? 2 : 4;
  constexpr bool SYNC = EXACT >= 3;
  ngd_d4 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; m++)
#pragma unroll
    for (int n = 0; n < WN; n++) acc[m][n] = (ngd_d4){0, 0, 0, 0};

  const uint64_t kstride = (uint64_t)n_ig * 64;
  const double *pa = PA + (uint64_t)ig0 * 64;  // wave-uniform bases (SGPR); the lane adds lane*8 bytes
  const double *pb = QB + (uint64_t)jg0 * 64;
  const uint32_t lane_off = lane * 8;

  constexpr int RING = DEPTH;
  double a[RING][4], bq[RING][WN];  // (the small-job form uses two of the four row operands)
  double wq[RING];  // bootstrap multiplicity of this lane's k (wk[4 kg + lane/16])
#pragma unroll
  for (int d = 0; d < RING; d++) wq[d] = 0;
  uint32_t lane_wk = (lane >> 4) * 8;  // (not const: a nested generic lambda must capture it)

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
  // WEIGHTED (one bootstrap replicate): the slice is a range of the LIST of k-groups that have a non-zero
  // weight (layout.hip); kidx() maps a list position to the k-group, a scalar load issued one trip ahead of
  // the fetch that needs it.  Every job of a slice walks the same list, so the slice still moves as one.
  auto kidx = [&](uint64_t pos) -> uint64_t { return (WEIGHTED && kgl) ? (uint64_t)kgl[pos] : pos; };
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
      const double *xw = wk_s + (kg - wk_kg0) * 4;
      asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(wq[d]) : "v"(lane_wk), "s"(xw));
    }
  };
  auto arrive = [&](int d) {  // wait until fetch #d (the oldest outstanding) has landed
    asm volatile("s_waitcnt vmcnt(%9)"
                 : "+v"(a[d][0]), "+v"(a[d][1]), "+v"(a[d][2]), "+v"(a[d][3]), "+v"(bq[d][0]),
                   "+v"(bq[d][1]), "+v"(bq[d][2]), "+v"(bq[d][3]), "+v"(wq[d])
                 : "n"(LPF * (DEPTH - 1)));
  };
  static_assert(WN == 4, "the asm fetch is written for 4+4 operands");
  static_assert(DEPTH >= 1 && DEPTH <= NGD_KG_TAIL, "tail padding must cover the run-ahead");

  // Prologue, k loop and drain as ONE unit per MFMA pattern (full: 16 tiles, triangular: 10).  The
  // two instantiations must not share a control-flow join while loads are in flight: the asm loads
  // are invisible to the compiler, so a register copy it placed at such a join would read a
  // register the load has not written yet.
  // same_c (a block ON the diagonal of an engine whose two operands are ONE image: single_image = 2, PA == QB and
  // ig0 == jg0): the block's row fragments ARE its column fragments -- four loads per k-group instead of eight, the row
  // operand is the column operand times the index weight.
  auto run = [&](auto rows_c, auto cols_c, auto tri_c, auto same_c) {
    constexpr int PM = decltype(rows_c)::value, PN = decltype(cols_c)::value;
    constexpr bool TRI = decltype(tri_c)::value, SAME = decltype(same_c)::value;
    static_assert(!SAME || (DEPTH == 1 && PM == PN), "shared operands: the one-deep form (its wait is vmcnt(0)), square blocks");
    auto fetch_r = [&](int d, uint64_t kg) {
      if constexpr (SAME) {
        const double *xb = pb + kg * kstride;
        asm volatile(
            "global_load_dwordx2 %0, %4, %5\n\t"
            "global_load_dwordx2 %1, %4, %5 offset:512\n\t"
            "global_load_dwordx2 %2, %4, %5 offset:1024\n\t"
            "global_load_dwordx2 %3, %4, %5 offset:1536"
            : "=&v"(bq[d][0]), "=&v"(bq[d][1]), "=&v"(bq[d][2]), "=&v"(bq[d][3])
            : "v"(lane_off), "s"(xb));
        if (WEIGHTED) {
          const double *xw = wk_s + (kg - wk_kg0) * 4;
          asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(wq[d]) : "v"(lane_wk), "s"(xw));
        }
      } else {
        fetch(d, kg);
      }
    };
    uint64_t nxt[DEPTH];  // the k-group the next refill of buffer d fetches
#pragma unroll
    for (int d = 0; d < DEPTH; d++) fetch_r(d, kidx(kg0 + d));
#pragma unroll
    for (int d = 0; d < DEPTH; d++) nxt[d] = kidx(kg0 + d + DEPTH);
    for (uint64_t kg = kg0; kg < kg1; kg += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
        if constexpr (SAME) {  // (DEPTH == 1: everything outstanding is this fetch)
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[d][0]), "+v"(bq[d][1]), "+v"(bq[d][2]), "+v"(bq[d][3]), "+v"(wq[d]));
#pragma unroll
          for (int m = 0; m < WM; m++) a[d][m] = bq[d][m];
        } else {
          arrive(d);
        }
        if (DEPTH == 1 || kg + d < kg1) {  // a slice need not be a whole number of ring trips
          if (WEIGHTED) {  // bootstrap multiplicity of the site (ngsDist.cpp:426-434)
            const double w = wq[d];
#pragma unroll
            for (int m = 0; m < WM; m++) a[d][m] *= w;
          }
#if defined(NGD_QB_PROBE)
          // Diagnostic build (tools/qb_probe.sh), never shipped: what forming q = score . p at consume time would cost
          // instead of reading the second image QB.  With the contraction index ordered (site quad, genotype, site) a
          // lane holds the three genotypes of ONE site in three consecutive fragments, so q takes 9 FMAs per lane and
          // column group per 3 k-groups (6 with a zero score diagonal): NGD_QB_PROBE = 12 (or 8) neutral FP64 FMAs per
          // k-group on the B fragments stand in for them (x * 1 + 0: results unchanged).
#pragma unroll
          for (int q = 0; q < NGD_QB_PROBE; q++)
            asm volatile("v_fma_f64 %0, %0, 1.0, 0" : "+v"(bq[d][q % WN]));
#endif
#pragma unroll
          for (int m = 0; m < WM; m++)
#pragma unroll
            for (int n = 0; n < WN; n++)
              if (m < PM && n < PN && (!TRI || m <= n))
                acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d][m], bq[d][n], acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the refill BEHIND the MFMAs that read the buffer
        fetch_r(d, nxt[d]);
        __builtin_amdgcn_sched_barrier(0);
        nxt[d] = kidx(kg + d + 2 * DEPTH);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead
    __builtin_amdgcn_sched_barrier(0);
  };
  // The same unit for a block of PM x PN MFMA tiles (upper triangle if TRI) with a D-deep ring: only
  // the PM + PN operand fragments it needs are loaded.  (Deeper rings for the narrow shapes were tried:
  // D = 2 and 4 lose 10-20 % to D = 1 at n_ind = 200..300.)
  auto run_exact = [&](auto rows_c, auto cols_c, auto tri_c, auto depth_c) {
    constexpr int PM = decltype(rows_c)::value, PN = decltype(cols_c)::value, D = decltype(depth_c)::value;
    constexpr bool TRI = decltype(tri_c)::value;
    constexpr int LOADS = PM + PN + (WEIGHTED ? 1 : 0);
    const uint32_t lwk = lane_wk;  // named here so that the nested lambda captures it
    auto fetch_x = [&](int d, uint64_t kg) {
      const double *xa = pa + kg * kstride;
      const double *xb = pb + kg * kstride;
      if (PM > 0) load_frag<0>(a[d][0], lane_off, xa);
      if (PM > 1) load_frag<512>(a[d][1], lane_off, xa);
      if (PM > 2) load_frag<1024>(a[d][2], lane_off, xa);
      if (PM > 3) load_frag<1536>(a[d][3], lane_off, xa);
      if (PN > 0) load_frag<0>(bq[d][0], lane_off, xb);
      if (PN > 1) load_frag<512>(bq[d][1], lane_off, xb);
      if (PN > 2) load_frag<1024>(bq[d][2], lane_off, xb);
      if (PN > 3) load_frag<1536>(bq[d][3], lane_off, xb);
      if (WEIGHTED) {
        const double *xw = wk_s + (kg - wk_kg0) * 4;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(wq[d]) : "v"(lwk), "s"(xw));
      }
    };
    auto arrive_x = [&](int d) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS * (D - 1)) : "memory");
#pragma unroll
      for (int m = 0; m < PM; m++) asm volatile("" : "+v"(a[d][m]));  // pins the MFMAs behind the wait
#pragma unroll
      for (int n = 0; n < PN; n++) asm volatile("" : "+v"(bq[d][n]));
      if (WEIGHTED) asm volatile("" : "+v"(wq[d]));
    };
    if constexpr (EXACT == 4 && D == 2) {
      // In step, every wavefront of the workgroup is in the same phase: were the refill issued behind the MFMAs (as
      // below), the texture path would sit idle while the MFMAs run and the MFMA pipe while the ~90 fragment loads of
      // a k-group are processed ([measured] cfg 2: a k-group took 2400 cycles for 1470 of MFMAs, two k-groups in flight
      // or one).  So the NEXT k-group's fetch is issued first -- into the other register set, free since the last
      // trip -- and is processed under this k-group's MFMAs; one fetch in flight, waited for in full.
      fetch_x(0, kidx(kg0));
      for (uint64_t kg = kg0; kg < kg1; kg += 2) {
#pragma unroll
        for (int d = 0; d < 2; d++) {
#if !defined(NGD_DIAG_NOBARRIER)  // (NGD_DIAG_*: timing-only builds of tools/exact_diag.sh, results meaningless)
          asm volatile("s_barrier" ::: "memory");
#endif
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int m = 0; m < PM; m++) asm volatile("" : "+v"(a[d][m]));  // pins the MFMAs behind the wait
#pragma unroll
          for (int n = 0; n < PN; n++) asm volatile("" : "+v"(bq[d][n]));
          if (WEIGHTED) asm volatile("" : "+v"(wq[d]));
#if !defined(NGD_DIAG_NOLOAD)
          fetch_x(d ^ 1, kidx(kg + d + 1));  // (past the slice: the images' tail, never consumed)
#endif
          __builtin_amdgcn_sched_barrier(0);  // the fetch stays AHEAD of the MFMAs
#if defined(NGD_DIAG_NOMFMA)
          if (false) {
#else
          if (kg + d < kg1) {
#endif
            if (WEIGHTED) {
              const double w = wq[d];
#pragma unroll
              for (int m = 0; m < PM; m++) a[d][m] *= w;
            }
#pragma unroll
            for (int m = 0; m < PM; m++)
#pragma unroll
              for (int n = 0; n < PN; n++)
                if (!TRI || m <= n)
                  acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d][m], bq[d][n], acc[m][n], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
    uint64_t nxt[D];
#pragma unroll
    for (int d = 0; d < D; d++) fetch_x(d, kidx(kg0 + d));
#pragma unroll
    for (int d = 0; d < D; d++) nxt[d] = kidx(kg0 + d + D);
    for (uint64_t kg = kg0; kg < kg1; kg += D) {
#pragma unroll
      for (int d = 0; d < D; d++) {
        // every job of the slice is at this k-group (loads stay in flight); a ring trip's barriers are all executed,
        // also past the slice's last k-group, so that every wavefront of the workgroup meets the same number
        if (SYNC) asm volatile("s_barrier" ::: "memory");
        arrive_x(d);
        if (D == 1 || kg + d < kg1) {
          if (WEIGHTED) {
            const double w = wq[d];
#pragma unroll
            for (int m = 0; m < PM; m++) a[d][m] *= w;
          }
#pragma unroll
          for (int m = 0; m < PM; m++)
#pragma unroll
            for (int n = 0; n < PN; n++)
              if (!TRI || m <= n)
                acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d][m], bq[d][n], acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch_x(d, nxt[d]);
        __builtin_amdgcn_sched_barrier(0);
        nxt[d] = kidx(kg + d + 2 * D);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // EXACT = 5: the same blocks, operands through LDS (see the note at the kernel).  One barrier per k-group: behind it the
  // k-group's fragments are complete in stage[b] (written in the previous trip) and nobody reads stage[b ^ 1] any more,
  // which this trip fills with the NEXT k-group (fetched during the previous trip) before the fetch after next goes out.
  auto run_lds = [&](auto rows_c, auto cols_c, auto tri_c) {
    constexpr int PM = decltype(rows_c)::value, PN = decltype(cols_c)::value;
    constexpr bool TRI = decltype(tri_c)::value;
    if constexpr (EXACT == 5) {
      // [buffer][row groups 0..15 | column groups 16..31][lane]: dynamic LDS (one array for every block shape's code path)
      extern __shared__ double ngd_stage_lds[];
      double(*stage)[32][64] = (double(*)[32][64])ngd_stage_lds;
      typedef __attribute__((address_space(3))) const volatile double lds_cvd;
      const uint32_t nw = blockDim.x >> 6, n_igv = n_igv_touch, nf = 2 * n_igv;
      // This wavefront's share of a k-group's fragments: f = wave, wave + nw, ... -- DEPTH of them in every wavefront (the
      // launcher's ceil(nf / nw); an index past the last fragment repeats it: same bytes to the same place).  No branch
      // inside the k loop: the hand-issued loads are invisible to the compiler, and a register copy it placed at a join
      // would read a register the load has not written yet.
      constexpr int CNT = DEPTH;
      // PF k-groups of fetches in flight per wavefront (register sets of CNT doubles): with one, the workgroups hold
      // ~5 MB in flight device-wide and HBM delivers 3.5 TB/s; its latency under load wants ~12 MB for 5.5
#if !defined(NGD_LDS_PF)
#define NGD_LDS_PF 4
#endif
      constexpr int PF = NGD_LDS_PF, U = PF < 2 ? 2 : PF;  // the k loop is unrolled U times: buffer and set by position
      static_assert((PF == 1 || PF == 2 || PF == 4) && PF + U <= NGD_KG_TAIL, "the fetch runs PF + U k-groups past the slice at most");
      double st[PF][CNT];
      const double *src[CNT];
      uint32_t slot[CNT];
#pragma unroll
      for (int q = 0; q < CNT; q++) {
        uint32_t f = wave + q * nw;
        f = f < nf ? f : nf - 1;
        src[q] = f < n_igv ? PA + (uint64_t)f * 64 : QB + (uint64_t)(f - n_igv) * 64;
        slot[q] = f < n_igv ? f : 16 + (f - n_igv);
      }
      auto gload = [&](int set, uint64_t kg) {
#pragma unroll
        for (int q = 0; q < CNT; q++) load_frag<0>(st[set][q], lane_off, src[q] + kg * kstride);
      };
      auto lwrite = [&](uint32_t b, int set) {
#pragma unroll
        for (int q = 0; q < CNT; q++) {
          asm volatile("" : "+v"(st[set][q]));  // read only behind the wait
          stage[b][slot[q]][lane] = st[set][q];
        }
      };
      gload(0, kg0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lwrite(0, 0);
#pragma unroll
      for (int i = 0; i < PF; i++) gload(i, kg0 + 1 + i);  // set i holds k-group kg0 + 1 + i, then every PF-th after it
      for (uint64_t kg = kg0; kg < kg1; kg += U) {
#pragma unroll
        for (int t = 0; t < U; t++) {  // (every trip runs its barrier, LDS traffic and fetch; only the MFMAs end at kg1)
          const uint32_t b = t & 1;
          const int set = t % PF;
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the fetches stay in flight)
          double xa[PM], xb[PN];
#pragma unroll
          for (int m = 0; m < PM; m++) xa[m] = *(lds_cvd *)&stage[b][ig0 + m][lane];
#pragma unroll
          for (int n = 0; n < PN; n++) xb[n] = *(lds_cvd *)&stage[b][16 + jg0 + n][lane];
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT * (PF - 1)) : "memory");  // the oldest set, k-group kg + t + 1, is here
          lwrite(b ^ 1, set);
          __builtin_amdgcn_sched_barrier(0);
          gload(set, kg + t + 1 + PF);  // (past the slice: the images' tail, never consumed)
          __builtin_amdgcn_sched_barrier(0);
          if (kg + t < kg1) {
#pragma unroll
            for (int m = 0; m < PM; m++)
#pragma unroll
              for (int n = 0; n < PN; n++)
                if (!TRI || m <= n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[m], xb[n], acc[m][n], 0, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using std::integral_constant;
  typedef integral_constant<int, 1> I1;
  typedef integral_constant<int, 2> I2;
  typedef integral_constant<int, 3> I3;
  typedef integral_constant<int, 4> I4;

  // A block shape this form has no code path for (engine.hip builds none: ngd_create checks its job list against
  // ngd_mfma_shape_listed()): the block's sums leave as NaN and clk[2] tells the host, which fails the run with NGD_E_HIP
  // -- never a silent matrix of zeros.
  bool unlisted = false;
  if (kg0 < kg1) {
    if constexpr (EXACT == 0) {
      // tri: a block on the diagonal whose lower triangle is left out (ngd_config.exact_shapes = 7: 10 tiles of 16;
      // engine.hip packs such blocks into workgroups of their own, so that a workgroup's four jobs advance at one rate).
      // NOT the default: such a block runs through its slice 1.6 times as fast as the others, away from the k-groups
      // they hold in L2 -- [measured] cfg 5: the fabric delivers 20.1 GB a launch instead of 7.1 (L2 hit rate 0.42 for
      // 0.76), the clock falls 2.5 % and the pass is no shorter for its 8 % fewer MFMAs; holding the triangular blocks
      // back (a progress word per slice, naps) made the FULL blocks drift instead: 28 GB, and 217 GB for 39 at cfg 3
      // (profiles/r05_tri_diag_ab.txt).  Equal work per k-group is what keeps a slice's jobs together.
      if (!job.tri) run(I4{}, I4{}, std::false_type{}, std::false_type{});
      else if (DEPTH == 1 && PA == QB && ig0 == jg0) run(I4{}, I4{}, std::true_type{}, integral_constant<bool, DEPTH == 1>{});
      else run(I4{}, I4{}, std::true_type{}, std::false_type{});
    } else if constexpr (EXACT == 1 || EXACT == 3) {
      switch (shape) {  // rows | cols << 3 | tri << 6
        case 4 | 4 << 3:
          if constexpr (SYNC) run_exact(I4{}, I4{}, std::false_type{}, I1{}); else run(I4{}, I4{}, std::false_type{}, std::false_type{});
          break;
        case 4 | 3 << 3: run_exact(I4{}, I3{}, std::false_type{}, I1{}); break;
        case 4 | 2 << 3: run_exact(I4{}, I2{}, std::false_type{}, I1{}); break;
        case 4 | 1 << 3: run_exact(I4{}, I1{}, std::false_type{}, I1{}); break;
        case 4 | 4 << 3 | 1 << 6: run_exact(I4{}, I4{}, std::true_type{}, I1{}); break;
        case 3 | 3 << 3 | 1 << 6: run_exact(I3{}, I3{}, std::true_type{}, I1{}); break;
        case 2 | 2 << 3 | 1 << 6: run_exact(I2{}, I2{}, std::true_type{}, I1{}); break;
        case 1 | 1 << 3 | 1 << 6: run_exact(I1{}, I1{}, std::true_type{}, I1{}); break;
        default:  // any other shape: the full pattern is always right
          if constexpr (SYNC) run_exact(I4{}, I4{}, std::false_type{}, I1{}); else run(I4{}, I4{}, std::false_type{}, std::false_type{});
      }
    } else if constexpr (EXACT == 5) {
      switch (shape) {
        case 2 | 4 << 3: run_lds(I2{}, I4{}, std::false_type{}); break;
        case 2 | 3 << 3: run_lds(I2{}, I3{}, std::false_type{}); break;
        case 2 | 2 << 3: run_lds(I2{}, I2{}, std::false_type{}); break;
        case 2 | 1 << 3: run_lds(I2{}, I1{}, std::false_type{}); break;
        case 2 | 4 << 3 | 1 << 6: run_lds(I2{}, I4{}, std::true_type{}); break;
        case 2 | 3 << 3 | 1 << 6: run_lds(I2{}, I3{}, std::true_type{}); break;
        case 2 | 2 << 3 | 1 << 6: run_lds(I2{}, I2{}, std::true_type{}); break;
        case 1 | 1 << 3 | 1 << 6: run_lds(I1{}, I1{}, std::true_type{}); break;
        default: unlisted = true;
      }
    } else {  // EXACT 2 / 4: blocks of at most 2 x 4 tiles (engine.hip builds no other shapes for them), DEPTH k-groups in flight
      typedef integral_constant<int, DEPTH> ID;
      switch (shape) {
        case 2 | 4 << 3: run_exact(I2{}, I4{}, std::false_type{}, ID{}); break;
        case 2 | 3 << 3: run_exact(I2{}, I3{}, std::false_type{}, ID{}); break;
        case 2 | 2 << 3: run_exact(I2{}, I2{}, std::false_type{}, ID{}); break;
        case 2 | 1 << 3: run_exact(I2{}, I1{}, std::false_type{}, ID{}); break;
        case 2 | 4 << 3 | 1 << 6: run_exact(I2{}, I4{}, std::true_type{}, ID{}); break;
        case 2 | 3 << 3 | 1 << 6: run_exact(I2{}, I3{}, std::true_type{}, ID{}); break;
        case 2 | 2 << 3 | 1 << 6: run_exact(I2{}, I2{}, std::true_type{}, ID{}); break;
        case 1 | 1 << 3 | 1 << 6: run_exact(I1{}, I1{}, std::true_type{}, ID{}); break;
        default: unlisted = true;
      }
    }
  }

  if (unlisted) {  // (uniform)
#pragma unroll
    for (int m = 0; m < WM; m++)
#pragma unroll
      for (int n = 0; n < WN; n++) acc[m][n] = (ngd_d4){__builtin_nan(""), __builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
    if (clk && lane == 0) clk[2] = 1ull;
  }
  if (clk_wave) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      clk[0] = t1 - clk_t0;
      clk[1] = r1 - clk_r0;
    }
  }
  // D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*r
  double *out = slab + (uint64_t)ks * n_pad * n_pad;
#pragma unroll
  for (int m = 0; m < WM; m++)
#pragma unroll
    for (int n = 0; n < WN; n++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (EXACT && (m >= (int)(shape & 7) || n >= (int)((shape >> 3) & 7) || ((shape >> 6) && m > n))) continue;  // (uniform)
        if (!EXACT && (shape >> 6) && m > n) continue;  // (form 6: a diagonal block's lower triangle was not computed)
        const uint32_t i = (ig0 + m) * 16 + (lane >> 4) + 4 * r;
        const uint32_t j = (jg0 + n) * 16 + (lane & 15);
        // (resume: single-image engines walk a pass in ranges of k-groups, every slice a piece of each range -- the
        // block's sums over this range are added to what the launches over the earlier ranges left in its plane)
        double v = acc[m][n][r];
        if (resume) v = out[(uint64_t)i * n_pad + j] + v;
        out[(uint64_t)i * n_pad + j] = v;
      }
}

}  // namespace

void ngd_launch_accum_mfma(hipStream_t st, const ngd_geom &g, const double *PA, const double *QB,
                           const double *d_ws /* wk */, const uint32_t *d_kgl, const ngd_job *d_jobs, uint32_t n_wg,
                           int exact_shapes /* 3: n_wg = 1 workgroup of wg_waves wavefronts per slice */, uint32_t wg_waves, uint32_t n_ks, uint64_t kg_per_slice, uint64_t n_kg_eff,
                           uint64_t k_per_slice, uint32_t w_slice_stride, double *slab, unsigned long long *d_clk,
                           uint32_t ks0, uint32_t resume) {
  if (!n_wg) return;
  // EXACT = 3: a prefetching wavefront beside the jobs where a twelfth fits and the slices are plain k-group ranges
  uint32_t touch_igv = 0;
  const uint32_t n_igv = (uint32_t)((g.n_ind + 15) / 16);
  // EXACT = 5 (operands through LDS): plain k-group ranges, at most 16 groups a side and 4 fragments per wavefront;
  // weighted passes and masked slices of the same engine take the register form (EXACT = 4) of the same jobs
  if (exact_shapes == 5 && (d_ws || d_kgl || k_per_slice || n_igv > 16 || 2 * n_igv > 4 * wg_waves)) exact_shapes = 4;
  if (exact_shapes == 5) {
    touch_igv = n_igv;
  } else if (((exact_shapes == 3 && wg_waves <= 11) || (exact_shapes == 4 && wg_waves <= 15)) && !d_ws && !d_kgl && !k_per_slice &&
             n_igv <= 16) {  // (the prefetching wavefront keeps 24 + 2 n_igv loads in flight: the counter holds 63)
    touch_igv = n_igv;
    wg_waves += 1;
  }
  // n_ks is a multiple of 8 (see the deal in the kernel)
  // EXACT: one job per (single-wavefront) workgroup -- jobs of different shapes last differently, and a
  // wavefront that is done should not wait for three siblings before its slot is handed on
#define NGD_MFMA(W, D, P, X)                                                                                    \
  hipLaunchKernelGGL((k_accum_mfma<W, D, P, X>), dim3(n_wg * n_ks), dim3(X >= 3 ? 64 * wg_waves : X ? 64 : 256),       \
                     X == 5 ? 2 * 32 * 64 * sizeof(double) : 0, st, PA, QB, d_ws, d_kgl, d_jobs,                        \
                     n_wg, g.n_ig, g.n_pad, kg_per_slice, n_kg_eff, k_per_slice, w_slice_stride, slab, touch_igv, d_clk, ks0, resume)
  // No in-wave run-ahead (DEPTH 1), 3 wavefronts per SIMD: the third wavefront covers the others' load phases.
  // Measured against a 4-deep register ring at 2 wavefronts per SIMD (56.0 vs 51.0 ms on the same job layout) and
  // against LDS-staged operand panels (tools/experiments/accum_mfma_lds.hip; profiles/r01_cfg3_mfma_*): both lose.
  // (a 2-deep ring at 3 wavefronts per SIMD needs 168+ VGPRs and spills: not built)
  if (exact_shapes == 2) {  // blocks of at most 2 x 4 tiles: X2D k-groups of operands in flight per wavefront
#if !defined(NGD_EXACT2_DEPTH)
#define NGD_EXACT2_DEPTH 1
#endif
    constexpr int X2D = NGD_EXACT2_DEPTH;
    constexpr int X2W = X2D == 1 ? 6 : X2D == 2 ? 5 : 4;  // wavefronts per SIMD the registers allow
    if (d_ws) NGD_MFMA(true, X2D, (X2W > 5 ? 5 : X2W), 2); else NGD_MFMA(false, X2D, X2W, 2);  // (weighted: 84 registers at depth 1)
  } else if (exact_shapes == 5) {  // (DEPTH = fragments a wavefront fetches per k-group)
    switch ((2 * n_igv + wg_waves - 1) / wg_waves) {
      case 1: NGD_MFMA(false, 1, 4, 5); break;
      case 2: NGD_MFMA(false, 2, 4, 5); break;
      case 3: NGD_MFMA(false, 3, 4, 5); break;
      default: NGD_MFMA(false, 4, 4, 5); break;
    }
  } else if (exact_shapes == 4) {
#if !defined(NGD_EXACT4_DEPTH)
#define NGD_EXACT4_DEPTH 2  // two register sets: the next k-group's fetch is issued ahead of this one's MFMAs
#endif
    if (d_ws) NGD_MFMA(true, NGD_EXACT4_DEPTH, 4, 4); else NGD_MFMA(false, NGD_EXACT4_DEPTH, 4, 4);
  } else if (exact_shapes == 3) {
    if (d_ws) NGD_MFMA(true, 1, 3, 3); else NGD_MFMA(false, 1, 3, 3);
  } else if (exact_shapes) {
    if (d_ws) NGD_MFMA(true, 1, 3, 1); else NGD_MFMA(false, 1, 3, 1);
  } else {
    if (d_ws) NGD_MFMA(true, 1, 3, 0); else NGD_MFMA(false, 1, 3, 0);
  }
#undef NGD_MFMA
}
