// accum_em_table.hip -- gen_dist() without --indep_geno (reference ngsDist.cpp:340-353 around em2(),
// emOptim2.cpp:112-135), tiled so that everything that depends on ONE individual is computed once per
// tile and site instead of once per pair.
//
// For one site the EM iterate has a closed form (accum_em.hip): with a_k = GL1[x]*GL2[y],
//   sfs_t = a^t / S_t,  S_t = SUM_k a_k^t = A_t(i1) * A_t(i2),  A_t(i) = SUM_x GL_i[x]^t   (rank one),
//   lik_t = log(S_{t+1} / S_t),
// and the stopping rule |lik_t - lik_{t-1}| < tole (emOptim2.cpp:127) reads
//   R_t(i1) * R_t(i2) < e^tole,        R_t(i) = A_{t+1}(i) A_{t-1}(i) / A_t(i)^2   (>= 1),
// the returned sfs being sfs_T for the first such t = T (or T = maxIter = 50), so that the site's score-weighted
// sum (ngsDist.cpp:351-353) is
//   c = f_T(i1)' * score * f_T(i2),    f_t(i) = GL_i^t / A_t(i).
// R_t and f_t belong to ONE individual.  A workgroup owns a 64 x 64 tile of pairs and a slice of sites; per
// site it builds the tables of the tile's 64 + 64 individuals in LDS, CH EM steps at a time:
//   rows:    Q_t = e^tole / R_t  and f_t;        columns: R_t and g_t = score * f_t,
// then every wavefront (lane = column, 64 / NW rows each) finds T of its pairs by comparing R_t(column) with
// Q_t(row) -- one FP64 compare + one select per step and pair -- and adds f_T(row) . g_T(column).  Pairs
// that have not stopped within the CH steps wait for the next CH (a second round is the rule, a third
// rare; step 50 is forced).  Against the per-pair form (k_accum_em<fast>: ~16 FP64 instructions per
// step and pair, 46 per pair-site at the end) this is 2 per step and ~10 per round at the end.
//
// The default form (PACK, below) changes how the rounds AFTER a site's first are worked -- the pairs still searching
// then sit in a few columns and rows, so they are packed 8 columns x a wavefront's 8 rows to a unit instead of being
// scanned row by row with most lanes idle -- skips the 8-step blocks in which a row cannot stop at all, and orders a
// row's LDS reads so that it waits for one round trip; sums are bit-identical to the plain form's (shape 4).
//
// Same arithmetic contract as the fast form: the stopping step is the reference's except where the
// criterion is within rounding of the tolerance; sums agree to ~1e-13 relative (bar: 1e-9).
// Missing sites under --pairwise_del (ngsDist.cpp:335-338) take Q = +inf / R = 0 and f = g = 0 at
// every step: the pair stops at step 1 and adds exactly 0.
#include "ngd_internal.h"

namespace {

constexpr int TS = 64;        // tile edge (individuals)
constexpr int MAX_ITER = 50;  // ngsDist.cpp:349

typedef double ngd_d2 __attribute__((ext_vector_type(2)));

// LDS reads are written so that each becomes ONE ds_read_b64 / ds_read_b128: hipcc otherwise pairs 8-byte reads
// into ds_read2_b64 / ds_read2st64_b64, which take 8 LDS cycles for the 16 bytes per lane that ds_read_b128 moves
// in 4 and two ds_read_b64 in 2 + 2 (MI355X_MICROARCH.md, LDS table) -- and this kernel is bound by those cycles.
// (volatile: the pairing pass leaves such reads alone; the address space is spelled out because a volatile access
// through a generic pointer is not narrowed to LDS by the compiler)
typedef __attribute__((address_space(3))) const volatile double ngd_lds_cvd;
__device__ __forceinline__ double lds_b64(const double *p) { return *(ngd_lds_cvd *)p; }

template <int CH, bool PACK = false>
struct alignas(16) em_tables {
  static constexpr int RS = CH + 2;  // row-table stride in doubles: spreads the builders' stores over the banks, 16-B aligned
  double Qr[TS * RS];                // rows:    e^tole / R_t            [row][step]
  double Fr[3][TS * RS];             // rows:    f_t[x]                  [x][row][step]
  double Rc[CH * TS];                // columns: R_t                     [step][column]
  double Gc[3][CH * TS];             // columns: (score * f_t)[x]        [x][step][column]
  uint32_t more[2];                  // "some pair of the tile has not stopped yet", one word per round parity
  // PACK (later rounds worked in packed units, below): per wavefront, the contributions of one unit on their way from
  // the lanes that computed them to the lanes that own the pairs, and the unit's column list
  double unit_c[PACK ? 8 * 64 : 2];
  // bit i of word p: row i of the tile cannot stop at any of the steps the p-th row builder wrote this round
  // (its own factor R_t is still >= e^tole there, and a column's is never below 1): the plain scan skips those steps
  unsigned long long row_nostop[4];
  uint8_t unit_col[PACK ? 8 * 8 : 8];
  uint8_t poison[2][TS];  // RB > 1, kernel end: bit b = matrix b drew a site at which this row / column was all zero
};

// Workgroup barrier that waits for this wavefront's LDS traffic only.  __syncthreads() also drains vmcnt, i.e. the
// global loads of the NEXT site's likelihoods that are meant to stay in flight across the rounds of this one
// ([measured] 1000 x 2e4: 51.5 ms with __syncthreads(), the loads' latency exposed once per site and workgroup).
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Four steps of the search for a pair's stopping step T, the lanes that have stopped dropping out of EXEC:
//   for each step: EXEC &= !(R < Q);  n += 1 in the lanes still active
// so n counts the steps a lane has NOT stopped at (two VALU instructions per step: the compare writes EXEC, nothing
// reads a compare result as a select mask, which would cost a third instruction and two wait states per step).
// m: in = lanes still searching, out = those that have not stopped within these steps either.  EXEC is restored.
__device__ __forceinline__ void scan4(uint64_t &m, uint32_t &n, double r0, double r1, double r2, double r3, double q0,
                                      double q1, double q2, double q3) {
  uint64_t keep;
  asm volatile(
      "s_mov_b64 %[keep], exec\n\t"
      "s_and_b64 exec, exec, %[m]\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r0], %[q0]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r1], %[q1]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r2], %[q2]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r3], %[q3]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "s_mov_b64 %[m], exec\n\t"
      "s_mov_b64 exec, %[keep]"
      : [m] "+s"(m), [n] "+v"(n), [keep] "=&s"(keep)
      : [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [r3] "v"(r3), [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [q3] "v"(q3)
      : "vcc", "scc");  // v_cmpx writes VCC next to EXEC on gfx9; s_and_b64 writes SCC
}

// Eight steps in one block: EXEC is saved and restored once, and the second four steps are jumped over when every lane
// has stopped -- per step the same two VALU instructions as scan4, but 5 scalar instructions per 8 steps instead of 14
// (scalar instructions and taken branches cost a wavefront issue slots like vector ones).
__device__ __forceinline__ void scan8(uint64_t &m, uint32_t &n, double r0, double r1, double r2, double r3, double r4,
                                      double r5, double r6, double r7, double q0, double q1, double q2, double q3,
                                      double q4, double q5, double q6, double q7) {
  uint64_t keep;
  asm volatile(
      "s_mov_b64 %[keep], exec\n\t"
      "s_and_b64 exec, exec, %[m]\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r0], %[q0]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r1], %[q1]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r2], %[q2]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r3], %[q3]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "s_cbranch_execz .Lngd_scan8_%=\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r4], %[q4]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r5], %[q5]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r6], %[q6]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "v_cmpx_nlt_f64_e32 vcc, %[r7], %[q7]\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n"
      ".Lngd_scan8_%=:\n\t"
      "s_mov_b64 %[m], exec\n\t"
      "s_mov_b64 exec, %[keep]"
      : [m] "+s"(m), [n] "+v"(n), [keep] "=&s"(keep)
      : [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [r3] "v"(r3), [r4] "v"(r4), [r5] "v"(r5), [r6] "v"(r6), [r7] "v"(r7),
        [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [q3] "v"(q3), [q4] "v"(q4), [q5] "v"(q5), [q6] "v"(q6), [q7] "v"(q7)
      : "vcc", "scc");
}

__device__ __forceinline__ double rcp_nr(double a) {
  double y = __builtin_amdgcn_rcp(a);
  double e = __builtin_fma(-a, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-a, y, 1.0);
  return __builtin_fma(y, e, y);
}

// One wavefront's share of a table round: steps tfirst+1 .. tfirst+SEG of its 64 individuals (lane = individual),
// v = GL^tfirst.  One role per call, so that the whole share is one block of straight-line code: the SEG+2 power
// sums and their reciprocals are independent of each other and the scheduler can keep the FP64 pipe full.
template <int CH, int SEG, bool ROW, bool PACK>
__device__ __forceinline__ void build_round(em_tables<CH, PACK> &L, const double *v, const double *g, const ngd_score &sc,
                                            uint32_t lane, uint32_t seg, int tfirst, bool miss) {
  constexpr int RS = em_tables<CH, PACK>::RS;
  const double E = 0x1.0041919b7ee34p+0;  // exp(0.001), the tolerance of ngsDist.cpp:349
  // pw = GL^(tfirst+k), A[k] its sum, r[k] = 1/A[k]; k = 1 .. SEG are the steps written.  Step k needs A and r of
  // k-1, k, k+1; f_k leaves as soon as r[k] is known and the powers are not kept, so that a step's live range is
  // short ([measured] 1000 x 2e4: 50.1-50.3 ms against 51.5-51.8 with all SEG+2 powers and reciprocals formed first --
  // that form wanted 100+ VGPRs around the build and left the allocator no slack anywhere else).
  double pw[3] = {v[0], v[1], v[2]};
  double A[SEG + 2], r[SEG + 2];
  bool nostop = true;
  A[0] = (pw[0] + pw[1]) + pw[2];
  r[0] = rcp_nr(A[0]);
#pragma unroll
  for (int k = 1; k <= SEG + 1; k++) {
#pragma unroll
    for (int x = 0; x < 3; x++) pw[x] *= g[x];
    A[k] = (pw[0] + pw[1]) + pw[2];
    r[k] = rcp_nr(A[k]);
    if (k <= SEG) {  // f_k (rows) / g_k = score * f_k (columns)
      const uint32_t tt = seg * SEG + k - 1;
      double f[3];
#pragma unroll
      for (int x = 0; x < 3; x++) f[x] = miss ? 0.0 : pw[x] * r[k];
      if (ROW) {
#pragma unroll
        for (int x = 0; x < 3; x++) L.Fr[x][lane * RS + tt] = f[x];
      } else {
#pragma unroll
        for (int x = 0; x < 3; x++)
          L.Gc[x][tt * TS + lane] =
              __builtin_fma(sc.v[3 * x + 2], f[2], __builtin_fma(sc.v[3 * x + 1], f[1], sc.v[3 * x] * f[0]));
      }
    }
    if (k >= 2) {  // the stopping-rule factor of step k-1
      const int c = k - 1;
      const uint32_t tt = seg * SEG + c - 1;
      const bool force = miss || tfirst + c >= MAX_ITER;
      if (ROW) {
        double q = (E * (A[c] * A[c])) * (r[c + 1] * r[c - 1]);
        if (force) q = __builtin_inf();
        L.Qr[lane * RS + tt] = q;
        // a pair stops where R(column) < q, and the R the column builders write is 1 or more up to a few units in
        // the last place: below this q no column can stop
        if (PACK) nostop = nostop && q <= 1.0 - 0x1p-40;
      } else {
        double rr = (A[c + 1] * A[c - 1]) * (r[c] * r[c]);
        if (force) rr = 0.0;
        L.Rc[tt * TS + lane] = rr;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (PACK && ROW) {
    const unsigned long long none = __builtin_amdgcn_ballot_w64(nostop);
    if (lane == 0) L.row_nostop[seg] = none;
  }
}

typedef __attribute__((address_space(3))) volatile double ngd_lds_vd;
typedef __attribute__((address_space(3))) volatile uint8_t ngd_lds_vb;
typedef __attribute__((address_space(3))) const volatile ngd_d2 ngd_lds_cvd2;

// One row of the tile the plain way (lane = column): the row's CH thresholds are the same 16 bytes in every lane (one
// ds_read_b128 per two steps), the column's R_t sit in registers (R2).  Pairs of `todo` bit r that stop within the
// round add their term to accr and leave todo.
// skip: bit b = no pair of this row can stop in steps 8b+1 .. 8b+8 (row_nostop): that block is not searched.
// The LDS reads of a row are issued so that a wavefront waits for ONE round trip per row, not three: the thresholds of
// the second block go out before the first block is searched; the next row's first-block thresholds (rb_next, `pref`)
// go out behind this row's six table reads for its terms, so the wait for those leaves them in flight.  QA holds the
// first-block thresholds: put there by the previous row's call, or read here if `load`.  (Every read is volatile: issue
// order = program order.)
// RB matrices at once (the full data set and bootstrap replicates, or replicates only): the pair's term is added to
// accr[b] with the site's weight wv[b] in matrix b -- product first, then the sum, like the one-matrix kernel, so a
// matrix carries the same bits from either.
template <int CH, bool WEIGHTED, bool PACK, int RB>
__device__ __forceinline__ void scan_row(const em_tables<CH, PACK> &L, uint32_t rb /* row * RS */, uint32_t rb_next,
                                         uint32_t lane, int r, const double (&R2)[CH], ngd_d2 (&QA)[4], uint32_t &todo,
                                         double (&accr)[RB], const double (&wv)[RB], uint32_t skip, bool load,
                                         bool pref) {
  static_assert(CH == 16, "two blocks of eight steps");
  const bool mine = (todo >> r) & 1;
  uint64_t m = __builtin_amdgcn_ballot_w64(mine);
  ngd_d2 QB[4];
  if (load) {
#pragma unroll
    for (int h = 0; h < 4; h++) QA[h] = *(ngd_lds_cvd2 *)&L.Qr[rb + 2 * h];
  }
  // (also when the second block will be skipped: such a row is nearly always out of the round altogether, and an
  // unconditional read keeps a branch and eight register moves out of every row)
#pragma unroll
  for (int h = 0; h < 4; h++) QB[h] = *(ngd_lds_cvd2 *)&L.Qr[rb + 8 + 2 * h];
  uint32_t n = 8;  // steps survived (by the lanes in m; the others keep what they had when they stopped)
#if defined(NGD_EMT_DIAG_COARSE)
  // TIMING-ONLY build (results wrong): what the search would cost if only every other step were compared (the upper
  // bound of a coarse search + one refining compare: DESIGN.md section 8)
  n = 0;
  if ((skip & 3) != 3)
    scan8(m, n, R2[1], R2[3], R2[5], R2[7], R2[9], R2[11], R2[13], R2[15], QA[0][1], QA[1][1], QA[2][1], QA[3][1], QB[0][1],
          QB[1][1], QB[2][1], QB[3][1]);
  else n = 8;
  const uint32_t lim = 8;
  const int T = mine && n < lim ? 2 * (int)n + 2 : 0;
#else
  if (!(skip & 1)) {
    n = 0;
    scan8(m, n, R2[0], R2[1], R2[2], R2[3], R2[4], R2[5], R2[6], R2[7], QA[0][0], QA[0][1], QA[1][0], QA[1][1], QA[2][0],
          QA[2][1], QA[3][0], QA[3][1]);
  }
  // second block skipped: the lanes that are still searching (n = 8) stay so
  const uint32_t lim = (skip & 2) ? 8 : CH;
  if (m != 0 && !(skip & 2))
    scan8(m, n, R2[8], R2[9], R2[10], R2[11], R2[12], R2[13], R2[14], R2[15], QB[0][0], QB[0][1], QB[1][0], QB[1][1],
          QB[2][0], QB[2][1], QB[3][0], QB[3][1]);
  const int T = mine && n < lim ? (int)n + 1 : 0;
#endif
  const uint32_t ti = T ? T - 1 : 0;
  const uint32_t a = rb + ti, b = ti * TS + lane;
  const double f0 = lds_b64(&L.Fr[0][a]), g0 = lds_b64(&L.Gc[0][b]);
  const double f1 = lds_b64(&L.Fr[1][a]), g1 = lds_b64(&L.Gc[1][b]);
  const double f2 = lds_b64(&L.Fr[2][a]), g2 = lds_b64(&L.Gc[2][b]);
  if (pref) {
#pragma unroll
    for (int h = 0; h < 4; h++) QA[h] = *(ngd_lds_cvd2 *)&L.Qr[rb_next + 2 * h];
  }
  double c = f0 * g0;
  c = __builtin_fma(f1, g1, c);
  c = __builtin_fma(f2, g2, c);
  if (T) {  // (RB > 1: the term is finite -- the individuals that would make it NaN are handled apart, see `poison`)
#pragma unroll
    for (int b = 0; b < RB; b++) accr[b] = accr[b] + (WEIGHTED ? c * wv[b] : c);
    todo &= ~(1u << r);
  }
}

// Later rounds, what the plain rows leave: the pairs still searching sit in a few COLUMNS (the individuals whose own
// factor R_t has not come down yet: every row of such a column is still searching) plus a few stragglers, so a plain
// row scan would run with ~15 of its 64 lanes.  A packed unit takes 8 of those columns x this wavefront's 8 rows:
// lane = (column slot, row); both the row's thresholds and the column's R_t come from the tables by per-lane
// address; the search and the term are the plain row's, instruction for instruction (same bits).  The terms then travel
// inside the wavefront -- through its 512 bytes of unit_c -- to the lanes that own the pairs (lane = column), which
// add them in the same order as ever: one term per pair and site, 0.0 from the slots that had nothing to add.
template <int CH, bool WEIGHTED, int RB>
__device__ __forceinline__ void packed_units(em_tables<CH, true> &L, uint32_t wave, uint32_t lane, uint32_t &todo,
                                             double (&acc)[8][RB], const double (&wv)[RB]) {
  constexpr int RS = em_tables<CH, true>::RS;
  static_assert(CH % 8 == 0, "shape");
  ngd_lds_vd *uc = (ngd_lds_vd *)&L.unit_c[wave * 64];
  ngd_lds_vb *ul = (ngd_lds_vb *)&L.unit_col[wave * 8];
  const uint32_t slot = lane >> 3, r = lane & 7;
  const uint32_t rb = (wave * 8 + r) * RS;
  uint64_t pm = __builtin_amdgcn_ballot_w64(todo != 0);  // columns with a pair still searching
  while (pm) {
    // this unit's columns: the first 8 set bits of pm, slot = rank among them
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0));
    const bool sel = ((pm >> lane) & 1) && rank < 8;
    if (sel) ul[rank] = (uint8_t)lane;
    const uint32_t n_col = (uint32_t)__builtin_popcountll(pm);
    const bool valid = slot < n_col;  // (a wavefront's LDS operations execute in order: the list is there)
    const uint32_t col = valid ? (uint32_t)ul[slot] : 0;
    const uint32_t tb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(col << 2), (int)todo);
    const bool mine = valid && ((tb >> r) & 1);
    uint64_t m = __builtin_amdgcn_ballot_w64(mine);
    uint32_t n = 0;
#pragma unroll
    for (int b = 0; b < CH / 8; b++) {
      if (b && m == 0) break;
      double R[8];
      ngd_d2 Q[4];
#pragma unroll
      for (int k = 0; k < 8; k++) R[k] = lds_b64(&L.Rc[(8 * b + k) * TS + col]);
#pragma unroll
      for (int h = 0; h < 4; h++) Q[h] = *(ngd_lds_cvd2 *)&L.Qr[rb + 8 * b + 2 * h];
      scan8(m, n, R[0], R[1], R[2], R[3], R[4], R[5], R[6], R[7], Q[0][0], Q[0][1], Q[1][0], Q[1][1], Q[2][0], Q[2][1],
            Q[3][0], Q[3][1]);
    }
    const int T = mine && n < (uint32_t)CH ? (int)n + 1 : 0;
    const uint32_t ti = T ? T - 1 : 0;
    const uint32_t a = rb + ti, b = ti * TS + col;
    double c = lds_b64(&L.Fr[0][a]) * lds_b64(&L.Gc[0][b]);
    c = __builtin_fma(lds_b64(&L.Fr[1][a]), lds_b64(&L.Gc[1][b]), c);
    c = __builtin_fma(lds_b64(&L.Fr[2][a]), lds_b64(&L.Gc[2][b]), c);
    const uint64_t done = __builtin_amdgcn_ballot_w64(T != 0);  // bit = 8 * slot + row
    uc[lane] = T ? c : 0.0;  // (unweighted: the owner multiplies; 0 x weight adds nothing)
    if (sel) {  // the owners of this unit's columns collect their 8 rows
      ngd_lds_cvd2 *mail = (ngd_lds_cvd2 *)&L.unit_c[wave * 64 + rank * 8];
      const ngd_d2 v0 = mail[0], v1 = mail[1], v2 = mail[2], v3 = mail[3];
      const double v[8] = {v0[0], v0[1], v1[0], v1[1], v2[0], v2[1], v3[0], v3[1]};
#pragma unroll
      for (int q = 0; q < 8; q++)
#pragma unroll
        for (int b = 0; b < RB; b++) acc[q][b] = acc[q][b] + (WEIGHTED ? v[q] * wv[b] : v[q]);
      todo &= ~((uint32_t)(done >> (rank * 8)) & 0xffu);
    }
    pm &= ~__builtin_amdgcn_ballot_w64(sel);
  }
}

#if defined(NGD_EMT_STAMPS)  // diagnostic build: where a wavefront's cycles go (tools/em_stamps.py)
#define EMT_STAMP(slot)                                                              \
  do {                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                               \
    unsigned long long _t;                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");       \
    __builtin_amdgcn_sched_barrier(0);                                               \
    stamp_sum[slot] += (double)(_t - stamp_last);                                    \
    stamp_last = _t;                                                                 \
  } while (0)
#else
#define EMT_STAMP(slot) do { } while (0)
#endif

// NW wavefronts per workgroup (64 / NW rows each), CH steps per table round, WPS = waves per SIMD the
// register allocation is held to (workgroups per CU x NW / 4)
// PACK: from the second round of a site on, only the rows in which at least PACK_DENSE pairs are still searching are
// scanned the plain way; the rest goes through packed_units() ([measured] tools/em_dense_sweep.sh, 1000 x 2e4: 8: 46.2 ms,
// 16: 44.1, 24: 44.5, 32: 44.8, 48: 47.3, no plain rows at all: 61.4)
#if !defined(NGD_PACK_DENSE)
#define NGD_PACK_DENSE 24
#endif
constexpr int PACK_DENSE = NGD_PACK_DENSE;

// RB > 1: RB matrices in one pass (bootstrap replicates whose blocks are too small for per-block partial results:
// the EM of a (pair, site) does not depend on the replicate, only its weight does).  Wb[s][RB] = the site's weight in
// each matrix; the slab holds RB planes per slice.  The RB x 8 accumulators take the registers of a second workgroup:
// one workgroup per CU.
//
// SPILL: the terms leave the kernel instead of being summed over the slice.  A bootstrap replicate weights whole BLOCKS
// of sites (rnd_map_data, ngsDist.cpp:416-437), so the terms of `spill_q` consecutive sites that share a block (a
// divisor of the block size) are added up here first and leave as ONE term per (pair, unit): 1 / spill_q of the bytes
// written, read and contracted.  Unit k of the chunk of sites [site_base, n_sites_eff) and pair slot p go to
// slab[((k / 4) * n_pad + p / 16) * 64 + (k % 4) * 16 + p % 16] (n_pad = groups of 16 pair slots here) -- the
// fragment-major operand layout of ngd_internal.h with "individual" = pair slot, so that contract_mfma.hip contracts the
// chunk with any number of bootstrap weight vectors by FP64 MFMA.  Pair slots are dealt in groups of 16 consecutive
// columns of one row of a tile, and only to groups that hold a pair: rowpg[tile * 64 + row] + g is the slot group of the
// row's column group g (a diagonal tile's lower triangle and the columns beyond n_ind get none: 7.6 % of the groups
// at 1000 individuals; rowpg is the first live group's slot group minus that group's index).  Terms that are not finite (an all-zero individual: 0/0 in normalize(), as on the CPU) raise
// the flag word `nanlist`; the chunk is then sanitised before it is contracted (0 x NaN must not reach the matrices
// that do not draw the site).
template <int NW, int CH, int WPS, bool WEIGHTED, bool PDEL, bool PACK, int RB, bool SPILL = false>
__global__ __launch_bounds__(NW * 64, WPS) void k_accum_em_table(
    const double *__restrict__ PA, const uint32_t *__restrict__ ws, const double *__restrict__ Wb, ngd_score sc,
    const ngd_tile *__restrict__ tiles, uint32_t n_tiles, uint32_t n_ig, uint32_t n_pad, uint64_t n_ind,
    uint64_t n_sites_eff, uint64_t sites_per_slice, double *__restrict__ slab,
    unsigned long long *__restrict__ counters, uint64_t site_base = 0,
    unsigned long long *__restrict__ nanlist = nullptr, uint32_t spill_q = 1, const uint32_t *__restrict__ rowpg = nullptr) {

  constexpr int RPW = TS / NW;  // rows per wavefront
  constexpr int RS = em_tables<CH, PACK>::RS;
  static_assert(!PACK || NW == 8, "packed units: 8 wavefronts x 8 rows");
  static_assert(RB == 1 || (PACK && WEIGHTED), "several matrices per pass: the packed form, weighted");
  static_assert(!SPILL || (PACK && !WEIGHTED && RB == 1), "spilled terms: the packed form, unweighted");
  // rows per group (one uniform "anything left?" test per group; their table reads are in flight together)
  constexpr int GR = (WPS >= 4 || PACK) ? 1 : 4;
  static_assert(RPW % GR == 0 && CH % 4 == 0 && (CH % 8 == 0 || CH % 8 == 4), "shape");
  __shared__ em_tables<CH, PACK> L;
  const uint32_t tile = blockIdx.x % n_tiles;
  const uint32_t ks = blockIdx.x / n_tiles;
  const uint32_t I0 = tiles[tile].ti * TS, J0 = tiles[tile].tj * TS;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (counters[2..3]: shader-cycle and constant-rate counter deltas around one workgroup's work, for the clock the launch
  // ran at -- ngd_last_shader_clock(); scalar registers)
  const bool clk_wave = blockIdx.x == (gridDim.x >> 1) && wave == 0;
  unsigned long long clk_t0 = 0, clk_r0 = 0;
  if (clk_wave) {
    clk_t0 = __builtin_amdgcn_s_memtime();
    clk_r0 = __builtin_amdgcn_s_memrealtime();
  }
#if defined(NGD_EMT_YOUNG_FIRST)
  if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);  // A/B build: the younger half of the workgroup is served first
#endif
  const uint64_t s0 = site_base + (uint64_t)ks * sites_per_slice;
  uint64_t s1 = s0 + sites_per_slice;
  if (s1 > n_sites_eff) s1 = n_sites_eff;

  // scanning role: lane = column, rows wave*RPW .. +RPW-1
  const uint32_t j = J0 + lane;
  uint32_t live = 0;  // bit r: pair (I0 + wave*RPW + r, j) exists
#pragma unroll
  for (int r = 0; r < RPW; r++)
    if (I0 + wave * RPW + r < j && j < n_ind) live |= 1u << r;
  if constexpr (SPILL) {  // bit 8 + r: this lane's group of 16 columns holds a pair of row r (the groups that get pair slots)
    static_assert(RPW == 8, "live: 8 row bits + 8 group bits");
    uint32_t gl = 0;
#pragma unroll
    for (int r = 0; r < RPW; r++) {
      const unsigned long long m = __builtin_amdgcn_ballot_w64((live >> r) & 1);
      if ((m >> (lane & 48)) & 0xffffull) gl |= 1u << (8 + r);
    }
    live |= gl;
  }
  double acc[RPW][RB];
#pragma unroll
  for (int r = 0; r < RPW; r++)
#pragma unroll
    for (int b = 0; b < RB; b++) acc[r][b] = 0;

  // building role: every wavefront builds.  Wavefronts 0 .. NW/2-1 own the tile's rows, the others its columns
  // (lane = individual); wavefront p of a role writes steps p*SEG+1 .. (p+1)*SEG of each round, so a round's CH steps
  // are SEG dependent multiplications deep instead of CH.  All of it is wave-uniform control flow.
  constexpr int NP = NW / 2;     // wavefronts per role
  constexpr int SEG = CH / NP;   // steps per wavefront and round
  static_assert(CH % NP == 0, "shape");
  // The column builders do more per step (they apply the score), and of two wavefronts of a workgroup that share a SIMD
  // the older one is served first: NGD_EMT_SWAP_ROLES (A/B build) gives the columns to the older wavefronts.
#if defined(NGD_EMT_SWAP_ROLES)
  const bool is_row = wave >= NP;
  const uint32_t seg = is_row ? wave - NP : wave;
#else
  const bool is_row = wave < NP;
  const uint32_t seg = is_row ? wave : wave - NP;
#endif
  const uint32_t bind = (is_row ? I0 : J0) + lane;
  const double *pb = PA + (uint64_t)(bind >> 4) * 64 + (bind & 15);
  const uint64_t kstride = (uint64_t)n_ig * 64;  // doubles between consecutive k-groups
  auto load_site = [&](uint64_t s, double *g) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const uint64_t k = 3 * s + c;
      g[c] = pb[(k >> 2) * kstride + (k & 3) * 16];
    }
  };
  double gn[3] = {0, 0, 0};
  if (s0 < s1) load_site(s0, gn);
  if (tid < 2) L.more[tid] = 0;
  uint32_t round = 0;  // table rounds so far (all sites): parity selects the flag word
  uint32_t sites_done = 0;
  uint32_t poison = 0;  // RB > 1, lane = the individual this lane builds: matrices that drew a site where it is all zero

#if defined(NGD_EMT_STAMPS)
  double stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
  // (a 32-bit count in a scalar register: as a 64-bit bound the compiler kept s1 in vector registers and spilled it)
  const uint32_t n_mine = s0 < s1 ? __builtin_amdgcn_readfirstlane((uint32_t)(s1 - s0)) : 0;
  // SPILL: sites of the current unit worked so far, and the unit's index within the chunk
  uint32_t in_unit = 0, unit_k = SPILL ? __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)ks * sites_per_slice) / (SPILL ? spill_q : 1u))) : 0;
  for (uint32_t si = 0; si < n_mine; si++) {
    const uint64_t s = s0 + si;
    double wgt = 1.0;
    double wv[RB];
    EMT_STAMP(0);  // loop overhead, end-of-site
    double g[3] = {gn[0], gn[1], gn[2]};
#if defined(NGD_EMT_STAMPS)
    asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]));  // the wait for the loads happens here
    EMT_STAMP(0);  // waiting for this site's likelihoods (counted with the loop overhead)
#endif
    if (si + 1 < n_mine) load_site(s + 1, gn);  // in flight while this site is worked on
    if (WEIGHTED && RB == 1) {
      const uint32_t m = ws[s];
      if (m == 0) continue;  // site not drawn in this replicate (uniform across the workgroup)
      wgt = (double)m;
    }
    wv[0] = wgt;
    uint32_t nz = 1;  // bit b: matrix b draws this site
    if (RB > 1) {
      nz = 0;
#pragma unroll
      for (int b = 0; b < RB; b++) {
        wv[b] = Wb[s * RB + b];
        nz |= (wv[b] != 0.0 ? 1u : 0u) << b;
      }
      nz = __builtin_amdgcn_readfirstlane(nz);
      if (!nz) continue;  // drawn by none of these replicates
    }
    // v = g^(t0 + seg*SEG): the power one step before this wavefront's first step of the round; gch = g^CH
    double v[3] = {1.0, 1.0, 1.0}, gch[3];
    {
      double sq[3] = {g[0], g[1], g[2]};  // g^(2^b)
#pragma unroll
      for (int bit = 0; (1 << bit) <= CH; bit++) {
        if ((seg * SEG) >> bit & 1) {
#pragma unroll
          for (int x = 0; x < 3; x++) v[x] *= sq[x];
        }
        if (CH >> bit & 1) {  // compile time: CH = 16 -> g^16, CH = 12 -> g^8 * g^4
#pragma unroll
          for (int x = 0; x < 3; x++) gch[x] = (CH & ((1 << bit) - 1)) ? gch[x] * sq[x] : sq[x];
        }
#pragma unroll
        for (int x = 0; x < 3; x++) sq[x] *= sq[x];
      }
    }
    bool miss = PDEL && ngd_miss(g[0], g[1], g[2]);
    if (RB > 1 && !miss) {
      // Several matrices per pass: a NaN term (an all-zero individual: 0/0 in normalize(), as on the CPU) would reach the
      // matrices that do NOT draw this site as 0 x NaN.  Such an individual is tabled like a missing one (its pairs stop
      // at once and add exactly 0) and remembered per matrix that draws the site: its pairs are set to NaN at the end.
      // (also a sum too small for its reciprocal -- below 2^-1022 the refined rcp is inf: the one-matrix pass makes such
      // a pair non-finite too; the prepared input of the reference sums to 1, only a raw upload can get here)
      const double a0 = (g[0] + g[1]) + g[2];
      if (!(a0 >= 2.2250738585072014e-308 && a0 <= 1.7976931348623157e308)) {
        miss = true;
        poison |= nz;
      }
    }
    uint32_t todo = SPILL ? live & 0xffu : live;
    sites_done++;
    EMT_STAMP(2);  // per-site set-up (powers)
    // (every pair has stopped by step MAX_ITER, where the tables force it: the bound only restates that)
    for (int t0 = 0; t0 < MAX_ITER; t0 += CH, round++) {  // steps t0+1 .. t0+CH
      {
        const int tfirst = t0 + (int)seg * SEG;  // this wavefront's steps are tfirst+1 .. tfirst+SEG
        if (is_row) build_round<CH, SEG, true, PACK>(L, v, g, sc, lane, seg, tfirst, miss);
        else build_round<CH, SEG, false, PACK>(L, v, g, sc, lane, seg, tfirst, miss);
#pragma unroll
        for (int x = 0; x < 3; x++) v[x] *= gch[x];
      }
      EMT_STAMP(3);  // building
      wg_barrier();
      EMT_STAMP(4);  // barrier after building
      // the other parity's word: every wavefront has read it (it is behind that read), it is next written after the
      // next round's first barrier
      if (tid == 0) L.more[(round & 1) ^ 1] = 0;
      if constexpr (PACK) {
        if (__builtin_amdgcn_ballot_w64(todo != 0)) {
          // rows scanned the plain way: all that have a pair in the first round, the dense ones later
          const uint32_t least = t0 == 0 ? 1 : PACK_DENSE;
          uint32_t rows = 0;
#pragma unroll
          for (int r = 0; r < RPW; r++)
            if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64((todo >> r) & 1)) >= least) rows |= 1u << r;
          if (rows) {
            // the row builders' verdicts on this wavefront's 8 rows: first / second block of eight steps
            typedef const volatile __attribute__((address_space(3))) unsigned long long lds_cvu64;
            unsigned long long ns[4];
#pragma unroll
            for (int p = 0; p < 4; p++) ns[p] = *(lds_cvu64 *)&L.row_nostop[p];
            const unsigned long long nsa = ns[0] & ns[1], nsb = ns[2] & ns[3];
            const uint32_t sh = (wave & 3) * 8;
            const uint32_t ska = __builtin_amdgcn_readfirstlane((uint32_t)(wave < 4 ? nsa : nsa >> 32) >> sh) & 0xff;
            const uint32_t skb = __builtin_amdgcn_readfirstlane((uint32_t)(wave < 4 ? nsb : nsb >> 32) >> sh) & 0xff;
            double R2[CH];
#pragma unroll
            for (int tt = 0; tt < CH; tt++) R2[tt] = lds_b64(&L.Rc[tt * TS + lane]);
            const uint32_t go = rows & ~(ska & skb);     // rows searched in this round
            const uint32_t need_a = go & ~ska;           // ... that search their first block
            const uint32_t load = need_a & ~(go << 1);   // ... and whose predecessor is not there to fetch its thresholds
            const uint32_t pref = need_a >> 1;           // rows that fetch their successor's
            ngd_d2 QA[4];
#pragma unroll
            for (int r = 0; r < RPW; r++) {
#if defined(NGD_EMT_FAIR)
              // A/B build: of the two wavefronts of a workgroup that share a SIMD the older is served first and reaches
              // every barrier ~9 % earlier (cycle stamps); alternating their priorities row by row shares the SIMD evenly
              if (((r & 1) != 0) != (wave >= NW / 2)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
              if ((go >> r) & 1)
                scan_row<CH, WEIGHTED, PACK, RB>(L, (wave * RPW + r) * RS, (wave * RPW + r + 1) * RS, lane, r, R2, QA,
                                                 todo, acc[r], wv, ((ska >> r) & 1) | (((skb >> r) & 1) << 1),
                                                 (load >> r) & 1, (pref >> r) & 1);
            }
#if defined(NGD_EMT_FAIR)
            __builtin_amdgcn_s_setprio(0);
#endif
          }
          if (t0 != 0) packed_units<CH, WEIGHTED, RB>(L, wave, lane, todo, acc, wv);
        }
      } else if (__builtin_amdgcn_ballot_w64(todo != 0)) {
        double R2[CH];
#pragma unroll
        for (int tt = 0; tt < CH; tt++) R2[tt] = lds_b64(&L.Rc[tt * TS + lane]);
#pragma unroll
        for (int gr = 0; gr < RPW / GR; gr++) {
          const uint32_t gmask = ((1u << GR) - 1) << (gr * GR);
          if (__builtin_amdgcn_ballot_w64((todo & gmask) != 0) == 0) continue;  // these rows are finished in every lane
          int T[GR];
#pragma unroll
          for (int q = 0; q < GR; q++) {
            const uint32_t rb = (wave * RPW + gr * GR + q) * RS;
            ngd_d2 Q[CH / 2];  // the row's CH thresholds: the same 16 bytes in every lane, one ds_read_b128 per two steps
#pragma unroll
            for (int h = 0; h < CH / 2; h++) Q[h] = *(const ngd_d2 *)&L.Qr[rb + 2 * h];
            const bool mine = (todo >> (gr * GR + q)) & 1;
            uint64_t m = __builtin_amdgcn_ballot_w64(mine);
            uint32_t n = 0;
#pragma unroll
            for (int b = 0; b < CH / 8; b++) {
              if (b && m == 0) break;  // every lane has stopped
              scan8(m, n, R2[8 * b], R2[8 * b + 1], R2[8 * b + 2], R2[8 * b + 3], R2[8 * b + 4], R2[8 * b + 5],
                    R2[8 * b + 6], R2[8 * b + 7], Q[4 * b][0], Q[4 * b][1], Q[4 * b + 1][0], Q[4 * b + 1][1],
                    Q[4 * b + 2][0], Q[4 * b + 2][1], Q[4 * b + 3][0], Q[4 * b + 3][1]);
            }
            if (CH % 8 && m != 0) {  // (12 steps per round: the last four)
              constexpr int b = CH / 4 - 1;
              scan4(m, n, R2[4 * b], R2[4 * b + 1], R2[4 * b + 2], R2[4 * b + 3], Q[2 * b][0], Q[2 * b][1], Q[2 * b + 1][0],
                    Q[2 * b + 1][1]);
            }
            T[q] = mine && n < (uint32_t)CH ? (int)n + 1 : 0;
          }
#pragma unroll
          for (int q = 0; q < GR; q++) {
            const int r = gr * GR + q;
            const uint32_t ti = T[q] ? T[q] - 1 : 0;
            const uint32_t a = (wave * RPW + r) * RS + ti, b = ti * TS + lane;
            double c = lds_b64(&L.Fr[0][a]) * lds_b64(&L.Gc[0][b]);
            c = __builtin_fma(lds_b64(&L.Fr[1][a]), lds_b64(&L.Gc[1][b]), c);
            c = __builtin_fma(lds_b64(&L.Fr[2][a]), lds_b64(&L.Gc[2][b]), c);
            if (WEIGHTED) c = c * wgt;
            if (T[q]) {
              acc[r][0] = acc[r][0] + c;
              todo &= ~(1u << r);
            }
          }
        }
      }
#if defined(NGD_EMT_STAMPS)
      if (t0 == 0) EMT_STAMP(5); else EMT_STAMP(7);  // scanning: a site's first round / its later rounds
#endif
      const bool left = __builtin_amdgcn_ballot_w64(todo != 0) != 0;
      if (left && lane == 0) L.more[round & 1] = 1;
      wg_barrier();  // also the barrier that lets the next round overwrite the tables
#if defined(NGD_EMT_STAMPS)
      if (t0 == 0) EMT_STAMP(6); else EMT_STAMP(1);  // barrier after scanning: first round / later rounds
#endif
      if (*(const volatile __attribute__((address_space(3))) uint32_t *)&L.more[round & 1] == 0) { round++; break; }
    }
    if constexpr (SPILL) {
      // the unit's last site (or the slice's): the wavefront's 8 x 64 sums leave, a run of 128 B per live group of 16
      // columns (a slice starts at a unit boundary: sites_per_slice is a multiple of spill_q)
      if (++in_unit == spill_q || si + 1 == n_mine) {
        // one store per row: the row's slot base is wave-uniform (scalar base + the lane's offset in its group), the lanes
        // of groups without a pair are masked off (bits 8.. of `live`); dead pairs of a live group leave as 0.0
        double *dunit = slab + (uint64_t)(unit_k >> 2) * n_pad * 64 + (unit_k & 3) * 16;
        const uint32_t loff = (lane >> 4) * 64 + (lane & 15);
        uint32_t row0 = tile * 64 + wave * RPW;
        asm volatile("" : "+s"(row0));  // the rows' slot bases are fetched HERE (one scalar load of 32 B): held through the
                                        // site loop they would take 8 scalar registers from the search, which spills then
        bool bad = false;
#pragma unroll
        for (int r = 0; r < RPW; r++) {
          const double c = acc[r][0];
          acc[r][0] = 0;
          bad = bad || !(__builtin_fabs(c) <= 1.7976931348623157e308);
          // rowpg: the slot group column group 0 of the row would have (its first LIVE group's, minus that group's index)
          double *drow = dunit + (int64_t)(int32_t)rowpg[row0 + r] * 64;
          if ((live >> (8 + r)) & 1) drow[loff] = c;
        }
        // a term that is not finite (an all-zero individual: 0/0 in normalize(), as on the CPU): k_spill_sanitize
        // (contract_mfma.hip) then goes over this chunk before it is contracted
        if (bad) *(volatile unsigned long long *)nanlist = 1ull;
        in_unit = 0;
        unit_k++;
      }
    }
  }
  if (tid == 0) {  // work done, for the roofline accounting of bench.py: (tile, site) visits and table rounds
    atomicAdd(&counters[0], (unsigned long long)sites_done);
    atomicAdd(&counters[1], (unsigned long long)round);
    if (clk_wave) {
      counters[2] = __builtin_amdgcn_s_memtime() - clk_t0;
      counters[3] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
  }
#if defined(NGD_EMT_STAMPS)
#pragma unroll
  for (int r = 0; r < RPW && r < 8; r++) acc[r][0] = stamp_sum[r];
#endif
  if (RB > 1) {  // (uniform: RB is a template parameter)
    if (seg == 0) L.poison[is_row ? 0 : 1][lane] = (uint8_t)poison;  // every wavefront of a role saw the same individuals
    wg_barrier();
#pragma unroll
    for (int r = 0; r < RPW; r++) {
      const uint32_t pz = (uint32_t)L.poison[0][wave * RPW + r] | (uint32_t)L.poison[1][lane];
#pragma unroll
      for (int b = 0; b < RB; b++)
        if ((pz >> b) & 1) acc[r][b] = __builtin_nan("");
    }
  }
  if constexpr (!SPILL) {
#pragma unroll
    for (int r = 0; r < RPW; r++)
#pragma unroll
      for (int b = 0; b < RB; b++)
        slab[(((uint64_t)ks * RB + b) * n_pad + (I0 + wave * RPW + r)) * n_pad + j] = acc[r][b];
  }
}

}  // namespace

// shape: 0 = 8 wavefronts x 8 rows, 16 steps per round, 4 waves per SIMD, a site's later rounds in packed units
//            (default: [measured] 1000 x 2e4, ms per launch: 44.5; shape 4, the same with every round scanned row by
//            row: 49.2 -- same bits; shape 1: 59.6; 2: 51.9; 3: 51.3; k_accum_em<fast> 122.7);
//        1 = 4 wavefronts x 16 rows, 16 steps, 2 waves per SIMD (register-rich);  2 / 3 = 0's and 1's shapes with 12
//            steps per round, rows only
void ngd_launch_accum_em_table(hipStream_t st, const ngd_geom &g, const double *PA, const uint32_t *d_ws,
                               uint64_t n_sites_eff, const ngd_score &score, int pairwise_del, int shape,
                               const ngd_tile *d_tiles64, uint32_t n_tiles64, uint32_t n_ks, uint64_t sites_per_slice,
                               double *slab, unsigned long long *d_counters) {
  if (!n_tiles64) return;
  const bool w = d_ws != nullptr, p = pairwise_del != 0;
#define NGD_EMT(NW, CH, WPS, W, P, K)                                                                                 \
  hipLaunchKernelGGL((k_accum_em_table<NW, CH, WPS, W, P, K, 1>), dim3(n_tiles64 * n_ks), dim3(NW * 64), 0, st, PA, d_ws, \
                     nullptr, score, d_tiles64, n_tiles64, g.n_ig, g.n_pad, g.n_ind, n_sites_eff, sites_per_slice, slab,  \
                     d_counters)
#define NGD_EMT_WP(NW, CH, WPS, K)                                                 \
  do {                                                                             \
    if (w) { if (p) NGD_EMT(NW, CH, WPS, true, true, K); else NGD_EMT(NW, CH, WPS, true, false, K); }   \
    else   { if (p) NGD_EMT(NW, CH, WPS, false, true, K); else NGD_EMT(NW, CH, WPS, false, false, K); } \
  } while (0)
  switch (shape) {
    default: NGD_EMT_WP(8, 16, 4, true); break;
    case 1: NGD_EMT_WP(4, 16, 2, false); break;
    case 2: NGD_EMT_WP(8, 12, 4, false); break;
    case 3: NGD_EMT_WP(4, 12, 2, false); break;
    case 4: NGD_EMT_WP(8, 16, 4, false); break;
  }
#undef NGD_EMT_WP
#undef NGD_EMT
}

// Slices ks0 .. ks0 + n_sub - 1 of a plain (unweighted) pass on their own -- the sites of the others need not be resident yet
// (engine.hip: the full-data pass started during a staged load).  lds_pad: bytes of dynamic LDS a workgroup asks for on top
// of its tables: enough of it and a CU holds ONE workgroup instead of two, which leaves registers and wave slots for the
// preparation kernels of the pieces that are still arriving.
void ngd_launch_accum_em_table_slices(hipStream_t st, const ngd_geom &g, const double *PA, const ngd_score &score, int pairwise_del,
                                      int shape, const ngd_tile *d_tiles64, uint32_t n_tiles64, uint32_t ks0, uint32_t n_sub,
                                      uint64_t sites_per_slice, double *slab, unsigned long long *d_counters, uint32_t lds_pad) {
  if (!n_tiles64 || !n_sub) return;
  const uint64_t s_lo = (uint64_t)ks0 * sites_per_slice, s_hi = std::min<uint64_t>(g.n_sites, s_lo + (uint64_t)n_sub * sites_per_slice);
  double *out = slab + (uint64_t)ks0 * g.n_pad * g.n_pad;
  const bool p = pairwise_del != 0;
#define NGD_EMT_S(NW, CH, WPS, P, K)                                                                                       \
  hipLaunchKernelGGL((k_accum_em_table<NW, CH, WPS, false, P, K, 1>), dim3(n_tiles64 * n_sub), dim3(NW * 64), lds_pad, st, PA, \
                     nullptr, nullptr, score, d_tiles64, n_tiles64, g.n_ig, g.n_pad, g.n_ind, s_hi, sites_per_slice, out,    \
                     d_counters, s_lo)
#define NGD_EMT_SP(NW, CH, WPS, K) do { if (p) NGD_EMT_S(NW, CH, WPS, true, K); else NGD_EMT_S(NW, CH, WPS, false, K); } while (0)
  switch (shape) {
    default: NGD_EMT_SP(8, 16, 4, true); break;
    case 1: NGD_EMT_SP(4, 16, 2, false); break;
    case 2: NGD_EMT_SP(8, 12, 4, false); break;
    case 3: NGD_EMT_SP(4, 12, 2, false); break;
    case 4: NGD_EMT_SP(8, 16, 4, false); break;
  }
#undef NGD_EMT_SP
#undef NGD_EMT_S
}

// rb (4 or 8) matrices in one pass of the packed form; d_Wb is [n_sites][rb] doubles, slab [n_ks][rb][n_pad][n_pad]
void ngd_launch_accum_em_table_batch(hipStream_t st, const ngd_geom &g, const double *PA, const double *d_Wb, int rb,
                                     uint64_t n_sites_eff, const ngd_score &score, int pairwise_del,
                                     const ngd_tile *d_tiles64, uint32_t n_tiles64, uint32_t n_ks,
                                     uint64_t sites_per_slice, double *slab, unsigned long long *d_counters) {
  if (!n_tiles64) return;
#define NGD_EMTB(P, RB)                                                                                                  \
  hipLaunchKernelGGL((k_accum_em_table<8, 16, 2, true, P, true, RB>), dim3(n_tiles64 * n_ks), dim3(512), 0, st, PA,       \
                     nullptr, d_Wb, score, d_tiles64, n_tiles64, g.n_ig, g.n_pad, g.n_ind, n_sites_eff, sites_per_slice, \
                     slab, d_counters)
  if (rb == 8) {
    if (pairwise_del) NGD_EMTB(true, 8); else NGD_EMTB(false, 8);
  } else {
    if (pairwise_del) NGD_EMTB(true, 4); else NGD_EMTB(false, 4);
  }
#undef NGD_EMTB
}

// The terms of sites [s_lo, s_hi), added up over units of q consecutive sites, of every live pair slot into C
// (fragment-major, n_pg groups of 16 pair slots per k-group of 4 units; see the SPILL note at the kernel; d_rowpg is the
// slot group of every tile row's first live group).  sites_per_slice must be a multiple of q.  *d_nanflag is set to 1
// if a term of the chunk was not finite.
void ngd_launch_accum_em_table_spill(hipStream_t st, const ngd_geom &g, const double *PA, uint64_t s_lo, uint64_t s_hi,
                                     const ngd_score &score, int pairwise_del, const ngd_tile *d_tiles64,
                                     uint32_t n_tiles64, uint32_t n_ks, uint64_t sites_per_slice, uint32_t q,
                                     const uint32_t *d_rowpg, uint32_t n_pg, double *C, unsigned long long *d_counters,
                                     unsigned long long *d_nanflag) {
  if (!n_tiles64 || s_hi <= s_lo) return;
#define NGD_EMTS(P)                                                                                                      \
  hipLaunchKernelGGL((k_accum_em_table<8, 16, 4, false, P, true, 1, true>), dim3(n_tiles64 * n_ks), dim3(512), 0, st, PA, \
                     nullptr, nullptr, score, d_tiles64, n_tiles64, g.n_ig, n_pg, g.n_ind, s_hi, sites_per_slice, C,      \
                     d_counters, s_lo, d_nanflag, q, d_rowpg)
  if (pairwise_del) NGD_EMTS(true); else NGD_EMTS(false);
#undef NGD_EMTS
}
