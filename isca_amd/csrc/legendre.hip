// Legendre transforms (tools/spherical_fourier.F90:177-339) on the FP64 matrix cores of gfx950.
//
// Both directions are per-wavenumber dense contractions with hemispheric even/odd folding and triangular bounds:
//   analysis  (trans_fourier_to_spherical, :264-339)  S[n][c] = sum_j' P(m,n,j') w(j') (F[N(j')][c] +- F[S(j')][c])
//   synthesis (trans_spherical_to_fourier, :177-261)  F[S/N(j')][c] = sum_{n even} P S[n][c] -+ sum_{n odd} P S[n][c]
// done with v_mfma_f64_16x16x4_f64 (A: lane l = A[row l&15][k l>>4]; B: B[k l>>4][col l&15]; D: lane l, reg r = row (l>>4)+4r, col l&15).
//
// Structure (round 2; replaces the one-shot LDS kernels of round 1):
//   * every wavefront is an independent work item (wavenumber, 32-column group, group of row tiles); no LDS, no barriers;
//   * the Legendre table is stored on the device in MFMA fragment order -- one 512-byte piece per (k-step, tile), lane-contiguous --
//     and streamed from L2 one k-step ahead of its use (all column groups of a wavenumber run on the XCD whose L2 holds its table);
//   * the contraction index is the outer loop and every output tile of the item is accumulated at once, so the other operand
//     (Fourier rows from HBM in the analysis, spectral rows in the synthesis) streams through a small register ring, prefetched
//     a group of k-steps ahead, instead of being parked in registers for the whole item;
//   * a lane owns the (re, im) pair of one level-field: its two values are columns of two column tiles, so operands are read and
//     results written as 16-byte pieces, 256 contiguous bytes per row and 16 lanes;
//   * synthesis: the B operand (div, vor, u cos, v cos, T, grad T, ln ps, grad ln ps) is formed from the spectral state with one
//     complex multiply-add per neighbour from a per-wavenumber coefficient table (compute_ucos_vcos spherical.F90:409-469,
//     compute_gradient_cos :270-351), ~20 VALU instructions per k-step against 8-16 MFMAs.
#include "kernels.h"
#include <type_traits>
#include <cstdlib>
#include <cstdio>
#include <string>

namespace isca {

typedef double double4_t __attribute__((ext_vector_type(4)));

namespace {

// Template parameters of the kernels:
//   FD   analysis: prefetch distance in k-steps (= ring slots; vmcnt retires in order, so the table pieces and the Fourier rows
//        share one ring and one distance)
//   NTG  analysis: row tiles per parity and work item (a wavenumber with more is split into row groups)
//   NW   synthesis: wavefronts per block (= k-steps per group);  JTG  synthesis: latitude tiles per wavefront

// spectral-side row offset (in doubles, fits 31 bits) of latitude j for local wavenumber slot ml; Jl = 2^lg
__device__ __forceinline__ int frow32(int j, int ml, int C, int lg, int Ml) {
  return ((((j >> lg) * Ml + ml) << lg) | (j & ((1 << lg) - 1))) * C;
}

// Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  All work items of one wavenumber share
// that wavenumber's table, so they are mapped onto the same XCD; within an XCD the wavenumbers come in ascending order, i.e. the
// long rows of the triangle first.  T = blocks per wavenumber.
__device__ __forceinline__ bool leg_block(int Ml, int T, int &ml, int &sub) {
  const int lin = blockIdx.x, xcd = lin & 7, q = lin >> 3;
  const int grp = q / T;
  ml = xcd + 8 * grp;
  sub = q - grp * T;
  return ml < Ml;
}
unsigned leg_grid(int Ml, int T) { return (unsigned)(8 * ((Ml + 7) / 8) * T); }

template <int N> using IC = std::integral_constant<int, N>;

// Scheduling trace for kernel experiments (-DLEG_TRACE; tools/dev/leg_trace.py): every working wavefront records its start and end
// (wall_clock64, 10 ns ticks), the SIMD it ran on (HW_ID, XCC_ID) and what its item was.  The launch number given in
// ISCA_LEG_TRACE_AT is dumped to gpurun_out/leg_trace_{fwd,inv}.bin.
#ifdef LEG_TRACE
struct TraceRec { long long t0, t1, ta, tb; unsigned hw_id, xcc, ml, what; };   // ta: first k-steps done, tb: last MFMA issued

#define TRACE_ARG , TraceRec *trace
#define TRACE_BEGIN const long long tr_t0 = wall_clock64(); long long tr_ta = 0, tr_tb = 0;
#define TRACE_MARK_A tr_ta = wall_clock64();
#define TRACE_MARK_B tr_tb = wall_clock64();
#define TRACE_PARAM , long long &tr_ta, long long &tr_tb
#define TRACE_PASS , tr_ta, tr_tb
#define TRACE_END(what_)                                                                                             \
  if (trace && lane == 0) {                                                                                          \
    TraceRec r; r.t0 = tr_t0; r.t1 = wall_clock64(); r.ta = tr_ta; r.tb = tr_tb; r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);            \
    r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); r.ml = ml; r.what = (what_);                                \
    trace[blockIdx.x * (blockDim.x >> 6) + wave] = r;                                                                                \
  }
#else
#define TRACE_ARG
#define TRACE_BEGIN
#define TRACE_END(what_)
#define TRACE_MARK_A
#define TRACE_MARK_B
#define TRACE_PARAM
#define TRACE_PASS
#endif

struct LegFwdArgs {
  const double *frag;      // [Ml][KS/2][2][NTP][64][2]: the pieces of k-steps 2q and 2q + 1 side by side (one 16-byte load per lane brings both)
  const double *Fs;        // Fourier rows, spectral-side view
  double *S;               // [Ml][N1][C]
  const int *m_local;
  int C, full, CB;         // CB = blocks of 4 column groups per wavenumber
  int KS, NTP, RG;         // k-steps (Jh/4), tiles per parity in the table, row groups
};

// One analysis work item: NE even-parity and NO odd-parity row tiles (first tile t0 of each parity) for the 16*NCT columns c0...
template <int NE, int NO, int FD>
__device__ __forceinline__ void leg_fwd_item(const Geom &g, const LegFwdArgs &a, int ml, int t0, int c0, int lane TRACE_PARAM) {
  constexpr int TT = NE + NO;
  const int cl = lane & 15, kq = lane >> 4;
  const int c = c0 + 2 * cl;                          // this lane's (re, im) pair
  const bool cok = c < a.C;
  const int cc = cok ? c : 0;
  const int lg = g.log2Jl, Ml = g.Ml, C = a.C, J = g.J;
  // Table pieces come as 16-byte loads, two k-steps per request: a vector-memory instruction costs the CU's address unit the same ~16
  // cycles whether a lane asks for 8 or for 16 bytes (tools/micro/mfma_f64_probe.hip: a table streamed with 8-byte loads arrives at
  // 32 B/clk/CU, with 16-byte loads at 64), and the pieces are most of this kernel's requests.
  static_assert(FD % 2 == 0, "ring of k-step pairs");
  constexpr int FP = FD / 2;
  const int kp_stride = 2 * a.NTP * 128, par_stride = a.NTP * 128;
  const double *fr = a.frag + (size_t)ml * (a.KS / 2) * kp_stride + t0 * 128 + 2 * lane;
  const double *Fs = a.Fs;
  double4_t acc[TT][2];
#pragma unroll
  for (int t = 0; t < TT; ++t) { acc[t][0] = (double4_t){0., 0., 0., 0.}; acc[t][1] = (double4_t){0., 0., 0., 0.}; }
  double2 xs[FD], xn[FD];
  double2 af[FP][TT];
  auto load = [&](int ps, int kp) {                    // table pieces and Fourier rows of k-steps 2 kp, 2 kp + 1 (ps is a constant after unrolling)
    const double *p = fr + kp * kp_stride;
#pragma unroll
    for (int t = 0; t < NE; ++t) af[ps][t] = *(const double2 *)(p + t * 128);
#pragma unroll
    for (int t = 0; t < NO; ++t) af[ps][NE + t] = *(const double2 *)(p + par_stride + t * 128);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int jp = (2 * kp + h) * 4 + kq;
      xs[2 * ps + h] = *(const double2 *)(Fs + frow32(jp, ml, C, lg, Ml) + cc);
      xn[2 * ps + h] = *(const double2 *)(Fs + frow32(J - 1 - jp, ml, C, lg, Ml) + cc);
    }
  };
  auto block = [&](auto morec, int ks0) {              // FD k-steps; refills a pair slot with the pair FD k-steps on once both halves are consumed
    constexpr bool MORE = decltype(morec)::value != 0;
#pragma unroll
    for (int d = 0; d < FD; ++d) {
      const double2 be = make_double2(xn[d].x + xs[d].x, xn[d].y + xs[d].y);   // x_even = F(north) + F(south)   (:311)
      const double2 bo = make_double2(xn[d].x - xs[d].x, xn[d].y - xs[d].y);   // x_odd  = F(north) - F(south)   (:312)
#pragma unroll
      for (int t = 0; t < NE; ++t) {
        const double av = (d & 1) ? af[d >> 1][t].y : af[d >> 1][t].x;
        acc[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, be.x, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, be.y, acc[t][1], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < NO; ++t) {
        const double av = (d & 1) ? af[d >> 1][NE + t].y : af[d >> 1][NE + t].x;
        acc[NE + t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bo.x, acc[NE + t][0], 0, 0, 0);
        acc[NE + t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bo.y, acc[NE + t][1], 0, 0, 0);
      }
      if (MORE && (d & 1)) load(d >> 1, (ks0 + FD) / 2 + (d >> 1));
      __builtin_amdgcn_sched_barrier(0);               // keep the k-steps apart: a hoisted fold would wait for the whole ring
    }
  };
  const int KS = a.KS;                                 // a multiple of FD
#pragma unroll
  for (int ps = 0; ps < FP; ++ps) load(ps, ps);
  for (int ks0 = 0; ks0 < KS - FD; ks0 += FD) {
    block(IC<1>(), ks0);
#ifdef LEG_TRACE
    if (ks0 == 0) TRACE_MARK_A
#endif
  }
  block(IC<0>(), KS - FD);
  TRACE_MARK_B
  if (!cok) return;
  double *Sp = a.S + (ml * g.N1) * C + c;
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int par = t < NE ? 0 : 1, tl = t0 + (t < NE ? t : t - NE);
    const int n0 = 2 * (tl * 16 + kq) + par;           // rows n0 + 8 r
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n0 + 8 * r < g.N1) *(double2 *)(Sp + (n0 + 8 * r) * C) = make_double2(acc[t][0][r], acc[t][1][r]);
  }
}

template <int FD, int NTG, int WPS>
__global__ __launch_bounds__(256, WPS) void k_leg_fwd(Geom g, LegFwdArgs a TRACE_ARG) {
  TRACE_BEGIN
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  int ml, sub;
  if (!leg_block(g.Ml, a.CB * a.RG, ml, sub)) return;
  const int m = a.m_local[ml];
  if (m < 0) return;
  const int rg = sub / a.CB, cb = sub - rg * a.CB;
  const int c0 = (cb * 4 + wave) * 32;
  if (c0 >= a.C) return;
  const int nlim = a.full ? g.N1 : g.N1 - m;
  const int nt0 = (((nlim + 1) >> 1) + 15) >> 4, nt1 = ((nlim >> 1) + 15) >> 4;
  const int t0 = rg * NTG;
  const int ne = min(nt0 - t0, NTG), no = max(min(nt1 - t0, NTG), 0);
  if (ne <= 0) return;
  // the odd-parity half has as many row tiles as the even one or one fewer (none at all only behind a single even tile)
#define ITEM(E, O) if constexpr ((E) <= NTG) { if (ne == (E) && no == (O)) { leg_fwd_item<E, O, FD>(g, a, ml, t0, c0, lane TRACE_PASS); } }
  ITEM(1, 0) ITEM(1, 1) ITEM(2, 1) ITEM(2, 2) ITEM(3, 2) ITEM(3, 3) ITEM(4, 3) ITEM(4, 4) ITEM(5, 4) ITEM(5, 5) ITEM(6, 5) ITEM(6, 6)
#undef ITEM
  TRACE_END(ne * 8 + no)
}

// ---------------------------------------------------------------------------------------------------------------------
// Synthesis.  FUSED: the spectral rows are formed from the state with b(n) = (ar + i ai) z_c(n) + cm z_n(n-1) + cp z_n(n+1);
// (ar, ai, cm, cp) per (wavenumber, n, kind) come from a table that is zero outside the triangle, so no bounds tests are needed
// in the loop.   kinds: 0 copy | 1 u cos from (div; vor) | 2 v cos from (vor; div) | 3 d/dx cos | 4 d/dy cos
// ---------------------------------------------------------------------------------------------------------------------
struct LegInvArgs {
  const double *frag;      // [Ml][2][NKS/2][JT][64][2]: the pieces of k-steps 2q and 2q + 1 side by side (16-byte loads, see leg_fwd_item)
  const double *S;         // staged spectral rows [Ml][N1][C] (not FUSED)
  double *Fs;
  const int *m_local;
  const double *vor, *div, *ts, *lnps;   // FUSED: spectral state at the level being synthesised
  const double *scoef;     // FUSED: [Ml][NR][5][4]
  int C, full, CB;         // CB: 32-column groups (= blocks) per wavenumber
  int NKS, JT, NR;         // k-steps per parity in the table (NHP/4), latitude tiles (Jh/16), rows of scoef per wavenumber
  int dxf;                 // FUSED: the batch has no x-derivative columns (6 L + 2 level-fields: div, vor, u, v, dT/dy, T, ln ps, d ln ps / dy): the inverse FFT
                           // forms them from the Fourier rows of T and ln ps (FieldList::dx)
};
// level-field index of a fused synthesis column -> field number of the full batch (0 div, 1 vor, 2 u, 3 v, 4 T, 5 dT/dx, 6 dT/dy, 7 ln ps, 8, 9 its gradient) and level
__device__ __forceinline__ int leg_inv_field(int lf, int L, int dxf, int &k) {
  const int n3 = dxf ? 6 : 7;
  int f;
  if (lf < n3 * L) { f = lf / L; k = lf - f * L; if (dxf && f >= 4) f = (f == 4) ? 6 : 4; }      // (dxf: column group 4 is dT/dy, 5 is T)
  else { f = 7 + (lf - n3 * L); k = 0; if (dxf && f == 8) f = 9; }
  return f;
}

// ---------------------------------------------------------------------------------------------------------------------
// Block = (wavenumber, 32-column group); its NW wavefronts own JTG latitude tiles each (all of them
// together: NW * JTG = Jh / 16 where that is possible) and SHARE the spectral rows: in every group of NW k-steps each wavefront
// forms the rows of one k-step (gather from the state + stencil coefficients, FUSED) and leaves them in LDS, so the gather costs
// 1/NW of what it costs a wavefront working alone, and its loads have a whole group of MFMAs to arrive.  Table pieces stream
// through a ring of NW slots (one group ahead).  One barrier per group.
// ---------------------------------------------------------------------------------------------------------------------
template <int NW, int JTG, bool FUSED, bool NB>
__device__ __forceinline__ void leg_inv_coop(const Geom &g, const LegInvArgs &a, int ml, int m, int c0, int wave, int lane,
                                             double2 *Bbuf TRACE_PARAM) {
  const int cl = lane & 15, kq = lane >> 4;
  const int c = c0 + 2 * cl;
  const bool cok = c < a.C;
  const int C = a.C, N1 = g.N1, L = g.L;
  const int nlim = a.full ? N1 : N1 - m;
  const int nks0 = (((nlim + 1) >> 1) + 3) >> 2;
  const int NGR = (nks0 + NW - 1) / NW;
  const bool dup = wave * JTG >= a.JT;                // more wavefronts than tiles (small grids): repeat the last tiles, store nothing
  const int jt0 = dup ? a.JT - JTG : wave * JTG;
  constexpr bool PAIR = NW % 2 == 0;                  // table pieces as 16-byte loads of two k-steps (one wavefront per block: 8-byte loads)
  constexpr int NA = PAIR ? NW / 2 : NW;
  const int kp_stride = a.JT * 128, par_stride = (a.NKS / 2) * a.JT * 128;
  const double *fr = a.frag + (size_t)ml * 2 * par_stride + jt0 * 128 + 2 * lane;
  const double2 *pc = nullptr, *pn = nullptr;       // centre array, neighbour array at (ml, n = 0, level)
  int stride = 0;                                   // double2 per n
  const double4_t *pcoef = nullptr;
  int soff = 0;                                     // not FUSED: offset of (ml, n = 0, c) in S
  if (FUSED) {
    const int lfr = c >> 1, lf = cok ? lfr : 0;
    int k;
    const int f = leg_inv_field(lf, L, a.dxf, k);
    stride = (f < 7) ? L : 1;
    const size_t e0 = (f < 7) ? (size_t)ml * N1 * L + k : (size_t)ml * N1;
    const double2 *vor = (const double2 *)a.vor + e0, *div = (const double2 *)a.div + e0;
    const double2 *tt = (const double2 *)(f < 7 ? a.ts : a.lnps) + e0;
    int kind = 0;
    switch (f) {
      case 0: pc = div; pn = div; break;
      case 1: pc = vor; pn = vor; break;
      case 2: pc = div; pn = vor; kind = 1; break;
      case 3: pc = vor; pn = div; kind = 2; break;
      case 4: case 7: pc = tt; pn = tt; break;
      case 5: case 8: pc = tt; pn = tt; kind = 3; break;
      default: pc = tt; pn = tt; kind = 4; break;     // 6, 9
    }
    pcoef = (const double4_t *)a.scoef + (size_t)ml * a.NR * 5 + kind;
  } else {
    soff = ml * N1 * C + (cok ? c : 0);
  }
  double4_t acc[2][JTG][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int jt = 0; jt < JTG; ++jt) { acc[p][jt][0] = (double4_t){0., 0., 0., 0.}; acc[p][jt][1] = (double4_t){0., 0., 0., 0.}; }
  double2 af[NA][2][JTG];                              // PAIR: .x / .y = the piece of the even / odd k-step; else .x only
  double2 zc[2], zn[4];
  double4_t cf[2];
  int gks = 0;                                         // k-step whose rows are in zc / zn / cf
  auto gload = [&](int ks) {                           // request the rows of k-step ks (clamped to the last one of this wavenumber)
    gks = min(ks, nks0 - 1);
    const int n0 = 8 * gks + 2 * kq;                   // rows n0 (even parity) and n0 + 1 (odd parity)
    const int na = min(n0, N1 - 1), nb = min(n0 + 1, N1 - 1);
    if (FUSED) {
      zc[0] = pc[na * stride]; zc[1] = pc[nb * stride];
      cf[0] = pcoef[n0 * 5]; cf[1] = pcoef[(n0 + 1) * 5];              // table rows exist (zero) up to N1 + 15
      if (NB) {
        zn[0] = pn[max(n0 - 1, 0) * stride]; zn[1] = pn[na * stride];
        zn[2] = pn[nb * stride]; zn[3] = pn[min(n0 + 2, N1 - 1) * stride];
      }
    } else {
      zc[0] = *(const double2 *)(a.S + soff + na * C);
      zc[1] = *(const double2 *)(a.S + soff + nb * C);
    }
  };
  auto gstore = [&](int buf) {                         // rows of k-step gks -> LDS slot `wave` of buffer buf
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      double2 v;
      if (FUSED) {
        const double4_t q = cf[p];
        const double2 z = zc[p];
        v = make_double2(q[0] * z.x - q[1] * z.y, q[0] * z.y + q[1] * z.x);
        if (NB) {
          const double2 zm = zn[p], zp = zn[p + 2];            // one neighbour after the other, each a fused multiply-add:
          v.x = __builtin_fma(q[3], zp.x, __builtin_fma(q[2], zm.x, v.x));   // the roundings of the staged kernel (k_spec_synth_inputs)
          v.y = __builtin_fma(q[3], zp.y, __builtin_fma(q[2], zm.y, v.y));
        }
      } else {
        const bool in = 8 * gks + 2 * kq + p < nlim;
        v = make_double2(in ? zc[p].x : 0.0, in ? zc[p].y : 0.0);
      }
      Bbuf[((buf * NW + wave) * 2 + p) * 64 + lane] = v;
    }
  };
  auto afload = [&](int q, int ks) {                   // PAIR: slot q holds k-steps 2 ks', 2 ks' + 1 with ks' = ks (a pair index)
    if constexpr (PAIR) {
      const double *p = fr + min(ks, a.NKS / 2 - 1) * kp_stride;
#pragma unroll
      for (int jt = 0; jt < JTG; ++jt) { af[q][0][jt] = *(const double2 *)(p + jt * 128); af[q][1][jt] = *(const double2 *)(p + par_stride + jt * 128); }
    } else {
      const int kc = min(ks, a.NKS - 1);
      const double *p = fr + (kc >> 1) * kp_stride + (kc & 1);
#pragma unroll
      for (int jt = 0; jt < JTG; ++jt) { af[q][0][jt].x = p[jt * 128]; af[q][1][jt].x = p[par_stride + jt * 128]; }
    }
  };
  double2 bq[2][2];                                    // rows of the current and the next k-step, read from LDS one k-step ahead
  auto bread = [&](int q, int buf) {
    bq[q & 1][0] = Bbuf[((buf * NW + q) * 2 + 0) * 64 + lane];
    bq[q & 1][1] = Bbuf[((buf * NW + q) * 2 + 1) * 64 + lane];
  };
  auto piece = [&](int q, int par, int jt) -> double {
    if constexpr (PAIR) return (q & 1) ? af[q >> 1][par][jt].y : af[q >> 1][par][jt].x;
    else return af[q][par][jt].x;
  };
  auto mfmas = [&](int q) {
    const double2 b0 = bq[q & 1][0], b1 = bq[q & 1][1];
#pragma unroll
    for (int jt = 0; jt < JTG; ++jt) {
      const double av = piece(q, 0, jt);
      acc[0][jt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0.x, acc[0][jt][0], 0, 0, 0);
      acc[0][jt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0.y, acc[0][jt][1], 0, 0, 0);
    }
#pragma unroll
    for (int jt = 0; jt < JTG; ++jt) {
      const double av = piece(q, 1, jt);
      acc[1][jt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1.x, acc[1][jt][0], 0, 0, 0);
      acc[1][jt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1.y, acc[1][jt][1], 0, 0, 0);
    }
  };
  // ---- prologue: rows of group 0 into LDS, requests for group 1, table pieces of group 0.  Requests retire in order, so the
  // prologue leaves them queued as every pass of the loop does (table pieces, then rows): the waits inside the loop then are for
  // exactly the pieces needed, with everything requested later still in flight.
  gload(wave);
#pragma unroll
  for (int q = 0; q < NA; ++q) afload(q, q);
  __builtin_amdgcn_sched_barrier(0);
  gstore(0);
  gload(NW + wave);
  __syncthreads();
  TRACE_MARK_A
  // ---- all groups but the last: every k-step is inside the triangle; loads are unconditional (clamped), see k_leg_inv
  for (int G = 0; G < NGR - 1; ++G) {
    const int buf = G & 1;
    bread(0, buf);
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      if (q + 1 < NW) bread(q + 1, buf);
      mfmas(q);
      if constexpr (PAIR) { if (q & 1) afload(q >> 1, ((G + 1) * NW + q) >> 1); }      // both halves of the pair consumed: the pair one group on
      else afload(q, (G + 1) * NW + q);
      __builtin_amdgcn_sched_barrier(0);
    }
    gstore(1 - buf);                                   // rows of group G + 1 (requested one group ago)
    gload((G + 2) * NW + wave);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  // ---- last group: only the k-steps below nks0, nothing more to request
  {
    const int G = NGR - 1, buf = G & 1;
#pragma unroll
    for (int q = 0; q < NW; ++q)
      if (G * NW + q < nks0) { bread(q, buf); mfmas(q); }
  }
  TRACE_MARK_B
  if (!cok || dup) return;
  // rows jp = jt*16 + kq + 4r and their mirrors J-1-jp stay inside one 16-aligned latitude group (Jl % 16 == 0)
#pragma unroll
  for (int jt = 0; jt < JTG; ++jt) {
    const int jp = (jt0 + jt) * 16 + kq;
    double *south = a.Fs + frow32(jp, ml, C, g.log2Jl, g.Ml) + c;
    double *north = a.Fs + frow32(g.J - 1 - jp, ml, C, g.log2Jl, g.Ml) + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double ex = acc[0][jt][0][r], ey = acc[0][jt][1][r], ox = acc[1][jt][0][r], oy = acc[1][jt][1][r];
      *(double2 *)(south + 4 * r * C) = make_double2(ex - ox, ey - oy);      // southern row  (:235)
      *(double2 *)(north - 4 * r * C) = make_double2(ex + ox, ey + oy);      // northern mirror (:236)
    }
  }
}

template <int NW, int JTG, bool FUSED, int WPS>
__global__ __launch_bounds__(64 * NW, WPS) void k_leg_inv_coop(Geom g, LegInvArgs a TRACE_ARG) {
  TRACE_BEGIN
  __shared__ double2 Bbuf[2 * NW * 2 * 64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  int ml, cg;
  if (!leg_block(g.Ml, a.CB, ml, cg)) return;          // CB: column groups per wavenumber
  const int m = a.m_local[ml];
  if (m < 0) return;
  const int c0 = cg * 32;
  if (FUSED) {
    // columns that are plain copies of a state array (div, vor, T, ln ps) need no neighbour rows: block-uniform test
    const int L = g.L, lf0 = c0 >> 1, lf1 = min(lf0 + 15, (a.C >> 1) - 1);
    auto field = [&](int lf) { int k_; return leg_inv_field(lf, L, a.dxf, k_); };
    const int f0 = field(lf0), f1 = field(lf1);
    if (f0 == f1 && (f0 == 0 || f0 == 1 || f0 == 4 || f0 == 7)) leg_inv_coop<NW, JTG, true, false>(g, a, ml, m, c0, wave, lane, Bbuf TRACE_PASS);
    else leg_inv_coop<NW, JTG, true, true>(g, a, ml, m, c0, wave, lane, Bbuf TRACE_PASS);
  } else {
    leg_inv_coop<NW, JTG, false, false>(g, a, ml, m, c0, wave, lane, Bbuf TRACE_PASS);
  }
  TRACE_END(m)
}

// plain-FMA check kernels (legendre_impl = 1, and lat_max not a multiple of 32)
__device__ __forceinline__ size_t frow(const Geom &g, int j, int ml, int C) {   // spectral-side row of latitude j
  const int p = j / g.Jl, jl = j - p * g.Jl;
  return ((size_t)(p * g.Ml + ml) * g.Jl + jl) * C;
}
__global__ void k_leg_fwd_simple(Geom g, const int *__restrict__ m_local, const double *__restrict__ pw,
                                 const double *__restrict__ Fs, double *__restrict__ S, int C, int full) {
  const int c = blockIdx.x * 64 + threadIdx.x, n = blockIdx.y, ml = blockIdx.z;
  const int m = m_local[ml];
  if (m < 0 || c >= C) return;
  const int nlim = full ? g.N1 : g.N1 - m;
  if (n >= nlim) return;
  const int par = n & 1, nh = n >> 1;
  const double *A = pw + ((size_t)(ml * 2 + par) * g.Jh) * g.NHP + nh;
  double acc = 0.0;
  for (int jp = 0; jp < g.Jh; ++jp) {
    const double xs = Fs[frow(g, jp, ml, C) + c], xn = Fs[frow(g, g.J - 1 - jp, ml, C) + c];
    acc += (par ? (xn - xs) : (xn + xs)) * A[(size_t)jp * g.NHP];
  }
  S[((size_t)ml * g.N1 + n) * C + c] = acc;
}
__global__ void k_leg_inv_simple(Geom g, const int *__restrict__ m_local, const double *__restrict__ pinv,
                                 const double *__restrict__ S, double *__restrict__ Fs, int C, int full) {
  const int c = blockIdx.x * 64 + threadIdx.x, jp = blockIdx.y, ml = blockIdx.z;
  const int m = m_local[ml];
  if (m < 0 || c >= C) return;
  const int nlim = full ? g.N1 : g.N1 - m;
  double e = 0.0, o = 0.0;
  for (int n = 0; n < nlim; ++n) {
    const int par = n & 1, nh = n >> 1;
    const double p = pinv[((size_t)(ml * 2 + par) * g.NHP + nh) * g.Jh + jp];
    const double sv = S[((size_t)ml * g.N1 + n) * C + c];
    if (par) o += sv * p; else e += sv * p;
  }
  Fs[frow(g, jp, ml, C) + c] = e - o;
  Fs[frow(g, g.J - 1 - jp, ml, C) + c] = e + o;
}

int env_int(const char *name, int dflt) {          // experiments build only (kernels.h: exp_env)
  const char *v = exp_env(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

#ifdef LEG_TRACE
namespace {
struct TraceHost {
  TraceRec *dev = nullptr; size_t cap = 0; long count = 0; const char *name;
  explicit TraceHost(const char *n) : name(n) {}
  TraceRec *arm(size_t nwaves, hipStream_t s) {        // non-null for the one launch that is recorded
    static const long at = env_int("ISCA_LEG_TRACE_AT", -1);
    if (count++ != at) return nullptr;
    cap = nwaves;
    hipMalloc((void **)&dev, cap * sizeof(TraceRec));
    hipMemsetAsync(dev, 0, cap * sizeof(TraceRec), s);
    return dev;
  }
  void dump(hipStream_t s) {
    if (!dev) return;
    hipStreamSynchronize(s);
    std::vector<TraceRec> h(cap);
    hipMemcpy(h.data(), dev, cap * sizeof(TraceRec), hipMemcpyDeviceToHost);
    std::string path = std::string("gpurun_out/leg_trace_") + name + ".bin";
    if (FILE *f = fopen(path.c_str(), "wb")) { fwrite(h.data(), sizeof(TraceRec), cap, f); fclose(f); }
    hipFree(dev); dev = nullptr;
  }
};
TraceHost trace_fwd("fwd"), trace_inv("inv");
}
#define TRACE_LAUNCH(th, nblocks) , th.arm((size_t)(nblocks) * 4, s)
#define TRACE_DUMP(th) th.dump(s);
#else
#define TRACE_LAUNCH(th, nblocks)
#define TRACE_DUMP(th)
#endif

bool legendre_mfma_ok(const Geom &g, int impl) {
  // 32-bit shift/mask row addressing, latitude tiles of 16 inside one band, whole groups of k-steps, even column counts
  if (impl != 0 || g.log2Jl < 0 || g.Jl % 16) return false;
  return g.Jh % 16 == 0 && g.Jh >= 16;
}

// Device tables in fragment order, and the coefficient table of the fused synthesis (host side, at create).
void build_legendre_fragments(const Geom &g, const Tables &T, const std::vector<int> &m_local, std::vector<double> &fwd,
                              std::vector<double> &inv, std::vector<double> &scoef) {
  const int KS = g.Jh / 4, NTP = g.NHP / 16, NKS = g.NHP / 4, JT = g.Jh / 16, N1 = g.N1, NR = N1 + 16;
  fwd.assign((size_t)g.Ml * KS * 2 * NTP * 64, 0.0);
  inv.assign((size_t)g.Ml * 2 * NKS * JT * 64, 0.0);
  scoef.assign((size_t)g.Ml * NR * 5 * 4, 0.0);
  for (int ml = 0; ml < g.Ml; ++ml) {
    const int m = m_local[ml];
    if (m < 0) continue;
    auto P = [&](int n, int jp) { return n < N1 ? T.legendre[((size_t)jp * N1 + n) * g.M1 + m] : 0.0; };
    for (int ks = 0; ks < KS; ++ks)
      for (int par = 0; par < 2; ++par)
        for (int t = 0; t < NTP; ++t)
          for (int l = 0; l < 64; ++l) {
            const int n = 2 * (16 * t + (l & 15)) + par, jp = 4 * ks + (l >> 4);
            fwd[(((((size_t)ml * (KS / 2) + (ks >> 1)) * 2 + par) * NTP + t) * 64 + l) * 2 + (ks & 1)] = P(n, jp) * T.wts_hem[jp];
          }
    for (int par = 0; par < 2; ++par)
      for (int ks = 0; ks < NKS; ++ks)
        for (int jt = 0; jt < JT; ++jt)
          for (int l = 0; l < 64; ++l) {
            const int n = 2 * (4 * ks + (l >> 4)) + par, jp = 16 * jt + (l & 15);
            inv[(((((size_t)ml * 2 + par) * (NKS / 2) + (ks >> 1)) * JT + jt) * 64 + l) * 2 + (ks & 1)] = P(n, jp);
          }
    const int nlim = N1 - m;                          // the fused synthesis is the step's: triangular bounds
    for (int n = 0; n < nlim; ++n) {
      const size_t mn = (size_t)n * g.M1 + m;
      double *q = &scoef[((size_t)ml * NR + n) * 20];
      const double lo = n >= 1 ? 1.0 : 0.0, hi = n + 1 < N1 ? 1.0 : 0.0;
      q[0] = 1.0;                                                                                      // copy
      q[4 + 1] = T.coef_uvc[mn]; q[4 + 2] = lo * T.coef_uvm[mn]; q[4 + 3] = -hi * T.coef_uvp[mn];      // u cos: i uvc div + uvm vor(n-1) - uvp vor(n+1)
      q[8 + 1] = T.coef_uvc[mn]; q[8 + 2] = -lo * T.coef_uvm[mn]; q[8 + 3] = hi * T.coef_uvp[mn];      // v cos: i uvc vor - uvm div(n-1) + uvp div(n+1)
      q[12 + 1] = T.coef_dx[mn];                                                                       // d/dx cos: i dx z
      q[16 + 2] = -lo * T.coef_dym[mn]; q[16 + 3] = hi * T.coef_dyp[mn];                               // d/dy cos: -dym z(n-1) + dyp z(n+1)
    }
  }
}

#ifdef ISCA_EXPERIMENTS
// Tables of the fused analysis (kernels.hip k_fft_leg_fwd): the triangle's tiles of 16 n dealt round-robin to the four MFMA wavefronts of a
// block, and P(m, n, j') w(j') in the operand order of v_mfma_f64_4x4x4_4b_f64 -- A[b][i][k] in lane 16 k + 4 b + i with b < 2 the tile's even
// n (n = 16 T + 2 (4 (b & 1) + i)), b >= 2 its odd n, k the latitude pair inside the k-step -- per (wavefront, chunk of 8 pairs, tile), the
// pieces of the chunk's two k-steps side by side.  Returns NT (tiles per wavefront, padded to the kernel's instantiations) or 0.
int build_fused_fwd_tables(const Geom &g, const Tables &T, const std::vector<int> &m_local, std::vector<double> &frag, std::vector<int> &desc) {
  struct Tile { int ml, m, n0, nlim; };
  std::vector<Tile> tiles;
  for (int ml = 0; ml < g.Ml; ++ml) {
    const int m = m_local[ml];
    if (m < 0) continue;
    const int nlim = g.N1 - m;
    for (int n0 = 0; n0 < nlim; n0 += 16) tiles.push_back({ml, m, n0, nlim});
  }
  const int per = ((int)tiles.size() + 3) / 4;
  const int NT = per <= 24 ? 24 : (per <= 48 ? 48 : (per <= 72 ? 72 : 0));
  if (!NT) return 0;
  const int NCH = g.Jh / 8, N1 = g.N1;
  frag.assign((size_t)4 * NCH * NT * 64 * 2, 0.0);
  desc.assign((size_t)4 * NT * 2, 0);
  for (int w = 0; w < 4; ++w)
    for (int t = 0; t < NT; ++t) {
      const size_t gi = (size_t)4 * t + w;
      int *dq = &desc[((size_t)w * NT + t) * 2];
      if (gi >= tiles.size()) { dq[0] = -1; dq[1] = 0; continue; }
      const Tile &tl = tiles[gi];
      dq[0] = tl.ml; dq[1] = tl.n0 | (tl.nlim << 16);
      for (int c = 0; c < NCH; ++c)
        for (int l = 0; l < 64; ++l)
          for (int h = 0; h < 2; ++h) {
            const int k = l >> 4, b = (l >> 2) & 3, i = l & 3;
            const int n = tl.n0 + 2 * (4 * (b & 1) + i) + (b >> 1), jp = 8 * c + 4 * h + k;
            const double v = (n < tl.nlim) ? T.legendre[((size_t)jp * N1 + n) * g.M1 + tl.m] * T.wts_hem[jp] : 0.0;
            frag[((((size_t)w * NCH + c) * NT + t) * 64 + l) * 2 + h] = v;
          }
    }
  return NT;
}
#endif  // ISCA_EXPERIMENTS

void launch_legendre_forward(const Geom &g, const Dev &d, const double *Fs, double *S, int C, int full, int impl, hipStream_t s) {
  if (legendre_mfma_ok(g, impl) && C % 2 == 0) {
    LegFwdArgs a;
    a.frag = d.leg_fwd_frag; a.Fs = Fs; a.S = S; a.m_local = d.m_local; a.C = C; a.full = full;
    a.KS = g.Jh / 4; a.NTP = g.NHP / 16;
    a.CB = ((C + 31) / 32 + 3) / 4;
    constexpr int FD = 4;                            // measured against (8, 3), (4 / 8 / 16, 1 or 2), (4, 6): HISTORY.md
    // Row tiles per parity and work item.  3 (+3) is what a grid that fills the SIMDs wants: the Fourier rows are read once per row group.  A shard of
    // the sharded model (Ml = M1 / P wavenumbers) has far fewer items than the device has SIMDs -- 132 wavefronts at T85L40 on 8 ranks -- and each of
    // them is a serial chain of Jh / 4 k-steps: there one row tile per item gives three times the wavefronts a third of the chain each (the rows a
    // group re-reads come from L2).  ISCA_LEG_NTG: measurement switch.
    static const int ntg_env = env_int("ISCA_LEG_NTG", 0);
    const int items3 = g.Ml * ((C + 31) / 32) * ((a.NTP + 2) / 3);
    const int NTG = (ntg_env == 1 || ntg_env == 2 || ntg_env == 3) ? ntg_env : (items3 >= 512 ? 3 : 1);
    a.RG = (a.NTP + NTG - 1) / NTG;
    const dim3 grid(leg_grid(g.Ml, a.CB * a.RG));
    [[maybe_unused]] static const int fd = env_int("ISCA_LEG_FD", 0);       // experiment: ring depth 8 / 16 with one wavefront per SIMD
    if (NTG == 1) hipLaunchKernelGGL((k_leg_fwd<FD, 1, 4>), grid, dim3(256), 0, s, g, a TRACE_LAUNCH(trace_fwd, grid.x));
#ifdef ISCA_EXPERIMENTS
    else if (NTG == 2) hipLaunchKernelGGL((k_leg_fwd<FD, 2, 2>), grid, dim3(256), 0, s, g, a TRACE_LAUNCH(trace_fwd, grid.x));
    else if (fd == 8 && a.KS % 8 == 0) hipLaunchKernelGGL((k_leg_fwd<8, 3, 1>), grid, dim3(256), 0, s, g, a TRACE_LAUNCH(trace_fwd, grid.x));
    else if (fd == 16 && a.KS % 16 == 0) hipLaunchKernelGGL((k_leg_fwd<16, 3, 1>), grid, dim3(256), 0, s, g, a TRACE_LAUNCH(trace_fwd, grid.x));
#endif
    else hipLaunchKernelGGL((k_leg_fwd<FD, 3, 2>), grid, dim3(256), 0, s, g, a TRACE_LAUNCH(trace_fwd, grid.x));
    TRACE_DUMP(trace_fwd)
  } else {
    dim3 grid((C + 63) / 64, g.N1, g.Ml);
    hipLaunchKernelGGL(k_leg_fwd_simple, grid, dim3(64), 0, s, g, d.m_local, d.pw_fwd, Fs, S, C, full);
  }
}

void launch_legendre_inverse(const Geom &g, const Dev &d, const double *S, double *Fs, int C, int full, int impl, hipStream_t s, int fused_tl, int dxf) {
  const bool mfma = legendre_mfma_ok(g, impl) && C % 2 == 0;
  if (fused_tl >= 0 && !mfma) throw std::runtime_error("fused synthesis needs the MFMA Legendre kernel");
  if (mfma) {
    LegInvArgs a;
    a.frag = d.leg_inv_frag; a.S = S; a.Fs = Fs; a.m_local = d.m_local; a.C = C; a.full = full;
    a.vor = a.div = a.ts = a.lnps = nullptr; a.scoef = d.leg_scoef; a.dxf = dxf;
    if (fused_tl >= 0) {
      if (full) throw std::runtime_error("fused synthesis: triangular bounds only");
      a.vor = d.vors[fused_tl]; a.div = d.divs[fused_tl]; a.ts = d.ts[fused_tl]; a.lnps = d.lnps[fused_tl];
    }
    a.NKS = g.NHP / 4; a.JT = g.Jh / 16; a.NR = g.N1 + 16;
    static const int variant = env_int("ISCA_LEG_INV", 0);      // measurement switch: 100 + NW * 10 + JTG
    {
      // NW wavefronts of JTG latitude tiles: NW * JTG covers the Jh / 16 tiles when it can (T85: 4 x 1, T170: 4 x 2)
      int nw = a.JT >= 4 ? 4 : a.JT, jtg = a.JT / nw;
      if (variant >= 100) { nw = (variant - 100) / 10; jtg = (variant - 100) % 10; }
      if (nw * jtg < a.JT || jtg > 2) throw std::runtime_error("legendre_inverse: unsupported lat_max for the cooperative kernel");
      a.CB = (C + 31) / 32;
      const dim3 grid(leg_grid(g.Ml, a.CB));
#define LC(NW, JTG, WPS)                                                                                                                  \
  do {                                                                                                                                    \
    if (fused_tl >= 0) hipLaunchKernelGGL((k_leg_inv_coop<NW, JTG, true, WPS>), grid, dim3(64 * NW), 0, s, g, a TRACE_LAUNCH(trace_inv, grid.x * 2)); \
    else hipLaunchKernelGGL((k_leg_inv_coop<NW, JTG, false, WPS>), grid, dim3(64 * NW), 0, s, g, a TRACE_LAUNCH(trace_inv, grid.x * 2));  \
  } while (0)
      switch (nw * 10 + jtg) {
        case 11: LC(1, 1, 2); break;
        case 21: LC(2, 1, 2); break;
        case 41: LC(4, 1, 2); break;
        case 42: LC(4, 2, 2); break;
#ifdef ISCA_EXPERIMENTS
        case 81: LC(8, 1, 2); break;
        case 82: LC(8, 2, 2); break;
#endif
        default: throw std::runtime_error("legendre_inverse: unsupported cooperative shape");
      }
#undef LC
      TRACE_DUMP(trace_inv)
    }
  } else {
    dim3 grid((C + 63) / 64, g.Jh, g.Ml);
    hipLaunchKernelGGL(k_leg_inv_simple, grid, dim3(64), 0, s, g, d.m_local, d.p_inv, S, Fs, C, full);
  }
}

}  // namespace isca
