// gfx950 (CDNA4, wave64) kernels of the spectral dynamical core.  Written for MI355X only.
//
// Data layouts (device):
//   grid fields      [lev][lat_local][lon]            (= the reference's Fortran (lon,lat,lev))
//   Fourier buffers  [slot = q*Ml+ml][lat_local][C]   C doubles per row: 2*(level-field)+re/im; grid
//                    side indexes slot by owner (q,ml) of wavenumber m, spectral side by source rank
//                    p = j/Jl of latitude j -- the two coincide on one GPU, and differ by exactly one
//                    all-to-all otherwise (transforms.F90:970-1056).
//   spectral work    [ml][n][C]
//   spectral state   [ml][n][lev] complex
#include "kernels.h"
#include <cstdlib>

namespace isca {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cscale(double s, double2 a) { return make_double2(s * a.x, s * a.y); }
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ double2 ctimes_i(double2 a) { return make_double2(-a.y, a.x); }   // cmplx(-aimag, real)

// ---- pending fixer corrections (lazy fixers)
// The mass / energy / water corrections of compute_corrections (spectral_dynamics.F90:1213-1283) and the `future` term of the grid
// tracer's Robert filter (leapfrog_2level_B, :1484) can stay PENDING on a time level: the stored fields are the uncorrected ones and every
// reader applies the level's three scalars (pend[4*tl + 0..2] = mass factor, temperature correction, water factor; row 2 = identity) on
// the fly.  The products below must round exactly like the stored result of the materialising kernel would, whatever expression they
// feed, so they are kept out of the compiler's multiply-add contraction.
__device__ __forceinline__ double mul_nc(double a, double b) {
#pragma clang fp contract(off)
  const double r = a * b;
  return r;
}
// water factor of one value: applied where p_full >= water_correction_limit, i.e. from level `km` (of the column) down
__device__ __forceinline__ double water_corr(double q, int k, int km, double wfac) { return mul_nc(q, (k >= km) ? wfac : 1.0); }
// The column kernel keeps, per column, the number of levels above the water-correction limit of the last three steps in one word
// (byte 0: this step, byte 1: the step before, ...): a pending water factor belongs to the pressures of the step that produced the level.
__device__ __forceinline__ int kmask_byte(int word, int b) { return (word >> (8 * b)) & 0xff; }

// tables read through the constant address space stay scalar loads also behind stores that might alias (k_tracer_horiz, k_column_sig)
typedef const double __attribute__((address_space(4))) kdouble;

enum CoefId { C_EIG = 0, C_UVM, C_UVC, C_UVP, C_ALPM, C_ALPP, C_DYM, C_DX, C_DYP, C_MASK, C_DAMP, C_DAMP_VOR, C_DAMP_DIV, C_COUNT };

// =====================================================================================================
// Longitude FFT (grid_fourier.F90:129-179, fft99.F90:578-727): real <-> half-complex of length I = 2^p,
// done as one complex Stockham FFT of length I/2 per row in LDS + the even/odd split.
// One block = R rows (R consecutive level-fields at one latitude), 256 threads.
// =====================================================================================================
// ---- in-register radix-2/4/8 butterflies (INV: conjugate transform)
// First work item of a persistent FFT block.  Workgroups go round-robin over the 8 XCDs and neighbouring items (column groups gx,
// gx+1 of one latitude) touch the same 128-byte lines of the Fourier buffer (its 256-byte pieces are not line-aligned), so each
// XCD takes a contiguous run of items: the shared lines are then fetched once per L2 instead of once per block.
__device__ __forceinline__ int fft_first_item() {
  const int G = gridDim.x, b = blockIdx.x;
  return (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
}
template <bool INV> __device__ __forceinline__ double2 mul_mi(double2 a) {   // * (-i) forward, * (+i) inverse
  return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}
template <bool INV> __device__ __forceinline__ void dft4(double2 &a0, double2 &a1, double2 &a2, double2 &a3) {
  const double2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi<INV>(csub(a1, a3));
  a0 = cadd(t0, t2); a1 = cadd(t1, t3); a2 = csub(t0, t2); a3 = csub(t1, t3);
}
template <bool INV, int R> __device__ __forceinline__ void dftR(double2 (&v)[R]) {
  if constexpr (R == 2) {
    const double2 a = v[0], b = v[1];
    v[0] = cadd(a, b); v[1] = csub(a, b);
  } else if constexpr (R == 4) {
    dft4<INV>(v[0], v[1], v[2], v[3]);
  } else {
    dft4<INV>(v[0], v[2], v[4], v[6]);      // E0..E3 -> v0,v2,v4,v6
    dft4<INV>(v[1], v[3], v[5], v[7]);      // O0..O3 -> v1,v3,v5,v7
    const double h = 0.70710678118654752440;
    const double2 o1 = v[3], o3 = v[7];
    const double2 w1 = INV ? make_double2(h * (o1.x - o1.y), h * (o1.x + o1.y)) : make_double2(h * (o1.x + o1.y), h * (o1.y - o1.x));
    const double2 w2 = mul_mi<INV>(v[5]);
    const double2 w3 = INV ? make_double2(-h * (o3.x + o3.y), h * (o3.x - o3.y)) : make_double2(h * (o3.y - o3.x), -h * (o3.x + o3.y));
    const double2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], w0 = v[1];
    v[0] = cadd(e0, w0); v[4] = csub(e0, w0);
    v[1] = cadd(e1, w1); v[5] = csub(e1, w1);
    v[2] = cadd(e2, w2); v[6] = csub(e2, w2);
    v[3] = cadd(e3, w3); v[7] = csub(e3, w3);
  }
}
// LDS padding: 1 slot per 8 and one more per row (measured against no padding and 1 per 4/16/32: best of those)
__device__ __forceinline__ int fpad(int i) { return i + (i >> 3); }

// One Stockham pass of radix R at stride S over a row of NC complex points held in LDS (in place: all reads,
// barrier, all writes, barrier).  16 threads per row.  twl = LDS copy of exp(-2 pi i k / (2 NC)), k < 2 NC.
template <int NC, int R, int S, bool INV>
__device__ __forceinline__ void fft_pass(double2 *row, const double2 *twl, int tr) {
  constexpr int NB = NC / R, M = NB / S, PER = (NB + 15) / 16;
  double2 v[PER][R];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int b = tr + 16 * i;
    if (b < NB) {
      const int p = b / S, q = b - p * S;
#pragma unroll
      for (int j = 0; j < R; ++j) v[i][j] = row[fpad(q + S * (p + j * M))];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int b = tr + 16 * i;
    if (b < NB) {
      const int p = b / S, q = b - p * S;
      dftR<INV, R>(v[i]);
      row[fpad(q + S * (R * p))] = v[i][0];
#pragma unroll
      for (int j = 1; j < R; ++j) {
        double2 w = twl[2 * S * j * p];
        if (INV) w.y = -w.y;
        row[fpad(q + S * (R * p + j))] = cmul(v[i][j], w);
      }
    }
  }
  __syncthreads();
}
template <int NC, bool INV> __device__ __forceinline__ void fft_row(double2 *row, const double2 *twl, int tr) {
  if constexpr (NC == 8) { fft_pass<8, 8, 1, INV>(row, twl, tr); }
  else if constexpr (NC == 16) { fft_pass<16, 4, 1, INV>(row, twl, tr); fft_pass<16, 4, 4, INV>(row, twl, tr); }
  else if constexpr (NC == 32) { fft_pass<32, 8, 1, INV>(row, twl, tr); fft_pass<32, 4, 8, INV>(row, twl, tr); }
  else if constexpr (NC == 64) { fft_pass<64, 8, 1, INV>(row, twl, tr); fft_pass<64, 8, 8, INV>(row, twl, tr); }
  else if constexpr (NC == 128) { fft_pass<128, 8, 1, INV>(row, twl, tr); fft_pass<128, 4, 8, INV>(row, twl, tr); fft_pass<128, 4, 32, INV>(row, twl, tr); }
  else { fft_pass<256, 8, 1, INV>(row, twl, tr); fft_pass<256, 8, 8, INV>(row, twl, tr); fft_pass<256, 4, 64, INV>(row, twl, tr); }
}

// rows (level-fields) per block: 16 (256 threads) up to lon_max = 256, 8 (128 threads) at lon_max = 512; 16 threads per row
template <int NC> struct FftCfg { static constexpr int R = (NC >= 256) ? 8 : 16; };

// Both kernels are persistent: a block walks over (row group, latitude) work items gx + GX * jl with stride gridDim.x and
// requests the rows of its next item before it transforms the current one, so only the first load of a block is exposed.
template <int NC>
__global__ __launch_bounds__(FftCfg<NC>::R * 16) void k_fft_fwd(Geom g, FieldList fl, const double *__restrict__ cosm,
                                                 const int *__restrict__ slot_of_m, const double2 *__restrict__ tw,
                                                 double *__restrict__ Fg, int C, int GX) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = FftCfg<NC>::R, NT = R * 16, rs = NC + NC / 8 + 1;
  constexpr int PER = (NC + 15) / 16, PERM = (NC + 15) / 16;          // num_fourier + 1 <= NC
  double2 *buf = (double2 *)smem, *twl = buf + R * rs;
  const int t = threadIdx.x, r = t >> 4, tr = t & 15;
  for (int k = t; k < 2 * NC; k += NT) twl[k] = tw[k];
  const int rr = t % R;
  const double inv_n = 1.0 / (double)g.I;
  int slot[PERM];
#pragma unroll
  for (int i = 0; i < PERM; ++i) { const int m = t / R + 16 * i; slot[i] = slot_of_m[m < g.M1 ? m : 0]; }
  const int NG = GX * g.Jl;
  double2 zn[PER];
  double scale_n = 0.0;
  // rows of work item `item` into registers (a padding row re-reads a valid row and gets scale 0)
  auto request = [&](int item, double2 (&z)[PER], double &scale) {
    const int gx = item % GX, jl = item / GX;
    const int c = gx * R + r;
    const double2 *src = (const double2 *)(fl.g[0] + (size_t)jl * g.I);
    scale = 0.0;
    if (c < fl.ncol) {
      int f = 0;
      while (f + 1 < fl.nf && c >= fl.off[f + 1]) ++f;
      const int k = c - fl.off[f];
      src = (const double2 *)(fl.g[f] + ((size_t)k * g.Jl + jl) * g.I);
      scale = (fl.op[f] == OP_COSM) ? cosm[jl] : 1.0;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int n = tr + 16 * i; if (n < NC) z[i] = src[n]; }
  };
  int item = fft_first_item();
  if (item < NG) request(item, zn, scale_n);
  for (; item < NG; item += gridDim.x) {
    const int gx = item % GX, jl = item / GX;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int n = tr + 16 * i;
      if (n < NC) buf[r * rs + fpad(n)] = make_double2(zn[i].x * scale_n, zn[i].y * scale_n);
    }
    if (item + (int)gridDim.x < NG) request(item + gridDim.x, zn, scale_n);       // in flight during the transform
    __syncthreads();
    fft_row<NC, false>(buf + r * rs, twl, tr);
    // X[k] = E[k] + W_I^k O[k];  E = (Z[k]+conj Z[Nc-k])/2, O = -i (Z[k]-conj Z[Nc-k])/2 ; c(k) = X[k]/I
    const int cc = gx * R + rr;
#pragma unroll
    for (int i = 0; i < PERM; ++i) {
      const int m = t / R + 16 * i;
      if (m < g.M1) {
        const double2 zk = buf[rr * rs + fpad(m)];
        const double2 zc = cconj(buf[rr * rs + fpad((NC - m) & (NC - 1))]);
        const double2 e = cscale(0.5, cadd(zk, zc));
        const double2 dd = csub(zk, zc);
        const double2 o = make_double2(0.5 * dd.y, -0.5 * dd.x);
        double2 X = cadd(e, cmul(twl[m], o));
        X.x *= inv_n; X.y *= inv_n;
        if (cc < fl.ncol) *(double2 *)(Fg + ((size_t)slot[i] * g.Jl + jl) * C + 2 * cc) = X;
      }
    }
    __syncthreads();                                   // buf is rewritten by the next item
  }
}

template <int NC>
__global__ __launch_bounds__(FftCfg<NC>::R * 16) void k_fft_inv(Geom g, FieldList fl, const double *__restrict__ cosm,
                                                 const int *__restrict__ slot_of_m, const double2 *__restrict__ tw,
                                                 const double *__restrict__ Fg, int C, int GX) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = FftCfg<NC>::R, NT = R * 16, rs = NC + NC / 8 + 1;
  constexpr int PER = (NC + 15) / 16, PERM = (NC + 15) / 16;
  double2 *buf = (double2 *)smem, *twl = buf + R * rs;
  const int t = threadIdx.x, r = t >> 4, tr = t & 15;
  for (int k = t; k < 2 * NC; k += NT) twl[k] = tw[k];
  const int rr = t % R;
  int slot[PERM];
#pragma unroll
  for (int i = 0; i < PERM; ++i) { const int m = t / R + 16 * i; slot[i] = slot_of_m[m < g.M1 ? m : 0]; }
  const int NG = GX * g.Jl;
  double2 Xn[PERM];
  // truncated coefficients m = 0..M of work item `item` (wavenumbers above the truncation re-read m = 0 and are zeroed)
  auto request = [&](int item, double2 (&X)[PERM]) {
    const int gx = item % GX, jl = item / GX;
    const int ccl = min(gx * R + rr, fl.ncol - 1);
#pragma unroll
    for (int i = 0; i < PERM; ++i) X[i] = *(const double2 *)(Fg + ((size_t)slot[i] * g.Jl + jl) * C + 2 * ccl);
  };
  int item = fft_first_item();
  if (item < NG) request(item, Xn);
  for (; item < NG; item += gridDim.x) {
    const int gx = item % GX, jl = item / GX;
    {  // transforms.F90:424 zeroes everything above the truncation
      const int cc = gx * R + rr;
#pragma unroll
      for (int i = 0; i < PERM; ++i) {
        const int m = t / R + 16 * i;
        if (m < NC) {
          double2 x = (m < g.M1 && cc < fl.ncol) ? Xn[i] : make_double2(0., 0.);
          if (m == 0) x.y = 0.0;   // the real inverse FFT never references the imaginary part of the mean
          buf[rr * rs + fpad(m)] = x;
        }
      }
    }
    if (item + (int)gridDim.x < NG) request(item + gridDim.x, Xn);               // in flight during the transform
    __syncthreads();
    {  // Z'[k] = (X[k] + conj X[Nc-k]) + i conj(W^k) (X[k] - conj X[Nc-k]),  X[Nc] = 0 ; in place via registers
      double2 zz[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int k = tr + 16 * i;
        if (k < NC) {
          const double2 xk = buf[r * rs + fpad(k)];
          const double2 xc = (k == 0) ? make_double2(0., 0.) : cconj(buf[r * rs + fpad(NC - k)]);
          const double2 e = cadd(xk, xc);
          const double2 o = cmul(cconj(twl[k]), csub(xk, xc));
          zz[i] = make_double2(e.x - o.y, e.y + o.x);
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int k = tr + 16 * i;
        if (k < NC) buf[r * rs + fpad(k)] = zz[i];
      }
      __syncthreads();
    }
    fft_row<NC, true>(buf + r * rs, twl, tr);
    const int c = gx * R + r;
    if (c < fl.ncol) {
      int f = 0;
      while (f + 1 < fl.nf && c >= fl.off[f + 1]) ++f;
      const int k = c - fl.off[f];
      double2 *dst = (double2 *)(fl.g[f] + ((size_t)k * g.Jl + jl) * g.I);
      const int op = fl.op[f];
      const double scale = (op == OP_COSM) ? cosm[jl] : 1.0;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int n = tr + 16 * i;
        if (n < NC) {
          double2 z = buf[r * rs + fpad(n)];
          if (op == OP_EXP) { z.x = exp(z.x); z.y = exp(z.y); }
          else { z.x *= scale; z.y *= scale; }
          dst[n] = z;
        }
      }
    }
    __syncthreads();                                   // buf is rewritten by the next item
  }
}


// ---- lon_max = 256 / 512 (NC = 128 / 256): the same three Stockham passes (radix 8, 8 or 4, 4) with less LDS traffic and two
// block barriers per item.  The generic kernels above are LDS-bound: three in-place passes + the split make 5 writes and 6 reads of
// every row per transform.  Here a row is held by NC/8 threads with 8 elements each, element tr + TPR i in register i: that is
// the input set of the thread's first-pass butterfly (stride NC/8) and the output set of its last-pass butterflies (stride S3, one
// block per row), so the first pass runs on the registers the row was loaded into and -- inverse -- the last pass is stored from
// registers: 3 writes + 4 reads.  A row's threads sit in one wavefront, whose LDS operations execute in program order, so the
// passes need no block barrier (two per item remain, around the transposed access to the Fourier buffer).  Same butterflies, same
// twiddles, same order as fft_row: bit-identical results.
template <int NC> struct Fft3 {
  static constexpr int TPR = NC / 8, R = 256 / TPR, R2 = (NC == 128) ? 4 : 8, S3 = 8 * R2, NB2 = NC / R2, BF2 = NB2 / TPR, BF3 = (NC / 4) / TPR;
#ifndef FFT3_SWIZZLE
#define FFT3_SWIZZLE 1
#endif
  static constexpr int rs = FFT3_SWIZZLE ? NC : NC + NC / 8 + 1;
};
// Where element e of row r lives (in double2 = 16-byte slots).  Padded layout (the generic kernels', round 2): r * (NC + NC/8 + 1) + e + e/8 -- every
// READ of a row paid one extra LDS cycle (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.26-0.30; tools/dev/fft_lds_model.py reproduces 0.286: a read
// group spans elements 0-3 and 12-15 of one row and 4-11 of the next, and the padding slot at element 8 shifts one onto another).  XOR swizzle, no
// padding: the low three slot bits of e are XORed with bits 3..5 of e (the first pass's 8 lanes at a stride of 8 elements hit 8 slots) and with
// the row (the transposed phase's 8 lanes on 8 rows do too), slot bit 3 with bit 3 of the row (its 16 lanes on 16 rows): any aligned run of 16
// elements of a row still fills an aligned run of 16 slots.  Model: 0.024 forward / 0.091 inverse at lon_max = 256 (the reversed index N - k of the
// merge is a run that straddles two blocks).
template <int NC> struct FftRow {
  int base, rx, rb3;
  __device__ __forceinline__ explicit FftRow(int r) : base(r * Fft3<NC>::rs), rx(r & 7), rb3((r >> 3) & 1) {}
  __device__ __forceinline__ int at(int e) const {
    if (FFT3_SWIZZLE) return base + (e & ~15) + ((((e >> 3) ^ rb3) & 1) << 3) + ((e ^ (e >> 3) ^ rx) & 7);
    return base + fpad(e);
  }
};
template <int NC, bool INV, bool TWREG> struct FftTw {
  double2 w1[TWREG ? 8 : 1], w2[TWREG ? Fft3<NC>::BF2 : 1][TWREG ? Fft3<NC>::R2 : 1];
  const double2 *twl;
  __device__ __forceinline__ void init(const double2 *__restrict__ tw, const double2 *twl_, int tr) {
    twl = twl_;
    if constexpr (TWREG) {
#pragma unroll
      for (int j = 1; j < 8; ++j) w1[j] = get(tw, 2 * j * tr);
#pragma unroll
      for (int u = 0; u < Fft3<NC>::BF2; ++u)
#pragma unroll
        for (int j = 1; j < Fft3<NC>::R2; ++j) w2[u][j] = get(tw, 16 * j * ((tr + Fft3<NC>::TPR * u) >> 3));
    }
  }
  static __device__ __forceinline__ double2 get(const double2 *t, int k) { double2 w = t[k]; if (INV) w.y = -w.y; return w; }
  __device__ __forceinline__ double2 p1(int j, int tr) const { if constexpr (TWREG) return w1[j]; else return get(twl, 2 * j * tr); }
  __device__ __forceinline__ double2 p2(int u, int j, int p) const { if constexpr (TWREG) return w2[u][j]; else return get(twl, 16 * j * p); }
};
// z[i] = element tr + TPR i in; transform out in the same layout.  `ix` = where the row's elements live in buf.
template <int NC, bool INV, bool TWREG> __device__ __forceinline__ void fft3_row(double2 (&z)[8], double2 *buf, const FftRow<NC> &ix, const FftTw<NC, INV, TWREG> &w, int tr) {
  constexpr int TPR = Fft3<NC>::TPR, R2 = Fft3<NC>::R2, S3 = Fft3<NC>::S3, BF2 = Fft3<NC>::BF2, BF3 = Fft3<NC>::BF3;
  // pass 1: radix 8, stride 1, butterfly b = tr: inputs b + (NC/8) j = registers, outputs 8 b + j
  dftR<INV, 8>(z);
  buf[ix.at(8 * tr)] = z[0];
#pragma unroll
  for (int j = 1; j < 8; ++j) buf[ix.at(8 * tr + j)] = cmul(z[j], w.p1(j, tr));
  __builtin_amdgcn_wave_barrier();
  {  // pass 2: radix R2, stride 8, in place
    double2 v[BF2][R2];
    const int q = tr & 7;
#pragma unroll
    for (int u = 0; u < BF2; ++u) {
      const int p = (tr + TPR * u) >> 3;
#pragma unroll
      for (int j = 0; j < R2; ++j) v[u][j] = buf[ix.at(q + 8 * (p + (NC / (8 * R2)) * j))];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < BF2; ++u) {
      const int p = (tr + TPR * u) >> 3;
      dftR<INV, R2>(v[u]);
      buf[ix.at(q + 8 * (R2 * p))] = v[u][0];
#pragma unroll
      for (int j = 1; j < R2; ++j) buf[ix.at(q + 8 * (R2 * p + j))] = cmul(v[u][j], w.p2(u, j, p));
    }
  }
  __builtin_amdgcn_wave_barrier();
  // pass 3: radix 4, stride S3, one block: butterfly q = tr + TPR u, inputs / outputs q + S3 j = registers u + BF3 j; twiddle twl[0] = 1
  const double2 one = make_double2(1.0, INV ? -0.0 : 0.0);
#pragma unroll
  for (int u = 0; u < BF3; ++u) {
    double2 v[4];
    const int q = tr + TPR * u;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = buf[ix.at(q + S3 * j)];
    dftR<INV, 4>(v);
    z[u] = v[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) z[u + BF3 * j] = cmul(v[j], one);
  }
  __builtin_amdgcn_wave_barrier();
}

template <int NC, bool TWREG>
__global__ __launch_bounds__(256) void k_fft_fwd3(Geom g, FieldList fl, const double *__restrict__ cosm, const int *__restrict__ slot_of_m,
                                                  const double2 *__restrict__ tw, double *__restrict__ Fg, int C, int GX) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TPR = Fft3<NC>::TPR, R = Fft3<NC>::R, rs = Fft3<NC>::rs;
  double2 *buf = (double2 *)smem, *twl = buf + R * rs;
  int *slot = (int *)(twl + 2 * NC);                    // Fourier-buffer slot of wavenumber m (above the truncation: that of m = 0)
  const int t = threadIdx.x, r = t / TPR, tr = t % TPR;
  for (int k = t; k < 2 * NC; k += 256) twl[k] = tw[k];
  for (int m = t; m < NC; m += 256) slot[m] = slot_of_m[m < g.M1 ? m : 0];
  FftTw<NC, false, TWREG> w;
  w.init(tw, twl, tr);
  const int rr = t % R;
  const double inv_n = 1.0 / (double)g.I;
  const int NG = GX * g.Jl;
  double2 zn[8];
  double scale_n = 0.0;
  auto request = [&](int item, double2 (&z)[8], double &scale) {         // a padding row re-reads a valid row and gets scale 0
    const int gx = item % GX, jl = item / GX;
    const int c = gx * R + r;
    const double2 *src = (const double2 *)(fl.g[0] + (size_t)jl * g.I);
    scale = 0.0;
    if (c < fl.ncol) {
      int f = 0;
      while (f + 1 < fl.nf && c >= fl.off[f + 1]) ++f;
      const int k = c - fl.off[f];
      src = (const double2 *)(fl.g[f] + ((size_t)k * g.Jl + jl) * g.I);
      scale = (fl.op[f] == OP_COSM) ? cosm[jl] : 1.0;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = src[tr + TPR * i];
  };
  const FftRow<NC> ix(r), ixr(rr);
  int item = fft_first_item();
  if (item < NG) request(item, zn, scale_n);
  __syncthreads();                                     // twl, slot
  for (; item < NG; item += gridDim.x) {
    const int gx = item % GX, jl = item / GX;
    double2 z[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = make_double2(zn[i].x * scale_n, zn[i].y * scale_n);
    if (item + (int)gridDim.x < NG) request(item + gridDim.x, zn, scale_n);       // in flight during the transform
    fft3_row<NC, false, TWREG>(z, buf, ix, w, tr);
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[ix.at(tr + TPR * i)] = z[i];
    __syncthreads();
    // X[k] = E[k] + W_I^k O[k];  E = (Z[k]+conj Z[Nc-k])/2, O = -i (Z[k]-conj Z[Nc-k])/2 ; c(k) = X[k]/I
    const int cc = gx * R + rr;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = t / R + TPR * i;
      if (m < g.M1) {
        const double2 zk = buf[ixr.at(m)];
        const double2 zc = cconj(buf[ixr.at((NC - m) & (NC - 1))]);
        const double2 e = cscale(0.5, cadd(zk, zc));
        const double2 dd = csub(zk, zc);
        const double2 o = make_double2(0.5 * dd.y, -0.5 * dd.x);
        double2 X = cadd(e, cmul(twl[m], o));
        X.x *= inv_n; X.y *= inv_n;
        if (cc < fl.ncol) *(double2 *)(Fg + ((size_t)slot[m] * g.Jl + jl) * C + 2 * cc) = X;
      }
    }
    __syncthreads();                                   // the rows are rewritten by the next item
  }
}

// row c of a field list -> (field, level); FieldList::il_a / il_b: two fields whose rows alternate
__device__ __forceinline__ int fl_row(const FieldList &fl, int c, int &k) {
  if (fl.il_a >= 0) {
    const int base = fl.off[fl.il_a], d = c - base;
    if (d >= 0 && d < 2 * fl.nlev[fl.il_a]) { k = d >> 1; return (d & 1) ? fl.il_b : fl.il_a; }
  }
  int f = 0;
  while (f + 1 < fl.nf && c >= fl.off[f + 1]) ++f;
  k = c - fl.off[f];
  return f;
}
template <int NC, bool TWREG>
__global__ __launch_bounds__(256) void k_fft_inv3(Geom g, FieldList fl, const double *__restrict__ cosm, const int *__restrict__ slot_of_m,
                                                  const double2 *__restrict__ tw, const double *__restrict__ Fg, int C, int GX) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TPR = Fft3<NC>::TPR, R = Fft3<NC>::R, rs = Fft3<NC>::rs;
  double2 *buf = (double2 *)smem, *twl = buf + R * rs;
  int *slot = (int *)(twl + 2 * NC);
  const int t = threadIdx.x, r = t / TPR, tr = t % TPR;
  for (int k = t; k < 2 * NC; k += 256) twl[k] = tw[k];
  for (int m = t; m < NC; m += 256) slot[m] = slot_of_m[m < g.M1 ? m : 0];
  int *rowc = slot + NC;                               // [ncol] row of the field list -> buffer column | x-derivative flag << 30 (FieldList::nbuf)
  if (fl.nbuf)
    for (int c = t; c < fl.ncol; c += 256) {
      int k;
      const int f = fl_row(fl, c, k);
      rowc[c] = (fl.boff[f] + k) | (fl.dx[f] ? 1 << 30 : 0);
    }
  FftTw<NC, true, TWREG> w;
  w.init(tw, twl, tr);
  const int rr = t % R;
  const int NG = GX * g.Jl;
  double2 Xn[8];
  __syncthreads();                                     // twl, slot, rowc
  double dxn = -1.0;                                   // >= 0: the requested row is an x-derivative, i m dxn times the rows of the field it was read from
  auto request = [&](int item, double2 (&X)[8], double &dxs) {      // wavenumbers above the truncation re-read m = 0 and are zeroed
    const int gx = item % GX, jl = item / GX;
    int ccl = min(gx * R + rr, fl.ncol - 1);
    dxs = -1.0;
    if (fl.nbuf) {                                     // the row's field decides which buffer column holds its coefficients (table built below)
      const int rc = rowc[ccl];
      ccl = rc & 0x3fffffff;
      if (rc >> 30) dxs = fl.dxfac;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) X[i] = *(const double2 *)(Fg + ((size_t)slot[t / R + TPR * i] * g.Jl + jl) * C + 2 * ccl);
  };
  const FftRow<NC> ix(r), ixr(rr);
  int item = fft_first_item();
  if (item < NG) request(item, Xn, dxn);
  for (; item < NG; item += gridDim.x) {
    const int gx = item % GX, jl = item / GX;
    {  // transforms.F90:424 zeroes everything above the truncation
      const int cc = gx * R + rr;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = t / R + TPR * i;
        double2 x = (m < g.M1 && cc < fl.ncol) ? Xn[i] : make_double2(0., 0.);
        if (dxn >= 0.0) { const double dm = dxn * (double)m; x = make_double2(-dm * x.y, dm * x.x); }      // i m / a times the coefficient
        if (m == 0) x.y = 0.0;   // the real inverse FFT never references the imaginary part of the mean
        buf[ixr.at(m)] = x;
      }
    }
    if (item + (int)gridDim.x < NG) request(item + gridDim.x, Xn, dxn);          // in flight during the transform
    __syncthreads();
    double2 z[8];
    // Z'[k] = (X[k] + conj X[Nc-k]) + i conj(W^k) (X[k] - conj X[Nc-k]),  X[Nc] = 0
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = tr + TPR * i;
      const double2 xk = buf[ix.at(k)];
      const double2 xc = (k == 0) ? make_double2(0., 0.) : cconj(buf[ix.at(NC - k)]);
      const double2 e = cadd(xk, xc);
      const double2 o = cmul(cconj(twl[k]), csub(xk, xc));
      z[i] = make_double2(e.x - o.y, e.y + o.x);
    }
    __builtin_amdgcn_wave_barrier();
    fft3_row<NC, true, TWREG>(z, buf, ix, w, tr);
    const int c = gx * R + r;
    if (c < fl.ncol) {
      int k;
      const int f = fl_row(fl, c, k);
      double2 *dst = (double2 *)(fl.g[f] + ((size_t)k * g.Jl + jl) * g.I);
      const int op = fl.op[f];
      const double scale = (op == OP_COSM) ? cosm[jl] : 1.0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        double2 v = z[i];
        if (op == OP_EXP) { v.x = exp(v.x); v.y = exp(v.y); }
        else { v.x *= scale; v.y *= scale; }
        dst[tr + TPR * i] = v;
      }
    }
    __syncthreads();                                   // the rows are rewritten by the next item
  }
}

#ifdef ISCA_EXPERIMENTS      // (round 4: correct, 45 MB less traffic, 1.7x slower than the two kernels it replaces -- HISTORY.md)
// =====================================================================================================
// Fused analysis: longitude FFT + Legendre transform of the step's forward batch in ONE kernel -- the truncated Fourier rows
// (grid_fourier.F90:129-179 -> spherical_fourier.F90:264-339) never leave the chip.  One rank (no lat <-> m exchange to feed), lon_max = 256.
//
// A block owns TWO level-fields (4 real columns) for ALL latitudes and ALL wavenumbers, because the quadrature sums of every (m, n) must stay
// in accumulators while the block walks the latitudes: 8 wavefronts in two roles, pipelined over chunks of 8 latitude pairs --
//   wavefronts 0-3 (FFT):  a 16-thread group per (level-field, pair): loads the northern and the southern row, forms x_even = N + S and
//      x_odd = N - S on the GRID values (the FFT is linear: the fold of spherical_fourier.F90:311-312 done in front of it), transforms both
//      rows (the same three Stockham passes as k_fft_fwd3, wavefront-synchronous), splits, and leaves the coefficients m <= M in LDS as
//      E/O[pair][parity][m][column];
//   wavefronts 4-7 (MFMA): v_mfma_f64_4x4x4_4b_f64 -- four independent 4x4x4 products per instruction, here 8 even + 8 odd n of one
//      wavenumber x 4 latitude pairs x 4 columns (lane maps found with tools/micro/mfma_f64_probe.hip: A[b][i][k] lane 16k+4b+i,
//      B[b][k][j] lane 16k+4b+j, D[b][i][j] lane 16i+4b+j); a 16-column tile would be three quarters empty.  Each wavefront owns NT tiles
//      (16 n of one m; the host deals the triangle's tiles round-robin) with their accumulators in registers for the whole kernel, streams
//      its part of the fragment-ordered table (16-byte loads: the pieces of a chunk's two k-steps side by side) through a ring, and reads
//      B from the E/O buffer the FFT wavefronts filled during the chunk before.
// One block barrier per chunk.  Algorithmic bytes: the grid rows in, the spectral rows out; the table (2.4 MB at T85) comes from L2.
// =====================================================================================================
constexpr int FZ_PC = 8;            // latitude pairs per chunk (two k-steps of 4)
struct FusedFwdArgs {
  const double *frag;               // [4][NCH][NT][64][2]
  const int *desc;                  // [4][NT][2]: {ml (-1: padding), 16 * tile | nlim << 16}
  double *S;                        // [Ml][N1][C]
  int C, NCH, MP;                   // MP: odd wavenumber pitch of the E/O buffers (bank spread)
  int dbg;                          // measurement only (ISCA_FZ_DBG): 1 the FFT wavefronts, 2 the MFMA wavefronts skip their work (wrong results)
};
template <int NT>
__global__ __launch_bounds__(512, 2) void k_fft_leg_fwd(Geom g, FieldList fl, const double *__restrict__ cosm, const double2 *__restrict__ tw,
                                                        FusedFwdArgs a) {
  constexpr int NC = 128, TPR = 16, PF = 8;
  static_assert(NT % PF == 0, "ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double2 *buf = (double2 *)smem;                          // [16 rows][NC]: the FFT groups' rows (FftRow swizzle)
  double2 *twl = buf + 16 * NC;                            // [2 NC] exp(-2 pi i k / I)
  double *eo = (double *)(twl + 2 * NC);                   // [2 buffers][FZ_PC][2 parities][MP][4 columns]
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  for (int k = t; k < 2 * NC; k += 512) twl[k] = tw[k];
  const int MP = a.MP, NCH = a.NCH, J = g.J, I = g.I;
  const int EOB = FZ_PC * 2 * MP * 4;                      // doubles per E/O buffer
  __syncthreads();
  if (wave < 4) {
    // ------------------------------------------------------------------ FFT role
    const int r = t >> 4, tr = t & 15, lfl = r >> 3, jpl = r & 7;
    const int c = blockIdx.x * 2 + lfl;
    const double *base = fl.g[0];
    int op = OP_NONE;
    bool valid = false;
    if (c < fl.ncol) {
      int f = 0;
      while (f + 1 < fl.nf && c >= fl.off[f + 1]) ++f;
      base = fl.g[f] + (size_t)(c - fl.off[f]) * J * I;
      op = fl.op[f];
      valid = true;
    }
    FftTw<NC, false, true> w;
    w.init(tw, twl, tr);
    const FftRow<NC> ix(r);
    const double inv_n = 1.0 / (double)I;
    double2 zn[8], zs[8];
    double sn = 0.0, ss = 0.0;
    auto request = [&](int ch) {                           // the pair's two rows (a padding column re-reads valid rows and gets scale 0)
      const int jp = ch * FZ_PC + jpl;
      const double2 *ps = (const double2 *)(base + (size_t)jp * I), *pn = (const double2 *)(base + (size_t)(J - 1 - jp) * I);
#pragma unroll
      for (int i = 0; i < 8; ++i) { zs[i] = ps[tr + TPR * i]; zn[i] = pn[tr + TPR * i]; }
      ss = valid ? ((op == OP_COSM) ? cosm[jp] : 1.0) : 0.0;
      sn = valid ? ((op == OP_COSM) ? cosm[J - 1 - jp] : 1.0) : 0.0;
    };
    request(0);
    for (int ch = 0; ch <= NCH; ++ch) {
      if (ch < NCH && !(a.dbg & 1)) {
        double *dst = eo + (ch & 1) * EOB + (size_t)(jpl * 2) * MP * 4 + 2 * lfl;
        const double sgn_n = sn, sgn_s = ss;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
          double2 z[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const double2 vn = make_double2(zn[i].x * sgn_n, zn[i].y * sgn_n), vs = make_double2(zs[i].x * sgn_s, zs[i].y * sgn_s);
            // x_even = north + south, x_odd = north - south   (spherical_fourier.F90:311-312)
            z[i] = par ? make_double2(vn.x - vs.x, vn.y - vs.y) : make_double2(vn.x + vs.x, vn.y + vs.y);
          }
          if (par == 1 && ch + 1 < NCH) request(ch + 1);           // the next chunk's rows: in flight during the second transform and the wait at the barrier
          fft3_row<NC, false, true>(z, buf, ix, w, tr);
#pragma unroll
          for (int i = 0; i < 8; ++i) buf[ix.at(tr + TPR * i)] = z[i];
          __builtin_amdgcn_wave_barrier();
          // X[k] = E[k] + W_I^k O[k];  E = (Z[k]+conj Z[Nc-k])/2, O = -i (Z[k]-conj Z[Nc-k])/2 ; c(k) = X[k]/I   (as k_fft_fwd3)
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const int m = tr + TPR * i;
            if (m < g.M1) {
              const double2 zk = buf[ix.at(m)];
              const double2 zc = cconj(buf[ix.at((NC - m) & (NC - 1))]);
              const double2 e = cscale(0.5, cadd(zk, zc));
              const double2 dd = csub(zk, zc);
              const double2 o = make_double2(0.5 * dd.y, -0.5 * dd.x);
              double2 X = cadd(e, cmul(twl[m], o));
              X.x *= inv_n; X.y *= inv_n;
              *(double2 *)(dst + ((size_t)par * MP + m) * 4) = X;
            }
          }
          __builtin_amdgcn_wave_barrier();                         // the row is rewritten by the next transform
        }
      }
      __syncthreads();                                             // chunk ch is in E/O[ch & 1]; chunk ch - 1 has been consumed
    }
  } else {
    // ------------------------------------------------------------------ MFMA role
    const int wv = wave - 4;
    const int k = lane >> 4, b = (lane >> 2) & 3, j = lane & 3, par = b >> 1;
    const int boff = ((k * 2 + par) * MP) * 4 + j;                 // + 32 MP per k-step of the chunk, + EOB per buffer, + 4 m
    const int *dsc = a.desc + (size_t)wv * NT * 2;
    const double2 *fr = (const double2 *)a.frag + (size_t)wv * NCH * NT * 64 + lane;
    const int total = NCH * NT;
    double acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = 0.0;
    double2 ring[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) ring[d] = fr[(size_t)d * 64];
    // where a tile's wavenumber sits in the E/O rows (4 ml doubles), two tiles per register: a descriptor read inside the loop would put a
    // scalar-memory round trip in front of every LDS read (measured: 75 us for the kernel instead of 25)
    unsigned mo[NT / 2];
#pragma unroll
    for (int q = 0; q < NT / 2; ++q) mo[q] = (unsigned)(4 * max(dsc[4 * q], 0)) | ((unsigned)(4 * max(dsc[4 * q + 2], 0)) << 16);
    __syncthreads();                                               // (the FFT wavefronts' barrier of chunk 0)
    for (int ch = 0; ch < NCH; ++ch) {
      const double *eb = eo + (ch & 1) * EOB + boff;
      if (!(a.dbg & 2))
#pragma unroll
      for (int gq = 0; gq < NT / PF; ++gq) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
          const int q = gq * PF + d;
          const double *bp = eb + ((mo[q >> 1] >> (16 * (q & 1))) & 0xffffu);
          const double b0 = bp[0], b1 = bp[32 * MP];
          const double2 av = ring[d];
          const int nxt = min(ch * NT + q + PF, total - 1);        // clamped at the end: re-reads, no branch around the load
          ring[d] = fr[(size_t)nxt * 64];
          acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(av.x, b0, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(av.y, b1, acc[q], 0, 0, 0);
        }
      }
      __syncthreads();
    }
    // D[b][i][jj] sits in lane 16 i + 4 b + jj: n = 16 tile + 2 (4 (b & 1) + i) + (b >> 1), column 4 blockIdx.x + jj
    const int iD = lane >> 4, e = 4 * (b & 1) + iD;
    double *Sp = a.S + 4 * blockIdx.x + j;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int ml = dsc[2 * q], w1 = dsc[2 * q + 1];
      const int n = (w1 & 0xffff) + 2 * e + par, nlim = w1 >> 16;
      if (ml >= 0 && n < nlim) Sp[((size_t)ml * g.N1 + n) * a.C] = acc[q];
    }
  }
}
bool fused_forward_ok(const Geom &g) { return g.P == 1 && g.I == 256 && g.Jh % FZ_PC == 0 && g.M1 <= 96; }
void launch_fft_legendre_forward(const Geom &g, const Dev &d, const FieldList &fl, double *S, hipStream_t s) {
  FusedFwdArgs a;
  a.frag = d.fz_frag; a.desc = d.fz_desc; a.S = S; a.C = col_pitch(fl.ncol); a.NCH = g.Jh / FZ_PC; a.MP = g.M1 | 1;
  static const int dbg = exp_env("ISCA_FZ_DBG") ? atoi(exp_env("ISCA_FZ_DBG")) : 0;
  a.dbg = dbg;
  const size_t lds = (size_t)(16 * 128 + 2 * 128) * sizeof(double2) + (size_t)2 * FZ_PC * 2 * a.MP * 4 * sizeof(double);
  const dim3 grid((unsigned)((fl.ncol + 1) / 2));
#define LZ(N)                                                                                                              \
  do {                                                                                                                     \
    static bool attr = false;                                                                                              \
    if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void *)k_fft_leg_fwd<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; } \
    hipLaunchKernelGGL(k_fft_leg_fwd<N>, grid, dim3(512), lds, s, g, fl, d.cosm_lat_l, (const double2 *)d.tw, a);          \
  } while (0)
  if (d.fz_NT == 24) LZ(24); else if (d.fz_NT == 48) LZ(48); else if (d.fz_NT == 72) LZ(72);
  else throw std::runtime_error("fused analysis: unsupported tile count");
#undef LZ
}
#endif  // ISCA_EXPERIMENTS

// ---- lon_max = 2^a 3^b 5^c that is not a power of two (the lengths fft99's set99 factors, fft99.F90:83-120; its radix-3 / radix-5 passes
// :876-1228): the same real <-> half-complex scheme -- one complex Stockham transform of length NC = I/2 and the even/odd split -- with the
// factor list at run time: passes of radix 4, 2, 3, 5 out of place between two LDS copies of the row (the Stockham form proper: no
// in-register staging, so the pass loop needs no compile-time bounds).  Not a tuned path (the benchmark resolutions are powers of two):
// 16 threads per row, 16 / 8 / 4 rows per block (what fits 64 KB of LDS), one work item per block.
struct FftMixed { int nf, radix[12]; };
template <bool INV> __device__ __forceinline__ void dft3(double2 &a0, double2 &a1, double2 &a2) {
  const double s3 = INV ? 0.86602540378443864676 : -0.86602540378443864676;     // sin(-+2 pi / 3)
  const double2 t1 = cadd(a1, a2), t2 = make_double2(a0.x - 0.5 * t1.x, a0.y - 0.5 * t1.y);
  const double2 d = csub(a1, a2), t3 = make_double2(-s3 * d.y, s3 * d.x);          // i s3 (a1 - a2)
  a0 = cadd(a0, t1); a1 = cadd(t2, t3); a2 = csub(t2, t3);
}
template <bool INV> __device__ __forceinline__ void dft5(double2 (&v)[5]) {
  const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;         // cos(2 pi / 5), cos(4 pi / 5)
  const double s1 = INV ? 0.95105651629515357212 : -0.95105651629515357212;       // sin(-+2 pi / 5)
  const double s2 = INV ? 0.58778525229247312917 : -0.58778525229247312917;       // sin(-+4 pi / 5)
  const double2 a = cadd(v[1], v[4]), b = cadd(v[2], v[3]), c = csub(v[1], v[4]), d = csub(v[2], v[3]);
  const double2 x0 = v[0];
  const double2 p1 = make_double2(x0.x + c1 * a.x + c2 * b.x, x0.y + c1 * a.y + c2 * b.y);
  const double2 p2 = make_double2(x0.x + c2 * a.x + c1 * b.x, x0.y + c2 * a.y + c1 * b.y);
  const double2 q1 = make_double2(-(s1 * c.y + s2 * d.y), s1 * c.x + s2 * d.x);    // i (s1 c + s2 d)
  const double2 q2 = make_double2(-(s2 * c.y - s1 * d.y), s2 * c.x - s1 * d.x);    // i (s2 c - s1 d)
  v[0] = cadd(x0, cadd(a, b));
  v[1] = cadd(p1, q1); v[4] = csub(p1, q1);
  v[2] = cadd(p2, q2); v[3] = csub(p2, q2);
}
// one pass of radix R at stride S: y[q + S (R p + j)] = w^(S j p) DFT_R(x[q + S (p + M j)])_j, M = NC / (R S), w = exp(-+2 pi i / NC); tw[k] = exp(-2 pi i k / (2 NC))
template <bool INV, int R> __device__ __forceinline__ void mixed_pass(const double2 *x, double2 *y, const double2 *twl, int NC, int S, int tr) {
  const int NB = NC / R, M = NB / S;
  for (int b = tr; b < NB; b += 16) {
    const int p = b / S, q = b - p * S;
    double2 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = x[q + S * (p + j * M)];
    if constexpr (R == 3) dft3<INV>(v[0], v[1], v[2]);
    else if constexpr (R == 5) dft5<INV>(v);
    else dftR<INV, R>(v);
    y[q + S * (R * p)] = v[0];
#pragma unroll
    for (int j = 1; j < R; ++j) {
      double2 w = twl[2 * S * j * p];
      if (INV) w.y = -w.y;
      y[q + S * (R * p + j)] = cmul(v[j], w);
    }
  }
}
// all passes of a row; returns the buffer that holds the result
template <bool INV> __device__ __forceinline__ double2 *mixed_row(double2 *x, double2 *y, const double2 *twl, int NC, const FftMixed &fm, int tr) {
  int S = 1;
  for (int f = 0; f < fm.nf; ++f) {
    const int R = fm.radix[f];
    if (R == 4) mixed_pass<INV, 4>(x, y, twl, NC, S, tr);
    else if (R == 2) mixed_pass<INV, 2>(x, y, twl, NC, S, tr);
    else if (R == 3) mixed_pass<INV, 3>(x, y, twl, NC, S, tr);
    else mixed_pass<INV, 5>(x, y, twl, NC, S, tr);
    S *= R;
    __syncthreads();
    double2 *t = x; x = y; y = t;
  }
  return x;
}
static int fft_mixed_rows(int NC) { return NC <= 112 ? 16 : (NC <= 224 ? 8 : 4); }
__global__ __launch_bounds__(256) void k_fft_fwd_mixed(Geom g, FieldList fl, FftMixed fm, const double *__restrict__ cosm, const int *__restrict__ slot_of_m,
                                                                const double2 *__restrict__ tw, double *__restrict__ Fg, int C, int GX, int R) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NC = g.I / 2, NT = R * 16;
  double2 *bufa = (double2 *)smem, *bufb = bufa + R * NC, *twl = bufb + R * NC;
  const int t = threadIdx.x, r = t >> 4, tr = t & 15, rr = t % R;
  for (int k = t; k < 2 * NC; k += NT) twl[k] = tw[k];
  const double inv_n = 1.0 / (double)g.I;
  const int item = blockIdx.x, gx = item % GX, jl = item / GX;
  {
    const int c = gx * R + r;
    const double2 *src = (const double2 *)(fl.g[0] + (size_t)jl * g.I);
    double scale = 0.0;                     // a padding row re-reads a valid row and gets scale 0
    if (c < fl.ncol) {
      int f = 0;
      while (f + 1 < fl.nf && c >= fl.off[f + 1]) ++f;
      src = (const double2 *)(fl.g[f] + ((size_t)(c - fl.off[f]) * g.Jl + jl) * g.I);
      scale = (fl.op[f] == OP_COSM) ? cosm[jl] : 1.0;
    }
    for (int n = tr; n < NC; n += 16) { const double2 z = src[n]; bufa[r * NC + n] = make_double2(z.x * scale, z.y * scale); }
  }
  __syncthreads();
  const double2 *res = mixed_row<false>(bufa + r * NC, bufb + r * NC, twl, NC, fm, tr) - r * NC;      // (the same buffer for every row: nf is uniform)
  // X[k] = E[k] + W_I^k O[k];  E = (Z[k]+conj Z[Nc-k])/2, O = -i (Z[k]-conj Z[Nc-k])/2 ; c(k) = X[k]/I   (thread = (row rr, wavenumber m): Fourier rows are column-contiguous)
  const int cc = gx * R + rr;
  for (int m = t / R; m < g.M1; m += 16) {
    const double2 zk = res[rr * NC + m];
    const double2 zc = cconj(res[rr * NC + (m == 0 ? 0 : NC - m)]);
    const double2 e = cscale(0.5, cadd(zk, zc));
    const double2 dd = csub(zk, zc);
    const double2 o = make_double2(0.5 * dd.y, -0.5 * dd.x);
    double2 X = cadd(e, cmul(twl[m], o));
    X.x *= inv_n; X.y *= inv_n;
    if (cc < fl.ncol) *(double2 *)(Fg + ((size_t)slot_of_m[m] * g.Jl + jl) * C + 2 * cc) = X;
  }
}
__global__ __launch_bounds__(256) void k_fft_inv_mixed(Geom g, FieldList fl, FftMixed fm, const double *__restrict__ cosm, const int *__restrict__ slot_of_m,
                                                                const double2 *__restrict__ tw, const double *__restrict__ Fg, int C, int GX, int R) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NC = g.I / 2, NT = R * 16;
  double2 *bufa = (double2 *)smem, *bufb = bufa + R * NC, *twl = bufb + R * NC;
  const int t = threadIdx.x, r = t >> 4, tr = t & 15, rr = t % R;
  for (int k = t; k < 2 * NC; k += NT) twl[k] = tw[k];
  const int item = blockIdx.x, gx = item % GX, jl = item / GX;
  {  // truncated coefficients m = 0..M; transforms.F90:424 zeroes everything above the truncation
    const int cc = gx * R + rr, ccl = min(cc, fl.ncol - 1);
    for (int m = t / R; m < NC; m += 16) {
      double2 x = make_double2(0., 0.);
      if (m < g.M1 && cc < fl.ncol) x = *(const double2 *)(Fg + ((size_t)slot_of_m[m] * g.Jl + jl) * C + 2 * ccl);
      if (m == 0) x.y = 0.0;          // the real inverse FFT never references the imaginary part of the mean
      bufa[rr * NC + m] = x;
    }
  }
  __syncthreads();
  // Z'[k] = (X[k] + conj X[Nc-k]) + i conj(W^k) (X[k] - conj X[Nc-k]),  X[Nc] = 0
  for (int k = tr; k < NC; k += 16) {
    const double2 xk = bufa[r * NC + k];
    const double2 xc = (k == 0) ? make_double2(0., 0.) : cconj(bufa[r * NC + NC - k]);
    const double2 e = cadd(xk, xc);
    const double2 o = cmul(cconj(twl[k]), csub(xk, xc));
    bufb[r * NC + k] = make_double2(e.x - o.y, e.y + o.x);
  }
  __syncthreads();
  const double2 *res = mixed_row<true>(bufb + r * NC, bufa + r * NC, twl, NC, fm, tr);
  const int c = gx * R + r;
  if (c < fl.ncol) {
    int f = 0;
    while (f + 1 < fl.nf && c >= fl.off[f + 1]) ++f;
    double2 *dst = (double2 *)(fl.g[f] + ((size_t)(c - fl.off[f]) * g.Jl + jl) * g.I);
    const int op = fl.op[f];
    const double scale = (op == OP_COSM) ? cosm[jl] : 1.0;
    for (int n = tr; n < NC; n += 16) {
      double2 z = res[n];
      if (op == OP_EXP) { z.x = exp(z.x); z.y = exp(z.y); }
      else { z.x *= scale; z.y *= scale; }
      dst[n] = z;
    }
  }
}
// the factors of NC = lon_max / 2: fours first, then a two, then threes and fives (any order gives the transform; larger strides last keep the
// early passes' accesses contiguous); false when a prime factor above 5 is left (set99 refuses those lengths too)
bool fft_mixed_factors(int NC, FftMixed &fm) {
  fm.nf = 0;
  int n = NC;
  while (n % 4 == 0) { fm.radix[fm.nf++] = 4; n /= 4; }
  if (n % 2 == 0) { fm.radix[fm.nf++] = 2; n /= 2; }
  while (n % 3 == 0) { fm.radix[fm.nf++] = 3; n /= 3; }
  while (n % 5 == 0) { fm.radix[fm.nf++] = 5; n /= 5; }
  return n == 1 && fm.nf <= 12;
}
static void launch_fft_mixed(bool inverse, const Geom &g, const Dev &d, const FieldList &fl, double *Fg, hipStream_t s) {
  FftMixed fm;
  const int NC = g.I / 2;
  if ((g.I & 1) || !fft_mixed_factors(NC, fm)) throw std::runtime_error("fft: lon_max must be even with no prime factor above 5 (fft99.F90:83-120)");
  const int R = fft_mixed_rows(NC);
  const int C = col_pitch(fl.ncol), GX = (fl.ncol + R - 1) / R;
  const size_t lds = (size_t)(2 * R * NC + 2 * NC) * sizeof(double2);
  if (inverse) hipLaunchKernelGGL(k_fft_inv_mixed, dim3((unsigned)(GX * g.Jl)), dim3(R * 16), lds, s, g, fl, fm, d.cosm_lat_l, d.slot_of_m, (const double2 *)d.tw, (const double *)Fg, C, GX, R);
  else hipLaunchKernelGGL(k_fft_fwd_mixed, dim3((unsigned)(GX * g.Jl)), dim3(R * 16), lds, s, g, fl, fm, d.cosm_lat_l, d.slot_of_m, (const double2 *)d.tw, Fg, C, GX, R);
}

static int fft_rows(int NC) { return NC >= 256 ? 8 : 16; }
static unsigned fft_grid(int items) {                  // persistent blocks: at most 3 per CU of the 256 (LDS-limited residency; measured
  const int cap = 768;                                 // against 512 / 1024 / 1536), and the same number of items for every block
  const int rounds = (items + cap - 1) / cap;
  return (unsigned)((items + rounds - 1) / rounds);
}
static bool fft_old() { static const bool v = exp_env("ISCA_FFT_OLD") != nullptr; return v; }   // experiment: the generic LDS passes at lon_max >= 256
static bool fft_twreg() { static const bool v = exp_env("ISCA_FFT_TWLDS") == nullptr; return v; }    // pass twiddles in registers (default) or, experiment, read from LDS
static size_t fft3_lds_bytes(int NC) { return (size_t)(fft_rows(NC) * (NC + NC / 8 + 1) + 2 * NC) * sizeof(double2) + (NC + 7 * ISCA_MAX_LEVELS + 3) * sizeof(int); }      // rows, twiddles, slot of m, k_fft_inv3's row table
static dim3 fft3_grid(int items) {                     // persistent blocks of 256 threads: 2 per CU (~196 VGPRs; measured against 3 and 4 per CU)
  static const int cap = exp_env("ISCA_FFT_CAP") ? atoi(exp_env("ISCA_FFT_CAP")) : 512;
  const int rounds = (items + cap - 1) / cap;
  // a multiple of 8 blocks: fft_first_item then gives each XCD a contiguous run of items (neighbouring items share 128-byte lines of the Fourier buffer);
  // with any other count it falls back to item = block, i.e. neighbours on different L2s.  (ISCA_FFT_G8=0: the count as it was, for measurements.)
  static const bool g8 = !(exp_env("ISCA_FFT_G8") && atoi(exp_env("ISCA_FFT_G8")) == 0);
  const int G = (items + rounds - 1) / rounds;
  return dim3((unsigned)(g8 ? (G + 7) / 8 * 8 : G));
}
static size_t fft_lds_bytes(int NC) { return (size_t)(fft_rows(NC) * (NC + NC / 8 + 1) + 2 * NC) * sizeof(double2); }

void launch_fft_forward(const Geom &g, const Dev &d, const FieldList &fl, double *Fg, hipStream_t s) {
  const int C = col_pitch(fl.ncol), NC = g.I / 2;
  const int R = fft_rows(NC);
  const int GX = (fl.ncol + R - 1) / R;
  dim3 grid(fft_grid(GX * g.Jl));
  const size_t lds = fft_lds_bytes(NC);
#define LF3(N, TW) hipLaunchKernelGGL((k_fft_fwd3<N, TW>), fft3_grid(GX * g.Jl), dim3(256), fft3_lds_bytes(NC), s, g, fl, d.cosm_lat_l, d.slot_of_m, (const double2 *)d.tw, Fg, C, GX)
#define LF(N) hipLaunchKernelGGL(k_fft_fwd<N>, grid, dim3(R * 16), lds, s, g, fl, d.cosm_lat_l, d.slot_of_m, (const double2 *)d.tw, Fg, C, GX)
  switch (NC) {
    case 8: LF(8); break; case 16: LF(16); break; case 32: LF(32); break; case 64: LF(64); break;
    case 128:
#ifdef ISCA_EXPERIMENTS
      if (fft_old()) { LF(128); break; }
      if (!fft_twreg()) { LF3(128, false); break; }
#endif
      LF3(128, true); break;
    case 256:
#ifdef ISCA_EXPERIMENTS
      if (fft_old()) { LF(256); break; }
      if (!fft_twreg()) { LF3(256, false); break; }
#endif
      LF3(256, true); break;
    default: launch_fft_mixed(false, g, d, fl, Fg, s);
  }
#undef LF
#undef LF3
}
void launch_fft_inverse(const Geom &g, const Dev &d, const FieldList &fl, const double *Fg, hipStream_t s) {
  const int C = col_pitch(fl.nbuf ? fl.nbuf : fl.ncol), NC = g.I / 2;
  if (fl.nbuf && (NC < 128 || (NC & (NC - 1)) || fft_old())) throw std::runtime_error("fft: x-derivatives in Fourier space need the lon_max = 256 / 512 kernels");
  const int R = fft_rows(NC);
  const int GX = (fl.ncol + R - 1) / R;
  dim3 grid(fft_grid(GX * g.Jl));
  const size_t lds = fft_lds_bytes(NC);
#define LI3(N, TW) hipLaunchKernelGGL((k_fft_inv3<N, TW>), fft3_grid(GX * g.Jl), dim3(256), fft3_lds_bytes(NC), s, g, fl, d.cosm_lat_l, d.slot_of_m, (const double2 *)d.tw, Fg, C, GX)
#define LI(N) hipLaunchKernelGGL(k_fft_inv<N>, grid, dim3(R * 16), lds, s, g, fl, d.cosm_lat_l, d.slot_of_m, (const double2 *)d.tw, Fg, C, GX)
  switch (NC) {
    case 8: LI(8); break; case 16: LI(16); break; case 32: LI(32); break; case 64: LI(64); break;
    case 128:
#ifdef ISCA_EXPERIMENTS
      if (fft_old()) { LI(128); break; }
      if (!fft_twreg()) { LI3(128, false); break; }
#endif
      LI3(128, true); break;
    case 256:
#ifdef ISCA_EXPERIMENTS
      if (fft_old()) { LI(256); break; }
      if (!fft_twreg()) { LI3(256, false); break; }
#endif
      LI3(256, true); break;
    default: launch_fft_mixed(true, g, d, fl, const_cast<double *>(Fg), s);
  }
#undef LI
#undef LI3
}

// =====================================================================================================
// Spectral-space operators (tools/spherical.F90:270-600).  State arrays are [ml][n][lev] complex.
// =====================================================================================================
#define COEF(id, ml, n) coef[((size_t)(id) * g.Ml + (ml)) * g.N1 + (n)]

__global__ void k_spec_pack(Geom g, const double2 *__restrict__ st, double *__restrict__ S, int C, int coloff, int nlev) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)g.Ml * g.N1 * nlev;
  if (idx >= tot) return;
  const int k = idx % nlev;
  const size_t mn = idx / nlev;
  *(double2 *)(S + mn * C + 2 * (coloff + k)) = st[idx];
}
__global__ void k_spec_unpack(Geom g, const double *__restrict__ coef, const double *__restrict__ S,
                              double2 *__restrict__ st, int C, int coloff, int nlev, int mask) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)g.Ml * g.N1 * nlev;
  if (idx >= tot) return;
  const int k = idx % nlev;
  const size_t mn = idx / nlev;
  double2 v = *(const double2 *)(S + mn * C + 2 * (coloff + k));
  if (mask) { const double mk = coef[(size_t)C_MASK * g.Ml * g.N1 + mn]; v.x *= mk; v.y *= mk; }
  st[idx] = v;
}
// compute_ucos_vcos (spherical.F90:409-469)
__global__ void k_spec_ucos_vcos(Geom g, const double *__restrict__ coef, const double2 *__restrict__ vor,
                                 const double2 *__restrict__ div, double *__restrict__ S, int C, int col_u, int col_v, int nlev) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)g.Ml * g.N1 * nlev;
  if (idx >= tot) return;
  const int k = idx % nlev;
  const size_t mn = idx / nlev;
  const int n = mn % g.N1, ml = mn / g.N1;
  const double2 vo = vor[idx], dv = div[idx];
  double2 u = cscale(COEF(C_UVC, ml, n), ctimes_i(dv));
  double2 v = cscale(COEF(C_UVC, ml, n), ctimes_i(vo));
  if (n >= 1) {
    const double cm = COEF(C_UVM, ml, n);
    u = cadd(u, cscale(cm, vor[idx - nlev]));
    v = csub(v, cscale(cm, div[idx - nlev]));
  }
  if (n + 1 < g.N1) {
    const double cp = COEF(C_UVP, ml, n);
    u = csub(u, cscale(cp, vor[idx + nlev]));
    v = cadd(v, cscale(cp, div[idx + nlev]));
  }
  *(double2 *)(S + mn * C + 2 * (col_u + k)) = u;
  *(double2 *)(S + mn * C + 2 * (col_v + k)) = v;
}
// compute_vor_div (spherical.F90:472-561) + triangular truncation (:564-600)
__device__ __forceinline__ void alpha_pair(const Geom &g, const double *__restrict__ coef, const double *__restrict__ S,
                                           int C, size_t mn, int ml, int n, int cu, int cv, double2 &vor, double2 &div,
                                           bool mask = true) {
  const double2 U = *(const double2 *)(S + mn * C + 2 * cu), V = *(const double2 *)(S + mn * C + 2 * cv);
  const double dx = COEF(C_DX, ml, n);
  vor = cscale(dx, ctimes_i(V));
  div = cscale(dx, ctimes_i(U));
  if (n >= 1) {
    const double am = COEF(C_ALPM, ml, n);
    const double2 Um = *(const double2 *)(S + (mn - 1) * C + 2 * cu), Vm = *(const double2 *)(S + (mn - 1) * C + 2 * cv);
    vor = cadd(vor, cscale(am, Um));     // alpha(v,u,-1): - (-1) alpm u(n-1)
    div = csub(div, cscale(am, Vm));     // alpha(u,v,+1): - alpm v(n-1)
  }
  if (n + 1 < g.N1) {
    const double ap = COEF(C_ALPP, ml, n);
    const double2 Up = *(const double2 *)(S + (mn + 1) * C + 2 * cu), Vp = *(const double2 *)(S + (mn + 1) * C + 2 * cv);
    vor = csub(vor, cscale(ap, Up));
    div = cadd(div, cscale(ap, Vp));
  }
  if (mask) {
    const double mk = COEF(C_MASK, ml, n);
    vor = cscale(mk, vor);
    div = cscale(mk, div);
  }
}
__global__ void k_spec_vor_div(Geom g, const double *__restrict__ coef, const double *__restrict__ S, int C, int col_u,
                               int col_v, double2 *__restrict__ vor, double2 *__restrict__ div, int nlev, int mask) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)g.Ml * g.N1 * nlev;
  if (idx >= tot) return;
  const int k = idx % nlev;
  const size_t mn = idx / nlev;
  const int n = mn % g.N1, ml = mn / g.N1;
  double2 vo, dv;
  alpha_pair(g, coef, S, C, mn, ml, n, col_u + k, col_v + k, vo, dv, mask != 0);
  vor[idx] = vo;
  div[idx] = dv;
}
// compute_gradient_cos (spherical.F90:270-351)
__global__ void k_spec_gradient(Geom g, const double *__restrict__ coef, const double2 *__restrict__ st,
                                double *__restrict__ S, int C, int col_dx, int col_dy, int nlev) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)g.Ml * g.N1 * nlev;
  if (idx >= tot) return;
  const int k = idx % nlev;
  const size_t mn = idx / nlev;
  const int n = mn % g.N1, ml = mn / g.N1;
  const double2 dx = cscale(COEF(C_DX, ml, n), ctimes_i(st[idx]));
  double2 dy = make_double2(0., 0.);
  if (n >= 1) dy = cscale(-COEF(C_DYM, ml, n), st[idx - nlev]);
  if (n + 1 < g.N1) dy = cadd(dy, cscale(COEF(C_DYP, ml, n), st[idx + nlev]));
  *(double2 *)(S + mn * C + 2 * (col_dx + k)) = dx;
  *(double2 *)(S + mn * C + 2 * (col_dy + k)) = dy;
}

static inline dim3 grid1d(size_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

// levels [k0, k0 + nk) of a spectral array [ml][n][nlev], `copies` times one behind the other (the send blocks of the stand-alone transforms' gather, api.hip)
__global__ void k_spec_level_chunk(size_t mn_count, const double2 *__restrict__ src, double2 *__restrict__ dst, int nlev, int k0, int nk, int copies) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, blk = mn_count * nk;
  if (idx >= blk) return;
  const size_t mn = idx / nk;
  const int k = (int)(idx - mn * nk);
  const double2 v = src[mn * nlev + k0 + k];
  for (int c = 0; c < copies; ++c) dst[(size_t)c * blk + idx] = v;
}
void launch_spec_level_chunk(const Geom &g, const double *src, double *dst, int nlev, int k0, int nk, int copies, hipStream_t s) {
  const size_t mn = (size_t)g.Ml * g.N1;
  hipLaunchKernelGGL(k_spec_level_chunk, grid1d(mn * nk), dim3(256), 0, s, mn, (const double2 *)src, (double2 *)dst, nlev, k0, nk, copies);
}
void launch_spec_pack(const Geom &g, const double *state, double *S, int C, int coloff, int nlev, hipStream_t s) {
  hipLaunchKernelGGL(k_spec_pack, grid1d((size_t)g.Ml * g.N1 * nlev), dim3(256), 0, s, g, (const double2 *)state, S, C, coloff, nlev);
}
void launch_spec_unpack(const Geom &g, const Dev &d, const double *S, double *state, int C, int coloff, int nlev, int mask, hipStream_t s) {
  hipLaunchKernelGGL(k_spec_unpack, grid1d((size_t)g.Ml * g.N1 * nlev), dim3(256), 0, s, g, d.coef, S, (double2 *)state, C, coloff, nlev, mask);
}
void launch_spec_ucos_vcos(const Geom &g, const Dev &d, const double *vor, const double *div, double *S, int C, int col_u, int col_v, int nlev, hipStream_t s) {
  hipLaunchKernelGGL(k_spec_ucos_vcos, grid1d((size_t)g.Ml * g.N1 * nlev), dim3(256), 0, s, g, d.coef, (const double2 *)vor, (const double2 *)div, S, C, col_u, col_v, nlev);
}
void launch_spec_vor_div(const Geom &g, const Dev &d, const double *S, int C, int col_u, int col_v, double *vor, double *div, int nlev, hipStream_t s, int mask) {
  hipLaunchKernelGGL(k_spec_vor_div, grid1d((size_t)g.Ml * g.N1 * nlev), dim3(256), 0, s, g, d.coef, S, C, col_u, col_v, (double2 *)vor, (double2 *)div, nlev, mask);
}
// compute_laplacian (spherical.F90:354-406): (-eigen_laplacian)**power, 0 where the eigenvalue is 0 for power < 0
__global__ void k_spec_laplacian(Geom g, const double *__restrict__ coef, const double2 *__restrict__ in,
                                 double2 *__restrict__ out, int nlev, int power) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)g.Ml * g.N1 * nlev;
  if (idx >= tot) return;
  const size_t mn = idx / nlev;
  const double e = -coef[(size_t)C_EIG * g.Ml * g.N1 + mn];
  double f = 1.0, b = e;
  for (int p = power < 0 ? -power : power; p > 0; p >>= 1) { if (p & 1) f *= b; b *= b; }   // integer power by squaring
  if (power < 0) f = (e != 0.0) ? 1.0 / f : 0.0;
  out[idx] = cscale(f, in[idx]);
}
void launch_spec_laplacian(const Geom &g, const Dev &d, const double *in, double *out, int nlev, int power, hipStream_t s) {
  hipLaunchKernelGGL(k_spec_laplacian, grid1d((size_t)g.Ml * g.N1 * nlev), dim3(256), 0, s, g, d.coef, (const double2 *)in, (double2 *)out, nlev, power);
}
void launch_spec_gradient(const Geom &g, const Dev &d, const double *state, double *S, int C, int col_dx, int col_dy, int nlev, hipStream_t s) {
  hipLaunchKernelGGL(k_spec_gradient, grid1d((size_t)g.Ml * g.N1 * nlev), dim3(256), 0, s, g, d.coef, (const double2 *)state, S, C, col_dx, col_dy, nlev);
}

// -----------------------------------------------------------------------------------------------------
// S1: spectral tendencies from the forward batch (spectral_dynamics.F90:874,891,900-904).
// Forward-batch columns: [0,L) dt_u/cos, [L,2L) dt_v/cos, [2L,3L) dt_T, [3L,4L) Phi+KE, 4L: dt_ln_ps
// -----------------------------------------------------------------------------------------------------
__global__ void k_spec_tendencies(Geom g, const double *__restrict__ coef, const double *__restrict__ Sf, int C,
                                  double2 *__restrict__ dtvor, double2 *__restrict__ dtdiv, double2 *__restrict__ dtT,
                                  double2 *__restrict__ dtlp) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int L = g.L;
  const size_t tot = (size_t)g.Ml * g.N1 * L;
  if (idx >= tot) return;
  const int k = idx % L;
  const size_t mn = idx / L;
  const int n = mn % g.N1, ml = mn / g.N1;
  const double mk = COEF(C_MASK, ml, n);
  if (mk == 0.0) {
    dtvor[idx] = dtdiv[idx] = dtT[idx] = make_double2(0., 0.);
    if (k == 0) dtlp[mn] = make_double2(0., 0.);
    return;
  }
  double2 vo, dv;
  alpha_pair(g, coef, Sf, C, mn, ml, n, k, L + k, vo, dv);
  const double2 E = *(const double2 *)(Sf + mn * C + 2 * (3 * L + k));
  dv = cadd(dv, cscale(COEF(C_EIG, ml, n), E));     // dt_divs - laplacian(E),  laplacian = -eigen*E
  dtvor[idx] = vo;
  dtdiv[idx] = dv;
  dtT[idx] = *(const double2 *)(Sf + mn * C + 2 * (2 * L + k));
  if (k == 0) dtlp[mn] = *(const double2 *)(Sf + mn * C + 2 * (4 * L));
}

__device__ __forceinline__ double2 sel2(bool c, double2 v) { return make_double2(c ? v.x : 0.0, c ? v.y : 0.0); }   // by value: stays in registers

// wave-wide inclusive prefix sum over lanes 0..63 with DPP row shifts / row broadcasts (no LDS crossbar traffic)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
  (void)lane;
  v = dpp_add<0x111, 0xf>(v);     // row_shr:1
  v = dpp_add<0x112, 0xf>(v);     // row_shr:2
  v = dpp_add<0x114, 0xf>(v);     // row_shr:4
  v = dpp_add<0x118, 0xf>(v);     // row_shr:8
  v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 -> rows 1,3
  v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 -> rows 2,3
  return v;
}
__device__ __forceinline__ double wave_last(double v) {   // value held by lane 63
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// -----------------------------------------------------------------------------------------------------
// S2: one wavefront per retained (m,n), lane = level.  implicit_correction (implicit.F90:241-325) with
// the L x L wave matrix applied by lane-broadcast mat-vec, spectral damping (spectral_damping.F90:172-291),
// leapfrog_2level_A + the Robert filter completion leapfrog_2level_B (leapfrog.F90:58-105) for
// raw_filter_coeff = 1.  impl_vec rows: 0, 1 the weights of linear_tp_tendency (dlog_1, dlog_3; other values for vert_difference_option = 'mcm'),
// 2 dp_ref, 3 h, 4 dlogf = lph(k+1)-lpf(k), 5 dlog_3 (linear_geopotential)
// -----------------------------------------------------------------------------------------------------
struct SpecUpdateArgs {
  double2 *vors_p, *vors_c, *vors_f, *divs_p, *divs_c, *divs_f, *ts_p, *ts_c, *ts_f, *lnps_p, *lnps_c, *lnps_f;
  const int *active; int nactive;
  double2 *dtvor, *dtdiv, *dtT, *dtlp;
  const double *coef, *impl_vec, *wave_t, *Sf;
  const int *m_local;
  int C, fourier_inc;
  double delta_t, xi, ref_p, ref_t, robert, eddy_sponge, zmu_sponge, zmv_sponge;
  // raw_filter_coeff /= 1 (Robert-Asselin-Williams): the part of the filter that is known before the new level exists,
  // prev - 2 cur (leapfrog_2level_A's part_filt_*), is kept for the end of the step (leapfrog_2level_B); null when raw = 1
  double raw;
  double2 *part_vor, *part_div, *part_t, *part_lp;
};

__device__ __forceinline__ void lin_tp(double2 dv, int lane, int L, double dp, double dlog1, double dlog3, double ref_t,
                                       double2 &dt_p, double2 &dt_t) {
  // linear_tp_tendency (implicit.F90:414-480) with a uniform reference temperature
  const double2 dmean = (lane < L) ? cscale(dp, dv) : make_double2(0., 0.);
  double2 inc;
  inc.x = wave_incl_scan(dmean.x, lane);
  inc.y = wave_incl_scan(dmean.y, lane);
  const double2 before = csub(inc, dmean);
  const double f = -KAPPA * ref_t / dp;
  dt_t = make_double2(f * (before.x * dlog3 + dmean.x * dlog1), f * (before.y * dlog3 + dmean.y * dlog1));
  dt_p = make_double2(-wave_last(inc.x), -wave_last(inc.y));
}

// MODE 0 is the step.  The other modes run parts of the same code on caller data for the C-ABI entry points
// isca_implicit_correction / isca_compute_spectral_damping / isca_leapfrog (tendencies given in dtvor..dtlp):
enum { SU_GIVEN = 1, SU_NO_IMPLICIT = 2, SU_NO_DAMPING = 4, SU_STOP_IMPLICIT = 8, SU_STOP_DAMPING = 16 };
template <int MODE>
__global__ __launch_bounds__(256) void k_spec_update(Geom g, SpecUpdateArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double2 *xsb = (double2 *)smem;                    // [4][64] per-wavefront vector for the wave-matrix product
  double *ws = (double *)(xsb + 4 * 64);             // [L][L] transposed wave matrix of this block's total wavenumber
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int L = g.L;
  // the list is grouped by total wavenumber (padded), so the 4 wavefronts of a block share one matrix
  // workgroups go round-robin over the 8 XCDs: block b of the launch takes list position (b % 8) * (G / 8) + b / 8, so that each XCD works
  // on a contiguous run of total wavenumbers (G is a multiple of 8: the host pads the list)
  const int G = gridDim.x, bx = blockIdx.x;
  const int blk = (G & 7) ? bx : (bx & 7) * (G >> 3) + (bx >> 3);
  const int4 ent = ((const int4 *)a.active)[blk * 4 + wave];      // {n, ml, m, total wavenumber}: one scalar load
  // the block's wave matrix -> LDS: the first eight values per thread are requested here, in front of the state loads, and stored
  // below (a plain copy loop paid one round trip per 256 values -- seven in front of everything else at L = 40)
  const double *W = a.wave_t + (size_t)ent.w * L * L;    // L*L may be odd: 8-byte copies
  const int LL = L * L;
  double wv[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) wv[r] = W[min(r * 256 + (int)threadIdx.x, LL - 1)];
  const bool idle = ent.x < 0;                           // padding entry: helps with the copy, computes on (0,0), stores nothing
  const int n = idle ? 0 : ent.x, ml = ent.y;            // retained (m,n) only: outside the triangle the state stays zero
  const int mn = ml * g.N1 + n;
  const double *coef = a.coef;
  const bool act = lane < L && !idle;
  const int kk = (lane < L) ? lane : 0;
  const size_t idx = (size_t)mn * L + kk;
  const double dlog1 = a.impl_vec[0 * 64 + kk], dlog3 = a.impl_vec[1 * 64 + kk], dp = a.impl_vec[2 * 64 + kk];
  const double hk = a.impl_vec[3 * 64 + kk], dlogf = a.impl_vec[4 * 64 + kk], dlog3g = a.impl_vec[5 * 64 + kk];
  const double eig = COEF(C_EIG, ml, n), dmp = COEF(C_DAMP, ml, n);   // read before the first store (scalar path)
  const double dmp_v = COEF(C_DAMP_VOR, ml, n), dmp_d = COEF(C_DAMP_DIV, ml, n);
  const int mglob = ent.z;
  const double2 zero = make_double2(0., 0.);
  // unconditional loads (idx is valid for every lane) masked afterwards: `c ? lvalue : lvalue` on a double2 selects
  // an address and would push both operands into scratch memory
  const double2 dprev = sel2(act, a.divs_p[idx]), dcur = sel2(act, a.divs_c[idx]);
  const double2 tprev = sel2(act, a.ts_p[idx]), tcur = sel2(act, a.ts_c[idx]);
  const double2 vprev = sel2(act, a.vors_p[idx]), vcur = sel2(act, a.vors_c[idx]);
  const double2 lprev = a.lnps_p[mn], lcur = a.lnps_c[mn];
  // --- spectral tendencies of the forward batch (spectral_dynamics.F90:874,891,900-904)
  double2 dt_vor, dt_div, dt_t = zero, dt_lp;
  if (MODE & SU_GIVEN) {
    dt_vor = sel2(act, a.dtvor[idx]); dt_div = sel2(act, a.dtdiv[idx]); dt_t = sel2(act, a.dtT[idx]);
    dt_lp = a.dtlp[mn];
  } else {
    // compute_vor_div of (dt_u, dt_v)/cos (alpha_pair above, spherical.F90:472-561) with every load of the forward
    // batch issued together: rows n-1 / n+1 are clamped and their coefficients zeroed at the ends of the column
    const double *row = a.Sf + (size_t)mn * a.C;
    const int dm_ = (n >= 1) ? a.C : 0, dp_ = (n + 1 < g.N1) ? a.C : 0;
    const double2 U = *(const double2 *)(row + 2 * kk), V = *(const double2 *)(row + 2 * (L + kk));
    const double2 Um = *(const double2 *)(row - dm_ + 2 * kk), Vm = *(const double2 *)(row - dm_ + 2 * (L + kk));
    const double2 Up = *(const double2 *)(row + dp_ + 2 * kk), Vp = *(const double2 *)(row + dp_ + 2 * (L + kk));
    const double2 E = *(const double2 *)(row + 2 * (3 * L + kk));
    const double2 Tt = *(const double2 *)(row + 2 * (2 * L + kk));
    dt_lp = *(const double2 *)(row + 2 * (4 * L));
    const double dx = COEF(C_DX, ml, n), mk = COEF(C_MASK, ml, n);
    const double am = (n >= 1) ? COEF(C_ALPM, ml, n) : 0.0, ap = (n + 1 < g.N1) ? COEF(C_ALPP, ml, n) : 0.0;
    dt_vor = cscale(dx, ctimes_i(V));
    dt_div = cscale(dx, ctimes_i(U));
    dt_vor = cadd(dt_vor, cscale(am, Um));            // alpha(v,u,-1)
    dt_div = csub(dt_div, cscale(am, Vm));            // alpha(u,v,+1)
    dt_vor = csub(dt_vor, cscale(ap, Up));
    dt_div = cadd(dt_div, cscale(ap, Vp));
    dt_vor = sel2(act, cscale(mk, dt_vor));
    dt_div = sel2(act, cadd(cscale(mk, dt_div), cscale(eig, E)));   // dt_divs - laplacian(Phi+KE), laplacian = -eigen
    dt_t = sel2(act, Tt);
    if (a.dtvor) {                       // locals of the reference's step: stored only for the phase-by-phase API (get_state "s_dtvor" ...)
      if (act) { a.dtvor[idx] = dt_vor; a.dtdiv[idx] = dt_div; a.dtT[idx] = dt_t; }
      if (lane == 0 && !idle) a.dtlp[mn] = dt_lp;
    }
  }
  for (int base = 0; base < LL; base += 8 * 256) {       // (a second pass of eight from L = 46 up)
    if (base > 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) wv[r] = W[min(base + r * 256 + (int)threadIdx.x, LL - 1)];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int i = base + r * 256 + (int)threadIdx.x;
      if (i < LL) ws[i] = wv[r];
    }
  }
  double2 dps, dts;
  if (!(MODE & SU_NO_IMPLICIT)) {
    // --- adjust_dt_divs (:289-325)
    lin_tp(csub(dprev, dcur), lane, L, dp, dlog1, dlog3, a.ref_t, dps, dts);
    dt_t = cadd(dt_t, dts);
    dt_lp = cadd(dt_lp, cscale(1.0 / a.ref_p, dps));
    const double2 ts_temp = cadd(csub(tprev, tcur), cscale(a.xi, dt_t));
    const double2 ps_temp = cadd(csub(lprev, lcur), cscale(a.xi, dt_lp));
    {  // linear_geopotential (:329-359) with del_ln_p = 0: suffix sums of RDGAS*T'*dlog3 below the level
      const double2 av = sel2(act && lane >= 1, cscale(RDGAS * dlog3g, ts_temp));
      double2 inc;
      inc.x = wave_incl_scan(av.x, lane);
      inc.y = wave_incl_scan(av.y, lane);
      const double2 tot = make_double2(wave_last(inc.x), wave_last(inc.y));
      const double2 below = csub(tot, inc);                                  // sum over k' > k
      const double2 geo = cadd(below, cscale(RDGAS * dlogf, ts_temp));
      const double hp = hk * a.ref_p;
      dt_div = cadd(dt_div, cscale(eig, make_double2(geo.x + hp * ps_temp.x, geo.y + hp * ps_temp.y)));
    }
    {  // dt_divs <- wave_matrix(L) . dt_divs (:268-277); ws[k'][k] in LDS, x broadcast from LDS
      double2 *xs = xsb + wave * 64;
      xs[lane] = sel2(act, dt_div);
      __syncthreads();                                  // matrix copy + x vectors visible
      double2 out = zero;
  #pragma unroll 8
      for (int k2 = 0; k2 < L; ++k2) {
        const double2 x = xs[k2];
        const double w = ws[k2 * L + kk];
        out.x += w * x.x;
        out.y += w * x.y;
      }
      dt_div = sel2(act, out);
    }
    if (idle) return;
    lin_tp(dt_div, lane, L, dp, dlog1, dlog3, a.ref_t, dps, dts);
    dt_t = cadd(dt_t, cscale(a.xi, dts));
    dt_lp = cadd(dt_lp, cscale(a.xi / a.ref_p, dps));
  }
  if (idle) return;
  if (MODE & SU_STOP_IMPLICIT) {
    if (act) { a.dtdiv[idx] = dt_div; a.dtT[idx] = dt_t; }
    if (lane == 0) a.dtlp[mn] = dt_lp;
    return;
  }
  // --- damping
  if (!(MODE & SU_NO_DAMPING)) {
    const double cf = 1.0 / (1.0 + dmp * a.delta_t);
    dt_vor = cscale(1.0 / (1.0 + dmp_v * a.delta_t), csub(dt_vor, cscale(dmp_v, vprev)));
    dt_div = cscale(1.0 / (1.0 + dmp_d * a.delta_t), csub(dt_div, cscale(dmp_d, dprev)));
    dt_t = cscale(cf, csub(dt_t, cscale(dmp, tprev)));
    if (lane == 0 && (a.eddy_sponge != 0.0 || a.zmu_sponge != 0.0 || a.zmv_sponge != 0.0)) {   // sponge on the top level (:236-245, :281-290)
      const double sv = (mglob != 0) ? a.eddy_sponge * eig : a.zmu_sponge * eig;
      const double sd = (mglob != 0) ? a.eddy_sponge * eig : a.zmv_sponge * eig;
      dt_vor = cscale(1.0 / (1.0 + sv * a.delta_t), csub(dt_vor, cscale(sv, vprev)));
      dt_div = cscale(1.0 / (1.0 + sd * a.delta_t), csub(dt_div, cscale(sd, dprev)));
    }
  }
  if (MODE & SU_STOP_DAMPING) {
    if (act) { a.dtvor[idx] = dt_vor; a.dtdiv[idx] = dt_div; a.dtT[idx] = dt_t; }
    return;
  }
  // --- leapfrog_2level_A then the `current` half of _B (leapfrog.F90:58-105); rc = robert_coeff * raw_filter_coeff.  The `future`
  // half of _B (only with raw_filter_coeff /= 1) is applied at the end of the step, after the grid fields have been synthesised from
  // the unadjusted new level like the reference does (spectral_dynamics.F90:933-937 before :1031), by k_raw_adjust.
  const double rc = a.robert * a.raw, dtt = a.delta_t;
#define LEAP(PREV, CUR, DT, ARR, PART, IDX, GUARD)                                  \
  {                                                                                 \
    const double2 part = make_double2(PREV.x - 2.0 * CUR.x, PREV.y - 2.0 * CUR.y);  \
    const double2 nf = make_double2(PREV.x + dtt * DT.x, PREV.y + dtt * DT.y);      \
    double2 nc = make_double2(CUR.x + rc * part.x, CUR.y + rc * part.y);            \
    nc = make_double2(nc.x + rc * nf.x, nc.y + rc * nf.y);                          \
    if (GUARD) { ARR##_c[IDX] = nc; ARR##_f[IDX] = nf; if (PART) PART[IDX] = part; } \
  }
  LEAP(vprev, vcur, dt_vor, a.vors, a.part_vor, idx, act)
  LEAP(dprev, dcur, dt_div, a.divs, a.part_div, idx, act)
  LEAP(tprev, tcur, dt_t, a.ts, a.part_t, idx, act)
  LEAP(lprev, lcur, dt_lp, a.lnps, a.part_lp, mn, lane == 0)
#undef LEAP
}

// leapfrog_2level_B's future half for raw_filter_coeff /= 1 (leapfrog.F90:101-102, called from complete_robert_filter at the very end
// of spectral_dynamics, :1031): a(future) += robert (raw - 1) (part_filt + a(future)), on the retained coefficients of the four fields
__global__ void k_raw_adjust(size_t n3, size_t n2, double f, double2 *vor, double2 *div, double2 *ts, double2 *lnps,
                             const double2 *__restrict__ pv, const double2 *__restrict__ pd, const double2 *__restrict__ pt,
                             const double2 *__restrict__ pl) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  auto adj = [&](double2 *x, const double2 *p) { double2 v = x[i]; const double2 q = p[i]; v.x += f * (q.x + v.x); v.y += f * (q.y + v.y); x[i] = v; };
  if (i < n3) { adj(vor, pv); adj(div, pd); adj(ts, pt); }
  if (i < n2) adj(lnps, pl);
}
void launch_raw_adjust(const isca_dyn &h, int fut, hipStream_t s) {
  const Geom &g = h.g;
  const size_t n3 = (size_t)g.Ml * g.N1 * g.L, n2 = (size_t)g.Ml * g.N1;
  hipLaunchKernelGGL(k_raw_adjust, grid1d(n3), dim3(256), 0, s, n3, n2, h.cfg.robert_coeff * (h.cfg.raw_filter_coeff - 1.0), (double2 *)h.d.vors[fut],
                     (double2 *)h.d.divs[fut], (double2 *)h.d.ts[fut], (double2 *)h.d.lnps[fut], (const double2 *)h.d.part_vor,
                     (const double2 *)h.d.part_div, (const double2 *)h.d.part_t, (const double2 *)h.d.part_lp);
}

void launch_spec_tendencies(const isca_dyn &h, hipStream_t s) {
  const Geom &g = h.g;
  hipLaunchKernelGGL(k_spec_tendencies, grid1d((size_t)g.Ml * g.N1 * g.L), dim3(256), 0, s, g, h.d.coef, h.d.Sf, h.Cf,
                     (double2 *)h.d.s_dtvor, (double2 *)h.d.s_dtdiv, (double2 *)h.d.s_dtT, (double2 *)h.d.s_dtlp);
}
static SpecUpdateArgs spec_update_args(const isca_dyn &h, const StepScalars &sc) {
  SpecUpdateArgs a;
  a.vors_p = (double2 *)h.d.vors[sc.prev]; a.vors_c = (double2 *)h.d.vors[sc.cur]; a.vors_f = (double2 *)h.d.vors[sc.fut];
  a.divs_p = (double2 *)h.d.divs[sc.prev]; a.divs_c = (double2 *)h.d.divs[sc.cur]; a.divs_f = (double2 *)h.d.divs[sc.fut];
  a.ts_p = (double2 *)h.d.ts[sc.prev]; a.ts_c = (double2 *)h.d.ts[sc.cur]; a.ts_f = (double2 *)h.d.ts[sc.fut];
  a.lnps_p = (double2 *)h.d.lnps[sc.prev]; a.lnps_c = (double2 *)h.d.lnps[sc.cur]; a.lnps_f = (double2 *)h.d.lnps[sc.fut];
  a.active = h.d.mn_active; a.nactive = h.n_active;
  a.dtvor = (double2 *)h.d.s_dtvor; a.dtdiv = (double2 *)h.d.s_dtdiv; a.dtT = (double2 *)h.d.s_dtT; a.dtlp = (double2 *)h.d.s_dtlp;
  a.coef = h.d.coef; a.impl_vec = h.d.impl_vec; a.wave_t = h.d.wave_mat_t; a.m_local = h.d.m_local;
  a.Sf = h.d.Sf; a.C = h.Cf; a.fourier_inc = h.cfg.fourier_inc;
  a.delta_t = sc.delta_t; a.xi = sc.xi; a.ref_p = h.tab.ref_surf_p; a.ref_t = h.tab.ref_t; a.robert = h.cfg.robert_coeff;
  a.eddy_sponge = h.cfg.eddy_sponge_coeff; a.zmu_sponge = h.cfg.zmu_sponge_coeff; a.zmv_sponge = h.cfg.zmv_sponge_coeff;
  a.raw = h.cfg.raw_filter_coeff;
  a.part_vor = (double2 *)h.d.part_vor; a.part_div = (double2 *)h.d.part_div; a.part_t = (double2 *)h.d.part_t; a.part_lp = (double2 *)h.d.part_lp;
  return a;
}
void launch_spec_update(const isca_dyn &h, const StepScalars &sc, hipStream_t s) {
  const Geom &g = h.g;
  const SpecUpdateArgs a = spec_update_args(h, sc);
  const size_t lds = (size_t)4 * 64 * sizeof(double2) + (size_t)g.L * g.L * sizeof(double);
  SpecUpdateArgs b = a;
  if (!sc.keep_spec_tend) b.dtvor = b.dtdiv = b.dtT = b.dtlp = nullptr;
  // use_implicit = .false. (spectral_dynamics.F90:906): the same kernel without implicit_correction
  if (h.cfg.use_implicit) hipLaunchKernelGGL(k_spec_update<0>, dim3((unsigned)(h.n_active / 4)), dim3(256), lds, s, g, b);
  else hipLaunchKernelGGL(k_spec_update<SU_NO_IMPLICIT>, dim3((unsigned)(h.n_active / 4)), dim3(256), lds, s, g, b);
}
// parts of the same kernel on caller data: stage 0 implicit_correction, 1 spectral damping, 2 leapfrog A+B.
// st[v][t]: v = vors, divs, ts, ln_ps; t = previous, current, future.  dtend: dt_vors, dt_divs, dt_ts, dt_ln_ps.
void launch_spec_update_stage(const isca_dyn &h, int stage, double delta_t, double robert, double *const st[4][3],
                              double *const dtend[4], hipStream_t s) {
  const Geom &g = h.g;
  StepScalars sc{}; sc.delta_t = delta_t; sc.xi = delta_t * h.cfg.alpha_implicit;
  SpecUpdateArgs a = spec_update_args(h, sc);
  a.vors_p = (double2 *)st[0][0]; a.vors_c = (double2 *)st[0][1]; a.vors_f = (double2 *)st[0][2];
  a.divs_p = (double2 *)st[1][0]; a.divs_c = (double2 *)st[1][1]; a.divs_f = (double2 *)st[1][2];
  a.ts_p = (double2 *)st[2][0]; a.ts_c = (double2 *)st[2][1]; a.ts_f = (double2 *)st[2][2];
  a.lnps_p = (double2 *)st[3][0]; a.lnps_c = (double2 *)st[3][1]; a.lnps_f = (double2 *)st[3][2];
  a.dtvor = (double2 *)dtend[0]; a.dtdiv = (double2 *)dtend[1]; a.dtT = (double2 *)dtend[2]; a.dtlp = (double2 *)dtend[3];
  a.robert = robert;
  a.part_vor = a.part_div = a.part_t = a.part_lp = nullptr;
  const size_t lds = (size_t)4 * 64 * sizeof(double2) + (size_t)g.L * g.L * sizeof(double);
  const dim3 grid((unsigned)(h.n_active / 4)), block(256);
  if (stage == 0) hipLaunchKernelGGL(k_spec_update<SU_GIVEN | SU_STOP_IMPLICIT>, grid, block, lds, s, g, a);
  else if (stage == 1) hipLaunchKernelGGL(k_spec_update<SU_GIVEN | SU_NO_IMPLICIT | SU_STOP_DAMPING>, grid, block, lds, s, g, a);
  else hipLaunchKernelGGL(k_spec_update<SU_GIVEN | SU_NO_IMPLICIT | SU_NO_DAMPING>, grid, block, lds, s, g, a);
}

// -----------------------------------------------------------------------------------------------------
// S3: inverse-batch inputs from the spectral state at time level tl (spectral_dynamics.F90:933-937 for the
// new state merged with :855,:890 for the next step's gradients -- SURVEY Appendix B).
// Inverse-batch columns: [0,L) div, [L,2L) vor, [2L,3L) u cos, [3L,4L) v cos, [4L,5L) T, [5L,6L) dT/dx cos,
// [6L,7L) dT/dy cos, 7L ln ps, 7L+1 d(ln ps)/dx cos, 7L+2 d(ln ps)/dy cos
// -----------------------------------------------------------------------------------------------------
__global__ void k_spec_synth_inputs(Geom g, const double *__restrict__ coef, const double2 *__restrict__ vor,
                                    const double2 *__restrict__ div, const double2 *__restrict__ ts,
                                    const double2 *__restrict__ lnps, double *__restrict__ Si, int C) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int L = g.L;
  const size_t tot = (size_t)g.Ml * g.N1 * L;
  if (idx >= tot) return;
  const int k = idx % L;
  const size_t mn = idx / L;
  const int n = mn % g.N1, ml = mn / g.N1;
  const double2 vo = vor[idx], dv = div[idx], tt = ts[idx];
  double2 u = cscale(COEF(C_UVC, ml, n), ctimes_i(dv));
  double2 v = cscale(COEF(C_UVC, ml, n), ctimes_i(vo));
  const double dxc = COEF(C_DX, ml, n);
  const double2 tdx = cscale(dxc, ctimes_i(tt));
  double2 tdy = make_double2(0., 0.);
  if (n >= 1) {
    const double cm = COEF(C_UVM, ml, n);
    u = cadd(u, cscale(cm, vor[idx - L]));
    v = csub(v, cscale(cm, div[idx - L]));
    tdy = cscale(-COEF(C_DYM, ml, n), ts[idx - L]);
  }
  if (n + 1 < g.N1) {
    const double cp = COEF(C_UVP, ml, n);
    u = csub(u, cscale(cp, vor[idx + L]));
    v = cadd(v, cscale(cp, div[idx + L]));
    tdy = cadd(tdy, cscale(COEF(C_DYP, ml, n), ts[idx + L]));
  }
  double *row = Si + mn * C;
  *(double2 *)(row + 2 * (0 * L + k)) = dv;
  *(double2 *)(row + 2 * (1 * L + k)) = vo;
  *(double2 *)(row + 2 * (2 * L + k)) = u;
  *(double2 *)(row + 2 * (3 * L + k)) = v;
  *(double2 *)(row + 2 * (4 * L + k)) = tt;
  *(double2 *)(row + 2 * (5 * L + k)) = tdx;
  *(double2 *)(row + 2 * (6 * L + k)) = tdy;
  if (k == 0) {
    const double2 lp = lnps[mn];
    double2 ldy = make_double2(0., 0.);
    if (n >= 1) ldy = cscale(-COEF(C_DYM, ml, n), lnps[mn - 1]);
    if (n + 1 < g.N1) ldy = cadd(ldy, cscale(COEF(C_DYP, ml, n), lnps[mn + 1]));
    *(double2 *)(row + 2 * (7 * L + 0)) = lp;
    *(double2 *)(row + 2 * (7 * L + 1)) = cscale(dxc, ctimes_i(lp));
    *(double2 *)(row + 2 * (7 * L + 2)) = ldy;
  }
}
void launch_spec_synthesis_inputs(const isca_dyn &h, int tl, hipStream_t s) {
  const Geom &g = h.g;
  hipLaunchKernelGGL(k_spec_synth_inputs, grid1d((size_t)g.Ml * g.N1 * g.L), dim3(256), 0, s, g, h.d.coef,
                     (const double2 *)h.d.vors[tl], (const double2 *)h.d.divs[tl], (const double2 *)h.d.ts[tl],
                     (const double2 *)h.d.lnps[tl], h.d.Si, h.Ci);
}

// =====================================================================================================
// Fixer core (used by the fixer kernels further down AND by the column kernel, whose block 0 finishes the step before's fixers)
// =====================================================================================================
// block = 64 columns x NW wavefronts (level chunks), like the column kernel
constexpr int NRED = 10;   // 2 sums of the column kernel + 8 of k_fixer_sums
constexpr int NPART = 10;  // per block of k_fixer_sums: the 8 sums + min and max of the new temperatures
struct FixerArgs {
  double *red;                  // [0..9] global sums (all-reduced by the host when world_size > 1), [16..18] scalars out
  const double *pprev, *pfut;   // block partials: 2 per block (column kernel), 8 per block (k_fixer_sums)
  int nb, reduce_here;
  double *pend_fut;             // Dev::pend row of the new level
  int patch;               // k_fixer_finish: patch the (0,0) spectral coefficients (not on its second run of a step, after the water sums came in)
  double2 *lnps_fut, *lnps_cur, *ts_fut, *ts_cur;
  double *psg, *tg;
  double *tr_fut, *tr_cur, *tratm_fut;   // grid tracer (null when none)
  const int *kmask;
  int ml0;                 // local slot of m = 0, or -1
  double sumw_nlon;        // global_sum_of_wts * num_lon
  double robert;
  int do_mass, do_energy, do_water;
  double raw;                 // raw_filter_coeff; tr_part: prev - 2 cur of the tracer (RAW filter), null when raw = 1
  const double *tr_part;
};
// compute_corrections (spectral_dynamics.F90:1213-1283, with mj's water-correction limit) from the ten global sums
__device__ __forceinline__ void fixer_scalars(const double *r_, const FixerArgs &a, double &factor, double &tcorr, double &wfac) {
#pragma clang fp contract(off)      // k_fixer_apply and k_fixer_finish must get the same bits from the same sums
  factor = 1.0; tcorr = 0.0; wfac = 1.0;
  const double mean_ps_prev = r_[0] / a.sumw_nlon;
  const double mean_en_prev = r_[1] / a.sumw_nlon / GRAV;
  if (a.do_mass) factor = mean_ps_prev / (r_[2] / a.sumw_nlon);
  if (a.do_energy) {
    const double mean_en_tmp = (r_[3] + factor * r_[4]) / a.sumw_nlon / GRAV;
    tcorr = GRAV * (mean_en_prev - mean_en_tmp) / (CP_AIR * mean_ps_prev);
  }
  if (a.do_water && a.tr_fut) {
    const double nrm = 1.0 / a.sumw_nlon / GRAV;
    const double water_prev = r_[5] * nrm;
    const double water_tmp = (r_[6] + factor * r_[7]) * nrm;
    const double corr = (r_[8] + factor * r_[9]) * nrm;
    const double notc = water_tmp - corr;
    if (water_tmp > 0.) {
      wfac = water_prev / water_tmp;
      wfac = wfac * (1. + notc / corr) - notc / corr;
    }
  }
}
// the (0,0) coefficients of ln ps and T follow the grid corrections (:1231, :1241), also on the Robert-filtered `current` level (:1470-1473)
// (two halves: the coefficients are requested before the scalars are known, so that only their stores follow the reduction)
struct SpecPatch { double lf, lc, tf, tc; };
__device__ __forceinline__ SpecPatch fixer_patch_load(const Geom &g, const FixerArgs &a) {
  SpecPatch p = {0., 0., 0., 0.};
  if (a.ml0 < 0) return p;
  const size_t mn = (size_t)a.ml0 * g.N1;     // (m=0, n=0)
  const int k = min((int)threadIdx.x, g.L - 1);
  p.lf = a.lnps_fut[mn].x; p.lc = a.lnps_cur[mn].x; p.tf = a.ts_fut[mn * g.L + k].x; p.tc = a.ts_cur[mn * g.L + k].x;
  return p;
}
__device__ __forceinline__ void fixer_patch_spectral(const Geom &g, const FixerArgs &a, const SpecPatch &p, double factor, double tcorr) {
  if (a.ml0 < 0) return;
  const size_t mn = (size_t)a.ml0 * g.N1;
  const double s2 = sqrt(2.);
  const int k = threadIdx.x;
  if (k == 0 && a.do_mass) {
    const double dl = s2 * log(factor);
    a.lnps_fut[mn].x = p.lf + dl;
    a.lnps_cur[mn].x = p.lc + a.robert * a.raw * dl;
  }
  if (k < g.L && a.do_energy) {
    const double dtc = s2 * tcorr;
    a.ts_fut[mn * g.L + k].x = p.tf + dtc;
    a.ts_cur[mn * g.L + k].x = p.tc + a.robert * a.raw * dtc;
  }
}
// Totals of the block partials (2 per block from the column kernel, 8 sums + min / max of the new temperatures per block from k_fixer_sums) by ONE
// block, in an order that does not depend on the block's size: 512 VIRTUAL threads -- value c = v & 15 (0..1 the column kernel's sums, 2..9
// k_fixer_sums', 10 min, 11 max), group g = v >> 4 -- each fold the sets g, g + 32, ... of their value in ascending order; the 32 groups of a value are
// then folded in ascending order by one thread.  A real thread takes the virtual threads r, r + blockDim, ...: the same bits from a block of 256 or
// 512 threads -- k_fixer_reduce, k_fixer_finish, and block 0 of the next step's column kernel (the deferred finish, below).  32 loads in flight per
// virtual thread: one memory round trip up to 1024 sets (T85L40: 512), two at T170L60 (2048) -- this is on the critical path of that block 0.
constexpr int FT_GROUPS = 32;
__device__ __forceinline__ void fixer_totals(const double *__restrict__ pprev, const double *__restrict__ pfut, int nb,
                                             double (*sh)[16], double *tot, double &tmin, double &tmax) {
  const int NT = blockDim.x, r = threadIdx.x, c = r & 15, cc = min(c, NRED + 1);
  const double *p0 = cc < 2 ? pprev + cc : pfut + (cc - 2);
  const int st = cc < 2 ? 2 : NPART;
  auto fold = [&](double a, double b) { return cc < NRED ? a + b : (cc == NRED ? fmin(a, b) : fmax(a, b)); };
  constexpr int U = 32;
  for (int vt = r; vt < 16 * FT_GROUPS; vt += NT) {
    const int g = vt >> 4;
    double acc = cc < NRED ? 0.0 : (cc == NRED ? INFINITY : -INFINITY);
    for (int i0 = g; i0 < nb; i0 += U * FT_GROUPS) {
      double x[U];
#pragma unroll
      for (int q = 0; q < U; ++q) x[q] = p0[(size_t)st * min(i0 + FT_GROUPS * q, nb - 1)];
#pragma unroll
      for (int q = 0; q < U; ++q)
        if (i0 + FT_GROUPS * q < nb) acc = fold(acc, x[q]);
    }
    sh[g][c] = acc;
  }
  __syncthreads();
  if (r < NRED + 2) {                                 // one thread per value folds the groups in ascending order (c = r here)
    double x = sh[0][r];
    for (int g = 1; g < FT_GROUPS; ++g) x = fold(x, sh[g][r]);
    sh[0][r] = x;                                     // (thread r is the only reader of column r)
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NRED; ++k) tot[k] = sh[0][k];
  tmin = sh[0][NRED]; tmax = sh[0][NRED + 1];
}
// Lazy fixers: compute_corrections' three scalars, left PENDING on the new time level (pend_fut[0..2]) instead of applied to the grid fields -- every
// later reader of that level applies them (k_column, the tracer kernels; k_fixer_materialize for the host) --, and the (0,0) spectral patch.  One block
// (>= num_levels threads).  reduce_here: fold the block partials first (one rank); otherwise red[0..9] hold the all-reduced totals.  Returns the
// scalars to every thread.
__device__ __forceinline__ void fixer_finish_body(const Geom &g, const FixerArgs &a, double (*sh)[16], double &factor, double &tcorr, double &wfac) {
  double r_[NRED], tmn = INFINITY, tmx = -INFINITY;
  const SpecPatch sp = a.patch ? fixer_patch_load(g, a) : SpecPatch{0., 0., 0., 0.};
  const double mn_old = a.red[20], mx_old = a.red[21];
  if (a.reduce_here) fixer_totals(a.pprev, a.pfut, a.nb, sh, r_, tmn, tmx);
  else {
#pragma unroll
    for (int c = 0; c < NRED; ++c) r_[c] = a.red[c];
  }
  fixer_scalars(r_, a, factor, tcorr, wfac);
  if (threadIdx.x == 0) {
    for (int c = 0; c < NRED; ++c) a.red[c] = r_[c];
    a.red[16] = factor; a.red[17] = tcorr; a.red[18] = wfac;
    if (a.reduce_here) { a.red[20] = fmin(mn_old, tmn); a.red[21] = fmax(mx_old, tmx); }
    a.pend_fut[PEND_FACTOR] = factor; a.pend_fut[PEND_TCORR] = tcorr; a.pend_fut[PEND_WFAC] = wfac;
  }
  if (a.patch) fixer_patch_spectral(g, a, sp, factor, tcorr);
}

static FixerArgs fixer_args(const isca_dyn &h, const StepScalars &sc);

// The deferred finish's device-resident state (Dev::fin_args): the two argument sets a finish can have (future level 0 or 1; the current level is the
// other one), the sequence word block 0 publishes and the two scalars it announces.  ONE pointer in ColumnArgs: the column kernel's scalar
// registers are full (106 with 12-25 spilled), and every spilled scalar takes a vector register from a kernel that lives at the 128 limit.
struct alignas(64) DeferredFin { FixerArgs fa[2]; alignas(64) double val[2][2]; };
// val[seq & 1] = {mass factor, temperature correction} of launch `seq`, written by block 0 with ONE 16-byte write-through store; the slot reads {NaN, NaN}
// until then (block 0 of the launch before reset it), so the pair is its own announcement: no second store, no ordering between two stores -- under
// the load of 512 blocks' first requests every device-scope round trip costs microseconds, and this one is on the critical path of all of them.
__device__ __forceinline__ void fin_publish(double *slot, double f, double t) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 v = {f, t};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(slot), "v"(v) : "memory");
}
__device__ __forceinline__ void fin_peek(const double *slot, double &f, double &t) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(slot) : "memory");
  f = v.x; t = v.y;
}

// =====================================================================================================
// Grid-point column kernel: hs_forcing (hs_forcing.F90:148-272) at the PREVIOUS level with CURRENT
// pressures (atmosphere.F90:304-311), pressure_variables (press_and_geopot.F90:152-221), four_in_one
// (spectral_dynamics.F90:1038-1112), vert_advection second-centred/advective (vert_advection.F90:185-193,
// 467-470), horizontal T advection (transforms.F90:822-828), vorticity/Coriolis terms and Phi+KE
// (spectral_dynamics.F90:893-902), and the "previous" sums of the fixers (:1306-1338).
// One thread per column, 64 columns (one wavefront, consecutive longitudes) per block.
// =====================================================================================================
struct ColumnArgs {
  const double *u, *v, *t, *ps;            // current
  const double *up, *vp, *tp, *psp;        // previous
  const double *vor, *div, *dxT, *dyT, *dxlp, *dylp;
  double *dtu, *dtv, *dtT, *E, *dtlp, *wg_full, *partials, *wg, *psp_copy;
  int *kmask; const int *kmask_rd; double water_limit;   // kmask: null without the grid tracer; kmask_rd: the same array, always valid
  const double *phu, *phv, *pht;           // tendencies of the physics package when it is not hs_forcing (k_column<CH, true>)
  const double *surf_geop;                 // [Jl][I] lower boundary of the hydrostatic integral (press_and_geopot.F90:331)
  const double *tv;                        // virtual temperature of the current level (k_column<CH, EXT, true>: use_virtual_temperature)
  const double *pend_c, *pend_p;           // pending fixer scalars of the current / previous level (identity row when nothing is pending)
  int store_wg_full;                       // wg_full (omega; a diagnostic and restart field) is only stored by steps after which the host can look
  int vadv_skip;                           // bit 0 / 1: the vertical advection of u, v / of T is another scheme's (k_vert_advection_scheme adds it)
  const double *pk, *bk, *dpk, *dbk, *cosm, *coriolis, *rad_lat, *wts;
  double delta_t, tka, tks, vkf, sigma_b, t_zero, delh, delv, eps, t_strat, P00;
  int do_conserve_energy;
  const double *lh_lon, *lh_lat;           // local_heating_option = 'Isidoro' (hs_forcing.F90:728-769): srfamp x longitude factor [I], latitude factor [Jl]; null = off
  double lh_decay;                         // local_heating_vert_decay
  const double *sig;                       // [L][16] per-level constants on pure sigma levels (k_column_sig; api.hip: col_sig); hs_sin: [Jl] sin(lat)
  const double *hs_sin;
  double lnP00;                            // log(P00)
  // The deferred finish of the step BEFORE (k_column_sig; api.hip: fin_deferred).  The fixers' three scalars hang on ten global sums, so they used to be a
  // one-block kernel between k_fixer_sums and this one: 7.6 us + two kernel boundaries on the step's critical path (9-10 us of a 0.166 ms step, measured
  // by leaving it out).  Now block 0 of this kernel computes them (fixer_finish_body: same code, same bits), publishes factor and temperature correction
  // with agent-scope stores + a sequence word, and the other blocks -- whose field loads are in flight meanwhile -- wait for that word before their
  // first arithmetic.  fin_seq = 0: nothing deferred, the scalars are read from pend_c as before.
  unsigned fin_seq; int fin_fut;                              // fin_seq: 0 = nothing deferred, else 1 + the number of deferred launches before this one (its parity picks the slot);
                                                              // red[25] is set when a block gave up waiting
  DeferredFin *fin;
};

__device__ __forceinline__ void hs_level(const ColumnArgs &a, double dt, double ps, double p_full, double up, double vp,
                                         double tp, double sin_lat, double sin2, double cos2, double cos4,
                                         double &utnd, double &vtnd, double &ttnd, double lh_xy = 0.0) {
  // rayleigh_damping (:615-679), dissipative heating (:198-200), newtonian_damping (:508-611)
  const double sigma = p_full * (1. / ps);
  const bool bl = (sigma <= 1.0) && (sigma > a.sigma_b);
  const double vcoeff = -a.vkf / (1.0 - a.sigma_b);
  const double vfactr = bl ? vcoeff * (sigma - a.sigma_b) : 0.0;
  utnd = vfactr * up;
  vtnd = vfactr * vp;
  ttnd = 0.0;
  if (a.do_conserve_energy) ttnd = -((up + .5 * utnd * dt) * utnd + (vp + .5 * vtnd * dt) * vtnd) / CP_AIR;
  const double t_star = a.t_zero - a.delh * sin2 - a.eps * sin_lat;
  const double tstr = a.t_strat - a.eps * sin_lat;
  const double tcoeff = (a.tks - a.tka) / (1.0 - a.sigma_b);
  const double p_norm = p_full / a.P00;
  const double the = t_star - a.delv * cos2 * log(p_norm);
  double teq = the * pow(p_norm, KAPPA);
  teq = fmax(teq, tstr);
  const double tdamp = bl ? a.tka + cos4 * (tcoeff * (sigma - a.sigma_b)) : a.tka;
  ttnd = ttnd + (-tdamp * (tp - teq));
  if (a.lh_lon) ttnd = ttnd + lh_xy * exp((p_full - ps) / a.lh_decay);      // local_heating (:233-235, :760-761)
}

// Block = 64 consecutive columns x NW wavefronts; wavefront w owns the contiguous levels [w*CH, w*CH+CH).
// The two vertical scans (mass-divergence prefix, hydrostatic suffix) are chunk sums exchanged through LDS,
// everything else is local to a thread's <= CH levels, so all loads of a thread are independent and in flight
// together (8x the wavefronts and ~50 outstanding loads per lane instead of one level at a time).
// MCM: vert_difference_option = 'mcm' (press_and_geopot.F90:196-210, spectral_dynamics.F90:1084-1099): p_full = the mean of the two half levels,
// the pressure-gradient term with grad(p_s)/p_s, the conversion term with (sum above + half the layer's own)/p_full.
// (8 wavefronts of CH = ceil(L/8) levels: two per SIMD, up to 256 VGPRs.  More, thinner wavefronts per block -- 3 / 4 per SIMD at <= 170 / 128 VGPRs -- were
// measured in round 5 and spilled: HISTORY.md.)
template <int CH, bool EXT, bool VIRT, bool MCM = false>
__global__ __launch_bounds__(512) void k_column(Geom g, ColumnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L = g.L, I = g.I;
  const int tid = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;   // w in an SGPR: table lookups by level become scalar loads
  double *lds_dm = (double *)smem;          // [NW][64] chunk sums of dmean
  double *lds_a = lds_dm + NW * 64;         // [NW][64] chunk sums of RDGAS*T*dlog3
  double *lds_e = lds_a + NW * 64;          // [NW] energy partials
  int *lds_cnt = (int *)(lds_e + NW);       // [NW][64] levels with p_full < water_correction_limit
  const int col = blockIdx.x * 64 + tid;
  const int jl = col / I;
  const size_t c2 = (size_t)col, lev = (size_t)g.Jl * I;
  const int k0 = w * CH, nk = min(CH, L - k0);          // nk >= 1 by construction of NW
  const double tc_c = a.pend_c[PEND_TCORR], tc_p = a.pend_p[PEND_TCORR];       // T(level) = stored + pending temperature correction
  const double ps = mul_nc(a.ps[c2], a.pend_c[PEND_FACTOR]), psp = mul_nc(a.psp[c2], a.pend_p[PEND_FACTOR]);
  const int kmw_old = a.kmask_rd[c2];                   // (always a valid array: no load under a branch)
  const double dx_ps = ps * a.dxlp[c2], dy_ps = ps * a.dylp[c2];
  const bool top0 = (a.pk[0] == 0.0 && a.bk[0] == 0.0);
  const int ktop = (a.pk[0] == 0.0) ? 1 : 0;
  // per-level constants of my chunk, read before the first store: afterwards the compiler could not keep them on the
  // scalar path (possible aliasing with the outputs) and every level would wait on vector loads and on its stores
  double dpk_r[CH], dbk_r[CH], pk_r[CH + 1], bk_r[CH + 1];
#pragma unroll
  for (int i = 0; i <= CH; ++i) {
    const int k = min(k0 + i, L);
    pk_r[i] = a.pk[k]; bk_r[i] = a.bk[k];
    if (i < CH) { const int kk = min(k0 + i, L - 1); dpk_r[i] = a.dpk[kk]; dbk_r[i] = a.dbk[kk]; }
  }
  const double wts_j = a.wts[jl], cosm = a.cosm[jl], cor = a.coriolis[jl], rad_lat = a.rad_lat[jl];
  double u[CH], v[CH], t[CH], dm[CH];
  double tvv[VIRT ? CH : 1];                // virtual_t of four_in_one / compute_geopotential (spectral_dynamics.F90:857-868); t itself is advected
#define TV(i) (VIRT ? tvv[VIRT ? (i) : 0] : t[i])
  // every global load of the block is issued here, before the barrier of the vertical scans: one memory
  // round trip per block instead of two (the loads below the barrier could not start before it)
  // (chunks of more than 5 levels would not fit the register file that way: they read these six below the barrier)
  constexpr bool EARLY = CH <= 5;
  double upv[EARLY ? CH : 1], vpv[EARLY ? CH : 1], tpv[EARLY ? CH : 1], vov[EARLY ? CH : 1], dxv[EARLY ? CH : 1], dyv[EARLY ? CH : 1];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int k = k0 + (i < nk ? i : 0);
    const size_t q = c2 + (size_t)k * lev;
    u[i] = a.u[q]; v[i] = a.v[q]; t[i] = a.t[q] + tc_c;
    if (VIRT) tvv[VIRT ? i : 0] = a.tv[q];
    if (EARLY) { upv[i] = a.up[q]; vpv[i] = a.vp[q]; tpv[i] = a.tp[q] + tc_p; vov[i] = a.vor[q]; dxv[i] = a.dxT[q]; dyv[i] = a.dyT[q]; }
    dm[i] = a.div[q];
  }
  // neighbours across the chunk boundary for the centred vertical fluxes
  double um = 0., vm = 0., tm = 0., un = 0., vn = 0., tn = 0.;
  if (k0 > 0) { const size_t q = c2 + (size_t)(k0 - 1) * lev; um = a.u[q]; vm = a.v[q]; tm = a.t[q] + tc_c; }
  if (k0 + nk < L) { const size_t q = c2 + (size_t)(k0 + nk) * lev; un = a.u[q]; vn = a.v[q]; tn = a.t[q] + tc_c; }
#pragma unroll
  for (int i = 0; i < CH; ++i) {          // mass divergence of the layer (four_in_one :1064-1067)
    const double dp = dpk_r[i] + dbk_r[i] * ps;
    dm[i] = (i < nk) ? dm[i] * dp + dbk_r[i] * (u[i] * dx_ps + v[i] * dy_ps) : 0.0;
  }
  // ln p at my half levels k0..k0+nk and full levels (press_and_geopot.F90:165-194)
  double lph[CH + 1], lpf[CH];
  double pfm[MCM ? CH : 1];                 // 'mcm': p_full itself (the reference forms it first and takes its logarithm)
  {
    const double ph0 = pk_r[0] + bk_r[0] * ps;
    lph[0] = (top0 && k0 == 0) ? 0.0 : log(ph0);
    double ph_k = ph0;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int k = k0 + (i < nk ? i : 0);
      const double ph_n = pk_r[i + 1] + bk_r[i + 1] * ps;      // (levels past the chunk end are computed but never used)
      const double l_n = log(ph_n);
      lph[i + 1] = l_n;
      if (MCM) { const double pm = 0.5 * (ph_n + ph_k); pfm[MCM ? i : 0] = pm; lpf[i] = log(pm); }
      else if (top0 && k == 0) lpf[i] = l_n - 1.0;
      else lpf[i] = l_n - (1.0 - ph_k * (l_n - lph[i]) / (ph_n - ph_k));
      ph_k = ph_n;
    }
  }
  double csum = 0.0, asum = 0.0;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int k = k0 + (i < nk ? i : 0);
    csum += dm[i];
    asum += (i < nk && k >= ktop) ? RDGAS * TV(i) * (lph[i + 1] - lph[i]) : 0.0;
  }
  lds_dm[w * 64 + tid] = csum;
  lds_a[w * 64 + tid] = asum;
  __syncthreads();
  double base = 0.0, total = 0.0, below = 0.0;
  for (int ww = 0; ww < NW; ++ww) {
    const double x = lds_dm[ww * 64 + tid];
    total += x;
    if (ww < w) base += x;
    if (ww > w) below += lds_a[ww * 64 + tid];
  }
  const double sin_lat = sin(rad_lat);
  const double sin2 = sin_lat * sin_lat, cos2 = 1.0 - sin2, cos4 = cos2 * cos2;
  const double lnP00 = log(a.P00);
  const double vcoeff = -a.vkf / (1.0 - a.sigma_b), tcoeff = (a.tks - a.tka) / (1.0 - a.sigma_b);
  const double t_star = a.t_zero - a.delh * sin2 - a.eps * sin_lat, tstr = a.t_strat - a.eps * sin_lat;
  const double rps = 1. / ps;
  double dmean_tot = base;
  double wg_k = (k0 == 0) ? 0.0 : (-base + total * bk_r[0]);
  double e_prev = 0.0;
  int nbelow = 0;
  if (a.wg && w == 0) { a.wg[c2] = 0.0; a.psp_copy[c2] = psp; }
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    if (i < nk) {
      const int k = k0 + i;
      const size_t q = c2 + (size_t)k * lev;
      const double l_h0 = lph[i], l_h1 = lph[i + 1], l_f = lpf[i];
      const double upi = EARLY ? upv[EARLY ? i : 0] : a.up[q], vpi = EARLY ? vpv[EARLY ? i : 0] : a.vp[q], tpi = EARLY ? tpv[EARLY ? i : 0] : a.tp[q] + tc_p;
      const double voi = EARLY ? vov[EARLY ? i : 0] : a.vor[q], dxti = EARLY ? dxv[EARLY ? i : 0] : a.dxT[q], dyti = EARLY ? dyv[EARLY ? i : 0] : a.dyT[q];
      const double p_full = MCM ? pfm[MCM ? i : 0] : exp(l_f);
      double dt_u, dt_v, dt_t;
      if (EXT) {   // physics tendencies computed beforehand (idealized_moist_phys, atmosphere.F90:304-317)
        dt_u = a.phu[q]; dt_v = a.phv[q]; dt_t = a.pht[q];
      } else {
      // ---- hs_forcing at the previous level (rayleigh :615-679, dissipative heating :198-200, newtonian :508-611)
      const double sigma = p_full * rps;
      const bool bl = (sigma <= 1.0) && (sigma > a.sigma_b);
      const double vfactr = bl ? vcoeff * (sigma - a.sigma_b) : 0.0;
      dt_u = vfactr * upi; dt_v = vfactr * vpi; dt_t = 0.0;
      if (a.do_conserve_energy) dt_t = -((upi + .5 * dt_u * a.delta_t) * dt_u + (vpi + .5 * dt_v * a.delta_t) * dt_v) / CP_AIR;
      {
        const double lpn = l_f - lnP00;                       // log(p_full/P00)
        const double the = t_star - a.delv * cos2 * lpn;
        const double teq = fmax(the * exp(KAPPA * lpn), tstr);   // (p/P00)**kappa
        const double tdamp = bl ? a.tka + cos4 * (tcoeff * (sigma - a.sigma_b)) : a.tka;
        dt_t = dt_t + (-tdamp * (tpi - teq));
      }
      }
      {  // initialize_corrections (:1318-1321)
        const double ue = upi + dt_u * a.delta_t, ve = vpi + dt_v * a.delta_t;
        e_prev += (0.5 * (ue * ue + ve * ve) + CP_AIR * (tpi + dt_t * a.delta_t)) * (dpk_r[i] + dbk_r[i] * psp);
      }
      // ---- four_in_one (:1064-1083)
      const double dp = dpk_r[i] + dbk_r[i] * ps, dp_inv = 1 / dp;
      const double dlog_1 = l_h1 - l_f, dlog_2 = l_f - l_h0, dlog_3 = l_h1 - l_h0;
      const double x1 = (bk_r[i + 1] * dlog_1 + bk_r[i] * dlog_2) * dp_inv;
      const double x2 = MCM ? dx_ps * rps : x1 * dx_ps, x3 = MCM ? dy_ps * rps : x1 * dy_ps;
      const double uc = u[i], vc = v[i], tc = t[i], tvc = TV(i);
      dt_u = dt_u - RDGAS * tvc * x2;
      dt_v = dt_v - RDGAS * tvc * x3;
      const double x4 = MCM ? (dmean_tot + 0.5 * dm[i]) / p_full : (dmean_tot * dlog_3 + dm[i] * dlog_1) * dp_inv;
      const double x5 = x4 - uc * x2 - vc * x3;
      dt_t = dt_t - KAPPA * tvc * x5;
      if (a.store_wg_full) a.wg_full[q] = -x5 * p_full;
      nbelow += (p_full < a.water_limit) ? 1 : 0;
      dmean_tot = dmean_tot + dm[i];
      const double wg_n = (k + 1 < L) ? (-dmean_tot + total * bk_r[i + 1]) : 0.0;
      if (a.wg) a.wg[q + lev] = wg_n;
      // ---- vert_advection SECOND_CENTERED / ADVECTIVE_FORM (vert_advection.F90:185-193, 467-470)
      const double ukm = (i == 0) ? um : u[i > 0 ? i - 1 : 0], vkm = (i == 0) ? vm : v[i > 0 ? i - 1 : 0], tkm = (i == 0) ? tm : t[i > 0 ? i - 1 : 0];
      const double ukp = (i == nk - 1) ? un : u[i + 1 < CH ? i + 1 : i], vkp = (i == nk - 1) ? vn : v[i + 1 < CH ? i + 1 : i], tkp = (i == nk - 1) ? tn : t[i + 1 < CH ? i + 1 : i];
      {
        const double dw = wg_n - wg_k;
        const double fu0 = (k == 0) ? wg_k * uc : wg_k * (0.5 * (uc + ukm));
        const double fv0 = (k == 0) ? wg_k * vc : wg_k * (0.5 * (vc + vkm));
        const double ft0 = (k == 0) ? wg_k * tc : wg_k * (0.5 * (tc + tkm));
        const double fu1 = (k + 1 < L) ? wg_n * (0.5 * (ukp + uc)) : wg_n * uc;
        const double fv1 = (k + 1 < L) ? wg_n * (0.5 * (vkp + vc)) : wg_n * vc;
        const double ft1 = (k + 1 < L) ? wg_n * (0.5 * (tkp + tc)) : wg_n * tc;
        if (!(a.vadv_skip & 1)) {
          dt_u = dt_u + (-(fu1 - fu0 - uc * dw) / dp);
          dt_v = dt_v + (-(fv1 - fv0 - vc * dw) / dp);
        }
        if (!(a.vadv_skip & 2)) dt_t = dt_t + (-(ft1 - ft0 - tc * dw) / dp);
      }
      // ---- horizontal T advection (transforms.F90:828), vorticity/Coriolis terms (:895-896)
      dt_t = dt_t - uc * dxti - vc * dyti;
      const double av = voi + cor;
      dt_u = dt_u + av * vc;
      dt_v = dt_v - av * uc;
      a.dtu[q] = dt_u * cosm;
      a.dtv[q] = dt_v * cosm;
      a.dtT[q] = dt_t;
      wg_k = wg_n;
    }
  }
  if (w == NW - 1) a.dtlp[c2] = (0.0 - total) / ps;     // (dt_psg - dmean_tot)/psg (:873, :1102)
  // ---- hydrostatic integral bottom-up within the chunk, Phi + KE (:350-356, :902)
  {
    double gh = below + a.surf_geop[c2];
#pragma unroll
    for (int i = CH - 1; i >= 0; --i) {
      if (i < nk) {
        const size_t q = c2 + (size_t)(k0 + i) * lev;
        a.E[q] = gh + RDGAS * TV(i) * (lph[i + 1] - lpf[i]) + .5 * (u[i] * u[i] + v[i] * v[i]);
        if (k0 + i >= ktop) gh = gh + RDGAS * TV(i) * (lph[i + 1] - lph[i]);
      }
    }
  }
  // ---- block partial sums: mean_surf_press_previous, mean_energy_previous
  double s_en = wts_j * e_prev, s_ps = (w == 0) ? wts_j * psp : 0.0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    s_ps += __shfl_down(s_ps, off, 64);
    s_en += __shfl_down(s_en, off, 64);
  }
  if (tid == 0) lds_e[w] = s_en;
  lds_cnt[w * 64 + tid] = nbelow;
  __syncthreads();
  if (a.kmask && w == 0) {
    int cnt = 0;
    for (int ww = 0; ww < NW; ++ww) cnt += lds_cnt[ww * 64 + tid];
    a.kmask[c2] = (kmw_old << 8) | cnt;
  }
  if (threadIdx.x == 0) {
    double e = 0.0;
    for (int ww = 0; ww < NW; ++ww) e += lds_e[ww];
    a.partials[2 * blockIdx.x] = s_ps;
    a.partials[2 * blockIdx.x + 1] = e;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel for PURE SIGMA levels (pk = 0; the Held-Suarez and Frierson test cases' 'uneven_sigma' / 'even_sigma').  Every pressure of a
// column is then b x p_s, and what pressure_variables (press_and_geopot.F90:165-194) forms with a logarithm per half level -- ln p_half, the
// Simmons-Burridge ln p_full, their differences in four_in_one (spectral_dynamics.F90:1064-1083) and compute_geopotential (:350-356) -- differs
// from ln p_s by constants of the vertical coordinate (ColumnArgs::sig, api.hip "col_sig"): ONE logarithm and ONE exponential per column (for
// hs_forcing's (p/P00)**kappa) instead of a logarithm, two exponentials and seven divisions per level.  k_column spent 2 550 vector instructions
// per wavefront on those (SQ_INSTS_VALU, profiles/r05_T85L40_pmc_summary.csv) -- 19 us of issue time at T85L40 which, with one block per CU in
// lock step around its barrier, did not overlap the block's memory phase: a shard of 64 blocks took the same 21 us as a full round of 256
// (profiles/r06_T85L40_P8_kernel_stats.csv).  Mathematically the same expressions; the roundings differ (fewer of them), as they already do
// between this device's log / exp and the host's.
// sig[k] = {d1 = ln p_half(k+1) - ln p_full(k), d3 = ln p_half(k+1) - ln p_half(k), x1c = (b(k+1) d1 + b(k) d2) / db, 1/db, cf = p_full/p_s, cf**kappa,
//           ln cf, hs_forcing's Rayleigh factor and boundary-layer part of the Newtonian rate at sigma = cf, db, b(k+1), b(k)}
// ---------------------------------------------------------------------------------------------------------------------
// TWO: two blocks per CU (<= 128 registers): the six fields that are only needed below the barrier (previous u, v, T, vorticity, the two T gradients) are
// then requested below it -- a second memory round trip per block, which the CU's other block covers
template <int CH, bool EXT, bool VIRT, bool TWO = false>
__global__ __launch_bounds__(512, TWO ? 4 : 2) void k_column_sig(Geom g, ColumnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L = g.L, I = g.I;
  const int tid = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
  double *lds_dm = (double *)smem;          // [NW][64] chunk sums of dmean
  double *lds_a = lds_dm + NW * 64;         // [NW][64] chunk sums of RDGAS*T*dlog3
  double *lds_e = lds_a + NW * 64;          // [NW] energy partials
  int *lds_cnt = (int *)(lds_e + NW);       // [NW][64] levels with p_full < water_correction_limit
  const int col = blockIdx.x * 64 + tid;
  const int jl = col / I;
  const size_t c2 = (size_t)col, lev = (size_t)g.Jl * I;
  const int k0 = w * CH, nk = min(CH, L - k0);
  kdouble *sg = (kdouble *)a.sig + 16 * k0;              // constant address space: scalar loads wherever they are needed, also behind the stores
  double fac_c = 1.0, tc_c = 0.0;
#ifdef ISCA_EXPERIMENTS      // the deferred finish (HISTORY "Round 6": correct, and no faster than the one-block kernel once its cost to this kernel is counted)
  __shared__ double fin_sh[FT_GROUPS][16];
  if (a.fin_seq && blockIdx.x == 0) {                     // the deferred finish of the step before: this block computes and publishes
    double wf;
    fixer_finish_body(g, a.fin->fa[a.fin_fut], fin_sh, fac_c, tc_c, wf);
    if (threadIdx.x == 0) {
      // (a blown-up run's scalars are NaN, which is what an empty slot reads: published as infinities -- the waiting blocks go on, and the
      // host's valid-range check sees the NaN in red[16..17])
      fin_publish(a.fin->val[a.fin_seq & 1], fac_c == fac_c ? fac_c : INFINITY, tc_c == tc_c ? tc_c : INFINITY);
      fin_publish(a.fin->val[~a.fin_seq & 1], __builtin_nan(""), __builtin_nan(""));      // the next launch's slot: empty until its block 0 fills it
    }
  }
#endif
  const double tc_p = a.pend_p[PEND_TCORR];
  double ps = a.ps[c2];
  const double psp = mul_nc(a.psp[c2], a.pend_p[PEND_FACTOR]);
  const int kmw_old = a.kmask_rd[c2];
  const double dxl = a.dxlp[c2], dyl = a.dylp[c2];
  constexpr int ktop = 1;                                   // pk(1) = 0: the hydrostatic sum starts at the second level (press_and_geopot.F90:341-349)
  const double wts_j = a.wts[jl], cosm = a.cosm[jl], cor = a.coriolis[jl], sin_lat = a.hs_sin[jl];
  double u[CH], v[CH], t[CH], dm[CH];
  double tvv[VIRT ? CH : 1];
#define TV(i) (VIRT ? tvv[VIRT ? (i) : 0] : t[i])
  constexpr bool EARLY = CH <= 5 && !TWO;
  double upv[EARLY ? CH : 1], vpv[EARLY ? CH : 1], tpv[EARLY ? CH : 1], vov[EARLY ? CH : 1], dxv[EARLY ? CH : 1], dyv[EARLY ? CH : 1];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int k = k0 + (i < nk ? i : 0);
    const size_t q = c2 + (size_t)k * lev;
    u[i] = a.u[q]; v[i] = a.v[q]; t[i] = a.t[q];
    if (VIRT) tvv[VIRT ? i : 0] = a.tv[q];
    if (EARLY) { upv[i] = a.up[q]; vpv[i] = a.vp[q]; tpv[i] = a.tp[q] + tc_p; vov[i] = a.vor[q]; dxv[i] = a.dxT[q]; dyv[i] = a.dyT[q]; }
    dm[i] = a.div[q];
  }
  double um = 0., vm = 0., tm = 0., un = 0., vn = 0., tn = 0.;
  if (k0 > 0) { const size_t q = c2 + (size_t)(k0 - 1) * lev; um = a.u[q]; vm = a.v[q]; tm = a.t[q]; }
  if (k0 + nk < L) { const size_t q = c2 + (size_t)(k0 + nk) * lev; un = a.u[q]; vn = a.v[q]; tn = a.t[q]; }
  // In front of the barrier: what does not need the current level's pending scalars -- with p_s = f p_s(stored) the layer's mass divergence is p_s x dmr,
  // dmr = db (div + u dln p_s/dx + v dln p_s/dy) --, so that a deferred finish (below) is only waited for behind the barrier, with every load, the
  // chunk sums of dmr and their exchange through LDS done meanwhile.  (The hydrostatic chunk sums need the corrected temperature itself -- T(stored) + c
  // summed as such, or a run restarted from materialised fields would differ in the last bit --: they are exchanged behind the level loop.)
  double csum = 0.0;
#pragma unroll
  for (int i = 0; i < CH; ++i) {          // mass divergence of the layer over p_s (four_in_one :1064-1067), dp = db p_s
    const double dbk = sg[16 * i + 9];
    dm[i] = (i < nk) ? dbk * (dm[i] + (u[i] * dxl + v[i] * dyl)) : 0.0;
    csum += dm[i];
  }
  lds_dm[w * 64 + tid] = csum;
  __syncthreads();
  double base = 0.0, total = 0.0;
  for (int ww = 0; ww < NW; ++ww) {
    const double x = lds_dm[ww * 64 + tid];
    total += x;
    if (ww < w) base += x;
  }
  // the current level's pending scalars: from pend_c, or -- deferred finish -- from block 0 of this launch
#ifdef ISCA_EXPERIMENTS
  if (!a.fin_seq) { fac_c = a.pend_c[PEND_FACTOR]; tc_c = a.pend_c[PEND_TCORR]; }
  else if (blockIdx.x != 0) {
    double f = 1.0, tcv = 0.0;
    if (tid == 0) {
      const long long t0 = wall_clock64();
      const double *slot = a.fin->val[a.fin_seq & 1];
      for (;;) {
        fin_peek(slot, f, tcv);
        if (f == f && tcv == tcv) break;                                      // both halves of the pair have arrived
        if (wall_clock64() - t0 > 300000000) { f = 1.0; tcv = 0.0; a.fin->fa[0].red[25] = 1.0; break; }      // 3 s of the 100 MHz counter (processes that share the
                                                                              // device are time-sliced for milliseconds): block 0 never ran in
        __builtin_amdgcn_s_sleep(4);                                          // front of this one; raised at the host's next synchronisation (api.hip)
      }
    }
    fac_c = __shfl(f, 0, 64); tc_c = __shfl(tcv, 0, 64);
  }
#else
  fac_c = a.pend_c[PEND_FACTOR]; tc_c = a.pend_c[PEND_TCORR];
#endif
  ps = mul_nc(ps, fac_c);
#pragma unroll
  for (int i = 0; i < CH; ++i) { t[i] += tc_c; dm[i] *= ps; }
  tm += tc_c; tn += tc_c;
  base *= ps; total *= ps;
  {   // the hydrostatic chunk sum of my levels (compute_geopotential :350-356), read by the wavefronts above behind the level loop
    double asum = 0.0;
#pragma unroll
    for (int i = 0; i < CH; ++i) asum += (i < nk && k0 + i >= ktop) ? RDGAS * TV(i) * sg[16 * i + 1] : 0.0;
    lds_a[w * 64 + tid] = asum;
  }
  const double rps = 1. / ps;
  double lpn0 = 0.0, pkap = 0.0;
  if (!EXT) { lpn0 = log(ps) - a.lnP00; pkap = exp(KAPPA * lpn0); }     // ln(p_s/P00), (p_s/P00)**kappa
  const double sin2 = sin_lat * sin_lat, cos2 = 1.0 - sin2, cos4 = cos2 * cos2;
  const double t_star = a.t_zero - a.delh * sin2 - a.eps * sin_lat, tstr = a.t_strat - a.eps * sin_lat;
  double dmean_tot = base;
  double wg_k = (k0 == 0) ? 0.0 : (-base + total * sg[11]);
  double e_prev = 0.0;
  int nbelow = 0;
  if (a.wg && w == 0) { a.wg[c2] = 0.0; a.psp_copy[c2] = psp; }
  constexpr double RCP = 1.0 / CP_AIR;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    if (i < nk) {
      const int k = k0 + i;
      const size_t q = c2 + (size_t)k * lev;
      const double d1 = sg[16 * i + 0], d3 = sg[16 * i + 1], x1c = sg[16 * i + 2], rdbk = sg[16 * i + 3], cf = sg[16 * i + 4];
      const double dbk = sg[16 * i + 9], bkn = sg[16 * i + 10];
      const double upi = EARLY ? upv[EARLY ? i : 0] : a.up[q], vpi = EARLY ? vpv[EARLY ? i : 0] : a.vp[q], tpi = EARLY ? tpv[EARLY ? i : 0] : a.tp[q] + tc_p;
      const double voi = EARLY ? vov[EARLY ? i : 0] : a.vor[q], dxti = EARLY ? dxv[EARLY ? i : 0] : a.dxT[q], dyti = EARLY ? dyv[EARLY ? i : 0] : a.dyT[q];
      const double p_full = cf * ps;
      double dt_u, dt_v, dt_t;
      if (EXT) {
        dt_u = a.phu[q]; dt_v = a.phv[q]; dt_t = a.pht[q];
      } else {
        // ---- hs_forcing at the previous level (rayleigh :615-679, dissipative heating :198-200, newtonian :508-611); sigma = cf
        const double vfactr = sg[16 * i + 7];
        dt_u = vfactr * upi; dt_v = vfactr * vpi; dt_t = 0.0;
        if (a.do_conserve_energy) dt_t = -((upi + .5 * dt_u * a.delta_t) * dt_u + (vpi + .5 * dt_v * a.delta_t) * dt_v) * RCP;
        const double lpn = lpn0 + sg[16 * i + 6];              // log(p_full/P00)
        const double the = t_star - a.delv * cos2 * lpn;
        const double teq = fmax(the * (pkap * sg[16 * i + 5]), tstr);   // (p_full/P00)**kappa
        const double tdamp = a.tka + cos4 * sg[16 * i + 8];
        dt_t = dt_t + (-tdamp * (tpi - teq));
      }
      {  // initialize_corrections (:1318-1321)
        const double ue = upi + dt_u * a.delta_t, ve = vpi + dt_v * a.delta_t;
        e_prev += (0.5 * (ue * ue + ve * ve) + CP_AIR * (tpi + dt_t * a.delta_t)) * (dbk * psp);
      }
      // ---- four_in_one (:1064-1083): x1 dp_s/dx = x1c dln p_s/dx
      const double dp_inv = rdbk * rps;
      const double x2 = x1c * dxl, x3 = x1c * dyl;
      const double uc = u[i], vc = v[i], tc = t[i], tvc = TV(i);
      dt_u = dt_u - RDGAS * tvc * x2;
      dt_v = dt_v - RDGAS * tvc * x3;
      const double x4 = (dmean_tot * d3 + dm[i] * d1) * dp_inv;
      const double x5 = x4 - uc * x2 - vc * x3;
      dt_t = dt_t - KAPPA * tvc * x5;
      if (a.store_wg_full) a.wg_full[q] = -x5 * p_full;
      nbelow += (p_full < a.water_limit) ? 1 : 0;
      dmean_tot = dmean_tot + dm[i];
      const double wg_n = (k + 1 < L) ? (-dmean_tot + total * bkn) : 0.0;
      if (a.wg) a.wg[q + lev] = wg_n;
      // ---- vert_advection SECOND_CENTERED / ADVECTIVE_FORM (vert_advection.F90:185-193, 467-470)
      const double ukm = (i == 0) ? um : u[i > 0 ? i - 1 : 0], vkm = (i == 0) ? vm : v[i > 0 ? i - 1 : 0], tkm = (i == 0) ? tm : t[i > 0 ? i - 1 : 0];
      const double ukp = (i == nk - 1) ? un : u[i + 1 < CH ? i + 1 : i], vkp = (i == nk - 1) ? vn : v[i + 1 < CH ? i + 1 : i], tkp = (i == nk - 1) ? tn : t[i + 1 < CH ? i + 1 : i];
      {
        const double dw = wg_n - wg_k;
        const double fu0 = (k == 0) ? wg_k * uc : wg_k * (0.5 * (uc + ukm));
        const double fv0 = (k == 0) ? wg_k * vc : wg_k * (0.5 * (vc + vkm));
        const double ft0 = (k == 0) ? wg_k * tc : wg_k * (0.5 * (tc + tkm));
        const double fu1 = (k + 1 < L) ? wg_n * (0.5 * (ukp + uc)) : wg_n * uc;
        const double fv1 = (k + 1 < L) ? wg_n * (0.5 * (vkp + vc)) : wg_n * vc;
        const double ft1 = (k + 1 < L) ? wg_n * (0.5 * (tkp + tc)) : wg_n * tc;
        if (!(a.vadv_skip & 1)) {
          dt_u = dt_u + (-(fu1 - fu0 - uc * dw) * dp_inv);
          dt_v = dt_v + (-(fv1 - fv0 - vc * dw) * dp_inv);
        }
        if (!(a.vadv_skip & 2)) dt_t = dt_t + (-(ft1 - ft0 - tc * dw) * dp_inv);
      }
      // ---- horizontal T advection (transforms.F90:828), vorticity/Coriolis terms (:895-896)
      dt_t = dt_t - uc * dxti - vc * dyti;
      const double av = voi + cor;
      dt_u = dt_u + av * vc;
      dt_v = dt_v - av * uc;
      a.dtu[q] = dt_u * cosm;
      a.dtv[q] = dt_v * cosm;
      a.dtT[q] = dt_t;
      wg_k = wg_n;
    }
  }
  if (w == NW - 1) a.dtlp[c2] = (0.0 - total) * rps;    // (dt_psg - dmean_tot)/psg (:873, :1102)
  // ---- hydrostatic integral bottom-up within the chunk, Phi + KE (:350-356, :902)
  __syncthreads();                                        // (lds_a of every wavefront)
  {
    double below = 0.0;
    for (int ww = w + 1; ww < NW; ++ww) below += lds_a[ww * 64 + tid];
    double gh = below + a.surf_geop[c2];
#pragma unroll
    for (int i = CH - 1; i >= 0; --i) {
      if (i < nk) {
        const size_t q = c2 + (size_t)(k0 + i) * lev;
        a.E[q] = gh + RDGAS * TV(i) * sg[16 * i + 0] + .5 * (u[i] * u[i] + v[i] * v[i]);
        if (k0 + i >= ktop) gh = gh + RDGAS * TV(i) * sg[16 * i + 1];
      }
    }
  }
  // ---- block partial sums: mean_surf_press_previous, mean_energy_previous
  double s_en = wts_j * e_prev, s_ps = (w == 0) ? wts_j * psp : 0.0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    s_ps += __shfl_down(s_ps, off, 64);
    s_en += __shfl_down(s_en, off, 64);
  }
  if (tid == 0) lds_e[w] = s_en;
  lds_cnt[w * 64 + tid] = nbelow;
  __syncthreads();
  if (a.kmask && w == 0) {
    int cnt = 0;
    for (int ww = 0; ww < NW; ++ww) cnt += lds_cnt[ww * 64 + tid];
    a.kmask[c2] = (kmw_old << 8) | cnt;
  }
  if (threadIdx.x == 0) {
    double e = 0.0;
    for (int ww = 0; ww < NW; ++ww) e += lds_e[ww];
    a.partials[2 * blockIdx.x] = s_ps;
    a.partials[2 * blockIdx.x + 1] = e;
  }
}

#undef TV

// the two FixerArgs a deferred finish can have (future level 0 or 1; the current level is the other one), on the device: ColumnArgs::fin
size_t deferred_fixer_args_bytes() { return sizeof(DeferredFin); }
void upload_deferred_fixer_args(const isca_dyn &h) {
  DeferredFin df{};
  for (auto &slot : df.val) slot[0] = slot[1] = std::nan("");                 // both slots empty
  for (int fut = 0; fut < 2; ++fut) { StepScalars sc{}; sc.fut = fut; sc.cur = 1 - fut; sc.prev = fut; df.fa[fut] = fixer_args(h, sc); }
  (void)hipMemcpy(h.d.fin_args, &df, sizeof(df), hipMemcpyHostToDevice);
}
bool column_takes_deferred_finish(const isca_dyn &h) {
  // Experiments build only, and only when asked for (ISCA_DEFERRED_FINISH=1): measured in round 6 -- block 0 of the column kernel computing the fixers'
  // scalars while the other blocks wait for them is 5 us faster than the same kernel followed by k_fixer_finish, but the waiting code costs the
  // 128-register variant 4 spilled registers and 4-5 us whether it waits or not; against the kernel WITHOUT that code the step is the same.
  if (!exp_env("ISCA_DEFERRED_FINISH")) return false;
  return h.d.col_sig && h.cfg.vert_difference_option != 1 && !virtual_t_on(h) && h.cfg.physics == 0 && !hs_forcing_separate(h) && h.lazy_fix;
}
size_t column_partials_count(const isca_dyn &h) { return (size_t)h.g.Jl * h.g.I / 64; }

// virtual_t = T (1 + (rvgas/rdgas - 1) q) (spectral_dynamics.F90:436-438, 858; press_and_geopot.F90:248, 342)
__global__ void k_virtual_t(size_t n, const double *__restrict__ t, const double *__restrict__ q, double *__restrict__ tv) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tv[i] = t[i] * (1.0 + (RVGAS / RDGAS - 1.0) * q[i]);
}
bool virtual_t_on(const isca_dyn &h) { return h.cfg.use_virtual_temperature && h.tracer_on; }    // `.not. dry_model`
void launch_virtual_t(const isca_dyn &h, const double *t, const double *q, double *tv, hipStream_t s) {
  const size_t n = (size_t)h.g.L * h.g.Jl * h.g.I;
  hipLaunchKernelGGL(k_virtual_t, grid1d(n), dim3(256), 0, s, n, t, q, tv);
}
bool hs_forcing_separate(const isca_dyn &h) { return h.cfg.physics == 0 && h.cfg.local_heating_option != 0; }
void launch_column(const isca_dyn &h, const StepScalars &sc, hipStream_t s) {
  const Geom &g = h.g;
  const Dev &d = h.d;
  ColumnArgs a;
  a.u = d.ug[sc.cur]; a.v = d.vg[sc.cur]; a.t = d.tg[sc.cur]; a.ps = d.psg[sc.cur];
  a.up = d.ug[sc.prev]; a.vp = d.vg[sc.prev]; a.tp = d.tg[sc.prev]; a.psp = d.psg[sc.prev];
  a.vor = d.vorg; a.div = d.divg; a.dxT = d.dxT; a.dyT = d.dyT; a.dxlp = d.dxlp; a.dylp = d.dylp;
  a.dtu = d.g_dtu; a.dtv = d.g_dtv; a.dtT = d.g_dtT; a.E = d.g_E; a.dtlp = d.g_dtlp; a.wg_full = d.wg_full;
  a.partials = d.partials;
  a.pk = d.pk; a.bk = d.bk; a.dpk = d.dpk; a.dbk = d.dbk; a.cosm = d.cosm_lat_l; a.coriolis = d.coriolis_l;
  a.rad_lat = d.rad_lat_l; a.wts = d.wts_lat_l;
  a.delta_t = sc.delta_t; a.tka = h.tab.tka; a.tks = h.tab.tks; a.vkf = h.tab.vkf; a.sigma_b = h.cfg.sigma_b;
  a.t_zero = h.cfg.t_zero; a.delh = h.cfg.delh; a.delv = h.cfg.delv; a.eps = h.cfg.eps; a.t_strat = h.cfg.t_strat;
  a.P00 = h.cfg.P00; a.do_conserve_energy = h.cfg.do_conserve_energy;
  a.vadv_skip = (h.cfg.vert_advect_uv != 0 ? 1 : 0) | (h.cfg.vert_advect_t != 0 ? 2 : 0);
  a.wg = (h.tracer_on || a.vadv_skip) ? d.wg : nullptr; a.kmask = h.tracer_on ? d.kmask : nullptr; a.kmask_rd = d.kmask_old; a.psp_copy = d.psp_copy;
  a.water_limit = h.cfg.water_correction_limit;
  a.phu = d.ph_dtu; a.phv = d.ph_dtv; a.pht = d.ph_dtT; a.surf_geop = d.surf_geop;
  a.pend_c = d.pend + 4 * sc.cur; a.pend_p = d.pend + 4 * sc.prev;     // identity rows unless a level's fixers are pending (lazy fixers)
  a.store_wg_full = sc.store_wg_full;
  const int CH = (g.L + 7) / 8;                 // <= 8 wavefronts per block, CH levels each
  const int NW = (g.L + CH - 1) / CH;
  const size_t lds = (size_t)(2 * NW * 64 + NW) * sizeof(double) + (size_t)NW * 64 * sizeof(int);
  const dim3 grid((unsigned)column_partials_count(h)), block(64 * NW);
  a.tv = virtual_t_on(h) ? d.tv : nullptr;
  const bool ext = h.cfg.physics != 0 || hs_forcing_separate(h);      // the physics tendencies come from arrays (a package's, or k_hs_forcing_step's)
  if (a.tv) launch_virtual_t(h, a.t, d.tr[sc.cur], d.tv, s);       // grid_tracers(:,:,:,current,nhum) (spectral_dynamics.F90:858)
  a.sig = d.col_sig; a.hs_sin = d.hs_sin_l; a.lnP00 = std::log(h.cfg.P00);
  a.fin_seq = 0; a.fin_fut = 0; a.fin = (DeferredFin *)d.fin_args;
  if (h.fin_deferred) { a.fin_fut = h.fin_fut; a.fin_seq = h.fin_seq; }       // (api.hip has flushed it unless this launch is the kernel that takes it; it counts the launches)
  if (a.sig && h.cfg.vert_difference_option != 1) {             // pure sigma levels: the per-level logarithms are constants of the coordinate
    // Two blocks per CU (k_column_sig<.., TWO>: <= 128 registers, the six below-the-barrier fields requested there) for the plain Held-Suarez
    // instantiation with chunks of <= 5 levels, when the grid has at least two blocks per CU to interleave: T85L40 on one rank 40.0 -> 37.0 us
    // (HIP events; step 0.1769 -> 0.1735 ms, two A/B pairs in one gpurun call).  A smaller grid (a shard, T42) is one round of blocks whatever the
    // occupancy, and there the second memory round trip only adds latency.  ISCA_COLUMN_TWO=0|1 overrides (measurement).
    static const int two_env = exp_env("ISCA_COLUMN_TWO") ? atoi(exp_env("ISCA_COLUMN_TWO")) : -1;
    const bool two = two_env >= 0 ? two_env != 0 : (grid.x >= 512 && CH <= 5);
    if (two && !a.tv) {
#define LT2(N) do { if (ext) hipLaunchKernelGGL((k_column_sig<N, true, false, true>), grid, block, lds, s, g, a); \
                    else hipLaunchKernelGGL((k_column_sig<N, false, false, true>), grid, block, lds, s, g, a); } while (0)
      switch (CH) {
        case 1: LT2(1); break; case 2: LT2(2); break; case 3: LT2(3); break; case 4: LT2(4); break; case 5: LT2(5); break;
#ifdef ISCA_EXPERIMENTS
        case 6: LT2(6); break; case 7: LT2(7); break; default: LT2(8); break;      // (6..8 levels per wavefront spill 11-32 registers at 128: T170L60 229-234 against 191 us)
#else
        default: break;
#endif
      }
#undef LT2
      if (CH <= 5 || two_env >= 0) return;
    }
#define LS(N) do { \
    if (a.tv) { if (ext) hipLaunchKernelGGL((k_column_sig<N, true, true>), grid, block, lds, s, g, a); else hipLaunchKernelGGL((k_column_sig<N, false, true>), grid, block, lds, s, g, a); } \
    else if (ext) hipLaunchKernelGGL((k_column_sig<N, true, false>), grid, block, lds, s, g, a); \
    else hipLaunchKernelGGL((k_column_sig<N, false, false>), grid, block, lds, s, g, a); } while (0)
    switch (CH) {
      case 1: LS(1); break; case 2: LS(2); break; case 3: LS(3); break; case 4: LS(4); break;
      case 5: LS(5); break; case 6: LS(6); break; case 7: LS(7); break; default: LS(8); break;
    }
#undef LS
    return;
  }
#define LC(N) do { \
    if (h.cfg.vert_difference_option == 1) { \
      if (a.tv) { if (ext) hipLaunchKernelGGL((k_column<N, true, true, true>), grid, block, lds, s, g, a); else hipLaunchKernelGGL((k_column<N, false, true, true>), grid, block, lds, s, g, a); } \
      else if (ext) hipLaunchKernelGGL((k_column<N, true, false, true>), grid, block, lds, s, g, a); \
      else hipLaunchKernelGGL((k_column<N, false, false, true>), grid, block, lds, s, g, a); } \
    else if (a.tv) { if (ext) hipLaunchKernelGGL((k_column<N, true, true>), grid, block, lds, s, g, a); else hipLaunchKernelGGL((k_column<N, false, true>), grid, block, lds, s, g, a); } \
    else if (ext) hipLaunchKernelGGL((k_column<N, true, false>), grid, block, lds, s, g, a); \
    else hipLaunchKernelGGL((k_column<N, false, false>), grid, block, lds, s, g, a); } while (0)
  // The general kernel (hybrid levels, 'mcm') is instantiated for even chunk sizes only -- it is the fallback since round 6, and its 64 variants were
  // 1.4 MB of code: an odd ceil(L / 8) takes the next even chunk and fewer wavefronts.
  {
    const int CHg = (CH + 1) & ~1, NWg = (g.L + CHg - 1) / CHg;
    const size_t lds = (size_t)(2 * NWg * 64 + NWg) * sizeof(double) + (size_t)NWg * 64 * sizeof(int);
    const dim3 block(64 * NWg);
    switch (CHg) { case 2: LC(2); break; case 4: LC(4); break; case 6: LC(6); break; default: LC(8); break; }
  }
#undef LC
}

// ---- vert_advection with the schemes the column kernel does not fuse (vert_advection.F90:69-478; spectral_dynamics.F90:877-888):
// FOURTH_CENTERED (uniform-spacing form, second order next to the top and the ground), FINITE_VOLUME_LINEAR = VAN_LEER_LINEAR (slope_z
// with limiters, upstream value at the half time step), FINITE_VOLUME_PARABOLIC (compute_weights + slope_z(linear = .false.), Colella &
// Woodward's monotonicity constraint, the extension for Courant numbers > 1), all in ADVECTIVE_FORM:
//   rdt = -(flux(k+1) - flux(k) - r (w(k+1) - w(k))) / dz ,  dz = dpk + dbk ps of the CURRENT level (:876), w = the column kernel's wg.
// One thread per column, the column's values in private arrays: an option path (one more launch, ~4 field passes), not the default.
constexpr int VADV_MAXL = 64;
template <int SCHEME>
__device__ void vadv_column(int L, double dt, const double *w, const double *dz, const double *r, double *rdt) {
  double flux[VADV_MAXL + 1];
  flux[0] = w[0] * r[0];
  flux[L] = w[L] * r[L - 1];
  if (SCHEME == 0) {                 // SECOND_CENTERED (:175-180)
    for (int k = 1; k <= L - 1; ++k) flux[k] = w[k] * (0.5 * (r[k] + r[k - 1]));
  } else if (SCHEME == 1) {          // FOURTH_CENTERED
    const double c1 = 7. / 12., c2 = 1. / 12.;
    for (int k = 2; k <= L - 2; ++k) flux[k] = w[k] * (c1 * (r[k] + r[k - 1]) - c2 * (r[k + 1] + r[k - 2]));
    flux[1] = w[1] * (0.5 * (r[1] + r[0]));
    flux[L - 1] = w[L - 1] * (0.5 * (r[L - 1] + r[L - 2]));
  } else {
    // slope_z: SCHEME 2 linear = .true., SCHEME 3 linear = .false.; limiters on in both
    double slp[VADV_MAXL];
    for (int k = 1; k <= L - 2; ++k) {
      const double gk = (r[k] - r[k - 1]) / (dz[k] + dz[k - 1]), gk1 = (r[k + 1] - r[k]) / (dz[k + 1] + dz[k]);
      double s;
      if (SCHEME == 2) s = (gk1 + gk) * dz[k];
      else s = (gk1 * (2. * dz[k - 1] + dz[k]) + gk * (2. * dz[k + 1] + dz[k])) * dz[k] / (dz[k - 1] + dz[k] + dz[k + 1]);
      const double rmin = fmin(fmin(r[k - 1], r[k]), r[k + 1]), rmax = fmax(fmax(r[k - 1], r[k]), r[k + 1]);
      slp[k] = copysign(1.0, s) * fmin(fmin(fabs(s), 2. * (r[k] - rmin)), 2. * (rmax - r[k]));
    }
    slp[0] = 0.0; slp[L - 1] = 0.0;
    if (SCHEME == 2) {
      for (int k = 1; k <= L - 1; ++k) {
        double rst;
        if (w[k] >= 0.) { const double cn = dt * w[k] / dz[k - 1]; rst = r[k - 1] + 0.5 * slp[k - 1] * (1. - cn); }
        else { const double cn = -dt * w[k] / dz[k]; rst = r[k] - 0.5 * slp[k] * (1. - cn); }
        flux[k] = w[k] * rst;
      }
    } else {
      double rl[VADV_MAXL], rr[VADV_MAXL];
      for (int k = 2; k <= L - 2; ++k) {            // compute_weights (:600-645) on the fly; the interface between layers k-1 and k
        const double d1 = 1.0 / (dz[k - 1] + dz[k]), d2 = 1.0 / (dz[k - 2] + dz[k - 1] + dz[k] + dz[k + 1]);
        const double d3 = 1.0 / (2 * dz[k - 1] + dz[k]), d4 = 1.0 / (dz[k - 1] + 2 * dz[k]);
        const double n3 = dz[k - 2] + dz[k - 1], n4 = dz[k] + dz[k + 1];
        const double x = n3 * d3 - n4 * d4, y = 2.0 * dz[k - 1] * dz[k];
        const double z0 = dz[k - 1] * d1, z1 = z0 + x * y * d1 * d2, z2 = dz[k - 1] * n3 * d3 * d2, z3 = dz[k] * n4 * d4 * d2;
        rl[k] = r[k - 1] + z1 * (r[k] - r[k - 1]) - z2 * slp[k] + z3 * slp[k - 1];
        rr[k - 1] = rl[k];
      }
      rl[1] = r[1] - 0.5 * slp[1]; rr[L - 2] = r[L - 2] + 0.5 * slp[L - 2];
      rl[0] = r[0] - 0.5 * slp[0]; rr[0] = r[0] + 0.5 * slp[0];
      rl[L - 1] = r[L - 1] - 0.5 * slp[L - 1]; rr[L - 1] = r[L - 1] + 0.5 * slp[L - 1];
      for (int k = 0; k < L; ++k) {                  // Colella and Woodward (1984), Equation 1.10
        if ((rr[k] - r[k]) * (r[k] - rl[k]) <= 0.0) { rl[k] = r[k]; rr[k] = r[k]; }
        if (k == 0 || k == L - 1) continue;
        const double rm = rr[k] - rl[k], aa = rm * (r[k] - 0.5 * (rr[k] + rl[k])), bb = rm * rm / 6.;
        if (aa > bb) rl[k] = 3.0 * r[k] - 2.0 * rr[k];
        if (aa < -bb) rr[k] = 3.0 * r[k] - 2.0 * rl[k];
      }
      const double tt = 2. / 3.;
      for (int k = 1; k <= L - 1; ++k) {
        double rst, xx, cn, rsum = 0.;
        int kk;
        if (w[k] >= 0.) {
          cn = dt * w[k] / dz[k - 1]; kk = k - 1;
          if (cn > 1.) {
            double dzsum = 0.; const double dtw = dt * w[k];
            while (dzsum + dz[kk] < dtw) { if (kk == 0) break; dzsum += dz[kk]; rsum += r[kk]; --kk; }
            xx = (dtw - dzsum) / dz[kk];
          } else xx = cn;
          const double rm = rr[kk] - rl[kk];
          double r6 = 6.0 * (r[kk] - 0.5 * (rr[kk] + rl[kk]));
          if (kk == 0) r6 = 0.;
          rst = rr[kk] - 0.5 * xx * (rm - (1.0 - tt * xx) * r6);
        } else {
          cn = -dt * w[k] / dz[k]; kk = k;
          if (cn > 1.) {
            double dzsum = 0.; const double dtw = -dt * w[k];
            // (:414 reads `if (kk == ks) exit` in this branch too, which never stops a downward walk; a flow that crosses the ground within one
            //  step walks off the column there -- here it stops at the lowest layer, and valid_range_t reports the blown-up state)
            while (dzsum + dz[kk] < dtw) { if (kk == L - 1) break; dzsum += dz[kk]; rsum += r[kk]; ++kk; }
            xx = (dtw - dzsum) / dz[kk];
          } else xx = cn;
          const double rm = rr[kk] - rl[kk];
          double r6 = 6.0 * (r[kk] - 0.5 * (rr[kk] + rl[kk]));
          if (kk == L - 1) r6 = 0.;
          rst = rl[kk] + 0.5 * xx * (rm + (1.0 - tt * xx) * r6);
        }
        if (cn > 1.) rst = (xx * rst + rsum) / cn;
        flux[k] = w[k] * rst;
      }
    }
  }
  for (int k = 0; k < L; ++k) rdt[k] = -(flux[k + 1] - flux[k] - r[k] * (w[k + 1] - w[k])) / dz[k];
}
struct VadvArgs {
  const double *wg, *ps, *dpk, *dbk, *cosm;
  const double *f[3];      // u, v, T at the time level the scheme works on
  double *dt[3];           // dt_u cos^-1, dt_v cos^-1 (the column kernel's scaling: transforms.F90:764-770), dt_T
  int scheme[3];           // 0: nothing to add (second-centred: done by the column kernel)
  double delta_t;
};
__global__ __launch_bounds__(64) void k_vert_advection_scheme(Geom g, VadvArgs a) {
  const size_t lev = (size_t)g.Jl * g.I, c2 = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (c2 >= lev) return;
  const int L = g.L, jl = (int)(c2 / g.I);
  double w[VADV_MAXL + 1], dz[VADV_MAXL], r[VADV_MAXL], rdt[VADV_MAXL];
  const double ps = a.ps[c2];
  for (int k = 0; k <= L; ++k) w[k] = a.wg[c2 + (size_t)k * lev];
  w[0] = 0.0;                                         // (the column kernel stores the interfaces 1..L; the top one carries no flux)
  for (int k = 0; k < L; ++k) dz[k] = a.dpk[k] + a.dbk[k] * ps;
  for (int f = 0; f < 3; ++f) {
    if (!a.scheme[f]) continue;
    for (int k = 0; k < L; ++k) r[k] = a.f[f][c2 + (size_t)k * lev];
    if (a.scheme[f] == 1) vadv_column<1>(L, a.delta_t, w, dz, r, rdt);
    else if (a.scheme[f] == 2) vadv_column<2>(L, a.delta_t, w, dz, r, rdt);
    else vadv_column<3>(L, a.delta_t, w, dz, r, rdt);
    const double sc = (f < 2) ? a.cosm[jl] : 1.0;
    for (int k = 0; k < L; ++k) { const size_t q = c2 + (size_t)k * lev; a.dt[f][q] = a.dt[f][q] + rdt[k] * sc; }
  }
}
void launch_vert_advection_schemes(const isca_dyn &h, const StepScalars &sc, hipStream_t s) {
  const Geom &g = h.g;
  const Dev &d = h.d;
  if (g.L < 4 || g.L > VADV_MAXL) throw std::runtime_error("vert_advect_uv / vert_advect_t other than second_centered need 4..64 levels");
  VadvArgs a;
  a.wg = d.wg; a.ps = d.psg[sc.cur]; a.dpk = d.dpk; a.dbk = d.dbk; a.cosm = d.cosm_lat_l; a.delta_t = sc.delta_t;
  const int suv = h.cfg.vert_advect_uv, st = h.cfg.vert_advect_t;
  // fourth_centered acts on the current level, the finite-volume schemes on the previous one (spectral_dynamics.F90:878-879, 885-886)
  const int tuv = (suv == 1) ? sc.cur : sc.prev, tt = (st == 1) ? sc.cur : sc.prev;
  a.f[0] = d.ug[tuv]; a.f[1] = d.vg[tuv]; a.f[2] = d.tg[tt];
  a.dt[0] = d.g_dtu; a.dt[1] = d.g_dtv; a.dt[2] = d.g_dtT;
  a.scheme[0] = a.scheme[1] = suv; a.scheme[2] = st;
  const size_t lev = (size_t)g.Jl * g.I;
  hipLaunchKernelGGL(k_vert_advection_scheme, dim3((unsigned)((lev + 63) / 64)), dim3(64), 0, s, g, a);
}

// one field with a scheme other than second_centered: rdt += vert_advection(delta_t, wg, dp(ps), r) -- a 'spectral' tracer's advect_vert
// (spectral_dynamics.F90:1135-1141: fourth_centered on the current level, the finite-volume schemes on the previous one)
void launch_vert_advection_field(const isca_dyn &h, int scheme, const double *ps, const double *r, double *rdt, double delta_t, hipStream_t s) {
  const Geom &g = h.g;
  const Dev &d = h.d;
  if (g.L < 4 || g.L > VADV_MAXL) throw std::runtime_error("advect_vert other than second_centered needs 4..64 levels");
  VadvArgs a;
  a.wg = d.wg; a.ps = ps; a.dpk = d.dpk; a.dbk = d.dbk; a.cosm = d.cosm_lat_l; a.delta_t = delta_t;
  a.f[0] = a.f[1] = nullptr; a.dt[0] = a.dt[1] = nullptr; a.scheme[0] = a.scheme[1] = 0;
  a.f[2] = r; a.dt[2] = rdt; a.scheme[2] = scheme;
  const size_t lev = (size_t)g.Jl * g.I;
  hipLaunchKernelGGL(k_vert_advection_scheme, dim3((unsigned)((lev + 63) / 64)), dim3(64), 0, s, g, a);
}

// standalone hs_forcing on caller fields (for the C-ABI entry point / parity tests)
__global__ void k_hs_forcing(Geom g, ColumnArgs a, double dt, const double *__restrict__ p_half,
                             const double *__restrict__ p_full, const double *__restrict__ u, const double *__restrict__ v,
                             const double *__restrict__ t, double *udt, double *vdt, double *tdt) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lev = (size_t)g.Jl * g.I;
  if (idx >= lev * g.L) return;
  const size_t c2 = idx % lev;
  const int jl = c2 / g.I;
  const double sin_lat = sin(a.rad_lat[jl]);
  const double sin2 = sin_lat * sin_lat, cos2 = 1.0 - sin2, cos4 = cos2 * cos2;
  const double ps = p_half[(size_t)g.L * lev + c2];
  double ut, vt, tt;
  const double lh_xy = a.lh_lon ? a.lh_lon[c2 - (size_t)jl * g.I] * a.lh_lat[jl] : 0.0;
  hs_level(a, dt, ps, p_full[idx], u[idx], v[idx], t[idx], sin_lat, sin2, cos2, cos4, ut, vt, tt, lh_xy);
  udt[idx] += ut; vdt[idx] += vt; tdt[idx] += tt;
}
// hs_forcing of the STEP as a kernel of its own, for hs_forcing_nml options the fused column kernel does not carry (local_heating_option): previous-level
// u, v, T with the current level's pressures (atmosphere.F90:304-311), the tendencies into the arrays the column kernel reads a physics package's from
// (k_column<CH, EXT = true>).  One thread per (column, level); p_full as the column kernel forms it (exp of the Simmons-Burridge ln p_full, or 'mcm').
__global__ void k_hs_forcing_step(Geom g, ColumnArgs a, double dt, const double *__restrict__ psg, const double *__restrict__ u, const double *__restrict__ v,
                                  const double *__restrict__ t, double *udt, double *vdt, double *tdt, int mcm) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lev = (size_t)g.Jl * g.I;
  if (idx >= lev * g.L) return;
  const int k = (int)(idx / lev);
  const size_t c2 = idx - (size_t)k * lev;
  const int jl = c2 / g.I;
  const double sin_lat = sin(a.rad_lat[jl]);
  const double sin2 = sin_lat * sin_lat, cos2 = 1.0 - sin2, cos4 = cos2 * cos2;
  const double ps = psg[c2];
  const double ph_k = a.pk[k] + a.bk[k] * ps, ph_n = a.pk[k + 1] + a.bk[k + 1] * ps;
  double p_full;
  if (mcm) p_full = 0.5 * (ph_n + ph_k);
  else {
    const bool top0 = (a.pk[0] == 0.0 && a.bk[0] == 0.0);
    const double l_n = log(ph_n);
    p_full = (top0 && k == 0) ? exp(l_n - 1.0) : exp(l_n - (1.0 - ph_k * (l_n - log(ph_k)) / (ph_n - ph_k)));
  }
  const double lh_xy = a.lh_lon ? a.lh_lon[c2 - (size_t)jl * g.I] * a.lh_lat[jl] : 0.0;
  double ut, vt, tt;
  hs_level(a, dt, ps, p_full, u[idx], v[idx], t[idx], sin_lat, sin2, cos2, cos4, ut, vt, tt, lh_xy);
  udt[idx] = ut; vdt[idx] = vt; tdt[idx] = tt;
}
void launch_hs_forcing_step(const isca_dyn &h, const StepScalars &sc, hipStream_t s) {
  const Dev &d = h.d;
  ColumnArgs a = {};
  a.rad_lat = d.rad_lat_l; a.pk = d.pk; a.bk = d.bk;
  a.tka = h.tab.tka; a.tks = h.tab.tks; a.vkf = h.tab.vkf; a.sigma_b = h.cfg.sigma_b;
  a.t_zero = h.cfg.t_zero; a.delh = h.cfg.delh; a.delv = h.cfg.delv; a.eps = h.cfg.eps; a.t_strat = h.cfg.t_strat;
  a.P00 = h.cfg.P00; a.do_conserve_energy = h.cfg.do_conserve_energy;
  a.lh_lon = d.lh_lon; a.lh_lat = d.lh_lat_l; a.lh_decay = h.cfg.local_heating_vert_decay;
  hipLaunchKernelGGL(k_hs_forcing_step, grid1d((size_t)h.g.Jl * h.g.I * h.g.L), dim3(256), 0, s, h.g, a, sc.delta_t, d.psg[sc.cur], d.ug[sc.prev], d.vg[sc.prev],
                     d.tg[sc.prev], d.ph_dtu, d.ph_dtv, d.ph_dtT, h.cfg.vert_difference_option == 1 ? 1 : 0);
}
void launch_hs_forcing(const isca_dyn &h, double dt, const double *p_half, const double *p_full, const double *u,
                       const double *v, const double *t, double *udt, double *vdt, double *tdt, hipStream_t s) {
  ColumnArgs a = {};
  a.rad_lat = h.d.rad_lat_l;
  a.tka = h.tab.tka; a.tks = h.tab.tks; a.vkf = h.tab.vkf; a.sigma_b = h.cfg.sigma_b;
  a.t_zero = h.cfg.t_zero; a.delh = h.cfg.delh; a.delv = h.cfg.delv; a.eps = h.cfg.eps; a.t_strat = h.cfg.t_strat;
  a.P00 = h.cfg.P00; a.do_conserve_energy = h.cfg.do_conserve_energy;
  a.lh_lon = h.d.lh_lon; a.lh_lat = h.d.lh_lat_l; a.lh_decay = h.cfg.local_heating_vert_decay;
  hipLaunchKernelGGL(k_hs_forcing, grid1d((size_t)h.g.Jl * h.g.I * h.g.L), dim3(256), 0, s, h.g, a, dt, p_half, p_full, u, v, t, udt, vdt, tdt);
}

// compute_pressures_and_heights (press_and_geopot.F90:363-387)
__global__ void k_pressures_heights(Geom g, const double *__restrict__ pk, const double *__restrict__ bk,
                                    const double *__restrict__ t, const double *__restrict__ psg, const double *__restrict__ surf_geop,
                                    double *p_full, double *p_half, double *z_full, double *z_half, bool mcm) {
  const size_t c2 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lev = (size_t)g.Jl * g.I;
  if (c2 >= lev) return;
  const int L = g.L;
  const double ps = psg[c2];
  const bool top0 = (pk[0] == 0.0 && bk[0] == 0.0);
  // top-down pressures, bottom-up heights recomputing the logs (diagnostic path, not in the step)
  double ph_k = pk[0] + bk[0] * ps;
  p_half[c2] = ph_k;
  for (int k = 0; k < L; ++k) {
    const double ph_n = pk[k + 1] + bk[k + 1] * ps;
    p_half[c2 + (k + 1) * lev] = ph_n;
    const double l_k = (top0 && k == 0) ? 0.0 : log(ph_k), l_n = log(ph_n);
    double lf;
    if (top0 && k == 0) lf = l_n - 1.0;
    else lf = l_n - (1.0 - ph_k * (l_n - l_k) / (ph_n - ph_k));
    p_full[c2 + k * lev] = mcm ? 0.5 * (ph_n + ph_k) : exp(lf);
    ph_k = ph_n;
  }
  double gh = surf_geop[c2];
  z_half[c2 + (size_t)L * lev] = gh / GRAV;
  const int ktop = (pk[0] == 0.0) ? 1 : 0;
  for (int k = L - 1; k >= 0; --k) {
    const double ph0 = pk[k] + bk[k] * ps, ph1 = pk[k + 1] + bk[k + 1] * ps;
    const double l0 = (top0 && k == 0) ? 0.0 : log(ph0), l1 = log(ph1);
    double lf;
    if (mcm) lf = log(0.5 * (ph1 + ph0));
    else if (top0 && k == 0) lf = l1 - 1.0;
    else lf = l1 - (1.0 - ph0 * (l1 - l0) / (ph1 - ph0));
    const double tk = t[c2 + k * lev];
    z_full[c2 + k * lev] = (gh + RDGAS * tk * (l1 - lf)) / GRAV;
    if (k >= ktop) gh = gh + RDGAS * tk * (l1 - l0);
    z_half[c2 + k * lev] = (k >= ktop) ? gh / GRAV : 0.0;
  }
}
void launch_pressures_heights(const isca_dyn &h, const double *t, const double *ps, double *p_full, double *p_half,
                              double *z_full, double *z_half, hipStream_t s) {
  hipLaunchKernelGGL(k_pressures_heights, grid1d((size_t)h.g.Jl * h.g.I, 64), dim3(64), 0, s, h.g, h.d.pk, h.d.bk, t, ps, h.d.surf_geop, p_full, p_half, z_full, z_half, h.cfg.vert_difference_option == 1);
}

// pressure_variables (press_and_geopot.F90:152-221, simmons_and_burridge): p_half, ln_p_half, p_full, ln_p_full
__global__ void k_pressure_variables(Geom g, const double *__restrict__ pk, const double *__restrict__ bk,
                                     const double *__restrict__ psg, double *p_half, double *ln_p_half, double *p_full,
                                     double *ln_p_full, bool mcm) {
  const size_t c2 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lev = (size_t)g.Jl * g.I;
  if (c2 >= lev) return;
  const double ps = psg[c2];
  const bool top0 = (pk[0] == 0.0 && bk[0] == 0.0);
  double ph_k = pk[0] + bk[0] * ps, l_k = top0 ? 0.0 : log(ph_k);
  p_half[c2] = ph_k; ln_p_half[c2] = l_k;
  for (int k = 0; k < g.L; ++k) {
    const double ph_n = pk[k + 1] + bk[k + 1] * ps, l_n = log(ph_n);
    p_half[c2 + (k + 1) * lev] = ph_n; ln_p_half[c2 + (k + 1) * lev] = l_n;
    if (mcm) { const double pm = 0.5 * (ph_n + ph_k); p_full[c2 + k * lev] = pm; ln_p_full[c2 + k * lev] = log(pm); }       // press_and_geopot.F90:196-200
    else {
    const double lf = (top0 && k == 0) ? l_n - 1.0 : l_n - (1.0 - ph_k * (l_n - l_k) / (ph_n - ph_k));
    ln_p_full[c2 + k * lev] = lf; p_full[c2 + k * lev] = exp(lf);
    }
    ph_k = ph_n; l_k = l_n;
  }
}
// compute_geopotential (press_and_geopot.F90:327-359), dry; the handle's surface geopotential
__global__ void k_geopotential(Geom g, const double *__restrict__ pk, const double *__restrict__ surf_geop, const double *__restrict__ t,
                               const double *__restrict__ ln_p_half, const double *__restrict__ ln_p_full,
                               double *geopot_full, double *geopot_half) {
  const size_t c2 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lev = (size_t)g.Jl * g.I;
  if (c2 >= lev) return;
  const int L = g.L, ktop = (pk[0] == 0.0) ? 1 : 0;
  double gh = surf_geop[c2];
  geopot_half[c2 + (size_t)L * lev] = gh;                     // geopot_half(:,:,num_levels+1) = surf_geopotential (:331)
  if (ktop == 1) geopot_half[c2] = 0.0;
  for (int k = L - 1; k >= 0; --k) {
    const double tk = t[c2 + k * lev], l1 = ln_p_half[c2 + (k + 1) * lev];
    geopot_full[c2 + k * lev] = gh + RDGAS * tk * (l1 - ln_p_full[c2 + k * lev]);
    if (k >= ktop) { gh = gh + RDGAS * tk * (l1 - ln_p_half[c2 + k * lev]); geopot_half[c2 + k * lev] = gh; }
  }
}
void launch_pressure_variables(const isca_dyn &h, const double *ps, double *p_half, double *ln_p_half, double *p_full, double *ln_p_full, hipStream_t s) {
  hipLaunchKernelGGL(k_pressure_variables, grid1d((size_t)h.g.Jl * h.g.I, 64), dim3(64), 0, s, h.g, h.d.pk, h.d.bk, ps, p_half, ln_p_half, p_full, ln_p_full, h.cfg.vert_difference_option == 1);
}
void launch_geopotential(const isca_dyn &h, const double *t, const double *ln_p_half, const double *ln_p_full, double *gf, double *gh, hipStream_t s, const double *surf_geop) {
  hipLaunchKernelGGL(k_geopotential, grid1d((size_t)h.g.Jl * h.g.I, 64), dim3(64), 0, s, h.g, h.d.pk, surf_geop ? surf_geop : h.d.surf_geop, t, ln_p_half, ln_p_full, gf, gh);
}
// mass_weighted_global_integral (global_integral.F90:49-81): per-latitude sums of wts * sum_k field*dp, one block per row
__global__ void k_mass_weighted_rows(Geom g, const double *__restrict__ dpk, const double *__restrict__ dbk,
                                     const double *__restrict__ wts, const double *__restrict__ f,
                                     const double *__restrict__ psg, double *__restrict__ rows) {
  __shared__ double sh[1024];
  const int jl = blockIdx.x;
  const size_t lev = (size_t)g.Jl * g.I;
  double acc = 0.0;
  for (int i = threadIdx.x; i < g.I; i += blockDim.x) {
    const size_t c2 = (size_t)jl * g.I + i;
    const double ps = psg[c2];
    double col = 0.0;
    for (int k = 0; k < g.L; ++k) col += f[c2 + k * lev] * (dpk[k] + dbk[k] * ps);
    acc += col;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = blockDim.x >> 1; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) rows[jl] = wts[jl] * sh[0];
}
void launch_mass_weighted_rows(const isca_dyn &h, const double *f, const double *ps, double *rows, hipStream_t s) {
  hipLaunchKernelGGL(k_mass_weighted_rows, dim3(h.g.Jl), dim3(256), 0, s, h.g, h.d.dpk, h.d.dbk, h.d.wts_lat_l, f, ps, rows);
}

__global__ void k_hadv_combine(size_t n, const double *u, const double *v, const double *dx, const double *dy, double *tend) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tend[i] = tend[i] - u[i] * dx[i] - v[i] * dy[i];
}
void launch_hadv_combine(const Geom &g, const double *u, const double *v, const double *dx, const double *dy, double *tend, int nlev, hipStream_t s) {
  const size_t n = (size_t)g.Jl * g.I * nlev;
  hipLaunchKernelGGL(k_hadv_combine, grid1d(n), dim3(256), 0, s, n, u, v, dx, dy, tend);
}
__global__ void k_scale_rows(Geom g, const double *cosm, double *a, int nlev) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lev = (size_t)g.Jl * g.I;
  if (i < lev * nlev) a[i] *= cosm[(i % lev) / g.I];
}
void launch_scale_rows(const Geom &g, const Dev &d, double *a, int nlev, hipStream_t s) {
  hipLaunchKernelGGL(k_scale_rows, grid1d((size_t)g.Jl * g.I * nlev), dim3(256), 0, s, g, d.cosm_lat_l, a, nlev);
}

// =====================================================================================================
// Grid tracer (sphum): update_tracers 'grid' branch (spectral_dynamics.F90:1155-1180)
//   q0 = tr(prev) + dt * tracer_source_sink          (hs_forcing.F90:240-263, 683-724; sink on atmosphere_mod's copy)
//   horizontal: a_grid_horiz_advection (fv_advection.F90:126-207): van Leer on the sphere with the
//               semi-Lagrangian half-step predictors, polar mirror rows and integer Courant shifts in x
//   vertical:   vert_advection PPM / advective form (vert_advection.F90:301-438, 467-470)
// =====================================================================================================
struct TracerArgs {
  const double *ua, *va, *trp, *tratm_p, *ps_cur, *ps_prev, *wg;
  double *trh, *tr_fut, *tr_cur;
  const double *c, *cc, *dy, *dyy, *dyp, *dym, *dpk, *dbk, *wts;
  const double *rcdx, *rdyy, *rcdy, *rdy;   // 1/(dx c(j)), 1/dyy, 1/(c(j) dy(j)), 1/dy  (same index conventions)
  const double *ppm;                         // [6][L] pure-sigma PPM weights: slope A,B ; edge z1,z2,z3 ; (unused)
  const double *halo_lo, *halo_hi;           // [3 (q0,u,v)][L][2][I] rows j0-2,j0-1 / j0+Jl,j0+Jl+1 received from the neighbour bands
  double *send_lo, *send_hi;                 // same layout: my rows 0,1 / Jl-2,Jl-1
  const int *kmask, *kmask_old;              // the column kernel's word of this step / of the step before (horizontal kernel, halo rows)
  double *wcol;
  double dx, dt, flux, rdamp, robert;
  double *tr_part;
  // Lazy fixers (see mul_nc above).  The previous level's tracer is trp + rb * U(cur): the `future` term of its Robert filter
  // (leapfrog_2level_B) is still to be added, U(cur) = tr_b with the current level's pending water factor; rb = 0 (and tr_b = trp)
  // when nothing is pending.  tratm_p and tr_cur_rd carry the pending water factors of pend_a / pend_c; ps_cur the mass factor of pend_c.
  const double *tr_b, *tr_cur_rd;
  const double *pend_a, *pend_c;
  double rb;
  size_t halo_q;                             // offset of this tracer's q0 rows in the halo buffers (0: tracer 1; 3 + e field blocks: tracer e + 2)
  // filt_horiz: the horizontal kernel -- which holds the previous and the current level of its own rows anyway -- does what needs nothing of the
  // transport: the first half of the Robert filter (tr_cur, tr_part) and the water fixer's sum over q0 ("before": one weighted sum per block into
  // w0blk[level][row block]); the vertical kernel then reads two fields instead of five and writes one instead of two.  Off on the very first step
  // (previous and current level are the same storage there: a block would filter rows its neighbours still read) and with ISCA_TRACER_FILTER_IN_VERT=1.
  int filt_horiz;
  double *w0blk;
};

// tracer_source_sink (hs_forcing.F90:683-724): surface flux into the lowest level, linear sink
__device__ __forceinline__ double tr_source_sink(const TracerArgs &a, const Geom &g, int k, size_t c2, double tr_atm) {
  const double src = (k == g.L - 1) ? a.flux / (a.dpk[k] + a.dbk[k] * a.ps_cur[c2]) : 0.0;
  return src - a.rdamp * tr_atm;
}
// previous-level tracer and atmosphere_mod's copy with what is pending on them applied (kw: the column's water-mask word)
__device__ __forceinline__ double tr_prev_of(const TracerArgs &a, int k, int kw, double trp, double trb) {
  return fma(a.rb, water_corr(trb, k, kmask_byte(kw, 1), a.pend_c[PEND_WFAC]), trp);
}
__device__ __forceinline__ double tr_atm_of(const TracerArgs &a, int k, int kw, double tratm) {
  return water_corr(tratm, k, kmask_byte(kw, 2), a.pend_a[PEND_WFAC]);
}
// q0 from values already in registers (ps is only used at the lowest level)
__device__ __forceinline__ double tr_q0_of(const TracerArgs &a, const Geom &g, int k, double tr_prev, double tr_atm, double ps) {
  const double src = (k == g.L - 1) ? a.flux / (a.dpk[k] + a.dbk[k] * ps) : 0.0;
  return tr_prev + a.dt * (src - a.rdamp * tr_atm);
}
__device__ __forceinline__ double vl_limit(double slope, double qm, double q0, double qp) {
  const double q_min = fmin(fmin(qm, q0), qp), q_max = fmax(fmax(qm, q0), qp);
  return copysign(1.0, slope) * fmin(fmin(fabs(slope), 2.0 * (q0 - q_min)), 2.0 * (q_max - q0));
}

// Values of the lanes to the left / right (longitude i -+ 1) without LDS storage: a DPP wavefront shift, and for the first / last lane
// of a wavefront the neighbouring wavefront's edge value through a small LDS array (periodic in longitude: the block is one row).
__device__ __forceinline__ double dpp_from_left(double v) {    // lane l receives lane l-1 (lane 0: undefined)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);      // wave_shr:1
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_right(double v) {   // lane l receives lane l+1 (lane 63: undefined)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, false);      // wave_shl:1
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// one block = RB consecutive latitude rows of one level, one thread per longitude (lon_max = 2^p; single rank:
// all rows are local).  RB+4 source rows are staged once (q0, u, q + semi_x(q)), RB+2 rows of v.
// LDS holds only what is gathered at a data-dependent longitude (q0 of the source rows for semi_x, q2 and its slope for the van Leer
// x flux): 10 rows of lon_max doubles, 40 KB at lon_max = 512, so that a block fits beside the persistent FFT blocks of the main stream
// (with u and the flux row also in LDS it was 78 KB and found no room while they ran: 554 us next to the main stream at T170L60
// against 152 us alone).
constexpr int TR_RB = 4;
constexpr int TR_LDS_ROWS = TR_RB + 4 + 2;
// WPE: wavefronts per SIMD the register allocation must leave room for.  At lon_max = 512 a block is 8 wavefronts, two per SIMD: with
// 135 registers it finds no SIMD with room beside the main stream's Legendre blocks (2 x 128 registers per SIMD); held to 128 (WPE = 4, 7
// spilled dwords) the kernel runs beside them -- T170L60 step 1.155 -> 1.113 ms on the same box; at lon_max <= 256 (1 wavefront per
// SIMD and block) the unconstrained allocation is the faster one (0.277 vs 0.281 ms at T85L40).  WPE = 5 (96 registers) spills 39 dwords: slower.
// P2: lon_max is a power of two (longitudes wrap with a mask); otherwise (lon_max = 2^a 3^b 5^c, the mixed-radix FFT's lengths) with a remainder.
template <bool P2> __device__ __forceinline__ int wrap_lon(int x, int I) {
  if (P2) return x & (I - 1);
  const int w = x % I;
  return w < 0 ? w + I : w;
}
// The kernel stores early (the filtered current level, TracerArgs.filt_horiz) and reads its per-row tables after that: through the constant address
// space, so that they stay scalar loads (an ordinary load behind a store that may alias goes down the vector path: 133 -> 37 s_load without this).
struct TracerTabs {
  kdouble *cc, *dyp, *dym, *rcdx, *rdyy, *rcdy, *rdy;
  __device__ explicit TracerTabs(const TracerArgs &a)
      : cc((kdouble *)a.cc), dyp((kdouble *)a.dyp), dym((kdouble *)a.dym), rcdx((kdouble *)a.rcdx), rdyy((kdouble *)a.rdyy), rcdy((kdouble *)a.rcdy),
        rdy((kdouble *)a.rdy) {}
};
// LOCAL: one rank -- every source row is in my band (no halo buffers: one address form per row)
template <int WPE, bool P2 = true, bool LOCAL = false>
__global__ __launch_bounds__(512, WPE) void k_tracer_horiz(Geom g, TracerArgs a) {
  const TracerTabs T(a);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RB = TR_RB, NR = RB + 4;
  const int I = g.I, J = g.J;
  double *qs = (double *)smem;        // [NR][I] q0 of virtual rows j0-2..j0+RB+1 (rows across a pole already rotated by I/2)
  double *q2 = qs + NR * I;           // [I]
  double *sx = q2 + I;                // [I]
  __shared__ int any_big[RB];
  __shared__ double edge_first[2 * RB][8], edge_last[RB][8];      // lane 0 / lane 63 values of every wavefront: u of the RB rows; the x flux
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = (blockDim.x + 63) >> 6;
  const int last = min(63, I - 1 - 64 * (int)(threadIdx.x >> 6));   // the wavefront's last lane (a row, or its tail, shorter than a wavefront: lon_max = 32, 96, 160)
  const int wl = (wv + nwv - 1) % nwv, wr = (wv + 1) % nwv;        // wavefronts holding longitudes i - 1 of my lane 0 / i + 1 of my last lane
  // Workgroups go round-robin over the 8 XCDs; neighbouring row blocks share 4 of their 8 source rows, so the tile index is
  // permuted to give each XCD (= each L2) a contiguous run of row blocks (whole levels) instead of every eighth one.
  int tbx = blockIdx.x, tby = blockIdx.y;
  {
    const int total = gridDim.x * gridDim.y;
    if ((total & 7) == 0) {
      const int lin = blockIdx.x + gridDim.x * blockIdx.y, tile = (lin & 7) * (total >> 3) + (lin >> 3);
      tby = tile / (int)gridDim.x; tbx = tile - tby * gridDim.x;
    }
  }
  const int i = threadIdx.x, j0 = g.j0 + tbx * RB, k = tby;      // j0: global index of the block's first row
  const size_t lev = (size_t)g.Jl * I;
  int jsrc[NR];                                                                 // global source row of each virtual row
  bool mir[NR], loc[NR];
  // Every global load of the block first (branch-free: a row outside my band comes from the halo buffer, which
  // already holds q0), then the LDS writes.  A virtual row across a pole is the mirror row seen half-way round the
  // globe (fv_advection.F90:150-164): it is read rotated by I/2, so every later access is at the thread's own i and
  // only q0 and u (gathered at other longitudes) have to live in LDS.
  double tq[NR], ta[NR], tu[NR], tv[NR], tb[NR];
  int kw[NR];
  const size_t fs = (size_t)g.L * 2 * I;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int jv = j0 + r - 2;
    mir[r] = jv < 0 || jv >= J;
    jsrc[r] = jv < 0 ? -jv - 1 : (jv >= J ? 2 * J - 1 - jv : jv);
    const int is = mir[r] ? wrap_lon<P2>(i + (I >> 1), I) : i;
    const int jl = jsrc[r] - g.j0;                                              // local row, or in a neighbour's band
    loc[r] = LOCAL || (jl >= 0 && jl < g.Jl);
    const size_t c2r = (size_t)(loc[r] ? jl : 0) * I + is;
    const size_t q = (size_t)k * lev + c2r;
    const size_t o = ((size_t)k * 2 + (loc[r] ? 0 : (jl < 0 ? jl + 2 : jl - g.Jl))) * I + is;
    const double *hb = (jl < 0) ? a.halo_lo : a.halo_hi;
    const double *pq = loc[r] ? a.trp + q : hb + o + a.halo_q, *pu = loc[r] ? a.ua + q : hb + o + fs, *pv = loc[r] ? a.va + q : hb + o + 2 * fs;
    tq[r] = *pq; ta[r] = *(loc[r] ? a.tratm_p + q : pq); tu[r] = *pu; tv[r] = *pv;
    tb[r] = *(loc[r] ? a.tr_b + q : pq); kw[r] = a.kmask_old[c2r] << 8;                // what is pending on the previous level (halo rows arrive finished)
  }
  // my own rows (always local, never across a pole): the current level where it is not tr_b, the previous surface pressure
  const bool filt = a.filt_horiz != 0, cur_is_b = a.tr_cur_rd == a.tr_b;
  double tc[RB], pp[RB];
#pragma unroll
  for (int rr = 0; rr < RB; ++rr) {
    const int jl = min(j0 + rr - g.j0, g.Jl - 1);
    const size_t c2 = (size_t)jl * I + i;
    tc[rr] = (filt && !cur_is_b) ? a.tr_cur_rd[(size_t)k * lev + c2] : 0.0;
    pp[rr] = filt ? a.ps_prev[c2] : 0.0;
  }
  const double wfac_c = a.pend_c[PEND_WFAC], dpk_k = a.dpk[k], dbk_k = a.dbk[k];
  double wts_r[RB];
#pragma unroll
  for (int rr = 0; rr < RB; ++rr) wts_r[rr] = a.wts[min(j0 + rr - g.j0, g.Jl - 1)];
  double psr[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {                                                // surface flux only enters the lowest level
    const int is = mir[r] ? wrap_lon<P2>(i + (I >> 1), I) : i;
    psr[r] = (k == g.L - 1 && loc[r]) ? mul_nc(a.ps_cur[(size_t)(jsrc[r] - g.j0) * I + is], a.pend_c[PEND_FACTOR]) : 1.0;
  }
  double q0r[NR], vr[NR];                                                       // own-longitude values stay in registers
  double w0 = 0.0, newc[RB], part[RB];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const double tbc = water_corr(tb[r], k, kmask_byte(kw[r], 1), wfac_c), tpv = fma(a.rb, tbc, tq[r]);      // (= tr_prev_of)
    q0r[r] = loc[r] ? tr_q0_of(a, g, k, tpv, tr_atm_of(a, k, kw[r], ta[r]), psr[r]) : tq[r];
    vr[r] = mir[r] ? -tv[r] : tv[r];
    qs[r * I + i] = q0r[r];
    if (r >= 2 && r < 2 + RB) {
      // leapfrog part A on the grid tracer (spectral_dynamics.F90:1164-1167; robert includes raw_filter_coeff): the current level with what is pending on it
      const double tcv = cur_is_b ? tbc : water_corr(tc[r - 2], k, kmask_byte(kw[r], 1), wfac_c);
      part[r - 2] = tpv - 2.0 * tcv;                                            // part_filt_tr (:1164), for the future half of the RAW filter
      newc[r - 2] = tcv + a.robert * part[r - 2];
      if (j0 + r - 2 < g.j0 + g.Jl) w0 += wts_r[r - 2] * (q0r[r] * (dpk_k + dbk_k * pp[r - 2]));     // water before (initialize_corrections :1332-1333), area weight applied here
    }
  }
  if (filt) {
    q2[i] = w0;
#pragma unroll
    for (int rr = 0; rr < RB; ++rr)
      if (j0 + rr < g.j0 + g.Jl) {
        const size_t q = (size_t)k * lev + (size_t)(j0 + rr - g.j0) * I + i;
        a.tr_cur[q] = newc[rr];
        if (a.tr_part) a.tr_part[q] = part[rr];
      }
  }
#pragma unroll
  for (int rr = 0; rr < RB; ++rr) {
    if (lane == 0) edge_first[rr][wv] = tu[rr + 2];
    if (lane == last) edge_last[rr][wv] = tu[rr + 2];
  }
  if (i < RB) any_big[i] = 0;
  __syncthreads();
  if (a.filt_horiz && wv == 0) {                                                // (q2 is next written behind the second barrier)
    double t0 = 0.0;
    for (int x = lane; x < I; x += 64) t0 += q2[x];
    const int nact = min(I, 64);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double o = __shfl_down(t0, off, 64);
      if (lane + off < nact) t0 += o;
    }
    if (lane == 0) a.w0blk[(size_t)k * gridDim.x + tbx] = t0;
  }
  const double hdt = 0.5 * a.dt;
  double q1r[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {      // semi_x (:376-411) on each source row
    const double b = tu[r] * hdt * T.rcdx[jsrc[r]];
    const double fb = floor(b);
    const int il = wrap_lon<P2>(i - 1 - (int)fb, I), ir = wrap_lon<P2>(il + 1, I);
    const double bb = b - fb, qc = q0r[r];
    q1r[r] = qc + (bb * qs[r * I + il] + (1.0 - bb) * qs[r * I + ir] - qc);
  }
  const int im = wrap_lon<P2>(i - 1, I), ip = wrap_lon<P2>(i + 1, I);
  // u at the cell faces and the Courant number of the x flux: formed here for the block-wide flag of the integer part, and again in the row's own
  // iteration below (two lane shifts and three products) rather than kept in twelve registers across the whole loop
  auto faces = [&](int rr, double &uc_l, double &uc_r, double &b) {
    const double u_c = tu[rr + 2];
    double u_l = dpp_from_left(u_c), u_r = dpp_from_right(u_c);
    if (lane == 0) u_l = edge_last[rr][wl];
    if (lane == last) u_r = edge_first[rr][wr];
    uc_l = 0.5 * (u_l + u_c);
    uc_r = 0.5 * (u_c + u_r);
    b = uc_l * a.dt * T.rcdx[j0 + rr];
  };
#pragma unroll
  for (int rr = 0; rr < RB; ++rr) {
    double uc_l, uc_r, b;
    faces(rr, uc_l, uc_r, b);
    if (fabs(b) > 1.0) any_big[rr] = 1;
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < RB; ++rr) {      // (the row's result is stored inside the loop: the per-row tables are read through the constant address space)
    const int r = rr + 2, jg = j0 + rr;
    if (jg >= g.j0 + g.Jl) continue;
    const double q0c = q0r[r], q0m = q0r[r - 1], q0p = q0r[r + 1];
    const double va_c = vr[r];
    // semi_y (:415-433)
    q2[i] = q0c + ((va_c >= 0.0) ? va_c * hdt * (q0m - q0c) * T.rdyy[jg] : va_c * hdt * (q0c - q0p) * T.rdyy[jg + 1]);
    const double vc_lo = 0.5 * (vr[r - 1] + va_c), vc_hi = 0.5 * (va_c + vr[r + 1]);
    double uc_i, uc_p, b;
    faces(rr, uc_i, uc_p, b);
    const double rcdy = T.rcdy[jg];
    double dq = q0c * ((vc_hi * T.cc[jg + 1] - vc_lo * T.cc[jg]) * rcdy + (uc_p - uc_i) * T.rcdx[jg]);
    __syncthreads();
    // vanleer_x (:308-347): slope_x, integer part of the Courant number, fractional van Leer flux
    sx[i] = vl_limit(((q2[ip] - q2[i]) + (q2[i] - q2[im])) / 2, q2[im], q2[i], q2[ip]);
    double fxi = 0.0;
    if (any_big[rr]) {               // integer_flux_x (:494-527), modular form
      const int n_ = (int)fmin(fmax(b, -(double)I), (double)I);      // (a Courant number beyond one turn of the circle is a blown-up state: bounded walk)
      if (n_ >= 1) { for (int t = 1; t <= n_; ++t) fxi += q2[wrap_lon<P2>(i - t, I)]; }
      else if (n_ <= -1) { for (int t = 0; t < -n_; ++t) fxi -= q2[wrap_lon<P2>(i + t, I)]; }
    }
    __syncthreads();
    {
      const double bb = b - trunc(b);
      const int ii = wrap_lon<P2>(i - 1 - (int)floor(b), I);
      const double fl_i = fxi + bb * (q2[ii] + 0.5 * sx[ii] * (copysign(1.0, bb) - bb));
      double fl_r = dpp_from_right(fl_i);                 // the flux through the face at i + 1
      if (lane == 0) edge_first[RB + rr][wv] = fl_i;
      __syncthreads();
      if (lane == last) fl_r = edge_first[RB + rr][wr];
      dq = dq - (fl_r - fl_i) * (1.0 / a.dt);
    }
    {  // vanleer_sphere (:268-304) on q1 with slope_sphere (:546-565)
      double sl[3];
#pragma unroll
      for (int t = 1; t <= 3; ++t) {
        const int jf = jg + t - 1;     // Fortran row index j' = 0..J+1 of the virtual row
        sl[t - 1] = vl_limit((q1r[rr + t + 1] - q1r[rr + t]) * T.dyp[jf] + (q1r[rr + t] - q1r[rr + t - 1]) * T.dym[jf],
                             q1r[rr + t - 1], q1r[rr + t], q1r[rr + t + 1]);
      }
      double f_lo = (vc_lo >= 0.0) ? vc_lo * T.cc[jg] * (q1r[rr + 1] + 0.5 * sl[0] * (1.0 - a.dt * T.rdy[jg + 1] * vc_lo))
                                   : vc_lo * T.cc[jg] * (q1r[rr + 2] - 0.5 * sl[1] * (1.0 + a.dt * T.rdy[jg + 2] * vc_lo));
      double f_hi = (vc_hi >= 0.0) ? vc_hi * T.cc[jg + 1] * (q1r[rr + 2] + 0.5 * sl[1] * (1.0 - a.dt * T.rdy[jg + 2] * vc_hi))
                                   : vc_hi * T.cc[jg + 1] * (q1r[rr + 3] - 0.5 * sl[2] * (1.0 + a.dt * T.rdy[jg + 3] * vc_hi));
      if (jg == 0) f_lo = 0.0;
      if (jg == J - 1) f_hi = 0.0;
      dq = dq - rcdy * (f_hi - f_lo);
    }
    a.trh[(size_t)k * lev + (size_t)(jg - g.j0) * I + i] = q0c + a.dt * dq;
  }
}

// rows 0,1 and Jl-2,Jl-1 of (q0, u, v) for the neighbouring latitude bands (mpp_update_domains, fv_advection.F90:161-162,259)
__global__ void k_tracer_pack_halo(Geom g, TracerArgs a) {
  const int i = threadIdx.x, k = blockIdx.x, hr = blockIdx.y & 1, side = blockIdx.y >> 1;
  const int jl = side ? g.Jl - 2 + hr : hr;
  const size_t lev = (size_t)g.Jl * g.I, c2 = (size_t)jl * g.I + i, q = (size_t)k * lev + c2;
  double *dst = side ? a.send_hi : a.send_lo;
  const size_t o = ((size_t)k * 2 + hr) * g.I + i, fs = (size_t)g.L * 2 * g.I;
  const int kw = a.kmask_old[c2] << 8;
  const double ps = mul_nc(a.ps_cur[c2], a.pend_c[PEND_FACTOR]);
  dst[o + a.halo_q] = tr_q0_of(a, g, k, tr_prev_of(a, k, kw, a.trp[q], a.tr_b[q]), tr_atm_of(a, k, kw, a.tratm_p[q]), ps);
  if (a.halo_q == 0) { dst[o + fs] = a.ua[q]; dst[o + 2 * fs] = a.va[q]; }      // the winds once, with tracer 1
}

// PPM reconstruction of one cell from the column values around it (slope_z :505-568 with limiters, non-linear
// weights; edge values :304-336; Colella-Woodward limiter :354-370).  rv[0..6] = r(k-3..k+3), dv likewise.
__device__ __forceinline__ double ppm_slope(const double *rv, const double *dv, int k, int L) {   // rv/dv centred at index 0 -> cell k
  if (k < 1 || k > L - 2) return 0.0;
  const double gk = (rv[0] - rv[-1]) / (dv[0] + dv[-1]), gp = (rv[1] - rv[0]) / (dv[1] + dv[0]);
  double s_ = (gp * (2. * dv[-1] + dv[0]) + gk * (2. * dv[1] + dv[0])) * dv[0] / (dv[-1] + dv[0] + dv[1]);
  const double rmin = fmin(fmin(rv[-1], rv[0]), rv[1]), rmax = fmax(fmax(rv[-1], rv[0]), rv[1]);
  return copysign(1.0, s_) * fmin(fmin(fabs(s_), 2. * (rv[0] - rmin)), 2. * (rmax - rv[0]));
}
__device__ __forceinline__ double ppm_edge(const double *rv, const double *dv, double slp_k, double slp_km1) {
  // interface between cells k-1 and k (valid for 2 <= k <= L-2); rv/dv centred at cell k
  const double d1 = 1.0 / (dv[-1] + dv[0]), d2 = 1.0 / (dv[-2] + dv[-1] + dv[0] + dv[1]);
  const double d3 = 1.0 / (2 * dv[-1] + dv[0]), d4 = 1.0 / (dv[-1] + 2 * dv[0]);
  const double n3 = dv[-2] + dv[-1], n4 = dv[0] + dv[1];
  const double x = n3 * d3 - n4 * d4, y = 2.0 * dv[-1] * dv[0];
  const double z0 = dv[-1] * d1, z1 = z0 + x * y * d1 * d2, z2 = dv[-1] * n3 * d3 * d2, z3 = dv[0] * n4 * d4 * d2;
  return rv[-1] + z1 * (rv[0] - rv[-1]) - z2 * slp_k + z3 * slp_km1;
}
// limited (r_left, r_right) of cell k; rv/dv centred at cell k, valid offsets -3..+3 where inside the column
__device__ __forceinline__ void ppm_cell(const double *rv, const double *dv, int k, int L, double &left, double &right) {
  const double sk = ppm_slope(rv, dv, k, L);
  const double rk = rv[0];
  // left edge = interface (k-1,k), right edge = interface (k,k+1)
  if (k >= 2 && k <= L - 2) left = ppm_edge(rv, dv, sk, ppm_slope(rv - 1, dv - 1, k - 1, L));
  else left = rk - 0.5 * sk;                       // k = 0, 1, L-1 : linear (slope is 0 at the two ends)
  if (k + 1 >= 2 && k + 1 <= L - 2) right = ppm_edge(rv + 1, dv + 1, ppm_slope(rv + 1, dv + 1, k + 1, L), sk);
  else right = rk + 0.5 * sk;                      // k = 0, L-2, L-1
  if ((right - rk) * (rk - left) <= 0.0) { left = rk; right = rk; }
  if (k != 0 && k != L - 1) {
    const double rm = right - left;
    const double aa = rm * (rk - 0.5 * (right + left)), bb = rm * rm / 6.;
    if (aa > bb) left = 3.0 * rk - 2.0 * right;
    if (aa < -bb) right = 3.0 * rk - 2.0 * left;
  }
}
// same for an arbitrary cell, reading the column from global memory (only for vertical Courant numbers > 1)
__device__ void ppm_cell_global(const TracerArgs &a, const Geom &g, size_t c2, double ps, int kk, double &rc, double &left, double &right) {
  const size_t lev = (size_t)g.Jl * g.I;
  double rv[7], dv[7];
#pragma unroll
  for (int t = -3; t <= 3; ++t) {
    const int kc = min(max(kk + t, 0), g.L - 1);
    rv[t + 3] = a.trh[(size_t)kc * lev + c2];
    dv[t + 3] = a.dpk[kc] + a.dbk[kc] * ps;
  }
  rc = rv[3];
  ppm_cell(rv + 3, dv + 3, kk, g.L, left, right);
}

// Block = 64 columns x NW wavefronts, wavefront w owns levels [w*CH, w*CH+CH) like the column kernel; every thread
// reconstructs its CH+2 cells from CH+8 column values held in registers (no LDS, no barriers in the main part).
// Pure sigma coordinates (pk = 0: HYB = false): dz = dbk*ps, so the slope and edge weights of slope_z / compute_weights are independent
// of the column and come from the host table a.ppm.  Hybrid levels (pk /= 0: HYB = true): the weights are formed per column from the
// layer thicknesses the thread holds anyway (ppm_slope / ppm_edge, the helpers of the Courant > 1 path).
// FILT: the filter's first half and the "water before" sum are done here (a.filt_horiz off); otherwise the horizontal kernel has done them
template <int CH, int MAXW, bool HYB, bool FILT>      // MAXW = wavefronts per block: 8, or 12 (chunks of 5 levels) for 41..60 levels
__global__ __launch_bounds__(64 * MAXW) void k_tracer_vert(Geom g, TracerArgs a) {
  __shared__ double red[5][MAXW][64];
  const int L = g.L;
  const int tid = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;   // w in an SGPR: table lookups by level become scalar loads
  const size_t lev = (size_t)g.Jl * g.I;
  const size_t c2 = (size_t)blockIdx.x * 64 + tid;
  const int k0 = w * CH, nk = min(CH, L - k0);
  const double ps = mul_nc(a.ps_cur[c2], a.pend_c[PEND_FACTOR]);
  constexpr int NV = CH + 8;                     // cells k0-4 .. k0+CH+3
  double rv[NV], dv[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const int kc = min(max(k0 - 4 + t, 0), L - 1);
    rv[t] = a.trh[(size_t)kc * lev + c2];
    dv[t] = a.dpk[kc] + a.dbk[kc] * ps;
  }
  double wv[CH + 1];
#pragma unroll
  for (int t = 0; t <= CH; ++t) wv[t] = a.wg[(size_t)min(k0 + t, L) * lev + c2];
  // what the update at the end needs, requested now with everything else
  double tpv[FILT ? CH : 1], tav[FILT ? CH : 1], tcv[FILT ? CH : 1];
  const int kmw = a.kmask[c2], km = kmask_byte(kmw, 0);
  double psp = 0.0;
  if (FILT) {
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const size_t q = (size_t)min(k0 + t, L - 1) * lev + c2;
      tpv[t] = a.trp[q]; tav[t] = a.tratm_p[q]; tcv[t] = a.tr_cur_rd[q];
    }
#pragma unroll
    for (int t = 0; t < CH; ++t) {                 // what is pending on the two older levels (lazy fixers; identities otherwise)
      const int k = min(k0 + t, L - 1);
      tcv[t] = water_corr(tcv[t], k, kmask_byte(kmw, 1), a.pend_c[PEND_WFAC]);
      tpv[t] = fma(a.rb, tcv[t], tpv[t]);
      tav[t] = tr_atm_of(a, k, kmw, tav[t]);
    }
    psp = a.ps_prev[c2];
  }
  // limited slopes of cells k0-2 .. k0+CH+1 (rv index t+2), once each
  double sl[CH + 4];
#pragma unroll
  for (int t = 0; t < CH + 4; ++t) {
    const int kc = k0 - 2 + t;
    double s_ = 0.0;
    if (HYB) s_ = ppm_slope(rv + t + 2, dv + t + 2, kc, L);
    else if (kc >= 1 && kc <= L - 2) {
      const double rm1 = rv[t + 1], r0 = rv[t + 2], rp1 = rv[t + 3];
      s_ = a.ppm[0 * L + kc] * (rp1 - r0) + a.ppm[1 * L + kc] * (r0 - rm1);
      const double rmin = fmin(fmin(rm1, r0), rp1), rmax = fmax(fmax(rm1, r0), rp1);
      s_ = copysign(1.0, s_) * fmin(fmin(fabs(s_), 2. * (r0 - rmin)), 2. * (rmax - r0));
    }
    sl[t] = s_;
  }
  // edge values at interfaces k0-1 .. k0+CH+1 (interface kc lies between cells kc-1 and kc), once each
  double ed[CH + 3];
#pragma unroll
  for (int t = 0; t < CH + 3; ++t) {
    const int kc = k0 - 1 + t;                    // cell kc = rv index t+3, slope index t+1
    double e = 0.0;
    if (kc >= 2 && kc <= L - 2) {
      if (HYB) e = ppm_edge(rv + t + 3, dv + t + 3, sl[t + 1], sl[t]);
      else e = rv[t + 2] + a.ppm[2 * L + kc] * (rv[t + 3] - rv[t + 2]) - a.ppm[3 * L + kc] * sl[t + 1] + a.ppm[4 * L + kc] * sl[t];
    }
    ed[t] = e;
  }
  // limited parabolas of cells k0-1 .. k0+CH
  double rl[CH + 2], rr[CH + 2];
#pragma unroll
  for (int t = 0; t < CH + 2; ++t) {
    const int kc = k0 - 1 + t;
    double left = 0., right = 0.;
    if (kc >= 0 && kc < L) {
      const double rk = rv[t + 3], sk = sl[t + 1];
      left = (kc >= 2 && kc <= L - 2) ? ed[t] : rk - 0.5 * sk;
      right = (kc + 1 >= 2 && kc + 1 <= L - 2) ? ed[t + 1] : rk + 0.5 * sk;
      if ((right - rk) * (rk - left) <= 0.0) { left = rk; right = rk; }
      if (kc != 0 && kc != L - 1) {
        const double rm = right - left;
        const double aa = rm * (rk - 0.5 * (right + left)), bb = rm * rm / 6.;
        if (aa > bb) left = 3.0 * rk - 2.0 * right;
        if (aa < -bb) right = 3.0 * rk - 2.0 * left;
      }
    }
    rl[t] = left; rr[t] = right;
  }
  // fluxes at interfaces k0 .. k0+CH (:373-432)
  double r_last = 0.0;                                        // r(ke) when my chunk holds the lowest level (static indices only:
#pragma unroll                                                // a run-time index would move rv[] to scratch memory)
  for (int t = 0; t < CH; ++t) if (k0 + t == L - 1) r_last = rv[4 + t];
  double fx[CH + 1];
  const double tt = 2. / 3.;
#pragma unroll
  for (int t = 0; t <= CH; ++t) {
    const int k = k0 + t;
    const double wk = wv[t];
    double f = 0.0;
    if (k <= 0) f = wk * rv[4];                               // flux(ks) = w(ks) r(ks)
    else if (k >= L) f = wk * r_last;                         // flux(ke+1) = w(ke+1) r(ke)
    else if (t <= nk) {
      const bool up = wk >= 0.;
      const int kk = up ? k - 1 : k;                          // donor cell; local index kk-(k0-1) = t or t+1
      // donor-cell values picked with selects between two fixed registers (an index computed at run time would
      // make the compiler keep a copy of rv/dv in scratch memory)
      const double dvd = up ? dv[3 + t] : dv[4 + t];
      const double cn = (up ? a.dt * wk : -a.dt * wk) / dvd;
      double rc = up ? rv[3 + t] : rv[4 + t], left = up ? rl[t] : rl[t + 1], right = up ? rr[t] : rr[t + 1], xx = cn, rsum = 0.;
      int kd = kk;
      if (cn > 1.) {                                          // extension for Courant numbers > 1 (:385-396, :410-421)
        double dzsum = 0.;
        const double dtw = up ? a.dt * wk : -a.dt * wk;
        double dzk = dvd;
        while (dzsum + dzk < dtw) {
          if (kd == (up ? 0 : g.L - 1)) break;                // the reference stops at kk == 1 going up; going down (`kk == ks`, :414) it never does:
                                                              // a blown-up state must not walk off the column (memory fault) before valid_range_t reports it
          dzsum += dzk; rsum += a.trh[(size_t)kd * lev + c2];
          kd += up ? -1 : 1;
          dzk = a.dpk[kd] + a.dbk[kd] * ps;
        }
        xx = (dtw - dzsum) / dzk;
        if (kd != kk) ppm_cell_global(a, g, c2, ps, kd, rc, left, right);
      }
      const double rm = right - left;
      double r6 = 6.0 * (rc - 0.5 * (right + left));
      if (up ? (kd == 0) : (kd == L - 1)) r6 = 0.;
      double rst = up ? right - 0.5 * xx * (rm - (1.0 - tt * xx) * r6) : left + 0.5 * xx * (rm + (1.0 - tt * xx) * r6);
      if (cn > 1.) rst = (xx * rst + rsum) / cn;
      f = wk * rst;
    }
    fx[t] = f;
  }
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0., s4 = 0.;
  double newc[CH], newf[CH];
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    newc[t] = newf[t] = 0.0;
    if (t < nk) {
      const int k = k0 + t;
      const double rk = rv[4 + t];
      const double rdt = -(fx[t + 1] - fx[t] - rk * (wv[t + 1] - wv[t])) / dv[4 + t];
      const double trf = rk + a.dt * rdt;
      newf[t] = trf;
      if (FILT) {
        // tr(prev) aliases tr(fut) from the second step on (and tr(cur) on the first): everything was read at the top
        const double q0 = tr_q0_of(a, g, k, tpv[t], tav[t], ps);
        newc[t] = tcv[t] + a.robert * (tpv[t] - 2.0 * tcv[t]);      // leapfrog part A on the grid tracer (:1164-1167); robert includes raw_filter_coeff
        s0 += q0 * (a.dpk[k] + a.dbk[k] * psp);                     // water before (initialize_corrections :1332-1333)
      }
      // column sums of the water after (compute_corrections :1249-1262)
      const double msk = (k >= km) ? 1.0 : 0.0;
      s1 += trf * a.dpk[k]; s2 += trf * a.dbk[k];
      s3 += msk * trf * a.dpk[k]; s4 += msk * trf * a.dbk[k];
    }
  }
#pragma unroll
  for (int t = 0; t < CH; ++t)
    if (t < nk) {
      const size_t q = (size_t)(k0 + t) * lev + c2;
      a.tr_fut[q] = newf[t];
      if (FILT) {
        a.tr_cur[q] = newc[t];
        if (a.tr_part) a.tr_part[q] = tpv[t] - 2.0 * tcv[t];        // part_filt_tr (:1164), for the future half of the RAW filter
      }
    }
  red[0][w][tid] = s0; red[1][w][tid] = s1; red[2][w][tid] = s2; red[3][w][tid] = s3; red[4][w][tid] = s4;
  __syncthreads();
  if (w < 5 && (FILT || w > 0)) {
    double s_ = 0.0;
    for (int ww = 0; ww < NW; ++ww) s_ += red[w][ww][tid];
    a.wcol[(size_t)w * lev + c2] = s_;
  }
}

// A 'grid' tracer whose advect_vert is not finite_volume_parabolic (update_tracers, spectral_dynamics.F90:1161: vert_advection with the entry's
// scheme on tr_future after the horizontal step): one thread per column, the column in private arrays (vadv_column), then the same update,
// filter part and column sums as k_tracer_vert.  An option path: the reference's own field_tables all use finite_volume_parabolic.
template <int SCHEME>
__global__ __launch_bounds__(64) void k_tracer_vert_scheme(Geom g, TracerArgs a) {
  const size_t lev = (size_t)g.Jl * g.I, c2 = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (c2 >= lev) return;
  const int L = g.L;
  double w[VADV_MAXL + 1], dz[VADV_MAXL], r[VADV_MAXL], rdt[VADV_MAXL];
  double tpv[VADV_MAXL], tav[VADV_MAXL], tcv[VADV_MAXL];
  const double ps = mul_nc(a.ps_cur[c2], a.pend_c[PEND_FACTOR]), psp = a.ps_prev[c2];
  const int kmw = a.kmask[c2], km = kmask_byte(kmw, 0);
  for (int k = 0; k <= L; ++k) w[k] = a.wg[c2 + (size_t)k * lev];
  w[0] = 0.0;
  for (int k = 0; k < L; ++k) {         // tr(prev) aliases tr(fut) from the second step on: everything is read before the first store
    const size_t q = c2 + (size_t)k * lev;
    dz[k] = a.dpk[k] + a.dbk[k] * ps;
    r[k] = a.trh[q];
    tcv[k] = water_corr(a.tr_cur_rd[q], k, kmask_byte(kmw, 1), a.pend_c[PEND_WFAC]);
    tpv[k] = fma(a.rb, tcv[k], a.trp[q]);
    tav[k] = tr_atm_of(a, k, kmw, a.tratm_p[q]);
  }
  vadv_column<SCHEME>(L, a.dt, w, dz, r, rdt);
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0., s4 = 0.;
  for (int k = 0; k < L; ++k) {
    const size_t q = c2 + (size_t)k * lev;
    const double trf = r[k] + a.dt * rdt[k];
    a.tr_fut[q] = trf;
    if (!a.filt_horiz) {
      const double q0 = tr_q0_of(a, g, k, tpv[k], tav[k], ps);
      a.tr_cur[q] = tcv[k] + a.robert * (tpv[k] - 2.0 * tcv[k]);
      if (a.tr_part) a.tr_part[q] = tpv[k] - 2.0 * tcv[k];
      s0 += q0 * (a.dpk[k] + a.dbk[k] * psp);
    }
    const double msk = (k >= km) ? 1.0 : 0.0;
    s1 += trf * a.dpk[k]; s2 += trf * a.dbk[k];
    s3 += msk * trf * a.dpk[k]; s4 += msk * trf * a.dbk[k];
  }
  if (!a.filt_horiz) a.wcol[0 * lev + c2] = s0;
  a.wcol[1 * lev + c2] = s1; a.wcol[2 * lev + c2] = s2; a.wcol[3 * lev + c2] = s3; a.wcol[4 * lev + c2] = s4;
}

// robert_coeff of field_table entry k+1 (spectral_dynamics.F90:340-351): its own, or the dynamics' one
// hs_forcing's source and sink for field_table entry k (0-based; hs_forcing.F90:250-265): the entry's tracer_sms values or hs_forcing_nml's
static double tracer_sms_flux(const isca_dyn &h, int k) { return (k >= 0 && h.cfg.tracer_sms[k]) ? h.cfg.tracer_flux[k] : h.cfg.trflux; }
static double tracer_sms_rdamp(const isca_dyn &h, int k) {
  double r = (k >= 0 && h.cfg.tracer_sms[k]) ? h.cfg.tracer_sink[k] : h.cfg.trsink;       // tracer_source_sink, hs_forcing.F90:697-699
  if (r < 0.) r = -86400. * r;
  return r > 0. ? 1. / r : r;
}
// advect_vert of field_table entry k (0-based): the entry's, or the representation's standard one (grid: finite_volume_parabolic, spectral: second_centered)
int tracer_vert_scheme(const isca_dyn &h, int k) {
  const int v = h.cfg.tracer_advect_vert[k];
  return v >= 0 ? v : ((k > 0 && h.cfg.tracer_spectral[k]) ? 0 : 3);
}
static double tracer_robert(const isca_dyn &h, int k) {
  return h.cfg.tracer_robert_coeff[k] >= 0.0 ? h.cfg.tracer_robert_coeff[k] : h.cfg.robert_coeff;
}
static TracerArgs tracer_args(const isca_dyn &h, const StepScalars &sc) {
  const Dev &d = h.d;
  TracerArgs a;
  a.ua = d.ug[sc.cur]; a.va = d.vg[sc.cur]; a.trp = d.tr[sc.prev]; a.tratm_p = d.tr_atm[sc.prev];
  a.ps_cur = d.psg[sc.cur]; a.ps_prev = d.psp_copy; a.wg = d.wg;   // psg(prev) storage is rewritten by the synthesis running concurrently
  a.trh = d.trh; a.tr_fut = d.tr[sc.fut]; a.tr_cur = d.tr[sc.cur];
  a.c = d.fv_c; a.cc = d.fv_cc; a.dy = d.fv_dy; a.dyy = d.fv_dyy; a.dyp = d.fv_dyp; a.dym = d.fv_dym;
  a.dpk = d.dpk; a.dbk = d.dbk; a.wts = d.wts_lat_l; a.kmask = d.kmask; a.kmask_old = d.kmask_old; a.wcol = d.wcol;
  a.rcdx = d.fv_rcdx; a.rdyy = d.fv_rdyy; a.rcdy = d.fv_rcdy; a.rdy = d.fv_rdy; a.ppm = d.ppm_tab;
  a.dx = h.tab.fv_dx; a.dt = sc.delta_t; a.flux = tracer_sms_flux(h, 0);
  a.rdamp = tracer_sms_rdamp(h, 0);
  a.robert = h.cfg.robert_coeff * h.cfg.raw_filter_coeff;
  a.tr_part = d.tr_part;
  // pending fixers (identity rows unless lazy_fix): the mass factor on ps(cur), the water factors on the two older tracer levels
  a.pend_c = d.pend + 4 * sc.cur; a.pend_a = d.pend + 4 * sc.prev;
  a.tr_b = a.trp; a.tr_cur_rd = a.tr_cur; a.rb = 0.0;
  if (h.lazy_fix) {
    a.tr_fut = d.tr_atm[sc.fut];                                            // the new level is kept once, uncorrected, where atmosphere_mod's copy lives
    if (h.tr_state[sc.cur] == TR_NEW) a.tr_cur_rd = d.tr_atm[sc.cur];       // ... so that is where the current level is read from
    if (h.tr_state[sc.prev] == TR_FILT) { a.rb = h.cfg.robert_coeff; a.tr_b = d.tr_atm[sc.cur]; }   // leapfrog_2level_B's `a(current) += robert a(future)` of the last step
  }
  if (h.cfg.physics != 0) {     // sphum / the caller's tracer: the source is the physics tendency, q0 = tr(prev) + dt * dt_tracers (0 - (-1) x = x exactly)
    a.tratm_p = d.ph_dtq; a.flux = 0.0; a.rdamp = -1.0; a.pend_a = d.pend + PEND_IDENTITY;
  }
  a.halo_lo = d.halo_recv; a.halo_hi = d.halo_recv + halo_doubles(h.g, h.cfg.num_tracers);
  a.send_lo = d.halo_send; a.send_hi = d.halo_send + halo_doubles(h.g, h.cfg.num_tracers);
  a.halo_q = 0;
  a.filt_horiz = h.tr_filt_horiz ? 1 : 0; a.w0blk = d.w0blk;
  return a;
}
// tracer e + 2 of the field_table (a further 'grid' tracer): the same transport on its own time levels; the column sums go to a spare array
// (only tracer 1 is water), the filter's `future` term is added at the end of the step (k_tracer_finish), its halo rows have their own block
static TracerArgs further_tracer_args(const isca_dyn &h, const StepScalars &sc, const TracerArgs &a, int e) {
  TracerArgs b = a;
  b.trp = h.d.trx[sc.prev][e]; b.tr_cur = h.d.trx[sc.cur][e]; b.tr_fut = h.d.trx[sc.fut][e]; b.wcol = h.d.wcol_x; b.tr_part = nullptr;
  b.tr_b = b.trp; b.tr_cur_rd = b.tr_cur; b.rb = 0.0; b.pend_a = h.d.pend + PEND_IDENTITY;        // (more than one tracer: the fixers are applied eagerly)
  b.robert = tracer_robert(h, e + 1); b.w0blk = h.d.w0blk_x;
  b.halo_q = (size_t)(3 + e) * h.g.L * 2 * h.g.I;
  if (h.cfg.physics == 0) { b.tratm_p = h.d.trx_atm[sc.prev][e]; b.flux = tracer_sms_flux(h, e + 1); b.rdamp = tracer_sms_rdamp(h, e + 1); }   // hs_forcing's source and sink act on every tracer, each with its tracer_sms (hs_forcing.F90:248-265)
  else if (h.cfg.physics == 2) b.tratm_p = h.d.ph_dtqx[e];                   // the caller's dt_tracers(:,:,:,ntr)
  else { b.tratm_p = b.trp; b.flux = 0.0; b.rdamp = 0.0; }                   // idealized_moist_phys only has a tendency for sphum
  return b;
}
void launch_tracer_pack_halo(const isca_dyn &h, const StepScalars &sc, hipStream_t s) {
  const Geom &g = h.g;
  TracerArgs a = tracer_args(h, sc);
  hipLaunchKernelGGL(k_tracer_pack_halo, dim3(g.L, 4), dim3(g.I), 0, s, g, a);
  for (int e = 0; e + 1 < h.cfg.num_tracers; ++e) {
    if (h.cfg.tracer_spectral[e + 1]) continue;
    hipLaunchKernelGGL(k_tracer_pack_halo, dim3(g.L, 4), dim3(g.I), 0, s, g, further_tracer_args(h, sc, a, e));
  }
}
static void launch_tracer_horiz_kernel(const Geom &g, const TracerArgs &a, size_t ldsh, hipStream_t s) {
  const dim3 grid((g.Jl + TR_RB - 1) / TR_RB, g.L), block(g.I);
  if (g.I & (g.I - 1)) {        // lon_max with factors 3, 5: longitudes wrap with a remainder
    if (g.I > 256) hipLaunchKernelGGL((k_tracer_horiz<4, false>), grid, block, ldsh, s, g, a);
    else hipLaunchKernelGGL((k_tracer_horiz<1, false>), grid, block, ldsh, s, g, a);
  } else if (g.P == 1 && !exp_env("ISCA_TRACER_HORIZ_GENERIC")) {
    if (g.I > 256) hipLaunchKernelGGL((k_tracer_horiz<4, true, true>), grid, block, ldsh, s, g, a);
    else hipLaunchKernelGGL((k_tracer_horiz<1, true, true>), grid, block, ldsh, s, g, a);
  } else if (g.I > 256) hipLaunchKernelGGL(k_tracer_horiz<4>, grid, block, ldsh, s, g, a);
  else hipLaunchKernelGGL(k_tracer_horiz<1>, grid, block, ldsh, s, g, a);
}
static void launch_tracer_vert_kernel(const Geom &g, const TracerArgs &a, hipStream_t s, int scheme = 3) {
  const dim3 grid((unsigned)((size_t)g.Jl * g.I / 64));
  if (scheme != 3) {                              // the entry's advect_vert: 0 second_centered, 1 fourth_centered, 2 van_leer_linear
    if (g.L < 4 || g.L > VADV_MAXL) throw std::runtime_error("advect_vert other than finite_volume_parabolic needs 4..64 levels");
    if (scheme == 0) hipLaunchKernelGGL(k_tracer_vert_scheme<0>, grid, dim3(64), 0, s, g, a);
    else if (scheme == 1) hipLaunchKernelGGL(k_tracer_vert_scheme<1>, grid, dim3(64), 0, s, g, a);
    else hipLaunchKernelGGL(k_tracer_vert_scheme<2>, grid, dim3(64), 0, s, g, a);
    return;
  }
  if (g.L > 40 && g.L <= 60) {                  // 12 wavefronts of 5 levels (164 VGPRs, 3 wavefronts per SIMD) instead of 8 of 8 (207)
    const int NW = (g.L + 4) / 5;
#define LV(HYB) do { if (a.filt_horiz) hipLaunchKernelGGL((k_tracer_vert<5, 12, HYB, false>), grid, dim3(64 * NW), 0, s, g, a); \
                     else hipLaunchKernelGGL((k_tracer_vert<5, 12, HYB, true>), grid, dim3(64 * NW), 0, s, g, a); } while (0)
    if (a.ppm) LV(false); else LV(true);
#undef LV
    return;
  }
  const int CH = std::max(1, (g.L + 7) / 8), NW = (g.L + CH - 1) / CH;    // NW >= 5 needed for the 5 column sums
  const dim3 block(64 * NW);
#define LF(N, HYB) do { if (a.filt_horiz) hipLaunchKernelGGL((k_tracer_vert<N, 8, HYB, false>), grid, block, 0, s, g, a); \
                        else hipLaunchKernelGGL((k_tracer_vert<N, 8, HYB, true>), grid, block, 0, s, g, a); } while (0)
#define LT(N) do { if (a.ppm) LF(N, false); else LF(N, true); } while (0)   // a.ppm: the pure-sigma weight table, null with hybrid levels
  switch (CH) {
    case 1: LT(1); break; case 2: LT(2); break; case 3: LT(3); break; case 4: LT(4); break;
    case 5: LT(5); break; case 6: LT(6); break; case 7: LT(7); break; default: LT(8); break;
  }
#undef LT
#undef LF
}
// part 0: tracer 1's horizontal kernel (needs nothing of this step's column kernel); part 1: everything after it; -1: both
void launch_tracer(const isca_dyn &h, const StepScalars &sc, hipStream_t s, int part) {
  const Geom &g = h.g;
  TracerArgs a = tracer_args(h, sc);
  const size_t ldsh = (size_t)TR_LDS_ROWS * g.I * sizeof(double);
  if (part != 1) launch_tracer_horiz_kernel(g, a, ldsh, s);
  if (part == 0) return;
  launch_tracer_vert_kernel(g, a, s, tracer_vert_scheme(h, 0));
  // further 'grid' tracers of the field_table (update_tracers' loop, spectral_dynamics.F90:1132,1155-1180): the same transport, their own
  // time levels; the column sums go to a spare array (only tracer 1 is water) and the filter's `future` term is added at the end of the step
  for (int e = 0; e + 1 < h.cfg.num_tracers; ++e) {
    if (h.cfg.tracer_spectral[e + 1]) continue;
    const TracerArgs b = further_tracer_args(h, sc, a, e);
    launch_tracer_horiz_kernel(g, b, ldsh, s);
    launch_tracer_vert_kernel(g, b, s, tracer_vert_scheme(h, e + 1));
  }
}

// ---- tracers 2..num_tracers: the end-of-step pieces
// grid tracer: leapfrog_2level_B (spectral_dynamics.F90:1484) a(current) += robert a(future), and atmosphere_mod's copy of the new level
__global__ void k_tracer_finish(size_t n, double robert, const double *__restrict__ fut, double *__restrict__ cur, double *__restrict__ atm_fut) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double f = fut[i]; if (cur) cur[i] += robert * f; atm_fut[i] = f; }
}
void launch_tracer_finish(const isca_dyn &h, const StepScalars &sc, int e, hipStream_t s) {
  const size_t n = (size_t)h.g.L * h.g.Jl * h.g.I;
  // a spectral tracer's grid values at `current` are not filtered (only its coefficients are, :1148 and :1482)
  double *cur = h.cfg.tracer_spectral[e + 1] ? nullptr : h.d.trx[sc.cur][e];
  hipLaunchKernelGGL(k_tracer_finish, grid1d(n), dim3(256), 0, s, n, tracer_robert(h, e + 1), h.d.trx[sc.fut][e], cur, h.d.trx_atm[sc.fut][e]);
}
// vert_advection(dt, w, dz, r, rdt, scheme = SECOND_CENTERED, form = ADVECTIVE_FORM) (vert_advection.F90:158-193, 461-465) with
// dz = p_half(k+1) - p_half(k) = dpk + dbk ps; the tendency is added to rdt (update_tracers, spectral_dynamics.F90:1139-1141)
__global__ void k_vert_advection_centered(Geom g, const double *__restrict__ w, const double *__restrict__ ps, const double *__restrict__ dpk,
                                          const double *__restrict__ dbk, const double *__restrict__ r, double *__restrict__ rdt) {
  const size_t lev = (size_t)g.Jl * g.I, n = lev * g.L;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = (int)(i / lev);
  const size_t c2 = i - (size_t)k * lev;
  const double rk = r[i], w0 = w[i], w1 = w[i + lev];
  const double f0 = (k == 0) ? w0 * rk : w0 * (0.5 * (rk + r[i - lev]));
  const double f1 = (k == g.L - 1) ? w1 * rk : w1 * (0.5 * (r[i + lev] + rk));
  const double dz = dpk[k] + dbk[k] * ps[c2];
  rdt[i] = rdt[i] + -(f1 - f0 - rk * (w1 - w0)) / dz;
}
void launch_vert_advection_centered(const isca_dyn &h, const double *w, const double *ps, const double *r, double *rdt, hipStream_t s) {
  const size_t n = (size_t)h.g.L * h.g.Jl * h.g.I;
  hipLaunchKernelGGL(k_vert_advection_centered, grid1d(n), dim3(256), 0, s, h.g, w, ps, h.d.dpk, h.d.dbk, r, rdt);
}
// water_borrowing (atmos_spectral/model/water_borrowing.F90:38-136; hole_filling = 'on' of a 'spectral' tracer, spectral_dynamics.F90:1142-1144): a
// negative value q(i,k) of the PREVIOUS level is filled from its neighbours on the latitude circle (i-1, i+1, wrapping) and in the column (k-1, k+1)
// when together they hold enough (total = sum_nb q dp + q dp > 0): its tendency gets -q/dt, each neighbour's (ratio - 1) q_nb/dt, ratio = total /
// neighbouring water.  The reference sweeps the circle (alternating direction) and scatters; q is only read, so every hole's contribution is
// independent and the sweep fixes nothing but the order of the additions: here every cell GATHERS -- its own hole and the holes among its four
// neighbours, each evaluated from the hole's own five-point stencil.  dp = dpk + dbk p_s of the current level (update_tracers' p_half).
__device__ __forceinline__ double wb_ratio_m1(const Geom &g, const double *__restrict__ q, const double *__restrict__ dpk, const double *__restrict__ dbk,
                                              const double *__restrict__ psrow, size_t row, size_t lev, int i, int k, bool *fillable = nullptr) {
  // (ratio - 1) of the hole at (i, k) of latitude row `row` (offset of its first longitude in a level), or 0 when it is no hole / cannot be filled;
  // fillable: whether it is a hole that IS filled (negative value, total water of the stencil positive: water_borrowing.F90) -- ratio - 1 rounds to
  // exactly 0 for a hole 1e-16 of the neighbouring water deep, whose own fill term is still applied
  const int I = g.I, im = (i == 0) ? I - 1 : i - 1, ip = (i == I - 1) ? 0 : i + 1;
  const size_t o = (size_t)k * lev + row;
  const double qc = q[o + i];
  if (fillable) *fillable = false;
  if (!(qc < 0.)) return 0.0;
  double nb = q[o + im] * (dpk[k] + dbk[k] * psrow[im]);
  nb = nb + q[o + ip] * (dpk[k] + dbk[k] * psrow[ip]);
  if (k != 0) nb = nb + q[o - lev + i] * (dpk[k - 1] + dbk[k - 1] * psrow[i]);
  if (k != g.L - 1) nb = nb + q[o + lev + i] * (dpk[k + 1] + dbk[k + 1] * psrow[i]);
  const double total = nb + qc * (dpk[k] + dbk[k] * psrow[i]);
  if (fillable) *fillable = total > 0.;
  return (total > 0.) ? total / nb - 1.0 : 0.0;
}
__global__ void k_water_borrowing(Geom g, const double *__restrict__ dpk, const double *__restrict__ dbk, const double *__restrict__ ps,
                                  const double *__restrict__ q, double *__restrict__ dt_q, double delta_t) {
  const size_t lev = (size_t)g.Jl * g.I, n = lev * g.L;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int k = (int)(idx / lev);
  const size_t c2 = idx - (size_t)k * lev;
  const int I = g.I, jl = (int)(c2 / I), i = (int)(c2 - (size_t)jl * I);
  const size_t row = (size_t)jl * I;
  const double *psrow = ps + row;
  const int im = (i == 0) ? I - 1 : i - 1, ip = (i == I - 1) ? 0 : i + 1;
  const double qc = q[idx];
  double t = dt_q[idx];
  // the reference's order of events for one cell does not exist (they interleave with the sweep); this one is fixed: own hole, west, east, above, below
  bool own;
  (void)wb_ratio_m1(g, q, dpk, dbk, psrow, row, lev, i, k, &own);
  if (own) t = t - qc / delta_t;
  t = t + wb_ratio_m1(g, q, dpk, dbk, psrow, row, lev, im, k) * qc / delta_t;
  t = t + wb_ratio_m1(g, q, dpk, dbk, psrow, row, lev, ip, k) * qc / delta_t;
  if (k != 0) t = t + wb_ratio_m1(g, q, dpk, dbk, psrow, row, lev, i, k - 1) * qc / delta_t;
  if (k != g.L - 1) t = t + wb_ratio_m1(g, q, dpk, dbk, psrow, row, lev, i, k + 1) * qc / delta_t;
  dt_q[idx] = t;
}
void launch_water_borrowing(const isca_dyn &h, const double *ps, const double *q_prev, double *dt_q, double delta_t, hipStream_t s) {
  const size_t n = (size_t)h.g.L * h.g.Jl * h.g.I;
  hipLaunchKernelGGL(k_water_borrowing, grid1d(n), dim3(256), 0, s, h.g, h.d.dpk, h.d.dbk, ps, q_prev, dt_q, delta_t);
}
// leapfrog_2level_A / _B (leapfrog.F90:58-105) on caller arrays (the C-ABI entry points behind leapfrog_mod): every value is read before
// anything is written, so `fut` may be the storage of `prev` (two time levels) or of `cur` (the first step)
__global__ void k_leapfrog_a(size_t n, const double *prev, double *cur, double *fut, const double *__restrict__ dta, double delta_t,
                             double robert, double raw, double *part) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double p = prev[i], c = cur[i];
  const double pt = p - 2.0 * c;
  const double f = p + delta_t * dta[i], cn = c + robert * pt * raw;
  if (part) part[i] = pt;
  if (fut == cur) { cur[i] = f; }           // previous == current == future makes no sense; future == current: the new level wins (leapfrog.F90:74-76)
  else { cur[i] = cn; fut[i] = f; }
}
__global__ void k_leapfrog_b(size_t n, double *cur, double *fut, const double *__restrict__ part, double robert, double raw) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double c = cur[i], f = fut[i];
  cur[i] = c + robert * f * raw;
  fut[i] = f + robert * (part[i] + f) * (raw - 1.0);
}
void launch_leapfrog_a(size_t n, const double *prev, double *cur, double *fut, const double *dta, double delta_t, double robert, double raw,
                       double *part, hipStream_t s) {
  hipLaunchKernelGGL(k_leapfrog_a, grid1d(n), dim3(256), 0, s, n, prev, cur, fut, dta, delta_t, robert, raw, part);
}
void launch_leapfrog_b(size_t n, double *cur, double *fut, const double *part, double robert, double raw, hipStream_t s) {
  hipLaunchKernelGGL(k_leapfrog_b, grid1d(n), dim3(256), 0, s, n, cur, fut, part, robert, raw);
}
// spectral tracer: compute_spectral_damping (spectral_damping.F90:172-201, the coefficients temperature uses) on dt_trs, leapfrog_2level_A
// (leapfrog.F90:58-84) and _B's `current` half (:100); one thread per coefficient and level, each reads its three time levels first
// (previous aliases future from the second step on, current on the first)
__global__ void k_spec_tracer_update(Geom g, const double *__restrict__ coef, double delta_t, double robert, const double2 *__restrict__ dt_trs,
                                     const double2 *prev, double2 *cur, double2 *fut) {
  const size_t n = (size_t)g.Ml * g.N1 * g.L;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t mn = i / g.L;
  const double dmp = coef[(size_t)C_DAMP * g.Ml * g.N1 + mn];
  const double2 p = prev[i], c = cur[i];
  double2 dt = dt_trs[i];
  const double cf = 1.0 / (1.0 + dmp * delta_t);
  dt = cscale(cf, csub(dt, cscale(dmp, p)));
  const double2 part = make_double2(p.x - 2.0 * c.x, p.y - 2.0 * c.y);
  const double2 nf = make_double2(p.x + delta_t * dt.x, p.y + delta_t * dt.y);
  double2 nc = make_double2(c.x + robert * part.x, c.y + robert * part.y);
  nc = make_double2(nc.x + robert * nf.x, nc.y + robert * nf.y);
  cur[i] = nc; fut[i] = nf;
}
void launch_spec_tracer_update(const isca_dyn &h, const StepScalars &sc, int e, const double *dt_trs, hipStream_t s) {
  const Geom &g = h.g;
  const size_t n = (size_t)g.Ml * g.N1 * g.L;
  hipLaunchKernelGGL(k_spec_tracer_update, grid1d(n), dim3(256), 0, s, g, h.d.coef, sc.delta_t, tracer_robert(h, e + 1), (const double2 *)dt_trs,
                     (const double2 *)h.d.trxs[sc.prev][e], (double2 *)h.d.trxs[sc.cur][e], (double2 *)h.d.trxs[sc.fut][e]);
}

// The same two kernels on caller fields (C-ABI entry points isca_a_grid_horiz_advection / isca_vert_advection_ppm):
// no source/sink, so the kernels return q + dt * tendency of the pure advection operator.
void launch_fv_horiz_on(const isca_dyn &h, const double *u, const double *v, const double *q, const double *ps, double dt, double *q_new, hipStream_t s) {
  const Geom &g = h.g;
  StepScalars sc{}; sc.delta_t = dt;
  TracerArgs a = tracer_args(h, sc);
  a.ua = u; a.va = v; a.trp = q; a.tratm_p = q; a.trh = q_new; a.ps_cur = ps;   // ps only enters the (zero) surface flux
  a.flux = 0.0; a.rdamp = 0.0; a.dt = dt;
  a.tr_b = q; a.rb = 0.0; a.pend_a = a.pend_c = h.d.pend + PEND_IDENTITY;
  a.filt_horiz = 0;                            // (the transport alone: nothing of the model's time levels is touched)
  const size_t ldsh = (size_t)TR_LDS_ROWS * g.I * sizeof(double);
  launch_tracer_horiz_kernel(g, a, ldsh, s);
}
void launch_ppm_vert_on(const isca_dyn &h, double dt, const double *w, const double *ps, const double *r, double *r_new,
                        double *dummy_a, double *dummy_b, hipStream_t s) {
  const Geom &g = h.g;
  StepScalars sc{}; sc.delta_t = dt;
  TracerArgs a = tracer_args(h, sc);
  a.trh = const_cast<double *>(r); a.wg = w; a.ps_cur = ps; a.ps_prev = ps; a.trp = dummy_a; a.tratm_p = dummy_a;
  a.tr_cur = dummy_b; a.tr_fut = r_new; a.flux = 0.0; a.rdamp = 0.0; a.dt = dt;
  a.tr_b = dummy_a; a.tr_cur_rd = dummy_b; a.rb = 0.0; a.pend_a = a.pend_c = h.d.pend + PEND_IDENTITY;
  a.filt_horiz = 0;
  launch_tracer_vert_kernel(g, a, s);
}
// tracer_source_sink (hs_forcing.F90:683-724) on caller fields: rst += flux/dp at the lowest level - tr/sink
__global__ void k_tracer_source_sink(Geom g, TracerArgs a, const double *__restrict__ tr, double *__restrict__ rdt) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lev = (size_t)g.Jl * g.I;
  if (idx >= lev * g.L) return;
  const int k = idx / lev;
  const size_t c2 = idx % lev;
  rdt[idx] += tr_source_sink(a, g, k, c2, tr[idx]);
}
void launch_tracer_source_sink(const isca_dyn &h, const double *ps, const double *tr, double *rdt, hipStream_t s, int k) {
  StepScalars sc{};
  TracerArgs a = tracer_args(h, sc);
  a.ps_cur = ps;
  a.flux = tracer_sms_flux(h, k); a.rdamp = tracer_sms_rdamp(h, k);   // hs_forcing's own, whatever physics the handle steps with
  hipLaunchKernelGGL(k_tracer_source_sink, grid1d((size_t)h.g.Jl * h.g.I * h.g.L), dim3(256), 0, s, h.g, a, tr, rdt);
}

// =====================================================================================================
// Mass / energy fixers (spectral_dynamics.F90:1213-1302; global_integral.F90:49-81; transforms.F90:1059-1077)
//   red[0..1]  local sums of the column kernel: w*ps(prev), w*E(prev)
//   red[2..4]  local sums of the new state:     w*ps(fut), w*sum_k e_k dpk_k, w*sum_k e_k dbk_k ps(fut)
//   red[16] mass_correction_factor, red[17] temperature_correction, red[18] water_correction_factor
// =====================================================================================================
// block = 64 columns x NW wavefronts (level chunks), like the column kernel (NRED, NPART, FixerArgs, the totals and the scalars: in front of the column kernel)
__global__ __launch_bounds__(512) void k_fixer_sums(Geom g, const double *__restrict__ u, const double *__restrict__ v,
                                                    const double *__restrict__ t, const double *__restrict__ psg,
                                                    const double *__restrict__ dpk, const double *__restrict__ dbk,
                                                    const double *__restrict__ wts, double *__restrict__ partials, int CH,
                                                    const double *__restrict__ wcol, const double *__restrict__ w0blk, int n_w0) {
  __shared__ double sred[4][8];
  const int tid = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
  const int col = blockIdx.x * 64 + tid;
  const int jl = col / g.I;
  const size_t c2 = col, lev = (size_t)g.Jl * g.I;
  const int k0 = w * CH, nk = min(g.L, k0 + CH) - k0;
  const double wgt = wts[jl], ps = psg[c2];
  double uu[8], vv[8], tt[8];                       // CH <= 8: all loads of the thread in flight together
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const size_t q = c2 + (size_t)min(k0 + i, g.L - 1) * lev;
    uu[i] = u[q]; vv[i] = v[q]; tt[i] = t[q];
  }
  double t0 = 0., t1 = 0., t2 = 0., t3 = 0., t4 = 0.;
  if (wcol && w == 0) {   // water fixer column sums left by the tracer kernel: before, after (dpk part, dbk*ps part), masked
    t0 = w0blk ? (col < n_w0 ? w0blk[col] : 0.0) : wgt * wcol[c2];
    t1 = wgt * wcol[lev + c2]; t2 = wgt * wcol[2 * lev + c2] * ps;
    t3 = wgt * wcol[3 * lev + c2]; t4 = wgt * wcol[4 * lev + c2] * ps;
  }
  double sa = 0.0, sb = 0.0;
  double tmn = tt[0], tmx = tt[0];                  // valid_range_t check of the new temperatures (spectral_dynamics.F90:940)
  bool nan = false;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < nk) {
      const double e = 0.5 * (uu[i] * uu[i] + vv[i] * vv[i]) + CP_AIR * tt[i];
      sa += e * dpk[k0 + i];
      sb += e * dbk[k0 + i];
      tmn = fmin(tmn, tt[i]); tmx = fmax(tmx, tt[i]);
      nan = nan || !(tt[i] == tt[i]);
    }
  }
  if (nan) tmx = INFINITY;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    tmn = fmin(tmn, __shfl_xor(tmn, off, 64));
    tmx = fmax(tmx, __shfl_xor(tmx, off, 64));
  }
  if (tid == 0) { sred[2][w] = tmn; sred[3][w] = tmx; }
  double s0 = (w == 0) ? wgt * ps : 0.0, s1 = wgt * sa, s2 = wgt * sb * ps;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    s0 += __shfl_down(s0, off, 64);
    s1 += __shfl_down(s1, off, 64);
    s2 += __shfl_down(s2, off, 64);
  }
  if (tid == 0) { sred[0][w] = s1; sred[1][w] = s2; }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      t0 += __shfl_down(t0, off, 64); t1 += __shfl_down(t1, off, 64); t2 += __shfl_down(t2, off, 64);
      t3 += __shfl_down(t3, off, 64); t4 += __shfl_down(t4, off, 64);
    }
  }
  if (threadIdx.x == 0) {
    double a1 = 0.0, a2 = 0.0, bmn = sred[2][0], bmx = sred[3][0];
    for (int ww = 0; ww < NW; ++ww) { a1 += sred[0][ww]; a2 += sred[1][ww]; bmn = fmin(bmn, sred[2][ww]); bmx = fmax(bmx, sred[3][ww]); }
    double *p = partials + NPART * (size_t)blockIdx.x;
    p[0] = s0; p[1] = a1; p[2] = a2; p[8] = bmn; p[9] = bmx;
    p[3] = t0; p[4] = t1; p[5] = t2; p[6] = t3; p[7] = t4;
  }
}
// red[0..9] <- totals: for the all-reduce between the phases when world_size > 1, and for k_fixer_apply (the eager path)
__global__ __launch_bounds__(512) void k_fixer_reduce(const double *__restrict__ pprev, const double *__restrict__ pfut, int nb,
                                                      double *__restrict__ red) {
  __shared__ double sh[FT_GROUPS][16];
  double tot[NRED], tmn, tmx;
  fixer_totals(pprev, pfut, nb, sh, tot, tmn, tmx);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < NRED; ++c) red[c] = tot[c];
    red[20] = fmin(red[20], tmn); red[21] = fmax(red[21], tmx);      // running extremes of this rank's band
  }
}
// fixer_finish_body as a kernel: behind the all-reduce of a sharded step, and wherever the scalars are needed before the next column kernel runs
// (the host reads state, diagnostics, a physics package in front of the column kernel).  On the plain one-rank path block 0 of the NEXT step's
// column kernel does this instead (ColumnArgs::fin).
__global__ __launch_bounds__(512) void k_fixer_finish(Geom g, FixerArgs a) {
  __shared__ double sh[FT_GROUPS][16];
  double factor, tcorr, wfac;
  fixer_finish_body(g, a, sh, factor, tcorr, wfac);
}
// Every block sums the block partials itself (same fixed order everywhere; world_size > 1: reads the all-reduced
// red[0..9]), derives the fixer scalars and applies them to its slice of psg / tg / tracer; block 0 also patches the (0,0) spectral coefficients, including the Robert-filtered `current` level (:1231,1241,1470-1473).
// Scalars: red[16] mass factor, red[17] temperature correction, red[18] water factor.
__global__ __launch_bounds__(256) void k_fixer_apply(Geom g, FixerArgs a) {
  double r_[NRED];
#pragma unroll
  for (int c = 0; c < NRED; ++c) r_[c] = a.red[c];
  const SpecPatch sp = (blockIdx.x == 0) ? fixer_patch_load(g, a) : SpecPatch{0., 0., 0., 0.};
  double factor, tcorr, wfac;
  fixer_scalars(r_, a, factor, tcorr, wfac);
  // 32-bit element indices (a 3-D field has < 2^31 elements): the 64-bit i / lev would expand into a branchy routine
  const unsigned lev = (unsigned)(g.Jl * g.I), n3 = lev * (unsigned)g.L;
  const unsigned first = (blockIdx.x * 256u + threadIdx.x) * 2u, stride = gridDim.x * 512u;
  constexpr int UN = 4;                                   // independent 16-byte loads in flight per array and lane
  for (unsigned i0 = first; i0 < n3; i0 += UN * stride) {
    double2 tv[UN], fv[UN], cv[UN];
    int km0[UN], km1[UN], kk[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const unsigned i = min(i0 + u * stride, n3 - 2u);
      tv[u] = *(const double2 *)(a.tg + i);
      if (a.tr_fut) {
        kk[u] = (int)(i / lev);
        const unsigned c2 = i - (unsigned)kk[u] * lev;
        fv[u] = *(const double2 *)(a.tr_fut + i); cv[u] = *(const double2 *)(a.tr_cur + i);
        km0[u] = kmask_byte(a.kmask[c2], 0); km1[u] = kmask_byte(a.kmask[c2 + 1], 0);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const unsigned i = i0 + u * stride;
      if (i < n3) {
        *(double2 *)(a.tg + i) = make_double2(tv[u].x + tcorr, tv[u].y + tcorr);
        if (a.tr_fut) {   // water factor where p_full >= limit, then leapfrog part B (:1484) and atmosphere_mod's copy (:1028)
          double2 f = fv[u], c = cv[u];
          f.x = water_corr(f.x, kk[u], km0[u], wfac); f.y = water_corr(f.y, kk[u], km1[u], wfac);
          c.x = fma(a.robert * a.raw, f.x, c.x); c.y = fma(a.robert * a.raw, f.y, c.y);
          *(double2 *)(a.tratm_fut + i) = f;                  // atmosphere_mod's copy is taken before the filter is completed (:1028)
          if (a.tr_part) {                                    // leapfrog_2level_B's future half (leapfrog.F90:102)
            const double2 pt = *(const double2 *)(a.tr_part + i);
            f.x += a.robert * (a.raw - 1.0) * (pt.x + f.x); f.y += a.robert * (a.raw - 1.0) * (pt.y + f.y);
          }
          *(double2 *)(a.tr_fut + i) = f; *(double2 *)(a.tr_cur + i) = c;
        }
      }
    }
  }
  for (unsigned i = first; i < lev; i += stride) {
    double2 p = *(double2 *)(a.psg + i); p.x = mul_nc(p.x, factor); p.y = mul_nc(p.y, factor); *(double2 *)(a.psg + i) = p;
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) { a.red[16] = factor; a.red[17] = tcorr; a.red[18] = wfac; }
    fixer_patch_spectral(g, a, sp, factor, tcorr);
  }
}
// What is pending on the two time levels, applied in place (before the host reads or writes state, restart files, diagnostics):
// afterwards tg, psg, tr and tr_atm of both levels hold what the eager k_fixer_apply would have left.
struct MaterializeArgs {
  double *tg[2], *psg[2], *tr[2], *tr_atm[2];
  const double *pend;           // [2][4]
  const int *kmask;
  int tstate[2];                // tracer buffers: TR_MAT / TR_NEW / TR_FILT
  int mbyte[2];                 // byte of the water-mask word that belongs to each buffer
  int thermo[2];                // mass factor / temperature correction pending on this level
  double robert;
};
__global__ __launch_bounds__(256) void k_fixer_materialize(Geom g, MaterializeArgs a) {
  const unsigned lev = (unsigned)(g.Jl * g.I), n3 = lev * (unsigned)g.L;
  const unsigned i = (blockIdx.x * 256u + threadIdx.x) * 2u;
  if (i >= n3) return;
  double w[2], tc[2], fac[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) { fac[b] = a.pend[4 * b + PEND_FACTOR]; tc[b] = a.pend[4 * b + PEND_TCORR]; w[b] = a.pend[4 * b + PEND_WFAC]; }
  const int k = (int)(i / lev);
  const unsigned c2 = i - (unsigned)k * lev;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if (a.thermo[b]) {
      double2 t = *(double2 *)(a.tg[b] + i); t.x += tc[b]; t.y += tc[b]; *(double2 *)(a.tg[b] + i) = t;
      if (i < lev) { double2 p = *(double2 *)(a.psg[b] + i); p.x = mul_nc(p.x, fac[b]); p.y = mul_nc(p.y, fac[b]); *(double2 *)(a.psg[b] + i) = p; }
    }
  }
  if (a.tstate[0] == TR_MAT && a.tstate[1] == TR_MAT) return;
  const int kw0 = a.kmask[c2], kw1 = a.kmask[c2 + 1];
  double2 U[2], F[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    U[b] = *(const double2 *)(a.tr_atm[b] + i); F[b] = *(const double2 *)(a.tr[b] + i);
    if (a.tstate[b] != TR_MAT) {
      U[b].x = water_corr(U[b].x, k, kmask_byte(kw0, a.mbyte[b]), w[b]); U[b].y = water_corr(U[b].y, k, kmask_byte(kw1, a.mbyte[b]), w[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if (a.tstate[b] == TR_MAT) continue;
    double2 f = U[b];                                                       // TR_NEW: the level itself
    if (a.tstate[b] == TR_FILT) { f.x = fma(a.robert, U[1 - b].x, F[b].x); f.y = fma(a.robert, U[1 - b].y, F[b].y); }
    *(double2 *)(a.tr_atm[b] + i) = U[b]; *(double2 *)(a.tr[b] + i) = f;
  }
}

void launch_fixer_sums(const isca_dyn &h, int fut, hipStream_t s) {
  const Geom &g = h.g;
  const Dev &d = h.d;
  const int nb = (int)column_partials_count(h);
  double *p2 = d.partials + 2 * (size_t)nb;
  const int CH = (g.L + 7) / 8, NW = (g.L + CH - 1) / CH;
  const double *wcol = h.tracer_on ? d.wcol : (const double *)nullptr;
  const double *w0blk = (h.tracer_on && h.tr_filt_horiz) ? d.w0blk : (const double *)nullptr;
  const int n_w0 = g.L * ((g.Jl + TR_RB - 1) / TR_RB);
  hipLaunchKernelGGL(k_fixer_sums, dim3(nb), dim3(64 * NW), 0, s, g, d.ug[fut], d.vg[fut], d.tg[fut], d.psg[fut], d.dpk, d.dbk, d.wts_lat_l, p2, CH, wcol, w0blk, n_w0);
  // the totals for the all-reduce of red[0..9] between the phases (world_size > 1) and for k_fixer_apply (eager fixers); with lazy fixers
  // on one rank k_fixer_finish folds them itself
  if (g.P > 1 || !h.lazy_fix) {
    hipLaunchKernelGGL(k_fixer_reduce, dim3(1), dim3(nb > 256 ? 512 : 256), 0, s, d.partials, p2, nb, d.red);
  }
}
static FixerArgs fixer_args(const isca_dyn &h, const StepScalars &sc) {
  const Geom &g = h.g;
  FixerArgs a;
  const int nb = (int)column_partials_count(h);
  a.red = h.d.red; a.pprev = h.d.partials; a.pfut = h.d.partials + 2 * (size_t)nb; a.nb = nb;
  a.pend_fut = h.d.pend + 4 * sc.fut;
  a.reduce_here = (g.P == 1 && h.lazy_fix);      // k_fixer_finish folds the partials itself; otherwise k_fixer_reduce (+ the all-reduce) left red[0..9]
  a.lnps_fut = (double2 *)h.d.lnps[sc.fut]; a.lnps_cur = (double2 *)h.d.lnps[sc.cur];
  a.ts_fut = (double2 *)h.d.ts[sc.fut]; a.ts_cur = (double2 *)h.d.ts[sc.cur];
  a.psg = h.d.psg[sc.fut]; a.tg = h.d.tg[sc.fut];
  a.tr_fut = h.tracer_on ? h.d.tr[sc.fut] : nullptr; a.tr_cur = h.d.tr[sc.cur]; a.tratm_fut = h.d.tr_atm[sc.fut];
  a.kmask = h.d.kmask; a.do_water = h.cfg.do_water_correction;
  a.ml0 = h.ml_of_m0;
  double sumw = 0.0;
  for (double w : h.tab.wts_lat) sumw += w;
  a.sumw_nlon = sumw * g.I;
  a.robert = h.cfg.robert_coeff; a.raw = h.cfg.raw_filter_coeff; a.tr_part = h.d.tr_part;
  a.do_mass = h.cfg.do_mass_correction; a.do_energy = h.cfg.do_energy_correction;
  a.patch = 1;
  return a;
}
void launch_fixer_apply(const isca_dyn &h, const StepScalars &sc, hipStream_t s) {
  const Geom &g = h.g;
  const FixerArgs a = fixer_args(h, sc);
  const size_t n3 = (size_t)g.Jl * g.I * g.L;
  const unsigned nblk = (unsigned)std::min<size_t>(1024, (n3 / 2 + 255) / 256);
  hipLaunchKernelGGL(k_fixer_apply, dim3(nblk), dim3(256), 0, s, g, a);
}
void launch_fixer_finish(const isca_dyn &h, const StepScalars &sc, hipStream_t s) {
  const FixerArgs a = fixer_args(h, sc);
  hipLaunchKernelGGL(k_fixer_finish, dim3(1), dim3(a.nb > 256 ? 512 : 256), 0, s, h.g, a);
}
// tstate / thermo: what is pending on time levels 0 and 1; cur_level: the level whose water mask is byte 0 of the mask word (the newest)
void launch_fixer_materialize(const isca_dyn &h, hipStream_t s) {
  const Geom &g = h.g;
  const Dev &d = h.d;
  MaterializeArgs a;
  for (int b = 0; b < 2; ++b) {
    a.tg[b] = d.tg[b]; a.psg[b] = d.psg[b]; a.tr[b] = d.tr[b]; a.tr_atm[b] = d.tr_atm[b];
    a.tstate[b] = h.tracer_on ? h.tr_state[b] : TR_MAT; a.thermo[b] = h.thermo_pending[b] ? 1 : 0;
    a.mbyte[b] = (b == h.current) ? 0 : 1;
  }
  a.pend = d.pend; a.kmask = d.kmask; a.robert = h.cfg.robert_coeff;
  const unsigned n3 = (unsigned)(g.Jl * g.I * g.L);
  hipLaunchKernelGGL(k_fixer_materialize, dim3((n3 / 2 + 255) / 256), dim3(256), 0, s, g, a);
}

// =====================================================================================================
// Shallow-water sibling core (src/atmos_spectral_shallow): the grid-point and spectral parts of one step; the transforms
// are the 3-D core's own kernels run with one level.
// =====================================================================================================
// shallow_physics (shallow_physics.F90:179-194: linear drag, relaxation of h to h_eq, both at the PREVIOUS level) +
// grid part of shallow_dynamics (shallow_dynamics.F90:421-437): vorticity flux terms, h tendency, Bernoulli function; pv (:469)
__global__ void k_sw_grid_tend(SwGridArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int j = i / a.I;
  const double u = a.u[i], v = a.v[i], h = a.h[i];
  const double vorg = a.vor[i] + a.coriolis[j];
  a.tend_u[i] = (0.0 - a.kappa_m * a.up[i]) + vorg * v;
  a.tend_v[i] = (0.0 - a.kappa_m * a.vp[i]) - vorg * u;
  a.tend_h[i] = ((0.0 - a.kappa_t * (a.hp[i] - a.h_eq[i])) - u * a.dxh[i] - v * a.dyh[i]) - h * a.div[i];
  a.bg[i] = (h + a.deep[i]) + 0.5 * (u * u + v * v);
  a.pv[i] = vorg / h;
}
// barotropic_dynamics (barotropic_dynamics.F90:300-306): absolute vorticity and the vorticity-flux tendencies (physics is empty)
__global__ void k_bt_grid_tend(int n, int I, const double *u, const double *v, const double *vor, const double *coriolis, double *tend_u,
                               double *tend_v, double *pv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double q = vor[i] + coriolis[i / I];
  pv[i] = q;
  tend_u[i] = 0.0 + q * v[i];
  tend_v[i] = 0.0 - q * u[i];
}
void launch_bt_grid_tend(int n, int I, const double *u, const double *v, const double *vor, const double *coriolis, double *tend_u, double *tend_v,
                         double *pv, hipStream_t s) {
  hipLaunchKernelGGL(k_bt_grid_tend, grid1d((size_t)n), dim3(256), 0, s, n, I, u, v, vor, coriolis, tend_u, tend_v, pv);
}
void launch_sw_grid_tend(const SwGridArgs &a, hipStream_t s) { hipLaunchKernelGGL(k_sw_grid_tend, grid1d((size_t)a.n), dim3(256), 0, s, a); }

__device__ __forceinline__ void sw_leapfrog(bool first, double delta_t, double robert, double2 prev, double2 cur, double2 dt, double2 &fut,
                                            double2 &cur_new) {   // leapfrog_3d_complex with raw_filter_coeff = 1 (leapfrog.F90:217-247)
  const double2 pc = make_double2(prev.x - 2.0 * cur.x, prev.y - 2.0 * cur.y);
  if (first) {
    fut = make_double2(prev.x + delta_t * dt.x, prev.y + delta_t * dt.y);
    cur_new = make_double2(cur.x + robert * (pc.x + fut.x), cur.y + robert * (pc.y + fut.y));
  } else {
    const double2 c1 = make_double2(cur.x + robert * pc.x, cur.y + robert * pc.y);
    fut = make_double2(prev.x + delta_t * dt.x, prev.y + delta_t * dt.y);
    cur_new = make_double2(c1.x + robert * fut.x, c1.y + robert * fut.y);
  }
}
// spectral part of shallow_dynamics (:438-455): Laplacian of the Bernoulli function, implicit_correction (:476-495), spectral
// damping of the three tendencies (spectral_damping.F90:172-199), leapfrog with Robert filter.  mode 1: a spectral tracer instead
// (update_spec_tracer :505-511: damping + leapfrog of one field).
__global__ void k_sw_spec_update(Geom g, SwSpecArgs a) {
  const int mn = blockIdx.x * blockDim.x + threadIdx.x;
  if (mn >= g.Ml * g.N1) return;
  const double eig = a.coef[(size_t)C_EIG * g.Ml * g.N1 + mn], dmp = a.coef[(size_t)C_DAMP * g.Ml * g.N1 + mn] + a.damping_r;
  const double coeff = 1.0 / (1.0 + dmp * a.delta_t);
  if (a.mode == 1) {
    const double2 p = a.vor_p[mn], c = a.vor_c[mn];
    double2 dt = a.dt_vor[mn];
    dt = make_double2(coeff * (dt.x - dmp * p.x), coeff * (dt.y - dmp * p.y));
    if (a.stir) { const double2 st = a.stir[mn]; dt = make_double2(dt.x + st.x, dt.y + st.y); }
    double2 f, cn;
    sw_leapfrog(a.first, a.delta_t, a.robert, p, c, dt, f, cn);
    a.vor_c[mn] = cn; a.vor_f[mn] = f;
    return;
  }
  const double2 vp = a.vor_p[mn], vc = a.vor_c[mn], dp = a.div_p[mn], dc = a.div_c[mn], hp = a.h_p[mn], hc = a.h_c[mn];
  double2 dtv = a.dt_vor[mn], dtd = a.dt_div[mn], dth = a.dt_h[mn];
  const double2 bs = a.bs[mn];
  const double mu = 0.5 * a.delta_t, mu2 = mu * mu;
  dtd = make_double2(dtd.x - (-(eig * bs.x)), dtd.y - (-(eig * bs.y)));              // dt_divs - compute_laplacian(bs)
  dth = make_double2(dth.x + a.h_0 * (dc.x - dp.x), dth.y + a.h_0 * (dc.y - dp.y));
  dtd = make_double2(dtd.x - eig * (hc.x - hp.x), dtd.y - eig * (hc.y - hp.y));
  const double den = 1.0 + mu2 * eig * a.h_0;
  dtd = make_double2((dtd.x + mu * eig * dth.x) / den, (dtd.y + mu * eig * dth.y) / den);
  dth = make_double2(dth.x - mu * a.h_0 * dtd.x, dth.y - mu * a.h_0 * dtd.y);
  dtv = make_double2(coeff * (dtv.x - dmp * vp.x), coeff * (dtv.y - dmp * vp.y));
  if (a.stir) { const double2 st = a.stir[mn]; dtv = make_double2(dtv.x + st.x, dtv.y + st.y); }
  dtd = make_double2(coeff * (dtd.x - dmp * dp.x), coeff * (dtd.y - dmp * dp.y));
  dth = make_double2(coeff * (dth.x - dmp * hp.x), coeff * (dth.y - dmp * hp.y));
  double2 f, cn;
  sw_leapfrog(a.first, a.delta_t, a.robert, vp, vc, dtv, f, cn); a.vor_c[mn] = cn; a.vor_f[mn] = f;
  sw_leapfrog(a.first, a.delta_t, a.robert, dp, dc, dtd, f, cn); a.div_c[mn] = cn; a.div_f[mn] = f;
  sw_leapfrog(a.first, a.delta_t, a.robert, hp, hc, dth, f, cn); a.h_c[mn] = cn; a.h_f[mn] = f;
}
void launch_sw_spec_update(const Geom &g, const SwSpecArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_sw_spec_update, grid1d((size_t)g.Ml * g.N1), dim3(256), 0, s, g, a);
}
// stirring (stirring.F90:216-224): the freshly localised forcing loses its (0,0) coefficient and enters the AR(1) state
__global__ void k_sw_stir_update(int n, double bstir, int mn00, const double2 *__restrict__ fresh, double2 *s_stir) {
  const int mn = blockIdx.x * blockDim.x + threadIdx.x;
  if (mn >= n) return;
  double2 f = fresh[mn];
  if (mn == mn00) f = make_double2(0.0, 0.0);
  const double2 s = s_stir[mn];
  s_stir[mn] = make_double2(bstir * s.x + f.x, bstir * s.y + f.y);
}
void launch_sw_stir_update(const Geom &g, double bstir, int mn00, const double *fresh, double *s_stir, hipStream_t s) {
  const int n = g.Ml * g.N1;
  hipLaunchKernelGGL(k_sw_stir_update, grid1d((size_t)n), dim3(256), 0, s, n, bstir, mn00, (const double2 *)fresh, (double2 *)s_stir);
}
__global__ void k_sw_scale_grid(int n, const double *__restrict__ factor, double *field) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) field[i] = factor[i] * field[i];
}
void launch_sw_scale_grid(int n, const double *factor, double *field, hipStream_t s) {
  hipLaunchKernelGGL(k_sw_scale_grid, grid1d((size_t)n), dim3(256), 0, s, n, factor, field);
}
// update_grid_tracer (:521-530) after the van Leer step: Robert filter of the current level, new level stored
__global__ void k_sw_grid_tracer_filter(int n, double robert, const double *__restrict__ prev, double *cur, const double *__restrict__ adv,
                                        double *fut) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double p = prev[i], c = cur[i], f = adv[i];
  const double cn = c + robert * (p + f - 2.0 * c);
  cur[i] = cn; fut[i] = f;
}
void launch_sw_grid_tracer_filter(int n, double robert, const double *prev, double *cur, const double *adv, double *fut, hipStream_t s) {
  hipLaunchKernelGGL(k_sw_grid_tracer_filter, grid1d((size_t)n), dim3(256), 0, s, n, robert, prev, cur, adv, fut);
}
// tend = tend_in - u dx - v dy for the spectral tracer (update_spec_tracer: horizontal_advection with a zero tendency)
__global__ void k_sw_tracer_tend(int n, const double *u, const double *v, const double *dx, const double *dy, double *tend) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tend[i] = 0.0 - u[i] * dx[i] - v[i] * dy[i];
}
void launch_sw_tracer_tend(int n, const double *u, const double *v, const double *dx, const double *dy, double *tend, hipStream_t s) {
  hipLaunchKernelGGL(k_sw_tracer_tend, grid1d((size_t)n), dim3(256), 0, s, n, u, v, dx, dy, tend);
}

// =====================================================================================================
// Diagnostics: what spectral_diagnostics hands to diag_manager's send_data every step
// (spectral_dynamics.F90:1728-1790), accumulated on the device for the time means of the diag_table.
// u, v, T, ps, tracer: the new time level; omega = wg_full; vor, div: vorg/divg, already those of the new level
// (:933-934).
// =====================================================================================================
const char *const DIAG_NAMES[NDIAG] = {"ps", "ucomp", "vcomp", "temp", "vor", "div", "omega", "sphum", "ucomp_sq", "vcomp_sq",
                                       "ucomp_vcomp", "temp_sq", "ucomp_temp", "vcomp_temp", "omega_sq", "omega_temp",
                                       "ucomp_omega", "vcomp_omega", "vcomp_vor", "wspd",
                                       "precipitation", "t_surf"};     // idealized_moist_phys.F90:672, mixed_layer.F90:359
struct DiagArgs {
  const double *u, *v, *t, *ps, *vor, *div, *w, *tr, *precip, *t_surf;
  double *acc[NDIAG];
  unsigned mask;
  unsigned n3, n2;
};
__global__ __launch_bounds__(256) void k_diag_accumulate(DiagArgs a) {
  const unsigned i = (blockIdx.x * 256u + threadIdx.x) * 2u;
  if (i >= a.n3) return;
  const unsigned m = a.mask;
  double2 u = {0, 0}, v = {0, 0}, t = {0, 0}, vo = {0, 0}, dv = {0, 0}, w = {0, 0}, q = {0, 0};
  // inputs first (only those some selected field needs), then the read-modify-writes of the sums
  if (m & 0x91502u) u = *(const double2 *)(a.u + i);          // ucomp, ucomp_sq, ucomp_vcomp, ucomp_temp, ucomp_omega, wspd
  if (m & 0xE2604u) v = *(const double2 *)(a.v + i);          // vcomp, vcomp_sq, ucomp_vcomp, vcomp_temp, vcomp_omega, vcomp_vor, wspd
  if (m & 0x0B808u) t = *(const double2 *)(a.t + i);          // temp, temp_sq, ucomp_temp, vcomp_temp, omega_temp
  if (m & 0x40010u) vo = *(const double2 *)(a.vor + i);       // vor, vcomp_vor
  if (m & 0x00020u) dv = *(const double2 *)(a.div + i);
  if (m & 0x3C040u) w = *(const double2 *)(a.w + i);          // omega, omega_sq, omega_temp, ucomp_omega, vcomp_omega
  if ((m & 0x00080u) && a.tr) q = *(const double2 *)(a.tr + i);
#define ACC(bit, ex, ey)                                                                   \
  if (m & (1u << (bit))) { double2 s = *(double2 *)(a.acc[bit] + i); s.x += (ex); s.y += (ey); *(double2 *)(a.acc[bit] + i) = s; }
  ACC(1, u.x, u.y) ACC(2, v.x, v.y) ACC(3, t.x, t.y) ACC(4, vo.x, vo.y) ACC(5, dv.x, dv.y) ACC(6, w.x, w.y) ACC(7, q.x, q.y)
  ACC(8, u.x * u.x, u.y * u.y) ACC(9, v.x * v.x, v.y * v.y) ACC(10, u.x * v.x, u.y * v.y) ACC(11, t.x * t.x, t.y * t.y)
  ACC(12, u.x * t.x, u.y * t.y) ACC(13, v.x * t.x, v.y * t.y) ACC(14, w.x * w.x, w.y * w.y) ACC(15, w.x * t.x, w.y * t.y)
  ACC(16, u.x * w.x, u.y * w.y) ACC(17, w.x * v.x, w.y * v.y) ACC(18, v.x * vo.x, v.y * vo.y)
  ACC(19, sqrt(u.x * u.x + v.x * v.x), sqrt(u.y * u.y + v.y * v.y))
#undef ACC
  if ((m & 1u) && i < a.n2) { double2 s = *(double2 *)(a.acc[0] + i); const double2 p = *(const double2 *)(a.ps + i); s.x += p.x; s.y += p.y; *(double2 *)(a.acc[0] + i) = s; }
  if ((m & (1u << 20)) && i < a.n2) { double2 s = *(double2 *)(a.acc[20] + i); const double2 p = *(const double2 *)(a.precip + i); s.x += p.x; s.y += p.y; *(double2 *)(a.acc[20] + i) = s; }
  if ((m & (1u << 21)) && i < a.n2) { double2 s = *(double2 *)(a.acc[21] + i); const double2 p = *(const double2 *)(a.t_surf + i); s.x += p.x; s.y += p.y; *(double2 *)(a.acc[21] + i) = s; }
}
void launch_diag_accumulate(const isca_dyn &h, int fut, hipStream_t s) {
  const Geom &g = h.g;
  const Dev &d = h.d;
  DiagArgs a;
  a.u = d.ug[fut]; a.v = d.vg[fut]; a.t = d.tg[fut]; a.ps = d.psg[fut]; a.vor = d.vorg; a.div = d.divg; a.w = d.wg_full;
  a.tr = h.tracer_on ? d.tr[fut] : nullptr;
  a.precip = d.precip; a.t_surf = d.t_surf;
  for (int i = 0; i < NDIAG; ++i) a.acc[i] = d.diag_acc[i];
  a.mask = h.diag_mask;
  a.n2 = (unsigned)(g.Jl * g.I); a.n3 = a.n2 * (unsigned)g.L;
  hipLaunchKernelGGL(k_diag_accumulate, dim3((a.n3 / 2 + 255) / 256), dim3(256), 0, s, a);
}

}  // namespace isca
