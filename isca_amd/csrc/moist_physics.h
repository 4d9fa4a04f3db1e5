// Column physics of the moist configuration (config 3: Frierson grey-radiation aquaplanet), restated from the reference's
// atmos_param / coupler modules as functions of ONE column, usable on the device (moist.hip) and on the host (the CPU
// tests compile this header with g++ and compare every routine with the reference's outputs).  A column field is addressed as x[k * s] (s = level
// stride: lat*lon on the device grid layout [lev][lat][lon]); k = 0 is the model top.
// Options: those of exp/test_cases/frierson/frierson_test_case.py:49-170 (do_simple everywhere, no snow, no virtual
// temperature, grey radiation 'frierson', SIMPLE_BETTS_MILLER convection, diffusivity PBL, mixed-layer surface).
#pragma once
#include <cmath>
#include <type_traits>
#if defined(__clang__)
#pragma clang fp contract(off)      // the reference is built without FMA contraction; the convective regime tests are knife-edge
#endif
#if defined(__HIPCC__)
#define MP_HD __host__ __device__ __forceinline__
#define MP_UNROLL _Pragma("unroll 4")
#else
#define MP_HD inline
#define MP_UNROLL
#endif
#define MP_R __restrict__
// Level loops run in chunks of MP_U levels: all global loads of a chunk are issued first (one memory round trip per chunk
// instead of one per level: a column is a chain of dependent recurrences, and a GPU thread walking it level by level waits a
// full memory latency per level), then the recurrence is advanced, then the chunk's results are stored.
#ifndef MP_U
#define MP_U 8
#endif
#if defined(__HIPCC__)
#define MP_UNROLL_ALL _Pragma("unroll")
#else
#define MP_UNROLL_ALL
#endif

namespace moist {

// shared/constants/constants.F90
constexpr double GRAV = 9.80, RDGAS = 287.04, RVGAS = 461.50, KAPPA = 2.0 / 7.0, CP_AIR = RDGAS / KAPPA;
constexpr double HLV = 2.500e6, TFREEZE = 273.16, DENS_H2O = 1000., STEFAN = 5.6734e-8, VONKARM = 0.40;
constexpr double PSTD_MKS = 101325.0, EPSILO = RDGAS / RVGAS;   // d622 of sat_vapor_pres

// ------------------------------------------------------------------------------------------------
// sat_vapor_pres (shared/sat_vapor_pres/sat_vapor_pres_k.F90:161-265, 1132-1300), do_simple = .true.:
// table of es = 610.78 exp(-hlv/rvgas (1/T - 1/tfreeze)) every 0.1 K from -173 C to 350 C with first derivative and half
// second derivative; lookups are the reference's quadratic interpolation.  Built on the host (SatTableHost), read here.
// ------------------------------------------------------------------------------------------------
struct SatTable {
  const double *tab, *dtab, *d2tab;
  double tmin, dtinv, teps, dtres;
  int n;
};
MP_HD void lookup_es_des(const SatTable &t, double temp, double &es, double &des) {
  const double tmp = temp - t.tmin;
  const int ind = (int)(t.dtinv * (tmp + t.teps));
  // 'table overflow' is FATAL in the reference; NaN marks it here.  No branch in front of the table reads (clamped index, select
  // afterwards): behind one, the reads of several lookups in an unrolled loop could not be issued together.
  const bool overflow = ind < 0 || ind >= t.n;
  const int ic = overflow ? 0 : ind;
  const double t0 = t.tab[ic], t1 = t.dtab[ic], t2 = t.d2tab[ic];
  const double del = tmp - t.dtres * (double)ind;
  es = overflow ? NAN : t0 + del * (t1 + del * t2);
  des = overflow ? NAN : t1 + 2. * del * t2;
}
// the same for N temperatures at once, in three passes (indices, table reads, interpolation): written per value the compiler keeps each
// lookup's reads behind the previous lookup's wait, and a chunk of N levels costs N memory round trips instead of one
template <int N>
MP_HD void lookup_es_des_n(const SatTable &t, const double (&temp)[N], double (&es)[N], double (&des)[N]) {
  int ic[N]; double del[N], t0[N], t1[N], t2[N]; bool overflow[N];
  MP_UNROLL_ALL
  for (int i = 0; i < N; ++i) {
    const double tmp = temp[i] - t.tmin;
    const int ind = (int)(t.dtinv * (tmp + t.teps));
    overflow[i] = ind < 0 || ind >= t.n;
    ic[i] = overflow[i] ? 0 : ind;
    del[i] = tmp - t.dtres * (double)ind;
  }
  MP_UNROLL_ALL
  for (int i = 0; i < N; ++i) { t0[i] = t.tab[ic[i]]; t1[i] = t.dtab[ic[i]]; t2[i] = t.d2tab[ic[i]]; }
  MP_UNROLL_ALL
  for (int i = 0; i < N; ++i) {
    es[i] = overflow[i] ? NAN : t0[i] + del[i] * (t1[i] + del[i] * t2[i]);
    des[i] = overflow[i] ? NAN : t1[i] + 2. * del[i] * t2[i];
  }
}
MP_HD double lookup_es(const SatTable &t, double temp) { double e, d; lookup_es_des(t, temp, e, d); return e; }
// compute_qs (sat_vapor_pres_k.F90:457-540) without q: qs = eps es / (p - (1-eps) es), dqs/dT = eps p des / denom^2; N values at once
template <int N>
MP_HD void compute_qs_n(const SatTable &t, const double (&temp)[N], const double (&press)[N], double (&qs)[N], double (&dqsdT)[N]) {
  double es[N], des[N];
  lookup_es_des_n<N>(t, temp, es, des);
  MP_UNROLL_ALL
  for (int i = 0; i < N; ++i) {
    const double denom = press[i] - (1.0 - EPSILO) * es[i];
    qs[i] = (denom > 0.0) ? EPSILO * es[i] / denom : EPSILO;
    dqsdT[i] = EPSILO * press[i] * des[i] / (denom * denom);
  }
}

// ------------------------------------------------------------------------------------------------
// lscale_cond (atmos_param/lscale_cond/lscale_cond.F90:79-212) with do_simple, do_evap, hc = 1:
// saturation adjustment where q > qsat, re-evaporation of the falling precipitation in the layers below
// (precip_evap :215-252); returns the deltas (not rates) and the rain in kg/m2.
// ------------------------------------------------------------------------------------------------
// Half-level pressures reach the routines below either as an array (a pointer, read with the stride the routine is given: the reference's
// interface, the host tests, isca_idealized_moist_phys on caller columns) or -- inside the model's step -- as the vertical coordinate and the
// column's surface pressure, p_half(k) = pk(k) + bk(k) ps formed where it is needed: the same values (k_moist_pressures stored exactly this
// expression) without an array that every routine reads again from HBM (seven passes over a level array per step of the column kernel).
struct PHalfSigma { const double *pk, *bk; double ps; };
MP_HD double ph_at(const double *p, int s, int k) { return p[k * s]; }
MP_HD double ph_at(const PHalfSigma &p, int, int k) { return p.pk[k] + p.bk[k] * p.ps; }
// load(k, t, q, x, y): temperature and humidity of level k plus two values of the caller's that come from memory and are handed on
// to out(k, t_delta, q_delta, x, y) - so that everything a chunk of levels reads is requested before anything is stored.
template <class LOAD, class OUT, class PH>
MP_HD void lscale_cond(const SatTable &st, int L, LOAD load, const double *pfull, PH phalf, int s, OUT out, double &rain) {
  const double hlcp = HLV / CP_AIR;
  double exq = 0.0, precip = 0.0;
  double ph_k = ph_at(phalf, s, 0);
  for (int k0 = 0; k0 < L; k0 += MP_U) {
    double tk[MP_U], qk[MP_U], pf[MP_U], phn[MP_U], qsat[MP_U], dqsat[MP_U], tdo[MP_U], qdo[MP_U], ax[MP_U], ay[MP_U];
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = (k0 + i < L) ? k0 + i : L - 1;
      load(k, tk[i], qk[i], ax[i], ay[i]); pf[i] = pfull[k * s]; phn[i] = ph_at(phalf, s, k + 1);
    }
    compute_qs_n<MP_U>(st, tk, pf, qsat, dqsat);
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      double qd = 0.0, td = 0.0;
      if (k0 + i < L) {
        if ((qk[i] - qsat[i]) * qsat[i] > 0.0) {
          qd = (qsat[i] - qk[i]) / (1.0 + hlcp * dqsat[i]);
          td = -hlcp * qd;
        }
        const double pmass = (phn[i] - ph_k) / GRAV;
        if (qd < 0.0) exq = exq - qd * pmass;
        if (qd >= 0.0 && exq > 0.0) {                 // evaporate precip where needed
          exq = exq / pmass;
          double def = (qsat[i] - qk[i]) / (1. + hlcp * dqsat[i]);
          def = fmin(fmax(def, 0.0), exq);
          qd = qd + def;
          td = td - def * hlcp;
          exq = (exq - def) * pmass;
        }
        precip = precip - pmass * qd;
        ph_k = phn[i];
      }
      tdo[i] = td; qdo[i] = qd;
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i)
      if (k0 + i < L) out(k0 + i, tdo[i], qdo[i], ax[i], ay[i]);
  }
  rain = fmax(precip, 0.0);
}

// ------------------------------------------------------------------------------------------------
// two_stream_gray_rad 'frierson' (atmos_param/two_stream_gray_rad/two_stream_gray_rad.F90:386-655 down, :659-776 up)
// with do_seasonal = .false.: annual-mean insolation 0.25 S0 (1 + del_sol P2(lat) + del_sw sin lat), SW absorbed with
// optical depth sw_tau_0 (p/p0)^solar_exponent, grey LW with lw_tau_0(lat) (linear_tau p/p0 + (1-linear_tau)(p/p0)^wv_exponent).
// ------------------------------------------------------------------------------------------------
MP_HD double pow4(double x) { return x * x * x * x; }     // x**4 as the reference build expands it, ((x x) x) x
// (p/p0)**wv_exponent, (p/p0)**solar_exponent with REAL exponents (namelist variables): the library pow, except for the value every test case sets,
// 4, taken as three multiplications -- within 3 ulp of pow, and a third of gray_rad_down's time on the device (pow is ~0.4 us per level there).
MP_HD double gray_pow(double x, double e) { return (e == 4.0) ? pow4(x) : pow(x, e); }
struct GrayRadParams {
  double solar_constant = 1360.0, del_sol = 1.4, del_sw = 0.0, ir_tau_eq = 6.0, ir_tau_pole = 1.5, atm_abs = 0.0, odp = 1.0,
         sw_diff = 0.0, linear_tau = 0.1, wv_exponent = 4.0, solar_exponent = 4.0, diabatic_acce = 1.0;
};
// Downward pass: fills lw_down[0..L] (caller storage, stride sw), lw_dtrans[0..L-1] and sw_down[0..L] (stride ssw; the downward shortwave flux on
// the half levels, which the reference evaluates again in the upward pass: here that pass reads it), returns the surface fluxes.
// (p/p0)^wv_exponent and (p/p0)^solar_exponent are one pow when the two exponents are equal (the defaults: 4).
template <class PH>
MP_HD void gray_rad_down(const GrayRadParams &p, int L, double lat, double albedo, const double *t, PH p_half, int s,
                         double *lw_down, double *lw_dtrans, int sw, double *sw_down, int ssw, double &insolation, double &sw_tau_0,
                         double &net_surf_sw_down, double &surf_lw_down, int slw = -1, bool store_sw = true) {
  // slw: lw_down's own stride (default: lw_dtrans's); store_sw = false: sw_down is not written -- with atm_abs = 0 it is the insolation at every
  // half level, which gray_rad_up can be told (sw_uniform), and the caller may then keep lw_down where sw_down would have been (the device kernel: in LDS)
  const int slwd = slw < 0 ? sw : slw;
  const double sl = sin(lat), sl2 = sl * sl;
  const double p2 = (1. - 3. * sl2) / 4.;
  insolation = 0.25 * p.solar_constant * (1.0 + p.del_sol * p2 + p.del_sw * sl);
  sw_tau_0 = (1.0 - p.sw_diff * sl2) * p.atm_abs;
  double lw_tau_0 = p.ir_tau_eq + (p.ir_tau_pole - p.ir_tau_eq) * sl2;
  lw_tau_0 = lw_tau_0 * p.odp;
  const bool one_pow = p.solar_exponent == p.wv_exponent;
  const double ph_top = ph_at(p_half, s, 0);
  const double pw0 = gray_pow(ph_top / PSTD_MKS, p.wv_exponent);
  double tau_k = lw_tau_0 * (p.linear_tau * ph_top / PSTD_MKS + (1.0 - p.linear_tau) * pw0);
  lw_down[0] = 0.;
  if (store_sw) sw_down[0] = insolation * exp(-sw_tau_0 * (one_pow ? pw0 : gray_pow(ph_top / PSTD_MKS, p.solar_exponent)));
  double lwd = 0., swd_last = 0.;
  for (int k0 = 0; k0 < L; k0 += MP_U) {
    double ph[MP_U], tk[MP_U], tau_n[MP_U], swd[MP_U];
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = (k0 + i < L) ? k0 + i : L - 1;
      ph[i] = ph_at(p_half, s, k + 1); tk[i] = t[k * s];
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const double pw = gray_pow(ph[i] / PSTD_MKS, p.wv_exponent);
      tau_n[i] = lw_tau_0 * (p.linear_tau * ph[i] / PSTD_MKS + (1.0 - p.linear_tau) * pw);
      // (atm_abs = 0, the grey scheme's default: the shortwave optical depth is zero at every latitude and exp(-0 x) = 1 exactly -- not evaluated)
      swd[i] = (p.atm_abs == 0.0) ? insolation : insolation * exp(-sw_tau_0 * (one_pow ? pw : gray_pow(ph[i] / PSTD_MKS, p.solar_exponent)));
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      if (k0 + i < L) {
        const double dtr = exp(-(tau_n[i] - tau_k));
        lw_dtrans[(k0 + i) * sw] = dtr;
        if (store_sw) sw_down[(k0 + i + 1) * ssw] = swd[i];
        swd_last = swd[i];
        const double b = STEFAN * pow4(tk[i]);
        lwd = lwd * dtr + b * (1. - dtr);
        lw_down[(k0 + i + 1) * slwd] = lwd;
        tau_k = tau_n[i];
      }
    }
  }
  surf_lw_down = lwd;
  net_surf_sw_down = swd_last * (1. - albedo);
}
// Upward pass: temperature tendency of the radiative flux divergence, accumulated into tdt.  tdt_is_zero: the caller's tendency so far is zero and
// is not read (0 + x is still formed, so the stored bits are those of an accumulation into a zeroed array) -- tdt may then BE sw_down (same
// stride): level k's shortwave flux is read in the load phase of k's chunk, its heating stored at the chunk's end, and the chunks walk upward.
template <class PH>
MP_HD void gray_rad_up(const GrayRadParams &p, int L, double albedo, double t_surf, const double *t, PH p_half, int s,
                       const double *lw_down, const double *lw_dtrans, int sw, const double *sw_down, int ssw, double *tdt, int st,
                       bool tdt_is_zero = false, int slw = -1, bool sw_uniform = false, double sw_value = 0.0) {
  // slw, sw_uniform / sw_value: see gray_rad_down (tdt may then BE lw_down: a level's flux is read in the load phase of its chunk, like sw_down's)
  const int sl = slw < 0 ? sw : slw;
  const double b_surf = STEFAN * pow4(t_surf);
  const double ph_surf = ph_at(p_half, s, L), sw_surf = sw_uniform ? sw_value : sw_down[L * ssw];
  const double sw_up = albedo * sw_surf;
  double lw_up_n = b_surf;                                   // lw_up at half level k+1, integrating upward
  double flux_n = (lw_up_n - lw_down[L * sl]) + (sw_up - sw_surf);
  double ph_n = ph_surf;
  for (int k0 = L - 1; k0 >= 0; k0 -= MP_U) {
    double tk[MP_U], ph[MP_U], td[MP_U], swd[MP_U], dtrs[MP_U], lwd[MP_U];
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {       // everything the chunk reads, the downward pass's arrays included (they may live in global memory)
      const int k = (k0 - i >= 0) ? k0 - i : 0;
      tk[i] = t[k * s]; ph[i] = ph_at(p_half, s, k); swd[i] = sw_uniform ? sw_value : sw_down[k * ssw]; dtrs[i] = lw_dtrans[k * sw]; lwd[i] = lw_down[k * sl];
      td[i] = tdt_is_zero ? 0.0 : tdt[k * st];
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = k0 - i;
      if (k >= 0) {
        const double b = STEFAN * pow4(tk[i]);
        const double dtr = dtrs[i];
        const double lw_up_k = lw_up_n * dtr + b * (1.0 - dtr);
        const double flux_k = (lw_up_k - lwd[i]) + (sw_up - swd[i]);
        const double tdt_rad = p.diabatic_acce * (flux_n - flux_k) * GRAV / (CP_AIR * (ph_n - ph[i]));
        td[i] = td[i] + tdt_rad;
        lw_up_n = lw_up_k; flux_n = flux_k; ph_n = ph[i];
      }
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i)
      if (k0 - i >= 0) tdt[(k0 - i) * st] = td[i];
  }
}

// ------------------------------------------------------------------------------------------------
// qe_moist_convection: simplified Betts-Miller scheme of Frierson (2007)
// (atmos_param/qe_moist_convection/qe_moist_convection.F90:255-1084).  Levels are numbered 1..L (1 = top) inside, as in
// the reference, so that kLZB / kLCL keep their meaning.  Returns the deltas over the step (not rates) and rain in kg/m2.
// ------------------------------------------------------------------------------------------------
struct QeParams {
  double tau_bm = 7200., rhbm = .8, Tmin = 173., Tmax = 335., val_inc = 0.01, val_min = -1., val_max = 1.;
  const double *lcl_temp_table = nullptr;     // built by QeTablesHost (moist_tables.h)
  int table_size = 0;
};
constexpr double QE_SMALL = 1.e-10, QE_PREF = 1.e5;
MP_HD double qe_mixing_ratio(double vapor_pressure, double pressure) { return RDGAS * vapor_pressure / RVGAS / (pressure - vapor_pressure); }
MP_HD double qe_virtual_temp(double temp, double r) {
  const double q = r / (1.0 + r);
  return temp * (1.0 + q * (RVGAS / RDGAS - 1.0));
}
// get_lcl_temp (:1051-1082): linear interpolation in the table; out-of-range values are FATAL there, NaN here
MP_HD double qe_get_lcl_temp(const QeParams &p, double value) {
  if (value < p.val_min || value > p.val_max) return NAN;
  const int iv_floor = (int)floor((value - p.val_min) / p.val_inc) + 1;      // 1-based
  const double w_floor = (p.val_min + (iv_floor - 1) * p.val_inc);
  const double w_ceil = (value - w_floor) / p.val_inc;
  return p.lcl_temp_table[iv_floor] * w_ceil - p.lcl_temp_table[iv_floor - 1] * (w_ceil - 1);
}

// Work arrays of one column, 1-based.  The parcel's Tp, rp live in caller storage (QeParcel) and the relaxation deltas dT, dq take that
// storage over once the reference profiles have consumed the parcel (each level's Tp, rp are read before its dT, dq are written); the
// reference profiles themselves are only kept when the caller wants them (WANT_REF: the host tests; the device kernel does not).
// TVM: where the environment's virtual temperature lives -- 0: a thread-private array here; 1: caller storage (QeParcel::wTv); 2: nowhere: the
// parcel storage holds the environment's T and r of every level until the parcel's own values of that level are written, and each use of T_v(k)
// stands in front of that write (the LCL level's is carried in a register), so T_v is formed from them where it is read -- the same two operands, the
// same bits.  (The device kernel: the private array was 424 bytes of scratch per lane, written once and gathered from level by level.)
template <int LMAX, bool WANT_REF, int TVM = 0>
struct QeColumn {
  double Tv[TVM != 0 ? 1 : LMAX + 2], Tref[WANT_REF ? LMAX + 2 : 1], qref[WANT_REF ? LMAX + 2 : 1];
};
// The parcel's temperature and mixing ratio, written level by level in the ascent and read back by the reference profiles, live in
// caller storage: wTp[(k-1)*sw], wrp[(k-1)*sw] for level k = 1..L (LDS on the device; in thread-private arrays every store of the
// ascent is a memory operation that the loads of the next level then queue behind).
struct QeParcel {
  double *wTp, *wrp; int sw;
  double *wTv = nullptr;           // TV_EXT: the environment's virtual temperature in caller storage too (stride sw; the device kernel: a third LDS array
                                   // instead of a thread-private one, which was 424 bytes of scratch per lane)
  // Pure sigma levels inside the model's step (pk = 0: every pressure of a column is a constant of the vertical coordinate times p_s, as in
  // k_column_sig): the logarithms of pressure RATIOS between neighbouring levels that the parcel's path takes at every level are constants too,
  //   sig[k-1] = ln(p_full(k) / p_full(k+1)),  sig[L + k-1] = ln(p_half(k+1) / p_half(k)),  sig[2L + k-1] = ln(p_full(k) / p_full(L))   (k = 1..L),
  // built once in extended precision (moist.hip: moist_create).  Null: the ratios are divided and their logarithms taken per level and column
  // (caller pressures, hybrid levels, the host tests).  Two divisions and two logarithms fewer per level of the ascent: a third of its time.
  const double *sig = nullptr;
#ifdef MOIST_TIMING
  long long *marks = nullptr;      // timing builds (moist.hip): wall_clock64 stamps at the QE_MARK points of qe_moist_convection
#define QE_MARK(i) if (MOIST_TIMING == 5 && pc.marks) { const long long t_ = wall_clock64(); for (int i_ = i; i_ < 9; ++i_) pc.marks[i_] = t_; }
#else
#define QE_MARK(i)
#endif
  MP_HD double &Tp(int k) const { return wTp[(k - 1) * sw]; }
  MP_HD double &rp(int k) const { return wrp[(k - 1) * sw]; }
};

// deltaT / deltaq may BE the parcel storage (deltaT == pc.wTp, deltaq == pc.wrp, so == pc.sw): the deltas are then left where they are.
template <int LMAX, bool WANT_REF = true, class PH = const double *, int TVM = 0>
MP_HD void qe_moist_convection(const SatTable &st, const QeParams &P, int L, double dt, const double *Tin_, const double *qin_,
                               const double *p_full_, PH p_half_, int s, double *deltaT, double *deltaq, double &rain,
                               double &cape_out, double &cin_out, int &convflag, int &kLZB_out, int &kLCL_out, double *Tref_out,
                               double *qref_out, int so, const QeParcel &pc) {
  QeColumn<LMAX, WANT_REF, TVM> c;
  auto Tv_store = [&](int k, double tt, double r) {
    if constexpr (TVM == 1) pc.wTv[(k - 1) * pc.sw] = qe_virtual_temp(tt, r);
    else if constexpr (TVM == 0) c.Tv[k] = qe_virtual_temp(tt, r);
  };
  auto Tv = [&](int k) -> double {       // (TVM = 2: only for a level whose parcel values have not been written yet)
    if constexpr (TVM == 1) return pc.wTv[(k - 1) * pc.sw];
    else if constexpr (TVM == 0) return c.Tv[k];
    else return qe_virtual_temp(pc.Tp(k), pc.rp(k));
  };
  auto dT = [&](int k) -> double & { return pc.wTp[(k - 1) * pc.sw]; };      // valid from the reference-profile pass on
  auto dq = [&](int k) -> double & { return pc.wrp[(k - 1) * pc.sw]; };
  auto set_ref = [&](int k, double tref, double qref) { if (WANT_REF) { c.Tref[k] = tref; c.qref[k] = qref; } };
  auto Tin = [&](int k) { return Tin_[(k - 1) * s]; };
  auto qin = [&](int k) { return qin_[(k - 1) * s]; };
  auto rin = [&](int k) { const double q = qin_[(k - 1) * s]; return q / (1.0 - q); };
  auto pf = [&](int k) { return p_full_[(k - 1) * s]; };
  auto ph = [&](int k) { return ph_at(p_half_, s, k - 1); };
  const int ks = L;                                                       // k_surface
  // walk(k1, k2, body): body(k, p_half(k), p_half(k+1)) for k = k1..k2 while it returns true.  The half-level pressures of 8 levels are requested
  // together: k1 differs between the columns of a wavefront, so every request is a gather and a memory round trip -- one per chunk here, one per
  // level (or per four) in the plain loops the adjustment steps were until round 6 (up to 26 us per wavefront in the shallow branch's search).
  auto walk = [&](int k1, int k2, auto body) {
    constexpr int W = 8;
    for (int k0 = k1; k0 <= k2; k0 += W) {
      double pp[W + 1];
      MP_UNROLL_ALL
      for (int i = 0; i <= W; ++i) pp[i] = ph((k0 + i <= ks + 1) ? k0 + i : ks + 1);
      bool go = true;
      MP_UNROLL_ALL
      for (int i = 0; i < W; ++i)
        if (go && k0 + i <= k2) go = body(k0 + i, pp[i], pp[i + 1]);
      if (!go) return;
    }
  };
  auto set_nocape = [&](double &pLZB, int &kLZB, int &kLFC, double &CIN) {  // set_values_if_nocape (:1014-1030)
    pLZB = pf(1); kLZB = 0; kLFC = 0; CIN = 0.;
    MP_UNROLL
    for (int k = 1; k <= L; ++k) { pc.Tp(k) = Tin(k); pc.rp(k) = rin(k); }
  };
  auto to_model = [&](int k1, int k2) {                                   // set_profiles_to_full_model_values (:1034-1047)
    MP_UNROLL
    for (int k = k1; k <= k2; ++k) { set_ref(k, Tin(k), qin(k)); dT(k) = 0.; dq(k) = 0.; }
  };
  constexpr int IU = 20;                                // (first thing in the kernel, two arrays: the registers for 20 levels at once are free)
  for (int k0 = 1; k0 <= L; k0 += IU) {                 // chunks: the loads of IU levels are in flight together
    double tt[IU], qq[IU];
    MP_UNROLL_ALL
    for (int i = 0; i < IU; ++i) { const int k = (k0 + i <= L) ? k0 + i : L; tt[i] = Tin(k); qq[i] = qin(k); }
    MP_UNROLL_ALL
    for (int i = 0; i < IU; ++i) {
      const int k = k0 + i;
      if (k <= L) { const double r = qq[i] / (1.0 - qq[i]); pc.Tp(k) = tt[i]; pc.rp(k) = r; Tv_store(k, tt[i], r); }      // (deltaT, deltaq = 0: every exit below sets all levels)
    }
  }
  QE_MARK(1)
  // ---- CAPE_calculation (:383-446)
  bool nocape = true, saturated = false, skip = false;
  double CAPE = 0., CIN = 0., pLZB = 0., pLCL = 0.;
  int kLFC = 0, kLZB = 0, kLCL = 0;
  const double T0 = Tin(ks), r0 = rin(ks);
  const double rs = qe_mixing_ratio(lookup_es(st, T0), pf(ks));
  if (r0 >= rs) saturated = true;
  // ---- CAPE_below_LCL (:450-583)
  if (saturated) {
    pLCL = pf(ks); kLCL = ks;
    pc.Tp(ks) = T0 + (r0 - rs) / ((CP_AIR / (HLV + QE_SMALL)) + (HLV * rs) / RVGAS / (T0 * T0));
    pc.rp(ks) = qe_mixing_ratio(lookup_es(st, pc.Tp(ks)), pf(ks));
  } else {
    const double theta0 = Tin(ks) * pow(QE_PREF / pf(ks), KAPPA);
    double TLCL;
    if (r0 <= 0) {
      pLCL = pf(1);
      TLCL = theta0 * pow(pLCL / QE_PREF, KAPPA);
      skip = true;
    } else {
      const double value = log(pow(theta0, -1 / KAPPA) * QE_PREF * r0 / (RDGAS / RVGAS + r0));
      TLCL = qe_get_lcl_temp(P, value);
      pLCL = QE_PREF * pow(TLCL / theta0, 1. / KAPPA);
      if (pLCL < pf(1)) {
        pLCL = pf(1);
        TLCL = theta0 * pow(pLCL / QE_PREF, KAPPA);
      }
      int k = ks;
      double tv_lcl;                                    // T_v of the level the loop stops at: the LCL level's, read before the parcel's values of that level are written
      CIN = 0.;
      if (pc.sig) {      // (the table form: the parcel's dry adiabat from one logarithm per column and one exponential per level)
        const double *sg = pc.sig;
        const double lnpfs = log(pf(ks) / QE_PREF);
        double pfk = pf(ks), tvk = Tv(ks), lrel = sg[2 * L + ks - 1], lh = sg[L + ks - 1];        // next level requested one iteration ahead
        while (k >= 1 && pfk > pLCL) {
          const int kn = (k > 1) ? k - 1 : 1;
          const double pfn = pf(kn), tvn = Tv(kn), lreln = sg[2 * L + kn - 1], lhn = sg[L + kn - 1];
          const double Tpk = theta0 * exp(KAPPA * (lnpfs + lrel));
          pc.Tp(k) = Tpk;
          pc.rp(k) = qe_mixing_ratio(lookup_es(st, Tpk), pfk);
          CIN = CIN + RDGAS * (tvk - qe_virtual_temp(Tpk, r0)) * lh;
          k = k - 1;
          pfk = pfn; tvk = tvn; lrel = lreln; lh = lhn;
        }
        tv_lcl = tvk;
      } else {
        double pfk = pf(ks), ph1 = ph(ks + 1), phk = ph(ks), tvk = Tv(ks);        // next level requested one iteration ahead
        while (k >= 1 && pfk > pLCL) {
          const int kn = (k > 1) ? k - 1 : 1;
          const double pfn = pf(kn), phn = ph(kn), tvn = Tv(kn);
          const double Tpk = theta0 * pow(pfk / QE_PREF, KAPPA);
          pc.Tp(k) = Tpk;
          pc.rp(k) = qe_mixing_ratio(lookup_es(st, Tpk), pfk);
          CIN = CIN + RDGAS * (tvk - qe_virtual_temp(Tpk, r0)) * log(ph1 / phk);
          k = k - 1;
          pfk = pfn; ph1 = phk; phk = phn; tvk = tvn;
        }
        tv_lcl = tvk;
      }
      kLCL = k;
      if (kLCL >= 1) {
        double a = KAPPA * TLCL + (HLV / CP_AIR) * r0;
        double b = (HLV * HLV) * r0 / (CP_AIR * RVGAS * (TLCL * TLCL));
        double dtdlnp = a / (1.0 + b);
        pc.Tp(kLCL) = TLCL + dtdlnp * log(pf(kLCL) / pLCL) / 2;
        if ((pc.Tp(kLCL) < P.Tmin) && nocape) {
          skip = true;
          set_nocape(pLZB, kLZB, kLFC, CIN);
        } else {
          pc.rp(kLCL) = qe_mixing_ratio(lookup_es(st, pc.Tp(kLCL)), (pf(kLCL) + pLCL) / 2);
          a = KAPPA * pc.Tp(kLCL) + (HLV / CP_AIR) * pc.rp(kLCL);
          b = (HLV * HLV) * pc.rp(kLCL) / (CP_AIR * RVGAS * (pc.Tp(kLCL) * pc.Tp(kLCL)));
          dtdlnp = a / (1.0 + b);
          pc.Tp(kLCL) = TLCL + dtdlnp * log(pf(kLCL) / pLCL);
          if ((pc.Tp(kLCL) < P.Tmin) && nocape) {
            skip = true;
            set_nocape(pLZB, kLZB, kLFC, CIN);
          } else {
            pc.rp(kLCL) = qe_mixing_ratio(lookup_es(st, pc.Tp(kLCL)), pf(kLCL));
            const double tvp = qe_virtual_temp(pc.Tp(kLCL), pc.rp(kLCL));
            const double lh = pc.sig ? pc.sig[L + kLCL - 1] : log(ph(kLCL + 1) / ph(kLCL));
            if ((tvp < tv_lcl) && nocape) {
              CIN = CIN + RDGAS * (tv_lcl - tvp) * lh;
            } else {
              CAPE = CAPE + RDGAS * (tvp - tv_lcl) * lh;
              if (nocape) { nocape = false; kLFC = kLCL; }
            }
          }
        }
      }
    }
  }
  QE_MARK(2)
  // ---- CAPE_above_LCL (:587-668)
  if (skip) {
    if (nocape) set_nocape(pLZB, kLZB, kLFC, CIN);
  } else {
    if (kLCL - 1 >= 1) {
      // the parcel's previous level stays in registers; pressures and Tv of the next level are requested one iteration ahead, so their
      // latency hides behind this level's table-lookup chain.  TAB: the two logarithms of the level come from pc.sig (requested ahead as well)
      auto ascent = [&](auto tab) {
        constexpr bool TAB = decltype(tab)::value;
        [[maybe_unused]] const double *sg = pc.sig;
        double Tp1 = pc.Tp(kLCL), rp1 = pc.rp(kLCL);
        double pf1 = pf(kLCL), pfk = pf(kLCL - 1), tvk = Tv(kLCL - 1);
        [[maybe_unused]] double ph1 = 0., phk = 0., lf = 0., lh = 0.;
        if constexpr (TAB) { lf = sg[kLCL - 2]; lh = sg[L + kLCL - 2]; }
        else { ph1 = ph(kLCL); phk = ph(kLCL - 1); }
        for (int k = kLCL - 1; k >= 1; --k) {
          const int kn = (k > 1) ? k - 1 : 1;
          const double pfn = pf(kn), tvn = Tv(kn);
          [[maybe_unused]] double phn = 0., lfn = 0., lhn = 0.;
          if constexpr (TAB) { lfn = sg[kn - 1]; lhn = sg[L + kn - 1]; }
          else { phn = ph(kn); lf = log(pfk / pf1); }
          double a = KAPPA * Tp1 + (HLV / CP_AIR) * rp1;
          double b = (HLV * HLV) * rp1 / (CP_AIR * RVGAS * (Tp1 * Tp1));
          double dtdlnp = a / (1.0 + b);
          double Tpk = Tp1 + dtdlnp * lf / 2;
          pc.Tp(k) = Tpk;
          if ((Tpk < P.Tmin) && nocape) { set_nocape(pLZB, kLZB, kLFC, CIN); break; }
          double rpk = qe_mixing_ratio(lookup_es(st, Tpk), (pfk + pf1) / 2);
          a = KAPPA * Tpk + (HLV / CP_AIR) * rpk;
          b = (HLV * HLV) * rpk / (CP_AIR * RVGAS * (Tpk * Tpk));
          dtdlnp = a / (1.0 + b);
          Tpk = Tp1 + dtdlnp * lf;
          pc.Tp(k) = Tpk;
          if ((Tpk < P.Tmin) && nocape) { pc.rp(k) = rpk; set_nocape(pLZB, kLZB, kLFC, CIN); break; }
          rpk = qe_mixing_ratio(lookup_es(st, Tpk), pfk);
          pc.rp(k) = rpk;
          const double tvp = qe_virtual_temp(Tpk, rpk);
          if ((tvp < tvk) && !nocape) {
            kLZB = k + 1;
            break;
          }
          if constexpr (!TAB) lh = log(ph1 / phk);
          if ((tvp < tvk) && nocape) {
            CIN = CIN + RDGAS * (tvk - tvp) * lh;
          } else {
            CAPE = CAPE + RDGAS * (tvp - tvk) * lh;
            if (nocape) { nocape = false; kLFC = k; }
          }
          Tp1 = Tpk; rp1 = rpk; pf1 = pfk; pfk = pfn; tvk = tvn;
          if constexpr (TAB) { lf = lfn; lh = lhn; }
          else { ph1 = phk; phk = phn; }
        }
      };
      if (pc.sig) ascent(std::true_type{}); else ascent(std::false_type{});
    }
  }
  QE_MARK(3)
  cape_out = CAPE; cin_out = CIN; kLZB_out = kLZB; kLCL_out = kLCL;
  (void)kLFC; (void)pLZB;
  convflag = 0;
  double Pq = 0.;
  if (CAPE > 0) {
    convflag = 1;
    // ---- set_reference_profiles (:768-796)
    double Pt = 0.;
    const int kb = (kLZB < 1) ? 1 : kLZB, kmodel = (kLZB - 1 > 1) ? kLZB - 1 : 1;
    // set_reference_profiles (:768-796), set_profiles_to_full_model_values above the LZB, Pq_calculation (:710-733) and
    // Pt_calculation (:737-764) act on each level independently: one pass in chunks, the four steps of a level in the
    // reference's order, the two sums in theirs
    for (int k0 = 1; k0 <= L; k0 += MP_U) {
      double tp[MP_U], rp[MP_U], pfv[MP_U], qi[MP_U], ti[MP_U], ph0[MP_U], ph1[MP_U];
      MP_UNROLL_ALL
      for (int i = 0; i < MP_U; ++i) {
        const int k = (k0 + i <= L) ? k0 + i : L;
        tp[i] = pc.Tp(k); rp[i] = pc.rp(k); pfv[i] = pf(k); qi[i] = qin(k); ti[i] = Tin(k); ph0[i] = ph(k); ph1[i] = ph(k + 1);
      }
      MP_UNROLL_ALL
      for (int i = 0; i < MP_U; ++i) {
        const int k = k0 + i;
        if (k <= L) {
          double tref = tp[i], qref = 0.0;
          if (k >= kb) {
            const double eref = P.rhbm * pfv[i] * rp[i] / (rp[i] + (RDGAS / RVGAS));
            const double r = qe_mixing_ratio(eref, pfv[i]);       // (the reference also stores it as the parcel's rp, which nothing reads again)
            qref = r / (1 + r);
          }
          if (k <= kmodel) { tref = ti[i]; qref = qi[i]; dT(k) = 0.; dq(k) = 0.; }
          set_ref(k, tref, qref);
          if (k >= kb) {
            const double dqk = -(qi[i] - qref) * dt / P.tau_bm;
            dq(k) = dqk;
            Pq = Pq + dqk * (ph0[i] - ph1[i]);
            const double dTk = -(ti[i] - tref) * dt / P.tau_bm;
            dT(k) = dTk;
            Pt = Pt + (CP_AIR / (HLV + QE_SMALL)) * dTk * (ph1[i] - ph0[i]);
          }
        }
      }
    }
    Pq = Pq / GRAV;
    Pt = Pt / GRAV;
    QE_MARK(4)
    if ((Pq > 0) && (Pt > 0)) {
      convflag = 2;
      if (Pq > Pt) {                                  // do_change_time_scale_deepconv (:992-1008)
        const double invtau_q = Pt / Pq / P.tau_bm;
        MP_UNROLL
        for (int k = kLZB; k <= ks; ++k) dq(k) = P.tau_bm * invtau_q * dq(k);
        Pq = Pt;
      } else {                                        // do_change_Tref_deepconv (:957-988)
        double deltak = 0.;
        walk(kLZB, ks, [&](int k, double p0, double p1) { deltak = deltak - (dT(k) + (HLV / CP_AIR) * dq(k)) * (p1 - p0); return true; });
        deltak = deltak / (ph(ks + 1) - ph(kLZB));
        MP_UNROLL
        for (int k = kLZB; k <= ks; ++k) { if (WANT_REF) c.Tref[k] = c.Tref[k] + deltak * P.tau_bm / dt; dT(k) = dT(k) + deltak; }
      }
    } else if (Pt > 0) {
      // ---- do_shallow_convection (:800-840) with level_of_zero_precip (:844-888)
      int k = kLZB;
      bool found = false;
      walk(kLZB, ks, [&](int kk, double p0, double p1) {       // while ((Pq < 0.) && (k <= ks))
        if (!(Pq < 0.)) return false;
        Pq = Pq - dq(kk) * (p0 - p1) / GRAV;
        k = kk + 1;
        return true;
      });
      const int k_top = k - 1;
      if (Pq > 0.) found = true;
      if (k_top > kLZB) to_model(kLZB, k_top - 1);
      if (found) {                                    // change_Tref_LZB_shallowconv (:893-930)
        const double cc = Pq * GRAV / (dq(k_top) * (ph(k_top + 1) - ph(k_top)));
        dq(k_top) = dq(k_top) * cc;
        dT(k_top) = dT(k_top) * cc;
        double deltak = 0.;
        walk(k_top, ks, [&](int kk, double p0, double p1) { deltak = deltak + dT(kk) * (p0 - p1); return true; });
        deltak = deltak / (ph(ks + 1) - ph(k_top));
        if (k_top != ks)
          for (int kk = k_top; kk <= ks; ++kk) { dT(kk) = dT(kk) + deltak; if (WANT_REF) c.Tref[kk] = c.Tref[kk] + deltak * P.tau_bm / dt; }
      } else {
        if (k_top == kLZB) to_model(ks, ks); else to_model(kLZB, k_top);
      }
      Pq = 0.;
    } else {
      Pq = 0.;
      to_model(1, ks);
    }
  } else {
    Pq = 0.;
    to_model(1, ks);
  }
  rain = Pq;
  QE_MARK(5)
  if (deltaT != pc.wTp) {
    MP_UNROLL_ALL
    for (int k = 1; k <= LMAX; ++k) {          // fixed trip count: the reads of the work arrays are issued together
      if (k > L) break;
      deltaT[(k - 1) * so] = dT(k); deltaq[(k - 1) * so] = dq(k);
    }
  }
  if (WANT_REF) {
    for (int k = 1; k <= L; ++k) {
      if (Tref_out) Tref_out[(k - 1) * so] = c.Tref[k];
      if (qref_out) qref_out[(k - 1) * so] = c.qref[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Monin-Obukhov surface layer (atmos_param/monin_obukhov/monin_obukhov_kernel.F90:122-420, :423-520, :648-810) with the
// module defaults rich_crit = 2, drag_min = 1e-5, stable_option = 1, neutral = .false.  The reference iterates a row of
// points together but freezes each point once it has converged, which is this per-point loop.
// ------------------------------------------------------------------------------------------------
struct MoParams { double rich_crit = 2.0, drag_min = 1.e-05, zeta_trans = 0.5; };
MP_HD double mo_phi_t(const MoParams &p, double zeta) {
  if (zeta < 0.0) return 1.0 / sqrt(1 - 16.0 * zeta);      // (1 - 16 zeta)**(-0.5): within an ulp of pow, a quarter of its time on the device
  const double b_stab = 1.0 / p.rich_crit;
  return 1.0 + zeta * (5.0 + b_stab * zeta) / (1.0 + zeta);
}
MP_HD double mo_phi_m(const MoParams &p, double zeta) {
  if (zeta < 0.0) { const double x = 1.0 / sqrt(1 - 16.0 * zeta); return sqrt(x); }
  const double b_stab = 1.0 / p.rich_crit;
  return 1.0 + zeta * (5.0 + b_stab * zeta) / (1.0 + zeta);
}
MP_HD double mo_psi_m(const MoParams &p, double zeta, double zeta_0, double ln_z_z0) {
  const double b_stab = 1.0 / p.rich_crit;
  if (zeta < 0.0) {
    double x = sqrt(1 - 16.0 * zeta), x_0 = sqrt(1 - 16.0 * zeta_0);
    x = sqrt(x); x_0 = sqrt(x_0);
    const double x1 = 1.0 + x, x1_0 = 1.0 + x_0;
    const double num = x1 * x1 * (1.0 + x * x), denom = x1_0 * x1_0 * (1.0 + x_0 * x_0);
    const double y = atan(x) - atan(x_0);
    return ln_z_z0 - log(num / denom) + 2 * y;
  }
  return ln_z_z0 + (5.0 - b_stab) * log((1.0 + zeta) / (1.0 + zeta_0)) + b_stab * (zeta - zeta_0);
}
MP_HD double mo_psi_t(const MoParams &p, double zeta, double zeta_t, double ln_z_zt) {
  const double b_stab = 1.0 / p.rich_crit;
  if (zeta < 0.0) {
    const double x = sqrt(1 - 16.0 * zeta), x_t = sqrt(1 - 16.0 * zeta_t);
    return ln_z_zt - 2.0 * log((1.0 + x) / (1.0 + x_t));
  }
  return ln_z_zt + (5.0 - b_stab) * log((1.0 + zeta) / (1.0 + zeta_t)) + b_stab * (zeta - zeta_t);
}
// monin_obukhov_solve_zeta (:245-420) for one point
MP_HD void mo_solve_zeta(const MoParams &p, double rich, double z, double z0, double zt, double zq, double &f_m, double &f_t, double &f_q) {
  const double error = 1.e-04, zeta_min = 1.e-06;
  const int max_iter = 20;
  const double z_z0 = z / z0, z_zt = z / zt, z_zq = z / zq;
  const double ln_z_z0 = log(z_z0), ln_z_zt = log(z_zt), ln_z_zq = log(z_zq);
  double zeta = rich * ln_z_z0 * ln_z_z0 / ln_z_zt;
  if (rich >= 0.0) zeta = zeta / (1.0 - rich / p.rich_crit);
  f_m = f_t = f_q = 0.0;
  for (int iter = 1; iter <= max_iter; ++iter) {
    if (fabs(zeta) < zeta_min) { f_m = ln_z_z0; f_t = ln_z_zt; f_q = ln_z_zq; return; }
    const double rzeta = 1.0 / zeta, zeta_0 = zeta / z_z0, zeta_t = zeta / z_zt, zeta_q = zeta / z_zq;
    const double phi_m = mo_phi_m(p, zeta), phi_m_0 = mo_phi_m(p, zeta_0), phi_t = mo_phi_t(p, zeta), phi_t_0 = mo_phi_t(p, zeta_t);
    f_m = mo_psi_m(p, zeta, zeta_0, ln_z_z0);
    f_t = mo_psi_t(p, zeta, zeta_t, ln_z_zt);
    f_q = mo_psi_t(p, zeta, zeta_q, ln_z_zq);
    const double df_m = (phi_m - phi_m_0) * rzeta, df_t = (phi_t - phi_t_0) * rzeta;
    const double rich_1 = zeta * f_t / (f_m * f_m);
    const double d_rich = rich_1 * (rzeta + df_t / f_t - 2.0 * df_m / f_m);
    const double correction = (rich - rich_1) / d_rich;
    const double corr = fmin(fabs(correction), fabs(correction / zeta));
    if (corr > error) zeta = zeta + correction; else return;
  }
}
// monin_obukhov_drag_1d (:122-241)
MP_HD void mo_drag(const MoParams &p, double pt, double pt0, double z, double z0, double zt, double zq, double speed, double &drag_m,
                   double &drag_t, double &drag_q, double &u_star, double &b_star) {
  const double small = 1.e-04;
  const double r_crit = 0.95 * p.rich_crit;
  const double sqrt_drag_min = (p.drag_min != 0.0) ? sqrt(p.drag_min) : 0.0;
  const double delta_b = GRAV * (pt0 - pt) / pt0;
  const double rich = -z * delta_b / (speed * speed + small);
  const double zz = fmax(fmax(z, z0), fmax(zt, zq));
  if (rich >= r_crit) {
    drag_m = drag_t = drag_q = p.drag_min;
    u_star = sqrt_drag_min * speed; b_star = sqrt_drag_min * delta_b;
    return;
  }
  double fm, ft, fq;
  mo_solve_zeta(p, rich, zz, z0, zt, zq, fm, ft, fq);
  const double us = fmax(VONKARM / fm, sqrt_drag_min), bs = fmax(VONKARM / ft, sqrt_drag_min), qs = fmax(VONKARM / fq, sqrt_drag_min);
  drag_m = us * us; drag_t = us * bs; drag_q = us * qs;
  u_star = us * speed; b_star = bs * delta_b;
}

// ------------------------------------------------------------------------------------------------
// surface_flux (coupler/surface_flux.F90:290-700) over an ocean point with do_simple = .true., use_virtual_temp = .false.,
// old_dtaudv = .true., no bucket, gust_min = 0, surface at rest.
// ------------------------------------------------------------------------------------------------
struct SurfFlux {
  double flux_t, flux_q, flux_r, flux_u, flux_v, dhdt_surf, dedt_surf, dedq_surf, drdt_surf, dhdt_atm, dedq_atm, dtaudu_atm, dtaudv_atm,
         w_atm, u_star, b_star, q_star, cd_m, cd_t, cd_q, q_surf;
};
MP_HD void surface_flux(const SatTable &st, const MoParams &mo, double t_atm, double q_atm, double u_atm, double v_atm, double p_atm,
                        double z_atm, double p_surf, double t_surf, double rough_mom, double rough_heat, double rough_moist,
                        double rough_scale, double gust, SurfFlux &o) {
  const double del_temp = 0.1, del_temp_inv = 1.0 / del_temp;
  const double d622 = RDGAS / RVGAS, kappa = RDGAS / CP_AIR, d608 = 0.0;   // use_virtual_temp = .false.
  const double t_surf0 = t_surf, t_surf1 = t_surf0 + del_temp;
  const double e_sat = lookup_es(st, t_surf0), e_sat1 = lookup_es(st, t_surf1);
  const double q_sat = d622 * e_sat / p_surf, q_sat1 = d622 * e_sat1 / p_surf;     // do_simple
  const double q_surf0 = q_sat;
  const double p_ratio = pow(p_surf / p_atm, kappa);
  const double tv_atm = t_atm * (1.0 + d608 * q_atm);
  const double th_atm = t_atm * p_ratio, thv_atm = tv_atm * p_ratio, thv_surf = t_surf0 * (1.0 + d608 * q_surf0);
  const double u_dif = 0.0 - u_atm, v_dif = 0.0 - v_atm;
  const double w_gust = gust;
  const double w_atm = sqrt(u_dif * u_dif + v_dif * v_dif + w_gust * w_gust);
  double cd_m, cd_t, cd_q, u_star, b_star;
  mo_drag(mo, thv_atm, thv_surf, z_atm, rough_mom, rough_heat, rough_moist, w_atm, cd_m, cd_t, cd_q, u_star, b_star);
  const double lr = log(z_atm / rough_mom + 1) / log(z_atm / rough_scale + 1);
  cd_m = cd_m * (lr * lr);
  const double drag_t = cd_t * w_atm, drag_q = cd_q * w_atm, drag_m = cd_m * w_atm;
  const double rho = p_atm / (RDGAS * tv_atm);
  double rho_drag = CP_AIR * drag_t * rho;
  o.flux_t = rho_drag * (t_surf0 - th_atm);
  o.dhdt_surf = rho_drag;
  o.dhdt_atm = -rho_drag * p_ratio;
  rho_drag = drag_q * rho;
  o.flux_q = rho_drag * (q_surf0 - q_atm);
  o.dedq_surf = 0;
  o.dedt_surf = rho_drag * (q_sat1 - q_sat) * del_temp_inv;
  o.dedq_atm = -rho_drag;
  o.q_star = o.flux_q / (u_star * rho);
  o.q_surf = q_atm + o.flux_q / (rho * cd_q * w_atm);
  o.flux_r = STEFAN * pow4(t_surf);
  o.drdt_surf = 4 * STEFAN * (t_surf * t_surf * t_surf);
  rho_drag = drag_m * rho;
  o.flux_u = rho_drag * u_dif;
  o.flux_v = rho_drag * v_dif;
  o.dtaudv_atm = -rho_drag;            // old_dtaudv
  o.dtaudu_atm = -rho_drag;
  o.w_atm = w_atm; o.u_star = u_star; o.b_star = b_star; o.cd_m = cd_m; o.cd_t = cd_t; o.cd_q = cd_q;
}


// ------------------------------------------------------------------------------------------------
// Rayleigh sponge of damping_driver (atmos_param/damping_driver/damping_driver.f90:164-171, :594-637): levels
// 1..nlev_rayfric above sponge_pbottom are damped towards rest with rate rfactr ((p_b - p)/p_b)^2, the kinetic energy
// removed goes into heat (do_conserve_energy).  nlev_rayfric and rfactr are set up by the host (:411-420).
// ------------------------------------------------------------------------------------------------
struct RayleighParams { int nlev_rayfric = 0; double rfactr = 0.0, sponge_pbottom = 50.0; bool conserve_energy = true; };
// zero_in: the incoming tendencies of the sponge levels are zero and are not read (0 + x is still formed)
MP_HD void rayleigh_damping(const RayleighParams &p, double dt, const double *pfull, const double *u, const double *v, int s, double *udt,
                            double *vdt, int st, double *tdt, int stt, bool zero_in = false) {
  for (int k0 = 0; k0 < p.nlev_rayfric; k0 += MP_U) {
    double pf[MP_U], uk[MP_U], vk[MP_U], ud[MP_U], vd[MP_U], td[MP_U];
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = (k0 + i < p.nlev_rayfric) ? k0 + i : p.nlev_rayfric - 1;
      pf[i] = pfull[k * s]; uk[i] = u[k * s]; vk[i] = v[k * s];
      ud[i] = zero_in ? 0.0 : udt[k * st]; vd[i] = zero_in ? 0.0 : vdt[k * st]; td[i] = zero_in ? 0.0 : tdt[k * stt];
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = k0 + i;
      if (k < p.nlev_rayfric) {
        double ut = 0.0, vt = 0.0;
        if (pf[i] < p.sponge_pbottom) {
          const double d = p.sponge_pbottom - pf[i];
          const double fact = p.rfactr * (d * d) / (p.sponge_pbottom * p.sponge_pbottom);
          ut = -uk[i] * fact; vt = -vk[i] * fact;
        }
        udt[k * st] = ud[i] + ut;
        vdt[k * st] = vd[i] + vt;
        if (p.conserve_energy) tdt[k * stt] = td[i] + (-((uk[i] + .5 * dt * ut) * ut + (vk[i] + .5 * dt * vt) * vt) / CP_AIR);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Boundary-layer diffusivities: vert_turb_driver with do_diffusivity (atmos_param/vert_turb_driver/vert_turb_driver.F90:204-273)
// -> diffusivity with do_simple = .true., do_entrain = .false. (atmos_param/diffusivity/diffusivity.F90:283-360): boundary
// layer depth from the bulk Richardson number of the dry static energy profile (pbl_depth :364-444), Monin-Obukhov
// diffusivities in the inner layer, the Troen-Mahrt shape above (diffusivity_pbl :448-510; mo_diff =
// monin_obukhov_kernel.F90:42-120).  Inputs are the provisional fields x_prev + dt * dx/dt (use_tau = .false.);
// k_m[k], k_t[k] sit on the interface above full level k (k = 0 stays zero).
// ------------------------------------------------------------------------------------------------
struct DiffusivityParams { double frac_inner = 0.1, rich_crit_pbl = 1.0, small = 1.e-04, ustar_min = 1.e-10; };
MP_HD double mo_diff_m(const MoParams &mo, const DiffusivityParams &dp, double z, double u_star, double b_star) {
  const double uss = fmax(u_star, dp.ustar_min);
  const double zeta = -(VONKARM * b_star * z / (uss * uss));
  return VONKARM * uss * z / mo_phi_m(mo, zeta);
}
MP_HD double mo_diff_t(const MoParams &mo, const DiffusivityParams &dp, double z, double u_star, double b_star) {
  const double uss = fmax(u_star, dp.ustar_min);
  const double zeta = -(VONKARM * b_star * z / (uss * uss));
  return VONKARM * uss * z / mo_phi_t(mo, zeta);
}
// pbl_depth (:364-444), do_simple: the height where the bulk Richardson number of the provisional profile first exceeds
// rich_crit_pbl, searched upward from the lowest level.  Nothing is stored: each level is visited once.
// tdt(k), udt(k), vdt(k): the tendencies so far as functions of the level (memory reads inside them must be unconditional)
// kstop (optional): the level at which the search stopped (L - 1 if it never did).  The depth returned lies below that level's height, so every
// interface at or above it has no diffusion (diffusivity_pbl is zero from the depth upward): levels 0 .. kstop-1 are untouched by the implicit diffusion.
// pbl_depth_f2: two sets of tendency functions -- (tdt, udt, vdt) valid at every level and (tdt_lo, udt_lo, vdt_lo) valid at the levels k >= klim only
// (the device kernel's: below the sponge its tendency functions need no sponge terms, and a function that tests the level for them costs the chunk
// its single memory round trip: the compiler turns the test into a branch and sinks the load behind it); a chunk that lies at or below klim takes the
// second set.
template <class TDT, class UDT, class VDT, class TDT2, class UDT2, class VDT2>
MP_HD double pbl_depth_f2(const DiffusivityParams &dp, int L, double dt, const double *tm, const double *um, const double *vm, int s,
                          TDT tdt, UDT udt, VDT vdt, TDT2 tdt_lo, UDT2 udt_lo, VDT2 vdt_lo, int klim, const double *z_full, const double *z_half, int sz,
                          int *kstop = nullptr) {
  const double gcp = GRAV / CP_AIR;
  if (kstop) *kstop = L - 1;
  const double z_surf = z_half[L * sz];
  double tbot = 0.0, h1 = 0.0, rich1 = 0.0, h = 0.0, result = 0.0;
  bool found = false;
  auto chunk = [&](int k0, auto &ft, auto &fu, auto &fv) {
    double zf[MP_U], tt[MP_U], uu[MP_U], vv[MP_U], ta[MP_U], ua[MP_U], va[MP_U];
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = (k0 - i >= 0) ? k0 - i : 0;
      zf[i] = z_full[k * sz]; tt[i] = tm[k * s]; uu[i] = um[k * s]; vv[i] = vm[k * s];
      ta[i] = ft(k); ua[i] = fu(k); va[i] = fv(k);
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) { tt[i] = tt[i] + dt * ta[i]; uu[i] = uu[i] + dt * ua[i]; vv[i] = vv[i] + dt * va[i]; }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = k0 - i;
      if (k < 0 || found) break;
      const double zfa = zf[i] - z_surf;
      const double svcp = tt[i] + gcp * zfa;
      if (k == L - 1) tbot = svcp;
      const double rich2 = zfa * GRAV * (svcp - tbot) / tbot / (uu[i] * uu[i] + vv[i] * vv[i] + dp.small);
      if (k == L - 1) { h1 = zfa; h = h1; rich1 = rich2; continue; }
      if (rich2 > dp.rich_crit_pbl) {
        if (kstop) *kstop = k;
        result = zfa + (h1 - zfa) * (rich2 - dp.rich_crit_pbl) / (rich2 - rich1);
        found = true;
        break;
      }
      rich1 = rich2; h1 = zfa;
    }
  };
  for (int k0 = L - 1; k0 >= 0 && !found; k0 -= MP_U) {
    if (k0 - (MP_U - 1) >= klim) chunk(k0, tdt_lo, udt_lo, vdt_lo);
    else chunk(k0, tdt, udt, vdt);
  }
  return found ? result : h;
}
template <class TDT, class UDT, class VDT>
MP_HD double pbl_depth_f(const DiffusivityParams &dp, int L, double dt, const double *tm, const double *um, const double *vm, int s,
                         TDT tdt, UDT udt, VDT vdt, const double *z_full, const double *z_half, int sz, int *kstop = nullptr) {
  return pbl_depth_f2(dp, L, dt, tm, um, vm, s, tdt, udt, vdt, tdt, udt, vdt, L + 1, z_full, z_half, sz, kstop);
}
MP_HD double pbl_depth(const DiffusivityParams &dp, int L, double dt, const double *tm, const double *um, const double *vm, int s,
                       const double *tdt, const double *udt, const double *vdt, int st, const double *z_full, const double *z_half, int sz) {
  return pbl_depth_f(dp, L, dt, tm, um, vm, s, [&](int k) { return tdt[k * st]; }, [&](int k) { return udt[k * st]; },
                     [&](int k) { return vdt[k * st]; }, z_full, z_half, sz);
}
// diffusivity_pbl (:448-510) as a function of the interface: k_m, k_t on the interface above full level k (k >= 1; 0 at k = 0)
struct PblProfile {
  MoParams mo; DiffusivityParams dp;
  double h, h_inner, k_m_ref, k_t_ref, u_star, b_star, z_surf;
  const double *z_half; int sz;
  MP_HD void init(const MoParams &mo_, const DiffusivityParams &dp_, double h_, double u_star_, double b_star_, const double *z_half_, int sz_,
                  int L) {
    mo = mo_; dp = dp_; h = h_; u_star = u_star_; b_star = b_star_; z_half = z_half_; sz = sz_;
    z_surf = z_half[L * sz];
    h_inner = dp.frac_inner * h;
    k_m_ref = mo_diff_m(mo, dp, h_inner, u_star, b_star);
    k_t_ref = mo_diff_t(mo, dp, h_inner, u_star, b_star);
  }
  MP_HD double shape(double zm) const {
    const double r = 1.0 - (zm - h_inner) / (h - h_inner);
    return (zm / h_inner) * (r * r);
  }
  MP_HD double k_m_at(int k, double z_half_k) const {
    if (k == 0) return 0.0;
    const double zm = z_half_k - z_surf;
    if (zm < h_inner) return mo_diff_m(mo, dp, zm, u_star, b_star);
    if (zm < h) return k_m_ref * shape(zm);
    return 0.0;
  }
  MP_HD double k_t_at(int k, double z_half_k) const {
    if (k == 0) return 0.0;
    const double zm = z_half_k - z_surf;
    if (zm < h_inner) return mo_diff_t(mo, dp, zm, u_star, b_star);
    if (zm < h) return k_t_ref * shape(zm);
    return 0.0;
  }
  MP_HD double k_m(int k) const { return k_m_at(k, z_half[k * sz]); }
  MP_HD double k_t(int k) const { return k_t_at(k, z_half[k * sz]); }
  struct Km { const PblProfile &p; MP_HD double raw(int k) const { return p.z_half[k * p.sz]; } MP_HD double eval(int k, double z) const { return p.k_m_at(k, z); } };
  struct Kt { const PblProfile &p; MP_HD double raw(int k) const { return p.z_half[k * p.sz]; } MP_HD double eval(int k, double z) const { return p.k_t_at(k, z); } };
};

// ------------------------------------------------------------------------------------------------
// Implicit vertical diffusion (atmos_param/vert_diff/vert_diff.F90): downward sweep of the tridiagonal elimination for
// momentum (closed at the surface with the stress and its derivative, kinetic energy dissipated into heat) and for
// dry static energy / humidity (left open: Tri_surf hands the lowest-level increments to the surface model),
// gcm_vert_diff_down :270-406; the mixed-layer ocean closes the system (mixed_layer.F90:568-720) and
// gcm_vert_diff_up :410-467 back-substitutes.
// The reference builds mu, nu, the explicit tendencies, a/b/c, e/g and f in separate array passes (compute_mu :1033,
// compute_nu :1053, explicit_tend :1005, compute_e :951, compute_f :984, vert_diff_down_2 :814); here one downward loop per
// pair of fields evaluates the same expressions level by level with the neighbours carried in registers, and only e, f_1,
// f_2 (needed again by the upward sweep) are stored, in caller storage (w[k * sw]: LDS on the device).
// ------------------------------------------------------------------------------------------------
struct VdiffSurf { double dtmass, dflux_t, delta_t, dflux_q, delta_q, delta_u, delta_v; };
struct VdiffWork { double *e, *f1, *f2; int sw, sw2; };      // strides of e, f1 / of f2 (which may live elsewhere)

namespace vd {
// A diffusivity profile is handed to the sweeps in two steps, raw(k) = whatever has to come from memory for interface k and
// eval(k, raw) = the diffusivity from it, so that a chunk's loads are issued together and the (branchy) evaluation follows.
struct TableDiff {            // diffusivities already tabulated, tab[k * s]
  const double *tab; int s;
  MP_HD double raw(int k) const { return tab[k * s]; }
  MP_HD double eval(int, double v) const { return v; }
};
// diff_surface (:882-910)
MP_HD void diff_surface(double mu_delt, double nu, double e_n1, double f_delt_n1, double dflux_datmos, double &flux, double factor,
                        double &delta_xi) {
  const double fff = 1.0 / factor;
  const double dflux = -nu * (1.0 - e_n1);
  delta_xi = delta_xi + mu_delt * nu * f_delt_n1;
  delta_xi = (delta_xi + mu_delt * flux * fff) / (1.0 - mu_delt * (dflux + dflux_datmos * fff));
  flux = flux + dflux_datmos * delta_xi;
}
struct DownResult { double mu_delt_n, nu_n, e_n1, f1_delt_n1, f2_delt_n1, delta_1_n, delta_2_n; };
// vert_diff_down_2 for the pair (x1, x2) with tendencies (d1, d2), diffusivity diff(k) on the interface above level k.
// aux: a value per level the caller wants read in the load phase of the level's chunk and handed back in its store phase (NoAux: nothing)
struct NoAux {
  MP_HD double load(int) const { return 0.0; }
  MP_HD void store(int, double) const {}
};
// kb > 0: the caller knows that no interface at or above level kb carries diffusion (pbl_depth_f's kstop); the sweep starts at level kb (nothing is
// stored for the levels above, whose e = 0, f = the incoming tendency: vert_diff_passthrough) -- the values it produces are those of the whole sweep,
// except that a ZERO may come out with the other sign.
template <class X1, class X2, class D1, class D2, class DIFF, class PH, class AUX = NoAux>
MP_HD DownResult down_pair(int L, double delt, X1 x1, X2 x2, D1 d1, D2 d2, DIFF diff, const double *t, int s, PH p_half,
                           const double *z_full, int sp, const VdiffWork &w, AUX aux = AUX(), int kb = 0) {
  DownResult r;
  double fl1_k = 0.0, fl2_k = 0.0, nu_k = 0.0, e_prev = 0.0, f1_prev = 0.0, f2_prev = 0.0;
  double x1_k = x1(kb), x2_k = x2(kb), t_k = t[kb * s], z_k = z_full[kb * sp], ph_k = ph_at(p_half, sp, kb);
  for (int k0 = kb; k0 < L; k0 += MP_U) {
    double phn[MP_U], tn[MP_U], zn[MP_U], x1n[MP_U], x2n[MP_U], dd1[MP_U], dd2[MP_U], df[MP_U], zr[MP_U], ax[MP_U];
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {       // everything level k0+i needs from memory: its own tendencies, the fields of the level below
      const int k = (k0 + i < L) ? k0 + i : L - 1, kn = (k + 1 < L) ? k + 1 : L - 1;
      phn[i] = ph_at(p_half, sp, k + 1); tn[i] = t[kn * s]; zn[i] = z_full[kn * sp]; x1n[i] = x1(kn); x2n[i] = x2(kn);
      dd1[i] = d1(k); dd2[i] = d2(k); zr[i] = diff.raw(kn); ax[i] = aux.load(k);
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {       // the diffusivities from what was loaded (branches, but no memory access behind them)
      const int k = (k0 + i < L) ? k0 + i : L - 1, kn = (k + 1 < L) ? k + 1 : L - 1;
      df[i] = diff.eval(kn, zr[i]);
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = k0 + i;
      if (k < L) {
        const double ph_n = phn[i];
        double nu_n = 0.0, e1, e2, fl1_n = 0.0, fl2_n = 0.0;
        if (k < L - 1 && df[i] == 0.0 && nu_k == 0.0) {
          // No diffusion across either interface of this level -- every level above the boundary layer, three quarters of the column: the
          // general path below then computes nu = 0, a = c = -0, b = g = 1, e = 0 and f = the incoming tendencies with four divisions per level.
          // The same values without them (a zero tendency may come out as +0 where the general path leaves -0).
          e_prev = 0.0; f1_prev = dd1[i] + 0.0; f2_prev = dd2[i] + 0.0;
          w.e[k * w.sw] = e_prev; w.f1[k * w.sw] = f1_prev; w.f2[k * w.sw2] = f2_prev;
          fl1_k = 0.0; fl2_k = 0.0; nu_k = 0.0; x1_k = x1n[i]; x2_k = x2n[i]; t_k = tn[i]; z_k = zn[i]; ph_k = ph_n;
          continue;
        }
        const double mu = GRAV / (ph_n - ph_k);                       // compute_mu
        if (k < L - 1) {
          const double rho_half = 2.0 * ph_n / (RDGAS * (tn[i] + t_k));   // compute_nu, no virtual temperature
          nu_n = rho_half * df[i] / (z_k - zn[i]);
          fl1_n = nu_n * (x1n[i] - x1_k); fl2_n = nu_n * (x2n[i] - x2_k);             // explicit_tend
          e1 = dd1[i] + mu * (fl1_n - fl1_k); e2 = dd2[i] + mu * (fl2_n - fl2_k);
        } else {
          e1 = dd1[i] - mu * fl1_k; e2 = dd2[i] - mu * fl2_k;
        }
        const double a = (k < L - 1) ? -(mu * nu_n * delt) : 0.0;                      // compute_e
        const double c = (k > 0) ? -(mu * nu_k * delt) : 0.0;
        const double b = 1.0 - a - c;
        if (k == 0) {
          e_prev = -a / b; f1_prev = e1 / b; f2_prev = e2 / b;                         // compute_f
          w.e[0] = e_prev; w.f1[0] = f1_prev; w.f2[0] = f2_prev;
        } else if (k < L - 1) {
          const double g = 1.0 / (b + c * e_prev);
          e_prev = -a * g; f1_prev = (e1 - c * f1_prev) * g; f2_prev = (e2 - c * f2_prev) * g;
          w.e[k * w.sw] = e_prev; w.f1[k * w.sw] = f1_prev; w.f2[k * w.sw2] = f2_prev;
        } else {
          r.mu_delt_n = mu * delt; r.nu_n = nu_k; r.e_n1 = e_prev; r.f1_delt_n1 = f1_prev * delt; r.f2_delt_n1 = f2_prev * delt;
          r.delta_1_n = e1 * delt; r.delta_2_n = e2 * delt;
        }
        fl1_k = fl1_n; fl2_k = fl2_n; nu_k = nu_n; x1_k = x1n[i]; x2_k = x2n[i]; t_k = tn[i]; z_k = zn[i]; ph_k = ph_n;
      }
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i)
      if (k0 + i < L) aux.store(k0 + i, ax[i]);
  }
  return r;
}
}  // namespace vd

// uv_vert_diff (:560-623): dt_u, dt_v become the final tendencies, the dissipated kinetic energy is added to dt_t.
// du_in(k), dv_in(k), dt_in(k): the incoming tendencies as functions of the level (the device kernel knows dt_u, dt_v to be zero below the
// sponge and forms dt_t from its parts; memory reads inside them must be unconditional); the final ones are stored to dt_u, dt_v, dt_t.
template <class DUIN, class DVIN>
MP_HD void vert_diff_momentum_up_f(const vd::DownResult &r, int L, double delt, const double *u, const double *v, int s, double &tau_u, double &tau_v,
                                   double dtau_du, double dtau_dv, DUIN du_in, DVIN dv_in, double *dt_u, double *dt_v, double *dt_t, int st,
                                   double *diss_heat, int sh, const VdiffWork &w, VdiffSurf &S, int kb = 0);
// dt_in(k) is evaluated in the DOWNWARD sweep (it may read what that sweep's e, f then overwrite) and parked in dt_t for the upward one.
template <class DTIN>
struct DtPark {
  DTIN in; double *dt_t; int st;
  MP_HD double load(int k) const { return in(k); }
  MP_HD void store(int k, double v) const { dt_t[k * st] = v; }
};
template <class DIFFM, class DUIN, class DVIN, class DTIN, class PH>
MP_HD void vert_diff_momentum_f(int L, double delt, const double *u, const double *v, const double *t, int s, DIFFM diff_m, PH p_half,
                                const double *z_full, int sp, double &tau_u, double &tau_v, double dtau_du, double dtau_dv, DUIN du_in, DVIN dv_in,
                                DTIN dt_in, double *dt_u, double *dt_v, double *dt_t, int st, double *diss_heat, int sh, const VdiffWork &w,
                                VdiffSurf &S) {
  const vd::DownResult r = vd::down_pair(L, delt, [&](int k) { return u[k * s]; }, [&](int k) { return v[k * s]; }, du_in, dv_in, diff_m, t, s,
                                         p_half, z_full, sp, w, DtPark<DTIN>{dt_in, dt_t, st});
  vert_diff_momentum_up_f(r, L, delt, u, v, s, tau_u, tau_v, dtau_du, dtau_dv, du_in, dv_in, dt_u, dt_v, dt_t, st, diss_heat, sh, w, S);
}
// the surface closure and the upward sweep of uv_vert_diff (the second half of vert_diff_momentum_f: a function of its own so that a caller can
// do -- or time -- the two halves separately)
template <class DUIN, class DVIN>
MP_HD void vert_diff_momentum_up_f(const vd::DownResult &r, int L, double delt, const double *u, const double *v, int s, double &tau_u, double &tau_v,
                                   double dtau_du, double dtau_dv, DUIN du_in, DVIN dv_in, double *dt_u, double *dt_v, double *dt_t, int st,
                                   double *diss_heat, int sh, const VdiffWork &w, VdiffSurf &S, int kb) {
  double delta_u_n = r.delta_1_n, delta_v_n = r.delta_2_n;
  vd::diff_surface(r.mu_delt_n, r.nu_n, r.e_n1, r.f1_delt_n1, dtau_du, tau_u, 1.0, delta_u_n);
  vd::diff_surface(r.mu_delt_n, r.nu_n, r.e_n1, r.f2_delt_n1, dtau_dv, tau_v, 1.0, delta_v_n);
  S.delta_u = delta_u_n; S.delta_v = delta_v_n;
  double xu = delta_u_n / delt, xv = delta_v_n / delt;                              // vert_diff_up (:914-947)
  const double half_delt = 0.5 * delt, cp_inv = 1.0 / CP_AIR;
  for (int k0 = L - 1; k0 >= kb; k0 -= MP_U) {
    double uk[MP_U], vk[MP_U], du0[MP_U], dv0[MP_U], dt0[MP_U], dh[MP_U], ee[MP_U], ff1[MP_U], ff2[MP_U];
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = (k0 - i >= kb) ? k0 - i : kb, ke = (k < L - 1) ? k : L - 2;      // (e, f of the downward sweep: levels kb .. L-2; they may live in global memory)
      uk[i] = u[k * s]; vk[i] = v[k * s]; du0[i] = du_in(k); dv0[i] = dv_in(k); dt0[i] = dt_t[k * st];
      ee[i] = w.e[ke * w.sw]; ff1[i] = w.f1[ke * w.sw]; ff2[i] = w.f2[ke * w.sw2];
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = k0 - i;
      if (k >= kb) {
        if (k < L - 1) { const double e = ee[i]; xu = e * xu + ff1[i]; xv = e * xv + ff2[i]; }
        const double du = xu - du0[i], dv = xv - dv0[i];
        dh[i] = -cp_inv * ((uk[i] + half_delt * du) * du + (vk[i] + half_delt * dv) * dv);
        du0[i] = xu; dv0[i] = xv;
        dt0[i] = dt0[i] + dh[i];
      }
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = k0 - i;
      if (k >= kb) {
        dt_u[k * st] = du0[i]; dt_v[k * st] = dv0[i]; dt_t[k * st] = dt0[i];
        if (diss_heat) diss_heat[k * sh] = dh[i];
      }
    }
  }
}
template <class DIFFM, class PH>
MP_HD void vert_diff_momentum(int L, double delt, const double *u, const double *v, const double *t, int s, DIFFM diff_m, PH p_half,
                              const double *z_full, int sp, double &tau_u, double &tau_v, double dtau_du, double dtau_dv, double *dt_u,
                              double *dt_v, double *dt_t, int st, double *diss_heat, int sh, const VdiffWork &w, VdiffSurf &S) {
  vert_diff_momentum_f(L, delt, u, v, t, s, diff_m, p_half, z_full, sp, tau_u, tau_v, dtau_du, dtau_dv, [&](int k) { return dt_u[k * st]; },
                       [&](int k) { return dt_v[k * st]; }, [&](int k) { return dt_t[k * st]; }, dt_u, dt_v, dt_t, st, diss_heat, sh, w, S);
}
// vert_diff_down_2 for dry static energy and humidity + the Tri_surf hand-over (gcm_vert_diff_down :372-404)
template <class DIFFT, class PH>
MP_HD void vert_diff_heat_down(int L, double delt, const double *t, const double *q, int s, DIFFT diff_t, PH p_half,
                               const double *z_full, int sp, const double *dt_t, const double *dt_q, int st, const VdiffWork &w, VdiffSurf &S, int kb = 0) {
  const double gcp = GRAV / CP_AIR;
  const vd::DownResult r = vd::down_pair(L, delt, [&](int k) { return t[k * s] + z_full[k * sp] * gcp; }, [&](int k) { return q[k * s]; },
                                         [&](int k) { return dt_t[k * st]; }, [&](int k) { return dt_q[k * st]; }, diff_t, t, s, p_half, z_full, sp, w,
                                         vd::NoAux(), kb);
  S.delta_t = r.delta_1_n + r.mu_delt_n * r.nu_n * r.f1_delt_n1;
  S.dflux_t = -r.nu_n * (1.0 - r.e_n1);
  S.delta_q = r.delta_2_n + r.mu_delt_n * r.nu_n * r.f2_delt_n1;
  S.dflux_q = -r.nu_n * (1.0 - r.e_n1);
  S.dtmass = r.mu_delt_n;
}
// gcm_vert_diff_up: final dt_t, dt_q
MP_HD void vert_diff_up(int L, double delt, const VdiffWork &w, const VdiffSurf &S, double *dt_t, double *dt_q, int st, int kb = 0) {
  double xt = S.delta_t / delt, xq = S.delta_q / delt;
  dt_t[(L - 1) * st] = xt; dt_q[(L - 1) * st] = xq;
  for (int k0 = L - 2; k0 >= kb; k0 -= MP_U) {      // chunks: e, f_1, f_2 of MP_U levels requested together (they may live in global memory)
    double ee[MP_U], ff1[MP_U], ff2[MP_U], ot[MP_U], oq[MP_U];
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      const int k = (k0 - i >= kb) ? k0 - i : kb;
      ee[i] = w.e[k * w.sw]; ff1[i] = w.f1[k * w.sw]; ff2[i] = w.f2[k * w.sw2];
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i) {
      if (k0 - i >= kb) { xt = ee[i] * xt + ff1[i]; xq = ee[i] * xq + ff2[i]; }
      ot[i] = xt; oq[i] = xq;
    }
    MP_UNROLL_ALL
    for (int i = 0; i < MP_U; ++i)
      if (k0 - i >= kb) { dt_t[(k0 - i) * st] = ot[i]; dt_q[(k0 - i) * st] = oq[i]; }
  }
}
// The levels 0 .. kb-1 above the boundary layer, which the sweeps started at kb leave out: with e = 0 and f = the incoming tendency the whole sweeps
// give them dt_u = du_in + 0, dt_v = dv_in + 0, dt_t = dt_in + 0 (the dissipation of a zero wind increment is a zero), dt_q = dq_in + 0 -- a plain
// streaming pass, every load of a chunk in flight at once and no recurrence.
template <class DUIN, class DVIN, class DTIN, class DQIN>
MP_HD void vert_diff_passthrough(int ka, int kb, DUIN du_in, DVIN dv_in, DTIN dt_in, DQIN dq_in, double *dt_u, double *dt_v, double *dt_t, double *dt_q,
                                 int st) {      // levels ka .. kb-1
  constexpr int PU = MP_U;
  for (int k0 = ka; k0 < kb; k0 += PU) {
    double a[PU], b[PU], c[PU], d[PU];
    MP_UNROLL_ALL
    for (int i = 0; i < PU; ++i) {
      const int k = (k0 + i < kb) ? k0 + i : kb - 1;
      a[i] = du_in(k); b[i] = dv_in(k); c[i] = dt_in(k); d[i] = dq_in(k);
    }
    MP_UNROLL_ALL
    for (int i = 0; i < PU; ++i)
      if (k0 + i < kb) { const int k = k0 + i; dt_u[k * st] = a[i] + 0.0; dt_v[k * st] = b[i] + 0.0; dt_t[k * st] = c[i] + 0.0; dt_q[k * st] = d[i] + 0.0; }
  }
}

// mixed_layer (atmos_spectral/driver/solo/mixed_layer.F90:568-720): slab ocean of uniform heat capacity closing the implicit
// system; updates t_surf and the lowest-level increments.  dt is dt_atmos (not the leapfrog 2 dt).
struct MixedLayerParams { double heat_capacity = 2.5 * 1.035e3 * 3989.24495292815; bool evaporation = true; double ocean_qflux = 0.0; };
MP_HD void mixed_layer(const MixedLayerParams &p, double dt, double &t_surf, double flux_t, double flux_q, double flux_r, double net_sw,
                       double lw_down, VdiffSurf &S, double dhdt_surf, double dedt_surf, double drdt_surf, double dhdt_atm,
                       double dedq_atm) {
  const double inv_cp_air = 1.0 / CP_AIR;
  const double gamma_t = 1.0 / (1.0 - S.dtmass * (S.dflux_t + dhdt_atm * inv_cp_air));
  const double gamma_q = 1.0 / (1.0 - S.dtmass * (S.dflux_q + dedq_atm));
  const double fn_t = gamma_t * (S.delta_t + S.dtmass * flux_t * inv_cp_air);
  const double fn_q = gamma_q * (S.delta_q + S.dtmass * flux_q);
  const double en_t = gamma_t * S.dtmass * dhdt_surf * inv_cp_air;
  const double en_q = gamma_q * S.dtmass * dedt_surf;
  const double alpha_t = flux_t * inv_cp_air + dhdt_atm * inv_cp_air * fn_t;
  const double alpha_q = flux_q + dedq_atm * fn_q;
  const double alpha_lw = flux_r;
  const double beta_t = dhdt_surf * inv_cp_air + dhdt_atm * inv_cp_air * en_t;
  const double beta_q = dedt_surf + dedq_atm * en_q;
  const double beta_lw = drdt_surf;
  double corrected_flux = -net_sw - lw_down + alpha_t * CP_AIR + alpha_lw - p.ocean_qflux;
  double t_surf_dependence = beta_t * CP_AIR + beta_lw;
  if (p.evaporation) {
    corrected_flux = corrected_flux + alpha_q * HLV;
    t_surf_dependence = t_surf_dependence + beta_q * HLV;
  }
  const double eff_heat_capacity = p.heat_capacity + t_surf_dependence * dt;
  const double delta_t_surf = -corrected_flux * dt / eff_heat_capacity;
  t_surf = t_surf + delta_t_surf;
  S.delta_t = fn_t + en_t * delta_t_surf;
  if (p.evaporation) S.delta_q = fn_q + en_q * delta_t_surf;
}

}  // namespace moist
