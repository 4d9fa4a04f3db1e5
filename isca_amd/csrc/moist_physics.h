// Column physics of the moist configuration (config 3: Frierson grey-radiation aquaplanet), restated from the reference's
// atmos_param / coupler modules as functions of ONE column, usable on the device (k_moist_* kernels) and on the host
// (oracle/moist_host.cpp, the checker built for the CPU tests).  A column field is addressed as x[k * s] (s = level
// stride: lat*lon on the device grid layout [lev][lat][lon]); k = 0 is the model top.
// Options: those of exp/test_cases/frierson/frierson_test_case.py:49-170 (do_simple everywhere, no snow, no virtual
// temperature, grey radiation 'frierson', SIMPLE_BETTS_MILLER convection, diffusivity PBL, mixed-layer surface).
#pragma once
#include <cmath>
#if defined(__HIPCC__)
#define MP_HD __host__ __device__ __forceinline__
#else
#define MP_HD inline
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
  int ind = (int)(t.dtinv * (tmp + t.teps));
  if (ind < 0 || ind >= t.n) {                   // 'table overflow' is FATAL in the reference; NaN marks it here
    es = des = NAN;
    return;
  }
  const double del = tmp - t.dtres * (double)ind;
  es = t.tab[ind] + del * (t.dtab[ind] + del * t.d2tab[ind]);
  des = t.dtab[ind] + 2. * del * t.d2tab[ind];
}
MP_HD double lookup_es(const SatTable &t, double temp) { double e, d; lookup_es_des(t, temp, e, d); return e; }
// compute_qs (sat_vapor_pres_k.F90:457-540) without q: qs = eps es / (p - (1-eps) es), dqs/dT = eps p des / denom^2
MP_HD void compute_qs(const SatTable &t, double temp, double press, double &qs, double &dqsdT) {
  double es, des;
  lookup_es_des(t, temp, es, des);
  const double denom = press - (1.0 - EPSILO) * es;
  qs = (denom > 0.0) ? EPSILO * es / denom : EPSILO;
  dqsdT = EPSILO * press * des / (denom * denom);
}

// ------------------------------------------------------------------------------------------------
// lscale_cond (atmos_param/lscale_cond/lscale_cond.F90:79-212) with do_simple, do_evap, hc = 1:
// saturation adjustment where q > qsat, re-evaporation of the falling precipitation in the layers below
// (precip_evap :215-252); returns the deltas (not rates) and the rain in kg/m2.
// ------------------------------------------------------------------------------------------------
template <int LMAX>
MP_HD void lscale_cond(const SatTable &st, int L, const double *tin, const double *qin, const double *pfull, const double *phalf,
                       int s, double *tdel, double *qdel, double &rain) {
  const double hlcp = HLV / CP_AIR;
  double exq = 0.0, precip = 0.0;
  for (int k = 0; k < L; ++k) {
    double qsat, dqsat;
    compute_qs(st, tin[k * s], pfull[k * s], qsat, dqsat);
    double qd = 0.0, td = 0.0;
    if ((qin[k * s] - qsat) * qsat > 0.0) {
      qd = (qsat - qin[k * s]) / (1.0 + hlcp * dqsat);
      td = -hlcp * qd;
    }
    const double pmass = (phalf[(k + 1) * s] - phalf[k * s]) / GRAV;
    if (qd < 0.0) exq = exq - qd * pmass;
    if (qd >= 0.0 && exq > 0.0) {                 // evaporate precip where needed
      exq = exq / pmass;
      double def = (qsat - qin[k * s]) / (1. + hlcp * dqsat);
      def = fmin(fmax(def, 0.0), exq);
      qd = qd + def;
      td = td - def * hlcp;
      exq = (exq - def) * pmass;
    }
    qdel[k * s] = qd; tdel[k * s] = td;
    precip = precip - pmass * qd;
  }
  rain = fmax(precip, 0.0);
}

// ------------------------------------------------------------------------------------------------
// two_stream_gray_rad 'frierson' (atmos_param/two_stream_gray_rad/two_stream_gray_rad.F90:386-655 down, :659-776 up)
// with do_seasonal = .false.: annual-mean insolation 0.25 S0 (1 + del_sol P2(lat) + del_sw sin lat), SW absorbed with
// optical depth sw_tau_0 (p/p0)^solar_exponent, grey LW with lw_tau_0(lat) (linear_tau p/p0 + (1-linear_tau)(p/p0)^wv_exponent).
// ------------------------------------------------------------------------------------------------
struct GrayRadParams {
  double solar_constant = 1360.0, del_sol = 1.4, del_sw = 0.0, ir_tau_eq = 6.0, ir_tau_pole = 1.5, atm_abs = 0.0, odp = 1.0,
         sw_diff = 0.0, linear_tau = 0.1, wv_exponent = 4.0, solar_exponent = 4.0, diabatic_acce = 1.0;
};
// Downward pass: fills lw_down[0..L] (caller storage, unit stride) and lw_dtrans[0..L-1], returns the surface fluxes.
MP_HD void gray_rad_down(const GrayRadParams &p, int L, double lat, double albedo, const double *t, const double *p_half, int s,
                         double *lw_down, double *lw_dtrans, double &insolation, double &sw_tau_0, double &net_surf_sw_down,
                         double &surf_lw_down) {
  const double sl = sin(lat);
  const double p2 = (1. - 3. * sl * sl) / 4.;
  insolation = 0.25 * p.solar_constant * (1.0 + p.del_sol * p2 + p.del_sw * sl);
  sw_tau_0 = (1.0 - p.sw_diff * sl * sl) * p.atm_abs;
  double lw_tau_0 = p.ir_tau_eq + (p.ir_tau_pole - p.ir_tau_eq) * sl * sl;
  lw_tau_0 = lw_tau_0 * p.odp;
  double tau_k = lw_tau_0 * (p.linear_tau * p_half[0] / PSTD_MKS + (1.0 - p.linear_tau) * pow(p_half[0] / PSTD_MKS, p.wv_exponent));
  lw_down[0] = 0.;
  for (int k = 0; k < L; ++k) {
    const double ph = p_half[(k + 1) * s];
    const double tau_n = lw_tau_0 * (p.linear_tau * ph / PSTD_MKS + (1.0 - p.linear_tau) * pow(ph / PSTD_MKS, p.wv_exponent));
    lw_dtrans[k] = exp(-(tau_n - tau_k));
    const double tk = t[k * s];
    const double b = STEFAN * (tk * tk * tk * tk);
    lw_down[k + 1] = lw_down[k] * lw_dtrans[k] + b * (1. - lw_dtrans[k]);
    tau_k = tau_n;
  }
  surf_lw_down = lw_down[L];
  const double sw_surf = insolation * exp(-sw_tau_0 * pow(p_half[L * s] / PSTD_MKS, p.solar_exponent));
  net_surf_sw_down = sw_surf * (1. - albedo);
}
// Upward pass: temperature tendency of the radiative flux divergence, accumulated into tdt.
MP_HD void gray_rad_up(const GrayRadParams &p, int L, double albedo, double t_surf, const double *t, const double *p_half, int s,
                       const double *lw_down, const double *lw_dtrans, double insolation, double sw_tau_0, double *tdt, int st) {
  const double b_surf = STEFAN * (t_surf * t_surf * t_surf * t_surf);
  const double sw_up = albedo * (insolation * exp(-sw_tau_0 * pow(p_half[L * s] / PSTD_MKS, p.solar_exponent)));
  double lw_up_n = b_surf;                                   // lw_up at half level k+1, integrating upward
  double flux_n = (lw_up_n - lw_down[L]) + (sw_up - insolation * exp(-sw_tau_0 * pow(p_half[L * s] / PSTD_MKS, p.solar_exponent)));
  for (int k = L - 1; k >= 0; --k) {
    const double tk = t[k * s];
    const double b = STEFAN * (tk * tk * tk * tk);
    const double lw_up_k = lw_up_n * lw_dtrans[k] + b * (1.0 - lw_dtrans[k]);
    const double sw_down_k = insolation * exp(-sw_tau_0 * pow(p_half[k * s] / PSTD_MKS, p.solar_exponent));
    const double flux_k = (lw_up_k - lw_down[k]) + (sw_up - sw_down_k);
    const double tdt_rad = p.diabatic_acce * (flux_n - flux_k) * GRAV / (CP_AIR * (p_half[(k + 1) * s] - p_half[k * s]));
    tdt[k * st] = tdt[k * st] + tdt_rad;
    lw_up_n = lw_up_k; flux_n = flux_k;
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

template <int LMAX>
struct QeColumn {            // work arrays of one column, 1-based
  double Tp[LMAX + 2], rp[LMAX + 2], Tv[LMAX + 2], Tref[LMAX + 2], qref[LMAX + 2], dT[LMAX + 2], dq[LMAX + 2];
};

template <int LMAX>
MP_HD void qe_moist_convection(const SatTable &st, const QeParams &P, int L, double dt, const double *Tin_, const double *qin_,
                               const double *p_full_, const double *p_half_, int s, double *deltaT, double *deltaq, double &rain,
                               double &cape_out, double &cin_out, int &convflag, int &kLZB_out, int &kLCL_out, double *Tref_out,
                               double *qref_out, int so) {
  QeColumn<LMAX> c;
  auto Tin = [&](int k) { return Tin_[(k - 1) * s]; };
  auto qin = [&](int k) { return qin_[(k - 1) * s]; };
  auto rin = [&](int k) { const double q = qin_[(k - 1) * s]; return q / (1.0 - q); };
  auto pf = [&](int k) { return p_full_[(k - 1) * s]; };
  auto ph = [&](int k) { return p_half_[(k - 1) * s]; };
  const int ks = L;                                                       // k_surface
  auto set_nocape = [&](double &pLZB, int &kLZB, int &kLFC, double &CIN) {  // set_values_if_nocape (:1014-1030)
    pLZB = pf(1); kLZB = 0; kLFC = 0; CIN = 0.;
    for (int k = 1; k <= L; ++k) { c.Tp[k] = Tin(k); c.rp[k] = rin(k); }
  };
  auto to_model = [&](int k1, int k2) {                                   // set_profiles_to_full_model_values (:1034-1047)
    for (int k = k1; k <= k2; ++k) { c.Tref[k] = Tin(k); c.qref[k] = qin(k); c.dT[k] = 0.; c.dq[k] = 0.; }
  };
  for (int k = 1; k <= L; ++k) { c.dT[k] = 0.; c.dq[k] = 0.; c.Tp[k] = Tin(k); c.rp[k] = rin(k); c.Tv[k] = qe_virtual_temp(Tin(k), rin(k)); }
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
    c.Tp[ks] = T0 + (r0 - rs) / ((CP_AIR / (HLV + QE_SMALL)) + (HLV * rs) / RVGAS / (T0 * T0));
    c.rp[ks] = qe_mixing_ratio(lookup_es(st, c.Tp[ks]), pf(ks));
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
      CIN = 0.;
      while (k >= 1 && pf(k) > pLCL) {
        c.Tp[k] = theta0 * pow(pf(k) / QE_PREF, KAPPA);
        c.rp[k] = qe_mixing_ratio(lookup_es(st, c.Tp[k]), pf(k));
        CIN = CIN + RDGAS * (c.Tv[k] - qe_virtual_temp(c.Tp[k], r0)) * log(ph(k + 1) / ph(k));
        k = k - 1;
      }
      kLCL = k;
      if (kLCL >= 1) {
        double a = KAPPA * TLCL + (HLV / CP_AIR) * r0;
        double b = (HLV * HLV) * r0 / (CP_AIR * RVGAS * (TLCL * TLCL));
        double dtdlnp = a / (1.0 + b);
        c.Tp[kLCL] = TLCL + dtdlnp * log(pf(kLCL) / pLCL) / 2;
        if ((c.Tp[kLCL] < P.Tmin) && nocape) {
          skip = true;
          set_nocape(pLZB, kLZB, kLFC, CIN);
        } else {
          c.rp[kLCL] = qe_mixing_ratio(lookup_es(st, c.Tp[kLCL]), (pf(kLCL) + pLCL) / 2);
          a = KAPPA * c.Tp[kLCL] + (HLV / CP_AIR) * c.rp[kLCL];
          b = (HLV * HLV) * c.rp[kLCL] / (CP_AIR * RVGAS * (c.Tp[kLCL] * c.Tp[kLCL]));
          dtdlnp = a / (1.0 + b);
          c.Tp[kLCL] = TLCL + dtdlnp * log(pf(kLCL) / pLCL);
          if ((c.Tp[kLCL] < P.Tmin) && nocape) {
            skip = true;
            set_nocape(pLZB, kLZB, kLFC, CIN);
          } else {
            c.rp[kLCL] = qe_mixing_ratio(lookup_es(st, c.Tp[kLCL]), pf(kLCL));
            const double tvp = qe_virtual_temp(c.Tp[kLCL], c.rp[kLCL]);
            if ((tvp < c.Tv[kLCL]) && nocape) {
              CIN = CIN + RDGAS * (c.Tv[kLCL] - tvp) * log(ph(kLCL + 1) / ph(kLCL));
            } else {
              CAPE = CAPE + RDGAS * (tvp - c.Tv[kLCL]) * log(ph(kLCL + 1) / ph(kLCL));
              if (nocape) { nocape = false; kLFC = kLCL; }
            }
          }
        }
      }
    }
  }
  // ---- CAPE_above_LCL (:587-668)
  if (skip) {
    if (nocape) set_nocape(pLZB, kLZB, kLFC, CIN);
  } else {
    for (int k = kLCL - 1; k >= 1; --k) {
      double a = KAPPA * c.Tp[k + 1] + (HLV / CP_AIR) * c.rp[k + 1];
      double b = (HLV * HLV) * c.rp[k + 1] / (CP_AIR * RVGAS * (c.Tp[k + 1] * c.Tp[k + 1]));
      double dtdlnp = a / (1.0 + b);
      c.Tp[k] = c.Tp[k + 1] + dtdlnp * log(pf(k) / pf(k + 1)) / 2;
      if ((c.Tp[k] < P.Tmin) && nocape) { set_nocape(pLZB, kLZB, kLFC, CIN); break; }
      c.rp[k] = qe_mixing_ratio(lookup_es(st, c.Tp[k]), (pf(k) + pf(k + 1)) / 2);
      a = KAPPA * c.Tp[k] + (HLV / CP_AIR) * c.rp[k];
      b = (HLV * HLV) * c.rp[k] / (CP_AIR * RVGAS * (c.Tp[k] * c.Tp[k]));
      dtdlnp = a / (1.0 + b);
      c.Tp[k] = c.Tp[k + 1] + dtdlnp * log(pf(k) / pf(k + 1));
      if ((c.Tp[k] < P.Tmin) && nocape) { set_nocape(pLZB, kLZB, kLFC, CIN); break; }
      c.rp[k] = qe_mixing_ratio(lookup_es(st, c.Tp[k]), pf(k));
      const double tvp = qe_virtual_temp(c.Tp[k], c.rp[k]);
      if ((tvp < c.Tv[k]) && nocape) {
        CIN = CIN + RDGAS * (c.Tv[k] - tvp) * log(ph(k + 1) / ph(k));
      } else if ((tvp < c.Tv[k]) && !nocape) {
        kLZB = k + 1;
        break;
      } else {
        CAPE = CAPE + RDGAS * (tvp - c.Tv[k]) * log(ph(k + 1) / ph(k));
        if (nocape) { nocape = false; kLFC = k; }
      }
    }
  }
  cape_out = CAPE; cin_out = CIN; kLZB_out = kLZB; kLCL_out = kLCL;
  (void)kLFC; (void)pLZB;
  convflag = 0;
  double Pq = 0.;
  if (CAPE > 0) {
    convflag = 1;
    // ---- set_reference_profiles (:768-796)
    for (int k = 1; k <= L; ++k) c.Tref[k] = c.Tp[k];
    for (int k = (kLZB < 1 ? 1 : kLZB); k <= ks; ++k) {
      if (k < kLZB) continue;
      const double eref = P.rhbm * pf(k) * c.rp[k] / (c.rp[k] + (RDGAS / RVGAS));
      c.rp[k] = qe_mixing_ratio(eref, pf(k));
      c.qref[k] = c.rp[k] / (1 + c.rp[k]);
    }
    { const int kk = (kLZB - 1 > 1) ? kLZB - 1 : 1; to_model(1, kk); }
    // ---- Pq_calculation (:710-733), Pt_calculation (:737-764)
    double Pt = 0.;
    for (int k = kLZB; k <= ks; ++k) {
      if (k < 1) continue;
      c.dq[k] = -(qin(k) - c.qref[k]) * dt / P.tau_bm;
      Pq = Pq + c.dq[k] * (ph(k) - ph(k + 1));
    }
    Pq = Pq / GRAV;
    for (int k = kLZB; k <= ks; ++k) {
      if (k < 1) continue;
      c.dT[k] = -(Tin(k) - c.Tref[k]) * dt / P.tau_bm;
      Pt = Pt + (CP_AIR / (HLV + QE_SMALL)) * c.dT[k] * (ph(k + 1) - ph(k));
    }
    Pt = Pt / GRAV;
    if ((Pq > 0) && (Pt > 0)) {
      convflag = 2;
      if (Pq > Pt) {                                  // do_change_time_scale_deepconv (:992-1008)
        const double invtau_q = Pt / Pq / P.tau_bm;
        for (int k = kLZB; k <= ks; ++k) c.dq[k] = P.tau_bm * invtau_q * c.dq[k];
        Pq = Pt;
      } else {                                        // do_change_Tref_deepconv (:957-988)
        double deltak = 0.;
        for (int k = kLZB; k <= ks; ++k) deltak = deltak - (c.dT[k] + (HLV / CP_AIR) * c.dq[k]) * (ph(k + 1) - ph(k));
        deltak = deltak / (ph(ks + 1) - ph(kLZB));
        for (int k = kLZB; k <= ks; ++k) { c.Tref[k] = c.Tref[k] + deltak * P.tau_bm / dt; c.dT[k] = c.dT[k] + deltak; }
      }
    } else if (Pt > 0) {
      // ---- do_shallow_convection (:800-840) with level_of_zero_precip (:844-888)
      int k = kLZB;
      bool found = false;
      while ((Pq < 0.) && (k <= ks)) {
        Pq = Pq - c.dq[k] * (ph(k) - ph(k + 1)) / GRAV;
        k = k + 1;
      }
      const int k_top = k - 1;
      if (Pq > 0.) found = true;
      if (k_top > kLZB) to_model(kLZB, k_top - 1);
      if (found) {                                    // change_Tref_LZB_shallowconv (:893-930)
        const double cc = Pq * GRAV / (c.dq[k_top] * (ph(k_top + 1) - ph(k_top)));
        c.dq[k_top] = c.dq[k_top] * cc;
        c.dT[k_top] = c.dT[k_top] * cc;
        double deltak = 0.;
        for (int kk = k_top; kk <= ks; ++kk) deltak = deltak + c.dT[kk] * (ph(kk) - ph(kk + 1));
        deltak = deltak / (ph(ks + 1) - ph(k_top));
        if (k_top != ks)
          for (int kk = k_top; kk <= ks; ++kk) { c.dT[kk] = c.dT[kk] + deltak; c.Tref[kk] = c.Tref[kk] + deltak * P.tau_bm / dt; }
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
  for (int k = 1; k <= L; ++k) {
    deltaT[(k - 1) * so] = c.dT[k]; deltaq[(k - 1) * so] = c.dq[k];
    if (Tref_out) Tref_out[(k - 1) * so] = c.Tref[k];
    if (qref_out) qref_out[(k - 1) * so] = c.qref[k];
  }
}

}  // namespace moist
