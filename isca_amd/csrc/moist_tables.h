// Host-side construction of the lookup tables of the moist physics (built once per handle, uploaded to the device).
#pragma once
#include "moist_physics.h"
#include <vector>

namespace moist {

// sat_vapor_pres_init_k (shared/sat_vapor_pres/sat_vapor_pres_k.F90:161-265) with do_simple: tcmin_simple = -173,
// tcmax_simple = 350, esres = 10 (sat_vapor_pres.F90:529-534, 2307-2317)
struct SatTableHost {
  std::vector<double> tab, dtab, d2tab;
  double tmin = 0, dtinv = 0, teps = 0, dtres = 0;
  void build(int tcmin = -173, int tcmax = 350, int esres = 10, double es0 = 1.0) {
    const int n = (tcmax - tcmin) * esres + 1;
    tab.resize(n); dtab.resize(n); d2tab.resize(n);
    dtres = ((double)tcmax - (double)tcmin) / (double)(n - 1);
    tmin = (double)tcmin + TFREEZE;
    dtinv = 1. / dtres;
    teps = .5 * dtres;
    for (int i = 0; i < n; ++i) {
      const double tem = tmin + dtres * (double)i;
      tab[i] = es0 * 610.78 * std::exp(-HLV / RVGAS * (1. / tem - 1. / TFREEZE));
      dtab[i] = HLV * tab[i] / RVGAS / (tem * tem);
    }
    for (int i = 1; i < n - 1; ++i) d2tab[i] = 0.25 * dtinv * (dtab[i + 1] - dtab[i - 1]);
    d2tab[0] = 0.50 * dtinv * (dtab[1] - dtab[0]);
    d2tab[n - 1] = 0.50 * dtinv * (dtab[n - 1] - dtab[n - 2]);
  }
  SatTable view() const { return SatTable{tab.data(), dtab.data(), d2tab.data(), tmin, dtinv, teps, dtres, (int)tab.size()}; }
};

// qe_moist_convection_init (qe_moist_convection.F90:105-186): value range from Tmin/Tmax, table of LCL temperatures by Newton
// iteration (lcl_temp :1086-1150), each entry started from the previous one
struct QeTablesHost {
  std::vector<double> lcl;
  QeParams params;
  void build(const SatTable &st, double rhbm = 0.7, double Tmin = 160., double Tmax = 350., double tau_bm = 7200., double val_inc = 0.01) {
    params.rhbm = rhbm; params.Tmin = Tmin; params.Tmax = Tmax; params.tau_bm = tau_bm; params.val_inc = val_inc;
    const double esmin = lookup_es(st, Tmin), esmax = lookup_es(st, Tmax);
    params.val_min = std::log(esmin / std::pow(Tmin, 1.0 / KAPPA));
    params.val_max = std::log(esmax / std::pow(Tmax, 1.0 / KAPPA));
    const int n = (int)std::ceil((params.val_max - params.val_min) / val_inc);
    lcl.resize(n);
    double guess = Tmin;
    for (int k = 0; k < n; ++k) {
      const double value = params.val_min + k * val_inc;
      double T = guess, dT = 1.e-7 + 1.;
      int iter = 0;
      while ((std::fabs(dT) > 1.e-7) && (iter < 100)) {
        const double f = value - std::log(lookup_es(st, T) * std::pow(T, -1 / KAPPA));
        const double df = 1 / KAPPA * (1.0 / T) - HLV / RVGAS * (1.0 / (T * T));
        dT = f / df;
        T = T - dT;
        iter = iter + 1;
      }
      lcl[k] = T;
      guess = T;
    }
    params.lcl_temp_table = lcl.data();
    params.table_size = n;
  }
};

}  // namespace moist
