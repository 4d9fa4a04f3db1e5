// TEST INFRASTRUCTURE ONLY: host build of the column routines of isca_amd/csrc/moist_physics.h, so that the CPU test
// suite can compare them with the reference's outputs (tests/golden/moist_*.npz) without a GPU.  Built by
// oracle/build_moist_host.py (g++ -O2 -ffp-contract=off) into oracle/_ref/libmoist_host.so; never loaded by the product.
#include "../isca_amd/csrc/moist_physics.h"
#include "../isca_amd/csrc/moist_tables.h"
#include <vector>

using namespace moist;

static SatTableHost g_sat;
static SatTable sat() { if (g_sat.tab.empty()) g_sat.build(); return g_sat.view(); }

static QeTablesHost g_qe;
static QeParams qe() { if (g_qe.lcl.empty()) g_qe.build(sat()); return g_qe.params; }

extern "C" {
void mh_qe_moist_convection(int L, int ncol, double dt, const double *tin, const double *qin, const double *pfull, const double *phalf,
                            double *dT, double *dq, double *rain, double *cape, double *cin, double *flag, double *klzb, double *klcl,
                            double *tref, double *qref) {
  const SatTable st = sat();
  const QeParams P = qe();
  for (int c = 0; c < ncol; ++c) {
    int f, kz, kl;
    qe_moist_convection<64>(st, P, L, dt, tin + c, qin + c, pfull + c, phalf + c, ncol, dT + c, dq + c, rain[c], cape[c], cin[c], f, kz, kl,
                            tref + c, qref + c, ncol);
    flag[c] = f; klzb[c] = kz; klcl[c] = kl;
  }
}
// arrays are [lev][ncol] (column c at x[k*ncol + c])
void mh_lookup_es_des(int n, const double *t, double *es, double *des) {
  const SatTable st = sat();
  for (int i = 0; i < n; ++i) lookup_es_des(st, t[i], es[i], des[i]);
}
void mh_lscale_cond(int L, int ncol, const double *tin, const double *qin, const double *pfull, const double *phalf, double *tdel,
                    double *qdel, double *rain) {
  const SatTable st = sat();
  for (int c = 0; c < ncol; ++c) lscale_cond<64>(st, L, tin + c, qin + c, pfull + c, phalf + c, ncol, tdel + c, qdel + c, rain[c]);
}
void mh_gray_rad(int L, int ncol, double atm_abs, const double *lat, const double *albedo, const double *t_surf, const double *t,
                 const double *p_half, double *net_sw, double *lw_down_surf, double *tdt) {
  GrayRadParams p; p.atm_abs = atm_abs;
  std::vector<double> lwd(L + 1), ltr(L);
  for (int c = 0; c < ncol; ++c) {
    double ins, tau0;
    gray_rad_down(p, L, lat[c], albedo[c], t + c, p_half + c, ncol, lwd.data(), ltr.data(), ins, tau0, net_sw[c], lw_down_surf[c]);
    gray_rad_up(p, L, albedo[c], t_surf[c], t + c, p_half + c, ncol, lwd.data(), ltr.data(), ins, tau0, tdt + c, ncol);
  }
}
}
