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
  std::vector<double> wtp(L), wrp(L);
  for (int c = 0; c < ncol; ++c) {
    int f, kz, kl;
    qe_moist_convection<64>(st, P, L, dt, tin + c, qin + c, pfull + c, phalf + c, ncol, dT + c, dq + c, rain[c], cape[c], cin[c], f, kz, kl,
                            tref + c, qref + c, ncol, QeParcel{wtp.data(), wrp.data(), 1});
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
  for (int c = 0; c < ncol; ++c) lscale_cond(st, L, [&](int k, double &t, double &q, double &, double &) { t = tin[k * ncol + c]; q = qin[k * ncol + c]; }, pfull + c, phalf + c, ncol,
                                             [&](int k, double td, double qd, double, double) { tdel[k * ncol + c] = td; qdel[k * ncol + c] = qd; }, rain[c]);
}
void mh_gray_rad(int L, int ncol, double atm_abs, const double *lat, const double *albedo, const double *t_surf, const double *t,
                 const double *p_half, double *net_sw, double *lw_down_surf, double *tdt) {
  GrayRadParams p; p.atm_abs = atm_abs;
  std::vector<double> lwd(L + 1), ltr(L), swd(L + 1);
  for (int c = 0; c < ncol; ++c) {
    double ins, tau0;
    gray_rad_down(p, L, lat[c], albedo[c], t + c, p_half + c, ncol, lwd.data(), ltr.data(), 1, swd.data(), 1, ins, tau0, net_sw[c], lw_down_surf[c]);
    gray_rad_up(p, L, albedo[c], t_surf[c], t + c, p_half + c, ncol, lwd.data(), ltr.data(), 1, swd.data(), 1, tdt + c, ncol);
  }
}
// out: 21 doubles per column in the order of struct SurfFlux
void mh_surface_flux(int ncol, const double *t_atm, const double *q_atm, const double *u_atm, const double *v_atm, const double *p_atm,
                     const double *z_atm, const double *p_surf, const double *t_surf, double rough, double gust, double *out) {
  const SatTable st = sat();
  MoParams mo;
  for (int c = 0; c < ncol; ++c) {
    SurfFlux o;
    surface_flux(st, mo, t_atm[c], q_atm[c], u_atm[c], v_atm[c], p_atm[c], z_atm[c], p_surf[c], t_surf[c], rough, rough, rough, rough, gust, o);
    const double v[21] = {o.flux_t, o.flux_q, o.flux_r, o.flux_u, o.flux_v, o.dhdt_surf, o.dedt_surf, o.dedq_surf, o.drdt_surf, o.dhdt_atm,
                          o.dedq_atm, o.dtaudu_atm, o.dtaudv_atm, o.w_atm, o.u_star, o.b_star, o.q_star, o.cd_m, o.cd_t, o.cd_q, o.q_surf};
    for (int i = 0; i < 21; ++i) out[c * 21 + i] = v[i];
  }
}
void mh_rayleigh(int L, int ncol, int nlev_rayfric, double rfactr, double sponge_pbottom, double dt, const double *pfull, const double *u,
                 const double *v, double *udt, double *vdt, double *tdt) {
  RayleighParams p; p.nlev_rayfric = nlev_rayfric; p.rfactr = rfactr; p.sponge_pbottom = sponge_pbottom;
  for (int c = 0; c < ncol; ++c) rayleigh_damping(p, dt, pfull + c, u + c, v + c, ncol, udt + c, vdt + c, ncol, tdt + c, ncol);
}
void mh_diffusivity(int L, int ncol, double dt, const double *tm, const double *um, const double *vm, const double *tdt, const double *udt,
                    const double *vdt, const double *z_full, const double *z_half, const double *u_star, const double *b_star, double *h,
                    double *k_m, double *k_t) {
  MoParams mo; DiffusivityParams dp;
  for (int c = 0; c < ncol; ++c) {
    h[c] = pbl_depth(dp, L, dt, tm + c, um + c, vm + c, ncol, tdt + c, udt + c, vdt + c, ncol, z_full + c, z_half + c, ncol);
    PblProfile pr;
    pr.init(mo, dp, h[c], u_star[c], b_star[c], z_half + c, ncol, L);
    for (int k = 0; k < L; ++k) { k_m[k * ncol + c] = pr.k_m(k); k_t[k * ncol + c] = pr.k_t(k); }
  }
}
// surf: 7 doubles per column (VdiffSurf) after the downward sweep; surf_ml: the same after mixed_layer
void mh_vert_diff(int L, int ncol, double delt, double dt_atmos, const double *u, const double *v, const double *t, const double *q,
                  const double *diff_m, const double *diff_t, const double *p_half, const double *p_full, const double *z_full,
                  const double *flux_u, const double *flux_v, const double *dtau_du, const double *dtau_dv, double *dt_u, double *dt_v,
                  double *dt_t, double *dt_q, double *diss_heat, double *surf, double *t_surf, const double *flux_t, const double *flux_q,
                  const double *flux_r, const double *net_sw, const double *lw_down, const double *dhdt_surf, const double *dedt_surf,
                  const double *drdt_surf, const double *dhdt_atm, const double *dedq_atm, double *surf_ml, double *dt_t_down) {
  MixedLayerParams ml;
  std::vector<double> we(L), wf1(L), wf2(L);
  (void)p_full;
  for (int c = 0; c < ncol; ++c) {
    VdiffWork w{we.data(), wf1.data(), wf2.data(), 1, 1};
    VdiffSurf S;
    double tu = flux_u[c], tv = flux_v[c];
    vert_diff_momentum(L, delt, u + c, v + c, t + c, ncol, vd::TableDiff{diff_m + c, ncol}, p_half + c, z_full + c, ncol, tu, tv,
                       dtau_du[c], dtau_dv[c], dt_u + c, dt_v + c, dt_t + c, ncol, diss_heat + c, ncol, w, S);
    vert_diff_heat_down(L, delt, t + c, q + c, ncol, vd::TableDiff{diff_t + c, ncol}, p_half + c, z_full + c, ncol, dt_t + c,
                        dt_q + c, ncol, w, S);
    const double a[7] = {S.dtmass, S.dflux_t, S.delta_t, S.dflux_q, S.delta_q, S.delta_u, S.delta_v};
    for (int i = 0; i < 7; ++i) surf[c * 7 + i] = a[i];
    for (int k = 0; k < L; ++k) dt_t_down[k * ncol + c] = dt_t[k * ncol + c];
    mixed_layer(ml, dt_atmos, t_surf[c], flux_t[c], flux_q[c], flux_r[c], net_sw[c], lw_down[c], S, dhdt_surf[c], dedt_surf[c],
                drdt_surf[c], dhdt_atm[c], dedq_atm[c]);
    const double b[7] = {S.dtmass, S.dflux_t, S.delta_t, S.dflux_q, S.delta_q, S.delta_u, S.delta_v};
    for (int i = 0; i < 7; ++i) surf_ml[c * 7 + i] = b[i];
    vert_diff_up(L, delt, w, S, dt_t + c, dt_q + c, ncol);
  }
}
// The implicit diffusion with its sweeps limited to the boundary layer (down_pair / vert_diff_*_up with kb = pbl_depth_f's kstop, the levels above through
// vert_diff_passthrough): what the device kernel runs.  Same arguments as mh_vert_diff plus the start level per column.
void mh_pbl_kstop(int L, int ncol, double dt, const double *tm, const double *um, const double *vm, const double *tdt, const double *udt,
                  const double *vdt, const double *z_full, const double *z_half, int *kstop) {
  DiffusivityParams dp;
  for (int c = 0; c < ncol; ++c)
    pbl_depth_f(dp, L, dt, tm + c, um + c, vm + c, ncol, [&](int k) { return tdt[k * ncol + c]; }, [&](int k) { return udt[k * ncol + c]; },
                [&](int k) { return vdt[k * ncol + c]; }, z_full + c, z_half + c, ncol, kstop + c);
}
void mh_vert_diff_kb(int L, int ncol, double delt, double dt_atmos, const double *u, const double *v, const double *t, const double *q,
                     const double *diff_m, const double *diff_t, const double *p_half, const double *z_full, const double *flux_u,
                     const double *flux_v, const double *dtau_du, const double *dtau_dv, double *dt_u, double *dt_v, double *dt_t, double *dt_q,
                     double *diss_heat, double *t_surf, const double *flux_t, const double *flux_q, const double *flux_r, const double *net_sw,
                     const double *lw_down, const double *dhdt_surf, const double *dedt_surf, const double *drdt_surf, const double *dhdt_atm,
                     const double *dedq_atm, const int *kbs) {
  MixedLayerParams ml;
  std::vector<double> we(L), wf1(L), wf2(L);
  for (int c = 0; c < ncol; ++c) {
    const int kb = std::min(kbs[c], L - 2);
    VdiffWork w{we.data(), wf1.data(), wf2.data(), 1, 1};
    VdiffSurf S;
    double tu = flux_u[c], tv = flux_v[c];
    auto du_in = [&](int k) { return dt_u[k * ncol + c]; };
    auto dv_in = [&](int k) { return dt_v[k * ncol + c]; };
    auto dt_in = [&](int k) { return dt_t[k * ncol + c]; };
    auto dq_in = [&](int k) { return dt_q[k * ncol + c]; };
    vert_diff_passthrough(0, kb, du_in, dv_in, dt_in, dq_in, dt_u + c, dt_v + c, dt_t + c, dt_q + c, ncol);
    const vd::DownResult r = vd::down_pair(L, delt, [&](int k) { return u[k * ncol + c]; }, [&](int k) { return v[k * ncol + c]; }, du_in, dv_in,
                                           vd::TableDiff{diff_m + c, ncol}, t + c, ncol, p_half + c, z_full + c, ncol, w,
                                           DtPark<decltype(dt_in)>{dt_in, dt_t + c, ncol}, kb);
    vert_diff_momentum_up_f(r, L, delt, u + c, v + c, ncol, tu, tv, dtau_du[c], dtau_dv[c], du_in, dv_in, dt_u + c, dt_v + c, dt_t + c, ncol,
                            diss_heat + c, ncol, w, S, kb);
    vert_diff_heat_down(L, delt, t + c, q + c, ncol, vd::TableDiff{diff_t + c, ncol}, p_half + c, z_full + c, ncol, dt_t + c, dt_q + c, ncol, w, S, kb);
    mixed_layer(ml, dt_atmos, t_surf[c], flux_t[c], flux_q[c], flux_r[c], net_sw[c], lw_down[c], S, dhdt_surf[c], dedt_surf[c], drdt_surf[c],
                dhdt_atm[c], dedq_atm[c]);
    vert_diff_up(L, delt, w, S, dt_t + c, dt_q + c, ncol, kb);
  }
}
}
