// Moist physics package on the device: idealized_moist_phys (atmos_spectral/driver/solo/idealized_moist_phys.F90:819-1340)
// with the Frierson grey-radiation aquaplanet options, one thread per column, levels strided by the column count
// (consecutive lanes = consecutive longitudes: every level access of a wavefront is one coalesced row segment).
// The column routines themselves are in moist_physics.h (shared with the host build the CPU tests check against the
// reference); this file holds the driver kernel in the reference's call order, the table upload and the launchers.
#include <type_traits>
#include "core.h"
#include "kernels.h"
#include "moist.h"
#include "moist_tables.h"

namespace isca {

struct MoistArgs {
  int ncol, L, I;                       // columns, levels, columns per latitude row (for rad_lat_row)
  const double *up, *vp, *tp, *qp;      // previous time level [L][ncol]
  const double *pf_p, *ph_p;            // pressures of the previous level (convection, condensation)
    const double *pf_c, *ph_c; double *zf_c, *zh_c;   // pressures and heights of the current level (everything else)
  const double *rad_lat_row, *rad_lat_col;   // latitude by row (grid) or by column (caller fields)
  double *t_surf;                       // mixed-layer temperature, updated
  double *dtu, *dtv, *dtT, *dtq;        // tendencies out
  double *precip;                       // convective + large-scale rain rate [ncol] (kg/m2/s)
  const double *pk, *bk, *ps_p, *ps_c;  // in the model's step (SIG kernels): the half-level pressures are pk + bk ps, formed where they are needed; ph_p / ph_c unused
  const double *surf_geop;              // non-null: zf_c / zh_c hold the hydrostatic increments of k_moist_pressures and this kernel sums them (moist_heights_scan)
  int ktop;
  double *work;                         // [5][L+1][ncol]: the three work arrays when they do not fit LDS; the radiation's level arrays (0, 1, 3), the sponge's heating (4)
  double delta_t, dt_atmos, gust, albedo;
  double rough_mom, rough_heat, rough_moist;
  moist::SatTable sat;
  moist::QeParams qe;
  moist::GrayRadParams rad;
  moist::MoParams mo;
  moist::RayleighParams ray;
  moist::DiffusivityParams dif;
  moist::MixedLayerParams ml;
  int do_damping;
};

// Three work arrays of L+1 levels per column (0, 1: the parcel, then the convection's deltas, then the diffusion's e, f_1; 2: the radiative
// heating, then f_2) live in LDS when the block's 3 x 64 x (L+1) doubles fit the 64 KB a block may take without opting in (L <= 41); up to
// L = 63 arrays 0 and 1 do and array 2 is a global buffer with the grid layout; beyond that all three are.  LMAX only sizes the private arrays of the convection scheme.
// A column is a chain of latency-bound recurrences and a T85 grid is only 512 wavefronts of columns, so a block runs TWO wavefronts
// on its 64 columns where the chain allows it: wavefront 0 does the convection and the condensation (T85L40: 66 us in the first days after a
// cold start, 98 us with a mean of 85 and up to 119 for the convection alone after 35 days, when it rains) while wavefront 1 does the height
// sum, the radiation, the surface fluxes and the sponge (84 / 98 us); they meet at one barrier, after which wavefront 0 goes on alone with the
// boundary layer and the implicit diffusion (53-59 us), wavefront 1's heating entering dt_tg in the reference's order
// (dt_tg = ((conv + cond) + rad) + sponge).  With blockDim = 64 the same code runs the parts one after the other.
// Every pass over a level array that leaves the chip costs HBM time here (32 768 columns x 14 fields do not fit the L2s, so a re-read
// is a miss: rocprofv3 counted 712 MB per launch for 147 MB of fields, r03 profile): values that have one reader are handed over in LDS
// (the convection's deltas in the parcel's arrays), sums are formed where their result is stored (dt_tg above), known zeros are not
// stored and re-read (dt_ug, dt_vg below the sponge), and the convection scheme keeps one private level array (Tv) instead of five.
constexpr int MOIST_NX = 20;       // scalars handed from wavefront 1 to wavefront 0 through work array 0 (needs L + 1 >= MOIST_NX)
// Phase timing for kernel experiments (tools/dev/moist_phase_times.py): built with -DMOIST_TIMING=p the kernel stamps wall_clock64 (10 ns
// ticks) at the marks of phase p (1: convection + condensation, 2: radiation + surface flux + sponge, 3: after the barrier) and stores,
// in lane i of every wavefront, the time between marks i and i+1 in place of the precipitation.
#ifdef MOIST_TIMING
#define MT_DECL long long mt_[9]; for (int i_ = 0; i_ < 9; ++i_) mt_[i_] = wall_clock64();
#define MT(p, i) if (MOIST_TIMING == p) { const long long t_ = wall_clock64(); for (int i_ = i; i_ < 9; ++i_) mt_[i_] = t_; }
#define MT_STORE(p) if (MOIST_TIMING == p) { long long d_ = 0; for (int i_ = 0; i_ < 8; ++i_) if ((lane & 7) == i_) d_ = mt_[i_ + 1] - mt_[i_]; a.precip[c] = (double)d_; }
#else
#define MT_DECL
#define MT(p, i)
#define MT_STORE(p)
#endif
// The hydrostatic sum, bottom-up, of one column: z_full / z_half arrive as the layers' increments (k_moist_pressures) and leave as heights; 8 levels of
// increments requested together.  (Was a kernel of its own: 512 wavefronts, five memory round trips, 12 us; the radiation wavefront does it on the way.)
__device__ __forceinline__ void moist_heights_scan(double gh, int L, int ktop, double *z_full, double *z_half, size_t s) {
  z_half[(size_t)L * s] = gh / GRAV;
  for (int k0 = L - 1; k0 >= 0; k0 -= 8) {
    double zf[8], dz[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = (k0 - i >= 0) ? k0 - i : 0;
      zf[i] = z_full[(size_t)k * s]; dz[i] = z_half[(size_t)k * s];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k0 - i;
      if (k >= 0) {
        z_full[(size_t)k * s] = (gh + zf[i]) / GRAV;
        if (k >= ktop) gh = gh + dz[i];
        z_half[(size_t)k * s] = (k >= ktop) ? gh / GRAV : 0.0;
      }
    }
  }
}
template <int LMAX, int NLDS, bool SIG>      // NLDS: how many of the three work arrays live in LDS (3, 2: arrays 0 and 1, or 0); SIG: p_half from (pk, bk, ps)
__global__ __launch_bounds__(128) void k_moist_physics(MoistArgs a) {
  constexpr bool LDSW = NLDS >= 2;
  extern __shared__ __attribute__((aligned(16))) double lds_work[];
  const int lane = threadIdx.x & 63, role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nroles = blockDim.x >> 6;
  const int col = min(blockIdx.x * 64 + lane, a.ncol - 1);     // the tail lanes redo the last column (same values stored)
  const int L = a.L, s = a.ncol;
  const size_t c = (size_t)col;
  const int sw = LDSW ? 64 : a.ncol, sw2 = (NLDS == 3) ? 64 : a.ncol;
  double *w0 = LDSW ? lds_work + lane : a.work + c;
  double *w1 = w0 + (size_t)(L + 1) * sw;
  double *w2 = (NLDS == 3) ? w1 + (size_t)(L + 1) * sw : a.work + c + (size_t)2 * (L + 1) * a.ncol;
  // the radiation / sponge wavefront keeps its two level arrays and the scalars it hands over in the global work area, so that LDS
  // arrays 0 and 1 belong to the convection (parcel profile) before the barrier and to the implicit diffusion (e, f) after it
  double *r0 = a.work + c, *r1 = r0 + (size_t)(L + 1) * s, *r3 = r0 + (size_t)3 * (L + 1) * s, *r4 = r0 + (size_t)4 * (L + 1) * s;
  const double *tp = a.tp + c, *qp = a.qp + c, *up = a.up + c, *vp = a.vp + c;
  using PHT = typename std::conditional<SIG, moist::PHalfSigma, const double *>::type;
  PHT php, phc;                         // half-level pressures of the previous / current time level
  if constexpr (SIG) { php = moist::PHalfSigma{a.pk, a.bk, a.ps_p[c]}; phc = moist::PHalfSigma{a.pk, a.bk, a.ps_c[c]}; }
  else { php = a.ph_p + c; phc = a.ph_c + c; }
  double *dtu = a.dtu + c, *dtv = a.dtv + c, *dtT = a.dtT + c, *dtq = a.dtq + c;
  const double delta_t = a.delta_t;
  const int nray = a.do_damping ? a.ray.nlev_rayfric : 0;
  double t_surf = 0.0, net_sw = 0.0, lw_down_surf = 0.0;
  moist::SurfFlux sf;
  MT_DECL
  // ---- wavefront nroles-1: grey radiation down (:1054-1061), surface fluxes (:1077-1153), radiation up (:1156-1162): heating into work array 2
  if (role == nroles - 1) {
    if (a.surf_geop) moist_heights_scan(a.surf_geop[c], L, a.ktop, a.zf_c + c, a.zh_c + c, (size_t)s);      // (its first reader is the surface flux below)
    const double lat = a.rad_lat_col ? a.rad_lat_col[c] : a.rad_lat_row[col / a.I];
    t_surf = a.t_surf[c];
    double insolation, sw_tau_0;
    moist::gray_rad_down(a.rad, L, lat, a.albedo, tp, phc, s, r0, r1, r3, s, insolation, sw_tau_0, net_sw, lw_down_surf);
    MT(2, 1)
    const size_t low = (size_t)(L - 1) * s;
    moist::surface_flux(a.sat, a.mo, tp[low], qp[low], up[low], vp[low], a.pf_c[c + low], a.zf_c[c + low], moist::ph_at(phc, s, L), t_surf,
                        a.rough_mom, a.rough_heat, a.rough_moist, a.rough_mom, a.gust, sf);
    MT(2, 2)
    for (int k = 0; k < L; ++k) w2[k * sw2] = 0.0;
    moist::gray_rad_up(a.rad, L, a.albedo, t_surf, tp, phc, s, r0, r1, r3, s, w2, sw2);
    MT(2, 3)
    // ---- Rayleigh sponge (:1228-1237; on this wavefront because in a spun-up model the convection wavefront is the longer of the two in front of
    //      the barrier -- 107 against 90 us after 35 days, where the cold start's first days have 66 against 78): momentum tendencies of the sponge
    //      levels in place (below them dt_ug, dt_vg stay zero until the diffusion: not stored), its heating into work array 4
    for (int k = 0; k < nray; ++k) { dtu[k * s] = 0.0; dtv[k * s] = 0.0; r4[(size_t)k * s] = 0.0; }
    if (nray) moist::rayleigh_damping(a.ray, delta_t, a.pf_c + c, up, vp, s, dtu, dtv, s, r4, s);
    if (nroles == 2) {
      const double x[MOIST_NX] = {sf.flux_t, sf.flux_q, sf.flux_r, sf.flux_u, sf.flux_v, sf.dhdt_surf, sf.dedt_surf, sf.drdt_surf, sf.dhdt_atm,
                                  sf.dedq_atm, sf.dtaudu_atm, sf.dtaudv_atm, sf.u_star, sf.b_star, t_surf, net_sw, lw_down_surf, 0., 0., 0.};
#pragma unroll
      for (int i = 0; i < MOIST_NX; ++i) r0[(size_t)i * s] = x[i];
    }
    MT(2, 4) MT_STORE(2)
  }
  // ---- wavefront 0: convection (:862-880) and large-scale condensation on the convectively adjusted profile (:975-997).  The convection's
  //      deltas stay where the parcel was (LDS, or the private arrays): the condensation is their only reader, and it leaves
  //      (0 + conv_dt_tg) + cond_dt_tg in the same place for the diffusion below; dt_qg = (0 + conv) + cond goes to memory
  double ptp[LDSW ? 1 : LMAX], prp[LDSW ? 1 : LMAX];                    // work arrays in global memory: the parcel stays thread-private
  moist::QeParcel pc = LDSW ? moist::QeParcel{w0, w1, sw} : moist::QeParcel{ptp, prp, 1};
#if defined(MOIST_TIMING) && MOIST_TIMING == 5      // phase 5: inside the convection scheme (marks in moist_physics.h)
  pc.marks = mt_;
#endif
  if (role == 0) {
    double rain, cape, cin;
    int flag, klzb, klcl;
    moist::qe_moist_convection<LMAX, false, PHT>(a.sat, a.qe, L, delta_t, tp, qp, a.pf_p + c, php, s, pc.wTp, pc.wrp, rain, cape, cin, flag, klzb,
                                            klcl, nullptr, nullptr, pc.sw, pc);
    MT(1, 1)
#if defined(MOIST_TIMING) && MOIST_TIMING == 5
    { const long long t_ = wall_clock64(); for (int i_ = 6; i_ < 9; ++i_) mt_[i_] = t_; }
    MT_STORE(5)
#endif
    double precip = rain / delta_t;
    double rain_ls;
    moist::lscale_cond(a.sat, L,
                       [&](int k, double &t, double &q, double &ct, double &cq) {
                         ct = pc.wTp[k * pc.sw]; cq = pc.wrp[k * pc.sw];
                         t = ct + tp[k * s]; q = cq + qp[k * s];
                       },
                       a.pf_p + c, php, s,
                       [&](int k, double td, double qd, double ct, double cq) {
                         pc.wTp[k * pc.sw] = ct / delta_t + td / delta_t;
                         dtq[k * s] = cq / delta_t + qd / delta_t;
                       },
                       rain_ls);
    precip = precip + rain_ls / delta_t;
#if !defined(MOIST_TIMING) || (MOIST_TIMING != 2 && MOIST_TIMING != 5)
    if (a.precip) a.precip[c] = precip;
#endif
    MT(1, 2) MT_STORE(1)
  }
  // ---- the two wavefronts meet here; wavefront 0 goes on alone
  if (nroles == 2) {
    __syncthreads();
    if (role != 0) return;
    double x[MOIST_NX];
#pragma unroll
    for (int i = 0; i < MOIST_NX; ++i) x[i] = r0[(size_t)i * s];
    sf.flux_t = x[0]; sf.flux_q = x[1]; sf.flux_r = x[2]; sf.flux_u = x[3]; sf.flux_v = x[4]; sf.dhdt_surf = x[5]; sf.dedt_surf = x[6];
    sf.drdt_surf = x[7]; sf.dhdt_atm = x[8]; sf.dedq_atm = x[9]; sf.dtaudu_atm = x[10]; sf.dtaudv_atm = x[11]; sf.u_star = x[12];
    sf.b_star = x[13]; t_surf = x[14]; net_sw = x[15]; lw_down_surf = x[16];
  }
  MT(3, 0)
  // ---- dt_tg = ((conv + cond) + rad) + sponge, in that order (:880, :997, :1162, :1237), formed where it is read: by the boundary-layer depth
  //      (lowest levels only) and by the momentum diffusion's downward sweep, which reads each level's parts before its e, f overwrite them
  //      and stores the sum for the upward sweep.  dt_ug, dt_vg are zero below the sponge: known, not read.
  const int nr1 = max(nray - 1, 0);
  auto heat_in = [&](int k) {
    const double sp = r4[(size_t)min(k, nr1) * s];
    double x = pc.wTp[k * pc.sw] + w2[k * sw2];
    if (k < nray) x = x + sp;
    return x;
  };
  auto du_in = [&](int k) { const double x = dtu[(size_t)min(k, nr1) * s]; return (k < nray) ? x : 0.0; };
  auto dv_in = [&](int k) { const double x = dtv[(size_t)min(k, nr1) * s]; return (k < nray) ? x : 0.0; };
  MT(3, 1)
  // ---- boundary-layer diffusivities (:1242-1262), implicit vertical diffusion with the mixed layer (:1292-1330)
  {
    const double h = moist::pbl_depth_f(a.dif, L, delta_t, tp, up, vp, s, heat_in, du_in, dv_in, a.zf_c + c, a.zh_c + c, s);
    MT(3, 2)
    moist::PblProfile pbl;
    pbl.init(a.mo, a.dif, h, sf.u_star, sf.b_star, a.zh_c + c, s, L);
    const moist::VdiffWork w{w0, w1, w2, sw, sw2};
    moist::VdiffSurf S;
    double tau_u = sf.flux_u, tau_v = sf.flux_v;
    {
      const auto r = moist::vd::down_pair(L, delta_t, [&](int k) { return up[(size_t)k * s]; }, [&](int k) { return vp[(size_t)k * s]; }, du_in, dv_in,
                                          moist::PblProfile::Km{pbl}, tp, s, phc, a.zf_c + c, s, w,
                                          moist::DtPark<decltype(heat_in)>{heat_in, dtT, s});      // = vert_diff_momentum_f, its two halves
      MT(3, 3)
      moist::vert_diff_momentum_up_f(r, L, delta_t, up, vp, s, tau_u, tau_v, sf.dtaudu_atm, sf.dtaudv_atm, du_in, dv_in, dtu, dtv, dtT, s, nullptr, 0, w, S);
    }
    MT(3, 4)
    moist::vert_diff_heat_down(L, delta_t, tp, qp, s, moist::PblProfile::Kt{pbl}, phc, a.zf_c + c, s, dtT, dtq, s, w, S);
    MT(3, 5)
    moist::mixed_layer(a.ml, a.dt_atmos, t_surf, sf.flux_t, sf.flux_q, sf.flux_r, net_sw, lw_down_surf, S, sf.dhdt_surf, sf.dedt_surf,
                       sf.drdt_surf, sf.dhdt_atm, sf.dedq_atm);
    MT(3, 6)
    moist::vert_diff_up(L, delta_t, w, S, dtT, dtq, s);
  }
  a.t_surf[c] = t_surf;
  MT(3, 7) MT_STORE(3)
}

// mixed_layer_init with prescribe_initial_dist (mixed_layer.F90:455-460): t_surf = tconst - delta_T (3 sin^2 lat - 1)/3
__global__ void k_t_surf_init(int ncol, int I, const double *__restrict__ rad_lat_row, double tconst, double delta_T, double *t_surf) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncol) return;
  const double sl = sin(rad_lat_row[col / I]);
  t_surf[col] = tconst - delta_T * ((3. * (sl * sl)) - 1.) / 3.;
}

// ---------------------------------------------------------------------------------------------- host side
struct MoistState {
  moist::SatTableHost sat;
  moist::QeTablesHost qe;
  double *d_sat[3] = {nullptr, nullptr, nullptr}, *d_lcl = nullptr;
  moist::SatTable sat_dev;
  moist::QeParams qe_dev;
  moist::RayleighParams ray;
};

static void *dev_upload(const std::vector<double> &v) {
  void *p = nullptr;
  HIP_CHECK(hipMalloc(&p, v.size() * sizeof(double)));
  HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
  return p;
}

MoistState *moist_create(const isca_dyn_config &cfg, const Tables &tab) {
  MoistState *m = new MoistState();
  const isca_moist_config &mc = cfg.moist;
  m->sat.build();                                          // sat_vapor_pres_init, do_simple
  m->qe.build(m->sat.view(), mc.rhbm, mc.Tmin, mc.Tmax, mc.tau_bm, mc.val_inc);
  m->d_sat[0] = (double *)dev_upload(m->sat.tab); m->d_sat[1] = (double *)dev_upload(m->sat.dtab); m->d_sat[2] = (double *)dev_upload(m->sat.d2tab);
  m->d_lcl = (double *)dev_upload(m->qe.lcl);
  m->sat_dev = m->sat.view();
  m->sat_dev.tab = m->d_sat[0]; m->sat_dev.dtab = m->d_sat[1]; m->sat_dev.d2tab = m->d_sat[2];
  m->qe_dev = m->qe.params;
  m->qe_dev.lcl_temp_table = m->d_lcl;
  // damping_driver_init (damping_driver.f90:411-420) with the reference pressures of idealized_moist_phys_init (:620-629)
  const int L = cfg.num_levels;
  std::vector<double> lph, lpf;
  pressure_variables_1d(tab.pk, tab.bk, moist::PSTD_MKS, lph, lpf, cfg.vert_difference_option == 1);
  int best = 0;
  double bestv = INFINITY;
  for (int k = 0; k <= L; ++k) {
    const double pref = (k < L) ? std::exp(lpf[k]) : moist::PSTD_MKS;
    const double v = std::fabs(pref - 2 * mc.sponge_pbottom);
    if (v < bestv) { bestv = v; best = k; }
  }
  m->ray.nlev_rayfric = std::min(best + 1, L);
  m->ray.rfactr = (mc.trayfric > 0.0) ? (1. / mc.trayfric) : (1. / std::fabs(mc.trayfric)) * (1. / 86400.);
  m->ray.sponge_pbottom = mc.sponge_pbottom;
  m->ray.conserve_energy = mc.damping_conserve_energy != 0;
  return m;
}
void moist_destroy(MoistState *m) {
  if (!m) return;
  for (double *p : m->d_sat) if (p) (void)hipFree(p);
  if (m->d_lcl) (void)hipFree(m->d_lcl);
  delete m;
}

static MoistArgs moist_args(const isca_dyn &h) {
  const isca_moist_config &mc = h.cfg.moist;
  const MoistState &m = *h.moist;
  MoistArgs a{};
  a.L = h.g.L;
  a.albedo = mc.albedo_value;
  a.rough_mom = mc.roughness_mom; a.rough_heat = mc.roughness_heat; a.rough_moist = mc.roughness_moist;
  a.sat = m.sat_dev; a.qe = m.qe_dev;
  a.rad.solar_constant = mc.solar_constant; a.rad.del_sol = mc.del_sol; a.rad.del_sw = mc.del_sw; a.rad.ir_tau_eq = mc.ir_tau_eq;
  a.rad.ir_tau_pole = mc.ir_tau_pole; a.rad.atm_abs = mc.atm_abs; a.rad.odp = mc.odp; a.rad.sw_diff = mc.sw_diff;
  a.rad.linear_tau = mc.linear_tau; a.rad.wv_exponent = mc.wv_exponent; a.rad.solar_exponent = mc.solar_exponent; a.rad.diabatic_acce = 1.0;
  a.mo.rich_crit = mc.rich_crit; a.mo.drag_min = mc.drag_min;
  a.ray = m.ray; a.do_damping = mc.do_rayleigh;
  a.dif.frac_inner = mc.frac_inner; a.dif.rich_crit_pbl = mc.rich_crit_pbl;
  a.ml.heat_capacity = mc.depth * (1.035e3 * 3989.24495292815);       // depth*RHO_CP (mixed_layer.F90:514)
  a.ml.evaporation = mc.evaporation != 0; a.ml.ocean_qflux = 0.0;
  a.dt_atmos = h.cfg.dt_atmos;
  return a;
}
static void launch_moist_kernel(const MoistArgs &a, hipStream_t s) {
  const bool two = a.L + 1 >= MOIST_NX && !getenv("ISCA_MOIST_ONE_WAVE");       // two wavefronts per 64 columns (see the kernel)
  const dim3 grid((a.ncol + 63) / 64), block(two ? 128 : 64);
  const size_t lds1 = (size_t)64 * (a.L + 1) * sizeof(double);        // one work array of a block
  const bool glob = getenv("ISCA_MOIST_GLOBAL_WORK") != nullptr;
  int nlds = glob ? 0 : (3 * lds1 <= 65536 ? 3 : (2 * lds1 <= 65536 ? 2 : 0));            // L <= 41: all three; L <= 63: arrays 0 and 1 (parcel / deltas / e, f1)
  if (const char *e = getenv("ISCA_MOIST_LDS_ARRAYS")) nlds = std::min(nlds, atoi(e) >= 2 ? atoi(e) : 0);      // (tests: the variants of larger level counts at a small one)
#define LM(N)                                                                                          \
  do {                                                                                                 \
    if (a.pk) {                                                                                        \
      if (nlds == 3) hipLaunchKernelGGL((k_moist_physics<N, 3, true>), grid, block, 3 * lds1, s, a);      \
      else if (nlds == 2) hipLaunchKernelGGL((k_moist_physics<N, 2, true>), grid, block, 2 * lds1, s, a); \
      else hipLaunchKernelGGL((k_moist_physics<N, 0, true>), grid, block, 0, s, a);                       \
    } else {                                                                                           \
      if (nlds == 3) hipLaunchKernelGGL((k_moist_physics<N, 3, false>), grid, block, 3 * lds1, s, a);     \
      else if (nlds == 2) hipLaunchKernelGGL((k_moist_physics<N, 2, false>), grid, block, 2 * lds1, s, a);\
      else hipLaunchKernelGGL((k_moist_physics<N, 0, false>), grid, block, 0, s, a);                      \
    }                                                                                                  \
  } while (0)
  if (a.L <= 30) LM(32); else if (a.L <= 46) LM(48); else LM(64);
#undef LM
}

// physics of one step on the model state: previous-level fields, pressures of both levels, heights of the current one
// compute_pressures_and_heights (press_and_geopot.F90:363-387, flat surface, no virtual temperature) of both time levels
// (atmosphere.F90:296-303): the previous level needs pressures only (nothing reads its heights), the current one both.
struct PressArgs {
  const double *pk, *bk, *surf_geop;
  const double *t[2], *ps[2];
  double *p_full[2], *p_half[2], *z_full, *z_half;
  int ncol, L, store_half;
  int tl0;                 // first time level of the launch (1: the previous level's pressures are cached)
  int mcm;                 // vert_difference_option = 'mcm': p_full = the mean of the half levels (press_and_geopot.F90:196-200)
};
// Above 41 levels (the moist kernel's third work array in a global buffer, 131 072 columns at T170L60: the kernel is throughput-bound and every
// pass over a level array is 15 us of HBM time) the half-level pressures are not stored: k_moist_physics forms pk + bk ps where it needs them
// (T170L60: k_moist_pressures 131 -> 97 us, k_moist_physics 1227 -> 1187).  Up to 41 levels they are: at T85L40 the kernel is a latency chain
// that the scalar loads of pk, bk lengthen by what the 73 MB it no longer reads would have bought (161 -> 164 us).
static bool moist_sigma_half(int L) {      // ISCA_MOIST_PHALF=sigma|arrays: one or the other at any level count (tests, measurements)
  if (const char *e = getenv("ISCA_MOIST_PHALF")) return e[0] == 's';
  return (size_t)3 * 64 * (L + 1) * sizeof(double) > 65536;
}
// (1) level-parallel part: one thread per (column, group of MP_PK levels, time level): p_half, p_full, and for the current level the two
//     hydrostatic increments of each layer, left in z_full / z_half for the sum in k_moist_physics.  (p_half is stored only for the level counts whose k_moist_physics reads it: moist_sigma_half.)  (One thread per level computed every
//     log p_half twice -- the kernel is VALU-bound on log / exp / divide: 18 of its 23 us -- a group computes it once per half level.)
constexpr int MP_PK = 4;
__global__ __launch_bounds__(256) void k_moist_pressures(PressArgs a) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= a.ncol) return;
  const int k0 = blockIdx.y * MP_PK, tl = blockIdx.z + a.tl0, L = a.L;
  const size_t c = (size_t)col, s = (size_t)a.ncol;
  const double ps = a.ps[tl][c];
  const double *pk = a.pk, *bk = a.bk;
  const bool top0 = (pk[0] == 0.0 && bk[0] == 0.0);
  const int ktop = (pk[0] == 0.0) ? 1 : 0;
  double tk[MP_PK];
#pragma unroll
  for (int i = 0; i < MP_PK; ++i) tk[i] = (tl == 1) ? a.t[1][c + (size_t)min(k0 + i, L - 1) * s] : 0.0;
  double ph0 = pk[k0] + bk[k0] * ps;
  double l0 = (top0 && k0 == 0) ? 0.0 : log(ph0);
#pragma unroll
  for (int i = 0; i < MP_PK; ++i) {
    const int k = k0 + i;
    if (k < L) {
      const double ph1 = pk[k + 1] + bk[k + 1] * ps, l1 = log(ph1);
      double lf;
      if (a.mcm) lf = log(0.5 * (ph1 + ph0));
      else if (top0 && k == 0) lf = l1 - 1.0;
      else lf = l1 - (1.0 - ph0 * (l1 - l0) / (ph1 - ph0));
      a.p_full[tl][c + (size_t)k * s] = a.mcm ? 0.5 * (ph1 + ph0) : exp(lf);
      if (a.store_half) {
        a.p_half[tl][c + (size_t)k * s] = ph0;
        if (k == L - 1) a.p_half[tl][c + (size_t)L * s] = ph1;
      }
      if (tl == 1) {
        a.z_full[c + (size_t)k * s] = RDGAS * tk[i] * (l1 - lf);
        a.z_half[c + (size_t)k * s] = (k >= ktop) ? RDGAS * tk[i] * (l1 - l0) : 0.0;
      }
      ph0 = ph1; l0 = l1;
    }
  }
}
// the work buffer: two (p_full, p_half) areas, then z_full, z_half of the current level, ...
struct MoistWork { double *pf[2], *ph[2], *zf_c, *zh_c, *rest; };
static MoistWork moist_work_layout(const isca_dyn &h) {
  const size_t lev = (size_t)h.g.Jl * h.g.I;
  const int L = h.g.L;
  MoistWork w;
  w.pf[0] = h.d.moist_work; w.ph[0] = w.pf[0] + lev * L; w.pf[1] = w.ph[0] + lev * (L + 1); w.ph[1] = w.pf[1] + lev * L;
  w.zf_c = w.ph[1] + lev * (L + 1); w.zh_c = w.zf_c + lev * L;
  w.rest = w.zh_c + lev * (L + 1);
  return w;
}
void launch_moist_pressures(const isca_dyn &h, const StepScalars &sc, hipStream_t s, int slot_prev, int slot_cur, bool prev_cached) {
  const Dev &d = h.d;
  const size_t lev = (size_t)h.g.Jl * h.g.I;
  const MoistWork w = moist_work_layout(h);
  double *pf_p = w.pf[slot_prev], *ph_p = w.ph[slot_prev], *pf_c = w.pf[slot_cur], *ph_c = w.ph[slot_cur];
  double *zf_c = w.zf_c, *zh_c = w.zh_c;
  PressArgs a;
  a.pk = d.pk; a.bk = d.bk; a.ncol = (int)lev; a.L = h.g.L; a.surf_geop = d.surf_geop;
  a.t[0] = d.tg[sc.prev]; a.ps[0] = d.psg[sc.prev]; a.t[1] = d.tg[sc.cur]; a.ps[1] = d.psg[sc.cur];
  if (virtual_t_on(h)) {      // heights of the current level from its virtual temperature (atmosphere.F90:335-337: grid_tracers of that level)
    launch_virtual_t(h, d.tg[sc.cur], d.tr_atm[sc.cur], d.tv, s);
    a.t[1] = d.tv;
  }
  a.p_full[0] = pf_p; a.p_half[0] = ph_p; a.p_full[1] = pf_c; a.p_half[1] = ph_c; a.z_full = zf_c; a.z_half = zh_c;
  a.store_half = moist_sigma_half(h.g.L) ? 0 : 1;
  a.tl0 = prev_cached ? 1 : 0;
  a.mcm = h.cfg.vert_difference_option == 1;
  hipLaunchKernelGGL(k_moist_pressures, dim3((unsigned)((lev + 255) / 256), (h.g.L + MP_PK - 1) / MP_PK, prev_cached ? 1 : 2), dim3(256), 0, s, a);      // (the heights: summed by k_moist_physics)
}
void launch_moist_physics(const isca_dyn &h, const StepScalars &sc, hipStream_t s, int slot_prev, int slot_cur) {
  const Dev &d = h.d;
  const size_t lev = (size_t)h.g.Jl * h.g.I;
  const MoistWork w = moist_work_layout(h);
  double *pf_p = w.pf[slot_prev], *ph_p = w.ph[slot_prev], *pf_c = w.pf[slot_cur], *ph_c = w.ph[slot_cur];
  double *zf_c = w.zf_c, *zh_c = w.zh_c, *zf_p = w.rest, *zh_p = zf_p + lev * h.g.L;
  MoistArgs a = moist_args(h);
  a.ncol = (int)lev; a.I = h.g.I;
  a.up = d.ug[sc.prev]; a.vp = d.vg[sc.prev]; a.tp = d.tg[sc.prev]; a.qp = d.tr_atm[sc.prev];
  a.pf_p = pf_p; a.ph_p = ph_p; a.pf_c = pf_c; a.ph_c = ph_c; a.zf_c = zf_c; a.zh_c = zh_c;
  a.rad_lat_row = d.rad_lat_l; a.rad_lat_col = nullptr;
  a.surf_geop = d.surf_geop; a.ktop = (h.tab.pk[0] == 0.0) ? 1 : 0;
  if (moist_sigma_half(h.g.L)) { a.pk = d.pk; a.bk = d.bk; a.ps_p = d.psg[sc.prev]; a.ps_c = d.psg[sc.cur]; }
  a.t_surf = d.t_surf; a.dtu = d.ph_dtu; a.dtv = d.ph_dtv; a.dtT = d.ph_dtT; a.dtq = d.ph_dtq; a.precip = d.precip;
  a.work = zh_p + lev * (h.g.L + 1);
  a.delta_t = sc.delta_t;
  a.gust = h.phys_calls == 0 ? 1.0 : h.cfg.moist.constant_gust;    // gust = 1 until vert_turb_driver has run once (:592, :1262)
  launch_moist_kernel(a, s);
}
// the same on caller columns (device pointers, [lev][ncol])
void launch_moist_physics_on(const isca_dyn &h, int ncol, double delta_t, double gust, const double *rad_lat, const double *u, const double *v,
                             const double *t, const double *q, const double *ph_p, const double *pf_p, const double *ph_c, const double *pf_c,
                             const double *zh_c, const double *zf_c, double *t_surf, double *dtu, double *dtv, double *dtT, double *dtq,
                             double *precip, double *work, hipStream_t s) {
  MoistArgs a = moist_args(h);
  a.work = work;
  a.ncol = ncol; a.I = 1;
  a.up = u; a.vp = v; a.tp = t; a.qp = q; a.pf_p = pf_p; a.ph_p = ph_p; a.pf_c = pf_c; a.ph_c = ph_c; a.zf_c = const_cast<double *>(zf_c); a.zh_c = const_cast<double *>(zh_c);      // (heights given: not summed, not written)
  a.rad_lat_row = nullptr; a.rad_lat_col = rad_lat;
  a.t_surf = t_surf; a.dtu = dtu; a.dtv = dtv; a.dtT = dtT; a.dtq = dtq; a.precip = precip;
  a.delta_t = delta_t; a.gust = gust;
  launch_moist_kernel(a, s);
}
void launch_t_surf_init(const isca_dyn &h, hipStream_t s) {
  const int ncol = h.g.Jl * h.g.I;
  hipLaunchKernelGGL(k_t_surf_init, dim3((ncol + 255) / 256), dim3(256), 0, s, ncol, h.g.I, h.d.rad_lat_l, h.cfg.moist.tconst,
                     h.cfg.moist.delta_T, h.d.t_surf);
}
size_t moist_work_doubles(const Geom &g) { return (size_t)g.Jl * g.I * (size_t)(4 * g.L + 9 * (g.L + 1)); }

}  // namespace isca
