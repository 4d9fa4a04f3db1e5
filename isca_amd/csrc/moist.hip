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
  double *cc_dT, *cc_dq, *cc_precip;    // this step's (conv + cond) heating and moistening rates [L][ncol] and their rain rate [ncol]: written by k_moist_convcond or, a step ahead, by k_moist_physics
  const double *tn, *qn; double dt_next; double *nx_dT, *nx_dq, *nx_precip; int do_next;   // the NEXT step's convection inside k_moist_physics: its previous level's T, q (pressures: pf_c, ph_c), its delta_t, where its rates go
  const double *sig;                    // the convection's logarithm tables on pure sigma levels (moist_physics.h: QeParcel::sig), or null
  const double *pk, *bk, *ps_p, *ps_c;  // in the model's step (SIG kernels): the half-level pressures are pk + bk ps, formed where they are needed; ph_p / ph_c unused
  const double *surf_geop;              // non-null: zf_c / zh_c hold the hydrostatic increments of k_moist_pressures and this kernel sums them (moist_heights_scan)
  int ktop;
  double *work;                         // [5][L+1][ncol]: B's work arrays 0 and 1 (2 when it does not fit LDS), the sponge's heating (4)
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
  int full_sweeps;                      // ISCA_MOIST_FULL_SWEEPS: the implicit diffusion's sweeps over every level (not only the boundary layer's)
};

// Round 5: the convection runs a step AHEAD.
// qe_moist_convection and lscale_cond read only PREVIOUS-level fields (T, q and that level's pressures, idealized_moist_phys.F90:862-880, :975-997),
// and a step's previous level is the step before's current level, whose grid fields are final once THAT step's fixers have run.  So the convection
// of step n+1 does not depend on step n's dynamics -- nor on anything else step n's physics computes: in the kernel of step n it is the work of
// the second wavefront of each block of 64 columns, beside the chain that IS step n's:
//   wavefront B: grey radiation down and up, surface fluxes, then boundary layer + implicit diffusion with the mixed layer (this step; it reads the
//                (conv + cond) rates the kernel of the step before left in one of two buffer sets);
//   wavefront A: the height sum and the Rayleigh sponge of this step (flagged to B through LDS), then convection + condensation of the NEXT step
//                into the other buffer set.
// Rounds 1-4 ran convection -> condensation -> diffusion of one step as one chain per block (the radiation beside the first two): 178 us of
// dependent work at T85L40 once it rains, 194-197 us per launch.  Now the two chains are ~125-160 and ~137 us and independent.  The first step after a
// state write has nothing computed ahead: k_moist_convcond (the same column code, one wavefront per 64 columns) runs in front of it.
// Work arrays of L+1 levels per column: A's parcel (T, r) in LDS arrays 0 and 1; B's three (0: lw_down, then the diffusion's e; 1: lw_dtrans, then f_1;
// 2: sw_down, then the radiative heating, then f_2) -- array 2 in LDS, 0 and 1 in the global work area: B reads and writes them a chunk of levels at a
// time, beside loads it waits for anyway, and its sweeps cover the boundary layer only -- when 3 x 64 x (L+1) doubles fit the 64 KB a block may take
// without opting in (L <= 41); up to L = 63 the parcel is in LDS and B's arrays are global; beyond that everything is (the parcel thread-private).
// (Measured and not kept: a 166 K window of the saturation table in LDS instead of B's array -- the ascent's dependent lookups already hit the
// CU's L1, the kernel went 138 -> 157 us, HISTORY.md 9.)
// dt_tg = ((conv + cond) + rad) + sponge in the reference's order (:880, :997, :1162, :1237), formed where it is read.
// Phase timing for kernel experiments (tools/dev/moist_phase_times.py): built with -DMOIST_TIMING=p the kernels stamp wall_clock64 (10 ns
// ticks) at the marks of phase p (1: convection + condensation, 5: inside the convection scheme, 2: radiation + surface fluxes, 4: height sum + sponge,
// 3: boundary layer + diffusion) and store, in lane i of every wavefront, the time between marks i and i+1 in place of the precipitation.
#ifdef MOIST_TIMING
#define MT_DECL long long mt_[9]; for (int i_ = 0; i_ < 9; ++i_) mt_[i_] = wall_clock64();
#define MT(p, i) if (MOIST_TIMING == p) { const long long t_ = wall_clock64(); for (int i_ = i; i_ < 9; ++i_) mt_[i_] = t_; }
#define MT_STORE(p, dst) if (MOIST_TIMING == p) { long long d_ = 0; for (int i_ = 0; i_ < 8; ++i_) if ((lane & 7) == i_) d_ = mt_[i_ + 1] - mt_[i_]; (dst)[c] = (double)d_; }
#else
#define MT_DECL
#define MT(p, i)
#define MT_STORE(p, dst)
#endif
// The hydrostatic sum, bottom-up, of one column: z_full / z_half arrive as the layers' increments (k_moist_pressures) and leave as heights; 8 levels of
// increments requested together.  (Was a kernel of its own: 512 wavefronts, five memory round trips, 12 us.)
__device__ __forceinline__ void moist_heights_scan(double gh, int L, int ktop, double *z_full, double *z_half, size_t s) {
  z_half[(size_t)L * s] = gh / GRAV;
  constexpr int HU = 8;
  for (int k0 = L - 1; k0 >= 0; k0 -= HU) {
    double zf[HU], dz[HU];
#pragma unroll
    for (int i = 0; i < HU; ++i) {
      const int k = (k0 - i >= 0) ? k0 - i : 0;
      zf[i] = z_full[(size_t)k * s]; dz[i] = z_half[(size_t)k * s];
    }
#pragma unroll
    for (int i = 0; i < HU; ++i) {
      const int k = k0 - i;
      if (k >= 0) {
        z_full[(size_t)k * s] = (gh + zf[i]) / GRAV;
        if (k >= ktop) gh = gh + dz[i];
        z_half[(size_t)k * s] = (k >= ktop) ? gh / GRAV : 0.0;
      }
    }
  }
}
// convection (:862-880) and large-scale condensation on the convectively adjusted profile (:975-997) of ONE column: T, q, p_full, p_half of the
// previous time level of the step in question.  The convection's deltas stay where the parcel was (LDS, or the private arrays): the condensation is
// their only reader; (0 + conv_dt_tg) + cond_dt_tg, dt_qg = (0 + conv) + cond and the rain rate go to memory.
template <int LMAX, int TVM, class PHT>      // TVM: where the convection keeps the environment's virtual temperature (moist_physics.h: QeColumn)
__device__ __forceinline__ void moist_convcond_column(const MoistArgs &a, const moist::SatTable &sat, int L, int s, const double *tp, const double *qp, const double *pf, PHT php,
                                                      double delta_t, double *ccT, double *ccq, double &precip_out, moist::QeParcel &pc) {
  double rain, cape, cin;
  int flag, klzb, klcl;
  moist::qe_moist_convection<LMAX, false, PHT, TVM>(sat, a.qe, L, delta_t, tp, qp, pf, php, s, pc.wTp, pc.wrp, rain, cape, cin, flag, klzb, klcl,
                                                       nullptr, nullptr, pc.sw, pc);
  double precip = rain / delta_t;
  double rain_ls;
  moist::lscale_cond(sat, L,
                     [&](int k, double &t, double &q, double &ct, double &cq) {
                       ct = pc.wTp[k * pc.sw]; cq = pc.wrp[k * pc.sw];
                       t = ct + tp[k * s]; q = cq + qp[k * s];
                     },
                     pf, php, s,
                     [&](int k, double td, double qd, double ct, double cq) {
                       ccT[(size_t)k * s] = ct / delta_t + td / delta_t;
                       ccq[(size_t)k * s] = cq / delta_t + qd / delta_t;
                     },
                     rain_ls);
  precip_out = precip + rain_ls / delta_t;
}
// ---- convection + condensation alone, for a step that has nothing computed ahead (the first one, or after a state write), for
//      isca_idealized_moist_phys on caller columns and for ISCA_MOIST_NO_PIPELINE.  NLDS: 3 = parcel T, r and Tv in LDS; 2 = the parcel in LDS, Tv
//      thread-private; 0 = all thread-private.
template <int LMAX, int NLDS, bool SIG>
__global__ __launch_bounds__(64) void k_moist_convcond(MoistArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds_work[];
  const int lane = threadIdx.x & 63;
  const int col = min(blockIdx.x * 64 + lane, a.ncol - 1);     // the tail lanes redo the last column (same values stored)
  const int L = a.L, s = a.ncol;
  const size_t c = (size_t)col;
  using PHT = typename std::conditional<SIG, moist::PHalfSigma, const double *>::type;
  PHT php;                         // half-level pressures of the previous time level
  if constexpr (SIG) php = moist::PHalfSigma{a.pk, a.bk, a.ps_p[c]};
  else php = a.ph_p + c;
  MT_DECL
  double ptp[NLDS >= 2 ? 1 : LMAX], prp[NLDS >= 2 ? 1 : LMAX];
  moist::QeParcel pc = NLDS >= 2 ? moist::QeParcel{lds_work + lane, lds_work + lane + (size_t)(L + 1) * 64, 64} : moist::QeParcel{ptp, prp, 1};
  if (NLDS == 3) pc.wTv = lds_work + lane + (size_t)2 * (L + 1) * 64;
  pc.sig = a.sig;
#if defined(MOIST_TIMING) && MOIST_TIMING == 5      // phase 5: inside the convection scheme (marks in moist_physics.h)
  pc.marks = mt_;
#endif
  double precip;
  moist_convcond_column<LMAX, NLDS == 3 ? 1 : 2, PHT>(a, a.sat, L, s, a.tp + c, a.qp + c, a.pf_p + c, php, a.delta_t, a.cc_dT + c, a.cc_dq + c, precip, pc);
  a.cc_precip[c] = precip;
  MT(1, 1) MT_STORE(1, a.cc_precip) MT_STORE(5, a.cc_precip)
}

// ---- the physics of one step (idealized_moist_phys.F90:1054-1340 with this step's (conv + cond) rates from a.cc_*) and, with a.do_next, the convection
//      + condensation of the next one (its previous level = this step's current one: a.tn, a.qn with the pressures a.pf_c / a.ph_c) into a.nx_*
template <int LMAX, int NLDS, bool SIG>      // NLDS: 3 = the parcel and B's array 2 in LDS; 2 = the parcel; 0 = nothing.  SIG: p_half from (pk, bk, ps)
__global__ __launch_bounds__(128) void k_moist_physics(MoistArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds_work[];
  const int lane = threadIdx.x & 63, role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nroles = blockDim.x >> 6;
  const int col = min(blockIdx.x * 64 + lane, a.ncol - 1);     // the tail lanes redo the last column (same values stored)
  const int L = a.L, s = a.ncol;
  const size_t c = (size_t)col;
  const moist::SatTable &sat = a.sat;
  // B's work arrays: 0 and 1 in the global work area, 2 in LDS behind the parcel's two when it fits
  const int sw = a.ncol, sw2 = (NLDS == 3) ? 64 : a.ncol;
  double *w0 = a.work + c;
  double *w1 = w0 + (size_t)(L + 1) * sw;
  double *w2 = (NLDS == 3) ? lds_work + lane + (size_t)2 * (L + 1) * 64 : w0 + (size_t)2 * (L + 1) * sw;
  double *r4 = a.work + c + (size_t)4 * (L + 1) * s;           // the sponge's heating (its levels only)
  const double *tp = a.tp + c, *qp = a.qp + c, *up = a.up + c, *vp = a.vp + c;
  using PHT = typename std::conditional<SIG, moist::PHalfSigma, const double *>::type;
  PHT phc;                         // half-level pressures of the current time level
  if constexpr (SIG) phc = moist::PHalfSigma{a.pk, a.bk, a.ps_c[c]};
  else phc = a.ph_c + c;
  double *dtu = a.dtu + c, *dtv = a.dtv + c, *dtT = a.dtT + c, *dtq = a.dtq + c;
  const double *ccT = a.cc_dT + c, *ccq = a.cc_dq + c;
  const double delta_t = a.delta_t;
  const int nray = a.do_damping ? a.ray.nlev_rayfric : 0;
  MT_DECL
  // ---- wavefront A: convection + condensation of the NEXT step (nothing of this step's: the two wavefronts share no data)
  if (role == 0 && a.do_next) {
    double ptp[NLDS >= 2 ? 1 : LMAX], prp[NLDS >= 2 ? 1 : LMAX];
    moist::QeParcel pc = NLDS >= 2 ? moist::QeParcel{lds_work + lane, lds_work + lane + (size_t)(L + 1) * 64, 64} : moist::QeParcel{ptp, prp, 1};
    pc.sig = a.sig;
#if defined(MOIST_TIMING) && MOIST_TIMING == 5
    pc.marks = mt_;
#endif
    double precip;
    moist_convcond_column<LMAX, 2, PHT>(a, sat, L, s, a.tn + c, a.qn + c, a.pf_c + c, phc, a.dt_next, a.nx_dT + c, a.nx_dq + c, precip, pc);
    a.nx_precip[c] = precip;
    MT(1, 1) MT_STORE(1, a.precip) MT_STORE(5, a.precip)
  }
  if (nroles == 2 && role == 0) return;
  // ---- wavefront B (the only one of a one-wavefront block): this step.  The height sum (first reader: the surface fluxes)
  if (a.surf_geop) moist_heights_scan(a.surf_geop[c], L, a.ktop, a.zf_c + c, a.zh_c + c, (size_t)s);
  MT(2, 1)
  // ---- wavefront B: grey radiation down (:1054-1061) and up (:1156-1162; it needs the surface temperature, not the surface fluxes): the heating lands
  //      in work array 2, where the downward shortwave flux was
  double t_surf = a.t_surf[c];
  double net_sw, lw_down_surf;
  {
    const double lat = a.rad_lat_col ? a.rad_lat_col[c] : a.rad_lat_row[col / a.I];
    double insolation, sw_tau_0;
    if (NLDS == 3 && a.rad.atm_abs == 0.0) {
      // no shortwave absorption (the scheme's default): the downward shortwave flux is the insolation at every half level and is not stored; the
      // downward longwave flux takes its place in LDS instead of work array 0 in global memory (21 MB less per launch at T85L40)
      moist::gray_rad_down(a.rad, L, lat, a.albedo, tp, phc, s, w2, w1, sw, (double *)nullptr, 0, insolation, sw_tau_0, net_sw, lw_down_surf, sw2, false);
      MT(2, 2)
      moist::gray_rad_up(a.rad, L, a.albedo, t_surf, tp, phc, s, w2, w1, sw, (const double *)nullptr, 0, w2, sw2, true, sw2, true, insolation);
    } else {
      moist::gray_rad_down(a.rad, L, lat, a.albedo, tp, phc, s, w0, w1, sw, w2, sw2, insolation, sw_tau_0, net_sw, lw_down_surf);
      MT(2, 2)
      moist::gray_rad_up(a.rad, L, a.albedo, t_surf, tp, phc, s, w0, w1, sw, w2, sw2, w2, sw2, true);
    }
    MT(2, 3)
  }
  // ---- surface fluxes (:1077-1153)
  moist::SurfFlux sf;
  {
    const size_t low = (size_t)(L - 1) * s;
    moist::surface_flux(sat, a.mo, tp[low], qp[low], up[low], vp[low], a.pf_c[c + low], a.zf_c[c + low], moist::ph_at(phc, s, L), t_surf,
                        a.rough_mom, a.rough_heat, a.rough_moist, a.rough_mom, a.gust, sf);
  }
  MT(2, 4)
  // ---- the Rayleigh sponge (:1228-1237): momentum tendencies of the sponge levels in place (below them dt_ug, dt_vg stay zero until the diffusion:
  //      not stored), its heating into work array 4
  if (nray) {
    if (a.ray.conserve_energy) moist::rayleigh_damping(a.ray, delta_t, a.pf_c + c, up, vp, s, dtu, dtv, s, r4, s, true);
    else {
      for (int k = 0; k < nray; ++k) r4[(size_t)k * s] = 0.0;
      moist::rayleigh_damping(a.ray, delta_t, a.pf_c + c, up, vp, s, dtu, dtv, s, r4, s, true);
    }
  }
  MT(2, 5) MT_STORE(2, a.precip)
#if !defined(MOIST_TIMING)
  if (a.precip) a.precip[c] = a.cc_precip[c];
#endif
  MT(3, 0)
  // ---- dt_tg = ((conv + cond) + rad) + sponge, in that order (:880, :997, :1162, :1237), formed where it is read: by the boundary-layer depth
  //      (lowest levels only) and by the momentum diffusion's downward sweep, which reads each level's parts before its e, f overwrite them
  //      and stores the sum for the upward sweep.  dt_ug, dt_vg are zero below the sponge: known, not read.  Two forms of each: with the sponge's
  //      terms (the test of the level becomes a branch with the load behind it: one memory round trip per LEVEL, found in the ISA) for the top of the
  //      column, and without them for everything at or below level nray -- where the boundary layer is.
  const int nr1 = max(nray - 1, 0);
  auto heat_sp = [&](int k) {
    const double sp = r4[(size_t)min(k, nr1) * s];
    double x = ccT[(size_t)k * s] + w2[k * sw2];
    if (k < nray) x = x + sp;
    return x;
  };
  auto du_sp = [&](int k) { const double x = dtu[(size_t)min(k, nr1) * s]; return (k < nray) ? x : 0.0; };
  auto dv_sp = [&](int k) { const double x = dtv[(size_t)min(k, nr1) * s]; return (k < nray) ? x : 0.0; };
  auto heat_lo = [&](int k) { return ccT[(size_t)k * s] + w2[k * sw2]; };
  auto zero_lo = [](int) { return 0.0; };
  auto dq_in = [&](int k) { return ccq[(size_t)k * s]; };
  MT(3, 1)
  // ---- boundary-layer diffusivities (:1242-1262), implicit vertical diffusion with the mixed layer (:1292-1330)
  int kstop;
  const double h = moist::pbl_depth_f2(a.dif, L, delta_t, tp, up, vp, s, heat_sp, du_sp, dv_sp, heat_lo, zero_lo, zero_lo, nray, a.zf_c + c, a.zh_c + c, s, &kstop);
  // No interface at or above level kstop carries diffusion (the depth lies below that level's height): the four sweeps of the implicit diffusion
  // run over the boundary layer only -- from the highest kstop of the wavefront's 64 columns down, typically a quarter of the column -- and the
  // levels above take their tendencies in one streaming pass (moist_physics.h: vert_diff_passthrough).  a.full_sweeps (ISCA_MOIST_FULL_SWEEPS): all levels.
  int kb = kstop;
#pragma unroll
  for (int off = 32; off; off >>= 1) kb = min(kb, __shfl_xor(kb, off));
  kb = __builtin_amdgcn_readfirstlane(a.full_sweeps ? 0 : min(kb, L - 2));
  MT(3, 2)
  const int ksp = min(nray, kb);
  moist::vert_diff_passthrough(0, ksp, du_sp, dv_sp, heat_sp, dq_in, dtu, dtv, dtT, dtq, s);        // the sponge levels
  moist::vert_diff_passthrough(ksp, kb, zero_lo, zero_lo, heat_lo, dq_in, dtu, dtv, dtT, dtq, s);    // between the sponge and the boundary layer
  MT(3, 3)
  moist::PblProfile pbl;
  pbl.init(a.mo, a.dif, h, sf.u_star, sf.b_star, a.zh_c + c, s, L);
  const moist::VdiffWork w{w0, w1, w2, sw, sw2};
  moist::VdiffSurf S;
  double tau_u = sf.flux_u, tau_v = sf.flux_v;
  auto momentum = [&](auto heat_in, auto du_in, auto dv_in) {
    const auto r = moist::vd::down_pair(L, delta_t, [&](int k) { return up[(size_t)k * s]; }, [&](int k) { return vp[(size_t)k * s]; }, du_in, dv_in,
                                        moist::PblProfile::Km{pbl}, tp, s, phc, a.zf_c + c, s, w,
                                        moist::DtPark<decltype(heat_in)>{heat_in, dtT, s}, kb);      // = vert_diff_momentum_f, its two halves
    MT(3, 4)
    moist::vert_diff_momentum_up_f(r, L, delta_t, up, vp, s, tau_u, tau_v, sf.dtaudu_atm, sf.dtaudv_atm, du_in, dv_in, dtu, dtv, dtT, s, nullptr, 0, w, S, kb);
  };
  if (kb >= nray) momentum(heat_lo, zero_lo, zero_lo);       // (the boundary layer lies below the sponge)
  else momentum(heat_sp, du_sp, dv_sp);
  MT(3, 5)
  moist::vert_diff_heat_down(L, delta_t, tp, qp, s, moist::PblProfile::Kt{pbl}, phc, a.zf_c + c, s, dtT, ccq, s, w, S, kb);
  MT(3, 6)
  moist::mixed_layer(a.ml, a.dt_atmos, t_surf, sf.flux_t, sf.flux_q, sf.flux_r, net_sw, lw_down_surf, S, sf.dhdt_surf, sf.dedt_surf,
                     sf.drdt_surf, sf.dhdt_atm, sf.dedq_atm);
  MT(3, 7)
  moist::vert_diff_up(L, delta_t, w, S, dtT, dtq, s, kb);
  a.t_surf[c] = t_surf;
  MT(3, 8) MT_STORE(3, a.precip)
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
  double *d_sig = nullptr;
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
  // QeParcel::sig: with pk = 0 (and not 'mcm') p_half = b p_s and p_full = cf p_s with ln cf(k) = ln b(k+1) - (1 - b(k) a / (b(k+1) - b(k))),
  // a = ln(b(k+1) / b(k)) (press_and_geopot.F90:165-194; the top layer of a model top at zero pressure: ln cf = ln b(2) - 1, and its ratio of half
  // levels is infinite as log(p_half(2) / 0) is): what k_moist_pressures computes per column, here once and in extended precision
  {
    bool sigma = cfg.vert_difference_option != 1;
    for (double v : tab.pk) if (v != 0.0) sigma = false;
    if (sigma && !getenv("ISCA_MOIST_LOG_PER_LEVEL")) {
      const int L = cfg.num_levels;
      std::vector<long double> lcf(L);
      std::vector<double> sg((size_t)3 * L, 0.0);
      for (int k = 0; k < L; ++k) {
        const long double b0 = tab.bk[k], b1 = tab.bk[k + 1];
        if (b0 == 0.0L) { lcf[k] = logl(b1) - 1.0L; sg[L + k] = INFINITY; }
        else { const long double a = logl(b1 / b0); lcf[k] = logl(b1) - (1.0L - b0 * a / (b1 - b0)); sg[L + k] = (double)a; }
      }
      for (int k = 0; k < L; ++k) {
        if (k + 1 < L) sg[k] = (double)(lcf[k] - lcf[k + 1]);
        sg[2 * L + k] = (double)(lcf[k] - lcf[L - 1]);
      }
      m->d_sig = (double *)dev_upload(sg);
    }
  }
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
  if (m->d_sig) (void)hipFree(m->d_sig);
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
  a.full_sweeps = exp_env("ISCA_MOIST_FULL_SWEEPS") ? 1 : 0;
  return a;
}
// LDS per block: one work array is 64 x (L+1) doubles.  ISCA_MOIST_LDS_ARRAYS=0|2 (test hook): the variants of larger
// level counts at a small one; (experiments build) ISCA_MOIST_CC_LDS=0|2|3: the convection kernel's alone (its LDS footprint beside the dynamics kernels it runs under).
static int moist_nlds(int L, const char *own_env) {
  const size_t lds1 = (size_t)64 * (L + 1) * sizeof(double);
  int nlds = 3 * lds1 <= 65536 ? 3 : (2 * lds1 <= 65536 ? 2 : 0);      // L <= 41: all three; L <= 63: arrays 0 and 1
  if (const char *e = getenv("ISCA_MOIST_LDS_ARRAYS")) nlds = std::min(nlds, atoi(e) >= 2 ? atoi(e) : 0);       // test hook: 0 (all in global memory) | 2 | 3
  if (own_env) if (const char *e = exp_env(own_env)) nlds = std::min(nlds, atoi(e) >= 2 ? atoi(e) : 0);
  return nlds;
}
static void launch_moist_convcond_kernel(const MoistArgs &a, hipStream_t s) {
  const dim3 grid((a.ncol + 63) / 64), block(64);
  const size_t lds1 = (size_t)64 * (a.L + 1) * sizeof(double);
  const int nlds = moist_nlds(a.L, "ISCA_MOIST_CC_LDS");
#define LM(N)                                                                                            \
  do {                                                                                                   \
    if (a.pk) {                                                                                          \
      if (nlds == 3) hipLaunchKernelGGL((k_moist_convcond<N, 3, true>), grid, block, 3 * lds1, s, a);      \
      else if (nlds == 2) hipLaunchKernelGGL((k_moist_convcond<N, 2, true>), grid, block, 2 * lds1, s, a); \
      else hipLaunchKernelGGL((k_moist_convcond<N, 0, true>), grid, block, 0, s, a);                       \
    } else {                                                                                             \
      if (nlds == 3) hipLaunchKernelGGL((k_moist_convcond<N, 3, false>), grid, block, 3 * lds1, s, a);     \
      else if (nlds == 2) hipLaunchKernelGGL((k_moist_convcond<N, 2, false>), grid, block, 2 * lds1, s, a);\
      else hipLaunchKernelGGL((k_moist_convcond<N, 0, false>), grid, block, 0, s, a);                      \
    }                                                                                                    \
  } while (0)
  if (a.L <= 46) LM(48); else LM(64);      // (a 32-level instantiation existed until round 6: 225 KB of code per variant for nothing measurable at T21L25 / T42L25)
#undef LM
}
static void launch_moist_physics_kernel(const MoistArgs &a, hipStream_t s) {
  const bool two = !exp_env("ISCA_MOIST_ONE_WAVE");       // two wavefronts per 64 columns (see the kernel)
  const dim3 grid((a.ncol + 63) / 64), block(two ? 128 : 64);
  const size_t lds1 = (size_t)64 * (a.L + 1) * sizeof(double);
  const int nlds = moist_nlds(a.L, nullptr);
#define LM(N)                                                                                           \
  do {                                                                                                  \
    if (a.pk) {                                                                                         \
      if (nlds == 3) hipLaunchKernelGGL((k_moist_physics<N, 3, true>), grid, block, 3 * lds1, s, a);      \
      else if (nlds == 2) hipLaunchKernelGGL((k_moist_physics<N, 2, true>), grid, block, 2 * lds1, s, a); \
      else hipLaunchKernelGGL((k_moist_physics<N, 0, true>), grid, block, 0, s, a);                       \
    } else {                                                                                            \
      if (nlds == 3) hipLaunchKernelGGL((k_moist_physics<N, 3, false>), grid, block, 3 * lds1, s, a);     \
      else if (nlds == 2) hipLaunchKernelGGL((k_moist_physics<N, 2, false>), grid, block, 2 * lds1, s, a);\
      else hipLaunchKernelGGL((k_moist_physics<N, 0, false>), grid, block, 0, s, a);                      \
    }                                                                                                   \
  } while (0)
  if (a.L <= 46) LM(48); else LM(64);      // (a 32-level instantiation existed until round 6: 225 KB of code per variant for nothing measurable at T21L25 / T42L25)
#undef LM
}

// physics of one step on the model state: previous-level fields, pressures of both levels, heights of the current one
// compute_pressures_and_heights (press_and_geopot.F90:363-387, flat surface, no virtual temperature) of both time levels
// (atmosphere.F90:296-303): the previous level needs pressures only (nothing reads its heights), the current one both.
struct PressArgs {
  const double *pk, *bk, *surf_geop;
  const double *t[2], *ps[2];
  double *p_full[2], *p_half[2], *z_full, *z_half;
  // lazy fixers (Dev::m_t): the level's T, q, p_s with the pending mass factor, temperature correction and water factor applied, into the level's slot
  const double *q[2], *pend[2];
  double *t_out[2], *q_out[2], *ps_out[2];
  const int *kmask; int mbyte[2], lazy;
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
  double ps = a.ps[tl][c];
  const double *pk = a.pk, *bk = a.bk;
  const bool top0 = (pk[0] == 0.0 && bk[0] == 0.0);
  const int ktop = (pk[0] == 0.0) ? 1 : 0;
  double tk[MP_PK];
#pragma unroll
  for (int i = 0; i < MP_PK; ++i) tk[i] = (tl == 1 || a.lazy) ? a.t[tl][c + (size_t)min(k0 + i, L - 1) * s] : 0.0;
  if (a.lazy) {       // what k_fixer_apply would have left in the stored fields (compute_corrections, spectral_dynamics.F90:1213-1283), formed here instead
    double qk[MP_PK];
#pragma unroll
    for (int i = 0; i < MP_PK; ++i) qk[i] = a.q[tl][c + (size_t)min(k0 + i, L - 1) * s];
    const double fac = a.pend[tl][PEND_FACTOR], tc = a.pend[tl][PEND_TCORR], wfac = a.pend[tl][PEND_WFAC];
    const int km = (a.kmask[c] >> (8 * a.mbyte[tl])) & 0xff;        // levels above water_correction_limit in the step that made this level
    ps = ps * fac;
    if (blockIdx.y == 0) a.ps_out[tl][c] = ps;
#pragma unroll
    for (int i = 0; i < MP_PK; ++i) {
      const int k = k0 + i;
      tk[i] = tk[i] + tc;
      if (k < L) { a.t_out[tl][c + (size_t)k * s] = tk[i]; a.q_out[tl][c + (size_t)k * s] = qk[i] * ((k >= km) ? wfac : 1.0); }
    }
  }
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
  a.lazy = h.lazy_fix ? 1 : 0;
  if (a.lazy) {         // (between steps the mask word's byte 0 belongs to the current level, byte 1 to the previous one: launch_fixer_materialize)
    a.q[0] = d.tr_atm[sc.prev]; a.q[1] = d.tr_atm[sc.cur]; a.pend[0] = d.pend + 4 * sc.prev; a.pend[1] = d.pend + 4 * sc.cur;
    a.t_out[0] = d.m_t[slot_prev]; a.q_out[0] = d.m_q[slot_prev]; a.ps_out[0] = d.m_ps[slot_prev];
    a.t_out[1] = d.m_t[slot_cur]; a.q_out[1] = d.m_q[slot_cur]; a.ps_out[1] = d.m_ps[slot_cur];
    a.kmask = d.kmask; a.mbyte[0] = 1; a.mbyte[1] = 0;
  }
  a.mcm = h.cfg.vert_difference_option == 1;
  hipLaunchKernelGGL(k_moist_pressures, dim3((unsigned)((lev + 255) / 256), (h.g.L + MP_PK - 1) / MP_PK, prev_cached ? 1 : 2), dim3(256), 0, s, a);      // (the heights: summed by k_moist_physics)
}
// convection + condensation of the step whose PREVIOUS level is time level `level` (its pressures in slot pslot of the work area), leapfrog step
// delta_t, into buffer set ccslot (idealized_moist_phys.F90:862-880, :975-997)
void launch_moist_convcond(const isca_dyn &h, int level, int pslot, double delta_t, int ccslot, hipStream_t s) {
  const Dev &d = h.d;
  const size_t lev = (size_t)h.g.Jl * h.g.I;
  const MoistWork w = moist_work_layout(h);
  MoistArgs a = moist_args(h);
  a.ncol = (int)lev; a.I = h.g.I;
  a.tp = h.lazy_fix ? d.m_t[pslot] : d.tg[level]; a.qp = h.lazy_fix ? d.m_q[pslot] : d.tr_atm[level];
  a.pf_p = w.pf[pslot]; a.ph_p = w.ph[pslot];
  if (moist_sigma_half(h.g.L)) { a.pk = d.pk; a.bk = d.bk; a.ps_p = h.lazy_fix ? d.m_ps[pslot] : d.psg[level]; }
  a.sig = h.moist->d_sig;      // (the model's own pressures: k_moist_pressures)
  a.cc_dT = d.cc_dT[ccslot]; a.cc_dq = d.cc_dq[ccslot]; a.cc_precip = d.cc_precip[ccslot];
  a.delta_t = delta_t;
  launch_moist_convcond_kernel(a, s);
}
// the physics of one step on the model state: previous-level fields, pressures and heights of the current one, the convection's rates from buffer
// set ccslot; next: also the NEXT step's convection + condensation (previous level = this step's current one, delta_t = 2 dt_atmos) into the other set
void launch_moist_physics(const isca_dyn &h, const StepScalars &sc, hipStream_t s, int slot_cur, int ccslot, bool next) {
  const Dev &d = h.d;
  const size_t lev = (size_t)h.g.Jl * h.g.I;
  const MoistWork w = moist_work_layout(h);
  double *zf_p = w.rest, *zh_p = zf_p + lev * h.g.L;
  MoistArgs a = moist_args(h);
  a.ncol = (int)lev; a.I = h.g.I;
  a.up = d.ug[sc.prev]; a.vp = d.vg[sc.prev]; a.tp = d.tg[sc.prev]; a.qp = d.tr_atm[sc.prev];
  if (h.lazy_fix) { a.tp = d.m_t[1 - slot_cur]; a.qp = d.m_q[1 - slot_cur]; }       // (the previous level's slot is the other one)
  a.pf_c = w.pf[slot_cur]; a.ph_c = w.ph[slot_cur]; a.zf_c = w.zf_c; a.zh_c = w.zh_c;
  a.rad_lat_row = d.rad_lat_l; a.rad_lat_col = nullptr;
  a.surf_geop = d.surf_geop; a.ktop = (h.tab.pk[0] == 0.0) ? 1 : 0;
  if (moist_sigma_half(h.g.L)) { a.pk = d.pk; a.bk = d.bk; a.ps_c = h.lazy_fix ? d.m_ps[slot_cur] : d.psg[sc.cur]; }
  a.t_surf = d.t_surf; a.dtu = d.ph_dtu; a.dtv = d.ph_dtv; a.dtT = d.ph_dtT; a.dtq = d.ph_dtq; a.precip = d.precip;
  a.cc_dT = d.cc_dT[ccslot]; a.cc_dq = d.cc_dq[ccslot]; a.cc_precip = d.cc_precip[ccslot];
  a.do_next = next ? 1 : 0; a.sig = h.moist->d_sig;
  a.tn = d.tg[sc.cur]; a.qn = d.tr_atm[sc.cur]; a.dt_next = 2 * h.cfg.dt_atmos;
  if (h.lazy_fix) { a.tn = d.m_t[slot_cur]; a.qn = d.m_q[slot_cur]; }
  a.nx_dT = d.cc_dT[1 - ccslot]; a.nx_dq = d.cc_dq[1 - ccslot]; a.nx_precip = d.cc_precip[1 - ccslot];
  a.work = zh_p + lev * (h.g.L + 1);
  a.delta_t = sc.delta_t;
  a.gust = h.phys_calls == 0 ? 1.0 : h.cfg.moist.constant_gust;    // gust = 1 until vert_turb_driver has run once (:592, :1262)
  launch_moist_physics_kernel(a, s);
}
// the same on caller columns (device pointers, [lev][ncol]); work: 5 (L+1) ncol doubles, cc: (2 L + 1) ncol doubles
void launch_moist_physics_on(const isca_dyn &h, int ncol, double delta_t, double gust, const double *rad_lat, const double *u, const double *v,
                             const double *t, const double *q, const double *ph_p, const double *pf_p, const double *ph_c, const double *pf_c,
                             const double *zh_c, const double *zf_c, double *t_surf, double *dtu, double *dtv, double *dtT, double *dtq,
                             double *precip, double *work, double *cc, hipStream_t s) {
  MoistArgs a = moist_args(h);
  a.work = work;
  a.ncol = ncol; a.I = 1;
  a.up = u; a.vp = v; a.tp = t; a.qp = q; a.pf_p = pf_p; a.ph_p = ph_p; a.pf_c = pf_c; a.ph_c = ph_c; a.zf_c = const_cast<double *>(zf_c); a.zh_c = const_cast<double *>(zh_c);      // (heights given: not summed, not written)
  a.rad_lat_row = nullptr; a.rad_lat_col = rad_lat;
  a.t_surf = t_surf; a.dtu = dtu; a.dtv = dtv; a.dtT = dtT; a.dtq = dtq; a.precip = precip;
  a.cc_dT = cc; a.cc_dq = cc + (size_t)h.g.L * ncol; a.cc_precip = cc + (size_t)2 * h.g.L * ncol;
  a.delta_t = delta_t; a.gust = gust;
  launch_moist_convcond_kernel(a, s);
  launch_moist_physics_kernel(a, s);
}
void launch_t_surf_init(const isca_dyn &h, hipStream_t s) {
  const int ncol = h.g.Jl * h.g.I;
  hipLaunchKernelGGL(k_t_surf_init, dim3((ncol + 255) / 256), dim3(256), 0, s, ncol, h.g.I, h.d.rad_lat_l, h.cfg.moist.tconst,
                     h.cfg.moist.delta_T, h.d.t_surf);
}
size_t moist_work_doubles(const Geom &g) { return (size_t)g.Jl * g.I * (size_t)(4 * g.L + 9 * (g.L + 1)); }

}  // namespace isca
