// C-ABI of the MI355X spectral core (see include/isca_dyn.h for the reference interfaces replaced).
#include "kernels.h"
#include "moist.h"
#include <cstring>
#include <unistd.h>
#include <sys/stat.h>
#include <ctime>
#include <cmath>
#include <algorithm>
#include <mutex>
#include <cstdlib>
#include <memory>

using namespace isca;

static thread_local std::string g_last_error;
extern "C" const char *isca_last_error(void) { return g_last_error.c_str(); }
void isca_internal_set_error(const std::string &m) { g_last_error = m; }      // for the library's other compile units (restart_nc.cpp)

#define API_BEGIN try {
#define API_END                                   \
  }                                               \
  catch (const std::exception &e) {               \
    g_last_error = e.what();                      \
    return 1;                                     \
  }                                               \
  catch (...) {                                   \
    g_last_error = "unknown error";               \
    return 1;                                     \
  }                                               \
  return 0;

static void fail(const std::string &m) { throw std::runtime_error(m); }

// ---------------------------------------------------------------------------------------------------
// kernel timing (HIP events on the handle's stream)
// ---------------------------------------------------------------------------------------------------
struct Timed {
  isca_dyn *h;
  int id = -1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st;
  Timed(isca_dyn *h_, const char *name, hipStream_t st_ = nullptr, bool segment = false) : h(h_), st(st_ ? st_ : h_->stream) {
    if (!h->timer.enabled || h->timer.segments != segment) return;
    auto &t = h->timer;
    for (size_t i = 0; i < t.names.size(); ++i)
      if (t.names[i] == name) id = (int)i;
    if (id < 0) { id = (int)t.names.size(); t.names.push_back(name); t.ms.push_back(0); t.calls.push_back(0); }
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
  }
  ~Timed() {
    if (id < 0) return;
    hipEventRecord(e1, st);
    h->timer.ev.push_back(e0); h->timer.ev.push_back(e1); h->timer.ev_name.push_back(id);
  }
};
static void timer_collect(isca_dyn *h) {
  auto &t = h->timer;
  if (t.ev.empty()) return;
  hipStreamSynchronize(h->stream);
  if (h->stream2) hipStreamSynchronize(h->stream2);
  for (size_t i = 0; i < t.ev_name.size(); ++i) {
    float ms = 0;
    hipEventElapsedTime(&ms, t.ev[2 * i], t.ev[2 * i + 1]);
    t.ms[t.ev_name[i]] += ms; t.calls[t.ev_name[i]] += 1;
    hipEventDestroy(t.ev[2 * i]); hipEventDestroy(t.ev[2 * i + 1]);
  }
  t.ev.clear(); t.ev_name.clear();
}

// ---------------------------------------------------------------------------------------------------
template <typename T>
static T *dalloc(isca_dyn *h, size_t n, bool zero = true) {
  T *p = nullptr;
  HIP_CHECK(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)));
  if (zero) HIP_CHECK(hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)));
  h->allocs.push_back(p);
  return p;
}
template <typename T>
static T *dupload(isca_dyn *h, const std::vector<T> &v) {
  T *p = dalloc<T>(h, v.size(), false);
  if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

extern "C" int isca_dyn_config_default(isca_dyn_config *c) {
  if (!c) return 1;
  std::memset(c, 0, sizeof(*c));
  // spectral_dynamics.F90:152-206 defaults overridden by held_suarez_test_case.py:45-98 (T21)
  c->lon_max = 64; c->lat_max = 32; c->num_fourier = 21; c->num_spherical = 22; c->num_levels = 25;
  c->fourier_inc = 1; c->triang_trunc = 1; c->dt_atmos = 600.0;
  c->damping_order = 4; c->damping_coeff = 1.15740741e-4;
  c->robert_coeff = 0.04; c->raw_filter_coeff = 1.0; c->alpha_implicit = 0.5;
  c->reference_sea_level_press = 1.0e5; c->scale_heights = 6.0; c->exponent = 7.5; c->surf_res = 0.5;
  c->do_mass_correction = 1; c->do_energy_correction = 1; c->do_water_correction = 1;
  c->water_correction_limit = 200.e2; c->initial_temperature = 264.0; c->initial_sphum = 0.0;
  c->valid_range_t[0] = 100.; c->valid_range_t[1] = 800.; c->num_tracers = 1;
  c->t_zero = 315.; c->t_strat = 200.; c->delh = 60.; c->delv = 10.; c->eps = 0.; c->sigma_b = 0.7;
  c->ka = -40.; c->ks = -4.; c->kf = -1.; c->do_conserve_energy = 1; c->trflux = 1.e-5; c->trsink = -4.; c->P00 = 1.e5;
  c->rank = 0; c->world_size = 1; c->device = 0; c->stream = nullptr; c->legendre_impl = 0;
  // moist package (physics = 1): module defaults overridden by frierson_test_case.py:49-170
  c->physics = 0; c->vert_coord_input = 0;
  for (int k = 0; k < ISCA_MAX_TRACERS; ++k) { c->tracer_spectral[k] = 0; c->tracer_robert_coeff[k] = -1.0; }
  // hs_forcing_nml's local heating (hs_forcing.F90:87-94): off, with the module's default shape
  c->local_heating_option = 0; c->local_heating_srfamp = 0.0; c->local_heating_xwidth = 10.; c->local_heating_ywidth = 10.;
  c->local_heating_xcenter = 180.; c->local_heating_ycenter = 45.; c->local_heating_vert_decay = 1.e4;
  c->use_virtual_temperature = 0;
  c->vert_advect_uv = 0; c->vert_advect_t = 0; c->use_implicit = 1; c->make_symmetric = 0;
  c->vert_difference_option = 0;
  for (int k = 0; k < ISCA_MAX_TRACERS; ++k) { c->tracer_hole_filling[k] = 0; c->tracer_sms[k] = 0; c->tracer_flux[k] = 0.0; c->tracer_sink[k] = 0.0; c->tracer_advect_vert[k] = -1; }
  c->damping_option = 0; c->cutoff_wn = 15; c->damping_coeff_vor = c->damping_coeff_div = -1.0; c->damping_order_vor = c->damping_order_div = -1;
  isca_moist_config &m = c->moist;
  m.roughness_mom = m.roughness_heat = m.roughness_moist = 3.21e-05;
  m.solar_constant = 1360.0; m.del_sol = 1.4; m.del_sw = 0.0; m.ir_tau_eq = 6.0; m.ir_tau_pole = 1.5; m.atm_abs = 0.2; m.odp = 1.0;
  m.sw_diff = 0.0; m.linear_tau = 0.1; m.wv_exponent = 4.0; m.solar_exponent = 4.0;
  m.depth = 2.5; m.tconst = 285.; m.delta_T = 40.; m.albedo_value = 0.31; m.evaporation = 1;
  m.tau_bm = 7200.; m.rhbm = 0.7; m.Tmin = 160.; m.Tmax = 350.; m.val_inc = 0.01;
  m.do_rayleigh = 1; m.trayfric = -0.25; m.sponge_pbottom = 5000.; m.damping_conserve_energy = 1;
  m.constant_gust = 0.0; m.frac_inner = 0.1; m.rich_crit_pbl = 1.0; m.rich_crit = 2.0; m.drag_min = 1.e-05;
  c->radius = RADIUS_EARTH; c->omega = OMEGA_EARTH;
  return 0;
}

static const double PEND_ROWS[12] = {1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0};    // Dev::pend with nothing pending
// Lazy fixers: apply what is pending on the two time levels in place, so that the stored tg, psg, tr, tr_atm are the model's values
// (host reads and writes of the state, restart files, diagnostics, the transforms of complete_update / refresh_derived).
// the scalars of the last step's fixers, if their computation was left to the next column kernel (core.h: fin_deferred), now
static void flush_finish(isca_dyn *h) {
  if (!h->fin_deferred) return;
  h->fin_deferred = false;
  StepScalars sc{}; sc.prev = h->fin_prev; sc.cur = h->fin_cur; sc.fut = h->fin_fut;
  launch_fixer_finish(*h, sc, h->stream);
}
static void materialize(isca_dyn *h) {
  flush_finish(h);
  const bool any = h->thermo_pending[0] || h->thermo_pending[1] || h->tr_state[0] != isca::TR_MAT || h->tr_state[1] != isca::TR_MAT;
  if (!any) return;
  if (h->in_step) fail("the model state cannot be read or written in the middle of a step driven phase by phase");
  launch_fixer_materialize(*h, h->stream);
  HIP_CHECK(hipMemcpyAsync(h->d.pend, PEND_ROWS, sizeof(PEND_ROWS), hipMemcpyHostToDevice, h->stream));
  h->thermo_pending[0] = h->thermo_pending[1] = false;
  h->tr_state[0] = h->tr_state[1] = isca::TR_MAT;
}
static bool is_lazy_field(const std::string &nm) {
  return nm == "tg" || nm == "psg" || nm == "tr" || nm == "tr_atm" || nm == "p_full" || nm == "p_half" || nm == "z_full" || nm == "z_half";
}

static void deal_wavenumbers(int M1, int P, std::vector<int> &m_of_slot, int &Ml) {
  // boustrophedon over ranks: round r deals m = r*P .. r*P+P-1 left-to-right (r even) or right-to-left
  Ml = (M1 + P - 1) / P;
  m_of_slot.assign((size_t)P * Ml, -1);
  for (int idx = 0; idx < P * Ml; ++idx) {
    const int r = idx / P, pos = idx % P;
    const int q = (r % 2 == 0) ? pos : P - 1 - pos;
    if (idx < M1) m_of_slot[(size_t)q * Ml + r] = idx;
  }
}
extern "C" int isca_wavenumber_dealing(int num_fourier, int world_size, int *m_of_slot, int *m_local) {
  API_BEGIN
  if (num_fourier < 0 || world_size < 1 || !m_of_slot || !m_local) fail("invalid argument");
  std::vector<int> v; int Ml;
  deal_wavenumbers(num_fourier + 1, world_size, v, Ml);
  std::memcpy(m_of_slot, v.data(), v.size() * sizeof(int));
  *m_local = Ml;
  API_END
}

static void check_config(const isca_dyn_config &c) {
  // check_dynamics_nml (spectral_dynamics.F90:666-755) + what this implementation supports
  if (c.num_fourier <= 0 || c.num_spherical <= 0 || c.num_levels <= 0) fail("invalid resolution");
  if (c.fourier_inc <= 0) fail(std::to_string(c.fourier_inc) + " is an invalid value for fourier_inc.");
  if (c.num_spherical != c.num_fourier * c.fourier_inc + 1) fail("num_spherical must equal num_fourier * fourier_inc + 1");
  if (c.lon_max < 3 * c.num_fourier + 1) fail("number of longitude points is too small for number of fourier waves");
  if (2 * c.lat_max < (c.triang_trunc ? 3 : 5) * (c.num_spherical - 1) + 1) fail("number of latitude points is too small for number of meridional waves");
  {  // fft99's set99 (fft99.F90:83-120): an even length whose half has no prime factor above 5
    int n = c.lon_max;
    if (n < 16 || n > 512 || (n & 1)) fail("lon_max must be even, between 16 and 512");
    n /= 2;
    for (int f : {2, 3, 5}) while (n % f == 0) n /= f;
    if (n != 1) fail("lon_max / 2 must have no prime factor above 5 (fft99's set99)");
  }
  if (c.lat_max % 8) fail("lat_max must be a multiple of 8");
  if (c.num_levels > 64) fail("num_levels must be <= 64 (one wavefront lane per level in the spectral update)");
  if (!(c.raw_filter_coeff > 0.0 && c.raw_filter_coeff <= 1.0)) fail("raw_filter_coeff must be in (0, 1]");
  if (c.robert_coeff < 0. || c.robert_coeff > 1.) fail("invalid robert_coeff");
  if (c.damping_order < 0 || c.damping_coeff < 0.) fail("invalid damping");
  if (c.damping_option < 0 || c.damping_option > 2)
    fail("spectral_damping_init: damping_option must be 0 'resolution_dependent', 1 'exponential_cutoff' or 2 'resolution_independent'");
  if (c.damping_option == 1 && (c.cutoff_wn < 0 || c.cutoff_wn >= c.num_spherical - 1)) fail("spectral_damping_init: cutoff_wn outside the truncation");
  if ((c.do_energy_correction) && !c.do_mass_correction) fail("energy_correction requires mass_correction");
  if (c.world_size < 1 || c.rank < 0 || c.rank >= c.world_size) fail("invalid rank/world_size");
  if (c.lat_max % c.world_size) fail("lat_max must be divisible by world_size (spec_mpp.F90:69-75)");
  if (((c.lat_max / c.world_size) * c.lon_max) % 64) fail("local columns must be a multiple of 64");
  if (c.dt_atmos <= 0) fail("dt_atmos has not been specified");
  if (c.num_levels > ISCA_MAX_LEVELS) fail("num_levels exceeds ISCA_MAX_LEVELS");
  if (c.vert_coord_input) {
    for (int k = 0; k < c.num_levels; ++k)
      if (!(c.pk_input[k + 1] + c.bk_input[k + 1] * c.reference_sea_level_press > c.pk_input[k] + c.bk_input[k] * c.reference_sea_level_press))
        fail("vert_coordinate_nml: pk/bk must give increasing half-level pressures");
  }
  if (c.vert_advect_uv < 0 || c.vert_advect_uv > 3)
    fail("spectral_dynamics_init: \"" + std::to_string(c.vert_advect_uv) + "\" is not a valid value for vert_advect_uv.");
  if (c.vert_advect_t < 0 || c.vert_advect_t > 3)
    fail("spectral_dynamics_init: \"" + std::to_string(c.vert_advect_t) + "\" is not a valid value for vert_advect_t.");
  if (c.vert_difference_option < 0 || c.vert_difference_option > 1)      // press_and_geopot.F90:216-219
    fail("pressure_variables: \"" + std::to_string(c.vert_difference_option) + "\" is not a valid value for vert_difference_option");
  if (!(c.radius > 0.0)) fail("constants_nml: radius must be positive");
  if (c.physics < 0 || c.physics > 2) fail("physics must be 0 (hs_forcing), 1 (idealized_moist_phys) or 2 (tendencies supplied by the caller)");
  if (c.num_tracers < 0 || c.num_tracers > ISCA_MAX_TRACERS) fail("num_tracers must be 0.." + std::to_string(ISCA_MAX_TRACERS));
  for (int k = 0; k < c.num_tracers; ++k)
    if (c.tracer_advect_vert[k] < -1 || c.tracer_advect_vert[k] > 3)
      fail("spectral_dynamics_init: tracer_advect_vert must be -1 (the representation's standard scheme) or 0..3 (second_centered, fourth_centered, "
           "van_leer_linear, finite_volume_parabolic): any other advect_vert is invalid");
  if (c.local_heating_option != 0 && c.local_heating_option != 1)
    fail("hs_forcing_nml: local_heating_option must be 0 ('': none) or 1 ('Isidoro'); 'from_file' is not a supported value (interpolator_mod's data files)");
  if (c.local_heating_option == 1 && c.physics != 0) fail("hs_forcing_nml: local_heating_option belongs to hs_forcing (physics = 0)");
  if (c.local_heating_option == 1 && (c.local_heating_xwidth == 0. || c.local_heating_ywidth == 0. || c.local_heating_vert_decay == 0.))
    fail("hs_forcing_nml: local_heating_xwidth, local_heating_ywidth and local_heating_vert_decay must not be zero");
  if (c.num_tracers > 0 && c.tracer_spectral[0] != 0)
    fail("spectral_dynamics_init: the first tracer of the field_table (the humidity the water fixer and the physics know) is a 'grid' tracer here: "
         "numerical_representation 'spectral' is not a supported value for it");
  if (c.num_tracers > 1) {
    if (c.raw_filter_coeff != 1.0) fail("more than one tracer: raw_filter_coeff must be 1");
    for (int k = 1; k < c.num_tracers; ++k) {
      if (c.tracer_spectral[k] != 0 && c.tracer_spectral[k] != 1)
        fail("spectral_dynamics_init: tracer_spectral must be 0 ('grid') or 1 ('spectral'): any other numerical_representation is invalid");
      if (c.tracer_robert_coeff[k] > 1.0) fail("tracer_robert_coeff must be <= 1 (negative = robert_coeff)");
    }
  }
  if (c.physics == 1) {
    if (c.num_tracers < 1) fail("idealized_moist_phys needs the sphum tracer (num_tracers >= 1)");
    if (c.num_levels < 3 || c.num_levels > 62) fail("idealized_moist_phys: num_levels must be in 3..62");
    const isca_moist_config &m = c.moist;
    if (m.frac_inner <= 0. || m.frac_inner >= 1.) fail("diffusivity_init: frac_inner must be between 0 and 1");
    if (m.rich_crit_pbl < 0.) fail("diffusivity_init: rich_crit_pbl must be greater than or equal to zero");
    if (m.rich_crit <= 0.25) fail("monin_obukhov_init: rich_crit must be greater than 0.25");
    if (m.drag_min < 0.0) fail("monin_obukhov_init: drag_min must be >= 0.0");
    if (m.depth <= 0. || m.Tmin >= m.Tmax || m.val_inc <= 0.) fail("invalid moist physics parameters");
  }
}

// Legendre kernels with rectangular bounds (every n of every wavenumber) instead of the triangle n <= num_spherical - m they have built in:
// rhomboidal truncation, and fourier_inc /= 1 (the triangle is then n <= num_spherical - m * fourier_inc; what lies outside is zero anyway)
static int rect_bounds(const isca_dyn *h) { return (h->cfg.triang_trunc && h->cfg.fourier_inc == 1) ? 0 : 1; }
static void build_field_lists(isca_dyn *h) {
  const int L = h->g.L;
  Dev &d = h->d;
  FieldList &f = h->fl_fwd;
  f.nf = 5;
  double *fg[5] = {d.g_dtu, d.g_dtv, d.g_dtT, d.g_E, d.g_dtlp};
  int off = 0;
  for (int i = 0; i < 5; ++i) { f.g[i] = fg[i]; f.nlev[i] = (i < 4) ? L : 1; f.off[i] = off; f.op[i] = OP_NONE; off += f.nlev[i]; }
  f.ncol = off;
  h->Cf = col_pitch(f.ncol);
}
// inverse batch targets depend on the time level that receives the new state
static FieldList inverse_list(isca_dyn *h, int tl) {
  const int L = h->g.L;
  Dev &d = h->d;
  FieldList f;
  f.nf = 10;
  double *ig[10] = {d.divg, d.vorg, d.ug[tl], d.vg[tl], d.tg[tl], d.dxT, d.dyT, d.psg[tl], d.dxlp, d.dylp};
  const int ops[10] = {OP_NONE, OP_NONE, OP_COSM, OP_COSM, OP_NONE, OP_COSM, OP_COSM, OP_EXP, OP_COSM, OP_COSM};
  int off = 0;
  for (int i = 0; i < 10; ++i) { f.g[i] = ig[i]; f.nlev[i] = (i < 7) ? L : 1; f.off[i] = off; f.op[i] = ops[i]; off += f.nlev[i]; }
  f.ncol = off;
  if (h->dx_fourier) {
    // d/dx of T and of ln p_s are i m / a times their Fourier coefficients (compute_gradient_cos' x part, spherical.F90:270-301; coef_dx = m fourier_inc / a),
    // which the batch holds anyway: no Legendre synthesis, no Fourier rows and (sharded) no exchange volume of their own -- 6 L + 2 level-fields instead of
    // 7 L + 3.  Buffer columns: div, vor, u, v, dT/dy, T, ln p_s, d ln p_s/dy.  Rows of the list: the same order with dT/dx's rows ALTERNATING with T's (both
    // read T's coefficients: one work item, mostly one load instruction) and d ln p_s/dx next to ln p_s -- so rows and buffer columns run in the same order and
    // the rows in front of the T block map to the same column number: an item's run of columns starts where it did (128-byte lines), which a block of
    // derivative rows anywhere else in the list would shift for everything behind it (measured at T170L60: +88 MB of re-read lines per launch).
    double *gp[10] = {d.divg, d.vorg, d.ug[tl], d.vg[tl], d.dyT, d.tg[tl], d.dxT, d.psg[tl], d.dxlp, d.dylp};
    const int op2[10] = {OP_NONE, OP_NONE, OP_COSM, OP_COSM, OP_COSM, OP_NONE, OP_COSM, OP_EXP, OP_COSM, OP_COSM};
    const int off2[10] = {0, L, 2 * L, 3 * L, 4 * L, 5 * L, 5 * L, 7 * L, 7 * L + 1, 7 * L + 2};
    const int boff[10] = {0, L, 2 * L, 3 * L, 4 * L, 5 * L, 5 * L, 6 * L, 6 * L, 6 * L + 1};
    for (int i = 0; i < 10; ++i) { f.g[i] = gp[i]; f.op[i] = op2[i]; f.off[i] = off2[i]; f.nlev[i] = (i < 7) ? L : 1; f.boff[i] = boff[i]; f.dx[i] = (i == 6 || i == 8) ? 1 : 0; }
    f.nbuf = 6 * L + 2;
    f.dxfac = (double)h->cfg.fourier_inc / h->cfg.radius;
    f.il_a = 5; f.il_b = 6;
  }
  return f;
}

extern "C" int isca_dyn_destroy(isca_dyn_t *h) {
  if (!h) return 0;
  timer_collect(h);
  if (h->hist) isca_history_destroy(h);          // (files not closed by isca_dyn_diag_close: every record written so far is complete)
  if (h->comm) { hipStreamSynchronize(h->stream); delete h->comm; h->comm = nullptr; }
  for (void *p : h->allocs) hipFree(p);
  moist_destroy(h->moist);
  if (h->stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_fork0) hipEventDestroy(h->ev_fork0);
  if (h->ev_join) hipEventDestroy(h->ev_join);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  if (h->host_red) hipHostFree(h->host_red);
  delete h;
  return 0;
}

static void upload_wave_matrices(isca_dyn *h, double delta_t) {
  if (h->wave_dt == delta_t) return;
  if (h->tab.damping_exponential) {                // spectral_damping.F90:186-190: the effective coefficients of a step of this length
    const Geom &g = h->g;
    std::vector<double> e[3], cf((size_t)3 * g.Ml * g.N1, 0.0);
    h->tab.damping_effective(delta_t, e[0], e[1], e[2]);
    for (int id = 0; id < 3; ++id)
      for (int ml = 0; ml < g.Ml; ++ml) {
        const int m = h->h_m_local[ml];
        if (m < 0) continue;
        for (int n = 0; n < g.N1; ++n) cf[((size_t)id * g.Ml + ml) * g.N1 + n] = e[id][(size_t)n * g.M1 + m];
      }
    HIP_CHECK(hipMemcpyAsync(h->d.coef + (size_t)10 * g.Ml * g.N1, cf.data(), cf.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
  }
  h->tab.build_wave_matrices(h->cfg, delta_t);     // implicit.F90:260-264: rebuilt when dt changes
  const int L = h->g.L, nw = h->tab.n_wave;
  std::vector<double> wt((size_t)nw * L * L);
  for (int w = 0; w < nw; ++w)
    for (int k = 0; k < L; ++k)
      for (int k2 = 0; k2 < L; ++k2) wt[((size_t)w * L + k2) * L + k] = h->tab.wave_matrix[((size_t)w * L + k) * L + k2];
  HIP_CHECK(hipMemcpyAsync(h->d.wave_mat_t, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
  h->wave_dt = delta_t;
}

// valid_range_t (spectral_dynamics.F90:940-972, FATAL 'temperatures out of valid range'): the fixer kernels keep the
// running extremes of every new temperature field since the last check; looked at whenever the host synchronises.
static void reset_valid_range(isca_dyn *h) {
  const double init[2] = {INFINITY, -INFINITY};
  HIP_CHECK(hipMemcpy(h->d.red + 20, init, sizeof(init), hipMemcpyHostToDevice));
}
// The synchronisation point of a run of steps: the extremes come back through a pinned buffer and are reset by copies QUEUED behind the last
// kernel, so the host waits once (two blocking default-stream copies here cost 50-80 us per call -- 2 % of a 20-step window at T85L40).
static void sync_and_check_valid_range(isca_dyn *h) {
  if (!h->host_red) {
    HIP_CHECK(hipHostMalloc((void **)&h->host_red, 64 * sizeof(double)));
    h->host_red[32] = INFINITY; h->host_red[33] = -INFINITY;
  }
  flush_finish(h);                                     // (the check below looks at the fixers' scalars)
  HIP_CHECK(hipMemcpyAsync(h->host_red, h->d.red, 26 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(hipMemcpyAsync(h->d.red + 20, h->host_red + 32, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (h->comm) { h->comm->synchronize(h->stream); h->comm->check(); }      // (RCCL: polls the stream, the communicator's error state and a deadline, comm.h)
  else HIP_CHECK(hipStreamSynchronize(h->stream));
  const double *red = h->host_red;
  // (kernels.hip k_column_sig: a block gave up waiting for block 0's scalars -- the dispatcher did not start block 0 first, or the device was taken away
  // for seconds.  Part of the verdict below, so that on a sharded run every rank raises at this point.)
  const bool fin_lost = red[25] != 0.0;
  if (fin_lost) HIP_CHECK(hipMemsetAsync(h->d.red + 25, 0, sizeof(double), h->stream));
  const double tmin = red[20], tmax = red[21];
  const bool stepped = !(tmin > tmax);                       // (no step since the last check: nothing to judge)
  const bool range_bad = stepped && (!(tmin >= h->cfg.valid_range_t[0] && tmax <= h->cfg.valid_range_t[1]) || !std::isfinite(red[16]) || !std::isfinite(red[17]));
  const bool bad = fin_lost || range_bad;
  // Sharded with the library's communicator: a rank judges its own band, but error_mesg(..., FATAL) stops EVERY PE (spectral_dynamics.F90:940-972)
  // -- the verdict is summed over the ranks, so that all of them raise at this synchronisation point instead of the others going on into an
  // exchange their peer never joins (every rank reaches this point after the same number of steps: the step loop is collective).
  bool other = false;
  if (h->comm && h->g.P > 1) {
    h->host_red[40] = bad ? 1.0 : 0.0;
    HIP_CHECK(hipMemcpyAsync(h->d.red + 24, h->host_red + 40, sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->comm->all_reduce_sum(h->d.red + 24, 1, h->stream);
    HIP_CHECK(hipMemcpyAsync(h->host_red + 41, h->d.red + 24, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    h->comm->synchronize(h->stream);
    other = !bad && h->host_red[41] > 0.0;
  }
  if (range_bad) {
    char msg[160];
    snprintf(msg, sizeof(msg), "temperatures out of valid range (min %.3f, max %.3f, valid %.1f..%.1f)", tmin, tmax,
             h->cfg.valid_range_t[0], h->cfg.valid_range_t[1]);
    fail(msg);
  }
  if (fin_lost) fail("column kernel: the deferred fixer scalars of the step before never arrived (block 0 did not run first); results since the last synchronisation are invalid");
  if (other) fail("temperatures out of valid range on another rank's latitude band");
}

extern "C" int isca_dyn_create(const isca_dyn_config *cfg, isca_dyn_t **out) {
  isca_dyn *h = nullptr;
  try {
    if (!cfg || !out) fail("null argument");
    check_config(*cfg);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
      fail("no HIP device available: the MI355X spectral core has no CPU fallback");
    HIP_CHECK(hipSetDevice(cfg->device));
    h = new isca_dyn();
    h->cfg = *cfg;
    if (cfg->stream) { h->stream = (hipStream_t)cfg->stream; h->own_stream = false; }
    else { HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    h->tab.build(*cfg);
    Tables &T = h->tab;
    Geom &g = h->g;
    g.I = T.I; g.J = T.J; g.M1 = T.M1; g.N1 = T.N1; g.L = T.L;
    g.P = cfg->world_size; g.rank = cfg->rank; g.Jl = g.J / g.P; g.j0 = g.rank * g.Jl; g.Jh = g.J / 2;
    g.Ml = (g.M1 + g.P - 1) / g.P;
    g.NHP = (((g.N1 + 1) / 2) + 15) / 16 * 16;
    g.log2I = 0; while ((1 << g.log2I) < g.I) ++g.log2I;
    g.log2Jl = -1;
    if ((g.Jl & (g.Jl - 1)) == 0) { g.log2Jl = 0; while ((1 << g.log2Jl) < g.Jl) ++g.log2Jl; }
    if (g.M1 > g.I / 2) fail("num_fourier too large for lon_max");
    // ---- wavenumber dealing: boustrophedon over ranks for triangular load balance
    { int Ml; deal_wavenumbers(g.M1, g.P, h->h_m_of_slot, Ml); }
    h->h_slot_of_m.assign(g.M1, 0);
    for (int sl = 0; sl < g.P * g.Ml; ++sl)
      if (h->h_m_of_slot[sl] >= 0) h->h_slot_of_m[h->h_m_of_slot[sl]] = sl;
    h->h_m_local.assign(g.Ml, -1);
    for (int ml = 0; ml < g.Ml; ++ml) h->h_m_local[ml] = h->h_m_of_slot[(size_t)g.rank * g.Ml + ml];
    h->ml_of_m0 = -1;
    for (int ml = 0; ml < g.Ml; ++ml) if (h->h_m_local[ml] == 0) h->ml_of_m0 = ml;
    Dev &d = h->d;
    std::memset(&d, 0, sizeof(d));
    d.m_of_slot = dupload(h, h->h_m_of_slot);
    d.slot_of_m = dupload(h, h->h_slot_of_m);
    d.m_local = dupload(h, h->h_m_local);
    // ---- latitude tables of the local band
    auto band = [&](const std::vector<double> &v) { return std::vector<double>(v.begin() + g.j0, v.begin() + g.j0 + g.Jl); };
    d.cosm_lat_l = dupload(h, band(T.cosm_lat)); d.wts_lat_l = dupload(h, band(T.wts_lat));
    d.coriolis_l = dupload(h, band(T.coriolis)); d.sin_lat_l = dupload(h, band(T.sin_lat));
    d.rad_lat_l = dupload(h, band(T.rad_lat)); d.wts_lat_g = dupload(h, T.wts_lat);
    if (cfg->local_heating_option == 1) {      // hs_forcing_init (:373-384) + local_heating's factors of longitude and latitude (:751-758), srfamp folded into the first
      const double pi = 3.14159265358979323846, twopi = 2. * pi;
      const double xw = cfg->local_heating_xwidth * pi / 180., yw = cfg->local_heating_ywidth * pi / 180.;
      double xc = cfg->local_heating_xcenter * pi / 180.;
      xc = xc - twopi * std::floor(xc / twopi);           // "make sure xcenter falls in the range zero to 2*PI" (hs_forcing.F90:378-380)
      const double yc = cfg->local_heating_ycenter * pi / 180.;
      const double srfamp = cfg->local_heating_srfamp / 86400.;
      std::vector<double> fx(g.I), fy(T.rad_lat.size());
      for (int i = 0; i < g.I; ++i) {
        double lon = T.deg_lon[i] * pi / 180.;
        lon = lon - twopi * std::floor(lon / twopi);
        const double z = (lon - xc) / xw;
        fx[i] = srfamp * std::exp(-.5 * (z * z));
      }
      for (size_t j = 0; j < fy.size(); ++j) { const double z = (T.rad_lat[j] - yc) / yw; fy[j] = std::exp(-.5 * (z * z)); }
      d.lh_lon = dupload(h, fx); d.lh_lat_l = dupload(h, band(fy));
    }
    // ---- Legendre tables for the local wavenumbers, parity-split (spherical_fourier.F90:376-394)
    {
      std::vector<double> pw((size_t)g.Ml * 2 * g.Jh * g.NHP, 0.0), pi((size_t)g.Ml * 2 * g.NHP * g.Jh, 0.0);
      for (int ml = 0; ml < g.Ml; ++ml) {
        const int m = h->h_m_local[ml];
        if (m < 0) continue;
        for (int n = 0; n < g.N1; ++n) {
          const int par = n & 1, nh = n >> 1;
          for (int jp = 0; jp < g.Jh; ++jp) {
            const double p = T.legendre[((size_t)jp * g.N1 + n) * g.M1 + m];
            pw[(((size_t)ml * 2 + par) * g.Jh + jp) * g.NHP + nh] = p * T.wts_hem[jp];
            pi[(((size_t)ml * 2 + par) * g.NHP + nh) * g.Jh + jp] = p;
          }
        }
      }
      d.pw_fwd = dupload(h, pw);
      d.p_inv = dupload(h, pi);
      if (legendre_mfma_ok(g, cfg->legendre_impl)) {
        std::vector<double> ff, fi, sc;
        build_legendre_fragments(g, T, h->h_m_local, ff, fi, sc);
        d.leg_fwd_frag = dupload(h, ff); d.leg_inv_frag = dupload(h, fi); d.leg_scoef = dupload(h, sc);
      }
#ifdef ISCA_EXPERIMENTS
      // fused FFT + Legendre analysis of the step (one rank, lon_max = 256, triangular truncation): ISCA_FUSE_FFT_LEG=1 -- correct, 45 MB less traffic,
      // 1.7x slower than the two kernels it replaces (round 4, HISTORY.md)
      if (exp_env("ISCA_FUSE_FFT_LEG") && atoi(exp_env("ISCA_FUSE_FFT_LEG")) != 0 && fused_forward_ok(g) && cfg->triang_trunc && cfg->fourier_inc == 1) {
        std::vector<double> fz; std::vector<int> dz;
        const int nt = build_fused_fwd_tables(g, T, h->h_m_local, fz, dz);
        if (nt) { d.fz_frag = dupload(h, fz); d.fz_desc = dupload(h, dz); d.fz_NT = nt; h->fuse_fwd = true; }
      }
#endif
    }
    {  // coefficient tables per local m
      const std::vector<double> *src[13] = {&T.eigen, &T.coef_uvm, &T.coef_uvc, &T.coef_uvp, &T.coef_alpm, &T.coef_alpp,
                                            &T.coef_dym, &T.coef_dx, &T.coef_dyp, &T.tri_mask, &T.damping, &T.damping_vor, &T.damping_div};
      std::vector<double> cf((size_t)13 * g.Ml * g.N1, 0.0);     // rows 10-12 (damping): refreshed per delta_t with 'exponential_cutoff'
      for (int id = 0; id < 13; ++id)
        for (int ml = 0; ml < g.Ml; ++ml) {
          const int m = h->h_m_local[ml];
          if (m < 0) continue;
          for (int n = 0; n < g.N1; ++n) cf[((size_t)id * g.Ml + ml) * g.N1 + n] = (*src[id])[(size_t)n * g.M1 + m];
        }
      d.coef = dupload(h, cf);
    }
    {
      // retained (m,n) of my wavenumbers grouped by total wavenumber m+n, each group padded to a multiple of 4
      // one descriptor per wavefront of k_spec_update: {n (-1: padding), ml, global m, total wavenumber of the block's wave matrix}:
      // one 16-byte scalar load instead of the chain list entry -> m_local[ml] -> (m, n) arithmetic in front of the first state load
      std::vector<int> act;
      for (int Lw = 0; Lw < (cfg->triang_trunc ? cfg->num_spherical : cfg->num_spherical + cfg->fourier_inc * cfg->num_fourier); ++Lw) {
        for (int ml = 0; ml < g.Ml; ++ml) {
          const int m = h->h_m_local[ml], n = Lw - m * cfg->fourier_inc;
          if (m < 0 || n < 0 || n >= g.N1) continue;
          if (T.tri_mask[(size_t)n * g.M1 + m] != 0.0) { act.push_back(n); act.push_back(ml); act.push_back(m); act.push_back(Lw); }
        }
        while ((act.size() / 4) % 4) { act.push_back(-1); act.push_back(0); act.push_back(0); act.push_back(Lw); }
      }
      // whole blocks of padding up to a multiple of 8 blocks: k_spec_update gives every XCD a CONTIGUOUS eighth of the list (the blocks of one
      // total wavenumber share a wave matrix, those of neighbouring ones the rows n - 1 / n + 1 of the forward batch: one L2 fetches them once)
      const int last_lw = act.empty() ? 0 : act[act.size() - 1];
      while ((act.size() / 16) % 8) for (int q = 0; q < 4; ++q) { act.push_back(-1); act.push_back(0); act.push_back(0); act.push_back(last_lw); }
      h->n_active = (int)(act.size() / 4);
      d.mn_active = dupload(h, act);
    }
    d.pk = dupload(h, T.pk); d.bk = dupload(h, T.bk); d.dpk = dupload(h, T.dpk); d.dbk = dupload(h, T.dbk);
    {
      // rows 0, 1: the two weights of linear_tp_tendency, dt_t = -kappa T_ref (above * c3 + own * c1) / dp (implicit.F90:446-468) -- dlog_3 and dlog_1
      // of the reference column for simmons_and_burridge, dp / p_full and dp / (2 p_full) for 'mcm'; 2: dp; 3: h; 4, 5: dlog_f and dlog_3 of
      // linear_geopotential (:329-359), the same for both options
      std::vector<double> iv(6 * 64, 0.0);
      for (int k = 0; k < g.L; ++k) {
        const double dp = T.dpk[k] + T.dbk[k] * T.ref_surf_p;
        iv[0 * 64 + k] = T.ref_ln_p_half[k + 1] - T.ref_ln_p_full[k];
        iv[1 * 64 + k] = T.ref_ln_p_half[k + 1] - T.ref_ln_p_half[k];
        if (cfg->vert_difference_option == 1) {
          const double p_full_ref = 0.5 * (T.pk[k + 1] + T.pk[k]) + 0.5 * (T.bk[k + 1] + T.bk[k]) * T.ref_surf_p;
          iv[0 * 64 + k] = 0.5 * dp / p_full_ref; iv[1 * 64 + k] = dp / p_full_ref;
        }
        iv[2 * 64 + k] = dp;
        iv[3 * 64 + k] = T.h_impl[k];
        iv[4 * 64 + k] = T.ref_ln_p_half[k + 1] - T.ref_ln_p_full[k];
        iv[5 * 64 + k] = T.ref_ln_p_half[k + 1] - T.ref_ln_p_half[k];
      }
      for (int k = g.L; k < 64; ++k) iv[2 * 64 + k] = 1.0;
      d.impl_vec = dupload(h, iv);
    }
    d.wave_mat_t = dalloc<double>(h, (size_t)(cfg->triang_trunc ? cfg->num_spherical : cfg->num_spherical + cfg->fourier_inc * cfg->num_fourier) * g.L * g.L);
    {
      std::vector<double> tw((size_t)2 * g.I);
      for (int k = 0; k < g.I; ++k) { tw[2 * k] = T.tw_re[k]; tw[2 * k + 1] = T.tw_im[k]; }
      d.tw = dupload(h, tw);
    }
    // ---- state
    const size_t ng3 = (size_t)g.L * g.Jl * g.I, ng2 = (size_t)g.Jl * g.I;
    const size_t ns3 = (size_t)g.Ml * g.N1 * g.L * 2, ns2 = (size_t)g.Ml * g.N1 * 2;
    for (int t = 0; t < 2; ++t) {
      d.ug[t] = dalloc<double>(h, ng3); d.vg[t] = dalloc<double>(h, ng3); d.tg[t] = dalloc<double>(h, ng3);
      d.psg[t] = dalloc<double>(h, ng2); d.tr[t] = dalloc<double>(h, ng3);
      d.vors[t] = dalloc<double>(h, ns3); d.divs[t] = dalloc<double>(h, ns3); d.ts[t] = dalloc<double>(h, ns3);
      d.lnps[t] = dalloc<double>(h, ns2);
    }
    d.vorg = dalloc<double>(h, ng3); d.divg = dalloc<double>(h, ng3); d.dxT = dalloc<double>(h, ng3); d.dyT = dalloc<double>(h, ng3);
    d.dxlp = dalloc<double>(h, ng2); d.dylp = dalloc<double>(h, ng2); d.wg_full = dalloc<double>(h, ng3);
    d.surf_geop = dalloc<double>(h, ng2);
    HIP_CHECK(hipMemsetAsync(d.surf_geop, 0, ng2 * sizeof(double), h->stream));         // flat until isca_dyn_set_surf_geopotential
    d.g_dtu = dalloc<double>(h, ng3); d.g_dtv = dalloc<double>(h, ng3); d.g_dtT = dalloc<double>(h, ng3);
    d.g_E = dalloc<double>(h, ng3); d.g_dtlp = dalloc<double>(h, ng2);
    d.s_dtvor = dalloc<double>(h, ns3); d.s_dtdiv = dalloc<double>(h, ns3); d.s_dtT = dalloc<double>(h, ns3); d.s_dtlp = dalloc<double>(h, ns2);
    if (cfg->raw_filter_coeff != 1.0) {
      d.part_vor = dalloc<double>(h, ns3); d.part_div = dalloc<double>(h, ns3); d.part_t = dalloc<double>(h, ns3); d.part_lp = dalloc<double>(h, ns2);
      d.tr_part = dalloc<double>(h, ng3);
    }
    // ---- work buffers sized for the largest batch (7L+3 level-fields)
    h->cap_cols = 7 * g.L + 3;
    const size_t nF = (size_t)g.P * g.Ml * g.Jl * col_pitch(h->cap_cols), nS = (size_t)g.Ml * g.N1 * col_pitch(h->cap_cols);
    d.Ff_g = dalloc<double>(h, nF); d.Fi_s = dalloc<double>(h, nF);
    if (g.P == 1) { d.Ff_s = d.Ff_g; d.Fi_g = d.Fi_s; }
    else { d.Ff_s = dalloc<double>(h, nF); d.Fi_g = dalloc<double>(h, nF); }
    d.Sf = dalloc<double>(h, nS); d.Si = dalloc<double>(h, nS);
    d.partials = dalloc<double>(h, 12 * (ng2 / 64 + 1));
    d.red = dalloc<double>(h, 32);
    d.fin_args = dalloc<char>(h, deferred_fixer_args_bytes());
    reset_valid_range(h);
    d.wg = dalloc<double>(h, (size_t)(g.L + 1) * ng2); d.trh = dalloc<double>(h, ng3);
    d.tr_atm[0] = dalloc<double>(h, ng3); d.tr_atm[1] = dalloc<double>(h, ng3);
    d.halo_send = dalloc<double>(h, 2 * halo_doubles(g, cfg->num_tracers)); d.halo_recv = dalloc<double>(h, 2 * halo_doubles(g, cfg->num_tracers));
    d.kmask = dalloc<int>(h, ng2 + 2); d.kmask_old = dalloc<int>(h, ng2 + 2); d.wcol = dalloc<double>(h, 5 * ng2); d.psp_copy = dalloc<double>(h, ng2);
    d.w0blk = dalloc<double>(h, (size_t)g.L * (g.Jl / 4 + 1)); d.w0blk_x = dalloc<double>(h, (size_t)g.L * (g.Jl / 4 + 1));
    d.pend = dupload(h, std::vector<double>(PEND_ROWS, PEND_ROWS + 12));
    for (int e = 0; e + 1 < cfg->num_tracers; ++e) {     // tracers 2..: zero until set (cold start: spectral_init_cond.F90 leaves them 0)
      for (int t = 0; t < 2; ++t) {
        d.trx[t][e] = dalloc<double>(h, ng3); d.trx_atm[t][e] = dalloc<double>(h, ng3);
        HIP_CHECK(hipMemsetAsync(d.trx[t][e], 0, ng3 * sizeof(double), h->stream));
        HIP_CHECK(hipMemsetAsync(d.trx_atm[t][e], 0, ng3 * sizeof(double), h->stream));
        if (cfg->tracer_spectral[e + 1]) {
          d.trxs[t][e] = dalloc<double>(h, ns3);
          HIP_CHECK(hipMemsetAsync(d.trxs[t][e], 0, ns3 * sizeof(double), h->stream));
        }
      }
      if (!d.wcol_x) d.wcol_x = dalloc<double>(h, 5 * ng2);
    }
    {  // the tracer's side stream; ISCA_TRACER_PRIO=high|low asks for a queue priority other than the main stream's (measurement switch)
      int lo = 0, hi = 0;
      const char *pr = exp_env("ISCA_TRACER_PRIO");
      if (pr && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
        HIP_CHECK(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, pr[0] == 'h' ? hi : lo));
      else HIP_CHECK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    }
    // fork / join of the side stream: dependencies between kernels of ONE device -- no system-scope fence when the event is recorded (the default writes the
    // caches back and invalidates them for the host and other devices: measured 7.4 us between k_column and k_fft_fwd3 and 6.5 us in front of k_fixer_sums at
    // T85L40 against 0-1.7 us between the other kernels, tools/kernel_gaps.sh).  ISCA_EVENT_SYSTEM_FENCE=1: the default flags.
    const unsigned evf = hipEventDisableTiming | (exp_env("ISCA_EVENT_SYSTEM_FENCE") ? 0u : (unsigned)hipEventDisableSystemFence);
    HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork, evf));
    HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork0, evf));
    HIP_CHECK(hipEventCreateWithFlags(&h->ev_join, evf));
    d.fv_c = dupload(h, T.fv_c); d.fv_cc = dupload(h, T.fv_cc); d.fv_dy = dupload(h, T.fv_dy);
    d.fv_dyy = dupload(h, T.fv_dyy); d.fv_dyp = dupload(h, T.fv_dyp); d.fv_dym = dupload(h, T.fv_dym);
    {
      const int J = g.J;
      std::vector<double> rcdx(J), rdyy(J + 1), rcdy(J), rdy(J + 4), ppm((size_t)6 * g.L, 0.0);
      for (int j = 0; j < J; ++j) { rcdx[j] = 1.0 / (T.fv_dx * T.fv_c[j]); rcdy[j] = 1.0 / (T.fv_dy[j + 2] * T.fv_c[j]); }
      for (int j = 0; j <= J; ++j) rdyy[j] = 1.0 / T.fv_dyy[j];
      for (int j = 0; j < J + 4; ++j) rdy[j] = 1.0 / T.fv_dy[j];
      // slope_z (vert_advection.F90:525-531) and compute_weights (:604-619) with dz = dbk (pk = 0: ps cancels)
      const std::vector<double> &dz = T.dbk;
      const int L = g.L;
      for (int k = 1; k <= L - 2; ++k) {
        const double f = dz[k] / (dz[k - 1] + dz[k] + dz[k + 1]);
        ppm[0 * L + k] = (2. * dz[k - 1] + dz[k]) / (dz[k + 1] + dz[k]) * f;
        ppm[1 * L + k] = (2. * dz[k + 1] + dz[k]) / (dz[k] + dz[k - 1]) * f;
      }
      for (int k = 2; k <= L - 2; ++k) {
        const double d1 = 1.0 / (dz[k - 1] + dz[k]), d2 = 1.0 / (dz[k - 2] + dz[k - 1] + dz[k] + dz[k + 1]);
        const double d3 = 1.0 / (2 * dz[k - 1] + dz[k]), d4 = 1.0 / (dz[k - 1] + 2 * dz[k]);
        const double n3 = dz[k - 2] + dz[k - 1], n4 = dz[k] + dz[k + 1];
        const double x = n3 * d3 - n4 * d4, y = 2.0 * dz[k - 1] * dz[k];
        const double z0 = dz[k - 1] * d1;
        ppm[2 * L + k] = z0 + x * y * d1 * d2; ppm[3 * L + k] = dz[k - 1] * n3 * d3 * d2; ppm[4 * L + k] = dz[k] * n4 * d4 * d2;
      }
      d.fv_rcdx = dupload(h, rcdx); d.fv_rdyy = dupload(h, rdyy); d.fv_rcdy = dupload(h, rcdy); d.fv_rdy = dupload(h, rdy);
      bool sigma = true;
      for (double v : T.pk) if (v != 0.0) sigma = false;
      d.ppm_tab = sigma ? dupload(h, ppm) : nullptr;      // hybrid levels: the tracer kernel forms the weights per column
      // ---- the column kernel's per-level constants on pure sigma levels (k_column_sig).  With pk = 0 every pressure of a column is b x p_s, so the
      // logarithms of pressure_variables (press_and_geopot.F90:165-194) differ from ln p_s by constants of the vertical coordinate:
      //   ln p_half(k+1) - ln p_half(k) = ln(b(k+1)/b(k)) =: a,   ln p_half(k+1) - ln p_full(k) = 1 - b(k) a / (b(k+1) - b(k)) =: d1,   p_full = p_s exp(lcf).
      // The kernel then needs ONE logarithm and ONE exponential per column instead of one and two per level, and no division per level.  Formed in
      // extended precision (d1 is a difference of nearly equal numbers) and rounded once.  Not with vert_difference_option = 'mcm' (its own formulas).
      d.col_sig = nullptr;
      if (sigma && cfg->vert_difference_option != 1 && !getenv("ISCA_COLUMN_GENERIC")) {
        const int L = g.L;
        std::vector<double> cs((size_t)16 * (L + 8), 0.0);       // (+ 8 levels of zeros: the last wavefront's chunk may reach past L)
        const bool top0 = T.bk[0] == 0.0;
        const double vcoeff = -h->tab.vkf / (1.0 - cfg->sigma_b), tcoeff = (h->tab.tks - h->tab.tka) / (1.0 - cfg->sigma_b);
        for (int k = 0; k < L; ++k) {
          const long double b0 = T.bk[k], b1 = T.bk[k + 1], db = b1 - b0;
          long double d1, d3, lcf;
          if (top0 && k == 0) { d1 = 1.0L; d3 = 0.0L; lcf = logl(b1) - 1.0L; }     // ln p_half(top) := 0, ln p_full(top) = ln p_half(2) - 1 (:178-184); d3 only ever meets a zero factor
          else { const long double a = logl(b1 / b0); d1 = 1.0L - b0 * a / db; d3 = a; lcf = logl(b1) - d1; }
          const long double d2 = (top0 && k == 0) ? 0.0L : d3 - d1;
          double *q = &cs[(size_t)16 * k];
          q[0] = (double)d1; q[1] = (double)d3; q[2] = (double)((b1 * d1 + b0 * d2) / db); q[3] = (double)(1.0L / db);
          q[4] = (double)expl(lcf); q[5] = (double)expl((long double)KAPPA * lcf); q[6] = (double)lcf;
          const double sig = q[4];
          const bool bl = sig <= 1.0 && sig > cfg->sigma_b;                         // hs_forcing's boundary layer (hs_forcing.F90:560-575, 645-655)
          q[7] = bl ? vcoeff * (sig - cfg->sigma_b) : 0.0;
          q[8] = bl ? tcoeff * (sig - cfg->sigma_b) : 0.0;
          q[9] = T.dbk[k]; q[10] = T.bk[k + 1]; q[11] = T.bk[k];
        }
        d.col_sig = dupload(h, cs);
      }
      { std::vector<double> sl(g.Jl); for (int j = 0; j < g.Jl; ++j) sl[j] = std::sin(T.rad_lat[g.j0 + j]); d.hs_sin_l = dupload(h, sl); }
    }
    // Tracer kernels on the side stream (fork after the column kernel, join before the fixer sums) pay for themselves from about T85 up:
    // at T42L25 / T21L25 the two cross-stream waits cost more than the overlap hides (0.105 -> 0.094 and 0.100 -> 0.083 ms per step
    // with the tracer on the main stream), at T85L40 the side stream saves 0.04 ms.  One rank only: a sharded step hides them under its exchange.
    h->tracer_serial = exp_env("ISCA_TRACER_SERIAL") != nullptr ||
                       (g.P == 1 && (size_t)g.L * g.Jl * g.I < 500000 && getenv("ISCA_TRACER_CONCURRENT") == nullptr);
    // ISCA_TRACER_EARLY=1 (measurement switch): the horizontal tracer kernel starts beside the column kernel instead of after it.  At T170L60
    // the main stream waits ~130 us for the side stream at the join, but starting the tracer earlier only moves the contention: the column
    // kernel beside it takes 365 us instead of 258 and the step 0.979 against 0.974 ms (T85L40: 0.215 against 0.188) -- the step is bound by
    // the sum of the bytes, not by the order of the kernels.  Off by default.
    {
      const char *e = exp_env("ISCA_TRACER_EARLY");
      h->tracer_early = e && e[0] == '1';
    }
    // ISCA_TRACER_FILTER_IN_VERT=1 (measurement / test switch): the first half of the tracer's Robert filter and the water fixer's "before" sum stay in the
    // vertical kernel (TracerArgs.filt_horiz off), as they were until round 5
    h->tracer_filter_in_vert = getenv("ISCA_TRACER_FILTER_IN_VERT") != nullptr;
    // The grid tracer's transport kernels (van Leer with 2-row halos, PPM with a 5-level stencil) need >= 4 latitude rows per rank and
    // >= 5 levels.  A configuration that asks for the tracer where it cannot run is FATAL -- it used to be dropped without a word;
    // num_tracers = 0 is the way to run without one (field_table without tracers).
    const bool tracer_can = (g.Jl >= 4) && (g.L >= 5);
    if (cfg->num_tracers > 0 && !tracer_can)
      fail("spectral_dynamics_init: the grid tracer needs num_levels >= 5 and lat_max / world_size >= 4; "
           "set num_tracers = 0 to run without it");
    h->tracer_env_off = exp_env("ISCA_NO_TRACER") != nullptr;   // measurement switch, reported by isca_dyn_get_info("tracer_env_off")
    h->tracer_on = tracer_can && (cfg->num_tracers > 0) && !h->tracer_env_off;
    if (cfg->physics == 1) {                  // idealized_moist_phys_init: tables, surface state, tendency arrays
      if (!h->tracer_on) fail("idealized_moist_phys: the specific-humidity grid tracer is not available in this configuration");
      h->moist = moist_create(h->cfg, T);
      d.ph_dtu = dalloc<double>(h, ng3); d.ph_dtv = dalloc<double>(h, ng3); d.ph_dtT = dalloc<double>(h, ng3); d.ph_dtq = dalloc<double>(h, ng3);
      d.t_surf = dalloc<double>(h, ng2); d.precip = dalloc<double>(h, ng2);
      d.moist_work = dalloc<double>(h, moist_work_doubles(g));
      for (int i = 0; i < 2; ++i) { d.cc_dT[i] = dalloc<double>(h, ng3); d.cc_dq[i] = dalloc<double>(h, ng3); d.cc_precip[i] = dalloc<double>(h, ng2); }
      for (int i = 0; i < 2; ++i) { d.m_t[i] = dalloc<double>(h, ng3); d.m_q[i] = dalloc<double>(h, ng3); d.m_ps[i] = dalloc<double>(h, ng2); }       // (lazy fixers: Dev::m_t)
      h->cc_pipeline = !getenv("ISCA_MOIST_NO_PIPELINE");     // the next step's convection beside this step's physics (core.h: cc_valid)
      HIP_CHECK(hipMemsetAsync(d.precip, 0, ng2 * sizeof(double), h->stream));
      launch_t_surf_init(*h, h->stream);      // mixed_layer_init without restart file: the prescribed distribution
    }
    if (hs_forcing_separate(*h)) {            // hs_forcing as a kernel of its own in front of the column kernel (local_heating_option): its three tendency arrays
      d.ph_dtu = dalloc<double>(h, ng3); d.ph_dtv = dalloc<double>(h, ng3); d.ph_dtT = dalloc<double>(h, ng3);
    }
    if (cfg->physics == 2) {                  // the caller's physics: only the arrays its tendencies are handed over in (zero until then)
      d.ph_dtu = dalloc<double>(h, ng3); d.ph_dtv = dalloc<double>(h, ng3); d.ph_dtT = dalloc<double>(h, ng3); d.ph_dtq = dalloc<double>(h, ng3);
      for (double *p : {d.ph_dtu, d.ph_dtv, d.ph_dtT, d.ph_dtq}) HIP_CHECK(hipMemsetAsync(p, 0, ng3 * sizeof(double), h->stream));
      for (int e = 0; e + 1 < cfg->num_tracers; ++e) {
        d.ph_dtqx[e] = dalloc<double>(h, ng3);
        HIP_CHECK(hipMemsetAsync(d.ph_dtqx[e], 0, ng3 * sizeof(double), h->stream));
      }
    }
    for (int i = 0; i < 4; ++i) { d.scratch_g[i] = dalloc<double>(h, (size_t)(g.L + 1) * ng2); d.scratch_s[i] = dalloc<double>(h, (size_t)g.Ml * g.N1 * (g.L + 1) * 2); }
    build_field_lists(h);
    h->Ci = col_pitch(7 * g.L + 3);
    // the MFMA synthesis kernel can generate its B operand from the spectral state (no staged work buffer)
    h->fuse_synth = legendre_mfma_ok(g, cfg->legendre_impl) && cfg->triang_trunc && cfg->fourier_inc == 1;      // the fused gather has the triangle's bounds built in
    if (exp_env("ISCA_NO_FUSE_SYNTH")) h->fuse_synth = false;
    // x-derivatives in Fourier space: with the fused synthesis, the plain Robert filter (the RAW filter re-synthesises the gradients from the adjusted
    // level in a batch of their own) and the lon_max = 256 / 512 FFT kernels
    h->dx_fourier = h->fuse_synth && cfg->raw_filter_coeff == 1.0 && g.I >= 256 && (g.I & (g.I - 1)) == 0 && !exp_env("ISCA_NO_DX_FOURIER") && !exp_env("ISCA_FFT_OLD");
    if (h->dx_fourier) h->Ci = col_pitch(6 * g.L + 2);
    if (virtual_t_on(*h)) d.tv = dalloc<double>(h, ng3);
    // Lazy fixers (core.h): for the plain configurations -- one grid tracer at most, Robert filter without the RAW term, no virtual
    // temperature --; ISCA_EAGER_FIXERS keeps the pass over the fields.  With the moist package (round 5) its pressure kernel, which visits the
    // current level's T anyway, leaves T, q and p_s with the pending scalars applied for the physics kernels (Dev::m_t).
    // (a vertical advection scheme other than second-centred reads the stored previous level in a kernel of its own: eager fixers)
    const bool vadv_ext = cfg->vert_advect_uv != 0 || cfg->vert_advect_t != 0;
    const bool tr1_std = cfg->num_tracers < 1 || tracer_vert_scheme(*h, 0) == 3;       // (tracer 1 with another advect_vert: the option kernel reads stored levels)
    h->lazy_fix = cfg->raw_filter_coeff == 1.0 && cfg->num_tracers <= 1 && !virtual_t_on(*h) && !vadv_ext && tr1_std && !hs_forcing_separate(*h) &&
                  getenv("ISCA_EAGER_FIXERS") == nullptr;
    h->kernels_per_step = (h->fuse_synth ? 8 : 9) - (h->fuse_fwd ? 1 : 0) + (h->tracer_on ? 2 : 0) + (virtual_t_on(*h) ? 1 : 0) + (h->lazy_fix ? (column_takes_deferred_finish(*h) ? -1 : 0) : 1) + (vadv_ext ? 1 : 0);    // lazy fixers: sums + finish (the finish in the next column kernel's block 0 on the plain one-rank path); eager: sums, totals, apply
    upload_deferred_fixer_args(*h);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    HIP_CHECK(hipDeviceSynchronize());
    *out = h;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    if (h) isca_dyn_destroy(h);
    return 1;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// batched transforms on device buffers (single rank: the two Fourier-side views coincide)
// ---------------------------------------------------------------------------------------------------
static void require_single(isca_dyn *h, const char *what) {
  if (h->g.P != 1) fail(std::string(what) + ": only available with world_size == 1 (use the phase API)");
}
// The staged transform pair on more than one rank: the lat <-> m exchange (transpose_fourier / reverse_transpose_fourier, transforms.F90:970-1056)
// between the two stages, issued by the library's communicator like the step's own -- blocks of Ml * Jl * C doubles per peer, whatever the
// batch's column pitch C.  (Without a communicator the host drives the step phase by phase and these transforms are not reachable.)
static void staged_exchange(isca_dyn *h, const double *send, double *recv, int C, const char *what) {
  if (h->g.P == 1) return;
  if (!h->comm) fail(std::string(what) + ": on more than one rank the transform's exchange needs the library's communicator (isca_dyn_comm_init)");
  Timed t(h, what);
  h->comm->all_to_all(send, recv, (size_t)h->g.Ml * h->g.Jl * C, h->stream);
}
static void run_inverse(isca_dyn *h, const FieldList &fl, int full) {    // Si -> grid
  const int C = col_pitch(fl.ncol);
  { Timed t(h, "legendre_inv"); launch_legendre_inverse(h->g, h->d, h->d.Si, h->d.Fi_s, C, full, h->cfg.legendre_impl, h->stream); }
  staged_exchange(h, h->d.Fi_s, h->d.Fi_g, C, "all_to_all_inv");
  { Timed t(h, "fft_inv"); launch_fft_inverse(h->g, h->d, fl, h->d.Fi_g, h->stream); }
}
static void run_forward(isca_dyn *h, const FieldList &fl, int full) {    // grid -> Sf
  const int C = col_pitch(fl.ncol);
  { Timed t(h, "fft_fwd"); launch_fft_forward(h->g, h->d, fl, h->d.Ff_g, h->stream); }
  staged_exchange(h, h->d.Ff_g, h->d.Ff_s, C, "all_to_all_fwd");
  { Timed t(h, "legendre_fwd"); launch_legendre_forward(h->g, h->d, h->d.Ff_s, h->d.Sf, C, full, h->cfg.legendre_impl, h->stream); }
}
static FieldList single_list(double *gptr, int nlev, int op) {
  FieldList f; f.nf = 1; f.ncol = nlev; f.g[0] = gptr; f.nlev[0] = nlev; f.off[0] = 0; f.op[0] = op; return f;
}
static FieldList pair_list(double *a, double *b, int nlev, int op) {
  FieldList f; f.nf = 2; f.ncol = 2 * nlev; f.g[0] = a; f.g[1] = b; f.nlev[0] = f.nlev[1] = nlev; f.off[0] = 0; f.off[1] = nlev; f.op[0] = f.op[1] = op; return f;
}
// device spectral state [ml][n][nlev] -> grid (trans_spherical_to_grid)
static void dev_s2g(isca_dyn *h, const double *spec, double *grid, int nlev, int op) {
  FieldList fl = single_list(grid, nlev, op);
  launch_spec_pack(h->g, spec, h->d.Si, col_pitch(fl.ncol), 0, nlev, h->stream);
  run_inverse(h, fl, 1);
}
static void dev_g2s(isca_dyn *h, double *grid, double *spec, int nlev, int do_trunc, int op) {
  FieldList fl = single_list(grid, nlev, op);
  run_forward(h, fl, 1);
  launch_spec_unpack(h->g, h->d, h->d.Sf, spec, col_pitch(fl.ncol), 0, nlev, do_trunc, h->stream);
}
static void dev_uv_from_vd(isca_dyn *h, const double *vor, const double *div, double *u, double *v, int nlev) {
  FieldList fl = pair_list(u, v, nlev, OP_COSM);
  launch_spec_ucos_vcos(h->g, h->d, vor, div, h->d.Si, col_pitch(fl.ncol), 0, nlev, nlev, h->stream);
  run_inverse(h, fl, 1);
}
static void dev_vd_from_uv(isca_dyn *h, double *u, double *v, double *vor, double *div, int nlev) {
  FieldList fl = pair_list(u, v, nlev, OP_COSM);
  run_forward(h, fl, 1);
  launch_spec_vor_div(h->g, h->d, h->d.Sf, col_pitch(fl.ncol), 0, nlev, vor, div, nlev, h->stream);
}

// all grid fields at time level tl (+ vorg, divg, gradients) from the spectral state at tl
static void synthesize_level(isca_dyn *h, int tl) {
  FieldList fl = inverse_list(h, tl);
  if (h->fuse_synth) {     // the step's own synthesis kernel: a restarted run then continues bit for bit
    const int C = col_pitch(fl.nbuf ? fl.nbuf : fl.ncol);
    { Timed t(h, "legendre_inv"); launch_legendre_inverse(h->g, h->d, h->d.Si, h->d.Fi_s, C, 0, h->cfg.legendre_impl, h->stream, tl, h->dx_fourier); }
    staged_exchange(h, h->d.Fi_s, h->d.Fi_g, C, "all_to_all_inv");
    { Timed t(h, "fft_inv"); launch_fft_inverse(h->g, h->d, fl, h->d.Fi_g, h->stream); }
    return;
  }
  { Timed t(h, "spec_synth_inputs"); launch_spec_synthesis_inputs(*h, tl, h->stream); }
  run_inverse(h, fl, rect_bounds(h));
}

// host (m,n,lev) Fortran <-> device [ml][n][lev] complex
static void spec_host_to_dev(isca_dyn *h, const double *host, double *dev, int nlev) {
  const Geom &g = h->g;
  std::vector<double> tmp((size_t)g.Ml * g.N1 * nlev * 2, 0.0);
  for (int k = 0; k < nlev; ++k)
    for (int n = 0; n < g.N1; ++n)
      for (int ml = 0; ml < g.Ml; ++ml) {
        const int m = h->h_m_local[ml];
        if (m < 0) continue;
        const size_t src = (((size_t)k * g.N1 + n) * g.M1 + m) * 2, dst = (((size_t)ml * g.N1 + n) * nlev + k) * 2;
        tmp[dst] = host[src]; tmp[dst + 1] = host[src + 1];
      }
  HIP_CHECK(hipMemcpyAsync(dev, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
}
// fills only the wavenumbers this rank owns; others are left untouched in `host`
static void spec_dev_to_host(isca_dyn *h, const double *dev, double *host, int nlev) {
  const Geom &g = h->g;
  std::vector<double> tmp((size_t)g.Ml * g.N1 * nlev * 2);
  HIP_CHECK(hipMemcpyAsync(tmp.data(), dev, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
  for (int k = 0; k < nlev; ++k)
    for (int n = 0; n < g.N1; ++n)
      for (int ml = 0; ml < g.Ml; ++ml) {
        const int m = h->h_m_local[ml];
        if (m < 0) continue;
        const size_t dst = (((size_t)k * g.N1 + n) * g.M1 + m) * 2, src = (((size_t)ml * g.N1 + n) * nlev + k) * 2;
        host[dst] = tmp[src]; host[dst + 1] = tmp[src + 1];
      }
}
static void h2d(isca_dyn *h, double *dev, const double *host, size_t n) {
  HIP_CHECK(hipMemcpyAsync(dev, host, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
}
static void d2h(isca_dyn *h, double *host, const double *dev, size_t n) {
  HIP_CHECK(hipMemcpyAsync(host, dev, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
}
static void dcopy(isca_dyn *h, double *dst, const double *src, size_t n) {
  HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
}
// A stand-alone transform on more than one rank: the spectral RESULT on every rank with all its wavenumbers (the host arrays have the whole spectral window on every
// rank: get_spec_domain of the drop-in, DynCore's (n, m) arrays), gathered through the library's communicator -- an all-to-all in which every rank sends its own
// wavenumbers to everyone, through the buffers of the transform's lat -> m exchange, in chunks of levels that fit them.
static void spec_dev_to_host_all(isca_dyn *h, const double *dev, double *host, int nlev) {
  const Geom &g = h->g;
  if (g.P == 1) { spec_dev_to_host(h, dev, host, nlev); return; }
  if (!h->comm) fail("a stand-alone transform on more than one rank needs the library's communicator (isca_dyn_comm_init)");
  const size_t cap = (size_t)g.Ml * g.Jl * col_pitch(h->cap_cols);                       // doubles per peer in Ff_g / Ff_s
  const int kmax = (int)std::min<size_t>((size_t)nlev, cap / ((size_t)g.Ml * g.N1 * 2));
  if (kmax < 1) fail("spectral gather: the exchange buffers hold less than one level");
  std::vector<double> tmp;
  for (int k0 = 0; k0 < nlev; k0 += kmax) {
    const int nk = std::min(kmax, nlev - k0);
    const size_t blk = (size_t)g.Ml * g.N1 * nk * 2;
    launch_spec_level_chunk(g, dev, h->d.Ff_g, nlev, k0, nk, g.P, h->stream);                 // my [ml][n][k0..k0+nk) block, once per peer
    { Timed t(h, "all_to_all_fwd"); h->comm->all_to_all(h->d.Ff_g, h->d.Ff_s, blk, h->stream); }
    tmp.resize((size_t)g.P * blk);
    d2h(h, tmp.data(), h->d.Ff_s, tmp.size());
    for (int q = 0; q < g.P; ++q)
      for (int ml = 0; ml < g.Ml; ++ml) {
        const int m = h->h_m_of_slot[(size_t)q * g.Ml + ml];
        if (m < 0) continue;
        for (int n = 0; n < g.N1; ++n)
          for (int k = 0; k < nk; ++k) {
            const size_t src = (size_t)q * blk + (((size_t)ml * g.N1 + n) * nk + k) * 2, dst = (((size_t)(k0 + k) * g.N1 + n) * g.M1 + m) * 2;
            host[dst] = tmp[src]; host[dst + 1] = tmp[src + 1];
          }
      }
  }
  h->comm->check();
}
// (collective over the ranks when there is more than one)
static void require_single_or_comm(isca_dyn *h, const char *what) {
  if (h->g.P != 1 && !h->comm) fail(std::string(what) + ": on more than one rank this is a collective call through the library's communicator (isca_dyn_comm_init); "
                                    "without one it is only available with world_size == 1");
}

// ---------------------------------------------------------------------------------------------------
// cold start: spectral_initialize_fields.F90:45-135 + spectral_dynamics.F90:580-630
// ---------------------------------------------------------------------------------------------------
// a state write: the moist package's cached pressures and the convection computed ahead for the next step are of the old state
static void moist_invalidate(isca_dyn *h) {
  h->moist_pcache = false;
  h->cc_valid = false;
}
static void reset_pending(isca_dyn *h) {       // a state written from scratch has nothing pending on it
  HIP_CHECK(hipMemcpy(h->d.pend, PEND_ROWS, sizeof(PEND_ROWS), hipMemcpyHostToDevice));
  h->thermo_pending[0] = h->thermo_pending[1] = false;
  h->tr_state[0] = h->tr_state[1] = isca::TR_MAT;
  h->in_step = false;
  h->fin_deferred = false;
  moist_invalidate(h);
}
static void cold_start_single(isca_dyn *h) {
  const Geom &g = h->g;
  Dev &d = h->d;
  const int L = g.L;
  const size_t ng3 = (size_t)L * g.Jl * g.I, ng2 = (size_t)g.Jl * g.I;
  const size_t ns3 = (size_t)g.Ml * g.N1 * L * 2, ns2 = (size_t)g.Ml * g.N1 * 2;
  std::vector<double> vors0((size_t)L * g.N1 * g.M1 * 2, 0.0);
  const int pert[4][2] = {{1, 3}, {5, 3}, {1, 2}, {5, 2}};
  for (auto &mn : pert)
    if (mn[0] < g.M1 && mn[1] < g.N1)
      for (int k = L - 3; k < L; ++k)
        if (k >= 0) vors0[(((size_t)k * g.N1 + mn[1]) * g.M1 + mn[0]) * 2] = 1.e-7;
  spec_host_to_dev(h, vors0.data(), d.vors[0], L);
  HIP_CHECK(hipMemsetAsync(d.divs[0], 0, ns3 * sizeof(double), h->stream));
  dev_uv_from_vd(h, d.vors[0], d.divs[0], d.ug[0], d.vg[0], L);
  std::vector<double> tg(ng3, h->cfg.initial_temperature), lnp(ng2, std::log(h->cfg.reference_sea_level_press));
  if (!h->h_surf_geop.empty())      // ln_psg = log(initial_sea_level_press) - surf_geopotential/(rdgas*initial_temperature)  (spectral_initialize_fields.F90:85)
    for (size_t i = 0; i < ng2; ++i) lnp[i] -= h->h_surf_geop[(size_t)g.j0 * g.I + i] / (RDGAS * h->cfg.initial_temperature);
  h2d(h, d.tg[0], tg.data(), ng3);
  h2d(h, d.psg[0], lnp.data(), ng2);
  dev_g2s(h, d.tg[0], d.ts[0], L, 1, OP_NONE);
  dev_s2g(h, d.ts[0], d.tg[0], L, OP_NONE);
  dev_g2s(h, d.psg[0], d.lnps[0], 1, 1, OP_NONE);
  dev_vd_from_uv(h, d.ug[0], d.vg[0], d.vors[0], d.divs[0], L);
  // ug, vg, tg, psg = exp(ln ps), vorg, divg and the gradient fields in one synthesis batch
  synthesize_level(h, 0);
  for (auto pr : {std::make_pair(d.ug[1], d.ug[0]), std::make_pair(d.vg[1], d.vg[0]), std::make_pair(d.tg[1], d.tg[0])})
    dcopy(h, pr.first, pr.second, ng3);
  dcopy(h, d.psg[1], d.psg[0], ng2);
  dcopy(h, d.vors[1], d.vors[0], ns3); dcopy(h, d.divs[1], d.divs[0], ns3); dcopy(h, d.ts[1], d.ts[0], ns3);
  dcopy(h, d.lnps[1], d.lnps[0], ns2);
  std::vector<double> tr(ng3, h->cfg.initial_sphum);
  h2d(h, d.tr[0], tr.data(), ng3); h2d(h, d.tr[1], tr.data(), ng3);
  h2d(h, d.tr_atm[0], tr.data(), ng3); h2d(h, d.tr_atm[1], tr.data(), ng3);
  for (int e = 0; e < ISCA_MAX_TRACERS - 1; ++e) for (int t = 0; t < 2; ++t) {
    for (double *p : {d.trx[t][e], d.trx_atm[t][e]}) if (p) HIP_CHECK(hipMemsetAsync(p, 0, ng3 * sizeof(double), h->stream));
    if (d.trxs[t][e]) HIP_CHECK(hipMemsetAsync(d.trxs[t][e], 0, ns3 * sizeof(double), h->stream));
  }
  HIP_CHECK(hipMemsetAsync(d.wg_full, 0, ng3 * sizeof(double), h->stream));
  if (h->cfg.physics == 1) launch_t_surf_init(*h, h->stream);       // mixed_layer_init, prescribe_initial_dist
  HIP_CHECK(hipStreamSynchronize(h->stream));
  h->previous = 0; h->current = 0; h->step_count = 0; h->have_state = true; h->phys_calls = 0;
  reset_pending(h);
}

static const char *GRID3[] = {"ug", "vg", "tg", "tr"};
static double *state_ptr(isca_dyn *h, const std::string &name, int tlev, size_t &count, int &kind) {
  // kind: 0 grid (device layout = host layout), 1 spectral 3-D, 2 spectral 2-D
  const Geom &g = h->g;
  Dev &d = h->d;
  const size_t ng3 = (size_t)g.L * g.Jl * g.I, ng2 = (size_t)g.Jl * g.I;
  const int tl = (tlev == 0) ? h->previous : h->current;
  kind = 0;
  if (name == "ug") { count = ng3; return d.ug[tl]; }
  if (name == "vg") { count = ng3; return d.vg[tl]; }
  if (name == "tg") { count = ng3; return d.tg[tl]; }
  if (name == "tr") { count = ng3; return d.tr[tl]; }
  if (name == "tr_atm") { count = ng3; return d.tr_atm[tl]; }
  if (name == "trh") { count = ng3; return d.trh; }
  for (int e = 0; e < ISCA_MAX_TRACERS - 1; ++e) {       // "tr2".."tr8", "tr_atm2"..: tracers 2..num_tracers
    const std::string n = std::to_string(e + 2);
    if (name == "tr" + n || name == "tr_atm" + n || name == "trs" + n) {
      if (!d.trx[0][e]) fail("get/set_state: " + name + ": num_tracers is " + std::to_string(h->cfg.num_tracers));
      if (name[2] == 's') {
        if (!d.trxs[0][e]) fail("get/set_state: " + name + ": tracer " + n + " is a grid tracer");
        kind = 1; count = (size_t)g.L * g.N1 * g.M1 * 2; return d.trxs[tl][e];
      }
      count = ng3; return name[2] == '_' ? d.trx_atm[tl][e] : d.trx[tl][e];
    }
  }
  if (name == "psg") { count = ng2; return d.psg[tl]; }
  if (name == "vorg") { count = ng3; return d.vorg; }
  if (name == "divg") { count = ng3; return d.divg; }
  if (name == "wg_full") { count = ng3; return d.wg_full; }
  if (name == "surf_geopotential") { count = ng2; return d.surf_geop; }      // local band; to SET it use isca_dyn_set_surf_geopotential
  if (name == "t_surf" || name == "precip") {
    if (h->cfg.physics != 1) fail("get/set_state: " + name + " exists only with the moist physics package");
    count = ng2; return name == "t_surf" ? d.t_surf : d.precip;
  }
  if (name == "dt_ug" || name == "dt_vg" || name == "dt_tg" || name == "dt_sphum") {     // physics tendencies of the last step
    if (h->cfg.physics == 0) fail("get/set_state: " + name + " exists only with physics = 1 (moist package) or 2 (caller's physics)");
    count = ng3; return name == "dt_ug" ? d.ph_dtu : name == "dt_vg" ? d.ph_dtv : name == "dt_tg" ? d.ph_dtT : d.ph_dtq;
  }
  if (name == "dxT") { count = ng3; return d.dxT; }
  if (name == "dyT") { count = ng3; return d.dyT; }
  if (name == "dxlp") { count = ng2; return d.dxlp; }
  if (name == "dylp") { count = ng2; return d.dylp; }
  if (name == "g_dtu") { count = ng3; return d.g_dtu; }
  if (name == "g_dtv") { count = ng3; return d.g_dtv; }
  if (name == "g_dtT") { count = ng3; return d.g_dtT; }
  if (name == "g_E") { count = ng3; return d.g_E; }
  if (name == "g_dtlp") { count = ng2; return d.g_dtlp; }
  kind = 1;
  count = (size_t)g.L * g.N1 * g.M1 * 2;
  if (name == "vors") return d.vors[tl];
  if (name == "divs") return d.divs[tl];
  if (name == "ts") return d.ts[tl];
  if (name == "s_dtvor") return d.s_dtvor;
  if (name == "s_dtdiv") return d.s_dtdiv;
  if (name == "s_dtT") return d.s_dtT;
  kind = 2;
  count = (size_t)g.N1 * g.M1 * 2;
  if (name == "ln_ps") return d.lnps[tl];
  if (name == "s_dtlp") return d.s_dtlp;
  (void)GRID3;
  return nullptr;
}

extern "C" int isca_dyn_get_state(isca_dyn_t *h, const char *name, int time_level, double *host, size_t count) {
  API_BEGIN
  if (!h || !name || !host) fail("null argument");
  const Geom &g = h->g;
  const std::string nm(name);
  const size_t ng2 = (size_t)g.Jl * g.I;
  flush_finish(h);                                  // (the (0,0) coefficients of ts, ln_ps wait for it too, not only the lazy grid fields)
  if (is_lazy_field(nm)) materialize(h);
  if (nm == "p_full" || nm == "p_half" || nm == "z_full" || nm == "z_half") {
    // compute_pressures_and_heights of the requested level (atmosphere.F90:229-241, 331-338)
    const int tl = (time_level == 0) ? h->previous : h->current;
    double *pf = h->d.scratch_g[0], *ph = h->d.scratch_g[1], *zf = h->d.scratch_g[2], *zh = h->d.scratch_g[3];
    const double *tq = h->d.tg[tl];
    if (virtual_t_on(*h)) { launch_virtual_t(*h, h->d.tg[tl], h->d.tr_atm[tl], h->d.tv, h->stream); tq = h->d.tv; }      // atmosphere's tracer copy (atmosphere.F90:235-240, 335-337)
    launch_pressures_heights(*h, tq, h->d.psg[tl], pf, ph, zf, zh, h->stream);
    const bool half = (nm == "p_half" || nm == "z_half");
    const size_t need = ng2 * (g.L + (half ? 1 : 0));
    if (count != need) fail("get_state: wrong element count for " + nm);
    d2h(h, host, nm == "p_full" ? pf : nm == "p_half" ? ph : nm == "z_full" ? zf : zh, need);
    return 0;
  }
  size_t cnt; int kind;
  double *p = state_ptr(h, nm, time_level, cnt, kind);
  if (!p) fail("get_state: unknown field " + nm);
  if (cnt != count) fail("get_state: wrong element count for " + nm);
  if (kind == 0) d2h(h, host, p, cnt);
  else spec_dev_to_host(h, p, host, kind == 1 ? g.L : 1);
  API_END
}

extern "C" int isca_dyn_set_state(isca_dyn_t *h, const char *name, int time_level, const double *host, size_t count) {
  API_BEGIN
  if (!h || !name || !host) fail("null argument");
  flush_finish(h);
  if (is_lazy_field(name)) materialize(h);
  size_t cnt; int kind;
  double *p = state_ptr(h, name, time_level, cnt, kind);
  if (!p) fail(std::string("set_state: unknown field ") + name);
  if (cnt != count) fail(std::string("set_state: wrong element count for ") + name);
  if (kind == 0) h2d(h, p, host, cnt);
  else spec_host_to_dev(h, host, p, kind == 1 ? h->g.L : 1);
  h->have_state = true;
  moist_invalidate(h);
  API_END
}

static void raw_gradients_a(isca_dyn *h, int tl);
static void raw_phase_b(isca_dyn *h);
static int raw_pitch(const isca_dyn *h);
// vorg, divg and the gradient fields of the `current` level from its spectral state; the caller's grid
// u, v, T, ps of that level are kept bit for bit (the synthesis would reproduce them only to roundoff)
static void refresh_derived(isca_dyn *h) {
  materialize(h);
  Dev &d = h->d;
  const int tl = h->current;
  const size_t ng2 = (size_t)h->g.Jl * h->g.I, ng3 = ng2 * h->g.L;
  dcopy(h, d.scratch_g[0], d.ug[tl], ng3); dcopy(h, d.scratch_g[1], d.vg[tl], ng3);
  dcopy(h, d.scratch_g[2], d.tg[tl], ng3); dcopy(h, d.scratch_g[3], d.psg[tl], ng2);
  synthesize_level(h, tl);
  if (h->cfg.raw_filter_coeff != 1.0) {      // the running model takes the gradients of the (adjusted) level through the RAW phase's own
    raw_gradients_a(h, tl);                  // kernels (raw_filter_phase): the same here, so that a restarted run continues bit for bit
    staged_exchange(h, h->d.Fi_s, h->d.Fi_g, raw_pitch(h), "all_to_all_raw");
    raw_phase_b(h);
  }
  dcopy(h, d.ug[tl], d.scratch_g[0], ng3); dcopy(h, d.vg[tl], d.scratch_g[1], ng3);
  dcopy(h, d.tg[tl], d.scratch_g[2], ng3); dcopy(h, d.psg[tl], d.scratch_g[3], ng2);
}

// complete_update_of_future (spectral_dynamics.F90:1416-1454): rebuild the spectral side of a level from
// its grid fields, then every derived grid field of that level
extern "C" int isca_dyn_complete_update(isca_dyn_t *h, int time_level) {
  API_BEGIN
  if (!h) fail("null handle");
  if (h->g.P != 1 && !h->comm) require_single(h, "complete_update");      // (collective over the ranks with the library's communicator)
  materialize(h);
  const int tl = (time_level == 0) ? h->previous : h->current;
  Dev &d = h->d;
  const int L = h->g.L;
  const size_t ng2 = (size_t)h->g.Jl * h->g.I;
  dev_vd_from_uv(h, d.ug[tl], d.vg[tl], d.vors[tl], d.divs[tl], L);
  dev_g2s(h, d.tg[tl], d.ts[tl], L, 1, OP_NONE);
  std::vector<double> ps(ng2);
  d2h(h, ps.data(), d.psg[tl], ng2);
  for (auto &x : ps) x = std::log(x);
  h2d(h, d.scratch_g[0], ps.data(), ng2);
  dev_g2s(h, d.scratch_g[0], d.lnps[tl], 1, 1, OP_NONE);
  for (int e = 0; e < ISCA_MAX_TRACERS - 1; ++e)        // spectral tracers: their coefficients from the grid values handed in (:1447-1451)
    if (d.trxs[tl][e]) dev_g2s(h, d.trx[tl][e], d.trxs[tl][e], L, 1, OP_NONE);
  if (tl == h->current) refresh_derived(h);
  HIP_CHECK(hipStreamSynchronize(h->stream));
  API_END
}

// Restart support (spectral_dynamics.F90:509-575 read_restart, :1502-1531 write; atmosphere.F90:197-223).
// After the state arrays of both time levels have been set: restore the leapfrog time pointers
// ('previous'/'current' of the restart file, 0-based here) and the step counter ...
extern "C" int isca_dyn_set_time_pointers(isca_dyn_t *h, int previous, int current, long step_count) {
  API_BEGIN
  if (!h) fail("null handle");
  if (previous < 0 || previous > 1 || current < 0 || current > 1) fail("set_time_pointers: time levels are 0 or 1");
  if (step_count < 0) fail("set_time_pointers: negative step count");
  materialize(h);
  h->previous = previous; h->current = current; h->step_count = step_count;
  h->phys_calls = 0;         // idealized_moist_phys_init sets gust = 1 again after a restart
  moist_invalidate(h);
  API_END
}
// ... then rebuild what the step keeps between calls but the restart file does not hold.
extern "C" int isca_dyn_refresh_derived(isca_dyn_t *h) {
  API_BEGIN
  if (!h || !h->have_state) fail("refresh_derived: no state");
  if (h->g.P != 1 && !h->comm) require_single(h, "refresh_derived");      // (collective over the ranks with the library's communicator: the synthesis' exchange)
  refresh_derived(h);
  HIP_CHECK(hipStreamSynchronize(h->stream));
  API_END
}

// ---------------------------------------------------------------------------------------------------
// the time step: atmosphere.F90:276-352 -> spectral_dynamics.F90:780-1034
// ---------------------------------------------------------------------------------------------------
static StepScalars step_scalars(isca_dyn *h) {
  StepScalars sc;
  sc.prev = h->previous; sc.cur = h->current;
  sc.delta_t = (sc.prev == sc.cur) ? h->cfg.dt_atmos : 2 * h->cfg.dt_atmos;
  sc.fut = (sc.prev == sc.cur) ? 1 - sc.cur : sc.prev;
  sc.xi = sc.delta_t * h->cfg.alpha_implicit;
  return sc;
}
// the grid tracer's transport under two timer names (the horizontal and the vertical kernel; part as in launch_tracer)
static void timed_tracer(isca_dyn *h, const StepScalars &sc, hipStream_t st, int part = -1) {
  if (part != 1) { Timed t(h, "tracer_horiz", st); launch_tracer(*h, sc, st, 0); }
  if (part != 0) { Timed t(h, "tracer_vert", st); launch_tracer(*h, sc, st, 1); }
}
static void phase0(isca_dyn *h, const StepScalars &sc) {          // grid tendencies + longitude FFT
  if (h->fin_deferred && (!column_takes_deferred_finish(*h) || h->tracer_early)) flush_finish(h);      // (a kernel in front of the column kernel reads the scalars: the moist package, hs_forcing on its own)
  h->in_step = true;
  if (h->cfg.physics == 1) {
    // the previous level's pressures are what the step before computed for its current level (grid p_s of a level is final once its
    // fixers are applied; every state write drops the cache): only the current level's are computed then (k_moist_pressures 20 -> 11 us)
    const bool cached = h->moist_pcache && sc.prev != sc.cur && !exp_env("ISCA_MOIST_NO_PCACHE");
    const int slot_prev = cached ? h->moist_pslot : 0, slot_cur = 1 - slot_prev;
    { Timed t(h, "moist_pressures"); launch_moist_pressures(*h, sc, h->stream, slot_prev, slot_cur, cached); }
    // convection + condensation read the previous level only: the kernel of the step before has computed them beside its own chain (cc_valid;
    // the same condition as the cached pressures: no state write since), else they run here, in front
    int ccs = 0;
    if (h->cc_valid && cached) ccs = h->cc_slot;
    else { Timed t(h, "moist_convcond"); launch_moist_convcond(*h, sc.prev, slot_prev, sc.delta_t, ccs, h->stream); }
    { Timed t(h, "moist_physics"); launch_moist_physics(*h, sc, h->stream, slot_cur, ccs, h->cc_pipeline); }
    h->moist_pslot = slot_cur; h->moist_pcache = true;
    h->phys_calls++;
    h->cc_valid = h->cc_pipeline; h->cc_slot = 1 - ccs;      // (the next step's: its previous level is this step's current one, its delta_t the leapfrog's 2 dt)
  }
  // fork: the tracer's vertical kernel needs the column kernel's vertical velocity; its horizontal kernel only state that exists when the
  // step starts (the column kernel's mask word of the step BEFORE: kmask_old), so it can start beside the column kernel (tracer_early:
  // measured, not faster).  Joined before the fixer sums.
  const bool side = h->tracer_on && h->g.P == 1 && !h->tracer_serial, early = side && h->tracer_early;
  // (the filter's first half in the horizontal kernel: not on the first step, where previous and current level share their storage; not when that kernel
  // starts before the column kernel, which leaves the copy of the previous surface pressure it would sum with and reads the unfiltered current level)
  h->tr_filt_horiz = h->tracer_on && sc.prev != sc.cur && !early && !h->tracer_filter_in_vert;
  std::swap(h->d.kmask, h->d.kmask_old);          // the column kernel reads the old word and writes the new one
  if (early) {
    HIP_CHECK(hipEventRecord(h->ev_fork0, h->stream));
    HIP_CHECK(hipStreamWaitEvent(h->stream2, h->ev_fork0, 0));
    timed_tracer(h, sc, h->stream2, 0);
  }
  if (hs_forcing_separate(*h)) { Timed t(h, "hs_forcing"); launch_hs_forcing_step(*h, sc, h->stream); }       // (an hs_forcing_nml option the fused kernel does not carry)
  // (a deferred finish is taken by this launch's block 0.  fin_seq counts exactly those launches -- never 0, parity alternating also across the wrap --:
  // its parity picks the slot the pair is published in, and every such launch empties the other slot for the next one)
  if (h->fin_deferred) h->fin_seq = h->fin_seq >= 0xfffffffeu ? (h->fin_seq & 1u ? 2u : 1u) : h->fin_seq + 1u;
  { Timed t(h, "column"); launch_column(*h, sc, h->stream); }
  h->fin_deferred = false;
  if (h->cfg.vert_advect_uv != 0 || h->cfg.vert_advect_t != 0) { Timed t(h, "vert_advection"); launch_vert_advection_schemes(*h, sc, h->stream); }
  if (h->tracer_on) {
    if (h->g.P > 1) {
      Timed t(h, "tracer_halo"); launch_tracer_pack_halo(*h, sc, h->stream);     // the tracer itself runs once the halo rows are in
    } else if (h->tracer_serial) {
      timed_tracer(h, sc, h->stream);
    } else {
      HIP_CHECK(hipEventRecord(h->ev_fork, h->stream));
      HIP_CHECK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
      timed_tracer(h, sc, h->stream2, early ? 1 : -1);
      HIP_CHECK(hipEventRecord(h->ev_join, h->stream2));
    }
  }
#ifdef ISCA_EXPERIMENTS
  if (h->fuse_fwd) { Timed t(h, "fft_leg_fwd"); launch_fft_legendre_forward(h->g, h->d, h->fl_fwd, h->d.Sf, h->stream); } else
#endif
  { Timed t(h, "fft_fwd"); launch_fft_forward(h->g, h->d, h->fl_fwd, h->d.Ff_g, h->stream); }
}
// Sharded runs: the grid tracer's transport, issued when the neighbours' halo rows have arrived (fv_advection's mpp_update_domains) and
// BEFORE the lat -> m all-to-all, on the side stream: it then runs under that exchange and the spectral pipeline, like on one GPU.
static void phase_tracer(isca_dyn *h, const StepScalars &sc) {
  if (!h->tracer_on || h->g.P == 1) return;
  if (h->tracer_serial) { timed_tracer(h, sc, h->stream); return; }
  HIP_CHECK(hipEventRecord(h->ev_fork, h->stream));
  HIP_CHECK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
  timed_tracer(h, sc, h->stream2);
  HIP_CHECK(hipEventRecord(h->ev_join, h->stream2));
}
static void phase1(isca_dyn *h, const StepScalars &sc) {          // analysis, spectral update, synthesis
  if (!h->fuse_fwd) { Timed t(h, "legendre_fwd"); launch_legendre_forward(h->g, h->d, h->d.Ff_s, h->d.Sf, h->Cf, rect_bounds(h), h->cfg.legendre_impl, h->stream); }
  { Timed t(h, "spec_update"); launch_spec_update(*h, sc, h->stream); }
  if (h->fuse_synth) {
    Timed t(h, "legendre_inv"); launch_legendre_inverse(h->g, h->d, h->d.Si, h->d.Fi_s, h->Ci, 0, h->cfg.legendre_impl, h->stream, sc.fut, h->dx_fourier);
  } else {
    { Timed t(h, "spec_synth_inputs"); launch_spec_synthesis_inputs(*h, sc.fut, h->stream); }
    { Timed t(h, "legendre_inv"); launch_legendre_inverse(h->g, h->d, h->d.Si, h->d.Fi_s, h->Ci, rect_bounds(h), h->cfg.legendre_impl, h->stream); }
  }
}
static void phase2(isca_dyn *h, const StepScalars &sc) {          // inverse FFT + fixer sums
  FieldList fl = inverse_list(h, sc.fut);
  { Timed t(h, "fft_inv"); launch_fft_inverse(h->g, h->d, fl, h->d.Fi_g, h->stream); }
  if (h->tracer_on && !h->tracer_serial) HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  { Timed t(h, "fixer_sums"); launch_fixer_sums(*h, sc.fut, h->stream); }
}
// raw_filter_coeff /= 1: the reference completes the filter of the NEW level after its grid fields have been synthesised
// (complete_robert_filter, spectral_dynamics.F90:1031), so u, v, T, ps, vor, div of that level stay those of the unadjusted spectral
// state while the next step's gradients of T and ln ps (:855, :890) come from the adjusted one: a third transform phase.
// Two halves around the m -> lat exchange of the sharded model (on one rank the two Fourier views are the same buffer):
// raw_phase_a adjusts the new spectral level and synthesises the four gradient fields as far as the Fourier rows of my wavenumbers,
// raw_phase_b transforms the rows of my latitudes to the grid.
static FieldList raw_field_list(isca_dyn *h) {
  const Geom &g = h->g;
  Dev &d = h->d;
  FieldList fl;
  fl.nf = 4;
  double *gp[4] = {d.dxT, d.dyT, d.dxlp, d.dylp};
  int off = 0;
  for (int i = 0; i < 4; ++i) { fl.g[i] = gp[i]; fl.nlev[i] = i < 2 ? g.L : 1; fl.off[i] = off; fl.op[i] = OP_COSM; off += fl.nlev[i]; }
  fl.ncol = off;
  return fl;
}
static int raw_pitch(const isca_dyn *h) { return col_pitch(2 * h->g.L + 2); }
static void raw_gradients_a(isca_dyn *h, int tl) {        // gradients of T and ln ps of spectral level tl as far as the Fourier rows of my wavenumbers
  const Geom &g = h->g;
  Dev &d = h->d;
  const int C = raw_pitch(h);
  Timed t(h, "raw_gradients");
  launch_spec_gradient(g, d, d.ts[tl], d.Si, C, 0, g.L, g.L, h->stream);
  launch_spec_gradient(g, d, d.lnps[tl], d.Si, C, 2 * g.L, 2 * g.L + 1, 1, h->stream);
  launch_legendre_inverse(g, d, d.Si, d.Fi_s, C, rect_bounds(h), h->cfg.legendre_impl, h->stream);     // (rhomboidal: every n of every wavenumber)
}
static void raw_phase_a(isca_dyn *h, const StepScalars &sc) {
  { Timed t(h, "raw_adjust"); launch_raw_adjust(*h, sc.fut, h->stream); }
  raw_gradients_a(h, sc.fut);
}
static void raw_phase_b(isca_dyn *h) {
  Timed t(h, "raw_gradients_fft");
  launch_fft_inverse(h->g, h->d, raw_field_list(h), h->d.Fi_g, h->stream);
}
// A 'spectral' tracer's step (update_tracers, spectral_dynamics.F90:1133-1154, with num_steps = 1): the physics tendency, minus the
// horizontal advection of the current coefficients by the current winds (:1134), plus the second-centred vertical advection of the
// current grid values (:1139-1141), transformed, damped like temperature, stepped (:1144-1148) and synthesised (:1154).  These are
// the staged transforms on the work buffers the step has finished with; the field_table's dry default has no such tracer.
static void spectral_tracer_step(isca_dyn *h, const StepScalars &sc, int e) {
  const Geom &g = h->g;
  Dev &d = h->d;
  const size_t ng3 = (size_t)g.L * g.Jl * g.I;
  double *dt_tr = d.scratch_g[2], *dt_trs = d.scratch_s[1];
  if (h->cfg.physics == 2) dcopy(h, dt_tr, d.ph_dtqx[e], ng3);
  else {
    HIP_CHECK(hipMemsetAsync(dt_tr, 0, ng3 * sizeof(double), h->stream));
    if (h->cfg.physics == 0) launch_tracer_source_sink(*h, d.psg[sc.cur], d.trx_atm[sc.prev][e], dt_tr, h->stream, e + 1);
  }
  FieldList fl = pair_list(d.scratch_g[0], d.scratch_g[1], g.L, OP_COSM);
  launch_spec_gradient(g, d, d.trxs[sc.cur][e], d.Si, col_pitch(fl.ncol), 0, g.L, g.L, h->stream);
  run_inverse(h, fl, 1);
  launch_hadv_combine(g, d.ug[sc.cur], d.vg[sc.cur], d.scratch_g[0], d.scratch_g[1], dt_tr, g.L, h->stream);
  // vert_advection with the entry's advect_vert (:1135-1141): the centred schemes on the current level, the finite-volume ones on the previous
  const int vs = tracer_vert_scheme(*h, e + 1);
  if (vs == 0) launch_vert_advection_centered(*h, d.wg, d.psg[sc.cur], d.trx[sc.cur][e], dt_tr, h->stream);
  else launch_vert_advection_field(*h, vs, d.psg[sc.cur], d.trx[vs == 1 ? sc.cur : sc.prev][e], dt_tr, sc.delta_t, h->stream);
  if (h->cfg.tracer_hole_filling[e + 1]) launch_water_borrowing(*h, d.psg[sc.cur], d.trx[sc.prev][e], dt_tr, sc.delta_t, h->stream);     // :1142-1144
  dev_g2s(h, dt_tr, dt_trs, g.L, 1, OP_NONE);
  launch_spec_tracer_update(*h, sc, e, dt_trs, h->stream);
  dev_s2g(h, d.trxs[sc.fut][e], d.trx[sc.fut][e], g.L, OP_NONE);
}
// part 0: the whole phase; sharded with raw_filter_coeff /= 1 the exchange of the re-synthesised gradients' Fourier rows lies between
// part 1 (fixers, the filter's adjustment, Legendre synthesis) and part 2 (their FFT, everything else, pointer rotation)
static void phase3(isca_dyn *h, const StepScalars &sc, int part = 0) {          // fixers, pointer rotation
  const bool raw = h->cfg.raw_filter_coeff != 1.0;
  if (part != 2) {
    if (h->lazy_fix) {       // the scalars only: left pending on the new level (and, for the tracer's filter, on the current one)
      // the plain column kernel next: its block 0 finishes (ColumnArgs::fin; sharded: from the all-reduced red[0..9]) -- no launch here; otherwise
      // (diagnostics that read the level right away, a configuration whose column kernel cannot) the one-block kernel
      if (column_takes_deferred_finish(*h) && !h->diag_mask) {
        h->fin_deferred = true; h->fin_prev = sc.prev; h->fin_cur = sc.cur; h->fin_fut = sc.fut;
      } else { Timed t(h, "fixer_finish"); launch_fixer_finish(*h, sc, h->stream); }
      h->thermo_pending[sc.fut] = true;
      if (h->tracer_on) { h->tr_state[sc.cur] = isca::TR_FILT; h->tr_state[sc.fut] = isca::TR_NEW; }
    } else { Timed t(h, "fixer_apply"); launch_fixer_apply(*h, sc, h->stream); }
    if (raw) raw_phase_a(h, sc);
    if (part == 1) return;
  }
  h->in_step = false;
  if (raw) raw_phase_b(h);
  if (h->tracer_on && h->cfg.num_tracers > 1) {
    Timed t(h, "tracers_2_up");
    for (int e = 0; e + 1 < h->cfg.num_tracers; ++e) {     // tracers 2..num_tracers (their transport, if 'grid', ran beside tracer 1's)
      if (h->cfg.tracer_spectral[e + 1]) spectral_tracer_step(h, sc, e);
      launch_tracer_finish(*h, sc, e, h->stream);
    }
  }
  if (h->diag_mask) {   // spectral_diagnostics(Time_next, psg(future), ug(future), ...) at the end of atmosphere (atmosphere.F90:344)
    if (h->lazy_fix) {  // the diagnostics read the stored fields: with them on, the pending corrections are applied every step
      h->previous = sc.cur; h->current = sc.fut;     // (materialize takes the newest level from the time pointers)
      Timed t(h, "fixer_materialize"); materialize(h);
    }
    Timed t(h, "diagnostics"); launch_diag_accumulate(*h, sc.fut, h->stream);
    h->diag_count += 1;
  }
  h->previous = sc.cur;
  h->current = sc.fut;
  h->step_count += 1;
}

// A step that ends in an exception must not leave the handle "in the middle of a step" (every later get_state / set_time_pointers
// would refuse), nor its peers waiting in the next exchange (error_mesg(..., FATAL) stops every PE).
struct StepGuard {
  isca_dyn *h; bool ok = false;
  explicit StepGuard(isca_dyn *h_) : h(h_) {}
  void done() { ok = true; }
  ~StepGuard() {
    if (ok) return;
    h->in_step = false;
    if (h->comm) h->comm->abort();
  }
};

// One step of the latitude-band sharded model with the exchanges issued on the same stream through RCCL:
// lat -> m all-to-all (transpose_fourier), m -> lat all-to-all (reverse_transpose_fourier), the tracer's 2-row halo
// exchange (mpp_update_domains in fv_advection) and the all-reduce of the 10 fixer sums.  Nothing returns to the host.
static void sharded_step(isca_dyn *h, int store_wg_full = 1) {
  StepScalars sc = step_scalars(h);
  sc.store_wg_full = store_wg_full;
  const Geom &g = h->g;
  isca::Comm &c = *h->comm;
  upload_wave_matrices(h, sc.delta_t);
  { Timed seg(h, "seg_grid", nullptr, true); phase0(h, sc); }
  // ISCA_HALO_WITH_ALL_TO_ALL=1: the tracer's halo rows travel in the group of the lat -> m all-to-all (Comm::all_to_all_with_halo: one exchange, one
  // latency hop less on the step's critical path -- three instead of four); the tracer's transport then starts behind that exchange and runs under
  // the spectral phase instead of under the exchange.  Same results; which order is faster is a question for a node with more than one GPU.
  static const bool fold_halo = getenv("ISCA_HALO_WITH_ALL_TO_ALL") != nullptr;
  if (h->tracer_on && fold_halo) {
    const size_t n = halo_doubles(g, h->cfg.num_tracers);
    { Timed t(h, "all_to_all_fwd"); c.all_to_all_with_halo(h->d.Ff_g, h->d.Ff_s, (size_t)g.Ml * g.Jl * h->Cf, h->d.halo_send, h->d.halo_send + n,
                                                           h->d.halo_recv, h->d.halo_recv + n, n, h->stream); }
    phase_tracer(h, sc);
  } else {
    if (h->tracer_on) {   // the tracer's halo rows first (small), so that its transport runs under the all-to-all
      const size_t n = halo_doubles(g, h->cfg.num_tracers);
      { Timed t(h, "halo"); c.halo(h->d.halo_send, h->d.halo_send + n, h->d.halo_recv, h->d.halo_recv + n, n, h->stream); }
      phase_tracer(h, sc);
    }
    { Timed t(h, "all_to_all_fwd"); c.all_to_all(h->d.Ff_g, h->d.Ff_s, (size_t)g.Ml * g.Jl * h->Cf, h->stream); }
  }
  { Timed seg(h, "seg_spectral", nullptr, true); phase1(h, sc); }
  { Timed t(h, "all_to_all_inv"); c.all_to_all(h->d.Fi_s, h->d.Fi_g, (size_t)g.Ml * g.Jl * h->Ci, h->stream); }
  { Timed seg(h, "seg_fft_inv", nullptr, true); phase2(h, sc); }
  { Timed t(h, "all_reduce"); c.all_reduce_sum(h->d.red, 10, h->stream); }
  if (h->cfg.raw_filter_coeff != 1.0) {    // a third exchange: the Fourier rows of the gradients re-synthesised from the adjusted level
    phase3(h, sc, 1);
    { Timed t(h, "all_to_all_raw"); c.all_to_all(h->d.Fi_s, h->d.Fi_g, (size_t)g.Ml * g.Jl * raw_pitch(h), h->stream); }
    phase3(h, sc, 2);
  } else { Timed seg(h, "seg_fixers", nullptr, true); phase3(h, sc); }
}

extern "C" int isca_dyn_step(isca_dyn_t *h, int nsteps, int sync) {
  API_BEGIN
  if (!h) fail("null handle");
  if (!h->have_state) fail("isca_dyn_step: no state (call isca_dyn_cold_start or set_state first)");
  if (h->g.P > 1 && !h->comm)
    fail("isca_dyn_step: world_size > 1 needs isca_dyn_comm_init first (or drive isca_dyn_step_phase and the exchanges from the host)");
  if (h->cfg.physics == 2 && nsteps != 0)
    fail("isca_dyn_step: physics = 2 has no physics of its own: hand the tendencies to isca_dyn_dynamics, one call per step");
  StepGuard guard(h);
  for (int i = 0; i < nsteps; ++i) {
    // wg_full (omega) is an output only: the last step of the call stores it, and every step while a diagnostic of omega accumulates
    // (or the moist package runs, whose restart and diagnostics see it too)
    const int store_wg = (i == nsteps - 1) || (h->diag_mask & 0x3C040u) || h->hist_wg_full || h->cfg.physics == 1 || exp_env("ISCA_ALWAYS_WG_FULL");
    if (h->g.P > 1) sharded_step(h, store_wg);
    else {
      StepScalars sc = step_scalars(h);
      sc.store_wg_full = store_wg;
      upload_wave_matrices(h, sc.delta_t);
      phase0(h, sc); phase1(h, sc); phase2(h, sc); phase3(h, sc);
    }
    if (h->hist) isca_history_after_step(h);       // an open diag_table: its files' records as their intervals complete (history_nc.cpp)
  }
  guard.done();
  if (sync) sync_and_check_valid_range(h);
  API_END
}
// spectral_dynamics(Time, psg_final, ug_final, vg_final, tg_final, tracer_attributes, grid_tracers_final, time_level_out, dt_psg, dt_ug,
// dt_vg, dt_tg, dt_tracers, wg_full, p_full, p_half, z_full) (spectral_dynamics.F90:780-795) as atmosphere calls it (atmosphere.F90:325)
// after a physics package of the host's own: physics = 2.  The tendencies are the arrays the reference's physics fills between
// atmosphere.F90:300 and :321 -- (lon, lat_local, lev), accumulated by the caller, null = zero -- on the host or (on_device) in device
// memory.  set_tendencies only hands them over (for the phase-by-phase sharded driver); dynamics also runs the step.
static void stage_tendencies(isca_dyn *h, const double *dt_ug, const double *dt_vg, const double *dt_tg, const double *dt_tracers, int on_device) {
  if (h->cfg.physics != 2) fail("tendencies of a caller's physics need a handle created with physics = 2");
  const size_t ng3 = (size_t)h->g.L * h->g.Jl * h->g.I;
  const double *src[4] = {dt_ug, dt_vg, dt_tg, dt_tracers};
  double *dst[4] = {h->d.ph_dtu, h->d.ph_dtv, h->d.ph_dtT, h->d.ph_dtq};
  if (h->cfg.num_tracers == 0) src[3] = nullptr;      // a field_table without tracers: dt_tracers has no elements (a Fortran caller cannot pass NULL)
  for (int i = 0; i < 4; ++i) {
    if (!src[i]) HIP_CHECK(hipMemsetAsync(dst[i], 0, ng3 * sizeof(double), h->stream));
    else HIP_CHECK(hipMemcpyAsync(dst[i], src[i], ng3 * sizeof(double), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  }
  for (int e = 0; e + 1 < h->cfg.num_tracers; ++e) {              // dt_tracers(:,:,:,ntr): tracer index slowest
    if (!dt_tracers) HIP_CHECK(hipMemsetAsync(h->d.ph_dtqx[e], 0, ng3 * sizeof(double), h->stream));
    else HIP_CHECK(hipMemcpyAsync(h->d.ph_dtqx[e], dt_tracers + (size_t)(e + 1) * ng3, ng3 * sizeof(double),
                                  on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  }
  if (!on_device) HIP_CHECK(hipStreamSynchronize(h->stream));      // the caller may reuse its arrays
}
extern "C" int isca_dyn_set_tendencies(isca_dyn_t *h, const double *dt_ug, const double *dt_vg, const double *dt_tg, const double *dt_tracers,
                                       int on_device) {
  API_BEGIN
  if (!h) fail("null handle");
  stage_tendencies(h, dt_ug, dt_vg, dt_tg, dt_tracers, on_device);
  API_END
}
extern "C" int isca_dyn_dynamics(isca_dyn_t *h, const double *dt_ug, const double *dt_vg, const double *dt_tg, const double *dt_tracers,
                                 int on_device, int sync) {
  API_BEGIN
  if (!h) fail("null handle");
  if (!h->have_state) fail("isca_dyn_dynamics: no state (call isca_dyn_cold_start or set_state first)");
  if (h->g.P > 1 && !h->comm)
    fail("isca_dyn_dynamics: world_size > 1 needs isca_dyn_comm_init first (or isca_dyn_set_tendencies + isca_dyn_step_phase driven by the host)");
  StepGuard guard(h);
  stage_tendencies(h, dt_ug, dt_vg, dt_tg, dt_tracers, on_device);
  if (h->g.P > 1) sharded_step(h);
  else {
    const StepScalars sc = step_scalars(h);
    upload_wave_matrices(h, sc.delta_t);
    phase0(h, sc); phase1(h, sc); phase2(h, sc); phase3(h, sc);
  }
  if (h->hist) isca_history_after_step(h);
  guard.done();
  if (sync) sync_and_check_valid_range(h);
  API_END
}
// delta_t of the step about to be taken (dt_atmos on a first step or after a restart with previous == current, else 2 dt_atmos:
// atmosphere.F90:286-290): what the caller's physics receives as its time step
extern "C" int isca_dyn_delta_t(isca_dyn_t *h, double *delta_t) {
  API_BEGIN
  if (!h || !delta_t) fail("null argument");
  *delta_t = step_scalars(h).delta_t;
  API_END
}
// RCCL communicator for the sharded step.  Rank 0 obtains the 128-byte id and hands it to the other ranks by any
// means (the Python driver broadcasts it with torch.distributed); every rank then calls isca_dyn_comm_init.
extern "C" int isca_comm_get_unique_id(void *id128) {
  API_BEGIN
  if (!id128) fail("null argument");
  isca::Comm::unique_id(id128);
  API_END
}
extern "C" int isca_dyn_comm_init(isca_dyn_t *h, const void *id128) {
  API_BEGIN
  if (!h || !id128) fail("null argument");
  if (h->comm) fail("comm_init: communicator already initialised");
  HIP_CHECK(hipSetDevice(h->cfg.device));
  h->comm = isca::Comm::create(id128, h->cfg.rank, h->cfg.world_size);
  // (ISCA_COMM=peer: the ranks store into each other's receive buffers -- exported and opened here, collectively)
  h->comm->attach(h->d.Ff_s, h->d.Fi_g, h->tracer_on ? h->d.halo_recv : nullptr, h->tracer_on ? halo_doubles(h->g, h->cfg.num_tracers) : 0);
  API_END
}
// A host without a message-passing layer of its own (the Fortran drop-in on an mpp built without MPI; any launcher that only sets environment
// variables): the rank and the number of ranks from the environment, and the 128-byte id handed from rank 0 to the others through a file.
// isca_env_rank: ISCA_RANK / ISCA_WORLD_SIZE, else what torchrun (RANK / WORLD_SIZE / LOCAL_RANK), Open MPI (OMPI_COMM_WORLD_RANK / _SIZE /
// _LOCAL_RANK), MPICH / Slurm (PMI_RANK / PMI_SIZE, SLURM_PROCID / SLURM_NTASKS / SLURM_LOCALID) export; one rank when none is set.
extern "C" int isca_env_rank(int *rank, int *world_size, int *local_rank) {
  static const char *const names[][3] = {{"ISCA_RANK", "ISCA_WORLD_SIZE", "ISCA_LOCAL_RANK"}, {"RANK", "WORLD_SIZE", "LOCAL_RANK"},
                                         {"OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK"},
                                         {"PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID"}, {"SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID"}};
  int r = 0, w = 1, l = -1;
  for (const auto &n : names) {
    const char *a = getenv(n[0]), *b = getenv(n[1]);
    if (a && b && *a && *b) { r = atoi(a); w = atoi(b); if (const char *c = getenv(n[2])) l = atoi(c); break; }
  }
  if (w < 1 || r < 0 || r >= w) { g_last_error = "isca_env_rank: inconsistent rank / world size in the environment"; return 1; }
  if (rank) *rank = r;
  if (world_size) *world_size = w;
  if (local_rank) *local_rank = l < 0 ? r : l;
  return 0;
}
// collective: rank 0 draws the id (isca_comm_get_unique_id: RCCL's, or the host-staged exchange's with ISCA_COMM=ipc) and leaves it in the file
// ISCA_COMM_ID_FILE names (written under another name and renamed: never seen half-written); the others wait for the file (ISCA_IPC_TIMEOUT_S,
// 120 s); everybody then runs isca_dyn_comm_init and isca_dyn_comm_check.  The file is removed by rank 0 once every rank has answered the check.
extern "C" int isca_dyn_comm_init_env(isca_dyn_t *h) {
  API_BEGIN
  if (!h) fail("null argument");
  if (h->cfg.world_size == 1) return 0;
  const char *path = getenv("ISCA_COMM_ID_FILE");
  if (!path || !*path) fail("comm_init_env: ISCA_COMM_ID_FILE (a path every rank can read) is not set");
  unsigned char id[128];
  // A file left behind by a run that crashed before rank 0 removed it must not be taken for this launch's: the launcher's ISCA_COMM_NONCE (a job id;
  // any string, the same on every rank) is written behind the id and the other ranks accept only a file that carries it.  Without the variable the
  // file's age decides (below), which a relaunch within the minute can defeat: launchers should set the nonce or remove the file first.
  char nonce[64] = {0};
  if (const char *e = getenv("ISCA_COMM_NONCE")) snprintf(nonce, sizeof(nonce), "%s", e);
  if (h->cfg.rank == 0) {
    if (isca_comm_get_unique_id(id)) fail(g_last_error);
    remove(path);                              // (an earlier run's)
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, 128, f) != 128 || fwrite(nonce, 1, sizeof(nonce), f) != sizeof(nonce) || fclose(f) != 0 || rename(tmp.c_str(), path) != 0)
      fail(std::string("comm_init_env: cannot write ") + path);
  } else {
    const double limit = getenv("ISCA_IPC_TIMEOUT_S") ? atof(getenv("ISCA_IPC_TIMEOUT_S")) : 120.0;
    double waited = 0.0;
    const time_t born = time(nullptr);        // a file left behind by an earlier run (rank 0 removes its own after use; a crashed run may not have) is not this run's id:
    for (;;) {                                // only a file written after this process came up, give or take a minute between the ranks' starts, is accepted
      struct stat st;
      FILE *f = (stat(path, &st) == 0 && st.st_mtime + 60 >= born) ? fopen(path, "rb") : nullptr;
      if (f) {
        char theirs[sizeof(nonce)] = {0};
        const size_t n = fread(id, 1, 128, f), nn = fread(theirs, 1, sizeof(theirs), f);
        fclose(f);
        if (n == 128 && nn == sizeof(theirs) && memcmp(theirs, nonce, sizeof(nonce)) == 0) break;      // (another launch's nonce: keep waiting for ours)
      }
      if (waited > limit) fail(std::string("comm_init_env: rank 0 did not leave the communicator id in ") + path);
      usleep(20000); waited += 0.02;
    }
  }
  if (isca_dyn_comm_init(h, id)) fail(g_last_error);
  if (isca_dyn_comm_check(h)) fail(g_last_error);
  if (h->cfg.rank == 0) remove(path);          // (every rank has read it: the check is collective)
  API_END
}
// "rccl" or "ipc" (comm.h), "" without a communicator
extern "C" const char *isca_dyn_comm_kind(isca_dyn_t *h) { return h && h->comm ? h->comm->kind() : ""; }
// Cross-rank check of the communicator before the step trusts it: every collective of the sharded step moves rank-tagged
// patterns through the step's own buffers (all-to-all + halo in one group, plain all-to-all, all-reduce) and each rank verifies
// what it received.  Collective over all ranks; non-zero = this rank saw wrong data (the caller then falls back).
extern "C" int isca_dyn_comm_check(isca_dyn_t *h) {
  API_BEGIN
  if (!h || !h->comm) fail("comm_check: no communicator");
  const Geom &g = h->g;
  isca::Comm &c = *h->comm;
  const int P = g.P, me = g.rank;
  const size_t blk = (size_t)g.Ml * g.Jl * h->Cf;                   // doubles per peer of the forward exchange
  const size_t nh = h->tracer_on ? halo_doubles(g, h->cfg.num_tracers) : 0;
  std::vector<double> send(blk * P), recv(blk * P, -1.0);
  for (int q = 0; q < P; ++q)
    for (size_t i = 0; i < blk; ++i) send[q * blk + i] = 1000.0 * me + q + 1e-3 * (double)(i % 997);
  h2d(h, h->d.Ff_g, send.data(), send.size());
  std::vector<double> hs(2 * nh), hr(2 * nh, -1.0);
  for (size_t i = 0; i < nh; ++i) { hs[i] = 10.0 * me + 1 + 1e-3 * (double)(i % 991); hs[nh + i] = 10.0 * me + 2 + 1e-3 * (double)(i % 991); }
  if (nh) { h2d(h, h->d.halo_send, hs.data(), hs.size()); h2d(h, h->d.halo_recv, hr.data(), hr.size()); }
  // all collectives first (every rank takes part in each of them), verification afterwards
  c.all_to_all_with_halo(h->d.Ff_g, h->d.Ff_s, blk, h->d.halo_send, h->d.halo_send + nh, h->d.halo_recv, h->d.halo_recv + nh, nh, h->stream);
  d2h(h, recv.data(), h->d.Ff_s, recv.size());
  if (nh) d2h(h, hr.data(), h->d.halo_recv, hr.size());
  const size_t blk_i = (size_t)g.Ml * g.Jl * h->Ci;
  std::vector<double> s2(blk_i * P), r2(blk_i * P, -1.0);
  for (int q = 0; q < P; ++q)
    for (size_t i = 0; i < blk_i; ++i) s2[q * blk_i + i] = 7.0 * me + 0.5 * q + 1e-3 * (double)(i % 983);
  h2d(h, h->d.Fi_s, s2.data(), s2.size());
  c.all_to_all(h->d.Fi_s, h->d.Fi_g, blk_i, h->stream);
  d2h(h, r2.data(), h->d.Fi_g, r2.size());
  std::vector<double> red(10), keep(32);
  d2h(h, keep.data(), h->d.red, 32);
  for (int i = 0; i < 10; ++i) red[i] = (double)(me + 1) * (i + 1);
  h2d(h, h->d.red, red.data(), 10);
  c.all_reduce_sum(h->d.red, 10, h->stream);
  d2h(h, red.data(), h->d.red, 10);
  h2d(h, h->d.red, keep.data(), 32);
  for (int q = 0; q < P; ++q)
    for (size_t i = 0; i < blk; ++i)
      if (recv[q * blk + i] != 1000.0 * q + me + 1e-3 * (double)(i % 997)) fail("comm_check: all-to-all delivered wrong data");
  for (size_t i = 0; i < nh; ++i) {     // recv_lo came from rank-1's send_hi, recv_hi from rank+1's send_lo
    if (me > 0 && hr[i] != 10.0 * (me - 1) + 2 + 1e-3 * (double)(i % 991)) fail("comm_check: halo (lower neighbour) delivered wrong data");
    if (me < P - 1 && hr[nh + i] != 10.0 * (me + 1) + 1 + 1e-3 * (double)(i % 991)) fail("comm_check: halo (upper neighbour) delivered wrong data");
  }
  for (int q = 0; q < P; ++q)
    for (size_t i = 0; i < blk_i; ++i)
      if (r2[q * blk_i + i] != 7.0 * q + 0.5 * me + 1e-3 * (double)(i % 983)) fail("comm_check: inverse all-to-all delivered wrong data");
  for (int i = 0; i < 10; ++i)
    if (red[i] != 0.5 * P * (P + 1) * (i + 1)) fail("comm_check: all-reduce gave a wrong sum");
  API_END
}
// Exercises every collective of the sharded step on a communicator of this process alone (world_size 1):
// checks that RCCL can be loaded and that send/recv, grouped exchange and all-reduce run on the given device.
extern "C" int isca_comm_selftest(int device, double *max_err) {
  API_BEGIN
  HIP_CHECK(hipSetDevice(device));
  char id[isca::Comm::UNIQUE_ID_BYTES];
  isca::Comm::unique_id(id);
  std::unique_ptr<isca::Comm> cp(isca::Comm::create(id, 0, 1));
  isca::Comm &c = *cp;
  const size_t n = 4096;
  std::vector<double> a(n), b(n, 0.0), r(16);
  for (size_t i = 0; i < n; ++i) a[i] = 0.25 * (double)i - 7.0;
  for (int i = 0; i < 16; ++i) r[i] = 1.0 + i;
  double *da, *db, *dr;
  hipStream_t s;
  HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  HIP_CHECK(hipMalloc((void **)&da, n * sizeof(double))); HIP_CHECK(hipMalloc((void **)&db, n * sizeof(double)));
  HIP_CHECK(hipMalloc((void **)&dr, 16 * sizeof(double)));
  HIP_CHECK(hipMemcpyAsync(da, a.data(), n * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMemsetAsync(db, 0, n * sizeof(double), s));
  HIP_CHECK(hipMemcpyAsync(dr, r.data(), 16 * sizeof(double), hipMemcpyHostToDevice, s));
  c.all_to_all(da, db, n, s);
  c.all_reduce_sum(dr, 10, s);
  c.halo(da, da, db, db, 16, s);              // no neighbours: must be a no-op
  c.all_to_all_with_halo(da, db, n, da, da, db, db, 16, s);
  HIP_CHECK(hipMemcpyAsync(b.data(), db, n * sizeof(double), hipMemcpyDeviceToHost, s));
  std::vector<double> r2(16);
  HIP_CHECK(hipMemcpyAsync(r2.data(), dr, 16 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  double e = 0.0;
  for (size_t i = 0; i < n; ++i) e = std::max(e, std::fabs(a[i] - b[i]));
  for (int i = 0; i < 16; ++i) e = std::max(e, std::fabs(r[i] - r2[i]));
  if (max_err) *max_err = e;
  hipFree(da); hipFree(db); hipFree(dr); hipStreamDestroy(s);
  API_END
}
extern "C" int isca_dyn_synchronize(isca_dyn_t *h) {
  API_BEGIN
  sync_and_check_valid_range(h);
  API_END
}

extern "C" int isca_dyn_step_phase(isca_dyn_t *h, int phase) {
  API_BEGIN
  if (!h || !h->have_state) fail("isca_dyn_step_phase: no state");
  StepScalars sc = step_scalars(h);
  sc.keep_spec_tend = 1;
  try {
  switch (phase) {
    case 0: upload_wave_matrices(h, sc.delta_t); phase0(h, sc); break;
    case 1: phase1(h, sc); break;
    case 2: phase2(h, sc); break;
    case 3:
      if (h->g.P > 1 && h->cfg.raw_filter_coeff != 1.0) fail("step_phase(3): with raw_filter_coeff /= 1 a sharded step ends with phases 5, exchange 2, 6");
      phase3(h, sc); break;
    case 5: phase3(h, sc, 1); break;
    case 6: phase3(h, sc, 2); break;
    case 4: phase_tracer(h, sc); break;
    default: fail("invalid phase");
  }
  } catch (...) { h->in_step = false; throw; }       // a failed phase must not lock the state behind "in the middle of a step"
  API_END
}
extern "C" int isca_dyn_exchange_buffers(isca_dyn_t *h, int which, void **send, void **recv, size_t *bytes_per_peer) {
  API_BEGIN
  const Geom &g = h->g;
  const int C = which == 0 ? h->Cf : (which == 1 ? h->Ci : raw_pitch(h));
  if (which == 0) { *send = h->d.Ff_g; *recv = h->d.Ff_s; }
  else if (which == 1 || which == 2) { *send = h->d.Fi_s; *recv = h->d.Fi_g; }      // 2: the RAW filter's gradient batch (2 L + 2 level-fields)
  else fail("invalid buffer id");
  *bytes_per_peer = (size_t)g.Ml * g.Jl * C * sizeof(double);
  API_END
}
extern "C" int isca_dyn_halo_buffers(isca_dyn_t *h, void **send_lo, void **send_hi, void **recv_lo, void **recv_hi, size_t *bytes) {
  API_BEGIN
  const size_t n = halo_doubles(h->g, h->cfg.num_tracers);
  *send_lo = h->d.halo_send; *send_hi = h->d.halo_send + n; *recv_lo = h->d.halo_recv; *recv_hi = h->d.halo_recv + n;
  *bytes = h->tracer_on ? n * sizeof(double) : 0;
  API_END
}
extern "C" int isca_dyn_reduce_buffer(isca_dyn_t *h, void **buf, size_t *count) {
  API_BEGIN
  *buf = h->d.red; *count = 10;
  API_END
}

// get_topography (spectral_init_cond.F90:167-308) hands spectral_dynamics a surface geopotential; here the caller does: the GLOBAL
// (lon_max, lat_max) field, the same on every rank, before isca_dyn_cold_start or a restart's set_state calls.  It enters the initial
// surface pressure (spectral_initialize_fields.F90:85), the hydrostatic integral of the dynamics (press_and_geopot.F90:331) and the
// heights handed to the physics.  The reference's own smoothing / truncation of the field is the caller's business (isca_amd/atmosphere.py).
extern "C" int isca_dyn_set_surf_geopotential(isca_dyn_t *h, const double *global_field, size_t count) {
  API_BEGIN
  if (!h || !global_field) fail("null argument");
  const Geom &g = h->g;
  if (count != (size_t)g.J * g.I) fail("set_surf_geopotential: the field must have lon_max * lat_max values");
  h->h_surf_geop.assign(global_field, global_field + count);
  h2d(h, h->d.surf_geop, global_field + (size_t)g.j0 * g.I, (size_t)g.Jl * g.I);
  API_END
}

extern "C" int isca_dyn_cold_start(isca_dyn_t *h) {
  API_BEGIN
  if (!h) fail("null handle");
  if (h->g.P == 1) { cold_start_single(h); return 0; }
  // multi-rank: every rank computes the (cheap) global cold start on its own GPU with a single-rank
  // handle and keeps its latitude band / wavenumber set
  isca_dyn_config c1 = h->cfg;
  c1.rank = 0; c1.world_size = 1; c1.stream = nullptr;
  isca_dyn_t *g1 = nullptr;
  if (isca_dyn_create(&c1, &g1)) fail(std::string("cold start: ") + isca_last_error());
  try {
    if (!h->h_surf_geop.empty() && isca_dyn_set_surf_geopotential(g1, h->h_surf_geop.data(), h->h_surf_geop.size()))
      fail(std::string("cold start: ") + isca_last_error());
    cold_start_single(g1);
    const Geom &g = h->g;
    const Geom &G = g1->g;
    const size_t NG3 = (size_t)G.L * G.J * G.I, NG2 = (size_t)G.J * G.I;
    std::vector<double> big(NG3), loc((size_t)g.L * g.Jl * g.I);
    auto grid3 = [&](double *src, double *dst, int nlev) {
      d2h(g1, big.data(), src, (size_t)nlev * NG2);
      for (int k = 0; k < nlev; ++k)
        std::memcpy(&loc[(size_t)k * g.Jl * g.I], &big[((size_t)k * G.J + g.j0) * G.I], (size_t)g.Jl * g.I * sizeof(double));
      h2d(h, dst, loc.data(), (size_t)nlev * g.Jl * g.I);
    };
    for (int t = 0; t < 2; ++t) {
      grid3(g1->d.ug[t], h->d.ug[t], g.L); grid3(g1->d.vg[t], h->d.vg[t], g.L); grid3(g1->d.tg[t], h->d.tg[t], g.L);
      grid3(g1->d.tr[t], h->d.tr[t], g.L); grid3(g1->d.tr_atm[t], h->d.tr_atm[t], g.L); grid3(g1->d.psg[t], h->d.psg[t], 1);
    }
    if (h->cfg.physics == 1) grid3(g1->d.t_surf, h->d.t_surf, 1);
    grid3(g1->d.vorg, h->d.vorg, g.L); grid3(g1->d.divg, h->d.divg, g.L); grid3(g1->d.dxT, h->d.dxT, g.L);
    grid3(g1->d.dyT, h->d.dyT, g.L); grid3(g1->d.dxlp, h->d.dxlp, 1); grid3(g1->d.dylp, h->d.dylp, 1);
    std::vector<double> sp((size_t)G.L * G.N1 * G.M1 * 2);
    auto spec = [&](double *src, double *dst, int nlev) {
      spec_dev_to_host(g1, src, sp.data(), nlev);
      spec_host_to_dev(h, sp.data(), dst, nlev);
    };
    for (int t = 0; t < 2; ++t) {
      spec(g1->d.vors[t], h->d.vors[t], g.L); spec(g1->d.divs[t], h->d.divs[t], g.L);
      spec(g1->d.ts[t], h->d.ts[t], g.L); spec(g1->d.lnps[t], h->d.lnps[t], 1);
    }
    h->previous = 0; h->current = 0; h->step_count = 0; h->have_state = true; h->phys_calls = 0;
    reset_pending(h);
  } catch (...) { isca_dyn_destroy(g1); throw; }
  isca_dyn_destroy(g1);
  API_END
}

// ---------------------------------------------------------------------------------------------------
extern "C" int isca_dyn_get_table(isca_dyn_t *h, const char *name, double *host, size_t count) {
  API_BEGIN
  const Tables &T = h->tab;
  const std::string nm(name);
  const std::vector<double> *v = nullptr;
  std::vector<double> tmp;
  if (nm == "sin_lat") v = &T.sin_lat; else if (nm == "wts_lat") v = &T.wts_lat; else if (nm == "deg_lat") v = &T.deg_lat;
  else if (nm == "deg_lon") v = &T.deg_lon; else if (nm == "pk") v = &T.pk; else if (nm == "bk") v = &T.bk;
  else if (nm == "legendre") v = &T.legendre; else if (nm == "eigen_laplacian") v = &T.eigen;
  else if (nm == "sin_hem") v = &T.sin_hem; else if (nm == "wts_hem") v = &T.wts_hem;
  else if (nm == "lon_boundaries") v = &T.lon_boundaries; else if (nm == "lat_boundaries") v = &T.lat_boundaries;     // transforms.F90:313-325 (radians)
  else if (nm == "wave_matrix") {
    if (T.wave_dt < 0) fail("wave matrices not built yet (run a step)");
    const int L = T.L, nw = T.n_wave;
    tmp.resize((size_t)nw * L * L);
    for (int w = 0; w < nw; ++w) for (int k = 0; k < L; ++k) for (int k2 = 0; k2 < L; ++k2)
      tmp[((size_t)w * L + k2) * L + k] = T.wave_matrix[((size_t)w * L + k) * L + k2];   // Fortran (k,k2,w)
    v = &tmp;
  } else if (nm == "fixer") {
    tmp.resize(32);
    flush_finish(h);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    HIP_CHECK(hipMemcpy(tmp.data(), h->d.red, 32 * sizeof(double), hipMemcpyDeviceToHost));
    v = &tmp;
  }
  if (!v) fail("unknown table " + nm);
  if (v->size() != count) fail("get_table: wrong element count for " + nm + " (have " + std::to_string(v->size()) + ")");
  std::memcpy(host, v->data(), count * sizeof(double));
  API_END
}
extern "C" int isca_dyn_get_info(isca_dyn_t *h, const char *name, long *value) {
  API_BEGIN
  const std::string nm(name);
  if (nm == "step") *value = h->step_count; else if (nm == "previous") *value = h->previous;
  else if (nm == "current") *value = h->current; else if (nm == "lat_local") *value = h->g.Jl;
  else if (nm == "lat_start") *value = h->g.j0;
  else if (nm == "m_local") *value = h->g.Ml; else if (nm == "kernels_per_step") *value = h->kernels_per_step;
  else if (nm == "tracer") *value = h->tracer_on ? 1 : 0;
  else if (nm == "tracer_env_off") *value = h->tracer_env_off ? 1 : 0;
  else if (nm == "cf") *value = h->Cf; else if (nm == "ci") *value = h->Ci;
  else if (nm == "inverse_batch") *value = h->dx_fourier ? 6 * h->g.L + 2 : 7 * h->g.L + 3;      // level-fields of the step's Legendre synthesis
  else if (nm == "phys_calls") *value = h->phys_calls;
  else if (nm == "lazy_fixers") *value = h->lazy_fix ? 1 : 0;
  else fail("unknown info " + nm);
  API_END
}

// Handing over a RUNNING model (both time levels set through set_state, not a restart): what idealized_moist_phys_mod keeps beside the fields.
// "phys_calls": calls of the physics package since idealized_moist_phys_init -- 0 makes the next call the first one (gust = 1 m/s,
// idealized_moist_phys.F90:592; what set_time_pointers leaves, as a restart does), > 0 continues with vert_turb_driver's constant_gust (:1262).
extern "C" int isca_dyn_set_info(isca_dyn_t *h, const char *name, long value) {
  API_BEGIN
  if (!h || !name) fail("null argument");
  const std::string nm(name);
  if (nm == "phys_calls") {
    if (value < 0) fail("set_info: phys_calls must not be negative");
    h->phys_calls = value;
  } else fail("set_info: unknown or read-only info " + nm);
  API_END
}

// ---------------------------------------------------------------------------------------------------
// transforms_mod entry points on host buffers
// ---------------------------------------------------------------------------------------------------
static void check_nlev(isca_dyn *h, int nlev) {
  if (nlev < 1 || nlev > h->g.L + 1) fail("nlev must be between 1 and num_levels+1");
}
extern "C" int isca_trans_spherical_to_grid(isca_dyn_t *h, const double *spherical, double *grid, int nlev) {
  API_BEGIN
  require_single_or_comm(h, "trans_spherical_to_grid"); check_nlev(h, nlev);
  spec_host_to_dev(h, spherical, h->d.scratch_s[0], nlev);
  dev_s2g(h, h->d.scratch_s[0], h->d.scratch_g[0], nlev, OP_NONE);
  d2h(h, grid, h->d.scratch_g[0], (size_t)nlev * h->g.Jl * h->g.I);
  API_END
}
extern "C" int isca_trans_grid_to_spherical(isca_dyn_t *h, const double *grid, double *spherical, int nlev, int do_truncation) {
  API_BEGIN
  require_single_or_comm(h, "trans_grid_to_spherical"); check_nlev(h, nlev);
  h2d(h, h->d.scratch_g[0], grid, (size_t)nlev * h->g.Jl * h->g.I);
  dev_g2s(h, h->d.scratch_g[0], h->d.scratch_s[0], nlev, do_truncation, OP_NONE);
  spec_dev_to_host_all(h, h->d.scratch_s[0], spherical, nlev);
  API_END
}
// transforms.F90:555-596 trans_filter(grid [, filter]): grid -> spherical (truncated) -> optional real (m,n) factor -> grid
extern "C" int isca_trans_filter(isca_dyn_t *h, double *grid, const double *filter, int nlev) {
  API_BEGIN
  require_single_or_comm(h, "trans_filter"); check_nlev(h, nlev);
  const Geom &g = h->g;
  const size_t n = (size_t)nlev * g.Jl * g.I;
  h2d(h, h->d.scratch_g[0], grid, n);
  dev_g2s(h, h->d.scratch_g[0], h->d.scratch_s[0], nlev, 1, OP_NONE);
  if (filter) {
    std::vector<double> sp((size_t)nlev * g.N1 * g.M1 * 2);
    spec_dev_to_host(h, h->d.scratch_s[0], sp.data(), nlev);        // (my wavenumbers: all that goes back to the device)
    for (int k = 0; k < nlev; ++k)
      for (size_t q = 0; q < (size_t)g.N1 * g.M1; ++q) {
        sp[((size_t)k * g.N1 * g.M1 + q) * 2] *= filter[q];
        sp[((size_t)k * g.N1 * g.M1 + q) * 2 + 1] *= filter[q];
      }
    spec_host_to_dev(h, sp.data(), h->d.scratch_s[0], nlev);
  }
  dev_s2g(h, h->d.scratch_s[0], h->d.scratch_g[0], nlev, OP_NONE);
  d2h(h, grid, h->d.scratch_g[0], n);
  API_END
}
extern "C" int isca_vor_div_from_uv_grid(isca_dyn_t *h, const double *u, const double *v, double *vor, double *div, int nlev) {
  API_BEGIN
  require_single_or_comm(h, "vor_div_from_uv_grid"); check_nlev(h, nlev);
  const size_t n = (size_t)nlev * h->g.Jl * h->g.I;
  h2d(h, h->d.scratch_g[0], u, n); h2d(h, h->d.scratch_g[1], v, n);
  dev_vd_from_uv(h, h->d.scratch_g[0], h->d.scratch_g[1], h->d.scratch_s[0], h->d.scratch_s[1], nlev);
  spec_dev_to_host_all(h, h->d.scratch_s[0], vor, nlev); spec_dev_to_host_all(h, h->d.scratch_s[1], div, nlev);
  API_END
}
extern "C" int isca_uv_grid_from_vor_div(isca_dyn_t *h, const double *vor, const double *div, double *u, double *v, int nlev) {
  API_BEGIN
  require_single_or_comm(h, "uv_grid_from_vor_div"); check_nlev(h, nlev);
  const size_t n = (size_t)nlev * h->g.Jl * h->g.I;
  spec_host_to_dev(h, vor, h->d.scratch_s[0], nlev); spec_host_to_dev(h, div, h->d.scratch_s[1], nlev);
  dev_uv_from_vd(h, h->d.scratch_s[0], h->d.scratch_s[1], h->d.scratch_g[0], h->d.scratch_g[1], nlev);
  d2h(h, u, h->d.scratch_g[0], n); d2h(h, v, h->d.scratch_g[1], n);
  API_END
}
extern "C" int isca_horizontal_advection(isca_dyn_t *h, const double *field_spec, const double *u, const double *v, double *tendency, int nlev) {
  API_BEGIN
  require_single(h, "horizontal_advection"); check_nlev(h, nlev);
  if (nlev > h->g.L) fail("horizontal_advection: nlev must be <= num_levels");
  const size_t n = (size_t)nlev * h->g.Jl * h->g.I;
  Dev &d = h->d;
  spec_host_to_dev(h, field_spec, d.scratch_s[0], nlev);
  FieldList fl = pair_list(d.scratch_g[0], d.scratch_g[1], nlev, OP_COSM);
  launch_spec_gradient(h->g, d, d.scratch_s[0], d.Si, col_pitch(fl.ncol), 0, nlev, nlev, h->stream);
  run_inverse(h, fl, 1);
  h2d(h, d.scratch_g[2], u, n); h2d(h, d.scratch_g[3], v, n);
  // tendency buffer: reuse g_E as scratch only when no step is in flight (host-synchronous API)
  double *tend = d.g_E;
  h2d(h, tend, tendency, n);
  launch_hadv_combine(h->g, d.scratch_g[2], d.scratch_g[3], d.scratch_g[0], d.scratch_g[1], tend, nlev, h->stream);
  d2h(h, tendency, tend, n);
  API_END
}
// Legendre / Fourier stages exposed separately (spherical_fourier.F90:177,264; grid_fourier.F90:129,155).
// host fourier layout: (m, lat, lev) complex with m = 0..num_fourier
static void fourier_host_to_dev(isca_dyn *h, const double *host, double *dev, int nlev) {
  const Geom &g = h->g;
  const int C = col_pitch(nlev);
  std::vector<double> tmp((size_t)g.Ml * g.Jl * C, 0.0);
  for (int k = 0; k < nlev; ++k) for (int j = 0; j < g.J; ++j) for (int m = 0; m < g.M1; ++m) {
    const size_t src = (((size_t)k * g.J + j) * g.M1 + m) * 2, dst = ((size_t)h->h_slot_of_m[m] * g.Jl + j) * C + 2 * k;
    tmp[dst] = host[src]; tmp[dst + 1] = host[src + 1];
  }
  h2d(h, dev, tmp.data(), tmp.size());
}
static void fourier_dev_to_host(isca_dyn *h, const double *dev, double *host, int nlev) {
  const Geom &g = h->g;
  const int C = col_pitch(nlev);
  std::vector<double> tmp((size_t)g.Ml * g.Jl * C);
  d2h(h, tmp.data(), dev, tmp.size());
  for (int k = 0; k < nlev; ++k) for (int j = 0; j < g.J; ++j) for (int m = 0; m < g.M1; ++m) {
    const size_t dst = (((size_t)k * g.J + j) * g.M1 + m) * 2, src = ((size_t)h->h_slot_of_m[m] * g.Jl + j) * C + 2 * k;
    host[dst] = tmp[src]; host[dst + 1] = tmp[src + 1];
  }
}
extern "C" int isca_trans_spherical_to_fourier(isca_dyn_t *h, const double *spherical, double *fourier, int nlev) {
  API_BEGIN
  require_single(h, "trans_spherical_to_fourier"); check_nlev(h, nlev);
  spec_host_to_dev(h, spherical, h->d.scratch_s[0], nlev);
  launch_spec_pack(h->g, h->d.scratch_s[0], h->d.Si, col_pitch(nlev), 0, nlev, h->stream);
  launch_legendre_inverse(h->g, h->d, h->d.Si, h->d.Fi_s, col_pitch(nlev), 1, h->cfg.legendre_impl, h->stream);
  fourier_dev_to_host(h, h->d.Fi_s, fourier, nlev);
  API_END
}
extern "C" int isca_trans_fourier_to_spherical(isca_dyn_t *h, const double *fourier, double *spherical, int nlev) {
  API_BEGIN
  require_single(h, "trans_fourier_to_spherical"); check_nlev(h, nlev);
  fourier_host_to_dev(h, fourier, h->d.Ff_s, nlev);
  launch_legendre_forward(h->g, h->d, h->d.Ff_s, h->d.Sf, col_pitch(nlev), 1, h->cfg.legendre_impl, h->stream);
  launch_spec_unpack(h->g, h->d, h->d.Sf, h->d.scratch_s[0], col_pitch(nlev), 0, nlev, 0, h->stream);
  spec_dev_to_host(h, h->d.scratch_s[0], spherical, nlev);
  API_END
}
extern "C" int isca_trans_grid_to_fourier(isca_dyn_t *h, const double *grid, double *fourier, int nlev) {
  API_BEGIN
  require_single(h, "trans_grid_to_fourier"); check_nlev(h, nlev);
  h2d(h, h->d.scratch_g[0], grid, (size_t)nlev * h->g.Jl * h->g.I);
  FieldList fl = single_list(h->d.scratch_g[0], nlev, OP_NONE);
  launch_fft_forward(h->g, h->d, fl, h->d.Ff_g, h->stream);
  fourier_dev_to_host(h, h->d.Ff_g, fourier, nlev);
  API_END
}
extern "C" int isca_trans_fourier_to_grid(isca_dyn_t *h, const double *fourier, double *grid, int nlev) {
  API_BEGIN
  require_single(h, "trans_fourier_to_grid"); check_nlev(h, nlev);
  fourier_host_to_dev(h, fourier, h->d.Fi_g, nlev);
  FieldList fl = single_list(h->d.scratch_g[0], nlev, OP_NONE);
  launch_fft_inverse(h->g, h->d, fl, h->d.Fi_g, h->stream);
  d2h(h, grid, h->d.scratch_g[0], (size_t)nlev * h->g.Jl * h->g.I);
  API_END
}
extern "C" int isca_area_weighted_global_mean(isca_dyn_t *h, const double *field2d, double *mean) {
  API_BEGIN
  // transforms.F90:1059-1077 -- a host-side convenience (the step uses the fused device reduction)
  const Tables &T = h->tab;
  double s = 0.0, sw = 0.0;
  for (int j = 0; j < T.J; ++j) { sw += T.wts_lat[j]; for (int i = 0; i < T.I; ++i) s += T.wts_lat[j] * field2d[(size_t)j * T.I + i]; }
  *mean = s / (sw * T.I);
  API_END
}
extern "C" int isca_hs_forcing(isca_dyn_t *h, double dt, const double *p_half, const double *p_full, const double *u,
                               const double *v, const double *t, double *udt, double *vdt, double *tdt) {
  API_BEGIN
  const Geom &g = h->g;
  const size_t n3 = (size_t)g.L * g.Jl * g.I, n3h = (size_t)(g.L + 1) * g.Jl * g.I;
  std::vector<double *> dv;
  auto up = [&](const double *src, size_t n) { double *p; HIP_CHECK(hipMalloc((void **)&p, n * sizeof(double))); h2d(h, p, src, n); dv.push_back(p); return p; };
  double *dph = up(p_half, n3h), *dpf = up(p_full, n3), *du = up(u, n3), *dvv = up(v, n3), *dt_ = up(t, n3);
  double *dud = up(udt, n3), *dvd = up(vdt, n3), *dtd = up(tdt, n3);
  launch_hs_forcing(*h, dt, dph, dpf, du, dvv, dt_, dud, dvd, dtd, h->stream);
  d2h(h, udt, dud, n3); d2h(h, vdt, dvd, n3); d2h(h, tdt, dtd, n3);
  for (double *p : dv) hipFree(p);
  API_END
}

// ---------------------------------------------------------------------------------------------------
// Components of the step on caller fields.  These run the kernels (or the device functions) the step
// itself uses, so the reference's per-routine outputs can be compared directly.
// ---------------------------------------------------------------------------------------------------
namespace {
struct DevTmp {      // short-lived device buffers of one host-synchronous call
  isca_dyn *h; std::vector<double *> v;
  explicit DevTmp(isca_dyn *h_) : h(h_) {}
  double *alloc(size_t n, bool zero = false) {
    double *p; HIP_CHECK(hipMalloc((void **)&p, n * sizeof(double)));
    v.push_back(p);
    if (zero) HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(double), h->stream));
    return p;
  }
  double *up(const double *src, size_t n) { double *p = alloc(n); h2d(h, p, src, n); return p; }
  ~DevTmp() { hipStreamSynchronize(h->stream); for (double *p : v) hipFree(p); }
};
}  // namespace

// idealized_moist_phys on caller columns: host arrays [lev][ncol] (Fortran (col, lev)), half levels [lev+1][ncol]
extern "C" int isca_idealized_moist_phys(isca_dyn_t *h, int ncol, double delta_t, double gust, const double *rad_lat, const double *u_prev,
                                         const double *v_prev, const double *t_prev, const double *q_prev, const double *p_half_prev,
                                         const double *p_full_prev, const double *p_half_cur, const double *p_full_cur,
                                         const double *z_half_cur, const double *z_full_cur, double *t_surf, double *dt_u, double *dt_v,
                                         double *dt_t, double *dt_q, double *precip) {
  API_BEGIN
  if (!h) fail("null handle");
  if (h->cfg.physics != 1) fail("idealized_moist_phys: the handle was not created with physics = 1");
  if (ncol <= 0) fail("idealized_moist_phys: ncol must be positive");
  const int L = h->g.L;
  const size_t nf = (size_t)ncol * L, nh = (size_t)ncol * (L + 1);
  DevTmp tmp(h);
  const double *u = tmp.up(u_prev, nf), *v = tmp.up(v_prev, nf), *t = tmp.up(t_prev, nf), *q = tmp.up(q_prev, nf), *php = tmp.up(p_half_prev, nh),
               *pfp = tmp.up(p_full_prev, nf), *phc = tmp.up(p_half_cur, nh), *pfc = tmp.up(p_full_cur, nf), *zhc = tmp.up(z_half_cur, nh),
               *zfc = tmp.up(z_full_cur, nf), *lat = tmp.up(rad_lat, ncol);
  double *ts = tmp.up(t_surf, ncol), *du = tmp.alloc(nf), *dv = tmp.alloc(nf), *dt = tmp.alloc(nf), *dq = tmp.alloc(nf), *pr = tmp.alloc(ncol), *wk = tmp.alloc(5 * nh), *cc = tmp.alloc(2 * nf + ncol);
  launch_moist_physics_on(*h, ncol, delta_t, gust, lat, u, v, t, q, php, pfp, phc, pfc, zhc, zfc, ts, du, dv, dt, dq, pr, wk, cc, h->stream);
  d2h(h, t_surf, ts, ncol); d2h(h, dt_u, du, nf); d2h(h, dt_v, dv, nf); d2h(h, dt_t, dt, nf); d2h(h, dt_q, dq, nf);
  if (precip) d2h(h, precip, pr, ncol);
  API_END
}

// spherical.F90:354-406 compute_laplacian(spherical [, power])
extern "C" int isca_compute_laplacian(isca_dyn_t *h, const double *spherical, double *laplacian, int nlev, int power) {
  API_BEGIN
  check_nlev(h, nlev);
  spec_host_to_dev(h, spherical, h->d.scratch_s[0], nlev);
  launch_spec_laplacian(h->g, h->d, h->d.scratch_s[0], h->d.scratch_s[1], nlev, power, h->stream);
  spec_dev_to_host(h, h->d.scratch_s[1], laplacian, nlev);
  API_END
}
// spherical.F90:270-351 compute_gradient_cos / compute_lon_deriv_cos / compute_lat_deriv_cos (either output may be NULL)
extern "C" int isca_compute_gradient_cos(isca_dyn_t *h, const double *spherical, double *deriv_lon, double *deriv_lat, int nlev) {
  API_BEGIN
  check_nlev(h, nlev);
  Dev &d = h->d;
  spec_host_to_dev(h, spherical, d.scratch_s[0], nlev);
  launch_spec_gradient(h->g, d, d.scratch_s[0], d.Si, 4 * nlev, 0, nlev, nlev, h->stream);
  launch_spec_unpack(h->g, d, d.Si, d.scratch_s[1], 4 * nlev, 0, nlev, 0, h->stream);
  launch_spec_unpack(h->g, d, d.Si, d.scratch_s[2], 4 * nlev, nlev, nlev, 0, h->stream);
  if (deriv_lon) spec_dev_to_host(h, d.scratch_s[1], deriv_lon, nlev);
  if (deriv_lat) spec_dev_to_host(h, d.scratch_s[2], deriv_lat, nlev);
  API_END
}
// spherical.F90:409-469 compute_ucos_vcos
extern "C" int isca_compute_ucos_vcos(isca_dyn_t *h, const double *vorticity, const double *divergence, double *u_cos, double *v_cos, int nlev) {
  API_BEGIN
  check_nlev(h, nlev);
  Dev &d = h->d;
  spec_host_to_dev(h, vorticity, d.scratch_s[0], nlev); spec_host_to_dev(h, divergence, d.scratch_s[1], nlev);
  launch_spec_ucos_vcos(h->g, d, d.scratch_s[0], d.scratch_s[1], d.Si, 4 * nlev, 0, nlev, nlev, h->stream);
  launch_spec_unpack(h->g, d, d.Si, d.scratch_s[2], 4 * nlev, 0, nlev, 0, h->stream);
  launch_spec_unpack(h->g, d, d.Si, d.scratch_s[3], 4 * nlev, nlev, nlev, 0, h->stream);
  spec_dev_to_host(h, d.scratch_s[2], u_cos, nlev); spec_dev_to_host(h, d.scratch_s[3], v_cos, nlev);
  API_END
}
// spherical.F90:472-561 compute_vor_div (no truncation, unlike vor_div_from_uv_grid)
extern "C" int isca_compute_vor_div(isca_dyn_t *h, const double *u_div_cos, const double *v_div_cos, double *vorticity, double *divergence, int nlev) {
  API_BEGIN
  check_nlev(h, nlev);
  Dev &d = h->d;
  spec_host_to_dev(h, u_div_cos, d.scratch_s[0], nlev); spec_host_to_dev(h, v_div_cos, d.scratch_s[1], nlev);
  launch_spec_pack(h->g, d.scratch_s[0], d.Si, 4 * nlev, 0, nlev, h->stream);
  launch_spec_pack(h->g, d.scratch_s[1], d.Si, 4 * nlev, nlev, nlev, h->stream);
  launch_spec_vor_div(h->g, d, d.Si, 4 * nlev, 0, nlev, d.scratch_s[2], d.scratch_s[3], nlev, h->stream, 0);
  spec_dev_to_host(h, d.scratch_s[2], vorticity, nlev); spec_dev_to_host(h, d.scratch_s[3], divergence, nlev);
  API_END
}
// spherical.F90:564-600 triangular_truncation (default mask), in place
extern "C" int isca_triangular_truncation(isca_dyn_t *h, double *spherical, int nlev) {
  API_BEGIN
  check_nlev(h, nlev);
  if (!h->cfg.triang_trunc) fail("triangular_truncation: the handle carries the rhomboidal mask (triang_trunc = .false.)");
  Dev &d = h->d;
  spec_host_to_dev(h, spherical, d.scratch_s[0], nlev);
  launch_spec_pack(h->g, d.scratch_s[0], d.Si, 2 * nlev, 0, nlev, h->stream);
  launch_spec_unpack(h->g, d, d.Si, d.scratch_s[1], 2 * nlev, 0, nlev, 1, h->stream);
  spec_dev_to_host(h, d.scratch_s[1], spherical, nlev);
  API_END
}
// transforms.F90:599-648 divide_by_cos / divide_by_cos2, in place (power = 1 or 2)
extern "C" int isca_divide_by_cos(isca_dyn_t *h, double *grid, int nlev, int power) {
  API_BEGIN
  check_nlev(h, nlev);
  if (power != 1 && power != 2) fail("divide_by_cos: power must be 1 or 2");
  const size_t n = (size_t)nlev * h->g.Jl * h->g.I;
  h2d(h, h->d.scratch_g[0], grid, n);
  for (int p = 0; p < power; ++p) launch_scale_rows(h->g, h->d, h->d.scratch_g[0], nlev, h->stream);
  d2h(h, grid, h->d.scratch_g[0], n);
  API_END
}
// global_integral.F90:49-81 mass_weighted_global_integral(field, surf_press)
extern "C" int isca_mass_weighted_global_integral(isca_dyn_t *h, const double *field, const double *surf_press, double *integral) {
  API_BEGIN
  require_single(h, "mass_weighted_global_integral");
  const Geom &g = h->g;
  const size_t n2 = (size_t)g.Jl * g.I;
  DevTmp t(h);
  double *f = t.up(field, n2 * g.L), *ps = t.up(surf_press, n2), *rows = t.alloc(g.Jl);
  launch_mass_weighted_rows(*h, f, ps, rows, h->stream);
  std::vector<double> r(g.Jl);
  d2h(h, r.data(), rows, g.Jl);
  double s_ = 0.0, sw = 0.0;
  for (int j = 0; j < g.Jl; ++j) { s_ += r[j]; sw += h->tab.wts_lat[j]; }
  *integral = s_ / (sw * g.I) / GRAV;
  API_END
}
// press_and_geopot.F90:152-221 pressure_variables(p_half, ln_p_half, p_full, ln_p_full, surf_p)
extern "C" int isca_pressure_variables(isca_dyn_t *h, const double *surf_p, double *p_half, double *ln_p_half, double *p_full, double *ln_p_full) {
  API_BEGIN
  const Geom &g = h->g;
  const size_t n2 = (size_t)g.Jl * g.I;
  DevTmp t(h);
  double *ps = t.up(surf_p, n2), *ph = t.alloc(n2 * (g.L + 1)), *lph = t.alloc(n2 * (g.L + 1)), *pf = t.alloc(n2 * g.L), *lpf = t.alloc(n2 * g.L);
  launch_pressure_variables(*h, ps, ph, lph, pf, lpf, h->stream);
  d2h(h, p_half, ph, n2 * (g.L + 1)); d2h(h, ln_p_half, lph, n2 * (g.L + 1)); d2h(h, p_full, pf, n2 * g.L); d2h(h, ln_p_full, lpf, n2 * g.L);
  API_END
}
// press_and_geopot.F90:327-359 compute_geopotential(t, ln_p_half, ln_p_full, surf_geopotential = 0, geopot_full, geopot_half)
extern "C" int isca_compute_geopotential(isca_dyn_t *h, const double *t, const double *ln_p_half, const double *ln_p_full, double *geopot_full, double *geopot_half) {
  API_BEGIN
  const Geom &g = h->g;
  const size_t n2 = (size_t)g.Jl * g.I;
  DevTmp tmp(h);
  double *dt_ = tmp.up(t, n2 * g.L), *lph = tmp.up(ln_p_half, n2 * (g.L + 1)), *lpf = tmp.up(ln_p_full, n2 * g.L);
  double *gf = tmp.alloc(n2 * g.L), *gh = tmp.alloc(n2 * (g.L + 1), true);
  launch_geopotential(*h, dt_, lph, lpf, gf, gh, h->stream);
  d2h(h, geopot_full, gf, n2 * g.L); d2h(h, geopot_half, gh, n2 * (g.L + 1));
  API_END
}
// the same with the caller's arguments: surf_geopotential (null: the handle's own), q_grid (null: none given) -- press_and_geopot.F90:314-359
extern "C" int isca_compute_geopotential_surf(isca_dyn_t *h, const double *t, const double *ln_p_half, const double *ln_p_full, const double *surf_geopotential,
                                              const double *q, double *geopot_full, double *geopot_half) {
  API_BEGIN
  const Geom &g = h->g;
  const size_t n2 = (size_t)g.Jl * g.I;
  if (h->cfg.use_virtual_temperature && !q) fail("compute_geopotential: q_grid must be present when use_virtual_temperature=.true.");      // :343
  DevTmp tmp(h);
  double *dt_ = tmp.up(t, n2 * g.L), *lph = tmp.up(ln_p_half, n2 * (g.L + 1)), *lpf = tmp.up(ln_p_full, n2 * g.L);
  double *sg = surf_geopotential ? tmp.up(surf_geopotential, n2) : nullptr;
  if (h->cfg.use_virtual_temperature) {          // virtual_t = t_grid (1 + (rvgas/rdgas - 1) q_grid) (:340-341)
    double *dq = tmp.up(q, n2 * g.L), *tv = tmp.alloc(n2 * g.L);
    launch_virtual_t(*h, dt_, dq, tv, h->stream);
    dt_ = tv;
  }
  double *gf = tmp.alloc(n2 * g.L), *gh = tmp.alloc(n2 * (g.L + 1), true);
  launch_geopotential(*h, dt_, lph, lpf, gf, gh, h->stream, sg);
  d2h(h, geopot_full, gf, n2 * g.L); d2h(h, geopot_half, gh, n2 * (g.L + 1));
  API_END
}
// fv_advection.F90:126-207 a_grid_horiz_advection(u, v, q, dt, tendency): tendency += van Leer advective tendency.
// Runs the step's own tracer kernel without source/sink.
extern "C" int isca_a_grid_horiz_advection(isca_dyn_t *h, const double *u, const double *v, const double *q, double dt, double *tendency) {
  API_BEGIN
  require_single(h, "a_grid_horiz_advection");
  const Geom &g = h->g;
  if (g.Jl < 4) fail("a_grid_horiz_advection: needs at least 4 latitude rows");
  const size_t n3 = (size_t)g.L * g.Jl * g.I;
  DevTmp t(h);
  double *du = t.up(u, n3), *dv = t.up(v, n3), *dq = t.up(q, n3), *qn = t.alloc(n3);
  const std::vector<double> ones((size_t)g.Jl * g.I, h->cfg.reference_sea_level_press);
  double *ps = t.up(ones.data(), ones.size());
  launch_fv_horiz_on(*h, du, dv, dq, ps, dt, qn, h->stream);
  std::vector<double> out(n3);
  d2h(h, out.data(), qn, n3);
  for (size_t i = 0; i < n3; ++i) tendency[i] += (out[i] - q[i]) / dt;
  API_END
}
// vert_advection.F90:70-478 vert_advection(dt, w, dz, r, rdt, scheme = FINITE_VOLUME_PARABOLIC, form = ADVECTIVE_FORM)
// with dz = dpk + dbk * surf_p (pure sigma levels).  Runs the step's own tracer kernel.
extern "C" int isca_vert_advection_ppm(isca_dyn_t *h, double dt, const double *w, const double *surf_p, const double *r, double *rdt) {
  API_BEGIN
  const Geom &g = h->g;
  if (!h->d.fv_c) fail("vert_advection_ppm: not available");
  const size_t n2 = (size_t)g.Jl * g.I, n3 = n2 * g.L;
  DevTmp t(h);
  double *dw = t.up(w, n2 * (g.L + 1)), *ps = t.up(surf_p, n2), *dr = t.up(r, n3), *rn = t.alloc(n3);
  double *da = t.alloc(n3, true), *db = t.alloc(n3, true);
  launch_ppm_vert_on(*h, dt, dw, ps, dr, rn, da, db, h->stream);
  std::vector<double> out(n3);
  d2h(h, out.data(), rn, n3);
  for (size_t i = 0; i < n3; ++i) rdt[i] = (out[i] - r[i]) / dt;
  API_END
}
// hs_forcing.F90:683-724 tracer_source_sink(flux, sink, p_half, r, rdt): rdt += source - sink
extern "C" int isca_hs_tracer_source_sink(isca_dyn_t *h, const double *surf_p, const double *r, double *rdt) {
  API_BEGIN
  const Geom &g = h->g;
  const size_t n2 = (size_t)g.Jl * g.I, n3 = n2 * g.L;
  DevTmp t(h);
  double *ps = t.up(surf_p, n2), *dr = t.up(r, n3), *dd = t.up(rdt, n3);
  launch_tracer_source_sink(*h, ps, dr, dd, h->stream);
  d2h(h, rdt, dd, n3);
  API_END
}

// vert_advection.F90:70-478 vert_advection(dt, w, dz, r, rdt, scheme = SECOND_CENTERED, form = ADVECTIVE_FORM), dz = dpk + dbk * surf_p
extern "C" int isca_vert_advection_centered(isca_dyn_t *h, const double *w, const double *surf_p, const double *r, double *rdt) {
  API_BEGIN
  const Geom &g = h->g;
  const size_t n2 = (size_t)g.Jl * g.I, n3 = n2 * g.L;
  DevTmp t(h);
  double *dw = t.up(w, n2 * (g.L + 1)), *ps = t.up(surf_p, n2), *dr = t.up(r, n3), *dd = t.alloc(n3, true);
  launch_vert_advection_centered(*h, dw, ps, dr, dd, h->stream);
  d2h(h, rdt, dd, n3);
  API_END
}
// press_and_geopot.F90:363-387 compute_pressures_and_heights on caller fields (the handle's surface geopotential)
extern "C" int isca_compute_pressures_and_heights(isca_dyn_t *h, const double *t, const double *ps, const double *q, double *z_full, double *z_half,
                                                  double *p_full, double *p_half) {
  API_BEGIN
  if (!h || !t || !ps) fail("null argument");
  const Geom &g = h->g;
  const size_t n2 = (size_t)g.Jl * g.I, n3 = n2 * g.L, n3h = n2 * (g.L + 1);
  DevTmp tmp(h);
  double *dt_ = tmp.up(t, n3), *dps = tmp.up(ps, n2), *pf = tmp.alloc(n3), *ph = tmp.alloc(n3h), *zf = tmp.alloc(n3), *zh = tmp.alloc(n3h);
  const double *tq = dt_;
  if (q && h->cfg.use_virtual_temperature) {         // press_and_geopot.F90:246-256, 340-355
    double *dq = tmp.up(q, n3), *tv = tmp.alloc(n3);
    launch_virtual_t(*h, dt_, dq, tv, h->stream);
    tq = tv;
  }
  launch_pressures_heights(*h, tq, dps, pf, ph, zf, zh, h->stream);
  if (p_full) d2h(h, p_full, pf, n3);
  if (p_half) d2h(h, p_half, ph, n3h);
  if (z_full) d2h(h, z_full, zf, n3);
  if (z_half) d2h(h, z_half, zh, n3h);
  API_END
}
// leapfrog.F90:58-105 on caller arrays
extern "C" int isca_leapfrog_2level_a(isca_dyn_t *h, size_t n, const double *prev, double *cur, double *fut, const double *dt_a, double delta_t,
                                      double robert_coeff, double raw_filter_coeff, double *part) {
  API_BEGIN
  if (!h || !prev || !cur || !fut || !dt_a) fail("null argument");
  DevTmp t(h);
  double *dc = t.up(cur, n), *dp = (prev == cur) ? dc : t.up(prev, n);
  double *df = (fut == prev) ? dp : ((fut == cur) ? dc : t.alloc(n));
  double *dd = t.up(dt_a, n), *dpt = part ? t.alloc(n) : nullptr;
  launch_leapfrog_a(n, dp, dc, df, dd, delta_t, robert_coeff, raw_filter_coeff, dpt, h->stream);
  d2h(h, cur, dc, n);
  if (fut != cur) d2h(h, fut, df, n);
  if (part) d2h(h, part, dpt, n);
  API_END
}
extern "C" int isca_leapfrog_2level_b(isca_dyn_t *h, size_t n, double *cur, double *fut, const double *part, double robert_coeff, double raw_filter_coeff) {
  API_BEGIN
  if (!h || !cur || !fut || !part) fail("null argument");
  DevTmp t(h);
  double *dc = t.up(cur, n), *df = t.up(fut, n), *dp = t.up(part, n);
  launch_leapfrog_b(n, dc, df, dp, robert_coeff, raw_filter_coeff, h->stream);
  d2h(h, cur, dc, n); d2h(h, fut, df, n);
  API_END
}
// gauss_and_legendre.F90: host tables (init-time in the reference too)
extern "C" int isca_compute_gaussian(int n_hem, double *sin_hem, double *wts_hem) {
  API_BEGIN
  if (n_hem < 1 || !sin_hem || !wts_hem) fail("compute_gaussian: invalid argument");
  std::vector<double> s_, w_;
  compute_gaussian(n_hem, s_, w_);
  std::memcpy(sin_hem, s_.data(), n_hem * sizeof(double)); std::memcpy(wts_hem, w_.data(), n_hem * sizeof(double));
  API_END
}
extern "C" int isca_compute_legendre(int num_fourier, int fourier_inc, int num_spherical, const double *sin_lat, int n_lat, double *legendre) {
  API_BEGIN
  if (num_fourier < 0 || num_spherical < 0 || fourier_inc < 1 || n_lat < 1 || !sin_lat || !legendre) fail("compute_legendre: invalid argument");
  std::vector<double> leg;
  compute_legendre(num_fourier, num_spherical, std::vector<double>(sin_lat, sin_lat + n_lat), leg, fourier_inc);
  std::memcpy(legendre, leg.data(), leg.size() * sizeof(double));
  API_END
}

// The three stages of the spectral update on caller data, run by the step's own kernel (k_spec_update).
static void spec_stage(isca_dyn *h, int stage, double delta_t, double robert, const double *const host_st[4][3],
                       double *const host_dt[4], bool want_state) {
  require_single(h, "spectral stage");
  const Geom &g = h->g;
  if (g.L > 64) fail("spectral stages: num_levels <= 64");
  const size_t n3 = (size_t)g.Ml * g.N1 * g.L * 2, n2 = (size_t)g.Ml * g.N1 * 2;
  DevTmp t(h);
  double *st[4][3], *dt[4];
  for (int v = 0; v < 4; ++v) {
    const int nlev = (v == 3) ? 1 : g.L;
    for (int tl = 0; tl < 3; ++tl) {
      st[v][tl] = (tl == 2) ? st[v][0] : t.alloc(v == 3 ? n2 : n3, true);     // future shares the previous slot (leapfrog.F90:58)
      if (tl < 2 && host_st[v][tl]) spec_host_to_dev(h, host_st[v][tl], st[v][tl], nlev);
    }
    dt[v] = t.alloc(v == 3 ? n2 : n3, true);
    if (host_dt[v]) spec_host_to_dev(h, host_dt[v], dt[v], nlev);
  }
  if (stage == 0) upload_wave_matrices(h, delta_t);
  launch_spec_update_stage(*h, stage, delta_t, robert, st, dt, h->stream);
  for (int v = 0; v < 4; ++v) {
    const int nlev = (v == 3) ? 1 : g.L;
    if (!want_state && host_dt[v]) spec_dev_to_host(h, dt[v], host_dt[v], nlev);
    if (want_state) for (int tl = 0; tl < 2; ++tl)
      if (host_st[v][tl]) spec_dev_to_host(h, st[v][tl], const_cast<double *>(host_st[v][tl]), nlev);
  }
}
// implicit.F90:241-286 implicit_correction(dt_divs, dt_ts, dt_ln_ps, divs, ts, ln_ps, delta_t, previous, current)
extern "C" int isca_implicit_correction(isca_dyn_t *h, double *dt_divs, double *dt_ts, double *dt_ln_ps, const double *divs_previous,
                                        const double *divs_current, const double *ts_previous, const double *ts_current,
                                        const double *ln_ps_previous, const double *ln_ps_current, double delta_t) {
  API_BEGIN
  const double *const st[4][3] = {{nullptr, nullptr, nullptr}, {divs_previous, divs_current, nullptr},
                                  {ts_previous, ts_current, nullptr}, {ln_ps_previous, ln_ps_current, nullptr}};
  double *const dt[4] = {nullptr, dt_divs, dt_ts, dt_ln_ps};
  spec_stage(h, 0, delta_t, h->cfg.robert_coeff, st, dt, false);
  API_END
}
// spectral_damping.F90:172-291 compute_spectral_damping (which = 0), _vor (1), _div (2): dt <- (dt - c*field)/(1 + c*delta_t)
extern "C" int isca_compute_spectral_damping(isca_dyn_t *h, int which, const double *field_previous, double *dt_field, double delta_t) {
  API_BEGIN
  if (which < 0 || which > 2) fail("compute_spectral_damping: which = 0 (temperature/tracer), 1 (vorticity), 2 (divergence)");
  const int v = (which == 0) ? 2 : which - 1;
  const double *st[4][3] = {};
  double *dt[4] = {};
  st[v][0] = field_previous; dt[v] = dt_field;
  spec_stage(h, 1, delta_t, h->cfg.robert_coeff, st, dt, false);
  API_END
}
// leapfrog.F90:58-105 leapfrog_2level_A + leapfrog_2level_B with future = previous (raw_filter_coeff = 1):
// on return `previous` holds the new time level and `current` the Robert-filtered one
extern "C" int isca_leapfrog(isca_dyn_t *h, double *previous, double *current, const double *dt_field, double delta_t, double robert_coeff) {
  API_BEGIN
  const double *st[4][3] = {};
  double *dt[4] = {};
  st[2][0] = previous; st[2][1] = current; dt[2] = const_cast<double *>(dt_field);
  spec_stage(h, 2, delta_t, robert_coeff, st, dt, true);
  API_END
}

// ---------------------------------------------------------------------------------------------------
// Diagnostics (spectral_diagnostics + diag_manager's time averaging, spectral_dynamics.F90:1554-1867)
// ---------------------------------------------------------------------------------------------------
static int diag_index(const std::string &nm) {
  for (int i = 0; i < NDIAG; ++i) if (nm == DIAG_NAMES[i]) return i;
  return -1;
}
// select the fields to accumulate: comma-separated reference names ("ps,ucomp,vcomp,temp,vor,div"), "" switches off
extern "C" int isca_dyn_diag_select(isca_dyn_t *h, const char *names) {
  API_BEGIN
  if (!h || !names) fail("null argument");
  unsigned mask = 0;
  std::string all(names), tok;
  for (size_t i = 0; i <= all.size(); ++i) {
    if (i == all.size() || all[i] == ',') {
      while (!tok.empty() && tok.back() == ' ') tok.pop_back();
      while (!tok.empty() && tok.front() == ' ') tok.erase(tok.begin());
      if (!tok.empty()) {
        const int k = diag_index(tok);
        if (k < 0) fail("diag_select: unknown field '" + tok + "'");
        if (k == 7 && !h->tracer_on) fail("diag_select: no grid tracer in this configuration");
        if (k >= 20 && h->cfg.physics != 1) fail("diag_select: '" + tok + "' exists only with the moist physics package");
        mask |= 1u << k;
      }
      tok.clear();
    } else tok.push_back(all[i]);
  }
  const Geom &g = h->g;
  const size_t ng2 = (size_t)g.Jl * g.I, ng3 = ng2 * g.L;
  for (int k = 0; k < NDIAG; ++k)
    if ((mask >> k & 1u) && !h->d.diag_acc[k]) h->d.diag_acc[k] = dalloc<double>(h, diag_is_2d(k) ? ng2 : ng3);
  h->diag_mask = mask;
  for (int k = 0; k < NDIAG; ++k)
    if (mask >> k & 1u) HIP_CHECK(hipMemsetAsync(h->d.diag_acc[k], 0, (diag_is_2d(k) ? ng2 : ng3) * sizeof(double), h->stream));
  h->diag_count = 0;
  HIP_CHECK(hipStreamSynchronize(h->stream));
  API_END
}
// time mean of a selected field since the last reset (sum / number of steps) and that number; reset != 0 starts a new interval
extern "C" int isca_dyn_diag_read(isca_dyn_t *h, const char *name, double *host, size_t count, long *nsteps, int reset) {
  API_BEGIN
  if (!h || !name) fail("null argument");
  const int k = diag_index(name);
  if (k < 0 || !(h->diag_mask >> k & 1u)) fail(std::string("diag_read: field not selected: ") + name);
  const Geom &g = h->g;
  const size_t n = (size_t)g.Jl * g.I * (diag_is_2d(k) ? 1 : g.L);
  if (nsteps) *nsteps = h->diag_count;
  if (host) {
    if (count != n) fail(std::string("diag_read: wrong element count for ") + name);
    d2h(h, host, h->d.diag_acc[k], n);
    if (h->diag_count > 0) for (size_t i = 0; i < n; ++i) host[i] = host[i] / (double)h->diag_count;
  }
  if (reset) {
    for (int q = 0; q < NDIAG; ++q)
      if (h->diag_mask >> q & 1u) HIP_CHECK(hipMemsetAsync(h->d.diag_acc[q], 0, (size_t)g.Jl * g.I * (diag_is_2d(q) ? 1 : g.L) * sizeof(double), h->stream));
    h->diag_count = 0;
    HIP_CHECK(hipStreamSynchronize(h->stream));
  }
  API_END
}

// ---------------------------------------------------------------------------------------------------
// benchmarking helpers
// ---------------------------------------------------------------------------------------------------
extern "C" int isca_bench_transform_pair(isca_dyn_t *h, int nfields, int reps, double *pair_ms, double *kernel_ms) {
  API_BEGIN
  require_single(h, "bench_transform_pair");
  const Geom &g = h->g;
  if (nfields < 1 || nfields > h->cap_cols) fail("nfields out of range");
  double *grid = nullptr;
  const size_t ngrid = (size_t)nfields * g.Jl * g.I;
  HIP_CHECK(hipMalloc((void **)&grid, ngrid * sizeof(double)));
  // band-limited random coefficients already in Si (deterministic LCG), synthesised once
  {
    const int C = col_pitch(nfields);
    std::vector<double> s((size_t)g.Ml * g.N1 * C, 0.0);
    unsigned long long st = 20260927ULL;
    for (int ml = 0; ml < g.Ml; ++ml) for (int n = 0; n < g.N1 - 1 - h->h_m_local[ml]; ++n) for (int c = 0; c < 2 * nfields; ++c) {
      st = st * 6364136223846793005ULL + 1442695040888963407ULL;
      const double r = ((double)(st >> 11) / 9007199254740992.0) - 0.5;
      const double tot = h->h_m_local[ml] + n;
      s[((size_t)ml * g.N1 + n) * C + c] = (h->h_m_local[ml] == 0 && (c & 1)) ? 0.0 : r / ((1 + tot) * (1 + tot));
    }
    h2d(h, h->d.Si, s.data(), s.size());
  }
  FieldList fl = single_list(grid, nfields, OP_NONE);
  const int C = col_pitch(nfields);
  hipEvent_t ev[5][2];
  for (auto &e : ev) { hipEventCreate(&e[0]); hipEventCreate(&e[1]); }
  double acc[5] = {0, 0, 0, 0, 0};
  for (int r = -2; r < reps; ++r) {
    hipEventRecord(ev[4][0], h->stream);
    hipEventRecord(ev[0][0], h->stream); launch_legendre_inverse(g, h->d, h->d.Si, h->d.Fi_s, C, 0, h->cfg.legendre_impl, h->stream); hipEventRecord(ev[0][1], h->stream);
    hipEventRecord(ev[1][0], h->stream); launch_fft_inverse(g, h->d, fl, h->d.Fi_g, h->stream); hipEventRecord(ev[1][1], h->stream);
    hipEventRecord(ev[2][0], h->stream); launch_fft_forward(g, h->d, fl, h->d.Ff_g, h->stream); hipEventRecord(ev[2][1], h->stream);
    hipEventRecord(ev[3][0], h->stream); launch_legendre_forward(g, h->d, h->d.Ff_s, h->d.Sf, C, 0, h->cfg.legendre_impl, h->stream); hipEventRecord(ev[3][1], h->stream);
    hipEventRecord(ev[4][1], h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (r >= 0) for (int i = 0; i < 5; ++i) { float ms; hipEventElapsedTime(&ms, ev[i][0], ev[i][1]); acc[i] += ms; }
  }
  for (auto &e : ev) { hipEventDestroy(e[0]); hipEventDestroy(e[1]); }
  *pair_ms = acc[4] / reps;
  for (int i = 0; i < 4; ++i) kernel_ms[i] = acc[i] / reps;
  hipFree(grid);
  API_END
}

extern "C" int isca_dyn_kernel_times(isca_dyn_t *h, int enable, double *ms, int max, char *names, size_t names_len, int *n) {
  API_BEGIN
  timer_collect(h);
  auto &t = h->timer;
  int cnt = 0;
  std::string nm;
  for (size_t i = 0; i < t.names.size() && (int)i < max; ++i) {
    ms[i] = t.calls[i] ? t.ms[i] / t.calls[i] : 0.0;
    nm += t.names[i]; nm += ';';
    ++cnt;
  }
  if (names && names_len) { std::strncpy(names, nm.c_str(), names_len - 1); names[names_len - 1] = 0; }
  if (n) *n = cnt;
  t.names.clear(); t.ms.clear(); t.calls.clear();
  t.enabled = enable != 0;
  t.segments = enable == 2;          // 2: the sharded step's segments between its exchanges ("seg_*") instead of its kernels
  API_END
}
#include "shallow.inc"
