// Launchers for the gfx950 kernels of the spectral core (definitions in kernels.hip).
#pragma once
#include "core.h"

#include <cstdlib>
// Switches of measured-and-rejected variants (HISTORY.md has their numbers) exist only in a build with -DISCA_EXPERIMENTS (tools/build_variant.sh);
// the product looks none of them up and instantiates none of their kernels.  What the product does read from the environment is listed in
// DESIGN.md 7 "Environment": the communicator's configuration and eight test hooks that force a path a configuration would select by itself.
#ifdef ISCA_EXPERIMENTS
inline const char *exp_env(const char *name) { return getenv(name); }
#else
inline const char *exp_env(const char *) { return nullptr; }
#endif

namespace isca {

struct StepScalars {      // per-step scalars passed by value to kernels
  double delta_t;         // dt or 2 dt
  double xi;              // alpha_implicit * delta_t
  int prev, cur, fut;
  int store_wg_full = 1;   // 0: a step inside a run of steps of one isca_dyn_step call, whose omega nobody can read (and no diagnostic wants)
  int keep_spec_tend = 0;  // k_spec_update also stores dt_vors, dt_divs, dt_ts, dt_ln_ps (locals of spectral_dynamics; kept for the phase-by-phase API's get_state)
};

// ---- shallow-water sibling core (kernels.hip, "Shallow-water sibling core")
struct SwGridArgs {
  const double *u, *v, *vor, *div, *h, *up, *vp, *hp, *dxh, *dyh, *coriolis, *h_eq, *deep;
  double kappa_m, kappa_t;
  double *tend_u, *tend_v, *tend_h, *bg, *pv;
  int n, I;
};
struct SwSpecArgs {
  const double *coef;
  const double2 *vor_p, *div_p, *h_p;
  double2 *vor_c, *div_c, *h_c, *vor_f, *div_f, *h_f;
  const double2 *dt_vor, *dt_div, *dt_h, *bs;
  const double2 *stir;                        // stirring added to the vorticity tendency after the damping (null: none)
  double delta_t, robert, h_0, damping_r;     // damping_r: spectral_damping_init's damping_coeff_r (linear drag), 0 for shallow water
  int first, mode;
};
void launch_sw_grid_tend(const SwGridArgs &a, hipStream_t s);
void launch_bt_grid_tend(int n, int I, const double *u, const double *v, const double *vor, const double *coriolis, double *tend_u, double *tend_v,
                         double *pv, hipStream_t s);
void launch_sw_spec_update(const Geom &g, const SwSpecArgs &a, hipStream_t s);
void launch_sw_grid_tracer_filter(int n, double robert, const double *prev, double *cur, const double *adv, double *fut, hipStream_t s);
void launch_sw_stir_update(const Geom &g, double bstir, int mn00, const double *fresh, double *s_stir, hipStream_t s);
void launch_sw_scale_grid(int n, const double *factor, double *field, hipStream_t s);
void launch_sw_tracer_tend(int n, const double *u, const double *v, const double *dx, const double *dy, double *tend, hipStream_t s);

// ---- transforms
void launch_fft_forward(const Geom &g, const Dev &d, const FieldList &fl, double *Fg, hipStream_t s);
void launch_fft_inverse(const Geom &g, const Dev &d, const FieldList &fl, const double *Fg, hipStream_t s);
// fused forward transform of the step (grid -> spectral work rows, no Fourier buffer): tables built at create when the geometry fits
int build_fused_fwd_tables(const Geom &g, const Tables &T, const std::vector<int> &m_local, std::vector<double> &frag, std::vector<int> &desc);
void launch_fft_legendre_forward(const Geom &g, const Dev &d, const FieldList &fl, double *S, hipStream_t s);
bool fused_forward_ok(const Geom &g);
// legendre.hip: true when the MFMA kernels cover this geometry; fragment-ordered tables built at create
bool legendre_mfma_ok(const Geom &g, int impl);
void build_legendre_fragments(const Geom &g, const Tables &T, const std::vector<int> &m_local, std::vector<double> &fwd,
                              std::vector<double> &inv, std::vector<double> &scoef);
void launch_legendre_forward(const Geom &g, const Dev &d, const double *Fs, double *S, int C, int full, int impl, hipStream_t s);
// fused_tl >= 0: build the inverse-batch columns on the fly from the spectral state at that time level (S unused)
void launch_legendre_inverse(const Geom &g, const Dev &d, const double *S, double *Fs, int C, int full, int impl, hipStream_t s, int fused_tl = -1, int dxf = 0);

// ---- spectral-space kernels
// pack a spectral state array [Ml][N1][nlev] (complex) into columns of a work buffer, and back
void launch_spec_pack(const Geom &g, const double *state, double *S, int C, int coloff, int nlev, hipStream_t s);
void launch_spec_level_chunk(const Geom &g, const double *src, double *dst, int nlev, int k0, int nk, int copies, hipStream_t s);
void launch_spec_unpack(const Geom &g, const Dev &d, const double *S, double *state, int C, int coloff, int nlev, int mask, hipStream_t s);
// (vor,div) state -> (ucos,vcos) columns ; (ucos,vcos) columns -> masked (vor,div) state ; gradient_cos
void launch_spec_ucos_vcos(const Geom &g, const Dev &d, const double *vor, const double *div, double *S, int C, int col_u, int col_v, int nlev, hipStream_t s);
void launch_spec_vor_div(const Geom &g, const Dev &d, const double *S, int C, int col_u, int col_v, double *vor, double *div, int nlev, hipStream_t s, int mask = 1);
void launch_spec_laplacian(const Geom &g, const Dev &d, const double *in, double *out, int nlev, int power, hipStream_t s);
void launch_spec_gradient(const Geom &g, const Dev &d, const double *state, double *S, int C, int col_dx, int col_dy, int nlev, hipStream_t s);

// the time step in spectral space
void launch_spec_tendencies(const isca_dyn &h, hipStream_t s);                       // S1
void launch_spec_update(const isca_dyn &h, const StepScalars &sc, hipStream_t s);    // S2
void launch_spec_synthesis_inputs(const isca_dyn &h, int tl, hipStream_t s);         // S3
void launch_raw_adjust(const isca_dyn &h, int fut, hipStream_t s);                   // future half of the RAW filter (raw_filter_coeff /= 1)
void launch_spec_update_stage(const isca_dyn &h, int stage, double delta_t, double robert, double *const st[4][3],
                              double *const dtend[4], hipStream_t s);                 // parts of S2 on caller data

// ---- grid-space kernels
void launch_column(const isca_dyn &h, const StepScalars &sc, hipStream_t s);
// vert_advect_uv / vert_advect_t other than second_centered: the scheme on whole columns, added to the column kernel's tendencies
void launch_vert_advection_schemes(const isca_dyn &h, const StepScalars &sc, hipStream_t s);
bool virtual_t_on(const isca_dyn &h);
// rows of Dev::pend (lazy fixers, kernels.hip): what is pending on time level tl sits at pend[4 * tl + ...]; row 2 is the identity
constexpr int PEND_FACTOR = 0, PEND_TCORR = 1, PEND_WFAC = 2, PEND_IDENTITY = 8;
void launch_virtual_t(const isca_dyn &h, const double *t, const double *q, double *tv, hipStream_t s);
void launch_tracer(const isca_dyn &h, const StepScalars &sc, hipStream_t s, int part = -1);
void launch_tracer_finish(const isca_dyn &h, const StepScalars &sc, int e, hipStream_t s);
void launch_vert_advection_centered(const isca_dyn &h, const double *w, const double *ps, const double *r, double *rdt, hipStream_t s);
void launch_vert_advection_field(const isca_dyn &h, int scheme, const double *ps, const double *r, double *rdt, double delta_t, hipStream_t s);
int tracer_vert_scheme(const isca_dyn &h, int k);     // advect_vert of field_table entry k (0-based): 0 second_centered .. 3 finite_volume_parabolic
// water_borrowing (hole_filling = 'on' of a 'spectral' tracer): dt_q corrected from the previous level's values q_prev, dp of surface pressure ps
void launch_water_borrowing(const isca_dyn &h, const double *ps, const double *q_prev, double *dt_q, double delta_t, hipStream_t s);
void launch_leapfrog_a(size_t n, const double *prev, double *cur, double *fut, const double *dta, double delta_t, double robert, double raw,
                       double *part, hipStream_t s);
void launch_leapfrog_b(size_t n, double *cur, double *fut, const double *part, double robert, double raw, hipStream_t s);
void launch_spec_tracer_update(const isca_dyn &h, const StepScalars &sc, int e, const double *dt_trs, hipStream_t s);
void launch_tracer_pack_halo(const isca_dyn &h, const StepScalars &sc, hipStream_t s);   // rows for the neighbour bands   // grid tracer: van Leer + PPM + filter part A
bool hs_forcing_separate(const isca_dyn &h);       // an hs_forcing_nml option the fused column kernel does not carry: k_hs_forcing_step in front of it
void launch_hs_forcing_step(const isca_dyn &h, const StepScalars &sc, hipStream_t s);
size_t deferred_fixer_args_bytes();
void upload_deferred_fixer_args(const isca_dyn &h);      // at the end of create: Dev::fin_args
bool column_takes_deferred_finish(const isca_dyn &h);    // the step's column kernel is the plain pure-sigma one, whose block 0 can finish the step before's fixers
void launch_fixer_sums(const isca_dyn &h, int fut, hipStream_t s);          // R1: partial sums over the local band
void launch_fixer_apply(const isca_dyn &h, const StepScalars &sc, hipStream_t s);   // R2: reduce + scalars + apply
void launch_fixer_finish(const isca_dyn &h, const StepScalars &sc, hipStream_t s);  // R2 with lazy fixers: reduce + scalars, left pending on the new level
void launch_fixer_materialize(const isca_dyn &h, hipStream_t s);                    // apply what is pending on both time levels in place
void launch_hs_forcing(const isca_dyn &h, double dt, const double *p_half, const double *p_full, const double *u,
                       const double *v, const double *t, double *udt, double *vdt, double *tdt, hipStream_t s);
void launch_pressures_heights(const isca_dyn &h, const double *t, const double *ps, double *p_full, double *p_half,
                              double *z_full, double *z_half, hipStream_t s);
void launch_hadv_combine(const Geom &g, const double *u, const double *v, const double *dx, const double *dy, double *tend, int nlev, hipStream_t s);
void launch_scale_rows(const Geom &g, const Dev &d, double *a, int nlev, hipStream_t s);   // a *= cosm_lat (divide_by_cos)

// component entry points on caller fields
void launch_pressure_variables(const isca_dyn &h, const double *ps, double *p_half, double *ln_p_half, double *p_full, double *ln_p_full, hipStream_t s);
void launch_geopotential(const isca_dyn &h, const double *t, const double *ln_p_half, const double *ln_p_full, double *gf, double *gh, hipStream_t s, const double *surf_geop = nullptr);   // surf_geop: the caller's lower boundary, or the handle's own
void launch_mass_weighted_rows(const isca_dyn &h, const double *f, const double *ps, double *rows, hipStream_t s);
void launch_fv_horiz_on(const isca_dyn &h, const double *u, const double *v, const double *q, const double *ps, double dt, double *q_new, hipStream_t s);
void launch_ppm_vert_on(const isca_dyn &h, double dt, const double *w, const double *ps, const double *r, double *r_new,
                        double *dummy_a, double *dummy_b, hipStream_t s);
void launch_tracer_source_sink(const isca_dyn &h, const double *ps, const double *tr, double *rdt, hipStream_t s, int k = -1);   // k: the field_table entry whose tracer_sms applies (-1: hs_forcing_nml's trflux / trsink)

// spectral_diagnostics (spectral_dynamics.F90:1705-1867): add this step's fields to the running sums
constexpr int NDIAG = 22;
inline bool diag_is_2d(int k) { return k == 0 || k >= 20; }     // ps; moist package: precipitation, t_surf
extern const char *const DIAG_NAMES[NDIAG];     // reference field names, index = bit in the mask; diag_is_2d(k): (lat, lon) fields
void launch_diag_accumulate(const isca_dyn &h, int fut, hipStream_t s);

size_t column_partials_count(const isca_dyn &h);

}  // namespace isca
