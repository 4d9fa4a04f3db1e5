// Device-side state of one isca_dyn handle (one rank = one GPU = one latitude band + one m-set).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <stdexcept>
#include "tables.h"
#include "comm.h"

namespace isca {
struct MoistState;

#define HIP_CHECK(expr)                                                                              \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + \
                               ":" + std::to_string(__LINE__) + ")");                                \
  } while (0)

constexpr int MAX_FIELDS = 12;
// grid-tracer buffers of a time level under lazy fixers: materialised; just produced (tr_atm holds it uncorrected, tr is stale);
// Robert-filtered except for the `future` term (tr holds that, tr_atm the uncorrected unfiltered copy)
enum TracerState { TR_MAT = 0, TR_NEW = 1, TR_FILT = 2 };

// One batched transform = a list of level-fields.  Grid side: pointers to [nlev][Jl][I] arrays.
// Column index of (field f, level k, re/im) in the Fourier / spectral work buffers: 2*(off[f]+k)+ri.
enum GridOp { OP_NONE = 0, OP_COSM = 1, OP_EXP = 2 };
// Doubles per row of the Fourier / spectral work buffers holding `ncol` complex level-fields: padded to whole 128-byte lines, so that the
// FFT kernels' per-wavenumber pieces (8 or 16 columns) and the Legendre kernels' column tiles start on a line (the few padding columns are
// transformed like the others and never read back)
inline int col_pitch(int ncol) { return (2 * ncol + 15) & ~15; }

struct FieldList {
  int nf;
  int ncol;                 // real level-fields = sum(nlev)
  double *g[MAX_FIELDS];    // grid arrays
  int nlev[MAX_FIELDS];
  int off[MAX_FIELDS];
  int op[MAX_FIELDS];       // applied to grid values: inverse: after FFT; forward: before FFT
  // x-derivatives taken in Fourier space (inverse direction, k_fft_inv3): with nbuf != 0 the Fourier buffer holds nbuf columns, field f's rows come from
  // buffer columns boff[f] + k, and a field with dx[f] != 0 is i m dxfac times the rows of the field it names -- d/d(lambda) / a of a field whose
  // Fourier coefficients are in the buffer anyway (compute_gradient_cos' x part, spherical.F90:270-301, applied after the Legendre sum instead of before)
  int nbuf = 0;
  int boff[MAX_FIELDS] = {};
  int dx[MAX_FIELDS] = {};
  double dxfac = 0.0;
  // the rows of fields il_a and il_b alternate (a's level k, b's level k, a's level k + 1, ...) from row off[il_a] on: a field and its x-derivative side by
  // side, so that the two rows' reads of the same Fourier coefficients fall into one work item (and mostly one load instruction) of k_fft_inv3
  int il_a = -1, il_b = -1;
};

// geometry shared by all kernels
struct Geom {
  int I, J, Jl, j0;         // lon points, global lats, local lats, first local lat (global index)
  int M1, N1, L;            // num_fourier+1, num_spherical+1, levels
  int P, rank, Ml;          // ranks, this rank, local m slots (padded)
  int Jh;                   // J/2
  int NHP;                  // padded count of n per parity (multiple of 16)
  int log2I;
  int log2Jl;               // log2(Jl) when Jl is a power of two, else -1
};

struct Dev {
  // ---- tables
  double *cosm_lat_l, *wts_lat_l, *coriolis_l, *sin_lat_l, *rad_lat_l;   // [Jl] local latitudes
  double *wts_lat_g;                                                      // [J]
  int *m_of_slot;            // [P*Ml]   global m of (rank q, local slot), -1 if padding
  int *slot_of_m;            // [M1]     q*Ml + ml
  int *m_local;              // [Ml]     global m of my slots
  // Legendre tables for my m slots, parity-split, n-half index padded to NHP (plain-FMA check kernels)
  double *pw_fwd;            // [Ml][2][Jh][NHP]   P(m,n,j')*w(j'), n = 2*nh+par   (analysis A operand)
  double *p_inv;             // [Ml][2][NHP][Jh]   P(m,n,j')                       (synthesis A operand)
  // the same tables in MFMA fragment order (legendre.hip) and the coefficient table of the fused synthesis
  double *leg_fwd_frag;      // [Ml][Jh/4][2][NHP/16][64]
  double *leg_inv_frag;      // [Ml][2][NHP/4][Jh/16][64]
  double *leg_scoef;         // [Ml][N1+16][5][4]
  // fused FFT + Legendre analysis (kernels.hip k_fft_leg_fwd; one rank, lon_max = 256): the table in 4x4x4 fragment order per MFMA wavefront
  double *fz_frag = nullptr; // [4][Jh/8][NT][64][2]
  int *fz_desc = nullptr;    // [4][NT][2] {ml, 16 * tile | nlim << 16}, ml = -1: padding
  int fz_NT = 0;
  double *coef;              // [9][Ml][N1]: eigen,uvm,uvc,uvp,alpm,alpp,dym,dx,dyp ; [9]=mask ; [10]=damping
  double *pk, *bk, *dpk, *dbk;
  double *wave_mat_t;        // [num_spherical][L(k')][L(k)]  transposed wave matrices
  double *tau_t, *gamma_t;   // unused by the scan formulation; kept for checks
  int *mn_active;            // [n_active][4] {n, ml, m, total wavenumber} of the retained (m,n) of my wavenumbers, grouped by total wavenumber (padding: n = -1)
  double *impl_vec;          // [6][L+1]: ref_ln_p_half, ref_ln_p_full, h, dp_ref, ...
  double *tw;                // [I/2][2] twiddles exp(-2 pi i k/I)
  // ---- prognostic state
  double *ug[2], *vg[2], *tg[2], *psg[2], *tr[2];   // grid, two time levels
  double *vorg, *divg, *dxT, *dyT, *dxlp, *dylp;    // grid, at `current`
  double *diag_acc[32] = {};                        // diagnostics: running sums of the selected fields
  double *surf_geop = nullptr;   // [Jl][I] surface geopotential (get_topography, spectral_init_cond.F90:167-308); zero = flat
  double *wg_full;
  double *wg;                // [L+1][Jl][I] vertical mass flux at interfaces (four_in_one), for the tracer
  double *tr_atm[2];         // atmosphere_mod's own (never Robert-filtered) copy of the grid tracer
  double *trh;               // tracer after the horizontal van Leer step
  double *tv = nullptr;      // virtual temperature work array (use_virtual_temperature)
  // tracers 2..num_tracers ([e] = tracer e+2): grid values, atmosphere_mod's copy, spectral coefficients (spectral tracers only),
  // and the column sums the transport kernel writes for tracer 1's water fixer (unused here)
  double *trx[2][ISCA_MAX_TRACERS - 1] = {}, *trx_atm[2][ISCA_MAX_TRACERS - 1] = {}, *trxs[2][ISCA_MAX_TRACERS - 1] = {}, *wcol_x = nullptr, *ph_dtqx[ISCA_MAX_TRACERS - 1] = {};
  double *halo_send, *halo_recv;   // [2 sides][3 + more grid tracers][L][2][I] tracer halo rows (lo, hi): q0 of tracer 1, u, v, q0 of the further grid tracers
  double *psp_copy;          // [Jl][I] psg(previous) saved by the column kernel for the concurrent tracer stream
  int *kmask;                // [Jl][I] number of levels with p_full < water_correction_limit: byte 0 this step, byte 1 the step before, byte 2 ...
  int *kmask_old;            // the word of the step before (the column kernel reads it and writes kmask; phase0 swaps the two): what the horizontal
                             // tracer kernel reads, so that it does not depend on this step's column kernel
  double *pend;              // [3][4] fixer scalars PENDING on time level 0 / 1 (mass factor, temperature correction, water factor, -); row 2: identity
  double *wcol;              // [5][Jl][I] column sums for the water fixer
  double *w0blk = nullptr, *w0blk_x = nullptr;   // [L][row blocks] the horizontal tracer kernel's weighted sums of q0 (TracerArgs.filt_horiz); _x: further tracers (unused sums)
  double *fv_c, *fv_cc, *fv_dy, *fv_dyy, *fv_dyp, *fv_dym;   // fv_advection tables (global latitudes)
  double *fv_rcdx, *fv_rdyy, *fv_rcdy, *fv_rdy;              // reciprocals used by the kernels
  double *ppm_tab;           // [6][L] pure-sigma PPM slope / edge weights
  double *col_sig;           // [L][16] per-level constants of the column kernel on pure sigma levels (k_column_sig; null: hybrid levels, 'mcm')
  double *hs_sin_l;          // [Jl] sin(lat) as hs_forcing forms it (hs_forcing.F90:519: sin of the latitude in radians), host libm
  double *vors[2], *divs[2], *ts[2], *lnps[2];      // spectral [Ml][N1][L] complex ; lnps [Ml][N1]
  // ---- work
  double *g_dtu, *g_dtv, *g_dtT, *g_E, *g_dtlp;     // forward-batch grid inputs
  double *Ff_g, *Ff_s, *Fi_s, *Fi_g;                // Fourier buffers (grid side / spectral side)
  double *Sf, *Si;                                  // spectral work [Ml][N1][Cf], [Ml][N1][Ci]
  double *s_dtvor, *s_dtdiv, *s_dtT, *s_dtlp;       // spectral tendencies [Ml][N1][L]
  // raw_filter_coeff /= 1 only: prev - 2 cur of the spectral fields and of the grid tracer (leapfrog_2level_A's part_filt_*)
  double *part_vor = nullptr, *part_div = nullptr, *part_t = nullptr, *part_lp = nullptr, *tr_part = nullptr;
  double *partials;                                 // block partial sums
  double *red;                                      // [32] global sums [0..9] / fixer scalars [16..18] / [20..21] extremes of T / [25] a column block gave up waiting for the deferred finish
  void *fin_args;                                   // the deferred finish's device state (kernels.hip DeferredFin: two argument sets, the sequence word, the published scalars)
  double *scratch_g[4], *scratch_s[4];              // API transforms
  double *lh_lon = nullptr, *lh_lat_l = nullptr;     // hs_forcing's local_heating_option = 'Isidoro': srfamp x the longitude factor [I], the latitude factor [Jl]
  // ---- moist physics package (physics = 1)
  double *ph_dtu = nullptr, *ph_dtv = nullptr, *ph_dtT = nullptr, *ph_dtq = nullptr;   // tendencies returned by idealized_moist_phys
  double *t_surf = nullptr, *precip = nullptr;      // [Jl][I] mixed-layer temperature; rain rate of the last step
  double *moist_work = nullptr;                     // p_full/p_half/z_full/z_half of both time levels
  // lazy fixers with the moist package: T, q (atmosphere_mod's copy) and p_s of a time level with what is pending on it applied, written once by
  // k_moist_pressures when the level is the current one, in the pressure slot of that level (the physics kernels read these, not the stored fields)
  double *m_t[2] = {}, *m_q[2] = {}, *m_ps[2] = {};
  // k_moist_convcond -> k_moist_column, two sets (a step's convection is computed while the step before runs): (conv + cond) rates [L][Jl][I], rain [Jl][I]
  double *cc_dT[2] = {nullptr, nullptr}, *cc_dq[2] = {nullptr, nullptr}, *cc_precip[2] = {nullptr, nullptr};
};

// doubles per side of the tracer halo buffers: (q0, u, v) + one q0 per further grid tracer, two rows of every level each
inline size_t halo_doubles(const Geom &g, int num_tracers) { return (size_t)(3 + (num_tracers > 1 ? num_tracers - 1 : 0)) * g.L * 2 * g.I; }

struct KernelTimer {
  bool enabled = false;
  bool segments = false;        // one event pair per run of kernels between two exchanges (sharded step) instead of one per kernel
  std::vector<std::string> names;
  std::vector<double> ms;
  std::vector<long> calls;
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<int> ev_name;
};

}  // namespace isca

struct isca_history;                // history_nc.cpp: the open diag_table of a handle (isca_dyn_diag_open)

struct isca_dyn {
  isca_dyn_config cfg;
  isca::Tables tab;
  isca::Geom g;
  isca::Dev d;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t stream2 = nullptr;    // grid-tracer transport runs here, concurrently with the spectral pipeline
  hipEvent_t ev_fork = nullptr, ev_fork0 = nullptr, ev_join = nullptr;
  int previous = 0, current = 0;
  long step_count = 0;
  bool have_state = false;
  int Cf, Ci;                       // doubles per (m,j) row in forward / inverse Fourier buffers
  isca::FieldList fl_fwd, fl_inv;
  std::vector<void *> allocs;
  isca::KernelTimer timer;
  size_t nblocks_col = 0;
  int kernels_per_step = 0;
  double wave_dt = -1.0;
  int ml_of_m0 = -1;
  std::vector<int> h_m_local, h_slot_of_m, h_m_of_slot;
  std::vector<double> h_surf_geop;  // global (lat_max, lon_max) surface geopotential as handed over (empty = flat)
  int n_active = 0;
  bool fuse_synth = false;
  bool dx_fourier = false;          // the step's synthesis batch carries no x-derivative fields: k_fft_inv3 forms dT/dx and d ln ps/dx from the Fourier rows of T and ln ps
  bool fuse_fwd = false;            // FFT + Legendre analysis of the step's forward batch in one kernel (ISCA_FUSE_FFT_LEG)
  bool tr_filt_horiz = false;       // this step: the tracer filter's first half and the "water before" sum are the horizontal kernel's (TracerArgs.filt_horiz)
  bool tracer_filter_in_vert = false;   // ISCA_TRACER_FILTER_IN_VERT=1
  bool tracer_serial = false;       // debugging/profiling: run the tracer kernels on the main stream
  bool tracer_early = false;        // the horizontal tracer kernel forks BEFORE the column kernel (ISCA_TRACER_EARLY=1; see spectral_dynamics_init)
  bool tracer_on = false;           // advect the grid tracer
  bool tracer_env_off = false;      // ISCA_NO_TRACER was set when the handle was created
  int cap_cols = 0;                 // capacity (level-fields) of the Fourier/spectral work buffers
  unsigned diag_mask = 0;           // spectral_diagnostics fields being accumulated (bit = index in DIAG_NAMES)
  long diag_count = 0;              // send_data calls since the last reset
  isca::MoistState *moist = nullptr;   // tables of the moist physics package (physics = 1)
  long phys_calls = 0;              // calls of the physics since create / restart (gust is 1 m/s on the first)
  // moist package: the pressures of the step's CURRENT level are the next step's PREVIOUS-level pressures (the grid p_s of a level does not
  // change once its fixers are applied): slot moist_pslot of the work area holds them while moist_pcache is true (reset by every state write)
  int moist_pslot = 0;
  bool moist_pcache = false;
  // ... and that level's T, q are the next step's PREVIOUS-level fields, all the convection and the condensation read: k_moist_physics computes the
  // next step's beside this step's chain (cc_valid: buffer set cc_slot holds the rates of the step about to be taken; dropped by every state write,
  // after which k_moist_convcond runs in front of the step).  ISCA_MOIST_NO_PIPELINE: every step its own.
  bool cc_pipeline = false, cc_valid = false;
  int cc_slot = 0;
  isca::Comm *comm = nullptr;       // RCCL communicator of the sharded step (isca_dyn_comm_init), else the host drives the phases
  // Lazy fixers: compute_corrections' scalars and the grid tracer's leapfrog_2level_B stay pending on the new level and are applied by
  // the next steps' kernels as they read it (no pass over the fields at the end of the step); materialised for every host access.
  bool lazy_fix = false;
  int tr_state[2] = {0, 0};         // TracerState of the tracer buffers of time level 0 / 1
  bool thermo_pending[2] = {false, false};   // mass factor / temperature correction pending on psg / tg of time level 0 / 1
  // The finish of the last step's fixers (the three scalars, the (0,0) patch) is DEFERRED to block 0 of the next column kernel (kernels.hip ColumnArgs::fin)
  // on the plain one-rank path; anything else that needs the scalars first -- the host reading state, diagnostics, a physics package in front of the
  // column kernel -- runs k_fixer_finish instead (api.hip: flush_finish).  fin_prev / _cur / _fut: that step's time levels; fin_seq: the number of deferred launches so far (its parity picks the slot block 0 publishes in).
  bool fin_deferred = false;
  int fin_prev = 0, fin_cur = 0, fin_fut = 0;
  unsigned fin_seq = 0;
  bool in_step = false;             // between phase 0 and phase 3 of a step driven phase by phase
  double *host_red = nullptr;       // pinned: the fixer scalars / temperature extremes read back at a synchronisation point
  isca_history *hist = nullptr;     // history files being written (isca_dyn_diag_open): the step loops call isca_history_after_step
  bool hist_wg_full = false;        // ... one of them samples omega instantaneously (time_avg = .false.): every step stores wg_full, whichever completes a record
};
void isca_history_after_step(isca_dyn *h);      // history_nc.cpp
void isca_history_destroy(isca_dyn *h);
