/* isca_dyn.h -- C-ABI of the MI355X-native spectral dynamical core.
 *
 * Drop-in boundary for the hot path of ExeClim/Isca's src/atmos_spectral (paths below are relative
 * to the reference's src/):
 *   - atmosphere_mod          atmos_spectral/driver/solo/atmosphere.F90:78   (atmosphere_init/atmosphere/atmosphere_end)
 *   - spectral_dynamics_mod   atmos_spectral/model/spectral_dynamics.F90:95-98
 *   - transforms_mod          atmos_spectral/tools/transforms.F90:134-184
 *   - hs_forcing_mod          atmos_param/hs_forcing/hs_forcing.F90:65
 * The reference has no FFI: the boundary is its Fortran module interface.  A maintainer binds these
 * entry points with `bind(C)` interface blocks; our own Python host binds them
 * with ctypes (isca_amd/dyncore.py).
 *
 * Conventions
 *   - every real is fp64 (the reference builds with -r8); complex = interleaved (re,im) fp64 pairs.
 *   - host arrays use the reference's Fortran layouts: grid (lon, lat, lev), spectral (m, n, lev)
 *     with m fastest; n = meridional index (total wavenumber m+n), n = 0..num_spherical.
 *   - latitudes south -> north (spherical_fourier.F90:413-423).
 *   - all functions return 0 on success, non-zero on error (the reference aborts with FATAL;
 *     here the message is available from isca_last_error()).  Nothing falls back to the CPU: if no
 *     HIP device is usable isca_dyn_create fails.
 *   - a handle owns its device memory and one HIP stream; calls on a handle are serialised.
 */
#ifndef ISCA_DYN_H
#define ISCA_DYN_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct isca_dyn isca_dyn_t;

#define ISCA_MAX_LEVELS 128
#define ISCA_MAX_TRACERS 8

/* Parameters of the moist physics package, physics = 1: idealized_moist_phys with the options of the Frierson grey-radiation
 * aquaplanet (exp/test_cases/frierson/frierson_test_case.py:49-170): SIMPLE_BETTS_MILLER convection, lscale_cond,
 * two_stream_gray_rad 'frierson' without seasonal cycle, surface_flux over a mixed-layer ocean, Rayleigh sponge,
 * diffusivity boundary layer, implicit vertical diffusion; do_simple everywhere, no virtual temperature, no land.
 * Field names are the reference's namelist variables; defaults (isca_dyn_config_default) are the module defaults
 * overridden by the test case's values. */
typedef struct isca_moist_config {
  double roughness_mom, roughness_heat, roughness_moist;          /* idealized_moist_phys_nml */
  double solar_constant, del_sol, del_sw, ir_tau_eq, ir_tau_pole, atm_abs, odp, sw_diff, linear_tau, wv_exponent,
         solar_exponent;                                          /* two_stream_gray_rad_nml */
  double depth, tconst, delta_T, albedo_value;                    /* mixed_layer_nml (prescribe_initial_dist) */
  int evaporation;
  double tau_bm, rhbm, Tmin, Tmax, val_inc;                       /* qe_moist_convection_nml */
  int do_rayleigh;                                                /* damping_driver_nml */
  double trayfric, sponge_pbottom;
  int damping_conserve_energy;
  double constant_gust;                                           /* vert_turb_driver_nml */
  double frac_inner, rich_crit_pbl;                               /* diffusivity_nml */
  double rich_crit, drag_min;                                     /* monin_obukhov_nml */
} isca_moist_config;

/* namelist keys used on the path: spectral_dynamics_nml (spectral_dynamics.F90:152-224),
 * hs_forcing_nml (hs_forcing.F90:76-122), main_nml dt_atmos (atmos_model.F90:111). */
typedef struct isca_dyn_config {
  int lon_max, lat_max, num_fourier, num_spherical, num_levels;
  int fourier_inc;              /* spherical.F90:40,182: index m is zonal wavenumber m * fourier_inc (1 = the whole circle) */
  int triang_trunc;             /* 1: triangular truncation; 0: rhomboidal (spherical.F90:603-644) */
  double dt_atmos;              /* seconds */
  /* spectral_dynamics_nml */
  int damping_order;            /* see damping_option at the end of the struct */
  double damping_coeff;
  double eddy_sponge_coeff, zmu_sponge_coeff, zmv_sponge_coeff;
  double robert_coeff;
  double raw_filter_coeff;      /* in (0, 1]; 1.0 = the pure Robert filter (reference default); < 1: Robert-Asselin-Williams (leapfrog.F90:58-105) */
  double alpha_implicit;
  double reference_sea_level_press;
  double scale_heights, exponent, surf_res;   /* vert_coord_option = 'uneven_sigma' */
  int do_mass_correction, do_energy_correction, do_water_correction;
  double water_correction_limit;
  double initial_temperature;   /* spectral_init_cond_nml, default 264 */
  double initial_sphum;
  double valid_range_t[2];
  int num_tracers;              /* prognostic tracers of the field_table, 0..ISCA_MAX_TRACERS (dry field_table: 1 = sphum); see tracer_spectral below */
  /* hs_forcing_nml */
  double t_zero, t_strat, delh, delv, eps, sigma_b, ka, ks, kf;
  int do_conserve_energy;
  double trflux, trsink, P00;
  /* decomposition: latitude bands in grid space, zonal-wavenumber sets in spectral space
   * (spec_mpp.F90:61-80).  world_size must divide lat_max. */
  int rank, world_size;
  int device;                   /* HIP device ordinal */
  void *stream;                 /* hipStream_t to run on, or NULL for a private stream */
  int legendre_impl;            /* 0 = MFMA (default), 1 = plain-FMA check kernels */
  /* physics package called by atmosphere (atmosphere.F90:304-331): 0 = hs_forcing, 1 = idealized_moist_phys (Frierson);
   * with 1 the grid tracer is specific humidity: it feeds the physics and receives its tendency.
   * 2 = the caller's own physics: the library is spectral_dynamics_mod only and takes the tendencies (isca_dyn_dynamics). */
  int physics;
  /* vert_coord_option = 'input' (vert_coordinate_nml): != 0 -> pk_input/bk_input hold num_levels+1 values, top first */
  int vert_coord_input;
  double pk_input[ISCA_MAX_LEVELS + 1], bk_input[ISCA_MAX_LEVELS + 1];
  isca_moist_config moist;
  double radius, omega;         /* constants_nml: planetary radius (m) and rotation rate (1/s); defaults 6376.0e3, 7.2921150e-5 */
  /* spectral_damping_init (spectral_damping.F90:56-168): damping_option 0 = 'resolution_dependent', 1 = 'exponential_cutoff' (with
   * cutoff_wn; the effective coefficient then depends on the step's delta_t, :186-190), 2 = 'resolution_independent'; separate
   * coefficient / order for vorticity and divergence, negative = those of damping_coeff / damping_order (spectral_dynamics.F90:447-452) */
  int damping_option, cutoff_wn;
  double damping_coeff_vor, damping_coeff_div;
  int damping_order_vor, damping_order_div;
  /* field_table entries of tracers 2..num_tracers (spectral_dynamics_init, spectral_dynamics.F90:316-352; entry [k] describes tracer k+1,
   * entry [0] is ignored: tracer 1 is the grid tracer the water fixer and the physics know as sphum).  tracer_spectral: the entry's
   * numerical_representation, 0 = 'grid' (van Leer horizontally, advect_vert = finite_volume_parabolic like tracer 1), 1 = 'spectral'
   * (horizontal_advection of the spectral coefficients, advect_vert = second_centered, hole_filling = off: the defaults of :145-147;
   * damped like temperature, :1146).  tracer_robert_coeff: the entry's robert_coeff, negative = robert_coeff (:340-351).
   * More than one tracer: raw_filter_coeff = 1; a 'spectral' tracer on more than one rank needs the library's communicator (isca_dyn_comm_init: its
   * transforms' exchanges are issued like the step's own); further 'grid' tracers also run under the host-driven phase API.  State names "tr2".."tr8", "tr_atm2".., and "trs2".. (spectral). */
  int tracer_spectral[ISCA_MAX_TRACERS];
  double tracer_robert_coeff[ISCA_MAX_TRACERS];
  /* use_virtual_temperature (spectral_dynamics_nml, default .false.): T (1 + (rvgas/rdgas - 1) q), q = tracer 1, in the pressure-gradient
   * and energy-conversion terms of four_in_one (spectral_dynamics.F90:857-864), in compute_geopotential and in the heights of
   * compute_pressures_and_heights (press_and_geopot.F90:246-256, 340-355).  Ignored without tracer 1 (the reference's dry_model). */
  int use_virtual_temperature;
  /* vert_advect_uv / vert_advect_t (spectral_dynamics_nml, spectral_dynamics.F90:280-301): the scheme of vert_advection
   * (atmos_shared/vert_advection/vert_advection.F90:69-478, ADVECTIVE_FORM) for u, v and for T --
   * 0 'second_centered' (the default, fused into the column kernel), 1 'fourth_centered' (on the current level, :878,885),
   * 2 'van_leer_linear', 3 'finite_volume_parabolic' (both on the PREVIOUS level with the step's delta_t, :879,886).
   * A value other than 0 adds one kernel to the step (the scheme on whole columns). */
  int vert_advect_uv, vert_advect_t;
  /* use_implicit (spectral_dynamics_nml, default .true. = 1; spectral_dynamics.F90:469-481, 906): 0 = no implicit_correction of the
   * divergence / temperature / surface-pressure tendencies: explicit leapfrog of the gravity waves (needs a short dt_atmos). */
  int use_implicit;
  /* make_symmetric (spectral_dynamics_nml, default .false.; spherical.F90:185): the truncation mask also drops every zonal wavenumber
   * m > 0 -- a zonally symmetric model (exp/test_cases/axisymmetric). */
  int make_symmetric;
  /* vert_difference_option (spectral_dynamics_nml; spectral_dynamics.F90:1084, press_and_geopot.F90:196, 242, implicit.F90:404, 447):
   * 0 'simmons_and_burridge' (the default), 1 'mcm' -- full-level pressures at the arithmetic mean of the half levels, the pressure-gradient
   * term as R T grad(p_s)/p_s, the energy-conversion term with (sum above + half the layer's own mass divergence)/p_full, and the
   * matching linear operator of the semi-implicit scheme. */
  int vert_difference_option;
  /* hole_filling of the field_table entries of tracers 2..num_tracers ([k] = tracer k+1 as in tracer_spectral; 1 = 'on'): water_borrowing
   * (atmos_spectral/model/water_borrowing.F90:38-136) on a 'spectral' tracer's tendency -- a negative value of the previous level is filled from
   * its four neighbours on the latitude circle and in the column when together they hold enough (spectral_dynamics.F90:1142-1144).  Ignored for
   * 'grid' tracers like the reference (:364-367). */
  int tracer_hole_filling[ISCA_MAX_TRACERS];
  /* tracer_sms of the field_table entries ([k] = tracer k+1, tracer 1 included): hs_forcing's source and sink per tracer (hs_forcing.F90:250-265).
   * tracer_sms[k] = 0: hs_forcing_nml's trflux / trsink (no tracer_sms method in the entry); 1: tracer_flux[k] / tracer_sink[k] (the entry's
   * "flux=" / "sink=" parameters, each defaulting to the namelist value; sink < 0 in days like trsink; 'off' and 'none' are flux = sink = 0:
   * no tendency from hs_forcing).  Only with physics = 0 (hs_forcing is the physics). */
  int tracer_sms[ISCA_MAX_TRACERS];
  double tracer_flux[ISCA_MAX_TRACERS], tracer_sink[ISCA_MAX_TRACERS];
  /* advect_vert of the field_table entries ([k] = tracer k+1, tracer 1 included; spectral_dynamics.F90:395-408): -1 = the standard scheme of the
   * entry's representation (the one the fused kernels implement: 'grid' finite_volume_parabolic, 'spectral' second_centered), else
   * 0 second_centered, 1 fourth_centered, 2 van_leer_linear, 3 finite_volume_parabolic.  A 'grid' tracer applies it to the new level after the
   * horizontal step (:1161); a 'spectral' tracer to the current level (centred schemes) or the previous one (finite-volume schemes, :1135-1141).
   * A non-standard scheme runs vert_advection on whole columns in a kernel of its own (4..64 levels), like vert_advect_uv / vert_advect_t. */
  int tracer_advect_vert[ISCA_MAX_TRACERS];
  /* hs_forcing_nml: local_heating_option (hs_forcing.F90:87-94, 233-238, 728-769): 0 = '' (none), 1 = 'Isidoro' -- an analytic heat source added to the
   * temperature tendency after the Newtonian damping, srfamp exp(-((lon - xcenter)/xwidth)^2 / 2) exp(-((lat - ycenter)/ywidth)^2 / 2)
   * exp((p_full - p_s)/vert_decay): local_heating_srfamp in K/day, the widths and centres in degrees, local_heating_vert_decay in Pa.
   * ('from_file' needs interpolator_mod's data files: outside the device core.)  Only with physics = 0. */
  int local_heating_option;
  double local_heating_srfamp, local_heating_xwidth, local_heating_ywidth, local_heating_xcenter, local_heating_ycenter, local_heating_vert_decay;
} isca_dyn_config;

/* fills the defaults of the reference's namelists + the Held-Suarez test case values */
int isca_dyn_config_default(isca_dyn_config *cfg);
/* ABI guard for bindings: sizeof(isca_dyn_config), sizeof(isca_moist_config), sizeof(isca_shallow_config), sizeof(isca_barotropic_config) */
int isca_config_sizes(size_t *sizes, int n);

/* spectral_dynamics_init + atmosphere_init + hs_forcing_init (tables, device state; no fields yet) */
int isca_dyn_create(const isca_dyn_config *cfg, isca_dyn_t **out);
int isca_dyn_destroy(isca_dyn_t *h);
const char *isca_last_error(void);

/* get_topography (init/spectral_init_cond.F90:167-308): the surface geopotential spectral_init_cond hands to spectral_dynamics, as
 * data: the GLOBAL (lon_max, lat_max) field in m2/s2, the same on every rank, set before isca_dyn_cold_start (or before the set_state
 * calls of a restart).  Default: flat.  It enters the initial surface pressure (spectral_initialize_fields.F90:85), the hydrostatic
 * integral (press_and_geopot.F90:331) and the heights of isca_dyn_get_state("z_full" / "z_half").  get_surf_geopotential
 * (spectral_dynamics.F90:1342) = isca_dyn_get_state(h, "surf_geopotential", ...) (local band). */
int isca_dyn_set_surf_geopotential(isca_dyn_t *h, const double *global_field, size_t count);
/* get_topography with topography_option = 'input' (init/spectral_init_cond.F90:186-245) on data handed over: the GLOBAL (lon_max, lat_max) height field in m
 * (the file's topog_field_name, 'zsurf') and land mask (land_field_name; > 0 = land; may be NULL with ocean_topog_smoothing = 0).  The surface
 * geopotential g * height is spectrally truncated (ocean_topog_smoothing = 0, :231-235) or regularised over the ocean -- compute_lambda + regularize of
 * topog_regularization_mod (init/topog_regularization.F90:75-290; the namelist's default 0.93) -- and becomes the handle's surface geopotential.
 * lambda / fraction_smoothed (may be NULL): what get_topography prints.  world_size 1; before isca_dyn_cold_start. */
int isca_dyn_set_topography(isca_dyn_t *h, const double *height, const double *land_mask, double ocean_topog_smoothing, double *lambda, double *fraction_smoothed);
/* the two public routines of topog_regularization_mod on caller fields ((lon, lat) global; ocean_mask: 1 = ocean, 0 = land) */
int isca_topog_regularize(isca_dyn_t *h, double lambda, const double *ocean_mask, const double *field, double *smoothed, double *fraction_smoothed);
int isca_topog_compute_lambda(isca_dyn_t *h, double ocean_topog_smoothing, const double *ocean_mask, const double *field, double *lambda, double *fraction_smoothed);
/* read_data for a host without netCDF (the Fortran drop-in reads INPUT/<topog_file_name> with it): record `record` of a variable of a netCDF classic /
 * 64-bit-offset file as doubles; count_out = its number of values, copied to out when out != NULL and count >= count_out */
int isca_nc_read_variable(const char *path, const char *var_name, int record, double *out, size_t count, size_t *count_out);

/* read_restart_or_do_coldstart (spectral_dynamics.F90:580-630) + spectral_initialize_fields */
int isca_dyn_cold_start(isca_dyn_t *h);

/* atmosphere(Time) x nsteps: hs_forcing -> spectral_dynamics -> time-level rotation
 * (atmosphere.F90:276-352).  Asynchronous on the handle's stream unless sync != 0. */
int isca_dyn_step(isca_dyn_t *h, int nsteps, int sync);
/* Waits for the handle's stream and raises what the steps since the last synchronisation left behind (temperatures outside valid_range_t: the
 * reference's FATAL, spectral_dynamics.F90:940).  COLLECTIVE on a handle with a communicator (isca_dyn_comm_init): the verdict is summed over the
 * ranks so that every rank raises when one band is out of range -- every rank must call it (and pass the same `sync` to isca_dyn_step /
 * isca_dyn_dynamics) at the same point of the run, like the reference's error_mesg(..., FATAL), which stops all PEs. */
int isca_dyn_synchronize(isca_dyn_t *h);

/* The physics / dynamics seam of atmosphere.F90:300-329 for a host that keeps a physics package of its own (physics = 2):
 *   spectral_dynamics(Time, psg_final, ug_final, vg_final, tg_final, tracer_attributes, grid_tracers_final, time_level_out,
 *                     dt_psg, dt_ug, dt_vg, dt_tg, dt_tracers, wg_full, p_full, p_half, z_full)   (spectral_dynamics.F90:780-795)
 * One call = one step: the tendencies the host's physics accumulated ((lon, lat_local, lev) arrays, null = zero; host memory, or
 * device memory with on_device != 0) go through the implicit step; results come back through isca_dyn_get_state (u, v, T, ps, tracer of
 * the new level, "wg_full", "p_full", "p_half", "z_full").  The physics reads the PREVIOUS level's u, v, T, tracer ("tr_atm": the
 * never-filtered copy atmosphere_mod keeps, atmosphere.F90:95) and the CURRENT level's pressures, as atmosphere.F90:304-317 passes them,
 * and isca_dyn_delta_t gives its time step (dt_atmos on the first step, else 2 dt_atmos).  dt_psg: the physics packages of this path
 * leave it zero (atmosphere.F90:298); not carried.  dt_tracers is (lon, lat_local, lev, num_tracers): one block per tracer of the
 * field_table, tracer index slowest, like the reference's array.  isca_dyn_set_tendencies hands the arrays over without stepping, for a sharded run
 * driven phase by phase (isca_dyn_step_phase). */
int isca_dyn_dynamics(isca_dyn_t *h, const double *dt_ug, const double *dt_vg, const double *dt_tg, const double *dt_tracers,
                      int on_device, int sync);
int isca_dyn_set_tendencies(isca_dyn_t *h, const double *dt_ug, const double *dt_vg, const double *dt_tg, const double *dt_tracers,
                            int on_device);
int isca_dyn_delta_t(isca_dyn_t *h, double *delta_t);

/* --- multi-GPU: one step split at its two lat<->m exchange points (transforms.F90:970-1056).
 * phase 0: grid tendencies + FFT            -> send buffer "fwd" (+ tracer halo rows)
 * phase 1: Legendre analysis, spectral update, Legendre synthesis -> send buffer "inv"
 * phase 2: inverse FFT, fixer partial sums  -> 8 doubles to all-reduce (isca_dyn_reduce_buffer)
 * phase 3: fixers applied, time levels rotated.
 * raw_filter_coeff /= 1 (Robert-Asselin-Williams, leapfrog.F90:88-105) ends the step with phases 5 and 6 instead of 3: phase 5 = the fixers,
 *          the filter's adjustment of the new spectral level and the Legendre synthesis of the gradients taken from it -> send buffer
 *          "raw" (which = 2: 2 L + 2 level-fields); phase 6 = their FFT, time levels rotated.
 * phase 4 (between 0 and 1, after the halo rows of the grid tracer have been exchanged and before the "fwd" all-to-all): the tracer's
 *          transport, on the handle's side stream, so that it runs under the exchange; a no-op without the tracer.
 * The host performs the all-to-all / all-reduce between phases (isca_amd/parallel.py). */
int isca_dyn_step_phase(isca_dyn_t *h, int phase);
int isca_dyn_exchange_buffers(isca_dyn_t *h, int which /*0 fwd, 1 inv, 2 raw*/, void **send, void **recv,
                              size_t *bytes_per_peer);
int isca_dyn_reduce_buffer(isca_dyn_t *h, void **buf, size_t *count);
/* grid-tracer halo rows (the mpp_update_domains of fv_advection.F90:161-162,259): after phase 0 the host sends
 * send_lo to rank-1 (which receives it as recv_hi) and send_hi to rank+1 (as recv_lo); *bytes = 0 if no tracer. */
int isca_dyn_halo_buffers(isca_dyn_t *h, void **send_lo, void **send_hi, void **recv_lo, void **recv_hi, size_t *bytes);

/* Pure host function (no GPU needed): the dealing of zonal wavenumbers m = 0..num_fourier to `world_size`
 * ranks used by the lat<->m exchange (replaces the contiguous m-blocks of spec_mpp.F90:78-80 /
 * mpp_domains_define.inc:164-250 by a boustrophedon deal that balances the triangle).
 * m_of_slot[q*m_local + ml] = global m owned by rank q in local slot ml, or -1 for padding;
 * *m_local = slots per rank.  m_of_slot must hold world_size*ceil((num_fourier+1)/world_size) ints. */
int isca_wavenumber_dealing(int num_fourier, int world_size, int *m_of_slot, int *m_local);

/* state access in the reference's layouts.  name is one of:
 *  grid 3-D (lon,lat_local,lev):  "ug","vg","tg","vorg","divg","wg_full","p_full","z_full","tr"
 *  grid 3-D half levels (lev+1):  "p_half","z_half"
 *  grid 2-D:                      "psg"; moist physics: "t_surf" (mixed-layer temperature), "precip" (last step, kg/m2/s)
 *  spectral (m,n,lev) complex:    "vors","divs","ts";  (m,n): "ln_ps"     [global m; local m on world_size>1]
 * time_level: 0 = previous, 1 = current (ignored for single-level fields). */
int isca_dyn_get_state(isca_dyn_t *h, const char *name, int time_level, double *host, size_t count);
int isca_dyn_set_state(isca_dyn_t *h, const char *name, int time_level, const double *host, size_t count);
/* after set_state of grid fields: rebuild the spectral side like complete_update_of_future
 * (spectral_dynamics.F90:1416-1454) for the given time level */
int isca_dyn_complete_update(isca_dyn_t *h, int time_level);
/* Restart (replaces read_restart_or_do_coldstart's restart branch, spectral_dynamics.F90:509-575, and the
 * time_pointers of atmosphere.F90:207-211): set the leapfrog pointers first (0-based storage slots of
 * `previous` and `current`), then set_state every array of both levels, then refresh_derived, which rebuilds
 * from the spectral state of `current` the grid fields the step keeps between calls (vorg, divg, grad T,
 * grad ln ps) without touching the grid u, v, T, ps just set (world_size == 1). */
int isca_dyn_set_time_pointers(isca_dyn_t *h, int previous, int current, long step_count);
int isca_dyn_refresh_derived(isca_dyn_t *h);

/* tables: "sin_lat","wts_lat","deg_lat","deg_lon","pk","bk","legendre" (m,n,lat_max/2),
 * "eigen_laplacian" (m,n), "wave_matrix" (lev,lev,0:num_spherical-1) for the current delta_t */
int isca_dyn_get_table(isca_dyn_t *h, const char *name, double *host, size_t count);
int isca_dyn_get_info(isca_dyn_t *h, const char *name, long *value);   /* "step","previous","current","lat_local","lat_start","m_local","kernels_per_step","tracer","inverse_batch" (level-fields of the step's synthesis batch: 7 L + 3, or 6 L + 2 when the inverse FFT forms the x-derivatives), "lazy_fixers" (1: the fixers' corrections stay pending on a new level and are applied by its readers) */
/* What idealized_moist_phys_mod keeps beside the fields, for handing over a RUNNING model through set_state (a restart resets it, as
 * idealized_moist_phys_init does): "phys_calls" = calls of the physics since init -- 0: the next call is the first (gust = 1 m/s,
 * idealized_moist_phys.F90:592), > 0: vert_turb_driver's constant_gust (:1262).  Also readable through isca_dyn_get_info. */
int isca_dyn_set_info(isca_dyn_t *h, const char *name, long value);
/* Restart files written and read by the library itself, in the netCDF classic / 64-bit-offset format (what fms_io writes), without a netCDF
 * library: <directory>/spectral_dynamics.res.nc (spectral_dynamics_end, spectral_dynamics.F90:1502-1531: previous, current, pk, bk,
 * vors/divs/ts/ln_ps _real/_imag, ug, vg, tg, psg, every tracer by its field_table name (+ _real/_imag for a 'spectral' one), vorg, divg,
 * surf_geopotential), <directory>/atmosphere.res.nc (atmosphere_end, atmosphere.F90:362-375: time_pointers, ug, vg, tg, psg, atmosphere_mod's
 * tracer copies, wg_full) and, with the moist package, <directory>/mixed_layer.res.nc (t_surf; mixed_layer.F90:813) -- two records along Time,
 * one per leapfrog level, fms_io's xaxis_N / yaxis_N / zaxis_N naming.  isca_dyn_read_restart is the restart branch of
 * read_restart_or_do_coldstart (:509-575) + atmosphere_init (atmosphere.F90:197-223): resolution checks with the reference's messages, both time
 * levels, the time pointers, the file's surface geopotential, then isca_dyn_refresh_derived.  tracer_names: comma-separated field_table names of
 * tracers 1..num_tracers (NULL: "sphum", "tracer2", ...).  With more than one rank every rank writes / reads its own piece, named as fms_io names the
 * pieces of a distributed file (<name>.nc.NNNN): its latitude band of the grid fields, its zonal wavenumbers of the spectral arrays; the same number
 * of ranks reads them back.  The one-rank files are the ones isca_amd/restart.py writes and reads. */
int isca_dyn_write_restart(isca_dyn_t *h, const char *directory, const char *tracer_names);
int isca_dyn_read_restart(isca_dyn_t *h, const char *directory, const char *tracer_names);
int isca_dyn_restart_exists(const char *directory);     /* file_exist('INPUT/spectral_dynamics.res.nc') (spectral_dynamics.F90:512) */
/* the file layer alone (no device): writes a small fms_io-style file to out_path (if given); sums[0..2] = sum, first, last value of record
 * `record` of variable var_name of in_path (if given) */
int isca_restart_file_selftest(const char *out_path, const char *in_path, const char *var_name, int record, double *sums);

/* --- transforms_mod entry points (host buffers, Fortran layouts; nlev = size of 3rd dim) ---
 * On more than one rank the five grid <-> spherical routines (spherical_to_grid, grid_to_spherical, vor_div_from_uv_grid, uv_grid_from_vor_div, trans_filter)
 * are COLLECTIVE calls through the library's communicator (isca_dyn_comm_init): grid arrays are the rank's latitude band, spectral arrays the whole
 * (0:num_fourier, 0:num_spherical) window on every rank -- a spectral result is gathered on every rank (the reference hands each rank its window
 * ms:me, transforms.F90:970-1056).  The remaining ones stay world_size == 1. */
int isca_trans_spherical_to_grid(isca_dyn_t *h, const double *spherical, double *grid, int nlev);   /* transforms.F90:379 */
int isca_trans_grid_to_spherical(isca_dyn_t *h, const double *grid, double *spherical, int nlev, int do_truncation); /* :462 */
int isca_vor_div_from_uv_grid(isca_dyn_t *h, const double *u, const double *v, double *vor, double *div, int nlev);  /* :742 */
int isca_uv_grid_from_vor_div(isca_dyn_t *h, const double *vor, const double *div, double *u, double *v, int nlev);  /* :700 */
int isca_horizontal_advection(isca_dyn_t *h, const double *field_spec, const double *u, const double *v, double *tendency, int nlev); /* :808 */
int isca_trans_spherical_to_fourier(isca_dyn_t *h, const double *spherical, double *fourier, int nlev); /* spherical_fourier.F90:177; fourier (m, lat, lev) */
int isca_trans_fourier_to_spherical(isca_dyn_t *h, const double *fourier, double *spherical, int nlev); /* spherical_fourier.F90:264 */
int isca_trans_grid_to_fourier(isca_dyn_t *h, const double *grid, double *fourier, int nlev);           /* grid_fourier.F90:129; fourier (0:num_fourier, lat, lev) */
int isca_trans_fourier_to_grid(isca_dyn_t *h, const double *fourier, double *grid, int nlev);           /* grid_fourier.F90:155 */
int isca_trans_filter(isca_dyn_t *h, double *grid, const double *filter /* (m,n) real or NULL */, int nlev);             /* transforms.F90:555 trans_filter */
int isca_area_weighted_global_mean(isca_dyn_t *h, const double *field2d, double *mean);                 /* transforms.F90:1059 */

/* hs_forcing(...) on caller-supplied fields (hs_forcing.F90:148): tendencies are accumulated into udt,vdt,tdt */
int isca_hs_forcing(isca_dyn_t *h, double dt, const double *p_half, const double *p_full, const double *u,
                    const double *v, const double *t, double *udt, double *vdt, double *tdt);

/* idealized_moist_phys(...) on caller-supplied columns (idealized_moist_phys.F90:819-1340, Frierson options; handle created with
 * physics = 1): ncol independent columns, arrays (col, lev) / (col, lev+1) with col fastest; fields of the previous time level,
 * pressures of both levels, heights of the current one, as the reference passes them.  t_surf is updated in place (mixed_layer);
 * gust is the value left by the previous call of vert_turb_driver (1.0 on the first call, then constant_gust).
 * Returns the tendencies and the rain rate (kg/m2/s; precip may be NULL). */
int isca_idealized_moist_phys(isca_dyn_t *h, int ncol, double delta_t, double gust, const double *rad_lat, const double *u_prev,
                              const double *v_prev, const double *t_prev, const double *q_prev, const double *p_half_prev,
                              const double *p_full_prev, const double *p_half_cur, const double *p_full_cur, const double *z_half_cur,
                              const double *z_full_cur, double *t_surf, double *dt_u, double *dt_v, double *dt_t, double *dt_q,
                              double *precip);

/* --- diagnostics: what spectral_diagnostics sends to diag_manager every step (spectral_dynamics.F90:1705-1867),
 * accumulated on the device for the time means of the diag_table.  Field names as registered by the reference:
 * ps, ucomp, vcomp, temp, vor, div, omega, sphum, ucomp_sq, vcomp_sq, ucomp_vcomp, temp_sq, ucomp_temp, vcomp_temp,
 * omega_sq, omega_temp, ucomp_omega, vcomp_omega, vcomp_vor, wspd; with the moist package also the 2-D fields precipitation
 * (module atmosphere, idealized_moist_phys.F90:672) and t_surf (module mixed_layer, mixed_layer.F90:359). */
int isca_dyn_diag_select(isca_dyn_t *h, const char *comma_separated_names);     /* "" switches the accumulation off */
int isca_dyn_diag_read(isca_dyn_t *h, const char *name, double *host, size_t count, long *nsteps, int reset);   /* mean over the steps since the last reset; host may be NULL */
/* History files from the library: diag_manager's part for these fields, for a host without Python around it (the Fortran drop-in).  diag_open reads
 * a diag_table -- the path of the run directory's file, or its text -- in the reference's format (title, base date, "file", freq, "units", format,
 * "time_units", "long_name" lines and "module", "field", "output_name", "file", "time_sampling", time_avg, "other_opts", precision lines;
 * src/shared/diag_manager/diag_table.F90, src/extra/python/isca/diagtable.py:5-35), selects the union of its fields on the device, and from then on every
 * step of isca_dyn_step / isca_dyn_dynamics counts: at the end of each file's output interval one record of time means (time_avg = .true.) or of
 * samples (.false.) is appended to <directory>/<file>.nc (netCDF classic; lon, lat, pfull, phalf, time, average_T1/_T2/_DT, pk, bk and the fields with
 * long_name / units / cell_methods as spectral_dynamics.F90:1554-1700 registers them; with more than one rank every rank's band as <file>.nc.NNNN).
 * start_seconds: the model time at the first step (a restarted run's Time).  Entries of modules the device core does not hold are refused by name.
 * diag_close finishes the files (a handle destroyed without it leaves them complete up to the last record). */
int isca_dyn_diag_open(isca_dyn_t *h, const char *diag_table_path_or_text, const char *directory, double start_seconds);
int isca_dyn_diag_close(isca_dyn_t *h);

/* --- RCCL communicator of the sharded step (world_size > 1) ----------------------------------------
 * With a communicator, isca_dyn_step runs the whole sharded step on the handle's stream: the lat<->m all-to-alls
 * (replacing mpp_transmit in transpose_fourier / reverse_transpose_fourier, transforms.F90:990-1054), the tracer halo
 * rows (mpp_update_domains, fv_advection.F90:161-162,259) and the all-reduce of the fixer sums (transforms.F90:1059-1077)
 * as RCCL calls between the kernels, nothing returning to the host.  Rank 0 calls isca_comm_get_unique_id and
 * distributes the 128 bytes; every rank calls isca_dyn_comm_init (collective).  RCCL is loaded with dlopen. */
int isca_comm_get_unique_id(void *id128);
int isca_dyn_comm_init(isca_dyn_t *h, const void *id128);
/* collective over all ranks after isca_dyn_comm_init: rank-tagged patterns through every exchange of the sharded step; non-zero if
 * this rank received wrong data (the host driver then keeps the exchanges in torch.distributed) */
int isca_dyn_comm_check(isca_dyn_t *h);
int isca_comm_selftest(int device, double *max_err);     /* one-rank communicator: load RCCL, run every collective once */
/* With ISCA_COMM=ipc in the environment of the rank that draws the id, the communicator is a host-staged exchange through files
 * mapped by every rank instead of RCCL (isca_amd/csrc/comm_ipc.cpp): ranks may then SHARE one GPU, so the sharded C++ step loop --
 * the exchange schedule that stands for transpose_fourier / reverse_transpose_fourier (transforms.F90:970-1056),
 * mpp_update_domains (fv_advection.F90:161-162) and mpp_sum -- can be verified with 2, 4, 8 processes on a one-GPU box.
 * Every exchange synchronises stream and host: a verification vehicle, not a fast path.  ISCA_IPC_TIMEOUT_S (120): how long a
 * rank waits for the others before it stops with an error; a rank whose step throws releases its peers with an error. */
const char *isca_dyn_comm_kind(isca_dyn_t *h);           /* "rccl", "ipc", or "" without a communicator */
/* For a host without a message-passing layer of its own -- the Fortran drop-in when mpp is built without MPI (mpp_pe() = 0 everywhere), or any
 * launcher that only exports environment variables: the decomposition contract of spec_mpp.F90:61-80 / atmosphere_domain (atmosphere.F90:390) is then
 * carried by the environment.  isca_env_rank: rank, number of ranks and rank on the node from ISCA_RANK / ISCA_WORLD_SIZE / ISCA_LOCAL_RANK, else
 * torchrun's RANK / WORLD_SIZE / LOCAL_RANK, Open MPI's OMPI_COMM_WORLD_*, PMI_RANK / PMI_SIZE, SLURM_PROCID / SLURM_NTASKS (one rank when none is
 * set).  isca_dyn_comm_init_env (collective; no-op with one rank): rank 0 draws the id and leaves it in the file ISCA_COMM_ID_FILE names, the others
 * wait for it (a file older than the waiting process by more than a minute is an earlier run's and is ignored; rank 0 removes its file once every rank has
 * answered the check), then isca_dyn_comm_init + isca_dyn_comm_check on every rank. */
int isca_env_rank(int *rank, int *world_size, int *local_rank);
int isca_dyn_comm_init_env(isca_dyn_t *h);

/* --- components of the step on caller fields (world_size == 1) -------------------------------------
 * Each entry replaces one public routine the reference's callers use on its own, and runs the kernel or
 * device function the step itself uses.  Spectral arrays (m,n,lev) complex, grid arrays (lon,lat,lev). */
int isca_compute_laplacian(isca_dyn_t *h, const double *spherical, double *laplacian, int nlev, int power);   /* spherical.F90:354-406; power = 1 is the default */
int isca_compute_gradient_cos(isca_dyn_t *h, const double *spherical, double *deriv_lon, double *deriv_lat, int nlev);   /* spherical.F90:270-351; also compute_lon/lat_deriv_cos (NULL skips an output) */
int isca_compute_ucos_vcos(isca_dyn_t *h, const double *vorticity, const double *divergence, double *u_cos, double *v_cos, int nlev);     /* spherical.F90:409-469 */
int isca_compute_vor_div(isca_dyn_t *h, const double *u_div_cos, const double *v_div_cos, double *vorticity, double *divergence, int nlev);   /* spherical.F90:472-561 */
int isca_triangular_truncation(isca_dyn_t *h, double *spherical, int nlev);                             /* spherical.F90:564-600, in place */
int isca_divide_by_cos(isca_dyn_t *h, double *grid, int nlev, int power);                               /* transforms.F90:599-648 divide_by_cos (1) / divide_by_cos2 (2), in place */
int isca_mass_weighted_global_integral(isca_dyn_t *h, const double *field, const double *surf_press, double *integral);   /* global_integral.F90:49-81 */
int isca_pressure_variables(isca_dyn_t *h, const double *surf_p, double *p_half, double *ln_p_half, double *p_full, double *ln_p_full);   /* press_and_geopot.F90:152-221 */
int isca_compute_geopotential(isca_dyn_t *h, const double *t, const double *ln_p_half, const double *ln_p_full,
                              double *geopot_full, double *geopot_half);                              /* press_and_geopot.F90:327-359 on the handle's surface geopotential, without q_grid */
/* ... with the routine's own arguments: surf_geopotential [lat][lon] (NULL: the handle's), q_grid [lev][lat][lon] (NULL: not given; required, as in the
   reference :343, when the handle has use_virtual_temperature) */
int isca_compute_geopotential_surf(isca_dyn_t *h, const double *t, const double *ln_p_half, const double *ln_p_full, const double *surf_geopotential,
                                   const double *q_grid, double *geopot_full, double *geopot_half);   /* press_and_geopot.F90:314-359 */
int isca_a_grid_horiz_advection(isca_dyn_t *h, const double *u, const double *v, const double *q, double dt, double *tendency);   /* fv_advection.F90:126-207; tendency is accumulated */
int isca_vert_advection_ppm(isca_dyn_t *h, double dt, const double *w, const double *surf_p, const double *r, double *rdt);   /* vert_advection.F90:70-478, FINITE_VOLUME_PARABOLIC / ADVECTIVE_FORM, dz = dpk + dbk*surf_p */
int isca_hs_tracer_source_sink(isca_dyn_t *h, const double *surf_p, const double *r, double *rdt);      /* hs_forcing.F90:683-724; rdt is accumulated */
int isca_vert_advection_centered(isca_dyn_t *h, const double *w, const double *surf_p, const double *r, double *rdt);   /* vert_advection.F90:185-193, 467-470: SECOND_CENTERED / ADVECTIVE_FORM, dz = dpk + dbk*surf_p; rdt = the tendency */
/* press_and_geopot.F90:363-387 compute_pressures_and_heights(t_grid, ps_grid, surf_geopotential, z_full, z_half, p_full, p_half [, q_grid]):
 * the handle's surface geopotential; q (NULL = none) enters through the virtual temperature when use_virtual_temperature is set */
int isca_compute_pressures_and_heights(isca_dyn_t *h, const double *t, const double *ps, const double *q, double *z_full, double *z_half,
                                       double *p_full, double *p_half);
/* leapfrog.F90:58-105 on n real values (a complex array = 2 n of them): leapfrog_2level_A -- part = prev - 2 cur; cur += robert part raw;
 * fut = prev + delta_t dt_a (fut may be prev's storage) -- and leapfrog_2level_B -- cur += robert fut raw; fut += robert (part + fut) (raw - 1) */
int isca_leapfrog_2level_a(isca_dyn_t *h, size_t n, const double *prev, double *cur, double *fut, const double *dt_a, double delta_t,
                           double robert_coeff, double raw_filter_coeff, double *part);
int isca_leapfrog_2level_b(isca_dyn_t *h, size_t n, double *cur, double *fut, const double *part, double robert_coeff, double raw_filter_coeff);
/* gauss_and_legendre.F90:111-183 compute_gaussian(sin_hem, wts_hem, n_hem) and :47-108 compute_legendre(legendre, num_fourier, fourier_inc,
 * num_spherical, sin_lat, n_lat), legendre(0:num_fourier, 0:num_spherical, n_lat): host tables, no handle */
int isca_compute_gaussian(int n_hem, double *sin_hem, double *wts_hem);
int isca_compute_legendre(int num_fourier, int fourier_inc, int num_spherical, const double *sin_lat, int n_lat, double *legendre);

/* the three stages of the spectral update, run by the step's own kernel on caller data (num_levels 3-D arrays) */
int isca_implicit_correction(isca_dyn_t *h, double *dt_divs, double *dt_ts, double *dt_ln_ps, const double *divs_previous,
                             const double *divs_current, const double *ts_previous, const double *ts_current,
                             const double *ln_ps_previous, const double *ln_ps_current, double delta_t);   /* implicit.F90:241-286 */
int isca_compute_spectral_damping(isca_dyn_t *h, int which, const double *field_previous, double *dt_field, double delta_t);   /* spectral_damping.F90:172-291; which 0 generic, 1 vor, 2 div */
int isca_leapfrog(isca_dyn_t *h, double *previous, double *current, const double *dt_field, double delta_t, double robert_coeff);   /* leapfrog.F90:58-105, A then B, future = previous */

/* --- benchmarking helpers: transform pair with data resident in HBM ------------------------------
 * Runs `reps` (s2g, g2s) pairs over nfields level-fields on device buffers owned by the handle and
 * returns the average time of one pair in milliseconds measured with HIP events on the handle's
 * stream.  kernel_ms[0..3] = {legendre_inv, fft_inv, fft_fwd, legendre_fwd} per-launch averages. */
int isca_bench_transform_pair(isca_dyn_t *h, int nfields, int reps, double *pair_ms, double *kernel_ms);
/* per-kernel average milliseconds over the steps run since the last call (HIP events); names are
 * returned as a ';'-separated list in `names` */
int isca_dyn_kernel_times(isca_dyn_t *h, int enable, double *ms, int max, char *names, size_t names_len, int *n);

#ifdef __cplusplus
}
#endif
#endif
