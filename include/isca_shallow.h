/* isca_shallow.h -- C-ABI of the shallow-water sibling core (reference: src/atmos_spectral_shallow).
 *
 * Same transform kernels as the 3-D core (include/isca_dyn.h), one level.  Replaces
 *   atmosphere_mod (shallow)   atmos_spectral_shallow/atmosphere.F90:117-250   (atmosphere_init / atmosphere / atmosphere_end)
 *   shallow_dynamics_mod       atmos_spectral_shallow/shallow_dynamics.F90:217-530
 *   shallow_physics_mod        atmos_spectral_shallow/shallow_physics.F90:106-194
 * Conventions as in isca_dyn.h: fp64, grid (lon, lat), spectral (m, n) complex interleaved, 0 = success, message from
 * isca_last_error().  Single GPU (the 2-D model is launch-bound; it does not shard).
 */
#ifndef ISCA_SHALLOW_H
#define ISCA_SHALLOW_H
#include <stddef.h>
#include "isca_stirring.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct isca_shallow isca_shallow_t;

/* shallow_dynamics_nml (shallow_dynamics.F90:117-194), shallow_physics_nml (shallow_physics.F90:88-103), main_nml dt_atmos.
 * Not carried: reading init_cond_file itself (netCDF; the host passes its fields to isca_shallow_init_from_grid), fourier_inc /= 1,
 * rhomboidal truncation, the exponential damping option. */
typedef struct isca_shallow_config {
  int num_lon, num_lat, num_fourier, num_spherical;
  double dt_atmos;
  int damping_order;
  double damping_coeff, robert_coeff, robert_coeff_tracer;
  double h_0, u_deep_mag, n_merid_deep_flow, u_upper_mag_init;
  int spec_tracer, grid_tracer;
  double lon_centre_init_cyc, lat_centre_init_cyc, lon_centre_init_acyc, lat_centre_init_acyc;
  double init_vortex_radius_deg, init_vortex_vor_f, init_vortex_h_h_0;
  int add_initial_vortex_pair, add_initial_vortex_as_height;
  double valid_range_v[2];
  /* shallow_physics_nml; phys_h_0 is that namelist's own h_0 */
  double fric_damp_time, therm_damp_time, phys_h_0, h_amp, h_lon, h_lat, h_width, h_itcz, itcz_width;
  int device;
  isca_stirring_config stirring;
  double radius, omega;         /* constants_nml (the shallow-water test cases run a giant planet: 55000e3 m, 1.6e-4 1/s) */
} isca_shallow_config;

int isca_shallow_config_default(isca_shallow_config *cfg);
/* shallow_dynamics_init + shallow_physics_init: tables, h_eq, deep_geopot */
int isca_shallow_create(const isca_shallow_config *cfg, isca_shallow_t **out);
int isca_shallow_destroy(isca_shallow_t *h);
/* the Time == Time_init branch of shallow_dynamics_init (:330-408): initial h, vor, div, tracers */
int isca_shallow_cold_start(isca_shallow_t *h);
/* the initial_condition_from_input_file branch (:332-337): vorticity, divergence and height ANOMALY (h_0 is added) on the model grid,
 * as the reference's interpolator delivers them from init_cond_file; tracers start as in the cold start */
int isca_shallow_init_from_grid(isca_shallow_t *h, const double *vor, const double *div, const double *height);
/* atmosphere(Time) x nsteps (atmosphere.F90:164-200): shallow_physics -> shallow_dynamics -> time-level rotation;
 * checks valid_range_v on return ("meridional wind out of valid range") */
int isca_shallow_step(isca_shallow_t *h, int nsteps);
/* state: grid "u","v","vor","div","h","tr","trs" (time_level 0 = previous, 1 = current), "stream","pv","h_eq","deep_geopot";
 * spectral "vors","divs","hs","trss","stirs" as (m, n) complex */
int isca_shallow_get_state(isca_shallow_t *h, const char *name, int time_level, double *host, size_t count);
int isca_shallow_set_state(isca_shallow_t *h, const char *name, int time_level, const double *host, size_t count);
/* "previous", "current" (0/1 storage slots), "step" */
int isca_shallow_get_info(isca_shallow_t *h, const char *name, long *value);
/* restart branch: restore the time pointers after both levels of the spectral and grid state have been set */
/* the (0:num_fourier, 0:num_spherical, 2) uniform random numbers in [0,1) the NEXT step's stirring uses instead of drawing its own
 * (stirring.F90:205-210 calls random_number); consumed by that step.  The AR(1) state is the spectral field "stirs" of get/set_state. */
int isca_shallow_set_stirring_noise(isca_shallow_t *h, const double *ran, size_t count);
int isca_shallow_set_time_pointers(isca_shallow_t *h, int previous, int current, long step_count);

#ifdef __cplusplus
}
#endif
#endif
