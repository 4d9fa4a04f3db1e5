/* isca_barotropic.h -- C-ABI of the barotropic-vorticity sibling core (reference: src/atmos_spectral_barotropic).
 *
 * Replaces atmosphere_mod (barotropic) atmos_spectral_barotropic/atmosphere.F90:110-235, barotropic_dynamics_mod
 * barotropic_dynamics.F90:175-420 (barotropic_physics is empty in the reference).  Conventions as in isca_shallow.h.
 */
#ifndef ISCA_BAROTROPIC_H
#define ISCA_BAROTROPIC_H
#include <stddef.h>
#include "isca_stirring.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct isca_barotropic isca_barotropic_t;

/* barotropic_dynamics_nml (barotropic_dynamics.F90:100-141) + main_nml dt_atmos.  Not carried: fourier_inc /= 1,
 * rhomboidal truncation, exponential damping. */
typedef struct isca_barotropic_config {
  int num_lon, num_lat, num_fourier, num_spherical;
  double dt_atmos;
  int damping_order;
  double damping_coeff, damping_coeff_r, robert_coeff;
  double zeta_0;
  int m_0;
  double eddy_width, eddy_lat;
  int spec_tracer, grid_tracer;
  double valid_range_v[2];
  int initial_zonal_wind;       /* 0 = 'zero', 1 = 'two_jets' */
  int device;
  isca_stirring_config stirring;
  double radius, omega;         /* constants_nml (the shallow-water test cases run a giant planet: 55000e3 m, 1.6e-4 1/s) */
} isca_barotropic_config;

int isca_barotropic_config_default(isca_barotropic_config *cfg);
int isca_barotropic_create(const isca_barotropic_config *cfg, isca_barotropic_t **out);   /* barotropic_dynamics_init: tables */
int isca_barotropic_destroy(isca_barotropic_t *h);
int isca_barotropic_cold_start(isca_barotropic_t *h);    /* Time == Time_init branch (:236-277): jets + eddy perturbation, tracers */
int isca_barotropic_step(isca_barotropic_t *h, int nsteps);   /* atmosphere(Time) x nsteps; checks valid_range_v on return */
/* grid "u","v","vor","tr","trs" (time_level 0 = previous, 1 = current), "stream","pv","zonal_u_init" (lat); spectral "vors","trss","stirs" */
int isca_barotropic_get_state(isca_barotropic_t *h, const char *name, int time_level, double *host, size_t count);
int isca_barotropic_set_state(isca_barotropic_t *h, const char *name, int time_level, const double *host, size_t count);
int isca_barotropic_get_info(isca_barotropic_t *h, const char *name, long *value);
/* the (0:num_fourier, 0:num_spherical, 2) uniform random numbers in [0,1) the NEXT step's stirring uses instead of drawing its own
 * (stirring.F90:205-210 calls random_number); consumed by that step.  The AR(1) state is the spectral field "stirs" of get/set_state. */
int isca_barotropic_set_stirring_noise(isca_barotropic_t *h, const double *ran, size_t count);
int isca_barotropic_set_time_pointers(isca_barotropic_t *h, int previous, int current, long step_count);

#ifdef __cplusplus
}
#endif
#endif
