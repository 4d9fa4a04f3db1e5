/* isca_stirring.h -- stirring_nml of the sibling cores (reference: src/atmos_spectral_barotropic/stirring.F90:60-71): stochastic
 * vorticity forcing in a band of total wavenumbers, AR(1) in time (decay_time), localised in physical space, added to the vorticity
 * tendency between the spectral damping and the leapfrog step (barotropic_dynamics.F90:311, shallow_dynamics.F90:447). */
#ifndef ISCA_STIRRING_H
#define ISCA_STIRRING_H
typedef struct isca_stirring_config {
  double decay_time, amplitude, lat0, lon0, widthy, widthx, B;   /* amplitude = 0 (default): no stirring */
  int do_localize, n_total_forcing_max, n_total_forcing_min, zonal_forcing_min;
  unsigned long long seed;      /* of the library's own uniform random numbers (the reference uses the Fortran runtime's generator,
                                 * which differs between compilers; isca_*_set_stirring_noise supplies the numbers instead) */
} isca_stirring_config;
#endif
