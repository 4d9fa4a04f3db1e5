! Test infrastructure (oracle/): drives the REFERENCE's topog_regularization_mod (src/atmos_spectral/init/topog_regularization.F90: compute_lambda,
! regularize -- what get_topography does with ocean_topog_smoothing /= 0, spectral_init_cond.F90:230-245, 285-295) on a height field and a land mask
! handed in as raw files.  get_topography's own branches that call it read their inputs through netCDF (topography_option = 'input' /
! 'interpolated'), which this image cannot build; the two public routines are called directly instead, after spectral_dynamics_init has
! initialised the transforms.  Compiled in place with the reference by oracle/build_ref.py topog; never shipped, never used by the product.
program ref_topog_harness
use constants_mod,         only: constants_init, grav
use fms_mod,               only: fms_init
use time_manager_mod,      only: time_type, set_time, set_calendar_type, NO_CALENDAR
use field_manager_mod,     only: MODEL_ATMOS
use tracer_manager_mod,    only: register_tracers, get_number_tracers
use diag_manager_mod,      only: diag_manager_init
use tracer_type_mod,       only: tracer_type
use spectral_dynamics_mod, only: spectral_dynamics_init
use transforms_mod,        only: get_grid_domain
use topog_regularization_mod, only: compute_lambda, regularize
implicit none
real :: ocean_topog_smoothing = 0.8
namelist /topog_harness_nml/ ocean_topog_smoothing
type(time_type) :: Time, Time_step
type(tracer_type), allocatable, dimension(:) :: tracer_attributes
integer :: ntrace, ntprog, ntdiag, ntfamily, num_tracers, nhum, is, ie, js, je, unit
logical :: dry_model
real, allocatable, dimension(:,:) :: height, land, surf_geopotential, smoothed
logical, allocatable, dimension(:,:) :: ocean_mask
real :: lambda, fraction_smoothed

open(newunit=unit, file='topog_harness.nml', status='old', action='read')
read(unit, nml=topog_harness_nml)
close(unit)
call fms_init()
call constants_init()
call register_tracers(MODEL_ATMOS, ntrace, ntprog, ntdiag, ntfamily)
call set_calendar_type(NO_CALENDAR)
call diag_manager_init()
Time = set_time(0, 0); Time_step = set_time(600, 0)
call get_number_tracers(MODEL_ATMOS, num_prog=num_tracers)
allocate(tracer_attributes(num_tracers))
call spectral_dynamics_init(Time, Time_step, tracer_attributes, dry_model, nhum)
call get_grid_domain(is, ie, js, je)
allocate(height(is:ie,js:je), land(is:ie,js:je), surf_geopotential(is:ie,js:je), smoothed(is:ie,js:je), ocean_mask(is:ie,js:je))
open(newunit=unit, file='in_height.bin', access='stream', form='unformatted', status='old'); read(unit) height; close(unit)
open(newunit=unit, file='in_land.bin', access='stream', form='unformatted', status='old'); read(unit) land; close(unit)
where(land > 0.)          ! spectral_init_cond.F90:223-227
  ocean_mask = .false.
elsewhere
  ocean_mask = .true.
end where
surf_geopotential = grav*height
call compute_lambda(ocean_topog_smoothing, ocean_mask, surf_geopotential, lambda, fraction_smoothed)
call regularize(lambda, ocean_mask, surf_geopotential, smoothed, fraction_smoothed)
open(newunit=unit, file='out_smoothed.bin', access='stream', form='unformatted', status='replace'); write(unit) smoothed; close(unit)
open(newunit=unit, file='out_lambda_fraction.bin', access='stream', form='unformatted', status='replace'); write(unit) lambda, fraction_smoothed; close(unit)
write(*,'(a,2es24.16)') 'TOPOG lambda, fraction_smoothed =', lambda, fraction_smoothed
end program ref_topog_harness
