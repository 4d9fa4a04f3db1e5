! The physics / dynamics seam of atmosphere.F90:300-329 from the reference's language: the host keeps its physics package (here
! hs_forcing_mod, evaluated through the library's own entry points on the fields the library hands out, exactly where a maintainer
! would call the Fortran hs_forcing) and gives the tendencies to the device dynamics -- spectral_dynamics(..., dt_ug, dt_vg, dt_tg,
! dt_tracers, ...) (spectral_dynamics.F90:780-795) = isca_dyn_dynamics on a handle created with physics = 2.
! tests/test_gpu_fortran_binding.py compares the printed values with the reference run (tests/golden/run_T21L25.npz).
program drive_external_physics
use iso_c_binding
use isca_dyn_c
implicit none
type(isca_dyn_config) :: cfg
type(c_ptr) :: handle
integer, parameter :: nlon = 64, nlat = 32, nlev = 25, nsteps = 144
real(c_double), allocatable, dimension(:,:,:) :: um, vm, tm, rm, p_half, p_full, dt_ug, dt_vg, dt_tg, dt_tr, tg, ug
real(c_double) :: delta_t
integer :: istep
integer(c_size_t) :: n3, n3h

if(.not. check_abi()) then
  print *, 'FATAL: isca_dyn_c does not match the library (struct sizes differ)'; stop 2
endif
if(isca_dyn_config_default(cfg) /= 0) stop 3
cfg%lon_max = nlon; cfg%lat_max = nlat; cfg%num_fourier = 21; cfg%num_spherical = 22; cfg%num_levels = nlev
cfg%dt_atmos = 600.0d0; cfg%damping_order = 4; cfg%scale_heights = 6.0d0; cfg%exponent = 7.5d0; cfg%surf_res = 0.5d0
cfg%physics = 2                       ! atmosphere_mod keeps its physics; the library is spectral_dynamics_mod
if(isca_dyn_create(cfg, handle) /= 0) then
  print *, 'FATAL: ', isca_message(); stop 4
endif
if(isca_dyn_cold_start(handle) /= 0) then
  print *, 'FATAL: ', isca_message(); stop 5
endif
allocate(um(nlon,nlat,nlev), vm(nlon,nlat,nlev), tm(nlon,nlat,nlev), rm(nlon,nlat,nlev), p_full(nlon,nlat,nlev), p_half(nlon,nlat,nlev+1))
allocate(dt_ug(nlon,nlat,nlev), dt_vg(nlon,nlat,nlev), dt_tg(nlon,nlat,nlev), dt_tr(nlon,nlat,nlev), tg(nlon,nlat,nlev), ug(nlon,nlat,nlev))
n3 = size(um, kind=c_size_t); n3h = size(p_half, kind=c_size_t)
do istep = 1, nsteps
  ! atmosphere.F90:286-317: delta_t, the previous level's u, v, T, tracer and the current level's pressures go to the physics
  if(isca_dyn_delta_t(handle, delta_t) /= 0) stop 6
  if(isca_dyn_get_state(handle, 'ug'//c_null_char, 0_c_int, um, n3) /= 0) stop 7
  if(isca_dyn_get_state(handle, 'vg'//c_null_char, 0_c_int, vm, n3) /= 0) stop 7
  if(isca_dyn_get_state(handle, 'tg'//c_null_char, 0_c_int, tm, n3) /= 0) stop 7
  if(isca_dyn_get_state(handle, 'tr_atm'//c_null_char, 0_c_int, rm, n3) /= 0) stop 7
  if(isca_dyn_get_state(handle, 'p_half'//c_null_char, 1_c_int, p_half, n3h) /= 0) stop 8
  if(isca_dyn_get_state(handle, 'p_full'//c_null_char, 1_c_int, p_full, n3) /= 0) stop 8
  dt_ug = 0.; dt_vg = 0.; dt_tg = 0.; dt_tr = 0.
  if(isca_hs_forcing(handle, delta_t, p_half, p_full, um, vm, tm, dt_ug, dt_vg, dt_tg) /= 0) stop 9
  if(isca_hs_tracer_source_sink(handle, p_half(:,:,nlev+1), rm, dt_tr) /= 0) stop 10
  ! atmosphere.F90:325: spectral_dynamics with the accumulated tendencies
  if(isca_dyn_dynamics(handle, dt_ug, dt_vg, dt_tg, dt_tr, 0_c_int, 1_c_int) /= 0) then
    print *, 'FATAL: ', isca_message(); stop 11
  endif
enddo
if(isca_dyn_get_state(handle, 'tg'//c_null_char, 1_c_int, tg, n3) /= 0) stop 12
if(isca_dyn_get_state(handle, 'ug'//c_null_char, 1_c_int, ug, n3) /= 0) stop 12
if(isca_dyn_get_state(handle, 'tr'//c_null_char, 1_c_int, rm, n3) /= 0) stop 12
write(*,'(a,3es24.16)') 'FORTRAN_STATE Tmin,Tmax,maxabsU=', minval(tg), maxval(tg), maxval(abs(ug))
write(*,'(a,3es24.16)') 'FORTRAN_POINT tg(5,7,20),ug(33,12,3),tr(9,30,25)=', tg(5,7,20), ug(33,12,3), rm(9,30,25)
! isca_dyn_step has no physics to call on this handle: FATAL
if(isca_dyn_step(handle, 1_c_int, 1_c_int) /= 0) write(*,'(a,a)') 'FORTRAN_ERROR ', isca_message()
if(isca_dyn_destroy(handle) /= 0) stop 13
end program drive_external_physics
