! Example / test driver in the reference's language: the body of a replacement atmosphere_mod (atmosphere.F90:120-352) reduced to its
! calls -- configure from the namelist values of the Held-Suarez test case, cold start, step, read the state back in the reference's
! (lon, lat, lev) layout.  Prints values tests/test_gpu_fortran_binding.py compares with the reference run (tests/golden/run_T21L25.npz).
program drive_held_suarez
use iso_c_binding
use isca_dyn_c
implicit none
type(isca_dyn_config) :: cfg
type(c_ptr) :: handle
integer, parameter :: nlon = 64, nlat = 32, nlev = 25
real(c_double), allocatable :: tg(:,:,:), ug(:,:,:), psg(:,:)
complex(c_double_complex), allocatable :: spec(:,:,:)
real(c_double) :: mean_ps
integer :: nsteps

if(.not. check_abi()) then
  print *, 'FATAL: isca_dyn_c does not match the library (struct sizes differ)'; stop 2
endif
if(isca_dyn_config_default(cfg) /= 0) stop 3
cfg%lon_max = nlon; cfg%lat_max = nlat; cfg%num_fourier = 21; cfg%num_spherical = 22; cfg%num_levels = nlev
cfg%dt_atmos = 600.0d0; cfg%damping_order = 4; cfg%scale_heights = 6.0d0; cfg%exponent = 7.5d0; cfg%surf_res = 0.5d0
if(isca_dyn_create(cfg, handle) /= 0) then
  print *, 'FATAL: ', isca_message(); stop 4
endif
if(isca_dyn_cold_start(handle) /= 0) then
  print *, 'FATAL: ', isca_message(); stop 5
endif
nsteps = 144
if(isca_dyn_step(handle, int(nsteps, c_int), 1_c_int) /= 0) then
  print *, 'FATAL: ', isca_message(); stop 6
endif
allocate(tg(nlon, nlat, nlev), ug(nlon, nlat, nlev), psg(nlon, nlat), spec(0:21, 0:22, nlev))
if(isca_dyn_get_state(handle, 'tg'//c_null_char, 1_c_int, tg, size(tg, kind=c_size_t)) /= 0) stop 7
if(isca_dyn_get_state(handle, 'ug'//c_null_char, 1_c_int, ug, size(ug, kind=c_size_t)) /= 0) stop 8
if(isca_dyn_get_state(handle, 'psg'//c_null_char, 1_c_int, psg, size(psg, kind=c_size_t)) /= 0) stop 9
if(isca_area_weighted_global_mean(handle, psg, mean_ps) /= 0) stop 10
if(isca_trans_grid_to_spherical(handle, tg, spec, int(nlev, c_int), 1_c_int) /= 0) stop 11
write(*,'(a,3es24.16)') 'FORTRAN_STATE Tmin,Tmax,maxabsU=', minval(tg), maxval(tg), maxval(abs(ug))
write(*,'(a,2es24.16)') 'FORTRAN_POINT tg(5,7,20),ug(33,12,3)=', tg(5,7,20), ug(33,12,3)
write(*,'(a,es24.16)')  'FORTRAN_MEAN_PS ', mean_ps
write(*,'(a,2es24.16)') 'FORTRAN_SPEC ts(0,0,25) ', real(spec(0,0,25)), aimag(spec(0,0,25))
! the error convention: a FATAL of the reference arrives as a non-zero return and a message
if(isca_dyn_get_state(handle, 'no_such_field'//c_null_char, 1_c_int, tg, size(tg, kind=c_size_t)) /= 0) then
  write(*,'(a,a)') 'FORTRAN_ERROR ', isca_message()
endif
if(isca_dyn_destroy(handle) /= 0) stop 12
end program drive_held_suarez
