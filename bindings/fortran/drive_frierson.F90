! The moist model from Fortran: atmosphere_nml idealized_moist_model = .true. with the Frierson test case's namelist values
! (frierson_test_case.py:49-170) set through the bind(C) types (nested isca_moist_config, vert_coordinate_nml's bk as an array), 144 steps
! from the cold start.  tests/test_gpu_fortran_binding.py compares the printed values with the reference run (tests/golden/moist_run_T21L25.npz).
program drive_frierson
use iso_c_binding
use isca_dyn_c
implicit none
type(isca_dyn_config) :: cfg
type(c_ptr) :: handle
integer, parameter :: nlon = 64, nlat = 32, nlev = 25
real(c_double), allocatable :: tg(:,:,:), q(:,:,:), ts(:,:)
real(c_double), parameter :: bk(26) = (/ 0.000000d0, 0.0117665d0, 0.0196679d0, 0.0315244d0, 0.0485411d0, 0.0719344d0, 0.1027829d0, &
     0.1418581d0, 0.1894648d0, 0.2453219d0, 0.3085103d0, 0.3775033d0, 0.4502789d0, 0.5244989d0, 0.5977253d0, 0.6676441d0, 0.7322627d0, &
     0.7900587d0, 0.8400683d0, 0.8819111d0, 0.9157609d0, 0.9422770d0, 0.9625127d0, 0.9778177d0, 0.9897489d0, 1.0000000d0 /)

if(.not. check_abi()) stop 2
if(isca_dyn_config_default(cfg) /= 0) stop 3
cfg%lon_max = nlon; cfg%lat_max = nlat; cfg%num_fourier = 21; cfg%num_spherical = 22; cfg%num_levels = nlev
cfg%dt_atmos = 720.0d0; cfg%damping_order = 4; cfg%robert_coeff = 0.03d0; cfg%initial_sphum = 2.d-6
cfg%physics = 1                                   ! idealized_moist_model = .true.
cfg%vert_coord_input = 1; cfg%bk_input = 0.0d0; cfg%pk_input = 0.0d0; cfg%bk_input(1:26) = bk
cfg%moist%atm_abs = 0.2d0; cfg%moist%depth = 2.5d0; cfg%moist%albedo_value = 0.31d0; cfg%moist%rhbm = 0.7d0
cfg%moist%Tmin = 160.d0; cfg%moist%Tmax = 350.d0; cfg%moist%trayfric = -0.25d0; cfg%moist%sponge_pbottom = 5000.d0
if(isca_dyn_create(cfg, handle) /= 0) then
  print *, 'FATAL: ', isca_message(); stop 4
endif
if(isca_dyn_cold_start(handle) /= 0) stop 5
if(isca_dyn_step(handle, 144_c_int, 1_c_int) /= 0) then
  print *, 'FATAL: ', isca_message(); stop 6
endif
allocate(tg(nlon, nlat, nlev), q(nlon, nlat, nlev), ts(nlon, nlat))
if(isca_dyn_get_state(handle, 'tg'//c_null_char, 1_c_int, tg, size(tg, kind=c_size_t)) /= 0) stop 7
if(isca_dyn_get_state(handle, 'tr'//c_null_char, 1_c_int, q, size(q, kind=c_size_t)) /= 0) stop 8
if(isca_dyn_get_state(handle, 't_surf'//c_null_char, 1_c_int, ts, size(ts, kind=c_size_t)) /= 0) stop 9
write(*,'(a,4es24.16)') 'FORTRAN_MOIST Tmin,Tmax,qmax,q(10,16,25)=', minval(tg), maxval(tg), maxval(q), q(10,16,25)
write(*,'(a,2es24.16)') 'FORTRAN_TSURF min,max=', minval(ts), maxval(ts)
if(isca_dyn_destroy(handle) /= 0) stop 10
end program drive_frierson
