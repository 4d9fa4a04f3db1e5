! isca_dyn_c -- bind(C) view of include/isca_dyn.h for the reference's own language: what a maintainer adds to the reference to put the
! GPU core behind atmosphere_mod / spectral_dynamics_mod / transforms_mod (INTEGRATION.md).  The derived types mirror the C structs member
! for member; isca_config_sizes lets the caller verify that (check_abi below).  Compiled and run by tests/test_gpu_fortran_binding.py.
module isca_dyn_c
use iso_c_binding
implicit none
public

integer, parameter :: ISCA_MAX_LEVELS = 128
integer, parameter :: ISCA_MAX_TRACERS = 8

type, bind(C) :: isca_moist_config
  real(c_double) :: roughness_mom, roughness_heat, roughness_moist
  real(c_double) :: solar_constant, del_sol, del_sw, ir_tau_eq, ir_tau_pole, atm_abs, odp, sw_diff, linear_tau, wv_exponent, solar_exponent
  real(c_double) :: depth, tconst, delta_T, albedo_value
  integer(c_int) :: evaporation
  real(c_double) :: tau_bm, rhbm, Tmin, Tmax, val_inc
  integer(c_int) :: do_rayleigh
  real(c_double) :: trayfric, sponge_pbottom
  integer(c_int) :: damping_conserve_energy
  real(c_double) :: constant_gust
  real(c_double) :: frac_inner, rich_crit_pbl
  real(c_double) :: rich_crit, drag_min
end type

type, bind(C) :: isca_dyn_config
  integer(c_int) :: lon_max, lat_max, num_fourier, num_spherical, num_levels
  integer(c_int) :: fourier_inc
  integer(c_int) :: triang_trunc
  real(c_double) :: dt_atmos
  integer(c_int) :: damping_order
  real(c_double) :: damping_coeff
  real(c_double) :: eddy_sponge_coeff, zmu_sponge_coeff, zmv_sponge_coeff
  real(c_double) :: robert_coeff
  real(c_double) :: raw_filter_coeff
  real(c_double) :: alpha_implicit
  real(c_double) :: reference_sea_level_press
  real(c_double) :: scale_heights, exponent, surf_res
  integer(c_int) :: do_mass_correction, do_energy_correction, do_water_correction
  real(c_double) :: water_correction_limit
  real(c_double) :: initial_temperature
  real(c_double) :: initial_sphum
  real(c_double) :: valid_range_t(2)
  integer(c_int) :: num_tracers
  real(c_double) :: t_zero, t_strat, delh, delv, eps, sigma_b, ka, ks, kf
  integer(c_int) :: do_conserve_energy
  real(c_double) :: trflux, trsink, P00
  integer(c_int) :: rank, world_size
  integer(c_int) :: device
  type(c_ptr)    :: stream
  integer(c_int) :: legendre_impl
  integer(c_int) :: physics
  integer(c_int) :: vert_coord_input
  real(c_double) :: pk_input(ISCA_MAX_LEVELS + 1), bk_input(ISCA_MAX_LEVELS + 1)
  type(isca_moist_config) :: moist
  real(c_double) :: radius, omega
  integer(c_int) :: damping_option, cutoff_wn
  real(c_double) :: damping_coeff_vor, damping_coeff_div
  integer(c_int) :: damping_order_vor, damping_order_div
  integer(c_int) :: tracer_spectral(ISCA_MAX_TRACERS)
  real(c_double) :: tracer_robert_coeff(ISCA_MAX_TRACERS)
  integer(c_int) :: use_virtual_temperature
  integer(c_int) :: vert_advect_uv, vert_advect_t      ! 0 second_centered, 1 fourth_centered, 2 van_leer_linear, 3 finite_volume_parabolic
  integer(c_int) :: use_implicit
  integer(c_int) :: make_symmetric
  integer(c_int) :: vert_difference_option             ! 0 simmons_and_burridge, 1 mcm
  integer(c_int) :: tracer_hole_filling(ISCA_MAX_TRACERS)
  integer(c_int) :: tracer_sms(ISCA_MAX_TRACERS)       ! 1: the entry's own tracer_flux / tracer_sink instead of hs_forcing_nml's trflux / trsink
  real(c_double) :: tracer_flux(ISCA_MAX_TRACERS), tracer_sink(ISCA_MAX_TRACERS)
  integer(c_int) :: tracer_advect_vert(ISCA_MAX_TRACERS)   ! -1 the representation's standard scheme, 0 second_centered .. 3 finite_volume_parabolic
  integer(c_int) :: local_heating_option               ! hs_forcing_nml: 0 '' (none), 1 'Isidoro' (hs_forcing.F90:728-769)
  real(c_double) :: local_heating_srfamp, local_heating_xwidth, local_heating_ywidth, local_heating_xcenter, local_heating_ycenter, local_heating_vert_decay
end type

interface
  integer(c_int) function isca_dyn_config_default(cfg) bind(C)
    import; type(isca_dyn_config), intent(out) :: cfg
  end function
  integer(c_int) function isca_config_sizes(sizes, n) bind(C)
    import; integer(c_size_t), intent(out) :: sizes(*); integer(c_int), value :: n
  end function
  integer(c_int) function isca_dyn_create(cfg, h) bind(C)
    import; type(isca_dyn_config), intent(in) :: cfg; type(c_ptr), intent(out) :: h
  end function
  integer(c_int) function isca_dyn_destroy(h) bind(C)
    import; type(c_ptr), value :: h
  end function
  integer(c_int) function isca_dyn_cold_start(h) bind(C)
    import; type(c_ptr), value :: h
  end function
  integer(c_int) function isca_dyn_step(h, nsteps, sync) bind(C)
    import; type(c_ptr), value :: h; integer(c_int), value :: nsteps, sync
  end function
  ! wait for the steps queued with sync = 0 and check valid_range_t (spectral_dynamics.F90:940-972)
  integer(c_int) function isca_dyn_synchronize(h) bind(C)
    import; type(c_ptr), value :: h
  end function
  ! spectral_dynamics(..., dt_ug, dt_vg, dt_tg, dt_tracers, ...) after a physics package of the caller's own (physics = 2):
  ! (lon, lat, lev) tendency arrays; hs_forcing / tracer_source_sink of hs_forcing_mod on caller fields
  integer(c_int) function isca_dyn_dynamics(h, dt_ug, dt_vg, dt_tg, dt_tracers, on_device, sync) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: dt_ug(*), dt_vg(*), dt_tg(*), dt_tracers(*)
    integer(c_int), value :: on_device, sync
  end function
  integer(c_int) function isca_dyn_delta_t(h, delta_t) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(out) :: delta_t
  end function
  integer(c_int) function isca_hs_forcing(h, dt, p_half, p_full, u, v, t, udt, vdt, tdt) bind(C)
    import; type(c_ptr), value :: h; real(c_double), value :: dt
    real(c_double), intent(in) :: p_half(*), p_full(*), u(*), v(*), t(*); real(c_double), intent(inout) :: udt(*), vdt(*), tdt(*)
  end function
  integer(c_int) function isca_hs_tracer_source_sink(h, surf_p, r, rdt) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: surf_p(*), r(*); real(c_double), intent(inout) :: rdt(*)
  end function
  integer(c_int) function isca_dyn_get_state(h, name, time_level, host, count) bind(C)
    import; type(c_ptr), value :: h; character(kind=c_char), intent(in) :: name(*)
    integer(c_int), value :: time_level; real(c_double), intent(out) :: host(*); integer(c_size_t), value :: count
  end function
  integer(c_int) function isca_dyn_set_state(h, name, time_level, host, count) bind(C)
    import; type(c_ptr), value :: h; character(kind=c_char), intent(in) :: name(*)
    integer(c_int), value :: time_level; real(c_double), intent(in) :: host(*); integer(c_size_t), value :: count
  end function
  integer(c_int) function isca_dyn_get_table(h, name, host, count) bind(C)
    import; type(c_ptr), value :: h; character(kind=c_char), intent(in) :: name(*)
    real(c_double), intent(out) :: host(*); integer(c_size_t), value :: count
  end function
  integer(c_int) function isca_trans_spherical_to_grid(h, spherical, grid, nlev) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(in) :: spherical(*)
    real(c_double), intent(out) :: grid(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_trans_grid_to_spherical(h, grid, spherical, nlev, do_truncation) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: grid(*)
    complex(c_double_complex), intent(out) :: spherical(*); integer(c_int), value :: nlev, do_truncation
  end function
  integer(c_int) function isca_area_weighted_global_mean(h, field2d, mean) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: field2d(*); real(c_double), intent(out) :: mean
  end function
  ! ---- the rest of transforms_mod / spherical_mod on caller arrays (Fortran layouts, num_levels <= nlev)
  integer(c_int) function isca_vor_div_from_uv_grid(h, u, v, vor, div, nlev) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: u(*), v(*)
    complex(c_double_complex), intent(out) :: vor(*), div(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_uv_grid_from_vor_div(h, vor, div, u, v, nlev) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(in) :: vor(*), div(*)
    real(c_double), intent(out) :: u(*), v(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_horizontal_advection(h, field_spec, u, v, tendency, nlev) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(in) :: field_spec(*)
    real(c_double), intent(in) :: u(*), v(*); real(c_double), intent(inout) :: tendency(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_trans_spherical_to_fourier(h, spherical, fourier, nlev) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(in) :: spherical(*)
    complex(c_double_complex), intent(out) :: fourier(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_trans_fourier_to_spherical(h, fourier, spherical, nlev) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(in) :: fourier(*)
    complex(c_double_complex), intent(out) :: spherical(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_trans_grid_to_fourier(h, grid, fourier, nlev) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: grid(*)
    complex(c_double_complex), intent(out) :: fourier(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_trans_fourier_to_grid(h, fourier, grid, nlev) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(in) :: fourier(*)
    real(c_double), intent(out) :: grid(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_trans_filter(h, grid, filter, nlev) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(inout) :: grid(*); type(c_ptr), value :: filter; integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_divide_by_cos(h, grid, nlev, power) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(inout) :: grid(*); integer(c_int), value :: nlev, power
  end function
  integer(c_int) function isca_compute_laplacian(h, spherical, laplacian, nlev, power) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(in) :: spherical(*)
    complex(c_double_complex), intent(out) :: laplacian(*); integer(c_int), value :: nlev, power
  end function
  integer(c_int) function isca_compute_gradient_cos(h, spherical, deriv_lon, deriv_lat, nlev) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(in) :: spherical(*)
    complex(c_double_complex), intent(out) :: deriv_lon(*), deriv_lat(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_compute_ucos_vcos(h, vorticity, divergence, u_cos, v_cos, nlev) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(in) :: vorticity(*), divergence(*)
    complex(c_double_complex), intent(out) :: u_cos(*), v_cos(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_compute_vor_div(h, u_div_cos, v_div_cos, vorticity, divergence, nlev) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(in) :: u_div_cos(*), v_div_cos(*)
    complex(c_double_complex), intent(out) :: vorticity(*), divergence(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_triangular_truncation(h, spherical, nlev) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(inout) :: spherical(*); integer(c_int), value :: nlev
  end function
  integer(c_int) function isca_compute_gaussian(n_hem, sin_hem, wts_hem) bind(C)
    import; integer(c_int), value :: n_hem; real(c_double), intent(out) :: sin_hem(*), wts_hem(*)
  end function
  integer(c_int) function isca_compute_legendre(num_fourier, fourier_inc, num_spherical, sin_lat, n_lat, legendre) bind(C)
    import; integer(c_int), value :: num_fourier, fourier_inc, num_spherical, n_lat
    real(c_double), intent(in) :: sin_lat(*); real(c_double), intent(out) :: legendre(*)
  end function
  ! ---- the routines of the step on caller arrays (press_and_geopot_mod, implicit_mod, spectral_damping_mod, leapfrog_mod,
  !      vert_advection_mod, fv_advection_mod, global_integral_mod)
  integer(c_int) function isca_pressure_variables(h, surf_p, p_half, ln_p_half, p_full, ln_p_full) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: surf_p(*)
    real(c_double), intent(out) :: p_half(*), ln_p_half(*), p_full(*), ln_p_full(*)
  end function
  integer(c_int) function isca_compute_geopotential(h, t, ln_p_half, ln_p_full, geopot_full, geopot_half) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: t(*), ln_p_half(*), ln_p_full(*)
    real(c_double), intent(out) :: geopot_full(*), geopot_half(*)
  end function
  integer(c_int) function isca_compute_geopotential_surf(h, t, ln_p_half, ln_p_full, surf_geopotential, q_grid, geopot_full, geopot_half) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: t(*), ln_p_half(*), ln_p_full(*), surf_geopotential(*)
    type(c_ptr), value :: q_grid                 ! c_null_ptr: no q_grid given
    real(c_double), intent(out) :: geopot_full(*), geopot_half(*)
  end function
  integer(c_int) function isca_compute_pressures_and_heights(h, t, ps, q, z_full, z_half, p_full, p_half) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: t(*), ps(*); type(c_ptr), value :: q
    real(c_double), intent(out) :: z_full(*), z_half(*), p_full(*), p_half(*)
  end function
  integer(c_int) function isca_implicit_correction(h, dt_divs, dt_ts, dt_ln_ps, divs_previous, divs_current, ts_previous, ts_current, &
                                                   ln_ps_previous, ln_ps_current, delta_t) bind(C)
    import; type(c_ptr), value :: h; complex(c_double_complex), intent(inout) :: dt_divs(*), dt_ts(*), dt_ln_ps(*)
    complex(c_double_complex), intent(in) :: divs_previous(*), divs_current(*), ts_previous(*), ts_current(*), ln_ps_previous(*), ln_ps_current(*)
    real(c_double), value :: delta_t
  end function
  integer(c_int) function isca_compute_spectral_damping(h, which, field_previous, dt_field, delta_t) bind(C)
    import; type(c_ptr), value :: h; integer(c_int), value :: which; complex(c_double_complex), intent(in) :: field_previous(*)
    complex(c_double_complex), intent(inout) :: dt_field(*); real(c_double), value :: delta_t
  end function
  integer(c_int) function isca_leapfrog_2level_a(h, n, prev, cur, fut, dt_a, delta_t, robert_coeff, raw_filter_coeff, part) bind(C)
    import; type(c_ptr), value :: h; integer(c_size_t), value :: n; type(c_ptr), value :: prev, cur, fut, dt_a, part
    real(c_double), value :: delta_t, robert_coeff, raw_filter_coeff
  end function
  integer(c_int) function isca_leapfrog_2level_b(h, n, cur, fut, part, robert_coeff, raw_filter_coeff) bind(C)
    import; type(c_ptr), value :: h; integer(c_size_t), value :: n; type(c_ptr), value :: cur, fut, part
    real(c_double), value :: robert_coeff, raw_filter_coeff
  end function
  integer(c_int) function isca_vert_advection_ppm(h, dt, w, surf_p, r, rdt) bind(C)
    import; type(c_ptr), value :: h; real(c_double), value :: dt; real(c_double), intent(in) :: w(*), surf_p(*), r(*)
    real(c_double), intent(out) :: rdt(*)
  end function
  integer(c_int) function isca_vert_advection_centered(h, w, surf_p, r, rdt) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: w(*), surf_p(*), r(*); real(c_double), intent(out) :: rdt(*)
  end function
  integer(c_int) function isca_a_grid_horiz_advection(h, u, v, q, dt, tendency) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: u(*), v(*), q(*); real(c_double), value :: dt
    real(c_double), intent(inout) :: tendency(*)
  end function
  integer(c_int) function isca_mass_weighted_global_integral(h, field, surf_press, integral) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: field(*), surf_press(*); real(c_double), intent(out) :: integral
  end function
  integer(c_int) function isca_dyn_complete_update(h, time_level) bind(C)
    import; type(c_ptr), value :: h; integer(c_int), value :: time_level
  end function
  integer(c_int) function isca_dyn_set_surf_geopotential(h, global_field, count) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: global_field(*); integer(c_size_t), value :: count
  end function
  integer(c_int) function isca_dyn_get_info(h, name, value) bind(C)
    import; type(c_ptr), value :: h; character(kind=c_char), intent(in) :: name(*); integer(c_long), intent(out) :: value
  end function
  ! diag_manager's part for the fields the device accumulates: the run directory's diag_table, one record per output interval appended to <file>.nc
  integer(c_int) function isca_dyn_diag_open(h, diag_table, directory, start_seconds) bind(C)
    import; type(c_ptr), value :: h; character(kind=c_char), intent(in) :: diag_table(*), directory(*); real(c_double), value :: start_seconds
  end function
  integer(c_int) function isca_dyn_diag_close(h) bind(C)
    import; type(c_ptr), value :: h
  end function
  integer(c_int) function isca_dyn_set_info(h, name, value) bind(C)       ! "phys_calls": a running moist model handed over through set_state
    import; type(c_ptr), value :: h; character(kind=c_char), intent(in) :: name(*); integer(c_long), value :: value
  end function
  ! restart files (spectral_dynamics.res.nc, atmosphere.res.nc, mixed_layer.res.nc) written / read by the library's own netCDF-classic code:
  ! spectral_dynamics_end (spectral_dynamics.F90:1502-1531) + atmosphere_end (atmosphere.F90:362-375); read_restart_or_do_coldstart (:509-575)
  integer(c_int) function isca_dyn_write_restart(h, directory, tracer_names) bind(C)
    import; type(c_ptr), value :: h; character(kind=c_char), intent(in) :: directory(*), tracer_names(*)
  end function
  integer(c_int) function isca_dyn_read_restart(h, directory, tracer_names) bind(C)
    import; type(c_ptr), value :: h; character(kind=c_char), intent(in) :: directory(*), tracer_names(*)
  end function
  integer(c_int) function isca_dyn_restart_exists(directory) bind(C)
    import; character(kind=c_char), intent(in) :: directory(*)
  end function
  ! get_topography for topography_option = 'input' (spectral_init_cond.F90:186-245): height and land mask on the global grid -> the handle's surface geopotential
  ! (truncated, or regularised over the ocean: topog_regularization_mod); read_data of a netCDF classic file without netCDF
  integer(c_int) function isca_dyn_set_topography(h, height, land_mask, ocean_topog_smoothing, lambda, fraction_smoothed) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: height(*), land_mask(*); real(c_double), value :: ocean_topog_smoothing
    real(c_double), intent(out) :: lambda, fraction_smoothed
  end function
  integer(c_int) function isca_nc_read_variable(path, var_name, record, out, count, count_out) bind(C)
    import; character(kind=c_char), intent(in) :: path(*), var_name(*); integer(c_int), value :: record
    real(c_double), intent(out) :: out(*); integer(c_size_t), value :: count; integer(c_size_t), intent(out) :: count_out
  end function
  ! a host without MPI of its own: rank / ranks / rank on the node from the environment, the communicator's id through ISCA_COMM_ID_FILE
  integer(c_int) function isca_env_rank(rank, world_size, local_rank) bind(C)
    import; integer(c_int), intent(out) :: rank, world_size, local_rank
  end function
  integer(c_int) function isca_dyn_comm_init_env(h) bind(C)
    import; type(c_ptr), value :: h
  end function
  function isca_last_error() bind(C) result(msg)
    import; type(c_ptr) :: msg
  end function
end interface

contains

! the FATAL message of the library as a Fortran string (for error_mesg(routine, message, FATAL))
function isca_message() result(text)
  character(len=:), allocatable :: text
  character(kind=c_char), pointer :: p(:)
  type(c_ptr) :: cp
  integer :: n
  cp = isca_last_error()
  text = ''
  if(.not. c_associated(cp)) return
  call c_f_pointer(cp, p, (/4096/))
  n = 0
  do while(n < 4096)
    if(p(n+1) == c_null_char) exit
    n = n + 1
  enddo
  allocate(character(len=n) :: text)
  text = transfer(p(1:n), text)
end function

! .true. when this module's derived types have the size of the library's structs
logical function check_abi()
  integer(c_size_t) :: sizes(4)
  type(isca_dyn_config) :: cfg
  check_abi = .false.
  if(isca_config_sizes(sizes, 4_c_int) /= 0) return
  check_abi = (sizes(1) == c_sizeof(cfg)) .and. (sizes(2) == c_sizeof(cfg%moist))
end function

end module isca_dyn_c
