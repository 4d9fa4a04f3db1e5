! isca_siblings_c -- bind(C) view of include/isca_shallow.h, isca_barotropic.h and isca_stirring.h: the sibling cores behind the
! reference's src/atmos_spectral_shallow and src/atmos_spectral_barotropic atmosphere_mod (INTEGRATION.md, "Sibling cores").
module isca_siblings_c
use iso_c_binding
implicit none
public

type, bind(C) :: isca_stirring_config
  real(c_double) :: decay_time, amplitude, lat0, lon0, widthy, widthx, B
  integer(c_int) :: do_localize, n_total_forcing_max, n_total_forcing_min, zonal_forcing_min
  integer(c_long_long) :: seed
end type

type, bind(C) :: isca_shallow_config
  integer(c_int) :: num_lon, num_lat, num_fourier, num_spherical
  real(c_double) :: dt_atmos
  integer(c_int) :: damping_order
  real(c_double) :: damping_coeff, robert_coeff, robert_coeff_tracer
  real(c_double) :: h_0, u_deep_mag, n_merid_deep_flow, u_upper_mag_init
  integer(c_int) :: spec_tracer, grid_tracer
  real(c_double) :: lon_centre_init_cyc, lat_centre_init_cyc, lon_centre_init_acyc, lat_centre_init_acyc
  real(c_double) :: init_vortex_radius_deg, init_vortex_vor_f, init_vortex_h_h_0
  integer(c_int) :: add_initial_vortex_pair, add_initial_vortex_as_height
  real(c_double) :: valid_range_v(2)
  real(c_double) :: fric_damp_time, therm_damp_time, phys_h_0, h_amp, h_lon, h_lat, h_width, h_itcz, itcz_width
  integer(c_int) :: device
  type(isca_stirring_config) :: stirring
  real(c_double) :: radius, omega
end type

type, bind(C) :: isca_barotropic_config
  integer(c_int) :: num_lon, num_lat, num_fourier, num_spherical
  real(c_double) :: dt_atmos
  integer(c_int) :: damping_order
  real(c_double) :: damping_coeff, damping_coeff_r, robert_coeff
  real(c_double) :: zeta_0
  integer(c_int) :: m_0
  real(c_double) :: eddy_width, eddy_lat
  integer(c_int) :: spec_tracer, grid_tracer
  real(c_double) :: valid_range_v(2)
  integer(c_int) :: initial_zonal_wind
  integer(c_int) :: device
  type(isca_stirring_config) :: stirring
  real(c_double) :: radius, omega
end type

interface
  integer(c_int) function isca_shallow_config_default(cfg) bind(C)
    import; type(isca_shallow_config), intent(out) :: cfg
  end function
  integer(c_int) function isca_shallow_create(cfg, h) bind(C)
    import; type(isca_shallow_config), intent(in) :: cfg; type(c_ptr), intent(out) :: h
  end function
  integer(c_int) function isca_shallow_destroy(h) bind(C)
    import; type(c_ptr), value :: h
  end function
  integer(c_int) function isca_shallow_cold_start(h) bind(C)
    import; type(c_ptr), value :: h
  end function
  integer(c_int) function isca_shallow_step(h, nsteps) bind(C)
    import; type(c_ptr), value :: h; integer(c_int), value :: nsteps
  end function
  integer(c_int) function isca_shallow_get_state(h, name, time_level, host, count) bind(C)
    import; type(c_ptr), value :: h; character(kind=c_char), intent(in) :: name(*)
    integer(c_int), value :: time_level; real(c_double), intent(out) :: host(*); integer(c_size_t), value :: count
  end function
  integer(c_int) function isca_shallow_set_stirring_noise(h, ran, count) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: ran(*); integer(c_size_t), value :: count
  end function
  integer(c_int) function isca_barotropic_config_default(cfg) bind(C)
    import; type(isca_barotropic_config), intent(out) :: cfg
  end function
  integer(c_int) function isca_barotropic_create(cfg, h) bind(C)
    import; type(isca_barotropic_config), intent(in) :: cfg; type(c_ptr), intent(out) :: h
  end function
  integer(c_int) function isca_barotropic_destroy(h) bind(C)
    import; type(c_ptr), value :: h
  end function
  integer(c_int) function isca_barotropic_cold_start(h) bind(C)
    import; type(c_ptr), value :: h
  end function
  integer(c_int) function isca_barotropic_step(h, nsteps) bind(C)
    import; type(c_ptr), value :: h; integer(c_int), value :: nsteps
  end function
  integer(c_int) function isca_barotropic_get_state(h, name, time_level, host, count) bind(C)
    import; type(c_ptr), value :: h; character(kind=c_char), intent(in) :: name(*)
    integer(c_int), value :: time_level; real(c_double), intent(out) :: host(*); integer(c_size_t), value :: count
  end function
  integer(c_int) function isca_barotropic_set_stirring_noise(h, ran, count) bind(C)
    import; type(c_ptr), value :: h; real(c_double), intent(in) :: ran(*); integer(c_size_t), value :: count
  end function
end interface

end module isca_siblings_c
