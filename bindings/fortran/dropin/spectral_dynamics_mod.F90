! spectral_dynamics_mod -- the reference's module name and public procedures (atmos_spectral/model/spectral_dynamics.F90:95-98) in front
! of the MI355X core: spectral_dynamics_init reads the reference's namelists and field_table and creates the device core
! (isca_dyn_create, physics = 2: the host keeps its physics package), spectral_dynamics is isca_dyn_dynamics followed by the copies
! the reference returns to its caller (spectral_dynamics.F90:780-795, :1023-1028).
!
! With isca_dropin_mod%dropin_physics = 1 (atmosphere_mod, atmosphere_nml: idealized_moist_model) the core is created with the Frierson
! column chain and the namelist values idealized_moist_phys_init collected.  vert_coord_option: 'even_sigma', 'uneven_sigma' and 'input'
! (vert_coordinate_nml's pk, bk).
! vert_coord_option 'hybrid' / 'mcm' / 'v197' are formed here (named_vert_coord); topography_option 'gaussian' through the reference's own
! gaussian_topog_mod.  Restart files: INPUT/*.res.nc are read and RESTART/*.res.nc written by the library's own netCDF-classic code
! (isca_dyn_read_restart / isca_dyn_write_restart; the same files isca_amd/restart.py handles).  Not here:
! topography_option = 'interpolated' (the data files of topography_mod), initial_state_option other than 'quiescent'.  Each is refused with
! error_mesg(..., FATAL) naming the option.
module spectral_dynamics_mod

#ifdef INTERNAL_FILE_NML
use mpp_mod, only: input_nml_file
#else
use fms_mod, only: open_namelist_file
#endif
use iso_c_binding
use fms_mod,            only: error_mesg, FATAL, NOTE, check_nml_error, mpp_pe, mpp_root_pe, stdlog, lowercase, uppercase, close_file, file_exist
use constants_mod,      only: radius, omega, grav, pi
use gaussian_topog_mod, only: gaussian_topog_init
use time_manager_mod,   only: time_type, get_time
use field_manager_mod,  only: MODEL_ATMOS, parse
use tracer_manager_mod, only: get_number_tracers, get_tracer_names, query_method, get_tracer_index, NO_TRACER
use tracer_type_mod,    only: tracer_type
use isca_dyn_c
use isca_dropin_mod

implicit none
private

public :: spectral_dynamics_init, spectral_dynamics, spectral_dynamics_end, get_num_levels
public :: get_use_virtual_temperature, get_reference_sea_level_press, get_surf_geopotential
public :: get_pk_bk, complete_robert_filter, complete_update_of_future
public :: get_axis_id, spectral_diagnostics, get_initial_fields

character(len=32), parameter :: default_advect_vert = 'second_centered', default_representation = 'spectral', default_hole_filling = 'off'

! ---- spectral_dynamics_nml: the reference's variables and defaults (spectral_dynamics.F90:152-224)
logical :: do_mass_correction = .true., do_water_correction = .true., do_energy_correction = .true., use_virtual_temperature = .false., &
           use_implicit = .true., triang_trunc = .true., graceful_shutdown = .false., make_symmetric = .false.
integer :: damping_order = 2, damping_order_vor = -1, damping_order_div = -1, cutoff_wn = 15, lon_max = 128, lat_max = 64, &
           num_fourier = 42, num_spherical = 43, fourier_inc = 1, num_levels = 18, num_steps = 1
integer, dimension(2) :: print_interval = (/1, 0/)
character(len=64) :: vert_coord_option = 'even_sigma', damping_option = 'resolution_dependent', vert_advect_uv = default_advect_vert, &
                     vert_advect_t = default_advect_vert, vert_difference_option = 'simmons_and_burridge', initial_state_option = 'quiescent'
real :: damping_coeff = 1.15740741e-4, damping_coeff_vor = -1., damping_coeff_div = -1., eddy_sponge_coeff = 0., zmu_sponge_coeff = 0., &
        zmv_sponge_coeff = 0., robert_coeff = .04, alpha_implicit = .5, longitude_origin = 0., scale_heights = 4., surf_res = .1, &
        p_press = .1, p_sigma = .3, exponent = 2.5, ocean_topog_smoothing = .93, initial_sphum = 0.0, reference_sea_level_press = 101325., &
        water_correction_limit = 0.0, raw_filter_coeff = 1.0
logical :: json_logging = .false.
real, dimension(2) :: valid_range_t = (/100., 500./)
namelist /spectral_dynamics_nml/ use_virtual_temperature, damping_option, cutoff_wn, damping_order, damping_coeff, damping_order_vor,  &
                                 damping_coeff_vor, damping_order_div, damping_coeff_div, do_mass_correction, do_water_correction,     &
                                 do_energy_correction, vert_advect_uv, vert_advect_t, use_implicit, longitude_origin, robert_coeff,     &
                                 alpha_implicit, vert_difference_option, reference_sea_level_press, lon_max, lat_max, num_levels,       &
                                 num_fourier, num_spherical, fourier_inc, triang_trunc, vert_coord_option, scale_heights, surf_res,     &
                                 p_press, p_sigma, exponent, ocean_topog_smoothing, initial_sphum, valid_range_t, eddy_sponge_coeff,    &
                                 zmu_sponge_coeff, zmv_sponge_coeff, print_interval, num_steps, initial_state_option,                  &
                                 water_correction_limit, raw_filter_coeff, graceful_shutdown, json_logging, make_symmetric

! ---- spectral_init_cond_nml (init/spectral_init_cond.F90:68-74)
real :: initial_temperature = 264.
character(len=64) :: topography_option = 'flat', topog_file_name = 'topography.data.nc', topog_field_name = 'zsurf'
character(len=256) :: land_field_name = 'land_mask'
namelist /spectral_init_cond_nml/ initial_temperature, topography_option, topog_file_name, topog_field_name, land_field_name

! ---- hs_forcing_nml (atmos_param/hs_forcing/hs_forcing.F90:76-122): the library takes the Held-Suarez parameters when the core is
! created, so they are read here, ahead of hs_forcing_init
logical :: no_forcing = .false., do_conserve_energy = .true., relax_to_specified_wind = .false.
real :: t_zero = 315., t_strat = 200., delh = 60., delv = 10., eps = 0., sigma_b = 0.7, P00 = 1.e5, p_trop = 1.e4, alpha = 2./7, ka = -40., &
        ks = -4., kf = -1., trflux = 1.e-5, trsink = -4., local_heating_srfamp = 0.0, local_heating_xwidth = 10., local_heating_ywidth = 10., &
        local_heating_xcenter = 180., local_heating_ycenter = 45., local_heating_vert_decay = 1.e4, peri_time = 0.25, smaxis = 1.5e6, &
        albedo = 0.3, lapse = 6.5, h_a = 2, tau_s = 5, heat_capacity = 4.2e6, ml_depth = 1, spinup_time = 10800., orbital_period = 365.25
character(len=256) :: local_heating_option = '', local_heating_file = '', u_wind_file = 'u', v_wind_file = 'v', &
                      equilibrium_t_option = 'Held_Suarez', equilibrium_t_file = 'temp', stratosphere_t_option = 'extend_tp'
namelist /hs_forcing_nml/ no_forcing, t_zero, t_strat, delh, delv, eps, sigma_b, ka, ks, kf, do_conserve_energy, trflux, trsink,        &
                          local_heating_srfamp, local_heating_xwidth, local_heating_ywidth, local_heating_xcenter, local_heating_ycenter, &
                          local_heating_vert_decay, local_heating_option, local_heating_file, relax_to_specified_wind, u_wind_file,     &
                          v_wind_file, equilibrium_t_option, equilibrium_t_file, p_trop, alpha, peri_time, smaxis, albedo, lapse, h_a,  &
                          tau_s, orbital_period, heat_capacity, ml_depth, spinup_time, stratosphere_t_option, P00

! ---- vert_coordinate_nml (init/vert_coordinate.F90:80-83), read when vert_coord_option = 'input'
integer, parameter :: max_levels = 100
real, dimension(max_levels+1) :: pk = 0., bk = 0.
namelist /vert_coordinate_nml/ pk, bk

logical :: module_is_initialized = .false.
logical :: dry_model
integer :: nhum, num_tracers
character(len=256) :: tracer_name_list = ' '       ! the field_table names of the prognostic tracers, comma-separated: the restart files' variable names
real :: dt_real

contains

!===============================================================================================
subroutine read_nml(which)
  character(len=*), intent(in) :: which
  integer :: io, ierr, unit
#ifdef INTERNAL_FILE_NML
  if(which == 'dyn') read(input_nml_file, nml=spectral_dynamics_nml, iostat=io)
  if(which == 'ini') read(input_nml_file, nml=spectral_init_cond_nml, iostat=io)
  if(which == 'hs')  read(input_nml_file, nml=hs_forcing_nml, iostat=io)
  if(which == 'vc')  read(input_nml_file, nml=vert_coordinate_nml, iostat=io)
  if(which == 'vc')  ierr = check_nml_error(io, 'vert_coordinate_nml')
  if(which == 'dyn') ierr = check_nml_error(io, 'spectral_dynamics_nml')
  if(which == 'ini') ierr = check_nml_error(io, 'spectral_init_cond_nml')
  if(which == 'hs')  ierr = check_nml_error(io, 'hs_forcing_nml')
#else
  unit = open_namelist_file()
  ierr = 1
  do while(ierr /= 0)
    if(which == 'dyn') read(unit, nml=spectral_dynamics_nml, iostat=io, end=20)
    if(which == 'ini') read(unit, nml=spectral_init_cond_nml, iostat=io, end=20)
    if(which == 'hs')  read(unit, nml=hs_forcing_nml, iostat=io, end=20)
    if(which == 'vc')  read(unit, nml=vert_coordinate_nml, iostat=io, end=20)
    ierr = check_nml_error(io, which)
  enddo
20 call close_file(unit)
#endif
end subroutine read_nml

!===============================================================================================
subroutine spectral_dynamics_init(Time, Time_step_in, tracer_attributes, dry_model_out, nhum_out, ocean_mask)

type(time_type), intent(in) :: Time, Time_step_in
type(tracer_type), intent(inout), dimension(:) :: tracer_attributes
logical, intent(out) :: dry_model_out
integer, intent(out) :: nhum_out
logical, optional, intent(in), dimension(:,:) :: ocean_mask

type(isca_dyn_config) :: cfg
integer :: ntr, nsphum, nmix_rat, seconds, days, k
integer(c_int) :: env_rank, env_world, env_local
integer(c_long) :: info_val
real :: robert_coeff_tracers, sms_value
character(len=128) :: scheme, params
character(len=128) :: tname, longname, units

if(module_is_initialized) return

call read_nml('dyn'); call read_nml('ini'); call read_nml('hs')
if(mpp_pe() == mpp_root_pe()) write(stdlog(), nml=spectral_dynamics_nml)

if(damping_order_vor == -1 ) damping_order_vor = damping_order
if(damping_order_div == -1 ) damping_order_div = damping_order
if(damping_coeff_vor == -1.) damping_coeff_vor = damping_coeff
if(damping_coeff_div == -1.) damping_coeff_div = damping_coeff

! what the device core does not carry is refused by name (check_dynamics_nml's own tests are the library's: isca_dyn_create)
if(trim(vert_difference_option) /= 'simmons_and_burridge' .and. trim(vert_difference_option) /= 'mcm') &        ! press_and_geopot.F90:216-219
  call error_mesg('pressure_variables','"'//trim(vert_difference_option)//'" is not a valid value for vert_difference_option', FATAL)
if(trim(initial_state_option) /= 'quiescent') &
  call error_mesg('spectral_dynamics_init','"'//trim(initial_state_option)//'" is not a supported value for initial_state_option.', FATAL)
if(trim(topography_option) /= 'flat' .and. trim(topography_option) /= 'gaussian' .and. trim(topography_option) /= 'input') &
  call error_mesg('get_topography','"'//trim(topography_option)//'" is not a supported value for topography_option here (flat, gaussian, input; '// &
                  "'interpolated' needs the data files of topography_mod)", FATAL)
if(num_steps /= 1) call error_mesg('spectral_dynamics_init','num_steps must be 1.', FATAL)
if(longitude_origin /= 0.) call error_mesg('spectral_dynamics_init','longitude_origin must be 0.', FATAL)
if(dropin_physics /= 1 .and. (trim(equilibrium_t_option) /= 'Held_Suarez' .or. relax_to_specified_wind)) &
  call error_mesg('spectral_dynamics_init','hs_forcing_nml: only the Held-Suarez branch of hs_forcing (with local_heating_option = Isidoro, or no_forcing) '// &
                  'is carried by the device core.', FATAL)
if(dropin_physics /= 1 .and. trim(local_heating_option) /= '' .and. trim(local_heating_option) /= 'Isidoro') &
  call error_mesg('hs_forcing_nml','"'//trim(local_heating_option)//'"  is not a supported value for local_heating_option', FATAL)

call chk(isca_dyn_config_default(cfg), 'spectral_dynamics_init')
cfg%lon_max = lon_max; cfg%lat_max = lat_max; cfg%num_fourier = num_fourier; cfg%num_spherical = num_spherical
cfg%num_levels = num_levels; cfg%fourier_inc = fourier_inc; cfg%triang_trunc = merge(1, 0, triang_trunc)
call get_time(Time_step_in, seconds, days)
dt_real = 86400*days + seconds
cfg%dt_atmos = dt_real
if(dropin_physics /= 1 .and. trim(local_heating_option) == 'Isidoro') then       ! hs_forcing's analytic heat source (hs_forcing.F90:728-769)
  cfg%local_heating_option = 1; cfg%local_heating_srfamp = local_heating_srfamp; cfg%local_heating_vert_decay = local_heating_vert_decay
  cfg%local_heating_xwidth = local_heating_xwidth; cfg%local_heating_ywidth = local_heating_ywidth
  cfg%local_heating_xcenter = local_heating_xcenter; cfg%local_heating_ycenter = local_heating_ycenter
endif
cfg%damping_order = damping_order; cfg%damping_coeff = damping_coeff
cfg%damping_order_vor = damping_order_vor; cfg%damping_order_div = damping_order_div
cfg%damping_coeff_vor = damping_coeff_vor; cfg%damping_coeff_div = damping_coeff_div; cfg%cutoff_wn = cutoff_wn
select case(trim(damping_option))
  case('resolution_dependent');   cfg%damping_option = 0
  case('exponential_cutoff');     cfg%damping_option = 1
  case('resolution_independent'); cfg%damping_option = 2
  case default
    call error_mesg('spectral_damping_init','"'//trim(damping_option)//'" is not a valid value for damping_option.', FATAL)
end select
cfg%eddy_sponge_coeff = eddy_sponge_coeff; cfg%zmu_sponge_coeff = zmu_sponge_coeff; cfg%zmv_sponge_coeff = zmv_sponge_coeff
cfg%robert_coeff = robert_coeff; cfg%raw_filter_coeff = raw_filter_coeff; cfg%alpha_implicit = alpha_implicit
cfg%reference_sea_level_press = reference_sea_level_press
cfg%do_mass_correction = merge(1, 0, do_mass_correction); cfg%do_energy_correction = merge(1, 0, do_energy_correction)
cfg%do_water_correction = merge(1, 0, do_water_correction); cfg%water_correction_limit = water_correction_limit
cfg%initial_temperature = initial_temperature; cfg%initial_sphum = initial_sphum; cfg%valid_range_t = valid_range_t
cfg%use_virtual_temperature = merge(1, 0, use_virtual_temperature)
cfg%radius = radius; cfg%omega = omega
cfg%t_zero = t_zero; cfg%t_strat = t_strat; cfg%delh = delh; cfg%delv = delv; cfg%eps = eps; cfg%sigma_b = sigma_b
cfg%ka = ka; cfg%ks = ks; cfg%kf = kf; cfg%do_conserve_energy = merge(1, 0, do_conserve_energy)
cfg%trflux = trflux; cfg%trsink = trsink; cfg%P00 = P00
if(no_forcing) then      ! hs_forcing returns at once (hs_forcing.F90:174): zero coefficients give exactly zero tendencies, no tracer source or sink
  cfg%ka = 0.; cfg%ks = 0.; cfg%kf = 0.; cfg%trflux = 0.; cfg%trsink = 0.; cfg%local_heating_option = 0
endif
cfg%vert_advect_uv = advect_scheme(vert_advect_uv, 'vert_advect_uv'); cfg%vert_advect_t = advect_scheme(vert_advect_t, 'vert_advect_t')
cfg%use_implicit = merge(1, 0, use_implicit); cfg%make_symmetric = merge(1, 0, make_symmetric)
cfg%vert_difference_option = merge(1, 0, trim(vert_difference_option) == 'mcm')
cfg%physics = dropin_physics                ! 2: the caller keeps its physics package and spectral_dynamics receives its tendencies
select case(trim(vert_coord_option))        ! compute_vert_coord (init/vert_coordinate.F90:124-152)
  case('uneven_sigma')
    cfg%vert_coord_input = 0; cfg%scale_heights = scale_heights; cfg%exponent = exponent; cfg%surf_res = surf_res
  case('even_sigma')
    cfg%vert_coord_input = 1
    do k = 0, num_levels
      cfg%pk_input(k+1) = 0.; cfg%bk_input(k+1) = real(k)/real(num_levels)
    enddo
  case('input')                             ! read_namelist (init/vert_coordinate.F90:187-216): the half levels of vert_coordinate_nml
    call read_nml('vc')
    if(num_levels + 1 > size(cfg%pk_input)) call error_mesg('spectral_dynamics_init','more levels than the device core carries', FATAL)
    if(all(bk(1:num_levels+1) == 0.) .and. all(pk(1:num_levels+1) == 0.)) call error_mesg('read_namelist in vert_coordinate_mod', &
      'No levels specified in namelist vert_coordinate_nml or namelist is missing ', FATAL)
    cfg%vert_coord_input = 1
    cfg%pk_input(1:num_levels+1) = pk(1:num_levels+1); cfg%bk_input(1:num_levels+1) = bk(1:num_levels+1)
  case('hybrid', 'mcm', 'v197')             ! compute_vert_coord's other options (init/vert_coordinate.F90:124-152), formed here
    call named_vert_coord(cfg)
  case default
    call error_mesg('compute_vert_coord','"'//trim(vert_coord_option)//'" is not a valid value for vert_coord_option.', FATAL)
end select
if(dropin_physics == 1) then                ! the Frierson chain inside the device step: idealized_moist_phys_init's namelist values
  if(.not. dropin_moist_set) call error_mesg('spectral_dynamics_init','physics = 1 without idealized_moist_phys_init', FATAL)
  cfg%moist = dropin_moist
endif

! ---- the field_table, as the reference reads it (spectral_dynamics.F90:316-409)
call get_number_tracers(MODEL_ATMOS, num_prog=num_tracers)
if(num_tracers > ISCA_MAX_TRACERS) call error_mesg('spectral_dynamics_init','more prognostic tracers than the device core carries', FATAL)
if(size(tracer_attributes) < num_tracers) call error_mesg('spectral_dynamics_init','size(tracer_attributes) is too small', FATAL)
cfg%num_tracers = num_tracers
do ntr = 1, num_tracers
  call get_tracer_names(MODEL_ATMOS, ntr, tname, longname, units)
  tracer_attributes(ntr)%name = lowercase(tname)
  tracer_attributes(ntr)%numerical_representation = default_representation
  if(query_method('numerical_representation', MODEL_ATMOS, ntr, scheme)) tracer_attributes(ntr)%numerical_representation = scheme
  tracer_attributes(ntr)%advect_vert = default_advect_vert
  if(query_method('advect_vert', MODEL_ATMOS, ntr, scheme)) tracer_attributes(ntr)%advect_vert = scheme
  tracer_attributes(ntr)%hole_filling = default_hole_filling
  if(query_method('hole_filling', MODEL_ATMOS, ntr, scheme)) tracer_attributes(ntr)%hole_filling = scheme
  tracer_attributes(ntr)%robert_coeff = robert_coeff
  if(query_method('robert_filter', MODEL_ATMOS, ntr, scheme, params)) then
    if(uppercase(scheme) == 'OFF') then
      tracer_attributes(ntr)%robert_coeff = 0.0
    else if(parse(params, 'robert_coeff', robert_coeff_tracers) == 1) then
      tracer_attributes(ntr)%robert_coeff = robert_coeff_tracers
    endif
  endif
  select case(trim(tracer_attributes(ntr)%numerical_representation))
    case('spectral')
      if(ntr == 1) call error_mesg('spectral_dynamics_init', "the first tracer of the field_table is carried as a 'grid' tracer by the device core: "// &
                                   "numerical_representation 'spectral' is not a supported value for it", FATAL)
      tracer_attributes(ntr)%advect_horiz = 'spectral'; cfg%tracer_spectral(ntr) = 1
      if(lowercase(trim(tracer_attributes(ntr)%hole_filling)) == 'on') cfg%tracer_hole_filling(ntr) = 1      ! water_borrowing (spectral_dynamics.F90:1142)
      if(uppercase(trim(tracer_attributes(ntr)%advect_vert)) /= 'SECOND_CENTERED') cfg%tracer_advect_vert(ntr) = advect_scheme(tracer_attributes(ntr)%advect_vert, 'advect_vert')
    case('grid')
      tracer_attributes(ntr)%advect_horiz = 'van_leer'; cfg%tracer_spectral(ntr) = 0
      if(uppercase(trim(tracer_attributes(ntr)%advect_vert)) /= 'FINITE_VOLUME_PARABOLIC') cfg%tracer_advect_vert(ntr) = advect_scheme(tracer_attributes(ntr)%advect_vert, 'advect_vert')
    case default
      call error_mesg('spectral_dynamics_init', trim(tracer_attributes(ntr)%numerical_representation)//' is an invalid numerical_representation', FATAL)
  end select
  cfg%tracer_robert_coeff(ntr) = tracer_attributes(ntr)%robert_coeff
  if(query_method('tracer_sms', MODEL_ATMOS, ntr, scheme, params) .and. .not. no_forcing) then       ! hs_forcing's source and sink of this entry (hs_forcing.F90:251-261)
    cfg%tracer_sms(ntr) = 1; cfg%tracer_flux(ntr) = 0.; cfg%tracer_sink(ntr) = 0.      ! 'none' (no tendency) and 'off' (flux = sink = 0) come to the same
    if(uppercase(trim(scheme)) /= 'NONE' .and. uppercase(trim(scheme)) /= 'OFF') then
      cfg%tracer_flux(ntr) = trflux; cfg%tracer_sink(ntr) = trsink
      if(parse(params, 'flux', sms_value) == 1) cfg%tracer_flux(ntr) = sms_value
      if(parse(params, 'sink', sms_value) == 1) cfg%tracer_sink(ntr) = sms_value
    endif
  endif
enddo
nsphum   = get_tracer_index(MODEL_ATMOS, 'sphum')
nmix_rat = get_tracer_index(MODEL_ATMOS, 'mix_rat')
if(nsphum /= NO_TRACER .and. nmix_rat /= NO_TRACER) &
  call error_mesg('spectral_dynamics_init','sphum and mix_rat cannot both be specified as tracers at the same time', FATAL)
nhum = 0
if(nsphum /= NO_TRACER) nhum = nsphum
if(nmix_rat /= NO_TRACER) nhum = nmix_rat
dry_model = (nhum == 0)
if(.not. dry_model .and. nhum /= 1) &       ! the library's tracer 1 is the one the water correction and the virtual temperature act on
  call error_mesg('spectral_dynamics_init','the humidity tracer must be the first atmos_mod entry of the field_table', FATAL)
if(dry_model .and. num_tracers > 0) then
  if(do_water_correction) call error_mesg('compute_corrections','do_water_correction must be .false. in a dry model (default is .true.)', FATAL)
  cfg%initial_sphum = 0.0; cfg%use_virtual_temperature = 0
endif
dry_model_out = dry_model
nhum_out = nhum

! ---- the device core: read_restart_or_do_coldstart (spectral_dynamics.F90:509-630) -- INPUT/spectral_dynamics.res.nc (+ atmosphere.res.nc,
!      mixed_layer.res.nc), read by the library's own netCDF-classic reader, or the cold start
! the decomposition: rank / number of ranks from the environment (this mpp has no MPI: mpp_pe() is 0 everywhere), latitude bands to the ranks, the
! communicator's id through ISCA_COMM_ID_FILE; collective over the ranks (isca_dyn_comm_init_env is a no-op with one)
call chk(isca_env_rank(env_rank, env_world, env_local), 'spectral_dynamics_init')
cfg%rank = env_rank; cfg%world_size = env_world; cfg%device = env_local
my_rank = env_rank; num_ranks = env_world
call chk(isca_dyn_create(cfg, core), 'spectral_dynamics_init')
core_ready = .true.
graceful = graceful_shutdown
call chk(isca_dyn_comm_init_env(core), 'spectral_dynamics_init')
call chk(isca_dyn_get_info(core, 'lat_local'//c_null_char, info_val), 'spectral_dynamics_init'); je_loc = int(info_val)
call chk(isca_dyn_get_info(core, 'lat_start'//c_null_char, info_val), 'spectral_dynamics_init'); js_loc = int(info_val) + 1
je_loc = js_loc + je_loc - 1
tracer_name_list = ' '
do ntr = 1, num_tracers
  tracer_name_list = trim(tracer_name_list)//trim(tracer_attributes(ntr)%name)
  if(ntr < num_tracers) tracer_name_list = trim(tracer_name_list)//','
enddo
if(isca_dyn_restart_exists('INPUT'//c_null_char) /= 0) then
  call chk(isca_dyn_read_restart(core, 'INPUT'//c_null_char, trim(tracer_name_list)//c_null_char), 'spectral_dynamics_init')
else
  if(trim(topography_option) == 'gaussian') call gaussian_topography      ! get_topography (init/spectral_init_cond.F90:299-303)
  if(trim(topography_option) == 'input') call input_topography            ! (:186-245)
  call chk(isca_dyn_cold_start(core), 'spectral_dynamics_init')
endif
! spectral_diagnostics_init (:1554-1700) and diag_manager's bookkeeping for the fields it registers: the run directory's diag_table is the library's to
! read -- it accumulates the table's fields on the device every step and appends one record per output interval to <file_name>.nc in the run directory
! (csrc/history_nc.cpp); an entry of a module the device core does not hold is its FATAL, by name
if(file_exist('diag_table')) then
  call get_time(Time, seconds, days)
  call chk(isca_dyn_diag_open(core, 'diag_table'//c_null_char, '.'//c_null_char, real(days, c_double)*86400._c_double + real(seconds, c_double)), &
           'spectral_diagnostics_init')
endif
nlon = lon_max; nlat = lat_max; nlev = num_levels; nfour = num_fourier; nsph = num_spherical; ntrace = num_tracers
virtual_t = use_virtual_temperature; ref_sea_level_press = reference_sea_level_press
triang = triang_trunc; finc = fourier_inc
module_is_initialized = .true.

end subroutine spectral_dynamics_init

!===============================================================================================
! get_topography, topography_option = 'input' (init/spectral_init_cond.F90:186-245): the height field and the land mask of INPUT/<topog_file_name>
! (read_data by variable name: the library's netCDF-classic reader), then truncation or -- ocean_topog_smoothing /= 0, the default 0.93 --
! the regularisation over the ocean, both the library's (isca_dyn_set_topography)
subroutine input_topography
real(c_double), allocatable :: zs(:), land(:)
real(c_double) :: lam, frac
integer(c_size_t) :: n, nread
character(len=256) :: path
path = 'INPUT/'//trim(topog_file_name)
if(.not. file_exist(trim(path))) call error_mesg('get_topography','topography_option="'//trim(topography_option)//'"'// &
                     ' but '//trim(path)//' does not exist', FATAL)
n = int(lon_max, c_size_t)*int(lat_max, c_size_t)
allocate(zs(n), land(n))
call chk(isca_nc_read_variable(trim(path)//c_null_char, trim(topog_field_name)//c_null_char, 0_c_int, zs, n, nread), 'get_topography')
if(nread /= n) call error_mesg('get_topography','Topography file contains data on another grid than the atmos model grid', FATAL)
call chk(isca_nc_read_variable(trim(path)//c_null_char, trim(land_field_name)//c_null_char, 0_c_int, land, n, nread), 'get_topography')
if(nread /= n) call error_mesg('get_topography','Land file contains data on another grid than the atmos model grid', FATAL)
call chk(isca_dyn_set_topography(core, zs, land, real(ocean_topog_smoothing, c_double), lam, frac), 'get_topography')
if(ocean_topog_smoothing /= 0.) then
  print '(/,"Message from subroutine get_topography:")'
  print '("lambda=",1pe16.8,"  fraction_smoothed=",1pe16.8,/)', lam, frac
endif
end subroutine input_topography

!===============================================================================================
! get_topography, topography_option = 'gaussian': gaussian_topog_nml's mountains (shared/topography/gaussian_topog.F90, the reference's
! own module) on the core's grid, handed to the library as the surface geopotential before the cold start
subroutine gaussian_topography
real, allocatable :: dlon(:), dlat(:), zs(:,:)
real(c_double), allocatable :: buf(:)
allocate(dlon(lon_max), dlat(lat_max), zs(lon_max, lat_max), buf(lon_max*lat_max))
call get_table1('deg_lon', dlon); call get_table1('deg_lat', dlat)
call gaussian_topog_init(dlon*pi/180, dlat*pi/180, zs)
buf = reshape(grav*zs, (/lon_max*lat_max/))
call chk(isca_dyn_set_surf_geopotential(core, buf, size(buf, kind=c_size_t)), 'get_topography')
end subroutine gaussian_topography

!===============================================================================================
! compute_vert_coord (init/vert_coordinate.F90:89-157) for 'hybrid' (an uneven-sigma profile, compute_uneven_sigma with zero_top = .false.
! :248-273, used as sigma below p_sigma and as pressure above p_press, blended by transition() :161-183), 'mcm' (compute_old_model_sigma
! :296-310) and 'v197' (compute_v197_sigma :276-294): the half levels go to the library like vert_coordinate_nml's
subroutine named_vert_coord(cfg)
type(isca_dyn_config), intent(inout) :: cfg
real, parameter :: v197(19) = (/ 0.0, .0089163, .0342936, .0740741, .1262002, .1886145, .2592592, .3360768, .4170096, .5000000, .5829904, &
                                 .6639231, .7407407, .8113854, .8737997, .9259259, .9657064, .9910837, 1.0 /)
real, parameter :: mcm(15) = (/ 0.0, .03, .0707, .1311, .2102, .3036, .4062, .5138, .6226, .7284, .8255, .9066, .9640, .9933, 1.0 /)
real :: zeta, z, p, f
integer :: k
cfg%vert_coord_input = 1
cfg%pk_input = 0.; cfg%bk_input = 0.
select case(trim(vert_coord_option))
  case('v197')
    if(num_levels /= 18) call error_mesg('compute_v197_sigma','num_levels must be 18', FATAL)
    cfg%bk_input(1:19) = v197
  case('mcm')
    if(num_levels /= 14) call error_mesg('compute_old_model_sigma','num_levels must be 14', FATAL)
    cfg%bk_input(1:15) = mcm
  case('hybrid')
    if(scale_heights == 0. .or. exponent == 0.) call error_mesg('compute_vert_coord','zero is an invalid value for scale_heights / exponent.', FATAL)
    if(surf_res <= 0. .or. surf_res > 1.0) call error_mesg('compute_vert_coord','the namelist parameter surf_res must be < 1.0', FATAL)
    if(p_sigma < p_press) call error_mesg('compute_vert_coord','p_sigma must be greater than p_press', FATAL)
    do k = 1, num_levels + 1
      if(k <= num_levels) then
        zeta = 1. - (real(k-1)/real(num_levels))
        z = surf_res*zeta + (1.0 - surf_res)*(zeta**exponent)
        p = exp(-z*scale_heights)
      else
        p = 1.0
      endif
      if(p <= p_press) then
        f = 0.0
      else if(p >= p_sigma) then
        f = 1.0
      else
        f = (sin(0.5*pi*(p - p_press)/(p_sigma - p_press)))**2
      endif
      cfg%pk_input(k) = reference_sea_level_press*(0.0*f + p*(1.0 - f))
      cfg%bk_input(k) = p*f + 0.0*(1.0 - f)
    enddo
end select
end subroutine named_vert_coord

!===============================================================================================
! spectral_dynamics.F90:280-301
integer function advect_scheme(name, what)
character(len=*), intent(in) :: name, what
select case(uppercase(trim(name)))
  case('SECOND_CENTERED');         advect_scheme = 0
  case('FOURTH_CENTERED');         advect_scheme = 1
  case('VAN_LEER_LINEAR');         advect_scheme = 2
  case('FINITE_VOLUME_PARABOLIC'); advect_scheme = 3
  case default
    advect_scheme = -1
    call error_mesg('spectral_dynamics_init','"'//trim(name)//'"'//' is not a valid value for '//trim(what)//'.', FATAL)
end select
end function advect_scheme

!===============================================================================================
subroutine get_initial_fields(ug_out, vg_out, tg_out, psg_out, grid_tracers_out)
real, intent(out), dimension(:,:,:)   :: ug_out, vg_out, tg_out
real, intent(out), dimension(:,:)     :: psg_out
real, intent(out), dimension(:,:,:,:) :: grid_tracers_out
integer(c_long) :: step
integer :: ntr
call need_core('get_initial_fields')
call chk(isca_dyn_get_info(core, cstr('step'), step), 'get_initial_fields')
if(step /= 0) call error_mesg('get_initial_fields','This routine may be called only to get the initial values after a cold_start', FATAL)
call get_grid3('ug', 1, ug_out); call get_grid3('vg', 1, vg_out); call get_grid3('tg', 1, tg_out); call get_grid2('psg', 1, psg_out)
do ntr = 1, ntrace
  call get_grid3(tracer_name(ntr, .false.), 1, grid_tracers_out(:,:,:,ntr))
enddo
end subroutine get_initial_fields

! the library's name of tracer ntr: 'tr', 'tr2', ... ('tr_atm', 'tr_atm2', ...: the copy atmosphere_mod works with)
function tracer_name(ntr, atm) result(nm)
integer, intent(in) :: ntr
logical, intent(in) :: atm
character(len=8) :: nm
nm = merge('tr_atm', 'tr    ', atm)
if(ntr > 1) write(nm, '(a,i1)') trim(nm), ntr
end function tracer_name

!===============================================================================================
subroutine spectral_dynamics(Time, psg_final, ug_final, vg_final, tg_final, tracer_attributes, grid_tracers_final, &
                             time_level_out, dt_psg, dt_ug, dt_vg, dt_tg, dt_tracers, wg_full, p_full, p_half, z_full)

type(time_type), intent(in) :: Time
real, intent(out), dimension(:,:)       :: psg_final
real, intent(out), dimension(:,:,:)     :: ug_final, vg_final, tg_final
real, intent(out), dimension(:,:,:,:,:) :: grid_tracers_final
type(tracer_type), intent(inout), dimension(:) :: tracer_attributes
integer, intent(in)                     :: time_level_out
real, intent(inout), dimension(:,:)     :: dt_psg
real, intent(inout), dimension(:,:,:)   :: dt_ug, dt_vg, dt_tg
real, intent(inout), dimension(:,:,:,:) :: dt_tracers
real, intent(out),   dimension(:,:,:)   :: wg_full
real, intent(in),    dimension(:,:,:)   :: p_full, z_full
real, intent(in),    dimension(:,:,:)   :: p_half
integer :: ntr
real(c_double), allocatable :: tnd(:)

call need_core('spectral_dynamics')
if(any(dt_psg /= 0.)) call error_mesg('spectral_dynamics','a physics tendency of surface pressure is not carried by the device core', FATAL)
! one step: spectral_dynamics.F90:780-1034 on the device with the caller's tendencies
allocate(tnd(max(size(dt_tracers), 1)))
if(size(dt_tracers) > 0) tnd = reshape(dt_tracers, (/size(dt_tracers)/))
call chk(isca_dyn_dynamics(core, dt_ug, dt_vg, dt_tg, tnd, 0_c_int, 1_c_int), 'spectral_dynamics')
! psg_final = psg(:,:,current) etc. (:1023-1028): the new time level; the tracers as atmosphere_mod keeps them (not Robert-filtered)
call get_grid2('psg', 1, psg_final)
call get_grid3('ug', 1, ug_final); call get_grid3('vg', 1, vg_final); call get_grid3('tg', 1, tg_final)
do ntr = 1, ntrace
  call get_grid3(tracer_name(ntr, .true.), 1, grid_tracers_final(:,:,:,time_level_out,ntr))
enddo
call get_grid3('wg_full', 1, wg_full)
end subroutine spectral_dynamics

!===============================================================================================
! complete_robert_filter (:1456-1490) is part of the device step; complete_update_of_future (:1416-1454) re-derives the spectral side of
! the new level after the caller changed its grid fields
subroutine complete_robert_filter(tracer_attributes, part_filt_ln_ps, part_filt_vors, part_filt_divs, part_filt_ts, part_filt_trs, part_filt_tr)
type(tracer_type), intent(inout), dimension(:) :: tracer_attributes
complex, intent(in), dimension(:,:) :: part_filt_ln_ps
complex, intent(in), dimension(:,:,:) :: part_filt_vors, part_filt_divs, part_filt_ts
complex, intent(in), dimension(:,:,:,:) :: part_filt_trs
real, intent(in), dimension(:,:,:,:) :: part_filt_tr
call error_mesg('complete_robert_filter','the device step completes the Robert filter itself: this routine must not be called', FATAL)
end subroutine complete_robert_filter

subroutine complete_update_of_future(psg, ug, vg, tg, tracer_attributes, grid_tracers)
real, intent(in), dimension(:,:)     :: psg
real, intent(in), dimension(:,:,:)   :: ug, vg, tg
type(tracer_type), intent(in), dimension(:) :: tracer_attributes
real, intent(in), dimension(:,:,:,:) :: grid_tracers
integer :: ntr
call need_core('complete_update_of_future')
call chk(isca_dyn_set_state(core, cstr('psg'), 1_c_int, reshape(psg, (/size(psg)/)), size(psg, kind=c_size_t)), 'complete_update_of_future')
call chk(isca_dyn_set_state(core, cstr('ug'), 1_c_int, reshape(ug, (/size(ug)/)), size(ug, kind=c_size_t)), 'complete_update_of_future')
call chk(isca_dyn_set_state(core, cstr('vg'), 1_c_int, reshape(vg, (/size(vg)/)), size(vg, kind=c_size_t)), 'complete_update_of_future')
call chk(isca_dyn_set_state(core, cstr('tg'), 1_c_int, reshape(tg, (/size(tg)/)), size(tg, kind=c_size_t)), 'complete_update_of_future')
do ntr = 1, ntrace
  call chk(isca_dyn_set_state(core, cstr(tracer_name(ntr, .false.)), 1_c_int, reshape(grid_tracers(:,:,:,ntr), (/size(tg)/)), &
                              size(tg, kind=c_size_t)), 'complete_update_of_future')
  call chk(isca_dyn_set_state(core, cstr(tracer_name(ntr, .true.)), 1_c_int, reshape(grid_tracers(:,:,:,ntr), (/size(tg)/)), &
                              size(tg, kind=c_size_t)), 'complete_update_of_future')
enddo
call chk(isca_dyn_complete_update(core, 1_c_int), 'complete_update_of_future')
end subroutine complete_update_of_future

!===============================================================================================
subroutine spectral_dynamics_end(tracer_attributes, Time)
type(tracer_type), intent(in), dimension(:) :: tracer_attributes
type(time_type), intent(in), optional :: Time
if(.not. module_is_initialized) return
! RESTART/spectral_dynamics.res.nc (:1502-1531) and -- written by atmosphere_end and mixed_layer_end in the reference -- atmosphere.res.nc,
! mixed_layer.res.nc: one call of the library's writer
call chk(isca_dyn_write_restart(core, 'RESTART'//c_null_char, trim(tracer_name_list)//c_null_char), 'spectral_dynamics_end')
call chk(isca_dyn_diag_close(core), 'spectral_dynamics_end')        ! the history files (diag_manager_end's part for them)
call chk(isca_dyn_destroy(core), 'spectral_dynamics_end')
core = c_null_ptr; core_ready = .false.; module_is_initialized = .false.
end subroutine spectral_dynamics_end

! spectral_diagnostics (:1709-1867) sends 20 fields of the new time level to diag_manager after every step (atmosphere.F90:344).  Here the step that
! produced the level has already added them to the device's running sums, and the library writes the diag_table's files itself (spectral_dynamics_init:
! isca_dyn_diag_open; one record per output interval, csrc/history_nc.cpp) -- so a host that calls this routine as the reference's does gets the same
! history files, and nothing crosses PCIe here: the arguments, the host's copies of the new level, are not read.  What is checked is what the
! reference checks first: that the module is up.
subroutine spectral_diagnostics(Time, p_surf, u_grid, v_grid, t_grid, wg_full, tr_grid, time_level)
type(time_type), intent(in) :: Time
real, intent(in), dimension(:,:)       :: p_surf
real, intent(in), dimension(:,:,:)     :: u_grid, v_grid, t_grid, wg_full
real, intent(in), dimension(:,:,:,:,:) :: tr_grid
integer, intent(in) :: time_level
if(.not. module_is_initialized) call error_mesg('spectral_diagnostics','dynamics has not been initialized', FATAL)
if(size(t_grid,3) /= nlev) call error_mesg('spectral_diagnostics','the fields handed over do not have num_levels levels', FATAL)
end subroutine spectral_diagnostics

!===============================================================================================
subroutine get_num_levels(num_levels_out)
integer, intent(out) :: num_levels_out
call need_core('get_num_levels')
num_levels_out = nlev
end subroutine get_num_levels

subroutine get_use_virtual_temperature(use_virtual_temperature_out)
logical, intent(out) :: use_virtual_temperature_out
call need_core('get_use_virtual_temperature')
use_virtual_temperature_out = virtual_t
end subroutine get_use_virtual_temperature

subroutine get_reference_sea_level_press(reference_sea_level_press_out)
real, intent(out) :: reference_sea_level_press_out
call need_core('get_reference_sea_level_press')
reference_sea_level_press_out = ref_sea_level_press
end subroutine get_reference_sea_level_press

subroutine get_surf_geopotential(surf_geopotential_out)
real, intent(out), dimension(:,:) :: surf_geopotential_out
call need_core('get_surf_geopotential')
call get_grid2('surf_geopotential', 1, surf_geopotential_out)
end subroutine get_surf_geopotential

subroutine get_pk_bk(pk_out, bk_out)
real, intent(out), dimension(:) :: pk_out, bk_out
call need_core('get_pk_bk')
if(size(pk_out) /= nlev+1 .or. size(bk_out) /= nlev+1) call error_mesg('get_pk_bk','pk_out and bk_out must have num_levels+1 values', FATAL)
call get_table1('pk', pk_out); call get_table1('bk', bk_out)
end subroutine get_pk_bk

! the axes diag_manager would know the dynamics' fields by: none are registered from here
function get_axis_id()
integer, dimension(4) :: get_axis_id
get_axis_id = 0
end function get_axis_id

end module spectral_dynamics_mod
