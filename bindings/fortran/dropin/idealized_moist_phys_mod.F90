! idealized_moist_phys_mod -- the reference's module name (atmos_spectral/driver/solo/idealized_moist_phys.F90:1-1395) in front of the
! device's Frierson column chain.  In the reference atmosphere_init calls idealized_moist_phys_init (atmosphere.F90:246-262), which
! reads idealized_moist_phys_nml and initialises two_stream_gray_rad, mixed_layer, qe_moist_convection, lscale_cond, damping_driver,
! vert_turb_driver / diffusivity, surface_flux / monin_obukhov -- each reading its own namelist -- and atmosphere(Time) calls
! idealized_moist_phys every step (:313-319).  Here the chain runs INSIDE the device step (isca_dyn_config%physics = 1, HISTORY.md 9), so
! what is left on the host is the namelists: idealized_moist_phys_init reads every group the chain reads, with the reference's variable
! names and MODULE defaults, refuses the option values the device package does not implement ("... is not a supported value", the wording
! of isca_amd/atmosphere.py:_moist_config), and leaves the values in isca_dropin_mod%dropin_moist for spectral_dynamics_init.
! A variable of one of these groups that is not declared here is a namelist error (check_nml_error: FATAL), never ignored.
module idealized_moist_phys_mod

#ifdef INTERNAL_FILE_NML
use mpp_mod, only: input_nml_file
#else
use fms_mod, only: open_namelist_file
#endif
use iso_c_binding
use fms_mod, only: error_mesg, FATAL, check_nml_error, close_file, uppercase
use isca_dyn_c
use isca_dropin_mod

implicit none
private
public :: idealized_moist_phys_init, idealized_moist_phys_end

! ---- idealized_moist_phys_nml (idealized_moist_phys.F90:84-138, 140-151)
logical :: two_stream_gray = .true., do_rrtm_radiation = .false., do_socrates_radiation = .false., do_damping = .false., turb = .false., &
           do_virtual = .false., mixed_layer_bc = .false., do_simple = .false., do_cloud_simple = .false., bucket = .false., &
           gp_surface = .false., do_lcl_diffusivity_depth = .false., lwet_convection = .false., do_bm = .false., do_ras = .false.
character(len=256) :: convection_scheme = 'unset', land_option = 'none'
real :: roughness_mom = 0.05, roughness_heat = 0.05, roughness_moist = 0.05
namelist /idealized_moist_phys_nml/ two_stream_gray, do_rrtm_radiation, do_socrates_radiation, do_damping, turb, do_virtual, &
         mixed_layer_bc, do_simple, do_cloud_simple, bucket, gp_surface, do_lcl_diffusivity_depth, lwet_convection, do_bm, do_ras, &
         convection_scheme, land_option, roughness_mom, roughness_heat, roughness_moist

! ---- two_stream_gray_rad_nml (atmos_param/two_stream_gray_rad/two_stream_gray_rad.F90:72-140)
character(len=32) :: rad_scheme = 'frierson'
logical :: do_seasonal = .false.
real :: solar_constant = 1360.0, del_sol = 1.4, del_sw = 0.0, ir_tau_eq = 6.0, ir_tau_pole = 1.5, atm_abs = 0.0, odp = 1.0, sw_diff = 0.0, &
        linear_tau = 0.1, wv_exponent = 4.0, solar_exponent = 4.0
namelist /two_stream_gray_rad_nml/ rad_scheme, do_seasonal, solar_constant, del_sol, del_sw, ir_tau_eq, ir_tau_pole, atm_abs, odp, sw_diff, &
         linear_tau, wv_exponent, solar_exponent

! ---- mixed_layer_nml (atmos_spectral/driver/solo/mixed_layer.F90:84-132)
logical :: evaporation = .true., prescribe_initial_dist = .false., do_qflux = .false., do_sc_sst = .false., do_ape_sst = .false., &
           update_albedo_from_ice = .false.
real :: depth = 40.0, tconst = 305.0, delta_T = 40.0, albedo_value = 0.06
namelist /mixed_layer_nml/ evaporation, prescribe_initial_dist, do_qflux, do_sc_sst, do_ape_sst, update_albedo_from_ice, depth, tconst, &
         delta_T, albedo_value

! ---- qe_moist_convection_nml (atmos_param/qe_moist_convection/qe_moist_convection.F90:66-70)
real :: tau_bm = 7200., rhbm = .8, Tmin = 173., Tmax = 335., val_inc = 0.01
namelist /qe_moist_convection_nml/ tau_bm, rhbm, Tmin, Tmax, val_inc

! ---- lscale_cond_nml (atmos_param/lscale_cond/lscale_cond.F90), sat_vapor_pres_nml (shared/sat_vapor_pres/sat_vapor_pres.F90)
logical :: lc_do_simple = .false., do_evap = .false.
logical :: svp_do_simple = .false.

! ---- damping_driver_nml (atmos_param/damping_driver/damping_driver.f90:42-56)
logical :: do_rayleigh = .false., dd_do_conserve_energy = .false., do_cg_drag = .false., do_mg_drag = .false., do_topo_drag = .false.
real :: trayfric = 0., sponge_pbottom = 50.

! ---- vert_turb_driver_nml (atmos_param/vert_turb_driver/vert_turb_driver.F90:95-122), diffusivity_nml (diffusivity.F90:100-130)
logical :: do_mellor_yamada = .true., do_diffusivity = .false., vt_do_simple = .false., use_tau = .true., do_shallow_conv = .false., &
           do_molecular_diffusion = .false.
character(len=16) :: gust_scheme = 'constant'
real :: constant_gust = 1.0
logical :: do_entrain = .true., df_do_simple = .false., fixed_depth = .false., free_atm_diff = .false., pbl_mcm = .false.
real :: frac_inner = 0.1, rich_crit_pbl = 1.0

! ---- surface_flux_nml (coupler/surface_flux.F90:232-258), monin_obukhov_nml (atmos_param/monin_obukhov/monin_obukhov.F90:88-95)
logical :: use_virtual_temp = .true., sf_do_simple = .false., old_dtaudv = .false.
logical :: neutral = .false.
integer :: stable_option = 1
real :: rich_crit = 2.0, drag_min = 1.e-05

contains

! One group at a time; the groups whose variable names collide (do_simple of four modules, do_conserve_energy) are read through local
! namelists that give the module-level variables above their group's own names.
subroutine read_group(which)
character(len=*), intent(in) :: which
integer :: io, ierr, unit
logical :: do_simple, do_conserve_energy
namelist /lscale_cond_nml/ do_simple, do_evap
namelist /sat_vapor_pres_nml/ do_simple
namelist /damping_driver_nml/ do_rayleigh, trayfric, sponge_pbottom, do_conserve_energy, do_cg_drag, do_mg_drag, do_topo_drag
namelist /vert_turb_driver_nml/ do_mellor_yamada, do_diffusivity, do_simple, use_tau, do_shallow_conv, do_molecular_diffusion, gust_scheme, &
         constant_gust
namelist /diffusivity_nml/ do_entrain, do_simple, fixed_depth, free_atm_diff, pbl_mcm, frac_inner, rich_crit_pbl
namelist /surface_flux_nml/ use_virtual_temp, do_simple, old_dtaudv
namelist /monin_obukhov_nml/ neutral, stable_option, rich_crit, drag_min
select case(which)
  case('lscale_cond_nml');      do_simple = lc_do_simple
  case('sat_vapor_pres_nml');   do_simple = svp_do_simple
  case('vert_turb_driver_nml'); do_simple = vt_do_simple
  case('diffusivity_nml');      do_simple = df_do_simple
  case('surface_flux_nml');     do_simple = sf_do_simple
  case default;                 do_simple = .false.
end select
do_conserve_energy = dd_do_conserve_energy
#ifdef INTERNAL_FILE_NML
io = 0
select case(which)
  case('idealized_moist_phys_nml'); read(input_nml_file, nml=idealized_moist_phys_nml, iostat=io)
  case('two_stream_gray_rad_nml');  read(input_nml_file, nml=two_stream_gray_rad_nml, iostat=io)
  case('mixed_layer_nml');          read(input_nml_file, nml=mixed_layer_nml, iostat=io)
  case('qe_moist_convection_nml');  read(input_nml_file, nml=qe_moist_convection_nml, iostat=io)
  case('lscale_cond_nml');          read(input_nml_file, nml=lscale_cond_nml, iostat=io)
  case('sat_vapor_pres_nml');       read(input_nml_file, nml=sat_vapor_pres_nml, iostat=io)
  case('damping_driver_nml');       read(input_nml_file, nml=damping_driver_nml, iostat=io)
  case('vert_turb_driver_nml');     read(input_nml_file, nml=vert_turb_driver_nml, iostat=io)
  case('diffusivity_nml');          read(input_nml_file, nml=diffusivity_nml, iostat=io)
  case('surface_flux_nml');         read(input_nml_file, nml=surface_flux_nml, iostat=io)
  case('monin_obukhov_nml');        read(input_nml_file, nml=monin_obukhov_nml, iostat=io)
end select
ierr = check_nml_error(io, which)
#else
unit = open_namelist_file()
ierr = 1
do while(ierr /= 0)
  select case(which)
    case('idealized_moist_phys_nml'); read(unit, nml=idealized_moist_phys_nml, iostat=io, end=20)
    case('two_stream_gray_rad_nml');  read(unit, nml=two_stream_gray_rad_nml, iostat=io, end=20)
    case('mixed_layer_nml');          read(unit, nml=mixed_layer_nml, iostat=io, end=20)
    case('qe_moist_convection_nml');  read(unit, nml=qe_moist_convection_nml, iostat=io, end=20)
    case('lscale_cond_nml');          read(unit, nml=lscale_cond_nml, iostat=io, end=20)
    case('sat_vapor_pres_nml');       read(unit, nml=sat_vapor_pres_nml, iostat=io, end=20)
    case('damping_driver_nml');       read(unit, nml=damping_driver_nml, iostat=io, end=20)
    case('vert_turb_driver_nml');     read(unit, nml=vert_turb_driver_nml, iostat=io, end=20)
    case('diffusivity_nml');          read(unit, nml=diffusivity_nml, iostat=io, end=20)
    case('surface_flux_nml');         read(unit, nml=surface_flux_nml, iostat=io, end=20)
    case('monin_obukhov_nml');        read(unit, nml=monin_obukhov_nml, iostat=io, end=20)
  end select
  ierr = check_nml_error(io, which)
enddo
20 call close_file(unit)
#endif
select case(which)
  case('lscale_cond_nml');      lc_do_simple = do_simple
  case('sat_vapor_pres_nml');   svp_do_simple = do_simple
  case('vert_turb_driver_nml'); vt_do_simple = do_simple
  case('diffusivity_nml');      df_do_simple = do_simple
  case('surface_flux_nml');     sf_do_simple = do_simple
  case('damping_driver_nml');   dd_do_conserve_energy = do_conserve_energy
end select
end subroutine read_group

! only ONE value of these options is implemented on the device (isca_amd/atmosphere.py _MOIST_FIXED); the reference's default of several
! of them is another one, so an input.nml that leaves them out asks for something else and is refused the same way
subroutine want(group, name, have, need)
character(len=*), intent(in) :: group, name
logical, intent(in) :: have, need
character(len=8) :: v
if(have .eqv. need) return
v = merge('.true. ', '.false.', have)
call error_mesg('idealized_moist_phys_init', trim(group)//': "'//trim(v)//'" is not a supported value for '//trim(name)// &
                ' (the device physics package implements '//trim(merge('.true. ', '.false.', need))//' only)', FATAL)
end subroutine want

subroutine idealized_moist_phys_init
! (the reference's argument list -- Time, Time_step, nhum, rad_lon_2d, ... -- serves its host-side diagnostics and grids; the device
!  package takes its grid from the core)
call read_group('idealized_moist_phys_nml'); call read_group('two_stream_gray_rad_nml'); call read_group('mixed_layer_nml')
call read_group('qe_moist_convection_nml');  call read_group('lscale_cond_nml');         call read_group('sat_vapor_pres_nml')
call read_group('damping_driver_nml');       call read_group('vert_turb_driver_nml');    call read_group('diffusivity_nml')
call read_group('surface_flux_nml');         call read_group('monin_obukhov_nml')

call want('idealized_moist_phys_nml', 'two_stream_gray', two_stream_gray, .true.)
call want('idealized_moist_phys_nml', 'do_rrtm_radiation', do_rrtm_radiation, .false.)
call want('idealized_moist_phys_nml', 'do_socrates_radiation', do_socrates_radiation, .false.)
call want('idealized_moist_phys_nml', 'do_damping', do_damping, .true.)
call want('idealized_moist_phys_nml', 'turb', turb, .true.)
call want('idealized_moist_phys_nml', 'mixed_layer_bc', mixed_layer_bc, .true.)
call want('idealized_moist_phys_nml', 'do_virtual', do_virtual, .false.)
call want('idealized_moist_phys_nml', 'do_simple', do_simple, .true.)
call want('idealized_moist_phys_nml', 'do_cloud_simple', do_cloud_simple, .false.)
call want('idealized_moist_phys_nml', 'bucket', bucket, .false.)
call want('idealized_moist_phys_nml', 'gp_surface', gp_surface, .false.)
call want('idealized_moist_phys_nml', 'do_lcl_diffusivity_depth', do_lcl_diffusivity_depth, .false.)
if(uppercase(trim(convection_scheme)) /= 'SIMPLE_BETTS_MILLER') &
  call error_mesg('idealized_moist_phys_init', 'idealized_moist_phys_nml: "'//trim(convection_scheme)//'" is not a supported value for '// &
                  'convection_scheme (only "SIMPLE_BETTS_MILLER")', FATAL)
if(lwet_convection .or. do_bm .or. do_ras) call error_mesg('idealized_moist_phys_init', &
  'idealized_moist_phys_nml: lwet_convection / do_bm / do_ras are not supported (convection_scheme = "SIMPLE_BETTS_MILLER")', FATAL)
if(trim(land_option) /= 'none') call error_mesg('idealized_moist_phys_init', &
  'idealized_moist_phys_nml: "'//trim(land_option)//'" is not a supported value for land_option (only "none")', FATAL)
if(uppercase(trim(rad_scheme)) /= 'FRIERSON') call error_mesg('two_stream_gray_rad_init', &
  'two_stream_gray_rad_nml: "'//trim(rad_scheme)//'" is not a supported value for rad_scheme (only "frierson")', FATAL)
call want('two_stream_gray_rad_nml', 'do_seasonal', do_seasonal, .false.)
call want('mixed_layer_nml', 'prescribe_initial_dist', prescribe_initial_dist, .true.)
call want('mixed_layer_nml', 'do_qflux', do_qflux, .false.)
call want('mixed_layer_nml', 'do_sc_sst', do_sc_sst, .false.)
call want('mixed_layer_nml', 'do_ape_sst', do_ape_sst, .false.)
call want('mixed_layer_nml', 'update_albedo_from_ice', update_albedo_from_ice, .false.)
call want('lscale_cond_nml', 'do_simple', lc_do_simple, .true.)
call want('lscale_cond_nml', 'do_evap', do_evap, .true.)
call want('sat_vapor_pres_nml', 'do_simple', svp_do_simple, .true.)
call want('damping_driver_nml', 'do_cg_drag', do_cg_drag, .false.)
call want('damping_driver_nml', 'do_mg_drag', do_mg_drag, .false.)
call want('damping_driver_nml', 'do_topo_drag', do_topo_drag, .false.)
call want('vert_turb_driver_nml', 'do_mellor_yamada', do_mellor_yamada, .false.)
call want('vert_turb_driver_nml', 'do_diffusivity', do_diffusivity, .true.)
call want('vert_turb_driver_nml', 'do_simple', vt_do_simple, .true.)
call want('vert_turb_driver_nml', 'use_tau', use_tau, .false.)
call want('vert_turb_driver_nml', 'do_shallow_conv', do_shallow_conv, .false.)
call want('vert_turb_driver_nml', 'do_molecular_diffusion', do_molecular_diffusion, .false.)
if(trim(gust_scheme) /= 'constant') call error_mesg('vert_turb_driver_init', &
  'vert_turb_driver_nml: "'//trim(gust_scheme)//'" is not a supported value for gust_scheme (only "constant")', FATAL)
call want('diffusivity_nml', 'do_entrain', do_entrain, .false.)
call want('diffusivity_nml', 'do_simple', df_do_simple, .true.)
call want('diffusivity_nml', 'fixed_depth', fixed_depth, .false.)
call want('diffusivity_nml', 'free_atm_diff', free_atm_diff, .false.)
call want('diffusivity_nml', 'pbl_mcm', pbl_mcm, .false.)
call want('surface_flux_nml', 'use_virtual_temp', use_virtual_temp, .false.)
call want('surface_flux_nml', 'do_simple', sf_do_simple, .true.)
call want('surface_flux_nml', 'old_dtaudv', old_dtaudv, .true.)
call want('monin_obukhov_nml', 'neutral', neutral, .false.)
if(stable_option /= 1) call error_mesg('monin_obukhov_init', 'monin_obukhov_nml: stable_option must be 1', FATAL)

dropin_moist%roughness_mom = roughness_mom; dropin_moist%roughness_heat = roughness_heat; dropin_moist%roughness_moist = roughness_moist
dropin_moist%solar_constant = solar_constant; dropin_moist%del_sol = del_sol; dropin_moist%del_sw = del_sw
dropin_moist%ir_tau_eq = ir_tau_eq; dropin_moist%ir_tau_pole = ir_tau_pole; dropin_moist%atm_abs = atm_abs; dropin_moist%odp = odp
dropin_moist%sw_diff = sw_diff; dropin_moist%linear_tau = linear_tau; dropin_moist%wv_exponent = wv_exponent
dropin_moist%solar_exponent = solar_exponent
dropin_moist%depth = depth; dropin_moist%tconst = tconst; dropin_moist%delta_T = delta_T; dropin_moist%albedo_value = albedo_value
dropin_moist%evaporation = merge(1, 0, evaporation)
dropin_moist%tau_bm = tau_bm; dropin_moist%rhbm = rhbm; dropin_moist%Tmin = Tmin; dropin_moist%Tmax = Tmax; dropin_moist%val_inc = val_inc
dropin_moist%do_rayleigh = merge(1, 0, do_rayleigh); dropin_moist%trayfric = trayfric; dropin_moist%sponge_pbottom = sponge_pbottom
dropin_moist%damping_conserve_energy = merge(1, 0, dd_do_conserve_energy)
dropin_moist%constant_gust = constant_gust
dropin_moist%frac_inner = frac_inner; dropin_moist%rich_crit_pbl = rich_crit_pbl
dropin_moist%rich_crit = rich_crit; dropin_moist%drag_min = drag_min
dropin_moist_set = .true.
end subroutine idealized_moist_phys_init

subroutine idealized_moist_phys_end
dropin_moist_set = .false.
end subroutine idealized_moist_phys_end

end module idealized_moist_phys_mod
