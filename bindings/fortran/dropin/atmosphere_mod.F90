! atmosphere_mod -- the reference's driver interface (atmos_spectral/driver/solo/atmosphere.F90:78: atmosphere_init, atmosphere,
! atmosphere_end, atmosphere_domain) with the whole step on the device: atmosphere(Time) is ONE call of isca_dyn_step -- hs_forcing (or,
! with atmosphere_nml: idealized_moist_model = .true., the Frierson column chain whose namelists idealized_moist_phys_init reads),
! spectral_dynamics, the pressures and heights of the new level -- queued on the device's stream; nothing crosses PCIe.  The main program (atmos_solo/atmos_model.F90:115-142) calls these four names and nothing else of the core.
!
! Restart files (the reference's variable set, netCDF classic format): spectral_dynamics_init reads INPUT/*.res.nc when they exist, spectral_dynamics_end
! writes RESTART/spectral_dynamics.res.nc, atmosphere.res.nc and mixed_layer.res.nc through the library (this image has no netCDF for fms_io).
module atmosphere_mod

#ifdef INTERNAL_FILE_NML
use mpp_mod, only: input_nml_file
#else
use fms_mod, only: open_namelist_file
#endif
use iso_c_binding
use fms_mod,            only: error_mesg, FATAL, check_nml_error, close_file
use time_manager_mod,   only: time_type
use field_manager_mod,  only: MODEL_ATMOS
use tracer_manager_mod, only: get_number_tracers
use mpp_domains_mod,    only: domain2d, mpp_define_domains
use tracer_type_mod,    only: tracer_type
use spectral_dynamics_mod, only: spectral_dynamics_init, spectral_dynamics_end
use transforms_mod,     only: get_grid_domain
use idealized_moist_phys_mod, only: idealized_moist_phys_init, idealized_moist_phys_end
use isca_dyn_c
use isca_dropin_mod

implicit none
private
public :: atmosphere_init, atmosphere, atmosphere_end, atmosphere_domain

logical :: idealized_moist_model = .false.
namelist /atmosphere_nml/ idealized_moist_model

type(tracer_type), allocatable, dimension(:) :: tracer_attributes
logical :: module_is_initialized = .false.
logical :: dry_model
integer :: nhum
integer :: steps_queued = 0
integer, parameter :: sync_every = 256

contains

subroutine atmosphere_init(Time_init, Time, Time_step_in)
type(time_type), intent(in) :: Time_init, Time, Time_step_in
integer :: io, ierr, unit, num_tracers
if(module_is_initialized) return
#ifdef INTERNAL_FILE_NML
read(input_nml_file, nml=atmosphere_nml, iostat=io)
ierr = check_nml_error(io, 'atmosphere_nml')
#else
unit = open_namelist_file()
ierr = 1
do while(ierr /= 0)
  read(unit, nml=atmosphere_nml, iostat=io, end=20)
  ierr = check_nml_error(io, 'atmosphere_nml')
enddo
20 call close_file(unit)
#endif
call get_number_tracers(MODEL_ATMOS, num_prog=num_tracers)
allocate(tracer_attributes(num_tracers))
if(idealized_moist_model) then
  call idealized_moist_phys_init     ! atmosphere.F90:246-262: the namelists of the Frierson chain, which then runs inside the device step
  dropin_physics = 1
else
  dropin_physics = 0                 ! hs_forcing inside the device step (atmosphere.F90:304-311)
endif
call spectral_dynamics_init(Time, Time_step_in, tracer_attributes, dry_model, nhum)
if(idealized_moist_model .and. dry_model) call error_mesg('atmosphere_init', &
  'idealized_moist_phys: the field_table has no specific-humidity tracer (sphum / mix_rat)', FATAL)
steps_queued = 0
module_is_initialized = .true.
end subroutine atmosphere_init

subroutine atmosphere(Time)
type(time_type), intent(in) :: Time
if(.not. module_is_initialized) call error_mesg('atmosphere','atmosphere module is not initialized', FATAL)
! the step is QUEUED on the core's stream (sync = 0): the host returns at once and the main program's loop runs ahead of the device, as
! isca_amd's own step(n) does.  The device is waited for -- and valid_range_t checked (spectral_dynamics.F90:940-972: FATAL) -- every
! sync_every calls, whenever state is read (isca_dyn_get_state synchronises), and in atmosphere_end.
call chk(isca_dyn_step(core, 1_c_int, 0_c_int), 'atmosphere')
steps_queued = steps_queued + 1
if(steps_queued >= sync_every) then
  call chk(isca_dyn_synchronize(core), 'atmosphere')
  steps_queued = 0
endif
end subroutine atmosphere

subroutine atmosphere_end
if(.not. module_is_initialized) return
call chk(isca_dyn_synchronize(core), 'atmosphere_end')
if(idealized_moist_model) call idealized_moist_phys_end
call spectral_dynamics_end(tracer_attributes)
deallocate(tracer_attributes)
module_is_initialized = .false.
end subroutine atmosphere_end

subroutine atmosphere_domain(Domain)
type(domain2d), intent(inout) :: Domain
integer :: is_d, ie_d, js_d, je_d
call need_core('atmosphere_domain')
! the reference hands back the decomposed grid domain (atmosphere.F90:390: get_grid_domain's).  This mpp has no message passing (every process is
! PE 0 of 1), so the domain it can describe is this process's own: all longitudes by the latitude band the library gave it (one rank: the whole grid)
call get_grid_domain(is_d, ie_d, js_d, je_d)
call mpp_define_domains((/is_d, ie_d, js_d, je_d/), (/1, 1/), Domain)
end subroutine atmosphere_domain

end module atmosphere_mod
