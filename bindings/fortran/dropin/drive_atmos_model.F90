! The main program's view of the drop-in (atmos_solo/atmos_model.F90:115-142, 148-260): infrastructure initialisation, atmosphere_init,
! the time loop `do na = 1, num_atmos_calls: call atmosphere(Time); Time = Time + Time_step`, atmosphere_end -- with this directory's
! atmosphere_mod, i.e. every step one call into the MI355X library and no field crossing PCIe.  tests/test_gpu_fortran_dropin.py runs
! it for the reference's Held-Suarez test case (input.nml, field_table, diag_table of the run directory) and compares the printed
! state with the reference run.  Number of steps and dt_atmos: main_nml-like values in drive.nml.
program drive_atmos_model
use iso_c_binding
use constants_mod,      only: constants_init
use fms_mod,            only: fms_init, fms_end
use time_manager_mod,   only: time_type, set_time, set_calendar_type, NO_CALENDAR, operator(+)
use field_manager_mod,  only: MODEL_ATMOS
use tracer_manager_mod, only: register_tracers
use diag_manager_mod,   only: diag_manager_init
use atmosphere_mod,     only: atmosphere_init, atmosphere, atmosphere_end
use spectral_dynamics_mod, only: get_num_levels
use transforms_mod,     only: get_grid_domain, area_weighted_global_mean
use isca_dropin_mod,    only: get_grid3, get_grid2
implicit none
integer :: nsteps = 144, dt_atmos = 600
namelist /drive_nml/ nsteps, dt_atmos
type(time_type) :: Time, Time_init, Time_step
integer :: ntrace, ntprog, ntdiag, ntfamily, na, unit, is, ie, js, je, nlev
real, allocatable :: tg(:,:,:), ug(:,:,:), psg(:,:)

open(newunit=unit, file='drive.nml', status='old', action='read')
read(unit, nml=drive_nml)
close(unit)
call fms_init()
call constants_init()
call register_tracers(MODEL_ATMOS, ntrace, ntprog, ntdiag, ntfamily)
call set_calendar_type(NO_CALENDAR)
call diag_manager_init()
Time_init = set_time(0, 0); Time = Time_init; Time_step = set_time(dt_atmos, 0)
call atmosphere_init(Time_init, Time, Time_step)
do na = 1, nsteps
  call atmosphere(Time)
  Time = Time + Time_step
enddo
call get_grid_domain(is, ie, js, je)
call get_num_levels(nlev)
allocate(tg(is:ie, js:je, nlev), ug(is:ie, js:je, nlev), psg(is:ie, js:je))
call get_grid3('tg', 1, tg); call get_grid3('ug', 1, ug); call get_grid2('psg', 1, psg)
write(*,'(a,3es24.16)') 'DRIVE_STATE Tmin,Tmax,maxabsU=', minval(tg), maxval(tg), maxval(abs(ug))
write(*,'(a,es24.16)') 'DRIVE_MEAN_PS', area_weighted_global_mean(psg)
call atmosphere_end
end program drive_atmos_model
