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
use transforms_mod,     only: get_grid_domain, get_lat_max, area_weighted_global_mean
use isca_dropin_mod,    only: get_grid3, get_grid2
implicit none
integer :: nsteps = 144, dt_atmos = 600
namelist /drive_nml/ nsteps, dt_atmos
type(time_type) :: Time, Time_init, Time_step
integer :: ntrace, ntprog, ntdiag, ntfamily, na, unit, is, ie, js, je, nlev, latmax
real, allocatable :: tg(:,:,:), ug(:,:,:), psg(:,:), q(:,:,:)
integer(kind=8) :: c0, c1, c2, crate
integer :: warm

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
! the loop of atmos_model.F90:339-349; timed from step `warm` on (set-up and clock ramp outside), the device waited for at both ends
call get_grid_domain(is, ie, js, je)
call get_num_levels(nlev)
allocate(tg(is:ie, js:je, nlev), ug(is:ie, js:je, nlev), psg(is:ie, js:je), q(is:ie, js:je, nlev))
warm = min(nsteps/4, 2000)
call system_clock(count_rate=crate)
do na = 1, nsteps
  if(na == warm + 1) then
    call get_grid2('psg', 1, psg)             ! (reading state waits for the device)
    call system_clock(c0)
  endif
  call atmosphere(Time)
  Time = Time + Time_step
enddo
call system_clock(c1)
call get_grid2('psg', 1, psg)
call system_clock(c2)
write(*,'(a,i8,a,f12.6,a,f12.6)') 'DRIVE_TIMING steps=', nsteps - warm, ' seconds=', real(c2-c0,8)/real(crate,8), &
     ' ms_per_step=', 1.e3*real(c2-c0,8)/real(crate,8)/max(nsteps - warm, 1)
write(*,'(a,f12.6)') 'DRIVE_HOST_AHEAD_S ', real(c2-c1,8)/real(crate,8)     ! how far the loop ran ahead of the device
call get_grid3('tg', 1, tg); call get_grid3('ug', 1, ug)
write(*,'(a,3es24.16)') 'DRIVE_STATE Tmin,Tmax,maxabsU=', minval(tg), maxval(tg), maxval(abs(ug))
if(ntprog > 0) then
  call get_grid3('tr', 1, q)
  write(*,'(a,2es24.16)') 'DRIVE_TRACER qmax,q(10,16,nlev)=', maxval(q), q(is+9, min(js+15, je), nlev)
  if(nlev > 1) write(*,'(a,es24.16)') 'DRIVE_TRACER_ALOFT max q(:,:,nlev-1)=', maxval(q(:,:,nlev-1))    ! (what the vertical advection carried up)
endif
call get_lat_max(latmax)
write(*,'(a,2i6)') 'DRIVE_ROWS js,je=', js, je       ! this process's latitude band (all rows with one rank)
if(je - js + 1 == latmax) write(*,'(a,es24.16)') 'DRIVE_MEAN_PS', area_weighted_global_mean(psg)
call atmosphere_end
end program drive_atmos_model
