! oracle/ref_shallow_harness.F90 -- TEST INFRASTRUCTURE ONLY (never linked into the product).
!
! Driver program (ours) for the reference's shallow-water sibling core (src/atmos_spectral_shallow), compiled in place by
! oracle/build_ref.py shallow.  It calls only public module procedures -- shallow_dynamics_init, shallow_physics_init,
! stirring_init, then per step shallow_physics and shallow_dynamics with the time-level bookkeeping of
! src/atmos_spectral_shallow/atmosphere.F90:117-197 (whose own state is private, hence this driver) -- and dumps raw
! little-endian fp64 arrays (Fortran order) after the steps listed in harness.nml.
program ref_shallow_harness

use constants_mod,        only: constants_init
use fms_mod,              only: fms_init
use time_manager_mod,     only: time_type, set_time, set_calendar_type, NO_CALENDAR, operator(+)
use diag_manager_mod,     only: diag_manager_init
use transforms_mod,       only: get_grid_domain, get_spec_domain, get_deg_lat, get_num_fourier, get_num_spherical
use shallow_dynamics_mod, only: shallow_dynamics_init, shallow_dynamics, dynamics_type
use shallow_physics_mod,  only: shallow_physics_init, shallow_physics, phys_type
use stirring_mod,         only: stirring_init

implicit none

integer :: nsteps = 1, dt_atmos = 1200
integer, dimension(64) :: dump_steps = -1
logical :: dump_random = .false.      ! stirring on: write the random numbers each step's stirring call will draw (in_stir_ran.bin)
namelist /harness_nml/ nsteps, dt_atmos, dump_steps, dump_random

type(time_type)     :: Time, Time_init, Time_step
type(dynamics_type) :: Dyn
type(phys_type)     :: Phys
integer :: is, ie, js, je, ms, me, ns, ne, previous, current, future, istep, unit
real    :: dt_real, delta_t
integer(kind=8) :: c0, c1, crate
real, allocatable :: deg_lat(:), ran(:,:,:)
integer, allocatable :: seed(:)
integer :: nseed, nf, nsph, uran

open(newunit=unit, file='harness.nml', status='old', action='read')
read(unit, nml=harness_nml)
close(unit)

call fms_init()
call constants_init()
call set_calendar_type(NO_CALENDAR)
call diag_manager_init()
Time_init = set_time(0, 0)
Time      = Time_init
Time_step = set_time(dt_atmos, 0)
dt_real   = real(dt_atmos)

call shallow_dynamics_init(Dyn, Time, Time_init, dt_real)
call get_grid_domain(is, ie, js, je)
call get_spec_domain(ms, me, ns, ne)
call shallow_physics_init(Phys)
call stirring_init(dt_real, Time, 0, 0, 0, 0)
previous = 1; current = 1
allocate(deg_lat(js:je)); call get_deg_lat(deg_lat)
call dump1('tab_deg_lat.bin', deg_lat)
call dump2('tab_deep_geopot.bin', Dyn%Grid%deep_geopot)
call dump_state(0)
if(dump_random) then
  call get_num_fourier(nf); call get_num_spherical(nsph)
  allocate(ran(0:nf,0:nsph,2))
  call random_seed(size=nseed); allocate(seed(nseed))
  open(newunit=uran, file='in_stir_ran.bin', access='stream', form='unformatted', status='replace')
endif

call system_clock(c0, crate)
do istep = 1, nsteps
  Dyn%Tend%u = 0.0; Dyn%Tend%v = 0.0; Dyn%Tend%h = 0.0
  if(Dyn%grid_tracer) Dyn%Tend%tr  = 0.0
  if(Dyn%spec_tracer) Dyn%Tend%trs = 0.0
  if(dump_random) then        ! the numbers stirring() is about to draw: draw them, write them, rewind the generator
    call random_seed(get=seed); call random_number(ran); write(uran) ran; call random_seed(put=seed)
  endif
  if(istep == 1) then
    delta_t = dt_real; future = 2
  else
    delta_t = 2.0*dt_real; future = previous
  endif
  call shallow_physics(Time, Dyn%Tend%u, Dyn%Tend%v, Dyn%Tend%h, Dyn%Grid%u, Dyn%Grid%v, Dyn%Grid%h, delta_t, previous, current, Phys)
  call shallow_dynamics(Time, Time_init, Dyn, previous, current, future, delta_t)
  previous = current
  current  = future
  Time = Time + Time_step
  if(any(dump_steps == istep)) call dump_state(istep)
enddo
call system_clock(c1)
write(*,'(a,i8,a,f12.6,a,f12.6)') 'REF_TIMING steps=', nsteps, ' seconds=', real(c1-c0,8)/real(crate,8), &
      ' ms_per_step=', 1.d3*real(c1-c0,8)/real(crate,8)/max(nsteps,1)
write(*,'(a,3es24.16)') 'REF_STATE hmin,hmax,maxabsU=', minval(Dyn%Grid%h(:,:,current)), maxval(Dyn%Grid%h(:,:,current)), &
      maxval(abs(Dyn%Grid%u(:,:,current)))

contains

subroutine dump_state(n)
integer, intent(in) :: n
character(len=6) :: tag
write(tag,'(i6.6)') n
call dump2('st_u_'//tag//'.bin', Dyn%Grid%u(:,:,current));     call dump2('st_v_'//tag//'.bin', Dyn%Grid%v(:,:,current))
call dump2('st_vor_'//tag//'.bin', Dyn%Grid%vor(:,:,current)); call dump2('st_div_'//tag//'.bin', Dyn%Grid%div(:,:,current))
call dump2('st_h_'//tag//'.bin', Dyn%Grid%h(:,:,current))
if(Dyn%grid_tracer) call dump2('st_tr_'//tag//'.bin', Dyn%Grid%tr(:,:,current))
if(Dyn%spec_tracer) call dump2('st_trs_'//tag//'.bin', Dyn%Grid%trs(:,:,current))
call dumpc('st_vors_'//tag//'.bin', Dyn%Spec%vor(:,:,current)); call dumpc('st_hs_'//tag//'.bin', Dyn%Spec%h(:,:,current))
if(n > 0) then
  call dump2('st_stream_'//tag//'.bin', Dyn%Grid%stream); call dump2('st_pv_'//tag//'.bin', Dyn%Grid%pv)
endif
end subroutine dump_state

subroutine dump1(name, a)
character(len=*), intent(in) :: name
real, intent(in) :: a(:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='replace'); write(u) a; close(u)
end subroutine dump1
subroutine dump2(name, a)
character(len=*), intent(in) :: name
real, intent(in) :: a(:,:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='replace'); write(u) a; close(u)
end subroutine dump2
subroutine dumpc(name, a)
character(len=*), intent(in) :: name
complex, intent(in) :: a(:,:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='replace'); write(u) a; close(u)
end subroutine dumpc

end program ref_shallow_harness
