! oracle/ref_harness.F90 -- TEST INFRASTRUCTURE ONLY (never linked into the product).
!
! Driver program (ours) that calls ONLY the public module API of the reference's spectral core,
! compiled in place from /root/reference by oracle/build_ref.py, and dumps raw little-endian fp64
! arrays (Fortran order) to the current directory.  It follows the call order of the reference's
! per-step driver (src/atmos_spectral/driver/solo/atmosphere.F90:120-352, Held-Suarez branch) and
! of its main program's initialisation (src/atmos_solo/atmos_model.F90:148-260); it does not call
! the *_end routines because those write netCDF restarts (netCDF is absent in this image).
!
! Control: ./harness.nml (namelist &harness_nml) in the run directory, next to the reference's own
! input.nml / field_table / diag_table.
!   mode = 'run'     : cold start, nsteps steps; dumps ug,vg,tg,psg (+ re-derived spectral state)
!                      after the steps listed in dump_steps; prints wall time per step.
!   mode = 'kernels' : after init, reads in_*.bin written by oracle/make_golden.py and applies
!                      individual public routines (transforms, spectral operators, implicit
!                      correction, damping, pressure variables, vertical advection, hs_forcing).
program ref_harness

use constants_mod,         only: constants_init, pi
use fms_mod,               only: fms_init
use time_manager_mod,      only: time_type, set_time, set_calendar_type, NO_CALENDAR, operator(+)
use field_manager_mod,     only: MODEL_ATMOS
use tracer_manager_mod,    only: register_tracers, get_number_tracers
use diag_manager_mod,      only: diag_manager_init
use tracer_type_mod,       only: tracer_type
use spectral_dynamics_mod, only: spectral_dynamics_init, spectral_dynamics, get_num_levels, &
                                 get_initial_fields, get_surf_geopotential, get_pk_bk, get_axis_id
use transforms_mod,        only: get_grid_domain, get_spec_domain, get_deg_lon, get_deg_lat, &
                                 get_grid_boundaries, get_sin_lat, get_wts_lat, &
                                 trans_grid_to_spherical, trans_spherical_to_grid, &
                                 vor_div_from_uv_grid, uv_grid_from_vor_div, horizontal_advection, &
                                 compute_laplacian, compute_gradient_cos, compute_ucos_vcos, &
                                 compute_vor_div, area_weighted_global_mean, &
                                 compute_legendre, compute_gaussian, &
                                 trans_spherical_to_fourier, trans_fourier_to_spherical, &
                                 trans_grid_to_fourier, trans_fourier_to_grid, &
                                 get_eigen_laplacian
use press_and_geopot_mod,  only: compute_pressures_and_heights, pressure_variables, compute_geopotential
use hs_forcing_mod,        only: hs_forcing_init, hs_forcing
use implicit_mod,          only: implicit_correction
use spectral_damping_mod,  only: compute_spectral_damping, compute_spectral_damping_vor, &
                                 compute_spectral_damping_div
use vert_advection_mod,    only: vert_advection, SECOND_CENTERED, ADVECTIVE_FORM, FINITE_VOLUME_PARABOLIC
use fv_advection_mod,      only: a_grid_horiz_advection
use global_integral_mod,   only: mass_weighted_global_integral
use leapfrog_mod,          only: leapfrog, leapfrog_2level_A, leapfrog_2level_B

implicit none

character(len=16) :: mode = 'run'
integer :: nsteps = 1, dt_atmos = 600
integer, dimension(64) :: dump_steps = -1
logical :: dump_tables = .true.
! developed-state restart fixture: the module keeps its spectral arrays (and its Robert-filtered tracer levels) private, so from step
! track_from on the driver follows the filter itself through public routines -- s(k) = the spectral state of the grid fields of
! step k (vor_div_from_uv_grid, trans_grid_to_spherical: equal to the module's own to roundoff), filtered level
! F(k) = (s(k) + robert*(F(k-1) - 2 s(k))) + robert*s(k+1)  (leapfrog_2level_A/B, leapfrog.F90:58-105 with raw_filter_coeff = 1);
! the start value F = s is forgotten as robert**steps (0.04**16 ~ 4e-23) -- and at step dump_full_at writes BOTH time levels (rs_*).
integer :: track_from = -1, dump_full_at = -1
real    :: robert_coeff = 0.04      ! spectral_dynamics_nml's value (the module does not export it)
! the trip test's criterion (exp/test_cases/trip_test/trip_test_functions.py:173-189, 286-297): time means of ps, ucomp, vcomp, temp, vor, div over every
! mean_every steps, as spectral_diagnostics (spectral_dynamics.F90:1709-1867, called with the FUTURE level at the end of atmosphere: atmosphere.F90:344)
! and diag_manager (sum of the samples, divided by their number when the interval ends) form them; netCDF is absent here, so the driver accumulates itself.
! vor and div on the grid are module-private: they are re-formed from the level's winds through public routines (equal to the module's own to roundoff).
integer :: mean_every = 0
namelist /harness_nml/ mode, nsteps, dt_atmos, dump_steps, dump_tables, track_from, dump_full_at, robert_coeff, mean_every

type(time_type) :: Time, Time_step, Time_next
type(tracer_type), allocatable, dimension(:) :: tracer_attributes
integer :: ntrace, ntprog, ntdiag, ntfamily, num_tracers, nhum
logical :: dry_model
integer :: is, ie, js, je, ms, me, ns, ne, num_levels, nlon, nlat
integer :: previous, current, future, i, j, k, istep, unit, idump
real    :: delta_t, dt_real
integer(kind=8) :: c0, c1, crate
real(kind=8) :: t_loop

real, allocatable, dimension(:,:,:,:)   :: p_half, p_full, z_half, z_full, ug, vg, tg
real, allocatable, dimension(:,:,:,:,:) :: grid_tracers
real, allocatable, dimension(:,:,:)     :: psg, wg_full, dt_ug, dt_vg, dt_tg
real, allocatable, dimension(:,:,:,:)   :: dt_tracers
real, allocatable, dimension(:,:)       :: dt_psg, surf_geopotential
real, allocatable, dimension(:)         :: deg_lon, deg_lat, rad_lonb, rad_latb, pk, bk, sin_lat, wts_lat
real, allocatable, dimension(:,:)       :: rad_lon_2d, rad_lat_2d, rad_lonb_2d, rad_latb_2d
complex, allocatable, dimension(:,:,:)  :: sc_vor, sc_div, sc_t, sn_vor, sn_div, sn_t, f_vor, f_div, f_t
complex, allocatable, dimension(:,:)    :: sc_lp, sn_lp, f_lp
real, allocatable, dimension(:,:,:,:)   :: f_tr
real, allocatable, dimension(:,:,:)     :: mn_u, mn_v, mn_t, mn_vor, mn_div, tmp_vor, tmp_div
real, allocatable, dimension(:,:)       :: mn_ps
complex, allocatable, dimension(:,:,:)  :: tmp_vs, tmp_ds
integer :: mean_count = 0

open(newunit=unit, file='harness.nml', status='old', action='read')
read(unit, nml=harness_nml)
close(unit)

! ---- initialisation, order of atmos_model.F90:148-350 -----------------------------------------
call fms_init()
call constants_init()
call register_tracers(MODEL_ATMOS, ntrace, ntprog, ntdiag, ntfamily)
call set_calendar_type(NO_CALENDAR)
call diag_manager_init()
Time      = set_time(0, 0)
Time_step = set_time(dt_atmos, 0)
dt_real   = real(dt_atmos)

! ---- atmosphere_init, atmosphere.F90:151-266 ----------------------------------------------------
call get_number_tracers(MODEL_ATMOS, num_prog=num_tracers)
allocate(tracer_attributes(num_tracers))
call spectral_dynamics_init(Time, Time_step, tracer_attributes, dry_model, nhum)
call get_grid_domain(is, ie, js, je)
call get_spec_domain(ms, me, ns, ne)
call get_num_levels(num_levels)
nlon = ie-is+1; nlat = je-js+1

allocate(p_half(is:ie,js:je,num_levels+1,2), z_half(is:ie,js:je,num_levels+1,2))
allocate(p_full(is:ie,js:je,num_levels,2),   z_full(is:ie,js:je,num_levels,2))
allocate(wg_full(is:ie,js:je,num_levels), psg(is:ie,js:je,2))
allocate(ug(is:ie,js:je,num_levels,2), vg(is:ie,js:je,num_levels,2), tg(is:ie,js:je,num_levels,2))
allocate(grid_tracers(is:ie,js:je,num_levels,2,num_tracers))
allocate(dt_psg(is:ie,js:je), dt_ug(is:ie,js:je,num_levels), dt_vg(is:ie,js:je,num_levels))
allocate(dt_tg(is:ie,js:je,num_levels), dt_tracers(is:ie,js:je,num_levels,num_tracers))
allocate(deg_lon(is:ie), deg_lat(js:je), rad_lon_2d(is:ie,js:je), rad_lat_2d(is:ie,js:je))
allocate(rad_lonb_2d(is:ie+1,js:je+1), rad_latb_2d(is:ie+1,js:je+1), rad_lonb(is:ie+1), rad_latb(js:je+1))
allocate(surf_geopotential(is:ie,js:je), pk(num_levels+1), bk(num_levels+1), sin_lat(js:je), wts_lat(js:je))
p_half=0.; z_half=0.; p_full=0.; z_full=0.; wg_full=0.; psg=0.; ug=0.; vg=0.; tg=0.; grid_tracers=0.
dt_psg=0.; dt_ug=0.; dt_vg=0.; dt_tg=0.; dt_tracers=0.

call get_surf_geopotential(surf_geopotential)
previous = 1; current = 1
call get_initial_fields(ug(:,:,:,1), vg(:,:,:,1), tg(:,:,:,1), psg(:,:,1), grid_tracers(:,:,:,1,:))
if(dry_model) then
  call compute_pressures_and_heights(tg(:,:,:,current), psg(:,:,current), surf_geopotential, &
       z_full(:,:,:,current), z_half(:,:,:,current), p_full(:,:,:,current), p_half(:,:,:,current))
else
  call compute_pressures_and_heights(tg(:,:,:,current), psg(:,:,current), surf_geopotential, &
       z_full(:,:,:,current), z_half(:,:,:,current), p_full(:,:,:,current), p_half(:,:,:,current), &
       grid_tracers(:,:,:,current,nhum))
endif
call get_deg_lon(deg_lon)
do i=is,ie
  rad_lon_2d(i,:) = deg_lon(i)*pi/180.
enddo
call get_deg_lat(deg_lat)
do j=js,je
  rad_lat_2d(:,j) = deg_lat(j)*pi/180.
enddo
call get_grid_boundaries(rad_lonb, rad_latb)
do i=is,ie+1
  rad_lonb_2d(i,:) = rad_lonb(i)
enddo
do j=js,je+1
  rad_latb_2d(:,j) = rad_latb(j)
enddo
call hs_forcing_init(get_axis_id(), Time, rad_lonb_2d, rad_latb_2d, rad_lat_2d)

call get_pk_bk(pk, bk)
call get_sin_lat(sin_lat)
call get_wts_lat(wts_lat)

if(dump_tables) then
  call dump1('tab_pk.bin', pk);          call dump1('tab_bk.bin', bk)
  call dump1('tab_sin_lat.bin', sin_lat); call dump1('tab_wts_lat.bin', wts_lat)
  call dump1('tab_deg_lat.bin', deg_lat); call dump1('tab_deg_lon.bin', deg_lon)
  call dump1('tab_rad_latb.bin', rad_latb)
  call dump_legendre()
endif

if(trim(mode) == 'run') then
  call dump_state(0)
  idump = 1
  t_loop = 0.
  call system_clock(count_rate=crate)
  do istep = 1, nsteps
    call system_clock(c0)
    call one_step()
    call system_clock(c1)
    t_loop = t_loop + real(c1-c0,8)/real(crate,8)
    if(track_from >= 0 .and. istep >= track_from) call track_filter(istep == track_from)
    if(istep == dump_full_at) call dump_both_levels()
    if(any(dump_steps == istep)) call dump_state(istep)
    if(mean_every > 0) call accumulate_means(istep)
  enddo
  write(*,'(a,i8,a,f12.6,a,f12.6)') 'REF_TIMING steps=', nsteps, ' seconds=', t_loop, ' ms_per_step=', 1.e3*t_loop/max(nsteps,1)
  write(*,'(a,3es24.16)') 'REF_STATE Tmin,Tmax,maxabsU=', minval(tg(:,:,:,current)), maxval(tg(:,:,:,current)), maxval(abs(ug(:,:,:,current)))
else if(trim(mode) == 'kernels') then
  call run_kernels()
else
  write(*,*) 'unknown mode ', trim(mode)
  stop 2
endif

contains

!--------------------------------------------------------------------------------------------------
subroutine one_step()
! atmosphere.F90:286-349 (Held-Suarez branch), without spectral_diagnostics
dt_ug = 0.0; dt_vg = 0.0; dt_tg = 0.0; dt_psg = 0.0; dt_tracers = 0.0
if(current == previous) then
  delta_t = dt_real
else
  delta_t = 2*dt_real
endif
Time_next = Time + Time_step
call hs_forcing(1, ie-is+1, 1, je-js+1, delta_t, Time_next, rad_lon_2d, rad_lat_2d, &
                p_half(:,:,:,current ),       p_full(:,:,:,current   ), &
                    ug(:,:,:,previous),           vg(:,:,:,previous  ), &
                    tg(:,:,:,previous), grid_tracers(:,:,:,previous,:), &
                    ug(:,:,:,previous),           vg(:,:,:,previous  ), &
                    tg(:,:,:,previous), grid_tracers(:,:,:,previous,:), &
                 dt_ug(:,:,:         ),        dt_vg(:,:,:           ), &
                 dt_tg(:,:,:         ),   dt_tracers(:,:,:,:), z_full(:,:,:,current))
if(previous == current) then
  future = 3 - current
else
  future = previous
endif
call spectral_dynamics(Time, psg(:,:,future), ug(:,:,:,future), vg(:,:,:,future), &
                       tg(:,:,:,future), tracer_attributes, grid_tracers(:,:,:,:,:), future, &
                       dt_psg, dt_ug, dt_vg, dt_tg, dt_tracers, wg_full, &
                       p_full(:,:,:,current), p_half(:,:,:,current), z_full(:,:,:,current))
if(dry_model) then
  call compute_pressures_and_heights(tg(:,:,:,future), psg(:,:,future), surf_geopotential, &
       z_full(:,:,:,future), z_half(:,:,:,future), p_full(:,:,:,future), p_half(:,:,:,future))
else
  call compute_pressures_and_heights(tg(:,:,:,future), psg(:,:,future), surf_geopotential, &
       z_full(:,:,:,future), z_half(:,:,:,future), p_full(:,:,:,future), p_half(:,:,:,future), &
       grid_tracers(:,:,:,future,nhum))
endif
previous = current
current  = future
Time = Time_next
end subroutine one_step

!--------------------------------------------------------------------------------------------------
subroutine spec_of_current(vs, ds, tts, lps)
complex, intent(out) :: vs(ms:,ns:,:), ds(ms:,ns:,:), tts(ms:,ns:,:), lps(ms:,ns:)
real, allocatable :: lnpsg(:,:)
allocate(lnpsg(is:ie,js:je))
call vor_div_from_uv_grid(ug(:,:,:,current), vg(:,:,:,current), vs, ds)
call trans_grid_to_spherical(tg(:,:,:,current), tts)
lnpsg = log(psg(:,:,current))
call trans_grid_to_spherical(lnpsg, lps)
deallocate(lnpsg)
end subroutine spec_of_current

subroutine track_filter(first)
logical, intent(in) :: first
integer :: ntr
real :: rq
if(first) then
  allocate(sc_vor(ms:me,ns:ne,num_levels), sc_div(ms:me,ns:ne,num_levels), sc_t(ms:me,ns:ne,num_levels), sc_lp(ms:me,ns:ne))
  allocate(sn_vor(ms:me,ns:ne,num_levels), sn_div(ms:me,ns:ne,num_levels), sn_t(ms:me,ns:ne,num_levels), sn_lp(ms:me,ns:ne))
  allocate(f_vor(ms:me,ns:ne,num_levels), f_div(ms:me,ns:ne,num_levels), f_t(ms:me,ns:ne,num_levels), f_lp(ms:me,ns:ne))
  allocate(f_tr(is:ie,js:je,num_levels,num_tracers))
  call spec_of_current(sc_vor, sc_div, sc_t, sc_lp)
  f_vor = sc_vor; f_div = sc_div; f_t = sc_t; f_lp = sc_lp
  f_tr = grid_tracers(:,:,:,current,:)
  return
endif
! the step just taken made s(k+1) (= current now); sc_* is s(k) (= previous now), f_* the filtered level k-1
call spec_of_current(sn_vor, sn_div, sn_t, sn_lp)
f_vor = sc_vor + robert_coeff*(f_vor - 2.0*sc_vor); f_vor = f_vor + robert_coeff*sn_vor
f_div = sc_div + robert_coeff*(f_div - 2.0*sc_div); f_div = f_div + robert_coeff*sn_div
f_t   = sc_t   + robert_coeff*(f_t   - 2.0*sc_t  ); f_t   = f_t   + robert_coeff*sn_t
f_lp  = sc_lp  + robert_coeff*(f_lp  - 2.0*sc_lp ); f_lp  = f_lp  + robert_coeff*sn_lp
do ntr = 1, num_tracers
  rq = tracer_attributes(ntr)%robert_coeff
  f_tr(:,:,:,ntr) = grid_tracers(:,:,:,previous,ntr) + rq*(f_tr(:,:,:,ntr) - 2.0*grid_tracers(:,:,:,previous,ntr))
  f_tr(:,:,:,ntr) = f_tr(:,:,:,ntr) + rq*grid_tracers(:,:,:,current,ntr)
enddo
sc_vor = sn_vor; sc_div = sn_div; sc_t = sn_t; sc_lp = sn_lp
end subroutine track_filter

subroutine accumulate_means(n)
! after one_step the new level is `current`: the level spectral_diagnostics is handed (atmosphere.F90:344)
integer, intent(in) :: n
character(len=8) :: tag
if(.not.allocated(mn_u)) then
  allocate(mn_u(is:ie,js:je,num_levels), mn_v(is:ie,js:je,num_levels), mn_t(is:ie,js:je,num_levels), mn_vor(is:ie,js:je,num_levels), &
           mn_div(is:ie,js:je,num_levels), tmp_vor(is:ie,js:je,num_levels), tmp_div(is:ie,js:je,num_levels), mn_ps(is:ie,js:je))
  allocate(tmp_vs(ms:me,ns:ne,num_levels), tmp_ds(ms:me,ns:ne,num_levels))
  mn_u = 0.; mn_v = 0.; mn_t = 0.; mn_vor = 0.; mn_div = 0.; mn_ps = 0.; mean_count = 0
endif
call vor_div_from_uv_grid(ug(:,:,:,current), vg(:,:,:,current), tmp_vs, tmp_ds)
call trans_spherical_to_grid(tmp_vs, tmp_vor)
call trans_spherical_to_grid(tmp_ds, tmp_div)
mn_u = mn_u + ug(:,:,:,current); mn_v = mn_v + vg(:,:,:,current); mn_t = mn_t + tg(:,:,:,current)
mn_vor = mn_vor + tmp_vor; mn_div = mn_div + tmp_div; mn_ps = mn_ps + psg(:,:,current)
mean_count = mean_count + 1
if(mod(n, mean_every) /= 0) return
write(tag,'(i6.6)') n
call dump3('mean_ucomp_'//trim(tag)//'.bin', mn_u/real(mean_count));  call dump3('mean_vcomp_'//trim(tag)//'.bin', mn_v/real(mean_count))
call dump3('mean_temp_'//trim(tag)//'.bin', mn_t/real(mean_count));   call dump3('mean_vor_'//trim(tag)//'.bin', mn_vor/real(mean_count))
call dump3('mean_div_'//trim(tag)//'.bin', mn_div/real(mean_count));  call dump2('mean_ps_'//trim(tag)//'.bin', mn_ps/real(mean_count))
mn_u = 0.; mn_v = 0.; mn_t = 0.; mn_vor = 0.; mn_div = 0.; mn_ps = 0.; mean_count = 0
end subroutine accumulate_means

subroutine dump_both_levels()
! what a restart of spectral_dynamics_mod + atmosphere_mod holds (spectral_dynamics.F90:1502-1531, atmosphere.F90:362-375)
integer :: ntr
character(len=1) :: trno
call dumpc3('rs_vors_cur.bin', sc_vor);  call dumpc3('rs_divs_cur.bin', sc_div)
call dumpc3('rs_ts_cur.bin', sc_t);      call dumpc2('rs_lnps_cur.bin', sc_lp)
call dumpc3('rs_vors_prev.bin', f_vor);  call dumpc3('rs_divs_prev.bin', f_div)
call dumpc3('rs_ts_prev.bin', f_t);      call dumpc2('rs_lnps_prev.bin', f_lp)
call dump3('rs_ug_cur.bin', ug(:,:,:,current));   call dump3('rs_ug_prev.bin', ug(:,:,:,previous))
call dump3('rs_vg_cur.bin', vg(:,:,:,current));   call dump3('rs_vg_prev.bin', vg(:,:,:,previous))
call dump3('rs_tg_cur.bin', tg(:,:,:,current));   call dump3('rs_tg_prev.bin', tg(:,:,:,previous))
call dump2('rs_psg_cur.bin', psg(:,:,current));   call dump2('rs_psg_prev.bin', psg(:,:,previous))
call dump3('rs_wg_full.bin', wg_full)
do ntr = 1, num_tracers
  write(trno,'(i1)') ntr
  call dump3('rs_tr'//trno//'_cur.bin', grid_tracers(:,:,:,current,ntr))          ! the dynamics' and atmosphere_mod's newest level
  call dump3('rs_tr'//trno//'_prev_atm.bin', grid_tracers(:,:,:,previous,ntr))    ! atmosphere_mod's (unfiltered) previous level
  call dump3('rs_tr'//trno//'_prev_filt.bin', f_tr(:,:,:,ntr))                    ! the dynamics' Robert-filtered previous level
enddo
write(*,'(a,4es16.8)') 'REF_DEVELOPED max|u|,max|v|,Tmin,Tmax=', maxval(abs(ug(:,:,:,current))), maxval(abs(vg(:,:,:,current))), &
     minval(tg(:,:,:,current)), maxval(tg(:,:,:,current))
end subroutine dump_both_levels

!--------------------------------------------------------------------------------------------------
subroutine dump_state(n)
integer, intent(in) :: n
character(len=8) :: tag
complex, allocatable, dimension(:,:,:) :: vors, divs, ts
complex, allocatable, dimension(:,:)   :: lnps
real,    allocatable, dimension(:,:)   :: lnpsg
integer :: ntr
character(len=1) :: trno
write(tag,'(i6.6)') n
call dump3('st_ug_'//trim(tag)//'.bin', ug(:,:,:,current))
call dump3('st_vg_'//trim(tag)//'.bin', vg(:,:,:,current))
call dump3('st_tg_'//trim(tag)//'.bin', tg(:,:,:,current))
call dump2('st_psg_'//trim(tag)//'.bin', psg(:,:,current))
call dump3('st_wg_full_'//trim(tag)//'.bin', wg_full)
call dump3('st_p_full_'//trim(tag)//'.bin', p_full(:,:,:,current))
call dump3('st_z_full_'//trim(tag)//'.bin', z_full(:,:,:,current))
do ntr = 1, num_tracers
  write(trno,'(i1)') ntr
  call dump3('st_tr'//trno//'_'//trim(tag)//'.bin', grid_tracers(:,:,:,current,ntr))
enddo
! spectral state re-derived through the public API (the module-private arrays have no getter)
allocate(vors(ms:me,ns:ne,num_levels), divs(ms:me,ns:ne,num_levels), ts(ms:me,ns:ne,num_levels))
allocate(lnps(ms:me,ns:ne), lnpsg(is:ie,js:je))
call vor_div_from_uv_grid(ug(:,:,:,current), vg(:,:,:,current), vors, divs)
call trans_grid_to_spherical(tg(:,:,:,current), ts)
lnpsg = log(psg(:,:,current))
call trans_grid_to_spherical(lnpsg, lnps)
call dumpc3('st_vors_'//trim(tag)//'.bin', vors)
call dumpc3('st_divs_'//trim(tag)//'.bin', divs)
call dumpc3('st_ts_'//trim(tag)//'.bin', ts)
call dumpc2('st_lnps_'//trim(tag)//'.bin', lnps)
deallocate(vors, divs, ts, lnps, lnpsg)
end subroutine dump_state

!--------------------------------------------------------------------------------------------------
subroutine dump_legendre()
integer :: nhem, nf, nsph
real, allocatable :: sin_hem(:), wts_hem(:), leg(:,:,:), eig(:,:)
nhem = nlat/2; nf = me-ms; nsph = ne-ns
allocate(sin_hem(nhem), wts_hem(nhem), leg(0:nf,0:nsph,nhem), eig(0:nf,0:nsph))
call compute_gaussian(sin_hem, wts_hem, nhem)
call compute_legendre(leg, nf, 1, nsph, sin_hem, nhem)
call get_eigen_laplacian(eig)
call dump1('tab_sin_hem.bin', sin_hem)
call dump1('tab_wts_hem.bin', wts_hem)
call dump3('tab_legendre.bin', leg)
call dump2('tab_eigen_laplacian.bin', eig)
end subroutine dump_legendre

!--------------------------------------------------------------------------------------------------
subroutine run_kernels()
! Each block reads inputs written by oracle/make_golden.py and applies one public routine.
complex, allocatable, dimension(:,:,:) :: sa, sb, sc, sd
complex, allocatable, dimension(:,:)   :: s2a, s2b
complex, allocatable, dimension(:,:,:,:) :: s4a, s4b
complex, allocatable, dimension(:,:,:) :: s3lnps
real,    allocatable, dimension(:,:,:) :: ga, gb, gc, gd, ge, w, dp
real,    allocatable, dimension(:,:,:,:) :: tr4, trdt
real,    allocatable, dimension(:,:)   :: g2a
complex, allocatable, dimension(:,:,:,:) :: fs
complex, allocatable, dimension(:,:,:)   :: fg
integer :: kk
real :: dtk, gm

kk = num_levels
allocate(sa(ms:me,ns:ne,kk), sb(ms:me,ns:ne,kk), sc(ms:me,ns:ne,kk), sd(ms:me,ns:ne,kk))
allocate(ga(is:ie,js:je,kk), gb(is:ie,js:je,kk), gc(is:ie,js:je,kk), gd(is:ie,js:je,kk), ge(is:ie,js:je,kk))
allocate(s2a(ms:me,ns:ne), s2b(ms:me,ns:ne), g2a(is:ie,js:je))

! --- transforms ---
call readc3('in_spec_a.bin', sa)
call readc3('in_spec_b.bin', sb)
call trans_spherical_to_grid(sa, ga)
call dump3('out_s2g_a.bin', ga)
call trans_grid_to_spherical(ga, sc)
call dumpc3('out_g2s_s2g_a.bin', sc)
call read3('in_grid_a.bin', ga)
call read3('in_grid_b.bin', gb)
call trans_grid_to_spherical(ga, sc)
call dumpc3('out_g2s_a.bin', sc)
call trans_grid_to_spherical(ga, sc, do_truncation=.false.)
call dumpc3('out_g2s_a_notrunc.bin', sc)
! Legendre / Fourier stages separately
allocate(fs(ms:me, nlat, kk, 1), fg(0:nlon/2, nlat, kk))
call trans_spherical_to_fourier(sa, fs)
call dumpc3('out_s2f_a.bin', fs(:,:,:,1))
fg = trans_grid_to_fourier(ga)
call dumpc3('out_g2f_a.bin', fg)
! --- vor/div <-> u,v ---
call vor_div_from_uv_grid(ga, gb, sc, sd)
call dumpc3('out_vor_from_uv.bin', sc)
call dumpc3('out_div_from_uv.bin', sd)
call uv_grid_from_vor_div(sa, sb, gc, gd)
call dump3('out_u_from_vd.bin', gc)
call dump3('out_v_from_vd.bin', gd)
! --- spectral operators ---
sc = compute_laplacian(sa)
call dumpc3('out_laplacian_a.bin', sc)
call compute_gradient_cos(sa, sc, sd)
call dumpc3('out_gradcos_dx_a.bin', sc)
call dumpc3('out_gradcos_dy_a.bin', sd)
call compute_ucos_vcos(sa, sb, sc, sd)
call dumpc3('out_ucos.bin', sc)
call dumpc3('out_vcos.bin', sd)
call compute_vor_div(sa, sb, sc, sd)
call dumpc3('out_vor_from_ucos.bin', sc)
call dumpc3('out_div_from_ucos.bin', sd)
! --- horizontal advection: tendency(inout) -= u dT/dx + v dT/dy ---
gc = 0.
call horizontal_advection(sa, ga, gb, gc)
call dump3('out_hadv.bin', gc)
! --- global means ---
gm = area_weighted_global_mean(ga(:,:,1))
call dump1('out_gmean.bin', (/gm/))
call read2('in_ps.bin', g2a)
gm = mass_weighted_global_integral(ga, g2a)
call dump1('out_mwgi.bin', (/gm/))
! --- pressure variables / geopotential (press_and_geopot.F90) ---
allocate(w(is:ie,js:je,kk+1), dp(is:ie,js:je,kk+1))
call pressure_variables(w, dp, gc, gd, g2a)     ! p_half, ln_p_half, p_full, ln_p_full
call dump3('out_p_half.bin', w);   call dump3('out_ln_p_half.bin', dp)
call dump3('out_p_full.bin', gc);  call dump3('out_ln_p_full.bin', gd)
call read3('in_temp.bin', ge)
deallocate(w); allocate(w(is:ie,js:je,kk+1))
call compute_geopotential(ge, dp, gd, surf_geopotential, gc, w)
call dump3('out_geopot_full.bin', gc); call dump3('out_geopot_half.bin', w)
! --- hs_forcing on (u=ga, v=gb, T=ge) with pressures from in_ps ---
call pressure_variables(w, dp, gc, gd, g2a)
allocate(tr4(is:ie,js:je,kk,num_tracers), trdt(is:ie,js:je,kk,num_tracers))
tr4 = 0.; trdt = 0.
dt_ug = 0.; dt_vg = 0.; dt_tg = 0.
dtk = 2*dt_real
call hs_forcing(1, nlon, 1, nlat, dtk, Time, rad_lon_2d, rad_lat_2d, w, gc, ga, gb, ge, tr4, &
                ga, gb, ge, tr4, dt_ug, dt_vg, dt_tg, trdt, gd)
call dump3('out_hs_dt_u.bin', dt_ug); call dump3('out_hs_dt_v.bin', dt_vg); call dump3('out_hs_dt_t.bin', dt_tg)
if(num_tracers > 0) call dump3('out_hs_dt_tr.bin', trdt(:,:,:,1))
! --- vertical advection, second centred / advective form (vert_advection.F90:185-193,467-470) ---
call read3('in_wg.bin', w)      ! (lon,lat,kk+1)
do k=1,kk
  dp(:,:,k) = (pk(k+1)-pk(k)) + (bk(k+1)-bk(k))*g2a
enddo
call vert_advection(dtk, w, dp(:,:,1:kk), ge, gc, scheme=SECOND_CENTERED, form=ADVECTIVE_FORM)
call dump3('out_vadv.bin', gc)
! --- PPM vertical advection of a tracer-like field (vert_advection.F90:301-438) ---
call read3('in_q.bin', ga)
call vert_advection(dtk, w, dp(:,:,1:kk), ga, gc, scheme=FINITE_VOLUME_PARABOLIC, form=ADVECTIVE_FORM)
call dump3('out_vadv_ppm.bin', gc)
! --- van Leer horizontal advection on the sphere (fv_advection.F90:126-207), moderate and >1 Courant numbers ---
call read3('in_grid_a.bin', gd); call read3('in_grid_b.bin', gb)
gc = 0.
call a_grid_horiz_advection(gd, gb, ga, dtk, gc)
call dump3('out_hadv_fv.bin', gc)
gc = 0.
call a_grid_horiz_advection(gd, gb, ga, 40.*dtk, gc)
call dump3('out_hadv_fv_bigcfl.bin', gc)
call read3('in_grid_a.bin', ga)
! --- implicit correction (implicit.F90:241-286) ---
allocate(s4a(ms:me,ns:ne,kk,2), s4b(ms:me,ns:ne,kk,2), s3lnps(ms:me,ns:ne,2))
call readc3('in_spec_a.bin', s4a(:,:,:,1)); call readc3('in_spec_b.bin', s4a(:,:,:,2))
call readc3('in_spec_c.bin', s4b(:,:,:,1)); call readc3('in_spec_d.bin', s4b(:,:,:,2))
call readc2('in_spec2_a.bin', s3lnps(:,:,1)); call readc2('in_spec2_b.bin', s3lnps(:,:,2))
call readc3('in_spec_e.bin', sc); call readc3('in_spec_f.bin', sd); call readc2('in_spec2_c.bin', s2a)
call implicit_correction(sc, sd, s2a, s4a, s4b, s3lnps, dtk, 1, 2)
call dumpc3('out_impl_dt_divs.bin', sc); call dumpc3('out_impl_dt_ts.bin', sd); call dumpc2('out_impl_dt_lnps.bin', s2a)
! --- spectral damping (spectral_damping.F90:172-291) ---
call readc3('in_spec_e.bin', sc)
call compute_spectral_damping_vor(sa, sc, dtk); call dumpc3('out_damp_vor.bin', sc)
call readc3('in_spec_e.bin', sc)
call compute_spectral_damping_div(sa, sc, dtk); call dumpc3('out_damp_div.bin', sc)
call readc3('in_spec_e.bin', sc)
call compute_spectral_damping(sa, sc, dtk);     call dumpc3('out_damp.bin', sc)
! --- leapfrog + Robert filter (leapfrog.F90:58-105) ---
call readc3('in_spec_e.bin', sc)
call leapfrog_2level_A(s4a, sc, 1, 2, 1, dtk, 0.04, 1.0, sd)
call leapfrog_2level_B(s4a, sd, 2, 1, 0.04, 1.0)
call dumpc3('out_leap_l1.bin', s4a(:,:,:,1)); call dumpc3('out_leap_l2.bin', s4a(:,:,:,2))
end subroutine run_kernels

!--------------------------------------------------------------------------------------------------
subroutine dump1(name, a)
character(len=*), intent(in) :: name
real, intent(in) :: a(:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='replace')
write(u) a
close(u)
end subroutine dump1
subroutine dump2(name, a)
character(len=*), intent(in) :: name
real, intent(in) :: a(:,:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='replace')
write(u) a
close(u)
end subroutine dump2
subroutine dump3(name, a)
character(len=*), intent(in) :: name
real, intent(in) :: a(:,:,:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='replace')
write(u) a
close(u)
end subroutine dump3
subroutine dumpc2(name, a)
character(len=*), intent(in) :: name
complex, intent(in) :: a(:,:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='replace')
write(u) a
close(u)
end subroutine dumpc2
subroutine dumpc3(name, a)
character(len=*), intent(in) :: name
complex, intent(in) :: a(:,:,:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='replace')
write(u) a
close(u)
end subroutine dumpc3
subroutine read2(name, a)
character(len=*), intent(in) :: name
real, intent(out) :: a(:,:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='old')
read(u) a
close(u)
end subroutine read2
subroutine read3(name, a)
character(len=*), intent(in) :: name
real, intent(out) :: a(:,:,:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='old')
read(u) a
close(u)
end subroutine read3
subroutine readc2(name, a)
character(len=*), intent(in) :: name
complex, intent(out) :: a(:,:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='old')
read(u) a
close(u)
end subroutine readc2
subroutine readc3(name, a)
character(len=*), intent(in) :: name
complex, intent(out) :: a(:,:,:)
integer :: u
open(newunit=u, file=name, access='stream', form='unformatted', status='old')
read(u) a
close(u)
end subroutine readc3

end program ref_harness
