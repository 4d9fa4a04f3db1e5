! oracle/ref_moist_harness.F90 -- TEST INFRASTRUCTURE ONLY (never linked into the product).
!
! Driver program (ours) for the reference's moist configuration (config 3 of BASELINE.json: Frierson grey-radiation
! aquaplanet).  It calls only public module procedures, in the order of the reference's per-step driver with
! idealized_moist_model = .true. (src/atmos_spectral/driver/solo/atmosphere.F90:120-352):
!   idealized_moist_phys (driver/solo/idealized_moist_phys.F90:819-1395) -> spectral_dynamics -> pressures/heights,
! and dumps raw little-endian fp64 arrays (Fortran order).  Compiled in place from /root/reference by oracle/build_ref.py.
!
! Control: ./harness.nml (&harness_nml) next to the reference's own input.nml / field_table / diag_table.
!   mode = 'kernels': nsteps steps as in 'run', then the routines of the column chain are called one by one, in the order
!                   and with the arguments of idealized_moist_phys (:862-1340), on the state reached; inputs (k_in_*) and
!                   every output (k_*) are dumped.  The surface temperature, private to the reference's driver, is replaced
!                   by one this harness chooses (k_in_t_surf) -- the routines take it as an argument.
!   mode = 'run'  : cold start, nsteps steps; after the steps in dump_steps dumps ug, vg, tg, psg, sphum of the new level
!                   (st_*), and for the steps in phys_steps the inputs and outputs of that step's idealized_moist_phys call:
!                   ph_in_* = u, v, T, q at `previous` and `current`, p_half/p_full/z_half/z_full at both levels,
!                   ph_dt_* = the physics tendencies dt_ug, dt_vg, dt_tg, dt_tracers it returned.
!                   track_from / dump_full_at (round 5, as in ref_harness.F90): from step track_from on the driver follows the Robert filter
!                   itself through public routines -- s(k) = the spectral state of step k's grid fields, F(k) = (s(k) + r (F(k-1) - 2 s(k))) +
!                   r s(k+1) (leapfrog.F90:58-105), likewise the tracer's filtered level -- and at step dump_full_at writes BOTH time levels
!                   (rs_*) plus the mixed layer's surface temperature.  t_surf is module-private data of idealized_moist_phys_mod with no
!                   getter; it is READ (never written) through the object file's symbol by oracle/ref_peek.c.
program ref_moist_harness

use iso_c_binding,         only: c_ptr, c_f_pointer

use constants_mod,         only: constants_init, pi
use fms_mod,               only: fms_init
use time_manager_mod,      only: time_type, set_time, set_calendar_type, NO_CALENDAR, operator(+)
use field_manager_mod,     only: MODEL_ATMOS
use tracer_manager_mod,    only: register_tracers, get_number_tracers
use diag_manager_mod,      only: diag_manager_init
use tracer_type_mod,       only: tracer_type
use spectral_dynamics_mod, only: spectral_dynamics_init, spectral_dynamics, get_num_levels, &
                                 get_initial_fields, get_surf_geopotential, get_pk_bk, get_axis_id
use transforms_mod,        only: get_grid_domain, get_spec_domain, get_deg_lon, get_deg_lat, get_grid_boundaries, &
                                 vor_div_from_uv_grid, trans_grid_to_spherical
use press_and_geopot_mod,  only: compute_pressures_and_heights
use idealized_moist_phys_mod, only: idealized_moist_phys_init, idealized_moist_phys
use sat_vapor_pres_mod,    only: lookup_es, lookup_des
use qe_moist_convection_mod, only: qe_moist_convection
use lscale_cond_mod,       only: lscale_cond
use two_stream_gray_rad_mod, only: two_stream_gray_rad_down, two_stream_gray_rad_up
use surface_flux_mod,      only: surface_flux
use damping_driver_mod,    only: damping_driver
use vert_turb_driver_mod,  only: vert_turb_driver
use vert_diff_mod,         only: gcm_vert_diff_down, gcm_vert_diff_up, surf_diff_type
use mixed_layer_mod,       only: mixed_layer

implicit none

character(len=16) :: mode = 'run'
integer :: nsteps = 1, dt_atmos = 720
integer, dimension(64) :: dump_steps = -1, phys_steps = -1
integer :: track_from = -1, dump_full_at = -1
real    :: robert_coeff = 0.03            ! spectral_dynamics_nml's value in the run's input.nml (the module keeps it private)
namelist /harness_nml/ mode, nsteps, dt_atmos, dump_steps, phys_steps, track_from, dump_full_at, robert_coeff

interface
  function ref_peek_t_surf() bind(C, name='ref_peek_t_surf') result(p)
    import :: c_ptr
    type(c_ptr) :: p
  end function ref_peek_t_surf
end interface
complex, allocatable, dimension(:,:,:) :: sc_vor, sc_div, sc_t, sn_vor, sn_div, sn_t, f_vor, f_div, f_t
complex, allocatable, dimension(:,:)   :: sc_lp, sn_lp, f_lp
real,    allocatable, dimension(:,:,:,:) :: f_tr

type(time_type) :: Time, Time_step, Time_next
type(tracer_type), allocatable, dimension(:) :: tracer_attributes
integer :: ntrace, ntprog, ntdiag, ntfamily, num_tracers, nhum
logical :: dry_model
integer :: is, ie, js, je, ms, me, ns, ne, num_levels, nlon, nlat
integer :: previous, current, future, i, j, istep, unit
real    :: delta_t, dt_real
integer(kind=8) :: c0, c1, crate
real(kind=8) :: t_loop

real, allocatable, dimension(:,:,:,:)   :: p_half, p_full, z_half, z_full, ug, vg, tg
real, allocatable, dimension(:,:,:,:,:) :: grid_tracers
real, allocatable, dimension(:,:,:)     :: psg, wg_full, dt_ug, dt_vg, dt_tg
real, allocatable, dimension(:,:,:,:)   :: dt_tracers
real, allocatable, dimension(:,:)       :: dt_psg, surf_geopotential
real, allocatable, dimension(:)         :: deg_lon, deg_lat, rad_lonb, rad_latb, pk, bk
real, allocatable, dimension(:,:)       :: rad_lon_2d, rad_lat_2d, rad_lonb_2d, rad_latb_2d

open(newunit=unit, file='harness.nml', status='old', action='read')
read(unit, nml=harness_nml)
close(unit)

! ---- initialisation, order of atmos_model.F90:148-350 and atmosphere.F90:151-266 ------------------
call fms_init()
call constants_init()
call register_tracers(MODEL_ATMOS, ntrace, ntprog, ntdiag, ntfamily)
call set_calendar_type(NO_CALENDAR)
call diag_manager_init()
Time      = set_time(0, 0)
Time_step = set_time(dt_atmos, 0)
dt_real   = real(dt_atmos)

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
allocate(surf_geopotential(is:ie,js:je), pk(num_levels+1), bk(num_levels+1))
p_half=0.; z_half=0.; p_full=0.; z_full=0.; wg_full=0.; psg=0.; ug=0.; vg=0.; tg=0.; grid_tracers=0.
dt_psg=0.; dt_ug=0.; dt_vg=0.; dt_tg=0.; dt_tracers=0.

call get_surf_geopotential(surf_geopotential)
previous = 1; current = 1
call get_initial_fields(ug(:,:,:,1), vg(:,:,:,1), tg(:,:,:,1), psg(:,:,1), grid_tracers(:,:,:,1,:))
call compute_pressures_and_heights(tg(:,:,:,current), psg(:,:,current), surf_geopotential, &
     z_full(:,:,:,current), z_half(:,:,:,current), p_full(:,:,:,current), p_half(:,:,:,current), &
     grid_tracers(:,:,:,current,nhum))
call compute_pressures_and_heights(tg(:,:,:,previous), psg(:,:,previous), surf_geopotential, &
     z_full(:,:,:,previous), z_half(:,:,:,previous), p_full(:,:,:,previous), p_half(:,:,:,previous), &
     grid_tracers(:,:,:,previous,nhum))
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
call idealized_moist_phys_init(Time, Time_step, nhum, rad_lon_2d, rad_lat_2d, rad_lonb_2d, rad_latb_2d, tg(:,:,num_levels,current))
call get_pk_bk(pk, bk)
call dump1('tab_pk.bin', pk); call dump1('tab_bk.bin', bk); call dump1('tab_deg_lat.bin', deg_lat)

if(trim(mode) == 'run') then
  call dump_state(0)
  t_loop = 0.
  call system_clock(count_rate=crate)
  do istep = 1, nsteps
    call system_clock(c0)
    call one_step(any(phys_steps == istep))
    call system_clock(c1)
    t_loop = t_loop + real(c1-c0,8)/real(crate,8)
    if(track_from >= 0 .and. istep >= track_from) call track_filter(istep == track_from)
    if(istep == dump_full_at) call dump_both_levels()
    if(any(dump_steps == istep)) call dump_state(istep)
  enddo
  write(*,'(a,i8,a,f12.6,a,f12.6)') 'REF_TIMING steps=', nsteps, ' seconds=', t_loop, ' ms_per_step=', 1.e3*t_loop/max(nsteps,1)
  write(*,'(a,4es24.16)') 'REF_STATE Tmin,Tmax,maxabsU,qmax=', minval(tg(:,:,:,current)), maxval(tg(:,:,:,current)), &
        maxval(abs(ug(:,:,:,current))), maxval(grid_tracers(:,:,:,current,nhum))
else if(trim(mode) == 'kernels') then
  do istep = 1, nsteps
    call one_step(.false.)
  enddo
  call run_kernels()
else
  write(*,*) 'unknown mode ', trim(mode)
  stop 2
endif

contains

!--------------------------------------------------------------------------------------------------
subroutine run_kernels()
real, dimension(is:ie,js:je,num_levels) :: tin, qin, es, des, conv_dt, conv_dq, qref, tref, tg_tmp, qg_tmp, cond_dt, cond_dq, rad_dt
real, dimension(is:ie,js:je) :: rain, snow, cape, cin, invtau_q, invtau_t, net_sw, lw_down, albedo, t_surf
integer, dimension(is:ie,js:je) :: convflag, klzbs, klcls
logical, dimension(is:ie,js:je) :: coldT
integer :: ii
if(current == previous) then
  delta_t = dt_real
else
  delta_t = 2*dt_real
endif
coldT = .false.
! the spun-up state convects everywhere: dry out every fourth longitude so that the sample also holds columns without CAPE
! and shallow ones (the routines are column-local, so this is just another set of columns)
do ii = is, ie
  if(mod(ii,4) == 0) grid_tracers(ii,:,:,previous,nhum) = 0.15*grid_tracers(ii,:,:,previous,nhum)
  if(mod(ii,4) == 2) grid_tracers(ii,:,:,previous,nhum) = 0.55*grid_tracers(ii,:,:,previous,nhum)
enddo
tin = tg(:,:,:,previous); qin = grid_tracers(:,:,:,previous,nhum)
call dump3('k_in_t_prev.bin', tin); call dump3('k_in_q_prev.bin', qin)
call dump3('k_in_u_prev.bin', ug(:,:,:,previous)); call dump3('k_in_v_prev.bin', vg(:,:,:,previous))
call dump3('k_in_t_cur.bin', tg(:,:,:,current)); call dump3('k_in_q_cur.bin', grid_tracers(:,:,:,current,nhum))
call dump3('k_in_u_cur.bin', ug(:,:,:,current)); call dump3('k_in_v_cur.bin', vg(:,:,:,current))
call dump3('k_in_p_full_prev.bin', p_full(:,:,:,previous)); call dump3('k_in_p_half_prev.bin', p_half(:,:,:,previous))
call dump3('k_in_p_full_cur.bin', p_full(:,:,:,current)); call dump3('k_in_p_half_cur.bin', p_half(:,:,:,current))
call dump3('k_in_z_full_cur.bin', z_full(:,:,:,current)); call dump3('k_in_z_half_cur.bin', z_half(:,:,:,current))
call dump1('k_in_delta_t.bin', (/delta_t/))
! --- sat_vapor_pres (shared/sat_vapor_pres/sat_vapor_pres.F90: lookup_es, lookup_des)
call lookup_es(tin, es); call lookup_des(tin, des)
call dump3('k_es.bin', es); call dump3('k_des.bin', des)
! --- convection (:862-880)
call qe_moist_convection(delta_t, tin, qin, p_full(:,:,:,previous), p_half(:,:,:,previous), coldT, rain, snow, conv_dt, conv_dq, &
                         qref, convflag, klzbs, cape, cin, invtau_q, invtau_t, tref, klcls)
call dump2('k_conv_rain.bin', rain); call dump3('k_conv_dt.bin', conv_dt); call dump3('k_conv_dq.bin', conv_dq)
call dump3('k_conv_qref.bin', qref); call dump3('k_conv_tref.bin', tref); call dump2('k_conv_cape.bin', cape); call dump2('k_conv_cin.bin', cin)
call dump2('k_conv_flag.bin', real(convflag)); call dump2('k_conv_klzb.bin', real(klzbs)); call dump2('k_conv_klcl.bin', real(klcls))
call dump2('k_conv_invtau_q.bin', invtau_q); call dump2('k_conv_invtau_t.bin', invtau_t)
tg_tmp = conv_dt + tin
qg_tmp = conv_dq + qin
! --- large-scale condensation (:983-987)
rain = 0.; snow = 0.
call lscale_cond(tg_tmp, qg_tmp, p_full(:,:,:,previous), p_half(:,:,:,previous), coldT, rain, snow, cond_dt, cond_dq)
call dump2('k_cond_rain.bin', rain); call dump3('k_cond_dt.bin', cond_dt); call dump3('k_cond_dq.bin', cond_dq)
! --- grey radiation, downward then upward (:1054-1061, :1156-1162)
albedo = 0.31
call two_stream_gray_rad_down(is, js, Time, rad_lat_2d, rad_lon_2d, p_half(:,:,:,current), tin, net_sw, lw_down, albedo, qin)
call dump2('k_rad_net_sw_down.bin', net_sw); call dump2('k_rad_lw_down.bin', lw_down)
t_surf = tin(:,:,num_levels) + 1.5 + 6.0*cos(rad_lon_2d)      ! warmer and colder than the air: unstable and stable surface layers
call dump2('k_in_t_surf.bin', t_surf)
rad_dt = 0.
call two_stream_gray_rad_up(is, js, Time, rad_lat_2d, p_half(:,:,:,current), t_surf, tin, rad_dt, albedo)
call dump3('k_rad_dt.bin', rad_dt)
! --- large-scale condensation once more on a moistened profile, so that the adjustment and the re-evaporation act
qg_tmp = 1.6*qg_tmp
call dump3('k_in_cond2_q.bin', qg_tmp)
call lscale_cond(tg_tmp, qg_tmp, p_full(:,:,:,previous), p_half(:,:,:,previous), coldT, rain, snow, cond_dt, cond_dq)
call dump2('k_cond2_rain.bin', rain); call dump3('k_cond2_dt.bin', cond_dt); call dump3('k_cond2_dq.bin', cond_dq)
call run_kernels_surface(conv_dt, conv_dq, rad_dt, t_surf, albedo, net_sw, lw_down, klcls)
end subroutine run_kernels

!--------------------------------------------------------------------------------------------------
subroutine run_kernels_surface(conv_dt, conv_dq, rad_dt, t_surf_in, albedo_in, net_sw, lw_down, klcls)
! surface fluxes, Rayleigh sponge, boundary-layer diffusivities, implicit vertical diffusion with the mixed-layer surface
! (idealized_moist_phys.F90:1077-1340), continuing with the tendencies accumulated so far
real, dimension(is:ie,js:je,num_levels), intent(in) :: conv_dt, conv_dq, rad_dt
real, dimension(is:ie,js:je), intent(in) :: t_surf_in, albedo_in, net_sw, lw_down
integer, dimension(is:ie,js:je), intent(in) :: klcls
real, dimension(is:ie,js:je) :: t_surf, q_surf, u_surf, v_surf, rough, gust, flux_t, flux_q, flux_r, flux_u, flux_v, drag_m, drag_t, &
     drag_q, w_atm, ustar, bstar, qstar, dhdt_surf, dedt_surf, dedq_surf, drdt_surf, dhdt_atm, dedq_atm, dtaudu_atm, dtaudv_atm, &
     ex_del_m, ex_del_h, ex_del_q, temp_2m, u_10m, v_10m, q_2m, rh_2m, bucket_depth, depth_change_lh, depth_change_conv, &
     depth_change_cond, z_pbl, fracland, albedo
logical, dimension(is:ie,js:je) :: land, avail, convect
real, dimension(is:ie,js:je,num_levels) :: tdtlw, diff_t, diff_m, diss_heat
type(surf_diff_type) :: Tri_surf
integer :: n
n = num_levels
dt_ug = 0.; dt_vg = 0.; dt_tracers = 0.
dt_tg = conv_dt/delta_t + rad_dt
dt_tracers(:,:,:,nhum) = conv_dq/delta_t
t_surf = t_surf_in; albedo = albedo_in
q_surf = 0.; u_surf = 0.; v_surf = 0.; rough = 3.21e-05; gust = 1.0
land = .false.; avail = .true.; bucket_depth = 0.; depth_change_lh = 0.; depth_change_conv = 0.; depth_change_cond = 0.
call surface_flux(tg(:,:,n,previous), grid_tracers(:,:,n,previous,nhum), ug(:,:,n,previous), vg(:,:,n,previous), &
     p_full(:,:,n,current), z_full(:,:,n,current), p_half(:,:,n+1,current), t_surf, t_surf, q_surf, .false., bucket_depth, 0.15, &
     depth_change_lh, depth_change_conv, depth_change_cond, u_surf, v_surf, rough, rough, rough, rough, gust, &
     flux_t, flux_q, flux_r, flux_u, flux_v, drag_m, drag_t, drag_q, w_atm, ustar, bstar, qstar, dhdt_surf, dedt_surf, dedq_surf, &
     drdt_surf, dhdt_atm, dedq_atm, dtaudu_atm, dtaudv_atm, ex_del_m, ex_del_h, ex_del_q, temp_2m, u_10m, v_10m, q_2m, rh_2m, &
     delta_t, land, .not.land, avail)
call dump2('k_sf_flux_t.bin', flux_t); call dump2('k_sf_flux_q.bin', flux_q); call dump2('k_sf_flux_r.bin', flux_r)
call dump2('k_sf_flux_u.bin', flux_u); call dump2('k_sf_flux_v.bin', flux_v)
call dump2('k_sf_drag_m.bin', drag_m); call dump2('k_sf_drag_t.bin', drag_t); call dump2('k_sf_drag_q.bin', drag_q)
call dump2('k_sf_w_atm.bin', w_atm); call dump2('k_sf_ustar.bin', ustar); call dump2('k_sf_bstar.bin', bstar); call dump2('k_sf_qstar.bin', qstar)
call dump2('k_sf_dhdt_surf.bin', dhdt_surf); call dump2('k_sf_dedt_surf.bin', dedt_surf); call dump2('k_sf_dedq_surf.bin', dedq_surf)
call dump2('k_sf_drdt_surf.bin', drdt_surf); call dump2('k_sf_dhdt_atm.bin', dhdt_atm); call dump2('k_sf_dedq_atm.bin', dedq_atm)
call dump2('k_sf_dtaudu_atm.bin', dtaudu_atm); call dump2('k_sf_dtaudv_atm.bin', dtaudv_atm); call dump2('k_sf_q_surf.bin', q_surf)
! --- Rayleigh sponge (damping_driver :1228-1237)
z_pbl = 0.
call damping_driver(is, js, rad_lat_2d, Time+Time_step, delta_t, p_full(:,:,:,current), p_half(:,:,:,current), &
     z_full(:,:,:,current), z_half(:,:,:,current), ug(:,:,:,previous), vg(:,:,:,previous), tg(:,:,:,previous), &
     grid_tracers(:,:,:,previous,nhum), grid_tracers(:,:,:,previous,:), dt_ug, dt_vg, dt_tg, dt_tracers(:,:,:,nhum), dt_tracers, z_pbl)
call dump3('k_damp_dt_u.bin', dt_ug); call dump3('k_damp_dt_v.bin', dt_vg); call dump3('k_damp_dt_t.bin', dt_tg)
! --- boundary-layer diffusivities (vert_turb_driver :1242-1262)
tdtlw = 0.; fracland = 0.; convect = .false.
call vert_turb_driver(1, 1, Time, Time+Time_step, delta_t, tdtlw, fracland, p_half(:,:,:,current), p_full(:,:,:,current), &
     z_half(:,:,:,current), z_full(:,:,:,current), ustar, bstar, qstar, rough, rad_lat_2d, convect, &
     ug(:,:,:,current), vg(:,:,:,current), tg(:,:,:,current), grid_tracers(:,:,:,current,nhum), grid_tracers(:,:,:,current,:), &
     ug(:,:,:,previous), vg(:,:,:,previous), tg(:,:,:,previous), grid_tracers(:,:,:,previous,nhum), grid_tracers(:,:,:,previous,:), &
     dt_ug, dt_vg, dt_tg, dt_tracers(:,:,:,nhum), dt_tracers, klcls, .false., diff_t, diff_m, gust, z_pbl)
call dump3('k_turb_diff_t.bin', diff_t); call dump3('k_turb_diff_m.bin', diff_m); call dump2('k_turb_gust.bin', gust); call dump2('k_turb_z_pbl.bin', z_pbl)
! --- implicit vertical diffusion, downward sweep / mixed layer / upward sweep (:1292-1330)
allocate(Tri_surf%dtmass(is:ie,js:je), Tri_surf%dflux_t(is:ie,js:je), Tri_surf%delta_t(is:ie,js:je), Tri_surf%delta_u(is:ie,js:je), &
         Tri_surf%delta_v(is:ie,js:je), Tri_surf%sst_miz(is:ie,js:je), Tri_surf%dflux_tr(is:ie,js:je,num_tracers), &
         Tri_surf%delta_tr(is:ie,js:je,num_tracers))
Tri_surf%dtmass = 0.; Tri_surf%dflux_t = 0.; Tri_surf%delta_t = 0.; Tri_surf%delta_u = 0.; Tri_surf%delta_v = 0.
Tri_surf%sst_miz = 0.; Tri_surf%dflux_tr = 0.; Tri_surf%delta_tr = 0.
call gcm_vert_diff_down(1, 1, delta_t, ug(:,:,:,previous), vg(:,:,:,previous), tg(:,:,:,previous), grid_tracers(:,:,:,previous,nhum), &
     grid_tracers(:,:,:,previous,:), diff_m, diff_t, p_half(:,:,:,current), p_full(:,:,:,current), z_full(:,:,:,current), &
     flux_u, flux_v, dtaudu_atm, dtaudv_atm, dt_ug, dt_vg, dt_tg, dt_tracers(:,:,:,nhum), dt_tracers, diss_heat, Tri_surf)
call dump2('k_vd_dtmass.bin', Tri_surf%dtmass); call dump2('k_vd_dflux_t.bin', Tri_surf%dflux_t); call dump2('k_vd_delta_t.bin', Tri_surf%delta_t)
call dump2('k_vd_delta_q.bin', Tri_surf%delta_tr(:,:,nhum)); call dump2('k_vd_dflux_q.bin', Tri_surf%dflux_tr(:,:,nhum))
call dump3('k_vd_down_dt_u.bin', dt_ug); call dump3('k_vd_down_dt_t.bin', dt_tg); call dump3('k_vd_diss_heat.bin', diss_heat)
call mixed_layer(Time, Time+Time_step, js, je, t_surf, flux_t, flux_q, flux_r, dt_real, net_sw, lw_down, Tri_surf, &
     dhdt_surf, dedt_surf, dedq_surf, drdt_surf, dhdt_atm, dedq_atm, albedo)
call dump2('k_ml_t_surf.bin', t_surf); call dump2('k_ml_delta_t.bin', Tri_surf%delta_t); call dump2('k_ml_delta_q.bin', Tri_surf%delta_tr(:,:,nhum))
call gcm_vert_diff_up(1, 1, delta_t, Tri_surf, dt_tg, dt_tracers(:,:,:,nhum), dt_tracers)
call dump3('k_fin_dt_u.bin', dt_ug); call dump3('k_fin_dt_v.bin', dt_vg); call dump3('k_fin_dt_t.bin', dt_tg)
call dump3('k_fin_dt_q.bin', dt_tracers(:,:,:,nhum))
end subroutine run_kernels_surface

!--------------------------------------------------------------------------------------------------
subroutine one_step(dump_phys)
! atmosphere.F90:286-349, idealized_moist_model branch, without spectral_diagnostics
logical, intent(in) :: dump_phys
character(len=8) :: tag
dt_ug = 0.0; dt_vg = 0.0; dt_tg = 0.0; dt_psg = 0.0; dt_tracers = 0.0
if(current == previous) then
  delta_t = dt_real
else
  delta_t = 2*dt_real
endif
Time_next = Time + Time_step
if(dump_phys) then
  write(tag,'(i6.6)') istep
  call dump3('ph_in_u_prev_'//trim(tag)//'.bin', ug(:,:,:,previous)); call dump3('ph_in_u_cur_'//trim(tag)//'.bin', ug(:,:,:,current))
  call dump3('ph_in_v_prev_'//trim(tag)//'.bin', vg(:,:,:,previous)); call dump3('ph_in_v_cur_'//trim(tag)//'.bin', vg(:,:,:,current))
  call dump3('ph_in_t_prev_'//trim(tag)//'.bin', tg(:,:,:,previous)); call dump3('ph_in_t_cur_'//trim(tag)//'.bin', tg(:,:,:,current))
  call dump3('ph_in_q_prev_'//trim(tag)//'.bin', grid_tracers(:,:,:,previous,nhum))
  call dump3('ph_in_q_cur_'//trim(tag)//'.bin', grid_tracers(:,:,:,current,nhum))
  call dump2('ph_in_ps_prev_'//trim(tag)//'.bin', psg(:,:,previous)); call dump2('ph_in_ps_cur_'//trim(tag)//'.bin', psg(:,:,current))
  call dump3('ph_in_p_half_prev_'//trim(tag)//'.bin', p_half(:,:,:,previous)); call dump3('ph_in_p_half_cur_'//trim(tag)//'.bin', p_half(:,:,:,current))
  call dump3('ph_in_p_full_prev_'//trim(tag)//'.bin', p_full(:,:,:,previous)); call dump3('ph_in_p_full_cur_'//trim(tag)//'.bin', p_full(:,:,:,current))
  call dump3('ph_in_z_half_prev_'//trim(tag)//'.bin', z_half(:,:,:,previous)); call dump3('ph_in_z_half_cur_'//trim(tag)//'.bin', z_half(:,:,:,current))
  call dump3('ph_in_z_full_prev_'//trim(tag)//'.bin', z_full(:,:,:,previous)); call dump3('ph_in_z_full_cur_'//trim(tag)//'.bin', z_full(:,:,:,current))
endif
call idealized_moist_phys(Time, p_half, p_full, z_half, z_full, ug, vg, psg, wg_full, tg, grid_tracers, &
                          previous, current, dt_ug, dt_vg, dt_tg, dt_tracers)
if(dump_phys) then
  call dump3('ph_dt_u_'//trim(tag)//'.bin', dt_ug); call dump3('ph_dt_v_'//trim(tag)//'.bin', dt_vg)
  call dump3('ph_dt_t_'//trim(tag)//'.bin', dt_tg); call dump3('ph_dt_q_'//trim(tag)//'.bin', dt_tracers(:,:,:,nhum))
endif
if(previous == current) then
  future = 3 - current
else
  future = previous
endif
call spectral_dynamics(Time, psg(:,:,future), ug(:,:,:,future), vg(:,:,:,future), &
                       tg(:,:,:,future), tracer_attributes, grid_tracers(:,:,:,:,:), future, &
                       dt_psg, dt_ug, dt_vg, dt_tg, dt_tracers, wg_full, &
                       p_full(:,:,:,current), p_half(:,:,:,current), z_full(:,:,:,current))
call compute_pressures_and_heights(tg(:,:,:,future), psg(:,:,future), surf_geopotential, &
     z_full(:,:,:,future), z_half(:,:,:,future), p_full(:,:,:,future), p_half(:,:,:,future), &
     grid_tracers(:,:,:,future,nhum))
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

subroutine dump_both_levels()
! what a restart of spectral_dynamics_mod + atmosphere_mod + mixed_layer_mod holds (spectral_dynamics.F90:1502-1531, atmosphere.F90:362-375,
! mixed_layer.F90:735-745)
real, pointer :: ts_ref(:,:)
call dumpc3('rs_vors_cur.bin', sc_vor);  call dumpc3('rs_divs_cur.bin', sc_div)
call dumpc3('rs_ts_cur.bin', sc_t);      call dumpc2('rs_lnps_cur.bin', sc_lp)
call dumpc3('rs_vors_prev.bin', f_vor);  call dumpc3('rs_divs_prev.bin', f_div)
call dumpc3('rs_ts_prev.bin', f_t);      call dumpc2('rs_lnps_prev.bin', f_lp)
call dump3('rs_ug_cur.bin', ug(:,:,:,current));   call dump3('rs_ug_prev.bin', ug(:,:,:,previous))
call dump3('rs_vg_cur.bin', vg(:,:,:,current));   call dump3('rs_vg_prev.bin', vg(:,:,:,previous))
call dump3('rs_tg_cur.bin', tg(:,:,:,current));   call dump3('rs_tg_prev.bin', tg(:,:,:,previous))
call dump2('rs_psg_cur.bin', psg(:,:,current));   call dump2('rs_psg_prev.bin', psg(:,:,previous))
call dump3('rs_wg_full.bin', wg_full)
call dump3('rs_tr1_cur.bin', grid_tracers(:,:,:,current,nhum))          ! the dynamics' and atmosphere_mod's newest level
call dump3('rs_tr1_prev_atm.bin', grid_tracers(:,:,:,previous,nhum))    ! atmosphere_mod's (unfiltered) previous level
call dump3('rs_tr1_prev_filt.bin', f_tr(:,:,:,nhum))                    ! the dynamics' Robert-filtered previous level
call c_f_pointer(ref_peek_t_surf(), ts_ref, (/nlon, nlat/))
call dump2('rs_t_surf.bin', ts_ref)
write(*,'(a,5es16.8)') 'REF_DEVELOPED max|u|,max|v|,Tmin,Tmax,qmax=', maxval(abs(ug(:,:,:,current))), maxval(abs(vg(:,:,:,current))), &
     minval(tg(:,:,:,current)), maxval(tg(:,:,:,current)), maxval(grid_tracers(:,:,:,current,nhum))
write(*,'(a,2es16.8)') 'REF_DEVELOPED t_surf min,max=', minval(ts_ref), maxval(ts_ref)
end subroutine dump_both_levels

!--------------------------------------------------------------------------------------------------
subroutine dump_state(n)
integer, intent(in) :: n
character(len=8) :: tag
write(tag,'(i6.6)') n
call dump3('st_ug_'//trim(tag)//'.bin', ug(:,:,:,current))
call dump3('st_vg_'//trim(tag)//'.bin', vg(:,:,:,current))
call dump3('st_tg_'//trim(tag)//'.bin', tg(:,:,:,current))
call dump2('st_psg_'//trim(tag)//'.bin', psg(:,:,current))
call dump3('st_q_'//trim(tag)//'.bin', grid_tracers(:,:,:,current,nhum))
end subroutine dump_state

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

end program ref_moist_harness
