! The modules of the time step under the reference's names and public argument lists, each forwarding to the entry point of the MI355X
! library that runs the step's own kernel on the caller's arrays (include/isca_dyn.h "Components of the step on caller fields"):
!   press_and_geopot_mod (atmos_spectral/model/press_and_geopot.F90:47-65), implicit_mod (implicit.F90:41), spectral_damping_mod
!   (spectral_damping.F90:33-52), leapfrog_mod (leapfrog.F90:31-50), vert_advection_mod (atmos_shared/vert_advection/vert_advection.F90:32-43),
!   fv_advection_mod (fv_advection.F90:46-52), global_integral_mod (global_integral.F90:38), hs_forcing_mod (atmos_param/hs_forcing/
!   hs_forcing.F90:148-150, 276).
! The level tables (pk, bk), the reference temperature profile, the damping coefficients and the Held-Suarez parameters are those the
! device core was created with (spectral_dynamics_init).

module press_and_geopot_mod
use iso_c_binding
use fms_mod, only: error_mesg, FATAL
use isca_dyn_c
use isca_dropin_mod
implicit none
private
public :: pressure_variables, compute_geopotential, compute_pressures_and_heights
interface pressure_variables
  module procedure pressure_variables_3d
end interface
interface compute_pressures_and_heights
  module procedure compute_pressures_and_heights_3d
end interface
contains
subroutine pressure_variables_3d(p_half, ln_p_half, p_full, ln_p_full, surface_p)
real, intent(out), dimension(:,:,:) :: p_half, ln_p_half, p_full, ln_p_full
real, intent(in),  dimension(:,:)   :: surface_p
call need_core('pressure_variables')
call chk(isca_pressure_variables(core, surface_p, p_half, ln_p_half, p_full, ln_p_full), 'pressure_variables')
end subroutine pressure_variables_3d
! the caller's surface geopotential is the lower boundary (press_and_geopot.F90:331); q_grid enters with use_virtual_temperature (:340-347)
subroutine compute_geopotential(t_grid, ln_p_half, ln_p_full, surf_geopotential, geopot_full, geopot_half, q_grid)
real, intent(in),  dimension(:,:,:) :: t_grid, ln_p_half, ln_p_full
real, intent(in),  dimension(:,:)   :: surf_geopotential
real, intent(out), dimension(:,:,:) :: geopot_full, geopot_half
real, intent(in), optional, dimension(:,:,:) :: q_grid
real(c_double), allocatable, target :: q(:)
call need_core('compute_geopotential')
if(present(q_grid)) then
  allocate(q(size(q_grid))); q = reshape(q_grid, (/size(q_grid)/))
  call chk(isca_compute_geopotential_surf(core, t_grid, ln_p_half, ln_p_full, surf_geopotential, c_loc(q), geopot_full, geopot_half), 'compute_geopotential')
else       ! (with use_virtual_temperature the library answers like the reference: 'q_grid must be present when use_virtual_temperature=.true.')
  call chk(isca_compute_geopotential_surf(core, t_grid, ln_p_half, ln_p_full, surf_geopotential, c_null_ptr, geopot_full, geopot_half), 'compute_geopotential')
endif
end subroutine compute_geopotential
subroutine compute_pressures_and_heights_3d(t_grid, ps_grid, surf_geopotential, z_full, z_half, p_full, p_half, q_grid)
real, intent(in), dimension(:,:,:) :: t_grid
real, intent(in), dimension(:,:)   :: ps_grid, surf_geopotential
real, intent(in), optional, dimension(:,:,:), target :: q_grid
real, intent(out), dimension(size(t_grid,1),size(t_grid,2),size(t_grid,3)  ) :: z_full, p_full
real, intent(out), dimension(size(t_grid,1),size(t_grid,2),size(t_grid,3)+1) :: z_half, p_half
real(c_double), allocatable, target :: q(:)
call need_core('compute_pressures_and_heights')
if(present(q_grid)) then
  allocate(q(size(q_grid))); q = reshape(q_grid, (/size(q_grid)/))
  call chk(isca_compute_pressures_and_heights(core, t_grid, ps_grid, c_loc(q), z_full, z_half, p_full, p_half), 'compute_pressures_and_heights')
else
  call chk(isca_compute_pressures_and_heights(core, t_grid, ps_grid, c_null_ptr, z_full, z_half, p_full, p_half), 'compute_pressures_and_heights')
endif
end subroutine compute_pressures_and_heights_3d
end module press_and_geopot_mod

!==================================================================================================================================
module implicit_mod
use iso_c_binding
use isca_dyn_c
use isca_dropin_mod
implicit none
private
public :: implicit_correction
contains
subroutine implicit_correction(dt_divs, dt_ts, dt_ln_ps, divs, ts, ln_ps, dt_in, previous, current)
complex, intent(inout), dimension(0:,0:,:) :: dt_divs, dt_ts
complex, intent(inout), dimension(0:,0:) :: dt_ln_ps
complex, intent(in), dimension(0:,0:,:,:) :: divs, ts
complex, intent(in), dimension(0:,0:,:) :: ln_ps
real, intent(in) :: dt_in
integer, intent(in) :: previous, current
call need_core('implicit_correction')
call chk(isca_implicit_correction(core, dt_divs, dt_ts, dt_ln_ps, divs(:,:,:,previous), divs(:,:,:,current), ts(:,:,:,previous), &
                                  ts(:,:,:,current), ln_ps(:,:,previous), ln_ps(:,:,current), dt_in), 'implicit_correction')
end subroutine implicit_correction
end module implicit_mod

!==================================================================================================================================
module spectral_damping_mod
use iso_c_binding
use fms_mod, only: error_mesg, FATAL
use isca_dyn_c
use isca_dropin_mod
implicit none
private
public :: compute_spectral_damping, compute_spectral_damping_vor, compute_spectral_damping_div
interface compute_spectral_damping
  module procedure compute_spectral_damping_3d
end interface
contains
subroutine damp(which, spec, dt_spec, current_dt, routine)
integer, intent(in) :: which
complex, intent(in), dimension(:,:,:) :: spec
complex, intent(inout), dimension(:,:,:) :: dt_spec
real, intent(in) :: current_dt
character(len=*), intent(in) :: routine
call need_core(routine)
if(size(spec,3) /= nlev) call error_mesg(routine,'the device kernel works on num_levels levels', FATAL)
call chk(isca_compute_spectral_damping(core, int(which, c_int), spec, dt_spec, current_dt), routine)
end subroutine damp
subroutine compute_spectral_damping_3d(spec, dt_spec, current_dt)
complex, intent(in), dimension(:,:,:) :: spec
real,    intent(in) :: current_dt
complex, intent(inout), dimension(:,:,:) :: dt_spec
call damp(0, spec, dt_spec, current_dt, 'compute_spectral_damping')
end subroutine compute_spectral_damping_3d
subroutine compute_spectral_damping_vor(vor, dt_vor, current_dt)
complex, intent(in), dimension(:,:,:) :: vor
real,    intent(in) :: current_dt
complex, intent(inout), dimension(:,:,:) :: dt_vor
call damp(1, vor, dt_vor, current_dt, 'compute_spectral_damping_vor')
end subroutine compute_spectral_damping_vor
subroutine compute_spectral_damping_div(div, dt_div, current_dt)
complex, intent(in), dimension(:,:,:) :: div
real,    intent(in) :: current_dt
complex, intent(inout), dimension(:,:,:) :: dt_div
call damp(2, div, dt_div, current_dt, 'compute_spectral_damping_div')
end subroutine compute_spectral_damping_div
end module spectral_damping_mod

!==================================================================================================================================
module leapfrog_mod
use iso_c_binding
use isca_dyn_c
use isca_dropin_mod
implicit none
private
public :: leapfrog, leapfrog_2level_A, leapfrog_2level_B
interface leapfrog
  module procedure leapfrog_3d_complex
end interface
interface leapfrog_2level_A
  module procedure leapfrog_2level_A_3d_complex
end interface
interface leapfrog_2level_B
  module procedure leapfrog_2level_B_3d_complex
end interface
contains
! the time levels are slices of `a`: contiguous, so the library works on them in place (future may be the previous level's storage)
subroutine leapfrog_2level_A_3d_complex(a, dt_a, previous, current, future, delta_t, robert_coeff, raw_filter_coeff, prev_curr_part_raw_filter)
complex, intent(inout), dimension(:,:,:,:), target, contiguous :: a
complex, intent(in),    dimension(:,:,:  ), target, contiguous :: dt_a
integer, intent(in) :: previous, current, future
real,    intent(in) :: delta_t, robert_coeff, raw_filter_coeff
complex, intent(out), dimension(size(dt_a,1),size(dt_a,2),size(dt_a,3)), target :: prev_curr_part_raw_filter
call need_core('leapfrog_2level_A')
call chk(isca_leapfrog_2level_a(core, 2_c_size_t*size(dt_a, kind=c_size_t), c_loc(a(1,1,1,previous)), c_loc(a(1,1,1,current)), &
                                c_loc(a(1,1,1,future)), c_loc(dt_a), delta_t, robert_coeff, raw_filter_coeff, &
                                c_loc(prev_curr_part_raw_filter)), 'leapfrog_2level_A')
end subroutine leapfrog_2level_A_3d_complex
subroutine leapfrog_2level_B_3d_complex(a, part_filt_a, current, future, robert_coeff, raw_filter_coeff)
complex, intent(inout), dimension(:,:,:,:), target, contiguous :: a
integer, intent(in) :: current, future
real,    intent(in) :: robert_coeff, raw_filter_coeff
complex, intent(in), dimension(:,:,:), target, contiguous :: part_filt_a
call need_core('leapfrog_2level_B')
call chk(isca_leapfrog_2level_b(core, 2_c_size_t*size(part_filt_a, kind=c_size_t), c_loc(a(1,1,1,current)), c_loc(a(1,1,1,future)), &
                                c_loc(part_filt_a), robert_coeff, raw_filter_coeff), 'leapfrog_2level_B')
end subroutine leapfrog_2level_B_3d_complex
subroutine leapfrog_3d_complex(a, dt_a, previous, current, future, delta_t, robert_coeff, raw_filter_coeff)
complex, intent(inout), dimension(:,:,:,:), target, contiguous :: a
complex, intent(in),    dimension(:,:,:  ), target, contiguous :: dt_a
integer, intent(in) :: previous, current, future
real,    intent(in) :: delta_t, robert_coeff, raw_filter_coeff
complex, dimension(size(dt_a,1),size(dt_a,2),size(dt_a,3)) :: part
call leapfrog_2level_A_3d_complex(a, dt_a, previous, current, future, delta_t, robert_coeff, raw_filter_coeff, part)
call leapfrog_2level_B_3d_complex(a, part, current, future, robert_coeff, raw_filter_coeff)
end subroutine leapfrog_3d_complex
end module leapfrog_mod

!==================================================================================================================================
module vert_advection_mod
use iso_c_binding
use fms_mod, only: error_mesg, FATAL
use isca_dyn_c
use isca_dropin_mod
implicit none
private
public :: vert_advection, vert_advection_end
integer, parameter, public :: SECOND_CENTERED = 101, FOURTH_CENTERED = 102, FINITE_VOLUME_LINEAR = 103, FINITE_VOLUME_PARABOLIC = 104, &
                              FINITE_VOLUME_PARABOLIC2 = 105, SECOND_CENTERED_WTS = 106, FOURTH_CENTERED_WTS = 107, &
                              VAN_LEER_LINEAR = FINITE_VOLUME_LINEAR
integer, parameter, public :: FLUX_FORM = 201, ADVECTIVE_FORM = 202
integer, parameter, public :: WEIGHTED_TENDENCY = 1
interface vert_advection
  module procedure vert_advection_3d
end interface
contains
! The device kernels take the layer depths as dz = dpk + dbk * surf_p of the core's levels (what spectral_dynamics passes): the surface
! pressure is recovered from the lowest layer, and dz is checked against it.
subroutine vert_advection_3d(dt, w, dz, r, rdt, mask, scheme, form, flags)
real, intent(in)                    :: dt
real, intent(in),  dimension(:,:,:) :: w, dz, r
real, intent(out), dimension(:,:,:) :: rdt
real,    intent(in), optional :: mask(:,:,:)
integer, intent(in), optional :: scheme, form, flags
real, dimension(size(dz,1), size(dz,2)) :: surf_p
real, dimension(nlev+1) :: pk, bk
integer :: sch, k, L
call need_core('vert_advection')
sch = SECOND_CENTERED; if(present(scheme)) sch = scheme
if(present(mask) .or. present(flags)) call error_mesg('vert_advection','mask / flags are not available on the device', FATAL)
if(present(form)) then
  if(form /= ADVECTIVE_FORM) call error_mesg('vert_advection','only form = ADVECTIVE_FORM is available on the device', FATAL)
else
  call error_mesg('vert_advection','only form = ADVECTIVE_FORM is available on the device (the default is FLUX_FORM)', FATAL)
endif
L = size(dz, 3)
if(L /= nlev) call error_mesg('vert_advection','the device kernels work on num_levels levels', FATAL)
call get_table1('pk', pk); call get_table1('bk', bk)
surf_p = (dz(:,:,L) - (pk(L+1) - pk(L)))/(bk(L+1) - bk(L))
do k = 1, L
  if(any(abs(dz(:,:,k) - ((pk(k+1) - pk(k)) + (bk(k+1) - bk(k))*surf_p)) > 1.e-9*abs(dz(:,:,k)))) &
    call error_mesg('vert_advection','dz is not the layer depth dpk + dbk * surface pressure of the model levels', FATAL)
enddo
if(sch == SECOND_CENTERED) then
  call chk(isca_vert_advection_centered(core, w, surf_p, r, rdt), 'vert_advection')
else if(sch == FINITE_VOLUME_PARABOLIC) then
  call chk(isca_vert_advection_ppm(core, dt, w, surf_p, r, rdt), 'vert_advection')
else
  call error_mesg('vert_advection','only SECOND_CENTERED and FINITE_VOLUME_PARABOLIC are available on the device', FATAL)
endif
end subroutine vert_advection_3d
subroutine vert_advection_end
end subroutine vert_advection_end
end module vert_advection_mod

!==================================================================================================================================
module fv_advection_mod
use iso_c_binding
use fms_mod, only: error_mesg, FATAL
use isca_dyn_c
use isca_dropin_mod
implicit none
private
public :: a_grid_horiz_advection
interface a_grid_horiz_advection
  module procedure a_grid_horiz_advection_3d
end interface
contains
subroutine a_grid_horiz_advection_3d(ua, va, q, dt, dq_dt, flux)
real, intent(in),    dimension(:,:,:) :: ua, va, q
real, intent(in)                      :: dt
real, intent(inout), dimension(:,:,:) :: dq_dt
logical, optional, intent(in) :: flux
call need_core('a_grid_horiz_advection')
if(present(flux)) then
  if(flux) call error_mesg('a_grid_horiz_advection','flux = .true. is not available on the device', FATAL)
endif
if(size(q,3) /= nlev) call error_mesg('a_grid_horiz_advection','the device kernel works on num_levels levels', FATAL)
call chk(isca_a_grid_horiz_advection(core, ua, va, q, dt, dq_dt), 'a_grid_horiz_advection')
end subroutine a_grid_horiz_advection_3d
end module fv_advection_mod

!==================================================================================================================================
module global_integral_mod
use iso_c_binding
use isca_dyn_c
use isca_dropin_mod
implicit none
private
public :: mass_weighted_global_integral
contains
function mass_weighted_global_integral(field, surf_press)
real :: mass_weighted_global_integral
real, intent(in), dimension(:,:,:) :: field
real, intent(in), dimension(:,:)   :: surf_press
real(c_double) :: v
call need_core('mass_weighted_global_integral')
call chk(isca_mass_weighted_global_integral(core, field, surf_press, v), 'mass_weighted_global_integral')
mass_weighted_global_integral = v
end function mass_weighted_global_integral
end module global_integral_mod

!==================================================================================================================================
module hs_forcing_mod
use iso_c_binding
use fms_mod, only: error_mesg, FATAL
use time_manager_mod, only: time_type
use isca_dyn_c
use isca_dropin_mod
implicit none
private
public :: hs_forcing, hs_forcing_init, hs_forcing_end
logical :: module_is_initialized = .false.
contains
! hs_forcing_nml was read when the device core was created (spectral_dynamics_init): nothing is left to set up
subroutine hs_forcing_init(axes, Time, lonb, latb, lat)
integer, intent(in) :: axes(4)
type(time_type), intent(in) :: Time
real, intent(in), dimension(:,:) :: lat
real, intent(in), optional, dimension(:,:) :: lonb, latb
call need_core('hs_forcing_init')
module_is_initialized = .true.
end subroutine hs_forcing_init
subroutine hs_forcing(is, ie, js, je, dt, Time, lon, lat, p_half, p_full, u, v, t, r, um, vm, tm, rm, udt, vdt, tdt, rdt, zfull, mask, kbot)
integer, intent(in)                        :: is, ie, js, je
real, intent(in)                           :: dt
type(time_type), intent(in)                :: Time
real, intent(in),    dimension(:,:)        :: lon, lat
real, intent(in),    dimension(:,:,:)      :: p_half, p_full
real, intent(in),    dimension(:,:,:)      :: u, v, t, um, vm, tm, zfull
real, intent(in),    dimension(:,:,:,:)    :: r, rm
real, intent(inout), dimension(:,:,:)      :: udt, vdt, tdt
real, intent(inout), dimension(:,:,:,:)    :: rdt
real, intent(in),    dimension(:,:,:), optional :: mask
integer, intent(in), dimension(:,:),   optional :: kbot
integer :: n
if(.not. module_is_initialized) call error_mesg('hs_forcing','hs_forcing_init has not been called', FATAL)
if(present(mask) .or. present(kbot)) call error_mesg('hs_forcing','mask / kbot are not available on the device', FATAL)
if(is /= 1 .or. js /= 1 .or. ie - is + 1 /= nlon .or. je - js + 1 /= nlat) &
  call error_mesg('hs_forcing','the device kernel works on the whole (lon_max, lat_max) window', FATAL)
! rayleigh_damping, dissipative heating and newtonian_damping act on the fields of time level tau - 1 (:199-233) ...
call chk(isca_hs_forcing(core, dt, p_half, p_full, um, vm, tm, udt, vdt, tdt), 'hs_forcing')
! ... and every tracer gets tracer_source_sink (:240-265)
do n = 1, size(rdt, 4)
  call chk(isca_hs_tracer_source_sink(core, p_half(:,:,size(p_half,3)), rm(:,:,:,n), rdt(:,:,:,n)), 'hs_forcing')
enddo
end subroutine hs_forcing
subroutine hs_forcing_end
module_is_initialized = .false.
end subroutine hs_forcing_end
end module hs_forcing_mod
