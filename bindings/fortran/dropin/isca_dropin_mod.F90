! isca_dropin_mod -- what the drop-in modules of this directory share: the handle of the MI355X core (include/isca_dyn.h through
! bindings/fortran/isca_dyn_c.F90), its dimensions, and the FATAL convention (a non-zero return of the library becomes
! error_mesg(routine, message, FATAL), as every reference routine reports).
!
! The modules here carry the reference's module NAMES and public argument lists (spectral_dynamics_mod, transforms_mod,
! press_and_geopot_mod, hs_forcing_mod, implicit_mod, spectral_damping_mod, leapfrog_mod, vert_advection_mod, fv_advection_mod,
! global_integral_mod, tracer_type_mod, atmosphere_mod), so that a host that `use`s those names -- the reference's own
! atmos_model / atmosphere driver, or this repository's oracle/ref_harness.F90 -- links against the GPU library unchanged.  They are
! compiled inside the reference tree, with its infrastructure modules (fms_mod, time_manager_mod, tracer_manager_mod, ...), by
! oracle/build_ref.py dropin; every array crosses the boundary in its Fortran layout.
module isca_dropin_mod
use iso_c_binding
use isca_dyn_c
use fms_mod, only: error_mesg, FATAL
implicit none
public

type(c_ptr), save :: core = c_null_ptr      ! isca_dyn_t* of this process
logical, save :: core_ready = .false.
! spectral_dynamics_nml: graceful_shutdown (spectral_dynamics.F90:976-1005) -- when the step reports temperatures out of the valid range, the
! reference agrees on that over all PEs, ends the diagnostics (diag_manager_end: partially complete history files are written out) and only then
! raises the FATAL.  Here the library's verdict is already collective (every rank's step returns the error), so each rank closes its history files
! (isca_dyn_diag_close: the records accumulated so far) before its FATAL.
logical, save :: graceful = .false.
integer, save :: nlon = 0, nlat = 0, nlev = 0, nfour = 0, nsph = 0      ! lon_max, lat_max, num_levels, num_fourier, num_spherical
integer, save :: ntrace = 0                                                ! prognostic tracers of the field_table
! The decomposition (spec_mpp.F90:61-80, atmosphere_domain): this process holds the latitude rows js_loc..je_loc of every longitude -- all of them with
! one rank; the library deals the zonal wavenumbers itself.  Rank and number of ranks come from the environment (isca_env_rank: the mpp of this build
! has no MPI), the exchanges are the library's (RCCL over xGMI; ISCA_COMM=ipc: host-staged, ranks may share a GPU).
integer, save :: my_rank = 0, num_ranks = 1, js_loc = 1, je_loc = 0
logical, save :: virtual_t = .false.
integer, save :: dropin_physics = 2     ! isca_dyn_config%physics of the core spectral_dynamics_init creates: 2 = the caller's physics (spectral_dynamics
                                        ! receives its tendencies); atmosphere_mod sets 0 (hs_forcing inside the device step) or 1 (the Frierson chain)
real, save :: ref_sea_level_press = 101325.
logical, save :: triang = .true.        ! spectral_dynamics_nml: triang_trunc, fourier_inc of the core (get_triang_trunc, get_fourier_inc)
integer, save :: finc = 1
! atmosphere_nml: idealized_moist_model -- the namelist values idealized_moist_phys_init collected for isca_dyn_config%moist (physics = 1)
type(isca_moist_config), save :: dropin_moist
logical, save :: dropin_moist_set = .false.

contains

subroutine chk(ierr, routine)
  integer(c_int), intent(in) :: ierr
  character(len=*), intent(in) :: routine
  character(len=:), allocatable :: msg
  integer(c_int) :: closed
  if(ierr == 0) return
  msg = isca_message()
  if(graceful .and. core_ready) closed = isca_dyn_diag_close(core)      ! (its own failure would only replace the message that matters)
  call error_mesg(routine, msg, FATAL)
end subroutine chk

subroutine need_core(routine)
  character(len=*), intent(in) :: routine
  if(.not. core_ready) call error_mesg(routine, 'spectral_dynamics_init has not been called', FATAL)
end subroutine need_core

function cstr(s) result(c)
  character(len=*), intent(in) :: s
  character(kind=c_char, len=len_trim(s)+1) :: c
  c = trim(s)//c_null_char
end function cstr

! a (lon, lat [, lev]) field or a spectral (m, n [, lev]) field of time level tl (0 = previous, 1 = current) of the model state
subroutine get_grid3(name, tl, a)
  character(len=*), intent(in) :: name
  integer, intent(in) :: tl
  real, intent(out) :: a(:,:,:)
  real(c_double), allocatable :: buf(:)
  allocate(buf(size(a)))
  call chk(isca_dyn_get_state(core, cstr(name), int(tl, c_int), buf, size(buf, kind=c_size_t)), 'get_state '//name)
  a = reshape(buf, shape(a))
end subroutine get_grid3
subroutine get_grid2(name, tl, a)
  character(len=*), intent(in) :: name
  integer, intent(in) :: tl
  real, intent(out) :: a(:,:)
  real(c_double), allocatable :: buf(:)
  allocate(buf(size(a)))
  call chk(isca_dyn_get_state(core, cstr(name), int(tl, c_int), buf, size(buf, kind=c_size_t)), 'get_state '//name)
  a = reshape(buf, shape(a))
end subroutine get_grid2
subroutine get_table1(name, a)
  character(len=*), intent(in) :: name
  real, intent(out) :: a(:)
  real(c_double), allocatable :: buf(:)
  allocate(buf(size(a)))
  call chk(isca_dyn_get_table(core, cstr(name), buf, size(buf, kind=c_size_t)), 'get_table '//name)
  a = buf
end subroutine get_table1

end module isca_dropin_mod
