! transforms_mod -- the reference's module name and the public procedures of atmos_spectral/tools/transforms.F90:134-184 (with those it
! carries forward from spherical_fourier_mod, grid_fourier_mod, spherical_mod, gauss_and_legendre_mod and spec_mpp_mod) that have an entry
! point in the MI355X library: every array goes to the device, through the step's own kernels, and back (isca_trans_*, isca_compute_*).
! One process = the whole grid and every wavenumber here (is:ie = 1:lon_max, js:je = 1:lat_max, ms:me = 0:num_fourier, ns:ne =
! 0:num_spherical); the latitude-band / wavenumber-set decomposition of spec_mpp_mod lives inside the library (isca_amd/parallel.py).
! The module is initialised by spectral_dynamics_init (which creates the device core), not by transforms_init.
module transforms_mod
use iso_c_binding
use fms_mod, only: error_mesg, FATAL
use isca_dyn_c
use isca_dropin_mod
implicit none
private

public :: transforms_are_initialized, trans_spherical_to_grid, trans_grid_to_spherical, divide_by_cos, divide_by_cos2, trans_filter
public :: get_lat_max, get_triang_trunc, get_num_fourier, get_fourier_inc, get_num_spherical, get_grid_boundaries
public :: vor_div_from_uv_grid, uv_grid_from_vor_div, horizontal_advection, area_weighted_global_mean
public :: trans_spherical_to_fourier, trans_fourier_to_spherical, get_sin_lat, get_cos_lat, get_cosm_lat, get_cosm2_lat, get_deg_lat, get_wts_lat
public :: trans_grid_to_fourier, trans_fourier_to_grid, get_lon_max, get_deg_lon
public :: compute_laplacian, get_eigen_laplacian, compute_gradient_cos, compute_ucos_vcos, compute_vor_div, triangular_truncation
public :: compute_legendre, compute_gaussian
public :: get_grid_domain, get_spec_domain

interface trans_spherical_to_grid
  module procedure trans_spherical_to_grid_3d, trans_spherical_to_grid_2d
end interface
interface trans_grid_to_spherical
  module procedure trans_grid_to_spherical_3d, trans_grid_to_spherical_2d
end interface
interface trans_filter
  module procedure trans_filter_3d, trans_filter_2d
end interface
interface divide_by_cos
  module procedure divide_by_cos_3d, divide_by_cos_2d
end interface
interface divide_by_cos2
  module procedure divide_by_cos2_3d, divide_by_cos2_2d
end interface
interface uv_grid_from_vor_div
  module procedure uv_grid_from_vor_div_2d, uv_grid_from_vor_div_3d
end interface
interface vor_div_from_uv_grid
  module procedure vor_div_from_uv_grid_2d, vor_div_from_uv_grid_3d
end interface
interface horizontal_advection
  module procedure horizontal_advection_2d, horizontal_advection_3d
end interface
interface trans_spherical_to_fourier
  module procedure trans_spherical_to_fourier_3d
end interface
interface trans_fourier_to_spherical
  module procedure trans_fourier_to_spherical_3d
end interface
interface trans_grid_to_fourier
  module procedure trans_grid_to_fourier_3d
end interface
interface trans_fourier_to_grid
  module procedure trans_fourier_to_grid_3d
end interface
interface compute_laplacian
  module procedure compute_laplacian_2d, compute_laplacian_3d
end interface
interface compute_gradient_cos
  module procedure compute_gradient_cos_2d, compute_gradient_cos_3d
end interface
interface compute_ucos_vcos
  module procedure compute_ucos_vcos_2d, compute_ucos_vcos_3d
end interface
interface compute_vor_div
  module procedure compute_vor_div_2d, compute_vor_div_3d
end interface
interface triangular_truncation
  module procedure triangular_truncation_2d, triangular_truncation_3d
end interface

contains

logical function transforms_are_initialized()
  transforms_are_initialized = core_ready
end function transforms_are_initialized

! ---- domains and tables ------------------------------------------------------------------------------------------------------
subroutine get_grid_domain(is, ie, js, je)
integer, intent(out) :: is, ie, js, je
call need_core('get_grid_domain')
is = 1; ie = nlon; js = js_loc; je = je_loc          ! latitude bands (spec_mpp.F90:61-80): all rows with one rank
end subroutine get_grid_domain
subroutine get_spec_domain(ms, me, ns, ne)
integer, intent(out) :: ms, me, ns, ne
call need_core('get_spec_domain')
ms = 0; me = nfour; ns = 0; ne = nsph
end subroutine get_spec_domain
subroutine get_lat_max(lat_max_out)
integer, intent(out) :: lat_max_out
call need_core('get_lat_max'); lat_max_out = nlat
end subroutine get_lat_max
subroutine get_lon_max(lon_max_out)
integer, intent(out) :: lon_max_out
call need_core('get_lon_max'); lon_max_out = nlon
end subroutine get_lon_max
subroutine get_num_fourier(num_fourier_out)
integer, intent(out) :: num_fourier_out
call need_core('get_num_fourier'); num_fourier_out = nfour
end subroutine get_num_fourier
subroutine get_num_spherical(num_spherical_out)
integer, intent(out) :: num_spherical_out
call need_core('get_num_spherical'); num_spherical_out = nsph
end subroutine get_num_spherical
subroutine get_fourier_inc(fourier_inc_out)
integer, intent(out) :: fourier_inc_out
call need_core('get_fourier_inc'); fourier_inc_out = finc
end subroutine get_fourier_inc
subroutine get_triang_trunc(triang_trunc_out)
logical, intent(out) :: triang_trunc_out
real :: eig(0:nfour, 0:nsph)
call need_core('get_triang_trunc')
triang_trunc_out = triang
end subroutine get_triang_trunc

subroutine get_deg_lon(deg_lon_out)
real, intent(out), dimension(:) :: deg_lon_out
call need_core('get_deg_lon'); call get_table1('deg_lon', deg_lon_out)
end subroutine get_deg_lon
! the latitude tables of THIS process's rows (the reference's getters return (js:je); transforms.F90:853-914), or the whole table when asked for lat_max values
subroutine get_lat_rows(name, a)
character(len=*), intent(in) :: name
real, intent(out) :: a(:)
real, allocatable :: whole(:)
if(size(a) == nlat) then
  call get_table1(name, a)
else if(size(a) == je_loc - js_loc + 1) then
  allocate(whole(nlat)); call get_table1(name, whole); a = whole(js_loc:je_loc)
else
  call error_mesg('get_'//name, 'the argument does not have the number of local latitudes', FATAL)
endif
end subroutine get_lat_rows
subroutine get_deg_lat(deg_lat_out)
real, intent(out), dimension(:) :: deg_lat_out
call need_core('get_deg_lat'); call get_lat_rows('deg_lat', deg_lat_out)
end subroutine get_deg_lat
subroutine get_sin_lat(sin_lat_out)
real, intent(out), dimension(:) :: sin_lat_out
call need_core('get_sin_lat'); call get_lat_rows('sin_lat', sin_lat_out)
end subroutine get_sin_lat
subroutine get_wts_lat(wts_lat_out)
real, intent(out), dimension(:) :: wts_lat_out
call need_core('get_wts_lat'); call get_lat_rows('wts_lat', wts_lat_out)
end subroutine get_wts_lat
subroutine get_cos_lat(cos_lat_out)      ! spherical_fourier.F90:467-484: cos_lat = sqrt(1 - sin_lat**2); cosm = 1/cos; cosm2 = 1/cos**2
real, intent(out), dimension(:) :: cos_lat_out
call get_sin_lat(cos_lat_out); cos_lat_out = sqrt(1 - cos_lat_out*cos_lat_out)
end subroutine get_cos_lat
subroutine get_cosm_lat(cosm_lat_out)
real, intent(out), dimension(:) :: cosm_lat_out
call get_cos_lat(cosm_lat_out); cosm_lat_out = 1./cosm_lat_out
end subroutine get_cosm_lat
subroutine get_cosm2_lat(cosm2_lat_out)
real, intent(out), dimension(:) :: cosm2_lat_out
call get_cos_lat(cosm2_lat_out); cosm2_lat_out = 1./(cosm2_lat_out*cosm2_lat_out)
end subroutine get_cosm2_lat
subroutine get_grid_boundaries(lon_boundaries, lat_boundaries, global)
real, intent(out), dimension(:) :: lon_boundaries, lat_boundaries
logical, intent(in), optional :: global
call need_core('get_grid_boundaries')
if(size(lon_boundaries) /= nlon+1) call error_mesg('get_grid_boundaries','size(lon_boundaries) is incorrect.', FATAL)
if(size(lat_boundaries) /= nlat+1) call error_mesg('get_grid_boundaries','size(lat_boundaries) is incorrect.', FATAL)
call get_table1('lon_boundaries', lon_boundaries); call get_table1('lat_boundaries', lat_boundaries)
end subroutine get_grid_boundaries
subroutine get_eigen_laplacian(eigen_laplacian_out)
real, intent(out), dimension(:,:) :: eigen_laplacian_out
real(c_double), allocatable :: buf(:)
call need_core('get_eigen_laplacian')
allocate(buf(size(eigen_laplacian_out)))
call chk(isca_dyn_get_table(core, cstr('eigen_laplacian'), buf, size(buf, kind=c_size_t)), 'get_eigen_laplacian')
eigen_laplacian_out = reshape(buf, shape(eigen_laplacian_out))
end subroutine get_eigen_laplacian

! gauss_and_legendre.F90:47-183: host tables of the library (no device work; the reference computes them on the host as well)
subroutine compute_legendre(legendre, num_fourier, fourier_inc, num_spherical, sin_lat, n_lat)
integer, intent(in) :: num_fourier, fourier_inc, num_spherical, n_lat
real, intent(in), dimension(n_lat) :: sin_lat
real, intent(out), dimension(0:num_fourier, 0:num_spherical, n_lat) :: legendre
call chk(isca_compute_legendre(int(num_fourier, c_int), int(fourier_inc, c_int), int(num_spherical, c_int), sin_lat, int(n_lat, c_int), legendre), &
         'compute_legendre')
end subroutine compute_legendre
subroutine compute_gaussian(sin_hem, wts_hem, n_hem)
integer, intent(in) :: n_hem
real, intent(out), dimension(n_hem) :: sin_hem, wts_hem
call chk(isca_compute_gaussian(int(n_hem, c_int), sin_hem, wts_hem), 'compute_gaussian')
end subroutine compute_gaussian

! ---- grid <-> spherical ------------------------------------------------------------------------------------------------------
subroutine trans_spherical_to_grid_3d(spherical, grid)
complex, intent(in),  dimension(:,:,:) :: spherical
real,    intent(out), dimension(:,:,:) :: grid
call need_core('trans_spherical_to_grid')
call chk(isca_trans_spherical_to_grid(core, spherical, grid, size(grid, 3, kind=c_int)), 'trans_spherical_to_grid')
end subroutine trans_spherical_to_grid_3d
subroutine trans_spherical_to_grid_2d(spherical, grid)
complex, intent(in),  dimension(:,:) :: spherical
real,    intent(out), dimension(:,:) :: grid
call need_core('trans_spherical_to_grid')
call chk(isca_trans_spherical_to_grid(core, spherical, grid, 1_c_int), 'trans_spherical_to_grid')
end subroutine trans_spherical_to_grid_2d
subroutine trans_grid_to_spherical_3d(grid, spherical, do_truncation)
real,    intent(in),  dimension(:,:,:) :: grid
complex, intent(out), dimension(:,:,:) :: spherical
logical, intent(in), optional :: do_truncation
integer(c_int) :: trunc
call need_core('trans_grid_to_spherical')
trunc = 1; if(present(do_truncation)) trunc = merge(1, 0, do_truncation)
call chk(isca_trans_grid_to_spherical(core, grid, spherical, size(grid, 3, kind=c_int), trunc), 'trans_grid_to_spherical')
end subroutine trans_grid_to_spherical_3d
subroutine trans_grid_to_spherical_2d(grid, spherical, do_truncation)
real,    intent(in),  dimension(:,:) :: grid
complex, intent(out), dimension(:,:) :: spherical
logical, intent(in), optional :: do_truncation
integer(c_int) :: trunc
call need_core('trans_grid_to_spherical')
trunc = 1; if(present(do_truncation)) trunc = merge(1, 0, do_truncation)
call chk(isca_trans_grid_to_spherical(core, grid, spherical, 1_c_int, trunc), 'trans_grid_to_spherical')
end subroutine trans_grid_to_spherical_2d

subroutine trans_filter_3d(grid, filter)
real, intent(inout), dimension(:,:,:) :: grid
real, intent(in), optional, dimension(:,:), target :: filter
real(c_double), allocatable, target :: f(:)
call need_core('trans_filter')
if(present(filter)) then
  allocate(f(size(filter))); f = reshape(filter, (/size(filter)/))
  call chk(isca_trans_filter(core, grid, c_loc(f), size(grid, 3, kind=c_int)), 'trans_filter')
else
  call chk(isca_trans_filter(core, grid, c_null_ptr, size(grid, 3, kind=c_int)), 'trans_filter')
endif
end subroutine trans_filter_3d
subroutine trans_filter_2d(grid, filter)
real, intent(inout), dimension(:,:) :: grid
real, intent(in), optional, dimension(:,:) :: filter
real, dimension(size(grid,1), size(grid,2), 1) :: g3
g3(:,:,1) = grid
call trans_filter_3d(g3, filter)
grid = g3(:,:,1)
end subroutine trans_filter_2d

subroutine divide_by_cos_3d(grid)
real, intent(inout), dimension(:,:,:) :: grid
call need_core('divide_by_cos')
call chk(isca_divide_by_cos(core, grid, size(grid, 3, kind=c_int), 1_c_int), 'divide_by_cos')
end subroutine divide_by_cos_3d
subroutine divide_by_cos_2d(grid)
real, intent(inout), dimension(:,:) :: grid
call need_core('divide_by_cos')
call chk(isca_divide_by_cos(core, grid, 1_c_int, 1_c_int), 'divide_by_cos')
end subroutine divide_by_cos_2d
subroutine divide_by_cos2_3d(grid)
real, intent(inout), dimension(:,:,:) :: grid
call need_core('divide_by_cos2')
call chk(isca_divide_by_cos(core, grid, size(grid, 3, kind=c_int), 2_c_int), 'divide_by_cos2')
end subroutine divide_by_cos2_3d
subroutine divide_by_cos2_2d(grid)
real, intent(inout), dimension(:,:) :: grid
call need_core('divide_by_cos2')
call chk(isca_divide_by_cos(core, grid, 1_c_int, 2_c_int), 'divide_by_cos2')
end subroutine divide_by_cos2_2d

! ---- winds <-> vorticity, divergence; advection ----------------------------------------------------------------------------
subroutine uv_grid_from_vor_div_3d(vor_spec, div_spec, u_grid, v_grid)
complex, intent(in),  dimension(:,:,:) :: vor_spec, div_spec
real,    intent(out), dimension(:,:,:) :: u_grid, v_grid
call need_core('uv_grid_from_vor_div')
call chk(isca_uv_grid_from_vor_div(core, vor_spec, div_spec, u_grid, v_grid, size(u_grid, 3, kind=c_int)), 'uv_grid_from_vor_div')
end subroutine uv_grid_from_vor_div_3d
subroutine uv_grid_from_vor_div_2d(vor_spec, div_spec, u_grid, v_grid)
complex, intent(in),  dimension(:,:) :: vor_spec, div_spec
real,    intent(out), dimension(:,:) :: u_grid, v_grid
call need_core('uv_grid_from_vor_div')
call chk(isca_uv_grid_from_vor_div(core, vor_spec, div_spec, u_grid, v_grid, 1_c_int), 'uv_grid_from_vor_div')
end subroutine uv_grid_from_vor_div_2d
subroutine vor_div_from_uv_grid_3d(u_grid, v_grid, vor_spec, div_spec, triang)
real,    intent(in),  dimension(:,:,:) :: u_grid, v_grid
complex, intent(out), dimension(:,:,:) :: vor_spec, div_spec
logical, intent(in), optional :: triang
call need_core('vor_div_from_uv_grid')
if(present(triang)) then
  if(.not. triang) call error_mesg('vor_div_from_uv_grid','triang = .false. needs a core created with triang_trunc = .false.', FATAL)
endif
call chk(isca_vor_div_from_uv_grid(core, u_grid, v_grid, vor_spec, div_spec, size(u_grid, 3, kind=c_int)), 'vor_div_from_uv_grid')
end subroutine vor_div_from_uv_grid_3d
subroutine vor_div_from_uv_grid_2d(u_grid, v_grid, vor_spec, div_spec, triang)
real,    intent(in),  dimension(:,:) :: u_grid, v_grid
complex, intent(out), dimension(:,:) :: vor_spec, div_spec
logical, intent(in), optional :: triang
call need_core('vor_div_from_uv_grid')
call chk(isca_vor_div_from_uv_grid(core, u_grid, v_grid, vor_spec, div_spec, 1_c_int), 'vor_div_from_uv_grid')
end subroutine vor_div_from_uv_grid_2d
subroutine horizontal_advection_3d(field_spec, u_grid, v_grid, tendency)
complex, intent(in),    dimension(:,:,:) :: field_spec
real,    intent(in),    dimension(:,:,:) :: u_grid, v_grid
real,    intent(inout), dimension(:,:,:) :: tendency
call need_core('horizontal_advection')
call chk(isca_horizontal_advection(core, field_spec, u_grid, v_grid, tendency, size(u_grid, 3, kind=c_int)), 'horizontal_advection')
end subroutine horizontal_advection_3d
subroutine horizontal_advection_2d(field_spec, u_grid, v_grid, tendency)
complex, intent(in),    dimension(:,:) :: field_spec
real,    intent(in),    dimension(:,:) :: u_grid, v_grid
real,    intent(inout), dimension(:,:) :: tendency
call need_core('horizontal_advection')
call chk(isca_horizontal_advection(core, field_spec, u_grid, v_grid, tendency, 1_c_int), 'horizontal_advection')
end subroutine horizontal_advection_2d
function area_weighted_global_mean(field)
real :: area_weighted_global_mean
real, intent(in), dimension(:,:) :: field
real(c_double) :: mean
call need_core('area_weighted_global_mean')
call chk(isca_area_weighted_global_mean(core, field, mean), 'area_weighted_global_mean')
area_weighted_global_mean = mean
end function area_weighted_global_mean

! ---- the two stages of a transform separately (spherical_fourier.F90:177,264; grid_fourier.F90:129,155) -------------------
! fourier(ms:me, lat, lev, 1): one latitude block (the reference's 4th index counts the blocks of its y-decomposition)
subroutine trans_spherical_to_fourier_3d(spherical, fourier)
complex, intent(in),  dimension(:,:,:)   :: spherical
complex, intent(out), dimension(:,:,:,:) :: fourier
call need_core('trans_spherical_to_fourier')
if(size(fourier,2)*size(fourier,4) /= nlat .or. size(fourier,4) /= 1) &
  call error_mesg('trans_spherical_to_fourier','size(fourier,2) must be lat_max and size(fourier,4) 1', FATAL)
call chk(isca_trans_spherical_to_fourier(core, spherical, fourier, size(spherical, 3, kind=c_int)), 'trans_spherical_to_fourier')
end subroutine trans_spherical_to_fourier_3d
subroutine trans_fourier_to_spherical_3d(fourier, spherical)
complex, intent(out), dimension(:,:,:)   :: spherical
complex, intent(in),  dimension(:,:,:,:) :: fourier
call need_core('trans_fourier_to_spherical')
if(size(fourier,2)*size(fourier,4) /= nlat .or. size(fourier,4) /= 1) &
  call error_mesg('trans_fourier_to_spherical','size(fourier,2) must be lat_max and size(fourier,4) 1', FATAL)
call chk(isca_trans_fourier_to_spherical(core, fourier, spherical, size(spherical, 3, kind=c_int)), 'trans_fourier_to_spherical')
end subroutine trans_fourier_to_spherical_3d
! fourier(0:lon_max/2, lat, lev); the device keeps the wavenumbers 0..num_fourier every consumer truncates to: the others return as zero
function trans_grid_to_fourier_3d(grid) result(fourier)
real, intent(in), dimension(:,:,:) :: grid
complex, dimension(0:size(grid,1)/2, size(grid,2), size(grid,3)) :: fourier
complex(c_double_complex), allocatable :: f(:,:,:)
call need_core('trans_grid_to_fourier')
allocate(f(0:nfour, size(grid,2), size(grid,3)))
call chk(isca_trans_grid_to_fourier(core, grid, f, size(grid, 3, kind=c_int)), 'trans_grid_to_fourier')
fourier = (0., 0.)
fourier(0:nfour,:,:) = f
end function trans_grid_to_fourier_3d
function trans_fourier_to_grid_3d(fourier) result(grid)
complex, intent(in), dimension(0:,:,:) :: fourier
real, dimension(2*(size(fourier,1)-1), size(fourier,2), size(fourier,3)) :: grid
complex(c_double_complex), allocatable :: f(:,:,:)
call need_core('trans_fourier_to_grid')
allocate(f(0:nfour, size(fourier,2), size(fourier,3)))
f = fourier(0:nfour,:,:)
call chk(isca_trans_fourier_to_grid(core, f, grid, size(grid, 3, kind=c_int)), 'trans_fourier_to_grid')
end function trans_fourier_to_grid_3d

! ---- spectral operators (spherical.F90:270-600) -----------------------------------------------------------------------------
function compute_laplacian_3d(spherical, power) result(laplacian)
complex, intent(in), dimension(:,:,:) :: spherical
integer, optional :: power
complex, dimension(size(spherical,1), size(spherical,2), size(spherical,3)) :: laplacian
integer(c_int) :: p
call need_core('compute_laplacian')
p = 1; if(present(power)) p = power
call chk(isca_compute_laplacian(core, spherical, laplacian, size(spherical, 3, kind=c_int), p), 'compute_laplacian')
end function compute_laplacian_3d
function compute_laplacian_2d(spherical, power) result(laplacian)
complex, intent(in), dimension(:,:) :: spherical
integer, optional :: power
complex, dimension(size(spherical,1), size(spherical,2)) :: laplacian
integer(c_int) :: p
call need_core('compute_laplacian')
p = 1; if(present(power)) p = power
call chk(isca_compute_laplacian(core, spherical, laplacian, 1_c_int, p), 'compute_laplacian')
end function compute_laplacian_2d
subroutine compute_gradient_cos_3d(spherical, deriv_lon, deriv_lat)
complex, intent(in),  dimension(:,:,:) :: spherical
complex, intent(out), dimension(:,:,:) :: deriv_lon, deriv_lat
call need_core('compute_gradient_cos')
call chk(isca_compute_gradient_cos(core, spherical, deriv_lon, deriv_lat, size(spherical, 3, kind=c_int)), 'compute_gradient_cos')
end subroutine compute_gradient_cos_3d
subroutine compute_gradient_cos_2d(spherical, deriv_lon, deriv_lat)
complex, intent(in),  dimension(:,:) :: spherical
complex, intent(out), dimension(:,:) :: deriv_lon, deriv_lat
call need_core('compute_gradient_cos')
call chk(isca_compute_gradient_cos(core, spherical, deriv_lon, deriv_lat, 1_c_int), 'compute_gradient_cos')
end subroutine compute_gradient_cos_2d
subroutine compute_ucos_vcos_3d(vorticity, divergence, u_cos, v_cos)
complex, intent(in),  dimension(:,:,:) :: vorticity, divergence
complex, intent(out), dimension(:,:,:) :: u_cos, v_cos
call need_core('compute_ucos_vcos')
call chk(isca_compute_ucos_vcos(core, vorticity, divergence, u_cos, v_cos, size(vorticity, 3, kind=c_int)), 'compute_ucos_vcos')
end subroutine compute_ucos_vcos_3d
subroutine compute_ucos_vcos_2d(vorticity, divergence, u_cos, v_cos)
complex, intent(in),  dimension(:,:) :: vorticity, divergence
complex, intent(out), dimension(:,:) :: u_cos, v_cos
call need_core('compute_ucos_vcos')
call chk(isca_compute_ucos_vcos(core, vorticity, divergence, u_cos, v_cos, 1_c_int), 'compute_ucos_vcos')
end subroutine compute_ucos_vcos_2d
subroutine compute_vor_div_3d(u_cos, v_cos, vorticity, divergence)
complex, intent(in),  dimension(:,:,:) :: u_cos, v_cos
complex, intent(out), dimension(:,:,:) :: vorticity, divergence
call need_core('compute_vor_div')
call chk(isca_compute_vor_div(core, u_cos, v_cos, vorticity, divergence, size(u_cos, 3, kind=c_int)), 'compute_vor_div')
end subroutine compute_vor_div_3d
subroutine compute_vor_div_2d(u_cos, v_cos, vorticity, divergence)
complex, intent(in),  dimension(:,:) :: u_cos, v_cos
complex, intent(out), dimension(:,:) :: vorticity, divergence
call need_core('compute_vor_div')
call chk(isca_compute_vor_div(core, u_cos, v_cos, vorticity, divergence, 1_c_int), 'compute_vor_div')
end subroutine compute_vor_div_2d
subroutine triangular_truncation_3d(spherical, trunc)
complex, intent(inout), dimension(:,:,:) :: spherical
integer, intent(in), optional :: trunc
call need_core('triangular_truncation')
if(present(trunc)) call error_mesg('triangular_truncation','a truncation other than the model''s own is not available', FATAL)
call chk(isca_triangular_truncation(core, spherical, size(spherical, 3, kind=c_int)), 'triangular_truncation')
end subroutine triangular_truncation_3d
subroutine triangular_truncation_2d(spherical, trunc)
complex, intent(inout), dimension(:,:) :: spherical
integer, intent(in), optional :: trunc
call need_core('triangular_truncation')
if(present(trunc)) call error_mesg('triangular_truncation','a truncation other than the model''s own is not available', FATAL)
call chk(isca_triangular_truncation(core, spherical, 1_c_int), 'triangular_truncation')
end subroutine triangular_truncation_2d

end module transforms_mod
