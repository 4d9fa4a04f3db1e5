! No GPU needed: the derived types of isca_dyn_c against the library's structs, and a few defaults read back through them.
program check_abi_prog
use iso_c_binding
use isca_dyn_c
use isca_siblings_c
implicit none
type(isca_dyn_config) :: cfg
type(isca_shallow_config) :: sw
type(isca_barotropic_config) :: bt
integer(c_size_t) :: sizes(4)
if(.not. check_abi()) then
  print *, 'ABI_MISMATCH'; stop 2
endif
if(isca_dyn_config_default(cfg) /= 0) stop 3
write(*,'(a,5i6)') 'ABI_OK ', cfg%lon_max, cfg%lat_max, cfg%num_fourier, cfg%num_spherical, cfg%num_levels
if(isca_config_sizes(sizes, 4_c_int) /= 0) stop 4
if(sizes(3) /= c_sizeof(sw) .or. sizes(4) /= c_sizeof(bt)) then
  print *, 'ABI_MISMATCH siblings'; stop 5
endif
if(isca_shallow_config_default(sw) /= 0 .or. isca_barotropic_config_default(bt) /= 0) stop 6
write(*,'(a,3es16.8,i4)') 'SIBLINGS ', sw%h_0, sw%stirring%decay_time, bt%zeta_0, bt%m_0
write(*,'(a,4es16.8)') 'DEFAULTS ', cfg%robert_coeff, cfg%moist%atm_abs, cfg%radius, cfg%valid_range_t(2)
! the members at the end of the struct (a shifted member in front of them would show here)
write(*,'(a,2es16.8,5i4)') 'TAIL ', cfg%tracer_robert_coeff(ISCA_MAX_TRACERS), cfg%tracer_sink(ISCA_MAX_TRACERS), cfg%use_implicit, &
     cfg%tracer_hole_filling(ISCA_MAX_TRACERS), cfg%tracer_sms(1), cfg%tracer_advect_vert(1), cfg%tracer_advect_vert(ISCA_MAX_TRACERS)
end program check_abi_prog
