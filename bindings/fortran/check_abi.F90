! No GPU needed: the derived types of isca_dyn_c against the library's structs, and a few defaults read back through them.
program check_abi_prog
use iso_c_binding
use isca_dyn_c
implicit none
type(isca_dyn_config) :: cfg
if(.not. check_abi()) then
  print *, 'ABI_MISMATCH'; stop 2
endif
if(isca_dyn_config_default(cfg) /= 0) stop 3
write(*,'(a,5i6)') 'ABI_OK ', cfg%lon_max, cfg%lat_max, cfg%num_fourier, cfg%num_spherical, cfg%num_levels
write(*,'(a,4es16.8)') 'DEFAULTS ', cfg%robert_coeff, cfg%moist%atm_abs, cfg%radius, cfg%valid_range_t(2)
end program check_abi_prog
