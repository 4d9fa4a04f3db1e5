! tracer_type_mod -- the tracer attribute record of the dynamical core's interface (atmos_spectral/model/tracer_type.F90:22-37):
! the same components, because spectral_dynamics_init fills an array of them for its caller.
module tracer_type_mod
implicit none
private
public :: tracer_type, tracer_type_version, tracer_type_tagname
character(len=128) :: tracer_type_version = 'isca_amd drop-in tracer_type'
character(len=128) :: tracer_type_tagname = 'MI355X'
type tracer_type
  character(len=32) :: name, numerical_representation, advect_horiz, advect_vert, hole_filling
  real :: robert_coeff
end type
end module tracer_type_mod
