#!/bin/bash
# Build an experimental variant of the library with extra -D flags on kernels.hip:
#   tools/build_variant.sh NAME "-DEXP_FOO=1"  ->  isca_amd/lib/libisca_dyn_NAME.so   (select with ISCA_DYN_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
python -m isca_amd.build > /dev/null
L=isca_amd/lib
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -c isca_amd/csrc/kernels.hip -o $L/kernels_$1.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libisca_dyn_$1.so $L/tables.o $L/comm.o $L/kernels_$1.o $L/api.o -ldl
echo built $L/libisca_dyn_$1.so
