#!/bin/bash
# Build an experimental variant of the library with extra -D flags on one translation unit (default kernels.hip):
#   tools/build_variant.sh NAME "-DEXP_FOO=1" [unit]  ->  isca_amd/lib/libisca_dyn_NAME.so   (select with ISCA_DYN_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
python -m isca_amd.build > /dev/null
L=isca_amd/lib
U=${3:-kernels}
EXTRA=""
[ "$U" = "moist" ] && EXTRA="-ffp-contract=off"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $EXTRA $2 -c isca_amd/csrc/$U.hip -o $L/${U}_$1.o
OBJS=""
for o in tables comm comm_peer comm_ipc restart_nc history_nc topog kernels legendre moist api; do
  if [ "$o" = "$U" ]; then OBJS="$OBJS $L/${U}_$1.o"; else OBJS="$OBJS $L/$o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libisca_dyn_$1.so $OBJS -ldl
echo built $L/libisca_dyn_$1.so
