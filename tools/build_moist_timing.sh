#!/bin/bash
# Phase-timing builds of the moist kernels (moist.hip: MOIST_TIMING; read by tools/dev/moist_phase_times.py):
#   tools/build_moist_timing.sh  ->  isca_amd/lib/libisca_dyn_mt{1,2,3,5}.so   (the product's other objects, moist.o rebuilt with -DMOIST_TIMING=p)
set -e
cd "$(dirname "$0")/.."
L=isca_amd/lib
python -m isca_amd.build > /dev/null
for p in 1 2 3 5; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DMOIST_TIMING=$p -c isca_amd/csrc/moist.hip -o $L/moist_mt$p.o &
done
wait
OBJS=$(ls $L/*.o | grep -v moist)
for p in 1 2 3 5; do
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libisca_dyn_mt$p.so $OBJS $L/moist_mt$p.o -ldl
done
echo built $L/libisca_dyn_mt{1,2,3,5}.so
