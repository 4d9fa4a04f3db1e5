#!/bin/bash
# The library with the measured-and-rejected variants compiled in (-DISCA_EXPERIMENTS: kernels.h exp_env; HISTORY.md lists them with their numbers):
#   tools/build_experiments.sh  ->  isca_amd/lib/libisca_dyn_exp.so   (select with ISCA_DYN_LIB=<that path>; the ISCA_* switches of HISTORY.md then work)
set -e
cd "$(dirname "$0")/.."
L=isca_amd/lib
mkdir -p $L/exp
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DISCA_EXPERIMENTS -Wno-unused-result"
for u in kernels legendre api; do /opt/rocm/bin/hipcc $FLAGS -c isca_amd/csrc/$u.hip -o $L/exp/$u.o & done
/opt/rocm/bin/hipcc $FLAGS -ffp-contract=off -c isca_amd/csrc/moist.hip -o $L/exp/moist.o &
/opt/rocm/bin/hipcc $FLAGS -c isca_amd/csrc/comm_peer.hip -o $L/exp/comm_peer.o &
for u in comm comm_ipc restart_nc history_nc; do /opt/rocm/bin/hipcc $FLAGS -c isca_amd/csrc/$u.cpp -o $L/exp/$u.o & done
for u in tables topog; do /opt/rocm/bin/hipcc $FLAGS -ffp-contract=off -c isca_amd/csrc/$u.cpp -o $L/exp/$u.o & done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libisca_dyn_exp.so $L/exp/*.o -ldl
echo built $L/libisca_dyn_exp.so
