#!/bin/bash
# atmos_model's loop on the drop-in atmosphere_mod (bindings/fortran/dropin) with N processes, one latitude band each (INTEGRATION.md section 0,
# "A multi-rank Fortran host").  The mpp this image can build has no MPI, so rank and number of ranks travel in the environment (isca_env_rank) and the
# communicator's id through a file; under mpirun / srun / torchrun the launcher's own variables are read instead and only ISCA_COMM_ID_FILE is needed.
#
#   usage: examples/run_fortran_sharded.sh <run directory with input.nml, field_table, diag_table, drive.nml> <N> [ipc]
#
# "ipc": all ranks share GPU 0 through the library's host-staged exchange (verification on a one-GPU box); otherwise rank r drives GPU r over RCCL.
# RESTART/*.res.nc.NNNN of a run become INPUT/ of the next one (every rank reads its own piece).
set -e
RUN=${1:?run directory}; N=${2:?number of ranks}; MODE=${3:-rccl}
HERE=$(cd "$(dirname "$0")/.." && pwd)
EXE=$HERE/oracle/_ref/drive_atmos_model_gpu.x          # built by: python oracle/build_ref.py dropin_atmos
[ -x "$EXE" ] || { echo "$EXE not built"; exit 1; }
export ISCA_WORLD_SIZE=$N ISCA_COMM_ID_FILE=$RUN/comm_id HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f "$ISCA_COMM_ID_FILE"
cd "$RUN"
pids=()
for r in $(seq 0 $((N - 1))); do
  if [ "$MODE" = ipc ]; then L=0; export ISCA_COMM=ipc; else L=$r; fi
  ( ulimit -s unlimited; ISCA_RANK=$r ISCA_LOCAL_RANK=$L exec "$EXE" > rank$r.log 2>&1 ) &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
grep -h "DRIVE_ROWS\|DRIVE_STATE" rank*.log
exit $rc
