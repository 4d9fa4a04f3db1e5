"""Run segmentation with restart chaining, in process.

Host-side counterpart of the reference's Python harness for the hot path
(src/extra/python/isca/experiment.py:60-364): one `Experiment` owns a work directory (`run/` with
`INPUT/` and `RESTART/`), a data directory with one folder per run (`run0001`, ...) and the restart
archives (`restarts/res0001.tar.gz`, ...).  `run(i)` is one segment of `main_nml` length: it starts from
restart archive i-1 (or cold, for i == 1), advances the GPU core, and archives the new restart.  Where the
reference renders run.sh and launches the Fortran executable under mpirun, this drives
atmosphere_init / atmosphere / atmosphere_end of `isca_amd.atmosphere` directly.
The `diag_table` of the dynamics fields (time means accumulated on the device) is written as netCDF history files
into the run's data folder (`isca_amd/diag.py`).
"""
from __future__ import annotations

import copy
import logging
import math
import os
import shutil
import tarfile

from . import atmosphere as atm
from .diag import DiagCollector, DiagTable, History
from .dyncore import RESOLUTIONS, IscaError


class FailedRunError(Exception):
    pass


class Experiment:
    RESOLUTIONS = RESOLUTIONS                     # experiment.py:29-57
    runfmt = "run%04d"
    restartfmt = "res%04d.tar.gz"

    def __init__(self, name: str, workbase: str, database: str | None = None):
        self.name = name
        self.workdir = os.path.join(workbase, "experiment", name)
        self.rundir = os.path.join(self.workdir, "run")
        self.datadir = os.path.join(database if database else os.path.join(workbase, "data"), name)
        self.restartdir = os.path.join(self.datadir, "restarts")
        self.namelist: dict = {}
        self.diag_table = DiagTable()                 # experiment.py:79; history files land in the run's data folder
        self.resolution: str | None = None
        self.field_table_file: str | None = None      # experiment.py:75: the run's field_table; None = the dry default (one grid tracer, sphum)
        self.log = logging.getLogger("isca_amd.experiment")

    # ---- namelist handling (experiment.py:121-143)
    def set_resolution(self, res: str, num_levels: int | None = None):
        delta = dict(self.RESOLUTIONS[res])
        if num_levels is not None:
            delta["num_levels"] = num_levels
        self.update_namelist({"spectral_dynamics_nml": delta})

    def update_namelist(self, new_vals: dict):
        for sec, vals in new_vals.items():
            self.namelist.setdefault(sec, {}).update(vals)

    def write_namelist(self, outdir: str):
        def fmt(v):
            if isinstance(v, bool):
                return ".true." if v else ".false."
            if isinstance(v, str):
                return f"'{v}'"
            if isinstance(v, (list, tuple)):
                return ", ".join(fmt(x) for x in v)
            return repr(v)
        with open(os.path.join(outdir, "input.nml"), "w") as f:
            for sec, vals in self.namelist.items():
                f.write(f"&{sec}\n")
                for k, v in vals.items():
                    f.write(f"    {k} = {fmt(v)}\n")
                f.write("/\n\n")

    def get_restart_file(self, i: int) -> str:
        return os.path.join(self.restartdir, self.restartfmt % i)

    def get_outputdir(self, run: int) -> str:
        return os.path.join(self.datadir, self.runfmt % run)

    def check_for_existing_output(self, i: int) -> bool:
        return os.path.isdir(self.get_outputdir(i))

    def clear_rundir(self):
        shutil.rmtree(self.rundir, ignore_errors=True)
        os.makedirs(self.rundir)

    def steps_per_run(self) -> int:
        """Length of one segment: main_nml days/hours/minutes/seconds over dt_atmos (atmos_model.F90:285-300)."""
        m = self.namelist.get("main_nml", {})
        seconds = ((m.get("days", 0) * 24 + m.get("hours", 0)) * 60 + m.get("minutes", 0)) * 60 + m.get("seconds", 0)
        dt = m.get("dt_atmos", 0)
        if dt <= 0 or seconds <= 0 or seconds % dt:
            raise IscaError("main_nml: run length must be a positive multiple of dt_atmos")
        return int(seconds // dt)

    # ---- one segment (experiment.py:198-346)
    def run(self, i: int, restart_file: str | None = None, use_restart: bool = True, overwrite_data: bool = False):
        self.clear_rundir()
        indir, resdir, outdir = os.path.join(self.rundir, "INPUT"), os.path.join(self.rundir, "RESTART"), self.get_outputdir(i)
        if self.check_for_existing_output(i):
            if overwrite_data:
                self.log.warning("Data for run %d already exists and overwrite_data is True. Overwriting.", i)
                shutil.rmtree(outdir)
            else:
                self.log.warning("Data for run %d already exists but overwrite_data is False. Stopping.", i)
                return False
        for d in (indir, resdir, self.restartdir):
            os.makedirs(d, exist_ok=True)
        self.write_namelist(self.rundir)
        if use_restart and not restart_file and i == 1:
            use_restart = False                      # run 1 spins up from the namelist's initial conditions
        if use_restart:
            restart_file = restart_file or self.get_restart_file(i - 1)
            if not os.path.isfile(restart_file):
                raise IOError("Restart file not found, expecting file %r" % restart_file)
            self.extract_restart_archive(restart_file, indir)
        nsteps = self.steps_per_run()
        dt = self.namelist["main_nml"]["dt_atmos"]
        collector = None
        try:
            ft = None
            if self.field_table_file is not None:
                with open(self.field_table_file) as f:
                    ft = f.read()
            core = atm.atmosphere_init(copy.deepcopy(self.namelist), run_dir=self.rundir, field_table=ft)
            hist = [History(core, spec, dt, os.path.join(self.rundir, name + ".nc"), start_seconds=(i - 1) * nsteps * dt)
                    for name, spec in self.diag_table.files.items() if spec["fields"]]
            collector = DiagCollector(core, hist)         # one set of device sums, shared by all files of the table
            chunk = nsteps
            for h in hist:
                chunk = math.gcd(chunk, h.every)
            for _ in range(nsteps // chunk):
                atm.atmosphere(chunk)
                collector.after_steps(chunk)
            collector.close()
            atm.atmosphere_end()
        except IscaError as e:
            # spectral_dynamics_nml: graceful_shutdown (spectral_dynamics.F90:976-1005): the diagnostics are ended -- the history files get the records
            # accumulated so far -- before the error is raised; they stay in the run directory, like everything else of a failed run
            if collector is not None and self.namelist.get("spectral_dynamics_nml", {}).get("graceful_shutdown", False):
                try:
                    collector.close()
                except Exception as e2:       # (the failure that matters is e)
                    self.log.warning("graceful_shutdown: closing the history files failed: %s", e2)
            atm.atmosphere_end()
            self.log.error("Run %d failed: %s", i, e)
            raise FailedRunError(str(e))
        os.makedirs(outdir)
        for name, spec in self.diag_table.files.items():          # experiment.py:325-330: history files into the data folder
            src = os.path.join(self.rundir, name + ".nc")
            if os.path.exists(src):
                shutil.copy(src, os.path.join(outdir, name + ".nc"))
        self.make_restart_archive(self.get_restart_file(i), resdir)
        shutil.rmtree(resdir)
        self.write_namelist(outdir)
        self.clear_rundir()
        return True

    def make_restart_archive(self, archive_file: str, restart_directory: str):
        with tarfile.open(archive_file, "w:gz") as tar:
            tar.add(restart_directory, arcname=".")

    def extract_restart_archive(self, archive_file: str, input_directory: str):
        with tarfile.open(archive_file, "r:gz") as tar:
            tar.extractall(path=input_directory)

    def delete_restart(self, run: int):
        f = self.get_restart_file(run)
        if os.path.isfile(f):
            os.remove(f)

    def derive(self, new_experiment_name: str):
        e = Experiment(new_experiment_name, os.path.dirname(os.path.dirname(self.workdir)))
        e.datadir = os.path.join(os.path.dirname(self.datadir), new_experiment_name)
        e.restartdir = os.path.join(e.datadir, "restarts")
        e.namelist = copy.deepcopy(self.namelist)
        e.diag_table = self.diag_table.copy()
        return e
