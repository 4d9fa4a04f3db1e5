"""diag_table and history files for the dynamics diagnostics.

Counterpart, for the fields of the hot path, of the harness's DiagTable (src/extra/python/isca/diagtable.py) and of what
diag_manager does with the send_data calls of spectral_diagnostics (atmos_spectral/model/spectral_dynamics.F90:1554-1867):
time means over the output interval, one record per interval, written as netCDF-3 with the reference's dimension and
variable names (lon, lat, pfull, phalf, time; ps, ucomp, vcomp, temp, vor, div, ...; static pk, bk).
The sums are accumulated on the device every step (isca_dyn_diag_select / isca_dyn_diag_read).
"""
from __future__ import annotations

import os
import numpy as np
from scipy.io import netcdf_file

from .dyncore import IscaError

# name -> (long name, units) as registered by the reference (spectral_dynamics.F90:1604-1690)
FIELDS = {
    "ps": ("surface pressure", "pascals"), "ucomp": ("zonal wind component", "m/sec"),
    "vcomp": ("meridional wind component", "m/sec"), "temp": ("temperature", "deg_k"), "vor": ("vorticity", "sec**-1"),
    "div": ("divergence", "sec**-1"), "omega": ("dp/dt vertical velocity", "Pa/sec"), "sphum": ("specific humidity", "kg/kg"),
    "ucomp_sq": ("zonal wind squared", "(m/sec)**2"), "vcomp_sq": ("meridional wind squared", "(m/sec)**2"),
    "ucomp_vcomp": ("zonal wind * meridional wind", "(m/sec)**2"), "temp_sq": ("temperature squared", "deg_k**2"),
    "ucomp_temp": ("zonal wind * temperature", "m*K/sec"), "vcomp_temp": ("meridional wind * temperature", "m*K/sec"),
    "omega_sq": ("omega squared", "(Pa/sec)**2"), "omega_temp": ("dp/dt * temperature", "Pa*K/sec"),
    "ucomp_omega": ("vertical * zonal wind", "m*Pa/sec**2"), "vcomp_omega": ("vertical * meridional wind", "m*Pa/sec**2"),
    "vcomp_vor": ("meridional wind * vorticity", "m/sec**2"), "wspd": ("wind speed", "m/sec"),
}
# 2-D fields of the moist physics package and the module that registers them (idealized_moist_phys.F90:672, mixed_layer.F90:359)
MOIST_FIELDS = {"precipitation": ("atmosphere", "precipitation from resolved, parameterised and snow", "kg/m/m/s"),
                "t_surf": ("mixed_layer", "surface temperature", "K")}
FIELDS.update({k: v[1:] for k, v in MOIST_FIELDS.items()})
TWO_D = ("ps",) + tuple(MOIST_FIELDS)
STATIC = {"pk": ("vertical coordinate pressure values", "pascals"), "bk": ("vertical coordinate sigma values", "none")}
_SECONDS = {"seconds": 1, "minutes": 60, "hours": 3600, "days": 86400}


class DiagTable:
    """add_file / add_field as in diagtable.py:60-120 (one output file is supported per table entry)."""

    def __init__(self, base_date=(0, 0, 0, 0, 0, 0)):
        # the table's base-date line: "0 0 0 0 0 0" without a calendar (the harness's default, diagtable.py:13-17), "0001 1 1 0 0 0" with one;
        # diag_manager writes it into the time axis' units (diag_util.F90:1582-1585)
        self.base_date = tuple(int(x) for x in base_date)
        self.files: dict = {}

    def add_file(self, name, freq, units="hours", time_units=None):
        if units not in _SECONDS:
            raise IscaError(f"diag_table: unsupported frequency units {units!r}")
        self.files[name] = {"name": name, "freq": freq, "units": units, "time_units": time_units or units, "fields": [], "base_date": self.base_date}

    def add_field(self, module, name, time_avg=False, files=None):
        if name in MOIST_FIELDS:
            if module != MOIST_FIELDS[name][0]:
                raise IscaError(f"diag_table: field {name!r} belongs to module {MOIST_FIELDS[name][0]!r}")
        elif module != "dynamics":
            raise IscaError(f"diag_table: module {module!r} is outside the dynamical core (only 'dynamics', and precipitation / t_surf "
                            "of the moist package)")
        if name not in FIELDS and name not in STATIC:
            raise IscaError(f"diag_table: unknown dynamics field {name!r}")
        for f in (files or list(self.files)):
            self.files[f]["fields"].append({"name": name, "time_avg": bool(time_avg)})

    def is_valid(self):
        return len(self.files) > 0

    def copy(self):
        import copy
        d = DiagTable(self.base_date)
        d.files = copy.deepcopy(self.files)
        return d


class DiagCollector:
    """The device keeps ONE set of running sums per handle.  With several history files (different intervals, different field lists)
    every file needs its own means, so the sums are taken off the device once per chunk of steps -- the union of the fields of all
    files, read and reset in one go -- and each History adds the chunk to its own host-side sums (diag_manager keeps one buffer per
    output field and file for the same reason)."""

    def __init__(self, core, histories):
        self.core, self.histories = core, list(histories)
        self.names = sorted({nm for h in self.histories for nm in h.names})
        for h in self.histories:
            h.collector = self
        if self.names:
            core.diag_select(self.names)

    def after_steps(self, nsteps: int):
        """Call after every `nsteps` steps; nsteps must divide every file's steps per interval."""
        sums, cnt = {}, nsteps
        for nm in self.names:
            mean, cnt = self.core.diag_mean(nm)
            sums[nm] = mean * cnt
        if self.names:
            if cnt != nsteps:
                raise IscaError(f"diagnostics: the device accumulated {cnt} steps, {nsteps} expected")
            self.core.diag_reset(self.names[0])
        for h in self.histories:
            h.add_chunk(nsteps, sums)

    def close(self):
        for h in self.histories:
            h.close()
        if self.names:
            self.core.diag_select("")


class History:
    """One history file of a run: sums every step (on the device, handed over chunk by chunk), one record per output interval.
    A History on its own (no DiagCollector) owns the device sums: only one such file per handle."""

    def __init__(self, core, table_file: dict, dt_atmos: float, path: str, start_seconds: float = 0.0):
        self.core, self.spec, self.dt, self.path = core, table_file, float(dt_atmos), path
        self.interval = table_file["freq"] * _SECONDS[table_file["units"]]
        if self.interval % dt_atmos:
            raise IscaError("diag_table: output interval must be a multiple of dt_atmos")
        self.every = int(self.interval // dt_atmos)
        self.names = [f["name"] for f in table_file["fields"] if f["name"] in FIELDS]
        self.avg = {f["name"]: f["time_avg"] for f in table_file["fields"]}
        self.static = [f["name"] for f in table_file["fields"] if f["name"] in STATIC]
        self.t0 = float(start_seconds)
        self.elapsed_steps = 0
        self.records = []                       # (t1, t2, {name: array})
        self.collector = None
        self._sums, self._nsum = {}, 0
        self._own = None
        self.standalone()                       # owns the device sums until a DiagCollector takes the file over

    def after_steps(self, nsteps: int):
        """Stand-alone use: call after every `nsteps` steps of the model."""
        if self.collector is not self._own:
            raise IscaError("History.after_steps: this file belongs to a DiagCollector; call the collector's after_steps")
        self._own.after_steps(nsteps)

    def standalone(self):
        """A single file owning the device sums (what a one-file diag_table amounts to)."""
        self._own = DiagCollector(self.core, [self])
        return self

    def add_chunk(self, nsteps: int, sums: dict):
        for nm in self.names:
            if self.avg[nm]:
                self._sums[nm] = sums[nm] if nm not in self._sums else self._sums[nm] + sums[nm]
        self._nsum += nsteps
        self.elapsed_steps += nsteps
        if self.elapsed_steps % self.every == 0:
            self._flush()

    _STATE = {"ucomp": "ug", "vcomp": "vg", "temp": "tg", "ps": "psg", "vor": "vorg", "div": "divg", "omega": "wg_full", "sphum": "tr",
              "precipitation": "precip", "t_surf": "t_surf"}

    def _flush(self):
        rec = {}
        for nm in self.names:
            if self.avg[nm]:
                rec[nm] = self._sums[nm] / self._nsum
            elif nm in self._STATE:             # instantaneous sample at the end of the interval
                rec[nm] = self.core.get(self._STATE[nm])
            else:
                raise IscaError(f"diag_table: {nm} is only available as a time average")
        self._sums, self._nsum = {}, 0
        t2 = self.t0 + self.elapsed_steps * self.dt
        self.records.append((t2 - self.interval, t2, rec))

    def close(self):
        c, tu = self.core, self.spec["time_units"]
        scale = _SECONDS[tu]
        f = netcdf_file(self.path, "w", version=2)
        f.createDimension("time", None)
        for nm, n, vals, units, axis in (("lon", c.I, c.table("deg_lon"), "degrees_E", "X"), ("lat", c.J, c.table("deg_lat"), "degrees_N", "Y")):
            f.createDimension(nm, n)
            v = f.createVariable(nm, "d", (nm,)); v[:] = vals; v.units = units; v.cartesian_axis = axis
        pk, bk = c.table("pk"), c.table("bk")
        p_half = (pk + bk * c.cfg.reference_sea_level_press) / 100.0            # approx. pressure levels, hPa (:1583-1589)
        lnp = np.log(np.where(p_half > 0, p_half, 1.0))
        p_full = np.empty(c.L)
        for k in range(c.L):                                                     # Simmons-Burridge full levels of the reference surface pressure
            if p_half[k] == 0.0:
                p_full[k] = np.exp(lnp[k + 1] - 1.0)
            else:
                p_full[k] = np.exp(lnp[k + 1] - (1.0 - p_half[k] * (lnp[k + 1] - lnp[k]) / (p_half[k + 1] - p_half[k])))
        for nm, n, vals in (("phalf", c.L + 1, p_half), ("pfull", c.L, p_full)):
            f.createDimension(nm, n)
            v = f.createVariable(nm, "d", (nm,)); v[:] = vals; v.units = "hPa"; v.cartesian_axis = "Z"; v.positive = "down"
        tv = f.createVariable("time", "d", ("time",)); tv.units = "%s since %04d-%02d-%02d %02d:%02d:%02d" % ((tu,) + tuple(self.spec.get("base_date", (0, 0, 0, 0, 0, 0)))); tv.cartesian_axis = "T"
        t1v = f.createVariable("average_T1", "d", ("time",)); t2v = f.createVariable("average_T2", "d", ("time",))
        dtv = f.createVariable("average_DT", "d", ("time",))
        for nm in self.static:
            v = f.createVariable(nm, "d", ("phalf",)); v[:] = c.table(nm); v.long_name, v.units = STATIC[nm]
        out = {}
        for nm in self.names:
            dims = ("time", "lat", "lon") if nm in TWO_D else ("time", "pfull", "lat", "lon")
            v = f.createVariable(nm, "d", dims); v.long_name, v.units = FIELDS[nm]
            if self.avg[nm]:
                v.cell_methods = "time: mean"; v.time_avg_info = "average_T1,average_T2,average_DT"
            out[nm] = v
        for r, (t1, t2, rec) in enumerate(self.records):
            tv[r] = 0.5 * (t1 + t2) / scale; t1v[r] = t1 / scale; t2v[r] = t2 / scale; dtv[r] = (t2 - t1) / scale
            for nm in self.names:
                out[nm][r] = rec[nm]
        f.close()
        if self.collector is self._own and self.names:
            self.core.diag_select("")
