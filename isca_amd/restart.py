"""Restart files of the spectral core: the reference's variable set in netCDF-3.

Replaces, for the hot path, the restart branches of
  read_restart_or_do_coldstart   (src/atmos_spectral/model/spectral_dynamics.F90:509-575)
  spectral_dynamics_end          (spectral_dynamics.F90:1502-1531)
  atmosphere_init / _end         (src/atmos_spectral/driver/solo/atmosphere.F90:197-223, 362-375)
  mixed_layer_init / _end        (src/atmos_spectral/driver/solo/mixed_layer.F90:324-327, :813) -- moist package only: t_surf in
                                 <dir>/mixed_layer.res.nc
Files: <dir>/spectral_dynamics.res.nc and <dir>/atmosphere.res.nc with the reference's variable names
(`vors_real`, `vors_imag`, ..., `ug`, `psg`, `<tracer>`, `vorg`, `divg`, `surf_geopotential`, `previous`,
`current`, `pk`, `bk`; `time_pointers`, `wg_full`), two records along `Time` (one per leapfrog time level,
1-based `previous`/`current` pointing into them) and fms_io's axis naming (`xaxis_N`, `yaxis_N`, `zaxis_N`).
The reader goes by variable name only, as `read_data`/`field_size` do.  All data fp64, written exactly:
run(N) == run(N/2) + write + read + run(N/2) bit for bit (tests/test_gpu_parity.py).
"""
from __future__ import annotations

import os
import numpy as np
from scipy.io import netcdf_file

from .dyncore import DynCore, IscaError

SPEC3 = ("vors", "divs", "ts")
GRID3 = ("ug", "vg", "tg")


class _Writer:
    """One fms_io-style restart file: every variable is (Time, zaxis, yaxis, xaxis)."""

    def __init__(self, path):
        self.f = netcdf_file(path, "w", version=2)
        self.f.createDimension("Time", None)
        self.axes = {"x": [], "y": [], "z": []}
        self.nrec = 0

    def _axis(self, kind, n):
        sizes = self.axes[kind]
        if n not in sizes:
            sizes.append(n)
            name = f"{kind}axis_{len(sizes)}"
            self.f.createDimension(name, n)
            v = self.f.createVariable(name, "d", (name,))
            v[:] = np.arange(1, n + 1, dtype=np.float64)
            v.cartesian_axis = kind.upper()
        return f"{kind}axis_{sizes.index(n) + 1}"

    def put(self, name, levels):
        """levels: list of arrays (one per Time record), each of rank <= 3 ([z,] [y,] x)."""
        arrs = [np.atleast_1d(np.asarray(a, dtype=np.float64)) for a in levels]
        a0 = arrs[0]
        shp = (1,) * (3 - a0.ndim) + a0.shape
        dims = ("Time", self._axis("z", shp[0]), self._axis("y", shp[1]), self._axis("x", shp[2]))
        v = self.f.createVariable(name, "d", dims)
        for t, a in enumerate(arrs):
            v[t] = a.reshape(shp)
        self.nrec = max(self.nrec, len(arrs))

    def close(self):
        tv = self.f.createVariable("Time", "d", ("Time",))
        tv[:self.nrec] = np.arange(1, self.nrec + 1, dtype=np.float64)
        tv.cartesian_axis = "T"
        self.f.close()


def _levels(core: DynCore, name):
    """Both storage slots of a two-level array in file order (record nt = storage slot nt)."""
    prev, cur = core.info("previous"), core.info("current")
    out = [None, None]
    out[cur] = core.get(name, 1)
    if prev != cur:
        out[prev] = core.get(name, 0)
    else:
        out[1 - cur] = out[cur]            # cold start: both levels hold the same values (:617-625)
    return out


def _tracers(core: DynCore, tracer_name: str | None):
    """(file name, grid state name, atmosphere copy name, spectral state name or None) per prognostic tracer (spectral_dynamics.F90:1520-1527)"""
    if not core.info("tracer"):
        return []
    names = list(core.tracer_names)
    if tracer_name is not None:
        names[0] = tracer_name
    out = [(names[0], "tr", "tr_atm", None)]
    for k in range(1, core.cfg.num_tracers):
        out.append((names[k], f"tr{k + 1}", f"tr_atm{k + 1}", f"trs{k + 1}" if core.cfg.tracer_spectral[k] else None))
    return out


def write_restart(core: DynCore, directory: str, tracer_name: str | None = None):
    """spectral_dynamics_end + atmosphere_end: write both restart files into `directory`."""
    if core.cfg.world_size != 1:
        raise IscaError("write_restart: gather the bands on one rank first (world_size == 1 only)")
    os.makedirs(directory, exist_ok=True)
    prev, cur = core.info("previous"), core.info("current")
    tracers = _tracers(core, tracer_name)

    w = _Writer(os.path.join(directory, "spectral_dynamics.res.nc"))
    w.put("previous", [float(prev + 1)] * 2)
    w.put("current", [float(cur + 1)] * 2)
    w.put("pk", [core.table("pk")] * 2)
    w.put("bk", [core.table("bk")] * 2)
    for nm in SPEC3 + ("ln_ps",):
        lv = _levels(core, nm)
        w.put(nm + "_real", [np.ascontiguousarray(a.real) for a in lv])
        w.put(nm + "_imag", [np.ascontiguousarray(a.imag) for a in lv])
    for nm in GRID3 + ("psg",):
        w.put(nm, _levels(core, nm))
    for name, grid, _, spec in tracers:
        w.put(name, _levels(core, grid))
        if spec is not None:
            lv = _levels(core, spec)
            w.put(name + "_real", [np.ascontiguousarray(a.real) for a in lv])
            w.put(name + "_imag", [np.ascontiguousarray(a.imag) for a in lv])
    w.put("vorg", [core.get("vorg")])
    w.put("divg", [core.get("divg")])
    w.put("surf_geopotential", [core.get("surf_geopotential")])
    w.close()

    w = _Writer(os.path.join(directory, "atmosphere.res.nc"))
    w.put("time_pointers", [np.array([prev + 1.0, cur + 1.0])] * 2)
    for nm in GRID3 + ("psg",):
        w.put(nm, _levels(core, nm))
    for name, _, atm_copy, _ in tracers:
        w.put(name, _levels(core, atm_copy))
    w.put("wg_full", [core.get("wg_full")])
    w.close()

    if core.cfg.physics == 1:                                    # mixed_layer_end
        w = _Writer(os.path.join(directory, "mixed_layer.res.nc"))
        w.put("t_surf", [core.get("t_surf")])
        w.close()


def _read_all(path):
    f = netcdf_file(path, "r", mmap=False)
    try:
        return {k: np.array(v[:], dtype=np.float64) for k, v in f.variables.items()}
    finally:
        f.close()


def restart_exists(directory: str) -> bool:
    return os.path.exists(os.path.join(directory, "spectral_dynamics.res.nc"))


def read_restart(core: DynCore, directory: str, tracer_name: str | None = None):
    """The restart branch of spectral_dynamics_init/atmosphere_init: load both time levels into the device
    state, restore the leapfrog pointers and rebuild the derived grid fields."""
    if core.cfg.world_size != 1:
        raise IscaError("read_restart: world_size == 1 only (scatter with ShardedDynCore.load_from)")
    sd = _read_all(os.path.join(directory, "spectral_dynamics.res.nc"))
    at_path = os.path.join(directory, "atmosphere.res.nc")
    at = _read_all(at_path) if os.path.exists(at_path) else None
    L, J, I, N1, M1 = core.L, core.J, core.I, core.N1, core.M1

    siz = sd["vors_real"].shape                      # (Time, lev, n, m)
    if (siz[3] - 1, siz[2] - 1, siz[1]) != (M1 - 1, N1 - 1, L):
        raise IscaError("spectral_dynamics_init: Resolution of restart data does not match resolution specified on "
                        f"namelist. Restart data: num_fourier={siz[3] - 1}, num_spherical={siz[2] - 1}, num_levels={siz[1]}"
                        f"  Namelist: num_fourier={M1 - 1}, num_spherical={N1 - 1}, num_levels={L}")
    siz = sd["ug"].shape
    if (siz[3], siz[2]) != (I, J):
        raise IscaError("spectral_dynamics_init: Resolution of restart data does not match resolution specified on "
                        f"namelist. Restart data: lon_max={siz[3]}, lat_max={siz[2]}  Namelist: lon_max={I}, lat_max={J}")
    if at is not None and at["ug"].shape[2:] != (J, I):
        raise IscaError("atmosphere_init: Resolution of restart data does not match resolution specified on namelist.")
    prev = int(round(float(sd["previous"].ravel()[0]))) - 1
    cur = int(round(float(sd["current"].ravel()[0]))) - 1
    if at is not None:
        tp = at["time_pointers"].reshape(at["time_pointers"].shape[0], -1)[0]
        if (int(tp[0]) - 1, int(tp[1]) - 1) != (prev, cur):
            raise IscaError("read_restart: time pointers of atmosphere.res and spectral_dynamics.res differ")
    for nm in ("pk", "bk"):
        if not np.array_equal(sd[nm].reshape(sd[nm].shape[0], -1)[0], core.table(nm)):
            raise IscaError(f"read_restart: {nm} of the restart file differs from the vertical coordinate of the namelist")
    core.set_surf_geopotential(np.asarray(sd["surf_geopotential"]).reshape(-1, J, I)[0])       # spectral_dynamics.F90:575: the restart file's topography, not get_topography's

    core.set_time_pointers(prev, cur, 0 if prev == cur else 1)
    tracers = _tracers(core, tracer_name)
    for nt in (0, 1):
        tl = 0 if (nt == prev and prev != cur) else 1
        if prev != cur or nt == cur:
            for nm in SPEC3:
                core.set(nm, sd[nm + "_real"][nt] + 1j * sd[nm + "_imag"][nt], tl)
            core.set("ln_ps", sd["ln_ps_real"][nt].reshape(N1, M1) + 1j * sd["ln_ps_imag"][nt].reshape(N1, M1), tl)
            for nm in GRID3:
                if at is not None and not np.array_equal(at[nm][nt], sd[nm][nt]):
                    raise IscaError(f"read_restart: {nm} of atmosphere.res and spectral_dynamics.res differ")
                core.set(nm, sd[nm][nt], tl)
            core.set("psg", sd["psg"][nt].reshape(J, I), tl)
            for name, grid, atm_copy, spec in tracers:
                if name not in sd or (spec is not None and name + "_real" not in sd):
                    raise IscaError(f"read_restart: tracer {name} not in the restart file")
                core.set(grid, sd[name][nt], tl)
                core.set(atm_copy, (at if at is not None else sd)[name][nt], tl)
                if spec is not None:
                    core.set(spec, sd[name + "_real"][nt] + 1j * sd[name + "_imag"][nt], tl)
    if at is not None:
        core.set("wg_full", at["wg_full"][0])
    if core.cfg.physics == 1:                                    # mixed_layer_init: restart file, else the prescribed distribution
        ml_path = os.path.join(directory, "mixed_layer.res.nc")
        if os.path.exists(ml_path):
            ts = _read_all(ml_path)["t_surf"]
            if ts.shape[-2:] != (J, I):
                raise IscaError("mixed_layer_init: resolution of mixed_layer.res does not match the namelist")
            core.set("t_surf", ts.reshape(-1, J, I)[0])
    core.refresh_derived()
    # vorg, divg are restart variables of the reference too (spectral_dynamics.F90:1518-1519, read back :566-567): with raw_filter_coeff /= 1
    # they belong to the new level BEFORE the filter's adjustment (:933-934 vs :1031), which the adjusted spectral state cannot give back
    if "vorg" in sd and "divg" in sd:
        core.set("vorg", np.asarray(sd["vorg"]).reshape(-1, L, J, I)[0])
        core.set("divg", np.asarray(sd["divg"]).reshape(-1, L, J, I)[0])
