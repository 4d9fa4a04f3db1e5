#!/usr/bin/env python3
"""Exploratory consistency checks of option combinations on the GPU (not a test: prints, raises on the first inconsistency)."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isca_amd import dyncore, restart, atmosphere as atm, configs


def mk(**kw):
    base = dict(num_levels=8)
    base.update(kw)
    return dyncore.DynCore(dyncore.default_config("T21", **base))


def topo(dc):
    lat = np.deg2rad(dc.table("deg_lat"))[:, None]; lon = np.deg2rad(dc.table("deg_lon"))[None, :]
    dc.set_surf_geopotential(9.8 * 2000.0 * np.exp(-((lat - 0.6) / 0.3) ** 2 - ((lon - 2.0) / 0.5) ** 2))


def restart_roundtrip(make_core, nsteps=6, more=5, names_extra=()):
    a = make_core(); a.cold_start(); a.step(nsteps)
    d = tempfile.mkdtemp()
    restart.write_restart(a, d)
    a.step(more)
    b = make_core(flat=True)
    restart.read_restart(b, d)
    b.step(more)
    for k in ("ug", "tg", "psg", "tr", "vors") + tuple(names_extra):
        if not np.array_equal(a.get(k), b.get(k)):
            raise SystemExit(f"restart round trip differs in {k}: {np.abs(a.get(k) - b.get(k)).max()}")
    a.close(); b.close()


# 1. topography + three tracers + restart
def c1(flat=False):
    dc = mk(num_tracers=3, tracer_spectral=[0, 0, 1], tracer_robert_coeff=[-1.0, 0.05, -1.0])
    if not flat:
        topo(dc)
    return dc
restart_roundtrip(c1, names_extra=("tr2", "tr3", "trs3"))
print("1 ok: topography + three tracers restart bit-exact")

# 2. virtual temperature + topography + restart
def c2(flat=False):
    dc = mk(use_virtual_temperature=1)
    if not flat:
        topo(dc)
    return dc
restart_roundtrip(c2)
print("2 ok: virtual temperature + topography restart bit-exact")

# 3. physics = 2 with three tracers: dt_tracers for every tracer == fused hs forcing
ref = mk(num_tracers=3, tracer_spectral=[0, 0, 1]); ref.cold_start()
ext = mk(num_tracers=3, tracer_spectral=[0, 0, 1], physics=2); ext.cold_start()
for _ in range(12):
    dt = ext.delta_t()
    u, v, t = ext.get("ug", 0), ext.get("vg", 0), ext.get("tg", 0)
    ph, pf = ext.get("p_half", 1), ext.get("p_full", 1)
    du, dv, dT = ext.hs_forcing(dt, ph, pf, u, v, t)
    dq = np.stack([ext.hs_tracer_source_sink(ph[-1], ext.get(n, 0)) for n in ("tr_atm", "tr_atm2", "tr_atm3")])
    ext.dynamics(du, dv, dT, dq)
ref.step(12)
for k in ("ug", "tg", "tr", "tr2", "tr3"):
    e = np.abs(ext.get(k) - ref.get(k)).max() / max(np.abs(ref.get(k)).max(), 1e-300 if k != "ug" else 1.0)
    print("  physics=2, three tracers", k, e)
    assert e < 1e-10, k
ref.close(); ext.close()
print("3 ok")

# 4. moist + topography + virtual temperature: runs, restart bit-exact
def c4(flat=False):
    nml = configs.frierson(); nml["spectral_dynamics_nml"].update(dyncore.RESOLUTIONS["T21"]); nml["spectral_dynamics_nml"]["use_virtual_temperature"] = True
    dc = dyncore.DynCore(atm.config_from_namelist(nml))
    if not flat:
        topo(dc)
    return dc
a = c4(); a.cold_start(); a.step(20)
d = tempfile.mkdtemp(); restart.write_restart(a, d)
a.set_time_pointers(a.info("previous"), a.info("current"), a.info("step"))       # gust reset like a restarted run
a.step(5)
b = c4(flat=True); restart.read_restart(b, d); b.step(5)
for k in ("ug", "tg", "psg", "tr", "t_surf"):
    assert np.array_equal(a.get(k), b.get(k)), k
print("4 ok: moist + topography + virtual temperature, restart bit-exact; T range", a.get("tg").min(), a.get("tg").max())
a.close(); b.close()

# 5. levels that do not fill the wavefront chunks evenly: L = 13, 31, 47 run and restart
for L in (13, 31, 47):
    def c5(flat=False, L=L):
        return mk(num_levels=L)
    restart_roundtrip(c5, nsteps=4, more=3)
print("5 ok: L = 13, 31, 47")
print("ALL OK")
