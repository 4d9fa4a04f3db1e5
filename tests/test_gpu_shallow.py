"""Shallow-water sibling core on the GPU, through include/isca_shallow.h, against the reference's own
src/atmos_spectral_shallow (fixtures tests/golden/shallow_run_*.npz written by oracle/ref_shallow_harness.F90): a vortex pair on a
zonal flow over the default mass forcing, both tracers on.  fp64; bounds are fractions of each field's maximum."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from isca_amd import shallow             # noqa: E402
from isca_amd.dyncore import IscaError   # noqa: E402

NML = {"shallow_dynamics_nml": {"add_initial_vortex_pair": True, "u_upper_mag_init": 10.0, "u_deep_mag": 5.0}, "main_nml": {"dt_atmos": 1200}}
PAIRS = (("u", "u"), ("v", "v"), ("vor", "vor"), ("div", "div"), ("h", "h"), ("tr", "tr"), ("trs", "trs"), ("vors", "vors"), ("hs", "hs"),
         ("stream", "stream"), ("pv", "pv"))


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def test_shallow_trajectory_T21(golden_dir):
    g = np.load(os.path.join(golden_dir, "shallow_run_T21.npz"))
    sw = shallow.ShallowWater(shallow.config_from_namelist(NML, "T21"))
    assert rel(sw.get("deep_geopot"), g["tab_deep_geopot"]) < 1e-13
    sw.cold_start()
    for mine, ref in PAIRS[:9]:
        assert rel(sw.get(mine), g["st_%s_000000" % ref]) < 1e-13, mine       # initial condition
    done = 0
    tol = {1: 1e-12, 2: 1e-12, 10: 1e-11, 200: 1e-9}
    for n in (1, 2, 10, 200):
        sw.step(n - done)
        done = n
        err = {ref: rel(sw.get(mine), g["st_%s_%06d" % (ref, n)]) for mine, ref in PAIRS}
        err["div"] = float(np.abs(sw.get("div") - g["st_div_%06d" % n]).max() / np.abs(g["st_vor_%06d" % n]).max())   # div << vor: scale by vor
        print("shallow T21 step", n, {k: "%.1e" % v for k, v in err.items()})
        assert max(err.values()) < tol[n], (n, err)
    assert sw.info("step") == 200
    sw.close()


def test_shallow_T42_300_steps(golden_dir):
    g = np.load(os.path.join(golden_dir, "shallow_run_T42.npz"))
    sw = shallow.ShallowWater(shallow.config_from_namelist(NML, "T42"))
    sw.cold_start()
    sw.step(300)
    err = {k: rel(sw.get(k), g["st_%s_000300" % k]) for k in ("u", "v", "h", "vor", "tr", "trs")}
    print("shallow T42 step 300", err)
    assert max(err.values()) < 1e-9, err
    h = sw.get("h")
    assert abs(h.min() - g["final_hmin_hmax_maxabsU"][0]) < 1e-6 and abs(h.max() - g["final_hmin_hmax_maxabsU"][1]) < 1e-6
    sw.close()


def test_shallow_restart_and_module_mirror():
    """set_state + set_time_pointers continue a run bit for bit; the module-level mirror drives the same handle type."""
    a = shallow.ShallowWater(shallow.config_from_namelist(NML, "T21"))
    a.cold_start()
    a.step(7)
    b = shallow.ShallowWater(shallow.config_from_namelist(NML, "T21"))
    p, c = a.info("previous"), a.info("current")
    for tl, slot in ((0, p), (1, c)):
        b.set_time_pointers(slot, slot, 0)          # address storage slot `slot` as "current" to fill it
        for k in ("u", "v", "vor", "div", "h", "tr", "trs", "vors", "divs", "hs", "trss"):
            b.set(k, a.get(k, tl), 1)
    b.set_time_pointers(p, c, 7)
    a.step(5)
    b.step(5)
    for k in ("u", "v", "h", "vor", "tr", "trs", "vors", "hs"):
        assert np.array_equal(a.get(k), b.get(k)), k
    a.close(); b.close()
    m = shallow.atmosphere_init(NML, "T21")
    shallow.atmosphere(3)
    assert m.info("step") == 3 and np.isfinite(m.get("h")).all()
    shallow.atmosphere_end()
    with pytest.raises(IscaError, match="atmosphere_init has not been called"):
        shallow.atmosphere()


def test_shallow_errors():
    with pytest.raises(IscaError, match="not a supported value for fourier_inc"):
        shallow.config_from_namelist({"shallow_dynamics_nml": {"fourier_inc": 2}})
    sw = shallow.ShallowWater(shallow.config_from_namelist(NML, "T21"))
    with pytest.raises(IscaError, match="has not been initialized"):
        sw.step(1)
    sw.cold_start()
    with pytest.raises(IscaError, match="unknown field"):
        sw.get("nonsense")
    bad = shallow.ShallowWater(shallow.config_from_namelist(NML, "T21", valid_range_v=(-1e-3, 1e-3)))
    bad.cold_start()
    with pytest.raises(IscaError, match="meridional wind out of valid range"):
        bad.step(20)
    sw.close(); bad.close()


def test_shallow_stirring_vs_reference(golden_dir):
    """shallow_water_stirring_test: stirring added to the vorticity tendency of the shallow-water core, with the reference's own draws."""
    g = np.load(os.path.join(golden_dir, "shallow_stirring_T21.npz"))
    nml = {**NML, "stirring_nml": {"decay_time": 172800, "amplitude": 3.e-12, "lat0": 45., "lon0": 180., "widthy": 12., "widthx": 45., "B": 1.0}}
    sw = shallow.ShallowWater(shallow.config_from_namelist(nml, "T21"))
    sw.cold_start()
    for n in range(1, 41):
        sw.set_stirring_noise(g["in_stir_ran"][n - 1])
        sw.step(1)
        if n in (1, 40):
            err = {k: rel(sw.get(k), g["st_%s_%06d" % (k, n)]) for k in ("u", "v", "vor", "h", "vors")}
            print("shallow stirring step", n, {k: "%.1e" % v for k, v in err.items()})
            assert max(err.values()) < 1e-11, (n, err)
    sw.close()


def test_shallow_init_from_grid():
    """initial_condition_from_input_file (prescribed_ics_test): handing the cold start's own vorticity, divergence and height anomaly
    back through init_from_grid reproduces the cold start (same transforms), and a perturbed height changes the run."""
    a = shallow.ShallowWater(shallow.config_from_namelist(NML, "T21"))
    a.cold_start()
    b = shallow.ShallowWater(shallow.config_from_namelist({**NML, "shallow_dynamics_nml": {**NML["shallow_dynamics_nml"],
                                                                                             "initial_condition_from_input_file": True}}, "T21"))
    vor, div, h = a.get("vor"), a.get("div"), a.get("h")
    b.init_from_grid(vor, div, h - a.cfg.h_0)
    for k in ("u", "v", "vor", "div", "vors", "tr", "trs"):
        assert np.array_equal(a.get(k), b.get(k)), k
    assert rel(b.get("h"), h) < 1e-15 and rel(b.get("hs"), a.get("hs")) < 1e-15          # (h - h_0) + h_0 is not bit-exact
    a.step(20); b.step(20)
    assert rel(b.get("h"), a.get("h")) < 1e-12
    c = shallow.ShallowWater(shallow.config_from_namelist(NML, "T21"))
    c.init_from_grid(vor, div, h - a.cfg.h_0 + 50.0 * np.cos(np.deg2rad(np.arange(32) * 5.0))[:, None])
    c.step(20)
    assert rel(c.get("h"), a.get("h")) > 1e-6
    for m in (a, b, c):
        m.close()


def test_shallow_giant_planet(golden_dir):
    """constants_nml (radius 55000 km, omega 1.6e-4: the planet of shallow_water_test.py) reaches the tables, the Coriolis parameter, the
    van Leer metric terms and the initial condition."""
    g = np.load(os.path.join(golden_dir, "shallow_run_giant_T21.npz"))
    nml = {**NML, "constants_nml": {"radius": 55000.e3, "omega": 1.6e-4}}
    sw = shallow.ShallowWater(shallow.config_from_namelist(nml, "T21"))
    assert rel(sw.get("deep_geopot"), g["tab_deep_geopot"]) < 1e-13
    sw.cold_start()
    sw.step(1)
    assert max(rel(sw.get(k), g["st_%s_000001" % k]) for k in ("u", "vor", "h", "tr", "trs", "hs")) < 1e-12
    sw.step(99)
    err = {k: rel(sw.get(k), g["st_%s_000100" % k]) for k in ("u", "v", "vor", "h", "tr", "trs", "vors", "hs")}
    print("shallow water on the giant planet, 100 steps:", err)
    assert max(err.values()) < 1e-10, err
    sw.close()


def test_tracers_are_passive():
    """spec_tracer / grid_tracer off: the batches shrink, the dynamics do not change (bit for bit), for both cores."""
    on = shallow.ShallowWater(shallow.config_from_namelist(NML, "T21"))
    off = shallow.ShallowWater(shallow.config_from_namelist({**NML, "shallow_dynamics_nml": {**NML["shallow_dynamics_nml"], "spec_tracer": False,
                                                                                                  "grid_tracer": False}}, "T21"))
    for m in (on, off):
        m.cold_start(); m.step(25)
    for k in ("u", "v", "h", "vor", "div", "vors", "hs"):
        assert np.array_equal(on.get(k), off.get(k)), k
    on.close(); off.close()
    bon = shallow.Barotropic(shallow.barotropic_config_from_namelist({"main_nml": {"dt_atmos": 1200}}, "T21"))
    boff = shallow.Barotropic(shallow.barotropic_config_from_namelist({"main_nml": {"dt_atmos": 1200},
                                                                       "barotropic_dynamics_nml": {"spec_tracer": False, "grid_tracer": False}}, "T21"))
    for m in (bon, boff):
        m.cold_start(); m.step(25)
    for k in ("u", "v", "vor", "vors"):
        assert np.array_equal(bon.get(k), boff.get(k)), k
    bon.close(); boff.close()


def test_sibling_restart_files(tmp_path):
    """<core>_dynamics.res.nc with the reference's variable set (two records: previous, current) and stirring.res.nc: a run continued
    from the files equals the uninterrupted run bit for bit (with supplied stirring noise, so both draw the same numbers)."""
    from scipy.io import netcdf_file
    rng = np.random.default_rng(3)
    for kind in ("shallow", "barotropic"):
        nml = {"main_nml": {"dt_atmos": 1200}, "stirring_nml": {"amplitude": 3.e-12, "B": 1.0}}
        if kind == "shallow":
            nml["shallow_dynamics_nml"] = NML["shallow_dynamics_nml"]
            mk = lambda: shallow.ShallowWater(shallow.config_from_namelist(nml, "T21"))
        else:
            mk = lambda: shallow.Barotropic(shallow.barotropic_config_from_namelist(nml, "T21"))
        a = mk(); a.cold_start()
        noise = rng.random((12, 2, a.N1, a.M1))
        for n in range(6):
            a.set_stirring_noise(noise[n]); a.step(1)
        d = str(tmp_path / kind)
        shallow.write_restart(a, d)
        f = netcdf_file(os.path.join(d, kind + "_dynamics.res.nc"), "r", mmap=False)
        assert f.variables["vors_real"].shape[0] == 2 and f.variables["u"].shape[-2:] == (32, 64) and "tr" in f.variables and "trs_imag" in f.variables
        f.close()
        assert os.path.exists(os.path.join(d, "stirring.res.nc"))
        b = mk(); shallow.read_restart(b, d)
        for n in range(6, 12):
            for m in (a, b):
                m.set_stirring_noise(noise[n]); m.step(1)
        for k in ("u", "v", "vor", "tr", "trs", "vors", "stirs") + (("h", "hs") if kind == "shallow" else ()):
            assert np.array_equal(a.get(k), b.get(k)), (kind, k)
        a.close(); b.close()
    with pytest.raises(IscaError, match="restart does not exist"):
        shallow.read_restart(mk(), str(tmp_path / "nowhere"))
