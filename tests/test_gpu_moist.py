"""Moist configuration (BASELINE configs[3]: Frierson grey-radiation aquaplanet) on the GPU, through the C-ABI, against
outputs of the reference itself (tests/golden/moist_*.npz, produced by oracle/ref_moist_harness.F90 from the reference's
own Fortran):
  - idealized_moist_phys on the fixture columns (spun-up T21L25 state, dry / shallow / deep convection, stable and unstable
    surface layers) against the reference's routine-by-routine chain;
  - the moist model from its cold start against the reference trajectory after 1, 2, 10, 144 and 1440 steps.
Tolerances: the column routines are bit-identical on the host (tests/test_moist_cpu.py); on the device the transcendental
functions differ in the last bits, and the convection scheme amplifies that where a column sits next to a regime
boundary, so the column test allows 1e-9 of the field maximum and the trajectory tests what SURVEY 8d states.
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from isca_amd import dyncore            # noqa: E402
from isca_amd import atmosphere as atm  # noqa: E402

FRIERSON_BK = [0.0000000, 0.0117665, 0.0196679, 0.0315244, 0.0485411, 0.0719344, 0.1027829, 0.1418581, 0.1894648, 0.2453219, 0.3085103,
               0.3775033, 0.4502789, 0.5244989, 0.5977253, 0.6676441, 0.7322627, 0.7900587, 0.8400683, 0.8819111, 0.9157609, 0.9422770,
               0.9625127, 0.9778177, 0.9897489, 1.0000000]


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def moist_namelist(res="T21", dt=720.0):
    """The Frierson test case's namelist (isca_amd/configs.py = frierson_test_case.py:49-170) at resolution `res`: the namelist of the
    reference run behind the fixtures (oracle/make_golden.py moist_input_nml)."""
    from isca_amd import configs
    nml = configs.frierson()
    nml["main_nml"]["dt_atmos"] = dt
    nml["spectral_dynamics_nml"].update(dyncore.RESOLUTIONS[res])
    return nml


def moist_core(res="T21", dt=720.0, **kw):
    return dyncore.DynCore(atm.config_from_namelist(moist_namelist(res, dt), **kw))


def test_moist_physics_columns(golden_dir):
    g = np.load(os.path.join(golden_dir, "moist_kernels_T21L25.npz"))
    assert not g["k_cond_dt"].any() and not g["k_cond_dq"].any()     # so the harness chain equals the driver's (cond adds exact zeros)
    dc = moist_core()
    assert np.array_equal(dc.table("bk"), g["tab_bk"])
    delt = float(g["k_in_delta_t"][0])
    du, dv, dt_t, dt_q, ts, precip = dc.idealized_moist_phys(
        delt, 1.0, g["lat_of_col"], g["k_in_u_prev"], g["k_in_v_prev"], g["k_in_t_prev"], g["k_in_q_prev"], g["k_in_p_half_prev"],
        g["k_in_p_full_prev"], g["k_in_p_half_cur"], g["k_in_p_full_cur"], g["k_in_z_half_cur"], g["k_in_z_full_cur"], g["k_in_t_surf"])
    err = {"u": rel(du, g["k_fin_dt_u"]), "v": rel(dv, g["k_fin_dt_v"]), "t": rel(dt_t, g["k_fin_dt_t"]), "q": rel(dt_q, g["k_fin_dt_q"]),
           "t_surf": rel(ts, g["k_ml_t_surf"]), "rain": rel(precip, g["k_conv_rain"] / delt)}
    print("moist physics columns vs reference:", err)
    assert max(err.values()) < 1e-9, err
    dc.close()


def test_moist_work_arrays_in_global_memory(golden_dir):
    """More than 41 levels do not fit the three LDS work arrays: up to 63 the kernel keeps two of them there and the third in a global buffer,
    beyond that all three.  Same results, bit for bit (forced here at 25 levels through the test hook ISCA_MOIST_LDS_ARRAYS = 0 | 2)."""
    g = np.load(os.path.join(golden_dir, "moist_kernels_T21L25.npz"))
    dc = moist_core()
    args = (float(g["k_in_delta_t"][0]), 1.0, g["lat_of_col"], g["k_in_u_prev"], g["k_in_v_prev"], g["k_in_t_prev"], g["k_in_q_prev"],
            g["k_in_p_half_prev"], g["k_in_p_full_prev"], g["k_in_p_half_cur"], g["k_in_p_full_cur"], g["k_in_z_half_cur"], g["k_in_z_full_cur"],
            g["k_in_t_surf"])
    a = dc.idealized_moist_phys(*args)
    for var, val in (("ISCA_MOIST_LDS_ARRAYS", "0"), ("ISCA_MOIST_LDS_ARRAYS", "2")):
        os.environ[var] = val
        try:
            b = dc.idealized_moist_phys(*args)
        finally:
            del os.environ[var]
        for x, y in zip(a, b):
            assert np.array_equal(x, y), (var, val)
    dc.close()


def test_moist_half_level_pressures_on_the_fly(monkeypatch):
    """Above 41 levels the moist kernels do not store the half-level pressures: k_moist_physics forms pk + bk ps where it needs them
    (ISCA_MOIST_PHALF=sigma forces that at 25 levels, =arrays the stored ones).  The same model state bit for bit after 30 steps."""
    def run(mode):
        monkeypatch.setenv("ISCA_MOIST_PHALF", mode)
        dc = moist_core()
        dc.cold_start(); dc.step(30)
        out = {k: dc.get(k) for k in ("ug", "vg", "tg", "psg")}
        out["q"] = dc.get("tr", 1); out["t_surf"] = dc.get("t_surf")
        dc.close()
        return out
    a, b = run("arrays"), run("sigma")
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_moist_trajectory_T85L40(golden_dir):
    """BASELINE configs[3] at its full size: the Frierson model at T85L40 (uneven_sigma levels of the test case's scale_heights / exponent,
    dt = 300 s) from the cold start against the reference run after 1, 12 and 144 steps (12 hours), on the committed
    [3::4, ::8, ::8] sample (ps: [::4, ::4]).  Same error measure as the T21L25 trajectory."""
    g = np.load(os.path.join(golden_dir, "moist_run_T85L40.npz"))
    nml = moist_namelist("T85", float(g["meta_dt_atmos"]))
    nml["spectral_dynamics_nml"].update(num_levels=40, vert_coord_option="uneven_sigma")
    nml.pop("vert_coordinate_nml", None)
    dc = dyncore.DynCore(atm.config_from_namelist(nml))
    assert np.allclose(dc.table("bk"), g["tab_bk"], rtol=0, atol=1e-15)
    dc.cold_start()
    done = 0
    tol = {1: 1e-11, 12: 1e-9, 144: 1e-8}
    for n in (1, 12, 144):
        dc.step(n - done)
        done = n
        err = {}
        for mine, ref in (("ug", "ug"), ("vg", "vg"), ("tg", "tg"), ("tr", "q")):
            r3 = g["st_%s_%06d_s488" % (ref, n)]
            scale = max(float(np.abs(r3).max()), 1.0 if ref in ("ug", "vg") else 0.0)
            err[ref] = float(np.abs(dc.get(mine)[3::4, ::8, ::8] - r3).max()) / scale
        err["psg"] = rel(dc.get("psg")[::4, ::4], g["st_psg_%06d_s44" % n])
        print("moist T85L40 step", n, err)
        assert max(err.values()) < tol[n], (n, err)
    tmin, tmax, umax, qmax = g["final_Tmin_Tmax_maxabsU_qmax"]
    t = dc.get("tg")
    assert abs(t.min() - tmin) < 1e-6 and abs(t.max() - tmax) < 1e-6 and abs(dc.get("tr").max() - qmax) < 1e-8 * qmax + 1e-12
    dc.close()


def test_moist_trajectory_T21L25(golden_dir):
    """From the cold start, pointwise, while the comparison is meaningful (1.2 days): fraction of the field maximum (winds, which
    start from rest: of max(|u|, 1 m/s)).  Measured: step 1 1e-14, step 144 1e-10 (u), 2e-12 (T), 1.3e-9 (q) - the same size as the
    difference between two runs of the reference whose initial humidity differs by 5e-12 (PARITY.md)."""
    g = np.load(os.path.join(golden_dir, "moist_run_T21L25.npz"))
    dc = moist_core(dt=float(g["meta_dt_atmos"]))
    dc.cold_start()
    done = 0
    tol = {1: 1e-11, 2: 1e-10, 10: 1e-9, 144: 1e-8}
    for n in (1, 2, 10, 144):
        dc.step(n - done)
        done = n
        err = {}
        for mine, ref in (("ug", "ug"), ("vg", "vg"), ("tg", "tg"), ("tr", "q"), ("psg", "psg")):
            key = "st_%s_%06d" % (ref, n)
            if key in g.files:
                scale = max(float(np.abs(g[key]).max()), 1.0 if ref in ("ug", "vg") else 0.0)
                err[ref] = float(np.abs(dc.get(mine) - g[key]).max()) / scale
        print("moist T21L25 step", n, err)
        assert max(err.values()) < tol[n], (n, err)
    dc.close()


def developed_core(g, one_ulp=False):
    """A handle holding the developed moist state of tests/golden/moist_developed_T42L25.npz (one_ulp: every temperature one ulp up)."""
    dc = moist_core("T42", float(g["meta_dt_atmos"]))
    assert np.array_equal(dc.table("bk"), g["tab_bk"])
    dc.cold_start()
    dc.set_time_pointers(0, 1, int(g["meta_step0"]))
    for tl, tag in ((0, "prev"), (1, "cur")):                       # time_level 0 = previous, 1 = current
        for nm in ("vors", "divs", "ts"):
            dc.set(nm, g[f"rs_{nm}_{tag}"], tl)
        dc.set("ln_ps", g[f"rs_lnps_{tag}"], tl)
        for nm in ("ug", "vg", "tg", "psg"):
            a = g[f"rs_{nm}_{tag}"]
            dc.set(nm, np.nextafter(a, np.inf) if (one_ulp and nm == "tg") else a, tl)
    dc.set("tr", g["rs_tr1_prev_filt"], 0); dc.set("tr_atm", g["rs_tr1_prev_atm"], 0)
    dc.set("tr", g["rs_tr1_cur"], 1); dc.set("tr_atm", g["rs_tr1_cur"], 1)
    dc.set("wg_full", g["rs_wg_full"])
    dc.set("t_surf", g["rs_t_surf"])
    dc.refresh_derived()
    dc.set_info("phys_calls", int(g["meta_step0"]))
    return dc


def test_moist_developed_state_steps_vs_reference(golden_dir):
    """A DEVELOPED state of the reference's MOIST model handed over and stepped (tests/golden/moist_developed_T42L25.npz: day 30 of the T42L25
    Frierson run of oracle/ref_moist_harness.F90; it rains -- precipitation up to 1.2e-3 kg/m2/s, max |u| 58 m/s, q up to 1.4e-2, deep and shallow
    convection in most tropical columns, condensation with re-evaporation, an evolved mixed layer -- as `developed_T42L25.npz` does for the dry
    core: both time levels of the spectral and grid state, the dynamics' Robert-filtered humidity level and atmosphere_mod's copy, omega, the
    mixed layer's t_surf (module-private in the reference: read by oracle/ref_peek.c), and the gust state of a model that is RUNNING
    (phys_calls > 0: vert_turb_driver's constant_gust, not the first call's 1 m/s).  Then 1 and 10 more steps against the reference's own
    (idealized_moist_phys.F90:819-1395; the knife-edge branches of qe_moist_convection.F90:1053-1180): 1e-11 and 1e-10 of each field's maximum --
    or, where that is larger, twice the model's own response to a ONE-ULP perturbation of the handed-over temperatures, measured here with a second
    handle (a convecting state amplifies rounding: after 10 steps the 1-ulp twin differs by ~1e-10 in q; a comparison cannot be tighter than that)."""
    g = np.load(os.path.join(golden_dir, "moist_developed_T42L25.npz"))
    umax, vmax, tmin, tmax, qmax = g["developed_maxu_maxv_Tmin_Tmax_qmax"]
    assert umax > 25.0 and qmax > 0.01, (umax, qmax)              # the fixture IS a developed, moist flow

    dc, twin = developed_core(g, False), developed_core(g, True)
    done = 0
    for n, tol in ((1, 1e-11), (10, 1e-10)):
        dc.step(n - done); twin.step(n - done); done = n
        err, noise = {}, {}
        for k, gk in (("ug", "ug"), ("vg", "vg"), ("tg", "tg"), ("tr", "q")):
            ref = g[f"after{n}_{gk}_s222"]
            a = dc.get(k)
            err[k] = float(np.abs(a[::2, ::2, ::2] - ref).max() / np.abs(ref).max())
            noise[k] = float(np.abs(twin.get(k) - a).max() / np.abs(ref).max())
            lo, hi = g[f"after{n}_{gk}_minmax"]
            bound = max(tol, 2 * noise[k]) * max(abs(lo), abs(hi))
            assert abs(a.min() - lo) <= bound and abs(a.max() - hi) <= bound, (n, k, noise[k])
        err["psg"] = rel(dc.get("psg"), g[f"after{n}_psg"]); noise["psg"] = rel(twin.get("psg"), dc.get("psg"))
        print("developed moist T42L25 state +", n, "steps vs the reference:", err, "| response to 1 ulp in T:", noise, "| rain max", float(dc.get("precip").max()))
        for k in err:
            assert err[k] < max(tol, 2 * noise[k]), (n, k, err[k], noise[k])
    assert float(dc.get("precip").max()) > 1e-4                    # (kg/m2/s: it rains)
    dc.close(); twin.close()


def test_moist_convection_sigma_log_tables(golden_dir, monkeypatch):
    """On pure sigma levels the convection scheme takes ln(p_full(k) / p_full(k+1)) and ln(p_half(k+1) / p_half(k)) -- constants of the vertical
    coordinate there -- from a table built once in extended precision (moist_physics.h: QeParcel::sig) instead of dividing and taking two logarithms
    per level and column (ISCA_MOIST_LOG_PER_LEVEL=1: the per-level form, bit-identical to the host twin's).  The two forms differ in the last bit of
    a logarithm, which moves a column that sits on one of the scheme's knife-edge branches: on the developed, raining state one step apart by 6e-12
    of the humidity's maximum (measured; 1e-13 in the other fields, the response to one ulp in T is 4e-14) -- inside the 1e-11 the one-step comparison
    with the reference allows (test_moist_developed_state_steps_vs_reference runs the table form: 2e-12 in q)."""
    g = np.load(os.path.join(golden_dir, "moist_developed_T42L25.npz"))
    tab, twin = developed_core(g, False), developed_core(g, True)
    monkeypatch.setenv("ISCA_MOIST_LOG_PER_LEVEL", "1")
    per_level = developed_core(g, False)
    monkeypatch.delenv("ISCA_MOIST_LOG_PER_LEVEL")
    for dc in (tab, twin, per_level):
        dc.step(1)
    diff, noise = {}, {}
    for k in ("ug", "vg", "tg", "tr", "psg"):
        a, b, c = tab.get(k), per_level.get(k), twin.get(k)
        scale = float(np.abs(a).max())
        diff[k] = float(np.abs(a - b).max()) / scale
        noise[k] = float(np.abs(a - c).max()) / scale
    print("convection's log tables vs per-level logarithms, one step on the developed state:", diff, "| response to 1 ulp in T:", noise)
    assert float(tab.get("precip").max()) > 1e-4
    assert any(v > 0 for v in diff.values())                        # (the switch does select another path)
    for k in diff:
        assert diff[k] < 1e-11, (k, diff[k], noise[k])
    for dc in (tab, twin, per_level):
        dc.close()


def test_moist_convection_ahead_equals_in_step(monkeypatch):
    """The next step's convection + condensation run on a side stream under this step's dynamics (they read previous-level fields only, which are
    final a step earlier; moist.hip).  Same arithmetic on the same inputs: the state after 40 steps -- with a host read and a state write in
    between, which drop what was computed ahead -- equals the run that computes them inside each step (ISCA_MOIST_NO_PIPELINE) bit for bit."""
    def run(no_pipeline):
        if no_pipeline:
            monkeypatch.setenv("ISCA_MOIST_NO_PIPELINE", "1")
        else:
            monkeypatch.delenv("ISCA_MOIST_NO_PIPELINE", raising=False)
        dc = moist_core()
        dc.cold_start(); dc.step(1); dc.step(19)
        ts = dc.get("t_surf"); dc.set("t_surf", ts)                 # a state write between steps
        dc.step(20)
        out = {k: dc.get(k) for k in ("ug", "vg", "tg", "psg", "t_surf", "precip")}
        out["q"] = dc.get("tr", 1)
        dc.close()
        return out
    a, b = run(False), run(True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("sigma_half", [False, True])
def test_moist_lazy_fixers_equal_eager(monkeypatch, sigma_half):
    """The fixers' corrections left pending on the new level (lazy fixers) in the moist model too: the dynamics' kernels apply them as they read the
    level, and the physics' pressure kernel -- which visits the current level's T anyway -- leaves T, q (atmosphere_mod's copy) and p_s with the
    pending scalars applied for the column kernels.  Against ISCA_EAGER_FIXERS=1 (the pass over the fields at the end of every step, as until
    round 5): 60 steps from the cold start (it rains from step ~30 in this configuration's tropics), with a host read, a state write and a read of
    the pressures in between, bit for bit; also with the half-level pressures formed on the fly from the corrected p_s."""
    if sigma_half:
        monkeypatch.setenv("ISCA_MOIST_PHALF", "sigma")
    def run(eager, looks):
        (monkeypatch.setenv("ISCA_EAGER_FIXERS", "1") if eager else monkeypatch.delenv("ISCA_EAGER_FIXERS", raising=False))
        dc = moist_core()
        assert bool(dc.info("lazy_fixers")) == (not eager)
        dc.cold_start()
        done = 0
        for stop in looks + [60]:
            dc.step(stop - done); done = stop
            if looks:
                dc.get("tg"); dc.get("p_full")
                if stop == looks[-1]:
                    ts = dc.get("t_surf"); dc.set("t_surf", ts)
        out = {(k, tl): dc.get(k, tl) for k in ("ug", "vg", "tg", "psg", "tr", "tr_atm", "ts", "ln_ps") for tl in (0, 1)}
        out["t_surf"], out["precip"], out["fixer"] = dc.get("t_surf"), dc.get("precip"), dc.table("fixer")[16:19]
        dc.close()
        return out
    lazy, looked, eager = run(False, []), run(False, [1, 2, 17, 40]), run(True, [])
    assert eager["fixer"][0] != 1.0 and eager["fixer"][1] != 0.0 and eager["fixer"][2] != 1.0
    for k in eager:
        assert np.array_equal(lazy[k], eager[k]), k
        assert np.array_equal(looked[k], eager[k]), k


def test_moist_virtual_temperature(golden_dir):
    """use_virtual_temperature = .true. in the moist model (q ~ 1e-2: a 0.6 % change of the pressure-gradient and energy-conversion terms
    and of the heights the physics sees): 40 steps from the cold start against the reference run with the same flag, same error measure
    and tolerances as the T21L25 trajectory; without the flag the run is visibly a different one."""
    g = np.load(os.path.join(golden_dir, "moist_run_T21L25_virtual_t.npz"))
    nml = moist_namelist("T21", float(g["meta_dt_atmos"]))
    nml["spectral_dynamics_nml"]["use_virtual_temperature"] = True
    dc = dyncore.DynCore(atm.config_from_namelist(nml))
    assert dc.cfg.use_virtual_temperature == 1
    dc.cold_start()
    done = 0
    tol = {1: 1e-11, 2: 1e-10, 10: 1e-9, 40: 3e-9}
    for n in (1, 2, 10, 40):
        dc.step(n - done)
        done = n
        err = {}
        for mine, ref in (("ug", "ug"), ("tg", "tg"), ("tr", "q"), ("psg", "psg")):
            key = "st_%s_%06d" % (ref, n)
            scale = max(float(np.abs(g[key]).max()), 1.0 if ref == "ug" else 0.0)
            err[ref] = float(np.abs(dc.get(mine) - g[key]).max()) / scale
        print("moist, virtual temperature, step", n, err)
        assert max(err.values()) < tol[n], (n, err)
    dc.close()
    plain = moist_core(dt=float(g["meta_dt_atmos"]))
    plain.cold_start(); plain.step(40)
    assert float(np.abs(plain.get("ug") - g["st_ug_000040"]).max()) > 1e-4          # the flag matters: m/s after 40 steps
    plain.close()


def test_moist_climate_12day(golden_dir):
    """12 days (1440 steps): the convection scheme makes the model chaotic on this time scale - two runs of the REFERENCE differing
    by 5e-12 in the initial humidity end 0.4 m/s, 0.2 K, 7 Pa apart pointwise (zonal means 0.02 m/s, 0.01 K) - so the check is on
    zonal and global means, with bands a few times that spread."""
    g = np.load(os.path.join(golden_dir, "moist_run_T21L25_12day.npz"))
    dc = moist_core(dt=float(g["meta_dt_atmos"]))
    dc.cold_start()
    dc.step(1440)
    band = {"ug": 0.15, "vg": 0.15, "tg": 0.15, "q": 5e-4, "psg": 3.0}            # zonal means: m/s, K, kg/kg, Pa
    gm_rel = {"tg": 1e-5, "q": 1e-3, "psg": 1e-6}
    for mine, ref in (("ug", "ug"), ("vg", "vg"), ("tg", "tg"), ("tr", "q"), ("psg", "psg")):
        a, b = dc.get(mine), g["st_%s_001440" % ref]
        zm = float(np.abs(a.mean(axis=-1) - b.mean(axis=-1)).max())
        print("moist 12 days", ref, "max diff", float(np.abs(a - b).max()), "zonal-mean diff", zm, "global means", a.mean(), b.mean())
        assert zm < band[ref], (ref, zm)
        if ref in gm_rel:
            assert abs(a.mean() - b.mean()) < gm_rel[ref] * abs(b.mean()), ref
    t = dc.get("tg")
    assert abs(t.min() - g["final_Tmin_Tmax_maxabsU_qmax"][0]) < 1.0 and abs(t.max() - g["final_Tmin_Tmax_maxabsU_qmax"][1]) < 1.0
    dc.close()


def test_moist_surface_state_and_errors():
    dc = moist_core()
    dc.cold_start()
    lat = np.deg2rad(dc.table("deg_lat"))
    ts = dc.get("t_surf")
    assert ts.shape == (dc.Jl, dc.I) and rel(ts, np.repeat((285. - 40. * (3 * np.sin(lat) ** 2 - 1) / 3.)[:, None], dc.I, 1)) < 1e-14
    dc.step(5)
    assert np.abs(dc.get("t_surf") - ts).max() > 1e-4 and dc.get("precip").min() >= 0.0
    assert np.isfinite(dc.get("dt_tg")).all()
    dc.close()
    # more levels than the LDS work arrays hold (global-memory work arrays, the 64-level instance of the convection arrays)
    big = dyncore.DynCore(dyncore.default_config("T21", num_levels=50, physics=1, dt_atmos=600.0, initial_sphum=2e-6, scale_heights=8.0))
    big.cold_start()
    big.step(30)
    assert np.isfinite(big.get("tg")).all() and np.isfinite(big.get("tr")).all() and big.get("tr").max() > 2e-6
    big.close()
    hs = dyncore.DynCore(dyncore.default_config("T21"))
    with pytest.raises(dyncore.IscaError, match="moist physics"):
        hs.get("t_surf")
    hs.close()
    with pytest.raises(dyncore.IscaError, match="not a supported value"):
        atm.config_from_namelist({"atmosphere_nml": {"idealized_moist_model": True}, "two_stream_gray_rad_nml": {"rad_scheme": "byrne"}})
    with pytest.raises(dyncore.IscaError, match="frac_inner"):
        dyncore.DynCore(dyncore.default_config("T21", physics=1, moist={"frac_inner": 1.5}))


def test_moist_restart_round_trip(tmp_path):
    """mixed_layer.res.nc next to the dynamics restart files: a restarted run continues bit for bit like a run that was put into
    the restart state in memory (the reference re-initialises gust to 1 m/s at every start, so does set_time_pointers)."""
    from isca_amd import restart
    a = moist_core()
    a.cold_start()
    a.step(7)
    restart.write_restart(a, str(tmp_path))
    assert os.path.exists(tmp_path / "mixed_layer.res.nc")
    b = moist_core()
    restart.read_restart(b, str(tmp_path))
    assert np.array_equal(a.get("t_surf"), b.get("t_surf"))
    a.set_time_pointers(a.info("previous"), a.info("current"), a.info("step"))
    a.step(5)
    b.step(5)
    for k in ("ug", "vg", "tg", "psg", "tr", "t_surf", "vors", "ts"):
        assert np.array_equal(a.get(k), b.get(k)), k
    a.close(); b.close()


def test_moist_sharded_matches_single():
    """Two latitude bands on this box's GPU (gloo): the column physics is local to a band, the water fixer sums cross them."""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
           "29641", os.path.join(repo, "tests", "mp_sharded_check.py"), "--backend", "gloo", "--steps", "8", "--res", "T21", "--levels", "25",
           "--moist"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=repo)
    assert r.returncode == 0 and "SHARDED_CHECK OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_moist_diagnostics_and_history(tmp_path):
    """precipitation (module atmosphere) and t_surf (module mixed_layer) as device-side time means next to the dynamics fields,
    and in the history file the diag_table asks for."""
    from isca_amd.diag import DiagTable, History
    from scipy.io import netcdf_file
    dc = moist_core()
    dc.cold_start()
    dc.step(600)                                   # 5 days: the boundary layer has moistened and it rains
    dc.diag_select(["precipitation", "t_surf", "temp"])
    acc_p, acc_t = np.zeros((dc.Jl, dc.I)), np.zeros((dc.Jl, dc.I))
    for _ in range(6):
        dc.step(1)
        acc_p += dc.get("precip"); acc_t += dc.get("t_surf")
    mp, n = dc.diag_mean("precipitation")
    mt, _ = dc.diag_mean("t_surf")
    assert n == 6 and rel(mp, acc_p / 6) < 1e-14 and rel(mt, acc_t / 6) < 1e-15 and mp.max() > 0
    dc.diag_select("")
    tab = DiagTable()
    tab.add_file("atmos_6h", 2, "hours")
    tab.add_field("dynamics", "ps", time_avg=True)
    tab.add_field("atmosphere", "precipitation", time_avg=True)
    tab.add_field("mixed_layer", "t_surf", time_avg=True)
    with pytest.raises(dyncore.IscaError, match="belongs to module"):
        tab.add_field("dynamics", "t_surf")
    hist = History(dc, tab.files["atmos_6h"], 720.0, str(tmp_path / "atmos_6h.nc"))
    for _ in range(2):
        dc.step(10)
        hist.after_steps(10)
    hist.close()
    f = netcdf_file(str(tmp_path / "atmos_6h.nc"), "r", mmap=False)
    assert f.variables["precipitation"].shape == (2, dc.J, dc.I) and f.variables["t_surf"][:].min() > 200.0
    f.close()
    dc.close()
    hs = dyncore.DynCore(dyncore.default_config("T21"))
    with pytest.raises(dyncore.IscaError, match="moist physics package"):
        hs.diag_select(["t_surf"])
    hs.close()


def test_moist_experiment_segments(tmp_path):
    """Run segmentation with the moist namelist: the restart archive carries mixed_layer.res.nc, the second segment continues from it,
    the history file holds precipitation and t_surf."""
    import tarfile
    from isca_amd.experiment import Experiment
    from scipy.io import netcdf_file
    e = Experiment("frierson_mini", str(tmp_path))
    nml = moist_namelist()
    nml["main_nml"].update({"days": 0, "hours": 4, "dt_atmos": 720})
    e.update_namelist(nml)
    e.diag_table.add_file("atmos_2h", 2, "hours")
    e.diag_table.add_field("dynamics", "temp", time_avg=True)
    e.diag_table.add_field("atmosphere", "precipitation", time_avg=True)
    e.diag_table.add_field("mixed_layer", "t_surf", time_avg=True)
    assert e.run(1) and e.run(2)
    with tarfile.open(e.get_restart_file(2)) as tar:
        names = {os.path.basename(n) for n in tar.getnames()}
    assert {"spectral_dynamics.res.nc", "atmosphere.res.nc", "mixed_layer.res.nc"} <= names
    f = netcdf_file(os.path.join(e.get_outputdir(2), "atmos_2h.nc"), "r", mmap=False)
    assert f.variables["t_surf"].shape == (2, 32, 64) and f.variables["precipitation"].shape == (2, 32, 64)
    assert np.allclose(f.variables["average_T1"][:], [4.0, 6.0])
    f.close()
