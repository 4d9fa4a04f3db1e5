"""The module-level drop-in: bindings/fortran/dropin holds THIS repository's Fortran modules under the reference's names and public
argument lists -- spectral_dynamics_mod (spectral_dynamics.F90:95-98, :230, :780-795), transforms_mod (transforms.F90:134-184),
press_and_geopot_mod, hs_forcing_mod, implicit_mod, spectral_damping_mod, leapfrog_mod, vert_advection_mod, fv_advection_mod,
global_integral_mod, tracer_type_mod -- forwarding to the C-ABI of isca_amd/lib/libisca_dyn.so.  oracle/build_ref.py dropin compiles them
with the reference's own infrastructure modules (fms_mod, time_manager_mod, tracer_manager_mod, ... in place) and links them with
oracle/ref_harness.F90 UNCHANGED -- the driver that otherwise runs the reference itself and only `use`s those public names.  The binary
(oracle/_ref/ref_harness_gpu.x, built where the reference tree exists and carried to the GPU box like the other prebuilt files) must
reproduce the reference's own outputs (tests/golden) through the GPU."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(REPO, "oracle", "_ref", "ref_harness_gpu.x")


def _need_exe():
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/ref_harness_gpu.x was not built (needs the reference tree: python oracle/build_ref.py dropin)")


def test_reference_harness_runs_on_the_gpu_library(tmp_path, golden_dir):
    """configs[0]: T21L25 Held-Suarez, 144 steps, driven by the reference's call sequence (hs_forcing -> spectral_dynamics ->
    compute_pressures_and_heights, atmosphere.F90:286-349) from Fortran: the dumps equal the reference run's."""
    _need_exe()
    from oracle import make_golden as mg            # checker side: run directory of the reference's test case, dump reader
    d = str(tmp_path / "run")
    mg.prepare_rundir(d, "T21", 25, "run", nsteps=144, dt=600, dump_steps=(1, 2, 144))
    stdout = mg.run_harness(d, exe=EXE, timeout=900)
    out = mg.read_outputs(d, "T21", 25)
    g = np.load(os.path.join(golden_dir, "run_T21L25.npz"))
    # the tables the harness fetched through the module getters: bit for bit the reference's
    for k in ("tab_pk", "tab_bk", "tab_sin_lat", "tab_wts_lat", "tab_deg_lat", "tab_deg_lon", "tab_sin_hem", "tab_wts_hem", "tab_eigen_laplacian"):
        assert np.array_equal(out[k], g[k]), k
    assert np.max(np.abs(out["tab_rad_latb"] - g["tab_rad_latb"])) < 1e-15
    checked = 0
    for k in g.files:
        if not k.startswith("st_"):
            continue
        ref, mine = g[k], out[k]
        tol = 1e-9 * max(np.abs(ref).max(), 1.0 if "_ug_" in k or "_vg_" in k else 1e-300)
        assert np.max(np.abs(mine - ref)) < tol, (k, float(np.max(np.abs(mine - ref))))
        checked += 1
    assert checked >= 9
    tmin, tmax, umax = [float(x) for x in re.search(r"REF_STATE Tmin,Tmax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)", stdout).groups()]
    assert np.max(np.abs(np.array([tmin, tmax, umax]) - g["final_Tmin_Tmax_maxabsU"])) < 1e-9


def test_reference_harness_kernels_on_the_gpu_library(tmp_path, golden_dir):
    """Every public routine the harness exercises one by one (mode 'kernels': transforms and their stages, spectral operators, global means,
    pressure variables and geopotential, hs_forcing, vert_advection second-centred and PPM, a_grid_horiz_advection, implicit_correction,
    spectral damping, leapfrog_2level_A/B) through the drop-in modules, against the reference's outputs for the same inputs."""
    _need_exe()
    from oracle import make_golden as mg
    g = np.load(os.path.join(golden_dir, "kernels_T21L6.npz"))
    d = str(tmp_path / "kernels")
    mg.prepare_rundir(d, "T21", 6, "kernels", dt=600)
    for k in g.files:
        if k.startswith("in_"):
            np.ascontiguousarray(g[k]).tofile(os.path.join(d, k + ".bin"))
    mg.run_harness(d, exe=EXE, timeout=900)
    out = mg.read_outputs(d, "T21", 6)
    M1 = 22
    loose = {"out_hadv_fv_bigcfl": 1e-10, "out_vadv_ppm": 1e-11, "out_hadv_fv": 1e-11}
    checked = 0
    for k in g.files:
        if not k.startswith("out_"):
            continue
        ref, mine = g[k], out[k]
        if k == "out_g2f_a":                    # the device keeps the wavenumbers the model truncates to
            ref, mine = ref[..., :M1], mine[..., :M1]
        err = float(np.max(np.abs(mine - ref)) / max(np.max(np.abs(ref)), 1e-300))
        assert err < loose.get(k, 1e-12), (k, err)
        checked += 1
    assert checked >= 40
    assert np.array_equal(out["tab_legendre"], g["tab_legendre"])


def test_reference_harness_geopotential_over_topography(tmp_path, golden_dir):
    """press_and_geopot_mod: compute_geopotential with the caller's surface geopotential (press_and_geopot.F90:331) through the drop-in module: the
    harness's 'kernels' mode over the two Gaussian mountains of gaussian_topog_nml, against the reference's outputs for the same inputs."""
    _need_exe()
    from oracle import make_golden as mg
    g, t = np.load(os.path.join(golden_dir, "kernels_T21L6.npz")), np.load(os.path.join(golden_dir, "kernels_T21L6_topography.npz"))
    d = str(tmp_path / "kernels")
    mg.prepare_rundir(d, "T21", 6, "kernels", dt=600, extra_groups=mg.GAUSSIAN_TOPOG_GROUPS)
    for k in g.files:
        if k.startswith("in_"):
            np.ascontiguousarray(g[k]).tofile(os.path.join(d, k + ".bin"))
    mg.run_harness(d, exe=EXE, timeout=900)
    out = mg.read_outputs(d, "T21", 6)
    for k in ("out_geopot_full", "out_geopot_half"):
        err = float(np.max(np.abs(out[k] - t[k])) / np.max(np.abs(t[k])))
        assert err < 1e-13, (k, err)
    assert np.abs(t["out_geopot_half"][-1]).max() > 2.0e4 and np.abs(out["out_geopot_full"] - g["out_geopot_full"]).max() > 1.0e4


def test_atmos_model_loop_on_atmosphere_mod(tmp_path, golden_dir):
    """atmos_model's time loop (atmos_model.F90:115-142) on this repository's atmosphere_mod (atmosphere.F90:78): atmosphere_init reads the
    reference's input.nml / field_table, every atmosphere(Time) is one device step; 144 steps land on the reference run."""
    exe = os.path.join(REPO, "oracle", "_ref", "drive_atmos_model_gpu.x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/drive_atmos_model_gpu.x was not built (python oracle/build_ref.py dropin_atmos)")
    from oracle import make_golden as mg
    d = str(tmp_path / "run")
    mg.prepare_rundir(d, "T21", 25, "run", nsteps=144, dt=600)
    open(os.path.join(d, "drive.nml"), "w").write(" &drive_nml\n   nsteps = 144, dt_atmos = 600\n /\n")
    stdout = mg.run_harness(d, exe=exe, timeout=900)
    g = np.load(os.path.join(golden_dir, "run_T21L25.npz"))
    vals = [float(x) for x in re.search(r"DRIVE_STATE Tmin,Tmax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)", stdout).groups()]
    assert np.max(np.abs(np.array(vals) - g["final_Tmin_Tmax_maxabsU"])) < 1e-9
    from oracle.isca_oracle import Config, SpectralCore
    sc = SpectralCore(Config.resolution("T21", 25))
    (mean_ps,) = [float(x) for x in re.search(r"DRIVE_MEAN_PS\s*(\S+)", stdout).groups()]
    assert abs(mean_ps - sc.area_weighted_global_mean(g["st_psg_000144"])) < 1e-6


def test_atmos_model_loop_frierson_through_atmosphere_mod(tmp_path, golden_dir):
    """BASELINE configs[3]'s model behind the reference's module interface: atmosphere_nml idealized_moist_model = .true. makes this
    repository's atmosphere_mod call idealized_moist_phys_init (atmosphere.F90:246-262), which reads the Frierson test case's namelists
    IN FORTRAN -- idealized_moist_phys_nml, two_stream_gray_rad_nml, mixed_layer_nml, qe_moist_convection_nml, lscale_cond_nml,
    sat_vapor_pres_nml, damping_driver_nml, vert_turb_driver_nml, diffusivity_nml, surface_flux_nml, and vert_coordinate_nml's 25 levels --
    and creates the core with physics = 1; atmos_model's loop then queues one device step per atmosphere(Time).  144 steps against the
    reference run of the same input.nml (moist tolerances of tests/test_gpu_moist.py)."""
    exe = os.path.join(REPO, "oracle", "_ref", "drive_atmos_model_gpu.x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/drive_atmos_model_gpu.x was not built (python oracle/build_ref.py dropin_atmos)")
    from oracle import make_golden as mg
    d = str(tmp_path / "run")
    mg.prepare_moist_rundir(d, "T21", 144, dt=720)
    open(os.path.join(d, "drive.nml"), "w").write(" &drive_nml\n   nsteps = 144, dt_atmos = 720\n /\n")
    stdout = mg.run_harness(d, exe=exe, timeout=900)
    g = np.load(os.path.join(golden_dir, "moist_run_T21L25.npz"))
    tg, ug, q = g["st_tg_000144"], g["st_ug_000144"], g["st_q_000144"]
    tmin, tmax, umax = [float(x) for x in re.search(r"DRIVE_STATE Tmin,Tmax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)", stdout).groups()]
    assert abs(tmin - tg.min()) < 1e-7 and abs(tmax - tg.max()) < 1e-7 and abs(umax - np.abs(ug).max()) < 1e-7
    qmax, qpt = [float(x) for x in re.search(r"DRIVE_TRACER qmax,q\(10,16,nlev\)=\s*(\S+)\s+(\S+)", stdout).groups()]
    assert abs(qmax - q.max()) < 1e-10 and abs(qpt - q[24, 15, 9]) < 1e-10
    # an option the device package does not implement is refused by name, from Fortran, like the reference's own "not a valid value" FATALs
    nml = open(os.path.join(d, "input.nml")).read().replace("convection_scheme = 'SIMPLE_BETTS_MILLER'", "convection_scheme = 'FULL_BETTS_MILLER'")
    open(os.path.join(d, "input.nml"), "w").write(nml)
    with pytest.raises(RuntimeError, match="is not a supported value for convection_scheme"):
        mg.run_harness(d, exe=exe, timeout=300)


def test_atmosphere_mod_queues_steps(tmp_path):
    """atmosphere(Time) queues its step (isca_dyn_step(core, 1, sync = 0)): the main program's loop runs ahead of the device, and the
    drop-in's step costs what the library's own step(n) costs -- DRIVE_TIMING of a T42L25 run against DynCore.step on the same box."""
    exe = os.path.join(REPO, "oracle", "_ref", "drive_atmos_model_gpu.x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/drive_atmos_model_gpu.x was not built (python oracle/build_ref.py dropin_atmos)")
    import time
    from oracle import make_golden as mg
    from isca_amd import dyncore
    d = str(tmp_path / "run")
    mg.prepare_rundir(d, "T42", 25, "run", nsteps=1, dt=600)
    open(os.path.join(d, "drive.nml"), "w").write(" &drive_nml\n   nsteps = 4000, dt_atmos = 600\n /\n")
    stdout = mg.run_harness(d, exe=exe, timeout=900)
    ms_dropin = float(re.search(r"DRIVE_TIMING steps=\s*\d+\s+seconds=\s*\S+\s+ms_per_step=\s*(\S+)", stdout).group(1))
    dc = dyncore.DynCore(dyncore.default_config("T42", num_levels=25)); dc.cold_start(); dc.step(1000)
    t0 = time.perf_counter(); dc.step(3000); ms_lib = 1e3 * (time.perf_counter() - t0) / 3000
    dc.close()
    print("T42L25: atmosphere(Time) %.4f ms per step, DynCore.step(n) %.4f ms per step" % (ms_dropin, ms_lib))
    assert ms_dropin < 1.25 * ms_lib + 0.01, (ms_dropin, ms_lib)


ISIDORO = ("local_heating_option = 'Isidoro', local_heating_srfamp = 5.0, local_heating_xwidth = 25., local_heating_ywidth = 12., "
           "local_heating_xcenter = 120., local_heating_ycenter = 20., local_heating_vert_decay = 3.e4")


@pytest.mark.parametrize("fixture,levels,nsteps,extra,groups", [
    ("run_T21L8_topography", 8, 36, "", "topo"),
    ("run_T21L12_hybrid_option", 12, 24,
     "vert_coord_option = 'hybrid', p_press = 0.15, p_sigma = 0.45, scale_heights = 5.0, exponent = 3.0, surf_res = 0.3", ""),
    ("run_T21L8_vadv_finite_volume", 8, 36, "vert_advect_uv = 'van_leer_linear', vert_advect_t = 'finite_volume_parabolic'", ""),
    ("run_T21L8_symmetric", 8, 48, "make_symmetric = .true.", ""),
    ("run_T21L14_mcm_coord", 14, 36, "vert_difference_option = 'mcm', vert_coord_option = 'mcm'", ""),
    ("run_T21L8_no_forcing", 8, 48, "", "topo+no_forcing"),
    ("run_T21L8_isidoro", 8, 48, "", "isidoro"),
])
def test_atmosphere_mod_options_from_fortran(tmp_path, golden_dir, fixture, levels, nsteps, extra, groups):
    """Options the drop-in front end forwards instead of refusing, each from the reference's own input.nml through atmos_model's loop on
    this repository's atmosphere_mod, against the reference run's final extremes: topography_option = 'gaussian' (gaussian_topog_nml through
    the reference's gaussian_topog_mod, spectral_init_cond.F90:299-303), vert_coord_option = 'hybrid' (compute_vert_coord,
    vert_coordinate.F90:124-152, formed in Fortran), vert_advect_uv / vert_advect_t, make_symmetric, vert_difference_option = 'mcm' on the 'mcm' levels, hs_forcing_nml's no_forcing
    (hs_forcing.F90:174) over the Gaussian mountains, and its local_heating_option = 'Isidoro' (hs_forcing.F90:728-769)."""
    exe = os.path.join(REPO, "oracle", "_ref", "drive_atmos_model_gpu.x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/drive_atmos_model_gpu.x was not built (python oracle/build_ref.py dropin_atmos)")
    from oracle import make_golden as mg
    d = str(tmp_path / "run")
    mg.prepare_rundir(d, "T21", levels, "run", nsteps=nsteps, dt=600, extra=extra, extra_groups=mg.GAUSSIAN_TOPOG_GROUPS if groups.startswith("topo") else "",
                      hs_extra="no_forcing = .true." if groups.endswith("no_forcing") else ISIDORO if groups == "isidoro" else "")
    open(os.path.join(d, "drive.nml"), "w").write(f" &drive_nml\n   nsteps = {nsteps}, dt_atmos = 600\n /\n")
    stdout = mg.run_harness(d, exe=exe, timeout=900)
    g = np.load(os.path.join(golden_dir, fixture + ".npz"))
    vals = [float(x) for x in re.search(r"DRIVE_STATE Tmin,Tmax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)", stdout).groups()]
    assert np.max(np.abs(np.array(vals) - g["final_Tmin_Tmax_maxabsU"])) < 1e-9, (vals, g["final_Tmin_Tmax_maxabsU"])


def test_atmosphere_mod_tracer_sms_from_fortran(tmp_path, golden_dir):
    """tracer_sms of the field_table (hs_forcing.F90:251-261) read on the Fortran side (query_method + parse in spectral_dynamics_init of the drop-in)
    and handed to the library: six tracers, sphum with "flux=2.5e-5, sink=-2.0"; 40 steps of atmos_model's loop against the reference run."""
    exe = os.path.join(REPO, "oracle", "_ref", "drive_atmos_model_gpu.x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/drive_atmos_model_gpu.x was not built (python oracle/build_ref.py dropin_atmos)")
    from oracle import make_golden as mg
    d = str(tmp_path / "run")
    mg.prepare_rundir(d, "T21", 8, "run", nsteps=40, dt=600, field_table=mg.FIELD_TABLE_SMS)
    open(os.path.join(d, "drive.nml"), "w").write(" &drive_nml\n   nsteps = 40, dt_atmos = 600\n /\n")
    stdout = mg.run_harness(d, exe=exe, timeout=900)
    g = np.load(os.path.join(golden_dir, "run_T21L8_tracer_sms.npz"))
    q = g["st_tr1_000040"]                                 # (lev, lat, lon)
    qmax, qpt = [float(x) for x in re.search(r"DRIVE_TRACER qmax,q\(10,16,nlev\)=\s*(\S+)\s+(\S+)", stdout).groups()]
    assert abs(qmax - q.max()) < 1e-9 * q.max() and abs(qpt - q[-1, 15, 9]) < 1e-9 * q.max(), (qmax, q.max(), qpt, q[-1, 15, 9])
    plain = np.load(os.path.join(golden_dir, "run_T21L8_six_tracers.npz"))["st_tr1_000040"]
    assert abs(q.max() - plain.max()) > 0.1 * plain.max()          # the entry's own flux and sink matter
    # advect_vert of an entry other than the fused scheme (here sphum with van_leer_linear), also read on the Fortran side
    d = str(tmp_path / "vert")
    mg.prepare_rundir(d, "T21", 8, "run", nsteps=40, dt=600, field_table=mg.FIELD_TABLE.replace('"finite_volume_parabolic"', '"van_leer_linear"'))
    open(os.path.join(d, "drive.nml"), "w").write(" &drive_nml\n   nsteps = 40, dt_atmos = 600\n /\n")
    stdout = mg.run_harness(d, exe=exe, timeout=900)
    q = np.load(os.path.join(golden_dir, "run_T21L8_sphum_van_leer.npz"))["st_tr1_000040"]
    qmax, qpt = [float(x) for x in re.search(r"DRIVE_TRACER qmax,q\(10,16,nlev\)=\s*(\S+)\s+(\S+)", stdout).groups()]
    assert abs(qmax - q.max()) < 1e-9 * q.max() and abs(qpt - q[-1, 15, 9]) < 1e-9 * q.max(), (qmax, q.max(), qpt, q[-1, 15, 9])
    (aloft,) = [float(x) for x in re.search(r"DRIVE_TRACER_ALOFT max q\(:,:,nlev-1\)=\s*(\S+)", stdout).groups()]
    assert abs(aloft - q[-2].max()) < 1e-9 * q[-2].max(), (aloft, q[-2].max())
    assert abs(q[-2].max() - plain[-2].max()) > 1e-5 * plain[-2].max()      # (not what finite_volume_parabolic carries up)


@pytest.mark.parametrize("moist", [False, True])
def test_atmosphere_mod_restarts_from_fortran(tmp_path, golden_dir, moist):
    """The restart branch from the Fortran side (read_restart_or_do_coldstart, spectral_dynamics.F90:509-575; spectral_dynamics_end :1502-1531;
    atmosphere.F90:197-223, 362-375; mixed_layer.F90:324-327, 813): atmos_model's loop on this repository's atmosphere_mod runs 20 steps and ends --
    spectral_dynamics_end writes RESTART/spectral_dynamics.res.nc, atmosphere.res.nc (and mixed_layer.res.nc) through the library's netCDF-classic
    writer --, RESTART becomes the next run's INPUT as the harness does it (experiment.py:300-330), and 16 more steps land exactly where the uninterrupted
    36-step run lands.  The files carry the reference's variable set and are the ones isca_amd/restart.py reads."""
    exe = os.path.join(REPO, "oracle", "_ref", "drive_atmos_model_gpu.x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/drive_atmos_model_gpu.x was not built (python oracle/build_ref.py dropin_atmos)")
    from oracle import make_golden as mg
    from scipy.io import netcdf_file

    def run(d, nsteps):
        if moist:
            mg.prepare_moist_rundir(d, "T21", nsteps, dt=720)
        else:
            mg.prepare_rundir(d, "T21", 8, "run", nsteps=nsteps, dt=600)
        open(os.path.join(d, "drive.nml"), "w").write(f" &drive_nml\n   nsteps = {nsteps}, dt_atmos = {720 if moist else 600}\n /\n")
        return d

    def state(stdout):
        vals = [float(x) for x in re.search(r"DRIVE_STATE Tmin,Tmax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)", stdout).groups()]
        vals += [float(x) for x in re.search(r"DRIVE_TRACER qmax,q\(10,16,nlev\)=\s*(\S+)\s+(\S+)", stdout).groups()]
        return vals + [float(re.search(r"DRIVE_MEAN_PS\s*(\S+)", stdout).group(1))]

    from isca_amd import dyncore, restart, atmosphere as atm
    if moist:
        # idealized_moist_phys_init sets the gust back to 1 m/s at every start (constant_gust = 0 afterwards), in the reference as here: the model
        # that equals the restarted run is the one put into the restart state in memory (tests/test_gpu_moist.py::test_moist_restart_round_trip)
        dc = dyncore.DynCore(atm.config_from_namelist(atm.parse_namelist(mg.moist_input_nml("T21", 25)), dt_atmos=720.0))
        dc.cold_start(); dc.step(20)
        dc.set_time_pointers(dc.info("previous"), dc.info("current"), dc.info("step"))
        dc.step(16)
        t, u, q, ps = dc.get("tg"), dc.get("ug"), dc.get("tr"), dc.get("psg")
        whole = [t.min(), t.max(), np.abs(u).max(), q.max(), q[-1, 15, 9], dc.area_weighted_global_mean(ps)]
        dc.close()
    else:
        whole = state(mg.run_harness(run(str(tmp_path / "whole"), 36), exe=exe, timeout=900))
    d1 = run(str(tmp_path / "seg1"), 20)
    mg.run_harness(d1, exe=exe, timeout=900)
    files = ["spectral_dynamics.res.nc", "atmosphere.res.nc"] + (["mixed_layer.res.nc"] if moist else [])
    for fn in files:
        assert os.path.exists(os.path.join(d1, "RESTART", fn)), fn
    f = netcdf_file(os.path.join(d1, "RESTART", "spectral_dynamics.res.nc"), "r", mmap=False)
    for v in ("previous", "current", "pk", "bk", "vors_real", "vors_imag", "divs_real", "divs_imag", "ts_real", "ts_imag", "ln_ps_real", "ln_ps_imag",
              "ug", "vg", "tg", "psg", "sphum", "vorg", "divg", "surf_geopotential"):
        assert v in f.variables, v
    assert f.variables["psg"].shape == (2, 1, 32, 64) and f.variables["vors_real"].shape[2:] == (23, 22)
    f.close()
    d2 = run(str(tmp_path / "seg2"), 16)
    os.rename(os.path.join(d1, "RESTART"), os.path.join(d2, "INPUT"))
    cont = state(mg.run_harness(d2, exe=exe, timeout=900))
    print("36 steps:", whole, " 20 + restart + 16:", cont)
    assert cont[:5] == whole[:5] and abs(cont[5] - whole[5]) < 1e-9, (whole, cont)      # (the mean's summation order differs between Fortran's and the library's)
    # ... and the Python mirror reads the Fortran run's files
    if not moist:
        dc = dyncore.DynCore(dyncore.default_config("T21", num_levels=8))
        restart.read_restart(dc, os.path.join(d2, "INPUT"))
        assert dc.info("previous") != dc.info("current")
        dc.step(16)
        t, u = dc.get("tg"), dc.get("ug")
        assert [t.min(), t.max(), np.abs(u).max()] == whole[:3]
        dc.close()


@pytest.mark.parametrize("nranks,moist", [(2, False), (4, False), (2, True)])
def test_atmosphere_mod_sharded_fortran_host(tmp_path, nranks, moist):
    """A multi-rank Fortran host (the decomposition contract of spec_mpp.F90:61-80 / atmosphere_domain, atmosphere.F90:390): two processes of atmos_model's
    loop on this repository's atmosphere_mod, each holding a latitude band (get_grid_domain returns its rows), the library dealing the zonal
    wavenumbers and issuing the lat <-> m exchanges itself (transforms.F90:970-1056) -- rank and number of ranks from the environment
    (isca_env_rank: this mpp has no MPI), the communicator's id through ISCA_COMM_ID_FILE, ISCA_COMM=ipc because the two share this box's GPU.
    36 steps against the one-process run; then 20 steps, RESTART/*.res.nc.NNNN (every rank's piece of a distributed file), and 16 more from INPUT/:
    exactly where the two-rank run without the restart lands."""
    import subprocess
    exe = os.path.join(REPO, "oracle", "_ref", "drive_atmos_model_gpu.x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/drive_atmos_model_gpu.x was not built (python oracle/build_ref.py dropin_atmos)")
    from oracle import make_golden as mg

    def prepare(d, nsteps):
        if moist:
            mg.prepare_moist_rundir(d, "T21", nsteps, dt=720)
        else:
            mg.prepare_rundir(d, "T21", 8, "run", nsteps=nsteps, dt=600)
        open(os.path.join(d, "drive.nml"), "w").write(f" &drive_nml\n   nsteps = {nsteps}, dt_atmos = {720 if moist else 600}\n /\n")
        return d

    def run_ranks(d, nranks):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ISCA_COMM="ipc", ISCA_IPC_TIMEOUT_S="300", ISCA_WORLD_SIZE=str(nranks), ISCA_LOCAL_RANK="0",
                   ISCA_COMM_ID_FILE=os.path.join(d, "comm_id"))
        procs = [subprocess.Popen(f"ulimit -s unlimited; exec {exe}", shell=True, cwd=d, executable="/bin/bash", text=True, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, env=dict(env, ISCA_RANK=str(r))) for r in range(nranks)]
        outs = [p.communicate(timeout=600)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
        rows, st = [], []
        for o in outs:
            rows.append(tuple(int(x) for x in re.search(r"DRIVE_ROWS js,je=\s*(\d+)\s+(\d+)", o).groups()))
            st.append([float(x) for x in re.search(r"DRIVE_STATE Tmin,Tmax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)", o).groups()])
        return rows, [min(s[0] for s in st), max(s[1] for s in st), max(s[2] for s in st)]

    rows1, one = run_ranks(prepare(str(tmp_path / "one"), 36), 1)
    rows2, two = run_ranks(prepare(str(tmp_path / "two"), 36), nranks)
    per = 32 // nranks
    assert rows1 == [(1, 32)] and rows2 == [(r * per + 1, (r + 1) * per) for r in range(nranks)], (rows1, rows2)
    print("one rank:", one, f" {nranks} ranks:", two)
    assert max(abs(a - b) for a, b in zip(one, two)) < (1e-8 if moist else 1e-10), (one, two)
    d1 = prepare(str(tmp_path / "seg1"), 20)
    run_ranks(d1, nranks)
    for fn in ("spectral_dynamics.res.nc", "atmosphere.res.nc") + (("mixed_layer.res.nc",) if moist else ()):
        for r in range(nranks):
            assert os.path.exists(os.path.join(d1, "RESTART", f"{fn}.{r:04d}")), (fn, r)
    d2 = prepare(str(tmp_path / "seg2"), 16)
    os.rename(os.path.join(d1, "RESTART"), os.path.join(d2, "INPUT"))
    _, cont = run_ranks(d2, nranks)
    if moist:       # a start sets the gust back to 1 m/s (idealized_moist_phys_init; constant_gust = 0 afterwards), in the reference as here: the
        s1 = prepare(str(tmp_path / "seg1_one"), 20)          # restarted sharded run is compared with the restarted ONE-rank run, not with the uninterrupted one
        run_ranks(s1, 1)
        s2 = prepare(str(tmp_path / "seg2_one"), 16)
        os.rename(os.path.join(s1, "RESTART"), os.path.join(s2, "INPUT"))
        _, cont1 = run_ranks(s2, 1)
        assert max(abs(a - b) for a, b in zip(cont1, cont)) < 1e-8, (cont1, cont)
    else:
        assert cont == two, (two, cont)


HS_DIAG_TABLE = """"FMS Model results"
0 0 0 0 0 0
# = output files =
# file_name, output_freq, output_units, format, time_units, long_name
"atmos_6hourly", 6, "hours", 1, "days", "time",

# = diagnostic field entries =  (exp/test_cases/held_suarez/held_suarez_test_case.py:27-38, on 6-hourly means)
# module_name, field_name, output_name, file_name, time_sampling, time_avg, other_opts, precision
"dynamics", "ps", "ps", "atmos_6hourly", "all", .true., "none", 2,
"dynamics", "bk", "bk", "atmos_6hourly", "all", .false., "none", 2,
"dynamics", "pk", "pk", "atmos_6hourly", "all", .false., "none", 2,
"dynamics", "ucomp", "ucomp", "atmos_6hourly", "all", .true., "none", 2,
"dynamics", "vcomp", "vcomp", "atmos_6hourly", "all", .true., "none", 2,
"dynamics", "temp", "temp", "atmos_6hourly", "all", .true., "none", 2,
"dynamics", "vor", "vor", "atmos_6hourly", "all", .true., "none", 2,
"dynamics", "div", "div", "atmos_6hourly", "all", .true., "none", 2,
"""


@pytest.mark.parametrize("nranks", [1, 2])
def test_atmosphere_mod_writes_history_files(tmp_path, nranks):
    """The Fortran host gets its diagnostics: with the Held-Suarez test case's diag_table in the run directory, atmos_model's loop on the drop-in
    (spectral_dynamics_init -> isca_dyn_diag_open, spectral_dynamics_end -> isca_dyn_diag_close; the device accumulates, the library writes) leaves
    atmos_6hourly.nc -- the file the Python host mirror writes for the same run (isca_amd/diag.py), every variable equal to 1e-15; with two ranks one
    piece per rank (atmos_6hourly.nc.0000, .0001: diag_manager's naming of a distributed file), whose latitude bands together are that file to
    the sharded run's rounding."""
    import subprocess
    from scipy.io import netcdf_file
    exe = os.path.join(REPO, "oracle", "_ref", "drive_atmos_model_gpu.x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/drive_atmos_model_gpu.x was not built (python oracle/build_ref.py dropin_atmos)")
    from oracle import make_golden as mg
    from isca_amd import dyncore
    from isca_amd.diag import DiagTable, History
    nsteps, L = 72, 8
    d = str(tmp_path / "run")
    mg.prepare_rundir(d, "T21", L, "run", nsteps=nsteps, dt=600)
    open(os.path.join(d, "diag_table"), "w").write(HS_DIAG_TABLE)
    open(os.path.join(d, "drive.nml"), "w").write(f" &drive_nml\n   nsteps = {nsteps}, dt_atmos = 600\n /\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ISCA_COMM="ipc", ISCA_IPC_TIMEOUT_S="120", ISCA_WORLD_SIZE=str(nranks), ISCA_LOCAL_RANK="0",
               ISCA_COMM_ID_FILE=os.path.join(d, "comm_id"))
    procs = [subprocess.Popen(f"ulimit -s unlimited; exec {exe}", shell=True, cwd=d, executable="/bin/bash", text=True, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, env=dict(env, ISCA_RANK=str(r))) for r in range(nranks)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    # the same run through the Python host mirror
    diag = DiagTable()
    diag.add_file("atmos_6hourly", 6, "hours", time_units="days")
    for nm, avg in (("ps", True), ("bk", False), ("pk", False), ("ucomp", True), ("vcomp", True), ("temp", True), ("vor", True), ("div", True)):
        diag.add_field("dynamics", nm, time_avg=avg)
    c = dyncore.DynCore(dyncore.default_config("T21", num_levels=L)); c.cold_start()
    hist = History(c, diag.files["atmos_6hourly"], 600.0, str(tmp_path / "py.nc"))
    for _ in range(nsteps // 36):
        c.step(36); hist.after_steps(36)
    hist.close(); c.close()
    ref = netcdf_file(str(tmp_path / "py.nc"), "r", mmap=False)
    pieces = [netcdf_file(os.path.join(d, "atmos_6hourly.nc" + (f".{r:04d}" if nranks > 1 else "")), "r", mmap=False) for r in range(nranks)]
    assert set(pieces[0].variables) == set(ref.variables)
    tol = 1e-15 if nranks == 1 else 1e-11
    for nm, v in ref.variables.items():
        if "lat" in v.dimensions:
            ax = v.dimensions.index("lat")
            mine = np.concatenate([p.variables[nm][:] for p in pieces], axis=ax)
        else:
            mine = pieces[0].variables[nm][:]
        assert mine.shape == v.shape, nm
        scale = max(float(np.abs(v[:]).max()), 1e-300)
        assert float(np.abs(mine - v[:]).max()) <= tol * scale, (nm, float(np.abs(mine - v[:]).max()) / scale)
    assert ref.variables["temp"].shape[0] == 2 and pieces[0].variables["temp"].cell_methods == b"time: mean"
    for f in pieces + [ref]:
        f.close()


def test_atmosphere_mod_graceful_shutdown(tmp_path):
    """spectral_dynamics_nml: graceful_shutdown = .true. (spectral_dynamics.F90:976-1005: diag_manager_end before the FATAL) through the Fortran host:
    a run that blows up (dt_atmos far beyond the CFL limit) ends in the reference's FATAL 'temperatures out of valid range' -- not in a fault --, and the
    run directory's history file is closed and holds the records of the intervals completed before it."""
    import subprocess
    from scipy.io import netcdf_file
    exe = os.path.join(REPO, "oracle", "_ref", "drive_atmos_model_gpu.x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/drive_atmos_model_gpu.x was not built (python oracle/build_ref.py dropin_atmos)")
    from oracle import make_golden as mg
    d = str(tmp_path / "run")
    mg.prepare_rundir(d, "T21", 8, "run", nsteps=200, dt=21600, extra="graceful_shutdown = .true.")
    open(os.path.join(d, "diag_table"), "w").write(HS_DIAG_TABLE)
    open(os.path.join(d, "drive.nml"), "w").write(" &drive_nml\n   nsteps = 200, dt_atmos = 21600\n /\n")
    r = subprocess.run(f"ulimit -s unlimited; exec {exe}", shell=True, cwd=d, executable="/bin/bash", text=True, capture_output=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode != 0 and "temperatures out of valid range" in out and "FATAL" in out, out[-2000:]
    f = netcdf_file(os.path.join(d, "atmos_6hourly.nc"), "r", mmap=False)
    n = f.variables["temp"].shape[0]
    assert 1 <= n <= 200 and f.variables["time"].shape[0] == n, n      # (the host queues its steps: the range check runs at a synchronisation)
    f.close()


def test_atmosphere_mod_input_topography_from_fortran(tmp_path, golden_dir):
    """topography_option = 'input' from Fortran (get_topography, spectral_init_cond.F90:186-245): spectral_dynamics_init reads zsurf and land_mask of
    INPUT/topography.data.nc with the library's netCDF-classic reader and hands them to isca_dyn_set_topography -- regularised over the ocean with the
    namelist's ocean_topog_smoothing (topog_regularization_mod; fixture of the reference's routines: test_golden_ocean_topog_smoothing).  24 steps
    through atmos_model's loop land where the Python mirror lands with the same field and mask handed over."""
    exe = os.path.join(REPO, "oracle", "_ref", "drive_atmos_model_gpu.x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/drive_atmos_model_gpu.x was not built (python oracle/build_ref.py dropin_atmos)")
    from oracle import make_golden as mg
    from scipy.io import netcdf_file
    from isca_amd import atmosphere as atm, dyncore
    g = np.load(os.path.join(golden_dir, "topog_regularize_T21.npz"))
    smoothing = float(g["meta_ocean_topog_smoothing"])
    d = str(tmp_path / "run")
    groups = " &spectral_init_cond_nml\n    topography_option = 'input'\n /\n"
    mg.prepare_rundir(d, "T21", 8, "run", nsteps=24, dt=600, extra=f"ocean_topog_smoothing = {smoothing}", extra_groups=groups)
    open(os.path.join(d, "drive.nml"), "w").write(" &drive_nml\n   nsteps = 24, dt_atmos = 600\n /\n")
    f = netcdf_file(os.path.join(d, "INPUT", "topography.data.nc"), "w", version=2)        # what the harness's topography files look like: (lat, lon) fields
    f.createDimension("lat", 32); f.createDimension("lon", 64)
    z = f.createVariable("zsurf", "d", ("lat", "lon")); z[:] = g["in_height"]
    lm = f.createVariable("land_mask", "f", ("lat", "lon")); lm[:] = g["in_land"]
    f.close()
    stdout = mg.run_harness(d, exe=exe, timeout=900)
    lam = float(re.search(r"lambda=\s*(\S+)\s+fraction_smoothed=\s*(\S+)", stdout).group(1))
    assert abs(lam / float(g["out_lambda"]) - 1) < 1e-7, (lam, float(g["out_lambda"]))       # (printed with 9 digits)
    vals = [float(x) for x in re.search(r"DRIVE_STATE Tmin,Tmax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)", stdout).groups()]
    nml = atm.parse_namelist(open(os.path.join(d, "input.nml")).read())
    nml["main_nml"] = {"dt_atmos": 600}
    core = atm.atmosphere_init(nml, surf_height=g["in_height"], land_mask=g["in_land"])
    try:
        assert rel_(core.get("surf_geopotential"), g["out_smoothed_geopotential"]) < 1e-9
        atm.atmosphere(24)
        t, u = core.get("tg"), core.get("ug")
        assert [t.min(), t.max(), np.abs(u).max()] == vals, (vals, [t.min(), t.max(), np.abs(u).max()])
    finally:
        atm.atmosphere_end()


def rel_(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
