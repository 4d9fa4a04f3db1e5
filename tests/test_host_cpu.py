"""CPU-side checks of the product's host layer: the C-ABI library loads and exports every symbol that
include/isca_dyn.h declares, configuration errors follow the reference's FATAL conditions, and the
product never imports the oracle.  No compute calls (no GPU here)."""
import os, re, subprocess, sys
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from isca_amd import build, dyncore
    build.build(verbose=False)
    return dyncore.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(REPO, "include", "isca_dyn.h")).read()
    declared = set(re.findall(r"\b(isca_[a-z_0-9]+)\s*\(", hdr))
    declared.discard("isca_dyn_config")
    assert len(declared) >= 25
    from isca_amd import dyncore
    assert declared == set(dyncore.EXPORTED_SYMBOLS), declared ^ set(dyncore.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    # the sibling cores' headers
    from isca_amd import shallow
    for header, names in (("isca_shallow.h", shallow.EXPORTED_SYMBOLS), ("isca_barotropic.h", shallow.BAROTROPIC_SYMBOLS)):
        h = open(os.path.join(REPO, "include", header)).read()
        decl = set(re.findall(r"\b(isca_[a-z_0-9]+)\s*\(", h)) - {"isca_last_error"}      # mentioned in the header comment
        assert decl == set(names), decl ^ set(names)
        for name in decl:
            assert hasattr(lib, name), name
    import ctypes
    sizes = (ctypes.c_size_t * 4)()
    assert lib.isca_config_sizes(sizes, 4) == 0
    assert list(sizes) == [ctypes.sizeof(dyncore._CConfig), ctypes.sizeof(dyncore._CMoistConfig), ctypes.sizeof(shallow._CShallowConfig),
                           ctypes.sizeof(shallow._CBarotropicConfig)]
    for cls, header, tname in ((shallow._CShallowConfig, "isca_shallow.h", "isca_shallow_config"),
                               (shallow._CBarotropicConfig, "isca_barotropic.h", "isca_barotropic_config")):
        h = open(os.path.join(REPO, "include", header)).read()
        body = re.sub(r"/\*.*?\*/", "", h[h.index("typedef struct %s {" % tname) + len("typedef struct %s {" % tname):h.index("} %s;" % tname)], flags=re.S)
        members = []
        for decl in body.split(";"):
            m = re.match(r"(?:int|double|isca_stirring_config)\s*(.*)", decl.strip(), flags=re.S)
            if m:
                members += [re.sub(r"\[.*\]", "", x).strip() for x in m.group(1).split(",")]
        assert members == [f[0] for f in cls._fields_], (tname, members)


def test_config_struct_matches_header(lib):
    from isca_amd import dyncore
    hdr = open(os.path.join(REPO, "include", "isca_dyn.h")).read()
    start = hdr.index("typedef struct isca_dyn_config {") + len("typedef struct isca_dyn_config {")
    body = hdr[start:hdr.index("} isca_dyn_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    def members(text):
        names = []
        for decl in text.split(";"):
            m = re.match(r"(?:int|double|void \*|isca_moist_config)\s*(.*)", decl.strip(), flags=re.S)
            if m:
                names += [re.sub(r"\[.*\]|\*", "", x).strip() for x in m.group(1).split(",")]
        return names
    assert members(body) == [f[0] for f in dyncore._CConfig._fields_]
    mstart = hdr.index("typedef struct isca_moist_config {") + len("typedef struct isca_moist_config {")
    mbody = re.sub(r"/\*.*?\*/", "", hdr[mstart:hdr.index("} isca_moist_config;")], flags=re.S)
    assert members(mbody) == [f[0] for f in dyncore._CMoistConfig._fields_]
    c = dyncore.default_config("T85", num_levels=40, dt_atmos=300.0)
    assert (c.lon_max, c.lat_max, c.num_fourier, c.num_spherical) == (256, 128, 85, 86)
    assert c.damping_order == 4 and c.robert_coeff == 0.04 and c.reference_sea_level_press == 1.0e5


def test_no_device_fails_loudly(lib):
    """No CPU fallback: without a HIP device creation must fail with a message (skipped on a GPU box)."""
    import ctypes as C
    from isca_amd import dyncore
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    with pytest.raises(dyncore.IscaError, match="no HIP device|no CPU fallback|hip"):
        dyncore.DynCore(dyncore.default_config("T21"))


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "isca_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(root, f), errors="replace").read()
                assert "oracle" not in txt.replace("the oracle", "").lower() or f == "build.py", f


def test_namelist_to_config(lib):
    """The reference's input.nml keys (held_suarez_test_case.py:45-98) map onto the C config; unsupported option
    values are FATAL like in the reference's own checks."""
    from isca_amd import atmosphere as atm, dyncore
    nml_text = '''
 &spectral_dynamics_nml
    damping_order = 4,  water_correction_limit = 200.e2, reference_sea_level_press = 1.0e5,
    valid_range_t = 100., 800., vert_coord_option = 'uneven_sigma', scale_heights = 6.0, exponent = 7.5,
    surf_res = 0.5, lon_max = 128, lat_max = 64, num_fourier = 42, num_spherical = 43, num_levels = 25 /
 &hs_forcing_nml
    t_zero = 315., ka = -40., ks = -4., kf = -1., do_conserve_energy = .true. /
 &main_nml
    dt_atmos = 600, days = 30, calendar = 'thirty_day' /
'''
    c = atm.config_from_namelist(nml_text)
    assert (c.lon_max, c.lat_max, c.num_fourier, c.num_levels) == (128, 64, 42, 25)
    assert c.dt_atmos == 600.0 and c.water_correction_limit == 200.e2 and list(c.valid_range_t) == [100.0, 800.0]
    assert c.do_conserve_energy == 1 and c.ka == -40.0
    # what an input.nml leaves out takes the reference's module defaults (spectral_dynamics.F90:152-206), not the test-case preset
    m = atm.config_from_namelist({"main_nml": {"dt_atmos": 900}, "spectral_dynamics_nml": {"num_levels": 10}})
    assert (m.lon_max, m.num_fourier, m.damping_order, m.vert_coord_input) == (128, 42, 2, 1) and m.reference_sea_level_press == 101325.0
    assert (m.scale_heights, m.exponent, m.surf_res, m.water_correction_limit) == (4.0, 2.5, 0.1, 0.0) and list(m.valid_range_t) == [100.0, 500.0]
    assert [m.bk_input[k] for k in (0, 5, 10)] == [0.0, 0.5, 1.0]                        # vert_coord_option defaults to 'even_sigma'
    assert atm.config_from_namelist(None, "T21").scale_heights == 6.0                    # no namelist: the Held-Suarez test case
    assert atm.config_from_namelist({"spectral_dynamics_nml": {"use_implicit": False}}).use_implicit == 0 and m.use_implicit == 1
    assert atm.config_from_namelist({"spectral_dynamics_nml": {"make_symmetric": True}}).make_symmetric == 1 and m.make_symmetric == 0
    assert atm.config_from_namelist({"spectral_dynamics_nml": {"use_virtual_temperature": True}}).use_virtual_temperature == 1
    with pytest.raises(dyncore.IscaError, match="convection_scheme is not set"):          # moist options whose reference default is not implemented
        atm.config_from_namelist({"atmosphere_nml": {"idealized_moist_model": True}})
    from isca_amd import configs
    fr = atm.config_from_namelist(configs.frierson())
    assert fr.physics == 1 and fr.moist.depth == 2.5 and fr.moist.ir_tau_eq == 6.0 and fr.moist.tau_bm == 7200.0   # set / module default
    d = {"spectral_dynamics_nml": {"damping_order": 4, "vert_coord_option": "pressure"}}
    with pytest.raises(dyncore.IscaError, match="vert_coord_option"):
        atm.config_from_namelist(d, "T21")
    with pytest.raises(dyncore.IscaError, match="not initialized"):
        atm.atmosphere()


def test_restart_file_round_trip_on_host(tmp_path):
    """restart.py against a host stand-in for the device handle: variable set, record <-> time-level mapping,
    1-based pointers (spectral_dynamics.F90:1502-1531, atmosphere.F90:362-375) and the FATAL resolution check."""
    import types
    import numpy as np
    from isca_amd import restart
    from isca_amd.dyncore import IscaError
    from scipy.io import netcdf_file

    class Host:
        L, J, Jl, I, N1, M1 = 3, 8, 8, 16, 7, 6
        cfg = types.SimpleNamespace(world_size=1, physics=0, num_tracers=3, tracer_spectral=[0, 0, 1])
        tracer_names = ["sphum", "age_grid", "age_spec"]

        def __init__(self, seed=None):
            self.ptr = {"previous": 1, "current": 0, "tracer": 1, "step": 5}
            self.store, self.refreshed = {}, False
            if seed is not None:
                rng = np.random.default_rng(seed)
                for tl in (0, 1):
                    for nm in ("vors", "divs", "ts", "trs3"):
                        self.store[nm, tl] = rng.standard_normal((3, 7, 6)) + 1j * rng.standard_normal((3, 7, 6))
                    self.store["ln_ps", tl] = rng.standard_normal((7, 6)) + 1j * rng.standard_normal((7, 6))
                    for nm in ("ug", "vg", "tg", "tr", "tr_atm", "tr2", "tr_atm2", "tr3", "tr_atm3", "vorg", "divg", "wg_full"):
                        self.store[nm, tl] = rng.standard_normal((3, 8, 16))
                    self.store["psg", tl] = 1e5 + rng.standard_normal((8, 16))
                for nm in ("vorg", "divg", "wg_full"):
                    self.store[nm, 0] = self.store[nm, 1]
                self.store["surf_geopotential", 1] = 100.0 * rng.standard_normal((8, 16))

        def info(self, k): return self.ptr[k]
        def table(self, k): return np.linspace(0, 1, 4) if k == "bk" else np.zeros(4)
        def get(self, nm, tl=1): return self.store[nm, tl]
        def set(self, nm, v, tl=1): self.store[nm, tl] = np.array(v)
        def set_surf_geopotential(self, v): self.store["surf_geopotential", 1] = np.array(v)
        def set_time_pointers(self, p, c, s): self.ptr.update(previous=p, current=c, step=s)
        def refresh_derived(self): self.refreshed = True

    a = Host(seed=3)
    restart.write_restart(a, str(tmp_path))
    f = netcdf_file(str(tmp_path / "spectral_dynamics.res.nc"), "r", mmap=False)
    assert float(f.variables["previous"][0].ravel()[0]) == 2.0 and float(f.variables["current"][0].ravel()[0]) == 1.0
    # record 0 is Fortran time level 1 = storage slot 0 = `current` here
    assert np.array_equal(f.variables["ug"][0], a.store["ug", 1]) and np.array_equal(f.variables["ug"][1], a.store["ug", 0])
    dims = f.variables["vors_real"].dimensions
    assert dims[0] == "Time" and [d[:5] for d in dims[1:]] == ["zaxis", "yaxis", "xaxis"]
    assert f.variables["vors_real"].shape == (2, 3, 7, 6) and f.variables["psg"].shape == (2, 1, 8, 16)
    # every field_table tracer under its own name, a spectral one also as <name>_real / <name>_imag (spectral_dynamics.F90:1520-1527)
    assert np.array_equal(f.variables["age_grid"][1], a.store["tr2", 0]) and np.array_equal(f.variables["age_spec"][0], a.store["tr3", 1])
    assert np.array_equal(f.variables["age_spec_imag"][0], a.store["trs3", 1].imag) and "age_grid_real" not in f.variables
    f.close()
    b = Host()
    restart.read_restart(b, str(tmp_path))
    assert b.refreshed and (b.ptr["previous"], b.ptr["current"]) == (1, 0)
    for key, val in a.store.items():
        if key[0] in ("vorg", "divg") or key == ("wg_full", 0):
            continue                                   # rebuilt by refresh_derived on the device
        assert np.array_equal(b.store[key], val), key
    c = Host()
    c.L = 4
    with pytest.raises(IscaError, match="num_levels=   3|num_levels=3"):
        restart.read_restart(c, str(tmp_path))


def test_experiment_host_logic(tmp_path):
    """Run segmentation mirror (experiment.py:198-346): segment length, namelist file, restart chaining errors."""
    from isca_amd.experiment import Experiment
    from isca_amd.atmosphere import parse_namelist
    from isca_amd.dyncore import IscaError
    exp = Experiment("hs", str(tmp_path))
    exp.set_resolution("T42", 25)
    exp.update_namelist({"main_nml": {"days": 30, "hours": 0, "dt_atmos": 600, "calendar": "thirty_day"},
                         "spectral_dynamics_nml": {"valid_range_t": [100., 800.], "vert_coord_option": "uneven_sigma"},
                         "atmosphere_nml": {"idealized_moist_model": False}})
    assert exp.steps_per_run() == 30 * 144
    assert exp.namelist["spectral_dynamics_nml"]["lon_max"] == 128 and exp.namelist["spectral_dynamics_nml"]["num_levels"] == 25
    os.makedirs(exp.rundir)
    exp.write_namelist(exp.rundir)
    back = parse_namelist(open(os.path.join(exp.rundir, "input.nml")).read())
    assert back["main_nml"]["calendar"] == "thirty_day" and back["spectral_dynamics_nml"]["valid_range_t"] == [100.0, 800.0]
    assert back["atmosphere_nml"]["idealized_moist_model"] is False
    assert exp.get_restart_file(3).endswith(os.path.join("hs", "restarts", "res0003.tar.gz"))
    with pytest.raises(IOError, match="Restart file not found"):
        exp.run(2)                                    # no res0001.tar.gz
    exp.update_namelist({"main_nml": {"days": 0, "seconds": 700}})
    with pytest.raises(IscaError):
        exp.steps_per_run()


def test_diag_table_host_logic():
    from isca_amd.diag import DiagTable, FIELDS
    from isca_amd.dyncore import IscaError
    d = DiagTable()
    d.add_file("atmos_monthly", 30, "days", time_units="days")
    for nm in ("ps", "bk", "pk", "ucomp", "vcomp", "temp", "vor", "div"):          # held_suarez_test_case.py:30-38
        d.add_field("dynamics", nm, time_avg=nm not in ("bk", "pk"))
    assert [f["name"] for f in d.files["atmos_monthly"]["fields"]][:3] == ["ps", "bk", "pk"] and d.is_valid()
    with pytest.raises(IscaError):
        d.add_field("two_stream", "olr")                     # outside the dynamical core
    with pytest.raises(IscaError):
        d.add_field("dynamics", "no_such_field")
    assert {"ucomp_vcomp", "omega", "wspd", "vcomp_vor"} <= set(FIELDS)


def test_moist_namelist_mapping():
    """idealized_moist_model = .true.: the Frierson test case's namelists become the C config; options the device package does not
    implement are refused like unsupported namelist values (no GPU needed: only the config is built)."""
    from isca_amd import atmosphere as atm, dyncore
    bk = [0.0, 0.2, 0.5, 0.8, 1.0]
    from isca_amd import configs
    nml = configs.frierson()                       # the test case's option switches; some values changed, some left to the module defaults
    nml.update({"atmosphere_nml": {"idealized_moist_model": True}, "main_nml": {"dt_atmos": 720},
                "spectral_dynamics_nml": {"num_levels": 4, "vert_coord_option": "input", "robert_coeff": 0.03, "initial_sphum": 2e-6},
                "vert_coordinate_nml": {"bk": bk, "pk": [0.0] * 5},
                "two_stream_gray_rad_nml": {"rad_scheme": "frierson", "atm_abs": 0.2, "do_seasonal": False},
                "mixed_layer_nml": {"depth": 10.0, "albedo_value": 0.25, "delta_T": 30.0, "evaporation": True, "prescribe_initial_dist": True},
                "qe_moist_convection_nml": {"rhbm": 0.8, "Tmin": 150.0},
                "damping_driver_nml": {"do_rayleigh": True, "trayfric": -0.5, "do_conserve_energy": False}})
    c = atm.config_from_namelist(nml, resolution="T21")
    assert c.physics == 1 and c.vert_coord_input == 1 and [c.bk_input[i] for i in range(5)] == bk and c.num_levels == 4
    m = c.moist
    assert (m.atm_abs, m.depth, m.albedo_value, m.delta_T, m.evaporation) == (0.2, 10.0, 0.25, 30.0, 1)
    assert (m.rhbm, m.Tmin, m.Tmax, m.trayfric, m.damping_conserve_energy) == (0.8, 150.0, 335.0, -0.5, 0)
    assert m.roughness_mom == 3.21e-05 and m.rich_crit == 2.0 and m.tconst == 305.0      # test case value / module defaults
    text = "&atmosphere_nml idealized_moist_model = .true. /\n&two_stream_gray_rad_nml rad_scheme = 'byrne' /\n"
    with pytest.raises(dyncore.IscaError, match="not a supported value for rad_scheme"):
        atm.config_from_namelist(text)
    with pytest.raises(dyncore.IscaError, match="vert_coordinate_nml"):
        atm.config_from_namelist({"spectral_dynamics_nml": {"vert_coord_option": "input"}})
    with pytest.raises(dyncore.IscaError, match="num_levels\\+1"):
        atm.config_from_namelist({"spectral_dynamics_nml": {"vert_coord_option": "input", "num_levels": 7}, "vert_coordinate_nml": {"bk": bk}})
    with pytest.raises(dyncore.IscaError, match="not supported by the device physics"):
        atm.config_from_namelist({"atmosphere_nml": {"idealized_moist_model": True}, "mixed_layer_nml": {"land_depth": 2.0}})
    ev = atm.config_from_namelist({"spectral_dynamics_nml": {"num_levels": 4, "vert_coord_option": "even_sigma"}})
    assert ev.vert_coord_input == 1 and [ev.bk_input[i] for i in range(5)] == [0.0, 0.25, 0.5, 0.75, 1.0]
    mars = atm.config_from_namelist({"constants_nml": {"radius": 3389.5e3, "omega": 7.088e-5}})
    assert (mars.radius, mars.omega) == (3389.5e3, 7.088e-5)
    with pytest.raises(dyncore.IscaError, match="only radius and omega"):
        atm.config_from_namelist({"constants_nml": {"grav": 3.71}})
    dry = atm.config_from_namelist({"spectral_dynamics_nml": {"num_levels": 25, "vert_coord_option": "uneven_sigma"}})
    assert dry.physics == 0 and dry.vert_coord_input == 0


def test_sibling_core_namelists():
    """shallow_dynamics_nml / shallow_physics_nml / barotropic_dynamics_nml -> C configs (no GPU needed)."""
    from isca_amd import shallow
    from isca_amd.dyncore import IscaError
    c = shallow.config_from_namelist({"shallow_dynamics_nml": {"num_lon": 128, "num_lat": 64, "num_fourier": 42, "num_spherical": 43, "h_0": 2.e4,
                                                                 "robert_coeff": 0.03, "grid_tracer": False, "valid_range_v": [-500., 500.]},
                                      "shallow_physics_nml": {"h_0": 2.5e4, "therm_damp_time": -5.0, "del_h": 1.}, "main_nml": {"dt_atmos": 600}})
    assert (c.num_lon, c.num_fourier, c.h_0, c.phys_h_0, c.therm_damp_time, c.robert_coeff, c.grid_tracer, c.dt_atmos) == \
        (128, 42, 2.e4, 2.5e4, -5.0, 0.03, 0, 600.0)
    assert c.valid_range_v[1] == 500.0 and c.fric_damp_time == -20.0 and c.spec_tracer == 1
    b = shallow.barotropic_config_from_namelist({"barotropic_dynamics_nml": {"initial_zonal_wind": "zero", "m_0": 6, "damping_coeff_r": 1e-6}}, "T42")
    assert (b.initial_zonal_wind, b.m_0, b.damping_coeff_r, b.num_lat, b.zeta_0) == (0, 6, 1e-6, 64, 8.e-05)
    st = shallow.barotropic_config_from_namelist({"stirring_nml": {"amplitude": 3.e-11, "decay_time": 172800, "B": 1.0, "widthx": 45.}}).stirring
    assert (st.amplitude, st.decay_time, st.B, st.lat0, st.n_total_forcing_max, st.do_localize) == (3.e-11, 172800.0, 1.0, 45.0, 15, 1)
    with pytest.raises(IscaError, match="stirring_nml: unknown variable"):
        shallow.config_from_namelist({"stirring_nml": {"colour": "red"}})
    with pytest.raises(IscaError, match="unknown shallow-water configuration key"):
        shallow.config_from_namelist({"shallow_dynamics_nml": {"no_such_key": 1}})
    with pytest.raises(IscaError, match="not a supported value for triang_trunc"):
        shallow.barotropic_config_from_namelist({"barotropic_dynamics_nml": {"triang_trunc": False}})


def test_test_case_config_files(lib):
    """isca_amd/configs.py holds the namelists of the two reference test cases key for key; both map onto the C config."""
    from isca_amd import atmosphere as atm, configs
    hs = atm.config_from_namelist(configs.held_suarez(), resolution="T42", num_levels=25)     # exp.set_resolution('T42', 25) of the test case
    assert (hs.physics, hs.num_levels, hs.dt_atmos, hs.lat_max, hs.scale_heights, hs.ka, hs.initial_sphum) == (0, 25, 600.0, 64, 6.0, -40.0, 0.0)
    assert atm.config_from_namelist(configs.held_suarez(), resolution="T42").num_levels == 18    # spectral_dynamics_nml's own default
    fr = atm.config_from_namelist(configs.frierson(), resolution="T42")
    assert (fr.physics, fr.num_levels, fr.dt_atmos, fr.initial_sphum, fr.robert_coeff, fr.vert_coord_input) == (1, 25, 720.0, 2.e-6, 0.03, 1)
    assert fr.bk_input[1] == 0.0117665 and fr.moist.atm_abs == 0.2 and fr.moist.depth == 2.5 and fr.moist.trayfric == -0.25
    assert fr.moist.rhbm == 0.7 and fr.moist.Tmin == 160.0 and fr.moist.constant_gust == 0.0


def test_integration_doc_in_sync():
    """INTEGRATION.md prints bindings/fortran/isca_dyn_c.F90 (the block a maintainer copies): regenerate with
    tools/gen_integration_snippet.py when the module changes."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_integration_snippet.py"), "--check"])
    assert r.returncode == 0, "INTEGRATION.md is out of date: run python tools/gen_integration_snippet.py"


def test_fortran_binding_abi(lib, tmp_path):
    """bindings/fortran/isca_dyn_c.F90 (the bind(C) module for the reference's language) compiles with the image's flang and its
    derived types have the library's struct sizes; defaults read back through them (no GPU needed)."""
    flang = os.environ.get("FLANG", "/opt/rocm/lib/llvm/bin/flang")
    if not os.path.exists(flang):
        pytest.skip("no flang in this image")
    src, libdir = os.path.join(REPO, "bindings", "fortran"), os.path.join(REPO, "isca_amd", "lib")
    mod_o, exe = str(tmp_path / "isca_dyn_c.o"), str(tmp_path / "check_abi.x")
    sib_o = str(tmp_path / "isca_siblings_c.o")
    subprocess.run([flang, "-c", os.path.join(src, "isca_dyn_c.F90"), "-o", mod_o, "-module-dir", str(tmp_path)], check=True, capture_output=True)
    subprocess.run([flang, "-c", os.path.join(src, "isca_siblings_c.F90"), "-o", sib_o, "-module-dir", str(tmp_path)], check=True, capture_output=True)
    subprocess.run([flang, os.path.join(src, "check_abi.F90"), mod_o, sib_o, "-I", str(tmp_path), "-L", libdir, "-lisca_dyn", "-Wl,-rpath," + libdir,
                    "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ABI_OK" in r.stdout, r.stdout + r.stderr
    assert r.stdout.split("ABI_OK")[1].split()[:5] == ["64", "32", "21", "22", "25"]
    vals = [float(x) for x in r.stdout.split("DEFAULTS")[1].split()[:4]]
    assert vals == [0.04, 0.2, 6376.0e3, 800.0]
    tail = r.stdout.split("TAIL")[1].split()[:7]         # tracer_robert_coeff(8), tracer_sink(8), use_implicit, tracer_hole_filling(8), tracer_sms(1), tracer_advect_vert(1), (8)
    assert [float(x) for x in tail[:2]] == [-1.0, 0.0] and [int(x) for x in tail[2:]] == [1, 0, 0, -1, -1]
    sib = r.stdout.split("SIBLINGS")[1].split()[:4]
    assert [float(x) for x in sib[:3]] == [3.e4, 172800.0, 8.e-5] and int(sib[3]) == 4


def test_field_table_entries():
    """tracer_manager's field_table as spectral_dynamics_init reads it (spectral_dynamics.F90:316-409): representation, vertical
    scheme, the tracer's own robert_coeff; what the kernels do not implement is refused by name."""
    from isca_amd import atmosphere as atm
    from isca_amd.dyncore import IscaError
    text = '''# dry default plus two more
"TRACER", "atmos_mod", "sphum"
          "longname",  "specific humidity"
          "numerical_representation", "grid"
          "hole_filling", "off"
          "advect_vert", "finite_volume_parabolic"
          "robert_filter", "on"
          "profile_type", "fixed", "surface_value=0.0" /
"TRACER", "land_mod", "sphum"
          "longname", "not ours" /
"TRACER", "atmos_mod", "Age_Grid"
          "numerical_representation", "grid"
          "advect_vert", "finite_volume_parabolic"
          "robert_filter", "on", "robert_coeff=0.05" /
"TRACER", "atmos_mod", "age_spec"
          "robert_filter", "off" /
'''
    entries = atm.parse_field_table(text)
    assert [e["name"] for e in entries] == ["sphum", "age_grid", "age_spec"]
    keys, names = atm.tracers_from_field_table(entries, 0.03)
    assert keys == dict(num_tracers=3, tracer_spectral=[0, 0, 1], tracer_robert_coeff=[-1.0, 0.05, 0.0], tracer_hole_filling=[0, 0, 0]) and names[2] == "age_spec"
    # hole_filling = on: water_borrowing on a spectral tracer's tendency (spectral_dynamics.F90:1142); ignored with the reference's warning for a grid tracer (:364-367)
    k4, _ = atm.tracers_from_field_table(atm.parse_field_table(text + '"TRACER", "atmos_mod", "y"\n "hole_filling", "on" /'))
    assert k4["tracer_hole_filling"] == [0, 0, 0, 1] and k4["tracer_spectral"][3] == 1
    # advect_vert other than the representation's fused scheme goes over by number (tracer_advect_vert; -1 = the standard one)
    kv, _ = atm.tracers_from_field_table(atm.parse_field_table(text + '"TRACER", "atmos_mod", "x"\n "numerical_representation", "grid" /\n'
                                                               '"TRACER", "atmos_mod", "y"\n "advect_vert", "van_leer_linear" /'))
    assert kv["tracer_advect_vert"] == [-1, -1, -1, 0, 2] and "tracer_advect_vert" not in keys
    for bad, msg in (
                     ('"TRACER", "atmos_mod", "x"\n "numerical_representation", "wavelet" /', "invalid numerical_representation"),
                     ('"TRACER", "atmos_mod", "x"\n "advect_vert", "upwind" /', "invalid advect_vert"),
                     ('"TRACER", "atmos_mod", "x" /', "must be a grid tracer"),
                     (text * 3, "at most 8")):
        with pytest.raises(IscaError, match=msg):
            atm.tracers_from_field_table(atm.parse_field_table(bad))
    # the layout of the reference's own tables (src/extra/model/*/field_table): a '/' and commas inside quoted fields, tabs, parameter strings
    ref_style = ('"TRACER", "atmos_mod", "sphum"\n          "longname",  "specific humidity"\n          "units",     "kg/kg"\n'
                 '          "numerical_representation", "grid"\n\t  "hole_filling",             "off"\n'
                 '          "advect_vert",              "finite_volume_parabolic"\n          "robert_filter",            "on"\n'
                 '          "tracer_sms", "on", "flux=2.5e-5, sink=-2.0"\n          "profile_type", "fixed",   "surface_value=0.0" /\n'
                 '"TRACER", "atmos_mod", "age"\n "units", "m/s/day"\n "tracer_sms", "OFF" /\n"TRACER", "atmos_mod", "dust"\n "tracer_sms", "on", "sink=3600." /\n')
    ent = atm.parse_field_table(ref_style)
    assert [e["name"] for e in ent] == ["sphum", "age", "dust"] and ent[0]["methods"]["units"] == ("kg/kg", "")
    assert ent[0]["methods"]["tracer_sms"] == ("on", "flux=2.5e-5, sink=-2.0") and ent[0]["methods"]["profile_type"] == ("fixed", "surface_value=0.0")
    # tracer_sms (hs_forcing.F90:251-261): own flux / sink, the one left out from hs_forcing_nml (here trflux = 3e-5), 'off' = no source and no sink
    ksms, _ = atm.tracers_from_field_table(ent, None, 3.e-5, -5.0)
    assert ksms["tracer_sms"] == [1, 1, 1] and ksms["tracer_flux"] == [2.5e-5, 0.0, 3.e-5] and ksms["tracer_sink"] == [-2.0, 0.0, 3600.]
    assert "tracer_sms" not in keys
    # the humidity tracer is found by NAME (nhum = get_tracer_index('sphum' | 'mix_rat')): it must be tracer 1; without one the model is dry
    grid = '"TRACER", "atmos_mod", "%s"\n "numerical_representation", "grid"\n "advect_vert", "finite_volume_parabolic" /\n'
    with pytest.raises(IscaError, match="must be the first atmos_mod entry"):
        atm.tracers_from_field_table(atm.parse_field_table(grid % "age" + grid % "sphum"))
    keys, names = atm.tracers_from_field_table(atm.parse_field_table(grid % "age"))
    assert keys["_dry_model"] and keys["initial_sphum"] == 0.0 and keys["use_virtual_temperature"] is False and names == ["age"]
    keys, _ = atm.tracers_from_field_table(atm.parse_field_table(grid % "mix_rat"))
    assert "_dry_model" not in keys
    # a robert_coeff of its own on tracer 1 is compared with the dynamics' effective value: the module default 0.04 when the namelist omits it
    own = '"TRACER", "atmos_mod", "sphum"\n "numerical_representation", "grid"\n "advect_vert", "finite_volume_parabolic"\n "robert_filter", "on", "robert_coeff=%s" /'
    with pytest.raises(IscaError, match="robert_coeff of its own"):
        atm.tracers_from_field_table(atm.parse_field_table(own % "0.05"))
    atm.tracers_from_field_table(atm.parse_field_table(own % "0.04"))
    atm.tracers_from_field_table(atm.parse_field_table(own % "0.05"), 0.05)


def test_named_vertical_coordinates(golden_dir):
    """compute_vert_coord's 'hybrid', 'mcm' and 'v197' (init/vert_coordinate.F90:124-152, 276-310) in the host mirror: the hybrid levels
    bit for bit against the pk, bk the reference built for the same namelist."""
    import numpy as np
    from isca_amd import atmosphere as atm
    from isca_amd.dyncore import IscaError
    g = np.load(os.path.join(golden_dir, "run_T21L12_hybrid_option.npz"))
    pk, bk = atm.named_vert_coord("hybrid", 12, 5.0, 0.3, 3.0, 0.15, 0.45, 1.0e5)
    assert np.array_equal(pk, g["tab_pk"]) and np.array_equal(bk, g["tab_bk"]) and pk[0] > 0.0 and bk[-1] == 1.0
    c = atm.config_from_namelist({"spectral_dynamics_nml": dict(num_levels=12, vert_coord_option="hybrid", p_press=0.15, p_sigma=0.45, scale_heights=5.0,
                                                                exponent=3.0, surf_res=0.3, reference_sea_level_press=1.0e5)})
    assert c.vert_coord_input == 1 and [c.pk_input[k] for k in range(13)] == list(g["tab_pk"])
    assert atm.config_from_namelist({"spectral_dynamics_nml": dict(num_levels=18, vert_coord_option="v197")}).bk_input[9] == 0.5
    assert atm.config_from_namelist({"spectral_dynamics_nml": dict(num_levels=14, vert_coord_option="mcm")}).bk_input[1] == 0.03
    with pytest.raises(IscaError, match="It must be 18"):
        atm.config_from_namelist({"spectral_dynamics_nml": dict(num_levels=20, vert_coord_option="v197")})
    with pytest.raises(IscaError, match="p_sigma must be greater than p_press"):
        atm.named_vert_coord("hybrid", 12, 5.0, 0.3, 3.0, 0.5, 0.4, 1.0e5)


def test_model_time_calendars(tmp_path):
    """RESTART/atmos_model.res (atmos_model.F90:198-202, 397-406): the date a segment ends at in each of time_manager's calendars, and
    the calendar type a restart file carries overriding main_nml."""
    from isca_amd import atmosphere as atm
    day = 86400
    assert atm._date_after([0, 0, 3, 0, 0, 0], "no_calendar", 2 * day + 3661) == [0, 0, 5, 1, 1, 1]
    assert atm._date_after([1, 1, 1, 0, 0, 0], "thirty_day", 45 * day) == [1, 2, 16, 0, 0, 0]
    assert atm._date_after([1, 12, 30, 0, 0, 0], "thirty_day", day) == [2, 1, 1, 0, 0, 0]
    assert atm._date_after([2001, 12, 31, 23, 0, 0], "noleap", 7200) == [2002, 1, 1, 1, 0, 0]
    assert atm._date_after([2000, 2, 28, 0, 0, 0], "noleap", 2 * day) == [2000, 3, 2, 0, 0, 0]
    assert atm._date_after([1900, 2, 28, 0, 0, 0], "julian", 2 * day) == [1900, 3, 1, 0, 0, 0]          # every fourth year
    assert atm._date_after([1900, 2, 28, 0, 0, 0], "gregorian", 2 * day) == [1900, 3, 2, 0, 0, 0]       # not 1900
    assert atm._date_after([2000, 2, 28, 0, 0, 0], "gregorian", 2 * day) == [2000, 3, 1, 0, 0, 0]       # but 2000
    assert atm._date_after([2003, 1, 1, 0, 0, 0], "julian", 366 * day + 365 * day) == [2005, 1, 1, 0, 0, 0]   # 2004 is a leap year
    # a restart file's calendar type wins over the namelist's
    (tmp_path / "atmos_model.res").write_text("  2004     2    28     0     0     0        Current model time\n     2        (Calendar: ...)\n")
    atm._clock = {"calendar": "thirty_day", "date0": [0] * 6, "step0": 0}
    try:
        atm._read_model_time(str(tmp_path))
        assert atm._clock["calendar"] == "julian" and atm._clock["date0"] == [2004, 2, 28, 0, 0, 0]
        assert atm._date_after(atm._clock["date0"], atm._clock["calendar"], 2 * day) == [2004, 3, 1, 0, 0, 0]
    finally:
        atm._clock = None


def test_restart_file_layer_against_scipy(tmp_path):
    """The library's netCDF-classic writer and reader (csrc/restart_nc.cpp; no device needed): scipy reads the file it writes -- dimensions,
    fms_io's axis variables with cartesian_axis, two records -- and it reads files scipy writes (64-bit offset and classic, doubles and floats,
    the single-record-variable layout), failing with read_data's wording on a missing variable."""
    import ctypes as C
    import numpy as np
    from scipy.io import netcdf_file
    from isca_amd import dyncore
    lib = dyncore.load_library()
    sums = (C.c_double * 3)()
    out = str(tmp_path / "a.nc").encode()
    assert lib.isca_restart_file_selftest(out, out, b"two_level", 1, sums) == 0 and list(sums) == [121785.0, 2000.25, 2059.25]
    f = netcdf_file(out.decode(), "r", mmap=False)
    assert f.dimensions["Time"] is None and f.dimensions["zaxis_1"] == 3 and f.dimensions["yaxis_1"] == 4 and f.dimensions["xaxis_1"] == 5
    v = f.variables["two_level"]
    assert v.dimensions == ("Time", "zaxis_1", "yaxis_1", "xaxis_1") and v.shape == (2, 3, 4, 5)
    assert np.array_equal(v[:].reshape(2, 60), 1000.0 * np.arange(1, 3)[:, None] + np.arange(60)[None, :] + 0.25)
    assert np.array_equal(f.variables["one_level"][0].ravel(), -1.0 / np.arange(1, 21)) and f.variables["one_level"].dimensions[1] == "zaxis_2"
    assert np.array_equal(f.variables["scalar"][:].ravel(), [1.0, 2.0]) and np.array_equal(f.variables["Time"][:], [1.0, 2.0])
    assert np.array_equal(f.variables["yaxis_1"][:], [1, 2, 3, 4]) and f.variables["yaxis_1"].cartesian_axis == b"Y" and f.variables["Time"].cartesian_axis == b"T"
    f.close()
    for version in (1, 2):
        p = str(tmp_path / f"b{version}.nc")
        g = netcdf_file(p, "w", version=version)
        g.createDimension("Time", None); g.createDimension("xaxis_1", 7)
        x = g.createVariable("xaxis_1", "d", ("xaxis_1",)); x[:] = np.arange(7.0); x.cartesian_axis = "X"
        w = g.createVariable("probe", "d", ("Time", "xaxis_1")); w[0] = np.arange(7.0) * 1.5; w[1] = np.arange(7.0) * -2.25
        q = g.createVariable("probe32", "f", ("Time", "xaxis_1")); q[0] = np.arange(7.0); q[1] = np.arange(7.0) + 0.5
        g.close()
        for var, rec, want in ((b"probe", 0, [31.5, 0.0, 9.0]), (b"probe", 1, [-47.25, 0.0, -13.5]), (b"probe32", 1, [24.5, 0.5, 6.5]), (b"xaxis_1", 0, [21.0, 0.0, 6.0])):
            assert lib.isca_restart_file_selftest(None, p.encode(), var, rec, sums) == 0 and list(sums) == want, (version, var, rec, list(sums))
    p = str(tmp_path / "c.nc")
    g = netcdf_file(p, "w", version=1); g.createDimension("Time", None); g.createDimension("x", 3)
    w = g.createVariable("only", "d", ("Time", "x")); w[0] = [1, 2, 3]; w[1] = [4, 5, 6]; w[2] = [7, 8, 9]; g.close()
    assert lib.isca_restart_file_selftest(None, p.encode(), b"only", 2, sums) == 0 and list(sums) == [24.0, 7.0, 9.0]
    assert lib.isca_restart_file_selftest(None, p.encode(), b"nothere", 0, sums) == 1 and b"has no variable nothere" in lib.isca_last_error()
    open(p, "wb").write(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    assert lib.isca_restart_file_selftest(None, p.encode(), b"only", 0, sums) == 1 and b"netCDF-4" in lib.isca_last_error()


def test_env_rank_sources(monkeypatch):
    """isca_env_rank (the decomposition of a host without MPI of its own: the Fortran drop-in): ISCA_* first, then what torchrun, Open MPI, PMI and Slurm
    export; one rank when nothing is set; inconsistent values are an error."""
    import ctypes as C
    from isca_amd import dyncore
    lib = dyncore.load_library()
    names = ["ISCA_RANK", "ISCA_WORLD_SIZE", "ISCA_LOCAL_RANK", "RANK", "WORLD_SIZE", "LOCAL_RANK", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE",
             "OMPI_COMM_WORLD_LOCAL_RANK", "PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID", "SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID"]

    def ask(**env):
        for n in names:
            monkeypatch.delenv(n, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, str(v))
        r, w, l = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        rc = lib.isca_env_rank(C.byref(r), C.byref(w), C.byref(l))
        return rc, r.value, w.value, l.value

    assert ask() == (0, 0, 1, 0)
    assert ask(ISCA_RANK=3, ISCA_WORLD_SIZE=8, ISCA_LOCAL_RANK=0) == (0, 3, 8, 0)
    assert ask(RANK=5, WORLD_SIZE=8, LOCAL_RANK=5) == (0, 5, 8, 5)                       # torchrun
    assert ask(OMPI_COMM_WORLD_RANK=2, OMPI_COMM_WORLD_SIZE=4, OMPI_COMM_WORLD_LOCAL_RANK=2) == (0, 2, 4, 2)
    assert ask(PMI_RANK=1, PMI_SIZE=2) == (0, 1, 2, 1)                                   # no local rank given: the rank
    assert ask(SLURM_PROCID=6, SLURM_NTASKS=16, SLURM_LOCALID=6) == (0, 6, 16, 6)
    assert ask(ISCA_RANK=1, ISCA_WORLD_SIZE=2, RANK=7, WORLD_SIZE=8)[:3] == (0, 1, 2)    # ISCA_* wins
    rc = ask(ISCA_RANK=4, ISCA_WORLD_SIZE=4)[0]
    assert rc == 1 and b"inconsistent rank" in lib.isca_last_error()


def test_restart_file_reader_property(tmp_path):
    """The reader of csrc/restart_nc.cpp on files scipy writes with random dimensions, variable orders, types and record counts (hypothesis): every record of
    every variable comes back as written -- the header walk (names and attributes with their padding, 32- / 64-bit offsets) and the record layout
    (slabs of every record variable in definition order, 4-byte padding of short types, the single-record-variable case)."""
    import ctypes as C
    import numpy as np
    from hypothesis import given, settings, strategies as st, HealthCheck
    from scipy.io import netcdf_file
    from isca_amd import dyncore
    lib = dyncore.load_library()
    sums = (C.c_double * 3)()
    counter = [0]

    var = st.tuples(st.sampled_from("dfih"), st.booleans(), st.integers(1, 3), st.integers(1, 5), st.text("abcxyz_", min_size=1, max_size=9))

    @settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(version=st.sampled_from([1, 2]), nrec=st.integers(1, 4), variables=st.lists(var, min_size=1, max_size=5), seed=st.integers(0, 2 ** 31))
    def check(version, nrec, variables, seed):
        rng = np.random.default_rng(seed)
        counter[0] += 1
        path = str(tmp_path / f"p{counter[0]}.nc")
        f = netcdf_file(path, "w", version=version)
        f.createDimension("Time", None)
        want, names = {}, set()
        for i, (typ, rec, ny, nx, nm) in enumerate(variables):
            name = f"{nm}{i}"
            if name in names:
                continue
            names.add(name)
            f.createDimension(f"y{i}", ny); f.createDimension(f"x{i}", nx)
            dims = (("Time",) if rec else ()) + (f"y{i}", f"x{i}")
            v = f.createVariable(name, typ, dims)
            v.long_name = "x" * (i + 1)                      # attributes of every padding length
            if typ in "df":
                data = rng.standard_normal(((nrec,) if rec else ()) + (ny, nx)).astype(np.float64 if typ == "d" else np.float32)
            else:
                data = rng.integers(-1000, 1000, ((nrec,) if rec else ()) + (ny, nx)).astype(np.int32 if typ == "i" else np.int16)
            v[:] = data
            want[name] = (rec, data)
        f.close()
        for name, (rec, data) in want.items():
            for r in range(nrec if rec else 1):
                blk = np.asarray(data[r] if rec else data, dtype=np.float64).ravel()
                assert lib.isca_restart_file_selftest(None, path.encode(), name.encode(), r, sums) == 0, lib.isca_last_error()
                assert np.isclose(sums[0], blk.sum(), rtol=1e-12, atol=1e-9) and sums[1] == blk[0] and sums[2] == blk[-1], (name, r, list(sums), blk)
    check()


def test_hs_forcing_nml_no_forcing_and_inert_keys():
    """hs_forcing_nml in the Python mirror: no_forcing = .true. (hs_forcing.F90:174) becomes zero coefficients and no tracer source for any entry;
    values that belong to branches which are off are accepted, the branches themselves are refused by name."""
    from isca_amd import atmosphere as atm, configs
    from isca_amd.dyncore import IscaError
    nml = configs.held_suarez()
    nml["hs_forcing_nml"].update(no_forcing=True, local_heating_option="", local_heating_srfamp=3.0, relax_to_specified_wind=False, p_trop=2.e4)
    c = atm.config_from_namelist(nml, "T21", tracer_sms=[1], tracer_flux=[3.e-5], tracer_sink=[-2.0])
    assert (c.ka, c.ks, c.kf, c.trflux, c.trsink) == (0.0, 0.0, 0.0, 0.0, 0.0) and list(c.tracer_sms) == [0] * len(c.tracer_sms)
    nml["hs_forcing_nml"]["no_forcing"] = False
    c = atm.config_from_namelist(nml, "T21")
    assert (c.ka, c.kf, c.trflux) == (-40.0, -1.0, 1.e-5)
    iso = configs.held_suarez(); iso["hs_forcing_nml"].update(local_heating_option="Isidoro", local_heating_srfamp=5.0, local_heating_xcenter=120.0)
    c = atm.config_from_namelist(iso, "T21")
    assert (c.local_heating_option, c.local_heating_srfamp, c.local_heating_xcenter, c.local_heating_ycenter) == (1, 5.0, 120.0, 45.0)
    iso["hs_forcing_nml"]["no_forcing"] = True
    assert atm.config_from_namelist(iso, "T21").local_heating_option == 0
    for key, val in (("local_heating_option", "from_file"), ("relax_to_specified_wind", True)):
        bad = configs.held_suarez(); bad["hs_forcing_nml"][key] = val
        with pytest.raises(IscaError, match=key):
            atm.config_from_namelist(bad, "T21")
    bad = configs.held_suarez(); bad["hs_forcing_nml"]["equilibrium_t_option"] = "top_down"
    with pytest.raises(IscaError, match="equilibrium_t_option"):
        atm.config_from_namelist(bad, "T21")
