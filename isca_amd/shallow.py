"""Shallow-water sibling core: ctypes binding of include/isca_shallow.h and a host mirror of the reference's
`atmosphere_mod` for src/atmos_spectral_shallow (atmosphere.F90:117-250): namelist groups `shallow_dynamics_nml`,
`shallow_physics_nml`, `main_nml` as in exp/test_cases/shallow_water/*.py.  No CPU fallback: the library and a HIP device
are required.  Arrays: grid [lat, lon], spectral complex [n, m] (views of the reference's Fortran (lon,lat) / (m,n))."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .dyncore import IscaError, RESOLUTIONS, load_library


class _CStirringConfig(C.Structure):
    _fields_ = [("decay_time", C.c_double), ("amplitude", C.c_double), ("lat0", C.c_double), ("lon0", C.c_double), ("widthy", C.c_double),
                ("widthx", C.c_double), ("B", C.c_double), ("do_localize", C.c_int), ("n_total_forcing_max", C.c_int),
                ("n_total_forcing_min", C.c_int), ("zonal_forcing_min", C.c_int), ("seed", C.c_ulonglong)]


def _apply_stirring(c, nml):
    """stirring_nml -> cfg.stirring (names unchanged; `seed` is ours); constants_nml radius / omega."""
    for k, v in (nml.get("constants_nml") or {}).items():
        if k.lower() not in ("radius", "omega"):
            raise IscaError(f"constants_nml: {k} cannot be changed (only radius and omega)")
        setattr(c, k.lower(), v)
    for k, v in (nml.get("stirring_nml") or {}).items():
        k = {"b": "B"}.get(k.lower(), k.lower())
        if not hasattr(c.stirring, k):
            raise IscaError(f"stirring_nml: unknown variable {k!r}")
        setattr(c.stirring, k, int(v) if isinstance(v, bool) else v)


class _CShallowConfig(C.Structure):
    _fields_ = [
        ("num_lon", C.c_int), ("num_lat", C.c_int), ("num_fourier", C.c_int), ("num_spherical", C.c_int), ("dt_atmos", C.c_double),
        ("damping_order", C.c_int), ("damping_coeff", C.c_double), ("robert_coeff", C.c_double), ("robert_coeff_tracer", C.c_double),
        ("h_0", C.c_double), ("u_deep_mag", C.c_double), ("n_merid_deep_flow", C.c_double), ("u_upper_mag_init", C.c_double),
        ("spec_tracer", C.c_int), ("grid_tracer", C.c_int),
        ("lon_centre_init_cyc", C.c_double), ("lat_centre_init_cyc", C.c_double), ("lon_centre_init_acyc", C.c_double),
        ("lat_centre_init_acyc", C.c_double), ("init_vortex_radius_deg", C.c_double), ("init_vortex_vor_f", C.c_double),
        ("init_vortex_h_h_0", C.c_double), ("add_initial_vortex_pair", C.c_int), ("add_initial_vortex_as_height", C.c_int),
        ("valid_range_v", C.c_double * 2),
        ("fric_damp_time", C.c_double), ("therm_damp_time", C.c_double), ("phys_h_0", C.c_double), ("h_amp", C.c_double),
        ("h_lon", C.c_double), ("h_lat", C.c_double), ("h_width", C.c_double), ("h_itcz", C.c_double), ("itcz_width", C.c_double),
        ("device", C.c_int), ("stirring", _CStirringConfig), ("radius", C.c_double), ("omega", C.c_double),
    ]


EXPORTED_SYMBOLS = ["isca_shallow_config_default", "isca_shallow_create", "isca_shallow_destroy", "isca_shallow_cold_start",
                    "isca_shallow_step", "isca_shallow_get_state", "isca_shallow_set_state", "isca_shallow_get_info",
                    "isca_shallow_set_time_pointers", "isca_shallow_set_stirring_noise", "isca_shallow_init_from_grid"]
_UNSUPPORTED = {"fourier_inc": 1, "triang_trunc": True, "south_to_north": True, "damping_option": "resolution_dependent",
                "raw_filter_coeff": 1.0, "longitude_origin": 0.0}
_PHYS_RENAME = {"h_0": "phys_h_0"}


def _lib():
    lib = load_library()
    if not getattr(lib, "_shallow_bound", False):
        H, dp = C.c_void_p, C.POINTER(C.c_double)
        sig = {"isca_shallow_config_default": [C.POINTER(_CShallowConfig)], "isca_shallow_create": [C.POINTER(_CShallowConfig), C.POINTER(H)],
               "isca_shallow_destroy": [H], "isca_shallow_cold_start": [H], "isca_shallow_step": [H, C.c_int],
               "isca_shallow_get_state": [H, C.c_char_p, C.c_int, dp, C.c_size_t],
               "isca_shallow_set_state": [H, C.c_char_p, C.c_int, dp, C.c_size_t],
               "isca_shallow_get_info": [H, C.c_char_p, C.POINTER(C.c_long)],
               "isca_shallow_set_time_pointers": [H, C.c_int, C.c_int, C.c_long],
               "isca_shallow_set_stirring_noise": [H, dp, C.c_size_t], "isca_shallow_init_from_grid": [H, dp, dp, dp]}
        for name, args in sig.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, C.c_int
        lib._shallow_bound = True
    return lib


def config_from_namelist(namelist: dict | None = None, resolution: str | None = None, **overrides) -> _CShallowConfig:
    """shallow_dynamics_nml + shallow_physics_nml + main_nml(dt_atmos) -> C config; option values the device core does not
    implement are refused; stirring_nml switches the stochastic vorticity forcing on."""
    c = _CShallowConfig()
    _lib().isca_shallow_config_default(C.byref(c))
    kw: dict = {}
    if resolution is not None:
        r = RESOLUTIONS[resolution]
        kw.update(num_lon=r["lon_max"], num_lat=r["lat_max"], num_fourier=r["num_fourier"], num_spherical=r["num_spherical"])
    nml = {g.lower(): v for g, v in (namelist or {}).items()}
    _apply_stirring(c, nml)
    for k, v in nml.get("shallow_dynamics_nml", {}).items():
        k = k.lower()
        if k in _UNSUPPORTED:
            if (str(v).lower() != str(_UNSUPPORTED[k]).lower()) and v != _UNSUPPORTED[k]:
                raise IscaError(f'"{v}" is not a supported value for {k} (only "{_UNSUPPORTED[k]}")')
            continue
        if k in ("check_fourier_imag", "cutoff_wn", "init_cond_file", "input_file_div_name", "input_file_height_name", "input_file_vor_name",
                 "initial_condition_from_input_file"):       # the host reads the file and calls init_from_grid
            continue
        kw[k] = v
    for k, v in nml.get("shallow_physics_nml", {}).items():
        k = k.lower()
        if k == "del_h":
            continue
        kw[_PHYS_RENAME.get(k, k)] = v
    if "dt_atmos" in nml.get("main_nml", {}):
        kw["dt_atmos"] = nml["main_nml"]["dt_atmos"]
    kw.update(overrides)
    for k, v in kw.items():
        if k == "valid_range_v":
            c.valid_range_v[0], c.valid_range_v[1] = v
        elif not hasattr(c, k):
            raise IscaError(f"unknown shallow-water configuration key {k!r}")
        else:
            setattr(c, k, int(v) if isinstance(v, bool) else v)
    return c


class ShallowWater:
    GRID = ("u", "v", "vor", "div", "h", "tr", "trs", "stream", "pv", "h_eq", "deep_geopot")
    SPEC = ("vors", "divs", "hs", "trss", "stirs")
    _noise_fn = "isca_shallow_set_stirring_noise"

    def __init__(self, cfg: _CShallowConfig):
        self.lib = _lib()
        self.cfg = cfg
        self._h = C.c_void_p()
        self._check(self.lib.isca_shallow_create(C.byref(cfg), C.byref(self._h)))
        self.I, self.J, self.M1, self.N1 = cfg.num_lon, cfg.num_lat, cfg.num_fourier + 1, cfg.num_spherical + 1

    def _check(self, rc):
        if rc != 0:
            raise IscaError(self.lib.isca_last_error().decode())

    def close(self):
        if self._h:
            self.lib.isca_shallow_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cold_start(self):
        self._check(self.lib.isca_shallow_cold_start(self._h))

    def init_from_grid(self, vor, div, height):
        """initial_condition_from_input_file: grid vorticity, divergence and height anomaly (h_0 is added), e.g. the variables of
        init_cond_file read with scipy.io.netcdf_file on the model grid."""
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (vor, div, height)]
        if any(x.shape != (self.J, self.I) for x in a):
            raise IscaError("init_from_grid: fields must have the grid shape (lat, lon)")
        self._check(self.lib.isca_shallow_init_from_grid(self._h, *[x.ctypes.data_as(C.POINTER(C.c_double)) for x in a]))

    def step(self, nsteps: int = 1):
        self._check(self.lib.isca_shallow_step(self._h, int(nsteps)))

    def info(self, name: str) -> int:
        v = C.c_long()
        self._check(self.lib.isca_shallow_get_info(self._h, name.encode(), C.byref(v)))
        return v.value

    def get(self, name: str, time_level: int = 1):
        spec = name in self.SPEC
        a = np.zeros((self.N1, self.M1), dtype=np.complex128) if spec else np.zeros((self.J, self.I))
        v = a.view(np.float64)
        self._check(self.lib.isca_shallow_get_state(self._h, name.encode(), time_level, v.ctypes.data_as(C.POINTER(C.c_double)), v.size))
        return a

    def set(self, name: str, value, time_level: int = 1):
        spec = name in self.SPEC
        a = np.ascontiguousarray(value, dtype=np.complex128 if spec else np.float64)
        if a.shape != ((self.N1, self.M1) if spec else (self.J, self.I)):
            raise IscaError(f"set({name}): wrong shape {a.shape}")
        v = a.view(np.float64)
        self._check(self.lib.isca_shallow_set_state(self._h, name.encode(), time_level, v.ctypes.data_as(C.POINTER(C.c_double)), v.size))

    def set_time_pointers(self, previous: int, current: int, step_count: int = 0):
        self._check(self.lib.isca_shallow_set_time_pointers(self._h, previous, current, step_count))

    def set_stirring_noise(self, ran):
        """Uniform [0,1) numbers of shape (2, n, m) (Fortran ran_nmbrs(0:num_fourier, 0:num_spherical, 2)) for the next step's stirring."""
        a = np.ascontiguousarray(ran, dtype=np.float64)
        if a.shape != (2, self.N1, self.M1):
            raise IscaError("set_stirring_noise: shape must be (2, num_spherical+1, num_fourier+1)")
        self._check(getattr(self.lib, self._noise_fn)(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), a.size))


# ---- module-level mirror of atmosphere_mod (shallow): one instance, like the Fortran module
_model: ShallowWater | None = None


def atmosphere_init(namelist=None, resolution=None, **overrides):
    global _model
    if _model is None:
        _model = ShallowWater(config_from_namelist(namelist, resolution, **overrides))
        _model.cold_start()
    return _model


def atmosphere(nsteps: int = 1):
    if _model is None:
        raise IscaError("atmosphere: atmosphere_init has not been called")
    _model.step(nsteps)


def atmosphere_end():
    global _model
    if _model is None:
        raise IscaError("atmosphere_end: atmosphere_init has not been called.")
    _model.close()
    _model = None


# =====================================================================================================
# Barotropic-vorticity sibling core (include/isca_barotropic.h; reference src/atmos_spectral_barotropic)
# =====================================================================================================
class _CBarotropicConfig(C.Structure):
    _fields_ = [
        ("num_lon", C.c_int), ("num_lat", C.c_int), ("num_fourier", C.c_int), ("num_spherical", C.c_int), ("dt_atmos", C.c_double),
        ("damping_order", C.c_int), ("damping_coeff", C.c_double), ("damping_coeff_r", C.c_double), ("robert_coeff", C.c_double),
        ("zeta_0", C.c_double), ("m_0", C.c_int), ("eddy_width", C.c_double), ("eddy_lat", C.c_double),
        ("spec_tracer", C.c_int), ("grid_tracer", C.c_int), ("valid_range_v", C.c_double * 2), ("initial_zonal_wind", C.c_int),
        ("device", C.c_int), ("stirring", _CStirringConfig), ("radius", C.c_double), ("omega", C.c_double),
    ]


BAROTROPIC_SYMBOLS = ["isca_barotropic_config_default", "isca_barotropic_create", "isca_barotropic_destroy", "isca_barotropic_cold_start",
                      "isca_barotropic_step", "isca_barotropic_get_state", "isca_barotropic_set_state", "isca_barotropic_get_info",
                      "isca_barotropic_set_time_pointers", "isca_barotropic_set_stirring_noise"]


def _blib():
    lib = load_library()
    if not getattr(lib, "_barotropic_bound", False):
        H, dp = C.c_void_p, C.POINTER(C.c_double)
        sig = {"isca_barotropic_config_default": [C.POINTER(_CBarotropicConfig)],
               "isca_barotropic_create": [C.POINTER(_CBarotropicConfig), C.POINTER(H)], "isca_barotropic_destroy": [H],
               "isca_barotropic_cold_start": [H], "isca_barotropic_step": [H, C.c_int],
               "isca_barotropic_get_state": [H, C.c_char_p, C.c_int, dp, C.c_size_t],
               "isca_barotropic_set_state": [H, C.c_char_p, C.c_int, dp, C.c_size_t],
               "isca_barotropic_get_info": [H, C.c_char_p, C.POINTER(C.c_long)],
               "isca_barotropic_set_time_pointers": [H, C.c_int, C.c_int, C.c_long],
               "isca_barotropic_set_stirring_noise": [H, dp, C.c_size_t]}
        for name, args in sig.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, C.c_int
        lib._barotropic_bound = True
    return lib


def barotropic_config_from_namelist(namelist: dict | None = None, resolution: str | None = None, **overrides) -> _CBarotropicConfig:
    """barotropic_dynamics_nml + main_nml(dt_atmos) (exp/test_cases/barotropic_vorticity_equation) -> C config."""
    c = _CBarotropicConfig()
    _blib().isca_barotropic_config_default(C.byref(c))
    kw: dict = {}
    if resolution is not None:
        r = RESOLUTIONS[resolution]
        kw.update(num_lon=r["lon_max"], num_lat=r["lat_max"], num_fourier=r["num_fourier"], num_spherical=r["num_spherical"])
    nml = {g.lower(): v for g, v in (namelist or {}).items()}
    _apply_stirring(c, nml)
    for k, v in nml.get("barotropic_dynamics_nml", {}).items():
        k = k.lower()
        if k in _UNSUPPORTED:
            if (str(v).lower() != str(_UNSUPPORTED[k]).lower()) and v != _UNSUPPORTED[k]:
                raise IscaError(f'"{v}" is not a supported value for {k} (only "{_UNSUPPORTED[k]}")')
            continue
        if k in ("check_fourier_imag", "cutoff_wn"):
            continue
        if k == "initial_zonal_wind":
            if str(v) not in ("zero", "two_jets"):
                raise IscaError(f"barotropic_dynamics_init: {v} is not a valid value of initial_zonal_wind ")
            v = 1 if str(v) == "two_jets" else 0
        kw[k] = v
    if "dt_atmos" in nml.get("main_nml", {}):
        kw["dt_atmos"] = nml["main_nml"]["dt_atmos"]
    kw.update(overrides)
    for k, v in kw.items():
        if k == "valid_range_v":
            c.valid_range_v[0], c.valid_range_v[1] = v
        elif not hasattr(c, k):
            raise IscaError(f"unknown barotropic configuration key {k!r}")
        else:
            setattr(c, k, int(v) if isinstance(v, bool) else v)
    return c


class Barotropic(ShallowWater):
    GRID = ("u", "v", "vor", "tr", "trs", "stream", "pv")
    SPEC = ("vors", "trss", "stirs")
    _noise_fn = "isca_barotropic_set_stirring_noise"

    def __init__(self, cfg: _CBarotropicConfig):
        self.lib = _blib()
        self.cfg = cfg
        self._h = C.c_void_p()
        self._check(self.lib.isca_barotropic_create(C.byref(cfg), C.byref(self._h)))
        self.I, self.J, self.M1, self.N1 = cfg.num_lon, cfg.num_lat, cfg.num_fourier + 1, cfg.num_spherical + 1

    def close(self):
        if self._h:
            self.lib.isca_barotropic_destroy(self._h)
            self._h = C.c_void_p()

    def cold_start(self):
        self._check(self.lib.isca_barotropic_cold_start(self._h))

    def step(self, nsteps: int = 1):
        self._check(self.lib.isca_barotropic_step(self._h, int(nsteps)))

    def info(self, name: str) -> int:
        v = C.c_long()
        self._check(self.lib.isca_barotropic_get_info(self._h, name.encode(), C.byref(v)))
        return v.value

    def get(self, name: str, time_level: int = 1):
        if name == "zonal_u_init":
            a = np.zeros(self.J)
        else:
            a = np.zeros((self.N1, self.M1), dtype=np.complex128) if name in self.SPEC else np.zeros((self.J, self.I))
        v = a.view(np.float64)
        self._check(self.lib.isca_barotropic_get_state(self._h, name.encode(), time_level, v.ctypes.data_as(C.POINTER(C.c_double)), v.size))
        return a

    def set(self, name: str, value, time_level: int = 1):
        spec = name in self.SPEC
        a = np.ascontiguousarray(value, dtype=np.complex128 if spec else np.float64)
        v = a.view(np.float64)
        self._check(self.lib.isca_barotropic_set_state(self._h, name.encode(), time_level, v.ctypes.data_as(C.POINTER(C.c_double)), v.size))

    def set_time_pointers(self, previous: int, current: int, step_count: int = 0):
        self._check(self.lib.isca_barotropic_set_time_pointers(self._h, previous, current, step_count))


# =====================================================================================================
# Restart files of the sibling cores, with the reference's variable set (shallow_dynamics.F90:650-678, barotropic_dynamics.F90
# write_restart; stirring.F90:236-239): two records along Time, record 1 = `previous`, record 2 = `current`
# =====================================================================================================
def _restart_layout(model):
    if isinstance(model, Barotropic):
        return "barotropic_dynamics.res.nc", ("vors",), ("u", "v", "vor")
    return "shallow_dynamics.res.nc", ("vors", "divs", "hs"), ("u", "v", "vor", "div", "h")


def write_restart(model, directory: str):
    """shallow_dynamics_end / barotropic_dynamics_end (+ stirring_end): RESTART/<core>_dynamics.res.nc, RESTART/stirring.res.nc."""
    import os
    from .restart import _Writer
    os.makedirs(directory, exist_ok=True)
    fname, spec, grid = _restart_layout(model)
    w = _Writer(os.path.join(directory, fname))
    for nm in spec + (("trss",) if model.cfg.spec_tracer else ()):
        lv = [model.get(nm, 0), model.get(nm, 1)]
        out = "trs" if nm == "trss" else nm
        w.put(out + "_real", [np.ascontiguousarray(a.real) for a in lv])
        w.put(out + "_imag", [np.ascontiguousarray(a.imag) for a in lv])
    for nm in grid + (("trs",) if model.cfg.spec_tracer else ()) + (("tr",) if model.cfg.grid_tracer else ()):
        w.put(nm, [model.get(nm, 0), model.get(nm, 1)])
    w.close()
    if model.cfg.stirring.amplitude != 0.0:
        s = model.get("stirs")
        w = _Writer(os.path.join(directory, "stirring.res.nc"))
        w.put("stir_real", [np.ascontiguousarray(s.real)])
        w.put("stir_imag", [np.ascontiguousarray(s.imag)])
        w.close()


def read_restart(model, directory: str):
    """The Time /= Time_init branch of *_dynamics_init + atmosphere_init (previous = 1, current = 2) + stirring_init's restart."""
    import os
    from .restart import _read_all
    fname, spec, grid = _restart_layout(model)
    path = os.path.join(directory, fname)
    if not os.path.exists(path):
        raise IscaError("read_restart: restart does not exist")
    d = _read_all(path)
    J, I, N1, M1 = model.J, model.I, model.N1, model.M1
    if d["u"].shape[-2:] != (J, I) or d["vors_real"].shape[-2:] != (N1, M1):
        raise IscaError("read_restart: resolution of the restart file does not match the namelist")
    for rec, slot in ((0, 0), (1, 1)):
        model.set_time_pointers(slot, slot, 0)                      # address storage slot `slot` as "current" to fill it
        for nm in spec + (("trss",) if model.cfg.spec_tracer else ()):
            key = "trs" if nm == "trss" else nm
            model.set(nm, d[key + "_real"][rec].reshape(N1, M1) + 1j * d[key + "_imag"][rec].reshape(N1, M1), 1)
        for nm in grid + (("trs",) if model.cfg.spec_tracer else ()) + (("tr",) if model.cfg.grid_tracer else ()):
            model.set(nm, d[nm][rec].reshape(J, I), 1)
    model.set_time_pointers(0, 1, 1)
    spath = os.path.join(directory, "stirring.res.nc")
    if model.cfg.stirring.amplitude != 0.0 and os.path.exists(spath):
        s = _read_all(spath)
        model.set("stirs", s["stir_real"].reshape(N1, M1) + 1j * s["stir_imag"].reshape(N1, M1))
