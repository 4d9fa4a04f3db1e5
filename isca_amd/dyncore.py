"""ctypes binding of the C-ABI in include/isca_dyn.h (isca_amd/lib/libisca_dyn.so).

This is the host side of the MI355X spectral core.  It has no CPU fallback: if the shared library
is missing or no HIP device is usable, construction raises.

Array conventions at this level are the library's (= the reference's Fortran layouts seen from C):
grid fields are numpy arrays [lev, lat, lon], spectral fields complex [lev, n, m].
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field, fields as dc_fields

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ISCA_DYN_LIB") or os.path.join(_HERE, "lib", "libisca_dyn.so")   # override: kernel experiments


class IscaError(RuntimeError):
    """Raised where the reference calls error_mesg(..., FATAL)."""


MAX_LEVELS = 128          # ISCA_MAX_LEVELS
MAX_TRACERS = 8           # ISCA_MAX_TRACERS


class _CMoistConfig(C.Structure):
    _fields_ = [
        ("roughness_mom", C.c_double), ("roughness_heat", C.c_double), ("roughness_moist", C.c_double),
        ("solar_constant", C.c_double), ("del_sol", C.c_double), ("del_sw", C.c_double), ("ir_tau_eq", C.c_double),
        ("ir_tau_pole", C.c_double), ("atm_abs", C.c_double), ("odp", C.c_double), ("sw_diff", C.c_double), ("linear_tau", C.c_double),
        ("wv_exponent", C.c_double), ("solar_exponent", C.c_double),
        ("depth", C.c_double), ("tconst", C.c_double), ("delta_T", C.c_double), ("albedo_value", C.c_double), ("evaporation", C.c_int),
        ("tau_bm", C.c_double), ("rhbm", C.c_double), ("Tmin", C.c_double), ("Tmax", C.c_double), ("val_inc", C.c_double),
        ("do_rayleigh", C.c_int), ("trayfric", C.c_double), ("sponge_pbottom", C.c_double), ("damping_conserve_energy", C.c_int),
        ("constant_gust", C.c_double), ("frac_inner", C.c_double), ("rich_crit_pbl", C.c_double), ("rich_crit", C.c_double),
        ("drag_min", C.c_double),
    ]


class _CConfig(C.Structure):
    _fields_ = [
        ("lon_max", C.c_int), ("lat_max", C.c_int), ("num_fourier", C.c_int), ("num_spherical", C.c_int),
        ("num_levels", C.c_int), ("fourier_inc", C.c_int), ("triang_trunc", C.c_int), ("dt_atmos", C.c_double),
        ("damping_order", C.c_int), ("damping_coeff", C.c_double),
        ("eddy_sponge_coeff", C.c_double), ("zmu_sponge_coeff", C.c_double), ("zmv_sponge_coeff", C.c_double),
        ("robert_coeff", C.c_double), ("raw_filter_coeff", C.c_double), ("alpha_implicit", C.c_double),
        ("reference_sea_level_press", C.c_double),
        ("scale_heights", C.c_double), ("exponent", C.c_double), ("surf_res", C.c_double),
        ("do_mass_correction", C.c_int), ("do_energy_correction", C.c_int), ("do_water_correction", C.c_int),
        ("water_correction_limit", C.c_double), ("initial_temperature", C.c_double), ("initial_sphum", C.c_double),
        ("valid_range_t", C.c_double * 2), ("num_tracers", C.c_int),
        ("t_zero", C.c_double), ("t_strat", C.c_double), ("delh", C.c_double), ("delv", C.c_double),
        ("eps", C.c_double), ("sigma_b", C.c_double), ("ka", C.c_double), ("ks", C.c_double), ("kf", C.c_double),
        ("do_conserve_energy", C.c_int), ("trflux", C.c_double), ("trsink", C.c_double), ("P00", C.c_double),
        ("rank", C.c_int), ("world_size", C.c_int), ("device", C.c_int), ("stream", C.c_void_p),
        ("legendre_impl", C.c_int),
        ("physics", C.c_int), ("vert_coord_input", C.c_int),
        ("pk_input", C.c_double * (MAX_LEVELS + 1)), ("bk_input", C.c_double * (MAX_LEVELS + 1)),
        ("moist", _CMoistConfig), ("radius", C.c_double), ("omega", C.c_double),
        ("damping_option", C.c_int), ("cutoff_wn", C.c_int), ("damping_coeff_vor", C.c_double), ("damping_coeff_div", C.c_double),
        ("damping_order_vor", C.c_int), ("damping_order_div", C.c_int),
        ("tracer_spectral", C.c_int * MAX_TRACERS), ("tracer_robert_coeff", C.c_double * MAX_TRACERS),
        ("use_virtual_temperature", C.c_int),
        ("vert_advect_uv", C.c_int), ("vert_advect_t", C.c_int), ("use_implicit", C.c_int), ("make_symmetric", C.c_int),
        ("vert_difference_option", C.c_int), ("tracer_hole_filling", C.c_int * MAX_TRACERS),
        ("tracer_sms", C.c_int * MAX_TRACERS), ("tracer_flux", C.c_double * MAX_TRACERS), ("tracer_sink", C.c_double * MAX_TRACERS),
        ("tracer_advect_vert", C.c_int * MAX_TRACERS),
        ("local_heating_option", C.c_int), ("local_heating_srfamp", C.c_double), ("local_heating_xwidth", C.c_double),
        ("local_heating_ywidth", C.c_double), ("local_heating_xcenter", C.c_double), ("local_heating_ycenter", C.c_double),
        ("local_heating_vert_decay", C.c_double),
    ]


_lib = None


def load_library():
    """dlopen the in-tree HIP library; fails loudly (no fallback) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IscaError(f"{LIB_PATH} not found: build it with `python -m isca_amd.build` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    lib.isca_last_error.restype = C.c_char_p
    sizes = (C.c_size_t * 4)()
    lib.isca_config_sizes.argtypes, lib.isca_config_sizes.restype = [C.POINTER(C.c_size_t), C.c_int], C.c_int
    if lib.isca_config_sizes(sizes, 4) != 0 or (sizes[0], sizes[1]) != (C.sizeof(_CConfig), C.sizeof(_CMoistConfig)):
        raise IscaError(f"{LIB_PATH}: configuration structs of the library and of this binding differ (rebuild: python -m isca_amd.build)")
    dp = C.POINTER(C.c_double)
    H = C.c_void_p
    sig = {
        "isca_dyn_config_default": [C.POINTER(_CConfig)],
        "isca_dyn_create": [C.POINTER(_CConfig), C.POINTER(H)],
        "isca_dyn_destroy": [H],
        "isca_dyn_cold_start": [H],
        "isca_dyn_set_surf_geopotential": [H, dp, C.c_size_t],
        "isca_dyn_step": [H, C.c_int, C.c_int],
        "isca_dyn_synchronize": [H],
        "isca_dyn_dynamics": [H, dp, dp, dp, dp, C.c_int, C.c_int],
        "isca_dyn_set_tendencies": [H, dp, dp, dp, dp, C.c_int],
        "isca_dyn_delta_t": [H, dp],
        "isca_dyn_step_phase": [H, C.c_int],
        "isca_dyn_exchange_buffers": [H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
        "isca_dyn_reduce_buffer": [H, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
        "isca_dyn_halo_buffers": [H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
        "isca_wavenumber_dealing": [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "isca_dyn_get_state": [H, C.c_char_p, C.c_int, dp, C.c_size_t],
        "isca_dyn_set_state": [H, C.c_char_p, C.c_int, dp, C.c_size_t],
        "isca_dyn_complete_update": [H, C.c_int],
        "isca_dyn_set_time_pointers": [H, C.c_int, C.c_int, C.c_long],
        "isca_dyn_refresh_derived": [H],
        "isca_dyn_get_table": [H, C.c_char_p, dp, C.c_size_t],
        "isca_dyn_get_info": [H, C.c_char_p, C.POINTER(C.c_long)],
        "isca_dyn_set_info": [H, C.c_char_p, C.c_long],
        "isca_dyn_diag_open": [H, C.c_char_p, C.c_char_p, C.c_double],
        "isca_dyn_diag_close": [H],
        "isca_dyn_set_topography": [H, dp, dp, C.c_double, dp, dp],
        "isca_topog_regularize": [H, C.c_double, dp, dp, dp, dp],
        "isca_topog_compute_lambda": [H, C.c_double, dp, dp, dp, dp],
        "isca_nc_read_variable": [C.c_char_p, C.c_char_p, C.c_int, dp, C.c_size_t, C.POINTER(C.c_size_t)],
        "isca_env_rank": [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "isca_dyn_comm_init_env": [H],
        "isca_dyn_write_restart": [H, C.c_char_p, C.c_char_p],
        "isca_dyn_read_restart": [H, C.c_char_p, C.c_char_p],
        "isca_dyn_restart_exists": [C.c_char_p],
        "isca_restart_file_selftest": [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, dp],
        "isca_trans_spherical_to_grid": [H, dp, dp, C.c_int],
        "isca_trans_grid_to_spherical": [H, dp, dp, C.c_int, C.c_int],
        "isca_vor_div_from_uv_grid": [H, dp, dp, dp, dp, C.c_int],
        "isca_uv_grid_from_vor_div": [H, dp, dp, dp, dp, C.c_int],
        "isca_horizontal_advection": [H, dp, dp, dp, dp, C.c_int],
        "isca_trans_spherical_to_fourier": [H, dp, dp, C.c_int],
        "isca_trans_fourier_to_spherical": [H, dp, dp, C.c_int],
        "isca_trans_grid_to_fourier": [H, dp, dp, C.c_int],
        "isca_trans_fourier_to_grid": [H, dp, dp, C.c_int],
        "isca_area_weighted_global_mean": [H, dp, dp],
        "isca_hs_forcing": [H, C.c_double, dp, dp, dp, dp, dp, dp, dp, dp],
        "isca_trans_filter": [H, dp, dp, C.c_int],
        "isca_idealized_moist_phys": [H, C.c_int, C.c_double, C.c_double] + [dp] * 17,
        "isca_bench_transform_pair": [H, C.c_int, C.c_int, dp, dp],
        "isca_compute_laplacian": [H, dp, dp, C.c_int, C.c_int],
        "isca_compute_gradient_cos": [H, dp, dp, dp, C.c_int],
        "isca_compute_ucos_vcos": [H, dp, dp, dp, dp, C.c_int],
        "isca_compute_vor_div": [H, dp, dp, dp, dp, C.c_int],
        "isca_triangular_truncation": [H, dp, C.c_int],
        "isca_divide_by_cos": [H, dp, C.c_int, C.c_int],
        "isca_mass_weighted_global_integral": [H, dp, dp, dp],
        "isca_pressure_variables": [H, dp, dp, dp, dp, dp],
        "isca_compute_geopotential": [H, dp, dp, dp, dp, dp],
        "isca_compute_geopotential_surf": [H, dp, dp, dp, dp, dp, dp, dp],
        "isca_a_grid_horiz_advection": [H, dp, dp, dp, C.c_double, dp],
        "isca_vert_advection_ppm": [H, C.c_double, dp, dp, dp, dp],
        "isca_hs_tracer_source_sink": [H, dp, dp, dp],
        "isca_vert_advection_centered": [H, dp, dp, dp, dp],
        "isca_compute_pressures_and_heights": [H, dp, dp, dp, dp, dp, dp, dp],
        "isca_leapfrog_2level_a": [H, C.c_size_t, dp, dp, dp, dp, C.c_double, C.c_double, C.c_double, dp],
        "isca_leapfrog_2level_b": [H, C.c_size_t, dp, dp, dp, C.c_double, C.c_double],
        "isca_compute_gaussian": [C.c_int, dp, dp],
        "isca_compute_legendre": [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp],
        "isca_implicit_correction": [H, dp, dp, dp, dp, dp, dp, dp, dp, dp, C.c_double],
        "isca_compute_spectral_damping": [H, C.c_int, dp, dp, C.c_double],
        "isca_leapfrog": [H, dp, dp, dp, C.c_double, C.c_double],
        "isca_comm_get_unique_id": [C.c_char_p],
        "isca_dyn_comm_init": [H, C.c_char_p],
        "isca_comm_selftest": [C.c_int, dp],
        "isca_dyn_diag_select": [H, C.c_char_p],
        "isca_dyn_diag_read": [H, C.c_char_p, dp, C.c_size_t, C.POINTER(C.c_long), C.c_int],
        "isca_dyn_kernel_times": [H, C.c_int, dp, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.isca_dyn_comm_kind.argtypes, lib.isca_dyn_comm_kind.restype = [H], C.c_char_p
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "isca_last_error", "isca_dyn_config_default", "isca_dyn_create", "isca_dyn_destroy", "isca_dyn_cold_start", "isca_dyn_set_surf_geopotential",
    "isca_dyn_step", "isca_dyn_synchronize", "isca_dyn_dynamics", "isca_dyn_set_tendencies", "isca_dyn_delta_t", "isca_dyn_step_phase",
    "isca_dyn_exchange_buffers",
    "isca_dyn_reduce_buffer", "isca_dyn_halo_buffers", "isca_wavenumber_dealing", "isca_dyn_get_state", "isca_dyn_set_state", "isca_dyn_complete_update",
    "isca_dyn_set_time_pointers", "isca_dyn_refresh_derived",
    "isca_dyn_get_table", "isca_dyn_get_info", "isca_dyn_set_info", "isca_dyn_diag_open", "isca_dyn_diag_close", "isca_dyn_set_topography", "isca_topog_regularize", "isca_topog_compute_lambda", "isca_nc_read_variable", "isca_env_rank", "isca_dyn_comm_init_env", "isca_dyn_write_restart", "isca_dyn_read_restart", "isca_dyn_restart_exists", "isca_restart_file_selftest", "isca_trans_spherical_to_grid", "isca_trans_grid_to_spherical",
    "isca_vor_div_from_uv_grid", "isca_uv_grid_from_vor_div", "isca_horizontal_advection",
    "isca_trans_spherical_to_fourier", "isca_trans_fourier_to_spherical", "isca_trans_grid_to_fourier",
    "isca_trans_fourier_to_grid", "isca_area_weighted_global_mean", "isca_hs_forcing",
    "isca_bench_transform_pair", "isca_dyn_kernel_times",
    "isca_compute_laplacian", "isca_compute_gradient_cos", "isca_compute_ucos_vcos", "isca_compute_vor_div",
    "isca_triangular_truncation", "isca_divide_by_cos", "isca_mass_weighted_global_integral", "isca_pressure_variables",
    "isca_compute_geopotential", "isca_compute_geopotential_surf", "isca_a_grid_horiz_advection", "isca_vert_advection_ppm", "isca_hs_tracer_source_sink",
    "isca_implicit_correction", "isca_compute_spectral_damping", "isca_leapfrog",
    "isca_vert_advection_centered", "isca_compute_pressures_and_heights", "isca_leapfrog_2level_a", "isca_leapfrog_2level_b",
    "isca_compute_gaussian", "isca_compute_legendre",
    "isca_comm_get_unique_id", "isca_dyn_comm_init", "isca_comm_selftest", "isca_dyn_comm_check", "isca_dyn_comm_kind",
    "isca_dyn_diag_select", "isca_dyn_diag_read", "isca_idealized_moist_phys", "isca_trans_filter", "isca_config_sizes",
]

GRAV = 9.80          # shared/constants/constants.F90 (constants_nml default), as in csrc/tables.h

# RESOLUTIONS of the reference's Python harness (src/extra/python/isca/experiment.py:29-57)
RESOLUTIONS = {
    "T21": dict(lon_max=64, lat_max=32, num_fourier=21, num_spherical=22),
    "T42": dict(lon_max=128, lat_max=64, num_fourier=42, num_spherical=43),
    "T85": dict(lon_max=256, lat_max=128, num_fourier=85, num_spherical=86),
    "T170": dict(lon_max=512, lat_max=256, num_fourier=170, num_spherical=171),
    # lon_max with factors 3 and 5 (fft99 takes n/2 = 2^a 3^b 5^c; the mixed-radix FFT kernels)
    "T31": dict(lon_max=96, lat_max=48, num_fourier=31, num_spherical=32),
    "T53": dict(lon_max=160, lat_max=80, num_fourier=53, num_spherical=54),
    "T63": dict(lon_max=192, lat_max=96, num_fourier=63, num_spherical=64),
    "T127": dict(lon_max=384, lat_max=192, num_fourier=127, num_spherical=128),
    # small test resolutions (not in the reference's table)
    "T10": dict(lon_max=32, lat_max=16, num_fourier=10, num_spherical=11),
    "S10": dict(lon_max=32, lat_max=32, num_fourier=10, num_spherical=21, fourier_inc=2),         # zonal wavenumbers 0, 2, .., 20 on a 180-degree sector
    "R10": dict(lon_max=32, lat_max=32, num_fourier=10, num_spherical=11, triang_trunc=0),      # rhomboidal: 2 lat_max >= 5 (num_spherical - 1) + 1
}


def default_config(resolution: str | None = None, **overrides) -> _CConfig:
    lib = load_library()
    c = _CConfig()
    lib.isca_dyn_config_default(C.byref(c))
    if resolution is not None:
        overrides = {**RESOLUTIONS[resolution], **overrides}
    for k, v in overrides.items():
        if k == "valid_range_t":
            c.valid_range_t[0], c.valid_range_t[1] = v
        elif k in ("pk_input", "bk_input"):               # vert_coordinate_nml with vert_coord_option = 'input'
            if len(v) > MAX_LEVELS + 1:
                raise IscaError(f"{k}: more than {MAX_LEVELS + 1} half levels")
            arr = getattr(c, k)
            for i, x in enumerate(v):
                arr[i] = float(x)
            c.vert_coord_input = 1
        elif k in ("tracer_spectral", "tracer_robert_coeff", "tracer_hole_filling", "tracer_sms", "tracer_flux", "tracer_sink", "tracer_advect_vert"):   # field_table entries, [k] = tracer k+1
            if len(v) > MAX_TRACERS:
                raise IscaError(f"{k}: more than {MAX_TRACERS} tracers")
            arr = getattr(c, k)
            for i, x in enumerate(v):
                arr[i] = x
        elif k == "moist":
            for mk, mv in v.items():
                if not hasattr(c.moist, mk):
                    raise IscaError(f"unknown moist physics key {mk!r}")
                setattr(c.moist, mk, mv)
        elif not hasattr(c, k):
            raise IscaError(f"unknown configuration key {k!r}")
        else:
            setattr(c, k, v)
    return c


def wavenumber_dealing(num_fourier: int, world_size: int):
    """m_of_slot[q, ml] (global m or -1) of the boustrophedon deal used by the lat<->m exchange."""
    lib = load_library()
    ml = (num_fourier + 1 + world_size - 1) // world_size
    out = (C.c_int * (world_size * ml))()
    n = C.c_int()
    if lib.isca_wavenumber_dealing(num_fourier, world_size, out, C.byref(n)) != 0:
        raise IscaError(lib.isca_last_error().decode())
    return np.array(out[:], dtype=np.int64).reshape(world_size, n.value)


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class DynCore:
    """One handle = one GPU = one latitude band and one zonal-wavenumber set."""

    def __init__(self, cfg: _CConfig):
        self.lib = load_library()
        self.cfg = cfg
        self._h = C.c_void_p()
        self._check(self.lib.isca_dyn_create(C.byref(cfg), C.byref(self._h)))
        self.I, self.J, self.L = cfg.lon_max, cfg.lat_max, cfg.num_levels
        self.M1, self.N1 = cfg.num_fourier + 1, cfg.num_spherical + 1
        self.Jl = self.info("lat_local")
        self.tracer_names = ["sphum"] + [f"tracer{k + 1}" for k in range(1, max(cfg.num_tracers, 1))]   # field_table names (restart variables)

    # -- plumbing
    def _check(self, rc):
        if rc != 0:
            raise IscaError(self.lib.isca_last_error().decode())

    def close(self):
        if self._h:
            self.lib.isca_dyn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self, name: str) -> int:
        v = C.c_long()
        self._check(self.lib.isca_dyn_get_info(self._h, name.encode(), C.byref(v)))
        return v.value

    def comm_init_env(self):
        """world_size > 1 without a message-passing layer on the host: the communicator's id from rank 0 to the others through the file
        ISCA_COMM_ID_FILE names (ISCA_COMM=ipc: ranks that share one GPU), then the communicator's self-check; isca_dyn_step(n) is then the
        library's own sharded step loop.  Collective over the ranks."""
        self._check(self.lib.isca_dyn_comm_init_env(self._h))

    def set_info(self, name: str, value: int):
        """what idealized_moist_phys_mod keeps beside the fields ("phys_calls"): handing over a running model through set()"""
        self._check(self.lib.isca_dyn_set_info(self._h, name.encode(), int(value)))

    # -- shapes
    def _shape(self, name):
        L, Jl, I, N1, M1 = self.L, self.Jl, self.I, self.N1, self.M1
        if name in ("psg", "dxlp", "dylp", "g_dtlp", "t_surf", "precip", "surf_geopotential"):
            return (Jl, I), False
        if name in ("p_half", "z_half"):
            return (L + 1, Jl, I), False
        if name in ("vors", "divs", "ts", "s_dtvor", "s_dtdiv", "s_dtT", "trs2", "trs3", "trs4"):
            return (L, N1, M1), True
        if name in ("ln_ps", "s_dtlp"):
            return (N1, M1), True
        return (L, Jl, I), False

    def set_surf_geopotential(self, global_field):
        """get_topography's result (m2/s2) as a global [lat_max, lon_max] array, before cold_start / a restart's set calls."""
        a = np.ascontiguousarray(global_field, dtype=np.float64)
        if a.shape != (self.J, self.I):
            raise IscaError(f"set_surf_geopotential: shape {a.shape} != {(self.J, self.I)}")
        self._check(self.lib.isca_dyn_set_surf_geopotential(self._h, _dptr(a), a.size))

    def set_topography(self, surf_height, land_mask=None, ocean_topog_smoothing=0.0):
        """get_topography with topography_option = 'input' on data handed over (isca_dyn_set_topography): the [lat_max, lon_max] height field in m and, for
        ocean_topog_smoothing /= 0, the land mask (> 0 = land).  Returns (lambda, fraction smoothed) -- zeros when the field is only truncated."""
        hgt = np.ascontiguousarray(surf_height, dtype=np.float64)
        if hgt.shape != (self.J, self.I):
            raise IscaError(f"set_topography: shape {hgt.shape} != {(self.J, self.I)}")
        land = None if land_mask is None else np.ascontiguousarray(land_mask, dtype=np.float64)
        if land is not None and land.shape != hgt.shape:
            raise IscaError(f"set_topography: land mask of shape {land.shape} != {hgt.shape}")
        lam, frac = C.c_double(), C.c_double()
        self._check(self.lib.isca_dyn_set_topography(self._h, _dptr(hgt), None if land is None else _dptr(land), float(ocean_topog_smoothing),
                                                     C.cast(C.byref(lam), C.POINTER(C.c_double)), C.cast(C.byref(frac), C.POINTER(C.c_double))))
        if ocean_topog_smoothing != 0.0:
            print(f"\nMessage from subroutine get_topography:\nlambda={lam.value:16.8e}  fraction_smoothed={frac.value:16.8e}\n")
        return lam.value, frac.value

    def get(self, name: str, time_level: int = 1):
        shape, cplx = self._shape(name)
        a = np.zeros(shape, dtype=np.complex128 if cplx else np.float64)
        v = a.view(np.float64)
        self._check(self.lib.isca_dyn_get_state(self._h, name.encode(), time_level, _dptr(v), v.size))
        return a

    def set(self, name: str, value, time_level: int = 1):
        shape, cplx = self._shape(name)
        a = np.ascontiguousarray(value, dtype=np.complex128 if cplx else np.float64)
        if a.shape != shape:
            raise IscaError(f"set({name}): shape {a.shape} != {shape}")
        v = a.view(np.float64)
        self._check(self.lib.isca_dyn_set_state(self._h, name.encode(), time_level, _dptr(v), v.size))

    def table(self, name: str):
        n = {"sin_lat": self.J, "wts_lat": self.J, "deg_lat": self.J, "deg_lon": self.I, "pk": self.L + 1,
             "bk": self.L + 1, "legendre": (self.J // 2) * self.N1 * self.M1, "eigen_laplacian": self.N1 * self.M1,
             "sin_hem": self.J // 2, "wts_hem": self.J // 2, "fixer": 32,
             "wave_matrix": (self.cfg.num_spherical + (0 if self.cfg.triang_trunc else self.cfg.num_fourier)) * self.L * self.L}[name]
        a = np.zeros(n)
        self._check(self.lib.isca_dyn_get_table(self._h, name.encode(), _dptr(a), a.size))
        if name == "legendre":
            a = a.reshape(self.J // 2, self.N1, self.M1)
        elif name == "eigen_laplacian":
            a = a.reshape(self.N1, self.M1)
        elif name == "wave_matrix":
            a = a.reshape(-1, self.L, self.L)
        return a

    # -- model
    def cold_start(self):
        self._check(self.lib.isca_dyn_cold_start(self._h))

    def step(self, nsteps: int = 1, sync: bool = True):
        self._check(self.lib.isca_dyn_step(self._h, nsteps, 1 if sync else 0))

    def synchronize(self):
        self._check(self.lib.isca_dyn_synchronize(self._h))

    def step_phase(self, phase: int):
        self._check(self.lib.isca_dyn_step_phase(self._h, phase))

    def exchange_buffers(self, which: int):
        s, r, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._check(self.lib.isca_dyn_exchange_buffers(self._h, which, C.byref(s), C.byref(r), C.byref(n)))
        return s.value, r.value, n.value

    def halo_buffers(self):
        p = [C.c_void_p() for _ in range(4)]
        n = C.c_size_t()
        self._check(self.lib.isca_dyn_halo_buffers(self._h, *[C.byref(x) for x in p], C.byref(n)))
        return [x.value for x in p], n.value

    def reduce_buffer(self):
        b, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.isca_dyn_reduce_buffer(self._h, C.byref(b), C.byref(n)))
        return b.value, n.value

    def complete_update(self, time_level: int = 1):
        self._check(self.lib.isca_dyn_complete_update(self._h, time_level))

    def write_restart_files(self, directory: str, tracer_names=None):
        """spectral_dynamics_end + atmosphere_end (+ mixed_layer_end) by the library's own netCDF-classic writer (isca_dyn_write_restart): the same
        files isca_amd.restart.write_restart writes through scipy."""
        names = None if tracer_names is None else ",".join(tracer_names).encode()
        self._check(self.lib.isca_dyn_write_restart(self._h, str(directory).encode(), names))

    def read_restart_files(self, directory: str, tracer_names=None):
        """The restart branch of spectral_dynamics_init / atmosphere_init by the library's own reader (isca_dyn_read_restart)."""
        names = None if tracer_names is None else ",".join(tracer_names).encode()
        self._check(self.lib.isca_dyn_read_restart(self._h, str(directory).encode(), names))

    def set_time_pointers(self, previous: int, current: int, step_count: int = 0):
        self._check(self.lib.isca_dyn_set_time_pointers(self._h, previous, current, step_count))

    def refresh_derived(self):
        self._check(self.lib.isca_dyn_refresh_derived(self._h))

    def state(self):
        return {k: self.get(k) for k in ("ug", "vg", "tg", "psg", "vors", "divs", "ts", "ln_ps")}

    # -- transforms_mod
    def _nlev(self, a, nd):
        a = np.asarray(a)
        return (a[None] if a.ndim == nd - 1 else a), a.ndim == nd - 1

    def trans_spherical_to_grid(self, spherical):
        s, two_d = self._nlev(spherical, 3)
        s = np.ascontiguousarray(s, dtype=np.complex128)
        g = np.zeros((s.shape[0], self.Jl, self.I))
        self._check(self.lib.isca_trans_spherical_to_grid(self._h, _dptr(s.view(np.float64)), _dptr(g), s.shape[0]))
        return g[0] if two_d else g

    def trans_grid_to_spherical(self, grid, do_truncation=True):
        g, two_d = self._nlev(grid, 3)
        g = np.ascontiguousarray(g, dtype=np.float64)
        s = np.zeros((g.shape[0], self.N1, self.M1), dtype=np.complex128)
        self._check(self.lib.isca_trans_grid_to_spherical(self._h, _dptr(g), _dptr(s.view(np.float64)), g.shape[0],
                                                          1 if do_truncation else 0))
        return s[0] if two_d else s

    def trans_filter(self, grid, filter=None):
        """trans_filter: spectral truncation of a grid field, optionally times a real (n, m) factor per coefficient."""
        g, two_d = self._nlev(grid, 3)
        g = np.array(g, dtype=np.float64, copy=True)
        f = None if filter is None else np.ascontiguousarray(filter, dtype=np.float64)
        if f is not None and f.shape != (self.N1, self.M1):
            raise IscaError("trans_filter: filter must have the spectral shape (n, m)")
        self._check(self.lib.isca_trans_filter(self._h, _dptr(g), None if f is None else _dptr(f), g.shape[0]))
        return g[0] if two_d else g

    def vor_div_from_uv_grid(self, u, v):
        u, two_d = self._nlev(u, 3); v, _ = self._nlev(v, 3)
        u = np.ascontiguousarray(u, dtype=np.float64); v = np.ascontiguousarray(v, dtype=np.float64)
        vor = np.zeros((u.shape[0], self.N1, self.M1), dtype=np.complex128); div = np.zeros_like(vor)
        self._check(self.lib.isca_vor_div_from_uv_grid(self._h, _dptr(u), _dptr(v), _dptr(vor.view(np.float64)),
                                                       _dptr(div.view(np.float64)), u.shape[0]))
        return (vor[0], div[0]) if two_d else (vor, div)

    def uv_grid_from_vor_div(self, vor, div):
        vor, two_d = self._nlev(vor, 3); div, _ = self._nlev(div, 3)
        vor = np.ascontiguousarray(vor, dtype=np.complex128); div = np.ascontiguousarray(div, dtype=np.complex128)
        u = np.zeros((vor.shape[0], self.Jl, self.I)); v = np.zeros_like(u)
        self._check(self.lib.isca_uv_grid_from_vor_div(self._h, _dptr(vor.view(np.float64)), _dptr(div.view(np.float64)),
                                                       _dptr(u), _dptr(v), vor.shape[0]))
        return (u[0], v[0]) if two_d else (u, v)

    def horizontal_advection(self, field_spec, u, v, tendency):
        s = np.ascontiguousarray(field_spec, dtype=np.complex128)
        u = np.ascontiguousarray(u, dtype=np.float64); v = np.ascontiguousarray(v, dtype=np.float64)
        t = np.array(tendency, dtype=np.float64, order="C", copy=True)
        self._check(self.lib.isca_horizontal_advection(self._h, _dptr(s.view(np.float64)), _dptr(u), _dptr(v), _dptr(t), s.shape[0]))
        return t

    def trans_spherical_to_fourier(self, spherical):
        s = np.ascontiguousarray(spherical, dtype=np.complex128)
        f = np.zeros((s.shape[0], self.J, self.M1), dtype=np.complex128)
        self._check(self.lib.isca_trans_spherical_to_fourier(self._h, _dptr(s.view(np.float64)), _dptr(f.view(np.float64)), s.shape[0]))
        return f

    def trans_fourier_to_spherical(self, fourier):
        f = np.ascontiguousarray(fourier, dtype=np.complex128)
        s = np.zeros((f.shape[0], self.N1, self.M1), dtype=np.complex128)
        self._check(self.lib.isca_trans_fourier_to_spherical(self._h, _dptr(f.view(np.float64)), _dptr(s.view(np.float64)), f.shape[0]))
        return s

    def trans_grid_to_fourier(self, grid):
        g = np.ascontiguousarray(grid, dtype=np.float64)
        f = np.zeros((g.shape[0], self.J, self.M1), dtype=np.complex128)
        self._check(self.lib.isca_trans_grid_to_fourier(self._h, _dptr(g), _dptr(f.view(np.float64)), g.shape[0]))
        return f

    def trans_fourier_to_grid(self, fourier):
        f = np.ascontiguousarray(fourier, dtype=np.complex128)
        g = np.zeros((f.shape[0], self.J, self.I))
        self._check(self.lib.isca_trans_fourier_to_grid(self._h, _dptr(f.view(np.float64)), _dptr(g), f.shape[0]))
        return g

    def area_weighted_global_mean(self, field2d):
        a = np.ascontiguousarray(field2d, dtype=np.float64)
        out = C.c_double()
        self._check(self.lib.isca_area_weighted_global_mean(self._h, _dptr(a), C.cast(C.byref(out), C.POINTER(C.c_double))))
        return out.value

    # ---- physics = 2: the host keeps its own physics package and hands its tendencies to the dynamics (atmosphere.F90:300-329)
    def delta_t(self):
        """Time step the physics of the coming step receives (dt_atmos on a first step, else 2 dt_atmos: atmosphere.F90:286-290)."""
        v = C.c_double()
        self._check(self.lib.isca_dyn_delta_t(self._h, C.byref(v)))
        return v.value

    def _tend_ptrs(self, arrs):
        keep, ptrs = [], []
        ntr = max(self.cfg.num_tracers, 1)
        for i, a in enumerate(arrs):
            if a is None:
                ptrs.append(None)
            else:
                a = np.ascontiguousarray(a, dtype=np.float64)
                want = (self.L, self.Jl, self.I)
                if i == 3 and ntr > 1:                      # dt_tracers(:,:,:,ntr): one (lev, lat, lon) block per tracer, tracer index slowest
                    if a.shape != (ntr,) + want:
                        raise IscaError(f"dt_tracers must be a ({ntr}, lev, lat_local, lon) array: one block per tracer of the field_table")
                elif a.shape != want:
                    raise IscaError("physics tendencies must be (lev, lat_local, lon) arrays")
                keep.append(a); ptrs.append(_dptr(a))
        return keep, ptrs

    def set_tendencies(self, dt_ug=None, dt_vg=None, dt_tg=None, dt_tracers=None):
        keep, ptrs = self._tend_ptrs((dt_ug, dt_vg, dt_tg, dt_tracers))
        self._check(self.lib.isca_dyn_set_tendencies(self._h, *ptrs, 0))

    def dynamics(self, dt_ug=None, dt_vg=None, dt_tg=None, dt_tracers=None, sync=True):
        """spectral_dynamics (spectral_dynamics.F90:780-795) with the caller's physics tendencies: one step."""
        keep, ptrs = self._tend_ptrs((dt_ug, dt_vg, dt_tg, dt_tracers))
        self._check(self.lib.isca_dyn_dynamics(self._h, *ptrs, 0, 1 if sync else 0))

    def hs_forcing(self, dt, p_half, p_full, u, v, t, udt=None, vdt=None, tdt=None):
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (p_half, p_full, u, v, t)]
        outs = [np.zeros_like(arrs[2]) if x is None else np.array(x, dtype=np.float64, copy=True) for x in (udt, vdt, tdt)]
        self._check(self.lib.isca_hs_forcing(self._h, float(dt), *[_dptr(a) for a in arrs], *[_dptr(o) for o in outs]))
        return tuple(outs)

    def idealized_moist_phys(self, delta_t, gust, rad_lat, u_prev, v_prev, t_prev, q_prev, p_half_prev, p_full_prev, p_half_cur, p_full_cur,
                             z_half_cur, z_full_cur, t_surf):
        """idealized_moist_phys on independent columns: arrays [lev(+1), ncol]; returns dt_u, dt_v, dt_t, dt_q, t_surf_new, precip."""
        ins = [np.ascontiguousarray(x, dtype=np.float64) for x in (rad_lat, u_prev, v_prev, t_prev, q_prev, p_half_prev, p_full_prev,
                                                                      p_half_cur, p_full_cur, z_half_cur, z_full_cur)]
        L, ncol = ins[1].shape
        if L != self.L or any(a.shape != (L, ncol) for a in (ins[2], ins[3], ins[4], ins[6], ins[8], ins[10])) or \
                any(a.shape != (L + 1, ncol) for a in (ins[5], ins[7], ins[9])) or ins[0].shape != (ncol,):
            raise IscaError("idealized_moist_phys: inconsistent array shapes")
        ts = np.array(t_surf, dtype=np.float64, copy=True)
        outs = [np.zeros((L, ncol)) for _ in range(4)]
        precip = np.zeros(ncol)
        self._check(self.lib.isca_idealized_moist_phys(self._h, ncol, float(delta_t), float(gust), *[_dptr(a) for a in ins], _dptr(ts),
                                                       *[_dptr(o) for o in outs], _dptr(precip)))
        return (*outs, ts, precip)

    # -- components of the step on caller fields (spherical_mod / press_and_geopot_mod / fv_advection_mod / ...)
    def _spec(self, a):
        a, two_d = self._nlev(a, 3)
        return np.ascontiguousarray(a, dtype=np.complex128), two_d

    def _spec_call(self, fn, ins, nout, *extra):
        arrs, two_d = zip(*[self._spec(a) for a in ins])
        outs = [np.zeros_like(arrs[0]) for _ in range(nout)]
        self._check(fn(self._h, *[_dptr(a.view(np.float64)) for a in arrs], *[_dptr(o.view(np.float64)) for o in outs],
                       arrs[0].shape[0], *extra))
        outs = [o[0] if two_d[0] else o for o in outs]
        return outs[0] if nout == 1 else tuple(outs)

    def compute_laplacian(self, spherical, power: int = 1):
        return self._spec_call(self.lib.isca_compute_laplacian, [spherical], 1, int(power))

    def compute_gradient_cos(self, spherical):
        return self._spec_call(self.lib.isca_compute_gradient_cos, [spherical], 2)

    def compute_lon_deriv_cos(self, spherical):
        return self.compute_gradient_cos(spherical)[0]

    def compute_lat_deriv_cos(self, spherical):
        return self.compute_gradient_cos(spherical)[1]

    def compute_ucos_vcos(self, vorticity, divergence):
        return self._spec_call(self.lib.isca_compute_ucos_vcos, [vorticity, divergence], 2)

    def compute_vor_div(self, u_div_cos, v_div_cos):
        return self._spec_call(self.lib.isca_compute_vor_div, [u_div_cos, v_div_cos], 2)

    def triangular_truncation(self, spherical):
        a, two_d = self._spec(spherical)
        a = a.copy()
        self._check(self.lib.isca_triangular_truncation(self._h, _dptr(a.view(np.float64)), a.shape[0]))
        return a[0] if two_d else a

    def divide_by_cos(self, grid, power: int = 1):
        g, two_d = self._nlev(grid, 3)
        g = np.array(g, dtype=np.float64, copy=True, order="C")
        self._check(self.lib.isca_divide_by_cos(self._h, _dptr(g), g.shape[0], power))
        return g[0] if two_d else g

    def divide_by_cos2(self, grid):
        return self.divide_by_cos(grid, 2)

    def mass_weighted_global_integral(self, field, surf_press):
        f = np.ascontiguousarray(field, dtype=np.float64); ps = np.ascontiguousarray(surf_press, dtype=np.float64)
        if f.shape != (self.L, self.Jl, self.I) or ps.shape != (self.Jl, self.I):
            raise IscaError("mass_weighted_global_integral: field (lev,lat,lon) and surf_press (lat,lon) expected")
        out = C.c_double()
        self._check(self.lib.isca_mass_weighted_global_integral(self._h, _dptr(f), _dptr(ps), C.cast(C.byref(out), C.POINTER(C.c_double))))
        return out.value

    def pressure_variables(self, surf_p):
        ps = np.ascontiguousarray(surf_p, dtype=np.float64)
        if ps.shape != (self.Jl, self.I):
            raise IscaError("pressure_variables: surf_p (lat,lon) expected")
        ph = np.zeros((self.L + 1, self.Jl, self.I)); lph = np.zeros_like(ph)
        pf = np.zeros((self.L, self.Jl, self.I)); lpf = np.zeros_like(pf)
        self._check(self.lib.isca_pressure_variables(self._h, _dptr(ps), _dptr(ph), _dptr(lph), _dptr(pf), _dptr(lpf)))
        return ph, lph, pf, lpf

    def compute_geopotential(self, t, ln_p_half, ln_p_full, surf_geopotential=None, q_grid=None):
        """press_and_geopot.F90:314-359.  Without the two optional arguments: on the handle's own surface geopotential, no q_grid."""
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (t, ln_p_half, ln_p_full)]
        gf = np.zeros((self.L, self.Jl, self.I)); gh = np.zeros((self.L + 1, self.Jl, self.I))
        if surf_geopotential is None and q_grid is None:
            self._check(self.lib.isca_compute_geopotential(self._h, *[_dptr(x) for x in a], _dptr(gf), _dptr(gh)))
            return gf, gh
        opt = [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in (surf_geopotential, q_grid)]
        if opt[0] is not None and opt[0].shape != (self.Jl, self.I) or opt[1] is not None and opt[1].shape != a[0].shape:
            raise IscaError("compute_geopotential: surf_geopotential (lat,lon) / q_grid (lev,lat,lon) expected")
        self._check(self.lib.isca_compute_geopotential_surf(self._h, *[_dptr(x) for x in a], *[None if x is None else _dptr(x) for x in opt],
                                                            _dptr(gf), _dptr(gh)))
        return gf, gh

    def _grid3(self, *arrs):
        out = [np.ascontiguousarray(x, dtype=np.float64) for x in arrs]
        for x in out:
            if x.shape != (self.L, self.Jl, self.I):
                raise IscaError(f"grid field of shape {(self.L, self.Jl, self.I)} expected, got {x.shape}")
        return out

    def a_grid_horiz_advection(self, u, v, q, dt, tendency=None):
        u, v, q = self._grid3(u, v, q)
        tend = np.zeros_like(q) if tendency is None else np.array(tendency, dtype=np.float64, copy=True, order="C")
        self._check(self.lib.isca_a_grid_horiz_advection(self._h, _dptr(u), _dptr(v), _dptr(q), float(dt), _dptr(tend)))
        return tend

    def vert_advection_ppm(self, dt, w, surf_p, r):
        (r,) = self._grid3(r)
        w = np.ascontiguousarray(w, dtype=np.float64); ps = np.ascontiguousarray(surf_p, dtype=np.float64)
        if w.shape != (self.L + 1, self.Jl, self.I) or ps.shape != (self.Jl, self.I):
            raise IscaError("vert_advection_ppm: w (lev+1,lat,lon) and surf_p (lat,lon) expected")
        rdt = np.zeros_like(r)
        self._check(self.lib.isca_vert_advection_ppm(self._h, float(dt), _dptr(w), _dptr(ps), _dptr(r), _dptr(rdt)))
        return rdt

    def hs_tracer_source_sink(self, surf_p, r, rdt=None):
        (r,) = self._grid3(r)
        ps = np.ascontiguousarray(surf_p, dtype=np.float64)
        out = np.zeros_like(r) if rdt is None else np.array(rdt, dtype=np.float64, copy=True, order="C")
        self._check(self.lib.isca_hs_tracer_source_sink(self._h, _dptr(ps), _dptr(r), _dptr(out)))
        return out

    def _spec3(self, a, two_d=False):
        a = np.array(a, dtype=np.complex128, copy=True, order="C")
        want = (self.N1, self.M1) if two_d else (self.L, self.N1, self.M1)
        if a.shape != want:
            raise IscaError(f"spectral array of shape {want} expected, got {a.shape}")
        return a

    def implicit_correction(self, dt_divs, dt_ts, dt_ln_ps, divs, ts, ln_ps, delta_t):
        """divs, ts, ln_ps: (previous, current) pairs.  Returns the corrected (dt_divs, dt_ts, dt_ln_ps)."""
        o = [self._spec3(dt_divs), self._spec3(dt_ts), self._spec3(dt_ln_ps, True)]
        i = [self._spec3(divs[0]), self._spec3(divs[1]), self._spec3(ts[0]), self._spec3(ts[1]),
             self._spec3(ln_ps[0], True), self._spec3(ln_ps[1], True)]
        self._check(self.lib.isca_implicit_correction(self._h, *[_dptr(x.view(np.float64)) for x in o + i], float(delta_t)))
        return tuple(o)

    def compute_spectral_damping(self, field_previous, dt_field, delta_t, kind="t"):
        f, d = self._spec3(field_previous), self._spec3(dt_field)
        which = {"t": 0, "vor": 1, "div": 2}[kind]
        self._check(self.lib.isca_compute_spectral_damping(self._h, which, _dptr(f.view(np.float64)), _dptr(d.view(np.float64)), float(delta_t)))
        return d

    def leapfrog(self, previous, current, dt_field, delta_t, robert_coeff=0.04):
        """-> (new level, Robert-filtered current)"""
        p, c, d = self._spec3(previous), self._spec3(current), self._spec3(dt_field)
        self._check(self.lib.isca_leapfrog(self._h, *[_dptr(x.view(np.float64)) for x in (p, c, d)], float(delta_t), float(robert_coeff)))
        return p, c

    @staticmethod
    def comm_selftest(device: int = 0) -> float:
        """Load RCCL and run each collective of the sharded step once on a one-rank communicator; returns the max error."""
        lib = load_library()
        e = C.c_double(-1.0)
        if lib.isca_comm_selftest(device, C.cast(C.byref(e), C.POINTER(C.c_double))) != 0:
            raise IscaError(lib.isca_last_error().decode())
        return e.value

    # -- diagnostics (spectral_diagnostics + time averaging)
    def diag_select(self, names):
        """names: iterable of the reference's diagnostic field names (or a comma-separated string); () switches off."""
        txt = names if isinstance(names, str) else ",".join(names)
        self._check(self.lib.isca_dyn_diag_select(self._h, txt.encode()))

    def diag_mean(self, name: str, reset: bool = False):
        """-> (time mean since the last reset, number of steps in it)"""
        a = np.zeros((self.Jl, self.I) if name in ("ps", "precipitation", "t_surf") else (self.L, self.Jl, self.I))
        n = C.c_long()
        self._check(self.lib.isca_dyn_diag_read(self._h, name.encode(), _dptr(a), a.size, C.byref(n), 1 if reset else 0))
        return a, n.value

    def diag_open(self, diag_table: str, directory: str = ".", start_seconds: float = 0.0):
        """history files written by the library itself (csrc/history_nc.cpp): `diag_table` = the path of the reference-format table or its text"""
        self._check(self.lib.isca_dyn_diag_open(self._h, str(diag_table).encode(), str(directory).encode(), float(start_seconds)))

    def diag_close(self):
        self._check(self.lib.isca_dyn_diag_close(self._h))

    def diag_reset(self, name_of_any_selected_field: str):
        n = C.c_long()
        self._check(self.lib.isca_dyn_diag_read(self._h, name_of_any_selected_field.encode(), None, 0, C.byref(n), 1))
        return n.value

    # -- measurement
    def bench_transform_pair(self, nfields: int, reps: int = 20):
        pair = C.c_double()
        k = np.zeros(4)
        self._check(self.lib.isca_bench_transform_pair(self._h, nfields, reps, C.cast(C.byref(pair), C.POINTER(C.c_double)), _dptr(k)))
        return pair.value, dict(zip(("legendre_inv", "fft_inv", "fft_fwd", "legendre_fwd"), k.tolist()))

    def kernel_times(self, enable=True):
        ms = np.zeros(64)
        names = C.create_string_buffer(4096)
        n = C.c_int()
        # enable: False / 0 stop, True / 1 one HIP-event pair per kernel, 2 one per run of kernels between two exchanges of the sharded step ("seg_*")
        self._check(self.lib.isca_dyn_kernel_times(self._h, int(enable), _dptr(ms), 64, names, 4096, C.byref(n)))
        nm = [x for x in names.value.decode().split(";") if x]
        return dict(zip(nm, ms[: n.value].tolist()))
