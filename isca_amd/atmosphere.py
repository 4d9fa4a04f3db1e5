"""Host-side mirror of the reference's module procedures for the hot path.

Same names, argument meaning and error behaviour as
  atmosphere_mod          (src/atmos_spectral/driver/solo/atmosphere.F90:78,120-390)
  spectral_dynamics_mod   (src/atmos_spectral/model/spectral_dynamics.F90:95-98)
  transforms_mod          (src/atmos_spectral/tools/transforms.F90:134-184)
Like the Fortran modules this keeps ONE module-level instance ("module_is_initialized"); errors that are
FATAL in the reference raise IscaError.  Arrays are numpy views of the reference's Fortran layouts:
grid (lon,lat,lev) <-> [lev,lat,lon], spectral (m,n,lev) <-> [lev,n,m].
"""
from __future__ import annotations

import math
import os
import sys
import re
import numpy as np

from . import dyncore, restart
from .dyncore import IscaError, RESOLUTIONS

_core: dyncore.DynCore | None = None
_run_dir: str | None = None
_NML_GROUPS = ("spectral_dynamics_nml", "hs_forcing_nml", "main_nml")
# hs_forcing_nml values that belong to local_heating_option = 'from_file' / relax_to_specified_wind / equilibrium_t_option other than 'Held_Suarez' (hs_forcing.F90:86-118)
_HS_LOCAL_HEATING = ("local_heating_srfamp", "local_heating_xwidth", "local_heating_ywidth", "local_heating_xcenter", "local_heating_ycenter",
                     "local_heating_vert_decay")
_HS_UNUSED_PARAMETERS = ("local_heating_file", "u_wind_file", "v_wind_file", "equilibrium_t_file", "p_trop", "alpha",
                         "peri_time", "smaxis", "albedo", "lapse", "h_a", "tau_s", "orbital_period", "heat_capacity", "ml_depth", "spinup_time",
                         "stratosphere_t_option")

# ---- moist physics package (atmosphere_nml: idealized_moist_model = .true.; exp/test_cases/frierson/frierson_test_case.py:49-170)
# namelist variables handed to the C config (group -> {variable: isca_moist_config member})
_MOIST_KEYS = {
    "idealized_moist_phys_nml": {"roughness_mom": "roughness_mom", "roughness_heat": "roughness_heat", "roughness_moist": "roughness_moist"},
    "two_stream_gray_rad_nml": {k: k for k in ("solar_constant", "del_sol", "del_sw", "ir_tau_eq", "ir_tau_pole", "atm_abs", "odp", "sw_diff",
                                                "linear_tau", "wv_exponent", "solar_exponent")},
    "mixed_layer_nml": {"depth": "depth", "tconst": "tconst", "delta_t": "delta_T", "albedo_value": "albedo_value", "evaporation": "evaporation"},
    "qe_moist_convection_nml": {"tau_bm": "tau_bm", "rhbm": "rhbm", "tmin": "Tmin", "tmax": "Tmax", "val_inc": "val_inc"},
    "damping_driver_nml": {"do_rayleigh": "do_rayleigh", "trayfric": "trayfric", "sponge_pbottom": "sponge_pbottom",
                           "do_conserve_energy": "damping_conserve_energy"},
    "vert_turb_driver_nml": {"constant_gust": "constant_gust"},
    "diffusivity_nml": {"frac_inner": "frac_inner", "rich_crit_pbl": "rich_crit_pbl"},
    "monin_obukhov_nml": {"rich_crit": "rich_crit", "drag_min": "drag_min"},
}
# options the device package implements in exactly one way: any other value is refused, as an unsupported option
_MOIST_FIXED = {
    "idealized_moist_phys_nml": {"two_stream_gray": True, "convection_scheme": "SIMPLE_BETTS_MILLER", "do_damping": True, "turb": True,
                                 "mixed_layer_bc": True, "do_virtual": False, "do_simple": True, "do_rrtm_radiation": False,
                                 "do_socrates_radiation": False, "do_cloud_simple": False, "bucket": False, "gp_surface": False,
                                 "do_lcl_diffusivity_depth": False, "land_option": "none"},
    "two_stream_gray_rad_nml": {"rad_scheme": "frierson", "do_seasonal": False},
    "mixed_layer_nml": {"prescribe_initial_dist": True, "do_qflux": False, "do_sc_sst": False, "do_ape_sst": False,
                        "update_albedo_from_ice": False},
    "vert_turb_driver_nml": {"do_mellor_yamada": False, "do_diffusivity": True, "do_simple": True, "use_tau": False, "gust_scheme": "constant",
                             "do_shallow_conv": False, "do_molecular_diffusion": False},
    "diffusivity_nml": {"do_entrain": False, "do_simple": True, "fixed_depth": False, "free_atm_diff": False, "pbl_mcm": False},
    "surface_flux_nml": {"use_virtual_temp": False, "do_simple": True, "old_dtaudv": True},
    "lscale_cond_nml": {"do_simple": True, "do_evap": True},
    "sat_vapor_pres_nml": {"do_simple": True},
    "monin_obukhov_nml": {"neutral": False, "stable_option": 1},
    "damping_driver_nml": {"do_cg_drag": False, "do_mg_drag": False, "do_topo_drag": False},
}


# Module defaults of the reference's namelist variables (what applies when an input.nml leaves a variable out), as C config members:
# spectral_dynamics.F90:152-206, hs_forcing.F90:76-84; main_nml's dt_atmos defaults to 0 = "dt_atmos has not been specified"
# (atmos_model.F90:111, FATAL).  isca_dyn_config_default holds the Held-Suarez TEST CASE instead (resolution-less use, benchmarks).
_REF_DEFAULTS = dict(
    do_mass_correction=1, do_water_correction=1, do_energy_correction=1, triang_trunc=1, damping_order=2,
    lon_max=128, lat_max=64, num_fourier=42, num_spherical=43, fourier_inc=1, num_levels=18, damping_coeff=1.15740741e-4,
    eddy_sponge_coeff=0., zmu_sponge_coeff=0., zmv_sponge_coeff=0., robert_coeff=.04, alpha_implicit=.5, scale_heights=4., surf_res=.1,
    exponent=2.5, initial_sphum=0.0, reference_sea_level_press=101325., water_correction_limit=0.0, raw_filter_coeff=1.0,
    valid_range_t=(100., 500.), dt_atmos=0.0, cutoff_wn=15, damping_coeff_vor=-1., damping_coeff_div=-1., damping_order_vor=-1,
    damping_order_div=-1, vert_advect_uv=0, vert_advect_t=0, use_implicit=1, make_symmetric=0, vert_difference_option=0,
    t_zero=315., t_strat=200., delh=60., delv=10., eps=0., sigma_b=0.7, ka=-40., ks=-4., kf=-1., do_conserve_energy=1, trflux=1.e-5,
    trsink=-4., P00=1.e5)
_REF_VERT_COORD_OPTION = "even_sigma"            # spectral_dynamics.F90:175
_VERT_ADVECT_SCHEMES = {"SECOND_CENTERED": 0, "FOURTH_CENTERED": 1, "VAN_LEER_LINEAR": 2, "FINITE_VOLUME_PARABOLIC": 3}     # spectral_dynamics.F90:280-301
_VERT_DIFFERENCE_OPTIONS = {"simmons_and_burridge": 0, "mcm": 1}     # spectral_dynamics.F90:1063, 1084
_DAMPING_OPTIONS = {"resolution_dependent": 0, "exponential_cutoff": 1, "resolution_independent": 2}     # spectral_damping.F90:124-153
# moist package, isca_moist_config members: idealized_moist_phys.F90:136-138, two_stream_gray_rad.F90:72-82, mixed_layer.F90:84-95,
# qe_moist_convection.F90:66-70, damping_driver.f90:42-56, vert_turb_driver.F90:116, diffusivity.F90:127-128, monin_obukhov.F90:88-89
_MOIST_REF_DEFAULTS = dict(
    roughness_mom=0.05, roughness_heat=0.05, roughness_moist=0.05, solar_constant=1360.0, del_sol=1.4, del_sw=0.0, ir_tau_eq=6.0,
    ir_tau_pole=1.5, atm_abs=0.0, odp=1.0, sw_diff=0.0, linear_tau=0.1, wv_exponent=4.0, solar_exponent=4.0, depth=40.0, tconst=305.0,
    delta_T=40.0, albedo_value=0.06, evaporation=1, tau_bm=7200., rhbm=.8, Tmin=173., Tmax=335., val_inc=0.01, do_rayleigh=0, trayfric=0.,
    sponge_pbottom=50., damping_conserve_energy=0, constant_gust=1.0, frac_inner=0.1, rich_crit_pbl=1.0, rich_crit=2.0, drag_min=1.e-05)
# reference defaults of the options in _MOIST_FIXED that differ from the one value implemented here: an input.nml that leaves them out
# asks the reference for something this package does not do, so they must be given
_MOIST_MUST_SET = {
    "idealized_moist_phys_nml": {"convection_scheme": "unset", "do_damping": False, "turb": False, "mixed_layer_bc": False, "do_simple": False},
    "mixed_layer_nml": {"prescribe_initial_dist": False},
    "vert_turb_driver_nml": {"do_mellor_yamada": True, "do_diffusivity": False, "do_simple": False, "use_tau": True},
    "diffusivity_nml": {"do_entrain": True, "do_simple": False},
    "surface_flux_nml": {"use_virtual_temp": True, "do_simple": False, "old_dtaudv": False},
    "lscale_cond_nml": {"do_simple": False, "do_evap": False},
    "sat_vapor_pres_nml": {"do_simple": False},
}


def _moist_config(namelist: dict) -> dict:
    """idealized_moist_phys_init and friends: collect the namelist variables of the moist package, refuse what is not implemented."""
    mo: dict = dict(_MOIST_REF_DEFAULTS)
    for grp, fixed in _MOIST_FIXED.items():
        for k, v in (namelist.get(grp) or {}).items():
            k = k.lower()
            if k in fixed:
                want = fixed[k]
                same = str(v).lower() == str(want).lower() if isinstance(want, str) else (bool(v) == want if isinstance(want, bool) else v == want)
                if not same:
                    raise IscaError(f'{grp}: "{v}" is not a supported value for {k} (only "{want}")')
    for grp, keys in _MOIST_KEYS.items():
        for k, v in (namelist.get(grp) or {}).items():
            k = k.lower()
            if k in keys:
                mo[keys[k]] = int(v) if isinstance(v, bool) else v
            elif k not in _MOIST_FIXED.get(grp, {}) and grp != "idealized_moist_phys_nml":
                raise IscaError(f"{grp}: {k} is not supported by the device physics package")
    for grp, must in _MOIST_MUST_SET.items():
        given = {k.lower() for k in (namelist.get(grp) or {})}
        for k, ref_default in must.items():
            if k not in given:
                raise IscaError(f'{grp}: {k} is not set and defaults to "{ref_default}" in the reference; only '
                                f'"{_MOIST_FIXED[grp][k]}" is supported, so it has to be given')
    return mo


def parse_namelist(text: str) -> dict:
    """Minimal reader of the reference's input.nml format (&group key = value, ... /)."""
    out: dict = {}
    text = re.sub(r"!.*", "", text)
    pos = 0
    while True:
        m = re.search(r"&(\w+)", text[pos:])
        if not m:
            break
        grp = m.group(1).lower()
        i = pos + m.end()
        j, quote = i, None
        while j < len(text):                       # the terminating '/' outside quotes
            ch = text[j]
            if quote:
                quote = None if ch == quote else quote
            elif ch in "'\"":
                quote = ch
            elif ch == "/":
                break
            j += 1
        body = text[i:j]
        pos = j + 1
        d = out.setdefault(grp, {})
        keys = list(re.finditer(r"(\w+)\s*=", body))
        for n, km in enumerate(keys):
            val = body[km.end(): keys[n + 1].start() if n + 1 < len(keys) else len(body)]
            vals = [v.strip() for v in val.replace("\n", " ").split(",") if v.strip()]
            conv = []
            for v in vals:
                lv = v.lower()
                if lv in (".true.", "t", ".t."):
                    conv.append(True)
                elif lv in (".false.", "f", ".f."):
                    conv.append(False)
                elif v[0] in "'\"":
                    conv.append(v.strip("'\""))
                else:
                    try:
                        conv.append(int(v))
                    except ValueError:
                        conv.append(float(lv.replace("d", "e")))
            if conv:
                d[km.group(1).lower()] = conv[0] if len(conv) == 1 else conv
    return out


_V197_BK = [0.0, .0089163, .0342936, .0740741, .1262002, .1886145, .2592592, .3360768, .4170096, .5000000, .5829904, .6639231, .7407407,
            .8113854, .8737997, .9259259, .9657064, .9910837, 1.0]                                           # compute_v197_sigma (:276-294)
_MCM_BK = [0.0, .03, .0707, .1311, .2102, .3036, .4062, .5138, .6226, .7284, .8255, .9066, .9640, .9933, 1.0]  # compute_old_model_sigma (:296-310)


def named_vert_coord(option, num_levels, scale_heights, surf_res, exponent, p_press, p_sigma, reference_press):
    """(pk, bk) of vert_coord_option = 'hybrid' | 'mcm' | 'v197' (init/vert_coordinate.F90:89-157): 'hybrid' blends an uneven-sigma profile
    used as sigma (b) with the same profile used as pressure (a) through transition() = sin^2 between p_press and p_sigma."""
    if option == "v197" or option == "mcm":
        bk = _V197_BK if option == "v197" else _MCM_BK
        if num_levels != len(bk) - 1:
            raise IscaError(f"compute_{'v197' if option == 'v197' else 'old_model'}_sigma: num_levels={num_levels} It must be {len(bk) - 1}")
        return [0.0] * len(bk), list(bk)
    if scale_heights == 0. or exponent == 0.:
        raise IscaError("compute_vert_coord: zero is an invalid value for scale_heights / exponent.")
    if not (0. < surf_res <= 1.0):
        raise IscaError(f"compute_vert_coord: the namelist parameter surf_res must be < 1.0, but surf_res={surf_res}")
    if p_sigma < p_press:
        raise IscaError(f"compute_vert_coord: p_sigma must be greater than p_press, but p_sigma={p_sigma}  p_press={p_press}")
    s2 = 1.0 - surf_res
    prof = []
    for k in range(num_levels):                        # compute_uneven_sigma(..., zero_top = .false.) (:248-273)
        zeta = 1. - (float(k) / float(num_levels))
        z = surf_res * zeta + s2 * (zeta ** exponent)
        prof.append(math.exp(-z * scale_heights))
    prof.append(1.0)
    pk, bk = [], []
    for p in prof:                                     # transition (:161-183)
        if p <= p_press:
            f = 0.0
        elif p >= p_sigma:
            f = 1.0
        else:
            f = (math.sin(0.5 * math.pi * (p - p_press) / (p_sigma - p_press))) ** 2
        pk.append(reference_press * (0.0 * f + p * (1.0 - f)))
        bk.append(p * f + 0.0 * (1.0 - f))
    return pk, bk


def config_from_namelist(namelist: dict | str | None, resolution: str | None = None, **overrides):
    """Build the C config from namelist groups (dict as in held_suarez_test_case.py:45-98, or input.nml text).  Variables a given
    namelist leaves out take the reference's MODULE defaults (_REF_DEFAULTS), as they do when the reference reads that input.nml;
    namelist = None keeps the library's Held-Suarez test-case preset (isca_dyn_config_default)."""
    if isinstance(namelist, str):
        namelist = parse_namelist(namelist)
    kw: dict = {} if namelist is None else dict(_REF_DEFAULTS)
    have_nml = namelist is not None
    if have_nml and resolution is not None:          # Experiment.set_resolution: over the module defaults, under what the namelist sets itself
        kw.update(dyncore.RESOLUTIONS[resolution])
    namelist = {g.lower(): v for g, v in (namelist or {}).items()}
    moist = bool(namelist.get("atmosphere_nml", {}).get("idealized_moist_model", False))
    vc = namelist.get("vert_coordinate_nml")
    vco = str(namelist.get("spectral_dynamics_nml", {}).get("vert_coord_option", _REF_VERT_COORD_OPTION if have_nml else "uneven_sigma")).lower()
    if vco == "even_sigma":               # compute_even_sigma (init/vert_coordinate.F90:230-244): bk = (k-1)/num_levels, pk = 0
        nl = overrides.get("num_levels", namelist.get("spectral_dynamics_nml", {}).get("num_levels", kw.get("num_levels", 25)))
        kw["bk_input"] = [float(k) / float(nl) for k in range(nl)] + [1.0]
        kw["pk_input"] = [0.0] * (nl + 1)
    if vco in ("hybrid", "mcm", "v197"):     # compute_vert_coord's other options (init/vert_coordinate.F90:124-152): formed here, handed over like 'input'
        sd = {k.lower(): v for k, v in namelist.get("spectral_dynamics_nml", {}).items()}
        nl = overrides.get("num_levels", sd.get("num_levels", kw.get("num_levels", 25)))
        kw["pk_input"], kw["bk_input"] = named_vert_coord(
            vco, nl, sd.get("scale_heights", kw.get("scale_heights", 4.0)), sd.get("surf_res", kw.get("surf_res", 0.1)),
            sd.get("exponent", kw.get("exponent", 2.5)), sd.get("p_press", 0.1), sd.get("p_sigma", 0.3),
            sd.get("reference_sea_level_press", kw.get("reference_sea_level_press", 101325.0)))
    if vco == "input":
        if not vc or "bk" not in vc:
            raise IscaError("vert_coord_option = 'input' needs vert_coordinate_nml with bk (and pk)")
        bk = list(vc["bk"])
        kw["bk_input"] = bk
        kw["pk_input"] = list(vc.get("pk", [0.0] * len(bk)))
        if len(kw["pk_input"]) != len(bk):
            raise IscaError("vert_coordinate_nml: pk and bk must have the same length")
        nl = namelist.get("spectral_dynamics_nml", {}).get("num_levels", len(bk) - 1)
        if len(bk) != nl + 1:
            raise IscaError("vert_coordinate_nml: bk must hold num_levels+1 values")
    for k, v in namelist.get("constants_nml", {}).items():
        if k.lower() not in ("radius", "omega"):
            raise IscaError(f"constants_nml: {k} cannot be changed (only radius and omega)")
        kw[k.lower()] = v
    ic = {k.lower(): v for k, v in (namelist.get("spectral_init_cond_nml") or {}).items()}     # topography_option: atmosphere_init
    if "initial_temperature" in ic:
        kw["initial_temperature"] = ic["initial_temperature"]
    if moist:
        kw["physics"] = 1
        kw["moist"] = _moist_config(namelist)
    dopt = str(namelist.get("spectral_dynamics_nml", {}).get("damping_option", "resolution_dependent")).lower()
    if dopt not in _DAMPING_OPTIONS:
        raise IscaError(f'"{dopt}" is an invalid value for damping_option')                      # spectral_damping.F90:152-153
    kw["damping_option"] = _DAMPING_OPTIONS[dopt]
    unsupported = {"vert_coord_option": vco if vco in ("input", "even_sigma", "hybrid", "mcm", "v197") else "uneven_sigma", "damping_option": dopt,
                   "initial_state_option": "quiescent",
                   "equilibrium_t_option": "Held_Suarez"}
    no_forcing = False
    for grp in _NML_GROUPS:
        for k, v in (namelist or {}).get(grp, {}).items():
            k = k.lower()
            if k in unsupported:
                if str(v).lower() != unsupported[k].lower():
                    raise IscaError(f'"{v}" is not a supported value for {k} (only "{unsupported[k]}")')
                continue
            if k in ("use_virtual_temperature", "use_implicit", "make_symmetric"):
                kw[k] = int(bool(v))
                continue
            if k == "vert_difference_option":                # press_and_geopot.F90:164, 196, 216-219
                if str(v).lower() not in _VERT_DIFFERENCE_OPTIONS:
                    raise IscaError(f'"{v}" is not a valid value for vert_difference_option')
                kw[k] = _VERT_DIFFERENCE_OPTIONS[str(v).lower()]
                continue
            if k in ("vert_advect_uv", "vert_advect_t"):     # spectral_dynamics.F90:280-301
                if str(v).upper() not in _VERT_ADVECT_SCHEMES:
                    raise IscaError(f'"{v}" is not a valid value for {k}.')
                kw[k] = _VERT_ADVECT_SCHEMES[str(v).upper()]
                continue
            if k in ("p_press", "p_sigma"):           # vert_coord_option = 'hybrid' (used above)
                continue
            if grp == "hs_forcing_nml":
                if k == "no_forcing":                 # hs_forcing returns at once (hs_forcing.F90:174): no drag, no heating, no tracer source
                    no_forcing = bool(v)
                    continue
                if k == "local_heating_option" and str(v).strip() == "" or k == "relax_to_specified_wind" and not v:
                    continue
                if k == "local_heating_option" and str(v).strip() == "Isidoro":      # the analytic heat source (hs_forcing.F90:728-769); its parameters below
                    kw["local_heating_option"] = 1
                    continue
                if k == "local_heating_option":       # 'from_file' (interpolator_mod's data) or anything else: the reference's own message for an unknown value
                    raise IscaError(f'hs_forcing_nml: "{v}" is not a supported value for local_heating_option (only \'\' and \'Isidoro\')')
                if k == "relax_to_specified_wind":
                    raise IscaError(f"hs_forcing_nml: {k} = {v!r} is not carried by the device core (it needs interpolator_mod's wind files)")
                if k in _HS_LOCAL_HEATING:            # handed on: used when local_heating_option = 'Isidoro', without effect otherwise
                    kw[k] = float(v)
                    continue
                if k in _HS_UNUSED_PARAMETERS:        # values of the branches above, without effect while those are off
                    continue
            if k in ("days", "hours", "minutes", "seconds", "calendar", "current_date", "print_interval", "num_steps", "json_logging",
                     "graceful_shutdown", "ocean_topog_smoothing"):
                continue
            if k == "p00":
                k = "P00"
            if isinstance(v, bool):
                v = int(v)
            if isinstance(v, (list, tuple)) and len(v) == 1:       # e.g. initial_sphum = [2.e-6] (one value per tracer)
                v = v[0]
            kw[k] = tuple(v) if isinstance(v, list) else v
    kw.update(overrides)
    if no_forcing:      # zero coefficients give exactly zero tendencies in the fused forcing (0 * finite); no tracer source or sink for any entry
        kw.update(ka=0.0, ks=0.0, kf=0.0, trflux=0.0, trsink=0.0, local_heating_option=0)
        for k in ("tracer_sms", "tracer_flux", "tracer_sink"):
            kw.pop(k, None)
    return dyncore.default_config(resolution, **kw)


# ---------------------------------------------------------------- atmosphere_mod
def gaussian_topog(namelist: dict, deg_lon, deg_lat):
    """gaussian_topog_init (shared/topography/gaussian_topog.F90:135-160, :215-259): the sum of the gaussian_topog_nml mountains, in m,
    on the [lat, lon] grid."""
    nml = {k.lower(): v for k, v in (namelist.get("gaussian_topog_nml") or {}).items()}
    def arr(name):
        v = nml.get(name, [])
        return [float(x) for x in (v if isinstance(v, (list, tuple)) else [v])]
    height = arr("height")
    lon, lat = np.asarray(deg_lon) * np.pi / 180.0, np.asarray(deg_lat) * np.pi / 180.0
    z = np.zeros((lat.size, lon.size))
    tpi = 2.0 * np.pi
    dtr = tpi / 360.0
    for n, hgt in enumerate(height):
        if hgt == 0.0:
            continue
        par = lambda name: (arr(name)[n] if n < len(arr(name)) else 0.0) * dtr     # namelist arrays default to 0 (:86-92)
        olon, olat, wlon, wlat, rlon, rlat = par("olon"), par("olat"), par("wlon"), par("wlat"), par("rlon"), par("rlat")
        dy = np.abs(lat - olat)
        yy = np.maximum(0.0, dy - rlat) / wlat
        dx = np.abs(lon - olon)
        dx = np.minimum(dx, np.abs(dx - tpi))
        xx = np.maximum(0.0, dx - rlon) / wlon
        z += hgt * np.exp(-xx[None, :] ** 2 - yy[:, None] ** 2)
    return z


def _get_topography(namelist: dict, surf_height, land_mask=None):
    """get_topography (init/spectral_init_cond.F90:167-308): topography_option 'flat' | 'gaussian' (gaussian_topog_nml) | 'input' (the
    [lat, lon] height field `zsurf` and land mask `land_mask` of INPUT/<topog_file_name>, handed over as `surf_height` / `land_mask`:
    spectrally truncated with ocean_topog_smoothing = 0, else regularised over the ocean -- compute_lambda + regularize of
    topog_regularization_mod, :236-245; the namelist's DEFAULT is 0.93)."""
    nml = {g.lower(): {k.lower(): v for k, v in vals.items()} for g, vals in namelist.items()}
    opt = str(nml.get("spectral_init_cond_nml", {}).get("topography_option", "flat")).lower()
    c = _core
    if opt == "flat":
        if surf_height is not None:
            raise IscaError("surf_height given but topography_option = 'flat'")
        return
    if opt == "gaussian":
        z = gaussian_topog(nml, c.table("deg_lon"), c.table("deg_lat"))
        c.set_surf_geopotential(dyncore.GRAV * z)                                       # no truncation on this branch (:300-303)
    elif opt == "input":
        if surf_height is None:
            raise IscaError("get_topography: topography_option=\"input\" needs the height field (atmosphere_init(..., surf_height=array))")
        smoothing = float(nml.get("spectral_dynamics_nml", {}).get("ocean_topog_smoothing", 0.93))          # spectral_dynamics.F90:184
        c.set_topography(surf_height, land_mask, smoothing)                              # isca_dyn_set_topography: truncation (:231-235) or regularisation (:236-245)
    else:
        raise IscaError(f'"{opt}" is an invalid value for topography_option.')


# ---- field_table (FMS field_manager format; the atmosphere's entries as spectral_dynamics_init reads them, spectral_dynamics.F90:316-409)
def _field_table_fields(line: str) -> list[str]:
    """The comma-separated fields of one field_table line; a comma inside quotes belongs to the field ("flux=2.5e-5, sink=-2.0" is one parameter string)."""
    fields, cur, quote = [], "", None
    for ch in line:
        if quote:
            if ch == quote:
                quote = None
            else:
                cur += ch
        elif ch in "\"'":
            quote = ch
        elif ch == ",":
            fields.append(cur.strip()); cur = ""
        else:
            cur += ch
    fields.append(cur.strip())
    return fields


def parse_field_table(text: str) -> list[dict]:
    """Entries of a field_table: '"TRACER", "atmos_mod", "name"' followed by '"method", "value"[, "parameters"]' lines, closed by '/'.
    Returns one dict per atmos_mod tracer: name and {method: (value, parameters)}; other models' entries are skipped."""
    out = []
    body = "\n".join(ln.split("#", 1)[0] for ln in text.splitlines())
    entries, cur, quote = [], "", None             # an entry ends at a '/' outside quotes ("units", "kg/kg" is a field)
    for ch in body:
        if quote:
            quote = None if ch == quote else quote
        elif ch in "\"'":
            quote = ch
        elif ch == "/":
            entries.append(cur); cur = ""
            continue
        cur += ch
    entries.append(cur)
    for entry in entries:
        rows = [_field_table_fields(ln) for ln in entry.splitlines() if ln.strip()]
        if not rows:
            continue
        head = rows[0]
        if len(head) < 3 or head[0].upper() != "TRACER":
            raise IscaError(f"field_table: entry does not start with \"TRACER\", \"model\", \"name\": {rows[0]}")
        if head[1].lower() not in ("atmos_mod", "atmos"):
            continue
        methods = {r[0].lower(): (r[1] if len(r) > 1 else "", r[2] if len(r) > 2 else "") for r in rows[1:]}
        out.append(dict(name=head[2].lower(), methods=methods))
    return out


def tracers_from_field_table(entries: list[dict], robert_coeff: float | None = None, trflux: float = 1.e-5, trsink: float = -4.0):
    """(config keys, tracer names) for the library.  What the kernels implement is what the reference's own field_tables use: tracer 1 a
    'grid' tracer (van Leer + finite_volume_parabolic, the sphum entry), further tracers either that or 'spectral' with the defaults
    (advect_vert = second_centered; hole_filling = on runs water_borrowing on its tendency); anything else is refused by name rather than run as
    something else."""
    if len(entries) > dyncore.MAX_TRACERS:
        raise IscaError(f"field_table: {len(entries)} tracers, at most {dyncore.MAX_TRACERS} are carried")
    # The reference finds the humidity tracer by NAME (nhum = get_tracer_index('sphum') or 'mix_rat', spectral_dynamics.F90:316-332;
    # dry_model when neither exists); the library's tracer 1 is the one initial_sphum, the water correction, the virtual temperature
    # and the moist physics act on.  So the humidity entry has to come first, and a table without one runs as the dry model.
    hum = [k for k, e in enumerate(entries) if e["name"] in ("sphum", "mix_rat")]
    if hum and hum[0] != 0:
        raise IscaError(f"field_table: the humidity tracer ({entries[hum[0]]['name']}) must be the first atmos_mod entry "
                        f"(the first entry is {entries[0]['name']}): tracer 1 is the one the water correction, virtual temperature and moist physics use")
    spectral, robert, names, holes, sms, vert = [], [], [], [], [], []
    for k, e in enumerate(entries):
        m = e["methods"]
        rep = m.get("numerical_representation", ("spectral", ""))[0].lower()      # default_representation (:145)
        if rep not in ("grid", "spectral"):
            raise IscaError(f"spectral_dynamics_init: {rep} is an invalid numerical_representation")            # :359-360
        adv = m.get("advect_vert", ("second_centered", ""))[0].lower()
        if adv not in ("second_centered", "fourth_centered", "van_leer_linear", "finite_volume_parabolic"):
            raise IscaError(f"spectral_dynamics_init: {adv} is an invalid advect_vert")                           # :406-407
        # the scheme the fused kernels implement for the representation is handed over as -1, any other one by its number (tracer_advect_vert)
        std = "finite_volume_parabolic" if rep == "grid" else "second_centered"
        vert.append(-1 if adv == std else ("second_centered", "fourth_centered", "van_leer_linear", "finite_volume_parabolic").index(adv))
        hole = 1 if rep == "spectral" and m.get("hole_filling", ("off", ""))[0].lower() == "on" else 0    # water_borrowing (:1142); ignored for grid tracers (:364-367)
        if k == 0 and rep != "grid":
            raise IscaError(f"field_table: the first tracer ({e['name']}) must be a grid tracer")
        sms_k = None                                                                # hs_forcing's source and sink for this entry (hs_forcing.F90:251-261)
        if "tracer_sms" in m:
            scheme, params = m["tracer_sms"]
            if scheme.upper() in ("NONE", "OFF"):                                   # `cycle` / flux = sink = 0: no tendency either way
                sms_k = (0.0, 0.0)
            else:
                fl, sk = (None, None)
                for item in params.replace(" ", "").split(","):
                    if item.lower().startswith("flux="):
                        fl = float(item.split("=", 1)[1])
                    elif item.lower().startswith("sink="):
                        sk = float(item.split("=", 1)[1])
                sms_k = (fl, sk)
        sms.append(sms_k)
        rc = -1.0                                                                   # the dynamics' robert_coeff (:347,351)
        if "robert_filter" in m:
            scheme, params = m["robert_filter"]
            if scheme.lower() == "off":
                rc = 0.0                                                            # :341-342
            else:
                for item in params.replace(" ", "").split(","):
                    if item.lower().startswith("robert_coeff="):
                        rc = float(item.split("=", 1)[1])
        dyn_rc = 0.04 if robert_coeff is None else robert_coeff                    # spectral_dynamics_nml's module default (:166)
        if k == 0 and rc >= 0.0 and rc != dyn_rc:
            raise IscaError(f"field_table: tracer {e['name']}: a robert_coeff of its own is only available from the second tracer on")
        spectral.append(1 if rep == "spectral" else 0); robert.append(rc); names.append(e["name"]); holes.append(hole)
    keys = dict(num_tracers=len(entries), tracer_spectral=spectral, tracer_robert_coeff=robert, tracer_hole_filling=holes)
    if any(v >= 0 for v in vert):
        keys["tracer_advect_vert"] = vert
    if any(x is not None for x in sms):        # a parameter the entry leaves out keeps hs_forcing_nml's value (trflux, trsink: the arguments)
        keys.update(tracer_sms=[0 if x is None else 1 for x in sms],
                    tracer_flux=[0.0 if x is None else float(trflux if x[0] is None else x[0]) for x in sms],
                    tracer_sink=[0.0 if x is None else float(trsink if x[1] is None else x[1]) for x in sms])
    if entries and not hum:            # dry_model: no humidity anywhere -- tracer 1 starts at 0 like every other tracer, no virtual temperature
        keys.update(initial_sphum=0.0, use_virtual_temperature=False, _dry_model=True)
    return keys, names


def atmosphere_init(namelist=None, resolution: str | None = None, run_dir: str | None = None, field_table: str | None = None, **overrides):
    """atmosphere.F90:120-272: spectral_dynamics_init + (restart | cold start) + hs_forcing_init.

    `field_table`: the text of the run's field_table (tracer_manager); without it the dry default applies (one grid tracer, sphum).
    With `run_dir` the reference's file protocol applies: `run_dir/INPUT/spectral_dynamics.res.nc` (+
    `atmosphere.res.nc`) is read when present (atmosphere.F90:197-223, spectral_dynamics.F90:509-575),
    otherwise the model cold-starts; atmosphere_end() then writes `run_dir/RESTART/`."""
    global _core, _run_dir
    if _core is not None:
        return _core                                   # `if(module_is_initialized) return`
    surf_height = overrides.pop("surf_height", None)
    land_mask = overrides.pop("land_mask", None)
    names = None
    if field_table is not None:          # text of the run's field_table (Experiment: field_table_file); None = the dry default, one sphum grid tracer
        nml = parse_namelist(namelist) if isinstance(namelist, str) else (namelist or {})
        sd = {k.lower(): v for k, v in {g.lower(): v for g, v in nml.items()}.get("spectral_dynamics_nml", {}).items()}
        hs = {k.lower(): v for k, v in {g.lower(): v for g, v in nml.items()}.get("hs_forcing_nml", {}).items()}
        keys, names = tracers_from_field_table(parse_field_table(field_table), sd.get("robert_coeff"),
                                               overrides.get("trflux", hs.get("trflux", 1.e-5)), overrides.get("trsink", hs.get("trsink", -4.0)))
        if keys.pop("_dry_model", False):
            # initialize_corrections / compute_corrections (spectral_dynamics.F90:1245-1248, 1328-1331)
            if sd.get("do_water_correction", True) and overrides.get("do_water_correction", True):
                raise IscaError("compute_corrections: do_water_correction must be .false. in a dry model (default is .true.)")
            at = {k.lower(): v for k, v in {g.lower(): v for g, v in nml.items()}.get("atmosphere_nml", {}).items()}
            if at.get("idealized_moist_model", False):
                raise IscaError("idealized_moist_phys: the field_table has no specific-humidity tracer (sphum / mix_rat)")
            overrides = {**overrides, "initial_sphum": 0.0, "use_virtual_temperature": False}
        overrides = {**keys, **overrides}
    _core = dyncore.DynCore(config_from_namelist(namelist, resolution, **overrides))
    if names:
        _core.tracer_names = names
    _run_dir = run_dir
    try:
        _get_topography(parse_namelist(namelist) if isinstance(namelist, str) else (namelist or {}), surf_height, land_mask)
    except Exception:
        _core.close()
        _core = None
        raise
    _setup_progress_log(parse_namelist(namelist) if isinstance(namelist, str) else (namelist or {}))
    inp = None if run_dir is None else os.path.join(run_dir, "INPUT")
    try:
        if inp is not None and restart.restart_exists(inp):
            restart.read_restart(_core, inp)
            _read_model_time(inp)
        else:
            _core.cold_start()
        _clock["step0"] = _core.info("step")                # 0 after a cold start, 0 or 1 after a restart
    except Exception:
        _core.close()
        _core = None
        raise
    return _core


# ---- the progress line of spectral_diagnostics -> global_integrals (spectral_dynamics.F90:1836-1840, 1869-1912): every
# print_interval (spectral_dynamics_nml, (days, seconds), default one day) rank 0 prints "Integration completed through ..." or, with
# json_logging, the machine-readable line the harness's progress bar parses ({"day":, "second":, "max_speed":, "avg_T":}).
_progress: dict | None = None      # the printer (print_interval / json_logging asked for), or None
_clock: dict | None = None         # the model time of this run: calendar, the date it began at, the step count it began at -- always kept


_CALENDAR_ALIASES = {"none": "no_calendar", "thirty_day_months": "thirty_day", "no_leap": "noleap"}


def _setup_progress_log(namelist: dict):
    global _progress, _clock
    nml = {g.lower(): {k.lower(): v for k, v in vals.items()} for g, vals in namelist.items()}
    sd, mn = nml.get("spectral_dynamics_nml", {}), nml.get("main_nml", {})
    cal = str(mn.get("calendar", "no_calendar")).lower()
    cal = _CALENDAR_ALIASES.get(cal, cal)
    if cal not in _CALENDAR_TYPES:
        raise IscaError(f"main_nml: calendar = '{cal}' is not a supported value (no_calendar, thirty_day, julian, gregorian, noleap)")
    date = list(mn.get("current_date", [0, 0, 0, 0, 0, 0]))
    if cal == "no_calendar":                 # atmos_model.F90:218-223: date(1:2) = 0, date(3:6) = current_time
        date = [0, 0] + (list(mn.get("current_time", [0, 0, 0, 0])) + [0] * 4)[:4]
    _clock = {"calendar": cal, "date0": date, "step0": 0}
    if "print_interval" not in sd and not sd.get("json_logging", False):
        _progress = None                                   # library use without a namelist request: silent
        return
    pi = sd.get("print_interval", [1, 0])
    pi = list(pi) if isinstance(pi, (list, tuple)) else [pi, 0]
    every = (pi[0] * 86400 + (pi[1] if len(pi) > 1 else 0)) / float(_core.cfg.dt_atmos)
    if every < 1 or every != int(every):
        raise IscaError("spectral_dynamics_nml: print_interval must be a positive multiple of dt_atmos")
    _progress = {"every": int(every), "json": bool(sd.get("json_logging", False)), "out": sys.stdout}


# ---- the model time across run segments: the main program's RESTART/atmos_model.res (atmos_model.F90:198-202, 397-406) holds the date the
# run ended at and the calendar type; a run that finds INPUT/atmos_model.res continues from that date AND calendar (the file overrides
# main_nml, atmos_model.F90:198-202), whether or not a progress line was asked for.
_CALENDAR_TYPES = {"no_calendar": 0, "thirty_day": 1, "julian": 2, "gregorian": 3, "noleap": 4}
_CALENDAR_NAMES = {v: k for k, v in _CALENDAR_TYPES.items()}
_MONTH_DAYS = (31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31)


def _is_leap(calendar: str, year: int) -> bool:
    """time_manager.F90: julian -- every fourth year; gregorian -- the 100 / 400 year exceptions; noleap -- never"""
    if calendar == "julian":
        return year % 4 == 0
    if calendar == "gregorian":
        return year % 4 == 0 and (year % 100 != 0 or year % 400 == 0)
    return False


def _date_after(date0, calendar: str, secs: int):
    """date0 (year, month, day, hour, minute, second) advanced by secs.  no_calendar counts days in date(3); thirty_day has 12 months of
    30 days; julian / gregorian / noleap have the real months (with their leap-year rules)."""
    y0, m0, d0, h0, mi0, s0 = (list(date0) + [0] * 6)[:6]
    if calendar == "no_calendar":
        tot = ((d0 * 24 + h0) * 60 + mi0) * 60 + s0 + secs
        days, rem = divmod(tot, 86400)
        return [0, 0, days, rem // 3600, rem % 3600 // 60, rem % 60]
    if calendar == "thirty_day":
        tot = ((((y0 * 12 + max(m0, 1) - 1) * 30 + max(d0, 1) - 1) * 24 + h0) * 60 + mi0) * 60 + s0 + secs
        tot, sec = divmod(tot, 60); tot, mnt = divmod(tot, 60); tot, hr = divmod(tot, 24); tot, dy = divmod(tot, 30); yr, mo = divmod(tot, 12)
        return [yr, mo + 1, dy + 1, hr, mnt, sec]
    yr, mo, dy = y0, max(m0, 1), max(d0, 1)
    days, rem = divmod((h0 * 60 + mi0) * 60 + s0 + secs, 86400)
    dy += days
    while True:
        n = _MONTH_DAYS[mo - 1] + (1 if mo == 2 and _is_leap(calendar, yr) else 0)
        if dy <= n:
            break
        dy -= n
        mo += 1
        if mo > 12:
            mo, yr = 1, yr + 1
    return [yr, mo, dy, rem // 3600, rem % 3600 // 60, rem % 60]


def _read_model_time(inp: str):
    path = os.path.join(inp, "atmos_model.res")
    if _clock is None or not os.path.exists(path):
        return
    with open(path) as f:
        rows = [ln.split() for ln in f.read().splitlines() if ln.strip()]
    _clock["date0"] = [int(x) for x in rows[0][:6]]
    if len(rows) > 1:
        try:
            _clock["calendar"] = _CALENDAR_NAMES[int(rows[1][0])]
        except (KeyError, ValueError):
            raise IscaError(f"{path}: calendar type {rows[1][0]!r} is not one of 0..4")


def _write_model_time(resdir: str):
    if _clock is None:
        return
    secs = int(round((_core.info("step") - _clock["step0"]) * _core.cfg.dt_atmos))
    date = _date_after(_clock["date0"], _clock["calendar"], secs)
    os.makedirs(resdir, exist_ok=True)
    with open(os.path.join(resdir, "atmos_model.res"), "w") as f:
        f.write("%6d%6d%6d%6d%6d%6d        Current model time: year, month, day, hour, minute, second\n" % tuple(date))
        f.write("%6d        (Calendar: no_calendar=0, thirty_day_months=1, julian=2, gregorian=3, noleap=4)\n"
                % _CALENDAR_TYPES[_clock["calendar"]])


def global_integrals():
    """spectral_dynamics.F90:1869-1912: maximum wind speed and area mean of the lowest-level temperature of the current level."""
    c = _need()
    u, v, t = c.get("ug"), c.get("vg"), c.get("tg")
    return float(np.sqrt(u * u + v * v).max()), c.area_weighted_global_mean(t[-1])


def _progress_line():
    c, p, k = _core, _progress, _clock
    secs = int(round((c.info("step") - k["step0"]) * c.cfg.dt_atmos))      # since this run began; date0 is where it began (a restart: where the last one ended)
    max_speed, avg_t = global_integrals()
    date = _date_after(k["date0"], k["calendar"], secs)
    days, rem = date[2], (date[3] * 60 + date[4]) * 60 + date[5]
    if k["calendar"] == "no_calendar":
        if p["json"]:
            line = ' {"day":%6d  ,"second":%6d  ,"max_speed":%13.6E   ,"avg_T":%13.6E   }' % (days, rem, max_speed, avg_t)
        else:
            line = " Integration completed through%6d days%6d seconds" % (days, rem)
    else:                                                   # a calendar with months, from main_nml's current_date
        yr, mo, dy, hr, mnt, sec = date
        mo, dy = mo - 1, dy - 1
        if p["json"]:
            line = ' {"date": "%04d-%02d-%02d", "time": "%02d:%02d:%02d", "max_speed":%6.1f   ,"avg_T":%6.1f   }' % (
                yr, mo + 1, dy + 1, hr, mnt, sec, max_speed, avg_t)
        else:
            names = (" Jan", " Feb", " Mar", " Apr", " May", " Jun", " Jul", " Aug", " Sep", " Oct", " Nov", " Dec")
            line = " Integration completed through%5d%s%3d  %2d:%2d:%2d" % (yr, names[mo], dy + 1, hr, mnt, sec)
    print(line, file=p["out"], flush=True)


def atmosphere(nsteps: int = 1):
    """atmosphere.F90:276-352: one call advances the model by dt_atmos (nsteps calls here)."""
    if _core is None:
        raise IscaError("atmosphere: atmosphere module is not initialized")
    if _progress is None:
        _core.step(nsteps)
        return
    left = nsteps
    while left > 0:                                         # stop at every alarm of print_interval (counted from the start of this run)
        done = _core.info("step") - _clock["step0"]
        chunk = min(left, _progress["every"] - done % _progress["every"])
        _core.step(chunk)
        left -= chunk
        if (done + chunk) % _progress["every"] == 0:
            _progress_line()


def atmosphere_end():
    """atmosphere.F90:358-392: write RESTART/atmosphere.res.nc and RESTART/spectral_dynamics.res.nc, free."""
    global _core, _run_dir
    if _core is not None:
        try:
            if _run_dir is not None:
                restart.write_restart(_core, os.path.join(_run_dir, "RESTART"))
                _write_model_time(os.path.join(_run_dir, "RESTART"))
        finally:
            _core.close()
            _core = None
            _run_dir = None


def _need():
    if _core is None:
        raise IscaError("spectral_dynamics has not been initialized")
    return _core


# ---------------------------------------------------------------- spectral_dynamics_mod getters
def spectral_dynamics_init(namelist=None, resolution=None, **overrides):
    return atmosphere_init(namelist, resolution, **overrides)


def get_num_levels():
    return _need().L


def get_pk_bk():
    c = _need()
    return c.table("pk"), c.table("bk")


def get_field(name: str, time_level: str = "current"):
    return _need().get(name, 1 if time_level == "current" else 0)


def get_initial_fields():
    c = _need()
    if c.info("step") != 0:
        raise IscaError("get_initial_fields: This routine may be called only to get the initial values after a cold_start")
    return c.get("ug"), c.get("vg"), c.get("tg"), c.get("psg")


# ---------------------------------------------------------------- transforms_mod
def trans_spherical_to_grid(spherical):
    return _need().trans_spherical_to_grid(spherical)


def trans_grid_to_spherical(grid, do_truncation=True):
    return _need().trans_grid_to_spherical(grid, do_truncation)


def vor_div_from_uv_grid(u_grid, v_grid):
    return _need().vor_div_from_uv_grid(u_grid, v_grid)


def uv_grid_from_vor_div(vor_spec, div_spec):
    return _need().uv_grid_from_vor_div(vor_spec, div_spec)


def horizontal_advection(field_spec, u_grid, v_grid, tendency):
    return _need().horizontal_advection(field_spec, u_grid, v_grid, tendency)


def area_weighted_global_mean(field):
    return _need().area_weighted_global_mean(field)


def get_deg_lat():
    return _need().table("deg_lat")


def get_deg_lon():
    return _need().table("deg_lon")


def get_sin_lat():
    return _need().table("sin_lat")


def get_wts_lat():
    return _need().table("wts_lat")


# ---------------------------------------------------------------- spherical_mod operators re-exported by transforms_mod
def compute_laplacian(spherical, power=1):
    return _need().compute_laplacian(spherical, power)


def compute_gradient_cos(spherical):
    return _need().compute_gradient_cos(spherical)


def compute_lon_deriv_cos(spherical):
    return _need().compute_lon_deriv_cos(spherical)


def compute_lat_deriv_cos(spherical):
    return _need().compute_lat_deriv_cos(spherical)


def compute_ucos_vcos(vorticity, divergence):
    return _need().compute_ucos_vcos(vorticity, divergence)


def compute_vor_div(u_div_cos, v_div_cos):
    return _need().compute_vor_div(u_div_cos, v_div_cos)


def triangular_truncation(spherical):
    return _need().triangular_truncation(spherical)


def divide_by_cos(grid):
    return _need().divide_by_cos(grid, 1)


def divide_by_cos2(grid):
    return _need().divide_by_cos(grid, 2)


# ---------------------------------------------------------------- press_and_geopot_mod / global_integral_mod / advection
def pressure_variables(surf_p):
    """-> p_half, ln_p_half, p_full, ln_p_full (press_and_geopot.F90:152)"""
    return _need().pressure_variables(surf_p)


def compute_geopotential(t, ln_p_half, ln_p_full):
    return _need().compute_geopotential(t, ln_p_half, ln_p_full)


def mass_weighted_global_integral(field, surf_press):
    return _need().mass_weighted_global_integral(field, surf_press)


def a_grid_horiz_advection(u, v, q, dt, tendency=None):
    return _need().a_grid_horiz_advection(u, v, q, dt, tendency)


def vert_advection_ppm(dt, w, surf_p, r):
    return _need().vert_advection_ppm(dt, w, surf_p, r)
