#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the reference itself (oracle/_ref/ref_harness.x).

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs oracle/_ref built from /root/reference
by oracle/build_ref.py).  The committed .npz files are data: inputs + the reference's outputs.
Usage: python oracle/make_golden.py [--only NAME]
"""
import argparse, os, re, shutil, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
EXE = os.path.join(HERE, "_ref", "ref_harness.x")
GOLD = os.path.join(REPO, "tests", "golden")

# tracer table of the dry model (one grid tracer), a configuration input of the HS test case
FIELD_TABLE = '''"TRACER", "atmos_mod", "sphum"
          "longname",  "specific humidity"
          "units",     "kg/kg"
          "numerical_representation", "grid"
	  "hole_filling",             "off"
          "advect_vert",              "finite_volume_parabolic"
          "robert_filter",            "on"
          "profile_type", "fixed",   "surface_value=0.0" /
'''

# three tracers: sphum as above, a second grid tracer with its own robert_coeff, and a spectral tracer with the defaults of
# spectral_dynamics.F90:145-147 (advect_vert = second_centered, hole_filling = off).  hs_forcing's tracer_source_sink feeds all three.
FIELD_TABLE_3 = FIELD_TABLE + '''"TRACER", "atmos_mod", "age_grid"
          "longname",  "second grid tracer"
          "units",     "none"
          "numerical_representation", "grid"
          "advect_vert",              "finite_volume_parabolic"
          "robert_filter",            "on", "robert_coeff=0.05"
          "profile_type", "fixed",   "surface_value=0.0" /
"TRACER", "atmos_mod", "age_spec"
          "longname",  "spectral tracer"
          "units",     "none"
          "numerical_representation", "spectral"
          "profile_type", "fixed",   "surface_value=0.0" /
'''

# the same three tracers with hole_filling = on for the spectral one: water_borrowing (atmos_spectral/model/water_borrowing.F90) on its tendency
FIELD_TABLE_3_HOLES = FIELD_TABLE_3.replace('"numerical_representation", "spectral"', '"numerical_representation", "spectral"\n          "hole_filling", "on"')

# six tracers (more than the four the library carried before): the three above, a third grid tracer (robert_coeff = 0.08), a spectral one with
# robert_coeff = 0.02 and a spectral one with hole_filling = on -- every entry differs from the others in what update_tracers does with it
FIELD_TABLE_6 = FIELD_TABLE_3 + '''"TRACER", "atmos_mod", "grid_three"
          "numerical_representation", "grid"
          "advect_vert",              "finite_volume_parabolic"
          "robert_filter",            "on", "robert_coeff=0.08"
          "profile_type", "fixed",   "surface_value=0.0" /
"TRACER", "atmos_mod", "spec_two"
          "numerical_representation", "spectral"
          "robert_filter",            "on", "robert_coeff=0.02"
          "profile_type", "fixed",   "surface_value=0.0" /
"TRACER", "atmos_mod", "spec_holes"
          "numerical_representation", "spectral"
          "hole_filling", "on"
          "profile_type", "fixed",   "surface_value=0.0" /
'''

# tracer_sms (hs_forcing.F90:251-261): sphum with a flux and a sink of its own, a grid tracer with only "flux=" (the sink stays hs_forcing_nml's), a
# spectral tracer switched 'off', a spectral one with only "sink=" in seconds, a grid tracer with 'none' (entries 3 and 5 get nothing from hs_forcing
# and stay zero) and one without the method (trflux, trsink)
FIELD_TABLE_SMS = '''"TRACER", "atmos_mod", "sphum"
          "numerical_representation", "grid"
          "advect_vert",              "finite_volume_parabolic"
          "tracer_sms", "on", "flux=2.5e-5, sink=-2.0" /
"TRACER", "atmos_mod", "g_flux"
          "numerical_representation", "grid"
          "advect_vert",              "finite_volume_parabolic"
          "tracer_sms", "on", "flux=4.0e-5" /
"TRACER", "atmos_mod", "s_off"
          "numerical_representation", "spectral"
          "tracer_sms", "off" /
"TRACER", "atmos_mod", "s_sink"
          "numerical_representation", "spectral"
          "tracer_sms", "on", "sink=86400." /
"TRACER", "atmos_mod", "g_none"
          "numerical_representation", "grid"
          "advect_vert",              "finite_volume_parabolic"
          "tracer_sms", "none" /
"TRACER", "atmos_mod", "g_plain"
          "numerical_representation", "grid"
          "advect_vert",              "finite_volume_parabolic" /
'''

# advect_vert per entry (spectral_dynamics.F90:395-408): sphum as always, 'grid' tracers with second_centered (no advect_vert line: the module's default),
# fourth_centered and van_leer_linear, 'spectral' tracers with fourth_centered (current level), van_leer_linear and finite_volume_parabolic (previous level)
FIELD_TABLE_VERT = FIELD_TABLE + '''"TRACER", "atmos_mod", "g_second"
          "numerical_representation", "grid" /
"TRACER", "atmos_mod", "g_fourth"
          "numerical_representation", "grid"
          "advect_vert",              "fourth_centered" /
"TRACER", "atmos_mod", "g_vanleer"
          "numerical_representation", "grid"
          "advect_vert",              "van_leer_linear" /
"TRACER", "atmos_mod", "s_fourth"
          "numerical_representation", "spectral"
          "advect_vert",              "fourth_centered" /
"TRACER", "atmos_mod", "s_vanleer"
          "numerical_representation", "spectral"
          "advect_vert",              "van_leer_linear" /
"TRACER", "atmos_mod", "s_ppm"
          "numerical_representation", "spectral"
          "advect_vert",              "finite_volume_parabolic" /
'''

RES = {"S10": (32, 32, 10, 21), "R10": (32, 32, 10, 11), "T5": (16, 8, 5, 6), "T10": (32, 16, 10, 11), "T21": (64, 32, 21, 22),
       "T31": (96, 48, 31, 32), "T53": (160, 80, 53, 54),        # lon_max = 2^5 3 and 2^5 5: the radix-3 and radix-5 passes of fft99 (fft99.F90:876-1228)
       "T42": (128, 64, 42, 43), "T85": (256, 128, 85, 86), "T170": (512, 256, 170, 171)}


def input_nml(res, num_levels, extra="", extra_groups="", hs_extra=""):
    lon, lat, nf, ns = RES[res]
    # namelist of exp/test_cases/held_suarez/held_suarez_test_case.py:45-98
    return f""" &atmosphere_nml
    idealized_moist_model = .false.
 /
 &spectral_dynamics_nml
    damping_order = 4,
    water_correction_limit = 200.e2,
    reference_sea_level_press = 1.0e5,
    valid_range_t = 100., 800.,
    initial_sphum = 0.0,
    vert_coord_option = 'uneven_sigma',
    scale_heights = 6.0,
    exponent = 7.5,
    surf_res = 0.5,
    lon_max = {lon}, lat_max = {lat}, num_fourier = {nf}, num_spherical = {ns}, num_levels = {num_levels}
    {extra}
 /
 &hs_forcing_nml
    t_zero = 315., t_strat = 200., delh = 60., delv = 10., eps = 0., sigma_b = 0.7,
    ka = -40., ks = -4., kf = -1., do_conserve_energy = .true.
    {hs_extra}
 /
 &diag_manager_nml
    mix_snapshot_average_fields = .false.
 /
 &fms_nml
    domains_stack_size = 2000000
 /
{extra_groups}
"""


FRIERSON_BK = [0.000000, 0.0117665, 0.0196679, 0.0315244, 0.0485411, 0.0719344, 0.1027829, 0.1418581, 0.1894648, 0.2453219,
               0.3085103, 0.3775033, 0.4502789, 0.5244989, 0.5977253, 0.6676441, 0.7322627, 0.7900587, 0.8400683, 0.8819111,
               0.9157609, 0.9422770, 0.9625127, 0.9778177, 0.9897489, 1.0000000]
MOIST_EXE = os.path.join(HERE, "_ref", "ref_moist_harness.x")


def moist_input_nml(res, num_levels=25, extra=""):
    """namelist of exp/test_cases/frierson/frierson_test_case.py:49-170 (config 3 of BASELINE.json): its 25 levels from
    vert_coordinate_nml, or (num_levels != 25: the T85L40 size of BASELINE configs[3]) `uneven_sigma` levels with the test case's
    scale_heights / exponent / surf_res"""
    lon, lat, nf, ns = RES[res]
    bk = ", ".join(f"{b:.7f}" for b in FRIERSON_BK)
    pk = ", ".join("0.0" for _ in FRIERSON_BK)
    vco = "input" if num_levels == 25 else "uneven_sigma"
    return f""" &atmosphere_nml
    idealized_moist_model = .true.
 /
 &idealized_moist_phys_nml
    do_damping = .true., turb = .true., mixed_layer_bc = .true., do_virtual = .false., do_simple = .true.,
    roughness_mom = 3.21e-05, roughness_heat = 3.21e-05, roughness_moist = 3.21e-05,
    two_stream_gray = .true., convection_scheme = 'SIMPLE_BETTS_MILLER'
 /
 &vert_turb_driver_nml
    do_mellor_yamada = .false., do_diffusivity = .true., do_simple = .true., constant_gust = 0.0, use_tau = .false.
 /
 &diffusivity_nml
    do_entrain = .false., do_simple = .true.
 /
 &surface_flux_nml
    use_virtual_temp = .false., do_simple = .true., old_dtaudv = .true.
 /
 &mixed_layer_nml
    tconst = 285., prescribe_initial_dist = .true., evaporation = .true., depth = 2.5, albedo_value = 0.31
 /
 &qe_moist_convection_nml
    rhbm = 0.7, Tmin = 160., Tmax = 350.
 /
 &betts_miller_nml
    rhbm = .7, do_simp = .false., do_shallower = .true.
 /
 &lscale_cond_nml
    do_simple = .true., do_evap = .true.
 /
 &sat_vapor_pres_nml
    do_simple = .true.
 /
 &damping_driver_nml
    do_rayleigh = .true., trayfric = -0.25, sponge_pbottom = 5000., do_conserve_energy = .true.
 /
 &two_stream_gray_rad_nml
    rad_scheme = 'frierson', do_seasonal = .false., atm_abs = 0.2
 /
 &diag_manager_nml
    mix_snapshot_average_fields = .false.
 /
 &fms_nml
    domains_stack_size = 2000000
 /
 &spectral_dynamics_nml
    damping_order = 4, water_correction_limit = 200.e2, reference_sea_level_press = 1.0e5, num_levels = {num_levels},
    valid_range_t = 100., 800., initial_sphum = 2.e-6, vert_coord_option = '{vco}', surf_res = 0.5,
    scale_heights = 11.0, exponent = 7.0, robert_coeff = 0.03,
    lon_max = {lon}, lat_max = {lat}, num_fourier = {nf}, num_spherical = {ns}{extra}
 /
 &vert_coordinate_nml
    bk = {bk},
    pk = {pk}
 /
"""


def prepare_moist_rundir(d, res, nsteps, dt=720, dump_steps=(), phys_steps=(), mode="run", num_levels=25, extra="", harness_extra=""):
    os.makedirs(os.path.join(d, "INPUT"), exist_ok=True)
    os.makedirs(os.path.join(d, "RESTART"), exist_ok=True)
    open(os.path.join(d, "input.nml"), "w").write(moist_input_nml(res, num_levels, (",\n    " + extra) if extra else ""))
    open(os.path.join(d, "field_table"), "w").write(FIELD_TABLE)       # src/extra/model/isca/field_table: the same sphum entry
    open(os.path.join(d, "diag_table"), "w").write("isca_ref_harness\n0 0 0 0 0 0\n")
    fmt = lambda t: ", ".join(str(s) for s in t) if t else "-1"
    open(os.path.join(d, "harness.nml"), "w").write(
        f" &harness_nml\n   mode = '{mode}', nsteps = {nsteps}, dt_atmos = {int(dt)}, dump_steps = {fmt(dump_steps)}, phys_steps = {fmt(phys_steps)}{harness_extra}\n /\n")


def prepare_rundir(d, res, num_levels, mode, nsteps=1, dt=600, dump_steps=(), extra="", extra_groups="", field_table=FIELD_TABLE, hs_extra="", harness_extra=""):
    os.makedirs(os.path.join(d, "INPUT"), exist_ok=True)
    os.makedirs(os.path.join(d, "RESTART"), exist_ok=True)
    open(os.path.join(d, "input.nml"), "w").write(input_nml(res, num_levels, extra, extra_groups, hs_extra))
    open(os.path.join(d, "field_table"), "w").write(field_table)
    open(os.path.join(d, "diag_table"), "w").write("isca_ref_harness\n0 0 0 0 0 0\n")
    ds = ", ".join(str(s) for s in dump_steps) if dump_steps else "-1"
    open(os.path.join(d, "harness.nml"), "w").write(
        f" &harness_nml\n   mode = '{mode}', nsteps = {nsteps}, dt_atmos = {int(dt)}, dump_steps = {ds}{harness_extra}\n /\n")


def run_harness(d, exe=EXE, timeout=3600):
    r = subprocess.run(f"ulimit -s unlimited; exec {exe}", shell=True, cwd=d, capture_output=True,
                       text=True, timeout=timeout, executable="/bin/bash")
    if r.returncode != 0:
        raise RuntimeError("ref harness failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
    return r.stdout


def shapes(res, L):
    lon, lat, nf, ns = RES[res]
    return dict(grid=(L, lat, lon), grid2=(lat, lon), spec=(L, ns + 1, nf + 1), spec2=(ns + 1, nf + 1),
                gridh=(L + 1, lat, lon))


def read_outputs(d, res, L):
    sh = shapes(res, L)
    lon, lat, nf, ns = RES[res]
    out = {}
    for fn in sorted(os.listdir(d)):
        if not fn.endswith(".bin") or fn.startswith("in_"):
            continue
        name = fn[:-4]
        raw = np.fromfile(os.path.join(d, fn), dtype=np.float64)
        n = raw.size
        cands = [("spec", True), ("spec2", True), ("grid", False), ("grid2", False), ("gridh", False)]
        arr = None
        if name == "tab_legendre":
            arr = raw.reshape(lat // 2, ns + 1, nf + 1)
        elif name == "tab_eigen_laplacian":
            arr = raw.reshape(ns + 1, nf + 1)
        elif name == "out_s2f_a":
            arr = raw.view(np.complex128).reshape(L, lat, nf + 1)
        elif name == "out_g2f_a":
            arr = raw.view(np.complex128).reshape(L, lat, lon // 2 + 1)
        else:
            is_spec = bool(re.search(r"(vors|divs|_ts_|lnps|g2s|vor_from|div_from|laplacian|gradcos|ucos|vcos|impl|damp|leap)", name))
            for key, cplx in cands:
                if cplx != is_spec:
                    continue
                cnt = int(np.prod(sh[key])) * (2 if cplx else 1)
                if cnt == n:
                    arr = (raw.view(np.complex128) if cplx else raw).reshape(sh[key])
                    break
            if arr is None:
                arr = raw
        out[name] = arr
    return out


def rand_spec(rng, res, L, two_d=False):
    """Band-limited random coefficients: N(0,1)+iN(0,1), imag(m=0)=0, /(1+L)^2, truncated."""
    lon, lat, nf, ns = RES[res]
    shp = (ns + 1, nf + 1) if two_d else (L, ns + 1, nf + 1)
    s = rng.standard_normal(shp) + 1j * rng.standard_normal(shp)
    s[..., 0] = s[..., 0].real
    m = np.arange(nf + 1)[None, :]; n = np.arange(ns + 1)[:, None]
    tot = m + n
    s = s / (1.0 + tot) ** 2
    s = s * (tot <= ns - 1)
    return s


def golden_kernels(res, L, seed, extra_groups="", keep=None):
    rng = np.random.default_rng(seed)
    lon, lat, nf, ns = RES[res]
    sh = shapes(res, L)
    inp = {}
    for nm in "abcdef":
        inp[f"in_spec_{nm}"] = rand_spec(rng, res, L)
    for nm in "abc":
        inp[f"in_spec2_{nm}"] = rand_spec(rng, res, L, two_d=True)
    inp["in_grid_a"] = 10.0 * rng.standard_normal(sh["grid"])
    inp["in_grid_b"] = 10.0 * rng.standard_normal(sh["grid"])
    inp["in_ps"] = 1.0e5 * (1.0 + 0.03 * rng.standard_normal(sh["grid2"]))
    inp["in_temp"] = 260.0 + 20.0 * rng.standard_normal(sh["grid"])
    wg = 0.05 * rng.standard_normal(sh["gridh"]); wg[0] = 0; wg[-1] = 0
    inp["in_wg"] = wg
    inp["in_q"] = np.abs(1.0e-3 * (1.0 + rng.standard_normal(sh["grid"])))          # tracer-like positive field
    with tempfile.TemporaryDirectory(prefix="refk_") as d:
        prepare_rundir(d, res, L, "kernels", dt=600, extra_groups=extra_groups)
        for k, v in inp.items():
            np.ascontiguousarray(v).tofile(os.path.join(d, k + ".bin"))
        run_harness(d)
        out = read_outputs(d, res, L)
    meta = dict(res=res, num_levels=L, dt_atmos=600.0, seed=seed)
    out = {**{k: v for k, v in inp.items()}, **out}
    if keep is not None:
        out = {k: v for k, v in out.items() if keep(k)}
    return {**out, **{"meta_" + k: np.array(v) for k, v in meta.items()}}


def golden_moist_kernels(res="T21", L=25, nsteps=2400, dt=720, stride=13):
    """Column routines of the Frierson physics chain on a spun-up moist state (harness mode 'kernels' after `nsteps` steps
    of the reference moist model): inputs and outputs of every routine, kept for every `stride`-th column as [lev][col].
    MOIST_KERNELS_DIR=<dir> reuses an existing harness run directory instead of running the reference again (~2 min)."""
    lon, lat, nf, ns = RES[res]
    nc = lon * lat
    reuse = os.environ.get("MOIST_KERNELS_DIR")
    tmp = None
    if reuse:
        d = reuse
    else:
        tmp = tempfile.TemporaryDirectory(prefix="refmk_")
        d = tmp.name
        prepare_moist_rundir(d, res, nsteps, dt=dt, mode="kernels")
        run_harness(d, exe=MOIST_EXE)
    cols = np.arange(0, nc, stride)
    shallow = np.flatnonzero(np.fromfile(os.path.join(d, "k_conv_flag.bin")) == 1.0)[:24]       # + some shallow-convection columns
    cols = np.union1d(cols, shallow)
    out = {"cols": cols, "meta_res": np.array(res), "meta_num_levels": np.array(L), "meta_dt_atmos": np.array(float(dt)),
           "meta_nsteps": np.array(nsteps)}
    for fn in sorted(os.listdir(d)):
        if not fn.endswith(".bin"):
            continue
        raw = np.fromfile(os.path.join(d, fn), dtype=np.float64)
        name = fn[:-4]
        if name.startswith("tab_") or raw.size % nc:
            out[name] = raw
        else:
            out[name] = np.ascontiguousarray(raw.reshape(-1, nc)[:, cols]) if raw.size > nc else raw[cols].copy()
    out["lat_of_col"] = (out["tab_deg_lat"] * np.pi / 180.0)[cols // lon]          # rad_lat as the harness builds it
    if tmp:
        tmp.cleanup()
    return out


def golden_moist_run(res="T21", L=25, nsteps=144, dump_steps=(1, 2, 10, 144), dt=720, keep=None, extra=""):
    """The reference moist model (Frierson physics) from its cold start: grid state u, v, T, q, ps at `dump_steps`."""
    lon, lat, nf, ns = RES[res]
    with tempfile.TemporaryDirectory(prefix="refmr_") as d:
        prepare_moist_rundir(d, res, nsteps, dt=dt, dump_steps=dump_steps, num_levels=L, extra=extra)
        stdout = run_harness(d, exe=MOIST_EXE)
        out = {}
        for fn in sorted(os.listdir(d)):
            if fn.startswith("st_") and fn.endswith(".bin") and (keep is None or keep(fn[:-4])):
                raw = np.fromfile(os.path.join(d, fn))
                out[fn[:-4]] = raw.reshape((L, lat, lon) if raw.size == L * lat * lon else (lat, lon))
            elif fn.startswith("tab_") and fn.endswith(".bin"):
                out[fn[:-4]] = np.fromfile(os.path.join(d, fn))
    m = re.search(r"REF_STATE Tmin,Tmax,maxabsU,qmax=\s*(\S+)\s+(\S+)\s+(\S+)\s+(\S+)", stdout)
    out["final_Tmin_Tmax_maxabsU_qmax"] = np.array([float(x) for x in m.groups()])
    out.update({"meta_res": np.array(res), "meta_num_levels": np.array(L), "meta_dt_atmos": np.array(float(dt)), "meta_nsteps": np.array(nsteps)})
    return out


def golden_moist_developed(res="T42", L=25, day=30, dt=720, more=(1, 10), track=24):
    """A DEVELOPED state of the reference's MOIST model (configs[3]'s physics at T42L25) and the steps after it: `day` days from the cold
    start -- it rains, the convection scheme's deep / shallow branches are taken in most tropical columns --, then BOTH time levels
    (ref_moist_harness.F90: track_from / dump_full_at, the Robert-filtered levels followed through the public API), the mixed layer's
    t_surf (read through oracle/ref_peek.c) and `more` further steps.  Inputs: the full two-level state; outputs: strided samples + extremes."""
    lon, lat, nf, ns = RES[res]
    n0 = day * 86400 // dt
    with tempfile.TemporaryDirectory(prefix="refmdev_") as d:
        prepare_moist_rundir(d, res, n0 + max(more), dt=dt, dump_steps=[n0 + m for m in more], num_levels=L,
                             harness_extra=f", track_from = {n0 - track}, dump_full_at = {n0}, robert_coeff = 0.03")
        stdout = run_harness(d, exe=MOIST_EXE, timeout=6 * 3600)
        out = {}
        sh = shapes(res, L)
        for fn in sorted(os.listdir(d)):
            if not fn.endswith(".bin") or not (fn.startswith("rs_") or fn.startswith("st_")):
                continue
            name, raw = fn[:-4], np.fromfile(os.path.join(d, fn))
            if name.startswith("rs_"):
                if re.match(r"rs_(vors|divs|ts)_", name):
                    out[name] = raw.view(np.complex128).reshape(sh["spec"])
                elif name.startswith("rs_lnps"):
                    out[name] = raw.view(np.complex128).reshape(sh["spec2"])
                else:
                    out[name] = raw.reshape(sh["grid"] if raw.size == np.prod(sh["grid"]) else sh["grid2"])
            elif re.match(r"st_(ug|vg|tg|psg|q)_", name) and int(name[-6:]) > n0:
                a3 = raw.reshape(sh["grid"] if raw.size == np.prod(sh["grid"]) else sh["grid2"])
                key = "after%d_%s" % (int(name[-6:]) - n0, name[3:-7])
                out[key + ("_s222" if a3.ndim == 3 else "")] = np.ascontiguousarray(a3[::2, ::2, ::2] if a3.ndim == 3 else a3)
                out[key + "_minmax"] = np.array([a3.min(), a3.max()])
            elif fn.startswith("tab_"):
                pass
        for fn in ("tab_pk.bin", "tab_bk.bin"):
            out[fn[:-4]] = np.fromfile(os.path.join(d, fn))
    m = re.search(r"REF_DEVELOPED max\|u\|,max\|v\|,Tmin,Tmax,qmax=\s*(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)", stdout)
    out["developed_maxu_maxv_Tmin_Tmax_qmax"] = np.array([float(x) for x in m.groups()])
    out.update({"meta_res": np.array(res), "meta_num_levels": np.array(L), "meta_dt_atmos": np.array(float(dt)), "meta_step0": np.array(n0),
                "meta_more": np.array(more)})
    return out


SHALLOW_EXE = os.path.join(HERE, "_ref", "ref_shallow_harness.x")
SHALLOW_NML = """ &shallow_dynamics_nml
    num_lon = {lon}, num_lat = {lat}, num_fourier = {nf}, num_spherical = {ns},
    add_initial_vortex_pair = .true., u_upper_mag_init = 10.0, u_deep_mag = 5.0
 /
 &shallow_physics_nml
 /
 &fms_nml
    domains_stack_size = 600000
 /
 &diag_manager_nml
    mix_snapshot_average_fields = .false.
 /
"""


BAROTROPIC_EXE = os.path.join(HERE, "_ref", "ref_barotropic_harness.x")
BAROTROPIC_NML = """ &barotropic_dynamics_nml
    num_lon = {lon}, num_lat = {lat}, num_fourier = {nf}, num_spherical = {ns}
 /
 &fms_nml
    domains_stack_size = 600000
 /
 &diag_manager_nml
    mix_snapshot_average_fields = .false.
 /
"""


STIRRING_NML = """ &stirring_nml
    decay_time = 172800, amplitude = {amp}, lat0 = 45., lon0 = 180., widthy = 12., widthx = 45., B = 1.0
 /
"""


def golden_shallow_run(res="T21", nsteps=200, dump_steps=(1, 2, 10, 200), dt=1200, keep=None, nml=None, exe=None, state_re=None,
                       dump_random=False):
    """The reference shallow-water core (src/atmos_spectral_shallow) from its cold start with a vortex pair on a zonal flow over
    the default forcing: grid u, v, vor, div, h, both tracers, stream, pv and the spectral vor, h after `dump_steps`."""
    lon, lat, nf, ns = RES[res]
    with tempfile.TemporaryDirectory(prefix="refsw_") as d:
        os.makedirs(os.path.join(d, "INPUT")); os.makedirs(os.path.join(d, "RESTART"))
        open(os.path.join(d, "input.nml"), "w").write((nml or SHALLOW_NML).format(lon=lon, lat=lat, nf=nf, ns=ns))
        open(os.path.join(d, "field_table"), "w").write("")
        open(os.path.join(d, "diag_table"), "w").write("isca_ref_harness\n0 0 0 0 0 0\n")
        open(os.path.join(d, "harness.nml"), "w").write(
            f" &harness_nml\n   nsteps = {nsteps}, dt_atmos = {int(dt)}, dump_steps = {', '.join(str(s) for s in dump_steps)}"
            f"{', dump_random = .true.' if dump_random else ''}\n /\n")
        stdout = run_harness(d, exe=exe or SHALLOW_EXE)
        out = {}
        for fn in sorted(os.listdir(d)):
            if not fn.endswith(".bin") or (keep is not None and not keep(fn[:-4])):
                continue
            raw = np.fromfile(os.path.join(d, fn))
            name = fn[:-4]
            if raw.size == lon * lat:
                out[name] = raw.reshape(lat, lon)
            elif name == "in_stir_ran":                                   # (m, n, 2) per step -> [step, 2, n, m]
                out[name] = raw.reshape(nsteps, 2, ns + 1, nf + 1)
            elif raw.size == 2 * (nf + 1) * (ns + 1):
                out[name] = raw.view(np.complex128).reshape(ns + 1, nf + 1)
            else:
                out[name] = raw
    m = re.search(state_re or r"REF_STATE hmin,hmax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)", stdout)
    out["final_hmin_hmax_maxabsU" if state_re is None else "final_vormin_vormax_maxabsU"] = np.array([float(x) for x in m.groups()])
    out.update({"meta_res": np.array(res), "meta_dt_atmos": np.array(float(dt)), "meta_nsteps": np.array(nsteps)})
    return out


GAUSSIAN_TOPOG_GROUPS = """ &spectral_init_cond_nml
    topography_option = 'gaussian'
 /
 &gaussian_topog_nml
    height = 2500., 1500., olon = 90., 250., olat = 40., -30., wlon = 25., 20., wlat = 15., 12., rlon = 0., 5., rlat = 0., 3.
 /
"""


HYBRID_BK = [0.0, 0.0, 0.05, 0.15, 0.30, 0.50, 0.70, 0.87, 1.0]
HYBRID_PK = [0.0, 2000.0, 6000.0, 8000.0, 7000.0, 5000.0, 2500.0, 800.0, 0.0]
HYBRID_LEVELS_GROUP = """ &vert_coordinate_nml
    bk = %s,
    pk = %s
 /
""" % (", ".join(str(b) for b in HYBRID_BK), ", ".join(str(p) for p in HYBRID_PK))


def golden_run(res, L, nsteps, dump_steps, dt=600, keep=None, extra="", extra_groups="", field_table=FIELD_TABLE, hs_extra="", harness_extra=""):
    """`extra`: further spectral_dynamics_nml assignments (they follow the test case's own, so they win); `extra_groups`: whole
    namelist groups appended to input.nml"""
    with tempfile.TemporaryDirectory(prefix="refr_") as d:
        prepare_rundir(d, res, L, "run", nsteps=nsteps, dt=dt, dump_steps=dump_steps, extra=extra, extra_groups=extra_groups,
                       field_table=field_table, hs_extra=hs_extra, harness_extra=harness_extra)
        stdout = run_harness(d)
        out = read_outputs(d, res, L)
    if keep is not None:
        out = {k: v for k, v in out.items() if keep(k)}
    m = re.search(r"REF_STATE Tmin,Tmax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)", stdout)
    out["final_Tmin_Tmax_maxabsU"] = np.array([float(x) for x in m.groups()])
    meta = dict(res=res, num_levels=L, dt_atmos=float(dt), nsteps=nsteps, extra=extra + extra_groups)
    out.update({"meta_" + k: np.array(v) for k, v in meta.items()})
    return out


TOPOG_EXE = os.path.join(HERE, "_ref", "ref_topog_harness.x")


def topog_inputs(res, seed=20260930):
    """A synthetic land mask (two continents and an island chain) and height field (mountain ranges on the land, zero over the ocean) in m, [lat, lon]"""
    lon, lat, nf, ns = RES[res]
    rng = np.random.default_rng(seed)
    x = (np.arange(lon) + 0.5) / lon * 360.0
    y = -90.0 + (np.arange(lat) + 0.5) / lat * 180.0
    X, Y = np.meshgrid(x, y)
    land = ((np.abs(X - 90.0) < 40.0) & (np.abs(Y - 35.0) < 30.0)) | ((np.abs(X - 260.0) < 25.0) & (np.abs(Y + 10.0) < 45.0)) | \
           ((np.abs(Y + 60.0) < 8.0) & (np.abs(((X - 150.0) % 40.0) - 20.0) < 6.0))
    h = 3500.0 * np.exp(-((X - 95.0) / 15.0) ** 2 - ((Y - 35.0) / 9.0) ** 2) + 2500.0 * np.exp(-((X - 255.0) / 6.0) ** 2 - ((Y + 15.0) / 30.0) ** 2) + 300.0
    h = h * (1.0 + 0.05 * rng.standard_normal(h.shape))
    return np.where(land, h, 0.0), land.astype(np.float64)


def golden_topog(res="T21", smoothing=0.8):
    """topog_regularization_mod (ocean_topog_smoothing /= 0): compute_lambda + regularize of the reference on a synthetic height field and land mask"""
    height, land = topog_inputs(res)
    with tempfile.TemporaryDirectory(prefix="reft_") as d:
        prepare_rundir(d, res, 8, "run", nsteps=1, dt=600)
        open(os.path.join(d, "topog_harness.nml"), "w").write(f" &topog_harness_nml\n   ocean_topog_smoothing = {smoothing}\n /\n")
        height.tofile(os.path.join(d, "in_height.bin")); land.tofile(os.path.join(d, "in_land.bin"))
        stdout = run_harness(d, exe=TOPOG_EXE)
        smoothed = np.fromfile(os.path.join(d, "out_smoothed.bin")).reshape(height.shape)
        lam, frac = np.fromfile(os.path.join(d, "out_lambda_fraction.bin"))
    print(re.findall(r"Message from subroutine regularize.*", stdout)[-3:])
    return dict(in_height=height, in_land=land, out_smoothed_geopotential=smoothed, out_lambda=np.array(lam), out_fraction_smoothed=np.array(frac),
                meta_res=np.array(res), meta_ocean_topog_smoothing=np.array(smoothing))


def golden_developed(res="T42", L=25, day=60, dt=600, more=(1, 10), track=24):
    """A DEVELOPED state and the steps after it (configs[1] of BASELINE.json, T42L25 Held-Suarez): the reference runs `day` days from
    its cold start -- baroclinic eddies at finite amplitude, Courant numbers, polar rows, the sponge and the fixers all see weather --,
    dumps BOTH time levels (harness.nml dump_full_at; the filtered levels followed through the public API, ref_harness.F90
    track_filter), then `more` further steps.  Inputs: the full two-level state; outputs: strided samples + extremes."""
    lon, lat, nf, ns = RES[res]
    n0 = day * 86400 // dt
    with tempfile.TemporaryDirectory(prefix="refdev_") as d:
        prepare_rundir(d, res, L, "run", nsteps=n0 + max(more), dt=dt, dump_steps=[n0 + m for m in more])
        with open(os.path.join(d, "harness.nml"), "w") as f:
            f.write(f" &harness_nml\n   mode = 'run', nsteps = {n0 + max(more)}, dt_atmos = {int(dt)}, dump_steps = "
                    f"{', '.join(str(n0 + m) for m in more)}, dump_tables = .false., track_from = {n0 - track}, dump_full_at = {n0}\n /\n")
        stdout = run_harness(d, timeout=6 * 3600)
        out = {}
        sh = shapes(res, L)
        for fn in sorted(os.listdir(d)):
            if not fn.endswith(".bin") or not (fn.startswith("rs_") or fn.startswith("st_")):
                continue
            name, raw = fn[:-4], np.fromfile(os.path.join(d, fn))
            if name.startswith("rs_"):
                if re.match(r"rs_(vors|divs|ts)_", name):
                    out[name] = raw.view(np.complex128).reshape(sh["spec"])
                elif name.startswith("rs_lnps"):
                    out[name] = raw.view(np.complex128).reshape(sh["spec2"])
                else:
                    out[name] = raw.reshape(sh["grid"] if raw.size == np.prod(sh["grid"]) else sh["grid2"])
            elif re.match(r"st_(ug|vg|tg|psg|tr1)_", name) and int(name[-6:]) > n0:
                a3 = raw.reshape(sh["grid"] if raw.size == np.prod(sh["grid"]) else sh["grid2"])
                key = "after%d_%s" % (int(name[-6:]) - n0, name[3:-7])
                out[key + ("_s222" if a3.ndim == 3 else "")] = np.ascontiguousarray(a3[::2, ::2, ::2] if a3.ndim == 3 else a3)
                out[key + "_minmax"] = np.array([a3.min(), a3.max()])
    m = re.search(r"REF_DEVELOPED max\|u\|,max\|v\|,Tmin,Tmax=\s*(\S+)\s+(\S+)\s+(\S+)\s+(\S+)", stdout)
    out["developed_maxu_maxv_Tmin_Tmax"] = np.array([float(x) for x in m.groups()])
    out.update({"meta_res": np.array(res), "meta_num_levels": np.array(L), "meta_dt_atmos": np.array(float(dt)), "meta_step0": np.array(n0),
                "meta_more": np.array(more)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    jobs = {
        # small config: every public routine + a 50-step trajectory with full state
        "kernels_T10L8": lambda: golden_kernels("T10", 8, 20260927),
        "run_T10L8": lambda: golden_run("T10", 8, 50, (1, 2, 3, 10, 50)),
        # real resolution of configs[0], few levels, kernel level
        "kernels_T21L6": lambda: golden_kernels("T21", 6, 20260928),
        # configs[0] itself: T21L25 HS, 1 day (144 steps); keep grid state at steps 2 and 144
        "run_T21L25": lambda: golden_run(
            "T21", 25, 144, (2, 144),
            keep=lambda k: k.startswith("tab_") and k != "tab_legendre"
            or re.match(r"st_(ug|vg|tg|psg)_(000002|000144)$", k) is not None or k == "st_tr1_000144"),
        # the trip test's criterion on configs[0] (exp/test_cases/trip_test/trip_test_functions.py:173-189: atmos_daily with ps, bk, pk, ucomp, vcomp, temp,
        # vor, div; :286-297 compares every variable of the file): three daily means of the T21L25 Held-Suarez run, formed as spectral_diagnostics +
        # diag_manager form them (ref_harness.F90: mean_every)
        "trip_T21L25": lambda: golden_run(
            "T21", 25, 432, (), harness_extra=", mean_every = 144",
            keep=lambda k: k in ("tab_pk", "tab_bk") or k.startswith("mean_")),
        # configs[0] for 10 days (1440 steps): the long-run tolerance of SURVEY 8d (1e-7 relative; 1-ulp noise ~2e-10)
        "run_T21L25_10day": lambda: golden_run(
            "T21", 25, 1440, (1440,),
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_001440$", k) is not None),
        # three tracers (grid sphum, a second grid tracer with robert_coeff = 0.05, a spectral tracer): spectral_dynamics.F90:1132-1183
        "run_T21L8_three_tracers": lambda: golden_run(
            "T21", 8, 40, (1, 2, 3, 40), field_table=FIELD_TABLE_3,
            keep=lambda k: re.match(r"st_(ug|tg|psg|tr1|tr2|tr3)_", k) is not None),
        # use_virtual_temperature = .true.: dry core with the hs tracer as q (tiny effect, tight tolerance) and the moist model (q ~ 1e-2)
        "run_T21L8_virtual_t": lambda: golden_run(
            "T21", 8, 60, (2, 60), extra="use_virtual_temperature = .true.",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1|z_full)_", k) is not None),
        "moist_run_T21L25_virtual_t": lambda: golden_moist_run(
            nsteps=40, dump_steps=(1, 2, 10, 40), extra="use_virtual_temperature = .true.",
            keep=lambda k: re.match(r"st_(ug|tg|q|psg)_", k) is not None),
        # rhomboidal truncation (triang_trunc = .false.: every m keeps n = 0..num_spherical-1; 5/2 latitudes per meridional wave)
        "run_R10L8_rhomboidal": lambda: golden_run(
            "R10", 8, 36, (1, 2, 36), extra="triang_trunc = .false.",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1|vors|ts|lnps)_", k) is not None),
        # fourier_inc = 2: zonal wavenumbers 0, 2, .., 20 on a 180-degree sector of 32 longitudes, triangular truncation at 20
        "run_S10L8_fourier_inc2": lambda: golden_run(
            "S10", 8, 36, (1, 2, 36), extra="fourier_inc = 2",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_", k) is not None),
        # hybrid levels (vert_coord_option = 'input' with pk /= 0: pressure levels aloft, sigma at the ground) with the grid tracer
        "run_T21L8_hybrid": lambda: golden_run(
            "T21", 8, 48, (1, 2, 48), extra="vert_coord_option = 'input'", extra_groups=HYBRID_LEVELS_GROUP,
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1|z_full|p_full)_", k) is not None),
        # vert_coord_option = 'hybrid' (compute_vert_coord: uneven-sigma profile blended into pressure levels above p_press): levels + 24 steps
        "run_T21L12_hybrid_option": lambda: golden_run(
            "T21", 12, 24, (24,), extra="vert_coord_option = 'hybrid', p_press = 0.15, p_sigma = 0.45, scale_heights = 5.0, exponent = 3.0, surf_res = 0.3",
            keep=lambda k: k in ("tab_pk", "tab_bk") or re.match(r"st_(ug|tg|psg|tr1)_", k) is not None),
        # tables only (Gauss nodes/weights, Legendre) at T42; T85 kept as a strided sample
        # Frierson column physics (configs[3]'s chain) routine by routine on a spun-up T21L25 moist state
        "moist_kernels_T21L25": golden_moist_kernels,
        # configs[3]'s model at T21L25 from the cold start: 1.2 days with early steps, and 12 days (final state only)
        "moist_run_T21L25": lambda: golden_moist_run(
            keep=lambda k: re.match(r"st_(ug|tg|q|psg)_(000001|000002|000010)$", k) is not None or k.endswith("_000144")),
        "moist_run_T21L25_12day": lambda: golden_moist_run(nsteps=1440, dump_steps=(1440,), keep=lambda k: k.endswith("_001440")),
        # sibling core: shallow water at T21 (early steps + 200) and T42 (final state of 300 steps)
        "shallow_run_T21": golden_shallow_run,
        "shallow_run_T42": lambda: golden_shallow_run("T42", 300, (300,), keep=lambda k: not k.startswith("st_") or re.match(
            r"st_(u|v|h|vor|tr|trs)_000300$", k) is not None),
        # sibling core: barotropic vorticity equation (two jets + wavenumber-4 eddy), T21 early steps + 200, T42 final state
        "barotropic_run_T21": lambda: golden_shallow_run(nml=BAROTROPIC_NML, exe=BAROTROPIC_EXE,
                                                         state_re=r"REF_STATE vormin,vormax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)"),
        "barotropic_run_T42": lambda: golden_shallow_run("T42", 300, (300,), nml=BAROTROPIC_NML, exe=BAROTROPIC_EXE,
                                                         state_re=r"REF_STATE vormin,vormax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)",
                                                         keep=lambda k: not k.startswith("st_") or k.endswith("_000300")),
        # the stirring variants of both sibling test cases: 60 steps from rest / from the default state, with the random numbers the
        # reference's stirring() drew (the Fortran runtime's generator is compiler specific, so they are part of the fixture)
        "barotropic_stirring_T21": lambda: golden_shallow_run(
            "T21", 60, (1, 2, 60), exe=BAROTROPIC_EXE, dump_random=True, state_re=r"REF_STATE vormin,vormax,maxabsU=\s*(\S+)\s+(\S+)\s+(\S+)",
            nml=BAROTROPIC_NML.replace("num_spherical = {ns}", "num_spherical = {ns}, initial_zonal_wind = 'zero', zeta_0 = 0.0")
            + STIRRING_NML.format(amp="3.e-11"), keep=lambda k: k == "in_stir_ran" or re.match(r"st_(u|v|vor|vors)_0000(01|02|60)$", k)),
        "shallow_stirring_T21": lambda: golden_shallow_run(
            "T21", 40, (1, 40), dump_random=True, nml=SHALLOW_NML + STIRRING_NML.format(amp="3.e-12"),
            keep=lambda k: k == "in_stir_ran" or re.match(r"st_(u|v|vor|h|vors)_0000(01|40)$", k)),
        # the shallow-water test case's planet (constants_nml of exp/test_cases/shallow_water/shallow_water_test.py: a giant planet)
        "shallow_run_giant_T21": lambda: golden_shallow_run(
            "T21", 100, (1, 100), nml=SHALLOW_NML + " &constants_nml\n    radius = 55000.e3, omega = 1.6e-4\n /\n",
            keep=lambda k: k.startswith("tab_") or re.match(r"st_(u|v|vor|div|h|tr|trs|vors|hs)_000(001|100)$", k)),
        "tables_T42": lambda: golden_run("T42", 2, 0, (), keep=lambda k: k.startswith("tab_")),
        # options of spectral_damping_init (spectral_damping.F90:124-156) on short T21L8 runs (36 steps): the exponential cut-off filter,
        # whose effective coefficient depends on the step's delta_t, and separate vorticity / divergence coefficients and orders
        "run_T21L8_damping_exponential": lambda: golden_run(
            "T21", 8, 36, (36,), extra="damping_option = 'exponential_cutoff', cutoff_wn = 10, damping_order = 3, damping_coeff = 2.3e-4, "
            "damping_coeff_vor = 1.2e-4, damping_coeff_div = 4.6e-4", keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_000036$", k) is not None),
        "run_T21L8_hole_filling": lambda: golden_run(
            "T21", 8, 60, (1, 2, 3, 40, 60), field_table=FIELD_TABLE_3_HOLES,
            keep=lambda k: re.match(r"st_(ug|tg|psg|tr1|tr2|tr3)_", k) is not None),
        "run_T21L8_six_tracers": lambda: golden_run(
            "T21", 8, 40, (1, 2, 40), field_table=FIELD_TABLE_6,
            keep=lambda k: re.match(r"st_(ug|tg|psg|tr[1-6])_", k) is not None),
        "run_T21L8_tracer_sms": lambda: golden_run(
            "T21", 8, 40, (1, 2, 40), field_table=FIELD_TABLE_SMS,
            keep=lambda k: re.match(r"st_(ug|tg|psg|tr[1-6])_", k) is not None),
        "run_T21L8_tracer_advect_vert": lambda: golden_run(
            "T21", 8, 60, (1, 2, 3, 60), field_table=FIELD_TABLE_VERT,
            keep=lambda k: re.match(r"st_(ug|tg|psg|tr[1-7])_", k) is not None),
        # sphum itself with advect_vert = van_leer_linear (the water correction then acts on a tracer the option kernel advected)
        "run_T21L8_sphum_van_leer": lambda: golden_run(
            "T21", 8, 40, (1, 2, 40), field_table=FIELD_TABLE.replace('"finite_volume_parabolic"', '"van_leer_linear"'),
            keep=lambda k: re.match(r"st_(ug|tg|psg|tr1)_", k) is not None),
        # hs_forcing_nml: no_forcing = .true. (hs_forcing returns at once, hs_forcing.F90:174: no drag, no heating, no tracer source) over the two
        # Gaussian mountains -- the adiabatic adjustment of the isothermal rest state to the orography
        "run_T21L8_no_forcing": lambda: golden_run(
            "T21", 8, 48, (1, 2, 48), extra_groups=GAUSSIAN_TOPOG_GROUPS, hs_extra="no_forcing = .true.",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_", k) is not None),
        # hs_forcing_nml: local_heating_option = 'Isidoro' (hs_forcing.F90:728-769): a Gaussian heat source of 5 K/day at the ground over (120E, 20N),
        # decaying upward with a scale of 300 hPa, on top of the Held-Suarez forcing
        "run_T21L8_isidoro": lambda: golden_run(
            "T21", 8, 48, (1, 2, 48), hs_extra="local_heating_option = 'Isidoro', local_heating_srfamp = 5.0, local_heating_xwidth = 25., "
            "local_heating_ywidth = 12., local_heating_xcenter = 120., local_heating_ycenter = 20., local_heating_vert_decay = 3.e4",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_", k) is not None),
        "run_T21L8_damping_vor_div": lambda: golden_run(
            "T21", 8, 36, (36,), extra="damping_option = 'resolution_dependent', damping_order = 4, damping_coeff_vor = 3.0e-4, damping_order_vor = 2, "
            "damping_coeff_div = 6.0e-4, damping_order_div = 3", keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_000036$", k) is not None),
        # Robert-Asselin-Williams filter (leapfrog.F90:58-105 with raw_filter_coeff /= 1): the new level's spectral state is adjusted
        # AFTER its grid fields have been synthesised (spectral_dynamics.F90:1031)
        "run_T21L8_raw_filter": lambda: golden_run(
            "T21", 8, 36, (2, 3, 36), extra="raw_filter_coeff = 0.7", keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_0000(02|03|36)$", k) is not None),
        # topography: two Gaussian mountains (get_topography 'gaussian', spectral_init_cond.F90:299-303; gaussian_topog.F90:215-259) --
        # initial surface pressure over the orography, surf_geopotential in the hydrostatic integral
        "run_T21L8_topography": lambda: golden_run(
            "T21", 8, 36, (1, 36), extra_groups=GAUSSIAN_TOPOG_GROUPS,
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_0000(01|36)$", k) is not None),
        # vert_advect_uv / vert_advect_t other than second_centered (spectral_dynamics.F90:280-301, 877-888; vert_advection.F90:173-438):
        # fourth-centred for both; van Leer for the winds with the piecewise-parabolic scheme for temperature (both on the previous level)
        "run_T21L8_vadv_fourth": lambda: golden_run(
            "T21", 8, 36, (1, 2, 36), extra="vert_advect_uv = 'fourth_centered', vert_advect_t = 'fourth_centered'",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_0000(01|02|36)$", k) is not None),
        "run_T21L8_vadv_finite_volume": lambda: golden_run(
            "T21", 8, 36, (1, 2, 36), extra="vert_advect_uv = 'van_leer_linear', vert_advect_t = 'finite_volume_parabolic'",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_0000(01|02|36)$", k) is not None),
        "run_T21L8_vadv_ppm_uv": lambda: golden_run(
            "T21", 8, 36, (36,), extra="vert_advect_uv = 'finite_volume_parabolic', vert_advect_t = 'van_leer_linear'",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_000036$", k) is not None),
        # use_implicit = .false. (spectral_dynamics.F90:906): explicit leapfrog of the gravity waves, dt_atmos = 300 s
        "run_T21L8_explicit": lambda: golden_run(
            "T21", 8, 48, (1, 2, 48), dt=300, extra="use_implicit = .false.",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_0000(01|02|48)$", k) is not None),
        # make_symmetric = .true. (spherical.F90:185; exp/test_cases/axisymmetric): every zonal wavenumber m > 0 is truncated away
        "run_T21L8_symmetric": lambda: golden_run(
            "T21", 8, 48, (1, 48), extra="make_symmetric = .true.",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_0000(01|48)$", k) is not None),
        # vert_difference_option = 'mcm' (spectral_dynamics.F90:1084, press_and_geopot.F90:196, 242, implicit.F90:404, 447): mid-point full-level pressures,
        # the old model's four_in_one and linear operator; on the test case's sigma levels and on the 'mcm' vertical coordinate (vert_coordinate.F90:148)
        "run_T21L8_mcm": lambda: golden_run(
            "T21", 8, 48, (1, 2, 48), extra="vert_difference_option = 'mcm'",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1|z_full|p_full)_0000(01|02|48)$", k) is not None),
        "run_T21L14_mcm_coord": lambda: golden_run(
            "T21", 14, 36, (1, 36), extra="vert_difference_option = 'mcm', vert_coord_option = 'mcm'",
            keep=lambda k: k in ("tab_pk", "tab_bk") or re.match(r"st_(ug|vg|tg|psg|tr1|z_full|p_full)_0000(01|36)$", k) is not None),
        # lon_max with factors 3 and 5 (fft99's set99 takes n/2 = 2^a 3^b 5^c): every public routine at T31 (96 x 48), runs at T31 and T53 (160 x 80)
        "kernels_T31L6": lambda: golden_kernels("T31", 6, 20260929),
        # the same harness over the two Gaussian mountains: compute_geopotential with a surface geopotential other than zero (press_and_geopot.F90:331)
        "kernels_T21L6_topography": lambda: golden_kernels("T21", 6, 20260928, extra_groups=GAUSSIAN_TOPOG_GROUPS,
                                                           keep=lambda k: re.match(r"out_(geopot_full|geopot_half)$", k) is not None),      # (inputs: those of kernels_T21L6, same seed)
        "run_T31L8": lambda: golden_run(
            "T31", 8, 36, (1, 2, 36), keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_0000(01|02|36)$", k) is not None),
        "run_T53L8": lambda: golden_run(
            "T53", 8, 36, (1, 36), keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_0000(01|36)$", k) is not None),
        # ocean_topog_smoothing (topog_regularization.F90): the reference's compute_lambda + regularize on a synthetic land mask and height field
        "topog_regularize_T21": lambda: golden_topog("T21", 0.8),
        "topog_regularize_T42": lambda: golden_topog("T42", 0.9),
        "run_T21L8_damping_res_independent": lambda: golden_run(
            "T21", 8, 36, (36,), extra="damping_option = 'resolution_independent', damping_order = 2, damping_coeff = 2.0e16",
            keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_000036$", k) is not None),
    }
    for name, fn in jobs.items():
        if a.only and a.only != name:
            continue
        out = fn()
        path = os.path.join(GOLD, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {len(out)} arrays, {os.path.getsize(path)/1e6:.2f} MB")
    if not a.only or a.only == "run_T42L25":
        # configs[1]: T42L25 HS, 1 day; 3-D fields kept as the [::2, ::2, ::2] sample (every 2nd level, latitude, longitude)
        out = golden_run("T42", 25, 144, (144,), keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_000144$", k) is not None)
        for k in list(out):
            if k.startswith("st_") and out[k].ndim == 3:
                out[k + "_s222"] = np.ascontiguousarray(out.pop(k)[::2, ::2, ::2])
        path = os.path.join(GOLD, "run_T42L25.npz")
        np.savez_compressed(path, **out)
        print(f"run_T42L25: {os.path.getsize(path)/1e6:.2f} MB")
    if not a.only or a.only == "run_T85L40":
        # the benchmark configuration itself (T85L40, dt = 300 s), steps 20 and 288 (one day) from the cold start; 3-D fields kept
        # as the [::4, ::8, ::8] sample (every 4th level, 8th latitude and longitude), ps as [::4, ::4]   (~2 min of reference time)
        out = golden_run("T85", 40, 288, (20, 288), dt=300, keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_000(020|288)$", k) is not None)
        for k in list(out):
            if k.startswith("st_"):
                a3 = out.pop(k)
                out[k + ("_s488" if a3.ndim == 3 else "_s44")] = np.ascontiguousarray(a3[::4, ::8, ::8] if a3.ndim == 3 else a3[::4, ::4])
                # (round 6) EVERY grid point: the zonal sums of the field and of its square per (level, latitude) -- a localized error off the strided
                # sample moves them (a relative error > 1e-7 at one point moves the row's sum of squares by > 1e-9)
                out[k + "_rowsum"] = a3.sum(axis=-1); out[k + "_rowsq"] = (a3 * a3).sum(axis=-1)
        path = os.path.join(GOLD, "run_T85L40.npz")
        np.savez_compressed(path, **out)
        print(f"run_T85L40: {os.path.getsize(path)/1e6:.2f} MB")
    if a.only == "moist_run_T85L40":
        # BASELINE configs[3] at its full size: the Frierson model at T85L40 (uneven_sigma levels, dt = 300 s) from the cold start,
        # steps 1, 12 and 144 (12 hours); 3-D fields kept as the [3::4, ::8, ::8] sample (every 4th level up to the lowest one),
        # 2-D fields as [::4, ::4]   (~4 min of reference time)
        out = golden_moist_run("T85", 40, 144, (1, 12, 144), dt=300, keep=lambda k: re.match(r"st_(ug|vg|tg|q|psg)_000(001|012|144)$", k) is not None)
        for k in list(out):
            if k.startswith("st_"):
                a3 = out.pop(k)
                out[k + ("_s488" if a3.ndim == 3 else "_s44")] = np.ascontiguousarray(a3[3::4, ::8, ::8] if a3.ndim == 3 else a3[::4, ::4])
        out = {k: v for k, v in out.items() if not k.startswith("tab_") or v.size < 4096}
        path = os.path.join(GOLD, "moist_run_T85L40.npz")
        np.savez_compressed(path, **out)
        print(f"moist_run_T85L40: {os.path.getsize(path)/1e6:.2f} MB")
    if a.only == "run_T170L60":
        # BASELINE configs[4] at its full size: T170L60 Held-Suarez, dt = 150 s, steps 1, 8 and 96 (4 hours) from the cold start;
        # 3-D fields kept as the [5::6, ::16, ::16] sample (every 6th level up to the lowest one), ps as [::8, ::8]
        out = golden_run("T170", 60, 96, (1, 8, 96), dt=150, keep=lambda k: re.match(r"st_(ug|vg|tg|psg|tr1)_0000(01|08|96)$", k) is not None)
        for k in list(out):
            if k.startswith("st_"):
                a3 = out.pop(k)
                out[k + ("_s6gg" if a3.ndim == 3 else "_s88")] = np.ascontiguousarray(a3[5::6, ::16, ::16] if a3.ndim == 3 else a3[::8, ::8])
                out[k + "_rowsum"] = a3.sum(axis=-1); out[k + "_rowsq"] = (a3 * a3).sum(axis=-1)       # every grid point (see run_T85L40)
        path = os.path.join(GOLD, "run_T170L60.npz")
        np.savez_compressed(path, **out)
        print(f"run_T170L60: {os.path.getsize(path)/1e6:.2f} MB")
    if a.only and a.only.startswith("moist_developed_"):
        # the moist model's developed-state fixture (~20 min of reference time at T42L25; `--only moist_developed_T21L25_d2` is a 1-minute dry run of the recipe)
        m = re.match(r"moist_developed_(T\d+)L(\d+)(?:_d(\d+))?$", a.only)
        out = golden_moist_developed(m.group(1), int(m.group(2)), day=int(m.group(3) or 30))
        path = os.path.join(GOLD if not m.group(3) else "/tmp", a.only + ".npz")
        np.savez_compressed(path, **out)
        print(f"{a.only}: {os.path.getsize(path)/1e6:.2f} MB,", out["developed_maxu_maxv_Tmin_Tmax_qmax"])
        return
    if a.only == "developed_T42L25":
        # the developed-state restart fixture (~25 min of reference time; 20 MB: the full two-level state is the INPUT)
        out = golden_developed()
        path = os.path.join(GOLD, "developed_T42L25.npz")
        np.savez_compressed(path, **out)
        print(f"developed_T42L25: {os.path.getsize(path)/1e6:.2f} MB,", out["developed_maxu_maxv_Tmin_Tmax"])
    if not a.only or a.only == "tables_T85":
        out = golden_run("T85", 2, 0, (), keep=lambda k: k.startswith("tab_"))
        leg = out.pop("tab_legendre")
        out["tab_legendre_j0_j31_j63"] = leg[[0, 31, 63]]
        path = os.path.join(GOLD, "tables_T85.npz")
        np.savez_compressed(path, **out)
        print(f"tables_T85: {os.path.getsize(path)/1e6:.2f} MB")


if __name__ == "__main__":
    main()
