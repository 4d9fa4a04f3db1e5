"""Frierson column physics (isca_amd/csrc/moist_physics.h), host build, against the reference's own routines.

The fixture tests/golden/moist_kernels_T21L25.npz holds inputs and outputs of every routine of the chain
(sat_vapor_pres lookups, qe_moist_convection, lscale_cond, two_stream_gray_rad, surface_flux, damping_driver,
vert_turb_driver/diffusivity, gcm_vert_diff_down, mixed_layer, gcm_vert_diff_up) called by oracle/ref_moist_harness.F90 on
a spun-up T21L25 moist state of the reference (every 13th column).  The header is compiled for the host without FMA
contraction (oracle/build_moist_host.py), and every routine reproduces the reference to the last bit on this image;
the assertions allow 1e-13 so that a different libm does not fail them.
"""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "moist_kernels_T21L25.npz")
TOL = 1e-13


@pytest.fixture(scope="module")
def lib():
    from oracle.build_moist_host import build
    return ctypes.CDLL(build())


@pytest.fixture(scope="module")
def g():
    z = np.load(GOLD)
    return {k: np.ascontiguousarray(z[k]) for k in z.files}


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


D = ctypes.c_double


def test_sat_vapor_pres_tables(lib, g):
    t = g["k_in_t_prev"].ravel().copy()
    es, des = np.zeros_like(t), np.zeros_like(t)
    lib.mh_lookup_es_des(t.size, P(t), P(es), P(des))
    assert rel(es, g["k_es"].ravel()) <= TOL and rel(des, g["k_des"].ravel()) <= TOL


def test_qe_moist_convection(lib, g):
    L, nc = g["k_in_t_prev"].shape
    out3 = {k: np.zeros((L, nc)) for k in ("dt", "dq", "tref", "qref")}
    out2 = {k: np.zeros(nc) for k in ("rain", "cape", "cin", "flag", "klzb", "klcl")}
    lib.mh_qe_moist_convection(L, nc, D(float(g["k_in_delta_t"][0])), P(g["k_in_t_prev"]), P(g["k_in_q_prev"]), P(g["k_in_p_full_prev"]),
                               P(g["k_in_p_half_prev"]), P(out3["dt"]), P(out3["dq"]), P(out2["rain"]), P(out2["cape"]), P(out2["cin"]),
                               P(out2["flag"]), P(out2["klzb"]), P(out2["klcl"]), P(out3["tref"]), P(out3["qref"]))
    for k in out3:
        assert rel(out3[k], g["k_conv_" + k]) <= TOL, k
    for k in ("rain", "cape", "cin"):
        assert rel(out2[k], g["k_conv_" + k]) <= TOL, k
    for k in ("flag", "klzb", "klcl"):                      # regime and level indices must agree exactly
        assert np.array_equal(out2[k], g["k_conv_" + k]), k
    assert len(np.unique(g["k_conv_flag"])) >= 3            # the sample holds no, shallow and deep convection


def test_lscale_cond(lib, g):
    L, nc = g["k_in_t_prev"].shape
    t = g["k_in_t_prev"] + g["k_conv_dt"]
    for q, tag in ((g["k_in_q_prev"] + g["k_conv_dq"], "k_cond"), (g["k_in_cond2_q"], "k_cond2")):
        q = np.ascontiguousarray(q)
        td, qd, rain = np.zeros((L, nc)), np.zeros((L, nc)), np.zeros(nc)
        lib.mh_lscale_cond(L, nc, P(t), P(q), P(g["k_in_p_full_prev"]), P(g["k_in_p_half_prev"]), P(td), P(qd), P(rain))
        assert rel(td, g[tag + "_dt"]) <= TOL and rel(qd, g[tag + "_dq"]) <= TOL and rel(rain, g[tag + "_rain"]) <= TOL
    assert np.count_nonzero(g["k_cond2_dq"]) > 100 and g["k_cond2_rain"].max() > 0     # condensation and re-evaporation both act


def test_two_stream_gray_rad(lib, g):
    L, nc = g["k_in_t_prev"].shape
    alb = np.full(nc, 0.31)
    nsw, lwd, tdt = np.zeros(nc), np.zeros(nc), np.zeros((L, nc))
    lib.mh_gray_rad(L, nc, D(0.2), P(g["lat_of_col"]), P(alb), P(g["k_in_t_surf"]), P(g["k_in_t_prev"]), P(g["k_in_p_half_cur"]), P(nsw), P(lwd),
                    P(tdt))
    assert rel(nsw, g["k_rad_net_sw_down"]) <= TOL and rel(lwd, g["k_rad_lw_down"]) <= TOL and rel(tdt, g["k_rad_dt"]) <= 1e-12


SF = ["flux_t", "flux_q", "flux_r", "flux_u", "flux_v", "dhdt_surf", "dedt_surf", "dedq_surf", "drdt_surf", "dhdt_atm", "dedq_atm", "dtaudu_atm",
      "dtaudv_atm", "w_atm", "ustar", "bstar", "qstar", "drag_m", "drag_t", "drag_q", "q_surf"]


def test_surface_flux(lib, g):
    L, nc = g["k_in_t_prev"].shape
    low = lambda n: np.ascontiguousarray(g[n][L - 1])
    out = np.zeros((nc, len(SF)))
    lib.mh_surface_flux(nc, P(low("k_in_t_prev")), P(low("k_in_q_prev")), P(low("k_in_u_prev")), P(low("k_in_v_prev")), P(low("k_in_p_full_cur")),
                        P(low("k_in_z_full_cur")), P(np.ascontiguousarray(g["k_in_p_half_cur"][L])), P(g["k_in_t_surf"]), D(3.21e-5), D(1.0), P(out))
    for i, nm in enumerate(SF):
        assert rel(out[:, i], g["k_sf_" + nm]) <= TOL, nm
    assert (g["k_sf_bstar"] > 0).any() and (g["k_sf_bstar"] < 0).any()      # stable and unstable surface layers


def _tendencies_before_damping(g):
    delt = float(g["k_in_delta_t"][0])
    L, nc = g["k_in_t_prev"].shape
    return delt, np.zeros((L, nc)), np.zeros((L, nc)), g["k_conv_dt"] / delt + g["k_rad_dt"], g["k_conv_dq"] / delt


def _nlev_rayfric(g, L):
    ph = g["tab_pk"] + g["tab_bk"] * 101325.0                    # damping_driver_init's pref (idealized_moist_phys.F90:620-629)
    lnph = np.log(np.where(ph > 0, ph, 1.0))
    lnpf = np.array([lnph[k + 1] - 1.0 if ph[k] == 0 else (ph[k + 1] * lnph[k + 1] - ph[k] * lnph[k]) / (ph[k + 1] - ph[k]) - 1.0
                     for k in range(L)])
    pref = np.append(np.exp(lnpf), 101325.0)
    return int(np.argmin(np.abs(pref - 2 * 5000.0))) + 1


def test_damping_turbulence_diffusion_mixed_layer(lib, g):
    L, nc = g["k_in_t_prev"].shape
    delt, du, dv, dt_t, dt_q = _tendencies_before_damping(g)
    u, v, tm, q = g["k_in_u_prev"], g["k_in_v_prev"], g["k_in_t_prev"], g["k_in_q_prev"]
    pf, ph, zf, zh = g["k_in_p_full_cur"], g["k_in_p_half_cur"], g["k_in_z_full_cur"], g["k_in_z_half_cur"]
    lib.mh_rayleigh(L, nc, _nlev_rayfric(g, L), D((1. / 0.25) * (1. / 86400.)), D(5000.), D(delt), P(pf), P(u), P(v), P(du), P(dv), P(dt_t))
    assert rel(du, g["k_damp_dt_u"]) <= TOL and rel(dv, g["k_damp_dt_v"]) <= TOL and rel(dt_t, g["k_damp_dt_t"]) <= TOL
    assert np.abs(g["k_damp_dt_u"]).max() > 0
    h, km, kt = np.zeros(nc), np.zeros((L, nc)), np.zeros((L, nc))
    lib.mh_diffusivity(L, nc, D(delt), P(tm), P(u), P(v), P(dt_t), P(du), P(dv), P(zf), P(zh), P(g["k_sf_ustar"]), P(g["k_sf_bstar"]), P(h), P(km),
                       P(kt))
    assert rel(h, g["k_turb_z_pbl"]) <= TOL and rel(km, g["k_turb_diff_m"]) <= TOL and rel(kt, g["k_turb_diff_t"]) <= TOL
    assert np.abs(g["k_turb_gust"]).max() == 0.0                   # constant_gust = 0 after the first step
    diss, surf, surf_ml, dtd = np.zeros((L, nc)), np.zeros((nc, 7)), np.zeros((nc, 7)), np.zeros((L, nc))
    ts = g["k_in_t_surf"].copy()
    sf = lambda n: g["k_sf_" + n]
    lib.mh_vert_diff(L, nc, D(delt), D(delt / 2), P(u), P(v), P(tm), P(q), P(km), P(kt), P(ph), P(pf), P(zf), P(sf("flux_u")), P(sf("flux_v")),
                     P(sf("dtaudu_atm")), P(sf("dtaudv_atm")), P(du), P(dv), P(dt_t), P(dt_q), P(diss), P(surf), P(ts), P(sf("flux_t")),
                     P(sf("flux_q")), P(sf("flux_r")), P(g["k_rad_net_sw_down"]), P(g["k_rad_lw_down"]), P(sf("dhdt_surf")), P(sf("dedt_surf")),
                     P(sf("drdt_surf")), P(sf("dhdt_atm")), P(sf("dedq_atm")), P(surf_ml), P(dtd))
    for i, nm in enumerate(["dtmass", "dflux_t", "delta_t", "dflux_q", "delta_q"]):
        assert rel(surf[:, i], g["k_vd_" + nm]) <= TOL, nm
    assert rel(dtd, g["k_vd_down_dt_t"]) <= TOL and rel(diss, g["k_vd_diss_heat"]) <= TOL
    assert rel(ts, g["k_ml_t_surf"]) <= TOL and np.abs(ts - g["k_in_t_surf"]).max() > 1e-3
    assert rel(surf_ml[:, 2], g["k_ml_delta_t"]) <= TOL and rel(surf_ml[:, 4], g["k_ml_delta_q"]) <= TOL
    for a, nm in ((du, "u"), (dv, "v"), (dt_t, "t"), (dt_q, "q")):
        assert rel(a, g["k_fin_dt_" + nm]) <= TOL, nm


def test_diffusion_limited_to_the_boundary_layer(lib, g):
    """The device kernel runs the four sweeps of the implicit diffusion over the boundary layer only (from pbl_depth_f's kstop down; the levels above
    take their tendencies in a streaming pass): same results as the sweeps over the whole column, which the test above holds to the reference --
    every value equal (a zero may carry the other sign), on the 182 fixture columns, whose boundary layers are 1 to 9 levels deep."""
    import ctypes as C
    L, nc = g["k_in_t_prev"].shape
    delt, du, dv, dt_t, dt_q = _tendencies_before_damping(g)
    u, v, tm, q = g["k_in_u_prev"], g["k_in_v_prev"], g["k_in_t_prev"], g["k_in_q_prev"]
    pf, ph, zf, zh = g["k_in_p_full_cur"], g["k_in_p_half_cur"], g["k_in_z_full_cur"], g["k_in_z_half_cur"]
    lib.mh_rayleigh(L, nc, _nlev_rayfric(g, L), D((1. / 0.25) * (1. / 86400.)), D(5000.), D(delt), P(pf), P(u), P(v), P(du), P(dv), P(dt_t))
    h, km, kt = np.zeros(nc), np.zeros((L, nc)), np.zeros((L, nc))
    lib.mh_diffusivity(L, nc, D(delt), P(tm), P(u), P(v), P(dt_t), P(du), P(dv), P(zf), P(zh), P(g["k_sf_ustar"]), P(g["k_sf_bstar"]), P(h), P(km),
                       P(kt))
    kstop = np.zeros(nc, dtype=np.int32)
    lib.mh_pbl_kstop(L, nc, D(delt), P(tm), P(u), P(v), P(dt_t), P(du), P(dv), P(zf), P(zh), kstop.ctypes.data_as(C.POINTER(C.c_int)))
    # no diffusivity on any interface at or above kstop (k_m[k], k_t[k] sit on the interface above level k)
    for c in range(nc):
        assert not km[:kstop[c] + 1, c].any() and not kt[:kstop[c] + 1, c].any(), c
    assert kstop.min() >= L - 12 and kstop.max() <= L - 1 and len(set(kstop.tolist())) > 3, sorted(set(kstop.tolist()))
    sf = lambda n: g["k_sf_" + n]
    out = {}
    for mode in ("whole", "limited", "wave"):
        a = [x.copy() for x in (du, dv, dt_t, dt_q)]
        diss, surf, surf_ml, dtd = np.zeros((L, nc)), np.zeros((nc, 7)), np.zeros((nc, 7)), np.zeros((L, nc))
        ts = g["k_in_t_surf"].copy()
        if mode == "whole":
            lib.mh_vert_diff(L, nc, D(delt), D(delt / 2), P(u), P(v), P(tm), P(q), P(km), P(kt), P(ph), P(pf), P(zf), P(sf("flux_u")), P(sf("flux_v")),
                             P(sf("dtaudu_atm")), P(sf("dtaudv_atm")), P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(diss), P(surf), P(ts), P(sf("flux_t")),
                             P(sf("flux_q")), P(sf("flux_r")), P(g["k_rad_net_sw_down"]), P(g["k_rad_lw_down"]), P(sf("dhdt_surf")), P(sf("dedt_surf")),
                             P(sf("drdt_surf")), P(sf("dhdt_atm")), P(sf("dedq_atm")), P(surf_ml), P(dtd))
        else:       # "wave": the smallest kstop of a group of 64 columns, as a wavefront takes it
            kb = kstop.copy() if mode == "limited" else np.repeat([kstop[i:i + 64].min() for i in range(0, nc, 64)], 64)[:nc].astype(np.int32)
            lib.mh_vert_diff_kb(L, nc, D(delt), D(delt / 2), P(u), P(v), P(tm), P(q), P(km), P(kt), P(ph), P(zf), P(sf("flux_u")), P(sf("flux_v")),
                                P(sf("dtaudu_atm")), P(sf("dtaudv_atm")), P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(diss), P(ts), P(sf("flux_t")),
                                P(sf("flux_q")), P(sf("flux_r")), P(g["k_rad_net_sw_down"]), P(g["k_rad_lw_down"]), P(sf("dhdt_surf")),
                                P(sf("dedt_surf")), P(sf("drdt_surf")), P(sf("dhdt_atm")), P(sf("dedq_atm")), np.ascontiguousarray(kb).ctypes.data_as(C.POINTER(C.c_int)))
        out[mode] = a + [diss, ts]
    for mode in ("limited", "wave"):
        for x, y, nm in zip(out["whole"], out[mode], ("dt_u", "dt_v", "dt_t", "dt_q", "diss_heat", "t_surf")):
            assert np.array_equal(x, y), (mode, nm, float(np.abs(x - y).max()))
