#!/usr/bin/env python3
"""Measured parity of the HIP path, printed as a markdown table (run on the GPU box; output kept in PARITY.md).
Rows: committed reference outputs (tests/golden, produced by the reference Fortran itself) and the numpy oracle."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from isca_amd import dyncore                                    # noqa: E402
from oracle.isca_oracle import Config, SpectralCore             # noqa: E402  (checker only)

GOLD = os.path.join(REPO, "tests", "golden")
rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
rows = []


def row(what, against, err, bound):
    rows.append(f"| {what} | {against} | {err:.1e} | {bound:g} |")


def make(res, L, **kw):
    return dyncore.DynCore(dyncore.default_config(res, num_levels=L, **kw))


g = np.load(os.path.join(GOLD, "kernels_T21L6.npz"))
dc = make("T21", 6)
sa, sb, ga, gb = g["in_spec_a"], g["in_spec_b"], g["in_grid_a"], g["in_grid_b"]
row("Legendre+FFT synthesis `trans_spherical_to_grid` (T21L6)", "reference output", rel(dc.trans_spherical_to_grid(sa), g["out_s2g_a"]), 1e-12)
row("analysis `trans_grid_to_spherical`", "reference output", rel(dc.trans_grid_to_spherical(ga), g["out_g2s_a"]), 1e-12)
v, d = dc.vor_div_from_uv_grid(ga, gb)
row("`vor_div_from_uv_grid`", "reference output", max(rel(v, g["out_vor_from_uv"]), rel(d, g["out_div_from_uv"])), 1e-12)
u, vv = dc.uv_grid_from_vor_div(sa, sb)
row("`uv_grid_from_vor_div`", "reference output", max(rel(u, g["out_u_from_vd"]), rel(vv, g["out_v_from_vd"])), 1e-12)
row("`horizontal_advection`", "reference output", rel(dc.horizontal_advection(sa, ga, gb, np.zeros_like(ga)), g["out_hadv"]), 1e-12)
ut, vt, tt = dc.hs_forcing(1200.0, g["out_p_half"], g["out_p_full"], ga, gb, g["in_temp"])
row("`hs_forcing` (u, v, T tendencies)", "reference output", max(rel(ut, g["out_hs_dt_u"]), rel(vt, g["out_hs_dt_v"]), rel(tt, g["out_hs_dt_t"])), 1e-12)
o1, o2, o3 = dc.implicit_correction(g["in_spec_e"], g["in_spec_f"], g["in_spec2_c"], (sa, sb), (g["in_spec_c"], g["in_spec_d"]),
                                    (g["in_spec2_a"], g["in_spec2_b"]), 1200.0)
row("`implicit_correction` (step's own kernel)", "reference output", max(rel(o1, g["out_impl_dt_divs"]), rel(o2, g["out_impl_dt_ts"]), rel(o3, g["out_impl_dt_lnps"])), 1e-12)
row("`compute_spectral_damping`", "reference output", rel(dc.compute_spectral_damping(sa, g["in_spec_e"], 1200.0, "t"), g["out_damp"]), 1e-14)
new, filt = dc.leapfrog(sa, sb, g["in_spec_e"], 1200.0, 0.04)
row("`leapfrog_2level_A/B`", "reference output", max(rel(new, g["out_leap_l1"]), rel(filt, g["out_leap_l2"])), 1e-15)
q = g["in_q"]
row("PPM `vert_advection` (step's own kernel)", "reference output", rel(dc.vert_advection_ppm(1200.0, g["in_wg"], g["in_ps"], q), g["out_vadv_ppm"]), 1e-12)
row("`a_grid_horiz_advection`, Courant < 1", "reference output", rel(dc.a_grid_horiz_advection(ga, gb, q, 1200.0), g["out_hadv_fv"]), 1e-12)
row("`a_grid_horiz_advection`, Courant > 1", "reference output", rel(dc.a_grid_horiz_advection(ga, gb, q, 48000.0), g["out_hadv_fv_bigcfl"]), 1e-12)
dc.close()

for name, res, L, n, tol in (("run_T21L25", "T21", 25, 144, 1e-9), ("run_T21L25_10day", "T21", 25, 1440, 1e-7)):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    dc = make(res, L); dc.cold_start(); dc.step(n)
    tag = f"{n:06d}"
    e = {k: rel(dc.get(k), g[f"st_{k}_{tag}"]) for k in ("ug", "vg", "tg", "psg")}
    e["tr"] = rel(dc.get("tr"), g[f"st_tr1_{tag}"])
    row(f"{res}L{L} Held-Suarez, {n} steps from the cold start: u, v", "reference run", max(e["ug"], e["vg"]), tol)
    row("... T, ps, tracer", "reference run", max(e["tg"], e["psg"], e["tr"]), tol)
    dc.close()
g = np.load(os.path.join(GOLD, "run_T42L25.npz"))
dc = make("T42", 25); dc.cold_start(); dc.step(144)
row("T42L25 Held-Suarez, 1 day: u, v, T (sampled), ps", "reference run",
    max(rel(dc.get("ug")[::2, ::2, ::2], g["st_ug_000144_s222"]), rel(dc.get("vg")[::2, ::2, ::2], g["st_vg_000144_s222"]),
        rel(dc.get("tg")[::2, ::2, ::2], g["st_tg_000144_s222"]), rel(dc.get("psg"), g["st_psg_000144"])), 1e-9)
dc.close()

t0 = time.time()
dc = make("T42", 25); dc.cold_start(); dc.step(36)
sc = SpectralCore(Config(num_levels=25, **dyncore.RESOLUTIONS["T42"])); sc.cold_start()
for _ in range(36):
    sc.step()
c = sc.current
row("T42L25, 36 steps: spectral state (vors, divs, ts, ln_ps)", "numpy oracle",
    max(rel(dc.get("vors"), sc.vors[c]), rel(dc.get("divs"), sc.divs[c]), rel(dc.get("ts"), sc.ts[c]), rel(dc.get("ln_ps"), sc.ln_ps[c])), 1e-9)
dc.close()

print("| quantity | against | max relative error (L-inf) | stated bound |\n|---|---|---|---|")
print("\n".join(rows))
