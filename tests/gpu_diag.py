#!/usr/bin/env python3
"""Stage-by-stage parity report of the HIP path against the oracle (run on the GPU box).
Prints one line per check; exits non-zero if any check exceeds its tolerance.
Usage: python tests/gpu_diag.py [T21:25 T10:8 ...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isca_amd import dyncore
from oracle.isca_oracle import Config, SpectralCore

FAIL = []


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def check(name, a, b, tol, absolute=False):
    e = float(np.max(np.abs(a - b))) if absolute else rel(a, b)
    ok = np.isfinite(e) and e <= tol
    print(f"  {'ok  ' if ok else 'FAIL'} {name:34s} err={e:.3e} tol={tol:.0e}", flush=True)
    if not ok:
        FAIL.append(name)


def rand_spec(rng, sc, L):
    s = rng.standard_normal((L, sc.N1, sc.M1)) + 1j * rng.standard_normal((L, sc.N1, sc.M1))
    s[..., 0] = s[..., 0].real
    s = s / (1.0 + sc.spherical_wave) ** 2
    return s * sc.triangle_mask


def run(res, L, impl, nsteps=12):
    print(f"== {res} L{L} legendre_impl={impl}", flush=True)
    r = dyncore.RESOLUTIONS[res]
    ocfg = Config(num_levels=L, **r)
    sc = SpectralCore(ocfg)
    cfg = dyncore.default_config(res, num_levels=L, legendre_impl=impl)
    dc = dyncore.DynCore(cfg)
    # tables
    check("table sin_lat", dc.table("sin_lat"), sc.sin_lat, 0, True)
    check("table wts_lat", dc.table("wts_lat"), sc.wts_lat, 0, True)
    check("table legendre", dc.table("legendre"), sc.legendre, 0, True)
    check("table bk", dc.table("bk"), sc.bk, 0, True)
    rng = np.random.default_rng(7)
    sa, sb = rand_spec(rng, sc, L), rand_spec(rng, sc, L)
    ga, gb = 10 * rng.standard_normal((L, sc.J, sc.I)), 10 * rng.standard_normal((L, sc.J, sc.I))
    # stages
    check("fft fwd (grid->fourier)", dc.trans_grid_to_fourier(ga), sc.grid_to_fourier(ga)[..., : sc.M1], 1e-13)
    f = sc.spherical_to_fourier(sa)
    check("legendre inv (spec->fourier)", dc.trans_spherical_to_fourier(sa), f, 1e-13)
    full = np.zeros(f.shape[:-1] + (sc.I // 2 + 1,), dtype=complex); full[..., : sc.M1] = f
    check("fft inv (fourier->grid)", dc.trans_fourier_to_grid(f), sc.fourier_to_grid(full), 1e-13)
    fr = sc.grid_to_fourier(ga)[..., : sc.M1]
    check("legendre fwd (fourier->spec)", dc.trans_fourier_to_spherical(fr), sc.fourier_to_spherical(fr), 1e-13)
    check("trans_spherical_to_grid", dc.trans_spherical_to_grid(sa), sc.trans_spherical_to_grid(sa), 1e-13)
    check("trans_grid_to_spherical", dc.trans_grid_to_spherical(ga), sc.trans_grid_to_spherical(ga), 1e-13)
    check("trans_grid_to_spherical notrunc", dc.trans_grid_to_spherical(ga, False), sc.trans_grid_to_spherical(ga, False), 1e-13)
    check("2-D s2g", dc.trans_spherical_to_grid(sa[0]), sc.trans_spherical_to_grid(sa[0]), 1e-13)
    vo, dv = dc.vor_div_from_uv_grid(ga, gb); vo2, dv2 = sc.vor_div_from_uv_grid(ga, gb)
    check("vor_div_from_uv_grid vor", vo, vo2, 1e-13); check("vor_div_from_uv_grid div", dv, dv2, 1e-13)
    u, v = dc.uv_grid_from_vor_div(sa, sb); u2, v2 = sc.uv_grid_from_vor_div(sa, sb)
    check("uv_grid_from_vor_div u", u, u2, 1e-13); check("uv_grid_from_vor_div v", v, v2, 1e-13)
    check("horizontal_advection", dc.horizontal_advection(sa, ga, gb, np.zeros_like(ga)),
          sc.horizontal_advection(sa, ga, gb, np.zeros_like(ga)), 1e-13)
    ps = 1e5 * (1 + 0.03 * rng.standard_normal((sc.J, sc.I)))
    T = 260 + 20 * rng.standard_normal((L, sc.J, sc.I))
    ph, lph, pf, lpf = sc.pressure_variables(ps)
    ut, vt, tt = dc.hs_forcing(1200.0, ph, pf, ga, gb, T)
    u2, v2, t2 = sc.hs_forcing(1200.0, ph, pf, ga, gb, T)
    check("hs_forcing udt", ut, u2, 1e-13); check("hs_forcing vdt", vt, v2, 1e-13); check("hs_forcing tdt", tt, t2, 1e-12)
    # cold start + steps with intermediates
    dc.cold_start(); sc.cold_start()
    st = dc.state(); so = sc.state()
    for k in ("ug", "vg"):
        check(f"cold {k}", st[k], so[k], 1e-15, True)
    for k in ("tg", "psg", "ts", "ln_ps", "vors"):
        check(f"cold {k}", st[k], so[k], 1e-13)
    check("cold vorg", dc.get("vorg"), sc.vorg, 1e-20, True)
    for i in range(1, nsteps + 1):
        for ph_ in range(4):
            dc.step_phase(ph_)
            if i <= 3 and ph_ == 0:
                sc.step()
                for k in ("g_dtu", "g_dtv", "g_dtT", "g_E", "g_dtlp", "wg_full"):
                    d_, o_ = dc.get(k), sc.dbg[k]
                    scale = max(np.abs(o_).max(), 1e-300)
                    check(f"step{i} {k}", d_ / scale, o_ / scale, 1e-11 if k != "wg_full" else 1e-9, True)
            if i <= 3 and ph_ == 1:
                for k in ("s_dtvor", "s_dtdiv", "s_dtT", "s_dtlp"):
                    d_, o_ = dc.get(k), sc.dbg[k]
                    scale = max(np.abs(o_).max(), 1e-300)
                    check(f"step{i} {k}", d_ / scale, o_ / scale, 1e-10, True)
        if i > 3:
            sc.step()
        if i in (1, 2, 3, nsteps):
            st = dc.state(); so = sc.state()
            for k in ("ug", "vg"):
                check(f"step{i} {k}", st[k], so[k], 1e-10, True)
            for k in ("tg", "psg", "ts", "ln_ps"):
                check(f"step{i} {k}", st[k], so[k], 1e-11)
            check(f"step{i} vors", st["vors"], so["vors"], 1e-9)
            check(f"step{i} divs(abs)", st["divs"], so["divs"], 1e-16, True)
            fx = dc.table("fixer")
            print(f"     fixer factor-1={fx[16]-1:.3e} tcorr={fx[17]:.3e} water={fx[18]-1:.3e}")
    t0 = time.time(); dc.step(50, True); t1 = time.time()
    print(f"  timing: {1e3*(t1-t0)/50:.3f} ms/step (eager, host-inclusive)")
    dc.kernel_times(True); dc.step(20, True)
    kt = dc.kernel_times(False)
    print("  kernel ms:", {k: round(v, 4) for k, v in kt.items()}, "sum", round(sum(kt.values()), 4))
    dc.close()


if __name__ == "__main__":
    specs = sys.argv[1:] or ["T10:8:1", "T21:25:1", "T21:25:0", "T42:25:0"]
    for s in specs:
        res, L, impl = s.split(":")
        try:
            run(res, int(L), int(impl))
        except Exception as e:   # keep going: one GPU call should report as much as possible
            import traceback; traceback.print_exc()
            FAIL.append(f"{s}: {e}")
    print("FAILED:" if FAIL else "ALL OK", FAIL)
    sys.exit(1 if FAIL else 0)
