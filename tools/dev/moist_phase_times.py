"""Phase times of k_moist_physics from the MOIST_TIMING builds (tools/build_moist_timing.sh): lane i of
every wavefront stores the wall_clock64 ticks (10 ns) between marks i and i+1 of phase P in the precipitation field (moist.hip, MT macros)."""
import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
REPO = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
MARKS = {1: ["convection + condensation of the next step (wavefront A)"],
         2: ["height sum", "gray_rad_down", "gray_rad_up", "surface_flux", "rayleigh sponge"],
         5: ["init pass (Tv, parcel)", "below the LCL", "ascent above the LCL", "reference profiles + Pq, Pt", "deep / shallow adjustment", "(end)"],
         3: ["(lambdas)", "pbl_depth + start level", "passthrough", "pbl profile + momentum down", "momentum up", "vert_diff_heat_down", "mixed_layer", "vert_diff_up"]}
if len(sys.argv) > 1:
    import numpy as np
    from isca_amd import dyncore
    ph = int(sys.argv[1])
    cfg = dyncore.default_config("T85", num_levels=40, physics=1, dt_atmos=300.0, initial_sphum=2e-6, robert_coeff=0.03, scale_heights=11.0, exponent=7.0)
    dc = dyncore.DynCore(cfg); dc.cold_start(); dc.step(int(os.environ.get('MOIST_SPINUP', '400')))      # (10000 steps = 35 days: convection active everywhere)
    p = dc.get("precip").reshape(-1, 8) * 0.01        # us; column c holds mark interval c % 8
    tot = 0.0
    for i, nm in enumerate(MARKS[ph]):
        print(f"  phase {ph}  {nm:36s} mean {p[:, i].mean():7.1f} us   max {p[:, i].max():7.1f}")
        tot += p[:, i].mean()
    rows = p[:, :len(MARKS[ph])].sum(axis=1)
    print(f"  phase {ph}  total {tot:.1f} us   slowest wavefront {rows.max():.1f} us (its parts: {' '.join(f'{x:.1f}' for x in p[rows.argmax(), :len(MARKS[ph])])})")
else:
    only = os.environ.get('MOIST_PHASES')
    for v in (1, 2, 3, 5):
        if only and str(v) not in only.split(','): continue
        env = dict(os.environ, ISCA_DYN_LIB=os.path.join(REPO, "isca_amd", "lib", f"libisca_dyn_mt{v}.so"))
        subprocess.run([sys.executable, __file__, str(v)], env=env)
