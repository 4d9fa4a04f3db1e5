#!/usr/bin/env python3
"""The shallow-water sibling core: a vortex pair on a zonal flow (and optionally the stirring forcing) at T85:
   python examples/shallow_water.py [--days 10] [--stirring]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isca_amd import shallow

ap = argparse.ArgumentParser()
ap.add_argument("--days", type=int, default=10); ap.add_argument("--stirring", action="store_true")
a = ap.parse_args()
nml = {"main_nml": {"dt_atmos": 1200}, "shallow_dynamics_nml": {"add_initial_vortex_pair": True, "u_upper_mag_init": 10.0, "u_deep_mag": 5.0}}
if a.stirring:
    nml["stirring_nml"] = {"decay_time": 172800, "amplitude": 3.e-13, "lat0": 45., "lon0": 180., "widthy": 12., "widthx": 45., "B": 1.0}
model = shallow.atmosphere_init(nml, "T85")
for day in range(1, a.days + 1):
    shallow.atmosphere(72)
    h, vor = model.get("h"), model.get("vor")
    print(f"day {day:3d}  h in [{h.min():9.2f}, {h.max():9.2f}] m   enstrophy {np.mean(vor * vor):.4e}")
shallow.atmosphere_end()
