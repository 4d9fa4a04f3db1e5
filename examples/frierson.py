#!/usr/bin/env python3
"""The Frierson grey-radiation aquaplanet (moist physics package) in run segments:
   python examples/frierson.py [--res T42] [--runs 2] [--days 30] [--workdir /tmp/isca_amd_work]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isca_amd import configs
from isca_amd.experiment import Experiment

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="T42"); ap.add_argument("--runs", type=int, default=2); ap.add_argument("--days", type=int, default=30)
ap.add_argument("--workdir", default="/tmp/isca_amd_work")
a = ap.parse_args()

exp = Experiment("frierson_test_experiment", a.workdir)
exp.diag_table.add_file("atmos_monthly", 30, "days", time_units="days")
for name in ("ps", "sphum", "ucomp", "vcomp", "temp", "vor", "div"):
    exp.diag_table.add_field("dynamics", name, time_avg=True)
exp.diag_table.add_field("atmosphere", "precipitation", time_avg=True)
exp.diag_table.add_field("mixed_layer", "t_surf", time_avg=True)
exp.update_namelist(configs.frierson())
exp.update_namelist({"main_nml": {"days": a.days}})
exp.set_resolution(a.res)
exp.run(1, use_restart=False, overwrite_data=True)
for i in range(2, a.runs + 1):
    exp.run(i, overwrite_data=True)
print("output in", exp.datadir)
