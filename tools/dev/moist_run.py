"""Plain T85L40 moist run for profiling (no event timers)."""
import sys
sys.path.insert(0, '/root/repo')
from isca_amd import dyncore
cfg = dyncore.default_config("T85", num_levels=40, physics=1, dt_atmos=300.0, initial_sphum=2e-6, robert_coeff=0.03, scale_heights=11.0, exponent=7.0)
dc = dyncore.DynCore(cfg); dc.cold_start()
dc.step(300)
print("done", dc.get("tg").max())
