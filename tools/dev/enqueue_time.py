"""Host time to enqueue steps against the time the GPU needs for them (HISTORY.md 4, "Why the step is not a hipGraph")."""
import sys, time
sys.path.insert(0, "/root/repo")
from isca_amd import dyncore
dc = dyncore.DynCore(dyncore.default_config("T85", num_levels=40, dt_atmos=300.0))
dc.cold_start(); dc.step(200)
for n in (200, 1000, 1000):
    t0 = time.perf_counter(); dc.step(n, sync=False); t1 = time.perf_counter(); dc.lib.isca_dyn_synchronize(dc._h); t2 = time.perf_counter()
    print(f"{n} steps: isca_dyn_step returned after {1e6 * (t1 - t0) / n:.1f} us per step; the GPU finished after {1e6 * (t2 - t0) / n:.1f} us per step")
