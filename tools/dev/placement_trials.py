"""Does the placement of the device arrays change when a core is destroyed and created again in the same process?  (No: +-0.3 % at T85L40 and T170L60.)"""
import sys, time
sys.path.insert(0, '/root/repo')
from isca_amd import dyncore
for res, L, dt in (("T85", 40, 300.0), ("T170", 60, 150.0)):
    out = []
    for trial in range(6):
        dc = dyncore.DynCore(dyncore.default_config(res, num_levels=L, dt_atmos=dt)); dc.cold_start()
        dc.step(1500 if res == "T85" else 300)
        n = 1000 if res == "T85" else 200
        t0 = time.time(); dc.step(n); t = (time.time() - t0) / n * 1e3
        out.append(round(t, 4))
        dc.close()
    print(res, out, flush=True)
