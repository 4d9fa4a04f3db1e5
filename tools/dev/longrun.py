import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from isca_amd import dyncore
dc = dyncore.DynCore(dyncore.default_config("T85", num_levels=40, dt_atmos=300.0))
dc.cold_start()
t=time.time()
for i in range(10):
    dc.step(5000)
    T=dc.get("tg"); u=dc.get("ug"); ps=dc.get("psg")
    print(i, "steps", (i+1)*5000, "T", T.min(), T.max(), "maxU", np.abs(u).max(), "mean ps", ps.mean(), "tr max", dc.get("tr").max(), flush=True)
print("elapsed", time.time()-t)
