"""T85L40 moist (Frierson) configuration: ms/step and per-kernel times."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from isca_amd import dyncore
import ctypes as C
res = sys.argv[1] if len(sys.argv) > 1 else "T85"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
cfg = dyncore.default_config(res, num_levels=L, physics=1, dt_atmos=dt, initial_sphum=2e-6, robert_coeff=0.03, scale_heights=11.0, exponent=7.0)
dc = dyncore.DynCore(cfg); dc.cold_start()
mode = sys.argv[4] if len(sys.argv) > 4 else ""
if mode == "spunup":               # for the profiler (rocprofv3 --collection-period 9:2:1): 35 days of spin-up, then steps for ~7 s
    dc.step(10000)
    t0 = time.time(); dc.step(30000); t1 = time.time()
    print(f"{res}L{L} moist, steps 10000-40000: {(t1-t0)/30000*1e3:.4f} ms/step")
    sys.exit(0)
short = mode == "short"            # a short run for the counter passes
dc.step(60 if short else 200)
for rep in range(1 if short else 2):
    n_ = 40 if short else 500
    t0 = time.time(); dc.step(n_); t1 = time.time()
    if short:
        print(f"{res}L{L} moist (short): {(t1-t0)/n_*1e3:.4f} ms/step"); break
    print(f"{res}L{L} moist: {(t1-t0)/n_*1e3:.4f} ms/step", "Tmin/max", dc.get('tg').min(), dc.get('tg').max(), 'umax', np.abs(dc.get('ug')).max())
lib = dc.lib
ms = (C.c_double*64)(); names = C.create_string_buffer(4096); n = C.c_int()
lib.isca_dyn_kernel_times(dc._h, 1, ms, 64, names, 4096, C.byref(n))
dc.step(20 if short else 200)
lib.isca_dyn_kernel_times(dc._h, 0, ms, 64, names, 4096, C.byref(n))
nm = [x for x in names.value.decode().split(';') if x]
for i in range(min(n.value, len(nm))): print(f"  {nm[i]:20s} {ms[i]*1e3:8.1f} us")
