"""Moist step time as the model state evolves from the cold start (the moist kernel is slower once convection is active).
usage: python tools/dev/moist_time_vs_state.py [steps,steps,...]"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from isca_amd import dyncore
import ctypes as C
cfg = dyncore.default_config("T85", num_levels=40, physics=1, dt_atmos=300.0, initial_sphum=2e-6, robert_coeff=0.03, scale_heights=11.0, exponent=7.0)
dc = dyncore.DynCore(cfg); dc.cold_start()
done = 0
for upto in [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else '200,1000,3000,10000,30000'.split(','))]:
    dc.step(upto - done); done = upto
    t0 = time.time(); dc.step(300); t1 = time.time(); done += 300
    lib = dc.lib; ms = (C.c_double*64)(); names = C.create_string_buffer(4096); n = C.c_int()
    lib.isca_dyn_kernel_times(dc._h, 1, ms, 64, names, 4096, C.byref(n)); dc.step(100); done += 100
    lib.isca_dyn_kernel_times(dc._h, 0, ms, 64, names, 4096, C.byref(n))
    nm = [x for x in names.value.decode().split(';') if x]
    kt = {nm[i]: ms[i]*1e3 for i in range(min(n.value, len(nm)))}
    print(f"after {upto:6d} steps ({upto*300/86400:6.1f} days): {(t1-t0)/300*1e3:.4f} ms/step  moist_physics {kt.get('moist_physics',0):.1f} us  precip max {dc.get('precip').max():.3e}  Tmin {dc.get('tg').min():.1f}", flush=True)
