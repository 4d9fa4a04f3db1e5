"""Frierson configuration after a spin-up (it rains): ms/step and per-kernel HIP-event times for the environment this process was started with.
usage: python tools/dev/moist_ab.py [res L dt spinup_steps]     (A/B: run it once per variant of the ISCA_MOIST_* switches inside one gpurun call)"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from isca_amd import dyncore
import ctypes as C
res = sys.argv[1] if len(sys.argv) > 1 else "T85"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
spin = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
cfg = dyncore.default_config(res, num_levels=L, physics=1, dt_atmos=dt, initial_sphum=2e-6, robert_coeff=0.03, scale_heights=11.0, exponent=7.0)
dc = dyncore.DynCore(cfg); dc.cold_start()
dc.step(spin)
best = 1e9
for rep in range(3):
    t0 = time.time(); dc.step(1000); t1 = time.time()
    best = min(best, (t1 - t0) / 1000 * 1e3)
tg = dc.get("tg")
var = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("ISCA_MOIST")) or "defaults"
print(f"[{var}] {res}L{L} moist after {spin} steps: {best:.4f} ms/step   Tmin/max {tg.min():.3f} {tg.max():.3f} precip max {dc.get('precip').max():.3e}")
ms = (C.c_double * 64)(); names = C.create_string_buffer(4096); n = C.c_int()
dc.lib.isca_dyn_kernel_times(dc._h, 1, ms, 64, names, 4096, C.byref(n))
dc.step(200)
dc.lib.isca_dyn_kernel_times(dc._h, 0, ms, 64, names, 4096, C.byref(n))
nm = [x for x in names.value.decode().split(';') if x]
print("   " + "  ".join(f"{nm[i]} {ms[i]*1e3:.1f}" for i in range(min(n.value, len(nm)))))
dc.close()
