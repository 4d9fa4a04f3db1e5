"""What the grid tracer costs a T85L40 Held-Suarez step although its kernels run on the side stream: the step and the per-kernel event times with
num_tracers = 1 (the headline configuration) and num_tracers = 0, in one process (HISTORY.md "Round 6")."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from isca_amd import dyncore
for ntr in (1, 0):
    cfg = dyncore.default_config("T85", num_levels=40, dt_atmos=300.0, num_tracers=ntr)
    dc = dyncore.DynCore(cfg); dc.cold_start(); dc.step(2400)
    best = 1e9
    for rep in range(3):
        t0 = time.time(); dc.step(1000); best = min(best, (time.time() - t0))
    ms = (C.c_double * 64)(); names = C.create_string_buffer(4096); n = C.c_int()
    dc.lib.isca_dyn_kernel_times(dc._h, 1, ms, 64, names, 4096, C.byref(n)); dc.step(300)
    dc.lib.isca_dyn_kernel_times(dc._h, 0, ms, 64, names, 4096, C.byref(n))
    nm = [x for x in names.value.decode().split(';') if x]
    print(f"num_tracers={ntr}: {best:.4f} ms/step  " + "  ".join(f"{nm[i]} {ms[i]*1e3:.1f}" for i in range(min(n.value, len(nm)))))
    dc.close()
