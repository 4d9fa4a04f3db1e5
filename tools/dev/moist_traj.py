import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_gpu_moist import moist_core
G='/root/repo/tests/golden/'
g = np.load(G+'moist_run_T21L25.npz'); g2 = np.load(G+'moist_run_T21L25_12day.npz')
dc = moist_core(dt=720.0); dc.cold_start()
done=0
for n in (1,2,10,144,1440):
    dc.step(n-done); done=n
    gg = g if n<1440 else g2
    err={}
    for mine, ref in (("ug","ug"),("vg","vg"),("tg","tg"),("tr","q"),("psg","psg")):
        key=f"st_{ref}_{n:06d}"
        if key in gg.files:
            a=dc.get(mine); b=gg[key]
            err[ref]=(float(np.abs(a-b).max()), float(np.abs(b).max()))
            if n==1440:
                za=a.mean(axis=-1); zb=b.mean(axis=-1)
                err[ref+'_zm']=(float(np.abs(za-zb).max()), float(np.abs(zb).max()))
                err[ref+'_gm']=(float(a.mean()), float(b.mean()))
    print(n, err)
ts=dc.get('t_surf'); print('t_surf', ts.min(), ts.max(), 'precip max', dc.get('precip').max())
