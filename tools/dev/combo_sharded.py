#!/usr/bin/env python3
"""Exploratory: sharded (2 ranks on one GPU, gloo) vs single for option combinations.  torchrun --nproc-per-node 2 tools/dev/combo_sharded.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from isca_amd import dyncore
from isca_amd.parallel import ShardedDynCore
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")


def topo(dc):
    lat = np.deg2rad(dc.table("deg_lat"))[:, None]; lon = np.deg2rad(dc.table("deg_lon"))[None, :]
    dc.set_surf_geopotential(9.8 * 2000.0 * np.exp(-((lat - 0.6) / 0.3) ** 2 - ((lon - 2.0) / 0.5) ** 2))


CASES = {
    "topography + virtual temperature": dict(use_virtual_temperature=1),
    "exponential damping + sponge": dict(damping_option=1, cutoff_wn=10, damping_order=3, eddy_sponge_coeff=1e-5),
    "res-independent damping, no fixers": dict(damping_option=2, damping_order=2, damping_coeff=2e16, do_mass_correction=0, do_energy_correction=0, do_water_correction=0),
}
ok = True
for name, opts in CASES.items():
    sh = ShardedDynCore(dyncore.default_config("T21", num_levels=8, rank=rank, world_size=world, device=0, **opts))
    if "topography" in name:
        topo(sh)
    sh.cold_start(); sh.step(10)
    got = {k: sh.gather_grid(k) for k in ("ug", "tg", "tr", "psg")}
    if rank == 0:
        ref = dyncore.DynCore(dyncore.default_config("T21", num_levels=8, device=0, **opts))
        if "topography" in name:
            topo(ref)
        ref.cold_start(); ref.step(10)
        for k, v in got.items():
            r = ref.get(k)
            e = np.abs(v - r).max() / max(np.abs(r).max(), 1.0 if k == "ug" else 1e-300)
            print(f"{name}: {k} {e:.2e}")
            ok &= bool(e < 1e-10)
        ref.close()
    sh.close()
    dist.barrier()
if rank == 0:
    print("SHARDED COMBOS", "OK" if ok else "FAILED")
dist.destroy_process_group()
