"""world_size-2 (and 4) gloo tests on CPU of the N>1 path's host logic: the wavenumber dealing exported
by the C-ABI, and the exchange (isca_amd.parallel.exchange / allreduce_sum) with exactly the buffer layout
the device kernels use -- [peer][m_local][lat_local][column] -- driven by oracle math standing in for the
device phases, so that a sharded transform pair must reproduce the unsharded one."""
import os, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dealing_is_balanced_bijection():
    from isca_amd import build, dyncore
    build.build(verbose=False)
    for nf, P in ((21, 2), (42, 4), (85, 8), (85, 2), (170, 8), (10, 2)):
        deal = dyncore.wavenumber_dealing(nf, P)
        owned = deal[deal >= 0]
        assert sorted(owned.tolist()) == list(range(nf + 1))
        assert deal[0, 0] == 0                       # the fixers patch (m,n)=(0,0) on rank 0, slot 0
        # triangular work per rank (rows n <= N+1-m) within 15 % of the mean
        work = np.array([sum((nf + 2 - m) for m in row if m >= 0) for row in deal], dtype=float)
        assert work.max() / work.mean() < 1.15, (nf, P, work)


def _worker(rank, world, port, res, L, out):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isca_amd import dyncore, parallel
    from oracle.isca_oracle import Config, SpectralCore
    sc = SpectralCore(Config(num_levels=L, **dyncore.RESOLUTIONS[res]))
    deal = dyncore.wavenumber_dealing(sc.cfg.num_fourier, world)
    Ml, Jl = deal.shape[1], sc.J // world
    rng = np.random.default_rng(5)
    g = rng.standard_normal((L, sc.J, sc.I))             # same global field on every rank
    C = 2 * L
    # "phase 0" of this rank: FFT of its latitude band into the send buffer [q][ml][jl][C]
    band = g[:, rank * Jl:(rank + 1) * Jl]
    f = sc.grid_to_fourier(band)[..., : sc.M1]            # [L, Jl, M1]
    send = np.zeros((world, Ml, Jl, C))
    for q in range(world):
        for ml, m in enumerate(deal[q]):
            if m >= 0:
                send[q, ml, :, 0::2] = f[:, :, m].real.T
                send[q, ml, :, 1::2] = f[:, :, m].imag.T
    send_t, recv_t = torch.from_numpy(send.reshape(-1)), torch.zeros(send.size, dtype=torch.float64)
    parallel.exchange(send_t, recv_t)
    recv = recv_t.numpy().reshape(world, Ml, Jl, C)       # [source rank p][ml][jl][C]: all latitudes of my m's
    # "phase 1": Legendre analysis for my wavenumbers, truncation, synthesis back
    fm = np.zeros((L, sc.J, sc.M1), dtype=complex)
    for ml, m in enumerate(deal[rank]):
        if m >= 0:
            blk = recv[:, ml]                              # [p, jl, C]
            fm[:, :, m] = (blk[..., 0::2] + 1j * blk[..., 1::2]).reshape(sc.J, L).T
    s = sc.fourier_to_spherical(fm) * sc.triangle_mask
    mine = np.zeros(sc.M1, bool); mine[deal[rank][deal[rank] >= 0]] = True
    s[..., ~mine] = 0
    f2 = sc.spherical_to_fourier(s)                        # [L, J, M1], only my m's non-zero
    send2 = np.zeros((world, Ml, Jl, C))
    for p in range(world):
        for ml, m in enumerate(deal[rank]):
            if m >= 0:
                send2[p, ml, :, 0::2] = f2[:, p * Jl:(p + 1) * Jl, m].real.T
                send2[p, ml, :, 1::2] = f2[:, p * Jl:(p + 1) * Jl, m].imag.T
    s2, r2 = torch.from_numpy(send2.reshape(-1)), torch.zeros(send2.size, dtype=torch.float64)
    parallel.exchange(s2, r2)
    recv2 = r2.numpy().reshape(world, Ml, Jl, C)           # [owner q][ml][jl][C]
    fb = np.zeros((L, Jl, sc.I // 2 + 1), dtype=complex)
    for q in range(world):
        for ml, m in enumerate(deal[q]):
            if m >= 0:
                fb[:, :, m] = (recv2[q, ml][..., 0::2] + 1j * recv2[q, ml][..., 1::2]).T
    gb = sc.fourier_to_grid(fb)                            # my band of the filtered field
    # the all-reduce of the fixer sums
    red = torch.tensor([float(np.sum(sc.wts_lat[rank * Jl:(rank + 1) * Jl, None] * gb[0]))], dtype=torch.float64)
    parallel.allreduce_sum(red)
    ref = sc.trans_spherical_to_grid(sc.trans_grid_to_spherical(g))
    err = np.max(np.abs(gb - ref[:, rank * Jl:(rank + 1) * Jl])) / np.max(np.abs(ref))
    gm = red.item() / (np.sum(sc.wts_lat) * sc.I)
    out[rank] = (float(err), float(abs(gm - sc.area_weighted_global_mean(ref[0]))))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,res,L", [(2, "T21", 3), (4, "T21", 2)])
def test_sharded_transform_pair_gloo(world, res, L):
    from isca_amd import build
    build.build(verbose=False)
    mgr = mp.Manager(); out = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, res, L, out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        err, gerr = out[rank]
        assert err < 1e-13 and gerr < 1e-13, (rank, err, gerr)
