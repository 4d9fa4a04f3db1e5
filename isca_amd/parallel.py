"""Latitude-band sharding over the GPUs of one node: host side of the lat<->m exchange.

The reference's "transpose method" (tools/spec_mpp.F90:61-80, transforms.F90:970-1056): grid space is
split into latitude bands, spectral space into zonal-wavenumber sets, with ONE all-to-all between the
FFT and the Legendre stage of every (batched) transform, plus the all-reduce of the fixer sums
(transforms.F90:1059-1077 gathers instead).  Here each rank drives its own GPU through the phase API of
the C-ABI and the exchanges go through torch.distributed ("nccl" = RCCL over xGMI on the GPU box;
"gloo" with host staging for CPU-side tests).  Per step: 2 all-to-alls + 1 all-reduce of 5 doubles.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import dyncore


class _DevPtr:
    """__cuda_array_interface__ view of a device buffer owned by the C library."""
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (ptr, False), "version": 3}


def exchange(send, recv, group=None):
    """all_to_all_single of equal blocks [P][block]; stages through the host on non-NCCL backends."""
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        dist.all_to_all_single(recv, send, group=group)
        return
    P = dist.get_world_size(group)
    s = send.detach().cpu().reshape(P, -1)
    outs = [torch.empty_like(s[0]) for _ in range(P)]
    _gloo_all_to_all(outs, s, group)
    recv.copy_(torch.stack(outs).reshape(recv.shape).to(recv.device))


def _gloo_all_to_all(outs, s, group):
    """gloo has no all_to_all: P rounds of scatter (root q scatters its P blocks)."""
    import torch.distributed as dist
    P = dist.get_world_size(group)
    me = dist.get_rank(group)
    for root in range(P):
        dist.scatter(outs[root], [s[q].contiguous() for q in range(P)] if me == root else None, src=root, group=group)


def halo_exchange(send_lo, send_hi, recv_lo, recv_hi, group=None):
    """Nearest-neighbour exchange of the tracer halo rows: send_lo -> rank-1 (its recv_hi), send_hi -> rank+1 (its recv_lo)."""
    import torch
    import torch.distributed as dist
    P, me = dist.get_world_size(group), dist.get_rank(group)
    nccl = dist.get_backend(group) == "nccl"
    sl, sh = (send_lo, send_hi) if nccl else (send_lo.detach().cpu(), send_hi.detach().cpu())
    rl, rh = (recv_lo, recv_hi) if nccl else (torch.empty_like(sl), torch.empty_like(sh))
    ops = []
    if me > 0:
        ops += [dist.P2POp(dist.isend, sl, me - 1, group), dist.P2POp(dist.irecv, rl, me - 1, group)]
    if me < P - 1:
        ops += [dist.P2POp(dist.isend, sh, me + 1, group), dist.P2POp(dist.irecv, rh, me + 1, group)]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if not nccl:
        if me > 0:
            recv_lo.copy_(rl.to(recv_lo.device))
        if me < P - 1:
            recv_hi.copy_(rh.to(recv_hi.device))


def allreduce_sum(t, group=None):
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, group=group)
        return
    c = t.detach().cpu()
    dist.all_reduce(c, group=group)
    t.copy_(c.to(t.device))


class ShardedDynCore(dyncore.DynCore):
    """DynCore for world_size > 1: same interface, step() interleaves the device phases with the exchanges."""

    def __init__(self, cfg, group=None):
        import torch
        import torch.distributed as dist
        self.group = group
        assert cfg.world_size == dist.get_world_size(group) and cfg.rank == dist.get_rank(group)
        self._torch = torch
        # one explicit stream shared by our kernels and (through stream-ordered waits) torch's collectives
        self._stream = torch.cuda.Stream(device=cfg.device)
        cfg.stream = self._stream.cuda_stream
        super().__init__(cfg)
        self._bufs = []
        self._raw = cfg.raw_filter_coeff != 1.0        # Robert-Asselin-Williams: a third exchange at the end of the step (gradient batch)
        for which in (0, 1, 2) if self._raw else (0, 1):
            s, r, nbytes = self.exchange_buffers(which)
            tot = nbytes * cfg.world_size
            self._bufs.append((torch.as_tensor(_DevPtr(s, tot), device="cuda"), torch.as_tensor(_DevPtr(r, tot), device="cuda")))
        b, n = self.reduce_buffer()
        self._red = torch.as_tensor(_DevPtr(b, 8 * n), device="cuda")
        ptrs, hbytes = self.halo_buffers()
        self._halo = [torch.as_tensor(_DevPtr(p, hbytes), device="cuda") for p in ptrs] if hbytes else None
        # Native mode: the library issues the exchanges itself through RCCL on our stream and a whole run of steps is one
        # call (no Python, no torch between the kernels).  Default with the nccl backend; ISCA_COMM=torch|native|ipc|peer overrides
        # (ipc: the same C++ step loop over the library's host-staged exchange, for ranks that share one GPU -- csrc/comm_ipc.cpp; peer: over the
        # library's device-resident exchange between the GPUs of one node, stores into hipIpc-mapped peer buffers -- csrc/comm_peer.hip;
        # the library reads the variable itself when rank 0 draws the communicator id).
        self.native = False
        mode = os.environ.get("ISCA_COMM", "native" if dist.get_backend(group) == "nccl" else "torch")
        if mode not in ("native", "ipc", "peer", "torch"):
            raise dyncore.IscaError(f"ISCA_COMM={mode!r}: expected native, ipc, peer or torch")
        if mode in ("native", "ipc", "peer"):
            box = [None]
            if cfg.rank == 0:
                buf = C.create_string_buffer(128)
                if self.lib.isca_comm_get_unique_id(buf) == 0:
                    box[0] = buf.raw
                else:
                    box[0] = "ERR:" + self.lib.isca_last_error().decode()
            dist.broadcast_object_list(box, src=0, group=group)
            why = box[0] if not isinstance(box[0], bytes) else None
            if why is None:
                ok = self.lib.isca_dyn_comm_init(self._h, box[0]) == 0
                flags = [None] * cfg.world_size
                dist.all_gather_object(flags, ok if ok else self.lib.isca_last_error().decode(), group=group)
                bad = [f for f in flags if f is not True]
                if not bad:           # every rank has a communicator: move rank-tagged patterns through every exchange of the step
                    self.lib.isca_dyn_comm_check.argtypes, self.lib.isca_dyn_comm_check.restype = [C.c_void_p], C.c_int
                    ok = self.lib.isca_dyn_comm_check(self._h) == 0
                    dist.all_gather_object(flags, ok if ok else self.lib.isca_last_error().decode(), group=group)
                    bad = [f for f in flags if f is not True]
                if bad:
                    why = str(bad[0])
                else:
                    self.native = True
            if why is not None:
                self.close()       # the device handle and its stream: a failed attempt must not leak them (bench.py retries in the same process)
                # a silent fall-back would be a silent order-of-magnitude slowdown: torch.distributed between the phases has to be asked for
                raise dyncore.IscaError(f"native exchange ({mode}) not available ({why}); set ISCA_COMM=torch to drive the exchanges through "
                                        "torch.distributed between the device phases")

    def _torch_step(self):
        self.step_phase(0)                                          # grid tendencies + FFT (+ tracer halo rows)
        if self._halo is not None:
            halo_exchange(*self._halo, group=self.group)            # fv_advection's mpp_update_domains
            self.step_phase(4)                                      # tracer transport on the side stream: runs under the exchange
        exchange(self._bufs[0][0], self._bufs[0][1], self.group)    # lat -> m   (transpose_fourier)
        self.step_phase(1)                                          # Legendre, spectral update, Legendre
        exchange(self._bufs[1][0], self._bufs[1][1], self.group)    # m -> lat   (reverse_transpose_fourier)
        self.step_phase(2)                                          # inverse FFT + local fixer sums
        allreduce_sum(self._red, self.group)                        # global means
        if self._raw:        # leapfrog_2level_B's future half changes the new spectral level: its gradients are synthesised again
            self.step_phase(5)                                      # fixers, the filter's adjustment, Legendre synthesis of the gradients
            exchange(self._bufs[2][0], self._bufs[2][1], self.group)
            self.step_phase(6)                                      # their FFT, time-level rotation
        else:
            self.step_phase(3)                                      # fixers, time-level rotation

    def _agree(self, err):
        """error_mesg(..., FATAL) stops every PE: a rank whose band left valid_range_t must not leave the others in the next collective"""
        import torch.distributed as dist
        flags = [None] * self.cfg.world_size
        dist.all_gather_object(flags, err, group=self.group)
        bad = [(r, f) for r, f in enumerate(flags) if f is not None]
        if bad:
            raise dyncore.IscaError(f"rank {bad[0][0]}: {bad[0][1]}")

    def _run(self, native_call, torch_call, sync):
        err = None
        if self.native:
            try:
                native_call()
            except dyncore.IscaError as e:
                if not sync:
                    raise
                err = str(e)
        else:
            with self._torch.cuda.stream(self._stream):
                torch_call()
            if sync:
                self._stream.synchronize()
                try:
                    self.synchronize()             # valid-range check of the temperatures, like isca_dyn_step(sync)
                except dyncore.IscaError as e:
                    err = str(e)
        if sync:
            self._agree(err)

    def step(self, nsteps: int = 1, sync: bool = True):
        def torch_steps():
            for _ in range(nsteps):
                self._torch_step()
        self._run(lambda: dyncore.DynCore.step(self, nsteps, sync), torch_steps, sync)

    def dynamics(self, dt_ug=None, dt_vg=None, dt_tg=None, dt_tracers=None, sync=True):
        """physics = 2 on a sharded run: this rank's band of the tendencies, then one step"""
        def torch_step():
            self.set_tendencies(dt_ug, dt_vg, dt_tg, dt_tracers)
            self._torch_step()
        self._run(lambda: dyncore.DynCore.dynamics(self, dt_ug, dt_vg, dt_tg, dt_tracers, sync), torch_step, sync)

    def gather_grid(self, name, time_level=1):
        """all-gather a grid field to every rank (tests/diagnostics): [lev, lat_global, lon]"""
        import torch.distributed as dist
        loc = self._torch.from_numpy(np.ascontiguousarray(self.get(name, time_level)))
        parts = [self._torch.empty_like(loc) for _ in range(self.cfg.world_size)]
        if dist.get_backend(self.group) == "nccl":
            parts = [p.cuda() for p in parts]
            dist.all_gather(parts, loc.cuda(), group=self.group)
            parts = [p.cpu() for p in parts]
        else:
            dist.all_gather(parts, loc, group=self.group)
        return np.concatenate([p.numpy() for p in parts], axis=-2)

    # ---- restart files on a sharded run (the reference writes per-PE fragments and combines them with
    #      mppnccombine, experiment.py:318-323; here rank 0 writes the combined files directly)
    def gather_spectral(self, name, time_level=1):
        """global (lev, n, m) array on every rank: each rank contributes the wavenumbers it owns"""
        import torch.distributed as dist
        loc = self.get(name, time_level)                       # zeros where m is not ours
        t = self._torch.from_numpy(np.ascontiguousarray(loc).view(np.float64))
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
            dist.all_reduce(t, group=self.group)
            t = t.cpu()
        else:
            dist.all_reduce(t, group=self.group)               # x + 0 is exact
        return t.numpy().view(np.complex128).reshape(loc.shape)

    def write_restart(self, directory: str):
        from . import restart
        view = _GatheredView(self)
        if self.cfg.rank == 0:
            restart.write_restart(view, directory)

    def read_restart(self, directory: str):
        """Every rank reads the combined files through a single-rank handle on its own GPU (which also rebuilds
        the derived grid fields) and keeps its latitude band / wavenumber set."""
        from . import restart
        c1 = dyncore.default_config()
        for f, _ in self.cfg._fields_:
            setattr(c1, f, getattr(self.cfg, f))
        c1.rank, c1.world_size, c1.stream = 0, 1, None
        one = dyncore.DynCore(c1)
        one.tracer_names = list(self.tracer_names)
        try:
            restart.read_restart(one, directory)
            self.set_time_pointers(one.info("previous"), one.info("current"), one.info("step"))
            self.set_surf_geopotential(one.get("surf_geopotential"))       # the file's topography (global; every rank keeps its band)
            j0, jl = self.info("lat_start"), self.Jl
            levels = (0, 1) if one.info("previous") != one.info("current") else (1,)
            for tl in levels:
                for nm in ("vors", "divs", "ts", "ln_ps"):
                    self.set(nm, one.get(nm, tl), tl)
                names = ("ug", "vg", "tg", "psg") + (("tr", "tr_atm") if self.info("tracer") else ())
                names += tuple(f"{nm}{k + 1}" for k in range(1, self.cfg.num_tracers if self.info("tracer") else 0) for nm in ("tr", "tr_atm"))
                for nm in names:
                    self.set(nm, one.get(nm, tl)[..., j0:j0 + jl, :], tl)
            for nm in ("vorg", "divg", "dxT", "dyT", "dxlp", "dylp", "wg_full") + (("t_surf",) if self.cfg.physics == 1 else ()):
                self.set(nm, one.get(nm)[..., j0:j0 + jl, :])
        finally:
            one.close()


class _GatheredView:
    """What restart.write_restart needs from a handle, answered with globally gathered arrays."""

    def __init__(self, sh: ShardedDynCore):
        import types
        self._sh = sh
        ntr = sh.cfg.num_tracers if sh.info("tracer") else 0
        self.cfg = types.SimpleNamespace(world_size=1, physics=sh.cfg.physics, num_tracers=ntr, tracer_spectral=[0] * max(ntr, 1))
        self.tracer_names = list(sh.tracer_names)        # a sharded run carries grid tracers only
        self.L, self.J, self.Jl, self.I, self.N1, self.M1 = sh.L, sh.J, sh.J, sh.I, sh.N1, sh.M1
        self._cache = {}
        for nm in ("vors", "divs", "ts", "ln_ps"):
            for tl in (0, 1):
                self._cache[nm, tl] = sh.gather_spectral(nm, tl)
        grids = ["ug", "vg", "tg", "psg", "vorg", "divg", "wg_full", "surf_geopotential"] + (["tr", "tr_atm"] if sh.info("tracer") else []) + \
                [f"{nm}{k + 1}" for k in range(1, ntr) for nm in ("tr", "tr_atm")] + \
                (["t_surf"] if sh.cfg.physics == 1 else [])
        for nm in grids:
            for tl in (0, 1):
                self._cache[nm, tl] = sh.gather_grid(nm, tl)

    def info(self, k):
        return self._sh.info(k)

    def table(self, k):
        return self._sh.table(k)

    def get(self, name, time_level=1):
        return self._cache[name, time_level]
