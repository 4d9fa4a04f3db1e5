"""ocean_topog_smoothing /= 0: the regularisation of a topography over the ocean (Lindberg & Broccoli 1996), mirror of the reference's
topog_regularization_mod (src/atmos_spectral/init/topog_regularization.F90: compute_lambda :75-150, regularize :153-290,
topog_regularization_init :292-365), which get_topography calls for topography_option = 'input' / 'interpolated'
(init/spectral_init_cond.F90:236-245, 285-295).  Initialisation-time host arithmetic on the (n, m) coefficients; every transform and
global mean is the device's (DynCore.trans_grid_to_spherical / trans_spherical_to_grid / area_weighted_global_mean)."""
from __future__ import annotations

import numpy as np

from .dyncore import DynCore, IscaError

ITMAX, TOLERANCE = 1000, 1.0e-5            # topog_regularization.F90:57-58
TOL_LAMBDA, ITMAX_LAMBDA = 0.001, 20       # :84-85


class _Setup:
    """topog_regularization_init (:292-365): D(m,n) = mean over the OCEAN points of w_j P_mn(j)^2, the total wavenumbers, the Lanczos-like factors"""

    def __init__(self, core: DynCore, ocean_mask):
        if core.cfg.world_size != 1:
            raise IscaError("regularize: not coded for a decomposed grid (regularize, topog_regularization.F90:172-175)")
        self.core = core
        J, I, N1, M1 = core.J, core.I, core.N1, core.M1
        self.ocean = np.asarray(ocean_mask, dtype=bool)
        if self.ocean.shape != (J, I):
            raise IscaError(f"topog_regularization_init: Input argument ocean_mask has incorrect dimensions. shape(ocean_mask)={self.ocean.shape}  Should be {(J, I)}")
        nf = core.cfg.num_fourier
        leg2 = core.table("legendre") ** 2                        # (J/2, n, m); P(-x)^2 = P(x)^2: one hemisphere's table serves both
        wts = core.table("wts_lat")
        cnt = self.ocean.sum(axis=1).astype(np.float64)           # ocean points per latitude row
        w = wts * cnt
        wh = w[: J // 2] + w[::-1][: J // 2]                       # rows j and lat_max + 1 - j share a table entry
        self.D = np.tensordot(wh, leg2, axes=(0, 0)) / I
        n, m = np.meshgrid(np.arange(N1), np.arange(M1), indexing="ij")
        self.LL = ((n + m) * (n + m + 1)).astype(np.float64)      # (n + m)(n + m + 1)
        self.keep = (n <= min(nf, core.cfg.num_spherical)).astype(np.float64)      # n = ns..nmax, nmax = min(num_fourier, ne)
        facm = np.pi * np.arange(M1) / (2.0 * nf)
        self.sfac = np.ones(M1)
        self.sfac[1:] = np.sin(facm[1:]) / facm[1:]
        self.sfac = self.sfac[None, :]


def regularize(core: DynCore, lam: float, ocean_mask, unsmoothed, _setup: _Setup | None = None):
    """regularize (:153-290): (smoothed field, fraction smoothed) for the regularisation parameter `lam`"""
    S = _setup or _Setup(core, ocean_mask)
    ocean = S.ocean
    u = np.ascontiguousarray(unsmoothed, dtype=np.float64)
    if u.shape != ocean.shape:
        raise IscaError(f"regularize: Input argument unsmoothed_field has incorrect dimensions. shape(unsmoothed_field)={u.shape}  Should be {ocean.shape}")
    H = S.keep / (1.0 + lam * S.D * S.LL ** 2)
    b = core.trans_grid_to_spherical(u)
    a = S.keep * b / (1.0 + lam * S.LL ** 2)                       # (equation 6.3)
    dela = S.LL * a
    rough = core.trans_spherical_to_grid(dela)
    converg, cost = 1.0, 0.0
    smoothed = None
    for it in range(1, ITMAX + 1):
        if abs(converg) < TOLERANCE:
            break
        rough = np.where(ocean, rough, 0.0)                        # rough is zeroed out over land
        dr2 = S.LL * core.trans_grid_to_spherical(rough) * S.keep
        a = (a + H * (b - a) - lam * H * dr2) * S.sfac             # (m = 0: factor 1)
        smoothed = core.trans_spherical_to_grid(a)
        dela = S.LL * a * S.keep
        rough = core.trans_spherical_to_grid(dela)
        cost_field = np.where(ocean, (u - smoothed) ** 2 + lam * rough ** 2, 0.0)      # (equation 6.4)
        oldcost, cost = cost, core.area_weighted_global_mean(cost_field)
        if it > 1:
            converg = (oldcost - cost) / oldcost
    else:
        raise IscaError("regularize: Failure to converge")
    rb = core.trans_spherical_to_grid(S.LL * b * S.keep)
    lamcosti = core.area_weighted_global_mean(np.where(ocean, rb ** 2, 0.0))
    ra = core.trans_spherical_to_grid(dela)
    lamcost = core.area_weighted_global_mean(np.where(ocean, ra ** 2, 0.0))
    return smoothed, 1.0 - lamcost / lamcosti


def compute_lambda(core: DynCore, ocean_topog_smoothing: float, ocean_mask, unsmoothed):
    """compute_lambda (:75-150): the regularisation parameter for which regularize smooths the wanted fraction (secant iteration from 1e-7, 2e-7)"""
    S = _Setup(core, ocean_mask)
    want = float(ocean_topog_smoothing)
    l1, l2 = 1.0e-7, 2.0e-7
    _, f1 = regularize(core, l1, ocean_mask, unsmoothed, S)
    if abs(want - f1) < TOL_LAMBDA:
        return l1, f1
    _, f2 = regularize(core, l2, ocean_mask, unsmoothed, S)
    if abs(want - f2) < TOL_LAMBDA:
        return l2, f2
    if f1 > want or f2 > want:
        raise IscaError("compute_lambda: Iterative scheme for computing lambda may not work unless initial values of lambda_1 and lambda_2 are reduced.")
    l1 = ((f2 - want) * l1 + (want - f1) * l2) / (f2 - f1)
    if l1 < 0.0:
        raise IscaError("compute_lambda: Iterative scheme for finding lambda will not work unless initial values of lambda_1 and lambda_2 are reduced.")
    _, f1 = regularize(core, l1, ocean_mask, unsmoothed, S)
    for it in range(1, ITMAX_LAMBDA + 1):
        if abs(want - f1) < TOL_LAMBDA:
            return l1, f1
        l2 = ((f2 - want) * l1 + (want - f1) * l2) / (f2 - f1)
        if l2 < 0.0:
            raise IscaError(f"compute_lambda: Iterative scheme for finding lambda failed. lambda went negative on iteration number{it:8d}")
        _, f2 = regularize(core, l2, ocean_mask, unsmoothed, S)
        if abs(want - f2) < TOL_LAMBDA:
            return l2, f2
        l1 = ((f2 - want) * l1 + (want - f1) * l2) / (f2 - f1)
        _, f1 = regularize(core, l1, ocean_mask, unsmoothed, S)
    raise IscaError("compute_lambda: Cannot converge on a value of lambda. Perhaps more interations are needed.")
