"""ocean_topog_smoothing /= 0: the regularisation of a topography over the ocean (Lindberg & Broccoli 1996) -- host mirror of the reference's
topog_regularization_mod (src/atmos_spectral/init/topog_regularization.F90: compute_lambda :75-150, regularize :153-290), which get_topography
calls for topography_option = 'input' / 'interpolated' (init/spectral_init_cond.F90:236-245, 285-295).  The arithmetic is the library's
(isca_amd/csrc/topog.cpp: isca_topog_compute_lambda / isca_topog_regularize, the transforms and global means on the device); this module is
the module's two public names with numpy arguments."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .dyncore import DynCore, IscaError, _dptr


def _args(core: DynCore, ocean_mask, unsmoothed):
    ocean = np.ascontiguousarray(np.asarray(ocean_mask, dtype=bool), dtype=np.float64)
    u = np.ascontiguousarray(unsmoothed, dtype=np.float64)
    if ocean.shape != (core.J, core.I):
        raise IscaError(f"topog_regularization_init: Input argument ocean_mask has incorrect dimensions. shape(ocean_mask)={ocean.shape}  Should be {(core.J, core.I)}")
    if u.shape != ocean.shape:
        raise IscaError(f"regularize: Input argument unsmoothed_field has incorrect dimensions. shape(unsmoothed_field)={u.shape}  Should be {ocean.shape}")
    return ocean, u


def regularize(core: DynCore, lam: float, ocean_mask, unsmoothed):
    """regularize(lambda, ocean_mask, unsmoothed_field, smoothed_field, fraction_smoothed) -> (smoothed field, fraction smoothed)"""
    ocean, u = _args(core, ocean_mask, unsmoothed)
    out, frac = np.zeros_like(u), C.c_double()
    core._check(core.lib.isca_topog_regularize(core._h, float(lam), _dptr(ocean), _dptr(u), _dptr(out), C.cast(C.byref(frac), C.POINTER(C.c_double))))
    return out, frac.value


def compute_lambda(core: DynCore, ocean_topog_smoothing: float, ocean_mask, unsmoothed):
    """compute_lambda(ocean_topog_smoothing, ocean_mask, unsmoothed_field, lambda, actual_fraction_smoothed) -> (lambda, fraction smoothed)"""
    ocean, u = _args(core, ocean_mask, unsmoothed)
    lam, frac = C.c_double(), C.c_double()
    core._check(core.lib.isca_topog_compute_lambda(core._h, float(ocean_topog_smoothing), _dptr(ocean), _dptr(u), C.cast(C.byref(lam), C.POINTER(C.c_double)),
                                                   C.cast(C.byref(frac), C.POINTER(C.c_double))))
    return lam.value, frac.value
