"""TEST INFRASTRUCTURE ONLY: numpy restatement of the reference's two sibling cores, on the transforms / damping / van Leer
routines of oracle/isca_oracle.py (SpectralCore with one level).  Never imported by the product.

  ShallowOracle     src/atmos_spectral_shallow: shallow_dynamics.F90:217-533 (init :320-408, step :411-470, implicit_correction
                    :476-495, tracers :497-533), shallow_physics.F90:106-194, time-level bookkeeping of atmosphere.F90:164-197
  BarotropicOracle  src/atmos_spectral_barotropic: barotropic_dynamics.F90:175-372 (physics is empty)
Pinned against the reference's own runs (tests/golden/shallow_run_T21.npz, barotropic_run_T21.npz) in tests/test_oracle_vs_golden.py.
"""
import math

import numpy as np

from .isca_oracle import Config, SpectralCore



def _leapfrog(a, dt_a, prev, cur, fut, delta_t, robert):
    """leapfrog_3d_complex with raw_filter_coeff = 1 (leapfrog.F90:217-247); a = [level0, level1]."""
    pc = a[prev] - 2.0 * a[cur]
    if prev == cur:
        a[fut] = a[prev] + delta_t * dt_a
        a[cur] = a[cur] + robert * (pc + a[fut])
    else:
        a[cur] = a[cur] + robert * pc
        a[fut] = a[prev] + delta_t * dt_a
        a[cur] = a[cur] + robert * a[fut]


class _Sibling:
    def __init__(self, res, dt_atmos, damping_coeff=1.e-4, damping_order=4, robert_coeff=0.04, radius=6376.0e3, omega=7.2921150e-5):
        table = {"T21": (64, 32, 21, 22), "T42": (128, 64, 42, 43)}
        lon, lat, nf, ns = table[res]
        self.sc = SpectralCore(Config(lon_max=lon, lat_max=lat, num_fourier=nf, num_spherical=ns, num_levels=1, dt_atmos=dt_atmos,
                                      damping_coeff=damping_coeff, damping_order=damping_order, robert_coeff=robert_coeff, radius=radius,
                                      omega=omega))
        self.radius, self.omega = radius, omega
        self.dt, self.robert = float(dt_atmos), robert_coeff
        self.J, self.I = lat, lon
        self.coriolis = (2 * omega * self.sc.sin_lat)[:, None]
        self.previous = self.current = 0
        self.damping_r = 0.0
        self.stir_amp = 0.0

    def stirring_init(self, amplitude, decay_time=172800.0, lat0=45.0, lon0=180.0, widthy=12.0, widthx=45.0, B=0.0, n_max=15, n_min=9,
                      zonal_min=3):
        """stirring_init (atmos_spectral_barotropic/stirring.F90:75-170)."""
        sc = self.sc
        self.stir_amp = amplitude
        self.astir = math.sqrt(1.0 - math.exp(-2 * self.dt / decay_time))
        self.bstir = math.exp(-self.dt / decay_time)
        self.wave_mask = np.zeros((sc.N1, sc.M1), dtype=bool)
        for m in range(zonal_min + 1, n_max):
            for n in range(n_min + 1 - m, n_max - m):
                if 0 <= n < sc.N1 and 0 <= m < sc.M1:
                    self.wave_mask[n, m] = True
        xx = sc.deg_lon - lon0
        xx = xx - 360.0 * np.rint(xx / 360.0)
        self.localize = np.exp(-.5 * ((sc.deg_lat - lat0) / widthy) ** 2)[:, None] * (1 + B * np.exp(-.5 * (xx / widthx) ** 2))[None, :]
        self.s_stir = np.zeros((sc.N1, sc.M1), dtype=np.complex128)

    def stirring(self, dt_vors, ran):
        """stirring (:190-224) with the uniform numbers `ran` [2, n, m] the reference drew."""
        new = np.where(self.wave_mask, self.stir_amp * self.astir * ((2 * ran[0] - 1) + 1j * (2 * ran[1] - 1)), 0.0)
        new = self.g2s(self.localize * self.s2g(new))
        new[0, 0] = 0.0
        self.s_stir = self.bstir * self.s_stir + new
        return dt_vors + self.s_stir

    # 2-D wrappers of the (level, lat, lon) routines
    def g2s(self, g):
        return self.sc.trans_grid_to_spherical(g[None])[0]

    def s2g(self, s):
        return self.sc.trans_spherical_to_grid(s[None])[0]

    def uv_from_vd(self, vor, div):
        u, v = self.sc.uv_grid_from_vor_div(vor[None], div[None])
        return u[0], v[0]

    def vd_from_uv(self, u, v):
        vor, div = self.sc.vor_div_from_uv_grid(u[None], v[None])
        return vor[0], div[0]

    def hadv(self, spec, u, v, tend):
        return self.sc.horizontal_advection(spec[None], u[None], v[None], tend[None])[0]

    def damp(self, spec_prev, dt_spec, delta_t):
        d = self.sc.damping + self.damping_r
        return (1.0 / (1.0 + d * delta_t)) * (dt_spec - d * spec_prev)

    def tracer_bands(self):
        tr = np.zeros((self.J, self.I))
        lat = self.sc.deg_lat
        tr[(lat > 10.0) & (lat < 20.0)] = 1.0
        tr[lat > 70.0] = -1.0
        return tr

    def _levels(self):
        first = self.previous == self.current
        fut = 1 - self.current if first else self.previous
        return self.previous, self.current, fut, (self.dt if first else 2.0 * self.dt)

    def _tracers(self, p, c, f, delta_t, robert_grid):
        u, v = self.u[c], self.v[c]
        dt_tr = self.hadv(self.trss[c], u, v, np.zeros((self.J, self.I)))                 # update_spec_tracer
        dt_trs = self.damp(self.trss[p], self.g2s(dt_tr), delta_t)
        _leapfrog(self.trss, dt_trs, p, c, f, delta_t, self.robert)
        self.trs[f] = self.s2g(self.trss[f])
        q = self.tr[p]                                                                     # update_grid_tracer (zero source)
        fut = q + delta_t * self.sc.a_grid_horiz_advection(u[None], v[None], q[None], delta_t, np.zeros((1, self.J, self.I)))[0]
        self.tr[c] = self.tr[c] + robert_grid * (self.tr[p] + fut - 2.0 * self.tr[c])
        self.tr[f] = fut

    def stream(self):
        eig = self.sc.eigen_laplacian
        inv = np.where(eig != 0.0, -1.0 / np.where(eig != 0.0, eig, 1.0), 0.0)
        return self.s2g(inv * self.vors[self.previous])


class ShallowOracle(_Sibling):
    def __init__(self, res="T21", dt_atmos=1200.0, h_0=3.e4, u_deep_mag=0.0, n_merid_deep_flow=3.0, u_upper_mag_init=0.0,
                 add_initial_vortex_pair=False, robert_coeff_tracer=0.04, fric_damp_time=-20.0, therm_damp_time=-10.0, phys_h_0=3.e4,
                 h_amp=2.e4, h_lon=90.0, h_lat=25.0, h_width=15.0, h_itcz=1.e5, itcz_width=4.0, **kw):
        super().__init__(res, dt_atmos, **kw)
        self.h_0, self.robert_tracer = h_0, robert_coeff_tracer
        sc = self.sc
        lat, lon = sc.deg_lat[:, None], sc.deg_lon[None, :]
        fd = -fric_damp_time * 86400 if fric_damp_time < 0 else fric_damp_time
        td = -therm_damp_time * 86400 if therm_damp_time < 0 else therm_damp_time
        self.kappa_m = 1.0 / fd if fd != 0 else 0.0
        self.kappa_t = 1.0 / td if td != 0 else 0.0
        xx, yy = (lon - h_lon) / (h_width * 2.0), (lat - h_lat) / h_width
        self.h_eq = phys_h_0 + h_amp * np.maximum(1.e-10, np.exp(-(xx * xx + yy * yy))) + h_itcz * np.exp(-(lat / itcz_width) ** 2)
        d2r, nm = math.pi / 180.0, n_merid_deep_flow
        la = d2r * lat
        deep = -2. * self.omega * u_deep_mag * self.radius * (1. / (1. - nm ** 2)) * (
            -np.cos(nm * la) * np.cos(la) - nm * (np.sin(nm * la) * np.sin(la) - math.sin(nm * (2. * math.atan(1.)))))
        deep = np.repeat(deep, self.I, axis=1)
        self.deep = deep - sc.area_weighted_global_mean(deep)
        # initial state (:330-408), vortex pair as a height anomaly
        h = h_0 - self.deep
        vor = np.repeat(-((u_upper_mag_init * nm) / self.radius) * np.sin(la), self.I, axis=1)
        if add_initial_vortex_pair:
            def rad(lon0, lat0):
                return np.sqrt(np.minimum((lon - lon0) ** 2, (lon - lon0 - 360.) ** 2) + (lat - lat0) ** 2) / 5.0
            rc, ra = rad(0.0, 60.0), rad(180.0, 60.0)
            h = np.where(rc <= 2.0, h + 0.1 * -h_0 * np.exp(-rc ** 2), np.where(ra <= 2.0, h + 0.1 * h_0 * np.exp(-ra ** 2), h))
        two = lambda a: [a.copy(), a.copy()]
        self.vors, self.divs, self.hs = two(self.g2s(vor)), two(self.g2s(np.zeros_like(vor))), two(self.g2s(h))
        u, v = self.uv_from_vd(self.vors[0], self.divs[0])
        self.u, self.v, self.vor, self.div, self.h = two(u), two(v), two(vor), two(np.zeros_like(vor)), two(h)
        tr = self.tracer_bands()
        self.tr, self.trs, self.trss = two(tr), two(tr), two(self.g2s(tr))

    def step(self, ran=None):
        p, c, f, delta_t = self._levels()
        u, v = self.u[c], self.v[c]
        vorg = self.vor[c] + self.coriolis
        tu = (0.0 - self.kappa_m * self.u[p]) + vorg * v
        tv = (0.0 - self.kappa_m * self.v[p]) - vorg * u
        th = self.hadv(self.hs[c], u, v, 0.0 - self.kappa_t * (self.h[p] - self.h_eq)) - self.h[c] * self.div[c]
        dt_vors, dt_divs = self.vd_from_uv(tu, tv)
        dt_hs = self.g2s(th)
        bs = self.g2s(self.h[c] + self.deep + 0.5 * (u * u + v * v))
        eig = self.sc.eigen_laplacian
        dt_divs = dt_divs - (-eig * bs)
        mu = 0.5 * delta_t                                               # implicit_correction
        dt_hs = dt_hs + self.h_0 * (self.divs[c] - self.divs[p])
        dt_divs = dt_divs - eig * (self.hs[c] - self.hs[p])
        dt_divs = (dt_divs + mu * eig * dt_hs) / (1.0 + mu * mu * eig * self.h_0)
        dt_hs = dt_hs - mu * self.h_0 * dt_divs
        dt_vors = self.damp(self.vors[p], dt_vors, delta_t)
        dt_divs = self.damp(self.divs[p], dt_divs, delta_t)
        dt_hs = self.damp(self.hs[p], dt_hs, delta_t)
        if self.stir_amp != 0.0:
            dt_vors = self.stirring(dt_vors, ran)
        self.pv = vorg / self.h[c]
        for a, d in ((self.vors, dt_vors), (self.divs, dt_divs), (self.hs, dt_hs)):
            _leapfrog(a, d, p, c, f, delta_t, self.robert)
        self.vor[f], self.div[f] = self.s2g(self.vors[f]), self.s2g(self.divs[f])
        self._tracers(p, c, f, delta_t, self.robert_tracer)
        self.u[f], self.v[f] = self.uv_from_vd(self.vors[f], self.divs[f])
        self.h[f] = self.s2g(self.hs[f])
        self.previous, self.current = c, f


class BarotropicOracle(_Sibling):
    def __init__(self, res="T21", dt_atmos=1200.0, zeta_0=8.e-5, m_0=4, eddy_width=15.0, eddy_lat=45.0, damping_coeff_r=0.0,
                 initial_zonal_wind="two_jets", **kw):
        super().__init__(res, dt_atmos, **kw)
        self.damping_r = damping_coeff_r
        sc = self.sc
        cl, sl = sc.cos_lat, sc.sin_lat
        self.zonal_u_init = (25.0 * cl - 30.0 * cl ** 3 + 300.0 * sl ** 2 * cl ** 6) if initial_zonal_wind == "two_jets" else 0.0 * cl
        u = np.repeat(self.zonal_u_init[:, None], self.I, axis=1)
        vors, _ = self.vd_from_uv(u, np.zeros_like(u))
        vor = self.s2g(vors)
        yy = (sc.deg_lat[:, None] - eddy_lat) / eddy_width
        rad_lon = sc.deg_lon[None, :] * math.atan(1.0) / 45.0
        vor = vor + 0.5 * zeta_0 * cl[:, None] * np.exp(-yy * yy) * np.cos(m_0 * rad_lon)
        two = lambda a: [a.copy(), a.copy()]
        self.vors = two(self.g2s(vor))
        self.zero = np.zeros_like(self.vors[0])
        u, v = self.uv_from_vd(self.vors[0], self.zero)
        self.u, self.v, self.vor = two(u), two(v), two(vor)
        tr = self.tracer_bands()
        self.tr, self.trs, self.trss = two(tr), two(tr), two(self.g2s(tr))

    def step(self, ran=None):
        p, c, f, delta_t = self._levels()
        self.pv = self.vor[c] + self.coriolis
        tu, tv = 0.0 + self.pv * self.v[c], 0.0 - self.pv * self.u[c]
        dt_vors, _ = self.vd_from_uv(tu, tv)
        dt_vors = self.damp(self.vors[p], dt_vors, delta_t)
        if self.stir_amp != 0.0:
            dt_vors = self.stirring(dt_vors, ran)
        _leapfrog(self.vors, dt_vors, p, c, f, delta_t, self.robert)
        self.vor[f] = self.s2g(self.vors[f])
        self._tracers(p, c, f, delta_t, self.robert)
        self.u[f], self.v[f] = self.uv_from_vd(self.vors[f], self.zero)
        self.previous, self.current = c, f
