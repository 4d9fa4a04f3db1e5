"""CPU restatement (numpy, fp64) of the reference's spectral dynamical-core hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module, and only as the checker.  The product (isca_amd/) never imports it.

Pinned against the reference itself: oracle/make_golden.py runs oracle/_ref/ref_harness.x (the
reference's own Fortran, compiled in place by oracle/build_ref.py) and commits its outputs as
tests/golden/*.npz; tests/test_oracle_vs_golden.py checks every function here against them --
the public routines one by one, the default step, and one reference run per option Config carries
(damping options, RAW filter, mcm differencing, vertical coordinates, rhomboidal truncation,
fourier_inc, vert_advect_uv/t, use_implicit, make_symmetric, use_virtual_temperature, topography,
no_forcing as zero coefficients, further field_table tracers with robert_coeff / hole_filling /
tracer_sms / advect_vert, lon_max with factors 3 and 5).

Array conventions.  The reference is Fortran column-major: grid (lon, lat, lev), spectral
(m, n, lev) with n the meridional index (total wavenumber = m + n).  Here arrays are the SAME
memory viewed by numpy in C order, i.e. grid[lev, lat, lon] and spec[lev, n, m]; 2-D fields drop
the lev axis.  Latitudes run south to north.  All citations are relative to /root/reference/src.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# constants: shared/constants/constants.F90:83-86,250-254
RADIUS = 6376.0e3
OMEGA = 7.2921150e-5
GRAV = 9.80
RDGAS = 287.04
RVGAS = 461.50
KAPPA = 2.0 / 7.0
CP_AIR = RDGAS / KAPPA
PI = 3.14159265358979323846


@dataclass
class Config:
    """Namelist keys actually used on the path (spectral_dynamics.F90:152-224, hs_forcing.F90:76-122)."""
    lon_max: int = 64
    lat_max: int = 32
    num_fourier: int = 21
    num_spherical: int = 22
    num_levels: int = 25
    dt_atmos: float = 600.0
    # spectral_dynamics_nml (values of exp/test_cases/held_suarez/held_suarez_test_case.py:45-98)
    damping_order: int = 4
    damping_coeff: float = 1.15740741e-4
    damping_option: str = "resolution_dependent"
    cutoff_wn: int = 15
    damping_coeff_vor: float = -1.0
    damping_coeff_div: float = -1.0
    damping_order_vor: int = -1
    damping_order_div: int = -1
    robert_coeff: float = 0.04
    raw_filter_coeff: float = 1.0
    alpha_implicit: float = 0.5
    reference_sea_level_press: float = 1.0e5
    scale_heights: float = 6.0
    exponent: float = 7.5
    surf_res: float = 0.5
    vert_coord_option: str = "uneven_sigma"                  # or 'input' with pk_input, bk_input (num_levels + 1 values each)
    pk_input: tuple = ()
    bk_input: tuple = ()
    p_press: float = 0.1          # vert_coord_option = 'hybrid' (spectral_dynamics.F90:180-181)
    p_sigma: float = 0.3
    vert_difference_option: str = "simmons_and_burridge"     # or 'mcm' (spectral_dynamics.F90:1084, press_and_geopot.F90:196, implicit.F90:404, 447)
    vert_advect_uv: str = "second_centered"       # spectral_dynamics.F90:280-301, 877-888: centred schemes on the current level, finite-volume ones on the previous
    vert_advect_t: str = "second_centered"
    use_implicit: bool = True                     # :906: .false. skips implicit_correction
    triang_trunc: bool = True                     # .false.: rhomboidal (every m keeps n = 0..num_spherical-1; transforms.F90:516-520, 773-780)
    fourier_inc: int = 1                          # zonal wavenumbers 0, inc, 2 inc, .. on a 360/inc degree sector
    use_virtual_temperature: bool = False         # :857-871, press_and_geopot.F90:246-256, 340-348 (q = tracer 1)
    make_symmetric: bool = False                  # spherical.F90:185: the truncation mask also drops every m > 0
    do_mass_correction: bool = True
    do_energy_correction: bool = True
    do_water_correction: bool = True
    water_correction_limit: float = 200.0e2
    eddy_sponge_coeff: float = 0.0
    zmu_sponge_coeff: float = 0.0
    zmv_sponge_coeff: float = 0.0
    initial_temperature: float = 264.0
    valid_range_t: tuple = (100.0, 800.0)
    initial_sphum: float = 0.0
    radius: float = 6376.0e3      # constants_nml
    omega: float = 7.2921150e-5
    num_tracers: int = 1          # the dry field_table carries one grid tracer (sphum)
    # further field_table entries after sphum (update_tracers' loop, spectral_dynamics.F90:1132-1183): dicts with
    # kind = 'grid' | 'spectral', robert_coeff (None: the namelist's), hole_filling (spectral only: water_borrowing.F90),
    # advect_vert (None: grid finite_volume_parabolic, spectral second_centered),
    # sms = (flux, sink) of the entry's tracer_sms method (hs_forcing.F90:251-261; 'off' / 'none' = (0, 0); None inside: the namelist's)
    extra_tracers: tuple = ()
    sphum_sms: tuple = None       # the same for tracer 1
    sphum_advect_vert: str = "finite_volume_parabolic"
    # hs_forcing_nml
    t_zero: float = 315.0
    t_strat: float = 200.0
    delh: float = 60.0
    delv: float = 10.0
    eps: float = 0.0
    sigma_b: float = 0.7
    ka: float = -40.0
    ks: float = -4.0
    kf: float = -1.0
    do_conserve_energy: bool = True
    trflux: float = 1.0e-5
    trsink: float = -4.0
    P00: float = 1.0e5
    local_heating_option: str = ""          # '' or 'Isidoro' (hs_forcing.F90:87-94, 728-769)
    local_heating_srfamp: float = 0.0
    local_heating_xwidth: float = 10.0
    local_heating_ywidth: float = 10.0
    local_heating_xcenter: float = 180.0
    local_heating_ycenter: float = 45.0
    local_heating_vert_decay: float = 1.0e4

    @staticmethod
    def resolution(name: str, num_levels: int, **kw) -> "Config":
        # src/extra/python/isca/experiment.py:29-57
        table = {"T21": (64, 32, 21, 22), "T42": (128, 64, 42, 43),
                 "T85": (256, 128, 85, 86), "T170": (512, 256, 170, 171),
                 "T31": (96, 48, 31, 32), "T53": (160, 80, 53, 54)}        # lon_max with factors 3 and 5 (fft99's set99)
        lon, lat, nf, ns = table[name]
        return Config(lon_max=lon, lat_max=lat, num_fourier=nf, num_spherical=ns,
                      num_levels=num_levels, **kw)


# --------------------------------------------------------------------------------------------
# tables
# --------------------------------------------------------------------------------------------
def compute_gaussian(n_hem: int):
    """atmos_spectral/tools/gauss_and_legendre.F90:111-183 (Newton iteration, pole-most first)."""
    converg = 0.1 ** 15            # precision(real*8) = 15
    n = 2 * n_hem
    sin_hem = np.zeros(n_hem)
    wts_hem = np.zeros(n_hem)
    for i in range(1, n_hem + 1):
        z = math.cos(PI * (i - 0.25) / (n + 0.5))
        for _ in range(10):
            p1, p2 = 1.0, 0.0
            for j in range(1, n + 1):
                p3 = p2
                p2 = p1
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j
            pp = n * (z * p1 - p2) / (z * z - 1.0)
            z1 = z
            z = z1 - p1 / pp
            if abs(z - z1) < converg:
                break
        else:
            raise RuntimeError("abscissas failed to converge")
        sin_hem[i - 1] = z
        wts_hem[i - 1] = 2.0 / ((1.0 - z * z) * pp * pp)
    return sin_hem, wts_hem


def compute_legendre(num_fourier: int, num_spherical: int, sin_hem: np.ndarray):
    """gauss_and_legendre.F90:47-108, fourier_inc=1.  Returns leg[j, n, m] (Fortran (m,n,j))."""
    M1, N1 = num_fourier + 1, num_spherical + 1
    m = np.arange(M1, dtype=np.float64)[None, :]
    n = np.arange(N1, dtype=np.float64)[:, None]
    l2 = (m + n) ** 2
    m2 = m ** 2 + 0 * n
    with np.errstate(invalid="ignore", divide="ignore"):
        eps = np.sqrt((l2 - m2) / (4.0 * l2 - 1.0))          # eps[n, m]
    leg = np.zeros((len(sin_hem), N1, M1))
    b = np.zeros(M1)
    for mm in range(1, M1):
        b[mm] = math.sqrt(0.5 * (2.0 * mm + 1.0) / mm)
    for j, s in enumerate(sin_hem):
        c = math.sqrt(1 - s * s)
        poly = np.zeros((N1, M1))
        poly[0, 0] = math.sqrt(0.5)
        for mm in range(1, M1):
            poly[0, mm] = b[mm] * c * poly[0, mm - 1]
        poly[1, :] = s * poly[0, :] / eps[1, :]
        for nn in range(2, N1):
            poly[nn, :] = (s * poly[nn - 1, :] - eps[nn - 1, :] * poly[nn - 2, :]) / eps[nn, :]
        leg[j] = poly
    return leg


def compute_vert_coord(option, num_levels, scale_heights, surf_res, exponent, p_press, p_sigma, reference_press):
    """compute_vert_coord (init/vert_coordinate.F90:89-157) for the options formed from numbers: returns (pk, bk)."""
    if option == "even_sigma":                                                # :231-246
        return np.zeros(num_levels + 1), np.arange(num_levels + 1) / float(num_levels)
    if option == "uneven_sigma":
        return compute_uneven_sigma(num_levels, scale_heights, surf_res, exponent)
    if option == "hybrid":                                                     # :139-145 with transition(), :162-184
        _, b_sigma = compute_uneven_sigma(num_levels, scale_heights, surf_res, exponent, zero_top=False)
        a_sigma = np.zeros_like(b_sigma)
        b_press, a_press = np.zeros_like(b_sigma), b_sigma.copy()             # the same profile as a pure pressure coordinate
        x = (b_sigma - p_press) / (p_sigma - p_press)
        f = np.where(b_sigma <= p_press, 0.0, np.where(b_sigma >= p_sigma, 1.0, np.sin(0.5 * PI * x) ** 2))
        a = a_sigma * f + a_press * (1.0 - f)
        b = b_sigma * f + b_press * (1.0 - f)
        return reference_press * a, b
    if option == "mcm":                                                        # compute_old_model_sigma :309-322
        assert num_levels == 14
        return np.zeros(15), np.array([0.0, .03, .0707, .1311, .2102, .3036, .4062, .5138, .6226, .7284, .8255, .9066, .9640, .9933, 1.0])
    if option == "v197":                                                       # :290-306
        assert num_levels == 18
        return np.zeros(19), np.array([0.0, .0089163, .0342936, .0740741, .1262002, .1886145, .2592592, .3360768, .4170096, .5000000, .5829904,
                                       .6639231, .7407407, .8113854, .8737997, .9259259, .9657064, .9910837, 1.0])
    raise NotImplementedError(option)


def compute_uneven_sigma(num_levels, scale_heights, surf_res, exponent, zero_top=True):
    """atmos_spectral/init/vert_coordinate.F90:248-273; returns (pk, bk)."""
    b = np.zeros(num_levels + 1)
    s2 = 1.0 - surf_res
    for k in range(1, num_levels + 1):
        zeta = 1.0 - float(k - 1) / float(num_levels)
        z = surf_res * zeta + s2 * (zeta ** exponent)
        b[k - 1] = math.exp(-z * scale_heights)
    b[num_levels] = 1.0
    if zero_top:
        b[0] = 0.0
    return np.zeros(num_levels + 1), b


def invert_gauss_jordan(a):
    """model/matrix_invert.F90:38-130 restated as a plain inverse (fp64, pivoted)."""
    return np.linalg.inv(a)


class SpectralCore:
    """Tables + operators + the time step of spectral_dynamics_mod / atmosphere_mod (HS branch)."""

    def __init__(self, cfg: Config):
        self.cfg = c = cfg
        self.I, self.J, self.L = c.lon_max, c.lat_max, c.num_levels
        self.M1, self.N1 = c.num_fourier + 1, c.num_spherical + 1
        I, J, M1, N1 = self.I, self.J, self.M1, self.N1
        # --- Gaussian grid: spherical_fourier.F90:397-431 ---
        self.sin_hem, self.wts_hem = compute_gaussian(J // 2)
        self.sin_lat = np.concatenate([-self.sin_hem, self.sin_hem[::-1]])
        self.wts_lat = np.concatenate([self.wts_hem, self.wts_hem[::-1]])
        self.cos_lat = np.sqrt(1 - self.sin_lat * self.sin_lat)
        self.cosm_lat = 1.0 / self.cos_lat
        self.deg_lat = np.arcsin(self.sin_lat) * 180.0 / PI
        self.deg_lon = np.arange(I) * 360.0 / I                       # grid_fourier.F90:109-118
        self.rad_lat = self.deg_lat * PI / 180.0                      # atmosphere.F90:248-251
        # --- Legendre: spherical_fourier.F90:376-394 ---
        inc = c.fourier_inc                                                  # zonal wavenumbers 0, inc, 2 inc, ..: every inc-th column of the table (:100-104)
        self.legendre = compute_legendre(c.num_fourier * inc, c.num_spherical, self.sin_hem)[:, :, ::inc]   # [j,n,m]
        self.legendre_wts = self.legendre * self.wts_hem[:, None, None]
        # --- spherical.F90:137-216 ---
        m = inc * np.arange(M1, dtype=np.float64)[None, :] + 0 * np.arange(N1)[:, None]
        n = np.arange(N1, dtype=np.float64)[:, None] + 0 * m
        Lw = m + n
        self.fourier_wave, self.spherical_wave = m, Lw
        self.triangle_mask = np.where(Lw > c.num_spherical - 1, 0.0, 1.0)
        if not c.triang_trunc:                                               # rhomboidal_truncation without arguments: the row n = num_spherical (spherical.F90:620)
            self.triangle_mask = np.ones_like(Lw); self.triangle_mask[N1 - 1, :] = 0.0
        if c.make_symmetric:                                                 # spherical.F90:185
            self.triangle_mask = np.where(m > 0, 0.0, self.triangle_mask)
        with np.errstate(invalid="ignore", divide="ignore"):
            eps = np.sqrt((Lw ** 2 - m ** 2) / (4.0 * Lw ** 2 - 1.0))
            self.epsilon = eps
            self.eigen_laplacian = Lw * (Lw + 1.0) / (self.cfg.radius * self.cfg.radius)
            self.coef_uvm = np.where(Lw > 0, -self.cfg.radius * eps / np.where(Lw > 0, Lw, 1), 0.0)
            self.coef_uvc = np.where(Lw > 0, -self.cfg.radius * m / np.where(Lw > 0, Lw * (Lw + 1.0), 1), 0.0)
        z = np.zeros((N1, M1))
        self.coef_uvp = z.copy(); self.coef_alpp = z.copy(); self.coef_dyp = z.copy()
        self.coef_uvp[:-1] = -self.cfg.radius * eps[1:] / (Lw[:-1] + 1.0)
        self.coef_alpm = (Lw + 1.0) * eps / self.cfg.radius
        self.coef_alpp[:-1] = Lw[:-1] * eps[1:] / self.cfg.radius
        self.coef_dym = (Lw - 1.0) * eps / self.cfg.radius
        self.coef_dx = m / self.cfg.radius
        self.coef_dyp[:-1] = (Lw[:-1] + 2.0) * eps[1:] / self.cfg.radius
        # --- vertical coordinate + derived (spectral_dynamics.F90:456-462) ---
        if c.vert_coord_option == "input":                                  # vert_coordinate_nml's pk, bk (init/vert_coordinate.F90:139-146)
            self.pk, self.bk = np.asarray(c.pk_input, dtype=np.float64), np.asarray(c.bk_input, dtype=np.float64)
            assert self.pk.size == self.L + 1 and self.bk.size == self.L + 1
        else:
            self.pk, self.bk = compute_vert_coord(c.vert_coord_option, self.L, c.scale_heights, c.surf_res, c.exponent, c.p_press, c.p_sigma,
                                                  c.reference_sea_level_press)
        self.dpk = self.pk[1:] - self.pk[:-1]
        self.dbk = self.bk[1:] - self.bk[:-1]
        self.coriolis = 2 * self.cfg.omega * self.sin_lat                      # spectral_dynamics.F90:445
        # --- damping: spectral_damping.F90:124-156 ---
        eig = self.eigen_laplacian
        cv = c.damping_coeff if c.damping_coeff_vor < 0 else c.damping_coeff_vor
        cd = c.damping_coeff if c.damping_coeff_div < 0 else c.damping_coeff_div
        ov = c.damping_order if c.damping_order_vor < 0 else c.damping_order_vor
        od = c.damping_order if c.damping_order_div < 0 else c.damping_order_div
        ref = eig[c.num_spherical - 1, 0]
        self.damping_exponential = c.damping_option == "exponential_cutoff"
        self.damping_coeffs = {"t": c.damping_coeff, "vor": cv, "div": cd}
        if c.damping_option == "resolution_dependent":
            self.damping = c.damping_coeff * ((eig / ref) ** c.damping_order)
            self.damping_vor = cv * ((eig / ref) ** ov)
            self.damping_div = cd * ((eig / ref) ** od)
        elif c.damping_option == "exponential_cutoff":                      # spectral_damping.F90:129-146
            se, cut = np.sqrt(eig), eig[c.cutoff_wn, 0]
            d = np.where(eig / cut > 1, ((se - np.sqrt(cut)) / (np.sqrt(ref) - np.sqrt(cut))) ** c.damping_order, 0.0)
            self.damping, self.damping_vor, self.damping_div = d, d.copy(), d.copy()
        elif c.damping_option == "resolution_independent":
            self.damping = c.damping_coeff * (eig ** c.damping_order)
            self.damping_vor = cv * (eig ** ov)
            self.damping_div = cd * (eig ** od)
        else:
            raise ValueError(c.damping_option)
        self.damping_eddy_sponge = c.eddy_sponge_coeff * eig
        self.damping_zmu_sponge = c.zmu_sponge_coeff * eig[:, 0]
        self.damping_zmv_sponge = c.zmv_sponge_coeff * eig[:, 0]
        # --- implicit: implicit.F90:79-217 ---
        self.num_total_wavenumbers = c.num_spherical - 1 + (0 if c.triang_trunc else c.fourier_inc * c.num_fourier)     # spectral_dynamics.F90:430-434
        self._implicit_init()
        self._wave_dt = None
        # --- HS constants: hs_forcing.F90:391-410 ---
        self.tka = -1.0 / (86400 * c.ka) if c.ka < 0 else c.ka
        self.tks = -1.0 / (86400 * c.ks) if c.ks < 0 else c.ks
        self.vkf = -1.0 / (86400 * c.kf) if c.kf < 0 else c.kf
        self.trsink = -86400.0 * c.trsink if c.trsink < 0 else c.trsink
        self.surf_geopotential = np.zeros((J, I))
        self.step_count = 0

    # ----------------------------------------------------------------------------------------
    # transforms (tools/transforms.F90:379-533, spherical_fourier.F90:177-339, grid_fourier.F90)
    # ----------------------------------------------------------------------------------------
    def spherical_to_fourier(self, s):
        """spherical_fourier.F90:214-258.  s[...,n,m] -> f[...,j,m] (all J latitudes, S->N)."""
        P = self.legendre
        xe = np.einsum("...nm,jnm->...jm", s[..., 0::2, :], P[:, 0::2, :])
        xo = np.einsum("...nm,jnm->...jm", s[..., 1::2, :], P[:, 1::2, :])
        south = xe - xo            # row j  (southern hemisphere, pole-most first)
        north = xe + xo            # row J+1-j
        return np.concatenate([south, north[..., ::-1, :]], axis=-2)

    def fourier_to_spherical(self, f):
        """spherical_fourier.F90:300-336.  f[...,j,m] -> s[...,n,m] (full rectangle)."""
        Jh = self.J // 2
        fs = f[..., :Jh, :]
        fn = f[..., ::-1, :][..., :Jh, :]
        xe = fn + fs
        xo = fn - fs
        W = self.legendre_wts
        s = np.zeros(f.shape[:-2] + (self.N1, self.M1), dtype=np.complex128)
        s[..., 0::2, :] = np.einsum("...jm,jnm->...nm", xe, W[:, 0::2, :])
        s[..., 1::2, :] = np.einsum("...jm,jnm->...nm", xo, W[:, 1::2, :])
        return s

    def grid_to_fourier(self, g):
        """grid_fourier.F90:129-152 + fft.F90:578-586: c(k) = (1/I) sum_j x(j) exp(-2 pi i jk/I)."""
        return np.fft.rfft(g, axis=-1) / self.I          # [..., 0:I/2+1]

    def fourier_to_grid(self, f):
        """grid_fourier.F90:155-179: x(j) = sum_k c(k) exp(+2 pi i jk/I), Hermitian symmetry."""
        return np.fft.irfft(f, n=self.I, axis=-1) * self.I

    def trans_spherical_to_grid(self, s):
        f = self.spherical_to_fourier(s)
        full = np.zeros(f.shape[:-1] + (self.I // 2 + 1,), dtype=np.complex128)
        full[..., : self.M1] = f                       # transforms.F90:424 zero above trunc_fourier
        return self.fourier_to_grid(full)

    def trans_grid_to_spherical(self, g, do_truncation=True):
        f = self.grid_to_fourier(g)[..., : self.M1]
        s = self.fourier_to_spherical(f)
        if do_truncation:
            s = s * self.triangle_mask               # spherical.F90:579-581
        return s

    # ----------------------------------------------------------------------------------------
    # spectral operators (tools/spherical.F90:270-600)
    # ----------------------------------------------------------------------------------------
    @staticmethod
    def _i_times(x):
        return 1j * x         # cmplx(-aimag, real)

    def compute_lon_deriv_cos(self, s):
        return self.coef_dx * self._i_times(s)

    def compute_lat_deriv_cos(self, s):
        d = np.zeros_like(s)
        d[..., 1:, :] = -s[..., :-1, :] * self.coef_dym[1:]
        d[..., :-1, :] += s[..., 1:, :] * self.coef_dyp[:-1]
        return d

    def compute_gradient_cos(self, s):
        return self.compute_lon_deriv_cos(s), self.compute_lat_deriv_cos(s)

    def compute_laplacian(self, s):
        return s * (-self.eigen_laplacian)

    def compute_ucos_vcos(self, vor, div):
        u = self.coef_uvc * self._i_times(div)
        v = self.coef_uvc * self._i_times(vor)
        u[..., 1:, :] += self.coef_uvm[1:] * vor[..., :-1, :]
        v[..., 1:, :] -= self.coef_uvm[1:] * div[..., :-1, :]
        u[..., :-1, :] -= self.coef_uvp[:-1] * vor[..., 1:, :]
        v[..., :-1, :] += self.coef_uvp[:-1] * div[..., 1:, :]
        return u, v

    def compute_alpha_operator(self, a, b, isign):
        al = self.coef_dx * self._i_times(a)
        al[..., 1:, :] -= isign * self.coef_alpm[1:] * b[..., :-1, :]
        al[..., :-1, :] += isign * self.coef_alpp[:-1] * b[..., 1:, :]
        return al

    def compute_vor_div(self, ucos, vcos):
        return self.compute_alpha_operator(vcos, ucos, -1), self.compute_alpha_operator(ucos, vcos, +1)

    def divide_by_cos(self, g):
        return g * self.cosm_lat[:, None]

    def uv_grid_from_vor_div(self, vor, div):
        """transforms.F90:700-719."""
        us, vs = self.compute_ucos_vcos(vor, div)
        return (self.divide_by_cos(self.trans_spherical_to_grid(us)),
                self.divide_by_cos(self.trans_spherical_to_grid(vs)))

    def vor_div_from_uv_grid(self, u, v):
        """transforms.F90:742-783 (triang=.true.)."""
        dx = self.trans_grid_to_spherical(self.divide_by_cos(u), do_truncation=False)
        dy = self.trans_grid_to_spherical(self.divide_by_cos(v), do_truncation=False)
        vor, div = self.compute_vor_div(dx, dy)
        return vor * self.triangle_mask, div * self.triangle_mask

    def horizontal_advection(self, field_spec, u, v, tendency):
        """transforms.F90:808-831."""
        dx, dy = self.compute_gradient_cos(field_spec)
        dxg = self.divide_by_cos(self.trans_spherical_to_grid(dx))
        dyg = self.divide_by_cos(self.trans_spherical_to_grid(dy))
        return tendency - u * dxg - v * dyg

    def area_weighted_global_mean(self, field2d):
        """transforms.F90:1059-1077."""
        return float(np.sum(self.wts_lat[:, None] * field2d) / (np.sum(self.wts_lat) * self.I))

    def mass_weighted_global_integral(self, field, ps):
        """model/global_integral.F90:49-81."""
        p_half = self.pk[:, None, None] + self.bk[:, None, None] * ps
        dp = p_half[1:] - p_half[:-1]
        return self.area_weighted_global_mean(np.sum(field * dp, axis=0)) / GRAV

    # ----------------------------------------------------------------------------------------
    # column routines
    # ----------------------------------------------------------------------------------------
    def pressure_variables(self, ps):
        """model/press_and_geopot.F90:152-221 (simmons_and_burridge, pk(1)=bk(1)=0 branch too)."""
        ps = np.asarray(ps, dtype=np.float64)
        sh = (self.L + 1,) + ps.shape
        p_half = self.pk.reshape((-1,) + (1,) * ps.ndim) + self.bk.reshape((-1,) + (1,) * ps.ndim) * ps
        ln_p_half = np.zeros(sh)
        ln_p_full = np.zeros((self.L,) + ps.shape)
        if self.cfg.vert_difference_option == "mcm":            # :196-210
            p_full = 0.5 * (p_half[1:] + p_half[:-1])
            k0 = 1 if (self.pk[0] == 0.0 and self.bk[0] == 0.0) else 0
            ln_p_half[k0:] = np.log(p_half[k0:])
            return p_half, ln_p_half, p_full, np.log(p_full)
        if self.pk[0] == 0.0 and self.bk[0] == 0.0:
            ln_p_half[1:] = np.log(p_half[1:])
            for k in range(1, self.L):
                alpha = 1.0 - p_half[k] * (ln_p_half[k + 1] - ln_p_half[k]) / (p_half[k + 1] - p_half[k])
                ln_p_full[k] = ln_p_half[k + 1] - alpha
            ln_p_full[0] = ln_p_half[1] - 1.0
            ln_p_half[0] = 0.0
        else:
            ln_p_half[:] = np.log(p_half)
            for k in range(self.L):
                alpha = 1.0 - p_half[k] * (ln_p_half[k + 1] - ln_p_half[k]) / (p_half[k + 1] - p_half[k])
                ln_p_full[k] = ln_p_half[k + 1] - alpha
        p_full = np.exp(ln_p_full)
        return p_half, ln_p_half, p_full, ln_p_full

    def compute_geopotential(self, t, ln_p_half, ln_p_full, q=None):
        """press_and_geopot.F90:314-359; with use_virtual_temperature and q given, T (1 + (rvgas/rdgas - 1) q) (:340-348)."""
        L = self.L
        if self.cfg.use_virtual_temperature and q is not None:
            t = t * (1. + (RVGAS / RDGAS - 1.) * q)
        gh = np.zeros((L + 1,) + t.shape[1:])
        gh[L] = self.surf_geopotential if t.ndim == 3 else 0.0
        ktop = 1 if self.pk[0] == 0.0 else 0
        for k in range(L - 1, ktop - 1, -1):
            gh[k] = gh[k + 1] + RDGAS * t[k] * (ln_p_half[k + 1] - ln_p_half[k])
        gf = gh[1:] + RDGAS * t * (ln_p_half[1:] - ln_p_full)
        return gf, gh

    def compute_pressures_and_heights(self, t, ps, q=None):
        """press_and_geopot.F90:363-387."""
        p_half, ln_p_half, p_full, ln_p_full = self.pressure_variables(ps)
        zf, zh = self.compute_geopotential(t, ln_p_half, ln_p_full, q)
        return zf / GRAV, zh / GRAV, p_full, p_half

    def four_in_one(self, divg, u, v, t, ps, ln_p_half, ln_p_full, p_full, dx_ps, dy_ps,
                    dt_ps, dt_t, dt_u, dt_v):
        """spectral_dynamics.F90:1038-1112 (simmons_and_burridge).  Returns updated tendencies + wg."""
        L = self.L
        dmean_tot = np.zeros_like(ps)
        wg = np.zeros((L + 1,) + ps.shape)
        wg_full = np.zeros((L,) + ps.shape)
        dt_t = dt_t.copy(); dt_u = dt_u.copy(); dt_v = dt_v.copy()
        for k in range(L):
            dp = self.dpk[k] + self.dbk[k] * ps
            dp_inv = 1 / dp
            dlog_1 = ln_p_half[k + 1] - ln_p_full[k]
            dlog_2 = ln_p_full[k] - ln_p_half[k]
            dlog_3 = ln_p_half[k + 1] - ln_p_half[k]
            mcm = self.cfg.vert_difference_option == "mcm"      # :1084-1099
            x1 = (self.bk[k + 1] * dlog_1 + self.bk[k] * dlog_2) * dp_inv
            x2 = dx_ps * (1.0 / ps) if mcm else x1 * dx_ps
            x3 = dy_ps * (1.0 / ps) if mcm else x1 * dy_ps
            dt_u[k] = dt_u[k] - RDGAS * t[k] * x2
            dt_v[k] = dt_v[k] - RDGAS * t[k] * x3
            dmean = divg[k] * dp + self.dbk[k] * (u[k] * dx_ps + v[k] * dy_ps)
            x4 = (dmean_tot + 0.5 * dmean) / p_full[k] if mcm else (dmean_tot * dlog_3 + dmean * dlog_1) * dp_inv
            x5 = x4 - u[k] * x2 - v[k] * x3
            dt_t[k] = dt_t[k] - KAPPA * t[k] * x5
            wg_full[k] = -x5 * p_full[k]
            dmean_tot = dmean_tot + dmean
            wg[k + 1] = -dmean_tot
        dt_ps = dt_ps - dmean_tot
        for k in range(1, L):
            wg[k] = wg[k] + dmean_tot * self.bk[k]
        wg[0] = 0.0
        wg[L] = 0.0
        return dt_ps, wg, wg_full, dt_t, dt_u, dt_v

    @staticmethod
    def vert_advection_second_centered(w, dz, r):
        """atmos_shared/vert_advection/vert_advection.F90:162-193,467-470 (ADVECTIVE_FORM)."""
        L = r.shape[0]
        flux = np.zeros_like(w)
        flux[0] = w[0] * r[0]
        flux[L] = w[L] * r[L - 1]
        flux[1:L] = w[1:L] * (0.5 * (r[1:] + r[:-1]))
        return -(flux[1:] - flux[:-1] - r * (w[1:] - w[:-1])) / dz

    # ----------------------------------------------------------------------------------------
    # Held-Suarez forcing (atmos_param/hs_forcing/hs_forcing.F90:148-272,508-724)
    # ----------------------------------------------------------------------------------------
    def hs_forcing(self, dt, p_half, p_full, u, v, t, tr=None, tr_dt=None, sms=None):
        c = self.cfg
        ps = p_half[-1]
        rps = 1.0 / ps
        sigma = p_full * rps
        bl = (sigma <= 1.0) & (sigma > c.sigma_b)
        # rayleigh_damping :615-679
        vcoeff = -self.vkf / (1.0 - c.sigma_b)
        vfactr = np.where(bl, vcoeff * (sigma - c.sigma_b), 0.0)
        utnd = vfactr * u
        vtnd = vfactr * v
        tdt = np.zeros_like(t)
        if c.do_conserve_energy:                                           # :198-200
            tdt = tdt + (-((u + 0.5 * utnd * dt) * utnd + (v + 0.5 * vtnd * dt) * vtnd) / CP_AIR)
        # newtonian_damping :508-611
        sin_lat = np.sin(self.rad_lat)[:, None]
        sin_lat_2 = sin_lat * sin_lat
        cos_lat_2 = 1.0 - sin_lat_2
        cos_lat_4 = cos_lat_2 * cos_lat_2
        t_star = c.t_zero - c.delh * sin_lat_2 - c.eps * sin_lat
        tstr = c.t_strat - c.eps * sin_lat
        tcoeff = (self.tks - self.tka) / (1.0 - c.sigma_b)
        p_norm = p_full / c.P00
        the = t_star - c.delv * cos_lat_2 * np.log(p_norm)
        teq = np.maximum(the * p_norm ** KAPPA, tstr)
        tdamp = np.where(bl, self.tka + cos_lat_4 * (tcoeff * (sigma - c.sigma_b)), self.tka)
        tdt = tdt + (-tdamp * (t - teq))
        if c.local_heating_option == "Isidoro":                             # local_heating :233-238, :750-764
            rad = np.pi / 180.0
            xw, yw, xc, yc = c.local_heating_xwidth * rad, c.local_heating_ywidth * rad, c.local_heating_xcenter * rad, c.local_heating_ycenter * rad
            xc = xc - 2 * np.pi * np.floor(xc / (2 * np.pi))               # hs_forcing_init :378-380
            srfamp = c.local_heating_srfamp / 86400.0
            lon = np.arange(self.I) * 360.0 / self.I * rad
            lon = lon - 2 * np.pi * np.floor(lon / (2 * np.pi))
            lon_factor = np.exp(-0.5 * ((lon - xc) / xw) ** 2)[None, :]
            lat_factor = np.exp(-0.5 * ((self.rad_lat - yc) / yw) ** 2)[:, None]
            tdt = tdt + srfamp * lon_factor * lat_factor * np.exp((p_full - ps) / c.local_heating_vert_decay)
        elif c.local_heating_option != "":
            raise ValueError('"%s"  is not a valid value for local_heating_option' % c.local_heating_option)
        out = [utnd, vtnd, tdt]
        if tr is not None:                                                   # :240-263, 683-724
            rst = tr + dt * tr_dt
            flux, rdamp = c.trflux, self.trsink
            if sms is not None:                                              # the entry's tracer_sms: (flux, sink), :251-261 (None: the namelist's)
                flux = c.trflux if sms[0] is None else sms[0]
                rdamp = c.trsink if sms[1] is None else sms[1]
                rdamp = -86400.0 * rdamp if rdamp < 0 else rdamp             # tracer_source_sink :697-698
            rdamp = 1.0 / rdamp if rdamp > 0 else 0.0
            source = np.zeros_like(tr)
            source[-1] = flux / (p_half[-1] - p_half[-2])
            out.append(tr_dt + source - rdamp * rst)
        return tuple(out)

    # ----------------------------------------------------------------------------------------
    # semi-implicit (model/implicit.F90)
    # ----------------------------------------------------------------------------------------
    def _pressure_variables_1d(self, ps):
        ph, lph, pf, lpf = self.pressure_variables(np.array(ps))
        return ph, lph, pf, lpf

    def linear_tp_tendency(self, div):
        """implicit.F90:414-480.  div[k, ...] complex -> (dt_p_surf[...], dt_t[k, ...])."""
        L = self.L
        t_ref = self.ref_temperature_implicit
        dmean_tot = np.zeros(div.shape[1:], dtype=div.dtype)
        dt_t = np.zeros_like(div)
        vert_vel = np.zeros((L + 1,) + div.shape[1:], dtype=div.dtype)
        for k in range(L):
            dp = self.dpk[k] + self.dbk[k] * self.ref_surf_p_implicit
            dp_inv = 1 / dp
            dlog_1 = self.ref_ln_p_half[k + 1] - self.ref_ln_p_full[k]
            dlog_3 = self.ref_ln_p_half[k + 1] - self.ref_ln_p_half[k]
            dmean = div[k] * dp
            if self.cfg.vert_difference_option == "mcm":        # :447-456
                p_full_ref = 0.5 * (self.pk[k + 1] + self.pk[k]) + 0.5 * (self.bk[k + 1] + self.bk[k]) * self.ref_surf_p_implicit
                dt_t[k] = -(KAPPA * t_ref[k] / p_full_ref) * (dmean_tot + 0.5 * dmean)
            else:
                dt_t[k] = -KAPPA * t_ref[k] * (dmean_tot * dlog_3 + dmean * dlog_1) * dp_inv
            dmean_tot = dmean_tot + dmean
            vert_vel[k + 1] = -dmean_tot
        dt_p_surf = -dmean_tot
        temp = np.zeros_like(vert_vel)
        for k in range(1, L):
            vert_vel[k] = vert_vel[k] + dmean_tot * self.bk[k]
            temp[k] = -vert_vel[k] * (t_ref[k] - t_ref[k - 1])
        for k in range(L):
            dp = self.dpk[k] + self.dbk[k] * self.ref_surf_p_implicit
            dt_t[k] = dt_t[k] + 0.5 * (1 / dp) * (temp[k + 1] + temp[k])
        return dt_p_surf, dt_t

    def linear_geopotential(self, del_t, del_ln_p_half, del_ln_p_full):
        """implicit.F90:329-359."""
        L = self.L
        t = self.ref_temperature_implicit
        lph, lpf = self.ref_ln_p_half, self.ref_ln_p_full
        gh = np.zeros((L + 1,) + del_t.shape[1:], dtype=del_t.dtype)
        for k in range(L - 1, 0, -1):
            gh[k] = gh[k + 1] + RDGAS * (del_t[k] * (lph[k + 1] - lph[k])
                                         + t[k] * (del_ln_p_half[k + 1] - del_ln_p_half[k]))
        g = np.zeros_like(del_t)
        for k in range(L):
            g[k] = gh[k + 1] + RDGAS * (del_t[k] * (lph[k + 1] - lpf[k])
                                        + t[k] * (del_ln_p_half[k + 1] - del_ln_p_full[k]))
        return g

    def _implicit_init(self):
        c = self.cfg
        L = self.L
        self.ref_temperature_implicit = np.full(L, 300.0)          # spectral_dynamics.F90:473
        self.ref_surf_p_implicit = c.reference_sea_level_press
        pref = self.ref_surf_p_implicit
        _, self.ref_ln_p_half, _, self.ref_ln_p_full = self._pressure_variables_1d(pref)
        del_ln_p_half = np.zeros(L + 1)
        del_ln_p_half[1:] = self.bk[1:] / (self.pk[1:] + self.bk[1:] * pref)
        del_ln_p_half[0] = 1.0 / pref if self.pk[0] == 0.0 else self.bk[0] / (self.pk[0] + self.bk[0] * pref)
        eps = 1.0e-5
        _, _, _, l1 = self._pressure_variables_1d(pref * (1.0 - 0.5 * eps))
        _, _, _, l2 = self._pressure_variables_1d(pref * (1.0 + 0.5 * eps))
        del_ln_p_full = (l2 - l1) / (eps * pref)
        # build_matrix :171-217
        ident = np.eye(L)
        nu = np.zeros(L); tau = np.zeros((L, L)); gamma = np.zeros((L, L))
        for k in range(L):
            dt_p, dt_t = self.linear_tp_tendency(ident[:, k].copy())
            nu[k] = -dt_p
            tau[:, k] = -dt_t
            gamma[:, k] = self.linear_geopotential(ident[:, k].copy(), np.zeros(L + 1), np.zeros(L))
        t = self.ref_temperature_implicit
        dlog_1 = self.ref_ln_p_half[1:] - self.ref_ln_p_full
        dlog_2 = self.ref_ln_p_full - self.ref_ln_p_half[:-1]
        h1 = RDGAS * t * (self.bk[1:] * dlog_1 + self.bk[:-1] * dlog_2) / (self.dpk + self.dbk * pref)
        if c.vert_difference_option == "mcm":                   # pres_grad_funct :404-408
            h1 = RDGAS * t / pref
        h2 = self.linear_geopotential(np.zeros(L), del_ln_p_half, del_ln_p_full)
        self.h = h1 + h2
        self.div_mat = np.outer(self.h, nu) + gamma @ tau
        self.nu_vec, self.tau_mat, self.gamma_mat = nu, tau, gamma

    def build_wave_matrices(self, dt):
        """implicit.F90:221-237."""
        self.xi = dt * self.cfg.alpha_implicit
        ntw = self.num_total_wavenumbers
        L = self.L
        self.wave_matrix = np.zeros((ntw + 1, L, L))
        for Lw in range(ntw + 1):
            factor = self.xi * self.xi * Lw * (Lw + 1) / self.cfg.radius ** 2
            self.wave_matrix[Lw] = invert_gauss_jordan(np.eye(L) + factor * self.div_mat)
        self._wave_dt = dt

    def implicit_correction(self, dt_divs, dt_ts, dt_ln_ps, divs, ts, ln_ps, dt, previous, current):
        """implicit.F90:241-325.  divs/ts: [2][k,n,m]; ln_ps: [2][n,m].  Returns new tendencies."""
        if self._wave_dt != dt:
            self.build_wave_matrices(dt)
        xi, pref = self.xi, self.ref_surf_p_implicit
        # adjust_dt_divs :289-325
        dps, dts = self.linear_tp_tendency(divs[previous] - divs[current])
        dt_ts = dt_ts + dts
        dt_ln_ps = dt_ln_ps + dps / pref
        ts_temp = ts[previous] - ts[current] + xi * dt_ts
        ps_temp = ln_ps[previous] - ln_ps[current] + xi * dt_ln_ps
        zero_h = np.zeros((self.L + 1,) + ts_temp.shape[1:], dtype=ts_temp.dtype)
        geopot = self.linear_geopotential(ts_temp, zero_h, np.zeros_like(ts_temp))
        dt_divs = dt_divs + self.eigen_laplacian * (geopot + self.h[:, None, None] * ps_temp * pref)
        # per-(m,n) L x L matvec :268-277
        Lw = self.spherical_wave.astype(int)
        ntw = self.num_total_wavenumbers
        out = dt_divs.copy()
        ok = Lw <= ntw
        Wsel = self.wave_matrix[np.where(ok, Lw, 0)]              # [n,m,L,L]
        prod = np.einsum("nmab,bnm->anm", Wsel, dt_divs)
        out = np.where(ok[None], prod, dt_divs)
        dps, dts = self.linear_tp_tendency(out)
        dt_ts = dt_ts + xi * dts
        dt_ln_ps = dt_ln_ps + xi * dps / pref
        return out, dt_ts, dt_ln_ps

    # ----------------------------------------------------------------------------------------
    # damping + leapfrog (spectral_damping.F90:172-291, leapfrog.F90:58-105)
    # ----------------------------------------------------------------------------------------
    def compute_spectral_damping(self, spec, dt_spec, dt, kind="t"):
        d = {"t": self.damping, "vor": self.damping_vor, "div": self.damping_div}[kind]
        if self.damping_exponential:                                         # spectral_damping.F90:186-190
            d = (np.exp(np.log(dt * self.damping_coeffs[kind] + 1.0) * d) - 1.0) / dt
        coeff = 1.0 / (1.0 + d * dt)
        out = coeff * (dt_spec - d * spec)
        if kind in ("vor", "div"):
            es = self.damping_eddy_sponge
            zs = self.damping_zmu_sponge if kind == "vor" else self.damping_zmv_sponge
            top = out[0].copy()
            top[:, 1:] = (top[:, 1:] - es[:, 1:] * spec[0][:, 1:]) / (1.0 + es[:, 1:] * dt)
            top[:, 0] = (top[:, 0] - zs * spec[0][:, 0]) / (1.0 + zs * dt)
            out[0] = top
        return out

    # ----------------------------------------------------------------------------------------
    # model state and the step
    # ----------------------------------------------------------------------------------------
    def cold_start(self):
        """init/spectral_initialize_fields.F90:45-135 + spectral_dynamics.F90:580-630."""
        c = self.cfg
        L, J, I, N1, M1 = self.L, self.J, self.I, self.N1, self.M1
        st = {}
        vors = np.zeros((L, N1, M1), dtype=np.complex128)
        divs = np.zeros_like(vors)
        for (m, n) in ((1, 3), (5, 3), (1, 2), (5, 2)):
            if m < M1 and n < N1:
                vors[L - 3:L, n, m] = 1.0e-7
        ug, vg = self.uv_grid_from_vor_div(vors, divs)
        tg = np.full((L, J, I), c.initial_temperature)
        ln_psg = math.log(c.reference_sea_level_press) - self.surf_geopotential / (RDGAS * c.initial_temperature)
        ts = self.trans_grid_to_spherical(tg)
        tg = self.trans_spherical_to_grid(ts)
        ln_ps = self.trans_grid_to_spherical(ln_psg)
        psg = np.exp(self.trans_spherical_to_grid(ln_ps))
        vors, divs = self.vor_div_from_uv_grid(ug, vg)
        ug, vg = self.uv_grid_from_vor_div(vors, divs)
        self.vorg = self.trans_spherical_to_grid(vors)
        self.divg = self.trans_spherical_to_grid(divs)
        two = lambda a: [a.copy(), a.copy()]
        self.vors, self.divs, self.ts, self.ln_ps = two(vors), two(divs), two(ts), two(ln_ps)
        self.ug, self.vg, self.tg, self.psg = two(ug), two(vg), two(tg), two(psg)
        tr = np.full((L, J, I), c.initial_sphum)
        self.tr = two(tr)
        self.tr_atm = two(tr)
        # tracers other than sphum start from zero (allocate_fields :661-662; only sphum/mix_rat get initial_sphum, :581-584)
        self.xtr = []
        for spec in c.extra_tracers:
            z = np.zeros((L, J, I))
            x = dict(kind=spec.get("kind", "grid"), rc=spec.get("robert_coeff"), holes=bool(spec.get("hole_filling", False)), sms=spec.get("sms"), vert=spec.get("advect_vert"),
                     g=two(z), atm=two(z))
            if x["rc"] is None:
                x["rc"] = c.robert_coeff
            if x["kind"] == "spectral":
                x["s"] = two(self.trans_grid_to_spherical(z))                  # :627-628
            self.xtr.append(x)
        self.previous = 0
        self.current = 0
        self.step_count = 0
        # atmosphere_init :229-241
        self.p_full = [None, None]; self.p_half = [None, None]
        self.z_full = [None, None]; self.z_half = [None, None]
        self._pressures_and_heights(0)
        self.p_full[1], self.p_half[1] = self.p_full[0].copy(), self.p_half[0].copy()
        self.z_full[1], self.z_half[1] = self.z_full[0].copy(), self.z_half[0].copy()
        self.wg_full = np.zeros((L, J, I))

    def _pressures_and_heights(self, lev):
        zf, zh, pf, ph = self.compute_pressures_and_heights(self.tg[lev], self.psg[lev], self.tr_atm[lev])     # atmosphere.F90:229-241, 331-338
        self.z_full[lev], self.z_half[lev], self.p_full[lev], self.p_half[lev] = zf, zh, pf, ph

    def step(self, with_tracer=True):
        """One call of atmosphere (driver/solo/atmosphere.F90:276-352), HS branch, i.e.
        hs_forcing -> spectral_dynamics (spectral_dynamics.F90:780-1034) -> pressures/heights."""
        c = self.cfg
        prev, cur = self.previous, self.current
        delta_t = c.dt_atmos if prev == cur else 2 * c.dt_atmos
        fut = 1 - cur if prev == cur else prev
        # --- physics: u,v,T at PREVIOUS, p at CURRENT (atmosphere.F90:304-311)
        if with_tracer:
            # atmosphere_mod keeps its OWN grid_tracers copy (atmosphere.F90:95,326), which only ever receives
            # the new level (spectral_dynamics.F90:1028): physics sees the un-Robert-filtered tracer
            dt_u, dt_v, dt_t, dt_tr = self.hs_forcing(delta_t, self.p_half[cur], self.p_full[cur], self.ug[prev],
                                                      self.vg[prev], self.tg[prev], self.tr_atm[prev], np.zeros_like(self.tr[prev]), sms=c.sphum_sms)
        else:
            dt_u, dt_v, dt_t = self.hs_forcing(delta_t, self.p_half[cur], self.p_full[cur],
                                               self.ug[prev], self.vg[prev], self.tg[prev])
        for x in self.xtr if with_tracer else ():                            # hs_forcing's loop over rdt(:,:,:,n), hs_forcing.F90:248-266
            x["dt"] = self.hs_forcing(delta_t, self.p_half[cur], self.p_full[cur], self.ug[prev], self.vg[prev], self.tg[prev],
                                      x["atm"][prev], np.zeros_like(x["g"][prev]), sms=x["sms"])[3]
        dt_ps = np.zeros((self.J, self.I))
        # --- initialize_corrections :1306-1338
        if c.do_mass_correction:
            mean_ps_prev = self.area_weighted_global_mean(self.psg[prev])
        if c.do_energy_correction:
            energy = 0.5 * ((self.ug[prev] + dt_u * delta_t) ** 2 + (self.vg[prev] + dt_v * delta_t) ** 2) \
                + CP_AIR * (self.tg[prev] + dt_t * delta_t)
            mean_energy_prev = self.mass_weighted_global_integral(energy, self.psg[prev])
        if with_tracer and c.do_water_correction:
            mean_water_prev = self.mass_weighted_global_integral(self.tr[prev] + delta_t * dt_tr, self.psg[prev])
        # --- dynamics tendencies :853-904
        p_half, ln_p_half, p_full, ln_p_full = self.pressure_variables(self.psg[cur])
        dxs, dys = self.compute_gradient_cos(self.ln_ps[cur])
        dx_ps = self.divide_by_cos(self.psg[cur] * self.trans_spherical_to_grid(dxs))
        dy_ps = self.divide_by_cos(self.psg[cur] * self.trans_spherical_to_grid(dys))
        u, v, t = self.ug[cur], self.vg[cur], self.tg[cur]
        tv = t * (1.0 + (RVGAS / RDGAS - 1.0) * self.tr[cur]) if (c.use_virtual_temperature and with_tracer) else t     # :857-861
        dt_ps, wg, wg_full, dt_t, dt_u, dt_v = self.four_in_one(
            self.divg, u, v, tv, self.psg[cur], ln_p_half, ln_p_full, p_full, dx_ps, dy_ps,
            dt_ps, dt_t, dt_u, dt_v)
        phig_full, _ = self.compute_geopotential(t, ln_p_half, ln_p_full, self.tr[cur] if with_tracer else None)        # :866-871
        dt_ln_ps = self.trans_grid_to_spherical(dt_ps / self.psg[cur])
        dp = p_half[1:] - p_half[:-1]
        lev_uv = cur if c.vert_advect_uv.endswith("centered") else prev     # :877-888
        lev_t = cur if c.vert_advect_t.endswith("centered") else prev
        dt_u = dt_u + self.vert_advection(c.vert_advect_uv, delta_t, wg, dp, self.ug[lev_uv])
        dt_v = dt_v + self.vert_advection(c.vert_advect_uv, delta_t, wg, dp, self.vg[lev_uv])
        dt_t = dt_t + self.vert_advection(c.vert_advect_t, delta_t, wg, dp, self.tg[lev_t])
        dt_t = self.horizontal_advection(self.ts[cur], u, v, dt_t)
        dt_ts = self.trans_grid_to_spherical(dt_t)
        absvor = self.vorg + self.coriolis[None, :, None]
        dt_u = dt_u + absvor * v
        dt_v = dt_v - absvor * u
        dt_vors, dt_divs = self.vor_div_from_uv_grid(dt_u, dt_v)
        phis_plus_ke = self.trans_grid_to_spherical(phig_full + 0.5 * (u ** 2 + v ** 2))
        dt_divs = dt_divs - self.compute_laplacian(phis_plus_ke)
        # intermediates for stage-by-stage parity tests of the device pipeline
        self.dbg = dict(g_dtu=self.divide_by_cos(dt_u), g_dtv=self.divide_by_cos(dt_v), g_dtT=dt_t,
                        g_E=phig_full + 0.5 * (u ** 2 + v ** 2), g_dtlp=dt_ps / self.psg[cur], wg_full=wg_full,
                        s_dtvor=dt_vors, s_dtdiv=dt_divs, s_dtT=dt_ts, s_dtlp=dt_ln_ps)
        # --- implicit, damping, leapfrog :906-931
        if c.use_implicit:
            dt_divs, dt_ts, dt_ln_ps = self.implicit_correction(
                dt_divs, dt_ts, dt_ln_ps, self.divs, self.ts, self.ln_ps, delta_t, prev, cur)
        dt_vors = self.compute_spectral_damping(self.vors[prev], dt_vors, delta_t, "vor")
        dt_divs = self.compute_spectral_damping(self.divs[prev], dt_divs, delta_t, "div")
        dt_ts = self.compute_spectral_damping(self.ts[prev], dt_ts, delta_t, "t")
        rc, raw = c.robert_coeff, c.raw_filter_coeff
        part = {}
        for name, dta in (("ln_ps", dt_ln_ps), ("vors", dt_vors), ("divs", dt_divs), ("ts", dt_ts)):
            a = getattr(self, name)
            part[name] = a[prev] - 2.0 * a[cur]                    # leapfrog.F90:73
            newfut = a[prev] + delta_t * dta
            a[cur] = a[cur] + rc * part[name] * raw
            if prev == cur:
                pass                                                 # same storage: :75-77
            a[fut] = newfut
        # --- back to grid :933-938
        self.divg = self.trans_spherical_to_grid(self.divs[fut])
        self.vorg = self.trans_spherical_to_grid(self.vors[fut])
        self.ug[fut], self.vg[fut] = self.uv_grid_from_vor_div(self.vors[fut], self.divs[fut])
        self.tg[fut] = self.trans_spherical_to_grid(self.ts[fut])
        self.psg[fut] = np.exp(self.trans_spherical_to_grid(self.ln_ps[fut]))
        tmin, tmax = self.tg[fut].min(), self.tg[fut].max()
        if tmin < c.valid_range_t[0] or tmax > c.valid_range_t[1]:
            raise FloatingPointError("temperatures out of valid range")      # :940-972
        if with_tracer:                                                      # update_tracers :1007
            tr_cur_new, tr_future, part_tr = self.update_grid_tracer(self.tr[prev], self.tr[cur], dt_tr, u, v, wg, p_half, delta_t,
                                                                          advect_vert=c.sphum_advect_vert)
            if prev == cur:
                self.tr[cur] = tr_cur_new
            else:
                self.tr[cur] = tr_cur_new
            self.tr[fut] = tr_future
            for x in self.xtr:
                if x["kind"] == "spectral":
                    self._update_spectral_tracer(x, u, v, wg, p_half, delta_t, prev, cur, fut)
                else:
                    g = x["g"]
                    g[cur], g[fut], x["part"] = self.update_grid_tracer(g[prev], g[cur], x["dt"], u, v, wg, p_half, delta_t, rc=x["rc"],
                                                                        advect_vert=x["vert"] or "finite_volume_parabolic")
        # --- compute_corrections :1213-1302
        if c.do_mass_correction:
            mean_ps_tmp = self.area_weighted_global_mean(self.psg[fut])
            factor = mean_ps_prev / mean_ps_tmp
            self.psg[fut] = factor * self.psg[fut]
            self.ln_ps[fut][0, 0] += math.sqrt(2.0) * math.log(factor)
        if c.do_energy_correction:
            mean_energy_tmp = self.mass_weighted_global_integral(
                0.5 * (self.ug[fut] ** 2 + self.vg[fut] ** 2) + CP_AIR * self.tg[fut], self.psg[fut])
            tcorr = GRAV * (mean_energy_prev - mean_energy_tmp) / (CP_AIR * mean_ps_prev)
            self.tg[fut] = self.tg[fut] + tcorr
            self.ts[fut][:, 0, 0] += math.sqrt(2.0) * tcorr
        if with_tracer and c.do_water_correction:                              # :1245-1283
            q = self.tr[fut]
            mean_water_tmp = self.mass_weighted_global_integral(q, self.psg[fut])
            mask = (p_full >= c.water_correction_limit)
            corr = self.mass_weighted_global_integral(q * mask, self.psg[fut])
            notc = self.mass_weighted_global_integral(q * (~mask), self.psg[fut])
            if mean_water_tmp > 0.0:
                f = mean_water_prev / mean_water_tmp
                f = f * (1. + notc / corr) - notc / corr
                self.tr[fut] = np.where(mask, f * q, q)
        if with_tracer:
            self.tr_atm[fut] = self.tr[fut].copy()
            for x in self.xtr:
                x["atm"][fut] = x["g"][fut].copy()
        self.previous, self.current = cur, fut
        # --- complete_robert_filter :1456-1490 (leapfrog_2level_B with swapped pointers)
        for name in ("ln_ps", "vors", "divs", "ts"):
            a = getattr(self, name)
            a[cur] = a[cur] + rc * a[fut] * raw
            a[fut] = a[fut] + rc * (part[name] + a[fut]) * (raw - 1.0)
        if with_tracer:                                                      # leapfrog_2level_B on the grid tracer :1484
            self.tr[cur] = self.tr[cur] + rc * self.tr[fut] * raw
            self.tr[fut] = self.tr[fut] + rc * (part_tr + self.tr[fut]) * (raw - 1.0)
            for x in self.xtr:                                               # :1479-1486, each with its own robert_coeff
                a = x["s"] if x["kind"] == "spectral" else x["g"]
                a[cur] = a[cur] + x["rc"] * a[fut] * raw
                a[fut] = a[fut] + x["rc"] * (x["part"] + a[fut]) * (raw - 1.0)
        self.wg_full = wg_full
        self.p_full[cur], self.p_half[cur] = p_full, p_half       # intent(out) of spectral_dynamics
        self._pressures_and_heights(fut)                          # atmosphere.F90:331-338
        self.step_count += 1

    # ----------------------------------------------------------------------------------------
    # grid-tracer transport: model/fv_advection.F90 (van Leer on the sphere) and
    # atmos_shared/vert_advection/vert_advection.F90:301-438 (PPM), update_tracers spectral_dynamics.F90:1155-1180
    # ----------------------------------------------------------------------------------------
    def _fv_init(self):
        """fv_advection_init (fv_advection.F90:58-120) with the boundaries of transforms.F90:313-321."""
        if hasattr(self, "_fv"):
            return self._fv
        J, I = self.J, self.I
        yy = np.zeros(J + 1)
        yy[0] = -0.5 * PI
        sum_wts = 0.0
        for j in range(J - 1):
            sum_wts = sum_wts + self.wts_lat[j]
            yy[j + 1] = math.asin(sum_wts - 1.0)
        yy[J] = 0.5 * PI
        y = 0.5 * (yy[1:] + yy[:-1])
        fv = dict(c=np.cos(y), cc=np.cos(yy))
        dy = np.zeros(J + 4)                       # index offset 2: dy[-1..J+2] -> dy[j+1]
        dy[2:J + 2] = yy[1:] - yy[:-1]
        dy[0] = dy[3]; dy[1] = dy[2]; dy[J + 2] = dy[J + 1]; dy[J + 3] = dy[J]
        dyy = np.zeros(J + 1)                      # dyy(1:ny+1) -> dyy[j-1]
        dyy[1:J] = y[1:] - y[:-1]
        dyy[0] = 2 * (y[0] - yy[0]); dyy[J] = 2 * (yy[J] - y[J - 1])
        # dy_plus(0:ny+1), dy_minus(0:ny+1): index j -> [j]
        dyF = lambda j: dy[j + 1]                  # Fortran dy(j)
        fv["dy_plus"] = np.array([dyF(j) / (dyF(j) + dyF(j + 1)) for j in range(0, J + 2)])
        fv["dy_minus"] = np.array([dyF(j) / (dyF(j - 1) + dyF(j)) for j in range(0, J + 2)])
        fv["dy"] = dy * self.cfg.radius                     # Fortran dy(j) = fv['dy'][j+1]
        fv["dyy"] = dyy * self.cfg.radius                   # Fortran dyy(j) = fv['dyy'][j-1]
        fv["dx"] = (1.0 / self.cfg.fourier_inc) * 2.0 * PI * self.cfg.radius / float(I)      # fv_advection.F90:108: the grid spans 360/fourier_inc degrees
        self._fv = fv
        return fv

    @staticmethod
    def _find_cell_x(b):
        I = b.shape[-1]
        ii = np.arange(I)[None, None, :] - np.floor(b).astype(np.int64)      # 1-based: (i-1) - floor(b)
        ii = np.where(ii > I, ii - I, ii)
        ii = np.where(ii < 1, ii + I, ii)
        return ii                                                            # 1-based cell index

    def _semi_x(self, ua, q, dt):
        fv = self._fv_init()
        I = self.I
        b = ua * dt / (fv["dx"] * fv["c"][None, :, None])
        ii = self._find_cell_x(b)
        i_left = ii
        i_right = np.where(ii + 1 > I, 1, ii + 1)
        bb = b - np.floor(b)
        ql = np.take_along_axis(q, i_left - 1, axis=2)
        qr = np.take_along_axis(q, i_right - 1, axis=2)
        return bb * ql + (1.0 - bb) * qr - q

    @staticmethod
    def _slope_x(q):
        grad = q - np.roll(q, 1, axis=2)
        slope = (np.roll(grad, -1, axis=2) + grad) / 2
        qm, qp = np.roll(q, 1, axis=2), np.roll(q, -1, axis=2)
        q_min = np.minimum(np.minimum(qm, q), qp); q_max = np.maximum(np.maximum(qm, q), qp)
        return np.where(slope >= 0, 1.0, -1.0) * np.minimum(np.minimum(np.abs(slope), 2.0 * (q - q_min)), 2.0 * (q_max - q))

    def _vanleer_x(self, dq_dt, uc, q, dt):
        fv = self._fv_init()
        I = self.I
        b = uc * dt / (fv["dx"] * fv["c"][None, :, None])
        bb = b - np.trunc(b)                                   # b - int(b)
        flux = np.zeros_like(q)
        big = np.abs(b).max(axis=2) > 1.0                      # rows with |CFL| > 1: integer_flux_x (:494-527)
        for k, j in zip(*np.nonzero(big)):
            c_, q_ = b[k, j], q[k, j]
            iic = np.trunc(c_).astype(int)
            for i in range(1, I + 1):
                n_ = iic[i - 1]
                if n_ >= 1:
                    if i - n_ >= 1:
                        flux[k, j, i - 1] = np.sum(q_[i - n_ - 1:i - 1])
                    else:
                        flux[k, j, i - 1] = np.sum(q_[0:i - 1]) + np.sum(q_[i - n_ + I - 1:I])
                elif n_ <= -1:
                    if i - 1 - n_ <= I:
                        flux[k, j, i - 1] = -np.sum(q_[i - 1:i - 1 - n_])
                    else:
                        flux[k, j, i - 1] = -np.sum(q_[i - 1:I]) - np.sum(q_[0:i - 1 - n_ - I])
        s = self._slope_x(q)
        ii = self._find_cell_x(b)
        qq = np.take_along_axis(q, ii - 1, axis=2)
        ss = np.take_along_axis(s, ii - 1, axis=2)
        flux = flux + bb * (qq + 0.5 * ss * (np.where(bb >= 0, 1.0, -1.0) - bb))
        return dq_dt - (np.roll(flux, -1, axis=2) - flux) / dt

    def _slope_sphere(self, q):
        """q rows -2..J+1 (J+4 rows) -> slope rows -1..J (J+2 rows); dy_plus/minus index j = row+... (:546-565)"""
        fv = self._fv_init()
        J = self.J
        mid, up, dn = q[:, 1:J + 3], q[:, 2:J + 4], q[:, 0:J + 2]       # rows j=0..J+1 (Fortran), j+1, j-1
        slope = (up - mid) * fv["dy_plus"][None, :, None] + (mid - dn) * fv["dy_minus"][None, :, None]
        q_min = np.minimum(np.minimum(dn, mid), up); q_max = np.maximum(np.maximum(dn, mid), up)
        return np.where(slope >= 0, 1.0, -1.0) * np.minimum(np.minimum(np.abs(slope), 2.0 * (mid - q_min)), 2.0 * (q_max - mid))

    def _vanleer_sphere(self, dq_dt, vc, q, dt):
        """vc rows 1..J+1 (J+1 rows); q rows -1..J+2 (J+4 rows) (:268-304)"""
        fv = self._fv_init()
        J = self.J
        s = self._slope_sphere(q)                                       # rows 0..J+1
        dyF = fv["dy"]                                                  # Fortran dy(j) = dyF[j+1]
        flux = np.zeros((q.shape[0], J + 1, q.shape[2]))
        for jj in range(1, J + 2):                                      # Fortran j = 1..J+1
            v = vc[:, jj - 1]
            qm, sm = q[:, jj - 1 + 1], s[:, jj - 1]                     # q(j-1): row index j-1 -> array (j-1)+2-... see below
            # arrays: q index a = j + 1 (rows -1..J+2 -> 0..J+3), s index a = j (rows 0..J+1 -> 0..J+1)
            qm, sm = q[:, (jj - 1) + 1], s[:, (jj - 1)]
            q0, s0 = q[:, jj + 1], s[:, jj]
            dtdy_m = dt / dyF[(jj - 1) + 1]; dtdy_0 = dt / dyF[jj + 1]
            flux[:, jj - 1] = np.where(v >= 0.0, v * fv["cc"][jj - 1] * (qm + 0.5 * sm * (1.0 - dtdy_m * v)),
                                       v * fv["cc"][jj - 1] * (q0 - 0.5 * s0 * (1.0 + dtdy_0 * v)))
        flux[:, 0] = 0.0; flux[:, J] = 0.0
        dyc = 1.0 / (dyF[2:J + 2] * fv["c"])
        return dq_dt - dyc[None, :, None] * (flux[:, 1:] - flux[:, :-1])

    def a_grid_horiz_advection(self, ua, va, q, dt, dq_dt):
        """fv_advection.F90:126-207 + advection_sphere_3d :238-264 (single PE: polar mirror rows only)."""
        fv = self._fv_init()
        J, I = self.J, self.I
        sh = I // 2
        def with_halo(a, sign):                                          # rows -1..J+2
            out = np.zeros((a.shape[0], J + 4, I))
            out[:, 2:J + 2] = a
            out[:, 1] = sign * np.roll(a[:, 0], -sh, axis=1)             # row 0  <- ii(i)=i+nx/2 of row 1
            out[:, 0] = sign * np.roll(a[:, 1], -sh, axis=1)             # row -1 <- row 2
            out[:, J + 2] = sign * np.roll(a[:, J - 1], -sh, axis=1)
            out[:, J + 3] = sign * np.roll(a[:, J - 2], -sh, axis=1)
            return out
        vx = with_halo(va, -1.0); vx[:, 0] = 0.0; vx[:, J + 3] = 0.0     # vx(-1), vx(ny+2) stay zero in the reference
        qx = with_halo(q, 1.0)
        uc = 0.5 * (np.roll(ua, 1, axis=2) + ua)
        vc = 0.5 * (vx[:, 1:J + 2] + vx[:, 2:J + 3])                     # Fortran j=1..J+1: vx(j-1)+vx(j)
        cF, ccF, dyF = fv["c"], fv["cc"], fv["dy"]
        div = (vc[:, 1:] * ccF[None, 1:, None] - vc[:, :-1] * ccF[None, :-1, None]) / (cF * dyF[2:J + 2])[None, :, None]
        div = div + (np.roll(uc, -1, axis=2) - uc) / (cF[None, :, None] * fv["dx"])
        dq_dt = dq_dt + q * div
        # advection_sphere
        q1 = q + self._semi_x(ua, q, 0.5 * dt)
        qxm, qxp = qx[:, 1:J + 1], qx[:, 3:J + 3]                         # rows j-1, j+1 for j=1..J
        dyyF = fv["dyy"]                                                  # Fortran dyy(j) = dyyF[j-1]
        semi_y = np.where(va >= 0.0, va * dt * 0.5 * (qxm - q) / dyyF[None, 0:J, None],
                          va * dt * 0.5 * (q - qxp) / dyyF[None, 1:J + 1, None])
        q2 = q + semi_y
        q1h = with_halo(q1, 1.0)
        dq_dt = self._vanleer_x(dq_dt, uc, q2, dt)
        return self._vanleer_sphere(dq_dt, vc, q1h, dt)

    @staticmethod
    def _compute_weights(dz):
        L = dz.shape[0]
        zwt = np.zeros((4, L) + dz.shape[1:])
        for k in range(2, L - 1):                                         # Fortran k = 3..n-1
            d1 = 1.0 / (dz[k - 1] + dz[k]); d2 = 1.0 / (dz[k - 2] + dz[k - 1] + dz[k] + dz[k + 1])
            d3 = 1.0 / (2 * dz[k - 1] + dz[k]); d4 = 1.0 / (dz[k - 1] + 2 * dz[k])
            n3 = dz[k - 2] + dz[k - 1]; n4 = dz[k] + dz[k + 1]
            x = n3 * d3 - n4 * d4; y = 2.0 * dz[k - 1] * dz[k]
            zwt[0, k] = dz[k - 1] * d1
            zwt[1, k] = zwt[0, k] + x * y * d1 * d2
            zwt[2, k] = dz[k - 1] * n3 * d3 * d2
            zwt[3, k] = dz[k] * n4 * d4 * d2
        return zwt

    @staticmethod
    def _slope_z(r, dz):
        """slope_z(limit=.true., linear=.false.) vert_advection.F90:505-568"""
        L = r.shape[0]
        grad = np.zeros_like(r)
        grad[1:] = (r[1:] - r[:-1]) / (dz[1:] + dz[:-1])
        slope = np.zeros_like(r)
        slope[1:-1] = (grad[2:] * (2. * dz[:-2] + dz[1:-1]) + grad[1:-1] * (2. * dz[2:] + dz[1:-1])) \
            * dz[1:-1] / (dz[:-2] + dz[1:-1] + dz[2:])
        rmin = np.minimum(np.minimum(r[:-2], r[1:-1]), r[2:]); rmax = np.maximum(np.maximum(r[:-2], r[1:-1]), r[2:])
        slope[1:-1] = np.where(slope[1:-1] >= 0, 1.0, -1.0) * np.minimum(np.minimum(np.abs(slope[1:-1]), 2. * (r[1:-1] - rmin)), 2. * (rmax - r[1:-1]))
        slope[0] = 0.0; slope[-1] = 0.0
        return slope

    @staticmethod
    def vert_advection_fourth_centered(w, dz, r):
        """vert_advection_3d, scheme=FOURTH_CENTERED, no mask (:239-276): fourth order inside, second order at the two interfaces next to the ends."""
        L = r.shape[0]
        flux = np.zeros_like(w)
        flux[0] = w[0] * r[0]
        flux[L] = w[L] * r[L - 1]
        flux[2:L - 1] = w[2:L - 1] * (7. / 12. * (r[2:L - 1] + r[1:L - 2]) - 1. / 12. * (r[3:L] + r[0:L - 3]))
        flux[1] = w[1] * (0.5 * (r[1] + r[0]))
        flux[L - 1] = w[L - 1] * (0.5 * (r[L - 1] + r[L - 2]))
        return -(flux[1:] - flux[:-1] - r * (w[1:] - w[:-1])) / dz

    @staticmethod
    def vert_advection_van_leer(dt, w, dz, r):
        """vert_advection_3d, scheme=VAN_LEER_LINEAR = FINITE_VOLUME_LINEAR (:279-299) with slope_z(limit=.true., linear=.true.) (:505-568)."""
        L = r.shape[0]
        grad = np.zeros_like(r)
        grad[1:] = (r[1:] - r[:-1]) / (dz[1:] + dz[:-1])
        slope = np.zeros_like(r)
        slope[1:-1] = (grad[2:] + grad[1:-1]) * dz[1:-1]
        rmin = np.minimum(np.minimum(r[:-2], r[1:-1]), r[2:]); rmax = np.maximum(np.maximum(r[:-2], r[1:-1]), r[2:])
        slope[1:-1] = np.where(slope[1:-1] >= 0, 1.0, -1.0) * np.minimum(np.minimum(np.abs(slope[1:-1]), 2. * (r[1:-1] - rmin)), 2. * (rmax - r[1:-1]))
        flux = np.zeros_like(w)
        flux[0] = w[0] * r[0]
        flux[L] = w[L] * r[L - 1]
        wk = w[1:L]
        up = r[:-1] + 0.5 * slope[:-1] * (1. - dt * wk / dz[:-1])
        dn = r[1:] - 0.5 * slope[1:] * (1. + dt * wk / dz[1:])
        flux[1:L] = wk * np.where(wk >= 0., up, dn)
        return -(flux[1:] - flux[:-1] - r * (w[1:] - w[:-1])) / dz

    def vert_advection(self, scheme, dt, w, dz, r):
        """vert_advection with one of the four schemes spectral_dynamics_mod offers (spectral_dynamics.F90:280-301, 395-408), ADVECTIVE_FORM."""
        if scheme == "second_centered":
            return self.vert_advection_second_centered(w, dz, r)
        if scheme == "fourth_centered":
            return self.vert_advection_fourth_centered(w, dz, r)
        if scheme == "van_leer_linear":
            return self.vert_advection_van_leer(dt, w, dz, r)
        if scheme == "finite_volume_parabolic":
            return self.vert_advection_ppm(dt, w, dz, r)
        raise ValueError(scheme)

    def vert_advection_ppm(self, dt, w, dz, r):
        """vert_advection_3d, scheme=FINITE_VOLUME_PARABOLIC, form=ADVECTIVE_FORM (:301-438, :467-470)."""
        L = r.shape[0]
        zwt = self._compute_weights(dz)
        slp = self._slope_z(r, dz)
        r_left = np.zeros_like(r); r_right = np.zeros_like(r)
        for k in range(2, L - 1):
            r_left[k] = r[k - 1] + zwt[1, k] * (r[k] - r[k - 1]) - zwt[2, k] * slp[k] + zwt[3, k] * slp[k - 1]
            r_right[k - 1] = r_left[k]
        r_left[1] = r[1] - 0.5 * slp[1]; r_right[L - 2] = r[L - 2] + 0.5 * slp[L - 2]
        r_left[0] = r[0] - 0.5 * slp[0]; r_right[0] = r[0] + 0.5 * slp[0]
        r_left[L - 1] = r[L - 1] - 0.5 * slp[L - 1]; r_right[L - 1] = r[L - 1] + 0.5 * slp[L - 1]
        for k in range(L):                                                 # Colella-Woodward limiter (:354-370)
            t1 = (r_right[k] - r[k]) * (r[k] - r_left[k]) <= 0.0
            r_left[k] = np.where(t1, r[k], r_left[k]); r_right[k] = np.where(t1, r[k], r_right[k])
            if k == 0 or k == L - 1:
                continue
            rm = r_right[k] - r_left[k]
            a = rm * (r[k] - 0.5 * (r_right[k] + r_left[k])); b = rm * rm / 6.
            new_left = np.where(a > b, 3.0 * r[k] - 2.0 * r_right[k], r_left[k])
            rm2 = r_right[k] - new_left
            new_right = np.where(a < -b, 3.0 * r[k] - 2.0 * new_left, r_right[k])
            r_left[k], r_right[k] = new_left, new_right
        flux = np.zeros_like(w)
        flux[0] = w[0] * r[0]; flux[L] = w[L] * r[L - 1]
        tt = 2. / 3.
        for k in range(1, L):
            wk = w[k]
            cn_p = dt * wk / dz[k - 1]; cn_m = -dt * wk / dz[k]
            if np.any((wk >= 0) & (cn_p > 1.0)) or np.any((wk < 0) & (cn_m > 1.0)):
                raise NotImplementedError("vertical Courant number > 1 (vert_advection.F90:385-396) not needed by the oracle cases")
            kk = k - 1
            rm = r_right[kk] - r_left[kk]
            r6 = 6.0 * (r[kk] - 0.5 * (r_right[kk] + r_left[kk]))
            if kk == 0:
                r6 = 0.0 * r6
            rst_p = r_right[kk] - 0.5 * cn_p * (rm - (1.0 - tt * cn_p) * r6)
            kk = k
            rm = r_right[kk] - r_left[kk]
            r6 = 6.0 * (r[kk] - 0.5 * (r_right[kk] + r_left[kk]))
            if kk == L - 1:
                r6 = 0.0 * r6
            rst_m = r_left[kk] + 0.5 * cn_m * (rm + (1.0 - tt * cn_m) * r6)
            flux[k] = wk * np.where(wk >= 0., rst_p, rst_m)
        return -(flux[1:] - flux[:-1] - r * (w[1:] - w[:-1])) / dz

    def update_grid_tracer(self, tr_prev, tr_cur, dt_tr, u, v, wg, p_half, delta_t, rc=None, advect_vert="finite_volume_parabolic"):
        """update_tracers, 'grid' branch (spectral_dynamics.F90:1155-1180); returns (tr_cur filtered part A, tr_future)."""
        tr_future = tr_prev + delta_t * dt_tr
        dq = self.a_grid_horiz_advection(u, v, tr_future, delta_t, np.zeros_like(tr_future))
        tr_future = tr_future + delta_t * dq
        dp = p_half[1:] - p_half[:-1]
        tr_future = tr_future + delta_t * self.vert_advection(advect_vert, delta_t, wg, dp, tr_future)
        rc, raw = (self.cfg.robert_coeff if rc is None else rc), self.cfg.raw_filter_coeff
        part = tr_prev - 2.0 * tr_cur
        tr_cur_new = tr_cur + rc * part * raw
        return tr_cur_new, tr_future, part

    def _update_spectral_tracer(self, x, u, v, wg, p_half, delta_t, prev, cur, fut):
        """update_tracers, 'spectral' branch (spectral_dynamics.F90:1133-1154): spectral horizontal advection of the current coefficients,
        second-centred vertical advection of the current grid values, water_borrowing if asked for, damping like temperature's
        (compute_spectral_damping without a kind, spectral_damping.F90:172-200), leapfrog_2level_A, synthesis of the new level."""
        dt = self.horizontal_advection(x["s"][cur], u, v, x["dt"])
        dp = p_half[1:] - p_half[:-1]
        scheme = x["vert"] or "second_centered"                              # centred schemes on the current level, finite-volume ones on the previous (:1135-1141)
        dt = dt + self.vert_advection(scheme, delta_t, wg, dp, x["g"][cur if scheme.endswith("centered") else prev])
        if x["holes"]:
            dt = self.water_borrowing(dt, x["g"][prev], cur, p_half, delta_t)
        dts = self.compute_spectral_damping(x["s"][prev], self.trans_grid_to_spherical(dt), delta_t, "t")
        a, raw = x["s"], self.cfg.raw_filter_coeff
        x["part"] = a[prev] - 2.0 * a[cur]                                  # leapfrog.F90:73
        newfut = a[prev] + delta_t * dts
        a[cur] = a[cur] + x["rc"] * x["part"] * raw
        a[fut] = newfut
        x["g"][fut] = self.trans_spherical_to_grid(a[fut])

    @staticmethod
    def water_borrowing(dt_q, q, current, p_half, delta_t):
        """atmos_spectral/model/water_borrowing.F90:37-112: a negative cell whose four/six neighbours (east, west, above, below) hold enough
        water to cover it is brought to zero by the tendency and the neighbours are scaled down by the same mass; cells are visited west
        to east when `current` (1-based in the reference) is even, east to west when odd -- only the order of the additions depends on
        it, q itself is not changed inside the loop."""
        L, J, I = q.shape
        dp = p_half[1:] - p_half[:-1]
        out = dt_q.copy()
        cur1 = current + 1                                                   # the reference's time-level index
        ks, js, is_ = np.nonzero(q < 0.0)
        order = np.lexsort((is_ if cur1 % 2 == 0 else -is_, ks, js))        # j outermost, then k, then i in sweep direction
        for n in order:
            k, j, i = int(ks[n]), int(js[n]), int(is_[n])
            iw, ie = (i - 1) % I, (i + 1) % I
            nb = q[k, j, iw] * dp[k, j, iw]
            nb = nb + q[k, j, ie] * dp[k, j, ie]
            if k != 0:
                nb = nb + q[k - 1, j, i] * dp[k - 1, j, i]
            if k != L - 1:
                nb = nb + q[k + 1, j, i] * dp[k + 1, j, i]
            total = nb + q[k, j, i] * dp[k, j, i]
            if total > 0.0:
                ratio = total / nb
                out[k, j, i] -= q[k, j, i] / delta_t
                out[k, j, iw] += (ratio - 1) * q[k, j, iw] / delta_t
                out[k, j, ie] += (ratio - 1) * q[k, j, ie] / delta_t
                if k != 0:
                    out[k - 1, j, i] += (ratio - 1) * q[k - 1, j, i] / delta_t
                if k != L - 1:
                    out[k + 1, j, i] += (ratio - 1) * q[k + 1, j, i] / delta_t
        return out

    # convenience
    def state(self):
        cur = self.current
        return dict(ug=self.ug[cur], vg=self.vg[cur], tg=self.tg[cur], psg=self.psg[cur],
                    vors=self.vors[cur], divs=self.divs[cur], ts=self.ts[cur], ln_ps=self.ln_ps[cur])
