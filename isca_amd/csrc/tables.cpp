// Host tables for the MI355X spectral core.  Compiled with -ffp-contract=off so that the Gauss
// nodes / Legendre recursion reproduce the reference's fp64 results bit for bit (the reference's
// flang -O2 build emits no FMA on x86-64).
#include "tables.h"
#include <cmath>
#include <stdexcept>

namespace isca {

// atmos_spectral/tools/gauss_and_legendre.F90:111-183 (Newton iteration on P_n, pole-most node first)
void compute_gaussian(int n_hem, std::vector<double> &sin_hem, std::vector<double> &wts_hem) {
  double converg = 1.0;
  for (int i = 0; i < 15; ++i) converg *= 0.1;   // .1**precision(real*8)
  converg = std::pow(0.1, 15);
  const int n = 2 * n_hem;
  sin_hem.assign(n_hem, 0.0);
  wts_hem.assign(n_hem, 0.0);
  for (int i = 1; i <= n_hem; ++i) {
    double z = std::cos(PI * (i - 0.25) / (n + 0.5));
    double pp = 0.0;
    bool ok = false;
    for (int iter = 0; iter < 10; ++iter) {
      double p1 = 1.0, p2 = 0.0, p3;
      for (int j = 1; j <= n; ++j) {
        p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
      double z1 = z;
      z = z1 - p1 / pp;
      if (std::fabs(z - z1) < converg) { ok = true; break; }
    }
    if (!ok) throw std::runtime_error("compute_gaussian: abscissas failed to converge in itermax iterations");
    sin_hem[i - 1] = z;
    wts_hem[i - 1] = 2.0 / ((1.0 - z * z) * pp * pp);
  }
}

// gauss_and_legendre.F90:47-108; leg[j][n][m]: the polynomials of every zonal wavenumber up to num_fourier * fourier_inc by the
// recursions, every fourier_inc-th one kept
void compute_legendre(int num_fourier, int num_spherical, const std::vector<double> &sin_hem, std::vector<double> &leg, int fourier_inc) {
  const int M1 = num_fourier + 1, N1 = num_spherical + 1, nlat = (int)sin_hem.size();
  const int F1 = num_fourier * fourier_inc + 1;
  std::vector<double> eps((size_t)N1 * F1), poly((size_t)N1 * F1), b(F1, 0.0);
  for (int n = 0; n < N1; ++n)
    for (int m = 0; m < F1; ++m) {
      double m2 = (double)m * m, l2 = (double)(m + n) * (m + n);
      eps[(size_t)n * F1 + m] = std::sqrt((l2 - m2) / (4.0 * l2 - 1.0));
    }
  for (int m = 1; m < F1; ++m) b[m] = std::sqrt(0.5 * (2.0 * (double)m + 1.0) / (double)m);
  leg.assign((size_t)nlat * N1 * M1, 0.0);
  for (int j = 0; j < nlat; ++j) {
    const double s = sin_hem[j];
    const double c = std::sqrt(1 - s * s);
    poly[0] = std::sqrt(0.5);
    for (int m = 1; m < F1; ++m) poly[m] = b[m] * c * poly[m - 1];
    for (int m = 0; m < F1; ++m) poly[F1 + m] = s * poly[m] / eps[F1 + m];
    for (int n = 2; n < N1; ++n)
      for (int m = 0; m < F1; ++m)
        poly[(size_t)n * F1 + m] =
            (s * poly[(size_t)(n - 1) * F1 + m] - eps[(size_t)(n - 1) * F1 + m] * poly[(size_t)(n - 2) * F1 + m]) /
            eps[(size_t)n * F1 + m];
    for (int n = 0; n < N1; ++n)
      for (int m = 0; m < M1; ++m) leg[((size_t)j * N1 + n) * M1 + m] = poly[(size_t)n * F1 + (size_t)m * fourier_inc];
  }
}

// model/matrix_invert.F90:38-130: Gauss-Jordan elimination with pivoting
bool invert_matrix(std::vector<double> &a, int n) {
  std::vector<double> inv((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
  for (int col = 0; col < n; ++col) {
    int piv = col;
    double best = std::fabs(a[(size_t)col * n + col]);
    for (int r = col + 1; r < n; ++r)
      if (std::fabs(a[(size_t)r * n + col]) > best) { best = std::fabs(a[(size_t)r * n + col]); piv = r; }
    if (best == 0.0) return false;
    if (piv != col)
      for (int k = 0; k < n; ++k) {
        std::swap(a[(size_t)piv * n + k], a[(size_t)col * n + k]);
        std::swap(inv[(size_t)piv * n + k], inv[(size_t)col * n + k]);
      }
    const double d = 1.0 / a[(size_t)col * n + col];
    for (int k = 0; k < n; ++k) { a[(size_t)col * n + k] *= d; inv[(size_t)col * n + k] *= d; }
    for (int r = 0; r < n; ++r) {
      if (r == col) continue;
      const double f = a[(size_t)r * n + col];
      if (f == 0.0) continue;
      for (int k = 0; k < n; ++k) {
        a[(size_t)r * n + k] -= f * a[(size_t)col * n + k];
        inv[(size_t)r * n + k] -= f * inv[(size_t)col * n + k];
      }
    }
  }
  a.swap(inv);
  return true;
}


// model/press_and_geopot.F90:152-221 for a single column (simmons_and_burridge; 'mcm': :196-210)
void pressure_variables_1d(const std::vector<double> &pk, const std::vector<double> &bk, double ps,
                           std::vector<double> &ln_p_half, std::vector<double> &ln_p_full, bool mcm) {
  const int L = (int)pk.size() - 1;
  std::vector<double> p_half(L + 1);
  ln_p_half.assign(L + 1, 0.0);
  ln_p_full.assign(L, 0.0);
  for (int k = 0; k <= L; ++k) p_half[k] = pk[k] + bk[k] * ps;
  if (mcm) {
    for (int k = 0; k < L; ++k) ln_p_full[k] = std::log(0.5 * (p_half[k + 1] + p_half[k]));
    const bool top0 = pk[0] == 0.0 && bk[0] == 0.0;
    for (int k = top0 ? 1 : 0; k <= L; ++k) ln_p_half[k] = std::log(p_half[k]);
    return;
  }
  if (pk[0] == 0.0 && bk[0] == 0.0) {
    for (int k = 1; k <= L; ++k) ln_p_half[k] = std::log(p_half[k]);
    for (int k = 1; k < L; ++k) {
      double alpha = 1.0 - p_half[k] * (ln_p_half[k + 1] - ln_p_half[k]) / (p_half[k + 1] - p_half[k]);
      ln_p_full[k] = ln_p_half[k + 1] - alpha;
    }
    ln_p_full[0] = ln_p_half[1] - 1.0;
    ln_p_half[0] = 0.0;
  } else {
    for (int k = 0; k <= L; ++k) ln_p_half[k] = std::log(p_half[k]);
    for (int k = 0; k < L; ++k) {
      double alpha = 1.0 - p_half[k] * (ln_p_half[k + 1] - ln_p_half[k]) / (p_half[k + 1] - p_half[k]);
      ln_p_full[k] = ln_p_half[k + 1] - alpha;
    }
  }
}


void Tables::build(const isca_dyn_config &c) {
  radius = c.radius; omega = c.omega;
  I = c.lon_max; J = c.lat_max; M1 = c.num_fourier + 1; N1 = c.num_spherical + 1; L = c.num_levels;
  // --- Gaussian grid: spherical_fourier.F90:397-431
  compute_gaussian(J / 2, sin_hem, wts_hem);
  sin_lat.resize(J); wts_lat.resize(J); cos_lat.resize(J); cosm_lat.resize(J); deg_lat.resize(J);
  rad_lat.resize(J); coriolis.resize(J);
  for (int j = 0; j < J / 2; ++j) {
    sin_lat[j] = -sin_hem[j];
    sin_lat[J - 1 - j] = -sin_lat[j];
    wts_lat[j] = wts_hem[j];
    wts_lat[J - 1 - j] = wts_hem[j];
  }
  for (int j = 0; j < J; ++j) {
    cos_lat[j] = std::sqrt(1 - sin_lat[j] * sin_lat[j]);
    cosm_lat[j] = 1. / cos_lat[j];
    deg_lat[j] = std::asin(sin_lat[j]) * 180.0 / PI;
    rad_lat[j] = deg_lat[j] * PI / 180.;               // atmosphere.F90:248-251
    coriolis[j] = 2 * omega * sin_lat[j];              // spectral_dynamics.F90:445
  }
  deg_lon.resize(I);
  for (int i = 0; i < I; ++i) deg_lon[i] = i * (360.0 / (double)c.fourier_inc) / (double)I;   // grid_fourier.F90:105-118: a 360/fourier_inc sector
  compute_legendre(c.num_fourier, c.num_spherical, sin_hem, legendre, c.fourier_inc);
  // --- spherical.F90:137-216
  const size_t NM = (size_t)N1 * M1;
  eigen.assign(NM, 0); coef_uvm.assign(NM, 0); coef_uvc.assign(NM, 0); coef_uvp.assign(NM, 0);
  coef_alpm.assign(NM, 0); coef_alpp.assign(NM, 0); coef_dym.assign(NM, 0); coef_dx.assign(NM, 0);
  coef_dyp.assign(NM, 0); tri_mask.assign(NM, 1.0);
  std::vector<double> eps(NM);
  for (int n = 0; n < N1; ++n)
    for (int m = 0; m < M1; ++m) {
      const size_t q = (size_t)n * M1 + m;
      const double fw = m * c.fourier_inc, sw = fw + n;      // the zonal wavenumber of index m is m * fourier_inc (spherical.F90:182-183)
      // the model's truncation: triangle_mask (spherical.F90:190-195), or -- triang_trunc = .false. -- rhomboidal_truncation's
      // `spherical(:,num_spherical,:) = 0` (spherical.F90:622): only the extra row goes
      if (c.triang_trunc ? (sw > c.num_spherical - 1) : (n == c.num_spherical)) tri_mask[q] = 0.0;
      if (c.make_symmetric && m > 0) tri_mask[q] = 0.0;     // make_symmetric (spherical.F90:185): a zonally symmetric model
      eps[q] = std::sqrt((sw * sw - fw * fw) / (4.0 * sw * sw - 1.0));
      eigen[q] = sw * (sw + 1.0) / (radius * radius);
      if (sw > 0) {
        coef_uvm[q] = -radius * eps[q] / sw;
        coef_uvc[q] = -radius * fw / (sw * (sw + 1.0));
      }
      coef_alpm[q] = (sw + 1.0) * eps[q] / radius;
      coef_dym[q] = (sw - 1.0) * eps[q] / radius;
      coef_dx[q] = fw / radius;
    }
  for (int n = 0; n < N1 - 1; ++n)
    for (int m = 0; m < M1; ++m) {
      const size_t q = (size_t)n * M1 + m, qp = (size_t)(n + 1) * M1 + m;
      const double sw = m * c.fourier_inc + n;
      coef_uvp[q] = -radius * eps[qp] / (sw + 1.0);
      coef_alpp[q] = sw * eps[qp] / radius;
      coef_dyp[q] = (sw + 2.0) * eps[qp] / radius;
    }
  // --- spectral_damping_init (spectral_damping.F90:56-168)
  {
    const double cv = c.damping_coeff_vor < 0. ? c.damping_coeff : c.damping_coeff_vor, cd = c.damping_coeff_div < 0. ? c.damping_coeff : c.damping_coeff_div;
    const int ov = c.damping_order_vor < 0 ? c.damping_order : c.damping_order_vor, od = c.damping_order_div < 0 ? c.damping_order : c.damping_order_div;
    damping.assign(NM, 0); damping_vor.assign(NM, 0); damping_div.assign(NM, 0);
    damping_coeffs[0] = c.damping_coeff; damping_coeffs[1] = cv; damping_coeffs[2] = cd;
    damping_exponential = c.damping_option == 1;
    const double eref = eigen[(size_t)(c.num_spherical - 1) * M1 + 0];
    if (c.damping_option == 0) {                 // 'resolution_dependent' (:124-127)
      for (size_t q = 0; q < NM; ++q) {
        damping[q] = c.damping_coeff * std::pow(eigen[q] / eref, c.damping_order);
        damping_vor[q] = cv * std::pow(eigen[q] / eref, ov);
        damping_div[q] = cd * std::pow(eigen[q] / eref, od);
      }
    } else if (c.damping_option == 1) {          // 'exponential_cutoff' (:129-146): one exponent table for the three
      const double ecut = eigen[(size_t)c.cutoff_wn * M1 + 0], scut = std::sqrt(ecut), sref = std::sqrt(eref);
      for (size_t q = 0; q < NM; ++q) {
        const double v = (eigen[q] / ecut > 1.) ? std::pow((std::sqrt(eigen[q]) - scut) / (sref - scut), c.damping_order) : 0.0;
        damping[q] = damping_vor[q] = damping_div[q] = v;
      }
    } else {                                     // 'resolution_independent' (:148-151)
      for (size_t q = 0; q < NM; ++q) {
        damping[q] = c.damping_coeff * std::pow(eigen[q], c.damping_order);
        damping_vor[q] = cv * std::pow(eigen[q], ov);
        damping_div[q] = cd * std::pow(eigen[q], od);
      }
    }
  }
  // --- vertical coordinate: init/vert_coordinate.F90:248-273 ('uneven_sigma', zero_top)
  pk.assign(L + 1, 0.0); bk.assign(L + 1, 0.0);
  if (c.vert_coord_input) {                    // 'input': vert_coordinate_nml's pk, bk as given (vert_coordinate.F90:150-160)
    for (int k = 0; k <= L; ++k) { pk[k] = c.pk_input[k]; bk[k] = c.bk_input[k]; }
  } else {
    const double s2 = 1.0 - c.surf_res;
    for (int k = 1; k <= L; ++k) {
      const double zeta = 1. - ((double)(k - 1) / (double)L);
      const double z = c.surf_res * zeta + s2 * std::pow(zeta, c.exponent);
      bk[k - 1] = std::exp(-z * c.scale_heights);
    }
    bk[L] = 1.0;
    bk[0] = 0.0;
  }
  dpk.resize(L); dbk.resize(L);
  for (int k = 0; k < L; ++k) { dpk[k] = pk[k + 1] - pk[k]; dbk[k] = bk[k + 1] - bk[k]; }
  // --- implicit_init + build_matrix: model/implicit.F90:79-217 (ref T = 300 K: spectral_dynamics.F90:473)
  ref_t = 300.0;
  ref_surf_p = c.reference_sea_level_press;
  const bool mcm = c.vert_difference_option == 1;
  pressure_variables_1d(pk, bk, ref_surf_p, ref_ln_p_half, ref_ln_p_full, mcm);
  std::vector<double> del_ln_p_half(L + 1), del_ln_p_full(L), l1h, l1, l2h, l2;
  for (int k = 1; k <= L; ++k) del_ln_p_half[k] = bk[k] / (pk[k] + bk[k] * ref_surf_p);
  del_ln_p_half[0] = (pk[0] == 0.0) ? 1.0 / ref_surf_p : bk[0] / (pk[0] + bk[0] * ref_surf_p);
  const double epsv = 1.e-5;
  pressure_variables_1d(pk, bk, ref_surf_p * (1.0 - 0.5 * epsv), l1h, l1, mcm);
  pressure_variables_1d(pk, bk, ref_surf_p * (1.0 + 0.5 * epsv), l2h, l2, mcm);
  for (int k = 0; k < L; ++k) del_ln_p_full[k] = (l2[k] - l1[k]) / (epsv * ref_surf_p);
  // linear_tp_tendency_1d (implicit.F90:414-480) and linear_geopotential_1d (:329-359) on unit vectors
  auto tp_tend = [&](const std::vector<double> &div, double &dt_p, std::vector<double> &dt_t) {
    dt_t.assign(L, 0.0);
    std::vector<double> vv(L + 1, 0.0), temp(L + 1, 0.0);
    double dmean_tot = 0.0;
    for (int k = 0; k < L; ++k) {
      const double dp = dpk[k] + dbk[k] * ref_surf_p, dp_inv = 1 / dp;
      const double dlog_1 = ref_ln_p_half[k + 1] - ref_ln_p_full[k];
      const double dlog_3 = ref_ln_p_half[k + 1] - ref_ln_p_half[k];
      const double dmean = div[k] * dp;
      if (mcm) {       // implicit.F90:447-456
        const double p_full_ref = 0.5 * (pk[k + 1] + pk[k]) + 0.5 * (bk[k + 1] + bk[k]) * ref_surf_p;
        dt_t[k] = -(KAPPA * ref_t / p_full_ref) * (dmean_tot + 0.5 * dmean);
      } else
      dt_t[k] = -KAPPA * ref_t * (dmean_tot * dlog_3 + dmean * dlog_1) * dp_inv;
      dmean_tot = dmean_tot + dmean;
      vv[k + 1] = -dmean_tot;
    }
    dt_p = -dmean_tot;
    for (int k = 1; k < L; ++k) { vv[k] += dmean_tot * bk[k]; temp[k] = -vv[k] * (ref_t - ref_t); }
    for (int k = 0; k < L; ++k) {
      const double dp = dpk[k] + dbk[k] * ref_surf_p;
      dt_t[k] += .5 * (1 / dp) * (temp[k + 1] + temp[k]);
    }
  };
  auto lin_geopot = [&](const std::vector<double> &del_t, const std::vector<double> &dlh,
                        const std::vector<double> &dlf, std::vector<double> &g) {
    std::vector<double> gh(L + 1, 0.0);
    g.assign(L, 0.0);
    for (int k = L - 1; k >= 1; --k)
      gh[k] = gh[k + 1] + RDGAS * (del_t[k] * (ref_ln_p_half[k + 1] - ref_ln_p_half[k]) + ref_t * (dlh[k + 1] - dlh[k]));
    for (int k = 0; k < L; ++k)
      g[k] = gh[k + 1] + RDGAS * (del_t[k] * (ref_ln_p_half[k + 1] - ref_ln_p_full[k]) + ref_t * (dlh[k + 1] - dlf[k]));
  };
  tau_mat.assign((size_t)L * L, 0); gamma_mat.assign((size_t)L * L, 0); nu_vec.assign(L, 0);
  std::vector<double> unit(L), zero(L, 0.0), zero1(L + 1, 0.0), col, g;
  for (int k = 0; k < L; ++k) {
    unit.assign(L, 0.0); unit[k] = 1.0;
    double dtp;
    tp_tend(unit, dtp, col);
    nu_vec[k] = -dtp;
    for (int r = 0; r < L; ++r) tau_mat[(size_t)r * L + k] = -col[r];
    lin_geopot(unit, zero1, zero, g);
    for (int r = 0; r < L; ++r) gamma_mat[(size_t)r * L + k] = g[r];
  }
  std::vector<double> h2;
  lin_geopot(zero, del_ln_p_half, del_ln_p_full, h2);
  h_impl.assign(L, 0.0);
  for (int k = 0; k < L; ++k) {   // pres_grad_funct :389-411
    const double dlog_1 = ref_ln_p_half[k + 1] - ref_ln_p_full[k];
    const double dlog_2 = ref_ln_p_full[k] - ref_ln_p_half[k];
    const double h1 = mcm ? RDGAS * ref_t / ref_surf_p        // pres_grad_funct, 'mcm' (:404-408)
                          : RDGAS * ref_t * (bk[k + 1] * dlog_1 + bk[k] * dlog_2) / (dpk[k] + dbk[k] * ref_surf_p);
    h_impl[k] = h1 + h2[k];
  }
  div_mat.assign((size_t)L * L, 0.0);
  for (int k = 0; k < L; ++k)
    for (int kk = 0; kk < L; ++kk) {
      double s = h_impl[k] * nu_vec[kk];
      for (int q = 0; q < L; ++q) s = s + gamma_mat[(size_t)k * L + q] * tau_mat[(size_t)q * L + kk];
      div_mat[(size_t)k * L + kk] = s;
    }
  // --- hs_forcing_init: hs_forcing.F90:391-410
  tka = (c.ka < 0.) ? -1. / (86400 * c.ka) : c.ka;
  tks = (c.ks < 0.) ? -1. / (86400 * c.ks) : c.ks;
  vkf = (c.kf < 0.) ? -1. / (86400 * c.kf) : c.kf;
  trsink_s = (c.trsink < 0.) ? -86400. * c.trsink : c.trsink;
  // --- fv_advection_init (fv_advection.F90:58-120) with the cell boundaries of transforms.F90:313-321
  {
    std::vector<double> yy(J + 1), y(J);
    yy[0] = -.5 * PI;
    double sum_wts = 0.;
    for (int j = 0; j < J - 1; ++j) { sum_wts = sum_wts + wts_lat[j]; yy[j + 1] = std::asin(sum_wts - 1.); }
    yy[J] = .5 * PI;
    lat_boundaries = yy;                                 // get_grid_boundaries (transforms.F90:313-325), longitude_origin = 0
    lon_boundaries.resize(I + 1);
    for (int i = 0; i <= I; ++i) lon_boundaries[i] = ((i + 1) - 1.5) * (2 * PI / I);
    fv_c.resize(J); fv_cc.resize(J + 1); fv_dy.assign(J + 4, 0.0); fv_dyy.assign(J + 1, 0.0);
    fv_dyp.resize(J + 2); fv_dym.resize(J + 2);
    for (int j = 0; j < J; ++j) { y[j] = 0.5 * (yy[j + 1] + yy[j]); fv_c[j] = std::cos(y[j]); }
    for (int j = 0; j <= J; ++j) fv_cc[j] = std::cos(yy[j]);
    auto dyF = [&](int j) -> double & { return fv_dy[j + 1]; };      // Fortran index j = -1..J+2
    for (int j = 1; j <= J; ++j) dyF(j) = yy[j] - yy[j - 1];
    dyF(-1) = dyF(2); dyF(0) = dyF(1); dyF(J + 1) = dyF(J); dyF(J + 2) = dyF(J - 1);
    for (int j = 2; j <= J; ++j) fv_dyy[j - 1] = y[j - 1] - y[j - 2];
    fv_dyy[0] = 2 * (y[0] - yy[0]);
    fv_dyy[J] = 2 * (yy[J] - y[J - 1]);
    for (int j = 0; j <= J + 1; ++j) { fv_dyp[j] = dyF(j) / (dyF(j) + dyF(j + 1)); fv_dym[j] = dyF(j) / (dyF(j - 1) + dyF(j)); }
    for (auto &v : fv_dy) v = v * radius;
    for (auto &v : fv_dyy) v = v * radius;
    fv_dx = ((360.0 / (double)c.fourier_inc) / 360.0) * 2.0 * PI * radius / (double)I;      // fv_advection.F90:108 with degrees_lon = 360/fourier_inc
  }
  // --- FFT twiddles
  tw_re.resize(I); tw_im.resize(I);
  for (int k = 0; k < I; ++k) {
    const long double a = -2.0L * 3.141592653589793238462643383279502884L * (long double)k / (long double)I;
    tw_re[k] = (double)cosl(a);
    tw_im[k] = (double)sinl(a);
  }
}

// implicit.F90:221-237
// compute_spectral_damping (spectral_damping.F90:186-190, :216-220, :263-267): the coefficient applied in a step of length delta_t
void Tables::damping_effective(double delta_t, std::vector<double> &t, std::vector<double> &vor, std::vector<double> &div) const {
  t = damping; vor = damping_vor; div = damping_div;
  if (!damping_exponential) return;
  std::vector<double> *out[3] = {&t, &vor, &div};
  for (int i = 0; i < 3; ++i)
    for (double &v : *out[i]) v = (std::exp(std::log(delta_t * damping_coeffs[i] + 1.0) * v) - 1.0) / delta_t;
}

void Tables::build_wave_matrices(const isca_dyn_config &c, double dt) {
  xi = dt * c.alpha_implicit;
  const int ntw = c.triang_trunc ? c.num_spherical - 1 : c.num_spherical - 1 + c.fourier_inc * c.num_fourier;      // num_total_wavenumbers
  n_wave = ntw + 1;
  wave_matrix.assign((size_t)(ntw + 1) * L * L, 0.0);
  std::vector<double> a((size_t)L * L);
  for (int Lw = 0; Lw <= ntw; ++Lw) {
    const double factor = xi * xi * Lw * (Lw + 1) / (radius * radius);
    for (int k = 0; k < L; ++k)
      for (int kk = 0; kk < L; ++kk) a[(size_t)k * L + kk] = (k == kk ? 1.0 : 0.0) + factor * div_mat[(size_t)k * L + kk];
    if (!invert_matrix(a, L)) throw std::runtime_error("build_wave_matrices: singular matrix");
    for (size_t q = 0; q < (size_t)L * L; ++q) wave_matrix[(size_t)Lw * L * L + q] = a[q];
  }
  wave_dt = dt;
}

}  // namespace isca
