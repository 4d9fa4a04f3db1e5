// get_topography for a height field handed over (topography_option = 'input', init/spectral_init_cond.F90:186-245): the surface geopotential
// g * height, spectrally truncated (ocean_topog_smoothing = 0, :229-235) or regularised over the ocean -- topog_regularization_mod
// (init/topog_regularization.F90: compute_lambda :75-150, regularize :153-290, topog_regularization_init :292-365; Lindberg & Broccoli 1996).
// Initialisation-time host arithmetic on the (m, n) coefficients; every transform and global mean is the device's, through the library's own
// entry points (isca_trans_grid_to_spherical / isca_trans_spherical_to_grid / isca_area_weighted_global_mean).
#include <algorithm>
#include <cmath>
#include <complex>
#include <stdexcept>
#include <string>
#include <vector>
#include "core.h"

void isca_internal_set_error(const std::string &m);

namespace {
using cplx = std::complex<double>;
[[noreturn]] void fail(const std::string &m) { throw std::runtime_error(m); }
constexpr int ITMAX = 1000, ITMAX_LAMBDA = 20;          // topog_regularization.F90:57, :85
constexpr double TOLERANCE = 1.0e-5, TOL_LAMBDA = 0.001;

struct Setup {        // topog_regularization_init (:292-365)
  isca_dyn_t *h;
  int J, I, N1, M1;
  std::vector<char> ocean;            // [J][I]
  std::vector<double> D, LL, keep, sfac;   // [N1][M1]: mean over the ocean points of w_j P_mn(j)^2; (n + m)(n + m + 1); n <= nmax; sin(facm)/facm by m
  Setup(isca_dyn_t *h_, const double *ocean_mask) : h(h_), J(h_->g.J), I(h_->g.I), N1(h_->g.N1), M1(h_->g.M1) {
    if (h->cfg.world_size != 1) fail("regularize: subroutine regularize is not yet coded for 2-d decomposition. It was assumed that it will never be needed.");
    ocean.resize((size_t)J * I);
    for (size_t i = 0; i < ocean.size(); ++i) ocean[i] = ocean_mask[i] != 0.0;
    const int Jh = J / 2, nf = h->cfg.num_fourier;
    std::vector<double> leg((size_t)Jh * N1 * M1), wts(J);
    if (isca_dyn_get_table(h, "legendre", leg.data(), leg.size()) || isca_dyn_get_table(h, "wts_lat", wts.data(), wts.size())) fail(isca_last_error());
    std::vector<double> wh(Jh, 0.0);       // rows j and lat_max + 1 - j share a table entry (P(-x)^2 = P(x)^2)
    for (int j = 0; j < J; ++j) {
      int cnt = 0;
      for (int i = 0; i < I; ++i) cnt += ocean[(size_t)j * I + i];
      wh[j < Jh ? j : J - 1 - j] += wts[j] * cnt;
    }
    const size_t ns = (size_t)N1 * M1;
    D.assign(ns, 0.0); LL.resize(ns); keep.resize(ns); sfac.resize(ns);
    for (int jh = 0; jh < Jh; ++jh)
      for (size_t q = 0; q < ns; ++q) { const double p = leg[(size_t)jh * ns + q]; D[q] += wh[jh] * p * p; }
    const int nmax = std::min(nf, (int)h->cfg.num_spherical);
    for (int n = 0; n < N1; ++n)
      for (int m = 0; m < M1; ++m) {
        const size_t q = (size_t)n * M1 + m;
        D[q] /= I;
        LL[q] = (double)((n + m) * (n + m + 1));
        keep[q] = n <= nmax ? 1.0 : 0.0;
        const double facm = M_PI * m / (2.0 * nf);
        sfac[q] = m == 0 ? 1.0 : std::sin(facm) / facm;
      }
  }
  std::vector<cplx> g2s(const std::vector<double> &g) const {
    std::vector<cplx> s((size_t)N1 * M1);
    if (isca_trans_grid_to_spherical(h, g.data(), (double *)s.data(), 1, 1)) fail(isca_last_error());
    return s;
  }
  std::vector<double> s2g(const std::vector<cplx> &s) const {
    std::vector<double> g((size_t)J * I);
    if (isca_trans_spherical_to_grid(h, (const double *)s.data(), g.data(), 1)) fail(isca_last_error());
    return g;
  }
  double ocean_mean(const std::vector<double> &f) const {     // area_weighted_global_mean of a field that is zero over land
    double v;
    if (isca_area_weighted_global_mean(h, f.data(), &v)) fail(isca_last_error());
    return v;
  }
};

// regularize (:153-290)
double regularize(const Setup &S, double lam, const std::vector<double> &u, std::vector<double> &smoothed) {
  const size_t ns = S.LL.size(), ng = u.size();
  std::vector<double> H(ns);
  for (size_t q = 0; q < ns; ++q) H[q] = S.keep[q] / (1.0 + lam * S.D[q] * S.LL[q] * S.LL[q]);
  const std::vector<cplx> b = S.g2s(u);
  std::vector<cplx> a(ns), dela(ns);
  for (size_t q = 0; q < ns; ++q) { a[q] = S.keep[q] * b[q] / (1.0 + lam * S.LL[q] * S.LL[q]); dela[q] = S.LL[q] * a[q]; }      // (equation 6.3)
  std::vector<double> rough = S.s2g(dela), cost_field(ng);
  double converg = 1.0, cost = 0.0;
  int it = 1;
  for (; it <= ITMAX; ++it) {
    if (std::fabs(converg) < TOLERANCE) break;
    for (size_t i = 0; i < ng; ++i) if (!S.ocean[i]) rough[i] = 0.0;              // rough is zeroed out over land
    std::vector<cplx> dr2 = S.g2s(rough);
    for (size_t q = 0; q < ns; ++q) {
      const cplx d = S.LL[q] * dr2[q] * S.keep[q];
      a[q] = (a[q] + H[q] * (b[q] - a[q]) - lam * H[q] * d) * S.sfac[q];
      dela[q] = S.LL[q] * a[q] * S.keep[q];
    }
    smoothed = S.s2g(a);
    rough = S.s2g(dela);
    for (size_t i = 0; i < ng; ++i) { const double e = u[i] - smoothed[i]; cost_field[i] = S.ocean[i] ? e * e + lam * rough[i] * rough[i] : 0.0; }      // (equation 6.4)
    const double oldcost = cost;
    cost = S.ocean_mean(cost_field);
    if (it > 1) converg = (oldcost - cost) / oldcost;
  }
  if (it > ITMAX) fail("regularize: Failure to converge");
  std::vector<cplx> delb(ns);
  for (size_t q = 0; q < ns; ++q) delb[q] = S.LL[q] * b[q] * S.keep[q];
  std::vector<double> r = S.s2g(delb);
  for (size_t i = 0; i < ng; ++i) cost_field[i] = S.ocean[i] ? r[i] * r[i] : 0.0;
  const double lamcosti = S.ocean_mean(cost_field);
  r = S.s2g(dela);
  for (size_t i = 0; i < ng; ++i) cost_field[i] = S.ocean[i] ? r[i] * r[i] : 0.0;
  const double lamcost = S.ocean_mean(cost_field);
  return 1.0 - lamcost / lamcosti;
}

// compute_lambda (:75-150): secant iteration from 1e-7, 2e-7
void compute_lambda(const Setup &S, double want, const std::vector<double> &u, double &lambda, double &fraction) {
  std::vector<double> tmp;
  double l1 = 1.0e-7, l2 = 2.0e-7;
  double f1 = regularize(S, l1, u, tmp);
  if (std::fabs(want - f1) < TOL_LAMBDA) { lambda = l1; fraction = f1; return; }
  double f2 = regularize(S, l2, u, tmp);
  if (std::fabs(want - f2) < TOL_LAMBDA) { lambda = l2; fraction = f2; return; }
  if (f1 > want || f2 > want) fail("compute_lambda: Iterative scheme for computing lambda may not work unless initial values of lambda_1 and lambda_2 are reduced.");
  l1 = ((f2 - want) * l1 + (want - f1) * l2) / (f2 - f1);
  if (l1 < 0.0) fail("compute_lambda: Iterative scheme for finding lambda will not work unless initial values of lambda_1 and lambda_2 are reduced.");
  f1 = regularize(S, l1, u, tmp);
  for (int it = 1; it <= ITMAX_LAMBDA; ++it) {
    if (std::fabs(want - f1) < TOL_LAMBDA) { lambda = l1; fraction = f1; return; }
    l2 = ((f2 - want) * l1 + (want - f1) * l2) / (f2 - f1);
    if (l2 < 0.0) fail("compute_lambda: Iterative scheme for finding lambda failed. lambda went negative on iteration number" + std::to_string(it));
    f2 = regularize(S, l2, u, tmp);
    if (std::fabs(want - f2) < TOL_LAMBDA) { lambda = l2; fraction = f2; return; }
    l1 = ((f2 - want) * l1 + (want - f1) * l2) / (f2 - f1);
    f1 = regularize(S, l1, u, tmp);
  }
  fail("compute_lambda: Cannot converge on a value of lambda. Perhaps more interations are needed.");
}
}  // namespace

#define TP_BEGIN try {
#define TP_END } catch (const std::exception &e) { isca_internal_set_error(e.what()); return 1; } return 0;

extern "C" int isca_topog_regularize(isca_dyn_t *h, double lambda, const double *ocean_mask, const double *field, double *smoothed, double *fraction_smoothed) {
  TP_BEGIN
  if (!h || !ocean_mask || !field || !smoothed || !fraction_smoothed) fail("null argument");
  const Setup S(h, ocean_mask);
  const std::vector<double> u(field, field + (size_t)S.J * S.I);
  std::vector<double> out;
  *fraction_smoothed = regularize(S, lambda, u, out);
  std::copy(out.begin(), out.end(), smoothed);
  TP_END
}
extern "C" int isca_topog_compute_lambda(isca_dyn_t *h, double ocean_topog_smoothing, const double *ocean_mask, const double *field, double *lambda, double *fraction_smoothed) {
  TP_BEGIN
  if (!h || !ocean_mask || !field || !lambda || !fraction_smoothed) fail("null argument");
  const Setup S(h, ocean_mask);
  compute_lambda(S, ocean_topog_smoothing, std::vector<double>(field, field + (size_t)S.J * S.I), *lambda, *fraction_smoothed);
  TP_END
}
// get_topography, topography_option = 'input' (:186-245): height in m, land mask (> 0: land; may be NULL when ocean_topog_smoothing = 0), both (lon, lat) global
extern "C" int isca_dyn_set_topography(isca_dyn_t *h, const double *height, const double *land_mask, double ocean_topog_smoothing, double *lambda, double *fraction_smoothed) {
  TP_BEGIN
  if (!h || !height) fail("null argument");
  if (h->cfg.world_size != 1) fail("get_topography: an 'input' topography is transformed with the one-rank transforms; hand a sharded run the finished field (isca_dyn_set_surf_geopotential)");
  const size_t ng = (size_t)h->g.J * h->g.I;
  std::vector<double> geop(ng);
  for (size_t i = 0; i < ng; ++i) geop[i] = isca::GRAV * height[i];               // surf_geopotential = grav*surf_height (:229)
  if (lambda) *lambda = 0.0;
  if (fraction_smoothed) *fraction_smoothed = 0.0;
  if (ocean_topog_smoothing == 0.0) {          // spectrally truncate the topography (:231-235)
    if (isca_trans_filter(h, geop.data(), nullptr, 1)) fail(isca_last_error());
  } else {
    if (!land_mask) fail("get_topography: ocean_topog_smoothing /= 0 needs the land mask of the topography file (land_field_name); ocean_topog_smoothing = 0 only truncates");
    std::vector<double> ocean(ng);
    for (size_t i = 0; i < ng; ++i) ocean[i] = land_mask[i] > 0.0 ? 0.0 : 1.0;   // where(land_ones > 0.) ocean_mask = .false. (:223-227)
    const Setup S(h, ocean.data());
    double lam, frac;
    compute_lambda(S, ocean_topog_smoothing, geop, lam, frac);
    std::vector<double> sm;
    frac = regularize(S, lam, geop, sm);
    geop.swap(sm);
    if (lambda) *lambda = lam;
    if (fraction_smoothed) *fraction_smoothed = frac;
  }
  if (isca_dyn_set_surf_geopotential(h, geop.data(), ng)) fail(isca_last_error());
  TP_END
}
