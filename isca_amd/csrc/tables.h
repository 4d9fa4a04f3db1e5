// Host-side init-time tables of the spectral core (fp64), built once per handle.
// Restated from the reference's math (citations per function in tables.cpp).
#pragma once
#include <vector>
#include <cstddef>
#include "../../include/isca_dyn.h"

namespace isca {

constexpr double RADIUS_EARTH = 6376.0e3;   // shared/constants/constants.F90:254 (defaults of constants_nml)
constexpr double OMEGA_EARTH = 7.2921150e-5;
constexpr double GRAV = 9.80;
constexpr double RDGAS = 287.04;
constexpr double RVGAS = 461.50;      // constants.F90: gas constant of water vapour (use_virtual_temperature)
constexpr double KAPPA = 2.0 / 7.0;
constexpr double CP_AIR = RDGAS / KAPPA;
constexpr double PI = 3.14159265358979323846;

struct Tables {
  int I, J, M1, N1, L;
  int n_wave = 0;                                // wave matrices: total wavenumbers 0..num_total_wavenumbers (spectral_dynamics.F90:430-434)
  std::vector<double> sin_hem, wts_hem;          // [J/2], pole-most first
  std::vector<double> lon_boundaries, lat_boundaries;   // [I+1], [J+1] cell edges in radians (transforms.F90:313-325)
  std::vector<double> sin_lat, wts_lat, cos_lat, cosm_lat, deg_lat, rad_lat, deg_lon, coriolis;  // [J] / [I]
  std::vector<double> legendre;                  // [J/2][N1][M1]  (Fortran (m,n,j))
  // spherical.F90 coefficient tables, [N1][M1]
  std::vector<double> eigen, coef_uvm, coef_uvc, coef_uvp, coef_alpm, coef_alpp, coef_dym, coef_dx, coef_dyp, tri_mask;
  // spectral_damping.F90:124-156: tables for T (and tracers), vorticity, divergence, [N1][M1].  With 'exponential_cutoff' they hold the
  // EXPONENT of the filter and the effective coefficient depends on the step's delta_t (damping_effective)
  std::vector<double> damping, damping_vor, damping_div;
  bool damping_exponential = false;
  double damping_coeffs[3] = {0., 0., 0.};        // damping_coeff, _vor, _div as used by the exponential form
  void damping_effective(double delta_t, std::vector<double> &t, std::vector<double> &vor, std::vector<double> &div) const;
  std::vector<double> pk, bk, dpk, dbk;          // [L+1], [L]
  // implicit.F90
  std::vector<double> ref_ln_p_half, ref_ln_p_full, h_impl, div_mat;   // [L+1],[L],[L],[L*L] (row-major k,kk)
  std::vector<double> tau_mat, gamma_mat, nu_vec;                      // [L*L],[L*L],[L]
  double ref_surf_p, ref_t;
  double radius = RADIUS_EARTH, omega = OMEGA_EARTH;   // constants_nml
  std::vector<double> wave_matrix;               // [n_wave][L][L] for wave_dt
  double wave_dt = -1.0, xi = 0.0;
  // hs
  double tka, tks, vkf, trsink_s;
  // fv_advection_init (model/fv_advection.F90:58-120): c[J], cc[J+1], dy[J+4] (Fortran dy(j) = dy[j+1]),
  // dyy[J+1] (dyy(j) = dyy[j-1]), dy_plus/minus[J+2] (index j = 0..J+1), dx
  std::vector<double> fv_c, fv_cc, fv_dy, fv_dyy, fv_dyp, fv_dym;
  double fv_dx;
  // FFT twiddles exp(-2 pi i k / I), k < I/2
  std::vector<double> tw_re, tw_im;

  void build(const isca_dyn_config &c);
  void build_wave_matrices(const isca_dyn_config &c, double dt);
};

// pressure_variables for one column (press_and_geopot.F90:152-221): ln p at half and full levels
void pressure_variables_1d(const std::vector<double> &pk, const std::vector<double> &bk, double ps, std::vector<double> &ln_p_half,
                           std::vector<double> &ln_p_full, bool mcm = false);
void compute_gaussian(int n_hem, std::vector<double> &sin_hem, std::vector<double> &wts_hem);
void compute_legendre(int num_fourier, int num_spherical, const std::vector<double> &sin_hem, std::vector<double> &leg, int fourier_inc = 1);
bool invert_matrix(std::vector<double> &a, int n);   // Gauss-Jordan with pivoting; returns false if singular

}  // namespace isca
