// Moist physics package (idealized_moist_phys, Frierson options) on the device: see moist.hip
#pragma once
#include "core.h"

namespace isca {
struct StepScalars;
struct MoistState;
MoistState *moist_create(const isca_dyn_config &cfg, const Tables &tab);
void moist_destroy(MoistState *m);
size_t moist_work_doubles(const Geom &g);
void launch_moist_pressures(const isca_dyn &h, const StepScalars &sc, hipStream_t s);
void launch_moist_physics(const isca_dyn &h, const StepScalars &sc, hipStream_t s);
void launch_moist_physics_on(const isca_dyn &h, int ncol, double delta_t, double gust, const double *rad_lat, const double *u, const double *v,
                             const double *t, const double *q, const double *ph_p, const double *pf_p, const double *ph_c, const double *pf_c,
                             const double *zh_c, const double *zf_c, double *t_surf, double *dtu, double *dtv, double *dtT, double *dtq,
                             double *precip, double *work, hipStream_t s);
void launch_t_surf_init(const isca_dyn &h, hipStream_t s);
}  // namespace isca
