// Moist physics package (idealized_moist_phys, Frierson options) on the device: see moist.hip
#pragma once
#include "core.h"

namespace isca {
struct StepScalars;
struct MoistState;
MoistState *moist_create(const isca_dyn_config &cfg, const Tables &tab);
void moist_destroy(MoistState *m);
size_t moist_work_doubles(const Geom &g);
// slot_prev / slot_cur: which of the two (p_full, p_half) areas of the work buffer hold the previous / current level's pressures;
// prev_cached: the previous level's are already there (the step before computed them for its current level)
void launch_moist_pressures(const isca_dyn &h, const StepScalars &sc, hipStream_t s, int slot_prev, int slot_cur, bool prev_cached);
// k_moist_convcond: convection + condensation of the step whose previous level is `level` (pressures in slot pslot), into buffer set ccslot;
// launch_moist_physics: k_moist_physics, the rest of the chain, reading that set, and (next) the NEXT step's convection + condensation beside it
// into the other set (moist.hip)
void launch_moist_convcond(const isca_dyn &h, int level, int pslot, double delta_t, int ccslot, hipStream_t s);
void launch_moist_physics(const isca_dyn &h, const StepScalars &sc, hipStream_t s, int slot_cur, int ccslot, bool next);
void launch_moist_physics_on(const isca_dyn &h, int ncol, double delta_t, double gust, const double *rad_lat, const double *u, const double *v,
                             const double *t, const double *q, const double *ph_p, const double *pf_p, const double *ph_c, const double *pf_c,
                             const double *zh_c, const double *zf_c, double *t_surf, double *dtu, double *dtv, double *dtT, double *dtq,
                             double *precip, double *work, double *cc, hipStream_t s);
void launch_t_surf_init(const isca_dyn &h, hipStream_t s);
}  // namespace isca
