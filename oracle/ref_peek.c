/* oracle/ref_peek.c -- TEST INFRASTRUCTURE ONLY (linked into oracle/_ref/ref_moist_harness.x, never into the product).
 *
 * idealized_moist_phys_mod keeps the mixed layer's surface temperature t_surf as module-private allocatable data with no getter
 * (src/atmos_spectral/driver/solo/idealized_moist_phys.F90:176, :477), and the image cannot write the restart file that would
 * carry it (fms_io needs netCDF).  To hand a DEVELOPED state of the reference's moist model over to the GPU build, the harness
 * READS that array through the object file's symbol: flang emits a module variable as _QM<module>E<name>, and an allocatable's
 * descriptor starts with the base address (CFI_cdesc_t layout: base_addr first).  Nothing is written through the pointer, no
 * reference source is touched, nothing stands in for anything. */
extern void *_QMidealized_moist_phys_modEt_surf;     /* first word of the descriptor = base address of t_surf(is:ie, js:je) */
void *ref_peek_t_surf(void) { return _QMidealized_moist_phys_modEt_surf; }
