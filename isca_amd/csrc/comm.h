// RCCL (the ROCm NCCL) bound at run time with dlopen: the library has no link-time dependency on it, single-GPU use never
// loads it, and inside a PyTorch process the already loaded librccl.so.1 is the one that answers.
// Replaces, for the sharded step, mpp_transmit in transpose_fourier / reverse_transpose_fourier
// (atmos_spectral/tools/transforms.F90:990-1054), mpp_update_domains in fv_advection.F90:161-162,259 and the global
// sums behind area_weighted_global_mean (transforms.F90:1059-1077).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <cstddef>

namespace isca {

struct RcclApi;

class Comm {
 public:
  static constexpr int UNIQUE_ID_BYTES = 128;
  static void unique_id(void *id128);                                    // throws std::runtime_error
  Comm(const void *id128, int rank, int world);                          // collective over all ranks
  ~Comm();
  int rank() const { return rank_; }
  int world() const { return world_; }
  // equal blocks: block p of `send` goes to rank p, block q of `recv` comes from rank q (count doubles each)
  void all_to_all(const double *send, double *recv, size_t count, hipStream_t s);
  // rows for the neighbouring latitude bands: lo <-> rank-1, hi <-> rank+1 (no wrap-around)
  void halo(const double *send_lo, const double *send_hi, double *recv_lo, double *recv_hi, size_t count, hipStream_t s);
  // both of the above in one RCCL group (one fused send/recv kernel instead of two)
  void all_to_all_with_halo(const double *send, double *recv, size_t count, const double *send_lo, const double *send_hi,
                            double *recv_lo, double *recv_hi, size_t halo_count, hipStream_t s);
  void all_reduce_sum(double *buf, size_t count, hipStream_t s);         // in place

 private:
  void *comm_ = nullptr;
  int rank_, world_;
};

}  // namespace isca
