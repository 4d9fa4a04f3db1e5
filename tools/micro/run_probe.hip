// Probe: does the column kernel's access pattern get faster with longer contiguous runs?  Same work split as k_column (8 wavefronts x 5
// levels, NA arrays in, NO arrays out, 256 KB level stride, footprint beyond the MALL), but each lane moves V doubles of V adjacent
// columns (V = 1: 512-byte runs per wavefront access, as today; V = 2: 1 KB; V = 4: 2 KB).  Usage: run_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int V> struct Vec;
template <> struct Vec<1> { typedef double T; };
template <> struct Vec<2> { typedef double2 T; };
template <> struct Vec<4> { typedef double4 T; };
__device__ inline void add(double &a, double b) { a += b; }
__device__ inline void add(double2 &a, double2 b) { a.x += b.x; a.y += b.y; }
__device__ inline void add(double4 &a, double4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

template <int V, int NA, int NO, int CH>
__global__ __launch_bounds__(512) void probe(const double *__restrict__ in, double *__restrict__ out, size_t ls, size_t as, int L) {
  typedef typename Vec<V>::T T;
  const int tid = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t c2 = ((size_t)blockIdx.x * 64 + tid) * V;
  T acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int k = w * CH + i;
    acc[i] = T{};
    if (k < L) {
#pragma unroll
      for (int a = 0; a < NA; ++a) add(acc[i], *(const T *)(in + a * as + k * ls + c2));
    }
  }
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int k = w * CH + i;
    if (k < L) {
#pragma unroll
      for (int o = 0; o < NO; ++o) *(T *)(out + o * as + k * ls + c2) = acc[i];
    }
  }
}

// the same work on a [lat][lev][lon]-like layout: the 40 levels of a block's 64 columns are one contiguous 20 KB piece per array
template <int NA, int NO, int CH>
__global__ __launch_bounds__(512) void probe_blocked(const double *__restrict__ in, double *__restrict__ out, size_t as, int L) {
  const int tid = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t base = (size_t)blockIdx.x * 64 * L + tid;
  double acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int k = w * CH + i;
    acc[i] = 0;
    if (k < L) {
#pragma unroll
      for (int a = 0; a < NA; ++a) acc[i] += in[a * as + base + (size_t)k * 64];
    }
  }
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int k = w * CH + i;
    if (k < L) {
#pragma unroll
      for (int o = 0; o < NO; ++o) out[o * as + base + (size_t)k * 64] = acc[i];
    }
  }
}

// fewer, fatter streams: the NA input fields interleaved component-wise in NG arrays of NA/NG components ([lev][col][comp]), same for the
// outputs -- a wavefront access is then a contiguous run of 64 * comps * 8 bytes
template <int NA, int NO, int NG, int CH>
__global__ __launch_bounds__(512) void probe_interleaved(const double *__restrict__ in, double *__restrict__ out, size_t ls, size_t as, int L) {
  constexpr int CI = NA / NG, CO = NO / NG;
  const int tid = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t c2 = (size_t)blockIdx.x * 64 + tid;
  double acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int k = w * CH + i;
    acc[i] = 0;
    if (k < L) {
#pragma unroll
      for (int gph = 0; gph < NG; ++gph)
#pragma unroll
        for (int cmp = 0; cmp < CI; ++cmp) acc[i] += in[(size_t)gph * as * CI + ((size_t)k * ls + c2) * CI + cmp];
    }
  }
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int k = w * CH + i;
    if (k < L) {
#pragma unroll
      for (int gph = 0; gph < NG; ++gph)
#pragma unroll
        for (int cmp = 0; cmp < CO; ++cmp) out[(size_t)gph * as * CO + ((size_t)k * ls + c2) * CO + cmp] = acc[i];
    }
  }
}

template <int V>
int run(const char *name) {
  const int L = 40, ncol = 256 * 128, NA = 10, NO = 6, NSET = 6;
  const size_t ls = ncol, as = ls * L;
  double *in, *out;
  CK(hipMalloc(&in, as * NA * NSET * sizeof(double)));
  CK(hipMalloc(&out, as * NO * NSET * sizeof(double)));
  CK(hipMemset(in, 0, as * NA * NSET * sizeof(double)));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 60;
  for (int warm = 0; warm < 2; ++warm) {
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) {
      const int set = r % NSET;
      if (V == 5) hipLaunchKernelGGL((probe_interleaved<NA, NO, 2, 5>), dim3(ncol / 64), dim3(512), 0, 0, in + set * as * NA, out + set * as * NO, ls, as, L);
      else if (V == 0) hipLaunchKernelGGL((probe_blocked<NA, NO, 5>), dim3(ncol / 64), dim3(512), 0, 0, in + set * as * NA, out + set * as * NO, as, L);
      else hipLaunchKernelGGL((probe<((V == 0 || V == 5) ? 1 : V), NA, NO, 5>), dim3(ncol / 64 / ((V == 0 || V == 5) ? 1 : V)), dim3(512), 0, 0, in + set * as * NA, out + set * as * NO, ls, as, L);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)(NA + NO) * as * 8;
  printf("%-28s %7.1f us per launch  %6.0f GB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
  hipFree(in); hipFree(out);
  return 0;
}
int main() {
  if (run<1>("1 column / lane (512 B runs)")) return 1;
  if (run<2>("2 columns / lane (1 KB runs)")) return 1;
  if (run<4>("4 columns / lane (2 KB runs)")) return 1;
  if (run<0>("blocked: 20 KB per block/array")) return 1;
  if (run<5>("interleaved: 2x(5 in), 2x(3 out)")) return 1;
  return 0;
}
