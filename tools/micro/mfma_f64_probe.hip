// Probe for pricing a fused FFT <-> Legendre stage (HISTORY.md 4 "Fused FFT + Legendre: priced"): what the FP64 matrix instructions of
// gfx950 cost when the output tile is narrow.  A fused block owns a few level-fields (2-8 columns), so the 16-column tile of
// v_mfma_f64_16x16x4_f64 is mostly empty; v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction) is the alternative.
//   part 1: lane mapping of the 4x4x4 form, found empirically (one-hot A and B lanes -> which D lanes are non-zero);
//   part 2: issue cost in cycles of both forms (independent accumulators back to back; 1, 2 and 4 wavefronts per SIMD);
//   part 3: the table-streaming loop of a fused block: every block streams the WHOLE fragment-ordered Legendre table of T85 (or T170)
//           from L2 while it issues the MFMAs -- 4496 (T85) of them per block, accumulators in registers, B operand from LDS.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_probe.hip -o tools/micro/mfma_f64_probe.x ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---- part 1
__global__ void k_map(unsigned long long *mask) {          // mask[la * 64 + lb] = lanes whose D is non-zero for one-hot A lane la, B lane lb
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) mask[la * 64 + lb] = m;
    }
}

// ---- part 2
template <int FORM, int NACC>
__global__ void k_rate(double *out, long long *cycles, int iters) {
  const int lane = threadIdx.x & 63;
  const double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane;
  double4_t acc16[NACC];
  double acc4[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) { acc16[i] = (double4_t){0., 0., 0., 0.}; acc4[i] = 0.0; }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (FORM == 16) acc16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[i], 0, 0, 0);
      else acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += (FORM == 16) ? acc16[i][0] + acc16[i][3] : acc4[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---- part 3: the Legendre half of a fused analysis block.  NT tiles (16 n x 4 columns each) per wavefront, KS k-steps of 4 latitude
// pairs; per (tile, k-step): one 512-byte table fragment from global memory (L2), the B operand (4 latitudes x 4 columns, the same for
// the four 4x4 products) from LDS, one MFMA.  Table = [KS][tiles of the whole triangle][64] doubles, shared by all blocks.
template <int NT, int PF>
__global__ __launch_bounds__(256) void k_stream(const double *__restrict__ frag, double *out, int KS, int tiles_total, long long *cycles) {
  static_assert(NT % PF == 0, "ring");
  __shared__ double B[64 * 16];                        // stand-in for the folded Fourier rows of one latitude chunk
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 16; i += 256) B[i] = 1.0 + 1e-6 * i;
  __syncthreads();
  double acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = 0.0;
  const double *fr = frag + (size_t)(wave * NT) * 64 + lane;
  const size_t ks_stride = (size_t)tiles_total * 64;
  double a[PF];                                        // ring: PF fragments (2 VGPRs each) in flight, refilled as they are consumed
#pragma unroll
  for (int d = 0; d < PF; ++d) a[d] = fr[d * 64];
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int g = 0; g < NT / PF; ++g) {
      // the group PF tiles ahead: (ks, g + 1), or (ks + 1, 0) -- clamped on the last pass (re-reads, no branch around the loads)
      const int ksn = (g + 1 < NT / PF) ? ks : min(ks + 1, KS - 1), gn = (g + 1 < NT / PF) ? g + 1 : 0;
      const double *pn = fr + ksn * ks_stride + gn * PF * 64;
      // B (the folded Fourier rows of one wavenumber, 4 latitudes x 4 columns) changes with the wavenumber, i.e. every 3-4 tiles: read
      // from LDS one group of 4 tiles ahead of its use
      double bn = B[((ks * 7 + g * PF) & 63) * 16 + (lane & 15)];
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        const int t = g * PF + d;
        double b = bn;
        if ((d & 3) == 0) { b = bn; bn = B[((ks * 7 + t + 4) & 63) * 16 + (lane & 15)]; }
        const double av = a[d];
        a[d] = pn[d * 64];
        acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, acc[t], 0, 0, 0);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < NT; ++t) s += acc[t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same with 16-byte loads: the fragments of k-steps 2q and 2q + 1 of one tile lie side by side ([KS/2][tiles][64][2]), one
// global_load_dwordx4 per lane brings both
template <int NT, int PF>
__global__ __launch_bounds__(256) void k_stream16(const double *__restrict__ frag, double *out, int KS, int tiles_total) {
  static_assert(NT % PF == 0, "ring");
  __shared__ double B[64 * 16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 16; i += 256) B[i] = 1.0 + 1e-6 * i;
  __syncthreads();
  double acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = 0.0;
  const double2 *fr = (const double2 *)frag + (size_t)(wave * NT) * 64 + lane;
  const size_t ks_stride = (size_t)tiles_total * 64;
  double2 a[PF];
#pragma unroll
  for (int d = 0; d < PF; ++d) a[d] = fr[d * 64];
  for (int ks = 0; ks < KS / 2; ++ks) {
#pragma unroll
    for (int g = 0; g < NT / PF; ++g) {
      const int ksn = (g + 1 < NT / PF) ? ks : min(ks + 1, KS / 2 - 1), gn = (g + 1 < NT / PF) ? g + 1 : 0;
      const double2 *pn = fr + ksn * ks_stride + gn * PF * 64;
      double2 bn = *(const double2 *)&B[((ks * 7 + g * PF) & 31) * 32 + 2 * (lane & 15)];
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        const int t = g * PF + d;
        double2 b = bn;
        if ((d & 3) == 0) { b = bn; bn = *(const double2 *)&B[((ks * 7 + t + 4) & 31) * 32 + 2 * (lane & 15)]; }
        const double2 av = a[d];
        a[d] = pn[d * 64];
        acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(av.x, b.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(av.y, b.y, acc[t], 0, 0, 0);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < NT; ++t) s += acc[t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  // ---- part 1
  unsigned long long *dmask;
  CK(hipMalloc((void **)&dmask, 4096 * 8));
  hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, dmask);
  std::vector<unsigned long long> mask(4096);
  CK(hipMemcpy(mask.data(), dmask, 4096 * 8, hipMemcpyDeviceToHost));
  // hypothesis: A lane = 16 b + 4 k + i?  find, for A lane la, the B lanes it meets and the D lanes it feeds
  printf("part 1: v_mfma_f64_4x4x4f64 lane map (A lane la x B lane lb -> D lanes)\n");
  for (int la : {0, 1, 4, 5, 16, 21, 63}) {
    printf("  A lane %2d meets B lanes:", la);
    for (int lb = 0; lb < 64; ++lb) if (mask[la * 64 + lb]) printf(" %d->D{", lb), [&] { for (int l = 0; l < 64; ++l) if (mask[la * 64 + lb] >> l & 1) printf("%d,", l); }(), printf("}");
    printf("\n");
  }
  // the mapping read off the one-hot runs: A[b][i][k] sits in lane 16 k + 4 b + i, B[b][k][j] in lane 16 k + 4 b + j, D[b][i][j] in lane 16 i + 4 b + j
  {
    int bad = 0;
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb) {
        const int ka = la >> 4, ba = (la >> 2) & 3, ia = la & 3, kb = lb >> 4, bb = (lb >> 2) & 3, jb = lb & 3;
        const unsigned long long want = (ka == kb && ba == bb) ? 1ull << (16 * ia + 4 * ba + jb) : 0ull;
        if (mask[la * 64 + lb] != want) ++bad;
      }
    printf("  mapping A: lane = 16 k + 4 b + i, B: lane = 16 k + 4 b + j, D: lane = 16 i + 4 b + j : %d mismatches of 4096\n", bad);
  }
  // ---- part 2
  double *dout; long long *dcyc;
  CK(hipMalloc((void **)&dout, 1 << 24)); CK(hipMalloc((void **)&dcyc, 1 << 16));
  printf("part 2: issue cost, %d MFMAs per wavefront, cycles per instruction (s_memtime ticks: 100 MHz -> converted with event time)\n", 8 * 20000);
  for (int wps : {1, 2, 4}) {
    const int threads = 64 * 4 * wps, blocks = 256, iters = 20000;
    const double ms16 = time_ms([&] { hipLaunchKernelGGL((k_rate<16, 8>), dim3(blocks), dim3(threads), 0, 0, dout, dcyc, iters); }, 3);
    const double ms4 = time_ms([&] { hipLaunchKernelGGL((k_rate<4, 8>), dim3(blocks), dim3(threads), 0, 0, dout, dcyc, iters); }, 3);
    const double n = 8.0 * iters * wps;      // per SIMD
    printf("  %d wavefront(s) per SIMD: 16x16x4 %.1f ns/instr/SIMD = %.2f TFLOP/s chip;  4x4x4 %.1f ns/instr/SIMD = %.2f TFLOP/s chip\n", wps,
           1e6 * ms16 / n, 2048.0 * n * 1024 / (ms16 * 1e-3) / 1e12, 1e6 * ms4 / n, 512.0 * n * 1024 / (ms4 * 1e-3) / 1e12);
  }
  // ---- part 3: T85: 281 tiles per k-step for the whole triangle, 16 k-steps; 4 wavefronts x 71 tiles; table 281 * 16 * 512 B = 2.3 MB
  {
    const int tiles = 288, KS = 16;
    double *frag;
    CK(hipMalloc((void **)&frag, (size_t)tiles * KS * 64 * 8 + 4096));
    std::vector<double> h((size_t)tiles * KS * 64, 1.0);
    CK(hipMemcpy(frag, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    for (int blocks : {1, 81, 161, 256}) {
      const double ms8 = time_ms([&] { hipLaunchKernelGGL((k_stream<72, 8>), dim3(blocks), dim3(256), 0, 0, frag, dout, KS, tiles, dcyc); }, 20);
      const double ms24 = time_ms([&] { hipLaunchKernelGGL((k_stream<72, 24>), dim3(blocks), dim3(256), 0, 0, frag, dout, KS, tiles, dcyc); }, 20);
      const double ms72 = time_ms([&] { hipLaunchKernelGGL((k_stream<72, 72>), dim3(blocks), dim3(256), 0, 0, frag, dout, KS, tiles, dcyc); }, 20);
      const double ms16a = time_ms([&] { hipLaunchKernelGGL((k_stream16<72, 8>), dim3(blocks), dim3(256), 0, 0, frag, dout, KS, tiles); }, 20);
      const double ms16b = time_ms([&] { hipLaunchKernelGGL((k_stream16<72, 24>), dim3(blocks), dim3(256), 0, 0, frag, dout, KS, tiles); }, 20);
      printf("part 3: 16-byte table loads (two k-steps per load), %3d blocks: %.2f / %.2f us with 8 / 24 loads in flight per wavefront\n", blocks, 1e3 * ms16a, 1e3 * ms16b);
      printf("part 3: T85 fused-analysis Legendre half, %3d blocks x (2 level-fields, all m; %d MFMA 4x4x4 and %.1f MB of table per block): "
             "%.2f / %.2f / %.2f us with 8 / 24 / 72 fragments in flight per wavefront\n", blocks, 4 * 72 * KS, 4 * 72 * KS * 512 / 1e6,
             1e3 * ms8, 1e3 * ms24, 1e3 * ms72);
    }
  }
  return 0;
}
