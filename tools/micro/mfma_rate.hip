// Microbenchmark: issue rate of v_mfma_f32_32x32x16_bf16 on gfx950 for (a) independent accumulators and
// (b) the split-precision pattern (three back-to-back MFMAs into the same accumulator), at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int DEP>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (__bf16)in[(threadIdx.x * 8 + j + i) & 1023];
      b[i][j] = (__bf16)in[(threadIdx.x * 8 + j + 7 * i + 3) & 1023];
    }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
      for (int d = 0; d < DEP; ++d)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + d) & 3], b[(i + 2 * d) & 3], acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int DEP>
void run(const char* name, int blocks_per_cu, const float* in, float* out) {
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, DEP>), dim3(grid), dim3(256), 0, 0, in, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, DEP>), dim3(grid), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)grid * 4 * iters * NACC * DEP;          // MFMAs
  printf("%-28s waves/SIMD %d : %.3f ms  %.0f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", name, blocks_per_cu, ms,
         mf * 32768.0 / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * NACC * DEP * blocks_per_cu));
}

int main() {
  float *in, *out;
  hipMalloc(&in, 4096); hipMalloc(&out, 256 * 8 * 256 * 4);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  for (int w : {1, 2, 4}) {
    run<8, 1>("8 independent acc", w, in, out);
    run<4, 1>("4 independent acc", w, in, out);
    run<2, 1>("2 independent acc", w, in, out);
    run<2, 3>("2 acc x 3 dependent", w, in, out);
    run<4, 3>("4 acc x 3 dependent", w, in, out);
    run<1, 3>("1 acc x 3 dependent", w, in, out);
  }
  return 0;
}
