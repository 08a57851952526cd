// Microbenchmark: v_mfma_f32_32x32x16_bf16 rate when the same wave also issues the other work of the
// split-precision k-loop per MFMA: V VALU ops (convert/sub/shift), R ds_read_b128, and per 6 MFMAs W ds_write_b64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V, int R, int W, int BAR>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[32768];
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (__bf16)in[(threadIdx.x * 8 + j + i) & 1023];
      b[i][j] = (__bf16)in[(threadIdx.x * 8 + j + 7 * i + 3) & 1023];
    }
  for (int i = threadIdx.x; i < 8192; i += 256) ((float*)smem)[i] = in[i & 1023];
  __syncthreads();
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v0 = in[threadIdx.x], v1 = in[threadIdx.x + 1], v2 = in[threadIdx.x + 2], v3 = in[threadIdx.x + 3];
  const int roff = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (R) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(smem + ((roff + 1024 * (3 * i + d) + 8192 * r) & 32767));
            a[(i + d + r) & 3] = t;
          }
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + d) & 3], b[(i + 2 * d) & 3], acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < V; ++q) {   // a dependent chain of cheap VALU ops on registers the MFMA does not touch
          if ((q & 3) == 0) v0 = v0 * 1.0001f + v1;
          else if ((q & 3) == 1) v1 = v1 - v2;
          else if ((q & 3) == 2) v2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v2) << 1);
          else v3 = v3 + v0;
        }
      }
    }
    if (W) {
#pragma unroll
      for (int w = 0; w < W; ++w)
        *reinterpret_cast<float2*>(smem + ((threadIdx.x * 8 + 2048 * w + 16384) & 32767)) = make_float2(v0, v1);
    }
    if (BAR) __syncthreads();
  }
  float s = v0 + v1 + v2 + v3;
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 4; ++i) s += (float)a[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V, int R, int W, int BAR>
void run(int blocks_per_cu, const float* in, float* out) {
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V, R, W, BAR>), dim3(grid), dim3(256), 0, 0, in, out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, R, W, BAR>), dim3(grid), dim3(256), 0, 0, in, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)grid * 4 * iters * 6;
  printf("VALU/MFMA %d  ds_read_b128/MFMA %d  ds_write_b64/6MFMA %d  barrier %d  waves/SIMD %d : %.3f ms  %.0f TFLOP/s (%.1f ns per MFMA per SIMD)\n",
         V, R, W, BAR, blocks_per_cu, ms, mf * 32768.0 / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * 6 * blocks_per_cu));
}

int main() {
  float *in, *out;
  (void)hipMalloc(&in, 8192); (void)hipMalloc(&out, 256 * 8 * 256 * 4);
  std::vector<float> h(2048);
  for (int i = 0; i < 2048; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  (void)hipMemcpy(in, h.data(), 8192, hipMemcpyHostToDevice);
  for (int w : {2, 4}) {
    run<0, 0, 0, 0>(w, in, out);
    run<3, 0, 0, 0>(w, in, out);
    run<6, 0, 0, 0>(w, in, out);
    run<9, 0, 0, 0>(w, in, out);
    run<0, 1, 0, 0>(w, in, out);
    run<0, 2, 0, 0>(w, in, out);
    run<6, 1, 0, 0>(w, in, out);
    run<6, 1, 4, 0>(w, in, out);
    run<6, 1, 4, 1>(w, in, out);
    run<0, 0, 0, 1>(w, in, out);
  }
  return 0;
}
