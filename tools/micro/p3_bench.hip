// Where a k-step of gemm_p3_kernel goes: the kernel itself on the gate's forward shape (1024 x 3072 x 3072, q32b operands filled
// with random bf16 halves), whole, without LDS-DMA, without MFMAs, and with in-kernel stamps of workgroup 0.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lirec_amd/csrc -I include tools/micro/p3_bench.hip -o tools/micro/p3_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_p3.hpp"
using namespace lirec;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_halves(unsigned short* p, long n, unsigned seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (unsigned short)(0x3c00u | (x & 0x7fu) | ((x >> 8) & 0x8000u));       // +-[2^-7, 2^-6)
  }
}

template <int KIND, int ABL>
static float run(const GemmGroup& g, int tiles, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p3_kernel<KIND, ABL>), dim3(tiles), dim3(512), 0, 0, g);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / iters < best) best = ms / iters;
  }
  return best * 1e3f;
}

int main() {
  const int M = 1024, N = 3072, K = 3072;
  unsigned char *A, *B; float *C, *bias; long long* stamps;
  CK(hipMalloc(&A, 4L * M * K)); CK(hipMalloc(&B, 4L * N * K)); CK(hipMalloc(&C, 4L * M * N)); CK(hipMalloc(&bias, 4L * N));
  CK(hipMalloc(&stamps, 8 * 8 * 128)); CK(hipMemset(stamps, 0, 8 * 8 * 128)); CK(hipMemset(bias, 0, 4L * N));
  fill_halves<<<1024, 256>>>((unsigned short*)A, 2L * M * K, 1u);
  fill_halves<<<1024, 256>>>((unsigned short*)B, 2L * N * K, 7u);
  GemmGroup g; memset(&g, 0, sizeof(g)); g.nprob = 1;
  GemmProblem p; memset(&p, 0, sizeof(p));
  p.A = (const float*)A; p.lda = K; p.B = (const float*)B; p.ldb = K; p.C = C; p.ldc = N; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.drop_scale = 1.f; p.slab = (float*)stamps; p.aux = C; p.ldaux = N;
  setvbuf(stdout, nullptr, _IONBF, 0);
  g.p[0] = p;
  const int tiles = (M / 128) * (N / 128);
  printf("gate forward shape %d x %d x %d, %d tiles of 128 x 128, 96 k-steps\n", M, N, K, tiles);
  printf("NT whole          %7.1f us\n", run<0, 0>(g, tiles, 20));
  printf("NT no LDS-DMA     %7.1f us\n", run<0, 1>(g, tiles, 20));
  printf("NT no MFMA        %7.1f us\n", run<0, 2>(g, tiles, 20));
  printf("NT neither        %7.1f us\n", run<0, 3>(g, tiles, 20));
  printf("NN whole          %7.1f us\n", run<1, 0>(g, tiles, 20));
  printf("NN no LDS-DMA     %7.1f us\n", run<1, 1>(g, tiles, 20));
  run<0, 4>(g, tiles, 1);
  std::vector<long long> st(8 * 128);
  CK(hipMemcpy(st.data(), stamps, 8 * 8 * 128, hipMemcpyDeviceToHost));
  printf("stamps of workgroup 0 (cycles, relative to the loader's top of step 0):\n");
  printf("%4s | loader: %8s %8s %8s %8s | compute: %8s %8s %8s\n", "t", "top", "landed", "barrier", "issued", "top", "barrier", "done");
  for (int t = 0; t < 96; t += (t < 8 ? 1 : 8)) {
    const long long* s = &st[8 * t]; const long long z = st[0];
    printf("%4d | %17lld %8lld %8lld %8lld | %17lld %8lld %8lld\n", t, s[0] - z, s[1] - z, s[2] - z, s[3] - z, s[4] - z, s[5] - z, s[6] - z);
  }
  return 0;
}
