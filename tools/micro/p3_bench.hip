// Where a k-step of gemm_p3_kernel goes: the kernel itself on the gate's three shapes (q32b operands filled with random bf16
// halves) -- forward 1024 x 3072 x 3072, data gradient 2 x (1024 x 1536 x 3072), weight gradient 3072 x 3072 x 1024 -- whole,
// without LDS-DMA, without MFMAs, and with in-kernel stamps of workgroup 0's first compute wave.
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

template <int MI, int NI, int EPI, int ABL>
static float run(const GemmGroup& g, int grid, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p3_kernel<MI, NI, EPI, false, ABL>), dim3(grid), dim3(512), 0, 0, g);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / iters < best) best = ms / iters;
  }
  return best * 1e3f;
}

static void setup(GemmGroup& g, int BM, int BN, int xm_force = -1) {
  int tn = 0;
  for (int i = 0; i < g.nprob; ++i) tn += g.p[i].N / BN;
  g.p3_tm = g.p[0].M / BM; g.p3_tn = tn; g.p3_xm = 0;
  double best = 0;
  for (int xm = 1; xm <= 8; xm *= 2) {
    const int xn = 8 / xm;
    if (g.p3_tm % xm || tn % xn) continue;
    const double cost = (double)(g.p3_tm / xm) * BM + (double)(tn / xn) * BN;
    if (g.p3_xm == 0 || cost < best) { best = cost; g.p3_xm = xm; }
  }
  if (xm_force >= 0) g.p3_xm = xm_force;
}

template <int MI, int NI>
static void shapes(const char* tag, unsigned char* A, unsigned char* B, float* C, float* bias, long long* stamps) {
  constexpr int BM = 32 * MI, BN = 32 * NI;
  const int n = 1024, N = 3072, K = 3072;
  printf("---- %s: tiles %d x %d, %d slots of %d bytes\n", tag, BM, BN, P3<MI, NI>::NSLOT, P3<MI, NI>::SLOT);
  GemmProblem p; memset(&p, 0, sizeof(p));
  p.drop_scale = 1.f; p.slab = (float*)stamps;
  {   // forward
    GemmGroup g; memset(&g, 0, sizeof(g)); g.nprob = 1;
    GemmProblem q = p; q.A = (const float*)A; q.lda = K; q.B = (const float*)B; q.ldb = K; q.C = C; q.ldc = N; q.bias = bias; q.M = n; q.N = N; q.K = K;
    g.p[0] = q; setup(g, BM, BN);
    const int tiles = g.p3_tm * g.p3_tn, grid = tiles < 256 ? tiles : 256;
    printf("forward  %d x %d x %d: %d tiles, grid %d, XCD blocks %d x %d\n", n, N, K, tiles, grid, g.p3_xm, g.p3_xm ? 8 / g.p3_xm : 0);
    printf("  whole          %7.1f us\n", run<MI, NI, 0, 0>(g, grid, 20));
    printf("  no LDS-DMA     %7.1f us\n", run<MI, NI, 0, 1>(g, grid, 20));
    printf("  no MFMA        %7.1f us\n", run<MI, NI, 0, 2>(g, grid, 20));
    printf("  neither        %7.1f us\n", run<MI, NI, 0, 3>(g, grid, 20));
    printf("  no fragment reads (MFMAs + LDS-DMA)            %7.1f us\n", run<MI, NI, 0, 16>(g, grid, 20));
    printf("  no fragment reads, no LDS-DMA (MFMAs alone)    %7.1f us\n", run<MI, NI, 0, 17>(g, grid, 20));
    printf("  nothing but the barriers                       %7.1f us\n", run<MI, NI, 0, 19>(g, grid, 20));
    printf("  fragment reads alone, no barriers              %7.1f us\n", run<MI, NI, 0, 35>(g, grid, 20));
    printf("  MFMAs + fragment reads, no barriers, no DMA    %7.1f us\n", run<MI, NI, 0, 33>(g, grid, 20));
    printf("  LDS-DMA alone (no reads, no MFMA)              %7.1f us\n", run<MI, NI, 0, 18>(g, grid, 20));
    GemmGroup g0 = g; g0.p3_xm = 0;
    printf("  whole, column-major tile order %7.1f us\n", run<MI, NI, 0, 0>(g0, grid, 20));
    CK(hipMemset(stamps, 0, 8 * 4 * 128));
    run<MI, NI, 0, 4>(g, grid, 1);
    std::vector<long long> st(4 * 128);
    CK(hipMemcpy(st.data(), stamps, 8 * 4 * 128, hipMemcpyDeviceToHost));
    printf("  stamps of workgroup 0, compute wave 0 (cycles from the top of step 0; even steps): top, barrier passed, MFMAs issued\n");
    for (int t = 0; t < 96; t += (t < 8 ? 2 : 16)) printf("  %4d | %8lld %8lld %8lld\n", t, st[4 * t] - st[0], st[4 * t + 1] - st[0], st[4 * t + 2] - st[0]);
  }
  {   // data gradient: two column ranges of the output
    GemmGroup g; memset(&g, 0, sizeof(g)); g.nprob = 2;
    for (int h = 0; h < 2; ++h) {
      GemmProblem q = p; q.A = (const float*)A; q.lda = N; q.B = (const float*)(B + 4096L * (h * 1536 / 32) * (N / 32)); q.ldb = N;
      q.C = C + h * 1536; q.ldc = K; q.aux = C + h * 1536; q.ldaux = K; q.M = n; q.N = 1536; q.K = N;
      g.p[h] = q;
    }
    setup(g, BM, BN);
    const int tiles = g.p3_tm * g.p3_tn, grid = tiles < 256 ? tiles : 256;
    printf("data gradient 2 x (%d x 1536 x %d): %d tiles, grid %d, XCD blocks %d x %d\n", n, N, tiles, grid, g.p3_xm, g.p3_xm ? 8 / g.p3_xm : 0);
    printf("  whole          %7.1f us\n", run<MI, NI, 1, 0>(g, grid, 20));
    printf("  no LDS-DMA     %7.1f us\n", run<MI, NI, 1, 1>(g, grid, 20));
  }
  {   // weight gradient
    GemmGroup g; memset(&g, 0, sizeof(g)); g.nprob = 1;
    GemmProblem q = p; q.A = (const float*)A; q.lda = n; q.B = (const float*)B; q.ldb = n; q.C = C; q.ldc = K; q.M = N; q.N = K; q.K = n; q.dbias = bias; q.dbias_set = 1;
    g.p[0] = q; setup(g, BM, BN);
    const int tiles = g.p3_tm * g.p3_tn, grid = tiles < 256 ? tiles : 256;
    printf("weight gradient %d x %d x %d: %d tiles, grid %d, XCD blocks %d x %d\n", N, K, n, tiles, grid, g.p3_xm, g.p3_xm ? 8 / g.p3_xm : 0);
    printf("  whole          %7.1f us\n", run<MI, NI, 2, 0>(g, grid, 20));
    printf("  no LDS-DMA     %7.1f us\n", run<MI, NI, 2, 1>(g, grid, 20));
    printf("  no MFMA        %7.1f us\n", run<MI, NI, 2, 2>(g, grid, 20));
    for (int xm = 0; xm <= 8; xm = xm ? xm * 2 : 1) {
      if (xm && (g.p3_tm % xm || g.p3_tn % (8 / xm))) continue;
      GemmGroup gx = g; gx.p3_xm = xm;
      printf("  whole, XCD blocks %d x %d  %7.1f us\n", xm, xm ? 8 / xm : 0, run<MI, NI, 2, 0>(gx, grid, 20));
    }
  }
}

int main() {
  unsigned char *A, *B; float *C, *bias; long long* stamps;
  CK(hipMalloc(&A, 4L * 3072 * 3072)); CK(hipMalloc(&B, 4L * 3072 * 3072)); CK(hipMalloc(&C, 4L * 3072 * 3072)); CK(hipMalloc(&bias, 4L * 3072));
  CK(hipMalloc(&stamps, 8 * 4 * 128)); CK(hipMemset(bias, 0, 4L * 3072));
  fill_halves<<<1024, 256>>>((unsigned short*)A, 2L * 3072 * 3072, 1u);
  fill_halves<<<1024, 256>>>((unsigned short*)B, 2L * 3072 * 3072, 7u);
  setvbuf(stdout, nullptr, _IONBF, 0);
  shapes<4, 3>("MI 4, NI 3", A, B, C, bias, stamps);
  return 0;
}
