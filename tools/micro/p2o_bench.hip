// The single-pass (gemm mode 3) layer-1 forward on q16b rows (gemm_p2.hpp, p2_nt_tile<.., XP = 1, ONE>) at T = 32: time, the
// no-DMA / DMA-only ablations and per-step stamps of both wave groups.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lirec_amd/csrc -I include tools/micro/p2o_bench.hip -o tools/micro/p2o_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "gemm_p2.hpp"

using namespace lirec;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill_kernel(unsigned* p, long n, unsigned seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    p[i] = (x & 0x007f007fu) | 0x3c003c00u;      // two small positive bf16 values
  }
}
__global__ void iota_kernel(int* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i; }
template <int ABL, bool K64 = false> static float run(const GemmGroup& g, int G, int nrep, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int r = 0; r < 7; ++r) {
    float ms;
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) {
      if (K64) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_ntg64_kernel<ABL>), dim3(G), dim3(512), 0, 0, g, nrep);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_ntg1_kernel<ABL, true>), dim3(G), dim3(512), 0, 0, g, nrep);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / iters);
  }
  std::sort(t.begin(), t.end());
  return 1e3f * t[3];
}
int main(int argc, char** argv) {
  const int valid = argc > 1 ? atoi(argv[1]) : 14000;
  const int J = 512, D = 6912, nseg = 4;
  const int in_dim[4] = {768, 2048, 2048, 2048}, in_off[4] = {0, 768, 2816, 4864};
  const int Mi = 2048;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  const int rc32 = (valid + 31) & ~31;
  unsigned char *Xc, *Xi, *Wq; float *H1c, *H1i, *bias;
  CK(hipMalloc(&Xc, (long)rc32 * D * 2)); CK(hipMalloc(&Xi, (long)Mi * D * 2)); CK(hipMalloc(&Wq, 2L * J * D * 4));
  CK(hipMalloc(&H1c, (long)rc32 * nseg * J * 4)); CK(hipMalloc(&H1i, (long)Mi * nseg * J * 4)); CK(hipMalloc(&bias, 2L * nseg * J * 4));
  fill_kernel<<<2048, 256>>>((unsigned*)Xc, (long)rc32 * D / 2, 1u); fill_kernel<<<2048, 256>>>((unsigned*)Xi, (long)Mi * D / 2, 2u);
  fill_kernel<<<2048, 256>>>((unsigned*)Wq, 2L * J * D, 3u); CK(hipMemset(bias, 0, 2L * nseg * J * 4));
  int* ident; CK(hipMalloc(&ident, (long)rc32 * 4)); iota_kernel<<<64, 256>>>(ident, rc32);
  int* d_count; CK(hipMalloc(&d_count, 4)); CK(hipMemcpy(d_count, &valid, 4, hipMemcpyHostToDevice));
  long long* st; const long nst = (long)G * 8 * 512;
  CK(hipMalloc(&st, nst * sizeof(long long))); CK(hipMemset(st, 0, nst * sizeof(long long)));
  GemmGroup g; memset(&g, 0, sizeof(g));
  for (int h = 0; h < 2; ++h)
    for (int i = 0; i < nseg; ++i) {
      GemmProblem p; memset(&p, 0, sizeof(p));
      p.A = (const float*)((h == 0 ? Xc : Xi) + 2048L * (in_off[i] / 32)); p.lda = D; p.srow = ident;
      const long wo = (long)h * J * D + (long)J * in_off[i];
      p.B = (const float*)(Wq + 4 * wo); p.ldb = in_dim[i];
      p.bias = bias + (h * nseg + i) * J;
      p.C = (h == 0 ? H1c : H1i) + (long)i * J; p.ldc = (long)nseg * J;
      p.M = h == 0 ? rc32 : Mi; p.N = J; p.K = in_dim[i]; p.dyn = h == 0 ? d_count : nullptr; p.drop_scale = 1.f;
      p.slab = (float*)st;
      g.p[g.nprob++] = p;
    }
  const int nrep = J / 256;
  printf("%d CUs; %d + %d rows (T = 32 shape), single-pass forward on q16b rows\n", G, valid, Mi);
  printf("  whole      %7.1f us\n", run<0>(g, G, nrep, 4));
  printf("  no LDS-DMA %7.1f us\n", run<16>(g, G, nrep, 4));
  printf("  DMA only   %7.1f us\n", run<32>(g, G, nrep, 4));
  // the same on q16c operands (rows AND weights as bf16, 64 of k per 128-byte row): the two-plane kernel, one MFMA per product
  GemmGroup g64 = g;
  {
    int k = 0;
    for (int h = 0; h < 2; ++h)
      for (int i = 0; i < nseg; ++i, ++k) {
        GemmProblem& p = g64.p[k];
        p.A = (const float*)((h == 0 ? Xc : Xi) + 4096L * (in_off[i] / 64)); p.lda = D / 2;
        const long wo = (long)h * J * D + (long)J * in_off[i];
        p.B = (const float*)(Wq + 2 * wo); p.ldb = in_dim[i] / 2; p.K = in_dim[i] / 2;
      }
  }
  printf("q16c rows and weights, 64 of k per step:\n");
  printf("  whole      %7.1f us\n", run<0, true>(g64, G, nrep, 4));
  printf("  no LDS-DMA %7.1f us\n", run<16, true>(g64, G, nrep, 4));
  printf("  DMA only   %7.1f us\n", run<32, true>(g64, G, nrep, 4));
  // the weight gradient's single-pass form on the same q16c rows (dZ1: hi plane, bf16): whole / without the requests / requests alone
  {
    unsigned short *Zc, *Zi; float *dW, *db, *slab, *dslab;
    CK(hipMalloc(&Zc, (long)rc32 * nseg * J * 2)); CK(hipMalloc(&Zi, (long)Mi * nseg * J * 2));
    fill_kernel<<<2048, 256>>>((unsigned*)Zc, (long)rc32 * nseg * J / 2, 5u); fill_kernel<<<2048, 256>>>((unsigned*)Zi, (long)Mi * nseg * J / 2, 6u);
    CK(hipMalloc(&dW, 2L * J * D * 4)); CK(hipMalloc(&db, 2L * nseg * J * 4));
    CK(hipMalloc(&slab, 2L * G * P2::SLAB * 4)); CK(hipMalloc(&dslab, 2L * G * 256 * 4));
    GemmGroup gw; memset(&gw, 0, sizeof(gw));
    gw.dyn_is_k = 1;
    for (int h = 0; h < 2; ++h)
      for (int i = 0; i < nseg; ++i) {
        GemmProblem w; memset(&w, 0, sizeof(w));
        const unsigned short* zh = h == 0 ? Zc : Zi;
        w.A = (const float*)(zh + (long)i * J); w.A_lo = zh + (long)i * J; w.lda = (long)nseg * J;
        w.B = (const float*)((h == 0 ? Xc : Xi) + 4096L * (in_off[i] / 64)); w.ldb = D; w.srow = ident;
        w.C = dW + ((long)h * J * D + (long)J * in_off[i]); w.ldc = in_dim[i];
        w.M = J; w.N = in_dim[i]; w.K = h == 0 ? rc32 : Mi; w.dyn = h == 0 ? d_count : nullptr;
        w.dbias = db + (h * nseg + i) * J; w.drop_scale = 1.f;
        gw.p[gw.nprob++] = w;
      }
    gw.p[0].slab = slab; gw.p[0].dbias_slab = dslab;
    auto tn = [&](int abl) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      std::vector<float> t;
      for (int r = 0; r < 7; ++r) {
        float ms;
        CK(hipEventRecord(e0));
        for (int it = 0; it < 4; ++it) {
          if (abl == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_kernel<16, true, 1, true>), dim3(G), dim3(512), 0, 0, gw, nrep);
          else if (abl == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_kernel<32, true, 1, true>), dim3(G), dim3(512), 0, 0, gw, nrep);
          else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_kernel<0, true, 1, true>), dim3(G), dim3(512), 0, 0, gw, nrep);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / 4);
      }
      std::sort(t.begin(), t.end());
      return 1e3f * t[3];
    };
    printf("weight gradient, single pass, q16c rows (32 rows of k per step):\n");
    printf("  whole      %7.1f us\n", tn(0));
    printf("  no LDS-DMA %7.1f us\n", tn(16));
    printf("  DMA only   %7.1f us\n", tn(32));
  }
  const bool k64_stamps = argc > 2 && atoi(argv[2]) == 64;
  for (int it = 0; it < 3; ++it) {
    if (k64_stamps) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_ntg64_kernel<1024>), dim3(G), dim3(512), 0, 0, g64, nrep);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_ntg1_kernel<1024, true>), dim3(G), dim3(512), 0, 0, g, nrep);
  }
  CK(hipDeviceSynchronize());
  std::vector<long long> h(nst);
  CK(hipMemcpy(h.data(), st, nst * sizeof(long long), hipMemcpyDeviceToHost));
  for (int b : {64, 65})
    for (int w = 0; w < 8; w += 4) {
      const long long* s5 = h.data() + (long)(b * 8 + w) * 512;
      printf("stamps block %d wave %d: step: barrier(even) even-half-work barrier(odd) odd-half-work | total (cycles)\n", b, w);
      for (int t = 0; t < 12 && s5[6 * t]; ++t)
        printf("   %2d: %6lld %6lld %6lld %6lld | %6lld\n", t, s5[6 * t + 1] - s5[6 * t], s5[6 * t + 2] - s5[6 * t + 1], s5[6 * t + 3] - s5[6 * t + 2],
               s5[6 * t + 4] - s5[6 * t + 3], t > 0 ? s5[6 * t] - s5[6 * (t - 1)] : 0LL);
      printf("   tile: entry -> first step landed %lld, k loop %lld, epilogue %lld\n", s5[501] - s5[500], s5[502] - s5[501], s5[503] - s5[502]);
    }
  return 0;
}
