// The layer-1 forward on rows fetched from the fp32 block itself (gemm_p2.hpp, p2_nt_tile XF: split in LDS by the loader waves,
// q32b rows written out for the weight gradient) beside the staged form, at the bench shape: bit-identical H1 and q32b rows, then
// times (interleaved rounds) and per-step stamps.  A slim sibling of p2_bench.hip (three kernel instantiations: builds in 2 min).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lirec_amd/csrc -I include tools/micro/p2x_bench.hip -o tools/micro/p2x_bench.bin
//   tools/micro/p2x_bench.bin [valid_ctx_rows=7221]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "gemm_p2.hpp"

using namespace lirec;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, long n, unsigned seed, int mode) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    float v = ((float)(x >> 8) / 8388608.0f - 1.0f);
    if (mode == 2) v = v > 0.f ? v : 0.f;          // post-relu features
    if (mode == 3) v *= 0.03f;                     // weights
    p[i] = v;
  }
}
__global__ void q32_kernel(const float* src, long ld, const int* rows, unsigned char* dst, long R, int C) {
  const long n8 = R * C / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long row = i / (C / 8); const int c8 = (int)(i - row * (C / 8));
    const float* q = src + (rows ? (long)rows[row] : row) * ld + 8 * c8;
    p2_store_q32b(dst, row, c8, C / 32, *reinterpret_cast<const f32x4*>(q), *reinterpret_cast<const f32x4*>(q + 4));
  }
}
__global__ void iota_kernel(int* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i; }
static float* dalloc_f(long n) { float* p; CK(hipMalloc(&p, n * sizeof(float))); return p; }

int main(int argc, char** argv) {
  const int valid = argc > 1 ? atoi(argv[1]) : 7221;
  const int J = 512, D = 6912, nseg = 4;
  const int in_dim[4] = {768, 2048, 2048, 2048}, in_off[4] = {0, 768, 2816, 4864};
  const int Mc = 18432, Mi = 1024;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  const int rc32 = (valid + 31) & ~31;
  printf("%d CUs; valid context rows %d of %d\n", G, valid, Mc);
  float* Xc = dalloc_f((long)Mc * D); float* Xi = dalloc_f((long)Mi * D);
  float* W = dalloc_f(2L * J * D); float* bias = dalloc_f(2L * nseg * J);
  unsigned char* Xcq = (unsigned char*)dalloc_f((long)rc32 * D); unsigned char* Xiq = (unsigned char*)dalloc_f((long)Mi * D);
  unsigned char* Xcq2 = (unsigned char*)dalloc_f((long)rc32 * D); unsigned char* Xiq2 = (unsigned char*)dalloc_f((long)Mi * D);
  unsigned char* Wq = (unsigned char*)dalloc_f(2L * J * D);
  float* H1c = dalloc_f((long)rc32 * nseg * J); float* H1i = dalloc_f((long)Mi * nseg * J);
  float* H1c2 = dalloc_f((long)rc32 * nseg * J); float* H1i2 = dalloc_f((long)Mi * nseg * J);
  int* d_count; CK(hipMalloc(&d_count, sizeof(int))); CK(hipMemcpy(d_count, &valid, sizeof(int), hipMemcpyHostToDevice));
  fill_kernel<<<2048, 256>>>(Xc, (long)Mc * D, 1u, 2); fill_kernel<<<2048, 256>>>(Xi, (long)Mi * D, 2u, 2);
  fill_kernel<<<2048, 256>>>(W, 2L * J * D, 3u, 3); fill_kernel<<<64, 256>>>(bias, 2L * nseg * J, 4u, 1);
  // the valid rows, spread over the block as the product's are; the list's tail repeats the last valid row (the staging pass's lists)
  std::vector<int> hl(rc32);
  for (int j = 0; j < rc32; ++j) { const int jj = j < valid ? j : valid - 1; hl[j] = (int)((long)jj * Mc / valid); }
  int* lst; CK(hipMalloc(&lst, (long)rc32 * 4)); CK(hipMemcpy(lst, hl.data(), (long)rc32 * 4, hipMemcpyHostToDevice));
  int* ident; CK(hipMalloc(&ident, (long)Mi * 4)); iota_kernel<<<8, 256>>>(ident, Mi);
  q32_kernel<<<2048, 256>>>(Xc, D, lst, Xcq, rc32, D); q32_kernel<<<2048, 256>>>(Xi, D, nullptr, Xiq, Mi, D);
  for (int h = 0; h < 2; ++h)
    for (int i = 0; i < nseg; ++i) {
      const long wo = (long)h * J * D + (long)J * in_off[i];
      q32_kernel<<<512, 256>>>(W + wo, in_dim[i], nullptr, Wq + 4 * wo, J, in_dim[i]);
    }
  GemmGroup gf, gx;
  memset(&gf, 0, sizeof(gf));
  for (int h = 0; h < 2; ++h)
    for (int i = 0; i < nseg; ++i) {
      GemmProblem p;
      memset(&p, 0, sizeof(p));
      p.A = (const float*)((h == 0 ? Xcq : Xiq) + 4096L * (in_off[i] / 32)); p.lda = D;
      const long wo = (long)h * J * D + (long)J * in_off[i];
      p.B = (const float*)(Wq + 4 * wo); p.ldb = in_dim[i];
      p.bias = bias + (h * nseg + i) * J;
      p.C = (h == 0 ? H1c : H1i) + (long)i * J; p.ldc = (long)nseg * J;
      p.M = h == 0 ? rc32 : Mi; p.N = J; p.K = in_dim[i];
      p.dyn = h == 0 ? d_count : nullptr;
      p.drop_scale = 1.f;
      gf.p[gf.nprob++] = p;
    }
  gx = gf;
  for (int k = 0; k < gf.nprob; ++k) {
    const int h = k / nseg, i = k % nseg;
    gx.p[k].srow = h == 0 ? lst : ident;
    gx.p[k].A = (h == 0 ? Xc : Xi) + in_off[i]; gx.p[k].lda = D;
    gx.p[k].xq_out = (h == 0 ? Xcq2 : Xiq2) + 4096L * (in_off[i] / 32); gx.p[k].ld_xq = D;
    gx.p[k].C = (h == 0 ? H1c2 : H1i2) + (long)i * J;
  }
  GemmGroup gx0 = gx;
  for (int k = 0; k < gf.nprob; ++k) gx0.p[k].xq_out = nullptr;
  const int nrep = J / 256;
  auto fwd = [&]() { hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_nt_kernel<0>), dim3(G), dim3(512), 0, 0, gf, nrep); };
  auto fwdx = [&](const GemmGroup& gg) { hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_ntx_kernel<0>), dim3(G), dim3(512), 0, 0, gg, nrep); };
  CK(hipMemset(H1c, 0xff, (long)rc32 * nseg * J * 4)); CK(hipMemset(H1c2, 0xff, (long)rc32 * nseg * J * 4));
  CK(hipMemset(H1i, 0xff, (long)Mi * nseg * J * 4)); CK(hipMemset(H1i2, 0xff, (long)Mi * nseg * J * 4));
  CK(hipMemset(Xcq2, 0xee, (long)rc32 * D * 4)); CK(hipMemset(Xiq2, 0xee, (long)Mi * D * 4));
  fwd(); fwdx(gx);
  CK(hipDeviceSynchronize());
  auto differ = [&](const void* a, const void* b, long n) {
    std::vector<unsigned> x(n), y(n);
    CK(hipMemcpy(x.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), b, n * 4, hipMemcpyDeviceToHost));
    long d = 0;
    for (long e = 0; e < n; ++e) d += x[e] != y[e];
    return d;
  };
  printf("fp32 rows split on the way in vs the staged form: H1 %ld + %ld elements differ; q32b rows written out: %ld + %ld words differ (tail rows included)\n",
         differ(H1c, H1c2, (long)valid * nseg * J), differ(H1i, H1i2, (long)Mi * nseg * J), differ(Xcq, Xcq2, (long)rc32 * D), differ(Xiq, Xiq2, (long)Mi * D));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ta, tx, tx0;
  for (int r = 0; r < 9; ++r) {
    float ms;
    CK(hipEventRecord(e0)); for (int it = 0; it < 4; ++it) fwd(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ta.push_back(ms / 4);
    CK(hipEventRecord(e0)); for (int it = 0; it < 4; ++it) fwdx(gx); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tx.push_back(ms / 4);
    CK(hipEventRecord(e0)); for (int it = 0; it < 4; ++it) fwdx(gx0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tx0.push_back(ms / 4);
  }
  std::sort(ta.begin(), ta.end()); std::sort(tx.begin(), tx.end()); std::sort(tx0.begin(), tx0.end());
  printf("forward: staged q32b rows median %.1f us (min %.1f) | fp32 rows split on the way in, q32b rows written out %.1f (%.1f) | the same, nothing written out %.1f (%.1f)\n",
         1e3 * ta[4], 1e3 * ta[0], 1e3 * tx[4], 1e3 * tx[0], 1e3 * tx0[4], 1e3 * tx0[0]);
  {
    long long* st; const long nst = (long)G * 8 * 512;
    CK(hipMalloc(&st, nst * sizeof(long long))); CK(hipMemset(st, 0, nst * sizeof(long long)));
    for (int i = 0; i < gx.nprob; ++i) gx.p[i].slab = reinterpret_cast<float*>(st);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_ntx_kernel<1024>), dim3(G), dim3(512), 0, 0, gx, nrep);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(nst);
    CK(hipMemcpy(h.data(), st, nst * sizeof(long long), hipMemcpyDeviceToHost));
    for (int w = 0; w < 8; w += 4) {
      const long long* s5 = h.data() + (long)(65 * 8 + w) * 512;
      printf("fp32-rows forward, stamps block 65 wave %d: step: barrier(even) even-half-work barrier(odd) odd-half-work | total (cycles)\n", w);
      for (int t = 0; t < 10 && s5[6 * t]; ++t)
        printf("   %2d: %6lld %6lld %6lld %6lld | %6lld\n", t, s5[6 * t + 1] - s5[6 * t], s5[6 * t + 2] - s5[6 * t + 1], s5[6 * t + 3] - s5[6 * t + 2],
               s5[6 * t + 4] - s5[6 * t + 3], t > 0 ? s5[6 * t] - s5[6 * (t - 1)] : 0LL);
      if (w == 4) {
        printf("   the loader group's even half: requests issued | wait for A(t+1) | conversion reads back | split + writes (+ stores) | fragment reads\n");
        for (int t = 1; t < 10 && s5[6 * t]; ++t)
          printf("   %2d: %6lld %6lld %6lld %6lld %6lld\n", t, s5[384 + 4 * t] - s5[6 * t + 1], s5[385 + 4 * t] - s5[384 + 4 * t], s5[386 + 4 * t] - s5[385 + 4 * t],
                 s5[387 + 4 * t] - s5[386 + 4 * t], s5[6 * t + 2] - s5[387 + 4 * t]);
      }
    }
  }
  return 0;
}
