// Stand-alone check + timing of the layer-1 planes kernels (lirec_amd/csrc/gemm_p2.hpp) at the bench shape.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lirec_amd/csrc -I include tools/micro/p2_bench.hip -o tools/micro/p2_bench.bin
//   tools/micro/p2_bench.bin [valid_ctx_rows=7096] [iters=20]
// Exact-integer operands (bit-exact against a naive kernel: catches every layout error), then random operands against an
// fp64 reference, then timings.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "gemm_p2.hpp"

using namespace lirec;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, long n, unsigned seed, int mode) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    float v;
    if (mode == 0) v = (float)((int)(x % 9u) - 4);                        // small integers: exact in bf16
    else if (mode == 1) { v = ((float)(x >> 8) / 8388608.0f - 1.0f); }     // uniform [-1, 1)
    else { v = ((float)(x >> 8) / 8388608.0f - 1.0f); v = v > 0.f ? v : 0.f; }   // post-relu features
    if (mode == 3) v = ((float)(x >> 8) / 8388608.0f - 1.0f) * 0.03f;      // weights
    p[i] = v;
  }
}
__global__ void split_kernel(const float* src, unsigned short* hi, unsigned short* lo, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
    uint2 h, l;
    split4(v, h, l);
    *reinterpret_cast<uint2*>(hi + 4 * i) = h;
    *reinterpret_cast<uint2*>(lo + 4 * i) = l;
  }
}
// fp32 row-major [R][C] (multiples of 32) -> q32b (gemm_p2.hpp)
__global__ void q32_kernel(const float* src, unsigned char* dst, long R, int C) {
  const long n8 = R * C / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long row = i / (C / 8); const int c8 = (int)(i - row * (C / 8));
    p2_store_q32b(dst, row, c8, C / 32, *reinterpret_cast<const f32x4*>(src + 8 * i), *reinterpret_cast<const f32x4*>(src + 8 * i + 4));
  }
}
__global__ void q16_kernel(const float* src, unsigned char* dst, long R, int C) {
  const long n8 = R * C / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long row = i / (C / 8); const int c8 = (int)(i - row * (C / 8));
    p2_store_q16b(dst, row, c8, C / 32, *reinterpret_cast<const f32x4*>(src + 8 * i), *reinterpret_cast<const f32x4*>(src + 8 * i + 4));
  }
}
__global__ void iota_kernel(int* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i; }
// reference forward: C[r][c] = relu(sum_k X[r][xoff+k] W[c][k] + b[c]) in fp64
__global__ void ref_nt_kernel(const float* X, long ldx, int xoff, const float* W, int K, const float* bias, int rows, int N, double* C, long ldc, int coff) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), r = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (r >= rows || c >= N) return;
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += (double)X[(long)r * ldx + xoff + k] * (double)W[(long)c * K + k];
  s += bias[c];
  C[(long)r * ldc + coff + c] = s > 0.0 ? s : 0.0;
}
// reference weight gradient: dW[m][n] = sum_r Z[r][zoff+m] X[r][xoff+n]
__global__ void ref_tn_kernel(const float* Z, long ldz, int zoff, const float* X, long ldx, int xoff, int rows, int M, int N, double* dW) {
  const int n = blockIdx.x * 16 + (threadIdx.x & 15), m = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (m >= M || n >= N) return;
  double s = 0.0;
  for (int r = 0; r < rows; ++r) s += (double)Z[(long)r * ldz + zoff + m] * (double)X[(long)r * ldx + xoff + n];
  dW[(long)m * N + n] = s;
}

static float* dalloc_f(long n) { float* p; CK(hipMalloc(&p, n * sizeof(float))); return p; }
static unsigned short* dalloc_h(long n) { unsigned short* p; CK(hipMalloc(&p, n * sizeof(unsigned short))); return p; }

int main(int argc, char** argv) {
  const int valid = argc > 1 ? atoi(argv[1]) : 7096;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  const int only = argc > 3 ? atoi(argv[3]) : -1;        // >= 0: timing of that ablation index only, no checks (profiling runs)
  const int J = 512, D = 6912, nseg = 4;
  const int in_dim[4] = {768, 2048, 2048, 2048}, in_off[4] = {0, 768, 2816, 4864};
  const int Mc = 18432, Mi = 1024;                       // static rows of the two heads
  const int rows_c = valid, rows_i = Mi;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  printf("device %s, %d CUs; valid ctx rows %d\n", prop.name, G, valid);

  // operands (fp32 originals + planes).  Feature planes hold only rows32(valid) rows like the library's.
  const int rc32 = (rows_c + 31) & ~31;
  float* Xc = dalloc_f((long)rc32 * D); float* Xi = dalloc_f((long)Mi * D);
  float* W = dalloc_f(2L * J * D);      // [head][seg rows...]: W1 of head h, seg i at W + h*J*D + J*in_off[i], [J][in_dim]
  float* bias = dalloc_f(2L * nseg * J);
  float* Zc = dalloc_f((long)rc32 * nseg * J); float* Zi = dalloc_f((long)Mi * nseg * J);
  unsigned char* Xcq = (unsigned char*)dalloc_f((long)rc32 * D); unsigned char* Xiq = (unsigned char*)dalloc_f((long)Mi * D);
  unsigned char* Wq = (unsigned char*)dalloc_f(2L * J * D);
  unsigned short *Zch = dalloc_h((long)rc32 * nseg * J), *Zcl = dalloc_h((long)rc32 * nseg * J), *Zih = dalloc_h((long)Mi * nseg * J), *Zil = dalloc_h((long)Mi * nseg * J);
  float* H1c = dalloc_f((long)Mc * nseg * J); float* H1i = dalloc_f((long)Mi * nseg * J);
  float* dW = dalloc_f(2L * J * D); float* db = dalloc_f(2L * nseg * J);
  float* slab = dalloc_f(2L * G * P2::SLAB); float* dslab = dalloc_f(2L * G * 256);
  int* d_count; CK(hipMalloc(&d_count, sizeof(int))); CK(hipMemcpy(d_count, &valid, sizeof(int), hipMemcpyHostToDevice));

  auto fill = [&](float* p, long n, unsigned seed, int mode) { fill_kernel<<<2048, 256>>>(p, n, seed, mode); };
  auto split = [&](const float* s, unsigned short* h, unsigned short* l, long n) { split_kernel<<<2048, 256>>>(s, h, l, n / 4); };

  GemmGroup gf, gw;
  auto build = [&]() {
    memset(&gf, 0, sizeof(gf)); memset(&gw, 0, sizeof(gw));
    for (int h = 0; h < 2; ++h)
      for (int i = 0; i < nseg; ++i) {
        GemmProblem p;
        memset(&p, 0, sizeof(p));
        const unsigned char* xq = h == 0 ? Xcq : Xiq;
        p.A = (const float*)(xq + 4096L * (in_off[i] / 32)); p.lda = D;
        const long wo = (long)h * J * D + (long)J * in_off[i];
        p.B = (const float*)(Wq + 4 * wo); p.ldb = in_dim[i];
        p.bias = bias + (h * nseg + i) * J;
        p.C = (h == 0 ? H1c : H1i) + (long)i * J; p.ldc = (long)nseg * J;
        p.M = h == 0 ? rc32 : Mi; p.N = J; p.K = in_dim[i];
        p.dyn = h == 0 ? d_count : nullptr;
        p.drop_scale = 1.f;
        gf.p[gf.nprob++] = p;
        GemmProblem w;
        memset(&w, 0, sizeof(w));
        const unsigned short* zh = h == 0 ? Zch : Zih; const unsigned short* zl = h == 0 ? Zcl : Zil;
        w.A = (const float*)(zh + (long)i * J); w.A_lo = zl + (long)i * J; w.lda = (long)nseg * J;
        w.B = (const float*)(xq + 4096L * (in_off[i] / 32)); w.ldb = D;
        w.C = dW + wo; w.ldc = in_dim[i];
        w.M = J; w.N = in_dim[i]; w.K = h == 0 ? rc32 : Mi;
        w.dyn = h == 0 ? d_count : nullptr;
        w.dbias = db + (h * nseg + i) * J;
        w.drop_scale = 1.f;
        gw.p[gw.nprob++] = w;
      }
    gw.p[0].slab = slab; gw.p[0].dbias_slab = dslab;
  };
  build();
  const int nrep = J / 256;
  int ntiles = 0;
  for (int i = 0; i < gw.nprob; ++i) ntiles += (gw.p[i].N / 256) * nrep;

  auto run_fwd = [&]() {
    if (gf.ablate == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_nt_kernel<16>), dim3(G), dim3(512), 0, 0, gf, nrep);
    else if (gf.ablate == 2048) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_nt_kernel<2048>), dim3(G), dim3(512), 0, 0, gf, nrep);
    else if (gf.ablate == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_nt_kernel<128>), dim3(G), dim3(512), 0, 0, gf, nrep);
    else if (gf.ablate == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_nt_kernel<32>), dim3(G), dim3(512), 0, 0, gf, nrep);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_nt_kernel<0>), dim3(G), dim3(512), 0, 0, gf, nrep);
  };
  auto run_tn = [&]() {
    if (gw.ablate == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_kernel<16>), dim3(G), dim3(512), 0, 0, gw, nrep);
    else if (gw.ablate == 2048) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_kernel<2048>), dim3(G), dim3(512), 0, 0, gw, nrep);
    else if (gw.ablate == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_kernel<32>), dim3(G), dim3(512), 0, 0, gw, nrep);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_kernel<0>), dim3(G), dim3(512), 0, 0, gw, nrep);
  };
  auto run_bwd = [&]() {
    run_tn();
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_reduce_kernel<false>), dim3(ntiles * 64), dim3(256), 0, 0, gw, nrep, G / nrep, AdamFuse{});
  };

  double* refC; CK(hipMalloc(&refC, (long)rc32 * nseg * J * sizeof(double)));
  double* refW; CK(hipMalloc(&refW, (long)J * 2048 * sizeof(double)));
  std::vector<float> hv; std::vector<double> hr;

  for (int mode = (only >= 0 ? 1 : 0); mode < 2; ++mode) {
    // mode 0: exact integers; mode 1: random
    fill(Xc, (long)rc32 * D, 1u, mode == 0 ? 0 : 2); fill(Xi, (long)Mi * D, 2u, mode == 0 ? 0 : 2);
    fill(W, 2L * J * D, 3u, mode == 0 ? 0 : 3); fill(bias, 2L * nseg * J, 4u, mode == 0 ? 0 : 1);
    fill(Zc, (long)rc32 * nseg * J, 5u, mode == 0 ? 0 : 1); fill(Zi, (long)Mi * nseg * J, 6u, mode == 0 ? 0 : 1);
    // rows beyond the valid count are zero in the planes (as the library's staging leaves them)
    if (rc32 > rows_c) { CK(hipMemset(Xc + (long)rows_c * D, 0, (long)(rc32 - rows_c) * D * 4)); CK(hipMemset(Zc + (long)rows_c * nseg * J, 0, (long)(rc32 - rows_c) * nseg * J * 4)); }
    q32_kernel<<<2048, 256>>>(Xc, Xcq, rc32, D); q32_kernel<<<2048, 256>>>(Xi, Xiq, Mi, D);
    for (int h = 0; h < 2; ++h)
      for (int i = 0; i < nseg; ++i) {
        const long wo = (long)h * J * D + (long)J * in_off[i];
        q32_kernel<<<512, 256>>>(W + wo, Wq + 4 * wo, J, in_dim[i]);
      }
    split(Zc, Zch, Zcl, (long)rc32 * nseg * J); split(Zi, Zih, Zil, (long)Mi * nseg * J);
    CK(hipMemset(H1c, 0xff, (long)Mc * nseg * J * 4)); CK(hipMemset(H1i, 0xff, (long)Mi * nseg * J * 4));
    CK(hipMemset(dW, 0, 2L * J * D * 4)); CK(hipMemset(db, 0, 2L * nseg * J * 4));
    run_fwd(); run_bwd();
    CK(hipDeviceSynchronize());
    if (only >= 0) break;
    // ---- forward check
    double worst = 0.0; long bad = 0;
    for (int h = 0; h < 2; ++h) {
      const int rows = h == 0 ? rows_c : rows_i;
      const float* X = h == 0 ? Xc : Xi;
      for (int i = 0; i < nseg; ++i)
        ref_nt_kernel<<<dim3(J / 16, (rows + 15) / 16), 256>>>(X, D, in_off[i], W + (long)h * J * D + (long)J * in_off[i], in_dim[i],
                                                              bias + (h * nseg + i) * J, rows, J, refC, (long)nseg * J, i * J);
      hv.resize((long)rows * nseg * J); hr.resize((long)rows * nseg * J);
      CK(hipMemcpy(hv.data(), h == 0 ? H1c : H1i, hv.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hr.data(), refC, hr.size() * 8, hipMemcpyDeviceToHost));
      double scale = 0.0;
      for (size_t e = 0; e < hr.size(); ++e) scale = fmax(scale, fabs(hr[e]));
      for (size_t e = 0; e < hr.size(); ++e) {
        const double d = fabs((double)hv[e] - hr[e]);
        if (!(d <= (mode == 0 ? 0.0 : 1e-5 * scale + 1e-4 * fabs(hr[e])))) { if (bad < 5) printf("  fwd mismatch head %d elem %zu (row %zu col %zu): %g vs %g\n", h, e, e / (nseg * J), e % (nseg * J), hv[e], hr[e]); ++bad; }
        worst = fmax(worst, d / (scale > 0 ? scale : 1));
      }
    }
    printf("mode %d forward: %ld mismatches, worst |d|/scale %.3g\n", mode, bad, worst);
    // ---- weight-gradient check
    worst = 0.0; bad = 0;
    for (int h = 0; h < 2; ++h) {
      const int rows = h == 0 ? rows_c : rows_i;
      for (int i = 0; i < nseg; ++i) {
        ref_tn_kernel<<<dim3(in_dim[i] / 16, J / 16), 256>>>(h == 0 ? Zc : Zi, (long)nseg * J, i * J, h == 0 ? Xc : Xi, D, in_off[i], rows, J, in_dim[i], refW);
        hv.resize((long)J * in_dim[i]); hr.resize((long)J * in_dim[i]);
        CK(hipMemcpy(hv.data(), dW + (long)h * J * D + (long)J * in_off[i], hv.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), refW, hr.size() * 8, hipMemcpyDeviceToHost));
        double scale = 0.0;
        for (size_t e = 0; e < hr.size(); ++e) scale = fmax(scale, fabs(hr[e]));
        for (size_t e = 0; e < hr.size(); ++e) {
          const double d = fabs((double)hv[e] - hr[e]);
          if (!(d <= (mode == 0 ? 0.0 : 2e-5 * scale + 1e-4 * fabs(hr[e])))) { if (bad < 5) printf("  dW mismatch head %d seg %d elem %zu (m %zu n %zu): %g vs %g\n", h, i, e, e / in_dim[i], e % in_dim[i], hv[e], hr[e]); ++bad; }
          worst = fmax(worst, d / (scale > 0 ? scale : 1));
        }
      }
    }
    printf("mode %d dW: %ld mismatches, worst |d|/scale %.3g\n", mode, bad, worst);
    // bias gradient: column sums of Z
    {
      std::vector<float> hdb(2L * nseg * J), hz;
      CK(hipMemcpy(hdb.data(), db, hdb.size() * 4, hipMemcpyDeviceToHost));
      long badb = 0; double wb = 0.0;
      for (int h = 0; h < 2; ++h) {
        const int rows = h == 0 ? rows_c : rows_i;
        hz.resize((long)rows * nseg * J);
        CK(hipMemcpy(hz.data(), h == 0 ? Zc : Zi, hz.size() * 4, hipMemcpyDeviceToHost));
        for (int c = 0; c < nseg * J; ++c) {
          double s = 0.0, sa = 0.0;
          for (int r = 0; r < rows; ++r) { s += hz[(long)r * nseg * J + c]; sa += fabs(hz[(long)r * nseg * J + c]); }
          const double d = fabs(hdb[h * nseg * J + c] - s);
          if (!(d <= (mode == 0 ? 0.0 : 1e-5 * sa + 1e-6))) { if (badb < 5) printf("  db mismatch head %d col %d: %g vs %g\n", h, c, hdb[h * nseg * J + c], s); ++badb; }
          wb = fmax(wb, d / (sa > 0 ? sa : 1));
        }
      }
      printf("mode %d db: %ld mismatches, worst %.3g\n", mode, badb, wb);
    }
  }

  // ---- one-plane variants (rows STORED as bf16, q16b, gathered): on exact-integer operands their results must equal the
  // two-plane kernels' bit for bit (the lo halves of such rows are zero); then their times
  {
    fill(Xc, (long)rc32 * D, 1u, 0); fill(Xi, (long)Mi * D, 2u, 0);
    fill(W, 2L * J * D, 3u, 0); fill(bias, 2L * nseg * J, 4u, 0);
    fill(Zc, (long)rc32 * nseg * J, 5u, 0); fill(Zi, (long)Mi * nseg * J, 6u, 0);
    if (rc32 > rows_c) { CK(hipMemset(Xc + (long)rows_c * D, 0, (long)(rc32 - rows_c) * D * 4)); CK(hipMemset(Zc + (long)rows_c * nseg * J, 0, (long)(rc32 - rows_c) * nseg * J * 4)); }
    q32_kernel<<<2048, 256>>>(Xc, Xcq, rc32, D); q32_kernel<<<2048, 256>>>(Xi, Xiq, Mi, D);
    unsigned char* Xc16 = (unsigned char*)dalloc_h((long)rc32 * D); unsigned char* Xi16 = (unsigned char*)dalloc_h((long)Mi * D);
    q16_kernel<<<2048, 256>>>(Xc, Xc16, rc32, D); q16_kernel<<<2048, 256>>>(Xi, Xi16, Mi, D);
    for (int h = 0; h < 2; ++h)
      for (int i = 0; i < nseg; ++i) {
        const long wo = (long)h * J * D + (long)J * in_off[i];
        q32_kernel<<<512, 256>>>(W + wo, Wq + 4 * wo, J, in_dim[i]);
      }
    split(Zc, Zch, Zcl, (long)rc32 * nseg * J); split(Zi, Zih, Zil, (long)Mi * nseg * J);
    int* ident; CK(hipMalloc(&ident, (long)Mc * 4));
    iota_kernel<<<64, 256>>>(ident, Mc);
    float* H1c2 = dalloc_f((long)Mc * nseg * J); float* H1i2 = dalloc_f((long)Mi * nseg * J);
    float* dW2 = dalloc_f(2L * J * D); float* db2 = dalloc_f(2L * nseg * J);
    CK(hipMemset(H1c, 0xff, (long)Mc * nseg * J * 4)); CK(hipMemset(H1i, 0xff, (long)Mi * nseg * J * 4));
    CK(hipMemset(H1c2, 0xff, (long)Mc * nseg * J * 4)); CK(hipMemset(H1i2, 0xff, (long)Mi * nseg * J * 4));
    CK(hipMemset(dW, 0, 2L * J * D * 4)); CK(hipMemset(db, 0, 2L * nseg * J * 4)); CK(hipMemset(dW2, 0, 2L * J * D * 4)); CK(hipMemset(db2, 0, 2L * nseg * J * 4));
    gf.ablate = gw.ablate = 0;
    run_fwd(); run_bwd();
    GemmGroup gf1 = gf, gw1 = gw;
    for (int k = 0; k < gf1.nprob; ++k) {
      const int h = k / nseg, i = k % nseg;
      gf1.p[k].A = (const float*)((h == 0 ? Xc16 : Xi16) + 2048L * (in_off[i] / 32)); gf1.p[k].srow = ident;
      gf1.p[k].C = (h == 0 ? H1c2 : H1i2) + (long)i * J;
      gw1.p[k].B = (const float*)((h == 0 ? Xc16 : Xi16) + 2048L * (in_off[i] / 32)); gw1.p[k].srow = ident;
      gw1.p[k].C = dW2 + ((long)h * J * D + (long)J * in_off[i]); gw1.p[k].dbias = db2 + (h * nseg + i) * J;
    }
    auto fwd1 = [&]() { hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_ntg1_kernel<0>), dim3(G), dim3(512), 0, 0, gf1, nrep); };
    auto bwd1 = [&]() {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_kernel<0, true, 1>), dim3(G), dim3(512), 0, 0, gw1, nrep);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_reduce_kernel<false>), dim3(ntiles * 64), dim3(256), 0, 0, gw1, nrep, G / nrep, AdamFuse{});
    };
    fwd1(); bwd1();
    CK(hipDeviceSynchronize());
    auto differ = [&](const float* a, const float* b, long n) {
      std::vector<float> x(n), y(n);
      CK(hipMemcpy(x.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), b, n * 4, hipMemcpyDeviceToHost));
      long d = 0;
      for (long e = 0; e < n; ++e) d += memcmp(&x[e], &y[e], 4) != 0;
      return d;
    };
    printf("one-plane (q16b rows) vs two-plane kernels on exact integers: forward %ld + %ld elements differ, dW %ld, db %ld\n",
           differ(H1c, H1c2, (long)rows_c * nseg * J), differ(H1i, H1i2, (long)Mi * nseg * J), differ(dW, dW2, 2L * J * D), differ(db, db2, 2L * nseg * J));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> tf, tb;
    for (int r = 0; r < 7; ++r) {
      float ms;
      CK(hipEventRecord(e0)); for (int it = 0; it < 4; ++it) fwd1(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1)); tf.push_back(ms / 4);
      CK(hipEventRecord(e0)); for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_kernel<0, true, 1>), dim3(G), dim3(512), 0, 0, gw1, nrep);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1)); tb.push_back(ms / 4);
    }
    std::sort(tf.begin(), tf.end()); std::sort(tb.begin(), tb.end());
    printf("one-plane     forward: median %.1f us (min %.1f)   dW1 gemm: median %.1f us (min %.1f)   (two MFMAs per product, half the row bytes)\n",
           1e3 * tf[3], 1e3 * tf[0], 1e3 * tb[3], 1e3 * tb[0]);
    // back to random operands for the timings below
    fill(Xc, (long)rc32 * D, 1u, 2); fill(Xi, (long)Mi * D, 2u, 2); fill(W, 2L * J * D, 3u, 3); fill(bias, 2L * nseg * J, 4u, 1);
    fill(Zc, (long)rc32 * nseg * J, 5u, 1); fill(Zi, (long)Mi * nseg * J, 6u, 1);
    if (rc32 > rows_c) { CK(hipMemset(Xc + (long)rows_c * D, 0, (long)(rc32 - rows_c) * D * 4)); CK(hipMemset(Zc + (long)rows_c * nseg * J, 0, (long)(rc32 - rows_c) * nseg * J * 4)); }
    q32_kernel<<<2048, 256>>>(Xc, Xcq, rc32, D); q32_kernel<<<2048, 256>>>(Xi, Xiq, Mi, D);
    for (int h = 0; h < 2; ++h)
      for (int i = 0; i < nseg; ++i) {
        const long wo = (long)h * J * D + (long)J * in_off[i];
        q32_kernel<<<512, 256>>>(W + wo, Wq + 4 * wo, J, in_dim[i]);
      }
    split(Zc, Zch, Zcl, (long)rc32 * nseg * J); split(Zi, Zih, Zil, (long)Mi * nseg * J);
    CK(hipDeviceSynchronize());
  }

  // ---- timing (random operands): variants interleaved in rounds (the chip's clock drifts with load; back-to-back blocks of
  // one variant each rank the variants by their position in the run)
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flops = 2.0 * ((double)rows_c + rows_i) * D * J;
  const int abl[6] = {0, 4, 16, 32, 2048, 128};
  const char* abn[6] = {"full", "no-k-loop", "no-dma", "dma-only", "full-no-nt", "full-hot"};
  {
    const int NR = 7;
    std::vector<float> tf[6], tb[6];
    for (int w = 0; w < 3; ++w) { gf.ablate = gw.ablate = 0; run_fwd(); run_bwd(); }
    for (int r = 0; r < NR; ++r)
      for (int abi = 0; abi < 6; ++abi) {
        if (only >= 0 && abi != only) continue;
        gf.ablate = gw.ablate = abl[abi];
        float ms;
        CK(hipEventRecord(e0)); for (int it = 0; it < 4; ++it) run_fwd(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); tf[abi].push_back(ms / 4);
        if (abi == 5) continue;
        CK(hipEventRecord(e0)); for (int it = 0; it < 4; ++it) run_tn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); tb[abi].push_back(ms / 4);
      }
    for (int abi = 0; abi < 6; ++abi) {
      if (tf[abi].empty()) continue;
      std::sort(tf[abi].begin(), tf[abi].end());
      printf("%-13s forward: median %.1f us (min %.1f)  %.0f TF algorithmic", abn[abi], 1e3 * tf[abi][tf[abi].size() / 2], 1e3 * tf[abi][0], flops / (tf[abi][tf[abi].size() / 2] * 1e-3) / 1e12);
      if (!tb[abi].empty()) { std::sort(tb[abi].begin(), tb[abi].end()); printf("   dW1 gemm: median %.1f us (min %.1f)", 1e3 * tb[abi][tb[abi].size() / 2], 1e3 * tb[abi][0]); }
      printf("\n");
    }
    gf.ablate = gw.ablate = 0;
    float ms;
    CK(hipEventRecord(e0)); for (int it = 0; it < iters; ++it) run_bwd(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("dW1 gemm + reduce: %.1f us\n", 1e3 * ms / iters);
  }
  // ---- per-step cycle stamps of the forward kernel (diagnostics build, ABL 1024): every wave of a few workgroups
  {
    long long* st; const long nst = (long)G * 8 * 512;
    CK(hipMalloc(&st, nst * sizeof(long long))); CK(hipMemset(st, 0, nst * sizeof(long long)));
    for (int i = 0; i < gf.nprob; ++i) gf.p[i].slab = reinterpret_cast<float*>(st);
    gf.ablate = 0;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_nt_kernel<1024>), dim3(G), dim3(512), 0, 0, gf, nrep);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(nst);
    CK(hipMemcpy(h.data(), st, nst * sizeof(long long), hipMemcpyDeviceToHost));
    const int blks[3] = {64, 65, 200};
    for (int bi = 0; bi < 3; ++bi)
      for (int w = 0; w < 8; w += 4) {
        const long long* s5 = h.data() + (long)(blks[bi] * 8 + w) * 512;
        printf("stamps block %d wave %d: step: barrier(even) even-half-work barrier(odd) odd-half-work | total (cycles)\n", blks[bi], w);
        for (int t = 0; t < 64 && s5[6 * t]; ++t)
          printf("   %2d: %6lld %6lld %6lld %6lld | %6lld\n", t, s5[6 * t + 1] - s5[6 * t], s5[6 * t + 2] - s5[6 * t + 1], s5[6 * t + 3] - s5[6 * t + 2],
                 s5[6 * t + 4] - s5[6 * t + 3], t > 0 ? s5[6 * t] - s5[6 * (t - 1)] : 0LL);
      }
    printf("tile phases (cycles; wave 0 / wave 4): entry -> first step landed -> k loop done -> epilogue done\n");
    for (int b = 0; b < G; b += 17) {
      const long long* s0 = h.data() + (long)(b * 8 + 0) * 512; const long long* s4 = h.data() + (long)(b * 8 + 4) * 512;
      if (!s0[500]) continue;
      printf("   block %3d: %6lld %7lld %6lld | %6lld %7lld %6lld\n", b, s0[501] - s0[500], s0[502] - s0[501], s0[503] - s0[502],
             s4[501] - s4[500], s4[502] - s4[501], s4[503] - s4[502]);
    }
    for (int i = 0; i < gf.nprob; ++i) gf.p[i].slab = nullptr;
  }
  // ---- per-workgroup time stamps (full kernels) ----------------------------------------------------------------------
  {
    long long* dbg; CK(hipMalloc(&dbg, 4L * G * sizeof(long long)));
    std::vector<long long> h(4L * G);
    for (int which = 0; which < 2; ++which) {
      GemmGroup& gg = which == 0 ? gf : gw;
      gg.ablate = 64; if (which == 0) gg.p[0].slab = reinterpret_cast<float*>(dbg); else gg.p[0].aux_out = reinterpret_cast<float*>(dbg);
      CK(hipMemset(dbg, 0, 4L * G * sizeof(long long)));
      hipEvent_t ev[22];
      for (int it = 0; it < 22; ++it) CK(hipEventCreate(&ev[it]));
      for (int it = 0; it < 21; ++it) {
        CK(hipEventRecord(ev[it]));
        if (which == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_nt_kernel<0>), dim3(G), dim3(512), 0, 0, gf, nrep); else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_p2_tn_kernel<0>), dim3(G), dim3(512), 0, 0, gw, nrep);
      }
      CK(hipEventRecord(ev[21]));
      CK(hipDeviceSynchronize());
      printf("%s per-launch (us):", which == 0 ? "forward" : "dW1 gemm");
      for (int it = 0; it < 21; ++it) { float ms; CK(hipEventElapsedTime(&ms, ev[it], ev[it + 1])); printf(" %.0f", ms * 1e3); }
      printf("\n");
      CK(hipMemcpy(h.data(), dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
      long long t0 = h[0];
      for (int L = 0; L < G; ++L) if (h[4 * L] && h[4 * L] < t0) t0 = h[4 * L];
      { long long tb = 0, te = 0; for (int L = 0; L < G; ++L) { if (h[4 * L] > tb) tb = h[4 * L]; if (h[4 * L + 1] > te) te = h[4 * L + 1]; }
        printf("%s last launch: first begin -> last begin %.1f us, -> last end %.1f us\n", which == 0 ? "forward" : "dW1", (tb - t0) * 0.01, (te - t0) * 0.01); }
      printf("%s: logical id: begin / duration (us), xcc, block\n", which == 0 ? "forward" : "dW1");
      for (int L = 0; L < G; ++L)
        printf("  %3d: %6.1f %6.1f  xcc %lld blk %lld\n", L, (h[4 * L] - t0) * 0.01, (h[4 * L + 1] - h[4 * L]) * 0.01, h[4 * L + 2], h[4 * L + 3]);
    }
  }
  return 0;
}
