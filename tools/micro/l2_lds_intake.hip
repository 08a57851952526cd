// How fast can ONE CU pull operand bytes into LDS when the bytes are cache-resident?  (DESIGN 4.4 priced the three 3072^2 gate
// GEMMs with "~35 GB/s per CU L2 -> LDS", a figure read off inside gemm_p2 with an HBM-sourced operand in the mix; this tool
// isolates it.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/l2_lds_intake.hip -o tools/micro/l2_lds_intake.bin
//   tools/micro/l2_lds_intake.bin            -> the table (stdout)
// One workgroup per CU (grid = CU count), W issuing waves (1, 2, 4, 8), every wave-instruction moves 1 KiB (dwordx4 per lane):
//   method dma : global_load_lds_dwordx4 (LDS-DMA) into a 64-KiB ring, D requests per wave in flight (counted vmcnt)
//   method reg : global_load_dwordx4 to registers, then ds_write_b128
// source:
//   shared  : every workgroup of the launch streams the SAME 2-MiB panel, again and again (L2 hits after the first pass; a weight
//             panel shared by the row tiles of a GEMM)
//   private : workgroup w streams its own 96-KiB panel again and again (32 x 96 KiB = 3 MiB per XCD: L2-resident, no sharing)
//   mall    : workgroup w walks its own 768-KiB slice of a 192-MiB buffer (beyond L2, inside the 256-MiB Infinity Cache)
//   hbm     : workgroup w walks its own slice of a 2-GiB buffer once (no reuse at all)
// mfma = m : every wave issues m v_mfma_f32_16x16x32_bf16 on register operands behind each 1-KiB request (0 = none; 3 = the
//            split-precision k-loop's ratio for a 128 x 128 tile: 12 MFMA 32x32x16-equivalents per four requests)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* sbase_, unsigned voff, unsigned lds_) {
  const unsigned long sb = (unsigned long)sbase_;
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(sb >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb);
  const void* sbase = (const void*)(((unsigned long)hi << 32) | (unsigned long)lo);
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}

// METHOD 0 = dma, 1 = reg.  Each wave owns an 8-KiB window of the ring (8 slots of 1 KiB) and keeps 8 requests in flight.
template <int METHOD, int MF>
__global__ __launch_bounds__(512) void intake_kernel(const unsigned char* __restrict__ src, long wg_stride, long panel, int iters,
                                                     float* __restrict__ sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[65536];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const unsigned char* base = src + (long)blockIdx.x * wg_stride;
  const unsigned lds0 = (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)ring + 8192u * wave;
  f32x4v acc[4];
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
  for (int n = 0; n < 4; ++n) acc[n] = f32x4v{0.f, 0.f, 0.f, 0.f};
  float keep = 0.f;
  // the wave's stream: 1-KiB pieces wave, wave + nw, ... of the panel, wrapping
  long off = (long)wave * 1024;
  const long step = (long)nw * 1024;
  for (int it = 0; it < iters; ++it) {
    if constexpr (METHOD == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        dma16(base + off, 16u * lane, lds0 + 1024u * u);
        off += step; if (off >= panel) off -= panel;
#pragma unroll
        for (int m = 0; m < MF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
      }
      // all but the newest eight have landed: two batches in flight at the steady state
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = *reinterpret_cast<const uint4*>(base + off + 16 * lane);
        off += step; if (off >= panel) off -= panel;
#pragma unroll
        for (int m = 0; m < MF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) *reinterpret_cast<uint4*>(ring + 8192 * wave + 1024 * u + 16 * lane) = v[u];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  keep = reinterpret_cast<const float*>(ring)[threadIdx.x];
  for (int n = 0; n < 4; ++n) keep += acc[n][0];
  if (keep == 123456.f) sink[0] = keep;
}

template <int METHOD, int MF>
static double run(const unsigned char* src, long wg_stride, long panel, int waves, int cus, long bytes_per_wg, float* sink) {
  const int iters = (int)(bytes_per_wg / ((long)waves * 8192));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(intake_kernel<METHOD, MF>), dim3(cus), dim3(64 * waves), 0, 0, src, wg_stride, panel, iters, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)iters * waves * 8192.0;
  return bytes / (best * 1e-3) / 1e9;           // GB/s per workgroup (= per CU)
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const long big = 2L << 30;
  unsigned char* buf;
  float* sink;
  CK(hipMalloc(&buf, big));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(buf, 1, big));
  printf("# %s, %d CUs; GB/s per CU (aggregate TB/s); every request 1 KiB per wave-instruction, 8-16 in flight per wave\n", prop.name, cus);
  printf("%-8s %-6s %-5s %5s %12s %10s\n", "source", "method", "waves", "mfma", "GB/s per CU", "TB/s chip");
  struct Src { const char* name; long stride, panel, bytes; } srcs[] = {
      {"shared", 0, 2L << 20, 24L << 20}, {"private", 96L << 10, 96L << 10, 24L << 20},
      {"mall", 768L << 10, 768L << 10, 12L << 20}, {"hbm", (big / cus) & ~1023L, (big / cus) & ~1023L, (big / cus) & ~1023L}};
  for (const Src& s : srcs) {
    for (int method = 0; method < 2; ++method) {
      for (int waves = 1; waves <= 8; waves *= 2) {
        for (int mf = 0; mf <= 3; mf += 3) {
          double r;
          if (method == 0) r = mf ? run<0, 3>(buf, s.stride, s.panel, waves, cus, s.bytes, sink) : run<0, 0>(buf, s.stride, s.panel, waves, cus, s.bytes, sink);
          else r = mf ? run<1, 3>(buf, s.stride, s.panel, waves, cus, s.bytes, sink) : run<1, 0>(buf, s.stride, s.panel, waves, cus, s.bytes, sink);
          printf("%-8s %-6s %-5d %5d %12.1f %10.2f\n", s.name, method == 0 ? "dma" : "reg", waves, mf, r, r * cus / 1e3);
        }
      }
    }
  }
  return 0;
}
