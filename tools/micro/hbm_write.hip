// What does the chip sustain for WRITES?  Every write-heavy kernel of the step (row staging: 257 MB written in 101 us; Adam: 221 MB
// in 85 us; un-pool: 66 MB in 26 us) sits at ~2.5 TB/s of writes whatever its read side does.  This tool measures plain streaming
// kernels over a 2-GiB buffer: write only (plain / non-temporal stores; 16 B and 8 B per lane), read only, copy (read + write), and
// "adam-like" (4 streams read, 3 written).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/hbm_write.hip -o tools/micro/hbm_write.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k_write(f32x4* __restrict__ dst, const f32x4* __restrict__ src, long n4, float v) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    if constexpr (MODE == 0) dst[i] = f32x4{v, v, v, v};                                             // plain 16-B stores
    else if constexpr (MODE == 1) __builtin_nontemporal_store(f32x4{v, v, v, v}, dst + i);          // non-temporal
    else if constexpr (MODE == 2) { reinterpret_cast<f32x2*>(dst)[2 * i] = f32x2{v, v}; reinterpret_cast<f32x2*>(dst)[2 * i + 1] = f32x2{v, v}; }
    else if constexpr (MODE == 3) { const f32x4 x = src[i]; if (x[0] == 12345.f) dst[0] = x; }       // read only
    else if constexpr (MODE == 4) dst[i] = src[i];                                                   // copy
    else if constexpr (MODE == 5) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);      // copy, nt both
    else if constexpr (MODE == 6) {                                                                  // adam-like: 4 read, 3 written (quarter-size streams)
      const long q = n4 >> 2;
      if (i < q) {
        const f32x4 p = dst[i], g = src[i], m = dst[i + q], w = dst[i + 2 * q];
        dst[i] = p + g; dst[i + q] = m + g; dst[i + 2 * q] = w + g * g;
      }
    }
  }
}

template <int MODE>
static double run(f32x4* dst, const f32x4* src, long n4, int blocks, double bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_write<MODE>), dim3(blocks), dim3(256), 0, 0, dst, src, n4, 1.5f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return bytes / (t[1] * 1e-3) / 1e12;
}

int main() {
  const long bytes = 2L << 30, n4 = bytes / 16;
  f32x4 *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  printf("2 GiB per stream; TB/s of the bytes NAMED (copy: read + written); second fastest of 5\n");
  for (int blocks : {1024, 4096, 16384}) {
    printf("grid %5d x 256:  write %.2f   write nt %.2f   write 8 B %.2f   read %.2f   copy %.2f (x2 = %.2f moved)   copy nt %.2f (%.2f)   adam-like %.2f moved (writes %.2f)\n", blocks,
           run<0>(a, b, n4, blocks, (double)bytes), run<1>(a, b, n4, blocks, (double)bytes), run<2>(a, b, n4, blocks, (double)bytes),
           run<3>(a, b, n4, blocks, (double)bytes), run<4>(a, b, n4, blocks, (double)bytes), 2 * run<4>(a, b, n4, blocks, (double)bytes),
           run<5>(a, b, n4, blocks, (double)bytes), 2 * run<5>(a, b, n4, blocks, (double)bytes),
           run<6>(a, b, n4, blocks, (double)bytes * 7 / 4), run<6>(a, b, n4, blocks, (double)bytes * 3 / 4));
  }
  float ms;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0)); CK(hipMemsetAsync(a, 1, bytes, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("hipMemsetAsync: %.2f TB/s\n", bytes / (ms * 1e-3) / 1e12);
  CK(hipEventRecord(e0, 0)); CK(hipMemcpyAsync(a, b, bytes, hipMemcpyDeviceToDevice, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("hipMemcpyAsync D2D: %.2f TB/s copied (%.2f moved)\n", bytes / (ms * 1e-3) / 1e12, 2 * bytes / (ms * 1e-3) / 1e12);
  return 0;
}
