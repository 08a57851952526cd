// How much does a cross-stream hand-over cost the signalling stream?  Chain of kernels A,B,A,B,... on `main`; after every A
// the side stream is made to wait for it and runs a small kernel C.  Variants: (0) no hand-over, (1) hipEventRecord after A +
// hipStreamWaitEvent, (2) A launched with hipExtLaunchKernelGGL(stopEvent) + hipStreamWaitEvent.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/event_bubble.hip -o tools/micro/event_bubble.bin
// [measured] per hand-over on the signalling stream: event record +6.0 us, kernel launched with a stop event +3.5 us, nothing 0.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
int main() {
  float *a, *b, *c;
  hipMalloc(&a, 4096); hipMalloc(&b, 4096); hipMalloc(&c, 4096);
  hipStream_t mainS, side;
  hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking); hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
  const int N = 200, IT = 20000;   // ~20 us kernels
  hipEvent_t ev[N];
  for (int i = 0; i < N; ++i) hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
  for (int variant = 0; variant < 3; ++variant) {
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (variant == 2) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, mainS, nullptr, ev[i], 0, a, IT);
        else hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, mainS, a, IT);
        if (variant == 1) hipEventRecord(ev[i], mainS);
        if (variant != 0) { hipStreamWaitEvent(side, ev[i], 0); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, side, c, IT / 4); }
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, mainS, b, IT);
      }
      hipStreamSynchronize(mainS);
      auto t1 = std::chrono::steady_clock::now();
      hipDeviceSynchronize();
      printf("variant %d: %.2f us per A+B pair on the main stream\n", variant, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
  }
  return 0;
}
