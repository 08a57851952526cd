// Command lists: a recorded train step re-issued from C.
//
// A step of the hot path is ~25 kernel launches whose arguments do not change from step to step once the activations
// sit at fixed addresses and the per-step scalars (dropout key offset, Adam step) live in device memory.  Driving them
// from Python costs ~0.9 ms of host time per step; a captured hipGraph of the same step replays with 10-20 us gaps at
// its fork / join points and costs the host as much again (measured: 1.13 ms per step against 1.05 ms for the eager
// loop).  So the library can RECORD the launches it makes -- kernel, grid, arguments by value, stream -- while an
// ordinary eager step runs, and re-issue the list later: plain launches on the caller's real streams (what the eager
// loop does, hence its timeline), with the host work of one C loop.
//
// Every launch in the library goes through lirec::launch(); while the calling thread records, a closure with the same
// arguments is appended to the list.  Cross-stream fork / join (lirec_stream_wait) and memsets are recorded the same way.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <vector>

namespace lirec {

struct CmdList {
  std::vector<std::function<void()>> cmds;
  // the stream each command is issued on (a stream wait: the SIGNALLING stream) and what it is -- 0 kernel launch / memset,
  // 1 stream wait, 2 a profiling bracket: what lirec_cmdlist_replay_lagged (the dependency fuzzer of the tests) goes by
  std::vector<hipStream_t> streams;
  std::vector<unsigned char> kinds;
  std::vector<hipEvent_t> events;          // owned: one per recorded stream wait
  void push(std::function<void()> f, hipStream_t s, int kind) {
    cmds.emplace_back(std::move(f)); streams.push_back(s); kinds.push_back((unsigned char)kind);
  }
  ~CmdList() { for (hipEvent_t e : events) (void)hipEventDestroy(e); }
};
extern thread_local CmdList* t_rec;        // non-null while this thread records (lirec_record_begin)
// Host-side dry run (lirec_debug_set bit 4194304; tests/test_host_asan.py): nothing is handed to the HIP runtime -- launches,
// memsets, event operations are skipped (and still recorded) -- so that the library's HOST code (argument validation, partition
// planning, the command lists' argument copies) can run under AddressSanitizer / UBSan in a container without a GPU.  Computes
// nothing; never set by the product.
extern bool g_dry;

template <class K, class... A>
inline void launch(K kernel, dim3 grid, dim3 block, unsigned shmem, hipStream_t s, A... args) {
  if (!g_dry) hipLaunchKernelGGL(kernel, grid, block, shmem, s, args...);
  if (t_rec) t_rec->push([=]() { if (!g_dry) hipLaunchKernelGGL(kernel, grid, block, shmem, s, args...); }, s, 0);
}

inline hipError_t memset_async(void* p, int v, size_t bytes, hipStream_t s) {
  const hipError_t e = g_dry ? hipSuccess : hipMemsetAsync(p, v, bytes, s);
  if (t_rec) t_rec->push([=]() { if (!g_dry) (void)hipMemsetAsync(p, v, bytes, s); }, s, 0);
  return e;
}

}  // namespace lirec
