// One translation unit per GEMM kernel family: compiled with
//   -DLIREC_INST_LAYOUT={0 NT,1 NN,2 TN} -DLIREC_INST_CORE={0 exact-f32 + naive, 1 bf16x3} [-DLIREC_INST_CFG={0..3}]
// (lirec_amd/build.py).  Holds the kernel instantiations and the plain launch function for them.
#include <hip/hip_runtime.h>

#include "gemm.hpp"
#include "gemm_bf16x3.hpp"
#include "gemm_p2.hpp"
#include "gemm_p3.hpp"
#include "gemm_launch.hpp"

#ifndef LIREC_INST_LAYOUT
#error "compile with -DLIREC_INST_LAYOUT=.. -DLIREC_INST_CORE=.."
#endif

#define LIREC_CAT_(a, b) a##b
#define LIREC_CAT(a, b) LIREC_CAT_(a, b)

namespace lirec {

constexpr int kL = LIREC_INST_LAYOUT;
// the tagged symbol exists only where a heavy call site uses it: tag 1 with NT, tag 2 with TN
constexpr int kTag = (kL == L_NT) ? 1 : (kL == L_TN ? 2 : 0);

#if LIREC_INST_CORE == 0

template <int WM, int WN>
static void go_f32(int variant, dim3 grid, hipStream_t s, const GemmGroup& g) {
  if (variant >= GV_TAGGED && variant <= GV_MAPPED && kTag != 0)
    lirec::launch(HIP_KERNEL_NAME(gemm_mfma_kernel<kL, WM, WN, kTag, true>), grid, dim3(256), 0, s, g);
  else if (variant != GV_SCALAR)
    lirec::launch(HIP_KERNEL_NAME(gemm_mfma_kernel<kL, WM, WN, 0, true>), grid, dim3(256), 0, s, g);
  else
    lirec::launch(HIP_KERNEL_NAME(gemm_mfma_kernel<kL, WM, WN, 0, false>), grid, dim3(256), 0, s, g);
}

void LIREC_CAT(launch_f32_L, LIREC_INST_LAYOUT)(bool big, int variant, dim3 grid, hipStream_t s, const GemmGroup& g) {
  if (big) go_f32<2, 2>(variant, grid, s, g);
  else go_f32<1, 1>(variant, grid, s, g);
}

void LIREC_CAT(launch_naive_L, LIREC_INST_LAYOUT)(dim3 grid, hipStream_t s, const GemmProblem& p) {
  lirec::launch(HIP_KERNEL_NAME(gemm_naive_kernel<kL>), grid, dim3(256), 0, s, p);
}

#elif LIREC_INST_CORE == 2

// layer 1 on q32b operands, LDS-DMA rings, persistent launch (gemm_p2.hpp): NT (forward) and TN (weight gradient + its
// slab reduce); `grid` = workgroups of the persistent launch (one per CU), nrep = 256-wide replicas of a range
#ifndef LIREC_INST_PART
#define LIREC_INST_PART 0
#endif
#if LIREC_INST_LAYOUT == 0 && LIREC_INST_PART == 0
void launch_p2_nt(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep) {
  lirec::launch(HIP_KERNEL_NAME(gemm_p2_nt_kernel<0>), grid, dim3(512), 0, s, g, nrep);
}
void launch_p3_fwd(int ni, dim3 grid, hipStream_t s, const GemmGroup& g) {
  if (ni == 4) {
    if (g.onepass) lirec::launch(HIP_KERNEL_NAME(gemm_p3_kernel<4, 4, 0, true>), grid, dim3(512), 0, s, g);
    else lirec::launch(HIP_KERNEL_NAME(gemm_p3_kernel<4, 4, 0, false>), grid, dim3(512), 0, s, g);
  } else {
    if (g.onepass) lirec::launch(HIP_KERNEL_NAME(gemm_p3_kernel<4, 3, 0, true>), grid, dim3(512), 0, s, g);
    else lirec::launch(HIP_KERNEL_NAME(gemm_p3_kernel<4, 3, 0, false>), grid, dim3(512), 0, s, g);
  }
}
#elif LIREC_INST_LAYOUT == 0 && LIREC_INST_PART == 1
void launch_p2_ntg(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep) {
  lirec::launch(HIP_KERNEL_NAME(gemm_p2_ntg_kernel<0>), grid, dim3(512), 0, s, g, nrep);
}
#elif LIREC_INST_LAYOUT == 0 && LIREC_INST_PART == 2
void launch_p2_ntg1(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep) {
  lirec::launch(HIP_KERNEL_NAME(gemm_p2_ntg1_kernel<0>), grid, dim3(512), 0, s, g, nrep);
}
#elif LIREC_INST_LAYOUT == 0 && LIREC_INST_PART == 3
void launch_p2_ntg64(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep) {
  lirec::launch(HIP_KERNEL_NAME(gemm_p2_ntg64_kernel<0>), grid, dim3(512), 0, s, g, nrep);
}
#elif LIREC_INST_LAYOUT == 1
void launch_p2_nn(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep) {
  lirec::launch(HIP_KERNEL_NAME(gemm_p2_nn_kernel<0>), grid, dim3(512), 0, s, g, nrep);
}
void launch_p3_dgrad(int ni, dim3 grid, hipStream_t s, const GemmGroup& g) {
  if (ni == 4) {
    if (g.onepass) lirec::launch(HIP_KERNEL_NAME(gemm_p3_kernel<4, 4, 1, true>), grid, dim3(512), 0, s, g);
    else lirec::launch(HIP_KERNEL_NAME(gemm_p3_kernel<4, 4, 1, false>), grid, dim3(512), 0, s, g);
  } else {
    if (g.onepass) lirec::launch(HIP_KERNEL_NAME(gemm_p3_kernel<4, 3, 1, true>), grid, dim3(512), 0, s, g);
    else lirec::launch(HIP_KERNEL_NAME(gemm_p3_kernel<4, 3, 1, false>), grid, dim3(512), 0, s, g);
  }
}
#elif LIREC_INST_LAYOUT == 2 && LIREC_INST_PART == 2
void launch_p2_tng1(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep) {
  lirec::launch(HIP_KERNEL_NAME(gemm_p2_tn_kernel<0, true, 1>), grid, dim3(512), 0, s, g, nrep);
}
void launch_p2_tng1o(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep) {
  lirec::launch(HIP_KERNEL_NAME(gemm_p2_tn_kernel<0, true, 1, true>), grid, dim3(512), 0, s, g, nrep);
}
#else
void launch_p3_wgrad(dim3 grid, hipStream_t s, const GemmGroup& g) {
  if (g.onepass) lirec::launch(HIP_KERNEL_NAME(gemm_p3_kernel<4, 3, 2, true>), grid, dim3(512), 0, s, g);
  else lirec::launch(HIP_KERNEL_NAME(gemm_p3_kernel<4, 3, 2, false>), grid, dim3(512), 0, s, g);
}
void launch_p2_tn(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep) {
  lirec::launch(HIP_KERNEL_NAME(gemm_p2_tn_kernel<0>), grid, dim3(512), 0, s, g, nrep);
}
void launch_p2_tng(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep) {
  lirec::launch(HIP_KERNEL_NAME(gemm_p2_tn_kernel<0, true>), grid, dim3(512), 0, s, g, nrep);
}
void launch_p2_tn_reduce(int tiles, int grid, hipStream_t s, const GemmGroup& g, int nrep, const AdamFuse* adam) {
  const AdamFuse none{};
  if (adam) lirec::launch(HIP_KERNEL_NAME(gemm_p2_tn_reduce_kernel<true>), dim3((unsigned)tiles * P2_RED_PARTS), dim3(256), 0, s, g, nrep, grid / nrep, *adam);
  else lirec::launch(HIP_KERNEL_NAME(gemm_p2_tn_reduce_kernel<false>), dim3((unsigned)tiles * P2_RED_PARTS), dim3(256), 0, s, g, nrep, grid / nrep, none);
}
#endif

#else

constexpr int kCfg = LIREC_INST_CFG;
constexpr int kThreads = 64 * TileCfg<kCfg>::WAVES_M * TileCfg<kCfg>::WAVES_N;

void LIREC_CAT(LIREC_CAT(LIREC_CAT(launch_bf_L, LIREC_INST_LAYOUT), _C), LIREC_INST_CFG)(int variant, dim3 grid, hipStream_t s,
                                                                                        const GemmGroup& g) {
  if constexpr (kL != L_NN) {
    if (variant == GV_MAPPED_XB && kL == L_TN) {
      lirec::launch(HIP_KERNEL_NAME(gemm_bf16x3_kernel<kL, kCfg, (kL == L_TN ? 3 : 0), true, true>), grid, dim3(kThreads), 0, s, g);
      return;
    }
    if (variant == GV_TAGGED_XB || variant == GV_MAPPED_XB) {
      lirec::launch(HIP_KERNEL_NAME(gemm_bf16x3_kernel<kL, kCfg, kTag, true, true>), grid, dim3(kThreads), 0, s, g);
      return;
    }
  }
  if (variant == GV_MAPPED && kL == L_TN)
    lirec::launch(HIP_KERNEL_NAME(gemm_bf16x3_kernel<kL, kCfg, (kL == L_TN ? 3 : 0), true>), grid, dim3(kThreads), 0, s, g);
  else if (variant >= GV_TAGGED && kTag != 0)
    lirec::launch(HIP_KERNEL_NAME(gemm_bf16x3_kernel<kL, kCfg, kTag, true>), grid, dim3(kThreads), 0, s, g);
  else if (variant != GV_SCALAR)
    lirec::launch(HIP_KERNEL_NAME(gemm_bf16x3_kernel<kL, kCfg, 0, true>), grid, dim3(kThreads), 0, s, g);
  else
    lirec::launch(HIP_KERNEL_NAME(gemm_bf16x3_kernel<kL, kCfg, 0, false>), grid, dim3(kThreads), 0, s, g);
}

#endif

}  // namespace lirec
