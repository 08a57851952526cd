// Launch entry points of the GEMM kernel instantiations.  Every (core, layout, tile config) lives in its
// own translation unit (gemm_inst.hip compiled once per combination, see lirec_amd/build.py) so that
// the library builds in parallel; the C-ABI translation unit only sees these plain functions.
#pragma once
#include <hip/hip_runtime.h>
#include "gemm.hpp"
#include "record.hpp"

namespace lirec {

// variant: 0 = element-wise staging (any alignment), 1 = dwordx4 staging, 2 = dwordx4 staging under the
// tagged symbol (a distinct kernel name for the heavy call sites, so a kernel trace tells them apart)
// 3 = tagged + every problem row-mapped (TN only; elsewhere the same as 2)
// 4 / 5 = 2 / 3 with the X operand stored as bf16 (bf16x3 core, NT and TN only)
enum { GV_SCALAR = 0, GV_VEC = 1, GV_TAGGED = 2, GV_MAPPED = 3, GV_TAGGED_XB = 4, GV_MAPPED_XB = 5 };

#define LIREC_DECL_LAUNCH(L)                                                                              \
  void launch_f32_L##L(bool big, int variant, dim3 grid, hipStream_t s, const GemmGroup& g);             \
  void launch_naive_L##L(dim3 grid, hipStream_t s, const GemmProblem& p);                                 \
  void launch_bf_L##L##_C0(int variant, dim3 grid, hipStream_t s, const GemmGroup& g);                   \
  void launch_bf_L##L##_C1(int variant, dim3 grid, hipStream_t s, const GemmGroup& g);                   \
  void launch_bf_L##L##_C3(int variant, dim3 grid, hipStream_t s, const GemmGroup& g);                   \
  void launch_bf_L##L##_C4(int variant, dim3 grid, hipStream_t s, const GemmGroup& g);
LIREC_DECL_LAUNCH(0)
LIREC_DECL_LAUNCH(1)
LIREC_DECL_LAUNCH(2)
void launch_bf_L2_C2(int variant, dim3 grid, hipStream_t s, const GemmGroup& g);     // 256 x 256: weight-gradient layout only
// layer 1 on q32b operands (gemm_p2.hpp): persistent launches of `grid` workgroups; `tiles` = 256 x 256 output tiles of the
// weight gradient (its reduce kernel's grid, `grid` = the GEMM launch's workgroups)
void launch_p2_nt(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep);
void launch_p2_ntg(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep);     // rows gathered through GemmProblem::srow
void launch_p2_tng(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep);
void launch_p2_tng1(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep);
void launch_p2_ntg1(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep);
void launch_p2_ntg64(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep);    // single pass: q16c rows and weights, 64 of k per step
void launch_p2_tng1o(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep);
// wave-specialised persistent kernel (gemm_p3.hpp), 128 x 96 tiles, operands k-contiguous q32b rows: the gate's three GEMMs
// (ni = 3: 128 x 96 tiles; ni = 4: 128 x 128 -- the host picks the one with fewer tile rounds x tile time: p3_pick_ni)
void launch_p3_fwd(int ni, dim3 grid, hipStream_t s, const GemmGroup& g);        // bias + relu + dropout
void launch_p3_dgrad(int ni, dim3 grid, hipStream_t s, const GemmGroup& g);      // (acc + beta C) * tanh' * dropout factor
void launch_p3_wgrad(dim3 grid, hipStream_t s, const GemmGroup& g);              // C = beta C + acc, bias gradient = row sums of A
void launch_p2_nn(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep);      // data gradient through k-major weights (gate dEE)
void launch_p2_tn(dim3 grid, hipStream_t s, const GemmGroup& g, int nrep);
void launch_p2_tn_reduce(int tiles, int grid, hipStream_t s, const GemmGroup& g, int nrep, const AdamFuse* adam = nullptr);
#undef LIREC_DECL_LAUNCH

}  // namespace lirec
