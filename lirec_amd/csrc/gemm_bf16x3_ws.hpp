// Wave-specialised variant of the split-precision GEMM core (NT layout: forward GEMMs).
//
// The one-role kernel (gemm_bf16x3.hpp) makes every wave do everything -- global loads, fp32 -> bf16 hi/lo split,
// LDS writes, fragment reads, MFMAs -- and the per-k-tile barrier keeps the waves of a workgroup in the same phase, so
// the matrix pipe (33 % busy by PMC), the VALU (~40 %) and the LDS take turns instead of overlapping.
// Here the two jobs belong to different waves of one 8-wave workgroup:
//   waves 4..7  PRODUCERS: stage k-tile kt+2 (global -> registers -> split -> LDS stage (kt+2)%3) and request kt+3;
//   waves 0..3  CONSUMERS: fragments of k-tile kt from LDS stage kt%3 -> MFMAs, each wave a 64x64 quarter of the
//               128x128 tile (4 accumulator tiles: 2/3 of the fragment bytes per MFMA of the 32x64 one-role waves).
// One barrier per k-tile orders both hand-offs (stage kt+2 complete; stage kt free); with three stages the consumers
// fetch the first fragments of tile kt+1 before that barrier, so no LDS latency is exposed behind it.  A SIMD holds one producer and
// one consumer wave, so its VALU/LDS-write work and its MFMA work come from different instruction streams and
// overlap by construction instead of by luck.  One workgroup per CU (the consumer's 64x64 tile + double-buffered
// fragments need > 128 VGPRs).
//
// STATUS: experimental -- reachable only with lirec_debug_set(.., force_cfg = 6); the launch policy never picks it.
// Bit-identical to the one-role kernel; on the K1 shape (dense, 18 432 rows) 0.73 ms against 0.56-0.60 ms.  The
// ablation masks (GemmGroup::ablate, tools/ablate_gemm.py) say why and what to do next: the CONSUMERS alone run at
// the MFMA rate the box sustains (0.24 ms of MFMA work; 0.49 ms with the per-tile overhead), the PRODUCERS alone need
// 0.62 ms (0.75 us per k-tile for one staging wave per SIMD, about twice their instruction-issue time; with a
// reload branch per chunk it was 1.3 us), and barriers + prologue + epilogue alone cost 0.26 ms: with one workgroup
// per CU nothing overlaps the epilogue (16 Philox calls per thread, on the 4 consumer waves only), while K = 768..2048
// per tile makes that fixed work as large as the MFMA work.  Next: two producer waves per SIMD (or LDS-DMA staging of
// pre-split planes), persistent workgroups whose producers run ahead into the next tile, the epilogue on all waves.
#pragma once
#include "gemm_bf16x3.hpp"

namespace lirec {

template <int LAYOUT, int TAG, bool VEC>
__global__ __launch_bounds__(512, 2) void gemm_bf16x3_ws_kernel(const GemmGroup g) {
  static_assert(LAYOUT == L_NT, "wave-specialised core: forward (NT) layout only");
  constexpr int BM = 128, BN = 128, BK = 32;
  using TA = OperandTile<true, BM>;
  using TB = OperandTile<true, BN>;
  constexpr int BUF = 2 * (TA::BYTES + TB::BYTES);     // A_hi, A_lo, B_hi, B_lo of one stage (32 KB)
  constexpr int NST = 3;                               // LDS stages: tile kt is consumed while kt+1 is already complete
                                                       // (its first fragments are prefetched) and kt+2 is being staged
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TileCoord tc = decode_tile<BM, BN>(g, xcd_remap(blockIdx.x, gridDim.x), false);
  const GemmProblem& p = g.p[tc.pi];
  const int m0 = tc.m0, n0 = tc.n0;
  const int M = tc.M, N = p.N, K = tc.k_end, kb = tc.k_begin;
  if (m0 >= M) return;                                 // row-compacted launch: nothing beyond the valid rows
  const int nk = (K - kb + BK - 1) / BK;

  if (wave >= 4) {
    // ------------------------------------------------------------------ producers
    const int pt = tid - 256;                          // 0..255: row pt>>3 (+ 32 i), k-quad pt&7
    const float* a_rowptr[4];
    const float* b_rowptr[4];
    bool a_ok[4], b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + (pt >> 3) + 32 * i, n = n0 + (pt >> 3) + 32 * i;
      a_ok[i] = m < M; b_ok[i] = n < N;
      a_rowptr[i] = p.A + phys_row(p, a_ok[i] ? m : 0) * p.lda;
      b_rowptr[i] = p.B + (long)(b_ok[i] ? n : 0) * p.ldb;
    }
    // LDS row pt>>3, 8-byte piece pt&7, 16-B slot swizzled by (row >> 2) & 3 (same image as the one-role kernel)
    const int store_off = (pt >> 3) * 64 + (((((pt & 7) >> 1) ^ ((pt >> 5) & 3)) << 4) | ((pt & 1) << 3));
    const int kq = 4 * (pt & 7);
    f32x4 ra[4], rb[4];
    // Branch-free: tiles beyond the last one are clamped to the last tile's address and written to a stage nobody
    // reads again, so the loop body has no run-time branch and every wait on a load is a counted one.
    const int k_last = kb + (nk > 0 ? nk - 1 : 0) * BK;
    auto ktile0 = [&](int t) { const int k = kb + t * BK; return k < k_last ? k : k_last; };
    const bool interior = (m0 + BM <= M) && (n0 + BN <= N) && (((K - kb) & (BK - 1)) == 0);
    auto load_tile = [&](int k0) {
      const int k = k0 + kq;
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = raw4<VEC>(a_rowptr[i] + k, a_ok[i] ? K - k : 0, p.A);
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[i] = raw4<VEC>(b_rowptr[i] + k, b_ok[i] ? K - k : 0, p.B);
    };
    // registers -> LDS stage `buf` (tile at k0); chunk i is re-loaded from the tile at k1 right after it is written
    auto store_tile = [&](int buf, int k0, int k1, auto edge_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      unsigned char* a_hi = smem + buf * BUF;
      unsigned char* a_lo = a_hi + TA::BYTES;
      unsigned char* b_hi = a_lo + TA::BYTES;
      unsigned char* b_lo = b_hi + TB::BYTES;
      const int k = k0 + kq, kn = k1 + kq;
      uint2 h, l;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (EDGE) split4(mask4(ra[i], a_ok[i] ? K - k : 0), h, l);
        else split4(ra[i], h, l);
        *reinterpret_cast<uint2*>(a_hi + store_off + 32 * 64 * i) = h;
        *reinterpret_cast<uint2*>(a_lo + store_off + 32 * 64 * i) = l;
        if constexpr (EDGE) ra[i] = raw4<VEC>(a_rowptr[i] + kn, a_ok[i] ? K - kn : 0, p.A);
        else ra[i] = raw4<VEC>(a_rowptr[i] + kn, 4, p.A);
        if constexpr (EDGE) split4(mask4(rb[i], b_ok[i] ? K - k : 0), h, l);
        else split4(rb[i], h, l);
        *reinterpret_cast<uint2*>(b_hi + store_off + 32 * 64 * i) = h;
        *reinterpret_cast<uint2*>(b_lo + store_off + 32 * 64 * i) = l;
        if constexpr (EDGE) rb[i] = raw4<VEC>(b_rowptr[i] + kn, b_ok[i] ? K - kn : 0, p.B);
        else rb[i] = raw4<VEC>(b_rowptr[i] + kn, 4, p.B);
      }
    };
    auto run = [&](auto edge_tag) {
      // prologue: tiles 0 and 1 -> stages 0 and 1, registers <- tile 2
      load_tile(kb);
      store_tile(0, kb, ktile0(1), edge_tag);
      store_tile(1, ktile0(1), ktile0(2), edge_tag);
      __syncthreads();
      for (int kt = 0; kt < nk; ++kt) {
        if (!(g.ablate & 2)) store_tile((kt + 2) % NST, ktile0(kt + 2), ktile0(kt + 3), edge_tag);
        __syncthreads();
      }
    };
    if (nk <= 0) { __syncthreads(); return; }
    if (interior) run(std::false_type{});
    else run(std::true_type{});
    return;
  }

  // -------------------------------------------------------------------- consumers
  constexpr int WM = 2, WN = 2;
  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int l31 = lane & 31, lh = lane >> 5;
  const int swz = (lane >> 2) & 3;
  const int s0 = 16 * (lh ^ (swz & 1)) + 32 * (swz >> 1);
  const int a_frag = (wm0 + l31) * 64 + s0, b_frag = (wn0 + l31) * 64 + s0;
  struct Frags { bf16x8 ah[WM], al[WM], bh[WN], bl[WN]; };
  auto read_frags = [&](Frags& f, int buf, int s) {
    const unsigned char* a_hi = smem + buf * BUF;
    const unsigned char* a_lo = a_hi + TA::BYTES;
    const unsigned char* b_hi = a_lo + TA::BYTES;
    const unsigned char* b_lo = b_hi + TB::BYTES;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      f.ah[i] = *reinterpret_cast<const bf16x8*>(a_hi + (a_frag ^ (32 * s)) + 32 * 64 * i);
      f.al[i] = *reinterpret_cast<const bf16x8*>(a_lo + (a_frag ^ (32 * s)) + 32 * 64 * i);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      f.bh[j] = *reinterpret_cast<const bf16x8*>(b_hi + (b_frag ^ (32 * s)) + 32 * 64 * j);
      f.bl[j] = *reinterpret_cast<const bf16x8*>(b_lo + (b_frag ^ (32 * s)) + 32 * 64 * j);
    }
  };
  auto mma = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
      }
  };
  __syncthreads();                                      // stages 0 and 1 are complete
  Frags f0, f1;
  if (nk > 0) read_frags(f0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt % NST;
    if (!(g.ablate & 1)) {
    read_frags(f1, buf, 1);                             // second k-step's fragments land under the first one's MFMAs
    mma(f0);
    if (kt + 1 < nk) read_frags(f0, (kt + 1) % NST, 0); // next tile (complete since the last barrier): no LDS wait after
    mma(f1);                                            // the barrier below
    }
    __syncthreads();                                    // stage kt+2 complete, stage kt free
  }
  gemm_epilogue<WM, WN, LAYOUT>(p, acc, m0, n0, wm0, wn0, lane, tc.split, M);
}

}  // namespace lirec
