// Split-precision GEMM on PRE-SPLIT bf16 planes, staged by LDS-DMA (gfx950).
//
// The two heavy contractions of the hot path -- layer 1 of both heads (K1: H1 = X W1^T, mlp/model.py:281,286,
// 291-292,307,313,319-320) and its weight gradient (K6: dW1 = dZ1^T X, autograd of the same lines) -- take their
// operands here as bf16 "planes": every fp32 operand a is stored once as hi = bf16_rne(a), lo = bf16_rne(a - hi)
// in two separate row-major bf16 arrays (lirec_stage_features for the feature rows, lirec_split_planes for the
// weights, the un-pooling kernel writes dZ1 that way).  The products are the same three MFMAs per fragment pair as
// gemm_bf16x3.hpp -- hi*hi + hi*lo + lo*hi into one fp32 accumulator, bit-identical results -- but the k-loop no
// longer converts anything: operands go global -> LDS with global_load_lds_dwordx4 (16 B per lane, no VGPRs, no
// VALU), three stages of LDS, two k-tiles of loads in flight across ONE raw s_barrier per k-tile behind a counted
// s_waitcnt vmcnt (cdna_hip_programming.md, "Pipelining across barriers").  The on-the-fly kernel spent 10 VALU per
// 4 elements on the split (VALU 43 % busy against MFMA 34 %) and kept its waves in lockstep on a
// load -> convert -> ds_write -> barrier chain.
//
// Tile 256 x 128 x 32, 8 waves as 4 x 2, each 64 x 64 (2 x 2 MFMA tiles of 32x32x16): 24 MFMAs and 8 ds_read_b128
// (or 16 ds_read_b64_tr_b16) per wave and k16-step.  LDS stage = 48 KiB: A_hi 16 | A_lo 16 | B_hi 8 | B_lo 8;
// three stages = 144 KiB, one workgroup per CU.
//   NT (K1): A[m][k], B[n][k], k contiguous.  Image [row][64 B]; the row's four 16-B slots are XOR-swizzled by
//       (row >> 2) & 3 (conflict-free for the fragment ds_read_b128, as in gemm_bf16x3.hpp).  LDS-DMA writes a wave's
//       64 x 16 B linearly, so the swizzle is applied to the SOURCE address: the lane that fills slot c' of row r
//       fetches chunk c' ^ swz(r) of that row's 64 bytes (same 64-B segment, coalescing unchanged).
//   TN (K6): A[k][m], B[k][n], m / n contiguous.  Image = sub-tiles of [32 k][128 cols] with 256-B rows, chunk ch of
//       row k at 256 k + 16 (ch ^ f(k)), f(k) = ((k & 3) << 2) | ((k >> 2) & 3) (layout (b) of the guide's T10:
//       conflict-free ds_read_b64_tr_b16 for the 32x32x16 operands); again the XOR goes on the source chunk.
// Epilogues, grouped launch, tile decode, XCD remap, split-K slabs: shared with gemm.hpp.  TN also reduces the bias
// gradient (column sums of A over k) on the matrix pipe: the waves of tile column 0 run one extra MFMA per A
// fragment against a fragment of ones.
#pragma once
#include "gemm.hpp"
#include "gemm_bf16x3.hpp"

namespace lirec {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// one LDS-DMA wave-instruction: lane l copies 16 bytes from `src` (per lane) to lds_base + 16 l (wave-uniform base)
__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_base) {
  __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_base, 16, 0, 0);
}

struct PlaneCfg {
  static constexpr int BM = 256, BN = 128, BK = 32;
  static constexpr int WAVES_M = 4, WAVES_N = 2, WM = 2, WN = 2, NTHR = 512;
  static constexpr int A_IMG = BM * BK * 2, B_IMG = BN * BK * 2;          // one plane of one operand: 16 KiB, 8 KiB
  static constexpr int STAGE = 2 * A_IMG + 2 * B_IMG;                      // 48 KiB
  static constexpr int NSTAGE = 3;
};

// AXB / BXB: the operand has no lo plane (a bf16-stored feature block: its low half is exactly zero)
template <int LAYOUT, bool AXB, bool BXB>
__global__ __launch_bounds__(512, 2) void gemm_planes_kernel(const GemmGroup g) {
  using Cf = PlaneCfg;
  constexpr int BM = Cf::BM, BN = Cf::BN, BK = Cf::BK, WM = Cf::WM, WN = Cf::WN;
  static_assert(LAYOUT == L_NT || LAYOUT == L_TN, "planes: forward (NT) and weight-gradient (TN) layouts");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[Cf::NSTAGE * Cf::STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ltile = launch_tile(g);
  if (ltile < 0) return;
  const TileCoord tc = decode_tile<BM, BN>(g, ltile, LAYOUT == L_TN);
  const GemmProblem& p = g.p[tc.pi];
  const int m0 = tc.m0, n0 = tc.n0;
  const int M = tc.M, N = p.N, K = tc.k_end, kb = tc.k_begin;
  if (m0 >= M) return;                            // row-compacted launch: nothing beyond the valid rows
  const int nk = (K - kb + BK - 1) / BK;          // (host guarantees whole k-tiles: K % 32 == 0, or zero-filled tails)

  // ---- LDS-DMA source addresses (per lane) and destinations (per wave) of one k-tile --------------------------
  // Every wave issues: 2 pieces of A_hi, 2 of A_lo, 1 of B_hi, 1 of B_lo (a piece = 1 KiB).
  const unsigned short* a_src[2];                  // element pointers into the hi plane at k = 0 (lo: same offset)
  const unsigned short* b_src;
  long a_lo_off, b_lo_off;                         // lo plane - hi plane, in elements
  int a_dst[2], b_dst;                             // byte offsets inside an image
  long a_kstep, b_kstep;                           // elements per k-tile
  {
    const unsigned short* Ah = reinterpret_cast<const unsigned short*>(p.A);
    const unsigned short* Bh = reinterpret_cast<const unsigned short*>(p.B);
    a_lo_off = AXB ? 0 : reinterpret_cast<const unsigned short*>(p.A_lo) - Ah;
    b_lo_off = BXB ? 0 : reinterpret_cast<const unsigned short*>(p.B_lo) - Bh;
    if constexpr (LAYOUT == L_NT) {
      // piece j of a plane = rows 16 j .. 16 j + 15; lane -> row (lane >> 2), LDS slot lane & 3, source chunk
      // (lane & 3) ^ swz, swz = (row >> 2) & 3 = (lane >> 4) & 3 (the pieces start at multiples of 16 rows)
      const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = 32 * wave + 16 * q + (lane >> 2);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;                  // rows beyond M: any valid row (their outputs are never stored)
        a_src[q] = Ah + (long)gm * p.lda + kb + 8 * chunk;
        a_dst[q] = (32 * wave + 16 * q) * 64;
      }
      const int rowb = 16 * wave + (lane >> 2);
      int gn = n0 + rowb;
      gn = gn < N ? gn : N - 1;
      b_src = Bh + (long)gn * p.ldb + kb + 8 * chunk;
      b_dst = (16 * wave) * 64;
      a_kstep = BK; b_kstep = BK;
    } else {
      // sub-tile [32 k][128 cols], 256-B rows; piece j of a sub-tile = k-rows 4 j .. 4 j + 3; lane -> k-row
      // (lane >> 4), LDS chunk lane & 15, source chunk (lane & 15) ^ f(k)
      // A (256 cols = 2 sub-tiles, 16 pieces): wave w takes pieces 2 w, 2 w + 1 -> sub-tile w >> 2, k-rows 8 (w & 3) ..
      // B (128 cols = 1 sub-tile, 8 pieces): wave w takes piece w -> k-rows 4 w ..
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int k = 8 * (wave & 3) + 4 * q + (lane >> 4);
        const int f = ((k & 3) << 2) | ((k >> 2) & 3);
        const int col = 128 * (wave >> 2) + 8 * ((lane & 15) ^ f);
        int gm = m0 + col;                         // (M is a multiple of 8 on this path; whole chunks are in or out)
        gm = gm + 8 <= M ? gm : 0;
        a_src[q] = Ah + (long)(kb + k) * p.lda + gm;
        a_dst[q] = (wave >> 2) * 8192 + (8 * (wave & 3) + 4 * q) * 256;
      }
      const int k = 4 * wave + (lane >> 4);
      const int f = ((k & 3) << 2) | ((k >> 2) & 3);
      int gn = n0 + 8 * ((lane & 15) ^ f);
      gn = gn + 8 <= N ? gn : 0;
      b_src = Bh + (long)(kb + k) * p.ldb + gn;
      b_dst = (4 * wave) * 256;
      a_kstep = (long)BK * p.lda; b_kstep = (long)BK * p.ldb;
    }
  }
  // piece q (0 .. PER_TILE-1) of k-tile t into `stage`
  constexpr int PER_TILE = 2 + (AXB ? 0 : 2) + 1 + (BXB ? 0 : 1);     // LDS-DMA instructions per wave and k-tile
  auto issue_piece = [&](int t, unsigned char* stage, int q) {
    unsigned char* a_hi = stage;
    unsigned char* a_lo = a_hi + Cf::A_IMG;
    unsigned char* b_hi = a_lo + Cf::A_IMG;
    unsigned char* b_lo = b_hi + Cf::B_IMG;
    const long ao = (long)t * a_kstep, bo = (long)t * b_kstep;
    constexpr int NA = AXB ? 2 : 4;
    if (q < 2) glds16(a_src[q] + ao, a_hi + a_dst[q]);
    else if (q < NA) glds16(a_src[q - 2] + ao + a_lo_off, a_lo + a_dst[q - 2]);
    else if (q == NA) glds16(b_src + bo, b_hi + b_dst);
    else glds16(b_src + bo + b_lo_off, b_lo + b_dst);
  };
  auto issue = [&](int t, unsigned char* stage) {
#pragma unroll
    for (int q = 0; q < PER_TILE; ++q) issue_piece(t, stage, q);
  };

  // ---- fragment read offsets ---------------------------------------------------------------------------------
  const int wm0 = (wave / Cf::WAVES_N) * 32 * WM, wn0 = (wave % Cf::WAVES_N) * 32 * WN;
  const int l31 = lane & 31, lh = lane >> 5;
  int a_frag, b_frag;          // NT: byte offset of (row, slot lh ^ ...) for k16-step 0; TN: of the first transpose read
  int tn_f[2] = {0, 0};        // TN: swizzle of the two transpose reads (t = 0, 1) -- independent of the k16-step
  if constexpr (LAYOUT == L_NT) {
    const int swz = (lane >> 2) & 3;
    a_frag = (wm0 + l31) * 64 + (((lh) ^ swz) << 4);          // slot (2 s + lh) ^ swz: the s bit is XORed in below
    b_frag = (wn0 + l31) * 64 + (((lh) ^ swz) << 4);
  } else {
    const int q = (lane & 15) >> 2, pq = lane & 3, gq = (lane >> 4) & 1;
    tn_f[0] = (q << 2) | ((2 * lh + 0) & 3);
    tn_f[1] = (q << 2) | ((2 * lh + 1) & 3);
    // row (8 lh + q) of the k16-step; chunk base = ((col & 127) >> 3) with col = w0 + 16 g + 4 p (+ 32 i later)
    a_frag = (wm0 >> 7) * 8192 + (8 * lh + q) * 256 + 8 * (pq & 1);
    b_frag = (8 * lh + q) * 256 + 8 * (pq & 1);
    // the chunk index part is kept separately: chunk(i) = ((w0 & 127) >> 3) + 4 i + 2 g + (p >> 1)
    tn_f[0] |= 0; tn_f[1] |= 0;
    (void)gq;
  }
  const int tn_chunk_a = ((wm0 & 127) >> 3) + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
  const int tn_chunk_b = ((wn0 & 127) >> 3) + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);

  auto frag_a = [&](const unsigned char* img, int i, int s) -> bf16x8 {
    if constexpr (LAYOUT == L_NT) {
      return *reinterpret_cast<const bf16x8*>(img + ((a_frag + 32 * 64 * i) ^ (32 * s)));
    } else {
      const unsigned char* base = img + a_frag + 16 * s * 256;
      const int ch = tn_chunk_a + 4 * i;
      const s16x4 x = lds_tr16(base + 16 * (ch ^ tn_f[0]));
      const s16x4 y = lds_tr16(base + 4 * 256 + 16 * (ch ^ tn_f[1]));
      const s16x8 v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
      return *reinterpret_cast<const bf16x8*>(&v);
    }
  };
  auto frag_b = [&](const unsigned char* img, int j, int s) -> bf16x8 {
    if constexpr (LAYOUT == L_NT) {
      return *reinterpret_cast<const bf16x8*>(img + ((b_frag + 32 * 64 * j) ^ (32 * s)));
    } else {
      const unsigned char* base = img + b_frag + 16 * s * 256;
      const int ch = tn_chunk_b + 4 * j;
      const s16x4 x = lds_tr16(base + 16 * (ch ^ tn_f[0]));
      const s16x4 y = lds_tr16(base + 4 * 256 + 16 * (ch ^ tn_f[1]));
      const s16x8 v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
      return *reinterpret_cast<const bf16x8*>(&v);
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // row-mapped forward launch: the original row ids of this wave's 64 rows (dropout counters), requested now and
  // first touched in the epilogue -- the epilogue then has no load in front of its Philox calls and stores
  int rid_lane = 0;
  const bool have_rid = (LAYOUT == L_NT) && p.rowmap != nullptr;
  if (have_rid) { const int r = m0 + wm0 + lane; rid_lane = p.rowmap[r < M ? r : M - 1]; }
  // bias gradient of the weight-gradient layout: column sums of A over k, on the matrix pipe (tile column 0, the
  // waves of wave column 0): A_frag x ones
  const bool do_dbias = (LAYOUT == L_TN) && p.dbias != nullptr && tc.tn == 0 && (wave % Cf::WAVES_N) == 0;
  f32x16 accb[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  // One k-tile: the MFMAs of `stage`, with the LDS-DMA pieces of tile t_next (into `next_stage`, free since the barrier)
  // dealt out between them -- one piece behind every few MFMAs.  An LDS-DMA instruction costs the issuing wave
  // 60-180 cycles; issued as a block right behind the barrier (the first version) all eight waves sat in that block
  // together with the matrix pipe idle: loop time = MFMA time + DMA-issue time.  Behind an MFMA the same cycles ride in
  // the shadow of the matrix pipe.
  auto compute = [&](const unsigned char* stage, bool dma, int t_next, unsigned char* next_stage) {
    const unsigned char* a_hi = stage;
    const unsigned char* a_lo = a_hi + Cf::A_IMG;
    const unsigned char* b_hi = a_lo + Cf::A_IMG;
    const unsigned char* b_lo = b_hi + Cf::B_IMG;
    // the fragments of BOTH k16-steps are requested before the first MFMA (64 VGPRs): the second step's LDS reads are
    // in flight under the first step's MFMAs
    bf16x8 ah[2][WM], al[2][WM], bh[2][WN], bl[2][WN];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < WM; ++i) { ah[s][i] = frag_a(a_hi, i, s); if constexpr (!AXB) al[s][i] = frag_a(a_lo, i, s); }
#pragma unroll
      for (int j = 0; j < WN; ++j) { bh[s][j] = frag_b(b_hi, j, s); if constexpr (!BXB) bl[s][j] = frag_b(b_lo, j, s); }
    }
    constexpr int NGRP = 2 * WM * WN;                   // (s, i, j) groups of up to three MFMAs
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if constexpr (!AXB) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s][i], bh[s][j], acc[i][j], 0, 0, 0);
          if constexpr (!BXB) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][i], bl[s][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][i], bh[s][j], acc[i][j], 0, 0, 0);
          const int grp = (s * WM + i) * WN + j;
          // pieces [grp PER_TILE / NGRP, (grp + 1) PER_TILE / NGRP) follow this group
#pragma unroll
          for (int q = grp * PER_TILE / NGRP; q < (grp + 1) * PER_TILE / NGRP; ++q)
            if (dma) issue_piece(t_next, next_stage, q);
        }
      if constexpr (LAYOUT == L_TN) {
        if (do_dbias) {
#pragma unroll
          for (int i = 0; i < WM; ++i) {
            if constexpr (!AXB) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s][i], ones, accb[i], 0, 0, 0);
            accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][i], ones, accb[i], 0, 0, 0);
          }
        }
      }
    }
  };

  // ---- k-loop: 3-stage ring, loads two k-tiles ahead, one barrier per k-tile -----------------------------------
  // Before tile t is read:  this wave's loads of tile t have landed (counted vmcnt: the PER_TILE instructions of
  // tile t + 1 may stay in flight), then the barrier -- every wave's loads of tile t have landed AND every wave has
  // finished reading tile t - 1, whose stage the loads of tile t + 2, issued right behind the barrier, overwrite.
  if (nk > 0 && !(g.ablate & 4)) {
    unsigned char* s0 = smem;
    unsigned char* s1 = smem + Cf::STAGE;
    unsigned char* s2 = smem + 2 * Cf::STAGE;
    // (diagnostics, lirec_debug_set: 16 = no LDS-DMA is issued, 32 = no LDS reads / MFMAs; results are garbage)
    const bool do_issue = !(g.ablate & 16), do_compute = !(g.ablate & 32);
    if (do_issue) {
      issue(0, s0);
      if (nk > 1) issue(1, s1);
    }
    for (int t = 0; t < nk; ++t) {
      if (t + 1 < nk) {
        if constexpr (PER_TILE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (PER_TILE == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      const bool dma = do_issue && t + 2 < nk;
      if (do_compute) compute(s0, dma, t + 2, s2);
      else if (dma) issue(t + 2, s2);
      unsigned char* tmp = s0; s0 = s1; s1 = s2; s2 = tmp;
    }
  }

  gemm_epilogue<WM, WN, LAYOUT>(p, acc, m0, n0, wm0, wn0, lane, tc.split, M, have_rid, rid_lane);

  if constexpr (LAYOUT == L_TN) {
    if (do_dbias && (l31 == 0)) {
      // every column of accb holds the row sums; lanes 0 and 32 own the two row halves of each 4-row group
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < M) {
            if (p.ksplit > 1) p.dbias_slab[(long)tc.split * p.M + m] = accb[i][r];
            else p.dbias[m] += accb[i][r];
          }
        }
    }
  }
}

}  // namespace lirec
