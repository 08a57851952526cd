// The gate GEMMs (GatingUnit, mlp/model.py:349-354: 3072 x 3072 weights against ~1024 candidate rows) on q32b operands with
// WAVE-SPECIALISED workgroups (gfx950): forward, data gradient and weight gradient are ONE kernel form.
//
// Why a kernel of its own.  tools/micro/l2_lds_intake.hip (profiles/r04_l2_lds_intake.txt): a CU takes 131 GB/s of cache-resident
// operand bytes into LDS when four or more waves issue LDS-DMA back to back; ONE issuing wave reaches 33.5, because a
// global_load_lds_dwordx4 holds the issuing wave for 60-140 cycles.  In gemm_p2.hpp every wave both issues its share of the DMA and
// computes: at 256-row tiles the requests hide behind 96 MFMAs, at the tiles a 1024-row problem leaves for 256 CUs they do not.
// Here the two jobs belong to different waves: waves 0-3 COMPUTE (2 x 2, one per SIMD, each owns its matrix pipe), waves 4-7
// LOAD (LDS-DMA only; their issue stalls cost the compute waves nothing).
//
// Round 5 (what changed against the 128 x 128 / three-layout kernel of round 4, HISTORY):
//   * EVERY operand is k-contiguous q32b rows -- C[M][N] = A[M][K] . B[N][K]^T, gemm_p2's swizzled [row][128 B] LDS image for
//     both, fragments by plain ds_read_b128.  The data gradient reads the weights through a TRANSPOSED q32b copy (Wg^T, staged
//     with Wg on the side stream), the weight gradient reads dZg^T and EE^T (written by the same staging launches that write dZg
//     and EE: split_q32b_dual_kernel).  The k-major images and transposed LDS reads of round 4 pinned the tile to multiples of
//     64 columns (their chunk swizzle) -- and with it the forward / data gradient to 192 tiles on 256 CUs;
//   * the tile is a template parameter: wave tile (16 MI) x (16 NI), workgroup tile (32 MI) x (32 NI).  128 x 96 (MI 4, NI 3)
//     cuts the 1024 x 3072 forward / data gradient into exactly 256 tiles, the 3072 x 3072 weight gradient into 768 = three
//     full rounds;
//   * the compute waves' loop is software-pipelined over WHOLE k-steps: the fragments of step t + 1 are read (into a second
//     register set) while the MFMAs of step t issue, so no MFMA ever waits for an LDS read, and a slot is free one barrier
//     earlier than before (its fragments are in registers when the step that multiplies them starts): with the same five
//     slots four k-steps are in flight instead of three.  tools/micro/p3_bench.hip: the old loop spent 1040 cycles per k-step
//     for 768 of MFMA, ~670 of them with the MFMAs removed (LDS round trips exposed once per row of fragments);
//   * PERSISTENT over tiles: a workgroup walks its tiles in one launch, the loaders run ahead into the next tile while the
//     compute waves store the current one (the weight gradient's three rounds pay one prologue);
//   * tiles are dealt to the XCDs as rectangular blocks of the output (g.p3_xm x 8 / g.p3_xm XCD grid): an XCD's workgroups share
//     a few row panels and a few column panels, walking k together.
// EPI 0 (forward): p.A rows [M][K], p.B weights [N][K]; bias + relu + dropout (Philox words drawn here).
// EPI 1 (data gradient dEE = dZg Wg): p.A = dZg rows [M][K = N_gate], p.B = rows of Wg^T; (acc + beta C) * tanh' * dropout factor.
// EPI 2 (weight gradient dWg = dZg^T EE): p.A = dZg^T [M = N_gate][K = n], p.B = EE^T [N = K_gate][n]; C = beta C + acc; the bias
//   gradient (row sums of p.A) rides along on the matrix pipe (A fragment x ones): the 2 MI sixteen-row fragments of a row panel
//   are dealt to the column tiles tn = 0 .. 2 MI - 1 of that panel, one each (one extra MFMA pair per k-step in one wave).
// ONE: gemm mode 3 (BASELINE config 5's arithmetic) -- one MFMA per product: bf16(a) bf16(b), fp32 accumulate.  There are no lo
//   halves: (r6) the operands are staged as q16c (bf16 values, 64-column blocks of 4 KiB, 128-byte rows: split_q32b_dual_kernel's
//   fmt16c), a k-step covers 64 of k, and an image row's 128 bytes are ONE WHOLE LINE of the operand (chunks 0-3: k 0-31 of the
//   step, chunks 4-7: k 32-63) -- the addressing of the three-pass form on a matrix of half the columns, the same images, swizzle and
//   fragment reads, two MFMAs (one per 32 of k) where that form issues three for half the k.  (Round 5 fetched the hi halves of
//   two consecutive 4-KiB blocks of q32b operands: the same bytes into LDS, but as 64-byte half lines, twice the lines through
//   the fabric into L2, and a staging pass that wrote lo halves nothing read.)
#pragma once
#include "gemm_p2.hpp"

namespace lirec {

template <int MI, int NI>
struct P3 {
  static_assert(MI >= NI && NI >= 1, "the next step's B fragments ride with the first NI rows of A fragments");
  static constexpr int BM = 32 * MI, BN = 32 * NI, BK = 32, NTHR = 512;
  static constexpr int AIMG = BM * 128, BIMG = BN * 128, SLOT = AIMG + BIMG;
  static constexpr int NSLOT = (160 * 1024 / SLOT) < 5 ? (160 * 1024 / SLOT) : 5;
  static constexpr int LDS_BYTES = NSLOT * SLOT;
  static constexpr int REQ = MI + NI;                        // LDS-DMA requests (1 KiB each) per loader wave and k-step
  static_assert(NSLOT >= 3 && (NSLOT - 1) * REQ <= 63, "ring depth / vmcnt range");
};

// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N - 1>) (the scheduling builtins want constant arguments)
template <int I> struct P3Const { static constexpr int value = I; };
template <int N, int I = 0, class F> __device__ __forceinline__ void p3_static_for(F&& f) {
  if constexpr (I < N) { f(P3Const<I>{}); p3_static_for<N, I + 1>(f); }
}

template <int N> __device__ __forceinline__ void p3_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `steps` k-steps' requests of this wave are outstanding
template <int REQ, int MAXS> __device__ __forceinline__ void p3_wait_steps(int steps) {
  if (steps <= 0) p3_vmcnt<0>();
  else if (steps == 1 || MAXS == 1) p3_vmcnt<REQ>();
  else if (steps == 2 || MAXS == 2) p3_vmcnt<2 * REQ>();
  else if (steps == 3 || MAXS == 3) p3_vmcnt<3 * REQ>();
  else p3_vmcnt<4 * REQ>();
}

// The r-th tile of workgroup b: (problem, tm, tn within the problem); false when the workgroup has no r-th tile.
// The tile space is g.p3_tm row tiles x g.p3_tn column tiles, the problems of the group side by side along tn (same M, same K).
template <int BN>
__device__ __forceinline__ bool p3_tile_of(const GemmGroup& g, int b, int G, int r, int& pi, int& tm, int& tn) {
  const int TM = g.p3_tm, TN = g.p3_tn;
  int tng;
  if (g.p3_xm > 0) {
    // XCD x = b & 7 owns the block (x / xn, x % xn) of the output, its G / 8 workgroups walk the block row by row
    const int xn = 8 / g.p3_xm, x = b & 7, w = b >> 3, W = G >> 3;
    const int bm = TM / g.p3_xm, bn = TN / xn, i = w + r * W;
    if (i >= bm * bn) return false;
    tm = (x / xn) * bm + i / bn;
    tng = (x % xn) * bn + i % bn;
  } else {
    const int L = b + r * G;
    if (L >= TM * TN) return false;
    tng = L / TM;
    tm = L - tng * TM;
  }
  pi = 0;
  for (int i = 0; i + 1 < g.nprob; ++i) {
    const int t = g.p[i].N / BN;
    if (tng >= t) { tng -= t; pi = i + 1; } else break;
  }
  tn = tng;
  return true;
}

template <int MI, int NI, int EPI, bool ONE, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_p3_kernel(const GemmGroup g) {
  using T = P3<MI, NI>;
  constexpr int NS = T::NSLOT;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[T::LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = gridDim.x, b = blockIdx.x;
  const int nk = ONE ? g.p[0].K >> 6 : g.p[0].K >> 5;           // (the same for every problem of the group: host; ONE: 64 of k per step)
  int nmine = 0;
  {
    int pi, tm, tn;
    while (p3_tile_of<T::BN>(g, b, G, nmine, pi, tm, tn)) ++nmine;
  }
  const int total = nmine * nk;
  if (total == 0) return;
  const unsigned lds0 = p2_lds_addr(smem);

  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves
    // Request j of an image = its rows 8 j .. 8 j + 7 (1 KiB): lane l lands at row 8 j + (l >> 3), chunk position l & 7, and
    // fetches the source chunk that belongs there: (l & 7) ^ ((row >> 1) & 7).  Loader lw issues requests MI lw .. MI lw + MI - 1
    // of the A image and NI lw .. NI lw + NI - 1 of the B image.
    const int lw = wave - 4;
    unsigned a_off[MI], b_off[NI];
    // (ONE: the row's 128 bytes are 64 consecutive bf16 values of a q16c operand -- the same chunk arithmetic)
    auto src_off = [&](int ri) -> unsigned {
      const unsigned sc = (unsigned)((lane & 7) ^ ((ri >> 1) & 7));
      return (unsigned)(ri & 31) * 128u + 16u * sc;
    };
#pragma unroll
    for (int q = 0; q < MI; ++q) a_off[q] = src_off(8 * (MI * lw + q) + (lane >> 3));
#pragma unroll
    for (int q = 0; q < NI; ++q) b_off[q] = src_off(8 * (NI * lw + q) + (lane >> 3));
    // issue cursor: tile r_i, k-step kt_i of it
    int r_i = -1, kt_i = nk;
    const unsigned char* a_base = nullptr;
    const unsigned char* b_base = nullptr;
    long a_bs = 0, b_bs = 0;                                    // bytes between 32-row blocks of the operands
    auto issue = [&](int slot) {
      if (kt_i == nk) {
        kt_i = 0; ++r_i;
        int pi, tm, tn;
        (void)p3_tile_of<T::BN>(g, b, G, r_i, pi, tm, tn);
        const GemmProblem& p = g.p[pi];
        a_bs = (long)(p.lda >> (ONE ? 6 : 5)) * 4096; b_bs = (long)(p.ldb >> (ONE ? 6 : 5)) * 4096;      // (ONE: 64-column blocks)
        a_base = reinterpret_cast<const unsigned char*>(p.A) + (long)(MI * tm) * a_bs;
        b_base = reinterpret_cast<const unsigned char*>(p.B) + (long)(NI * tn) * b_bs;
      }
      if constexpr ((ABL & 1) == 0) {
        const unsigned so = lds0 + (unsigned)slot * T::SLOT;
#pragma unroll
        for (int q = 0; q < NI; ++q) {
          const int j = NI * lw + q;
          p2_dma16(b_base + (long)(j >> 2) * b_bs + 4096L * kt_i, b_off[q], so + T::AIMG + (unsigned)j * 1024u);
        }
#pragma unroll
        for (int q = 0; q < MI; ++q) {
          const int j = MI * lw + q;
          p2_dma16(a_base + (long)(j >> 2) * a_bs + 4096L * kt_i, a_off[q], so + (unsigned)j * 1024u);
        }
      }
      ++kt_i;
    };
    int islot = 0;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      if (u < total) issue(islot);
      islot = islot == NS - 1 ? 0 : islot + 1;
    }
    // in front of the first barrier: step 0 has landed (all but the steps behind it)
    p3_wait_steps<T::REQ, NS - 1>(total - 1 < NS - 1 ? total - 1 : NS - 1);
    __builtin_amdgcn_s_barrier();
    // in front of barrier s: steps <= s + 1 have landed; behind it: step s + NS goes into the slot of step s, whose fragments
    // the compute waves took into registers during step s - 1
    for (int s = 0; s < total; ++s) {
      const int left = total - s - 2;
      p3_wait_steps<T::REQ, NS - 2>(left < NS - 2 ? left : NS - 2);
      if constexpr ((ABL & 32) == 0) __builtin_amdgcn_s_barrier();
      if (s + NS < total) issue(islot);
      islot = islot == NS - 1 ? 0 : islot + 1;
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int wr = wave >> 1, wc = wave & 1, gq = lane >> 4, l15 = lane & 15;
  const int frag = l15 * 128 + ((gq ^ ((l15 >> 1) & 7)) << 4);
  const int lo_d = 64 - 2 * (frag & 64);                       // lo address = hi address ^ 64
  const int a_frag = frag + (MI * wr) * 2048, b_frag = T::AIMG + frag + (NI * wc) * 2048;
  struct Frags { bf16x8 ah[MI], al[MI], bh[NI], bl[NI]; };
  Frags f0, f1;
  f32x4v acc[MI][NI];
  f32x4v accb = f32x4v{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int n = 0; n < NI; ++n) acc[i][n] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // current tile
  int r_c = 0, kt = 0, pi = 0, tm = 0, tn = 0;
  (void)p3_tile_of<T::BN>(g, b, G, 0, pi, tm, tn);
  // (EPI 2) which sixteen-row fragment's bias gradient this wave carries in the current tile: -1 = none
  auto db_of = [&](int pi_, int tn_) -> int {
    if constexpr (EPI != 2) return -1;
    return (g.p[pi_].dbias != nullptr && wc == 0 && tn_ < 2 * MI && tn_ / MI == wr) ? tn_ % MI : -1;
  };
  int db_i = db_of(pi, tn);

  auto epilogue = [&]() {
    const GemmProblem& p = g.p[pi];
    const bool drop = p.thresh != 0u;
    unsigned key_lo = p.seed_lo, key_hi = p.seed_hi;
    if (drop) apply_seed_offset(key_lo, key_hi, p.seed_dev);
    const int row0 = T::BM * tm + 16 * MI * wr + 4 * gq, col0 = T::BN * tn + 16 * NI * wc + l15;
    if constexpr (EPI == 0) {
      float bias_n[NI];
#pragma unroll
      for (int n = 0; n < NI; ++n) bias_n[n] = p.bias ? p.bias[col0 + 16 * n] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row4 = row0 + 16 * i;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
          const int col = col0 + 16 * n;
          unsigned w[4] = {0u, 0u, 0u, 0u};
          if (drop) philox4((unsigned)(p.drop_col_off + col), (unsigned)(row4 >> 2), p.site, 0u, key_lo, key_hi, w);
          float* cp = p.C + (long)row4 * p.ldc + col;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = relu_f(acc[i][n][j] + bias_n[n]);
            if (drop) v = w[j] >= p.thresh ? v * p.drop_scale : 0.f;
            cp[(long)j * p.ldc] = v;
          }
        }
      }
    } else if constexpr (EPI == 1) {
      const bool has_beta = p.beta != 0.f;
      // (r6) The tanh outputs (and, beta != 0, the old values) of HALF the wave's row fragments are requested together, then
      // that half is finished: two (128 x 128 tiles: four) exposed round trips per tile instead of MI x NI = 12 (16) (one per 16 x 16 block, each behind a
      // compiler barrier: ~1.3 k cycles apiece of a tile's exposed tail).  Half, not all: one fragment register set of the
      // software-pipelined k-loop is live across the epilogue, and 2 x 48 more values would not fit beside it.
      // (the 128 x 128 tile's sixteen more accumulators leave room for one fragment row at a time)
      constexpr int IH = NI <= 3 ? (MI + 1) / 2 : 1;
      constexpr int NH = (MI + IH - 1) / IH;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        float ax[IH][NI][4], old[IH][NI][4];
#pragma unroll
        for (int ii = 0; ii < IH; ++ii) {
          const int i = h * IH + ii;
          if (i >= MI) continue;
          const int row4 = row0 + 16 * i;
#pragma unroll
          for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              ax[ii][n][j] = p.aux[(long)(row4 + j) * p.ldaux + col0 + 16 * n];
              old[ii][n][j] = has_beta ? p.C[(long)(row4 + j) * p.ldc + col0 + 16 * n] : 0.f;
            }
        }
#pragma unroll
        for (int ii = 0; ii < IH; ++ii) {
          const int i = h * IH + ii;
          if (i >= MI) continue;
          const int row4 = row0 + 16 * i;
#pragma unroll
          for (int n = 0; n < NI; ++n) {
            const int col = col0 + 16 * n;
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (drop) philox4((unsigned)(p.drop_col_off + col), (unsigned)(row4 >> 2), p.site, 0u, key_lo, key_hi, w);
            float* cp = p.C + (long)row4 * p.ldc + col;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = acc[i][n][j] + p.beta * old[ii][n][j];
              const bool keep = !drop || w[j] >= p.thresh;
              const float f = 1.f - ax[ii][n][j] * ax[ii][n][j];
              v *= keep ? f * p.drop_scale : 0.f;
              cp[(long)j * p.ldc] = v;
            }
          }
        }
        __asm__ volatile("" ::: "memory");
      }
    } else {
      const bool has_beta = p.beta != 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row4 = row0 + 16 * i;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
          float* cp = p.C + (long)row4 * p.ldc + col0 + 16 * n;
          float old[4] = {0.f, 0.f, 0.f, 0.f};
          if (has_beta) {
#pragma unroll
            for (int j = 0; j < 4; ++j) old[j] = cp[(long)j * p.ldc];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) cp[(long)j * p.ldc] = acc[i][n][j] + p.beta * old[j];
        }
      }
      if (db_i >= 0 && l15 == 0) {
        const int row4 = row0 + 16 * db_i;
#pragma unroll
        for (int j = 0; j < 4; ++j) p.dbias[row4 + j] = p.dbias_set ? accb[j] : p.dbias[row4 + j] + accb[j];
      }
    }
  };

  auto read_frags = [&](const unsigned char* sp, Frags& f) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      f.ah[i] = *reinterpret_cast<const bf16x8*>(sp + a_frag + i * 2048);
      f.al[i] = *reinterpret_cast<const bf16x8*>(sp + a_frag + i * 2048 + lo_d);
    }
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      f.bh[n] = *reinterpret_cast<const bf16x8*>(sp + b_frag + n * 2048);
      f.bl[n] = *reinterpret_cast<const bf16x8*>(sp + b_frag + n * 2048 + lo_d);
    }
  };

  // One k-step: the MFMAs of this step from the fragments in `cur` (read during the previous step), and -- spread over the rows
  // of MFMAs, in front of each row -- the reads of the NEXT step's fragments from the slot at `sn` into `nxt`.  That slot landed
  // before this step's barrier.  Behind the very last step the reads fetch a stale slot and are never used.
  auto step = [&](const unsigned char* sn, const Frags& cur, Frags& nxt) {
    p3_static_for<MI>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr ((ABL & 16) == 0) {
        nxt.ah[i] = *reinterpret_cast<const bf16x8*>(sn + a_frag + i * 2048);
        nxt.al[i] = *reinterpret_cast<const bf16x8*>(sn + a_frag + i * 2048 + lo_d);
        if (i < NI) {
          nxt.bh[i] = *reinterpret_cast<const bf16x8*>(sn + b_frag + i * 2048);
          nxt.bl[i] = *reinterpret_cast<const bf16x8*>(sn + b_frag + i * 2048 + lo_d);
        }
      } else {
        nxt.ah[i] = cur.ah[i]; nxt.al[i] = cur.al[i];
        if (i < NI) { nxt.bh[i] = cur.bh[i]; nxt.bl[i] = cur.bl[i]; }
      }
      if constexpr ((ABL & 2) == 0) {
        if constexpr (!ONE) {
#pragma unroll
          for (int n = 0; n < NI; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.al[i], cur.bh[n], acc[i][n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < NI; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.ah[i], cur.bl[n], acc[i][n], 0, 0, 0);
        } else {
          // (the "lo" halves of the images hold the hi halves of the second column block of this 64-wide step)
#pragma unroll
          for (int n = 0; n < NI; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.al[i], cur.bl[n], acc[i][n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < NI; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.ah[i], cur.bh[n], acc[i][n], 0, 0, 0);
        if constexpr (EPI == 2) {
          if (i == db_i) {
            accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.al[i], ones, accb, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.ah[i], ones, accb, 0, 0, 0);
          }
        }
      } else {
        acc[i][0][0] += (float)cur.ah[i][0] + (float)cur.al[i][0] + (float)cur.bh[i % NI][0] + (float)cur.bl[i % NI][0];     // (keeps the reads alive)
      }
      if constexpr (EPI != 2 && (ABL & 16) == 0) {
        // ONE read between two MFMAs (an MFMA holds the SIMD's issue port for half of its 16 cycles: a read in the other half is
        // free; the reads in a clump in front of the row drained the matrix pipe for ~50 cycles per row: 772 cycles per k-step for
        // 576 of MFMA, tools/micro/p3_bench.hip).  (The bias-gradient branch of EPI 2 splits the row into basic blocks: no groups.)
        constexpr int NR = i < NI ? 4 : 2, NM = ONE ? 2 * NI : 3 * NI, NP = NR < NM ? NR : NM;
        p3_static_for<NP>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
        if constexpr (NR > NP) __builtin_amdgcn_sched_group_barrier(0x100, NR - NP, 0);
        if constexpr (NM > NP) __builtin_amdgcn_sched_group_barrier(0x008, NM - NP, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // the step's tail: every read of this step has returned (the slot it read may be refilled behind the next barrier); at the
  // end of a tile, its epilogue and the next tile's bookkeeping
  auto tail = [&]() {
    __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0), as the builtin: the compiler's own wait tracking sees it
    if (++kt == nk) {
      epilogue();
      kt = 0; ++r_c;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int n = 0; n < NI; ++n) acc[i][n] = f32x4v{0.f, 0.f, 0.f, 0.f};
      accb = f32x4v{0.f, 0.f, 0.f, 0.f};
      if (r_c < nmine) {
        (void)p3_tile_of<T::BN>(g, b, G, r_c, pi, tm, tn);
        db_i = db_of(pi, tn);
      }
    }
  };

  __builtin_amdgcn_s_barrier();                                 // step 0 has landed
  read_frags(smem, f0);
  __builtin_amdgcn_s_waitcnt(0xc07f);                           // lgkmcnt(0): slot 0 is refilled behind the next barrier
  int nslot = 1;                                                // slot of the step after the current one
  for (int s = 0; s < total; s += 2) {
    long long c0 = 0;
    if constexpr ((ABL & 4) != 0) c0 = __builtin_readcyclecounter();
    if constexpr ((ABL & 32) == 0) __builtin_amdgcn_s_barrier();
    if constexpr ((ABL & 4) != 0) {
      // stamps of workgroup 0, compute wave 0: [s][0..2] = top of the step, barrier passed, MFMAs issued
      if (b == 0 && wave == 0 && lane == 0 && s < 128) {
        long long* st = reinterpret_cast<long long*>(g.p[0].slab) + 4 * s;
        st[0] = c0; st[1] = __builtin_readcyclecounter();
      }
    }
    step(smem + nslot * T::SLOT, f0, f1);
    nslot = nslot == NS - 1 ? 0 : nslot + 1;
    if constexpr ((ABL & 4) != 0) {
      if (b == 0 && wave == 0 && lane == 0 && s < 128) reinterpret_cast<long long*>(g.p[0].slab)[4 * s + 2] = __builtin_readcyclecounter();
    }
    tail();
    if (s + 1 < total) {
      if constexpr ((ABL & 32) == 0) __builtin_amdgcn_s_barrier();
      step(smem + nslot * T::SLOT, f1, f0);
      nslot = nslot == NS - 1 ? 0 : nslot + 1;
      tail();
    }
  }
}

}  // namespace lirec
