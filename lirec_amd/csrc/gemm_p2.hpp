// Layer-1 GEMM pair on pre-split bf16 operands, second generation (gfx950): K1 (H1 = X W1^T, mlp/model.py:281,286,291-292,
// 307,313,319-320) and K6 (dW1 = dZ1^T X, autograd of the same lines).
//
// What the round-2 counters said about the first planes kernel and the on-the-fly core: both move
// 1.3-1.8 GB from L2 to the CUs per launch (256x128 / 128x128 tiles) and the chip delivers ~10 TB/s of that; the epilogue
// of a one-workgroup-per-CU tile is exposed; and 256 fixed tiles of unequal depth (K = 768 / 2048) leave a quarter of the
// chip idle for half of the launch.  This kernel is built around those numbers:
//   * 256 x 256 x 32 tiles (0.9 GB of L2 -> LDS traffic for the bench shape), 8 waves as 2 x 4, each (16 MF) x 64 outputs
//     as MF x 4 tiles of v_mfma_f32_16x16x32_bf16 (the 16-row MFMA lets a tile be any multiple of 32 rows tall and holds a
//     higher clock than the 32x32x16 form under load, MI355X_MICROARCH.md DVFS item 7);
//   * the split-precision product of gemm_bf16x3.hpp -- hi*hi + hi*lo + lo*hi into one fp32 accumulator -- on operands
//     split ONCE.  The feature rows and the weights are kept in the "q32" form: the fp32 matrix's own footprint, every 32
//     consecutive elements of a row stored as 64 B of hi halves followed by 64 B of lo halves, so that one k-step of a row
//     is one whole 128-B line (separate hi / lo planes made every LDS-DMA request a 64-B half line: the forward's DMA ran at
//     6 TB/s).  dZ1 (written by the un-pool kernel) stays in two planes: its rows are read 256 B at a time;
//   * LDS = two rings, all 160 KiB: three 32-KiB slots for the operand whose fragments are streamed through the k-step (A),
//     two for the operand whose fragments are hoisted into registers at the start of the step (B).  A second barrier behind
//     the B reads releases B's slot early, so BOTH operands are requested two k-steps ahead (the chip-wide L2 -> LDS rate makes
//     a 64-KiB request per CU take ~1.7 us, more than one k-step of MFMAs);
//   * LDS-DMA as inline asm with hand-counted s_waitcnt vmcnt (see p2_dma16);
//   * a PERSISTENT launch of one workgroup per CU with a device-side 1-D partition of the work: every workgroup gets the
//     same number of (32-row block x k-step) units.  Forward (NT): the cut runs along the rows of each (head, segment)
//     problem -- whole k, so no partial sums -- and a workgroup's row range becomes one or more tiles of 32 MF rows;
//     weight gradient (TN): the cut runs along k (the rows that are reduced over), partial tiles go to slabs that a
//     reduce kernel sums in a fixed order (deterministic), tiles that fall into one workgroup's range whole are added
//     to dW1 directly.  The device-side row count of the compact context rows (GemmProblem::dyn) enters the partition on
//     the device: no host read-back.  The two column tiles of a problem (J = 512) are handled by adjacent workgroups of one
//     XCD over the same rows at the same time, so the feature rows come out of HBM once.
// LDS images (conflict-free by construction, checked with exact-integer operands in tools/micro/p2_bench.hip)
//   NT, both operands: [256 rows][128 B = 4 hi chunks | 4 lo chunks]; chunk c of row r at slot c ^ ((r >> 1) & 7): the 16
//       lanes a ds_read_b128 services together (rows lane & 15, chunk lane >> 4) hit 16 different 16-B slots of the 256-B bank row;
//   TN, B (feature rows, q32): [32 k][1 KiB]; chunk ch of row k at (ch & ~15) | ((ch & 15) ^ f(k)),
//       f(k) = ((k & 3) << 2) | ((k >> 2) & 3); TN, A (dZ1 planes): hi | lo images, each two sub-tiles of [32 k][128 cols] with
//       256-B rows, chunk ch of row k at ch ^ f(k): conflict-free ds_read_b64_tr_b16 (guide T10 (b)).
#pragma once
#include "gemm.hpp"
#include "gemm_bf16x3.hpp"
#include "p2_partition.hpp"

namespace lirec {

// One LDS-DMA wave-instruction (global_load_lds_dwordx4): lane l copies 16 bytes from sbase + voff (per lane) to LDS address
// lds + 16 l.  Written as inline asm on purpose: hipcc's waitcnt pass tracks the builtin form as a pending LDS write and,
// unable to tell the ring slot being filled from the one being read, drains it (s_waitcnt vmcnt(0)) in front of the first
// fragment read of every k-step; behind asm it knows nothing, and every wait of the k-loop is the counted one written below.
// (Hidden operations only make the compiler's own vmcnt waits -- epilogue loads -- stricter: the counter is in-order.)
__device__ __forceinline__ void p2_dma16(const void* sbase_, unsigned voff, unsigned lds_) {
  // (the base and the LDS address are wave-uniform by construction; said explicitly -- hipcc's uniformity analysis loses
  //  them behind the divergent branches of the in-loop dropout tasks and would hand the asm a VGPR)
  const unsigned long sb = (unsigned long)sbase_;
  const unsigned sb_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(sb >> 32)), sb_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb);
  const void* sbase = (const void*)(((unsigned long)sb_hi << 32) | (unsigned long)sb_lo);
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ unsigned p2_lds_addr(const unsigned char* p) {
  return (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)p;
}
// the same with the non-temporal cache policy (streamed-once operands)
__device__ __forceinline__ void p2_dma16_nt(const void* sbase_, unsigned voff, unsigned lds_) {
  const unsigned long sb = (unsigned long)sbase_;
  const unsigned sb_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(sb >> 32)), sb_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb);
  const void* sbase = (const void*)(((unsigned long)sb_hi << 32) | (unsigned long)sb_lo);
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
// the same with a per-lane 64-bit source address (rows gathered through an index: GemmProblem::srow)
__device__ __forceinline__ void p2_dma16_v(const unsigned char* addr, unsigned lds_) {
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(addr), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void p2_dma16_v_nt(const unsigned char* addr, unsigned lds_) {
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(addr), "s"(lds) : "memory", "m0");
}
// byte offset of storage row s in a q32b matrix of `ld` columns (relative to its first column block)
__device__ __forceinline__ long p2_row_off(int s, long ld) { return ((long)(s >> 5) * (ld >> 5)) * 4096 + (long)(s & 31) * 128; }
// (the same in q16b storage: 2-KiB blocks, 64-byte rows)
__device__ __forceinline__ long p2_row_off16(int s, long ld) { return ((long)(s >> 5) * (ld >> 5)) * 2048 + (long)(s & 31) * 64; }
// (q16c storage: 4-KiB blocks of 64 columns, 128-byte rows)
__device__ __forceinline__ long p2_row_off16c(int s, long ld) { return ((long)(s >> 5) * (ld >> 6)) * 4096 + (long)(s & 31) * 128; }
typedef int i32x4v __attribute__((ext_vector_type(4)));
// four consecutive ints at a wave-uniform address, through the scalar cache (the list was written by an earlier launch)
__device__ __forceinline__ i32x4v p2_sload4(const int* p) {
  return *reinterpret_cast<const __attribute__((address_space(4))) i32x4v*>((unsigned long)p);
}
template <int N> __device__ __forceinline__ void p2_wait_vm() {
  static_assert(N >= 0 && N <= 8, "counts used by the k-loops");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
__device__ __forceinline__ void p2_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct P2 {
  static constexpr int BM = 256, BN = 256, BK = 32, NTHR = 512;
  static constexpr int SLOT = 32768;            // one operand, one k-step (hi + lo)
  static constexpr int IMG = 16384;             // TN A: one plane of the slot
  static constexpr int A0 = 0, B0 = 3 * SLOT, LDS_BYTES = 5 * SLOT;
  static constexpr int SLAB = 256 * 256;        // floats of one partial tile
};

// logical workgroup id: the G / 8 workgroups an XCD hosts get consecutive ids (speed only)
__device__ __forceinline__ int p2_logical_id() {
  const int G = gridDim.x, b = blockIdx.x;
  return (G & 7) == 0 ? (b & 7) * (G >> 3) + (b >> 3) : b;
}
// (r T < 2^32 for every shape the host admits: r <= a few hundred workgroups, T = k-steps of all tiles; 32-bit on purpose --
//  the reduce kernel evaluates this a dozen times per workgroup and a 64-bit division is a ~100-instruction sequence)
__device__ __forceinline__ long p2_cut(long r, long T, long Gr) { return (long)((unsigned)r * (unsigned)T / (unsigned)Gr); }

typedef float f32x4v __attribute__((ext_vector_type(4)));

// (q32b, the storage of the feature rows and the first-layer weights for these kernels: p2_store_q32b in gemm_bf16x3.hpp)

// -----------------------------------------------------------------------------------------------------------------
// forward: one tile of 32 MF rows x 256 columns, all of k.  p.A / p.B: q32b matrices at the segment's first column block
// (byte pointers; lda / ldb = columns of the whole matrix, i.e. 32 x its column blocks).
// -----------------------------------------------------------------------------------------------------------------
// XP = planes of the row operand: 2 = q32b (hi and lo halves), 1 = q16b (rows STORED as bf16: the stored value is the hi half, there
// is no lo half -- gathered rows only): A image rows of 64 bytes, two requests per 32-row block and k-step, one fragment read and
// TWO MFMAs per product (hi x lo(W), hi x hi(W)).
// ONE (with XP = 1): gemm mode 3, BASELINE config 5's arithmetic -- one MFMA per product (bf16(x) bf16(w), fp32 accumulate).
// XF (with GATHER, XP = 2): the rows are fetched STRAIGHT FROM THE fp32 BLOCK (p.A = the block at the segment's first column, lda = its
// row pitch in elements, srow = physical rows): a k-step of a row is the same 128 B either way, so the LDS-DMA requests and the A
// image keep their shape -- the image just arrives holding 32 floats per row instead of 32 hi | 32 lo halves -- and the loader wave
// that requested a row block turns it into the q32b image IN PLACE (two ds_read_b128, split4 x 2, two ds_write_b128 per 8 floats;
// every read of a row is issued before the first write to it, LDS executes a wave's instructions in order) once its requests have
// landed, in the half-step it otherwise spends waiting for the other group's MFMAs.  The multiply path does not change, the
// results are the staged path's bit for bit.  The same registers go out to p.xq_out (when given) as the q32b rows the weight
// gradient reads later -- the copy the row staging pass used to make; `emit` = which of the wave's two 16-row halves of a block
// THIS workgroup stores (the column-tile workgroups of a row range share the job).
// MEASURED, NOT USED BY THE LIBRARY (tools/micro/p2x_bench.hip, profiles/r05_p2x_bench.txt; the library instantiates no XF kernel):
// bit-identical, and 194 us (205 with the rows written out) against the staged form's 153 -- the loader group's half-step grows
// from ~740 to ~3100 cycles (requests 520 | wait for rows that now arrive with a DRAM-page miss each, one k-step of ring depth too
// few: 700-1300 | conversion reads 300 | split + ds_write_b128 + stores 1100-1300 | fragment reads 230) and the other group's
// 1700 cycles of MFMAs no longer cover it.  What it would save is the staging pass's row copy (~70 of its 100 us): not enough.
// The same arithmetic rules out building the dZ1 operand of the weight gradient in registers (p2_tn_piece's A image is the same 32
// KiB per k-step through the same ds_write path, plus the loads of dHbar and the sign bits in front of it).
template <int MF, int ABL, bool GATHER = false, int XP = 2, bool ONE = false, bool XF = false, bool K64 = false>
__device__ __forceinline__ void p2_nt_tile(const GemmProblem& p, unsigned char* smem, int row0_, int Mvalid, int ct_,
                                           int lane, int wave, int ablate, int emit = 0) {
  static_assert(XP == 2 || (XP == 1 && GATHER), "one-plane rows are gathered from q16b storage");
  static_assert(!ONE || XP == 1, "the single-pass mode runs on the one-plane form");
  static_assert(!XF || (GATHER && XP == 2), "fp32 rows are fetched through a row list and become the two-plane image");
  static_assert(!K64 || (XP == 2 && !ONE && !XF), "64 of k per step: the two-plane kernel's images, the halves holding k 0-31 | k 32-63");
  // (wave-uniform by construction; said explicitly so that the LDS-DMA base addresses are SGPR pairs)
  const int row0 = __builtin_amdgcn_readfirstlane(row0_), ct = __builtin_amdgcn_readfirstlane(ct_);
  const int wr = wave >> 2, wc = wave & 3, g = lane >> 4, l15 = lane & 15;
  const int nk = p.K >> 5;
  // ---- Two wave groups half a k-step apart (the two waves of every SIMD belong to different groups):
  //   group X = waves 0-3: tile rows [0, 16 MF), and the LOADER of the B operand (wave j: image rows [64 j, 64 j + 64));
  //   group Y = waves 4-7: tile rows [16 MF, 32 MF), and the loader of the A operand (wave j: row blocks j and j + 4).
  // A k-step is two half-steps with one barrier in front of each.  In the EVEN half X multiplies step t (its B fragments and
  // first A fragment already in registers) while Y requests A(t + 2), reads the B fragments of step t and its own first A
  // fragment and waits for A(t + 1); in the ODD half Y multiplies step t while X requests B(t + 2) and reads its fragments of
  // step t + 1.  The matrix pipe of a SIMD always has ONE wave feeding it: what used to be exposed between two k-steps -- the
  // counted wait, the barrier, the fragment reads, their latency, the second barrier: ~990 of 4140 cycles per step with both
  // waves of a SIMD in step (tools/micro/p2_bench.hip, stamps) -- and every LDS-DMA issue stall (60-180 cycles each) now sit
  // in the shadow of the other group's MFMAs.  Slots as before: A three, B two.  A(s) is in use from the barrier in front of the
  // odd half s - 1 (X reads its first fragment) to the one in front of the even half s + 1: A(t + 2) may be requested in the
  // even half t and has a k-step to land; B(s) from the odd half s - 1 to the end of the even half s: B(t + 2) is requested in
  // the odd half t.  Same products in the same order per output element: the results are bit-identical to the in-step form.
  const int role = wave >> 2, wj = wave & 3;
  unsigned off2[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = 8 * q + (lane >> 3);
    const int sc = (lane & 7) ^ ((r >> 1) & 7);               // source chunk of LDS chunk lane & 7 (rows + 16: same swizzle)
    off2[q] = (unsigned)r * 128u + 16u * sc;
  }
  const unsigned lds0 = p2_lds_addr(smem);
  // this wave's two row blocks of the operand it loads: b = wj, wj + 4 of A (those below MF); 2 wj, 2 wj + 1 of B
  const int blk0 = role == 0 ? 2 * wj : wj, blk1 = role == 0 ? 2 * wj + 1 : wj + 4;
  const bool has0 = role == 0 || blk0 < MF, has1 = role == 0 || blk1 < MF;
  const int nreq = ((role == 1 && XP == 1) ? 2 : 4) * ((has0 ? 1 : 0) + (has1 ? 1 : 0));     // requests per k-step of this wave
  const unsigned char* src0 = role == 0 ? reinterpret_cast<const unsigned char*>(p.B) + (long)(8 * ct + blk0) * (p.ldb >> 5) * 4096
                                        : reinterpret_cast<const unsigned char*>(p.A) + (long)((row0 >> 5) + blk0) * (p.lda >> 5) * 4096;
  const unsigned char* src1 = role == 0 ? reinterpret_cast<const unsigned char*>(p.B) + (long)(8 * ct + blk1) * (p.ldb >> 5) * 4096
                                        : reinterpret_cast<const unsigned char*>(p.A) + (long)((row0 >> 5) + blk1) * (p.lda >> 5) * 4096;
  constexpr int ARB = XP == 1 ? 64 : 128;                     // bytes of an A image row
  const unsigned dst0 = lds0 + (role == 0 ? P2::B0 + (32 * blk0) * 128 : P2::A0 + (32 * blk0) * ARB);
  const unsigned dst1 = lds0 + (role == 0 ? P2::B0 + (32 * blk1) * 128 : P2::A0 + (32 * blk1) * ARB);
  // GATHER: the A rows come straight from a q32b matrix through GemmProblem::srow -- this lane's four image rows of each block
  // (8 q + lane / 8) as byte addresses of their k-step-0 chunk; a k-step further is one 4-KiB column block further
  const unsigned char* arow[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
  unsigned aoff[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  if constexpr (GATHER) {
    if (role == 1) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int b = u == 0 ? blk0 : blk1;
        if (b < MF) {
          if constexpr (XP == 1) {
            // q16b: a request = 16 image rows x 64 B; lane -> row 16 q + lane / 4, LDS chunk lane & 3 <- source chunk ^ f(row)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int img = 16 * q + (lane >> 2);
              const int sidx = p.srow[row0 + 32 * b + img];
              const int sc = (lane & 3) ^ ((img >> 1) & 3);
              arow[u][q] = reinterpret_cast<const unsigned char*>(p.A) + p2_row_off16(sidx, p.lda) + 16 * sc;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int img = 8 * q + (lane >> 3);
              const int sidx = p.srow[row0 + 32 * b + img];
              const int sc = (lane & 7) ^ ((img >> 1) & 7);
              // (XF: a 32-bit byte offset into the block -- the base and the k-step stay on the scalar side)
              if constexpr (XF) aoff[u][q] = (unsigned)sidx * (unsigned)(p.lda * 4) + 16u * sc;
              else arow[u][q] = reinterpret_cast<const unsigned char*>(p.A) + p2_row_off(sidx, p.lda) + 16 * sc;
            }
          }
        }
      }
    }
  }
  // the requests of k-step t of this wave's operand into slot `slot` of that operand's ring
  auto issue = [&](int t, int slot) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!(u == 0 ? has0 : has1)) continue;
      const unsigned char* sb = u == 0 ? src0 : src1;
      const unsigned db = (u == 0 ? dst0 : dst1) + slot * P2::SLOT;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (role == 0) {
          p2_dma16(sb + 4096 * t + (q >> 1) * 2048, off2[q & 1], db + q * 1024);
        } else {
          if constexpr (XP == 1) {
            if (q < 2) p2_dma16_v(arow[u][q] + 2048L * t, db + q * 1024);
          } else if constexpr (GATHER) {
            // (default cache policy: a piece row is shared by many logical rows -- it should stay in L2 / the Infinity Cache)
            // (XF: a row of the block is read by this launch's column-tile workgroups at about the same time and not again)
            if constexpr (XF) p2_dma16_nt(reinterpret_cast<const unsigned char*>(p.A) + 128L * t, aoff[u][q], db + q * 1024);
            else p2_dma16_v(arow[u][q] + 4096L * t, db + q * 1024);
          } else if constexpr ((ABL & 2048) == 0) {
            // (the feature rows are streamed: each line is used by this launch's two column-tile workgroups at about the same
            //  time and never again -- non-temporal policy, 152 vs 158 us in interleaved rounds; diagnostics bit 2048: off)
            p2_dma16_nt(sb + 4096 * t + (q >> 1) * 2048, off2[q & 1], db + q * 1024);
          } else {
            p2_dma16(sb + 4096 * t + (q >> 1) * 2048, off2[q & 1], db + q * 1024);
          }
        }
      }
    }
  };
  // all but this wave's newest `n` requests have landed (n = 0, 4, 8)
  auto wait_but = [&](int n) { if (n >= 8) p2_wait_vm<8>(); else if (n >= 4) p2_wait_vm<4>(); else if (n >= 2) p2_wait_vm<2>(); else p2_wait_vm<0>(); };
  // ---- XF: this loader wave's row blocks of k-step t, in A slot `slot`: 32 floats per row -> 32 hi | 32 lo halves, in place.
  // Lane -> (row, c): c = lane & 3 = the row's floats 8 c .. 8 c + 7 (source chunks 2 c, 2 c + 1 -> chunks c and 4 + c).
  // (two halves: the reads are issued BEFORE the k-step's LDS-DMA requests -- whose issue stalls cover the LDS round trip -- and
  //  the split, the writes and the stores behind them)
  f32x4 cva[2][2], cvb[2][2];
  // (j = lane / 4 -> row: swizzle class bit 2 = j bit 0, so the two rows of a ds_write_b128 lane group (8 contiguous lanes, banks
  //  mod 128 B) write disjoint halves of the bank row; row parity = j bit 1 and class bit 0 = j bit 2, so the four rows of a
  //  ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...: j of even / odd bit parity) cover both 128-B halves x both chunk parities)
  const int cj = lane >> 2, cc = lane & 3;
  const int csw = ((cj & 1) << 2) | (((cj >> 3) & 1) << 1) | ((cj >> 2) & 1);
  const int cr16 = 2 * csw + ((cj >> 1) & 1);
  auto convert_read = [&](int slot) {
    if constexpr (XF) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!(u == 0 ? has0 : has1)) continue;
        const unsigned char* rp = smem + P2::A0 + slot * P2::SLOT + (32 * (u == 0 ? blk0 : blk1) + cr16) * 128;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          cva[u][q] = *reinterpret_cast<const f32x4*>(rp + q * 2048 + (((2 * cc) ^ csw) << 4));
          cvb[u][q] = *reinterpret_cast<const f32x4*>(rp + q * 2048 + (((2 * cc + 1) ^ csw) << 4));
        }
      }
    }
  };
  auto convert_write = [&](int t, int slot) {
    if constexpr (XF) {
      p2_wait_lgkm0();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!(u == 0 ? has0 : has1)) continue;
        const int b = u == 0 ? blk0 : blk1;
        unsigned char* rp = smem + P2::A0 + slot * P2::SLOT + (32 * b + cr16) * 128;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          uint2 h0, l0, h1, l1;
          split4(cva[u][q], h0, l0);
          split4(cvb[u][q], h1, l1);
          const uint4 hi = make_uint4(h0.x, h0.y, h1.x, h1.y), lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
          *reinterpret_cast<uint4*>(rp + q * 2048 + ((cc ^ csw) << 4)) = hi;
          *reinterpret_cast<uint4*>(rp + q * 2048 + (((4 + cc) ^ csw) << 4)) = lo;
          if ((emit >> q) & 1) {
            // (the matrix is < 4 GiB: a 32-bit lane offset, the k-step on the scalar side)
            const int R = row0 + 32 * b + 16 * q + cr16;
            const unsigned eo = ((unsigned)(R >> 5) * (unsigned)(p.ld_xq >> 5)) * 4096u + (unsigned)(R & 31) * 128u + (unsigned)cc * 16u;
            unsigned char* o = p.xq_out + 4096L * t + eo;
            *reinterpret_cast<uint4*>(o) = hi;
            *reinterpret_cast<uint4*>(o + 64) = lo;
          }
        }
      }
    }
  };
  // ---- fragment addresses: row l15 of the fragment, chunk g (hi) / g + 4 (lo = hi address ^ 64) ----------------------------
  const int frag = l15 * 128 + ((g ^ ((l15 >> 1) & 7)) << 4);
  const int lo_d = 64 - 2 * (frag & 64);                     // lo address = hi address ^ 64
  // (one-plane A image: 64-byte rows, four chunks, swizzle (row >> 1) & 3; a fragment = 1 KiB)
  constexpr int AFB = XP == 1 ? 1024 : 2048;                  // bytes of one 16-row A fragment in the image
  const int a_frag = P2::A0 + (XP == 1 ? l15 * 64 + ((g ^ ((l15 >> 1) & 3)) << 4) : frag) + (wr * MF) * AFB;
  const int b_frag = P2::B0 + frag + (wc * 4) * 2048;

  f32x4v acc[MF][4];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = f32x4v{0.f, 0.f, 0.f, 0.f};

  if (!(ablate & 4)) {
    constexpr bool di = !(ABL & 16), dc = !(ABL & 32);        // diagnostics builds: 16 = no LDS-DMA, 32 = no reads / MFMAs
    bf16x8 bh[4], bl[4], ah, al;
    // the B fragments of the step in B slot `bs` and this wave's first A fragment of the step in A slot `as`
    auto read_frags = [&](int as, int bs) {
      const unsigned char* ap = smem + a_frag + as * P2::SLOT;
      const unsigned char* bp = smem + b_frag + bs * P2::SLOT;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        bh[n] = *reinterpret_cast<const bf16x8*>(bp + n * 2048);
        bl[n] = *reinterpret_cast<const bf16x8*>(bp + n * 2048 + lo_d);
      }
      ah = *reinterpret_cast<const bf16x8*>(ap);
      if constexpr (XP == 2) al = *reinterpret_cast<const bf16x8*>(ap + lo_d);
      p2_wait_lgkm0();
    };
    // this wave's MFMAs of one k-step: A slot `as`, the fragments above in registers
    auto multiply = [&](int as) {
      const unsigned char* ap = smem + a_frag + as * P2::SLOT;
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        bf16x8 ah_n, al_n;
        if (i + 1 < MF) {
          ah_n = *reinterpret_cast<const bf16x8*>(ap + (i + 1) * AFB);
          if constexpr (XP == 2) al_n = *reinterpret_cast<const bf16x8*>(ap + (i + 1) * AFB + lo_d);
        }
        if constexpr (K64) {
          // (the "lo" halves of both images hold k 32-63 of the step: one MFMA per product, two k-blocks per step)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[n], acc[i][n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl[n], acc[i][n], 0, 0, 0);
        } else {
        if constexpr (XP == 2) {
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[n], acc[i][n], 0, 0, 0);
        }
        if constexpr (!ONE) {
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[n], acc[i][n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[n], acc[i][n], 0, 0, 0);
        }
        if (i + 1 < MF) { ah = ah_n; if constexpr (XP == 2) al = al_n; }
        // issue order inside the group: the NEXT fragment's reads in front of this one's MFMAs (left alone hipcc sinks every
        // read to just before its first use and waits lgkmcnt(0) there); nothing crosses the group's end
        if (i + 1 < MF) __builtin_amdgcn_sched_group_barrier(0x100, XP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, ONE ? 4 : (K64 ? 8 : 4 * (XP + 1)), 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    long long* stamp = nullptr;
    if constexpr ((ABL & 1024) != 0) {
      if (lane == 0) stamp = reinterpret_cast<long long*>(p.slab) + (long)(blockIdx.x * 8 + wave) * 512;
      if (stamp) stamp[500] = __builtin_readcyclecounter();     // tile entered (behind the partition decode)
    }
    // prologue: steps 0 and 1 of both operands; then X takes its fragments of step 0
    if (di) {
      issue(0, 0);
      if (nk > 1) issue(1, 1);
    }
    if (nk > 1) wait_but(nreq); else p2_wait_vm<0>();
    if constexpr (!XF) __builtin_amdgcn_s_barrier();       // (XF: each group's branch below has it -- the loader group converts step 0 first)
    if constexpr ((ABL & 1024) != 0) { if (stamp) stamp[501] = __builtin_readcyclecounter(); }      // step 0 has landed
    if constexpr (!XF) { if (role == 0 && dc) read_frags(0, 0); }
    // (two loops, one per group, each straight-line: with the groups' halves as branches of ONE loop body hipcc kept a second set
    //  of the forty fragment registers for the merge and the 256-row tile spilled)
    int as = 0;                                              // A slot of the step this wave multiplies next
    if (role == 0) {
      if constexpr (XF) { __builtin_amdgcn_s_barrier(); if (dc) read_frags(0, 0); }
      for (int t = 0; t < nk; ++t) {
        if constexpr ((ABL & 1024) != 0) { if (stamp && t < 64) stamp[6 * t] = __builtin_readcyclecounter(); }
        __builtin_amdgcn_s_barrier();                        // ---- even half: X multiplies
        if constexpr ((ABL & 1024) != 0) { if (stamp && t < 64) stamp[6 * t + 1] = __builtin_readcyclecounter(); }
        if (dc) multiply(as);
        p2_wait_vm<0>();                                     // B(t + 1), requested a k-step ago, has landed
        if constexpr ((ABL & 1024) != 0) { if (stamp && t < 64) stamp[6 * t + 2] = __builtin_readcyclecounter(); }
        __builtin_amdgcn_s_barrier();                        // ---- odd half: X loads and reads
        if constexpr ((ABL & 1024) != 0) { if (stamp && t < 64) stamp[6 * t + 3] = __builtin_readcyclecounter(); }
        if (di && t + 2 < nk) issue(t + 2, t & 1);           // every wave of Y has its B fragments of step t: the slot is free
        as = as == 2 ? 0 : as + 1;
        if (dc) read_frags(as, (t + 1) & 1);                 // (behind the last step: a stale slot, never used)
        if constexpr ((ABL & 1024) != 0) { if (stamp && t < 64) stamp[6 * t + 4] = __builtin_readcyclecounter(); }
      }
    } else {
      if constexpr (XF) { if (di) { convert_read(0); convert_write(0, 0); p2_wait_lgkm0(); } __builtin_amdgcn_s_barrier(); }
      for (int t = 0; t < nk; ++t) {
        if constexpr ((ABL & 1024) != 0) { if (stamp && t < 64) stamp[6 * t] = __builtin_readcyclecounter(); }
        __builtin_amdgcn_s_barrier();                        // ---- even half: Y loads and reads
        if constexpr ((ABL & 1024) != 0) { if (stamp && t < 64) stamp[6 * t + 1] = __builtin_readcyclecounter(); }
        int issued = 0;
        if constexpr (!XF) { if (di && t + 2 < nk) { issue(t + 2, as == 0 ? 2 : as - 1); issued = nreq; } }      // slot (t + 2) % 3: A(t - 1) is spent
        else issued = (di && t + 2 < nk) ? nreq : 0;
        if constexpr (XF) {
          // (the conversion first: the fragment registers of the step just multiplied are dead here, those of the next not yet
          //  loaded.  Two k-steps of requests stay in flight, as in the staged form: the block's rows arrive with HBM's latency)
          if (issued) issue(t + 2, as == 0 ? 2 : as - 1);
          if constexpr ((ABL & 1024) != 0) { if (stamp && t < 16) stamp[384 + 4 * t] = __builtin_readcyclecounter(); }
          wait_but(issued);                                    // A(t + 1) has landed
          if constexpr ((ABL & 1024) != 0) { if (stamp && t < 16) stamp[385 + 4 * t] = __builtin_readcyclecounter(); }
          const bool cv = di && t + 1 < nk;
          const int ns = as == 2 ? 0 : as + 1;
          if (cv) { convert_read(ns); p2_wait_lgkm0(); }
          if constexpr ((ABL & 1024) != 0) { if (stamp && t < 16) stamp[386 + 4 * t] = __builtin_readcyclecounter(); }
          if (cv) { convert_write(t + 1, ns); p2_wait_lgkm0(); }      // (visible to the other group behind the odd barrier)
          if constexpr ((ABL & 1024) != 0) { if (stamp && t < 16) stamp[387 + 4 * t] = __builtin_readcyclecounter(); }
          __builtin_amdgcn_sched_barrier(0);
          if (dc) read_frags(as, t & 1);                       // (its closing wait covers the conversion's writes)
        } else {
          if (dc) read_frags(as, t & 1);
          wait_but(issued);                                    // A(t + 1) has landed
        }
        if constexpr ((ABL & 1024) != 0) { if (stamp && t < 64) stamp[6 * t + 2] = __builtin_readcyclecounter(); }
        __builtin_amdgcn_s_barrier();                        // ---- odd half: Y multiplies
        if constexpr ((ABL & 1024) != 0) { if (stamp && t < 64) stamp[6 * t + 3] = __builtin_readcyclecounter(); }
        if (dc) multiply(as);
        as = as == 2 ? 0 : as + 1;
        if constexpr ((ABL & 1024) != 0) { if (stamp && t < 64) stamp[6 * t + 4] = __builtin_readcyclecounter(); }
      }
    }
    __builtin_amdgcn_s_barrier();               // the next tile's first requests overwrite the slots
    if constexpr ((ABL & 1024) != 0) { if (stamp) stamp[502] = __builtin_readcyclecounter(); }      // k loop done
  }

  // ---- epilogue: bias, relu, dropout, store ----------------------------------------------------------------------------
  // Dropout: p.aux, when given, holds the KEEP BITS of the launch -- one byte per (four consecutive rows, column), bit j =
  // row 4 q + j is kept, pitch p.ldaux, column index = drop_col_off + col -- written by the row-staging pass
  // (stage_rows_q32b_kernel): with one workgroup per CU nothing hides this epilogue, and the Philox calls (one per four
  // outputs, original row ids through the row map) were 60 us of the launch when they sat here.  (Producing them inside the
  // k-loop, in the shadow of the matrix pipe, was tried twice: hipcc answers the extra live ranges with 2 KB of scratch per
  // lane.)  Without the bytes the words are computed here (any caller, same results).
  const bool drop = p.thresh != 0u;
  const unsigned char* keep = reinterpret_cast<const unsigned char*>(p.aux);
  unsigned key_lo = p.seed_lo, key_hi = p.seed_hi;
  if (drop && !keep) apply_seed_offset(key_lo, key_hi, p.seed_dev);
  const bool mapped = drop && !keep && p.rowmap != nullptr;
  float bias_n[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) bias_n[n] = p.bias ? p.bias[256 * ct + 64 * wc + 16 * n + l15] : 0.f;
  if (drop && keep) {
    // (r6) The production path -- keep bytes given -- as a loop of its own: ALL keep bytes of the wave's MF row fragments are
    // requested first (4 MF byte loads, no branch between them), then the fragments are stored.  In the shared loop below hipcc
    // kept each fragment's four loads in front of that fragment's stores: MF exposed round trips of ~1.5 k cycles each, most of
    // the tile's 11-13 k cycle epilogue (the stamps of tools/micro/p2_bench.hip); one remains.
    const unsigned char* kp = keep + p.drop_col_off + 256 * ct + 64 * wc + l15;
    const int qlast = (Mvalid - 1) >> 2;
    unsigned kbs[MF][4];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      int q = (row0 + (wr * MF + i) * 16 + 4 * g) >> 2;
      q = q < qlast ? q : qlast;                              // (a fragment beyond the valid rows: a duplicate, never used)
#pragma unroll
      for (int n = 0; n < 4; ++n) kbs[i][n] = kp[(long)q * p.ldaux + 16 * n];
    }
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int row4 = row0 + (wr * MF + i) * 16 + 4 * g;
      if (row4 >= Mvalid) continue;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float* cp = p.C + (long)row4 * p.ldc + 256 * ct + 64 * wc + 16 * n + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = relu_f(acc[i][n][j] + bias_n[n]);
          v = ((kbs[i][n] >> j) & 1u) ? v * p.drop_scale : 0.f;
          if (row4 + j < Mvalid) cp[(long)j * p.ldc] = v;
        }
      }
    }
  } else
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int row4 = row0 + (wr * MF + i) * 16 + 4 * g;
    if (row4 >= Mvalid) continue;
    unsigned kb[4] = {15u, 15u, 15u, 15u};
    if (drop && keep) {
#pragma unroll
      for (int n = 0; n < 4; ++n) kb[n] = keep[(long)(row4 >> 2) * p.ldaux + p.drop_col_off + 256 * ct + 64 * wc + 16 * n + l15];
    }
    unsigned rid[4] = {0u, 0u, 0u, 0u};
    if (mapped) {
#pragma unroll
      for (int j = 0; j < 4; ++j) rid[j] = (unsigned)p.rowmap[row4 + j < Mvalid ? row4 + j : Mvalid - 1];
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int col = 256 * ct + 64 * wc + 16 * n + l15;
      if (drop && !keep) {
        unsigned w[4];
        if (!mapped) {
          philox4((unsigned)(p.drop_col_off + col), (unsigned)(row4 >> 2), p.site, 0u, key_lo, key_hi, w);
        } else {
          unsigned rnd[4];
          unsigned blk = rid[0] >> 2;
          philox4((unsigned)(p.drop_col_off + col), blk, p.site, 0u, key_lo, key_hi, rnd);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if ((rid[j] >> 2) != blk) {
              blk = rid[j] >> 2;
              philox4((unsigned)(p.drop_col_off + col), blk, p.site, 0u, key_lo, key_hi, rnd);
            }
            const unsigned k = rid[j] & 3u;
            w[j] = k == 0u ? rnd[0] : (k == 1u ? rnd[1] : (k == 2u ? rnd[2] : rnd[3]));
          }
        }
        kb[n] = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) kb[n] |= (w[j] >= p.thresh ? 1u : 0u) << j;
      }
      float* cp = p.C + (long)row4 * p.ldc + col;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = relu_f(acc[i][n][j] + bias_n[n]);
        if (drop) v = ((kb[n] >> j) & 1u) ? v * p.drop_scale : 0.f;
        if (row4 + j < Mvalid) cp[(long)j * p.ldc] = v;
      }
    }
  }
  if constexpr ((ABL & 1024) != 0) {
    if (lane == 0 && !(ablate & 4)) (reinterpret_cast<long long*>(p.slab) + (long)(blockIdx.x * 8 + wave) * 512)[503] = __builtin_readcyclecounter();
  }
}

// -----------------------------------------------------------------------------------------------------------------
// data gradient through a wide weight matrix (the gate, dEE = dZg Wg: mlp/model.py:349-354 backward): one tile of 32 MF rows x
// 256 columns, all of k.  p.A: the rows (dZg), q32b, as in the forward tile; p.B: the weights as they are stored for the
// FORWARD -- q32b [k][columns], the reduced index is the ROW index -- at the problem's first column block (ldb = columns of the
// whole matrix): the B image and its fragments are the weight-gradient kernel's (k-major rows, transposed LDS reads), so ONE
// staged copy of Wg serves the forward and this launch.  Epilogue: tanh-dropout backward (EPI_TANH_BWD of gemm.hpp):
//   C = (acc + beta C) * keep / (1 - p) * (1 - aux^2),  keep from Philox (site, drop_col_off + col, row).
// -----------------------------------------------------------------------------------------------------------------
template <int MF, int ABL>
__device__ __forceinline__ void p2_nn_tile(const GemmProblem& p, unsigned char* smem, int row0_, int Mvalid, int ct_,
                                           int lane, int wave, int ablate) {
  const int row0 = __builtin_amdgcn_readfirstlane(row0_), ct = __builtin_amdgcn_readfirstlane(ct_);
  const int wr = wave >> 2, wc = wave & 3, g = lane >> 4, l15 = lane & 15;
  const int nk = p.K >> 5;
  // ---- LDS-DMA.  A: as the forward tile (wave w fills image rows [32 w, 32 w + 32)).  B: wave w fills k-rows 4 w + q, one
  // instruction (1 KiB = 256 columns) each; LDS chunk `lane` of row k <- source chunk lane ^ f(k) (gemm_p2_tn's B image)
  unsigned off2[2], b_off[4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = 8 * q + (lane >> 3);
    const int sc = (lane & 7) ^ ((r >> 1) & 7);
    off2[q] = (unsigned)r * 128u + 16u * sc;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned sc = (unsigned)(lane ^ ((q << 2) | (wave & 3)));
    b_off[q] = (sc >> 3) * 4096u + (unsigned)(4 * wave + q) * 128u + (sc & 7u) * 16u;
  }
  const unsigned char* a_base = reinterpret_cast<const unsigned char*>(p.A) + (long)((row0 >> 5) + wave) * (p.lda >> 5) * 4096;
  const unsigned char* b_base = reinterpret_cast<const unsigned char*>(p.B) + 8L * 4096 * ct;
  const long b_step = (long)(p.ldb >> 5) * 4096;
  const unsigned lds0 = p2_lds_addr(smem);
  const unsigned dstw = lds0 + (32 * wave) * 128;
  const unsigned b_dst = lds0 + P2::B0 + (4 * wave) * 1024;
  const bool load_a = wave < MF;
  auto issue_one = [&](int j, int t, int aslot, int bslot) {
    const int q = j & 3;
    if (j < 4) {
      p2_dma16(b_base + (long)t * b_step, b_off[q], b_dst + bslot * P2::SLOT + q * 1024);
    } else if (load_a) {
      p2_dma16(a_base + 4096 * t + (q >> 1) * 2048, off2[q & 1], dstw + P2::A0 + aslot * P2::SLOT + q * 1024);
    }
  };
  auto issue_all = [&](int t, int aslot, int bslot) {
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_one(j, t, aslot, bslot);
  };
  const int frag = l15 * 128 + ((g ^ ((l15 >> 1) & 7)) << 4);
  const int lo_d = 64 - 2 * (frag & 64);
  const int a_frag = P2::A0 + frag + (wr * MF) * 2048;
  // transposed B fragment reads (gemm_p2_tn's): two per fragment, rows 8 g + 4 t + q4
  const int q4 = l15 >> 2, pp = lane & 3;
  int tb[2], tx[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int k = 8 * g + 4 * t + q4;
    const int f = (q4 << 2) | ((2 * g + t) & 3);
    tb[t] = P2::B0 + wc * 256 + k * 1024 + 8 * (pp & 1);
    tx[t] = ((pp >> 1) ^ f) << 4;
  }
  auto frag2 = [&](const unsigned char* p0, const unsigned char* p1, int cb) -> bf16x8 {
    const s16x4 x = lds_tr16(p0 + ((cb << 4) ^ tx[0]));
    const s16x4 y = lds_tr16(p1 + ((cb << 4) ^ tx[1]));
    const s16x8 v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    return *reinterpret_cast<const bf16x8*>(&v);
  };

  f32x4v acc[MF][4];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = f32x4v{0.f, 0.f, 0.f, 0.f};

  if (!(ablate & 4)) {
    issue_all(0, 0, 0);
    if (nk > 1) issue_all(1, 1, 1);
    int as = 0;
    for (int t = 0; t < nk; ++t) {
      if (t + 1 < nk) { if (load_a) p2_wait_vm<8>(); else p2_wait_vm<4>(); }
      else p2_wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      const unsigned char* ap = smem + a_frag + as * P2::SLOT;
      const unsigned char* b0 = smem + tb[0] + (t & 1) * P2::SLOT;
      const unsigned char* b1 = smem + tb[1] + (t & 1) * P2::SLOT;
      bf16x8 bh[4], bl[4], ah, al;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int cb = ((n >> 1) & 1) * 8 + 2 * (n & 1);
        bh[n] = frag2(b0, b1, cb);
        bl[n] = frag2(b0, b1, cb | 4);
      }
      ah = *reinterpret_cast<const bf16x8*>(ap);
      al = *reinterpret_cast<const bf16x8*>(ap + lo_d);
      p2_wait_lgkm0();
      __builtin_amdgcn_s_barrier();
      const bool pre = t + 2 < nk;
      const int as2 = as == 0 ? 2 : as - 1;
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        bf16x8 ah_n, al_n;
        if (i + 1 < MF) {
          ah_n = *reinterpret_cast<const bf16x8*>(ap + (i + 1) * 2048);
          al_n = *reinterpret_cast<const bf16x8*>(ap + (i + 1) * 2048 + lo_d);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[n], acc[i][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[n], acc[i][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[n], acc[i][n], 0, 0, 0);
        if (i + 1 < MF) { ah = ah_n; al = al_n; }
        if (i + 1 < MF) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) {
#pragma unroll
          for (int j = i * 8 / MF; j < (i + 1) * 8 / MF; ++j) issue_one(j, t + 2, as2, t & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      as = as == 2 ? 0 : as + 1;
    }
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: (acc + beta C) * tanh' * dropout factor
  const bool drop = p.thresh != 0u;
  unsigned key_lo = p.seed_lo, key_hi = p.seed_hi;
  if (drop) apply_seed_offset(key_lo, key_hi, p.seed_dev);
  const bool has_beta = p.beta != 0.f;
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int row4 = row0 + (wr * MF + i) * 16 + 4 * g;
    if (row4 >= Mvalid) continue;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int col = 256 * ct + 64 * wc + 16 * n + l15;
      float ax[4], old[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = row4 + j < Mvalid ? row4 + j : Mvalid - 1;
        ax[j] = p.aux[(long)r * p.ldaux + col];
        if (has_beta) old[j] = p.C[(long)r * p.ldc + col];
      }
      unsigned w[4] = {0u, 0u, 0u, 0u};
      if (drop) philox4((unsigned)(p.drop_col_off + col), (unsigned)(row4 >> 2), p.site, 0u, key_lo, key_hi, w);
      float* cp = p.C + (long)row4 * p.ldc + col;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[i][n][j] + p.beta * old[j];
        const bool keep = !drop || w[j] >= p.thresh;
        const float f = 1.f - ax[j] * ax[j];
        v *= keep ? f * p.drop_scale : 0.f;
        if (row4 + j < Mvalid) cp[(long)j * p.ldc] = v;
      }
      __asm__ volatile("" ::: "memory");
    }
  }
}

// Forward-type launch (KIND 0: gemm_p2_nt_kernel, forward tiles; KIND 1: gemm_p2_nn_kernel, data-gradient tiles): gridDim.x
// workgroups (one per CU), every problem N = 256 nrep columns, K a multiple of 32.
template <int ABL, int KIND>
__device__ __forceinline__ void p2_rows_kernel_body(const GemmGroup& g, const int nrep, unsigned char* smem) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // diagnostics (ablate bit 64): per-workgroup begin / end stamps of the 100 MHz clock and the XCD id into g.p[0].slab
  const long long t_begin = (g.ablate & 64) ? (long long)wall_clock64() : 0;
  // ---- partition (every workgroup runs the same scalar arithmetic; the device-side row counts enter here) ----------------
  // A tile of MF row blocks costs P2_COST_TILE + P2_COST_RB MF units per k-step plus P2_TILE_FIXED for its fill and epilogue
  // (p2_partition.hpp: the k-loop runs at the matrix pipes' rate); a problem is cut into chunks of the largest number of row
  // blocks whose cost stays under a bound C; C is the smallest bound for which the chunks of all problems fit the grid.
  // (the row counts are read ONCE -- a scalar load from device memory per use made the bisection cost 60 us -- and every loop
  //  over the problems is fully unrolled so that these stay in registers)
  int rbv[LIREC_MAX_PROB], ksv[LIREC_MAX_PROB], rowsv[LIREC_MAX_PROB], nchv[LIREC_MAX_PROB];
#pragma unroll
  for (int i = 0; i < LIREC_MAX_PROB; ++i) {
    rowsv[i] = i < g.nprob ? dyn_limit(g.p[i], g.p[i].M) : 0;
    rbv[i] = (rowsv[i] + 31) >> 5;
    ksv[i] = i < g.nprob ? (g.p[i].K >> 5) : 1;
  }
  // (g.nt_bound: the bound left by the staging launch that produced this launch's operands -- same inputs, same search;
  //  g.nt_bound_val: computed on the host from static row counts)
  const int Chi = g.nt_bound_val ? g.nt_bound_val
                : (g.nt_bound ? __builtin_amdgcn_readfirstlane(*g.nt_bound) : p2_nt_search(rbv, ksv, (int)gridDim.x, nrep, lane));
  int Wtot = 0;
#pragma unroll
  for (int i = 0; i < LIREC_MAX_PROB; ++i) {
    nchv[i] = 0;
    if (rbv[i] == 0) continue;
    const int gm = p2_nt_gmax(Chi, ksv[i], rbv[i]);
    nchv[i] = (int)((unsigned)(rbv[i] + gm - 1) / (unsigned)gm);
    Wtot += nchv[i] * nrep;
  }
  // logical workgroup id: the workgroups an XCD hosts get consecutive ids (speed only), and when the launch has fewer work
  // items than workgroups every XCD takes its share of the items (ceil(W / 8)) instead of the first XCDs taking 32 each
  int L = blockIdx.x;
  {
    const int G = gridDim.x, b = blockIdx.x;
    if ((G & 7) == 0) {
      int per = (Wtot + 7) >> 3;
      per = per > (G >> 3) ? (G >> 3) : per;
      L = (b >> 3) < per ? (b & 7) * per + (b >> 3) : Wtot;
    }
  }
  int first = 0;
#pragma unroll
  for (int i = 0; i < LIREC_MAX_PROB; ++i) {
    const GemmProblem& p = g.p[i];
    const int rows = rowsv[i], rb = rbv[i];
    if (rb == 0) continue;
    const int nch = nchv[i];
    if (L >= first && L < first + nch * nrep) {
      int j, ct;
      if (g.nt_ct_major) { ct = (L - first) / nch; j = (L - first) - ct * nch; }
      else { j = (L - first) / nrep; ct = (L - first) - j * nrep; }
      const int rb0 = (int)((unsigned)(j * rb) / (unsigned)nch), rb1 = (int)((unsigned)((j + 1) * rb) / (unsigned)nch);
      const int nrb = rb1 - rb0;
      if (nrb > 0) {
        const int ntile = (nrb + 7) >> 3, base = nrb / ntile, rem = nrb - base * ntile;
        int r = rb0;
        for (int tl = 0; tl < ntile; ++tl) {
          const int mf = base + (tl < rem ? 1 : 0);
#define LIREC_P2_TILE(MFV)                                                                           \
  do {                                                                                               \
    if constexpr (KIND == 0) p2_nt_tile<MFV, ABL>(p, smem, 32 * r, rows, ct, lane, wave, g.ablate);  \
    else if constexpr (KIND == 2) p2_nt_tile<MFV, ABL, true>(p, smem, 32 * r, rows, ct, lane, wave, g.ablate);  \
    else if constexpr (KIND == 3) p2_nt_tile<MFV, ABL, true, 1>(p, smem, 32 * r, rows, ct, lane, wave, g.ablate);  \
    else if constexpr (KIND == 4) p2_nt_tile<MFV, ABL, true, 1, true>(p, smem, 32 * r, rows, ct, lane, wave, g.ablate);  \
    else if constexpr (KIND == 5) p2_nt_tile<MFV, ABL, true, 2, false, true>(p, smem, 32 * r, rows, ct, lane, wave, g.ablate,  \
                                                                             p.xq_out ? (nrep == 1 ? 3 : (ct < 2 ? 1 << ct : 0)) : 0);  \
    else if constexpr (KIND == 6) p2_nt_tile<MFV, ABL, true, 2, false, false, true>(p, smem, 32 * r, rows, ct, lane, wave, g.ablate);  \
    else p2_nn_tile<MFV, ABL>(p, smem, 32 * r, rows, ct, lane, wave, g.ablate);                      \
  } while (0)
          switch (mf) {
            case 1: LIREC_P2_TILE(1); break;
            case 2: LIREC_P2_TILE(2); break;
            case 3: LIREC_P2_TILE(3); break;
            case 4: LIREC_P2_TILE(4); break;
            case 5: LIREC_P2_TILE(5); break;
            case 6: LIREC_P2_TILE(6); break;
            case 7: LIREC_P2_TILE(7); break;
            default: LIREC_P2_TILE(8); break;
          }
#undef LIREC_P2_TILE
          r += mf;
        }
      }
    }
    first += nch * nrep;
  }
  if ((g.ablate & 64) && threadIdx.x == 0 && L < Wtot) {
    long long* dbg = reinterpret_cast<long long*>(g.p[0].slab) + 4L * L;
    dbg[0] = t_begin; dbg[1] = (long long)wall_clock64(); dbg[2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15; dbg[3] = blockIdx.x;
  }
}

template <int ABL>
__global__ __launch_bounds__(512, 2) void gemm_p2_nt_kernel(const GemmGroup g, const int nrep) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[P2::LDS_BYTES];
  p2_rows_kernel_body<ABL, 0>(g, nrep, smem);
}
// (rows gathered through GemmProblem::srow: every problem of the launch)
template <int ABL>
__global__ __launch_bounds__(512, 2) void gemm_p2_ntg_kernel(const GemmGroup g, const int nrep) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[P2::LDS_BYTES];
  p2_rows_kernel_body<ABL, 2>(g, nrep, smem);
}
// (rows gathered from q16b storage -- bf16-stored features: one plane.  ONE: round 5's single-pass form on q16b rows and q32b
//  weights -- since round 6 the library runs that mode on q16c operands, gemm_p2_ntg64_kernel below; this instantiation is kept
//  for tools/micro/p2o_bench.hip, which measures the two side by side)
template <int ABL, bool ONE = false>
__global__ __launch_bounds__(512, 2) void gemm_p2_ntg1_kernel(const GemmGroup g, const int nrep) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[P2::LDS_BYTES];
  p2_rows_kernel_body<ABL, ONE ? 4 : 3>(g, nrep, smem);
}
// (gemm mode 3 on operands stored as q16c -- bf16 values, 64 of k per 128-byte row: the two-plane kernel with one MFMA per
//  product; the host hands every problem over with K, lda and ldb halved, i.e. counted in 64-k "q32b columns")
template <int ABL>
__global__ __launch_bounds__(512, 2) void gemm_p2_ntg64_kernel(const GemmGroup g, const int nrep) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[P2::LDS_BYTES];
  p2_rows_kernel_body<ABL, 6>(g, nrep, smem);
}
// (rows fetched from the fp32 block itself and split on the way in: p2_nt_tile, XF)
template <int ABL>
__global__ __launch_bounds__(512, 2) void gemm_p2_ntx_kernel(const GemmGroup g, const int nrep) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[P2::LDS_BYTES];
  p2_rows_kernel_body<ABL, 5>(g, nrep, smem);
}
template <int ABL>
__global__ __launch_bounds__(512, 2) void gemm_p2_nn_kernel(const GemmGroup g, const int nrep) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[P2::LDS_BYTES];
  p2_rows_kernel_body<ABL, 1>(g, nrep, smem);
}

// -----------------------------------------------------------------------------------------------------------------
// weight gradient: one piece = k-steps [ks0, ks1) of the 256 x 256 tile (mt, nt) of problem p.
// p.A / p.A_lo: dZ1 planes (bf16 elements, lda in elements) at the segment's first column; p.B: the feature rows, q32b, at
// the segment's first column block (ldb = columns of the whole matrix).
// -----------------------------------------------------------------------------------------------------------------
// XP = 1: the feature rows are STORED as bf16 (q16b, gathered): their image is one k-major plane in the dZ1 operand's own format
// ([128-column sub-tile][32 k][256 B], transposed reads), four requests per loader wave and k-step, two MFMAs per product.
// ONE (gemm mode 3): one MFMA per product, and the rows are read from q16c storage (the same image; a request's 16 lanes of a k-row
// then fetch two whole 128-byte lines instead of four 64-byte halves).
template <bool DBIAS, int ABL, bool GATHER = false, int XP = 2, bool ONE = false>
__device__ __forceinline__ void p2_tn_piece(const GemmProblem& p, unsigned char* smem, int mt_, int nt_, int ks0_, int ks1_,
                                            bool whole, float* slab, float* dslab, int lane, int wave, int ablate) {
  constexpr int MF = 8;
  const int mt = __builtin_amdgcn_readfirstlane(mt_), nt = __builtin_amdgcn_readfirstlane(nt_);
  const int ks0 = __builtin_amdgcn_readfirstlane(ks0_), ks1 = __builtin_amdgcn_readfirstlane(ks1_);
  const int wr = wave >> 2, wc = wave & 3, g = lane >> 4, l15 = lane & 15;
  const unsigned short* Ah = reinterpret_cast<const unsigned short*>(p.A);
  const long a_lo = reinterpret_cast<const unsigned short*>(p.A_lo) - Ah;
  // Two wave groups half a k-step apart, as in the forward tile (p2_nt_tile): X = waves 0-3 (output rows [0, 128) of the tile)
  // multiplies in the even half of a k-step and is the LOADER of B (the feature rows: wave j requests k-rows 8 j .. 8 j + 7,
  // one instruction = 1 KiB = 256 columns each); Y = waves 4-7 (rows [128, 256)) multiplies in the odd half and loads A (dZ1:
  // wave j requests k-rows 8 j .. 8 j + 7 of both 128-column sub-tiles and both planes).
  const int role = wave >> 2, wj = wave & 3;
  const unsigned lds0 = p2_lds_addr(smem);
  unsigned a_off[2][2], a_dst[2][2], b_off[8];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = 8 * wj + 4 * q + (lane >> 4);
      const int f = ((k & 3) << 2) | ((k >> 2) & 3);
      const int col = 128 * u + 8 * ((lane & 15) ^ f);
      a_off[u][q] = 2u * (unsigned)(k * (int)p.lda + col);
      a_dst[u][q] = lds0 + P2::A0 + u * 8192 + (8 * wj + 4 * q) * 256;
    }
  // (image row k, LDS chunk `lane` <- source chunk sc = lane ^ f(k): column block sc >> 3, chunk sc & 7 of row k)
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int k = 8 * wj + q;
    const unsigned sc = (unsigned)(lane ^ (((k & 3) << 2) | ((k >> 2) & 3)));
    // (GATHER: the row's own offset comes from the index, per request)
    b_off[q] = (sc >> 3) * 4096u + (GATHER ? 0u : (unsigned)k * 128u) + (sc & 7u) * 16u;
  }
  static_assert(XP == 2 || (XP == 1 && GATHER), "one-plane rows are gathered from q16b storage");
  static_assert(!ONE || XP == 1, "the single-pass mode runs on the one-plane form");
  // (XP = 1) request (u, h): sub-tile u, k-rows 8 wj + 4 h + lane / 16; LDS chunk lane & 15 of the row <- source chunk ^ f(k):
  // columns 128 u + 8 sc of the tile = q16b column block 4 u + (sc >> 2), chunk sc & 3 of the row's 64 bytes
  unsigned b1_off[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = 8 * wj + 4 * h + (lane >> 4);
    const unsigned sc = (unsigned)((lane & 15) ^ (((k & 3) << 2) | ((k >> 2) & 3)));
    // (ONE -- the single-pass mode -- reads q16c storage: 64-column block sc >> 3 of the sub-tile's two, chunk sc & 7 of the row's 128 bytes)
    b1_off[h] = ONE ? (sc >> 3) * 4096u + (sc & 7u) * 16u : (sc >> 2) * 2048u + (sc & 3u) * 16u;
  }
  const unsigned b_dst = lds0 + P2::B0 + (XP == 1 ? (8 * wj) * 256 : (8 * wj) * 1024);
  const unsigned short* a_base = Ah + 256 * mt;
  const unsigned char* b_base = reinterpret_cast<const unsigned char*>(p.B) + (XP == 1 ? 8 * 2048 : 8 * 4096) * nt;
  const long a_step = 32 * p.lda, b_step = (long)(p.ldb >> 5) * 4096;
  // GATHER: k-row 32 t + 8 wj + q of the reduction is storage row srow[.] of the q32b matrix at p.B -- eight per loader wave and
  // k-step, fetched through the scalar cache a half-step before the requests are made (`sr0`, `sr1`)
  i32x4v sr0 = {0, 0, 0, 0}, sr1 = {0, 0, 0, 0};
  // this wave's eight requests of k-step t into slot `slot` of its operand's ring
  auto issue = [&](int t, int slot) {
    if (role == 0 && XP == 1) {
      // this lane's k-row of each half: entry lane / 16 of sr0 (h = 0) / sr1 (h = 1)
      const int v = lane >> 4;
      const int s0 = v == 0 ? sr0[0] : (v == 1 ? sr0[1] : (v == 2 ? sr0[2] : sr0[3]));
      const int s1 = v == 0 ? sr1[0] : (v == 1 ? sr1[1] : (v == 2 ? sr1[2] : sr1[3]));
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned char* src = b_base + (ONE ? p2_row_off16c(h == 0 ? s0 : s1, p.ldb) : p2_row_off16(h == 0 ? s0 : s1, p.ldb)) + b1_off[h] + u * 4 * 2048;
          p2_dma16_v(src, b_dst + slot * P2::SLOT + u * 8192 + h * 1024);
        }
    } else if (role == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if constexpr (GATHER) {
          const int v = q & 3;
          const int sidx = q < 4 ? (v == 0 ? sr0[0] : (v == 1 ? sr0[1] : (v == 2 ? sr0[2] : sr0[3])))
                                 : (v == 0 ? sr1[0] : (v == 1 ? sr1[1] : (v == 2 ? sr1[2] : sr1[3])));
          p2_dma16(b_base + p2_row_off(sidx, p.ldb), b_off[q], b_dst + slot * P2::SLOT + q * 1024);
        } else if constexpr ((ABL & 2048) == 0) p2_dma16_nt(b_base + (long)t * b_step, b_off[q], b_dst + slot * P2::SLOT + q * 1024);
        else p2_dma16(b_base + (long)t * b_step, b_off[q], b_dst + slot * P2::SLOT + q * 1024);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < (ONE ? 2 : 4); ++q) {             // (ONE: the hi plane alone -- no product uses dZ1's lo halves)
          const unsigned short* ab = a_base + (long)t * a_step + (q >> 1) * a_lo;
          p2_dma16(ab, a_off[u][q & 1], a_dst[u][q & 1] + slot * P2::SLOT + (q >> 1) * P2::IMG);
        }
    }
  };
  // transposed fragment reads: two per fragment (t = 0, 1), rows 8 g + 4 t + q4, columns base + 4 pp
  const int q4 = l15 >> 2, pp = lane & 3;
  int ta[2], tb[2], tx[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int k = 8 * g + 4 * t + q4;
    const int f = (q4 << 2) | ((2 * g + t) & 3);
    ta[t] = P2::A0 + wr * 8192 + k * 256 + 8 * (pp & 1);
    tb[t] = XP == 1 ? P2::B0 + (wc >> 1) * 8192 + k * 256 + 8 * (pp & 1) : P2::B0 + wc * 256 + k * 1024 + 8 * (pp & 1);
    tx[t] = ((pp >> 1) ^ f) << 4;
  }
  auto frag2 = [&](const unsigned char* p0, const unsigned char* p1, int cb) -> bf16x8 {
    const s16x4 x = lds_tr16(p0 + ((cb << 4) ^ tx[0]));
    const s16x4 y = lds_tr16(p1 + ((cb << 4) ^ tx[1]));
    const s16x8 v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    return *reinterpret_cast<const bf16x8*>(&v);
  };

  f32x4v acc[MF][4];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // bias gradient: column sums of A over k on the matrix pipe (A fragment x ones), tile column 0; wave column wc takes the
  // row fragments 2 wc, 2 wc + 1 of its wave row: two extra accumulators per wave
  f32x4v accb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) accb[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  if (!(ablate & 4)) {
    constexpr bool di = !(ABL & 16), dc = !(ABL & 32);
    bf16x8 bh[4], bl[4], ah, al;
    // the B fragments of the step in B slot `bs` and this wave's first A fragment of the step in A slot `as`
    auto read_frags = [&](int as, int bs) {
      const unsigned char* a0 = smem + ta[0] + as * P2::SLOT;
      const unsigned char* a1 = smem + ta[1] + as * P2::SLOT;
      const unsigned char* b0 = smem + tb[0] + bs * P2::SLOT;
      const unsigned char* b1 = smem + tb[1] + bs * P2::SLOT;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if constexpr (XP == 1) {
          bh[n] = frag2(b0, b1, 2 * (4 * (wc & 1) + n));     // fragment n of the wave's 64 columns in its 128-column sub-tile
        } else {
          const int cb = ((n >> 1) & 1) * 8 + 2 * (n & 1);
          bh[n] = frag2(b0, b1, cb);
          bl[n] = frag2(b0, b1, cb | 4);
        }
      }
      ah = frag2(a0, a1, 0);
      if constexpr (!ONE) al = frag2(a0 + P2::IMG, a1 + P2::IMG, 0);
      p2_wait_lgkm0();
    };
    auto multiply = [&](int as) {
      const unsigned char* a0 = smem + ta[0] + as * P2::SLOT;
      const unsigned char* a1 = smem + ta[1] + as * P2::SLOT;
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        bf16x8 ah_n, al_n;
        if (i + 1 < MF) { ah_n = frag2(a0, a1, 2 * (i + 1)); if constexpr (!ONE) al_n = frag2(a0 + P2::IMG, a1 + P2::IMG, 2 * (i + 1)); }
        if constexpr (!ONE) {
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[n], acc[i][n], 0, 0, 0);
        }
        if constexpr (XP == 2) {
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[n], acc[i][n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[n], acc[i][n], 0, 0, 0);
        if constexpr (DBIAS) {
          if ((i >> 1) == wc) {
            // (ONE: the bias gradient sums what the products see -- dZ1 rounded to bf16)
            if constexpr (!ONE) accb[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, ones, accb[i & 1], 0, 0, 0);
            accb[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ones, accb[i & 1], 0, 0, 0);
          }
        }
        if (i + 1 < MF) { ah = ah_n; if constexpr (!ONE) al = al_n; }
        if constexpr (!DBIAS) {
          if (i + 1 < MF) __builtin_amdgcn_sched_group_barrier(0x100, ONE ? 2 : 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, ONE ? 4 : 4 * (XP + 1), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // prologue: steps ks0, ks0 + 1 of both operands; then X takes its fragments of step ks0
    if (di) {
      if constexpr (GATHER) { if (role == 0) { sr0 = p2_sload4(p.srow + 32 * ks0 + 8 * wj); sr1 = p2_sload4(p.srow + 32 * ks0 + 8 * wj + 4); } }
      issue(ks0, 0);
      if (ks0 + 1 < ks1) {
        if constexpr (GATHER) { if (role == 0) { sr0 = p2_sload4(p.srow + 32 * (ks0 + 1) + 8 * wj); sr1 = p2_sload4(p.srow + 32 * (ks0 + 1) + 8 * wj + 4); } }
        issue(ks0 + 1, 1);
      }
    }
    // (all but the step ks0 + 1 requests of this wave: 4 for the loader of one-plane rows / of the hi plane alone, else 8)
    if (ks0 + 1 < ks1) { if ((role == 0 && XP == 1) || (role == 1 && ONE)) p2_wait_vm<4>(); else p2_wait_vm<8>(); } else p2_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (role == 0 && dc) read_frags(0, 0);
    int as = 0, bs = 0;
    if (role == 0) {
      for (int t = ks0; t < ks1; ++t) {
        __builtin_amdgcn_s_barrier();                        // ---- even half: X multiplies
        // (GATHER: the row list entries of step t + 2, fetched here so that they are back when the requests are made)
        if constexpr (GATHER) { if (t + 2 < ks1) { sr0 = p2_sload4(p.srow + 32 * (t + 2) + 8 * wj); sr1 = p2_sload4(p.srow + 32 * (t + 2) + 8 * wj + 4); } }
        if (dc) multiply(as);
        p2_wait_vm<0>();                                     // B(t + 1), requested a k-step ago, has landed
        __builtin_amdgcn_s_barrier();                        // ---- odd half: X loads and reads
        if (di && t + 2 < ks1) issue(t + 2, bs);             // every wave of Y has its B fragments of step t: the slot is free
        as = as == 2 ? 0 : as + 1;
        bs ^= 1;
        if (dc) read_frags(as, bs);                          // (behind the last step: a stale slot, never used)
      }
    } else {
      for (int t = ks0; t < ks1; ++t) {
        __builtin_amdgcn_s_barrier();                        // ---- even half: Y loads and reads
        const bool pre = di && t + 2 < ks1;
        if (pre) issue(t + 2, as == 0 ? 2 : as - 1);         // slot (t + 2) % 3: A(t - 1) is spent
        if (dc) read_frags(as, bs);
        if (pre) { if constexpr (ONE) p2_wait_vm<4>(); else p2_wait_vm<8>(); } else p2_wait_vm<0>();      // A(t + 1) has landed
        __builtin_amdgcn_s_barrier();                        // ---- odd half: Y multiplies
        if (dc) multiply(as);
        as = as == 2 ? 0 : as + 1;
        bs ^= 1;
      }
    }
    __builtin_amdgcn_s_barrier();
  }

  // ---- output: a whole tile is added to C (and the bias gradient) directly, a partial one goes to its slab -------------
  // (element (i, n, j) -> local row (wr MF + i) 16 + 4 g + j, local column 64 wc + 16 n + l15)
  float* obase = whole ? p.C + (long)(256 * mt + wr * MF * 16 + 4 * g) * p.ldc + 256 * nt + 64 * wc + l15
                       : slab + (long)(wr * MF * 16 + 4 * g) * 256 + 64 * wc + l15;
  const long old = whole ? p.ldc : 256;
  if (whole && p.beta == 0.f) {                 // (gradients overwritten: nothing to read back)
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) obase[(long)(16 * i + j) * old + 16 * n] = acc[i][n][j];
  } else if (whole) {
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      float o[4][4];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[n][j] = obase[(long)(16 * i + j) * old + 16 * n];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) obase[(long)(16 * i + j) * old + 16 * n] = o[n][j] + acc[i][n][j];
    }
  } else {
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) obase[(long)(16 * i + j) * old + 16 * n] = acc[i][n][j];
  }
  if constexpr (DBIAS) {
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int ml = (wr * MF + i) * 16 + 4 * g;            // local row of element 0
      if ((i >> 1) == wc && l15 == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (whole) p.dbias[256 * mt + ml + j] = p.dbias_set ? accb[i & 1][j] : p.dbias[256 * mt + ml + j] + accb[i & 1][j];
          else dslab[ml + j] = accb[i & 1][j];
        }
      }
    }
  }
}

// The partition of the weight-gradient launch, shared by the GEMM and the reduce kernel.  Strips = (problem, column tile nt),
// cost = k-steps of the problem; `nrep` = M / 256 row tiles are handled by nrep adjacent workgroups over the same range.
// In the partition's coordinate every tile is P2_PIECE_FIXED units longer than its k-steps: the head of the tile stands for
// what a piece costs besides its k-steps (pipeline fill, the 256-KiB slab), so a workgroup that starts a tile gets
// correspondingly fewer k-steps.
#define P2_PIECE_FIXED 5
__device__ __forceinline__ int p2_tn_ks(const GemmProblem& p) { return (dyn_limit(p, p.K) + 31) >> 5; }
__device__ __forceinline__ long p2_tn_len(const GemmProblem& p) { const int ks = p2_tn_ks(p); return ks > 0 ? ks + P2_PIECE_FIXED : 0; }
__device__ __forceinline__ long p2_tn_total(const GemmGroup& g) {
  long T = 0;
  for (int i = 0; i < g.nprob; ++i) T += p2_tn_len(g.p[i]) * (g.p[i].N >> 8);
  return T;
}
// k-steps [k0, k1) of the tile at [S, S + len) that fall to the range [a, b)
__device__ __forceinline__ void p2_tn_ksteps(long a, long b, long S, long len, int& k0, int& k1) {
  long u0 = (a > S ? a : S) - S - P2_PIECE_FIXED, u1 = (b < S + len ? b : S + len) - S - P2_PIECE_FIXED;
  k0 = (int)(u0 > 0 ? u0 : 0); k1 = (int)(u1 > 0 ? u1 : 0);
}

// slabs: g.p[0].slab = [2 * Gr * nrep][256 x 256] floats, g.p[0].dbias_slab = [2 * Gr * nrep][256]
template <int ABL, bool GATHER = false, int XP = 2, bool ONE = false>
__global__ __launch_bounds__(512, 2) void gemm_p2_tn_kernel(const GemmGroup g, const int nrep) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[P2::LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int L = p2_logical_id();
  const long Gr = gridDim.x / nrep;
  int rho = L / nrep;
  const int rep = L - rho * nrep;
  if (rho >= Gr) return;
  const long long t_begin = (g.ablate & 64) ? (long long)wall_clock64() : 0;
  const long T = p2_tn_total(g);
  // Which ranges an XCD hosts.  Consecutive ranges walk ONE column strip's rows; the workgroups that read the same rows of
  // dZ1 -- the same k-range of different strips -- lie a strip's length apart.  With consecutive ranges per XCD (16 here) an
  // XCD holds 4 strips x 4 quarters and every dZ1 quarter is fetched by each group of four strips again (4.5 x the planes'
  // size per launch, FETCH_SIZE); dealing every S-th range to an XCD, S = ranges per (longest) strip rounded to a power of two,
  // puts the same quarter of 16 strips there instead.  Measured: FETCH_SIZE of the launch 521 -> 489 MB (the workgroups of an XCD do
  // not walk in step, so the 4 MB L2 keeps less of a slice than the arithmetic hopes for), time unchanged (not fabric-bound).
  if ((Gr & 7) == 0 && !(g.ablate & 4096)) {
    long lmax = 0;
    for (int i = 0; i < g.nprob; ++i) { const long l = p2_tn_len(g.p[i]); lmax = l > lmax ? l : lmax; }
    const long per_strip = T > 0 ? (lmax * Gr + T / 2) / T : 1;          // ranges per longest strip
    const int S = per_strip >= 6 ? 8 : (per_strip >= 3 ? 4 : (per_strip >= 2 ? 2 : 1));
    const int per = (int)(Gr >> 3), x = rho / per, j = rho - x * per;
    rho = (x / S) * per * S + S * j + (x % S);
  }
  const long a = p2_cut(rho, T, Gr), b = p2_cut(rho + 1, T, Gr);
  long P = 0;
  for (int i = 0; i < g.nprob; ++i) {
    const GemmProblem& p = g.p[i];
    const long ks = p2_tn_ks(p), len = p2_tn_len(p), cost = len * (p.N >> 8);
    if (cost > 0 && a < P + cost && b > P) {
      const long lo = (a > P ? a : P) - P, hi = (b < P + cost ? b : P + cost) - P;
      const int s_first = (int)(lo / len), s_last = (int)((hi - 1) / len);
      for (int s = s_first; s <= s_last; ++s) {
        const long S = P + (long)s * len;
        int k0, k1;
        p2_tn_ksteps(a, b, S, len, k0, k1);
        if (k1 <= k0) continue;
        const bool whole = (k0 == 0 && k1 == ks);
        const long sid = ((long)rho * 2 + (a >= S ? 0 : 1)) * nrep + rep;
        float* dsl = g.p[0].dbias_slab ? g.p[0].dbias_slab + sid * 256 : nullptr;
        if (s == 0 && p.dbias != nullptr)
          p2_tn_piece<true, ABL, GATHER, XP, ONE>(p, smem, rep, s, k0, k1, whole, g.p[0].slab + sid * P2::SLAB, dsl, lane, wave, g.ablate);
        else
          p2_tn_piece<false, ABL, GATHER, XP, ONE>(p, smem, rep, s, k0, k1, whole, g.p[0].slab + sid * P2::SLAB, dsl, lane, wave, g.ablate);
      }
    }
    P += cost;
  }
  if ((g.ablate & 64) && threadIdx.x == 0) {
    long long* dbg = reinterpret_cast<long long*>(g.p[0].aux_out) + 4L * L;
    dbg[0] = t_begin; dbg[1] = (long long)wall_clock64(); dbg[2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15; dbg[3] = blockIdx.x;
  }
}

// Sums the partial tiles of the launch above into C (+=) and the bias gradients, in ascending workgroup order.
// grid = (number of 256 x 256 output tiles of all problems) x P2_RED_PARTS workgroups of 256 threads, a thread owns P2_RED_Q
// float4 of the tile (the search for the tile's pieces -- a few dozen integer divisions -- is paid once per thread: with one
// float4 per thread it was most of the kernel); `Gr` = the GEMM launch's gridDim.x / nrep.
#define P2_RED_Q 4
#define P2_RED_PARTS (64 / P2_RED_Q)
// The first-layer bucket's Adam update folded into the slab reduce (the recorded step, which issues backward and update as a
// unit): the thread that owns four elements of dW1 -- summed from the slabs, or left in C by the workgroup that had the tile whole --
// updates the parameters and moments at the same offsets of their flat buffers right away (same arithmetic, in the same order, as
// adam_kernel: bit-identical), still stores the gradient (it stays observable), and writes the new weights' q32b form into the
// next forward's operand buffer (GemmProblem::aux_out, optional): the separate reduce launch's 28 MB round trip through the
// gradient buffer, the Adam launch over this bucket and the W1 split of the next step's staging pass are gone.
template <bool ADAM>
static __global__ __launch_bounds__(256) void gemm_p2_tn_reduce_kernel(const GemmGroup g, const int nrep, const int Gr_, const AdamFuse ad) {
  const long Gr = Gr_;
  int tile = blockIdx.x / P2_RED_PARTS;
  const int part = blockIdx.x - tile * P2_RED_PARTS;
  // tile -> (problem, nt, rep)
  int pi = 0;
  long P = 0;
  for (; pi < g.nprob; ++pi) {
    const int nt_all = (g.p[pi].N >> 8) * nrep;
    if (tile < nt_all) break;
    tile -= nt_all;
    P += p2_tn_len(g.p[pi]) * (g.p[pi].N >> 8);
  }
  if (pi >= g.nprob) return;
  const GemmProblem& p = g.p[pi];
  const int nt = tile / nrep, rep = tile - nt * nrep;
  const int tid = threadIdx.x;
  const long ks = p2_tn_ks(p), len = p2_tn_len(p), T = p2_tn_total(g);
  const bool do_db = p.dbias != nullptr && nt == 0 && part == 0;
  float step_size = ad.step_size, bc2_sqrt = ad.bc2_sqrt;
  if constexpr (ADAM) {
    if (ad.step_dev) {      // step kept on the device (replays): the same double-precision bias corrections as adam_kernel
      const double t = (double)*ad.step_dev;
      step_size = (float)((double)ad.lr / (1.0 - pow((double)ad.beta1, t)));
      bc2_sqrt = (float)sqrt(1.0 - pow((double)ad.beta2, t));
    }
  }
  // what happens to the final gradient of this thread's float4 q at C + e: stored (unless C holds it already), and, ADAM, applied
  auto finish = [&](f32x4* cp, const f32x4 o, bool store) {
    if (store) *cp = o;
    if constexpr (ADAM) {
      const long off = reinterpret_cast<const float*>(cp) - ad.g;
      const f32x4 pn = adam4(ad, step_size, bc2_sqrt, off, o);
      if (p.aux_out) {      // the new weights as q32b [M][N] (rows of C): 8 bytes of hi halves, 8 of lo
        const long row = (reinterpret_cast<const float*>(cp) - p.C) / p.ldc, col = (reinterpret_cast<const float*>(cp) - p.C) - row * p.ldc;
        if (ad.wq16c) {     // (single-pass mode: bf16 values, 64-column blocks)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.aux_out) + (((row >> 5) * (p.ldc >> 6) + (col >> 6)) * 32 + (row & 31)) * 128 + (col & 63) * 2) = hi4(pn);
        } else {
          uint2 h2, l2;
          split4(pn, h2, l2);
          unsigned char* q = reinterpret_cast<unsigned char*>(p.aux_out) + (((row >> 5) * (p.ldc >> 5) + (col >> 5)) * 32 + (row & 31)) * 128 + (col & 31) * 2;
          *reinterpret_cast<uint2*>(q) = h2;
          *reinterpret_cast<uint2*>(q + 64) = l2;
        }
      }
    }
  };
  auto finish_db = [&](float db, bool store) {
    float* bp = p.dbias + 256 * rep + tid;
    if (store) *bp = db;
    if constexpr (ADAM) (void)adam1(ad, step_size, bc2_sqrt, bp - ad.g, db);
  };
  long e[P2_RED_Q];
#pragma unroll
  for (int q = 0; q < P2_RED_Q; ++q) e[q] = ((long)(part * P2_RED_Q + q) * 256 + tid) * 4;
  auto cptr = [&](int q) { return reinterpret_cast<f32x4*>(p.C + (long)(256 * rep + (int)(e[q] >> 8)) * p.ldc + 256 * nt + (int)(e[q] & 255)); };
  if (ks <= 0 || T <= 0) {
    // No row to reduce over (the device-side count of the compact context rows is 0): the GEMM launch skipped this problem.
    // Accumulating gradients (beta = 1) are then right as they stand; OVERWRITTEN ones (beta = 0: the recorded step, which has
    // no zeroing pass) must be stored as zeros, or the previous step's values would reach the optimiser.
#pragma unroll
    for (int q = 0; q < P2_RED_Q; ++q) {
      f32x4* cp = cptr(q);
      if (p.beta == 0.f) finish(cp, f32x4{0.f, 0.f, 0.f, 0.f}, true);
      else if (ADAM) finish(cp, *cp, false);
    }
    if (do_db) {
      if (p.dbias_set) finish_db(0.f, true);
      else if (ADAM) finish_db(p.dbias[256 * rep + tid], false);
    }
    return;
  }
  const long S = P + (long)nt * len;
  long r = (long)((unsigned)S * (unsigned)Gr / (unsigned)T);
  if (r > Gr - 1) r = Gr - 1;
  while (r + 1 < Gr && p2_cut(r + 1, T, Gr) <= S) ++r;
  while (r > 0 && p2_cut(r, T, Gr) > S) --r;
  // this thread's float4 q: row = (part * P2_RED_Q + q) * 4 + tid / 64, col = 4 (tid % 64)
  f32x4 v[P2_RED_Q];
#pragma unroll
  for (int q = 0; q < P2_RED_Q; ++q) v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db = 0.f;
  bool any = false, whole = false;
  for (; r < Gr; ++r) {
    const long ar = p2_cut(r, T, Gr), br = p2_cut(r + 1, T, Gr);
    if (ar >= S + len) break;
    int k0, k1;
    p2_tn_ksteps(ar, br, S, len, k0, k1);
    if (k1 <= k0) continue;
    if (k0 == 0 && k1 == ks) { whole = true; break; }      // a whole tile: its workgroup has added it to C already
    const long sid = (r * 2 + (ar >= S ? 0 : 1)) * nrep + rep;
    const float* sl = g.p[0].slab + sid * P2::SLAB;
    f32x4 s[P2_RED_Q];
#pragma unroll
    for (int q = 0; q < P2_RED_Q; ++q) s[q] = *reinterpret_cast<const f32x4*>(sl + e[q]);
#pragma unroll
    for (int q = 0; q < P2_RED_Q; ++q) { v[q][0] += s[q][0]; v[q][1] += s[q][1]; v[q][2] += s[q][2]; v[q][3] += s[q][3]; }
    if (do_db) db += g.p[0].dbias_slab[sid * 256 + tid];
    any = true;
  }
  if (whole || !any) {
    if constexpr (ADAM) {
#pragma unroll
      for (int q = 0; q < P2_RED_Q; ++q) { f32x4* cp = cptr(q); finish(cp, *cp, false); }
      if (do_db) finish_db(p.dbias[256 * rep + tid], false);
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < P2_RED_Q; ++q) {
    f32x4* cp = cptr(q);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (p.beta != 0.f) o = *cp;
    o[0] += v[q][0]; o[1] += v[q][1]; o[2] += v[q][2]; o[3] += v[q][3];
    finish(cp, o, true);
  }
  if (do_db) finish_db(p.dbias_set ? db : p.dbias[256 * rep + tid] + db, true);
}

}  // namespace lirec
