// Split-precision GEMM core for gfx950: fp32 in / fp32 out on the bf16 MFMA pipe.
//
// Each fp32 operand element is split on the fly into two bf16 pieces,
//   a = a_hi + a_lo,  a_hi = bf16_rne(a),  a_lo = bf16_rne(a - a_hi)
// (16 mantissa bits kept, full fp32 exponent range, no scaling needed), and every
// product is formed as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  by three
// v_mfma_f32_32x32x16_bf16 into ONE fp32 accumulator (the dropped a_lo*b_lo term is
// ~2^-18 |ab|, the order of the split's own truncation).  Per-product error ~2^-17,
// random in sign, so sums over K = 768..18432 terms land at ~1e-6..1e-5 of the
// output scale -- inside the 1e-4 parity bar (the parity tests run this core
// against the same reference golden vectors as the exact f32-MFMA core).  bf16 MFMA is
// 16x the f32-input MFMA rate, so three passes give up to 16/3 = 5.3x the f32 core's
// ceiling.
//
// Same problem descriptors, layouts, row selection, grouped launch, XCD remap, split-K
// and epilogues as gemm.hpp.  What differs is the on-chip operand path:
//   * k-contiguous global operands (A of NT/NN, B of NT): LDS tile [row][k] bf16, row
//     pitch 64 B (no padding), the row's four 16-B slots XOR-swizzled by (row >> 2) & 3; a lane's
//     MFMA fragment (8 consecutive k of one row) is one ds_read_b128.  Conflict-free both ways:
//     the 16 lanes a ds_read_b128 services together hold rows whose (row & 3, slot ^ swizzle)
//     pairs are all distinct, and the 16 lanes of a staging ds_write_b64 cover two whole rows =
//     128 contiguous bytes (a padded 80-B pitch -- the first version -- was conflict-free for
//     the reads only: PMC showed 28 % of the LDS cycles of the NT kernels were bank conflicts);
//   * m/n-contiguous global operands (A and B of dW = dY^T X, B of dX = dY W): LDS tile
//     [k][col] bf16 -- a straight, vectorised copy of the global tile -- and the fragment
//     is gathered by two ds_read_b64_tr_b16 (hardware 4x16 transpose read), so no
//     register transposes are needed; pitch = 2*cols + 64 B puts the four k-rows one read
//     touches on four different 64-B bank quarters;
//   * every tile has a hi and a lo image; conversion happens once per staged element,
//     between the global load and the LDS write.
// The profile that shaped it (rocprofv3 PMC, first version): 13 VALU instructions per
// MFMA -- VALU-issue bound on predication + conversion -- hence (a) tiles that lie fully
// inside the problem take a path with no predication at all, (b) no register transposes.
#pragma once
#include <type_traits>
#include "gemm.hpp"

namespace lirec {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// 10 VALU instructions per 4 elements: 2 v_cvt_pk_bf16_f32 (hi, round-to-nearest-even), the hi halves widened
// back to fp32 with one shift / one mask per element (written as bit operations: left to
// __builtin_convertvector hipcc converts every element a second time), 2 v_pk_add_f32 for the exact
// remainders, 2 v_cvt_pk_bf16_f32 for lo.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const f32x4 v, uint2& hi, uint2& lo) {
  const f32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
  const bf16x2 h01 = __builtin_convertvector(v01, bf16x2), h23 = __builtin_convertvector(v23, bf16x2);
  const unsigned w01 = __builtin_bit_cast(unsigned, h01), w23 = __builtin_bit_cast(unsigned, h23);
  const f32x2 f01 = {__builtin_bit_cast(float, w01 << 16), __builtin_bit_cast(float, w01 & 0xffff0000u)};
  const f32x2 f23 = {__builtin_bit_cast(float, w23 << 16), __builtin_bit_cast(float, w23 & 0xffff0000u)};
  const f32x2 r01 = v01 - f01, r23 = v23 - f23;
  const bf16x2 l01 = __builtin_convertvector(r01, bf16x2), l23 = __builtin_convertvector(r23, bf16x2);
  hi = make_uint2(w01, w23);
  lo = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}

// single-pass mode (gemm mode 3): the hi halves alone -- 2 VALU instructions per 4 elements
__device__ __forceinline__ uint2 hi4(const f32x4 v) {
  const f32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
  const bf16x2 h01 = __builtin_convertvector(v01, bf16x2), h23 = __builtin_convertvector(v23, bf16x2);
  return make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
}

// q32b ("blocked q32"): the storage of the feature rows and the first-layer weights for these kernels.  An [R][C] fp32
// matrix (R, C multiples of 32) is cut into 32 x 32 blocks, block (rb, cb) at byte ((rb * (C / 32) + cb) * 4096; inside a block
// row r (0..31) holds 128 B: the 32 hi halves (bf16_rne(a)) then the 32 lo halves (bf16_rne(a - hi)).  Same footprint as the
// fp32 matrix.  Why blocked: one k-step of a 32-row group is ONE contiguous 4 KiB (the row-major form touched 32 different
// DRAM pages for 128 B each, every k-step again: the feature stream crawled at 2 TB/s), and consecutive k-steps / column
// blocks are consecutive 4 KiB chunks.
// One call = 8 consecutive elements of row `row`, columns 8 c8 .. 8 c8 + 7.
__device__ __forceinline__ void p2_store_q32b(unsigned char* dst, long row, int c8, int cblocks, const f32x4 a, const f32x4 b) {
  uint2 h0, l0, h1, l1;
  split4(a, h0, l0);
  split4(b, h1, l1);
  unsigned char* blk = dst + (((row >> 5) * cblocks + (c8 >> 2)) * 32 + (row & 31)) * 128 + (c8 & 3) * 16;
  *reinterpret_cast<uint4*>(blk) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4*>(blk + 64) = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// q16b ("blocked bf16"): the ONE-PLANE sibling of q32b, for features that are stored as bf16 (BASELINE config 5): 32 x 32 blocks of
// 2 KiB, block (rb, cb) at byte (rb * (C / 32) + cb) * 2048, row r of a block = 64 B = its 32 bf16 values (the stored value IS the
// hi half; there is no lo half).  Half the fp32 footprint; one k-step of a 32-row group = one contiguous 2 KiB.
__device__ __forceinline__ void p2_store_q16b(unsigned char* dst, long row, int c8, int cblocks, const f32x4 a, const f32x4 b) {
  const uint2 h0 = hi4(a), h1 = hi4(b);
  unsigned char* blk = dst + (((row >> 5) * cblocks + (c8 >> 2)) * 32 + (row & 31)) * 64 + (c8 & 3) * 16;
  *reinterpret_cast<uint4*>(blk) = make_uint4(h0.x, h0.y, h1.x, h1.y);
}

// q16c: the same values in 32 x 64 blocks of 4 KiB -- block (rb, cb) at byte (rb * (C / 64) + cb) * 4096, row r of a block = 128 B =
// its 64 bf16 values.  The storage of the SINGLE-PASS mode (gemm mode 3), rows and first-layer weights alike: a k-step of the
// forward kernel is then 64 of k, one whole 128-byte line per row for BOTH operands (q16b rows are 64-byte half lines, and the
// weights came as q32b with a lo half no product used), and byte for byte the addressing of a q32b matrix of C / 2 columns --
// the two-plane kernel runs on it unchanged, its "hi" chunks holding k 0-31 of the step and its "lo" chunks k 32-63.
__device__ __forceinline__ long p2_q16c_off(long row, int c8, int cblocks64) {
  return (((row >> 5) * cblocks64 + (c8 >> 3)) * 32 + (row & 31)) * 128 + (c8 & 7) * 16;
}
__device__ __forceinline__ void p2_store_q16c(unsigned char* dst, long row, int c8, int cblocks64, const f32x4 a, const f32x4 b) {
  const uint2 h0 = hi4(a), h1 = hi4(b);
  *reinterpret_cast<uint4*>(dst + p2_q16c_off(row, c8, cblocks64)) = make_uint4(h0.x, h0.y, h1.x, h1.y);
}

__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(reinterpret_cast<const s16x4*>(p)));
}

// One operand's LDS image.  EXT = tile extent along the non-reduced index (BM or BN);
// KC = the operand is k-contiguous in global memory.
template <bool KC, int EXT>
struct OperandTile {
  static constexpr int BK = 32;
  static constexpr int PITCH = KC ? (BK * 2) : (EXT * 2 + 64);        // bytes per LDS row
  static constexpr int BYTES = KC ? EXT * PITCH : BK * PITCH;         // one image (hi or lo)
  static constexpr int NCH = EXT / 32;                                // float4 chunks per thread per k-tile
};

// Tile configurations (CFG): waves are arranged WAVES_M x WAVES_N, each owns WM x WN MFMA tiles of 32x32.
//   0:  64 x  64, 4 waves (small problems)        1: 128 x 128, 4 waves, 2 workgroups per CU
//   2: 256 x 256, 8 waves, 1 workgroup per CU (all 160 KiB of LDS): the same global bytes per k-tile as two
//      128x128 workgroups but twice the MFMAs -- the staging loads of a CU drain at only ~16 B/clk
//      (PMC: neither prefetch depth nor pre-converted operands moved the 128x128 kernel), so bytes per
//      flop is what sets the rate of the large GEMMs.
template <int CFG> struct TileCfg;
template <> struct TileCfg<0> { static constexpr int WAVES_M = 2, WAVES_N = 2, WM = 1, WN = 1; };
template <> struct TileCfg<1> { static constexpr int WAVES_M = 2, WAVES_N = 2, WM = 2, WN = 2; };
template <> struct TileCfg<2> { static constexpr int WAVES_M = 2, WAVES_N = 4, WM = 4, WN = 2; };
//   3: 128 x 128, 8 waves of 32x64 (<= 128 registers): two workgroups per CU = 4 waves per SIMD, so the
//      load / convert / MFMA phases of different waves overlap
template <> struct TileCfg<3> { static constexpr int WAVES_M = 4, WAVES_N = 2, WM = 1, WN = 2; };
//   4: 256 x 128, 8 waves of 64x64, one workgroup per CU (kept for experiments: never the fastest on this path)
template <> struct TileCfg<4> { static constexpr int WAVES_M = 4, WAVES_N = 2, WM = 2, WN = 2; };

// XB: the X operand (A of NT, B of TN) is stored as bf16 (BASELINE config 5, "bf16 storage"): its hi image is the
// stored value itself, its lo image is zero -- no split, no lo fragments, two MFMAs per product instead of three.
template <int LAYOUT, int CFG, int TAG, bool VEC, bool XB = false>
__global__ __launch_bounds__(64 * TileCfg<CFG>::WAVES_M * TileCfg<CFG>::WAVES_N, (CFG == 3 ? 4 : 2)) void gemm_bf16x3_kernel(const GemmGroup g) {
  constexpr int WAVES_M = TileCfg<CFG>::WAVES_M, WAVES_N = TileCfg<CFG>::WAVES_N;
  constexpr int WM = TileCfg<CFG>::WM, WN = TileCfg<CFG>::WN;
  constexpr int NTHR = 64 * WAVES_M * WAVES_N;
  constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N, BK = 32;
  constexpr bool A_KC = (LAYOUT != L_TN);
  constexpr bool B_KC = (LAYOUT == L_NT);
  static_assert(!XB || (VEC && LAYOUT != L_NN), "bf16 X: dword-aligned staging, forward or weight-gradient layout");
  constexpr bool AXB = XB && LAYOUT == L_NT, BXB = XB && LAYOUT == L_TN;     // which operand is the bf16 X
  using TA = OperandTile<A_KC, BM>;
  using TB = OperandTile<B_KC, BN>;
  constexpr int BUF = 2 * (TA::BYTES + TB::BYTES);     // A_hi, A_lo, B_hi, B_lo
  constexpr int RP = NTHR / 8;                         // k-contiguous: rows staged per pass
  constexpr int QA = BM / 4, QB = BN / 4;              // n-contiguous: column quads per k-row
  constexpr int KSA = NTHR / QA, KSB = NTHR / QB;      //               k-rows staged per pass
  constexpr int CA = A_KC ? BM / RP : BK / KSA;        // float4 chunks per thread per k-tile
  constexpr int CB = B_KC ? BN / RP : BK / KSB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ltile = launch_tile(g);
  if (ltile < 0) return;
  const TileCoord tc = decode_tile<BM, BN>(g, ltile, LAYOUT == L_TN);
  const GemmProblem& p = g.p[tc.pi];
  const int m0 = tc.m0, n0 = tc.n0;
  const int M = tc.M, N = p.N, K = tc.k_end, kb = tc.k_begin;
  if (m0 >= M) return;                           // row-compacted launch: nothing beyond the valid rows
  // interior tile: no row/column/k predicate can fire -> unpredicated loads and stores
  const bool interior = (m0 + BM <= M) && (n0 + BN <= N) && (((K - kb) & (BK - 1)) == 0);

  // ---- staging assignment ------------------------------------------------------
  // k-contiguous: chunk i -> row (tid>>3) + RP i, k-quad tid&7
  // n-contiguous: chunk i -> k-row (tid / (EXT/4)) + KS i, column quad tid % (EXT/4)
  const float* a_rowptr[A_KC ? CA : 1];
  const float* b_rowptr[B_KC ? CB : 1];
  bool a_rowok[A_KC ? CA : 1], b_rowok[B_KC ? CB : 1];
  if constexpr (A_KC) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int m = m0 + (tid >> 3) + RP * i;
      a_rowok[i] = m < M;
      const long arow = (LAYOUT == L_NT ? phys_row(p, a_rowok[i] ? m : 0) : (long)(a_rowok[i] ? m : 0)) * p.lda;
      // (bf16 X: the same element offset, half the bytes)
      a_rowptr[i] = AXB ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.A) + 2 * arow) : p.A + arow;
    }
  }
  if constexpr (B_KC) {
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int n = n0 + (tid >> 3) + RP * i;
      b_rowok[i] = n < N;
      b_rowptr[i] = p.B + (long)(b_rowok[i] ? n : 0) * p.ldb;
    }
  }
  const int a_cq = 4 * (tid % QA), a_kr = tid / QA;
  const int b_cq = 4 * (tid % QB), b_kr = tid / QB;

  f32x4 ra[CA], rb[CB];                          // one k-tile of staging registers (rolling, see mainloop)
  // TN with a row map: the row id of the NEXT load of each B chunk, fetched one tile ahead and left
  // untouched until then, so that the feature-row load never waits on the index load it depends on
  // (the memory counter is in-order: consuming the index at once would wait for every load before it)
  int b_nextrow[(LAYOUT == L_TN) ? CB : 1];
  // TAG 3 = the row-mapped weight-gradient launch, TAG 2 = the unmapped one: a compile-time property there,
  // so the k-loop carries no run-time branch on it (a branch in the loop makes hipcc drain the memory counter)
  const bool tn_mapped = (LAYOUT == L_TN) && (TAG == 3 ? true : (TAG == 2 ? false : p.rowmap != nullptr));
  float dbias_acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_dbias = (LAYOUT == L_TN) && p.dbias != nullptr && tc.tn == 0;

  // global -> registers, one staged chunk (0..CA-1: A, CA..CA+CB-1: B) of the k-tile starting at k0
  constexpr int NCHUNK = CA + CB;
  auto load_chunk = [&](int k0, int c, auto edge_tag) __attribute__((always_inline)) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    if (c < CA) {
      const int i = c;
      if constexpr (A_KC) {
        const int k = k0 + 4 * (tid & 7);
        if constexpr (AXB) ra[i] = raw4_bf16(a_rowptr[i], k, !EDGE || (a_rowok[i] && k < K), p.A);
        else if constexpr (EDGE) ra[i] = raw4<VEC>(a_rowptr[i] + k, a_rowok[i] ? K - k : 0, p.A);
        else ra[i] = raw4<VEC>(a_rowptr[i] + k, 4, p.A);
      } else {
        const int k = k0 + a_kr + KSA * i;
        if constexpr (EDGE) ra[i] = raw4<VEC>(p.A + (long)k * p.lda + m0 + a_cq, (k < K) ? M - (m0 + a_cq) : 0, p.A);
        else ra[i] = raw4<VEC>(p.A + (long)k * p.lda + m0 + a_cq, 4, p.A);
      }
    } else {
      const int i = c - CA;
      if constexpr (B_KC) {
        const int k = k0 + 4 * (tid & 7);
        if constexpr (EDGE) rb[i] = raw4<VEC>(b_rowptr[i] + k, b_rowok[i] ? K - k : 0, p.B);
        else rb[i] = raw4<VEC>(b_rowptr[i] + k, 4, p.B);
      } else {
        const int k = k0 + b_kr + KSB * i;
        const bool kok = !EDGE || k < K;
        long row;
        if constexpr (LAYOUT == L_TN) {
          if (tn_mapped) {
            row = sel_row(p, b_nextrow[i]);                          // id fetched while the previous tile was staged
          } else {
            row = phys_row(p, kok ? k : 0);
          }
        } else {
          row = (long)k;
        }
        if constexpr (BXB) rb[i] = raw4_bf16(p.B, row * p.ldb + n0 + b_cq, !EDGE || (kok && n0 + b_cq < N), p.B);
        else if constexpr (EDGE) rb[i] = raw4<VEC>(p.B + row * p.ldb + n0 + b_cq, kok ? N - (n0 + b_cq) : 0, p.B);
        else rb[i] = raw4<VEC>(p.B + row * p.ldb + n0 + b_cq, 4, p.B);
        if constexpr (LAYOUT == L_TN) {
          if (tn_mapped) {                                           // issued behind the feature load it does not feed
            const int kn = k + BK;
            b_nextrow[i] = p.rowmap[kn < K ? kn : (K > 0 ? K - 1 : 0)];
          }
        }
      }
    }
  };

  // k-contiguous images: thread -> row tid>>3 (+ RP i), 8-byte piece tid&7 of the row's 64 B; the piece's
  // 16-B slot is swizzled by (row >> 2) & 3 (RP is a multiple of 16, so the swizzle is the same for every i)
  static_assert(RP % 16 == 0, "swizzle must not depend on the chunk index");
  const int kc_store_off = (tid >> 3) * 64 + (((((tid & 7) >> 1) ^ ((tid >> 5) & 3)) << 4) | ((tid & 1) << 3));

  // one staged chunk: registers -> LDS stage `buf` (predicate on edge tiles, split into hi/lo, write).
  // Chunks 0..CA-1 belong to A, CA..CA+CB-1 to B.
  auto store_chunk = [&](int buf, int k0, int c, auto edge_tag, auto one_tag) __attribute__((always_inline)) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    constexpr bool ONE = decltype(one_tag)::value;      // single pass: operands rounded to bf16 once, no lo images
    unsigned char* a_hi = smem + buf * BUF;
    unsigned char* a_lo = a_hi + TA::BYTES;
    unsigned char* b_hi = a_lo + TA::BYTES;
    unsigned char* b_lo = b_hi + TB::BYTES;
    uint2 h, l;
    if (c < CA) {
      const int i = c;
      f32x4 v = ra[i];
      if constexpr (AXB) {
        // stored bf16: the two dwords ARE the hi image; there is no lo image
        { const float fx = v.x, fy = v.y; h = make_uint2(__float_as_uint(fx), __float_as_uint(fy)); }
        if constexpr (EDGE) { if (!(a_rowok[i] && k0 + 4 * (tid & 7) < K)) h = make_uint2(0u, 0u); }
        *reinterpret_cast<uint2*>(a_hi + kc_store_off + RP * i * TA::PITCH) = h;
      } else if constexpr (A_KC) {
        if constexpr (EDGE) v = mask4(v, a_rowok[i] ? K - (k0 + 4 * (tid & 7)) : 0);
        const int off = kc_store_off + RP * i * TA::PITCH;
        if constexpr (ONE) { *reinterpret_cast<uint2*>(a_hi + off) = hi4(v); }
        else {
          split4(v, h, l);
          *reinterpret_cast<uint2*>(a_hi + off) = h;
          *reinterpret_cast<uint2*>(a_lo + off) = l;
        }
      } else {
        if constexpr (EDGE) v = mask4(v, (k0 + a_kr + KSA * i < K) ? M - (m0 + a_cq) : 0);
        if (do_dbias) {
          const int kr = k0 + a_kr + KSA * i;
          const float rs = p.rowscale ? ((kr < K) ? p.rowscale[kr] : 0.f) : 1.f;
          dbias_acc[0] += v.x * rs; dbias_acc[1] += v.y * rs; dbias_acc[2] += v.z * rs; dbias_acc[3] += v.w * rs;
        }
        const int off = (a_kr + KSA * i) * TA::PITCH + 2 * a_cq;
        if constexpr (ONE) { *reinterpret_cast<uint2*>(a_hi + off) = hi4(v); }
        else {
          split4(v, h, l);
          *reinterpret_cast<uint2*>(a_hi + off) = h;
          *reinterpret_cast<uint2*>(a_lo + off) = l;
        }
      }
    } else {
      const int i = c - CA;
      f32x4 v = rb[i];
      if constexpr (B_KC) {
        if constexpr (EDGE) v = mask4(v, b_rowok[i] ? K - (k0 + 4 * (tid & 7)) : 0);
        const int off = kc_store_off + RP * i * TB::PITCH;
        if constexpr (ONE) { *reinterpret_cast<uint2*>(b_hi + off) = hi4(v); }
        else {
          split4(v, h, l);
          *reinterpret_cast<uint2*>(b_hi + off) = h;
          *reinterpret_cast<uint2*>(b_lo + off) = l;
        }
      } else if constexpr (BXB) {
        { const float fx = v.x, fy = v.y; h = make_uint2(__float_as_uint(fx), __float_as_uint(fy)); }
        if constexpr (EDGE) { if (!(k0 + b_kr + KSB * i < K && n0 + b_cq < N)) h = make_uint2(0u, 0u); }
        *reinterpret_cast<uint2*>(b_hi + (b_kr + KSB * i) * TB::PITCH + 2 * b_cq) = h;
      } else {
        if constexpr (EDGE) v = mask4(v, (k0 + b_kr + KSB * i < K) ? N - (n0 + b_cq) : 0);
        const int off = (b_kr + KSB * i) * TB::PITCH + 2 * b_cq;
        if constexpr (ONE) { *reinterpret_cast<uint2*>(b_hi + off) = hi4(v); }
        else {
          split4(v, h, l);
          *reinterpret_cast<uint2*>(b_hi + off) = h;
          *reinterpret_cast<uint2*>(b_lo + off) = l;
        }
      }
    }
  };
  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm0 = (wave / WAVES_N) * 32 * WM, wn0 = (wave % WAVES_N) * 32 * WN;
  const int l31 = lane & 31, lh = lane >> 5;
  // fragment base offsets inside one image
  //   k-contiguous tile [row][k]:  row = w0 + 32 i + (lane&31), bytes 16*(lane>>5) + 32 s
  //   n-contiguous tile [k][col]:  two transpose reads t = 0,1 of the 4 x 16 block
  //       rows 16 s + 8 (lane>>5) + 4 t + q,  cols w0 + 32 i + 16 ((lane>>4)&1) + 4 p .. +3
  //       with q = (lane&15) >> 2, p = lane & 3 (the lane then receives column (lane&15) of those rows)
  // k-contiguous: slot (2 s + lh) ^ swz with swz = (row >> 2) & 3 = (lane >> 2) & 3 (tile origins are
  // multiples of 32): byte offset 16 (lh ^ (swz & 1)) + 32 (s ^ (swz >> 1)) -- the k-step flips one bit
  const int kc_swz = (lane >> 2) & 3;
  const int kc_s0 = 16 * (lh ^ (kc_swz & 1)) + 32 * (kc_swz >> 1);
  const int a_frag = A_KC ? (wm0 + l31) * TA::PITCH + kc_s0
                          : (8 * lh + ((lane & 15) >> 2)) * TA::PITCH + 2 * (wm0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
  const int b_frag = B_KC ? (wn0 + l31) * TB::PITCH + kc_s0
                          : (8 * lh + ((lane & 15) >> 2)) * TB::PITCH + 2 * (wn0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));

  auto frag_a = [&](const unsigned char* img, int i, int s) -> bf16x8 {
    if constexpr (A_KC) {
      return *reinterpret_cast<const bf16x8*>(img + (a_frag ^ (32 * s)) + 32 * TA::PITCH * i);
    } else {
      const unsigned char* q = img + a_frag + 64 * i + 16 * s * TA::PITCH;
      const s16x4 x = lds_tr16(q), y = lds_tr16(q + 4 * TA::PITCH);
      const s16x8 v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
      return *reinterpret_cast<const bf16x8*>(&v);
    }
  };
  auto frag_b = [&](const unsigned char* img, int j, int s) -> bf16x8 {
    if constexpr (B_KC) {
      return *reinterpret_cast<const bf16x8*>(img + (b_frag ^ (32 * s)) + 32 * TB::PITCH * j);
    } else {
      const unsigned char* q = img + b_frag + 64 * j + 16 * s * TB::PITCH;
      const s16x4 x = lds_tr16(q), y = lds_tr16(q + 4 * TB::PITCH);
      const s16x8 v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
      return *reinterpret_cast<const bf16x8*>(&v);
    }
  };

  // One k-tile: the MFMAs of the tile in LDS stage `buf`, with the staging of later tiles dealt out
  // between the MFMA groups (a 32x32x16 MFMA holds the SIMD's issue port for only 8 of its 32 cycles, so
  // the convert/split VALU work rides in the shadow of the matrix pipe instead of forming a serial phase
  // behind it -- the ablation build showed the two phases were purely additive before).  ROLLING
  // REGISTERS: after group g, chunk c(g) of tile kt+1 -- requested one whole iteration ago -- is
  // converted and written to the other LDS stage, and its registers are at once re-loaded with the
  // same chunk of tile kt+2.  One register set, every load has a full iteration to land, and the
  // counted vmcnt the compiler derives is the constant NCHUNK-1.
  constexpr int NGROUP = 2 * WM * WN;
  // chunks [g NCHUNK / NGROUP, (g + 1) NCHUNK / NGROUP) follow group g: spread evenly whatever the two counts
  auto ktile = [&](int buf, int k0_next, int k0_next2, auto store_tag, auto load_tag, auto edge_tag, auto one_tag) __attribute__((always_inline)) {
    constexpr bool STORE_NEXT = decltype(store_tag)::value, LOAD_NEXT2 = decltype(load_tag)::value;
    constexpr bool ONE = decltype(one_tag)::value;
    const unsigned char* a_hi = smem + buf * BUF;
    const unsigned char* a_lo = a_hi + TA::BYTES;
    const unsigned char* b_hi = a_lo + TA::BYTES;
    const unsigned char* b_lo = b_hi + TB::BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) { ah[i] = frag_a(a_hi, i, s); if constexpr (!AXB && !ONE) al[i] = frag_a(a_lo, i, s); }
#pragma unroll
      for (int j = 0; j < WN; ++j) { bh[j] = frag_b(b_hi, j, s); if constexpr (!BXB && !ONE) bl[j] = frag_b(b_lo, j, s); }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if constexpr (!AXB && !ONE) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          if constexpr (!BXB && !ONE) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          const int grp = (s * WM + i) * WN + j;
#pragma unroll
          for (int c = grp * NCHUNK / NGROUP; c < (grp + 1) * NCHUNK / NGROUP; ++c) {
            if constexpr (STORE_NEXT) store_chunk(buf ^ 1, k0_next, c, edge_tag, one_tag);
            if constexpr (LOAD_NEXT2) load_chunk(k0_next2, c, edge_tag);
          }
        }
    }
  };

  auto mainloop = [&](auto edge_tag, auto one_tag) __attribute__((always_inline)) {
    const int nk = (K - kb + BK - 1) / BK;
    if (nk <= 0) return;                         // a k-chunk beyond the valid rows: the partial tile is zero
    if constexpr (LAYOUT == L_TN) {
      if (tn_mapped) {
#pragma unroll
        for (int i = 0; i < CB; ++i) {
          const int k = kb + b_kr + KSB * i;
          b_nextrow[i] = p.rowmap[k < K ? k : (K > 0 ? K - 1 : 0)];
        }
      }
    }
    // prologue: tile 0 -> LDS stage 0, tile 1 -> registers
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) load_chunk(kb, c, edge_tag);
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      store_chunk(0, kb, c, edge_tag, one_tag);
      if (nk > 1) load_chunk(kb + BK, c, edge_tag);
    }
    __syncthreads();
    // steady state (no branches inside a k-tile, so the compiler keeps counted vmcnt waits), then the
    // two tail iterations: nothing left to request, nothing left to stage
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
      ktile(kt & 1, kb + (kt + 1) * BK, kb + (kt + 2) * BK, std::true_type{}, std::true_type{}, edge_tag, one_tag);
      __syncthreads();
    }
    if (kt + 1 < nk) {
      ktile(kt & 1, kb + (kt + 1) * BK, 0, std::true_type{}, std::false_type{}, edge_tag, one_tag);
      __syncthreads();
      ++kt;
    }
    ktile(kt & 1, 0, 0, std::false_type{}, std::false_type{}, edge_tag, one_tag);
    __syncthreads();
  };
  // (diagnostics: lirec_debug_set(4, cfg) skips the k-loop -- what is left is the per-tile fixed work)
  // g.onepass (gemm mode 3, the large GEMMs only): operands rounded to bf16 once, ONE MFMA per product, fp32 accumulate --
  // the single-pass leg of BASELINE config 5; never the headline arithmetic
  // (not instantiated for the generic weight-gradient kernels of the 128x128x4-wave / 256-row tiles: no single-pass call site
  //  selects them, and the second copy of their loop pushed hipcc into 2 KB of scratch per lane; they stay three-pass)
  constexpr bool ONE_OK = !(LAYOUT == L_TN && TAG == 0 && (CFG == 1 || CFG == 2 || CFG == 4));
  bool one = false;
  if constexpr (ONE_OK) one = g.onepass != 0;
  if (g.ablate & 4) { /* no k-loop */ }
  else if (one) {
    if constexpr (ONE_OK) { if (interior) mainloop(std::false_type{}, std::true_type{}); else mainloop(std::true_type{}, std::true_type{}); }
  }
  else if (interior) mainloop(std::false_type{}, std::false_type{});
  else mainloop(std::true_type{}, std::false_type{});

  gemm_epilogue<WM, WN, LAYOUT>(p, acc, m0, n0, wm0, wn0, lane, tc.split, M);

  if constexpr (LAYOUT == L_TN) {
    if (do_dbias) {
      // a thread summed columns m0 + a_cq .. +3 over its k-rows; the threads that share a column
      // quad are tid, tid + QA, tid + 2 QA, ... : reduce through LDS (the stages are free now)
      float* red = reinterpret_cast<float*>(smem);
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 4; ++c) red[tid * 4 + c] = dbias_acc[c];
      __syncthreads();
      if (tid < QA) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < KSA; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += red[(tid + QA * r) * 4 + c];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int m = m0 + 4 * tid + c;
          if (m < M) {
            if (p.ksplit > 1) p.dbias_slab[(long)tc.split * p.M + m] = v[c];
            else p.dbias[m] = p.dbias_set ? v[c] : p.dbias[m] + v[c];
          }
        }
      }
    }
  }
}

}  // namespace lirec
