// asr_gemm_f32: general f32 GEMM on the f32-input MFMA (see gemm_core.h); desc.compute = 1 rounds the operands to
// bf16 on their way into the bf16 MFMA (f32 storage, accumulation and epilogue: the mixed-precision mode).
//
// This file is compiled four times (the kernels dominate the library's build time): as gemm.hip with GEMM_BF = 0 (the f32-MFMA
// kernels and the C entry point), through gemm_bf16.hip with GEMM_BF = 1 (the bf16-operand kernels of --mixed-precision) and
// through gemm_split9.hip / gemm_split6.hip with GEMM_BF = 2 / 3 (f32 products as nine / six exact bf16 pair products on the
// bf16 matrix pipe: gemm_core.h run_split).
#include <stdlib.h>

#include "gemm_core.h"

#ifndef GEMM_BF
#define GEMM_BF 0
#endif

// Tile order.  Workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the blocks that share an
// XCD's L2 - a speed assumption only), so the linear id is first remapped to p = (position of b inside
// its XCD's own sequence, XCD-major): every XCD then owns one contiguous range of p.  p walks the
// tiles of the SMALLER operand fastest, so the concurrently resident blocks of an XCD share the tile
// of the larger operand (read from HBM once) while the smaller operand stays L2 resident.
__device__ __forceinline__ void tile_of_block(int tiles_m, int tiles_n, int walk_n, int* bm, int* bn) {
  const int total = tiles_m * tiles_n;
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
  const int p = xcd * (total >> 3) + min(xcd, total & 7) + idx;   // bijection on [0, total)
  if (walk_n) { *bm = p / tiles_n; *bn = p - *bm * tiles_n; }
  else { *bn = p / tiles_m; *bm = p - *bn * tiles_m; }
}

template <class AL, class BL, int TA, int TB, int BM, int BN, int WAVES_M, int WAVES_N, int BF>
__global__ __launch_bounds__(256) void gemm_kernel(AL al, BL bl, GemmEpilogue ep, int K, long sAz, long sBz, long sCz, long sAscale,
                                                   int tiles_m, int tiles_n, int walk_n, int split_k, int k_chunk) {
  using T = GemmTile<TA, TB, BM, BN, WAVES_M, WAVES_N>;
  constexpr int LBF = BF >= 2 ? 2 : BF;                        // LDS image: f32 tiles (0, 1) or three bf16 planes (2, 3)
  __shared__ __attribute__((aligned(16))) float As[T::template a_lds<LBF>()];
  __shared__ __attribute__((aligned(16))) float Bs[T::template b_lds<LBF>()];
  const int z = blockIdx.z / split_k, zs = blockIdx.z % split_k;
  al.offset_z((long)z * sAz, (long)z * sAscale);
  bl.offset_z((long)z * sBz, 0);
  ep.C += (long)z * sCz;
  if ((z != 0 && sCz == 0) || zs != 0) ep.bias = nullptr;  // split-K: bias is added once
  const int kbeg = zs * k_chunk, kend = min(K, kbeg + k_chunk);
  if (kbeg >= K && !(K == 0 && zs == 0)) return;
  int bm, bn;
  tile_of_block(tiles_m, tiles_n, walk_n, &bm, &bn);
  if constexpr (BF == 2) T::template run_split<9>(al, bl, ep, kbeg, kend, bm * BM, bn * BN, As, Bs);
  else if constexpr (BF == 3) T::template run_split<6>(al, bl, ep, kbeg, kend, bm * BM, bn * BN, As, Bs);
  else T::template run<BF>(al, bl, ep, kbeg, kend, bm * BM, bn * BN, As, Bs);
}

struct GemmPlan {
  const asr_gemm_desc* d;
  const float* A; const float* B;
  int a_rows, a_cols, b_rows, b_cols;
  GemmEpilogue ep;
  hipStream_t st;
};

template <class AL, class BL, int TA, int TB, int BM, int BN, int WAVES_M, int WAVES_N>
static void launch_cfg(const GemmPlan& g, const AL& al, const BL& bl) {
  const asr_gemm_desc* d = g.d;
  const int tm = asr_cdiv(d->M, BM), tn = asr_cdiv(d->N, BN);
  const int sk = d->split_k > 1 ? d->split_k : 1;
  int k_chunk = asr_cdiv(asr_cdiv(d->K, sk), GEMM_BK) * GEMM_BK;  // BK-aligned partitions
  if (k_chunk <= 0) k_chunk = GEMM_BK;
  // the operand with fewer bytes stays L2 resident; walk its tiles fastest
  const int walk_n = ((long)d->K * d->N <= (long)d->M * d->K) ? 1 : 0;
  dim3 grid((unsigned)(tm * tn), 1, (unsigned)(d->batch * sk));
  hipLaunchKernelGGL((gemm_kernel<AL, BL, TA, TB, BM, BN, WAVES_M, WAVES_N, GEMM_BF>), grid, dim3(256), 0, g.st, al, bl, g.ep, d->K, d->stride_a,
                     d->stride_b, d->stride_c, d->stride_a_scale, tm, tn, walk_n, sk, k_chunk);
}

template <class AL, class BL, int TA, int TB>
static void launch_t(const GemmPlan& g, const AL& al, const BL& bl) {
  const asr_gemm_desc* d = g.d;
  if (d->N <= 32) { launch_cfg<AL, BL, TA, TB, 256, 32, 4, 1>(g, al, bl); return; }
  if (d->M <= 32) { launch_cfg<AL, BL, TA, TB, 32, 256, 1, 4>(g, al, bl); return; }
  // pick the tile by a wave-quantisation model: workgroups are dealt over 256 CUs in rounds of
  // (up to) 3 resident per CU; bigger tiles have the higher per-flop efficiency (less LDS staging)
  const long z = (long)d->batch * (d->split_k > 1 ? d->split_k : 1);
  auto score = [&](int bm, int bn, double tile_eff) {
    const long wgs = (long)asr_cdiv(d->M, bm) * asr_cdiv(d->N, bn) * z;
    const long rounds = (wgs + 255) / 256;                 // per-CU sequential tiles (at equal sharing)
    const double useful = (double)d->M * d->N * z;         // useful output elements
    // fewer than ~1.5 workgroups per CU: nothing overlaps a workgroup's prologue / epilogue (measured: proj
    // 7968x512x512 56 -> 51 us, keys 32 -> 28 us, vocab dY 178 -> 165 us with the next smaller tile)
    const double lonely = wgs < 384 ? 0.88 : 1.0;
    return lonely * tile_eff * useful / ((double)rounds * 256 * bm * bn);
  };
  static const int forced = getenv("ASR_GEMM_TILE") ? atoi(getenv("ASR_GEMM_TILE")) : 0;   // tuning aid: 1..4
  if (forced == 1) { launch_cfg<AL, BL, TA, TB, 128, 128, 2, 2>(g, al, bl); return; }
  if (forced == 2) { launch_cfg<AL, BL, TA, TB, 128, 64, 2, 2>(g, al, bl); return; }
  if (forced == 3) { launch_cfg<AL, BL, TA, TB, 64, 128, 2, 2>(g, al, bl); return; }
  if (forced == 4) { launch_cfg<AL, BL, TA, TB, 64, 64, 2, 2>(g, al, bl); return; }
  // (a 256 x 128 tile for the bf16 path spills: 72-93 TF against 314-414 TF with 128 x 128 on the las_large shapes)
  // per-tile efficiency: the bigger tile stages less per flop, but its fixed cost per tile (first loads, the C write of all
  // co-resident tiles at once: ~7 K-iterations' worth at 128 x 128 against ~3 at 64 x 64, tests/tools/gemm_k_sweep.py) only
  // amortises over a long K loop.  With <= 32 K-iterations per workgroup (every product of the las_small / deepspeech steps:
  // K or K / split_k <= 1024) the small tile wins inside the training step (las_small: 4.4 -> 4.1 ms of GEMMs)
  const int sk1 = d->split_k > 1 ? d->split_k : 1;
  const int k_iters = asr_cdiv(asr_cdiv(d->K, sk1), GEMM_BK);
  static const double e64s = getenv("ASR_GEMM_E64") ? atof(getenv("ASR_GEMM_E64")) : 1.02;
  double e128 = 1.0, e64 = k_iters <= 32 ? e64s : 0.80, e128x64 = k_iters <= 32 ? 0.97 : 0.92, e64x128 = e128x64;
  if (GEMM_BF >= 2) {
    // the split evaluation has 2.7x less matrix time per K tile to hide the same staging behind: a K tile costs LDS traffic rather than
    // MFMA issue, and a wave that owns two output tiles reads 3/4 of the fragments per MFMA that a one-tile wave reads.  Measured on
    // the step's shapes (tests/tools/bench_gemm_modes.py with ASR_GEMM_TILE = 1..4): 128 x 64 wins the short-K products (encoder
    // input projection 75 -> 66 us, vocabulary projection 147 -> 131), 128 x 128 the long contractions, 64 x 64 only the small ones
    e128 = k_iters <= 32 ? 0.97 : 1.0; e128x64 = 1.0; e64x128 = 0.97; e64 = 0.90;
  }
  const double s128 = score(128, 128, e128), s12864 = score(128, 64, e128x64), s64128 = score(64, 128, e64x128), s64 = score(64, 64, e64);
  if (s128 >= s12864 && s128 >= s64128 && s128 >= s64) launch_cfg<AL, BL, TA, TB, 128, 128, 2, 2>(g, al, bl);
  else if (s12864 >= s64128 && s12864 >= s64) launch_cfg<AL, BL, TA, TB, 128, 64, 2, 2>(g, al, bl);
  else if (s64128 >= s64) launch_cfg<AL, BL, TA, TB, 64, 128, 2, 2>(g, al, bl);
  else launch_cfg<AL, BL, TA, TB, 64, 64, 2, 2>(g, al, bl);
}

template <class AL, class BL>
static void launch_l(const GemmPlan& g, const AL& al, const BL& bl) {
  const asr_gemm_desc* d = g.d;
  if (!d->trans_a && !d->trans_b) launch_t<AL, BL, 0, 0>(g, al, bl);
  else if (!d->trans_a && d->trans_b) launch_t<AL, BL, 0, 1>(g, al, bl);
  else if (d->trans_a && !d->trans_b) launch_t<AL, BL, 1, 0>(g, al, bl);
  else launch_t<AL, BL, 1, 1>(g, al, bl);
}

static inline bool aligned16(const void* p, long ld, long stride) {
  return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0) && (stride % 4 == 0);
}

#if GEMM_BF == 1
int asr_gemm_launch_bf16(const GemmPlan& g) {
#elif GEMM_BF == 2
int asr_gemm_launch_split9(const GemmPlan& g) {
#elif GEMM_BF == 3
int asr_gemm_launch_split6(const GemmPlan& g) {
#else
int asr_gemm_launch_bf16(const GemmPlan& g);
int asr_gemm_launch_split9(const GemmPlan& g);
int asr_gemm_launch_split6(const GemmPlan& g);
static int asr_gemm_launch_f32(const GemmPlan& g) {
#endif
  const asr_gemm_desc* d = g.d;
  const float* A = g.A;
  const float* B = g.B;
  const bool a_fast = d->K > 0 && aligned16(A, d->lda, d->stride_a) && g.a_cols % 4 == 0 && g.a_cols >= 4;
  const bool b_fast = d->K > 0 && aligned16(B, d->ldb, d->stride_b) && g.b_cols % 4 == 0 && g.b_cols >= 4;
  // (rows-per-group 1 has no 32-bit reciprocal: ceil(2^32 / 1) wraps to 0 - it takes the plain loader)
  const bool s_fast = d->a_scale && (((uintptr_t)d->a_scale & 15) == 0) && d->stride_a_scale % 4 == 0 && d->a_rpg > 1 &&
                      (long)g.a_rows * d->a_rpg < 4294967296L;
  if (a_fast && b_fast && !d->a_scale) {
    launch_l(g, FastLoader{A, d->lda, g.a_rows, g.a_cols}, FastLoader{B, d->ldb, g.b_rows, g.b_cols});
  } else if (a_fast && b_fast && s_fast) {
    const uint32_t magic = (uint32_t)((4294967296ULL + (uint64_t)d->a_rpg - 1) / (uint64_t)d->a_rpg);
    launch_l(g, FastScaledLoader{A, d->lda, g.a_rows, g.a_cols, d->a_scale, magic}, FastLoader{B, d->ldb, g.b_rows, g.b_cols});
  } else {
    PlainLoader al{A, d->lda, g.a_rows, g.a_cols, aligned16(A, d->lda, d->stride_a), d->a_scale, d->a_rpg};
    PlainLoader bl{B, d->ldb, g.b_rows, g.b_cols, aligned16(B, d->ldb, d->stride_b), nullptr, 1};
    launch_l(g, al, bl);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}


#if !GEMM_BF
extern "C" int asr_gemm_f32(const asr_gemm_desc* d, const float* A, const float* B, float* C, void* stream) {
  ASR_CHECK(d && A && B && C, ASR_ERR_ARG, "asr_gemm_f32: null argument");
  ASR_CHECK(d->M > 0 && d->N > 0 && d->K >= 0 && d->batch >= 1, ASR_ERR_SHAPE, "asr_gemm_f32: bad M/N/K/batch %d %d %d %d",
            d->M, d->N, d->K, d->batch);
  GemmPlan g;
  g.d = d; g.A = A; g.B = B; g.st = (hipStream_t)stream;
  g.a_rows = d->trans_a ? d->K : d->M; g.a_cols = d->trans_a ? d->M : d->K;
  g.b_rows = d->trans_b ? d->N : d->K; g.b_cols = d->trans_b ? d->K : d->N;
  ASR_CHECK(d->lda >= g.a_cols && d->ldb >= g.b_cols && d->ldc >= d->N, ASR_ERR_SHAPE,
            "asr_gemm_f32: leading dimension smaller than row length (lda %ld ldb %ld ldc %ld)", d->lda, d->ldb, d->ldc);
  ASR_CHECK((long)d->batch * (d->split_k > 1 ? d->split_k : 1) <= 65535, ASR_ERR_SHAPE, "asr_gemm_f32: batch*split_k > 65535");
  ASR_CHECK(!(d->split_k > 1 && !d->accumulate), ASR_ERR_ARG, "asr_gemm_f32: split_k > 1 accumulates atomically: set accumulate and pre-zero C");
  ASR_CHECK(d->compute >= 0 && d->compute <= 3, ASR_ERR_ARG,
            "asr_gemm_f32: compute must be 0 (f32 MFMA), 1 (bf16 operands), 2 (f32 as nine bf16 pair products) or 3 (six), got %d", d->compute);
  ASR_CHECK(!(d->a_scale && d->a_rpg <= 0) && !(d->c_scale && d->c_rpg <= 0), ASR_ERR_ARG,
            "asr_gemm_f32: group scale needs rows-per-group > 0");
  int mode = d->accumulate ? 1 : 0;
  if (d->batch > 1 && d->stride_c == 0) {
    ASR_CHECK(d->accumulate, ASR_ERR_ARG, "asr_gemm_f32: batch>1 with stride_c==0 (split-K) requires accumulate=1");
    mode = 2;
  }
  if (d->accumulate == 2 || d->split_k > 1) mode = 2;
  g.ep = GemmEpilogue{C, d->ldc, d->M, d->N, d->alpha, d->bias, d->c_scale, d->c_rpg, mode, d->relu, nullptr, 0u, 0.f};

  if (d->compute == 1) return asr_gemm_launch_bf16(g);
  if (d->compute == 2) return asr_gemm_launch_split9(g);
  if (d->compute == 3) return asr_gemm_launch_split6(g);
  return asr_gemm_launch_f32(g);
}
#endif
