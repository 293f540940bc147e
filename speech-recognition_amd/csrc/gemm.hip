// asr_gemm_f32: general f32 GEMM on the f32-input MFMA (see gemm_core.h).
#include "gemm_core.h"

template <int TA, int TB, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void gemm_kernel(PlainLoader al, PlainLoader bl, GemmEpilogue ep, int K,
                                                   long sAz, long sBz, long sCz, long sAscale, int tiles_m,
                                                   int split_k, int k_chunk) {
  using T = GemmTile<TA, TB, BM, BN, WAVES_M, WAVES_N>;
  __shared__ __attribute__((aligned(16))) float As[T::A_ELEMS];
  __shared__ __attribute__((aligned(16))) float Bs[T::B_ELEMS];
  const int z = blockIdx.z / split_k, zs = blockIdx.z % split_k;
  al.p += (long)z * sAz;
  bl.p += (long)z * sBz;
  ep.C += (long)z * sCz;
  if (al.scale != nullptr) al.scale += (long)z * sAscale;
  if ((z != 0 && sCz == 0) || zs != 0) ep.bias = nullptr;  // split-K: bias is added once
  const int kbeg = zs * k_chunk, kend = min(K, kbeg + k_chunk);
  if (kbeg >= K && !(K == 0 && zs == 0)) return;
  // consecutive blocks walk M first: neighbours share the same B (weight) tile in L2
  const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;
  T::run(al, bl, ep, kbeg, kend, bm * BM, bn * BN, As, Bs);
}

template <int TA, int TB, int BM, int BN, int WAVES_M, int WAVES_N>
static void launch_cfg(const PlainLoader& al, const PlainLoader& bl, const GemmEpilogue& ep, const asr_gemm_desc* d,
                       hipStream_t st) {
  const int tm = asr_cdiv(d->M, BM), tn = asr_cdiv(d->N, BN);
  const int sk = d->split_k > 1 ? d->split_k : 1;
  const int k_chunk = asr_cdiv(asr_cdiv(d->K, sk), GEMM_BK) * GEMM_BK;  // BK-aligned partitions
  dim3 grid((unsigned)(tm * tn), 1, (unsigned)(d->batch * sk));
  hipLaunchKernelGGL((gemm_kernel<TA, TB, BM, BN, WAVES_M, WAVES_N>), grid, dim3(256), 0, st, al, bl, ep, d->K,
                     d->stride_a, d->stride_b, d->stride_c, d->stride_a_scale, tm, sk, k_chunk > 0 ? k_chunk : GEMM_BK);
}

template <int TA, int TB>
static void launch_t(const PlainLoader& al, const PlainLoader& bl, const GemmEpilogue& ep, const asr_gemm_desc* d,
                     hipStream_t st) {
  const long big = (long)asr_cdiv(d->M, 128) * asr_cdiv(d->N, 128) * d->batch * (d->split_k > 1 ? d->split_k : 1);
  if (d->N <= 32) launch_cfg<TA, TB, 256, 32, 4, 1>(al, bl, ep, d, st);
  else if (d->M <= 32) launch_cfg<TA, TB, 32, 256, 1, 4>(al, bl, ep, d, st);
  else if (big >= 192) launch_cfg<TA, TB, 128, 128, 2, 2>(al, bl, ep, d, st);
  else launch_cfg<TA, TB, 64, 64, 2, 2>(al, bl, ep, d, st);
}

static inline int aligned16(const void* p, long ld) { return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0); }

extern "C" int asr_gemm_f32(const asr_gemm_desc* d, const float* A, const float* B, float* C, void* stream) {
  ASR_CHECK(d && A && B && C, ASR_ERR_ARG, "asr_gemm_f32: null argument");
  ASR_CHECK(d->M > 0 && d->N > 0 && d->K >= 0 && d->batch >= 1, ASR_ERR_SHAPE, "asr_gemm_f32: bad M/N/K/batch %d %d %d %d",
            d->M, d->N, d->K, d->batch);
  const int a_rows = d->trans_a ? d->K : d->M, a_cols = d->trans_a ? d->M : d->K;
  const int b_rows = d->trans_b ? d->N : d->K, b_cols = d->trans_b ? d->K : d->N;
  ASR_CHECK(d->lda >= a_cols && d->ldb >= b_cols && d->ldc >= d->N, ASR_ERR_SHAPE,
            "asr_gemm_f32: leading dimension smaller than row length (lda %ld ldb %ld ldc %ld)", d->lda, d->ldb, d->ldc);
  ASR_CHECK((long)d->batch * (d->split_k > 1 ? d->split_k : 1) <= 65535, ASR_ERR_SHAPE, "asr_gemm_f32: batch*split_k > 65535");
  ASR_CHECK(!(d->split_k > 1 && !d->accumulate), ASR_ERR_ARG, "asr_gemm_f32: split_k > 1 accumulates atomically: set accumulate and pre-zero C");
  ASR_CHECK(!(d->a_scale && d->a_rpg <= 0) && !(d->c_scale && d->c_rpg <= 0), ASR_ERR_ARG,
            "asr_gemm_f32: group scale needs rows-per-group > 0");
  hipStream_t st = (hipStream_t)stream;
  PlainLoader al{A, d->lda, a_rows, a_cols, aligned16(A, d->lda) && (d->stride_a % 4 == 0), d->a_scale, d->a_rpg};
  PlainLoader bl{B, d->ldb, b_rows, b_cols, aligned16(B, d->ldb) && (d->stride_b % 4 == 0), nullptr, 1};
  int mode = d->accumulate ? 1 : 0;
  if (d->batch > 1 && d->stride_c == 0) {
    ASR_CHECK(d->accumulate, ASR_ERR_ARG, "asr_gemm_f32: batch>1 with stride_c==0 (split-K) requires accumulate=1");
    mode = 2;
  }
  if (d->accumulate == 2 || d->split_k > 1) mode = 2;
  GemmEpilogue ep{C, d->ldc, d->M, d->N, d->alpha, d->bias, d->c_scale, d->c_rpg, mode, d->relu, nullptr, 0u, 0.f};
  if (d->K == 0) {
    // nothing to accumulate; for store mode the result is bias only - still run with K=0 (loop skipped)
  }
  if (!d->trans_a && !d->trans_b) launch_t<0, 0>(al, bl, ep, d, st);
  else if (!d->trans_a && d->trans_b) launch_t<0, 1>(al, bl, ep, d, st);
  else if (d->trans_a && !d->trans_b) launch_t<1, 0>(al, bl, ep, d, st);
  else launch_t<1, 1>(al, bl, ep, d, st);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
