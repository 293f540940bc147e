// Memory-bound layer kernels of the LAS / DeepSpeech2 training step (gfx950): frame mask,
// BatchNormalization(+ReLU) forward/backward, column sums (bias gradients), embedding gather /
// scatter-add, row-wise dropout, fills.  All are one or two passes over their tensors with
// 16-byte accesses where the layout allows; reductions are wave shuffles + one atomic per block.
#include "common.h"

// ------------------------------------------------------------------------------------------ fill / scale
__global__ void fill_kernel(float* p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
extern "C" int asr_fill_f32(float* p, long n, float value, void* stream) {
  ASR_CHECK(p && n >= 0, ASR_ERR_ARG, "asr_fill_f32: bad argument");
  if (n == 0) return ASR_OK;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)min((long)2048, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, n, value);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// ------------------------------------------------------------------------------------------ frame mask
// Listener._audio_mask (las.py:205-217) / Convolution._audio_mask (deepspeech2.py:68-78):
// out[b][j] = any over frames [j*group, (j+1)*group) of any(x[b, frame, :] != 0.0).  One wave per j.
__global__ __launch_bounds__(256) void frame_mask_kernel(const float* x, int B, int T, int FC, int group, int Tout, uint8_t* out) {
  const int lane = threadIdx.x & 63;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wid >= (long)B * Tout) return;
  const int b = (int)(wid / Tout), j = (int)(wid % Tout);
  const float* p = x + ((long)b * T + (long)j * group) * FC;
  const long n = (long)group * FC;
  bool any = false;
  for (long i = lane; i < n; i += 64) any |= (p[i] != 0.0f);
  const unsigned long long bal = __ballot(any);
  if (lane == 0) out[wid] = bal != 0ull ? 1 : 0;
}
extern "C" int asr_frame_mask(const float* x, int B, int T, int FC, int group, int Tout, uint8_t* out, void* stream) {
  ASR_CHECK(x && out, ASR_ERR_ARG, "asr_frame_mask: null argument");
  ASR_CHECK(B > 0 && T > 0 && FC > 0 && group > 0 && Tout >= 0 && (long)Tout * group <= T, ASR_ERR_SHAPE,
            "asr_frame_mask: Tout*group (%d*%d) exceeds T=%d", Tout, group, T);
  if (Tout == 0) return ASR_OK;
  const long waves = (long)B * Tout;
  hipLaunchKernelGGL(frame_mask_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, B, T, FC, group, Tout, out);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// ------------------------------------------------------------------------------------------ column sums
// out[c] (+)= sum_r A[r][c]  (bias gradients).  Block = 64 columns x 4 row-lanes; atomics across row chunks.
__global__ __launch_bounds__(256) void colsum_kernel(const float* A, int M, int N, long lda, float* out, int rows_per_block) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (c < N) {
    // eight rows in flight per thread: the loop is a chain of dependent-looking loads otherwise (one memory round trip per row:
    // 14.6 us for a [2688 x 384] bias gradient that is 4 MB of data)
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    int r = r0 + w;
    for (; r + 28 < r1; r += 32) {
      const float* q = A + (long)r * lda + c;
      const float v0 = q[0], v1 = q[4 * lda], v2 = q[8 * lda], v3 = q[12 * lda], v4 = q[16 * lda], v5 = q[20 * lda], v6 = q[24 * lda], v7 = q[28 * lda];
      p0 += v0 + v4; p1 += v1 + v5; p2 += v2 + v6; p3 += v3 + v7;
    }
    for (; r < r1; r += 4) p0 += A[(long)r * lda + c];
    s = (p0 + p1) + (p2 + p3);
  }
  red[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < N) atomicAdd(&out[c], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
extern "C" int asr_colsum(const float* A, int M, int N, long lda, float* out, void* stream) {
  ASR_CHECK(A && out && M > 0 && N > 0 && lda >= N, ASR_ERR_ARG, "asr_colsum: bad argument");
  const int rpb = 128;                                    // (32 rows per block: 26 us - the atomics of four times as many blocks cost more than the shorter chains save)
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)asr_cdiv(N, 64), (unsigned)asr_cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, A, M, N,
                     lda, out, rpb);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// ------------------------------------------------------------------------------------------ one-column products of the attention (las.py:46-59)
// The hoisted attention has three products with a dimension of ONE - s0 = K bq, d bq = K^T ds0 and the rank-1 term ds0 (x) bq of dK.
// Through the MFMA GEMM each of them is a launch of 24-42 us that moves 8 MB; as memory-bound kernels they take 5-8.
// out[c] (+)= sum_r w[r] A[r][c]: colsum_kernel with a weight per row
__global__ __launch_bounds__(256) void colsum_w_kernel(const float* A, int M, int N, long lda, const float* w, float* out, int rows_per_block) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), wv = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (c < N) {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    int r = r0 + wv;
    for (; r + 12 < r1; r += 16) {
      const float* q = A + (long)r * lda + c;
      const float v0 = q[0], v1 = q[4 * lda], v2 = q[8 * lda], v3 = q[12 * lda];
      p0 = fmaf(w[r], v0, p0); p1 = fmaf(w[r + 4], v1, p1); p2 = fmaf(w[r + 8], v2, p2); p3 = fmaf(w[r + 12], v3, p3);
    }
    for (; r < r1; r += 4) p0 = fmaf(w[r], A[(long)r * lda + c], p0);
    s = (p0 + p1) + (p2 + p3);
  }
  red[wv][threadIdx.x & 63] = s;
  __syncthreads();
  if (wv == 0 && c < N) atomicAdd(&out[c], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
extern "C" int asr_colsum_weighted(const float* A, int M, int N, long lda, const float* w, float* out, void* stream) {
  ASR_CHECK(A && w && out && M > 0 && N > 0 && lda >= N, ASR_ERR_ARG, "asr_colsum_weighted: bad argument");
  const int rpb = 128;
  hipLaunchKernelGGL(colsum_w_kernel, dim3((unsigned)asr_cdiv(N, 64), (unsigned)asr_cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, A, M, N, lda, w,
                     out, rpb);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
// y[r] = A[r][:] . x  (one wave per row, 4 rows per workgroup)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* A, int M, int K, long lda, const float* x, float* y) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= M) return;
  const float* a = A + (long)r * lda;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s = fmaf(a[k], x[k], s);
  s = wave_sum(s);
  if (lane == 0) y[r] = s;
}
extern "C" int asr_rowdot(const float* A, int M, int K, long lda, const float* x, float* y, void* stream) {
  ASR_CHECK(A && x && y && M > 0 && K > 0 && lda >= K, ASR_ERR_ARG, "asr_rowdot: bad argument");
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)asr_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, A, M, K, lda, x, y);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
// C[r][c] += u[r] v[c]
__global__ __launch_bounds__(256) void rank1_add_kernel(float* C, int M, int N, long ldc, const float* u, const float* v) {
  const long n4 = N >> 2, total = (long)M * n4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / n4, c = (i % n4) * 4;
    float4* dst = reinterpret_cast<float4*>(C + r * ldc + c);
    const float4 vv = *reinterpret_cast<const float4*>(v + c);
    const float ur = u[r];
    float4 t = *dst;
    t.x = fmaf(ur, vv.x, t.x); t.y = fmaf(ur, vv.y, t.y); t.z = fmaf(ur, vv.z, t.z); t.w = fmaf(ur, vv.w, t.w);
    *dst = t;
  }
}
extern "C" int asr_rank1_add(float* C, int M, int N, long ldc, const float* u, const float* v, void* stream) {
  ASR_CHECK(C && u && v && M > 0 && N > 0 && N % 4 == 0 && ldc >= N && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)v & 15) == 0, ASR_ERR_ARG,
            "asr_rank1_add: bad argument (N and ldc multiples of 4, 16-byte aligned C and v)");
  const long total = (long)M * (N >> 2);
  const unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(rank1_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, C, M, N, ldc, u, v);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// ------------------------------------------------------------------------------------------ batch norm
// BatchNormalization(axis=-1, eps=1e-3, momentum=0.99) in training mode (las.py:170,193;
// deepspeech2.py:112,118): biased batch statistics over all M rows (padded frames included).
// pass 1: per-column sum / sum of squares accumulated in double (atomics across row chunks)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* x, int M, int C, long ld, double* stats, int rows_per_block) {
  __shared__ double red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  double s = 0.0, q = 0.0;
  if (c < C) {
    int r = r0 + w;
    for (; r + 12 < r1; r += 16) {                        // four rows in flight per thread
      const float* p = x + (long)r * ld + c;
      const double v0 = p[0], v1 = p[4 * ld], v2 = p[8 * ld], v3 = p[12 * ld];
      s += (v0 + v1) + (v2 + v3);
      q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
    }
    for (; r < r1; r += 4) { const double v = x[(long)r * ld + c]; s += v; q += v * v; }
  }
  red[0][w][l] = s; red[1][w][l] = q;
  __syncthreads();
  if (w == 0 && c < C) {
    atomicAdd(&stats[c], red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l]);
    atomicAdd(&stats[C + c], red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l]);
  }
}
// pass 2: normalise (+ReLU), publish mean / rstd for backward, update the moving statistics
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* x, int M, int C, long ld, const double* stats, const float* gamma,
                                                       const float* beta, float eps, float momentum, int relu, float* y, long ldy,
                                                       float* mean_out, float* rstd_out, float* moving_mean, float* moving_var) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  if (c >= C) return;
  const double mu = stats[c] / M;
  double var = stats[C + c] / M - mu * mu;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma[c] * rstd, bb = beta[c] - mean * g;
  if (blockIdx.y == 0 && w == 0) {
    mean_out[c] = mean; rstd_out[c] = rstd;
    if (moving_mean) moving_mean[c] = moving_mean[c] * momentum + mean * (1.f - momentum);
    if (moving_var) moving_var[c] = moving_var[c] * momentum + (float)var * (1.f - momentum);
  }
  const int rows_per_block = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (int r = r0 + w; r < r1; r += 4) {
    float v = x[(long)r * ld + c] * g + bb;
    if (relu) v = fmaxf(v, 0.f);
    y[(long)r * ldy + c] = v;
  }
}
// inference mode: moving statistics
__global__ __launch_bounds__(256) void bn_infer_kernel(const float* x, int M, int C, long ld, const float* gamma, const float* beta,
                                                       const float* mm, const float* mv, float eps, int relu, float* y, long ldy) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  if (c >= C) return;
  const float g = gamma[c] / sqrtf(mv[c] + eps), bb = beta[c] - mm[c] * g;
  const int rows_per_block = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (int r = r0 + w; r < r1; r += 4) {
    float v = x[(long)r * ld + c] * g + bb;
    if (relu) v = fmaxf(v, 0.f);
    y[(long)r * ldy + c] = v;
  }
}
extern "C" int asr_bn_fwd(const float* x, int M, int C, long ld, const float* gamma, const float* beta, float eps, float momentum,
                          int relu, int training, float* y, long ldy, float* mean_out, float* rstd_out, float* moving_mean,
                          float* moving_var, double* stats_ws, void* stream) {
  ASR_CHECK(x && gamma && beta && y && M > 0 && C > 0, ASR_ERR_ARG, "asr_bn_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int rpb = 128;
  dim3 grid((unsigned)asr_cdiv(C, 64), (unsigned)asr_cdiv(M, rpb));
  if (training) {
    ASR_CHECK(mean_out && rstd_out && stats_ws, ASR_ERR_ARG, "asr_bn_fwd: training needs mean/rstd/stats buffers");
    if (asr_zero_async(stats_ws, sizeof(double) * 2 * C, st) != hipSuccess) { asr_set_error("asr_bn_fwd: memset failed"); return ASR_ERR_HIP; }
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, st, x, M, C, ld, stats_ws, rpb);
    hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(256), 0, st, x, M, C, ld, stats_ws, gamma, beta, eps, momentum, relu, y, ldy,
                       mean_out, rstd_out, moving_mean, moving_var);
  } else {
    ASR_CHECK(moving_mean && moving_var, ASR_ERR_ARG, "asr_bn_fwd: inference needs the moving statistics");
    hipLaunchKernelGGL(bn_infer_kernel, grid, dim3(256), 0, st, x, M, C, ld, gamma, beta, moving_mean, moving_var, eps, relu, y, ldy);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// backward: dyr = dy * (y > 0 if relu);  sums[c] = sum dyr, sums[C+c] = sum dyr * xhat
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* x, const float* y, const float* dy, int M, int C, long ld,
                                                           long ldy, long lddy, const float* mean, const float* rstd, int relu,
                                                           double* sums, int rows_per_block) {
  __shared__ double red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  double s = 0.0, q = 0.0;
  if (c < C) {
    const float mu = mean[c], rs = rstd[c];
#pragma unroll 4                                          // (the rows' loads in flight together: the loop is latency bound otherwise)
    for (int r = r0 + w; r < r1; r += 4) {
      float d = dy[(long)r * lddy + c];
      if (relu && !(y[(long)r * ldy + c] > 0.f)) d = 0.f;
      s += d; q += (double)d * ((x[(long)r * ld + c] - mu) * rs);
    }
  }
  red[0][w][l] = s; red[1][w][l] = q;
  __syncthreads();
  if (w == 0 && c < C) {
    atomicAdd(&sums[c], red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l]);
    atomicAdd(&sums[C + c], red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l]);
  }
}
// dx = gamma * rstd * (dyr - mean(dyr) - xhat * mean(dyr * xhat));  dgamma += sum dyr xhat;  dbeta += sum dyr
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* x, const float* y, const float* dy, int M, int C, long ld,
                                                           long ldy, long lddy, const float* mean, const float* rstd,
                                                           const float* gamma, int relu, const double* sums, float* dx, long lddx,
                                                           float* dgamma, float* dbeta) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  if (c >= C) return;
  const float mu = mean[c], rs = rstd[c], g = gamma[c];
  const float m1 = (float)(sums[c] / M), m2 = (float)(sums[C + c] / M);
  if (blockIdx.y == 0 && w == 0) {
    dbeta[c] += (float)sums[c];
    dgamma[c] += (float)sums[C + c];
  }
  const int rows_per_block = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
#pragma unroll 4
  for (int r = r0 + w; r < r1; r += 4) {
    float d = dy[(long)r * lddy + c];
    if (relu && !(y[(long)r * ldy + c] > 0.f)) d = 0.f;
    const float xh = (x[(long)r * ld + c] - mu) * rs;
    dx[(long)r * lddx + c] = g * rs * (d - m1 - xh * m2);
  }
}
extern "C" int asr_bn_bwd(const float* x, const float* y, const float* dy, int M, int C, long ld, long ldy, long lddy,
                          const float* mean, const float* rstd, const float* gamma, int relu, float* dx, long lddx, float* dgamma,
                          float* dbeta, double* sums_ws, void* stream) {
  ASR_CHECK(x && dy && mean && rstd && gamma && dx && dgamma && dbeta && sums_ws && (!relu || y), ASR_ERR_ARG, "asr_bn_bwd: null argument");
  hipStream_t st = (hipStream_t)stream;
  const int rpb = 128;
  dim3 grid((unsigned)asr_cdiv(C, 64), (unsigned)asr_cdiv(M, rpb));
  if (asr_zero_async(sums_ws, sizeof(double) * 2 * C, st) != hipSuccess) { asr_set_error("asr_bn_bwd: memset failed"); return ASR_ERR_HIP; }
  hipLaunchKernelGGL(bn_bwd_stats_kernel, grid, dim3(256), 0, st, x, y, dy, M, C, ld, ldy, lddy, mean, rstd, relu, sums_ws, rpb);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, st, x, y, dy, M, C, ld, ldy, lddy, mean, rstd, gamma, relu, sums_ws, dx, lddx,
                     dgamma, dbeta);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// ------------------------------------------------------------------------------------------ row-wise dropout
// y[r][k] = x[r][k] * mult(stream0 + stream_step * (r % period), (r / period) * idx_ld + idx_off + k).
// Decoder sites (las.py:278,291): rows r = b*U + i, one stream per step i, element index b*H + k.
struct RowDrop { uint32_t stream0, stream_step; int period; long idx_ld; int idx_off; float rate; };
__device__ __forceinline__ float rowdrop_mult(const RowDrop& d, uint32_t seed, int r, int k) {
  if (d.rate <= 0.f) return 1.f;
  // period > 0: rows are batch-major (r = b * period + step); period < 0: step-major (r = step * |period| + b)
  const int step = d.period > 0 ? r % d.period : r / (-d.period);
  const int b = d.period > 0 ? r / d.period : r % (-d.period);
  const AsrRngKey key = asr_rng_key(seed, d.stream0 + d.stream_step * (uint32_t)step);
  return asr_drop_mult(key, (uint32_t)((long)b * d.idx_ld + d.idx_off + k), asr_drop_threshold(d.rate), 1.f / (1.f - d.rate));
}
__global__ void dropout_rows_kernel(const float* x, long ldx, float* y, long ldy, int R, int K, const uint32_t* seed, RowDrop d) {
  const uint32_t sd = seed[0];
  const long n = (long)R * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / K), k = (int)(i % K);
    y[(long)r * ldy + k] = x[(long)r * ldx + k] * rowdrop_mult(d, sd, r, k);
  }
}
extern "C" int asr_dropout_rows(const float* x, long ldx, float* y, long ldy, int R, int K, const uint32_t* seed, uint32_t stream0,
                                uint32_t stream_step, int period, long idx_ld, int idx_off, float rate, void* stream) {
  ASR_CHECK(x && y && seed && R > 0 && K > 0 && period != 0, ASR_ERR_ARG, "asr_dropout_rows: bad argument");
  RowDrop d{stream0, stream_step, period, idx_ld, idx_off, rate};
  const long n = (long)R * K;
  hipLaunchKernelGGL(dropout_rows_kernel, dim3((unsigned)min((long)2048, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy,
                     R, K, seed, d);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// flat in-place dropout on a contiguous tensor (gradient of a Keras Dropout layer): x[i] *= mult(stream, i)
__global__ void dropout_flat_kernel(float* x, long n, const uint32_t* seed, uint32_t stream_id, float rate) {
  const AsrRngKey key = asr_rng_key(seed[0], stream_id);
  const uint32_t thr = asr_drop_threshold(rate);
  const float sc = 1.f / (1.f - rate);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    x[i] *= asr_drop_mult(key, (uint32_t)i, thr, sc);
}
extern "C" int asr_dropout_flat(float* x, long n, const uint32_t* seed, uint32_t stream_id, float rate, void* stream) {
  ASR_CHECK(x && seed && n >= 0, ASR_ERR_ARG, "asr_dropout_flat: bad argument");
  if (n == 0 || rate <= 0.f) return ASR_OK;
  hipLaunchKernelGGL(dropout_flat_kernel, dim3((unsigned)min((long)4096, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, seed,
                     stream_id, rate);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// [B, D] inverted-dropout multiplier table (Keras RNN input dropout: one mask per batch row, constant
// over time, las.py:94,102) -> consumed by the GEMM's a_scale / c_scale group scaling
__global__ void dropout_table_kernel(float* out, long n, const uint32_t* seed, uint32_t stream_id, float rate) {
  const AsrRngKey key = asr_rng_key(seed[0], stream_id);
  const uint32_t thr = asr_drop_threshold(rate);
  const float sc = 1.f / (1.f - rate);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = asr_drop_mult(key, (uint32_t)i, thr, sc);
}
extern "C" int asr_dropout_table(float* out, long n, const uint32_t* seed, uint32_t stream_id, float rate, void* stream) {
  ASR_CHECK(out && seed && n > 0 && rate > 0.f && rate < 1.f, ASR_ERR_ARG, "asr_dropout_table: bad argument");
  hipLaunchKernelGGL(dropout_table_kernel, dim3((unsigned)min((long)1024, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n, seed,
                     stream_id, rate);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// several tables in one launch (blockIdx.y = table): the input-dropout tables of all BiRNN layers of a step depend on the seed only
#define ASR_TABLES_MANY 16
struct DropTables { float* out[ASR_TABLES_MANY]; long n[ASR_TABLES_MANY]; uint32_t stream_id[ASR_TABLES_MANY]; float rate[ASR_TABLES_MANY]; };
__global__ void dropout_tables_kernel(DropTables t, const uint32_t* seed) {
  const int k = blockIdx.y;
  const AsrRngKey key = asr_rng_key(seed[0], t.stream_id[k]);
  const uint32_t thr = asr_drop_threshold(t.rate[k]);
  const float sc = 1.f / (1.f - t.rate[k]);
  float* out = t.out[k];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < t.n[k]; i += (long)gridDim.x * blockDim.x)
    out[i] = asr_drop_mult(key, (uint32_t)i, thr, sc);
}
extern "C" int asr_dropout_tables(int ntables, float* const* outs, const long* ns, const uint32_t* stream_ids, const float* rates,
                                  const uint32_t* seed, void* stream) {
  ASR_CHECK(outs && ns && stream_ids && rates && seed && ntables > 0, ASR_ERR_ARG, "asr_dropout_tables: null argument");
  for (int t0 = 0; t0 < ntables; t0 += ASR_TABLES_MANY) {
    DropTables t{};
    const int n = ntables - t0 < ASR_TABLES_MANY ? ntables - t0 : ASR_TABLES_MANY;
    long nmax = 0;
    for (int k = 0; k < n; ++k) {
      ASR_CHECK(outs[t0 + k] && ns[t0 + k] > 0 && rates[t0 + k] > 0.f && rates[t0 + k] < 1.f, ASR_ERR_ARG, "asr_dropout_tables: bad table %d", t0 + k);
      t.out[k] = outs[t0 + k]; t.n[k] = ns[t0 + k]; t.stream_id[k] = stream_ids[t0 + k]; t.rate[k] = rates[t0 + k];
      nmax = ns[t0 + k] > nmax ? ns[t0 + k] : nmax;
    }
    hipLaunchKernelGGL(dropout_tables_kernel, dim3((unsigned)min((long)64, (nmax + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, t, seed);
    ASR_LAUNCH_CHECK();
  }
  return ASR_OK;
}

// ------------------------------------------------------------------------------------------ embedding
// Embedding gather (las.py:258,278) fused with up to two dropout sites; one wave per row.
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const float* E, const int32_t* tok, int R, int Hd, int V, float* out, long ldo,
                                                            const uint32_t* seed, RowDrop d1, RowDrop d2) {
  const int lane = threadIdx.x & 63;
  const long r = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= R) return;
  int t = tok[r];
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);
  const uint32_t sd = seed ? seed[0] : 0u;
  for (int k = lane; k < Hd; k += 64) {
    float v = E[(long)t * Hd + k];
    if (seed) v = v * rowdrop_mult(d1, sd, (int)r, k) * rowdrop_mult(d2, sd, (int)r, k);
    out[r * ldo + k] = v;
  }
}
// scatter-add of the (dropout-masked) input gradient into dE; float atomics, 256 contiguous bytes per wave op
__global__ __launch_bounds__(256) void embedding_bwd_kernel(float* dE, const int32_t* tok, int R, int Hd, int V, const float* dx, long ldd,
                                                            const uint32_t* seed, RowDrop d1, RowDrop d2) {
  const int lane = threadIdx.x & 63;
  const long r = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= R) return;
  int t = tok[r];
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);
  const uint32_t sd = seed ? seed[0] : 0u;
  for (int k = lane; k < Hd; k += 64) {
    float v = dx[r * ldd + k];
    if (seed) v = v * rowdrop_mult(d1, sd, (int)r, k) * rowdrop_mult(d2, sd, (int)r, k);
    atomicAdd(&dE[(long)t * Hd + k], v);
  }
}
extern "C" int asr_embedding(int backward, float* E_or_dE, const int32_t* tok, int R, int Hd, int V, float* x_or_dx, long ld,
                             const uint32_t* seed, const asr_rowdrop* drop1, const asr_rowdrop* drop2, void* stream) {
  ASR_CHECK(E_or_dE && tok && x_or_dx && R > 0 && Hd > 0 && V > 0, ASR_ERR_ARG, "asr_embedding: bad argument");
  RowDrop d1{0, 0, 1, 0, 0, 0.f}, d2{0, 0, 1, 0, 0, 0.f};
  if (drop1) d1 = RowDrop{drop1->stream0, drop1->stream_step, drop1->period, drop1->idx_ld, drop1->idx_off, drop1->rate};
  if (drop2) d2 = RowDrop{drop2->stream0, drop2->stream_step, drop2->period, drop2->idx_ld, drop2->idx_off, drop2->rate};
  ASR_CHECK(d1.period != 0 && d2.period != 0, ASR_ERR_ARG, "asr_embedding: dropout period must be != 0");
  const bool any = (d1.rate > 0.f || d2.rate > 0.f);
  ASR_CHECK(!(any && !seed), ASR_ERR_ARG, "asr_embedding: dropout needs a device seed");
  dim3 grid((unsigned)((R + 3) / 4));
  if (!backward)
    hipLaunchKernelGGL(embedding_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)E_or_dE, tok, R, Hd, V, x_or_dx, ld,
                       any ? seed : nullptr, d1, d2);
  else
    hipLaunchKernelGGL(embedding_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, E_or_dE, tok, R, Hd, V, (const float*)x_or_dx, ld,
                       any ? seed : nullptr, d1, d2);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// out[r] = tokens[r*ld + col] (teacher forcing) or argmax of the previous logits row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* x, long ld, int R, int N, int32_t* out) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int r = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int c = threadIdx.x; c < N; c += 256) {
    const float v = x[(long)r * ld + c];
    if (v > best || (v == best && c < idx)) { best = v; idx = c; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if (lane == 0) { bv[w] = best; bi[w] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i)
      if (bv[i] > best || (bv[i] == best && bi[i] < idx)) { best = bv[i]; idx = bi[i]; }
    out[r] = idx == 0x7fffffff ? 0 : idx;
  }
}
extern "C" int asr_argmax_rows(const float* x, long ld, int R, int N, int32_t* out, void* stream) {
  ASR_CHECK(x && out && R > 0 && N > 0, ASR_ERR_ARG, "asr_argmax_rows: bad argument");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, x, ld, R, N, out);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// out[i * out_stride] = tok[i] != pad  (decoder step mask, las.py:276)
__global__ void token_mask_kernel(const int32_t* tok, long n, int pad, uint8_t* out, long out_stride) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i * out_stride] = tok[i] != pad ? 1 : 0;
}
extern "C" int asr_token_mask(const int32_t* tok, long n, int pad, uint8_t* out, long out_stride, void* stream) {
  ASR_CHECK(tok && out && n > 0 && out_stride > 0, ASR_ERR_ARG, "asr_token_mask: bad argument");
  hipLaunchKernelGGL(token_mask_kernel, dim3((unsigned)min((long)1024, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tok, n, pad, out,
                     out_stride);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}


// ------------------------------------------------------------------------------------------ bf16 weight images
// dst[i] = bf16(src[i]) (round to nearest even): the images the wide recurrent step kernels read under mixed precision.
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(src + i);
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 o;
      o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
      *reinterpret_cast<bf16x4*>(dst + i) = o;
    } else {
      for (long k = i; k < n; ++k) dst[k] = (__bf16)src[k];
    }
  }
}

extern "C" int asr_f32_to_bf16(const float* src, void* dst, long n, void* stream) {
  ASR_CHECK(src && dst, ASR_ERR_ARG, "asr_f32_to_bf16: null argument");
  ASR_CHECK(n > 0, ASR_ERR_SHAPE, "asr_f32_to_bf16: n must be > 0");
  ASR_CHECK((((uintptr_t)src & 15) == 0) && (((uintptr_t)dst & 7) == 0), ASR_ERR_ARG, "asr_f32_to_bf16: src must be 16-byte and dst 8-byte aligned");
  const long blocks = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)(blocks < 4096 ? (blocks > 0 ? blocks : 1) : 4096)), dim3(256), 0, (hipStream_t)stream, src,
                     static_cast<__bf16*>(dst), n);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// dst[i] = float(src[i]): the way back for gradient buckets that crossed the fabric as bf16 (training.GradientExchange)
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(src + i);
      *reinterpret_cast<float4*>(dst + i) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    } else {
      for (long k = i; k < n; ++k) dst[k] = (float)src[k];
    }
  }
}

extern "C" int asr_bf16_to_f32(const void* src, float* dst, long n, void* stream) {
  ASR_CHECK(src && dst, ASR_ERR_ARG, "asr_bf16_to_f32: null argument");
  ASR_CHECK(n > 0, ASR_ERR_SHAPE, "asr_bf16_to_f32: n must be > 0");
  ASR_CHECK((((uintptr_t)dst & 15) == 0) && (((uintptr_t)src & 7) == 0), ASR_ERR_ARG, "asr_bf16_to_f32: dst must be 16-byte and src 8-byte aligned");
  const long blocks = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)(blocks < 4096 ? (blocks > 0 ? blocks : 1) : 4096)), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const __bf16*>(src), dst, n);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// ------------------------------------------------------------------------------------------ diagnostics
// Keeps `blocks` workgroups of `threads` threads busy for `microseconds` (bounded by the 100 MHz real-time counter, so it
// always ends): a stand-in for a foreign kernel (an RCCL channel) that holds compute units while the one-launch sweeps run.
__global__ void occupy_kernel(long ticks, unsigned* sink) {
  const long t0 = (long)__builtin_amdgcn_s_memrealtime();
  unsigned spins = 0;
  while ((long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); ++spins; }
  if (sink != nullptr && threadIdx.x == 0 && blockIdx.x == 0) *sink = spins;
}

extern "C" int asr_debug_occupy(int blocks, int threads, int microseconds, void* stream) {
  ASR_CHECK(blocks > 0 && threads > 0 && threads <= 1024 && microseconds >= 0 && microseconds <= 2000000, ASR_ERR_ARG, "asr_debug_occupy: bad argument");
  hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), 0, (hipStream_t)stream, (long)microseconds * 100, nullptr);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// The memory-side co-tenant: `blocks` workgroups of 256 threads copy their slice of `buf` (16-byte loads and stores, `bytes` in
// all, first half -> second half) over and over until `microseconds` have passed (real-time counter: it always ends).  Stands in
// for the weight-gradient products / RCCL reductions whose memory traffic - not their compute units - exposed the two BPTT-sweep
// races of round 3.
__global__ __launch_bounds__(256) void stream_memory_kernel(float4* buf, long n4_half, long ticks) {
  const long t0 = (long)__builtin_amdgcn_s_memrealtime();
  const long per = (n4_half + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per, hi = lo + per < n4_half ? lo + per : n4_half;
  while ((long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    for (long i = lo + threadIdx.x; i < hi; i += 256 * 4) {
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const long j = i + (long)k * 256; v[k] = j < hi ? buf[j] : float4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int k = 0; k < 4; ++k) { const long j = i + (long)k * 256; if (j < hi) buf[n4_half + j] = v[k]; }
    }
  }
}

extern "C" int asr_debug_stream_memory(float* buf, long bytes, int blocks, int microseconds, void* stream) {
  ASR_CHECK(buf && bytes >= 64 && (((uintptr_t)buf & 15) == 0) && blocks > 0 && blocks <= 4096 && microseconds >= 0 && microseconds <= 2000000,
            ASR_ERR_ARG, "asr_debug_stream_memory: bad argument");
  hipLaunchKernelGGL(stream_memory_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(buf), bytes / 32,
                     (long)microseconds * 100);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
