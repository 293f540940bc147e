// Attention of the LAS decoder (las.py:43-59, called once per decoder step at las.py:282) for gfx950.
//
// The reference's "AdditiveAttention" is a projected dot product:
//     e[b,t] = (h[b] Wq + bq) . (enc[b,t] Wk + bk) - 1e9 (1 - mask[b,t]);  p = softmax_t(e);  ctx = p . enc
// and it re-projects the keys at every decoder step.  Here the loop-invariant parts are hoisted
// (batched GEMMs outside the step loop, see models/las.py):
//     K  = enc Wk + bk          [B,T,Hd]
//     Kq = K Wq^T               [B,T,Hd]     s0 = K bq   [B,T]
//     e[b,t] = h[b] . Kq[b,t] + s0[b,t] - 1e9 (1 - mask)
// so one step is two small kernels: scores (wave-level dot products over all 256 CUs) and
// softmax + context (block-level max/sum reductions, one workgroup per (batch row, 64-column slice)).
// The backward step mirrors them: dp = dctx . enc ; ds = p (dp - <p,dp>) ; dh = ds . Kq.
// These kernels are latency-bound (a step moves ~25 MB): every wave issues the loads of several
// independent rows before it reduces any of them.
// Two one-launch variants were built, verified and dropped: splitting T over workgroups with an in-launch combine
// (attention_fused.hip, opt-in: 17.5 us against 5 + 9 us) and, without any hand-off, one 16-wave workgroup per (row,
// column chunk) that recomputes the row's scores itself (2, 4 or 8 chunks: las_small 16.99 -> 17.12-17.14 ms per step):
// a single CU streaming its 320-510 KB through three dependent phases takes longer than the two wide launches plus
// the 1.7 us boundary between them.
#include <stdlib.h>

#include "common.h"

#define ATT_RPW 4   // rows per wave in the dot-product kernels
#define ATT_UNR 8   // independent loads in flight per lane in the weighted-sum kernels

// out[b,t] = v[b] . M[b,t,:] (+ s0[b,t]) (- 1e9 (1 - mask[b,t]))     grid (ceil(T/16), B)
template <bool VEC>
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const float* v, long ldv, const float* M, const float* s0, const uint8_t* mask,
                                                          int T, int K, float* out) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* vb = v + (long)b * ldv;
  const int t0 = blockIdx.x * (4 * ATT_RPW) + w * ATT_RPW;
  float s[ATT_RPW];
#pragma unroll
  for (int r = 0; r < ATT_RPW; ++r) s[r] = 0.f;
  if (VEC) {
    for (int k = lane * 4; k < K; k += 256) {
      const float4 hv = *reinterpret_cast<const float4*>(vb + k);
      float4 kv[ATT_RPW];
#pragma unroll
      for (int r = 0; r < ATT_RPW; ++r)
        kv[r] = (t0 + r < T) ? *reinterpret_cast<const float4*>(M + ((long)b * T + t0 + r) * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int r = 0; r < ATT_RPW; ++r) s[r] += hv.x * kv[r].x + hv.y * kv[r].y + hv.z * kv[r].z + hv.w * kv[r].w;
    }
  } else {
    for (int k = lane; k < K; k += 64) {
      const float hv = vb[k];
#pragma unroll
      for (int r = 0; r < ATT_RPW; ++r)
        if (t0 + r < T) s[r] = fmaf(hv, M[((long)b * T + t0 + r) * K + k], s[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < ATT_RPW; ++r) {
    const float tot = wave_sum(s[r]);
    const int t = t0 + r;
    if (lane == 0 && t < T) {
      float val = tot + (s0 ? s0[(long)b * T + t] : 0.f);
      if (mask) val -= 1e9f * (1.0f - (mask[(long)b * T + t] ? 1.0f : 0.0f));
      out[(long)b * T + t] = val;
    }
  }
}

// acc[c] = sum_t wgt[t] X[b,t,c]  for this block's 64 columns, 4 waves striding t, ATT_UNR loads in flight
__device__ __forceinline__ float weighted_colsum(const float* wgt, const float* Xb, int T, int D, int c, int w) {
  float acc = 0.f;
  if (c >= D) return 0.f;
  for (int t0 = w; t0 < T; t0 += 4 * ATT_UNR) {
    float x[ATT_UNR];
#pragma unroll
    for (int i = 0; i < ATT_UNR; ++i) { const int t = t0 + 4 * i; x[i] = t < T ? Xb[(long)t * D + c] : 0.f; }
#pragma unroll
    for (int i = 0; i < ATT_UNR; ++i) { const int t = t0 + 4 * i; if (t < T) acc = fmaf(wgt[t], x[i], acc); }
  }
  return acc;
}

// p = softmax(e[b,:]); ctx[b, c0:c0+64] = sum_t p[t] enc[b,t,c0:c0+64].   grid (ceil(D/64), B)
__global__ __launch_bounds__(256) void attn_softmax_ctx_kernel(const float* e, const float* enc, int T, int D, float* p_out, float* ctx,
                                                               long ldctx) {
  extern __shared__ float sm[];  // [T] probabilities + [4][64] partials + [16] reduction scratch
  float* p = sm;
  float* part = sm + T;
  float* red = part + 256;
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  float mx = -INFINITY;
  for (int t = threadIdx.x; t < T; t += 256) { const float v = e[(long)b * T + t]; p[t] = v; mx = fmaxf(mx, v); }
  mx = block_max(mx, red);
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) { const float v = expf(p[t] - mx); p[t] = v; s += v; }
  s = block_sum(s, red);
  const float inv = 1.f / s;
  __syncthreads();
  if (blockIdx.x == 0 && p_out)
    for (int t = threadIdx.x; t < T; t += 256) p_out[(long)b * T + t] = p[t] * inv;
  part[w * 64 + (threadIdx.x & 63)] = weighted_colsum(p, enc + (long)b * T * D, T, D, c, w);
  __syncthreads();
  if (w == 0 && c < D) ctx[(long)b * ldctx + c] = (part[threadIdx.x] + part[64 + threadIdx.x] + part[128 + threadIdx.x] + part[192 + threadIdx.x]) * inv;
}

// ds = p (dp - <p, dp>);  dh[b, c0:c0+64] (+)= sum_t ds[t] Kq[b,t,c0:c0+64].   grid (ceil(Hd/64), B)
__global__ __launch_bounds__(256) void attn_bwd_dh_kernel(const float* p, const float* dp, const float* Kq, int T, int Hd, float* ds_out,
                                                          float* dh, long lddh, int accumulate) {
  extern __shared__ float sm[];
  float* ds = sm;
  float* part = sm + T;
  float* red = part + 256;
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  float dot = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) dot = fmaf(p[(long)b * T + t], dp[(long)b * T + t], dot);
  dot = block_sum(dot, red);
  for (int t = threadIdx.x; t < T; t += 256) {
    const float v = p[(long)b * T + t] * (dp[(long)b * T + t] - dot);
    ds[t] = v;
    if (blockIdx.x == 0 && ds_out) ds_out[(long)b * T + t] = v;
  }
  __syncthreads();
  part[w * 64 + (threadIdx.x & 63)] = weighted_colsum(ds, Kq + (long)b * T * Hd, T, Hd, c, w);
  __syncthreads();
  if (w == 0 && c < Hd) {
    const float v = part[threadIdx.x] + part[64 + threadIdx.x] + part[128 + threadIdx.x] + part[192 + threadIdx.x];
    float* o = dh + (long)b * lddh + c;
    *o = accumulate ? *o + v : v;
  }
}

static size_t attn_smem(int T) { return sizeof(float) * ((size_t)T + 256 + 16); }
static inline bool vec4_ok(const void* a, long lda, const void* m, int K) {
  return (K % 4 == 0) && (lda % 4 == 0) && (((uintptr_t)a | (uintptr_t)m) & 15) == 0;
}
static void launch_rowdot(const float* v, long ldv, const float* M, const float* s0, const uint8_t* mask, int B, int T, int K, float* out,
                          hipStream_t st) {
  dim3 grid((unsigned)asr_cdiv(T, 4 * ATT_RPW), (unsigned)B);
  if (vec4_ok(v, ldv, M, K)) hipLaunchKernelGGL(attn_rowdot_kernel<true>, grid, dim3(256), 0, st, v, ldv, M, s0, mask, T, K, out);
  else hipLaunchKernelGGL(attn_rowdot_kernel<false>, grid, dim3(256), 0, st, v, ldv, M, s0, mask, T, K, out);
}

// One decoder step of attention, forward.  h [B,Hd] (row stride ldh), Kq [B,T,Hd], s0 [B,T] or NULL,
// mask [B,T] u8, enc [B,T,D].  Outputs: e scratch [B,T], p [B,T] (saved for backward), ctx [B,D] (row stride ldctx).
extern "C" int asr_attn_step_fwd(const float* h, long ldh, const float* Kq, const float* s0, const uint8_t* mask, const float* enc, int B,
                                 int T, int Hd, int D, float* e, float* p, float* ctx, long ldctx, void* stream) {
  ASR_CHECK(h && Kq && mask && enc && e && p && ctx, ASR_ERR_ARG, "asr_attn_step_fwd: null argument");
  ASR_CHECK(B > 0 && T > 0 && Hd > 0 && D > 0, ASR_ERR_SHAPE, "asr_attn_step_fwd: bad shape");
  ASR_CHECK(attn_smem(T) <= 64 * 1024, ASR_ERR_SHAPE, "asr_attn_step_fwd: T=%d too long for the LDS softmax (max ~16000 frames)", T);
  hipStream_t st = (hipStream_t)stream;
  launch_rowdot(h, ldh, Kq, s0, mask, B, T, Hd, e, st);
  hipLaunchKernelGGL(attn_softmax_ctx_kernel, dim3((unsigned)asr_cdiv(D, 64), (unsigned)B), dim3(256), attn_smem(T), st, (const float*)e, enc,
                     T, D, p, ctx, ldctx);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// Backward of one step: dctx [B,D] -> ds [B,T] (gradient wrt the scores, kept for the batched key
// gradients) and dh [B,Hd] (+)= ds . Kq.  dp is scratch [B,T].
extern "C" int asr_attn_step_bwd(const float* dctx, long lddctx, const float* p, const float* Kq, const float* enc, int B, int T, int Hd,
                                 int D, float* dp, float* ds, float* dh, long lddh, int accumulate, void* stream) {
  ASR_CHECK(dctx && p && Kq && enc && dp && ds && dh, ASR_ERR_ARG, "asr_attn_step_bwd: null argument");
  ASR_CHECK(B > 0 && T > 0 && Hd > 0 && D > 0 && attn_smem(T) <= 64 * 1024, ASR_ERR_SHAPE, "asr_attn_step_bwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  launch_rowdot(dctx, lddctx, enc, nullptr, nullptr, B, T, D, dp, st);
  hipLaunchKernelGGL(attn_bwd_dh_kernel, dim3((unsigned)asr_cdiv(Hd, 64), (unsigned)B), dim3(256), attn_smem(T), st, p, (const float*)dp, Kq,
                     T, Hd, ds, dh, lddh, accumulate);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// ------------------------------------------------------------------------------------------ bf16 streams (mixed precision)
// The same two-launch steps reading bf16 images of Kq and enc (asr_f32_to_bf16 once per training step): the streams are
// what these kernels move (LAS-large: 392 MB per decoder step), so the images halve their time; products and sums stay f32.
__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }

__global__ __launch_bounds__(256) void attn_rowdot16_kernel(const float* v, long ldv, const unsigned short* M, const float* s0, const uint8_t* mask,
                                                            int T, int K, float* out) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* vb = v + (long)b * ldv;
  const int t0 = blockIdx.x * (4 * ATT_RPW) + w * ATT_RPW;
  float s[ATT_RPW];
#pragma unroll
  for (int r = 0; r < ATT_RPW; ++r) s[r] = 0.f;
  for (int k = lane * 8; k < K; k += 512) {                      // K % 8 == 0: eight bf16 = 16 bytes per lane
    const float4 h0 = *reinterpret_cast<const float4*>(vb + k), h1 = *reinterpret_cast<const float4*>(vb + k + 4);
    uint4 kv[ATT_RPW];
#pragma unroll
    for (int r = 0; r < ATT_RPW; ++r)
      kv[r] = (t0 + r < T) ? *reinterpret_cast<const uint4*>(M + ((long)b * T + t0 + r) * K + k) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int r = 0; r < ATT_RPW; ++r) {
      s[r] += h0.x * __uint_as_float(kv[r].x << 16) + h0.y * __uint_as_float(kv[r].x & 0xffff0000u) +
              h0.z * __uint_as_float(kv[r].y << 16) + h0.w * __uint_as_float(kv[r].y & 0xffff0000u) +
              h1.x * __uint_as_float(kv[r].z << 16) + h1.y * __uint_as_float(kv[r].z & 0xffff0000u) +
              h1.z * __uint_as_float(kv[r].w << 16) + h1.w * __uint_as_float(kv[r].w & 0xffff0000u);
    }
  }
#pragma unroll
  for (int r = 0; r < ATT_RPW; ++r) {
    const float tot = wave_sum(s[r]);
    const int t = t0 + r;
    if (lane == 0 && t < T) {
      float val = tot + (s0 ? s0[(long)b * T + t] : 0.f);
      if (mask) val -= 1e9f * (1.0f - (mask[(long)b * T + t] ? 1.0f : 0.0f));
      out[(long)b * T + t] = val;
    }
  }
}

__device__ __forceinline__ float weighted_colsum16(const float* wgt, const unsigned short* Xb, int T, int D, int c, int w) {
  float acc = 0.f;
  if (c >= D) return 0.f;
  for (int t0 = w; t0 < T; t0 += 4 * ATT_UNR) {
    unsigned short x[ATT_UNR];
#pragma unroll
    for (int i = 0; i < ATT_UNR; ++i) { const int t = t0 + 4 * i; x[i] = t < T ? Xb[(long)t * D + c] : (unsigned short)0; }
#pragma unroll
    for (int i = 0; i < ATT_UNR; ++i) { const int t = t0 + 4 * i; if (t < T) acc = fmaf(wgt[t], bf2f(x[i]), acc); }
  }
  return acc;
}

__global__ __launch_bounds__(256) void attn_softmax_ctx16_kernel(const float* e, const unsigned short* enc, int T, int D, float* p_out, float* ctx,
                                                                 long ldctx) {
  extern __shared__ float sm[];
  float* p = sm;
  float* part = sm + T;
  float* red = part + 256;
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  float mx = -INFINITY;
  for (int t = threadIdx.x; t < T; t += 256) { const float v = e[(long)b * T + t]; p[t] = v; mx = fmaxf(mx, v); }
  mx = block_max(mx, red);
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) { const float v = expf(p[t] - mx); p[t] = v; s += v; }
  s = block_sum(s, red);
  const float inv = 1.f / s;
  __syncthreads();
  if (blockIdx.x == 0 && p_out)
    for (int t = threadIdx.x; t < T; t += 256) p_out[(long)b * T + t] = p[t] * inv;
  part[w * 64 + (threadIdx.x & 63)] = weighted_colsum16(p, enc + (long)b * T * D, T, D, c, w);
  __syncthreads();
  if (w == 0 && c < D) ctx[(long)b * ldctx + c] = (part[threadIdx.x] + part[64 + threadIdx.x] + part[128 + threadIdx.x] + part[192 + threadIdx.x]) * inv;
}

__global__ __launch_bounds__(256) void attn_bwd_dh16_kernel(const float* p, const float* dp, const unsigned short* Kq, int T, int Hd, float* ds_out,
                                                            float* dh, long lddh, int accumulate) {
  extern __shared__ float sm[];
  float* ds = sm;
  float* part = sm + T;
  float* red = part + 256;
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  float dot = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) dot = fmaf(p[(long)b * T + t], dp[(long)b * T + t], dot);
  dot = block_sum(dot, red);
  for (int t = threadIdx.x; t < T; t += 256) {
    const float v = p[(long)b * T + t] * (dp[(long)b * T + t] - dot);
    ds[t] = v;
    if (blockIdx.x == 0 && ds_out) ds_out[(long)b * T + t] = v;
  }
  __syncthreads();
  part[w * 64 + (threadIdx.x & 63)] = weighted_colsum16(ds, Kq + (long)b * T * Hd, T, Hd, c, w);
  __syncthreads();
  if (w == 0 && c < Hd) {
    const float v = part[threadIdx.x] + part[64 + threadIdx.x] + part[128 + threadIdx.x] + part[192 + threadIdx.x];
    float* o = dh + (long)b * lddh + c;
    *o = accumulate ? *o + v : v;
  }
}

// Wide rows (the las_large decoder: D = 2048, Hd = 1024): one column per lane makes 2-byte loads - 128 bytes per wave instruction, 2.6 TB/s
// on the 130 MB stream of enc.  Here a lane owns VW consecutive columns (16- or 8-byte loads: 1 KB / 512 B per wave instruction), a
// workgroup 64 VW columns of one batch row, its four waves split the rows of the clip.
template <int VW>
__device__ __forceinline__ void weighted_colsum16v(const float* wgt, const unsigned short* Xb, int T, int D, int c0, int w, float (&acc)[VW]) {
  typedef unsigned int uvec __attribute__((ext_vector_type(VW / 2)));
#pragma unroll
  for (int j = 0; j < VW; ++j) acc[j] = 0.f;
  for (int t0 = w; t0 < T; t0 += 4 * ATT_UNR) {
    uvec x[ATT_UNR];
#pragma unroll
    for (int i = 0; i < ATT_UNR; ++i) {
      const int t = t0 + 4 * i;
      if (t < T) x[i] = *reinterpret_cast<const uvec*>(Xb + (long)t * D + c0);
      else x[i] = uvec(0u);
    }
#pragma unroll
    for (int i = 0; i < ATT_UNR; ++i) {
      const int t = t0 + 4 * i;
      const float wt = t < T ? wgt[t] : 0.f;
#pragma unroll
      for (int j = 0; j < VW / 2; ++j) {
        acc[2 * j] = fmaf(wt, __uint_as_float(x[i][j] << 16), acc[2 * j]);
        acc[2 * j + 1] = fmaf(wt, __uint_as_float(x[i][j] & 0xffff0000u), acc[2 * j + 1]);
      }
    }
  }
}
template <int VW>
static size_t attn_smem_v(int T) { return sizeof(float) * ((size_t)T + 4 * 64 * VW + 16); }

template <int VW>
__global__ __launch_bounds__(256) void attn_softmax_ctx16v_kernel(const float* e, const unsigned short* enc, int T, int D, float* p_out, float* ctx,
                                                                  long ldctx) {
  extern __shared__ float sm[];
  float* p = sm;
  float* part = sm + T;
  float* red = part + 4 * 64 * VW;
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, c0 = (blockIdx.x * 64 + lane) * VW;
  float mx = -INFINITY;
  for (int t = threadIdx.x; t < T; t += 256) { const float v = e[(long)b * T + t]; p[t] = v; mx = fmaxf(mx, v); }
  mx = block_max(mx, red);
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) { const float v = expf(p[t] - mx); p[t] = v; s += v; }
  s = block_sum(s, red);
  const float inv = 1.f / s;
  __syncthreads();
  if (blockIdx.x == 0 && p_out)
    for (int t = threadIdx.x; t < T; t += 256) p_out[(long)b * T + t] = p[t] * inv;
  float acc[VW];
  weighted_colsum16v<VW>(p, enc + (long)b * T * D, T, D, c0, w, acc);
#pragma unroll
  for (int j = 0; j < VW; ++j) part[(w * 64 + lane) * VW + j] = acc[j];
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const int o = lane * VW + j;
      ctx[(long)b * ldctx + c0 + j] = (part[o] + part[64 * VW + o] + part[128 * VW + o] + part[192 * VW + o]) * inv;
    }
  }
}

template <int VW>
__global__ __launch_bounds__(256) void attn_bwd_dh16v_kernel(const float* p, const float* dp, const unsigned short* Kq, int T, int Hd, float* ds_out,
                                                             float* dh, long lddh, int accumulate) {
  extern __shared__ float sm[];
  float* ds = sm;
  float* part = sm + T;
  float* red = part + 4 * 64 * VW;
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, c0 = (blockIdx.x * 64 + lane) * VW;
  float dot = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) dot = fmaf(p[(long)b * T + t], dp[(long)b * T + t], dot);
  dot = block_sum(dot, red);
  for (int t = threadIdx.x; t < T; t += 256) {
    const float v = p[(long)b * T + t] * (dp[(long)b * T + t] - dot);
    ds[t] = v;
    if (blockIdx.x == 0 && ds_out) ds_out[(long)b * T + t] = v;
  }
  __syncthreads();
  float acc[VW];
  weighted_colsum16v<VW>(ds, Kq + (long)b * T * Hd, T, Hd, c0, w, acc);
#pragma unroll
  for (int j = 0; j < VW; ++j) part[(w * 64 + lane) * VW + j] = acc[j];
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const int o = lane * VW + j;
      const float v = part[o] + part[64 * VW + o] + part[128 * VW + o] + part[192 * VW + o];
      float* op = dh + (long)b * lddh + c0 + j;
      *op = accumulate ? *op + v : v;
    }
  }
}
// columns per lane for a row of `cols` bf16: the widest loads that still make ~a workgroup per compute unit (0 = the one-column kernels)
static int attn_vw(int cols, int B) {
  static const int on = getenv("ASR_ATTN_VEC") ? atoi(getenv("ASR_ATTN_VEC")) : 1;
  if (!on) return 0;
  if (cols % 512 == 0 && (long)(cols / 512) * B >= 192) return 8;
  if (cols % 256 == 0 && (long)(cols / 256) * B >= 192) return 4;
  return 0;
}

extern "C" int asr_attn_step_fwd_bf16(const float* h, long ldh, const void* Kq16, const float* s0, const uint8_t* mask, const void* enc16, int B,
                                      int T, int Hd, int D, float* e, float* p, float* ctx, long ldctx, void* stream) {
  ASR_CHECK(h && Kq16 && mask && enc16 && e && p && ctx, ASR_ERR_ARG, "asr_attn_step_fwd_bf16: null argument");
  ASR_CHECK(B > 0 && T > 0 && Hd > 0 && D > 0 && Hd % 8 == 0 && ldh % 4 == 0 && attn_smem(T) <= 64 * 1024, ASR_ERR_SHAPE,
            "asr_attn_step_fwd_bf16: bad shape (Hd %% 8, ldh %% 4 required)");
  ASR_CHECK(((((uintptr_t)h) | ((uintptr_t)Kq16)) & 15) == 0, ASR_ERR_ARG, "asr_attn_step_fwd_bf16: h and Kq16 must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_rowdot16_kernel, dim3((unsigned)asr_cdiv(T, 4 * ATT_RPW), (unsigned)B), dim3(256), 0, st, h, ldh,
                     static_cast<const unsigned short*>(Kq16), s0, mask, T, Hd, e);
  const int vw = ((uintptr_t)enc16 & 15) == 0 ? attn_vw(D, B) : 0;
  if (vw == 8)
    hipLaunchKernelGGL(attn_softmax_ctx16v_kernel<8>, dim3((unsigned)(D / 512), (unsigned)B), dim3(256), attn_smem_v<8>(T), st, (const float*)e,
                       static_cast<const unsigned short*>(enc16), T, D, p, ctx, ldctx);
  else if (vw == 4)
    hipLaunchKernelGGL(attn_softmax_ctx16v_kernel<4>, dim3((unsigned)(D / 256), (unsigned)B), dim3(256), attn_smem_v<4>(T), st, (const float*)e,
                       static_cast<const unsigned short*>(enc16), T, D, p, ctx, ldctx);
  else
    hipLaunchKernelGGL(attn_softmax_ctx16_kernel, dim3((unsigned)asr_cdiv(D, 64), (unsigned)B), dim3(256), attn_smem(T), st, (const float*)e,
                       static_cast<const unsigned short*>(enc16), T, D, p, ctx, ldctx);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_attn_step_bwd_bf16(const float* dctx, long lddctx, const float* p, const void* Kq16, const void* enc16, int B, int T, int Hd,
                                      int D, float* dp, float* ds, float* dh, long lddh, int accumulate, void* stream) {
  ASR_CHECK(dctx && p && Kq16 && enc16 && dp && ds && dh, ASR_ERR_ARG, "asr_attn_step_bwd_bf16: null argument");
  ASR_CHECK(B > 0 && T > 0 && Hd > 0 && D > 0 && D % 8 == 0 && lddctx % 4 == 0 && attn_smem(T) <= 64 * 1024, ASR_ERR_SHAPE,
            "asr_attn_step_bwd_bf16: bad shape (D %% 8, lddctx %% 4 required)");
  ASR_CHECK(((((uintptr_t)dctx) | ((uintptr_t)enc16)) & 15) == 0, ASR_ERR_ARG, "asr_attn_step_bwd_bf16: dctx and enc16 must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_rowdot16_kernel, dim3((unsigned)asr_cdiv(T, 4 * ATT_RPW), (unsigned)B), dim3(256), 0, st, dctx, lddctx,
                     static_cast<const unsigned short*>(enc16), (const float*)nullptr, (const uint8_t*)nullptr, T, D, dp);
  const int vw = ((uintptr_t)Kq16 & 15) == 0 ? attn_vw(Hd, B) : 0;
  if (vw == 8)
    hipLaunchKernelGGL(attn_bwd_dh16v_kernel<8>, dim3((unsigned)(Hd / 512), (unsigned)B), dim3(256), attn_smem_v<8>(T), st, p, (const float*)dp,
                       static_cast<const unsigned short*>(Kq16), T, Hd, ds, dh, lddh, accumulate);
  else if (vw == 4)
    hipLaunchKernelGGL(attn_bwd_dh16v_kernel<4>, dim3((unsigned)(Hd / 256), (unsigned)B), dim3(256), attn_smem_v<4>(T), st, p, (const float*)dp,
                       static_cast<const unsigned short*>(Kq16), T, Hd, ds, dh, lddh, accumulate);
  else
    hipLaunchKernelGGL(attn_bwd_dh16_kernel, dim3((unsigned)asr_cdiv(Hd, 64), (unsigned)B), dim3(256), attn_smem(T), st, p, (const float*)dp,
                       static_cast<const unsigned short*>(Kq16), T, Hd, ds, dh, lddh, accumulate);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
