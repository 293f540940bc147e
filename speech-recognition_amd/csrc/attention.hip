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
#include "common.h"

// e[b,t] = h[b] . Kq[b,t,:] + s0[b,t] - 1e9 (1 - mask[b,t]).   grid (ceil(T/32), B); a wave does 8 rows.
__global__ __launch_bounds__(256) void attn_scores_kernel(const float* h, long ldh, const float* Kq, const float* s0, const uint8_t* mask,
                                                          int T, int Hd, float* e) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* hb = h + (long)b * ldh;
  for (int i = 0; i < 8; ++i) {
    const int t = blockIdx.x * 32 + w * 8 + i;
    if (t >= T) break;
    const float* kr = Kq + ((long)b * T + t) * Hd;
    float s = 0.f;
    for (int k = lane; k < Hd; k += 64) s = fmaf(hb[k], kr[k], s);
    s = wave_sum(s);
    if (lane == 0) {
      float v = s + (s0 ? s0[(long)b * T + t] : 0.f);
      v -= 1e9f * (1.0f - (mask[(long)b * T + t] ? 1.0f : 0.0f));
      e[(long)b * T + t] = v;
    }
  }
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
  float acc = 0.f;
  if (c < D)
    for (int t = w; t < T; t += 4) acc = fmaf(p[t], enc[((long)b * T + t) * D + c], acc);
  part[w * 64 + (threadIdx.x & 63)] = acc;
  __syncthreads();
  if (w == 0 && c < D) ctx[(long)b * ldctx + c] = (part[threadIdx.x] + part[64 + threadIdx.x] + part[128 + threadIdx.x] + part[192 + threadIdx.x]) * inv;
}

// dp[b,t] = dctx[b] . enc[b,t,:]     grid (ceil(T/32), B)
__global__ __launch_bounds__(256) void attn_bwd_dp_kernel(const float* dctx, long ldd, const float* enc, int T, int D, float* dp) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* db = dctx + (long)b * ldd;
  for (int i = 0; i < 8; ++i) {
    const int t = blockIdx.x * 32 + w * 8 + i;
    if (t >= T) break;
    const float* er = enc + ((long)b * T + t) * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s = fmaf(db[k], er[k], s);
    s = wave_sum(s);
    if (lane == 0) dp[(long)b * T + t] = s;
  }
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
  float acc = 0.f;
  if (c < Hd)
    for (int t = w; t < T; t += 4) acc = fmaf(ds[t], Kq[((long)b * T + t) * Hd + c], acc);
  part[w * 64 + (threadIdx.x & 63)] = acc;
  __syncthreads();
  if (w == 0 && c < Hd) {
    const float v = part[threadIdx.x] + part[64 + threadIdx.x] + part[128 + threadIdx.x] + part[192 + threadIdx.x];
    float* o = dh + (long)b * lddh + c;
    *o = accumulate ? *o + v : v;
  }
}

static size_t attn_smem(int T) { return sizeof(float) * ((size_t)T + 256 + 16); }

// One decoder step of attention, forward.  h [B,Hd] (row stride ldh), Kq [B,T,Hd], s0 [B,T] or NULL,
// mask [B,T] u8, enc [B,T,D].  Outputs: e scratch [B,T], p [B,T] (saved for backward), ctx [B,D] (row stride ldctx).
extern "C" int asr_attn_step_fwd(const float* h, long ldh, const float* Kq, const float* s0, const uint8_t* mask, const float* enc, int B,
                                 int T, int Hd, int D, float* e, float* p, float* ctx, long ldctx, void* stream) {
  ASR_CHECK(h && Kq && mask && enc && e && p && ctx, ASR_ERR_ARG, "asr_attn_step_fwd: null argument");
  ASR_CHECK(B > 0 && T > 0 && Hd > 0 && D > 0, ASR_ERR_SHAPE, "asr_attn_step_fwd: bad shape");
  ASR_CHECK(attn_smem(T) <= 64 * 1024, ASR_ERR_SHAPE, "asr_attn_step_fwd: T=%d too long for the LDS softmax (max ~16000 frames)", T);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_scores_kernel, dim3((unsigned)asr_cdiv(T, 32), (unsigned)B), dim3(256), 0, st, h, ldh, Kq, s0, mask, T, Hd, e);
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
  hipLaunchKernelGGL(attn_bwd_dp_kernel, dim3((unsigned)asr_cdiv(T, 32), (unsigned)B), dim3(256), 0, st, dctx, lddctx, enc, T, D, dp);
  hipLaunchKernelGGL(attn_bwd_dh_kernel, dim3((unsigned)asr_cdiv(Hd, 64), (unsigned)B), dim3(256), attn_smem(T), st, p, (const float*)dp, Kq,
                     T, Hd, ds, dh, lddh, accumulate);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
