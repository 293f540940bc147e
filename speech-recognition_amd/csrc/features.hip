// Stand-alone SpecAugment and delta/delta-delta kernels on stored feature tensors (the --use-tfrecord
// route of run/train.py:70-74, where the dataset already holds log-mel frames).  Both are one pass
// over the tensor: HBM-bound, 16-byte accesses where the row length allows, one workgroup per
// (clip, 8-frame slab) so every XCD streams its own contiguous slabs.
#include "common.h"

#define SA_MAXBAND 16   // = FE_MAXBAND of the fused front end

struct SaArgs {
  float* x;
  const int32_t* n_frames;
  const uint32_t* seed;
  int B, T, v, C;
  int F, mF, Tm, mT;
  float p;
};

// data.py:282-301.  The draws (stream 3, index b*64 + ...) are those of logmel_kernel (frontend.hip).
__global__ __launch_bounds__(256) void spec_augment_kernel(SaArgs a) {
  __shared__ int bands[4][SA_MAXBAND];   // f0, f, t0, t
  const int b = blockIdx.y, tid = threadIdx.x;
  const int T_b = a.n_frames ? min(a.n_frames[b], a.T) : a.T;
  if (tid == 0) {
    for (int i = 0; i < SA_MAXBAND; ++i) { bands[0][i] = 0; bands[1][i] = 0; bands[2][i] = 0; bands[3][i] = 0; }
    const AsrRngKey key = asr_rng_key(a.seed[0], 3u /* STREAM_SPECAUG */);
    if (a.F > 0 && a.mF > 0) {
      for (int i = 0; i < a.mF; ++i) {
        const int f = asr_uniform_int(key, (uint32_t)(b * 64 + 2 * i), a.F);
        const int f0 = asr_uniform_int(key, (uint32_t)(b * 64 + 2 * i + 1), a.v - f);
        bands[0][i] = f0; bands[1][i] = f;
      }
    }
    if (a.Tm > 0 && a.mT > 0 && a.p > 0.f) {
      int applied = 0;
      const int max_maskable = (int)((float)T_b * a.p);
      for (int j = 0; j < a.mT; ++j) {
        int t = asr_uniform_int(key, (uint32_t)(b * 64 + 32 + 2 * j), a.Tm);
        t = max(min(t, max_maskable - applied), 0);
        applied += t;
        const int tt0 = asr_uniform_int(key, (uint32_t)(b * 64 + 32 + 2 * j + 1), T_b - t);
        bands[2][j] = tt0; bands[3][j] = t;
      }
    }
  }
  __syncthreads();
  const int row = a.v * a.C;
  const int t_lo = blockIdx.x * 8, t_hi = min(t_lo + 8, T_b);
  // only elements inside a band are touched (a store of 0.0); everything else stays as it is
  for (int t = t_lo; t < t_hi; ++t) {
    bool tz = false;
    for (int j = 0; j < a.mT; ++j) tz |= (t >= bands[2][j] && t < bands[2][j] + bands[3][j]);
    float* xr = a.x + ((long)b * a.T + t) * row;
    for (int e = tid; e < row; e += 256) {
      const int m = e / a.C;
      bool z = tz;
      for (int i = 0; i < a.mF; ++i) z |= (m >= bands[0][i] && m < bands[0][i] + bands[1][i]);
      if (z) xr[e] = 0.f;
    }
  }
}

extern "C" int asr_spec_augment(const asr_logmel_cfg* cfg, float* x, const int32_t* n_frames, int B, int T, int C,
                                const uint32_t* seed, void* stream) {
  ASR_CHECK(cfg && x && seed, ASR_ERR_ARG, "asr_spec_augment: null argument");
  ASR_CHECK(B > 0 && T > 0 && C > 0 && cfg->num_mel_bins > 0, ASR_ERR_SHAPE, "asr_spec_augment: bad shape B=%d T=%d C=%d v=%d", B, T, C, cfg->num_mel_bins);
  ASR_CHECK(cfg->sa_mF <= SA_MAXBAND && cfg->sa_mT <= SA_MAXBAND, ASR_ERR_SHAPE, "asr_spec_augment: m_F/m_T > %d", SA_MAXBAND);
  if (!cfg->sa_enable) return ASR_OK;
  SaArgs a{};
  a.x = x; a.n_frames = n_frames; a.seed = seed;
  a.B = B; a.T = T; a.v = cfg->num_mel_bins; a.C = C;
  const bool use_f = cfg->sa_F > 0 && cfg->sa_mF > 0, use_t = cfg->sa_T > 0 && cfg->sa_mT > 0 && cfg->sa_p > 0.f;
  a.F = use_f ? cfg->sa_F : 0; a.mF = use_f ? cfg->sa_mF : 0;
  a.Tm = use_t ? cfg->sa_T : 0; a.mT = use_t ? cfg->sa_mT : 0; a.p = cfg->sa_p;
  hipLaunchKernelGGL(spec_augment_kernel, dim3((unsigned)asr_cdiv(T, 8), (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// data.py:319-324: delta[t] = x[t] - x[t-1], deltas[t] = delta[t] - delta[t-1], x[-1] = delta[-1] = 0,
// same subtraction order as the reference.  One thread per (frame, bin): three 4-byte reads that hit
// the same lines as the neighbouring frames' (L1/L2), one 12-byte write; rows are contiguous.
__global__ __launch_bounds__(256) void delta_kernel(const float* __restrict__ x, const int32_t* __restrict__ n_frames, int T, int v,
                                                    float* __restrict__ out) {
  const int b = blockIdx.y;
  const int T_b = n_frames ? min(n_frames[b], T) : T;
  const long per_clip = (long)T * v;
  const float* xb = x + (long)b * per_clip;
  float* ob = out + (long)b * per_clip * 3;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < per_clip; e += (long)gridDim.x * 256) {
    const int t = (int)(e / v);
    float x0 = 0.f, d0 = 0.f, dd = 0.f;
    if (t < T_b) {
      x0 = xb[e];
      const float x1 = t >= 1 ? xb[e - v] : 0.f, x2 = t >= 2 ? xb[e - 2L * v] : 0.f;
      d0 = x0 - x1;
      dd = d0 - (x1 - x2);
    }
    ob[3 * e] = x0; ob[3 * e + 1] = d0; ob[3 * e + 2] = dd;
  }
}

extern "C" int asr_delta_accelerate(const float* x, const int32_t* n_frames, int B, int T, int v, float* out, void* stream) {
  ASR_CHECK(x && out, ASR_ERR_ARG, "asr_delta_accelerate: null argument");
  ASR_CHECK(B > 0 && T > 0 && v > 0, ASR_ERR_SHAPE, "asr_delta_accelerate: bad shape B=%d T=%d v=%d", B, T, v);
  const long per_clip = (long)T * v;
  const unsigned gx = (unsigned)std::min<long>(asr_cdiv(per_clip, 256), 2048);
  hipLaunchKernelGGL(delta_kernel, dim3(gx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, n_frames, T, v, out);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
