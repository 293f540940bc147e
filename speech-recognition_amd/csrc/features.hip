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

// ------------------------------------------------------------------------------------------ time warp
// data.py:275-280: tfa.image.sparse_image_warp(audio [T, v, C], src (t_s, v/2), dst (t_d, v/2), num_boundary_points=3)
// with the defaults interpolation_order = 2, regularization_weight = 0  ([TF-sem], tensorflow-addons):
//   control points = the moved point + 12 zero-flow points on the image boundary (the border of a 4 x 4 grid of
//   linspace(0, T-1, 4) x linspace(0, v-1, 4));  a polyharmonic spline of order 2 (phi(r2) = 0.5 r2 log max(r2, 1e-10)
//   on squared distances, plus an affine term) interpolates the flows at the DESTINATION points; every pixel is then
//   read at (t - flow_t, f - flow_f) with bilinear interpolation (floors clamped to [0, size-2], weights to [0, 1]).
// Only the time component of the flow is non-zero (all control flows are (dt, 0), so the frequency system has a
// zero right-hand side).  The 16 x 16 spline system is solved per clip by one thread in float64 with partial
// pivoting (TensorFlow solves it in float32); the evaluation follows TensorFlow's float32 formulas.
#define TW_N 13          // control points: 1 moved + 12 boundary
#define TW_DIM (TW_N + 3)

struct TwArgs {
  const float* x;
  const int32_t* n_frames;
  const uint32_t* seed;
  float* coef;           // [B][32]: w[13], v[3], control y[13] (x positions are re-derived), flag
  float* out;
  int B, T, v, C, W;
};

__device__ __forceinline__ float tw_phi(float r) { return 0.5f * r * logf(fmaxf(r, 1e-10f)); }

__device__ __forceinline__ void tw_points(int T_b, int v, float dst, float* cy, float* cx) {
  cy[0] = dst; cx[0] = (float)(v / 2);
  int n = 1;
  for (int iy = 0; iy < 4; ++iy)
    for (int ix = 0; ix < 4; ++ix) {
      if (iy != 0 && iy != 3 && ix != 0 && ix != 3) continue;
      // np.linspace(0, size - 1, 4) in float64, then cast to float32
      cy[n] = (float)((double)(T_b - 1) * iy / 3.0);
      cx[n] = (float)((double)(v - 1) * ix / 3.0);
      ++n;
    }
}

__global__ __launch_bounds__(64) void time_warp_solve_kernel(TwArgs a) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  float* co = a.coef + (long)b * 32;
  const int T_b = a.n_frames ? min(a.n_frames[b], a.T) : a.T;
  co[31] = 0.f;                                                  // flag: 0 = copy the clip unchanged
  if (T_b <= 2 * a.W || T_b < 2 || a.v < 2) return;              // the reference's draw range is empty there
  const AsrRngKey key = asr_rng_key(a.seed[0], 5u /* STREAM_TIMEWARP */);
  const int src = a.W + asr_uniform_int(key, (uint32_t)(2 * b), T_b - 2 * a.W);          // uniform((), W, T - W)
  const int dst = src - a.W + asr_uniform_int(key, (uint32_t)(2 * b + 1), 2 * a.W);      // + uniform((), -W, W)
  float cy[TW_N], cx[TW_N];
  tw_points(T_b, a.v, (float)dst, cy, cx);
  double M[TW_DIM][TW_DIM + 1];
  for (int i = 0; i < TW_DIM; ++i)
    for (int j = 0; j <= TW_DIM; ++j) M[i][j] = 0.0;
  for (int i = 0; i < TW_N; ++i) {
    for (int j = 0; j < TW_N; ++j) {
      const double dy = (double)cy[i] - cy[j], dx = (double)cx[i] - cx[j];
      const double r = dy * dy + dx * dx;
      M[i][j] = 0.5 * r * log(fmax(r, 1e-10));
    }
    M[i][TW_N] = cy[i]; M[i][TW_N + 1] = cx[i]; M[i][TW_N + 2] = 1.0;
    M[TW_N][i] = cy[i]; M[TW_N + 1][i] = cx[i]; M[TW_N + 2][i] = 1.0;
  }
  M[0][TW_DIM] = (double)(dst - src);                            // flow of the moved point; the boundary points stay
  for (int c = 0; c < TW_DIM; ++c) {
    int piv = c;
    for (int r = c + 1; r < TW_DIM; ++r)
      if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
    if (fabs(M[piv][c]) < 1e-12) return;                         // coincident control points: leave the clip as it is
    if (piv != c)
      for (int j = 0; j <= TW_DIM; ++j) { const double t = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = t; }
    for (int r = c + 1; r < TW_DIM; ++r) {
      const double f = M[r][c] / M[c][c];
      for (int j = c; j <= TW_DIM; ++j) M[r][j] -= f * M[c][j];
    }
  }
  double sol[TW_DIM];
  for (int r = TW_DIM - 1; r >= 0; --r) {
    double s = M[r][TW_DIM];
    for (int j = r + 1; j < TW_DIM; ++j) s -= M[r][j] * sol[j];
    sol[r] = s / M[r][r];
  }
  for (int i = 0; i < TW_DIM; ++i) co[i] = (float)sol[i];
  co[16] = (float)dst;
  co[31] = 1.f;
}

__global__ __launch_bounds__(256) void time_warp_apply_kernel(TwArgs a) {
  __shared__ float w[TW_DIM], cy[TW_N], cx[TW_N];
  __shared__ int on;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int T_b = a.n_frames ? min(a.n_frames[b], a.T) : a.T;
  const float* co = a.coef + (long)b * 32;
  if (tid == 0) {
    on = co[31] != 0.f;
    for (int i = 0; i < TW_DIM; ++i) w[i] = co[i];
    if (on) tw_points(T_b, a.v, co[16], cy, cx);
  }
  __syncthreads();
  const int row = a.v * a.C;
  const float* img = a.x + (long)b * a.T * row;
  float* dst = a.out + (long)b * a.T * row;
  const int t_lo = blockIdx.x * 8, t_hi = min(t_lo + 8, a.T);
  for (int t = t_lo; t < t_hi; ++t) {
    for (int e = tid; e < row; e += 256) {
      float val;
      if (!on || t >= T_b) {
        val = img[(long)t * row + e];
      } else {
        const int f = e / a.C, c = e - f * a.C;
        float flow = w[TW_N] * (float)t + w[TW_N + 1] * (float)f + w[TW_N + 2];
        float rbf = 0.f;
        for (int i = 0; i < TW_N; ++i) {
          const float dy = (float)t - cy[i], dx = (float)f - cx[i];
          rbf = fmaf(tw_phi(dy * dy + dx * dx), w[i], rbf);
        }
        flow += rbf;
        const float qy = (float)t - flow;
        const float fy = fminf(fmaxf(0.f, floorf(qy)), (float)(T_b - 2));
        const float ay = fminf(fmaxf(0.f, qy - fy), 1.f);
        const int iy = (int)fy, ix = min(f, a.v - 2);
        const float ax = (float)(f - ix);
        const float tl = img[((long)iy * a.v + ix) * a.C + c], tr = img[((long)iy * a.v + ix + 1) * a.C + c];
        const float bl = img[((long)(iy + 1) * a.v + ix) * a.C + c], br = img[((long)(iy + 1) * a.v + ix + 1) * a.C + c];
        const float top = ax * (tr - tl) + tl, bot = ax * (br - bl) + bl;
        val = ay * (bot - top) + top;
      }
      dst[(long)t * row + e] = val;
    }
  }
}

extern "C" int asr_time_warp(const float* x, const int32_t* n_frames, int B, int T, int v, int C, int W, const uint32_t* seed, float* coef,
                             float* out, void* stream) {
  ASR_CHECK(x && seed && coef && out, ASR_ERR_ARG, "asr_time_warp: null argument");
  ASR_CHECK(x != out, ASR_ERR_ARG, "asr_time_warp: x and out must not alias (the warp gathers)");
  ASR_CHECK(B > 0 && T > 0 && v > 0 && C > 0 && W > 0, ASR_ERR_SHAPE, "asr_time_warp: bad shape B=%d T=%d v=%d C=%d W=%d", B, T, v, C, W);
  TwArgs a{x, n_frames, seed, coef, out, B, T, v, C, W};
  hipLaunchKernelGGL(time_warp_solve_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(time_warp_apply_kernel, dim3((unsigned)asr_cdiv(T, 8), (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
