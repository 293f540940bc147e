// Conv2D (VALID, NHWC, HWIO kernel, linear) forward / filter-gradient / input-gradient as implicit
// GEMMs on the f32 MFMA core (gemm_core.h).  Replaces tf.keras.layers.Conv2D at las.py:163-164,
// 183-184 and deepspeech2.py:47-50, 57-59 and the gradients TF derives for them.
//
//   forward      Y[pos][o]   = sum_kc im2col[pos][kc] W[kc][o] + b[o]            (NN, A gathered)
//   bwd filter   dW[kc][o]  += sum_pos im2col[pos][kc] dY[pos][o]                (TN, A gathered, split-K)
//   bwd data     dX[ipos][c] = sum_{r,s,o} dY[b,(h-r)/sh,(w-s)/sw,o] W[r,s,c,o]   (NT, both gathered)
// pos = (b, ho, wo); kc = (r, s, c) with c fastest - exactly the HWIO kernel's row index.
#include "gemm_core.h"

struct ConvGeom {
  int B, H, W, C, kh, kw, sh, sw, O, Ho, Wo;
};

// stored logical matrix im2col [R = B*Ho*Wo][Cc = kh*kw*C]
struct Im2colLoader {
  const float* x;
  ConvGeom g;
  int R, Cc, vec_ok;
  __device__ __forceinline__ void fetch4(int r, int c, float (&v)[4]) const {
    if (r >= R || c >= Cc) { v[0] = v[1] = v[2] = v[3] = 0.f; return; }
    const int wo = r % g.Wo, t = r / g.Wo, ho = t % g.Ho, b = t / g.Ho;
    const long base = (((long)b * g.H + (long)ho * g.sh) * g.W + (long)wo * g.sw) * g.C;
    const int rowlen = g.kw * g.C;  // one kernel row (s, c) is contiguous in the input
    if (vec_ok && c + 3 < Cc) {
      const int kr = c / rowlen, rem = c - kr * rowlen;       // rem..rem+3 stay inside the row (C % 4 == 0)
      const float4 t4 = *reinterpret_cast<const float4*>(x + base + (long)kr * g.W * g.C + rem);
      v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cc = c + i;
        if (cc < Cc) {
          const int kr = cc / rowlen, rem = cc - kr * rowlen;
          v[i] = x[base + (long)kr * g.W * g.C + rem];
        } else v[i] = 0.f;
      }
    }
  }
};

// bwd-data A operand: stored logical matrix G [R = B*H*W][Cc = kh*kw*O],
// G[(b,h,w)][(r,s,o)] = dY[b, (h-r)/sh, (w-s)/sw, o] when the division is exact and in range, else 0
struct DyGatherLoader {
  const float* dy;
  ConvGeom g;
  int R, Cc, vec_ok;
  __device__ __forceinline__ void fetch4(int r, int c, float (&v)[4]) const {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (r >= R || c >= Cc) return;
    const int w = r % g.W, t = r / g.W, h = t % g.H, b = t / g.H;
    if (vec_ok && c + 3 < Cc) {   // O % 4 == 0 and c % 4 == 0: the four o share (r, s) and are contiguous
      const int o = c % g.O, rs = c / g.O, s = rs % g.kw, kr = rs / g.kw;
      const int hh = h - kr, ww = w - s;
      if (hh < 0 || ww < 0 || hh % g.sh != 0 || ww % g.sw != 0) return;
      const int ho = hh / g.sh, wo = ww / g.sw;
      if (ho >= g.Ho || wo >= g.Wo) return;
      const float4 t4 = *reinterpret_cast<const float4*>(dy + (((long)b * g.Ho + ho) * g.Wo + wo) * g.O + o);
      v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cc = c + i;
      if (cc >= Cc) continue;
      const int o = cc % g.O, rs = cc / g.O, s = rs % g.kw, kr = rs / g.kw;
      const int hh = h - kr, ww = w - s;
      if (hh < 0 || ww < 0 || hh % g.sh != 0 || ww % g.sw != 0) continue;
      const int ho = hh / g.sh, wo = ww / g.sw;
      if (ho >= g.Ho || wo >= g.Wo) continue;
      v[i] = dy[(((long)b * g.Ho + ho) * g.Wo + wo) * g.O + o];
    }
  }
};

// bwd-data B operand: stored logical matrix Wt [R = C][Cc = kh*kw*O], Wt[c][(r,s,o)] = W[r,s,c,o]
struct WtLoader {
  const float* w;
  ConvGeom g;
  int R, Cc, vec_ok;
  __device__ __forceinline__ void fetch4(int r, int c, float (&v)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cc = c + i;
      if (r < R && cc < Cc) {
        const int o = cc % g.O, rs = cc / g.O;
        v[i] = w[((long)rs * g.C + r) * g.O + o];
      } else v[i] = 0.f;
    }
  }
};

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_fwd_kernel(Im2colLoader al, PlainLoader bl, GemmEpilogue ep, int K, int tiles_m) {
  using T = GemmTile<0, 0, BM, BN, WAVES_M, WAVES_N>;
  __shared__ __attribute__((aligned(16))) float As[T::A_ELEMS];
  __shared__ __attribute__((aligned(16))) float Bs[T::B_ELEMS];
  const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;
  T::run(al, bl, ep, 0, K, bm * BM, bn * BN, As, Bs);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_bwd_filter_kernel(Im2colLoader al, PlainLoader bl, GemmEpilogue ep, int K,
                                                              int tiles_m, int k_chunk) {
  using T = GemmTile<1, 0, BM, BN, WAVES_M, WAVES_N>;
  __shared__ __attribute__((aligned(16))) float As[T::A_ELEMS];
  __shared__ __attribute__((aligned(16))) float Bs[T::B_ELEMS];
  const int kbeg = blockIdx.z * k_chunk, kend = min(K, kbeg + k_chunk);
  if (kbeg >= K) return;
  const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;
  T::run(al, bl, ep, kbeg, kend, bm * BM, bn * BN, As, Bs);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_bwd_data_kernel(DyGatherLoader al, WtLoader bl, GemmEpilogue ep, int K, int tiles_m) {
  using T = GemmTile<0, 1, BM, BN, WAVES_M, WAVES_N>;
  __shared__ __attribute__((aligned(16))) float As[T::A_ELEMS];
  __shared__ __attribute__((aligned(16))) float Bs[T::B_ELEMS];
  const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;
  T::run(al, bl, ep, 0, K, bm * BM, bn * BN, As, Bs);
}

static int conv_geom(const asr_conv_desc* d, ConvGeom* g) {
  ASR_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->O > 0 && d->kh > 0 && d->kw > 0 && d->sh > 0 && d->sw > 0,
            ASR_ERR_SHAPE, "conv2d: non-positive dimension");
  ASR_CHECK(d->H >= d->kh && d->W >= d->kw, ASR_ERR_SHAPE, "conv2d: input %dx%d smaller than kernel %dx%d", d->H, d->W, d->kh, d->kw);
  g->B = d->B; g->H = d->H; g->W = d->W; g->C = d->C; g->kh = d->kh; g->kw = d->kw; g->sh = d->sh; g->sw = d->sw; g->O = d->O;
  g->Ho = (d->H - d->kh) / d->sh + 1;
  g->Wo = (d->W - d->kw) / d->sw + 1;
  ASR_CHECK((long)g->B * g->H * g->W < 2147483647L && (long)g->kh * g->kw * (g->C > g->O ? g->C : g->O) < 2147483647L, ASR_ERR_SHAPE,
            "conv2d: index space exceeds int32");
  return ASR_OK;
}
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int asr_conv2d_out_dims(const asr_conv_desc* d, int* Ho, int* Wo) {
  ASR_CHECK(d && Ho && Wo, ASR_ERR_ARG, "asr_conv2d_out_dims: null argument");
  ConvGeom g;
  int rc = conv_geom(d, &g);
  if (rc) return rc;
  *Ho = g.Ho; *Wo = g.Wo;
  return ASR_OK;
}

extern "C" int asr_conv2d_fwd(const asr_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                              const uint32_t* drop_seed, uint32_t drop_stream, float drop_rate, void* stream) {
  ASR_CHECK(d && x && w && y, ASR_ERR_ARG, "asr_conv2d_fwd: null argument");
  ConvGeom g;
  int rc = conv_geom(d, &g);
  if (rc) return rc;
  const int M = g.B * g.Ho * g.Wo, N = g.O, K = g.kh * g.kw * g.C;
  Im2colLoader al{x, g, M, K, (g.C % 4 == 0) && al16(x)};
  PlainLoader bl{w, (long)N, K, N, (N % 4 == 0) && al16(w), nullptr, 1};
  GemmEpilogue ep{y, (long)N, M, N, 1.f, bias, nullptr, 1, 0, 0, (drop_rate > 0.f ? drop_seed : nullptr), drop_stream, drop_rate};
  ASR_CHECK(!(drop_rate > 0.f && !drop_seed), ASR_ERR_ARG, "asr_conv2d_fwd: dropout needs a device seed");
  hipStream_t st = (hipStream_t)stream;
  if (N <= 32) {
    const int tm = asr_cdiv(M, 256), tn = asr_cdiv(N, 32);
    hipLaunchKernelGGL((conv_fwd_kernel<256, 32, 4, 1>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, al, bl, ep, K, tm);
  } else {
    const int tm = asr_cdiv(M, 128), tn = asr_cdiv(N, 64);
    hipLaunchKernelGGL((conv_fwd_kernel<128, 64, 2, 2>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, al, bl, ep, K, tm);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// dW += im2col^T dY (dW must be zeroed or hold a running sum: atomic accumulation over K partitions)
extern "C" int asr_conv2d_bwd_filter(const asr_conv_desc* d, const float* x, const float* dy, float* dw, void* stream) {
  ASR_CHECK(d && x && dy && dw, ASR_ERR_ARG, "asr_conv2d_bwd_filter: null argument");
  ConvGeom g;
  int rc = conv_geom(d, &g);
  if (rc) return rc;
  const int M = g.kh * g.kw * g.C, N = g.O, K = g.B * g.Ho * g.Wo;
  Im2colLoader al{x, g, K, M, (g.C % 4 == 0) && al16(x)};
  PlainLoader bl{dy, (long)N, K, N, (N % 4 == 0) && al16(dy), nullptr, 1};
  GemmEpilogue ep{dw, (long)N, M, N, 1.f, nullptr, nullptr, 1, 2, 0, nullptr, 0u, 0.f};
  hipStream_t st = (hipStream_t)stream;
  const bool narrow = N <= 32;
  const int tm = asr_cdiv(M, narrow ? 256 : 64), tn = asr_cdiv(N, narrow ? 32 : 64);
  int splits = 1024 / (tm * tn);
  if (splits < 1) splits = 1;
  int k_chunk = asr_cdiv(asr_cdiv(K, splits), GEMM_BK) * GEMM_BK;
  if (k_chunk < 4 * GEMM_BK) k_chunk = 4 * GEMM_BK;
  splits = asr_cdiv(K, k_chunk);
  ASR_CHECK(splits <= 65535, ASR_ERR_SHAPE, "asr_conv2d_bwd_filter: too many K partitions");
  dim3 grid((unsigned)(tm * tn), 1, (unsigned)splits);
  if (narrow) hipLaunchKernelGGL((conv_bwd_filter_kernel<256, 32, 4, 1>), grid, dim3(256), 0, st, al, bl, ep, K, tm, k_chunk);
  else hipLaunchKernelGGL((conv_bwd_filter_kernel<64, 64, 2, 2>), grid, dim3(256), 0, st, al, bl, ep, K, tm, k_chunk);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_conv2d_bwd_data(const asr_conv_desc* d, const float* dy, const float* w, float* dx, void* stream) {
  ASR_CHECK(d && dy && w && dx, ASR_ERR_ARG, "asr_conv2d_bwd_data: null argument");
  ConvGeom g;
  int rc = conv_geom(d, &g);
  if (rc) return rc;
  const int M = g.B * g.H * g.W, N = g.C, K = g.kh * g.kw * g.O;
  DyGatherLoader al{dy, g, M, K, (g.O % 4 == 0) && al16(dy)};
  WtLoader bl{w, g, N, K, 0};
  GemmEpilogue ep{dx, (long)N, M, N, 1.f, nullptr, nullptr, 1, 0, 0, nullptr, 0u, 0.f};
  hipStream_t st = (hipStream_t)stream;
  if (N <= 32) {
    const int tm = asr_cdiv(M, 256), tn = asr_cdiv(N, 32);
    hipLaunchKernelGGL((conv_bwd_data_kernel<256, 32, 4, 1>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, al, bl, ep, K, tm);
  } else {
    const int tm = asr_cdiv(M, 128), tn = asr_cdiv(N, 64);
    hipLaunchKernelGGL((conv_bwd_data_kernel<128, 64, 2, 2>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, al, bl, ep, K, tm);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
