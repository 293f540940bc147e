// Conv2D (VALID, NHWC, HWIO kernel, linear) forward / filter-gradient / input-gradient as implicit
// GEMMs on the f32 MFMA core (gemm_core.h).  Replaces tf.keras.layers.Conv2D at las.py:163-164,
// 183-184 and deepspeech2.py:47-50, 57-59 and the gradients TF derives for them.
//
//   forward      Y[pos][o]   = sum_kc im2col[pos][kc] W[kc][o] + b[o]            (NN, A gathered)
//   bwd filter   dW[kc][o]  += sum_pos im2col[pos][kc] dY[pos][o]                (TN, A gathered, split-K)
//   bwd data     dX[ipos][c] = sum_{r,s,o} dY[b,(h-r)/sh,(w-s)/sw,o] W[r,s,c,o]   (NT, both gathered; one
//                stride-1 correlation per (h % sh, w % sw) class, see DyClassLoader)
// pos = (b, ho, wo); kc = (r, s, c) with c fastest - exactly the HWIO kernel's row index.
#include <stdlib.h>

#include <stdlib.h>
#include <string.h>

#include "gemm_core.h"

struct ConvGeom {
  int B, H, W, C, kh, kw, sh, sw, O, Ho, Wo;
};

// stored logical matrix im2col [R = B*Ho*Wo][Cc = kh*kw*C]
template <int VEC>   // VEC: C % 4 == 0 and x 16-byte aligned - a float4 of columns stays inside one kernel row
struct Im2colLoader {
  const float* x;
  ConvGeom g;
  int R, Cc;
  AsrDiv dWo, dHo, dRow;            // divisors Wo, Ho and kw*C (one kernel row (s, c) is contiguous in the input)
  struct Row { long base; int ok; };
  struct Col { long off[VEC ? 1 : 4]; int n; };   // !VEC: one offset per element (a group of 4 may straddle kernel rows)
  __device__ __forceinline__ Row row(int r) const {
    uint32_t t, wo, b, ho;
    dWo.divmod((uint32_t)min(r, R - 1), t, wo);
    dHo.divmod(t, b, ho);
    return Row{(((long)b * g.H + (long)ho * g.sh) * g.W + (long)wo * g.sw) * g.C, r < R};
  }
  __device__ __forceinline__ Col col(int c) const {
    Col cl;
    cl.n = max(0, min(4, Cc - c));
#pragma unroll
    for (int i = 0; i < (VEC ? 1 : 4); ++i) {
      uint32_t kr, rem;
      dRow.divmod((uint32_t)max(0, min(c + i, VEC ? Cc - 4 : Cc - 1)), kr, rem);   // clamped: the load is always legal
      cl.off[i] = (long)kr * g.W * g.C + rem;
    }
    return cl;
  }
  __device__ __forceinline__ void fetch4(const Row& rw, const Col& cl, float (&v)[4]) const {
    if (VEC) {                      // Cc % 4 == 0 too (C % 4 == 0): a fetch is all in or all out
      const float4 t4 = *reinterpret_cast<const float4*>(x + rw.base + cl.off[0]);
      const bool ok = rw.ok && cl.n == 4;
      v[0] = ok ? t4.x : 0.f; v[1] = ok ? t4.y : 0.f; v[2] = ok ? t4.z : 0.f; v[3] = ok ? t4.w : 0.f;
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (rw.ok && i < cl.n) ? x[rw.base + cl.off[i]] : 0.f;
  }
};

// bwd-data by stride class.  Input positions with h % sh == ph and w % sw == pw receive only the filter
// taps r = ph + sh*i, s = pw + sw*j, and for all of them the same ones: with h = hq*sh + ph, w = wq*sw + pw
//   dX[b, h, w, c] = sum_{i, j, o} dY[b, hq - i, wq - j, o] * W[ph + sh*i, pw + sw*j, c, o]
// i.e. one stride-1 correlation per class over the sub-sampled filter - no multiplications by the zeros a
// strided gather would insert (sh*sw times fewer MFMAs than the dense form).
// Rows are ordered (b, wq, hq) and columns (j, i, o): a tile of consecutive rows then covers one or two
// values of wq, for which only the taps j in [wq - (Wo-1), wq] can hit dY at all, and those taps are one
// contiguous K range - the kernel restricts its K loop to it (the frequency axis is short against the
// kernel width in DeepSpeech2: 40 % of the taps fall off the edge).
// A operand of class (ph, pw): G [R = B*Wq*Hq][Cc = nS*nR*O], G[(b,wq,hq)][(j,i,o)] = dY[b, hq-i, wq-j, o] or 0
template <int VEC>    // VEC: O % 4 == 0 and 16-byte aligned dY / W (the four o of a fetch share their tap)
struct DyClassLoader {
  const float* dy;
  ConvGeom g;
  int Hq, Wq, nR;     // class grid and taps per kernel column
  int R, Cc;
  AsrDiv dHq, dWq, dO, dnR;
  struct Row { long bbase; int hq, wq, ok; };
  struct Col { int o[VEC ? 1 : 4], i[VEC ? 1 : 4], j[VEC ? 1 : 4]; int n; };
  __device__ __forceinline__ Row row(int r) const {
    uint32_t t, hq, b, wq;
    dHq.divmod((uint32_t)min(r, R - 1), t, hq);
    dWq.divmod(t, b, wq);
    return Row{(long)b * g.Ho * g.Wo * g.O, (int)hq, (int)wq, r < R};
  }
  __device__ __forceinline__ Col col(int c) const {
    Col cl;
    cl.n = max(0, min(4, Cc - c));
#pragma unroll
    for (int e = 0; e < (VEC ? 1 : 4); ++e) {
      uint32_t ij, o, j, i;
      dO.divmod((uint32_t)max(0, min(c + e, VEC ? Cc - 4 : Cc - 1)), ij, o);
      dnR.divmod(ij, j, i);
      cl.o[e] = (int)o; cl.i[e] = (int)i; cl.j[e] = (int)j;
    }
    return cl;
  }
  __device__ __forceinline__ void fetch4(const Row& rw, const Col& cl, float (&v)[4]) const {
    if (VEC) {
      const int ho = rw.hq - cl.i[0], wo = rw.wq - cl.j[0];
      const bool ok = rw.ok && cl.n == 4 && ho >= 0 && wo >= 0 && ho < g.Ho && wo < g.Wo;
      const int hc = min(max(ho, 0), g.Ho - 1), wc = min(max(wo, 0), g.Wo - 1);     // clamped: the load is always legal
      const float4 t4 = *reinterpret_cast<const float4*>(dy + rw.bbase + ((long)hc * g.Wo + wc) * g.O + cl.o[0]);
      v[0] = ok ? t4.x : 0.f; v[1] = ok ? t4.y : 0.f; v[2] = ok ? t4.z : 0.f; v[3] = ok ? t4.w : 0.f;
      return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ho = rw.hq - cl.i[e], wo = rw.wq - cl.j[e];
      const bool ok = rw.ok && e < cl.n && ho >= 0 && wo >= 0 && ho < g.Ho && wo < g.Wo;
      v[e] = ok ? dy[rw.bbase + ((long)ho * g.Wo + wo) * g.O + cl.o[e]] : 0.f;
    }
  }
};

// B operand of class (ph, pw): Wt [R = C][Cc = nS*nR*O], Wt[c][(j,i,o)] = W[ph + sh*i, pw + sw*j, c, o]
template <int VEC>
struct WtClassLoader {
  const float* w;
  ConvGeom g;
  int ph, pw, nR;
  int R, Cc;
  AsrDiv dO, dnR;
  struct Row { long off; int ok; };
  struct Col { long off[VEC ? 1 : 4]; int n; };
  __device__ __forceinline__ Row row(int r) const { return Row{(long)min(r, R - 1) * g.O, r < R}; }
  __device__ __forceinline__ Col col(int c) const {
    Col cl;
    cl.n = max(0, min(4, Cc - c));
#pragma unroll
    for (int e = 0; e < (VEC ? 1 : 4); ++e) {
      uint32_t ij, o, j, i;
      dO.divmod((uint32_t)max(0, min(c + e, VEC ? Cc - 4 : Cc - 1)), ij, o);
      dnR.divmod(ij, j, i);
      cl.off[e] = ((long)(ph + g.sh * (int)i) * g.kw + (pw + g.sw * (int)j)) * g.C * g.O + o;
    }
    return cl;
  }
  __device__ __forceinline__ void fetch4(const Row& rw, const Col& cl, float (&v)[4]) const {
    if (VEC) {     // the four o are contiguous in the HWIO kernel
      const float4 t4 = *reinterpret_cast<const float4*>(w + cl.off[0] + rw.off);
      const bool ok = rw.ok && cl.n == 4;
      v[0] = ok ? t4.x : 0.f; v[1] = ok ? t4.y : 0.f; v[2] = ok ? t4.z : 0.f; v[3] = ok ? t4.w : 0.f;
      return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (rw.ok && e < cl.n) ? w[cl.off[e] + rw.off] : 0.f;
  }
};

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int VEC, int SP>
__global__ __launch_bounds__(256) void conv_fwd_kernel(Im2colLoader<VEC> al, PlainLoader bl, GemmEpilogue ep, int K, int tiles_m) {
  using T = GemmTile<0, 0, BM, BN, WAVES_M, WAVES_N, BK>;
  __shared__ __attribute__((aligned(16))) float As[T::template a_lds<(SP ? 2 : 0)>()];
  __shared__ __attribute__((aligned(16))) float Bs[T::template b_lds<(SP ? 2 : 0)>()];
  const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;
  if constexpr (SP) T::template run_split<SP>(al, bl, ep, 0, K, bm * BM, bn * BN, As, Bs);
  else T::template run<0>(al, bl, ep, 0, K, bm * BM, bn * BN, As, Bs);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int VEC, int SP>
__global__ __launch_bounds__(256) void conv_bwd_filter_kernel(Im2colLoader<VEC> al, PlainLoader bl, GemmEpilogue ep, int K,
                                                              int tiles_m, int k_chunk) {
  using T = GemmTile<1, 0, BM, BN, WAVES_M, WAVES_N, BK>;
  __shared__ __attribute__((aligned(16))) float As[T::template a_lds<(SP ? 2 : 0)>()];
  __shared__ __attribute__((aligned(16))) float Bs[T::template b_lds<(SP ? 2 : 0)>()];
  const int kbeg = blockIdx.z * k_chunk, kend = min(K, kbeg + k_chunk);
  if (kbeg >= K) return;
  const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;
  if constexpr (SP) T::template run_split<SP>(al, bl, ep, kbeg, kend, bm * BM, bn * BN, As, Bs);
  else T::template run<0>(al, bl, ep, kbeg, kend, bm * BM, bn * BN, As, Bs);
}

// all stride classes of one data-gradient in ONE launch (class = blockIdx.z): their tiles fill the chip
// together instead of leaving a ragged tail per class
#define CONV_MAX_CLASSES 4
template <int VEC>
struct ConvClass {
  DyClassLoader<VEC> al;
  WtClassLoader<VEC> bl;
  GemmEpilogue ep;
  int K, tiles_m, tiles, k_per_j;
};
template <int VEC>
struct ConvClassSet { ConvClass<VEC> c[CONV_MAX_CLASSES]; };

// blockIdx.y = K partition (gridDim.y > 1: the partitions of a tile add up atomically into a pre-zeroed dX - few large tiles leave the
// last round of workgroups mostly empty: deepspeech conv3 has 558 tiles of 363 K steps for 512 resident workgroups)
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int VEC, int SP>
__global__ __launch_bounds__(256) void conv_bwd_data_kernel(ConvClassSet<VEC> cs) {
  using T = GemmTile<0, 1, BM, BN, WAVES_M, WAVES_N, BK>;
  __shared__ __attribute__((aligned(16))) float As[T::template a_lds<(SP ? 2 : 0)>()];
  __shared__ __attribute__((aligned(16))) float Bs[T::template b_lds<(SP ? 2 : 0)>()];
  const ConvClass<VEC>& cc = cs.c[blockIdx.z];
  if ((int)blockIdx.x >= cc.tiles) return;
  DyClassLoader<VEC> al = cc.al;
  WtClassLoader<VEC> bl = cc.bl;
  const int K = cc.K, tiles_m = cc.tiles_m, k_per_j = cc.k_per_j;
  const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;
  int kbeg = 0, kend = K;
  if (k_per_j > 0) {   // taps j outside [wq - (Wo-1), wq] read only zeros for every row of this tile: skip them
    const int r0 = bm * BM, r1 = min(r0 + BM, al.R) - 1;
    const int t0 = r0 / al.Hq, t1 = r1 / al.Hq;
    if (t0 / al.Wq == t1 / al.Wq) {                 // the tile stays inside one clip: wq runs from t0 % Wq to t1 % Wq
      const int j_lo = max(0, t0 % al.Wq - (al.g.Wo - 1)), j_hi = min(K / k_per_j - 1, t1 % al.Wq);
      kbeg = j_lo * k_per_j;
      kend = max(kbeg, (j_hi + 1) * k_per_j);
      al.Cc = kend;      // the loaders zero-fill from kend on, so the last K tile may be partial
      bl.Cc = kend;
    }
  }
  if (gridDim.y > 1) {
    const int chunk = ((kend - kbeg + (int)gridDim.y - 1) / (int)gridDim.y + BK - 1) / BK * BK;
    kbeg += (int)blockIdx.y * chunk;
    if (kbeg >= kend) return;                       // (dX is pre-zeroed: nothing to add)
    kend = min(kend, kbeg + chunk);
    al.Cc = kend;
    bl.Cc = kend;
  }
  if constexpr (SP) T::template run_split<SP>(al, bl, cc.ep, kbeg, kend, bm * BM, bn * BN, As, Bs);
  else T::template run<0>(al, bl, cc.ep, kbeg, kend, bm * BM, bn * BN, As, Bs);
}

// How the convolutions evaluate their f32 products (asr_set_f32_product_mode; the same three evaluations as asr_gemm_desc.compute 0 / 2 / 3:
// f32 MFMA, nine or six bf16 pair products of exact three-way operand splits on the bf16 MFMA - gemm_core.h run_split)
static int g_conv_mode = -1;
static int conv_mode() {
  if (g_conv_mode < 0) {
    const char* e = getenv("ASR_GEMM_F32");
    g_conv_mode = (e && !strcmp(e, "split9")) ? 2 : ((e && !strcmp(e, "mfma")) ? 0 : 3);      // default: six pairs (ops.py has the same default)
  }
  return g_conv_mode;
}
int conv_product_mode() { return conv_mode(); }               // (conv_halo.hip)
extern "C" int asr_set_f32_product_mode(int mode) {
  ASR_CHECK(mode == 0 || mode == 2 || mode == 3, ASR_ERR_ARG, "asr_set_f32_product_mode: 0 (f32 MFMA), 2 (nine bf16 pair products) or 3 (six), got %d", mode);
  const int old = conv_mode();
  g_conv_mode = mode;
  return old;
}
#define CONV_UNPACK(...) __VA_ARGS__
#define CONV_GO(KERN, P, ...)                                                             \
  do {                                                                                    \
    const int mode__ = conv_mode();                                                       \
    if (mode__ == 2) hipLaunchKernelGGL((KERN<CONV_UNPACK P, 9>), __VA_ARGS__);            \
    else if (mode__ == 3) hipLaunchKernelGGL((KERN<CONV_UNPACK P, 6>), __VA_ARGS__);       \
    else hipLaunchKernelGGL((KERN<CONV_UNPACK P, 0>), __VA_ARGS__);                        \
  } while (0)

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
  PlainLoader bl{w, (long)N, K, N, (N % 4 == 0) && al16(w), nullptr, 1};
  GemmEpilogue ep{y, (long)N, M, N, 1.f, bias, nullptr, 1, 0, 0, (drop_rate > 0.f ? drop_seed : nullptr), drop_stream, drop_rate};
  ASR_CHECK(!(drop_rate > 0.f && !drop_seed), ASR_ERR_ARG, "asr_conv2d_fwd: dropout needs a device seed");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (g.C % 4 == 0) && al16(x);
  const AsrDiv dWo = asr_make_div(g.Wo), dHo = asr_make_div(g.Ho), dRow = asr_make_div(g.kw * g.C);
  Im2colLoader<1> av{x, g, M, K, dWo, dHo, dRow};
  Im2colLoader<0> as{x, g, M, K, dWo, dHo, dRow};
  static const int fwd_small = getenv("ASR_CONV_FWD_SMALL") ? atoi(getenv("ASR_CONV_FWD_SMALL")) : 1;
  static const int narrow_bm = getenv("ASR_CONV_FWD_BM") ? atoi(getenv("ASR_CONV_FWD_BM")) : 128;
  if (N <= 32 && narrow_bm == 128) {
    const int tm = asr_cdiv(M, 128), tn = asr_cdiv(N, 32);
    if (vec) CONV_GO(conv_fwd_kernel, (128, 32, 4, 1, 32, 1), dim3((unsigned)(tm * tn)), dim3(256), 0, st, av, bl, ep, K, tm);
    else CONV_GO(conv_fwd_kernel, (128, 32, 4, 1, 32, 0), dim3((unsigned)(tm * tn)), dim3(256), 0, st, as, bl, ep, K, tm);
  } else if (N <= 32) {
    const int tm = asr_cdiv(M, 256), tn = asr_cdiv(N, 32);
    if (vec) CONV_GO(conv_fwd_kernel, (256, 32, 4, 1, 32, 1), dim3((unsigned)(tm * tn)), dim3(256), 0, st, av, bl, ep, K, tm);
    else CONV_GO(conv_fwd_kernel, (256, 32, 4, 1, 32, 0), dim3((unsigned)(tm * tn)), dim3(256), 0, st, as, bl, ep, K, tm);
  } else if (vec && fwd_small) {
    // 64 x 64 tiles: deepspeech conv3 (M = 40320, N = 96) makes 630 workgroups of 128 x 64 - a full round and a 23 % one at two per CU;
    // 1260 smaller ones balance better (922 -> 840 us); one 128-wide column tile (im2col gathered once) measured 1209 us
    const int tm = asr_cdiv(M, 64), tn = asr_cdiv(N, 64);
    CONV_GO(conv_fwd_kernel, (64, 64, 2, 2, 32, 1), dim3((unsigned)(tm * tn)), dim3(256), 0, st, av, bl, ep, K, tm);
  } else {
    const int tm = asr_cdiv(M, 128), tn = asr_cdiv(N, 64);
    if (vec) CONV_GO(conv_fwd_kernel, (128, 64, 2, 2, 32, 1), dim3((unsigned)(tm * tn)), dim3(256), 0, st, av, bl, ep, K, tm);
    else CONV_GO(conv_fwd_kernel, (128, 64, 2, 2, 32, 0), dim3((unsigned)(tm * tn)), dim3(256), 0, st, as, bl, ep, K, tm);
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
  PlainLoader bl{dy, (long)N, K, N, (N % 4 == 0) && al16(dy), nullptr, 1};
  GemmEpilogue ep{dw, (long)N, M, N, 1.f, nullptr, nullptr, 1, 2, 0, nullptr, 0u, 0.f};
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (g.C % 4 == 0) && al16(x);
  const AsrDiv dWo = asr_make_div(g.Wo), dHo = asr_make_div(g.Ho), dRow = asr_make_div(g.kw * g.C);
  Im2colLoader<1> av{x, g, K, M, dWo, dHo, dRow};
  Im2colLoader<0> as{x, g, K, M, dWo, dHo, dRow};
  const bool narrow = N <= 32;
  const bool tiny = M <= 32;     // a 3x3x3 first-layer filter: 27 rows - a 32-row tile wastes far fewer MFMAs than a 256-row one
  static const int wide_env = getenv("ASR_CONV_DW_WIDE") ? atoi(getenv("ASR_CONV_DW_WIDE")) : 1;
  const bool wide = !tiny && !narrow && N > 64 && N <= 128 && wide_env != 0;   // 96 output channels: ONE 128-wide column tile gathers im2col once
  const int wbm = wide_env == 2 ? 64 : 128;
  static const int dw_bm = getenv("ASR_CONV_DW_BM") ? atoi(getenv("ASR_CONV_DW_BM")) : 128;
  const int tm = asr_cdiv(M, tiny ? 32 : (narrow ? dw_bm : (wide ? wbm : 64))), tn = asr_cdiv(N, tiny ? 128 : (narrow ? 32 : (wide ? 128 : 64)));
  int splits = 1024 / (tm * tn);
  if (splits < 1) splits = 1;
  const int bk = 32;                         // K partitions are whole K tiles of the kernel used
  int k_chunk = asr_cdiv(asr_cdiv(K, splits), bk) * bk;
  if (k_chunk < 4 * bk) k_chunk = 4 * bk;
  splits = asr_cdiv(K, k_chunk);
  ASR_CHECK(splits <= 65535, ASR_ERR_SHAPE, "asr_conv2d_bwd_filter: too many K partitions");
  dim3 grid((unsigned)(tm * tn), 1, (unsigned)splits);
  if (tiny && vec) CONV_GO(conv_bwd_filter_kernel, (32, 128, 1, 4, 32, 1), grid, dim3(256), 0, st, av, bl, ep, K, tm, k_chunk);
  else if (tiny) CONV_GO(conv_bwd_filter_kernel, (32, 128, 1, 4, 32, 0), grid, dim3(256), 0, st, as, bl, ep, K, tm, k_chunk);
  else if (narrow && vec && dw_bm == 128) CONV_GO(conv_bwd_filter_kernel, (128, 32, 4, 1, 32, 1), grid, dim3(256), 0, st, av, bl, ep, K, tm, k_chunk);
  else if (narrow && dw_bm == 128) CONV_GO(conv_bwd_filter_kernel, (128, 32, 4, 1, 32, 0), grid, dim3(256), 0, st, as, bl, ep, K, tm, k_chunk);
  else if (narrow && vec) CONV_GO(conv_bwd_filter_kernel, (256, 32, 4, 1, 32, 1), grid, dim3(256), 0, st, av, bl, ep, K, tm, k_chunk);
  else if (narrow) CONV_GO(conv_bwd_filter_kernel, (256, 32, 4, 1, 32, 0), grid, dim3(256), 0, st, as, bl, ep, K, tm, k_chunk);
  else if (wide && vec && wbm == 128) CONV_GO(conv_bwd_filter_kernel, (128, 128, 2, 2, 32, 1), grid, dim3(256), 0, st, av, bl, ep, K, tm, k_chunk);
  else if (wide && vec) CONV_GO(conv_bwd_filter_kernel, (64, 128, 2, 2, 32, 1), grid, dim3(256), 0, st, av, bl, ep, K, tm, k_chunk);
  else if (vec) CONV_GO(conv_bwd_filter_kernel, (64, 64, 2, 2, 32, 1), grid, dim3(256), 0, st, av, bl, ep, K, tm, k_chunk);
  else CONV_GO(conv_bwd_filter_kernel, (64, 64, 2, 2, 32, 0), grid, dim3(256), 0, st, as, bl, ep, K, tm, k_chunk);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_conv2d_bwd_data(const asr_conv_desc* d, const float* dy, const float* w, float* dx, void* stream) {
  ASR_CHECK(d && dy && w && dx, ASR_ERR_ARG, "asr_conv2d_bwd_data: null argument");
  ConvGeom g;
  int rc = conv_geom(d, &g);
  if (rc) return rc;
  const int N = g.C;
  hipStream_t st = (hipStream_t)stream;
  const int o_vec = (g.O % 4 == 0) && al16(dy) && al16(w);
  const bool narrow = N <= 32;
  static const int dx_bm = getenv("ASR_CONV_DX_BM") ? atoi(getenv("ASR_CONV_DX_BM")) : 128;
  const int BMh = narrow ? dx_bm : 128, BNh = narrow ? 32 : 64;
  // classes in chunks of CONV_MAX_CLASSES per launch (sh * sw <= 4 for every shipped model: one launch)
  ConvClassSet<1> sv{};
  ConvClassSet<0> ss{};
  int n = 0, max_tiles = 0;
  // K partitions: when all classes together make fewer than ~8 workgroups per CU and K is long, split K (atomic accumulation into a
  // zeroed dX).  deepspeech conv3: 558 tiles x 363 K steps -> 4 partitions (1447 -> see DESIGN.md); conv2: 1598 -> 2
  int splits = 1;
  {
    static const int split_env = getenv("ASR_CONV_DX_SPLIT") ? atoi(getenv("ASR_CONV_DX_SPLIT")) : -1;
    long tiles_all = 0;
    int kmax = 0;
    for (int ph = 0; ph < g.sh; ++ph)
      for (int pw = 0; pw < g.sw; ++pw) {
        const int Hq = (g.H - ph + g.sh - 1) / g.sh, Wq = (g.W - pw + g.sw - 1) / g.sw;
        if (Hq <= 0 || Wq <= 0) continue;
        const int nR = ph < g.kh ? (g.kh - ph + g.sh - 1) / g.sh : 0, nS = pw < g.kw ? (g.kw - pw + g.sw - 1) / g.sw : 0;
        tiles_all += (long)asr_cdiv((long)g.B * Hq * Wq, BMh) * asr_cdiv(N, BNh);
        kmax = nR * nS * g.O > kmax ? nR * nS * g.O : kmax;
      }
    if (kmax >= 2048 && tiles_all > 0 && tiles_all < 2048) splits = (int)((2048 + tiles_all - 1) / tiles_all);
    if (splits > 4) splits = 4;
    if (split_env >= 1) splits = split_env;
    if ((long)g.sh * g.sw > CONV_MAX_CLASSES) splits = 1;   // (several launches over one dX: keep the plain stores)
  }
  if (splits > 1 && asr_zero_async(dx, sizeof(float) * (size_t)g.B * g.H * g.W * g.C, st) != hipSuccess) {
    asr_set_error("asr_conv2d_bwd_data: zeroing dX failed");
    return ASR_ERR_HIP;
  }
  auto flush = [&]() {
    if (n == 0) return;
    dim3 grid((unsigned)max_tiles, (unsigned)splits, (unsigned)n);
    if (narrow && o_vec && BMh == 128) CONV_GO(conv_bwd_data_kernel, (128, 32, 4, 1, 32, 1), grid, dim3(256), 0, st, sv);
    else if (narrow && BMh == 128) CONV_GO(conv_bwd_data_kernel, (128, 32, 4, 1, 32, 0), grid, dim3(256), 0, st, ss);
    else if (narrow && o_vec) CONV_GO(conv_bwd_data_kernel, (256, 32, 4, 1, 32, 1), grid, dim3(256), 0, st, sv);
    else if (narrow) CONV_GO(conv_bwd_data_kernel, (256, 32, 4, 1, 32, 0), grid, dim3(256), 0, st, ss);
    else if (o_vec) CONV_GO(conv_bwd_data_kernel, (128, 64, 2, 2, 32, 1), grid, dim3(256), 0, st, sv);
    else CONV_GO(conv_bwd_data_kernel, (128, 64, 2, 2, 32, 0), grid, dim3(256), 0, st, ss);
    n = 0; max_tiles = 0;
  };
  for (int ph = 0; ph < g.sh; ++ph)
    for (int pw = 0; pw < g.sw; ++pw) {
      const int Hq = (g.H - ph + g.sh - 1) / g.sh, Wq = (g.W - pw + g.sw - 1) / g.sw;
      if (Hq <= 0 || Wq <= 0) continue;
      const int nR = ph < g.kh ? (g.kh - ph + g.sh - 1) / g.sh : 0, nS = pw < g.kw ? (g.kw - pw + g.sw - 1) / g.sw : 0;
      const int M = g.B * Hq * Wq, K = nR * nS * g.O;     // K == 0 (stride > kernel): the class is written as zeros
      const int nRd = nR > 0 ? nR : 1;
      const AsrDiv dHq = asr_make_div(Hq), dWq = asr_make_div(Wq), dO = asr_make_div(g.O), dnR = asr_make_div(nRd);
      const int k_per_j = (K > 0 && (nR * g.O) % 4 == 0) ? nR * g.O : 0;   // K ranges start on a float4 boundary
      const int tm = asr_cdiv(M, BMh), tn = asr_cdiv(N, BNh);
      GemmEpilogue ep{dx, (long)N, M, N, 1.f, nullptr, nullptr, 1, splits > 1 ? 2 : 0, 0, nullptr, 0u, 0.f, 1, Hq, Wq, g.sh, g.sw, ph, pw, g.H, g.W};
      sv.c[n] = ConvClass<1>{DyClassLoader<1>{dy, g, Hq, Wq, nRd, M, K, dHq, dWq, dO, dnR}, WtClassLoader<1>{w, g, ph, pw, nRd, N, K, dO, dnR}, ep,
                             K, tm, tm * tn, k_per_j};
      ss.c[n] = ConvClass<0>{DyClassLoader<0>{dy, g, Hq, Wq, nRd, M, K, dHq, dWq, dO, dnR}, WtClassLoader<0>{w, g, ph, pw, nRd, N, K, dO, dnR}, ep,
                             K, tm, tm * tn, k_per_j};
      max_tiles = tm * tn > max_tiles ? tm * tn : max_tiles;
      if (++n == CONV_MAX_CLASSES) flush();
    }
  flush();
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
