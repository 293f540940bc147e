// Convolutions whose kernel slides with stride 1 along W (deepspeech2.py:47-50: conv2 / conv3, kernels 21 x 11, strides (2, 1)), forward and
// input gradient, with the input ROW staged once per kernel row and shared by all kw taps - the "LDS halo" form of the implicit GEMM.
//
// Why: conv.hip gathers im2col tiles - output position x one tap's 32 channels per K tile - so every input element is fetched, SPLIT into its
// three bf16 planes (the f32 products run as bf16 pair products, gemm_core.h run_split) and written to LDS once per tap that touches it:
// 11 times along W.  With 32 output channels a K tile carries 12 MFMAs per wave against ~120 vector instructions of splitting per lane, so the
// kernels were bound by the split, not by the matrix pipe (conv2 forward: 67 GFLOP in 747 us = 22 % of the bf16 pipe for six pair products).
//
// Here a workgroup owns RG output ROW GROUPS (a row group = the Wout positions of one (b, h)) x 32 output channels.  For each (kernel row r,
// 32-channel chunk) it stages the RG input rows it needs - each Wout + S - 1 positions x 32 channels, split ONCE - as an image
// [group][position][32 ch] of three bf16 planes, and the S taps of that kernel row read it at row offsets 0 .. S - 1: output (g, w), tap s
// reads image row g * GW + w + s.  The weights arrive pre-split and pre-ordered (asr_conv2d_halo_pack: the LDS image of every (r, chunk) block,
// one memory-bound pass over the kernel per step) and are copied 16 bytes at a time.  Per (r, chunk): 24 S MFMAs per wave (S = 11: 264)
// against ~45 split elements per lane - the split is amortised S-fold and the matrix pipe sets the time.
//
// The input gradient is the same computation on dY: with h = sh * hq + ph (conv.hip, DyClassLoader)
//   dX[b, sh hq + ph, w, c] = sum_{i, s, o} dY[b, hq - i, w - s, o] W[ph + sh i, s, c, o]
// one launch per class ph: input = dY (zero outside), taps along W reversed (s' = S - 1 - s reads position w + s' - (S - 1)), input row
// hq - i, reduction over o, 32-wide output tiles over c.
#include <stdlib.h>
#include <string.h>

#include "gemm_core.h"

typedef unsigned short bf16_t;

struct HaloArgs {
  const float* in;              // [B][Hin][Win][Cin]
  const bf16_t* wp;             // packed weights: [n tile][J = NR * NC][S][3 planes][32 n][32 k], each plane in LDS order (halo_b_off)
  float* out;                   // [B][Hfull][Wout][N]
  const float* bias;            // [N] or null
  int B, Hin, Win, Cin;
  int Hout, Wout, N;            // row groups per batch entry, positions per row group, output channels
  int S, pad, GW;               // taps along W; image position u holds input w = u - pad; GW = Wout + S - 1 positions per row group
  int RG;                       // row groups per workgroup
  int NR, NC;                   // J = NR kernel rows x NC 32-channel chunks
  int h_mul, r_mul;             // input row of (row group h, kernel row jr) = h * h_mul + jr * r_mul
  int Hfull, oh_mul, oh_off;    // stored output row of row group h = h * oh_mul + oh_off (of Hfull)
  int groups;                   // B * Hout
  int dbg;                      // timing experiments (ASR_CONV_HALO_DBG): 1 no products, 2 one staging only, 3 one load only
  AsrDiv dGW, dWout, dHout;
};

__device__ __forceinline__ int halo_a_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// TERMS = 6 / 9 (gemm_core.h run_split).  256 threads: wave v owns output rows 64 v .. 64 v + 63 of the tile (two 32 x 32 MFMA tiles) x 32 channels.
template <int TERMS, int MAXF, int MAXB>
__global__ __launch_bounds__(256) void conv_halo_kernel(HaloArgs a) {
  extern __shared__ __attribute__((aligned(16))) char halo_lds[];
  using T = GemmTile<0, 0, 128, 32, 4, 1>;                       // (split3 / pack2 only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int rows = a.RG * a.GW;                                  // image rows
  const int a_plane = rows * 64;
  char* Ab = halo_lds;
  char* Bb = halo_lds + 3 * a_plane;
  const int g0 = blockIdx.x * a.RG, n0 = blockIdx.y * 32;
  const int J = a.NR * a.NC;
  const bf16_t* wp = a.wp + (long)blockIdx.y * J * a.S * 3072;   // 3 planes x 32 x 32 elements per tap
  const int nfa = (rows * 8 + 255) >> 8;                         // float4 fetches per thread and (r, chunk): image row idx >> 3, channels 4 (idx & 7)
  const int nfb = (a.S * 384 + 255) >> 8;                        // 16-byte weight chunks per thread
  float4 ra[MAXF];
  uint4 rb[MAXB];
  auto gload = [&](int j) {
    const int jr = j / a.NC, cc = j - jr * a.NC;
#pragma unroll
    for (int f = 0; f < MAXF; ++f) {
      ra[f] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < nfa) {
        const int idx = tid + 256 * f;
        uint32_t g, u, b, h;
        a.dGW.divmod((uint32_t)min(idx >> 3, rows - 1), g, u);
        a.dHout.divmod((uint32_t)min(g0 + (int)g, a.groups - 1), b, h);
        const int w = (int)u - a.pad, hin = (int)h * a.h_mul + jr * a.r_mul;
        const bool ok = (idx >> 3) < rows && g0 + (int)g < a.groups && w >= 0 && w < a.Win && hin >= 0 && hin < a.Hin;
        if (ok) ra[f] = *reinterpret_cast<const float4*>(a.in + (((long)b * a.Hin + hin) * a.Win + w) * a.Cin + cc * 32 + 4 * (idx & 7));
      }
    }
    const uint4* src = reinterpret_cast<const uint4*>(wp + (long)j * a.S * 3072);
#pragma unroll
    for (int f = 0; f < MAXB; ++f) {
      const int idx = tid + 256 * f;
      rb[f] = (f < nfb && idx < a.S * 384) ? src[idx] : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int f = 0; f < MAXF; ++f) {
      const int idx = tid + 256 * f, row = idx >> 3, q = idx & 7;
      if (f < nfa && row < rows) {
        unsigned h[4], m[4], l[4];
        T::split3(ra[f].x, h[0], m[0], l[0]); T::split3(ra[f].y, h[1], m[1], l[1]);
        T::split3(ra[f].z, h[2], m[2], l[2]); T::split3(ra[f].w, h[3], m[3], l[3]);
        const int off = halo_a_off(row, q >> 1) | ((q & 1) << 3);
        *reinterpret_cast<uint2*>(Ab + off) = make_uint2(T::pack2(h[0], h[1]), T::pack2(h[2], h[3]));
        *reinterpret_cast<uint2*>(Ab + a_plane + off) = make_uint2(T::pack2(m[0], m[1]), T::pack2(m[2], m[3]));
        *reinterpret_cast<uint2*>(Ab + 2 * a_plane + off) = make_uint2(T::pack2(l[0], l[1]), T::pack2(l[2], l[3]));
      }
    }
#pragma unroll
    for (int f = 0; f < MAXB; ++f) {
      const int idx = tid + 256 * f;
      if (f < nfb && idx < a.S * 384) *reinterpret_cast<uint4*>(Bb + idx * 16) = rb[f];
    }
  };
  // this lane's two output rows (MFMA A rows): tile row m -> (group, position) -> image row of tap 0; rows beyond the tile read image row 0
  int r0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uint32_t g, w;
    a.dWout.divmod((uint32_t)(wave * 64 + i * 32 + l31), g, w);
    r0[i] = (int)g < a.RG ? (int)g * a.GW + (int)w : 0;
  }
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int boff = l31 * 64;                                     // B row n = l31 of a tap's plane
  const int bsw = (l31 >> 2) & 3;

  bf16x8 fa0[3][2], fb0[3], fa1[3][2], fb1[3];
  auto rd = [&](int s, int kk, bf16x8 (&fa)[3][2], bf16x8 (&fb)[3]) {
    const char* Bs = Bb + s * 6144 + boff + (((2 * kk + lh) ^ bsw) << 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = halo_a_off(r0[i] + s, 2 * kk + lh);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fa[pl][i] = *reinterpret_cast<const bf16x8*>(Ab + pl * a_plane + off);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fb[pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * 2048);
  };
  auto mm = [&](const bf16x8 (&fa)[3][2], const bf16x8 (&fb)[3]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16 c = acc[i];
      if (TERMS == 9) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[1], c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0], c, 0, 0, 0);
      acc[i] = c;
    }
  };
  if (J > 0) gload(0);
  for (int j = 0; j < J; ++j) {
    __syncthreads();                                             // every wave is done with the image of j - 1
    if (a.dbg != 2 || j == 0) lstore();
    __syncthreads();
    if (j + 1 < J && (a.dbg != 3 || j == 0)) gload(j + 1);       // in flight during the products of j
    if (a.dbg == 1) continue;
    // the S taps x two 16-deep steps of this image, fragments read ONE STEP AHEAD (one wave per SIMD: nothing else hides the LDS latency)
    rd(0, 0, fa0, fb0);
    for (int s = 0; s < a.S; ++s) {
      rd(s, 1, fa1, fb1);
      mm(fa0, fb0);
      if (s + 1 < a.S) rd(s + 1, 0, fa0, fb0);
      mm(fa1, fb1);
    }
  }
  // C: column n = l31, row = (r & 3) + 8 (r >> 2) + 4 lh of the 32 x 32 tile
  const int n = n0 + l31;
  const float bv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      uint32_t g, w, b, h;
      a.dWout.divmod((uint32_t)m, g, w);
      if ((int)g >= a.RG || g0 + (int)g >= a.groups || n >= a.N) continue;
      a.dHout.divmod((uint32_t)(g0 + (int)g), b, h);
      a.out[(((long)b * a.Hfull + (long)h * a.oh_mul + a.oh_off) * a.Wout + w) * a.N + n] = acc[i][r] + bv;
    }
}

// ------------------------------------------------------------------------------------------ weights: split + LDS order, once per step
// dst[nt][j = jr * NC + cc][s][plane][n][k-chunk ^ swizzle(n)][8 k]  <-  w[(ky(jr) * kw + kx(s)) * C * O + c * O + o]:
//   forward   (n, k) = (o, c): ky = jr, kx = s
//   gradient  (n, k) = (c, o): ky = ph + sh * jr, kx = S - 1 - s          (class ph; the reduction runs over o)
__global__ __launch_bounds__(256) void conv_halo_pack_kernel(const float* w, bf16_t* dst, int kw, int C, int O, int NR, int NC, int S, int NT, int grad, int ph,
                                                             int sh) {
  using T = GemmTile<0, 0, 128, 32, 4, 1>;
  const long total = (long)NT * NR * NC * S * 1024;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int k = (int)(e & 31), n = (int)((e >> 5) & 31);
    long t = e >> 10;
    const int s = (int)(t % S); t /= S;
    const int cc = (int)(t % NC); t /= NC;
    const int jr = (int)(t % NR);
    const int nt = (int)(t / NR);
    const int ky = grad ? ph + sh * jr : jr, kx = grad ? S - 1 - s : s;
    const int c = grad ? nt * 32 + n : cc * 32 + k, o = grad ? cc * 32 + k : nt * 32 + n;
    const float v = (c < C && o < O) ? w[(((long)ky * kw + kx) * C + c) * O + o] : 0.f;
    unsigned h, m, l;
    T::split3(v, h, m, l);
    bf16_t* blk = dst + (((long)nt * NR * NC + (long)jr * NC + cc) * S + s) * 3072;
    const int off = n * 32 + ((((k >> 3) ^ ((n >> 2) & 3)) << 3) | (k & 7));
    blk[off] = (bf16_t)(h >> 16);
    blk[1024 + off] = (bf16_t)(m >> 16);
    blk[2048 + off] = (bf16_t)(l >> 16);
  }
}

// ------------------------------------------------------------------------------------------ host side
struct HaloPlan {
  int ok, RG, GW, rows, NR_max, NT, NC, S;
  size_t lds, ws;
};
// which: 0 forward, 1 input gradient
static HaloPlan halo_plan(const asr_conv_desc* d, int which) {
  HaloPlan p{};
  static const int on = getenv("ASR_CONV_HALO") ? atoi(getenv("ASR_CONV_HALO")) : 1;
  // (the input gradient through this kernel measures SLOWER than conv.hip's stride-class kernels on the deepspeech shapes - 783 against 750 us
  // and 1113 against 772 us: its image carries S - 1 zero positions per row group and its staging is not overlapped yet; ASR_CONV_HALO_DX=1 routes it here)
  static const int on_dx = getenv("ASR_CONV_HALO_DX") ? atoi(getenv("ASR_CONV_HALO_DX")) : 0;
  if (!on || (which == 1 && !on_dx) || d->sw != 1 || d->kw < 2 || d->kw > 12 || d->W < d->kw || d->H < d->kh) return p;
  const int Wo = d->W - d->kw + 1;
  const int Cin = which ? d->O : d->C, N = which ? d->C : d->O, Wout = which ? d->W : Wo;
  if (Cin % 32 != 0 || N < 1 || Wout > 128) return p;
  p.S = d->kw; p.GW = Wout + d->kw - 1; p.RG = 256 / Wout; p.rows = p.RG * p.GW;
  p.NC = Cin / 32; p.NT = (N + 31) / 32;
  p.NR_max = which ? (d->kh + d->sh - 1) / d->sh : d->kh;
  p.lds = (size_t)p.rows * 192 + (size_t)p.S * 6144;
  if (p.lds > 158 * 1024 || (p.rows * 8 + 255) / 256 > 14 || (p.S * 384 + 255) / 256 > 18) return p;
  // one packed kernel per class of the gradient (sh of them), one for the forward pass
  p.ws = (size_t)(which ? d->sh : 1) * p.NT * p.NR_max * p.NC * p.S * 3072 * sizeof(bf16_t);
  p.ok = 1;
  return p;
}
extern "C" long asr_conv2d_halo_workspace(const asr_conv_desc* d, int which) {
  if (!d || which < 0 || which > 1) return 0;
  const HaloPlan p = halo_plan(d, which);
  return p.ok ? (long)p.ws : 0;
}
int conv_product_mode();        // conv.hip: 0 f32 MFMA, 2 nine pairs, 3 six pairs

template <int TERMS>
static void halo_launch(const HaloArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  auto kern = conv_halo_kernel<TERMS, 14, 18>;
  static unsigned long long seen = 0;
  if (asr_first_use_on_device(seen)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
}

// y = conv(x, w) + bias through the halo kernel.  ws: asr_conv2d_halo_workspace(d, 0) bytes (the packed kernel; rewritten by every call).
extern "C" int asr_conv2d_fwd_halo(const asr_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* ws, long ws_bytes,
                                   void* stream) {
  ASR_CHECK(d && x && w && y && ws, ASR_ERR_ARG, "asr_conv2d_fwd_halo: null argument");
  const HaloPlan p = halo_plan(d, 0);
  const int mode = conv_product_mode();
  ASR_CHECK(p.ok && mode != 0, ASR_ERR_UNSUPPORTED, "asr_conv2d_fwd_halo: this geometry / product mode takes asr_conv2d_fwd (asr_conv2d_halo_workspace returns 0)");
  ASR_CHECK(ws_bytes >= (long)p.ws, ASR_ERR_ARG, "asr_conv2d_fwd_halo: workspace of %ld bytes, %zu needed", ws_bytes, p.ws);
  ASR_CHECK((((uintptr_t)x | (uintptr_t)ws) & 15) == 0, ASR_ERR_ARG, "asr_conv2d_fwd_halo: x and the workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int Ho = (d->H - d->kh) / d->sh + 1, Wo = d->W - d->kw + 1;
  ASR_CHECK((long)d->B * d->H * d->W * d->C < 2147483647L * 4 && (long)d->B * Ho < 2147483647L, ASR_ERR_SHAPE, "asr_conv2d_fwd_halo: index space");
  const long total = (long)p.NT * d->kh * p.NC * p.S * 1024;
  hipLaunchKernelGGL(conv_halo_pack_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, st, w,
                     static_cast<bf16_t*>(ws), d->kw, d->C, d->O, d->kh, p.NC, p.S, p.NT, 0, 0, d->sh);
  HaloArgs a{};
  a.in = x; a.wp = static_cast<const bf16_t*>(ws); a.out = y; a.bias = bias;
  a.B = d->B; a.Hin = d->H; a.Win = d->W; a.Cin = d->C;
  a.Hout = Ho; a.Wout = Wo; a.N = d->O;
  a.S = p.S; a.pad = 0; a.GW = p.GW; a.RG = p.RG; a.NR = d->kh; a.NC = p.NC;
  a.h_mul = d->sh; a.r_mul = 1;
  a.Hfull = Ho; a.oh_mul = 1; a.oh_off = 0;
  a.groups = d->B * Ho;
  a.dbg = getenv("ASR_CONV_HALO_DBG") ? atoi(getenv("ASR_CONV_HALO_DBG")) : 0;
  a.dGW = asr_make_div(p.GW); a.dWout = asr_make_div(Wo); a.dHout = asr_make_div(Ho);
  dim3 grid((unsigned)asr_cdiv(a.groups, p.RG), (unsigned)p.NT);
  if (mode == 2) halo_launch<9>(a, grid, p.lds, st); else halo_launch<6>(a, grid, p.lds, st);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// dx = full correlation of dy with w (overwrites dx), one launch per stride class along H.
extern "C" int asr_conv2d_bwd_data_halo(const asr_conv_desc* d, const float* dy, const float* w, float* dx, void* ws, long ws_bytes, void* stream) {
  ASR_CHECK(d && dy && w && dx && ws, ASR_ERR_ARG, "asr_conv2d_bwd_data_halo: null argument");
  const HaloPlan p = halo_plan(d, 1);
  const int mode = conv_product_mode();
  ASR_CHECK(p.ok && mode != 0, ASR_ERR_UNSUPPORTED, "asr_conv2d_bwd_data_halo: this geometry / product mode takes asr_conv2d_bwd_data");
  ASR_CHECK(ws_bytes >= (long)p.ws, ASR_ERR_ARG, "asr_conv2d_bwd_data_halo: workspace of %ld bytes, %zu needed", ws_bytes, p.ws);
  ASR_CHECK((((uintptr_t)dy | (uintptr_t)ws) & 15) == 0, ASR_ERR_ARG, "asr_conv2d_bwd_data_halo: dy and the workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int Ho = (d->H - d->kh) / d->sh + 1, Wo = d->W - d->kw + 1;
  const size_t per_class = (size_t)p.NT * p.NR_max * p.NC * p.S * 3072;
  for (int ph = 0; ph < d->sh; ++ph) {
    const int Hq = (d->H - ph + d->sh - 1) / d->sh;
    const int nR = ph < d->kh ? (d->kh - ph + d->sh - 1) / d->sh : 0;
    if (Hq <= 0) continue;
    bf16_t* wpk = static_cast<bf16_t*>(ws) + (size_t)ph * per_class;
    if (nR > 0) {
      const long total = (long)p.NT * nR * p.NC * p.S * 1024;
      hipLaunchKernelGGL(conv_halo_pack_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, st, w, wpk, d->kw,
                         d->C, d->O, nR, p.NC, p.S, p.NT, 1, ph, d->sh);
    }
    HaloArgs a{};
    a.in = dy; a.wp = wpk; a.out = dx; a.bias = nullptr;
    a.B = d->B; a.Hin = Ho; a.Win = Wo; a.Cin = d->O;
    a.Hout = Hq; a.Wout = d->W; a.N = d->C;
    a.S = p.S; a.pad = p.S - 1; a.GW = p.GW; a.RG = p.RG; a.NR = nR; a.NC = p.NC;       // (nR = 0: no taps - the class is written as zeros)
    a.h_mul = 1; a.r_mul = -1;
    a.Hfull = d->H; a.oh_mul = d->sh; a.oh_off = ph;
    a.groups = d->B * Hq;
    a.dbg = getenv("ASR_CONV_HALO_DBG") ? atoi(getenv("ASR_CONV_HALO_DBG")) : 0;
    a.dGW = asr_make_div(p.GW); a.dWout = asr_make_div(d->W); a.dHout = asr_make_div(Hq);
    dim3 grid((unsigned)asr_cdiv(a.groups, p.RG), (unsigned)p.NT);
    if (mode == 2) halo_launch<9>(a, grid, p.lds, st); else halo_launch<6>(a, grid, p.lds, st);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
