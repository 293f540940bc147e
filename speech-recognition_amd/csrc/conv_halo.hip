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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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

#define HALO_MAX_CLASSES 4
struct HaloArgsSet { HaloArgs c[HALO_MAX_CLASSES]; };          // blockIdx.z: the stride classes of an input gradient share one launch

__device__ __forceinline__ int halo_a_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// TERMS = 6 / 9 (gemm_core.h run_split).  256 threads: wave v owns output rows 64 v .. 64 v + 63 of the tile (two 32 x 32 MFMA tiles) x 32 channels.
template <int TERMS, int MAXF, int MAXB>
__global__ __launch_bounds__(256) void conv_halo_kernel(HaloArgsSet as) {
  extern __shared__ __attribute__((aligned(16))) char halo_lds[];
  const HaloArgs a = as.c[blockIdx.z];                          // (by value: one scalar load per field, none inside the loops)
  if ((int)blockIdx.x * a.RG >= a.groups) return;
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

// ------------------------------------------------------------------------------------------ second version: staging under the products
// The kernel above stages an image (loads, split, LDS stores, two barriers) and THEN multiplies: with one workgroup per CU nothing else runs
// while it stages - 4.5 of the 9.4 us a (kernel row, chunk) costs on the deepspeech shapes (ASR_CONV_HALO_DBG).  Here
//   * the input image has TWO buffers: the loads for (r, chunk) j + 1 go out when j starts, and their split + LDS stores are spread over the taps
//     of j, two fetches per tap from the fourth tap on, between that tap's MFMAs (the vector instructions fill the cycles in which a matrix
//     instruction leaves the issue free);
//   * the weights stream through a ring of three tap blocks (6 KB each): the block of tap t + 2 is loaded when tap t starts and stored when it
//     ends, so it is visible one barrier before it is needed and the fragments of tap t + 1 can be read ahead;
//   * one barrier per tap (24 MFMAs per wave).
template <int I, int N, class F>
__device__ __forceinline__ void halo_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    halo_static_for<I + 1, N>(f);
  }
}
// SC: the number of taps, at compile time - with the peeled last (r, chunk) the steady state is straight-line code, which is what lets the compiler
// wait for the loads it needs by COUNT (behind any branch it waits for all of them, the weight block just requested included: measured, no gain)
#define HALO_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// wait until at most N loads are outstanding; the registers named are its outputs as far as the compiler knows, so nothing that reads them moves above it
template <int N>
__device__ __forceinline__ void halo_wait(u32x4& x, u32x4& y) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N) : "memory"); }
__device__ __forceinline__ void halo_touch(f32x4& x) { asm volatile("" : "+v"(x)); }
template <class V>
__device__ __forceinline__ void halo_gload(V& dst, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
template <int TERMS, int MAXF, int SC>
__global__ __launch_bounds__(256) void conv_halo2_kernel(HaloArgsSet as) {
  extern __shared__ __attribute__((aligned(16))) char halo_lds[];
  const HaloArgs a = as.c[blockIdx.z];                          // (by value: one scalar load per field, none inside the loops)
  if ((int)blockIdx.x * a.RG >= a.groups) return;
  using T = GemmTile<0, 0, 128, 32, 4, 1>;                       // (split3 / pack2 only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int rows = a.RG * a.GW;
  // (one spare row per plane and 2 KB behind the ring take the stores of the lanes that have nothing to store: no branch around a store or a load
  // anywhere in the steady state - behind a branch the compiler waits for ALL outstanding loads instead of counting)
  const int a_plane = (rows + 1) * 64, a_buf = 3 * a_plane;
  char* ring = halo_lds + 2 * a_buf;
  char* dummy_b = ring + 3 * 6144;
  const int g0 = blockIdx.x * a.RG, n0 = blockIdx.y * 32;
  constexpr int S = SC;
  const int J = a.NR * a.NC, total = J * S;
  const uint4* wp = reinterpret_cast<const uint4*>(a.wp + (long)blockIdx.y * total * 3072);   // 384 16-byte chunks per tap
  const int nfa = (rows * 8 + 255) >> 8;
  const long row_stride = (long)a.Win * a.Cin;
  // what a fetch needs that does not change with (r, chunk)
  long abase[MAXF];
  int ahh[MAXF], aoff[MAXF];
  bool aok[MAXF], arow[MAXF];
#pragma unroll
  for (int f = 0; f < MAXF; ++f) {
    const int idx = tid + 256 * f, row = idx >> 3, q = idx & 7;
    uint32_t g, u, b, h;
    a.dGW.divmod((uint32_t)min(row, rows - 1), g, u);
    a.dHout.divmod((uint32_t)min(g0 + (int)g, a.groups - 1), b, h);
    const int w = (int)u - a.pad;
    arow[f] = f < nfa && row < rows;
    aok[f] = arow[f] && g0 + (int)g < a.groups && w >= 0 && w < a.Win;
    abase[f] = ((long)b * a.Hin * a.Win + max(0, min(w, a.Win - 1))) * a.Cin + 4 * q;
    ahh[f] = (int)h * a.h_mul;
    aoff[f] = arow[f] ? (halo_a_off(row, q >> 1) | ((q & 1) << 3)) : rows * 64 + 8 * q;
  }
  // The steady state's loads are inline assembly and its waits are written out (HALO_WAIT): hipcc branches around a load whose result is
  // selected away and, behind any branch, waits for ALL outstanding loads - the weight block just requested included (measured: 724 us against
  // 576).  Issue order per image: tap 0: weight block (2 loads), image fetches (MAXF loads); taps 1 ..: weight block (2).  vmcnt counts in order.
  f32x4 ra[MAXF];
  bool aval[MAXF];
  u32x4 rb[2][2];                                               // two weight blocks in flight: [tap parity][chunk]
  auto a_load = [&](int j) {
    const int jr = j / a.NC, cc = j - jr * a.NC;
#pragma unroll
    for (int f = 0; f < MAXF; ++f) {
      const int hin = ahh[f] + jr * a.r_mul;
      aval[f] = aok[f] && hin >= 0 && hin < a.Hin;
      const float* p = a.in + (aval[f] ? abase[f] + (long)hin * row_stride + cc * 32 : 0L);       // (a lane with nothing to load reads x[0..3])
      halo_gload(ra[f], p);
    }
  };
  auto a_store = [&](auto fc, char* Ad) {
    constexpr int f = decltype(fc)::value;
    if constexpr (f < MAXF) {
      {
        unsigned h[4], m[4], l[4];
        const f32x4 v = aval[f] ? ra[f] : (f32x4){0.f, 0.f, 0.f, 0.f};
        T::split3(v.x, h[0], m[0], l[0]); T::split3(v.y, h[1], m[1], l[1]);
        T::split3(v.z, h[2], m[2], l[2]); T::split3(v.w, h[3], m[3], l[3]);
        *reinterpret_cast<uint2*>(Ad + aoff[f]) = make_uint2(T::pack2(h[0], h[1]), T::pack2(h[2], h[3]));
        *reinterpret_cast<uint2*>(Ad + a_plane + aoff[f]) = make_uint2(T::pack2(m[0], m[1]), T::pack2(m[2], m[3]));
        *reinterpret_cast<uint2*>(Ad + 2 * a_plane + aoff[f]) = make_uint2(T::pack2(l[0], l[1]), T::pack2(l[2], l[3]));
      }
    }
  };
  auto b_load = [&](int t, auto set) {
    constexpr int R = decltype(set)::value;
    const uint4* src = wp + (long)t * 384;
    const uint4* p0 = src + tid;
    const uint4* p1 = src + (tid < 128 ? 256 + tid : tid);
    halo_gload(rb[R][0], p0);
    halo_gload(rb[R][1], p1);
  };
  auto b_store = [&](int slot, auto set) {
    constexpr int R = decltype(set)::value;
    char* dst = ring + slot * 6144;
    *reinterpret_cast<u32x4*>(dst + tid * 16) = rb[R][0];
    *reinterpret_cast<u32x4*>(tid < 128 ? dst + (256 + tid) * 16 : dummy_b + (tid - 128) * 16) = rb[R][1];
  };
  using R0 = std::integral_constant<int, 0>;
  using R1 = std::integral_constant<int, 1>;
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
  const int boff = l31 * 64, bsw = (l31 >> 2) & 3;
  bf16x8 fa0[3][2], fb0[3], fa1[3][2], fb1[3];
  auto rd = [&](const char* Ab, int slot, int s, int kk, bf16x8 (&fa)[3][2], bf16x8 (&fb)[3]) {
    const char* Bs = ring + slot * 6144 + boff + (((2 * kk + lh) ^ bsw) << 4);
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

  if (total > 0) {
    // prologue: image 0 and the weight blocks of taps 0 and 1
    a_load(0);
    b_load(0, R0{}); HALO_WAIT_ALL(); b_store(0, R0{});
    if (total > 1) { b_load(1, R0{}); HALO_WAIT_ALL(); b_store(1, R0{}); }
    HALO_WAIT_ALL();
    halo_static_for<0, MAXF>([&](auto fc) { a_store(fc, halo_lds); });
    if (total > 2) b_load(2, R1{});                             // (tap 0 stores the set of the odd taps)
    __syncthreads();
    int slot = 0;                                               // ring slot of the current tap
    rd(halo_lds, 0, 0, 0, fa0, fb0);
    // PAR: parity of the image's first tap (the register set of a weight block is the parity of the tap that requested it)
    auto image = [&](int j, auto more_c, auto par_c) {
      constexpr bool more = decltype(more_c)::value;            // another image follows (steady state)
      constexpr int PAR = decltype(par_c)::value;
      const char* Ab = halo_lds + (j & 1) * a_buf;
      char* An = halo_lds + ((j + 1) & 1) * a_buf;
      halo_static_for<0, S>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        // the weight block of tap t + 3 is requested here (register set s & 1) and the one of tap t + 2, requested a tap ago (the other set),
        // goes into the ring at the end of this tap: a block has a whole tap (24 MFMAs per wave) and more to arrive
        constexpr bool wnext = more || s + 2 < S;               // tap t + 2 exists
        constexpr bool wload = more || s + 3 < S;               // tap t + 3 exists ...
        const int t = j * S + s;
        if constexpr (wload) b_load(t + 3, std::integral_constant<int, ((s + PAR) & 1)>{});
        if constexpr (more && s == 0) a_load(j + 1);            // (behind the weight block: its store at the end of this tap then waits by count, not for these)
        rd(Ab, slot, s, 1, fa1, fb1);
        mm(fa0, fb0);
        // (every image fetch is older than the weight block stored at the end of tap 1: from tap 2 on they have all arrived)
        if constexpr (s >= 3 && more) a_store(std::integral_constant<int, 2 * (s - 3)>{}, An);
        if constexpr (s + 1 < S) rd(Ab, slot == 2 ? 0 : slot + 1, s + 1, 0, fa0, fb0);   // (the next tap's weights have been visible since the last barrier)
        mm(fa1, fb1);
        if constexpr (s >= 3 && more) a_store(std::integral_constant<int, 2 * (s - 3) + 1>{}, An);
        if constexpr (s + 1 == S && more) {
          // the fetches the taps did not get to (all of them when S < 4)
          halo_static_for<(S >= 4 ? 2 * (S - 3) : 0), MAXF>([&](auto fc) { a_store(fc, An); });
        }
        if constexpr (wnext) {
          // the block requested a tap ago: younger than it are this tap's block (2 loads, if requested) and, in tap 0, the image fetches
          constexpr int younger = (wload ? 2 : 0) + ((more && s == 0) ? MAXF : 0);
          halo_wait<younger>(rb[(s + 1 + PAR) & 1][0], rb[(s + 1 + PAR) & 1][1]);
          if constexpr (more && s == 1) halo_static_for<0, MAXF>([&](auto fc) { halo_touch(ra[decltype(fc)::value]); });
        }
        if constexpr (wnext) b_store(slot == 0 ? 2 : slot - 1, std::integral_constant<int, ((s + 1 + PAR) & 1)>{});   // slot of tap t + 2 = (slot + 2) % 3
        __syncthreads();
        slot = slot == 2 ? 0 : slot + 1;
        if constexpr (s + 1 == S && more) rd(An, slot, 0, 0, fa0, fb0);   // first fragments of the next image (complete behind this barrier)
      });
    };
    int j = 0;
    for (; j + 2 < J; j += 2) {
      image(j, std::true_type{}, R0{});
      image(j + 1, std::true_type{}, std::integral_constant<int, (S & 1)>{});
    }
    if (j + 1 < J) { image(j, std::true_type{}, R0{}); ++j; if constexpr (S & 1) image(j, std::false_type{}, R1{}); else image(j, std::false_type{}, R0{}); }
    else image(j, std::false_type{}, R0{});
  }
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
  int ok, ver, RG, GW, rows, NR_max, NT, NC, S;
  size_t lds, ws;
};
// which: 0 forward, 1 input gradient
static int g_halo_force = -1;
static int halo_force() {
  if (g_halo_force < 0) g_halo_force = getenv("ASR_CONV_HALO_FORCE") ? atoi(getenv("ASR_CONV_HALO_FORCE")) : 0;
  return g_halo_force;
}
// tests / tuning: 1 routes every eligible geometry through the row-staged kernels whatever the fill of its last round of workgroups; returns the previous setting
extern "C" int asr_conv2d_halo_force(int on) {
  const int old = halo_force();
  if (on >= 0) g_halo_force = on ? 1 : 0;
  return old;
}
static HaloPlan halo_plan(const asr_conv_desc* d, int which) {
  HaloPlan p{};
  static const int on = getenv("ASR_CONV_HALO") ? atoi(getenv("ASR_CONV_HALO")) : 1;
  static const int on_dx = getenv("ASR_CONV_HALO_DX") ? atoi(getenv("ASR_CONV_HALO_DX")) : 1;
  const int force = halo_force();                                // tests / tuning: skip the occupancy gate below
  if (!on || (which == 1 && !on_dx) || d->sw != 1 || d->kw < 2 || d->kw > 12 || d->W < d->kw || d->H < d->kh || d->sh > HALO_MAX_CLASSES) return p;
  const int Wo = d->W - d->kw + 1, Ho = (d->H - d->kh) / d->sh + 1;
  const int Cin = which ? d->O : d->C, N = which ? d->C : d->O, Wout = which ? d->W : Wo;
  if (Cin % 32 != 0 || N < 1 || Wout > 128) return p;
  static const int ver_env = getenv("ASR_CONV_HALO_V") ? atoi(getenv("ASR_CONV_HALO_V")) : 0;
  p.S = d->kw; p.GW = Wout + d->kw - 1;
  p.NC = Cin / 32; p.NT = (N + 31) / 32;
  p.NR_max = which ? (d->kh + d->sh - 1) / d->sh : d->kh;
  // row groups in all: the classes of an input gradient run in one launch
  long groups = 0;
  if (which) { for (int ph = 0; ph < d->sh; ++ph) groups += (long)d->B * ((d->H - ph + d->sh - 1) / d->sh); }
  else groups = (long)d->B * Ho;
  // version 1: one image buffer + the S tap blocks of a kernel row; version 2 (compiled for the 11 taps of the deepspeech kernels): two image
  // buffers + a ring of three tap blocks, the staging under the products - 0.85 of version 1's time per workgroup, but fewer row groups fit
  const int rg1 = 256 / Wout;
  int rg2 = (160 * 1024 - 3 * 6144 - 2048 - 384) / (384 * p.GW);
  if (rg2 > rg1) rg2 = rg1;
  auto rounds = [&](int rg) { return rg < 1 ? 1L << 40 : (((groups + rg - 1) / rg + d->sh * (which ? 1 : 0)) * p.NT + 255) / 256; };
  const bool can2 = d->kw == 11 && rg2 >= 1 && (rg2 * p.GW * 8 + 255) / 256 <= 14;
  const bool can1 = (size_t)rg1 * p.GW * 192 + (size_t)p.S * 6144 <= 158 * 1024 && (rg1 * p.GW * 8 + 255) / 256 <= 14 && (p.S * 384 + 255) / 256 <= 18;
  // Version 2 is NOT taken by default any more (end of round 4): with kernels of another stream running beside it - the arrangement of
  // tests/tools/exp/ds2_beside_dbg.py, conv2's input gradient beside two filter-gradient kernels - single workgroups of it produced garbage
  // (5 600-36 400 of 13 M elements around 1e26, in 3 of 6 trials; version 1 and the general kernel: 0 of 12 and 0 of 9 in the same arrangement;
  // alone it has never differed from the float64 reference).  Its hand-counted waits and LDS hand-overs look right on paper and the cause is not
  // found, so a training step does not depend on it: ASR_CONV_HALO_V=2 asks for it explicitly, and the parity tests (asr_conv2d_halo_force)
  // keep exercising it.  Cost: deepspeech conv2 forward 488 -> ~600 us, input gradient 589 -> ~650 us.
  if (ver_env == 1) p.ver = can1 ? 1 : 0;
  else if (ver_env == 2 || (ver_env == 0 && force)) p.ver = (can2 && (ver_env == 2 || !can1 || 0.85 * rounds(rg2) <= 1.0 * rounds(rg1))) ? 2 : (can1 ? 1 : 0);
  else p.ver = can1 ? 1 : 0;
  if (!p.ver) return p;
  p.RG = p.ver == 2 ? rg2 : rg1;
  p.rows = p.RG * p.GW;
  p.lds = p.ver == 2 ? (size_t)(p.rows + 1) * 384 + 3 * 6144 + 2048 : (size_t)p.rows * 192 + (size_t)p.S * 6144;
  // One workgroup per CU, every workgroup the same length: a launch pays for whole rounds of 256.  The general kernels (conv.hip) run several
  // smaller workgroups per CU and lose little to the last round, so this path is taken only where its rounds are mostly full (deepspeech conv3's
  // input gradient makes 2.2 rounds of long workgroups here: 780 us against 772 there; conv2's 6.5: 590 against 750).
  const long wgs = ((groups + p.RG - 1) / p.RG) * p.NT;
  const double fill = (double)wgs / (double)(((wgs + 255) / 256) * 256);
  if (!force && fill < (which ? 0.85 : 0.70)) return p;
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
static void halo_launch(const HaloArgsSet& a, dim3 grid, size_t lds, int ver, hipStream_t st) {
  if (ver == 2) {
    auto kern = conv_halo2_kernel<TERMS, 14, 11>;
    static unsigned long long seen = 0;
    if (asr_first_use_on_device(seen)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    return;
  }
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
  dim3 grid((unsigned)asr_cdiv(a.groups, p.RG), (unsigned)p.NT, 1);
  HaloArgsSet set{};
  set.c[0] = a;
  if (mode == 2) halo_launch<9>(set, grid, p.lds, p.ver, st); else halo_launch<6>(set, grid, p.lds, p.ver, st);
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
  HaloArgsSet set{};
  int ncls = 0, max_tiles = 0;
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
    HaloArgs& a = set.c[ncls++];
    a = HaloArgs{};
    a.in = dy; a.wp = wpk; a.out = dx; a.bias = nullptr;
    a.B = d->B; a.Hin = Ho; a.Win = Wo; a.Cin = d->O;
    a.Hout = Hq; a.Wout = d->W; a.N = d->C;
    a.S = p.S; a.pad = p.S - 1; a.GW = p.GW; a.RG = p.RG; a.NR = nR; a.NC = p.NC;       // (nR = 0: no taps - the class is written as zeros)
    a.h_mul = 1; a.r_mul = -1;
    a.Hfull = d->H; a.oh_mul = d->sh; a.oh_off = ph;
    a.groups = d->B * Hq;
    a.dbg = getenv("ASR_CONV_HALO_DBG") ? atoi(getenv("ASR_CONV_HALO_DBG")) : 0;
    a.dGW = asr_make_div(p.GW); a.dWout = asr_make_div(d->W); a.dHout = asr_make_div(Hq);
    const int tiles = asr_cdiv(a.groups, p.RG);
    max_tiles = tiles > max_tiles ? tiles : max_tiles;
  }
  if (ncls > 0) {
    dim3 grid((unsigned)max_tiles, (unsigned)p.NT, (unsigned)ncls);
    if (mode == 2) halo_launch<9>(set, grid, p.lds, p.ver, st); else halo_launch<6>(set, grid, p.lds, p.ver, st);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
