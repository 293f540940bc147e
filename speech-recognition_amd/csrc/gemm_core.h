// f32 MFMA GEMM core for gfx950: C[M,N] (+)= alpha * op(A)[M,K] * op(B)[K,N] (+ bias), exact f32
// (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain).  The tile loaders are policies so that the
// same main loop serves plain matrices, im2col views (conv fwd / bwd-filter) and the gathered
// views of conv bwd-data.
//
// Tiling: 256 threads = 4 waves arranged WAVES_M x WAVES_N; each wave owns MI x NI tiles of 32x32;
// BK = 32.  A and B tiles are staged global -> registers -> LDS with the next tile's global
// loads in flight during the MFMAs of the current one.
//
// LDS images (so that every MFMA operand read is a conflict-free ds_read_b32):
//   operand whose K index is contiguous in memory  -> [mn][k] rows of BK+1 floats (odd stride)
//   operand whose M/N index is contiguous          -> [k][mn] rows of BM/BN floats
#pragma once
#include "common.h"

#define GEMM_BK 32

// ---- loader policies.  A loader presents a stored matrix [R][Cc] in two phases so that the main loop can
// hoist whatever does not change along K out of it:
//   Row row(r)   - everything derived from the stored row index      (address base, validity)
//   Col col(c)   - everything derived from the stored column c (a multiple of 4)
//   fetch4(row, col, v) reads stored[r][c..c+3] with zero fill outside [R, Cc)
// For an operand whose K index is the stored column (A with TA == 0, B with TB == 1) the rows are fixed per
// thread and only col() is evaluated per K tile; for the transposed storage it is the other way round.
// Divisions by run-time constants use AsrDiv (multiply-high + one correction, exact for all 32-bit x).
struct AsrDiv {
  uint32_t d, magic;   // magic = floor(2^32 / d) (d >= 2), unused for d == 1
  __device__ __forceinline__ void divmod(uint32_t x, uint32_t& q, uint32_t& r) const {
    if (d == 1) { q = x; r = 0; return; }
    q = __umulhi(x, magic);
    r = x - q * d;
    if (r >= d) { q += 1; r -= d; }
  }
};
static inline AsrDiv asr_make_div(uint32_t d) {
  AsrDiv v;
  v.d = d ? d : 1;
  v.magic = v.d > 1 ? (uint32_t)(4294967296ULL / v.d) : 0u;
  return v;
}

// FastLoader: 16-byte aligned rows, Cc % 4 == 0 - branch-free (clamped address + select).
struct FastLoader {
  const float* p;
  long ld;
  int R, Cc;
  struct Row { long off; int ok; };
  struct Col { int cc; int ok; };
  __device__ __forceinline__ void offset_z(long da, long) { p += da; }
  __device__ __forceinline__ Row row(int r) const { return Row{(long)min(r, R - 1) * ld, r < R}; }
  __device__ __forceinline__ Col col(int c) const { return Col{min(c, Cc - 4), c < Cc}; }
  __device__ __forceinline__ void fetch4(const Row& rw, const Col& cl, float (&v)[4]) const {
    const float4 t = *reinterpret_cast<const float4*>(p + rw.off + cl.cc);
    const bool ok = rw.ok && cl.ok;
    v[0] = ok ? t.x : 0.f; v[1] = ok ? t.y : 0.f; v[2] = ok ? t.z : 0.f; v[3] = ok ? t.w : 0.f;
  }
};
// FastScaledLoader: FastLoader times a row-group table scale[(r / rpg)][Cc] (Keras RNN input dropout);
// the division is a multiply-high by magic = ceil(2^32 / rpg) (exact for r * rpg < 2^32, host checked).
struct FastScaledLoader {
  const float* p;
  long ld;
  int R, Cc;
  const float* scale;
  uint32_t magic;
  struct Row { long off; long soff; int ok; };
  struct Col { int cc; int ok; };
  __device__ __forceinline__ void offset_z(long da, long ds) { p += da; scale += ds; }
  __device__ __forceinline__ Row row(int r) const {
    const int rr = min(r, R - 1);
    return Row{(long)rr * ld, (long)__umulhi((uint32_t)rr, magic) * Cc, r < R};
  }
  __device__ __forceinline__ Col col(int c) const { return Col{min(c, Cc - 4), c < Cc}; }
  __device__ __forceinline__ void fetch4(const Row& rw, const Col& cl, float (&v)[4]) const {
    const float4 t = *reinterpret_cast<const float4*>(p + rw.off + cl.cc);
    const float4 s = *reinterpret_cast<const float4*>(scale + rw.soff + cl.cc);
    const bool ok = rw.ok && cl.ok;
    v[0] = ok ? t.x * s.x : 0.f; v[1] = ok ? t.y * s.y : 0.f; v[2] = ok ? t.z * s.z : 0.f; v[3] = ok ? t.w * s.w : 0.f;
  }
};
// PlainLoader: any alignment / size, optional scale (slow path for odd shapes).
struct PlainLoader {
  __device__ __forceinline__ void offset_z(long da, long ds) { p += da; if (scale != nullptr) scale += ds; }
  const float* p;
  long ld;
  int R, Cc;       // stored rows / cols
  int vec_ok;      // base and ld 16-byte aligned
  const float* scale;  // optional group scale [R / rpg][Cc]
  int rpg;
  struct Row { int r; };
  struct Col { int c; };
  __device__ __forceinline__ Row row(int r) const { return Row{r}; }
  __device__ __forceinline__ Col col(int c) const { return Col{c}; }
  __device__ __forceinline__ void fetch4(const Row& rw, const Col& cl, float (&v)[4]) const {
    const int r = rw.r, c = cl.c;
    if (r < R && c + 3 < Cc && vec_ok) {
      const float4 t = *reinterpret_cast<const float4*>(p + (long)r * ld + c);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (r < R && c + i < Cc) ? p[(long)r * ld + c + i] : 0.f;
    }
    if (scale != nullptr && r < R) {
      const float* s = scale + (long)(r / rpg) * Cc + c;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (c + i < Cc) v[i] *= s[i];
    }
  }
};

struct GemmEpilogue {
  float* C;
  long ldc;
  int M, N;
  float alpha;
  const float* bias;     // [N] or null
  const float* c_scale;  // optional [M / c_rpg][N]
  int c_rpg;
  int mode;              // 0 store, 1 accumulate (+=), 2 atomic add
  int relu;
  // optional inverted dropout on the result, element index = r * N + c (Keras Dropout after a layer)
  const uint32_t* drop_seed;
  uint32_t drop_stream;
  float drop_rate;
  // optional row remap (conv data-gradient by stride class): logical row (b, wq, hq) -> stored row
  // (b, hq*sh + ph, wq*sw + pw) of a [B, H, W] grid
  int rm_on, rm_Hq, rm_Wq, rm_sh, rm_sw, rm_ph, rm_pw, rm_H, rm_W;
  // stored row of logical row r (evaluated once per row by the caller, not once per element)
  __device__ __forceinline__ long map_row(int r) const {
    if (!rm_on) return r;
    const int rc = min(r, M - 1);
    const int hq = rc % rm_Hq, t = rc / rm_Hq, wq = t % rm_Wq, b = t / rm_Wq;
    return ((long)b * rm_H + (long)hq * rm_sh + rm_ph) * rm_W + (long)wq * rm_sw + rm_pw;
  }
  __device__ __forceinline__ void put(int r, long rr, int c, float v) const {
    if (r >= M || c >= N) return;
    v *= alpha;
    if (bias != nullptr) v += bias[c];
    if (c_scale != nullptr) v *= c_scale[(long)(r / c_rpg) * N + c];
    if (relu) v = fmaxf(v, 0.f);
    if (drop_seed != nullptr) {
      const AsrRngKey key = asr_rng_key(drop_seed[0], drop_stream);
      v *= asr_drop_mult(key, (uint32_t)((long)r * N + c), asr_drop_threshold(drop_rate), 1.f / (1.f - drop_rate));
    }
    float* dst = C + rr * ldc + c;
    if (mode == 0) *dst = v;
    else if (mode == 1) *dst += v;
    else atomicAdd(dst, v);
  }
};

// TA: 0 -> A stored [M][K] (k contiguous); 1 -> A stored [K][M] (m contiguous). Same for TB with
// 0 -> B stored [K][N] (n contiguous); 1 -> B stored [N][K] (k contiguous).
//
// K pairing: within every group of 8 consecutive k, MFMA step e (0..3) multiplies k = 8g+e (lanes
// 0-31) and k = 8g+4+e (lanes 32-63) - any pairing is valid as long as A and B agree, and this one
// lets a k-contiguous operand be read from LDS as ONE ds_read_b128 per lane per 8 k (its 4 values
// are the lane's operands of the 4 steps).  LDS images:
//   k-contiguous operand : float4 granules [k/4][mn ^ (k/4)]  - b128 stores from the coalesced global
//                          load mapping and b128 reads, both conflict-free through the XOR swizzle
//   mn-contiguous operand: [k][mn] floats, b128 stores, b32 reads (consecutive lanes -> consecutive mn)
template <int TA, int TB, int BM, int BN, int WAVES_M, int WAVES_N, int BK_ = GEMM_BK>
struct GemmTile {
  static constexpr int BK = BK_;   // 32, or 64 for the narrow-N conv tiles (twice the MFMAs between barriers)
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  static constexpr int MI = WM / 32, NI = WN / 32;
  static constexpr int A_KC = (TA == 0);  // A has k contiguous
  static constexpr int B_KC = (TB == 1);  // B has k contiguous
  static constexpr int A_ELEMS = BM * BK;
  static constexpr int B_ELEMS = BN * BK;
  static constexpr int A_V4 = BM * BK / 4 / 256;  // float4 fetches per thread per tile
  static constexpr int B_V4 = BN * BK / 4 / 256;
  // LDS floats per operand by arithmetic mode: BF 0 / 1 keep the f32 tile; BF 2 (three-way bf16 split, below) keeps three bf16 planes
  template <int BF> static constexpr int a_lds() { return BF == 2 ? BM * 48 : A_ELEMS; }     // 3 planes x 32 k x 2 B = 192 B = 48 floats per row
  template <int BF> static constexpr int b_lds() { return BF == 2 ? BN * 48 : B_ELEMS; }
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(A_V4 >= 1 && B_V4 >= 1, "tile too small for 256 threads");
  static_assert(256 % (BM / 4) == 0 && 256 % (BN / 4) == 0 && 256 % (BK / 4) == 0, "per-thread fixed fetch column");

  // ------------------------------------------------------------------------------------------------------------------
  // BF = 2: f32 products on the bf16 matrix pipe.  gfx950 multiplies f32 on the matrix cores at 1/16 of its bf16 rate
  // (MI355X_MICROARCH.md), so an f32 product is evaluated as a sum of bf16 x bf16 products instead: every operand element is split
  // EXACTLY into three bf16 values x = x1 + x2 + x3 (8 + 8 + 8 significant bits, by truncation: each remainder is exactly
  // representable), once per element as the tile goes to LDS, and a * b = sum over the nine pairs a_i * b_j, each pair an exact
  // product inside v_mfma_f32_32x32x16_bf16 (f32 accumulation).  TERMS = 9 reproduces every bit of every product a * b (2^-32
  // relative) - more than an f32 FMA chain keeps; TERMS = 6 drops the three pairs of weight <= 2^-24 (a2 b3, a3 b2, a3 b3: error
  // <= 3 * 2^-24 |a b| per product, the size of one f32 rounding).  Accumulation is f32 in both, as in the f32 MFMA.  Cost per
  // 32 x 32 x 16 block: 9 (6) bf16 MFMAs of 8 passes against 8 f32 MFMAs of 16 passes.  Inf / NaN operands give NaN (inf - inf
  // in the split); finite values, zeros and denormals are exact.
  // LDS image: per operand three planes [mn][32 k] of bf16 (64-byte rows), the four 16-byte chunks of a row XOR-swizzled with
  // (mn >> 2) & 3 - a lane's MFMA operand (8 consecutive k) is ONE ds_read_b128 and the reads are conflict-free.
  static __device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xFFFF0000u;
    l = __float_as_uint(r1 - __uint_as_float(m));          // <= 8 significant bits: the low half of the pattern is zero
  }
  // the upper halves of two f32 patterns as one dword (v_perm_b32: bytes 2, 3 of each)
  static __device__ __forceinline__ unsigned pack2(unsigned lo_elem, unsigned hi_elem) { return __builtin_amdgcn_perm(hi_elem, lo_elem, 0x07060302u); }

  // store V4 float4 fetches of a k-contiguous operand: fetch i covers row (tid + 256 i) / 8, k = 4 ((tid + 256 i) % 8) .. + 3
  template <int V4, int BMN>
  static __device__ __forceinline__ void split_store_kc(const float (&r)[V4][4], char* base, int tid) {
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      const int f = tid + i * 256, row = f >> 3, q = f & 7;
      unsigned h[4], m[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split3(r[i][e], h[e], m[e], l[e]);
      const int off = row * 64 + ((((q >> 1) ^ ((row >> 2) & 3)) << 4) | ((q & 1) << 3));
      *reinterpret_cast<uint2*>(base + off) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
      *reinterpret_cast<uint2*>(base + BMN * 64 + off) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
      *reinterpret_cast<uint2*>(base + 2 * BMN * 64 + off) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
    }
  }
  // store V4 float4 fetches of an mn-contiguous operand: the thread holds k = kb .. kb + V4 - 1 (kb = V4 (tid / (BMN / 4))) of the four
  // rows mn = 4 (tid % (BMN / 4)) .. + 3: V4 consecutive k of one row are V4 bf16 = one 2 / 4 / 8 / 16-byte LDS store per plane
  template <int V4, int BMN>
  static __device__ __forceinline__ void split_store_mc(const float (&r)[V4][4], char* base, int tid) {
    const int kb = V4 * (tid / (BMN / 4)), mn0 = 4 * (tid % (BMN / 4));
    unsigned h[V4][4], m[V4][4], l[V4][4];
#pragma unroll
    for (int i = 0; i < V4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) split3(r[i][e], h[i][e], m[i][e], l[i][e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = mn0 + e;
      char* dst = base + row * 64 + ((((kb >> 3) ^ ((row >> 2) & 3)) << 4) | ((kb & 7) << 1));
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        char* d = dst + pl * BMN * 64;
        auto v = [&](int i) { return pl == 0 ? h[i][e] : (pl == 1 ? m[i][e] : l[i][e]); };
        if constexpr (V4 == 1) *reinterpret_cast<unsigned short*>(d) = (unsigned short)(v(0) >> 16);
        else if constexpr (V4 == 2) *reinterpret_cast<unsigned*>(d) = pack2(v(0), v(1));
        else if constexpr (V4 == 4) *reinterpret_cast<uint2*>(d) = make_uint2(pack2(v(0), v(1)), pack2(v(2), v(3)));
        else *reinterpret_cast<uint4*>(d) = make_uint4(pack2(v(0), v(1)), pack2(v(2), v(3)), pack2(v(4), v(5)), pack2(v(6), v(7)));
      }
    }
  }

  template <int TERMS, class AL, class BL>
  static __device__ __forceinline__ void run_split(const AL& al, const BL& bl, const GemmEpilogue& ep, int kbeg, int kend,
                                                   int m0, int n0, float* As, float* Bs) {
    static_assert(BK == 32, "the split path stages 32-deep K tiles");
    static_assert(A_V4 == 1 || A_V4 == 2 || A_V4 == 4 || A_V4 == 8, "fetches per thread");
    static_assert(B_V4 == 1 || B_V4 == 2 || B_V4 == 4 || B_V4 == 8, "fetches per thread");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    float ra[A_V4][4], rb[B_V4][4];
    // a wave with ONE output tile (64 x 64 workgroup tile) has no independent MFMA to put between two dependent ones: its two 16-deep
    // steps of a K tile then accumulate into two accumulators (16 more registers), added at the end
    constexpr bool DUAL = MI * NI == 1;
    f32x16 acc[MI][NI], acc2[1][1];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[0][0][r] = 0.f;
    char* Ab = reinterpret_cast<char*>(As);
    char* Bb = reinterpret_cast<char*>(Bs);
    typename AL::Row arow[A_KC ? A_V4 : 1];
    typename BL::Row brow[B_KC ? B_V4 : 1];
    const typename AL::Col acol = al.col(m0 + 4 * (tid % (BM / 4)));
    const typename BL::Col bcol = bl.col(n0 + 4 * (tid % (BN / 4)));
    if (A_KC) {
#pragma unroll
      for (int i = 0; i < A_V4; ++i) arow[i] = al.row(m0 + (tid + i * 256) / (BK / 4));
    }
    if (B_KC) {
#pragma unroll
      for (int i = 0; i < B_V4; ++i) brow[i] = bl.row(n0 + (tid + i * 256) / (BK / 4));
    }
    auto gload = [&](int k0) {
      if (A_KC) {
        const typename AL::Col kc = al.col(k0 + 4 * (tid % (BK / 4)));
#pragma unroll
        for (int i = 0; i < A_V4; ++i) al.fetch4(arow[i], kc, ra[i]);
      } else {                                             // (k rows A_V4 (tid / (BM / 4)) + i: consecutive k per thread, see split_store_mc)
#pragma unroll
        for (int i = 0; i < A_V4; ++i) al.fetch4(al.row(k0 + A_V4 * (tid / (BM / 4)) + i), acol, ra[i]);
      }
      if (B_KC) {
        const typename BL::Col kc = bl.col(k0 + 4 * (tid % (BK / 4)));
#pragma unroll
        for (int i = 0; i < B_V4; ++i) bl.fetch4(brow[i], kc, rb[i]);
      } else {
#pragma unroll
        for (int i = 0; i < B_V4; ++i) bl.fetch4(bl.row(k0 + B_V4 * (tid / (BN / 4)) + i), bcol, rb[i]);
      }
    };
    auto lstore = [&]() {
      if (A_KC) split_store_kc<A_V4, BM>(ra, Ab, tid); else split_store_mc<A_V4, BM>(ra, Ab, tid);
      if (B_KC) split_store_kc<B_V4, BN>(rb, Bb, tid); else split_store_mc<B_V4, BN>(rb, Bb, tid);
    };
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int l31 = lane & 31, lh = lane >> 5;
    gload(kbeg);
    for (int kt = 0; kt < nk; ++kt) {
      lstore();
      __syncthreads();
      if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {                     // two 16-deep MFMA steps per K tile: chunk 2 s2 + lh of the row
        bf16x8 a8[3][MI], b8[3][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int m = wm * WM + i * 32 + l31;
          const int off = m * 64 + ((((2 * s2 + lh) ^ ((m >> 2) & 3))) << 4);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) a8[pl][i] = *reinterpret_cast<const bf16x8*>(Ab + pl * BM * 64 + off);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int n = wn * WN + j * 32 + l31;
          const int off = n * 64 + ((((2 * s2 + lh) ^ ((n >> 2) & 3))) << 4);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) b8[pl][j] = *reinterpret_cast<const bf16x8*>(Bb + pl * BN * 64 + off);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            // pairs by weight: (1,1) 1, (1,2) (2,1) 2^-8, (1,3) (2,2) (3,1) 2^-16 | (2,3) (3,2) 2^-24, (3,3) 2^-32
            // (one accumulator: a second one for the low-order pairs would double the accumulator registers and cost the 128 x 128
            // tile its occupancy; the pairs go in by rising weight inside each 16-deep block instead)
            f32x16 c = (DUAL && s2 == 1) ? acc2[0][0] : acc[i][j];
            if (TERMS == 9) {
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[2][i], b8[2][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[1][i], b8[2][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[2][i], b8[1][j], c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][i], b8[2][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[1][i], b8[1][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[2][i], b8[0][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][i], b8[1][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[1][i], b8[0][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][i], b8[0][j], c, 0, 0, 0);
            if (DUAL && s2 == 1) acc2[0][0] = c; else acc[i][j] = c;
          }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const long srow = ep.map_row(row);
#pragma unroll
        for (int j = 0; j < NI; ++j) ep.put(row, srow, n0 + wn * WN + j * 32 + l31, DUAL ? acc[i][j][r] + acc2[0][0][r] : acc[i][j][r]);
      }
  }

  // BF = 1: the operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) as they leave LDS and the
  // products run on v_mfma_f32_32x32x16_bf16 with the same f32 accumulators - "mixed precision" with f32 storage,
  // f32 accumulation and f32 epilogue.  One bf16 MFMA takes k = 16s + 8*(lane>>5) + j (j = 0..7) of the K tile.
  template <int BF, class AL, class BL>
  static __device__ __forceinline__ void run(const AL& al, const BL& bl, const GemmEpilogue& ep, int kbeg, int kend,
                                             int m0, int n0, float* As, float* Bs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    float ra[A_V4][4], rb[B_V4][4];
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4* As4 = reinterpret_cast<float4*>(As);
    float4* Bs4 = reinterpret_cast<float4*>(Bs);

    // the index that is fixed per thread along K (stored row for a k-contiguous operand, stored column
    // otherwise) is decoded once; the other one once per K tile
    typename AL::Row arow[A_KC ? A_V4 : 1];
    typename BL::Row brow[B_KC ? B_V4 : 1];
    // (256 threads and BM/4, BN/4, BK/4 all divide 256: the fixed column of a thread is the same for all its fetches)
    const typename AL::Col acol = al.col(m0 + 4 * (tid % (BM / 4)));
    const typename BL::Col bcol = bl.col(n0 + 4 * (tid % (BN / 4)));
    if (A_KC) {
#pragma unroll
      for (int i = 0; i < A_V4; ++i) arow[i] = al.row(m0 + (tid + i * 256) / (BK / 4));
    }
    if (B_KC) {
#pragma unroll
      for (int i = 0; i < B_V4; ++i) brow[i] = bl.row(n0 + (tid + i * 256) / (BK / 4));
    }
    auto gload = [&](int k0) {
      if (A_KC) {
        const typename AL::Col kc = al.col(k0 + 4 * (tid % (BK / 4)));
#pragma unroll
        for (int i = 0; i < A_V4; ++i) al.fetch4(arow[i], kc, ra[i]);
      } else {
#pragma unroll
        for (int i = 0; i < A_V4; ++i) al.fetch4(al.row(k0 + (tid + i * 256) / (BM / 4)), acol, ra[i]);
      }
      if (B_KC) {
        const typename BL::Col kc = bl.col(k0 + 4 * (tid % (BK / 4)));
#pragma unroll
        for (int i = 0; i < B_V4; ++i) bl.fetch4(brow[i], kc, rb[i]);
      } else {
#pragma unroll
        for (int i = 0; i < B_V4; ++i) bl.fetch4(bl.row(k0 + (tid + i * 256) / (BN / 4)), bcol, rb[i]);
      }
    };
    auto lstore = [&]() {
#pragma unroll
      for (int i = 0; i < A_V4; ++i) {
        const int f = tid + i * 256;
        const float4 v = make_float4(ra[i][0], ra[i][1], ra[i][2], ra[i][3]);
        if (A_KC) { const int r = f / (BK / 4), q = f % (BK / 4); As4[q * BM + (r ^ q)] = v; }
        else      { const int r = f / (BM / 4), q = f % (BM / 4); As4[r * (BM / 4) + q] = v; }
      }
#pragma unroll
      for (int i = 0; i < B_V4; ++i) {
        const int f = tid + i * 256;
        const float4 v = make_float4(rb[i][0], rb[i][1], rb[i][2], rb[i][3]);
        if (B_KC) { const int r = f / (BK / 4), q = f % (BK / 4); Bs4[q * BN + (r ^ q)] = v; }
        else      { const int r = f / (BN / 4), q = f % (BN / 4); Bs4[r * (BN / 4) + q] = v; }
      }
    };

    // K range [kbeg, kend): kbeg is a multiple of BK (split-K partitions are BK aligned); the loaders
    // zero-fill past their own extent
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int l31 = lane & 31, lh = lane >> 5;
    gload(kbeg);
    for (int kt = 0; kt < nk; ++kt) {
      lstore();
      __syncthreads();
      if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
      if (BF) {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
          bf16x8 a8[MI], b8[NI];
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int m = wm * WM + i * 32 + l31;
            float t[8];
            if (A_KC) {
              const int q = 4 * s + 2 * lh;
              const float4 u = As4[q * BM + (m ^ q)], w = As4[(q + 1) * BM + (m ^ (q + 1))];
              t[0] = u.x; t[1] = u.y; t[2] = u.z; t[3] = u.w; t[4] = w.x; t[5] = w.y; t[6] = w.z; t[7] = w.w;
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) t[e] = As[(16 * s + 8 * lh + e) * BM + m];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) a8[i][e] = (__bf16)t[e];
          }
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int n = wn * WN + j * 32 + l31;
            float t[8];
            if (B_KC) {
              const int q = 4 * s + 2 * lh;
              const float4 u = Bs4[q * BN + (n ^ q)], w = Bs4[(q + 1) * BN + (n ^ (q + 1))];
              t[0] = u.x; t[1] = u.y; t[2] = u.z; t[3] = u.w; t[4] = w.x; t[5] = w.y; t[6] = w.z; t[7] = w.w;
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) t[e] = Bs[(16 * s + 8 * lh + e) * BN + n];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) b8[j][e] = (__bf16)t[e];
          }
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
        }
      } else {
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        float a[MI][4], b[NI][4];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int m = wm * WM + i * 32 + l31;
          if (A_KC) {
            const int q = 2 * g + lh;
            const float4 t = As4[q * BM + (m ^ q)];
            a[i][0] = t.x; a[i][1] = t.y; a[i][2] = t.z; a[i][3] = t.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[i][e] = As[(8 * g + 4 * lh + e) * BM + m];
          }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int n = wn * WN + j * 32 + l31;
          if (B_KC) {
            const int q = 2 * g + lh;
            const float4 t = Bs4[q * BN + (n ^ q)];
            b[j][0] = t.x; b[j][1] = t.y; b[j][2] = t.z; b[j][3] = t.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) b[j][e] = Bs[(8 * g + 4 * lh + e) * BN + n];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
      }
      }
      __syncthreads();
    }
    // C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const long srow = ep.map_row(row);
#pragma unroll
        for (int j = 0; j < NI; ++j) ep.put(row, srow, n0 + wn * WN + j * 32 + l31, acc[i][j][r]);
      }
  }
};
