// asr_gemm_bf16_nt: C[M,N] (+)= alpha * A[M,K] * B[N,K]^T (+ bias) with bf16 OPERANDS IN MEMORY, f32 accumulation and an f32 C -
// the dense contractions of the wide models under --mixed-precision (run/train.py:62-66; BASELINE configs[4] las_large).
//
// Why a second GEMM: gemm_core.h's BF = 1 mode keeps f32 operands in memory and LDS and rounds them to bf16 as fragments leave
// LDS - right for models whose activations must stay f32 for the f32 kernels around the product, but the staging then moves
// four bytes per operand element through global -> register -> LDS -> register for a two-byte MFMA operand: 300-420 TFLOP/s on
// the las_large shapes against a 2.5 PFLOP/s pipe.  Here both operands arrive as bf16 images, k-contiguous
// ("NT": A rows and B rows both run along K), so that a lane's 16-byte LDS read IS its MFMA operand (8 consecutive k):
//   v_mfma_f32_32x32x16_bf16: A lane l = (row l & 31, k 8 (l >> 5) .. + 7), B likewise with the column; C col = l & 31,
//   row = (r & 3) + 8 (r >> 2) + 4 (l >> 5).
// Default schedule (configuration 12, K a multiple of 64 with at least three K tiles): tile 256 x 128 x 64, 8 waves as 4 x 2, each
// 64 x 64 = 2 x 2 MFMA tiles; the tiles go global -> LDS DIRECTLY (global_load_lds_dwordx4, the swizzle applied to the per-lane
// source chunk) into THREE buffers, two tiles ahead, with a counted s_waitcnt and a raw s_barrier so that the loads stay in
// flight across the barrier (one barrier per K tile): 913 TF on the weight-gradient shape against 789 for the fallback.
// Fallback (any K multiple of 8; configuration 0): tile 128 x 128 x 64, 4 waves as 2 x 2, global -> registers -> LDS staging with
// the next tile's loads in flight during the MFMAs (as gemm_core.h).  LDS rows of 64 bf16 = 128 B, the 16-byte chunk index XORed
// with ((row >> 1) & 7) so that each 16-lane group of a ds_read_b128 covers the bank row exactly once (g16_off).
// The images are produced by memory-bound conversion passes (f32 -> bf16, optionally times the Keras input-dropout row-group
// table, optionally TRANSPOSED: the weight-gradient products contract over the rows of both activations): ~0.1 ms per
// 31936 x 2048 activation against ~1-3 ms of product saved.  Epilogue = gemm_core.h's (bias, ReLU, row-group scale, +=, atomics).
#include "gemm16.h"

// WM x WN waves, each TM x TN MFMA tiles of 32 x 32: block tile (32 WM TM) x (32 WN TN) x 64.  NBUF LDS buffers: with 2 the tile for step
// kt + 1 is written into the other buffer BEFORE the MFMAs of step kt and the loads of tile kt + 2 are issued right after (one barrier per
// K step, loads in flight for a whole MFMA phase); with 1 the write waits behind the MFMAs (two barriers, half the LDS: more blocks per CU).
// STG 1: the tiles go global -> LDS directly (global_load_lds_dwordx4: a wave instruction fills 8 tile rows = 1024 contiguous bytes, the XOR
// swizzle applied to the per-lane SOURCE chunk) - no staging registers, no ds_write pass; needs whole K tiles (K, k_chunk multiples of 64) and
// clamps out-of-range rows to the last valid one (the epilogue drops them).
template <int WM, int WN, int TM, int TN, int NBUF, int STG>
__global__ __launch_bounds__(64 * WM * WN) void gemm16_nt_kernel(const bf16_t* A, long lda, const bf16_t* B, long ldb, GemmEpilogue ep, int M, int N, int K,
                                                                 int tiles_m, int tiles_n, int split_k, int k_chunk, long sAz, long sBz, long sCz) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, NT = 64 * WM * WN;
  constexpr int CA = BM * 8 / NT, CB = BN * 8 / NT;             // 16-byte chunks per thread and K tile
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "staging map");
  extern __shared__ __attribute__((aligned(16))) unsigned char g16_smem[];
  const int z = blockIdx.z / split_k, zs = blockIdx.z % split_k;
  A += (long)z * sAz; B += (long)z * sBz; ep.C += (long)z * sCz;
  if (zs != 0 || (z != 0 && sCz == 0)) ep.bias = nullptr;
  const int kbeg = zs * k_chunk, kend = min(K, kbeg + k_chunk);
  if (kbeg >= K) return;
  // XCD-aware tile order (gemm.hip): blocks b, b + 8 share an XCD; every XCD walks a contiguous range, the smaller operand fastest
  int bm, bn;
  g16_tile(blockIdx.x, tiles_m, tiles_n, bm, bn);
  const int m0 = bm * BM, n0 = bn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  // staging: chunk f = tid + NT i -> (row f / 8, chunk f % 8): 8 consecutive lanes cover one 128-byte row segment
  u32x4_t ra[CA], rb[CB];
  const int srow = tid >> 3, sch = tid & 7;
  auto gload = [&](int k0) {
    const int k = k0 + 8 * sch;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int ma = m0 + srow + (NT / 8) * i;
      ra[i] = (u32x4_t){0u, 0u, 0u, 0u};
      if (ma < M && k < kend) ra[i] = *reinterpret_cast<const u32x4_t*>(A + (long)ma * lda + k);     // (K, kbeg multiples of 8: whole chunks)
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int nb = n0 + srow + (NT / 8) * i;
      rb[i] = (u32x4_t){0u, 0u, 0u, 0u};
      if (nb < N && k < kend) rb[i] = *reinterpret_cast<const u32x4_t*>(B + (long)nb * ldb + k);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* As = g16_smem + buf * (BM + BN) * 128;
    unsigned char* Bs = As + BM * 128;
#pragma unroll
    for (int i = 0; i < CA; ++i) *reinterpret_cast<u32x4_t*>(As + g16_off(srow + (NT / 8) * i, sch)) = ra[i];
#pragma unroll
    for (int i = 0; i < CB; ++i) *reinterpret_cast<u32x4_t*>(Bs + g16_off(srow + (NT / 8) * i, sch)) = rb[i];
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto compute = [&](int buf) {
    const unsigned char* As = g16_smem + buf * (BM + BN) * 128;
    const unsigned char* Bs = As + BM * 128;
#pragma unroll
    for (int s = 0; s < G16_BK / 16; ++s) {              // MFMA k-step s: this lane's 8 k = chunk 2 s + lh
      bf16x8 a8[TM], b8[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a8[i] = *reinterpret_cast<const bf16x8*>(As + g16_off((wm * TM + i) * 32 + l31, 2 * s + lh));
#pragma unroll
      for (int j = 0; j < TN; ++j) b8[j] = *reinterpret_cast<const bf16x8*>(Bs + g16_off((wn * TN + j) * 32 + l31, 2 * s + lh));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nk = (kend - kbeg + G16_BK - 1) / G16_BK;
  if constexpr (STG == 1) {
    constexpr int NW = WM * WN;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "glds map");
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    static_assert(NW % 2 == 0, "the source swizzle below takes the row group's parity from the wave");
    // dest row within its 8-row group; SOURCE chunk for dest position lane & 7 of row 8 g + r8, g = wave + NW i (g_swz of that row)
    const int r8 = lane >> 3, c8 = (lane & 7) ^ ((4 * (wave & 1) + (r8 >> 1)) & 7);
    auto stage = [&](int k0, int buf) {
      unsigned char* As = g16_smem + buf * (BM + BN) * 128;
      unsigned char* Bs = As + BM * 128;
#pragma unroll
      for (int i = 0; i < BM / 8 / NW; ++i) {
        const int g = wave + NW * i;
        const int ma = min(m0 + 8 * g + r8, M - 1);
        __builtin_amdgcn_global_load_lds(A + (long)ma * lda + k0 + 8 * c8, (lds_ptr_t)(As + g * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < BN / 8 / NW; ++i) {
        const int g = wave + NW * i;
        const int nb = min(n0 + 8 * g + r8, N - 1);
        __builtin_amdgcn_global_load_lds(B + (long)nb * ldb + k0 + 8 * c8, (lds_ptr_t)(Bs + g * 1024), 16, 0, 0);
      }
    };
    if constexpr (NBUF == 3) {
      // Three buffers, loads TWO tiles ahead and in flight ACROSS the barrier: a counted wait (the youngest tile's loads may stay
      // outstanding: a wave's loads return in order) and a raw s_barrier - __syncthreads() would drain the LDS-DMA queue first.
      // Tile kt + 2 goes into the buffer compute(kt - 1) read: behind the barrier every wave is done with it.
      constexpr int LPT = BM / 8 / NW + BN / 8 / NW;            // direct-to-LDS loads per tile and wave
      stage(kbeg, 0);
      if (nk > 1) stage(kbeg + G16_BK, 1);
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) stage(kbeg + (kt + 2) * G16_BK, (kt + 2) % 3);
        compute(kt % 3);
      }
    } else if constexpr (NBUF == 2) {
      stage(kbeg, 0);
      for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                        // tile kt landed for everyone; everyone is done with compute(kt - 1)
        if (kt + 1 < nk) stage(kbeg + (kt + 1) * G16_BK, (kt + 1) & 1);
        compute(kt & 1);
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) {
        stage(kbeg + kt * G16_BK, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        compute(0);
        __syncthreads();
      }
    }
  } else if constexpr (NBUF == 2) {
    gload(kbeg);
    lstore(0);
    if (nk > 1) gload(kbeg + G16_BK);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) lstore((kt + 1) & 1);
      if (kt + 2 < nk) gload(kbeg + (kt + 2) * G16_BK);
      compute(kt & 1);
      __syncthreads();
    }
  } else {
    gload(kbeg);
    for (int kt = 0; kt < nk; ++kt) {
      lstore(0);
      __syncthreads();
      if (kt + 1 < nk) gload(kbeg + (kt + 1) * G16_BK);
      compute(0);
      __syncthreads();
    }
  }
  // (compile-time indices: with 4 x 2 accumulator tiles the plain unrolled loops were NOT fully unrolled and the whole accumulator array
  // went to scratch - 576 bytes per lane, 150 TF)
  g16_static_for<0, TM>([&](auto i) {
    g16_static_for<0, 16>([&](auto r) {
      constexpr int ic = decltype(i)::value, rc = decltype(r)::value;
      const int row = m0 + (wm * TM + ic) * 32 + (rc & 3) + 8 * (rc >> 2) + 4 * lh;
      const long srow2 = ep.map_row(row);
      g16_static_for<0, TN>([&](auto j) {
        constexpr int jc = decltype(j)::value;
        ep.put(row, srow2, n0 + (wn * TN + jc) * 32 + l31, acc[ic][jc][rc]);
      });
    });
  });
}

static hipError_t g16_attr_status = hipSuccess;                 // last hipFuncSetAttribute result of g16_launch (checked by the caller)
template <int WM, int WN, int TM, int TN, int NBUF, int STG>
static void g16_launch(const G16Launch& g) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  constexpr size_t smem = (size_t)NBUF * (BM + BN) * 128;
  auto kern = gemm16_nt_kernel<WM, WN, TM, TN, NBUF, STG>;
  // per device (a process may drive several): the attribute lives with the device's copy of the code object
  static bool attr[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr[dev]) {
    g16_attr_status = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (g16_attr_status != hipSuccess && smem > 64 * 1024) return;       // the caller reports it (asr_gemm_bf16_nt)
    attr[dev] = true;
  }
  const asr_gemm_desc* d = g.d;
  const int tm = asr_cdiv(d->M, BM), tn = asr_cdiv(d->N, BN);
  int k_chunk = asr_cdiv(asr_cdiv(d->K, g.sk), G16_BK) * G16_BK;
  if (k_chunk <= 0) k_chunk = G16_BK;
  dim3 grid((unsigned)(tm * tn), 1, (unsigned)(d->batch * g.sk));
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, g.st, g.A, d->lda, g.B, d->ldb, g.ep, d->M, d->N, d->K, tm, tn, g.sk, k_chunk, d->stride_a,
                     d->stride_b, d->stride_c);
}
static int g16_cfg = -1;
static int g16_config() {
  if (g16_cfg < 0) { const char* e = getenv("ASR_G16_CFG"); g16_cfg = e ? atoi(e) : 100; }   // 100 = by shape
  return g16_cfg;
}
extern "C" int asr_gemm_bf16_config(int cfg) {                   // tuning / tests: select the tile configuration (-1: leave), returns the previous one
  const int old = g16_config();
  if (cfg >= 0) g16_cfg = cfg;
  return old;
}

extern "C" int asr_gemm_bf16_nt(const asr_gemm_desc* d, const void* A16, const void* B16, float* C, void* stream) {
  ASR_CHECK(d && A16 && B16 && C, ASR_ERR_ARG, "asr_gemm_bf16_nt: null argument");
  ASR_CHECK(d->M > 0 && d->N > 0 && d->K > 0 && d->batch >= 1, ASR_ERR_SHAPE, "asr_gemm_bf16_nt: bad M/N/K/batch %d %d %d %d", d->M, d->N, d->K, d->batch);
  ASR_CHECK(!d->trans_a && d->trans_b, ASR_ERR_ARG, "asr_gemm_bf16_nt: operands must be k-contiguous (trans_a = 0, trans_b = 1): A [M,K], B [N,K]");
  ASR_CHECK(d->K % 8 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && ((uintptr_t)A16 & 15) == 0 && ((uintptr_t)B16 & 15) == 0 &&
                d->stride_a % 8 == 0 && d->stride_b % 8 == 0,
            ASR_ERR_SHAPE, "asr_gemm_bf16_nt: K, leading dimensions and batch strides must be multiples of 8 elements, operands 16-byte aligned");
  ASR_CHECK(d->lda >= d->K && d->ldb >= d->K && d->ldc >= d->N, ASR_ERR_SHAPE, "asr_gemm_bf16_nt: leading dimension smaller than the row length");
  ASR_CHECK(!d->a_scale, ASR_ERR_ARG, "asr_gemm_bf16_nt: fold the A row-group scale into the bf16 image (asr_f32_to_bf16_image)");
  ASR_CHECK(!(d->c_scale && d->c_rpg <= 0), ASR_ERR_ARG, "asr_gemm_bf16_nt: group scale needs rows-per-group > 0");
  const int sk = d->split_k > 1 ? d->split_k : 1;
  ASR_CHECK(!(sk > 1 && !d->accumulate), ASR_ERR_ARG, "asr_gemm_bf16_nt: split_k > 1 accumulates atomically: set accumulate and pre-zero C");
  ASR_CHECK((long)d->batch * sk <= 65535, ASR_ERR_SHAPE, "asr_gemm_bf16_nt: batch * split_k > 65535");
  int mode = d->accumulate ? 1 : 0;
  if (d->batch > 1 && d->stride_c == 0) {
    ASR_CHECK(d->accumulate, ASR_ERR_ARG, "asr_gemm_bf16_nt: batch > 1 with stride_c == 0 (split-K over the batch) requires accumulate = 1");
    mode = 2;
  }
  if (d->accumulate == 2 || sk > 1) mode = 2;
  GemmEpilogue ep{C, d->ldc, d->M, d->N, d->alpha, d->bias, d->c_scale, d->c_rpg, mode, d->relu, nullptr, 0u, 0.f};
  G16Launch g{static_cast<const bf16_t*>(A16), static_cast<const bf16_t*>(B16), ep, d, sk, (hipStream_t)stream};
  const bool whole = d->K % G16_BK == 0;                        // (k_chunk is a multiple of 64 by construction)
  int cfg = g16_config();
  // by shape: the deep pipelines pay with many K tiles per workgroup (weight / input gradients, K >= 2048); products with a
  // short K and a large C ([8128 x 1024] x [1024 x 16000]: 419 against 476 us) are bound by writing C and want more, smaller workgroups
  if (cfg == 100) {
    cfg = (d->K / sk >= 2048) ? 12 : 0;
    if (whole && d->K / sk >= 2048 && d->M >= 256 && d->N >= 256) {
      cfg = 15;                                                  // the eight-phase 256 x 256 schedule (gemm16_8p.hip)
      // A grid short of the chip (the 256 x 256 tiles of a weight gradient number 64-128): split K further where C is accumulated anyway -
      // the partial sums then meet in atomic adds ([2048 x 31936] x [31936 x 4096]: 615 us in one piece, 448 us in two)
      const long tiles = (long)asr_cdiv(d->M, 256) * asr_cdiv(d->N, 256) * d->batch;
      if (mode != 0) {
        int s2 = sk;
        while (tiles * s2 * 2 <= 288 && d->K / (s2 * 2) >= 2048) s2 *= 2;
        if (s2 != sk) { g.sk = s2; g.ep.mode = 2; }
      }
    }
  }
  if (!whole && cfg >= 4) cfg = 0;
  if (cfg >= 12 && cfg <= 14 && asr_cdiv(d->K, sk) < 3 * G16_BK) cfg = 0;   // (the three-buffer pipeline wants at least three K tiles)
  switch (cfg) {
    case 1: g16_launch<2, 2, 2, 2, 2, 0>(g); break;
    case 2: g16_launch<4, 2, 2, 2, 2, 0>(g); break;              // 256 x 128, 8 waves
    case 3: g16_launch<4, 2, 2, 2, 1, 0>(g); break;
    case 4: g16_launch<2, 2, 2, 2, 1, 1>(g); break;              // direct-to-LDS staging
    case 5: g16_launch<2, 2, 2, 2, 2, 1>(g); break;
    case 6: g16_launch<4, 2, 2, 2, 1, 1>(g); break;
    case 7: g16_launch<4, 2, 2, 2, 2, 1>(g); break;
    case 8: g16_launch<2, 4, 4, 2, 2, 1>(g); break;              // 256 x 256, 8 waves of 128 x 64, direct-to-LDS, two buffers (128 KB LDS)
    case 9: g16_launch<2, 4, 4, 2, 1, 1>(g); break;
    case 10: g16_launch<2, 2, 4, 2, 2, 1>(g); break;             // 256 x 128, 4 waves of 128 x 64
    case 12: g16_launch<4, 2, 2, 2, 3, 1>(g); break;             // 256 x 128, 8 waves, THREE buffers, loads in flight across the barrier
    case 13: g16_launch<2, 2, 2, 2, 3, 1>(g); break;             // 128 x 128, 4 waves, three buffers
    case 14: g16_launch<2, 2, 4, 2, 3, 1>(g); break;             // 256 x 128, 4 waves of 128 x 64, three buffers
    case 11: g16_launch<2, 2, 4, 2, 1, 1>(g); break;
    case 15: g16_attr_status = g16_launch_8p(g); break;                            // 256 x 256, 8 waves, eight-phase schedule
    default: g16_launch<2, 2, 2, 2, 1, 0>(g); break;             // 128 x 128, 4 waves, one LDS buffer, register staging (any K; the fallback of 12)
  }
  if (g16_attr_status != hipSuccess && cfg != 0) {
    // this device refused the large-LDS attribute: nothing was launched; the 32 KB configuration needs no attribute
    g16_attr_status = hipSuccess;
    (void)hipGetLastError();
    g16_launch<2, 2, 2, 2, 1, 0>(g);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// ------------------------------------------------------------------------------------------ bf16 images of f32 operands
// Source row r of a (possibly 3-D) operand: batch r / rpb at stride bstr, row r % rpb at stride lds (rpb = R: plain 2-D).
// dst[r][c] = bf16(src[r][c] * scale[(r / rpg)][c])   (scale optional: the Keras RNN input-dropout table, one row per `rpg` source rows)
__global__ __launch_bounds__(256) void bf16_image_kernel(const float* src, long lds, int R, int Cc, int rpb, long bstr, const float* scale, int rpg,
                                                         bf16_t* dst, long ldd) {
  const long n4 = Cc >> 2, total = (long)R * n4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / n4, c = (i % n4) * 4;
    float4 v = *reinterpret_cast<const float4*>(src + (r / rpb) * bstr + (r % rpb) * lds + c);
    if (scale) {
      const float4 s = *reinterpret_cast<const float4*>(scale + (r / rpg) * Cc + c);
      v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
    }
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *reinterpret_cast<bf16x4*>(dst + r * ldd + c) = o;
  }
}
// dst[c][r] = bf16(src[r][c] * scale[(r / rpg)][c]): 64 x 64 tiles through LDS, coalesced on both sides.  (A 128 x 64 tile written as packed
// bf16 pairs measured 1.5x SLOWER - 4-way LDS conflicts on the paired read and half the blocks in flight; this pass runs at ~3.1 TB/s.)
// Destination column of source row r: (r / rpb) * drpb + r % rpb + dshift - the batches of a 3-D source can land at a different pitch and
// offset (the recurrent-kernel gradient pairs h[b, t -+ 1] with ds[b, t]: the shifted h image then shares ds's transposed image).
// tb_nb > 0: TIME-MAJOR columns.  The R = rows x tb_nb source rows are enumerated r = t * tb_nb + b (a tile's 64 rows are 64 batches of
// one time step: their 256-byte pieces lie a batch stride apart), the destination column is r + dshift, the scale row is b.
__global__ __launch_bounds__(256) void bf16_image_t_kernel(const float* src, long lds, int R, int Cc, int rpb, long bstr, const float* scale, int rpg,
                                                           bf16_t* dst, long ldd, int drpb, long dshift, int tb_nb) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < Cc) {
      if (tb_nb) {
        const int t = r / tb_nb, b = r % tb_nb;
        v = src[(long)b * bstr + (long)t * lds + c];
        if (scale) v *= scale[(long)b * Cc + c];
      } else {
        v = src[(long)(r / rpb) * bstr + (long)(r % rpb) * lds + c];
        if (scale) v *= scale[(long)(r / rpg) * Cc + c];
      }
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < R) {
      const __bf16 b = (__bf16)tile[tx][i];
      const long col = tb_nb ? (long)r + dshift : (long)(r / rpb) * drpb + r % rpb + dshift;
      dst[(long)c * ldd + col] = __builtin_bit_cast(bf16_t, b);
    }
  }
}
extern "C" int asr_f32_to_bf16_image(const float* src, long ld_src, int rows, int cols, int rows_per_batch, long batch_stride, const float* scale,
                                     int rows_per_group, int transpose, void* dst, long ld_dst, int dst_rows_per_batch, int dst_shift, void* stream) {
  ASR_CHECK(src && dst && rows > 0 && cols > 0 && ld_src >= cols, ASR_ERR_ARG, "asr_f32_to_bf16_image: bad argument");
  const int rpb = rows_per_batch > 0 ? rows_per_batch : rows;
  ASR_CHECK(!(scale && rows_per_group <= 0), ASR_ERR_ARG, "asr_f32_to_bf16_image: scale needs rows_per_group > 0");
  const int drpb = dst_rows_per_batch > 0 ? dst_rows_per_batch : rpb;
  ASR_CHECK((dst_rows_per_batch == 0 && dst_shift == 0) || (transpose && dst_shift >= 0 && drpb >= rpb + dst_shift), ASR_ERR_ARG,
            "asr_f32_to_bf16_image: destination batching is for the transposed image, with rows_per_batch + shift <= dst_rows_per_batch");
  ASR_CHECK(ld_dst >= (transpose ? (long)asr_cdiv(rows, rpb) * drpb : cols), ASR_ERR_SHAPE, "asr_f32_to_bf16_image: destination rows too short");
  hipStream_t st = (hipStream_t)stream;
  if (transpose) {
    hipLaunchKernelGGL(bf16_image_t_kernel, dim3((unsigned)asr_cdiv(cols, 64), (unsigned)asr_cdiv(rows, 64)), dim3(256), 0, st, src, ld_src, rows, cols, rpb,
                       batch_stride, scale, rows_per_group, static_cast<bf16_t*>(dst), ld_dst, drpb, (long)dst_shift, 0);
  } else {
    ASR_CHECK(cols % 4 == 0 && ld_src % 4 == 0 && batch_stride % 4 == 0 && ld_dst % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0 &&
                  (!scale || ((uintptr_t)scale & 15) == 0),
              ASR_ERR_SHAPE, "asr_f32_to_bf16_image: the straight image needs columns / leading dimensions in multiples of 4 and aligned buffers");
    const long total = (long)rows * (cols >> 2);
    const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(bf16_image_kernel, dim3(grid), dim3(256), 0, st, src, ld_src, rows, cols, rpb, batch_stride, scale, rows_per_group,
                       static_cast<bf16_t*>(dst), ld_dst);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_f32_to_bf16_image_tb(const float* src, long ld_src, int nbatch, int rows, int cols, long batch_stride, const float* scale, void* dst,
                                        long ld_dst, long dst_shift, void* stream) {
  ASR_CHECK(src && dst && nbatch > 0 && rows > 0 && cols > 0 && ld_src >= cols && dst_shift >= 0, ASR_ERR_ARG, "asr_f32_to_bf16_image_tb: bad argument");
  ASR_CHECK((long)nbatch * rows < 2147483647L && ld_dst >= (long)nbatch * rows + dst_shift, ASR_ERR_SHAPE, "asr_f32_to_bf16_image_tb: destination rows too short");
  const int R = nbatch * rows;
  hipLaunchKernelGGL(bf16_image_t_kernel, dim3((unsigned)asr_cdiv(cols, 64), (unsigned)asr_cdiv(R, 64)), dim3(256), 0, (hipStream_t)stream, src, ld_src, R, cols,
                     1, batch_stride, scale, 1, static_cast<bf16_t*>(dst), ld_dst, 1, dst_shift, nbatch);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
