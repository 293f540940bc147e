// asr_gemm_bf16_nt, tile configuration 15 (gemm16.hip holds the entry point, the other configurations and the image passes).
#include "gemm16.h"

// ------------------------------------------------------------------------------------------ 256 x 256 x 64, eight phases per pair of K tiles
// The schedule of cdna_hip_programming.md 5 ("The 256^2 8-phase template"), rebuilt here on this file's operand images (32x32x16 MFMA, 128-byte
// LDS rows, chunk ^ g16_swz(row)): 8 waves as 2 (M) x 4 (N); the tile's A rows and B rows are split in HALVES of 128 (A0 A1 B0 B1, 16 KB each per K tile,
// two K tiles resident = 128 KB); a wave owns 64 rows of each A half and 32 columns of each B half, i.e. four 64 x 32 output QUADRANTS, and one
// phase = {read the fragments the next quadrant needs, issue ONE half tile of direct-to-LDS loads, barrier, 8 MFMAs (quadrant x K = 64), barrier}:
//   phase  reads          computes   stages (K tile t is the one computed)     last read of the staged buffer
//   q0     B0 (4), A0 (8)  A0 x B0    A1 of t + 1                               q2 of t - 1
//   q1     B1 (4)          A0 x B1    B0 of t + 2                               q0 (retired BEFORE q0's first barrier: lgkmcnt(8), B reads first)
//   q2     A1 (8)          A1 x B1    A0 of t + 2                               q0
//   q3     -               A1 x B0    B1 of t + 2; wait: at most 3 half tiles (6 loads) outstanding => tile t + 1 has landed, read from q0 on
// The loads are never drained inside the loop (counted s_waitcnt vmcnt, raw s_barrier: __syncthreads() would wait for the LDS-DMA queue), and the
// waves wr = 1 run ONE BARRIER BEHIND the waves wr = 0 (every SIMD holds one wave of each group): while one group issues its MFMAs the other issues
// its LDS reads and loads.  Hazards under that stagger, group 0 ahead: a buffer is read by the late group at most one barrier after the early
// group's matching read, and every restaging above is issued >= 2 barriers after the early group's read retired (q1's B0: the lgkmcnt(8) before
// q0's first barrier retires the late group's B0 reads before the early group passes its second); a landed tile is read one phase after the wait
// that retires it (the late group's wait precedes its first barrier of q3 = the early group's second).
// Needs whole K tiles (K and the K chunk multiples of 64); rows beyond M / N are clamped on load and dropped by the epilogue.
// ABL: timing experiments only (results wrong unless 0 or 1) - 1 no stagger, 2 no direct-to-LDS loads inside the loop, 3 no fragment reads inside
// the loop, 4 no MFMAs, 5 no priority raise, 6 neither stagger nor priority
template <int ABL, int MF>
__global__ __launch_bounds__(512) void gemm16_8p_kernel(const bf16_t* A, long lda, const bf16_t* B, long ldb, GemmEpilogue ep, int M, int N, int K,
                                                        int tiles_m, int tiles_n, int split_k, int k_chunk, long sAz, long sBz, long sCz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char g16_smem[];     // [buffer 2][A0 A1 B0 B1][128 rows][128 B]
  const int z = blockIdx.z / split_k, zs = blockIdx.z % split_k;
  A += (long)z * sAz; B += (long)z * sBz; ep.C += (long)z * sCz;
  if (zs != 0 || (z != 0 && sCz == 0)) ep.bias = nullptr;
  const int kbeg = zs * k_chunk, kend = min(K, kbeg + k_chunk);
  if (kbeg >= K) return;
  int bm, bn;
  g16_tile(blockIdx.x, tiles_m, tiles_n, bm, bn);
  const int m0 = bm * 256, n0 = bn * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  // staging: a wave instruction fills 8 rows (1 KB); wave w fills the row groups w and w + 8 of every half tile
  const int r8 = lane >> 3, c8 = (lane & 7) ^ ((4 * (wave & 1) + (r8 >> 1)) & 7);     // source chunk for dest position lane & 7 (g16_swz of row 8 g + r8)
  long asrc[2][2], bsrc[2][2];                                 // element offsets of this lane's source chunk, [half][group]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = h * 128 + 8 * (wave + 8 * j) + r8;
      asrc[h][j] = (long)min(m0 + row, M - 1) * lda + kbeg + 8 * c8;
      bsrc[h][j] = (long)min(n0 + row, N - 1) * ldb + kbeg + 8 * c8;
    }
  auto stage = [&](auto opnd, auto half, int kt) {             // operand 0 = A, 1 = B; into buffer kt & 1
    constexpr int X = decltype(opnd)::value, H = decltype(half)::value;
    if (ABL == 2 && kt >= 2) return;
    unsigned char* dst = g16_smem + (kt & 1) * 65536 + (X * 2 + H) * 16384 + wave * 1024;
    const bf16_t* src = X ? B : A;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds(src + (X ? bsrc[H][j] : asrc[H][j]) + (long)kt * 64, (lds_ptr_t)(dst + j * 8192), 16, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // fragment reads: this lane's byte offsets inside a half tile.  MF = 32 (v_mfma_f32_32x32x16_bf16): lane = (row l & 31, chunk 2 s + (l >> 5)) for
  // k-step s of 4, a wave's quadrant = 2 m tiles x 1 n tile; MF = 16 (v_mfma_f32_16x16x32_bf16): lane = (row l & 15, chunk 4 s + (l >> 4)) for k-step
  // s of 2, quadrant = 4 m tiles x 2 n tiles.  Same 8 + 4 reads and the same arithmetic per phase; the 16 x 16 form holds a higher clock on
  // random operands (MI355X_MICROARCH.md, DVFS give-back (7)).
  constexpr int KS = MF == 32 ? 4 : 2, MT = MF == 32 ? 2 : 4, NT = MF == 32 ? 1 : 2;
  const int lrow = MF == 32 ? l31 : (lane & 15), lk = MF == 32 ? lh : (lane >> 4);
  int xo[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) xo[s] = g16_off(lrow, (8 / KS) * s + lk);
  const int arow = wr * 64 * 128, brow = 32768 + wc * 32 * 128;
  bf16x8 a[MT][KS], b0[NT][KS], b1[NT][KS];
  bool first = true;
  auto rdA = [&](int buf, auto half) {
    if (ABL == 3 && !first) return;
    const unsigned char* base = g16_smem + buf * 65536 + decltype(half)::value * 16384 + arow;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int s = 0; s < KS; ++s) a[i][s] = *reinterpret_cast<const bf16x8*>(base + i * MF * 128 + xo[s]);
  };
  auto rdB = [&](int buf, auto half, bf16x8 (&b)[NT][KS]) {
    if (ABL == 3 && !first) return;
    const unsigned char* base = g16_smem + buf * 65536 + decltype(half)::value * 16384 + brow;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int s = 0; s < KS; ++s) b[j][s] = *reinterpret_cast<const bf16x8*>(base + j * MF * 128 + xo[s]);
  };
  typedef typename std::conditional<MF == 32, f32x16, f32x4>::type acc_t;
  acc_t acc[2][MT][2][NT];                                     // [A half][m tile][B half][n tile]
#pragma unroll
  for (int i = 0; i < 2 * MT * 2 * NT; ++i)
#pragma unroll
    for (int r = 0; r < (MF == 32 ? 16 : 4); ++r) (&acc[0][0][0][0])[i][r] = 0.f;
  auto quad = [&](auto ha, auto hb, bf16x8 (&b)[NT][KS]) {     // first barrier .. second barrier of a phase
    constexpr int HA = decltype(ha)::value, HB = decltype(hb)::value;
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ABL != 5 && ABL != 6) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if constexpr (ABL == 4) asm volatile("" ::"v"(a[i][s]), "v"(b[j][s]));
          else if constexpr (MF == 32) acc[HA][i][HB][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][s], b[j][s], acc[HA][i][HB][j], 0, 0, 0);
          else acc[HA][i][HB][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][s], b[j][s], acc[HA][i][HB][j], 0, 0, 0);
        }
    if (ABL != 5 && ABL != 6) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  };
  const int nk = (kend - kbeg) / G16_BK;
  // prologue: tile 0 and the first three half tiles of tile 1 (the steady state keeps three half tiles in flight)
  stage(I1{}, I0{}, 0); stage(I0{}, I0{}, 0); stage(I1{}, I1{}, 0); stage(I0{}, I1{}, 0);
  if (nk > 1) {
    stage(I1{}, I0{}, 1); stage(I0{}, I0{}, 1); stage(I1{}, I1{}, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (ABL != 1 && ABL != 6 && wr == 1) __builtin_amdgcn_s_barrier();                   // the stagger: the waves wr = 1 stay one barrier behind from here on
  auto tile = [&](int t, auto steady) {
    constexpr bool ST = decltype(steady)::value;               // steady state: tiles t + 1 and t + 2 exist
    const int buf = t & 1;
    rdB(buf, I0{}, b0);
    __builtin_amdgcn_sched_barrier(0);
    rdA(buf, I0{});
    if (ST || t + 1 < nk) stage(I0{}, I1{}, t + 1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");         // the four B0 reads (issued first) have returned: B0 may be restaged next phase
    quad(I0{}, I0{}, b0);
    rdB(buf, I1{}, b1);
    if (ST || t + 2 < nk) stage(I1{}, I0{}, t + 2);
    quad(I0{}, I1{}, b1);
    rdA(buf, I1{});
    if (ST || t + 2 < nk) stage(I0{}, I0{}, t + 2);
    quad(I1{}, I1{}, b1);
    if (ST || t + 2 < nk) {
      stage(I1{}, I1{}, t + 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    quad(I1{}, I0{}, b0);
    first = false;
  };
  int t = 0;
  for (; t + 2 < nk; ++t) tile(t, std::true_type{});
  for (; t < nk; ++t) tile(t, std::false_type{});
  if (ABL != 1 && ABL != 6 && wr == 0) __builtin_amdgcn_s_barrier();                   // (the barrier the late group is still owed)
  // C: MF = 32: col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5); MF = 16: col = l & 15, row = 4 (l >> 4) + r
  g16_static_for<0, 2 * MT>([&](auto i) {
    g16_static_for<0, (MF == 32 ? 16 : 4)>([&](auto r) {
      constexpr int ic = decltype(i)::value, rc = decltype(r)::value, ha = ic / MT, mt = ic % MT;
      const int row = m0 + ha * 128 + wr * 64 + mt * MF + (MF == 32 ? (rc & 3) + 8 * (rc >> 2) + 4 * lh : 4 * lk + rc);
      const long srow2 = ep.map_row(row);
      g16_static_for<0, 2 * NT>([&](auto j) {
        constexpr int jc = decltype(j)::value, hb = jc / NT, nt = jc % NT;
        ep.put(row, srow2, n0 + hb * 128 + wc * 32 + nt * MF + lrow, acc[ha][mt][hb][nt][rc]);
      });
    });
  });
}

hipError_t g16_launch_8p(const G16Launch& g) {
  static const int abl = getenv("ASR_G16_8P_ABL") ? atoi(getenv("ASR_G16_8P_ABL")) : 0;
  static const int mf = getenv("ASR_G16_8P_MFMA") ? atoi(getenv("ASR_G16_8P_MFMA")) : 16;
  auto kern = abl == 1 ? gemm16_8p_kernel<1, 16> : abl == 2 ? gemm16_8p_kernel<2, 16> : abl == 3 ? gemm16_8p_kernel<3, 16> : abl == 4 ? gemm16_8p_kernel<4, 16>
            : abl == 5 ? gemm16_8p_kernel<5, 16> : abl == 6 ? gemm16_8p_kernel<6, 16> : mf == 32 ? gemm16_8p_kernel<0, 32> : gemm16_8p_kernel<0, 16>;
  static unsigned long long seen = 0;
  if (asr_first_use_on_device(seen)) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { seen = 0; return e; }
  }
  const asr_gemm_desc* d = g.d;
  const int tm = asr_cdiv(d->M, 256), tn = asr_cdiv(d->N, 256);
  int k_chunk = asr_cdiv(asr_cdiv(d->K, g.sk), G16_BK) * G16_BK;
  if (k_chunk <= 0) k_chunk = G16_BK;
  dim3 grid((unsigned)(tm * tn), 1, (unsigned)(d->batch * g.sk));
  hipLaunchKernelGGL(kern, grid, dim3(512), 128 * 1024, g.st, g.A, d->lda, g.B, d->ldb, g.ep, d->M, d->N, d->K, tm, tn, g.sk, k_chunk,
                     d->stride_a, d->stride_b, d->stride_c);
  return hipSuccess;
}
