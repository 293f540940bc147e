// Backward step of the recurrent cells (gradient of rnn.hip's forward step) for gfx950.
//
// The quantity handed from one backward step to the next is ds, the gradient with respect to the
// gate pre-activation sums ("slots", [B, NS*H], written in place over the saved activations) - the
// mirror image of the forward pass handing h around.  A step's workgroup owns 16 hidden units x 16
// batch rows and computes, for those units only,
//     dh[b, j] = sum_c ds_consumer[b, c] * W_consumer[j, c]          (consumer = the cell(s) that read h_j)
// straight from the Keras-layout weights (row j of the recurrent / input kernel is contiguous), with
// the c axis split over the workgroup's 16 waves (v_mfma_f32_16x16x4_f32, both operands loaded as one
// float4 per lane per 16 columns), reduced through LDS, followed by the gate-gradient math of the
// owned units.  Compared with a K-split over workgroups (partial-sum slabs) this writes 16x less
// data per launch - and a dependent launch on MI355X is priced mostly by the dirty lines it leaves.
// A second "linear" mode of the same kernel only forms the sums (gradients of cell inputs that are
// not states: the attention context, the initial states).
#include <stdlib.h>

#include "common.h"

#define CELL_LSTM 0
#define CELL_GRU 1
#define CELL_RNN 2
#define BW_NW 16      // waves per workgroup
#ifndef BW_CH
#define BW_CH 4
#endif       // 16-column blocks a wave keeps in flight

struct BackSrc {
  const float* D; long ldd;
  const float* W; long ldw;
  const unsigned short* W16;     // bf16 image of W (same layout) or NULL
  int nseg; int d_col0[4], w_col0[4], len[4];
  const float* oscale[4]; long oscale_ld;   // optional per-segment multiplier [B, n_units] on the segment's product (recurrent dropout: one mask per gate)
  int vec;
  float drop_rate; uint32_t drop_stream; long drop_ld; int drop_off;
};
struct Bwd2Dir {
  int n_units;
  BackSrc src[2];                      // [0] -> gradient wrt the state h, [1] -> gradient wrt the emitted output y
  const float* addA; long addA_ld;
  const float* addB; long addB_ld;
  float* direct; long direct_ld;       // carried part of dh (masked rows / GRU z * dh): read, then overwritten
  float* out; long out_ld;             // linear mode when non-null
  float* dc; long dc_ld;
  float* dy_carry; long dy_carry_ld;
  const uint8_t* mask; long mask_ld;
  const float* saved; long saved_ld;
  const float* h_prev; long h_prev_ld;
  const float* c_prev; long c_prev_ld;
  const float* c_out; long c_out_ld;
  float* dslots; long dslots_ld;
};
struct Bwd2Args { Bwd2Dir d[2]; int B; const uint32_t* seed; };

// partial[16 rows x 16 units] (this wave's share of the column blocks) of  D[rows, cols] x W[units, cols]^T
__device__ __forceinline__ f32x4 back_partial(const BackSrc& s, int b0, int unit0, int B, int n_units, int wave, int li, int lq) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (s.D == nullptr) return acc;
  const int brow = b0 + li, urow = unit0 + li;
  const bool aok = brow < B, bok = urow < n_units;
  for (int g = 0; g < s.nseg; ++g) {
    const int len = s.len[g], nb = (len + 15) >> 4;
    const float* dr = s.D + (long)brow * s.ldd + s.d_col0[g];
    const float* wr = s.W + (long)urow * s.ldw + s.w_col0[g];
    const f32x4 before = acc;
    for (int j0 = wave; j0 < nb; j0 += BW_NW * BW_CH) {
      float4 av[BW_CH], bv[BW_CH];
#pragma unroll
      for (int i = 0; i < BW_CH; ++i) {
        const int jb = j0 + BW_NW * i, k = 16 * jb + 4 * lq;
        av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (jb < nb) {
          if (s.vec && k + 3 < len) {
            if (aok) av[i] = *reinterpret_cast<const float4*>(dr + k);
            if (bok) bv[i] = *reinterpret_cast<const float4*>(wr + k);
          } else {
            if (aok) { av[i].x = k < len ? dr[k] : 0.f; av[i].y = k + 1 < len ? dr[k + 1] : 0.f; av[i].z = k + 2 < len ? dr[k + 2] : 0.f; av[i].w = k + 3 < len ? dr[k + 3] : 0.f; }
            if (bok) { bv[i].x = k < len ? wr[k] : 0.f; bv[i].y = k + 1 < len ? wr[k + 1] : 0.f; bv[i].z = k + 2 < len ? wr[k + 2] : 0.f; bv[i].w = k + 3 < len ? wr[k + 3] : 0.f; }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < BW_CH; ++i) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bv[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bv[i].w, acc, 0, 0, 0);
      }
    }
    if (s.oscale[g] != nullptr) {
      // this segment's product times its own mask (Keras implementation 1: the consumer read h * mask_g through gate g's columns of
      // its recurrent kernel); C/D map: this lane holds rows 4 lq + r of column li
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int br = b0 + 4 * lq + r, un = unit0 + li;
        const float m = (br < B && un < n_units) ? s.oscale[g][(long)br * s.oscale_ld + un] : 0.f;
        acc[r] = before[r] + (acc[r] - before[r]) * m;
      }
    }
  }
  return acc;
}

// Element-wise part of one backward step for the (batch row b, unit j) pair, from the two matrix-product sums.
struct BwdOperands {
  bool m;
  float carry, svv[4], cpv, cov, dcv, hpv, addAv, addBv, dirv;
};

template <int CELL>
__device__ __forceinline__ void load_bwd_operands(const Bwd2Dir& d, int b, int j, int H, bool linear, BwdOperands& o) {
  constexpr int NSV = CELL == CELL_RNN ? 1 : 4;
  o.m = true; o.carry = 0.f; o.cpv = 0.f; o.cov = 0.f; o.dcv = 0.f; o.hpv = 0.f; o.addAv = 0.f; o.addBv = 0.f; o.dirv = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) o.svv[g] = 0.f;
  if (d.addA) o.addAv = d.addA[(long)b * d.addA_ld + j];
  if (d.addB) o.addBv = d.addB[(long)b * d.addB_ld + j];
  if (d.direct) o.dirv = d.direct[(long)b * d.direct_ld + j];
  if (!linear) {
    o.m = d.mask ? d.mask[(long)b * d.mask_ld] != 0 : true;
    o.carry = d.dy_carry ? d.dy_carry[(long)b * d.dy_carry_ld + j] : 0.f;
    const float* sv = d.saved + (long)b * d.saved_ld + j;
#pragma unroll
    for (int g = 0; g < NSV; ++g) o.svv[g] = sv[(long)g * H];
    if (CELL == CELL_LSTM) {
      o.cpv = d.c_prev ? d.c_prev[(long)b * d.c_prev_ld + j] : 0.f;
      o.cov = d.c_out[(long)b * d.c_out_ld + j];
      o.dcv = d.dc[(long)b * d.dc_ld + j];
    }
    if (CELL == CELL_GRU) o.hpv = d.h_prev ? d.h_prev[(long)b * d.h_prev_ld + j] : 0.f;
  }
}

template <int CELL>
__device__ __forceinline__ void bwd_finish(const Bwd2Dir& d, const uint32_t* seed, int b, int j, int H, bool linear, const BwdOperands& op,
                                           float sa, float sb) {
  constexpr int NSV = CELL == CELL_RNN ? 1 : 4;
  const bool m = op.m;
  // (recurrent dropout: `sa` arrives with the per-gate masks applied by back_partial; the GRU carry z * h_prev takes the unmasked
  // state - Keras implementation 1, which its cells use whenever recurrent_dropout != 0)
  const float carry = op.carry, cpv = op.cpv, cov = op.cov, dcv = op.dcv, hpv = op.hpv, addAv = op.addAv, addBv = op.addBv, dirv = op.dirv;
  const float* svv = op.svv;
  struct { const uint32_t* seed; } a{seed};
  if (d.src[1].D != nullptr && d.src[1].drop_rate > 0.f) {
    const BackSrc& s = d.src[1];
    const AsrRngKey key = asr_rng_key(a.seed[0], s.drop_stream);
    sb *= asr_drop_mult(key, (uint32_t)((long)b * s.drop_ld + s.drop_off + j), asr_drop_threshold(s.drop_rate), 1.f / (1.f - s.drop_rate));
  }
  const float dh_state = sa + addAv + dirv;
  const float dy = sb + addBv;
  if (linear) {
    d.out[(long)b * d.out_ld + j] = dh_state + dy;
    return;
  }
  float ds[4] = {0.f, 0.f, 0.f, 0.f};
  float dir = 0.f;
  if (!m) {
    dir = dh_state;                       // state carried unchanged through a masked step
    if (d.dy_carry) d.dy_carry[(long)b * d.dy_carry_ld + j] = carry + dy;
  } else {
    const float dh = dh_state + dy + carry;
    if (d.dy_carry) d.dy_carry[(long)b * d.dy_carry_ld + j] = 0.f;
    if (CELL == CELL_LSTM) {
      const float ig = svv[0], fg = svv[NSV > 1 ? 1 : 0], gg = svv[NSV > 2 ? 2 : 0], og = svv[NSV > 3 ? 3 : 0];
      const float tc = tanhf_(cov);
      const float dct = dcv + dh * og * (1.f - tc * tc);
      ds[0] = dct * gg * ig * (1.f - ig);
      ds[1] = dct * cpv * fg * (1.f - fg);
      ds[2] = dct * ig * (1.f - gg * gg);
      ds[3] = dh * tc * og * (1.f - og);
      d.dc[(long)b * d.dc_ld + j] = dct * fg;
    } else if (CELL == CELL_GRU) {
      const float z = svv[0], r = svv[NSV > 1 ? 1 : 0], hh = svv[NSV > 2 ? 2 : 0], arh = svv[NSV > 3 ? 3 : 0];
      const float dahh = dh * (1.f - z) * (1.f - hh * hh);
      ds[0] = dh * (hpv - hh) * z * (1.f - z);
      ds[1] = dahh * arh * r * (1.f - r);
      ds[2] = dahh;
      ds[3] = dahh * r;
      dir = dh * z;
    } else {
      const float hn = svv[0];
      ds[0] = dh * (1.f - hn * hn);
    }
  }
  float* o = d.dslots + (long)b * d.dslots_ld + j;
#pragma unroll
  for (int g = 0; g < NSV; ++g) o[(long)g * H] = ds[g];
  if (d.direct) d.direct[(long)b * d.direct_ld + j] = dir;
}

template <int CELL>
__global__ __launch_bounds__(64 * BW_NW) void rnn_step_bwd_kernel(Bwd2Args a) {
  __shared__ float part[2][BW_NW][256];
  const Bwd2Dir& d = a.d[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int unit0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int B = a.B, H = d.n_units;
  // thread (row, un) of the first 256 threads owns one (batch row, unit) pair for the gate math
  const int row = tid >> 4, un = tid & 15;
  const int b = b0 + row, j = unit0 + un;
  const bool owner = tid < 256 && b < B && j < H;
  const bool linear = d.out != nullptr;

  // operands of the element-wise part: loads issued before the matrix products
  BwdOperands op;
  if (owner) load_bwd_operands<CELL>(d, b, j, H, linear, op);

  const f32x4 pa = back_partial(d.src[0], b0, unit0, B, H, wave, li, lq);
  const f32x4 pb = back_partial(d.src[1], b0, unit0, B, H, wave, li, lq);
#pragma unroll
  for (int r = 0; r < 4; ++r) {        // C/D map: col = lane&15 (unit), row = 4*(lane>>4) + r (batch row)
    part[0][wave][(lq * 4 + r) * 16 + li] = pa[r];
    part[1][wave][(lq * 4 + r) * 16 + li] = pb[r];
  }
  __syncthreads();
  if (!owner) return;
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int w = 0; w < BW_NW; ++w) { sa += part[0][w][tid]; sb += part[1][w][tid]; }
  bwd_finish<CELL>(d, a.seed, b, j, H, linear, op, sa, sb);
}

// ------------------------------------------------------------------------------------------ wide backward step
// Wide cells with several batch tiles (the mirror of rnn_step_fwd_wide_kernel, rnn.hip): a workgroup owns NU x 16 units
// and NT x 16 batch rows, so a weight row is read once per NT tiles and a ds row once per NU tiles.  The column axis is
// split over the 16 waves as above; every wave leaves one 16 x 16 slab per (source, row tile, unit tile) in LDS.
// Measured on las_large (ms per training step, narrow kernel 225.9): NT,NU = 2,1: 219.6   1,2: 215.3 (chosen)
// 2,2: 259.3   4,1: 276.1 - as in the forward kernel, tiles that leave fewer than one workgroup per CU lose more
// to exposed latency than they save in L2 traffic.
// WBF = 1 (mixed precision): weights from the bf16 image, ds rounded to bf16 in registers, one v_mfma_f32_16x16x32_bf16 per
// pair of column blocks (see rnn_step_fwd_wide_kernel).
template <int NT, int NU, int WBF>
__device__ __forceinline__ void back_partial_wide(const BackSrc& s, int b0, int unit0, int B, int n_units, int wave, int li, int lq,
                                                  f32x4 (&acc)[NT][NU]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (s.D == nullptr) return;
  for (int g = 0; g < s.nseg; ++g) {
    const int len = s.len[g], nb = (len + 15) >> 4;
    // two column blocks in flight per wave: with 8 (bf16) / 4 (f32) las_large ran 226 / 317 ms per step against 186 / 309
    for (int j0 = wave; j0 < nb; j0 += BW_NW * 2) {
      float4 av[2][NT], bv[WBF ? 1 : 2][NU];
      uint2 bh[2][NU];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int jb = j0 + BW_NW * i, k = 16 * jb + 4 * lq;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int brow = b0 + 16 * t + li;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (jb < nb && brow < B) {
            const float* dr = s.D + (long)brow * s.ldd + s.d_col0[g];
            if (s.vec && k + 3 < len) v = *reinterpret_cast<const float4*>(dr + k);
            else { v.x = k < len ? dr[k] : 0.f; v.y = k + 1 < len ? dr[k + 1] : 0.f; v.z = k + 2 < len ? dr[k + 2] : 0.f; v.w = k + 3 < len ? dr[k + 3] : 0.f; }
          }
          av[i][t] = v;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int urow = unit0 + 16 * u + li;
          if (WBF) {
            uint2 h = make_uint2(0u, 0u);
            if (jb < nb && urow < n_units) {
              const unsigned short* wr = s.W16 + (long)urow * s.ldw + s.w_col0[g];
              if (s.vec && k + 3 < len) h = *reinterpret_cast<const uint2*>(wr + k);
              else {
                const unsigned w0 = k < len ? wr[k] : 0u, w1 = k + 1 < len ? wr[k + 1] : 0u, w2 = k + 2 < len ? wr[k + 2] : 0u, w3 = k + 3 < len ? wr[k + 3] : 0u;
                h = make_uint2(w0 | (w1 << 16), w2 | (w3 << 16));
              }
            }
            bh[i][u] = h;
          } else {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jb < nb && urow < n_units) {
              const float* wr = s.W + (long)urow * s.ldw + s.w_col0[g];
              if (s.vec && k + 3 < len) v = *reinterpret_cast<const float4*>(wr + k);
              else { v.x = k < len ? wr[k] : 0.f; v.y = k + 1 < len ? wr[k + 1] : 0.f; v.z = k + 2 < len ? wr[k + 2] : 0.f; v.w = k + 3 < len ? wr[k + 3] : 0.f; }
            }
            bv[i][u] = v;
          }
        }
      }
      if (WBF) {
        bf16x8 a8[NT], b8[NU];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          a8[t][0] = (__bf16)av[0][t].x; a8[t][1] = (__bf16)av[0][t].y; a8[t][2] = (__bf16)av[0][t].z; a8[t][3] = (__bf16)av[0][t].w;
          a8[t][4] = (__bf16)av[1][t].x; a8[t][5] = (__bf16)av[1][t].y; a8[t][6] = (__bf16)av[1][t].z; a8[t][7] = (__bf16)av[1][t].w;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const uint4 raw = make_uint4(bh[0][u].x, bh[0][u].y, bh[1][u].x, bh[1][u].y);
          b8[u] = __builtin_bit_cast(bf16x8, raw);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int u = 0; u < NU; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[t], b8[u], acc[t][u], 0, 0, 0);
      } else
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][t].x, bv[i][u].x, acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][t].y, bv[i][u].y, acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][t].z, bv[i][u].z, acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][t].w, bv[i][u].w, acc[t][u], 0, 0, 0);
          }
    }
  }
}

template <int CELL, int NT, int NU, int WBF>
__global__ __launch_bounds__(64 * BW_NW) void rnn_step_bwd_wide_kernel(Bwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) float bw_smem[];   // [2 sources][BW_NW][NT * NU][256]
  const Bwd2Dir& d = a.d[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int unit0 = blockIdx.x * 16 * NU, b0 = blockIdx.y * 16 * NT;
  const int B = a.B, H = d.n_units;
  const bool linear = d.out != nullptr;
  constexpr int NTHR = 64 * BW_NW, NPAIR = 256 * NT * NU, NP = (NPAIR + NTHR - 1) / NTHR;
  constexpr int SLAB = NT * NU * 256;
  const bool two = d.src[1].D != nullptr;

  // pair p -> (row tile t, unit tile u, row, unit): p = ((t * NU + u) * 16 + row) * 16 + un  (= the slab layout)
  BwdOperands op[NP];
  bool owner[NP];
#pragma unroll
  for (int r = 0; r < NP; ++r) {
    const int p = tid + NTHR * r;
    const int tile = p >> 8, row = (p >> 4) & 15, un = p & 15;
    const int b = b0 + 16 * (tile / NU) + row, j = unit0 + 16 * (tile % NU) + un;
    owner[r] = p < NPAIR && b < B && j < H;
    if (owner[r]) load_bwd_operands<CELL>(d, b, j, H, linear, op[r]);
  }

  f32x4 acc[NT][NU];
  back_partial_wide<NT, NU, WBF>(d.src[0], b0, unit0, B, H, wave, li, lq, acc);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) bw_smem[(long)wave * SLAB + (t * NU + u) * 256 + (lq * 4 + r) * 16 + li] = acc[t][u][r];
  if (two) {
    back_partial_wide<NT, NU, WBF>(d.src[1], b0, unit0, B, H, wave, li, lq, acc);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) bw_smem[(long)(BW_NW + wave) * SLAB + (t * NU + u) * 256 + (lq * 4 + r) * 16 + li] = acc[t][u][r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NP; ++r) {
    if (!owner[r]) continue;
    const int p = tid + NTHR * r;
    const int tile = p >> 8, row = (p >> 4) & 15, un = p & 15;
    const int b = b0 + 16 * (tile / NU) + row, j = unit0 + 16 * (tile % NU) + un;
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < BW_NW; ++w) sa += bw_smem[(long)w * SLAB + p];
    if (two) {
#pragma unroll
      for (int w = 0; w < BW_NW; ++w) sb += bw_smem[(long)(BW_NW + w) * SLAB + p];
    }
    bwd_finish<CELL>(d, a.seed, b, j, H, linear, op[r], sa, sb);
  }
}

// ------------------------------------------------------------------------------------------ staged backward step
// The wide kernel above loads every MFMA operand straight from global memory, one float4 per lane: a wave instruction
// then touches 16 different rows (64-byte pieces), and a wave owning 16 column blocks of a 4096-column contraction
// makes 8 dependent round trips per step.  Here the workgroup (32 units x 16 rows, 16 waves) walks the column axis in
// chunks of KC columns staged through LDS like a GEMM: coalesced 16-byte loads (a wave reads whole 1-KB row pieces),
// the next chunk prefetched into registers while the current one is multiplied, fragments read from padded LDS rows.
template <int WBF>
struct Staged {
  static constexpr int CW = WBF ? 32 : 16;          // columns per wave per chunk (one bf16 MFMA / four f32 MFMA steps)
  static constexpr int KC = BW_NW * CW;             // 512 / 256
  static constexpr int A_LD = KC + 4;               // floats: rows 4 banks apart -> conflict-free b128 reads
  static constexpr int B_LD = WBF ? KC + 8 : KC + 4;   // bf16 elements / floats
  static constexpr int A_FLOATS = 16 * A_LD;
  static constexpr int B_BYTES = 32 * B_LD * (WBF ? 2 : 4);
  static constexpr int A_V = 16 * KC / 4 / (64 * BW_NW);      // float4 per thread per chunk (2 / 1)
  static constexpr int B_V = WBF ? 32 * KC / 8 / (64 * BW_NW) : 32 * KC / 4 / (64 * BW_NW);   // 16-byte pieces per thread (2 / 2)
  static size_t smem_bytes(bool two) { return sizeof(float) * (two ? 2 : 1) * BW_NW * 2 * 256 + sizeof(float) * A_FLOATS + B_BYTES; }
};

template <int WBF>
__device__ __forceinline__ void back_partial_staged(const BackSrc& s, int b0, int unit0, int B, int n_units, float* As, void* Bsv,
                                                    f32x4 (&acc)[2]) {
  using G = Staged<WBF>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (s.D == nullptr) return;
  float* Bf = static_cast<float*>(Bsv);
  unsigned short* Bh = static_cast<unsigned short*>(Bsv);
  float4 ra[G::A_V];
  uint4 rb[G::B_V];
  auto gload = [&](int g, int k0) {
    const int len = s.len[g];
#pragma unroll
    for (int i = 0; i < G::A_V; ++i) {
      const int idx = tid + 64 * BW_NW * i, row = idx / (G::KC / 4), c = k0 + 4 * (idx % (G::KC / 4));
      ra[i] = (b0 + row < B && c < len) ? *reinterpret_cast<const float4*>(s.D + (long)(b0 + row) * s.ldd + s.d_col0[g] + c)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < G::B_V; ++i) {
      const int idx = tid + 64 * BW_NW * i;
      if (WBF) {
        const int un = idx / (G::KC / 8), c = k0 + 8 * (idx % (G::KC / 8));
        rb[i] = (unit0 + un < n_units && c < len) ? *reinterpret_cast<const uint4*>(s.W16 + (long)(unit0 + un) * s.ldw + s.w_col0[g] + c)
                                                  : make_uint4(0u, 0u, 0u, 0u);
      } else {
        const int un = idx / (G::KC / 4), c = k0 + 4 * (idx % (G::KC / 4));
        rb[i] = (unit0 + un < n_units && c < len) ? *reinterpret_cast<const uint4*>(s.W + (long)(unit0 + un) * s.ldw + s.w_col0[g] + c)
                                                  : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < G::A_V; ++i) {
      const int idx = tid + 64 * BW_NW * i, row = idx / (G::KC / 4), c4 = idx % (G::KC / 4);
      *reinterpret_cast<float4*>(As + row * G::A_LD + 4 * c4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < G::B_V; ++i) {
      const int idx = tid + 64 * BW_NW * i;
      if (WBF) {
        const int un = idx / (G::KC / 8), c8 = idx % (G::KC / 8);
        *reinterpret_cast<uint4*>(Bh + un * G::B_LD + 8 * c8) = rb[i];
      } else {
        const int un = idx / (G::KC / 4), c4 = idx % (G::KC / 4);
        *reinterpret_cast<uint4*>(Bf + un * G::B_LD + 4 * c4) = rb[i];
      }
    }
  };
  // chunk list: the segments one after the other, each in steps of KC columns
  int g = 0, k0 = 0;
  gload(g, k0);
  for (;;) {
    lstore();
    __syncthreads();
    int gn = g, kn = k0 + G::KC;
    if (kn >= s.len[g]) { gn = g + 1; kn = 0; }
    const bool more = gn < s.nseg;
    if (more) gload(gn, kn);
    if (WBF) {
      const float* ar = As + li * G::A_LD + 32 * wave + 8 * lq;
      const float4 x0 = *reinterpret_cast<const float4*>(ar), x1 = *reinterpret_cast<const float4*>(ar + 4);
      bf16x8 a8;
      a8[0] = (__bf16)x0.x; a8[1] = (__bf16)x0.y; a8[2] = (__bf16)x0.z; a8[3] = (__bf16)x0.w;
      a8[4] = (__bf16)x1.x; a8[5] = (__bf16)x1.y; a8[6] = (__bf16)x1.z; a8[7] = (__bf16)x1.w;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint4 raw = *reinterpret_cast<const uint4*>(Bh + (16 * u + li) * G::B_LD + 32 * wave + 8 * lq);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(bf16x8, raw), acc[u], 0, 0, 0);
      }
    } else {
      const float4 x = *reinterpret_cast<const float4*>(As + li * G::A_LD + 16 * wave + 4 * lq);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4 w = *reinterpret_cast<const float4*>(Bf + (16 * u + li) * G::B_LD + 16 * wave + 4 * lq);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, w.x, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, w.y, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, w.z, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, w.w, acc[u], 0, 0, 0);
      }
    }
    __syncthreads();
    if (!more) break;
    g = gn; k0 = kn;
  }
}

template <int CELL, int WBF>
__global__ __launch_bounds__(64 * BW_NW) void rnn_step_bwd_staged_kernel(Bwd2Args a) {
  using G = Staged<WBF>;
  extern __shared__ __attribute__((aligned(16))) float bw_smem[];   // [sources][BW_NW][2 tiles][256] slabs, then the A and B chunks
  const Bwd2Dir& d = a.d[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int unit0 = blockIdx.x * 32, b0 = blockIdx.y * 16;
  const int B = a.B, H = d.n_units;
  const bool linear = d.out != nullptr;
  bool two = false;                                              // the launch sizes LDS for the widest direction
  for (int z = 0; z < (int)gridDim.z; ++z) two = two || a.d[z].src[1].D != nullptr;
  constexpr int SLAB = 2 * 256;
  float* As = bw_smem + (two ? 2 : 1) * BW_NW * SLAB;
  void* Bs = As + G::A_FLOATS;

  // pair p (threads 0..511) -> (unit tile u, row, unit): p = (u * 16 + row) * 16 + un  (= the slab layout)
  const int p = tid, tile = p >> 8, row = (p >> 4) & 15, un = p & 15;
  const int b = b0 + row, j = unit0 + 16 * tile + un;
  const bool owner = p < 512 && b < B && j < H;
  BwdOperands op;
  if (owner) load_bwd_operands<CELL>(d, b, j, H, linear, op);

  f32x4 acc[2];
  back_partial_staged<WBF>(d.src[0], b0, unit0, B, H, As, Bs, acc);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) bw_smem[(long)wave * SLAB + u * 256 + (lq * 4 + r) * 16 + li] = acc[u][r];
  const bool mine = d.src[1].D != nullptr;
  if (two) {                                                     // (uniform over the launch: every workgroup takes the barriers inside)
    back_partial_staged<WBF>(d.src[1], b0, unit0, B, H, As, Bs, acc);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) bw_smem[(long)(BW_NW + wave) * SLAB + u * 256 + (lq * 4 + r) * 16 + li] = acc[u][r];
  }
  __syncthreads();
  if (!owner) return;
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int w = 0; w < BW_NW; ++w) sa += bw_smem[(long)w * SLAB + p];
  if (mine) {
#pragma unroll
    for (int w = 0; w < BW_NW; ++w) sb += bw_smem[(long)(BW_NW + w) * SLAB + p];
  }
  bwd_finish<CELL>(d, a.seed, b, j, H, linear, op, sa, sb);
}

// ------------------------------------------------------------------------------------------ host side
static inline int cell_nsaved(int cell) { return cell == CELL_RNN ? 1 : 4; }

static int fill_src(BackSrc* o, const asr_rnn_back_src* s) {
  *o = BackSrc{};
  if (!s->D) return ASR_OK;
  ASR_CHECK(s->W && s->nseg >= 1 && s->nseg <= 2, ASR_ERR_ARG, "rnn backward source: weights / segments missing");
  o->D = s->D; o->ldd = s->ldd; o->W = s->W; o->ldw = s->ldw; o->nseg = s->nseg;
  o->W16 = static_cast<const unsigned short*>(s->W16);
  bool vec = (((uintptr_t)s->D | (uintptr_t)s->W) & 15) == 0 && s->ldd % 4 == 0 && s->ldw % 4 == 0;
  for (int g = 0; g < s->nseg; ++g) {
    o->d_col0[g] = s->d_col0[g]; o->w_col0[g] = s->w_col0[g]; o->len[g] = s->len[g];
    vec = vec && s->d_col0[g] % 4 == 0 && s->w_col0[g] % 4 == 0;
  }
  o->vec = vec;
  o->drop_rate = s->drop_rate; o->drop_stream = s->drop_stream; o->drop_ld = s->drop_ld; o->drop_off = s->drop_off;
  return ASR_OK;
}

static int fill_dir(Bwd2Dir* d, const asr_rnn_step_bwd* s, int rnn_type, const uint32_t* seed) {
  *d = Bwd2Dir{};
  d->n_units = s->n_units;
  int rc = fill_src(&d->src[0], &s->srcA);
  if (rc) return rc;
  rc = fill_src(&d->src[1], &s->srcB);
  if (rc) return rc;
  ASR_CHECK(!((s->srcA.drop_rate > 0.f || s->srcB.drop_rate > 0.f) && !seed), ASR_ERR_ARG, "rnn backward: dropout needs a device seed");
  ASR_CHECK(!(s->srcA.D && s->srcA.drop_rate > 0.f), ASR_ERR_UNSUPPORTED, "rnn backward: dropout is only supported on source B");
  d->addA = s->addA; d->addA_ld = s->addA_ld; d->addB = s->addB; d->addB_ld = s->addB_ld;
  d->direct = s->direct; d->direct_ld = s->direct_ld; d->out = s->out; d->out_ld = s->out_ld;
  d->dc = s->dc; d->dc_ld = s->dc_ld; d->dy_carry = s->dy_carry; d->dy_carry_ld = s->dy_carry_ld;
  d->mask = s->mask; d->mask_ld = s->mask_ld; d->saved = s->saved; d->saved_ld = s->saved_ld;
  d->h_prev = s->h_prev; d->h_prev_ld = s->h_prev_ld; d->c_prev = s->c_prev; d->c_prev_ld = s->c_prev_ld;
  d->c_out = s->c_out; d->c_out_ld = s->c_out_ld; d->dslots = s->dslots; d->dslots_ld = s->dslots_ld;
  if (!s->out) {
    ASR_CHECK(s->saved && s->dslots, ASR_ERR_ARG, "rnn backward: saved/dslots missing");
    ASR_CHECK(rnn_type != CELL_LSTM || (s->dc && s->c_out), ASR_ERR_ARG, "rnn backward: LSTM needs dc and c_out");
  }
  return ASR_OK;
}

template <int NT, int NU>
static void launch_bwd_wide(int rnn_type, const Bwd2Args& a, int ndir, int nu, bool two, hipStream_t st) {
  dim3 grid((unsigned)asr_cdiv(nu, 16 * NU), (unsigned)asr_cdiv(a.B, 16 * NT), (unsigned)ndir);
  const size_t smem = sizeof(float) * (two ? 2 : 1) * BW_NW * NT * NU * 256;
  auto go = [&](auto kern) {
    static unsigned long long attr = 0;
    if (asr_first_use_on_device(attr)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    hipLaunchKernelGGL(kern, grid, dim3(64 * BW_NW), smem, st, a);
  };
  bool bf = true;                                                // every used source carries a bf16 image of its weights
  for (int i = 0; i < ndir; ++i)
    for (int k = 0; k < 2; ++k) bf = bf && (a.d[i].src[k].D == nullptr || a.d[i].src[k].W16 != nullptr);
  if (bf) {
    if (rnn_type == CELL_LSTM) go(rnn_step_bwd_wide_kernel<CELL_LSTM, NT, NU, 1>);
    else if (rnn_type == CELL_GRU) go(rnn_step_bwd_wide_kernel<CELL_GRU, NT, NU, 1>);
    else go(rnn_step_bwd_wide_kernel<CELL_RNN, NT, NU, 1>);
    return;
  }
  if (rnn_type == CELL_LSTM) go(rnn_step_bwd_wide_kernel<CELL_LSTM, NT, NU, 0>);
  else if (rnn_type == CELL_GRU) go(rnn_step_bwd_wide_kernel<CELL_GRU, NT, NU, 0>);
  else go(rnn_step_bwd_wide_kernel<CELL_RNN, NT, NU, 0>);
}

static int launch_bwd(int rnn_type, const Bwd2Args& a, int ndir, hipStream_t st) {
  int nu = a.d[0].n_units;
  for (int i = 1; i < ndir; ++i) nu = a.d[i].n_units > nu ? a.d[i].n_units : nu;
  // wide cells with several batch tiles: 32 units x 16 rows per workgroup (rnn_step_bwd_wide_kernel); ASR_RNN_WIDE=0 turns it off
  static const int wide = getenv("ASR_RNN_WIDE") ? atoi(getenv("ASR_RNN_WIDE")) : 1;
  static const int min_h = getenv("ASR_RNN_WIDE_MIN_H") ? atoi(getenv("ASR_RNN_WIDE_MIN_H")) : 512;   // tests lower it
  // ... and long rows to contract over (las_small's d context step has 512 units but only 1024 columns: narrow is faster there)
  int cols = 0;
  for (int i = 0; i < ndir; ++i)
    for (int k = 0; k < 2; ++k) {
      int c = 0;
      for (int g = 0; g < a.d[i].src[k].nseg; ++g) c += a.d[i].src[k].D ? a.d[i].src[k].len[g] : 0;
      cols = c > cols ? c : cols;
    }
  bool rmult = false;                                            // recurrent dropout: narrow kernel only
  for (int i = 0; i < ndir; ++i) rmult = rmult || a.d[i].src[0].oscale[0] != nullptr;
  if (wide && !rmult && nu >= min_h && a.B > 16 && (cols >= 2048 || min_h < 512)) {
    bool two = false;
    for (int i = 0; i < ndir; ++i) two = two || a.d[i].src[1].D != nullptr;
    static const int staged = getenv("ASR_RNN_STAGED") ? atoi(getenv("ASR_RNN_STAGED")) : 1;
    bool bf = true, ok = staged != 0;
    for (int i = 0; i < ndir; ++i)
      for (int k = 0; k < 2; ++k) {
        const BackSrc& sc = a.d[i].src[k];
        if (!sc.D) continue;
        bf = bf && sc.W16 != nullptr;
        ok = ok && sc.vec;
        for (int g = 0; g < sc.nseg; ++g) ok = ok && sc.len[g] % 4 == 0;
      }
    for (int i = 0; i < ndir && ok && bf; ++i)
      for (int k = 0; k < 2; ++k) {
        const BackSrc& sc = a.d[i].src[k];
        if (!sc.D) continue;
        ok = ok && sc.ldw % 8 == 0 && (((uintptr_t)sc.W16) & 15) == 0;
        for (int g = 0; g < sc.nseg; ++g) ok = ok && sc.w_col0[g] % 8 == 0 && sc.len[g] % 8 == 0;
      }
    if (ok) {
      dim3 grid((unsigned)asr_cdiv(nu, 32), (unsigned)asr_cdiv(a.B, 16), (unsigned)ndir);
      auto go = [&](auto kern, size_t smem) {
        static unsigned long long attr = 0;
        if (asr_first_use_on_device(attr)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        hipLaunchKernelGGL(kern, grid, dim3(64 * BW_NW), smem, st, a);
      };
      if (bf) {
        const size_t smem = Staged<1>::smem_bytes(two);
        if (rnn_type == CELL_LSTM) go(rnn_step_bwd_staged_kernel<CELL_LSTM, 1>, smem);
        else if (rnn_type == CELL_GRU) go(rnn_step_bwd_staged_kernel<CELL_GRU, 1>, smem);
        else go(rnn_step_bwd_staged_kernel<CELL_RNN, 1>, smem);
      } else {
        const size_t smem = Staged<0>::smem_bytes(two);
        if (rnn_type == CELL_LSTM) go(rnn_step_bwd_staged_kernel<CELL_LSTM, 0>, smem);
        else if (rnn_type == CELL_GRU) go(rnn_step_bwd_staged_kernel<CELL_GRU, 0>, smem);
        else go(rnn_step_bwd_staged_kernel<CELL_RNN, 0>, smem);
      }
      ASR_LAUNCH_CHECK();
      return ASR_OK;
    }
    launch_bwd_wide<1, 2>(rnn_type, a, ndir, nu, two, st);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
  }
  dim3 grid((unsigned)asr_cdiv(nu, 16), (unsigned)asr_cdiv(a.B, 16), (unsigned)ndir);
  dim3 block(64 * BW_NW);
  if (rnn_type == CELL_LSTM) hipLaunchKernelGGL(rnn_step_bwd_kernel<CELL_LSTM>, grid, block, 0, st, a);
  else if (rnn_type == CELL_GRU) hipLaunchKernelGGL(rnn_step_bwd_kernel<CELL_GRU>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(rnn_step_bwd_kernel<CELL_RNN>, grid, block, 0, st, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_rnn_cell_bwd(int rnn_type, int B, int ndir, const asr_rnn_step_bwd* steps, const uint32_t* seed, void* stream) {
  ASR_CHECK(steps, ASR_ERR_ARG, "asr_rnn_cell_bwd: null argument");
  ASR_CHECK(rnn_type >= 0 && rnn_type <= 2, ASR_ERR_UNSUPPORTED, "rnn_type: %d is invalid!", rnn_type);
  ASR_CHECK(B > 0 && (ndir == 1 || ndir == 2), ASR_ERR_SHAPE, "asr_rnn_cell_bwd: B %d ndir %d", B, ndir);
  Bwd2Args a{};
  a.B = B; a.seed = seed;
  for (int i = 0; i < ndir; ++i) {
    ASR_CHECK(steps[i].n_units > 0, ASR_ERR_SHAPE, "asr_rnn_cell_bwd: n_units must be > 0");
    int rc = fill_dir(&a.d[i], &steps[i], rnn_type, seed);
    if (rc) return rc;
  }
  return launch_bwd(rnn_type, a, ndir, (hipStream_t)stream);
}

// column segments of the saved/dslots buffer that multiply the recurrent kernel U [H, G*H]
static void rec_segments(int rnn_type, int H, asr_rnn_back_src* s) {
  if (rnn_type == CELL_GRU) {
    s->nseg = 2;
    s->d_col0[0] = 0; s->w_col0[0] = 0; s->len[0] = 2 * H;
    s->d_col0[1] = 3 * H; s->w_col0[1] = 2 * H; s->len[1] = H;
  } else {
    s->nseg = 1;
    s->d_col0[0] = 0; s->w_col0[0] = 0; s->len[0] = (rnn_type == CELL_LSTM ? 4 : 1) * H;
  }
}

// Whole BiRNN layer backward-through-time.  On return saved[d] holds ds (gradient wrt the gate sums)
// for the batched dW / dU / dX GEMMs, dh0[d] the gradient wrt the initial h, dc[d] wrt the initial c.
extern "C" int asr_rnn_seq_bwd(const asr_rnn_seq* s, const asr_rnn_seq_grad* gs, void* stream) {
  ASR_CHECK(s && gs, ASR_ERR_ARG, "asr_rnn_seq_bwd: null argument");
  ASR_CHECK(s->rnn_type >= 0 && s->rnn_type <= 2, ASR_ERR_UNSUPPORTED, "rnn_type: %d is invalid!", s->rnn_type);
  const int B = s->B, T = s->T, H = s->H;
  const int NS = cell_nsaved(s->rnn_type), NG = s->rnn_type == CELL_LSTM ? 4 : (s->rnn_type == CELL_GRU ? 3 : 1);
  const bool lstm = s->rnn_type == CELL_LSTM;
  hipStream_t st = (hipStream_t)stream;
  for (int d = 0; d < s->ndir; ++d) {
    ASR_CHECK(s->saved[d] && s->U[d] && gs->direct[d] && gs->dy, ASR_ERR_ARG, "asr_rnn_seq_bwd: null buffer (dir %d)", d);
    ASR_CHECK(!lstm || gs->dc[d], ASR_ERR_ARG, "asr_rnn_seq_bwd: LSTM needs a dc buffer (dir %d)", d);
    ASR_CHECK(!s->mask || gs->dy_carry[d], ASR_ERR_ARG, "asr_rnn_seq_bwd: masked sequences need dy_carry (dir %d)", d);
    if (asr_zero_async(gs->direct[d], sizeof(float) * (size_t)B * H, st) != hipSuccess) { asr_set_error("asr_rnn_seq_bwd: memset failed"); return ASR_ERR_HIP; }
  }
  const long ld_s = (long)T * NS * H;
  for (int step = T - 1; step >= -1; --step) {   // step == -1: gradient wrt the initial h (linear mode)
    Bwd2Args a{};
    a.B = B; a.seed = nullptr;
    bool any = false;
    for (int d = 0; d < s->ndir; ++d) {
      const bool rev = s->reverse[d] != 0;
      asr_rnn_step_bwd sb{};
      sb.n_units = H;
      const int tn_step = step + 1;                         // the step processed just before (in backward order)
      if (tn_step <= T - 1) {
        const int tn = rev ? T - 1 - tn_step : tn_step;
        sb.srcA.D = s->saved[d] + (long)tn * NS * H; sb.srcA.ldd = ld_s;
        sb.srcA.W = s->U[d]; sb.srcA.ldw = s->ldu[d] ? s->ldu[d] : (long)NG * H; sb.srcA.W16 = s->U16[d];
        rec_segments(s->rnn_type, H, &sb.srcA);
      } else {
        sb.addA = gs->dh_last[d]; sb.addA_ld = gs->dh_last_ld[d];
      }
      sb.direct = gs->direct[d]; sb.direct_ld = H;
      if (step >= 0) {
        const int t = rev ? T - 1 - step : step;
        const int tp = rev ? t + 1 : t - 1;
        sb.addB = gs->dy + (long)t * gs->dy_ld + s->y_col[d]; sb.addB_ld = (long)T * gs->dy_ld;
        sb.dc = lstm ? gs->dc[d] : nullptr; sb.dc_ld = H;
        sb.dy_carry = s->mask ? gs->dy_carry[d] : nullptr; sb.dy_carry_ld = H;
        sb.mask = s->mask ? s->mask + t : nullptr; sb.mask_ld = T;
        sb.saved = s->saved[d] + (long)t * NS * H; sb.saved_ld = ld_s;
        if (step == 0) {
          sb.h_prev = s->h0[d]; sb.h_prev_ld = s->h0_ld[d];
          sb.c_prev = lstm ? s->c0[d] : nullptr; sb.c_prev_ld = s->c0_ld[d];
        } else {
          sb.h_prev = s->hseq[d] + (long)tp * H; sb.h_prev_ld = (long)T * H;
          sb.c_prev = lstm ? s->cseq[d] + (long)tp * H : nullptr; sb.c_prev_ld = (long)T * H;
        }
        sb.c_out = lstm ? s->cseq[d] + (long)t * H : nullptr; sb.c_out_ld = (long)T * H;
        sb.dslots = s->saved[d] + (long)t * NS * H; sb.dslots_ld = ld_s;
      } else {
        if (!gs->dh0[d]) { sb.n_units = 0; }
        sb.out = gs->dh0[d]; sb.out_ld = gs->dh0_ld[d];
      }
      if (sb.n_units > 0) {
        int rc = fill_dir(&a.d[d], &sb, s->rnn_type, nullptr);
        if (rc) return rc;
        if (s->rec_mult[d] && a.d[d].src[0].D) {
          // recurrent dropout: one mask per gate ([NG][B,H] tables) on the state the consumer read - one column segment per gate
          BackSrc& r = a.d[d].src[0];
          const long gsz = (long)B * H;
          if (s->rnn_type == CELL_GRU) {                     // ds slots z, r, (x-part), h~ ; recurrent kernel columns z, r, h~
            r.nseg = 3;
            r.d_col0[0] = 0; r.w_col0[0] = 0; r.d_col0[1] = H; r.w_col0[1] = H; r.d_col0[2] = 3 * H; r.w_col0[2] = 2 * H;
          } else {
            r.nseg = NG;
            for (int g = 0; g < NG; ++g) { r.d_col0[g] = g * H; r.w_col0[g] = g * H; }
          }
          for (int g = 0; g < r.nseg; ++g) { r.len[g] = H; r.oscale[g] = s->rec_mult[d] + g * gsz; }
          r.oscale_ld = H;
          r.vec = r.vec && H % 4 == 0;
        }
        any = true;
      } else {
        a.d[d] = Bwd2Dir{};   // n_units == 0: every thread of that direction exits as a non-owner
      }
    }
    if (!any) continue;
    int rc = launch_bwd(s->rnn_type, a, s->ndir, st);
    if (rc) return rc;
  }
  return ASR_OK;
}
