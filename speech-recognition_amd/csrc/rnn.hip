// Recurrent cells for gfx950: Keras LSTM / GRU(reset_after) / SimpleRNN with K.rnn mask semantics
// (las.py:62-126 BiRNN, las.py:259-262,285-288 decoder cells, deepspeech2.py:109-119).
//
// This file: the ONE-LAUNCH-PER-TIME-STEP forward kernels (both directions of a BiRNN ride in
// blockIdx.z) - used for the LAS decoder cells, and for whole layers whose shape the persistent
// one-launch-per-layer kernel (rnn_persist.hip) does not take.  The step sequence is meant to be captured
// in a hipGraph by the caller.
//
// Work split: workgroup (q, bt, dir) owns hidden units [4q, 4q+4) x batch rows [16bt, 16bt+16).
// The per-step product  [16 x Ktot] x [Ktot x 16 gate columns]  runs on v_mfma_f32_16x16x4_f32 with
// K split across the 4 waves and reduced through LDS; wave 0 then does the gate math.  Weights are
// re-packed once per optimizer step into MFMA fragment order (asr_rnn_pack) so every B-operand
// load is one coalesced 256-byte wave access of an L2-resident 16 KB slice.
// (Wide cells with several batch tiles take rnn_step_fwd_wide_kernel below instead.)
//
// "slot" = one of the 16 packed gate columns: slot = 4*s + u, u = unit within the group,
//   LSTM s = gate i,f,c~,o        GRU s = z, r, x-part of h~, recurrent part of h~        RNN s = 0 only.
//
// The backward step kernels live in rnn_bwd.hip (and rnn_persist_bwd.hip for whole layers).
#include <stdlib.h>

#include "common.h"

#define CELL_LSTM 0
#define CELL_GRU 1
#define CELL_RNN 2

static __host__ __device__ inline int cell_ngates(int cell) { return cell == CELL_LSTM ? 4 : (cell == CELL_GRU ? 3 : 1); }
static __host__ __device__ inline int cell_nsaved(int cell) { return cell == CELL_RNN ? 1 : 4; }

// ------------------------------------------------------------------------------------------ pack
// The packed K axis is the concatenation of the segments, each padded to a multiple of 16 ("blocks").
// Forward image  Wp [Q][KB][64 lanes][4]: lane (lq = lane>>4, slot = lane&15), element e holds
//   W[k = 16*blk + 4*lq + e][slot] - so one float4 per lane feeds 4 consecutive MFMA 16x16x4 steps and
//   matches a float4 of the activation row loaded by the same lane (the k labelling inside a block is
//   free as long as A and B agree).
struct PackArgs {
  const float* W[ASR_RNN_MAXSEG];
  long ldw[ASR_RNN_MAXSEG];
  int K[ASR_RNN_MAXSEG];
  int ks0[ASR_RNN_MAXSEG];
  int is_rec[ASR_RNN_MAXSEG];
  int nseg, KSt, NT, H, Q, cell;
  float* Wp;
};

__device__ __forceinline__ float pack_value(const PackArgs& a, int q, int kcol, int slot) {
  // kcol indexes the packed (segment-padded) K axis
  const int blk = kcol >> 4;
  int s = -1;
  for (int i = 0; i < a.nseg; ++i)
    if (blk >= a.ks0[i] && blk < a.ks0[i] + (a.K[i] + 15) / 16) s = i;
  if (s < 0) return 0.f;
  const int k = kcol - 16 * a.ks0[s];
  const int g = slot >> 2, u = slot & 3, j = 4 * q + u;
  if (k >= a.K[s] || j >= a.H) return 0.f;
  const float* W = a.W[s];
  const long ld = a.ldw[s];
  if (a.cell == CELL_LSTM) return W[k * ld + (long)g * a.H + j];
  if (a.cell == CELL_GRU) {
    if (g < 2) return W[k * ld + (long)g * a.H + j];
    if (g == 2) return a.is_rec[s] ? 0.f : W[k * ld + 2L * a.H + j];
    return a.is_rec[s] ? W[k * ld + 2L * a.H + j] : 0.f;
  }
  return g == 0 ? W[k * ld + j] : 0.f;
}

__global__ void rnn_pack_kernel(PackArgs a) {
  const long nf = (long)a.Q * a.KSt * 256;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const long r = i >> 8;
    const int blk = (int)(r % a.KSt), q = (int)(r / a.KSt);
    a.Wp[i] = pack_value(a, q, 16 * blk + 4 * (lane >> 4) + e, lane & 15);
  }
}

// every cell of a model in one launch (blockIdx.y = cell): the images are refreshed once per optimizer step, and eight 5-us
// launches of a few hundred KB each are mostly launch latency
#define ASR_PACK_MANY 12
struct PackManyArgs { PackArgs c[ASR_PACK_MANY]; };
__global__ void rnn_pack_many_kernel(PackManyArgs m) {
  const PackArgs& a = m.c[blockIdx.y];
  const long nf = (long)a.Q * a.KSt * 256;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const long r = i >> 8;
    const int blk = (int)(r % a.KSt), q = (int)(r / a.KSt);
    a.Wp[i] = pack_value(a, q, 16 * blk + 4 * (lane >> 4) + e, lane & 15);
  }
}

// ------------------------------------------------------------------------------------------ forward step
struct FwdSeg {
  const float* x; long ld; int K; int ks0; int vec;
  int drop; uint32_t drop_stream; float drop_rate; long drop_ld; int drop_off;
  const float* mult; long mult_ld;      // optional multiplier tables [slots][B, K] on this segment (Keras recurrent dropout: constant over time)
  long mult_gs;                         // floats between the tables of two gates: tf.keras switches its cells to implementation 1 whenever
                                        // recurrent_dropout != 0 - ONE MASK PER GATE on h_tm1 - so the product is taken once per gate slot
};
struct FwdDir {
  FwdSeg seg[ASR_RNN_MAXSEG];
  int nseg, KSt;
  const float* Wp;
  const unsigned short* Wp16;   // bf16 image of Wp or NULL
  const float* pre; long pre_ld;
  const float* bias; const float* bias_rec;
  const float* h_prev; long h_prev_ld;
  const float* c_prev; long c_prev_ld;
  const float* y_prev; long y_prev_ld;
  const uint8_t* mask; long mask_ld;
  float* h_out; long h_out_ld;
  float* c_out; long c_out_ld;
  float* y_out; long y_out_ld;
  float* saved; long saved_ld;
};
struct FwdArgs { FwdDir d[2]; int B, H; const uint32_t* seed; };

#ifndef RNN_CH
#define RNN_CH 4
#endif  // 16-wide K blocks a wave keeps in flight (loads first, then the MFMAs)

// Gate math of one (batch row b, hidden unit j) pair from the pre-activations `pre` (+ biases), the recurrent
// sums s[4] (slot order of the packed weights) and the previous state; writes h / c / y / saved.
template <int CELL>
__device__ __forceinline__ void cell_finish(const FwdDir& d, int H, int b, int j, bool m, float hp, float hpm, float yp, float cp,
                                            const float* pre, const float* br, const float* s) {
  // hpm = hp times the recurrent-dropout multiplier (= hp without recurrent dropout): [TF-sem] GRUCell rebinds h_tm1 to the
  // masked value, so the z * h_tm1 carry sees the multiplier; the state carried through a masked step does not
  constexpr int NSV = CELL == CELL_RNN ? 1 : 4;
  float hn, c2, sv[4];
  float prel[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int NGL = CELL == CELL_LSTM ? 4 : (CELL == CELL_GRU ? 3 : 1);
#pragma unroll
  for (int g = 0; g < NGL; ++g) prel[g] = pre[g];
  asr_cell_forward<CELL>(prel, s, br, hpm, cp, hn, c2, sv);
  if (d.saved) {
    float* o = d.saved + (long)b * d.saved_ld + j;
#pragma unroll
    for (int g = 0; g < NSV; ++g) o[(long)g * H] = sv[g];
  }
  if (CELL == CELL_LSTM && d.c_out) d.c_out[(long)b * d.c_out_ld + j] = m ? c2 : cp;
  if (d.h_out) d.h_out[(long)b * d.h_out_ld + j] = m ? hn : hp;
  if (d.y_out) d.y_out[(long)b * d.y_out_ld + j] = m ? hn : yp;
}

template <int CELL>
__global__ __launch_bounds__(256) void rnn_step_fwd_kernel(FwdArgs a) {
  __shared__ float part[4][16 * 17];
  const FwdDir& d = a.d[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int q = blockIdx.x, b0 = blockIdx.y * 16;
  const int H = a.H, B = a.B;

  // wave 0 owns the gate math of the 16 x 4 (row, unit) pairs: fetch its operands now, so their
  // latency hides under the matrix product instead of following the barrier
  const int bi = lane >> 2, u = lane & 3;
  const int b = b0 + bi, j = 4 * q + u;
  const bool live = b < B && j < H;
  constexpr int NG = CELL == CELL_LSTM ? 4 : (CELL == CELL_GRU ? 3 : 1);
  bool m = true;
  float hp = 0.f, yp = 0.f, cp = 0.f, pre[NG], br[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < NG; ++g) pre[g] = 0.f;
  if (wave == 0 && live) {
    m = d.mask ? d.mask[(long)b * d.mask_ld] != 0 : true;
    hp = d.h_prev ? d.h_prev[(long)b * d.h_prev_ld + j] : 0.f;
    yp = d.y_prev ? d.y_prev[(long)b * d.y_prev_ld + j] : 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float v = d.pre ? d.pre[(long)b * d.pre_ld + (long)g * H + j] : 0.f;
      if (d.bias) v += d.bias[(long)g * H + j];
      pre[g] = v;
    }
    if (CELL == CELL_LSTM) cp = d.c_prev ? d.c_prev[(long)b * d.c_prev_ld + j] : 0.f;
    if (CELL == CELL_GRU && d.bias_rec) { br[0] = d.bias_rec[j]; br[1] = d.bias_rec[H + j]; br[2] = d.bias_rec[2L * H + j]; }
  }
  const float hpm = hp;                 // (Keras implementation 1: the GRU carry z * h_tm1 takes the UNMASKED state)

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float4* wp = reinterpret_cast<const float4*>(d.Wp) + (long)q * d.KSt * 64 + lane;
  const int brow = b0 + li;
  const uint32_t seedv = a.seed ? a.seed[0] : 0u;
  for (int s = 0; s < d.nseg; ++s) {
    const FwdSeg& sg = d.seg[s];
    const int nb = (sg.K + 15) >> 4;
    const float* xr = sg.x ? sg.x + (long)brow * sg.ld : nullptr;
    const bool rowok = sg.x != nullptr && brow < B;
    for (int j0 = wave; j0 < nb; j0 += 4 * RNN_CH) {
      float4 av[RNN_CH], bv[RNN_CH];
#pragma unroll
      for (int i = 0; i < RNN_CH; ++i) {
        const int jb = j0 + 4 * i;
        av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (jb < nb) {
          const int k = 16 * jb + 4 * lq;
          bv[i] = wp[(long)(sg.ks0 + jb) * 64];
          if (rowok) {
            if (sg.vec && k + 3 < sg.K) av[i] = *reinterpret_cast<const float4*>(xr + k);
            else {
              av[i].x = k < sg.K ? xr[k] : 0.f;
              av[i].y = k + 1 < sg.K ? xr[k + 1] : 0.f;
              av[i].z = k + 2 < sg.K ? xr[k + 2] : 0.f;
              av[i].w = k + 3 < sg.K ? xr[k + 3] : 0.f;
            }
          }
        }
      }
      if (sg.drop) {
        const AsrRngKey key = asr_rng_key(seedv, sg.drop_stream);
        const uint32_t thr = asr_drop_threshold(sg.drop_rate);
        const float dscale = 1.f / (1.f - sg.drop_rate);
#pragma unroll
        for (int i = 0; i < RNN_CH; ++i) {
          const uint32_t idx = (uint32_t)((long)brow * sg.drop_ld + sg.drop_off + 16 * (j0 + 4 * i) + 4 * lq);
          av[i].x *= asr_drop_mult(key, idx, thr, dscale);
          av[i].y *= asr_drop_mult(key, idx + 1, thr, dscale);
          av[i].z *= asr_drop_mult(key, idx + 2, thr, dscale);
          av[i].w *= asr_drop_mult(key, idx + 3, thr, dscale);
        }
      }
      if (sg.mult) {
        // per-gate masks: the packed columns of a tile are (slot, unit) = (li >> 2, li & 3); the A operand of an MFMA cannot depend
        // on the output column, so the product runs once per slot with that slot's mask on h and every lane keeps the result of
        // ITS column's slot.  Mask index of a packed slot: LSTM i, f, c~, o -> 0..3; GRU z, r, (x-part: no recurrent term), h~ -> 0, 1, -, 2
        constexpr int NSLOT = CELL == CELL_RNN ? 1 : 4;
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
          if (CELL == CELL_GRU && sl == 2) continue;
          const int mi = CELL == CELL_GRU && sl == 3 ? 2 : sl;
          const float* mr = sg.mult + (long)mi * sg.mult_gs + (long)brow * sg.mult_ld;
          f32x4 accg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < RNN_CH; ++i) {
            const int k = 16 * (j0 + 4 * i) + 4 * lq;
            float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rowok) {
              if (k < sg.K) m4.x = mr[k];
              if (k + 1 < sg.K) m4.y = mr[k + 1];
              if (k + 2 < sg.K) m4.z = mr[k + 2];
              if (k + 3 < sg.K) m4.w = mr[k + 3];
            }
            accg = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x * m4.x, bv[i].x, accg, 0, 0, 0);
            accg = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y * m4.y, bv[i].y, accg, 0, 0, 0);
            accg = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z * m4.z, bv[i].z, accg, 0, 0, 0);
            accg = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w * m4.w, bv[i].w, accg, 0, 0, 0);
          }
          if ((li >> 2) == sl || NSLOT == 1) acc += accg;
        }
      } else {
#pragma unroll
      for (int i = 0; i < RNN_CH; ++i) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bv[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bv[i].w, acc, 0, 0, 0);
      }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave][(lq * 4 + r) * 17 + li] = acc[r];
  __syncthreads();
  if (wave != 0) return;
  if (!live) return;
  float s[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
    s[g] = part[0][bi * 17 + g * 4 + u] + part[1][bi * 17 + g * 4 + u] + part[2][bi * 17 + g * 4 + u] + part[3][bi * 17 + g * 4 + u];
  cell_finish<CELL>(d, H, b, j, m, hp, hpm, yp, cp, pre, br, s);
}

// ------------------------------------------------------------------------------------------ wide forward step
// Wide cells (H >= 512) with more than one batch tile (LAS-large: H = 1024, B = 64).  The narrow kernel above re-reads
// every packed weight slice once per batch tile and the whole state once per 4 units: 256 MB of L2 -> CU traffic per step at
// H = 1024, B = 64, which is what bounds it (36.9 us per step = 6.9 TB/s).  Here a workgroup owns WD_NQ weight slices
// (4 * WD_NQ units) x NT batch tiles, K split over NW waves and reduced through LDS: 96 MB per step at NT = 2, WD_NQ = 4.
// Occupancy matters as much as traffic - measured on las_large (ms per training step, narrow 255.3):
//   NT,NQ,NW = 4,2,4: 276   2,2,4: 245.6   2,2,8: 232.2   2,2,16: 232.2   1,2,8: 237.9   4,2,16: 238.3   1,4,16: 227.4
//   2,4,16: 225.5 (chosen: forward step 36.9 -> 23.6 us).  Summing the waves' partials with LDS atomics instead of a
//   slab per wave was far slower (319).  An LDS-staged variant in the style of rnn_step_bwd_staged_kernel (8 waves, one
//   output tile each over the whole K axis, activations staged once, no cross-wave sum) was slower too (161.5 -> 191 ms
//   mixed): with the weights already in fragment order the loads were never the problem here, and 8 waves per CU
//   stream them with too little in flight.
#define WD_CH 2

// WBF = 1 (mixed precision): the weights come from their bf16 image and the inputs are rounded to bf16 in registers; one
// v_mfma_f32_16x16x32_bf16 then covers the two 16-wide K blocks a wave has in flight (the k labelling inside an MFMA
// is free as long as both operands agree: elements 0-3 of a lane's fragment are its float4 of the first block,
// elements 4-7 that of the second) - an eighth of the matrix-pipe time and half the weight bytes.
template <int CELL, int NT, int WD_NQ, int NW, int WBF>
__global__ __launch_bounds__(64 * NW) void rnn_step_fwd_wide_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wd_smem[];
  float(*part)[NT][WD_NQ][16 * 17] = reinterpret_cast<float(*)[NT][WD_NQ][16 * 17]>(wd_smem);
  const FwdDir& d = a.d[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int H = a.H, B = a.B;
  const int Q = (H + 3) >> 2, q0 = blockIdx.x * WD_NQ;
  const int row0 = blockIdx.y * NT * 16;
  constexpr int NG = CELL == CELL_LSTM ? 4 : (CELL == CELL_GRU ? 3 : 1);
  constexpr int NTHR = 64 * NW;
  constexpr int NP = (NT * 16 * 4 * WD_NQ + NTHR - 1) / NTHR;    // (row, unit) pairs per thread

  bool live[NP], m[NP];
  float hp[NP], yp[NP], cp[NP], pre[NP][NG], br[NP][3];
#pragma unroll
  for (int r = 0; r < NP; ++r) {
    const int p = tid + NTHR * r;
    const int bl = p / (4 * WD_NQ), b = row0 + bl, uu = p % (4 * WD_NQ), j = 4 * (q0 + (uu >> 2)) + (uu & 3);
    live[r] = b < B && bl < NT * 16 && j < H;
    m[r] = true; hp[r] = 0.f; yp[r] = 0.f; cp[r] = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) pre[r][g] = 0.f;
    br[r][0] = br[r][1] = br[r][2] = 0.f;
    if (live[r]) {
      m[r] = d.mask ? d.mask[(long)b * d.mask_ld] != 0 : true;
      hp[r] = d.h_prev ? d.h_prev[(long)b * d.h_prev_ld + j] : 0.f;
      yp[r] = d.y_prev ? d.y_prev[(long)b * d.y_prev_ld + j] : 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        float v = d.pre ? d.pre[(long)b * d.pre_ld + (long)g * H + j] : 0.f;
        if (d.bias) v += d.bias[(long)g * H + j];
        pre[r][g] = v;
      }
      if (CELL == CELL_LSTM) cp[r] = d.c_prev ? d.c_prev[(long)b * d.c_prev_ld + j] : 0.f;
      if (CELL == CELL_GRU && d.bias_rec) { br[r][0] = d.bias_rec[j]; br[r][1] = d.bias_rec[H + j]; br[r][2] = d.bias_rec[2L * H + j]; }
    }
  }

  f32x4 acc[NT][WD_NQ];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int qi = 0; qi < WD_NQ; ++qi) acc[t][qi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float4* wp[WD_NQ];
  const uint2* wp16[WD_NQ];
#pragma unroll
  for (int qi = 0; qi < WD_NQ; ++qi) {
    wp[qi] = reinterpret_cast<const float4*>(d.Wp) + (long)min(q0 + qi, Q - 1) * d.KSt * 64 + lane;
    wp16[qi] = reinterpret_cast<const uint2*>(d.Wp16) + (long)min(q0 + qi, Q - 1) * d.KSt * 64 + lane;
  }
  const uint32_t seedv = a.seed ? a.seed[0] : 0u;
  for (int s = 0; s < d.nseg; ++s) {
    const FwdSeg& sg = d.seg[s];
    const int nb = (sg.K + 15) >> 4;
    for (int j0 = wave; j0 < nb; j0 += NW * WD_CH) {
      float4 av[WD_CH][NT], bv[WBF ? 1 : WD_CH][WD_NQ];
      uint2 bh[WD_CH][WD_NQ];
#pragma unroll
      for (int i = 0; i < WD_CH; ++i) {
        const int jb = j0 + NW * i;
        const int k = 16 * jb + 4 * lq;
#pragma unroll
        for (int qi = 0; qi < WD_NQ; ++qi) {
          if (WBF) bh[i][qi] = (jb < nb && q0 + qi < Q) ? wp16[qi][(long)(sg.ks0 + jb) * 64] : make_uint2(0u, 0u);
          else bv[i][qi] = (jb < nb && q0 + qi < Q) ? wp[qi][(long)(sg.ks0 + jb) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int brow = row0 + 16 * t + li;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (jb < nb && sg.x != nullptr && brow < B) {
            const float* xr = sg.x + (long)brow * sg.ld;
            if (sg.vec && k + 3 < sg.K) v = *reinterpret_cast<const float4*>(xr + k);
            else {
              v.x = k < sg.K ? xr[k] : 0.f;
              v.y = k + 1 < sg.K ? xr[k + 1] : 0.f;
              v.z = k + 2 < sg.K ? xr[k + 2] : 0.f;
              v.w = k + 3 < sg.K ? xr[k + 3] : 0.f;
            }
          }
          av[i][t] = v;
        }
      }
      if (sg.drop) {
        const AsrRngKey key = asr_rng_key(seedv, sg.drop_stream);
        const uint32_t thr = asr_drop_threshold(sg.drop_rate);
        const float dscale = 1.f / (1.f - sg.drop_rate);
#pragma unroll
        for (int i = 0; i < WD_CH; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const uint32_t idx = (uint32_t)((long)(row0 + 16 * t + li) * sg.drop_ld + sg.drop_off + 16 * (j0 + NW * i) + 4 * lq);
            av[i][t].x *= asr_drop_mult(key, idx, thr, dscale);
            av[i][t].y *= asr_drop_mult(key, idx + 1, thr, dscale);
            av[i][t].z *= asr_drop_mult(key, idx + 2, thr, dscale);
            av[i][t].w *= asr_drop_mult(key, idx + 3, thr, dscale);
          }
      }
      if (WBF) {
        static_assert(WD_CH == 2, "one bf16 MFMA per pair of blocks");
        bf16x8 a8[NT], b8[WD_NQ];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          a8[t][0] = (__bf16)av[0][t].x; a8[t][1] = (__bf16)av[0][t].y; a8[t][2] = (__bf16)av[0][t].z; a8[t][3] = (__bf16)av[0][t].w;
          a8[t][4] = (__bf16)av[1][t].x; a8[t][5] = (__bf16)av[1][t].y; a8[t][6] = (__bf16)av[1][t].z; a8[t][7] = (__bf16)av[1][t].w;
        }
#pragma unroll
        for (int qi = 0; qi < WD_NQ; ++qi) {
          const uint4 raw = make_uint4(bh[0][qi].x, bh[0][qi].y, bh[1][qi].x, bh[1][qi].y);
          b8[qi] = __builtin_bit_cast(bf16x8, raw);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int qi = 0; qi < WD_NQ; ++qi) acc[t][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[t], b8[qi], acc[t][qi], 0, 0, 0);
      } else {
#pragma unroll
      for (int i = 0; i < WD_CH; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int qi = 0; qi < WD_NQ; ++qi) {
            acc[t][qi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][t].x, bv[i][qi].x, acc[t][qi], 0, 0, 0);
            acc[t][qi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][t].y, bv[i][qi].y, acc[t][qi], 0, 0, 0);
            acc[t][qi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][t].z, bv[i][qi].z, acc[t][qi], 0, 0, 0);
            acc[t][qi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][t].w, bv[i][qi].w, acc[t][qi], 0, 0, 0);
          }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int qi = 0; qi < WD_NQ; ++qi)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave][t][qi][(lq * 4 + r) * 17 + li] = acc[t][qi][r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NP; ++r) {
    if (!live[r]) continue;
    const int p = tid + NTHR * r;
    const int bl = p / (4 * WD_NQ), b = row0 + bl, uu = p % (4 * WD_NQ), qi = uu >> 2, u = uu & 3, j = 4 * (q0 + qi) + u;
    const int t = bl >> 4, bi = bl & 15;
    float s4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += part[w][t][qi][bi * 17 + g * 4 + u];
      s4[g] = v;
    }
    cell_finish<CELL>(d, H, b, j, m[r], hp[r], hp[r], yp[r], cp[r], pre[r], br[r], s4);
  }
}

// ------------------------------------------------------------------------------------------ host side
static int cell_from_name(int rnn_type) { return rnn_type; }

extern "C" int asr_rnn_geometry(int rnn_type, int H, int nseg, const int* K, asr_rnn_geom* g) {
  ASR_CHECK(g && K, ASR_ERR_ARG, "asr_rnn_geometry: null argument");
  ASR_CHECK(rnn_type >= 0 && rnn_type <= 2, ASR_ERR_UNSUPPORTED, "rnn_type: %d is invalid!", rnn_type);
  ASR_CHECK(nseg >= 1 && nseg <= ASR_RNN_MAXSEG && H > 0, ASR_ERR_SHAPE, "asr_rnn_geometry: nseg %d / H %d", nseg, H);
  g->Q = asr_cdiv(H, 4);
  int ks = 0;
  for (int i = 0; i < ASR_RNN_MAXSEG; ++i) g->ks0[i] = 0;
  for (int i = 0; i < nseg; ++i) { g->ks0[i] = ks; ks += asr_cdiv(K[i], 16); }
  g->KSt = ks;
  g->wp_floats = (long)g->Q * g->KSt * 256;
  return ASR_OK;
}

extern "C" int asr_rnn_pack(int rnn_type, int H, int nseg, const float* const* W, const long* ldw, const int* K,
                            const int* is_rec, float* Wp, void* stream) {
  ASR_CHECK(W && ldw && K && is_rec && Wp, ASR_ERR_ARG, "asr_rnn_pack: null argument");
  asr_rnn_geom g;
  int rc = asr_rnn_geometry(rnn_type, H, nseg, K, &g);
  if (rc) return rc;
  PackArgs a{};
  for (int i = 0; i < nseg; ++i) { a.W[i] = W[i]; a.ldw[i] = ldw[i]; a.K[i] = K[i]; a.ks0[i] = g.ks0[i]; a.is_rec[i] = is_rec[i]; }
  a.nseg = nseg; a.KSt = g.KSt; a.NT = g.KSt; a.H = H; a.Q = g.Q; a.cell = cell_from_name(rnn_type); a.Wp = Wp;
  const long n = g.wp_floats;
  hipLaunchKernelGGL(rnn_pack_kernel, dim3((unsigned)min((long)2048, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_rnn_pack_many(int ncells, const asr_rnn_pack_desc* cells, void* stream) {
  ASR_CHECK(cells && ncells > 0, ASR_ERR_ARG, "asr_rnn_pack_many: null argument");
  for (int c0 = 0; c0 < ncells; c0 += ASR_PACK_MANY) {
    PackManyArgs m{};
    const int n = ncells - c0 < ASR_PACK_MANY ? ncells - c0 : ASR_PACK_MANY;
    long nmax = 0;
    for (int c = 0; c < n; ++c) {
      const asr_rnn_pack_desc& d = cells[c0 + c];
      ASR_CHECK(d.Wp && d.nseg >= 1 && d.nseg <= ASR_RNN_MAXSEG, ASR_ERR_ARG, "asr_rnn_pack_many: bad cell %d", c0 + c);
      asr_rnn_geom g;
      int rc = asr_rnn_geometry(d.rnn_type, d.H, d.nseg, d.K, &g);
      if (rc) return rc;
      PackArgs& a = m.c[c];
      for (int i = 0; i < d.nseg; ++i) {
        ASR_CHECK(d.W[i], ASR_ERR_ARG, "asr_rnn_pack_many: null weight (cell %d, segment %d)", c0 + c, i);
        a.W[i] = d.W[i]; a.ldw[i] = d.ldw[i]; a.K[i] = d.K[i]; a.ks0[i] = g.ks0[i]; a.is_rec[i] = d.is_rec[i];
      }
      a.nseg = d.nseg; a.KSt = g.KSt; a.NT = g.KSt; a.H = d.H; a.Q = g.Q; a.cell = cell_from_name(d.rnn_type); a.Wp = d.Wp;
      nmax = g.wp_floats > nmax ? g.wp_floats : nmax;
    }
    hipLaunchKernelGGL(rnn_pack_many_kernel, dim3((unsigned)min((long)256, (nmax + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, m);
    ASR_LAUNCH_CHECK();
  }
  return ASR_OK;
}

static void fill_fwd_dir(FwdDir* d, const asr_rnn_step_fwd* s) {
  *d = FwdDir{};
  d->nseg = s->nseg; d->KSt = s->KSt; d->Wp = s->Wp; d->Wp16 = static_cast<const unsigned short*>(s->Wp16);
  for (int i = 0; i < s->nseg; ++i) {
    d->seg[i].x = s->seg_x[i]; d->seg[i].ld = s->seg_ld[i]; d->seg[i].K = s->seg_K[i]; d->seg[i].ks0 = s->seg_ks0[i];
    d->seg[i].vec = (((uintptr_t)s->seg_x[i] & 15) == 0) && (s->seg_ld[i] % 4 == 0);
    d->seg[i].drop = s->seg_drop_rate[i] > 0.f; d->seg[i].drop_stream = s->seg_drop_stream[i];
    d->seg[i].drop_rate = s->seg_drop_rate[i]; d->seg[i].drop_ld = s->seg_drop_ld[i]; d->seg[i].drop_off = s->seg_drop_off[i];
  }
  d->pre = s->pre; d->pre_ld = s->pre_ld; d->bias = s->bias; d->bias_rec = s->bias_rec;
  d->h_prev = s->h_prev; d->h_prev_ld = s->h_prev_ld; d->c_prev = s->c_prev; d->c_prev_ld = s->c_prev_ld;
  d->y_prev = s->y_prev; d->y_prev_ld = s->y_prev_ld; d->mask = s->mask; d->mask_ld = s->mask_ld;
  d->h_out = s->h_out; d->h_out_ld = s->h_out_ld; d->c_out = s->c_out; d->c_out_ld = s->c_out_ld;
  d->y_out = s->y_out; d->y_out_ld = s->y_out_ld; d->saved = s->saved; d->saved_ld = s->saved_ld;
}

template <int NT, int NQ, int NW>
static void launch_fwd_wide(int rnn_type, const FwdArgs& a, int ndir, hipStream_t st) {
  dim3 grid((unsigned)asr_cdiv(asr_cdiv(a.H, 4), NQ), (unsigned)asr_cdiv(a.B, 16 * NT), (unsigned)ndir);
  const size_t smem = sizeof(float) * NW * NT * NQ * 16 * 17;
  auto go = [&](auto kern) {
    static unsigned long long attr = 0;                                  // one flag per kernel instantiation (generic lambda)
    if (asr_first_use_on_device(attr)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), smem, st, a);
  };
  bool bf = true;
  for (int i = 0; i < ndir; ++i) bf = bf && a.d[i].Wp16 != nullptr;
  if (bf) {
    if (rnn_type == CELL_LSTM) go(rnn_step_fwd_wide_kernel<CELL_LSTM, NT, NQ, NW, 1>);
    else if (rnn_type == CELL_GRU) go(rnn_step_fwd_wide_kernel<CELL_GRU, NT, NQ, NW, 1>);
    else go(rnn_step_fwd_wide_kernel<CELL_RNN, NT, NQ, NW, 1>);
    return;
  }
  if (rnn_type == CELL_LSTM) go(rnn_step_fwd_wide_kernel<CELL_LSTM, NT, NQ, NW, 0>);
  else if (rnn_type == CELL_GRU) go(rnn_step_fwd_wide_kernel<CELL_GRU, NT, NQ, NW, 0>);
  else go(rnn_step_fwd_wide_kernel<CELL_RNN, NT, NQ, NW, 0>);
}

static int launch_fwd(int rnn_type, const FwdArgs& a, int ndir, hipStream_t st) {
  // wide cells with several batch tiles: 16 units x 32 rows per workgroup (see rnn_step_fwd_wide_kernel); ASR_RNN_WIDE=0 turns it off
  static const int wide = getenv("ASR_RNN_WIDE") ? atoi(getenv("ASR_RNN_WIDE")) : 1;
  static const int min_h = getenv("ASR_RNN_WIDE_MIN_H") ? atoi(getenv("ASR_RNN_WIDE_MIN_H")) : 512;   // tests lower it
  bool mult = false;                                             // recurrent dropout: narrow kernel only
  for (int i = 0; i < ndir; ++i)
    for (int sg = 0; sg < a.d[i].nseg; ++sg) mult = mult || a.d[i].seg[sg].mult != nullptr;
  if (wide && !mult && a.H >= min_h && a.B > 16) {
    launch_fwd_wide<2, 4, 16>(rnn_type, a, ndir, st);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
  }
  dim3 grid((unsigned)asr_cdiv(a.H, 4), (unsigned)asr_cdiv(a.B, 16), (unsigned)ndir);
  if (rnn_type == CELL_LSTM) hipLaunchKernelGGL(rnn_step_fwd_kernel<CELL_LSTM>, grid, dim3(256), 0, st, a);
  else if (rnn_type == CELL_GRU) hipLaunchKernelGGL(rnn_step_fwd_kernel<CELL_GRU>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(rnn_step_fwd_kernel<CELL_RNN>, grid, dim3(256), 0, st, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_rnn_cell_fwd(int rnn_type, int B, int H, int ndir, const asr_rnn_step_fwd* steps, const uint32_t* seed,
                                void* stream) {
  ASR_CHECK(steps, ASR_ERR_ARG, "asr_rnn_cell_fwd: null argument");
  ASR_CHECK(rnn_type >= 0 && rnn_type <= 2, ASR_ERR_UNSUPPORTED, "rnn_type: %d is invalid!", rnn_type);
  ASR_CHECK(B > 0 && H > 0 && (ndir == 1 || ndir == 2), ASR_ERR_SHAPE, "asr_rnn_cell_fwd: B %d H %d ndir %d", B, H, ndir);
  FwdArgs a{};
  a.B = B; a.H = H; a.seed = seed;
  for (int i = 0; i < ndir; ++i) {
    ASR_CHECK(steps[i].nseg >= 1 && steps[i].nseg <= ASR_RNN_MAXSEG && steps[i].Wp, ASR_ERR_ARG, "asr_rnn_cell_fwd: bad segments");
    fill_fwd_dir(&a.d[i], &steps[i]);
    for (int s = 0; s < steps[i].nseg; ++s)
      ASR_CHECK(!(a.d[i].seg[s].drop && !seed), ASR_ERR_ARG, "asr_rnn_cell_fwd: dropout needs a device seed");
  }
  return launch_fwd(rnn_type, a, ndir, (hipStream_t)stream);
}

// Whole BiRNN layer forward: T dependent step launches, both directions per launch (las.py:108-126).
extern "C" int asr_rnn_seq_fwd(const asr_rnn_seq* s, void* stream) {
  ASR_CHECK(s, ASR_ERR_ARG, "asr_rnn_seq_fwd: null argument");
  ASR_CHECK(s->rnn_type >= 0 && s->rnn_type <= 2, ASR_ERR_UNSUPPORTED, "rnn_type: %d is invalid!", s->rnn_type);
  ASR_CHECK(s->B > 0 && s->T > 0 && s->H > 0 && (s->ndir == 1 || s->ndir == 2), ASR_ERR_SHAPE, "asr_rnn_seq_fwd: bad B/T/H/ndir");
  const int B = s->B, T = s->T, H = s->H;
  const int NG = cell_ngates(s->rnn_type), NS = cell_nsaved(s->rnn_type);
  const bool lstm = s->rnn_type == CELL_LSTM;
  asr_rnn_geom g;
  int K1[1] = {H};
  asr_rnn_geometry(s->rnn_type, H, 1, K1, &g);
  for (int d = 0; d < s->ndir; ++d) {
    ASR_CHECK(s->pre[d] && s->Wp[d] && s->hseq[d] && s->y && (!lstm || s->cseq[d]), ASR_ERR_ARG, "asr_rnn_seq_fwd: null buffer (dir %d)", d);
  }
  for (int step = 0; step < T; ++step) {
    FwdArgs a{};
    a.B = B; a.H = H; a.seed = nullptr;
    for (int d = 0; d < s->ndir; ++d) {
      const bool rev = s->reverse[d] != 0;
      const int t = rev ? T - 1 - step : step;
      const int tp = rev ? t + 1 : t - 1;  // time index of the previous processing step
      FwdDir& fd = a.d[d];
      fd = FwdDir{};
      fd.nseg = 1; fd.KSt = g.KSt; fd.Wp = s->Wp[d]; fd.Wp16 = static_cast<const unsigned short*>(s->Wp16[d]);
      fd.pre = s->pre[d] + (long)t * NG * H; fd.pre_ld = (long)T * NG * H;
      fd.bias = nullptr; fd.bias_rec = s->bias_rec[d];
      float* hseq = s->hseq[d];
      float* cseq = s->cseq[d];
      if (step == 0) {
        fd.h_prev = s->h0[d]; fd.h_prev_ld = s->h0_ld[d];
        fd.c_prev = lstm ? s->c0[d] : nullptr; fd.c_prev_ld = s->c0_ld[d];
        fd.y_prev = nullptr;
      } else {
        fd.h_prev = hseq + (long)tp * H; fd.h_prev_ld = (long)T * H;
        fd.c_prev = lstm ? cseq + (long)tp * H : nullptr; fd.c_prev_ld = (long)T * H;
        fd.y_prev = s->y + (long)tp * s->y_ld + s->y_col[d]; fd.y_prev_ld = (long)T * s->y_ld;
      }
      fd.seg[0].x = fd.h_prev; fd.seg[0].ld = fd.h_prev_ld; fd.seg[0].K = H; fd.seg[0].ks0 = 0;
      fd.seg[0].vec = (((uintptr_t)fd.h_prev & 15) == 0) && (fd.h_prev_ld % 4 == 0);
      fd.seg[0].mult = s->rec_mult[d]; fd.seg[0].mult_ld = H;       // deepspeech2.py:95-107 recurrent_dropout: h_tm1 * mask_g[B,H], one per gate
      fd.seg[0].mult_gs = (long)B * H;
      fd.mask = s->mask ? s->mask + t : nullptr; fd.mask_ld = T;
      fd.h_out = hseq + (long)t * H; fd.h_out_ld = (long)T * H;
      fd.c_out = lstm ? cseq + (long)t * H : nullptr; fd.c_out_ld = (long)T * H;
      fd.y_out = s->y + (long)t * s->y_ld + s->y_col[d]; fd.y_out_ld = (long)T * s->y_ld;
      fd.saved = s->saved[d] ? s->saved[d] + (long)t * NS * H : nullptr; fd.saved_ld = (long)T * NS * H;
    }
    int rc = launch_fwd(s->rnn_type, a, s->ndir, (hipStream_t)stream);
    if (rc) return rc;
  }
  return ASR_OK;
}

