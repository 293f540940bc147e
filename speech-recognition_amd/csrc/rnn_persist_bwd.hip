// Persistent backward-through-time kernel for a whole (Bi)RNN layer on gfx950: ONE launch runs all T
// steps of rnn_bwd.hip's step kernel (same arithmetic and summation order; equal to fp32 rounding).
//
// Workgroup = 16 hidden units x 16 batch rows, 16 waves; its 16 rows of the recurrent kernel stay in
// registers for the whole sequence, the cell-state gradient, the masked-step carries and the GRU
// z*dh term live in the owner threads' registers.  What moves between workgroups each step is ds (the
// gradient wrt the gate sums, [16 rows x NS*H] per group): every workgroup publishes its [NS][16][16]
// block as whole-line write-through (`sc1`) wave stores into an exchange buffer, drains with
// `vmcnt(0)` and each storing wave adds 1 to the group's counter (relaxed, agent scope); consumers
// poll the counter, pass a workgroup barrier and read the slab with `global_load_dwordx4 ... sc1`
// (cdna_hip_programming.md Guideline 16, third row of the valid-forms table).  The exchange buffer is
// double-buffered by step parity; every spin is bounded and raises the error word on time-out.
#include "common.h"

#define CELL_LSTM 0
#define CELL_GRU 1
#define CELL_RNN 2
#define PB_NW 16
#define PB_CH 4    // K blocks per wave in registers: (recurrent columns) <= 16 * 16 * 4 = 1024

struct PBDir {
  const float* U; long ldu;
  float* saved;                 // [B,T,NS*H]: activations in, ds out
  const float* hseq; const float* cseq;
  const float* h0; long h0_ld; const float* c0; long c0_ld;
  const float* dh_last; long dh_last_ld;
  float* dc;                    // [B,H] in: d/d final c, out: d/d initial c (LSTM)
  float* dh0; long dh0_ld;
  int reverse, y_col;
};
struct PBArgs {
  PBDir d[2];
  int B, T, H;
  const uint8_t* mask;
  const float* dy; long dy_ld;
  float* xbuf; unsigned* counters; unsigned* err;
  int spin_limit;
};

__device__ __forceinline__ bool pb_wait(unsigned* c, unsigned target, int limit) {
  for (int i = 0; i < limit; ++i) {
    const unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

template <int CELL>
__global__ __launch_bounds__(64 * PB_NW) void rnn_seq_bwd_persist_kernel(PBArgs a) {
  __shared__ float part[PB_NW][256];
  __shared__ int abort_flag;
  const PBDir& d = a.d[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int ut = blockIdx.x, unit0 = ut * 16, b0 = blockIdx.y * 16, NU = gridDim.x;
  const int B = a.B, T = a.T, H = a.H;
  constexpr int NS = CELL == CELL_RNN ? 1 : 4;
  const int group = blockIdx.z * gridDim.y + blockIdx.y;
  unsigned* counter = a.counters + group * 32;
  float* xb = a.xbuf + (long)group * 2 * NU * 1024;      // [parity][NU][NS<=4][16 rows][16 units]

  // recurrent-kernel column blocks of this wave: (ds column, kernel column) pairs, 16 wide
  const int nb0 = (CELL == CELL_GRU ? 2 : (CELL == CELL_LSTM ? 4 : 1)) * H / 16;   // first segment
  const int nb = nb0 + (CELL == CELL_GRU ? H / 16 : 0);
  float4 bw[PB_CH];
  int aoff[PB_CH];                                          // float offset of this lane's float4 inside one parity half
#pragma unroll
  for (int i = 0; i < PB_CH; ++i) {
    const int jb = wave + PB_NW * i;
    bw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    aoff[i] = 0;
    if (jb < nb) {
      const int dcol = jb < nb0 ? 16 * jb : 3 * H + 16 * (jb - nb0);
      const int wcol = jb < nb0 ? 16 * jb : 2 * H + 16 * (jb - nb0);
      bw[i] = *reinterpret_cast<const float4*>(d.U + (long)(unit0 + li) * d.ldu + wcol + 4 * lq);
      const int g = dcol / H, uo = dcol - g * H;
      aoff[i] = (((uo >> 4) * 4 + g) * 16 + li) * 16 + 4 * lq;
    }
  }
  // owner threads (first 256): one (row, unit) pair each; per-pair state in registers
  const int row = tid >> 4, un = tid & 15;
  const int b = b0 + row, j = unit0 + un;
  const bool owner = tid < 256;
  const bool live = owner && b < B;
  float dcv = 0.f, carry = 0.f, dirv = 0.f;
  if (live && CELL == CELL_LSTM) dcv = d.dc[(long)b * H + j];
  if (tid == 0) abort_flag = 0;
  __syncthreads();

  for (int step = T - 1; step >= -1; --step) {
    const bool cell = step >= 0;
    const int t = cell ? (d.reverse ? T - 1 - step : step) : 0;
    const int tp = d.reverse ? t + 1 : t - 1;
    // prefetch the element-wise operands of this step
    bool m = true;
    float svv[NS], cpv = 0.f, cov = 0.f, hpv = 0.f, dyv = 0.f, addAv = 0.f;
#pragma unroll
    for (int g = 0; g < NS; ++g) svv[g] = 0.f;
    if (live && cell) {
      m = a.mask ? a.mask[(long)b * T + t] != 0 : true;
      const float* sv = d.saved + ((long)b * T + t) * NS * H + j;
#pragma unroll
      for (int g = 0; g < NS; ++g) svv[g] = sv[(long)g * H];
      dyv = a.dy[((long)b * T + t) * a.dy_ld + d.y_col + j];
      if (CELL == CELL_LSTM) {
        cov = d.cseq[((long)b * T + t) * H + j];
        cpv = step == 0 ? (d.c0 ? d.c0[(long)b * d.c0_ld + j] : 0.f) : d.cseq[((long)b * T + tp) * H + j];
      }
      if (CELL == CELL_GRU) hpv = step == 0 ? (d.h0 ? d.h0[(long)b * d.h0_ld + j] : 0.f) : d.hseq[((long)b * T + tp) * H + j];
    }
    const bool has_src = step < T - 1;                       // ds of the step processed just before
    if (!has_src && live && d.dh_last) addAv = d.dh_last[(long)b * d.dh_last_ld + j];
    float sa = 0.f;
    if (has_src) {
      const int done = T - 1 - step;                         // steps published so far by every workgroup
      if (wave == PB_NW - 1 && lane == 0) {
        if (!pb_wait(counter, (unsigned)(4 * NU) * (unsigned)done, a.spin_limit)) abort_flag = 1;
      }
      __syncthreads();
      if (abort_flag) break;
      const float* src = xb + (long)((step + 1) & 1) * NU * 1024;
      const float* p0 = src + aoff[0];
      const float* p1 = src + aoff[1];
      const float* p2 = src + aoff[2];
      const float* p3 = src + aoff[3];
      f32x4 av[PB_CH];
      asm volatile(
          "global_load_dwordx4 %0, %4, off sc1\n\t"
          "global_load_dwordx4 %1, %5, off sc1\n\t"
          "global_load_dwordx4 %2, %6, off sc1\n\t"
          "global_load_dwordx4 %3, %7, off sc1\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(av[0]), "=&v"(av[1]), "=&v"(av[2]), "=&v"(av[3])
          : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
          : "memory");
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < PB_CH; ++i) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bw[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bw[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bw[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bw[i].w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave][(lq * 4 + r) * 16 + li] = acc[r];
      __syncthreads();
      if (owner) {
#pragma unroll
        for (int w = 0; w < PB_NW; ++w) sa += part[w][tid];
      }
    }
    if (owner) {
      const float dh_state = sa + addAv + dirv;
      if (!cell) {
        if (live && d.dh0) d.dh0[(long)b * d.dh0_ld + j] = dh_state;
      } else {
        float ds[4] = {0.f, 0.f, 0.f, 0.f};
        float dir = 0.f;
        if (live) {
          if (!m) {
            dir = dh_state;
            carry += dyv;
          } else {
            const float dh = dh_state + dyv + carry;
            carry = 0.f;
            if (CELL == CELL_LSTM) {
              const float ig = svv[0], fg = svv[NS > 1 ? 1 : 0], gg = svv[NS > 2 ? 2 : 0], og = svv[NS > 3 ? 3 : 0];
              const float tc = tanhf_(cov);
              const float dct = dcv + dh * og * (1.f - tc * tc);
              ds[0] = dct * gg * ig * (1.f - ig);
              ds[1] = dct * cpv * fg * (1.f - fg);
              ds[2] = dct * ig * (1.f - gg * gg);
              ds[3] = dh * tc * og * (1.f - og);
              dcv = dct * fg;
            } else if (CELL == CELL_GRU) {
              const float z = svv[0], r = svv[NS > 1 ? 1 : 0], hh = svv[NS > 2 ? 2 : 0], arh = svv[NS > 3 ? 3 : 0];
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
          float* o = d.saved + ((long)b * T + t) * NS * H + j;
#pragma unroll
          for (int g = 0; g < NS; ++g) o[(long)g * H] = ds[g];
          dirv = dir;
        }
        // publish ds: layout [NS][16 rows][16 units]; a wave (4 rows x 16 units) writes 256 contiguous bytes per slot
        float* dst = xb + (long)(step & 1) * NU * 1024 + (long)ut * 1024 + row * 16 + un;
        if (NS == 4) {
          asm volatile(
              "global_store_dword %0, %1, off sc1\n\t"
              "global_store_dword %0, %2, off offset:1024 sc1\n\t"
              "global_store_dword %0, %3, off offset:2048 sc1\n\t"
              "global_store_dword %0, %4, off offset:3072 sc1\n\t"
              "s_waitcnt vmcnt(0)" ::"v"(dst), "v"(ds[0]), "v"(ds[1]), "v"(ds[2]), "v"(ds[3]) : "memory");
        } else {
          asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(ds[0]) : "memory");
        }
        if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (live && CELL == CELL_LSTM && !abort_flag) d.dc[(long)b * H + j] = dcv;
  if (abort_flag && tid == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

extern "C" long asr_rnn_persist_bwd_ws_floats(int B, int H, int ndir) {
  const long groups = (long)ndir * asr_cdiv(B, 16), NU = asr_cdiv(H, 16);
  return groups * 2 * NU * 1024 + groups * 32 + 32;
}

extern "C" int asr_rnn_persist_bwd_supported(int rnn_type, int B, int T, int H, int ndir) {
  if (rnn_type < 0 || rnn_type > 2 || B <= 0 || T < 2 || H <= 0 || H % 16 != 0) return 0;
  const int cols = (rnn_type == CELL_LSTM ? 4 : (rnn_type == CELL_GRU ? 3 : 1)) * H;
  if (cols > 16 * PB_NW * PB_CH) return 0;
  return (long)asr_cdiv(H, 16) * asr_cdiv(B, 16) * ndir <= 256 ? 1 : 0;
}

// Same contract as asr_rnn_seq_bwd (rnn_bwd.hip) in one launch.  gs->direct / gs->dy_carry are not used
// (those carries live in registers).  ws: asr_rnn_persist_bwd_ws_floats() floats; the uint32 at
// ws[ws_floats - 32] is non-zero after the call if a hand-off timed out.
extern "C" int asr_rnn_seq_bwd_persist(const asr_rnn_seq* s, const asr_rnn_seq_grad* gs, float* ws, void* stream) {
  ASR_CHECK(s && gs && ws, ASR_ERR_ARG, "asr_rnn_seq_bwd_persist: null argument");
  ASR_CHECK(asr_rnn_persist_bwd_supported(s->rnn_type, s->B, s->T, s->H, s->ndir), ASR_ERR_UNSUPPORTED,
            "asr_rnn_seq_bwd_persist: shape not supported");
  const int B = s->B, T = s->T, H = s->H;
  const bool lstm = s->rnn_type == CELL_LSTM;
  const int NG = lstm ? 4 : (s->rnn_type == CELL_GRU ? 3 : 1);
  hipStream_t st = (hipStream_t)stream;
  const long groups = (long)s->ndir * asr_cdiv(B, 16), NU = asr_cdiv(H, 16);
  PBArgs a{};
  a.B = B; a.T = T; a.H = H; a.mask = s->mask; a.dy = gs->dy; a.dy_ld = gs->dy_ld;
  a.xbuf = ws;
  a.counters = reinterpret_cast<unsigned*>(ws + groups * 2 * NU * 1024);
  a.err = a.counters + groups * 32;
  a.spin_limit = 1 << 18;   // ~0.3 s of polling: a live hand-off takes microseconds, start-up skew at most milliseconds
  ASR_CHECK(gs->dy, ASR_ERR_ARG, "asr_rnn_seq_bwd_persist: dy missing");
  for (int d = 0; d < s->ndir; ++d) {
    ASR_CHECK(s->saved[d] && s->U[d] && s->hseq[d] && (!lstm || (gs->dc[d] && s->cseq[d])), ASR_ERR_ARG, "asr_rnn_seq_bwd_persist: null buffer (dir %d)", d);
    ASR_CHECK(!s->rec_mult[d], ASR_ERR_UNSUPPORTED, "asr_rnn_seq_bwd_persist: recurrent dropout is not supported");
    const long ldu = s->ldu[d] ? s->ldu[d] : (long)NG * H;
    ASR_CHECK((((uintptr_t)s->U[d]) & 15) == 0 && ldu % 4 == 0, ASR_ERR_ARG, "asr_rnn_seq_bwd_persist: recurrent kernel must be 16-byte aligned");
    PBDir& p = a.d[d];
    p.U = s->U[d]; p.ldu = ldu; p.saved = s->saved[d]; p.hseq = s->hseq[d]; p.cseq = s->cseq[d];
    p.h0 = s->h0[d]; p.h0_ld = s->h0_ld[d]; p.c0 = s->c0[d]; p.c0_ld = s->c0_ld[d];
    p.dh_last = gs->dh_last[d]; p.dh_last_ld = gs->dh_last_ld[d];
    p.dc = gs->dc[d]; p.dh0 = gs->dh0[d]; p.dh0_ld = gs->dh0_ld[d];
    p.reverse = s->reverse[d]; p.y_col = s->y_col[d];
  }
  if (asr_zero_async(a.counters, sizeof(unsigned) * (groups * 32 + 32), st) != hipSuccess) { asr_set_error("asr_rnn_seq_bwd_persist: memset failed"); return ASR_ERR_HIP; }
  dim3 grid((unsigned)NU, (unsigned)asr_cdiv(B, 16), (unsigned)s->ndir);
  dim3 block(64 * PB_NW);
  if (s->rnn_type == CELL_LSTM) hipLaunchKernelGGL(rnn_seq_bwd_persist_kernel<CELL_LSTM>, grid, block, 0, st, a);
  else if (s->rnn_type == CELL_GRU) hipLaunchKernelGGL(rnn_seq_bwd_persist_kernel<CELL_GRU>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(rnn_seq_bwd_persist_kernel<CELL_RNN>, grid, block, 0, st, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
