// Persistent forward kernel for a whole (Bi)RNN layer on gfx950: ONE launch runs all T time steps.
//
// Why: with one launch per step (rnn.hip) a step costs ~6 us, most of it the launch boundary (wave
// launch, kernel-argument fetch, end-of-kernel write-back), not the 0.3 us of MFMA work.  Here the
// workgroups of a (direction, 16-row batch tile) group stay resident, keep their slice of the
// recurrent kernel in registers for the whole sequence, and hand h_t to each other through global
// memory with the placement-independent protocol of cdna_hip_programming.md Guideline 16 (table row
// "agent-scope atomic adds + sc1 poll + workgroup barrier + sc1 stores / sc1 dwordx4 loads"):
//   producer: every workgroup writes its 16 x 4 slice of h_t as ONE 256-byte wave store (two whole
//             128-B lines, write-through `sc1`) into an exchange buffer, waits `vmcnt(0)`, then
//             lane 0 does a relaxed agent-scope atomic add on the group's counter;
//   consumer: one lane polls the counter (relaxed agent-scope = `sc1` load) until all Q workgroups of
//             the group have published step s-1, workgroup barrier, then every wave reads the 16 x H
//             slab of h_{t-1} with `global_load_dwordx4 ... sc1` (bypasses this CU's L1).
// The exchange buffer is double-buffered by step parity: a workgroup can only be one step ahead of
// the slowest member of its group.  Every spin is bounded: on time-out the workgroup raises the
// error word and leaves, the others follow one time-out later; the host falls back to / reports.
// Residency: grid = Q x (B/16) x ndir workgroups of 256 threads, required <= 256 (one per CU).
#include "common.h"

#define CELL_LSTM 0
#define CELL_GRU 1
#define CELL_RNN 2
#define PS_MAXB 4      // K blocks per wave held in registers: H <= 16 * 4 * PS_MAXB = 256

struct PDir {
  const float* pre; const float* Wp; const float* bias_rec;
  const float* h0; long h0_ld; const float* c0; long c0_ld;
  float* hseq; float* cseq; float* saved;
  int reverse, y_col;
};
struct PArgs {
  PDir d[2];
  int B, T, H, KB;
  const uint8_t* mask;
  float* y; long y_ld;
  float* xbuf;          // [groups][2][Q][16][4] exchange buffer
  unsigned* counters;   // [groups] * 32 words apart
  unsigned* err;
  int spin_limit;
};

__device__ __forceinline__ bool ps_wait(unsigned* c, unsigned target, int limit) {
  for (int i = 0; i < limit; ++i) {
    const unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

template <int CELL>
__global__ __launch_bounds__(256) void rnn_seq_fwd_persist_kernel(PArgs a) {
  __shared__ float part[4][16 * 17];
  __shared__ int abort_flag;
  const PDir& d = a.d[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int q = blockIdx.x, b0 = blockIdx.y * 16, Q = gridDim.x;
  const int B = a.B, T = a.T, H = a.H;
  const int group = blockIdx.z * gridDim.y + blockIdx.y;
  unsigned* counter = a.counters + group * 32;
  float* xb = a.xbuf + (long)group * 2 * Q * 64;
  constexpr int NG = CELL == CELL_LSTM ? 4 : (CELL == CELL_GRU ? 3 : 1);
  constexpr int NS = CELL == CELL_RNN ? 1 : 4;

  // this wave's slice of the packed recurrent kernel stays in registers for the whole sequence
  const float4* wp = reinterpret_cast<const float4*>(d.Wp) + (long)q * a.KB * 64 + lane;
  float4 bw[PS_MAXB];
#pragma unroll
  for (int i = 0; i < PS_MAXB; ++i) {
    const int jb = wave + 4 * i;
    bw[i] = jb < a.KB ? wp[(long)jb * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // gate-math ownership (wave 0): lane -> (row bi, unit u); recurrent state lives in registers
  const int bi = lane >> 2, u = lane & 3;
  const int b = b0 + bi, j = 4 * q + u;
  const bool live = b < B && j < H;
  float hp = 0.f, cp = 0.f, yp = 0.f, br[3] = {0.f, 0.f, 0.f};
  if (wave == 0 && live) {
    hp = d.h0 ? d.h0[(long)b * d.h0_ld + j] : 0.f;
    if (CELL == CELL_LSTM) cp = d.c0 ? d.c0[(long)b * d.c0_ld + j] : 0.f;
    if (CELL == CELL_GRU && d.bias_rec) { br[0] = d.bias_rec[j]; br[1] = d.bias_rec[H + j]; br[2] = d.bias_rec[2L * H + j]; }
  }
  if (tid == 0) abort_flag = 0;
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = d.reverse ? T - 1 - s : s;
    // operands of the gate math that do not depend on the exchange
    bool m = true;
    float pre[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) pre[g] = 0.f;
    if (wave == 0 && live) {
      m = a.mask ? a.mask[(long)b * T + t] != 0 : true;
      const float* pr = d.pre + ((long)b * T + t) * NG * H + j;
#pragma unroll
      for (int g = 0; g < NG; ++g) pre[g] = pr[(long)g * H];
    }
    // wait for h_{s-1} of the whole group
    if (s > 0 && wave == 1 && lane == 0) {
      if (!ps_wait(counter, (unsigned)Q * (unsigned)s, a.spin_limit)) abort_flag = 1;
    }
    __syncthreads();
    if (abort_flag) break;

    // A operand: 16 rows x H of h_{s-1}
    f32x4 av[PS_MAXB];
    if (s == 0) {
#pragma unroll
      for (int i = 0; i < PS_MAXB; ++i) {
        const int jb = wave + 4 * i;
        av[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (jb < a.KB && d.h0 != nullptr && b0 + li < B) {
          const float* hr = d.h0 + (long)(b0 + li) * d.h0_ld + 16 * jb + 4 * lq;
          av[i] = (f32x4){hr[0], hr[1], hr[2], hr[3]};
        }
      }
    } else {
      // exchange layout [Q][16 rows][4 units]: the float4 of (row li, units 16jb+4lq..+3) is workgroup 4jb+lq's
      const float* src = xb + (long)((s - 1) & 1) * Q * 64;
      const float* p0 = src + ((long)(4 * (wave + 0) + lq) * 16 + li) * 4;
      const float* p1 = src + ((long)(4 * (wave + 4) + lq) * 16 + li) * 4;
      const float* p2 = src + ((long)(4 * (wave + 8) + lq) * 16 + li) * 4;
      const float* p3 = src + ((long)(4 * (wave + 12) + lq) * 16 + li) * 4;
      // blocks beyond KB alias block 0 of this wave (valid memory); their weights are zero
      if (wave + 4 >= a.KB) p1 = p0;
      if (wave + 8 >= a.KB) p2 = p0;
      if (wave + 12 >= a.KB) p3 = p0;
      if (wave >= a.KB) { p0 = src; p1 = src; p2 = src; p3 = src; }
      asm volatile(
          "global_load_dwordx4 %0, %4, off sc1\n\t"
          "global_load_dwordx4 %1, %5, off sc1\n\t"
          "global_load_dwordx4 %2, %6, off sc1\n\t"
          "global_load_dwordx4 %3, %7, off sc1\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(av[0]), "=&v"(av[1]), "=&v"(av[2]), "=&v"(av[3])
          : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
          : "memory");
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PS_MAXB; ++i) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bw[i].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bw[i].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bw[i].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bw[i].w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][(lq * 4 + r) * 17 + li] = acc[r];
    __syncthreads();

    if (wave == 0) {
      float hnew = hp;
      if (live) {
        float sg[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          sg[g] = part[0][bi * 17 + g * 4 + u] + part[1][bi * 17 + g * 4 + u] + part[2][bi * 17 + g * 4 + u] + part[3][bi * 17 + g * 4 + u];
        float hn, cn = cp;
        float* sv = d.saved ? d.saved + ((long)b * T + t) * NS * H + j : nullptr;
        if (CELL == CELL_LSTM) {
          const float ig = sigmoidf_(pre[0] + sg[0]), fg = sigmoidf_(pre[1] + sg[1]);
          const float gg = tanhf_(pre[2] + sg[2]), og = sigmoidf_(pre[3] + sg[3]);
          const float c2 = fg * cp + ig * gg;
          hn = og * tanhf_(c2);
          cn = m ? c2 : cp;
          if (sv) { sv[0] = ig; sv[H] = fg; sv[2L * H] = gg; sv[3L * H] = og; }
          d.cseq[((long)b * T + t) * H + j] = cn;
        } else if (CELL == CELL_GRU) {
          const float z = sigmoidf_(pre[0] + sg[0] + br[0]);
          const float r = sigmoidf_(pre[1] + sg[1] + br[1]);
          const float arh = sg[3] + br[2];
          const float hh = tanhf_(pre[2] + sg[2] + r * arh);
          hn = z * hp + (1.f - z) * hh;
          if (sv) { sv[0] = z; sv[H] = r; sv[2L * H] = hh; sv[3L * H] = arh; }
        } else {
          hn = tanhf_(pre[0] + sg[0]);
          if (sv) sv[0] = hn;
        }
        hnew = m ? hn : hp;
        yp = m ? hn : yp;
        cp = cn;
        hp = hnew;
        d.hseq[((long)b * T + t) * H + j] = hnew;
        a.y[((long)b * T + t) * a.y_ld + d.y_col + j] = yp;
      }
      // publish this workgroup's 16 x 4 slice: ONE 256-byte wave store (lane = row*4 + unit), then signal
      float* dst = xb + (long)(s & 1) * Q * 64 + (long)q * 64 + lane;
      asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(hnew) : "memory");
      if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (abort_flag && tid == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// scratch the caller provides: exchange buffer + counters + error word (floats)
extern "C" long asr_rnn_persist_ws_floats(int B, int H, int ndir) {
  const long groups = (long)ndir * asr_cdiv(B, 16), Q = asr_cdiv(H, 4);
  return groups * 2 * Q * 64 + groups * 32 + 32;
}

// 1 when the persistent kernel can run this layer (otherwise use asr_rnn_seq_fwd)
extern "C" int asr_rnn_persist_supported(int rnn_type, int B, int T, int H, int ndir) {
  if (rnn_type < 0 || rnn_type > 2 || B <= 0 || T < 2 || H <= 0 || H % 16 != 0 || H > 64 * PS_MAXB) return 0;
  return (long)asr_cdiv(H, 4) * asr_cdiv(B, 16) * ndir <= 256 ? 1 : 0;
}

// Same contract as asr_rnn_seq_fwd (rnn.hip), one launch.  ws: asr_rnn_persist_ws_floats() floats; its
// last 32 words hold the error word (non-zero after the call = a hand-off timed out: results invalid).
extern "C" int asr_rnn_seq_fwd_persist(const asr_rnn_seq* s, float* ws, void* stream) {
  ASR_CHECK(s && ws, ASR_ERR_ARG, "asr_rnn_seq_fwd_persist: null argument");
  ASR_CHECK(asr_rnn_persist_supported(s->rnn_type, s->B, s->T, s->H, s->ndir), ASR_ERR_UNSUPPORTED,
            "asr_rnn_seq_fwd_persist: shape not supported (need H %% 16 == 0, H <= %d, <= 256 workgroups)", 64 * PS_MAXB);
  const int B = s->B, T = s->T, H = s->H;
  const bool lstm = s->rnn_type == CELL_LSTM;
  hipStream_t st = (hipStream_t)stream;
  const long groups = (long)s->ndir * asr_cdiv(B, 16), Q = asr_cdiv(H, 4);
  PArgs a{};
  a.B = B; a.T = T; a.H = H; a.KB = asr_cdiv(H, 16);
  a.mask = s->mask; a.y = s->y; a.y_ld = s->y_ld;
  a.xbuf = ws;
  a.counters = reinterpret_cast<unsigned*>(ws + groups * 2 * Q * 64);
  a.err = a.counters + groups * 32;
  a.spin_limit = 1 << 20;
  for (int d = 0; d < s->ndir; ++d) {
    ASR_CHECK(s->pre[d] && s->Wp[d] && s->hseq[d] && s->y && (!lstm || s->cseq[d]), ASR_ERR_ARG, "asr_rnn_seq_fwd_persist: null buffer (dir %d)", d);
    ASR_CHECK(!s->rec_mult[d], ASR_ERR_UNSUPPORTED, "asr_rnn_seq_fwd_persist: recurrent dropout is not supported");
    PDir& p = a.d[d];
    p.pre = s->pre[d]; p.Wp = s->Wp[d]; p.bias_rec = s->bias_rec[d];
    p.h0 = s->h0[d]; p.h0_ld = s->h0_ld[d]; p.c0 = s->c0[d]; p.c0_ld = s->c0_ld[d];
    p.hseq = s->hseq[d]; p.cseq = s->cseq[d]; p.saved = s->saved[d];
    p.reverse = s->reverse[d]; p.y_col = s->y_col[d];
  }
  // counters + error word are zeroed on every call (a memset node when captured)
  if (hipMemsetAsync(a.counters, 0, sizeof(unsigned) * (groups * 32 + 32), st) != hipSuccess) { asr_set_error("asr_rnn_seq_fwd_persist: memset failed"); return ASR_ERR_HIP; }
  dim3 grid((unsigned)Q, (unsigned)asr_cdiv(B, 16), (unsigned)s->ndir);
  if (s->rnn_type == CELL_LSTM) hipLaunchKernelGGL(rnn_seq_fwd_persist_kernel<CELL_LSTM>, grid, dim3(256), 0, st, a);
  else if (s->rnn_type == CELL_GRU) hipLaunchKernelGGL(rnn_seq_fwd_persist_kernel<CELL_GRU>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(rnn_seq_fwd_persist_kernel<CELL_RNN>, grid, dim3(256), 0, st, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
