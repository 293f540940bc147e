// Persistent forward kernel for a whole (Bi)RNN layer on gfx950: ONE launch runs all T time steps.
//
// Why: with one launch per step (rnn.hip) a step costs ~6 us, most of it the launch boundary (wave
// launch, kernel-argument fetch, end-of-kernel write-back), not the 0.3 us of MFMA work.  Here the
// workgroups of a (direction, 16-row batch tile) group stay resident, keep their slice of the
// recurrent kernel in registers for the whole sequence, and hand h_t to each other through global
// memory with data-tagged granules (cdna_hip_programming.md Guideline 16, recipe R2: "the data IS the
// flag"): every value travels as one naturally aligned 8-byte {value, tag = step + 1} granule written
// by ONE write-through (`sc1`) store - no separate flag, no fence, no drain.
//   producer: wave 0 of every workgroup stores its 16 x 4 slice of h_t as one 512-byte wave store
//             (`global_store_dwordx2 ... sc1`, four whole 128-B lines);
//   consumer: every wave re-reads the granules it needs (`global_load_dwordx4 ... sc1`, two granules
//             per load, bypassing this CU's L1) until all their tags equal the step it waits for.
// The exchange buffer is double-buffered by step parity (a workgroup can be at most one step ahead of
// the slowest member of its group) and zeroed before every launch, so a stale or never-written
// granule can never carry the awaited tag.  Every spin is bounded: on time-out the workgroup raises
// the error word and leaves, the others follow one time-out later.
// Residency: grid = Q x (B/16) x ndir workgroups of 256 threads, required <= 256 (one per CU).
#include "common.h"

#define CELL_LSTM 0
#define CELL_GRU 1
#define CELL_RNN 2
#define PS_MAXB 4      // K blocks per wave held in registers: H <= 16 * 4 * PS_MAXB = 256
typedef float f32x2 __attribute__((ext_vector_type(2)));

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
  float* xbuf;          // [groups][2][Q][16 rows][4 units] granules of 2 floats {value, tag}
  unsigned* err;
  int spin_limit;
};

template <int CELL>
__global__ __launch_bounds__(256) void rnn_seq_fwd_persist_kernel(PArgs a) {
  __shared__ float part[2][4][16 * 17];
  __shared__ int abort_flag;
  const PDir& d = a.d[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int q = blockIdx.x, b0 = blockIdx.y * 16, Q = gridDim.x;
  const int B = a.B, T = a.T, H = a.H;
  const int group = blockIdx.z * gridDim.y + blockIdx.y;
  float* xb = a.xbuf + (long)group * 2 * Q * 128;        // 128 floats = 64 granules per workgroup and parity
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
  // granule addresses of this lane's operands: block jb -> workgroup 4jb+lq, row li, units 0..3 = 8 floats
  int goff[PS_MAXB];
#pragma unroll
  for (int i = 0; i < PS_MAXB; ++i) {
    const int jb = wave + 4 * i;
    goff[i] = jb < a.KB ? ((4 * jb + lq) * 16 + li) * 8 : -1;
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

    // A operand: 16 rows x H of h_{s-1}
    float av[PS_MAXB][4];
    if (s == 0) {
#pragma unroll
      for (int i = 0; i < PS_MAXB; ++i) {
        const int jb = wave + 4 * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) av[i][e] = 0.f;
        if (jb < a.KB && d.h0 != nullptr && b0 + li < B) {
          const float* hr = d.h0 + (long)(b0 + li) * d.h0_ld + 16 * jb + 4 * lq;
#pragma unroll
          for (int e = 0; e < 4; ++e) av[i][e] = hr[e];
        }
      }
    } else {
      const float* src = xb + (long)((s - 1) & 1) * Q * 128;
      const float* p[PS_MAXB];
#pragma unroll
      for (int i = 0; i < PS_MAXB; ++i) p[i] = src + (goff[i] >= 0 ? goff[i] : 0);
      const unsigned want = (unsigned)s;               // tag of the data published at step s-1
      f32x4 g0, g1, g2, g3, g4, g5, g6, g7;
      int spins = 0;
      for (;;) {
        asm volatile(
            "global_load_dwordx4 %0, %8, off sc1\n\t"
            "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
            "global_load_dwordx4 %2, %9, off sc1\n\t"
            "global_load_dwordx4 %3, %9, off offset:16 sc1\n\t"
            "global_load_dwordx4 %4, %10, off sc1\n\t"
            "global_load_dwordx4 %5, %10, off offset:16 sc1\n\t"
            "global_load_dwordx4 %6, %11, off sc1\n\t"
            "global_load_dwordx4 %7, %11, off offset:16 sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3), "=&v"(g4), "=&v"(g5), "=&v"(g6), "=&v"(g7)
            : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
            : "memory");
        bool ok = true;
        if (goff[0] >= 0) ok = ok && __float_as_uint(g0.y) == want && __float_as_uint(g0.w) == want && __float_as_uint(g1.y) == want && __float_as_uint(g1.w) == want;
        if (goff[1] >= 0) ok = ok && __float_as_uint(g2.y) == want && __float_as_uint(g2.w) == want && __float_as_uint(g3.y) == want && __float_as_uint(g3.w) == want;
        if (goff[2] >= 0) ok = ok && __float_as_uint(g4.y) == want && __float_as_uint(g4.w) == want && __float_as_uint(g5.y) == want && __float_as_uint(g5.w) == want;
        if (goff[3] >= 0) ok = ok && __float_as_uint(g6.y) == want && __float_as_uint(g6.w) == want && __float_as_uint(g7.y) == want && __float_as_uint(g7.w) == want;
        if (__all(ok)) break;
        if (++spins > a.spin_limit || *(volatile int*)&abort_flag) { abort_flag = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      av[0][0] = g0.x; av[0][1] = g0.z; av[0][2] = g1.x; av[0][3] = g1.z;
      av[1][0] = g2.x; av[1][1] = g2.z; av[1][2] = g3.x; av[1][3] = g3.z;
      av[2][0] = g4.x; av[2][1] = g4.z; av[2][2] = g5.x; av[2][3] = g5.z;
      av[3][0] = g6.x; av[3][1] = g6.z; av[3][2] = g7.x; av[3][3] = g7.z;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PS_MAXB; ++i) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][0], bw[i].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][1], bw[i].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][2], bw[i].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][3], bw[i].w, acc, 0, 0, 0);
    }
    float(*pt)[16 * 17] = part[s & 1];                 // double-buffered: a wave may run one step ahead of wave 0
#pragma unroll
    for (int r = 0; r < 4; ++r) pt[wave][(lq * 4 + r) * 17 + li] = acc[r];
    __syncthreads();
    if (abort_flag) break;

    if (wave == 0) {
      float hnew = hp;
      if (live) {
        float sg[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          sg[g] = pt[0][bi * 17 + g * 4 + u] + pt[1][bi * 17 + g * 4 + u] + pt[2][bi * 17 + g * 4 + u] + pt[3][bi * 17 + g * 4 + u];
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
      }
      // publish the 16 x 4 slice first (it is on every other workgroup's critical path): one 512-byte wave
      // store of {value, tag} granules, lane = row*4 + unit
      float* dst = xb + (long)(s & 1) * Q * 128 + (long)q * 128 + lane * 2;
      const float tagf = __uint_as_float((unsigned)(s + 1));
      f32x2 gran = {hnew, tagf};
      asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(gran) : "memory");
      if (live) {
        d.hseq[((long)b * T + t) * H + j] = hnew;
        a.y[((long)b * T + t) * a.y_ld + d.y_col + j] = yp;
      }
    }
  }
  if (abort_flag && tid == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// scratch the caller provides: granule exchange buffer + error word (floats)
extern "C" long asr_rnn_persist_ws_floats(int B, int H, int ndir) {
  const long groups = (long)ndir * asr_cdiv(B, 16), Q = asr_cdiv(H, 4);
  return groups * 2 * Q * 128 + 32;
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
  const long ws_floats = groups * 2 * Q * 128 + 32;
  PArgs a{};
  a.B = B; a.T = T; a.H = H; a.KB = asr_cdiv(H, 16);
  a.mask = s->mask; a.y = s->y; a.y_ld = s->y_ld;
  a.xbuf = ws;
  a.err = reinterpret_cast<unsigned*>(ws + ws_floats - 32);
  a.spin_limit = 1 << 18;   // ~0.3 s of polling: a live hand-off takes microseconds, start-up skew at most milliseconds
  for (int d = 0; d < s->ndir; ++d) {
    ASR_CHECK(s->pre[d] && s->Wp[d] && s->hseq[d] && s->y && (!lstm || s->cseq[d]), ASR_ERR_ARG, "asr_rnn_seq_fwd_persist: null buffer (dir %d)", d);
    ASR_CHECK(!s->rec_mult[d], ASR_ERR_UNSUPPORTED, "asr_rnn_seq_fwd_persist: recurrent dropout is not supported");
    PDir& p = a.d[d];
    p.pre = s->pre[d]; p.Wp = s->Wp[d]; p.bias_rec = s->bias_rec[d];
    p.h0 = s->h0[d]; p.h0_ld = s->h0_ld[d]; p.c0 = s->c0[d]; p.c0_ld = s->c0_ld[d];
    p.hseq = s->hseq[d]; p.cseq = s->cseq[d]; p.saved = s->saved[d];
    p.reverse = s->reverse[d]; p.y_col = s->y_col[d];
  }
  // every tag and the error word are zeroed on every call (by a kernel: see asr_zero_async)
  if (asr_zero_async(ws, sizeof(float) * ws_floats, st) != hipSuccess) { asr_set_error("asr_rnn_seq_fwd_persist: memset failed"); return ASR_ERR_HIP; }
  dim3 grid((unsigned)Q, (unsigned)asr_cdiv(B, 16), (unsigned)s->ndir);
  if (s->rnn_type == CELL_LSTM) hipLaunchKernelGGL(rnn_seq_fwd_persist_kernel<CELL_LSTM>, grid, dim3(256), 0, st, a);
  else if (s->rnn_type == CELL_GRU) hipLaunchKernelGGL(rnn_seq_fwd_persist_kernel<CELL_GRU>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(rnn_seq_fwd_persist_kernel<CELL_RNN>, grid, dim3(256), 0, st, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
