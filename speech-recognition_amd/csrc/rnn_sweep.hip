// One-launch forward sweep of a whole (Bi)RNN layer on gfx950 (las.py:62-126 BiRNN, deepspeech2.py:109-119): all T
// time steps of both directions in ONE kernel, the recurrent kernel resident in registers, h_t handed from
// workgroup to workgroup through a small exchange buffer in global memory.
//
// Work split (same as the per-step kernel of rnn.hip, so the arithmetic and its order are identical): the workgroups
// of a (direction, 16-row batch tile) GROUP each own NQ slices of 4 hidden units x all gates; per step a workgroup
// multiplies the group's h_{t-1} [16 x H] with its [H x 16 NQ] slice of U on v_mfma_f32_16x16x4_f32 (K split over the
// 4 waves, reduced through LDS), does the gate math of its 16 x 4 NQ (row, unit) pairs and publishes their h_t.
//
// Hand-off ("the data is the flag", cdna_hip_programming.md Guideline 16 R2, without a tag word): h values are always
// finite, so a slot that holds the SENTINEL (a NaN bit pattern no arithmetic produces) has not been written yet.
//   exchange buffer, per group: [4 slots][Q][16 rows][4 units] f32; step s publishes into slot s % 4;
//   producer: ONE `global_store_dwordx4 ... sc1` per (row, 4-unit slice): a slice's 16 rows are two whole 128-B lines;
//   consumer: every wave re-reads the 16-byte pieces it needs with `global_load_dwordx4 ... sc1` (L1 bypassed) until none
//             of their 4 words is the sentinel - every word validates itself, so no ordering between words is assumed;
//   re-arming: after publishing step s a workgroup overwrites ITS OWN slices of slot (s + 2) % 4 (which held h_{s-2})
//             with the sentinel.  Having gathered all of h_{s-1} it knows that every workgroup has finished step s - 1
//             and therefore consumed h_{s-2}.  The next publish (step s + 1) is issued only after the gather of step
//             s + 1, whose `s_waitcnt vmcnt(0)` also retires that re-arming store; so by the time any consumer can look
//             at slot (s + 2) % 4 for h_{s+2} (it must first have seen this workgroup's h_{s+1}) the sentinel - or the
//             new value - is what memory holds.  Four slots instead of three keep that wait off the critical path.
// Wave roles: gfx950 counts a wave's loads and stores in ONE in-order counter (vmcnt), so a wave that has just published
// cannot consume the loads of its next gather before the write-through acknowledgement of that store has come back
// (measured: +0.34 us per step).  The workgroup therefore has 4 GATHER waves (K split four ways: gather, MFMA, partial sums
// to LDS; they never store to global memory), NQ GATE waves (gate math, publish, re-arm) and ONE WRITER wave that stores the
// layer's own outputs (h, y, c, saved activations / backward coefficients) a step behind, from records the gate waves leave in
// LDS: in a gate wave those stores - to lines nobody has touched yet - would be retired by the `s_waitcnt vmcnt(0)` in front of
// its NEXT publish (round 3: the forward sweep went from 467 to 506 us per las_small layer when the 16-byte coefficient
// stores joined the gate waves' queue).
// Compared with {value, tag} granules this halves the bytes every step moves across the fabric (16 KB instead of 32 KB
// per workgroup at H = 256) and the number of load instructions per gather.
// The buffer is filled with the sentinel before every launch; every spin is bounded: on time-out the workgroup raises
// the error word (and the caller's sticky error flag) and leaves, its peers follow one time-out later.
// Residency: all workgroups of a group must be resident together - 256-thread workgroups with < 20 KB of LDS, several
// fit per CU, so a grid of <= 256 is co-resident even when another stream (RCCL) occupies part of the chip.
#include <stdlib.h>

#include "sweep_common.h"

#define CELL_LSTM 0
#define CELL_GRU 1
#define CELL_RNN 2
#define SW_MAXB 4                  // K blocks per wave held in registers: H <= 16 * 4 * SW_MAXB = 256
#define SW_SLOTS 4
#define SW_SENT 0x7FC0DEADu        // quiet NaN with a payload: never the result of an arithmetic instruction

struct SwDir {
  const float* pre; const float* Wp; const float* bias_rec;
  const float* h0; long h0_ld; const float* c0; long c0_ld;
  float* hseq; float* cseq; float* saved;
  float* coef;          // [B,T,H,CW] backward coefficients for the BPTT sweep (asr_rnn_seq.coef) or NULL
  int reverse, y_col;
};
struct SwArgs {
  SwDir d[2];
  int B, T, H, KB;
  const uint8_t* mask;
  float* y; long y_ld;
  float* xbuf;          // [groups][SW_SLOTS][Q][16 rows][4 units]
  unsigned* err;        // per-launch error word (zeroed by the launch)
  float* err_flag;      // caller's sticky flag (set to 1.0f on time-out, never cleared here) or NULL
  int spin_limit;
  int dbg;              // timing experiments only (ASR_SWEEP_DBG): 2 no wait, 4 no publish
  int delay;            // s_sleep(2) periods before a gather's first poll
  // XCD-local placement (speed only): the grid is 1-D, 8 x nx blocks; block b belongs to XCD class b % 8 = group, member b / 8.
  // Blocks b and b + 8 share an XCD under the round-robin dispatch observed on gfx950 (not promised): the members check it at run
  // time through `ids` and publish with plain stores (the line stays in the XCD's L2: 0.28 us per hand-off against 0.46-0.69
  // with write-through stores, tests/tools/micro/pingpong.hip) only if they all sit on one XCD.
  int xcd, nx, ny, ngroups;
  float* ids;           // [ngroups][nx][4] (inside the sentinel-filled workspace)
  int prio;             // s_setprio level of every wave
};

static int g_spin_limit = 1 << 20;   // ~1.2 s of polling: a deadlock detector, not a latency bound.  A live hand-off takes microseconds (16 000
                                     // individually timed launches: none over 1.2x the median, tests/tools/stall_hunt.py); one unexplained
                                     // time-out at the earlier limit of 0.3 s was seen in ~10^5 launches on a shared host
extern "C" void asr_rnn_sweep_set_spin_limit(int polls) { g_spin_limit = polls; }
int asr_sweep_prio(void) {
  static const int v = getenv("ASR_SWEEP_PRIO") ? atoi(getenv("ASR_SWEEP_PRIO")) : 3;
  return v;
}
extern "C" int asr_rnn_sweep_spin_limit(void) { return g_spin_limit; }

__global__ void sw_fill_kernel(uint32_t* p, size_t n, uint32_t v, uint32_t* zero_words, int nzero, uint32_t expected) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
  if (blockIdx.x == 0 && (int)threadIdx.x < nzero) zero_words[threadIdx.x] = (int)threadIdx.x == SWD_EXPECTED ? expected : 0u;
}

// One wave that waits (bounded by the 100 MHz s_memrealtime clock) until a sweep has counted all of its workgroups in: put in
// front of side-stream work that is meant to run BESIDE that sweep, so that the sweep is resident before the other work can
// take compute units (sweep_common.h).  A pure scheduling hint: when the time is up the stream simply goes on.
__global__ void sweep_gate_kernel(const unsigned* diag, unsigned long long max_ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const unsigned arrived = __hip_atomic_load(diag + SWD_ARRIVED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned expected = __hip_atomic_load(diag + SWD_EXPECTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (expected != 0u && arrived >= expected) break;
    if (__builtin_amdgcn_s_memrealtime() - t0 > max_ticks) break;
    __builtin_amdgcn_s_sleep(32);
  }
}
extern "C" int asr_sweep_gate(const float* diag_words, int max_microseconds, void* stream) {
  ASR_CHECK(diag_words && max_microseconds >= 0, ASR_ERR_ARG, "asr_sweep_gate: bad argument");
  hipLaunchKernelGGL(sweep_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<const unsigned*>(diag_words),
                     (unsigned long long)max_microseconds * 100ull);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

#define SW_REC 16                  // floats of a gate lane's output record: h, y, c, -, saved[4], coefficients[8]
template <int CELL, int NQ>
__global__ __launch_bounds__(64 * (5 + NQ)) void rnn_sweep_fwd_kernel(SwArgs a) {
  __shared__ float part[2][4][NQ][16 * 17];
  __shared__ __attribute__((aligned(16))) float outs[2][NQ][64][SW_REC];   // by step parity: written by the gate waves, stored by the writer a step later
  __shared__ int abort_flag;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool gate_wave = wv < NQ;                       // waves [0, NQ): gate math + publish; [NQ, NQ + 4): gather + MFMA; NQ + 4: writer
  const bool gather_wave = wv >= NQ && wv < NQ + 4, writer_wave = wv == NQ + 4;
  const int wave = gather_wave ? wv - NQ : 0;           // K-split index of a gather wave
  const int li = lane & 15, lq = lane >> 4;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z, gx = gridDim.x, gy = gridDim.y;
  if (a.xcd) {
    const int g = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (g >= a.ngroups || idx >= a.nx) return;           // (before any barrier: the whole workgroup leaves)
    bx = idx; by = g % a.ny; bz = g / a.ny; gx = a.nx; gy = a.ny;
  }
  if (tid == 0) swd_arrive(a.err);                       // start handshake (sweep_common.h)
  swd_setprio(a.prio);
  const int q0 = bx * NQ, b0 = by * 16, Q = gx * NQ;
  const int B = a.B, T = a.T, H = a.H;
  const int group = bz * gy + by;
  const SwDir& d = a.d[bz];
  const long slot_floats = (long)Q * 64;                                  // one slot: [Q][16][4]
  float* xb = a.xbuf + (long)group * SW_SLOTS * slot_floats;
  constexpr int NG = CELL == CELL_LSTM ? 4 : (CELL == CELL_GRU ? 3 : 1);
  constexpr int NS = CELL == CELL_RNN ? 1 : 4;

  // this wave's share of the packed recurrent kernel stays in registers for the whole sequence
  float4 bw[NQ][SW_MAXB];
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    const float4* wp = reinterpret_cast<const float4*>(d.Wp) + (long)(q0 + n) * a.KB * 64 + lane;
#pragma unroll
    for (int i = 0; i < SW_MAXB; ++i) {
      const int jb = wave + 4 * i;
      bw[n][i] = (gather_wave && jb < a.KB) ? wp[(long)jb * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // float offset (inside a slot) of this lane's 16-byte piece of K block jb: slice 4jb+lq, row li
  int goff[SW_MAXB];
#pragma unroll
  for (int i = 0; i < SW_MAXB; ++i) {
    const int jb = wave + 4 * i;
    goff[i] = jb < a.KB ? ((4 * jb + lq) * 16 + li) * 4 : -1;
  }
  // gate-math ownership: wave n < NQ owns slice q0+n; lane -> (row bi, unit u); recurrent state lives in registers
  const int bi = lane >> 2, u = lane & 3;
  const int qn = q0 + (gate_wave ? wv : 0);
  const int b = b0 + bi, j = 4 * qn + u;
  const bool live = gate_wave && b < B && j < H;
  float hp = 0.f, cp = 0.f, yp = 0.f, br[3] = {0.f, 0.f, 0.f};
  if (live) {
    hp = d.h0 ? d.h0[(long)b * d.h0_ld + j] : 0.f;
    if (CELL == CELL_LSTM) cp = d.c0 ? d.c0[(long)b * d.c0_ld + j] : 0.f;
    if (CELL == CELL_GRU && d.bias_rec) { br[0] = d.bias_rec[j]; br[1] = d.bias_rec[H + j]; br[2] = d.bias_rec[2L * H + j]; }
  }
  const long pub_off = ((long)qn * 16 + bi) * 4;                          // this (slice, row)'s 16 bytes inside a slot
  __shared__ int local_mode;
  if (tid == 0) { abort_flag = 0; local_mode = 0; }
  __syncthreads();
  if (a.xcd && wv == 0) {
    // every member publishes the XCD it runs on; all members read all of them (also a start barrier: the group is resident)
    const int my = (int)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 15) + 1;      // HW_REG_XCC_ID, 1-based
    float* idp = a.ids + ((long)group * a.nx) * 4;
    if (lane == 0) {
      const float f = (float)my;
      const f32x4 v = {f, f, f, f};
      float* dst = idp + (long)bx * 4;
      asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
    }
    bool same = true;
    for (int i0 = 0; i0 < a.nx && same; i0 += 64) {
      const int i = i0 + lane;
      const float* src = idp + (long)(i < a.nx ? i : 0) * 4;
      int spins = 0;
      for (;;) {
        f32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src) : "memory");
        const bool fresh = __float_as_uint(v.x) != SW_SENT && __float_as_uint(v.w) != SW_SENT;
        if (__all(fresh)) { same = same && __all(v.x == (float)my); break; }
        if (++spins > a.spin_limit) { abort_flag = 3; same = false; break; }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if (lane == 0) local_mode = same ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (tid == 64 * NQ && !swd_wait_all(a.err, a.spin_limit)) abort_flag = 15;   // (a gather wave's lane: wave 0 may be busy with the id exchange)
  __syncthreads();
  const bool local = local_mode != 0 && !(a.dbg & 32);
  if (a.xcd && tid == 0) {                             // diagnosis: err[2] = workgroups that publish XCD-locally, err[3] = all
    atomicAdd(a.err + 2, local ? 1u : 0u);
    atomicAdd(a.err + 3, 1u);
  }

  // writer wave: lane l stores the records of gate lane l of every gate wave (same (row, unit) mapping)
  auto write_outputs = [&](int sp) {
    const int tp = d.reverse ? T - 1 - sp : sp;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int jn = 4 * (q0 + n) + u;
      if (b < B && jn < H) {
        const float* rec = &outs[sp & 1][n][lane][0];
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rec);
        const long bt = (long)b * T + tp, o = bt * H + jn;
        d.hseq[o] = r0.x;
        a.y[bt * a.y_ld + d.y_col + jn] = r0.y;
        if (CELL == CELL_LSTM) d.cseq[o] = r0.z;
        if (d.saved) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(rec + 4);
          float* sv2 = d.saved + bt * NS * H + jn;
          sv2[0] = sv.x;
          if (NS == 4) { sv2[(long)H] = sv.y; sv2[2L * H] = sv.z; sv2[3L * H] = sv.w; }
        }
        if (d.coef) {
          constexpr int CW = CELL == CELL_RNN ? 4 : 8;
          float* cf = d.coef + o * CW;
          *reinterpret_cast<f32x4*>(cf) = *reinterpret_cast<const f32x4*>(rec + 8);
          if (CW == 8) *reinterpret_cast<f32x4*>(cf + 4) = *reinterpret_cast<const f32x4*>(rec + 12);
        }
      }
    }
  };

  for (int s = 0; s < T; ++s) {
    const int t = d.reverse ? T - 1 - s : s;
    // operands of the gate math that do not depend on the exchange
    bool m = true;
    float pre[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) pre[g] = 0.f;
    if (live) {
      m = a.mask ? a.mask[(long)b * T + t] != 0 : true;
      const float* pr = d.pre + ((long)b * T + t) * NG * H + j;
#pragma unroll
      for (int g = 0; g < NG; ++g) pre[g] = pr[(long)g * H];
    }

    if (gather_wave) {
    // A operand: 16 rows x H of h_{s-1}
    f32x4 av[SW_MAXB];
    if (s == 0) {
#pragma unroll
      for (int i = 0; i < SW_MAXB; ++i) {
        const int jb = wave + 4 * i;
        av[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (jb < a.KB && d.h0 != nullptr && b0 + li < B) {
          const float* hr = d.h0 + (long)(b0 + li) * d.h0_ld + 16 * jb + 4 * lq;
#pragma unroll
          for (int e = 0; e < 4; ++e) av[i][e] = hr[e];
        }
      }
    } else {
      const float* src = xb + (long)((s - 1) & (SW_SLOTS - 1)) * slot_floats;
      const float* p0 = src + (goff[0] >= 0 ? goff[0] : 0);
      const float* p1 = src + (goff[1] >= 0 ? goff[1] : 0);
      const float* p2 = src + (goff[2] >= 0 ? goff[2] : 0);
      const float* p3 = src + (goff[3] >= 0 ? goff[3] : 0);
      int spins = 0;
      for (int w = 0; w < a.delay; ++w) __builtin_amdgcn_s_sleep(2);
      for (;;) {
        asm volatile(
            "global_load_dwordx4 %0, %4, off sc1\n\t"
            "global_load_dwordx4 %1, %5, off sc1\n\t"
            "global_load_dwordx4 %2, %6, off sc1\n\t"
            "global_load_dwordx4 %3, %7, off sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(av[0]), "=&v"(av[1]), "=&v"(av[2]), "=&v"(av[3])
            : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
            : "memory");
        bool ok = true;
#pragma unroll
        for (int i = 0; i < SW_MAXB; ++i)
          if (goff[i] >= 0)
            ok = ok && __float_as_uint(av[i].x) != SW_SENT && __float_as_uint(av[i].y) != SW_SENT && __float_as_uint(av[i].z) != SW_SENT &&
                 __float_as_uint(av[i].w) != SW_SENT;
        if (__all(ok) || (a.dbg & 2)) break;
        if (lds_peek(&abort_flag)) break;
        if (++spins > a.spin_limit) { abort_flag = 1 | (s << 8); break; }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int i = 0; i < SW_MAXB; ++i)
        if (goff[i] < 0) av[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 acc[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < SW_MAXB; ++i) {
#pragma unroll
      for (int n = 0; n < NQ; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bw[n][i].x, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NQ; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bw[n][i].y, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NQ; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bw[n][i].z, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NQ; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bw[n][i].w, acc[n], 0, 0, 0);
    }
    float(*ptw)[NQ][16 * 17] = part[s & 1];             // double-buffered: a gather wave may run one step ahead of the gate waves
#pragma unroll
    for (int n = 0; n < NQ; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) ptw[wave][n][(lq * 4 + r) * 17 + li] = acc[n][r];
    }
    float(*pt)[NQ][16 * 17] = part[s & 1];
    // LDS-only barrier: __syncthreads() would also wait for the gate waves' output stores of the previous step to be acknowledged
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (abort_flag) break;

    if (writer_wave && s > 0) write_outputs(s - 1);       // the records of step s - 1 are complete (the gate waves passed this barrier after writing them)
    if (gate_wave) {
      float hnew = hp;
      float sgv[4] = {0.f, 0.f, 0.f, 0.f};
      float cn = cp;
      if (live) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          sgv[g] = pt[0][wv][bi * 17 + g * 4 + u] + pt[1][wv][bi * 17 + g * 4 + u] + pt[2][wv][bi * 17 + g * 4 + u] +
                   pt[3][wv][bi * 17 + g * 4 + u];
        float hn, c2, prel[4] = {0.f, 0.f, 0.f, 0.f}, sums[4] = {sgv[0], sgv[1], sgv[2], sgv[3]};
#pragma unroll
        for (int g = 0; g < NG; ++g) prel[g] = pre[g];
        asr_cell_forward<CELL>(prel, sums, br, hp, cp, hn, c2, sgv);      // sgv <- the activations saved for backward
        if (CELL == CELL_LSTM) cn = m ? c2 : cp;
        hnew = m ? hn : hp;
        yp = m ? hn : yp;
      }
      // publish first (it is on every other workgroup's critical path): lanes u == 0 gather their row's 4 units and store
      // them as ONE 16-byte write-through store; then re-arm slot (s + 2) % 4 the same way
      f32x4 pub;
      pub.x = hnew;
      pub.y = __shfl_down(hnew, 1, 64);
      pub.z = __shfl_down(hnew, 2, 64);
      pub.w = __shfl_down(hnew, 3, 64);
      if (u == 0 && !(a.dbg & 4)) {
        float* dst = xb + (long)(s & (SW_SLOTS - 1)) * slot_floats + pub_off;
        float* rearm = xb + (long)((s + 2) & (SW_SLOTS - 1)) * slot_floats + pub_off;
        const float sf = __uint_as_float(SW_SENT);
        const f32x4 sent = {sf, sf, sf, sf};
        if (local)                                        // the group sits on one XCD: plain stores keep the line in its L2
          asm volatile(
              "s_waitcnt vmcnt(0)\n\t"
              "global_store_dwordx4 %0, %1, off\n\t"
              "global_store_dwordx4 %2, %3, off" ::"v"(dst), "v"(pub), "v"(rearm), "v"(sent) : "memory");
        else
          asm volatile(
              "s_waitcnt vmcnt(0)\n\t"                    // retires the re-arming store of the previous step (a step old: no stall)
              "global_store_dwordx4 %0, %1, off sc1\n\t"
              "global_store_dwordx4 %2, %3, off sc1" ::"v"(dst), "v"(pub), "v"(rearm), "v"(sent) : "memory");
      }
      if (live) {
        // this step's outputs as a record in LDS; the writer wave stores them after the next barrier
        float* rec = &outs[s & 1][wv][lane][0];
        *reinterpret_cast<f32x4*>(rec) = (f32x4){hnew, yp, cn, 0.f};
        if (d.saved) *reinterpret_cast<f32x4*>(rec + 4) = (f32x4){sgv[0], sgv[1], sgv[2], sgv[3]};
        if (d.coef) {
          // the element-wise backward of this (row, step, unit) as coefficients (asr_rnn_seq.coef): everything the BPTT sweep would
          // otherwise recompute from seven scalar loads per unit and step (four activations, c_t, c_{t-1} / h_{t-1}, the mask) on ITS
          // critical path; here it is a dozen multiplications behind the publish.  A masked step carries the state gradients through
          f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = {0.f, 0.f, 0.f, 0.f};
          if constexpr (CELL == CELL_LSTM) {
            if (m) {
              const float ig = sgv[0], fg = sgv[1], gg = sgv[2], og = sgv[3], tc = tanhf_(cn);
              k0 = (f32x4){og * (1.f - tc * tc), fg, tc * og * (1.f - og), 1.f};
              k1 = (f32x4){gg * ig * (1.f - ig), cp * fg * (1.f - fg), ig * (1.f - gg * gg), 0.f};
            } else {
              k0.y = 1.f;
            }
          } else if constexpr (CELL == CELL_GRU) {
            if (m) {
              const float z = sgv[0], r = sgv[1], hh = sgv[2], arh = sgv[3];
              const float e = (1.f - z) * (1.f - hh * hh);
              k0 = (f32x4){(hp - hh) * z * (1.f - z), e * arh * r * (1.f - r), e, e * r};
              k1 = (f32x4){z, 1.f, 0.f, 0.f};
            }
          } else {
            if (m) k0 = (f32x4){1.f - hnew * hnew, 1.f, 0.f, 0.f};
          }
          *reinterpret_cast<f32x4*>(rec + 8) = k0;
          *reinterpret_cast<f32x4*>(rec + 12) = k1;
        }
        cp = cn;
        hp = hnew;
      }
    }
  }
  // the last step's records (every wave left the loop at the same barrier, by time-out or by count)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (writer_wave && !abort_flag && T > 0) write_outputs(T - 1);
  if (abort_flag && tid == 0) {
    __hip_atomic_store(a.err, (unsigned)abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    swd_record(a.err, (unsigned)abort_flag, local ? 1 : 0);
    if (a.err_flag) __hip_atomic_store(reinterpret_cast<unsigned*>(a.err_flag), 0x3F800000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) swd_depart(a.err);
}

// Workgroups of `kernel` the chip can hold at once (occupancy API x compute units), or -1 when no device is visible (the CPU-only
// build check).  The API is known to answer one block per CU too many near the SGPR limits (MI355X_MICROARCH.md, residency):
// the answer is capped at 4 per CU, and callers keep a quarter of it free for whatever else runs beside the sweep (RCCL).
long asr_sweep_capacity(const void* kernel, int threads) {
  int dev = 0, cus = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); return -1; }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, threads, 0) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return (long)(per < 4 ? per : 4) * cus;
}

template <int NQ>
static long sw_fwd_capacity(int rnn_type) {
  static long cache[3] = {0, 0, 0};                               // 0 = not asked yet
  if (cache[rnn_type] == 0) {
    const void* k = rnn_type == CELL_LSTM ? reinterpret_cast<const void*>(rnn_sweep_fwd_kernel<CELL_LSTM, NQ>)
                  : rnn_type == CELL_GRU ? reinterpret_cast<const void*>(rnn_sweep_fwd_kernel<CELL_GRU, NQ>)
                                         : reinterpret_cast<const void*>(rnn_sweep_fwd_kernel<CELL_RNN, NQ>);
    cache[rnn_type] = asr_sweep_capacity(k, 64 * (5 + NQ));
  }
  return cache[rnn_type];
}

static bool sw_xcd_enabled() {
  static const int v = getenv("ASR_SWEEP_XCD") ? atoi(getenv("ASR_SWEEP_XCD")) : 1;
  return v != 0;
}

static int sw_nq(int B, int H, int ndir) {
  // slices per workgroup: 1 (one workgroup per 4 units, shortest matrix phase) unless that grid would not fit the chip
  static const int forced = getenv("ASR_SWEEP_NQ") ? atoi(getenv("ASR_SWEEP_NQ")) : 0;
  if ((forced == 1 || forced == 2) && (H / 4) % forced == 0) return forced;
  const long wgs = (long)(H / 4) * asr_cdiv(B, 16) * ndir;
  // XCD-local placement wants a group on the 32 compute units of one XCD, one workgroup each (measured on the las_small layer:
  // 64 single-slice workgroups per XCD 2.05 us per step, 32 two-slice ones 1.83; without the placement 1.99 / 2.08)
  if (sw_xcd_enabled() && (long)ndir * asr_cdiv(B, 16) <= 8 && H / 4 > 32 && H / 8 <= 32 && (H / 4) % 2 == 0) return 2;
  return wgs <= 256 ? 1 : 2;
}

// scratch the caller provides: exchange buffer + 32 words holding the per-launch error word (floats)
extern "C" long asr_rnn_sweep_ws_floats(int B, int H, int ndir) {
  const long groups = (long)ndir * asr_cdiv(B, 16), Q = asr_cdiv(H, 4);
  return groups * SW_SLOTS * Q * 64 + groups * Q * 4 + 32;       // exchange slots, XCD ids (one piece per workgroup), error words
}

// 1 when the one-launch sweep can run this layer (otherwise use asr_rnn_seq_fwd)
extern "C" int asr_rnn_sweep_supported(int rnn_type, int B, int T, int H, int ndir) {
  if (rnn_type < 0 || rnn_type > 2 || B <= 0 || T < 2 || H <= 0 || H % 16 != 0 || H > 64 * SW_MAXB) return 0;
  if (ndir != 1 && ndir != 2) return 0;
  const long wgs = (long)(H / 4) * asr_cdiv(B, 16) * ndir;
  if (wgs > 512) return 0;                                               // two slices per workgroup above 256
  // every workgroup of the launch has to be resident at once (they wait for each other): ask the device, keep a quarter free
  const int nq = sw_nq(B, H, ndir);
  const long cap = nq == 1 ? sw_fwd_capacity<1>(rnn_type) : sw_fwd_capacity<2>(rnn_type);
  if (cap >= 0 && (wgs / nq) * 4 > cap * 3) return 0;
  return 1;
}

template <int NQ>
static void sw_launch(int rnn_type, dim3 grid, hipStream_t st, const SwArgs& a) {
  if (rnn_type == CELL_LSTM) hipLaunchKernelGGL((rnn_sweep_fwd_kernel<CELL_LSTM, NQ>), grid, dim3(64 * (5 + NQ)), 0, st, a);
  else if (rnn_type == CELL_GRU) hipLaunchKernelGGL((rnn_sweep_fwd_kernel<CELL_GRU, NQ>), grid, dim3(64 * (5 + NQ)), 0, st, a);
  else hipLaunchKernelGGL((rnn_sweep_fwd_kernel<CELL_RNN, NQ>), grid, dim3(64 * (5 + NQ)), 0, st, a);
}

// Same contract as asr_rnn_seq_fwd (rnn.hip), one launch.  ws: asr_rnn_sweep_ws_floats() floats; the uint32 at
// ws[ws_floats - 32] is non-zero after the call if a hand-off timed out (results invalid); err_flag (optional, device):
// set to 1.0f in that case and never cleared by the library.
extern "C" int asr_rnn_sweep_fwd(const asr_rnn_seq* s, float* ws, float* err_flag, void* stream) {
  ASR_CHECK(s && ws, ASR_ERR_ARG, "asr_rnn_sweep_fwd: null argument");
  ASR_CHECK(asr_rnn_sweep_supported(s->rnn_type, s->B, s->T, s->H, s->ndir), ASR_ERR_UNSUPPORTED,
            "asr_rnn_sweep_fwd: shape not supported (need H %% 16 == 0, H <= %d, <= 512 unit slices x batch tiles)", 64 * SW_MAXB);
  const int B = s->B, T = s->T, H = s->H;
  const bool lstm = s->rnn_type == CELL_LSTM;
  hipStream_t st = (hipStream_t)stream;
  const long groups = (long)s->ndir * asr_cdiv(B, 16), Q = H / 4;
  const long xslots = groups * SW_SLOTS * Q * 64;
  const long xfloats = xslots + groups * Q * 4;           // + one id piece per (possible) workgroup
  SwArgs a{};
  a.B = B; a.T = T; a.H = H; a.KB = H / 16;
  a.mask = s->mask; a.y = s->y; a.y_ld = s->y_ld;
  a.xbuf = ws;
  a.err = reinterpret_cast<unsigned*>(ws + xfloats);
  a.err_flag = err_flag;
  a.spin_limit = g_spin_limit;
  a.dbg = getenv("ASR_SWEEP_DBG") ? atoi(getenv("ASR_SWEEP_DBG")) : 0;
  a.prio = asr_sweep_prio();
  // the first poll of a gather cannot succeed before the publish of the step has crossed the fabric (~1 us): polling earlier only
  // adds traffic in front of it (measured on las_small: 2.38 us per step with no delay, 1.98 with 12 x 128 cycles, 2.15 with 16)
  a.delay = getenv("ASR_SWEEP_DELAY") ? atoi(getenv("ASR_SWEEP_DELAY")) : 12;
  for (int d = 0; d < s->ndir; ++d) {
    ASR_CHECK(s->pre[d] && s->Wp[d] && s->hseq[d] && s->y && (!lstm || s->cseq[d]), ASR_ERR_ARG, "asr_rnn_sweep_fwd: null buffer (dir %d)", d);
    ASR_CHECK(!s->rec_mult[d], ASR_ERR_UNSUPPORTED, "asr_rnn_sweep_fwd: recurrent dropout is not supported (use asr_rnn_seq_fwd)");
    SwDir& p = a.d[d];
    p.pre = s->pre[d]; p.Wp = s->Wp[d]; p.bias_rec = s->bias_rec[d];
    p.h0 = s->h0[d]; p.h0_ld = s->h0_ld[d]; p.c0 = s->c0[d]; p.c0_ld = s->c0_ld[d];
    p.hseq = s->hseq[d]; p.cseq = s->cseq[d]; p.saved = s->saved[d]; p.coef = s->coef[d];
    p.reverse = s->reverse[d]; p.y_col = s->y_col[d];
  }
  const int nq = sw_nq(B, H, s->ndir);
  dim3 grid((unsigned)(Q / nq), (unsigned)asr_cdiv(B, 16), (unsigned)s->ndir);
  const unsigned expected = (unsigned)((Q / nq) * groups);         // workgroups that take part (XCD mode launches spare ones that leave at once)
  const bool xcd_env = sw_xcd_enabled();
  // one group per XCD: at most 8 groups, and a group's workgroups must fit the 32 compute units of one XCD two at a time
  const long cap_all = nq == 1 ? sw_fwd_capacity<1>(s->rnn_type) : sw_fwd_capacity<2>(s->rnn_type);
  if (xcd_env && groups <= 8 && cap_all > 0 && Q / nq <= 32 && (Q / nq) * 4 <= (cap_all / 8) * 3) {
    a.xcd = 1; a.nx = (int)(Q / nq); a.ny = asr_cdiv(B, 16); a.ngroups = (int)groups;
    a.ids = ws + xslots;
    grid = dim3((unsigned)(8 * a.nx), 1, 1);
    if (!getenv("ASR_SWEEP_DELAY")) a.delay = 9;           // hand-offs inside an XCD are shorter (measured optimum 9-10 x 128 cycles)
  }
  // every exchange word is re-armed with the sentinel and the diagnosis words cleared on every call, by an ordinary kernel
  // (graph-capturable; see asr_zero_async for why not a memset node)
  {
    const size_t n = (size_t)xfloats;
    const unsigned fgrid = (unsigned)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
    hipLaunchKernelGGL(sw_fill_kernel, dim3(fgrid), dim3(256), 0, st, reinterpret_cast<uint32_t*>(ws), n, SW_SENT, a.err, 16, expected);
    ASR_LAUNCH_CHECK();
  }
  if (nq == 1) sw_launch<1>(s->rnn_type, grid, st, a);
  else sw_launch<2>(s->rnn_type, grid, st, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
