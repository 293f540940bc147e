// One-launch backward-through-time sweep of a whole (Bi)RNN layer on gfx950: the mirror of rnn_sweep.hip.
//
// Per step the recurrence needs  dh_{t-1}[16 x H] = ds_t[16 x G*H] x U^T  for every (direction, 16-row batch tile) GROUP, then
// the element-wise gate gradients ds_{t-1} = f(dh_{t-1}, saved activations).  The product is decomposed in TWO dimensions
// over a GxG square of workgroups per group (SUMMA style).  With KU = H/G hidden units per "unit group":
//     workgroup (i, j):  computes ds_t for the units of group i (16 rows x KU units x gates; the gate math is element-wise, so
//                        the G workgroups of row i repeat it - far cheaper than moving its result),
//                        multiplies it with its resident [KU gates-columns x KU] block of U^T -> the partial dh of the
//                        units of group j that is due to the units of group i: one [16 x KU] block,
//                        PUBLISHES that block once; the G workgroups (j, *) of row j all read it.
//     To start step t-1 a workgroup gathers the G blocks (i, i') that make up dh for ITS unit group: 16 x H floats (16 KB at
//     H = 256) - the same volume the forward sweep gathers - while it publishes only 16 x KU floats (2 KB).
// Splitting the product over output units only (the step kernels, the first persistent kernel) makes every workgroup gather
// all of ds_t (64 KB per workgroup and step); splitting it over the contraction only makes every workgroup SEND 16 KB in
// H/4 separate blocks (measured: 3.9 - 5.0 us per step, bound by the write-through stores).  The square needs neither.
//
// Hand-off = the forward sweep's: blocks are self-validating (a SENTINEL NaN pattern marks "not written yet"), written with
// 16-byte write-through stores, polled with `global_load_dwordx4 ... sc1`; step p publishes into slot (p+1) % 6, which the
// gather of step p+1 reads, and the PUBLISHER re-arms its block THREE steps later.  Why three: in the square a workgroup gathers
// only the G blocks of ITS row - not everybody's, as in the forward sweep's all-gather - so the return of its gather of step p
// proves only that the G senders of that row have finished their gather of step p-1, and, one hop further back, that EVERY
// workgroup has finished its gather of step p-2.  The readers of a publisher's block sit in another row: all that is certain
// about them at publish time p is "gather p-2 done", i.e. the block published at step p-3 has been consumed.  (Rounds 1-2
// re-armed one step earlier, on the all-gather's argument: a reader that was still polling its gather p-1 for some OTHER, late
// sender then saw the sentinel come back over a block it had already seen fresh, and waited for ever - the "lost hand-off"
// time-outs that showed once other kernels shared the memory system with the sweep; sweep_common.h's record caught one: all 256
// workgroups resident, gather stuck at step 70.)  Six slots: the sentinel stored at step p+3 is retired by the publisher's
// `s_waitcnt vmcnt(0)` of step p+4, before its publish of step p+4; a reader polls the slot again for step p+7, after it has
// consumed that publisher's block of step p+5 - so what it finds there is the sentinel or the new block, never the old one.
// All stores to one word come from one wave in program order.  (Do NOT let the reader re-arm: two agents storing to the same
// word race even when one store is issued only after the other was seen retired.)
//
// Wave roles (64 x (4 + NT) threads).  gfx950 counts a wave's loads and stores in ONE in-order counter (vmcnt), so a wave with
// write-through stores in flight cannot consume a later load before their acknowledgements are back:
//   waves 0-3  GATHER + OWNER: sleep until the blocks are due, poll the G blocks, add them up, fetch the NEXT step's coefficient packs
//              (buffer loads, at the start of the local work), gate-gradient math of the 16 x KU (row, unit) pairs, ds to LDS, product;
//   waves 4..  PUBLISH (one per 16-unit output tile): add the gather waves' four partial blocks (the contraction is split over them),
//              publish the [16 x KU] block, re-arm the block of three steps ago, write ds out of place (each column of the square its
//              share of a row's pieces) and sum the bias gradients.  They never load from global memory.
// The roles hand over through LDS counters (a workgroup barrier would make the gather waves wait for store
// acknowledgements); LDS buffers are double-buffered by step parity and protected by causality through the exchange
// (a gather of step p+2 cannot complete before this workgroup's own publish of step p+1).
// Every spin is bounded; on time-out the error word (who gave up | step << 8) and the caller's sticky flag are raised.
#include <stdio.h>
#include <stdlib.h>

#include "sweep_common.h"

#define CELL_LSTM 0
#define CELL_GRU 1
#define CELL_RNN 2
#define SB_SLOTS 6
#define SB_SENT 0x7FC0DEADu

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Stage timeline of ONE workgroup (block 0 of group 0; ASR_SWEEP_DBG bit 128; read back with asr_debug_sweep_trace): per step eight
// s_memrealtime stamps (10 ns ticks) - 0 gather entered, 1 gather complete, 2 gate gradients done, 3 partial block in LDS,
// 4 publish wave saw all four partial blocks, 5 publish + re-arm issued.  Timing aid only.
#define SB_TRACE_STEPS 512
#define SB_TRACE_ALL_STEPS 64                                       // ASR_SWEEP_DBG bit 256: EVERY workgroup of group 0 stamps "gather complete" and
#define SB_TRACE_ALL_FIRST 64                                       // "publish issued" of steps [64, 128): [workgroup <= 64][step][2] behind the stage stamps
__device__ unsigned long long sb_trace[SB_TRACE_STEPS * 8 + 64 * SB_TRACE_ALL_STEPS * 2];

struct SbDir {
  const float* U; long ldu;
  const float* coef;            // [B,T,H,CW]: the element-wise backward as coefficients, written by the forward sweep (asr_rnn_seq.coef)
  float* ds;                    // [B,T,NS*H]: gate-sum gradients out (a buffer nobody in this launch reads)
  float* db; float* db_rec;     // optional bias gradients (+=): db[NG*H] = sum over rows and steps of the input-side slots; GRU also db_rec[3H]
                                // (recurrent bias: slots z, r and the recurrent part of h~).  Saves a pass over ds per direction (asr_colsum)
  const float* dh_last; long dh_last_ld;
  float* dc;                    // [B,H] in: d/d final c, out: d/d initial c (LSTM)
  float* dh0; long dh0_ld;
  int reverse, y_col;
};
struct SbArgs {
  SbDir d[2];
  int B, T, H, G;               // G x G workgroups per group
  const uint8_t* mask;
  const float* dy; long dy_ld;
  float* xbuf;                  // [groups][SB_SLOTS][G rows][G senders][NT][64 lanes][4]
  long xbytes;
  unsigned* err; float* err_flag;
  int spin_limit;
  int dbg;                      // timing experiments only (ASR_SWEEP_DBG): 1 no re-arm, 2 no wait, 4 no publish, 16 no ds stores
  int delay;                    // 10 ns ticks between entering a gather and its first poll
  int xcd, nx, ny, ngroups;     // XCD-local placement (see rnn_sweep.hip): 1-D grid, block b -> group b % 8, member b / 8
  float* ids;                   // [ngroups][nx][4]
  int prio;                     // s_setprio level of every wave
  int rowxcd;                   // 1: block index -> (i = bx % G, j = bx / G)
};

// abort_flag doubles as the diagnosis: 0 = running, else (who gave up first) | (step << 8): 1 gather, 2 owner waiting for the other
// gather waves, 3 publish wave waiting for the owner step, 4 publish wave waiting for its contraction partners
__device__ __forceinline__ bool sb_wait(lds_flag_t c, int target, lds_flag_t abort_flag, int limit, int code) {
  for (int i = 0; *c < target; ++i) {
    if (*abort_flag) return false;
    if (i > limit) { *abort_flag = code; return false; }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}

template <int CELL, int NT>      // NT: 16-unit tiles per unit group (KU = 16 NT)
__global__ __launch_bounds__(64 * (4 + NT)) void rnn_sweep_bwd_kernel(SbArgs a) {
  constexpr int NS = CELL == CELL_RNN ? 1 : 4;                      // saved / ds slots per unit
  constexpr int NGR = CELL == CELL_LSTM ? 4 : (CELL == CELL_GRU ? 3 : 1);   // ds slots that multiply the recurrent kernel
  constexpr int KU = 16 * NT;                                       // units per unit group
  constexpr int UW = 4 * NT;                                        // units per gather wave (= its MFMA k-steps: one unit x 4 gate slots each)
  constexpr int LP = 4 / NT;                                        // lanes that share a position (they split the senders)
  constexpr int TRLD = 4 * UW + 4;                                  // padded row of a wave's ds image
  // ds of a gather wave's units, [row][unit * 4 + gate slot]: written and read back (as MFMA A operands) by the gather wave itself,
  // read once more by the publish waves, which store ds to memory AFTER their publish.  Three copies by step: the gather wave
  // rewrites copy p % 3 at step p + 3, when its gather has returned - which implies this workgroup's publish of step p + 1, and
  // the publish waves stored step p's ds before that
  __shared__ __attribute__((aligned(16))) float tr[3][4][16][TRLD];
  __shared__ __attribute__((aligned(16))) float part[2][4][NT][256];   // partial dh blocks of the gather waves (MFMA C layout), by step parity
  __shared__ int abort_flag;
  __shared__ int g_done[4];                                         // per gather wave: steps whose partial block is in LDS
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z, gy = gridDim.y;
  if (a.xcd) {
    const int g = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (g >= a.ngroups || idx >= a.nx) return;                     // (before any barrier: the whole workgroup leaves)
    bx = idx; by = g % a.ny; bz = g / a.ny; gy = a.ny;
  }
  if (threadIdx.x == 0) swd_arrive(a.err);                          // start handshake (sweep_common.h)
  swd_setprio(a.prio);
  const SbDir& d = a.d[bz];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool gather_wave = wv < 4;
  const int li = lane & 15, lq = lane >> 4;
  // (i, j): ds of unit group i, partial dh of unit group j.  Consecutive blocks go to consecutive XCDs (round-robin dispatch), so with
  // i = bx % G the G workgroups of a row - which all fetch the SAME coefficient packs, a gigabyte per las_small launch when each of them
  // misses in an L2 of its own - sit on one XCD and share its L2 (a.rowxcd; 0 = the placement until round 4, column j on XCD j)
  const int G = a.G, gi_ = a.rowxcd ? bx % G : bx / G, gj_ = a.rowxcd ? bx / G : bx % G;
  const int b0 = by * 16;
  const int B = a.B, T = a.T, H = a.H;
  const int group = bz * gy + by;
  const long blk = (long)NT * 256;                                  // floats per block
  const long slot_floats = (long)G * G * blk;
  float* xb = a.xbuf + (long)group * SB_SLOTS * slot_floats;
  const int lds_limit = a.spin_limit > (1 << 20) ? a.spin_limit : (a.spin_limit << 4);   // LDS polls are ~16x shorter than fabric polls
  __shared__ int local_mode;
  if (tid == 0) { abort_flag = 0; local_mode = 0; }
  if (tid < 4) g_done[tid] = 0;
  __syncthreads();
  if (a.xcd && wv == 4) {
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
        const bool fresh = __float_as_uint(v.x) != SB_SENT && __float_as_uint(v.w) != SB_SENT;
        if (__all(fresh)) { same = same && __all(v.x == (float)my); break; }
        if (++spins > a.spin_limit) { abort_flag = 5; same = false; break; }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if (lane == 0) local_mode = same ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (tid == 0 && !swd_wait_all(a.err, a.spin_limit)) abort_flag = 15;   // the whole grid is resident before the first step
  __syncthreads();
  const bool local = local_mode != 0 && !(a.dbg & 32);
  if (a.xcd && tid == 0) {                             // diagnosis: err[2] = workgroups that publish XCD-locally, err[3] = all
    atomicAdd(a.err + 2, local ? 1u : 0u);
    atomicAdd(a.err + 3, 1u);
  }

  if (gather_wave) {
    // ------------------------------------------------------------------------------------------ GATHER + GATE GRADIENTS + PRODUCT
    // Wave w owns the units UW w .. UW w + UW - 1 of group i for all 16 rows.  Lane = (position pl, part): a position is one
    // 16-byte piece of a block (rows 4 plq .. 4 plq + 3 of one unit); its LP lanes split the G senders and add up with two
    // cross-lane shuffles (no LDS, no hand-over), then each finishes NT of the 4 rows.
    const int pl = lane / LP, pt = lane % LP;
    const int ul = pl >> 2, plq = pl & 3;                           // unit inside the wave, row quad
    const int un = UW * wv + ul;                                    // unit inside the group
    const int j = gi_ * KU + un;                                    // hidden unit
    const bool writer = gj_ == 0;                                   // column 0 of the square writes the layer's outputs
    int brow[NT];
    bool live[NT];
    float dcv[NT], carry[NT], dirv[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) {
      brow[r] = b0 + 4 * plq + pt * NT + r;
      live[r] = brow[r] < B;
      dcv[r] = (live[r] && CELL == CELL_LSTM) ? d.dc[(long)brow[r] * H + j] : 0.f;
      carry[r] = 0.f; dirv[r] = 0.f;
    }
    // resident B operands: k-step ks = unit ks of this wave, k-row lq = gate slot; the lane (li, lq) holds
    // U[unit 16 t + li of group j][gate column lq of unit (UW w + ks) of group i]   (GRU: k-rows z, r, recurrent part of h~; row 3 unused)
    float bwv[NT][UW];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ks = 0; ks < UW; ++ks)
        bwv[t][ks] = lq < NGR ? d.U[(long)(gj_ * KU + 16 * t + li) * d.ldu + (long)lq * H + gi_ * KU + UW * wv + ks] : 0.f;
    const long row_off = (long)gi_ * G * blk;                       // row i inside a slot
    const long pos_off = ((long)(un >> 4) * 64 + plq * 16 + (un & 15)) * 4;   // this position inside a block
    // element-wise operands are fetched one step ahead (a wave's loads retire in order: fetched at the top of their own step they
    // would sit in front of the gather's polls)
    // (two 16-byte coefficient loads + dy per row instead of the seven / eight scalar loads of rounds 1-2 - activations, c_t, c_{t-1},
    // mask: what the operand loads cost in front of the polls was measured at ~0.8 us of the 3.4 us step)
    constexpr int CW = CELL == CELL_RNN ? 4 : 8;
    struct Operands { f32x4 k0, k1; float dyv; };
    // Buffer loads: (byte offset of the lane's (row, unit), fixed for the whole sweep, in a VGPR) + (byte offset of the step, wave-uniform,
    // in an SGPR).  NO vector address arithmetic per step - with flat pointers the compiler computed each row's address in registers it
    // had just named as a load destination and put `s_waitcnt vmcnt(0)` between the loads (round 4, from the ISA: the gather wave sat
    // out the whole memory latency of its coefficient loads, 1.5 us from "partial block in LDS" to "next gather entered").  Rows beyond
    // B point past the buffer: the hardware returns zeros.
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.coef), 0, (int)((long)B * T * H * CW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)((long)B * T * a.dy_ld * 4), 0x00020000);
    unsigned cvo[NT], yvo[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) {
      cvo[r] = live[r] ? (unsigned)((((long)brow[r] * T) * H + j) * CW * 4) : 0x80000000u;
      yvo[r] = live[r] ? (unsigned)((((long)brow[r] * T) * a.dy_ld + d.y_col + j) * 4) : 0x80000000u;
    }
    const unsigned cstep = (unsigned)H * CW * 4, ystep = (unsigned)a.dy_ld * 4;   // bytes per time step
    typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    auto fetch = [&](int p, Operands (&o)[NT]) {
      // straight-line: behind the last step the loads repeat the last step's (valid, unused) - a branch here would end in register
      // copies at its join, and the compiler waits for the loads in front of those.  Only the components the gate gradients read are
      // loaded: a destination register whose value is dead gets reused while the load is in flight, behind another wait
      const int step = p < T ? T - 1 - p : 0;
      const int t = d.reverse ? T - 1 - step : step;
      const unsigned cs = (unsigned)t * cstep, ys = (unsigned)t * ystep;
#pragma unroll
      for (int r = 0; r < NT; ++r) {
        o[r].k1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (CELL == CELL_LSTM) {                          // k0 = {A, f, Co, m}, k1 = {Ci, Cf, Cg, -}
          o[r].k0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(crs, (int)cvo[r], (int)cs, 0));
          const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(crs, (int)(cvo[r] + 16u), (int)cs, 0);
          o[r].k1.x = __uint_as_float(v.x); o[r].k1.y = __uint_as_float(v.y); o[r].k1.z = __uint_as_float(v.z);
        } else if constexpr (CELL == CELL_GRU) {                    // k0 = {Cz, Cr, E, E r}, k1 = {z, m, -, -}
          o[r].k0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(crs, (int)cvo[r], (int)cs, 0));
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(crs, (int)(cvo[r] + 16u), (int)cs, 0);
          o[r].k1.x = __uint_as_float(v.x); o[r].k1.y = __uint_as_float(v.y);
        } else {                                                    // k0 = {1 - h^2, m, -, -}
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(crs, (int)cvo[r], (int)cs, 0);
          o[r].k0 = (f32x4){__uint_as_float(v.x), __uint_as_float(v.y), 0.f, 0.f};
        }
        o[r].dyv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, (int)yvo[r], (int)ys, 0));
      }
    };
    Operands opA[NT], opB[NT];
    fetch(0, opA);
    int nfail = 0;                                                   // gathers of the sweep whose first poll came too early (diagnosis)
    // ds goes OUT OF PLACE.  The G workgroups of a square's row all read the same saved activations, one step ahead, and a row's
    // writer has no proof of where its row mates are: its gather returns blocks of column i, i.e. of the workgroups (i', i) - only
    // one of which is a row mate.  Writing ds over `saved` one step late (the first version of this kernel) therefore raced
    // with a slow row mate's fetch: seen as garbage gradients once other kernels ran beside the sweep (round 3, tests/tools/dbg_overlap.py).
    // The step body exists twice (steps 2q and 2q + 1) with the two operand register sets swapped: one loop body that fetches into
    // the set it has just read ends in loop-carried register copies, and the compiler waits for the loads in front of those.
    auto one_step = [&](const int p, const Operands (&cur)[NT], Operands (&nxt)[NT]) -> bool {   // false: the sweep is over (or aborted)
      const bool cell = p < T;
      const int step = T - 1 - p;
      const int t = cell ? (d.reverse ? T - 1 - step : step) : 0;
      float sa[NT];
#pragma unroll
      for (int r = 0; r < NT; ++r) sa[r] = 0.f;
      const bool tracing = (a.dbg & 128) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wv == 0 && p < SB_TRACE_STEPS;
      if (tracing && lane == 0) sb_trace[p * 8 + 0] = __builtin_amdgcn_s_memrealtime();
      if (p > 0) {
        const unsigned long long t_enter = __builtin_amdgcn_s_memrealtime();
        const float* src = xb + (long)(p % SB_SLOTS) * slot_floats + row_off + pos_off;
        const bool u0 = pt < G, u1 = pt + LP < G, u2 = pt + 2 * LP < G, u3 = pt + 3 * LP < G;
        const float* p0 = src + (u0 ? (long)pt * blk : 0);
        const float* p1 = src + (u1 ? (long)(pt + LP) * blk : 0);
        const float* p2 = src + (u2 ? (long)(pt + 2 * LP) * blk : 0);
        const float* p3 = src + (u3 ? (long)(pt + 3 * LP) * blk : 0);
        f32x4 v0, v1, v2, v3;
        int spins = 0;
        // Poll as late as the data allows: every poll round of the chip is 4 MB of device-scope loads on the fabric the publishes travel
        // on, and the hand-off gets slower with them (round 4, tests/tools/exp/bptt_knobs.py: 3.47 us per step polling from the start
        // of the gather, 2.77 with the first poll timed to arrive just behind the data, 2.98 a quarter of a microsecond either side:
        // a poll that comes too early costs a whole round trip, one that comes late costs its lateness).  The sleep is a FIXED time
        // (a.delay, 10 ns ticks of s_memrealtime - not cycles: the clock moves between and during the launches of a training step).
        // Three self-tuning versions were built and measured and lost to it inside the training step, where it matters: steering by the
        // rate of failed first polls (that rate has a floor - 5 % alone, 14 % in the step - whatever the delay, and every threshold drifts
        // late there: 180 ticks, 10.1 ms per step against 9.58), the same with the learned delay carried from launch to launch, and
        // timing the poll from the arrival the idle publish wave measures with one-piece probes (the workgroups wait for each other:
        // whoever polls late makes the others' data late, and the measured arrival runs away - 5.5 us per step at a lead of 0.2 us).
        if (a.delay > 0) {
          const unsigned long long target = t_enter + (unsigned long long)a.delay;
          while (__builtin_amdgcn_s_memrealtime() < target) __builtin_amdgcn_s_sleep(1);
        }
        for (;;) {
          asm volatile(
              "global_load_dwordx4 %0, %4, off sc1\n\t"
              "global_load_dwordx4 %1, %5, off sc1\n\t"
              "global_load_dwordx4 %2, %6, off sc1\n\t"
              "global_load_dwordx4 %3, %7, off sc1\n\t"
              "s_waitcnt vmcnt(0)"
              : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
              : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
              : "memory");
          auto fresh = [](const f32x4& v) {
            return __float_as_uint(v.x) != SB_SENT && __float_as_uint(v.y) != SB_SENT && __float_as_uint(v.z) != SB_SENT && __float_as_uint(v.w) != SB_SENT;
          };
          const bool ok = (!u0 || fresh(v0)) && (!u1 || fresh(v1)) && (!u2 || fresh(v2)) && (!u3 || fresh(v3));
          if (__all(ok) || (a.dbg & 2)) break;
          if (lds_peek(&abort_flag)) break;
          if (++spins > a.spin_limit) { abort_flag = 1 | (p << 8); break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (lds_peek(&abort_flag)) return false;
        nfail += spins > 0 ? 1 : 0;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (u0) acc += v0;
        if (u1) acc += v1;
        if (u2) acc += v2;
        if (u3) acc += v3;
        // the LP lanes of the position hold sums over disjoint senders
        // (the partners sit in one quad: DPP quad_perm swaps, not ds_bpermute round trips through the LDS crossbar)
        auto quad_xor = [](float v, int s2) {
          const int i = __float_as_int(v);
          return __int_as_float(s2 == 1 ? __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false)      // quad_perm [1,0,3,2]
                                        : __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
        };
#pragma unroll
        for (int s2 = 1; s2 < LP; s2 <<= 1) {
          acc.x += quad_xor(acc.x, s2);
          acc.y += quad_xor(acc.y, s2);
          acc.z += quad_xor(acc.z, s2);
          acc.w += quad_xor(acc.w, s2);
        }
#pragma unroll
        for (int r = 0; r < NT; ++r) {
          const int c = pt * NT + r;                                // component = row inside the quad
          sa[r] = c == 0 ? acc.x : (c == 1 ? acc.y : (c == 2 ? acc.z : acc.w));
        }
      }
      if (tracing && lane == 0) sb_trace[p * 8 + 1] = __builtin_amdgcn_s_memrealtime();
      if ((a.dbg & 256) && group == 0 && wv == 0 && lane == 0 && p >= SB_TRACE_ALL_FIRST && p < SB_TRACE_ALL_FIRST + SB_TRACE_ALL_STEPS && bx < 64)
        sb_trace[SB_TRACE_STEPS * 8 + (bx * SB_TRACE_ALL_STEPS + p - SB_TRACE_ALL_FIRST) * 2] = __builtin_amdgcn_s_memrealtime();
      // this step's operands were fetched a step ago (older than the gather's polls in the wave's in-order queue: already here); the
      // next step's go out NOW, at the start of the local work: a wave's loads return in order, so whatever is still in flight when
      // the next gather starts delays its first poll by the rest of a memory latency (issued behind the product, as until round 4,
      // that was ~1 us of every step - sweep_trace.py)
      f32x4 k0[NT], k1[NT];
      float dyv[NT], addAv[NT];
#pragma unroll
      for (int r = 0; r < NT; ++r) {
        k0[r] = cur[r].k0; k1[r] = cur[r].k1; dyv[r] = cur[r].dyv; addAv[r] = 0.f;
        if (p == 0 && live[r] && d.dh_last) addAv[r] = d.dh_last[(long)brow[r] * d.dh_last_ld + j];
      }
      fetch(p + 1, nxt);
      float ds[NT][4];
#pragma unroll
      for (int r = 0; r < NT; ++r) {
        ds[r][0] = ds[r][1] = ds[r][2] = ds[r][3] = 0.f;
        const float dh_state = sa[r] + addAv[r] + dirv[r];
        if (!cell) {
          if (live[r] && writer && d.dh0) d.dh0[(long)brow[r] * d.dh0_ld + j] = dh_state;
          continue;
        }
        float dir = 0.f;
        if (live[r]) {
          const bool m = (CELL == CELL_LSTM ? k0[r].w : (CELL == CELL_GRU ? k1[r].y : k0[r].y)) != 0.f;   // the step mask rides in the coefficients
          if (!m) {
            dir = dh_state;
            carry[r] += dyv[r];
          } else {
            const float dh = dh_state + dyv[r] + carry[r];
            carry[r] = 0.f;
            if constexpr (CELL == CELL_LSTM) {                      // k0 = {A, f, Co, m}, k1 = {Ci, Cf, Cg, 0}
              const float dct = dcv[r] + dh * k0[r].x;
              ds[r][0] = dct * k1[r].x;
              ds[r][1] = dct * k1[r].y;
              ds[r][2] = dct * k1[r].z;
              ds[r][3] = dh * k0[r].z;
              dcv[r] = dct * k0[r].y;
            } else if constexpr (CELL == CELL_GRU) {                // k0 = {Cz, Cr, E, E r}, k1 = {z, m, 0, 0}
              ds[r][0] = dh * k0[r].x;
              ds[r][1] = dh * k0[r].y;
              ds[r][2] = dh * k0[r].z;
              ds[r][3] = dh * k0[r].w;
              dir = dh * k1[r].x;
            } else {                                                // k0 = {1 - h^2, m, 0, 0}
              ds[r][0] = dh * k0[r].x;
            }
          }
          dirv[r] = dir;
        }
      }
      if (!cell) return false;
      if (tracing && lane == 0) sb_trace[p * 8 + 2] = __builtin_amdgcn_s_memrealtime();
      // the wave's ds image in its own LDS rows, gate slots in the order the recurrent kernel's column blocks take them
      // (GRU: z, r, r (.) d(a_hh); the input-side slot 2 does not multiply U), then read back as MFMA A operands: lane (li, lq) =
      // (row, gate slot) of unit ks - the same wave wrote it, its LDS accesses execute in order
#pragma unroll
      for (int r = 0; r < NT; ++r) {
        f32x4 img;
        img.x = ds[r][0];
        img.y = CELL == CELL_RNN ? 0.f : ds[r][1];
        img.z = CELL == CELL_LSTM ? ds[r][2] : (CELL == CELL_GRU ? ds[r][3] : 0.f);
        img.w = CELL == CELL_LSTM ? ds[r][3] : (CELL == CELL_GRU ? ds[r][2] : 0.f);   // GRU: the input-side slot rides along (k-row 3 of U^T is zero)
        *reinterpret_cast<f32x4*>(&tr[p % 3][wv][4 * plq + pt * NT + r][4 * ul]) = img;
      }
      f32x4 acc[NT];
#pragma unroll
      for (int t2 = 0; t2 < NT; ++t2) acc[t2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < UW; ++ks) {
        const float av = tr[p % 3][wv][li][4 * ks + lq];
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2) acc[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bwv[t2][ks], acc[t2], 0, 0, 0);
      }
#pragma unroll
      for (int t2 = 0; t2 < NT; ++t2) *reinterpret_cast<f32x4*>(&part[p & 1][wv][t2][lane * 4]) = acc[t2];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) lds_poke(&g_done[wv], p + 1);
      if (tracing && lane == 0) sb_trace[p * 8 + 3] = __builtin_amdgcn_s_memrealtime();
      // (ds goes to memory from the publish waves: in this wave the stores would sit in front of the next gather's polls in the
      // in-order memory counter - measured +0.76 us per step)
      return true;
    };
    for (int p = 0; p <= T; p += 2) {                                // p = T: only the gradient wrt the initial state
      if (!one_step(p, opA, opB)) break;
      if (!one_step(p + 1, opB, opA)) break;
    }
    if (lane == 0) {                                                 // poll statistics (diagnosis words 25-27, sweep_common.h)
      atomicAdd(a.err + 25, (unsigned)nfail);
      atomicAdd(a.err + 26, 1u);
    }
    if (writer && CELL == CELL_LSTM && !abort_flag) {
#pragma unroll
      for (int r = 0; r < NT; ++r)
        if (live[r]) d.dc[(long)brow[r] * H + j] = dcv[r];
    }
  } else {
    // ------------------------------------------------------------------------------------------ PUBLISH (one wave per output tile)
    const int nt_ = wv - 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, (int)a.xbytes, 0x00020000);
    const long my_blk = ((long)gj_ * G + gi_) * blk + (long)nt_ * 256 + lane * 4;   // block (row j, sender i), this wave's tile
    const u32x4 sent = {SB_SENT, SB_SENT, SB_SENT, SB_SENT};
    // ds leaves the chip once per step from the G workgroups of row i together: the NS x NT (wave, piece) units of a workgroup are dealt
    // round the G columns (unit m to column m % G).  Until round 4 column 0 wrote all of it - four more stores per lane and step than
    // its row mates, 1.39 against 1.11 us of local work (tests/tools/sweep_trace_all.py), and every cycle of hand-offs that passes
    // through a column-0 workgroup (all of them, within two steps) ran at ITS pace.
    bool mine[NS];
    bool writer = false;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      mine[k] = (k * NT + nt_) % G == gj_ && !(a.dbg & 16);
      writer = writer || mine[k];
    }
    f32x4 bsum[NS];                                                   // this lane's pieces summed over the steps: the bias gradient
#pragma unroll
    for (int k = 0; k < NS; ++k) bsum[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < T; ++p) {
      bool ok = true;
      for (int i = 0;; ++i) {
        const int v = lds_peek(&g_done[lane & 3]);
        if (__all(v >= p + 1)) break;
        if (lds_peek(&abort_flag)) { ok = false; break; }
        if (i > lds_limit) { abort_flag = 3 | (p << 8); ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) break;
      const bool tracing = (a.dbg & 128) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wv == 4 && p < SB_TRACE_STEPS;
      if (tracing && lane == 0) sb_trace[p * 8 + 4] = __builtin_amdgcn_s_memrealtime();
      f32x4 acc = *reinterpret_cast<const f32x4*>(&part[p & 1][0][nt_][lane * 4]);
      acc += *reinterpret_cast<const f32x4*>(&part[p & 1][1][nt_][lane * 4]);
      acc += *reinterpret_cast<const f32x4*>(&part[p & 1][2][nt_][lane * 4]);
      acc += *reinterpret_cast<const f32x4*>(&part[p & 1][3][nt_][lane * 4]);
      // retire the stores of the previous step (publish + sentinel, a whole exchange round old), then publish and re-arm.
      // (a counted wait - only the stores up to step p - 2 retired - shortens this wave's part of the step by 0.3 us but not the step)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");
      const long dst = ((long)group * SB_SLOTS + (p + 1) % SB_SLOTS) * slot_floats + my_blk;
      const long old = ((long)group * SB_SLOTS + (p + SB_SLOTS - 2) % SB_SLOTS) * slot_floats + my_blk;   // the block of step p - 3
      if (local) {                                                   // the group sits on one XCD: plain stores keep the lines in its L2
        if (!(a.dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), rsrc, (int)(dst * 4), 0, 0);
        if (p >= 3 && !(a.dbg & 1)) __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)(old * 4), 0, 0);
      } else {
        if (!(a.dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), rsrc, (int)(dst * 4), 0, 16);   // aux 16 = sc1
        if (p >= 3 && !(a.dbg & 1)) __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)(old * 4), 0, 16);
      }
      if (tracing && lane == 0) sb_trace[p * 8 + 5] = __builtin_amdgcn_s_memrealtime();
      if ((a.dbg & 256) && group == 0 && wv == 4 && lane == 0 && p >= SB_TRACE_ALL_FIRST && p < SB_TRACE_ALL_FIRST + SB_TRACE_ALL_STEPS && bx < 64)
        sb_trace[SB_TRACE_STEPS * 8 + (bx * SB_TRACE_ALL_STEPS + p - SB_TRACE_ALL_FIRST) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
      if (writer) {
        // this step's ds, [B, T, NS * H] row-major: 16-byte pieces (row, gate, 4 units), consecutive lanes on consecutive pieces of a
        // (row, gate) run; the next step's `s_waitcnt vmcnt(0)` retires them a whole exchange round later
        constexpr int PPR = NS * 4 * NT;                               // pieces per batch row
        const int step = T - 1 - p, t = d.reverse ? T - 1 - step : step;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          if (!mine[k]) continue;
          const int q = nt_ * 64 + lane + k * 64 * NT;                 // (16 * PPR = NS * 64 NT pieces: NS per lane, the same ones every step)
          const int row = q / PPR, rem = q % PPR, gate = rem / (4 * NT), u4 = rem % (4 * NT);
          const int sl = CELL == CELL_GRU ? (gate == 2 ? 3 : (gate == 3 ? 2 : gate)) : gate;   // image slot of this output slot
          const float* src = &tr[p % 3][u4 / NT][row][16 * (u4 % NT) + sl];
          const f32x4 v = {src[0], src[4], src[8], src[12]};
          if (b0 + row < B) *reinterpret_cast<f32x4*>(d.ds + ((long)(b0 + row) * T + t) * NS * H + (long)gate * H + gi_ * KU + 4 * u4) = v;
          bsum[k] += v;                                                // (rows beyond B carry zeros)
        }
      }
    }
    if (writer && d.db && !lds_peek(&abort_flag)) {
      // bias gradient.  A lane's NS pieces are the same (gate, unit quad) of different rows (64 NT k is a multiple of PPR), and so are
      // the lanes PPR apart: sum them in registers and across the wave first, then ONE atomic per bias element and wave (summing
      // every lane's pieces with atomics - 16 colliding adds per element - cost the GRU layers more than the pass over ds it replaces)
      constexpr int PPR = NS * 4 * NT;
      f32x4 tot = bsum[0];
#pragma unroll
      for (int k = 1; k < NS; ++k) tot += bsum[k];
#pragma unroll
      for (int s2 = PPR; s2 < 64; s2 <<= 1) {
        tot.x += __shfl_xor(tot.x, s2, 64); tot.y += __shfl_xor(tot.y, s2, 64);
        tot.z += __shfl_xor(tot.z, s2, 64); tot.w += __shfl_xor(tot.w, s2, 64);
      }
      if (lane < PPR) {
        const int rem = (nt_ * 64 + lane) % PPR, gate = rem / (4 * NT), u4 = rem % (4 * NT);
        const int j0 = gi_ * KU + 4 * u4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = tot[e];
          if (CELL == CELL_GRU) {                                      // slots z, r, x-part of h~, recurrent part of h~
            if (gate < 3) atomicAdd(d.db + (long)gate * H + j0 + e, v);
            if (gate != 2 && d.db_rec) atomicAdd(d.db_rec + (long)(gate == 3 ? 2 : gate) * H + j0 + e, v);
          } else {
            atomicAdd(d.db + (long)gate * H + j0 + e, v);
          }
        }
      }
    }
  }
  __syncthreads();
  if (abort_flag && tid == 0) {
    __hip_atomic_store(a.err, (unsigned)abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    swd_record(a.err, (unsigned)abort_flag, local ? 1 : 0);
    if (a.err_flag) __hip_atomic_store(reinterpret_cast<unsigned*>(a.err_flag), 0x3F800000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) swd_depart(a.err);
}

extern "C" int asr_debug_sweep_trace(unsigned long long* out, int n) {
  if (!out || n <= 0 || n > SB_TRACE_STEPS * 8 + 64 * SB_TRACE_ALL_STEPS * 2) return ASR_ERR_ARG;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sb_trace), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? ASR_OK : ASR_ERR_HIP;
}

// geometry: tiles per unit group (1 or 2) and the side G of the workgroup square of one (direction, batch tile) group
static bool sb_geometry(int B, int H, int ndir, int* nt, int* G) {
  if (H <= 0 || H % 16 != 0 || H > 256) return false;
  const long groups = (long)ndir * asr_cdiv(B, 16);
  static const int forced = getenv("ASR_SWEEP_BWD_NT") ? atoi(getenv("ASR_SWEEP_BWD_NT")) : 0;
  static const int xcd_env = getenv("ASR_SWEEP_XCD") ? atoi(getenv("ASR_SWEEP_XCD")) : 1;
  // XCD-local placement (one group per XCD) wants the square on the 32 compute units of one XCD, one workgroup each: take the
  // larger unit group when that makes it fit (DeepSpeech2 layer, H = 128: 8 x 8 workgroups 2.9 us per step, 4 x 4 2.35)
  if (!forced && xcd_env && groups <= 8 && H % 32 == 0 && (H / 16) * (H / 16) > 32 && (H / 32) * (H / 32) <= 32) {
    *nt = 2; *G = H / 32;
    return true;
  }
  for (int t = 1; t <= 2; ++t) {
    if (H % (16 * t) != 0) continue;
    if (forced && forced != t && H % (16 * forced) == 0) continue;
    const int g = H / (16 * t);
    if (groups * g * g <= 256 || t == 2) {
      if (groups * g * g > 512) return false;
      *nt = t; *G = g;
      return true;
    }
  }
  // H % 32 != 0 and the single-tile square does not fit one workgroup per CU: take it up to two per CU
  const int g = H / 16;
  if (groups * g * g > 512) return false;
  *nt = 1; *G = g;
  return true;
}

extern "C" long asr_rnn_sweep_bwd_ws_floats(int B, int H, int ndir) {
  int nt = 1, G = 1;
  if (!sb_geometry(B, H, ndir, &nt, &G)) return 32;
  const long groups = (long)ndir * asr_cdiv(B, 16);
  return groups * SB_SLOTS * G * G * nt * 256 + groups * G * G * 4 + 32;   // exchange slots, XCD ids (one piece per workgroup), error words
}

template <int NT>
static long sb_capacity(int rnn_type) {
  static long cache[3] = {0, 0, 0};
  if (cache[rnn_type] == 0) {
    const void* k = rnn_type == CELL_LSTM ? reinterpret_cast<const void*>(rnn_sweep_bwd_kernel<CELL_LSTM, NT>)
                  : rnn_type == CELL_GRU ? reinterpret_cast<const void*>(rnn_sweep_bwd_kernel<CELL_GRU, NT>)
                                         : reinterpret_cast<const void*>(rnn_sweep_bwd_kernel<CELL_RNN, NT>);
    cache[rnn_type] = asr_sweep_capacity(k, 64 * (4 + NT));
  }
  return cache[rnn_type];
}

extern "C" int asr_rnn_sweep_bwd_supported(int rnn_type, int B, int T, int H, int ndir) {
  if (rnn_type < 0 || rnn_type > 2 || B <= 0 || T < 2) return 0;
  if (ndir != 1 && ndir != 2) return 0;
  int nt, G;
  if (!sb_geometry(B, H, ndir, &nt, &G)) return 0;
  if ((long)B * T * H * 32 >= 2147483647L) return 0;                 // coefficient packs / dy are read through 32-bit buffer offsets
  // all workgroups must be resident together: ask the device, keep a quarter of its capacity free (see rnn_sweep.hip)
  const long wgs = (long)ndir * asr_cdiv(B, 16) * G * G;
  const long cap = nt == 1 ? sb_capacity<1>(rnn_type) : sb_capacity<2>(rnn_type);
  if (cap >= 0 && wgs * 4 > cap * 3) return 0;
  return 1;
}

template <int NT>
static void sb_launch(int rnn_type, dim3 grid, hipStream_t st, const SbArgs& a) {
  if (rnn_type == CELL_LSTM) hipLaunchKernelGGL((rnn_sweep_bwd_kernel<CELL_LSTM, NT>), grid, dim3(64 * (4 + NT)), 0, st, a);
  else if (rnn_type == CELL_GRU) hipLaunchKernelGGL((rnn_sweep_bwd_kernel<CELL_GRU, NT>), grid, dim3(64 * (4 + NT)), 0, st, a);
  else hipLaunchKernelGGL((rnn_sweep_bwd_kernel<CELL_RNN, NT>), grid, dim3(64 * (4 + NT)), 0, st, a);
}

// Same contract as asr_rnn_seq_bwd (rnn_bwd.hip) in one launch.  gs->direct / gs->dy_carry are not used (those carries
// live in registers).  ws: asr_rnn_sweep_bwd_ws_floats() floats; the uint32 at ws[ws_floats - 32] is non-zero after
// the call if a hand-off timed out; err_flag as for asr_rnn_sweep_fwd.
extern "C" int asr_rnn_sweep_bwd(const asr_rnn_seq* s, const asr_rnn_seq_grad* gs, float* ws, float* err_flag, void* stream) {
  ASR_CHECK(s && gs && ws, ASR_ERR_ARG, "asr_rnn_sweep_bwd: null argument");
  ASR_CHECK(asr_rnn_sweep_bwd_supported(s->rnn_type, s->B, s->T, s->H, s->ndir), ASR_ERR_UNSUPPORTED, "asr_rnn_sweep_bwd: shape not supported");
  const int B = s->B, T = s->T, H = s->H;
  const bool lstm = s->rnn_type == CELL_LSTM;
  const int NG = lstm ? 4 : (s->rnn_type == CELL_GRU ? 3 : 1);
  hipStream_t st = (hipStream_t)stream;
  int nt = 1, G = 1;
  sb_geometry(B, H, s->ndir, &nt, &G);
  const long groups = (long)s->ndir * asr_cdiv(B, 16);
  const long xslots = groups * SB_SLOTS * G * G * nt * 256;
  const long xfloats = xslots + groups * G * G * 4;
  ASR_CHECK(xfloats * 4 < 2147483647L, ASR_ERR_SHAPE, "asr_rnn_sweep_bwd: exchange buffer beyond 2 GB");
  SbArgs a{};
  a.B = B; a.T = T; a.H = H; a.G = G; a.mask = s->mask; a.dy = gs->dy; a.dy_ld = gs->dy_ld;
  a.xbuf = ws; a.xbytes = xfloats * 4;
  a.err = reinterpret_cast<unsigned*>(ws + xfloats);
  a.err_flag = err_flag;
  a.spin_limit = asr_rnn_sweep_spin_limit();
  a.dbg = getenv("ASR_SWEEP_DBG") ? atoi(getenv("ASR_SWEEP_DBG")) : 0;
  a.prio = asr_sweep_prio();
  a.rowxcd = getenv("ASR_SWEEP_BWD_ROWXCD") ? atoi(getenv("ASR_SWEEP_BWD_ROWXCD")) : 1;
  // sleep in front of a gather's first poll, in 10 ns ticks (see the gather).  las_small layer, chip-wide square of 64 workgroups per group, rows
  // on XCDs: 70 / 80 / 90 / 100 / 110 ticks = 9.44 / 9.33 / 9.35 / 9.40 / 9.53 ms per training step (9.76 before round 4, when the next step's
  // operand loads sat in front of the polls and were the delay), 2.66 us per dependent step alone at 90.
  a.delay = getenv("ASR_SWEEP_BWD_DELAY") ? atoi(getenv("ASR_SWEEP_BWD_DELAY")) : 85;
  ASR_CHECK(gs->dy, ASR_ERR_ARG, "asr_rnn_sweep_bwd: dy missing");
  ASR_CHECK(gs->dy_ld >= (long)s->ndir * H && (long)B * T * gs->dy_ld * 4 < 2147483647L, ASR_ERR_SHAPE,
            "asr_rnn_sweep_bwd: dy is read through 32-bit buffer offsets (B T dy_ld floats beyond 2 GB, or dy_ld < ndir H)");
  for (int d = 0; d < s->ndir; ++d) {
    ASR_CHECK(s->coef[d] && s->U[d] && (!lstm || gs->dc[d]), ASR_ERR_ARG, "asr_rnn_sweep_bwd: null buffer (dir %d): the BPTT sweep reads the coefficients the forward sweep wrote (s->coef)", d);
    ASR_CHECK(gs->ds[d] && (const float*)gs->ds[d] != s->coef[d], ASR_ERR_ARG, "asr_rnn_sweep_bwd: g->ds[%d] missing", d);
    ASR_CHECK(!s->rec_mult[d], ASR_ERR_UNSUPPORTED, "asr_rnn_sweep_bwd: recurrent dropout is not supported (use asr_rnn_seq_bwd)");
    SbDir& p = a.d[d];
    p.U = s->U[d]; p.ldu = s->ldu[d] ? s->ldu[d] : (long)NG * H; p.coef = s->coef[d]; p.ds = gs->ds[d]; p.db = gs->db[d]; p.db_rec = gs->db_rec[d];
    p.dh_last = gs->dh_last[d]; p.dh_last_ld = gs->dh_last_ld[d];
    p.dc = gs->dc[d]; p.dh0 = gs->dh0[d]; p.dh0_ld = gs->dh0_ld[d];
    p.reverse = s->reverse[d]; p.y_col = s->y_col[d];
  }
  {
    const size_t n = (size_t)xfloats;
    const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(sw_fill_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<uint32_t*>(ws), n, SB_SENT, a.err, 16,
                       (unsigned)(groups * G * G));
    ASR_LAUNCH_CHECK();
  }
  dim3 grid((unsigned)(G * G), (unsigned)asr_cdiv(B, 16), (unsigned)s->ndir);
  static const int xcd_env = getenv("ASR_SWEEP_XCD") ? atoi(getenv("ASR_SWEEP_XCD")) : 1;
  const long cap_all = nt == 1 ? sb_capacity<1>(s->rnn_type) : sb_capacity<2>(s->rnn_type);
  // XCD-local placement only with one workgroup per compute unit of the XCD (32): two per CU measured slower than the chip-wide
  // placement (las_small square of 64: 4.4 against 3.4 us per step) - an explicit bound, not the occupancy answer, which moves with
  // every change of the kernel's register count
  // (round 4, with the poll delay in place: the las_small square forced onto one XCD at two workgroups per CU = 3.02-3.07 us per step, the
  // chip-wide placement 2.99-3.07: what the L2-local hand-off saves, the doubled work per compute unit costs)
  if (xcd_env && groups <= 8 && cap_all > 0 && G * G <= 32 && (long)G * G * 4 <= (cap_all / 8) * 3) {
    a.xcd = 1; a.nx = G * G; a.ny = asr_cdiv(B, 16); a.ngroups = (int)groups;
    a.ids = ws + xslots;
    grid = dim3((unsigned)(8 * a.nx), 1, 1);
    // hand-offs that stay inside an XCD's L2 are short and their polls cheap: no sleep in front of them (DeepSpeech2 layer, H = 128:
    // 1.95 us per step polling at once, 2.5 with the steered sleep, which waits for the 90th percentile of a short wait)
    if (!getenv("ASR_SWEEP_BWD_DELAY")) a.delay = 0;
  }
  if (nt == 1) sb_launch<1>(s->rnn_type, grid, st, a);
  else sb_launch<2>(s->rnn_type, grid, st, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
