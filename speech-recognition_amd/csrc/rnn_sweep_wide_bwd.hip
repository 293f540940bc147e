// One-launch backward-through-time sweep of a WIDE bidirectional LSTM layer (H = 1024, B <= 64) under mixed precision on gfx950: the
// las_large encoder (las_large.yml: encoder_hidden_dim 1024; reference models/las.py:90-106 differentiated), whose BPTT ran as
// T' launches of rnn_step_bwd_staged_kernel (16.6 us each, 2 377 launches = 39.5 of the 129 ms las_large step: every launch re-reads
// the 16 MB of bf16 recurrent weights of both directions and 1 MB of ds per workgroup through L2).
//
// Decomposition.  dh_{t-1}[b, j] = sum_k ds_t[b, k] U[j, k] over the 4H gate columns k.  An all-gather of ds_t (1-D split over j)
// would move 512 KB per workgroup and step - four times the forward sweep's volume, which already bounds that kernel.  Instead the
// product is cut in BOTH dimensions over a 32 x 4 grid of workgroups per direction (128 per direction, 256 for the layer: one per CU):
// workgroup (i, j) owns the gate columns of unit group i (32 units x 4 gates = 128 columns) and the output units of column block j
// (256 units).  Per step it
//   1. gathers, for ITS 32 units, the partial sums of all 32 workgroups (., j(i)) of the column block j(i) = i / 8 its units lie in:
//      32 senders x [64 rows x 32 units] bf16 = 128 KB - the forward sweep's volume - and adds them up in f32: dh_t for its units;
//   2. does the element-wise gate gradients of its (row, unit) pairs (4 per thread; the 4 workgroups of a grid row repeat them -
//      cheaper than another hand-off) and leaves ds_t [64 x 128] as a bf16 image in LDS; column j also writes gate j of ds_t (f32) out;
//   3. multiplies: partial[256 units x 64 rows] = U[J_j, K_i] (resident bf16 A operands, 32 VGPRs per lane for the whole sequence)
//      x ds_t^T (B operands from the LDS image) on v_mfma_f32_16x16x32_bf16 - the product is taken transposed so that a lane's
//      accumulators are 4 consecutive units of ONE batch row and pack into the exchange piece without a transpose;
//   4. publishes its partial block as bf16 (32 KB: 8 slices x [64 rows x 4 pieces x 16 B], a slice = the 32 units one consumer row
//      needs, contiguous) with write-through stores and re-arms the block it wrote three steps earlier.
// bf16 partial sums: the exchange rounds each of the 32 partial sums of a dh element to bf16 (f32 would double the gather, the bound
// of this kernel); the operands of the product are bf16 already (mixed precision), the sum over senders and all gate math are f32.
//
// Hand-off: the sentinel protocol of the other sweeps (a NaN bf16 pair marks "not written yet", 16-byte L1-bypassing polls, the
// publisher re-arms).  The re-arm lag follows rnn_sweep_bwd.hip's argument: a workgroup gathers from the 32 workgroups of ONE grid
// column and is gathered by the 32 workgroups of 8 grid ROWS, so a completed gather of step p proves "my senders finished gather
// p - 1" and, because their senders together are the whole grid, "everybody finished gather p - 2": six slots, the block of step
// p - 3 is re-armed when step p is published.
// All 8 waves do every phase (gather, gate math, product, publish).  Measured per step at B = 64, T' = 499 (tests/tools/bench_wide_sweep.py):
// 9.9 us; 6.5 with neither waits nor publishes; publishes + sentinels cost ~3.5 us of visibility latency (16 MB of write-through
// stores per step chip-wide) that is NOT the in-order acknowledgement wait of the f32 sweeps: with 8 gather waves + 4 product/publish
// waves (768 threads, 166 VGPRs, the first probe delayed by the product's duration) the step took 10.0 us, so the simpler
// one-role structure stays.  Also measured: all four gates of ds from column 0 only (+1.9 us: everybody waits for the slowest
// workgroups), 4-byte sentinels (+0.65 us: partial-line write-through), no probe before the gather (+1.9 us of retry traffic).
// LSTM only; H = 1024; B <= 64; masks, chained final-state gradients and dh0 / dc as in asr_rnn_seq_bwd.
#include <stdlib.h>

#include "sweep_common.h"

#include "decoder_sweep_common.h"

#define WB_SLOTS 6
#define WB_GI 32                 // unit groups (grid rows): 32 units each
#define WB_GJ 4                  // column blocks (grid columns): 256 units each
#define WB_BLOCK_WORDS 8192      // one published block: 8 slices x 64 rows x 4 pieces x 4 words (32 KB)
#define WB_SLOT_WORDS ((long)WB_GJ * WB_GI * WB_BLOCK_WORDS)

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

struct WbDir {
  const float* U; long ldu;      // recurrent kernel [H, 4H]
  const float* saved;            // [B,T,4H] gate activations i, f, c~, o
  const float* cseq;             // [B,T,H]
  const float* c0; long c0_ld;   // initial cell state or null
  float* ds;                     // [B,T,4H] out (f32), or null when the bf16 images below are wanted instead
  unsigned short* ds16;          // optional: the same gate-sum gradients as a bf16 image [B T, 4H] (the A operand of dX = ds W^T)
  unsigned short* ds16T;         // optional: ... transposed, [4H][ld16T] with column t * B + b (the B operand of dW = x^T ds and dU = h^T ds:
  long ld16T;                    //   time-major columns, so that a step's 64 rows are ONE 128-byte run per gate column)
  float* db;                     // optional: bias gradient [4H] += column sums of ds (summed in registers over the steps)
  const float* dh_last; long dh_last_ld;
  float* dc;                     // [B,H] in: d/d final c, out: d/d initial c
  float* dh0; long dh0_ld;
  int reverse, y_col;
};
struct WbArgs {
  WbDir d[2];
  int B, T, H;
  const uint8_t* mask;
  const float* dy; long dy_ld;
  uint32_t* xbuf; long xbytes;   // [ndir][6 slots][4 column blocks][32 senders][8 slices][64 rows][4 pieces][4 words]
  unsigned* err; float* err_flag;
  int spin_limit, dbg, prio;
  int rowxcd;                    // block -> (row = x % 32, column = x / 32)
};

__device__ __forceinline__ bool wb_fresh(const u32x4& v) { return v.x != DS_SENT && v.y != DS_SENT && v.z != DS_SENT && v.w != DS_SENT; }
__device__ __forceinline__ float wb_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float wb_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t wb_pack2(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
// byte offset of 16-byte chunk c (0..15) of a row of the LDS ds image (256-byte rows, chunk index XORed with the row: the 16 rows a
// product fragment reads at one k offset fall on 16 different chunk positions)
__device__ __forceinline__ int wb_img(int row, int c) { return row * 256 + ((c ^ (row & 15)) << 4); }

__global__ __launch_bounds__(512) void rnn_sweepw_bwd_kernel(WbArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char img[2][64 * 256];     // ds_t as bf16 [row][k = gate * 32 + unit], by step parity
  // time-outs are raised during a step's gather and acted upon after that step's barrier, by ALL waves or none: one flag per step parity -
  // a fast wave that gives up in the gather of step p + 1 must not change what a slow wave reads right behind the barrier of step p
  __shared__ int abort_par[2];
  const WbDir& d = a.d[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  // grid row (unit group), grid column (output block).  Consecutive blocks go to consecutive XCDs (round-robin dispatch): with a.rowxcd
  // the 4 workgroups of a grid row - which gather the SAME 128 KB and fetch the same saved activations every step - share an XCD and its L2
  const int gi = a.rowxcd ? (int)(blockIdx.x & 31) : (int)(blockIdx.x >> 2), gj = a.rowxcd ? (int)(blockIdx.x >> 5) : (int)(blockIdx.x & 3);
  const int B = a.B, T = a.T, H = a.H;
  if (tid == 0) { abort_par[0] = 0; abort_par[1] = 0; swd_arrive(a.err); }
  swd_setprio(a.prio);

  // ---- element-wise role: lane = (row-in-wave rl, piece q, half sp); 4 consecutive units of one batch row ----
  const int rl = lane >> 3, q = (lane >> 1) & 3, sp = lane & 1;
  const int row = 8 * wv + rl;
  const int ug = 16 * sp + 4 * q;                                 // first of the thread's 4 units inside the group
  const int j0 = 32 * gi + ug;                                    // ... as hidden units
  const bool live = row < B;
  const bool writer = gj == 0;
  f32x4 dcv = {0.f, 0.f, 0.f, 0.f}, carry = {0.f, 0.f, 0.f, 0.f}, dirv = {0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};                              // this thread's share of the bias gradient (gate gj, its 4 units, its row)
  if (live) dcv = *reinterpret_cast<const f32x4*>(d.dc + (long)row * H + j0);

  // ---- product role: wave w owns output units 32 w .. 32 w + 31 of column block j (two M tiles); resident A operands ----
  // A[m = unit li of tile][k = 32 ks + 8 lq + e] = U[256 j + 32 w + 16 mt + li][gate ks, unit 32 i + 8 lq + e]
  u32x4 wreg[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float* ur = d.U + (long)(256 * gj + 32 * wv + 16 * mt + li) * d.ldu + (long)ks * H + 32 * gi + 8 * lq;
      const f32x4 u0 = *reinterpret_cast<const f32x4*>(ur), u1 = *reinterpret_cast<const f32x4*>(ur + 4);
      wreg[mt][ks] = (u32x4){wb_pack2(u0.x, u0.y), wb_pack2(u0.z, u0.w), wb_pack2(u1.x, u1.y), wb_pack2(u1.z, u1.w)};
    }

  if (tid == 0 && !swd_wait_all(a.err, a.spin_limit)) abort_par[0] = 15;   // the whole grid is resident before the first step (acted upon behind step 0's barrier)
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, (int)a.xbytes, 0x00020000);
  const long dir_words = (long)blockIdx.z * WB_SLOTS * WB_SLOT_WORDS;
  // gather: column block jc = i / 8, slice i % 8 of every sender's block; this thread takes senders 16 sp .. 16 sp + 15, piece (row, q)
  const long g_base = ((long)((gi >> 3) * WB_GI) * 8 + (gi & 7)) * 1024;                                   // words, wave-uniform
  const long g_lane = (long)(16 * sp) * 8 * 1024 + (long)(row * 4 + q) * 4;                                // words, per lane
  const unsigned g_voff = (unsigned)(g_lane * 4);                                                          // bytes
  // publish: block (column block j, sender i), slice = wave, piece (row = 16 nt + li, q = lq)
  const long p_off = ((long)(gj * WB_GI + gi) * 8 + wv) * 1024 + (long)(li * 4 + lq) * 4;

  struct Operands { f32x4 sv[4], ct, cp, dy; bool m; };
  auto fetch = [&](int p, Operands& o) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    o.sv[0] = o.sv[1] = o.sv[2] = o.sv[3] = z; o.ct = z; o.cp = z; o.dy = z; o.m = true;
    if (!live || p >= T) return;
    const int s = T - 1 - p, t = d.reverse ? T - 1 - s : s;
    const long bt = (long)row * T + t;
    o.m = a.mask ? a.mask[bt] != 0 : true;
    const float* sv = d.saved + bt * 4 * H + j0;
#pragma unroll
    for (int g = 0; g < 4; ++g) o.sv[g] = *reinterpret_cast<const f32x4*>(sv + (long)g * H);
    o.ct = *reinterpret_cast<const f32x4*>(d.cseq + bt * H + j0);
    if (s > 0) o.cp = *reinterpret_cast<const f32x4*>(d.cseq + ((long)row * T + (d.reverse ? t + 1 : t - 1)) * H + j0);
    else if (d.c0) o.cp = *reinterpret_cast<const f32x4*>(d.c0 + (long)row * d.c0_ld + j0);
    o.dy = *reinterpret_cast<const f32x4*>(a.dy + bt * a.dy_ld + d.y_col + j0);
  };
  Operands op;
  fetch(0, op);

  for (int p = 0; p <= T; ++p) {                                   // p = T: only the gradient wrt the initial state
    const bool cell = p < T;
    f32x4 dh = {0.f, 0.f, 0.f, 0.f};
    const lds_flag_t abort_now = lds_flag(&abort_par[p & 1]);
    if (p > 0) {
      // ---------------------------------------------------------------------------------------- gather (32 senders, 16 per thread)
      const uint32_t* src0 = a.xbuf + dir_words + (long)(p % WB_SLOTS) * WB_SLOT_WORDS + g_base;      // wave-uniform
      const uint32_t* src = src0 + g_lane;
      // One probe piece until it is fresh (256 workgroups re-reading 128 KB each per retry would sit in front of the very publishes
      // they wait for: issuing the whole gather first and probing only after a miss measured 11.8 against 9.9 us per step), then the
      // whole gather, 16 loads per lane in flight.
      bool ok2 = true;
      for (int sp2 = 0;; ++sp2) {
        u32x4 pv;
        const uint32_t* pp = src + 15l * 8 * 1024;                  // the last sender of this lane's half
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(pv) : "v"(pp) : "memory");
        if (__all(wb_fresh(pv)) || (a.dbg & 2)) break;
        if (*abort_now) { ok2 = false; break; }
        if (sp2 > a.spin_limit) { *abort_now = 1 | (p << 8); ok2 = false; break; }
        __builtin_amdgcn_s_sleep(4);
      }
      float acc8[8];
      for (int spins = 0; ok2; ++spins) {
        u32x4 v[16];
#pragma unroll
        for (int s16 = 0; s16 < 16; ++s16) {
          // wave-uniform sender base in SGPRs + ONE 32-bit lane offset for all 16 loads
          const uint32_t* sb = src0 + (long)s16 * 8 * 1024;
          asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=&v"(v[s16]) : "v"(g_voff), "s"(sb) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]),
                       "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])::"memory");
        bool ok = true;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc8[e] = 0.f;
#pragma unroll
        for (int s16 = 0; s16 < 16; ++s16) {
          ok = ok && wb_fresh(v[s16]);
          acc8[0] += wb_lo(v[s16].x); acc8[1] += wb_hi(v[s16].x); acc8[2] += wb_lo(v[s16].y); acc8[3] += wb_hi(v[s16].y);
          acc8[4] += wb_lo(v[s16].z); acc8[5] += wb_hi(v[s16].z); acc8[6] += wb_lo(v[s16].w); acc8[7] += wb_hi(v[s16].w);
        }
        if (__all(ok) || (a.dbg & 2)) break;
        if (*abort_now) { ok2 = false; break; }
        if (spins > a.spin_limit) { *abort_now = 2 | (p << 8); ok2 = false; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (ok2) {
        // the partner lane (sp ^ 1) summed the other 16 senders: this lane finishes the first (sp = 0) / last (sp = 1) four units
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float mine = sp ? acc8[4 + e] : acc8[e], send = sp ? acc8[e] : acc8[4 + e];
          dh[e] = mine + __shfl_xor(send, 1, 64);
        }
      }
    }
    // ------------------------------------------------------------------------------------------ gate gradients (rnn_bwd.hip bwd_finish)
    f32x4 dstate = dh + dirv;
    if (p == 0 && live && d.dh_last) dstate += *reinterpret_cast<const f32x4*>(d.dh_last + (long)row * d.dh_last_ld + j0);
    if (!cell) {
      if (live && writer && d.dh0 && !*abort_now) *reinterpret_cast<f32x4*>(d.dh0 + (long)row * d.dh0_ld + j0) = dstate;
      break;
    }
    f32x4 ds[4];
    ds[0] = ds[1] = ds[2] = ds[3] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live) {
      if (!op.m) {
        dirv = dstate;                                              // state carried unchanged through a masked step
        carry += op.dy;
      } else {
        const f32x4 dhv = dstate + op.dy + carry;
        carry = (f32x4){0.f, 0.f, 0.f, 0.f};
        dirv = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ig = op.sv[0][e], fg = op.sv[1][e], gg = op.sv[2][e], og = op.sv[3][e];
          const float tc = tanhf_(op.ct[e]);
          const float dct = dcv[e] + dhv[e] * og * (1.f - tc * tc);
          ds[0][e] = dct * gg * ig * (1.f - ig);
          ds[1][e] = dct * op.cp[e] * fg * (1.f - fg);
          ds[2][e] = dct * ig * (1.f - gg * gg);
          ds[3][e] = dhv[e] * tc * og * (1.f - og);
          dcv[e] = dct * fg;
        }
      }
    }
    {
      unsigned char* im = img[p & 1];
#pragma unroll
      for (int g = 0; g < 4; ++g) {                                // k = 32 g + ug .. + 3: chunk 4 g + ug / 8, half (ug / 4) & 1
        const bf16x4_t h = {(__bf16)ds[g][0], (__bf16)ds[g][1], (__bf16)ds[g][2], (__bf16)ds[g][3]};
        *reinterpret_cast<bf16x4_t*>(im + wb_img(row, 4 * g + (ug >> 3)) + ((ug >> 2) & 1) * 8) = h;
      }
    }
    if (live && !(a.dbg & 16)) {
      // the layer's ds: the four workgroups of a grid row hold the same values - column j stores gate j (one 16-byte store per lane;
      // all four gates from column 0 made its 32 workgroups 1.9 us per step slower than the rest, and everybody waits for the slowest)
      const int s = T - 1 - p, t = d.reverse ? T - 1 - s : s;
      const long o = ((long)row * T + t) * 4 * H + (long)gj * H + j0;
      const f32x4 mine = gj == 0 ? ds[0] : (gj == 1 ? ds[1] : (gj == 2 ? ds[2] : ds[3]));
      if (d.ds) *reinterpret_cast<f32x4*>(d.ds + o) = mine;
      if (d.ds16) {                                                  // the straight bf16 image: what asr_f32_to_bf16_image would make of ds
        const bf16x4_t h = {(__bf16)mine[0], (__bf16)mine[1], (__bf16)mine[2], (__bf16)mine[3]};
        *reinterpret_cast<bf16x4_t*>(d.ds16 + o) = h;
      }
      bsum += mine;
    }
    fetch(p + 1, op);                                              // next step's operands: in flight during the product and the next gather
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS-only barrier: the image is complete
    if (*abort_now) break;                                          // (written before the barrier only: every wave reads the same value)
    if (d.ds16T) {
      // the transposed bf16 image, read back out of the LDS image: thread (unit u, row quad rq) takes gate gj of unit u for rows
      // 4 rq .. 4 rq + 3 and stores them at [gate column][t B + row] - the 16 threads of a unit write one 128-byte run
      const int s = T - 1 - p, t = d.reverse ? T - 1 - s : s;
      const int u = tid >> 4, rq = tid & 15, k = 32 * gj + u;
      const unsigned char* im = img[p & 1];
      unsigned short v4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v4[e] = *reinterpret_cast<const unsigned short*>(im + wb_img(4 * rq + e, k >> 3) + (k & 7) * 2);
      unsigned short* o = d.ds16T + (long)(gj * H + 32 * gi + u) * d.ld16T + (long)t * B + 4 * rq;
      if ((B & 3) == 0) {
        if (4 * rq < B) *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)v4[0] | ((unsigned)v4[1] << 16), (unsigned)v4[2] | ((unsigned)v4[3] << 16));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * rq + e < B) o[e] = v4[e];
      }
    }
    // ------------------------------------------------------------------------------------------ product + publish
    f32x4 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      const unsigned char* im = img[p & 1];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 bfrag[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bfrag[nt] = *reinterpret_cast<const bf16x8*>(im + wb_img(16 * nt + li, 4 * ks + lq));
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const bf16x8 a8 = __builtin_bit_cast(bf16x8, wreg[mt][ks]);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, bfrag[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }
    // retire this wave's earlier stores (the previous step's publish and sentinel, a whole exchange round old), then publish
    // step p into slot p + 1 and re-arm the block of step p - 3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(a.dbg & 4)) {
      const long dst = dir_words + (long)((p + 1) % WB_SLOTS) * WB_SLOT_WORDS + p_off;
      const long old = dir_words + (long)((p + WB_SLOTS - 2) % WB_SLOTS) * WB_SLOT_WORDS + p_off;
      const u32x4 sent = {DS_SENT, DS_SENT, DS_SENT, DS_SENT};
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const u32x4 pc = {wb_pack2(acc[0][nt][0], acc[0][nt][1]), wb_pack2(acc[0][nt][2], acc[0][nt][3]),
                          wb_pack2(acc[1][nt][0], acc[1][nt][1]), wb_pack2(acc[1][nt][2], acc[1][nt][3])};
        __builtin_amdgcn_raw_buffer_store_b128(pc, rsrc, (int)((dst + nt * 256) * 4), 0, 16);       // aux 16 = sc1 (write-through)
      }
      if (p >= 3 && !(a.dbg & 1)) {      // (whole 16-byte pieces: re-arming one word per piece - enough for the freshness test - measured 0.65 us per step SLOWER)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)((old + nt * 256) * 4), 0, 16);
      }
    }
  }
  __syncthreads();
  const int abort_flag = abort_par[0] | abort_par[1];
  if (live && writer && !abort_flag) *reinterpret_cast<f32x4*>(d.dc + (long)row * H + j0) = dcv;
  if (d.db && !abort_flag) {
    // bias gradient: the lanes 8 apart hold other rows of the same 4 units - sum them in the wave, then one atomic per element and wave
    // (rows beyond B carry zeros)
#pragma unroll
    for (int s2 = 8; s2 < 64; s2 <<= 1) {
      bsum.x += __shfl_xor(bsum.x, s2, 64); bsum.y += __shfl_xor(bsum.y, s2, 64);
      bsum.z += __shfl_xor(bsum.z, s2, 64); bsum.w += __shfl_xor(bsum.w, s2, 64);
    }
    if (rl == 0) {
      float* o = d.db + (long)gj * H + j0;
      atomicAdd(o, bsum.x); atomicAdd(o + 1, bsum.y); atomicAdd(o + 2, bsum.z); atomicAdd(o + 3, bsum.w);
    }
  }
  if (abort_flag && tid == 0) {
    __hip_atomic_store(a.err, (unsigned)abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    swd_record(a.err, (unsigned)abort_flag, 0);
    if (a.err_flag) __hip_atomic_store(reinterpret_cast<unsigned*>(a.err_flag), 0x3F800000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) swd_depart(a.err);
}

extern "C" int asr_rnn_sweep_wide_bwd_supported(int rnn_type, int B, int T, int H, int ndir) {
  if (rnn_type != 0 || B <= 0 || B > 64 || T < 2 || H != 1024) return 0;
  if (ndir != 1 && ndir != 2) return 0;
  return 1;
}

extern "C" long asr_rnn_sweep_wide_bwd_ws_floats(int B, int H, int ndir) {
  (void)B; (void)H;
  return (long)ndir * WB_SLOTS * WB_SLOT_WORDS + 32;
}

// Same contract as asr_rnn_sweep_bwd (rnn_sweep_bwd.hip) for a wide LSTM layer under mixed precision: reads the saved gate
// activations and cell states of the forward pass (s->saved, s->cseq - NOT coefficient packs), writes the gate-sum gradients OUT OF
// PLACE to gs->ds ([B,T,4H]; the grid row's workgroups all read `saved`), gs->dc in/out, gs->dh0 out; bias gradients are not summed
// here (gs->db ignored).  ds, the recurrent weights and the exchanged partial sums are rounded to bf16, everything else is f32.
extern "C" int asr_rnn_sweep_wide_bwd(const asr_rnn_seq* s, const asr_rnn_seq_grad* gs, float* ws, float* err_flag, void* stream) {
  ASR_CHECK(s && gs && ws, ASR_ERR_ARG, "asr_rnn_sweep_wide_bwd: null argument");
  ASR_CHECK(asr_rnn_sweep_wide_bwd_supported(s->rnn_type, s->B, s->T, s->H, s->ndir), ASR_ERR_UNSUPPORTED, "asr_rnn_sweep_wide_bwd: shape not supported");
  const int wgs = WB_GI * WB_GJ * s->ndir;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
    ASR_CHECK(cus >= wgs, ASR_ERR_UNSUPPORTED, "asr_rnn_sweep_wide_bwd: needs %d compute units, device has %d", wgs, cus);
  const int B = s->B, T = s->T, H = s->H;
  hipStream_t st = (hipStream_t)stream;
  WbArgs a{};
  a.B = B; a.T = T; a.H = H;
  a.mask = s->mask; a.dy = gs->dy; a.dy_ld = gs->dy_ld;
  ASR_CHECK(gs->dy && gs->dy_ld % 4 == 0 && ((uintptr_t)gs->dy & 15) == 0, ASR_ERR_ARG, "asr_rnn_sweep_wide_bwd: dy must be 16-byte aligned with ld % 4 == 0");
  const long xwords = (long)s->ndir * WB_SLOTS * WB_SLOT_WORDS;
  a.xbuf = reinterpret_cast<uint32_t*>(ws); a.xbytes = xwords * 4;
  ASR_CHECK(a.xbytes < (1l << 31), ASR_ERR_SHAPE, "asr_rnn_sweep_wide_bwd: exchange buffer exceeds a buffer descriptor");
  a.err = reinterpret_cast<unsigned*>(ws + xwords);
  a.err_flag = err_flag;
  a.spin_limit = asr_rnn_sweep_spin_limit();
  a.dbg = getenv("ASR_SWEEP_DBG") ? atoi(getenv("ASR_SWEEP_DBG")) : 0;
  a.prio = asr_sweep_prio();
  a.rowxcd = getenv("ASR_SWEEP_WIDE_BWD_ROWXCD") ? atoi(getenv("ASR_SWEEP_WIDE_BWD_ROWXCD")) : 1;
  for (int d = 0; d < s->ndir; ++d) {
    ASR_CHECK(s->saved[d] && s->cseq[d] && s->U[d] && gs->dc[d], ASR_ERR_ARG, "asr_rnn_sweep_wide_bwd: null buffer (dir %d)", d);
    ASR_CHECK(gs->ds[d] || gs->ds16[d] || gs->ds16T[d], ASR_ERR_ARG, "asr_rnn_sweep_wide_bwd: no destination for the gate-sum gradients: ds, ds16 or ds16T (dir %d)", d);
    ASR_CHECK(gs->ds[d] != s->saved[d], ASR_ERR_ARG, "asr_rnn_sweep_wide_bwd: ds must not alias the saved activations (dir %d)", d);
    ASR_CHECK((((uintptr_t)gs->ds16[d] | (uintptr_t)gs->ds16T[d]) & 7) == 0 && (!gs->ds16T[d] || (gs->ds16T_ld % 4 == 0 && gs->ds16T_ld >= (long)s->B * s->T)),
              ASR_ERR_ARG, "asr_rnn_sweep_wide_bwd: ds16 / ds16T must be 8-byte aligned, ds16T_ld a multiple of 4 and >= B T (dir %d)", d);
    ASR_CHECK(!s->rec_mult[d], ASR_ERR_UNSUPPORTED, "asr_rnn_sweep_wide_bwd: recurrent dropout is not supported (use asr_rnn_seq_bwd)");
    ASR_CHECK(s->ldu[d] % 4 == 0 && s->y_col[d] % 4 == 0 && (!s->c0[d] || s->c0_ld[d] % 4 == 0) && (!gs->dh_last[d] || gs->dh_last_ld[d] % 4 == 0) &&
                  (!gs->dh0[d] || gs->dh0_ld[d] % 4 == 0),
              ASR_ERR_SHAPE, "asr_rnn_sweep_wide_bwd: leading dimensions / column offsets must be multiples of 4 (dir %d)", d);
    // every one of these is read or written with 16-byte vector accesses: a misaligned view (an odd-offset slice of a state tensor)
    // must come back as an argument error, not as a fault on the device
    const void* vec[] = {s->U[d], s->saved[d], s->cseq[d], s->c0[d], gs->ds[d], gs->dc[d], gs->dh_last[d], gs->dh0[d]};      // (null = unused: passes)
    for (const void* q : vec)
      ASR_CHECK(((uintptr_t)q & 15) == 0, ASR_ERR_ARG, "asr_rnn_sweep_wide_bwd: U, saved, cseq, c0, ds, dc, dh_last and dh0 must be 16-byte aligned (dir %d)", d);
    WbDir& p = a.d[d];
    p.U = s->U[d]; p.ldu = s->ldu[d];
    p.saved = s->saved[d]; p.cseq = s->cseq[d];
    p.c0 = s->c0[d]; p.c0_ld = s->c0_ld[d];
    p.ds = gs->ds[d];
    p.ds16 = static_cast<unsigned short*>(gs->ds16[d]); p.ds16T = static_cast<unsigned short*>(gs->ds16T[d]); p.ld16T = gs->ds16T_ld;
    p.db = gs->db[d];
    p.dh_last = gs->dh_last[d]; p.dh_last_ld = gs->dh_last_ld[d];
    p.dc = gs->dc[d];
    p.dh0 = gs->dh0[d]; p.dh0_ld = gs->dh0_ld[d];
    p.reverse = s->reverse[d]; p.y_col = s->y_col[d];
  }
  {
    const size_t n = (size_t)xwords;
    hipLaunchKernelGGL(sw_fill_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<uint32_t*>(ws), n, DS_SENT, a.err, 16, (unsigned)wgs);
    ASR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(rnn_sweepw_bwd_kernel, dim3((unsigned)(WB_GI * WB_GJ), 1, (unsigned)s->ndir), dim3(512), 0, st, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
