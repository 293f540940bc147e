// One-launch forward sweep of the LAS decoder under teacher forcing (las.py:267-292 AttendAndSpeller.call looped by
// las.py:368-377): all U steps of {attention, decoder LSTM 0, decoder LSTM 1} in ONE kernel on gfx950.
//
// With one launch per kernel a decoder step is 4 dependent launches (scores, softmax+context, two cells): ~32 us, of which
// the arithmetic is a few hundred nanoseconds - the rest is launch boundaries and every kernel re-reading its operands
// (the attention streams Kq and enc: 24.5 MB per step on las_small) from L2 / Infinity Cache.  Here 256 workgroups stay
// resident for the whole sequence and keep everything that does not change between steps on chip:
//   * attention operands: workgroup w owns batch row b = w / 8 and time chunk c = w % 8 (<= 32 encoder frames): its
//     [TC x Hd] slice of Kq and [TC x D] slice of enc live in LDS (96 KB at las_small) for all U steps;
//   * cell weights: workgroup w also owns, for decoder layer w / 128 and batch tile (w / 64) % 2, the 4 hidden units
//     q = w % 64 x 4 gates: its columns of the packed kernels (asr_rnn_pack images) live in registers.
// A step is four hand-offs between workgroups (all of the forward sweep's kind: self-validating 16-byte pieces, a sentinel
// NaN pattern marks "not written yet", 4 slots, the publisher re-arms what it wrote two steps earlier - see rnn_sweep.hip):
//   h1(i-1) --> [A] every (b, c) workgroup: scores of its chunk, chunk-local softmax, partial context  --> partial (m, l, ctx_c)
//           --> [S] the 8 workgroups of a row each combine one 1/8 slice of the context (flash-style rescale)  --> ctx slice
//           --> [L0] layer-0 workgroups: gates = pre0 (embedding half, batched outside) + drop(ctx) Wc + h1(i-1) U0 --> h0, c0
//           --> [L1] layer-1 workgroups: gates = b1 + drop(y0) W1 + h0 U1                                            --> h1(i), c1(i)
// (las.py:285-288: the state is threaded layer to layer inside a step and from the last layer to layer 0 of the next step;
//  pad-token rows carry their state and emit zeros.)  Everything the backward pass and the batched GEMMs after the loop
// need is written where the per-step kernels write it: p, ctx, the gate activations, y / h / c of both layers.
// Wave roles (320 threads): waves 0-3 gather, multiply and reduce and never store to global memory; wave 4 does the
// gate math, publishes and writes the saved tensors (a wave's loads and stores retire through one in-order counter, so a
// publishing wave must not be the one that polls).  Roles hand over through LDS counters.
// Restrictions (the caller falls back to the per-step kernels otherwise): LSTM, 2 decoder layers, teacher forcing, B <= 32,
// Hd % 16 == 0 and <= 256, D % 32 == 0 and <= 512, T' <= 256.  Every spin is bounded.
#include <stdlib.h>

#include "common.h"

#include "decoder_sweep_common.h"

struct DsArgs {
  int B, U, T2, Hd, D, TC, FS;            // FS = D / 8 features per context slice
  int Bs, b0;                               // row stride of the step-major tensors / first batch row of this launch (batches > 32 run in passes)
  const float* Kq; const float* enc; const float* s0; const uint8_t* mask;
  const float* h_init; const float* c_init;
  const float* Wp0; int KSt0, kc0, kh0;    // layer 0 image: block offsets of the context and state segments
  const float* Wp1; int KSt1, kx1, kh1;    // layer 1 image: input and state segments
  const float* pre0; const float* bias1;
  const uint8_t* tokmask;
  const uint32_t* seed; float rate; uint32_t stream0; uint32_t stream_step;
  float* p; float* ctx; float* hin; float* cin;
  float* y0; float* saved0; float* h0; float* c0;
  float* y1; float* saved1;
  float* xbuf; long xbytes;
  long o_h1, o_c1, o_h0, o_c0, o_part, o_ctx, slot_floats;   // float offsets of the six exchanges inside a slot
  unsigned* err; float* err_flag;
  int spin_limit, delay;
  int dbg;                                  // timing experiments only (ASR_DECODER_SWEEP_DBG): 2 = gathers do not wait
  int prio;                                 // s_setprio level of every wave
};

// LONG: the general instance - chunks of more than DS_MAXTC frames (T' > 256: streamed-frame code) and / or a batch that runs in
// passes (row stride != rows of this launch, row offset != 0).  The kernel sits at the 256-register limit: the common instance
// (T' <= 256, B <= 32: the benchmark's) must not pay for the general paths in spills, so they are compiled only into LONG
// Stage timeline (timing aid, ASR_DECODER_SWEEP_TRACE=1; read back with asr_debug_decoder_trace, tests/tools/decoder_trace.py): s_memrealtime stamps
// (10 ns ticks) of two workgroups - block 0 (attention chunk (0, 0) + a layer-0 cell) and block 128 (attention chunk (16, 0) + a layer-1
// cell) - 16 per step: gather wave 0: 0 step entered, 1 h1 gathered (+ the layer-0 state product), 2 chunk scores visible, 3 partial
// context handed to the publish wave, 4 the row's partials gathered, 5 slice combined, 6 cell operands gathered, 7 partial sums in LDS;
// publish wave: 8 partial published, 9 slice published, 10 cell state published, 11 saved tensors stored.  Compiled into an
// instance of its own (TRACE): the production kernel sits at the 256-register limit and must not carry the stamps.
#define DSF_TRACE_STEPS 128
__device__ unsigned long long dsf_trace[2 * DSF_TRACE_STEPS * 16];
#define DS_STAMP(k)                                                                                             \
  do {                                                                                                          \
    if (TRACE && twg >= 0 && lane == 0 && i < DSF_TRACE_STEPS) dsf_trace[(twg * DSF_TRACE_STEPS + i) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)

template <bool LONG, bool TRACE = false>
__global__ __launch_bounds__(320) void decoder_sweep_fwd_kernel(DsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int twg = !TRACE ? -1 : ((blockIdx.x == 0 && (wv == 0 || wv == 4)) ? 0 : ((blockIdx.x == 128 && (wv == 0 || wv == 4)) ? 1 : -1));
  const bool pub_wave = wv == 4;
  const int li = lane & 15, lq = lane >> 4;
  const int w = blockIdx.x;
  const int B = a.B, U = a.U, T2 = a.T2, Hd = a.Hd, D = a.D, TC = a.TC, FS = a.FS;
  const int Bs_ = LONG ? a.Bs : B, b0_ = LONG ? a.b0 : 0;
  const int Q = Hd >> 2, KBH = Hd >> 4, KBC = D >> 4;
  // attention role
  // (block -> role placements measured in round 4, consecutive blocks going to consecutive XCDs: the 8 chunk workgroups of a batch row on one
  // XCD 15.8 -> 16.4 us per decoder step, the 64 slice workgroups of a (layer, tile) on two XCDs 15.7 -> 16.6: both lose HERE and win in the
  // backward sweep, which has them)
  const int ab = w >> 3, ac = w & 7;
  const bool attn = ab < B;
  const int t_lo = ac * TC, nt = max(0, min(TC, T2 - t_lo));      // this chunk's frames
  const int atile = ab >> 4, arow = ab & 15;
  // cell role
  const int layer = w >> 7, tile = (w >> 6) & 1, q = w & 63;
  const bool cell = q < Q && tile * 16 < B;
  // LDS carve (floats)
  const int KQLD = Hd + 4, ENLD = D + 4;
  float* kq_s = lds;                                   // [TC][KQLD]
  float* enc_s = kq_s + DS_MAXTC * KQLD;               // [TC][ENLD]
  float* s0m = enc_s + DS_MAXTC * ENLD;                // [64] score offsets: s0 - 1e9 (1 - mask)
  float* hrow = s0m + DS_MAXTC2;                       // [4 waves][Hd] the attention row of h1, one copy per gather wave
  float* esc = hrow + 4 * 256;                         // [64] scores of the chunk
  float* pt = esc + DS_MAXTC2;                         // [4 waves][64] chunk-local exp(e - m_c), one copy per gather wave
  float* pbuf = pt + 4 * DS_MAXTC2;                    // [4 + D] partial to publish: {m_c, l_c, 0, 0}, ctx_c
  float* sl = pbuf + 4 + 512;                          // [8][4 + FS] gathered stats + slices (FS <= 64)
  float* cslice = sl + 8 * 68;                         // [64] combined context slice
  float* pn = cslice + 64;                             // [64] normalised probabilities of the chunk
  float* part = pn + DS_MAXTC2;                        // [2][4][16 * 17] matrix partial sums (double buffered by step parity)
  float* hblk = part + 2 * 4 * 272;                    // [64] previous state of the owned (row, unit) pairs
  float* cblk = hblk + 64;                             // [64] previous cell state of the owned pairs
  int* flags = reinterpret_cast<int*>(cblk + 64);      // abort | cC (slice ready) | per-wave progress words of the four hand-over points
  const lds_flag_t abort_flag = lds_flag(flags);          // (LDS-typed: ds_read / ds_write, not flat accesses - sweep_common.h)
  const lds_flag_t cC = abort_flag + 1;
  const lds_flag_t cE = abort_flag + 4, cA = abort_flag + 8, cB = abort_flag + 12, cD = abort_flag + 16;   // scores, partial, slices gathered, sums
  if (tid < 20) flags[tid] = 0;
  if (tid < DS_MAXTC2) esc[tid] = -INFINITY;
  if (tid == 0) swd_arrive(a.err);                       // start handshake (sweep_common.h)
  swd_setprio(a.prio);

  // ---- resident operands ----
  if (attn) {
    const int ntr = min(nt, DS_MAXTC);                  // frames resident in LDS; frames [ntr, nt) are read from memory every step
    const float* kqg = a.Kq + ((long)ab * T2 + t_lo) * Hd;
    const float* eng = a.enc + ((long)ab * T2 + t_lo) * D;
    for (int i = tid; i < ntr * (Hd >> 2); i += 320) {
      const int t = i / (Hd >> 2), k4 = i % (Hd >> 2);
      *reinterpret_cast<float4*>(kq_s + t * KQLD + 4 * k4) = *reinterpret_cast<const float4*>(kqg + (long)t * Hd + 4 * k4);
    }
    for (int i = tid; i < ntr * (D >> 2); i += 320) {
      const int t = i / (D >> 2), k4 = i % (D >> 2);
      *reinterpret_cast<float4*>(enc_s + t * ENLD + 4 * k4) = *reinterpret_cast<const float4*>(eng + (long)t * D + 4 * k4);
    }
    if (tid < DS_MAXTC2) {
      float v = -INFINITY;                              // frames beyond the chunk take no part
      if (tid < nt) {
        const long bt = (long)ab * T2 + t_lo + tid;
        v = (a.s0 ? a.s0[bt] : 0.f) - 1e9f * (1.0f - (a.mask[bt] ? 1.0f : 0.0f));
      }
      s0m[tid] = v;
    }
  }
  const int kbase_c = layer == 0 ? a.kc0 : a.kx1, kbase_h = layer == 0 ? a.kh0 : a.kh1;
  const int nbx = layer == 0 ? KBC : KBH;              // K blocks of the input segment (context / y0)
  const int wave = pub_wave ? 0 : wv;
  float4 bwx[DS_MAXCB], bwh[DS_MAXHB];
#pragma unroll
  for (int i = 0; i < DS_MAXCB; ++i) bwx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < DS_MAXHB; ++i) bwh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cell && !pub_wave) {
    const float4* wp = reinterpret_cast<const float4*>(layer == 0 ? a.Wp0 : a.Wp1) + (long)q * (layer == 0 ? a.KSt0 : a.KSt1) * 64 + lane;
#pragma unroll
    for (int i = 0; i < DS_MAXCB; ++i) {
      const int jb = wave + 4 * i;
      if (jb < nbx) bwx[i] = wp[(long)(kbase_c + jb) * 64];
    }
#pragma unroll
    for (int i = 0; i < DS_MAXHB; ++i) {
      const int jb = wave + 4 * i;
      if (jb < KBH) bwh[i] = wp[(long)(kbase_h + jb) * 64];
    }
  }
  if (tid == 0 && !swd_wait_all(a.err, a.spin_limit)) *abort_flag = 15;   // the whole grid is resident before the first step
  __syncthreads();

  float* xb = a.xbuf;
  const int lds_limit = a.spin_limit > (1 << 20) ? a.spin_limit : (a.spin_limit << 4);
  const float scale = a.rate > 0.f ? 1.f / (1.f - a.rate) : 1.f;
  const uint32_t thresh = asr_drop_threshold(a.rate);
  const uint32_t seedv = (a.seed && a.rate > 0.f) ? a.seed[0] : 0u;

  if (!pub_wave) {
    // ================================================================================================= GATHER waves
    for (int i = 0; i < U; ++i) {
      const long slot_prev = (long)((i + 3) & 3) * a.slot_floats, slot_cur = (long)(i & 3) * a.slot_floats;
      DS_STAMP(0);
      // ---- (1) h1 of the previous step: the attention row (every wave keeps its own LDS copy) and, for layer-0 cells, the tile ----
      f32x4 hv[5];
      f32x4 acc_h = {0.f, 0.f, 0.f, 0.f};
      {
        unsigned o5[5];
        bool use[5];
        const float* h1x = xb + slot_prev + a.o_h1;            // wave-uniform base
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int jb = wave + 4 * k;
          use[k] = i > 0 && cell && layer == 0 && jb < KBH;
          o5[k] = use[k] ? (unsigned)((((long)tile * Q + 4 * jb + lq) * 16 + li) * 16) : 0u;
        }
        use[4] = i > 0 && attn && lane < Q;
        o5[4] = use[4] ? (unsigned)((((long)atile * Q + lane) * 16 + arow) * 16) : 0u;
        if (i > 0) {
          if (!ds_gather5(h1x, o5, use, hv, abort_flag, a.spin_limit, a.delay, 1 | (i << 8))) break;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int jb = wave + 4 * k;
            hv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (cell && layer == 0 && jb < KBH && tile * 16 + li < B)
              hv[k] = *reinterpret_cast<const f32x4*>(a.h_init + (long)(tile * 16 + li) * Hd + 16 * jb + 4 * lq);
          }
          hv[4] = (attn && lane < Q) ? *reinterpret_cast<const f32x4*>(a.h_init + (long)ab * Hd + 4 * lane) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (attn && lane < Q) *reinterpret_cast<f32x4*>(hrow + wave * 256 + 4 * lane) = hv[4];
        if (cell && layer == 0) {
          // the owned pairs' previous state for the gate wave: block q, rows 0..15 (held by the wave / lanes that loaded slice q)
#pragma unroll
          for (int k = 0; k < 4; ++k)                            // (static register indices: a run-time pick would go through scratch)
            if (k == (q >> 4) && wave == ((q >> 2) & 3) && lq == (q & 3)) *reinterpret_cast<f32x4*>(hblk + li * 4) = hv[k];
#pragma unroll
          for (int k = 0; k < DS_MAXHB; ++k) {
            if (wave + 4 * k < KBH) {
              acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].x, bwh[k].x, acc_h, 0, 0, 0);
              acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].y, bwh[k].y, acc_h, 0, 0, 0);
              acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].z, bwh[k].z, acc_h, 0, 0, 0);
              acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].w, bwh[k].w, acc_h, 0, 0, 0);
            }
          }
        }
      }
      DS_STAMP(1);
      // ---- (2) scores of the chunk, chunk-local softmax, partial context ----
      if (attn) {
        // thread (t = tid / 8, kg = tid % 8): partial dot over the k's congruent to kg (float4 granularity)
        const int t = tid >> 3, kg = tid & 7;
        float dot = 0.f;
        if (t < nt) {
          const float* kr = kq_s + t * KQLD;
          const float* hr = hrow + wave * 256;
          for (int k4 = kg; k4 < (Hd >> 2); k4 += 8) {
            const float4 kv = *reinterpret_cast<const float4*>(kr + 4 * k4);
            const float4 hh = *reinterpret_cast<const float4*>(hr + 4 * k4);
            dot += hh.x * kv.x + hh.y * kv.y + hh.z * kv.z + hh.w * kv.w;
          }
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        dot += __shfl_xor(dot, 4, 64);
        if (kg == 0) esc[t] = dot + s0m[t];
        if (LONG && nt > DS_MAXTC) {                              // chunks longer than the LDS holds (T' > 256): the other frames' keys from memory
          const int t2 = DS_MAXTC + t;
          float dot2 = 0.f;
          if (t2 < nt) {                                  // (addresses rebuilt here: nothing of the rare path stays live across the step)
            const float* kr = a.Kq + ((long)ab * T2 + t_lo + t2) * Hd;
            const float* hr = hrow + wave * 256;
            for (int k4 = kg; k4 < (Hd >> 2); k4 += 8) {
              const float4 kv = *reinterpret_cast<const float4*>(kr + 4 * k4);
              const float4 hh = *reinterpret_cast<const float4*>(hr + 4 * k4);
              dot2 += hh.x * kv.x + hh.y * kv.y + hh.z * kv.z + hh.w * kv.w;
            }
          }
          dot2 += __shfl_xor(dot2, 1, 64);
          dot2 += __shfl_xor(dot2, 2, 64);
          dot2 += __shfl_xor(dot2, 4, 64);
          if (kg == 0) esc[t2] = dot2 + s0m[t2];
        }
        ds_mark(cE, wave, i + 1);
        if (!ds_wait4(cE, i + 1, abort_flag, lds_limit, 5 | (i << 8))) break;
        DS_STAMP(2);
        // every lane evaluates the chunk's softmax statistics itself from broadcast LDS reads of the <= 32 scores (no cross-lane
        // traffic: a chain of ten dependent wave shuffles costs more than 32 hardware exponentials); lanes 0-31 of every wave leave
        // the weights in the wave's own LDS copy, from where the context loop reads them as broadcasts
        float m = -INFINITY;
#pragma unroll
        for (int t4 = 0; t4 < DS_MAXTC / 4; ++t4) {
          const float4 e4 = *reinterpret_cast<const float4*>(esc + 4 * t4);
          m = fmaxf(fmaxf(m, fmaxf(e4.x, e4.y)), fmaxf(e4.z, e4.w));
        }
        if (LONG && nt > DS_MAXTC) {                              // (T' > 256: the second half of the chunk; wave-uniform)
#pragma unroll
          for (int t4 = DS_MAXTC / 4; t4 < DS_MAXTC2 / 4; ++t4) {
            const float4 e4 = *reinterpret_cast<const float4*>(esc + 4 * t4);
            m = fmaxf(fmaxf(m, fmaxf(e4.x, e4.y)), fmaxf(e4.z, e4.w));
          }
        }
        float l = 0.f;
#pragma unroll
        for (int t4 = 0; t4 < DS_MAXTC / 4; ++t4) {
          const float4 e4 = *reinterpret_cast<const float4*>(esc + 4 * t4);
          l += (e4.x == -INFINITY ? 0.f : fast_exp_(e4.x - m)) + (e4.y == -INFINITY ? 0.f : fast_exp_(e4.y - m)) +
               (e4.z == -INFINITY ? 0.f : fast_exp_(e4.z - m)) + (e4.w == -INFINITY ? 0.f : fast_exp_(e4.w - m));
        }
        if (LONG && nt > DS_MAXTC) {
#pragma unroll
          for (int t4 = DS_MAXTC / 4; t4 < DS_MAXTC2 / 4; ++t4) {
            const float4 e4 = *reinterpret_cast<const float4*>(esc + 4 * t4);
            l += (e4.x == -INFINITY ? 0.f : fast_exp_(e4.x - m)) + (e4.y == -INFINITY ? 0.f : fast_exp_(e4.y - m)) +
                 (e4.z == -INFINITY ? 0.f : fast_exp_(e4.z - m)) + (e4.w == -INFINITY ? 0.f : fast_exp_(e4.w - m));
          }
        }
        float* ptw_ = pt + wave * DS_MAXTC2;
        {
          const float e1 = esc[lane];
          ptw_[lane] = e1 == -INFINITY ? 0.f : fast_exp_(e1 - m);      // frames beyond the chunk carry -inf
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // partial context: thread -> features 2 tid, 2 tid + 1 (one 8-byte LDS read per frame + one broadcast read of the weight)
        float c0 = 0.f, c1 = 0.f;
        const int f0 = 2 * tid;
        if (f0 < D) {
#pragma unroll 8
          for (int t2 = 0; t2 < (LONG ? (nt < DS_MAXTC ? nt : DS_MAXTC) : nt); ++t2) {      // (a plain `nt` bound where it can be: the min() costs the common instance 32 bytes of spills)
            const float2 ev2 = *reinterpret_cast<const float2*>(enc_s + t2 * ENLD + f0);
            const float pw = ptw_[t2];
            c0 = fmaf(pw, ev2.x, c0);
            c1 = fmaf(pw, ev2.y, c1);
          }
#pragma unroll 8
          for (int t2 = DS_MAXTC; LONG && t2 < nt; ++t2) {          // (T' > 256 only) the streamed frames: one coalesced 8-byte load per thread and frame
            const float2 ev2 = *reinterpret_cast<const float2*>(a.enc + ((long)ab * T2 + t_lo + t2) * D + f0);
            const float pw = ptw_[t2];
            c0 = fmaf(pw, ev2.x, c0);
            c1 = fmaf(pw, ev2.y, c1);
          }
        }
        if (f0 < D) { pbuf[4 + f0] = c0; pbuf[4 + f0 + 1] = c1; }
        if (tid == 0) { pbuf[0] = m; pbuf[1] = l; pbuf[2] = 0.f; pbuf[3] = 0.f; }
      }
      ds_mark(cA, wave, i + 1);
      DS_STAMP(3);
      // ---- (3) the row's 8 partials: statistics + this workgroup's feature slice; combine.  A layer-0 cell's previous cell state
      //          (c1 of the previous step, its own block) rides in the same gather ----
      {
        const int npc = 1 + (FS >> 2);                            // pieces per chunk: stats + slice
        const int cc = tid / npc, kk = tid % npc;
        const bool mine = attn && tid < DS_NC * npc;
        const float* xs = xb;                                    // both pieces are addressed from the start of the exchange buffer
        const unsigned px = (unsigned)((slot_cur + a.o_part + ((long)(attn ? ab : 0) * DS_NC + (mine ? cc : 0)) * (4 + D) +
                                        (mine && kk > 0 ? 4 + FS * ac + 4 * (kk - 1) : 0)) * 4);
        const bool wantc = cell && layer == 0 && i > 0 && wave == 0 && lane < 16;
        const unsigned pc = wantc ? (unsigned)((slot_prev + a.o_c1 + (((long)tile * Q + q) * 16 + lane) * 4) * 4) : px;
        const unsigned o5[5] = {px, pc, px, px, px};
        const bool use[5] = {mine, wantc, false, false, false};
        f32x4 sv[5];
        if (!ds_gather5(xs, o5, use, sv, abort_flag, a.spin_limit, a.delay, 2 | (i << 8))) break;
        if (mine) *reinterpret_cast<f32x4*>(sl + cc * 68 + 4 * kk) = sv[0];
        if (cell && layer == 0 && wave == 0 && lane < 16) {
          f32x4 cvv = sv[1];
          if (i == 0) cvv = tile * 16 + lane < B ? *reinterpret_cast<const f32x4*>(a.c_init + (long)(tile * 16 + lane) * Hd + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(cblk + lane * 4) = cvv;
        }
      }
      DS_STAMP(4);
      if (attn) {
        ds_mark(cB, wave, i + 1);
        if (wave == 0) {
          if (!ds_wait4(cB, i + 1, abort_flag, lds_limit, 6 | (i << 8))) break;
          float m = -INFINITY;
#pragma unroll
          for (int c2 = 0; c2 < DS_NC; ++c2) m = fmaxf(m, sl[c2 * 68]);
          float L = 0.f, val = 0.f, alpha_own = 0.f;
#pragma unroll
          for (int c2 = 0; c2 < DS_NC; ++c2) {
            const float mc = sl[c2 * 68];
            const float al = (mc == -INFINITY) ? 0.f : fast_exp_(mc - m);
            L = fmaf(al, sl[c2 * 68 + 1], L);
            if (lane < FS) val = fmaf(al, sl[c2 * 68 + 4 + lane], val);
            if (c2 == ac) alpha_own = al;
          }
          const float inv = 1.f / L;
          if (lane < FS) cslice[lane] = val * inv;
          pn[lane] = pt[lane] * alpha_own * inv;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) *cC = i + 1;
        }
      }
      DS_STAMP(5);
      // ---- (4) the cell of this workgroup's layer ----
      if (cell) {
        f32x4 acc0 = acc_h, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (layer == 0) {
          // dropout multipliers of this lane's context operands first (they do not depend on the gather)
          const AsrRngKey key = asr_rng_key(seedv, a.stream0 + a.stream_step * (uint32_t)i + 2u);
          f32x4 cv[8];
          unsigned o8[8];
          bool use[8];
          const float* cx = xb + slot_cur + a.o_ctx;
          const bool rowok = tile * 16 + li < B;                 // rows beyond B have no attention workgroup: their context is zero, never awaited
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int jb = wave + 4 * k;
            use[k] = jb < KBC && rowok;
            o8[k] = use[k] ? (unsigned)((((long)tile * (D >> 2) + 4 * jb + lq) * 16 + li) * 16) : 0u;
          }
          // one keep-bit per operand (bit 4k + e): the 32 multipliers would otherwise occupy 32 registers across the gather
          unsigned keep = 0xFFFFFFFFu;
          if (a.rate > 0.f) {
            keep = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const uint32_t idx = (uint32_t)((long)(b0_ + tile * 16 + li) * (Hd + D) + Hd + 16 * (wave + 4 * k) + 4 * lq);
#pragma unroll
              for (int e = 0; e < 4; ++e) keep |= (asr_rng_u32(key, idx + e) >= thresh ? 1u : 0u) << (4 * k + e);
            }
          }
          if (!ds_gather8(cx, o8, use, cv, abort_flag, a.spin_limit, a.delay, 3 | (i << 8))) break;
          DS_STAMP(6);
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (!use[k]) cv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (wave + 4 * k < KBC) {
              const float m0 = (keep >> (4 * k)) & 1u ? scale : 0.f, m1 = (keep >> (4 * k + 1)) & 1u ? scale : 0.f;
              const float m2 = (keep >> (4 * k + 2)) & 1u ? scale : 0.f, m3 = (keep >> (4 * k + 3)) & 1u ? scale : 0.f;
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[k].x * m0, bwx[k].x, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[k].y * m1, bwx[k].y, acc1, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[k].z * m2, bwx[k].z, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[k].w * m3, bwx[k].w, acc1, 0, 0, 0);
            }
          }
        } else {
          // layer 1: x = drop(y0) with y0 = tokmask ? h0 : 0, state = h0
          const AsrRngKey key = asr_rng_key(seedv, a.stream0 + a.stream_step * (uint32_t)i + 3u);
          const float* hx = xb + slot_cur + a.o_h0;
          unsigned o5[5];
          bool use[5];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int jb = wave + 4 * k;
            use[k] = jb < KBH;
            o5[k] = use[k] ? (unsigned)((((long)tile * Q + 4 * jb + lq) * 16 + li) * 16) : 0u;
          }
          use[4] = wave == 0 && lane < 16;                       // c0 of the owned pairs
          o5[4] = use[4] ? (unsigned)(((a.o_c0 - a.o_h0) + (((long)tile * Q + q) * 16 + lane) * 4) * 4) : 0u;
          const int brow = tile * 16 + li;
          const bool rowm = brow < B ? a.tokmask[(long)i * Bs_ + brow] != 0 : false;
          unsigned keep = rowm ? 0xFFFFu : 0u;                   // one keep-bit per operand (bit 4k + e)
          if (rowm && a.rate > 0.f) {
            keep = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const uint32_t idx = (uint32_t)((long)(b0_ + brow) * Hd + 16 * (wave + 4 * k) + 4 * lq);
#pragma unroll
              for (int e = 0; e < 4; ++e) keep |= (asr_rng_u32(key, idx + e) >= thresh ? 1u : 0u) << (4 * k + e);
            }
          }
          if (!ds_gather5(hx, o5, use, hv, abort_flag, a.spin_limit, a.delay, 4 | (i << 8))) break;
          DS_STAMP(6);
          if (wave == 0 && lane < 16) *reinterpret_cast<f32x4*>(cblk + lane * 4) = hv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)                            // (static register indices: a run-time pick would go through scratch)
            if (k == (q >> 4) && wave == ((q >> 2) & 3) && lq == (q & 3)) *reinterpret_cast<f32x4*>(hblk + li * 4) = hv[k];
#pragma unroll
          for (int k = 0; k < DS_MAXHB; ++k) {
            if (wave + 4 * k < KBH) {
              const float m0 = (keep >> (4 * k)) & 1u ? scale : 0.f, m1 = (keep >> (4 * k + 1)) & 1u ? scale : 0.f;
              const float m2 = (keep >> (4 * k + 2)) & 1u ? scale : 0.f, m3 = (keep >> (4 * k + 3)) & 1u ? scale : 0.f;
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].x, bwh[k].x, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].x * m0, bwx[k].x, acc1, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].y, bwh[k].y, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].y * m1, bwx[k].y, acc1, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].z, bwh[k].z, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].z * m2, bwx[k].z, acc1, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].w, bwh[k].w, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[k].w * m3, bwx[k].w, acc1, 0, 0, 0);
            }
          }
        }
        const f32x4 acc = acc0 + acc1;
        float* ptw = part + (i & 1) * 4 * 272 + wave * 272;
#pragma unroll
        for (int r = 0; r < 4; ++r) ptw[(lq * 4 + r) * 17 + li] = acc[r];
      }
      ds_mark(cD, wave, i + 1);
      DS_STAMP(7);
    }
  } else {
    // ================================================================================================= PUBLISH wave
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, (int)a.xbytes, 0x00020000);
    const u32x4 sent = {DS_SENT, DS_SENT, DS_SENT, DS_SENT};
    const int bi = lane >> 2, u = lane & 3;
    const int brow = tile * 16 + bi, j = 4 * q + u;
    const bool live = cell && brow < B;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (cell && layer == 1)
#pragma unroll
      for (int g = 0; g < 4; ++g) bias[g] = a.bias1[(long)g * Hd + j];
    for (int i = 0; i < U; ++i) {
      const long slot_cur = (long)(i & 3) * a.slot_floats, slot_old = (long)((i + 2) & 3) * a.slot_floats;
      // gate-math operands that do not depend on the exchange
      bool m = true;
      float pre[4] = {bias[0], bias[1], bias[2], bias[3]};
      if (live) {
        m = a.tokmask[(long)i * Bs_ + brow] != 0;
        if (layer == 0) {
          const float* pr = a.pre0 + ((long)i * Bs_ + brow) * 4 * Hd + j;
#pragma unroll
          for (int g = 0; g < 4; ++g) pre[g] = pr[(long)g * Hd];
        }
      }
      // ---- the chunk's partial: {m, l}, ctx_c ----
      // ONE wait per step for this wave's stores: it retires every sentinel of the previous step before any publish of this one (a
      // slot re-armed at step i is rewritten at step i + 2 and polled again only by readers that have consumed this wave's step
      // i + 1 pieces).  Rounds 1-2 also waited in front of the slice and the cell publish - i.e. for the write-through
      // acknowledgement of the stores issued a moment earlier in the same step, a fabric round trip each on the step's critical chain
      // (placed while the gather waves are still busy with the attention stage: the wait is off the chain)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!ds_wait4(cA, i + 1, abort_flag, lds_limit, 7 | (i << 8))) break;
      if (attn) {
        const int npieces = 1 + (D >> 2);
        const long base = a.o_part + ((long)ab * DS_NC + ac) * (4 + D);
        for (int pc = lane; pc < npieces; pc += 64) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(pbuf + 4 * pc);
          __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)((slot_cur + base + 4 * pc) * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)((slot_old + base + 4 * pc) * 4), 0, 16);
        }
      }
      DS_STAMP(8);
      // ---- the combined context slice ----
      if (attn) {
        if (!ds_wait(cC, i + 1, abort_flag, lds_limit, 8 | (i << 8))) break;
        asm volatile("" ::: "memory");                   // (compiler barrier only: keeps the slice's LDS reads behind the wait)
        if (lane < (FS >> 2)) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(cslice + 4 * lane);
          const long off = a.o_ctx + (((long)atile * (D >> 2) + (FS >> 2) * ac + lane) * 16 + arow) * 4;
          __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)((slot_cur + off) * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)((slot_old + off) * 4), 0, 16);
        }
        if (lane < FS) a.ctx[((long)i * Bs_ + ab) * D + FS * ac + lane] = cslice[lane];
        if (lane < nt) a.p[((long)i * Bs_ + ab) * T2 + t_lo + lane] = pn[lane];
      }
      DS_STAMP(9);
      // ---- the cell ----
      if (!ds_wait4(cD, i + 1, abort_flag, lds_limit, 9 | (i << 8))) break;
      if (cell) {
        const float* pp = part + (i & 1) * 4 * 272;
        float s[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          s[g] = pp[bi * 17 + g * 4 + u] + pp[272 + bi * 17 + g * 4 + u] + pp[2 * 272 + bi * 17 + g * 4 + u] + pp[3 * 272 + bi * 17 + g * 4 + u];
        const float hp = hblk[bi * 4 + u], cp = cblk[bi * 4 + u];
        float hn, c2, sv[4];
        const float br[3] = {0.f, 0.f, 0.f};
        asr_cell_forward<0>(pre, s, br, hp, cp, hn, c2, sv);
        const float hnew = m ? hn : hp, cnew = m ? c2 : cp, y = m ? hn : 0.f;
        f32x4 ph, pc4;
        ph.x = hnew; ph.y = __shfl_down(hnew, 1, 64); ph.z = __shfl_down(hnew, 2, 64); ph.w = __shfl_down(hnew, 3, 64);
        pc4.x = cnew; pc4.y = __shfl_down(cnew, 1, 64); pc4.z = __shfl_down(cnew, 2, 64); pc4.w = __shfl_down(cnew, 3, 64);
        asm volatile("" ::: "memory");
        if (u == 0) {
          const long blk = (((long)tile * Q + q) * 16 + bi) * 4;
          const long oh = layer == 0 ? a.o_h0 : a.o_h1, oc = layer == 0 ? a.o_c0 : a.o_c1;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ph), rsrc, (int)((slot_cur + oh + blk) * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pc4), rsrc, (int)((slot_cur + oc + blk) * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)((slot_old + oh + blk) * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)((slot_old + oc + blk) * 4), 0, 16);
        }
        DS_STAMP(10);
        if (live) {
          const long row = (long)i * Bs_ + brow;
          float* svp = (layer == 0 ? a.saved0 : a.saved1) + row * 4 * Hd + j;
#pragma unroll
          for (int g = 0; g < 4; ++g) svp[(long)g * Hd] = sv[g];
          (layer == 0 ? a.y0 : a.y1)[row * Hd + j] = y;
          if (layer == 0) {
            a.h0[row * Hd + j] = hnew;
            a.c0[row * Hd + j] = cnew;
          } else {
            a.hin[((long)(i + 1) * Bs_ + brow) * Hd + j] = hnew;
            a.cin[((long)(i + 1) * Bs_ + brow) * Hd + j] = cnew;
          }
        }
      }
      DS_STAMP(11);
    }
  }
  __syncthreads();
  if (*abort_flag && tid == 0) {
    __hip_atomic_store(a.err, (unsigned)*abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // err[1]: 0x7fffffff - the EARLIEST (step, stage) that gave up anywhere; err[2]: the workgroup that reported it last
    __hip_atomic_fetch_max(a.err + 1, 0x7fffffffu - (unsigned)(((*abort_flag >> 8) << 8) | (*abort_flag & 255)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.err[32 + w] = (unsigned)*abort_flag;               // per-workgroup record (diagnosis: tests/tools/dbg_decoder_sweep.py)
    swd_record(a.err, (unsigned)*abort_flag, 0);
    if (a.err_flag) __hip_atomic_store(reinterpret_cast<unsigned*>(a.err_flag), 0x3F800000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) swd_depart(a.err);
}

static size_t ds_lds_bytes(int Hd, int D) {
  const size_t floats = (size_t)DS_MAXTC * (Hd + 4) + (size_t)DS_MAXTC * (D + 4) + DS_MAXTC2 + 4 * 256 + DS_MAXTC2 + 4 * DS_MAXTC2 + (4 + 512) + 8 * 68 + 64 +
                        DS_MAXTC2 + 2 * 4 * 272 + 64 + 64 + 32;
  return floats * sizeof(float);
}

static void ds_layout(int Hd, int D, long* o_h1, long* o_c1, long* o_h0, long* o_c0, long* o_part, long* o_ctx, long* slot) {
  const long hsz = 2L * Hd * 16;                         // [2 tiles][Hd / 4][16][4]
  *o_h1 = 0; *o_c1 = hsz; *o_h0 = 2 * hsz; *o_c0 = 3 * hsz;
  *o_part = 4 * hsz;
  *o_ctx = *o_part + 32L * DS_NC * (4 + D);
  *slot = *o_ctx + 2L * D * 16;
}

extern "C" int asr_decoder_sweep_supported(int rnn_type, int num_layers, int B, int U, int T2, int Hd, int D) {
  // B <= 64: batches of more than 32 rows run as two launches of <= 32 rows (asr_decoder_sweep_fwd does that itself); T' <= 512:
  // chunks of up to 64 frames, the first 32 of each resident in LDS, the rest streamed (libri_config.yml max_audio_length 2048 frames: T' = 511)
  if (rnn_type != 0 || num_layers != 2 || B <= 0 || B > 64 || U < 1 || T2 < 1 || T2 > DS_NC * DS_MAXTC2) return 0;
  if (Hd <= 0 || Hd % 16 != 0 || Hd > 256 || D <= 0 || D % 32 != 0 || D > 512) return 0;
  // 256 workgroups that wait for each other, ~100 KB of LDS each: one per compute unit, all resident at once.  The occupancy answer
  // is a query, not a reservation (single tenant assumed: another process holding LDS on one CU stalls the grid until it leaves or
  // the spin limit reports it - sweep_common.h says which); a device that cannot hold the grid even when empty is refused here
  static long cap = 0;
  if (cap == 0) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_sweep_fwd_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int dev = 0, cus = 0, per = 0;
    cap = -1;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, reinterpret_cast<const void*>(decoder_sweep_fwd_kernel<false, false>), 320, ds_lds_bytes(256, 512)) == hipSuccess)
      cap = (long)per * cus;
    (void)hipGetLastError();
  }
  if (cap >= 0 && cap < 256) return 0;
  return 1;
}

extern "C" long asr_decoder_sweep_ws_floats(int Hd, int D) {
  long a, b, c, d, e, f, slot;
  ds_layout(Hd, D, &a, &b, &c, &d, &e, &f, &slot);
  return DS_SLOTS * slot + 32 + 256;                     // exchange, error words, per-workgroup abort record
}

extern "C" int asr_decoder_sweep_fwd(const asr_decoder_sweep* s, float* ws, float* err_flag, void* stream) {
  ASR_CHECK(s && ws, ASR_ERR_ARG, "asr_decoder_sweep_fwd: null argument");
  ASR_CHECK(asr_decoder_sweep_supported(0, 2, s->B, s->U, s->T2, s->Hd, s->D), ASR_ERR_UNSUPPORTED, "asr_decoder_sweep_fwd: shape not supported");
  ASR_CHECK(s->Kq && s->enc && s->mask && s->h_init && s->c_init && s->Wp0 && s->Wp1 && s->pre0 && s->bias1 && s->tokmask && s->p && s->ctx &&
                s->hin && s->cin && s->y0 && s->saved0 && s->h0 && s->c0 && s->y1 && s->saved1,
            ASR_ERR_ARG, "asr_decoder_sweep_fwd: null buffer");
  ASR_CHECK(!(s->drop_rate > 0.f && !s->seed), ASR_ERR_ARG, "asr_decoder_sweep_fwd: dropout needs a device seed");
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
    ASR_CHECK(cus >= 256, ASR_ERR_UNSUPPORTED, "asr_decoder_sweep_fwd: needs 256 compute units (one resident workgroup each), device has %d", cus);
  hipStream_t st = (hipStream_t)stream;
  DsArgs a{};
  a.B = s->B; a.U = s->U; a.T2 = s->T2; a.Hd = s->Hd; a.D = s->D;
  a.TC = asr_cdiv(s->T2, DS_NC); a.FS = s->D / DS_NC;
  a.Kq = s->Kq; a.enc = s->enc; a.s0 = s->s0; a.mask = s->mask; a.h_init = s->h_init; a.c_init = s->c_init;
  a.Wp0 = s->Wp0; a.KSt0 = s->KSt0; a.kc0 = s->ks0_ctx; a.kh0 = s->ks0_h;
  a.Wp1 = s->Wp1; a.KSt1 = s->KSt1; a.kx1 = s->ks1_x; a.kh1 = s->ks1_h;
  a.pre0 = s->pre0; a.bias1 = s->bias1; a.tokmask = s->tokmask;
  a.seed = s->seed; a.rate = s->drop_rate; a.stream0 = s->drop_stream0; a.stream_step = s->drop_stream_step;
  a.p = s->p; a.ctx = s->ctx; a.hin = s->hin; a.cin = s->cin; a.y0 = s->y0; a.saved0 = s->saved0; a.h0 = s->h0; a.c0 = s->c0;
  a.y1 = s->y1; a.saved1 = s->saved1;
  ds_layout(s->Hd, s->D, &a.o_h1, &a.o_c1, &a.o_h0, &a.o_c0, &a.o_part, &a.o_ctx, &a.slot_floats);
  const long xfloats = DS_SLOTS * a.slot_floats;
  a.xbuf = ws; a.xbytes = xfloats * 4;
  a.err = reinterpret_cast<unsigned*>(ws + xfloats);
  a.err_flag = err_flag;
  a.spin_limit = asr_rnn_sweep_spin_limit();
  a.prio = asr_sweep_prio();
  a.delay = getenv("ASR_DECODER_SWEEP_DELAY") ? atoi(getenv("ASR_DECODER_SWEEP_DELAY")) : 8;   // negative: timing experiment, gathers do not wait
  const size_t smem = ds_lds_bytes(s->Hd, s->D);
  static unsigned long long attr = 0;
  if (asr_first_use_on_device(attr)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_sweep_fwd_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_sweep_fwd_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  // one launch per 32 batch rows (the grid is 32 rows x 8 chunks of attention and 2 batch tiles of cells): the passes are independent
  // chains over the same U steps, each as long as one launch - twice the time for B = 64, against 9 launches per step on the fallback
  a.Bs = s->B;
  for (int b0 = 0; b0 < s->B; b0 += 32) {
    const long T2 = s->T2, Hd = s->Hd, D = s->D;
    a.B = s->B - b0 < 32 ? s->B - b0 : 32; a.b0 = b0;
    a.Kq = s->Kq + b0 * T2 * Hd; a.enc = s->enc + b0 * T2 * D; a.s0 = s->s0 ? s->s0 + b0 * T2 : nullptr; a.mask = s->mask + b0 * T2;
    a.h_init = s->h_init + b0 * Hd; a.c_init = s->c_init + b0 * Hd;
    a.pre0 = s->pre0 + b0 * 4 * Hd; a.tokmask = s->tokmask + b0;
    a.p = s->p + b0 * T2; a.ctx = s->ctx + b0 * D; a.hin = s->hin + b0 * Hd; a.cin = s->cin + b0 * Hd;
    a.y0 = s->y0 + b0 * Hd; a.saved0 = s->saved0 + b0 * 4 * Hd; a.h0 = s->h0 + b0 * Hd; a.c0 = s->c0 + b0 * Hd;
    a.y1 = s->y1 + b0 * Hd; a.saved1 = s->saved1 + b0 * 4 * Hd;
    const size_t n = (size_t)xfloats;
    hipLaunchKernelGGL(sw_fill_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, st,
                       reinterpret_cast<uint32_t*>(ws), n, DS_SENT, a.err, 16, 256u);
    (void)asr_zero_async(a.err + 32, 256 * sizeof(unsigned), st);
    ASR_LAUNCH_CHECK();
    static const int trace = getenv("ASR_DECODER_SWEEP_TRACE") ? atoi(getenv("ASR_DECODER_SWEEP_TRACE")) : 0;
    if (a.TC > DS_MAXTC || a.Bs != a.B || a.b0 != 0) hipLaunchKernelGGL((decoder_sweep_fwd_kernel<true, false>), dim3(256), dim3(320), smem, st, a);
    else if (trace) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_sweep_fwd_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((decoder_sweep_fwd_kernel<false, true>), dim3(256), dim3(320), smem, st, a);
    } else hipLaunchKernelGGL((decoder_sweep_fwd_kernel<false, false>), dim3(256), dim3(320), smem, st, a);
    ASR_LAUNCH_CHECK();
  }
  return ASR_OK;
}

// Timing aid: the stage stamps of the last traced forward decoder sweep (ASR_DECODER_SWEEP_TRACE=1): n <= 2 * 128 * 16 words.
extern "C" int asr_debug_decoder_trace(unsigned long long* out, int n) {
  if (!out || n <= 0 || n > 2 * DSF_TRACE_STEPS * 16) return ASR_ERR_ARG;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dsf_trace), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? ASR_OK : ASR_ERR_HIP;
}
