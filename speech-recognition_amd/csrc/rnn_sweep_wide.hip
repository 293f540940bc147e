// One-launch forward sweep of a WIDE bidirectional LSTM layer (H up to 1024, B up to 64) under mixed precision on gfx950: the
// las_large encoder (las_large.yml: encoder_hidden_dim 1024), where the f32 sweep of rnn_sweep.hip cannot keep the recurrent
// kernel on chip (4 M f32 weights per direction) and the layer ran as T' launches of rnn_step_fwd_wide_kernel (13.5 us each).
//
// Residency: bf16 weights.  The recurrent kernel of one direction is 8 MB in bf16; 128 workgroups per direction (one per CU,
// 256 in all for a bidirectional layer) each own 8 hidden units x 4 gates = 32 columns x H rows = 64 KB of it, held in the
// registers of 4 gather waves (64 VGPRs each) for all T' steps.  A workgroup therefore multiplies ALL batch rows (up to 4 tiles
// of 16) with its columns: per step [64 x H] x [H x 32] on v_mfma_f32_16x16x32_bf16 (64 MFMAs per wave, K split over the waves),
// f32 accumulation, f32 gate math and f32 cell state - the same arithmetic as the mixed-precision step kernels (state operand
// rounded to bf16 on its way into the product).
//
// Exchange: h_t travels as bf16.  A 16-byte piece = one batch row x the 8 units of one workgroup = exactly one A operand of
// the MFMA (lane = (row, k group of 8)), so a gather lane loads its operands ready-made: 32 pieces per lane and step, 128 KB
// per workgroup (H = 1024, B = 64) - the all-gather of a weights-resident split, served by L2 / Infinity Cache.  Layout
// [slot][unit group][64 rows][16 B]: a workgroup publishes one contiguous 1 KB block.  Hand-off as in rnn_sweep.hip: the data
// is the flag (a bf16 pair 0x7FC0DEAD - a NaN - marks "not written yet"), 4 slots, the publisher re-arms what it wrote two
// steps earlier, one probe piece is polled until fresh before the 32 loads go out, every spin bounded.
// Wave roles (384 threads): waves 0-1 gate math + publish + the layer's outputs (32 rows x 8 units each), waves 2-5 gather +
// product (a quarter of K each).  LSTM only; H % 128 == 0, 256 < H <= 1024; B <= 64.
#include <stdlib.h>

#include "common.h"

#include "decoder_sweep_common.h"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct WwDir {
  const float* pre; const float* Wp;
  const float* h0; long h0_ld; const float* c0; long c0_ld;
  float* hseq; float* cseq; float* saved;
  int reverse, y_col;
};
struct WwArgs {
  WwDir d[2];
  int B, T, H, MT;               // MT = batch tiles of 16 rows
  const uint8_t* mask;
  float* y; long y_ld;
  uint32_t* xbuf; long xbytes;   // [ndir][4 slots][H / 8][64 rows][4 words]
  long slot_words;
  unsigned* err; float* err_flag;
  int spin_limit, dbg, prio;
};

__device__ __forceinline__ uint32_t ww_pack2(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bool ww_fresh(const u32x4& v) { return v.x != DS_SENT && v.y != DS_SENT && v.z != DS_SENT && v.w != DS_SENT; }

// slot e of a piece holds unit perm(e) of the group: words are (unit j, unit j + 4) pairs - what one gate lane owns
__device__ __forceinline__ int ww_perm(int e) { return (e >> 1) + 4 * (e & 1); }

template <int KS>                // MFMA k-steps per gather wave: H = 128 KS
__global__ __launch_bounds__(384) void rnn_sweepw_fwd_kernel(WwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ww_lds[];
  float(*part)[4][4][2][16 * 17] = reinterpret_cast<float(*)[4][4][2][16 * 17]>(ww_lds);   // [parity][gather wave][batch tile][unit quad][row x column]
  // the workgroup's 64 KB of bf16 weights as ready-made B operands: [gather wave][k-step][unit quad][lane] x 16 bytes (each lane reads
  // back exactly what it wrote; 128 VGPRs of a gather lane hold the step's A operands, so the weights live one level further out)
  u32x4* wlds = reinterpret_cast<u32x4*>(ww_lds + 2 * 4 * 4 * 2 * 16 * 17);
  __shared__ int abort_flag;
  const WwDir& d = a.d[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool gate_wave = wv < 2;
  const int li = lane & 15, lq = lane >> 4;
  const int kp = blockIdx.x;                            // unit group: units 8 kp .. 8 kp + 7
  const int B = a.B, T = a.T, H = a.H, MT = a.MT;
  const int KB = H / 16;                                // K blocks of the packed f32 image
  uint32_t* xb = a.xbuf + (long)blockIdx.z * DS_SLOTS * a.slot_words;
  if (tid == 0) { abort_flag = 0; swd_arrive(a.err); }   // start handshake (sweep_common.h)
  swd_setprio(a.prio);

  // ---- gather waves: resident B operands, built once from the packed f32 image (asr_rnn_pack) ----
  const int w = gate_wave ? 0 : wv - 2;
  if (!gate_wave) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float* wq = d.Wp + (long)(2 * kp + nt) * KB * 256;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kpi = (w * KS + ks) * 4 + lq;         // the piece (unit group) this lane's k group comes from
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 8 * kpi + ww_perm(e);
          f[e] = (__bf16)wq[((long)(k >> 4) * 64 + ((k >> 2) & 3) * 16 + li) * 4 + (k & 3)];
        }
        wlds[((w * KS + ks) * 2 + nt) * 64 + lane] = __builtin_bit_cast(u32x4, f);
      }
    }
  }
  // ---- gate waves: lane -> (row bi of a tile, unit pair up / up + 4); wave g owns batch tiles 2g, 2g + 1 ----
  const int bi = lane >> 2, up = lane & 3;
  float hp[2][2], cp[2][2], yp[2][2];
  bool live[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 16 * (2 * wv + m) + bi, j = 8 * kp + up + 4 * h;
      live[m][h] = gate_wave && row < B;
      hp[m][h] = 0.f; cp[m][h] = 0.f; yp[m][h] = 0.f;
      if (live[m][h]) {
        hp[m][h] = d.h0 ? d.h0[(long)row * d.h0_ld + j] : 0.f;
        cp[m][h] = d.c0 ? d.c0[(long)row * d.c0_ld + j] : 0.f;
      }
    }
  if (tid == 0 && !swd_wait_all(a.err, a.spin_limit)) abort_flag = 15;   // the whole grid is resident before the first step
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, (int)a.xbytes, 0x00020000);
  const long dir_words = (long)blockIdx.z * DS_SLOTS * a.slot_words;

  int nretry = 0;                                          // whole gathers that found a stale piece (diagnosis word 25)
  for (int s = 0; s < T; ++s) {
    const int t = d.reverse ? T - 1 - s : s;
    // gate waves: operands that do not depend on the exchange
    bool mk[2];
    float pre[2][2][4];
    if (gate_wave) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int row = 16 * (2 * wv + m) + bi;
        mk[m] = true;
        if (row < B) mk[m] = a.mask ? a.mask[(long)row * T + t] != 0 : true;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int g = 0; g < 4; ++g) pre[m][h][g] = 0.f;
          if (live[m][h]) {
            const float* pr = d.pre + ((long)row * T + t) * 4 * H + 8 * kp + up + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[m][h][g] = pr[(long)g * H];
          }
        }
      }
    } else {
      // ------------------------------------------------------------------------------------------ gather + product
      u32x4 av[4][KS];
      if (s == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            av[mt][ks] = (u32x4){0u, 0u, 0u, 0u};
            const int row = 16 * mt + li, kpi = (w * KS + ks) * 4 + lq;
            if (d.h0 && row < B) {
              const float* hr = d.h0 + (long)row * d.h0_ld + 8 * kpi;
              av[mt][ks] = (u32x4){ww_pack2(hr[0], hr[4]), ww_pack2(hr[1], hr[5]), ww_pack2(hr[2], hr[6]), ww_pack2(hr[3], hr[7])};
            }
          }
      } else {
        const uint32_t* src = xb + (long)((s - 1) & 3) * a.slot_words;
        // one probe piece until it is fresh, then everything
        {
          const uint32_t* pp = src + ((long)((w * KS) * 4 + lq) * 64 + li) * 4;
          bool ok2 = true;
          for (int spins = 0;; ++spins) {
            u32x4 v;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(pp) : "memory");
            if (__all(ww_fresh(v)) || (a.dbg & 2)) break;
            if (lds_peek(&abort_flag)) { ok2 = false; break; }
            if (spins > a.spin_limit) { abort_flag = 1 | (s << 8); ok2 = false; break; }
            __builtin_amdgcn_s_sleep(4);
          }
          if (!ok2) goto wave_sync;                      // (never leave the step loop in front of the workgroup barrier)
        }
        bool done = false;
        // addresses: wave-uniform slot base in SGPRs + one 32-bit byte offset per k-step; the batch tile is the instruction's
        // immediate offset (256 bytes per tile) - 8 address registers for 32 loads
        unsigned vo[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vo[ks] = (unsigned)((((long)((w * KS + ks) * 4 + lq) * 64 + li) * 4) * 4);
        // a k-step's pieces come from 4 of the 128 senders; the probe watched 16 of them, and one time in five a gather still finds
        // somebody's piece stale (diagnosis word 25): only the k-steps that held one are read again, not the wave's whole 32 KB
        bool need[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) need[ks] = true;
        for (int spins = 0; !done; ++spins) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            if (!need[ks]) continue;                        // (wave-uniform)
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc1" : "+v"(av[0][ks]) : "v"(vo[ks]), "s"(src) : "memory");
            if (MT > 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:256 sc1" : "+v"(av[1][ks]) : "v"(vo[ks]), "s"(src) : "memory");
            if (MT > 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:512 sc1" : "+v"(av[2][ks]) : "v"(vo[ks]), "s"(src) : "memory");
            if (MT > 3) asm volatile("global_load_dwordx4 %0, %1, %2 offset:768 sc1" : "+v"(av[3][ks]) : "v"(vo[ks]), "s"(src) : "memory");
          }
          // the loads above are in flight: tie every destination to the wait so that nothing reads them earlier
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            if (mt < MT) {
              if constexpr (KS == 8) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(av[mt][0]), "+v"(av[mt][1]), "+v"(av[mt][2]), "+v"(av[mt][3]), "+v"(av[mt][4]), "+v"(av[mt][5]),
                             "+v"(av[mt][6]), "+v"(av[mt][7])::"memory");
              } else if constexpr (KS == 6) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(av[mt][0]), "+v"(av[mt][1]), "+v"(av[mt][2]), "+v"(av[mt][3]), "+v"(av[mt][4]), "+v"(av[mt][5])::"memory");
              } else {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(av[mt][0]), "+v"(av[mt][1]), "+v"(av[mt][2]), "+v"(av[mt][3])::"memory");
              }
            }
          }
          bool ok = true;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            if (!need[ks]) continue;
            bool okk = true;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
              if (mt < MT) okk = okk && ww_fresh(av[mt][ks]);
            const bool fresh_k = __all(okk);
            need[ks] = !fresh_k;
            ok = ok && fresh_k;
          }
          if (ok || (a.dbg & 2)) { done = true; break; }
          ++nretry;
          if (lds_peek(&abort_flag)) break;
          if (spins > a.spin_limit) { abort_flag = 2 | (s << 8); break; }
          __builtin_amdgcn_s_sleep(2);
        }
        if (!done) goto wave_sync;
      }
      float(*ptw)[4][2][16 * 17] = part[s & 1];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if (mt < MT) {
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a8 = __builtin_bit_cast(bf16x8, av[mt][ks]);
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, wlds[((w * KS + ks) * 2 + 0) * 64 + lane]);
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, wlds[((w * KS + ks) * 2 + 1) * 64 + lane]);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b1, acc1, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ptw[w][mt][0][(lq * 4 + r) * 17 + li] = acc0[r];
            ptw[w][mt][1][(lq * 4 + r) * 17 + li] = acc1[r];
          }
        }
      }
    }
  wave_sync:
    float(*pt)[4][2][16 * 17] = part[s & 1];
    // LDS-only barrier: __syncthreads() would also wait for the gate waves' output stores of the previous step to be acknowledged
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (abort_flag) break;

    if (gate_wave) {
      uint32_t word[2];
      float o_h[2][2], o_c[2][2], o_sv[2][2][4];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int mt = 2 * wv + m;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          o_h[m][h] = hp[m][h];
          o_c[m][h] = cp[m][h];
          o_sv[m][h][0] = o_sv[m][h][1] = o_sv[m][h][2] = o_sv[m][h][3] = 0.f;
          if (live[m][h]) {
            float sums[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
              sums[g] = pt[0][mt][h][bi * 17 + g * 4 + up] + pt[1][mt][h][bi * 17 + g * 4 + up] + pt[2][mt][h][bi * 17 + g * 4 + up] +
                        pt[3][mt][h][bi * 17 + g * 4 + up];
            float hn, c2;
            const float br[3] = {0.f, 0.f, 0.f};
            asr_cell_forward<0>(pre[m][h], sums, br, hp[m][h], cp[m][h], hn, c2, o_sv[m][h]);
            o_c[m][h] = mk[m] ? c2 : cp[m][h];
            o_h[m][h] = mk[m] ? hn : hp[m][h];
            yp[m][h] = mk[m] ? hn : yp[m][h];
          }
        }
        word[m] = ww_pack2(o_h[m][0], o_h[m][1]);
      }
      // publish first (it is on every other workgroup's critical path): lane up == 0 of a row collects the four words of the row's
      // piece; then re-arm the slot of two steps ago; the wait retires the previous step's stores (a step old: no stall)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        u32x4 pc;
        pc.x = word[m];
        pc.y = (uint32_t)__shfl_down((int)word[m], 1, 64);
        pc.z = (uint32_t)__shfl_down((int)word[m], 2, 64);
        pc.w = (uint32_t)__shfl_down((int)word[m], 3, 64);
        if (up == 0 && !(a.dbg & 4)) {
          const long off = ((long)kp * 64 + 16 * (2 * wv + m) + bi) * 4;
          const u32x4 sent = {DS_SENT, DS_SENT, DS_SENT, DS_SENT};
          __builtin_amdgcn_raw_buffer_store_b128(pc, rsrc, (int)((dir_words + (long)(s & 3) * a.slot_words + off) * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)((dir_words + (long)((s + 2) & 3) * a.slot_words + off) * 4), 0, 16);
        }
      }
      // the layer's own outputs
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (live[m][h]) {
            const int row = 16 * (2 * wv + m) + bi, j = 8 * kp + up + 4 * h;
            const long o = ((long)row * T + t) * H + j;
            d.hseq[o] = o_h[m][h];
            a.y[((long)row * T + t) * a.y_ld + d.y_col + j] = yp[m][h];
            d.cseq[o] = o_c[m][h];
            if (d.saved) {
              float* sv2 = d.saved + ((long)row * T + t) * 4 * H + j;
#pragma unroll
              for (int g = 0; g < 4; ++g) sv2[(long)g * H] = o_sv[m][h][g];
            }
            cp[m][h] = o_c[m][h];
            hp[m][h] = o_h[m][h];
          }
    }
  }
  __syncthreads();
  if (!gate_wave && lane == 0) { atomicAdd(a.err + 25, (unsigned)nretry); atomicAdd(a.err + 26, 1u); }
  if (abort_flag && tid == 0) {
    __hip_atomic_store(a.err, (unsigned)abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    swd_record(a.err, (unsigned)abort_flag, 0);
    if (a.err_flag) __hip_atomic_store(reinterpret_cast<unsigned*>(a.err_flag), 0x3F800000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) swd_depart(a.err);
}

extern "C" int asr_rnn_sweep_wide_supported(int rnn_type, int B, int T, int H, int ndir) {
  if (rnn_type != 0 || B <= 0 || B > 64 || T < 2 || H <= 256 || H > 1024 || H % 128 != 0) return 0;
  if (H / 128 != 4 && H / 128 != 6 && H / 128 != 8) return 0;
  if (ndir != 1 && ndir != 2) return 0;
  if ((long)(H / 8) * ndir > 256) return 0;              // one workgroup per compute unit
  return 1;
}

extern "C" long asr_rnn_sweep_wide_ws_floats(int B, int H, int ndir) {
  (void)B;
  return (long)ndir * DS_SLOTS * (H / 8) * 64 * 4 + 32;
}

// Same contract as asr_rnn_seq_fwd (rnn.hip) for a wide LSTM layer under mixed precision, in one launch; the state operand of
// the recurrent product is rounded to bf16 (as in rnn_step_fwd_wide_kernel with bf16 weights).  Error word / err_flag as for
// asr_rnn_sweep_fwd.
extern "C" int asr_rnn_sweep_wide_fwd(const asr_rnn_seq* s, float* ws, float* err_flag, void* stream) {
  ASR_CHECK(s && ws, ASR_ERR_ARG, "asr_rnn_sweep_wide_fwd: null argument");
  ASR_CHECK(asr_rnn_sweep_wide_supported(s->rnn_type, s->B, s->T, s->H, s->ndir), ASR_ERR_UNSUPPORTED, "asr_rnn_sweep_wide_fwd: shape not supported");
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
    ASR_CHECK(cus >= (s->H / 8) * s->ndir, ASR_ERR_UNSUPPORTED, "asr_rnn_sweep_wide_fwd: needs %d compute units, device has %d", (s->H / 8) * s->ndir, cus);
  const int B = s->B, T = s->T, H = s->H;
  hipStream_t st = (hipStream_t)stream;
  WwArgs a{};
  a.B = B; a.T = T; a.H = H; a.MT = asr_cdiv(B, 16);
  a.mask = s->mask; a.y = s->y; a.y_ld = s->y_ld;
  a.slot_words = (long)(H / 8) * 64 * 4;
  const long xwords = (long)s->ndir * DS_SLOTS * a.slot_words;
  a.xbuf = reinterpret_cast<uint32_t*>(ws); a.xbytes = xwords * 4;
  a.err = reinterpret_cast<unsigned*>(ws + xwords);
  a.err_flag = err_flag;
  a.spin_limit = asr_rnn_sweep_spin_limit();
  a.dbg = getenv("ASR_SWEEP_DBG") ? atoi(getenv("ASR_SWEEP_DBG")) : 0;
  a.prio = asr_sweep_prio();
  for (int d = 0; d < s->ndir; ++d) {
    ASR_CHECK(s->pre[d] && s->Wp[d] && s->hseq[d] && s->cseq[d] && s->y, ASR_ERR_ARG, "asr_rnn_sweep_wide_fwd: null buffer (dir %d)", d);
    ASR_CHECK(!s->rec_mult[d], ASR_ERR_UNSUPPORTED, "asr_rnn_sweep_wide_fwd: recurrent dropout is not supported (use asr_rnn_seq_fwd)");
    WwDir& p = a.d[d];
    p.pre = s->pre[d]; p.Wp = s->Wp[d];
    p.h0 = s->h0[d]; p.h0_ld = s->h0_ld[d]; p.c0 = s->c0[d]; p.c0_ld = s->c0_ld[d];
    p.hseq = s->hseq[d]; p.cseq = s->cseq[d]; p.saved = s->saved[d];
    p.reverse = s->reverse[d]; p.y_col = s->y_col[d];
  }
  {
    const size_t n = (size_t)xwords;
    const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(sw_fill_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<uint32_t*>(ws), n, DS_SENT, a.err, 16,
                       (unsigned)((H / 8) * s->ndir));
    ASR_LAUNCH_CHECK();
  }
  dim3 grid((unsigned)(H / 8), 1, (unsigned)s->ndir);
  const int ks = H / 128;
  const size_t smem = sizeof(float) * 2 * 4 * 4 * 2 * 16 * 17 + (size_t)4 * ks * 2 * 64 * 16;
  static unsigned long long attr = 0;
  if (asr_first_use_on_device(attr)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_sweepw_fwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_sweepw_fwd_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_sweepw_fwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  }
  if (ks == 8) hipLaunchKernelGGL(rnn_sweepw_fwd_kernel<8>, grid, dim3(384), smem, st, a);
  else if (ks == 6) hipLaunchKernelGGL(rnn_sweepw_fwd_kernel<6>, grid, dim3(384), smem, st, a);
  else hipLaunchKernelGGL(rnn_sweepw_fwd_kernel<4>, grid, dim3(384), smem, st, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
