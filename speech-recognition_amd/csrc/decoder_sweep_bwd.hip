// One-launch BACKWARD sweep of the LAS decoder under teacher forcing: the mirror of decoder_sweep.hip (las.py:267-292 looped by
// las.py:368-377, differentiated): all U steps of {decoder LSTM 1, decoder LSTM 0, attention} backwards in ONE kernel on gfx950.
//
// Per step i (from U-1 down to 0) the chain of dependent products is
//     dH(i), dC(i)  --[L1]-->  ds1 = gate gradients of layer 1;  dh0 = ds1 U1^T + drop (.) (ds1 W1^T)   (state + output of layer 0)
//                   --[L0]-->  ds0;  dH(i-1) part = ds0 U0^T;  dctx = drop (.) (ds0 Wc^T)
//                   --[A ]-->  dp = dctx enc^T, de = p (.) (dp - dctx . ctx)   [sum_t p dp = dctx . ctx: no exchange for the softmax],
//                              dq = de Kq  (the other part of dH(i-1))
// Pad-token rows carry dH and dC through unchanged.  256 workgroups of 512 threads stay resident; each has two roles:
//   * attention: workgroup w owns batch row b = w / 8 and time chunk c = w % 8; its slices of Kq and enc live in LDS for all steps;
//   * cell: layer w / 128, batch tile (w / 64) % 2, position (gi, gj) = ((w % 64) / 8, w % 8) in a G x G square (SUMMA, as in
//     rnn_sweep_bwd.hip, G = Hd / KU, KU = 16 NT hidden units per unit group): the workgroup repeats the element-wise gate
//     gradients of unit group gi, multiplies them with its resident [4 KU x KU] blocks of the transposed kernels and publishes
//     the partial input gradient of unit group gj (layer 0 also the partial context gradient of the D / G features gj).
// Three hand-offs per step, all of the forward sweep's kind (self-validating 16-byte pieces, NaN sentinel = "not written yet",
// 4 slots, the publisher re-arms what it wrote two steps earlier; every block has exactly one writer wave):
//   P1/C1: layer 1 -> layer 0 (partial dh0 blocks, dc)     P0/C0/PC: layer 0 -> layer 1 of the next step (partial dH, dc) and
//   -> attention (partial dctx, row-major)                   Q: attention -> layer 1 of the next step (dq per (row, chunk)).
// Wave roles: waves 0-3 gather, add up and do the element-wise math (they never store to the exchange); waves 4-7 multiply
// (contraction split over the waves, summed through LDS), publish, and write the saved results (ds, de, dctx) to memory.
// ds is written OUT OF PLACE: the G workgroups of a row all read the same saved activations one step ahead.
// Restrictions (the caller falls back to the per-step kernels): LSTM, 2 layers, B <= 32, Hd = 16..128 step 16 or 160..256 step
// 32, D / 4 a power of two <= 128, D % (16 G) == 0 with D / G <= 64, T' <= 256, 256 compute units.  Every spin is bounded.
#include <stdlib.h>

#include "common.h"

#include "decoder_sweep_common.h"

struct DbArgs {
  int B, U, T2, Hd, D, TC, G, FC, FS;     // FC = D / G context features per square column, FS = D / 8 features per chunk (saved dctx slice)
  int Bs, b0;                               // row stride of the step-major tensors / first batch row of this launch (batches > 32 run in passes)
  const float* Kq; const float* enc;
  const float* p; const float* ctx;
  const float* saved0; const float* saved1;
  float* ds0; float* ds1;
  const float* cin; const float* c0;
  const uint8_t* tokmask;
  const float* dy1; long dy1_ld;
  const float* U1; const float* W1; const float* U0; const float* W0c; long ldw;
  const uint32_t* seed; float rate; uint32_t stream0, stream_step;
  float* de; float* dctx; float* dhs; float* dc;
  float* de_sum;                            // optional [B,T2]: sum over the steps of de (the gradient wrt the constant score term s0 = K bq)
  float* xbuf; long xbytes;
  long o_p1, o_c1, o_p0, o_c0, o_pc, o_q, slot_floats;
  unsigned* err; float* err_flag;
  int spin_limit, delay, dbg, prio;
  int rowxcd;                               // block -> (row, chunk) mapping of the attention role
  int cellxcd;                              // block -> (gi, gj) mapping of the cell role
};

// up to 9 self-validating pieces per lane (layer 1: 4 partial-dH pieces, 4 dq pieces, 1 dc piece)
__device__ __forceinline__ bool db_gather9(const float* base, const unsigned (&off)[9], const bool (&use)[9], f32x4 (&v)[9], lds_flag_t abort_flag,
                                           int limit, int delay, int code) {
  if (delay >= 0 && delay < 1000) {
    // the probe piece is chosen with wave-uniform branches over STATIC indices (a run-time pick from the offset array goes
    // through scratch memory: a memory round trip in front of every gather)
    if (__any(use[0])) { if (!ds_probe(base, off[0], use[0], abort_flag, limit, code)) return false; }
    else if (__any(use[4])) { if (!ds_probe(base, off[4], use[4], abort_flag, limit, code)) return false; }
    else if (__any(use[8])) { if (!ds_probe(base, off[8], use[8], abort_flag, limit, code)) return false; }
  }
  for (int spins = 0;; ++spins) {
    asm volatile(
        "s_nop 4\n\t"
        "global_load_dwordx4 %0, %9, %18 sc1\n\t"
        "global_load_dwordx4 %1, %10, %18 sc1\n\t"
        "global_load_dwordx4 %2, %11, %18 sc1\n\t"
        "global_load_dwordx4 %3, %12, %18 sc1\n\t"
        "global_load_dwordx4 %4, %13, %18 sc1\n\t"
        "global_load_dwordx4 %5, %14, %18 sc1\n\t"
        "global_load_dwordx4 %6, %15, %18 sc1\n\t"
        "global_load_dwordx4 %7, %16, %18 sc1\n\t"
        "global_load_dwordx4 %8, %17, %18 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8])
        : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]), "v"(off[8]), "s"(base)
        : "memory");
    bool ok = true;
    int bad = 15;
#pragma unroll
    for (int i = 8; i >= 0; --i)
      if (use[i] && !ds_fresh(v[i])) { ok = false; bad = i; }
    if (__all(ok) || delay < 0) return true;
    if (*abort_flag) return false;
    if (spins > limit) {
      const unsigned long long ball = __ballot(!ok);
      const int fl = __ffsll((long long)ball) - 1;
      *abort_flag = code | (__shfl(bad, fl, 64) << 16) | (fl << 20);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// the five-piece gather of decoder_sweep_common.h with the static probe choice
__device__ __forceinline__ bool db_gather5(const float* base, const unsigned (&off)[5], const bool (&use)[5], f32x4 (&v)[5], lds_flag_t abort_flag,
                                           int limit, int delay, int code) {
  if (delay >= 0 && delay < 1000) {
    if (__any(use[0])) { if (!ds_probe(base, off[0], use[0], abort_flag, limit, code)) return false; }
    else if (__any(use[4])) { if (!ds_probe(base, off[4], use[4], abort_flag, limit, code)) return false; }
  }
  for (int spins = 0;; ++spins) {
    asm volatile(
        "s_nop 4\n\t"
        "global_load_dwordx4 %0, %5, %10 sc1\n\t"
        "global_load_dwordx4 %1, %6, %10 sc1\n\t"
        "global_load_dwordx4 %2, %7, %10 sc1\n\t"
        "global_load_dwordx4 %3, %8, %10 sc1\n\t"
        "global_load_dwordx4 %4, %9, %10 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
        : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "s"(base)
        : "memory");
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (use[i]) ok = ok && ds_fresh(v[i]);
    if (__all(ok) || delay < 0) return true;
    if (*abort_flag) return false;
    if (spins > limit) { *abort_flag = code; return false; }
    __builtin_amdgcn_s_sleep(1);
  }
}

__device__ __forceinline__ float db_pick(const f32x4& v, int idx) { return idx == 0 ? v.x : (idx == 1 ? v.y : (idx == 2 ? v.z : v.w)); }

// abort codes | (step << 8): 1 layer-1 gather, 2 layer-0 gather, 3 context-gradient gather, 5-9 LDS hand-overs of the gather waves,
// 10 publish waiting for the owner, 11 publish waiting for its contraction partners, 12 attention publish
template <int NT>
__global__ __launch_bounds__(512) void decoder_sweep_bwd_kernel(DbArgs a) {
  constexpr int KU = 16 * NT, NL = 4 * KU, KP = 4 / NT, KPW = NL / 4 / KP, LD = NL + 4;
  constexpr int NPOS = 64 * NT, TPP = 256 / NPOS;
  constexpr int CTM = NT == 2 ? 2 : 4;                 // context tiles per publish wave (at most)
  constexpr int NJ = 1 + CTM;                          // products per publish wave: recurrent tile + (layer 1: input tile | layer 0: context tiles)
  constexpr int QPR = KU / 4, CP = 16 / QPR, QM = 8 / CP;   // dq gather: quads per row, chunk parts, pieces per thread
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int w = blockIdx.x;
  const int B = a.B, U = a.U, T2 = a.T2, Hd = a.Hd, D = a.D, TC = a.TC, G = a.G, FC = a.FC, FS = a.FS;
  // attention role
  // consecutive blocks go to consecutive XCDs (round-robin dispatch): with a.rowxcd the 8 chunk workgroups of a batch row - which exchange
  // their partials with each other every step - share an XCD (block = chunk * 32 + row); 0 = block = row * 8 + chunk, one chunk per XCD
  const int ab = a.rowxcd ? (w & 31) : (w >> 3), ac = a.rowxcd ? (w >> 5) : (w & 7);
  const bool attn = ab < B;
  const int t_lo = ac * TC, nt = max(0, min(TC, T2 - t_lo));
  // cell role
  // (a.cellxcd: row gi of a cell square on XCD gi - its 8 workgroups gather the same blocks and fetch the same saved activations, as
  // in rnn_sweep_bwd.hip; 0: column gj on XCD gj)
  const int layer = w >> 7, tile = (w >> 6) & 1, gi = a.cellxcd ? (w & 7) : ((w & 63) >> 3), gj = a.cellxcd ? ((w & 63) >> 3) : (w & 7);
  const bool cell = gi < G && gj < G && tile * 16 < B;
  const bool writer = gj == 0;
  const bool diag = gi == gj;
  const int CT = FC / 16 / NT;                         // context tiles of a publish wave (layer 0)
  // LDS carve (floats)
  const int KQLD = Hd + 4, ENLD = D + 4;
  float* kq_s = lds;                                   // [32][KQLD]
  float* enc_s = kq_s + DS_MAXTC * KQLD;               // [32][ENLD]
  float* red = enc_s + DS_MAXTC * ENLD;                // [256][4] gather partial sums
  float* qs = red + 1024;                              // [CP][16][KU] dq partial sums
  float* dsl = qs + 1024;                              // [16][LD] labelled ds image of the step
  float* pass_s = dsl + 16 * LD;                       // [NPOS][4] state gradient carried past the gates (pad-token rows)
  float* dcs = pass_s + NPOS * 4;                      // [NPOS][4] dc handed to the next cell
  float* sp = dcs + NPOS * 4;                          // [(KP-1) NT NJ][256] contraction partials of the publish waves
  float* ctxT = sp + (KP - 1) * NT * NJ * 256;         // [16][68] partial context gradient, transposed to row-major
  float* dcx = ctxT + 16 * 68;                         // [256][4] context-gradient partial sums
  float* dcv_s = dcx + 1024;                           // [512] context gradient of the row
  float* dotw = dcv_s + 512;                           // [8]
  float* des = dotw + 8;                               // [64] score gradients of the chunk
  float* ps = des + DS_MAXTC2;                         // [64] attention weights of the chunk
  float* dqs = ps + DS_MAXTC2;                         // [256] partial query gradient
  int* flags = reinterpret_cast<int*>(dqs + 256);      // [64]
  const lds_flag_t abort_flag = lds_flag(flags);          // (LDS-typed: ds_read / ds_write, not flat accesses - sweep_common.h)
  const lds_flag_t cG = abort_flag + 4, cO = abort_flag + 8, cS = abort_flag + 12, cA1 = abort_flag + 16, cA2 = abort_flag + 20, cA3 = abort_flag + 24, cA4 = abort_flag + 28;
  if (tid < 64) flags[tid] = 0;
  if (tid == 0) swd_arrive(a.err);                       // start handshake (sweep_common.h)
  swd_setprio(a.prio);

  // ---- resident attention operands ----
  const int ntr = min(nt, DS_MAXTC);                    // frames resident in LDS; frames [ntr, nt) are read from memory every step (T' > 256)
  const float* kqg = a.Kq + ((long)(attn ? ab : 0) * T2 + t_lo) * Hd;
  const float* eng = a.enc + ((long)(attn ? ab : 0) * T2 + t_lo) * D;
  if (attn) {
    for (int i = tid; i < ntr * (Hd >> 2); i += 512) {
      const int t = i / (Hd >> 2), k4 = i % (Hd >> 2);
      *reinterpret_cast<float4*>(kq_s + t * KQLD + 4 * k4) = *reinterpret_cast<const float4*>(kqg + (long)t * Hd + 4 * k4);
    }
    for (int i = tid; i < ntr * (D >> 2); i += 512) {
      const int t = i / (D >> 2), k4 = i % (D >> 2);
      *reinterpret_cast<float4*>(enc_s + t * ENLD + 4 * k4) = *reinterpret_cast<const float4*>(eng + (long)t * D + 4 * k4);
    }
  }
  if (tid == 0 && !swd_wait_all(a.err, a.spin_limit)) *abort_flag = 15;   // the whole grid is resident before the first step
  __syncthreads();

  float* xb = a.xbuf;
  const int lds_limit = a.spin_limit > (1 << 20) ? a.spin_limit : (a.spin_limit << 4);
  const float scale = a.rate > 0.f ? 1.f / (1.f - a.rate) : 1.f;
  const uint32_t thresh = asr_drop_threshold(a.rate);
  const uint32_t seedv = (a.seed && a.rate > 0.f) ? a.seed[0] : 0u;
  const long blkf = (long)NT * 256;

  if (wv < 4) {
    // ================================================================================================= GATHER + OWNER waves
    // cell role: position (nt_, plq, pli) = rows 4 plq .. 4 plq + 3 of unit 16 nt_ + pli of group gi; TPP threads share a position
    // and finish NT of its rows each
    const int pos = tid % NPOS, sub = tid / NPOS;
    const int nt_ = pos >> 6, plq = (pos >> 4) & 3, pli = pos & 15;
    const int un = 16 * nt_ + pli, j = gi * KU + un;
    int brow[NT];
    bool live[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) {
      brow[r] = tile * 16 + 4 * plq + sub * NT + r;
      live[r] = cell && brow[r] < B;
    }
    struct Operands { bool m; float sv[4], cp, co, dy; };
    auto fetch = [&](int p, Operands (&o)[NT]) {
      const int i = U - 1 - p;
#pragma unroll
      for (int r = 0; r < NT; ++r) {
        o[r].m = true; o[r].cp = 0.f; o[r].co = 0.f; o[r].dy = 0.f;
        o[r].sv[0] = o[r].sv[1] = o[r].sv[2] = o[r].sv[3] = 0.f;
        if (live[r] && p < U) {
          const long row = (long)i * a.Bs + brow[r];
          o[r].m = a.tokmask[row] != 0;
          const float* sv = (layer == 0 ? a.saved0 : a.saved1) + row * 4 * Hd + j;
#pragma unroll
          for (int g = 0; g < 4; ++g) o[r].sv[g] = sv[(long)g * Hd];
          if (layer == 1) {
            o[r].dy = a.dy1[row * a.dy1_ld + j];
            o[r].co = a.cin[((long)(i + 1) * a.Bs + brow[r]) * Hd + j];  // c1(i)
            o[r].cp = a.c0[row * Hd + j];                               // layer 1 starts from layer 0's state
          } else {
            o[r].co = a.c0[row * Hd + j];
            o[r].cp = a.cin[row * Hd + j];                              // c1(i-1)
          }
        }
      }
    };
    Operands nxt[NT];
    fetch(0, nxt);
    // attention role: the weights of the chunk and the row's context, fetched one step ahead
    float pre_p = 0.f;
    float2 pre_c = make_float2(0.f, 0.f);
    auto fetch_attn = [&](int p) {
      const int i = U - 1 - p;
      if (attn && p < U) {
        const long row = (long)i * a.Bs + ab;
        if (tid < nt) pre_p = a.p[row * T2 + t_lo + tid];
        if (2 * tid < D) pre_c = *reinterpret_cast<const float2*>(a.ctx + row * D + 2 * tid);
      }
    };
    fetch_attn(0);

    for (int p = 0; p <= U; ++p) {
      const int i = U - 1 - p;
      const long slot_cur = (long)(p & 3) * a.slot_floats, slot_prev = (long)((p + 3) & 3) * a.slot_floats;
      // ------------------------------------------------------------------------------------------ cell role
      if (cell && (p < U || (layer == 1 && writer))) {
        float sa[NT], dcin[NT];
#pragma unroll
        for (int r = 0; r < NT; ++r) { sa[r] = 0.f; dcin[r] = 0.f; }
        if (layer == 1 ? p > 0 : true) {
          f32x4 accP = {0.f, 0.f, 0.f, 0.f}, accQ = {0.f, 0.f, 0.f, 0.f}, cpc = {0.f, 0.f, 0.f, 0.f};
          const int npieces = G * NPOS;
          if (layer == 1) {
            unsigned off[9];
            bool use[9];
            f32x4 v[9];
            const float* base = xb + slot_prev;
            const long rowb = a.o_p0 + ((long)(tile * G + gi) * G) * blkf;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int f = tid + 256 * m;
              use[m] = f < npieces;
              off[m] = use[m] ? (unsigned)((rowb + (long)f * 4) * 4) : 0u;
            }
            const int qrow = tid >> 4, cp = (tid & 15) / QPR, quad = tid % QPR;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              use[4 + m] = m < QM && tile * 16 + qrow < B;
              off[4 + m] = use[4 + m] ? (unsigned)((a.o_q + ((long)(tile * 16 + qrow) * DS_NC + cp + CP * m) * Hd + gi * KU + 4 * quad) * 4) : 0u;
            }
            use[8] = true;
            off[8] = (unsigned)((a.o_c0 + (long)(tile * G + gi) * blkf + (long)pos * 4) * 4);
            if (!db_gather9(base, off, use, v, abort_flag, a.spin_limit, a.delay, 1 | (p << 8))) break;
#pragma unroll
            for (int m = 0; m < 4; ++m)
              if (use[m]) accP += v[m];
#pragma unroll
            for (int m = 0; m < 4; ++m)
              if (use[4 + m]) accQ += v[4 + m];
            cpc = v[8];
            *reinterpret_cast<f32x4*>(qs + ((long)cp * 16 + qrow) * KU + 4 * quad) = accQ;
          } else {
            unsigned off[5];
            bool use[5];
            f32x4 v[5];
            const float* base = xb + slot_cur;
            const long rowb = a.o_p1 + ((long)(tile * G + gi) * G) * blkf;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int f = tid + 256 * m;
              use[m] = f < npieces;
              off[m] = use[m] ? (unsigned)((rowb + (long)f * 4) * 4) : 0u;
            }
            use[4] = true;
            off[4] = (unsigned)((a.o_c1 + (long)(tile * G + gi) * blkf + (long)pos * 4) * 4);
            if (!db_gather5(base, off, use, v, abort_flag, a.spin_limit, a.delay, 2 | (p << 8))) break;
#pragma unroll
            for (int m = 0; m < 4; ++m)
              if (use[m]) accP += v[m];
            cpc = v[4];
          }
          *reinterpret_cast<f32x4*>(red + tid * 4) = accP;
          ds_mark(cG, wv, p + 1);
          if (!ds_wait4(cG, p + 1, abort_flag, lds_limit, 5 | (p << 8))) break;
#pragma unroll
          for (int r = 0; r < NT; ++r) {
#pragma unroll
            for (int k = 0; k < TPP; ++k) sa[r] += red[(pos + k * NPOS) * 4 + sub * NT + r];
            if (layer == 1) {
#pragma unroll
              for (int c2 = 0; c2 < CP; ++c2) sa[r] += qs[((long)c2 * 16 + 4 * plq + sub * NT + r) * KU + un];
            }
            dcin[r] = db_pick(cpc, sub * NT + r);
          }
        }
        if (p == U) {                                   // gradient wrt the decoder's initial state (layer 1 writers only)
#pragma unroll
          for (int r = 0; r < NT; ++r)
            if (live[r]) {
              a.dhs[(long)brow[r] * Hd + j] = sa[r];
              a.dc[(long)brow[r] * Hd + j] = dcin[r];
            }
        } else {
#pragma unroll
          for (int r = 0; r < NT; ++r) {
            float ds[4] = {0.f, 0.f, 0.f, 0.f};
            float pass = 0.f, dcout = 0.f;
            if (live[r]) {
              if (!nxt[r].m) {
                pass = sa[r];                           // pad-token row: state carried unchanged, output was zero
                dcout = dcin[r];
              } else {
                const float dh = sa[r] + nxt[r].dy;
                const float ig = nxt[r].sv[0], fg = nxt[r].sv[1], gg = nxt[r].sv[2], og = nxt[r].sv[3];
                const float tc = tanhf_(nxt[r].co);
                const float dct = dcin[r] + dh * og * (1.f - tc * tc);
                ds[0] = dct * gg * ig * (1.f - ig);
                ds[1] = dct * nxt[r].cp * fg * (1.f - fg);
                ds[2] = dct * ig * (1.f - gg * gg);
                ds[3] = dh * tc * og * (1.f - og);
                dcout = dct * fg;
              }
            }
            const int rr = 4 * plq + sub * NT + r;
#pragma unroll
            for (int g = 0; g < 4; ++g) dsl[rr * LD + g * KU + un] = ds[g];
            pass_s[pos * 4 + sub * NT + r] = pass;
            dcs[pos * 4 + sub * NT + r] = dcout;
          }
          ds_mark(cO, wv, p + 1);
          fetch(p + 1, nxt);
        }
      }
      if (p == U) break;
      // ------------------------------------------------------------------------------------------ attention role
      if (attn) {
        if (tid < DS_MAXTC2) ps[tid] = tid < nt ? pre_p : 0.f;
        const float2 cx = pre_c;
        {
          const int QD = D >> 2, SH = 256 / QD;           // piece = (sender, feature quad); SH senders in flight per pass
          const int sg = tid / QD, fq = tid % QD;
          unsigned off[5];
          bool use[5];
          f32x4 v[5];
          const float* base = xb + slot_cur;
          const long rowb = a.o_pc + ((long)ab * G) * D;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int snd = sg + SH * m;
            use[m] = snd < G;
            off[m] = use[m] ? (unsigned)((rowb + (long)snd * D + 4 * fq) * 4) : 0u;
          }
          use[4] = false; off[4] = 0u;
          if (!db_gather5(base, off, use, v, abort_flag, a.spin_limit, a.delay, 3 | (p << 8))) break;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m < 4; ++m)
            if (use[m]) acc += v[m];
          *reinterpret_cast<f32x4*>(dcx + tid * 4) = acc;
        }
        ds_mark(cA1, wv, p + 1);
        if (!ds_wait4(cA1, p + 1, abort_flag, lds_limit, 6 | (p << 8))) break;
        // the row's context gradient (through the input dropout of layer 0) and its dot product with the context
        float dotp = 0.f;
        {
          const int f0 = 2 * tid;
          if (f0 < D) {
            const int QD = D >> 2, SH = 256 / QD;
            float d0 = 0.f, d1 = 0.f;
            for (int sg = 0; sg < SH && sg < G; ++sg) {
              const float2 t2 = *reinterpret_cast<const float2*>(dcx + (sg * QD + (f0 >> 2)) * 4 + (f0 & 3));
              d0 += t2.x; d1 += t2.y;
            }
            if (a.rate > 0.f) {
              const AsrRngKey key = asr_rng_key(seedv, a.stream0 + a.stream_step * (uint32_t)i + 2u);
              const uint32_t idx = (uint32_t)((long)(a.b0 + ab) * (Hd + D) + Hd + f0);
              d0 *= asr_drop_mult(key, idx, thresh, scale);
              d1 *= asr_drop_mult(key, idx + 1, thresh, scale);
            }
            *reinterpret_cast<float2*>(dcv_s + f0) = make_float2(d0, d1);
            dotp = d0 * cx.x + d1 * cx.y;
          }
#pragma unroll
          for (int s = 32; s >= 1; s >>= 1) dotp += __shfl_xor(dotp, s, 64);
          if (lane == 0) dotw[wv] = dotp;
        }
        ds_mark(cA2, wv, p + 1);
        if (!ds_wait4(cA2, p + 1, abort_flag, lds_limit, 7 | (p << 8))) break;
        {
          const float dot = dotw[0] + dotw[1] + dotw[2] + dotw[3];
          const int t = tid >> 3, kg = tid & 7;
          float dp = 0.f;
          if (t < nt) {
            const float* er = enc_s + t * ENLD;
            for (int k4 = kg; k4 < (D >> 2); k4 += 8) {
              const float4 ev = *reinterpret_cast<const float4*>(er + 4 * k4);
              const float4 dv = *reinterpret_cast<const float4*>(dcv_s + 4 * k4);
              dp += dv.x * ev.x + dv.y * ev.y + dv.z * ev.z + dv.w * ev.w;
            }
          }
          dp += __shfl_xor(dp, 1, 64);
          dp += __shfl_xor(dp, 2, 64);
          dp += __shfl_xor(dp, 4, 64);
          if (kg == 0) des[t] = t < nt ? ps[t] * (dp - dot) : 0.f;
          if (nt > DS_MAXTC) {                              // the frames the LDS does not hold: their encoder rows from memory
            const int t2 = DS_MAXTC + t;
            float dp2 = 0.f;
            if (t2 < nt) {
              const float* er = eng + (long)t2 * D;
              for (int k4 = kg; k4 < (D >> 2); k4 += 8) {
                const float4 ev = *reinterpret_cast<const float4*>(er + 4 * k4);
                const float4 dv = *reinterpret_cast<const float4*>(dcv_s + 4 * k4);
                dp2 += dv.x * ev.x + dv.y * ev.y + dv.z * ev.z + dv.w * ev.w;
              }
            }
            dp2 += __shfl_xor(dp2, 1, 64);
            dp2 += __shfl_xor(dp2, 2, 64);
            dp2 += __shfl_xor(dp2, 4, 64);
            if (kg == 0) des[t2] = t2 < nt ? ps[t2] * (dp2 - dot) : 0.f;
          }
        }
        ds_mark(cA3, wv, p + 1);
        if (!ds_wait4(cA3, p + 1, abort_flag, lds_limit, 8 | (p << 8))) break;
        if (tid < Hd) {
          float dq = 0.f;
          for (int t = 0; t < ntr; ++t) dq = fmaf(des[t], kq_s[t * KQLD + tid], dq);
#pragma unroll 8
          for (int t = ntr; t < nt; ++t) dq = fmaf(des[t], kqg[(long)t * Hd + tid], dq);     // (T' > 256) streamed keys, coalesced over tid
          dqs[tid] = dq;
        }
        ds_mark(cA4, wv, p + 1);
        fetch_attn(p + 1);
      }
    }
  } else {
    // ================================================================================================= PUBLISH waves
    const int sw = wv - 4, nt_ = sw % NT, kp = sw / NT;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, (int)a.xbytes, 0x00020000);
    const u32x4 sent = {DS_SENT, DS_SENT, DS_SENT, DS_SENT};
    // resident B operands: k-step ks of this wave carries ds label cl = kp * 4 KPW + lq * KPW + ks <-> (gate cl / KU, unit cl % KU of
    // group gi); the lane (li, lq) holds W[output row of this lane][that gate column]
    float bw[NJ][KPW];
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
      for (int ks = 0; ks < KPW; ++ks) bw[jb][ks] = 0.f;
    if (cell) {
#pragma unroll
      for (int ks = 0; ks < KPW; ++ks) {
        const int cl = kp * 4 * KPW + lq * KPW + ks, g = cl / KU, cu = cl % KU;
        const long col = (long)g * Hd + gi * KU + cu;
        const long orow = (long)(gj * KU + 16 * nt_ + li) * a.ldw;
        if (layer == 1) {
          bw[0][ks] = a.U1[orow + col];
          bw[1][ks] = a.W1[orow + col];
        } else {
          bw[0][ks] = a.U0[orow + col];
#pragma unroll
          for (int x = 0; x < CTM; ++x)
            if (x < CT) bw[1 + x][ks] = a.W0c[(long)(gj * FC + 16 * (nt_ + NT * x) + li) * a.ldw + col];
        }
      }
    }
    const long my_p = ((long)(tile * G + gj) * G + gi) * blkf + (long)nt_ * 256 + lane * 4;   // block (row gj, sender gi), this wave's tile
    float de_acc = 0.f;                                  // (wave 3 of an attention workgroup) this lane's frame: de summed over the steps
    for (int p = 0; p < U; ++p) {
      const int i = U - 1 - p;
      const long slot_cur = (long)(p & 3) * a.slot_floats, slot_old = (long)((p + 2) & 3) * a.slot_floats;
      if (cell) {
        if (!ds_wait4(cO, p + 1, abort_flag, lds_limit, 10 | (p << 8))) break;
        float av[KPW];
#pragma unroll
        for (int ks = 0; ks < KPW; ++ks) av[ks] = dsl[li * LD + kp * 4 * KPW + lq * KPW + ks];
        // the writer column also saves ds (row-major [4 Hd] per batch row): quads of the image, NT per lane
        f32x4 dsv[NT];
        if (writer) {
#pragma unroll
          for (int e = 0; e < NT; ++e) {
            const int qd = sw * 64 + lane + 256 * e, row = qd / (NL / 4), qir = qd % (NL / 4);
            dsv[e] = *reinterpret_cast<const f32x4*>(dsl + row * LD + 4 * qir);
          }
        }
        f32x4 acc[NJ];
#pragma unroll
        for (int jb = 0; jb < NJ; ++jb) acc[jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int njobs = layer == 1 ? 2 : 1 + CT;
#pragma unroll
        for (int ks = 0; ks < KPW; ++ks) {
#pragma unroll
          for (int jb = 0; jb < NJ; ++jb)
            if (jb < njobs) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bw[jb][ks], acc[jb], 0, 0, 0);
        }
        if (KP > 1 && kp > 0) {
#pragma unroll
          for (int jb = 0; jb < NJ; ++jb)
            if (jb < njobs) *reinterpret_cast<f32x4*>(sp + ((long)((kp - 1) * NT + nt_) * NJ + jb) * 256 + lane * 4) = acc[jb];
        }
        ds_mark(cS, sw, p + 1);
        if (kp == 0) {
          if (KP > 1) {
            if (!ds_wait4(cS, p + 1, abort_flag, lds_limit, 11 | (p << 8))) break;
#pragma unroll
            for (int k = 0; k < KP - 1; ++k)
#pragma unroll
              for (int jb = 0; jb < NJ; ++jb)
                if (jb < njobs) acc[jb] += *reinterpret_cast<const f32x4*>(sp + ((long)(k * NT + nt_) * NJ + jb) * 256 + lane * 4);
          }
          f32x4 vh = acc[0];
          if (diag) vh += *reinterpret_cast<const f32x4*>(pass_s + (nt_ * 64 + lane) * 4);
          if (layer == 1) {
            f32x4 dm = {1.f, 1.f, 1.f, 1.f};
            if (a.rate > 0.f) {
              const AsrRngKey key = asr_rng_key(seedv, a.stream0 + a.stream_step * (uint32_t)i + 3u);
              const int k = gj * KU + 16 * nt_ + li;
#pragma unroll
              for (int r = 0; r < 4; ++r) dm[r] = asr_drop_mult(key, (uint32_t)((long)(a.b0 + tile * 16 + 4 * lq + r) * Hd + k), thresh, scale);
            }
            vh += dm * acc[1];
          }
          f32x4 dcp[NT];
          if (sw == 0 && writer) {
#pragma unroll
            for (int n = 0; n < NT; ++n) dcp[n] = *reinterpret_cast<const f32x4*>(dcs + (lane + 64 * n) * 4);
          }
          f32x4 cpiece[CTM];
          if (layer == 0) {
#pragma unroll
            for (int x = 0; x < CTM; ++x)
              if (x < CT) {
                const int tx = nt_ + NT * x;
#pragma unroll
                for (int r = 0; r < 4; ++r) ctxT[(4 * lq + r) * 68 + 16 * tx + li] = acc[1 + x][r];
              }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int x = 0; x < CTM; ++x)
              if (x < CT) cpiece[x] = *reinterpret_cast<const f32x4*>(ctxT + (lane >> 2) * 68 + 16 * (nt_ + NT * x) + 4 * (lane & 3));
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");
          const long op = layer == 1 ? a.o_p1 : a.o_p0, oc = layer == 1 ? a.o_c1 : a.o_c0;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vh), rsrc, (int)((slot_cur + op + my_p) * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)((slot_old + op + my_p) * 4), 0, 16);
          if (layer == 0) {
#pragma unroll
            for (int x = 0; x < CTM; ++x)
              if (x < CT) {
                const long o = a.o_pc + ((long)(tile * 16 + (lane >> 2)) * G + gi) * D + gj * FC + 16 * (nt_ + NT * x) + 4 * (lane & 3);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, cpiece[x]), rsrc, (int)((slot_cur + o) * 4), 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)((slot_old + o) * 4), 0, 16);
              }
          }
          if (sw == 0 && writer) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const long o = oc + (long)(tile * G + gi) * blkf + (long)(lane + 64 * n) * 4;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dcp[n]), rsrc, (int)((slot_cur + o) * 4), 0, 16);
              __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)((slot_old + o) * 4), 0, 16);
            }
          }
        }
        // the layer's ds to memory AFTER the publish: in front of it these stores (to lines nobody has touched yet) would have to be
        // acknowledged before the publish's `s_waitcnt vmcnt(0)` lets the block out - a memory round trip on the critical chain of
        // every step (rounds 1-2 had them there); here they are retired by the next step's wait, a whole exchange round later
        if (writer) {
          float* dso = (layer == 0 ? a.ds0 : a.ds1);
#pragma unroll
          for (int e = 0; e < NT; ++e) {
            const int qd = sw * 64 + lane + 256 * e, row = qd / (NL / 4), qir = qd % (NL / 4);
            const int lab = 4 * qir, g = lab / KU, cu = lab % KU;
            const int br = tile * 16 + row;
            if (br < B) *reinterpret_cast<f32x4*>(dso + ((long)i * a.Bs + br) * 4 * Hd + (long)g * Hd + gi * KU + cu) = dsv[e];
          }
        }
      }
      if (attn && sw == 3) {
        if (!ds_wait4(cA4, p + 1, abort_flag, lds_limit, 12 | (p << 8))) break;
        f32x4 qv = {0.f, 0.f, 0.f, 0.f};
        if (lane < (Hd >> 2)) qv = *reinterpret_cast<const f32x4*>(dqs + 4 * lane);
        const float dcv = lane < FS ? dcv_s[FS * ac + lane] : 0.f;
        const float dev = des[lane];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");
        if (lane < (Hd >> 2)) {
          const long o = a.o_q + ((long)ab * DS_NC + ac) * Hd + 4 * lane;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, qv), rsrc, (int)((slot_cur + o) * 4), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (int)((slot_old + o) * 4), 0, 16);
        }
        const long row = (long)i * a.Bs + ab;
        if (lane < FS) a.dctx[row * D + FS * ac + lane] = dcv;
        if (lane < nt) a.de[row * T2 + t_lo + lane] = dev;
        if (lane < nt) de_acc += dev;
      }
    }
    if (attn && sw == 3 && a.de_sum && lane < nt && !*abort_flag) a.de_sum[(long)ab * T2 + t_lo + lane] = de_acc;
  }
  __syncthreads();
  if (*abort_flag && tid == 0) {
    __hip_atomic_store(a.err, (unsigned)*abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_max(a.err + 1, 0x7fffffffu - (unsigned)(((*abort_flag >> 8) << 8) | (*abort_flag & 255)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.err[32 + w] = (unsigned)*abort_flag;
    swd_record(a.err, (unsigned)*abort_flag, 0);
    if (a.err_flag) __hip_atomic_store(reinterpret_cast<unsigned*>(a.err_flag), 0x3F800000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) swd_depart(a.err);
}

static bool db_geometry(int Hd, int D, int* nt, int* G) {
  if (Hd <= 0 || Hd % 16 != 0 || Hd > 256) return false;
  const int t = Hd > 128 ? 2 : 1;
  if (Hd % (16 * t) != 0) return false;
  const int g = Hd / (16 * t);
  if (g < 1 || g > 8) return false;
  const int qd = D / 4;
  if (D <= 0 || D % 4 != 0 || qd > 128 || (qd & (qd - 1)) != 0 || qd < 4) return false;
  if (D % (16 * g) != 0) return false;
  const int fc = D / g;
  if (fc > 64 || (fc / 16) % t != 0) return false;
  if (D % 8 != 0 || D / 8 > 64) return false;
  *nt = t; *G = g;
  return true;
}

static size_t db_lds_bytes(int Hd, int D, int nt) {
  const int KU = 16 * nt, NL = 4 * KU, KP = 4 / nt, NPOS = 64 * nt, NJ = 1 + (nt == 2 ? 2 : 4);
  const size_t floats = (size_t)DS_MAXTC * (Hd + 4) + (size_t)DS_MAXTC * (D + 4) + 1024 + 1024 + 16 * (NL + 4) + 2 * NPOS * 4 + (size_t)(KP - 1) * nt * NJ * 256 +
                        16 * 68 + 1024 + 512 + 8 + DS_MAXTC2 + DS_MAXTC2 + 256 + 64;
  return floats * sizeof(float);
}

static void db_layout(int Hd, int D, int nt, int G, long* o_p1, long* o_c1, long* o_p0, long* o_c0, long* o_pc, long* o_q, long* slot) {
  const long blkf = (long)nt * 256;
  const long psz = 2L * G * G * blkf, csz = 2L * G * blkf;
  *o_p1 = 0; *o_c1 = psz; *o_p0 = psz + csz; *o_c0 = 2 * psz + csz;
  *o_pc = 2 * psz + 2 * csz;
  *o_q = *o_pc + 32L * G * D;
  *slot = *o_q + 32L * DS_NC * Hd;
}

extern "C" int asr_decoder_sweep_bwd_supported(int rnn_type, int num_layers, int B, int U, int T2, int Hd, int D) {
  if (rnn_type != 0 || num_layers != 2 || B <= 0 || B > 64 || U < 1 || T2 < 1 || T2 > DS_NC * DS_MAXTC2) return 0;   // (see asr_decoder_sweep_supported)
  int nt, G;
  if (!db_geometry(Hd, D, &nt, &G)) return 0;
  if (db_lds_bytes(Hd, D, nt) > 160 * 1024) return 0;
  return 1;
}

extern "C" long asr_decoder_sweep_bwd_ws_floats(int Hd, int D) {
  int nt, G;
  if (!db_geometry(Hd, D, &nt, &G)) return 32 + 256;
  long a, b, c, d, e, f, slot;
  db_layout(Hd, D, nt, G, &a, &b, &c, &d, &e, &f, &slot);
  return DS_SLOTS * slot + 32 + 256;
}

extern "C" int asr_decoder_sweep_bwd(const asr_decoder_sweep_grad* s, float* ws, float* err_flag, void* stream) {
  ASR_CHECK(s && ws, ASR_ERR_ARG, "asr_decoder_sweep_bwd: null argument");
  ASR_CHECK(asr_decoder_sweep_bwd_supported(0, 2, s->B, s->U, s->T2, s->Hd, s->D), ASR_ERR_UNSUPPORTED, "asr_decoder_sweep_bwd: shape not supported");
  ASR_CHECK(s->Kq && s->enc && s->p && s->ctx && s->saved0 && s->saved1 && s->cin && s->c0 && s->tokmask && s->dy1 && s->U1 && s->W1 && s->U0 && s->W0 &&
                s->ds0 && s->ds1 && s->de && s->dctx && s->dh_init && s->dc_init,
            ASR_ERR_ARG, "asr_decoder_sweep_bwd: null buffer");
  ASR_CHECK(s->ds0 != s->saved0 && s->ds1 != s->saved1, ASR_ERR_ARG, "asr_decoder_sweep_bwd: ds must not alias the saved activations");
  ASR_CHECK(!(s->drop_rate > 0.f && !s->seed), ASR_ERR_ARG, "asr_decoder_sweep_bwd: dropout needs a device seed");
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
    ASR_CHECK(cus >= 256, ASR_ERR_UNSUPPORTED, "asr_decoder_sweep_bwd: needs 256 compute units (one resident workgroup each), device has %d", cus);
  hipStream_t st = (hipStream_t)stream;
  int nt = 1, G = 1;
  db_geometry(s->Hd, s->D, &nt, &G);
  DbArgs a{};
  a.B = s->B; a.U = s->U; a.T2 = s->T2; a.Hd = s->Hd; a.D = s->D;
  a.TC = asr_cdiv(s->T2, DS_NC); a.G = G; a.FC = s->D / G; a.FS = s->D / DS_NC;
  a.Kq = s->Kq; a.enc = s->enc; a.p = s->p; a.ctx = s->ctx;
  a.saved0 = s->saved0; a.saved1 = s->saved1; a.ds0 = s->ds0; a.ds1 = s->ds1;
  a.cin = s->cin; a.c0 = s->c0; a.tokmask = s->tokmask; a.dy1 = s->dy1; a.dy1_ld = s->dy1_ld;
  a.U1 = s->U1; a.W1 = s->W1; a.U0 = s->U0; a.W0c = s->W0 + (long)s->Hd * 4 * s->Hd; a.ldw = 4L * s->Hd;
  a.seed = s->seed; a.rate = s->drop_rate; a.stream0 = s->drop_stream0; a.stream_step = s->drop_stream_step;
  a.de = s->de; a.dctx = s->dctx; a.dhs = s->dh_init; a.dc = s->dc_init;
  db_layout(s->Hd, s->D, nt, G, &a.o_p1, &a.o_c1, &a.o_p0, &a.o_c0, &a.o_pc, &a.o_q, &a.slot_floats);
  const long xfloats = DS_SLOTS * a.slot_floats;
  ASR_CHECK(xfloats * 4 < (1L << 31), ASR_ERR_UNSUPPORTED, "asr_decoder_sweep_bwd: exchange buffer too large");
  a.xbuf = ws; a.xbytes = xfloats * 4;
  a.err = reinterpret_cast<unsigned*>(ws + xfloats);
  a.err_flag = err_flag;
  a.spin_limit = asr_rnn_sweep_spin_limit();
  a.prio = asr_sweep_prio();
  // (las_small geometry: backward sweep 14.72 -> 14.36 us per decoder step with the rows on XCDs; the forward sweep loses, 15.8 -> 16.4, and keeps 0)
  a.cellxcd = getenv("ASR_DECODER_SWEEP_CELLXCD") ? atoi(getenv("ASR_DECODER_SWEEP_CELLXCD")) : 1;   // (14.5 -> 14.15 us per decoder step)
  a.rowxcd = getenv("ASR_DECODER_SWEEP_ROWXCD") ? atoi(getenv("ASR_DECODER_SWEEP_ROWXCD")) : 1;
  a.delay = getenv("ASR_DECODER_SWEEP_BWD_DELAY") ? atoi(getenv("ASR_DECODER_SWEEP_BWD_DELAY")) : 4;
  const size_t smem = db_lds_bytes(s->Hd, s->D, nt);
  static unsigned long long attr = 0;
  if (asr_first_use_on_device(attr)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_sweep_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_sweep_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  a.Bs = s->B;
  for (int b0 = 0; b0 < s->B; b0 += 32) {                 // one launch per 32 batch rows (see asr_decoder_sweep_fwd)
    const long T2 = s->T2, Hd = s->Hd, D = s->D;
    a.B = s->B - b0 < 32 ? s->B - b0 : 32; a.b0 = b0;
    a.Kq = s->Kq + b0 * T2 * Hd; a.enc = s->enc + b0 * T2 * D; a.p = s->p + b0 * T2; a.ctx = s->ctx + b0 * D;
    a.saved0 = s->saved0 + b0 * 4 * Hd; a.saved1 = s->saved1 + b0 * 4 * Hd; a.ds0 = s->ds0 + b0 * 4 * Hd; a.ds1 = s->ds1 + b0 * 4 * Hd;
    a.cin = s->cin + b0 * Hd; a.c0 = s->c0 + b0 * Hd; a.tokmask = s->tokmask + b0; a.dy1 = s->dy1 + b0 * s->dy1_ld;
    a.de = s->de + b0 * T2; a.dctx = s->dctx + b0 * D; a.dhs = s->dh_init + b0 * Hd; a.dc = s->dc_init + b0 * Hd;
    a.de_sum = s->de_sum ? s->de_sum + b0 * T2 : nullptr;
    const size_t n = (size_t)xfloats;
    hipLaunchKernelGGL(sw_fill_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, st,
                       reinterpret_cast<uint32_t*>(ws), n, DS_SENT, a.err, 16, 256u);
    (void)asr_zero_async(a.err + 32, 256 * sizeof(unsigned), st);
    ASR_LAUNCH_CHECK();
    if (nt == 1) hipLaunchKernelGGL(decoder_sweep_bwd_kernel<1>, dim3(256), dim3(512), smem, st, a);
    else hipLaunchKernelGGL(decoder_sweep_bwd_kernel<2>, dim3(256), dim3(512), smem, st, a);
    ASR_LAUNCH_CHECK();
  }
  return ASR_OK;
}
