// One-launch attention steps for the LAS decoder (las.py:43-59): the same arithmetic as attention.hip's
// two-kernel steps, restructured so that a decoder step costs one dependent launch instead of two.
//
// Grid = (NCH chunks of the T' axis) x (B batch rows) = 256 workgroups for las_small: every CU takes one
// (row, chunk) pair, i.e. 32 key rows: scores -> chunk-local softmax statistics (max m_w, sum s_w of
// exp(e - m_w)) -> partial context sum_t exp(e_t - m_w) enc[t] over all D columns.  The partials (2 + D
// floats per chunk) are combined by the LAST workgroup of the row to arrive at an agent-scope ticket
// (cdna_hip_programming.md section 5, "In-launch split-K reduction": plain stores, s_waitcnt vmcnt(0), barrier,
// one agent release fence before the relaxed ticket add; the reducer takes one agent acquire fence and
// then reads the slabs with plain loads).  The ticket counters only ever increase (the last arriver is
// the one that draws a value = NCH-1 mod NCH), so they are zeroed once at allocation and need no per-call reset -
// a captured graph replays the kernel unchanged.
//   softmax over chunks:  M = max_w m_w,  S = sum_w s_w exp(m_w - M),  p_t = exp(e_t - m_w(t)) exp(m_w(t) - M) / S
// Backward: dp_t = dctx . enc[t];  with dot = sum_t p_t dp_t:  ds_t = p_t (dp_t - dot),
//   dh = sum_t ds_t Kq[t] = sum_w (A_w - dot * Bv_w),  A_w = sum_t p_t dp_t Kq[t],  Bv_w = sum_t p_t Kq[t].
#include "common.h"

#define AF_NCH 8          // chunks per batch row (power of two: the ticket test is a mask)
#define AF_MAXROWS 64     // rows per chunk held in LDS
#define AF_RPW 4          // score rows in flight per wave

struct AfArgs {
  const float* v; long ldv;          // query vector per batch row: h (forward) or dctx (backward)
  const float* Kq; const float* s0; const uint8_t* mask; const float* enc;
  const float* p_in;                 // backward: attention probabilities
  int B, T, Hd, D, TC;
  float* scratch; unsigned* tickets;
  float* p; float* out; long ldout;  // forward: p [B,T], ctx [B,D];  backward: ds [B,T], dh [B,Hd]
  int accumulate;
};

// s[r] = v . M[row0 + r, :K] for AF_RPW rows of one wave (K % 4 == 0, 16-byte aligned rows)
__device__ __forceinline__ void af_rowdots(const float* v, const float* Mb, int K, int row0, int nrows, float (&s)[AF_RPW]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int r = 0; r < AF_RPW; ++r) s[r] = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 hv = *reinterpret_cast<const float4*>(v + k);
    float4 kv[AF_RPW];
#pragma unroll
    for (int r = 0; r < AF_RPW; ++r)
      kv[r] = (row0 + r < nrows) ? *reinterpret_cast<const float4*>(Mb + (long)(row0 + r) * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < AF_RPW; ++r) s[r] += hv.x * kv[r].x + hv.y * kv[r].y + hv.z * kv[r].z + hv.w * kv[r].w;
  }
#pragma unroll
  for (int r = 0; r < AF_RPW; ++r) s[r] = wave_sum(s[r]);
}

// publish this workgroup's slab and draw a ticket; returns true in every thread of the last arriver
__device__ __forceinline__ bool af_arrive(unsigned* ticket, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (old & (AF_NCH - 1)) == AF_NCH - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}

__global__ __launch_bounds__(256) void attn_fused_fwd_kernel(AfArgs a) {
  __shared__ float sh[AF_MAXROWS + 2 * AF_NCH + 16 + 4];   // chunk p | per-chunk scale | m,s staging | reduction scratch | flag
  float* pl = sh;
  float* scale = sh + AF_MAXROWS;
  float* red = sh + AF_MAXROWS + 2 * AF_NCH;
  int* flag = reinterpret_cast<int*>(sh + AF_MAXROWS + 2 * AF_NCH + 16);
  const int w = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = a.T, D = a.D, TC = a.TC;
  const int t0 = w * TC, nrows = max(0, min(TC, T - t0));
  const float* vb = a.v + (long)b * a.ldv;
  // 1. scores of this chunk
  for (int r0 = wave * AF_RPW; r0 < nrows; r0 += 4 * AF_RPW) {
    float s[AF_RPW];
    af_rowdots(vb, a.Kq + ((long)b * T + t0) * a.Hd, a.Hd, r0, nrows, s);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < AF_RPW; ++r) {
        const int t = t0 + r0 + r;
        if (r0 + r < nrows) {
          float val = s[r] + (a.s0 ? a.s0[(long)b * T + t] : 0.f);
          val -= 1e9f * (1.0f - (a.mask[(long)b * T + t] ? 1.0f : 0.0f));
          pl[r0 + r] = val;
        }
      }
    }
  }
  __syncthreads();
  // 2. chunk-local softmax statistics
  float mx = -INFINITY;
  for (int r = tid; r < nrows; r += 256) mx = fmaxf(mx, pl[r]);
  mx = block_max(mx, red);
  float sum = 0.f;
  for (int r = tid; r < nrows; r += 256) { const float e = expf(pl[r] - mx); pl[r] = e; sum += e; }
  sum = block_sum(sum, red);
  __syncthreads();
  // 3. partial context over all D columns (coalesced rows of enc), 8 rows in flight
  float* slab = a.scratch + ((long)b * AF_NCH + w) * (2 + D);
  const float* eb = a.enc + ((long)b * T + t0) * D;
  for (int c = tid; c < D; c += 256) {
    float acc = 0.f;
    for (int r0 = 0; r0 < nrows; r0 += 8) {
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = r0 + i < nrows ? eb[(long)(r0 + i) * D + c] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = fmaf(r0 + i < nrows ? pl[r0 + i] : 0.f, x[i], acc);
    }
    slab[2 + c] = acc;
  }
  if (tid == 0) { slab[0] = nrows > 0 ? mx : -INFINITY; slab[1] = nrows > 0 ? sum : 0.f; }
  for (int r = tid; r < nrows; r += 256) a.p[(long)b * T + t0 + r] = pl[r];      // unnormalised, rescaled by the reducer
  // 4. the last chunk of this row to arrive combines
  if (!af_arrive(a.tickets + b, flag)) return;
  const float* sb = a.scratch + (long)b * AF_NCH * (2 + D);
  if (tid < AF_NCH) { scale[tid] = sb[(long)tid * (2 + D)]; scale[AF_NCH + tid] = sb[(long)tid * (2 + D) + 1]; }
  __syncthreads();
  float M = -INFINITY;
#pragma unroll
  for (int i = 0; i < AF_NCH; ++i) M = fmaxf(M, scale[i]);
  float S = 0.f, sc[AF_NCH];
#pragma unroll
  for (int i = 0; i < AF_NCH; ++i) { sc[i] = scale[AF_NCH + i] > 0.f ? expf(scale[i] - M) : 0.f; S += scale[AF_NCH + i] * sc[i]; }
  const float inv = 1.f / S;
  for (int c = tid; c < D; c += 256) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < AF_NCH; ++i) acc = fmaf(sc[i], sb[(long)i * (2 + D) + 2 + c], acc);
    a.out[(long)b * a.ldout + c] = acc * inv;
  }
  for (int t = tid; t < T; t += 256) a.p[(long)b * T + t] *= sc[t / TC] * inv;
}

__global__ __launch_bounds__(256) void attn_fused_bwd_kernel(AfArgs a) {
  __shared__ float sh[2 * AF_MAXROWS + AF_NCH + 16 + 4];    // p dp | p | per-chunk dots | reduction scratch | flag
  float* pd = sh;
  float* pp = sh + AF_MAXROWS;
  float* dots = sh + 2 * AF_MAXROWS;
  float* red = sh + 2 * AF_MAXROWS + AF_NCH;
  int* flag = reinterpret_cast<int*>(sh + 2 * AF_MAXROWS + AF_NCH + 16);
  const int w = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = a.T, Hd = a.Hd, TC = a.TC;
  const int t0 = w * TC, nrows = max(0, min(TC, T - t0));
  const float* vb = a.v + (long)b * a.ldv;       // dctx row
  // 1. dp_t = dctx . enc[t]; keep p_t and p_t dp_t for the chunk, publish dp (as ds, finished by the reducer)
  for (int r0 = wave * AF_RPW; r0 < nrows; r0 += 4 * AF_RPW) {
    float s[AF_RPW];
    af_rowdots(vb, a.enc + ((long)b * T + t0) * a.D, a.D, r0, nrows, s);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < AF_RPW; ++r)
        if (r0 + r < nrows) {
          const float pt = a.p_in[(long)b * T + t0 + r0 + r];
          pp[r0 + r] = pt;
          pd[r0 + r] = pt * s[r];
          a.p[(long)b * T + t0 + r0 + r] = s[r];          // dp for now
        }
    }
  }
  __syncthreads();
  float dot = 0.f;
  for (int r = tid; r < nrows; r += 256) dot += pd[r];
  dot = block_sum(dot, red);
  // 2. A_w[c] = sum_t p dp Kq[t,c], Bv_w[c] = sum_t p Kq[t,c]
  float* slab = a.scratch + ((long)b * AF_NCH + w) * (1 + 2 * Hd);
  const float* kb = a.Kq + ((long)b * T + t0) * Hd;
  for (int c = tid; c < Hd; c += 256) {
    float accA = 0.f, accB = 0.f;
    for (int r0 = 0; r0 < nrows; r0 += 8) {
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = r0 + i < nrows ? kb[(long)(r0 + i) * Hd + c] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool ok = r0 + i < nrows;
        accA = fmaf(ok ? pd[r0 + i] : 0.f, x[i], accA);
        accB = fmaf(ok ? pp[r0 + i] : 0.f, x[i], accB);
      }
    }
    slab[1 + c] = accA;
    slab[1 + Hd + c] = accB;
  }
  if (tid == 0) slab[0] = dot;
  if (!af_arrive(a.tickets + b, flag)) return;
  const float* sb = a.scratch + (long)b * AF_NCH * (1 + 2 * Hd);
  if (tid < AF_NCH) dots[tid] = sb[(long)tid * (1 + 2 * Hd)];
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < AF_NCH; ++i) tot += dots[i];
  for (int c = tid; c < Hd; c += 256) {
    float accA = 0.f, accB = 0.f;
#pragma unroll
    for (int i = 0; i < AF_NCH; ++i) { accA += sb[(long)i * (1 + 2 * Hd) + 1 + c]; accB += sb[(long)i * (1 + 2 * Hd) + 1 + Hd + c]; }
    const float v = accA - tot * accB;
    float* o = a.out + (long)b * a.ldout + c;
    *o = a.accumulate ? *o + v : v;
  }
  for (int t = tid; t < T; t += 256) a.p[(long)b * T + t] = a.p_in[(long)b * T + t] * (a.p[(long)b * T + t] - tot);   // ds
}

static int af_check(int B, int T, int Hd, int D, const void* p0, const void* p1, long ld) {
  ASR_CHECK(B > 0 && T > 0 && Hd > 0 && D > 0, ASR_ERR_SHAPE, "asr_attn_fused: bad shape");
  ASR_CHECK(Hd % 4 == 0 && D % 4 == 0 && ld % 4 == 0 && ((((uintptr_t)p0) | ((uintptr_t)p1)) & 15) == 0, ASR_ERR_SHAPE,
            "asr_attn_fused: Hd, D and the row strides must be multiples of 4 and the buffers 16-byte aligned");
  ASR_CHECK(asr_cdiv(T, AF_NCH) <= AF_MAXROWS, ASR_ERR_SHAPE, "asr_attn_fused: T=%d > %d frames (use asr_attn_step_fwd/bwd)", T, AF_NCH * AF_MAXROWS);
  return ASR_OK;
}

extern "C" long asr_attn_fused_ws_floats(int B, int Hd, int D) {
  const long per = 2 + (long)(D > 2 * Hd ? D : 2 * Hd);
  return (long)B * AF_NCH * per;
}
extern "C" int asr_attn_fused_supported(int T, int Hd, int D) {
  return (T > 0 && Hd > 0 && D > 0 && Hd % 4 == 0 && D % 4 == 0 && asr_cdiv(T, AF_NCH) <= AF_MAXROWS) ? 1 : 0;
}

// Same contract as asr_attn_step_fwd, one launch.  scratch: asr_attn_fused_ws_floats() floats; tickets: B uint32 words
// that are zero when first used and are only ever passed to these two functions (they count up by AF_NCH per call).
extern "C" int asr_attn_fused_fwd(const float* h, long ldh, const float* Kq, const float* s0, const uint8_t* mask, const float* enc, int B,
                                  int T, int Hd, int D, float* scratch, uint32_t* tickets, float* p, float* ctx, long ldctx, void* stream) {
  ASR_CHECK(h && Kq && mask && enc && scratch && tickets && p && ctx, ASR_ERR_ARG, "asr_attn_fused_fwd: null argument");
  int rc = af_check(B, T, Hd, D, h, Kq, ldh);
  if (rc) return rc;
  AfArgs a{};
  a.v = h; a.ldv = ldh; a.Kq = Kq; a.s0 = s0; a.mask = mask; a.enc = enc;
  a.B = B; a.T = T; a.Hd = Hd; a.D = D; a.TC = asr_cdiv(T, AF_NCH);
  a.scratch = scratch; a.tickets = tickets; a.p = p; a.out = ctx; a.ldout = ldctx;
  hipLaunchKernelGGL(attn_fused_fwd_kernel, dim3(AF_NCH, (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// Same contract as asr_attn_step_bwd (no dp scratch needed), one launch.
extern "C" int asr_attn_fused_bwd(const float* dctx, long lddctx, const float* p, const float* Kq, const float* enc, int B, int T, int Hd,
                                  int D, float* scratch, uint32_t* tickets, float* ds, float* dh, long lddh, int accumulate, void* stream) {
  ASR_CHECK(dctx && p && Kq && enc && scratch && tickets && ds && dh, ASR_ERR_ARG, "asr_attn_fused_bwd: null argument");
  int rc = af_check(B, T, Hd, D, dctx, enc, lddctx);
  if (rc) return rc;
  AfArgs a{};
  a.v = dctx; a.ldv = lddctx; a.Kq = Kq; a.enc = enc; a.p_in = p;
  a.B = B; a.T = T; a.Hd = Hd; a.D = D; a.TC = asr_cdiv(T, AF_NCH);
  a.scratch = scratch; a.tickets = tickets; a.p = ds; a.out = dh; a.ldout = lddh; a.accumulate = accumulate;
  hipLaunchKernelGGL(attn_fused_bwd_kernel, dim3(AF_NCH, (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
