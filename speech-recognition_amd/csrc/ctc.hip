// CTC loss and gradient for gfx950.
// tf.nn.ctc_loss(labels dense, logits [B,T,V] batch-major, label_length, logit_length = T, blank_index)
// exactly as CTCLoss.call uses it (measure.py:32-42): label_length = count(y != pad), every row uses
// the full T frames (the encoder mask is ignored), per-sample negative log-likelihood / label_length,
// Keras mean over the batch.  Three kernels:
//   ctc_rows     one workgroup per (b,t): log-sum-exp of the logits row (staged once in LDS) and the
//                log-probabilities of the 2L+1 extended-label symbols gathered into lp [B,T,S]
//   ctc_lattice  one workgroup per sample: alpha recursion (rolling rows in LDS; rows also streamed to
//                global for the second pass), the loss, then the beta recursion producing the state
//                posteriors gamma[b,t,s] = P(state s at time t | labels)
//   ctc_grad     one workgroup per (b,t): dlogits = scale_b (softmax - scatter(gamma)), written in place
// S = 2 L + 1;  ext[s] = blank for even s, labels[b][s/2] for odd s.  HBM traffic: the logits are read
// twice and written once; everything else is O(B T L).
#include "common.h"

#define CTC_NEG (-1e30f)
__device__ __forceinline__ float logaddexp_(float a, float b) {
  const float m = fmaxf(a, b);
  if (m <= CTC_NEG) return CTC_NEG;
  return m + log1pf(expf(-fabsf(a - b)));
}
// log(e^a + e^b + e^c) for the lattice recursions: one max, three hardware exponentials, one hardware logarithm (the two nested
// logaddexp calls cost two expf + two log1pf per state and step on the dependent chain).  Absolute error ~1e-7 in the log domain.
__device__ __forceinline__ float ctc_lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m <= CTC_NEG) return CTC_NEG;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}
__device__ __forceinline__ int ctc_label_len(const int32_t* lab, int L, int pad) {
  int n = 0;
  for (int i = 0; i < L; ++i) n += (lab[i] != pad) ? 1 : 0;  // measure.py:36 count_nonzero(y != pad)
  return n;
}
__device__ __forceinline__ int ctc_sym(const int32_t* lab, int s, int blank, int V) {
  int sym = (s & 1) ? lab[s >> 1] : blank;
  return sym < 0 ? 0 : (sym >= V ? V - 1 : sym);
}

template <bool IN_LDS>
__global__ __launch_bounds__(256) void ctc_rows_kernel(const float* logits, long ld, const int32_t* labels, int T, int V, int L, int blank,
                                                       float* lse_out, float* lp) {
  extern __shared__ float row[];
  __shared__ float red[16];
  const int r = blockIdx.x, b = r / T, tid = threadIdx.x;
  const float* x = logits + (long)r * ld;
  float mx = -INFINITY;
  for (int c = tid; c < V; c += 256) { const float v = x[c]; if (IN_LDS) row[c] = v; mx = fmaxf(mx, v); }
  mx = block_max(mx, red);
  float s = 0.f;
  for (int c = tid; c < V; c += 256) s += expf((IN_LDS ? row[c] : x[c]) - mx);
  s = block_sum(s, red);
  const float lse = mx + logf(s);
  if (tid == 0) lse_out[r] = lse;
  const int32_t* lab = labels + (long)b * L;
  const int S = 2 * L + 1;
  for (int sidx = tid; sidx < S; sidx += 256) {
    const int sym = ctc_sym(lab, sidx, blank, V);
    lp[(long)r * S + sidx] = (IN_LDS ? row[sym] : x[sym]) - lse;
  }
}

// Barrier for the rolling LDS rows of the lattice: orders LDS only.  __syncthreads() also waits for every outstanding global
// store (the alpha / gamma rows written each step) to be acknowledged - a global-memory round trip on each of the 2 T dependent
// steps; nothing in the lattice needs those stores before the kernel ends, except alpha[t][s], which the SAME thread reads back.
__device__ __forceinline__ void ctc_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(256) void ctc_lattice_kernel(const float* lp, float* alpha, float* gamma, const int32_t* labels, int T, int L,
                                                          int pad, float* per_sample, float* stats, float inv_batch) {
  extern __shared__ float sh[];  // two rolling rows of S
  const int b = blockIdx.x, tid = threadIdx.x;
  const int S = 2 * L + 1;
  const int32_t* lab = labels + (long)b * L;
  const int len = ctc_label_len(lab, L, pad);
  const int Sb = 2 * len + 1;
  const float* lpb = lp + (long)b * T * S;
  float* al = alpha + (long)b * T * S;
  float* ga = gamma + (long)b * T * S;
  float* r0 = sh;
  float* r1 = sh + S;
  // ---- alpha[t][s]: log-probability of all prefixes ending in state s at time t (emission at t included)
  for (int s = tid; s < S; s += 256) {
    const float v = (s < 2 && s < Sb) ? lpb[s] : CTC_NEG;
    r0[s] = v;
    al[s] = v;
  }
  ctc_lds_barrier();
  // The emission log-probabilities of the NEXT frame are loaded before the barrier of the current one: the barrier orders memory, so
  // a load placed after it would add a global-memory round trip to each of the T dependent steps (measured 1.4 us per step)
  const bool one_pass = S <= 256;                         // one state per thread: the prefetch registers below cover the row
  // whether state tid may be entered from tid - 2 / left towards tid + 2 (distinct neighbouring labels): constant over time
  const bool skip_a = tid < Sb && tid >= 2 && (tid & 1) && lab[tid >> 1] != lab[(tid >> 1) - 1];
  const bool skip_b = tid + 2 < Sb && (tid & 1) && lab[(tid >> 1) + 1] != lab[tid >> 1];
  float lp_nx = (one_pass && tid < S && T > 1) ? lpb[(long)S + tid] : 0.f;
  for (int t = 1; t < T; ++t) {
    const float* prev = (t & 1) ? r0 : r1;
    float* cur = (t & 1) ? r1 : r0;
    const float lp_t = lp_nx;
    if (one_pass && tid < S && t + 1 < T) lp_nx = lpb[(long)(t + 1) * S + tid];
    for (int s = tid; s < S; s += 256) {
      float v = CTC_NEG;
      if (s < Sb) {
        const bool skip = one_pass ? skip_a : (s >= 2 && (s & 1) && lab[s >> 1] != lab[(s >> 1) - 1]);
        v = ctc_lse3(prev[s], s >= 1 ? prev[s - 1] : CTC_NEG, skip ? prev[s - 2] : CTC_NEG);
        v = v <= CTC_NEG ? CTC_NEG : v + (one_pass ? lp_t : lpb[(long)t * S + s]);
      }
      cur[s] = v;
      al[(long)t * S + s] = v;
    }
    ctc_lds_barrier();
  }
  const float* last = ((T - 1) & 1) ? r1 : r0;
  float ll = last[Sb - 1];
  if (Sb >= 2) ll = logaddexp_(ll, last[Sb - 2]);
  ctc_lds_barrier();
  if (tid == 0) {
    const float per = -ll / (float)len;   // measure.py:41 divides by the label length, unguarded like the reference
    per_sample[b] = per;
    atomicAdd(&stats[0], per * inv_batch);
  }
  // ---- beta[t][s]: log-probability of completing the labelling from state s after time t
  for (int s = tid; s < S; s += 256) {
    const float v = (s < Sb && s >= Sb - 2) ? 0.f : CTC_NEG;
    r0[s] = v;
    const float a = al[(long)(T - 1) * S + s];
    ga[(long)(T - 1) * S + s] = (a > CTC_NEG && v > CTC_NEG) ? expf(a + v - ll) : 0.f;
  }
  ctc_lds_barrier();
  // same here: the three emission values and alpha of the next iteration are in registers before the barrier
  float b0n = 0.f, b1n = 0.f, b2n = 0.f, an = 0.f;
  auto bload = [&](int t) {
    const float* lpn = lpb + (long)(t + 1) * S;
    b0n = lpn[tid];
    b1n = tid + 1 < S ? lpn[tid + 1] : 0.f;
    b2n = tid + 2 < S ? lpn[tid + 2] : 0.f;
    an = al[(long)t * S + tid];
  };
  if (one_pass && tid < S && T >= 2) bload(T - 2);
  for (int t = T - 2, k = 0; t >= 0; --t, ++k) {
    const float* nxt = (k & 1) ? r1 : r0;
    float* cur = (k & 1) ? r0 : r1;
    const float* lpn = lpb + (long)(t + 1) * S;
    const float c0 = b0n, c1 = b1n, c2 = b2n, ca = an;
    if (one_pass && tid < S && t >= 1) bload(t - 1);
    for (int s = tid; s < S; s += 256) {
      float v = CTC_NEG;
      if (s < Sb) {
        const bool skip = one_pass ? skip_b : (s + 2 < Sb && (s & 1) && lab[(s >> 1) + 1] != lab[s >> 1]);
        const float t0 = nxt[s] <= CTC_NEG ? CTC_NEG : nxt[s] + (one_pass ? c0 : lpn[s]);
        const float t1 = (s + 1 < Sb && nxt[s + 1] > CTC_NEG) ? nxt[s + 1] + (one_pass ? c1 : lpn[s + 1]) : CTC_NEG;
        const float t2 = (skip && nxt[s + 2] > CTC_NEG) ? nxt[s + 2] + (one_pass ? c2 : lpn[s + 2]) : CTC_NEG;
        v = ctc_lse3(t0, t1, t2);
      }
      cur[s] = v;
      const float a = one_pass ? ca : al[(long)t * S + s];
      ga[(long)t * S + s] = (a > CTC_NEG && v > CTC_NEG) ? expf(a + v - ll) : 0.f;
    }
    ctc_lds_barrier();
  }
}

template <bool IN_LDS>
__global__ __launch_bounds__(256) void ctc_grad_kernel(float* logits, long ld, const int32_t* labels, const float* lse, const float* gamma,
                                                       int T, int V, int L, int blank, int pad, float scale) {
  extern __shared__ float row[];
  const int r = blockIdx.x, b = r / T, tid = threadIdx.x;
  float* x = logits + (long)r * ld;
  const int32_t* lab = labels + (long)b * L;
  const int len = ctc_label_len(lab, L, pad);
  const int Sb = 2 * len + 1, S = 2 * L + 1;
  const float sc = scale / (float)len;
  const float l = lse[r];
  for (int c = tid; c < V; c += 256) {
    const float v = sc * expf(x[c] - l);
    if (IN_LDS) row[c] = v; else x[c] = v;
  }
  __syncthreads();
  for (int s = tid; s < Sb; s += 256) {
    const float gsub = -sc * gamma[(long)r * S + s];
    const int sym = ctc_sym(lab, s, blank, V);
    if (IN_LDS) atomicAdd(&row[sym], gsub); else atomicAdd(&x[sym], gsub);
  }
  if (IN_LDS) {
    __syncthreads();
    for (int c = tid; c < V; c += 256) x[c] = row[c];
  }
}

// workspace (floats): lse [B*T] + lp, alpha, gamma [B*T*S each] + per-sample loss [B]
extern "C" long asr_ctc_workspace_floats(int B, int T, int L) { return (long)B * T * (1 + 3L * (2 * L + 1)) + B; }

// logits [B*T, V] (row stride ld) are overwritten by grad_scale * d(mean_b nll_b/len_b)/d logits when
// write_grad.  stats[0] += the loss (zero it first); per_sample [B] receives nll_b / len_b.
extern "C" int asr_ctc_loss(float* logits, long ld, const int32_t* labels, int B, int T, int V, int L, int blank, int pad, float* ws,
                            float* per_sample, float* stats, int write_grad, float grad_scale, void* stream) {
  ASR_CHECK(logits && labels && ws && stats && per_sample, ASR_ERR_ARG, "asr_ctc_loss: null argument");
  ASR_CHECK(B > 0 && T > 0 && V > 0 && L > 0 && ld >= V && blank >= 0 && blank < V, ASR_ERR_SHAPE, "asr_ctc_loss: bad shape / blank index");
  hipStream_t st = (hipStream_t)stream;
  const int S = 2 * L + 1;
  const long n = (long)B * T * S;
  float* lse = ws;
  float* lp = lse + (long)B * T;
  float* alpha = lp + n;
  float* gamma = alpha + n;
  const size_t rowbytes = sizeof(float) * (size_t)V;
  const bool in_lds = rowbytes <= 144 * 1024;
  static unsigned long long attr = 0;
  if (asr_first_use_on_device(attr)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_rows_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_grad_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  }
  ASR_CHECK(sizeof(float) * 2 * S <= 64 * 1024, ASR_ERR_SHAPE, "asr_ctc_loss: label length %d too long", L);
  if (in_lds) hipLaunchKernelGGL(ctc_rows_kernel<true>, dim3((unsigned)(B * T)), dim3(256), rowbytes, st, (const float*)logits, ld, labels, T, V, L, blank, lse, lp);
  else hipLaunchKernelGGL(ctc_rows_kernel<false>, dim3((unsigned)(B * T)), dim3(256), 0, st, (const float*)logits, ld, labels, T, V, L, blank, lse, lp);
  hipLaunchKernelGGL(ctc_lattice_kernel, dim3((unsigned)B), dim3(256), sizeof(float) * 2 * S, st, (const float*)lp, alpha, gamma, labels, T, L, pad,
                     per_sample, stats, 1.f / (float)B);
  if (write_grad) {
    const float scale = grad_scale / (float)B;
    if (in_lds) hipLaunchKernelGGL(ctc_grad_kernel<true>, dim3((unsigned)(B * T)), dim3(256), rowbytes, st, logits, ld, labels, (const float*)lse, (const float*)gamma, T, V, L, blank, pad, scale);
    else hipLaunchKernelGGL(ctc_grad_kernel<false>, dim3((unsigned)(B * T)), dim3(256), 0, st, logits, ld, labels, (const float*)lse, (const float*)gamma, T, V, L, blank, pad, scale);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// out[r, :] = x[r, :] * (mask[r] != 0)   (deepspeech2.py:176 `* mask[:, :, None]`, and its gradient)
__global__ void mask_rows_kernel(const float* x, long ldx, const uint8_t* mask, int R, int C, float* out, long ldo) {
  const long n = (long)R * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / C), c = (int)(i % C);
    out[(long)r * ldo + c] = mask[r] ? x[(long)r * ldx + c] : 0.f;
  }
}
extern "C" int asr_mask_rows(const float* x, long ldx, const uint8_t* mask, int R, int C, float* out, long ldo, void* stream) {
  ASR_CHECK(x && mask && out && R > 0 && C > 0, ASR_ERR_ARG, "asr_mask_rows: bad argument");
  const long n = (long)R * C;
  hipLaunchKernelGGL(mask_rows_kernel, dim3((unsigned)min((long)2048, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, mask, R, C, out, ldo);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
