// Greedy decoding kernels (search.py:23-81 LAS greedy search, search.py:223-252 DeepSpeech2 best path).
// Both are row reductions over the vocabulary (HBM-bound: each logit is read once, 16 bytes per lane when
// the row is aligned) followed by a few words of per-row state, so a whole decode runs without a host
// round trip per step.
#include "common.h"

// max, first arg-max (tf.math.top_k / arg-max tie rule: lowest index) and log-sum-exp of one row.
// `last` >= 0 names an index that competes as if it were stored after the end of the row (the CTC blank,
// which search.py:237-239 moves to the last class before the arg-max).
__device__ __forceinline__ void row_top1(const float* __restrict__ x, int V, int last, float* red, int* redi, float& mx_out, int& am_out,
                                         float& lse_out) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float mx = -INFINITY;
  int am = 0x7fffffff;
  for (int c = tid; c < V; c += 256) {
    const float v = x[c];
    const int key = (c == last) ? V : c;
    if (v > mx || (v == mx && key < am)) { mx = v; am = key; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oi = __shfl_xor(am, o, 64);
    if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
  }
  if (lane == 0) { red[w] = mx; redi[w] = am; }
  __syncthreads();
  mx = red[0]; am = redi[0];
  for (int i = 1; i < 4; ++i)
    if (red[i] > mx || (red[i] == mx && redi[i] < am)) { mx = red[i]; am = redi[i]; }
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < V; c += 256) s += expf(x[c] - mx);
  s = block_sum(s, red);
  mx_out = mx;
  am_out = am;
  lse_out = mx + logf(s);
}

// One LAS greedy step (search.py:41-55) for row b = blockIdx.x.
__global__ __launch_bounds__(256) void greedy_update_kernel(const float* logits, long ld, int V, int cur_len, int eos, int pad,
                                                            int32_t* next_tok, uint8_t* ended, float* log_ppl, int32_t* seq_len) {
  __shared__ float red[8];
  __shared__ int redi[4];
  const int b = blockIdx.x;
  float mx, lse;
  int am;
  row_top1(logits + (long)b * ld, V, -1, red, redi, mx, am, lse);
  if (threadIdx.x == 0) {
    const bool was_ended = ended[b] != 0;
    const float lp = mx - lse;                                   // top-1 of log_softmax
    if (!was_ended) log_ppl[b] += lp;
    const int tok = was_ended ? pad : am;
    if (tok == eos) { ended[b] = 1; seq_len[b] = cur_len + 1; }
    next_tok[b] = tok;
  }
}

extern "C" int asr_greedy_update(const float* logits, long ld, int B, int V, int cur_len, int eos, int pad, int32_t* next_tok,
                                 uint8_t* ended, float* log_ppl, int32_t* seq_len, void* stream) {
  ASR_CHECK(logits && next_tok && ended && log_ppl && seq_len, ASR_ERR_ARG, "asr_greedy_update: null argument");
  ASR_CHECK(B > 0 && V > 0 && ld >= V, ASR_ERR_SHAPE, "asr_greedy_update: bad shape B=%d V=%d ld=%ld", B, V, ld);
  hipLaunchKernelGGL(greedy_update_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logits, ld, V, cur_len, eos, pad, next_tok,
                     ended, log_ppl, seq_len);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// DeepSpeech2 best path, stage 1: per frame the winning class (blank competing as the last class) and its
// log-probability.
__global__ __launch_bounds__(256) void ctc_best_rows_kernel(const float* logits, long ld, int V, int blank, int32_t* best, float* best_lp) {
  __shared__ float red[8];
  __shared__ int redi[4];
  const long r = blockIdx.x;
  float mx, lse;
  int am;
  row_top1(logits + r * ld, V, blank, red, redi, mx, am, lse);
  if (threadIdx.x == 0) {
    best[r] = am;                                                // V means blank
    best_lp[r] = mx - lse;
  }
}

// stage 2: collapse repeats, drop blanks ([TF-sem] tf.nn.ctc_greedy_decoder, merge_repeated=True), one wave per
// utterance: lanes take 64 frames at a time, a ballot marks the emitting frames, popcount gives each its slot.
__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int32_t* best, const float* best_lp, int T, int V, int32_t* tokens,
                                                          int32_t* lengths, float* neg_sum) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int32_t* row = best + (long)b * T;
  int32_t* out = tokens + (long)b * T;
  int count = 0;
  float acc = 0.f;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    bool emit = false;
    int cls = V;
    if (t < T) {
      cls = row[t];
      const int prev = t > 0 ? row[t - 1] : -1;
      emit = cls != V && cls != prev;
      acc += best_lp[(long)b * T + t];
    }
    const unsigned long long m = __ballot(emit);
    if (emit) out[count + __popcll(m & ((1ull << lane) - 1))] = cls;
    count += __popcll(m);
  }
  for (int t = count + lane; t < T; t += 64) out[t] = 0;          // tf.sparse.to_dense pads with 0
  acc = wave_sum(acc);
  if (lane == 0) { lengths[b] = count; neg_sum[b] = -acc; }
}

extern "C" int asr_ctc_greedy(const float* logits, long ld, int B, int T, int V, int blank, int32_t* best, float* best_lp,
                              int32_t* tokens, int32_t* lengths, float* neg_sum_logits, void* stream) {
  ASR_CHECK(logits && best && best_lp && tokens && lengths && neg_sum_logits, ASR_ERR_ARG, "asr_ctc_greedy: null argument");
  ASR_CHECK(B > 0 && T > 0 && V > 0 && ld >= V && blank >= 0 && blank < V, ASR_ERR_SHAPE, "asr_ctc_greedy: bad shape B=%d T=%d V=%d blank=%d", B, T, V, blank);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ctc_best_rows_kernel, dim3((unsigned)((long)B * T)), dim3(256), 0, st, logits, ld, V, blank, best, best_lp);
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3((unsigned)B), dim3(64), 0, st, (const int32_t*)best, (const float*)best_lp, T, V, tokens, lengths,
                     neg_sum_logits);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// LAS beam search (search.py:83-209).  Per decoding step two launches:
//   beam_topk_kernel    one workgroup per hypothesis row: log_softmax of the row and its k best classes in
//                       tf.math.top_k order (descending, ties -> lowest index), k row scans out of L2;
//   beam_select_kernel  one workgroup per utterance: the beam*k candidates' accumulated log-probabilities
//                       (search.py:136-138), length penalty (search.py:159-160), the stable top-beam of
//                       log_prob * penalty (search.py:162) and the gather of the token histories
//                       (search.py:165-175) from the `in` to the `out` half of a ping-pong state.
// The reference leaves its loop when every row holds an EOS (search.py:122-125): each workgroup re-derives
// that from the `in` flags and then only forwards the state, so the host may look at the flags rarely.
struct BeamState {
  int32_t* hist;      // [R, ldh] token histories
  float* log_ppl;     // [R]
  uint8_t* ended;     // [R] row holds an EOS
  int32_t* slen;      // [R] first EOS index + 1 (valid when ended)
};

__global__ __launch_bounds__(256) void beam_topk_kernel(const float* logits, long ld, int V, int k, float* lp_out, int32_t* tok_out) {
  __shared__ float red[8];
  __shared__ int redi[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* x = logits + (long)r * ld;
  float mx, lse;
  int am;
  row_top1(x, V, -1, red, redi, mx, am, lse);
  const float logs = lse - mx;                                   // log(sum(exp(x - max)))
  float pv = mx;
  int pi = am;
  if (tid == 0) { lp_out[(long)r * k] = (mx - mx) - logs; tok_out[(long)r * k] = am; }
  for (int m = 1; m < k; ++m) {
    // best element strictly after (pv, pi) in (value descending, index ascending) order
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = tid; c < V; c += 256) {
      const float v = x[c];
      const bool after = v < pv || (v == pv && c > pi);
      if (after && (v > bv || (v == bv && c < bi))) { bv = v; bi = c; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { red[w] = bv; redi[w] = bi; }
    __syncthreads();
    bv = red[0]; bi = redi[0];
    for (int i = 1; i < 4; ++i)
      if (redi[i] != 0x7fffffff && (bi == 0x7fffffff || red[i] > bv || (red[i] == bv && redi[i] < bi))) { bv = red[i]; bi = redi[i]; }
    pv = bv; pi = bi;
    if (tid == 0) {
      const bool have = bi != 0x7fffffff;                        // k > V: repeat the worst class
      lp_out[(long)r * k + m] = have ? (bv - mx) - logs : -INFINITY;
      tok_out[(long)r * k + m] = have ? bi : 0;
    }
  }
}

#define BEAM_MAX 32

__global__ __launch_bounds__(256) void beam_select_kernel(const float* lp, const int32_t* tok, BeamState in, BeamState out, int ldh, int R,
                                                          int beam, int cur_len, int eos, double alpha, double beta, int32_t* next_tok,
                                                          int32_t* parent, int32_t* final_len) {
  __shared__ float s_score[BEAM_MAX * BEAM_MAX];
  __shared__ float s_lp[BEAM_MAX * BEAM_MAX];
  __shared__ uint8_t s_taken[BEAM_MAX * BEAM_MAX];
  __shared__ int s_pick[BEAM_MAX];
  __shared__ float red[8];
  __shared__ int redi[4];
  __shared__ int s_live;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int row0 = b * beam;
  const bool first = cur_len == 1;                               // search.py:141: the rows are still one per utterance
  if (tid == 0) s_live = 0;
  __syncthreads();
  {
    int live = 0;
    for (int r = tid; r < R; r += 256)
      if (!in.ended[r] && (!first || r % beam == 0)) live = 1;
    if (live) s_live = 1;
  }
  __syncthreads();
  if (!s_live) {                                                 // search.py:122-125: the loop is over, carry the state
    for (int i = tid; i < beam * ldh; i += 256) out.hist[(long)row0 * ldh + i] = in.hist[(long)row0 * ldh + i];
    if (tid < beam) {
      out.log_ppl[row0 + tid] = in.log_ppl[row0 + tid];
      out.ended[row0 + tid] = in.ended[row0 + tid];
      out.slen[row0 + tid] = in.slen[row0 + tid];
      next_tok[row0 + tid] = in.hist[(long)(row0 + tid) * ldh + cur_len - 1];
      parent[row0 + tid] = row0 + tid;
    }
    return;
  }
  if (b == 0 && tid == 0) *final_len = cur_len + 1;
  const int C = first ? beam : beam * beam;
  for (int c = tid; c < C; c += 256) {
    const int j = first ? 0 : c / beam, m = c % beam;
    const int r = row0 + j;
    const bool e = in.ended[r] != 0;
    const float l = (e ? 0.f : lp[(long)r * beam + m]) + in.log_ppl[r];          // search.py:137-138
    const int len = e ? in.slen[r] : cur_len + 1;
    const float pen = (float)pow((double)(1 + len) / (1.0 + beta), alpha);        // search.py:159 (float64 in TF)
    s_lp[c] = l;
    s_score[c] = first ? -(float)c : l * pen;                    // first step: top_k order itself (search.py:141-153)
    s_taken[c] = 0;
  }
  __syncthreads();
  for (int q = 0; q < beam; ++q) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = tid; c < C; c += 256) {
      if (s_taken[c]) continue;
      const float v = s_score[c];
      if (bi == 0x7fffffff || v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    if (lane == 0) { red[w] = bv; redi[w] = bi; }
    __syncthreads();
    if (tid == 0) {
      bv = red[0]; bi = redi[0];
      for (int i = 1; i < 4; ++i)
        if (redi[i] != 0x7fffffff && (bi == 0x7fffffff || red[i] > bv || (red[i] == bv && redi[i] < bi))) { bv = red[i]; bi = redi[i]; }
      s_pick[q] = bi;
      s_taken[bi] = 1;
    }
    __syncthreads();
  }
  for (int i = tid; i < beam * cur_len; i += 256) {              // search.py:165-175: gather the parents' histories
    const int q = i / cur_len, t = i % cur_len;
    const int c = s_pick[q];
    const int j = first ? 0 : c / beam;
    out.hist[(long)(row0 + q) * ldh + t] = in.hist[(long)(row0 + j) * ldh + t];
  }
  if (tid < beam) {
    const int q = tid, c = s_pick[q];
    const int j = first ? 0 : c / beam, m = c % beam;
    const int r = row0 + j;
    const bool e = in.ended[r] != 0;
    const int t = tok[(long)r * beam + m];
    out.hist[(long)(row0 + q) * ldh + cur_len] = t;
    out.log_ppl[row0 + q] = s_lp[c];
    out.ended[row0 + q] = e || t == eos;
    out.slen[row0 + q] = e ? in.slen[r] : cur_len + 1;
    next_tok[row0 + q] = t;
    parent[row0 + q] = r;
  }
}

extern "C" int asr_beam_topk(const float* logits, long ld, int R, int V, int k, float* lp, int32_t* tok, void* stream) {
  ASR_CHECK(logits && lp && tok, ASR_ERR_ARG, "asr_beam_topk: null argument");
  ASR_CHECK(R > 0 && V > 0 && ld >= V && k > 0 && k <= BEAM_MAX, ASR_ERR_SHAPE, "asr_beam_topk: bad shape R=%d V=%d k=%d (k <= %d)", R, V, k, BEAM_MAX);
  hipLaunchKernelGGL(beam_topk_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, ld, V, k, lp, tok);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_beam_select(const float* lp, const int32_t* tok, int B, int beam, int cur_len, int ld_hist, int eos, double alpha,
                               double beta, const int32_t* hist_in, const float* ppl_in, const uint8_t* ended_in, const int32_t* slen_in,
                               int32_t* hist_out, float* ppl_out, uint8_t* ended_out, int32_t* slen_out, int32_t* next_tok, int32_t* parent,
                               int32_t* final_len, void* stream) {
  ASR_CHECK(lp && tok && hist_in && ppl_in && ended_in && slen_in && hist_out && ppl_out && ended_out && slen_out && next_tok && parent && final_len,
            ASR_ERR_ARG, "asr_beam_select: null argument");
  ASR_CHECK(B > 0 && beam > 0 && beam <= BEAM_MAX && cur_len >= 1 && ld_hist > cur_len, ASR_ERR_SHAPE,
            "asr_beam_select: bad shape B=%d beam=%d (<= %d) cur_len=%d ld_hist=%d", B, beam, BEAM_MAX, cur_len, ld_hist);
  ASR_CHECK(hist_in != hist_out && ppl_in != ppl_out && ended_in != ended_out, ASR_ERR_ARG, "asr_beam_select: in and out state must differ");
  BeamState in{const_cast<int32_t*>(hist_in), const_cast<float*>(ppl_in), const_cast<uint8_t*>(ended_in), const_cast<int32_t*>(slen_in)};
  BeamState out{hist_out, ppl_out, ended_out, slen_out};
  hipLaunchKernelGGL(beam_select_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, lp, tok, in, out, ld_hist, B * beam, beam, cur_len,
                     eos, alpha, beta, next_tok, parent, final_len);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// DeepSpeechSearcher.beam_search, device half (search.py:268-272): out[r, :] = log_softmax over V + 1 classes of
// the row with the blank appended as the last class and its old slot masked by -1e9.
__global__ __launch_bounds__(256) void ctc_log_softmax_kernel(const float* logits, long ld, int V, int blank, float* out) {
  __shared__ float red[8];
  const long r = blockIdx.x;
  const int tid = threadIdx.x;
  const float* x = logits + r * ld;
  float* o = out + r * (long)(V + 1);
  float mx = -INFINITY;
  for (int c = tid; c <= V; c += 256) {
    const float v = c == V ? x[blank] : (c == blank ? x[c] + -1e9f : x[c]);
    mx = fmaxf(mx, v);
  }
  mx = block_max(mx, red);
  float s = 0.f;
  for (int c = tid; c <= V; c += 256) {
    const float v = c == V ? x[blank] : (c == blank ? x[c] + -1e9f : x[c]);
    s += expf(v - mx);
  }
  s = block_sum(s, red);
  const float logs = logf(s);
  for (int c = tid; c <= V; c += 256) {
    const float v = c == V ? x[blank] : (c == blank ? x[c] + -1e9f : x[c]);
    o[c] = (v - mx) - logs;
  }
}

extern "C" int asr_ctc_log_softmax(const float* logits, long ld, long R, int V, int blank, float* out, void* stream) {
  ASR_CHECK(logits && out, ASR_ERR_ARG, "asr_ctc_log_softmax: null argument");
  ASR_CHECK(R > 0 && V > 0 && ld >= V && blank >= 0 && blank < V, ASR_ERR_SHAPE, "asr_ctc_log_softmax: bad shape R=%ld V=%d blank=%d", R, V, blank);
  hipLaunchKernelGGL(ctc_log_softmax_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, ld, V, blank, out);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
