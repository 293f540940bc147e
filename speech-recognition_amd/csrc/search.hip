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
