// Masked sparse softmax cross-entropy + accuracy for gfx950, forward and gradient in one pass over the
// logits (measure.py:4-21 SparseCategoricalCrossentropy, measure.py:45-69 SparseCategoricalAccuracy, and
// the Keras SUM_OVER_BATCH_SIZE mean over the kept tokens).  HBM-bound: each logits row is read once
// into LDS, reduced with wave shuffles, and the gradient is written once in place.  (CTC: ctc.hip.)
#include "common.h"

// stats layout (float): [0] sum of per-token NLL / n_valid (= the loss), [1] number of correct argmax, [2] n_valid
// One workgroup per row r = (b, u).  Row cached in LDS when it fits (V <= 36864), otherwise re-read.
template <bool IN_LDS>
__global__ __launch_bounds__(256) void softmax_xent_kernel(float* logits, long ld, const int32_t* labels, int R, int V, int ignore_index,
                                                           float* stats, int write_grad, float grad_scale) {
  extern __shared__ float row[];
  __shared__ float red[16];
  __shared__ int redi[4];
  __shared__ float redv[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float* x = logits + (long)r * ld;
  // number of kept tokens (every block recomputes it: R ints, L2 resident)
  float cnt = 0.f;
  for (int i = tid; i < R; i += 256) cnt += (labels[i] != ignore_index) ? 1.f : 0.f;
  cnt = block_sum(cnt, red);
  const int y = labels[r];
  const bool keep = (y != ignore_index);
  if (!keep) {
    if (write_grad)
      for (int c = tid; c < V; c += 256) x[c] = 0.f;
    return;
  }
  float mx = -INFINITY;
  int am = 0x7fffffff;
  for (int c = tid; c < V; c += 256) {
    const float v = x[c];
    if (IN_LDS) row[c] = v;
    if (v > mx) { mx = v; am = c; }
  }
  // argmax with lowest-index tie break (tf.argmax) and the row max
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oi = __shfl_xor(am, o, 64);
    if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
  }
  if (lane == 0) { redv[w] = mx; redi[w] = am; }
  __syncthreads();
  mx = redv[0]; am = redi[0];
  for (int i = 1; i < 4; ++i)
    if (redv[i] > mx || (redv[i] == mx && redi[i] < am)) { mx = redv[i]; am = redi[i]; }
  float s = 0.f;
  for (int c = tid; c < V; c += 256) s += expf((IN_LDS ? row[c] : x[c]) - mx);
  s = block_sum(s, red);
  const float lse = mx + logf(s);
  const float inv_cnt = cnt > 0.f ? 1.f / cnt : 0.f;
  if (tid == 0) {
    // a label outside [0, V) never indexes memory: NaN loss, as TF's GPU kernel reports it
    const float xy = (y >= 0 && y < V) ? (IN_LDS ? row[y] : x[y]) : NAN;
    atomicAdd(&stats[0], (lse - xy) * inv_cnt);
    atomicAdd(&stats[1], am == y ? 1.f : 0.f);
    if (r == 0) stats[2] = cnt;
  }
  if (write_grad) {
    const float g = grad_scale * inv_cnt;
    for (int c = tid; c < V; c += 256) {
      const float pz = expf((IN_LDS ? row[c] : x[c]) - lse);
      x[c] = (pz - (c == y ? 1.f : 0.f)) * g;
    }
  }
}

// The same row pass for the common layout (V and the row stride multiples of 4, 16-byte aligned rows, row fits in LDS): 16-byte loads
// and stores, four of them in flight per thread (the scalar kernel above keeps one 4-byte load per thread in flight and reaches
// 1.7 TB/s on the [2048 x 16000] logits of a las_small step), and ONE exponential per element - the second pass leaves exp(x - max) in
// LDS and the gradient pass scales it by 1 / sum.
__global__ __launch_bounds__(256) void softmax_xent_vec_kernel(float* logits, long ld, const int32_t* labels, int R, int V, int ignore_index,
                                                               float* stats, int write_grad, float grad_scale) {
  extern __shared__ float row[];
  __shared__ float red[16];
  __shared__ int redi[4];
  __shared__ float redv[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float4* x4 = reinterpret_cast<float4*>(logits + (long)r * ld);
  float4* row4 = reinterpret_cast<float4*>(row);
  const int V4 = V >> 2;
  float cnt = 0.f;
  for (int i = tid; i < R; i += 256) cnt += (labels[i] != ignore_index) ? 1.f : 0.f;
  cnt = block_sum(cnt, red);
  const int y = labels[r];
  if (y == ignore_index) {
    if (write_grad)
      for (int c = tid; c < V4; c += 256) x4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const bool y_ok = y >= 0 && y < V;
  const float xy = (tid == 0 && y_ok) ? logits[(long)r * ld + y] : 0.f;     // (before the row is overwritten by its gradient)
  float mx = -INFINITY;
  int am = 0x7fffffff;
  auto upd = [&](float v, int c) { if (v > mx || (v == mx && c < am)) { mx = v; am = c; } };
  int c = tid;
  for (; c + 768 < V4; c += 1024) {
    const float4 a = x4[c], b = x4[c + 256], d = x4[c + 512], e = x4[c + 768];
    row4[c] = a; row4[c + 256] = b; row4[c + 512] = d; row4[c + 768] = e;
    upd(a.x, 4 * c); upd(a.y, 4 * c + 1); upd(a.z, 4 * c + 2); upd(a.w, 4 * c + 3);
    upd(b.x, 4 * (c + 256)); upd(b.y, 4 * (c + 256) + 1); upd(b.z, 4 * (c + 256) + 2); upd(b.w, 4 * (c + 256) + 3);
    upd(d.x, 4 * (c + 512)); upd(d.y, 4 * (c + 512) + 1); upd(d.z, 4 * (c + 512) + 2); upd(d.w, 4 * (c + 512) + 3);
    upd(e.x, 4 * (c + 768)); upd(e.y, 4 * (c + 768) + 1); upd(e.z, 4 * (c + 768) + 2); upd(e.w, 4 * (c + 768) + 3);
  }
  for (; c < V4; c += 256) {
    const float4 a = x4[c];
    row4[c] = a;
    upd(a.x, 4 * c); upd(a.y, 4 * c + 1); upd(a.z, 4 * c + 2); upd(a.w, 4 * c + 3);
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oi = __shfl_xor(am, o, 64);
    if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
  }
  if (lane == 0) { redv[w] = mx; redi[w] = am; }
  __syncthreads();
  mx = redv[0]; am = redi[0];
  for (int i = 1; i < 4; ++i)
    if (redv[i] > mx || (redv[i] == mx && redi[i] < am)) { mx = redv[i]; am = redi[i]; }
  float s = 0.f;
  for (int k = tid; k < V4; k += 256) {                    // (each thread revisits the granules it wrote: no barrier needed)
    float4 v = row4[k];
    v.x = expf(v.x - mx); v.y = expf(v.y - mx); v.z = expf(v.z - mx); v.w = expf(v.w - mx);
    row4[k] = v;
    s += (v.x + v.y) + (v.z + v.w);
  }
  s = block_sum(s, red);
  const float inv_cnt = cnt > 0.f ? 1.f / cnt : 0.f;
  if (tid == 0) {
    atomicAdd(&stats[0], (y_ok ? (mx + logf(s)) - xy : NAN) * inv_cnt);
    atomicAdd(&stats[1], am == y ? 1.f : 0.f);
    if (r == 0) stats[2] = cnt;
  }
  if (write_grad) {
    const float g = grad_scale * inv_cnt, gs = g / s;
    const int y4 = y_ok ? (y >> 2) : -1, ye = y & 3;
    for (int k = tid; k < V4; k += 256) {
      float4 v = row4[k];
      v.x *= gs; v.y *= gs; v.z *= gs; v.w *= gs;
      if (k == y4) { if (ye == 0) v.x -= g; else if (ye == 1) v.y -= g; else if (ye == 2) v.z -= g; else v.w -= g; }
      x4[k] = v;
    }
  }
}

// logits [R, V] (row stride ld) are overwritten by d loss / d logits when write_grad != 0.
// stats (3 floats, device) must be zeroed by the caller before the call.
extern "C" int asr_softmax_xent(float* logits, long ld, const int32_t* labels, int R, int V, int ignore_index, float* stats, int write_grad,
                                float grad_scale, void* stream) {
  ASR_CHECK(logits && labels && stats && R > 0 && V > 0 && ld >= V, ASR_ERR_ARG, "asr_softmax_xent: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t bytes = sizeof(float) * (size_t)V;
  if (bytes <= 144 * 1024 && V % 4 == 0 && ld % 4 == 0 && ((uintptr_t)logits & 15) == 0) {
    static unsigned long long attr_v = 0;
    if (asr_first_use_on_device(attr_v)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_xent_vec_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    }
    hipLaunchKernelGGL(softmax_xent_vec_kernel, dim3((unsigned)R), dim3(256), bytes, st, logits, ld, labels, R, V, ignore_index, stats, write_grad,
                       grad_scale);
  } else if (bytes <= 144 * 1024) {
    static unsigned long long attr = 0;
    if (asr_first_use_on_device(attr)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_xent_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    }
    hipLaunchKernelGGL(softmax_xent_kernel<true>, dim3((unsigned)R), dim3(256), bytes, st, logits, ld, labels, R, V, ignore_index, stats,
                       write_grad, grad_scale);
  } else {
    hipLaunchKernelGGL(softmax_xent_kernel<false>, dim3((unsigned)R), dim3(256), 0, st, logits, ld, labels, R, V, ignore_index, stats, write_grad,
                       grad_scale);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
