// Optimizer for gfx950: Keras Adam (run/train.py:159-168) fused with the reference's LRScheduler
// (utils.py:11-35) over ONE flat parameter buffer (parameters, gradients and both moments are
// contiguous fp32 arrays of equal length, so the step is a single HBM-bound pass: 4 reads + 3 writes
// per element).  Step count and dropout seed live in device memory so that a captured hipGraph
// advances them on every replay.
#include "common.h"

// state (device, int32 x 4): [0] iterations (number of steps applied), [1] dropout seed, [2] sticky error word (set by
// advance_state when the step's skip flag was raised; cleared only by the host), [3] reserved
struct AdamArgs {
  float* p; const float* g; float* m; float* v;
  long n;
  const int32_t* state;
  asr_lr_schedule lr;
  float beta1, beta2, eps, grad_scale;
  const float* skip;       // optional: a non-zero value means the step's gradients are invalid - leave everything untouched
};

__device__ __forceinline__ float lr_at(const asr_lr_schedule& s, float step) {
  // utils.py:28-35: min(step * inc, max_lr - (step - warmup) * dec) clipped below at min_lr
  const float st = step + (float)s.offset_steps;
  const float lr = fminf(st * s.increasing_delta, s.max_learning_rate - (st - (float)s.warmup_steps) * s.decreasing_delta);
  return fmaxf(lr, s.min_learning_rate);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  if (a.skip != nullptr && *a.skip != 0.f) return;
  const int it = a.state[0];
  const float t = (float)(it + 1);
  const float lr = lr_at(a.lr, (float)it);
  // [TF-sem] Keras Adam: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  theta -= lr_t * m / (sqrt(v) + eps)
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)a.beta2, (double)t)) / (1.0 - pow((double)a.beta1, (double)t)));
  const long n4 = a.n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  float4* p4 = reinterpret_cast<float4*>(a.p);
  const float4* g4 = reinterpret_cast<const float4*>(a.g);
  float4* m4 = reinterpret_cast<float4*>(a.m);
  float4* v4 = reinterpret_cast<float4*>(a.v);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 p = p4[i], g = g4[i], m = m4[i], v = v4[i];
#define ADAM1(c)                                              \
    {                                                         \
      const float gg = g.c * a.grad_scale;                    \
      m.c = a.beta1 * m.c + (1.f - a.beta1) * gg;             \
      v.c = a.beta2 * v.c + (1.f - a.beta2) * gg * gg;        \
      p.c -= lr_t * m.c / (sqrtf(v.c) + a.eps);               \
    }
    ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
    p4[i] = p; m4[i] = m; v4[i] = v;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const float gg = a.g[i] * a.grad_scale;
    const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * gg;
    const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * gg * gg;
    a.m[i] = m; a.v[i] = v;
    a.p[i] -= lr_t * m / (sqrtf(v) + a.eps);
  }
}

__global__ void advance_state_kernel(int32_t* state, const float* skip) {
  if (skip != nullptr && *skip != 0.f) { state[2] |= 1; return; }   // an invalid step is not counted and stays on record
  state[0] += 1;
  state[1] = (int32_t)asr_fmix32((uint32_t)state[1] + 0x9E3779B9u);
}

extern "C" int asr_adam_step(float* params, const float* grads, float* m, float* v, long n, const int32_t* state,
                             const asr_lr_schedule* lr, float beta1, float beta2, float eps, float grad_scale, const float* skip_flag,
                             void* stream) {
  ASR_CHECK(params && grads && m && v && state && lr && n > 0, ASR_ERR_ARG, "asr_adam_step: bad argument");
  ASR_CHECK((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) == 0, ASR_ERR_ARG,
            "asr_adam_step: buffers must be 16-byte aligned");
  AdamArgs a{params, grads, m, v, n, state, *lr, beta1, beta2, eps, grad_scale, skip_flag};
  const long blocks = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks))), dim3(256), 0, (hipStream_t)stream, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// iterations += 1; seed = fmix32(seed + golden)  -- run once at the end of every training step.  skip_flag (optional, device): a
// non-zero value (a hand-off of a one-launch sweep timed out somewhere in the step, on any rank: the flag travels inside the
// last gradient bucket) sets state[2] instead - sticky until the host clears it
extern "C" int asr_advance_state(int32_t* state, const float* skip_flag, void* stream) {
  ASR_CHECK(state, ASR_ERR_ARG, "asr_advance_state: null argument");
  hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, skip_flag);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// host-side mirror of utils.py:11-27 (schedule constants) so bindings need no arithmetic of their own
extern "C" int asr_lr_schedule_init(asr_lr_schedule* s, long total_steps, double max_learning_rate, double min_learning_rate,
                                    double warmup_rate, long warmup_steps, long offset_steps) {
  ASR_CHECK(s, ASR_ERR_ARG, "asr_lr_schedule_init: null argument");
  const long ws = warmup_steps ? warmup_steps : (long)(total_steps * warmup_rate) + 1;
  ASR_CHECK(total_steps != ws, ASR_ERR_ARG, "asr_lr_schedule_init: total_steps == warmup_steps (division by zero in the reference too)");
  s->warmup_steps = (int)ws;
  s->increasing_delta = ws ? (float)(max_learning_rate / ws) : 1e12f;
  s->decreasing_delta = (float)((max_learning_rate - min_learning_rate) / (double)(total_steps - ws));
  s->max_learning_rate = (float)max_learning_rate;
  s->min_learning_rate = (float)min_learning_rate;
  s->offset_steps = (int)offset_steps;
  return ASR_OK;
}
