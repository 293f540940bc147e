// Shared device/host helpers for libasr_mi355x (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/asr_mi355x.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define ASR_WAVE 64

// ---------------------------------------------------------------- host-side status plumbing
void asr_set_error(const char* fmt, ...);
// (Every entry point validates its arguments with ASR_CHECK before it launches anything.  A pure argument check: it does NOT touch the
// runtime's sticky error - an asynchronous failure left by another library (torch, RCCL) must stay visible to whoever launched it.
// The one stale error this library knows of - hipErrorNoDevice from a probe made before the device was initialised - is dropped
// once, by asr_runtime_init(), which the binding calls right after loading.)
#define ASR_CHECK(cond, code, ...)          \
  do {                                      \
    if (!(cond)) {                          \
      asr_set_error(__VA_ARGS__);           \
      return (code);                        \
    }                                       \
  } while (0)
// The opt-in to more than 64 KiB of dynamic LDS (hipFuncSetAttribute) lives with each DEVICE's copy of the code object, so a call site sets
// it on the first use per device, not per process: `seen` is that call site's bit mask over device ordinals.  (A race sets it twice.)
static inline bool asr_first_use_on_device(unsigned long long& seen) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (seen & bit) return false;
  seen |= bit;
  return true;
}
#define ASR_LAUNCH_CHECK()                                              \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      asr_set_error("%s: HIP launch error: %s", __func__, hipGetErrorString(e__)); \
      return ASR_ERR_HIP;                                               \
    }                                                                   \
  } while (0)

static inline int asr_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Zero `bytes` (a multiple of 4) of device memory with an ordinary kernel on `st`.  Used instead of
// hipMemsetAsync wherever the call may be captured into a hipGraph: memset nodes were observed to run out
// of order with the kernel nodes that follow them when another process shares the GPU (stale accumulators
// surviving into the next replay), kernel nodes of one stream never are.
static __global__ void asr_zero_kernel(uint32_t* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t asr_zero_async(void* p, size_t bytes, hipStream_t st) {
  const size_t n = bytes / 4;
  if (n == 0) return hipSuccess;
  const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(asr_zero_kernel, dim3(grid), dim3(256), 0, st, static_cast<uint32_t*>(p), n);
  return hipGetLastError();
}

// ---------------------------------------------------------------- stateless RNG (spec: oracle/rng.py)
__host__ __device__ __forceinline__ uint32_t asr_fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
struct AsrRngKey { uint32_t k1, k2; };
__host__ __device__ __forceinline__ AsrRngKey asr_rng_key(uint32_t seed, uint32_t stream) {
  AsrRngKey k;
  k.k1 = asr_fmix32(seed ^ (stream * 0x9E3779B1u + 0x7F4A7C15u));
  k.k2 = asr_fmix32(k.k1 + 0x6A09E667u + stream);
  return k;
}
__host__ __device__ __forceinline__ uint32_t asr_rng_u32(AsrRngKey k, uint32_t idx) {
  return asr_fmix32(((idx ^ k.k1) * 0x9E3779B1u) + k.k2);
}
__host__ __device__ __forceinline__ uint32_t asr_drop_threshold(float rate) {
  return rate > 0.f ? (uint32_t)(long long)((double)rate * 4294967296.0) : 0u;
}
// inverted-dropout multiplier for element idx
__device__ __forceinline__ float asr_drop_mult(AsrRngKey k, uint32_t idx, uint32_t thresh, float scale) {
  return asr_rng_u32(k, idx) >= thresh ? scale : 0.f;
}
__host__ __device__ __forceinline__ int asr_uniform_int(AsrRngKey k, uint32_t idx, int n) {
  if (n <= 0) return 0;
  return (int)(((uint64_t)asr_rng_u32(k, idx) * (uint64_t)(uint32_t)n) >> 32);
}

// ---------------------------------------------------------------- wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block reductions for <= 16 waves; `red` must hold >= 16 floats of LDS; result broadcast to all threads
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}

// Gate non-linearities on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each, absolute
// error of the results ~1e-7).  The gate math of a recurrent step runs on ONE wave per workgroup and
// sits on the step's critical path, so the ~100-instruction ocml expf/tanhf forms cost ~1 us per step.
// Contraction is pinned off in these helpers: the step kernels and the one-launch sweeps inline them into different
// surroundings and must produce the same bits (the tests compare them with torch.equal).
__device__ __forceinline__ float fast_exp_(float x) {
#pragma clang fp contract(off)
  return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
}
__device__ __forceinline__ float sigmoidf_(float x) {
#pragma clang fp contract(off)
  return __builtin_amdgcn_rcpf(1.f + fast_exp_(-x));
}
__device__ __forceinline__ float tanhf_(float x) {
#pragma clang fp contract(off)
  const float t = fast_exp_(-2.f * fabsf(x));
  return copysignf((1.f - t) * __builtin_amdgcn_rcpf(1.f + t), x);
}

// Forward gate math of one (batch row, hidden unit) pair - Keras LSTM / GRU(reset_after) / SimpleRNN cells (las.py:10-17 via
// tf.keras.layers.*).  CELL: 0 LSTM, 1 GRU, 2 SimpleRNN.  pre[g]: input projection (+ bias) per gate,
// s[slot]: recurrent sums in the packed slot order (LSTM i,f,c~,o; GRU z, r, x-part of h~ (unused), recurrent part of h~),
// br: GRU recurrent bias, hpm: previous h times the recurrent-dropout multiplier (= hp without it), cp: previous c.
// Out: hn (new h before the step mask), c2 (new c before the step mask), sv[4] the activations saved for backward.
template <int CELL>
__device__ __forceinline__ void asr_cell_forward(const float* pre, const float* s, const float* br, float hpm, float cp, float& hn, float& c2,
                                                 float (&sv)[4]) {
#pragma clang fp contract(off)
  if constexpr (CELL == 0) {
    const float ig = sigmoidf_(pre[0] + s[0]), fg = sigmoidf_(pre[1] + s[1]);
    const float gg = tanhf_(pre[2] + s[2]), og = sigmoidf_(pre[3] + s[3]);
    c2 = fg * cp + ig * gg;
    hn = og * tanhf_(c2);
    sv[0] = ig; sv[1] = fg; sv[2] = gg; sv[3] = og;
  } else if constexpr (CELL == 1) {
    const float z = sigmoidf_(pre[0] + s[0] + br[0]);
    const float r = sigmoidf_(pre[1] + s[1] + br[1]);
    const float arh = s[3] + br[2];
    const float hh = tanhf_(pre[2] + s[2] + r * arh);
    hn = z * hpm + (1.f - z) * hh;
    c2 = 0.f;
    sv[0] = z; sv[1] = r; sv[2] = hh; sv[3] = arh;
  } else {
    hn = tanhf_(pre[0] + s[0]);
    c2 = 0.f;
    sv[0] = hn; sv[1] = 0.f; sv[2] = 0.f; sv[3] = 0.f;
  }
}
