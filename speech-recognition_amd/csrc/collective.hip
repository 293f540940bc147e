// asr_allreduce_bucket: the data-parallel gradient exchange of run/train.py (tf.distribute.MirroredStrategy's implicit all-reduce,
// utils.py:142-153) as an entry point of this library, issued straight into RCCL on the caller's stream: no Python between the
// backward segment that completes a gradient bucket and the collective that sums it over the replicas, and - RCCL collectives being
// capturable - the whole data-parallel step can be ONE hipGraph (backward segments on the compute stream, bucket all-reduces forked
// onto the communication stream inside the capture).
//
// RCCL is resolved at run time (dlopen): the library must keep loading on a box without RCCL, and in a PyTorch process it must bind
// to the copy PyTorch has already loaded (soname librccl.so.1) rather than bring a second one.  One process per GPU; the unique id
// travels by whatever the host side has (torch.distributed's store in training.py, an MPI broadcast, a file).
//
// Wire format: f32, or - under mixed precision (the las_large configuration) - bf16: the bucket is rounded into a
// caller-provided bf16 staging buffer, summed as bf16 by RCCL (half the xGMI bytes) and written back to the f32 bucket.
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include "common.h"

typedef struct { char internal[128]; } asr_nccl_id;                 // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_id)(asr_nccl_id*);
typedef int (*fn_init_rank)(void**, int, asr_nccl_id, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);

static struct {
  void* h;
  fn_get_id get_id; fn_init_rank init_rank; fn_destroy destroy; fn_allreduce allreduce; fn_errstr errstr;
  int tried;
} g_rccl;

// ncclDataType_t / ncclRedOp_t values of rccl.h (stable ABI since NCCL 2.10): ncclFloat32 = 7, ncclBfloat16 = 9, ncclSum = 0
enum { ASR_NCCL_F32 = 7, ASR_NCCL_BF16 = 9, ASR_NCCL_SUM = 0 };

static bool rccl_load() {
  if (g_rccl.tried) return g_rccl.h != nullptr;
  g_rccl.tried = 1;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (h) break; }   // a copy the process already holds
  for (const char* n : names) { if (h) break; h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
  if (!h) return false;
  g_rccl.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
  g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
  g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  g_rccl.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
  g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  if (!g_rccl.get_id || !g_rccl.init_rank || !g_rccl.destroy || !g_rccl.allreduce) return false;
  g_rccl.h = h;
  return true;
}

#define RCCL_CHECK(call, what)                                                                       \
  do {                                                                                               \
    const int r__ = (call);                                                                          \
    if (r__ != 0) {                                                                                  \
      asr_set_error("%s: RCCL error %d (%s)", what, r__, g_rccl.errstr ? g_rccl.errstr(r__) : "?");  \
      return ASR_ERR_HIP;                                                                            \
    }                                                                                                \
  } while (0)

extern "C" int asr_comm_available(void) { return rccl_load() ? 1 : 0; }

extern "C" int asr_comm_unique_id(void* id128) {
  ASR_CHECK(id128, ASR_ERR_ARG, "asr_comm_unique_id: null argument");
  ASR_CHECK(rccl_load(), ASR_ERR_UNSUPPORTED, "asr_comm_unique_id: librccl.so.1 cannot be loaded");
  RCCL_CHECK(g_rccl.get_id(static_cast<asr_nccl_id*>(id128)), "asr_comm_unique_id");
  return ASR_OK;
}

extern "C" int asr_comm_init(const void* id128, int nranks, int rank, void** comm) {
  ASR_CHECK(id128 && comm, ASR_ERR_ARG, "asr_comm_init: null argument");
  ASR_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, ASR_ERR_ARG, "asr_comm_init: rank %d of %d", rank, nranks);
  ASR_CHECK(rccl_load(), ASR_ERR_UNSUPPORTED, "asr_comm_init: librccl.so.1 cannot be loaded");
  asr_nccl_id id;
  memcpy(&id, id128, sizeof(id));
  *comm = nullptr;
  RCCL_CHECK(g_rccl.init_rank(comm, nranks, id, rank), "asr_comm_init");
  return ASR_OK;
}

extern "C" int asr_comm_destroy(void* comm) {
  if (!comm) return ASR_OK;
  ASR_CHECK(rccl_load(), ASR_ERR_UNSUPPORTED, "asr_comm_destroy: librccl.so.1 cannot be loaded");
  RCCL_CHECK(g_rccl.destroy(comm), "asr_comm_destroy");
  return ASR_OK;
}

__global__ __launch_bounds__(256) void bucket_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
    bf4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    reinterpret_cast<bf4*>(dst)[i] = o;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (__bf16)src[i];
}
__global__ __launch_bounds__(256) void bucket_from_bf16_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
    const bf4 v = reinterpret_cast<const bf4*>(src)[i];
    reinterpret_cast<float4*>(dst)[i] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (float)src[i];
}

// bucket[0, n) <- sum over the ranks of `comm` of bucket (in place), on `stream`.  wire_bf16: NULL = the values travel as f32;
// else a device buffer of n bf16 (16-byte aligned) through which they travel as bf16 (rounded to nearest even before the sum,
// which RCCL accumulates in bf16's range with f32 arithmetic per pair).  Asynchronous, capturable into a hipGraph.
extern "C" int asr_allreduce_bucket(void* comm, float* bucket, long n, void* wire_bf16, void* stream) {
  ASR_CHECK(comm && bucket, ASR_ERR_ARG, "asr_allreduce_bucket: null argument");
  ASR_CHECK(n > 0, ASR_ERR_SHAPE, "asr_allreduce_bucket: n must be > 0");
  ASR_CHECK(((uintptr_t)bucket & 15) == 0 && ((uintptr_t)wire_bf16 & 15) == 0, ASR_ERR_ARG, "asr_allreduce_bucket: buffers must be 16-byte aligned");
  ASR_CHECK(rccl_load(), ASR_ERR_UNSUPPORTED, "asr_allreduce_bucket: librccl.so.1 cannot be loaded");
  hipStream_t st = (hipStream_t)stream;
  if (!wire_bf16) {
    RCCL_CHECK(g_rccl.allreduce(bucket, bucket, (size_t)n, ASR_NCCL_F32, ASR_NCCL_SUM, comm, st), "asr_allreduce_bucket");
    return ASR_OK;
  }
  const long blocks = (n / 4 + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 2048 ? (blocks > 0 ? blocks : 1) : 2048);
  hipLaunchKernelGGL(bucket_to_bf16_kernel, dim3(grid), dim3(256), 0, st, bucket, static_cast<__bf16*>(wire_bf16), n);
  ASR_LAUNCH_CHECK();
  RCCL_CHECK(g_rccl.allreduce(wire_bf16, wire_bf16, (size_t)n, ASR_NCCL_BF16, ASR_NCCL_SUM, comm, st), "asr_allreduce_bucket");
  hipLaunchKernelGGL(bucket_from_bf16_kernel, dim3(grid), dim3(256), 0, st, static_cast<const __bf16*>(wire_bf16), bucket, n);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
