// Hand-off latency between two workgroups on DIFFERENT XCDs under every cache-policy combination of the store and the polling load
// (sc0 / sc1 / nt bits), with the message size of the sweeps' gathers (NL 16-byte loads per lane, 64 lanes).
// hipcc --offload-arch=gfx950 -O2 -o /tmp/pingpong2 tests/tools/micro/pingpong2.hip && /tmp/pingpong2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NL 4

template <int SM>
__device__ __forceinline__ void put(float* p, f32x4 v) {
  if (SM == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (SM == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if (SM == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int LM>
__device__ __forceinline__ f32x4 get(const float* p) {
  f32x4 v;
  if (LM == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  else if (LM == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  else if (LM == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int SM, int LM>
__global__ void __launch_bounds__(64) pingpong(float* buf, int peer, int rounds, int* xcc, long long* cycles) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (lane == 0) xcc[b] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7;
  if (b != 0 && b != peer) return;
  float* a = buf + lane * 4;
  float* c = buf + 65536 + lane * 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  bool dead = false;
  for (int i = 1; i <= rounds && !dead; ++i) {
    const float tag = (float)i;
    float* mine = b == 0 ? a : c;
    const float* theirs = b == 0 ? c : a;
    if (b != 0) {
      int spins = 0;
      for (;;) {
        f32x4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) v[k] = get<LM>(theirs + k * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NL; ++k) ok = ok && v[k].x == tag && v[k].w == tag;
        if (__all(ok) || (dead = ++spins > (1 << 16))) break;
      }
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) put<SM>(mine + k * 1024, (f32x4){tag, tag, tag, tag});
    if (b == 0) {
      int spins = 0;
      for (;;) {
        f32x4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) v[k] = get<LM>(theirs + k * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NL; ++k) ok = ok && v[k].x == tag && v[k].w == tag;
        if (__all(ok) || (dead = ++spins > (1 << 16))) break;
      }
    }
  }
  if (b == 0 && lane == 0) *cycles = dead ? -1 : (long long)(__builtin_amdgcn_s_memrealtime() - t0);
}

template <int SM, int LM>
static void run(float* buf, int* xcc, long long* cyc, int peer) {
  const int rounds = 2000;
  int hx[64];
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(buf, 0, 1 << 20);
    pingpong<SM, LM><<<64, 64>>>(buf, peer, rounds, xcc, cyc);
    hipDeviceSynchronize();
    hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
    long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    static const char* sn[] = {"sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt"};
    if (rep) {
      if (hc < 0) printf("store %-10s load %-10s peer %2d xcc %d/%d: TIMED OUT (stale read)\n", sn[SM], sn[LM], peer, hx[0], hx[peer]);
      else printf("store %-10s load %-10s peer %2d xcc %d/%d: %.0f ns per one-way hand-off (%d x 1 KB)\n", sn[SM], sn[LM], peer, hx[0], hx[peer], hc * 10.0 / rounds / 2, NL);
    }
  }
}

int main() {
  float* buf; int* xcc; long long* cyc;
  hipMalloc(&buf, 1 << 20); hipMalloc(&xcc, 4096); hipMalloc(&cyc, 8);
  for (int peer : {1, 4, 8}) {
    run<0, 0>(buf, xcc, cyc, peer); run<0, 1>(buf, xcc, cyc, peer); run<0, 2>(buf, xcc, cyc, peer); run<0, 3>(buf, xcc, cyc, peer);
    run<1, 0>(buf, xcc, cyc, peer); run<1, 1>(buf, xcc, cyc, peer); run<1, 2>(buf, xcc, cyc, peer); run<1, 3>(buf, xcc, cyc, peer);
    run<2, 0>(buf, xcc, cyc, peer); run<2, 1>(buf, xcc, cyc, peer); run<2, 2>(buf, xcc, cyc, peer); run<2, 3>(buf, xcc, cyc, peer);
    run<3, 0>(buf, xcc, cyc, peer); run<3, 1>(buf, xcc, cyc, peer); run<3, 2>(buf, xcc, cyc, peer); run<3, 3>(buf, xcc, cyc, peer);
  }
  return 0;
}
