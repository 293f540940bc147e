// Hand-off latency between two workgroups: same XCD vs different XCD, sc1 vs plain stores (loads always sc1: L1-bypassing).
// hipcc --offload-arch=gfx950 -O2 -o /tmp/pingpong tests/tools/micro/pingpong.hip && /tmp/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int STORE_SC1>
__device__ __forceinline__ void put(float* p, f32x4 v) {
  if (STORE_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int LOAD_PLAIN>
__device__ __forceinline__ f32x4 get(const float* p) {
  f32x4 v;
  if (LOAD_PLAIN) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// block 0 = ping, block `peer` = pong; each round: ping writes slot a (value i), pong waits for it and writes slot b, ping waits.
// Every lane of wave 0 moves its own 16 bytes (a 1 KB message per direction), like one piece of the sweeps' hand-offs.
template <int STORE_SC1, int LOAD_PLAIN>
__global__ void __launch_bounds__(64) pingpong(float* buf, int peer, int rounds, int* xcc, long long* cycles) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (lane == 0) xcc[b] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7;   // HW_REG_XCC_ID bits [3:0]
  if (b != 0 && b != peer) return;
  float* a = buf + lane * 4;
  float* c = buf + 4096 + lane * 4;
  long long t0 = __builtin_readcyclecounter();
  bool dead = false;
  for (int i = 1; i <= rounds && !dead; ++i) {
    const float tag = (float)i;
    if (b == 0) {
      put<STORE_SC1>(a, (f32x4){tag, tag, tag, tag});
      int spins = 0;
      for (;;) { f32x4 v = get<LOAD_PLAIN>(c); if (__all(v.x == tag && v.w == tag) || (dead = ++spins > (1 << 16))) break; }
    } else {
      int spins = 0;
      for (;;) { f32x4 v = get<LOAD_PLAIN>(a); if (__all(v.x == tag && v.w == tag) || (dead = ++spins > (1 << 16))) break; }
      put<STORE_SC1>(c, (f32x4){tag, tag, tag, tag});
    }
  }
  if (b == 0 && lane == 0) *cycles = dead ? -1 : (long long)(__builtin_readcyclecounter() - t0);
}

int main() {
  float* buf; int* xcc; long long* cyc;
  hipMalloc(&buf, 1 << 16); hipMalloc(&xcc, 4096); hipMalloc(&cyc, 8);
  const int rounds = 2000, grid = 64;
  int hx[64];
  for (int mode = 0; mode < 3; ++mode)
    for (int peer : {8, 16, 1, 4}) {
      const int sc1 = mode != 1, lp = mode == 2;
      for (int rep = 0; rep < 2; ++rep) {
        hipMemset(buf, 0, 1 << 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) pingpong<1, 0><<<grid, 64>>>(buf, peer, rounds, xcc, cyc); else if (mode == 1) pingpong<0, 0><<<grid, 64>>>(buf, peer, rounds, xcc, cyc); else pingpong<1, 1><<<grid, 64>>>(buf, peer, rounds, xcc, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
        long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        if (rep && hc < 0) printf("store %-5s load %-9s peer %2d  xcc %d/%d: TIMED OUT (stale read)\n", sc1 ? "sc1" : "plain", lp ? "inv+plain" : "sc1", peer, hx[0], hx[peer]);
        else if (rep) printf("store %-5s load %-9s peer %2d  xcc(ping)=%d xcc(pong)=%d  %.0f ns per one-way hand-off (1 KB message)\n", sc1 ? "sc1" : "plain", lp ? "inv+plain" : "sc1", peer, hx[0], hx[peer],
                        ms * 1e6 / rounds / 2);
      }
    }
  return 0;
}
