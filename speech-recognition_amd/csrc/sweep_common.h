// Shared by the six one-launch sweeps (rnn_sweep.hip, rnn_sweep_bwd.hip, rnn_sweep_wide.hip, rnn_sweep_wide_bwd.hip, decoder_sweep.hip,
// decoder_sweep_bwd.hip): the 32 diagnosis words that sit behind every sweep's exchange buffer, the start handshake and the
// record a workgroup leaves when one of its bounded spins gives up.
//
// A sweep's workgroups wait for each other, so a launch only makes progress once ALL of them are resident.  Nothing in HIP
// promises that (the occupancy API is a query, not a reservation): another stream's workgroups may hold the compute units a
// late workgroup needs, and the resident ones then spin until it arrives.  The spins are bounded (asr_rnn_sweep_set_spin_limit),
// so the worst case is a skipped training step, never a hang - but a time-out must say WHICH of the two things happened:
//   * a workgroup was not resident (arrivals < expected when the first workgroup gave up), or
//   * every workgroup was there and a hand-off was lost all the same (protocol or memory-system fault).
// Every workgroup therefore counts itself in at its first instruction (word 4) and the first one to give up records what it
// saw (words 8-15).  The arrival count is also what asr_sweep_gate polls: work that should run BESIDE a sweep (weight-gradient
// GEMMs on a side stream) is released only once the sweep is resident, so that it cannot take the sweep's compute units first.
//
//   word 0       error word of the launch: 0, or (code | step << 8 ...) of a workgroup that gave up (kernel specific)
//   word 1       decoder sweeps: 0x7fffffff - the earliest (step, stage) that gave up anywhere
//   words 2, 3   encoder sweeps in XCD mode: workgroups that publish XCD-locally / all workgroups
//   word 4       arrivals of this launch (the last workgroup to leave puts it back to 0)
//   word 5       expected workgroups (written by sw_fill_kernel before the launch)
//   word 6       departures
//   words 8-15   record of the FIRST workgroup that gave up: 8 its abort word, 9 block (x | y << 12 | z << 24), 10 HW_REG_XCC_ID,
//                11 arrivals at that moment, 12 XCD-local mode (1) or write-through (0), 13 wave, 14 s_memrealtime (low word), 15 taken
//   words 16-23  STICKY copy of the first such record since the host last cleared it (words 0-15 are re-armed by every launch, and
//                the launch that failed is rarely the last one before anybody looks); word 24 launches that gave up since then
//   words 25, 26 encoder BPTT sweep, poll statistics (accumulated until the host clears them): gathers whose first poll came too early,
//                gather waves that reported
#pragma once
#include "common.h"

// Progress words and abort flags live in LDS and are polled by the other waves of the workgroup.  They must be reached through
// LDS-TYPED pointers: `volatile int*` is a generic pointer, address-space inference leaves volatile accesses alone, and the access
// becomes flat_load / flat_store ... sc0 sc1 + s_waitcnt vmcnt(0) - the texture path's latency instead of a ds_read's, and an entry in
// the wave's in-order memory counter in front of its polls (round 4: 186 such instructions in the decoder sweep, the BPTT sweep's
// "partial block ready" mark waited for its own flat store).  Through these types the same source compiles to ds_read_b32 / ds_write_b32.
typedef __attribute__((address_space(3))) int lds_int_t;
typedef volatile lds_int_t* lds_flag_t;
__device__ __forceinline__ lds_flag_t lds_flag(int* p) { return (lds_flag_t)p; }
__device__ __forceinline__ int lds_peek(const int* p) { return *(const volatile lds_int_t*)p; }
__device__ __forceinline__ void lds_poke(int* p, int v) { *(volatile lds_int_t*)p = v; }

#define SWD_ARRIVED 4
#define SWD_EXPECTED 5
#define SWD_DEPARTED 6
#define SWD_RECORD 8

extern "C" int asr_rnn_sweep_spin_limit(void);
// wave priority of the sweeps' waves (s_setprio 0-3): they are latency bound and share compute units with throughput work released
// beside them (asr_sweep_gate); at equal priority that work's back-to-back MFMAs delay every dependent instruction of a step
int asr_sweep_prio(void);
__device__ __forceinline__ void swd_setprio(int prio) {
  if (prio >= 3) __builtin_amdgcn_s_setprio(3);
  else if (prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (prio == 1) __builtin_amdgcn_s_setprio(1);
}
long asr_sweep_capacity(const void* kernel, int threads);
// fills p[0..n) with v, zeroes zero_words[0..nzero) and stores the number of workgroups the sweep will run in zero_words[SWD_EXPECTED]
__global__ void sw_fill_kernel(uint32_t* p, size_t n, uint32_t v, uint32_t* zero_words, int nzero, uint32_t expected);

// one lane per workgroup, first thing in the kernel (fire and forget: nobody waits for the returned value)
__device__ __forceinline__ void swd_arrive(unsigned* err) {
  __hip_atomic_fetch_add(err + SWD_ARRIVED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane per workgroup, after the workgroup's own set-up and before its first step: wait until the whole grid has counted itself in.
// The waits of the steps are bounded in POLLS (a lost hand-off shows within ~a second); this one is bounded in TIME and generously
// (20 s of the 100 MHz s_memrealtime clock): a foreign kernel that holds compute units - an RCCL collective, a co-tenant - only
// delays the launch, it does not fail it.  Separates the two failure modes for good: after this, every workgroup IS resident.
// spin_limit <= 0 (the tests' forced time-outs) does not wait at all.  false = the grid did not assemble (the caller aborts).
__device__ __forceinline__ bool swd_wait_all(unsigned* err, int spin_limit) {
  if (spin_limit <= 0) return true;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const unsigned expected = __hip_atomic_load(err + SWD_EXPECTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned arrived = __hip_atomic_load(err + SWD_ARRIVED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived >= expected) return true;
    if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) return false;
    __builtin_amdgcn_s_sleep(8);
  }
}
// one lane per workgroup, last thing in the kernel: the last one out re-arms the handshake for asr_sweep_gate
// (returns true to the last workgroup out: everything the others did before their departure is visible to it)
__device__ __forceinline__ bool swd_depart(unsigned* err) {
  const unsigned expected = __hip_atomic_load(err + SWD_EXPECTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned n = __hip_atomic_fetch_add(err + SWD_DEPARTED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (expected != 0 && n + 1 == expected) {
    __hip_atomic_store(err + SWD_ARRIVED, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(err + SWD_DEPARTED, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
  return false;
}
// one lane of a workgroup that gave up: only the first caller of the launch leaves a record
__device__ __forceinline__ void swd_record(unsigned* err, unsigned word, int local_mode) {
  if (__hip_atomic_fetch_add(err + SWD_RECORD + 7, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  err[SWD_RECORD + 0] = word;
  err[SWD_RECORD + 1] = blockIdx.x | (blockIdx.y << 12) | (blockIdx.z << 24);
  err[SWD_RECORD + 2] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 15u;       // HW_REG_XCC_ID
  err[SWD_RECORD + 3] = __hip_atomic_load(err + SWD_ARRIVED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  err[SWD_RECORD + 4] = (unsigned)local_mode;
  err[SWD_RECORD + 5] = threadIdx.x >> 6;
  err[SWD_RECORD + 6] = (unsigned)__builtin_amdgcn_s_memrealtime();
  if (__hip_atomic_fetch_add(err + 24, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
#pragma unroll
    for (int i = 0; i < 7; ++i) err[16 + i] = err[SWD_RECORD + i];
    err[16 + 7] = __hip_atomic_load(err + SWD_EXPECTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
