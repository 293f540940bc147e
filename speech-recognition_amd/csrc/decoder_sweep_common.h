// Shared pieces of the one-launch decoder sweeps (decoder_sweep.hip forward, decoder_sweep_bwd.hip backward): the sentinel
// hand-off primitives (self-validating 16-byte pieces, bounded polls) and the LDS progress words.  See rnn_sweep.hip for the protocol.
#pragma once
#include "sweep_common.h"

#ifndef DS_PROBE_SLEEP
#define DS_PROBE_SLEEP 4            // 64-cycle periods between two polls of the probe piece
#endif
#define DS_SLOTS 4
#define DS_SENT 0x7FC0DEADu
#define DS_NC 8                    // time chunks per batch row
#define DS_MAXTC 32                // encoder frames of a chunk RESIDENT in LDS
#define DS_MAXTC2 64               // encoder frames per chunk: the ones beyond DS_MAXTC are streamed from L2 / Infinity Cache every step
#define DS_MAXHB 4                 // K blocks of h per gather wave: Hd <= 256
#define DS_MAXCB 8                 // K blocks of the context per gather wave: D <= 512

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ bool ds_fresh(const f32x4& v) {
  return __float_as_uint(v.x) != DS_SENT && __float_as_uint(v.y) != DS_SENT && __float_as_uint(v.z) != DS_SENT && __float_as_uint(v.w) != DS_SENT;
}
// code | (step << 8): 1 h1 gather, 2 partial gather, 3 context gather, 4 h0 gather, 5-8 LDS hand-overs
__device__ __forceinline__ bool ds_wait(lds_flag_t c, int target, lds_flag_t abort_flag, int limit, int code) {
  for (int i = 0; *c < target; ++i) {
    if (*abort_flag) return false;
    if (i > limit) { *abort_flag = code; return false; }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}
// While the data is not there yet only ONE of the pieces is polled (the pieces of a step are published within a fraction of a
// microsecond of each other, so the first one is a good predictor): 256 workgroups x 4 waves re-reading 5-8 KB each per microsecond
// for the several microseconds a role waits for its turn would put terabytes per second of polls in front of the publishes.
// Addresses are (wave-uniform base in SGPRs, 32-bit byte offset per lane): half the address registers of flat pointers.
__device__ __forceinline__ bool ds_probe(const float* base, unsigned off, bool use, lds_flag_t abort_flag, int limit, int code) {
  for (int spins = 0;; ++spins) {
    f32x4 v;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(off), "s"(base) : "memory");
    const bool ok = !use || ds_fresh(v);
    if (__all(ok)) return true;
    if (*abort_flag) return false;
    if (spins > limit) { *abort_flag = code; return false; }
    __builtin_amdgcn_s_sleep(DS_PROBE_SLEEP);
  }
}
// up to 5 self-validating 16-byte pieces per lane: re-read until none of the USED ones holds the sentinel
__device__ __forceinline__ bool ds_gather5(const float* base, const unsigned (&off)[5], const bool (&use)[5], f32x4 (&v)[5], lds_flag_t abort_flag,
                                           int limit, int delay, int code) {
  if (delay >= 0 && delay < 1000) {
    // the probe piece is the first one any lane uses - chosen with wave-uniform branches over STATIC indices (a run-time pick
    // from the offset array goes through scratch memory: a memory round trip in front of every gather)
    if (__any(use[0])) { if (!ds_probe(base, off[0], use[0], abort_flag, limit, code)) return false; }
    else if (__any(use[1])) { if (!ds_probe(base, off[1], use[1], abort_flag, limit, code)) return false; }
    else if (__any(use[2])) { if (!ds_probe(base, off[2], use[2], abort_flag, limit, code)) return false; }
    else if (__any(use[3])) { if (!ds_probe(base, off[3], use[3], abort_flag, limit, code)) return false; }
    else if (__any(use[4])) { if (!ds_probe(base, off[4], use[4], abort_flag, limit, code)) return false; }
  }
  for (int spins = 0;; ++spins) {
    asm volatile(
        "s_nop 4\n\t"                                      // the base may have just been written by scalar ALU code the compiler does not
                                                          // know a memory instruction reads (cdna_hip_programming.md 5.7, item 2)
        "global_load_dwordx4 %0, %5, %10 sc1\n\t"
        "global_load_dwordx4 %1, %6, %10 sc1\n\t"
        "global_load_dwordx4 %2, %7, %10 sc1\n\t"
        "global_load_dwordx4 %3, %8, %10 sc1\n\t"
        "global_load_dwordx4 %4, %9, %10 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
        : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "s"(base)
        : "memory");
    bool ok = true;
    int bad = 7;
#pragma unroll
    for (int i = 4; i >= 0; --i)
      if (use[i] && !ds_fresh(v[i])) { ok = false; bad = i; }
    if (__all(ok) || delay < 0) return true;
    if (*abort_flag) return false;
    if (spins > limit) {                                  // diagnosis: which piece of which lane never arrived
      const unsigned long long ball = __ballot(!ok);
      const int fl = __ffsll((long long)ball) - 1;
      *abort_flag = code | (__shfl(bad, fl, 64) << 16) | (fl << 20);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ bool ds_gather8(const float* base, const unsigned (&off)[8], const bool (&use)[8], f32x4 (&v)[8], lds_flag_t abort_flag,
                                           int limit, int delay, int code) {
  if (delay >= 0 && delay < 1000 && !ds_probe(base, off[0], use[0], abort_flag, limit, code)) return false;
  for (int spins = 0;; ++spins) {
    asm volatile(
        "s_nop 4\n\t"
        "global_load_dwordx4 %0, %8, %16 sc1\n\t"
        "global_load_dwordx4 %1, %9, %16 sc1\n\t"
        "global_load_dwordx4 %2, %10, %16 sc1\n\t"
        "global_load_dwordx4 %3, %11, %16 sc1\n\t"
        "global_load_dwordx4 %4, %12, %16 sc1\n\t"
        "global_load_dwordx4 %5, %13, %16 sc1\n\t"
        "global_load_dwordx4 %6, %14, %16 sc1\n\t"
        "global_load_dwordx4 %7, %15, %16 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]), "s"(base)
        : "memory");
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (use[i]) ok = ok && ds_fresh(v[i]);
    if (__all(ok) || delay < 0) return true;
    if (*abort_flag) return false;
    if (spins > limit) { *abort_flag = code; return false; }
    __builtin_amdgcn_s_sleep(1);
  }
}
// Hand-overs inside the workgroup: every gather wave keeps its OWN progress word per hand-over point (the step it has finished,
// plus one); a waiter needs all four.  (One shared cumulative counter is not enough here: a gather wave that has nothing to
// gather at some point of the step runs ahead, and its increments for LATER steps would complete the count of an earlier one.)
__device__ __forceinline__ void ds_mark(lds_flag_t c4, int wave, int value) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) c4[wave] = value;
}
__device__ __forceinline__ bool ds_wait4(lds_flag_t c4, int target, lds_flag_t abort_flag, int limit, int code) {
  for (int i = 0;; ++i) {
    const int v = c4[threadIdx.x & 3];
    if (__all(v >= target)) return true;
    if (*abort_flag) return false;
    if (i > limit) { *abort_flag = code; return false; }
    __builtin_amdgcn_s_sleep(1);
  }
}

