// Shared by gemm16.hip (tile configurations 0-14) and gemm16_8p.hip (the eight-phase 256 x 256 schedule): the LDS image of a bf16 operand tile.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "gemm_core.h"

typedef unsigned short bf16_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

#define G16_BK 64

template <int I, int N, class F>
__device__ __forceinline__ void g16_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    g16_static_for<I + 1, N>(f);
  }
}

// byte offset of chunk ch (8 bf16) of a tile row.  The swizzle is chosen for ds_read_b128's LANE GROUPS (MI355X_MICROARCH.md, LDS: a wave's read is
// served in four groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 - and the 16 lanes of a group must fall on 16
// different 16-byte slots of the 256-byte bank row): a fragment read has lane = row, so a group's rows r must differ in (r & 1, (r >> 1) & 7 ^ ch),
// which they do for both groups.  (Rounds 3-4 XORed with r & 7: rows r and r + 24 of a group then shared a slot - every read 2-way conflicted.)
__device__ __forceinline__ int g16_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int g16_off(int row, int ch) { return row * 128 + 16 * (ch ^ g16_swz(row)); }

// Tile order.  Workgroups b, b + 8, ... share an XCD (and its 4 MB L2), so every XCD walks a contiguous range of the tile sequence p; the sequence
// itself runs down GROUPS of 4 tile rows, column by column, so that the ~32 tiles an XCD has in flight form a 4 x 8 block - 12 operand strips
// per K step through the fabric for 32 workgroups instead of the 33 of a 1 x 32 row of tiles (round 4; measured on [8192]^3: see DESIGN.md).
__device__ __forceinline__ void g16_tile(int bid, int tiles_m, int tiles_n, int& bm, int& bn) {
  const int total = tiles_m * tiles_n;
  const int xcd = bid & 7, idx = bid >> 3;
  const int p = xcd * (total >> 3) + min(xcd, total & 7) + idx;
  const int per = 4 * tiles_n, grp = p / per, q = p - grp * per;
  const int rows = min(4, tiles_m - 4 * grp);
  bn = q / rows;
  bm = 4 * grp + (q - bn * rows);
}

struct G16Launch {
  const bf16_t *A, *B; GemmEpilogue ep; const asr_gemm_desc* d; int sk; hipStream_t st;
};
// gemm16_8p.hip: returns the hipFuncSetAttribute status (not hipSuccess: nothing was launched)
hipError_t g16_launch_8p(const G16Launch& g);
