// LDS tile image shared by the fused attention kernels (forward and backward).
#pragma once
#include "common.h"

namespace evgpt {

constexpr int TK = 64;  // keys per tile

template <int D> __device__ __forceinline__ int swz(int row);
// Two access patterns must both be conflict-free (LDS serves 256 B per cycle):
//   * ds_read_b128 of one chunk column over 16 consecutive rows (K fragments):
//     the 16 rows must land on 16 distinct 16-byte slots of the 256-byte bank row;
//   * ds_read_tr16_b64 (V^T fragments): 32 lanes per cycle read 32 contiguous
//     bytes (one aligned chunk pair) of each of 8 consecutive rows: the 8 rows must
//     land on 8 distinct 32-byte slots.
// D = 128 (256-byte rows): chunk bits [3:1] ^= row & 7, bit 0 ^= row bit 3.
// D = 64 (128-byte rows, two per bank row; row bit 0 picks the half): chunk bits
// [2:1] ^= (row >> 1) & 3, bit 0 ^= row bit 3.
template <> __device__ __forceinline__ int swz<128>(int row) { return ((row & 7) << 1) | ((row >> 3) & 1); }
template <> __device__ __forceinline__ int swz<64>(int row) { return (((row >> 1) & 3) << 1) | ((row >> 3) & 1); }

// byte offset of 16-byte chunk `ch` of row `row` inside a [TK][D] bf16 tile
template <int D>
__device__ __forceinline__ int tile_off(int row, int ch) {
  return row * (D * 2) + ((ch ^ swz<D>(row)) << 4);
}

// Image of a tile that is ONLY read by rows (the forward kernels' K tile: fragment row = lane & 15,
// chunk = 4 ks + (lane >> 4)).  ds_read_b128 serves a wave in four groups of 16 lanes that are not
// contiguous -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same plus 32 -- so one group holds
// all 16 rows, eight of them at chunk c and eight at chunk c ^ 1.  The swizzle above (built for the
// transposed reads) puts those on 8 slots twice: every K read was a 2-way conflict, a third of the
// forward kernels' LDS cycles.  XOR with the full row index (D = 128) gives the 16 rows 16 slots for
// either chunk; D = 64 has two rows per 256-byte bank row, so the XOR is the row PAIR index and
// the row's low bit picks the half.
template <int D> __device__ __forceinline__ int swz_k(int row);
template <> __device__ __forceinline__ int swz_k<128>(int row) { return row & 15; }
template <> __device__ __forceinline__ int swz_k<64>(int row) { return (row >> 1) & 7; }
template <int D>
__device__ __forceinline__ int tile_off_k(int row, int ch) {
  return row * (D * 2) + ((ch ^ swz_k<D>(row)) << 4);
}

}  // namespace evgpt
