// 4-bit codebook weights: register-resident table lookup.
//
// A 32-bit word carries 8 codes: byte j = code of weight j (low nibble) and of
// weight 4+j (high nibble).  The 16 bf16 codebook entries are held as two byte
// planes of four words each; v_perm_b32 picks a byte out of eight, so entries
// 0..7 and 8..15 are looked up separately and a third permute selects between
// the two by bit 3 of the code.  Two more permutes interleave the planes into
// bf16 pairs, in the order v_dot2c_f32_bf16 wants (weights 2m, 2m+1 against the
// x word m).
#pragma once
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

// v_perm_b32: selector bytes 0..3 take bytes of `lo_src`, 4..7 bytes of `hi_src`
__device__ __forceinline__ unsigned int perm_b32(unsigned int hi_src, unsigned int lo_src,
                                                 unsigned int sel) {
  return __builtin_amdgcn_perm(hi_src, lo_src, sel);
}

// 8 codes -> 4 words of bf16 pairs: out[m] = (value[2m], value[2m+1])
__device__ __forceinline__ void w4_expand8(unsigned int w, const W4Code& cb, unsigned int (&out)[4]) {
  const unsigned int idx[2] = {w & 0x0F0F0F0Fu, (w >> 4) & 0x0F0F0F0Fu};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned int s7 = idx[i] & 0x07070707u;
    const unsigned int alo = perm_b32(cb.lo[1], cb.lo[0], s7);  // entries 0..7
    const unsigned int blo = perm_b32(cb.lo[3], cb.lo[2], s7);  // entries 8..15
    const unsigned int ahi = perm_b32(cb.hi[1], cb.hi[0], s7);
    const unsigned int bhi = perm_b32(cb.hi[3], cb.hi[2], s7);
    const unsigned int pick = ((idx[i] >> 1) & 0x04040404u) | 0x03020100u;  // byte j of a or of b
    const unsigned int lo = perm_b32(blo, alo, pick);
    const unsigned int hi = perm_b32(bhi, ahi, pick);
    out[2 * i] = perm_b32(hi, lo, 0x05010400u);
    out[2 * i + 1] = perm_b32(hi, lo, 0x07030602u);
  }
}

}  // namespace evgpt
