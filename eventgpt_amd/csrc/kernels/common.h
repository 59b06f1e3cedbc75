// Shared device helpers for the gfx950 (CDNA4) kernels.
//
// Conventions used by every kernel in this directory:
//   * a wavefront is 64 lanes; block sizes are multiples of 64;
//   * bf16 tensors are addressed as `bf16_t` (= unsigned short) and moved in
//     16-byte vectors (`bf16x8`); arithmetic is fp32;
//   * launchers take raw device pointers + a hipStream_t, allocate nothing and
//     never synchronise, so every launch can be captured into a hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evgpt {

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) float f32x4_t;      // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;    // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

constexpr int kWave = 64;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(static_cast<unsigned int>(v) << 16);
}
// Round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 and keeps NaNs NaN.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = static_cast<__bf16>(f);
  return __builtin_bit_cast(bf16_t, b);
}
// One v_cvt_pk_bf16_f32 for the pair (two scalar casts + shift + or are four instructions).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

// dot of two packed bf16 pairs accumulated in fp32 (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2_bf16(unsigned int a, unsigned int b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a),
                                         __builtin_bit_cast(bf16x2_t, b), acc, false);
}
__device__ __forceinline__ float dot8_bf16(const u32x4_t& a, const u32x4_t& b, float acc) {
  acc = dot2_bf16(a[0], b[0], acc);
  acc = dot2_bf16(a[1], b[1], acc);
  acc = dot2_bf16(a[2], b[2], acc);
  acc = dot2_bf16(a[3], b[3], acc);
  return acc;
}

// Wave-wide sum, returned in every lane.  Six DPP adds walk the sum into lane 63 (quad
// swaps, half-row and row mirrors, then row_bcast:15 / row_bcast:31 carry row totals across
// the four 16-lane rows) and a v_readlane hands it back as a scalar.  The __shfl_xor
// butterfly costs six ds_bpermute plus their address arithmetic per value — in the decode
// GEMVs, whose waves stream only 8 KB each, that was a quarter of all VALU instructions and
// all of the LDS-pipe traffic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(moved);
}
// Sum over aligned groups of 8 or 16 lanes (a DPP row is 16 lanes), result in every lane of
// the group: 3 or 4 VALU adds, no LDS-pipe traffic.
template <int LANES>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(LANES == 8 || LANES == 16, "group of 8 or 16 lanes");
  v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);   // row_half_mirror: 8-lane sums
  if constexpr (LANES == 16) v = dpp_add<0x140, 0xF>(v);   // row_mirror: 16-lane sums
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);   // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);   // row_mirror: every lane holds its row's sum
  v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide sum for blocks of up to 1024 threads; `scratch` holds >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += scratch[i];
  return t;
}

// streamed-once 16-byte load (weights): non-temporal so it does not displace
// the activations / KV cache that are re-read from L2.
__device__ __forceinline__ u32x4_t load_stream16(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
}
__device__ __forceinline__ u32x4_t load16(const void* p) {
  return *reinterpret_cast<const u32x4_t*>(p);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.f + __expf(-1.702f * x)); }

enum Act : int { ACT_NONE = 0, ACT_GELU = 1, ACT_QUICK_GELU = 2, ACT_SILU = 3, ACT_RELU = 4 };

__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case ACT_GELU: return gelu_f(x);
    case ACT_QUICK_GELU: return quick_gelu_f(x);
    case ACT_SILU: return silu_f(x);
    case ACT_RELU: return fmaxf(x, 0.f);
    default: return x;
  }
}

// Bijective XCD-aware block remap: blocks that share `id % 8` sit on one XCD,
// so give each XCD a contiguous range of logical tiles (neighbouring tiles then
// share an L2).  Valid for any grid size.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7;
  const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (id >> 3);
}

}  // namespace evgpt
