// 4-bit codebook weights -> bf16, for the paths that run the MFMA GEMM on them
// (prefill, verifies and batches above the GEMV's row limit): one thread expands
// one 16-byte load (32 weights, one scale) into 64 bytes of bf16.
#include "w4.h"

namespace evgpt {

__global__ __launch_bounds__(256) void w4_dequant_kernel(const unsigned char* __restrict__ packed,
                                                         const float* __restrict__ absmax, W4Code code,
                                                         bf16_t* __restrict__ out, long long n_chunks) {
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  const u32x4_t w = load_stream16(packed + c * 16);
  const float s = absmax[c >> 1];
  u32x4_t* dst = reinterpret_cast<u32x4_t*>(out + c * 32);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned int e[4];
    w4_expand8(w[j], code, e);
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(bf16lo(e[i]) * s, bf16hi(e[i]) * s);
    dst[j] = o;
  }
}

void launch_w4_dequant(const unsigned char* packed, const float* absmax, const W4Code& code,
                       bf16_t* out, long long n_rows, int K, hipStream_t st) {
  const long long n_chunks = n_rows * (K / 32);
  if (n_chunks == 0) return;
  const unsigned int blocks = static_cast<unsigned int>((n_chunks + 255) / 256);
  hipLaunchKernelGGL(w4_dequant_kernel, dim3(blocks), dim3(256), 0, st, packed, absmax, code, out, n_chunks);
}

}  // namespace evgpt
