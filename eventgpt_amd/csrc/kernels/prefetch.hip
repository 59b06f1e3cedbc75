// Pull a byte range of a weight matrix into the die-level cache ahead of the kernel that streams it.
//
// A decode step is a chain of weight-streaming launches; while the attention launches of a layer run
// (latency-bound, a few MB of K/V) and across launch boundaries the HBM interface is idle.  This
// kernel, issued on a second stream next to them, reads the next projection's weights with
// default-policy loads and drops the values: the lines stay in the 256 MiB memory-side cache, from
// which the projection then reads faster than from HBM.
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

__global__ __launch_bounds__(256) void l3_prefetch_kernel(const u32x4_t* __restrict__ src, long long n16,
                                                          unsigned int* __restrict__ sink) {
  const long long stride = gridDim.x * 256ll;
  long long i = blockIdx.x * 256ll + threadIdx.x;
  unsigned int acc = 0;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    u32x4_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[i + j * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
  }
  for (; i < n16; i += stride) {
    const u32x4_t v = src[i];
    acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  // the loads must not be optimised away: a store nobody expects to happen depends on them
  if (acc == 0x9e3779b9u && blockIdx.x == 0x7fffffff) *sink = acc;
}

void launch_l3_prefetch(const void* src, long long bytes, unsigned int* sink, int blocks, hipStream_t st) {
  const long long n16 = bytes / 16;
  if (n16 <= 0) return;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(l3_prefetch_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const u32x4_t*>(src), n16, sink);
}

}  // namespace evgpt
