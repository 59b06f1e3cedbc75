// Rank-r projection of a low-rank adapter: C [M, R] = alpha * A [M, K] · W [R, K]^T with R <= 16.
//
// The output is 16 columns wide and the contraction is the model width, so a tiled GEMM spends its
// time on a 64- or 128-column W tile that is mostly padding and on a split-K reduction pass.  Here a
// block owns 16 rows of A and its 16 waves share the contraction: wave w takes the 32-column K-steps
// w, w + 16, ..., loads the A and W fragments of a step straight from memory in the MFMA operand
// layout (row = lane % 16, eight consecutive k at 8 * (lane / 16)), and the 16 partial 16 x 16
// accumulators are summed through LDS.  A [836, 4096] input is 53 blocks of 1024 threads with every
// load of a wave in flight at once; W (128 KB at r = 16) stays in L2.
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

constexpr int LR_WAVES = 16;

__global__ __launch_bounds__(LR_WAVES * 64) void lowrank_out_kernel(const bf16_t* __restrict__ A,
                                                                   const bf16_t* __restrict__ W,
                                                                   bf16_t* __restrict__ C, int M, int R, int K,
                                                                   float alpha) {
  __shared__ float red[LR_WAVES][256];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int frow = lane & 15, fq = lane >> 4;
  const int m = blockIdx.x * 16 + frow;
  const bool a_ok = m < M, w_ok = frow < R;
  const bf16_t* a_row = A + static_cast<size_t>(a_ok ? m : 0) * K;
  const bf16_t* w_row = W + static_cast<size_t>(w_ok ? frow : 0) * K;
  const u32x4_t z4 = {0u, 0u, 0u, 0u};
  const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, z4);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int UN = 8;   // K = 4096: the eight steps of a wave are issued together
  // the loop bound is the same for all lanes of a wave (an MFMA needs every lane's operands, so no lane
  // may leave early); a lane whose chunk of the last step lies past K contributes zeros
  for (int kb = wid * 32; kb < K; kb += UN * LR_WAVES * 32) {
    bf16x8_t af[UN], wf[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = kb + u * LR_WAVES * 32 + fq * 8;
      const bool in = k < K;          // K is a multiple of 8: a chunk is inside or outside as a whole
      af[u] = (in && a_ok) ? *reinterpret_cast<const bf16x8_t*>(a_row + k) : zero;
      wf[u] = (in && w_ok) ? *reinterpret_cast<const bf16x8_t*>(w_row + k) : zero;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af[u], acc, 0, 0, 0);
  }
  // acc[e] = sum over this wave's steps of  W[r = 4 * fq + e] · A[m = frow]
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wid][lane * 4 + e] = acc[e];
  __syncthreads();
  const int t = threadIdx.x;
  if (t < 256) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < LR_WAVES; ++w) s += red[w][t];
    const int ln = t >> 2, e = t & 3;
    const int r = 4 * (ln >> 4) + e, mm = blockIdx.x * 16 + (ln & 15);
    if (mm < M && r < R) C[static_cast<size_t>(mm) * R + r] = f32_to_bf16(s * alpha);
  }
}

void launch_lowrank_out(const bf16_t* A, const bf16_t* W, bf16_t* C, int M, int R, int K, float alpha,
                        hipStream_t st) {
  if (M <= 0) return;
  hipLaunchKernelGGL(lowrank_out_kernel, dim3((M + 15) / 16), dim3(LR_WAVES * 64), 0, st, A, W, C, M, R, K, alpha);
}

}  // namespace evgpt
