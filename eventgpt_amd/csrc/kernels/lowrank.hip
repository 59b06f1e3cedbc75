// Rank-r projection of a low-rank adapter: C [M, R] = alpha * A [M, K] · W [R, K]^T with R <= 16.
//
// The output is 16 columns wide and the contraction is the model width, so a tiled GEMM spends its
// time on a 64- or 128-column W tile that is mostly padding and on a split-K reduction pass.  Here a
// block owns 16 rows of A and its 16 waves share the contraction: wave w takes the 32-column K-steps
// w, w + 16, ..., loads the A and W fragments of a step straight from memory in the MFMA operand
// layout (row = lane % 16, eight consecutive k at 8 * (lane / 16)), and the 16 partial 16 x 16
// accumulators are summed through LDS.  A [836, 4096] input is 53 blocks of 1024 threads with every
// load of a wave in flight at once; W (128 KB at r = 16) stays in L2.
//
// A may be a column block of a wider matrix (row stride lda), and the adapter's input dropout can be
// applied while A is loaded (DROP): element (m, k) is zeroed unless its bit of lowrank_keep8(seed,
// chunk (m K + k) / 8) is set -- the 1 / (1 - p) factor rides on alpha -- so the dropped copy of the
// layer input is never written.  The backward recomputes the same mask from the seed
// (lowrank_mask_kernel: m (.) x for the weight gradient of A, acc += m (.) g for the input gradient).
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

constexpr int LR_WAVES = 16;

// keep bits of the 8 elements of 16-byte chunk `chunk`: two 64-bit mixes, one 16-bit field per element
__device__ __forceinline__ unsigned int lowrank_keep8(unsigned long long seed, unsigned long long chunk,
                                                      unsigned int thresh16) {
  unsigned int bits = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    unsigned long long z = seed + (2 * chunk + h) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
#pragma unroll
    for (int f = 0; f < 4; ++f)
      bits |= (static_cast<unsigned int>(z >> (16 * f)) & 0xffffu) >= thresh16 ? 1u << (4 * h + f) : 0u;
  }
  return bits;
}
// zero the dropped elements of a chunk (element e: low / high half of word e / 2)
__device__ __forceinline__ u32x4_t lowrank_apply8(u32x4_t v, unsigned int keep) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned int m = ((keep >> (2 * j)) & 1u ? 0x0000ffffu : 0u) | ((keep >> (2 * j + 1)) & 1u ? 0xffff0000u : 0u);
    v[j] &= m;
  }
  return v;
}
static unsigned int lowrank_thresh16(float p) {
  if (p <= 0.f) return 0u;
  const double t = static_cast<double>(p) * 65536.0;
  return t >= 65535.0 ? 65535u : static_cast<unsigned int>(t);
}

template <bool DROP>
__global__ __launch_bounds__(LR_WAVES * 64) void lowrank_out_kernel(const bf16_t* __restrict__ A,
                                                                   const bf16_t* __restrict__ W,
                                                                   bf16_t* __restrict__ C, int M, int R, int K,
                                                                   long long lda, float alpha, unsigned int thresh16,
                                                                   unsigned long long seed) {
  __shared__ float red[LR_WAVES][256];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int frow = lane & 15, fq = lane >> 4;
  const int m = blockIdx.x * 16 + frow;
  const bool a_ok = m < M, w_ok = frow < R;
  const bf16_t* a_row = A + static_cast<size_t>(a_ok ? m : 0) * lda;
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
    if constexpr (DROP) {
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int k = kb + u * LR_WAVES * 32 + fq * 8;
        const unsigned long long chunk = (static_cast<unsigned long long>(a_ok ? m : 0) * K + (k < K ? k : 0)) >> 3;
        af[u] = __builtin_bit_cast(bf16x8_t, lowrank_apply8(__builtin_bit_cast(u32x4_t, af[u]),
                                                            lowrank_keep8(seed, chunk, thresh16)));
      }
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

void launch_lowrank_out(const bf16_t* A, const bf16_t* W, bf16_t* C, int M, int R, int K, long long lda, float alpha,
                        float drop_p, unsigned long long seed, hipStream_t st) {
  if (M <= 0) return;
  const unsigned int t16 = lowrank_thresh16(drop_p);
  if (t16)
    hipLaunchKernelGGL(lowrank_out_kernel<true>, dim3((M + 15) / 16), dim3(LR_WAVES * 64), 0, st, A, W, C, M, R, K, lda,
                       alpha / (1.f - drop_p), t16, seed);
  else
    hipLaunchKernelGGL(lowrank_out_kernel<false>, dim3((M + 15) / 16), dim3(LR_WAVES * 64), 0, st, A, W, C, M, R, K, lda,
                       alpha, 0u, 0ull);
}

// y = m (.) x  (ADD = false)  or  y += m (.) x  (ADD = true) over contiguous [rows][K] tensors, m the keep mask
// of lowrank_keep8 (no 1 / (1 - p) factor: the callers carry it on the rank-r intermediate)
template <bool ADD>
__global__ __launch_bounds__(256) void lowrank_mask_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                           long long n8, unsigned int thresh16,
                                                           unsigned long long seed) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += gridDim.x * 256ll) {
    u32x4_t v = lowrank_apply8(load16(x + i * 8), lowrank_keep8(seed, static_cast<unsigned long long>(i), thresh16));
    if constexpr (ADD) {
      const u32x4_t a = load16(y + i * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        v[j] = pack_bf16x2(bf16lo(a[j]) + bf16lo(v[j]), bf16hi(a[j]) + bf16hi(v[j]));
    }
    *reinterpret_cast<u32x4_t*>(y + i * 8) = v;
  }
}
void launch_lowrank_mask(const bf16_t* x, bf16_t* y, long long n, float drop_p, unsigned long long seed, bool add,
                         hipStream_t st) {
  if (n <= 0) return;
  long long b = (n / 8 + 255) / 256;
  const unsigned blocks = static_cast<unsigned>(b > 8192 ? 8192 : (b < 1 ? 1 : b));
  const unsigned int t16 = lowrank_thresh16(drop_p);
  if (add) hipLaunchKernelGGL(lowrank_mask_kernel<true>, dim3(blocks), dim3(256), 0, st, x, y, n / 8, t16, seed);
  else hipLaunchKernelGGL(lowrank_mask_kernel<false>, dim3(blocks), dim3(256), 0, st, x, y, n / 8, t16, seed);
}

}  // namespace evgpt
