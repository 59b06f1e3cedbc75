// Backward kernels of the training path's fused elementwise ops (forward = the inference
// kernels).  Each replaces 8-12 eager elementwise launches per call site and layer.
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

// two block-wide sums at once; `scratch` holds >= 32 floats
__device__ __forceinline__ void block_sum2(float& a, float& b, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  a = wave_sum(a); b = wave_sum(b);
  __syncthreads();
  if (lane == 0) { scratch[wid] = a; scratch[16 + wid] = b; }
  __syncthreads();
  float ta = 0.f, tb = 0.f;
  for (int i = 0; i < nw; ++i) { ta += scratch[i]; tb += scratch[16 + i]; }
  a = ta; b = tb;
}

// y = x * rstd * w  =>  dx = rstd * (g w) - x * rstd^3 * sum(g w x) / H,   dw += g * x * rstd
// One block walks `rows_per_block` rows; a thread owns the same column chunks in every row, so
// the weight gradient is summed in registers and leaves the block as one row of a
// [blocks][H] partial buffer that the caller sums over blocks (atomics on one [H] vector
// made this kernel 75 us for 836 rows; the partial rows are ~10 us including the sum).
constexpr int RB_MAX_CHUNKS = 4;   // 256 threads x 4 chunks x 8 columns = 8192 columns
template <bool DW>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          const bf16_t* __restrict__ g, bf16_t* __restrict__ dx,
                                                          float* __restrict__ dw, int rows, int H, float eps,
                                                          int rows_per_block) {
  __shared__ float scratch[32];
  const int tid = threadIdx.x, nchunk = H >> 3;
  float dwacc[RB_MAX_CHUNKS][8];
#pragma unroll
  for (int i = 0; i < RB_MAX_CHUNKS; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) dwacc[i][k] = 0.f;
  u32x4_t wv[RB_MAX_CHUNKS];
#pragma unroll
  for (int i = 0; i < RB_MAX_CHUNKS; ++i) {
    const int c = tid + i * 256;
    wv[i] = c < nchunk ? load16(w + c * 8) : u32x4_t{0, 0, 0, 0};
  }
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, rows);
  for (int r = r0; r < r1; ++r) {
    const bf16_t* xr = x + static_cast<size_t>(r) * H;
    const bf16_t* gr = g + static_cast<size_t>(r) * H;
    u32x4_t xv[RB_MAX_CHUNKS], gv[RB_MAX_CHUNKS];
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int i = 0; i < RB_MAX_CHUNKS; ++i) {
      const int c = tid + i * 256;
      const bool ok = c < nchunk;
      xv[i] = ok ? load16(xr + c * 8) : u32x4_t{0, 0, 0, 0};
      gv[i] = ok ? load16(gr + c * 8) : u32x4_t{0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = bf16lo(xv[i][j]), x1 = bf16hi(xv[i][j]);
        ss += x0 * x0 + x1 * x1;
        dot += bf16lo(gv[i][j]) * bf16lo(wv[i][j]) * x0 + bf16hi(gv[i][j]) * bf16hi(wv[i][j]) * x1;
      }
    }
    block_sum2(ss, dot, scratch);
    const float rstd = rsqrtf(ss / static_cast<float>(H) + eps);
    const float coef = rstd * rstd * rstd * dot / static_cast<float>(H);
#pragma unroll
    for (int i = 0; i < RB_MAX_CHUNKS; ++i) {
      const int c = tid + i * 256;
      if (c >= nchunk) continue;
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = bf16lo(xv[i][j]), x1 = bf16hi(xv[i][j]);
        const float g0 = bf16lo(gv[i][j]), g1 = bf16hi(gv[i][j]);
        o[j] = pack_bf16x2(rstd * g0 * bf16lo(wv[i][j]) - x0 * coef, rstd * g1 * bf16hi(wv[i][j]) - x1 * coef);
        if constexpr (DW) { dwacc[i][2 * j] += g0 * x0 * rstd; dwacc[i][2 * j + 1] += g1 * x1 * rstd; }
      }
      *reinterpret_cast<u32x4_t*>(dx + static_cast<size_t>(r) * H + c * 8) = o;
    }
  }
  if constexpr (DW) {
#pragma unroll
    for (int i = 0; i < RB_MAX_CHUNKS; ++i) {
      const int c = tid + i * 256;
      if (c >= nchunk) continue;
#pragma unroll
      for (int k = 0; k < 8; ++k) dw[static_cast<size_t>(blockIdx.x) * H + c * 8 + k] = dwacc[i][k];
    }
  }
}

// rows handled by one block, and the number of blocks (= rows of the dw partial buffer)
int rmsnorm_bwd_rows_per_block(int rows, bool with_dw) {
  int rpb = (rows + 1023) / 1024;
  if (with_dw && rpb < 4) rpb = 4;
  return rpb < 1 ? 1 : rpb;
}
int rmsnorm_bwd_blocks(int rows, bool with_dw) {
  const int rpb = rmsnorm_bwd_rows_per_block(rows, with_dw);
  return (rows + rpb - 1) / rpb;
}

void launch_rmsnorm_bwd(const bf16_t* x, const bf16_t* w, const bf16_t* g, bf16_t* dx, float* dw, int rows,
                        int H, float eps, hipStream_t st) {
  if (rows <= 0) return;
  const int rpb = rmsnorm_bwd_rows_per_block(rows, dw != nullptr);
  const int blocks = (rows + rpb - 1) / rpb;
  if (dw) hipLaunchKernelGGL(rmsnorm_bwd_kernel<true>, dim3(blocks), dim3(256), 0, st, x, w, g, dx, dw, rows, H, eps, rpb);
  else hipLaunchKernelGGL(rmsnorm_bwd_kernel<false>, dim3(blocks), dim3(256), 0, st, x, w, g, dx, dw, rows, H, eps, rpb);
}

// h = silu(gate) * up with gate_up = [gate | up]:  d gate = gh * up * s (1 + gate (1 - s)),  d up = gh * gate * s
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gate_up, const bf16_t* __restrict__ gh,
                                                         bf16_t* __restrict__ dgu, long long M, int I) {
  const int cpr = I >> 3;  // 8-column chunks per row
  const long long total = M * cpr;
  for (long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long m = t / cpr;
    const int c = static_cast<int>(t % cpr);
    const size_t row2 = static_cast<size_t>(m) * 2 * I;
    const u32x4_t gv = load16(gate_up + row2 + c * 8);
    const u32x4_t uv = load16(gate_up + row2 + I + c * 8);
    const u32x4_t hv = load16(gh + static_cast<size_t>(m) * I + c * 8);
    u32x4_t og, ou;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float dgv[2], duv[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float gt = k ? bf16hi(gv[j]) : bf16lo(gv[j]);
        const float up = k ? bf16hi(uv[j]) : bf16lo(uv[j]);
        const float go = k ? bf16hi(hv[j]) : bf16lo(hv[j]);
        const float s = 1.f / (1.f + __expf(-gt));
        dgv[k] = go * up * s * (1.f + gt * (1.f - s));
        duv[k] = go * gt * s;
      }
      og[j] = pack_bf16x2(dgv[0], dgv[1]);
      ou[j] = pack_bf16x2(duv[0], duv[1]);
    }
    *reinterpret_cast<u32x4_t*>(dgu + row2 + c * 8) = og;
    *reinterpret_cast<u32x4_t*>(dgu + row2 + I + c * 8) = ou;
  }
}

void launch_swiglu_bwd(const bf16_t* gate_up, const bf16_t* gh, bf16_t* dgu, long long M, int I, hipStream_t st) {
  const long long total = M * (I >> 3);
  if (total <= 0) return;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(static_cast<unsigned int>(blocks)), dim3(256), 0, st, gate_up, gh, dgu, M, I);
}

}  // namespace evgpt
