// Token cross-entropy on bf16 logits without an fp32 copy of them, and the row gather / scatter that
// restricts the LM head to supervised positions.
//
// Instruction tuning supervises the answer tokens only -- a few dozen of a sample's ~840 positions
// (reference loss: pipeline/adapter_train/train_lora_adapter.py:692-707; the HF model it trains
// computes logits for every position and an fp32 softmax over all of them).  Here the supervised
// rows of the final hidden states are gathered first (rows_gather; backward rows_scatter), the LM
// head GEMM runs on those rows, and the loss is three small kernels:
//   ce_fwd     per row: log-sum-exp of the bf16 logits in one pass (running max / sum per thread,
//              merged across the block) and  loss = lse - logit[label];
//   ce_reduce  mean over the rows whose label is not the ignore index, and 1 / count;
//   ce_bwd     d logits = (exp(logit - lse) - [column == label]) * upstream / count, written into a
//              row-padded buffer (columns rounded up to 64, zeros in the padding) so that the
//              input-gradient GEMM reads 16-byte chunks; the upstream gradient and 1 / count are read
//              from device memory, so nothing here waits for the host.
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

__device__ __forceinline__ void lse_merge(float& m, float& s, float om, float os) {
  const float nm = fmaxf(m, om);
  if (nm == -INFINITY) { m = nm; s = 0.f; return; }
  s = s * __expf(m - nm) + os * __expf(om - nm);
  m = nm;
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const bf16_t* __restrict__ logits, long long ld,
                                                     const int* __restrict__ labels, float* __restrict__ lse,
                                                     float* __restrict__ loss, int V, int ignore) {
  __shared__ float sm[8], ss[8];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bf16_t* lr = logits + row * ld;
  float m = -INFINITY, s = 0.f;
  for (int v = tid; v < V; v += 256) {
    const float x = bf16_to_f32(lr[v]);
    if (x > m) { s = s * __expf(m - x) + 1.f; m = x; }      // (exp(-inf - x) = 0 on the first element)
    else s += __expf(x - m);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) lse_merge(m, s, __shfl_xor(m, off, 64), __shfl_xor(s, off, 64));
  if (lane == 0) { sm[wid] = m; ss[wid] = s; }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 4; ++i) lse_merge(m, s, sm[i], ss[i]);
    const float l = m + __logf(s);
    lse[row] = l;
    const int lab = labels[row];
    loss[row] = (lab == ignore || lab < 0 || lab >= V) ? 0.f : l - bf16_to_f32(lr[lab]);
  }
}

// stat[0] = mean loss over the counted rows, stat[1] = 1 / max(count, 1); with row weights
// stat[0] = sum_i w_i loss_i and stat[1] = 1 (the weights carry the normalisation); one block
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ loss, const int* __restrict__ labels,
                                                        const float* __restrict__ wrow, int M, int V, int ignore,
                                                        float* __restrict__ stat) {
  __shared__ float red[16];
  float s = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) {
    const int lab = labels[i];
    if (lab != ignore && lab >= 0 && lab < V) { s += wrow ? loss[i] * wrow[i] : loss[i]; c += 1.f; }
  }
  s = block_sum(s, red);
  c = block_sum(c, red);
  if (threadIdx.x == 0) {
    const float inv = wrow ? 1.f : 1.f / fmaxf(c, 1.f);
    stat[0] = s * inv;
    stat[1] = inv;
  }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const bf16_t* __restrict__ logits, long long ld,
                                                     const int* __restrict__ labels, const float* __restrict__ lse,
                                                     const float* __restrict__ upstream, const float* __restrict__ stat,
                                                     const float* __restrict__ wrow, bf16_t* __restrict__ dlogits,
                                                     long long ldd, int V, int ignore) {
  const int row = blockIdx.x, tid = threadIdx.x;
  const int lab = labels[row];
  const bool live = !(lab == ignore || lab < 0 || lab >= V);
  const float scale = live ? upstream[0] * stat[1] * (wrow ? wrow[row] : 1.f) : 0.f;
  const float l = lse[row];
  const bf16_t* lr = logits + row * ld;
  unsigned int* dr = reinterpret_cast<unsigned int*>(dlogits + row * ldd);    // ldd is even: pairs are aligned
  const int pairs = static_cast<int>(ldd >> 1);
  for (int i = tid; i < pairs; i += 256) {
    float d[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = 2 * i + k;
      d[k] = (v < V && live) ? (__expf(bf16_to_f32(lr[v]) - l) - (v == lab ? 1.f : 0.f)) * scale : 0.f;
    }
    dr[i] = pack_bf16x2(d[0], d[1]);
  }
}

void launch_ce_fwd(const bf16_t* logits, long long ld, const int* labels, float* lse, float* loss, float* stat, int M,
                   int V, int ignore, hipStream_t st, const float* wrow) {
  if (M <= 0) return;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(M), dim3(256), 0, st, logits, ld, labels, lse, loss, V, ignore);
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, st, loss, labels, wrow, M, V, ignore, stat);
}
void launch_ce_bwd(const bf16_t* logits, long long ld, const int* labels, const float* lse, const float* upstream,
                   const float* stat, bf16_t* dlogits, long long ldd, int M, int V, int ignore, hipStream_t st,
                   const float* wrow) {
  if (M <= 0) return;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(M), dim3(256), 0, st, logits, ld, labels, lse, upstream, stat, wrow, dlogits,
                     ldd, V, ignore);
}

// out[i] = src[idx[i]] (rows of K bf16, K % 8 == 0)
__global__ __launch_bounds__(256) void rows_gather_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx,
                                                          bf16_t* __restrict__ out, int n, int K, int n_src) {
  const int cpr = K >> 3;
  for (long long t = blockIdx.x * 256ll + threadIdx.x; t < static_cast<long long>(n) * cpr; t += gridDim.x * 256ll) {
    const int i = static_cast<int>(t / cpr), c = static_cast<int>(t % cpr);
    int r = idx[i];
    r = r < 0 ? 0 : (r >= n_src ? n_src - 1 : r);
    *reinterpret_cast<u32x4_t*>(out + static_cast<size_t>(i) * K + c * 8) = load16(src + static_cast<size_t>(r) * K + c * 8);
  }
}
// out [R][K]: row idx[i] = g[i], every other row zero; idx ascending and unique
__global__ __launch_bounds__(256) void rows_scatter_kernel(const bf16_t* __restrict__ g, const int* __restrict__ idx,
                                                           bf16_t* __restrict__ out, int n, int K, int R) {
  const int cpr = K >> 3;
  for (int r = blockIdx.x; r < R; r += gridDim.x) {
    int lo = 0, hi = n;               // first position with idx >= r
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (idx[mid] < r) lo = mid + 1; else hi = mid;
    }
    const bool hit = lo < n && idx[lo] == r;
    for (int c = threadIdx.x; c < cpr; c += 256)
      *reinterpret_cast<u32x4_t*>(out + static_cast<size_t>(r) * K + c * 8) =
          hit ? load16(g + static_cast<size_t>(lo) * K + c * 8) : u32x4_t{0, 0, 0, 0};
  }
}
void launch_rows_gather(const bf16_t* src, const int* idx, bf16_t* out, int n, int K, int n_src, hipStream_t st) {
  if (n <= 0) return;
  long long blocks = (static_cast<long long>(n) * (K >> 3) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rows_gather_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, src, idx, out, n, K, n_src);
}
void launch_rows_scatter(const bf16_t* g, const int* idx, bf16_t* out, int n, int K, int R, hipStream_t st) {
  if (R <= 0) return;
  hipLaunchKernelGGL(rows_scatter_kernel, dim3(R > 4096 ? 4096 : R), dim3(256), 0, st, g, idx, out, n, K, R);
}

}  // namespace evgpt
