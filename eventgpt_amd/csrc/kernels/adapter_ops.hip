// Kernels of the speculative-decoding draft head (hidden-state adapter -> verifier lm_head
// -> argmax) that keep a whole draft/verify iteration on the device:
//   * gather_rows_dev: rows of a [N][H] matrix selected by a DEVICE-resident index (the
//     verifier's step counter, or the number of tokens it accepted last round), so a captured
//     iteration reads "the drafter states aligned with the current token" without the host
//     ever learning where the sequence stands;
//   * argmax_from_partials: finishes the row argmax of the lm_head GEMM's per-slab partials
//     into int32 draft tokens (lowest index on ties);
//   * attn_d512: self-attention for the adapters' 8 x 512 heads.  A head_dim of 512 is exactly
//     one wave's worth of 16-byte lanes (64 x 8 bf16), so a wave owns one query row, keeps q
//     and the output row in registers, and walks the keys with an online softmax; the
//     sequences here are the few draft positions of one round (1..64 rows), for which a
//     tensor-core tile would be almost entirely padding.
// (Reference: the adapters' nn.MultiheadAttention, pipeline/adapter_train/hidden_adapter.py:522-560,
// driven from the SD loop at pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:996-1001.)
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

// out[i][:] = src[clamp(index[0] + offset + i, 0, n_valid - 1)][:],  i < rows
__global__ __launch_bounds__(256) void gather_rows_dev_kernel(const bf16_t* __restrict__ src,
                                                              const int* __restrict__ index, int offset,
                                                              const int* __restrict__ n_valid_dev, int n_rows_src,
                                                              bf16_t* __restrict__ out, int H) {
  const int i = blockIdx.x;
  int nv = n_valid_dev ? n_valid_dev[0] : n_rows_src;
  nv = nv < 1 ? 1 : (nv > n_rows_src ? n_rows_src : nv);
  int r = index[0] + offset + i;
  r = r < 0 ? 0 : (r >= nv ? nv - 1 : r);
  for (int c = threadIdx.x; c < (H >> 3); c += 256)
    *reinterpret_cast<u32x4_t*>(out + static_cast<size_t>(i) * H + c * 8) =
        load16(src + static_cast<size_t>(r) * H + c * 8);
}

void launch_gather_rows_dev(const bf16_t* src, const int* index, int offset, const int* n_valid_dev,
                            int n_rows_src, bf16_t* out, int rows, int H, hipStream_t st) {
  hipLaunchKernelGGL(gather_rows_dev_kernel, dim3(rows), dim3(256), 0, st, src, index, offset, n_valid_dev,
                     n_rows_src, out, H);
}

// token[r] = index of the largest partial of row r (lowest index on ties); one wave per row
__global__ __launch_bounds__(64) void argmax_from_partials_kernel(const float* __restrict__ pval,
                                                                  const int* __restrict__ pidx, int n_part,
                                                                  int* __restrict__ token, int vocab) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < n_part; i += 64) {
    const float v = pval[static_cast<size_t>(r) * n_part + i];
    const int ix = pidx[static_cast<size_t>(r) * n_part + i];
    if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) token[r] = (bi < 0 || bi >= vocab) ? 0 : bi;
}

void launch_argmax_from_partials(const float* pval, const int* pidx, int rows, int n_part, int* token,
                                 int vocab, hipStream_t st) {
  hipLaunchKernelGGL(argmax_from_partials_kernel, dim3(rows), dim3(64), 0, st, pval, pidx, n_part, token, vocab);
}

// Multimodal splice: out[i][:] = embed[map[i]] (map >= 0), feats[-1 - map[i]] (negative) or
// zeros (padding, map <= pad).  One block per output row; the layout itself (which position
// takes which row) is a few dozen integers computed by the host from the prompt ids.
__global__ __launch_bounds__(256) void splice_rows_kernel(const bf16_t* __restrict__ embed, int vocab,
                                                          const bf16_t* __restrict__ feats, int n_feat,
                                                          const int* __restrict__ map, int pad,
                                                          bf16_t* __restrict__ out, int H) {
  const long long i = blockIdx.x;
  const int m = map[i];
  const bf16_t* src = nullptr;
  if (m >= 0) src = embed + static_cast<size_t>(m < vocab ? m : vocab - 1) * H;
  else if (m > pad) { const int r = -1 - m; src = feats + static_cast<size_t>(r < n_feat ? r : n_feat - 1) * H; }
  for (int c = threadIdx.x; c < (H >> 3); c += 256)
    *reinterpret_cast<u32x4_t*>(out + i * H + c * 8) = src ? load16(src + c * 8) : u32x4_t{0, 0, 0, 0};
}

void launch_splice_rows(const bf16_t* embed, int vocab, const bf16_t* feats, int n_feat, const int* map,
                        int pad, bf16_t* out, long long rows, int H, hipStream_t st) {
  hipLaunchKernelGGL(splice_rows_kernel, dim3(static_cast<unsigned>(rows)), dim3(256), 0, st, embed, vocab, feats,
                     n_feat, map, pad, out, H);
}

// q/k/v: element (b, s, h, d) at base + b*sb + s*ss + h*512 + d (any strides with contiguous d,
// e.g. the q / k / v thirds of a packed in-projection); out [B][S][H][512] contiguous.
__global__ __launch_bounds__(256) void attn_d512_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, bf16_t* __restrict__ out,
                                                        long long sb, long long ss, int S, int H, int causal,
                                                        float scale) {
  constexpr int D = 512;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.z;
  const long long bh_row = blockIdx.x * 4ll + wid;           // index inside this batch: h * S + qi
  if (bh_row >= static_cast<long long>(H) * S) return;       // wave-uniform
  const int h = static_cast<int>(bh_row / S), qi = static_cast<int>(bh_row % S);
  const bf16_t* qp = q + b * sb + static_cast<long long>(qi) * ss + h * D + lane * 8;
  const u32x4_t qv = load16(qp);
  const int n_keys = causal ? qi + 1 : S;
  float m = -INFINITY, l = 0.f;
  float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int U = 4;                                       // keys in flight
  for (int j0 = 0; j0 < n_keys; j0 += U) {
    u32x4_t kv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u < n_keys ? j0 + u : n_keys - 1;
      kv[u] = load16(k + b * sb + static_cast<long long>(j) * ss + h * D + lane * 8);
      vv[u] = load16(v + b * sb + static_cast<long long>(j) * ss + h * D + lane * 8);
    }
    float sc[U], mn = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sc[u] = wave_sum(dot8_bf16(qv, kv[u], 0.f)) * scale;
      if (j0 + u < n_keys) mn = fmaxf(mn, sc[u]);
    }
    const float a = (m == -INFINITY) ? 0.f : __expf(m - mn);
    l *= a;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] *= a;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float p = j0 + u < n_keys ? __expf(sc[u] - mn) : 0.f;
      l += p;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[2 * i] += p * bf16lo(vv[u][i]);
        o[2 * i + 1] += p * bf16hi(vv[u][i]);
      }
    }
    m = mn;
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  const u32x4_t w = {pack_bf16x2(o[0] * inv, o[1] * inv), pack_bf16x2(o[2] * inv, o[3] * inv),
                     pack_bf16x2(o[4] * inv, o[5] * inv), pack_bf16x2(o[6] * inv, o[7] * inv)};
  *reinterpret_cast<u32x4_t*>(out + ((static_cast<long long>(b) * S + qi) * H + h) * D + lane * 8) = w;
}

void launch_attn_d512(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, long long sb,
                      long long ss, int B, int S, int H, int causal, float scale, hipStream_t st) {
  const long long rows = static_cast<long long>(H) * S;
  hipLaunchKernelGGL(attn_d512_kernel, dim3(static_cast<unsigned>((rows + 3) / 4), 1, B), dim3(256), 0, st, q, k,
                     v, out, sb, ss, S, H, causal, scale);
}

}  // namespace evgpt
