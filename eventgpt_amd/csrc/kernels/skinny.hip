// Consumers of split-K GEMM partials for the few-row regime (speculative verify,
// batched decode with <= 32 rows).
//
// With a handful of activation rows a projection is weight streaming: the GEMM
// splits K over blocks to fill the chip and leaves fp32 slabs
// partial[s][M][N].  Summing the slabs is a pass over M*N floats anyway, so the
// op that follows the projection rides on that pass instead of being its own
// launch and its own round trip through HBM:
//   * o_proj / down_proj : sum -> + residual -> (RMSNorm of the next block)
//   * gate_up            : sum -> SiLU(gate) * up
//   * qkv                : sum -> RoPE -> q rows / KV-cache append
// Each consumer rounds the summed projection to bf16 first, exactly where the
// unfused GEMM epilogue would have, so fused and unfused paths agree bit for bit.
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

// 8 consecutive columns starting at element e0 of the [M][N] slab, summed over
// the slabs in slab order and rounded to bf16 (returned as 4 packed pairs).
// NS > 0: the slab count is a compile-time constant, so all 2*NS loads are issued
// before the first add (these kernels are a few L2 round trips; a runtime-bound
// loop would take them one after the other).  NS == 0: any count.
template <int NS>
__device__ __forceinline__ u32x4_t sum8_bf16(const float* __restrict__ partial, size_t slab,
                                             int nslab, size_t e0) {
  f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  if constexpr (NS > 0) {
    f32x4_t t0[NS], t1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      t0[s] = *reinterpret_cast<const f32x4_t*>(partial + s * slab + e0);
      t1[s] = *reinterpret_cast<const f32x4_t*>(partial + s * slab + e0 + 4);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      a[0] += t0[s][0]; a[1] += t0[s][1]; a[2] += t0[s][2]; a[3] += t0[s][3];
      b[0] += t1[s][0]; b[1] += t1[s][1]; b[2] += t1[s][2]; b[3] += t1[s][3];
    }
  } else {
    for (int s = 0; s < nslab; ++s) {
      const f32x4_t t0 = *reinterpret_cast<const f32x4_t*>(partial + s * slab + e0);
      const f32x4_t t1 = *reinterpret_cast<const f32x4_t*>(partial + s * slab + e0 + 4);
      a[0] += t0[0]; a[1] += t0[1]; a[2] += t0[2]; a[3] += t0[3];
      b[0] += t1[0]; b[1] += t1[1]; b[2] += t1[2]; b[3] += t1[3];
    }
  }
  return u32x4_t{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]),
                 pack_bf16x2(b[2], b[3])};
}

// expands `LAUNCH(NS)` with NS = nslab when it is 1, 2, 4 or 8, else NS = 0 (runtime count)
#define EVGPT_NSLAB_SWITCH(nslab, LAUNCH) \
  switch (nslab) {                        \
    case 1: LAUNCH(1); break;             \
    case 2: LAUNCH(2); break;             \
    case 4: LAUNCH(4); break;             \
    case 8: LAUNCH(8); break;             \
    default: LAUNCH(0); break;            \
  }

// x_out[m] = bf16(sum) + res[m];  xn_out[m] = RMSNorm(x_out[m]) * norm_w  (if norm_w)
// One block per row.  x_out may alias res.
template <int NS>
__global__ __launch_bounds__(256) void reduce_residual_norm_kernel(
    const float* __restrict__ partial, int nslab, int M, int N, const bf16_t* res, bf16_t* x_out,
    const bf16_t* __restrict__ norm_w, bf16_t* __restrict__ xn_out, float eps) {
  __shared__ float red[16];
  const int m = blockIdx.x;
  const size_t slab = static_cast<size_t>(M) * N, base = static_cast<size_t>(m) * N;
  const int nchunk = N >> 3;
  float ss = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    u32x4_t v = sum8_bf16<NS>(partial, slab, nslab, base + c * 8);
    if (res) {
      const u32x4_t r = load16(res + base + c * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[i] = pack_bf16x2(bf16lo(v[i]) + bf16lo(r[i]), bf16hi(v[i]) + bf16hi(r[i]));
    }
    *reinterpret_cast<u32x4_t*>(x_out + base + c * 8) = v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float lo = bf16lo(v[i]), hi = bf16hi(v[i]);
      ss += lo * lo + hi * hi;
    }
  }
  if (!norm_w) return;  // block-uniform
  ss = block_sum(ss, red);
  const float rstd = rsqrtf(ss / static_cast<float>(N) + eps);
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    const u32x4_t v = load16(x_out + base + c * 8);  // this thread's own stores
    const u32x4_t g = load16(norm_w + c * 8);
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float lo = bf16_to_f32(f32_to_bf16(bf16lo(v[i]) * rstd)) * bf16lo(g[i]);
      const float hi = bf16_to_f32(f32_to_bf16(bf16hi(v[i]) * rstd)) * bf16hi(g[i]);
      o[i] = pack_bf16x2(lo, hi);
    }
    *reinterpret_cast<u32x4_t*>(xn_out + base + c * 8) = o;
  }
}

// x_out[m] = bf16(sum + bias) + res[m];  y_out[m] = LayerNorm(x_out[m]) * ln_w + ln_b  (if ln_w): the
// consumer of a ViT block's out-projection and second MLP projection (bias and residual of the projection,
// then the LayerNorm the next projection reads).  One wave per row of up to 64 * 8 * NC elements; the row
// stays in registers between the statistics and the normalisation (as layernorm_wave_kernel).
template <int NS, int NC>
__global__ __launch_bounds__(256) void reduce_residual_layernorm_kernel(
    const float* __restrict__ partial, int nslab, int M, int N, const bf16_t* __restrict__ bias, const bf16_t* res,
    bf16_t* x_out, const bf16_t* __restrict__ ln_w, const bf16_t* __restrict__ ln_b, bf16_t* __restrict__ y_out,
    float eps) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= M) return;
  const size_t slab = static_cast<size_t>(M) * N, base = static_cast<size_t>(m) * N;
  const int nchunk = N >> 3;
  u32x4_t v[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c >= nchunk) { v[i] = u32x4_t{0, 0, 0, 0}; continue; }
    const size_t e0 = base + c * 8;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int ns = NS ? NS : nslab;
#pragma unroll
    for (int sidx = 0; sidx < (NS ? NS : 1); ++sidx) {
      for (int s2 = sidx; s2 < ns; s2 += (NS ? NS : 1)) {
        const f32x4_t t0 = *reinterpret_cast<const f32x4_t*>(partial + s2 * slab + e0);
        const f32x4_t t1 = *reinterpret_cast<const f32x4_t*>(partial + s2 * slab + e0 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] += t0[j]; a[4 + j] += t1[j]; }
      }
    }
    if (bias) {
      const u32x4_t bv = load16(bias + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[2 * j] += bf16lo(bv[j]); a[2 * j + 1] += bf16hi(bv[j]); }
    }
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(a[2 * j], a[2 * j + 1]);
    if (res) {
      const u32x4_t r = load16(res + e0);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(bf16lo(o[j]) + bf16lo(r[j]), bf16hi(o[j]) + bf16hi(r[j]));
    }
    *reinterpret_cast<u32x4_t*>(x_out + e0) = o;
    v[i] = o;
  }
  if (!ln_w) return;   // wave-uniform
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sm += bf16lo(v[i][j]) + bf16hi(v[i][j]);
  const float mean = wave_sum(sm) / static_cast<float>(N);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    if (lane + 64 * i < nchunk) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = bf16lo(v[i][j]) - mean, hi = bf16hi(v[i][j]) - mean;
        ss += lo * lo + hi * hi;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / static_cast<float>(N) + eps);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c >= nchunk) continue;
    const u32x4_t g = load16(ln_w + c * 8);
    const u32x4_t bb = ln_b ? load16(ln_b + c * 8) : u32x4_t{0, 0, 0, 0};
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack_bf16x2((bf16lo(v[i][j]) - mean) * rstd * bf16lo(g[j]) + bf16lo(bb[j]),
                         (bf16hi(v[i][j]) - mean) * rstd * bf16hi(g[j]) + bf16hi(bb[j]));
    *reinterpret_cast<u32x4_t*>(y_out + base + c * 8) = o;
  }
}

// partial [nslab][M][2I] laid out [gate | up] -> out [M][I]
template <int NS>
__global__ __launch_bounds__(256) void reduce_swiglu_kernel(const float* __restrict__ partial,
                                                            int nslab, long long M, int I,
                                                            bf16_t* __restrict__ out) {
  const int cpr = I >> 3;
  const long long total = M * cpr;
  const size_t slab = static_cast<size_t>(M) * 2 * I;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const long long m = i / cpr;
    const int c = static_cast<int>(i % cpr);
    const u32x4_t g = sum8_bf16<NS>(partial, slab, nslab, static_cast<size_t>(m) * 2 * I + c * 8);
    const u32x4_t u = sum8_bf16<NS>(partial, slab, nslab, static_cast<size_t>(m) * 2 * I + I + c * 8);
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack_bf16x2(silu_f(bf16lo(g[j])) * bf16lo(u[j]), silu_f(bf16hi(g[j])) * bf16hi(u[j]));
    *reinterpret_cast<u32x4_t*>(out + m * I + c * 8) = o;
  }
}

// partial [nslab][B*S][(Hq+2Hkv)*D] -> q_out [B][S][Hq*D], k/v cache rows at pos0[b]+s.
// Thread = one 8-wide chunk of the first half of a head and its rotate-half partner.
template <int NS>
__global__ __launch_bounds__(256) void reduce_rope_kv_kernel(
    const float* __restrict__ partial, int nslab, bf16_t* __restrict__ q_out,
    bf16_t* __restrict__ k_cache, bf16_t* __restrict__ v_cache, const float* __restrict__ cos_tab,
    const float* __restrict__ sin_tab, const int* __restrict__ pos0, int S, int Hq, int Hkv, int D,
    int l_max) {
  const int hh = D >> 1, cph = hh >> 3;
  const int heads = Hq + 2 * Hkv;
  const int per_tok = heads * cph;
  const int b = blockIdx.y, B = gridDim.y;
  const size_t slab = static_cast<size_t>(B) * S * heads * D;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < static_cast<long long>(S) * per_tok;
       i += gridDim.x * 256ll) {
    const int s = static_cast<int>(i / per_tok);
    const int rem = static_cast<int>(i % per_tok);
    const int h = rem / cph, c = rem % cph;
    int pos = (pos0 ? pos0[b] : 0) + s;
    // device-side positions: a row that would land outside the cache is dropped
    const bool in_cache = pos >= 0 && pos < l_max;
    pos = pos < 0 ? 0 : (pos >= l_max ? l_max - 1 : pos);
    if (h >= Hq && !in_cache) continue;
    const size_t src = (static_cast<size_t>(b) * S + s) * heads * D + h * D + c * 8;
    const u32x4_t a = sum8_bf16<NS>(partial, slab, nslab, src);
    const u32x4_t bb = sum8_bf16<NS>(partial, slab, nslab, src + hh);
    bf16_t* dst;
    if (h < Hq) {
      dst = q_out + (static_cast<size_t>(b) * S + s) * Hq * D + h * D;
    } else if (h < Hq + Hkv) {
      dst = k_cache + ((static_cast<size_t>(b) * Hkv + (h - Hq)) * l_max + pos) * D;
    } else {
      dst = v_cache + ((static_cast<size_t>(b) * Hkv + (h - Hq - Hkv)) * l_max + pos) * D;
      *reinterpret_cast<u32x4_t*>(dst + c * 8) = a;
      *reinterpret_cast<u32x4_t*>(dst + hh + c * 8) = bb;
      continue;
    }
    const float* cs = cos_tab + static_cast<size_t>(pos) * hh + c * 8;
    const float* sn = sin_tab + static_cast<size_t>(pos) * hh + c * 8;
    u32x4_t o1, o2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float c0 = cs[2 * j], c1 = cs[2 * j + 1], s0 = sn[2 * j], s1 = sn[2 * j + 1];
      const float x1l = bf16lo(a[j]), x1h = bf16hi(a[j]), x2l = bf16lo(bb[j]), x2h = bf16hi(bb[j]);
      o1[j] = pack_bf16x2(x1l * c0 - x2l * s0, x1h * c1 - x2h * s1);
      o2[j] = pack_bf16x2(x2l * c0 + x1l * s0, x2h * c1 + x1h * s1);
    }
    *reinterpret_cast<u32x4_t*>(dst + c * 8) = o1;
    *reinterpret_cast<u32x4_t*>(dst + hh + c * 8) = o2;
  }
}

// ------------------------------------------------------------- row metrics
// cos[r] = <a_r, b_r> / (|a_r| |b_r|),  mse[r] = mean_j (a_rj - b_rj)^2 for rows of
// length H in ONE pass over both tensors (the evaluation metrics and the alignment
// loss need both; separate normalise / multiply / reduce passes read the pair five
// times).  One wave per row, 16-byte loads, fp32 accumulation.  T = float or bf16.
template <typename T>
__global__ __launch_bounds__(256) void row_metrics_kernel(const T* __restrict__ a,
                                                          const T* __restrict__ b,
                                                          float* __restrict__ cos_out,
                                                          float* __restrict__ mse_out, long long rows,
                                                          int H, float eps, float* __restrict__ stats) {
  const long long r = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  constexpr int EPV = 16 / sizeof(T);   // elements per 16-byte vector
  const T* ar = a + r * H;
  const T* br = b + r * H;
  float dot = 0.f, na = 0.f, nb = 0.f, se = 0.f;
  for (int c = lane; c < H / EPV; c += 64) {
    float x[EPV], y[EPV];
    if constexpr (sizeof(T) == 4) {
      const f32x4_t va = *reinterpret_cast<const f32x4_t*>(ar + c * EPV);
      const f32x4_t vb = *reinterpret_cast<const f32x4_t*>(br + c * EPV);
#pragma unroll
      for (int i = 0; i < 4; ++i) { x[i] = va[i]; y[i] = vb[i]; }
    } else {
      const u32x4_t va = load16(ar + c * EPV), vb = load16(br + c * EPV);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[2 * i] = bf16lo(va[i]); x[2 * i + 1] = bf16hi(va[i]);
        y[2 * i] = bf16lo(vb[i]); y[2 * i + 1] = bf16hi(vb[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < EPV; ++i) {
      dot += x[i] * y[i]; na += x[i] * x[i]; nb += y[i] * y[i];
      const float d = x[i] - y[i];
      se += d * d;
    }
  }
  dot = wave_sum(dot); na = wave_sum(na); nb = wave_sum(nb); se = wave_sum(se);
  if (lane == 0) {
    if (stats) {   // the raw sums: a loss function's backward needs them
      *reinterpret_cast<f32x4_t*>(stats + r * 4) = f32x4_t{dot, na, nb, se};
      return;
    }
    cos_out[r] = dot / (fmaxf(sqrtf(na), eps) * fmaxf(sqrtf(nb), eps));
    mse_out[r] = se / static_cast<float>(H);
  }
}

// Gradient of a loss that is a function of the per-row sums above (masked MSE + cosine):
//   grad[r][i] = A_r (p_i - t_i) + B_r t_i + C_r p_i,   coef[r] = {A_r, B_r, C_r}
// in one pass over the pair (autograd through normalise / multiply / reduce writes and re-reads
// a dozen full-size temporaries).
template <typename T>
__global__ __launch_bounds__(256) void row_affine_grad_kernel(const T* __restrict__ p, const T* __restrict__ t,
                                                              const float* __restrict__ coef, T* __restrict__ grad,
                                                              long long rows, int H) {
  constexpr int EPV = 16 / sizeof(T);
  const int cpr = H / EPV;
  const long long total = rows * cpr;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += gridDim.x * 256ll) {
    const long long r = idx / cpr;
    const float A = coef[r * 3], B = coef[r * 3 + 1], C = coef[r * 3 + 2];
    if constexpr (sizeof(T) == 4) {
      const f32x4_t vp = reinterpret_cast<const f32x4_t*>(p)[idx], vt = reinterpret_cast<const f32x4_t*>(t)[idx];
      f32x4_t g;
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = A * (vp[i] - vt[i]) + B * vt[i] + C * vp[i];
      reinterpret_cast<f32x4_t*>(grad)[idx] = g;
    } else {
      const u32x4_t vp = reinterpret_cast<const u32x4_t*>(p)[idx], vt = reinterpret_cast<const u32x4_t*>(t)[idx];
      u32x4_t g;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pl = bf16lo(vp[i]), ph = bf16hi(vp[i]), tl = bf16lo(vt[i]), th = bf16hi(vt[i]);
        g[i] = pack_bf16x2(A * (pl - tl) + B * tl + C * pl, A * (ph - th) + B * th + C * ph);
      }
      reinterpret_cast<u32x4_t*>(grad)[idx] = g;
    }
  }
}

void launch_row_affine_grad(const void* p, const void* t, const float* coef, void* grad, long long rows, int H,
                            bool bf16, hipStream_t st) {
  const long long total = rows * (H / (bf16 ? 8 : 4));
  int blocks = static_cast<int>((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) return;
  if (bf16)
    hipLaunchKernelGGL(row_affine_grad_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, static_cast<const bf16_t*>(p),
                       static_cast<const bf16_t*>(t), coef, static_cast<bf16_t*>(grad), rows, H);
  else
    hipLaunchKernelGGL(row_affine_grad_kernel<float>, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(p),
                       static_cast<const float*>(t), coef, static_cast<float*>(grad), rows, H);
}

void launch_row_metrics_f32(const float* a, const float* b, float* cos_out, float* mse_out,
                            long long rows, int H, float eps, hipStream_t st, float* stats) {
  hipLaunchKernelGGL(row_metrics_kernel<float>, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256),
                     0, st, a, b, cos_out, mse_out, rows, H, eps, stats);
}
void launch_row_metrics_bf16(const bf16_t* a, const bf16_t* b, float* cos_out, float* mse_out,
                             long long rows, int H, float eps, hipStream_t st, float* stats) {
  hipLaunchKernelGGL(row_metrics_kernel<bf16_t>, dim3(static_cast<unsigned>((rows + 3) / 4)),
                     dim3(256), 0, st, a, b, cos_out, mse_out, rows, H, eps, stats);
}

void launch_reduce_residual_norm(const float* partial, int nslab, int M, int N, const bf16_t* res,
                                 bf16_t* x_out, const bf16_t* norm_w, bf16_t* xn_out, float eps,
                                 hipStream_t st) {
#define L_(NS)                                                                                   \
  hipLaunchKernelGGL(reduce_residual_norm_kernel<NS>, dim3(M), dim3(256), 0, st, partial, nslab, M, \
                     N, res, x_out, norm_w, xn_out, eps)
  EVGPT_NSLAB_SWITCH(nslab, L_)
#undef L_
}

// N <= 2048 (a multiple of 8); returns false for wider rows (nothing launched)
bool launch_reduce_residual_layernorm(const float* partial, int nslab, int M, int N, const bf16_t* bias,
                                      const bf16_t* res, bf16_t* x_out, const bf16_t* ln_w, const bf16_t* ln_b,
                                      bf16_t* y_out, float eps, hipStream_t st) {
  const int nchunk = N >> 3;
  if (nchunk > 256 || M < 1) return false;
#define L_(NS)                                                                                                      \
  do {                                                                                                              \
    if (nchunk <= 128)                                                                                              \
      hipLaunchKernelGGL((reduce_residual_layernorm_kernel<NS, 2>), dim3((M + 3) / 4), dim3(256), 0, st, partial,  \
                         nslab, M, N, bias, res, x_out, ln_w, ln_b, y_out, eps);                                    \
    else                                                                                                            \
      hipLaunchKernelGGL((reduce_residual_layernorm_kernel<NS, 4>), dim3((M + 3) / 4), dim3(256), 0, st, partial,  \
                         nslab, M, N, bias, res, x_out, ln_w, ln_b, y_out, eps);                                    \
  } while (0)
  EVGPT_NSLAB_SWITCH(nslab, L_)
#undef L_
  return true;
}

void launch_reduce_swiglu(const float* partial, int nslab, long long M, int I, bf16_t* out,
                          hipStream_t st) {
  const long long work = M * (I / 8);
  int blocks = static_cast<int>((work + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
#define L_(NS) \
  hipLaunchKernelGGL(reduce_swiglu_kernel<NS>, dim3(blocks), dim3(256), 0, st, partial, nslab, M, I, out)
  EVGPT_NSLAB_SWITCH(nslab, L_)
#undef L_
}

void launch_reduce_rope_kv(const float* partial, int nslab, bf16_t* q_out, bf16_t* k_cache,
                           bf16_t* v_cache, const float* cos_tab, const float* sin_tab,
                           const int* pos0, int B, int S, int Hq, int Hkv, int D, int l_max,
                           hipStream_t st) {
  const long long work = static_cast<long long>(S) * (Hq + 2 * Hkv) * (D / 16);
  int blocks = static_cast<int>((work + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
#define L_(NS)                                                                                  \
  hipLaunchKernelGGL(reduce_rope_kv_kernel<NS>, dim3(blocks, B), dim3(256), 0, st, partial, nslab, \
                     q_out, k_cache, v_cache, cos_tab, sin_tab, pos0, S, Hq, Hkv, D, l_max)
  EVGPT_NSLAB_SWITCH(nslab, L_)
#undef L_
}

}  // namespace evgpt
