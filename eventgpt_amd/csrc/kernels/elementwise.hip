// Memory-bound row kernels: RMSNorm (+residual), LayerNorm, RoPE + KV-cache
// append for prefill, SwiGLU, spatio-temporal pooling, argmax / token commit.
// All bf16 traffic is 16 bytes per lane; arithmetic is fp32.
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

// ---------------------------------------------------------------- RMSNorm
// One 256-thread block per row.  Optional fused residual: r = x + res is written
// to res_out (bf16-rounded, the value later layers see) and normalised.
template <bool HAS_RES>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x,
                                                      const bf16_t* __restrict__ res,
                                                      const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ y,
                                                      bf16_t* __restrict__ res_out, int H,
                                                      float eps) {
  __shared__ float red[16];
  const size_t base = static_cast<size_t>(blockIdx.x) * H;
  const int nchunk = H >> 3;
  float ss = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    u32x4_t v = load16(x + base + c * 8);
    if constexpr (HAS_RES) {
      u32x4_t r = load16(res + base + c * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[i] = pack_bf16x2(bf16lo(v[i]) + bf16lo(r[i]), bf16hi(v[i]) + bf16hi(r[i]));
      *reinterpret_cast<u32x4_t*>(res_out + base + c * 8) = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float lo = bf16lo(v[i]), hi = bf16hi(v[i]);
      ss += lo * lo + hi * hi;
    }
  }
  ss = block_sum(ss, red);
  const float rstd = rsqrtf(ss / static_cast<float>(H) + eps);
  const bf16_t* src = HAS_RES ? res_out : x;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    u32x4_t v = load16(src + base + c * 8);
    u32x4_t g = load16(w + c * 8);
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float lo = bf16_to_f32(f32_to_bf16(bf16lo(v[i]) * rstd)) * bf16lo(g[i]);
      const float hi = bf16_to_f32(f32_to_bf16(bf16hi(v[i]) * rstd)) * bf16hi(g[i]);
      o[i] = pack_bf16x2(lo, hi);
    }
    *reinterpret_cast<u32x4_t*>(y + base + c * 8) = o;
  }
}

void launch_rmsnorm(const bf16_t* x, const bf16_t* res, const bf16_t* w, bf16_t* y,
                    bf16_t* res_out, int rows, int H, float eps, hipStream_t st) {
  if (res) hipLaunchKernelGGL(rmsnorm_kernel<true>, dim3(rows), dim3(256), 0, st, x, res, w, y, res_out, H, eps);
  else hipLaunchKernelGGL(rmsnorm_kernel<false>, dim3(rows), dim3(256), 0, st, x, res, w, y, res_out, H, eps);
}

// --------------------------------------------------------------- LayerNorm
// `act` (ACT_*) is applied to the normalised output (the adapters' LN -> GELU).
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x,
                                                        const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ b,
                                                        bf16_t* __restrict__ y, int H, float eps, int act) {
  __shared__ float red[16];
  const size_t base = static_cast<size_t>(blockIdx.x) * H;
  const int nchunk = H >> 3;
  float s = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    u32x4_t v = load16(x + base + c * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) s += bf16lo(v[i]) + bf16hi(v[i]);
  }
  const float mean = block_sum(s, red) / static_cast<float>(H);
  float ss = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    u32x4_t v = load16(x + base + c * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float lo = bf16lo(v[i]) - mean, hi = bf16hi(v[i]) - mean;
      ss += lo * lo + hi * hi;
    }
  }
  const float rstd = rsqrtf(block_sum(ss, red) / static_cast<float>(H) + eps);
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    u32x4_t v = load16(x + base + c * 8);
    u32x4_t g = load16(w + c * 8);
    u32x4_t bb = b ? load16(b + c * 8) : u32x4_t{0, 0, 0, 0};
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      o[i] = pack_bf16x2(apply_act((bf16lo(v[i]) - mean) * rstd * bf16lo(g[i]) + bf16lo(bb[i]), act),
                         apply_act((bf16hi(v[i]) - mean) * rstd * bf16hi(g[i]) + bf16hi(bb[i]), act));
    *reinterpret_cast<u32x4_t*>(y + base + c * 8) = o;
  }
}

// Rows of up to 64*8*NC elements: one wave per row, the row stays in registers between
// the statistics and the normalisation, and the only synchronisation is the wave
// reduction — a short row (a ViT's 1024) is a single memory round trip instead of
// three passes with block barriers.  Same arithmetic as the block kernel.
template <int NC>
__global__ __launch_bounds__(256) void layernorm_wave_kernel(const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ b,
                                                             bf16_t* __restrict__ y, int rows, int H,
                                                             float eps, int act) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const size_t base = static_cast<size_t>(r) * H;
  const int nchunk = H >> 3;
  u32x4_t v[NC], g[NC], bb[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    const bool ok = c < nchunk;
    const int cc = ok ? c : 0;
    v[i] = load16(x + base + cc * 8);
    g[i] = load16(w + cc * 8);
    bb[i] = b ? load16(b + cc * 8) : u32x4_t{0, 0, 0, 0};
    if (!ok) v[i] = u32x4_t{0, 0, 0, 0};
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += bf16lo(v[i][j]) + bf16hi(v[i][j]);
  const float mean = wave_sum(s) / static_cast<float>(H);
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
  const float rstd = rsqrtf(wave_sum(ss) / static_cast<float>(H) + eps);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack_bf16x2(apply_act((bf16lo(v[i][j]) - mean) * rstd * bf16lo(g[i][j]) + bf16lo(bb[i][j]), act),
                           apply_act((bf16hi(v[i][j]) - mean) * rstd * bf16hi(g[i][j]) + bf16hi(bb[i][j]), act));
      *reinterpret_cast<u32x4_t*>(y + base + c * 8) = o;
    }
  }
}

void launch_layernorm(const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, int rows,
                      int H, float eps, hipStream_t st, int act) {
  const int nchunk = H >> 3;
  if (rows >= 64 && nchunk <= 128) {
    hipLaunchKernelGGL(layernorm_wave_kernel<2>, dim3((rows + 3) / 4), dim3(256), 0, st, x, w, b, y,
                       rows, H, eps, act);
  } else if (rows >= 64 && nchunk <= 256) {
    hipLaunchKernelGGL(layernorm_wave_kernel<4>, dim3((rows + 3) / 4), dim3(256), 0, st, x, w, b, y,
                       rows, H, eps, act);
  } else {
    hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(256), 0, st, x, w, b, y, H, eps, act);
  }
}

// ------------------------------------------------- RoPE + KV append (prefill)
// qkv: [B][S][(Hq+2Hkv)*D] straight out of the fused QKV GEMM.  One thread
// handles one 8-element chunk of the first half of a head and its partner chunk
// in the second half (rotate-half pairing i <-> i + D/2).  q goes to q_out
// [B][S][Hq*D]; k, v go into the cache [B][Hkv][Lmax][D] at pos0[b] + s.
__global__ __launch_bounds__(256) void rope_kv_prefill_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ q_out, bf16_t* __restrict__ k_cache,
    bf16_t* __restrict__ v_cache, const float* __restrict__ cos_tab,
    const float* __restrict__ sin_tab, const int* __restrict__ pos0, int S, int Hq, int Hkv,
    int D, int l_max) {
  const int hh = D >> 1, cph = hh >> 3;             // chunks per half head
  const int heads = Hq + 2 * Hkv;
  const int per_tok = heads * cph;
  const long long total = static_cast<long long>(gridDim.y) * S * per_tok;
  const int b = blockIdx.y;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < static_cast<long long>(S) * per_tok;
       i += gridDim.x * 256ll) {
    const int s = static_cast<int>(i / per_tok);
    const int rem = static_cast<int>(i % per_tok);
    const int h = rem / cph, c = rem % cph;
    int pos = (pos0 ? pos0[b] : 0) + s;
    // positions come from a device counter: a row that would land outside the
    // cache is dropped (its q row still gets a finite value from the last slot)
    const bool in_cache = pos >= 0 && pos < l_max;
    pos = pos < 0 ? 0 : (pos >= l_max ? l_max - 1 : pos);
    const bf16_t* src = qkv + (static_cast<size_t>(b) * S + s) * heads * D + h * D + c * 8;
    u32x4_t a = load16(src), bb = load16(src + hh);
    bf16_t* dst;
    if (h >= Hq && !in_cache) continue;
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
  (void)total;
}

void launch_rope_kv_prefill(const bf16_t* qkv, bf16_t* q_out, bf16_t* k_cache, bf16_t* v_cache,
                            const float* cos_tab, const float* sin_tab, const int* pos0, int B,
                            int S, int Hq, int Hkv, int D, int l_max, hipStream_t st) {
  const long long work = static_cast<long long>(S) * (Hq + 2 * Hkv) * (D / 16);
  int blocks = static_cast<int>((work + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(rope_kv_prefill_kernel, dim3(blocks, B), dim3(256), 0, st, qkv, q_out,
                     k_cache, v_cache, cos_tab, sin_tab, pos0, S, Hq, Hkv, D, l_max);
}

// RoPE on a standalone [B,S,H,D] tensor with explicit positions (module path).
__global__ __launch_bounds__(256) void rope_kernel(const bf16_t* __restrict__ x,
                                                   bf16_t* __restrict__ y,
                                                   const long long* __restrict__ positions,
                                                   const float* __restrict__ cos_tab,
                                                   const float* __restrict__ sin_tab,
                                                   long long tokens, int H, int D) {
  const int hh = D >> 1, cph = hh >> 3;
  const long long total = tokens * H * cph;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const long long t = i / (H * cph);
    const int rem = static_cast<int>(i % (H * cph));
    const int h = rem / cph, c = rem % cph;
    const long long pos = positions[t];
    const size_t off = (static_cast<size_t>(t) * H + h) * D + c * 8;
    u32x4_t a = load16(x + off), bb = load16(x + off + hh);
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
    *reinterpret_cast<u32x4_t*>(y + off) = o1;
    *reinterpret_cast<u32x4_t*>(y + off + hh) = o2;
  }
}

void launch_rope(const bf16_t* x, bf16_t* y, const long long* positions, const float* cos_tab,
                 const float* sin_tab, long long tokens, int H, int D, hipStream_t st) {
  const long long work = tokens * H * (D / 16);
  int blocks = static_cast<int>((work + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(rope_kernel, dim3(blocks), dim3(256), 0, st, x, y, positions, cos_tab,
                     sin_tab, tokens, H, D);
}

// RoPE + split of a packed [tokens][(Hq+2Hkv)*D] projection into q [tokens][Hq][D], k and v
// [tokens][Hkv][D] (training forward over a grouped q/k/v matrix), and its inverse for the
// backward pass (MERGE: the three gradients are rotated back -- `sign` = -1 -- and written
// into one packed gradient).  One launch either way instead of two RoPE launches plus three
// strided copies / a concatenation.
template <bool MERGE>
__global__ __launch_bounds__(256) void rope_qkv_kernel(
    const bf16_t* __restrict__ packed_in, bf16_t* __restrict__ packed_out, bf16_t* q, bf16_t* k, bf16_t* v,
    const long long* __restrict__ positions, const float* __restrict__ cos_tab,
    const float* __restrict__ sin_tab, long long tokens, int Hq, int Hkv, int D, int max_pos, float sign) {
  const int hh = D >> 1, cph = hh >> 3;
  const int heads = Hq + 2 * Hkv, per_tok = heads * cph;
  const long long total = tokens * per_tok;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const long long t = i / per_tok;
    const int rem = static_cast<int>(i % per_tok);
    const int h = rem / cph, c = rem % cph;
    const size_t po = (static_cast<size_t>(t) * heads + h) * D + c * 8;
    bf16_t* sp;
    if (h < Hq) sp = q + (static_cast<size_t>(t) * Hq + h) * D + c * 8;
    else if (h < Hq + Hkv) sp = k + (static_cast<size_t>(t) * Hkv + (h - Hq)) * D + c * 8;
    else sp = v + (static_cast<size_t>(t) * Hkv + (h - Hq - Hkv)) * D + c * 8;
    const bf16_t* src = MERGE ? sp : packed_in + po;
    bf16_t* dst = MERGE ? packed_out + po : sp;
    const u32x4_t a = load16(src), bb = load16(src + hh);
    if (h >= Hq + Hkv) {   // v: copied
      *reinterpret_cast<u32x4_t*>(dst) = a;
      *reinterpret_cast<u32x4_t*>(dst + hh) = bb;
      continue;
    }
    long long pos = positions[t];
    pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
    const float* cs = cos_tab + static_cast<size_t>(pos) * hh + c * 8;
    const float* sn = sin_tab + static_cast<size_t>(pos) * hh + c * 8;
    u32x4_t o1, o2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float c0 = cs[2 * j], c1 = cs[2 * j + 1], s0 = sign * sn[2 * j], s1 = sign * sn[2 * j + 1];
      const float x1l = bf16lo(a[j]), x1h = bf16hi(a[j]), x2l = bf16lo(bb[j]), x2h = bf16hi(bb[j]);
      o1[j] = pack_bf16x2(x1l * c0 - x2l * s0, x1h * c1 - x2h * s1);
      o2[j] = pack_bf16x2(x2l * c0 + x1l * s0, x2h * c1 + x1h * s1);
    }
    *reinterpret_cast<u32x4_t*>(dst) = o1;
    *reinterpret_cast<u32x4_t*>(dst + hh) = o2;
  }
}

void launch_rope_qkv(const bf16_t* packed_in, bf16_t* packed_out, bf16_t* q, bf16_t* k, bf16_t* v,
                     const long long* positions, const float* cos_tab, const float* sin_tab, long long tokens,
                     int Hq, int Hkv, int D, int max_pos, bool merge, hipStream_t st) {
  const long long work = tokens * (Hq + 2 * Hkv) * (D / 16);
  int blocks = static_cast<int>((work + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (merge)
    hipLaunchKernelGGL(rope_qkv_kernel<true>, dim3(blocks), dim3(256), 0, st, packed_in, packed_out, q, k, v,
                       positions, cos_tab, sin_tab, tokens, Hq, Hkv, D, max_pos, -1.f);
  else
    hipLaunchKernelGGL(rope_qkv_kernel<false>, dim3(blocks), dim3(256), 0, st, packed_in, packed_out, q, k, v,
                       positions, cos_tab, sin_tab, tokens, Hq, Hkv, D, max_pos, 1.f);
}

// ------------------------------------------------------------------ SwiGLU
// in: [M][2I] laid out [gate | up]; out: [M][I]
__global__ __launch_bounds__(256) void swiglu_kernel(const bf16_t* __restrict__ in,
                                                     bf16_t* __restrict__ out, long long M, int I) {
  const int cpr = I >> 3;
  const long long total = M * cpr;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const long long m = i / cpr;
    const int c = static_cast<int>(i % cpr);
    u32x4_t g = load16(in + m * 2 * I + c * 8);
    u32x4_t u = load16(in + m * 2 * I + I + c * 8);
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack_bf16x2(silu_f(bf16lo(g[j])) * bf16lo(u[j]), silu_f(bf16hi(g[j])) * bf16hi(u[j]));
    *reinterpret_cast<u32x4_t*>(out + m * I + c * 8) = o;
  }
}

void launch_swiglu(const bf16_t* in, bf16_t* out, long long M, int I, hipStream_t st) {
  const long long work = M * (I / 8);
  int blocks = static_cast<int>((work + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(swiglu_kernel, dim3(blocks), dim3(256), 0, st, in, out, M, I);
}

// ------------------------------------------------- spatio-temporal pooling
// feats [T][S][C] -> out [(Tn + S)][C]: rows 0..Tn-1 are per-frame means over S
// (zero rows if Tn > T), rows Tn.. are per-token means over T.  Both reductions
// come out of one launch without atomics: the first `sp_blocks` blocks give one
// thread per (token, 8-channel chunk) summing over the T frames; the remaining
// blocks each own (frame, 32 chunks) and split the S-long sum over 8 row lanes
// that combine through LDS.
__global__ __launch_bounds__(256) void st_pool_kernel(const bf16_t* __restrict__ f,
                                                      bf16_t* __restrict__ out, int T, int S,
                                                      int C, int Tn, int sp_blocks) {
  const int cpr = C >> 3;
  if (static_cast<int>(blockIdx.x) < sp_blocks) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < S * cpr; i += sp_blocks * 256) {
      const int s = i / cpr, c = i % cpr;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int t = 0; t < T; ++t) {
        u32x4_t v = load16(f + (static_cast<size_t>(t) * S + s) * C + c * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[2 * j] += bf16lo(v[j]); acc[2 * j + 1] += bf16hi(v[j]); }
      }
      const float inv = 1.f / static_cast<float>(T);
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(acc[2 * j] * inv, acc[2 * j + 1] * inv);
      *reinterpret_cast<u32x4_t*>(out + static_cast<size_t>(Tn + s) * C + c * 8) = o;
    }
    return;
  }
  __shared__ float part[8][32][8];
  const int groups = (cpr + 31) / 32;
  const int tb = blockIdx.x - sp_blocks;
  const int t = tb / groups, cg = tb % groups;
  const int sl = threadIdx.x >> 5, c = cg * 32 + (threadIdx.x & 31);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (t < T && c < cpr) {
    // a row lane walks S/8 rows: eight loads are issued before the first is summed, or the walk
    // is one memory round trip per row (26 us for 577 rows)
    constexpr int U = 8;
    int s = sl;
    for (; s + 8 * (U - 1) < S; s += 8 * U) {
      u32x4_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = load16(f + (static_cast<size_t>(t) * S + s + 8 * u) * C + c * 8);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[2 * j] += bf16lo(v[u][j]); acc[2 * j + 1] += bf16hi(v[u][j]); }
    }
    for (; s < S; s += 8) {
      u32x4_t v = load16(f + (static_cast<size_t>(t) * S + s) * C + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[2 * j] += bf16lo(v[j]); acc[2 * j + 1] += bf16hi(v[j]); }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[sl][threadIdx.x & 31][j] = acc[j];
  __syncthreads();
  if (sl == 0 && c < cpr) {
    const float inv = 1.f / static_cast<float>(S);
    float tot[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      tot[j] = 0.f;
      for (int k = 0; k < 8; ++k) tot[j] += part[k][threadIdx.x & 31][j];
    }
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(tot[2 * j] * inv, tot[2 * j + 1] * inv);
    *reinterpret_cast<u32x4_t*>(out + static_cast<size_t>(t) * C + c * 8) = o;
  }
}

void launch_st_pool(const bf16_t* f, bf16_t* out, int T, int S, int C, int Tn, hipStream_t st) {
  const int cpr = C / 8;
  int sp_blocks = (S * cpr + 255) / 256;
  if (sp_blocks > 2048) sp_blocks = 2048;
  const int t_blocks = Tn * ((cpr + 31) / 32);
  hipLaunchKernelGGL(st_pool_kernel, dim3(sp_blocks + t_blocks), dim3(256), 0, st, f, out, T, S, C,
                     Tn, sp_blocks);
}

// ----------------------------------------------------------------- argmax
// Row-wise argmax of an fp32 or bf16 [rows][N] matrix, lowest index on ties.
template <typename T>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const T* __restrict__ x,
                                                          long long* __restrict__ out, int N) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const T* row = x + static_cast<size_t>(blockIdx.x) * N;
  float bv = -INFINITY;
  int bi = 0;
  for (int i = threadIdx.x; i < N; i += 256) {
    float v;
    if constexpr (sizeof(T) == 2) v = bf16_to_f32(row[i]); else v = row[i];
    if (v > bv) { bv = v; bi = i; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
    out[blockIdx.x] = bi;
  }
}

void launch_argmax_rows_f32(const float* x, long long* out, int rows, int N, hipStream_t st) {
  hipLaunchKernelGGL(argmax_rows_kernel<float>, dim3(rows), dim3(256), 0, st, x, out, N);
}
void launch_argmax_rows_bf16(const bf16_t* x, long long* out, int rows, int N, hipStream_t st) {
  hipLaunchKernelGGL(argmax_rows_kernel<bf16_t>, dim3(rows), dim3(256), 0, st, x, out, N);
}

// ------------------------------------------- decode step: commit the token
// Finishes the lm_head argmax from per-block partials, or takes a token chosen
// by the sampler, then (a) appends it to the output ring at the device-side
// step counter, (b) sets the EOS flag, (c) gathers its embedding row into the
// residual-stream buffer for the next step and (d) advances the position.
// One block per sequence; everything the host would otherwise do between two
// decode steps, so a replayed graph needs no host round trip.  `active` (optional,
// one int per sequence) masks slots that must not advance: a batch scheduler
// admits or parks sequences by editing it between replays.
__global__ __launch_bounds__(256) void commit_token_kernel(
    const float* __restrict__ part_val, const int* __restrict__ part_idx, int n_part,
    const int* __restrict__ sampled, const bf16_t* __restrict__ embed, bf16_t* __restrict__ x,
    int* __restrict__ out_tokens, int max_out, int* __restrict__ step, int* __restrict__ pos,
    int* __restrict__ done, int* __restrict__ cur_token, int eos, int H, int vocab,
    const int* __restrict__ active) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int tok_s;
  const int b = blockIdx.x;
  if (active && !active[b]) return;   // block-uniform
  if (sampled) {
    if (threadIdx.x == 0) tok_s = sampled[b];
  } else {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n_part; i += 256) {
      const float v = part_val[b * n_part + i];
      const int ix = part_idx[b * n_part + i];
      if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w)
        if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
      tok_s = bi;
    }
  }
  __syncthreads();
  int tok = tok_s;
  if (tok < 0 || tok >= vocab) tok = 0;  // all-NaN logits must not turn into a wild gather
  const int st = step[b];
  if (threadIdx.x == 0) {
    if (st < max_out) out_tokens[b * max_out + st] = tok;
    cur_token[b] = tok;
    if (tok == eos) done[b] = 1;
  }
  // embedding row for the next step
  const int nchunk = H >> 3;
  for (int c = threadIdx.x; c < nchunk; c += 256)
    *reinterpret_cast<u32x4_t*>(x + static_cast<size_t>(b) * H + c * 8) =
        load16(embed + static_cast<size_t>(tok) * H + c * 8);
  __syncthreads();
  if (threadIdx.x == 0) {
    step[b] = st + 1;
    pos[b] = pos[b] + 1;
  }
}

void launch_commit_token(const float* part_val, const int* part_idx, int n_part,
                         const int* sampled, const bf16_t* embed, bf16_t* x, int* out_tokens,
                         int max_out, int* step, int* pos, int* done, int* cur_token, int eos,
                         int B, int H, int vocab, hipStream_t st, const int* active) {
  hipLaunchKernelGGL(commit_token_kernel, dim3(B), dim3(256), 0, st, part_val, part_idx, n_part,
                     sampled, embed, x, out_tokens, max_out, step, pos, done, cur_token, eos, H, vocab,
                     active);
}

}  // namespace evgpt

namespace evgpt {

// ------------------------------------------------------------- ViT front end
// patchify: pixels [T][C][S][S] bf16 -> rows [T*G*G][Kp] in the conv weight's
// (c, ph, pw) order, zero-padded from C*P*P to Kp (a multiple of 64) so the
// patch embedding is one MFMA GEMM.  One thread per output element pair.
__global__ __launch_bounds__(256) void patchify_kernel(const bf16_t* __restrict__ px,
                                                       bf16_t* __restrict__ out, int T, int C,
                                                       int S, int P, int Kp) {
  const int G = S / P, K = C * P * P;
  const long long total = static_cast<long long>(T) * G * G * Kp;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int k = static_cast<int>(i % Kp);
    const long long row = i / Kp;
    bf16_t v = 0;
    if (k < K) {
      const int t = static_cast<int>(row / (G * G)), g = static_cast<int>(row % (G * G));
      const int gy = g / G, gx = g % G;
      const int c = k / (P * P), py = (k % (P * P)) / P, pxx = k % P;
      v = px[((static_cast<size_t>(t) * C + c) * S + gy * P + py) * S + gx * P + pxx];
    }
    out[i] = v;
  }
}

void launch_patchify(const bf16_t* px, bf16_t* out, int T, int C, int S, int P, int Kp,
                     hipStream_t st) {
  const long long total = static_cast<long long>(T) * (S / P) * (S / P) * Kp;
  int blocks = static_cast<int>((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(patchify_kernel, dim3(blocks), dim3(256), 0, st, px, out, T, C, S, P, Kp);
}

// embed + pre-LayerNorm: row (t, n) = LN( n == 0 ? cls + pos[0] : tok[t][n-1] + pos[n] ).
// tok: [T][N-1][H] (patch GEMM output), pos: [N][H], out: [T][N][H].
__global__ __launch_bounds__(256) void vit_embed_ln_kernel(
    const bf16_t* __restrict__ tok, const bf16_t* __restrict__ cls, const bf16_t* __restrict__ pos,
    const bf16_t* __restrict__ w, const bf16_t* __restrict__ b, bf16_t* __restrict__ out, int N,
    int H, float eps) {
  __shared__ float red[16];
  __shared__ __attribute__((aligned(16))) float rowbuf[2048];
  const int t = blockIdx.x / N, n = blockIdx.x % N;
  const bf16_t* src = n == 0 ? cls : tok + (static_cast<size_t>(t) * (N - 1) + (n - 1)) * H;
  const bf16_t* pp = pos + static_cast<size_t>(n) * H;
  float s = 0.f;
  for (int i = threadIdx.x; i < H; i += 256) {
    // sum is rounded to bf16 like the separate embedding add it replaces
    const float v = bf16_to_f32(f32_to_bf16(bf16_to_f32(src[i]) + bf16_to_f32(pp[i])));
    rowbuf[i] = v;
    s += v;
  }
  const float mean = block_sum(s, red) / static_cast<float>(H);
  float ss = 0.f;
  for (int i = threadIdx.x; i < H; i += 256) { const float d = rowbuf[i] - mean; ss += d * d; }
  const float rstd = rsqrtf(block_sum(ss, red) / static_cast<float>(H) + eps);
  bf16_t* dst = out + static_cast<size_t>(blockIdx.x) * H;
  for (int i = threadIdx.x; i < H; i += 256)
    dst[i] = f32_to_bf16((rowbuf[i] - mean) * rstd * bf16_to_f32(w[i]) + bf16_to_f32(b[i]));
}

void launch_vit_embed_ln(const bf16_t* tok, const bf16_t* cls, const bf16_t* pos, const bf16_t* w,
                         const bf16_t* b, bf16_t* out, int T, int N, int H, float eps,
                         hipStream_t st) {
  hipLaunchKernelGGL(vit_embed_ln_kernel, dim3(T * N), dim3(256), 0, st, tok, cls, pos, w, b, out,
                     N, H, eps);
}

}  // namespace evgpt
