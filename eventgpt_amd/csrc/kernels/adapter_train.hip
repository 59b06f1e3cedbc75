// Training kernels of the hidden-state adapters (reference modules:
// pipeline/adapter_train/hidden_adapter.py:495-663 AttentionAdapter, :670-958 EAGLEStyleAdapter,
// :965-1199 FusedEAGLEAdapter; there every op is a PyTorch / cuDNN library call):
//
//   * attn512_fwd / attn512_bwd -- multi-head self-attention with head_dim 512 (8 heads over a
//     4096-wide state) for sequences of at most 64 positions, forward and backward on MFMA.  One
//     workgroup owns one (batch, head): the whole score matrix is a single 64 x 64 tile, so the
//     forward keeps the probabilities (8 KB per head) for the backward instead of recomputing
//     them, and nothing is split over keys.  The 512-deep contractions stream through LDS in
//     four 128-column chunks; products that contract over the sequence read their operand with
//     the transposing LDS read.  Key-padding masks, causal masks and dropout on the probabilities
//     (a counter-based hash of (seed, batch, head, row, key), recomputed in the backward) are
//     applied in registers.
//   * layernorm_bwd, gelu / gelu_bwd, dropout -- the elementwise pieces around it.
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

// ------------------------------------------------------------------ counter-based dropout mask
__device__ __forceinline__ unsigned int drop_hash(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return static_cast<unsigned int>((z ^ (z >> 31)) >> 40);   // 24 uniform bits
}
__device__ __forceinline__ bool drop_keep(unsigned long long seed, unsigned long long idx, unsigned int thresh24) {
  return drop_hash(seed, idx) >= thresh24;
}
static unsigned int drop_threshold(float p) {
  if (p <= 0.f) return 0u;
  const double t = static_cast<double>(p) * 16777216.0;
  return t >= 16777215.0 ? 16777215u : static_cast<unsigned int>(t);
}

// ------------------------------------------------------------------------------ LDS tile images
// [64][128] bf16 (256-byte rows): 16-byte chunk c of row r sits at chunk c ^ t128_sw(r).  The XOR
// is even, so a chunk pair stays together (what the transposing read needs), and the 8 rows a
// half-wave gathers at once fall on 8 distinct 32-byte slots.
__device__ __forceinline__ int t128_sw(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }
__device__ __forceinline__ int t128_off(int row, int ch) { return row * 256 + ((ch ^ t128_sw(row)) << 4); }
// [64][64] bf16 (128-byte rows), row fragments only
__device__ __forceinline__ int t64_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

// rows r0 + (lane & 15), k = 32 ks + 8 (lane >> 4) .. + 7 of a [64][128] image (operand whose
// contraction runs along the image's columns)
__device__ __forceinline__ bf16x8_t t128_rows(const unsigned char* img, int r0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(img + t128_off(r0 + (lane & 15), ks * 4 + (lane >> 4)));
}
__device__ __forceinline__ bf16x8_t t64_rows(const unsigned char* img, int r0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(img + t64_off(r0 + (lane & 15), ks * 4 + (lane >> 4)));
}
// columns cb + (lane & 15), k = image rows 32 ks + 8 (lane >> 4) .. + 7 (operand whose contraction
// runs down the image's rows): two transposing reads
__device__ __forceinline__ bf16x8_t t128_cols(const unsigned char* img, int cb, int ks, int lane) {
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const int fr = lane & 15, g = lane >> 4, q = fr >> 2, pp = fr & 3;
  const int row1 = ks * 32 + 8 * g + q, row2 = row1 + 4;
  const int chunk = (cb >> 3) + (pp >> 1);
  const unsigned char* a1 = img + row1 * 256 + ((chunk ^ t128_sw(row1)) << 4) + 8 * (pp & 1);
  const unsigned char* a2 = img + row2 * 256 + ((chunk ^ t128_sw(row2)) << 4) + 8 * (pp & 1);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)a1);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)a2);
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// rows 0..S-1 (zeros below) x columns [c0, c0 + 128) of one head into a [64][128] image
__device__ __forceinline__ void t128_load(unsigned char* img, const bf16_t* base, long long row_stride, int S,
                                          int c0, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * 256;
    const int row = idx >> 4, ch = idx & 15;
    const u32x4_t v = row < S ? load16(base + static_cast<long long>(row) * row_stride + c0 + ch * 8)
                              : u32x4_t{0, 0, 0, 0};
    *reinterpret_cast<u32x4_t*>(img + t128_off(row, ch)) = v;
  }
}

struct Attn512Args {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;   // element (b, s, h, d) at b*sb + s*ss + h*512 + d
  long long sb, ss;
  bf16_t* out;                // forward: [B][S][H][512]
  bf16_t* probs;              // [B][H][64][64] softmax probabilities before dropout (forward writes, backward reads)
  const unsigned char* kpm;   // [B][S], 1 = key is padding; or null
  const bf16_t* d_out;        // backward: [B][S][H][512]
  bf16_t* dq; bf16_t* dk; bf16_t* dv;                   // backward outputs, addressed like q / k / v
  int S, H, causal;
  float scale, inv_keep;
  unsigned int drop_thresh;
  unsigned long long seed;
};

constexpr int A512_D = 512, A512_T = 64;

// ------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(256) void attn512_fwd_kernel(Attn512Args p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384 + 8192];
  unsigned char* imgA = smem;
  unsigned char* imgB = smem + 16384;
  unsigned char* imgP = smem + 32768;
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, g = lane >> 4;
  const int S = p.S;
  const long long head = static_cast<long long>(b) * p.sb + static_cast<long long>(h) * A512_D;

  // scores, transposed: acc[n][r] = Q[16w + fr] . K[16n + 4g + r]
  f32x4_t acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < 4; ++c) {
    t128_load(imgA, p.q + head, p.ss, S, c * 128, tid);
    t128_load(imgB, p.k + head, p.ss, S, c * 128, tid);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t bq = t128_rows(imgA, 16 * w, ks, lane);
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t128_rows(imgB, 16 * n, ks, lane), bq, acc[n], 0, 0, 0);
    }
    __syncthreads();
  }

  // softmax over the keys of query row 16w + fr: in-lane over (n, r), then over the four lane groups
  const int qrow = 16 * w + fr;
  float mx = -INFINITY;
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * n + 4 * g + r;
      const bool dead = key >= S || (p.causal && key > qrow) || (p.kpm && p.kpm[b * S + key]);
      acc[n][r] = dead ? -INFINITY : acc[n][r] * p.scale;
      mx = fmaxf(mx, acc[n][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[n][r] = mx == -INFINITY ? 0.f : __expf(acc[n][r] - mx);
      sum += acc[n][r];
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = (sum > 0.f && qrow < S) ? 1.f / sum : 0.f;
  const unsigned long long row_id = (static_cast<unsigned long long>(b) * p.H + h) * A512_T + qrow;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    float pr[4], pd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pr[r] = bf16_to_f32(f32_to_bf16(acc[n][r] * inv));     // what the backward reads back
      const bool keep = p.drop_thresh == 0 || drop_keep(p.seed, row_id * A512_T + 16 * n + 4 * g + r, p.drop_thresh);
      pd[r] = keep ? pr[r] * p.inv_keep : 0.f;
    }
    const int key0 = 16 * n + 4 * g;
    if (p.probs) {
      const u32x2_t o = {pack_bf16x2(pr[0], pr[1]), pack_bf16x2(pr[2], pr[3])};
      *reinterpret_cast<u32x2_t*>(p.probs + row_id * A512_T + key0) = o;
    }
    const u32x2_t o = {pack_bf16x2(pd[0], pd[1]), pack_bf16x2(pd[2], pd[3])};
    *reinterpret_cast<u32x2_t*>(imgP + t64_off(qrow, key0 >> 3) + (key0 & 7) * 2) = o;
  }
  __syncthreads();

  // out^T tile: o[t][r] = sum_key V[key][128c + 16t + 4g + r] * P[qrow][key]
  for (int c = 0; c < 4; ++c) {
    t128_load(imgA, p.v + head, p.ss, S, c * 128, tid);
    __syncthreads();
    f32x4_t o[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) o[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t bp = t64_rows(imgP, 16 * w, ks, lane);
#pragma unroll
      for (int t = 0; t < 8; ++t)
        o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t128_cols(imgA, 16 * t, ks, lane), bp, o[t], 0, 0, 0);
    }
    if (qrow < S) {
      bf16_t* dst = p.out + ((static_cast<long long>(b) * S + qrow) * p.H + h) * A512_D + c * 128 + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const u32x2_t ov = {pack_bf16x2(o[t][0], o[t][1]), pack_bf16x2(o[t][2], o[t][3])};
        *reinterpret_cast<u32x2_t*>(dst + 16 * t) = ov;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ backward
// dP = dO V^T;  with the dropout mask M / keep:  Pd = P M,  dP' = dP M,  delta = rowsum(P dP'),
// dS = scale * P (dP' - delta);  dV = Pd^T dO,  dQ = dS K,  dK = dS^T Q.
constexpr int A512_BWD_SMEM = 3 * 16384 + 3 * 8192;

__global__ __launch_bounds__(256) void attn512_bwd_kernel(Attn512Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* img0 = smem;
  unsigned char* img1 = smem + 16384;
  unsigned char* img2 = smem + 2 * 16384;
  unsigned char* imgDS = smem + 3 * 16384;           // dS   [qrow][key]
  unsigned char* imgDST = imgDS + 8192;              // dS^T [key][qrow]
  unsigned char* imgPT = imgDST + 8192;              // Pd^T [key][qrow]
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, g = lane >> 4;
  const int S = p.S;
  const long long head = static_cast<long long>(b) * p.sb + static_cast<long long>(h) * A512_D;
  const bf16_t* go = p.d_out + (static_cast<long long>(b) * S * p.H + h) * A512_D;
  const long long go_stride = static_cast<long long>(p.H) * A512_D;

  // dP, laid out like the forward's scores: acc[n][r] = dO[16w + fr] . V[16n + 4g + r]
  f32x4_t acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < 4; ++c) {
    t128_load(img0, p.v + head, p.ss, S, c * 128, tid);
    t128_load(img1, go, go_stride, S, c * 128, tid);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t bg = t128_rows(img1, 16 * w, ks, lane);
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t128_rows(img0, 16 * n, ks, lane), bg, acc[n], 0, 0, 0);
    }
    __syncthreads();
  }

  const int qrow = 16 * w + fr;
  const unsigned long long row_id = (static_cast<unsigned long long>(b) * p.H + h) * A512_T + qrow;
  float pr[4][4], dpm[4][4], keepf[4][4];
  float delta = 0.f;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int key0 = 16 * n + 4 * g;
    const u32x2_t pv = *reinterpret_cast<const u32x2_t*>(p.probs + row_id * A512_T + key0);
    pr[n][0] = bf16lo(pv[0]); pr[n][1] = bf16hi(pv[0]); pr[n][2] = bf16lo(pv[1]); pr[n][3] = bf16hi(pv[1]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool keep = p.drop_thresh == 0 || drop_keep(p.seed, row_id * A512_T + key0 + r, p.drop_thresh);
      keepf[n][r] = keep ? p.inv_keep : 0.f;
      dpm[n][r] = acc[n][r] * keepf[n][r];
      delta += pr[n][r] * dpm[n][r];
    }
  }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int key0 = 16 * n + 4 * g;
    float ds[4];
    bf16_t dsb[4], pdb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ds[r] = pr[n][r] * (dpm[n][r] - delta) * p.scale;
      dsb[r] = f32_to_bf16(ds[r]);
      pdb[r] = f32_to_bf16(pr[n][r] * keepf[n][r]);
    }
    const u32x2_t o = {pack_bf16x2(ds[0], ds[1]), pack_bf16x2(ds[2], ds[3])};
    *reinterpret_cast<u32x2_t*>(imgDS + t64_off(qrow, key0 >> 3) + (key0 & 7) * 2) = o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = key0 + r;
      *reinterpret_cast<bf16_t*>(imgDST + t64_off(key, qrow >> 3) + (qrow & 7) * 2) = dsb[r];
      *reinterpret_cast<bf16_t*>(imgPT + t64_off(key, qrow >> 3) + (qrow & 7) * 2) = pdb[r];
    }
  }
  __syncthreads();

  // wave w owns sequence positions 16w .. 16w + 15 of all three outputs; o[t][r] is element
  // (position 16w + fr, column 128c + 16t + 4g + r)
  const int pos = 16 * w + fr;
  for (int c = 0; c < 4; ++c) {
    t128_load(img0, p.k + head, p.ss, S, c * 128, tid);
    t128_load(img1, p.q + head, p.ss, S, c * 128, tid);
    t128_load(img2, go, go_stride, S, c * 128, tid);
    __syncthreads();
    const long long dst_off = head + static_cast<long long>(pos) * p.ss + c * 128 + 4 * g;
    auto product = [&](const unsigned char* img_cols, const unsigned char* img_rows, bf16_t* dst) {
      f32x4_t o[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t bb = t64_rows(img_rows, 16 * w, ks, lane);
#pragma unroll
        for (int t = 0; t < 8; ++t)
          o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t128_cols(img_cols, 16 * t, ks, lane), bb, o[t], 0, 0, 0);
      }
      if (pos < S) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const u32x2_t ov = {pack_bf16x2(o[t][0], o[t][1]), pack_bf16x2(o[t][2], o[t][3])};
          *reinterpret_cast<u32x2_t*>(dst + dst_off + 16 * t) = ov;
        }
      }
    };
    product(img2, imgPT, p.dv);     // dV[key]  = sum_q dO[q]  Pd[q][key]
    product(img0, imgDS, p.dq);     // dQ[q]    = sum_key K[key] dS[q][key]
    product(img1, imgDST, p.dk);    // dK[key]  = sum_q Q[q]   dS[q][key]
    __syncthreads();
  }
}

static void attn512_fill(Attn512Args& a, float scale, float drop_p, unsigned long long seed) {
  a.scale = scale;
  a.drop_thresh = drop_threshold(drop_p);
  a.inv_keep = a.drop_thresh ? 1.f / (1.f - drop_p) : 1.f;
  a.seed = seed;
}

void launch_attn512_fwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, long long sb, long long ss, bf16_t* out,
                        bf16_t* probs, const unsigned char* kpm, int B, int S, int H, int causal, float scale,
                        float drop_p, unsigned long long seed, hipStream_t st) {
  Attn512Args a{};
  a.q = q; a.k = k; a.v = v; a.sb = sb; a.ss = ss; a.out = out; a.probs = probs; a.kpm = kpm;
  a.S = S; a.H = H; a.causal = causal;
  attn512_fill(a, scale, drop_p, seed);
  hipLaunchKernelGGL(attn512_fwd_kernel, dim3(H, B), dim3(256), 0, st, a);
}

void launch_attn512_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, long long sb, long long ss,
                        const bf16_t* d_out, const bf16_t* probs, bf16_t* dq, bf16_t* dk, bf16_t* dv, int B, int S,
                        int H, float scale, float drop_p, unsigned long long seed, hipStream_t st) {
  Attn512Args a{};
  a.q = q; a.k = k; a.v = v; a.sb = sb; a.ss = ss; a.d_out = d_out; a.probs = const_cast<bf16_t*>(probs);
  a.dq = dq; a.dk = dk; a.dv = dv; a.S = S; a.H = H;
  attn512_fill(a, scale, drop_p, seed);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn512_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        A512_BWD_SMEM);
    attr_set = true;
  }
  hipLaunchKernelGGL(attn512_bwd_kernel, dim3(H, B), dim3(256), A512_BWD_SMEM, st, a);
}

// keep mask of the attention dropout, [B][H][64][64] bytes (1 = kept): the oracle of the tests
__global__ __launch_bounds__(256) void attn512_mask_kernel(unsigned char* out, long long n, unsigned int thresh,
                                                           unsigned long long seed) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
    out[i] = (thresh == 0 || drop_keep(seed, static_cast<unsigned long long>(i), thresh)) ? 1 : 0;
}
void launch_attn512_mask(unsigned char* out, long long n, float drop_p, unsigned long long seed, hipStream_t st) {
  if (n <= 0) return;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(attn512_mask_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, out, n,
                     drop_threshold(drop_p), seed);
}

// ---------------------------------------------------------------------------- elementwise ops
// mode 0: y = gelu(x);  mode 1: y = g * gelu'(x)  (erf form, as nn.GELU());
// mode 2: y = keep ? x / (1 - p) : 0  (forward and backward of dropout: same mask from (seed, index))
template <int MODE>
__global__ __launch_bounds__(256) void adapter_elementwise_kernel(const bf16_t* __restrict__ x,
                                                                  const bf16_t* __restrict__ g,
                                                                  bf16_t* __restrict__ y, long long n8,
                                                                  unsigned int thresh, float inv_keep,
                                                                  unsigned long long seed) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += gridDim.x * 256ll) {
    const u32x4_t xv = load16(x + i * 8);
    u32x4_t gv = xv, o;
    if constexpr (MODE == 1) gv = load16(g + i * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float r[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float xe = k ? bf16hi(xv[j]) : bf16lo(xv[j]);
        if constexpr (MODE == 0) {
          r[k] = gelu_f(xe);
        } else if constexpr (MODE == 1) {
          const float ge = k ? bf16hi(gv[j]) : bf16lo(gv[j]);
          const float cdf = 0.5f * (1.f + erff(xe * 0.70710678118654752f));
          const float pdf = 0.3989422804014327f * __expf(-0.5f * xe * xe);
          r[k] = ge * (cdf + xe * pdf);
        } else {
          r[k] = drop_keep(seed, static_cast<unsigned long long>(i) * 8 + 2 * j + k, thresh) ? xe * inv_keep : 0.f;
        }
      }
      o[j] = pack_bf16x2(r[0], r[1]);
    }
    *reinterpret_cast<u32x4_t*>(y + i * 8) = o;
  }
}

static unsigned ew_blocks(long long n8) {
  long long b = (n8 + 255) / 256;
  return static_cast<unsigned>(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}
void launch_gelu(const bf16_t* x, bf16_t* y, long long n, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(adapter_elementwise_kernel<0>, dim3(ew_blocks(n / 8)), dim3(256), 0, st, x, nullptr, y, n / 8, 0u, 1.f, 0ull);
}
void launch_gelu_bwd(const bf16_t* x, const bf16_t* g, bf16_t* dx, long long n, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(adapter_elementwise_kernel<1>, dim3(ew_blocks(n / 8)), dim3(256), 0, st, x, g, dx, n / 8, 0u, 1.f, 0ull);
}
void launch_dropout(const bf16_t* x, bf16_t* y, long long n, float p, unsigned long long seed, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(adapter_elementwise_kernel<2>, dim3(ew_blocks(n / 8)), dim3(256), 0, st, x, nullptr, y, n / 8,
                     drop_threshold(p), 1.f / (1.f - p), seed);
}

// LayerNorm backward.  y = (x - mu) rstd w + b:
//   dx = rstd (g w - mean(g w) - xh mean(g w xh)),  dw += g xh,  db += g,   xh = (x - mu) rstd.
// A block walks rows_per_block rows; a thread owns the same column chunks in every row, so the
// parameter gradients are summed in registers and leave as one row each of [blocks][H] partials.
constexpr int LN_MAX_CHUNKS = 4;
__device__ __forceinline__ void ln_block_sum2(float& a, float& b, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  a = wave_sum(a); b = wave_sum(b);
  __syncthreads();
  if (lane == 0) { scratch[wid] = a; scratch[8 + wid] = b; }
  __syncthreads();
  a = scratch[0] + scratch[1] + scratch[2] + scratch[3];
  b = scratch[8] + scratch[9] + scratch[10] + scratch[11];
}
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ g, bf16_t* __restrict__ dx,
                                                            float* __restrict__ dw, float* __restrict__ db, int rows,
                                                            int H, float eps, int rows_per_block) {
  __shared__ float scratch[16];
  const int tid = threadIdx.x, nchunk = H >> 3;
  float dwacc[LN_MAX_CHUNKS][8], dbacc[LN_MAX_CHUNKS][8];
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) { dwacc[i][k] = 0.f; dbacc[i][k] = 0.f; }
  float wv[LN_MAX_CHUNKS][8];
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = tid + i * 256;
    const u32x4_t t = c < nchunk ? load16(w + c * 8) : u32x4_t{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) { wv[i][2 * j] = bf16lo(t[j]); wv[i][2 * j + 1] = bf16hi(t[j]); }
  }
  const float inv_h = 1.f / static_cast<float>(H);
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, rows);
  for (int r = r0; r < r1; ++r) {
    const bf16_t* xr = x + static_cast<size_t>(r) * H;
    const bf16_t* gr = g + static_cast<size_t>(r) * H;
    float xe[LN_MAX_CHUNKS][8], ge[LN_MAX_CHUNKS][8];
    float s1 = 0.f, zero = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
      const int c = tid + i * 256;
      const bool ok = c < nchunk;
      const u32x4_t xv = ok ? load16(xr + c * 8) : u32x4_t{0, 0, 0, 0};
      const u32x4_t gv = ok ? load16(gr + c * 8) : u32x4_t{0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xe[i][2 * j] = bf16lo(xv[j]); xe[i][2 * j + 1] = bf16hi(xv[j]);
        ge[i][2 * j] = bf16lo(gv[j]); ge[i][2 * j + 1] = bf16hi(gv[j]);
        s1 += xe[i][2 * j] + xe[i][2 * j + 1];
      }
    }
    ln_block_sum2(s1, zero, scratch);
    const float mu = s1 * inv_h;
    float s2 = 0.f;
    zero = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
      const bool ok = tid + i * 256 < nchunk;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        xe[i][k] = ok ? xe[i][k] - mu : 0.f;
        s2 += xe[i][k] * xe[i][k];
      }
    }
    ln_block_sum2(s2, zero, scratch);
    const float rstd = rsqrtf(s2 * inv_h + eps);
    float a = 0.f, bsum = 0.f;      // sum(g w), sum(g w xh)
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        xe[i][k] *= rstd;                       // xh
        const float gw = ge[i][k] * wv[i][k];
        a += gw;
        bsum += gw * xe[i][k];
        dwacc[i][k] += ge[i][k] * xe[i][k];
        dbacc[i][k] += ge[i][k];
      }
    ln_block_sum2(a, bsum, scratch);
    const float ma = a * inv_h, mb = bsum * inv_h;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
      const int c = tid + i * 256;
      if (c >= nchunk) continue;
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack_bf16x2(rstd * (ge[i][2 * j] * wv[i][2 * j] - ma - xe[i][2 * j] * mb),
                           rstd * (ge[i][2 * j + 1] * wv[i][2 * j + 1] - ma - xe[i][2 * j + 1] * mb));
      *reinterpret_cast<u32x4_t*>(dx + static_cast<size_t>(r) * H + c * 8) = o;
    }
  }
  if (dw) {
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
      const int c = tid + i * 256;
      if (c >= nchunk) continue;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        dw[static_cast<size_t>(blockIdx.x) * H + c * 8 + k] = dwacc[i][k];
        db[static_cast<size_t>(blockIdx.x) * H + c * 8 + k] = dbacc[i][k];
      }
    }
  }
}

int layernorm_bwd_blocks(int rows) {
  int rpb = (rows + 511) / 512;
  if (rpb < 4) rpb = 4;
  return (rows + rpb - 1) / rpb;
}
void launch_layernorm_bwd(const bf16_t* x, const bf16_t* w, const bf16_t* g, bf16_t* dx, float* dw, float* db,
                          int rows, int H, float eps, hipStream_t st) {
  if (rows <= 0) return;
  const int blocks = layernorm_bwd_blocks(rows);
  const int rpb = (rows + blocks - 1) / blocks;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, st, x, w, g, dx, dw, db, rows, H, eps, rpb);
}

}  // namespace evgpt
