// Fused AdamW update of one optimizer shard: a single pass over the gradient shard, the
// fp32 master weights and both moments that also writes the refreshed parameter shard in
// its storage type.  The unfused sequence is ten elementwise launches with fp32
// temporaries (~56 bytes of traffic per parameter); this is 30 (bf16 parameters).
//
//   g  = grad * clip                       m = b1 m + (1-b1) g      v = b2 v + (1-b2) g^2
//   w *= 1 - lr*wd                         w -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

template <typename T> __device__ __forceinline__ float ld_as_f32(const T* p);
template <> __device__ __forceinline__ float ld_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_as_f32<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st_from_f32(T* p, float v);
template <> __device__ __forceinline__ void st_from_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_from_f32<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

__device__ __forceinline__ float adamw_one(float g, float& m, float& v, float w, const AdamWArgs& a) {
  g *= a.clip;
  m = a.beta1 * m + (1.f - a.beta1) * g;
  v = a.beta2 * v + (1.f - a.beta2) * g * g;
  const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
  return w * a.decay - a.step_size * (m / denom);
}

template <typename T>
__global__ __launch_bounds__(256) void fused_adamw_kernel(const T* __restrict__ grad, float* __restrict__ master,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          T* __restrict__ param_out, long long n, AdamWArgs a) {
  if (a.clip_dev) a.clip *= *a.clip_dev;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x * 4;
  for (long long i0 = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) * 4; i0 < n; i0 += stride) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long i = i0 + k;
      if (i >= n) break;
      float mi = m[i], vi = v[i];
      const float w = adamw_one(ld_as_f32(grad + i), mi, vi, master[i], a);
      m[i] = mi; v[i] = vi; master[i] = w;
      st_from_f32(param_out + i, w);
    }
  }
}

// 8 elements per thread with 16-byte accesses (all pointers 16-byte aligned, n % 8 == 0)
template <typename T>
__global__ __launch_bounds__(256) void fused_adamw_vec_kernel(const T* __restrict__ grad, float* __restrict__ master,
                                                              float* __restrict__ m, float* __restrict__ v,
                                                              T* __restrict__ param_out, long long n8, AdamWArgs a) {
  if (a.clip_dev) a.clip *= *a.clip_dev;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; c < n8; c += stride) {
    float g[8], mm[8], vv[8], w[8];
    if constexpr (sizeof(T) == 2) {
      const u32x4_t gv = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(grad) + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) { g[2 * j] = bf16lo(gv[j]); g[2 * j + 1] = bf16hi(gv[j]); }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(grad) + 2 * c + h);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[4 * h + j] = t[j];
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4_t tm = reinterpret_cast<const f32x4_t*>(m)[2 * c + h];
      const f32x4_t tv = reinterpret_cast<const f32x4_t*>(v)[2 * c + h];
      const f32x4_t tw = reinterpret_cast<const f32x4_t*>(master)[2 * c + h];
#pragma unroll
      for (int j = 0; j < 4; ++j) { mm[4 * h + j] = tm[j]; vv[4 * h + j] = tv[j]; w[4 * h + j] = tw[j]; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = adamw_one(g[k], mm[k], vv[k], w[k], a);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      reinterpret_cast<f32x4_t*>(m)[2 * c + h] = f32x4_t{mm[4 * h], mm[4 * h + 1], mm[4 * h + 2], mm[4 * h + 3]};
      reinterpret_cast<f32x4_t*>(v)[2 * c + h] = f32x4_t{vv[4 * h], vv[4 * h + 1], vv[4 * h + 2], vv[4 * h + 3]};
      reinterpret_cast<f32x4_t*>(master)[2 * c + h] = f32x4_t{w[4 * h], w[4 * h + 1], w[4 * h + 2], w[4 * h + 3]};
    }
    if constexpr (sizeof(T) == 2) {
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(w[2 * j], w[2 * j + 1]);
      reinterpret_cast<u32x4_t*>(param_out)[c] = o;
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        reinterpret_cast<f32x4_t*>(param_out)[2 * c + h] = f32x4_t{w[4 * h], w[4 * h + 1], w[4 * h + 2], w[4 * h + 3]};
    }
  }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
static void launch_t(const T* grad, float* master, float* m, float* v, T* out, long long n,
                     const AdamWArgs& a, hipStream_t st) {
  if (n <= 0) return;
  if (n % 8 == 0 && aligned16(grad) && aligned16(master) && aligned16(m) && aligned16(v) && aligned16(out)) {
    const long long n8 = n / 8;
    long long vb = (n8 + 255) / 256;
    if (vb > 8192) vb = 8192;
    hipLaunchKernelGGL(fused_adamw_vec_kernel<T>, dim3(static_cast<unsigned int>(vb)), dim3(256), 0, st, grad,
                       master, m, v, out, n8, a);
    return;
  }
  long long blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fused_adamw_kernel<T>, dim3(static_cast<unsigned int>(blocks)), dim3(256), 0, st, grad,
                     master, m, v, out, n, a);
}

void launch_fused_adamw_bf16(const bf16_t* grad, float* master, float* m, float* v, bf16_t* out, long long n,
                             const AdamWArgs& a, hipStream_t st) {
  launch_t<bf16_t>(grad, master, m, v, out, n, a, st);
}
void launch_fused_adamw_f32(const float* grad, float* master, float* m, float* v, float* out, long long n,
                            const AdamWArgs& a, hipStream_t st) {
  launch_t<float>(grad, master, m, v, out, n, a, st);
}

// ---- sum of squares of a gradient shard (global-norm clipping), two deterministic stages:
// `blocks` partial sums, then one block adds them to acc[0] (fp32; accumulates across buckets).
template <typename T>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const T* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  const long long stride = static_cast<long long>(gridDim.x) * 256 * 8;
  for (long long i0 = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 8; i0 < n; i0 += stride) {
    if (i0 + 8 <= n) {
      if constexpr (sizeof(T) == 2) {
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(x + i0));
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float a = bf16lo(v[j]), b = bf16hi(v[j]); s += a * a + b * b; }
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(x + i0) + h);
          s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
      }
    } else {
      for (long long i = i0; i < n; ++i) { const float a = ld_as_f32(x + i); s += a * a; }
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int n, float* __restrict__ acc) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) acc[0] += s;
}

int sumsq_blocks(long long n) {
  long long b = (n + 256 * 8 * 4 - 1) / (256 * 8 * 4);   // at least four 16-byte loads per thread
  return static_cast<int>(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

void launch_sumsq(const void* x, bool bf16, long long n, float* part, float* acc, hipStream_t st) {
  const int blocks = sumsq_blocks(n);
  if (bf16) hipLaunchKernelGGL(sumsq_partial_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, static_cast<const bf16_t*>(x), n, part);
  else hipLaunchKernelGGL(sumsq_partial_kernel<float>, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(x), n, part);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, part, blocks, acc);
}

}  // namespace evgpt
