// Timing probe for a persistent decode-layer kernel: one block per CU streams its share of every
// weight matrix of a 7B-geometry decoder through per-wave direct-to-LDS rings, with the grid
// barriers a fused layer needs between its phases (qkv | attention | combine | o_proj | gate/up |
// down).  The arithmetic is a plain dot product per weight row against an LDS-resident vector; the
// probe answers two questions before the real kernel is written: what streaming rate the ring
// structure reaches without barriers, and what the barriers cost with the weights still streaming.
//
//   persist_probe [layers=32] [iters=10]
//
// Every device wait is bounded (spin limit -> error word, the block carries on and exits).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../eventgpt_amd/csrc/kernels/common.h"

using namespace evgpt;

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int NB = 256;   // blocks (one per CU)
constexpr int NW = 8;     // waves per block
constexpr int CH = 1024;  // bytes per wave-wide 16-byte load
constexpr int XBYTES = 22528;  // LDS vector: 11264 bf16 (down_proj input, zero padded)

struct Probe {
  const unsigned char* w[4];  // qkv, o, gate_up, down of layer 0
  size_t lstride[4];          // bytes between layers
  int rows[4];
  int rowbytes[4];
  int layers;
  const unsigned char* kv;    // [layers][kv_bytes]
  size_t kv_bytes;
  unsigned short* vec[4];     // outputs of the four projections (inputs of the next one)
  float* part;                // attention partials [NB][130]
  unsigned int* bar;
  unsigned int* err;
  int barriers;               // 0: none, 1: after each projection, 2: + the two attention phases
  int spin_limit;
};

template <int AUX>
__device__ __forceinline__ void dma16(const unsigned char* base, int bytes, unsigned char* lds, int voff) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, 0, 0, AUX);
}

template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else static_assert(N < 0, "add the literal");
}

// position in a wave's chunk stream
struct Cursor {
  int layer, m, r, r_end, c, nc;
  __device__ __forceinline__ void start_matrix(const Probe& p, int g) {
    const int nwaves = NB * NW;
    r = static_cast<int>(static_cast<long long>(g) * p.rows[m] / nwaves);
    r_end = static_cast<int>(static_cast<long long>(g + 1) * p.rows[m] / nwaves);
    c = 0;
    nc = (p.rowbytes[m] + CH - 1) / CH;
  }
  __device__ __forceinline__ bool done(const Probe& p) const { return layer >= p.layers; }
  // returns true when the step left a matrix (phase boundary)
  __device__ __forceinline__ bool advance(const Probe& p, int g) {
    if (++c < nc) return false;
    c = 0;
    if (++r < r_end) return false;
    if (++m == 4) { m = 0; ++layer; }
    if (layer < p.layers) start_matrix(p, g);
    return true;
  }
};

__device__ __forceinline__ void grid_sync(const Probe& p, unsigned int& target, int& dead) {
  wait_vm<0>();
  __syncthreads();
  target += gridDim.x;
  if (threadIdx.x == 0 && !dead) {
    __hip_atomic_fetch_add(p.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(p.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > p.spin_limit) {
        __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}

template <int D, int AUX>
__global__ __launch_bounds__(NW * 64, 1) void probe_kernel(Probe p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xl = smem;                       // XBYTES
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* ring = smem + XBYTES + wid * (D * CH);
  const int g = blockIdx.x * NW + wid;
  unsigned int target = 0;
  int dead = 0;

  // x vector: something finite
  for (int i = tid; i < XBYTES / 4; i += NW * 64) reinterpret_cast<unsigned int*>(xl)[i] = 0x3c003c00u;
  __syncthreads();

  Cursor is{0, 0, 0, 0, 0, 0}, cs{0, 0, 0, 0, 0, 0};
  is.start_matrix(p, g);
  cs.start_matrix(p, g);
  auto issue = [&](int slot) {
    if (is.done(p)) {   // keep the in-flight count uniform: a zero-length buffer reads nothing
      dma16<AUX>(p.w[0], 0, ring + slot * CH, lane * 16);
      return;
    }
    const unsigned char* rowp = p.w[is.m] + is.layer * p.lstride[is.m] + static_cast<size_t>(is.r) * p.rowbytes[is.m];
    dma16<AUX>(rowp, p.rowbytes[is.m], ring + slot * CH, is.c * CH + lane * 16);
    is.advance(p, g);
  };
#pragma unroll
  for (int s = 0; s < D; ++s) issue(s);

  float acc = 0.f;
  int slot = 0;
  while (!cs.done(p)) {
    wait_vm<D - 1>();
    const u32x4_t wv = *reinterpret_cast<const u32x4_t*>(ring + slot * CH + lane * 16);
    const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(xl + cs.c * CH + lane * 16);
    acc = dot8_bf16(wv, xv, acc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue(slot);
    slot = slot + 1 == D ? 0 : slot + 1;
    const int m = cs.m, r = cs.r;
    const bool row_end = cs.c + 1 == cs.nc;
    if (row_end) {
      const float s = wave_sum(acc);
      acc = 0.f;
      if (lane == 0) {
        const int n_out = m == 2 ? p.rows[2] / 2 : p.rows[m];
        __hip_atomic_store(p.vec[m] + (r % n_out), f32_to_bf16(s * 1e-3f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const bool phase_end = cs.advance(p, g);
    if (phase_end && p.barriers) {
      grid_sync(p, target, dead);
      if (m == 0 && p.barriers >= 2) {
        // attention stand-in: every wave reads its slice of this layer's K/V, the block leaves a
        // partial, barrier, every block merges eight partials, barrier
        const int layer = cs.layer;   // advance() moved to the o_proj phase of the same layer
        const size_t per_wave = p.kv_bytes / (NB * NW) / CH * CH;
        const unsigned char* kp = p.kv + static_cast<size_t>(layer) * p.kv_bytes + static_cast<size_t>(g) * per_wave;
        float a = 0.f;
        for (size_t o = 0; o < per_wave; o += CH) {
          const u32x4_t v = load16(kp + o + lane * 16);
          a += bf16lo(v[0]) + bf16hi(v[3]);
        }
        a = wave_sum(a);
        if (lane == 0)
          __hip_atomic_store(p.part + (blockIdx.x * NW + wid), a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_sync(p, target, dead);
        float b = 0.f;
        for (int i = tid; i < 8 * 130; i += NW * 64)
          b += __hip_atomic_load(p.part + ((blockIdx.x / 8) * 8 * NW + i) % (NB * NW), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 16)
          __hip_atomic_store(p.vec[0] + blockIdx.x * 16 + tid, f32_to_bf16(b * 1e-3f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_sync(p, target, dead);
      }
      // the next projection's input vector, written by every block
      const int nx = (cs.done(p) ? 8192 : p.rowbytes[cs.m]) / 4;
      const unsigned int* src = reinterpret_cast<const unsigned int*>(p.vec[m]);
      const int avail = (m == 2 ? p.rows[2] / 2 : p.rows[m]) / 2;
      for (int i = tid; i < nx; i += NW * 64)
        reinterpret_cast<unsigned int*>(xl)[i] = __hip_atomic_load(src + (i % avail), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x3fff3fffu;
      __syncthreads();
    }
  }
  wait_vm<0>();
}

// nothing but barriers: what one grid-wide rendezvous costs with idle memory pipes
__global__ __launch_bounds__(NW * 64, 1) void barrier_only_kernel(Probe p, int n) {
  unsigned int target = 0;
  int dead = 0;
  for (int i = 0; i < n; ++i) grid_sync(p, target, dead);
}

// plain streaming read of the same bytes: the box's ceiling
__global__ __launch_bounds__(256) void stream_kernel(const u32x4_t* __restrict__ src, size_t n16, float* out) {
  float a = 0.f;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4_t v0 = load_stream16(src + i), v1 = load_stream16(src + i + stride);
    const u32x4_t v2 = load_stream16(src + i + 2 * stride), v3 = load_stream16(src + i + 3 * stride);
    a += bf16lo(v0[0]) + bf16lo(v1[1]) + bf16lo(v2[2]) + bf16lo(v3[3]);
  }
  if (a == 123.456f) out[0] = a;
}

__global__ void fill_kernel(unsigned int* p, size_t n) {
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) {
    unsigned int h = static_cast<unsigned int>(i) * 2654435761u;
    h ^= h >> 15;
    // two bf16 values in (-0.03, 0.03)
    p[i] = (0x3c000000u | (h & 0x807f0000u)) | ((0x3c00u | ((h >> 3) & 0x807fu)));
  }
}

__global__ void zero_kernel(unsigned int* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

template <int D, int AUX>
static float run(const Probe& p, int iters, unsigned int* h_err) {
  const size_t smem = XBYTES + NW * D * CH;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<D, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) {
    zero_kernel<<<1, 64>>>(p.bar, 2);
    probe_kernel<D, AUX><<<NB, NW * 64, smem>>>(p);
  }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) {
    zero_kernel<<<1, 64>>>(p.bar, 1);
    probe_kernel<D, AUX><<<NB, NW * 64, smem>>>(p);
  }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(h_err, p.err, 4, hipMemcpyDeviceToHost));
  return ms / iters;
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 32;
  const int iters = argc > 2 ? atoi(argv[2]) : 10;
  Probe p{};
  const int rows[4] = {12288, 4096, 22016, 4096};
  const int rowbytes[4] = {8192, 8192, 8192, 22016};
  size_t total = 0;
  unsigned char* wbuf[4];
  for (int m = 0; m < 4; ++m) {
    p.rows[m] = rows[m];
    p.rowbytes[m] = rowbytes[m];
    p.lstride[m] = static_cast<size_t>(rows[m]) * rowbytes[m];
    const size_t bytes = p.lstride[m] * layers;
    CK(hipMalloc(&wbuf[m], bytes));
    fill_kernel<<<4096, 256>>>(reinterpret_cast<unsigned int*>(wbuf[m]), bytes / 4);
    p.w[m] = wbuf[m];
    total += bytes;
  }
  p.layers = layers;
  p.kv_bytes = static_cast<size_t>(636) * 4096 * 2 * 2 / (NB * NW * CH) * (NB * NW * CH);
  unsigned char* kv;
  CK(hipMalloc(&kv, p.kv_bytes * layers));
  fill_kernel<<<4096, 256>>>(reinterpret_cast<unsigned int*>(kv), p.kv_bytes * layers / 4);
  p.kv = kv;
  for (int m = 0; m < 4; ++m) {
    CK(hipMalloc(&p.vec[m], 32768));
    fill_kernel<<<32, 256>>>(reinterpret_cast<unsigned int*>(p.vec[m]), 32768 / 4);
  }
  CK(hipMalloc(&p.part, NB * NW * 4 + 4096));
  CK(hipMalloc(&p.bar, 64));
  CK(hipMalloc(&p.err, 64));
  zero_kernel<<<1, 64>>>(p.bar, 16);
  zero_kernel<<<1, 64>>>(p.err, 16);
  p.spin_limit = 200000;
  CK(hipDeviceSynchronize());
  printf("weights %.2f GB over %d layers, kv %.1f MB per layer\n", total / 1e9, layers, p.kv_bytes / 1e6);
  fflush(stdout);

  // ceiling: plain streaming loads over the qkv + gate_up buffers
  {
    float* out;
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t bytes = p.lstride[2] * layers;
    for (int grid : {2048, 4096, 8192}) {
      stream_kernel<<<grid, 256>>>(reinterpret_cast<const u32x4_t*>(wbuf[2]), bytes / 16, out);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) stream_kernel<<<grid, 256>>>(reinterpret_cast<const u32x4_t*>(wbuf[2]), bytes / 16, out);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("plain stream grid %5d: %.3f ms for %.2f GB = %.2f TB/s\n", grid, ms / 5, bytes / 1e9, bytes / (ms / 5) / 1e9);
      fflush(stdout);
    }
  }

  {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int n = 1000;
    for (int rep = 0; rep < 2; ++rep) {
      zero_kernel<<<1, 64>>>(p.bar, 1);
      CK(hipEventRecord(e0));
      barrier_only_kernel<<<NB, NW * 64>>>(p, n);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
    }
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned int e = 0;
    CK(hipMemcpy(&e, p.err, 4, hipMemcpyDeviceToHost));
    printf("barrier only: %.2f us per grid barrier (256 blocks, %d barriers) err=%u\n", ms * 1e3 / n, n, e);
    fflush(stdout);
    if (e) return 2;
  }

  unsigned int err = 0;
  for (int barriers = 0; barriers <= 2; ++barriers) {
    p.barriers = barriers;
    float ms;
    ms = run<16, 0>(p, iters, &err);
    printf("ring 16 KB/wave           barriers=%d: %.3f ms  (%.2f TB/s)  err=%u\n", barriers, ms, total / ms / 1e9, err);
    fflush(stdout);
    if (err) break;
    ms = run<16, 2>(p, iters, &err);
    printf("ring 16 KB/wave, nt loads barriers=%d: %.3f ms  (%.2f TB/s)  err=%u\n", barriers, ms, total / ms / 1e9, err);
    fflush(stdout);
    if (err) break;
    ms = run<8, 0>(p, iters, &err);
    printf("ring  8 KB/wave           barriers=%d: %.3f ms  (%.2f TB/s)  err=%u\n", barriers, ms, total / ms / 1e9, err);
    fflush(stdout);
    if (err) break;
  }
  return err ? 2 : 0;
}
