// Tensor-parallel communication kernels.
//
// 1. One-shot all-reduce over peer-mapped buffers (xGMI is a full mesh of point-to-point
//    links, so for the latency-bound messages of a decode step -- [B,4096] partial sums,
//    16 KB .. 512 KB -- every rank PUSHES its partial into a slot of each peer's receive
//    buffer (all 7 links busy at once, posted writes, no read round trip), raises a flag
//    per peer, polls its OWN flag words (local memory), and sums the W slots in rank order
//    -- the result is bit-identical on every rank.  The same launch first sums the local
//    split-K slabs of the projection and afterwards adds the residual and, when a block
//    owns whole rows, applies the next RMSNorm: no separate launches around the collective.
//
//    Protocol (per block region `blk`, double-buffered by the parity of a per-region epoch
//    that lives in device memory, so a captured launch can be replayed):
//       e = epoch[blk] + 1
//       store partial -> peer p's data[e&1][me][blk]      (every p != me)
//       every storing wave drains its stores; barrier; system-scope release
//       flag store     -> peer p's flag[e&1][me][blk] = e
//       poll MY flag[e&1][p][blk] == e for every p != me  (bounded; a timeout sets err[0])
//       system-scope acquire; barrier; sum slots in rank order; epilogue; epoch[blk] = e
//    A rank can be at most one call ahead of a peer for a region (it needs the peer's flag
//    of call e to leave call e, and the peer raises it only after it has finished reading
//    call e-1), so two buffers per region are enough; regions are addressed by block index
//    with a fixed stride, so calls of different sizes never overlap another block's region.
//    The receive buffers and flags are allocated uncached (fine-grained), so neither remote
//    writes nor local polls depend on cache write-back / invalidation of a remote agent.
//    Every wait is bounded: on a timeout the kernel records it, finishes with whatever
//    arrived and exits normally; the host reads err[0] and raises.
//
//    A two-shot variant (reduce-scatter + all-gather, both as direct pushes) carries the
//    bandwidth-bound bf16 messages of a sharded prefill; it is described at its kernel below.
//
// 2. Helpers for the RCCL path and the vocab-parallel head: slab pre-sum, (max, index)
//    pair kernels for the distributed argmax.
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

__device__ __forceinline__ unsigned poll_u32(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int NS>
__device__ __forceinline__ f32x4_t sum_slabs4(const float* __restrict__ partial, long long slab,
                                              int nslab, long long e0) {
  f32x4_t a = {0.f, 0.f, 0.f, 0.f};
  if constexpr (NS > 0) {
    f32x4_t t[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) t[s] = *reinterpret_cast<const f32x4_t*>(partial + s * slab + e0);
#pragma unroll
    for (int s = 0; s < NS; ++s) { a[0] += t[s][0]; a[1] += t[s][1]; a[2] += t[s][2]; a[3] += t[s][3]; }
  } else {
    for (int s = 0; s < nslab; ++s) {
      const f32x4_t t = *reinterpret_cast<const f32x4_t*>(partial + s * slab + e0);
      a[0] += t[0]; a[1] += t[1]; a[2] += t[2]; a[3] += t[3];
    }
  }
  return a;
}

// WT: slots summed (the world size rounded up to 2, 4 or 8)
template <int NS, int WT>
__global__ __launch_bounds__(256) void oneshot_allreduce_kernel(ARCallSet calls) {
  const ARCall& c = calls.c[blockIdx.y];
  const CommState* __restrict__ cs = c.cs;
  __shared__ float red[16];
  const int tid = threadIdx.x, blk = blockIdx.x;
  const int W = cs->world, me = cs->rank;
  const int row = blk / c.bpr, part = blk % c.bpr;
  const long long e0 = static_cast<long long>(row) * c.N + static_cast<long long>(part) * c.chunk;
  const int nq = min(c.chunk, c.N - part * c.chunk) >> 2;   // float4s of this region
  unsigned e = cs->epoch[blk] + 1u;
  if (e == 0u) e = 1u;                                       // 0 is the "never written" flag value
  const unsigned parity = e & 1u;
  const long long slot_stride = static_cast<long long>(COMM_MAX_BLOCKS) * COMM_CHUNK_MAX;  // floats per source rank
  const long long my_slot = (static_cast<long long>(parity) * COMM_MAX_RANKS + me) * slot_stride +
                            static_cast<long long>(blk) * COMM_CHUNK_MAX;

  // ---- 1. local partial of this region (sum of the projection's split-K slabs)
  constexpr int QPT = COMM_CHUNK_MAX / 4 / 256;   // float4s per thread
  f32x4_t v[QPT];
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int q = tid + i * 256;
    v[i] = q < nq ? sum_slabs4<NS>(c.partial, c.slab_stride, c.nslab, e0 + q * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  // ---- 2. push to every peer's slot [parity][me][blk]
  for (int p = 0; p < W; ++p) {
    if (p == me) continue;
    float* dst = cs->data[p] + my_slot;
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
      const int q = tid + i * 256;
      if (q < nq) *reinterpret_cast<f32x4_t*>(dst + q * 4) = v[i];
    }
  }
  // ---- 3. publish: every storing wave drains, then one wave releases and raises the flags
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long long flag_idx = (static_cast<long long>(parity) * COMM_MAX_RANKS + me) * COMM_MAX_BLOCKS + blk;
  if (tid < 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid < W && tid != me)
      __hip_atomic_store(cs->flag[tid] + flag_idx, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- 4. wait for the peers (my own flag words; lane p waits for rank p)
    if (tid < W && tid != me) {
      const unsigned* f = cs->flag[me] + (static_cast<long long>(parity) * COMM_MAX_RANKS + tid) * COMM_MAX_BLOCKS + blk;
      unsigned spins = 0;
      while (poll_u32(f) != e) {
        if (++spins > cs->spin_limit) {          // give up: record it and go on with what arrived
          atomicAdd(cs->err, 1u);
          cs->err[1] = e; cs->err[2] = static_cast<unsigned>(tid); cs->err[3] = static_cast<unsigned>(blk);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // ---- 5. sum the W contributions in rank order (identical on every rank)
  const float* mine = cs->data[me] + static_cast<long long>(parity) * COMM_MAX_RANKS * slot_stride +
                      static_cast<long long>(blk) * COMM_CHUNK_MAX;
  f32x4_t acc[QPT];
  {
    f32x4_t t[QPT][WT];
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
      const int q = tid + i * 256;
      const int qq = q < nq ? q : 0;
#pragma unroll
      for (int p = 0; p < WT; ++p) {
        // every one of the COMM_MAX_RANKS slots is allocated: slots of absent ranks (and my own,
        // which stays in registers) are read and dropped rather than branched around
        t[i][p] = *reinterpret_cast<const f32x4_t*>(mine + p * slot_stride + qq * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
      acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < WT; ++p) {
        if (p >= W) continue;
        const f32x4_t s = p == me ? v[i] : t[i][p];
        acc[i][0] += s[0]; acc[i][1] += s[1]; acc[i][2] += s[2]; acc[i][3] += s[3];
      }
    }
  }
  // ---- 6. epilogue: x = bf16(sum) + res; optional RMSNorm when the block owns the whole row
  float ss = 0.f;
  u32x2_t xo[QPT];
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int q = tid + i * 256;
    if (q >= nq) continue;
    float y[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
    if (c.sum_out) *reinterpret_cast<f32x4_t*>(c.sum_out + e0 + q * 4) = acc[i];
    if (c.x_out) {
      if (c.res) {
        const u32x2_t r = *reinterpret_cast<const u32x2_t*>(c.res + e0 + q * 4);
        y[0] = bf16_to_f32(f32_to_bf16(y[0])) + bf16lo(r[0]); y[1] = bf16_to_f32(f32_to_bf16(y[1])) + bf16hi(r[0]);
        y[2] = bf16_to_f32(f32_to_bf16(y[2])) + bf16lo(r[1]); y[3] = bf16_to_f32(f32_to_bf16(y[3])) + bf16hi(r[1]);
      }
      xo[i] = u32x2_t{pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
      *reinterpret_cast<u32x2_t*>(c.x_out + e0 + q * 4) = xo[i];
      const float a0 = bf16lo(xo[i][0]), a1 = bf16hi(xo[i][0]), a2 = bf16lo(xo[i][1]), a3 = bf16hi(xo[i][1]);
      ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
    }
  }
  if (c.xn_out && c.bpr == 1) {   // block-uniform
    ss = block_sum(ss, red);
    const float rstd = rsqrtf(ss / static_cast<float>(c.N) + c.eps);
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
      const int q = tid + i * 256;
      if (q >= nq) continue;
      const u32x2_t g = *reinterpret_cast<const u32x2_t*>(c.norm_w + q * 4);
      const float a0 = bf16_to_f32(f32_to_bf16(bf16lo(xo[i][0]) * rstd)) * bf16lo(g[0]);
      const float a1 = bf16_to_f32(f32_to_bf16(bf16hi(xo[i][0]) * rstd)) * bf16hi(g[0]);
      const float a2 = bf16_to_f32(f32_to_bf16(bf16lo(xo[i][1]) * rstd)) * bf16lo(g[1]);
      const float a3 = bf16_to_f32(f32_to_bf16(bf16hi(xo[i][1]) * rstd)) * bf16hi(g[1]);
      *reinterpret_cast<u32x2_t*>(c.xn_out + e0 + q * 4) = u32x2_t{pack_bf16x2(a0, a1), pack_bf16x2(a2, a3)};
    }
  }
  if (tid == 0) cs->epoch[blk] = e;
}

// Region geometry of a call: rows of up to COMM_CHUNK_MAX floats are one block each (the block
// can then normalise the row); a handful of rows are cut into pieces so that more than a few
// CUs push at once.  Returns 0 when the message does not fit the buffers.
int comm_plan(int M, int N, int* chunk, int* bpr) {
  if (M < 1 || N < 4 || (N & 3)) return 0;
  int b = 1;
  if (N > COMM_CHUNK_MAX) b = (N + COMM_CHUNK_MAX - 1) / COMM_CHUNK_MAX;
  if (M <= 3) { while (b < 8 && (N / (b * 2)) >= 256 && (N % (b * 2 * 4)) == 0) b *= 2; }
  int ch = (N + b - 1) / b;
  ch = (ch + 3) & ~3;
  if (ch > COMM_CHUNK_MAX || static_cast<long long>(M) * b > COMM_MAX_BLOCKS) return 0;
  *chunk = ch; *bpr = b;
  return M * b;
}

template <int WT>
static void launch_ar_w(const ARCallSet& calls, dim3 g, int nslab, hipStream_t st) {
  const dim3 b(256);
  switch (nslab) {
    case 1: hipLaunchKernelGGL((oneshot_allreduce_kernel<1, WT>), g, b, 0, st, calls); break;
    case 2: hipLaunchKernelGGL((oneshot_allreduce_kernel<2, WT>), g, b, 0, st, calls); break;
    case 4: hipLaunchKernelGGL((oneshot_allreduce_kernel<4, WT>), g, b, 0, st, calls); break;
    case 8: hipLaunchKernelGGL((oneshot_allreduce_kernel<8, WT>), g, b, 0, st, calls); break;
    default: hipLaunchKernelGGL((oneshot_allreduce_kernel<0, WT>), g, b, 0, st, calls); break;
  }
}

void launch_oneshot_allreduce(const ARCallSet& calls, int n_virtual, int blocks, int nslab, int world,
                              hipStream_t st) {
  const dim3 g(blocks, n_virtual);
  if (world <= 2) launch_ar_w<2>(calls, g, nslab, st);
  else if (world <= 4) launch_ar_w<4>(calls, g, nslab, st);
  else launch_ar_w<8>(calls, g, nslab, st);
}

// ------------------------------------------------------------------------------------------
// 1b. Two-shot all-reduce for bandwidth-bound messages (the [S, 4096] products of a sharded
//     prefill: 5 MB at 636 tokens).  One-shot would make every rank receive W - 1 whole messages;
//     here rank r owns slice r of the message:
//       scatter: every rank pushes piece b of slice p into peer p's slot [parity][me], flag
//       reduce : the owner sums its own piece and the W - 1 staged ones in rank order (fp32)
//       gather : and pushes the bf16 result into every peer's second slot [parity][me], flag;
//                each rank copies the W - 1 arrived pieces into its output.
//     Every rank sends and receives (W - 1) / W of the message twice, spread over all links at
//     once (a ring moves the same bytes through one link per hop).  The sum of an element is
//     computed by exactly one rank, so the result is bit-identical everywhere.  Flags, parity,
//     bounded waits and the error word work as in the one-shot kernel; a block depends only on
//     the blocks of the SAME index on the other ranks, never on another block of its own grid.
template <int WT>
__global__ __launch_bounds__(256) void twoshot_allreduce_kernel(TSCallSet calls) {
  const TSCall& c = calls.c[blockIdx.y];
  const Comm2State* __restrict__ cs = c.cs;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int W = cs->world, me = cs->rank;
  unsigned e = cs->epoch[b] + 1u;
  if (e == 0u) e = 1u;
  const unsigned parity = e & 1u;
  const long long slice = static_cast<long long>(c.pieces) * COMM2_PIECE;
  const long long in_slice = static_cast<long long>(b) * COMM2_PIECE + tid * 8;    // this lane's 8 elements
  const long long my_slot = (static_cast<long long>(parity) * W + me) * cs->slice_cap + in_slice;
  const long long flag_plane = static_cast<long long>(W) * cs->pieces_cap;          // words per (phase, parity)
  auto flag_of = [&](int phase, int src) {
    return (static_cast<long long>(phase) * 2 + parity) * flag_plane + static_cast<long long>(src) * cs->pieces_cap + b;
  };
  auto publish_and_wait = [&](int phase) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < 64) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid < W && tid != me)
        __hip_atomic_store(cs->flag[tid] + flag_of(phase, me), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (tid < W && tid != me) {
        const unsigned* f = cs->flag[me] + flag_of(phase, tid);
        unsigned spins = 0;
        while (poll_u32(f) != e) {
          if (++spins > cs->spin_limit) {        // give up: record it and go on with what arrived
            atomicAdd(cs->err, 1u);
            cs->err[1] = e; cs->err[2] = static_cast<unsigned>(tid); cs->err[3] = static_cast<unsigned>(b);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  };

  // ---- scatter: piece b of slice p goes to rank p
  u32x4_t mine = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int p = 0; p < WT; ++p) {
    if (p >= W) continue;
    const long long el = static_cast<long long>(p) * slice + in_slice;
    const u32x4_t v = el < c.E ? load16(c.x + el) : u32x4_t{0u, 0u, 0u, 0u};
    if (p == me) mine = v;
    else *reinterpret_cast<u32x4_t*>(cs->stage1[p] + my_slot) = v;
  }
  publish_and_wait(0);

  // ---- reduce my slice's piece in rank order, fp32
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  {
    u32x4_t t[WT];
#pragma unroll
    for (int q = 0; q < WT; ++q) {
      // slots of absent ranks (and my own, which stays in registers) are read and dropped
      const int qq = q < W ? q : 0;
      t[q] = *reinterpret_cast<const u32x4_t*>(cs->stage1[me] + (static_cast<long long>(parity) * W + qq) * cs->slice_cap + in_slice);
    }
#pragma unroll
    for (int q = 0; q < WT; ++q) {
      if (q >= W) continue;
      const u32x4_t v = q == me ? mine : t[q];
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[2 * i] += bf16lo(v[i]); acc[2 * i + 1] += bf16hi(v[i]); }
    }
  }
  const u32x4_t r = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]),
                     pack_bf16x2(acc[6], acc[7])};
  const long long my_el = static_cast<long long>(me) * slice + in_slice;
  if (my_el < c.E) *reinterpret_cast<u32x4_t*>(c.out + my_el) = r;
  // ---- gather: the reduced piece goes to every peer
#pragma unroll
  for (int p = 0; p < WT; ++p) {
    if (p >= W || p == me) continue;
    *reinterpret_cast<u32x4_t*>(cs->stage2[p] + my_slot) = r;
  }
  publish_and_wait(1);
#pragma unroll
  for (int q = 0; q < WT; ++q) {
    if (q >= W || q == me) continue;
    const long long el = static_cast<long long>(q) * slice + in_slice;
    if (el < c.E)
      *reinterpret_cast<u32x4_t*>(c.out + el) =
          *reinterpret_cast<const u32x4_t*>(cs->stage2[me] + (static_cast<long long>(parity) * W + q) * cs->slice_cap + in_slice);
  }
  if (tid == 0) cs->epoch[b] = e;
}

void launch_twoshot_allreduce(const TSCallSet& calls, int n_virtual, int pieces, int world, hipStream_t st) {
  const dim3 g(pieces, n_virtual), b(256);
  if (world <= 2) hipLaunchKernelGGL(twoshot_allreduce_kernel<2>, g, b, 0, st, calls);
  else if (world <= 4) hipLaunchKernelGGL(twoshot_allreduce_kernel<4>, g, b, 0, st, calls);
  else hipLaunchKernelGGL(twoshot_allreduce_kernel<8>, g, b, 0, st, calls);
}

// ---------------------------------------------------------------- RCCL-path helpers
// out[M*N] = sum_s partial[s][M*N]   (the message of the library all-reduce)
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ partial, int nslab,
                                                        long long n, float* __restrict__ out) {
  for (long long q = blockIdx.x * 256ll + threadIdx.x; q * 4 < n; q += gridDim.x * 256ll)
    *reinterpret_cast<f32x4_t*>(out + q * 4) = sum_slabs4<0>(partial, n, nslab, q * 4);
}

void launch_sum_slabs(const float* partial, int nslab, long long n, float* out, hipStream_t st) {
  int blocks = static_cast<int>((n / 4 + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sum_slabs_kernel, dim3(blocks), dim3(256), 0, st, partial, nslab, n, out);
}

// ---------------------------------------------------------------- distributed argmax
// Per-row argmax partials of this rank's vocabulary shard -> one (value, global index) pair
// per row, written as two 32-bit words into pair[row] (the all-gather's send buffer).
__global__ __launch_bounds__(64) void argmax_pair_local_kernel(const float* __restrict__ part_val,
                                                               const int* __restrict__ part_idx,
                                                               int n_part, int index_offset,
                                                               unsigned* __restrict__ pair) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < n_part; i += 64) {
    const float v = part_val[static_cast<long long>(b) * n_part + i];
    const int ix = part_idx[static_cast<long long>(b) * n_part + i];
    if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) {
    pair[2 * b] = __float_as_uint(bv);
    pair[2 * b + 1] = static_cast<unsigned>(bi == 0x7fffffff ? 0x7fffffff : bi + index_offset);
  }
}

// pairs [W][B][2] (gathered) -> token[b] = index of the largest value, lowest index on ties
__global__ __launch_bounds__(64) void argmax_pair_select_kernel(const unsigned* __restrict__ pairs, int W,
                                                                int B, int* __restrict__ token) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int w = 0; w < W; ++w) {
    const float v = __uint_as_float(pairs[(static_cast<long long>(w) * B + b) * 2]);
    const int ix = static_cast<int>(pairs[(static_cast<long long>(w) * B + b) * 2 + 1]);
    if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
  }
  token[b] = bi == 0x7fffffff ? 0 : bi;
}

void launch_argmax_pair_local(const float* part_val, const int* part_idx, int B, int n_part,
                              int index_offset, unsigned* pair, hipStream_t st) {
  hipLaunchKernelGGL(argmax_pair_local_kernel, dim3(B), dim3(64), 0, st, part_val, part_idx, n_part,
                     index_offset, pair);
}

void launch_argmax_pair_select(const unsigned* pairs, int W, int B, int* token, hipStream_t st) {
  hipLaunchKernelGGL(argmax_pair_select_kernel, dim3((B + 63) / 64), dim3(64), 0, st, pairs, W, B, token);
}

}  // namespace evgpt
