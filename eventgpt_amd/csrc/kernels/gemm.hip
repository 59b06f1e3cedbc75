// bf16 MFMA GEMM with fused epilogue:  C[M,N] = act(A[M,K] · W[N,K]^T + bias) + res
//
// Both operands are K-contiguous (activations row-major, weights in the
// nn.Linear [out][in] layout), which is the natural layout for MFMA A *and* B
// fragments: every fragment is one 16-byte ds_read_b128.
//
// Structure (gfx950):
//   * block tile BM x BN x 64, 4 waves; each wave owns a (BM/WGM) x (BN/WGN)
//     sub-tile built from 16x16x32 bf16 MFMAs;
//   * operands are staged HBM -> LDS with direct-to-LDS 16-byte loads
//     (buffer_load_dwordx4 ... lds through a wave-uniform descriptor that is rebased
//     per K-tile, so the loop carries no per-lane address arithmetic), multi-buffered:
//     the loads of K-tiles t+1.. are issued before the MFMAs of tile t, one barrier
//     per K-tile;
//   * LDS rows are 128 B, so rows two apart share banks; a 16-byte-chunk XOR
//     swizzle  chunk' = chunk ^ ((row >> 1) & 7)  makes every ds_read_b128
//     16-lane group hit 16 distinct slots.  Direct-to-LDS writes are
//     lane-linear, so the swizzle is applied to the per-lane *source* address
//     and again on the read;
//   * the MFMA is issued as  D = W_frag · A_frag  (weights as the A operand) so
//     each lane ends up with 4 consecutive output columns of one row and the
//     epilogue stores 8 bytes per lane;
//   * blockIdx -> tile mapping is XCD-aware: the 8 XCDs have private L2s and
//     blocks are dealt round-robin, so each XCD is given a contiguous run of
//     tiles that share weight panels.
// Any M, N; K must be a multiple of 64 (callers pad).  Rows past M / N are
// clamped on load and masked on store.
//
// Operand layouts for training (template flags AT / BT): an operand may instead be given
// K-STRIDED, i.e. stored [K][M] (A^T) or [K][N] (W^T) -- the layouts of the backward
// products  dX = dY · W  (W [N][K] is the K-strided "weight" of a contraction over N) and
// dW = dY^T · X  (both operands strided in the token dimension).  Such a tile is staged
// as [64 k-rows][BM or BN columns] with the same direct-to-LDS loads and its MFMA
// fragments are gathered with the transposing read ds_read_b64_tr_b16: two reads give a
// lane its column's 8 consecutive k, the same fragment a ds_read_b128 gives for a
// K-contiguous operand, so any mix of layouts feeds the same MFMA.  The image is
// XOR-swizzled (on the source address and on the read) so that the 8 k-rows a half-wave
// gathers at once fall on 8 distinct 32-byte bank slots.
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

constexpr int BK = 64;

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

struct GemmParams {
  const bf16_t* A;     // [M][lda]
  const bf16_t* W;     // [N][ldw]
  bf16_t* C;           // [M][ldc]
  const bf16_t* bias;  // [N] or null
  const bf16_t* res;   // [M][ldc] or null (may alias C)
  float* Cf;           // optional fp32 output [M][ldc] (C may be null)
  int M, N, K, lda, ldw, ldc, act;
  // optional row-argmax partials: [M][n_part] (value, index), n_part = tiles_n * WGN,
  // taken on the fp32 accumulators so C is never written.
  float* amax_val;
  int* amax_idx;
  // split-K: blockIdx = tile * splitk + ks; each split covers k_per_split (a
  // multiple of 64) of K and stores its raw fp32 tile to partial[ks][M][N]; a
  // second launch sums the slabs and applies the epilogue.
  int splitk, k_per_split;
  float* partial;
  int raw_partial;     // store the fp32 slab even when splitk == 1 (a fused consumer sums it)
  // K-strided operands only: k-rows that exist in memory.  A padded contraction (K rounded up
  // to 64) re-reads the last real row for k >= k_rows; the other operand holds zeros there.
  int ka_rows, kw_rows;
};

// wait until at most `n` of this wave's direct-to-LDS loads are still in flight
template <int N> __device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 24, "vmcnt literal");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else static_assert(N == 0, "add the literal");
}

// swizzle of the 16-byte chunk index of k-row `row` in a K-strided tile whose rows hold BX
// columns: a half-wave's transposed read touches k-rows {8g+q} or {8g+4+q} (g in {0,1} or
// {2,3}, q = 0..3), which must land on 8 distinct 32-byte slots of the 256-byte bank row
template <int BX> __device__ __forceinline__ int swz_t(int row) {
  if constexpr (BX == 128) return ((row & 3) | (((row >> 3) & 1) << 2)) << 1;          // 256-B rows
  else return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;                         // 128-B rows, 2 per bank row
}

typedef __attribute__((ext_vector_type(8))) short s16x8_t;

// fragment of a K-strided tile: columns cb + (lane & 15), k = 32*ks + 8*(lane >> 4) + 0..7
template <int BX>
__device__ __forceinline__ bf16x8_t frag_t(const unsigned char* tile, int cb, int ks, int lane) {
  const int fr = lane & 15, g = lane >> 4, q = fr >> 2, pp = fr & 3;
  const int row1 = ks * 32 + 8 * g + q, row2 = row1 + 4;
  const int chunk = (cb >> 3) + (pp >> 1);
  const unsigned char* a1 = tile + row1 * (BX * 2) + ((chunk ^ swz_t<BX>(row1)) << 4) + 8 * (pp & 1);
  const unsigned char* a2 = tile + row2 * (BX * 2) + ((chunk ^ swz_t<BX>(row2)) << 4) + 8 * (pp & 1);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)a1);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)a2);
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// 16 bytes per lane from `src + voff` (range-checked against `bytes`: past it, zeros) straight
// into LDS at `lds + 16 * lane`.  Kept out of the kernel template: the descriptor type exists
// only in the device pass, and a template body the host pass cannot instantiate loses its
// launch stub.
__device__ __forceinline__ void buffer_to_lds16(const unsigned char* src, int bytes, unsigned char* lds, int voff) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, 0, 0, 0);
}

template <int BM, int BN, int WGM, int WGN, int STAGES, bool AT = false, bool BT = false>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
  static_assert(!AT || BM == 64 || BM == 128, "K-strided A needs a 64- or 128-row tile");
  static_assert(!BT || BN == 64 || BN == 128, "K-strided W needs a 64- or 128-column tile");
  constexpr int WM = BM / WGM, WN = BN / WGN;  // per-wave tile
  constexpr int TM = WM / 16, TN = WN / 16;    // 16x16 tiles per wave
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  static_assert(WGM * WGN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n * p.splitk);
  const int tile = bid / p.splitk, ksplit = bid % p.splitk;
  const int k_begin = ksplit * p.k_per_split;
  const int k_len = min(p.k_per_split, p.K - k_begin);
  // consecutive tiles walk down M first: they share one weight panel
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wid / WGN, wc = wid % WGN;

  // ---- staging: one wave-instruction = 1 KiB of the LDS image (8 rows x 128 B of a
  // K-contiguous tile, or 1024 / (2 BX) k-rows of a K-strided one).  The loads are
  // buffer_load ... lds: the per-lane source offset (row, swizzled chunk) is computed ONCE,
  // and each K-tile only rebases the wave-uniform buffer descriptor (scalar ALU) -- the K
  // loop issues no vector address arithmetic for staging (global_load_lds with per-lane
  // 64-bit addresses cost ~20 VALU instructions per load, more than the loop's MFMAs).
  // The descriptor's range check returns zeros for rows past M / N / the contraction
  // length, so edge tiles need no clamping.
  const int srow = lane >> 3, schunk = lane & 7;
  const int wid_u = __builtin_amdgcn_readfirstlane(wid);
  int voff_a[BM / 32], voff_w[BN / 32];
#pragma unroll
  for (int i = 0; i < BM / 32; ++i) {
    const int rg = wid_u * (BM / 32) + i;
    if constexpr (AT) {
      constexpr int CPR = BM / 8;
      const int lin = rg * 64 + lane;                 // 16-byte chunk index in the image
      const int row = lin / CPR, cl = (lin % CPR) ^ swz_t<BM>(row);
      voff_a[i] = (row * p.lda + m0 + cl * 8) * 2;
    } else {
      const int row = rg * 8 + srow;
      const int c = schunk ^ ((row >> 1) & 7);
      voff_a[i] = ((m0 + row) * p.lda + c * 8) * 2;
    }
  }
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) {
    const int rg = wid_u * (BN / 32) + i;
    if constexpr (BT) {
      constexpr int CPR = BN / 8;
      const int lin = rg * 64 + lane;
      const int row = lin / CPR, cl = (lin % CPR) ^ swz_t<BN>(row);
      voff_w[i] = (row * p.ldw + n0 + cl * 8) * 2;
    } else {
      const int row = rg * 8 + srow;
      const int c = schunk ^ ((row >> 1) & 7);
      voff_w[i] = ((n0 + row) * p.ldw + c * 8) * 2;
    }
  }
  // bytes of each operand that exist in memory (everything past them reads as zero)
  const long long bytes_a = 2ll * (AT ? p.ka_rows : p.M) * p.lda;
  const long long bytes_w = 2ll * (BT ? p.kw_rows : p.N) * p.ldw;
  auto stage = [&](int buf, int kt) {
    unsigned char* base = smem + buf * STAGE;
    const int k0 = k_begin + kt * BK;
    const long long adv_a = AT ? 2ll * k0 * p.lda : 2ll * k0;
    const long long adv_w = BT ? 2ll * k0 * p.ldw : 2ll * k0;
    const long long left_a = bytes_a - adv_a, left_w = bytes_w - adv_w;
    const int lim_a = static_cast<int>(left_a > 0 ? left_a : 0), lim_w = static_cast<int>(left_w > 0 ? left_w : 0);
    const unsigned char* src_a = reinterpret_cast<const unsigned char*>(p.A) + adv_a;
    const unsigned char* src_w = reinterpret_cast<const unsigned char*>(p.W) + adv_w;
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) {  // BM/8 row-groups over 4 waves
      const int rg = wid_u * (BM / 32) + i;
      buffer_to_lds16(src_a, lim_a, base + rg * 1024, voff_a[i]);
    }
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
      const int rg = wid_u * (BN / 32) + i;
      buffer_to_lds16(src_w, lim_w, base + A_BYTES + rg * 1024, voff_w[i]);
    }
  };

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int fsw = (frow >> 1) & 7;  // swizzle term: tile bases are multiples of 16 rows
  const int nk = k_len / BK;

  // STAGES-deep ring, STAGES-1 K-tiles in flight.  Direct-to-LDS loads are
  // tracked by vmcnt, so "tile kt has landed" is a *counted* wait that leaves the
  // younger tiles' loads outstanding, followed by a raw s_barrier (a
  // __syncthreads() would drain them all).  After the barrier every wave has
  // finished reading the buffer of tile kt-1, which is the one re-staged next.
  constexpr int G = (BM + BN) / 32;  // direct-to-LDS instructions per wave per stage
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage(s, s);

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = min(STAGES - 2, nk - 1 - kt);
    if (ahead <= 0) wait_vm<0>();
    else if (ahead == 1) wait_vm<G>();
    else if (ahead == 2) wait_vm<2 * G>();
    else wait_vm<3 * G>();
    __builtin_amdgcn_s_barrier();
    if (kt + STAGES - 1 < nk) {
      int nb = cur + STAGES - 1;
      if (nb >= STAGES) nb -= STAGES;
      stage(nb, kt + STAGES - 1);
    }
    const unsigned char* a_base = smem + cur * STAGE + (wr * WM + frow) * 128;
    const unsigned char* b_base = smem + cur * STAGE + A_BYTES + (wc * WN + frow) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = ((ks * 4 + fq) ^ fsw) * 16;
      bf16x8_t af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (AT) af[i] = frag_t<BM>(smem + cur * STAGE, wr * WM + i * 16, ks, lane);
        else af[i] = *reinterpret_cast<const bf16x8_t*>(a_base + i * 16 * 128 + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (BT) bf[j] = frag_t<BN>(smem + cur * STAGE + A_BYTES, wc * WN + j * 16, ks, lane);
        else bf[j] = *reinterpret_cast<const bf16x8_t*>(b_base + j * 16 * 128 + off);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    cur = cur + 1 == STAGES ? 0 : cur + 1;
  }

  // ---- epilogue: lane holds C[m][n .. n+3]
  if (p.splitk > 1 || p.raw_partial) {
    float* slab = p.partial + static_cast<size_t>(ksplit) * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wr * WM + i * 16 + frow;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wc * WN + j * 16 + fq * 4;
        float* dst = slab + static_cast<size_t>(m) * p.N + n;
        if (n + 3 < p.N && (p.N & 3) == 0) {
          *reinterpret_cast<f32x4_t*>(dst) = acc[i][j];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < p.N) dst[r] = acc[i][j][r];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wr * WM + i * 16 + frow;
    if (p.amax_val) {
      // row max over this wave's WN columns: in-lane over j and r, then across the
      // 4 lanes (fq) that share the row
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wc * WN + j * 16 + fq * 4 + r;
          float v = acc[i][j][r];
          if (p.bias && n < p.N) v += bf16_to_f32(p.bias[n]);
          if (n < p.N && (v > bv || (v == bv && n < bi))) { bv = v; bi = n; }
        }
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (fq == 0 && m < p.M) {
        const int n_part = tiles_n * WGN;
        p.amax_val[static_cast<size_t>(m) * n_part + tn * WGN + wc] = bv;
        p.amax_idx[static_cast<size_t>(m) * n_part + tn * WGN + wc] = bi;
      }
    }
    if (m >= p.M || (!p.C && !p.Cf)) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wc * WN + j * 16 + fq * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const size_t o = static_cast<size_t>(m) * p.ldc + n;
      const bool full = (n + 3 < p.N) && ((p.ldc & 3) == 0);
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += bf16_to_f32(p.bias[n + r]);
      }
      if (p.act != ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
      }
      if (p.res) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) v[r] = bf16_to_f32(f32_to_bf16(v[r])) + bf16_to_f32(p.res[o + r]);
      }
      if (p.Cf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < p.N) p.Cf[o + r] = v[r];
      }
      if (p.C) {
        if (full) {
          u32x2_t w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<u32x2_t*>(p.C + o) = w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < p.N) p.C[o + r] = f32_to_bf16(v[r]);
        }
      }
    }
  }
}

// out = act(sum_s partial[s] + bias) + res, 4 columns per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
  const long long quads = (static_cast<long long>(p.M) * p.N + 3) / 4;
  const size_t slab = static_cast<size_t>(p.M) * p.N;
  for (long long q = blockIdx.x * 256ll + threadIdx.x; q < quads; q += gridDim.x * 256ll) {
    const size_t e0 = static_cast<size_t>(q) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = (p.N & 3) == 0;
    for (int s = 0; s < p.splitk; ++s) {
      if (vec) {
        const f32x4_t t = *reinterpret_cast<const f32x4_t*>(p.partial + s * slab + e0);
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
      } else {
        for (int r = 0; r < 4; ++r) if (e0 + r < slab) v[r] += p.partial[s * slab + e0 + r];
      }
    }
    for (int r = 0; r < 4; ++r) {
      const size_t e = e0 + r;
      if (e >= slab) break;
      const int m = static_cast<int>(e / p.N), n = static_cast<int>(e % p.N);
      float x = v[r];
      if (p.bias) x += bf16_to_f32(p.bias[n]);
      x = apply_act(x, p.act);
      const size_t o = static_cast<size_t>(m) * p.ldc + n;
      if (p.res) x = bf16_to_f32(f32_to_bf16(x)) + bf16_to_f32(p.res[o]);
      if (p.Cf) p.Cf[o] = x;
      if (p.C) p.C[o] = f32_to_bf16(x);
    }
  }
}

template <int BM, int BN, int WGM, int WGN, int STAGES, bool AT = false, bool BT = false>
static void launch_cfg(const GemmParams& p, hipStream_t st) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.splitk;
  constexpr size_t smem = static_cast<size_t>(STAGES) * (BM + BN) * BK * 2;
  static bool attr_set = false;
  if (smem > 64 * 1024 && !attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, WGM, WGN, STAGES, AT, BT>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WGM, WGN, STAGES, AT, BT>), dim3(tiles), dim3(256), smem,
                     st, p);
}

// Tile choice: the largest tile that still puts >= ~1.5 blocks on each of the 256
// CUs; shapes that cannot (prefill of a few hundred rows against a 1024- or
// 4096-wide matrix, verify batches) take smaller tiles with a deeper ring, since
// a grid below one block per CU is bound by the latency of its own K loop.
enum GemmCfg { CFG_128x128 = 0, CFG_64x128 = 1, CFG_64x64 = 2, CFG_32x64 = 3, CFG_32x128 = 4 };

static int g_forced_cfg = -1;
void gemm_force_config(int cfg) { g_forced_cfg = cfg; }

static GemmCfg pick_cfg(int M, int N) {
  if (g_forced_cfg >= 0) return static_cast<GemmCfg>(g_forced_cfg);
  auto tiles = [&](int bm, int bn) {
    return static_cast<long long>((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  };
  // thresholds from benchmarks/bench_gemm.py on MI355X
  if (M <= 32) return tiles(32, 128) >= 256 ? CFG_32x128 : CFG_32x64;
  if (tiles(128, 128) >= 384) return CFG_128x128;
  if (tiles(64, 128) >= 300) return CFG_64x128;
  if (tiles(64, 64) >= 300) return CFG_64x64;
  return CFG_32x64;
}

static long long cfg_tiles(GemmCfg c, int M, int N) {
  static const int bm[] = {128, 64, 64, 32, 32}, bn[] = {128, 128, 64, 64, 128};
  return static_cast<long long>((M + bm[c] - 1) / bm[c]) * ((N + bn[c] - 1) / bn[c]);
}

static int g_forced_splitk = 0;
void gemm_force_splitk(int sk) { g_forced_splitk = sk; }

// Split K when the output grid alone cannot fill the chip and K is long enough
// that every split still runs >= 8 K-steps.
int gemm_splitk(int M, int N, int K) {
  if (g_forced_splitk > 0) return (K / 64) >= g_forced_splitk ? g_forced_splitk : 1;
  // measured (benchmarks/bench_gemm.py): the extra reduce pass only pays for
  // weight-streaming shapes with a handful of rows (batched decode, verify)
  if (M > 32) return 1;
  const long long t = cfg_tiles(pick_cfg(M, N), M, N);
  if (t >= 256) return 1;
  // one block per CU.  Twice as many splits stream the weights 5-20 % faster in
  // isolation (6 rows: qkv 4.4 -> 5.3 TB/s), but the consumer then sums twice as
  // many slabs and a whole verify pass / batched decode step comes out 2-3 % slower.
  int sk = static_cast<int>((256 + t - 1) / t);
  const int max_by_k = (K / 64) / 8;
  if (sk > max_by_k) sk = max_by_k;
  if (sk > 16) sk = 16;
  return sk < 2 ? 1 : sk;
}

int gemm_argmax_partials(int M, int N) {
  switch (pick_cfg(M, N)) {  // tiles_n * WGN of the chosen config
    case CFG_128x128: case CFG_64x128: return ((N + 127) / 128) * 2;
    case CFG_64x64: return ((N + 63) / 64) * 2;
    case CFG_32x64: return ((N + 63) / 64) * 4;
    default: return ((N + 127) / 128) * 4;
  }
}

void launch_gemm_bf16(const bf16_t* A, const bf16_t* W, bf16_t* C, float* Cf, const bf16_t* bias,
                      const bf16_t* res, int M, int N, int K, int lda, int ldw, int ldc, int act,
                      float* amax_val, int* amax_idx, float* splitk_ws, int splitk,
                      hipStream_t st) {
  GemmParams p{A, W, C, bias, res, Cf, M, N, K, lda, ldw, ldc, act, amax_val, amax_idx, 1, K,
               nullptr, 0, K, K};
  if (splitk > 1 && splitk_ws && !amax_val) {
    const int ksteps = K / 64;
    p.splitk = splitk;
    p.k_per_split = ((ksteps + splitk - 1) / splitk) * 64;
    p.splitk = (K + p.k_per_split - 1) / p.k_per_split;  // drop empty trailing splits
    p.partial = splitk_ws;
  }
  switch (pick_cfg(M, N)) {
    case CFG_128x128: launch_cfg<128, 128, 2, 2, 2>(p, st); break;
    case CFG_64x128: launch_cfg<64, 128, 2, 2, 3>(p, st); break;
    case CFG_64x64: launch_cfg<64, 64, 2, 2, 4>(p, st); break;
    case CFG_32x64: launch_cfg<32, 64, 1, 4, 4>(p, st); break;
    default: launch_cfg<32, 128, 1, 4, 4>(p, st); break;
  }
  if (p.splitk > 1) {
    const long long quads = (static_cast<long long>(M) * N + 3) / 4;
    int blocks = static_cast<int>((quads + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
  }
}

// C[M,N] = opA · opW^T with K-strided operands: a_t -> A is stored [K][M] (lda = its row
// stride), w_t -> W is stored [K][N].  Used by the backward products of a linear layer.
// M (if a_t) and N (if w_t) must be multiples of 8 and >= 8; K a multiple of 64; ka_rows /
// kw_rows <= K are the k-rows the K-strided operands really have (see GemmParams).
template <bool AT, bool BT>
static void launch_t(const GemmParams& p, hipStream_t st) {
  auto tiles = [&](int bm, int bn) {
    return static_cast<long long>((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
  };
  if (tiles(128, 128) >= 384) launch_cfg<128, 128, 2, 2, 2, AT, BT>(p, st);
  else if (tiles(64, 128) >= 300) launch_cfg<64, 128, 2, 2, 3, AT, BT>(p, st);
  else launch_cfg<64, 64, 2, 2, 4, AT, BT>(p, st);
}

void launch_gemm_bf16_t(const bf16_t* A, const bf16_t* W, bf16_t* C, float* Cf, const bf16_t* bias,
                        const bf16_t* res, int M, int N, int K, int lda, int ldw, int ldc, int act,
                        bool a_t, bool w_t, int ka_rows, int kw_rows, hipStream_t st) {
  GemmParams p{A, W, C, bias, res, Cf, M, N, K, lda, ldw, ldc, act, nullptr, nullptr, 1, K, nullptr, 0,
               ka_rows, kw_rows};
  if (a_t && w_t) launch_t<true, true>(p, st);
  else if (a_t) launch_t<true, false>(p, st);
  else if (w_t) launch_t<false, true>(p, st);
  else launch_t<false, false>(p, st);
}

// Projection that stops at the fp32 split-K slabs partial[nslab][M][N]; returns
// nslab (>= 1).  `partial` holds gemm_splitk(M, N, K) * M * N floats.
int launch_gemm_partial(const bf16_t* A, const bf16_t* W, float* partial, int M, int N, int K,
                        int lda, int ldw, hipStream_t st) {
  GemmParams p{A, W, nullptr, nullptr, nullptr, nullptr, M, N, K, lda, ldw, N, ACT_NONE, nullptr,
               nullptr, 1, K, partial, 1, K, K};
  const int splitk = gemm_splitk(M, N, K);
  if (splitk > 1) {
    const int ksteps = K / 64;
    p.k_per_split = ((ksteps + splitk - 1) / splitk) * 64;
    p.splitk = (K + p.k_per_split - 1) / p.k_per_split;
  }
  switch (pick_cfg(M, N)) {
    case CFG_128x128: launch_cfg<128, 128, 2, 2, 2>(p, st); break;
    case CFG_64x128: launch_cfg<64, 128, 2, 2, 3>(p, st); break;
    case CFG_64x64: launch_cfg<64, 64, 2, 2, 4>(p, st); break;
    case CFG_32x64: launch_cfg<32, 64, 1, 4, 4>(p, st); break;
    default: launch_cfg<32, 128, 1, 4, 4>(p, st); break;
  }
  return p.splitk;
}

}  // namespace evgpt
