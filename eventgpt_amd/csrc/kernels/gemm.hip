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
#include <type_traits>

#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

constexpr int BK = 64;

static int g_group_m = 8;             // row tiles per group of the kernels' tile walk (0: whole columns)
// split-K workgroups of the 4-wave kernel ordered split by split for more than 32 rows (636 x 4096 x 11008 in
// three splits: 61.9 -> 58.9 us; 64 x 128 tiles 93.9 -> 73.6); the few-row decode / verify products keep tile order
static int g_split_major = 1;
void gemm_split_major(int on) { g_split_major = on; }

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
  // Gated-linear-unit mode (K-contiguous operands, no K split): W is [gate rows | up rows], each
  // glu_half rows; the output has glu_half (= N) columns  silu(x·gate_n) * (x·up_n).  A block stages,
  // for every wave column, half of its tile rows from the gate region and the matching half from the
  // up region, so a lane holds both factors of its output columns and the intermediate [M, 2I]
  // tensor is never written.
  int glu_half;
  // K-contiguous operands only: the columns a row really has when that is not a multiple of 64
  // (0: K itself).  K is then the contraction rounded up to 64, and the last K-tile's chunks past
  // k_cols -- which in memory are the start of the next row -- are staged as zeros.
  int k_cols;
  // RoPE + cache-append epilogue of a q | k | v projection (ROPE instance of the 128 x 128 tile, head_dim
  // 128 so that a column tile is one head): rotated q rows go to rope_q [M][Hq * 128], rotated k and
  // plain v rows into the cache slices rope_k / rope_v [Hkv][l_max][128] at position pos0[0] + row.
  bf16_t* rope_q; bf16_t* rope_k; bf16_t* rope_v;
  const float* rope_cos; const float* rope_sin;     // [l_max][64]
  const int* rope_pos0;
  int rope_hq, rope_hkv, rope_lmax;
  // row tiles per group of the tile walk (0: whole columns); set by the launchers
  int group_m;
  int rows_epilogue;   // 8-wave kernel: bf16 output stored as whole lines through LDS (set by launch_256)
  // 4-wave kernel with split K: workgroup index = split * tiles + tile instead of tile * splits + split
  int split_major;
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

// Tile index -> (row tile, column tile).  Consecutive tiles walk down a group of about `group_m` row tiles, then
// move one column tile over, group after group (group_m <= 0 or few row tiles: whole columns).  The workgroups
// of an XCD that run side by side then cover a near-square patch of the output and share its operand panels
// in their L2 -- about 8 + 4 panels for 32 tiles instead of 32 + 1 when a tall output is walked column by
// column.  The groups are balanced (27 row tiles at group_m = 8: 7, 7, 7, 6).  Only from 3 group_m row tiles on:
// every group streams the whole weight matrix again, which a tall, compute-bound product takes from the
// die-level cache, while a few hundred rows are bound by exactly that stream (636-token prefill with groups of
// five 64-row tiles: 11.55 -> 11.89 ms; 8192^3 on the 8-wave kernel 1182 -> 1441 TFLOP/s, 4643 x 12288 x 4096 on
// 128 x 128 tiles 929 -> 1070, 19 row tiles of 256: no change either way).
__device__ __forceinline__ void tile_walk(int tile, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
  if (group_m > 0 && tiles_m >= 3 * group_m) {
    const int ngroups = (tiles_m + group_m - 1) / group_m;
    const int gsize = (tiles_m + ngroups - 1) / ngroups;
    const int per_group = gsize * tiles_n;
    const int g = tile / per_group, within = tile % per_group;
    const int first = g * gsize;
    const int rows = min(gsize, tiles_m - first);
    tm = first + within % rows;
    tn = within / rows;
  } else {
    tm = tile % tiles_m;
    tn = tile / tiles_m;
  }
}

// Epilogue shared by the GEMM kernels: lane holds C[m][n .. n+3] of each 16x16 tile (m = row
// lane & 15, n = 4 * (lane >> 4)); wave (wr, wc) owns the WM x WN sub-tile at (wr*WM, wc*WN).
template <int TM, int TN, int WM, int WN, int WGN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4_t (&acc)[TM][TN], int m0, int n0, int wr,
                                              int wc, int tn, int tiles_n, int ksplit, int lane) {
  const int frow = lane & 15, fq = lane >> 4;
  if constexpr (TN % 2 == 0) {
    if (p.glu_half) {
      // tile columns j < TN/2 are gate, j + TN/2 the matching up columns; both factors are rounded to
      // bf16 first (what a separate activation kernel would read back)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wr * WM + i * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN / 2; ++j) {
          const int n = n0 + wc * (WN / 2) + j * 16 + fq * 4;
          if (n >= p.N) continue;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            v[r] = silu_f(bf16_to_f32(f32_to_bf16(acc[i][j][r]))) * bf16_to_f32(f32_to_bf16(acc[i][j + TN / 2][r]));
          const size_t o = static_cast<size_t>(m) * p.ldc + n;
          if (n + 3 < p.N && (p.ldc & 3) == 0) {
            u32x2_t w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<u32x2_t*>(p.C + o) = w;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < p.N) p.C[o + r] = f32_to_bf16(v[r]);
          }
        }
      }
      return;
    }
  }
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

// bf16 output of the 8-wave kernel through LDS.  The MFMA layout gives a lane 4 consecutive columns of 16
// rows, so a direct store instruction writes 16 pieces of 32 bytes in 16 cache lines and a wave's 128-byte
// row segment arrives as four partial writes: a 256 x 256 tile is 4 096 of them, and with the stores left out
// the kernel runs 23 080 x 4 096 x 1 024 at 1 143 instead of 778 TFLOP/s.  Here a wave writes 64 rows of its
// 128 x 64 block to LDS (8 bytes per lane, 16-byte chunks XOR-swizzled by the row), reads them back as 16 bytes
// per lane -- 8 lanes per row -- and stores 8 rows of 128 contiguous bytes per instruction; the residual is
// added on the way out, read in the same whole lines.  The scratch is the wave's own staging slices (4 x 2 KB)
// of the ring buffer the next tile's K-tile 1 goes to: nobody else writes them, the wave itself stages into
// them after these stores have their data, and every reader is past the barrier that ended the K loop.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmParams& p, f32x4_t (&acc)[TM][TN], int m0, int n0,
                                                   int wr, int wc, int lane, unsigned char* scratch) {
  static_assert(TM == 8 && TN == 4, "128 x 64 per wave");
  const int frow = lane & 15, fq = lane >> 4;
  const int rl = lane >> 3, cl = lane & 7;          // read side: row within 8, 16-byte chunk of the 128-byte row
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int il = 0; il < 4; ++il) {
      const int i = pass * 4 + il;
      unsigned char* slice = scratch + il * 16384;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wc * 64 + j * 16 + fq * 4;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += bf16_to_f32(p.bias[n + r]);
        }
        if (p.act != ACT_NONE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
        }
        const u32x2_t w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        const int chunk = (j * 2 + (fq >> 1)) ^ (frow & 7);
        *reinterpret_cast<u32x2_t*>(slice + frow * 128 + chunk * 16 + (fq & 1) * 8) = w;
      }
    }
    // one wave's LDS operations execute in issue order; the compiler must keep them in this order too
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned char* slice = scratch + (q >> 1) * 16384;
      const int r = (q & 1) * 8 + rl;
      u32x4_t d = *reinterpret_cast<const u32x4_t*>(slice + r * 128 + ((cl ^ (r & 7)) << 4));
      const int m = m0 + wr * 128 + pass * 64 + q * 8 + rl;
      const int n = n0 + wc * 64 + cl * 8;
      if (m < p.M && n < p.N) {           // N % 8 == 0: a chunk is inside or outside as a whole
        const size_t o = static_cast<size_t>(m) * p.ldc + n;
        if (p.res) {
          const u32x4_t rv = *reinterpret_cast<const u32x4_t*>(p.res + o);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            d[e] = pack_bf16x2(bf16lo(d[e]) + bf16lo(rv[e]), bf16hi(d[e]) + bf16hi(rv[e]));
        }
        *reinterpret_cast<u32x4_t*>(p.C + o) = d;
      }
    }
    asm volatile("" ::: "memory");
  }
}

// 16 bytes per lane from `src + voff` (range-checked against `bytes`: past it, zeros) straight
// into LDS at `lds + 16 * lane`.  Kept out of the kernel template: the descriptor type exists
// only in the device pass, and a template body the host pass cannot instantiate loses its
// launch stub.
__device__ __forceinline__ void buffer_to_lds16(const unsigned char* src, int bytes, unsigned char* lds, int voff) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, 0, 0, 0);
}

// KT: a K-contiguous operand's rows end inside the last K-tile (GemmParams::k_cols > 0).  A template
// parameter, not a run-time test: with the extra staging path compiled into every instance the 64-row
// configurations ran 5-15 % slower on contractions that never needed it.
// RoPE epilogue for a 128 x 128 tile of 2 x 2 waves (each 64 x 64): the rotate-half partner of column d
// of a head is column d + 64, i.e. the same lane and register of the wave next door, so the two wave
// columns swap their bf16-rounded tiles through LDS (free once the K loop has ended) and each lane
// rotates and stores its own 4-column runs.  Same rounding and cache rules as rope_kv_prefill_kernel.
template <int TM, int TN>
__device__ __forceinline__ void gemm_rope_epilogue(const GemmParams& p, f32x4_t (&acc)[TM][TN], unsigned char* smem,
                                                   int m0, int n0, int wid, int lane) {
  static_assert(TM == 4 && TN == 4, "128 x 128 tile, 2 x 2 waves");
  const int wr = wid >> 1, wc = wid & 1, frow = lane & 15, fq = lane >> 4;
  const int head = n0 >> 7;
  const bool is_v = head >= p.rope_hq + p.rope_hkv;
  float* xch = reinterpret_cast<float*>(smem);        // [4 waves][64 values][64 lanes]
  __syncthreads();                                    // every wave is done with the operand tiles
  if (!is_v) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          xch[(wid * 64 + i * 16 + j * 4 + r) * 64 + lane] = bf16_to_f32(f32_to_bf16(acc[i][j][r]));
  }
  __syncthreads();
  const int pos_base = p.rope_pos0 ? p.rope_pos0[0] : 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wr * 64 + i * 16 + frow;
    if (m >= p.M) continue;
    int pos = pos_base + m;
    const bool in_cache = pos >= 0 && pos < p.rope_lmax;
    pos = pos < 0 ? 0 : (pos >= p.rope_lmax ? p.rope_lmax - 1 : pos);
    bf16_t* dst;
    if (head < p.rope_hq) dst = p.rope_q + (static_cast<size_t>(m) * p.rope_hq + head) * 128;
    else if (!in_cache) continue;
    else if (!is_v) dst = p.rope_k + (static_cast<size_t>(head - p.rope_hq) * p.rope_lmax + pos) * 128;
    else dst = p.rope_v + (static_cast<size_t>(head - p.rope_hq - p.rope_hkv) * p.rope_lmax + pos) * 128;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int dl = j * 16 + fq * 4;               // column inside this wave's half of the head
      float v[4];
      if (is_v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      } else {
        const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(p.rope_cos + static_cast<size_t>(pos) * 64 + dl);
        const f32x4_t sn = *reinterpret_cast<const f32x4_t*>(p.rope_sin + static_cast<size_t>(pos) * 64 + dl);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float own = bf16_to_f32(f32_to_bf16(acc[i][j][r]));
          const float oth = xch[((wid ^ 1) * 64 + i * 16 + j * 4 + r) * 64 + lane];
          v[r] = wc == 0 ? own * cs[r] - oth * sn[r] : own * cs[r] + oth * sn[r];
        }
      }
      const u32x2_t w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>(dst + wc * 64 + dl) = w;
    }
  }
}

template <int BM, int BN, int WGM, int WGN, int STAGES, bool AT = false, bool BT = false, bool KT = false,
          bool ROPE = false>
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
  const int bn_out = p.glu_half ? BN / 2 : BN;      // output columns per tile
  const int tiles_n = (p.N + bn_out - 1) / bn_out, tiles_m = (p.M + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n * p.splitk);
  // split K: tile-major (the splits of a tile are neighbours) or split-major (an XCD works inside one K range:
  // with a few hundred rows that range of A stays in its L2 while the weights stream past)
  const int ntile = tiles_m * tiles_n;
  const int tile = p.split_major ? bid % ntile : bid / p.splitk;
  const int ksplit = p.split_major ? bid / ntile : bid % p.splitk;
  const int k_begin = ksplit * p.k_per_split;
  const int k_len = min(p.k_per_split, p.K - k_begin);
  // consecutive tiles walk down (a group of) M first: they share one weight panel
  int tm, tn;
  tile_walk(tile, tiles_m, tiles_n, p.group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * bn_out;
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
      int src = n0 + row;
      if (p.glu_half) {    // wave column's tile rows: first half gate, second half the matching up rows
        const int wcol = row / WN, within = row % WN;
        src = (within >= WN / 2 ? p.glu_half - WN / 2 : 0) + n0 + wcol * (WN / 2) + within;
      }
      voff_w[i] = (src * p.ldw + c * 8) * 2;
    }
  }
  // bytes of each operand that exist in memory (everything past them reads as zero)
  const long long bytes_a = 2ll * (AT ? p.ka_rows : p.M) * p.lda;
  const long long bytes_w = 2ll * (BT ? p.kw_rows : (p.glu_half ? 2 * p.glu_half : p.N)) * p.ldw;
  auto stage = [&](int buf, int kt) {
    unsigned char* base = smem + buf * STAGE;
    const int k0 = k_begin + kt * BK;
    const long long adv_a = AT ? 2ll * k0 * p.lda : 2ll * k0;
    const long long adv_w = BT ? 2ll * k0 * p.ldw : 2ll * k0;
    const long long left_a = bytes_a - adv_a, left_w = bytes_w - adv_w;
    const int lim_a = static_cast<int>(left_a > 0 ? left_a : 0), lim_w = static_cast<int>(left_w > 0 ? left_w : 0);
    const unsigned char* src_a = reinterpret_cast<const unsigned char*>(p.A) + adv_a;
    const unsigned char* src_w = reinterpret_cast<const unsigned char*>(p.W) + adv_w;
    if (KT && k0 + BK > p.k_cols) {
      // the one K-tile that runs past the end of a K-contiguous operand's rows (block-uniform
      // branch): chunks at or past k_cols get an offset outside the descriptor's range
      constexpr int kOut = 0x7ffffff0;
#pragma unroll
      for (int i = 0; i < BM / 32; ++i) {
        const int rg = wid_u * (BM / 32) + i;
        int vo = voff_a[i];
        if constexpr (!AT) {
          const int row = rg * 8 + srow;
          if (k0 + (schunk ^ ((row >> 1) & 7)) * 8 >= p.k_cols) vo = kOut;
        }
        buffer_to_lds16(src_a, lim_a, base + rg * 1024, vo);
      }
#pragma unroll
      for (int i = 0; i < BN / 32; ++i) {
        const int rg = wid_u * (BN / 32) + i;
        int vo = voff_w[i];
        if constexpr (!BT) {
          const int row = rg * 8 + srow;
          if (k0 + (schunk ^ ((row >> 1) & 7)) * 8 >= p.k_cols) vo = kOut;
        }
        buffer_to_lds16(src_w, lim_w, base + A_BYTES + rg * 1024, vo);
      }
      return;
    }
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

  if constexpr (ROPE) {
    static_assert(BM == 128 && BN == 128 && WGM == 2 && WGN == 2 && STAGES * (BM + BN) * BK * 2 >= 65536, "rope tile");
    gemm_rope_epilogue<TM, TN>(p, acc, smem, m0, n0, wid, lane);
  } else {
    gemm_epilogue<TM, TN, WM, WN, WGN>(p, acc, m0, n0, wr, wc, tn, tiles_n, ksplit, lane);
  }
}

// ---------------------------------------------------------------------------------------
// Up to 32 rows against a weight matrix, fp32 split-K slabs out (batched decode steps, verify passes): the
// product is a stream of the weights.  The 32 x 64 configuration of the kernel above fetches 128 bytes of each
// of 64 weight rows per K-tile, the next 128 bytes of the same rows a K-tile later; this one takes K-tiles of
// 128, so a staging instruction reads 4 rows x 256 contiguous bytes (16 lanes per row; the 16-byte chunks of a
// row are permuted by chunk ^ (row & 15), which puts the 16 rows of an MFMA fragment read on 16 different
// 16-byte slots of the 256-byte bank row).  Three stages of 24 KB, two workgroups per CU.
constexpr int FR_BK = 128, FR_W_BYTES = 64 * FR_BK * 2;

// (a 16-row x image for up to 16 rows and four stages instead of three measure within 2 % of this; four stages
// with the 32-row image, one workgroup per CU, 10 % behind)
__global__ __launch_bounds__(256) void gemm_fewrow_kernel(GemmParams p) {
  constexpr int AM = 32, STAGES = 3;   // rows of the x image, K-tiles in the ring
  constexpr int A_BYTES = AM * FR_BK * 2, STAGE = A_BYTES + FR_W_BYTES;
  constexpr int TM = AM / 16;      // 16-row blocks of x
  constexpr int AI = AM / 16;      // x staging instructions per wave (4 rows each, shared by the four waves)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + 63) / 64;
  const int bid = xcd_remap(blockIdx.x, tiles_n * p.splitk);
  const int tile = bid / p.splitk, ksplit = bid % p.splitk;   // the splits of a tile are neighbours
  const int n0 = tile * 64;
  const int k_begin = ksplit * p.k_per_split;
  const int k_len = min(p.k_per_split, p.K - k_begin);
  const int nk = k_len / FR_BK;

  const int r4 = lane >> 4, slot = lane & 15;
  int voff_a[AI], voff_w[4];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = (wid * AI + i) * 4 + r4;
    voff_a[i] = (row * p.lda + ((slot ^ (row & 15)) * 8)) * 2;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wid * 4 + i) * 4 + r4;
    voff_w[i] = ((n0 + row) * p.ldw + ((slot ^ (row & 15)) * 8)) * 2;
  }
  // rows past M / N lie past the descriptor's range and arrive as zeros
  const long long bytes_a = 2ll * p.M * p.lda, bytes_w = 2ll * p.N * p.ldw;
  auto stage = [&](int buf, int kt) {
    unsigned char* base = smem + buf * STAGE;
    const long long adv = 2ll * (k_begin + kt * FR_BK);
    const long long left_a = bytes_a - adv, left_w = bytes_w - adv;
    const int lim_a = static_cast<int>(left_a > 0 ? left_a : 0), lim_w = static_cast<int>(left_w > 0 ? left_w : 0);
    const unsigned char* src_a = reinterpret_cast<const unsigned char*>(p.A) + adv;
    const unsigned char* src_w = reinterpret_cast<const unsigned char*>(p.W) + adv;
#pragma unroll
    for (int i = 0; i < 4; ++i) buffer_to_lds16(src_w, lim_w, base + A_BYTES + (wid * 4 + i) * 1024, voff_w[i]);
#pragma unroll
    for (int i = 0; i < AI; ++i) buffer_to_lds16(src_a, lim_a, base + (wid * AI + i) * 1024, voff_a[i]);
  };

  f32x4_t acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  constexpr int G = 4 + AI;   // staging instructions per wave and stage
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage(s, s);

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = min(STAGES - 2, nk - 1 - kt);   // younger tiles that may still be on their way
    if (ahead <= 0) wait_vm<0>();
    else if (ahead == 1) wait_vm<G>();
    else wait_vm<2 * G>();
    __builtin_amdgcn_s_barrier();
    if (kt + STAGES - 1 < nk) {
      int nb = cur + STAGES - 1;
      if (nb >= STAGES) nb -= STAGES;
      stage(nb, kt + STAGES - 1);
    }
    const unsigned char* a_img = smem + cur * STAGE + frow * 256;
    const unsigned char* w_img = smem + cur * STAGE + A_BYTES + (wid * 16 + frow) * 256;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = ((ks * 4 + fq) ^ frow) * 16;
      const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(w_img + off);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(a_img + i * 16 * 256 + off);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[i], 0, 0, 0);
      }
    }
    cur = cur + 1 == STAGES ? 0 : cur + 1;
  }

  // a lane holds 4 consecutive columns of one row: one 16-byte store per 16 x 16 block
  float* slab = p.partial + static_cast<size_t>(ksplit) * p.M * p.N;
  const int n = n0 + wid * 16 + fq * 4;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = i * 16 + frow;
    if (m < p.M && n < p.N) *reinterpret_cast<f32x4_t*>(slab + static_cast<size_t>(m) * p.N + n) = acc[i];
  }
}

static void launch_fewrow(const GemmParams& p, hipStream_t st) {
  constexpr size_t smem = 3 * (32 * FR_BK * 2 + FR_W_BYTES);
  static const bool attr = [] {   // 72 KB: more than a launch may ask for without opting in
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fewrow_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)) == hipSuccess;
  }();
  (void)attr;
  const int tiles_n = (p.N + 63) / 64;
  hipLaunchKernelGGL(gemm_fewrow_kernel, dim3(tiles_n * p.splitk), dim3(256), smem, st, p);
}

// ---------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 x 4, each 128 x 64 of C), one workgroup per CU: the kernel
// for GEMMs with thousands of rows.  The 4-wave kernel above stops near 900 TFLOP/s however it
// is pipelined, because one wave per SIMD has to alternate between feeding the LDS pipe and
// the MFMA pipe; here each SIMD holds two waves that run HALF A PHASE APART, so one issues
// MFMAs while the other issues its LDS reads and the staging loads.
//
// A K-tile is four phases; a phase is  [LDS reads + staging loads] barrier [16 MFMAs] barrier.
// The two wave rows (wr = 0 / 1) are offset by one barrier: wr = 1 passes one extra barrier
// before the loop and wr = 0 one after it, so between any two barriers one row is in its MFMA
// part and the other in its load part.
//
// LDS (128 KB) holds two K-tiles, each as four 16 KB images: A-sub0, A-sub1, W-sub0, W-sub1.
// "sub h" is what the phases consume together: A-sub h = rows {wr*128 + h*64 .. +64} of both wave
// rows (128 rows), W-sub h = columns {wc*64 + h*32 .. +32} of the four wave columns (128 rows of W).
//   phase 1: read W-sub0 + A-sub0      MFMA A0 x W0     stage A-sub1 of tile t+1
//   phase 2: read W-sub1               MFMA A0 x W1     stage W-sub0 of tile t+2
//   phase 3: read A-sub1               MFMA A1 x W1     stage A-sub0 of tile t+2
//   phase 4: (operands in registers)   MFMA A1 x W0     stage W-sub1 of tile t+2, then wait
// Ordering rules this schedule is built on (P = phase counter over the whole loop):
//   * an image read in phase P may be re-staged from phase P+1 on: every wave waits for its
//     LDS reads (lgkmcnt) before the first barrier of the phase it issued them in, and the
//     late wave row passes that barrier as the second barrier of the early row's phase P,
//     which the early row has behind it when it issues in phase P+1;
//   * staged data is read one phase after the wait that retires it: every wave waits
//     (counted vmcnt: the three youngest images stay in flight) before the first barrier of
//     its phase 4, and the reads of the next tile start after that phase's second barrier,
//     which no wave passes before both rows have done their wait.
// Loads past the end of K are issued anyway: the buffer descriptor's range check turns them
// into zero fills of images nobody reads, which keeps the vmcnt arithmetic uniform.
// BMT = 128: the same schedule on a 128 x 256 tile (each wave 64 x 64; the A images are 64 rows, one staging
// instruction per wave; 8 MFMAs per phase) for row counts that fill 256-row tiles badly.  K-contiguous A only.
template <bool AT, bool BT, int BMT = 256>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmParams p) {
  static_assert(BMT == 256 || (BMT == 128 && !AT), "128-row variant: K-contiguous A");
  constexpr int BM = BMT, BN = 256, A_HALF = BM * 64, W_HALF = 16384, TILE = 2 * A_HALF + 2 * W_HALF;
  constexpr int IS = BM / 64;          // 16-row tiles of one wave in one A image
  constexpr int LA = BM / 128;         // staging instructions per wave and A image
  constexpr int QR = BM / 4;           // rows of one wave row in one A image
  constexpr int TM = 2 * IS, TN = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int bn_out = p.glu_half ? BN / 2 : BN;      // gated mode: W-sub0 holds gate rows, W-sub1 the up rows
  const int tiles_n = (p.N + bn_out - 1) / bn_out, tiles_m = (p.M + BM - 1) / BM;
  const int total = tiles_m * tiles_n * p.splitk;

  // The grid is at most one workgroup per CU and a workgroup walks the tiles blockIdx.x, blockIdx.x + gridDim.x,
  // ... (gridDim.x is a multiple of 8 or the whole tile count, so a workgroup stays on its XCD's range of
  // xcd_remap): when a tile's K loop has ended the LDS is free, so the first K-tile of the NEXT tile is staged
  // before the epilogue and lands while the epilogue's stores drain.
  struct Tile {
    int m0, n0, tn, ksplit, k_begin, nk;
    int voff_a[2], voff_w[2];      // per-lane source offsets of sub 0; sub 1 and the K-tile move the descriptor base
  };
  auto locate = [&](int t, Tile& c) {
    const int bid = xcd_remap(t, total);
    const int tile = bid / p.splitk;
    c.ksplit = bid % p.splitk;
    c.k_begin = c.ksplit * p.k_per_split;
    c.nk = min(p.k_per_split, p.K - c.k_begin) / BK;
    int tm, tnn;
    tile_walk(tile, tiles_m, tiles_n, p.group_m, tm, tnn);
    c.tn = tnn;
    c.m0 = tm * BM;
    c.n0 = c.tn * bn_out;
    // ---- staging: image = 16 wave-instructions of 1 KB, two per wave (one for a 64-row A image)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rg = wid * 2 + i;
      if constexpr (AT) {      // image [64 k][128 columns]: columns = two runs of 64 (wave rows 0 / 1)
        const int lin = rg * 64 + lane;
        const int row = lin >> 4, cl = (lin & 15) ^ swz_t<128>(row);
        c.voff_a[i] = (row * p.lda + c.m0 + (cl >> 3) * 128 + (cl & 7) * 8) * 2;
      } else {                 // image [BM / 2 rows][64 k]: rows = two runs of QR (wave rows 0 / 1)
        const int lr = (wid * LA + (i < LA ? i : 0)) * 8 + (lane >> 3);
        const int cc = (lane & 7) ^ ((lr >> 1) & 7);
        c.voff_a[i] = ((c.m0 + (lr / QR) * (BM / 2) + (lr % QR)) * p.lda + cc * 8) * 2;
      }
      if constexpr (BT) {      // columns = four runs of 32 (wave columns 0..3)
        const int lin = rg * 64 + lane;
        const int row = lin >> 4, cl = (lin & 15) ^ swz_t<128>(row);
        c.voff_w[i] = (row * p.ldw + c.n0 + (cl >> 2) * 64 + (cl & 3) * 8) * 2;
      } else {
        const int lr = rg * 8 + (lane >> 3);
        const int cc = (lane & 7) ^ ((lr >> 1) & 7);
        c.voff_w[i] = ((c.n0 + (lr >> 5) * (p.glu_half ? 32 : 64) + (lr & 31)) * p.ldw + cc * 8) * 2;
      }
    }
  };
  const long long bytes_a = 2ll * (AT ? p.ka_rows : p.M) * p.lda;
  const long long bytes_w = 2ll * (BT ? p.kw_rows : (p.glu_half ? 2 * p.glu_half : p.N)) * p.ldw;
  // which: 0 = A-sub0, 1 = A-sub1, 2 = W-sub0, 3 = W-sub1; kt may run past nk (zero fill)
  auto stage_of = [&](const Tile& c, int which, int kt) {
    const int k0 = c.k_begin + kt * BK, h = which & 1;
    if (which < 2) {
      unsigned char* dst = smem + (kt & 1) * TILE + which * A_HALF + wid * (LA * 1024);
      const long long adv = AT ? 2ll * k0 * p.lda + 2ll * h * 64 : 2ll * k0 + 2ll * h * QR * p.lda;
      const long long left = bytes_a - adv;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(p.A) + adv;
      const int lim = static_cast<int>(left > 0 ? left : 0);
      buffer_to_lds16(src, lim, dst, c.voff_a[0]);
      if constexpr (LA == 2) buffer_to_lds16(src, lim, dst + 1024, c.voff_a[1]);
    } else {
      unsigned char* dst = smem + (kt & 1) * TILE + 2 * A_HALF + (which - 2) * W_HALF + wid * 2048;
      const long long adv = BT ? 2ll * k0 * p.ldw + 2ll * h * 32
                               : 2ll * k0 + 2ll * h * (p.glu_half ? p.glu_half : 32) * p.ldw;
      const long long left = bytes_w - adv;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(p.W) + adv;
      const int lim = static_cast<int>(left > 0 ? left : 0);
      buffer_to_lds16(src, lim, dst, c.voff_w[0]);
      buffer_to_lds16(src, lim, dst + 1024, c.voff_w[1]);
    }
  };
  Tile cur;
  int t = blockIdx.x;
  locate(t, cur);
  auto stage = [&](int which, int kt) { stage_of(cur, which, kt); };

  f32x4_t acc[TM][TN];

  const int frow = lane & 15, fq = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  // per-lane read offsets inside an image (K-contiguous operands), one per 32-wide k-step
  const int rd_a[2] = {(wr * QR + frow) * 128 + ((fq ^ fsw) << 4), (wr * QR + frow) * 128 + (((4 + fq) ^ fsw) << 4)};
  const int rd_w[2] = {(wc * 32 + frow) * 128 + ((fq ^ fsw) << 4), (wc * 32 + frow) * 128 + (((4 + fq) ^ fsw) << 4)};

  bf16x8_t af[2][IS][2], wf[2][2][2];     // [sub][16-row tile][k-step]
  auto read_a = [&](int buf, int h) {
    const unsigned char* img = smem + buf * TILE + h * A_HALF;
#pragma unroll
    for (int i = 0; i < IS; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (AT) af[h][i][ks] = frag_t<128>(img, wr * 64 + i * 16, ks, lane);
        else af[h][i][ks] = *reinterpret_cast<const bf16x8_t*>(img + rd_a[ks] + i * 2048);
      }
  };
  auto read_w = [&](int buf, int h) {
    const unsigned char* img = smem + buf * TILE + 2 * A_HALF + h * W_HALF;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (BT) wf[h][j][ks] = frag_t<128>(img, wc * 32 + j * 16, ks, lane);
        else wf[h][j][ks] = *reinterpret_cast<const bf16x8_t*>(img + rd_w[ks] + j * 2048);
      }
  };
  auto mma = [&](int ha, int hw) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < IS; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[ha * IS + i][hw * 2 + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[hw][j][ks], af[ha][i][ks], acc[ha * IS + i][hw * 2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // workgroup barrier that neither the compiler's passes nor its scheduler move code across
  auto fence = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads have returned
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // The barrier that opens a phase's MFMA part.  The fragments read in this phase are needed only after it, so
  // their latency is spent in the barrier: the wait for them follows it.  Only reads of an image that is
  // re-staged in the NEXT phase have to be back before the barrier (the staging row passes it before it
  // issues): KEEP = how many of the youngest reads may still be in flight there (-1: none need to be back).
  auto open_mma = [&](auto keep_c) {
    constexpr int KEEP = decltype(keep_c)::value;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KEEP == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if constexpr (KEEP == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    else if constexpr (KEEP == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // reads of A issued after the reads of W in phase 1 (K-contiguous A: one ds_read_b128 per fragment)
  constexpr int A_READS = AT ? 0 : IS * 2;

  auto tile_body = [&](auto buf_c, int kt) {
    constexpr int buf = decltype(buf_c)::value;
    // phase 1: W-sub0 is re-staged in phase 2, so its reads (issued first) are back before the barrier
    read_w(buf, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(buf, 0);
    stage(1, kt + 1);
    open_mma(std::integral_constant<int, A_READS>{});
    mma(0, 0);
    fence();
    // phase 2 (W-sub1 is re-staged in phase 4)
    read_w(buf, 1);
    stage(2, kt + 2);
    open_mma(std::integral_constant<int, -1>{});
    mma(0, 1);
    fence();
    // phase 3 (A-sub1 is re-staged in phase 1 of the next tile)
    read_a(buf, 1);
    stage(0, kt + 2);
    open_mma(std::integral_constant<int, -1>{});
    mma(1, 1);
    fence();
    // phase 4
    stage(3, kt + 2);
    wait_vm<4 + LA>();
    fence();
    mma(1, 0);
    fence();
  };

  // ---- first tile: K-tile 0 whole
  stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
  for (;;) {
    // the three images of K-tile 1 the steady state keeps in flight.  The wait below counts this wave's
    // outstanding vector-memory operations; loads complete in issue order among themselves, so once at most
    // these three images' loads are outstanding the four images of K-tile 0 have landed, whatever the
    // previous tile's epilogue stores (issued between the two groups) are doing.
    stage(2, 1); stage(0, 1); stage(3, 1);
    wait_vm<4 + LA>();
    fence();
    if (wr == 1) fence();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < cur.nk; kt += 2) {
      tile_body(std::integral_constant<int, 0>{}, kt);
      if (kt + 1 < cur.nk) tile_body(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (wr == 0) fence();   // both wave rows are past their last LDS read
    wait_vm<0>();           // zero fills past K have landed: the LDS can be staged again (or handed on)
    const int m0 = cur.m0, n0 = cur.n0, tn = cur.tn, ksplit = cur.ksplit;
    t += gridDim.x;
    const bool more = t < total;
    if (more) {
      locate(t, cur);
      stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
    }
    if constexpr (BMT == 256) {
      // plain bf16 output with whole 16-byte chunks per row: whole-line stores through LDS
      if (p.rows_epilogue) gemm_epilogue_rows<TM, TN>(p, acc, m0, n0, wr, wc, lane, smem + TILE + wid * 2048);
      else gemm_epilogue<TM, TN, BM / 2, 64, 4>(p, acc, m0, n0, wr, wc, tn, tiles_n, ksplit, lane);
    } else {
      gemm_epilogue<TM, TN, BM / 2, 64, 4>(p, acc, m0, n0, wr, wc, tn, tiles_n, ksplit, lane);
    }
    if (!more) break;
  }
}

static bool g_use_256 = true;
void gemm_enable_256(bool on) { g_use_256 = on; }
static bool g_persist_256 = true;     // false: one workgroup per tile (A/B runs)
void gemm_persist_256(bool on) { g_persist_256 = on; }
void gemm_group_m(int g) { g_group_m = g; }

template <bool AT, bool BT, int BM = 256>
static void launch_256(const GemmParams& p, hipStream_t st) {
  const int bn_out = p.glu_half ? 128 : 256;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + bn_out - 1) / bn_out) * p.splitk;
  constexpr int smem = 2 * (2 * BM * 64 + 2 * 16384);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<AT, BT, BM>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  // one workgroup per CU; more tiles than CUs are walked inside the kernel (a multiple of 8 workgroups, so
  // that a workgroup keeps its blockIdx.x % 8 -- its XCD's range of tiles -- from tile to tile)
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      n = 256;
    n = n / 8 * 8;
    return n < 8 ? 8 : n;
  }();
  const int grid = (g_persist_256 && tiles > cus) ? cus : tiles;
  GemmParams q = p;
  q.group_m = g_group_m;
  static const bool rows_on = [] { const char* e = getenv("EVGPT_GEMM_ROWS_EPILOGUE"); return !e || atoi(e) != 0; }();
  q.rows_epilogue = rows_on && BM == 256 && p.C && !p.Cf && !p.amax_val && !p.glu_half && p.splitk == 1 &&
                    !p.raw_partial && !p.rope_cos && p.N % 8 == 0 && p.ldc % 8 == 0 &&
                    (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 &&
                    (!p.res || (reinterpret_cast<uintptr_t>(p.res) & 15) == 0);
  hipLaunchKernelGGL((gemm256_kernel<AT, BT, BM>), dim3(grid), dim3(512), smem, st, q);
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

template <int BM, int BN, int WGM, int WGN, int STAGES, bool AT = false, bool BT = false, bool KT = false,
          bool ROPE = false>
static void launch_cfg(const GemmParams& p, hipStream_t st) {
  const int bn_out = p.glu_half ? BN / 2 : BN;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + bn_out - 1) / bn_out) * p.splitk;
  constexpr size_t smem = static_cast<size_t>(STAGES) * (BM + BN) * BK * 2;
  static bool attr_set = false;
  if (smem > 64 * 1024 && !attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, WGM, WGN, STAGES, AT, BT, KT, ROPE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
    attr_set = true;
  }
  GemmParams q = p;
  q.group_m = g_group_m;
  q.split_major = (g_split_major && p.splitk > 1 && p.M > 32) ? 1 : 0;
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WGM, WGN, STAGES, AT, BT, KT, ROPE>), dim3(tiles), dim3(256), smem,
                     st, q);
}

// Tile choice: the largest tile that still puts >= ~1.5 blocks on each of the 256
// CUs; shapes that cannot (prefill of a few hundred rows against a 1024- or
// 4096-wide matrix, verify batches) take smaller tiles with a deeper ring, since
// a grid below one block per CU is bound by the latency of its own K loop.
enum GemmCfg { CFG_128x128 = 0, CFG_64x128 = 1, CFG_64x64 = 2, CFG_32x64 = 3, CFG_32x128 = 4, CFG_256x256 = 5,
               CFG_128x256 = 6 };

static int g_forced_cfg = -1;
void gemm_force_config(int cfg) { g_forced_cfg = cfg; }

// The 8-wave kernel pays once there are rows to fill 256-row tiles and about half a round of
// tiles (measured on MI355X, TFLOP/s, 8-wave vs best 4-wave config: 2115 x 4096 x 11008, 144 tiles,
// 913 vs 751; 4643 x 12288 x 4096 1190 vs 930; 20352 x 4096 x 4096 1230 vs 905; but 636 rows 717 vs
// 948 and 2885 x 1024 x 4096, 48 tiles, 282 vs 567).
// K (0 = not known to the caller): a 256 x 256 tile's prologue and epilogue are one workgroup's alone (one per
// CU), so over a short contraction the 4-wave 128 x 128 tile -- two workgroups per CU, each hiding the other's
// ends -- is ahead.  With the grouped tile walk, TFLOP/s 8-wave vs 128 x 128: 23080 x 3072 x 1024 782 vs 816,
// 11540 x 3072 x 1024 674 vs 771, 4616 x 4096 x 1024 547 vs 707, 4616 x 4096 x 2048 717 vs 882, but
// 6688 x 4096 x 2048 980 vs 945, 23080 x 4096 x 2048 1102 vs 1003 and 4643 x 12288 x 4096 1253 vs 1070.
static bool use_256(int M, int N, int K = 0) {
  const long long t = static_cast<long long>((M + 255) / 256) * ((N + 255) / 256);
  // (Those were measured with the direct epilogue.  With whole-line stores through LDS -- gemm_epilogue_rows --
  // the 8-wave kernel is ahead over short contractions too once its tiles make more than half a round: 8-wave vs
  // 128 x 128, 2885 x 3072 x 1024 (144 tiles) 603 vs 508, 2885 x 4096 x 256 347 vs 291, 4616 x 3072 x 1024
  // 866 vs 667, 6688 x 4096 x 1024 961 vs 775, 11540 x 1024 x 1024 (184) 731 vs 634, 23080 x 4096 x 1024 879 vs
  // 653; behind below that: 6688 x 1024 x 1024 (108 tiles) 463 vs 552, 4616 x 1024 x 4096 (76) 485 vs 627.)
  if (K > 0 && !(K >= 4096 || (K >= 2048 && M >= 6144)) && t < 140) return false;
  // 768..1023 rows are three or four row tiles (up to a quarter of the last one empty): still ahead
  // once the tiles come near one round (836 x 12288 x 4096: 914 vs 833 TFLOP/s; K-strided W,
  // 836 x 11008 x 4096: 686 vs 582), behind for 64 tiles (836 x 4096: 338 vs 700)
  return g_use_256 && ((M >= 1024 && t >= 128) || (M >= 768 && t >= 160));
}

static int g_forced_splitk = 0;
static bool g_tail_splitk = true;
// set while gemm_splitk_slabs / launch_gemm_partial plan a product whose caller sums the slabs itself
static thread_local bool g_for_slabs = false;

// 768..1024 rows into a few thousand columns over a long contraction (one ~840-token sample against
// the 4096-wide matrices with K = 11008 / 12288 / 22016): 48..128 tiles of 256x256 are a fraction of a
// round, but with K split over the rest of the round the 8-wave kernel is ahead of every 4-wave
// choice (benchmarks/bench_gemm_splitk256.py, TFLOP/s, 4 splits vs the best 4-wave configuration:
// 836x4096x11008 878 vs 811, x12288 926 vs 844, x22016 1038 vs 871; at 636 rows -- three row tiles,
// a quarter of the last one used -- it is behind, 725 vs 807).  Returns the split count, or 0.
static int narrow_256_splits(int M, int N, int K) {
  if (!g_use_256 || g_forced_cfg >= 0 || g_forced_splitk > 0 || M < 768 || M > 1024 || K < 8192) return 0;
  const long long t = static_cast<long long>((M + 255) / 256) * ((N + 255) / 256);
  if (t < 48 || t > 128) return 0;
  const int s = static_cast<int>(256 / t);
  return s > 4 ? 4 : s;
}

// K = 0: the caller cannot split K (argmax partials, K-strided operands)
// k_real: the contraction length when the caller knows it (the tile choice depends on it), also where K = 0
static GemmCfg pick_cfg(int M, int N, int K = 0, int k_real = 0) {
  if (k_real == 0) k_real = K;
  if (g_forced_cfg >= 0) return static_cast<GemmCfg>(g_forced_cfg);
  auto tiles = [&](int bm, int bn) {
    return static_cast<long long>((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  };
  // thresholds from benchmarks/bench_gemm.py on MI355X
  if (M <= 32) return tiles(32, 128) >= 256 ? CFG_32x128 : CFG_32x64;
  if (use_256(M, N, k_real)) return CFG_256x256;
  if (K && narrow_256_splits(M, N, K) >= 2) return CFG_256x256;
  // Narrow outputs over a long contraction (a few hundred rows into 4096 columns over 11008, the
  // ViT's 4096 -> 1024 projection): the larger tile, its grid filled by splitting K (gemm_splitk)
  // instead of by shrinking the tile.  636 x 4096 x 11008: 128x128 with 3 splits 806 TFLOP/s,
  // 64x128 unsplit 608; 577 x 1024 x 4096: 64x128 with 3 splits 237, 32x64 unsplit 184.
  if (M <= 1024 && g_forced_splitk == 0) {
    // (a caller that consumes the fp32 slabs itself -- g_for_slabs -- pays no reduction pass, and the same
    // choice wins from K = 4096 on: 636 x 4096 x 4096 as 160 tiles x 3 splits 28 us, 64 x 128 unsplit 36.7;
    // with the reduction pass it loses there, 519 vs 571 TFLOP/s)
    if (K >= (g_for_slabs ? 4096 : 8192) && tiles(128, 128) >= 96 && tiles(128, 128) < 256) return CFG_128x128;
    if (K >= 4096 && tiles(64, 128) >= 48 && tiles(64, 128) < 128) return CFG_64x128;
  }
  if (tiles(128, 128) >= 384) return CFG_128x128;
  if (tiles(64, 128) >= 300) return CFG_64x128;
  if (tiles(64, 64) >= 300) return CFG_64x64;
  return CFG_32x64;
}

static long long cfg_tiles(GemmCfg c, int M, int N) {
  static const int bm[] = {128, 64, 64, 32, 32, 256, 128}, bn[] = {128, 128, 64, 64, 128, 256, 256};
  return static_cast<long long>((M + bm[c] - 1) / bm[c]) * ((N + bn[c] - 1) / bn[c]);
}

void gemm_force_splitk(int sk) { g_forced_splitk = sk; }

// Split K when the output grid alone cannot fill the chip and K is long enough
// that every split still runs >= 8 K-steps.
int gemm_splitk(int M, int N, int K) {
  if (g_forced_splitk > 0) return (K / 64) >= g_forced_splitk ? g_forced_splitk : 1;
  // measured (benchmarks/bench_gemm.py): the extra reduce pass only pays for
  // weight-streaming shapes with a handful of rows (batched decode, verify)
  if (M > 32) {
    if (M > 1024) {
      // A single partial round of 256 x 256 tiles over a long contraction: with K split s ways the tiles
      // make ceil(t s / 256) rounds of 1 / s the length.  Taken when that model promises a third off and
      // s <= 3 (the fp32 slabs cost about one more pass over the output per split): 2115 x 4096 x 11008,
      // 144 tiles, s = 3: 231-262 -> 209 us; 1500 rows, 96 tiles, s = 2: 152 -> 130 us; 2885 rows, 192
      // tiles: every split count measures behind and the model agrees.
      if (g_tail_splitk && K >= 8192 && use_256(M, N, K)) {
        const long long t = static_cast<long long>((M + 255) / 256) * ((N + 255) / 256);
        if (t < 256) {
          for (int s = 2; s <= 3; ++s) {
            const long long rounds = (t * s + 255) / 256;
            if (3 * rounds <= 2 * s) return s;     // rounds / s <= 2 / 3
          }
        }
      }
      return 1;
    }
    if (!use_256(M, N, K) && narrow_256_splits(M, N, K) >= 2) return narrow_256_splits(M, N, K);
    const GemmCfg cfg = pick_cfg(M, N, K);
    const long long tl = cfg_tiles(cfg, M, N);
    int s = 1;   // fill the co-resident slots (2 blocks of 128x128, 3 of 64x128 per CU)
    if (tl <= 32 && K >= 2048) {
      // a narrow output over a long contraction (low-rank projections: hundreds of rows into 16
      // columns) is a few dozen blocks walking the whole K: split until the grid is a few hundred
      // blocks, keeping >= 4 K-steps per split
      const long long want = 256 / tl, most = K / 256;
      s = static_cast<int>(want < most ? want : most);
      return s < 1 ? 1 : (s > 16 ? 16 : s);
    }
    if (cfg == CFG_128x128 && K >= (g_for_slabs ? 4096 : 8192) && tl >= 96 && tl < 256) s = static_cast<int>(512 / tl);
    else if (cfg == CFG_64x128 && K >= 4096 && tl >= 48 && tl < 128) s = static_cast<int>(256 / tl);
    return s < 1 ? 1 : (s > 4 ? 4 : s);
  }
  const long long t = cfg_tiles(pick_cfg(M, N), M, N);
  if (t >= 512) return 1;
  // At least one block per CU, and at least four splits while the tiles are fewer than two per CU
  // (the 7B decoder's qkv and gate/up: 192 and 344 tiles of 32 x 64): batched decode steps with every
  // projection split four ways against one block per CU (qkv 2, gate/up unsplit), ms per step on one box:
  // 8 streams 4.08 -> 3.84, 16 streams 4.61 -> 4.39, 32 streams 5.75 -> 5.50-5.58.  More than that loses
  // again (six or eight splits everywhere: 6.1-6.2 at 32 streams): the consumers sum every slab, and
  // their all-loads-first path exists for 1, 2, 4 and 8 slabs only (three or six for one projection:
  // 5.68-5.77 against 5.66 at 32 streams, 3.86-3.93 against 3.80 ms per verify pass of five drafts).
  int sk = static_cast<int>((256 + t - 1) / t);
  if (sk < 4) sk = 4;
  const int max_by_k = (K / 64) / 8;
  if (sk > max_by_k) sk = max_by_k;
  if (sk > 16) sk = 16;
  return sk < 2 ? 1 : sk;
}

// One workgroup per CU means whole rounds of 256 tiles.  When the 8-wave kernel's last round would be
// under half full (and there are only one or two full rounds for it to hide behind), the columns of the
// whole rounds (n1) go first and the remaining ones are a second launch: on the 4-wave kernel, or --
// *tail_s >= 2, contraction long enough -- on the 8-wave kernel again with K split tail_s ways so that
// the few tiles spread over the idle CUs (4643 x 4096 x 11008: 304 tiles = 247 + 57 x 4 splits).
static bool last_round_split(int M, int N, int K, int* n1, int* tail_s) {
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
  const long long total = static_cast<long long>(tiles_m) * tiles_n;
  const int rounds = static_cast<int>(total / 256), rem = static_cast<int>(total % 256);
  const int cols1 = rounds >= 1 ? (rounds * 256) / tiles_m : 0;     // column tiles of the whole rounds
  if (!(rounds >= 1 && rounds <= 2 && rem > 0 && rem < 128 && cols1 >= 1 && cols1 < tiles_n)) return false;
  *n1 = cols1 * 256;
  const int rem_tiles = tiles_m * (tiles_n - cols1);
  int s = 256 / rem_tiles;
  if (s > 4) s = 4;
  // two splits (86..128 leftover tiles) measure level with or behind the 4-wave kernel on those columns
  // (811 x 22016 x 4096: 160 vs 157 us, 1000 rows: 171 vs 164); three or four are ahead
  *tail_s = (K >= 4096 && K % 64 == 0 && s >= 3 && g_tail_splitk) ? s : 0;
  return true;
}

// fp32 workspace (floats) the last-round split of launch_gemm_bf16 wants for this shape, 0 if none
long long gemm_tail_ws_floats(int M, int N, int K) {
  if (g_forced_cfg >= 0 || g_forced_splitk > 0 || !use_256(M, N, K)) return 0;
  int n1 = 0, s = 0;
  if (!last_round_split(M, N, K, &n1, &s) || s < 2) return 0;
  return static_cast<long long>(s) * M * (N - n1);
}
void gemm_enable_tail_splitk(bool on) { g_tail_splitk = on; }

int gemm_argmax_partials(int M, int N) {
  switch (pick_cfg(M, N)) {  // tiles_n * WGN of the chosen config
    case CFG_256x256: case CFG_128x256: return ((N + 255) / 256) * 4;
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
  // a contraction that is not a multiple of 64 (a multiple of 8: 16-byte chunks) runs to the next
  // one, its last K-tile staged with zeros past the rows' real end (GemmParams::k_cols)
  const int k_true = K;
  K = (K + 63) / 64 * 64;
  GemmParams p{A, W, C, bias, res, Cf, M, N, K, lda, ldw, ldc, act, amax_val, amax_idx, 1, K,
               nullptr, 0, K, K};
  p.k_cols = k_true == K ? 0 : k_true;
  if (splitk > 1 && splitk_ws && !amax_val) {
    const int ksteps = K / 64;
    p.splitk = splitk;
    p.k_per_split = ((ksteps + splitk - 1) / splitk) * 64;
    p.splitk = (K + p.k_per_split - 1) / p.k_per_split;  // drop empty trailing splits
    p.partial = splitk_ws;
  }
  GemmCfg cfg = pick_cfg(M, N, (splitk_ws && !amax_val) ? K : 0, amax_val ? 0 : K);
  if (p.k_cols) {
    // short last K-tile: the two instances compiled with the masked staging path (these are the
    // low-rank products and other short contractions; the tile choice matters little for them)
    if (M <= 32) launch_cfg<32, 64, 1, 4, 4, false, false, true>(p, st);
    else launch_cfg<64, 128, 2, 2, 3, false, false, true>(p, st);
    if (p.splitk > 1) {
      const long long quads = (static_cast<long long>(M) * N + 3) / 4;
      int blocks = static_cast<int>((quads + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    }
    return;
  }
  // One workgroup per CU means whole rounds of 256 tiles: a grid of 304 tiles takes as long as
  // one of 512.  When the last round would be under half full (and there are few rounds for it
  // to hide in), the 8-wave kernel takes the columns that make whole rounds and the 4-wave
  // kernel, several blocks per CU, the remaining ones.
  if (cfg == CFG_256x256 && g_forced_cfg < 0 && p.splitk == 1 && !amax_val) {
    int n1 = 0, tail_s = 0;
    if (last_round_split(M, N, K, &n1, &tail_s)) {
      GemmParams a = p, b = p;
      a.N = n1;
      b.N = N - n1;
      b.W = W + static_cast<size_t>(n1) * ldw;
      if (C) b.C = C + n1;
      if (Cf) b.Cf = Cf + n1;
      if (bias) b.bias = bias + n1;
      if (res) b.res = res + n1;
      launch_256<false, false>(a, st);
      if (tail_s >= 2 && splitk_ws) {
        // the few tiles of the last round stay on the 8-wave kernel, their K split over the idle CUs
        const int ksteps = K / 64;
        b.k_per_split = ((ksteps + tail_s - 1) / tail_s) * 64;
        b.splitk = (K + b.k_per_split - 1) / b.k_per_split;
        b.partial = splitk_ws;
        launch_256<false, false>(b, st);
        const long long quads = (static_cast<long long>(b.M) * b.N + 3) / 4;
        int blocks = static_cast<int>((quads + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, b);
      } else {
        launch_cfg<128, 128, 2, 2, 2>(b, st);
      }
      return;
    }
  }
  switch (cfg) {
    case CFG_128x128: launch_cfg<128, 128, 2, 2, 2>(p, st); break;
    case CFG_64x128: launch_cfg<64, 128, 2, 2, 3>(p, st); break;
    case CFG_64x64: launch_cfg<64, 64, 2, 2, 4>(p, st); break;
    case CFG_32x64: launch_cfg<32, 64, 1, 4, 4>(p, st); break;
    case CFG_256x256: launch_256<false, false>(p, st); break;
    case CFG_128x256: launch_256<false, false, 128>(p, st); break;
    default: launch_cfg<32, 128, 1, 4, 4>(p, st); break;
  }
  if (p.splitk > 1) {
    const long long quads = (static_cast<long long>(M) * N + 3) / 4;
    int blocks = static_cast<int>((quads + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
  }
}

// q | k | v projection of a prefill with RoPE and the cache append in the epilogue (head_dim 128): A [M][K],
// W [(Hq + 2 Hkv) * 128][K]; rotated q -> q_out [M][Hq * 128], k / v -> cache slices [Hkv][l_max][128] at
// pos0[0] + row.  Returns false (nothing launched) for shapes the fused tile does not take.
bool launch_gemm_qkv_rope(const bf16_t* A, const bf16_t* W, int M, int K, bf16_t* q_out, bf16_t* k_cache,
                          bf16_t* v_cache, const float* cos_tab, const float* sin_tab, const int* pos0, int Hq,
                          int Hkv, int l_max, hipStream_t st) {
  const int N = (Hq + 2 * Hkv) * 128;
  if (K % 64 || M < 1 || g_forced_cfg >= 0) return false;
  GemmParams p{A, W, nullptr, nullptr, nullptr, nullptr, M, N, K, K, K, N, ACT_NONE, nullptr, nullptr, 1, K,
               nullptr, 0, K, K};
  p.rope_q = q_out; p.rope_k = k_cache; p.rope_v = v_cache; p.rope_cos = cos_tab; p.rope_sin = sin_tab;
  p.rope_pos0 = pos0; p.rope_hq = Hq; p.rope_hkv = Hkv; p.rope_lmax = l_max;
  launch_cfg<128, 128, 2, 2, 2, false, false, false, true>(p, st);
  return true;
}

// dX = dY W of a few hundred tokens is a narrow output (tokens x 4096) over a contraction of
// 12288 or 22016: the 128 x 128 tile with K split over the co-resident slots, as in pick_cfg.
int gemm_splitk_t(int M, int N, int K) {
  if (g_forced_splitk > 0) return (K / 64) >= g_forced_splitk ? g_forced_splitk : 1;
  const long long t = static_cast<long long>((M + 127) / 128) * ((N + 127) / 128);
  if (g_forced_cfg >= 0 || M > 1024) return 1;
  if (!use_256(M, N, K) && narrow_256_splits(M, N, K) >= 2) return narrow_256_splits(M, N, K);
  if (t <= 32 && K >= 2048) {   // narrow output, long contraction (input gradient of a low-rank B): as in gemm_splitk
    const long long want = 256 / t, most = K / 256;
    const int s = static_cast<int>(want < most ? want : most);
    return s < 1 ? 1 : (s > 16 ? 16 : s);
  }
  if (K < 8192 || t < 96 || t >= 256) return 1;
  const int s = static_cast<int>(512 / t);
  return s < 1 ? 1 : (s > 4 ? 4 : s);
}

// H[M, I] = silu(A · Wgate^T) * (A · Wup^T) with W = [gate rows | up rows] ([2I, K]): the gate/up
// projection of a SwiGLU MLP with the activation applied in the epilogue (GemmParams::glu_half).
void launch_gemm_glu(const bf16_t* A, const bf16_t* W, bf16_t* H, int M, int I, int K, int lda, int ldw, int ldc,
                     hipStream_t st) {
  GemmParams p{A, W, H, nullptr, nullptr, nullptr, M, I, K, lda, ldw, ldc, ACT_NONE, nullptr, nullptr, 1, K,
               nullptr, 0, K, K};
  p.glu_half = I;
  auto tiles = [&](int bm, int bn) { return static_cast<long long>((M + bm - 1) / bm) * ((2 * I + bn - 1) / bn); };
  if (g_forced_cfg < 0 && use_256(M, 2 * I, K) && g_tail_splitk && I % 128 == 0) {
    // a handful of 256 x 128 output tiles past the last full round (2115 rows x 11008: 774 = 3 x 256 + 6)
    // would cost a whole further round of the 8-wave kernel: their columns go to 64 x 64 tiles instead
    const int tiles_m = (M + 255) / 256, tiles_n = I / 128;
    const long long total = static_cast<long long>(tiles_m) * tiles_n;
    const int rounds = static_cast<int>(total / 256), rem = static_cast<int>(total % 256);
    const int cols1 = rounds >= 1 ? (rounds * 256) / tiles_m : 0;
    if (rounds >= 1 && rem > 0 && rem <= 48 && cols1 >= 1 && cols1 < tiles_n) {
      const int n1 = cols1 * 128;
      GemmParams a = p, b = p;
      a.N = n1;
      b.N = I - n1;
      b.W = W + static_cast<size_t>(n1) * ldw;     // gate rows n1.., up rows glu_half + n1.. relative to it
      b.C = H + n1;
      launch_256<false, false>(a, st);
      launch_cfg<64, 64, 2, 2, 4>(b, st);
      return;
    }
  }
  if (g_forced_cfg == CFG_256x256 || (g_forced_cfg < 0 && use_256(M, 2 * I, K))) launch_256<false, false>(p, st);
  else if (g_forced_cfg == CFG_128x128 || (g_forced_cfg < 0 && tiles(128, 128) >= 384)) launch_cfg<128, 128, 2, 2, 2>(p, st);
  else if (g_forced_cfg == CFG_64x64) launch_cfg<64, 64, 2, 2, 4>(p, st);
  else launch_cfg<64, 128, 2, 2, 3>(p, st);
}

// C[M,N] = opA · opW^T with K-strided operands: a_t -> A is stored [K][M] (lda = its row
// stride), w_t -> W is stored [K][N].  Used by the backward products of a linear layer.
// M (if a_t) and N (if w_t) must be multiples of 8 and >= 8; K a multiple of 64; ka_rows /
// kw_rows <= K are the k-rows the K-strided operands really have (see GemmParams); a K-contiguous
// operand has lda / ldw <= K columns (a multiple of 8), staged as zeros past them.
template <bool AT, bool BT>
static void launch_t(const GemmParams& p, hipStream_t st) {
  auto tiles = [&](int bm, int bn) {
    return static_cast<long long>((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
  };
  const bool cols_ok = (!AT || p.M % 8 == 0) && (!BT || p.N % 8 == 0);
  if constexpr (!(AT && BT)) {
    if (p.k_cols) {   // short last K-tile of the K-contiguous operand: the instance with the masked staging path
      launch_cfg<64, 128, 2, 2, 3, AT, BT, true>(p, st);
      return;
    }
  }
  if (p.splitk > 1 && cols_ok && !use_256(p.M, p.N, p.K) && narrow_256_splits(p.M, p.N, p.K) >= 2) launch_256<AT, BT>(p, st);
  else if (p.splitk > 1) launch_cfg<128, 128, 2, 2, 2, AT, BT>(p, st);
  else if ((g_forced_cfg == CFG_256x256 || (g_forced_cfg < 0 && use_256(p.M, p.N, p.K))) && cols_ok) launch_256<AT, BT>(p, st);
  else if (tiles(128, 128) >= 384) launch_cfg<128, 128, 2, 2, 2, AT, BT>(p, st);
  else if (tiles(64, 128) >= 300) launch_cfg<64, 128, 2, 2, 3, AT, BT>(p, st);
  else launch_cfg<64, 64, 2, 2, 4, AT, BT>(p, st);
}

void launch_gemm_bf16_t(const bf16_t* A, const bf16_t* W, bf16_t* C, float* Cf, const bf16_t* bias,
                        const bf16_t* res, int M, int N, int K, int lda, int ldw, int ldc, int act,
                        bool a_t, bool w_t, int ka_rows, int kw_rows, hipStream_t st, float* splitk_ws, int splitk) {
  GemmParams p{A, W, C, bias, res, Cf, M, N, K, lda, ldw, ldc, act, nullptr, nullptr, 1, K, nullptr, 0,
               ka_rows, kw_rows};
  // the K-contiguous operand (at most one here) may end before the padded contraction does
  const int cols = !a_t ? lda : (!w_t ? ldw : K);
  p.k_cols = cols < K ? cols : 0;
  if (splitk > 1 && splitk_ws) {
    const int ksteps = K / 64;
    p.k_per_split = ((ksteps + splitk - 1) / splitk) * 64;
    p.splitk = (K + p.k_per_split - 1) / p.k_per_split;
    p.partial = splitk_ws;
  }
  if (a_t && w_t) launch_t<true, true>(p, st);
  else if (a_t) launch_t<true, false>(p, st);
  else if (w_t) launch_t<false, true>(p, st);
  else launch_t<false, false>(p, st);
  if (p.splitk > 1) {
    const long long quads = (static_cast<long long>(M) * N + 3) / 4;
    int blocks = static_cast<int>((quads + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
  }
}

// Projection that stops at the fp32 split-K slabs partial[nslab][M][N]; returns
// nslab (>= 1).  `partial` holds gemm_splitk(M, N, K) * M * N floats.
// K splits of launch_gemm_partial (its caller sums the slabs, so a split costs no extra pass)
int gemm_splitk_slabs(int M, int N, int K) {
  g_for_slabs = true;
  const int s = gemm_splitk(M, N, K);
  g_for_slabs = false;
  return s;
}

int launch_gemm_partial(const bf16_t* A, const bf16_t* W, float* partial, int M, int N, int K,
                        int lda, int ldw, hipStream_t st, int max_split) {
  GemmParams p{A, W, nullptr, nullptr, nullptr, nullptr, M, N, K, lda, ldw, N, ACT_NONE, nullptr,
               nullptr, 1, K, partial, 1, K, K};
  int splitk = gemm_splitk_slabs(M, N, K);
  if (max_split > 0 && splitk > max_split) splitk = max_split;
  // up to 32 rows: the weight-streaming kernel (EVGPT_FEWROW_GEMM=0: the 32 x 64 tile of the general kernel).
  // Per layer of the 7B decoder at 32 / 6 rows: 88.3 -> 84.1 / 80.4 -> 77.1 us; decode step at 8 / 32 streams
  // 3.82 -> 3.70 / 5.50 -> 5.39 ms
  static const bool fewrow = [] { const char* e = getenv("EVGPT_FEWROW_GEMM"); return !(e && e[0] == '0'); }();
  if (fewrow && M <= 32 && g_forced_cfg < 0 && K % FR_BK == 0 && N % 4 == 0 && lda % 8 == 0 && ldw % 8 == 0) {
    const int ksteps = K / FR_BK;
    if (splitk > ksteps) splitk = ksteps;
    p.k_per_split = ((ksteps + splitk - 1) / splitk) * FR_BK;
    p.splitk = (K + p.k_per_split - 1) / p.k_per_split;
    launch_fewrow(p, st);
    return p.splitk;
  }
  if (splitk > 1) {
    const int ksteps = K / 64;
    p.k_per_split = ((ksteps + splitk - 1) / splitk) * 64;
    p.splitk = (K + p.k_per_split - 1) / p.k_per_split;
  }
  // more than 32 rows: the tile the split count was planned for (gemm_splitk picks with K in view)
  g_for_slabs = true;
  const GemmCfg cfg = M > 32 ? pick_cfg(M, N, K) : pick_cfg(M, N);
  g_for_slabs = false;
  switch (cfg) {
    case CFG_128x128: launch_cfg<128, 128, 2, 2, 2>(p, st); break;
    case CFG_64x128: launch_cfg<64, 128, 2, 2, 3>(p, st); break;
    case CFG_64x64: launch_cfg<64, 64, 2, 2, 4>(p, st); break;
    case CFG_32x64: launch_cfg<32, 64, 1, 4, 4>(p, st); break;
    case CFG_256x256: launch_256<false, false>(p, st); break;
    case CFG_128x256: launch_256<false, false, 128>(p, st); break;
    default: launch_cfg<32, 128, 1, 4, 4>(p, st); break;
  }
  return p.splitk;
}

}  // namespace evgpt
