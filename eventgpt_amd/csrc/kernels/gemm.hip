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
//     (global_load_lds_dwordx4), double buffered: the loads of K-tile t+1 are
//     issued before the MFMAs of tile t, one barrier per K-tile;
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
  // optional row-argmax partials: [M][n_part] (value, index), n_part = tiles_n * WGN.
  // Values are the bf16-rounded outputs, so the result equals argmax over a
  // materialised bf16 C without ever writing it.
  float* amax_val;
  int* amax_idx;
};

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
  constexpr int WM = BM / WGM, WN = BN / WGN;  // per-wave tile
  constexpr int TM = WM / 16, TN = WN / 16;    // 16x16 tiles per wave
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  static_assert(WGM * WGN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  // consecutive tiles walk down M first: they share one weight panel
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wid / WGN, wc = wid % WGN;

  // ---- staging: one wave-instruction = 8 rows x 128 B, lane -> (row, chunk')
  const int srow = lane >> 3, schunk = lane & 7;
  auto stage = [&](int buf, int kt) {
    unsigned char* base = smem + buf * STAGE;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) {  // BM/8 row-groups over 4 waves
      const int rg = wid * (BM / 32) + i;
      const int row = rg * 8 + srow;
      const int c = schunk ^ ((row >> 1) & 7);
      int gr = m0 + row; gr = gr < p.M ? gr : p.M - 1;
      const bf16_t* src = p.A + static_cast<size_t>(gr) * p.lda + k0 + c * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + rg * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
      const int rg = wid * (BN / 32) + i;
      const int row = rg * 8 + srow;
      const int c = schunk ^ ((row >> 1) & 7);
      int gr = n0 + row; gr = gr < p.N ? gr : p.N - 1;
      const bf16_t* src = p.W + static_cast<size_t>(gr) * p.ldw + k0 + c * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + A_BYTES + rg * 1024), 16, 0, 0);
    }
  };

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int fsw = (frow >> 1) & 7;  // swizzle term: tile bases are multiples of 16 rows
  const int nk = p.K / BK;

  stage(0, 0);
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) & lgkmcnt(0)
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const unsigned char* a_base = smem + cur * STAGE + (wr * WM + frow) * 128;
    const unsigned char* b_base = smem + cur * STAGE + A_BYTES + (wc * WN + frow) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = ((ks * 4 + fq) ^ fsw) * 16;
      bf16x8_t af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const bf16x8_t*>(a_base + i * 16 * 128 + off);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const bf16x8_t*>(b_base + j * 16 * 128 + off);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();  // drains the direct-to-LDS loads of tile kt+1 and orders buffer reuse
    cur ^= 1;
  }

  // ---- epilogue: lane holds C[m][n .. n+3]
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
          v = bf16_to_f32(f32_to_bf16(v));
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

template <int BM, int BN, int WGM, int WGN>
static void launch_cfg(const GemmParams& p, hipStream_t st) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const size_t smem = 2 * (BM + BN) * BK * 2;
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WGM, WGN>), dim3(tiles), dim3(256), smem, st, p);
}

int gemm_argmax_partials(int M, int N) {
  // must mirror the tile choice in launch_gemm_bf16
  if (M <= 32) return ((N + 127) / 128) * 4;
  return ((N + 127) / 128) * 2;
}

void launch_gemm_bf16(const bf16_t* A, const bf16_t* W, bf16_t* C, float* Cf, const bf16_t* bias,
                      const bf16_t* res, int M, int N, int K, int lda, int ldw, int ldc, int act,
                      float* amax_val, int* amax_idx, hipStream_t st) {
  GemmParams p{A, W, C, bias, res, Cf, M, N, K, lda, ldw, ldc, act, amax_val, amax_idx};
  // Tile choice: fill 256 CUs.  Small M (prefill of a few hundred rows, verify
  // batches) uses shorter tiles so the grid is not a fraction of the chip.
  const long long t128 = static_cast<long long>((M + 127) / 128) * ((N + 127) / 128);
  if (M <= 32) launch_cfg<32, 128, 1, 4>(p, st);
  else if (M <= 64 || t128 < 256) launch_cfg<64, 128, 2, 2>(p, st);
  else launch_cfg<128, 128, 2, 2>(p, st);
}

}  // namespace evgpt
