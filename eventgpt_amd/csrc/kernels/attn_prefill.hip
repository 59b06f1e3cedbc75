// Fused (flash-style) attention for many query rows: LLM prefill (causal,
// head_dim 128), speculative-verify batches (causal with an offset into a long
// cache) and the ViT encoder (non-causal, head_dim 64, 577 tokens).
//
// One block = 4 waves = 64 query rows of one (batch, head); each wave owns 16
// rows.  K/V tiles of 64 keys are staged through LDS (register-staged: the
// global loads of tile t+1 are issued before the MFMAs of tile t and written to
// LDS after the barrier).  The score tile is computed TRANSPOSED,
//     S^T[key][q] = K · Q^T        (K rows = MFMA A operand, Q^T = B operand)
// so that the query index lives on the lane (lane & 15) and a lane's 16
// accumulator values are 16 different keys of ONE query.  Consequences:
//   * softmax row statistics are in-lane reductions + 2 cross-lane shuffles;
//   * the exponentiated tile, converted to bf16, already *is* the B operand of
//     the second product  O^T[d][q] = V^T · P^T  — no LDS round trip for P
//     (the contraction index is permuted consistently on both operands);
//   * V^T fragments come from the row-major V tile with the hardware
//     transposing read ds_read_b64_tr_b16, so V is staged exactly like K;
//   * rescaling O by exp(m_old - m_new) is a per-lane scalar multiply.
// LDS images are XOR-swizzled at 16-byte-chunk granularity so both the row
// reads (ds_read_b128) and the transposed reads spread over the banks.
#include "common.h"
#include "attn_common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

struct AttnParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
  long long q_sb, q_ss, q_sh;   // element strides: batch, sequence, head
  long long k_sb, k_ss, k_sh;
  long long v_sb, v_ss, v_sh;
  long long o_sb, o_ss, o_sh;
  const int* kv_len;            // optional device-side key count per batch
  int Sq, Sk, Hq, Hkv;
  int q_offset;                 // absolute position of query row 0 minus key 0 (causal)
  int q_offset_from_len;        // if set: q_offset = kv_len[b] - Sq
  float scale_log2e;
  // split-KV mode (few query rows against a long cache): block x = key split
  float* part_o;                // [B][Hq][nsplit][Sq][D] unnormalised fp32
  float* part_ml;               // [B][Hq][nsplit][Sq][2] (running max in log2 units, sum)
  int nsplit;
  float* lse;                   // optional [B][Hq][Sq]: log2 of the row's softmax denominator in the
                                // scaled-log2 domain (max + log2(sum)); the backward recomputes P from it
};

// GROUPS = 2: the block is two such groups of 4 waves over the SAME 64 query rows; group 0 takes the
// even key tiles of the block's range and group 1 the odd ones, each through its own K / V tile
// buffers, and group 1's running (max, sum, O) is merged into group 0's at the end through LDS.  A
// few hundred query rows of a few heads are fewer blocks than CUs, each a serial chain over its key
// tiles; two groups halve the chain and double the waves (used for the ViT, see launch_attn_prefill).
template <int D, bool CAUSAL, bool SPLIT = false, int GROUPS = 1>
__global__ __launch_bounds__(256 * GROUPS) void attn_prefill_kernel(AttnParams p) {
  constexpr int CPR = D / 8;                  // 16-byte chunks per row
  constexpr int TILE_BYTES = TK * D * 2;
  constexpr int LD_PER_THREAD = TK * CPR / 256;  // chunks each thread stages per tile
  constexpr int KS = D / 32;                  // k-steps of the QK^T product
  constexpr int DT = D / 16;                  // 16-wide output column tiles
  static_assert(GROUPS == 1 || GROUPS == 2, "one or two groups of four waves");
  static_assert(GROUPS == 1 || (DT * 4 + 2) * 256 * 4 <= GROUPS * 2 * TILE_BYTES, "merge buffer fits the tile buffers");
  __shared__ __attribute__((aligned(16))) unsigned char lds[GROUPS * 2 * TILE_BYTES];
  const int grp = GROUPS == 1 ? 0 : static_cast<int>(threadIdx.x >> 8);
  unsigned char* k_lds = lds + grp * 2 * TILE_BYTES;
  unsigned char* v_lds = k_lds + TILE_BYTES;

  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;   // within the group
  const int fr = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int hk = h / (p.Hq / p.Hkv);
  // a device-side key count can never address past the buffer capacity p.Sk
  int Sk = p.kv_len ? p.kv_len[b] : p.Sk;
  Sk = Sk > p.Sk ? p.Sk : (Sk < 0 ? 0 : Sk);
  const int q_off = p.q_offset_from_len ? (Sk - p.Sq) : p.q_offset;
  const int qb0 = SPLIT ? 0 : blockIdx.x * 64;  // first query row of the block
  const int qi = qb0 + wid * 16 + fr;        // this lane's query row
  const int qi_c = qi < p.Sq ? qi : p.Sq - 1;

  // Q^T fragments straight from global: Q[qi][ks*32 + 8g .. +8]
  bf16x8_t qf[KS];
  {
    const bf16_t* qp = p.q + b * p.q_sb + static_cast<long long>(qi_c) * p.q_ss + h * p.q_sh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 32 + g * 8);
  }

  const bf16_t* kbase = p.k + b * p.k_sb + hk * p.k_sh;
  const bf16_t* vbase = p.v + b * p.v_sb + hk * p.v_sh;

  // keys this block needs: all of them, or up to its last query's diagonal
  int k_end = Sk;
  if (CAUSAL) {
    const int last_q = min(qb0 + 63, p.Sq - 1);
    k_end = min(Sk, last_q + q_off + 1);
  }
  int ntiles = (k_end + TK - 1) / TK;
  int t_begin = 0;
  if (SPLIT) {
    const int per = (ntiles + p.nsplit - 1) / p.nsplit;
    t_begin = min(static_cast<int>(blockIdx.x) * per, ntiles);
    ntiles = min(t_begin + per, ntiles);
  }
  // waves whose 16 query rows are all padding only help staging the tiles
  const bool wave_active = qb0 + wid * 16 < p.Sq;

  // Two register sets: the loads of tile t+2 are issued while tile t is computed, so a tile has
  // two compute phases to arrive before it is written to LDS (with one set the write at the top
  // of every iteration waited out most of an L2 round trip: 2.5 us per 64-key tile).
  u32x4_t kreg[2][LD_PER_THREAD], vreg[2][LD_PER_THREAD];
  auto prefetch = [&](u32x4_t (&kr)[LD_PER_THREAD], u32x4_t (&vr)[LD_PER_THREAD], int t) {
#pragma unroll
    for (int i = 0; i < LD_PER_THREAD; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / CPR, ch = idx % CPR;
      int key = t * TK + row; key = key < Sk ? key : Sk - 1;
      kr[i] = load16(kbase + static_cast<long long>(key) * p.k_ss + ch * 8);
      vr[i] = load16(vbase + static_cast<long long>(key) * p.v_ss + ch * 8);
    }
  };
  auto commit = [&](const u32x4_t (&kr)[LD_PER_THREAD], const u32x4_t (&vr)[LD_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < LD_PER_THREAD; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / CPR, ch = idx % CPR;
      *reinterpret_cast<u32x4_t*>(k_lds + tile_off<D>(row, ch)) = kr[i];
      *reinterpret_cast<u32x4_t*>(v_lds + tile_off<D>(row, ch)) = vr[i];
    }
  };

  f32x4_t o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;

  auto compute = [&](int t) {
    // ---- S^T tile: 4 sub-tiles of 16 keys
    f32x4_t st[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) st[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)      // k-step outermost: four independent accumulators in turn
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(
            k_lds + tile_off<D>(n * 16 + fr, ks * 4 + g));
        st[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], st[n], 0, 0, 0);
      }

    // ---- online softmax on the RAW scores (lane = one query; 16 keys in registers).  The scale
    // is positive, so the maximum commutes with it and exp2(s * scale - m * scale) is one fma and
    // one v_exp_f32 per score; keys are masked only in tiles that cross the diagonal or the end
    // of the keys; the accumulators (AGPRs) are rescaled only when some row's maximum moved.
    const int key0 = t * TK + g * 4;
    const int last_key = t * TK + TK - 1;
    if (last_key > Sk - 1 || (CAUSAL && last_key > qb0 + wid * 16 + q_off)) {
      const int lim = CAUSAL ? min(Sk - 1, qi + q_off) : Sk - 1;  // last visible key
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (key0 + n * 16 + r > lim) st[n][r] = -INFINITY;
    }
    float mt = fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3]));
#pragma unroll
    for (int n = 1; n < 4; ++n) mt = fmaxf(mt, fmaxf(fmaxf(st[n][0], st[n][1]), fmaxf(st[n][2], st[n][3])));
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    const float msafe = mn == -INFINITY ? 0.f : mn;
    if (__builtin_amdgcn_ballot_w64(mn != m) != 0) {
      const float alpha = __builtin_amdgcn_exp2f((m - msafe) * p.scale_log2e);  // m = -inf -> 0
      l *= alpha;
#pragma unroll
      for (int i = 0; i < DT; ++i) o[i] *= alpha;
      m = mn;
    }
    const float ms = msafe * p.scale_log2e;
    float psum = 0.f;
    unsigned int pk[4][2];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(fmaf(st[n][r], p.scale_log2e, -ms)); psum += e[r]; }
      pk[n][0] = pack_bf16x2(e[0], e[1]);
      pk[n][1] = pack_bf16x2(e[2], e[3]);
    }
    l += psum;

    // ---- O^T += V^T · P^T, two k-steps of 32 keys; slot (g, j) <-> key
    //      32s + 4g + j (j < 4) or 32s + 16 + 4g + (j - 4)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4_t pw = {pk[2 * s][0], pk[2 * s][1], pk[2 * s + 1][0], pk[2 * s + 1][1]};
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
      const int q4 = fr >> 2, p4 = fr & 3;
      const int r0 = s * 32 + g * 4 + q4, r1 = r0 + 16;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int ch = dt * 2 + (p4 >> 1);
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4_t __attribute__((address_space(3)))*)(v_lds + tile_off<D>(r0, ch) + 8 * (p4 & 1)));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4_t __attribute__((address_space(3)))*)(v_lds + tile_off<D>(r1, ch) + 8 * (p4 & 1)));
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        const s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pf,
                                                        o[dt], 0, 0, 0);
      }
    }
  };

  // this group's tiles: first, first + GROUPS, ...; the trip count is group 0's (block-uniform:
  // the loop contains barriers), a group past its last tile only keeps step
  const int first = t_begin + grp;
  const int n_mine = first < ntiles ? (ntiles - first + GROUPS - 1) / GROUPS : 0;
  const int n_iter = t_begin < ntiles ? (ntiles - t_begin + GROUPS - 1) / GROUPS : 0;
  if (0 < n_mine) prefetch(kreg[0], vreg[0], first);
  if (1 < n_mine) prefetch(kreg[1], vreg[1], first + GROUPS);
  for (int k = 0; k < n_iter; k += 2) {
    __syncthreads();
    if (k < n_mine) commit(kreg[0], vreg[0]);
    __syncthreads();
    if (k + 2 < n_mine) prefetch(kreg[0], vreg[0], first + GROUPS * (k + 2));
    if (wave_active && k < n_mine) compute(first + GROUPS * k);
    if (k + 1 >= n_iter) break;
    __syncthreads();
    if (k + 1 < n_mine) commit(kreg[1], vreg[1]);
    __syncthreads();
    if (k + 3 < n_mine) prefetch(kreg[1], vreg[1], first + GROUPS * (k + 3));
    if (wave_active && k + 1 < n_mine) compute(first + GROUPS * (k + 1));
  }

  // ---- finalise: l is a per-lane partial over this lane's key slots
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if constexpr (GROUPS == 2) {
    // group 1 hands its state over in lane-private columns of the (now idle) tile buffers; lane
    // (wave, lane) of group 0 holds the same query row and output columns as its counterpart
    __syncthreads();
    float* xch = reinterpret_cast<float*>(lds);
    const int col = tid;   // wave * 64 + lane within the group
    if (grp == 1) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[(dt * 4 + r) * 256 + col] = o[dt][r];
      xch[(DT * 4) * 256 + col] = m;
      xch[(DT * 4 + 1) * 256 + col] = l;
    }
    __syncthreads();
    if (grp == 1) return;
    const float m1 = xch[(DT * 4) * 256 + col], l1 = xch[(DT * 4 + 1) * 256 + col];
    const float mn = fmaxf(m, m1);
    const float msafe = mn == -INFINITY ? 0.f : mn;
    const float a0 = __builtin_amdgcn_exp2f((m - msafe) * p.scale_log2e);    // -inf -> 0
    const float a1 = __builtin_amdgcn_exp2f((m1 - msafe) * p.scale_log2e);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] = o[dt][r] * a0 + xch[(dt * 4 + r) * 256 + col] * a1;
    l = l * a0 + l1 * a1;
    m = mn;
  }
  if (SPLIT) {
    if (qi < p.Sq) {
      const long long slot = ((static_cast<long long>(b) * p.Hq + h) * p.nsplit + blockIdx.x) * p.Sq + qi;
      float* po = p.part_o + slot * D;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        *reinterpret_cast<f32x4_t*>(po + dt * 16 + g * 4) = o[dt];
      if (g == 0) { p.part_ml[slot * 2] = m * p.scale_log2e; p.part_ml[slot * 2 + 1] = l; }   // m is kept unscaled
    }
    return;
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  if (p.lse && g == 0 && qi < p.Sq)
    p.lse[(static_cast<long long>(b) * p.Hq + h) * p.Sq + qi] = l > 0.f ? m * p.scale_log2e + log2f(l) : INFINITY;
  if (qi < p.Sq) {
    bf16_t* op = p.o + b * p.o_sb + static_cast<long long>(qi) * p.o_ss + h * p.o_sh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      u32x2_t w = {pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv),
                   pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv)};
      *reinterpret_cast<u32x2_t*>(op + dt * 16 + g * 4) = w;
    }
  }
}

// Same computation for long query ranges: one block = 128 query rows, each wave owns TWO 16-row
// query tiles, so every K fragment (ds_read_b128) and every V^T fragment (two transposing reads)
// feeds two MFMAs instead of one -- with 16 rows per wave the kernel above reads 32 KB of LDS
// per wave and key tile for 32 MFMAs and is bound by the LDS pipe.  K / V tiles are staged
// straight into LDS (buffer_load ... lds, swizzle applied to the per-lane source offset; keys past
// the end come back as zeros from the descriptor's range check) into a two-tile ring: the loads
// of tile t+1 are in flight while tile t is computed, and no staging registers are held.
__device__ __forceinline__ void attn_buffer_to_lds16(const unsigned char* src, int bytes, unsigned char* lds, int voff) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)lds, 16, voff, 0, 0, 0);
}

constexpr int AW_STAGES = 4;   // K / V tiles in the LDS ring: three in flight while one is computed

// AW_WAVES = 8: 256 query rows per block (long prompts); 4: 128 rows (a few hundred tokens, the ViT)
// QT: 16-row query tiles per wave (2: every K / V fragment feeds two MFMAs; 1: twice the blocks)
template <int D, bool CAUSAL, int AW_WAVES, int QT = 2>
__global__ __launch_bounds__(AW_WAVES * 64) void attn_prefill_wide_kernel(AttnParams p) {
  constexpr int CPR = D / 8;
  constexpr int TILE_BYTES = TK * D * 2;
  constexpr int NIW = TILE_BYTES / 1024 / AW_WAVES;   // 1 KB staging instructions per wave and tile
  constexpr int KS = D / 32;
  constexpr int DT = D / 16;
  constexpr int BR = AW_WAVES * 16 * QT;              // query rows per block
  static_assert(NIW >= 1, "a tile is at least one staging instruction per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [stage][K | V]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, g = lane >> 4;
  // grid = (heads, query blocks, batch): consecutive blocks are the same query block of different
  // heads, so under causal masking the whole grid starts in order of decreasing work
  const int b = blockIdx.z, h = blockIdx.x;
  const int hk = h / (p.Hq / p.Hkv);
  int Sk = p.kv_len ? p.kv_len[b] : p.Sk;
  Sk = Sk > p.Sk ? p.Sk : (Sk < 0 ? 0 : Sk);
  const int q_off = p.q_offset_from_len ? (Sk - p.Sq) : p.q_offset;
  // causal: the blocks with the most keys start first
  const int qb0 = (CAUSAL ? gridDim.y - 1 - blockIdx.y : blockIdx.y) * BR;
  const int qw0 = qb0 + wid * 16 * QT;       // first query row of this wave

  bf16x8_t qf[QT][KS];
  int qi[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qi[qt] = qw0 + qt * 16 + fr;
    const int qc = qi[qt] < p.Sq ? qi[qt] : p.Sq - 1;
    const bf16_t* qp = p.q + b * p.q_sb + static_cast<long long>(qc) * p.q_ss + h * p.q_sh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qt][ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 32 + g * 8);
  }

  const unsigned char* kbase = reinterpret_cast<const unsigned char*>(p.k + b * p.k_sb + hk * p.k_sh);
  const unsigned char* vbase = reinterpret_cast<const unsigned char*>(p.v + b * p.v_sb + hk * p.v_sh);
  const long long k_bytes = Sk > 0 ? 2ll * ((Sk - 1) * p.k_ss + D) : 0;
  const long long v_bytes = Sk > 0 ? 2ll * ((Sk - 1) * p.v_ss + D) : 0;
  int voff_k[NIW], voff_v[NIW];
#pragma unroll
  for (int j = 0; j < NIW; ++j) {
    const int idx = (wid * NIW + j) * 64 + lane;
    const int row = idx / CPR, ch = (idx % CPR) ^ swz<D>(row);
    voff_k[j] = static_cast<int>((row * p.k_ss + ch * 8) * 2);
    voff_v[j] = static_cast<int>((row * p.v_ss + ch * 8) * 2);
  }
  auto stage = [&](int buf, int t) {
    unsigned char* kd = lds + buf * 2 * TILE_BYTES + wid * NIW * 1024;
    const long long ka = 2ll * t * TK * p.k_ss, va = 2ll * t * TK * p.v_ss;
    const int kl = static_cast<int>(k_bytes > ka ? k_bytes - ka : 0), vl = static_cast<int>(v_bytes > va ? v_bytes - va : 0);
#pragma unroll
    for (int j = 0; j < NIW; ++j) attn_buffer_to_lds16(kbase + ka, kl, kd + j * 1024, voff_k[j]);
#pragma unroll
    for (int j = 0; j < NIW; ++j) attn_buffer_to_lds16(vbase + va, vl, kd + TILE_BYTES + j * 1024, voff_v[j]);
  };

  int k_end = Sk;
  if (CAUSAL) k_end = min(Sk, min(qb0 + BR - 1, p.Sq - 1) + q_off + 1);
  const int ntiles = (k_end + TK - 1) / TK;
  const bool wave_active = qw0 < p.Sq;
  const int q_hi = min(qw0 + 16 * QT - 1, p.Sq - 1);   // last query row of this wave

  f32x4_t o[QT][DT];
  float m[QT], l[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m[qt] = -INFINITY; l[qt] = 0.f;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[qt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  // The loop is software-pipelined into two phases per key tile,
  //   S: softmax of tile t (VALU)            M: P·V of tile t, then K·Q^T of tile t+1 (MFMA)
  // each closed by a barrier, and the two halves of the block (waves 0-3 / 4-7, one of each per
  // SIMD) run ONE BARRIER APART: while one half is in its MFMA phase the other is in its softmax,
  // so the SIMD's matrix and vector pipes work at the same time instead of taking turns.
  f32x4_t st[QT][4];
  unsigned int pk[QT][4][2];
  auto qk = [&](const unsigned char* k_lds) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) st[qt][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // k-step outermost: eight independent accumulators between two MFMAs on the same one
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(k_lds + tile_off<D>(n * 16 + fr, ks * 4 + g));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          st[qt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], st[qt][n], 0, 0, 0);
      }
  };
  auto softmax = [&](int t) {
    // Online softmax on the RAW scores (the scale is positive, so the maximum commutes with it):
    // exp2(s * scale - m * scale) is one fma and one v_exp_f32 per score.  The softmax, not the
    // MFMA, bounds this kernel (16 scores per lane and query tile, 4 cycles per VALU instruction),
    // so everything avoidable is avoided: keys are masked only in tiles that cross the diagonal
    // or the end of the keys, and the accumulators are rescaled only when some row's maximum
    // moved (they live in AGPRs; a rescale moves all of them through VGPRs).
    const int key0 = t * TK + g * 4;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q_lo = qw0 + qt * 16;                     // first query row of the tile (wave-uniform)
      const int last_key = t * TK + TK - 1;
      const bool need_mask = last_key > Sk - 1 || (CAUSAL && last_key > q_lo + q_off);
      if (need_mask) {
        const int lim = CAUSAL ? min(Sk - 1, qi[qt] + q_off) : Sk - 1;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + n * 16 + r > lim) st[qt][n][r] = -INFINITY;
      }
      float mt = fmaxf(fmaxf(st[qt][0][0], st[qt][0][1]), fmaxf(st[qt][0][2], st[qt][0][3]));
#pragma unroll
      for (int n = 1; n < 4; ++n)
        mt = fmaxf(mt, fmaxf(fmaxf(st[qt][n][0], st[qt][n][1]), fmaxf(st[qt][n][2], st[qt][n][3])));
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float mn = fmaxf(m[qt], mt);
      const float msafe = mn == -INFINITY ? 0.f : mn;
      if (__builtin_amdgcn_ballot_w64(mn != m[qt]) != 0) {   // some row's maximum moved
        const float alpha = __builtin_amdgcn_exp2f((m[qt] - msafe) * p.scale_log2e);   // m = -inf -> 0
        l[qt] *= alpha;
#pragma unroll
        for (int i = 0; i < DT; ++i) o[qt][i] *= alpha;
        m[qt] = mn;
      }
      const float ms = msafe * p.scale_log2e;
      float psum = 0.f;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(fmaf(st[qt][n][r], p.scale_log2e, -ms)); psum += e[r]; }
        pk[qt][n][0] = pack_bf16x2(e[0], e[1]);
        pk[qt][n][1] = pack_bf16x2(e[2], e[3]);
      }
      l[qt] += psum;
    }
  };
  auto pv = [&](const unsigned char* v_lds) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8_t pf[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        u32x4_t pw = {pk[qt][2 * s2][0], pk[qt][2 * s2][1], pk[qt][2 * s2 + 1][0], pk[qt][2 * s2 + 1][1]};
        pf[qt] = __builtin_bit_cast(bf16x8_t, pw);
      }
      const int q4 = fr >> 2, p4 = fr & 3;
      const int r0 = s2 * 32 + g * 4 + q4, r1 = r0 + 16;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int ch = dt * 2 + (p4 >> 1);
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4_t __attribute__((address_space(3)))*)(v_lds + tile_off<D>(r0, ch) + 8 * (p4 & 1)));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4_t __attribute__((address_space(3)))*)(v_lds + tile_off<D>(r1, ch) + 8 * (p4 & 1)));
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        const s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pf[qt], o[qt][dt], 0, 0, 0);
      }
    }
  };
  auto fence = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mine = [&](int t) { return wave_active && (!CAUSAL || t * TK <= q_hi + q_off); };
  auto wait_tile = [&]() {   // all but the youngest tile's loads of this wave have landed
    static_assert(2 * NIW == 2 || 2 * NIW == 4 || 2 * NIW == 8, "vmcnt literal");
    if constexpr (2 * NIW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (2 * NIW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  };

  // Ring of AW_STAGES tiles: while tile t is multiplied, t+1 is being scored, t+2 is landing and
  // t+3 has just been requested.  Tiles past the last one are requested too: their out-of-range
  // loads zero-fill a stage nobody reads and keep the count of loads in flight uniform.
  //   * a stage is re-requested at the start of phase M of tile t for tile t+3: its previous
  //     content, tile t-1, was last read in phase M of tile t-1, which the late half finishes
  //     at the barrier the early half passes on entering phase M of tile t;
  //   * tile t+2 is waited for at the end of phase M of tile t, by every wave before the
  //     barrier that closes it, and first read in phase M of tile t+1.
  const bool late = wid >= AW_WAVES / 2;
#pragma unroll
  for (int i = 0; i < AW_STAGES - 1; ++i) stage(i, i);
  wait_tile();                       // tiles 0 and 1
  fence();
  if (ntiles > 0 && mine(0)) qk(lds);
  if (late) fence();
  for (int t = 0; t < ntiles; ++t) {
    if (mine(t)) softmax(t);
    fence();
    stage((t + AW_STAGES - 1) % AW_STAGES, t + AW_STAGES - 1);
    if (mine(t)) pv(lds + (t % AW_STAGES) * 2 * TILE_BYTES + TILE_BYTES);
    if (t + 1 < ntiles && mine(t + 1)) qk(lds + ((t + 1) % AW_STAGES) * 2 * TILE_BYTES);
    wait_tile();                     // tile t+2
    fence();
  }
  if (!late) fence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // zero fills land before the LDS is handed on

#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float lt = l[qt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    if (qi[qt] >= p.Sq) continue;
    if (p.lse && g == 0)
      p.lse[(static_cast<long long>(b) * p.Hq + h) * p.Sq + qi[qt]] = lt > 0.f ? m[qt] * p.scale_log2e + log2f(lt) : INFINITY;
    bf16_t* op = p.o + b * p.o_sb + static_cast<long long>(qi[qt]) * p.o_ss + h * p.o_sh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      u32x2_t w = {pack_bf16x2(o[qt][dt][0] * inv, o[qt][dt][1] * inv),
                   pack_bf16x2(o[qt][dt][2] * inv, o[qt][dt][3] * inv)};
      *reinterpret_cast<u32x2_t*>(op + dt * 16 + g * 4) = w;
    }
  }
}

// Merge the key splits of one query row: thread d owns output column d.
template <int D>
__global__ __launch_bounds__(D) void attn_split_combine_kernel(AttnParams p) {
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
  const long long base = (static_cast<long long>(b) * p.Hq + h) * p.nsplit;
  // Every load is issued before the first use and none sits behind a branch: the
  // kernel is a handful of L2 round trips and they must overlap.  Splits past
  // nsplit re-read split 0 and are masked afterwards.
  constexpr int MAXS = 16;
  float ms[MAXS], ls[MAXS], os[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const long long slot = (base + (s < p.nsplit ? s : 0)) * p.Sq + qi;
    ms[s] = p.part_ml[slot * 2];
    ls[s] = p.part_ml[slot * 2 + 1];
    os[s] = p.part_o[slot * D + d];
  }
  float mx = -INFINITY;
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    ms[s] = s < p.nsplit ? ms[s] : -INFINITY;
    mx = fmaxf(mx, ms[s]);
  }
  float acc = 0.f, l = 0.f;
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const float w = (ms[s] == -INFINITY) ? 0.f : exp2f(ms[s] - mx);
    l += w * ls[s];
    acc += w * os[s];
  }
  bf16_t* op = p.o + b * p.o_sb + static_cast<long long>(qi) * p.o_ss + h * p.o_sh;
  op[d] = f32_to_bf16(l > 0.f ? acc / l : 0.f);
}

static int g_attn_groups = 0;  // key-tile groups per 64-row block: 0 = by shape, 1 / 2 = forced (tests, tuning)
void attn_force_groups(int g) { g_attn_groups = g; }
static int g_attn_wide = 1;   // 0 = never, 1 = by shape, 2 = whenever the strides allow (tests, tuning)
static int g_attn_wide4_min_rows = 1400;  // fewest query rows that take the 128-row blocks (mode 1): measured crossover 1300-1500 (1024 rows 18.3 vs 18.0 ms per 7B prefill, 1500 rows 25.9 vs 26.2; 636 rows 12.6 vs 12.3, ViT frame 2.50 vs 2.24)
void attn_enable_wide(int mode) { g_attn_wide = mode; }
void attn_wide4_min_rows(int rows) { g_attn_wide4_min_rows = rows; }

void launch_attn_prefill(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o,
                         const long long* strides /*12: q,k,v,o x (b,s,h)*/, const int* kv_len,
                         int B, int Sq, int Sk, int Hq, int Hkv, int D, int causal, int q_offset,
                         int q_offset_from_len, float scale, hipStream_t st, float* part_o,
                         float* part_ml, int nsplit, float* lse) {
  AttnParams p;
  p.part_o = part_o; p.part_ml = part_ml; p.nsplit = nsplit; p.lse = lse;
  p.q = q; p.k = k; p.v = v; p.o = o;
  p.q_sb = strides[0]; p.q_ss = strides[1]; p.q_sh = strides[2];
  p.k_sb = strides[3]; p.k_ss = strides[4]; p.k_sh = strides[5];
  p.v_sb = strides[6]; p.v_ss = strides[7]; p.v_sh = strides[8];
  p.o_sb = strides[9]; p.o_ss = strides[10]; p.o_sh = strides[11];
  p.kv_len = kv_len; p.Sq = Sq; p.Sk = Sk; p.Hq = Hq; p.Hkv = Hkv;
  p.q_offset = q_offset; p.q_offset_from_len = q_offset_from_len;
  p.scale_log2e = scale * 1.4426950408889634f;
  if (nsplit > 0) {
    // few query rows (Sq <= 64), causal, long cache: split the keys over blocks
    dim3 sgrid(nsplit, Hq, B), cgrid(Sq, Hq, B);
    if (D == 128) {
      hipLaunchKernelGGL((attn_prefill_kernel<128, true, true>), sgrid, dim3(256), 0, st, p);
      hipLaunchKernelGGL((attn_split_combine_kernel<128>), cgrid, dim3(128), 0, st, p);
    } else {
      hipLaunchKernelGGL((attn_prefill_kernel<64, true, true>), sgrid, dim3(256), 0, st, p);
      hipLaunchKernelGGL((attn_split_combine_kernel<64>), cgrid, dim3(64), 0, st, p);
    }
    return;
  }
  dim3 block(256);
  // Wide blocks (two query tiles per wave, K / V ring in LDS): 256 rows once those fill the chip,
  // 128 rows for anything from two such blocks up; K / V must allow 16-byte row chunks
  const bool strides_ok = strides[4] % 8 == 0 && strides[7] % 8 == 0 && strides[4] * 2ll * Sk < (1ll << 31) &&
                          strides[7] * 2ll * Sk < (1ll << 31);
  const long long blocks256 = static_cast<long long>((Sq + 255) / 256) * Hq * B;
  int wide = 0;             // waves per block of the ring-staged kernel (0: not used)
  constexpr int qt = 2;     // query tiles per wave
  if (strides_ok && g_attn_wide >= 1) {
    if (Sq >= 512 && blocks256 >= 256) wide = 8;
    else if (Sq >= g_attn_wide4_min_rows || g_attn_wide == 2) wide = 4;
  }
  if (wide) {
    const int rows = wide * 16 * qt;
    dim3 wgrid(Hq, (Sq + rows - 1) / rows, B), wblock(wide * 64);
    auto go = [&](auto kern, int d) {
      const int smem = AW_STAGES * 2 * TK * d * 2;
      static bool attr_set = false;     // one flag per kernel: the lambda is instantiated per kernel type
      if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
      }
      hipLaunchKernelGGL(kern, wgrid, wblock, smem, st, p);
    };
    if (wide == 8) {
      if (D == 128) { if (causal) go(attn_prefill_wide_kernel<128, true, 8>, 128); else go(attn_prefill_wide_kernel<128, false, 8>, 128); }
      else { if (causal) go(attn_prefill_wide_kernel<64, true, 8>, 64); else go(attn_prefill_wide_kernel<64, false, 8>, 64); }
    } else {
      if (D == 128) { if (causal) go(attn_prefill_wide_kernel<128, true, 4>, 128); else go(attn_prefill_wide_kernel<128, false, 4>, 128); }
      else { if (causal) go(attn_prefill_wide_kernel<64, true, 4>, 64); else go(attn_prefill_wide_kernel<64, false, 4>, 64); }
    }
    return;
  }
  dim3 grid((Sq + 63) / 64, Hq, B);
  // Two key-tile groups per block for the ViT's shape class -- head_dim 64, non-causal, so few blocks
  // that half the chip idles (577 tokens x 16 heads: 160) -- while all blocks are resident at once:
  // 12.3 -> 9.9 us per layer.  Causal head_dim-128 blocks do not gain (636 tokens x 32 heads: 23.9 ->
  // 25.1 us, 836: 30.4 -> 33.7): they are bound by the LDS pipe, not by the chain over their key
  // tiles, and the diagonal splits unevenly (benchmarks/bench_attn_groups.py).
  const long long nblocks = static_cast<long long>(grid.x) * Hq * B;
  const bool two = g_attn_groups == 2 ||
                   (g_attn_groups == 0 && D == 64 && !causal && nblocks <= 512 && Sk >= 4 * TK);
  if (two) {
    dim3 block2(512);
    if (D == 128) {
      if (causal) hipLaunchKernelGGL((attn_prefill_kernel<128, true, false, 2>), grid, block2, 0, st, p);
      else hipLaunchKernelGGL((attn_prefill_kernel<128, false, false, 2>), grid, block2, 0, st, p);
    } else {
      if (causal) hipLaunchKernelGGL((attn_prefill_kernel<64, true, false, 2>), grid, block2, 0, st, p);
      else hipLaunchKernelGGL((attn_prefill_kernel<64, false, false, 2>), grid, block2, 0, st, p);
    }
    return;
  }
  if (D == 128) {
    if (causal) hipLaunchKernelGGL((attn_prefill_kernel<128, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((attn_prefill_kernel<128, false>), grid, block, 0, st, p);
  } else {
    if (causal) hipLaunchKernelGGL((attn_prefill_kernel<64, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((attn_prefill_kernel<64, false>), grid, block, 0, st, p);
  }
}

}  // namespace evgpt
