// Flash-attention backward for the training paths (causal head_dim 128 of the decoder,
// non-causal head_dim 64 of the vision tower): dQ, dK, dV from Q, K, V, dO and the forward's
// per-row log-sum-exp, without ever materialising the score matrix.
//
// P is recomputed tile by tile:  P = exp2(S * scale*log2e - LSE),  dS = P * (dP - delta) * scale
// with dP = dO V^T and delta = rowsum(dO * O).  Two kernels, each the forward kernel's
// structure (attn_prefill.hip) with the roles of the operands rearranged, so that every
// product is an MFMA whose transposed operand comes from ds_read_b64_tr_b16 on the same
// swizzled LDS image and whose "probability" operand never leaves the registers:
//
//   * dq kernel -- one block per 64 query rows of a (batch, head); K and V tiles stream
//     through LDS.  S^T = K Q^T and dP^T = V dO^T put the query on the lane, so LSE / delta
//     are per-lane scalars and dS^T (bf16) is directly the B operand of
//     dQ^T[d][q] += K^T[d][key] dS^T[key][q]  (K^T by transposed reads of the K tile).
//   * dkv kernel -- one block per 64 keys of a (batch, kv head); Q and dO tiles stream
//     through LDS.  S = Q K^T and dP = dO V^T put the KEY on the lane (K / V fragments of the
//     wave's 16 keys stay in registers), P and dS are the B operands of
//     dV^T[d][key] += dO^T[d][q] P[q][key]   and   dK^T[d][key] += Q^T[d][q] dS[q][key].
//     dK and dV are owned by the block: no atomics, bitwise reproducible.  Grouped KV heads
//     loop over the query heads of the group.
//
// Recomputing S and dP in both kernels costs 7 products instead of 5; in exchange dQ needs
// no cross-block float atomics (which run at a sixth of the store rate on this chip) and no
// ordered hand-off.  At the sequence lengths trained here (<= 2K tokens) attention is a few
// percent of a step.
#include "common.h"
#include "attn_common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

struct AttnBwdParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* d_o;
  const float* lse; const float* delta;
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  int Sq, Sk, Hq, Hkv, q_offset;
  float scale, scale_log2e;
};

// delta[b][h][q] = sum_d dO[b][q][h][d] * O[b][q][h][d]
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o,
                                                         const bf16_t* __restrict__ d_o,
                                                         float* __restrict__ delta, long long rows, int Sq,
                                                         int Hq) {
  constexpr int LPR = D / 8;                 // lanes per row
  const long long r = (blockIdx.x * 256ll + threadIdx.x) / LPR;   // row = (b, q, h)
  const int dl = threadIdx.x % LPR;
  float s = 0.f;
  if (r < rows) {
    const u32x4_t a = load16(o + r * D + dl * 8), g = load16(d_o + r * D + dl * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) s += bf16lo(a[i]) * bf16lo(g[i]) + bf16hi(a[i]) * bf16hi(g[i]);
  }
  s = group_sum<LPR>(s);
  if (r < rows && dl == 0) {
    const long long h = r % Hq, bq = r / Hq, qq = bq % Sq, b = bq / Sq;
    delta[(b * Hq + h) * Sq + qq] = s;
  }
}

// stage one [64][D] tile (rows row0.. of a [*, S, H, D] tensor's head) into the swizzled image
template <int D>
__device__ __forceinline__ void tile_fetch(u32x4_t (&reg)[TK * (D / 8) / 256], const bf16_t* base,
                                           long long row_stride, int row0, int n_rows, int tid) {
  constexpr int CPR = D / 8;
#pragma unroll
  for (int i = 0; i < TK * CPR / 256; ++i) {
    const int idx = tid + i * 256;
    const int row = idx / CPR, ch = idx % CPR;
    int r = row0 + row; r = r < n_rows ? r : n_rows - 1;
    reg[i] = load16(base + static_cast<long long>(r) * row_stride + ch * 8);
  }
}
template <int D>
__device__ __forceinline__ void tile_commit(const u32x4_t (&reg)[TK * (D / 8) / 256], unsigned char* lds, int tid) {
  constexpr int CPR = D / 8;
#pragma unroll
  for (int i = 0; i < TK * CPR / 256; ++i) {
    const int idx = tid + i * 256;
    *reinterpret_cast<u32x4_t*>(lds + tile_off<D>(idx / CPR, idx % CPR)) = reg[i];
  }
}

// acc[dt] += T^T[d][x] * Pb[x][lane column]: T is a [64][D] LDS tile read transposed, pk the
// packed bf16 weights of this lane's column for the 64 x-rows (the forward kernel's P layout)
template <int D>
__device__ __forceinline__ void mma_tr(f32x4_t (&acc)[D / 16], const unsigned char* lds,
                                       const unsigned int (&pk)[4][2], int fr, int g) {
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const u32x4_t pw = {pk[2 * s][0], pk[2 * s][1], pk[2 * s + 1][0], pk[2 * s + 1][1]};
    const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
    const int q4 = fr >> 2, p4 = fr & 3;
    const int r0 = s * 32 + g * 4 + q4, r1 = r0 + 16;
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) {
      const int ch = dt * 2 + (p4 >> 1);
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (s16x4_t __attribute__((address_space(3)))*)(lds + tile_off<D>(r0, ch) + 8 * (p4 & 1)));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (s16x4_t __attribute__((address_space(3)))*)(lds + tile_off<D>(r1, ch) + 8 * (p4 & 1)));
      const s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pf, acc[dt], 0, 0, 0);
    }
  }
}

// st[n] = rows n*16.. of the LDS tile (A operand, ds_read_b128) times the register fragments
template <int D>
__device__ __forceinline__ void mma_rows(f32x4_t (&st)[4], const unsigned char* lds,
                                         const bf16x8_t (&bfrag)[D / 32], int fr, int g) {
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    st[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) {
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(lds + tile_off<D>(n * 16 + fr, ks * 4 + g));
      st[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag[ks], st[n], 0, 0, 0);
    }
  }
}

// ------------------------------------------------------------------------ dQ
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdParams p) {
  constexpr int TILE_BYTES = TK * D * 2, KS = D / 32, DT = D / 16, LD = TK * (D / 8) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE_BYTES];
  unsigned char* k_lds = lds;
  unsigned char* v_lds = lds + TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, hk = h / (p.Hq / p.Hkv);
  const int qb0 = blockIdx.x * 64, qi = qb0 + wid * 16 + fr, qi_c = qi < p.Sq ? qi : p.Sq - 1;
  const long long q_row = static_cast<long long>(p.Hq) * D, k_row = static_cast<long long>(p.Hkv) * D;
  const bf16_t* qp = p.q + (static_cast<long long>(b) * p.Sq + qi_c) * q_row + h * D;
  const bf16_t* dop = p.d_o + (static_cast<long long>(b) * p.Sq + qi_c) * q_row + h * D;
  bf16x8_t qf[KS], dof[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 32 + g * 8);
    dof[ks] = *reinterpret_cast<const bf16x8_t*>(dop + ks * 32 + g * 8);
  }
  const long long stat = (static_cast<long long>(b) * p.Hq + h) * p.Sq + qi_c;
  const float lse = p.lse[stat], dlt = p.delta[stat];
  const bf16_t* kbase = p.k + static_cast<long long>(b) * p.Sk * k_row + hk * D;
  const bf16_t* vbase = p.v + static_cast<long long>(b) * p.Sk * k_row + hk * D;
  int k_end = p.Sk;
  if (CAUSAL) k_end = min(p.Sk, min(qb0 + 63, p.Sq - 1) + p.q_offset + 1);
  const int ntiles = (k_end + TK - 1) / TK;
  const bool wave_active = qb0 + wid * 16 < p.Sq;

  f32x4_t acc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t kreg[LD], vreg[LD];
  if (ntiles > 0) { tile_fetch<D>(kreg, kbase, k_row, 0, p.Sk, tid); tile_fetch<D>(vreg, vbase, k_row, 0, p.Sk, tid); }
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();                        // every wave is done with the previous tile
    tile_commit<D>(kreg, k_lds, tid);
    tile_commit<D>(vreg, v_lds, tid);
    __syncthreads();
    if (t + 1 < ntiles) {
      tile_fetch<D>(kreg, kbase, k_row, (t + 1) * TK, p.Sk, tid);
      tile_fetch<D>(vreg, vbase, k_row, (t + 1) * TK, p.Sk, tid);
    }
    if (!wave_active) continue;
    f32x4_t st[4], dp[4];
    mma_rows<D>(st, k_lds, qf, fr, g);      // S^T[key][q]
    mma_rows<D>(dp, v_lds, dof, fr, g);     // dP^T[key][q]
    const int key0 = t * TK + g * 4;
    const int lim = CAUSAL ? min(p.Sk - 1, qi + p.q_offset) : p.Sk - 1;
    unsigned int pk[4][2];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + n * 16 + r;
        const float pr = key <= lim ? exp2f(st[n][r] * p.scale_log2e - lse) : 0.f;
        ds[r] = pr * (dp[n][r] - dlt) * p.scale;
      }
      pk[n][0] = pack_bf16x2(ds[0], ds[1]);
      pk[n][1] = pack_bf16x2(ds[2], ds[3]);
    }
    mma_tr<D>(acc, k_lds, pk, fr, g);       // dQ^T[d][q] += K^T dS^T
  }
  if (qi < p.Sq) {
    bf16_t* dqp = p.dq + (static_cast<long long>(b) * p.Sq + qi) * q_row + h * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const u32x2_t w = {pack_bf16x2(acc[dt][0], acc[dt][1]), pack_bf16x2(acc[dt][2], acc[dt][3])};
      *reinterpret_cast<u32x2_t*>(dqp + dt * 16 + g * 4) = w;
    }
  }
}

// ------------------------------------------------------------------- dK, dV
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdParams p) {
  constexpr int TILE_BYTES = TK * D * 2, KS = D / 32, DT = D / 16, LD = TK * (D / 8) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE_BYTES];
  __shared__ __attribute__((aligned(16))) float s_lse[TK], s_dlt[TK];
  unsigned char* q_lds = lds;
  unsigned char* do_lds = lds + TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, hk = blockIdx.y, rep = p.Hq / p.Hkv;
  const int kb0 = blockIdx.x * 64, ki = kb0 + wid * 16 + fr, ki_c = ki < p.Sk ? ki : p.Sk - 1;
  const long long q_row = static_cast<long long>(p.Hq) * D, k_row = static_cast<long long>(p.Hkv) * D;
  const bf16_t* kp = p.k + (static_cast<long long>(b) * p.Sk + ki_c) * k_row + hk * D;
  const bf16_t* vp = p.v + (static_cast<long long>(b) * p.Sk + ki_c) * k_row + hk * D;
  bf16x8_t kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + ks * 32 + g * 8);
    vf[ks] = *reinterpret_cast<const bf16x8_t*>(vp + ks * 32 + g * 8);
  }
  // first query tile that can see this key block: q + q_offset >= kb0
  int t_begin = 0;
  if (CAUSAL) t_begin = max(0, kb0 - p.q_offset) / TK;
  const int ntiles = (p.Sq + TK - 1) / TK;
  const bool wave_active = kb0 + wid * 16 < p.Sk;

  f32x4_t dk[DT], dv[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) { dk[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  u32x4_t qreg[LD], doreg[LD];
  for (int hh = 0; hh < rep; ++hh) {
    const int h = hk * rep + hh;
    const bf16_t* qbase = p.q + static_cast<long long>(b) * p.Sq * q_row + h * D;
    const bf16_t* dobase = p.d_o + static_cast<long long>(b) * p.Sq * q_row + h * D;
    const float* lse_b = p.lse + (static_cast<long long>(b) * p.Hq + h) * p.Sq;
    const float* dlt_b = p.delta + (static_cast<long long>(b) * p.Hq + h) * p.Sq;
    if (t_begin < ntiles) {
      tile_fetch<D>(qreg, qbase, q_row, t_begin * TK, p.Sq, tid);
      tile_fetch<D>(doreg, dobase, q_row, t_begin * TK, p.Sq, tid);
    }
    for (int t = t_begin; t < ntiles; ++t) {
      __syncthreads();
      tile_commit<D>(qreg, q_lds, tid);
      tile_commit<D>(doreg, do_lds, tid);
      if (tid < TK) {
        const int qq = min(t * TK + tid, p.Sq - 1);
        s_lse[tid] = lse_b[qq];
        s_dlt[tid] = dlt_b[qq];
      }
      __syncthreads();
      if (t + 1 < ntiles) {
        tile_fetch<D>(qreg, qbase, q_row, (t + 1) * TK, p.Sq, tid);
        tile_fetch<D>(doreg, dobase, q_row, (t + 1) * TK, p.Sq, tid);
      }
      if (!wave_active) continue;
      f32x4_t st[4], dp[4];
      mma_rows<D>(st, q_lds, kf, fr, g);     // S[q][key]: key on the lane
      mma_rows<D>(dp, do_lds, vf, fr, g);    // dP[q][key]
      unsigned int pp[4][2], pd[4][2];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(&s_lse[n * 16 + g * 4]);
        const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(&s_dlt[n * 16 + g * 4]);
        float pr[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = t * TK + n * 16 + g * 4 + r;
          const bool vis = qq < p.Sq && ki < p.Sk && (!CAUSAL || ki <= qq + p.q_offset);
          pr[r] = vis ? exp2f(st[n][r] * p.scale_log2e - l4[r]) : 0.f;
          ds[r] = pr[r] * (dp[n][r] - d4[r]) * p.scale;
        }
        pp[n][0] = pack_bf16x2(pr[0], pr[1]); pp[n][1] = pack_bf16x2(pr[2], pr[3]);
        pd[n][0] = pack_bf16x2(ds[0], ds[1]); pd[n][1] = pack_bf16x2(ds[2], ds[3]);
      }
      mma_tr<D>(dv, do_lds, pp, fr, g);      // dV^T[d][key] += dO^T P
      mma_tr<D>(dk, q_lds, pd, fr, g);       // dK^T[d][key] += Q^T dS
    }
  }
  if (ki < p.Sk) {
    bf16_t* dkp = p.dk + (static_cast<long long>(b) * p.Sk + ki) * k_row + hk * D;
    bf16_t* dvp = p.dv + (static_cast<long long>(b) * p.Sk + ki) * k_row + hk * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const u32x2_t wk = {pack_bf16x2(dk[dt][0], dk[dt][1]), pack_bf16x2(dk[dt][2], dk[dt][3])};
      const u32x2_t wv = {pack_bf16x2(dv[dt][0], dv[dt][1]), pack_bf16x2(dv[dt][2], dv[dt][3])};
      *reinterpret_cast<u32x2_t*>(dkp + dt * 16 + g * 4) = wk;
      *reinterpret_cast<u32x2_t*>(dvp + dt * 16 + g * 4) = wv;
    }
  }
}

template <int D>
static void launch_bwd_d(const AttnBwdParams& p, const bf16_t* o, float* delta, int B, int causal, hipStream_t st) {
  const long long rows = static_cast<long long>(B) * p.Sq * p.Hq;
  const long long threads = rows * (D / 8);
  hipLaunchKernelGGL(attn_delta_kernel<D>, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, st,
                     o, p.d_o, delta, rows, p.Sq, p.Hq);
  const dim3 gq((p.Sq + 63) / 64, p.Hq, B), gk((p.Sk + 63) / 64, p.Hkv, B), blk(256);
  if (causal) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<D, true>), gq, blk, 0, st, p);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, true>), gk, blk, 0, st, p);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<D, false>), gq, blk, 0, st, p);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, false>), gk, blk, 0, st, p);
  }
}

void launch_attn_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* o, const bf16_t* d_o,
                     const float* lse, float* delta, bf16_t* dq, bf16_t* dk, bf16_t* dv, int B, int Sq,
                     int Sk, int Hq, int Hkv, int D, int causal, int q_offset, float scale, hipStream_t st) {
  AttnBwdParams p;
  p.q = q; p.k = k; p.v = v; p.d_o = d_o; p.lse = lse; p.delta = delta; p.dq = dq; p.dk = dk; p.dv = dv;
  p.Sq = Sq; p.Sk = Sk; p.Hq = Hq; p.Hkv = Hkv; p.q_offset = q_offset;
  p.scale = scale; p.scale_log2e = scale * 1.4426950408889634f;
  if (D == 128) launch_bwd_d<128>(p, o, delta, B, causal, st);
  else launch_bwd_d<64>(p, o, delta, B, causal, st);
}

}  // namespace evgpt
