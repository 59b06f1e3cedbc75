// Decode-time weight-streaming GEMV family for batch <= 4 token rows.
//
//   y[b][n] = epilogue( sum_k W[n][k] * prologue(x)[b][k] )
//
// Regime: one token per sequence reads every weight exactly once (13.5 GB per
// token for a 7B model), so these kernels are pure HBM streaming and what
// matters is bytes in flight per CU.  Design:
//   * weights go straight from HBM to VGPRs with non-temporal 16-byte loads
//     (no LDS round trip: a weight byte is used by exactly one wave);
//   * a *unit* of R=4 output rows is shared by KS waves that each stream 1/KS of
//     the K range and combine through LDS.  KS is chosen per shape so that every
//     launch puts >= 24 waves on each of the 256 CUs: a 4096-row matrix is only
//     1024 units, which at one wave per unit is 4 waves per CU and measured
//     2.9 TB/s; with KS=4 the same matrix keeps 16+ waves per CU streaming;
//   * activations: with a fused RMSNorm the block stages the normalised vector
//     in LDS once (norm and projection are one launch); without a norm each lane
//     reads its x chunk straight from L2 — no barrier before the first weight
//     load;
//   * v_dot2c_f32_bf16 does the MACs without bf16->f32 unpacking;
//   * epilogues fuse what would otherwise be separate launches: bias/act,
//     residual add, SwiGLU (gate/up row pairs), RoPE + KV-cache append (q/k row
//     pairs j, j+D/2), fp32 logits + per-block argmax partials.
// Positions / cache lengths are read from device memory so a captured graph can
// be replayed for every token.
//
// 4-bit weights (template Q = 1): the same kernels stream a 16-entry-codebook
// format (NF4 or any other table): [N][K/2] code bytes + one fp32 scale per 64
// weights.  A lane's 16-byte load is 32 weights that share one scale, so the
// scale is applied once per load; codes become bf16 pairs for v_dot2c through
// v_perm_b32 lookups in a register-resident table (no LDS traffic for the
// table).  x is always staged in LDS in this mode (a lane needs 64 bytes of x per
// 16 bytes of weights), XOR-swizzled so that the 64-byte lane stride is
// conflict-free.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "../evgpt_kernels.h"
#include "w4.h"

namespace evgpt {

enum { P_NONE = 0, P_RMSNORM = 1 };
enum { E_STORE = 0, E_RESIDUAL = 1, E_SWIGLU = 2, E_QKV_ROPE = 3, E_LOGITS = 4 };

struct GemvParams {
  const bf16_t* W;       // [N][K]
  const bf16_t* x;       // [B][K]
  const bf16_t* norm_w;  // [K]           (P_RMSNORM)
  float eps;
  int N, K;
  // E_STORE / E_RESIDUAL / E_SWIGLU
  bf16_t* y;             // [B][N_out]
  float* yf;             // E_STORE: raw fp32 sums [B][N] instead of y (tensor-parallel partials)
  const bf16_t* res;     // [B][N]        (E_RESIDUAL; may alias y)
  const bf16_t* bias;    // [N] or null
  int act;
  // E_QKV_ROPE
  bf16_t* q_out;         // [B][Hq*D]
  bf16_t* k_cache;       // [B][Hkv][Lmax][D]
  bf16_t* v_cache;
  const float* cos_tab;  // [max_pos][D/2]
  const float* sin_tab;
  const int* pos;        // [B] device-resident positions, or [1] with rows_one_seq
  int rows_one_seq;      // 1: the B rows are positions pos[0]+b of ONE sequence (cache slot 0)
  int n_q_heads, n_kv_heads, head_dim, l_max;
  // E_LOGITS
  float* logits;         // [B][N] or null
  float* part_val;       // [B][gridDim.x]
  int* part_idx;
  // Q = 1: W holds [N][K/2] code bytes (byte j of each 32-bit word = codes of weights
  // j and 4+j of that word's 8 consecutive weights, low nibble first)
  const float* absmax;   // [N][K/64]
  W4Code code;           // byte planes of the 16 bf16 codebook entries
};

// Measured on the 7B decode step (ms per token, same box): PRE/UNROLL/OCC = 1/4/4: 2.72,
// 2/4/4: 3.13 (spills), 4/6/1: 2.90, 4/8/1: 2.94, 4/6/3: 4.39 (spills) -- prefetching more of
// the first unit ahead of the norm, or issuing more than four iterations at once, costs
// more in registers (fewer resident blocks) than it returns.
#ifndef EVGPT_GEMV_PRE
#define EVGPT_GEMV_PRE 1       // weight chunk-iterations requested ahead of the fused norm
#endif
#ifndef EVGPT_GEMV_UNROLL
#define EVGPT_GEMV_UNROLL 4    // most chunk-iterations whose loads are all issued before the first use
#endif
#ifndef EVGPT_GEMV_WFIRST
#define EVGPT_GEMV_WFIRST 0    // 1: request the PRE weight chunks before the norm's x / norm-weight chunks
#endif
#ifndef EVGPT_GEMV_SIMPLE
#define EVGPT_GEMV_SIMPLE 0    // 1: plain loop unrolled by UNROLL instead of the issue-everything-first shots
#endif
#ifndef EVGPT_GEMV_OCC
#define EVGPT_GEMV_OCC 4       // waves per SIMD a one-row bf16 kernel's register budget must admit
#endif
constexpr int R = 4;           // rows per unit
constexpr int GEMV_WAVES = 4;  // waves per block

template <int E>
__device__ __forceinline__ void unit_rows(const GemvParams& p, int u, int (&row)[R]) {
  if constexpr (E == E_SWIGLU) {
    const int half = p.N >> 1;  // rows [0,half) = gate, [half,N) = up
    row[0] = 2 * u; row[1] = 2 * u + 1; row[2] = half + 2 * u; row[3] = half + 2 * u + 1;
  } else if constexpr (E == E_QKV_ROPE) {
    const int hd = p.head_dim, hh = hd >> 1;
    const int rot_rows = (p.n_q_heads + p.n_kv_heads) * hd;  // q and k regions rotate
    const int rot_units = rot_rows >> 2;
    if (u < rot_units) {
      const int units_per_head = hd >> 2;  // 2 pairs per unit
      const int h = u / units_per_head, j = (u % units_per_head) * 2;
      const int base = h * hd;
      row[0] = base + j; row[1] = base + j + 1; row[2] = base + hh + j; row[3] = base + hh + j + 1;
    } else {
      const int r0 = rot_rows + (u - rot_units) * 4;
      row[0] = r0; row[1] = r0 + 1; row[2] = r0 + 2; row[3] = r0 + 3;
    }
  } else {
    row[0] = 4 * u; row[1] = 4 * u + 1; row[2] = 4 * u + 2; row[3] = 4 * u + 3;
  }
}

// The same maps for RR rows per unit (RR / 2 rotate-half or gate / up pairs): RR = 4 gives unit_rows.
template <int E, int RR>
__device__ __forceinline__ void unit_rows_n(const GemvParams& p, int u, int (&row)[RR]) {
  constexpr int NP = RR / 2;
  if constexpr (E == E_SWIGLU) {
    static_assert(RR % 2 == 0, "gate / up pairs");
    const int half = p.N >> 1;
#pragma unroll
    for (int i = 0; i < NP; ++i) { row[i] = NP * u + i; row[NP + i] = half + NP * u + i; }
  } else if constexpr (E == E_QKV_ROPE) {
    static_assert(RR % 2 == 0, "rotate-half pairs");
    const int hd = p.head_dim, hh = hd >> 1;
    const int rot_rows = (p.n_q_heads + p.n_kv_heads) * hd;
    const int rot_units = rot_rows / RR;
    if (u < rot_units) {
      const int units_per_head = hh / NP;
      const int h = u / units_per_head, j = (u % units_per_head) * NP;
      const int base = h * hd;
#pragma unroll
      for (int i = 0; i < NP; ++i) { row[i] = base + j + i; row[NP + i] = base + hh + j + i; }
    } else {
      const int r0 = rot_rows + (u - rot_units) * RR;
#pragma unroll
      for (int r = 0; r < RR; ++r) row[r] = r0 + r;
    }
  } else {
#pragma unroll
    for (int r = 0; r < RR; ++r) row[r] = RR * u + r;
  }
}

// x chunk i (16 bytes) sits at this slot of the LDS image.  Q = 1 lanes read chunks
// 4c..4c+3 of consecutive c: the XOR spreads a 16-lane group over all 64 banks.
template <int Q>
__device__ __forceinline__ int xs_slot(int i) {
  if constexpr (Q) return i ^ ((i >> 4) & 3);
  else return i;
}

template <int B, int P, int E, int KS, int Q>
// 4-bit: the expansion is VALU work that needs other waves to hide under, so the
// register budget is capped for 4 waves per SIMD (the compiler otherwise hoists every
// row's expansion and ends at 1-2)
__global__ __launch_bounds__(GEMV_WAVES * 64, Q ? (B <= 2 ? 4 : 3) : (B == 1 ? EVGPT_GEMV_OCC : (B == 2 ? 3 : 2))) void gemv_kernel(GemvParams p) {
  static_assert(KS == 1 || KS == 2 || KS == 4, "KS waves per unit");
  static_assert(E != E_LOGITS || KS == 1, "logits epilogue runs one wave per unit");
  constexpr int UPB = GEMV_WAVES / KS;  // units per block iteration
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS carve: [partials: GEMV_WAVES*R*B floats][argmax scratch][x image (P_RMSNORM)]
  float* part = reinterpret_cast<float*>(smem);
  constexpr int PART_BYTES = ((GEMV_WAVES * R * B * 4 + 15) / 16) * 16;
  float* red = reinterpret_cast<float*>(smem + PART_BYTES);          // 16 floats
  u32x4_t* xs = reinterpret_cast<u32x4_t*>(smem + PART_BYTES + 64);  // [B][K/8]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ks = wid % KS, ub_slot = wid / KS;
  const int nchunk = p.K >> 3;              // 16-byte chunks of one x row
  const int nwc = Q ? (p.K >> 5) : nchunk;  // 16-byte chunks of one weight row
  const size_t row_bytes = Q ? static_cast<size_t>(p.K >> 1) : static_cast<size_t>(p.K) * 2;
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(p.W);
  const int nblk = p.K >> 6;                // scales per weight row (Q)
  constexpr bool STAGE = (P == P_RMSNORM) || Q;  // x goes through LDS

  // this wave's slice of the K range, in 16-byte weight chunks
  const int cper = (nwc + KS - 1) / KS;
  const int c_begin = ks * cper;
  const int c_end = min(c_begin + cper, nwc);
  const int n_units = (p.N + R - 1) / R;
  const int n_ub = (n_units + UPB - 1) / UPB;

  // ---------------- prologue: normalise x into LDS (fused RMSNorm) ----------------
  // A wave's loads return in issue order, so the x / norm-weight chunks of the fused norm
  // are requested FIRST and the weights right behind them: the weights do not depend on
  // x, and with up to PRE chunk-iterations of this wave's first unit already in flight
  // the HBM round trip of the weight bytes overlaps the reduction instead of following
  // it (a launch whose blocks are all resident at once has nothing else to hide it).
  constexpr int PRE = Q ? 0 : EVGPT_GEMV_PRE;  // 4-bit: the registers go to the expansion
  constexpr int XC = 2;  // x chunks a thread keeps in registers between the two norm passes
  const bool fast_norm = (P == P_RMSNORM) && nchunk <= XC * GEMV_WAVES * 64;
  constexpr int XB = Q ? 1 : B;   // 4-bit: one x row live at a time (registers go to the expansion)
  u32x4_t nw[XC], xrow[XB][XC];
  auto load_xrow = [&](int b, u32x4_t (&dst)[XC]) {
#pragma unroll
    for (int i = 0; i < XC; ++i) {
      const int c = tid + i * GEMV_WAVES * 64;
      dst[i] = load16(p.x + (static_cast<size_t>(b) * p.K + (c < nchunk ? c : 0) * 8));
      if (c >= nchunk) dst[i] = u32x4_t{0, 0, 0, 0};
    }
  };
  auto request_norm_operands = [&]() {
    if constexpr (P == P_RMSNORM) {
      if (fast_norm) {
#pragma unroll
        for (int i = 0; i < XC; ++i) {
          const int c = tid + i * GEMV_WAVES * 64;
          nw[i] = load16(p.norm_w + (c < nchunk ? c : 0) * 8);
        }
        if constexpr (!Q) {
#pragma unroll
          for (int b = 0; b < B; ++b) load_xrow(b, xrow[b]);
        }
      }
    }
  };
  if constexpr (!EVGPT_GEMV_WFIRST) request_norm_operands();
  u32x4_t pre[PRE > 0 ? PRE : 1][R];
  float pre_sc[PRE > 0 ? PRE : 1][R];
  if constexpr (STAGE && PRE > 0) {
    const int u = blockIdx.x * UPB + ub_slot;
    int row[R];
    unit_rows<E>(p, u < n_units ? u : n_units - 1, row);
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
      const int c = c_begin + lane + 64 * i;
      const int cc = c < c_end ? c : 0;  // out-of-slice lanes load a valid address and drop it
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int rr = row[r] < p.N ? row[r] : p.N - 1;
        pre[i][r] = load_stream16(Wb + static_cast<size_t>(rr) * row_bytes + static_cast<size_t>(cc) * 16);
        pre_sc[i][r] = 0.f;
      }
    }
  }
  if constexpr (EVGPT_GEMV_WFIRST) request_norm_operands();
  if constexpr (P == P_RMSNORM) {
    if (fast_norm) {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        if constexpr (Q) load_xrow(b, xrow[0]);
        const u32x4_t (&xr)[XC] = xrow[Q ? 0 : b];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < XC; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = bf16lo(xr[i][j]), hi = bf16hi(xr[i][j]);
            ss += lo * lo + hi * hi;
          }
        ss = block_sum(ss, red);
        const float rstd = rsqrtf(ss / static_cast<float>(p.K) + p.eps);
#pragma unroll
        for (int i = 0; i < XC; ++i) {
          const int c = tid + i * GEMV_WAVES * 64;
          u32x4_t o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // HF order: round x*rstd to bf16, then multiply by the bf16 weight
            const float lo = bf16_to_f32(f32_to_bf16(bf16lo(xr[i][j]) * rstd)) * bf16lo(nw[i][j]);
            const float hi = bf16_to_f32(f32_to_bf16(bf16hi(xr[i][j]) * rstd)) * bf16hi(nw[i][j]);
            o[j] = pack_bf16x2(lo, hi);
          }
          if (c < nchunk) xs[b * nchunk + xs_slot<Q>(c)] = o;
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        float ss = 0.f;
        for (int c = tid; c < nchunk; c += blockDim.x) {
          u32x4_t v = load16(p.x + (static_cast<size_t>(b) * p.K + c * 8));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float lo = bf16lo(v[i]), hi = bf16hi(v[i]);
            ss += lo * lo + hi * hi;
          }
        }
        ss = block_sum(ss, red);
        const float rstd = rsqrtf(ss / static_cast<float>(p.K) + p.eps);
        for (int c = tid; c < nchunk; c += blockDim.x) {
          u32x4_t v = load16(p.x + (static_cast<size_t>(b) * p.K + c * 8));
          u32x4_t w = load16(p.norm_w + c * 8);
          u32x4_t o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float lo = bf16_to_f32(f32_to_bf16(bf16lo(v[i]) * rstd)) * bf16lo(w[i]);
            const float hi = bf16_to_f32(f32_to_bf16(bf16hi(v[i]) * rstd)) * bf16hi(w[i]);
            o[i] = pack_bf16x2(lo, hi);
          }
          xs[b * nchunk + xs_slot<Q>(c)] = o;
        }
      }
    }
    __syncthreads();
  } else if constexpr (Q) {
#pragma unroll
    for (int b = 0; b < B; ++b)
      for (int c = tid; c < nchunk; c += blockDim.x)
        xs[b * nchunk + xs_slot<Q>(c)] = load16(p.x + (static_cast<size_t>(b) * p.K + c * 8));
    __syncthreads();
  }

  float best_val[B];
  int best_idx[B];
  if constexpr (E == E_LOGITS) {
#pragma unroll
    for (int b = 0; b < B; ++b) { best_val[b] = -INFINITY; best_idx[b] = 0; }
  }

  // Epilogue operands that do not depend on the dot products are requested early, so
  // the tail of a block is one memory round trip (the store), not a chain of them:
  // positions here, the residual rows at the top of each unit.
  int pos_in[B];
  if constexpr (E == E_QKV_ROPE) {
#pragma unroll
    for (int b = 0; b < B; ++b) pos_in[b] = p.rows_one_seq ? p.pos[0] + b : p.pos[b];
  }

  // block-uniform trip count: the partial-sum exchange below contains a barrier
  for (int ub = blockIdx.x; ub < n_ub; ub += gridDim.x) {
    const int u = ub * UPB + ub_slot;
    const bool unit_ok = u < n_units;
    int row[R];
    unit_rows<E>(p, unit_ok ? u : n_units - 1, row);
    const unsigned char* wp[R];
    const float* sp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int rr = row[r] < p.N ? row[r] : p.N - 1;  // clamp; out-of-range rows are discarded
      wp[r] = Wb + static_cast<size_t>(rr) * row_bytes;
      if constexpr (Q) sp[r] = p.absmax + static_cast<size_t>(rr) * nblk;
    }
    float acc[R][B];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
    bf16_t res_in[R][B];
    if constexpr (E == E_RESIDUAL) {
      if (lane == 0 && ks == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int b = 0; b < B; ++b)
            res_in[r][b] = row[r] < p.N ? p.res[static_cast<size_t>(b) * p.N + row[r]] : bf16_t(0);
      }
    }

    float cs_in[2][B], sn_in[2][B];
    if constexpr (E == E_QKV_ROPE) {
      if (lane == 0 && ks == 0 && row[0] < (p.n_q_heads + p.n_kv_heads) * p.head_dim) {
        const int hh = p.head_dim >> 1, j = row[0] % p.head_dim;
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const int pc = pos_in[b] < 0 ? 0 : (pos_in[b] >= p.l_max ? p.l_max - 1 : pos_in[b]);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            cs_in[i][b] = p.cos_tab[static_cast<size_t>(pc) * hh + j + i];
            sn_in[i][b] = p.sin_tab[static_cast<size_t>(pc) * hh + j + i];
          }
        }
      }
    }

    // xg: this lane's x chunk per row, already in registers (no fused norm, bf16 weights)
    auto consume = [&](const u32x4_t (&wv)[R], const float (&sc)[R], int c, const u32x4_t* xg) {
      if constexpr (Q) {
        u32x4_t xv[B][4];
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) xv[b][j] = xs[b * nchunk + xs_slot<Q>(4 * c + j)];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          // one row at a time: without the fence the scheduler interleaves the four
          // rows' expansions (64 more live registers) and the kernel spills
          __builtin_amdgcn_sched_barrier(0);
          float s[B];
#pragma unroll
          for (int b = 0; b < B; ++b) s[b] = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned int e[4];
            w4_expand8(wv[r][j], p.code, e);
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
              for (int i = 0; i < 4; ++i) s[b] = dot2_bf16(e[i], xv[b][j][i], s[b]);
          }
#pragma unroll
          for (int b = 0; b < B; ++b) acc[r][b] = fmaf(sc[r], s[b], acc[r][b]);
        }
      } else {
#pragma unroll
        for (int b = 0; b < B; ++b) {
          u32x4_t xv;
          if constexpr (P == P_RMSNORM) xv = xs[b * nchunk + c];
          else xv = xg[b];
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r][b] = dot8_bf16(wv[r], xv, acc[r][b]);
        }
      }
    };
    int c = c_begin + lane;
    // chunk-iterations of this wave's slice (wave-uniform: ks comes from the wave id)
    int it_left = __builtin_amdgcn_readfirstlane((c_end - c_begin + 63) >> 6);
    if constexpr (STAGE && PRE > 0) {
      if (ub == blockIdx.x) {  // the chunks requested ahead of the prologue
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
          if (c < c_end) consume(pre[i], pre_sc[i], c, nullptr);
          c += 64;
        }
        it_left = it_left > PRE ? it_left - PRE : 0;
      }
    }
    // Every load of up to UN chunk-iterations is issued before the first dot product: a
    // slice of n iterations is one memory round trip when n <= UN (a loop unrolled by a
    // fixed factor runs its remainder one dependent round trip per iteration, which for
    // a 2- or 5-iteration slice was most of the kernel).  4-bit: one iteration is already
    // 2048 weights per row-quad and the expansion needs the registers.
    constexpr int UN = Q ? 1 : EVGPT_GEMV_UNROLL;
    auto run = [&](auto it_tag) {
      constexpr int IT = decltype(it_tag)::value;
      u32x4_t wv[IT][R];
      float sc[IT][R];
      u32x4_t xg[IT][B];
      int cc[IT];
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        cc[i] = c + 64 * i < c_end ? c + 64 * i : c_end - 1;  // lanes past the slice re-read its last chunk
        if constexpr (!STAGE) {
#pragma unroll
          for (int b = 0; b < B; ++b) xg[i][b] = load16(p.x + (static_cast<size_t>(b) * p.K + cc[i] * 8));
        }
      }
#pragma unroll
      for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          wv[i][r] = load_stream16(wp[r] + static_cast<size_t>(cc[i]) * 16);
          if constexpr (Q) sc[i][r] = __builtin_nontemporal_load(sp[r] + (cc[i] >> 1));
          else sc[i][r] = 0.f;
        }
#pragma unroll
      for (int i = 0; i < IT; ++i)
        if (c + 64 * i < c_end) consume(wv[i], sc[i], c + 64 * i, xg[i]);
      c += 64 * IT;
    };
    if constexpr (Q) {
      // 4-bit: one iteration at a time in a plain per-lane loop (the expansion of one
      // iteration already fills the register budget of 4 waves per SIMD)
      for (; c < c_end; c += 64) {
        u32x4_t wv[R];
        float sc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          wv[r] = load_stream16(wp[r] + static_cast<size_t>(c) * 16);
          sc[r] = __builtin_nontemporal_load(sp[r] + (c >> 1));
        }
        consume(wv, sc, c, nullptr);
      }
      it_left = 0;
    } else if constexpr (EVGPT_GEMV_SIMPLE) {
#pragma unroll(EVGPT_GEMV_UNROLL)
      for (; c < c_end; c += 64) {
        u32x4_t wv[R], xg[B];
        float sc[R];
        if constexpr (!STAGE) {
#pragma unroll
          for (int b = 0; b < B; ++b) xg[b] = load16(p.x + (static_cast<size_t>(b) * p.K + c * 8));
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          wv[r] = load_stream16(wp[r] + static_cast<size_t>(c) * 16);
          sc[r] = 0.f;
        }
        consume(wv, sc, c, xg);
      }
      it_left = 0;
    }
    while (it_left > 0) {
      // n <= UN: one shot; n <= 2 UN: two even shots; else UN at a time
      const int n = it_left <= UN ? it_left : (it_left <= 2 * UN ? (it_left + 1) >> 1 : UN);
      switch (n) {
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: if constexpr (UN >= 2) run(std::integral_constant<int, 2>{}); break;
        case 3: if constexpr (UN >= 3) run(std::integral_constant<int, 3>{}); break;
        case 4: if constexpr (UN >= 4) run(std::integral_constant<int, 4>{}); break;
        case 5: if constexpr (UN >= 5) run(std::integral_constant<int, 5>{}); break;
        case 6: if constexpr (UN >= 6) run(std::integral_constant<int, 6>{}); break;
        case 7: if constexpr (UN >= 7) run(std::integral_constant<int, 7>{}); break;
        default: if constexpr (UN >= 8) run(std::integral_constant<int, 8>{}); break;
      }
      it_left -= n;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = wave_sum(acc[r][b]);

    if constexpr (KS > 1) {
      // combine the KS partial sums of each unit through LDS; wave ks==0 finishes
      __syncthreads();  // previous iteration's readers are done with `part`
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int b = 0; b < B; ++b) part[(wid * R + r) * B + b] = acc[r][b];
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int k2 = 1; k2 < KS; ++k2)
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < B; ++b) acc[r][b] += part[((wid + k2) * R + r) * B + b];
      }
    }
    if (!unit_ok || ks != 0) continue;

    // ------------------------------ epilogues ------------------------------
    if constexpr (E == E_LOGITS) {
#pragma unroll
      for (int b = 0; b < B; ++b)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (row[r] < p.N) {
            // fp32 logits: rounding them to bf16 would turn near-ties into exact
            // ties, and a greedy decode must agree with a batched verify of it
            const float v = acc[r][b];
            if (lane == 0 && p.logits) p.logits[static_cast<size_t>(b) * p.N + row[r]] = v;
            if (v > best_val[b]) { best_val[b] = v; best_idx[b] = row[r]; }
          }
        }
    } else if (lane == 0) {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        if constexpr (E == E_STORE) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (row[r] < p.N) {
              float v = acc[r][b];
              if (p.yf) { p.yf[static_cast<size_t>(b) * p.N + row[r]] = v; continue; }
              if (p.bias) v += bf16_to_f32(p.bias[row[r]]);
              v = apply_act(v, p.act);
              p.y[static_cast<size_t>(b) * p.N + row[r]] = f32_to_bf16(v);
            }
          }
        } else if constexpr (E == E_RESIDUAL) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (row[r] < p.N) {
              const size_t o = static_cast<size_t>(b) * p.N + row[r];
              float v = acc[r][b];
              if (p.bias) v += bf16_to_f32(p.bias[row[r]]);
              v = apply_act(v, p.act);
              v = bf16_to_f32(f32_to_bf16(v)) + bf16_to_f32(res_in[r][b]);
              p.y[o] = f32_to_bf16(v);
            }
          }
        } else if constexpr (E == E_SWIGLU) {
          const int half = p.N >> 1;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (row[i] < half) {
              const float g = bf16_to_f32(f32_to_bf16(acc[i][b]));
              const float up = bf16_to_f32(f32_to_bf16(acc[2 + i][b]));
              p.y[static_cast<size_t>(b) * half + row[i]] = f32_to_bf16(silu_f(g) * up);
            }
          }
        } else if constexpr (E == E_QKV_ROPE) {
          const int hd = p.head_dim, hh = hd >> 1;
          const int q_rows = p.n_q_heads * hd, k_rows = p.n_kv_heads * hd;
          // the position is a device counter: clamp it into the cache so a
          // sequence that ran past its capacity overwrites the last slot
          // instead of writing outside the allocation
          const int pos_raw = pos_in[b];
          const int pos = pos_raw < 0 ? 0 : (pos_raw >= p.l_max ? p.l_max - 1 : pos_raw);
          const int slot = p.rows_one_seq ? 0 : b;   // cache slot of this row's sequence
          // a row past the cache of a multi-row pass is dropped, never wrapped onto a live slot
          const bool kv_ok = !p.rows_one_seq || (pos_raw >= 0 && pos_raw < p.l_max);
          if (row[0] < q_rows + k_rows) {
            const int h_all = row[0] / hd, j = row[0] % hd;  // j < hh
            bf16_t* dst;
            if (h_all < p.n_q_heads) {
              dst = p.q_out + static_cast<size_t>(b) * q_rows + h_all * hd;
            } else {
              if (!kv_ok) continue;
              const int h = h_all - p.n_q_heads;
              dst = p.k_cache + ((static_cast<size_t>(slot) * p.n_kv_heads + h) * p.l_max + pos) * hd;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const float c = cs_in[i][b], s = sn_in[i][b];
              const float x1 = bf16_to_f32(f32_to_bf16(acc[i][b]));
              const float x2 = bf16_to_f32(f32_to_bf16(acc[2 + i][b]));
              dst[j + i] = f32_to_bf16(x1 * c - x2 * s);
              dst[j + i + hh] = f32_to_bf16(x2 * c + x1 * s);
            }
          } else {
            if (!kv_ok) continue;
            const int vr = row[0] - q_rows - k_rows;
            const int h = vr / hd, j = vr % hd;
            bf16_t* dst = p.v_cache + ((static_cast<size_t>(slot) * p.n_kv_heads + h) * p.l_max + pos) * hd;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[j + r] = f32_to_bf16(acc[r][b]);
          }
        }
      }
    }
  }

  if constexpr (E == E_LOGITS) {
    // block-level argmax partial: (value, lowest index on ties)
    float* sv = part;                                  // [GEMV_WAVES][B]
    int* si = reinterpret_cast<int*>(part) + GEMV_WAVES * B;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int b = 0; b < B; ++b) { sv[wid * B + b] = best_val[b]; si[wid * B + b] = best_idx[b]; }
    }
    __syncthreads();
    if (tid < B) {
      float bv = sv[tid];
      int bi = si[tid];
      for (int w = 1; w < GEMV_WAVES; ++w)
        if (sv[w * B + tid] > bv || (sv[w * B + tid] == bv && si[w * B + tid] < bi)) {
          bv = sv[w * B + tid]; bi = si[w * B + tid];
        }
      p.part_val[tid * gridDim.x + blockIdx.x] = bv;
      p.part_idx[tid * gridDim.x + blockIdx.x] = bi;
    }
  }
}

// The one-row bf16 instance of the family: the kernel every decode step of a single stream spends its
// time in.  Same layout, prologues and epilogues as gemv_kernel above, with the plainer schedule that
// measures fastest for exactly this case -- the first weight chunks requested ahead of the norm's
// operands, a loop unrolled four chunk-iterations deep, no register cap -- 7B decode step, same box and
// same weight placement: 2.669 ms per token against 2.692 with gemv_kernel<1, ...> in its place (and
// 2.621 against 2.729 with four waves per unit for every matrix).
// Build variants run back to back on one box (7B decode step, ms per token; -DEVGPT_ROW1_...):
// PRE / UNROLL / OCC = 1 / 4 / 1: 2.619; PRE 0: 2.765; PRE 2: 2.674; UNROLL 2: 2.639; UNROLL 8: 2.625;
// OCC 2: 2.620.
#ifndef EVGPT_ROW1_PRE
#define EVGPT_ROW1_PRE 1      // weight chunk-iterations requested ahead of the fused norm
#endif
#ifndef EVGPT_ROW1_UNROLL
#define EVGPT_ROW1_UNROLL 4   // chunk-iterations the main loop is unrolled by
#endif
#ifndef EVGPT_ROW1_OCC
#define EVGPT_ROW1_OCC 1      // blocks per CU the register budget must admit (1: no cap)
#endif
// RR: rows per unit.  The 4096-row residual projections (o_proj, down_proj) are 1024 one-unit blocks at four
// rows per unit -- all resident at once, every block's reduce / store tail exposed at the end of the launch;
// two rows per unit (plain row order only) make 2048 blocks of half the work.
template <int P, int E, int KS, int RR = 4>
__global__ __launch_bounds__(GEMV_WAVES * 64, EVGPT_ROW1_OCC) void gemv_row1_kernel(GemvParams p) {
  static_assert(RR == 4 || RR == 2 || ((E == E_RESIDUAL || E == E_STORE) && RR == 1), "rows per unit");
  constexpr int R = RR;          // shadows the family's constant inside this kernel
  constexpr int NP = RR / 2;     // rotate-half / gate-up pairs per unit
  constexpr int B = 1, Q = 0;
  static_assert(KS == 1 || KS == 2 || KS == 4, "KS waves per unit");
  static_assert(E != E_LOGITS || KS == 1, "logits epilogue runs one wave per unit");
  constexpr int UPB = GEMV_WAVES / KS;  // units per block iteration
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS carve: [partials: GEMV_WAVES*R*B floats][argmax scratch][x image (P_RMSNORM)]
  float* part = reinterpret_cast<float*>(smem);
  constexpr int PART_BYTES = ((GEMV_WAVES * R * B * 4 + 15) / 16) * 16;
  float* red = reinterpret_cast<float*>(smem + PART_BYTES);          // 16 floats
  u32x4_t* xs = reinterpret_cast<u32x4_t*>(smem + PART_BYTES + 64);  // [B][K/8]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ks = wid % KS, ub_slot = wid / KS;
  const int nchunk = p.K >> 3;              // 16-byte chunks of one x row
  const int nwc = Q ? (p.K >> 5) : nchunk;  // 16-byte chunks of one weight row
  const size_t row_bytes = Q ? static_cast<size_t>(p.K >> 1) : static_cast<size_t>(p.K) * 2;
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(p.W);
  const int nblk = p.K >> 6;                // scales per weight row (Q)
  constexpr bool STAGE = (P == P_RMSNORM) || Q;  // x goes through LDS

  // this wave's slice of the K range, in 16-byte weight chunks
  const int cper = (nwc + KS - 1) / KS;
  const int c_begin = ks * cper;
  const int c_end = min(c_begin + cper, nwc);
  const int n_units = (p.N + R - 1) / R;
  const int n_ub = (n_units + UPB - 1) / UPB;

  // ---------------- prologue: normalise x into LDS (fused RMSNorm) ----------------
  // The weights do not depend on x, so the first PRE chunk-iterations of this
  // wave's first unit are requested before the norm: the HBM round trip of the
  // first weight bytes then overlaps the reduction instead of following it.
  constexpr int PRE = EVGPT_ROW1_PRE;  // 4-bit: the registers go to the expansion
  u32x4_t pre[PRE > 0 ? PRE : 1][R];
  float pre_sc[PRE > 0 ? PRE : 1][R];
  if constexpr (STAGE) {
    const int u = blockIdx.x * UPB + ub_slot;
    int row[R];
    unit_rows_n<E, RR>(p, u < n_units ? u : n_units - 1, row);
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
      const int c = c_begin + lane + 64 * i;
      const int cc = c < c_end ? c : 0;  // out-of-slice lanes load a valid address and drop it
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int rr = row[r] < p.N ? row[r] : p.N - 1;
        pre[i][r] = load_stream16(Wb + static_cast<size_t>(rr) * row_bytes + static_cast<size_t>(cc) * 16);
        if constexpr (Q)
          pre_sc[i][r] = __builtin_nontemporal_load(p.absmax + static_cast<size_t>(rr) * nblk + (cc >> 1));
      }
    }
  }
  if constexpr (P == P_RMSNORM) {
    constexpr int XC = 2;  // x chunks a thread keeps in registers between the two passes
    if (nchunk <= XC * GEMV_WAVES * 64) {
      u32x4_t nw[XC];
#pragma unroll
      for (int i = 0; i < XC; ++i) {
        const int c = tid + i * GEMV_WAVES * 64;
        nw[i] = c < nchunk ? load16(p.norm_w + c * 8) : u32x4_t{0, 0, 0, 0};
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        u32x4_t xv[XC];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < XC; ++i) {
          const int c = tid + i * GEMV_WAVES * 64;
          xv[i] = c < nchunk ? load16(p.x + (static_cast<size_t>(b) * p.K + c * 8))
                             : u32x4_t{0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = bf16lo(xv[i][j]), hi = bf16hi(xv[i][j]);
            ss += lo * lo + hi * hi;
          }
        }
        ss = block_sum(ss, red);
        const float rstd = rsqrtf(ss / static_cast<float>(p.K) + p.eps);
#pragma unroll
        for (int i = 0; i < XC; ++i) {
          const int c = tid + i * GEMV_WAVES * 64;
          u32x4_t o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // HF order: round x*rstd to bf16, then multiply by the bf16 weight
            const float lo = bf16_to_f32(f32_to_bf16(bf16lo(xv[i][j]) * rstd)) * bf16lo(nw[i][j]);
            const float hi = bf16_to_f32(f32_to_bf16(bf16hi(xv[i][j]) * rstd)) * bf16hi(nw[i][j]);
            o[j] = pack_bf16x2(lo, hi);
          }
          if (c < nchunk) xs[b * nchunk + xs_slot<Q>(c)] = o;
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        float ss = 0.f;
        for (int c = tid; c < nchunk; c += blockDim.x) {
          u32x4_t v = load16(p.x + (static_cast<size_t>(b) * p.K + c * 8));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float lo = bf16lo(v[i]), hi = bf16hi(v[i]);
            ss += lo * lo + hi * hi;
          }
        }
        ss = block_sum(ss, red);
        const float rstd = rsqrtf(ss / static_cast<float>(p.K) + p.eps);
        for (int c = tid; c < nchunk; c += blockDim.x) {
          u32x4_t v = load16(p.x + (static_cast<size_t>(b) * p.K + c * 8));
          u32x4_t w = load16(p.norm_w + c * 8);
          u32x4_t o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float lo = bf16_to_f32(f32_to_bf16(bf16lo(v[i]) * rstd)) * bf16lo(w[i]);
            const float hi = bf16_to_f32(f32_to_bf16(bf16hi(v[i]) * rstd)) * bf16hi(w[i]);
            o[i] = pack_bf16x2(lo, hi);
          }
          xs[b * nchunk + xs_slot<Q>(c)] = o;
        }
      }
    }
    __syncthreads();
  } else if constexpr (Q) {
#pragma unroll
    for (int b = 0; b < B; ++b)
      for (int c = tid; c < nchunk; c += blockDim.x)
        xs[b * nchunk + xs_slot<Q>(c)] = load16(p.x + (static_cast<size_t>(b) * p.K + c * 8));
    __syncthreads();
  }

  float best_val[B];
  int best_idx[B];
  if constexpr (E == E_LOGITS) {
#pragma unroll
    for (int b = 0; b < B; ++b) { best_val[b] = -INFINITY; best_idx[b] = 0; }
  }

  // Epilogue operands that do not depend on the dot products are requested early, so
  // the tail of a block is one memory round trip (the store), not a chain of them:
  // positions here, the residual rows at the top of each unit.
  int pos_in[B];
  if constexpr (E == E_QKV_ROPE) {
#pragma unroll
    for (int b = 0; b < B; ++b) pos_in[b] = p.rows_one_seq ? p.pos[0] + b : p.pos[b];
  }

  // block-uniform trip count: the partial-sum exchange below contains a barrier
  for (int ub = blockIdx.x; ub < n_ub; ub += gridDim.x) {
    const int u = ub * UPB + ub_slot;
    const bool unit_ok = u < n_units;
    int row[R];
    unit_rows_n<E, RR>(p, unit_ok ? u : n_units - 1, row);
    const unsigned char* wp[R];
    const float* sp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int rr = row[r] < p.N ? row[r] : p.N - 1;  // clamp; out-of-range rows are discarded
      wp[r] = Wb + static_cast<size_t>(rr) * row_bytes;
      if constexpr (Q) sp[r] = p.absmax + static_cast<size_t>(rr) * nblk;
    }
    float acc[R][B];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
    bf16_t res_in[R][B];
    if constexpr (E == E_RESIDUAL) {
      if (lane == 0 && ks == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int b = 0; b < B; ++b)
            res_in[r][b] = row[r] < p.N ? p.res[static_cast<size_t>(b) * p.N + row[r]] : bf16_t(0);
      }
    }

    float cs_in[NP > 0 ? NP : 1][B], sn_in[NP > 0 ? NP : 1][B];
    if constexpr (E == E_QKV_ROPE) {
      if (lane == 0 && ks == 0 && row[0] < (p.n_q_heads + p.n_kv_heads) * p.head_dim) {
        const int hh = p.head_dim >> 1, j = row[0] % p.head_dim;
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const int pc = pos_in[b] < 0 ? 0 : (pos_in[b] >= p.l_max ? p.l_max - 1 : pos_in[b]);
#pragma unroll
          for (int i = 0; i < NP; ++i) {
            cs_in[i][b] = p.cos_tab[static_cast<size_t>(pc) * hh + j + i];
            sn_in[i][b] = p.sin_tab[static_cast<size_t>(pc) * hh + j + i];
          }
        }
      }
    }

    auto consume = [&](const u32x4_t (&wv)[R], const float (&sc)[R], int c) {
      if constexpr (Q) {
        u32x4_t xv[B][4];
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) xv[b][j] = xs[b * nchunk + xs_slot<Q>(4 * c + j)];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          // one row at a time: without the fence the scheduler interleaves the four
          // rows' expansions (64 more live registers) and the kernel spills
          __builtin_amdgcn_sched_barrier(0);
          float s[B];
#pragma unroll
          for (int b = 0; b < B; ++b) s[b] = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned int e[4];
            w4_expand8(wv[r][j], p.code, e);
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
              for (int i = 0; i < 4; ++i) s[b] = dot2_bf16(e[i], xv[b][j][i], s[b]);
          }
#pragma unroll
          for (int b = 0; b < B; ++b) acc[r][b] = fmaf(sc[r], s[b], acc[r][b]);
        }
      } else {
#pragma unroll
        for (int b = 0; b < B; ++b) {
          u32x4_t xv;
          if constexpr (P == P_RMSNORM) xv = xs[b * nchunk + c];
          else xv = load16(p.x + (static_cast<size_t>(b) * p.K + c * 8));
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r][b] = dot8_bf16(wv[r], xv, acc[r][b]);
        }
      }
    };
    int c = c_begin + lane;
    if constexpr (STAGE) {
      if (ub == blockIdx.x) {  // the chunks requested ahead of the prologue
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
          if (c < c_end) consume(pre[i], pre_sc[i], c);
          c += 64;
        }
      }
    }
    // bf16: 4 iterations (16 loads) in flight per lane; 4-bit: one iteration is already
    // 2048 weights per row-quad and the expansion needs the registers
#pragma unroll(EVGPT_ROW1_UNROLL)
    for (; c < c_end; c += 64) {
      u32x4_t wv[R];
      float sc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        wv[r] = load_stream16(wp[r] + static_cast<size_t>(c) * 16);
        if constexpr (Q) sc[r] = __builtin_nontemporal_load(sp[r] + (c >> 1));
      }
      consume(wv, sc, c);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = wave_sum(acc[r][b]);

    if constexpr (KS > 1) {
      // combine the KS partial sums of each unit through LDS; wave ks==0 finishes
      __syncthreads();  // previous iteration's readers are done with `part`
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int b = 0; b < B; ++b) part[(wid * R + r) * B + b] = acc[r][b];
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int k2 = 1; k2 < KS; ++k2)
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < B; ++b) acc[r][b] += part[((wid + k2) * R + r) * B + b];
      }
    }
    if (!unit_ok || ks != 0) continue;

    // ------------------------------ epilogues ------------------------------
    if constexpr (E == E_LOGITS) {
#pragma unroll
      for (int b = 0; b < B; ++b)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (row[r] < p.N) {
            // fp32 logits: rounding them to bf16 would turn near-ties into exact
            // ties, and a greedy decode must agree with a batched verify of it
            const float v = acc[r][b];
            if (lane == 0 && p.logits) p.logits[static_cast<size_t>(b) * p.N + row[r]] = v;
            if (v > best_val[b]) { best_val[b] = v; best_idx[b] = row[r]; }
          }
        }
    } else if (lane == 0) {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        if constexpr (E == E_STORE) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (row[r] < p.N) {
              float v = acc[r][b];
              if (p.yf) { p.yf[static_cast<size_t>(b) * p.N + row[r]] = v; continue; }
              if (p.bias) v += bf16_to_f32(p.bias[row[r]]);
              v = apply_act(v, p.act);
              p.y[static_cast<size_t>(b) * p.N + row[r]] = f32_to_bf16(v);
            }
          }
        } else if constexpr (E == E_RESIDUAL) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (row[r] < p.N) {
              const size_t o = static_cast<size_t>(b) * p.N + row[r];
              float v = acc[r][b];
              if (p.bias) v += bf16_to_f32(p.bias[row[r]]);
              v = apply_act(v, p.act);
              v = bf16_to_f32(f32_to_bf16(v)) + bf16_to_f32(res_in[r][b]);
              p.y[o] = f32_to_bf16(v);
            }
          }
        } else if constexpr (E == E_SWIGLU) {
          const int half = p.N >> 1;
#pragma unroll
          for (int i = 0; i < NP; ++i) {
            if (row[i] < half) {
              const float g = bf16_to_f32(f32_to_bf16(acc[i][b]));
              const float up = bf16_to_f32(f32_to_bf16(acc[NP + i][b]));
              p.y[static_cast<size_t>(b) * half + row[i]] = f32_to_bf16(silu_f(g) * up);
            }
          }
        } else if constexpr (E == E_QKV_ROPE) {
          const int hd = p.head_dim, hh = hd >> 1;
          const int q_rows = p.n_q_heads * hd, k_rows = p.n_kv_heads * hd;
          // the position is a device counter: clamp it into the cache so a
          // sequence that ran past its capacity overwrites the last slot
          // instead of writing outside the allocation
          const int pos_raw = pos_in[b];
          const int pos = pos_raw < 0 ? 0 : (pos_raw >= p.l_max ? p.l_max - 1 : pos_raw);
          const int slot = p.rows_one_seq ? 0 : b;   // cache slot of this row's sequence
          // a row past the cache of a multi-row pass is dropped, never wrapped onto a live slot
          const bool kv_ok = !p.rows_one_seq || (pos_raw >= 0 && pos_raw < p.l_max);
          if (row[0] < q_rows + k_rows) {
            const int h_all = row[0] / hd, j = row[0] % hd;  // j < hh
            bf16_t* dst;
            if (h_all < p.n_q_heads) {
              dst = p.q_out + static_cast<size_t>(b) * q_rows + h_all * hd;
            } else {
              if (!kv_ok) continue;
              const int h = h_all - p.n_q_heads;
              dst = p.k_cache + ((static_cast<size_t>(slot) * p.n_kv_heads + h) * p.l_max + pos) * hd;
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
              const float c = cs_in[i][b], s = sn_in[i][b];
              const float x1 = bf16_to_f32(f32_to_bf16(acc[i][b]));
              const float x2 = bf16_to_f32(f32_to_bf16(acc[NP + i][b]));
              dst[j + i] = f32_to_bf16(x1 * c - x2 * s);
              dst[j + i + hh] = f32_to_bf16(x2 * c + x1 * s);
            }
          } else {
            if (!kv_ok) continue;
            const int vr = row[0] - q_rows - k_rows;
            const int h = vr / hd, j = vr % hd;
            bf16_t* dst = p.v_cache + ((static_cast<size_t>(slot) * p.n_kv_heads + h) * p.l_max + pos) * hd;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[j + r] = f32_to_bf16(acc[r][b]);
          }
        }
      }
    }
  }

  if constexpr (E == E_LOGITS) {
    // block-level argmax partial: (value, lowest index on ties)
    float* sv = part;                                  // [GEMV_WAVES][B]
    int* si = reinterpret_cast<int*>(part) + GEMV_WAVES * B;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int b = 0; b < B; ++b) { sv[wid * B + b] = best_val[b]; si[wid * B + b] = best_idx[b]; }
    }
    __syncthreads();
    if (tid < B) {
      float bv = sv[tid];
      int bi = si[tid];
      for (int w = 1; w < GEMV_WAVES; ++w)
        if (sv[w * B + tid] > bv || (sv[w * B + tid] == bv && si[w * B + tid] < bi)) {
          bv = sv[w * B + tid]; bi = si[w * B + tid];
        }
      p.part_val[tid * gridDim.x + blockIdx.x] = bv;
      p.part_idx[tid * gridDim.x + blockIdx.x] = bi;
    }
  }
}

// ---------------------------------------------------------------------------------------
// One to three rows of x against 4-bit weights (single-stream decode, short verifies, two or three streams).
//
// A 7B projection is 8-45 MB of codes: the whole launch lasts 5-15 us, so what it costs is
// set by how early its bytes are requested and by the work per byte, not by steady-state
// streaming.  This kernel therefore
//   * looks codes up instead of permuting bytes: one code byte (weights j and 4+j of a word)
//     indexes an LDS table of 256 bf16 pairs, kept once per bank so that the 32 lanes of an
//     access group never meet in a bank (2 address instructions + 1 lookup + 1 v_dot2c per
//     two weights, against ~7 instructions by v_perm_b32); the x image in LDS holds the
//     matching pairs (x_j, x_4+j);
//   * gives each wave whole rows (no cross-wave sums, no barrier after the prologue) two at
//     a time -- a gate / up pair, a rotate-half pair or two plain rows -- in steps of
//     W4R_NI x 2048 weights per row, and keeps two steps in flight: the loads of the first
//     two are issued before x is staged and normalised, every later step is requested two
//     steps before it is used.
constexpr int W4R_WAVES = 8;                // waves per block (they share the table and the x image)
constexpr int W4R_NI = 2;                   // 64-lane chunk-iterations per step
constexpr int W4_TAB_BYTES = 256 * 32 * 4;  // 256 code bytes x 32 bank copies of one bf16 pair

// The x image the table path multiplies against: a chunk (x0..x7) is stored as the pairs
// (x0,x4) (x1,x5) (x2,x6) (x3,x7), the two weights one code byte carries.
__device__ __forceinline__ u32x4_t w4_x_image(u32x4_t o) {
  return u32x4_t{perm_b32(o[2], o[0], 0x05040100u), perm_b32(o[2], o[0], 0x07060302u),
                 perm_b32(o[3], o[1], 0x05040100u), perm_b32(o[3], o[1], 0x07060302u)};
}

// bf16 pair of code byte i of w; tab_lane points at this lane's bank copy of entry 0
__device__ __forceinline__ unsigned int w4_lookup(const unsigned char* tab_lane, unsigned int w, int i) {
  return *reinterpret_cast<const unsigned int*>(tab_lane + ((w >> (8 * i)) & 0xFFu) * 128u);
}

// the row pair ("group") g of a projection
template <int E>
__device__ __forceinline__ void w4r_group_rows(const GemvParams& p, int g, int (&row)[2]) {
  if constexpr (E == E_SWIGLU) {
    row[0] = g; row[1] = (p.N >> 1) + g;            // gate row, up row
  } else if constexpr (E == E_QKV_ROPE) {
    const int hd = p.head_dim, hh = hd >> 1;
    const int rot_groups = ((p.n_q_heads + p.n_kv_heads) * hd) >> 1;
    if (g < rot_groups) {
      const int h = g / hh, j = g - h * hh;
      row[0] = h * hd + j; row[1] = h * hd + hh + j;  // the two halves a rotation mixes
    } else {
      row[0] = 2 * g; row[1] = 2 * g + 1;             // v rows
    }
  } else {
    row[0] = 2 * g; row[1] = 2 * g + 1;
  }
}

#ifndef EVGPT_ROWS_BF16_DEFAULT
// bit b: b x rows of bf16 weights go to the rows kernel.  7B decode step, ms, the kernels it replaces vs this
// one: 1 row 2.549 / 2.584 (gemv_row1_kernel stays), 2 rows 2.929 / 2.789, 3 rows 3.334 / 3.057, 4 rows 3.606
// (few-row GEMM path) / 3.389.
#define EVGPT_ROWS_BF16_DEFAULT 28
#endif
constexpr int ROWS_NI_BF16 = 4;   // bf16 weights: 64-lane chunk-iterations (512 weights each) per step
template <int B, int NI>
struct RowsStep {
  u32x4_t w[2][NI];
  float sc[2][NI];          // 4-bit weights only
  unsigned int aux[B][2];   // epilogue operands of the group, per x row: residual pair / cos, sin bits
};

// B x rows (the single-stream step, short speculative verifies, a few streams) share every weight
// register -- with 4-bit weights every lookup: a code byte is looked up once and multiplied into B sums.
// Q = 0 streams bf16 weights through the same pipeline (no table, plain x images, one 16-byte x chunk
// per 16 bytes of weights and x row).
template <int P, int E, int B, int Q>
__global__ __launch_bounds__(W4R_WAVES * 64, (B == 1 && Q) ? 2 : 1) void gemv_rows_kernel(GemvParams p) {
  static_assert(E != E_LOGITS, "the LM head has a kernel of its own");
  static_assert(B >= 1 && B <= 4 && (Q == 0 || B <= 3), "x rows");
  constexpr bool W4 = Q != 0;
  constexpr int NI = W4 ? W4R_NI : ROWS_NI_BF16;   // chunk-iterations per step
  constexpr int IT_SHIFT = W4 ? 5 : 3;             // weights per 16-byte chunk: 32 / 8
  using Step = RowsStep<B, NI>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS carve: [block-sum scratch 64 B][table 32 KB (4-bit)][B x images of K * 2 B]
  float* red = reinterpret_cast<float*>(smem);
  const unsigned char* tab = smem + 64;
  unsigned char* xs_bytes = smem + 64 + (W4 ? W4_TAB_BYTES : 0);
  const unsigned int x_bytes = static_cast<unsigned int>(p.K) * 2u;   // one image
  // the wave index as a scalar: everything derived from it (this wave's rows, its step count,
  // the loop below) is then uniform to the compiler, which branches instead of masking lanes
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nchunk = p.K >> 3;   // 16-byte chunks of x
  const int nwc = p.K >> IT_SHIFT;   // 16-byte chunks of one weight row
  const int nblk = p.K >> 6;         // scales per weight row (4-bit)
  const size_t row_bytes = W4 ? static_cast<size_t>(p.K >> 1) : static_cast<size_t>(p.K) * 2;
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(p.W);
  const int nslab = (nwc + 64 * NI - 1) / (64 * NI);
  const int n_groups = p.N >> 1;
  const int nw = gridDim.x * W4R_WAVES;
  const int gw = blockIdx.x * W4R_WAVES + wid;   // neighbouring waves stream neighbouring rows
  const int n_steps = gw < n_groups ? ((n_groups - gw + nw - 1) / nw) * nslab : 0;

  // Every request below is issued unconditionally, so that the count of loads in flight at
  // each wait is the same on every path (a load under a condition makes the compiler wait
  // for everything): a step past the wave's last one reads chunk 0 of row 0 in all lanes
  // (one cache line) and is never used.
  // A load from an address the compiler can see is the same in all lanes is moved to a scalar
  // register at once, i.e. waited for where it is issued.  The epilogue operands are such loads
  // (positions, one residual pair, one cos / sin per group and x row), so their addresses get a zero
  // the compiler cannot see through and the values stay in vector registers until used.
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  int pos_raw[B];
#pragma unroll
  for (int b = 0; b < B; ++b) pos_raw[b] = 0;
  if constexpr (E == E_QKV_ROPE) {   // the oldest loads: cos / sin wait for them alone
#pragma unroll
    for (int b = 0; b < B; ++b) pos_raw[b] = p.rows_one_seq ? p.pos[vzero] + b : p.pos[b + vzero];
  }

  // Row bases are uniform (scalar registers); a lane adds an unsigned 32-bit offset, which is the
  // addressing form that needs no 64-bit vector arithmetic per load.
  auto issue = [&](Step& st, int g, int slab) {
    const bool live = g < n_groups;
    int row[2];
    w4r_group_rows<E>(p, live ? g : 0, row);
    unsigned int woff[NI], soff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = (slab * NI + i) * 64 + lane;
      const int cc = live ? (c < nwc ? c : nwc - 1) : 0;   // lanes past the row re-read its last chunk and drop it
      woff[i] = static_cast<unsigned int>(cc) * 16u;
      soff[i] = static_cast<unsigned int>(cc >> 1) * 4u;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const unsigned char* wp = Wb + static_cast<size_t>(row[r]) * row_bytes;
#pragma unroll
      for (int i = 0; i < NI; ++i) st.w[r][i] = load_stream16(wp + woff[i]);
      if constexpr (W4) {
        const unsigned char* sp = reinterpret_cast<const unsigned char*>(p.absmax + static_cast<size_t>(row[r]) * nblk);
#pragma unroll
        for (int i = 0; i < NI; ++i)
          st.sc[r][i] = __builtin_nontemporal_load(reinterpret_cast<const float*>(sp + soff[i]));
      }
    }
  };
  // epilogue operands of group g, requested with the weights (a load inside the epilogue
  // would have to wait for every weight request before it)
  auto issue_aux = [&](Step& st, int g) {
    const int gg = g < n_groups ? g : 0;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      st.aux[b][0] = 0; st.aux[b][1] = 0;
      if constexpr (E == E_RESIDUAL) {   // rows 2g, 2g + 1 of x row b
        st.aux[b][0] = *reinterpret_cast<const unsigned int*>(p.res + static_cast<size_t>(b) * p.N + 2 * gg + vzero);
      } else if constexpr (E == E_QKV_ROPE) {
        const int hh = p.head_dim >> 1;
        const int rot_groups = ((p.n_q_heads + p.n_kv_heads) * p.head_dim) >> 1;
        const int pc = pos_raw[b] < 0 ? 0 : (pos_raw[b] >= p.l_max ? p.l_max - 1 : pos_raw[b]);
        const int j = gg < rot_groups ? gg % hh : 0;
        st.aux[b][0] = __float_as_uint(p.cos_tab[static_cast<size_t>(pc) * hh + j + vzero]);
        st.aux[b][1] = __float_as_uint(p.sin_tab[static_cast<size_t>(pc) * hh + j + vzero]);
      }
    }
  };

  // ---- the first requests: x and the norm weights (they gate the prologue), then two steps of weights
  constexpr int XC = 3;   // x chunks per thread and row: K <= 3 * 512 * 8
  u32x4_t xr[B][XC], nwt[XC];
#pragma unroll
  for (int i = 0; i < XC; ++i) {
    const int c = tid + i * W4R_WAVES * 64;
    const int cc = c < nchunk ? c : 0;
#pragma unroll
    for (int b = 0; b < B; ++b) xr[b][i] = load16(p.x + static_cast<size_t>(b) * p.K + cc * 8);
    if constexpr (P == P_RMSNORM) nwt[i] = load16(p.norm_w + cc * 8);
    else nwt[i] = u32x4_t{0, 0, 0, 0};
  }
  Step b0, b1;
  int g0 = gw, s0 = 0;          // the step held in b0
  int g1 = gw, s1 = 1;          // the step held in b1
  if (s1 == nslab) { s1 = 0; g1 += nw; }
  // fences: the compiler may not interleave the requests of two steps (the waits count on their order)
  __builtin_amdgcn_sched_barrier(0);
  issue(b0, g0, s0);
  issue_aux(b0, g0);     // same order as in the loop: the waits are sized for the shorter of the two
  __builtin_amdgcn_sched_barrier(0);
  issue(b1, g1, s1);
  issue_aux(b1, g1);
  __builtin_amdgcn_sched_barrier(0);

  // ---- table: entry t = (value[t & 15], value[t >> 4]), 32 copies 4 bytes apart
  if (W4 && tid < 256) {
    unsigned int e[4];
    w4_expand8(static_cast<unsigned int>(tid), p.code, e);   // weight 0 = code t & 15, weight 4 = code t >> 4
    const unsigned int v = (e[0] & 0xFFFFu) | (e[2] << 16);
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(smem + 64 + tid * 128);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = u32x4_t{v, v, v, v};
  }

  // ---- x images (fused RMSNorm in HF order: round x * rstd to bf16, then multiply by the bf16 weight)
#pragma unroll
  for (int b = 0; b < B; ++b) {
    u32x4_t* xs = reinterpret_cast<u32x4_t*>(xs_bytes + b * x_bytes);
    if constexpr (P == P_RMSNORM) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < XC; ++i) {
        if (tid + i * W4R_WAVES * 64 < nchunk) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = bf16lo(xr[b][i][j]), hi = bf16hi(xr[b][i][j]);
            ss += lo * lo + hi * hi;
          }
        }
      }
      ss = block_sum(ss, red);
      const float rstd = rsqrtf(ss / static_cast<float>(p.K) + p.eps);
#pragma unroll
      for (int i = 0; i < XC; ++i) {
        const int c = tid + i * W4R_WAVES * 64;
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = bf16_to_f32(f32_to_bf16(bf16lo(xr[b][i][j]) * rstd)) * bf16lo(nwt[i][j]);
          const float hi = bf16_to_f32(f32_to_bf16(bf16hi(xr[b][i][j]) * rstd)) * bf16hi(nwt[i][j]);
          o[j] = pack_bf16x2(lo, hi);
        }
        if (c < nchunk) xs[xs_slot<Q>(c)] = W4 ? w4_x_image(o) : o;
      }
    } else {
#pragma unroll
      for (int i = 0; i < XC; ++i) {
        const int c = tid + i * W4R_WAVES * 64;
        if (c < nchunk) xs[xs_slot<Q>(c)] = W4 ? w4_x_image(xr[b][i]) : xr[b][i];
      }
    }
  }
  __syncthreads();   // the last barrier: no wave waits for another below

  const unsigned char* tab_lane = tab + (lane & 31) * 4;
  float acc[2][B];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int b = 0; b < B; ++b) acc[r][b] = 0.f;

  // Where this lane's x chunks of iteration 0 sit in an image (bytes).  4-bit: four chunks per weight chunk,
  // iteration i of slab s (s * NI + i) * 4096 bytes on (the swizzle of xs_slot depends on the lane alone);
  // bf16: one chunk, 1024 bytes per iteration.
  unsigned int xoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) xoff[j] = W4 ? static_cast<unsigned int>(xs_slot<1>(4 * lane + j)) * 16u : lane * 16u;

  // 4-bit: one step = four batches (iteration, row) of 16 lookups and 16 B v_dot2c.  The lookups of a
  // batch are issued a batch ahead of their dot products, behind fences the scheduler may not
  // move code across: left alone it emits eight lookups, waits for them and runs eight dependent
  // dot products, and the LDS round trip of every such group (sixteen per step) is exposed.
  // No branch around the arithmetic either (a skipped wait is a wait the compiler repeats in
  // full later): a lane past the end of the row multiplies its re-read chunk by a zero scale
  // (bf16: its sum is dropped by a select).
  auto lookups = [&](const u32x4_t& w, unsigned int (&e)[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) e[4 * j + k] = w4_lookup(tab_lane, w[j], k);
  };
  auto xread = [&](u32x4_t (&xv)[B][4], unsigned int xit) {
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        xv[b][j] = *reinterpret_cast<const u32x4_t*>(xs_bytes + b * x_bytes + xit + xoff[j]);
  };
  auto dots = [&](const unsigned int (&e)[16], const u32x4_t (&xv)[B][4], float scale, float (&a)[B]) {
#pragma unroll
    for (int b = 0; b < B; ++b) {
      float s0 = 0.f, s1 = 0.f;   // two chains: a v_dot2c does not wait for the one before it
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0 = dot2_bf16(e[4 * j + 0], xv[b][j][0], s0);
        s1 = dot2_bf16(e[4 * j + 1], xv[b][j][1], s1);
        s0 = dot2_bf16(e[4 * j + 2], xv[b][j][2], s0);
        s1 = dot2_bf16(e[4 * j + 3], xv[b][j][3], s1);
      }
      a[b] = fmaf(scale, s0 + s1, a[b]);
    }
  };
  static_assert(W4R_NI == 2, "the 4-bit batch schedule below is written out for two iterations per step");
  auto consume = [&](const Step& st, int slab) {
    if constexpr (!W4) {
      // bf16: the x chunks of all iterations first (one LDS round trip per step), then the dot products
      u32x4_t xv[NI][B];
      bool ok[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int it = slab * NI + i;
        ok[i] = it * 64 + lane < nwc;
        const unsigned int xit = ok[i] ? static_cast<unsigned int>(it) * 1024u : 0u;
#pragma unroll
        for (int b = 0; b < B; ++b)
          xv[i][b] = *reinterpret_cast<const u32x4_t*>(xs_bytes + b * x_bytes + xit + xoff[0]);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int b = 0; b < B; ++b) {
            const float t = dot8_bf16(st.w[r][i], xv[i][b], 0.f);
            acc[r][b] += ok[i] ? t : 0.f;
          }
    } else {
    bool ok[2];
    unsigned int xit[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int it = slab * W4R_NI + i;
      ok[i] = it * 64 + lane < nwc;
      xit[i] = ok[i] ? static_cast<unsigned int>(it) * 4096u : 0u;
    }
    unsigned int ea[16], eb[16];
    if constexpr (B == 1) {
      u32x4_t xv0[B][4], xv1[B][4];
      xread(xv0, xit[0]);
      lookups(st.w[0][0], ea);
      __builtin_amdgcn_sched_barrier(0);
      lookups(st.w[1][0], eb);
      xread(xv1, xit[1]);
      __builtin_amdgcn_sched_barrier(0);
      dots(ea, xv0, ok[0] ? st.sc[0][0] : 0.f, acc[0]);
      __builtin_amdgcn_sched_barrier(0);
      lookups(st.w[0][1], ea);
      __builtin_amdgcn_sched_barrier(0);
      dots(eb, xv0, ok[0] ? st.sc[1][0] : 0.f, acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      lookups(st.w[1][1], eb);
      __builtin_amdgcn_sched_barrier(0);
      dots(ea, xv1, ok[1] ? st.sc[0][1] : 0.f, acc[0]);
      __builtin_amdgcn_sched_barrier(0);
      dots(eb, xv1, ok[1] ? st.sc[1][1] : 0.f, acc[1]);
    } else {
      // more x rows: one set of x registers, re-read for the second iteration after its last use for the first
      u32x4_t xv[B][4];
      xread(xv, xit[0]);
      lookups(st.w[0][0], ea);
      __builtin_amdgcn_sched_barrier(0);
      lookups(st.w[1][0], eb);
      __builtin_amdgcn_sched_barrier(0);
      dots(ea, xv, ok[0] ? st.sc[0][0] : 0.f, acc[0]);
      __builtin_amdgcn_sched_barrier(0);
      lookups(st.w[0][1], ea);
      __builtin_amdgcn_sched_barrier(0);
      dots(eb, xv, ok[0] ? st.sc[1][0] : 0.f, acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      xread(xv, xit[1]);
      lookups(st.w[1][1], eb);
      __builtin_amdgcn_sched_barrier(0);
      dots(ea, xv, ok[1] ? st.sc[0][1] : 0.f, acc[0]);
      __builtin_amdgcn_sched_barrier(0);
      dots(eb, xv, ok[1] ? st.sc[1][1] : 0.f, acc[1]);
    }
    }
  };

  // the row sums of group g are complete: epilogue by lane 0 (stores only)
  auto finish = [&](const Step& st, int g) {
    float a0[B], a1[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      a0[b] = wave_sum(acc[0][b]); a1[b] = wave_sum(acc[1][b]);
      acc[0][b] = 0.f; acc[1][b] = 0.f;
    }
    if (lane != 0) return;
    int row[2];
    w4r_group_rows<E>(p, g, row);
#pragma unroll
    for (int b = 0; b < B; ++b) {
      if constexpr (E == E_STORE) {
        const float a[2] = {a0[b], a1[b]};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if (p.yf) p.yf[static_cast<size_t>(b) * p.N + row[r]] = a[r];
          else p.y[static_cast<size_t>(b) * p.N + row[r]] = f32_to_bf16(a[r]);
        }
      } else if constexpr (E == E_RESIDUAL) {
        const float a[2] = {a0[b], a1[b]};
        const float res[2] = {bf16lo(st.aux[b][0]), bf16hi(st.aux[b][0])};
        unsigned int out = 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const float sum = bf16_to_f32(f32_to_bf16(a[r])) + res[r];
          out |= static_cast<unsigned int>(__builtin_bit_cast(unsigned short, f32_to_bf16(sum))) << (16 * r);
        }
        *reinterpret_cast<unsigned int*>(p.y + static_cast<size_t>(b) * p.N + row[0]) = out;
      } else if constexpr (E == E_SWIGLU) {
        const float gate = bf16_to_f32(f32_to_bf16(a0[b])), up = bf16_to_f32(f32_to_bf16(a1[b]));
        p.y[static_cast<size_t>(b) * (p.N >> 1) + row[0]] = f32_to_bf16(silu_f(gate) * up);
      } else if constexpr (E == E_QKV_ROPE) {
        const int hd = p.head_dim, hh = hd >> 1;
        const int q_rows = p.n_q_heads * hd, k_rows = p.n_kv_heads * hd;
        // the position is a device counter: clamp it into the cache so a sequence that ran
        // past its capacity overwrites the last slot instead of writing outside the allocation
        const int pos = pos_raw[b] < 0 ? 0 : (pos_raw[b] >= p.l_max ? p.l_max - 1 : pos_raw[b]);
        const int slot = p.rows_one_seq ? 0 : b;   // cache slot of this row's sequence
        // a row past the cache of a multi-row pass is dropped, never wrapped onto a live slot
        const bool kv_ok = !p.rows_one_seq || (pos_raw[b] >= 0 && pos_raw[b] < p.l_max);
        if (row[0] < q_rows + k_rows) {
          const int h_all = row[0] / hd, j = row[0] - h_all * hd;   // j < hh
          bf16_t* dst;
          if (h_all < p.n_q_heads) {
            dst = p.q_out + static_cast<size_t>(b) * q_rows + h_all * hd;
          } else {
            if (!kv_ok) continue;
            dst = p.k_cache + ((static_cast<size_t>(slot) * p.n_kv_heads + (h_all - p.n_q_heads)) * p.l_max + pos) * hd;
          }
          const float c = __uint_as_float(st.aux[b][0]), sn = __uint_as_float(st.aux[b][1]);
          const float x1 = bf16_to_f32(f32_to_bf16(a0[b])), x2 = bf16_to_f32(f32_to_bf16(a1[b]));
          dst[j] = f32_to_bf16(x1 * c - x2 * sn);
          dst[j + hh] = f32_to_bf16(x2 * c + x1 * sn);
        } else {
          if (!kv_ok) continue;
          const int vr = row[0] - q_rows - k_rows;
          const int h = vr / hd, j = vr - h * hd;
          bf16_t* dst = p.v_cache + ((static_cast<size_t>(slot) * p.n_kv_heads + h) * p.l_max + pos) * hd;
          dst[j] = f32_to_bf16(a0[b]);
          dst[j + 1] = f32_to_bf16(a1[b]);
        }
      }
    }
  };

  auto advance2 = [&](int& g, int& s) {   // two steps on
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (++s == nslab) { s = 0; g += nw; }
  };

  // Pairs of steps; an odd last step follows the loop (a break between the two halves would
  // give the loop a second exit whose request count the waits of the first half had to assume).
  int t = 0;
  for (; t + 1 < n_steps; t += 2) {
    consume(b0, s0);
    if (s0 == nslab - 1) finish(b0, g0);
    advance2(g0, s0);
    __builtin_amdgcn_sched_barrier(0);
    issue(b0, g0, s0);
    issue_aux(b0, g0);
    __builtin_amdgcn_sched_barrier(0);
    consume(b1, s1);
    if (s1 == nslab - 1) finish(b1, g1);
    advance2(g1, s1);
    __builtin_amdgcn_sched_barrier(0);
    issue(b1, g1, s1);
    issue_aux(b1, g1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (t < n_steps) {
    consume(b0, s0);
    if (s0 == nslab - 1) finish(b0, g0);
  }
}

// shapes the rows kernel takes: 1..3 x rows of 4-bit weights or (EVGPT_ROWS_BF16: the x rows it takes)
// bf16 weights, row pairs, whole scale blocks, the table and the x images inside a CU's LDS
static bool rows_kernel_ok(const GemvParams& p, int B, int E) {
  static const bool on = [] { const char* e = getenv("EVGPT_W4_ROW1"); return !e || atoi(e) != 0; }();
  static const int max_b = [] { const char* e = getenv("EVGPT_W4_ROWS_MAX"); return e ? atoi(e) : 3; }();
  // bf16: bit b set = take b x rows
  static const int bf16_rows = [] { const char* e = getenv("EVGPT_ROWS_BF16"); return e ? atoi(e) : EVGPT_ROWS_BF16_DEFAULT; }();
  if (E == E_LOGITS || B < 1) return false;
  if (p.absmax) {
    if (!on || B > 3 || B > max_b) return false;
    if (p.K % 64 != 0) return false;
  } else {
    // one row without the fused norm (o_proj, down_proj) goes here as well: 7B decode step, three alternating
    // runs, 2.548 -> 2.542 ms (the norm-fused projections measure the other way: 2.549 -> 2.584 with all four)
    static const int res1 = [] { const char* e = getenv("EVGPT_ROWS_BF16_RES1"); return e ? atoi(e) : 1; }();
    // bit E of EVGPT_ROWS_BF16_E1 sends one row with that epilogue here whatever the prologue.  7B decode step,
    // two alternating runs: none 2.540 / 2.544, q|k|v (bit 3) 2.527 / 2.528, gate/up (bit 2) 2.545 / 2.544,
    // both 2.539 / 2.537: q|k|v is 6 144 row pairs = three per wave exactly.
    static const int e1 = [] { const char* e = getenv("EVGPT_ROWS_BF16_E1"); return e ? atoi(e) : 8; }();
    const bool take = ((bf16_rows >> B) & 1) || (B == 1 && res1 && !p.norm_w && (E == E_RESIDUAL || E == E_STORE)) ||
                      (B == 1 && ((e1 >> E) & 1));
    if (B > 4 || !take || p.K % 8 != 0) return false;
  }
  if (p.bias || p.act != ACT_NONE) return false;   // no load may sit in this kernel's epilogue
  if (p.N % 2 != 0) return false;
  if (E == E_QKV_ROPE && p.head_dim % 2 != 0) return false;
  if (64 + (p.absmax ? W4_TAB_BYTES : 0) + static_cast<size_t>(B) * p.K * 2 > 128 * 1024) return false;
  return p.K <= 3 * W4R_WAVES * 64 * 8;   // the x chunks a block keeps in registers through the norm
}

template <int P, int E, int B, int Q>
static void launch_rows_b(const GemvParams& p, hipStream_t st) {
  const size_t smem = 64 + (Q ? W4_TAB_BYTES : 0) + static_cast<size_t>(B) * p.K * 2;
  // one block per CU (two would hold their tables and x images at once; 256 measured ahead of 512);
  // fewer blocks when there are fewer row pairs than waves
  static const int cap = [] { const char* e = getenv("EVGPT_W4_ROW1_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 256; }();
  static const bool attr = [] {   // more than the 64 KB a launch may ask for without opting in (several long x rows)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_rows_kernel<P, E, B, Q>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
  }();
  (void)attr;
  const int groups = p.N >> 1;
  const int want = (groups + W4R_WAVES - 1) / W4R_WAVES;
  int blocks = want < cap ? want : cap;
  // bf16: every wave the same number of row pairs -- 11 008 pairs over 2 048 waves are 5 or 6 each and the
  // launch lasts as long as 6, so fewer blocks (ceil(pairs / waves) pairs per wave, a few CUs idle) lose
  // nothing: three streams 3.055 -> 3.013 ms per step.  4-bit weights: level (1.48-1.50 either way), kept
  // as it was; the other rounding (more blocks, some CUs host two) is 10 % slower for both.
  static const int mode = [] { const char* e = getenv("EVGPT_ROWS_BALANCE"); return e ? atoi(e) : -1; }();
  if ((mode < 0 ? Q == 0 : mode == 1) && want > cap) {
    const int per_wave = (groups + cap * W4R_WAVES - 1) / (cap * W4R_WAVES);
    blocks = (groups + per_wave * W4R_WAVES - 1) / (per_wave * W4R_WAVES);
  }
  hipLaunchKernelGGL((gemv_rows_kernel<P, E, B, Q>), dim3(blocks), dim3(W4R_WAVES * 64), smem, st, p);
}

template <int P, int E>
static void launch_rows(const GemvParams& p, int B, hipStream_t st) {
  if (p.absmax) {
    switch (B) {
      case 1: launch_rows_b<P, E, 1, 1>(p, st); break;
      case 2: launch_rows_b<P, E, 2, 1>(p, st); break;
      default: launch_rows_b<P, E, 3, 1>(p, st); break;
    }
  } else {
    switch (B) {
      case 1: launch_rows_b<P, E, 1, 0>(p, st); break;
      case 2: launch_rows_b<P, E, 2, 0>(p, st); break;
      case 3: launch_rows_b<P, E, 3, 0>(p, st); break;
      default: launch_rows_b<P, E, 4, 0>(p, st); break;
    }
  }
}

static int g_forced_ks = 0;
void gemv_force_ks(int ks) { g_forced_ks = ks; }   // 0 = automatic; for tuning runs

// EVGPT_GEMV_KS="NxK:ks,NxK:ks" overrides the choice for single shapes (tuning runs)
static int env_ks(int N, int K) {
  static const char* e = getenv("EVGPT_GEMV_KS");
  if (!e) return 0;
  for (const char* s = e; *s;) {
    int n = 0, k = 0, ks = 0, used = 0;
    if (sscanf(s, "%dx%d:%d%n", &n, &k, &ks, &used) != 3) break;
    if (n == N && k == K) return ks;
    s += used;
    if (*s == ',') ++s;
  }
  return 0;
}

// waves-per-unit so that the launch has >= ~6000 waves (24 per CU)
static int pick_ks(int N, int K, bool one_row_bf16) {
  if (g_forced_ks == 1 || g_forced_ks == 2 || g_forced_ks == 4) return g_forced_ks;
  const int ek = env_ks(N, K);
  if (ek == 1 || ek == 2 || ek == 4) return ek;
  const int units = (N + R - 1) / R;
  // One bf16 row (single-stream decode): four waves per unit for every matrix of the 7B decoder,
  // also the 12288- and 22016-row ones that two waves per unit would already spread over 6000+
  // waves.  7B decode step on one box with the weights placed alike, ms per token: qkv 2 -> 4 waves
  // 2.681 -> 2.659, gate/up 2 -> 4 2.681 -> 2.653, both 2.665 -> 2.621; one wave per unit 2.72 / 2.80;
  // two for down_proj 2.77, two for o_proj 2.74.  More rows or 4-bit weights go the other way
  // (two rows 3.17 -> 3.40 ms per verify pass, three 3.47 -> 4.08, NF4 step 1.766 -> 1.789) and keep
  // the wave-count rule.
  if (one_row_bf16 && units < 6000) return 4;
  if (units >= 6000) return 1;
  if (units >= 3000) return 2;
  return 4;
}

static int gemv_grid(int N, int ks) {
  const int units = (N + R - 1) / R;
  const int upb = GEMV_WAVES / ks;
  const int blocks = (units + upb - 1) / upb;
  // 8 blocks per CU are resident.  Measured on MI355X (7B decode step): letting
  // the dispatcher hand out one-unit blocks beats a 2048-block grid whose blocks loop
  // (2.79 vs 2.82 ms per token; 5504 one-unit blocks for gate/up vs 4096 looping ones
  // 2.616 vs 2.621), and an even split of units over fewer blocks is slower than both.
  // EVGPT_GEMV_CAP overrides for tuning.
  static const int cap = [] {
    const char* e = getenv("EVGPT_GEMV_CAP");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 8192;
  }();
  return blocks < cap ? blocks : cap;
}

template <int P, int E, int KS, int Q>
static void launch_bk(const GemvParams& p, int B, hipStream_t st) {
  const size_t part = ((GEMV_WAVES * R * B * 4 + 15) / 16) * 16;
  const size_t smem = part + 64 + ((P == P_RMSNORM || Q) ? static_cast<size_t>(B) * (p.K / 8) * 16 : 0);
  // 4-bit blocks stage x for every projection and do a quarter of the streaming per unit:
  // 1024 looping blocks (4 resident per CU) amortise that (1.92 -> 1.87 ms per 7B step)
  static const int cap4 = [] { const char* e = getenv("EVGPT_W4_CAP"); return e && atoi(e) > 0 ? atoi(e) : 1024; }();
  const int full = gemv_grid(p.N, KS);
  dim3 g(Q && full > cap4 ? cap4 : full), b(GEMV_WAVES * 64);
  switch (B) {
    case 1:
      if constexpr (Q == 0) {
        if constexpr (KS == 4 && E != E_LOGITS) {      // measured at four waves per unit only
          // Rows per unit of the single-stream step (one unit per block at four waves per unit).  7B decode step,
          // alternating on one box: residual projections (o_proj, down_proj: no prologue) 4 / 2 / 1 rows per unit
          // 2.630 / 2.601 / 2.577 ms per token; the projections with the fused norm (qkv, gate/up) at one
          // rotate-half or gate / up pair per unit instead of two 2.60 -> 2.74: every block stages and normalises
          // x, and twice the blocks pay that twice.  EVGPT_ROW1_RR / EVGPT_ROW1_RR_PAIR override for tuning runs
          // (residual: 1, 2, 4; rope / SwiGLU epilogues: 2, 4).
          static const int rr_res = [] { const char* e = getenv("EVGPT_ROW1_RR"); return e ? atoi(e) : 1; }();
          static const int rr_pair = [] { const char* e = getenv("EVGPT_ROW1_RR_PAIR"); return e ? atoi(e) : 4; }();
          constexpr bool plain = E == E_RESIDUAL || E == E_STORE;
          constexpr int upb = GEMV_WAVES / KS;          // units per block
          const int rr = plain ? rr_res : rr_pair;
          const bool fits = p.N % 4 == 0 && (E != E_QKV_ROPE || p.head_dim % 4 == 0);
          if (fits && rr == 2) {
            hipLaunchKernelGGL((gemv_row1_kernel<P, E, KS, 2>), dim3(((p.N + 1) / 2 + upb - 1) / upb), b, smem, st, p);
            break;
          }
          if constexpr (plain) {
            if (fits && rr == 1) {
              hipLaunchKernelGGL((gemv_row1_kernel<P, E, KS, 1>), dim3((p.N + upb - 1) / upb), b, smem, st, p);
              break;
            }
          }
        }
        hipLaunchKernelGGL((gemv_row1_kernel<P, E, KS>), g, b, smem, st, p);
      } else {
        hipLaunchKernelGGL((gemv_kernel<1, P, E, KS, Q>), g, b, smem, st, p);
      }
      break;
    case 2: hipLaunchKernelGGL((gemv_kernel<2, P, E, KS, Q>), g, b, smem, st, p); break;
    case 3: hipLaunchKernelGGL((gemv_kernel<3, P, E, KS, Q>), g, b, smem, st, p); break;
    case 4:
      // 4-bit weights keep 16 x registers per row and are routed to <= 3 rows by the caller
      if constexpr (Q == 0) hipLaunchKernelGGL((gemv_kernel<4, P, E, KS, Q>), g, b, smem, st, p);
      break;
    default: break;
  }
}

template <int P, int E>
static void launch_b(const GemvParams& p, int B, hipStream_t st) {
  if constexpr (E == E_LOGITS) {
    launch_bk<P, E, 1, 0>(p, B, st);
  } else if (rows_kernel_ok(p, B, E)) {
    launch_rows<P, E>(p, B, st);
  } else if (p.absmax) {
    // a 4-bit row is a quarter of the bytes, so half the waves per unit keep the per-wave
    // work (and the share of a block's prologue) comparable; measured on the 7B decode
    // step: 2.12 ms with the bf16 choice, 1.92 ms with half (EVGPT_W4_KS_SHIFT to tune)
    static const int shift = [] { const char* e = getenv("EVGPT_W4_KS_SHIFT"); return e ? atoi(e) : 1; }();
    int ks = pick_ks(p.N, p.K, false) >> shift;
    switch (ks < 1 ? 1 : ks) {
      case 1: launch_bk<P, E, 1, 1>(p, B, st); break;
      case 2: launch_bk<P, E, 2, 1>(p, B, st); break;
      default: launch_bk<P, E, 4, 1>(p, B, st); break;
    }
  } else {
    switch (pick_ks(p.N, p.K, B == 1)) {
      case 1: launch_bk<P, E, 1, 0>(p, B, st); break;
      case 2: launch_bk<P, E, 2, 0>(p, B, st); break;
      default: launch_bk<P, E, 4, 0>(p, B, st); break;
    }
  }
}

int gemv_max_batch() { return 4; }

int gemv_logits_grid(int N) { return gemv_grid(N, 1); }

int gemv_w4_max_batch() { return 3; }

static void set_w4(GemvParams& p, const W4Desc* w4) {
  if (w4) { p.absmax = w4->absmax; p.code = w4->code; }
}

void launch_gemv(const bf16_t* W, const bf16_t* x, bf16_t* y, const bf16_t* bias,
                 const bf16_t* res, const bf16_t* norm_w, float eps, int act, int B, int N,
                 int K, hipStream_t st, const W4Desc* w4, float* y_f32) {
  GemvParams p{};
  set_w4(p, w4);
  p.yf = y_f32;
  p.W = W; p.x = x; p.y = y; p.bias = bias; p.res = res; p.norm_w = norm_w; p.eps = eps;
  p.act = act; p.N = N; p.K = K;
  if (res) {
    if (norm_w) launch_b<P_RMSNORM, E_RESIDUAL>(p, B, st);
    else launch_b<P_NONE, E_RESIDUAL>(p, B, st);
  } else {
    if (norm_w) launch_b<P_RMSNORM, E_STORE>(p, B, st);
    else launch_b<P_NONE, E_STORE>(p, B, st);
  }
}

void launch_gemv_swiglu(const bf16_t* W_gate_up, const bf16_t* x, bf16_t* y,
                        const bf16_t* norm_w, float eps, int B, int inter, int K,
                        hipStream_t st, const W4Desc* w4) {
  GemvParams p{};
  set_w4(p, w4);
  p.W = W_gate_up; p.x = x; p.y = y; p.norm_w = norm_w; p.eps = eps; p.N = 2 * inter; p.K = K;
  if (norm_w) launch_b<P_RMSNORM, E_SWIGLU>(p, B, st);
  else launch_b<P_NONE, E_SWIGLU>(p, B, st);
}

void launch_gemv_qkv_rope(const bf16_t* W_qkv, const bf16_t* x, const bf16_t* norm_w, float eps,
                          bf16_t* q_out, bf16_t* k_cache, bf16_t* v_cache, const float* cos_tab,
                          const float* sin_tab, const int* pos, int B, int n_q_heads,
                          int n_kv_heads, int head_dim, int l_max, int K, hipStream_t st,
                          int rows_one_seq, const W4Desc* w4) {
  GemvParams p{};
  set_w4(p, w4);
  p.rows_one_seq = rows_one_seq;
  p.W = W_qkv; p.x = x; p.norm_w = norm_w; p.eps = eps; p.K = K;
  p.N = (n_q_heads + 2 * n_kv_heads) * head_dim;
  p.q_out = q_out; p.k_cache = k_cache; p.v_cache = v_cache; p.cos_tab = cos_tab;
  p.sin_tab = sin_tab; p.pos = pos; p.n_q_heads = n_q_heads; p.n_kv_heads = n_kv_heads;
  p.head_dim = head_dim; p.l_max = l_max;
  if (norm_w) launch_b<P_RMSNORM, E_QKV_ROPE>(p, B, st);
  else launch_b<P_NONE, E_QKV_ROPE>(p, B, st);
}

void launch_gemv_logits(const bf16_t* W, const bf16_t* x, const bf16_t* norm_w, float eps,
                        float* logits, float* part_val, int* part_idx, int B, int N, int K,
                        hipStream_t st) {
  GemvParams p{};
  p.W = W; p.x = x; p.norm_w = norm_w; p.eps = eps; p.N = N; p.K = K;
  p.logits = logits; p.part_val = part_val; p.part_idx = part_idx;
  if (norm_w) launch_b<P_RMSNORM, E_LOGITS>(p, B, st);
  else launch_b<P_NONE, E_LOGITS>(p, B, st);
}

}  // namespace evgpt
