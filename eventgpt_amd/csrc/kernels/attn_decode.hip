// Single-query (decode) attention over a static KV cache, split along the
// key axis ("flash-decoding") so that 32 heads still fill 256 CUs.
//
// Memory-bound: every K and V byte is read once.  K/V rows go straight to
// VGPRs (16 B per lane, a 128-wide row = 16 lanes, so one wave-instruction
// covers 4 keys); no LDS staging since nothing is reused across waves.  Each
// 16-lane group keeps its own online-softmax state; groups, waves and finally
// splits are merged with the usual (m, l, o) rescaling.  The cache length is
// read from device memory so the launch can be replayed from a graph.
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

constexpr int AD_WAVES = 4;

// part_o : [B][Hq][nsplit][D] fp32, un-normalised
// part_ml: [B][Hq][nsplit][2] fp32 (running max, running sum)
// FINAL (one split): the block holds the whole row and writes the normalised output itself.
template <int D, bool FINAL = false>
__global__ __launch_bounds__(AD_WAVES * 64) void attn_decode_split_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k_cache,
    const bf16_t* __restrict__ v_cache, float* __restrict__ part_o, float* __restrict__ part_ml,
    const int* __restrict__ pos, int Hq, int Hkv, int l_max, float scale, bf16_t* __restrict__ out) {
  constexpr int LPK = D / 8;       // lanes per key row
  constexpr int KPW = 64 / LPK;    // keys per wave-instruction
  const int h = blockIdx.x, split = blockIdx.y, b = blockIdx.z, nsplit = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int sub = lane / LPK, dl = lane % LPK;
  // the current token's K/V were appended by the QKV kernel; never read past the cache
  const int len = min(max(pos[b] + 1, 1), l_max);
  const int hk = h / (Hq / Hkv);

  // this split's key range, in units of KPW-key groups
  const int ngroups = (len + KPW - 1) / KPW;
  const int gper = (ngroups + nsplit - 1) / nsplit;
  const int g0 = split * gper;
  const int g1 = min(g0 + gper, ngroups);

  // q stays packed: the score is four v_dot2 per lane and one multiply by the scale
  const u32x4_t qv = load16(q + (static_cast<size_t>(b) * Hq + h) * D + dl * 8);
  const bf16_t* kb = k_cache + (static_cast<size_t>(b) * Hkv + hk) * l_max * D + dl * 8;
  const bf16_t* vb = v_cache + (static_cast<size_t>(b) * Hkv + hk) * l_max * D + dl * 8;

  float m = -INFINITY, l = 0.f;
  float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};

  constexpr int U = 4;  // groups in flight per wave: 8 loads of 1 KiB
  for (int g = g0 + wid * U; g < g1; g += AD_WAVES * U) {
    u32x4_t kv[U], vv[U];
    int key[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      key[u] = (g + u) * KPW + sub;
      const bool ok = (g + u) < g1 && key[u] < len;
      const int kk = ok ? key[u] : 0;
      kv[u] = load16(kb + static_cast<size_t>(kk) * D);
      vv[u] = load16(vb + static_cast<size_t>(kk) * D);
      if (!ok) key[u] = -1;
    }
    // scores of the U keys first, then ONE rescale of the running state by the new maximum
    // (per key it is eight multiplies more), then the U weighted value rows
    float sc[U];
    float mn = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sc[u] = group_sum<LPK>(dot8_bf16(qv, kv[u], 0.f)) * scale;   // the LPK lanes of one key row
      if (key[u] >= 0) mn = fmaxf(mn, sc[u]);
    }
    const float a = (mn == -INFINITY) ? 1.f : __expf(m - mn);   // m == -inf: state is all zeros
    l *= a;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] *= a;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float p = key[u] >= 0 ? __expf(sc[u] - mn) : 0.f;
      l += p;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[2 * i] += p * bf16lo(vv[u][i]);
        o[2 * i + 1] += p * bf16hi(vv[u][i]);
      }
    }
    m = mn;
  }

  // Every 16-lane (8-lane for D = 64) group of every wave holds a partial (m, l, o): all of them
  // go to LDS and are merged in one pass.  Merging the groups of a wave with cross-lane
  // exchanges first costs 20 ds_bpermute per wave on the CU's shared LDS pipe.
  constexpr int NP = AD_WAVES * KPW;
  __shared__ __attribute__((aligned(16))) float s_o[NP][D];
  __shared__ float s_m[NP], s_l[NP];
  {
    const int pidx = wid * KPW + sub;
    *reinterpret_cast<f32x4_t*>(&s_o[pidx][dl * 8]) = f32x4_t{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4_t*>(&s_o[pidx][dl * 8 + 4]) = f32x4_t{o[4], o[5], o[6], o[7]};
    if (dl == 0) { s_m[pidx] = m; s_l[pidx] = l; }
  }
  __syncthreads();
  if (tid < D) {
    float mm = s_m[0];
#pragma unroll
    for (int w = 1; w < NP; ++w) mm = fmaxf(mm, s_m[w]);
    float ll = 0.f, oo = 0.f;
#pragma unroll
    for (int w = 0; w < NP; ++w) {
      const float c = (s_m[w] == -INFINITY) ? 0.f : __expf(s_m[w] - mm);
      ll += s_l[w] * c;
      oo += s_o[w][tid] * c;
    }
    if constexpr (FINAL) {
      out[(static_cast<size_t>(b) * Hq + h) * D + tid] = f32_to_bf16(oo / ll);   // len >= 1: ll > 0
    } else {
      const size_t pi = (static_cast<size_t>(b) * Hq + h) * nsplit + split;
      part_o[pi * D + tid] = oo;
      if (tid == 0) { part_ml[pi * 2] = mm; part_ml[pi * 2 + 1] = ll; }
    }
  }
}

// out[b][h*D + d] = sum_s w_s o_s[d] / sum_s w_s l_s,  w_s = exp(m_s - max m)
template <int D>
__global__ __launch_bounds__(D) void attn_decode_combine_kernel(const float* __restrict__ part_o,
                                                                const float* __restrict__ part_ml,
                                                                bf16_t* __restrict__ out, int Hq,
                                                                int nsplit) {
  const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  const size_t base = (static_cast<size_t>(b) * Hq + h) * nsplit;
  constexpr int MAXS = 16;
  // issue every load before the first use: the kernel is a handful of dependent
  // L2 round trips, so they must overlap
  float ms[MAXS], ls[MAXS], os[MAXS];
  // (splits past nsplit re-read split 0 and are masked afterwards: a select between a
  // load and a constant would put every load behind its own branch)
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const size_t slot = base + (s < nsplit ? s : 0);
    ms[s] = part_ml[slot * 2];
    ls[s] = part_ml[slot * 2 + 1];
    os[s] = part_o[slot * D + d];
  }
  float mm = -INFINITY;
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    ms[s] = s < nsplit ? ms[s] : -INFINITY;
    mm = fmaxf(mm, ms[s]);
  }
  float ll = 0.f, oo = 0.f;
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const float c = (ms[s] == -INFINITY) ? 0.f : __expf(ms[s] - mm);
    ll += ls[s] * c;
    oo += os[s] * c;
  }
  out[(static_cast<size_t>(b) * Hq + h) * D + d] = f32_to_bf16(oo / ll);
}

void launch_attn_decode(const bf16_t* q, const bf16_t* k_cache, const bf16_t* v_cache,
                        bf16_t* out, float* part_o, float* part_ml, const int* pos, int B, int Hq,
                        int Hkv, int D, int l_max, int nsplit, float scale, hipStream_t st) {
  dim3 grid(Hq, nsplit, B), block(AD_WAVES * 64);
  if (nsplit == 1) {   // many streams: no partial results, no second launch
    if (D == 128)
      hipLaunchKernelGGL((attn_decode_split_kernel<128, true>), grid, block, 0, st, q, k_cache, v_cache,
                         part_o, part_ml, pos, Hq, Hkv, l_max, scale, out);
    else
      hipLaunchKernelGGL((attn_decode_split_kernel<64, true>), grid, block, 0, st, q, k_cache, v_cache,
                         part_o, part_ml, pos, Hq, Hkv, l_max, scale, out);
    return;
  }
  if (D == 128) {
    hipLaunchKernelGGL((attn_decode_split_kernel<128, false>), grid, block, 0, st, q, k_cache, v_cache,
                       part_o, part_ml, pos, Hq, Hkv, l_max, scale, out);
    hipLaunchKernelGGL(attn_decode_combine_kernel<128>, dim3(Hq, B), dim3(128), 0, st, part_o,
                       part_ml, out, Hq, nsplit);
  } else {
    hipLaunchKernelGGL((attn_decode_split_kernel<64, false>), grid, block, 0, st, q, k_cache, v_cache,
                       part_o, part_ml, pos, Hq, Hkv, l_max, scale, out);
    hipLaunchKernelGGL(attn_decode_combine_kernel<64>, dim3(Hq, B), dim3(64), 0, st, part_o,
                       part_ml, out, Hq, nsplit);
  }
}

}  // namespace evgpt
