// Speculative-decoding verify/accept and the on-device sampler.
//
// verify_accept: the verifier has just run ONE forward over the G+1 tokens
// [last, d1..dG] and its lm_head GEMM left per-row argmax partials.  A single
// block finishes the row argmaxes, finds the longest draft prefix the verifier
// agrees with, appends the accepted tokens plus the verifier's own next token
// (the correction, or the bonus token when everything matched) to the output
// ring, handles an EOS inside the accepted run, rolls the KV cache back by
// moving the device-side length counter, and gathers the next embedding.  The
// host never sees the logits or the draft; it can read {n_accepted, next,
// n_emitted} later, or not at all.  (Behaviour of the reference's host-side
// loop: pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:569-637.)
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

constexpr int VA_MAX_ROWS = 64;

__global__ __launch_bounds__(256) void verify_accept_kernel(
    const float* __restrict__ pval, const int* __restrict__ pidx, int n_part,
    const int* __restrict__ draft, int G, int* __restrict__ out_tokens, int max_out,
    int* __restrict__ step, int* __restrict__ pos, int* __restrict__ done,
    int* __restrict__ cur_token, int* __restrict__ result, int eos,
    const bf16_t* __restrict__ embed, bf16_t* __restrict__ x, int H, int vocab,
    const int* __restrict__ limit, int* __restrict__ stats) {
  __shared__ int amax[VA_MAX_ROWS];
  __shared__ int next_s;
  // A host that replays whole draft/verify iterations without reading anything back passes
  // `limit`: once the sequence is done, or limit[0] tokens are out, a replay changes nothing
  // (its forward pass only wrote cache rows past the final position).  result[3..5] mirror
  // step / done / pos so one small read tells the host where the sequence stands.
  if (limit && (done[0] || step[0] >= limit[0])) {       // block-uniform
    if (threadIdx.x == 0) { result[0] = 0; result[2] = 0; result[3] = step[0]; result[4] = done[0]; result[5] = pos[0]; }
    return;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rows = G + 1;
  // one wave per row: reduce that row's partials (lowest index on ties)
  for (int r = wid; r < rows; r += 4) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < n_part; i += 64) {
      const float v = pval[static_cast<size_t>(r) * n_part + i];
      const int ix = pidx[static_cast<size_t>(r) * n_part + i];
      if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) amax[r] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int n_acc = 0;
    while (n_acc < G && draft[n_acc] == amax[n_acc]) ++n_acc;
    int st = step[0];
    int emitted = 0;
    bool hit_eos = false;
    // accepted draft tokens, then the verifier's token at the first mismatch / bonus
    for (int i = 0; i <= n_acc && !hit_eos; ++i) {
      const int tok = i < n_acc ? draft[i] : amax[n_acc];
      if (st < max_out) out_tokens[st] = tok;
      ++st; ++emitted;
      if (tok == eos) { hit_eos = true; n_acc = i < n_acc ? i + 1 : n_acc; }
    }
    const int next = hit_eos ? eos : amax[n_acc];
    // cache slots pos..pos+G were written by the verify forward; keep the last
    // token's slot plus one per accepted draft token
    pos[0] = pos[0] + (hit_eos ? emitted : n_acc + 1);
    step[0] = st;
    cur_token[0] = next;
    if (hit_eos) done[0] = 1;
    result[0] = n_acc; result[1] = next; result[2] = emitted;
    result[3] = st; result[4] = hit_eos ? 1 : done[0]; result[5] = pos[0];
    if (stats) { stats[0] += G; stats[1] += n_acc; stats[2] += 1; }
    next_s = next;
  }
  __syncthreads();
  int tok = next_s;
  if (tok < 0 || tok >= vocab) tok = 0;
  for (int c = threadIdx.x; c < (H >> 3); c += 256)
    *reinterpret_cast<u32x4_t*>(x + c * 8) = load16(embed + static_cast<size_t>(tok) * H + c * 8);
}

void launch_verify_accept(const float* pval, const int* pidx, int n_part, const int* draft, int G,
                          int* out_tokens, int max_out, int* step, int* pos, int* done,
                          int* cur_token, int* result, int eos, const bf16_t* embed, bf16_t* x,
                          int H, int vocab, hipStream_t st, const int* limit, int* stats) {
  hipLaunchKernelGGL(verify_accept_kernel, dim3(1), dim3(256), 0, st, pval, pidx, n_part, draft, G,
                     out_tokens, max_out, step, pos, done, cur_token, result, eos, embed, x, H, vocab,
                     limit, stats);
}

// ------------------------------------------------- token-level SD: drafter re-sync
// Start of one draft/verify round of token-level speculative decoding between two models that
// share a vocabulary (reference: feasible/naive_speculative_decoding.py:7-60, which re-runs both
// models over the whole sequence instead).  The verifier's counters say where the sequence
// stands; the drafter's cache is valid up to the verifier's position minus two at worst (after a
// round whose drafts were all accepted it has not ingested the last of them).  So every round
// the drafter begins with a TWO-row pass over [token before the current one, current token] at
// positions [pos - 1, pos]: row 0 re-writes a slot that is usually already right, row 1 yields
// the first draft.  This kernel stages those two embedding rows and points the drafter's
// counters at the verifier's position (plus the difference of the two prompts' lengths); no
// host read, no conditional catch-up pass.
__global__ __launch_bounds__(256) void naive_sd_sync_kernel(
    const int* __restrict__ t_pos, const int* __restrict__ t_step, const int* __restrict__ t_cur,
    const int* __restrict__ t_out, int max_out, int offset, const bf16_t* __restrict__ embed, int vocab,
    const bf16_t* __restrict__ prompt_last, bf16_t* __restrict__ rows, int* __restrict__ d_pos0,
    int* __restrict__ d_pos, int* __restrict__ d_step, int* __restrict__ d_done, int H, int l_max) {
  const int r = blockIdx.x;   // 0: the token before the current one, 1: the current token
  const int st = t_step[0];
  const bf16_t* src;
  if (r == 0 && (st < 2 || st - 2 >= max_out)) {
    src = prompt_last;          // nothing but the first token is out: the slot before it is the prompt's last row
  } else {
    int tok = r == 0 ? t_out[st - 2] : t_cur[0];
    if (tok < 0 || tok >= vocab) tok = 0;
    src = embed + static_cast<size_t>(tok) * H;
  }
  for (int c = threadIdx.x; c < (H >> 3); c += 256)
    *reinterpret_cast<u32x4_t*>(rows + static_cast<size_t>(r) * H + c * 8) = load16(src + c * 8);
  if (r == 0 && threadIdx.x == 0) {
    int p = t_pos[0] + offset;                 // the drafter's slot of the current token
    p = p < 1 ? 1 : (p > l_max - 1 ? l_max - 1 : p);
    d_pos0[0] = p - 1;
    d_pos[0] = p;
    d_step[0] = 0;
    d_done[0] = 0;
  }
}

void launch_naive_sd_sync(const int* t_pos, const int* t_step, const int* t_cur, const int* t_out, int max_out,
                          int offset, const bf16_t* embed, int vocab, const bf16_t* prompt_last, bf16_t* rows,
                          int* d_pos0, int* d_pos, int* d_step, int* d_done, int H, int l_max, hipStream_t st) {
  hipLaunchKernelGGL(naive_sd_sync_kernel, dim3(2), dim3(256), 0, st, t_pos, t_step, t_cur, t_out, max_out, offset,
                     embed, vocab, prompt_last, rows, d_pos0, d_pos, d_step, d_done, H, l_max);
}

// ---------------------------------------------------------------- sampler
// Temperature + top-k + nucleus (top-p) sampling of one token per row, entirely
// on the device: one 1024-thread block per row keeps the row's probabilities in
// registers, finds the k-th largest probability exactly by bisecting on its bit
// pattern (positive floats order like their bits; ties at the k-th value are all
// kept), then the probability threshold whose tail mass reaches top_p of what is
// left by bisection (no sort), then walks the kept set with a counter-based
// uniform draw.  The random stream is a function of (seed, step[b], row), so replaying
// a captured graph still draws fresh numbers as the device-side step advances.
__device__ __forceinline__ unsigned int mix32(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return static_cast<unsigned int>((z ^ (z >> 31)) >> 32);
}

constexpr int SP_THREADS = 1024;
constexpr int SP_MAX_PER_THREAD = 40;  // vocab up to 40960

__global__ __launch_bounds__(SP_THREADS) void sample_top_p_kernel(
    const float* __restrict__ logits, int* __restrict__ out, int V, float inv_temp, int top_k, float top_p,
    unsigned long long seed, const int* __restrict__ step) {
  __shared__ float red[16];
  __shared__ float scan[SP_THREADS];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = logits + static_cast<size_t>(b) * V;
  float pr[SP_MAX_PER_THREAD];
  // thread t owns the contiguous range [t*per, (t+1)*per) so a prefix scan over
  // threads walks the vocabulary in index order
  const int per = (V + SP_THREADS - 1) / SP_THREADS;
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < SP_MAX_PER_THREAD; ++i) {
    const int idx = tid * per + i;
    pr[i] = (i < per && idx < V) ? row[idx] * inv_temp : -INFINITY;
    mx = fmaxf(mx, pr[i]);
  }
  mx = wave_max(mx);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = red[0];
  for (int i = 1; i < SP_THREADS / 64; ++i) mx = fmaxf(mx, red[i]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < SP_MAX_PER_THREAD; ++i) { pr[i] = __expf(pr[i] - mx); sum += pr[i]; }
  float total = block_sum(sum, red);
  if (top_k > 0 && top_k < V) {
    // largest t with count(bits(p) >= t) >= k; p in [0, 1] so t in [0, bits(1.0f)]
    unsigned int lo = 0u, hi = 0x3F800000u + 1u;  // invariant: count(>= lo) >= k, count(>= hi) < k
    while (hi - lo > 1u) {
      const unsigned int mid = lo + ((hi - lo) >> 1);
      float c = 0.f;
#pragma unroll
      for (int i = 0; i < SP_MAX_PER_THREAD; ++i) c += __float_as_uint(pr[i]) >= mid ? 1.f : 0.f;
      c = block_sum(c, red);
      if (c >= static_cast<float>(top_k)) lo = mid; else hi = mid;
    }
    const float kth = __uint_as_float(lo);
    sum = 0.f;
#pragma unroll
    for (int i = 0; i < SP_MAX_PER_THREAD; ++i) { if (pr[i] < kth) pr[i] = 0.f; sum += pr[i]; }
    total = block_sum(sum, red);
  }
  // bisection on the threshold: keep = {p >= thr}; want the largest thr whose
  // kept mass still reaches top_p * total
  float thr = 0.f;
  if (top_p < 1.f) {
    float lo = 0.f, hi = 1.f;  // p of the max element is exp(0) = 1
    const float want = top_p * total;
    for (int it = 0; it < 24; ++it) {
      const float mid = 0.5f * (lo + hi);
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < SP_MAX_PER_THREAD; ++i) m += pr[i] >= mid ? pr[i] : 0.f;
      m = block_sum(m, red);
      if (m >= want) lo = mid; else hi = mid;
    }
    thr = lo;
  }
  float kept = 0.f;
#pragma unroll
  for (int i = 0; i < SP_MAX_PER_THREAD; ++i) { if (pr[i] < thr) pr[i] = 0.f; kept += pr[i]; }
  // inclusive scan of per-thread kept mass
  scan[tid] = kept;
  __syncthreads();
  for (int off = 1; off < SP_THREADS; off <<= 1) {
    const float v = tid >= off ? scan[tid - off] : 0.f;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  const float mass = scan[SP_THREADS - 1];
  const unsigned int rnd = mix32(seed ^ (static_cast<unsigned long long>(step[b]) << 20) ^
                                 (static_cast<unsigned long long>(b) << 52));
  // u in [0, mass): exactly one thread's [prefix_before, prefix_after) holds it
  const float u = static_cast<float>(rnd >> 8) * (1.f / 16777216.f) * mass;
  const float after = scan[tid];
  const float before = tid > 0 ? scan[tid - 1] : 0.f;
  if (kept > 0.f && u >= before && u < after) {
    float run = before;
    int pick = -1;
#pragma unroll
    for (int i = 0; i < SP_MAX_PER_THREAD; ++i) {
      if (pr[i] > 0.f && (pick < 0 || u >= run)) pick = tid * per + i;  // last kept index with start <= u
      run += pr[i];
    }
    out[b] = pick;
  }
}

void launch_sample_top_p(const float* logits, int* out, int B, int V, float temperature, int top_k,
                         float top_p, unsigned long long seed, const int* step, hipStream_t st) {
  hipLaunchKernelGGL(sample_top_p_kernel, dim3(B), dim3(SP_THREADS), 0, st, logits, out, V,
                     1.f / temperature, top_k, top_p, seed, step);
}

}  // namespace evgpt
