// torch op registration for the gfx950 kernels (namespace `evgpt::`).
//
// Two kinds of ops:
//   * functional ops (rmsnorm, linear, attention, ...) that allocate their
//     output — used by the nn.Module path;
//   * in-place "engine" ops that read and write caller-owned static buffers and
//     take device-resident positions — used by the serving engine and safe to
//     capture into a hipGraph (no allocation, no host sync).
// Host-side shape checks live here so a kernel is never launched on operands
// that do not match what its indexing assumes.
#include <algorithm>
#include <cmath>
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "evgpt_kernels.h"

namespace {

using at::Tensor;
using evgpt::bf16_t;

hipStream_t cur_stream() { return c10::hip::getCurrentHIPStream().stream(); }

const bf16_t* bptr(const Tensor& t) { return reinterpret_cast<const bf16_t*>(t.data_ptr()); }
bf16_t* bptr_mut(const Tensor& t) { return reinterpret_cast<bf16_t*>(t.data_ptr()); }
const bf16_t* bptr_opt(const c10::optional<Tensor>& t) {
  return t.has_value() && t->defined() ? bptr(*t) : nullptr;
}

void check_bf16(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must be a GPU tensor");
  TORCH_CHECK(t.scalar_type() == at::kBFloat16, name, " must be bfloat16");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
  TORCH_CHECK((reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0, name, " must be 16-byte aligned");
}
void check_i32(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kInt && t.is_contiguous(), name,
              " must be a contiguous int32 GPU tensor");
}
void check_f32(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(), name,
              " must be a contiguous float32 GPU tensor");
}

// a [rows, cols] bf16 matrix whose rows may be strided (a column block of a wider matrix)
void check_rows(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kBFloat16 && t.dim() == 2, name, ": 2-D bf16 GPU tensor");
  TORCH_CHECK(t.stride(1) == 1 && t.stride(0) >= t.size(1) && t.stride(0) % 8 == 0 &&
              (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0,
              name, ": unit column stride, row stride a multiple of 8, 16-byte aligned");
  TORCH_CHECK(t.size(0) * t.stride(0) < (1ll << 30), name, ": addressed with 32-bit byte offsets (< 2 GiB)");
}

// ------------------------------------------------------------------ norms
Tensor rmsnorm(const Tensor& x, const Tensor& w, double eps) {
  check_bf16(x, "x"); check_bf16(w, "weight");
  const int64_t H = x.size(-1);
  TORCH_CHECK(H % 8 == 0 && w.numel() == H, "rmsnorm: hidden size must be a multiple of 8");
  Tensor y = at::empty_like(x);
  const int64_t rows = x.numel() / H;
  if (rows > 0)
    evgpt::launch_rmsnorm(bptr(x), nullptr, bptr(w), bptr_mut(y), nullptr, (int)rows, (int)H,
                          (float)eps, cur_stream());
  return y;
}

std::tuple<Tensor, Tensor> add_rmsnorm(const Tensor& x, const Tensor& res, const Tensor& w,
                                       double eps) {
  check_bf16(x, "x"); check_bf16(res, "residual"); check_bf16(w, "weight");
  TORCH_CHECK(x.sizes() == res.sizes(), "add_rmsnorm: shape mismatch");
  const int64_t H = x.size(-1);
  TORCH_CHECK(H % 8 == 0 && w.numel() == H);
  Tensor y = at::empty_like(x), r = at::empty_like(x);
  const int64_t rows = x.numel() / H;
  if (rows > 0)
    evgpt::launch_rmsnorm(bptr(x), bptr(res), bptr(w), bptr_mut(y), bptr_mut(r), (int)rows, (int)H,
                          (float)eps, cur_stream());
  return {y, r};
}

Tensor layernorm(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b, double eps, int64_t act) {
  check_bf16(x, "x"); check_bf16(w, "weight");
  if (b.has_value()) check_bf16(*b, "bias");
  const int64_t H = x.size(-1);
  TORCH_CHECK(H % 8 == 0 && w.numel() == H);
  Tensor y = at::empty_like(x);
  const int64_t rows = x.numel() / H;
  if (rows > 0)
    evgpt::launch_layernorm(bptr(x), bptr(w), bptr_opt(b), bptr_mut(y), (int)rows, (int)H,
                            (float)eps, cur_stream(), (int)act);
  return y;
}

// ------------------------------------------------------------------ linear
// y = act(x @ w^T + bias) + res.  rows <= 4 -> weight-streaming GEMV, else MFMA GEMM.
Tensor linear(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& bias,
              const c10::optional<Tensor>& res, int64_t act, bool out_f32) {
  check_bf16(x, "x"); check_bf16(w, "weight");
  TORCH_CHECK(w.dim() == 2, "weight must be [N, K]");
  const int64_t K = x.size(-1), N = w.size(0);
  TORCH_CHECK(w.size(1) == K, "linear: K mismatch (", w.size(1), " vs ", K, ")");
  TORCH_CHECK(K % 8 == 0, "linear: K must be a multiple of 8");
  const int64_t M = x.numel() / K;
  TORCH_CHECK(M * K < (1ll << 30) && N * K < (1ll << 30), "linear: operands are addressed with 32-bit byte offsets (< 2 GiB each)");
  if (bias.has_value()) { check_bf16(*bias, "bias"); TORCH_CHECK(bias->numel() == N); }
  auto out_sizes = x.sizes().vec();
  out_sizes.back() = N;
  if (res.has_value()) {
    check_bf16(*res, "residual");
    TORCH_CHECK(res->numel() == M * N, "linear: residual shape mismatch");
  }
  Tensor y = at::empty(out_sizes, x.options().dtype(out_f32 ? at::kFloat : at::kBFloat16));
  if (M == 0) return y;
  auto st = cur_stream();
  const bool gemv_ok = M <= evgpt::gemv_max_batch() && !out_f32 && N % 4 == 0 &&
                       (int64_t)M * K * 2 <= 96 * 1024;
  if (gemv_ok) {
    evgpt::launch_gemv(bptr(w), bptr(x), bptr_mut(y), bptr_opt(bias), bptr_opt(res), nullptr, 0.f,
                       (int)act, (int)M, (int)N, (int)K, st);
  } else {
    const int sk = evgpt::gemm_splitk((int)M, (int)N, (int)K);
    Tensor ws;
    if (sk > 1) {
      ws = at::empty({sk * M * N}, x.options().dtype(at::kFloat));
    } else {
      const int64_t tail = evgpt::gemm_tail_ws_floats((int)M, (int)N, (int)K);
      if (tail > 0) ws = at::empty({tail}, x.options().dtype(at::kFloat));
    }
    evgpt::launch_gemm_bf16(bptr(x), bptr(w), out_f32 ? nullptr : bptr_mut(y),
                            out_f32 ? y.data_ptr<float>() : nullptr, bptr_opt(bias),
                            bptr_opt(res), (int)M, (int)N, (int)K, (int)K, (int)K, (int)N, (int)act,
                            nullptr, nullptr, ws.defined() ? ws.data_ptr<float>() : nullptr, sk, st);
  }
  return y;
}

// h [.., I] = silu(x · Wgate^T) * (x · Wup^T), w_gate_up = [gate rows | up rows] ([2I, K]); more than 32 rows
// (fewer take the weight-streaming / split-K paths, whose consumers apply the activation)
Tensor linear_swiglu(const Tensor& x, const Tensor& w) {
  check_bf16(x, "x"); check_bf16(w, "w_gate_up");
  const int64_t K = x.size(-1), I = w.size(0) / 2, M = K ? x.numel() / K : 0;
  TORCH_CHECK(w.dim() == 2 && w.size(1) == K && w.size(0) % 2 == 0, "linear_swiglu: w_gate_up is [2I, K]");
  TORCH_CHECK(K % 64 == 0 && I % 32 == 0, "linear_swiglu: K % 64 == 0 and I % 32 == 0");
  TORCH_CHECK(2 * I * K * 2 < (1ll << 31) && M * K * 2 < (1ll << 31), "linear_swiglu: operands over 2 GiB");
  auto out_sizes = x.sizes().vec();
  out_sizes.back() = I;
  Tensor h = at::empty(out_sizes, x.options());
  if (M > 0)
    evgpt::launch_gemm_glu(bptr(x), bptr(w), bptr_mut(h), (int)M, (int)I, (int)K, (int)K, (int)K, (int)I, cur_stream());
  return h;
}

// C [M,N] = opA(a) · opW(w)^T on the MFMA GEMM with either operand K-strided:
//   a_t = false: a is [M,K]      a_t = true: a is [Kr,M] (the contraction runs over a's rows)
//   w_t = false: w is [N,K]      w_t = true: w is [Kr,N]
// These are the layouts of a linear layer's backward: dX = dY · W is (a = dY, w = W, w_t) and
// dW = dY^T · X is (a = dY, w = X, both transposed).  K is the contraction rounded up to 64; an
// operand may end up to 63 before it (K-strided: any number of rows Kr <= K; K-contiguous: a
// multiple of 8 columns): the staging loads return zeros past an operand's real end.
// `out`: write the product into this tensor (same dtype and element count as the result) instead of a
// new one; it may be `res` itself, which makes the call an accumulation (each element is read and
// written by the same lane).
Tensor linear_t(const Tensor& a, const Tensor& w, bool a_t, bool w_t, const c10::optional<Tensor>& bias,
                const c10::optional<Tensor>& res, int64_t act, bool out_f32, int64_t K,
                const c10::optional<Tensor>& out) {
  TORCH_CHECK(a.dim() == 2 && w.dim() == 2 && (a_t || w_t), "linear_t: 2-D operands, at least one K-strided");
  check_rows(a, "a"); check_rows(w, "w");     // a K-strided operand may be a column block of a wider matrix
  TORCH_CHECK((a_t || a.is_contiguous()) && (w_t || w.is_contiguous()), "linear_t: a K-contiguous operand must be contiguous");
  TORCH_CHECK(K % 64 == 0 && K >= 64, "linear_t: padded contraction length must be a multiple of 64");
  const int64_t M = a_t ? a.size(1) : a.size(0), N = w_t ? w.size(1) : w.size(0);
  int64_t ka_rows = K, kw_rows = K;
  if (a_t) {
    TORCH_CHECK(M % 8 == 0 && M >= 8, "linear_t: K-strided A needs M % 8 == 0");
    ka_rows = a.size(0);
  } else TORCH_CHECK(a.size(1) % 8 == 0 && a.size(1) >= 8, "linear_t: K-contiguous A needs a multiple of 8 columns");
  if (w_t) {
    TORCH_CHECK(N % 8 == 0 && N >= 8, "linear_t: K-strided W needs N % 8 == 0");
    kw_rows = w.size(0);
  } else TORCH_CHECK(w.size(1) % 8 == 0 && w.size(1) >= 8, "linear_t: K-contiguous W needs a multiple of 8 columns");
  // the contraction lengths the two operands really have (a K-contiguous one: its columns)
  const int64_t ka = a_t ? ka_rows : a.size(1), kw = w_t ? kw_rows : w.size(1);
  TORCH_CHECK(ka >= 1 && kw >= 1 && ka <= K && kw <= K && K - std::min(ka, kw) < 64,
              "linear_t: K is the contraction rounded up to 64");
  // Both kinds of operand end where their data ends: K-strided ones at their last row (the staging
  // loads are range-checked against the rows that exist), K-contiguous ones at their last column
  // (the last K-tile's chunks past it are given out-of-range offsets).  Either way the kernel sees
  // zeros there, so no operand needs padding.
  if (bias.has_value()) { check_bf16(*bias, "bias"); TORCH_CHECK(bias->numel() == N); }
  if (res.has_value()) { check_bf16(*res, "residual"); TORCH_CHECK(res->numel() == M * N); }
  Tensor y;
  if (out.has_value()) {
    y = *out;
    TORCH_CHECK(y.is_cuda() && y.is_contiguous() && y.numel() == M * N &&
                y.scalar_type() == (out_f32 ? at::kFloat : at::kBFloat16), "linear_t: out must be a contiguous [M, N] "
                "tensor of the result's dtype");
  } else {
    y = at::empty({M, N}, a.options().dtype(out_f32 ? at::kFloat : at::kBFloat16));
  }
  if (M == 0 || N == 0) return y;
  const int sk = evgpt::gemm_splitk_t((int)M, (int)N, (int)K);
  Tensor ws;
  if (sk > 1) ws = at::empty({sk * M * N}, a.options().dtype(at::kFloat));
  evgpt::launch_gemm_bf16_t(bptr(a), bptr(w), out_f32 ? nullptr : bptr_mut(y), out_f32 ? y.data_ptr<float>() : nullptr,
                            bptr_opt(bias), bptr_opt(res), (int)M, (int)N, (int)K, (int)a.stride(0), (int)w.stride(0),
                            (int)N, (int)act, a_t, w_t, (int)ka_rows, (int)kw_rows, cur_stream(),
                            sk > 1 ? ws.data_ptr<float>() : nullptr, sk);
  return y;
}

// t [.., R] = alpha * x [.., K] · w [R, K]^T for a low-rank adapter's R <= 16 rows: one launch, no split-K pass.
// x: contiguous, or 2-D with strided rows.  drop_p > 0: x is masked while it is loaded (keep mask of
// (seed, element index), 1 / (1 - p) folded into alpha); lowrank_mask reproduces m (.) x for the backward.
Tensor lowrank_out(const Tensor& x, const Tensor& w, double alpha, double drop_p, int64_t seed) {
  check_bf16(w, "w");
  TORCH_CHECK(w.dim() == 2 && x.dim() >= 1, "lowrank_out: w is [R, K]");
  const bool strided = x.dim() == 2 && !x.is_contiguous();
  if (strided) check_rows(x, "x"); else check_bf16(x, "x");
  const int64_t K = x.size(-1), R = w.size(0), M = K ? x.numel() / K : 0;
  TORCH_CHECK(w.size(1) == K && K % 8 == 0 && K >= 8, "lowrank_out: K mismatch or K % 8 != 0");
  TORCH_CHECK(R >= 1 && R <= 16, "lowrank_out: 1 <= R <= 16");
  TORCH_CHECK(M < (1ll << 31) / 16 && x.numel() < (1ll << 40), "lowrank_out: too many rows");
  TORCH_CHECK(drop_p >= 0 && drop_p < 1, "lowrank_out: 0 <= drop_p < 1");
  auto sizes = x.sizes().vec();
  sizes.back() = R;
  Tensor y = at::empty(sizes, x.options());
  if (M > 0)
    evgpt::launch_lowrank_out(bptr(x), bptr(w), bptr_mut(y), (int)M, (int)R, (int)K, strided ? x.stride(0) : K,
                              (float)alpha, (float)drop_p, (unsigned long long)seed, cur_stream());
  return y;
}

// m (.) x with lowrank_out's keep mask (no rescaling); contiguous, a multiple of 8 elements
Tensor lowrank_mask(const Tensor& x, double drop_p, int64_t seed) {
  check_bf16(x, "x");
  TORCH_CHECK(x.numel() % 8 == 0 && drop_p >= 0 && drop_p < 1, "lowrank_mask: a multiple of 8 elements");
  Tensor y = at::empty_like(x);
  evgpt::launch_lowrank_mask(bptr(x), bptr_mut(y), x.numel(), (float)drop_p, (unsigned long long)seed, false, cur_stream());
  return y;
}
// acc += m (.) x in place
void lowrank_mask_add_(const Tensor& acc, const Tensor& x, double drop_p, int64_t seed) {
  check_bf16(x, "x"); check_bf16(acc, "acc");
  TORCH_CHECK(x.numel() % 8 == 0 && acc.numel() == x.numel() && drop_p >= 0 && drop_p < 1, "lowrank_mask_add_: shapes");
  evgpt::launch_lowrank_mask(bptr(x), bptr_mut(acc), x.numel(), (float)drop_p, (unsigned long long)seed, true, cur_stream());
}

// out [M, N] (rows may be strided: a column block of a wider matrix) = x [M, K] · w [N, K]^T (+ out when
// accumulate); x's rows may be strided too.  The low-rank update written straight onto its columns of a
// grouped projection.
void linear_rows_(const Tensor& x, const Tensor& w, const Tensor& out, bool accumulate) {
  check_rows(x, "x"); check_rows(out, "out"); check_bf16(w, "weight");
  const int64_t M = x.size(0), K = x.size(1), N = w.size(0);
  TORCH_CHECK(w.dim() == 2 && w.size(1) == K && K % 8 == 0 && out.size(0) == M && out.size(1) == N && N % 4 == 0 &&
              out.stride(0) % 4 == 0, "linear_rows_: x [M, K], w [N, K], out [M, N], K % 8 == 0");
  TORCH_CHECK(N * K < (1ll << 30), "linear_rows_: weight < 2 GiB");
  if (M == 0) return;
  evgpt::launch_gemm_bf16(bptr(x), bptr(w), bptr_mut(out), nullptr, nullptr, accumulate ? bptr(out) : nullptr, (int)M,
                          (int)N, (int)K, (int)x.stride(0), (int)K, (int)out.stride(0), 0, nullptr, nullptr, nullptr, 1,
                          cur_stream());
}

// ---- few-row projections: GEMM to fp32 split-K slabs + a fused consumer ----
// partial [nslab, M, N] fp32 (nslab >= 1); the consumers below sum the slabs.
Tensor linear_partial(const Tensor& x, const Tensor& w, int64_t max_split) {
  check_bf16(x, "x"); check_bf16(w, "weight");
  const int64_t K = x.size(-1), N = w.size(0);
  TORCH_CHECK(w.dim() == 2 && w.size(1) == K, "linear_partial: K mismatch");
  TORCH_CHECK(K % 64 == 0 && N % 8 == 0, "linear_partial: K % 64 == 0 and N % 8 == 0");
  const int64_t M = x.numel() / K;
  TORCH_CHECK(M >= 1, "linear_partial: empty input");
  int sk = evgpt::gemm_splitk_slabs((int)M, (int)N, (int)K);
  if (max_split > 0 && sk > max_split) sk = (int)max_split;
  Tensor ws = at::empty({sk, M, N}, x.options().dtype(at::kFloat));
  const int used = evgpt::launch_gemm_partial(bptr(x), bptr(w), ws.data_ptr<float>(), (int)M, (int)N,
                                              (int)K, (int)K, (int)K, cur_stream(), (int)max_split);
  return used == sk ? ws : ws.narrow(0, 0, used);
}

// x_out = bf16(sum partial) + res; xn = RMSNorm(x_out) * norm_w when norm_w is given
// (else xn is an empty tensor).  x_out may be `res` itself (in place).
std::tuple<Tensor, Tensor> reduce_residual_norm(const Tensor& partial, const c10::optional<Tensor>& res,
                                                const c10::optional<Tensor>& norm_w, double eps,
                                                const c10::optional<Tensor>& out) {
  check_f32(partial, "partial");
  TORCH_CHECK(partial.dim() == 3, "partial must be [nslab, M, N]");
  const int64_t S = partial.size(0), M = partial.size(1), N = partial.size(2);
  TORCH_CHECK(N % 8 == 0, "reduce_residual_norm: N % 8 == 0");
  auto bf = partial.options().dtype(at::kBFloat16);
  Tensor x = out.has_value() ? *out : at::empty({M, N}, bf);
  check_bf16(x, "out");
  TORCH_CHECK(x.numel() == M * N, "reduce_residual_norm: out shape");
  if (res.has_value()) { check_bf16(*res, "res"); TORCH_CHECK(res->numel() == M * N); }
  Tensor xn;
  if (norm_w.has_value()) {
    check_bf16(*norm_w, "norm_w");
    TORCH_CHECK(norm_w->numel() == N);
    xn = at::empty({M, N}, bf);
  } else {
    xn = at::empty({0}, bf);
  }
  evgpt::launch_reduce_residual_norm(partial.data_ptr<float>(), (int)S, (int)M, (int)N, bptr_opt(res),
                                     bptr_mut(x), bptr_opt(norm_w),
                                     norm_w.has_value() ? bptr_mut(xn) : nullptr, (float)eps,
                                     cur_stream());
  return {x, xn};
}

// x = bf16(sum partial + bias) + res; y = LayerNorm(x) * ln_w + ln_b when ln_w is given (else an empty tensor).
// Rows of up to 2048 elements.
std::tuple<Tensor, Tensor> reduce_residual_layernorm(const Tensor& partial, const c10::optional<Tensor>& bias,
                                                     const c10::optional<Tensor>& res,
                                                     const c10::optional<Tensor>& ln_w,
                                                     const c10::optional<Tensor>& ln_b, double eps) {
  check_f32(partial, "partial");
  TORCH_CHECK(partial.dim() == 3, "partial must be [nslab, M, N]");
  const int64_t S = partial.size(0), M = partial.size(1), N = partial.size(2);
  TORCH_CHECK(N % 8 == 0 && N <= 2048 && M >= 1 && S >= 1, "reduce_residual_layernorm: N % 8 == 0, N <= 2048");
  auto bf = partial.options().dtype(at::kBFloat16);
  if (bias.has_value()) { check_bf16(*bias, "bias"); TORCH_CHECK(bias->numel() == N); }
  if (res.has_value()) { check_bf16(*res, "res"); TORCH_CHECK(res->numel() == M * N); }
  if (ln_w.has_value()) { check_bf16(*ln_w, "ln_w"); TORCH_CHECK(ln_w->numel() == N); }
  if (ln_b.has_value()) { check_bf16(*ln_b, "ln_b"); TORCH_CHECK(ln_b->numel() == N && ln_w.has_value()); }
  Tensor x = at::empty({M, N}, bf);
  Tensor y = ln_w.has_value() ? at::empty({M, N}, bf) : at::empty({0}, bf);
  const bool ok = evgpt::launch_reduce_residual_layernorm(
      partial.data_ptr<float>(), (int)S, (int)M, (int)N, bptr_opt(bias), bptr_opt(res), bptr_mut(x), bptr_opt(ln_w),
      bptr_opt(ln_b), ln_w.has_value() ? bptr_mut(y) : nullptr, (float)eps, cur_stream());
  TORCH_CHECK(ok, "reduce_residual_layernorm: row too wide");
  return {x, y};
}

Tensor reduce_swiglu(const Tensor& partial) {
  check_f32(partial, "partial");
  TORCH_CHECK(partial.dim() == 3 && partial.size(2) % 16 == 0, "partial must be [nslab, M, 2I]");
  const int64_t S = partial.size(0), M = partial.size(1), I = partial.size(2) / 2;
  Tensor y = at::empty({M, I}, partial.options().dtype(at::kBFloat16));
  evgpt::launch_reduce_swiglu(partial.data_ptr<float>(), (int)S, M, (int)I, bptr_mut(y), cur_stream());
  return y;
}

// partial [nslab, B*S, (Hq+2Hkv)*D] -> q_out [B,S,Hq*D] + cache rows at pos0[b]+s
void reduce_rope_kv(const Tensor& partial, const Tensor& q_out, const Tensor& k_cache,
                    const Tensor& v_cache, const Tensor& cos, const Tensor& sin, const Tensor& pos0,
                    int64_t B, int64_t S, int64_t Hq, int64_t Hkv, int64_t D) {
  check_f32(partial, "partial"); check_bf16(q_out, "q_out"); check_bf16(k_cache, "k_cache");
  check_bf16(v_cache, "v_cache"); check_f32(cos, "cos"); check_f32(sin, "sin");
  check_i32(pos0, "pos0");
  TORCH_CHECK(partial.dim() == 3 && partial.size(1) == B * S && partial.size(2) == (Hq + 2 * Hkv) * D,
              "reduce_rope_kv: partial must be [nslab, B*S, (Hq+2Hkv)*D]");
  TORCH_CHECK(D % 16 == 0 && q_out.numel() == B * S * Hq * D && pos0.numel() == B);
  TORCH_CHECK(k_cache.dim() == 4 && k_cache.size(0) >= B && k_cache.size(1) == Hkv &&
              k_cache.size(3) == D && v_cache.sizes() == k_cache.sizes() && k_cache.is_contiguous() &&
              v_cache.is_contiguous(), "reduce_rope_kv: cache must be contiguous [B,Hkv,Lmax,D]");
  TORCH_CHECK(cos.size(-1) == D / 2 && sin.size(-1) == D / 2 && cos.size(0) >= k_cache.size(2));
  evgpt::launch_reduce_rope_kv(partial.data_ptr<float>(), (int)partial.size(0), bptr_mut(q_out),
                               bptr_mut(k_cache), bptr_mut(v_cache), cos.data_ptr<float>(),
                               sin.data_ptr<float>(), pos0.data_ptr<int>(), (int)B, (int)S, (int)Hq,
                               (int)Hkv, (int)D, (int)k_cache.size(2), cur_stream());
}

// (cos [rows], mse [rows]) of two equally shaped tensors whose last dim is the row
std::tuple<Tensor, Tensor> row_metrics(const Tensor& a, const Tensor& b, double eps) {
  TORCH_CHECK(a.is_cuda() && b.is_cuda() && a.is_contiguous() && b.is_contiguous() &&
              a.sizes() == b.sizes() && a.scalar_type() == b.scalar_type(),
              "row_metrics: two contiguous GPU tensors of one shape and dtype");
  const int64_t H = a.size(-1), rows = H ? a.numel() / H : 0;
  auto out_sizes = a.sizes().vec();
  out_sizes.pop_back();
  Tensor c = at::empty(out_sizes, a.options().dtype(at::kFloat));
  Tensor m = at::empty(out_sizes, a.options().dtype(at::kFloat));
  if (rows == 0) return {c, m};
  if (a.scalar_type() == at::kFloat) {
    TORCH_CHECK(H % 4 == 0, "row_metrics: row length must be a multiple of 4 (fp32)");
    evgpt::launch_row_metrics_f32(a.data_ptr<float>(), b.data_ptr<float>(), c.data_ptr<float>(),
                                  m.data_ptr<float>(), rows, (int)H, (float)eps, cur_stream());
  } else if (a.scalar_type() == at::kBFloat16) {
    TORCH_CHECK(H % 8 == 0, "row_metrics: row length must be a multiple of 8 (bf16)");
    evgpt::launch_row_metrics_bf16(bptr(a), bptr(b), c.data_ptr<float>(), m.data_ptr<float>(), rows,
                                   (int)H, (float)eps, cur_stream());
  } else {
    TORCH_CHECK(false, "row_metrics: float32 or bfloat16 only");
  }
  return {c, m};
}

// [rows, 4] = (dot, |a|^2, |b|^2, |a-b|^2) per row of two equally shaped tensors
Tensor row_stats(const Tensor& a, const Tensor& b) {
  TORCH_CHECK(a.is_cuda() && b.is_cuda() && a.is_contiguous() && b.is_contiguous() &&
              a.sizes() == b.sizes() && a.scalar_type() == b.scalar_type(),
              "row_stats: two contiguous GPU tensors of one shape and dtype");
  const int64_t H = a.size(-1), rows = H ? a.numel() / H : 0;
  Tensor s = at::zeros({rows, 4}, a.options().dtype(at::kFloat));
  if (rows == 0) return s;
  if (a.scalar_type() == at::kFloat) {
    TORCH_CHECK(H % 4 == 0, "row_stats: row length must be a multiple of 4 (fp32)");
    evgpt::launch_row_metrics_f32(a.data_ptr<float>(), b.data_ptr<float>(), nullptr, nullptr, rows, (int)H, 0.f,
                                  cur_stream(), s.data_ptr<float>());
  } else if (a.scalar_type() == at::kBFloat16) {
    TORCH_CHECK(H % 8 == 0, "row_stats: row length must be a multiple of 8 (bf16)");
    evgpt::launch_row_metrics_bf16(bptr(a), bptr(b), nullptr, nullptr, rows, (int)H, 0.f, cur_stream(),
                                   s.data_ptr<float>());
  } else {
    TORCH_CHECK(false, "row_stats: float32 or bfloat16 only");
  }
  return s;
}

// grad[r] = A_r (p_r - t_r) + B_r t_r + C_r p_r with coef [rows, 3] fp32
Tensor row_affine_grad(const Tensor& p, const Tensor& t, const Tensor& coef) {
  check_f32(coef, "coef");
  TORCH_CHECK(p.is_cuda() && t.is_cuda() && p.is_contiguous() && t.is_contiguous() && p.sizes() == t.sizes() &&
              p.scalar_type() == t.scalar_type(), "row_affine_grad: two contiguous GPU tensors of one shape and dtype");
  const int64_t H = p.size(-1), rows = H ? p.numel() / H : 0;
  const bool bf = p.scalar_type() == at::kBFloat16;
  TORCH_CHECK(bf || p.scalar_type() == at::kFloat, "row_affine_grad: float32 or bfloat16 only");
  TORCH_CHECK(H % (bf ? 8 : 4) == 0 && coef.numel() == rows * 3, "row_affine_grad: row length / coefficient count");
  Tensor g = at::empty_like(p);
  if (rows) evgpt::launch_row_affine_grad(p.data_ptr(), t.data_ptr(), coef.data_ptr<float>(), g.data_ptr(), rows, (int)H, bf,
                                          cur_stream());
  return g;
}

Tensor swiglu(const Tensor& gate_up) {
  check_bf16(gate_up, "gate_up");
  const int64_t I2 = gate_up.size(-1);
  TORCH_CHECK(I2 % 16 == 0, "swiglu: last dim must be a multiple of 16");
  auto sizes = gate_up.sizes().vec();
  sizes.back() = I2 / 2;
  Tensor y = at::empty(sizes, gate_up.options());
  const int64_t M = gate_up.numel() / I2;
  if (M > 0) evgpt::launch_swiglu(bptr(gate_up), bptr_mut(y), M, (int)(I2 / 2), cur_stream());
  return y;
}

Tensor rope(const Tensor& x, const Tensor& positions, const Tensor& cos, const Tensor& sin) {
  check_bf16(x, "x"); check_f32(cos, "cos"); check_f32(sin, "sin");
  TORCH_CHECK(x.dim() == 4, "rope: x must be [B,S,H,D]");
  TORCH_CHECK(positions.is_cuda() && positions.scalar_type() == at::kLong &&
              positions.is_contiguous() && positions.numel() == x.size(0) * x.size(1),
              "rope: positions must be a contiguous int64 [B,S] GPU tensor");
  const int64_t H = x.size(2), D = x.size(3);
  TORCH_CHECK(D % 16 == 0 && cos.size(-1) == D / 2 && sin.size(-1) == D / 2);
  Tensor y = at::empty_like(x);
  if (x.numel() > 0)
    evgpt::launch_rope(bptr(x), bptr_mut(y), reinterpret_cast<const long long*>(positions.data_ptr<int64_t>()), cos.data_ptr<float>(),
                       sin.data_ptr<float>(), x.size(0) * x.size(1), (int)H, (int)D, cur_stream());
  return y;
}

// q [B,S,Hq,D], k, v [B,S,Hkv,D] = split + RoPE of a packed projection [B,S,(Hq+2Hkv)*D]
std::vector<Tensor> rope_qkv_split(const Tensor& qkv, const Tensor& positions, const Tensor& cos, const Tensor& sin,
                                   int64_t Hq, int64_t Hkv, int64_t D) {
  check_bf16(qkv, "qkv"); check_f32(cos, "cos"); check_f32(sin, "sin");
  TORCH_CHECK(qkv.dim() == 3 && qkv.size(2) == (Hq + 2 * Hkv) * D, "rope_qkv_split: qkv must be [B,S,(Hq+2Hkv)*D]");
  const int64_t B = qkv.size(0), S = qkv.size(1);
  TORCH_CHECK(positions.is_cuda() && positions.scalar_type() == at::kLong && positions.is_contiguous() &&
              positions.numel() == B * S, "rope_qkv_split: positions must be a contiguous int64 [B,S] GPU tensor");
  TORCH_CHECK(D % 16 == 0 && cos.dim() == 2 && cos.size(1) == D / 2 && sin.sizes() == cos.sizes() && cos.is_contiguous() &&
              sin.is_contiguous(), "rope_qkv_split: cos / sin tables are [positions, D/2]");
  Tensor q = at::empty({B, S, Hq, D}, qkv.options()), k = at::empty({B, S, Hkv, D}, qkv.options()),
         v = at::empty({B, S, Hkv, D}, qkv.options());
  if (qkv.numel() > 0)
    evgpt::launch_rope_qkv(bptr(qkv), nullptr, bptr_mut(q), bptr_mut(k), bptr_mut(v),
                           reinterpret_cast<const long long*>(positions.data_ptr<int64_t>()), cos.data_ptr<float>(),
                           sin.data_ptr<float>(), B * S, (int)Hq, (int)Hkv, (int)D, (int)cos.size(0), false, cur_stream());
  return {q, k, v};
}

// the packed gradient [B,S,(Hq+2Hkv)*D] of rope_qkv_split's input from the gradients of its outputs
Tensor rope_qkv_merge(const Tensor& gq, const Tensor& gk, const Tensor& gv, const Tensor& positions, const Tensor& cos,
                      const Tensor& sin) {
  check_bf16(gq, "gq"); check_bf16(gk, "gk"); check_bf16(gv, "gv"); check_f32(cos, "cos"); check_f32(sin, "sin");
  TORCH_CHECK(gq.dim() == 4 && gk.dim() == 4 && gv.sizes() == gk.sizes() && gq.size(0) == gk.size(0) &&
              gq.size(1) == gk.size(1) && gq.size(3) == gk.size(3), "rope_qkv_merge: gq [B,S,Hq,D], gk / gv [B,S,Hkv,D]");
  const int64_t B = gq.size(0), S = gq.size(1), Hq = gq.size(2), Hkv = gk.size(2), D = gq.size(3);
  TORCH_CHECK(positions.is_cuda() && positions.scalar_type() == at::kLong && positions.is_contiguous() &&
              positions.numel() == B * S, "rope_qkv_merge: positions must be a contiguous int64 [B,S] GPU tensor");
  TORCH_CHECK(D % 16 == 0 && cos.dim() == 2 && cos.size(1) == D / 2 && sin.sizes() == cos.sizes() && cos.is_contiguous() &&
              sin.is_contiguous(), "rope_qkv_merge: cos / sin tables are [positions, D/2]");
  Tensor g = at::empty({B, S, (Hq + 2 * Hkv) * D}, gq.options());
  if (g.numel() > 0)
    evgpt::launch_rope_qkv(nullptr, bptr_mut(g), bptr_mut(gq), bptr_mut(gk), bptr_mut(gv),
                           reinterpret_cast<const long long*>(positions.data_ptr<int64_t>()), cos.data_ptr<float>(),
                           sin.data_ptr<float>(), B * S, (int)Hq, (int)Hkv, (int)D, (int)cos.size(0), true, cur_stream());
  return g;
}

// --------------------------------------------------------------- attention
void fill_strides_bshd(const Tensor& t, long long* s) {
  TORCH_CHECK(t.stride(3) == 1, "attention: head_dim must be contiguous");
  s[0] = t.stride(0); s[1] = t.stride(1); s[2] = t.stride(2);
}

// q [B,Sq,Hq,D], k/v [B,Sk,Hkv,D] (any strides with contiguous D)
Tensor attention(const Tensor& q, const Tensor& k, const Tensor& v, bool causal, double scale,
                 int64_t q_offset) {
  TORCH_CHECK(q.is_cuda() && q.scalar_type() == at::kBFloat16 && k.scalar_type() == at::kBFloat16 &&
              v.scalar_type() == at::kBFloat16, "attention: bf16 GPU tensors required");
  TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.dim() == 4);
  const int64_t B = q.size(0), Sq = q.size(1), Hq = q.size(2), D = q.size(3);
  const int64_t Sk = k.size(1), Hkv = k.size(2);
  TORCH_CHECK(D == 64 || D == 128, "attention: head_dim must be 64 or 128");
  TORCH_CHECK(k.size(0) == B && v.size(0) == B && v.size(1) == Sk && v.size(2) == Hkv &&
              k.size(3) == D && v.size(3) == D && Hq % Hkv == 0, "attention: shape mismatch");
  TORCH_CHECK(Sk >= 1 && Sq >= 1);
  Tensor o = at::empty({B, Sq, Hq, D}, q.options());
  long long s[12];
  fill_strides_bshd(q, s); fill_strides_bshd(k, s + 3); fill_strides_bshd(v, s + 6);
  fill_strides_bshd(o, s + 9);
  TORCH_CHECK(q.stride(1) % 8 == 0 && q.stride(2) % 8 == 0 && k.stride(1) % 8 == 0 &&
              k.stride(2) % 8 == 0 && v.stride(1) % 8 == 0 && v.stride(2) % 8 == 0,
              "attention: strides must keep 16-byte alignment");
  evgpt::launch_attn_prefill(bptr(q), bptr(k), bptr(v), bptr_mut(o), s, nullptr, (int)B, (int)Sq,
                             (int)Sk, (int)Hq, (int)Hkv, (int)D, causal ? 1 : 0, (int)q_offset, 0,
                             (float)scale, cur_stream());
  return o;
}

// Training forward: attention plus the per-row log-sum-exp the backward recomputes P from.
// q [B,Sq,Hq,D], k/v [B,Sk,Hkv,D] contiguous.  Returns (o [B,Sq,Hq,D], lse [B,Hq,Sq] fp32).
std::tuple<Tensor, Tensor> attention_fwd_lse(const Tensor& q, const Tensor& k, const Tensor& v, bool causal,
                                             double scale, int64_t q_offset) {
  check_bf16(q, "q"); check_bf16(k, "k"); check_bf16(v, "v");
  TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.sizes() == k.sizes());
  const int64_t B = q.size(0), Sq = q.size(1), Hq = q.size(2), D = q.size(3), Sk = k.size(1), Hkv = k.size(2);
  TORCH_CHECK((D == 64 || D == 128) && k.size(0) == B && k.size(3) == D && Hq % Hkv == 0 && Sq >= 1 && Sk >= 1,
              "attention_fwd_lse: shape mismatch");
  Tensor o = at::empty_like(q);
  Tensor lse = at::empty({B, Hq, Sq}, q.options().dtype(at::kFloat));
  long long s[12];
  fill_strides_bshd(q, s); fill_strides_bshd(k, s + 3); fill_strides_bshd(v, s + 6); fill_strides_bshd(o, s + 9);
  evgpt::launch_attn_prefill(bptr(q), bptr(k), bptr(v), bptr_mut(o), s, nullptr, (int)B, (int)Sq, (int)Sk, (int)Hq,
                             (int)Hkv, (int)D, causal ? 1 : 0, (int)q_offset, 0, (float)scale, cur_stream(), nullptr,
                             nullptr, 0, lse.data_ptr<float>());
  return {o, lse};
}

// (dq, dk, dv) of the attention above from the upstream gradient d_o
std::tuple<Tensor, Tensor, Tensor> attention_bwd(const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& o,
                                                 const Tensor& d_o, const Tensor& lse, bool causal, double scale,
                                                 int64_t q_offset) {
  check_bf16(q, "q"); check_bf16(k, "k"); check_bf16(v, "v"); check_bf16(o, "o"); check_bf16(d_o, "d_o");
  check_f32(lse, "lse");
  TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.sizes() == k.sizes() && o.sizes() == q.sizes() &&
              d_o.sizes() == q.sizes());
  const int64_t B = q.size(0), Sq = q.size(1), Hq = q.size(2), D = q.size(3), Sk = k.size(1), Hkv = k.size(2);
  TORCH_CHECK((D == 64 || D == 128) && k.size(0) == B && k.size(3) == D && Hq % Hkv == 0 && lse.numel() == B * Hq * Sq,
              "attention_bwd: shape mismatch");
  Tensor dq = at::empty_like(q), dk = at::empty_like(k), dv = at::empty_like(v);
  Tensor delta = at::empty({B, Hq, Sq}, q.options().dtype(at::kFloat));
  evgpt::launch_attn_bwd(bptr(q), bptr(k), bptr(v), bptr(o), bptr(d_o), lse.data_ptr<float>(), delta.data_ptr<float>(),
                         bptr_mut(dq), bptr_mut(dk), bptr_mut(dv), (int)B, (int)Sq, (int)Sk, (int)Hq, (int)Hkv, (int)D,
                         causal ? 1 : 0, (int)q_offset, (float)scale, cur_stream());
  return {dq, dk, dv};
}

// Attention of q [B,Sq,Hq,D] against a static cache [B,Hkv,Lmax,D]; the key
// count per sequence comes from device memory (kv_len[b]) and the causal
// offset is kv_len[b] - Sq.  Writes into `out` [B,Sq,Hq,D].
void attention_cache(const Tensor& q, const Tensor& k_cache, const Tensor& v_cache,
                     const Tensor& out, const Tensor& kv_len, double scale) {
  check_bf16(q, "q"); check_bf16(k_cache, "k_cache"); check_bf16(v_cache, "v_cache");
  check_bf16(out, "out"); check_i32(kv_len, "kv_len");
  TORCH_CHECK(q.dim() == 4 && k_cache.dim() == 4 && out.sizes() == q.sizes());
  const int64_t B = q.size(0), Sq = q.size(1), Hq = q.size(2), D = q.size(3);
  const int64_t Hkv = k_cache.size(1), Lmax = k_cache.size(2);
  TORCH_CHECK(k_cache.size(0) == B && k_cache.size(3) == D && v_cache.sizes() == k_cache.sizes());
  TORCH_CHECK(kv_len.numel() == B && (D == 64 || D == 128) && Hq % Hkv == 0);
  long long s[12] = {Sq * Hq * D, Hq * D, D, Hkv * Lmax * D, D, Lmax * D,
                     Hkv * Lmax * D, D, Lmax * D, Sq * Hq * D, Hq * D, D};
  // a handful of query rows against a long cache (speculative verify): one block
  // per head would stream the whole cache serially, so split the keys over blocks
  const int nsplit = (Sq <= 64 && Lmax >= 512) ? 16 : 0;
  Tensor part_o, part_ml;
  if (nsplit > 0) {
    part_o = at::empty({B, Hq, nsplit, Sq, D}, q.options().dtype(at::kFloat));
    part_ml = at::empty({B, Hq, nsplit, Sq, 2}, q.options().dtype(at::kFloat));
  }
  evgpt::launch_attn_prefill(bptr(q), bptr(k_cache), bptr(v_cache), bptr_mut(out), s,
                             kv_len.data_ptr<int>(), (int)B, (int)Sq, (int)Lmax, (int)Hq, (int)Hkv,
                             (int)D, 1, 0, 1, (float)scale, cur_stream(),
                             nsplit ? part_o.data_ptr<float>() : nullptr,
                             nsplit ? part_ml.data_ptr<float>() : nullptr, nsplit);
}

Tensor st_pool(const Tensor& f, int64_t n_temporal) {
  check_bf16(f, "features");
  TORCH_CHECK(f.dim() == 3 && f.size(2) % 8 == 0, "st_pool: features must be [T,S,C], C % 8 == 0");
  const int64_t T = f.size(0), S = f.size(1), C = f.size(2);
  Tensor out = at::empty({n_temporal + S, C}, f.options());
  evgpt::launch_st_pool(bptr(f), bptr_mut(out), (int)T, (int)S, (int)C, (int)n_temporal,
                        cur_stream());
  return out;
}

Tensor argmax_rows(const Tensor& x) {
  TORCH_CHECK(x.is_cuda() && x.is_contiguous() && x.dim() >= 1);
  const int64_t N = x.size(-1), rows = x.numel() / N;
  auto sizes = x.sizes().vec();
  sizes.pop_back();
  Tensor out = at::empty(sizes, x.options().dtype(at::kLong));
  if (rows == 0) return out;
  if (x.scalar_type() == at::kFloat)
    evgpt::launch_argmax_rows_f32(x.data_ptr<float>(), reinterpret_cast<long long*>(out.data_ptr<int64_t>()), (int)rows, (int)N,
                                  cur_stream());
  else if (x.scalar_type() == at::kBFloat16)
    evgpt::launch_argmax_rows_bf16(bptr(x), reinterpret_cast<long long*>(out.data_ptr<int64_t>()), (int)rows, (int)N,
                                   cur_stream());
  else
    TORCH_CHECK(false, "argmax_rows: float32 or bfloat16 only");
  return out;
}

// ------------------------------------------------------------ engine ops
// x [B,H] -> q_out [B,Hq*D], append K/V rows at pos[b] in caches [B,Hkv,Lmax,D]
void decode_qkv(const Tensor& x, const Tensor& norm_w, const Tensor& w_qkv, const Tensor& q_out,
                const Tensor& k_cache, const Tensor& v_cache, const Tensor& cos, const Tensor& sin,
                const Tensor& pos, double eps, int64_t Hq, int64_t Hkv, int64_t D, bool one_seq) {
  // one_seq: the rows of x are consecutive positions pos[0], pos[0]+1, ... of the single
  // sequence whose cache slice [1,Hkv,Lmax,D] is passed (speculative verify); otherwise
  // row b is sequence b at pos[b] (decode).
  check_bf16(x, "x"); check_bf16(norm_w, "norm_w"); check_bf16(w_qkv, "w_qkv");
  check_bf16(q_out, "q_out"); check_bf16(k_cache, "k_cache"); check_bf16(v_cache, "v_cache");
  check_f32(cos, "cos"); check_f32(sin, "sin"); check_i32(pos, "pos");
  const int64_t B = x.size(0), K = x.size(1);
  TORCH_CHECK(x.dim() == 2 && B <= evgpt::gemv_max_batch() && K % 8 == 0 && norm_w.numel() == K);
  TORCH_CHECK(w_qkv.size(0) == (Hq + 2 * Hkv) * D && w_qkv.size(1) == K, "decode_qkv: bad w_qkv");
  TORCH_CHECK(D % 16 == 0 && q_out.numel() == B * Hq * D && pos.numel() == (one_seq ? 1 : B));
  TORCH_CHECK(k_cache.dim() == 4 && k_cache.size(0) == (one_seq ? 1 : B) && k_cache.size(1) == Hkv &&
              k_cache.size(3) == D && v_cache.sizes() == k_cache.sizes());
  TORCH_CHECK(cos.size(-1) == D / 2 && cos.size(0) >= k_cache.size(2),
              "decode_qkv: rope table shorter than the cache");
  TORCH_CHECK(B * K * 2 <= 96 * 1024);
  evgpt::launch_gemv_qkv_rope(bptr(w_qkv), bptr(x), bptr(norm_w), (float)eps, bptr_mut(q_out),
                              bptr_mut(k_cache), bptr_mut(v_cache), cos.data_ptr<float>(),
                              sin.data_ptr<float>(), pos.data_ptr<int>(), (int)B, (int)Hq, (int)Hkv,
                              (int)D, (int)k_cache.size(2), (int)K, cur_stream(), one_seq ? 1 : 0);
}

// (dx, dw): dw is the fp32 weight gradient when want_dw, else an empty tensor
std::tuple<Tensor, Tensor> rmsnorm_bwd(const Tensor& x, const Tensor& w, const Tensor& g, double eps,
                                       bool want_dw) {
  check_bf16(x, "x"); check_bf16(w, "weight"); check_bf16(g, "grad");
  const int64_t H = x.size(-1), rows = x.numel() / H;
  TORCH_CHECK(H % 8 == 0 && H <= 8192 && w.numel() == H && g.numel() == x.numel(),
              "rmsnorm_bwd: hidden size must be a multiple of 8 up to 8192");
  Tensor dx = at::empty_like(x);
  Tensor part;
  if (want_dw && rows > 0)
    part = at::empty({(int64_t)evgpt::rmsnorm_bwd_blocks((int)rows, true), H}, x.options().dtype(at::kFloat));
  evgpt::launch_rmsnorm_bwd(bptr(x), bptr(w), bptr(g), bptr_mut(dx), part.defined() ? part.data_ptr<float>() : nullptr,
                            (int)rows, (int)H, (float)eps, cur_stream());
  Tensor dw = part.defined() ? part.sum(0) : at::zeros({want_dw ? H : 0}, x.options().dtype(at::kFloat));
  return std::make_tuple(dx, dw);
}

// ---- training kernels of the hidden-state adapters (adapter_train.hip)
// qkv [B, S, 3 * H * 512] packed in-projection -> (out [B, S, H * 512], probs [B, H, 64, 64])
std::tuple<Tensor, Tensor> attn512_fwd(const Tensor& qkv, int64_t H, bool causal, const c10::optional<Tensor>& kpm,
                                       double scale, double drop_p, int64_t seed, bool save_probs) {
  check_bf16(qkv, "qkv");
  TORCH_CHECK(qkv.dim() == 3 && qkv.is_contiguous() && H >= 1 && qkv.size(2) == 3 * H * 512 && qkv.size(1) >= 1 &&
              qkv.size(1) <= 64 && drop_p >= 0 && drop_p < 1, "attn512_fwd: qkv [B, S <= 64, 3 * H * 512] contiguous bf16");
  const int64_t B = qkv.size(0), S = qkv.size(1), E = H * 512;
  const unsigned char* mask = nullptr;
  if (kpm.has_value() && kpm->defined()) {
    TORCH_CHECK(kpm->is_cuda() && kpm->is_contiguous() && kpm->numel() == B * S &&
                (kpm->scalar_type() == at::kBool || kpm->scalar_type() == at::kByte), "attn512_fwd: kpm bool [B, S]");
    mask = reinterpret_cast<const unsigned char*>(kpm->data_ptr());
  }
  Tensor out = at::empty({B, S, E}, qkv.options());
  Tensor probs = save_probs ? at::empty({B, H, 64, 64}, qkv.options()) : at::empty({0}, qkv.options());
  if (B > 0)
    evgpt::launch_attn512_fwd(bptr(qkv), bptr(qkv) + E, bptr(qkv) + 2 * E, S * 3 * E, 3 * E, bptr_mut(out),
                              save_probs ? bptr_mut(probs) : nullptr, mask, (int)B, (int)S, (int)H, causal ? 1 : 0,
                              (float)scale, (float)drop_p, (unsigned long long)seed, cur_stream());
  return {out, probs};
}

Tensor attn512_bwd(const Tensor& qkv, const Tensor& d_out, const Tensor& probs, int64_t H, double scale, double drop_p,
                   int64_t seed) {
  check_bf16(qkv, "qkv"); check_bf16(d_out, "d_out"); check_bf16(probs, "probs");
  TORCH_CHECK(qkv.dim() == 3 && qkv.is_contiguous() && qkv.size(2) == 3 * H * 512 && qkv.size(1) <= 64, "attn512_bwd: qkv");
  const int64_t B = qkv.size(0), S = qkv.size(1), E = H * 512;
  TORCH_CHECK(d_out.is_contiguous() && d_out.numel() == B * S * E && probs.is_contiguous() &&
              probs.numel() == B * H * 64 * 64, "attn512_bwd: d_out [B, S, H * 512], probs [B, H, 64, 64]");
  Tensor dqkv = at::empty_like(qkv);
  if (B > 0)
    evgpt::launch_attn512_bwd(bptr(qkv), bptr(qkv) + E, bptr(qkv) + 2 * E, S * 3 * E, 3 * E, bptr(d_out), bptr(probs),
                              bptr_mut(dqkv), bptr_mut(dqkv) + E, bptr_mut(dqkv) + 2 * E, (int)B, (int)S, (int)H,
                              (float)scale, (float)drop_p, (unsigned long long)seed, cur_stream());
  return dqkv;
}

// keep mask [B, H, 64, 64] (uint8) of the attention dropout for (drop_p, seed)
Tensor attn512_mask(int64_t B, int64_t H, double drop_p, int64_t seed, const Tensor& like) {
  Tensor out = at::empty({B, H, 64, 64}, like.options().dtype(at::kByte));
  evgpt::launch_attn512_mask(out.data_ptr<uint8_t>(), out.numel(), (float)drop_p, (unsigned long long)seed, cur_stream());
  return out;
}

static void check_ew(const Tensor& x, const char* what) {
  check_bf16(x, what);
  TORCH_CHECK(x.is_contiguous() && x.numel() % 8 == 0 && (reinterpret_cast<uintptr_t>(x.data_ptr()) & 15) == 0,
              what, ": contiguous bf16, a multiple of 8 elements, 16-byte aligned");
}
Tensor gelu_fwd(const Tensor& x) {
  check_ew(x, "gelu");
  Tensor y = at::empty_like(x);
  evgpt::launch_gelu(bptr(x), bptr_mut(y), x.numel(), cur_stream());
  return y;
}
Tensor gelu_bwd(const Tensor& x, const Tensor& g) {
  check_ew(x, "gelu_bwd x"); check_ew(g, "gelu_bwd grad");
  TORCH_CHECK(g.numel() == x.numel());
  Tensor dx = at::empty_like(x);
  evgpt::launch_gelu_bwd(bptr(x), bptr(g), bptr_mut(dx), x.numel(), cur_stream());
  return dx;
}
// y = keep ? x / (1 - p) : 0 with the keep mask a function of (seed, element index): the same call on the
// incoming gradient is the backward
Tensor dropout_fwd(const Tensor& x, double p, int64_t seed) {
  check_ew(x, "dropout");
  TORCH_CHECK(p >= 0 && p < 1);
  Tensor y = at::empty_like(x);
  evgpt::launch_dropout(bptr(x), bptr_mut(y), x.numel(), (float)p, (unsigned long long)seed, cur_stream());
  return y;
}

// (dx, dw, db): fp32 parameter gradients when want_params, else empty tensors
std::tuple<Tensor, Tensor, Tensor> layernorm_bwd(const Tensor& x, const Tensor& w, const Tensor& g, double eps,
                                                 bool want_params) {
  check_bf16(x, "x"); check_bf16(w, "weight"); check_bf16(g, "grad");
  const int64_t H = x.size(-1), rows = x.numel() / H;
  TORCH_CHECK(H % 8 == 0 && H <= 8192 && w.numel() == H && g.numel() == x.numel(),
              "layernorm_bwd: hidden size must be a multiple of 8 up to 8192");
  Tensor dx = at::empty_like(x);
  auto f32 = x.options().dtype(at::kFloat);
  Tensor pw, pb;
  if (want_params && rows > 0) {
    const int64_t nb = evgpt::layernorm_bwd_blocks((int)rows);
    pw = at::empty({nb, H}, f32);
    pb = at::empty({nb, H}, f32);
  }
  evgpt::launch_layernorm_bwd(bptr(x), bptr(w), bptr(g), bptr_mut(dx), pw.defined() ? pw.data_ptr<float>() : nullptr,
                              pb.defined() ? pb.data_ptr<float>() : nullptr, (int)rows, (int)H, (float)eps, cur_stream());
  Tensor dw = pw.defined() ? pw.sum(0) : at::zeros({want_params ? H : 0}, f32);
  Tensor db = pb.defined() ? pb.sum(0) : at::zeros({want_params ? H : 0}, f32);
  return std::make_tuple(dx, dw, db);
}

// ---- loss.hip
// logits [M, V] bf16 (any row stride), labels int32 [M] -> (lse [M], loss rows [M], stat [2] = mean, 1 / count)
static const float* row_weights(const c10::optional<Tensor>& w, int64_t M) {
  if (!w.has_value() || !w->defined()) return nullptr;
  check_f32(*w, "row_weight");
  TORCH_CHECK(w->numel() == M, "row_weight: one weight per row");
  return w->data_ptr<float>();
}
std::tuple<Tensor, Tensor, Tensor> ce_fwd(const Tensor& logits, const Tensor& labels, int64_t ignore,
                                          const c10::optional<Tensor>& row_weight) {
  check_bf16(logits, "logits"); check_i32(labels, "labels");
  TORCH_CHECK(logits.dim() == 2 && logits.stride(1) == 1 && labels.numel() == logits.size(0) && logits.size(1) >= 1,
              "ce_fwd: logits [M, V] with unit column stride, labels [M]");
  const int64_t M = logits.size(0), V = logits.size(1);
  auto f32 = logits.options().dtype(at::kFloat);
  Tensor lse = at::empty({M}, f32), loss = at::empty({M}, f32), stat = at::zeros({2}, f32);
  evgpt::launch_ce_fwd(bptr(logits), logits.stride(0), labels.data_ptr<int>(), lse.data_ptr<float>(),
                       loss.data_ptr<float>(), stat.data_ptr<float>(), (int)M, (int)V, (int)ignore, cur_stream(),
                       row_weights(row_weight, M));
  return {lse, loss, stat};
}
// d logits [M, round_up(V, 64)] (padding columns zero) for the upstream gradient `upstream` (device scalar)
Tensor ce_bwd(const Tensor& logits, const Tensor& labels, const Tensor& lse, const Tensor& upstream, const Tensor& stat,
              int64_t ignore, const c10::optional<Tensor>& row_weight) {
  check_bf16(logits, "logits"); check_i32(labels, "labels"); check_f32(lse, "lse"); check_f32(upstream, "upstream");
  check_f32(stat, "stat");
  TORCH_CHECK(logits.dim() == 2 && logits.stride(1) == 1 && labels.numel() == logits.size(0) &&
              lse.numel() == logits.size(0) && upstream.numel() >= 1 && stat.numel() >= 2, "ce_bwd: shapes");
  const int64_t M = logits.size(0), V = logits.size(1), Vp = (V + 63) / 64 * 64;
  Tensor d = at::empty({M, Vp}, logits.options());
  evgpt::launch_ce_bwd(bptr(logits), logits.stride(0), labels.data_ptr<int>(), lse.data_ptr<float>(),
                       upstream.data_ptr<float>(), stat.data_ptr<float>(), bptr_mut(d), Vp, (int)M, (int)V, (int)ignore,
                       cur_stream(), row_weights(row_weight, M));
  return d;
}
Tensor rows_gather(const Tensor& src, const Tensor& idx) {
  check_bf16(src, "src"); check_i32(idx, "idx");
  TORCH_CHECK(src.dim() == 2 && src.is_contiguous() && src.size(1) % 8 == 0 && src.size(0) >= 1, "rows_gather: src [R, K % 8 == 0]");
  Tensor out = at::empty({idx.numel(), src.size(1)}, src.options());
  evgpt::launch_rows_gather(bptr(src), idx.data_ptr<int>(), bptr_mut(out), (int)idx.numel(), (int)src.size(1),
                            (int)src.size(0), cur_stream());
  return out;
}
// [R, K] with row idx[i] = g[i] and zeros elsewhere; idx ascending and unique
Tensor rows_scatter(const Tensor& g, const Tensor& idx, int64_t R) {
  check_bf16(g, "g"); check_i32(idx, "idx");
  TORCH_CHECK(g.dim() == 2 && g.is_contiguous() && g.size(1) % 8 == 0 && g.size(0) == idx.numel() && R >= 0, "rows_scatter: g [n, K % 8 == 0]");
  Tensor out = at::empty({R, g.size(1)}, g.options());
  evgpt::launch_rows_scatter(bptr(g), idx.data_ptr<int>(), bptr_mut(out), (int)idx.numel(), (int)g.size(1), (int)R, cur_stream());
  return out;
}

Tensor swiglu_bwd(const Tensor& gate_up, const Tensor& gh) {
  check_bf16(gate_up, "gate_up"); check_bf16(gh, "grad");
  const int64_t I2 = gate_up.size(-1), M = gate_up.numel() / I2;
  TORCH_CHECK(I2 % 16 == 0 && gh.numel() == M * (I2 / 2), "swiglu_bwd: shapes");
  Tensor dgu = at::empty_like(gate_up);
  evgpt::launch_swiglu_bwd(bptr(gate_up), bptr(gh), bptr_mut(dgu), M, (int)(I2 / 2), cur_stream());
  return dgu;
}

// One optimizer shard: grad/param_out in the parameters' storage type (bf16 or fp32), fp32
// master weights and moments, all of the same length; updates master, moments and param_out.
void fused_adamw(const Tensor& grad, const Tensor& master, const Tensor& exp_avg, const Tensor& exp_avg_sq,
                 const Tensor& param_out, double lr, double beta1, double beta2, double eps,
                 double weight_decay, int64_t step, double clip, const c10::optional<Tensor>& clip_dev) {
  check_f32(master, "master"); check_f32(exp_avg, "exp_avg"); check_f32(exp_avg_sq, "exp_avg_sq");
  const int64_t n = master.numel();
  TORCH_CHECK(grad.is_cuda() && grad.is_contiguous() && param_out.is_cuda() && param_out.is_contiguous() &&
              grad.numel() == n && param_out.numel() == n && exp_avg.numel() == n && exp_avg_sq.numel() == n,
              "fused_adamw: all tensors are contiguous device tensors of one length");
  TORCH_CHECK(grad.scalar_type() == param_out.scalar_type() && step >= 1, "fused_adamw: dtype/step");
  evgpt::AdamWArgs a;
  a.clip_dev = nullptr;
  if (clip_dev.has_value()) {
    check_f32(*clip_dev, "clip_dev");
    TORCH_CHECK(clip_dev->numel() == 1, "fused_adamw: clip_dev is one device scalar");
    a.clip_dev = clip_dev->data_ptr<float>();
  }
  a.clip = (float)clip; a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps;
  a.decay = (float)(1.0 - lr * weight_decay);
  a.step_size = (float)(lr / (1.0 - std::pow(beta1, (double)step)));
  a.inv_sqrt_bc2 = (float)(1.0 / std::sqrt(1.0 - std::pow(beta2, (double)step)));
  if (grad.scalar_type() == at::kBFloat16)
    evgpt::launch_fused_adamw_bf16(bptr(grad), master.data_ptr<float>(), exp_avg.data_ptr<float>(),
                                   exp_avg_sq.data_ptr<float>(), bptr_mut(param_out), n, a, cur_stream());
  else if (grad.scalar_type() == at::kFloat)
    evgpt::launch_fused_adamw_f32(grad.data_ptr<float>(), master.data_ptr<float>(), exp_avg.data_ptr<float>(),
                                  exp_avg_sq.data_ptr<float>(), param_out.data_ptr<float>(), n, a, cur_stream());
  else
    TORCH_CHECK(false, "fused_adamw: bfloat16 or float32 parameters");
}

// acc[0] += sum(x^2) over a contiguous bf16 / fp32 device tensor (16-byte aligned), deterministic
void sumsq_accumulate(const Tensor& x, const Tensor& acc) {
  check_f32(acc, "acc");
  TORCH_CHECK(x.is_cuda() && x.is_contiguous() && (x.scalar_type() == at::kBFloat16 || x.scalar_type() == at::kFloat),
              "sumsq_accumulate: contiguous bf16 / fp32 device tensor");
  TORCH_CHECK(reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 == 0, "sumsq_accumulate: 16-byte aligned input");
  const int64_t n = x.numel();
  if (n == 0) return;
  Tensor part = at::empty({evgpt::sumsq_blocks(n)}, acc.options());
  evgpt::launch_sumsq(x.data_ptr(), x.scalar_type() == at::kBFloat16, n, part.data_ptr<float>(), acc.data_ptr<float>(),
                      cur_stream());
}

// ---- 4-bit codebook weights: `packed` uint8 [N, K/2], `absmax` fp32 [N, K/64], `code` a
// host bf16 tensor with the 16 codebook entries (read here on the host, so it must not
// live on the device: these ops run inside graph captures)
static evgpt::W4Desc w4_desc(const Tensor& packed, const Tensor& absmax, const Tensor& code,
                             int64_t N, int64_t K) {
  TORCH_CHECK(packed.is_cuda() && packed.scalar_type() == at::kByte && packed.is_contiguous(),
              "w4: packed must be a contiguous uint8 device tensor");
  check_f32(absmax, "absmax");
  TORCH_CHECK(K % 64 == 0 && packed.numel() == N * (K / 2) && absmax.numel() == N * (K / 64),
              "w4: packed/absmax do not match [", N, ",", K, "]");
  TORCH_CHECK(!code.is_cuda() && code.scalar_type() == at::kBFloat16 && code.numel() == 16 &&
              code.is_contiguous(), "w4: code must be a host bf16 tensor of 16 entries");
  const uint16_t* c = reinterpret_cast<const uint16_t*>(code.data_ptr());
  evgpt::W4Desc d{};
  d.absmax = absmax.data_ptr<float>();
  for (int e = 0; e < 16; ++e) {
    d.code.lo[e >> 2] |= static_cast<unsigned int>(c[e] & 0xFF) << (8 * (e & 3));
    d.code.hi[e >> 2] |= static_cast<unsigned int>(c[e] >> 8) << (8 * (e & 3));
  }
  return d;
}

Tensor w4_dequant(const Tensor& packed, const Tensor& absmax, const Tensor& code,
                  const c10::optional<Tensor>& out) {
  TORCH_CHECK(packed.dim() == 2, "w4_dequant: packed [N, K/2]");
  const int64_t N = packed.size(0), K = packed.size(1) * 2;
  const evgpt::W4Desc d = w4_desc(packed, absmax, code, N, K);
  Tensor o = out.has_value() ? *out : at::empty({N, K}, packed.options().dtype(at::kBFloat16));
  check_bf16(o, "out");
  TORCH_CHECK(o.numel() == N * K, "w4_dequant: out size");
  evgpt::launch_w4_dequant(packed.data_ptr<uint8_t>(), d.absmax, d.code, bptr_mut(o), N, (int)K,
                           cur_stream());
  return o;
}

void decode_qkv_w4(const Tensor& x, const Tensor& norm_w, const Tensor& packed, const Tensor& absmax,
                   const Tensor& code, const Tensor& q_out, const Tensor& k_cache, const Tensor& v_cache,
                   const Tensor& cos, const Tensor& sin, const Tensor& pos, double eps, int64_t Hq,
                   int64_t Hkv, int64_t D, bool one_seq) {
  check_bf16(x, "x"); check_bf16(norm_w, "norm_w");
  check_bf16(q_out, "q_out"); check_bf16(k_cache, "k_cache"); check_bf16(v_cache, "v_cache");
  check_f32(cos, "cos"); check_f32(sin, "sin"); check_i32(pos, "pos");
  const int64_t B = x.size(0), K = x.size(1);
  TORCH_CHECK(x.dim() == 2 && B <= evgpt::gemv_w4_max_batch() && norm_w.numel() == K);
  const evgpt::W4Desc d = w4_desc(packed, absmax, code, (Hq + 2 * Hkv) * D, K);
  TORCH_CHECK(D % 16 == 0 && q_out.numel() == B * Hq * D && pos.numel() == (one_seq ? 1 : B));
  TORCH_CHECK(k_cache.dim() == 4 && k_cache.size(0) == (one_seq ? 1 : B) && k_cache.size(1) == Hkv &&
              k_cache.size(3) == D && v_cache.sizes() == k_cache.sizes());
  TORCH_CHECK(cos.size(-1) == D / 2 && cos.size(0) >= k_cache.size(2),
              "decode_qkv_w4: rope table shorter than the cache");
  TORCH_CHECK(B * K * 2 <= 96 * 1024);
  evgpt::launch_gemv_qkv_rope(reinterpret_cast<const evgpt::bf16_t*>(packed.data_ptr<uint8_t>()), bptr(x),
                              bptr(norm_w), (float)eps, bptr_mut(q_out), bptr_mut(k_cache),
                              bptr_mut(v_cache), cos.data_ptr<float>(), sin.data_ptr<float>(),
                              pos.data_ptr<int>(), (int)B, (int)Hq, (int)Hkv, (int)D,
                              (int)k_cache.size(2), (int)K, cur_stream(), one_seq ? 1 : 0, &d);
}

// y (in place) = y + W4 @ xin
void decode_gemv_residual_w4(const Tensor& packed, const Tensor& absmax, const Tensor& code,
                             const Tensor& xin, const Tensor& y) {
  check_bf16(xin, "xin"); check_bf16(y, "y");
  TORCH_CHECK(xin.dim() == 2 && packed.dim() == 2);
  const int64_t B = xin.size(0), K = xin.size(1), N = packed.size(0);
  const evgpt::W4Desc d = w4_desc(packed, absmax, code, N, K);
  TORCH_CHECK(y.numel() == B * N && N % 4 == 0);
  TORCH_CHECK(B <= evgpt::gemv_w4_max_batch() && B * K * 2 <= 96 * 1024);
  evgpt::launch_gemv(reinterpret_cast<const evgpt::bf16_t*>(packed.data_ptr<uint8_t>()), bptr(xin),
                     bptr_mut(y), nullptr, bptr(y), nullptr, 0.f, 0, (int)B, (int)N, (int)K,
                     cur_stream(), &d);
}

void decode_mlp_up_w4(const Tensor& packed, const Tensor& absmax, const Tensor& code, const Tensor& x,
                      const Tensor& norm_w, double eps, const Tensor& h) {
  check_bf16(x, "x"); check_bf16(norm_w, "norm_w"); check_bf16(h, "h");
  TORCH_CHECK(x.dim() == 2 && packed.dim() == 2);
  const int64_t B = x.size(0), K = x.size(1), I = packed.size(0) / 2;
  const evgpt::W4Desc d = w4_desc(packed, absmax, code, 2 * I, K);
  TORCH_CHECK(h.numel() == B * I && I % 2 == 0 && norm_w.numel() == K);
  TORCH_CHECK(B <= evgpt::gemv_w4_max_batch() && B * K * 2 <= 96 * 1024);
  evgpt::launch_gemv_swiglu(reinterpret_cast<const evgpt::bf16_t*>(packed.data_ptr<uint8_t>()), bptr(x),
                            bptr_mut(h), bptr(norm_w), (float)eps, (int)B, (int)I, (int)K, cur_stream(),
                            &d);
}

void decode_attn(const Tensor& q, const Tensor& k_cache, const Tensor& v_cache, const Tensor& out,
                 const Tensor& part_o, const Tensor& part_ml, const Tensor& pos, int64_t nsplit,
                 double scale) {
  check_bf16(q, "q"); check_bf16(k_cache, "k_cache"); check_bf16(v_cache, "v_cache");
  check_bf16(out, "out"); check_f32(part_o, "part_o"); check_f32(part_ml, "part_ml");
  check_i32(pos, "pos");
  const int64_t B = k_cache.size(0), Hkv = k_cache.size(1), Lmax = k_cache.size(2),
                D = k_cache.size(3);
  TORCH_CHECK(D == 64 || D == 128, "decode_attn: head_dim 64 or 128");
  const int64_t Hq = q.numel() / (B * D);
  TORCH_CHECK(Hq * B * D == q.numel() && Hq % Hkv == 0 && out.numel() == q.numel());
  TORCH_CHECK(part_o.numel() >= B * Hq * nsplit * D && part_ml.numel() >= B * Hq * nsplit * 2);
  TORCH_CHECK(pos.numel() == B && nsplit >= 1 && nsplit <= 16, "decode_attn: 1..16 splits");
  evgpt::launch_attn_decode(bptr(q), bptr(k_cache), bptr(v_cache), bptr_mut(out),
                            part_o.data_ptr<float>(), part_ml.data_ptr<float>(),
                            pos.data_ptr<int>(), (int)B, (int)Hq, (int)Hkv, (int)D, (int)Lmax,
                            (int)nsplit, (float)scale, cur_stream());
}

// y (in place) = y + W @ xin      xin [B,K], y [B,N]
void decode_gemv_residual(const Tensor& w, const Tensor& xin, const Tensor& y) {
  check_bf16(w, "w"); check_bf16(xin, "xin"); check_bf16(y, "y");
  const int64_t B = xin.size(0), K = xin.size(1), N = w.size(0);
  TORCH_CHECK(xin.dim() == 2 && w.size(1) == K && y.numel() == B * N && N % 4 == 0 && K % 8 == 0);
  TORCH_CHECK(B <= evgpt::gemv_max_batch() && B * K * 2 <= 96 * 1024);
  evgpt::launch_gemv(bptr(w), bptr(xin), bptr_mut(y), nullptr, bptr(y), nullptr, 0.f, 0, (int)B,
                     (int)N, (int)K, cur_stream());
}

// y = W @ xin (+ y when accumulate).  Tensor-parallel row-parallel projections
// use accumulate on rank 0 only, so the all-reduce that follows adds the residual once.
void decode_gemv(const Tensor& w, const Tensor& xin, const Tensor& y, bool accumulate) {
  check_bf16(w, "w"); check_bf16(xin, "xin"); check_bf16(y, "y");
  const int64_t B = xin.size(0), K = xin.size(1), N = w.size(0);
  TORCH_CHECK(xin.dim() == 2 && w.size(1) == K && y.numel() == B * N && N % 4 == 0 && K % 8 == 0);
  TORCH_CHECK(B <= evgpt::gemv_max_batch() && B * K * 2 <= 96 * 1024);
  evgpt::launch_gemv(bptr(w), bptr(xin), bptr_mut(y), nullptr, accumulate ? bptr(y) : nullptr,
                     nullptr, 0.f, 0, (int)B, (int)N, (int)K, cur_stream());
}

// out_f32 [1,B,N] = W @ xin as raw fp32 sums: the local partial of a row-parallel projection,
// summed over the ranks (and rounded once) by the all-reduce that follows
void decode_gemv_partial(const Tensor& w, const Tensor& xin, const Tensor& out_f32) {
  check_bf16(w, "w"); check_bf16(xin, "xin"); check_f32(out_f32, "out");
  const int64_t B = xin.size(0), K = xin.size(1), N = w.size(0);
  TORCH_CHECK(xin.dim() == 2 && w.size(1) == K && out_f32.numel() == B * N && N % 4 == 0 && K % 8 == 0);
  TORCH_CHECK(B <= evgpt::gemv_max_batch() && B * K * 2 <= 96 * 1024);
  evgpt::launch_gemv(bptr(w), bptr(xin), nullptr, nullptr, nullptr, nullptr, 0.f, 0, (int)B, (int)N, (int)K,
                     cur_stream(), nullptr, out_f32.data_ptr<float>());
}

// h = silu(Wg rmsnorm(x)) * (Wu rmsnorm(x));  w_gate_up [2I,K] = [gate rows | up rows]
void decode_mlp_up(const Tensor& w_gate_up, const Tensor& x, const Tensor& norm_w, double eps,
                   const Tensor& h) {
  check_bf16(w_gate_up, "w_gate_up"); check_bf16(x, "x"); check_bf16(norm_w, "norm_w");
  check_bf16(h, "h");
  const int64_t B = x.size(0), K = x.size(1), I = w_gate_up.size(0) / 2;
  TORCH_CHECK(x.dim() == 2 && w_gate_up.size(1) == K && h.numel() == B * I && I % 2 == 0 &&
              K % 8 == 0 && norm_w.numel() == K);
  TORCH_CHECK(B <= evgpt::gemv_max_batch() && B * K * 2 <= 96 * 1024);
  evgpt::launch_gemv_swiglu(bptr(w_gate_up), bptr(x), bptr_mut(h), bptr(norm_w), (float)eps, (int)B,
                            (int)I, (int)K, cur_stream());
}

int64_t logits_partials(int64_t N) { return evgpt::gemv_logits_grid((int)N); }

void decode_logits(const Tensor& w, const Tensor& x, const c10::optional<Tensor>& norm_w,
                   double eps, const c10::optional<Tensor>& logits, const Tensor& part_val,
                   const Tensor& part_idx) {
  check_bf16(w, "w"); check_bf16(x, "x"); check_f32(part_val, "part_val");
  check_i32(part_idx, "part_idx");
  const int64_t B = x.size(0), K = x.size(1), N = w.size(0);
  TORCH_CHECK(x.dim() == 2 && w.size(1) == K && K % 8 == 0 && B <= evgpt::gemv_max_batch());
  const int64_t np = evgpt::gemv_logits_grid((int)N);
  TORCH_CHECK(part_val.numel() >= B * np && part_idx.numel() >= B * np);
  float* lp = nullptr;
  if (logits.has_value() && logits->defined()) {
    check_f32(*logits, "logits");
    TORCH_CHECK(logits->numel() == B * N);
    lp = logits->data_ptr<float>();
  }
  if (norm_w.has_value()) { check_bf16(*norm_w, "norm_w"); TORCH_CHECK(norm_w->numel() == K); }
  evgpt::launch_gemv_logits(bptr(w), bptr(x), bptr_opt(norm_w), (float)eps, lp,
                            part_val.data_ptr<float>(), part_idx.data_ptr<int>(), (int)B, (int)N,
                            (int)K, cur_stream());
}

void commit_token(const Tensor& part_val, const Tensor& part_idx, int64_t n_part,
                  const c10::optional<Tensor>& sampled, const Tensor& embed, const Tensor& x,
                  const Tensor& out_tokens, const Tensor& step, const Tensor& pos,
                  const Tensor& done, const Tensor& cur_token, int64_t eos,
                  const c10::optional<Tensor>& active) {
  check_f32(part_val, "part_val"); check_i32(part_idx, "part_idx"); check_bf16(embed, "embed");
  check_bf16(x, "x"); check_i32(out_tokens, "out_tokens"); check_i32(step, "step");
  check_i32(pos, "pos"); check_i32(done, "done"); check_i32(cur_token, "cur_token");
  const int64_t B = x.size(0), H = x.size(1);
  TORCH_CHECK(embed.size(1) == H && H % 8 == 0 && out_tokens.dim() == 2 && out_tokens.size(0) == B);
  TORCH_CHECK(step.numel() == B && pos.numel() == B && done.numel() == B && cur_token.numel() == B);
  const int* sp = nullptr;
  if (sampled.has_value() && sampled->defined()) { check_i32(*sampled, "sampled"); sp = sampled->data_ptr<int>(); }
  else TORCH_CHECK(part_val.numel() >= B * n_part && part_idx.numel() >= B * n_part);
  const int* ap = nullptr;
  if (active.has_value() && active->defined()) {
    check_i32(*active, "active");
    TORCH_CHECK(active->numel() == B, "commit_token: one active flag per sequence");
    ap = active->data_ptr<int>();
  }
  evgpt::launch_commit_token(part_val.data_ptr<float>(), part_idx.data_ptr<int>(), (int)n_part, sp,
                             bptr(embed), bptr_mut(x), out_tokens.data_ptr<int>(),
                             (int)out_tokens.size(1), step.data_ptr<int>(), pos.data_ptr<int>(),
                             done.data_ptr<int>(), cur_token.data_ptr<int>(), (int)eos, (int)B,
                             (int)H, (int)embed.size(0), cur_stream(), ap);
}

// qkv [B,S,(Hq+2Hkv)*D] -> q_out [B,S,Hq*D] (rotated) + cache append at pos0[b]+s
void prefill_rope_kv(const Tensor& qkv, const Tensor& q_out, const Tensor& k_cache,
                     const Tensor& v_cache, const Tensor& cos, const Tensor& sin,
                     const Tensor& pos0, int64_t Hq, int64_t Hkv, int64_t D) {
  check_bf16(qkv, "qkv"); check_bf16(q_out, "q_out"); check_bf16(k_cache, "k_cache");
  check_bf16(v_cache, "v_cache"); check_f32(cos, "cos"); check_f32(sin, "sin");
  check_i32(pos0, "pos0");
  TORCH_CHECK(qkv.dim() == 3 && qkv.size(2) == (Hq + 2 * Hkv) * D && D % 16 == 0);
  const int64_t B = qkv.size(0), S = qkv.size(1);
  TORCH_CHECK(q_out.numel() == B * S * Hq * D && pos0.numel() == B);
  TORCH_CHECK(k_cache.size(0) == B && k_cache.size(1) == Hkv && k_cache.size(3) == D &&
              v_cache.sizes() == k_cache.sizes() && k_cache.size(2) >= S);
  TORCH_CHECK(cos.size(0) >= k_cache.size(2) && cos.size(-1) == D / 2);
  evgpt::launch_rope_kv_prefill(bptr(qkv), bptr_mut(q_out), bptr_mut(k_cache), bptr_mut(v_cache),
                                cos.data_ptr<float>(), sin.data_ptr<float>(), pos0.data_ptr<int>(),
                                (int)B, (int)S, (int)Hq, (int)Hkv, (int)D, (int)k_cache.size(2),
                                cur_stream());
}

// x [S, K] @ w_qkv^T with RoPE + cache append in the GEMM's epilogue (head_dim 128, one sequence): q_out
// [S, Hq * 128], k / v into k_cache / v_cache [1, Hkv, Lmax, 128] at pos0[0] + row.  Returns false (and
// launches nothing) when the fused tile does not take the shape: the caller then runs GEMM + prefill_rope_kv.
bool linear_qkv_rope(const Tensor& x, const Tensor& w_qkv, const Tensor& q_out, const Tensor& k_cache,
                     const Tensor& v_cache, const Tensor& cos, const Tensor& sin, const Tensor& pos0, int64_t Hq,
                     int64_t Hkv) {
  check_bf16(x, "x"); check_bf16(w_qkv, "w_qkv"); check_bf16(q_out, "q_out"); check_bf16(k_cache, "k_cache");
  check_bf16(v_cache, "v_cache"); check_f32(cos, "cos"); check_f32(sin, "sin"); check_i32(pos0, "pos0");
  const int64_t K = x.size(-1), S = x.numel() / K;
  if (K % 64 || S < 1 || k_cache.dim() != 4 || k_cache.size(0) != 1 || k_cache.size(3) != 128) return false;
  TORCH_CHECK(w_qkv.dim() == 2 && w_qkv.size(1) == K && w_qkv.size(0) == (Hq + 2 * Hkv) * 128, "linear_qkv_rope: w_qkv");
  TORCH_CHECK(q_out.numel() == S * Hq * 128 && k_cache.size(1) == Hkv && v_cache.sizes() == k_cache.sizes() &&
              pos0.numel() >= 1 && cos.size(-1) == 64 && sin.size(-1) == 64 && cos.size(0) >= k_cache.size(2) &&
              sin.size(0) >= k_cache.size(2), "linear_qkv_rope: shapes");
  return evgpt::launch_gemm_qkv_rope(bptr(x), bptr(w_qkv), (int)S, (int)K, bptr_mut(q_out), bptr_mut(k_cache),
                                     bptr_mut(v_cache), cos.data_ptr<float>(), sin.data_ptr<float>(),
                                     pos0.data_ptr<int>(), (int)Hq, (int)Hkv, (int)k_cache.size(2), cur_stream());
}

// Row argmax of (x @ w^T + bias) without materialising the product: the GEMM
// epilogue leaves per-(row, column-slab) partials; returns (val [M,P], idx [M,P]).
std::tuple<Tensor, Tensor> linear_argmax_partials(const Tensor& x, const Tensor& w,
                                                   const c10::optional<Tensor>& bias) {
  check_bf16(x, "x"); check_bf16(w, "weight");
  const int64_t K = x.size(-1), N = w.size(0), M = x.numel() / K;
  TORCH_CHECK(w.dim() == 2 && w.size(1) == K && K % 64 == 0 && M >= 1,
              "linear_argmax: K must be a multiple of 64");
  if (bias.has_value()) { check_bf16(*bias, "bias"); TORCH_CHECK(bias->numel() == N); }
  const int64_t P = evgpt::gemm_argmax_partials((int)M, (int)N);
  Tensor val = at::empty({M, P}, x.options().dtype(at::kFloat));
  Tensor idx = at::empty({M, P}, x.options().dtype(at::kInt));
  evgpt::launch_gemm_bf16(bptr(x), bptr(w), nullptr, nullptr, bptr_opt(bias), nullptr, (int)M,
                          (int)N, (int)K, (int)K, (int)K, (int)N, 0, val.data_ptr<float>(),
                          idx.data_ptr<int>(), nullptr, 1, cur_stream());
  return {val, idx};
}

void gemm_force_config(int64_t cfg) { evgpt::gemm_force_config((int)cfg); }
void gemm_enable_256(bool on) { evgpt::gemm_enable_256(on); }
void gemm_persist_256(bool on) { evgpt::gemm_persist_256(on); }
void gemm_group_m(int64_t g) { evgpt::gemm_group_m((int)g); }
void gemm_split_major(int64_t on) { evgpt::gemm_split_major((int)on); }
void attn_force_groups(int64_t g) { evgpt::attn_force_groups((int)g); }
void gemm_enable_tail_splitk(bool on) { evgpt::gemm_enable_tail_splitk(on); }
void attn_enable_wide(int64_t mode) { evgpt::attn_enable_wide((int)mode); }
void attn_wide4_min_rows(int64_t rows) { evgpt::attn_wide4_min_rows((int)rows); }
void gemm_force_splitk(int64_t sk) { evgpt::gemm_force_splitk((int)sk); }
void gemv_force_ks(int64_t ks) { evgpt::gemv_force_ks((int)ks); }

// ------------------------------------------------------------ event front end
// x,y,p int32 [n]; returns uint8 [T,H,W,3].  mode 0: last event wins, 1: negative wins.
Tensor event_raster(const Tensor& x, const Tensor& y, const Tensor& p, int64_t n_slices, int64_t H,
                    int64_t W, int64_t mode) {
  check_i32(x, "x"); check_i32(y, "y"); check_i32(p, "p");
  const int64_t n = x.numel();
  TORCH_CHECK(y.numel() == n && p.numel() == n && n_slices >= 1 && H >= 1 && W >= 1);
  TORCH_CHECK(n < (1ll << 30), "event_raster: at most 2^30 events per call");
  Tensor keys = at::full({n_slices, H, W}, mode == 0 ? -1 : 0, x.options());
  Tensor img = at::empty({n_slices, H, W, 3}, x.options().dtype(at::kByte));
  evgpt::launch_event_raster(x.data_ptr<int>(), y.data_ptr<int>(), p.data_ptr<int>(), (int)n,
                             (int)n_slices, (int)H, (int)W, keys.data_ptr<int>(),
                             img.data_ptr<uint8_t>(), (int)mode, cur_stream());
  return img;
}

// returns (hist int32 [bins,2,H,W], voxel fp32 [bins,H,W])
std::tuple<Tensor, Tensor> event_voxel(const Tensor& x, const Tensor& y, const Tensor& p,
                                       const Tensor& t, int64_t t0, int64_t span, int64_t bins,
                                       int64_t H, int64_t W, bool signed_mode, bool by_count) {
  check_i32(x, "x"); check_i32(y, "y"); check_i32(p, "p");
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kLong && t.is_contiguous());
  const int64_t n = x.numel();
  TORCH_CHECK(y.numel() == n && p.numel() == n && t.numel() == n && bins >= 1);
  Tensor hist = at::zeros({bins, 2, H, W}, x.options());
  Tensor vox = at::zeros({bins, H, W}, x.options().dtype(at::kFloat));
  evgpt::launch_event_voxel(x.data_ptr<int>(), y.data_ptr<int>(), p.data_ptr<int>(),
                            reinterpret_cast<const long long*>(t.data_ptr<int64_t>()), (int)n,
                            (long long)t0, (long long)span, (int)bins, (int)H, (int)W,
                            hist.data_ptr<int>(), vox.data_ptr<float>(), signed_mode ? 1 : 0,
                            by_count ? 1 : 0, cur_stream());
  return {hist, vox};
}

// Same outputs for a stream whose bins are index ranges: bounds int32 [bins+1] on the device,
// non-decreasing, bounds[0] >= 0, bounds[bins] <= n (checked by the caller that built them).
std::tuple<Tensor, Tensor> event_voxel_ranges(const Tensor& x, const Tensor& y, const Tensor& p,
                                              const Tensor& bounds, int64_t H, int64_t W, bool signed_mode,
                                              int64_t chunks) {
  check_i32(x, "x"); check_i32(y, "y"); check_i32(p, "p"); check_i32(bounds, "bounds");
  const int64_t n = x.numel(), bins = bounds.numel() - 1;
  TORCH_CHECK(y.numel() == n && p.numel() == n && bins >= 1 && H >= 1 && W >= 1);
  TORCH_CHECK(evgpt::event_voxel_band_rows((int)H, (int)W) >= 1, "event_voxel_ranges: image too wide for an LDS band");
  TORCH_CHECK(bins * 2 * H * W < (1ll << 31) && bins <= 65535);
  Tensor hist = at::zeros({bins, 2, H, W}, x.options());
  Tensor vox = at::zeros({bins, H, W}, x.options().dtype(at::kFloat));
  if (n > 0)
    evgpt::launch_event_voxel_ranges(x.data_ptr<int>(), y.data_ptr<int>(), p.data_ptr<int>(), bounds.data_ptr<int>(),
                                     (int)n, (int)bins, (int)H, (int)W, hist.data_ptr<int>(), vox.data_ptr<float>(),
                                     signed_mode ? 1 : 0, cur_stream(), (int)std::min<int64_t>(std::max<int64_t>(chunks, 0), 1024));
  return {hist, vox};
}

// img uint8 [T,H,W,3] -> bf16 [T,3,S,S]; tables are int32 GPU tensors
Tensor clip_preprocess(const Tensor& img, const Tensor& hstart, const Tensor& hcount,
                       const Tensor& hweight, const Tensor& vstart, const Tensor& vcount,
                       const Tensor& vweight, int64_t S, int64_t resize_h, int64_t resize_w,
                       int64_t top, int64_t left, std::vector<double> mean,
                       std::vector<double> stdv) {
  TORCH_CHECK(img.is_cuda() && img.scalar_type() == at::kByte && img.is_contiguous() &&
              img.dim() == 4 && img.size(3) == 3, "clip_preprocess: uint8 [T,H,W,3] expected");
  for (const Tensor* tb : {&hstart, &hcount, &hweight, &vstart, &vcount, &vweight}) check_i32(*tb, "table");
  const int64_t T = img.size(0), H = img.size(1), W = img.size(2);
  TORCH_CHECK(mean.size() == 3 && stdv.size() == 3);
  TORCH_CHECK(top >= 0 && left >= 0 && top + S <= resize_h && left + S <= resize_w);
  const bool need_h = resize_w != W, need_v = resize_h != H;
  TORCH_CHECK(!need_h || (hstart.numel() == resize_w && hcount.numel() == resize_w &&
                          hweight.dim() == 2 && hweight.size(0) == resize_w));
  TORCH_CHECK(!need_v || (vstart.numel() == resize_h && vcount.numel() == resize_h &&
                          vweight.dim() == 2 && vweight.size(0) == resize_h));
  Tensor out = at::empty({T, 3, S, S}, img.options().dtype(at::kBFloat16));
  float m[3] = {(float)mean[0], (float)mean[1], (float)mean[2]};
  float sd[3] = {(float)stdv[0], (float)stdv[1], (float)stdv[2]};
  evgpt::launch_clip_preproc(img.data_ptr<uint8_t>(), bptr_mut(out), hstart.data_ptr<int>(),
                             hcount.data_ptr<int>(), hweight.data_ptr<int>(),
                             vstart.data_ptr<int>(), vcount.data_ptr<int>(),
                             vweight.data_ptr<int>(), (int)T, (int)H, (int)W, (int)S,
                             (int)resize_h, (int)resize_w, (int)top, (int)left,
                             need_h ? (int)hweight.size(1) : 1, need_v ? (int)vweight.size(1) : 1,
                             m, sd, cur_stream());
  return out;
}

// pixels [T,C,S,S] -> patch rows [T*G*G, Kp] (zero padded to a multiple of 64)
Tensor patchify(const Tensor& px, int64_t P) {
  check_bf16(px, "pixels");
  TORCH_CHECK(px.dim() == 4 && px.size(2) == px.size(3) && px.size(2) % P == 0);
  const int64_t T = px.size(0), C = px.size(1), S = px.size(2), G = S / P;
  const int64_t Kp = ((C * P * P + 63) / 64) * 64;
  Tensor out = at::empty({T * G * G, Kp}, px.options());
  evgpt::launch_patchify(bptr(px), bptr_mut(out), (int)T, (int)C, (int)S, (int)P, (int)Kp,
                         cur_stream());
  return out;
}

// tok [T,N-1,H] + cls/pos embeddings + pre-LayerNorm -> [T,N,H]
Tensor vit_embed_ln(const Tensor& tok, const Tensor& cls, const Tensor& pos, const Tensor& w,
                    const Tensor& b, double eps) {
  check_bf16(tok, "tok"); check_bf16(cls, "cls"); check_bf16(pos, "pos"); check_bf16(w, "w");
  check_bf16(b, "b");
  TORCH_CHECK(tok.dim() == 3 && pos.dim() == 2);
  const int64_t T = tok.size(0), N = tok.size(1) + 1, H = tok.size(2);
  TORCH_CHECK(pos.size(0) == N && pos.size(1) == H && cls.numel() == H && w.numel() == H &&
              b.numel() == H && H <= 2048, "vit_embed_ln: shape mismatch (hidden <= 2048)");
  Tensor out = at::empty({T, N, H}, tok.options());
  evgpt::launch_vit_embed_ln(bptr(tok), bptr(cls), bptr(pos), bptr(w), bptr(b), bptr_mut(out),
                             (int)T, (int)N, (int)H, (float)eps, cur_stream());
  return out;
}

// ---------------------------------------------------------- speculative decode
void verify_accept(const Tensor& pval, const Tensor& pidx, const Tensor& draft, int64_t G,
                   const Tensor& out_tokens, const Tensor& step, const Tensor& pos,
                   const Tensor& done, const Tensor& cur_token, const Tensor& result, int64_t eos,
                   const Tensor& embed, const Tensor& x, const c10::optional<Tensor>& limit,
                   const c10::optional<Tensor>& stats) {
  check_f32(pval, "pval"); check_i32(pidx, "pidx"); check_i32(draft, "draft");
  check_i32(out_tokens, "out_tokens"); check_i32(step, "step"); check_i32(pos, "pos");
  check_i32(done, "done"); check_i32(cur_token, "cur_token"); check_i32(result, "result");
  check_bf16(embed, "embed"); check_bf16(x, "x");
  TORCH_CHECK(G >= 0 && G + 1 <= 64, "verify_accept: at most 63 draft tokens");
  TORCH_CHECK(pval.dim() == 2 && pval.size(0) >= G + 1 && pidx.sizes() == pval.sizes());
  TORCH_CHECK(draft.numel() >= G && result.numel() >= 6, "verify_accept: result holds 6 ints");
  const int* lim = nullptr;
  int* stp = nullptr;
  if (limit.has_value() && limit->defined()) { check_i32(*limit, "limit"); TORCH_CHECK(limit->numel() >= 1); lim = limit->data_ptr<int>(); }
  if (stats.has_value() && stats->defined()) { check_i32(*stats, "stats"); TORCH_CHECK(stats->numel() >= 3); stp = stats->data_ptr<int>(); }
  const int64_t H = embed.size(1);
  TORCH_CHECK(x.numel() >= H && H % 8 == 0);
  evgpt::launch_verify_accept(pval.data_ptr<float>(), pidx.data_ptr<int>(), (int)pval.size(1),
                              draft.data_ptr<int>(), (int)G, out_tokens.data_ptr<int>(),
                              (int)out_tokens.size(-1), step.data_ptr<int>(), pos.data_ptr<int>(),
                              done.data_ptr<int>(), cur_token.data_ptr<int>(),
                              result.data_ptr<int>(), (int)eos, bptr(embed), bptr_mut(x), (int)H,
                              (int)embed.size(0), cur_stream(), lim, stp);
}

// ---- draft head of a device-resident speculative-decoding iteration (adapter_ops.hip)
// rows src[clamp(index[0] + offset + i, 0, n_valid - 1)], i < rows; index / n_valid live on the device
Tensor gather_rows_dev(const Tensor& src, const Tensor& index, int64_t offset, const c10::optional<Tensor>& n_valid,
                       int64_t rows) {
  check_bf16(src, "src"); check_i32(index, "index");
  TORCH_CHECK(src.dim() == 2 && src.size(0) >= 1 && src.size(1) % 8 == 0 && rows >= 1 && index.numel() >= 1);
  const int* nv = nullptr;
  if (n_valid.has_value() && n_valid->defined()) { check_i32(*n_valid, "n_valid"); nv = n_valid->data_ptr<int>(); }
  Tensor out = at::empty({rows, src.size(1)}, src.options());
  evgpt::launch_gather_rows_dev(bptr(src), index.data_ptr<int>(), (int)offset, nv, (int)src.size(0), bptr_mut(out),
                                (int)rows, (int)src.size(1), cur_stream());
  return out;
}

// embeds [B,L,H]: row (b,l) = embed[map] / feats[-1 - map] / zeros (map <= -2^30), see splice_rows_kernel
Tensor splice_rows(const Tensor& embed, const Tensor& feats, const Tensor& map) {
  check_bf16(embed, "embed"); check_bf16(feats, "feats"); check_i32(map, "map");
  TORCH_CHECK(embed.dim() == 2 && feats.dim() == 2 && feats.size(1) == embed.size(1) && embed.size(1) % 8 == 0 &&
              feats.size(0) >= 1 && map.dim() == 2, "splice_rows: embed [V,H], feats [R,H], map [B,L]");
  Tensor out = at::empty({map.size(0), map.size(1), embed.size(1)}, embed.options());
  if (map.numel() > 0)
    evgpt::launch_splice_rows(bptr(embed), (int)embed.size(0), bptr(feats), (int)feats.size(0), map.data_ptr<int>(),
                              -(1 << 30), bptr_mut(out), map.numel(), (int)embed.size(1), cur_stream());
  return out;
}

void argmax_from_partials(const Tensor& pval, const Tensor& pidx, const Tensor& token, int64_t vocab) {
  check_f32(pval, "pval"); check_i32(pidx, "pidx"); check_i32(token, "token");
  TORCH_CHECK(pval.dim() == 2 && pidx.sizes() == pval.sizes() && token.numel() >= pval.size(0) && vocab >= 1);
  evgpt::launch_argmax_from_partials(pval.data_ptr<float>(), pidx.data_ptr<int>(), (int)pval.size(0),
                                     (int)pval.size(1), token.data_ptr<int>(), (int)vocab, cur_stream());
}

// self-attention with head_dim 512 (one wave per query row); q/k/v are [B,S,H,512] views with
// equal strides and contiguous heads (e.g. the thirds of a packed in-projection [B,S,3,H,512])
Tensor attn_d512(const Tensor& q, const Tensor& k, const Tensor& v, bool causal, double scale) {
  for (const Tensor* t : {&q, &k, &v}) {
    TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kBFloat16 && t->dim() == 4 && t->size(3) == 512 &&
                t->stride(3) == 1 && t->stride(2) == 512 && t->sizes() == q.sizes() && t->strides() == q.strides() &&
                (reinterpret_cast<uintptr_t>(t->data_ptr()) & 15) == 0 && t->stride(1) % 8 == 0 && t->stride(0) % 8 == 0,
                "attn_d512: [B,S,H,512] bf16 views with equal strides and contiguous heads");
  }
  const int64_t B = q.size(0), S = q.size(1), H = q.size(2);
  TORCH_CHECK(S >= 1 && S <= 4096);
  Tensor out = at::empty({B, S, H, 512}, q.options());
  evgpt::launch_attn_d512(bptr(q), bptr(k), bptr(v), bptr_mut(out), q.stride(0), q.stride(1), (int)B, (int)S, (int)H,
                          causal ? 1 : 0, (float)scale, cur_stream());
  return out;
}

// drafter re-sync of a token-level speculative-decoding round (sd.hip: naive_sd_sync_kernel)
void naive_sd_sync(const Tensor& t_pos, const Tensor& t_step, const Tensor& t_cur, const Tensor& t_out, int64_t offset,
                   const Tensor& embed, const Tensor& prompt_last, const Tensor& rows, const Tensor& d_pos0,
                   const Tensor& d_pos, const Tensor& d_step, const Tensor& d_done, int64_t l_max) {
  for (const Tensor* t : {&t_pos, &t_step, &t_cur, &t_out, &d_pos0, &d_pos, &d_step, &d_done}) check_i32(*t, "counter");
  check_bf16(embed, "embed"); check_bf16(prompt_last, "prompt_last"); check_bf16(rows, "rows");
  const int64_t H = embed.size(1);
  TORCH_CHECK(embed.dim() == 2 && H % 8 == 0 && prompt_last.numel() == H && rows.numel() == 2 * H && rows.is_contiguous() &&
              prompt_last.is_contiguous() && t_out.is_contiguous() && l_max >= 2, "naive_sd_sync: rows [2,H], prompt_last [H]");
  evgpt::launch_naive_sd_sync(t_pos.data_ptr<int>(), t_step.data_ptr<int>(), t_cur.data_ptr<int>(), t_out.data_ptr<int>(),
                              (int)t_out.size(-1), (int)offset, bptr(embed), (int)embed.size(0), bptr(prompt_last),
                              bptr_mut(rows), d_pos0.data_ptr<int>(), d_pos.data_ptr<int>(), d_step.data_ptr<int>(),
                              d_done.data_ptr<int>(), (int)H, (int)l_max, cur_stream());
}

void sample_top_p(const Tensor& logits, const Tensor& out, double temperature, double top_p,
                  int64_t seed, const Tensor& step, int64_t top_k) {
  check_f32(logits, "logits"); check_i32(out, "out"); check_i32(step, "step");
  TORCH_CHECK(logits.dim() == 2 && logits.size(1) <= 40960 && temperature > 0);
  const int64_t B = logits.size(0);
  TORCH_CHECK(out.numel() >= B && step.numel() >= B);
  evgpt::launch_sample_top_p(logits.data_ptr<float>(), out.data_ptr<int>(), (int)B,
                             (int)logits.size(1), (float)temperature, (int)std::max<int64_t>(top_k, 0), (float)top_p,
                             (unsigned long long)seed, step.data_ptr<int>(), cur_stream());
}

}  // namespace

TORCH_LIBRARY(evgpt, m) {
  m.def("rmsnorm(Tensor x, Tensor w, float eps) -> Tensor", &rmsnorm);
  m.def("add_rmsnorm(Tensor x, Tensor res, Tensor w, float eps) -> (Tensor, Tensor)", &add_rmsnorm);
  m.def("layernorm(Tensor x, Tensor w, Tensor? b, float eps, int act=0) -> Tensor", &layernorm);
  m.def("linear(Tensor x, Tensor w, Tensor? bias, Tensor? res, int act, bool out_f32) -> Tensor", &linear);
  m.def("linear_swiglu(Tensor x, Tensor w_gate_up) -> Tensor", &linear_swiglu);
  m.def("linear_partial(Tensor x, Tensor w, int max_split=0) -> Tensor", &linear_partial);
  m.def("lowrank_out(Tensor x, Tensor w, float alpha, float drop_p=0.0, int seed=0) -> Tensor", &lowrank_out);
  m.def("lowrank_mask(Tensor x, float drop_p, int seed) -> Tensor", &lowrank_mask);
  m.def("lowrank_mask_add_(Tensor(a!) acc, Tensor x, float drop_p, int seed) -> ()", &lowrank_mask_add_);
  m.def("linear_rows_(Tensor x, Tensor w, Tensor(a!) out, bool accumulate) -> ()", &linear_rows_);
  m.def("linear_t(Tensor a, Tensor w, bool a_t, bool w_t, Tensor? bias, Tensor? res, int act, bool out_f32, int K, Tensor(a!)? out=None) -> Tensor", &linear_t);
  m.def("gemv_max_batch() -> int", []() -> int64_t { return evgpt::gemv_max_batch(); });
  m.def("reduce_residual_layernorm(Tensor partial, Tensor? bias, Tensor? res, Tensor? ln_w, Tensor? ln_b, float eps) -> (Tensor, Tensor)", &reduce_residual_layernorm);
  m.def("reduce_residual_norm(Tensor partial, Tensor? res, Tensor? norm_w, float eps, Tensor(a!)? out) -> (Tensor, Tensor)", &reduce_residual_norm);
  m.def("reduce_swiglu(Tensor partial) -> Tensor", &reduce_swiglu);
  m.def("reduce_rope_kv(Tensor partial, Tensor(a!) q_out, Tensor(b!) k_cache, Tensor(c!) v_cache, Tensor cos, Tensor sin, Tensor pos0, int B, int S, int Hq, int Hkv, int D) -> ()", &reduce_rope_kv);
  m.def("swiglu(Tensor gate_up) -> Tensor", &swiglu);
  m.def("rope(Tensor x, Tensor positions, Tensor cos, Tensor sin) -> Tensor", &rope);
  m.def("rope_qkv_split(Tensor qkv, Tensor positions, Tensor cos, Tensor sin, int Hq, int Hkv, int D) -> Tensor[]", &rope_qkv_split);
  m.def("rope_qkv_merge(Tensor gq, Tensor gk, Tensor gv, Tensor positions, Tensor cos, Tensor sin) -> Tensor", &rope_qkv_merge);
  m.def("attention(Tensor q, Tensor k, Tensor v, bool causal, float scale, int q_offset) -> Tensor", &attention);
  m.def("attention_fwd_lse(Tensor q, Tensor k, Tensor v, bool causal, float scale, int q_offset) -> (Tensor, Tensor)", &attention_fwd_lse);
  m.def("attention_bwd(Tensor q, Tensor k, Tensor v, Tensor o, Tensor d_o, Tensor lse, bool causal, float scale, int q_offset) -> (Tensor, Tensor, Tensor)", &attention_bwd);
  m.def("attention_cache(Tensor q, Tensor k_cache, Tensor v_cache, Tensor(a!) out, Tensor kv_len, float scale) -> ()", &attention_cache);
  m.def("st_pool(Tensor f, int n_temporal) -> Tensor", &st_pool);
  m.def("argmax_rows(Tensor x) -> Tensor", &argmax_rows);
  m.def("row_metrics(Tensor a, Tensor b, float eps) -> (Tensor, Tensor)", &row_metrics);
  m.def("row_stats(Tensor a, Tensor b) -> Tensor", &row_stats);
  m.def("row_affine_grad(Tensor p, Tensor t, Tensor coef) -> Tensor", &row_affine_grad);
  m.def("decode_qkv(Tensor x, Tensor norm_w, Tensor w_qkv, Tensor(a!) q_out, Tensor(b!) k_cache, Tensor(c!) v_cache, Tensor cos, Tensor sin, Tensor pos, float eps, int Hq, int Hkv, int D, bool one_seq=False) -> ()", &decode_qkv);
  m.def("decode_attn(Tensor q, Tensor k_cache, Tensor v_cache, Tensor(a!) out, Tensor(b!) part_o, Tensor(c!) part_ml, Tensor pos, int nsplit, float scale) -> ()", &decode_attn);
  m.def("decode_gemv_residual(Tensor w, Tensor xin, Tensor(a!) y) -> ()", &decode_gemv_residual);
  m.def("rmsnorm_bwd(Tensor x, Tensor weight, Tensor grad, float eps, bool want_dw=False) -> (Tensor, Tensor)", &rmsnorm_bwd);
  m.def("swiglu_bwd(Tensor gate_up, Tensor grad) -> Tensor", &swiglu_bwd);
  m.def("ce_fwd(Tensor logits, Tensor labels, int ignore, Tensor? row_weight=None) -> (Tensor, Tensor, Tensor)", &ce_fwd);
  m.def("ce_bwd(Tensor logits, Tensor labels, Tensor lse, Tensor upstream, Tensor stat, int ignore, Tensor? row_weight=None) -> Tensor", &ce_bwd);
  m.def("rows_gather(Tensor src, Tensor idx) -> Tensor", &rows_gather);
  m.def("rows_scatter(Tensor g, Tensor idx, int R) -> Tensor", &rows_scatter);
  m.def("attn512_fwd(Tensor qkv, int H, bool causal, Tensor? kpm, float scale, float drop_p, int seed, bool save_probs) -> (Tensor, Tensor)", &attn512_fwd);
  m.def("attn512_bwd(Tensor qkv, Tensor d_out, Tensor probs, int H, float scale, float drop_p, int seed) -> Tensor", &attn512_bwd);
  m.def("attn512_mask(int B, int H, float drop_p, int seed, Tensor like) -> Tensor", &attn512_mask);
  m.def("gelu_fwd(Tensor x) -> Tensor", &gelu_fwd);
  m.def("gelu_bwd(Tensor x, Tensor grad) -> Tensor", &gelu_bwd);
  m.def("dropout_fwd(Tensor x, float p, int seed) -> Tensor", &dropout_fwd);
  m.def("layernorm_bwd(Tensor x, Tensor weight, Tensor grad, float eps, bool want_params=True) -> (Tensor, Tensor, Tensor)", &layernorm_bwd);
  m.def("sumsq_accumulate(Tensor x, Tensor(a!) acc) -> ()", &sumsq_accumulate);
  m.def("fused_adamw(Tensor grad, Tensor(a!) master, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, Tensor(d!) param_out, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float clip, Tensor? clip_dev=None) -> ()", &fused_adamw);
  m.def("w4_dequant(Tensor packed, Tensor absmax, Tensor code, Tensor(a!)? out=None) -> Tensor", &w4_dequant);
  m.def("gemv_w4_max_batch() -> int", []() { return (int64_t)evgpt::gemv_w4_max_batch(); });
  m.def("decode_qkv_w4(Tensor x, Tensor norm_w, Tensor packed, Tensor absmax, Tensor code, Tensor(a!) q_out, Tensor(b!) k_cache, Tensor(c!) v_cache, Tensor cos, Tensor sin, Tensor pos, float eps, int Hq, int Hkv, int D, bool one_seq=False) -> ()", &decode_qkv_w4);
  m.def("decode_gemv_residual_w4(Tensor packed, Tensor absmax, Tensor code, Tensor xin, Tensor(a!) y) -> ()", &decode_gemv_residual_w4);
  m.def("decode_mlp_up_w4(Tensor packed, Tensor absmax, Tensor code, Tensor x, Tensor norm_w, float eps, Tensor(a!) h) -> ()", &decode_mlp_up_w4);
  m.def("decode_gemv(Tensor w, Tensor xin, Tensor(a!) y, bool accumulate) -> ()", &decode_gemv);
  m.def("decode_mlp_up(Tensor w_gate_up, Tensor x, Tensor norm_w, float eps, Tensor(a!) h) -> ()", &decode_mlp_up);
  m.def("logits_partials(int N) -> int", &logits_partials);
  m.def("decode_logits(Tensor w, Tensor x, Tensor? norm_w, float eps, Tensor(a!)? logits, Tensor(b!) part_val, Tensor(c!) part_idx) -> ()", &decode_logits);
  m.def("commit_token(Tensor part_val, Tensor part_idx, int n_part, Tensor? sampled, Tensor embed, Tensor(a!) x, Tensor(b!) out_tokens, Tensor(c!) step, Tensor(d!) pos, Tensor(e!) done, Tensor(f!) cur_token, int eos, Tensor? active=None) -> ()", &commit_token);
  m.def("linear_argmax_partials(Tensor x, Tensor w, Tensor? bias) -> (Tensor, Tensor)", &linear_argmax_partials);
  m.def("event_raster(Tensor x, Tensor y, Tensor p, int n_slices, int H, int W, int mode) -> Tensor", &event_raster);
  m.def("event_voxel(Tensor x, Tensor y, Tensor p, Tensor t, int t0, int span, int bins, int H, int W, bool signed_mode, bool by_count) -> (Tensor, Tensor)", &event_voxel);
  m.def("event_voxel_ranges(Tensor x, Tensor y, Tensor p, Tensor bounds, int H, int W, bool signed_mode, int chunks=0) -> (Tensor, Tensor)", &event_voxel_ranges);
  m.def("clip_preprocess(Tensor img, Tensor hstart, Tensor hcount, Tensor hweight, Tensor vstart, Tensor vcount, Tensor vweight, int S, int resize_h, int resize_w, int top, int left, float[] mean, float[] stdv) -> Tensor", &clip_preprocess);
  m.def("gemm_force_config(int cfg) -> ()", &gemm_force_config);
  m.def("gemm_enable_256(bool on) -> ()", &gemm_enable_256);
  m.def("gemm_persist_256(bool on) -> ()", &gemm_persist_256);
  m.def("gemm_group_m(int g) -> ()", &gemm_group_m);
  m.def("gemm_split_major(int on) -> ()", &gemm_split_major);
  m.def("attn_force_groups(int groups) -> ()", &attn_force_groups);
  m.def("gemm_enable_tail_splitk(bool on) -> ()", &gemm_enable_tail_splitk);
  m.def("attn_enable_wide(int mode) -> ()", &attn_enable_wide);
  m.def("attn_wide4_min_rows(int rows) -> ()", &attn_wide4_min_rows);
  m.def("gemm_force_splitk(int sk) -> ()", &gemm_force_splitk);
  m.def("gemv_force_ks(int ks) -> ()", &gemv_force_ks);
  m.def("decode_gemv_partial(Tensor w, Tensor xin, Tensor(a!) out_f32) -> ()", &decode_gemv_partial);
  m.def("patchify(Tensor px, int P) -> Tensor", &patchify);
  m.def("vit_embed_ln(Tensor tok, Tensor cls, Tensor pos, Tensor w, Tensor b, float eps) -> Tensor", &vit_embed_ln);
  m.def("verify_accept(Tensor pval, Tensor pidx, Tensor draft, int G, Tensor(a!) out_tokens, Tensor(b!) step, Tensor(c!) pos, Tensor(d!) done, Tensor(e!) cur_token, Tensor(f!) result, int eos, Tensor embed, Tensor(g!) x, Tensor? limit=None, Tensor(h!)? stats=None) -> ()", &verify_accept);
  m.def("splice_rows(Tensor embed, Tensor feats, Tensor map) -> Tensor", &splice_rows);
  m.def("gather_rows_dev(Tensor src, Tensor index, int offset, Tensor? n_valid, int rows) -> Tensor", &gather_rows_dev);
  m.def("argmax_from_partials(Tensor pval, Tensor pidx, Tensor(a!) token, int vocab) -> ()", &argmax_from_partials);
  m.def("attn_d512(Tensor q, Tensor k, Tensor v, bool causal, float scale) -> Tensor", &attn_d512);
  m.def("naive_sd_sync(Tensor t_pos, Tensor t_step, Tensor t_cur, Tensor t_out, int offset, Tensor embed, Tensor prompt_last, Tensor(a!) rows, Tensor(b!) d_pos0, Tensor(c!) d_pos, Tensor(d!) d_step, Tensor(e!) d_done, int l_max) -> ()", &naive_sd_sync);
  m.def("sample_top_p(Tensor logits, Tensor(a!) out, float temperature, float top_p, int seed, Tensor step, int top_k=0) -> ()", &sample_top_p);
  m.def("linear_qkv_rope(Tensor x, Tensor w_qkv, Tensor(a!) q_out, Tensor(b!) k_cache, Tensor(c!) v_cache, Tensor cos, Tensor sin, Tensor pos0, int Hq, int Hkv) -> bool", &linear_qkv_rope);
  m.def("prefill_rope_kv(Tensor qkv, Tensor(a!) q_out, Tensor(b!) k_cache, Tensor(c!) v_cache, Tensor cos, Tensor sin, Tensor pos0, int Hq, int Hkv, int D) -> ()", &prefill_rope_kv);
}
