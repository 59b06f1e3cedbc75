// Peer-mapped communication buffers for the one-shot and two-shot all-reduces (comm.hip).
//
// Each rank allocates its receive buffer and flag words as UNCACHED device memory
// (hipExtMallocWithFlags + hipDeviceMallocUncached: remote posted writes and local polls
// are coherent without cache maintenance on either agent), exports them with
// hipIpcGetMemHandle, and maps every peer's buffers with hipIpcOpenMemHandle.  The 64-byte
// handles travel through the process group as plain byte tensors (Python side,
// parallel/comm.py).  Everything here runs once at engine construction: nothing in this
// file is called from a captured region.
#include <cstring>
#include <vector>

#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime.h>
#include <torch/library.h>

#include "evgpt_kernels.h"

namespace {

using at::Tensor;

#define EVGPT_HIP_CHECK(expr)                                                              \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    TORCH_CHECK(e_ == hipSuccess, #expr, " failed: ", hipGetErrorString(e_));             \
  } while (0)

// zero-filled uncached device buffer of `nbytes` bytes (uint8 tensor that owns it)
Tensor comm_alloc(int64_t nbytes, int64_t device) {
  TORCH_CHECK(nbytes > 0 && nbytes % 16 == 0, "comm_alloc: size must be a positive multiple of 16");
  int prev = 0;
  EVGPT_HIP_CHECK(hipGetDevice(&prev));
  EVGPT_HIP_CHECK(hipSetDevice((int)device));
  void* p = nullptr;
  EVGPT_HIP_CHECK(hipExtMallocWithFlags(&p, (size_t)nbytes, hipDeviceMallocUncached));
  EVGPT_HIP_CHECK(hipMemset(p, 0, (size_t)nbytes));
  EVGPT_HIP_CHECK(hipDeviceSynchronize());
  EVGPT_HIP_CHECK(hipSetDevice(prev));
  auto opts = at::TensorOptions().dtype(at::kByte).device(at::kCUDA, (c10::DeviceIndex)device);
  return at::from_blob(p, {nbytes}, [](void* q) { (void)hipFree(q); }, opts);
}

// 64-byte IPC handle of a buffer from comm_alloc, as a host uint8 tensor
Tensor comm_ipc_handle(const Tensor& buf) {
  TORCH_CHECK(buf.is_cuda() && buf.scalar_type() == at::kByte && buf.is_contiguous());
  hipIpcMemHandle_t h;
  EVGPT_HIP_CHECK(hipIpcGetMemHandle(&h, buf.data_ptr()));
  Tensor out = at::empty({(int64_t)sizeof(h)}, at::kByte);
  std::memcpy(out.data_ptr(), &h, sizeof(h));
  return out;
}

// maps a peer's buffer into this process; returns the device address
int64_t comm_ipc_open(const Tensor& handle, int64_t device) {
  TORCH_CHECK(!handle.is_cuda() && handle.scalar_type() == at::kByte &&
              handle.numel() == (int64_t)sizeof(hipIpcMemHandle_t) && handle.is_contiguous());
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle.data_ptr(), sizeof(h));
  int prev = 0;
  EVGPT_HIP_CHECK(hipGetDevice(&prev));
  EVGPT_HIP_CHECK(hipSetDevice((int)device));
  void* p = nullptr;
  EVGPT_HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  EVGPT_HIP_CHECK(hipSetDevice(prev));
  return (int64_t)(uintptr_t)p;
}

void comm_ipc_close(int64_t ptr) {
  if (ptr) (void)hipIpcCloseMemHandle((void*)(uintptr_t)ptr);
}

std::tuple<int64_t, int64_t, int64_t> comm_plan(int64_t M, int64_t N) {
  int chunk = 0, bpr = 0;
  const int blocks = evgpt::comm_plan((int)M, (int)N, &chunk, &bpr);
  return {blocks, chunk, bpr};
}

int64_t comm_data_bytes() { return evgpt::COMM_DATA_FLOATS * 4; }
int64_t comm_flag_bytes() { return evgpt::COMM_FLAG_WORDS * 4; }

// Device-resident CommState of one rank.  data_ptrs / flag_ptrs: W device addresses (this
// rank's own buffers at index `rank`); epoch: int32 [COMM_MAX_BLOCKS]; err: int32 [4].
Tensor comm_make_state(std::vector<int64_t> data_ptrs, std::vector<int64_t> flag_ptrs, const Tensor& epoch,
                       const Tensor& err, int64_t rank, int64_t world, int64_t spin_limit) {
  TORCH_CHECK(world >= 1 && world <= evgpt::COMM_MAX_RANKS && rank >= 0 && rank < world,
              "comm_make_state: world must be 1..", evgpt::COMM_MAX_RANKS);
  TORCH_CHECK((int64_t)data_ptrs.size() == world && (int64_t)flag_ptrs.size() == world);
  TORCH_CHECK(epoch.is_cuda() && epoch.scalar_type() == at::kInt && epoch.is_contiguous() &&
              epoch.numel() >= evgpt::COMM_MAX_BLOCKS, "comm_make_state: epoch int32 [64]");
  TORCH_CHECK(err.is_cuda() && err.scalar_type() == at::kInt && err.is_contiguous() && err.numel() >= 4);
  TORCH_CHECK(spin_limit >= 1 && spin_limit < (1ll << 31));
  evgpt::CommState s{};
  for (int64_t p = 0; p < world; ++p) {
    TORCH_CHECK(data_ptrs[p] != 0 && flag_ptrs[p] != 0 && (data_ptrs[p] & 15) == 0, "comm_make_state: null/unaligned peer buffer");
    s.data[p] = reinterpret_cast<float*>((uintptr_t)data_ptrs[p]);
    s.flag[p] = reinterpret_cast<unsigned*>((uintptr_t)flag_ptrs[p]);
  }
  s.epoch = reinterpret_cast<unsigned*>(epoch.data_ptr<int>());
  s.err = reinterpret_cast<unsigned*>(err.data_ptr<int>());
  s.rank = (int)rank; s.world = (int)world; s.spin_limit = (unsigned)spin_limit;
  Tensor out = at::empty({(int64_t)sizeof(s)}, epoch.options().dtype(at::kByte));
  EVGPT_HIP_CHECK(hipMemcpy(out.data_ptr(), &s, sizeof(s), hipMemcpyHostToDevice));
  return out;
}

const evgpt::bf16_t* bp(const c10::optional<Tensor>& t) {
  return t.has_value() && t->defined() ? reinterpret_cast<const evgpt::bf16_t*>(t->data_ptr()) : nullptr;
}

void check_bf(const Tensor& t, int64_t n, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kBFloat16 && t.is_contiguous() && t.numel() == n &&
              (reinterpret_cast<uintptr_t>(t.data_ptr()) & 7) == 0, name, ": contiguous bf16 of ", n, " elements");
}

// All-reduce of partial [nslab, M, N] fp32 over the ranks of `state`, fused with
//   x_out = bf16(sum) + res      and, when norm_w is given and a block owns a row,
//   xn_out = RMSNorm(x_out) * norm_w.
// One list entry per rank driven by THIS launch: one in production (each process launches
// its own rank), several when one process plays every rank of a virtual group (tests).
// Returns false (and launches nothing) if the message does not fit the buffers.
bool comm_allreduce(std::vector<Tensor> state, std::vector<Tensor> partial,
                    std::vector<c10::optional<Tensor>> res, std::vector<c10::optional<Tensor>> x_out,
                    const c10::optional<Tensor>& norm_w, std::vector<c10::optional<Tensor>> xn_out,
                    std::vector<c10::optional<Tensor>> sum_out, double eps, int64_t world) {
  TORCH_CHECK(world >= 1 && world <= evgpt::COMM_MAX_RANKS, "comm_allreduce: world size");
  const int64_t V = (int64_t)state.size();
  TORCH_CHECK(V >= 1 && V <= evgpt::COMM_MAX_RANKS && (int64_t)partial.size() == V && (int64_t)res.size() == V &&
              (int64_t)x_out.size() == V && (int64_t)xn_out.size() == V && (int64_t)sum_out.size() == V,
              "comm_allreduce: one entry per driven rank");
  const Tensor& p0 = partial[0];
  TORCH_CHECK(p0.dim() == 3, "comm_allreduce: partial must be [nslab, M, N]");
  const int64_t ns = p0.size(0), M = p0.size(1), N = p0.size(2);
  int chunk = 0, bpr = 0;
  const int blocks = evgpt::comm_plan((int)M, (int)N, &chunk, &bpr);
  if (blocks == 0) return false;
  if (norm_w.has_value()) check_bf(*norm_w, N, "norm_w");
  evgpt::ARCallSet calls{};
  for (int64_t v = 0; v < V; ++v) {
    TORCH_CHECK(state[v].is_cuda() && state[v].scalar_type() == at::kByte &&
                state[v].numel() == (int64_t)sizeof(evgpt::CommState), "comm_allreduce: bad state");
    const Tensor& p = partial[v];
    TORCH_CHECK(p.is_cuda() && p.scalar_type() == at::kFloat && p.is_contiguous() && p.sizes() == p0.sizes() &&
                (reinterpret_cast<uintptr_t>(p.data_ptr()) & 15) == 0, "comm_allreduce: partials must be equal-shaped fp32");
    evgpt::ARCall& c = calls.c[v];
    c.cs = reinterpret_cast<const evgpt::CommState*>(state[v].data_ptr());
    c.partial = p.data_ptr<float>();
    c.slab_stride = M * N; c.nslab = (int)ns; c.M = (int)M; c.N = (int)N; c.chunk = chunk; c.bpr = bpr;
    c.eps = (float)eps;
    if (res[v].has_value()) { check_bf(*res[v], M * N, "res"); c.res = bp(res[v]); }
    if (x_out[v].has_value()) { check_bf(*x_out[v], M * N, "x_out"); c.x_out = const_cast<evgpt::bf16_t*>(bp(x_out[v])); }
    if (xn_out[v].has_value() && norm_w.has_value()) {
      TORCH_CHECK(bpr == 1 && x_out[v].has_value(), "comm_allreduce: the fused norm needs whole rows per block (N <= ",
                  evgpt::COMM_CHUNK_MAX, ", M >= 4) and x_out");
      check_bf(*xn_out[v], M * N, "xn_out");
      c.norm_w = bp(norm_w); c.xn_out = const_cast<evgpt::bf16_t*>(bp(xn_out[v]));
    }
    if (sum_out[v].has_value()) {
      const Tensor& s = *sum_out[v];
      TORCH_CHECK(s.is_cuda() && s.scalar_type() == at::kFloat && s.is_contiguous() && s.numel() == M * N);
      c.sum_out = s.data_ptr<float>();
    }
  }
  evgpt::launch_oneshot_allreduce(calls, (int)V, blocks, (int)ns, (int)world,
                                  c10::hip::getCurrentHIPStream().stream());
  return true;
}

// Device-resident Comm2State of one rank (two-shot all-reduce).  stage1 / stage2 / flag: W device
// addresses each (this rank's own buffers at index `rank`); every stage buffer holds
// 2 * world * pieces_cap * COMM2_PIECE bf16, every flag buffer 2 * 2 * world * pieces_cap words.
Tensor comm2_make_state(std::vector<int64_t> stage1, std::vector<int64_t> stage2, std::vector<int64_t> flags,
                        const Tensor& epoch, const Tensor& err, int64_t pieces_cap, int64_t rank, int64_t world,
                        int64_t spin_limit) {
  TORCH_CHECK(world >= 1 && world <= evgpt::COMM_MAX_RANKS && rank >= 0 && rank < world, "comm2_make_state: world size");
  TORCH_CHECK((int64_t)stage1.size() == world && (int64_t)stage2.size() == world && (int64_t)flags.size() == world);
  TORCH_CHECK(pieces_cap >= 1 && pieces_cap <= (1 << 20), "comm2_make_state: pieces_cap");
  TORCH_CHECK(epoch.is_cuda() && epoch.scalar_type() == at::kInt && epoch.is_contiguous() && epoch.numel() >= pieces_cap,
              "comm2_make_state: epoch int32 [pieces_cap]");
  TORCH_CHECK(err.is_cuda() && err.scalar_type() == at::kInt && err.is_contiguous() && err.numel() >= 4);
  TORCH_CHECK(spin_limit >= 1 && spin_limit < (1ll << 31));
  evgpt::Comm2State s{};
  for (int64_t p = 0; p < world; ++p) {
    TORCH_CHECK(stage1[p] != 0 && stage2[p] != 0 && flags[p] != 0 && (stage1[p] & 15) == 0 && (stage2[p] & 15) == 0,
                "comm2_make_state: null/unaligned peer buffer");
    s.stage1[p] = reinterpret_cast<evgpt::bf16_t*>((uintptr_t)stage1[p]);
    s.stage2[p] = reinterpret_cast<evgpt::bf16_t*>((uintptr_t)stage2[p]);
    s.flag[p] = reinterpret_cast<unsigned*>((uintptr_t)flags[p]);
  }
  s.epoch = reinterpret_cast<unsigned*>(epoch.data_ptr<int>());
  s.err = reinterpret_cast<unsigned*>(err.data_ptr<int>());
  s.pieces_cap = (int)pieces_cap;
  s.slice_cap = pieces_cap * evgpt::COMM2_PIECE;
  s.rank = (int)rank; s.world = (int)world; s.spin_limit = (unsigned)spin_limit;
  Tensor out = at::empty({(int64_t)sizeof(s)}, epoch.options().dtype(at::kByte));
  EVGPT_HIP_CHECK(hipMemcpy(out.data_ptr(), &s, sizeof(s), hipMemcpyHostToDevice));
  return out;
}

int64_t comm2_piece() { return evgpt::COMM2_PIECE; }

// Two-shot all-reduce of equal-shaped bf16 tensors, one list entry per rank driven by THIS launch
// (one in production).  out[v] may be x[v].  Returns false (and launches nothing) if the message
// does not fit `pieces_cap` pieces per slice.
bool comm2_allreduce(std::vector<Tensor> state, std::vector<Tensor> x, std::vector<Tensor> out, int64_t pieces_cap,
                     int64_t world) {
  TORCH_CHECK(world >= 2 && world <= evgpt::COMM_MAX_RANKS, "comm2_allreduce: 2..8 ranks");
  const int64_t V = (int64_t)state.size();
  TORCH_CHECK(V >= 1 && V <= evgpt::COMM_MAX_RANKS && (int64_t)x.size() == V && (int64_t)out.size() == V,
              "comm2_allreduce: one entry per driven rank");
  const int64_t E = x[0].numel();
  TORCH_CHECK(E >= 8 && E % 8 == 0, "comm2_allreduce: a multiple of 8 elements");
  const int64_t per_rank = (E + world - 1) / world;
  const int64_t pieces = (per_rank + evgpt::COMM2_PIECE - 1) / evgpt::COMM2_PIECE;
  if (pieces > pieces_cap) return false;
  evgpt::TSCallSet calls{};
  for (int64_t v = 0; v < V; ++v) {
    TORCH_CHECK(state[v].is_cuda() && state[v].scalar_type() == at::kByte &&
                state[v].numel() == (int64_t)sizeof(evgpt::Comm2State), "comm2_allreduce: bad state");
    check_bf(x[v], E, "x");
    check_bf(out[v], E, "out");
    TORCH_CHECK((reinterpret_cast<uintptr_t>(x[v].data_ptr()) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(out[v].data_ptr()) & 15) == 0, "comm2_allreduce: 16-byte aligned tensors");
    evgpt::TSCall& c = calls.c[v];
    c.cs = reinterpret_cast<const evgpt::Comm2State*>(state[v].data_ptr());
    c.x = reinterpret_cast<const evgpt::bf16_t*>(x[v].data_ptr());
    c.out = reinterpret_cast<evgpt::bf16_t*>(out[v].data_ptr());
    c.E = E; c.pieces = (int)pieces;
  }
  evgpt::launch_twoshot_allreduce(calls, (int)V, (int)pieces, (int)world, c10::hip::getCurrentHIPStream().stream());
  return true;
}

Tensor sum_slabs(const Tensor& partial) {
  TORCH_CHECK(partial.is_cuda() && partial.scalar_type() == at::kFloat && partial.is_contiguous() &&
              partial.dim() == 3 && (partial.size(1) * partial.size(2)) % 4 == 0, "sum_slabs: fp32 [nslab, M, N]");
  Tensor out = at::empty({1, partial.size(1), partial.size(2)}, partial.options());
  evgpt::launch_sum_slabs(partial.data_ptr<float>(), (int)partial.size(0), partial.size(1) * partial.size(2),
                          out.data_ptr<float>(), c10::hip::getCurrentHIPStream().stream());
  return out;
}

// (val [B,P], idx [B,P]) argmax partials of a vocabulary shard -> pair [B,2] int32 words
void argmax_pair_local(const Tensor& part_val, const Tensor& part_idx, int64_t B, int64_t n_part,
                       int64_t index_offset, const Tensor& pair) {
  TORCH_CHECK(part_val.is_cuda() && part_val.scalar_type() == at::kFloat && part_val.is_contiguous() &&
              part_idx.is_cuda() && part_idx.scalar_type() == at::kInt && part_idx.is_contiguous() &&
              part_val.numel() >= B * n_part && part_idx.numel() >= B * n_part && n_part >= 1);
  TORCH_CHECK(pair.is_cuda() && pair.scalar_type() == at::kInt && pair.is_contiguous() && pair.numel() >= 2 * B);
  evgpt::launch_argmax_pair_local(part_val.data_ptr<float>(), part_idx.data_ptr<int>(), (int)B, (int)n_part,
                                  (int)index_offset, reinterpret_cast<unsigned*>(pair.data_ptr<int>()),
                                  c10::hip::getCurrentHIPStream().stream());
}

void argmax_pair_select(const Tensor& pairs, int64_t W, int64_t B, const Tensor& token) {
  TORCH_CHECK(pairs.is_cuda() && pairs.scalar_type() == at::kInt && pairs.is_contiguous() && pairs.numel() >= W * B * 2);
  TORCH_CHECK(token.is_cuda() && token.scalar_type() == at::kInt && token.is_contiguous() && token.numel() >= B);
  evgpt::launch_argmax_pair_select(reinterpret_cast<const unsigned*>(pairs.data_ptr<int>()), (int)W, (int)B,
                                   token.data_ptr<int>(), c10::hip::getCurrentHIPStream().stream());
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(evgpt, m) {
  m.def("comm_alloc(int nbytes, int device) -> Tensor", &comm_alloc);
  m.def("comm_ipc_handle(Tensor buf) -> Tensor", &comm_ipc_handle);
  m.def("comm_ipc_open(Tensor handle, int device) -> int", &comm_ipc_open);
  m.def("comm_ipc_close(int ptr) -> ()", &comm_ipc_close);
  m.def("comm_plan(int M, int N) -> (int, int, int)", &comm_plan);
  m.def("comm_data_bytes() -> int", &comm_data_bytes);
  m.def("comm_flag_bytes() -> int", &comm_flag_bytes);
  m.def("comm_make_state(int[] data_ptrs, int[] flag_ptrs, Tensor epoch, Tensor err, int rank, int world, int spin_limit) -> Tensor", &comm_make_state);
  m.def("comm_allreduce(Tensor[] state, Tensor[] partial, Tensor?[] res, Tensor?[] x_out, Tensor? norm_w, Tensor?[] xn_out, Tensor?[] sum_out, float eps, int world) -> bool", &comm_allreduce);
  m.def("comm2_make_state(int[] stage1, int[] stage2, int[] flags, Tensor epoch, Tensor err, int pieces_cap, int rank, int world, int spin_limit) -> Tensor", &comm2_make_state);
  m.def("comm2_piece() -> int", &comm2_piece);
  m.def("comm2_allreduce(Tensor[] state, Tensor[] x, Tensor(a!)[] out, int pieces_cap, int world) -> bool", &comm2_allreduce);
  m.def("sum_slabs(Tensor partial) -> Tensor", &sum_slabs);
  m.def("argmax_pair_local(Tensor part_val, Tensor part_idx, int B, int n_part, int index_offset, Tensor(a!) pair) -> ()", &argmax_pair_local);
  m.def("argmax_pair_select(Tensor pairs, int W, int B, Tensor(a!) token) -> ()", &argmax_pair_select);
}
