// Launcher declarations for the gfx950 kernels.  Launchers take raw device
// pointers and a stream, never allocate and never synchronise (graph-safe).
#pragma once
#include <hip/hip_runtime.h>

namespace evgpt {

typedef unsigned short bf16_t;

// ---- gemv.hip (decode, batch <= gemv_max_batch())
int gemv_max_batch();
int gemv_logits_grid(int N);
// 4-bit weights with a 16-entry bf16 codebook: lo[g] / hi[g] hold the low / high bytes
// of entries 4g..4g+3 (entry 4g in byte 0).  A launcher given a W4Desc reads `W` as
// [N][K/2] code bytes and `absmax` as [N][K/64] fp32 scales (K % 64 == 0).
struct W4Code { unsigned int lo[4], hi[4]; };
struct W4Desc { const float* absmax; W4Code code; };
int gemv_w4_max_batch();
void launch_w4_dequant(const unsigned char* packed, const float* absmax, const W4Code& code,
                       bf16_t* out, long long n_rows, int K, hipStream_t st);
void launch_gemv(const bf16_t* W, const bf16_t* x, bf16_t* y, const bf16_t* bias,
                 const bf16_t* res, const bf16_t* norm_w, float eps, int act, int B, int N,
                 int K, hipStream_t st, const W4Desc* w4 = nullptr, float* y_f32 = nullptr);
void gemv_force_ks(int ks);   // 0 = automatic; 1, 2 or 4 waves per 4-row unit (tuning runs)
void launch_gemv_swiglu(const bf16_t* W_gate_up, const bf16_t* x, bf16_t* y,
                        const bf16_t* norm_w, float eps, int B, int inter, int K, hipStream_t st,
                        const W4Desc* w4 = nullptr);
void launch_gemv_qkv_rope(const bf16_t* W_qkv, const bf16_t* x, const bf16_t* norm_w, float eps,
                          bf16_t* q_out, bf16_t* k_cache, bf16_t* v_cache, const float* cos_tab,
                          const float* sin_tab, const int* pos, int B, int n_q_heads,
                          int n_kv_heads, int head_dim, int l_max, int K, hipStream_t st,
                          int rows_one_seq = 0, const W4Desc* w4 = nullptr);
void launch_gemv_logits(const bf16_t* W, const bf16_t* x, const bf16_t* norm_w, float eps,
                        float* logits, float* part_val, int* part_idx, int B, int N, int K,
                        hipStream_t st);

// ---- train_bwd.hip
int rmsnorm_bwd_blocks(int rows, bool with_dw);   // rows of the [blocks][H] fp32 dw partial buffer
void launch_rmsnorm_bwd(const bf16_t* x, const bf16_t* w, const bf16_t* g, bf16_t* dx, float* dw, int rows,
                        int H, float eps, hipStream_t st);
void launch_swiglu_bwd(const bf16_t* gate_up, const bf16_t* gh, bf16_t* dgu, long long M, int I, hipStream_t st);

// ---- adapter_train.hip (training of the hidden-state adapters)
// q / k / v: element (b, s, h, d) at b*sb + s*ss + h*512 + d (e.g. the thirds of a packed in-projection);
// S <= 64; probs [B][H][64][64] bf16 (softmax before dropout; may be null in the forward); kpm [B][S]
// bytes, 1 = padded key, or null; drop_p in [0, 1); the dropout mask is a function of (seed, b, h, row, key)
void launch_attn512_fwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, long long sb, long long ss, bf16_t* out,
                        bf16_t* probs, const unsigned char* kpm, int B, int S, int H, int causal, float scale,
                        float drop_p, unsigned long long seed, hipStream_t st);
void launch_attn512_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, long long sb, long long ss,
                        const bf16_t* d_out, const bf16_t* probs, bf16_t* dq, bf16_t* dk, bf16_t* dv, int B, int S,
                        int H, float scale, float drop_p, unsigned long long seed, hipStream_t st);
void launch_attn512_mask(unsigned char* out, long long n, float drop_p, unsigned long long seed, hipStream_t st);
void launch_gelu(const bf16_t* x, bf16_t* y, long long n, hipStream_t st);                       // n % 8 == 0
void launch_gelu_bwd(const bf16_t* x, const bf16_t* g, bf16_t* dx, long long n, hipStream_t st);
void launch_dropout(const bf16_t* x, bf16_t* y, long long n, float p, unsigned long long seed, hipStream_t st);
int layernorm_bwd_blocks(int rows);    // rows of the [blocks][H] fp32 dw / db partial buffers
void launch_layernorm_bwd(const bf16_t* x, const bf16_t* w, const bf16_t* g, bf16_t* dx, float* dw, float* db,
                          int rows, int H, float eps, hipStream_t st);

// ---- loss.hip (token cross-entropy on bf16 logits; gather / scatter of the supervised rows)
// logits [M][ld] bf16, labels [M] (ignore, or anything outside [0, V), counts as unsupervised);
// lse / loss [M] fp32; stat [2] fp32 = {mean loss over supervised rows, 1 / their count}, or with row
// weights wrow [M] = {sum_i w_i loss_i, 1}
void launch_ce_fwd(const bf16_t* logits, long long ld, const int* labels, float* lse, float* loss, float* stat, int M,
                   int V, int ignore, hipStream_t st, const float* wrow = nullptr);
// dlogits [M][ldd] (ldd even, >= V; columns >= V are zeroed) = (softmax - onehot) * upstream[0] * stat[1] (* wrow[i])
void launch_ce_bwd(const bf16_t* logits, long long ld, const int* labels, const float* lse, const float* upstream,
                   const float* stat, bf16_t* dlogits, long long ldd, int M, int V, int ignore, hipStream_t st,
                   const float* wrow = nullptr);
void launch_rows_gather(const bf16_t* src, const int* idx, bf16_t* out, int n, int K, int n_src, hipStream_t st);
void launch_rows_scatter(const bf16_t* g, const int* idx, bf16_t* out, int n, int K, int R, hipStream_t st);

// ---- optim.hip
// clip_dev (optional): a device scalar multiplied into `clip` by the kernel itself, so a clip factor
// derived from the global gradient norm never has to visit the host
struct AdamWArgs {
  float clip, beta1, beta2, eps, decay /* 1 - lr*wd */, step_size /* lr / bc1 */, inv_sqrt_bc2;
  const float* clip_dev;
};
void launch_fused_adamw_bf16(const bf16_t* grad, float* master, float* m, float* v, bf16_t* out, long long n,
                             const AdamWArgs& a, hipStream_t st);
void launch_fused_adamw_f32(const float* grad, float* master, float* m, float* v, float* out, long long n,
                            const AdamWArgs& a, hipStream_t st);

int sumsq_blocks(long long n);
void launch_sumsq(const void* x, bool bf16, long long n, float* part, float* acc, hipStream_t st);

// ---- gemm.hip
int gemm_argmax_partials(int M, int N);
void gemm_force_config(int cfg);  // -1 = automatic; for tuning runs
void attn_force_groups(int g);      // key-tile groups per 64-row attention block: 0 by shape (default), 1, 2
void attn_wide4_min_rows(int rows); // fewest query rows that take the 128-row wide blocks
void attn_enable_wide(int mode);   // 128-query-row prefill attention: 0 never, 1 by shape (default), 2 always
void gemm_group_m(int g);          // row tiles per group of the 8-wave kernel's tile walk (0: whole columns)
void gemm_split_major(int on);     // split-K workgroups of the 4-wave kernel ordered split by split
void gemm_persist_256(bool on);   // the 8-wave kernel walks several tiles per workgroup (default on)
void gemm_enable_256(bool on);    // the 256 x 256 8-wave kernel for large GEMMs (default on)
void launch_gemm_bf16(const bf16_t* A, const bf16_t* W, bf16_t* C, float* Cf, const bf16_t* bias,
                      const bf16_t* res, int M, int N, int K, int lda, int ldw, int ldc, int act,
                      float* amax_val, int* amax_idx, float* splitk_ws, int splitk,
                      hipStream_t st);
void launch_gemm_glu(const bf16_t* A, const bf16_t* W, bf16_t* H, int M, int I, int K, int lda, int ldw, int ldc,
                     hipStream_t st);
bool launch_gemm_qkv_rope(const bf16_t* A, const bf16_t* W, int M, int K, bf16_t* q_out, bf16_t* k_cache,
                          bf16_t* v_cache, const float* cos_tab, const float* sin_tab, const int* pos0, int Hq,
                          int Hkv, int l_max, hipStream_t st);
int gemm_splitk_slabs(int M, int N, int K);   // K splits of launch_gemm_partial (slabs summed by the caller)
int gemm_splitk_t(int M, int N, int K);   // K splits for launch_gemm_bf16_t (1 = none)
void launch_gemm_bf16_t(const bf16_t* A, const bf16_t* W, bf16_t* C, float* Cf, const bf16_t* bias,
                        const bf16_t* res, int M, int N, int K, int lda, int ldw, int ldc, int act,
                        bool a_t, bool w_t, int ka_rows, int kw_rows, hipStream_t st, float* splitk_ws = nullptr, int splitk = 1);
// max_split > 0 caps the number of K slabs (1: one slab, e.g. a tensor-parallel rank whose
// contraction is already a slice and whose slab goes straight into a collective)
int launch_gemm_partial(const bf16_t* A, const bf16_t* W, float* partial, int M, int N, int K,
                        int lda, int ldw, hipStream_t st, int max_split = 0);
int gemm_splitk(int M, int N, int K);   // planned split count (1 = none); ws = splitk*M*N floats
long long gemm_tail_ws_floats(int M, int N, int K);   // unsplit GEMMs: workspace the last-round K split wants (0: none)
void gemm_enable_tail_splitk(bool on);
void gemm_force_splitk(int sk);         // 0 = automatic; for tuning runs

// ---- event_ops.hip
void launch_event_raster(const int* x, const int* y, const int* p, int n_events, int n_slices,
                         int H, int W, int* keys, unsigned char* img, int mode, hipStream_t st);
int event_voxel_band_rows(int H, int W);
void launch_event_voxel_ranges(const int* x, const int* y, const int* p, const int* bounds, int n_events,
                               int bins, int H, int W, int* out_hist, float* out_vox, int signed_mode,
                               hipStream_t st, int chunks_override = 0);
void launch_event_voxel(const int* x, const int* y, const int* p, const long long* t, int n_events,
                        long long t0, long long span, int bins, int H, int W, int* out_hist,
                        float* out_vox, int signed_mode, int by_count, hipStream_t st);
void launch_clip_preproc(const unsigned char* img, bf16_t* out, const int* hstart,
                         const int* hcount, const int* hweight, const int* vstart,
                         const int* vcount, const int* vweight, int T, int H, int W, int S,
                         int resize_h, int resize_w, int top, int left, int ksh, int ksv,
                         const float* mean, const float* stdv, hipStream_t st);

// ---- adapter_ops.hip (device-resident speculative-decoding draft head)
void launch_gather_rows_dev(const bf16_t* src, const int* index, int offset, const int* n_valid_dev,
                            int n_rows_src, bf16_t* out, int rows, int H, hipStream_t st);
void launch_argmax_from_partials(const float* pval, const int* pidx, int rows, int n_part, int* token,
                                 int vocab, hipStream_t st);
void launch_splice_rows(const bf16_t* embed, int vocab, const bf16_t* feats, int n_feat, const int* map,
                        int pad, bf16_t* out, long long rows, int H, hipStream_t st);
void launch_attn_d512(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, long long sb,
                      long long ss, int B, int S, int H, int causal, float scale, hipStream_t st);

// ---- sd.hip
void launch_verify_accept(const float* pval, const int* pidx, int n_part, const int* draft, int G,
                          int* out_tokens, int max_out, int* step, int* pos, int* done,
                          int* cur_token, int* result, int eos, const bf16_t* embed, bf16_t* x,
                          int H, int vocab, hipStream_t st, const int* limit = nullptr,
                          int* stats = nullptr);
void launch_sample_top_p(const float* logits, int* out, int B, int V, float temperature, int top_k,
                         float top_p, unsigned long long seed, const int* step,
                         hipStream_t st);
void launch_naive_sd_sync(const int* t_pos, const int* t_step, const int* t_cur, const int* t_out, int max_out,
                          int offset, const bf16_t* embed, int vocab, const bf16_t* prompt_last, bf16_t* rows,
                          int* d_pos0, int* d_pos, int* d_step, int* d_done, int H, int l_max, hipStream_t st);

// ---- elementwise.hip
void launch_rmsnorm(const bf16_t* x, const bf16_t* res, const bf16_t* w, bf16_t* y,
                    bf16_t* res_out, int rows, int H, float eps, hipStream_t st);
void launch_layernorm(const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, int rows,
                      int H, float eps, hipStream_t st, int act = 0);
void launch_rope_kv_prefill(const bf16_t* qkv, bf16_t* q_out, bf16_t* k_cache, bf16_t* v_cache,
                            const float* cos_tab, const float* sin_tab, const int* pos0, int B,
                            int S, int Hq, int Hkv, int D, int l_max, hipStream_t st);
// C [M, R] = alpha * A [M, K] · W [R, K]^T for R <= 16 and K % 8 == 0 (rank-r projection of a low-rank adapter)
void launch_lowrank_out(const bf16_t* A, const bf16_t* W, bf16_t* C, int M, int R, int K, long long lda, float alpha,
                        float drop_p, unsigned long long seed, hipStream_t st);
void launch_lowrank_mask(const bf16_t* x, bf16_t* y, long long n, float drop_p, unsigned long long seed, bool add,
                         hipStream_t st);
void launch_rope_qkv(const bf16_t* packed_in, bf16_t* packed_out, bf16_t* q, bf16_t* k, bf16_t* v,
                     const long long* positions, const float* cos_tab, const float* sin_tab, long long tokens,
                     int Hq, int Hkv, int D, int max_pos, bool merge, hipStream_t st);
void launch_rope(const bf16_t* x, bf16_t* y, const long long* positions, const float* cos_tab,
                 const float* sin_tab, long long tokens, int H, int D, hipStream_t st);
void launch_swiglu(const bf16_t* in, bf16_t* out, long long M, int I, hipStream_t st);
void launch_st_pool(const bf16_t* f, bf16_t* out, int T, int S, int C, int Tn, hipStream_t st);
void launch_argmax_rows_f32(const float* x, long long* out, int rows, int N, hipStream_t st);
void launch_argmax_rows_bf16(const bf16_t* x, long long* out, int rows, int N, hipStream_t st);
void launch_commit_token(const float* part_val, const int* part_idx, int n_part,
                         const int* sampled, const bf16_t* embed, bf16_t* x, int* out_tokens,
                         int max_out, int* step, int* pos, int* done, int* cur_token, int eos,
                         int B, int H, int vocab, hipStream_t st,
                         const int* active = nullptr);

void launch_patchify(const bf16_t* px, bf16_t* out, int T, int C, int S, int P, int Kp,
                     hipStream_t st);
void launch_vit_embed_ln(const bf16_t* tok, const bf16_t* cls, const bf16_t* pos, const bf16_t* w,
                         const bf16_t* b, bf16_t* out, int T, int N, int H, float eps,
                         hipStream_t st);

// ---- skinny.hip (consumers of split-K partials)
void launch_reduce_residual_norm(const float* partial, int nslab, int M, int N, const bf16_t* res,
                                 bf16_t* x_out, const bf16_t* norm_w, bf16_t* xn_out, float eps,
                                 hipStream_t st);
void launch_row_metrics_f32(const float* a, const float* b, float* cos_out, float* mse_out,
                            long long rows, int H, float eps, hipStream_t st, float* stats = nullptr);
void launch_row_metrics_bf16(const bf16_t* a, const bf16_t* b, float* cos_out, float* mse_out,
                             long long rows, int H, float eps, hipStream_t st, float* stats = nullptr);
void launch_row_affine_grad(const void* p, const void* t, const float* coef, void* grad, long long rows, int H,
                            bool bf16, hipStream_t st);
bool launch_reduce_residual_layernorm(const float* partial, int nslab, int M, int N, const bf16_t* bias,
                                      const bf16_t* res, bf16_t* x_out, const bf16_t* ln_w, const bf16_t* ln_b,
                                      bf16_t* y_out, float eps, hipStream_t st);
void launch_reduce_swiglu(const float* partial, int nslab, long long M, int I, bf16_t* out,
                          hipStream_t st);
void launch_reduce_rope_kv(const float* partial, int nslab, bf16_t* q_out, bf16_t* k_cache,
                           bf16_t* v_cache, const float* cos_tab, const float* sin_tab,
                           const int* pos0, int B, int S, int Hq, int Hkv, int D, int l_max,
                           hipStream_t st);

// ---- comm.hip (tensor-parallel collectives on peer-mapped buffers, vocab-parallel argmax)
constexpr int COMM_MAX_RANKS = 8;       // slots per receive buffer
constexpr int COMM_MAX_BLOCKS = 64;     // block regions per call
constexpr int COMM_CHUNK_MAX = 4096;    // floats per region (one row of a 4096-wide model)
// Receive buffer of one rank: float data[2][COMM_MAX_RANKS][COMM_MAX_BLOCKS][COMM_CHUNK_MAX]
// and unsigned flag[2][COMM_MAX_RANKS][COMM_MAX_BLOCKS], both uncached device memory.
constexpr long long COMM_DATA_FLOATS = 2ll * COMM_MAX_RANKS * COMM_MAX_BLOCKS * COMM_CHUNK_MAX;
constexpr long long COMM_FLAG_WORDS = 2ll * COMM_MAX_RANKS * COMM_MAX_BLOCKS;
struct CommState {            // lives in device memory, one per rank
  float* data[COMM_MAX_RANKS];      // data[p]: rank p's receive buffer (peer-mapped; [rank] is mine)
  unsigned* flag[COMM_MAX_RANKS];   // flag[p]: rank p's flag words
  unsigned* epoch;                  // [COMM_MAX_BLOCKS] calls completed per block region (local)
  unsigned* err;                    // [4]: timeouts, then epoch / peer / region of the last one (local)
  int rank, world;
  unsigned spin_limit;              // polls before a wait gives up
};
struct ARCall {
  const CommState* cs;
  const float* partial;       // [nslab][M][N] local partial sums
  long long slab_stride;      // M*N
  int nslab, M, N, chunk, bpr;
  const bf16_t* res;          // [M][N] or null
  bf16_t* x_out;              // [M][N] bf16(sum) + res, or null
  const bf16_t* norm_w;       // [N]: RMSNorm of x_out into xn_out (only when a block owns a row)
  bf16_t* xn_out;
  float* sum_out;             // optional fp32 [M][N] reduced sum
  float eps;
};
struct ARCallSet { ARCall c[COMM_MAX_RANKS]; };   // one per rank driven by this launch (1 in production)
int comm_plan(int M, int N, int* chunk, int* bpr);   // blocks of a call, 0 if it does not fit
void launch_oneshot_allreduce(const ARCallSet& calls, int n_virtual, int blocks, int nslab, int world,
                              hipStream_t st);
// Two-shot all-reduce of prefill-sized bf16 messages: reduce-scatter + all-gather, both as direct
// pushes into peer-mapped staging slots.  A block moves one PIECE (256 lanes x 16 bytes) of every
// rank's slice; staging and flags are uncached device memory, double-buffered by call parity.
constexpr int COMM2_PIECE = 2048;       // bf16 elements per block
struct Comm2State {           // device memory, one per rank
  bf16_t* stage1[COMM_MAX_RANKS];   // scatter slots  [2][world][slice_cap] of rank p (peer-mapped; [rank] is mine)
  bf16_t* stage2[COMM_MAX_RANKS];   // gather slots, same shape
  unsigned* flag[COMM_MAX_RANKS];   // [2 phases][2][world][pieces_cap]
  unsigned* epoch;                  // [pieces_cap] calls completed per piece (local)
  unsigned* err;                    // [4]: timeouts, then epoch / peer / piece of the last one (local)
  long long slice_cap;              // elements per (parity, source rank) slot: pieces_cap * COMM2_PIECE
  int pieces_cap;
  int rank, world;
  unsigned spin_limit;
};
struct TSCall {
  const Comm2State* cs;
  const bf16_t* x;            // [E] this rank's addend
  bf16_t* out;                // [E] the sum (may be x)
  long long E;                // elements, a multiple of 8
  int pieces;                 // pieces per slice: slice s = elements [s, s + 1) * pieces * COMM2_PIECE
};
struct TSCallSet { TSCall c[COMM_MAX_RANKS]; };
void launch_twoshot_allreduce(const TSCallSet& calls, int n_virtual, int pieces, int world, hipStream_t st);
void launch_sum_slabs(const float* partial, int nslab, long long n, float* out, hipStream_t st);
void launch_argmax_pair_local(const float* part_val, const int* part_idx, int B, int n_part,
                              int index_offset, unsigned* pair, hipStream_t st);
void launch_argmax_pair_select(const unsigned* pairs, int W, int B, int* token, hipStream_t st);

// ---- attn_decode.hip
void launch_attn_decode(const bf16_t* q, const bf16_t* k_cache, const bf16_t* v_cache,
                        bf16_t* out, float* part_o, float* part_ml, const int* pos, int B, int Hq,
                        int Hkv, int D, int l_max, int nsplit, float scale, hipStream_t st);

// ---- attn_prefill.hip
void launch_attn_prefill(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o,
                         const long long* strides, const int* kv_len, int B, int Sq, int Sk,
                         int Hq, int Hkv, int D, int causal, int q_offset, int q_offset_from_len,
                         float scale, hipStream_t st, float* part_o = nullptr,
                         float* part_ml = nullptr, int nsplit = 0, float* lse = nullptr);

// ---- attn_bwd.hip: flash-attention backward (recomputes P from the forward's LSE)
// q/do/dq [B,Sq,Hq,D], k/v/dk/dv [B,Sk,Hkv,D], o [B,Sq,Hq,D], all contiguous bf16;
// lse [B,Hq,Sq] from the forward; delta [B,Hq,Sq] scratch.
void launch_attn_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* o, const bf16_t* d_o,
                     const float* lse, float* delta, bf16_t* dq, bf16_t* dk, bf16_t* dv, int B, int Sq,
                     int Sk, int Hq, int Hkv, int D, int causal, int q_offset, float scale, hipStream_t st);

}  // namespace evgpt
