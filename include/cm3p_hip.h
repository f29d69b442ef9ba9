/*
 * cm3p_hip.h - C ABI of libcm3p_hip.so: the MI355X (gfx950) kernels behind CM3P's contrastive training hot path.
 *
 * The reference (OliBomby/CM3P) is pure Python: it has no FFI, and its hot path runs inside PyTorch ops called from
 * ref:cm3p/modeling_cm3p.py and the third-party encoder TF:models/modernbert/modeling_modernbert.py (TF: = the
 * `transformers` package the reference depends on).  Each entry point below therefore names the reference Python
 * call it replaces.  The Python binding that a maintainer adds is `cm3p_amd/_lib.py` (ctypes); see INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, dtype codes; `stream` is a hipStream_t passed as void*.
 *   - every function returns CM3P_OK (0) or a negative CM3P_ERR_* code; nothing throws, nothing allocates,
 *     nothing synchronises; workspaces are passed in by the caller; no global state.
 *   - all pointers are device pointers, 16-byte aligned; "bf16" buffers are raw uint16 bit patterns.
 *   - activations are row-major [tokens, features]; tokens = batch * seq.
 *   - a function may be called from any host thread as long as the caller owns the stream.
 */
#ifndef CM3P_HIP_H
#define CM3P_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CM3P_OK 0
#define CM3P_ERR_INVALID (-1) /* bad argument: null pointer, unsupported shape, misaligned buffer */
#define CM3P_ERR_LAUNCH (-2)  /* the HIP runtime refused the launch */

#define CM3P_F32 0
#define CM3P_BF16 1

/* ABI version of this header; cm3p_abi_version() must return it. */
#define CM3P_ABI_VERSION 16
int cm3p_abi_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * LayerNorm without bias: y = (x - mean) / sqrt(var + eps) * weight.
 * Replaces nn.LayerNorm(H, eps, bias=False) at TF:models/modernbert/modeling_modernbert.py:61,312,314,420.
 * x: [rows, H] (fp32 or bf16).  Writes y_f32 and/or y_bf16 (either may be NULL), mean/rstd [rows] (may be NULL).
 * H % 4 == 0, H <= 2048.
 */
int cm3p_layernorm_fwd(const void* x, int x_dtype, const float* weight, float* y_f32, void* y_bf16, float* mean,
                       float* rstd, int64_t rows, int H, float eps, void* stream);

/* Number of rows of `dw_partial` ([blocks, H] fp32 workspace) that cm3p_layernorm_bwd / cm3p_embed_ln_bwd need. */
int cm3p_layernorm_bwd_blocks(int64_t rows);

/* Backward of the above (autograd of the same nn.LayerNorm).  dx = dres + LN'(dy) where `dres` (may be NULL) is the
 * gradient already flowing on the residual stream: the pre-norm residual x + f(LN(x)) of
 * TF:...modeling_modernbert.py:331-332.  dx_f32 may alias dres.  dw[H] is fully overwritten. */
int cm3p_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* weight, const float* mean,
                       const float* rstd, const float* dres, float* dx_f32, void* dx_bf16, float* dw_partial, float* dw,
                       int64_t rows, int H, void* stream);

/* Token embedding lookup + optional audio-embedding scatter + LayerNorm.
 * Replaces ModernBertEmbeddings.forward (TF:...modeling_modernbert.py:64-71) fed by
 * CM3PBeatmapTransformer.forward's `inputs_embeds[input_ids == audio_token_id] = audio_embeds`
 * (ref:cm3p/modeling_cm3p.py:592,603-605).  slot[t] >= 0 selects override_rows[slot[t]] instead of table[ids[t]];
 * slot / override_rows may both be NULL.  vocab = rows of `table`: an id outside [0, vocab) reads as a zero row and receives no
 * gradient (nn.Embedding raises a device-side assert there; these kernels must not fault). */
int cm3p_embed_ln_fwd(const int64_t* ids, const void* table, int table_dtype, const int32_t* slot,
                      const void* override_rows, int override_dtype, const float* weight, float* y_f32, void* y_bf16,
                      float* mean, float* rstd, int64_t T, int H, float eps, int64_t vocab, void* stream);

/* Backward: d_table[ids[t]] += row gradient (fp32 atomics; the caller zeroes d_table; row `padding_idx` gets none,
 * as nn.Embedding(padding_idx=...) does), d_override[slot[t]] = row gradient.  Either may be NULL. */
int cm3p_embed_ln_bwd(const float* dy, const int64_t* ids, const void* table, int table_dtype, const int32_t* slot,
                      const void* override_rows, int override_dtype, const float* weight, const float* mean,
                      const float* rstd, float* d_table, float* d_override, float* dw_partial, float* dw, int64_t T, int H,
                      int64_t padding_idx, int64_t vocab, void* stream);

/* The same backward without atomics (the default of the Python host): the caller sorts the tokens by id and the kernels visit them
 * in that order, so every sum has a fixed order - the embedding gradient is reproducible bit for bit - and tokens that share an
 * id do not serialise on one row of d_table.
 *   order  [T] int64: token indices sorted by ids[.] ascending, STABLE (ascending token index inside an id);
 *   run_of [T] int32: run number of sorted position p; a new run starts at p = 0, wherever ids[order[p]] != ids[order[p - 1]] and
 *                     at every multiple of cm3p_embed_ln_bwd_sorted_chunk() (64), i.e. run_of = cumsum(start flags) - 1;
 *   run_rows [R, H] fp32 and run_ids [R] int64: workspaces with R >= run_of[T - 1] + 1 (never more than min(T, vocab + T / 64 + 1));
 *   dw_partial: [ceil(ceil(T / 64) / 4), H] fp32.  d_table [vocab, H] is fully written (no zeroing by the caller). */
int cm3p_embed_ln_bwd_sorted_chunk(void);
int cm3p_embed_ln_bwd_sorted(const float* dy, const int64_t* ids, const int64_t* order, const int32_t* run_of, const void* table,
                             int table_dtype, const int32_t* slot, const void* override_rows, int override_dtype, const float* weight,
                             const float* mean, const float* rstd, float* d_table, float* d_override, float* run_rows, int64_t* run_ids,
                             float* dw_partial, float* dw, int64_t T, int H, int64_t padding_idx, int64_t vocab, void* stream);

/* The two index arrays cm3p_embed_ln_bwd_sorted takes, made on the device from the ids alone (a stable counting sort; replaces the
 * torch.sort / cumsum sequence of r03 - integer bookkeeping with no counterpart in the reference, whose nn.Embedding backward
 * (TF:models/modernbert/modeling_modernbert.py:64-71) scatters with atomics): order[T] int64 and run_of[T] int32 exactly as specified
 * there, keys = clamp(ids, -1, vocab).  workspace: cm3p_token_order_workspace_ints(T, vocab) int32 values; that query returns 0 when
 * the vocabulary is too large for the kernel's per-block histogram (vocab + 2 > 12288): the caller then sorts by other means. */
int64_t cm3p_token_order_workspace_ints(int64_t T, int64_t vocab);
int cm3p_token_order(const int64_t* ids, int64_t T, int64_t vocab, int64_t* order, int32_t* run_of, int32_t* workspace, void* stream);

/* slot[t] = rank of token t among the tokens equal to audio_token_id, in row-major (b, s) order, else -1;
 * count[0] = how many there are.  The integer side of ref:cm3p/modeling_cm3p.py:604-605 (bit-exact). */
int cm3p_audio_slots(const int64_t* ids, int64_t T, int64_t audio_token_id, int32_t* slot, int32_t* count, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM with fp32 accumulation:  C[m, n] = sum_k A(m, k) * B(n, k)   (+ R[m, n])
 * Replaces the bias-free nn.Linear calls Wqkv / Wo / Wi / Wo (TF:...modeling_modernbert.py:84,87,246,259,271,300,90-91)
 * and their autograd: forward y = x W^T (a_kc=1, b_kc=1), input gradient dx = dy W (a_kc=1, b_kc=0) and
 * weight gradient dW = dy^T x (a_kc=0, b_kc=0).
 *   a_kc = 1: A is [M, lda] with k contiguous;   a_kc = 0: A is [K, lda] with m contiguous (element (m,k) at A[k*lda+m]).
 *   b_kc likewise for B over (n, k).
 *   epilogue: CM3P_EPI_BF16 (C bf16), CM3P_EPI_F32 (C fp32), CM3P_EPI_F32_RESID (C fp32 = R + acc; R fp32 [M, ldc], may
 *   alias C), CM3P_EPI_F32_BIAS (C fp32 = acc + R[n]; R fp32 [N]: nn.Linear's bias added while the tile is stored - the
 *   decoder of the MLM head, ref:cm3p/modeling_cm3p.py:767,991; forward orientation a_kc = b_kc = 1 only).
 *   Constraints: contiguous extents and leading dimensions are multiples of 8 elements.
 */
#define CM3P_EPI_BF16 0
#define CM3P_EPI_F32 1
#define CM3P_EPI_F32_RESID 2
#define CM3P_EPI_F32_BIAS 5
/* split_k > 1 (CM3P_EPI_F32 only, ldc == N): the contraction is cut into split_k ranges whose fp32 partial tiles go to
 * `workspace` (split_k * M * N floats) and are then summed in a fixed order - used for dW, whose contraction runs over
 * all tokens while its output is only a few hundred tiles. */
int cm3p_gemm_bf16(const void* A, const void* B, void* C, const float* R, int64_t M, int64_t N, int64_t K, int64_t lda,
                   int64_t ldb, int64_t ldc, int a_kc, int b_kc, int epilogue, int split_k, float* workspace, void* stream);

/* Host-only, process-wide.  The big-shape GEMM (csrc/gemm8p.hip) is a persistent kernel: by default one 512-thread workgroup per CU walks the
 * work items with a fixed stride.  workgroups > 0 launches that many instead (clamped to the number of work items): with MORE workgroups than
 * CUs the surplus is dispatched wherever a CU comes free, which is what a data-parallel step wants while RCCL's channel workgroups hold
 * CUs (the reference gets the same from DistributedDataParallel over NCCL, ref:train.py:360-375: its GEMMs are not persistent) - measured
 * with 32 CUs held: +26..58 % per GEMM with one workgroup per CU, +7..10 % with 1024.  Safe for any value: no
 * gemm8p instance waits on another workgroup (split-K partials are summed by a separate launch).  0 restores the default; until
 * the first call the environment variable CM3P_G8P_GRID (read once) supplies the value.  Returns CM3P_ERR_INVALID for workgroups < 0. */
int cm3p_gemm8p_set_grid(int workgroups);
int cm3p_gemm8p_get_grid(void);

/* Host-only.  The hand-scheduled kernels carry compile-time TIMING probes (macros CM3P_ABL, CM3P_FABL, CM3P_BABL, CM3P_G256_ABL,
 * CM3P_G8P_ABL: builds that skip barriers, loads or stores and whose results are wrong by construction; the _ablate.sh scripts under tools/ubench).
 * Returns a bit mask of the objects in THIS library that were built with any of them set: 0 for every library that may be used
 * for results (cm3p_amd/_lib.py refuses to load anything else; tests/test_cabi.py). */
int cm3p_build_ablation_flags(void);

/* Debug builds only (libcm3p_hip_audit.so, -DCM3P_DMA_AUDIT=1; bit 5 of cm3p_build_ablation_flags): every LDS-DMA staging helper of the
 * GEMM and attention kernels then records, per operand id (csrc/common.h: 0 / 1 GEMM A / B, 2 / 3 the two tile matrices of an attention
 * ring, 4 / 5 statistics or mask rows), the lowest first byte and the highest last byte it reads into buf[id][0] / buf[id][1]
 * (uint64, atomicMin / atomicMax: the caller initialises them to ~0 and 0); NULL switches the recording off.  Not stream-ordered: call
 * it between synchronised launches.  The shipped library returns CM3P_ERR_INVALID.  tests/test_dma_audit_gpu.py asserts on the edge
 * shapes that every recorded address lies inside the tensor the caller handed over (the r03 over-read of the GEMM's staging stream). */
int cm3p_debug_set_dma_audit(void* buf);

/* Fused Wqkv projection + rotary embedding: qkv[M, N] (bf16) = x[M, K] Wqkv[N, K]^T with apply_rotary_pos_emb applied to the
 * first rope_cols (= 2H: the q and k thirds) columns (TF:...modeling_modernbert.py:271-280).  The 256 x 256 kernel rotates the
 * bf16-rounded projection in fp32 while it stores the staged rows (what the reference's autocast path computes: rotary on the
 * bf16 linear output); the 128 x 128 kernel for small shapes rotates the fp32 accumulators before rounding.  Both are inside
 * the tolerance of the tests.  cos/sin: [n_pos, 32] fp32 from cm3p_rope_table; token row m uses table row m (per_batch != 0)
 * or m % S.  Heads are 64 wide; N and rope_cols are multiples of 64.
 * q_scale: the first rope_cols / 2 columns (the q third) are multiplied by q_scale in fp32 BEFORE the single bf16 rounding of the
 * rotated value.  With q_scale = scale * log2(e) the attention kernels (q_prescaled = 1) get scores that are already in the
 * exp2 units of their softmax: no per-score multiply and no second rounding of q - the same number of roundings as the
 * reference's path, which rounds the rotated q once and scales inside SDPA in fp32.  q_scale = 1 leaves q as the reference's. */
int cm3p_qkv_gemm_rope(const void* x, const void* Wqkv, void* qkv, int64_t M, int64_t N, int64_t K, const float* cos_tab,
                       const float* sin_tab, int S, int per_batch, int rope_cols, float q_scale, void* stream);

/* Host-only: the split_k the library recommends for a weight-gradient GEMM of this shape (sizes the workspace). */
int cm3p_gemm_wgrad_splits(int64_t M, int64_t N, int64_t K);

/* fp32 -> bf16 cast of n elements (n % 4 == 0): the autocast weight / activation cast in front of a bf16 linear. */
int cm3p_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream);
/* The same cast of a [rows, cols] matrix plus its transpose y_t [cols, rows] in one pass (identical bf16 values).  The input-gradient
 * GEMM dx = dy W then reads W^T with the contraction index contiguous (a_kc = b_kc = 1), the faster operand form of the 256 x 256
 * kernel - nn.Linear's autograd does the same by handing the transposed view to the BLAS.  rows, cols multiples of 8. */
int cm3p_cast_f32_bf16_t(const float* x, void* y, void* y_t, int64_t rows, int64_t cols, void* stream);
/* The same for n matrices in ONE launch (all the projection weights of a tower at the start of a training forward).  table: n rows of
 * six int64 on the device - source (fp32 [rows, cols]), bf16 copy, transposed bf16 copy [cols, rows] (device addresses, 16-byte
 * aligned), rows, cols (multiples of 8), index of the matrix's first 64 x 64 block - with first-block indices ascending from 0;
 * total_blocks = the sum of ceil(rows / 64) * ceil(cols / 64).  The table is the caller's (it must outlive the launch). */
int cm3p_cast_f32_bf16_t_multi(const int64_t* table, int n, int64_t total_blocks, void* stream);
/* y_f32 (and y_bf16 if not NULL) = a_f32 + b (b fp32 or bf16); n % 4 == 0.  Residual-gradient join for layer 0,
 * whose attn_norm is nn.Identity (TF:...modeling_modernbert.py:309-310). */
int cm3p_add_f32(const float* a, const void* b, int b_dtype, float* y_f32, void* y_bf16, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Rotary position embedding.
 * cm3p_rope_table: cos/sin[p, j] = cos/sin(float(position_ids[p]) * inv_freq[j]), j < half_dim, fp32
 *   (ModernBertRotaryEmbedding.forward, TF:...modeling_modernbert.py:146-163; the caller supplies inv_freq computed
 *   as :141).
 * cm3p_rope_apply: in place on the q and k thirds of a packed qkv buffer [B, S, 3, nh, 64] (bf16):
 *   x' = x*cos + rotate_half(x)*sin in fp32, rounded back to bf16 (apply_rotary_pos_emb, :196-219).
 *   pos_batch_stride = 0 when one position row serves every batch element, else S.  inverse != 0 applies the
 *   transpose rotation (the backward pass).
 */
int cm3p_rope_table(const int64_t* position_ids, int64_t n_pos, const float* inv_freq, int half_dim, float* cos_out,
                    float* sin_out, void* stream);
int cm3p_rope_apply(void* qkv, const float* cos_tab, const float* sin_tab, int B, int S, int nh, int64_t pos_batch_stride,
                    int inverse, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Non-causal flash attention, head_dim 64, bf16 in / fp32 softmax / bf16 out.
 * Replaces sdpa_attention_forward -> F.scaled_dot_product_attention(q, k, v, attn_mask, scale, is_causal=False)
 * (TF:integrations/sdpa_attention.py:153-163, called from TF:...modeling_modernbert.py:286-297) together with the
 * mask the reference materialises (TF:masking_utils.py:141-179): key kv is visible to query q of batch b iff
 *   key_mask[b, kv] != 0  AND  (window < 0  OR  |q - kv| <= window).
 * The (B,1,S,S) mask is never built.  Rows with no visible key produce exact zeros.
 *   qkv: [B, S, 3, nh, 64] bf16 (q and k already rotated);  out: [B, S, nh, 64] bf16;  lse: [B, nh, S] fp32
 *   (natural-log sum-exp of the scaled scores; +inf for rows with no visible key);  key_mask: [B, S] bytes or NULL.
 */
/* q_prescaled != 0: the q third of qkv already holds q * scale * log2(e) (cm3p_qkv_gemm_rope with that q_scale); the kernels
 * then skip the scaling.  q_prescaled == 0: plain q; the kernels apply scale * log2(e) in fp32 on the score accumulators (one
 * extra multiply per score) - q is never re-rounded to bf16 either way.  `scale` is always the softmax scale (1 / sqrt(64)). */
int cm3p_attn_fwd(const void* qkv, void* out, float* lse, const uint8_t* key_mask, int B, int S, int nh, int window,
                  float scale, int q_prescaled, void* stream);
/* Host-only: which forward kernel cm3p_attn_fwd / cm3p_attn_fwd_varlen launches for this call - 1: the pipelined global kernel of
 * csrc/attention_fwd.hip (window < 0, pre-scaled q, row offsets within 32 bits, CM3P_ATTN_FWD_IMPL not "wave3"), 0: attn_fwd_kernel of
 * csrc/attention.hip.  The ONE place the routing is decided; callers that label launches (profiler tags) ask instead of re-deriving it. */
int cm3p_attn_fwd_impl(int S, int nh, int window, int q_prescaled);

/* output_attentions: the attention probabilities [B, nh, S, S] fp32 of one layer from qkv and the lse cm3p_attn_fwd stored - what
 * the reference returns as `attentions` (TF switches to eager_attention_forward for such a call,
 * TF:models/modernbert/modeling_modernbert.py:133-170).  An inspection path (plain fp32 arithmetic, B * nh * S * S * 4 bytes of
 * output).  Invisible keys: exact zeros; a row with no visible key: uniform 1 / S, as the eager path's finite additive mask leaves it.
 * Padded layout only. */
int cm3p_attn_probs(const void* qkv, const float* lse, const uint8_t* key_mask, float* probs, int B, int S, int nh, int window, float scale,
                    int q_prescaled, void* stream);
/* Backward.  delta: [B, nh, S] fp32 workspace.  dqkv: [B, S, 3, nh, 64] bf16, fully overwritten.
 * If cos_tab/sin_tab are not NULL the inverse rotary rotation is applied to dq and dk before they are stored (the
 * backward of apply_rotary_pos_emb), with pos_batch_stride = 0 (one position row for all batches) or S.
 * stages: which of the backward's kernels to launch - CM3P_ATTN_BWD_DQ (the dq third of dqkv, and delta),
 * CM3P_ATTN_BWD_DKV (the dk and dv thirds; reads the delta a DQ stage wrote earlier on the same stream), or both (3).
 * Callers that time kernels one by one issue the stages as two calls; the results are identical.
 * dqkv's q third is the gradient w.r.t. the UN-scaled rotated q in both q_prescaled modes (the chain rule through q_scale is
 * applied inside), i.e. what the Wqkv GEMM's backward expects. */
#define CM3P_ATTN_BWD_DQ 1
#define CM3P_ATTN_BWD_DKV 2
int cm3p_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                  const uint8_t* key_mask, int B, int S, int nh, int window, float scale, const float* cos_tab,
                  const float* sin_tab, int64_t pos_batch_stride, int stages, int q_prescaled, void* stream);

/* Head sizes other than 64 (csrc/attention_generic.hip): the same attention contract - qkv [B, S, 3, nh, head_dim] bf16 with q and k already
 * rotated, out [B, S, nh, head_dim] bf16, lse / delta [B, nh, S] fp32, the mask rule and the exact zeros / lse = +inf of rows without a
 * visible key as cm3p_attn_fwd / cm3p_attn_bwd - as plain fp32 kernels (one thread per query or key row, no matrix cores) for head_dim 16
 * and 32 (and 64, as a cross-check of the MFMA kernels).  They exist so that EVERY configuration of the reference runs, e.g. its own tiny
 * test configuration (hidden 64, 4 heads; BASELINE.json configs[0]); the default towers are all head_dim 64 and never come here.  q is
 * plain (no pre-scaling), dqkv's q / k thirds are gradients w.r.t. the ROTATED q / k: cm3p_rope_apply_generic(..., inverse = 1) on dqkv
 * completes the backward of apply_rotary_pos_emb (TF:models/modernbert/modeling_modernbert.py:188-219), whose forward it also is
 * (in place on the q and k thirds of a packed qkv; cos / sin [n_pos, head_dim / 2] from cm3p_rope_table). */
int cm3p_attn_generic_supported(int head_dim);
int cm3p_attn_fwd_generic(const void* qkv, void* out, float* lse, const uint8_t* key_mask, int B, int S, int nh, int head_dim, int window,
                          float scale, void* stream);
int cm3p_attn_bwd_generic(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                          const uint8_t* key_mask, int B, int S, int nh, int head_dim, int window, float scale, void* stream);
int cm3p_rope_apply_generic(void* qkv, const float* cos_tab, const float* sin_tab, int B, int S, int nh, int head_dim, int64_t pos_batch_stride,
                            int inverse, void* stream);

/* The same backward for GLOBAL layers (no window) as one key-parallel kernel that executes each of the five matrix products
 * once (cm3p_attn_bwd with window < 0 runs a query-parallel and a key-parallel kernel that both recompute the scores: seven).
 * Padded batches: cu_seqlens = NULL, total = 0, S = the padded length, key_mask [B, S] or NULL, lse [B, nh, S].
 * Unpadded batches: cu_seqlens [B + 1], S = max_seqlen, total rows, key_mask NULL, pos_batch_stride 0, lse [nh, total], rotary
 * tables per token (as cm3p_attn_bwd_varlen).
 * stages: CM3P_ATTN_BWD_FUSED_PREP (delta and the per-tile score offsets -> workspace), _MAIN (dk and dv thirds of dqkv; one bf16
 * partial dq per PAIR of 256-key blocks -> workspace: two launches, see _MAIN_EVEN / _MAIN_ODD), _REDUCE (the dq third of dqkv from
 * the partials: the two key blocks of a slab are summed in bf16 by the memory side's packed add (_MAIN_ODD, one more rounding than a
 * pure fp32 sum), the slabs in fp32 in a fixed order by _REDUCE: deterministic); a caller issues all three in this order on one stream (7), or one by one to time them.
 * workspace: caller-owned device memory of at least cm3p_attn_bwd_fused_workspace_bytes(B, S, nh) bytes, 16-byte aligned;
 * contents are scratch (nothing is carried between calls). */
#define CM3P_ATTN_BWD_FUSED_PREP 1
#define CM3P_ATTN_BWD_FUSED_MAIN 2      /* = _MAIN_EVEN then _MAIN_ODD */
#define CM3P_ATTN_BWD_FUSED_REDUCE 4
#define CM3P_ATTN_BWD_FUSED_MAIN_EVEN 8 /* the main kernel over key blocks 0, G, 2 G ... (G = cm3p_attn_bwd_fused_slab_group): each STORES its bf16 dq partial to slab kblk / G */
#define CM3P_ATTN_BWD_FUSED_MAIN_ODD 16 /* ... over the other key blocks, one launch per position in the group: each ADDS its partial to the same slab (one packed-bf16 atomic
                                           add per element, after the store by stream order: deterministic); must follow _MAIN_EVEN */
#define CM3P_ATTN_BWD_FUSED_MAIN_ADD1 32 /* the adding launches one by one (callers that time kernels): _ADD1 << (p - 1) = position p of the group, p = 1 .. G - 1 */
int64_t cm3p_attn_bwd_fused_workspace_bytes(int B, int S, int nh);
/* Host-only: how many consecutive 256-key blocks share one dQ slab at sequence length S (2, or 4 for long sequences: the first block of
 * a group stores its partial, the others add theirs with packed-bf16 atomics in launches of their own; the environment variable
 * CM3P_FUSED_SLAB_GROUP=2|4 overrides).  _MAIN_EVEN = the storing launch, _MAIN_ODD = the adding launches, in order. */
int cm3p_attn_bwd_fused_slab_group(int S);
int cm3p_attn_bwd_fused(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, const uint8_t* key_mask,
                        const int* cu_seqlens, int B, int S, int64_t total, int nh, float scale, const float* cos_tab,
                        const float* sin_tab, int64_t pos_batch_stride, int stages, int q_prescaled, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * GeGLU: g = gelu_erf(h[:, :I]) * h[:, I:]   (ModernBertMLP.forward, TF:...modeling_modernbert.py:89-91).
 * h: [T, 2I] bf16, g: [T, I] bf16; I % 8 == 0.  Backward: dh from dg and h.
 */
int cm3p_geglu_fwd(const void* h, void* g, int64_t T, int I, void* stream);

/* The Wi projection and GeGLU in one kernel, for forward-only calls (evaluation, embedding extraction: nothing keeps h and g for a
 * backward pass; replaces nn.Linear Wi + act * gate of ModernBertMLP.forward, TF:...modeling_modernbert.py:89-91, in that mode):
 *   a[t, j] = gelu_erf(bf16(x[t] . Wi[j])) * bf16(x[t] . Wi[I + j]),  j < I      - the values cm3p_gemm_bf16 + cm3p_geglu_fwd give, bit for bit.
 * x: [T, K] bf16; a: [T, I] bf16; w_interleaved: [2I, K] bf16 = Wi with its rows reordered so that every 64 consecutive rows are
 * 32 rows of the first half followed by the 32 matching rows of the second half: row 64 q + r is Wi[32 q + r] for r < 32 and
 * Wi[I + 32 q + r - 32] for r >= 32 (cm3p_amd/encoder.py makes the copy once per weight version).
 * Shapes of the 256 x 256 ring kernel only: K % 64 == 0, I % 32 == 0, T % 8 == 0, ceil(T / 256) * ceil(2I / 256) >= 200;
 * anything else is CM3P_ERR_INVALID and the caller keeps the two-kernel path. */
int cm3p_gemm_geglu(const void* x, const void* w_interleaved, void* a, int64_t T, int64_t I, int64_t K, void* stream);
int cm3p_geglu_bwd(const void* dg, const void* h, void* dh, int64_t T, int I, void* stream);
/* y = gelu_erf(x) elementwise on bf16, and dx = dy * gelu'(x)  (nn.functional.gelu at ref:cm3p/modeling_cm3p.py:478,501-502). */
int cm3p_gelu_fwd(const void* x, void* y, int64_t n, void* stream);
int cm3p_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Audio front end: Conv1d(kernel 3, padding 1, stride 1|2) as im2col + cm3p_gemm_bf16, then bias + GELU
 * (CM3PAudioEncoder.forward, ref:cm3p/modeling_cm3p.py:488-489,501-504).
 * cm3p_im2col_k3: x is [B, C, T_in] fp32 (x_token_major = 0, the mel input) or [B, T_in, C] bf16 (x_token_major = 1);
 *   patches [B*T_out, C*3] bf16 with column c*3 + kk = x[b, c, t*stride + kk - 1] (zero outside), matching the
 *   Conv1d weight [C_out, C, 3] viewed as [C_out, C*3].  T_out = (T_in - 1) / stride + 1.
 * cm3p_col2im_k3: the transpose for token-major inputs: dx [B, T_in, C] bf16 from dpatches.
 * cm3p_bias_gelu_fwd: a = gelu_erf(z + bias), z fp32 [R, C]; writes bf16 and/or fp32.
 * cm3p_bias_gelu_bwd: dz = da * gelu'(z + bias) (bf16) and dbias[C] = column sums of dz; db_partial is a
 *   [cm3p_bias_gelu_bwd_blocks(R), C] fp32 workspace.
 */
int cm3p_im2col_k3(const void* x, int x_token_major, void* patches, int B, int C, int T_in, int T_out, int stride, void* stream);
int cm3p_col2im_k3(const void* dpatches, void* dx, int B, int C, int T_in, int T_out, int stride, void* stream);
int cm3p_bias_gelu_fwd(const float* z, const float* bias, void* a_bf16, float* a_f32, int64_t R, int C, void* stream);
int cm3p_bias_gelu_bwd_blocks(int64_t R);
int cm3p_bias_gelu_bwd(const void* da, int da_dtype, const float* z, const float* bias, void* dz_bf16, float* db_partial,
                       float* dbias, int64_t R, int C, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Pooling of the last hidden state (ref:cm3p/modeling_cm3p.py:385-396, 631-642).
 *   cls != 0: pooled[b] = h[b, 0];  else pooled[b] = sum_s h[b,s]*m[b,s] / max(sum_s m[b,s], 1e-9)  (m all ones if NULL).
 * h: [Bn, S, H] fp32, mask: [Bn, S] int64 or NULL, pooled: [Bn, H] fp32.  partial: fp32 workspace
 * [Bn, cm3p_pool_chunks(S), H]; count: [Bn] fp32, sum of the mask row (saved for the backward pass).
 */
int cm3p_pool_chunks(int S);
int cm3p_pool_fwd(const float* h, const int64_t* mask, float* pooled, float* partial, float* count, int Bn, int S, int H,
                  int cls, void* stream);
int cm3p_pool_bwd(const float* dpooled, const int64_t* mask, const float* count, float* dh, int Bn, int S, int H, int cls,
                  void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Contrastive head, fp32 throughout (ref:cm3p/modeling_cm3p.py:27-62, 958-985).
 */
/* C[m, n] (+)= alpha * sum_k A[m*a_rs + k*a_cs] * B[n*b_rs + k*b_cs]; generic strides, small problems only. */
int cm3p_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int64_t a_rs, int64_t a_cs, int64_t b_rs,
                  int64_t b_cs, int64_t ldc, float alpha, int accumulate, void* stream);
/* y = x / sqrt(sum x^2) per row, no eps (_get_vector_norm, ref:cm3p/modeling_cm3p.py:54-62,960,972); norm[rows] saved. */
int cm3p_l2norm_fwd(const float* x, float* y, float* norm, int rows, int D, void* stream);
int cm3p_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx, int rows, int D, void* stream);
/* Cross-entropy over `rows` rows of `cols` logits addressed as logits[row_offset[r] + c*col_stride]
 * (row_offset NULL -> r*row_stride), target[r] in [0, cols): nn.functional.cross_entropy (ref:cm3p/modeling_cm3p.py:27-29)
 * on a strided view, so `similarity`, `similarity.t()`, `similarity[arange, idx]` and
 * `similarity.permute(2,0,1).reshape(B,-1)` (:41-50) need no copies.
 * loss_rows[r] = logsumexp - logit[target].  If dlogits != NULL: dlogits[same address] += grad_scale * (softmax - onehot). */
int cm3p_cross_entropy(const float* logits, int rows, int cols, int64_t row_stride, int64_t col_stride,
                       const int64_t* row_offset, const int64_t* target, float grad_scale, float* loss_rows, float* dlogits,
                       void* stream);
/* y = x * exp(*log_scale): `logits * self.logit_scale.exp()` (ref:cm3p/modeling_cm3p.py:977) without a host read. */
int cm3p_scale_exp(const float* x, const float* log_scale, float* y, int64_t n, void* stream);
/* y = x * (*scale), scale a device scalar: chain rule through a scalar loss without a host read. */
int cm3p_scale_by(const float* x, const float* scale, float* y, int64_t n, void* stream);
/* out[0] = sum a[i]*b[i] (fixed order): d logit_scale = <dlogits, logits>. */
int cm3p_dot_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
/* out[0] (+)= scale * sum x[i] (fixed order): the mean over rows inside cross_entropy and the /2 of cm3p_loss. */
int cm3p_sum_f32(const float* x, float* out, int64_t n, float scale, int accumulate, void* stream);
/* Masked-LM head (CM3PPredictionHead + decoder + ForMaskedLMLoss; ref:cm3p/modeling_cm3p.py:987-996,1229-1238,
 * TF:loss/loss_utils.py:32-46,74-91) - the pieces the encoder kernels do not already cover:
 * cm3p_cross_entropy_masked: cross_entropy(ignore_index) over contiguous rows of `cols` logits with row pitch
 *   row_stride (>= cols; the pad columns get zero gradient).  loss_rows[r] = 0 for ignored rows.  dlogits (may be NULL) is
 *   fully written: grad_scale * inv_count[0] * (softmax - onehot).
 * cm3p_inv_valid_count: inv_count[0] = 1 / max(#(target != ignore_index), 1) - the "mean" denominator, kept on the device.
 * cm3p_add_bias_f32: x[r, :] += bias.   cm3p_colsum_f32: out[c] = sum_r x[r, c] (fixed order; partial is a
 *   [cm3p_colsum_blocks(rows), cols] workspace) - the decoder bias gradient. */
int cm3p_cross_entropy_masked(const float* logits, int64_t rows, int cols, int64_t row_stride, const int64_t* target,
                              int64_t ignore_index, float grad_scale, const float* inv_count, float* loss_rows, float* dlogits,
                              void* stream);
int cm3p_inv_valid_count(const int64_t* target, int64_t n, int64_t ignore_index, float* inv_count, void* stream);
/* The training path of the same loss without an fp32 gradient of the logits ([B*S, vocab] fp32 = 1.66 GB at B=32, S=4096):
 * cm3p_ce_masked_stats: loss_rows[r] = lse_r - x[r, target_r] and lse_rows[r] = lse_r for labelled rows, both 0 for ignored
 *   rows (whose logits are not read).
 * cm3p_ce_masked_dlogits_bf16: dlogits_bf16[r, c] = bf16(s * (exp(x[r, c] - lse_r) - [c == target_r])), s = scale_a[0] * scale_b[0]
 *   (device scalars: the incoming loss gradient and 1 / #labelled), zero for ignored rows and pad columns - the operand of the
 *   decoder's dgrad / wgrad GEMMs; colsum[c] = sum_r of the unrounded values (the decoder bias gradient, fixed order).
 *   row_stride % 4 == 0; partial: [cm3p_ce_masked_dlogits_blocks(rows), row_stride] fp32 workspace. */
int cm3p_ce_masked_stats(const float* logits, int64_t rows, int cols, int64_t row_stride, const int64_t* target, int64_t ignore_index,
                         float* loss_rows, float* lse_rows, void* stream);
int cm3p_ce_masked_dlogits_blocks(int64_t rows);
int cm3p_ce_masked_dlogits_bf16(const float* logits, int64_t rows, int cols, int64_t row_stride, const int64_t* target, int64_t ignore_index,
                                const float* lse_rows, const float* scale_a, const float* scale_b, void* dlogits_bf16, float* partial,
                                float* colsum, void* stream);
int cm3p_add_bias_f32(float* x, const float* bias, int64_t rows, int cols, void* stream);
int cm3p_colsum_blocks(int64_t rows);
int cm3p_colsum_f32(const float* x, float* partial, float* out, int64_t rows, int cols, void* stream);

/* Classifier variant (CM3PForBeatmapClassification, ref:cm3p/modeling_cm3p.py:1196-1218): mean MSELoss (kind 0) or
 * BCEWithLogitsLoss (kind 1) over n elements; out[0] = loss, dx (may be NULL) = its gradient.  CrossEntropyLoss reuses
 * cm3p_cross_entropy. */
int cm3p_pointwise_loss(const float* x, const float* y, float* out, float* dx, int64_t n, int kind, void* stream);

/* idx[b] = first v with classes[b, v] == 0, else 0: `(classes == 0).int().argmax(dim=1)` (ref:cm3p/modeling_cm3p.py:40). */
int cm3p_first_zero_index(const int64_t* classes, int B, int V, int64_t* idx, void* stream);

/* ---- Unpadded ("varlen") execution (ref:cm3p/modeling_cm3p.py:65-134 _unpad_cm3p_input / _pad_cm3p_output, :911-931;
 * the flash_attn_varlen path of TF:models/modernbert/modeling_modernbert.py).  Valid tokens are packed back to back:
 * sequence b owns rows cu_seqlens[b] .. cu_seqlens[b+1]-1 (int32, B+1 entries, device) of qkv [total, 3, nh, 64],
 * out [total, nh, 64] and of the per-token rotary tables [total, 32]; lse / delta are [nh, total].  No key mask: every packed
 * token is valid.  Same kernels, same window rule and same results on the valid tokens as cm3p_attn_fwd / cm3p_attn_bwd.
 * cm3p_gather_rows_f32: dst[i, :] = src[idx[i], :];  cm3p_scatter_rows_f32: dst[idx[i], :] = src[i, :] (dst pre-zeroed by the
 * caller = _pad_cm3p_output); rows of H fp32 values, H % 4 == 0. */
int cm3p_attn_fwd_varlen(const void* qkv, void* out, float* lse, const int* cu_seqlens, int B, int max_seqlen, int64_t total,
                         int nh, int window, float scale, int q_prescaled, void* stream);
int cm3p_attn_bwd_varlen(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                         const int* cu_seqlens, int B, int max_seqlen, int64_t total, int nh, int window, float scale,
                         const float* cos_tab, const float* sin_tab, int stages, int q_prescaled, void* stream);
int cm3p_gather_rows_f32(const float* src, const int64_t* idx, float* dst, int64_t n, int H, void* stream);
int cm3p_scatter_rows_f32(const float* src, const int64_t* idx, float* dst, int64_t n, int H, void* stream);

/* ---- Muon optimizer step (ref:utils/muon_utils.py:35-57 zeropower_via_newtonschulz5, :138-203 Muon.step) -----------
 * The step that follows the hot path in every training recipe >= v2 (ref:configs/train/v2.yaml:9).
 *
 * cm3p_gemm_bf16_batched: C_b = bf16(alpha * A_b B_b^T + beta * R_b) for b < batch, operands as in cm3p_gemm_bf16, matrix b
 *   at base + b * stride (elements).  R (bf16, C's layout) may be NULL.  One Newton-Schulz iteration over a group of
 *   same-shaped weights is three calls: A = X X^T;  B = b A + c A A;  X' = a X + B X  (:50-53).
 * Same-shaped weights form a group.  Parameters, gradients and optimizer state stay in torch's own allocations and are
 * passed as DEVICE tables of int64 addresses (fp32, contiguous); the iterate X is a bf16 workspace [n_mat][x_stride]
 * holding each rows x cols matrix at row pitch ldx (extents rounded up to 8, padding zero).
 * cm3p_muon_momentum: buf = momentum*buf + g; u = nesterov ? g + momentum*buf : g (as :163-164 does); X = bf16(u);
 *   partials[mat, 0..cm3p_muon_partials(rows, cols)) = fixed-order partial sums of X^2 (:160-164, :46).
 *   aligned16 != 0 promises that every address in both tables is 16-byte aligned (enables 16-byte accesses).
 * cm3p_muon_normalize: X /= bf16(bf16(||X||) + eps), in the reference's bf16 arithmetic (:47).
 * cm3p_muon_apply: p += neg_lr * float(bf16(X * shape_scale)), shape_scale = sqrt(max(1, rows/cols)) (:173-176).
 * cm3p_adamw_multi: the reference's AdamW branch for all other parameters in one launch (:178-203):
 *   m1 = lerp(m1, g, w1); m2 = lerp(m2, g*g, w2); p = p*decay + step_alpha * m1 / (eps + sqrt(m2)).
 *   The host passes w1 = 1-beta1, w2 = 1-beta2, decay = 1 - adamw_lr*wd, step_alpha = -lr/scale exactly as :195-203 does. */
int cm3p_gemm_bf16_batched(const void* A, const void* B, void* C, const void* R, int batch, int64_t M, int64_t N, int64_t K,
                           int64_t lda, int64_t ldb, int64_t ldc, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                           int64_t stride_r, int a_kc, int b_kc, float alpha, float beta, void* stream);
int cm3p_muon_partials(int rows, int cols);
int cm3p_muon_momentum(const int64_t* g_ptrs, const int64_t* buf_ptrs, void* X, float* partials, int n_mat, int rows, int cols,
                       int ldx, int64_t x_stride, float momentum, int nesterov, int aligned16, void* stream);
int cm3p_muon_normalize(void* X, const float* partials, int n_mat, int rows, int cols, int64_t x_stride, float eps, void* stream);
int cm3p_muon_apply(const int64_t* p_ptrs, const void* X, int n_mat, int rows, int cols, int ldx, int64_t x_stride,
                    float shape_scale, float neg_lr, void* stream);
int cm3p_adamw_multi(const int64_t* p_ptrs, const int64_t* g_ptrs, const int64_t* m1_ptrs, const int64_t* m2_ptrs,
                     const int64_t* numels, int n_tensors, int64_t max_numel, float w1, float w2, float eps, float decay,
                     float step_alpha, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CM3P_HIP_H */
