/*
 * valley_hip.h — C ABI of libvalley_hip.so, the MI355X (gfx950) kernels behind Valley's
 * visual-token hot path.
 *
 * The reference (RupertLuo/Valley) has no FFI/plugin layer: its hot path is Python calling
 * HuggingFace modules (valley/model/valley_model.py:163-254, 292-305).  This header is the boundary
 * a maintainer binds instead of those module calls (ctypes stub in INTEGRATION.md;
 * valley_amd/lib.py is that binding).  Each entry point cites the reference / third-party
 * call it replaces.  "hf:" = transformers/models/… (the un-vendored dependency pinned at
 * pyproject.toml:19).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator in practice);
 *     nothing is allocated, freed or retained; no global state; thread-safe per distinct stream.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised.
 *   - "bf16" in a name or comment = the library's 16-bit STORAGE type, raw uint16_t: bfloat16 in libvalley_hip.so, IEEE
 *     fp16 in libvalley_hip_f16.so (the same sources built with -DVLY_FP16=1: the reference infers in fp16,
 *     valley/inference/run_valley.py:39) — vly_storage_dtype() tells which; the residual stream and norm parameters are fp32.
 *   - row-major everywhere; leading dimensions are in ELEMENTS.
 *   - return 0 on success, -22 (EINVAL) for unsupported shapes/alignments,
 *     -(1000+hipError_t) if a launch failed.  vly_last_error() gives a thread-local message.
 */
#ifndef VALLEY_HIP_H
#define VALLEY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLY_ABI_VERSION 7   /* 2: + the fp32 "precise" entry points (vly_*_f32); 3: + vly_storage_dtype; 4: + vly_gemv_rmsnorm_bf16,
                               vly_decode_attention_split, vly_gemv_attnmerge_bf16; 5: + vly_decode_layers(_supported),
                               vly_decode_attention_merged; 6: vly_decode_layers(_supported) and tile hint 297 moved to the
                               EXPERIMENTAL library (libvalley_hip_exp.so, section at the end), vly_gemv_bf16 takes M <= 16; 7: + vly_split3_f32, vly_norm_split3_f32,
                               tile hints 397 / 398 / 497 of vly_gemm_bf16 */

/* epilogues of vly_gemm_bf16 */
#define VLY_EPI_NONE        0   /* C = A W^T (+bias) (+residual)                                   */
#define VLY_EPI_QUICK_GELU  1   /* C = q(A W^T + bias), q(x) = x*sigmoid(1.702x)  hf:activations.py  */
#define VLY_EPI_SWIGLU      2   /* W rows interleaved (gate_j, up_j): C[:, j] = silu(g_j) * u_j,
                                   C is N/2 wide                     hf:llama/modeling_llama.py:171 */
#define VLY_EPI_RELU        3   /* C = max(A W^T + bias, 0): the FFN of the v3 temporal transformer layer
                                   (torch.nn.TransformerEncoderLayer default activation)            */
#define VLY_EPI_QKV_ROPE    4   /* vly_gemm_bf16_qkv_rope only: the fused q|k|v projection whose epilogue applies
                                   rotate-half RoPE to q and k and appends k, v to the KV cache            */
/* output dtypes */
#define VLY_OUT_BF16 0
#define VLY_OUT_F32  1

/* pooling modes of vly_pool_tokens (valley/model/valley_model.py:206-209) */
#define VLY_POOL_MEAN 0
#define VLY_POOL_MAX  1
#define VLY_POOL_IMPORTANCE 2   /* v2: softmax over frames of Linear(256*H -> 1) scores, weighted sum (:113-121) */

/* ldw value meaning "W is in the block layout of vly_pack_weight_bf16" (vly_gemm_bf16, vly_gemm_bf16_splitk2) */
#define VLY_LDW_PACKED64 (-64)

int         vly_abi_version(void);
const char *vly_last_error(void);
/* 16-bit storage type of every half-precision tensor this library reads and writes: VLY_STORAGE_BF16 (libvalley_hip.so)
 * or VLY_STORAGE_FP16 (libvalley_hip_f16.so).  The host must allocate and convert accordingly (torch.bfloat16 /
 * torch.float16 behind run_valley.py:39's ``model.to(torch.float16)``). */
#define VLY_STORAGE_BF16 0
#define VLY_STORAGE_FP16 1
int         vly_storage_dtype(void);

/* GEMM with fused epilogue:  C[M,N] = epi(A[M,K] · W[N,K]^T + bias[N]) + residual[M,N]
 *   A, W bf16; bias fp32 or NULL; residual fp32 or NULL (ldr); C bf16 or fp32 (out_dtype).
 *   W is an nn.Linear weight as stored ([out,in] row-major) — no transpose is ever materialised.
 *   Requires K % 64 == 0, lda/ldw % 8 == 0, N % 4 == 0 (N % 8 for SWIGLU), 16-byte aligned bases.
 *   Replaces every nn.Linear / Conv2d-as-GEMM on the path: CLIP patch embedding
 *   (hf:clip/modeling_clip.py:148-154,209), q/k/v/out_proj (:293-296), fc1/fc2 (:343-344),
 *   mm_projector (valley_model.py:54-55,190), Llama q/k/v/o (hf:llama/modeling_llama.py:230-241),
 *   gate/up/down (:166-168), lm_head (valley_model.py:264,305).
 *   tile_hint: 0 = auto, otherwise the kernel a tuner picked (valley_amd/tuned/gfx950.json; every hint computes the same
 *   product up to fp32 summation order).  BM x BN tiles, K tile 64:
 *     1 = 256x256, 2 = 128x128, 3 = 256x128, 4 = 128x256, 5 = 192x256, 6 = 192x192 (8 waves, two LDS stages);
 *     7 = 128x192, 8 = 192x128 (80 KB of LDS: TWO workgroups share a CU and cover each other's prologue / epilogue);
 *     9 = 256x256 with 16 waves;   51, 53, 54, 55 = tiles 1, 3, 4, 5 with the wave role split (the two waves of a SIMD run
 *     one phase apart);   73, 74, 76 = tiles 3, 4, 6 with THREE stages (two K tiles of LDS-DMA in flight), 83, 84, 86 = the
 *     same with the role split;   93 / 94 = 256x128 / 128x256, 16 waves, three stages;
 *     97 / 98 / 99 = 256x256 / 224x256 / 192x256 with FOUR waves (one per SIMD, a (BM/2) x 128 accumulator block each, 512
 *     registers per lane; DESIGN.md "the 4-wave kernels");   197 / 198 / 199 = the same tiles in the PERSISTENT kernel (one
 *     workgroup per CU walks the tiles, register-only epilogue, outputs through a clipping buffer descriptor — the hot
 *     path's kernel; bf16 outputs whose rows are not 16-byte aligned or whose width is not a multiple of 8, bf16 +
 *     residual, outputs of 2 GB or more and K < 128 fall back to 97 / 98 / 99; not for the split-K pair).
 *   ldw = VLY_LDW_PACKED64: W points to the block layout written by vly_pack_weight_bf16. */
int vly_gemm_bf16(const void *A, const void *W, const float *bias, const float *residual, void *C,
                  int M, int N, int K, int lda, int ldw, int ldc, int ldr,
                  int epilogue, int out_dtype, int tile_hint, void *stream);

/* The fused q|k|v projection of a Llama layer with RoPE and the KV-cache append in its epilogue
 *   (hf:llama/modeling_llama.py:230-238 projections, :127-157 rotate-half RoPE, :261-262 cache update): A [M = B*S, K = H]
 *   bf16, W = [q; k; v] rows [3H, H]; row m is token s = m % S of sequence b = m / S at position past_len + s.
 *   q columns: rotated, written to qkv[m, 0:H] (row stride ldc; the k / v columns of that buffer are NOT written);
 *   k columns: rotated, written to kcache[b, head, past_len + s, :]; v columns: written to vcache likewise
 *   (caches bf16 [B, heads, ctx_max, 128]).  The rotation is applied to the bf16-rounded projection with the fp32
 *   cos / sin tables [pos, 64], i.e. bit-identical to vly_gemm_bf16 followed by vly_rope_kv.  Needs a tile whose width is
 *   a multiple of 128 (a head's two halves in one tile: every tile_hint whose tile is 128 or 256 columns wide, whole-K-tile
 *   loops only) and
 *   16-byte aligned qkv rows; other tile hints return -22. */
int vly_gemm_bf16_qkv_rope(const void *A, const void *W, void *qkv, void *kcache, void *vcache, const float *cos_table,
                           const float *sin_table, int M, int H, int K, int lda, int ldw, int ldc, int S, int heads,
                           int past_len, int ctx_max, int tile_hint, void *stream);

/* Split-K by two in ONE launch:  C0 = A[:, :K/2] . W[:, :K/2]^T + bias,  C1 = A[:, K/2:] . W[:, K/2:]^T, both bf16
 *   [M,N] with row stride ldc; the consumer adds them (vly_add2_rmsnorm).  The grid holds every tile twice, so a
 *   projection with fewer tiles than CUs (Llama o_proj / down_proj at M = 1312: 176-224 tiles) fills the chip with
 *   paired workgroups; the partial sums travel as bf16 (2 x 10.7 MB at c2), not as an fp32 round trip.
 *   Same constraints as vly_gemm_bf16 (K >= 128); tile_hint as there (0 = 192x128). */
int vly_gemm_bf16_splitk2(const void *A, const void *W, const float *bias, void *C0, void *C1,
                          int M, int N, int K, int lda, int ldw, int ldc, int tile_hint, void *stream);

/* Weight repack for the prefill GEMMs (done once at load, like the q|k|v fusion): W [N,K] row-major bf16 ->
 *   packed [K/64][ceil(N/64)][64 rows][64 k], rows >= N zero-filled; ceil(N/64)*64*K elements.  In this layout the
 *   K tile of a weight panel that a workgroup stages per iteration is ONE contiguous run in HBM (8 KB per 64 rows)
 *   instead of one 128-byte line out of each 2*K-byte row: +1..3 % on the large GEMMs, +6..12 % on the N = 4096
 *   projections at M = 1312 (profiles/history/r01/r01_ab_lib_v17.jsonl).  The GEMV (decode) and stream-K kernels read row-major
 *   weights only, so a model that decodes keeps both copies (valley_amd.ops.PackedWeight). */
int vly_pack_weight_bf16(const void *W, void *packed, int N, int K, int ldw, void *stream);

/* Persistent stream-K variant of vly_gemm_bf16 (same math, epilogues and constraints): the
 *   (tile, k-tile) iteration space is cut into equal contiguous ranges over CUs x occupancy persistent
 *   workgroups, so small-M problems (configs[1]: M = 1312) keep every CU busy; partial tiles are
 *   combined through fp32 slabs in `workspace` (>= vly_gemm_streamk_workspace_bytes(), 16-byte
 *   aligned, ZEROED once by the caller, then owned by this entry point on one stream at a time).
 *   `epoch` must be non-zero and different on every launch that shares the workspace.
 *   Deterministic for a given shape, but the in-tile summation order depends on M (not
 *   batch-invariant like vly_gemm_bf16).
 *   tile_hint 298 / 299 (round 3): the persistent 4-wave kernel of vly_gemm_bf16 (hints 198 / 199: 224 / 192 x 256
 *   tiles, one workgroup per CU) with its REMAINDER ROUND split along K — the tiles past the last whole round of CUs
 *   are cut into S equal K slices (S picked per shape so that S x remainder fills whole rounds), run first; the
 *   slices of a tile hand a running fp32 sum down a chain of slabs in `workspace`, the last one runs the epilogue.
 *   These hints also take ldw = VLY_LDW_PACKED64.  Shipped for the 7B prefill shapes (M = 1312): gate|up 216.8 ->
 *   194 us, q|k|v 138.6 -> 125 us.  tile_hint 297 (round 4): the same for the 256 x 256 tile (hint 197) — the owning slice adds
 *   the chain's running sum to its accumulators with exact f32-input MFMAs before a plain epilogue. */
int    vly_gemm_bf16_streamk(const void *A, const void *W, const float *bias, const float *residual, void *C,
                             int M, int N, int K, int lda, int ldw, int ldc, int ldr,
                             int epilogue, int out_dtype, int tile_hint,
                             void *workspace, size_t workspace_bytes, unsigned epoch, void *stream);
size_t vly_gemm_streamk_workspace_bytes(void);
int    vly_gemm_streamk_tile_for(int M, int N, int K);

/* The tile configuration vly_gemm_bf16 picks for (M,N) when tile_hint == 0 (1/2/3 as above, 197 when the
 * problem has at least 192 tiles of 256x256); lets a
 *   caller label launches when it profiles (bench.py's per-kernel roofline). */
int vly_gemm_tile_for(int M, int N);

/* LayerNorm over the last dim of an fp32 [M,D] tensor -> bf16 (GEMM input) and optionally fp32.
 *   hf:clip/modeling_clip.py:605,642 (pre_layrnorm), :363,368 (layer_norm1/2).  D % 256 == 0, D <= 8192. */
int vly_layernorm(const float *x, const float *gamma, const float *beta, void *y_bf16, float *y_f32,
                  int M, int D, float eps, void *stream);

/* RMSNorm (fp32 statistics) of fp32 [M,D] -> bf16.  hf:llama/modeling_llama.py:61-66.  Same D limits. */
int vly_rmsnorm(const float *x, const float *gamma, void *y_bf16, int M, int D, float eps, void *stream);

/* Residual update fused with the following norm:  h[M,D] (fp32, in place) += delta (bf16, the output of the
 *   sub-layer's last GEMM), then y = LayerNorm/RMSNorm(h) -> bf16.  gamma == NULL: add only (the last
 *   residual update of a stack).  Moves the `residual + hidden_states` of hf:clip/modeling_clip.py:376,381
 *   and hf:llama/modeling_llama.py:317,323 out of the GEMM epilogue (64-byte row pieces, exposed) into a
 *   streaming kernel.  Same D limits as vly_layernorm. */
int vly_add_layernorm(float *h, const void *delta_bf16, const float *gamma, const float *beta, void *y_bf16,
                      int M, int D, float eps, void *stream);
int vly_add_rmsnorm(float *h, const void *delta_bf16, const float *gamma, void *y_bf16, int M, int D, float eps,
                    void *stream);

/* im2col for the 14x14/stride-14 patch conv: images bf16 [F,3,224,224] -> patches bf16 [F*256, 640]
 *   (k = c*196 + ky*14 + kx, columns 588..639 zero) so that Conv2d becomes vly_gemm_bf16 with the
 *   zero-padded [1024,640] weight.  hf:clip/modeling_clip.py:209-210. */
int vly_patchify(const void *images_bf16, void *patches_bf16, int F, void *stream);

/* CLS concat + position add + pre_layrnorm: patch_out fp32 [F*256,1024] (the patch GEMM's fp32
 *   output) -> residual stream fp32 [F*257,1024].  hf:clip/modeling_clip.py:212-217 then :642. */
int vly_vit_embed_ln(const float *patch_out_f32, const float *cls, const float *pos,
                     const float *gamma, const float *beta, float *h_f32, int F, float eps, void *stream);

/* ViT self-attention, non-causal, no mask, 16 heads x 64, N = 257 tokens per frame:
 *   qkv bf16 [F*257, 3072] (q | k | v, biases already added) -> out bf16 [F*257, 1024].
 *   softmax in fp32, scale 64^-0.5.  qkv and out 16-byte aligned (16-byte loads and stores).
 *   hf:clip/modeling_clip.py:258-277, 317-330. */
int vly_vit_attention(const void *qkv_bf16, void *out_bf16, int F, void *stream);

/* Temporal pooling + per-frame CLS pick for B clips of T frames:
 *   feats fp32 [B,T,257,W] -> out bf16 [B, 256+T, W]; rows 0..255 = mean/max over T of patch
 *   tokens, rows 256.. = CLS token of each frame.  valley/model/valley_model.py:206-215.
 *   mode VLY_POOL_IMPORTANCE needs `scores` fp32 [B,T] (vly_temporal_scores); NULL otherwise. */
int vly_pool_tokens(const float *feats_f32, void *out_bf16, int B, int T, int W, int mode,
                    const float *scores, void *stream);

/* v2 temporal-importance frame scores: scores[f] = w . flatten(feats[f, 1:257, :]) + bias[0],
 *   feats fp32 [F,257,W], w fp32 [256*W].  valley/model/valley_model.py:42,115-116. */
int vly_temporal_scores(const float *feats_f32, const float *w, const float *bias, float *scores,
                        int F, int W, void *stream);

/* v3 "temporal transformer delta" pooling glue (valley/model/valley_model.py:123-133; the GEMMs of the
 *   1-layer nn.TransformerEncoderLayer(d_model=H, nhead=8, post-LN, ReLU FFN) go through vly_gemm_bf16):
 *   vly_delta_prep     feats fp32 [B,T,257,H] (projected), pos fp32 [>=T,H] ->
 *                      x_all bf16 [B*256*T, H] = patch[t,p] + pos[t] in (p-major, t-minor) order (:124-129),
 *                      x_last (bf16 and fp32) [B*256, H] = its t = T-1 rows, mean fp32 [B*256, H] (:131)
 *   vly_delta_attention  q bf16 [nseq,H] (last step only), kv bf16 [nseq*T, 2H] -> out bf16 [nseq,H];
 *                      softmax over the T keys, head_dim H/nhead <= 1024, T <= 32
 *   vly_delta_finish   out bf16 [B,256+T,H]: rows<256 = delta + mean (:132), rows>=256 = frame CLS (:215) */
int vly_delta_prep(const float *feats_f32, const float *pos_f32, void *x_all_bf16, void *x_last_bf16,
                   float *x_last_f32, float *mean_f32, int B, int T, int H, void *stream);
int vly_delta_attention(const void *q_bf16, const void *kv_bf16, void *out_bf16, int nseq, int T, int H,
                        int nhead, void *stream);
int vly_delta_finish(const float *delta_f32, const float *mean_f32, const float *feats_f32, void *out_bf16,
                     int B, int T, int H, void *stream);
/* The same three with fp32 operands throughout (the fp32 "precise" mode, round 4: nothing rounded; GEMMs through vly_gemm_f32,
 *   norms through vly_norm_f32).  x_last has no 16-bit twin here. */
int vly_delta_prep_f32(const float *feats_f32, const float *pos_f32, float *x_all_f32, float *x_last_f32, float *mean_f32,
                       int B, int T, int H, void *stream);
int vly_delta_attention_f32(const float *q_f32, const float *kv_f32, float *out_f32, int nseq, int T, int H, int nhead,
                            void *stream);
int vly_delta_finish_f32(const float *delta_f32, const float *mean_f32, const float *feats_f32, float *out_f32, int B, int T,
                         int H, void *stream);

/* Token-embedding gather + visual-token splice -> fp32 residual stream:
 *   row_map int32 [R]: v >= 0 -> embed_table[v];  v < 0 -> visual[-v-1].
 *   embed_table bf16 [V,H], visual bf16 [NV,H], out fp32 [R,H].
 *   valley/model/valley_model.py:160 and the row copies of :228,:242 (the integer index logic and
 *   its ValueErrors stay on the host, valley_amd/splice.py). */
int vly_embed_splice(const int32_t *row_map, const void *embed_bf16, const void *visual_bf16,
                     float *out_f32, int R, int H, void *stream);

/* vly_add_rmsnorm with two bf16 deltas (the split-K partials of vly_gemm_bf16_splitk2): h += d0 + d1; y = rmsnorm(h). */
int vly_add2_rmsnorm(float *h, const void *delta0_bf16, const void *delta1_bf16, const float *gamma, void *y_bf16,
                     int M, int D, float eps, void *stream);
/* vly_add_layernorm with two bf16 deltas (CLIP out_proj / fc2 split-K partials); gamma NULL = add only. */
int vly_add2_layernorm(float *h, const void *delta0_bf16, const void *delta1_bf16, const float *gamma, const float *beta,
                       void *y_bf16, int M, int D, float eps, void *stream);

/* RoPE (rotate-half) on q and k of a fused qkv buffer + KV-cache append:
 *   qkv bf16 [B*S, 3*heads*128]; q rotated in place; rotated k and v written to
 *   kcache/vcache bf16 [B, heads, ctx_max, 128] at positions past_len..past_len+S-1.
 *   position of row s = past_len + s (independent of padding).  cos/sin: fp32 tables
 *   [ctx_max, 64] indexed by absolute position (built by the host exactly as
 *   hf:llama/modeling_llama.py:95-124).  Rotation :127-157, KV append :261-262 (legacy tuple cat
 *   at the pinned commit).
 *   past_len_dev (nullable): device int32 that overrides past_len at execution time, so that a
 *   hipGraph-captured decode step can be replayed while the position advances on the device. */
int vly_rope_kv(void *qkv_bf16, void *kcache_bf16, void *vcache_bf16, const float *cos_table,
                const float *sin_table, int B, int S, int heads, int past_len, const int32_t *past_len_dev,
                int ctx_max, void *stream);

/* Causal self-attention over the KV cache, head_dim 128:
 *   q from qkv bf16 [B*S, 3*heads*128] (first third), K/V from the caches, kv_len = past_len+S.
 *   key_valid uint8 [B, key_valid_stride >= kv_len] or NULL (1 = attend; the padding half of HF's
 *   additive mask).  out bf16 [B*S, heads*128].  S == 1 takes the HBM-bound decode kernel (one
 *   workgroup per head streams the head's K and V once; serve/model_worker.py:380-387).
 *   past_len_dev: as in vly_rope_kv.
 *   hf:llama/modeling_llama.py:191-213, mask = causal AND padding (:386-397 / masking_utils). */
int vly_llama_attention(const void *qkv_bf16, const void *kcache_bf16, const void *vcache_bf16,
                        const uint8_t *key_valid, int key_valid_stride, void *out_bf16, int B, int S,
                        int heads, int past_len, const int32_t *past_len_dev, int ctx_max, void *stream);

/* HF's ``output_attentions`` for one layer: the softmax probabilities of vly_llama_attention's problem,
 *   out fp32 [B, heads, S, kv_len = past_len + S] (zero where a key is masked; a query with no visible key gets a row of
 *   zeros).  qkv holds the ROTATED q in its first third (after vly_rope_kv / vly_gemm_bf16_qkv_rope), kcache the rotated K.
 *   inputs_f32 = 1: fp32 q|k|v rows and fp32 cache (the precise engines).  kv_len <= 16000 (its scores live in LDS).  A separate pass, not a hot path.
 *   valley_model.py:281,324-330 -> hf:llama/modeling_llama.py:191-213 (eager attention returns attn_weights). */
int vly_llama_attention_probs(const void *qkv, const void *kcache, const uint8_t *key_valid, int key_valid_stride,
                              float *out, int B, int S, int heads, int past_len, int ctx_max, int inputs_f32, void *stream);

/* One-token decode step of a layer's attention, fused:  vly_rope_kv (S = 1) + vly_llama_attention (S = 1) in one
 *   launch.  qkv bf16 [B, 3*heads*128] holds the new token's UNROTATED q|k|v; the rotated k and v are appended
 *   to the caches at position past_len, the new query attends over positions 0..past_len.  Same arithmetic
 *   as the two separate calls (serve/model_worker.py:380-387 -> hf:llama/modeling_llama.py:127-157,191-213,
 *   261-262).  past_len_dev as in vly_rope_kv.  qkv is not modified. */
int vly_decode_attention(const void *qkv_bf16, void *kcache_bf16, void *vcache_bf16, const float *cos_table,
                         const float *sin_table, const uint8_t *key_valid, int key_valid_stride, void *out_bf16,
                         int B, int heads, int past_len, const int32_t *past_len_dev, int ctx_max, void *stream);

/* The same step for a batch whose rows are INDEPENDENT sequences at their own positions (continuous batching over one
 *   captured decode step, serve/model_worker.py:380-387 run for several requests at once): row b appends at
 *   past_len_rows[b] (device int32 [B], clamped to ctx_max-1) and attends over 0..past_len_rows[b]; key_valid rows
 *   (stride >= ctx_max) mask padding and retired slots. */
int vly_decode_attention_rows(const void *qkv_bf16, void *kcache_bf16, void *vcache_bf16, const float *cos_table,
                              const float *sin_table, const uint8_t *key_valid, int key_valid_stride, void *out_bf16,
                              int B, int heads, const int32_t *past_len_rows, int ctx_max, void *stream);

/* The same step split over the keys (flash-decoding): every head is VLY_DECODE_SPLITS workgroups, each over a 64-aligned
 *   quarter of the keys, so a batch-1 step keeps 4 x heads CUs busy instead of `heads`.  Writes, per (row, head, split),
 *   132 fp32 = {running max (log2 domain), sum of exponentials, 0, 0, unnormalised P·V[128]} to `partials`
 *   ([B][heads][VLY_DECODE_SPLITS][132], 16-byte aligned); vly_gemv_attnmerge_bf16 — the o projection — merges them.
 *   past_len_dev_stride: 0 = one device position for the batch (as vly_decode_attention), 1 = one per row
 *   (as vly_decode_attention_rows).  Same RoPE / append arithmetic as vly_decode_attention; the softmax sums are
 *   associated per split, so outputs agree with it to fp32 rounding, not bit for bit. */
#define VLY_DECODE_SPLITS 4
#ifdef VLY_EXPERIMENTAL      /* round 3's form: libvalley_hip_exp.so only (the merged launch below is the default) */
int vly_decode_attention_split(const void *qkv_bf16, void *kcache_bf16, void *vcache_bf16, const float *cos_table,
                               const float *sin_table, const uint8_t *key_valid, int key_valid_stride, float *partials,
                               int B, int heads, int past_len, const int32_t *past_len_dev, int past_len_dev_stride,
                               int ctx_max, void *stream);
#endif

/* The same launch, with the merge done by the LAST of a head's VLY_DECODE_SPLITS workgroups to finish (round 4): it combines the
 *   head's partials (the arithmetic of vly_gemv_attnmerge_bf16's prologue, in split order: bit-identical) and writes the
 *   attention output out[B, heads*128] in the 16-bit storage type, which a plain vly_gemv_bf16 (the o projection) then reads.
 *   arrivals: B * heads uint32 of device memory, zero before the first launch; every launch leaves them zero.  `partials` is
 *   still the workspace [B][heads][VLY_DECODE_SPLITS][132].  (HF LlamaAttention.forward behind serve/model_worker.py:380-387.) */
int vly_decode_attention_merged(const void *qkv_bf16, void *kcache_bf16, void *vcache_bf16, const float *cos_table,
                                const float *sin_table, const uint8_t *key_valid, int key_valid_stride, float *partials,
                                void *out_bf16, uint32_t *arrivals, int B, int heads, int past_len,
                                const int32_t *past_len_dev, int past_len_dev_stride, int ctx_max, void *stream);

/* Weight-streaming GEMV for decode (M <= 16 rows):  same contract as vly_gemm_bf16
 *   (epilogues, residual, out dtype) but HBM-bound by construction: every weight byte is read once.
 *   M <= 2: VALU dot products; 3 <= M <= 16: the same stream through MFMA 16x16x32 (sixteen weight rows x the M
 *   activation rows per instruction) when K % 64 == 0, N % 4 == 0 and the rows are 16-byte aligned — the step of
 *   several concurrent requests (the reference's worker admits 5, serve/model_worker.py:467-474) then costs what one
 *   request's does.  Each output row depends on its own activation row only.  M > 8 needs the MFMA form.
 *   serve/model_worker.py:380-387 (one-token forward). */
int vly_gemv_bf16(const void *A, const void *W, const float *bias, const float *residual, void *C,
                  int M, int N, int K, int lda, int ldw, int ldc, int ldr,
                  int epilogue, int out_dtype, void *stream);

/* The same GEMV with the RMSNorm that feeds it folded in:  C = epi(rmsnorm(H; gamma, eps) · W^T + bias) + residual,
 *   H the fp32 residual stream [M, K] (row stride ldh), M <= 2, 2048 <= K <= 6144.  Replaces the pairs
 *   input_layernorm -> q|k|v, post_attention_layernorm -> gate|up and norm -> lm_head of a decode step
 *   (HF LlamaDecoderLayer.forward / LlamaModel.norm + lm_head behind serve/model_worker.py:380-387): two launches and two
 *   kernel boundaries per layer less.  Bit-identical to vly_rmsnorm followed by vly_gemv_bf16.  C must not overlap H
 *   (-22): every workgroup re-reads the H row for the norm while others write C; `residual` may be any other buffer. */
int vly_gemv_rmsnorm_bf16(const float *H, const float *gamma, float eps, const void *W, const float *bias,
                          const float *residual, void *C, int M, int N, int K, int ldh, int ldw, int ldc, int ldr,
                          int epilogue, int out_dtype, void *stream);

/* The o projection of a decode step with the merge of vly_decode_attention_split's partials folded in:
 *   C = (merge(partials) as bf16 [M, heads*128]) · W^T + bias + residual, M <= 2 rows, 2048 <= K = heads*128 <= 6144.
 *   (HF LlamaAttention.forward: attn_output -> o_proj, behind serve/model_worker.py:380-387.) */
#ifdef VLY_EXPERIMENTAL      /* libvalley_hip_exp.so only */
int vly_gemv_attnmerge_bf16(const float *partials, const void *W, const float *bias, const float *residual, void *C,
                            int M, int N, int heads, int ldw, int ldc, int ldr, int out_dtype, void *stream);
#endif

/* fp32 -> bf16 (round-to-nearest-even) over n contiguous elements, n % 8 == 0: the `.to(dtype)`
 *   between an fp32 tensor and a GEMM input (only used on the `max`-pooling path, where the
 *   projector has to see every token, valley_model.py:190,209). */
int vly_cast_f32_bf16(const float *x, void *y_bf16, long n, void *stream);

/* Frame preprocessing in front of the path (valley/util/data_util.py:262-281): Pillow's 8-bit separable
 *   bilinear resample restricted to the centre crop, then /255 and CLIP mean/std.
 *   vly_resize_h_u8:   in u8 [T,H,W,3] -> out u8 [T,H,OW,3]; bounds int32 [OW,2] (first input x, tap
 *                      count), taps int32 [OW,ksize] (22-bit fixed point) for the OW kept columns.
 *   vly_resize_v_norm: in u8 [T,H,W,3] -> out bf16|fp32 [T,3,OS,OS]; vertical taps for the OS kept rows,
 *                      columns x_off .. x_off+OS-1, out = (clip8(acc>>22)/255 - mean[c]) / std[c]. */
int vly_resize_h_u8(const uint8_t *in, const int32_t *bounds, const int32_t *taps, uint8_t *out,
                    int T, int H, int W, int OW, int ksize, void *stream);
int vly_resize_v_norm(const uint8_t *in, const int32_t *bounds, const int32_t *taps, const float *mean,
                      const float *stdv, void *out, int T, int H, int W, int x_off, int OS, int ksize,
                      int out_f32, void *stream);

/* p[i] += delta for i < n (the device-side position counter of a captured decode step). */
int vly_incr_i32(int32_t *p, int n, int delta, void *stream);

/* argmax over the last dim of fp32 [M,N] (row stride ld >= N) -> int32 [M]; first maximal index.
 *   serve/model_worker.py:389-391. */
int vly_argmax(const float *x, int32_t *idx, int M, int N, int ld, void *stream);

/* C[M,N] = epi(A[M,K] W[N,K]^T + bias), bf16 out, for FEW rows (M <= 256; N % 32 == 0; K % 128 == 0, K % 512 == 0 from
 *   K = 2048 on; epilogue NONE, QUICK_GELU or
 *   RELU): the latency-optimised form of vly_gemm_bf16 for the F-row remainders of the tall ViT GEMMs — one 32x32 block
 *   of C per workgroup, K split over its four (K >= 2048: sixteen) waves, fragments loaded straight from global memory (gemm_skinny.hip).
 *   Same math as vly_gemm_bf16 up to fp32 summation order. */
int vly_gemm_skinny_bf16(const void *A_bf16, const void *W_bf16, const float *bias, void *C_bf16, int M, int N, int K,
                         int lda, int ldw, int ldc, int epilogue, void *stream);

/* ---- fp32 "precise" path (VALLEY_PRECISION=fp32): the same operators with fp32 tensors end to end and contractions on
 * the exact f32-input MFMA (v_mfma_f32_32x32x2_f32).  It exists to demonstrate the north star's "logits within 1e-3 of
 * the reference", which bf16 operands cannot reach; ~1/16 of the bf16 path's rate by construction. ------------------- */

/* C[M,N] = epi(A[M,K] . W[N,K]^T + bias) + residual, all fp32 (nn.Linear / the patch conv as a GEMM;
 *   hf:clip/modeling_clip.py:148-154,293-350, hf:llama/modeling_llama.py:160-173,230-241).  K % 16 == 0, N % 4 == 0,
 *   16-byte aligned rows; SWIGLU as in vly_gemm_bf16 (interleaved rows, N/2-wide output, no residual); residual may
 *   alias C. */
int vly_gemm_f32(const float *A, const float *W, const float *bias, const float *residual, float *C,
                 int M, int N, int K, int lda, int ldw, int ldc, int ldr, int epilogue, void *stream);

/* Split-operand images for the fp32 engines' GEMMs (round 6; VALLEY_F32_GEMM=x3): out3[M, 3 Kp] (16-bit storage) from fp32 x[M, K]
 *   with x = hi + lo, hi = rn16(x), lo = rn16(x - hi): order 0 -> [hi | hi | lo] (activations), order 1 -> [hi | lo | hi] (weights),
 *   every segment Kp >= K wide, pad columns zero.  vly_gemm_bf16 over K' = 3 Kp with fp32 output then equals the fp32 product up to
 *   2^-16 relative per term (a_hi w_hi + a_hi w_lo + a_lo w_hi, each exact in fp32) — the arithmetic of hf:clip/modeling_clip.py:293-350
 *   / hf:llama/modeling_llama.py:160-173,230-241 at a third of the 16-bit MFMA rate instead of a sixteenth.  epilogue applies the
 *   PRODUCING GEMM's activation first (vly_gemm_f32's expressions): NONE, QUICK_GELU, RELU, or SWIGLU (x then has 2 K interleaved
 *   (gate, up) columns).  K % 4 == 0, Kp % 4 == 0, ldx % 4 == 0, x 16-byte aligned. */
int vly_split3_f32(const float *x, int ldx, void *out3_half, int M, int K, int Kp, int epilogue, int order, void *stream);

/* out = softmax(q k^T * head_dim^-0.5 [+ mask]) v in fp32 (hf:clip/modeling_clip.py:259-277 without mask;
 *   hf:llama/modeling_llama.py:191-213 with causal + key-validity mask: key j visible to query i iff
 *   j <= i + past_len and key_valid[b][j]).  q / out row r of batch b, head h: base + b*batch_stride + r*row_stride + h*hd;
 *   k / v row j: base + b*kv_batch_stride + h*kv_head_stride + j*kv_row_stride (covers the q|k|v GEMM buffer of the
 *   ViT and the [B,heads,ctx,hd] fp32 KV cache).  head_dim 64 or 128.  Fully masked query rows come out as zeros. */
int vly_attention_f32(const float *q, long q_batch_stride, int q_row_stride, const float *k, const float *v,
                      long kv_batch_stride, long kv_head_stride, int kv_row_stride, const uint8_t *key_valid,
                      int key_valid_stride, float *out, long out_batch_stride, int out_row_stride, int B, int heads,
                      int n_q, int n_kv, int head_dim, int causal, int past_len, void *stream);

/* y = LayerNorm(x; gamma, beta, eps) (rms == 0) or gamma * x * rsqrt(mean(x^2) + eps) (rms != 0, beta unused), fp32
 *   [M,D] -> fp32 [M,D]; y may alias x.  hf:clip/modeling_clip.py:605,642; hf:llama/modeling_llama.py:51-67. */
int vly_norm_f32(const float *x, const float *gamma, const float *beta, float *y, int M, int D, float eps, int rms,
                 void *stream);

/* vly_norm_f32 fused with vly_split3_f32 (order 0, no activation): out3[M, 3 Kp] = [hi | hi | lo] of LayerNorm / RMSNorm(x) — the GEMM
 *   operand of the split-operand engine without the fp32 round trip through memory.  hf:clip/modeling_clip.py:605,642;
 *   hf:llama/modeling_llama.py:51-67. */
int vly_norm_split3_f32(const float *x, const float *gamma, const float *beta, void *out3_half, int M, int D, int Kp, float eps, int rms,
                        void *stream);

/* vly_rope_kv with fp32 q|k|v rows [B*S, 3*heads*128] and fp32 caches [B,heads,ctx_max,128] (hf:llama 127-157). */
int vly_rope_kv_f32(float *qkv, float *kcache, float *vcache, const float *cos_table, const float *sin_table,
                    int B, int S, int heads, int past_len, int ctx_max, void *stream);

/* vly_patchify at fp32: [F,3,224,224] -> [F*256, k_padded] (k_padded >= 588, a multiple of 16; pad columns zero). */
int vly_patchify_f32(const float *images, float *patches, int F, int k_padded, void *stream);

/* vly_pool_tokens with an fp32 result [B, 256+T, W] (valley_model.py:206-215, 113-121). */
int vly_pool_tokens_f32(const float *feats, float *out, int B, int T, int W, int mode, const float *scores, void *stream);

/* vly_embed_splice with an fp32 embedding table [V,H] and fp32 visual tokens (valley_model.py:160, 195-247). */
int vly_embed_splice_f32(const int32_t *row_map, const float *embed, const float *visual, float *out, int R, int H,
                         void *stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * EXPERIMENTAL entry points (these and the two #ifdef VLY_EXPERIMENTAL prototypes above): exported by libvalley_hip_exp.so / libvalley_hip_exp_f16.so only (the sources built with -DVLY_EXPERIMENTAL=1,
 * valley_amd/build.py; VALLEY_EXPERIMENTAL=1 makes the Python binding load it).  Built to parity and measured BEHIND the default
 * path they would replace, kept as tested experiments: no default path calls them, the shipped libraries
 * (libvalley_hip.so, libvalley_hip_f16.so) do not carry them.  That library also accepts tile hint 297 of
 * vly_gemm_bf16_streamk and the VLY_VIT_ATTN=4 / VLY_LLAMA_ATTN=1 kernels.
 * --------------------------------------------------------------------------------------------------------------------- */
#ifdef VLY_EXPERIMENTAL
/* ALL decoder layers of a batch-1/2 decode step in ONE persistent launch (round 4; decode_step.hip): per layer the five
 *   phases input_layernorm + q|k|v, RoPE + KV append + split attention, merge + o + residual, post_attention_layernorm +
 *   gate|up + SwiGLU, down + residual — the launches vly_gemv_rmsnorm_bf16 / vly_decode_attention_split /
 *   vly_gemv_attnmerge_bf16 / vly_gemv_rmsnorm_bf16 / vly_gemv_bf16 of a layer, with the same arithmetic (the step is
 *   BIT-identical to them) — separated by grid barriers inside the launch, the weight stream running across every barrier.
 *   Replaces the loop body of serve/model_worker.py:380-387 (one-token forward) x num_hidden_layers.
 *     layers_dev : DEVICE array of n_layers descriptors (weights in the layouts of the entry points above: q|k|v fused
 *                  [3H, H], o [H, H], gate|up row-interleaved [2I, H], down [H, I], 16-bit storage type, row-major,
 *                  16-byte aligned; the two RMSNorm weights fp32 [H]; this layer's K / V cache [B, heads, ctx_max, 128])
 *     h          : fp32 [B, H] residual stream, in (token embeddings) and out (input of the final norm)
 *     qkv_scratch (16-bit [B, 3H]), partials (fp32 [B, heads, VLY_DECODE_SPLITS, 132]), mlp_scratch (fp32 [B, I]): workspaces
 *     pos_dev    : device int32, the position of the new token (pos_stride 0: one value; 1: one per batch row)
 *     sync       : VLY_DECODE_SYNC_WORDS uint32 of device memory, 64-byte aligned, private to this stream and to this
 *                  (n_layers) — zeroed ONCE by the caller before the first launch; the barrier counters continue from launch to
 *                  launch.  After a launch completed, sync[VLY_DECODE_SYNC_ABORT] != 0 means a workgroup gave up waiting at a
 *                  grid barrier (not every workgroup was resident: another kernel was holding CUs) and h is invalid — every
 *                  wait is bounded, the launch always ends; zero the words again before the next launch.
 *   Needs the whole GPU: one 1024-thread workgroup per CU, all resident.  B <= 2, heads * 128 == H, (H, I) in the 7B / 13B
 *   classes (vly_decode_layers_supported); -22 otherwise. */
typedef struct vly_decode_layer {
    const void *w_qkv, *w_o, *w_gu, *w_down;
    const float *ln1, *ln2;
    void *kcache, *vcache;
} vly_decode_layer;
#define VLY_DECODE_SYNC_WORDS 512
#define VLY_DECODE_SYNC_ABORT 272
int vly_decode_layers_supported(int B, int H, int heads, int I);
int vly_decode_layers(const vly_decode_layer *layers_dev, int n_layers, float *h, void *qkv_scratch, float *partials,
                      float *mlp_scratch, const float *cos_table, const float *sin_table, const uint8_t *key_valid,
                      int key_valid_stride, const int32_t *pos_dev, int pos_stride, int B, int H, int heads, int I, float eps,
                      int ctx_max, uint32_t *sync, void *stream);
#endif  /* VLY_EXPERIMENTAL */

#ifdef __cplusplus
}
#endif
#endif /* VALLEY_HIP_H */
