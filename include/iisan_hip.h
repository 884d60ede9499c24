/*
 * iisan_hip.h — C ABI of libiisan_hip.so: the MI355X (gfx950) implementation of IISAN's data-parallel hot path.
 *
 * The reference (GAIR-Lab/IISAN) has no FFI: its "operator interface" for this path is the Python nn.Module
 * surface of the Code_Uncached and Code_Cached model/ packages.  Each entry point below replaces the arithmetic of one reference
 * forward (or the autograd backward PyTorch derives from it); the reference lines are cited per function.
 * `iisan_amd/` binds these with ctypes behind modules of the same names.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to row-major contiguous memory owned by the caller (16-byte aligned),
 *     except arguments documented as "host";
 *   - the library never allocates device memory, never synchronises and never owns buffers: scratch comes in
 *     through `ws`/`ws_bytes`, sized by the matching *_ws_bytes() query;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); functions are re-entrant per stream;
 *   - return 0 on success, IISAN_EBADSHAPE / IISAN_EWORKSPACE / IISAN_EHIP (<0) otherwise, with a message in
 *     iisan_last_error() (thread local);
 *   - `dtype16` selects the 16-bit MFMA operand type of the frozen encoders: IISAN_F16 (default; the reference
 *     itself runs them under fp16 autocast, Code_Uncached/run.py:409) or IISAN_BF16.  Accumulation, the
 *     residual stream, LayerNorm/softmax statistics, taps and everything trainable are fp32.
 */
#ifndef IISAN_HIP_H
#define IISAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IISAN_OK 0
#define IISAN_EBADSHAPE (-1)
#define IISAN_EWORKSPACE (-2)
#define IISAN_EHIP (-3)

#define IISAN_F16 0
#define IISAN_BF16 1
#define IISAN_F32 2   /* storage type of a packed tap store only (iisan_gather_taps) */

#define IISAN_MAX_LAYERS 48
#define IISAN_MAX_SIDE 16

const char* iisan_version(void);
const char* iisan_arch(void);            /* "gfx950" */
const char* iisan_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Frozen encoders (no-grad).  Weights are packed once by the host (iisan_amd/encoders.py): matrices in the 16-bit
 * operand type, [out,in] row-major exactly like torch.nn.Linear.weight; vectors in fp32.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* qkv_w;  const float* qkv_b;   /* [3D,D] 16-bit, [3D]  (q;k;v stacked)                          */
    const void* o_w;    const float* o_b;     /* [D,D], [D]                                                     */
    const void* fc1_w;  const float* fc1_b;   /* [F,D], [F]                                                     */
    const void* fc2_w;  const float* fc2_b;   /* [D,F], [D]                                                     */
    const float* ln1_w; const float* ln1_b;   /* ViT: layernorm_before;  BERT: attention.output.LayerNorm       */
    const float* ln2_w; const float* ln2_b;   /* ViT: layernorm_after;   BERT: output.LayerNorm                 */
    /* OPTIONAL fp32 masters of qkv_w / fc1_w (null = none).  The ViT executor with fp16 operands folds LayerNorm 1 / 2 into
     * these two matrices at the start of every call (gamma-scaled, centred rows); given the masters it rounds the
     * folded rows to fp16 ONCE instead of re-rounding the 16-bit copies (tap error against the reference 1.09e-3 -> see DESIGN). */
    const float* qkv_w32; const float* fc1_w32;
} iisan_layer_weights;

typedef struct {
    int32_t hidden, layers, heads, mlp;       /* D, L, H, F                                                     */
    int32_t image, patch, channels;           /* 224, 16, 3                                                     */
    int32_t dtype16;
    float   eps;
    int32_t full_blocks;                      /* dead-work policy, see below: 0 = default, 1 = every block on every token */
    const void*  patch_w;                     /* [D, C*P*P] 16-bit (Conv2d kernel flattened (c,py,px))          */
    const float* patch_b;                     /* [D]                                                            */
    const float* cls_token;                   /* [D]                                                            */
    const float* pos_emb;                     /* [T, D], T = (image/patch)^2 + 1                                */
    iisan_layer_weights layer[IISAN_MAX_LAYERS];
    /* OPTIONAL (null = none): the LayerNorm-folded QKV / FC1 weights of every layer, filled ONCE by iisan_vit_fold_layernorm into a
     * caller-owned device buffer of iisan_vit_fold_bytes() bytes.  Without it every forward call folds them again into its workspace
     * (300 MB of traffic, ~0.1 ms for ViT-B; the result is the same). */
    const void* folded;
} iisan_vit_weights;

typedef struct {
    int32_t hidden, layers, heads, mlp;
    int32_t vocab, max_pos;
    int32_t dtype16;
    float   eps;
    int32_t full_blocks;                      /* as in iisan_vit_weights                                        */
    const float* word_emb;                    /* [V, D] fp32 (gather only)                                      */
    const float* pos_emb;                     /* [max_pos, D]                                                   */
    const float* type_emb;                    /* [2, D] (row 0 used)                                            */
    const float* emb_ln_w; const float* emb_ln_b;
    iisan_layer_weights layer[IISAN_MAX_LAYERS];
} iisan_bert_weights;

/* Replaces Vit_Encoder.forward + `[:,0]` tap selection (Code_Uncached/model/encoders.py:29-31, model.py:210,212):
 * images fp32 [M,C,R,R] -> taps fp32 [M, n_taps, D], taps[:,k] = CLS row of hidden state tap_layers[k]
 * (0 = embeddings, l = output of layer l; the last one is before the final LayerNorm).  `tap_layers` is a host
 * array.  `chunk_items` (0 = whole batch) bounds the activation working set. */
/* The folded weights of `w` (LayerNorm 1 / 2 of every layer folded into qkv_w / fc1_w: gamma-scaled, centred, fp16, rounded
 * sum-preservingly from the fp32 masters when the layer has them, + the folded biases): [layers][(3D + F) x D] fp16, then
 * [layers][3D + F] fp32.  0 bytes when the executor would not use them (bf16 operands). */
size_t iisan_vit_fold_bytes(const iisan_vit_weights* w);
int iisan_vit_fold_layernorm(const iisan_vit_weights* w, void* folded, size_t bytes, void* stream);
/* fp16 operands (dtype16 = IISAN_F16), production sizes: the executor folds LayerNorm 1 / 2 of every block into the QKV / FC1
 * weights at the start of the call (workspace: +99 MB for ViT-B) and runs no LayerNorm / residual-add kernel inside blocks 1..L-1
 * (DESIGN 6g); the taps stay inside the same tolerance against the reference (tests/test_gpu_encoders.py). */
size_t iisan_vit_forward_taps_ws_bytes(const iisan_vit_weights* w, int64_t M, int64_t chunk_items);
int iisan_vit_forward_taps(const iisan_vit_weights* w, const float* images, int64_t M,
                           const int32_t* tap_layers, int32_t n_taps, float* taps,
                           int64_t chunk_items, void* ws, size_t ws_bytes, void* stream);
/* The same from RAW uint8 pixels [M,C,R,R] (0..255): ToTensor + Normalize(.5,.5) of the reference's transform
 * (Code_Uncached/data_utils/dataset.py:46-50) are applied in fp32 inside the patch-extraction kernel — taps identical to
 * the fp32 entry point fed with the normalised image, a quarter of the input bytes. */
int iisan_vit_forward_taps_u8(const iisan_vit_weights* w, const uint8_t* images, int64_t M,
                              const int32_t* tap_layers, int32_t n_taps, float* taps,
                              int64_t chunk_items, void* ws, size_t ws_bytes, void* stream);

/* Replaces Text_Encoder/Bert_Encoder.forward + tap selection (encoders.py:81-91,148-159, model.py:211,213):
 * text int64 [M, 2W] (W ids then W attention-mask values) -> taps fp32 [M, n_taps, D]. */
size_t iisan_bert_forward_taps_ws_bytes(const iisan_bert_weights* w, int64_t M, int32_t words, int64_t chunk_items);
int iisan_bert_forward_taps(const iisan_bert_weights* w, const int64_t* text, int64_t M, int32_t words,
                            const int32_t* tap_layers, int32_t n_taps, float* taps,
                            int64_t chunk_items, void* ws, size_t ws_bytes, void* stream);

/* Dead-work policy of the two executors above (`full_blocks` of the weights struct — part of the call, not process state).
 * 0 (default): blocks deeper than the deepest tapped hidden state are not run, and in the last live block attention / O /
 * MLP / LayerNorm run for the CLS row of every item only (K and V still for all tokens): the path consumes nothing but
 * `hidden_states[i][:, 0]`, so the taps are the same values.  1 = run every block on every token exactly like HF ViTModel /
 * BertModel do (what `bench.py` measures: SURVEY 8d counts that work). */

/* ------------------------------------------------------------------------------------------------------------
 * Side network (IISANAdaptedMModel.forward, Code_Uncached/model/model.py:209-271; Cached model.py:300-349).
 * All fp32.  Parameters arrive as a host array of device pointers in the order documented at
 * iisan_amd/ops.py::SIDE_PARAM_ORDER:
 *   for tower in (cv, text, mm): for k in 0..n_side-1: fc_down.weight, fc_down.bias, fc_up.weight, fc_up.bias
 *   gates: cv[0..n), text[0..n), mm[0..n)      (each a 1-element tensor; ignored when gated == 0)
 *   fc_cv.w, fc_cv.b, fc_bert.w, fc_bert.b, fc_mm.w, fc_mm.b,
 *   head_cv.w, head_cv.b (classifier / cv_pre_fc), head_text.w, head_text.b (title.fc / bert_pre_fc),
 *   fc_mm_down.w, fc_mm_down.b
 * Gradients are written to a parallel host array of device pointers (same order), ACCUMULATING (+=) into them;
 * the caller zeroes them.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_side;           /* SANBs of the image tower (7; 6 with remove_first); also of the text tower unless versa */
    int32_t dim_cv, dim_text; /* tap widths (768, 768; Versa e.g. 1024 / 8192)                                  */
    int32_t down;             /* adapter bottleneck (64)                                                        */
    int32_t emb;              /* embedding_dim (64)                                                             */
    int32_t gated;            /* fusion_method == "gated"                                                       */
    int32_t gelu;             /* adapter_activation == "GELU"                                                   */
    int32_t remove_first;     /* args.remove_first == "TRUE": states start at taps[:,first_index] (model.py:215-218) */
    int32_t tap_stride_cv;    /* taps are [M, tap_stride, D]; tap_index[k] selects the layer of SANB k          */
    int32_t tap_stride_text;
    int32_t tap_index[IISAN_MAX_SIDE];     /* tap-axis index used by image SANB k (and text SANB k unless versa) */
    int32_t first_index;                   /* tap index that seeds the states when remove_first                */
    /* ---- Versa (Code_Cached_Asym/model/model.py:257-429); all zero for the Uncached / Cached variants ---- */
    int32_t versa;            /* 1: asymmetric towers: heads are fc_cv[E,Dc], fc_bert[E,Dt], fc_mm[d,d] then cv_pre_fc[E,E],
                                 bert_pre_fc[E,E], fc_mm_down[E,d], d = min(Dc,Dt); group layer-drop; dim-align        */
    int32_t n_side_text;      /* SANBs of the text tower                                                        */
    int32_t tap_index_text[IISAN_MAX_SIDE];
    int32_t first_index_text;
    int32_t taps_exact16;     /* 1: the caller guarantees that every tap value is exactly representable in fp16 (taps cached in fp16, as
                                 preprocess_*.py of Code_Cached_Asym writes them): the split-operand dim-align products then take the tap
                                 with scale 1, skip its amax pass and (round 6) do not build the lo plane of its image — a tap that is NOT
                                 exact in fp16 would be rounded to fp16 there.  0 = unknown (always safe)                   */
} iisan_side_cfg;

/* Parameter table (host array of device pointers), n_mm = min(n_cv, n_text):
 *   image SANBs k=0..n_cv-1:  fc_down.weight, fc_down.bias, fc_up.weight, fc_up.bias
 *   text  SANBs k=0..n_t-1 :  same          mm SANBs i=0..n_mm-1: same
 *   gates: image[n_cv], text[n_t], mm[n_mm]   (1-element tensors; ignored when gated == 0)
 *   versa && dim_cv != dim_text: dim-align down_project_list[i].weight, .bias for i=0..n_mm-1
 *   fc_cv.w,.b   fc_bert.w,.b   fc_mm.w,.b
 *   head_cv.w,.b (classifier | cv_pre_fc)   head_text.w,.b (title.fc | bert_pre_fc)   fc_mm_down.w,.b          */
int32_t iisan_side_net_num_params(const iisan_side_cfg* cfg);
size_t iisan_side_net_ws_bytes(const iisan_side_cfg* cfg, int64_t M);        /* saved activations + scratch    */
/* out: item3 fp32 [M, 3*emb] = cat[cv, text, mm] (the com_dense input, model.py:69) */
/* `fwd_token` (out, host): what the backward call needs to know about THIS forward call — which kernel routes filled the
 * workspace.  The caller carries it to iisan_side_net_bwd together with the workspace (the library keeps no per-call state:
 * SURVEY 8b "stateless").  A backward call whose token does not match the routes it would take itself returns IISAN_EBADSHAPE. */
int iisan_side_net_fwd(const iisan_side_cfg* cfg, const float* taps_cv, const float* taps_text, int64_t M,
                       const void* const* params, float* item3, void* ws, size_t ws_bytes, uint64_t* fwd_token, void* stream);
int iisan_side_net_bwd(const iisan_side_cfg* cfg, const float* taps_cv, const float* taps_text, int64_t M,
                       const void* const* params, const float* d_item3, void* const* grads,
                       void* ws, size_t ws_bytes, uint64_t fwd_token, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Plain fp32 Linear (com_dense, Code_Uncached/model/model.py:36-37,69): y = x W^T + b ; bwd accumulates dW, db.
 * ---------------------------------------------------------------------------------------------------------- */
int iisan_linear_fwd(const float* x, const float* w, const float* b, float* y, int64_t M, int32_t K, int32_t N,
                     void* stream);
int iisan_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                     int64_t M, int32_t K, int32_t N, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * SASRec user encoder (User_Encoder.forward, Code_Uncached/model/encoders.py:60-65; modules.py:6-96).  fp32.
 * params order (host array of device pointers): position_embedding.weight, layer_norm.{weight,bias}, then per
 * block: w_Q, w_K, w_V, fc (weights), attn layer_norm.{weight,bias}, w_1.{weight,bias}, w_2.{weight,bias},
 * ffn layer_norm.{weight,bias}.   x: [B,S,E]; log_mask: [B,S]; y: [B,S,E].
 * With emb == 64, seq <= 16 and 1, 2 or 4 heads (the reference configuration is 64 / 10 / 2) each direction is ONE launch
 * (+ a fixed-order reducer of per-workgroup parameter-gradient partial sums in the backward: bit-reproducible, no atomics);
 * other shapes take one launch per operator.  Either way the forward keeps every intermediate the backward needs in `ws`
 * (same slots), and gradients ACCUMULATE (+=) into the caller's tensors.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t seq, emb, heads, blocks;       /* 10, 64, 2, 2                                                     */
    float   dropout;                        /* 0 => identity (eval)                                             */
    uint64_t seed;                          /* dropout stream (ignored when dropout == 0)                       */
} iisan_sasrec_cfg;

size_t iisan_sasrec_ws_bytes(const iisan_sasrec_cfg* cfg, int64_t B);
int iisan_sasrec_fwd(const iisan_sasrec_cfg* cfg, const float* x, const float* log_mask, int64_t B,
                     const void* const* params, float* y, void* ws, size_t ws_bytes, void* stream);
int iisan_sasrec_bwd(const iisan_sasrec_cfg* cfg, const float* x, const float* log_mask, int64_t B,
                     const void* const* params, const float* dy, float* dx, void* const* grads,
                     void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * In-batch debiased cross-entropy (ModelMM.forward, Code_Uncached/model/model.py:81-104), fused: the [T,M]
 * logits are never materialised.  ids int64 [bs*(S+1)], score fp32 [bs*(S+1),E], prec fp32 [bs*S,E],
 * log_mask fp32 [bs,S], pop_prob fp32 [n_pop = item_num+1].  loss: 1 float.  row_lse: [bs*S] scratch kept for bwd.
 * The library never syncs, so a bad INPUT VALUE cannot come back as a return code: an id outside [0, n_pop) is
 * not dereferenced and makes the loss NaN (the reference would raise an IndexError at model.py:63).
 * From 2^24 logits on (bs = 1024: 10,240 x 11,264) the two passes run on the 16-bit matrix cores with score / prec split
 * into fp16 hi + lo planes (csrc/ce.hip ce16_*: the accuracy of an fp32 product to ~2^-22; the workspace also holds the
 * operand images and the per-range partial results: ~95 MB at bs = 1024); below, on the f32-input matrix cores.
 * ---------------------------------------------------------------------------------------------------------- */
size_t iisan_inbatch_ce_ws_bytes(int64_t bs, int32_t S);
int iisan_inbatch_ce_fwd(const int64_t* ids, const float* score, const float* prec, const float* log_mask,
                         const float* pop_prob, int64_t n_pop, int64_t bs, int32_t S, int32_t E, float* loss,
                         void* ws, size_t ws_bytes, uint64_t* fwd_token, void* stream);
/* d_loss: host scalar multiplier (upstream gradient).  d_score [M,E] and d_prec [T,E] are overwritten.  `fwd_token`: the value
 * the forward call on this workspace returned (it says whether that call left d_prec for d_loss = 1 in the workspace, which
 * this call then only scales); 0 or a token of another call shape is IISAN_EBADSHAPE. */
int iisan_inbatch_ce_bwd(const int64_t* ids, const float* score, const float* prec, const float* log_mask,
                         const float* pop_prob, int64_t bs, int32_t S, int32_t E, float d_loss,
                         float* d_score, float* d_prec, void* ws, size_t ws_bytes, uint64_t fwd_token, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Eval scoring (eval_model + metrics_topK, Code_Uncached/data_utils/metrics.py:59-67,198-207): for each user
 * the 1-based rank of `target` among items 1..item_num by score = prec . item_emb, history scored -inf, ties
 * towards the lower item id.  history: int32 [U, hist_stride] padded with 0.  ranks: int32 [U]; a target outside
 * 1..item_num is not dereferenced and yields rank -1, history ids outside that range are ignored.
 * Limits (IISAN_EBADSHAPE otherwise): E == 64, hist_stride <= 256 (the history correction of a user lives in LDS), prec and
 * item_emb 16-byte aligned.
 * ---------------------------------------------------------------------------------------------------------- */
int iisan_score_rank(const float* prec, const float* item_emb, int64_t U, int64_t n_items_plus1, int32_t E,
                     const int32_t* history, int32_t hist_stride, const int32_t* target, int32_t* ranks,
                     void* stream);

/* The recommendation list itself (north_star "top-k indices"; SURVEY 8b `score_topk`): topk_ids int32 [U, k] = the first k item ids
 * of `order = torch.argsort(y_score, descending=True)` in metrics_topK (Code_Uncached/data_utils/metrics.py:59-60) over the row
 * eval_model builds at metrics.py:198-206 (history scored -inf, column 0 dropped, id = position + 1), ties towards the lower item id
 * (= a stable descending argsort; the reference leaves tie order unspecified).  Scores come out of the same MFMA chain as
 * iisan_score_rank's: a target that call ranks r <= k sits at topk_ids[u, r-1].  topk_scores fp32 [U, k] (nullable): the scores of
 * those items.  history ids outside 1..item_num are ignored, any hist_stride >= 0 is accepted.  When fewer than k items remain outside
 * a user's history the trailing slots hold id 0 / score -inf (the reference's argsort lists the -inf items there).
 * Limits (IISAN_EBADSHAPE otherwise): E == 64, 1 <= k <= 16, prec and item_emb 16-byte aligned; workspace from
 * iisan_score_topk_ws_bytes (0 bytes when one workgroup per 32 users covers the whole catalogue). */
size_t iisan_score_topk_ws_bytes(int64_t U, int64_t n_items_plus1, int32_t k);
int iisan_score_topk(const float* prec, const float* item_emb, int64_t U, int64_t n_items_plus1, int32_t E,
                     const int32_t* history, int32_t hist_stride, int32_t k, int32_t* topk_ids, float* topk_scores,
                     void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused Adam over a flat fp32 parameter buffer with per-segment learning rates (torch.optim.Adam defaults, the
 * optimiser of Code_Uncached/run.py:323-336).  seg_end (host, int64[n_seg]) are exclusive end offsets.
 * ---------------------------------------------------------------------------------------------------------- */
int iisan_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end,
                    const float* seg_lr, int32_t n_seg, int32_t step, float beta1, float beta2, float eps,
                    float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Primitive kernels, exported for unit tests and micro-benchmarks only.
 * ---------------------------------------------------------------------------------------------------------- */
/* C = epilogue(A[M,K] · W[N,K]^T + bias): 16-bit operands, fp32 accumulate.
 * mode 0: out16 = acc+bias     1: out16 = gelu_erf(acc+bias)     2: out32 = acc+bias+resid32 (resid may alias out)
 * A must be readable for ceil(M/256)*256 rows (tiles are loaded whole; rows >= M are never stored). */
int iisan_gemm16(int32_t dtype16, int32_t mode, const void* A, const void* W, const float* bias, void* out,
                 const float* resid, int64_t M, int32_t N, int32_t K, void* stream);
/* LayerNorm over rows of width 768: out16 (nullable) / out32 (nullable) */
int iisan_layernorm768(int32_t dtype16, const float* x, const float* g, const float* b, float eps,
                       void* out16, float* out32, int64_t rows, void* stream);
/* softmax(QK^T/sqrt(64) + key_bias)V per (item, head); qkv 16-bit HEAD-MAJOR [items, H, 3 (q|k|v), S, 64] (what the QKV
 * GEMM epilogue of the encoders writes); ctx 16-bit token-major [items*S, H*64];
 * key_bias fp32 [items,S] or NULL (values < 0 mark masked keys) */
int iisan_attention16(int32_t dtype16, const void* qkv, const float* key_bias, void* ctx, int64_t items, int32_t S,
                      int32_t heads, void* stream);
/* the same for the CLS query (token 0) of every item only: ctx_cls 16-bit [items, H*64].  Used by the encoder executors
 * in the last live block, where only `hidden_states[i][:, 0]` is consumed (Code_Uncached/model/model.py:210-213) */
int iisan_attention_cls16(int32_t dtype16, const void* qkv, const float* key_bias, void* ctx_cls, int64_t items,
                          int32_t S, int32_t heads, void* stream);
/* fp32 MFMA GEMM: C[M,N] = op(A) op(B) (+bias)(+relu); ta: A stored [K,M]; tb: B stored [K,N] (else [N,K]);
 * accumulate != 0: C += (atomic, split-K capable) */
int iisan_gemm32(const float* A, const float* B, const float* bias, float* C, int64_t M, int32_t N, int64_t K,
                 int32_t ta, int32_t tb, int32_t relu, int32_t accumulate, void* stream);
/* The same product on the 16-bit matrix cores with split operands (csrc/split.hip): each fp32 operand becomes two fp16
 * planes (hi + lo, power-of-two scale from the tensor's amax) and  A·B^T ~= Ah·Bh^T + Ah·Bl^T + Al·Bh^T  runs as one
 * fp16 MFMA GEMM over 3K with fp32 accumulation — a few fp32 ulps from the FMA chain, ~5x the f32 matrix rate.  Used by
 * the side network for its large Linear layers (fc_* 768x768, Versa dim-align).  No activation epilogue; N % 8 == 0;
 * operands 16-byte aligned; workspace from iisan_gemm_x3_ws_bytes. */
size_t iisan_gemm_x3_ws_bytes(int64_t M, int32_t N, int64_t K);
int iisan_gemm_x3(const float* A, const float* B, const float* bias, float* C, int64_t M, int32_t N, int64_t K,
                  int32_t ta, int32_t tb, int32_t accumulate, void* ws, size_t ws_bytes, void* stream);
/* Packed device-resident tap store (replaces the 22 `torch.load` calls per sample of
 * Code_Cached/data_utils/dataset.py:29-34,77-90): table [rows, row_elems] in fp32 / fp16 / bf16 (store_dtype), row i =
 * the selected CLS taps of item i flattened ([n_sel, D]); out[m, :] = fp32(table[ids[m], :]).  ids int64 on device,
 * clamped to [0, rows); row_elems must be a multiple of 8. */
int iisan_gather_taps(int32_t store_dtype, const void* table, int64_t rows, const int64_t* ids, float* out, int64_t M,
                      int64_t row_elems, void* stream);
/* f32 -> 16-bit conversion helper used when packing weights */
int iisan_cast16(int32_t dtype16, const float* src, void* dst, int64_t n, void* stream);

/* Kernel-level entry points of the encoder GEMM's epilogue families (tests/test_gpu_primitives.py, tools/gemm_*.py): the
 * products behind HF ViTLayer's QKV / O / FC1 / FC2 dense layers as the executors launch them. */
/* iisan_gemm16 with explicit leading dimensions (mode 0 / 1) */
int iisan_gemm16_ld(int32_t dtype16, int32_t mode, const void* A, const void* W, const float* bias, void* out, int64_t M,
                    int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldo, void* stream);
/* fp16 operands, fp32 output, optional split-K (ksplit > 1: atomic accumulation into a caller-zeroed out) */
int iisan_gemm16_f32(const void* A, const void* W, float* out, int64_t M, int32_t N, int32_t K, int32_t ksplit,
                     void* stream);
/* LN(x) W^T + b with the LayerNorm applied in the epilogue: A = x [M,K] fp16 (the residual stream), W / bias = the folded set
 * (iisan_fold_ln_weights), rowstat [ceil(M/256)*256] = rstd per row (NULL: plain product); mode 1 (GELU) or 4 (head-major QKV,
 * S tokens per item, N / 192 heads) */
int iisan_gemm16_lna(int32_t mode, const void* A, const void* W, const float* bias, void* out, const float* rowstat, int64_t M,
                     int32_t N, int32_t K, int32_t S, void* stream);
/* gamma-folded, centred weights of one LayerNorm -> Linear pair: Wf[n,:] = fp16(g * W[n,:] - mean_k(g * W[n,:])), bf = b_lin + W beta
 * (w32 != 0: W is the fp32 master, else the 16-bit copy) */
int iisan_fold_ln_weights(const void* W, int32_t w32, const float* bias, const float* g, const float* b, void* Wf, float* bf,
                          int32_t N, void* stream);
/* x16 <- fp16(x16 + A W^T + bias) in place (fp16 residual stream, N <= 1024) + per-64-column-slice row sums / sums of squares into
 * rowpart [N/64][Mpad][2]; rows m with m % S == 0 (CLS rows) receive the delta alone */
int iisan_gemm16_stream(const void* A, const void* W, const float* bias, void* x16, float* rowpart, int64_t M, int32_t N,
                        int32_t K, int32_t S, void* stream);
/* rowpart -> rstd per row; folds the CLS rows' deltas into their fp32 stream xc [items, N] */
int iisan_stream_stats_finalize(const float* rowpart, int32_t nslots, int64_t Mpad, void* x16, float* xc, float* rstat,
                                float eps, int64_t items, int32_t Ttok, void* stream);
/* host-side predicate (no device touched): does the persistent 256x256 kernel take this product? */
int32_t iisan_gemm16_h256_applicable(int32_t mode, int64_t M, int32_t N, int32_t K, int32_t qkv_S, int32_t qkv_heads,
                                     int32_t qkv_which0);

/* ------------------------------------------------------------------------------------------------------------
 * DEV section — process-wide development switches and measurement hooks.  NOT part of the product contract: a
 * product process calls none of these and the library defaults ARE the product routes.  Every other entry point
 * above is stateless and re-entrant per stream; these are the only process-global state the library has.
 * ---------------------------------------------------------------------------------------------------------- */
/* Named switches (kernel-family selection for golden-pinned reference runs, ablation bits, A/B routes).  Names are
 * registered by the kernel files (csrc/common.h: IISAN_DEV_KNOB); iisan_dev_state(buf, cap, 1) lists them all.
 * set: 0 or IISAN_EBADSHAPE (unknown name).  get: the value, INT64_MIN for an unknown name.
 * state: writes "name=value,..." of every switch NOT at its library default (all != 0: every switch) and returns the
 * length needed; "" means the product routes are in force.  reset: every switch back to its default.
 * Names of the form "count:<kernel family>" (round 6) are launch COUNTERS, not switches: get reads, set / reset zero them, state lists them only
 * with all != 0.  tests/test_gpu_trainable.py pins the default dispatch of the bench shapes with them. */
int32_t iisan_dev_set(const char* name, int64_t value);
int64_t iisan_dev_get(const char* name);
size_t iisan_dev_state(char* buf, size_t cap, int32_t all);
void iisan_dev_reset(void);
/* Per-launch HIP-event timing of one kernel class on the launching stream (bench.py's roofline.dominant_kernel):
 * cls 0 = off, 1 = the 16-bit encoder GEMM, 2 = the f32-matrix-core family of the trainable side.  only_stream: with on != 0
 * only launches on `stream` are timed.  collect: waits for the recorded events, returns the launch count, fills total
 * milliseconds and FLOPs; last_bytes: algorithmic bytes of the launches of the last collect. */
void iisan_timing_enable(int32_t cls);
void iisan_timing_only_stream(void* stream, int32_t on);
int64_t iisan_timing_collect(double* total_ms, double* total_flops);
double iisan_timing_last_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* IISAN_HIP_H */
