// Host-side executors of the two frozen encoders: enqueue the kernel chain of one ViT / BERT forward on the
// caller's stream and write the per-layer CLS taps in one pass (no hidden state other than the live one is kept).
// Replaces Vit_Encoder.forward / Bert_Encoder.forward + `hidden_states[i][:,0]`
// (Code_Uncached/model/encoders.py:29-31,81-91,148-159; model.py:210-213) — and, run over a catalogue, what
// Code_Cached/preprocess_vectors.py:68-112 precomputes.
#include "common.h"

int launch_vit_im2col(int dtype16, const void* img, int img_u8, void* out, int64_t M, int C, int R, int p, hipStream_t s);
int launch_vit_cls_rows(float* X, const float* cls, const float* pos, int64_t M, int T, int D, hipStream_t s);
int launch_bert_embed_ln(int dtype16, const int64_t* text, const float* word, const float* pos, const float* type0,
                         const float* g, const float* b, float eps, float* X, void* H, float* key_bias, int64_t M,
                         int W, int vocab, hipStream_t s, void* X16 = nullptr, float* Xc = nullptr);
int launch_gather_cls(const float* X, float* taps, int64_t M, int T, int D, int n_taps, int k, hipStream_t s);
int launch_gather_rows16(const void* H, void* out, int64_t M, int T, int D, hipStream_t s);   // out[m] = H[m*T] (16-bit rows)

namespace {

struct EncBufs {
    float* X;       // [Tp, D] fp32 residual stream (resid32 mode; in the default mixed mode: the patch embedding's output only,
                    // aliased onto QKV, which is first written after block 0's LN1 has consumed it)
    void* X16;      // [Tp, D] fp16: residual stream of the non-CLS rows (mixed mode)
    float* Xc;      // [Mcp, D] fp32: residual stream of the CLS rows, compact (mixed mode) = the hidden states' tapped rows
    void* H;        // [Tp, D] 16-bit: LayerNorm output, reused as the attention context
    void* QKV;      // [Tp, 3D] 16-bit
    void* F1;       // [Tp, max(F, patch_dim)] 16-bit: GELU(FC1) (and the patch matrix during embedding)
    void* D16;      // [Tp, D] 16-bit: output of the O GEMM (a residual DELTA, added to the fp32 stream by LayerNorm kernels)
    void* D16b;     // [Tp, D] 16-bit: output of the FC2 GEMM
    float* KB;      // [Mc, W] key bias (BERT)
    void* Cls;      // [2, Mcp, D] 16-bit: CLS rows of LN(x) and their Q projection in the CLS-only last block
    int64_t Mcp;    // items per chunk rounded up to a GEMM row tile
    // LayerNorm applied by the consuming product (ViT, fp16, mixed stream; g_ln_fold):
    void* Wf;       // [layers][3D + F, D] fp16: gamma-folded, centred QKV and FC1 weights
    float* Bf;      // [layers][3D + F]: folded biases
    float* RS;      // [Tp]: rstd of every row of the stream
    float* RP;      // [D / 64][Tp][2]: per-slice row sums of the products that add into the stream (EPI_STREAM16)
};

// 1 = keep the whole residual stream in fp32 (rounds 1-3); 0 (default) = fp32 for the CLS rows, fp16 for the others (rowops.hip).
// Bench / test knob (tools/enc_time.py A/B, tests/test_gpu_encoders.py); the *_ws_bytes queries follow it.
int g_resid32 = 0;
// 1 (default): the pre-LN tower (ViT) with fp16 operands and the mixed stream never materialises LayerNorm(x): the add kernels write
// the stream and rstd per row, the QKV / FC1 products read the stream itself against gamma-folded, centred weights and apply
// rstd in their epilogues (Gemm16Args::rowstat) — 12 instead of 16 bytes per token row and block in the HBM-bound kernels.
// 2 (default): ... and no add kernel either: the O / FC2 products add into the fp16 stream in their epilogues (EPI_STREAM16) and leave
// per-slice row sums, a small kernel turns those into rstd and folds the CLS rows' deltas into their fp32 stream.
// 0: the LayerNorm image of rounds 1-4.  Bench / test knob.
int g_ln_fold = 2;

size_t carve(WsCarver& c, EncBufs& b, int64_t tokens, int64_t items, int D, int F, int64_t kb_elems, int fold_layers = 0, bool fold_ws = true) {
    const int64_t Tp = ceil_div(tokens, 256) * 256;     // GEMM A operands are read in 256-row tiles
    b.Mcp = ceil_div(items, 256) * 256;
    b.Cls = c.take<uint16_t>((size_t)2 * b.Mcp * D);
    b.Xc = c.take<float>((size_t)b.Mcp * D);
    b.X16 = c.take<uint16_t>((size_t)Tp * D);
    b.X = g_resid32 ? c.take<float>((size_t)Tp * D) : nullptr;
    b.H = c.take<uint16_t>((size_t)Tp * D);
    b.QKV = c.take<uint16_t>((size_t)Tp * 3 * D);
    if (!b.X) b.X = (float*)b.QKV;                      // 6 bytes per element >= the 4 the fp32 embeddings need
    b.F1 = c.take<uint16_t>((size_t)Tp * F);
    b.D16 = c.take<uint16_t>((size_t)Tp * D);
    b.D16b = c.take<uint16_t>((size_t)Tp * D);
    b.KB = c.take<float>((size_t)(kb_elems > 0 ? kb_elems : 1));
    b.Wf = nullptr; b.Bf = nullptr; b.RS = nullptr; b.RP = nullptr;
    if (fold_layers > 0) {
        if (fold_ws) {          // (not when the caller's weights struct carries the folded set)
            b.Wf = c.take<uint16_t>((size_t)fold_layers * (3 * D + F) * D);
            b.Bf = c.take<float>((size_t)fold_layers * (3 * D + F));
        }
        b.RS = c.take<float>((size_t)Tp);
        b.RP = c.take<float>((size_t)(D / 64) * Tp * 2);
    }
    return c.off;
}

Gemm16Args gemm_args(int mode, const void* A, int K, const void* W, const float* bias, void* out, int N, int64_t M, int qkv_S = 0,
                     int qkv_heads = 0, int qkv_which0 = 0, const float* rowstat = nullptr) {
    Gemm16Args a{};
    a.A = A; a.W = W; a.bias = bias; a.out = out;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldo = N;
    a.qkv_S = qkv_S; a.qkv_heads = qkv_heads; a.qkv_which0 = qkv_which0;
    a.rowstat = rowstat;
    return a;
}

int gemm(int dt, int mode, const void* A, int K, const void* W, const float* bias, void* out, int N, const float* resid,
         int64_t M, hipStream_t s, const float* pos = nullptr, int patch_P = 0, int qkv_S = 0, int qkv_heads = 0,
         int qkv_which0 = 0) {
    Gemm16Args a = gemm_args(mode, A, K, W, bias, out, N, M, qkv_S, qkv_heads, qkv_which0);
    a.resid = resid; a.pos = pos; a.patch_P = patch_P;
    return launch_gemm16(dt, mode, a, s);
}

// LN(x) W^T + b with the LayerNorm in the epilogue: A = the stream, W / bias = the folded set, rowstat = the rows' rstd
int gemm_ln(int dt, int mode, const void* X16, int K, const void* Wf, const float* bf, const float* rs, void* out, int N,
            int64_t M, hipStream_t s, int qkv_S = 0, int qkv_heads = 0, int qkv_which0 = 0) {
    return launch_gemm16(dt, mode, gemm_args(mode, X16, K, Wf, bf, out, N, M, qkv_S, qkv_heads, qkv_which0, rs), s);
}

// Last live block: K and V for every token (head-major; the q third of the buffer is left untouched), Q for the CLS rows
// only.  Hc (in) / Qc (out) are compact [mc, D] 16-bit views: Hc = the CLS rows of the block's 16-bit input H.
int kv_all_q_cls(int dt, const iisan_layer_weights& L, const void* H, void* QKV, const void* Hc, void* Qc, int64_t tok,
                 int64_t mc, int T, int heads, int D, hipStream_t s) {
    const size_t e16 = 2;
    IISAN_TRY(gemm(dt, EPI_QKVH16, H, D, (const char*)L.qkv_w + (size_t)D * D * e16, L.qkv_b + D, QKV, 2 * D, nullptr, tok, s,
                   nullptr, 0, T, heads, 1));
    IISAN_TRY(gemm(dt, EPI_OUT16, Hc, D, L.qkv_w, L.qkv_b, Qc, D, nullptr, mc, s));
    return IISAN_OK;
}

int tap_index(const int32_t* tap_layers, int n, int layer) {
    for (int k = 0; k < n; ++k)
        if (tap_layers[k] == layer) return k;
    return -1;
}

// process-wide OVERRIDE of the weights structs' `full_blocks` field (bench.py, tests; not declared in include/iisan_hip.h):
// 1 = run every block of the tower on every token, as the reference does
int g_full_blocks = 0;

int max_tap(const int32_t* tap_layers, int n) {
    int m = 0;
    for (int k = 0; k < n; ++k) m = tap_layers[k] > m ? tap_layers[k] : m;
    return m;
}

int check_common(int hidden, int layers, int heads, int mlp, int n_taps, const int32_t* tap_layers) {
    IISAN_CHECK_SHAPE(hidden == 768, "encoder hidden size must be 768 (got %d): the side network is 768 wide "
                      "(Code_Uncached/model/model.py:171)", hidden);
    IISAN_CHECK_SHAPE(heads > 0 && hidden == heads * 64, "encoder head_dim must be 64 (hidden %d, heads %d)", hidden, heads);
    IISAN_CHECK_SHAPE(layers >= 1 && layers <= IISAN_MAX_LAYERS, "encoder layers %d out of range", layers);
    IISAN_CHECK_SHAPE(mlp % 128 == 0, "encoder mlp size %d must be a multiple of 128", mlp);
    IISAN_CHECK_SHAPE(n_taps >= 1 && tap_layers, "need at least one tap layer");
    for (int k = 0; k < n_taps; ++k)
        IISAN_CHECK_SHAPE(tap_layers[k] >= 0 && tap_layers[k] <= layers, "tap layer %d outside 0..%d", tap_layers[k], layers);
    return IISAN_OK;
}

}  // namespace

IISAN_DEV_KNOB(full_blocks, g_full_blocks);
IISAN_DEV_KNOB(resid32, g_resid32);
IISAN_DEV_KNOB(ln_fold, g_ln_fold);
static int vit_fold_layers(const iisan_vit_weights* w) {
    return (!g_resid32 && g_ln_fold && w->dtype16 == IISAN_F16) ? w->layers : 0;
}
// the folded set of one tower: [layers][(3D + F) x D] fp16, then [layers][3D + F] fp32 (the caller's `folded` buffer or the workspace)
static size_t vit_fold_w_elems(const iisan_vit_weights* w) { return (size_t)(3 * w->hidden + w->mlp) * w->hidden; }
static int vit_fold_all(const iisan_vit_weights* w, int layers, bool last_fc1, char* Wf, float* Bf, hipStream_t s) {
    const int D = w->hidden, F = w->mlp;
    LnFoldJob jobs[32];
    int nj = 0;
    for (int l = 0; l < layers; ++l) {
        const iisan_layer_weights& L = w->layer[l];
        char* wq = Wf + (size_t)l * vit_fold_w_elems(w) * 2;
        float* bq = Bf + (size_t)l * (3 * D + F);
        jobs[nj++] = LnFoldJob{L.qkv_w32 ? (const void*)L.qkv_w32 : L.qkv_w, L.qkv_b, L.ln1_w, L.ln1_b, wq, bq, 3 * D, L.qkv_w32 ? 1 : 0};
        if (l + 1 < layers || last_fc1)
            jobs[nj++] = LnFoldJob{L.fc1_w32 ? (const void*)L.fc1_w32 : L.fc1_w, L.fc1_b, L.ln2_w, L.ln2_b, wq + (size_t)3 * D * D * 2, bq + 3 * D, F, L.fc1_w32 ? 1 : 0};
        if (nj >= 31 || l + 1 == layers) { IISAN_TRY(launch_fold_ln_weights(jobs, nj, s)); nj = 0; }
    }
    return IISAN_OK;
}

extern "C" size_t iisan_vit_fold_bytes(const iisan_vit_weights* w) {
    if (w->dtype16 != IISAN_F16 || w->hidden != 768) return 0;
    return (size_t)w->layers * (vit_fold_w_elems(w) * 2 + (size_t)(3 * w->hidden + w->mlp) * 4);
}

extern "C" int iisan_vit_fold_layernorm(const iisan_vit_weights* w, void* folded, size_t bytes, void* stream) {
    const size_t need = iisan_vit_fold_bytes(w);
    IISAN_CHECK_SHAPE(need > 0, "vit_fold_layernorm: fp16 operands and hidden size 768 only");
    if (!folded || bytes < need) {
        iisan_set_error("vit_fold_layernorm: buffer too small (%zu < %zu)", bytes, need);
        return IISAN_EWORKSPACE;
    }
    IISAN_CHECK_SHAPE(w->layers >= 1 && w->layers <= IISAN_MAX_LAYERS, "vit_fold_layernorm: layers %d out of range", w->layers);
    return vit_fold_all(w, w->layers, true, (char*)folded, (float*)((char*)folded + (size_t)w->layers * vit_fold_w_elems(w) * 2), (hipStream_t)stream);
}

extern "C" size_t iisan_vit_forward_taps_ws_bytes(const iisan_vit_weights* w, int64_t M, int64_t chunk_items) {
    const int64_t Mc = (chunk_items > 0 && chunk_items < M) ? chunk_items : M;
    const int P = (w->image / w->patch) * (w->image / w->patch);
    const int pd = w->channels * w->patch * w->patch;
    WsCarver c(nullptr, 0);
    EncBufs b;
    return carve(c, b, Mc * (P + 1), Mc, w->hidden, w->mlp > pd ? w->mlp : pd, 0, vit_fold_layers(w), w->folded == nullptr);
}

static int vit_forward_taps_impl(const iisan_vit_weights* w, const void* images, int img_u8, int64_t M,
                                 const int32_t* tap_layers, int32_t n_taps, float* taps, int64_t chunk_items,
                                 void* ws, size_t ws_bytes, void* stream);

extern "C" int iisan_vit_forward_taps(const iisan_vit_weights* w, const float* images, int64_t M,
                                      const int32_t* tap_layers, int32_t n_taps, float* taps, int64_t chunk_items,
                                      void* ws, size_t ws_bytes, void* stream) {
    return vit_forward_taps_impl(w, images, 0, M, tap_layers, n_taps, taps, chunk_items, ws, ws_bytes, stream);
}

extern "C" int iisan_vit_forward_taps_u8(const iisan_vit_weights* w, const uint8_t* images, int64_t M,
                                         const int32_t* tap_layers, int32_t n_taps, float* taps, int64_t chunk_items,
                                         void* ws, size_t ws_bytes, void* stream) {
    IISAN_CHECK_SHAPE(w->image % 8 == 0, "vit u8: image side %d must be a multiple of 8", w->image);
    return vit_forward_taps_impl(w, images, 1, M, tap_layers, n_taps, taps, chunk_items, ws, ws_bytes, stream);
}

static int vit_forward_taps_impl(const iisan_vit_weights* w, const void* images, int img_u8, int64_t M,
                                 const int32_t* tap_layers, int32_t n_taps, float* taps, int64_t chunk_items,
                                 void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    IISAN_CHECK_SHAPE(M > 0, "vit_forward_taps: M must be positive");
    IISAN_TRY(check_common(w->hidden, w->layers, w->heads, w->mlp, n_taps, tap_layers));
    IISAN_CHECK_SHAPE(w->patch % 8 == 0 && w->image % w->patch == 0, "vit: patch %d / image %d unsupported", w->patch, w->image);
    const int D = w->hidden, F = w->mlp, P = (w->image / w->patch) * (w->image / w->patch), T = P + 1;
    const int pd = w->channels * w->patch * w->patch;
    IISAN_CHECK_SHAPE(pd % 64 == 0, "vit: patch vector length %d must be a multiple of 64", pd);
    IISAN_CHECK_SHAPE(T <= 224, "vit: %d tokens per image > 224 not supported", T);
    const int64_t Mc = (chunk_items > 0 && chunk_items < M) ? chunk_items : M;
    WsCarver c(ws, ws_bytes);
    EncBufs b;
    carve(c, b, Mc * T, Mc, D, F > pd ? F : pd, 0, vit_fold_layers(w), w->folded == nullptr);
    if (w->folded && b.RS) {       // the caller folded once (iisan_vit_fold_layernorm)
        b.Wf = const_cast<void*>(w->folded);
        b.Bf = (float*)((char*)b.Wf + (size_t)w->layers * vit_fold_w_elems(w) * 2);
    }
    if (c.overflow || !ws) {
        iisan_set_error("vit_forward_taps: workspace too small (%zu < %zu)", ws_bytes, c.off);
        return IISAN_EWORKSPACE;
    }
    const int dt = w->dtype16;
    const bool full_blocks = g_full_blocks || w->full_blocks;
    const int live = full_blocks ? w->layers : max_tap(tap_layers, n_taps);
    // LayerNorm in the epilogues of the QKV / FC1 products (g_ln_fold): only where those products run on the kernel that has it, for
    // every chunk size of this call (the last chunk may be shorter)
    bool lna = b.Wf != nullptr && live > 0;
    for (int64_t mc : {Mc, M % Mc == 0 ? Mc : M % Mc}) {
        const int64_t tok = mc * T;
        lna = lna && gemm16_takes_rowstat(dt, EPI_QKVH16, gemm_args(EPI_QKVH16, nullptr, D, nullptr, nullptr, nullptr, 3 * D, tok, T, w->heads)) &&
              gemm16_takes_rowstat(dt, EPI_GELU16, gemm_args(EPI_GELU16, nullptr, D, nullptr, nullptr, nullptr, F, tok)) &&
              (full_blocks || gemm16_takes_rowstat(dt, EPI_QKVH16, gemm_args(EPI_QKVH16, nullptr, D, nullptr, nullptr, nullptr, 2 * D, tok, T, w->heads, 1)));
    }
    // ... and the residual adds in the epilogues of the O / FC2 products
    bool lnb = lna && g_ln_fold >= 2;
    for (int64_t mc : {Mc, M % Mc == 0 ? Mc : M % Mc}) {
        const int64_t tok = mc * T;
        Gemm16Args o = gemm_args(EPI_STREAM16, nullptr, D, nullptr, nullptr, nullptr, D, tok, T), f = gemm_args(EPI_STREAM16, nullptr, F, nullptr, nullptr, nullptr, D, tok, T);
        o.rowpart = f.rowpart = b.RP;
        lnb = lnb && gemm16_runs_h256(dt, EPI_STREAM16, o) && gemm16_runs_h256(dt, EPI_STREAM16, f);
    }
    // x += A W^T + bias in the stream (fp16 rows in place, the CLS rows' fp32 stream through the finalize step); rstd of the new rows
    auto gemm_stream = [&](const void* A, int K, const void* W, const float* bias, int64_t mc) {
        const int64_t tok = mc * T;
        Gemm16Args a = gemm_args(EPI_STREAM16, A, K, W, bias, b.X16, D, tok, T);
        a.rowpart = b.RP;
        IISAN_TRY(launch_gemm16(dt, EPI_STREAM16, a, s));
        return launch_stream_stats_finalize(b.RP, D / 64, ceil_div(tok, 256) * 256, b.X16, b.Xc, b.RS, w->eps, mc, T, s);
    };
    const size_t fold_w = (size_t)(3 * D + F) * D;          // elements of one layer's folded weights
    auto Wf_qkv = [&](int l) { return (char*)b.Wf + (size_t)l * fold_w * 2; };
    auto Wf_fc1 = [&](int l) { return Wf_qkv(l) + (size_t)3 * D * D * 2; };
    auto bf_qkv = [&](int l) { return b.Bf + (size_t)l * (3 * D + F); };
    auto bf_fc1 = [&](int l) { return bf_qkv(l) + 3 * D; };
    if (lna && !w->folded) IISAN_TRY(vit_fold_all(w, live, full_blocks, (char*)b.Wf, b.Bf, s));
    const int64_t img_elems = (int64_t)w->channels * w->image * w->image;
    for (int64_t m0 = 0; m0 < M; m0 += Mc) {
        const int64_t mc = (M - m0 < Mc) ? M - m0 : Mc;
        const int64_t tok = mc * T;
        float* tp = taps + m0 * n_taps * D;
        const void* img0 = img_u8 ? (const void*)((const uint8_t*)images + m0 * img_elems)
                                  : (const void*)((const float*)images + m0 * img_elems);
        IISAN_TRY(launch_vit_im2col(dt, img0, img_u8, b.F1, mc, w->channels, w->image, w->patch, s));
        const bool mixed = !g_resid32;
        // patch embedding: resid32 = fp32 token rows with the position embedding added (128x128 kernel); mixed = 16-bit rows on the
        // production GEMM (gemm16_h256, 0.40 against 0.59 ms at bs = 128), the position table is added by block 0's LayerNorm
        if (mixed) IISAN_TRY(gemm(dt, EPI_PATCH16, b.F1, pd, w->patch_w, w->patch_b, b.D16b, D, nullptr, mc * P, s, nullptr, P));
        else IISAN_TRY(gemm(dt, EPI_PATCH32, b.F1, pd, w->patch_w, w->patch_b, b.X, D, nullptr, mc * P, s, w->pos_emb, P));
        // CLS rows: cls + pos[0] — into the fp32 stream (token-major) or into the compact fp32 CLS stream
        IISAN_TRY(launch_vit_cls_rows(mixed ? b.Xc : b.X, w->cls_token, w->pos_emb, mc, mixed ? 1 : T, D, s));
        // the tapped rows of the current hidden state: CLS rows of X (stride T) or the compact CLS stream itself
        auto tap = [&](int k) { return mixed ? launch_gather_cls(b.Xc, tp, mc, 1, D, n_taps, k, s) : launch_gather_cls(b.X, tp, mc, T, D, n_taps, k, s); };
        int k = tap_index(tap_layers, n_taps, 0);
        if (k >= 0) IISAN_TRY(tap(k));
        // The O / FC2 GEMMs emit 16-bit deltas; the residual add is fused into the NEXT LayerNorm kernel (HBM-bound) instead
        // of a read-modify-write GEMM epilogue.  `pending` = a delta not yet added to the stream.
        // Residual bookkeeping (pre-LN tower): LN2 computes LN(x + dO) WITHOUT writing x back; the next LN1 adds both deltas,
        // (x + dO) + dF in fp32 — the same value — and writes x once per block instead of twice.
        const void* pend_o = nullptr;     // deltas not yet added to the stream
        const void* pend_f = nullptr;
        // Blocks after the deepest tapped hidden state are dead code (Versa configurations tap a prefix of the tower),
        // and in the last LIVE block only the CLS token's output is consumed: K/V are computed for every token, but
        // attention, O, LN2, FC1, FC2 and the closing residual add run on one row per item.
        constexpr int MX_ADD_STAT = MX_D1 | MX_RESV | MX_STAT;       // x += d (fp16 stream, fp32 CLS rows); row statistics
        for (int l = 0; l < live; ++l) {
            const iisan_layer_weights& L = w->layer[l];
            // x += pending deltas of block l-1 ; h = LN1(x)            -> the stream is hidden state l
            if (lna) {
                // ... without the image h: the stream + (rstd, -mean rstd) per row; LN1 is applied by the QKV product's epilogue.
                // (x + dO is in the stream already: this block's predecessor wrote it in its LN2 step)
                if (l == 0)
                    IISAN_TRY(launch_layernorm768_mixed(dt, MX_SRC32 | MX_POSROW | MX_ADD_STAT, w->pos_emb, b.X16, b.Xc, b.D16b, nullptr, nullptr, nullptr, w->eps, nullptr, mc, T, s, b.RS));
                else if (!lnb)       // (lnb: the FC2 product of block l - 1 added into the stream and left its statistics)
                    IISAN_TRY(launch_layernorm768_mixed(dt, MX_ADD_STAT, nullptr, b.X16, b.Xc, pend_f, nullptr, nullptr, nullptr, w->eps, nullptr, mc, T, s, b.RS));
            } else if (!mixed)
                IISAN_TRY(launch_add2_layernorm768(dt, b.X, pend_o, pend_f, L.ln1_w, L.ln1_b, w->eps, pend_o ? b.X : nullptr, b.H, nullptr, tok, s));
            else if (l == 0)     // position table + 16-bit patch embedding -> fp16 stream of the patch rows (the CLS rows are in Xc already) + LN image
                IISAN_TRY(launch_layernorm768_mixed(dt, MX_SRC32 | MX_POSROW | MX_D1 | MX_RESV | MX_LN, w->pos_emb, b.X16, b.Xc, b.D16b, nullptr, L.ln1_w, L.ln1_b, w->eps, b.H, mc, T, s));
            else
                IISAN_TRY(launch_layernorm768_mixed(dt, MX_D1 | MX_D2 | MX_RESV | MX_LN, nullptr, b.X16, b.Xc, pend_o, pend_f, L.ln1_w, L.ln1_b, w->eps, b.H, mc, T, s));
            pend_o = pend_f = nullptr;
            k = tap_index(tap_layers, n_taps, l);
            if (k >= 0 && l > 0) IISAN_TRY(tap(k));
            if (lnb && (l + 1 < live || full_blocks)) {
                IISAN_TRY(gemm_ln(dt, EPI_QKVH16, b.X16, D, Wf_qkv(l), bf_qkv(l), b.RS, b.QKV, 3 * D, tok, s, T, w->heads));
                IISAN_TRY(launch_attention16(dt, b.QKV, nullptr, b.H, mc, T, w->heads, s));
                IISAN_TRY(gemm_stream(b.H, D, L.o_w, L.o_b, mc));
                IISAN_TRY(gemm_ln(dt, EPI_GELU16, b.X16, D, Wf_fc1(l), bf_fc1(l), b.RS, b.F1, F, tok, s));
                IISAN_TRY(gemm_stream(b.F1, F, L.fc2_w, L.fc2_b, mc));
            } else if (lna && (l + 1 < live || full_blocks)) {
                IISAN_TRY(gemm_ln(dt, EPI_QKVH16, b.X16, D, Wf_qkv(l), bf_qkv(l), b.RS, b.QKV, 3 * D, tok, s, T, w->heads));
                IISAN_TRY(launch_attention16(dt, b.QKV, nullptr, b.H, mc, T, w->heads, s));
                IISAN_TRY(gemm(dt, EPI_OUT16, b.H, D, L.o_w, L.o_b, b.D16, D, nullptr, tok, s));
                // x += dO (written: the FC1 product reads the stream); LN2 in the FC1 product's epilogue
                IISAN_TRY(launch_layernorm768_mixed(dt, MX_ADD_STAT, nullptr, b.X16, b.Xc, b.D16, nullptr, nullptr, nullptr, w->eps, nullptr, mc, T, s, b.RS));
                IISAN_TRY(gemm_ln(dt, EPI_GELU16, b.X16, D, Wf_fc1(l), bf_fc1(l), b.RS, b.F1, F, tok, s));
                IISAN_TRY(gemm(dt, EPI_OUT16, b.F1, F, L.fc2_w, L.fc2_b, b.D16b, D, nullptr, tok, s));
                pend_f = b.D16b;
            } else if (l + 1 < live || full_blocks) {
                IISAN_TRY(gemm(dt, EPI_QKVH16, b.H, D, L.qkv_w, L.qkv_b, b.QKV, 3 * D, nullptr, tok, s, nullptr, 0, T, w->heads));
                IISAN_TRY(launch_attention16(dt, b.QKV, nullptr, b.H, mc, T, w->heads, s));
                IISAN_TRY(gemm(dt, EPI_OUT16, b.H, D, L.o_w, L.o_b, b.D16, D, nullptr, tok, s));
                // h = LN2(x + dO)   (x itself is updated by the next LN1)
                if (!mixed)
                    IISAN_TRY(launch_add_layernorm768(dt, b.X, b.D16, L.ln2_w, L.ln2_b, w->eps, nullptr, b.H, nullptr, tok, s));
                else
                    IISAN_TRY(launch_layernorm768_mixed(dt, MX_D1 | MX_LN, nullptr, b.X16, b.Xc, b.D16, nullptr, L.ln2_w, L.ln2_b, w->eps, b.H, mc, T, s));
                IISAN_TRY(gemm(dt, EPI_GELU16, b.H, D, L.fc1_w, L.fc1_b, b.F1, F, nullptr, tok, s));
                IISAN_TRY(gemm(dt, EPI_OUT16, b.F1, F, L.fc2_w, L.fc2_b, b.D16b, D, nullptr, tok, s));
                pend_o = b.D16;
                pend_f = b.D16b;
            } else {
                // CLS rows only: compact [mc, D] 16-bit views in their own scratch
                void* Hc = b.Cls;
                void* Qc = (char*)b.Cls + (size_t)b.Mcp * D * 2;
                if (lna) {
                    // K / V of every token from the stream (LN1 in the epilogue); the CLS rows' LN1 image from their fp32 stream
                    IISAN_TRY(launch_layernorm768(dt, b.Xc, L.ln1_w, L.ln1_b, w->eps, Hc, nullptr, mc, s));
                    IISAN_TRY(gemm_ln(dt, EPI_QKVH16, b.X16, D, Wf_qkv(l) + (size_t)D * D * 2, bf_qkv(l) + D, b.RS, b.QKV, 2 * D, tok, s, T, w->heads, 1));
                    IISAN_TRY(gemm(dt, EPI_OUT16, Hc, D, L.qkv_w, L.qkv_b, Qc, D, nullptr, mc, s));
                } else {
                    IISAN_TRY(launch_gather_rows16(b.H, Hc, mc, T, D, s));            // CLS rows of LN1(x)
                    IISAN_TRY(kv_all_q_cls(dt, L, b.H, b.QKV, Hc, Qc, tok, mc, T, w->heads, D, s));
                }
                IISAN_TRY(launch_attention_cls16(dt, b.QKV, nullptr, b.H, mc, T, w->heads, s, Qc));
                float* Xc = (float*)b.QKV;      // free once the CLS attention has run (stream order)
                IISAN_TRY(gemm(dt, EPI_OUT16, b.H, D, L.o_w, L.o_b, b.D16, D, nullptr, mc, s));
                if (mixed) IISAN_TRY(launch_gather_cls(b.Xc, Xc, mc, 1, D, 1, 0, s)); else IISAN_TRY(launch_gather_cls(b.X, Xc, mc, T, D, 1, 0, s));
                IISAN_TRY(launch_add_layernorm768(dt, Xc, b.D16, L.ln2_w, L.ln2_b, w->eps, Xc, b.H, nullptr, mc, s));
                IISAN_TRY(gemm(dt, EPI_GELU16, b.H, D, L.fc1_w, L.fc1_b, b.F1, F, nullptr, mc, s));
                IISAN_TRY(gemm(dt, EPI_OUT16, b.F1, F, L.fc2_w, L.fc2_b, b.D16, D, nullptr, mc, s));
                IISAN_TRY(launch_add_layernorm768(dt, Xc, b.D16, nullptr, nullptr, w->eps, Xc, nullptr, nullptr, mc, s));
                k = tap_index(tap_layers, n_taps, live);     // hidden state `live` (before any final LayerNorm)
                IISAN_TRY(launch_gather_cls(Xc, tp, mc, 1, D, n_taps, k, s));
            }
        }
        if (full_blocks) {
            k = tap_index(tap_layers, n_taps, w->layers);
            if (k >= 0) {
                if (lnb) {}     // the stream is hidden state `layers` already
                else if (lna)   // x + dO is in the stream: the CLS rows take dF
                    IISAN_TRY(launch_layernorm768_mixed(dt, MX_D1 | MX_RESV | MX_CLSONLY, nullptr, b.X16, b.Xc, pend_f, nullptr, nullptr, nullptr, w->eps, nullptr, mc, T, s));
                else if (!mixed)
                    IISAN_TRY(launch_add2_layernorm768(dt, b.X, pend_o, pend_f, nullptr, nullptr, w->eps, b.X, nullptr, nullptr, tok, s));
                else            // only the CLS rows of the last hidden state are consumed
                    IISAN_TRY(launch_layernorm768_mixed(dt, MX_D1 | MX_D2 | MX_RESV | MX_CLSONLY, nullptr, b.X16, b.Xc, pend_o, pend_f, nullptr, nullptr, w->eps, nullptr, mc, T, s));
                IISAN_TRY(tap(k));
            }
        }
    }
    return IISAN_OK;
}

extern "C" size_t iisan_bert_forward_taps_ws_bytes(const iisan_bert_weights* w, int64_t M, int32_t words, int64_t chunk_items) {
    const int64_t Mc = (chunk_items > 0 && chunk_items < M) ? chunk_items : M;
    WsCarver c(nullptr, 0);
    EncBufs b;
    return carve(c, b, Mc * words, Mc, w->hidden, w->mlp, Mc * words);
}

extern "C" int iisan_bert_forward_taps(const iisan_bert_weights* w, const int64_t* text, int64_t M, int32_t words,
                                       const int32_t* tap_layers, int32_t n_taps, float* taps, int64_t chunk_items,
                                       void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    IISAN_CHECK_SHAPE(M > 0, "bert_forward_taps: M must be positive");
    IISAN_TRY(check_common(w->hidden, w->layers, w->heads, w->mlp, n_taps, tap_layers));
    IISAN_CHECK_SHAPE(words >= 1 && words <= w->max_pos && words <= 224, "bert: %d words unsupported", words);
    const int D = w->hidden, F = w->mlp, T = words;
    const int64_t Mc = (chunk_items > 0 && chunk_items < M) ? chunk_items : M;
    WsCarver c(ws, ws_bytes);
    EncBufs b;
    carve(c, b, Mc * T, Mc, D, F, Mc * T);
    if (c.overflow || !ws) {
        iisan_set_error("bert_forward_taps: workspace too small (%zu < %zu)", ws_bytes, c.off);
        return IISAN_EWORKSPACE;
    }
    const int dt = w->dtype16;
    for (int64_t m0 = 0; m0 < M; m0 += Mc) {
        const int64_t mc = (M - m0 < Mc) ? M - m0 : Mc;
        const int64_t tok = mc * T;
        float* tp = taps + m0 * n_taps * D;
        const bool mixed = !g_resid32;
        // post-LN tower: the stream IS the LayerNorm output.  resid32: X fp32 [tok, D]; mixed: Xc fp32 (CLS rows) + X16 fp16.  With fp16
        // operands the 16-bit image the products read and the fp16 stream are the same values: one buffer, written once (g_ln_fold; the add +
        // LayerNorm kernels move 6 instead of 8 bytes per element)
        const bool alias = mixed && dt == IISAN_F16 && g_ln_fold;
        void* IMG = alias ? b.X16 : b.H;
        IISAN_TRY(launch_bert_embed_ln(dt, text + m0 * 2 * words, w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_w,
                                       w->emb_ln_b, w->eps, mixed ? nullptr : b.X, IMG, b.KB, mc, T, w->vocab, s, b.X16, b.Xc));
        auto tap = [&](int k) { return mixed ? launch_gather_cls(b.Xc, tp, mc, 1, D, n_taps, k, s) : launch_gather_cls(b.X, tp, mc, T, D, n_taps, k, s); };
        // x = LN(x + delta): stream and 16-bit image out
        auto add_ln = [&](const float* g, const float* be) {
            return mixed ? launch_layernorm768_mixed(dt, MX_D1 | MX_LN | MX_RESY | (alias ? MX_ALIAS : 0), nullptr, b.X16, b.Xc, b.D16, nullptr, g, be, w->eps, IMG, mc, T, s)
                         : launch_add_layernorm768(dt, b.X, b.D16, g, be, w->eps, nullptr, b.H, b.X, tok, s);
        };
        int k = tap_index(tap_layers, n_taps, 0);
        if (k >= 0) IISAN_TRY(tap(k));
        const bool full_blocks = g_full_blocks || w->full_blocks;
        const int live = full_blocks ? w->layers : max_tap(tap_layers, n_taps);     // see the ViT executor
        for (int l = 0; l < live; ++l) {
            const iisan_layer_weights& L = w->layer[l];
            // a = LN(x + O(attn(x)))
            if (l + 1 < live || full_blocks) {
                IISAN_TRY(gemm(dt, EPI_QKVH16, IMG, D, L.qkv_w, L.qkv_b, b.QKV, 3 * D, nullptr, tok, s, nullptr, 0, T, w->heads));
                IISAN_TRY(launch_attention16(dt, b.QKV, b.KB, b.H, mc, T, w->heads, s));
                IISAN_TRY(gemm(dt, EPI_OUT16, b.H, D, L.o_w, L.o_b, b.D16, D, nullptr, tok, s));
                IISAN_TRY(add_ln(L.ln1_w, L.ln1_b));
                // x = LN(a + FC2(gelu(FC1 a)))
                IISAN_TRY(gemm(dt, EPI_GELU16, IMG, D, L.fc1_w, L.fc1_b, b.F1, F, nullptr, tok, s));
                IISAN_TRY(gemm(dt, EPI_OUT16, b.F1, F, L.fc2_w, L.fc2_b, b.D16, D, nullptr, tok, s));
                IISAN_TRY(add_ln(L.ln2_w, L.ln2_b));
                k = tap_index(tap_layers, n_taps, l + 1);
                if (k >= 0) IISAN_TRY(tap(k));
            } else {
                // post-LN tower: the block input H is the 16-bit image of X, so the CLS rows of H are gathered directly
                void* Hc = b.Cls;
                void* Qc = (char*)b.Cls + (size_t)b.Mcp * D * 2;
                IISAN_TRY(launch_gather_rows16(IMG, Hc, mc, T, D, s));
                IISAN_TRY(kv_all_q_cls(dt, L, IMG, b.QKV, Hc, Qc, tok, mc, T, w->heads, D, s));
                float* Xc = (float*)b.QKV;
                IISAN_TRY(launch_attention_cls16(dt, b.QKV, b.KB, b.H, mc, T, w->heads, s, Qc));
                IISAN_TRY(gemm(dt, EPI_OUT16, b.H, D, L.o_w, L.o_b, b.D16, D, nullptr, mc, s));
                if (mixed) IISAN_TRY(launch_gather_cls(b.Xc, Xc, mc, 1, D, 1, 0, s)); else IISAN_TRY(launch_gather_cls(b.X, Xc, mc, T, D, 1, 0, s));
                IISAN_TRY(launch_add_layernorm768(dt, Xc, b.D16, L.ln1_w, L.ln1_b, w->eps, nullptr, b.H, Xc, mc, s));
                IISAN_TRY(gemm(dt, EPI_GELU16, b.H, D, L.fc1_w, L.fc1_b, b.F1, F, nullptr, mc, s));
                IISAN_TRY(gemm(dt, EPI_OUT16, b.F1, F, L.fc2_w, L.fc2_b, b.D16, D, nullptr, mc, s));
                IISAN_TRY(launch_add_layernorm768(dt, Xc, b.D16, L.ln2_w, L.ln2_b, w->eps, nullptr, nullptr, Xc, mc, s));
                k = tap_index(tap_layers, n_taps, live);
                IISAN_TRY(launch_gather_cls(Xc, tp, mc, 1, D, n_taps, k, s));
            }
        }
    }
    return IISAN_OK;
}
