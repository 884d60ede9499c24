// Eval scoring: rank of each user's target item among all items (Hit@10 / nDCG@10 inputs).
// Replaces the per-user Python loop of eval_model + the full argsort of metrics_topK
// (Code_Uncached/data_utils/metrics.py:59-67,198-207): scores = prec · item_emb^T, history -> -inf, column 0
// dropped, rank = 1 + #{items ahead of the target}.  Ties are resolved towards the lower item id (the reference's
// argsort leaves tie order unspecified).  HBM/L2-bound: the item table ([n,64] fp32, 5 MB for
// Scientific) is streamed once per user block from L2; integer-exact counting, no sort.
#include "common.h"

namespace {

__device__ __forceinline__ float dot64(const float* __restrict__ a, const float* p) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const f4 t = *(const f4*)(a + 4 * v);
        s = fmaf(t[0], p[4 * v], s);
        s = fmaf(t[1], p[4 * v + 1], s);
        s = fmaf(t[2], p[4 * v + 2], s);
        s = fmaf(t[3], p[4 * v + 3], s);
    }
    return s;
}

__device__ __forceinline__ bool ahead(float s, int64_t c, float st, int64_t t) { return s > st || (s == st && c < t); }

__global__ __launch_bounds__(256) void score_rank_kernel(const float* __restrict__ prec, const float* __restrict__ item_emb,
                                                         int64_t n_items, const int32_t* __restrict__ history, int hist_stride,
                                                         const int32_t* __restrict__ target, int32_t* __restrict__ ranks) {
    __shared__ float sp[64];
    __shared__ int red[256];
    const int64_t u = blockIdx.x;
    if (threadIdx.x < 64) sp[threadIdx.x] = prec[u * 64 + threadIdx.x];
    __syncthreads();
    float p[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) p[e] = sp[e];
    const int64_t t = target[u];
    if (t <= 0 || t >= n_items) {            // not an item: never dereferenced, reported as rank -1
        if (threadIdx.x == 0) ranks[u] = -1;
        return;
    }
    const int32_t* hist = history + u * hist_stride;
    bool t_in_hist = false;
    for (int h = 0; h < hist_stride; ++h) t_in_hist |= (hist[h] == (int32_t)t && hist[h] != 0);
    const float st = t_in_hist ? -INFINITY : dot64(item_emb + t * 64, p);
    int cnt = 0;
    for (int64_t c = 1 + threadIdx.x; c < n_items; c += 256) cnt += ahead(dot64(item_emb + c * 64, p), c, st, t) ? 1 : 0;
    // history corrections: a history item contributes with score -inf instead of its raw score
    for (int h = threadIdx.x; h < hist_stride; h += 256) {
        const int64_t c = hist[h];
        if (c <= 0 || c >= n_items) continue;
        bool dup = false;
        for (int k = 0; k < h; ++k) dup |= hist[k] == hist[h];
        if (dup) continue;
        if (ahead(dot64(item_emb + c * 64, p), c, st, t)) cnt -= 1;
        if (ahead(-INFINITY, c, st, t)) cnt += 1;
    }
    red[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) ranks[u] = 1 + red[0];
}

}  // namespace

extern "C" int iisan_score_rank(const float* prec, const float* item_emb, int64_t U, int64_t n_items_plus1, int32_t E,
                                const int32_t* history, int32_t hist_stride, const int32_t* target, int32_t* ranks,
                                void* stream) {
    IISAN_CHECK_SHAPE(E == 64, "score_rank: embedding_dim must be 64 (got %d)", E);
    IISAN_CHECK_SHAPE(U > 0 && n_items_plus1 > 1 && hist_stride >= 0, "score_rank: empty problem");
    hipLaunchKernelGGL(score_rank_kernel, dim3((unsigned)U), dim3(256), 0, (hipStream_t)stream, prec, item_emb, n_items_plus1,
                       history, hist_stride, target, ranks);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}
