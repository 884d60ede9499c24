// Eval scoring: rank of each user's target item among all items (Hit@10 / nDCG@10 inputs).
// Replaces the per-user Python loop of eval_model + the full argsort of metrics_topK
// (Code_Uncached/data_utils/metrics.py:59-67,198-207): scores = prec · item_emb^T, history -> -inf, column 0
// dropped, rank = 1 + #{items ahead of the target}.  Ties are resolved towards the lower item id (the reference's
// argsort leaves tie order unspecified).  Integer-exact counting, no sort, the [U, N] scores never exist.
//
// Round 3: the score product runs on the f32 matrix cores (`v_mfma_f32_16x16x4_f32`, exact fp32 FMAs), 32 users per
// workgroup.  The first version gave every user its own workgroup whose 256 threads each streamed whole 256-byte item rows:
// the 5.2 MB table was re-read from L2 once per USER (62.8 GB per Scientific eval pass) and every score was a 64-step
// dependent VALU chain.  Now a workgroup holds the user vectors of 32 users as B fragments in registers (loop invariant),
// its four waves walk disjoint 16-item tiles (A fragments: 64 contiguous bytes per lane and tile, next tile prefetched),
// and one item-row load feeds two MFMAs: the table is read once per 32 users (1.9 GB) and item splits across blockIdx.y
// fill the chip, partial counts meeting in `ranks` through integer atomics (exact, order-free).
//
// Exactness: a rank is a count of comparisons `score(c) > score(target)`; every score of a user — the target's, its history
// items', every other item's — comes out of the SAME 16-MFMA chain with the same contraction order (lane group g contracts
// k = 16g .. 16g+15; the four groups meet in the MFMA's own fixed reduction), so equal inputs give equal bits wherever the
// item sits in a tile.  The target's and the history items' scores are taken from the diagonal of extra tiles whose row i
// is "the h-th item of user i".  History: a history item contributes with score -inf instead of its raw score; if the
// target itself is in the history its score is -inf as well (metrics.py:204-205: reproduced, not fixed).
#include "common.h"

namespace {

constexpr int UB = 32;          // users per workgroup (two 16-user B fragment sets)
constexpr int MAXH = 256;       // history entries per user (33 KB of LDS for their raw scores); longer exclusion lists: IISAN_EBADSHAPE

__device__ __forceinline__ bool ahead(float s, int c, float st, int t) { return s > st || (s == st && c < t); }

// one 16-item tile against both user sets: acc[u][r] = score(item 4*(lane/16) + r of the tile, user 16u + lane%16)
__device__ __forceinline__ void tile_scores(const f4 (&a)[4], const f4 (&b)[2][4], f4 (&acc)[2]) {
    acc[0] = (f4){0.f, 0.f, 0.f, 0.f};
    acc[1] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v][e], b[0][v][e], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v][e], b[1][v][e], acc[1], 0, 0, 0);
        }
}

__device__ __forceinline__ void load_row16(const float* __restrict__ row, int g, f4 (&a)[4]) {
#pragma unroll
    for (int v = 0; v < 4; ++v) a[v] = *(const f4*)(row + 16 * g + 4 * v);
}

__global__ __launch_bounds__(256) void score_rank_mfma_kernel(const float* __restrict__ prec, const float* __restrict__ item_emb,
                                                              int n_items, const int32_t* __restrict__ history, int hist_stride,
                                                              const int32_t* __restrict__ target, int32_t* __restrict__ ranks,
                                                              int U, int tiles_per_split) {
    __shared__ float s_sc[MAXH + 1][UB];      // [h][user]: raw score of history entry h (h < hist_stride) / of the target (h = hist_stride)
    __shared__ float s_st[UB];                // the target's effective score (-inf when it is in the history)
    __shared__ int s_t[UB];                   // target id (0 = invalid: rank -1)
    __shared__ int s_cnt[UB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int u0 = blockIdx.x * UB;

    // the users' vectors as B fragments: lane (j, g) holds prec[u0 + 16u + j][16g .. 16g+15]
    f4 b[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int uu = u0 + 16 * u + j;
        if (uu < U) load_row16(prec + (int64_t)uu * 64, g, b[u]);
        else
#pragma unroll
            for (int v = 0; v < 4; ++v) b[u][v] = (f4){0.f, 0.f, 0.f, 0.f};
    }
    if (threadIdx.x < UB) {
        const int uu = u0 + (int)threadIdx.x;
        int t = uu < U ? target[uu] : 0;
        if (t <= 0 || t >= n_items) t = 0;                // not an item: never dereferenced, reported as rank -1
        s_t[threadIdx.x] = t;
        s_cnt[threadIdx.x] = 0;
    }
    __syncthreads();

    // ---- phase 1: raw scores of every user's target and history items (diagonals of gathered tiles), tiles spread over the waves.
    // Tile (h, half): row i = entry h of user u0 + 16*half + i; only the user set `half` is of interest.
    const int n_ent = hist_stride + 1;
    for (int e = wave; e < 2 * n_ent; e += 4) {
        const int h = e >> 1, half = e & 1;
        const int uu = u0 + 16 * half + j;               // row j of the tile belongs to this user
        int c = 0;
        if (uu < U) c = h < hist_stride ? history[(int64_t)uu * hist_stride + h] : s_t[16 * half + j];
        if (c < 0 || c >= n_items) c = 0;
        f4 a[4], acc[2];
        load_row16(item_emb + (int64_t)c * 64, g, a);
        tile_scores(a, b, acc);
        // D[i][jj] sits in lane (jj, i / 4), element i % 4: the diagonal i == jj
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r == j) s_sc[h][16 * half + j] = half ? acc[1][r] : acc[0][r];
    }
    __syncthreads();
    if (threadIdx.x < UB) {
        const int x = threadIdx.x, uu = u0 + x, t = s_t[x];
        bool in_hist = false;
        if (uu < U && t > 0)
            for (int h = 0; h < hist_stride; ++h) in_hist |= history[(int64_t)uu * hist_stride + h] == t;
        s_st[x] = in_hist ? -INFINITY : s_sc[hist_stride][x];
    }
    __syncthreads();

    // ---- phase 2: count the items ahead of the target over this workgroup's item range
    float st[2]; int tt[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { st[u] = s_st[16 * u + j]; tt[u] = s_t[16 * u + j]; }
    const int n_tiles = (n_items + 15) >> 4;
    const int tile_lo = blockIdx.y * tiles_per_split;
    int tile_hi = tile_lo + tiles_per_split;
    if (tile_hi > n_tiles) tile_hi = n_tiles;
    int cnt[2] = {0, 0};
    f4 a[4], an[4];
    int tile = tile_lo + wave;
    auto row_of = [&](int tl) { int c = tl * 16 + j; return item_emb + (int64_t)(c < n_items ? c : 0) * 64; };
    if (tile < tile_hi) load_row16(row_of(tile), g, a);
    for (; tile < tile_hi; tile += 4) {
        const bool more = tile + 4 < tile_hi;
        if (more) load_row16(row_of(tile + 4), g, an);          // next tile's rows in flight during this tile's MFMAs
        f4 acc[2];
        tile_scores(a, b, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = tile * 16 + 4 * g + r;
            const bool item = c >= 1 && c < n_items;
#pragma unroll
            for (int u = 0; u < 2; ++u) cnt[u] += (item && c != tt[u] && ahead(acc[u][r], c, st[u], tt[u])) ? 1 : 0;
        }
        if (more)
#pragma unroll
            for (int v = 0; v < 4; ++v) a[v] = an[v];
    }
    // lanes (j, g = 0..3) hold partial counts of the same user
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        cnt[u] += __shfl_xor(cnt[u], 16, 64);
        cnt[u] += __shfl_xor(cnt[u], 32, 64);
        if (g == 0 && cnt[u]) atomicAdd(&s_cnt[16 * u + j], cnt[u]);
    }
    __syncthreads();

    // ---- phase 3: history corrections (first item split only) and the result
    if (threadIdx.x < UB) {
        const int x = threadIdx.x, uu = u0 + x;
        if (uu >= U) return;
        const int t = s_t[x];
        if (t == 0) {
            if (blockIdx.y == 0) ranks[uu] = -1;          // the launch zeroes `ranks` first; the other splits add nothing
            return;
        }
        int c_all = s_cnt[x];
        if (blockIdx.y == 0) {
            const float stx = s_st[x];
            const int32_t* hist = history + (int64_t)uu * hist_stride;
            for (int h = 0; h < hist_stride; ++h) {
                const int c = hist[h];
                if (c <= 0 || c >= n_items || c == t) continue;
                bool dup = false;
                for (int k = 0; k < h; ++k) dup |= hist[k] == c;
                if (dup) continue;
                if (ahead(s_sc[h][x], c, stx, t)) c_all -= 1;         // counted above with its raw score ...
                if (ahead(-INFINITY, c, stx, t)) c_all += 1;          // ... but it scores -inf
            }
            c_all += 1;                                               // rank = 1 + #ahead
        }
        if (c_all) atomicAdd(&ranks[uu], c_all);
    }
}

// invalid targets: a split other than 0 must not add to the -1 another block wrote; handled by giving those users no count
// (t == 0 returns before the atomic) — so `ranks` only needs zeroing before the launch.
__global__ void zero_ranks_kernel(int32_t* __restrict__ r, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r[i] = 0;
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Recommendation list: the first k item ids of `order = argsort(score, descending)` over the history-masked score row without column 0
// (metrics_topK, Code_Uncached/data_utils/metrics.py:59-60, on the row eval_model builds at :198-206) — what the rank above is a position in.
// Same score tiles as score_rank_mfma_kernel (one 16-MFMA chain per score: the bits of a score are those the rank kernel compares, so a
// target ranked r <= k sits at position r - 1 of this list), same tie rule (lower item id first = a stable descending argsort), and the
// [U, N] scores never exist either:
//   * every wave keeps, per user, a short candidate list in LDS and the list's k-th best entry as a register threshold; a score is
//     appended only when it is ahead of that threshold (one compare per score in the common case);
//   * a list that could overflow with the next tile is COMPACTED by the wave itself: entries that are in the user's exclusion list are
//     dropped, every other entry counts the entries ahead of it (the order is total: scores, then ids) and moves to the slot of that
//     count if it is < k — the list comes out sorted, the threshold is its last entry;
//   * the four waves' lists meet the same way per workgroup, the item splits' lists in topk_merge_kernel: every step is a selection by
//     exact comparisons in a fixed total order, so the result does not depend on launch geometry or timing.
// Fewer than k candidates (item_num - |history| < k): the remaining slots hold id 0 / score -inf (the reference's argsort lists the -inf
// items there, in ascending id order for a stable sort: iisan_amd/evaluate.py fills them in).
constexpr int TK_MAX = 16;          // k <= 16
constexpr int TK_CAP = 48;          // candidate slots per (wave, user): a tile appends at most 16, compaction when more than TK_CAP - 16 are used
constexpr int TK_HS = 64;           // exclusion-list entries per user kept in LDS (longer lists are read from global memory)
constexpr int TK_MAX_SPLITS = 128;  // item splits per user block (topk_merge_kernel stages splits * k entries in LDS)

// A candidate is ONE 64-bit key: the score's bits mapped monotonically to unsigned (high word), ~id (low word) — `a is ahead of b`
// (higher score, or equal score and lower id: `ahead` above) is the single unsigned compare key(a) > key(b); 0 marks an empty / excluded
// slot (every real key is larger).  -0.0 is canonicalised to +0.0 first: the two compare equal as floats and must tie on the id.
typedef unsigned long long tkey;
__device__ __forceinline__ tkey tk_key(float s, int c) {
    uint32_t b = __float_as_uint(s + 0.0f);
    b ^= (b >> 31) ? 0xffffffffu : 0x80000000u;
    return ((tkey)b << 32) | (uint32_t)(~c);
}
__device__ __forceinline__ int tk_id(tkey k) { return (int)~(uint32_t)k; }
__device__ __forceinline__ float tk_score(tkey k) {
    uint32_t b = (uint32_t)(k >> 32);
    b ^= (b >> 31) ? 0x80000000u : 0xffffffffu;
    return __uint_as_float(b);
}
// lanes of ONE wave exchanging data through LDS: the hardware runs a wave's LDS operations in order, the compiler must too
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave: drop the excluded entries of cand[0 .. n) of user x = lane & 31, select and sort its k best (n = this lane's user's count; lanes
// l and l + 32 serve the same user and split its entries).  Returns the user's new count.  Cost: every entry counts the entries ahead of it —
// one 8-byte LDS read (the 32 users side by side: conflict-free), one 64-bit compare and one add-with-carry per pair, the reads requested
// eight at a time.  (The first version compared (score, id) pairs with one dependent LDS read per step: 1.6 ms per Scientific eval pass,
// matrix pipe 13 % busy, waves parked 52 % of the time — profiles/r6_eval_kernel_stats.md tells the before / after.)
__device__ __forceinline__ int topk_compact(tkey (*cand)[UB], tkey (*tmp)[UB], int n, int lane, int k, int u0, int U,
                                            const int (*s_hist)[TK_HS + 1], const int32_t* __restrict__ history, int hist_stride) {
    const int x = lane & 31, half = lane >> 5;
    const int uu = u0 + x;
    int nmax = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(nmax, o, 64); nmax = t > nmax ? t : nmax; }
    nmax = (nmax + 7) & ~7;                                      // wave-uniform, <= TK_CAP
    int n_ex = 0;
    for (int e = half; e < n; e += 2) {                         // exclusion: `score[history] = -inf` (metrics.py:204-205)
        const int c = tk_id(cand[e][x]);
        bool ex = false;
        if (hist_stride <= TK_HS) {
#pragma unroll 4
            for (int h = 0; h < hist_stride; ++h) ex |= s_hist[x][h] == c;
        } else if (uu < U) {
#pragma unroll 4
            for (int h = 0; h < hist_stride; ++h) ex |= history[(int64_t)uu * hist_stride + h] == c;
        }
        if (ex) { cand[e][x] = 0; ++n_ex; }
    }
    for (int e = n + half; e < nmax; e += 2) cand[e][x] = 0;    // the tail up to the wave's longest list: stale slots must not count
    n_ex += __shfl_xor(n_ex, 32, 64);
    wave_sync();
    for (int e = half; e < n; e += 2) {
        const tkey ke = cand[e][x];
        if (ke == 0) continue;
        int r = 0;
        for (int f0 = 0; f0 < nmax; f0 += 8) {
            tkey kf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) kf[i] = cand[f0 + i][x];
#pragma unroll
            for (int i = 0; i < 8; ++i) r += kf[i] > ke ? 1 : 0;
        }
        if (r < k) tmp[r][x] = ke;
    }
    wave_sync();
    const int nv = n - n_ex;
    const int m = nv < k ? nv : k;
    for (int e = half; e < m; e += 2) cand[e][x] = tmp[e][x];
    wave_sync();
    return m;
}

template <bool DIRECT>
__global__ __launch_bounds__(256) void score_topk_kernel(const float* __restrict__ prec, const float* __restrict__ item_emb, int n_items,
                                                         const int32_t* __restrict__ history, int hist_stride, int U, int tiles_per_split,
                                                         int k, tkey* __restrict__ part, int32_t* __restrict__ ids, float* __restrict__ scores) {
    __shared__ tkey s_cand[4][TK_CAP][UB];          // 48 KiB
    __shared__ tkey s_tmp[4][TK_MAX][UB];           // 16 KiB
    __shared__ int s_hist[UB][TK_HS + 1];
    __shared__ int s_cnt[4][UB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int u0 = blockIdx.x * UB;

    f4 b[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int uu = u0 + 16 * u + j;
        if (uu < U) load_row16(prec + (int64_t)uu * 64, g, b[u]);
        else
#pragma unroll
            for (int v = 0; v < 4; ++v) b[u][v] = (f4){0.f, 0.f, 0.f, 0.f};
    }
    if (hist_stride <= TK_HS)
        for (int i = threadIdx.x; i < UB * hist_stride; i += 256) {
            const int x = i / hist_stride, h = i - x * hist_stride;
            s_hist[x][h] = u0 + x < U ? history[(int64_t)(u0 + x) * hist_stride + h] : 0;
        }
    __syncthreads();

    tkey (*cand)[UB] = s_cand[wave];
    tkey (*tmp)[UB] = s_tmp[wave];
    float st[2] = {-INFINITY, -INFINITY};
    int tt[2] = {0x7fffffff, 0x7fffffff};
    int cnt[2] = {0, 0};                                         // list lengths of users j and 16 + j: the same value in the four lanes (g) of a j
    const bool live[2] = {u0 + j < U, u0 + 16 + j < U};
    const unsigned long long below = (1ull << (16 * g)) - 1;     // the lanes (j, g' < g)

    auto compact = [&]() {
        const int m = topk_compact(cand, tmp, (lane & 16) ? cnt[1] : cnt[0], lane, k, u0, U, s_hist, history, hist_stride);
        cnt[0] = __shfl(m, j, 64);                               // lane x = j serves user j, lane 16 + j user 16 + j
        cnt[1] = __shfl(m, 16 + j, 64);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (cnt[u] >= k) { const tkey t = cand[k - 1][16 * u + j]; st[u] = tk_score(t); tt[u] = tk_id(t); }
    };

    const int n_tiles = (n_items + 15) >> 4;
    const int tile_lo = blockIdx.y * tiles_per_split;
    int tile_hi = tile_lo + tiles_per_split;
    if (tile_hi > n_tiles) tile_hi = n_tiles;
    f4 a[4], an[4];
    int tile = tile_lo + wave;
    auto row_of = [&](int tl) { int c = tl * 16 + j; return item_emb + (int64_t)(c < n_items ? c : 0) * 64; };
    if (tile < tile_hi) load_row16(row_of(tile), g, a);
    for (; tile < tile_hi; tile += 4) {
        const bool more = tile + 4 < tile_hi;
        if (more) load_row16(row_of(tile + 4), g, an);
        f4 acc[2];
        tile_scores(a, b, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = tile * 16 + 4 * g + r;
            const bool item = c >= 1 && c < n_items;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // append without atomics: the four lanes (g) of a user learn from one ballot how many of them pass and where each one's
                // entry goes; exclusion lists are applied at compaction (all 64 lanes busy there, one or two here)
                const bool pass = item && live[u] && ahead(acc[u][r], c, st[u], tt[u]);
                const unsigned long long m4 = (__ballot(pass) >> j) & 0x0001000100010001ull;
                if (pass) cand[cnt[u] + __popcll(m4 & below)][16 * u + j] = tk_key(acc[u][r], c);
                cnt[u] += __popcll(m4);
            }
        }
        if (__any(cnt[0] > TK_CAP - 16 || cnt[1] > TK_CAP - 16)) { wave_sync(); compact(); }
        if (more)
#pragma unroll
            for (int v = 0; v < 4; ++v) a[v] = an[v];
    }
    wave_sync();
    compact();
    if (lane < 32) s_cnt[wave][lane] = (lane & 16) ? cnt[1] : cnt[0];
    __syncthreads();

    // ---- the four waves' sorted lists -> the workgroup's k best per user: thread (x, p) takes entries p, p + 8, ... of the concatenation
    {
        const int x = threadIdx.x & 31, p = threadIdx.x >> 5;
        const int uu = u0 + x;
        const int n0 = s_cnt[0][x], n1 = s_cnt[1][x], n2 = s_cnt[2][x], n3 = s_cnt[3][x];
        const int nall = n0 + n1 + n2 + n3;
        auto entry = [&](int e) {
            if (e < n0) return s_cand[0][e][x];
            e -= n0;
            if (e < n1) return s_cand[1][e][x];
            e -= n1;
            if (e < n2) return s_cand[2][e][x];
            return s_cand[3][e - n2][x];
        };
        tkey* dst_part = DIRECT ? nullptr : part + ((int64_t)blockIdx.y * U + uu) * k;
        auto put = [&](int r, tkey v) {
            if (DIRECT) { ids[(int64_t)uu * k + r] = v ? tk_id(v) : 0; if (scores) scores[(int64_t)uu * k + r] = v ? tk_score(v) : -INFINITY; }
            else dst_part[r] = v;
        };
        if (uu < U) {
            for (int e = p; e < nall; e += 8) {
                const tkey ke = entry(e);
                int r = 0;
                for (int f = 0; f < nall; ++f) r += entry(f) > ke ? 1 : 0;
                if (r < k) put(r, ke);
            }
            for (int r = (nall < k ? nall : k) + p; r < k; r += 8) put(r, 0);
        }
    }
}

// the item splits' lists of one user -> ids / scores (one wave per user; at most TK_MAX_SPLITS * TK_MAX entries)
__global__ __launch_bounds__(64) void topk_merge_kernel(const tkey* __restrict__ part, int splits, int U, int k, int32_t* __restrict__ ids,
                                                        float* __restrict__ scores) {
    __shared__ tkey s_e[TK_MAX_SPLITS * TK_MAX];
    const int u = blockIdx.x, lane = threadIdx.x;
    const int n = splits * k;
    int nv = 0;
    for (int e = lane; e < n; e += 64) {
        const int sp = e / k, i = e - sp * k;
        const tkey v = part[((int64_t)sp * U + u) * k + i];
        s_e[e] = v;
        nv += v != 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nv += __shfl_xor(nv, o, 64);
    __syncthreads();
    for (int e = lane; e < n; e += 64) {
        const tkey ke = s_e[e];
        if (ke == 0) continue;
        int r = 0;
        for (int f = 0; f < n; ++f) r += s_e[f] > ke ? 1 : 0;
        if (r < k) { ids[(int64_t)u * k + r] = tk_id(ke); if (scores) scores[(int64_t)u * k + r] = tk_score(ke); }
    }
    for (int r = (nv < k ? nv : k) + lane; r < k; r += 64) { ids[(int64_t)u * k + r] = 0; if (scores) scores[(int64_t)u * k + r] = -INFINITY; }
}

// item splits of the top-k launch: two workgroups per CU fit (73 KiB of LDS each).  Longer item streams per wave mean fewer appended
// candidates (a stream's threshold is its own k-th best), more splits mean better-filled rounds: among 1 .. 8 splits (more only when
// there are fewer user blocks than workgroup slots) the count with the best-filled last round wins, ties to the fewer splits.
static int topk_splits(int64_t U, int64_t n_items_plus1) {
    const int64_t ublocks = ceil_div(U, UB), n_tiles = ceil_div(n_items_plus1, 16), slots = (int64_t)2 * iisan_cu_count();
    int64_t cap = n_tiles / 64;                                  // at least 16 tiles (256 items) per wave
    if (cap > TK_MAX_SPLITS) cap = TK_MAX_SPLITS;
    if (cap < 1) cap = 1;
    int64_t best = 1;
    if (ublocks >= slots) {
        double best_fill = 0.0;
        for (int64_t sp = 1; sp <= 8 && sp <= cap; ++sp) {
            const int64_t wgs = ublocks * sp;
            const double fill = (double)wgs / (double)(ceil_div(wgs, slots) * slots);
            if (fill > best_fill + 0.02) { best_fill = fill; best = sp; }
        }
    } else {
        best = ceil_div(slots, ublocks);
        if (best > cap) best = cap;
    }
    const int64_t per = ceil_div(n_tiles, best);
    return (int)ceil_div(n_tiles, per);
}

}  // namespace

extern "C" int iisan_score_rank(const float* prec, const float* item_emb, int64_t U, int64_t n_items_plus1, int32_t E,
                                const int32_t* history, int32_t hist_stride, const int32_t* target, int32_t* ranks,
                                void* stream) {
    hipStream_t s = (hipStream_t)stream;
    IISAN_CHECK_SHAPE(E == 64, "score_rank: embedding_dim must be 64 (got %d)", E);
    IISAN_CHECK_SHAPE(U > 0 && n_items_plus1 > 1 && hist_stride >= 0, "score_rank: empty problem");
    IISAN_CHECK_SHAPE(hist_stride <= MAXH, "score_rank: at most %d history entries per user (got %d)", MAXH, hist_stride);
    IISAN_CHECK_SHAPE(U < (1ll << 31) - UB && n_items_plus1 < (1ll << 31) - 16, "score_rank: problem too large for 32-bit indices");
    IISAN_CHECK_SHAPE((((uintptr_t)prec | (uintptr_t)item_emb) & 15) == 0, "score_rank: prec and item_emb must be 16-byte aligned");
    const int ublocks = (int)ceil_div(U, UB);
    const int n_tiles = (int)ceil_div(n_items_plus1, 16);
    // item splits: about four workgroups per CU in total, at least 16 tiles (4 per wave) each
    int splits = (int)ceil_div((int64_t)4 * iisan_cu_count(), ublocks);
    if (splits > n_tiles / 16) splits = n_tiles / 16;
    if (splits < 1) splits = 1;
    const int per = (int)ceil_div(n_tiles, splits);
    splits = (int)ceil_div(n_tiles, per);
    hipLaunchKernelGGL(zero_ranks_kernel, dim3((unsigned)ceil_div(U, 256)), dim3(256), 0, s, ranks, U);
    IISAN_LAUNCH_OK();
    hipLaunchKernelGGL(score_rank_mfma_kernel, dim3((unsigned)ublocks, (unsigned)splits), dim3(256), 0, s, prec, item_emb,
                       (int)n_items_plus1, history, hist_stride, target, ranks, (int)U, per);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

extern "C" size_t iisan_score_topk_ws_bytes(int64_t U, int64_t n_items_plus1, int32_t k) {
    if (U <= 0 || n_items_plus1 <= 1 || k <= 0) return 0;
    const int splits = topk_splits(U, n_items_plus1);
    return splits > 1 ? align_up((size_t)splits * (size_t)U * (size_t)k * sizeof(tkey), 256) : 0;
}

extern "C" int iisan_score_topk(const float* prec, const float* item_emb, int64_t U, int64_t n_items_plus1, int32_t E,
                                const int32_t* history, int32_t hist_stride, int32_t k, int32_t* topk_ids, float* topk_scores,
                                void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    IISAN_CHECK_SHAPE(E == 64, "score_topk: embedding_dim must be 64 (got %d)", E);
    IISAN_CHECK_SHAPE(U > 0 && n_items_plus1 > 1 && hist_stride >= 0, "score_topk: empty problem");
    IISAN_CHECK_SHAPE(k >= 1 && k <= TK_MAX, "score_topk: k must be in 1..%d (got %d)", TK_MAX, k);
    IISAN_CHECK_SHAPE(hist_stride == 0 || history, "score_topk: hist_stride %d without a history array", hist_stride);
    IISAN_CHECK_SHAPE(U < (1ll << 31) - UB && n_items_plus1 < (1ll << 31) - 16 && U * k < (1ll << 31), "score_topk: problem too large for 32-bit indices");
    IISAN_CHECK_SHAPE((((uintptr_t)prec | (uintptr_t)item_emb) & 15) == 0, "score_topk: prec and item_emb must be 16-byte aligned");
    const int ublocks = (int)ceil_div(U, UB);
    const int n_tiles = (int)ceil_div(n_items_plus1, 16);
    const int splits = topk_splits(U, n_items_plus1);
    const int per = (int)ceil_div(n_tiles, splits);
    const size_t need = iisan_score_topk_ws_bytes(U, n_items_plus1, k);
    if (ws_bytes < need || (need && !ws)) {
        iisan_set_error("score_topk: workspace too small (%zu < %zu)", ws_bytes, need);
        return IISAN_EWORKSPACE;
    }
    if (splits == 1) {
        hipLaunchKernelGGL(score_topk_kernel<true>, dim3((unsigned)ublocks, 1), dim3(256), 0, s, prec, item_emb, (int)n_items_plus1, history,
                           hist_stride, (int)U, per, k, (tkey*)nullptr, topk_ids, topk_scores);
        IISAN_LAUNCH_OK();
        return IISAN_OK;
    }
    hipLaunchKernelGGL(score_topk_kernel<false>, dim3((unsigned)ublocks, (unsigned)splits), dim3(256), 0, s, prec, item_emb, (int)n_items_plus1,
                       history, hist_stride, (int)U, per, k, (tkey*)ws, topk_ids, topk_scores);
    IISAN_LAUNCH_OK();
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)U), dim3(64), 0, s, (const tkey*)ws, splits, (int)U, k, topk_ids, topk_scores);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}
