// Split-operand GEMM for the big TRAINABLE products (fc_* 768x768 layers of the side network, Versa's dim-align
// projections and their backward products): fp32 operands are split on the device into two fp16 planes,
//     x * s = hi + lo,   hi = fp16(x*s),  lo = fp16(x*s - hi),   s = power of two chosen from the tensor's amax,
// and   A·B^T  ~=  (Ah·Bh^T + Ah·Bl^T + Al·Bh^T) / (sA sB)   runs as ONE 16-bit MFMA GEMM on operand images that hold each plane once,
//     A2 = [Ah | Al]   B2 = [Bh | Bl]     (rows of 2*Kp 16-bit elements)
// whose K-step stages the four tiles and runs the three tile products from them (gemm16_x3.hip; rounds 2 - 5: K' = 3K images [Ah|Ah|Al] x
// [Bh|Bl|Bh] through the encoder GEMM kernel — a third more operand bytes for the same MFMAs).  hi and lo together carry 22 mantissa bits of every
// element within 2^-16 of the tensor's amax (below that the error floor is 2^-38 of amax), the dropped lo·lo term is
// 2^-22 relative: the product is within a few fp32 ulps of an fp32 FMA chain, at 1/3 of the 16-bit MFMA rate instead of
// the f32-input matrix cores' 1/16 (MI355X_MICROARCH.md: 2.5 PF vs 157 TF).  The reference computes these Linear
// layers in fp32 (Code_Cached/model/model.py:333-347, Code_Cached_Asym/model/model.py:400-416).
//
// Everything stays on the stream: amax -> scale -> split are kernels, the scales reach the GEMM epilogue through device
// memory (never the host), so the library still never synchronises.
#include "common.h"

namespace {

struct SplitArgs {
    const float* x;        // source, row-major [rows, cols], leading dimension ld (floats)
    int64_t rows, cols, ld;
    _Float16* out;         // [out_rows, 2*kp]: hi plane | lo plane
    int64_t out_rows, kp;  // padded operand rows / padded K (multiple of 64)
    int32_t trans;         // 1: the operand is x^T (operand row = source column, K runs over source rows)
    uint32_t* amax_bits;   // in: bit pattern of max|x| (amax kernel)
    float* inv_scale;      // out: 1/s
    int32_t* lo_flag;      // out: set to 1 when any lo element is non-zero (pre-zeroed; plain stores of the same value)
    int32_t skip_lo;       // 1: the caller vouches that every element is exact in fp16 at scale 1 (taps cached in fp16, amax slot preset): the lo plane is
                           // neither computed nor written and the flag stays 0 — the product never reads it (23 MB per [1408, 8192] tap image)
};

__device__ __forceinline__ void amax_body(const float* __restrict__ x, int64_t rows, int64_t cols, int64_t ld, uint32_t* amax_bits,
                                          int64_t block, int64_t nblocks) {
    // cols % 4 == 0 and 16-byte aligned rows are checked on the host.  ONE atomic per workgroup and at most 512
    // workgroups: same-address atomics serialise (the first version issued one per wave from 2048 workgroups — 8192
    // atomics on one word — and took 103 us for a 35 MB tensor, 10x its HBM time).
    __shared__ float red[4];
    const int64_t c4 = cols / 4, total = rows * c4;
    float m = 0.f;
    const int64_t stride = nblocks * blockDim.x;
    int64_t i = block * blockDim.x + threadIdx.x;
    if (ld == cols) {
        // contiguous rows (weights, whole activations): a flat array — no 64-bit division per load, eight loads in flight per thread
        // (round 6: Versa's seven [1024, 8192] dim-align weights in one launch ran at 3.0 TB/s through the generic loop below)
        const f4* x4 = (const f4*)x;
        for (; i + 7 * stride < total; i += 8 * stride) {
            f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x4[i + u * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
        }
        for (; i < total; i += stride) {
            const f4 v = x4[i];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
        i = total;
    }
    // four loads in flight per thread (one per iteration left the pass latency-bound: 12.6 us for 35 MB)
    for (; i + 3 * stride < total; i += 4 * stride) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = i + u * stride, r = q / c4, c = (q - r * c4) * 4;
            v[u] = *(const f4*)(x + r * ld + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
    }
    for (; i < total; i += stride) {
        const int64_t r = i / c4, c = (i - r * c4) * 4;
        const f4 v = *(const f4*)(x + r * ld + c);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f) atomicMax(amax_bits, __float_as_uint(m));   // non-negative floats order like their bit patterns
    }
}

__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                   uint32_t* amax_bits) {
    amax_body(x, rows, cols, ld, amax_bits, blockIdx.x, gridDim.x);
}

// the amax of several tensors in ONE launch (blockIdx.y = tensor): the operands of a group of split-operand products — the three fc
// layers of a direction, Versa's seven dim-align weights — paid a launch of 5 - 16 us each, latency-bound (a 33.5 MB weight at 2.1 TB/s)
__global__ __launch_bounds__(256) void amax_batch_kernel(AmaxBatch b) {
    const int z = blockIdx.y;
    amax_body(b.x[z], b.rows[z], b.cols[z], b.ld[z], b.out[z], blockIdx.x, gridDim.x);
}

// s = 2^(13 - floor(log2 amax)): the largest element lands in [2^13, 2^14) (fp16 max 65504); amax == 0 or denormal -> s = 1
__device__ __forceinline__ float scale_of(uint32_t amax_bits) {
    const int e = (int)(amax_bits >> 23) & 0xff;
    if (e == 0 || e == 0xff) return 1.0f;
    return __uint_as_float((uint32_t)(267 - e) << 23);
}

__device__ __forceinline__ void split1(float xs, _Float16& hi, _Float16& lo) {
    hi = (_Float16)xs;
    lo = (_Float16)(xs - (float)hi);
}

// operand row = source row: one thread per 8 consecutive K elements
__device__ __forceinline__ void split_rows_body(const SplitArgs& a, int64_t block, int64_t nblocks) {
    const float s = scale_of(*a.amax_bits);
    if (block == 0 && threadIdx.x == 0) *a.inv_scale = 1.0f / s;
    const int64_t g8 = a.kp / 8, total = a.out_rows * g8;
    int any_lo = 0;
    for (int64_t i = block * blockDim.x + threadIdx.x; i < total; i += nblocks * blockDim.x) {
        const int64_t r = i / g8, k = (i - r * g8) * 8;
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)0.f; lo[e] = (_Float16)0.f; }
        if (r < a.rows && k < a.cols) {
            const float* p = a.x + r * a.ld + k;
            if (k + 8 <= a.cols) {
                const f4 v0 = *(const f4*)p, v1 = *(const f4*)(p + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 h, l;
                    split1(v0[e] * s, h, l); hi[e] = h; lo[e] = l;
                    split1(v1[e] * s, h, l); hi[4 + e] = h; lo[4 + e] = l;
                }
            } else {
                for (int e = 0; e < 8 && k + e < a.cols; ++e) {
                    _Float16 h, l;
                    split1(p[e] * s, h, l); hi[e] = h; lo[e] = l;
                }
            }
        }
        _Float16* o = a.out + r * 2 * a.kp + k;
        *(h8*)o = hi;
        if (a.skip_lo) continue;
        *(h8*)(o + a.kp) = lo;
        const u4 lb = __builtin_bit_cast(u4, lo);
        any_lo |= ((lb[0] | lb[1] | lb[2] | lb[3]) & 0x7fff7fffu) != 0;
    }
    if (__syncthreads_or(any_lo) && threadIdx.x == 0) *a.lo_flag = 1;
}
__global__ __launch_bounds__(256) void split_rows_kernel(SplitArgs a) { split_rows_body(a, blockIdx.x, gridDim.x); }

// operand row = source COLUMN (x^T): 64 x 64 tiles transposed through LDS; K runs over the source rows.  One tile (bx, by) of the
// (kp / 64) x ceil(out_rows / 64) tile grid; returns whether any lo element of it is non-zero (thread-local)
__device__ __forceinline__ int split_cols_tile(const SplitArgs& a, float (&T)[64][65], float s, int64_t bx, int64_t by) {
    const int64_t r0 = bx * 64;      // source rows  = K index
    const int64_t c0 = by * 64;      // source cols  = operand rows
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = (tid >> 4) + 16 * i, cc = (tid & 15) * 4;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        const int64_t r = r0 + rr, c = c0 + cc;
        if (r < a.rows && c < a.cols) {
            const float* p = a.x + r * a.ld + c;
            if (c + 4 <= a.cols) v = *(const f4*)p;
            else for (int e = 0; e < 4 && c + e < a.cols; ++e) v[e] = p[e];
        }
        T[rr][cc] = v[0]; T[rr][cc + 1] = v[1]; T[rr][cc + 2] = v[2]; T[rr][cc + 3] = v[3];
    }
    __syncthreads();
    int any_lo = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int w = tid + 256 * i;          // 512 work items: operand row cc (64) x K group kg (8)
        const int cc = w >> 3, kg = (w & 7) * 8;
        const int64_t orow = c0 + cc;
        if (orow >= a.out_rows) continue;
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 h, l;
            split1(T[kg + e][cc] * s, h, l);
            hi[e] = h; lo[e] = l;
        }
        _Float16* o = a.out + orow * 2 * a.kp + r0 + kg;
        *(h8*)o = hi;
        if (a.skip_lo) continue;
        *(h8*)(o + a.kp) = lo;
        const u4 lb = __builtin_bit_cast(u4, lo);
        any_lo |= ((lb[0] | lb[1] | lb[2] | lb[3]) & 0x7fff7fffu) != 0;
    }
    return any_lo;
}
__global__ __launch_bounds__(256) void split_cols_kernel(SplitArgs a) {
    __shared__ float T[64][65];
    const float s = scale_of(*a.amax_bits);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.inv_scale = 1.0f / s;
    const int any_lo = split_cols_tile(a, T, s, blockIdx.x, blockIdx.y);
    if (__syncthreads_or(any_lo) && threadIdx.x == 0) *a.lo_flag = 1;
}

// the operand images of a GROUP of split-operand products in one launch (blockIdx.y = operand; round 6: Versa's seven dim-align products
// paid two launches each, forward and weight gradient — 28 launches of 5 - 16 us per step)
struct SplitBatch { SplitArgs a[16]; };
__global__ __launch_bounds__(256) void split_batch_kernel(SplitBatch b) {
    __shared__ float T[64][65];
    const SplitArgs& a = b.a[blockIdx.y];
    if (!a.trans) { split_rows_body(a, blockIdx.x, gridDim.x); return; }
    const float s = scale_of(*a.amax_bits);
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.inv_scale = 1.0f / s;
    const int64_t tx = a.kp / 64, ty = (a.out_rows + 63) / 64;
    int any_lo = 0;
    for (int64_t t = blockIdx.x; t < tx * ty; t += gridDim.x) {
        const int64_t by = t / tx, bx = t - by * tx;
        any_lo |= split_cols_tile(a, T, s, bx, by);
        __syncthreads();            // T is reused by the next tile
    }
    if (__syncthreads_or(any_lo) && threadIdx.x == 0) *a.lo_flag = 1;
}

// the split-K sums of a group's products in one launch (blockIdx.y = product)
struct ReduceX3 { const float* P; int32_t ks; int64_t stride; const float* bias; const float* resid; float* C; int64_t M; int32_t N, ldc; };
struct ReduceX3Batch { ReduceX3 r[8]; };
__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(ReduceX3Batch b) {
    const ReduceX3& r = b.r[blockIdx.y];
    const int n4 = r.N / 4;
    const int64_t total = r.M * n4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / n4;
        const int n = (int)(i - m * n4) * 4;
        f4 v = *(const f4*)(r.P + m * r.N + n);
        for (int y = 1; y < r.ks; ++y) v += *(const f4*)(r.P + y * r.stride + m * r.N + n);
        if (r.bias) v += *(const f4*)(r.bias + n);
        if (r.resid) v += *(const f4*)(r.resid + m * r.ldc + n);
        *(f4*)(r.C + m * r.ldc + n) = v;
    }
}

// C[m][n] = sum_y P[y][m][n] (+ bias[n]) (+ resid[m][n]); 16-byte lanes, fixed summation order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ P, int ks, int64_t stride, const float* __restrict__ bias,
                                                            const float* resid, float* C, int64_t M, int N, int ldc) {
    const int n4 = N / 4;
    const int64_t total = M * n4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / n4;
        const int n = (int)(i - m * n4) * 4;
        f4 v = *(const f4*)(P + m * N + n);
        for (int y = 1; y < ks; ++y) v += *(const f4*)(P + y * stride + m * N + n);
        if (bias) v += *(const f4*)(bias + n);
        if (resid) v += *(const f4*)(resid + m * ldc + n);
        *(f4*)(C + m * ldc + n) = v;
    }
}

}  // namespace

static int g_amax_blocks = 0;            // > 0: workgroups per tensor of the batched amax pass (sweeps); 0 = by the rule in launch_amax_batch
IISAN_DEV_KNOB(amax_blocks, g_amax_blocks);
int launch_amax_batch(const AmaxBatch& b, int n, hipStream_t s) {
    if (n <= 0) return IISAN_OK;
    int64_t blocks = 1;
    for (int i = 0; i < n; ++i) {
        IISAN_CHECK_SHAPE(b.cols[i] % 4 == 0 && b.ld[i] % 4 == 0 && ((uintptr_t)b.x[i] & 15) == 0, "amax: source rows must be 16-byte aligned");
        const int64_t w = ceil_div(b.rows[i] * (b.cols[i] / 4), 256 * 4);
        if (w > blocks) blocks = w;
    }
    // workgroups per tensor (one atomic each, and the slots of a launch share a cache line: same-line atomics serialise).  Kernel trace of the Cached
    // step (six / three [11264, 768] and [768, 768] tensors per launch, average of the two launches): 512 -> 34.5 us, 256 -> 24.2, 128 -> 19.5, 64 -> 20.3;
    // Versa's seven [1024, 8192] weights: 45 us whatever the count
    const int64_t cap = g_amax_blocks > 0 ? g_amax_blocks : (n >= 2 ? 128 : 512);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(amax_batch_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, s, b);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

namespace {

int split_operand(const float* x, bool trans, int64_t op_rows, int64_t K, int64_t ld, _Float16* out,
                  int64_t out_rows, int64_t kp, uint32_t* amax_bits, float* inv_scale, int32_t* lo_flag, hipStream_t s,
                  bool amax_ready = false, bool skip_lo = false) {
    SplitArgs a{};
    a.lo_flag = lo_flag;
    a.skip_lo = (skip_lo && amax_ready) ? 1 : 0;
    a.x = x; a.ld = ld; a.out = out; a.out_rows = out_rows; a.kp = kp; a.trans = trans ? 1 : 0;
    a.amax_bits = amax_bits; a.inv_scale = inv_scale;
    a.rows = trans ? K : op_rows;
    a.cols = trans ? op_rows : K;
    IISAN_CHECK_SHAPE(a.cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0, "split: source rows must be 16-byte aligned");
    int64_t blocks = ceil_div(a.rows * (a.cols / 4), 256 * 4);
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    if (!amax_ready) {
        hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, a.rows, a.cols, ld, amax_bits);
        IISAN_LAUNCH_OK();
    }
    if (!trans) {
        int64_t b2 = ceil_div(out_rows * (kp / 8), 256);
        if (b2 > 8192) b2 = 8192;
        hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)b2), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(split_cols_kernel, dim3((unsigned)(kp / 64), (unsigned)ceil_div(out_rows, 64)), dim3(256), 0, s, a);
    }
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

}  // namespace

int launch_gemm16_f32(const Gemm16Args& a, int ksplit, hipStream_t s);      // gemm16.hip

// K-splits of a product.  Partial products go to the workspace and a small reducer adds them (fp32 atomics run at ~20 G/s chip-wide: a
// split-K epilogue on them took 430 us for 8 M outputs, tools/g16f32_time.py).  Round 6: chosen by a cost estimate instead of "double until 384
// workgroups" — the chip holds 2 x CUs workgroups of gemm16_x3p_kernel at a time, a split count that spills a few workgroups into another round
// pays a whole round for them (Versa's dim-align product: 88 tiles x 8 splits = 704 workgroups on 512 slots, two rounds of 32 K-steps; five
// splits = 440 workgroups, ONE round of 52), and every split costs a pass over its partial (written and read back: ~8 bytes per output).
//   cost(ks) = ceil(tiles ks / slots) x (steps / ks + 4) K-step times (~1 us)  +  [ks > 1] ks x (8 M N bytes at ~5 TB/s)
int x3p_tile_n(int64_t M, int64_t N);          // gemm16_x3.hip: 128 or 192 columns per tile
static int g_x3_ks = 0;                  // > 0: this many K splits for every product (sweeps); 0 = by the cost estimate
IISAN_DEV_KNOB(x3_force_ks, g_x3_ks);
static int x3_ksplit(int64_t mp, int64_t np, int64_t kp) {
    if (g_x3_ks > 0) return g_x3_ks > 16 ? 16 : g_x3_ks;
    const int bn = x3p_tile_n(mp, np);
    const int64_t tiles = (mp / 128) * ceil_div(np, bn), steps = kp / 32 * bn / 128, slots = (int64_t)2 * iisan_cu_count();
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= 16; ++ks) {
        if (ks > 1 && steps / ks < 12) break;
        const double rounds = (double)ceil_div(tiles * ks, slots);
        const double cost = rounds * ((double)steps / ks + 4.0) + (ks > 1 ? ks * (8.0 * (double)mp * (double)np / 5e6) : 0.0);
        if (cost < best_cost * 0.97) { best_cost = cost; best = ks; }        // (ties and near-ties: the fewer splits)
    }
    return best;
}

size_t gemm_x3_ws_bytes(int64_t M, int64_t N, int64_t K) {
    const int64_t kp = ceil_div(K, 64) * 64, mp = ceil_div(M, 128) * 128, np = ceil_div(N, 128) * 128;
    const int ks = x3_ksplit(mp, np, kp);
    return align_up((size_t)mp * 2 * kp * 2, 256) + align_up((size_t)np * 2 * kp * 2, 256) + 256 +
           (ks > 1 ? align_up((size_t)ks * M * N * 4, 256) : 0);
}

// Worth the four extra small launches (amax + split per operand)?  Only the big products.
static bool x3_shape_ok(const Gemm32Prob& p, int flags) {
    if (flags & ~(G32_TA | G32_TB | G32_ACCUM | G32_HINT_B_EXACT16)) return false;            // no activation / dropout epilogues
    if (p.act_src || p.N % 8 || p.ldc % 4) return false;
    if (((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15) return false;
    if (p.lda % 4 || p.ldb % 4) return false;
    if (p.resid && (p.ldr != p.ldc)) return false;
    return true;
}
// Below this many FLOPs the extra passes (memset, amax + split per operand, ~7 dependent launches) cost more than the
// matrix rate gains.  Measured (tools/x3_time.py, MI355X; gemm32 reaches 83-98 TF on these shapes):
//   [11264,768]x[768,768] fwd / dX 106 us vs 151-160 (1.4-1.5x), its dW 135 vs 160 (1.2x), Versa dim-align [1408,8192]->1024
//   172 vs 265 (1.5x) and its dW 133 vs 241 (1.8x); [4373,768]x[768,768] (the distinct ids of a bs = 1024 batch) 61 vs 86
//   (1.4x), its dW 74 vs 75; [2816,768]x[768,768] 54 vs 50-55 (even); [1408,768]x[768,768] 45 vs 30 us (0.7x: stays on
//   the f32 cores).
// ONE default for the product, the tests and the bench (dev switch x3 = 1 restores exactly this value): products of at least
// 4 GFLOP take the split-operand route ([4373, 768]x[768, 768] 61 vs 86 us; [2816, 768]x[768, 768] even)
constexpr double X3_DEFAULT_MIN_FLOPS = 4e9;
static double g_x3_min_flops = X3_DEFAULT_MIN_FLOPS;
void gemm_x3_set_min_flops(double f) { g_x3_min_flops = f < 0 ? X3_DEFAULT_MIN_FLOPS : f; }
double gemm_x3_get_min_flops() { return g_x3_min_flops; }
bool gemm_x3_applicable(const Gemm32Prob& p, int flags) {
    return x3_shape_ok(p, flags) && 2.0 * (double)p.M * (double)p.N * (double)p.K >= g_x3_min_flops;
}

// C[M,N] (=|+=) op(A)[M,K] · op(B)[K,N] + bias (+ resid), same operand conventions as launch_gemm32
// (A stored [M,K] or, G32_TA, [K,M]; B stored [N,K] or, G32_TB, [K,N]); G32_ACCUM: C += (read-modify-write, no atomics).
// `ws` >= gemm_x3_ws_bytes(M, N, K).
static int launch_gemm_x3_any(const Gemm32Prob& p, int flags, void* ws, size_t ws_bytes, hipStream_t s);
int launch_gemm_x3(const Gemm32Prob& p, int flags, void* ws, size_t ws_bytes, hipStream_t s) {
    return launch_gemm_x3_any(p, flags, ws, ws_bytes, s);
}
static int64_t g_cnt_x3 = 0;                  // split-operand products launched (route counter, common.h)
IISAN_DEV_COUNTER(gemm_x3, g_cnt_x3);
static int launch_gemm_x3_any(const Gemm32Prob& p, int flags_in, void* ws, size_t ws_bytes, hipStream_t s) {
    ++g_cnt_x3;
    int flags = flags_in;
    IISAN_CHECK_SHAPE(x3_shape_ok(p, flags), "gemm_x3: unsupported problem (flags 0x%x, N %d)", flags, p.N);
    IISAN_CHECK_SHAPE(ws && ws_bytes >= gemm_x3_ws_bytes(p.M, p.N, p.K), "gemm_x3: workspace too small");
    const int64_t kp = ceil_div(p.K, 64) * 64, mp = ceil_div(p.M, 128) * 128, np = ceil_div(p.N, 128) * 128;
    char* w = (char*)ws;
    _Float16* A16 = (_Float16*)w;
    w += align_up((size_t)mp * 2 * kp * 2, 256);
    _Float16* B16 = (_Float16*)w;
    w += align_up((size_t)np * 2 * kp * 2, 256);
    char* z48 = p.x3_zeroed ? (char*)p.x3_zeroed : w;
    uint32_t* amax = (uint32_t*)z48;          // [0] A, [1] B
    float* inv = (float*)(z48 + 16);          // [0] A, [1] B
    int32_t* lo_flag = (int32_t*)(z48 + 32);  // [0] A, [1] B: any non-zero lo element
    if (!p.x3_zeroed) IISAN_HIP_OK(hipMemsetAsync(amax, 0, 48, s));
    // (G32_HINT_B_EXACT16 told the old route which operand's lo plane to put last in K' so that the GEMM could drop it; the plane-sharing
    //  kernel reads both flags and skips whichever lo plane is all zeros — an operand that is exact in fp16, like taps cached in fp16)
    flags &= ~G32_HINT_B_EXACT16;
    IISAN_TRY(split_operand(p.A, (flags & G32_TA) != 0, p.M, p.K, p.lda, A16, mp, kp, p.amax_a ? p.amax_a : amax, inv, lo_flag, s,
                            p.amax_a && p.amax_a_ready, p.exact16_a != 0));
    // B operand rows = N: stored [N,K] by default, [K,N] under G32_TB (then the operand is the source transposed)
    IISAN_TRY(split_operand(p.B, (flags & G32_TB) != 0, p.N, p.K, p.ldb, B16, np, kp, p.amax_b ? p.amax_b : amax + 1, inv + 1, lo_flag + 1, s,
                            p.amax_b && p.amax_b_ready, p.exact16_b != 0));
    X3pArgs g{};
    g.A2 = A16; g.B2 = B16; g.M = p.M; g.N = p.N; g.kp = (int32_t)kp;
    g.bias = p.bias; g.resid = p.resid; g.out = p.C; g.ldo = p.ldc;
    g.inv_a = inv; g.inv_b = inv + 1; g.lo_a = lo_flag; g.lo_b = lo_flag + 1;
    const bool accum = (flags & G32_ACCUM) != 0;
    const int ks = x3_ksplit(mp, np, kp);
    if (ks == 1) {
        // "+=": the previous value of C is the residual (read-modify-write by the one lane that owns the element)
        if (accum) { IISAN_CHECK_SHAPE(!p.resid, "gemm_x3: accumulate and residual together"); g.resid = p.C; }
        return launch_gemm16_x3p(g, 1, s);
    }
    float* P = (float*)(w + 256);
    g.out = P; g.ldo = p.N; g.bias = nullptr; g.resid = nullptr; g.split_stride = p.M * (int64_t)p.N;
    IISAN_TRY(launch_gemm16_x3p(g, ks, s));
    int64_t blocks = ceil_div(p.M * (p.N / 4), 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, P, ks, g.split_stride, p.bias,
                       accum ? p.C : p.resid, p.C, p.M, p.N, p.ldc);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

// A GROUP of split-operand products that share flags (the three fc layers of a direction, Versa's dim-align products): ONE launch builds
// every operand image, the products follow, ONE launch adds the split-K partials of all of them.  Needs every product's zeroed block and
// both amax slots filled (sidenet.hip: gemm_group batches the amax passes) and the workspace to hold the images of the whole group at once
// (gemm_x3_group_ws_bytes); otherwise — or with the dev knob x3_group = 0 — the products go one by one through launch_gemm_x3.
static int g_x3_group = 8;              // products per group (<= 8); 0 / 1: one by one
IISAN_DEV_KNOB(x3_group, g_x3_group);
int gemm_x3_group_max() { return g_x3_group < 1 ? 1 : (g_x3_group > 8 ? 8 : g_x3_group); }
static int64_t g_cnt_x3_group = 0;
IISAN_DEV_COUNTER(gemm_x3_group, g_cnt_x3_group);
size_t gemm_x3_group_ws_bytes(const int64_t* M, const int64_t* N, const int64_t* K, int n) {
    size_t t = 0;
    for (int i = 0; i < n; ++i) t += align_up(gemm_x3_ws_bytes(M[i], N[i], K[i]), 256);
    return t;
}
int launch_gemm_x3_group(const Gemm32Prob* probs, int n, int flags_in, void* ws, size_t ws_bytes, hipStream_t s) {
    bool grouped = g_x3_group >= 2 && n >= 2 && n <= 8 && ws;
    size_t need = 0;
    for (int i = 0; i < n && grouped; ++i) {
        const Gemm32Prob& p = probs[i];
        grouped = p.x3_zeroed && p.amax_a && p.amax_b && p.amax_a_ready && p.amax_b_ready && x3_shape_ok(p, flags_in);
        need += align_up(gemm_x3_ws_bytes(p.M, p.N, p.K), 256);
    }
    if (!grouped || need > ws_bytes) {
        for (int i = 0; i < n; ++i) IISAN_TRY(launch_gemm_x3_any(probs[i], flags_in, ws, ws_bytes, s));
        return IISAN_OK;
    }
    ++g_cnt_x3_group;
    g_cnt_x3 += n;
    const int flags = flags_in & ~G32_HINT_B_EXACT16;
    const bool accum = (flags & G32_ACCUM) != 0;
    SplitBatch sb{};
    ReduceX3Batch rb{};
    X3pArgs g[8];
    int ks[8];
    int nred = 0;
    int64_t split_blocks = 1, red_blocks = 1;
    char* w = (char*)ws;
    for (int i = 0; i < n; ++i) {
        const Gemm32Prob& p = probs[i];
        const int64_t kp = ceil_div(p.K, 64) * 64, mp = ceil_div(p.M, 128) * 128, np = ceil_div(p.N, 128) * 128;
        char* w0 = w;
        _Float16* A16 = (_Float16*)w;
        w += align_up((size_t)mp * 2 * kp * 2, 256);
        _Float16* B16 = (_Float16*)w;
        w += align_up((size_t)np * 2 * kp * 2, 256);
        float* P = (float*)(w + 256);
        w = w0 + align_up(gemm_x3_ws_bytes(p.M, p.N, p.K), 256);
        char* z48 = (char*)p.x3_zeroed;
        float* inv = (float*)(z48 + 16);
        int32_t* lo_flag = (int32_t*)(z48 + 32);
        for (int o = 0; o < 2; ++o) {
            SplitArgs& a = sb.a[2 * i + o];
            const bool trans = (flags & (o ? G32_TB : G32_TA)) != 0;
            const int64_t op_rows = o ? p.N : p.M;
            a.x = o ? p.B : p.A; a.ld = o ? p.ldb : p.lda; a.out = o ? B16 : A16; a.out_rows = o ? np : mp; a.kp = kp; a.trans = trans ? 1 : 0;
            a.amax_bits = o ? p.amax_b : p.amax_a; a.inv_scale = inv + o; a.lo_flag = lo_flag + o;
            a.skip_lo = (o ? p.exact16_b : p.exact16_a) ? 1 : 0;
            a.rows = trans ? p.K : op_rows;
            a.cols = trans ? op_rows : p.K;
            IISAN_CHECK_SHAPE(a.cols % 4 == 0 && a.ld % 4 == 0 && ((uintptr_t)a.x & 15) == 0, "split: source rows must be 16-byte aligned");
            const int64_t blocks = trans ? (kp / 64) * ceil_div(a.out_rows, 64) : ceil_div(a.out_rows * (kp / 8), 256);
            if (blocks > split_blocks) split_blocks = blocks;
        }
        g[i] = X3pArgs{};
        g[i].A2 = A16; g[i].B2 = B16; g[i].M = p.M; g[i].N = p.N; g[i].kp = (int32_t)kp;
        g[i].bias = p.bias; g[i].resid = p.resid; g[i].out = p.C; g[i].ldo = p.ldc;
        g[i].inv_a = inv; g[i].inv_b = inv + 1; g[i].lo_a = lo_flag; g[i].lo_b = lo_flag + 1;
        ks[i] = x3_ksplit(mp, np, kp);
        if (ks[i] == 1) {
            if (accum) { IISAN_CHECK_SHAPE(!p.resid, "gemm_x3: accumulate and residual together"); g[i].resid = p.C; }
        } else {
            g[i].out = P; g[i].ldo = p.N; g[i].bias = nullptr; g[i].resid = nullptr; g[i].split_stride = p.M * (int64_t)p.N;
            ReduceX3& r = rb.r[nred++];
            r.P = P; r.ks = ks[i]; r.stride = g[i].split_stride; r.bias = p.bias; r.resid = accum ? p.C : p.resid; r.C = p.C; r.M = p.M; r.N = p.N; r.ldc = p.ldc;
            const int64_t blocks = ceil_div(p.M * (p.N / 4), 256);
            if (blocks > red_blocks) red_blocks = blocks;
        }
    }
    if (split_blocks > 4096) split_blocks = 4096;
    hipLaunchKernelGGL(split_batch_kernel, dim3((unsigned)split_blocks, (unsigned)(2 * n)), dim3(256), 0, s, sb);
    IISAN_LAUNCH_OK();
    for (int i = 0; i < n; ++i) IISAN_TRY(launch_gemm16_x3p(g[i], ks[i], s));
    if (nred) {
        if (red_blocks > 2048) red_blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3((unsigned)red_blocks, (unsigned)nred), dim3(256), 0, s, rb);
        IISAN_LAUNCH_OK();
    }
    return IISAN_OK;
}

extern "C" size_t iisan_gemm_x3_ws_bytes(int64_t M, int32_t N, int64_t K) { return gemm_x3_ws_bytes(M, N, K); }

extern "C" int iisan_gemm_x3(const float* A, const float* B, const float* bias, float* C, int64_t M, int32_t N, int64_t K,
                             int32_t ta, int32_t tb, int32_t accumulate, void* ws, size_t ws_bytes, void* stream) {
    Gemm32Prob p{};
    p.A = A; p.B = B; p.bias = bias; p.resid = nullptr; p.act_src = nullptr; p.C = C;
    p.M = M; p.N = N; p.K = K;
    p.lda = ta ? (int32_t)M : (int32_t)K;
    p.ldb = tb ? N : (int32_t)K;
    p.ldc = N; p.ldr = N;
    const int flags = (ta ? G32_TA : 0) | (tb ? G32_TB : 0) | (accumulate ? G32_ACCUM : 0);
    // the exported entry point takes any size (tests); the side network asks gemm_x3_applicable() first
    IISAN_CHECK_SHAPE(M > 0 && N > 0 && K > 0 && N % 8 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0, "gemm_x3: bad shape");
    Gemm32Prob q = p;
    return launch_gemm_x3_any(q, flags, ws, ws_bytes, (hipStream_t)stream);
}
