// Internal helpers shared by the gfx950 kernels of libiisan_hip.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/iisan_hip.h"

// ---- vector types (wave64; MFMA operand fragments) ----------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

#define WAVE 64

// ---- error plumbing ------------------------------------------------------------------------------------------
void iisan_set_error(const char* fmt, ...);

#define IISAN_CHECK_SHAPE(cond, ...)                                                                             \
    do {                                                                                                         \
        if (!(cond)) {                                                                                           \
            iisan_set_error(__VA_ARGS__);                                                                        \
            return IISAN_EBADSHAPE;                                                                              \
        }                                                                                                        \
    } while (0)

#define IISAN_HIP_OK(expr)                                                                                       \
    do {                                                                                                         \
        hipError_t _e = (expr);                                                                                  \
        if (_e != hipSuccess) {                                                                                  \
            iisan_set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__, hipGetErrorString(_e));         \
            return IISAN_EHIP;                                                                                   \
        }                                                                                                        \
    } while (0)

#define IISAN_LAUNCH_OK() IISAN_HIP_OK(hipGetLastError())

// ---- development switches (dev.cpp; include/iisan_hip.h DEV section) ------------------------------------------------------
// A kernel file's process-wide route / ablation variable becomes reachable by name through iisan_dev_set / iisan_dev_get:
//     static int g_ln_fold = 2;   IISAN_DEV_KNOB(ln_fold, g_ln_fold);
// the value at registration time is the library default (iisan_dev_reset, iisan_dev_state).
struct IisanDevKnob {
    const char* name;
    int64_t def;
    int64_t (*get)();
    void (*set)(int64_t);
};
int iisan_dev_register(const IisanDevKnob& k);
#define IISAN_DEV_KNOB(NAME, VAR)                                                                                \
    static const int iisan_dev_reg_##NAME = iisan_dev_register(IisanDevKnob{                                     \
        #NAME, (int64_t)(VAR), [] { return (int64_t)(VAR); }, [](int64_t v) { VAR = (decltype(VAR))v; }})
// ROUTE COUNTERS (round 6): "how often did this kernel family launch" — read through iisan_dev_get("count:<name>"), zeroed by iisan_dev_set(..., 0) /
// iisan_dev_reset; never part of iisan_dev_state (def = INT64_MIN marks a counter).  tests/test_gpu_trainable.py pins the DEFAULT dispatch of the
// bench shapes with them: a threshold that silently sends the headline to another kernel family fails a test.
#define IISAN_DEV_COUNTER(NAME, VAR)                                                                             \
    static const int iisan_dev_cnt_##NAME = iisan_dev_register(IisanDevKnob{                                     \
        "count:" #NAME, INT64_MIN, [] { return (int64_t)(VAR); }, [](int64_t v) { VAR = (decltype(VAR))v; }})
// ... with a side effect behind the store (SET is a statement using the new value `v`)
#define IISAN_DEV_KNOB_FN(NAME, VAR, SET)                                                                        \
    static const int iisan_dev_reg_##NAME = iisan_dev_register(IisanDevKnob{                                     \
        #NAME, (int64_t)(VAR), [] { return (int64_t)(VAR); }, [](int64_t v) { SET; }})

#define IISAN_TRY(expr)                                                                                          \
    do {                                                                                                         \
        int _r = (expr);                                                                                         \
        if (_r != IISAN_OK) return _r;                                                                           \
    } while (0)

// per-device host caches (api.cpp).  iisan_cu_count(): compute units of the CURRENT device, asked of the runtime once per
// device.  OncePerDevice: "this kernel's dynamic-LDS limit has been raised" is a fact about one device — a process that drives
// a second device must raise it there too (a process-wide `static bool` made the second device's launches fail).
constexpr int IISAN_MAX_DEVICES = 64;
int iisan_cu_count();
struct OncePerDevice {
    bool done[IISAN_MAX_DEVICES] = {};
    bool first() {                                       // true exactly once per device (single issuing thread per process)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= IISAN_MAX_DEVICES) return true;
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bump allocator over the caller's workspace (host side; the library never allocates device memory).
struct WsCarver {
    char* base;
    size_t cap, off;
    bool overflow;
    WsCarver(void* p, size_t n) : base((char*)p), cap(n), off(0), overflow(false) {}
    template <typename T>
    T* take(size_t count) {
        size_t bytes = align_up(count * sizeof(T), 256);
        if (base && off + bytes > cap) overflow = true;
        T* r = base ? (T*)(base + off) : nullptr;
        off += bytes;
        return r;
    }
};

// ---- 16-bit operand type traits -------------------------------------------------------------------------------
struct F16 {
    typedef _Float16 elem;
    typedef h8 v8;
    typedef h4 v4;
    static __device__ __forceinline__ f4 mfma(v8 a, v8 b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ elem from_f32(float x) { return (_Float16)x; }
    static __device__ __forceinline__ float to_f32(elem x) { return (float)x; }
};
struct BF16 {
    typedef __bf16 elem;
    typedef b8 v8;
    typedef b4 v4;
    static __device__ __forceinline__ f4 mfma(v8 a, v8 b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ elem from_f32(float x) { return (__bf16)x; }
    static __device__ __forceinline__ float to_f32(elem x) { return (float)x; }
};

// ---- device helpers ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// GELU(erf) for results that are rounded to a 16-bit operand anyway (FC1 epilogue of the frozen encoders), with NO
// transcendental:   gelu(x) = max(x, 0) - a * R(a)^8,   a = min(|x|, 5.5),   R = degree-5 polynomial,
// where R(a)^8 approximates 0.5 * erfc(a / sqrt 2) (a minimax fit of the gelu error itself over [0, 5.5], three
// squarings; beyond 5.5 the term is < 1.4e-8 |x|).  Max |gelu error| 1.7e-6 over [-12, 12] in fp32 arithmetic
// (tools/gelu_fit.py; an fp16 result near 0.01 already rounds by 3.8e-6).  History: libm erff (branchy); A&S 7.1.26
// (rcp + exp, ~84 VALU cycles per value); A&S 7.1.28 (1 + a1 z + .. + a6 z^6)^-16 packed (6 fma + 4 mul + rcp, ~45
// cycles; the FC1 epilogue stayed VALU-issue bound, 23 instructions per pair); this form is 5 fma + 3 mul + min + max +
// fma = 15 instructions per pair with the bias add and the conversion.
constexpr float GELU_X0 = 5.5f;
// min(|x|, X0) and max(x, 0) as ONE instruction each: fminf / fmaxf put a canonicalising `v_max x, x` in front (two of
// the 12 VALU slots per value)
__device__ __forceinline__ float gelu_absmin(float x) {
    float r;
    asm("v_min_f32 %0, |%1|, %2" : "=v"(r) : "v"(x), "v"(GELU_X0));
    return r;
}
__device__ __forceinline__ float gelu_relu(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
constexpr float GELU_C0 = 9.170038104e-01f, GELU_C1 = -9.149338305e-02f, GELU_C2 = -3.171720356e-02f, GELU_C3 = -1.111552469e-03f,
                GELU_C4 = 1.940784161e-03f, GELU_C5 = -1.910830324e-04f;
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float a = gelu_absmin(x);
    float p = fmaf(a, GELU_C5, GELU_C4);
    p = fmaf(p, a, GELU_C3);
    p = fmaf(p, a, GELU_C2);
    p = fmaf(p, a, GELU_C1);
    p = fmaf(p, a, GELU_C0);
    p = p * p; p = p * p; p = p * p;
    return fmaf(-a, p, gelu_relu(x));
}
// The same on 8 pairs at once on the packed-fp32 path (v_pk_fma_f32 / v_pk_mul_f32), written step by step across the
// pairs so that every Horner / squaring step is 8 INDEPENDENT instructions: with one wave per SIMD doing VALU work the
// dependent chain of a single evaluation (~8 cycles of latency per step) is otherwise exposed.
typedef float f2 __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ void gelu_erf_fast2xN(f2* x) {
    f2 a[NP], p[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) { a[k][0] = gelu_absmin(x[k][0]); a[k][1] = gelu_absmin(x[k][1]); }
#pragma unroll
    for (int k = 0; k < NP; ++k) p[k] = __builtin_elementwise_fma(a[k], (f2)GELU_C5, (f2)GELU_C4);
#pragma unroll
    for (int k = 0; k < NP; ++k) p[k] = __builtin_elementwise_fma(p[k], a[k], (f2)GELU_C3);
#pragma unroll
    for (int k = 0; k < NP; ++k) p[k] = __builtin_elementwise_fma(p[k], a[k], (f2)GELU_C2);
#pragma unroll
    for (int k = 0; k < NP; ++k) p[k] = __builtin_elementwise_fma(p[k], a[k], (f2)GELU_C1);
#pragma unroll
    for (int k = 0; k < NP; ++k) p[k] = __builtin_elementwise_fma(p[k], a[k], (f2)GELU_C0);
#pragma unroll
    for (int sq = 0; sq < 3; ++sq)
#pragma unroll
        for (int k = 0; k < NP; ++k) p[k] = p[k] * p[k];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        f2 mx;
        mx[0] = gelu_relu(x[k][0]); mx[1] = gelu_relu(x[k][1]);
        x[k] = __builtin_elementwise_fma(-a[k], p[k], mx);
    }
}
__device__ __forceinline__ void gelu_erf_fast2x8(f2 (&x)[8]) { gelu_erf_fast2xN<8>(x); }
// d/dx gelu_erf
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Counter-based dropout: the keep/scale factor of element `idx` at dropout site `site` is a pure function of
// (seed, site, idx), so the backward pass regenerates the forward mask instead of storing it.  24-bit uniform from a
// 64-bit mix (two rounds of xor-shift-multiply); tests/test_gpu_trainable.py re-implements it in numpy.
__device__ __forceinline__ float drop_scale(uint64_t seed, uint32_t site, uint64_t idx, uint32_t thr24, float inv_keep) {
    uint64_t x = seed + (uint64_t)site * 0x9E3779B97F4A7C15ull + idx * 0xD1B54A32D192ED03ull;
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return ((uint32_t)(x >> 8) & 0xFFFFFFu) >= thr24 ? inv_keep : 0.f;
}
struct DropCfg {           // p == 0 (thr24 == 0): identity
    uint64_t seed;
    uint32_t site, thr24;
    float inv_keep;
};
static inline DropCfg make_drop(uint64_t seed, uint32_t site, float p) {
    DropCfg d;
    d.seed = seed; d.site = site;
    d.thr24 = p > 0.f ? (uint32_t)(p * 16777216.0f) : 0u;
    d.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    return d;
}

// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform base + lane*16 (guide §5)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware bijective remap of a linear workgroup id: the dispatcher places id b on XCD b%8; give every XCD a
// contiguous range of logical tiles so neighbouring tiles (sharing operand panels) hit the same L2 (guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// per-launch HIP-event timing of the dominant kernel (timing.cpp); used by bench.py only
bool iisan_timing_on(hipStream_t s);   // class 1 (gemm16) enabled for launches on stream s
int iisan_timing_class();       // 0 = off, 1 = gemm16, 2 = the gemm32.hip family
void iisan_timing_pre(hipStream_t s, double flops, double bytes);
void iisan_timing_post(hipStream_t s);

// internal launchers shared between translation units -----------------------------------------------------------
struct Gemm16Args {
    const void* A;      // [Mpad, K] 16-bit
    const void* W;      // [N, K] 16-bit
    const float* bias;  // [N] or null
    void* out;          // mode-dependent
    const float* resid; // fp32 [*, N] (modes 2,3)
    const float* pos;   // patch mode: pos_emb [P+1, N]
    int64_t M;          // valid rows
    int32_t N, K;
    int32_t lda, ldw, ldo;
    int32_t patch_P;    // >0: row remap m -> (m/P)*(P+1) + 1 + m%P, += pos[1 + m%P]
    int32_t qkv_S, qkv_heads;   // EPI_QKVH16: tokens per item and heads of the head-major QKV layout
    int32_t qkv_which0;         // EPI_QKVH16: first of q|k|v (0..2) the N = (3 - which0)*64*heads columns hold (1 = K and V only)
    int32_t debug;      // ablation bits for micro-benchmarks: 1 = skip epilogue stores, 2 = skip steady-state DMA
    int32_t walk_c, walk_h;     // gemm16_h256 tile walk: panels of walk_c column tiles, sub-slabs of walk_h row tiles per XCD (0 = row-major list)
    // LayerNorm applied ALGEBRAICALLY in the epilogue (gemm16_h256 only, EPI_QKVH16 / EPI_GELU16, fp16 operands): A holds the
    // un-normalised rows x, W the folded rows Wf[n][k] = gamma[k] W[n][k] - mean_k(gamma W[n]) (fold_ln_weights: centred, so that
    // sum_k x[k] Wf[n][k] = sum_k (x[k] - mean(x)) gamma[k] W[n][k]), and
    //     LN(x) W^T + b  =  rstd_m * acc[m][n] + bias'[n],     bias' = b + W beta (in `bias`)
    // rowstat = [Mpad] fp32: rstd of every row of A; null = plain product
    const float* rowstat;
    float* rowpart;     // EPI_STREAM16: [N / 64][Mpad][2] partial row statistics (Mpad = M rounded up to 256)
    // EPI_F32 (split-operand GEMM of the trainable path, split.hip): out fp32 = acc * inv_a[0] * inv_b[0] (+ bias) (+ resid)
    const float* inv_a; const float* inv_b;     // device scalars (reciprocal operand scales), null = 1
    int32_t atomic;     // 1: accumulate into out with fp32 atomics (bench knob only: ~20 G atomics/s chip-wide, far too slow)
    int64_t split_stride;   // EPI_F32 split-K: K-split y writes its partial product to out + y*split_stride (bias / resid: reducer)
    const int32_t* skip_last_third;   // EPI_F32: device flag; when it reads 0 the last third of K is all zeros and is not multiplied
};
enum { EPI_OUT16 = 0, EPI_GELU16 = 1, EPI_RESID32 = 2, EPI_PATCH32 = 3, EPI_QKVH16 = 4, EPI_F32 = 5, EPI_PATCH16 = 6, EPI_STREAM16 = 7 };
// EPI_STREAM16 (gemm16_h256 only, fp16, N = ldo): the residual add of a pre-LN tower in the epilogue — out IS the fp16 stream x [Mpad, N],
// read and written in place: x[m] = fp16(x[m] + acc[m] + bias), and `rowpart` [N / 64][Mpad] (sum, sum of squares) receives every
// row's statistics over each 64-column slice (reduced in a fixed order by stream_stats_finalize, rowops.hip).  Rows m = item * qkv_S
// (the CLS rows, whose stream is the fp32 `xc` of the executor) receive the DELTA fp16(acc + bias) alone: the finalize step adds it to
// their fp32 stream and writes the rounded sum back.
// EPI_PATCH16: 16-bit output, patch row m of image m / P -> token row m + m / P + 1 (bias added; the position embedding is added by
// the LayerNorm kernel that reads the rows — rowops.hip, MX_POSROW)
// EPI_QKVH16: 16-bit output scattered head-major, out[item][head][q|k|v][token][64] (item = m / S): every (item, head)
// slice the attention kernel streams is then one contiguous block instead of 128-byte pieces at a 4.6 KB stride.
int launch_gemm16(int dtype16, int mode, const Gemm16Args& a, hipStream_t s);
bool gemm16_runs_h256(int dtype16, int mode, const Gemm16Args& a);       // gemm16.hip: would this product run on gemm16_h256 (EPI_STREAM16 exists only there)?
bool gemm16_takes_rowstat(int dtype16, int mode, const Gemm16Args& a);   // gemm16.hip: would this product run on the kernel that applies LayerNorm in its epilogue?
int launch_layernorm768(int dtype16, const float* x, const float* g, const float* b, float eps, void* out16,
                        float* out32, int64_t rows, hipStream_t s);
// x (+ delta16) -> [sum32 = x + delta] -> LayerNorm -> out16 / out32 (any output may be null; g == null: no LayerNorm)
int launch_add_layernorm768(int dtype16, const float* x, const void* delta16, const float* g, const float* b, float eps,
                            float* sum32, void* out16, float* out32, int64_t rows, hipStream_t s);
// ... with a second 16-bit delta: v = (x + delta16) + delta16b (either may be null)
int launch_add2_layernorm768(int dtype16, const float* x, const void* delta16, const void* delta16b, const float* g,
                             const float* b, float eps, float* sum32, void* out16, float* out32, int64_t rows, hipStream_t s);
// mixed-precision residual stream (rowops.hip: layernorm768_mixed_kernel): CLS rows fp32 in `xc` [items, 768], every other token
// row fp16 in `x16` [items * Ttok, 768]; V = which operands exist
enum { MX_D1 = 1, MX_D2 = 2, MX_LN = 4, MX_RESV = 8, MX_RESY = 16, MX_SRC32 = 32, MX_CLSONLY = 64, MX_POSROW = 128, MX_STAT = 256, MX_ALIAS = 512 };
// MX_ALIAS (with MX_RESY, fp16 operands): out16 IS x16 — the post-LN tower's stream and its LayerNorm image are the same fp16 values, written
// once (every row, the CLS rows too; their fp32 copy still goes to xc)
// MX_STAT (with MX_RESV, without MX_LN): no LayerNorm image; every row's sum goes to the fp16 stream (the CLS rows as well: the stream
// is the next GEMM's A operand) and `stat` [rows] receives rstd of the ROUNDED row — Gemm16Args::rowstat
// MX_POSROW (with MX_SRC32 | MX_D1): the fp32 source is a [Ttok, 768] table indexed by the TOKEN (the position embedding) and the delta is
// the patch embedding of that token; CLS rows take neither (their stream holds cls + pos[0] already)
int launch_layernorm768_mixed(int dtype16, int V, const float* x32, void* x16, float* xc, const void* delta16, const void* delta16b,
                              const float* g, const float* b, float eps, void* out16, int64_t items, int Ttok, hipStream_t s,
                              float* stat = nullptr);
// LayerNorm folded into the weights of the product that consumes it (rowops.hip): Wf[n][k] = fp16(gamma[k] W[n][k] - mean_k(gamma W[n]))
// with sum_k Wf[n][k] = 0 to the last bit that matters, bf[n] = bias[n] + sum_k beta[k] W[n][k]; K = 768, fp16 weights.  Up to 32 jobs in one launch.
struct LnFoldJob { const void* W; const float* bias; const float* g; const float* b; void* Wf; float* bf; int32_t N;
                   int32_t w32;    // 1: W is fp32 (the caller's master copy), 0: fp16
};
int launch_fold_ln_weights(const LnFoldJob* jobs, int n, hipStream_t s);
// after an EPI_STREAM16 product: rstat[m] = rstd of row m from the `nslots` partial sums of rowpart [nslots][Mpad][2]; the CLS rows
// (m = item * Ttok): xc[item] += the fp16 delta the product left in x16[m]; x16[m] = fp16(xc[item]); rstat[m] from the rounded row
int launch_stream_stats_finalize(const float* rowpart, int nslots, int64_t Mpad, void* x16, float* xc, float* rstat, float eps,
                                 int64_t items, int Ttok, hipStream_t s);
int launch_attention16(int dtype16, const void* qkv, const float* key_bias, void* ctx, int64_t items, int S,
                       int heads, hipStream_t s);
// CLS query only: ctx_cls [items, heads*64] (last executed encoder block)
int launch_attention_cls16(int dtype16, const void* qkv, const float* key_bias, void* ctx_cls, int64_t items, int S,
                           int heads, hipStream_t s, const void* q_cls = nullptr);   // q_cls: [items, heads*64] 16-bit CLS queries

// split-operand fp32 GEMM on the 16-bit matrix cores (split.hip); same problem description and TA/TB/ACCUM flags as gemm32
struct Gemm32Prob;
size_t gemm_x3_ws_bytes(int64_t M, int64_t N, int64_t K);
bool gemm_x3_applicable(const Gemm32Prob& p, int flags);
int launch_gemm_x3(const Gemm32Prob& p, int flags, void* ws, size_t ws_bytes, hipStream_t s);
// a group of products with the same flags: one launch for all operand images, one for all split-K sums (split.hip)
size_t gemm_x3_group_ws_bytes(const int64_t* M, const int64_t* N, const int64_t* K, int n);
int launch_gemm_x3_group(const Gemm32Prob* probs, int n, int flags, void* ws, size_t ws_bytes, hipStream_t s);
int gemm_x3_group_max();          // products per group (dev knob x3_group)

// split-operand GEMM on shared planes (gemm16_x3.hip): images A2 [Mpad, 2 kp] = [hi | lo], B2 [Npad, 2 kp]; out fp32 = (Ah Bh^T + Ah Bl^T + Al Bh^T) * inv_a * inv_b
// (+ bias) (+ resid), or raw split-K partials at out + y * split_stride; lo_a / lo_b: device flags "the lo plane has a non-zero element"
struct X3pArgs {
    const _Float16* A2; const _Float16* B2;
    int64_t M; int32_t N, kp;
    const float* bias; const float* resid; float* out; int32_t ldo;
    const float* inv_a; const float* inv_b;
    const int32_t* lo_a; const int32_t* lo_b;
    int64_t split_stride;
};
int launch_gemm16_x3p(const X3pArgs& a, int ksplit, hipStream_t s);

// amax of up to 16 tensors in one launch (split.hip): x row-major [rows, cols] with leading dimension ld; out = the caller's zeroed amax slot
struct AmaxBatch { const float* x[16]; int64_t rows[16], cols[16], ld[16]; uint32_t* out[16]; };
int launch_amax_batch(const AmaxBatch& b, int n, hipStream_t s);

struct Gemm32Prob {
    const float* A; const float* B; const float* bias; const float* resid; const float* act_src; float* C;
    int64_t M; int32_t N; int64_t K;
    int32_t lda, ldb, ldc, ldr;
    DropCfg drop;          // G32_DROPOUT: C = dropout(acc + bias [act]) (+ resid); element index = m*ldc + n
    int64_t ksplit_stride; // internal (split-K through a scratch buffer): K-split y writes its raw partial product at C + y*ksplit_stride
    // gemm_x3 only: caller-owned amax slots of the operands (bit pattern of max|x|, zeroed by the caller before their first
    // use) and whether they already hold the tensor's amax — a tensor read by several products (an activation by its forward
    // and weight-gradient products, a weight by forward and dX) then pays its amax pass once.  null = private slot, computed here.
    uint32_t* amax_a; uint32_t* amax_b;
    int32_t amax_a_ready, amax_b_ready;
    // gemm_x3 only: the operand is exact in fp16 at scale 1 AND its amax slot is preset and ready (taps cached in fp16, cfg->taps_exact16): its lo
    // plane is not written (it would be all zeros; the product skips a plane whose flag stays 0)
    int32_t exact16_a, exact16_b;
    uint32_t* x3_zeroed;   // gemm_x3 only: 12 words the caller has zeroed for this product alone (private amax, 1/scale, lo flags); null = zeroed here
    // weight-gradient products (G32_TA | G32_TB | G32_ACCUM, A stored [K, M]): colsum_a[m] += sum_k A[k][m] — the bias gradient that
    // belongs to the same dY.  gemm32_dw_kernel sums the A tiles it stages anyway (round 3 re-read dY in a colsum launch of its
    // own: 0.94 GB and 20 launches per Cached step); every other route falls back to that launch.  null = not wanted.
    float* colsum_a;
};
// fusion-fed down projection of the separate SANB launches (gemm32.hip: gemm32_n64f_kernel):  F = fuse(a, b, prev) is formed in the
// registers that feed the product, written out once, and  U = F·W^T + bias,  A = act(U)  leave together
struct N64FDesc {
    const float* a; const float* b; const float* prev; int64_t lda, ldb, ldp; const float* gate; int32_t type;   // as FuseTower (sidenet.hip)
    float* F;                      // [M, K] (ld K)
    const float* W; int32_t ldw;   // [64, K]
    const float* bias;             // [64]
    float* U; float* A;            // [M, 64]: pre-activation, activation
    int64_t M; int32_t K;
};
bool gemm32_n64f_ok(const N64FDesc* d, int n);
int launch_gemm32_n64f(const N64FDesc* d, int n, int gelu, hipStream_t s);
// gate-fused dF product of the separate SANB launches (gemm32.hip: gemm32_k64_kernel<true, true>)
struct K64Gate { const float* gate; const float* ga; const float* go; int64_t ldga, ldgo; float* dgate; int32_t scale_prev, store; float* d2; int64_t ldd2; int32_t d2_is_b; };
bool gemm32_k64_gate_ok(const Gemm32Prob* probs, const K64Gate* gates, int nprob);
int launch_gemm32_k64_gate(const Gemm32Prob* probs, const K64Gate* gates, int nprob, hipStream_t s);
// Scratch for split-K of skinny long-K products that are NOT "+=" (their epilogue — bias, activation, masks — has to see the
// complete sum, so the partial products meet in a buffer and a reducer applies it).  Registered by an executor for the
// duration of its call (stream-ordered use; one host thread per process issues the work); null = never split.
void gemm32_set_scratch(float* ws, size_t floats);
// flags for launch_gemm32
enum {
    G32_TA = 1,        // A stored [K, M]
    G32_TB = 2,        // B stored [K, N] (default [N, K], torch Linear weight)
    G32_RELU = 4,      // C = relu(.)
    G32_GELU = 8,      // C = gelu_erf(.)
    G32_ACCUM = 16,    // C += (atomicAdd; enables split-K)
    G32_MUL_RELU_MASK = 32,  // C = (.) * (act_src > 0)
    G32_MUL_GELU_GRAD = 64,  // C = (.) * gelu'(act_src)
    G32_PREACT = 128,
    G32_HINT_B_EXACT16 = 1024,   // gemm_x3 only: the B operand is probably exact in fp16 (cached taps): its lo plane goes last
    G32_DROPOUT = 256, // scale by the dropout keep factor before the residual add  // also store the pre-activation into act_src (as float* out) -- fwd of GELU adapters
};
int launch_gemm32(const Gemm32Prob* probs, int nprob, int flags, hipStream_t s);
