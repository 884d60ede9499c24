// Row-wise HBM-bound kernels of the frozen encoders: LayerNorm, patch extraction, embedding gather, CLS taps.
// Roofline: HBM (one read + one write of each row); one wave64 per 768-wide row, 16-byte lane accesses,
// wavefront-shuffle reductions, fp32 statistics (two-pass mean / centred variance in registers).
#include "common.h"

namespace {

// ---- (residual add +) LayerNorm over rows of 768 fp32 (HF nn.LayerNorm, eps inside the sqrt) ----------------------
// v = x + delta16 (delta optional: the 16-bit output of the preceding O / FC2 GEMM, so the fp32 read-modify-write of
// the residual stream happens HERE, in an HBM-bound kernel, instead of stalling the MFMA pipeline in a GEMM epilogue);
// sum32 <- v (optional, may alias x); out32 / out16 <- LN(v) (optional, out32 may alias x).  g == nullptr: add only.
// Which optional operands exist is COMPILE TIME (V bit 0: delta, 1: delta2, 2: sum32, 3: LayerNorm outputs): with run-time
// null tests every load sat behind its own branch and hipcc waited for each one before issuing the next — one load in flight
// per wave, six HBM round trips per row (ISA: load, s_waitcnt vmcnt(0), load, s_waitcnt vmcnt(0), ...).  Now all loads of a
// row are requested first.
template <typename T, int V>
__global__ __launch_bounds__(256) void layernorm768_kernel(const float* x, const typename T::elem* __restrict__ delta,
                                                           const typename T::elem* __restrict__ delta2,
                                                           const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                           float* sum32, typename T::elem* __restrict__ out16,
                                                           float* out32, int64_t rows) {
    constexpr bool D1 = V & 1, D2 = (V & 2) != 0, SUM = (V & 4) != 0, LN = (V & 8) != 0;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    // The fp32 residual stream and the 16-bit deltas are read once and not touched again for gigabytes: non-temporal
    // accesses (measured: 220.7 -> 201.8 us average per launch over a step's 49 launches)
    const float* xr = x + row * 768;
    f4 v[3];
    typename T::v4 d1[3], d2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[i] = __builtin_nontemporal_load((const f4*)(xr + i * 256 + lane * 4));
        if (D1) d1[i] = __builtin_nontemporal_load((const typename T::v4*)(delta + row * 768 + i * 256 + lane * 4));
        if (D2) d2[i] = __builtin_nontemporal_load((const typename T::v4*)(delta2 + row * 768 + i * 256 + lane * 4));
    }
    f4 gg[3], bb[3];
    if (LN) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { gg[i] = *(const f4*)(g + i * 256 + lane * 4); bb[i] = *(const f4*)(b + i * 256 + lane * 4); }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (D1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] += T::to_f32(d1[i][e]);
        }
        if (D2) {       // added AFTER delta, in fp32: (x + delta) + delta2 — the same value as two passes produce
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] += T::to_f32(d2[i][e]);
        }
        if (SUM) __builtin_nontemporal_store(v[i], (f4*)(sum32 + row * 768 + i * 256 + lane * 4));
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    if (!LN) return;
    const float mean = wave_sum(s) * (1.0f / 768.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / 768.0f) + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = i * 256 + lane * 4;
        f4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * gg[i][e] + bb[i][e];
        if (out32) *(f4*)(out32 + row * 768 + c) = y;
        if (out16) {
            typename T::v4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = T::from_f32(y[e]);
            *(typename T::v4*)(out16 + row * 768 + c) = o;
        }
    }
}


// ---- the same with a MIXED-PRECISION residual stream (round 4) ------------------------------------------------------
// The path consumes only the CLS row of every hidden state (Code_Uncached/model/model.py:210-213), and every token row already
// reaches the next GEMM through a 16-bit LayerNorm image.  So the residual stream is kept in fp32 for the CLS rows only
// (`xc` [items, 768], compact: the taps are copies of it) and in fp16 for the patch / word rows (`x16` [rows, 768]; always
// IEEE half, whatever the MFMA operand type — bf16 would keep 8 bits).  A non-CLS row's rounding (2^-11, once per block)
// reaches the CLS row only through the attention average over the item's tokens: measured on the golden ViT-B inputs with
// fp32 arithmetic everywhere else, +9e-5 relative on every tap (all rows in fp16: 6.8e-4) against the 8.8e-4 the 16-bit operands
// cost and the 1.5e-3 budget.  Bytes per token row and ViT block: LN1 10 + LN2 6 = 16 instead of 14 + 8 = 22 (BERT: 8
// instead of 12 per LayerNorm) in kernels that run at the HBM rate.
// HALF a wave per token row (32 lanes x three 8-element pieces), a workgroup = 8 consecutive tokens of ONE item (the CLS test costs no
// division per lane): every access of the fp16 stream, the 16-bit deltas and the 16-bit image is a full 16-byte lane access.  (The
// first version gave a row to a whole wave — 12 elements per lane as three 8-byte pieces: 4.3 TB/s on LN1 where the fp32 kernel's
// 16-byte accesses reach 6.7; MI355X_MICROARCH.md: 8-byte accesses run at 0.54-0.70 x the 16-byte rate.)  V: which operands exist.
template <typename T, int V>
__global__ __launch_bounds__(256) void layernorm768_mixed_kernel(const float* __restrict__ x32, _Float16* x16, float* xc,
                                                                 const typename T::elem* __restrict__ delta,
                                                                 const typename T::elem* __restrict__ delta2,
                                                                 const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                                 typename T::elem* __restrict__ out16, int64_t items, int Ttok,
                                                                 float* __restrict__ stat) {
    constexpr bool D1 = (V & MX_D1) != 0, D2 = (V & MX_D2) != 0, LN = (V & MX_LN) != 0, RESV = (V & MX_RESV) != 0,
                   RESY = (V & MX_RESY) != 0, SRC32 = (V & MX_SRC32) != 0, CLSONLY = (V & MX_CLSONLY) != 0, POSROW = (V & MX_POSROW) != 0,
                   STAT = (V & MX_STAT) != 0, ALIAS = (V & MX_ALIAS) != 0;
    typedef typename T::v8 V8;
    const int lane = threadIdx.x & 31, half = threadIdx.x >> 5;          // 8 half-waves per workgroup
    int64_t item;
    int tok;
    if (CLSONLY) {
        item = (int64_t)blockIdx.x * 8 + half; tok = 0;
        if (item >= items) return;
    } else {
        const int bpi = (Ttok + 7) >> 3;
        item = blockIdx.x / bpi;
        tok = (int)(blockIdx.x - item * bpi) * 8 + half;
        if (tok >= Ttok) return;
    }
    const int64_t row = item * Ttok + tok;
    const bool cls = tok == 0;                                            // uniform within the half-wave
    V8 d1[3], d2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (D1) d1[i] = __builtin_nontemporal_load((const V8*)(delta + row * 768 + i * 256 + lane * 8));
        if (D2) d2[i] = __builtin_nontemporal_load((const V8*)(delta2 + row * 768 + i * 256 + lane * 8));
    }
    float v[3][8];
    if (cls) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f4 a = *(const f4*)(xc + item * 768 + i * 256 + lane * 8), c = *(const f4*)(xc + item * 768 + i * 256 + lane * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = c[e]; }
        }
    } else if (SRC32) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int64_t srow = POSROW ? (int64_t)tok : row;            // POSROW: the position-embedding table (L2-resident), by token
            const f4 a = POSROW ? *(const f4*)(x32 + srow * 768 + i * 256 + lane * 8) : __builtin_nontemporal_load((const f4*)(x32 + srow * 768 + i * 256 + lane * 8));
            const f4 c = POSROW ? *(const f4*)(x32 + srow * 768 + i * 256 + lane * 8 + 4) : __builtin_nontemporal_load((const f4*)(x32 + srow * 768 + i * 256 + lane * 8 + 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = c[e]; }
        }
    } else {
        h8 xh[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) xh[i] = __builtin_nontemporal_load((const h8*)(x16 + row * 768 + i * 256 + lane * 8));
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = (float)xh[i][e];
    }
    // STAT: the fp16 stream is the next product's A operand — the CLS rows go there too, and the statistics are those of the
    // ROUNDED row (what the product multiplies); val is rounded in place
    auto put_resid = [&](int i, float (&val)[8]) {
        if (cls) {
            *(f4*)(xc + item * 768 + i * 256 + lane * 8) = (f4){val[0], val[1], val[2], val[3]};
            *(f4*)(xc + item * 768 + i * 256 + lane * 8 + 4) = (f4){val[4], val[5], val[6], val[7]};
        }
        if ((!cls || STAT) && !ALIAS) {
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)val[e];
            if (STAT) *(h8*)(x16 + row * 768 + i * 256 + lane * 8) = o;      // read again by the GEMM that follows
            else __builtin_nontemporal_store(o, (h8*)(x16 + row * 768 + i * 256 + lane * 8));
            if (STAT) {
#pragma unroll
                for (int e = 0; e < 8; ++e) val[e] = (float)o[e];
            }
        }
    };
    float s = 0.f;
    const bool use_d = !(POSROW && cls);          // POSROW: the CLS slot of the patch-embedding buffer was never written
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (D1 && use_d) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] += T::to_f32(d1[i][e]);
        }
        if (D2 && use_d) {       // added AFTER delta, in fp32: (x + delta) + delta2
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] += T::to_f32(d2[i][e]);
        }
        if (RESV && (STAT || !(POSROW && cls))) put_resid(i, v[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
    }
    if (!LN && !STAT) return;
    auto sum32 = [](float t) {          // over the 32 lanes of the half-wave
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        return t;
    };
    const float mean = sum32(s) * (1.0f / 768.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v[i][e] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(sum32(q) * (1.0f / 768.0f) + eps);
    if (STAT) {
        if (lane == 0) stat[row] = rstd;
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = i * 256 + lane * 8;
        const f4 g0 = *(const f4*)(g + c), g1 = *(const f4*)(g + c + 4), b0 = *(const f4*)(b + c), b1 = *(const f4*)(b + c + 4);
        float y[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = (v[i][e] - mean) * rstd * g0[e] + b0[e];
            y[4 + e] = (v[i][4 + e] - mean) * rstd * g1[e] + b1[e];
        }
        if (RESY) put_resid(i, y);
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = T::from_f32(y[e]);
        *(V8*)(out16 + row * 768 + c) = o;
    }
}

// ---- ViT patch extraction: images fp32 [M,C,R,R] -> patch matrix 16-bit [M*P, C*p*p], col = c*p*p + iy*p + ix ----
// one thread = 8 consecutive ix (two float4 reads, one 16-byte store)
// SRC = float: already normalised pixels; SRC = uint8_t: raw pixels, normalised here exactly as the reference's
// transform does in fp32 — ToTensor (x/255) then Normalize(mean .5, std .5) (`Code_Uncached/data_utils/dataset.py:46-50`)
// — so the 16-bit patch matrix is bit-identical to the fp32 route while the image crosses PCIe/HBM at a quarter the size.
template <typename T, typename SRC>
__global__ __launch_bounds__(256) void vit_im2col_kernel(const SRC* __restrict__ img, typename T::elem* __restrict__ out,
                                                         int64_t M, int C, int R, int p) {
    const int per_row = C * p * p / 8;        // threads per patch row
    const int gp = R / p;                      // patches per image side
    const int64_t total = M * gp * gp * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % per_row);
        const int64_t pr = i / per_row;        // patch row index = m*P + py*gp + px
        const int col = t * 8;
        const int c = col / (p * p), rem = col % (p * p), iy = rem / p, ix = rem % p;
        const int64_t m = pr / (gp * gp);
        const int pp = (int)(pr % (gp * gp)), py = pp / gp, px = pp % gp;
        const SRC* src = img + ((m * C + c) * R + (py * p + iy)) * (int64_t)R + px * p + ix;
        f4 a, b2;
        if constexpr (sizeof(SRC) == 4) {
            a = *(const f4*)src;
            b2 = *(const f4*)(src + 4);
        } else {
            typedef uint8_t U8 __attribute__((ext_vector_type(8)));
            const U8 u = *(const U8*)src;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = __fdiv_rn(__fdiv_rn((float)u[e], 255.0f) - 0.5f, 0.5f);
                b2[e] = __fdiv_rn(__fdiv_rn((float)u[4 + e], 255.0f) - 0.5f, 0.5f);
            }
        }
        typename T::v8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = T::from_f32(a[e]);
            o[4 + e] = T::from_f32(b2[e]);
        }
        *(typename T::v8*)(out + pr * (int64_t)(C * p * p) + col) = o;
    }
}

// CLS rows of the token matrix: X[m*T] = cls + pos[0]
__global__ void vit_cls_rows_kernel(float* __restrict__ X, const float* __restrict__ cls, const float* __restrict__ pos,
                                    int64_t M, int T, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * D) return;
    const int64_t m = i / D;
    const int d = (int)(i % D);
    X[m * T * D + d] = cls[d] + pos[d];
}

// ---- BERT embeddings: LN(word[id] + pos[t] + type[0]) -> X fp32 + H 16-bit; key bias from the attention mask ------
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_ln_kernel(const int64_t* __restrict__ text, const float* __restrict__ word,
                                                            const float* __restrict__ pos, const float* __restrict__ type0,
                                                            const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                            float* __restrict__ X, typename T::elem* __restrict__ H,
                                                            float* __restrict__ key_bias, int64_t M, int W, int vocab,
                                                            _Float16* __restrict__ X16, float* __restrict__ Xc) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M * W) return;
    const int64_t m = row / W;
    const int t = (int)(row % W);
    int64_t id = text[m * 2 * W + t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    if (lane == 0) key_bias[row] = text[m * 2 * W + W + t] != 0 ? 0.0f : -1.0f;
    const float* wr = word + id * 768;
    const float* pr = pos + (int64_t)t * 768;
    f4 v[3];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = i * 256 + lane * 4;
        const f4 a = *(const f4*)(wr + c), p2 = *(const f4*)(pr + c), ty = *(const f4*)(type0 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = a[e] + p2[e] + ty[e];
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = wave_sum(s) * (1.0f / 768.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / 768.0f) + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = i * 256 + lane * 4;
        const f4 gg = *(const f4*)(g + c), bb = *(const f4*)(b + c);
        f4 y;
        typename T::v4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
            o[e] = T::from_f32(y[e]);
        }
        if (X) {
            *(f4*)(X + row * 768 + c) = y;
        } else if (t == 0) {                 // mixed-precision residual stream: CLS rows fp32 (compact), the others fp16
            *(f4*)(Xc + m * 768 + c) = y;
        } else if ((const void*)H != (const void*)X16) {     // (H == X16: the fp16 image is the stream — one store below covers every row)
            typedef _Float16 hv4 __attribute__((ext_vector_type(4)));
            hv4 xh;
#pragma unroll
            for (int e = 0; e < 4; ++e) xh[e] = (_Float16)y[e];
            *(hv4*)(X16 + row * 768 + c) = xh;
        }
        *(typename T::v4*)(H + row * 768 + c) = o;
    }
}

// taps[m, k, :] = X[m*T, :]   (CLS row of every item; Code_Uncached/model/model.py:212-213)
__global__ void gather_cls_kernel(const float* __restrict__ X, float* __restrict__ taps, int64_t M, int T, int D,
                                  int n_taps, int k) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index
    const int d4 = D / 4;
    if (i >= M * d4) return;
    const int64_t m = i / d4;
    const int c = (int)(i % d4) * 4;
    *(f4*)(taps + (m * n_taps + k) * D + c) = *(const f4*)(X + m * T * D + c);
}

// 16-bit rows: out[m, :] = H[m*T, :]  (CLS rows of a token-major 16-bit activation), 16 bytes per thread
__global__ void gather_rows16_kernel(const uint4* __restrict__ H, uint4* __restrict__ out, int64_t M, int T, int D8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * D8) return;
    const int64_t m = i / D8;
    const int c = (int)(i - m * D8);
    out[m * D8 + c] = H[m * T * D8 + c];
}

template <typename T>
__global__ void cast16_kernel(const float* __restrict__ src, typename T::elem* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = T::from_f32(src[i]);
}

}  // namespace

int launch_add2_layernorm768(int dtype16, const float* x, const void* delta16, const void* delta16b, const float* g,
                             const float* b, float eps, float* sum32, void* out16, float* out32, int64_t rows, hipStream_t s) {
    if (rows <= 0) return IISAN_OK;
    dim3 grid((unsigned)ceil_div(rows, 4)), block(256);
    const int v = (delta16 ? 1 : 0) | (delta16b ? 2 : 0) | (sum32 ? 4 : 0) | (g ? 8 : 0);
#define LN768_CASE(V)                                                                                                          \
    case V:                                                                                                                    \
        if (dtype16 == IISAN_BF16)                                                                                             \
            hipLaunchKernelGGL((layernorm768_kernel<BF16, V>), grid, block, 0, s, x, (const __bf16*)delta16, (const __bf16*)delta16b, g, b, eps, sum32, (__bf16*)out16, out32, rows); \
        else                                                                                                                   \
            hipLaunchKernelGGL((layernorm768_kernel<F16, V>), grid, block, 0, s, x, (const _Float16*)delta16, (const _Float16*)delta16b, g, b, eps, sum32, (_Float16*)out16, out32, rows); \
        break
    switch (v) {
        LN768_CASE(0); LN768_CASE(1); LN768_CASE(2); LN768_CASE(3); LN768_CASE(4); LN768_CASE(5); LN768_CASE(6); LN768_CASE(7);
        LN768_CASE(8); LN768_CASE(9); LN768_CASE(10); LN768_CASE(11); LN768_CASE(12); LN768_CASE(13); LN768_CASE(14); LN768_CASE(15);
    }
#undef LN768_CASE
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}


// mixed-precision residual stream (layernorm768_mixed_kernel): V = MX_* flags of the operands that exist
int launch_layernorm768_mixed(int dtype16, int V, const float* x32, void* x16, float* xc, const void* delta16, const void* delta16b,
                              const float* g, const float* b, float eps, void* out16, int64_t items, int Ttok, hipStream_t s,
                              float* stat) {
    if (items <= 0) return IISAN_OK;
    IISAN_CHECK_SHAPE(!(V & MX_STAT) || stat, "layernorm768_mixed: MX_STAT needs the statistics buffer");
    const int64_t blocks = (V & MX_CLSONLY) ? ceil_div(items, 8) : items * ((Ttok + 7) / 8);
    IISAN_CHECK_SHAPE(blocks < (1ll << 31), "layernorm768_mixed: grid too large");
    dim3 grid((unsigned)blocks), block(256);
#define MX_CASE(VV)                                                                                                            \
    case VV:                                                                                                                   \
        if (dtype16 == IISAN_BF16)                                                                                             \
            hipLaunchKernelGGL((layernorm768_mixed_kernel<BF16, VV>), grid, block, 0, s, x32, (_Float16*)x16, xc, (const __bf16*)delta16, (const __bf16*)delta16b, g, b, eps, (__bf16*)out16, items, Ttok, stat); \
        else                                                                                                                   \
            hipLaunchKernelGGL((layernorm768_mixed_kernel<F16, VV>), grid, block, 0, s, x32, (_Float16*)x16, xc, (const _Float16*)delta16, (const _Float16*)delta16b, g, b, eps, (_Float16*)out16, items, Ttok, stat); \
        break
    switch (V) {
        MX_CASE(MX_SRC32 | MX_RESV | MX_LN);               // ViT block 0, LN1: fp32 embeddings -> fp16 stream + LN image
        MX_CASE(MX_SRC32 | MX_POSROW | MX_D1 | MX_RESV | MX_LN);   // ViT block 0, LN1: position table + 16-bit patch embedding -> stream + image
        MX_CASE(MX_D1 | MX_LN);                            // ViT LN2: LN(x + dO), x not written
        MX_CASE(MX_D1 | MX_D2 | MX_RESV | MX_LN);          // ViT LN1: x += dO + dF; LN
        MX_CASE(MX_D1 | MX_D2 | MX_RESV | MX_CLSONLY);     // ViT closing add of the CLS rows (hidden state 12)
        MX_CASE(MX_D1 | MX_LN | MX_RESY);                  // BERT: x = LN(x + d)
        MX_CASE(MX_D1 | MX_LN | MX_RESY | MX_ALIAS);       // BERT, fp16 operands: ... and the image IS the stream
        // LayerNorm applied by the consuming product (Gemm16Args::rowstat): the stream and the row statistics only
        MX_CASE(MX_SRC32 | MX_POSROW | MX_D1 | MX_RESV | MX_STAT);   // ViT block 0
        MX_CASE(MX_D1 | MX_RESV | MX_STAT);                // ViT LN1 (x += dF) and LN2 (x += dO)
        MX_CASE(MX_D1 | MX_RESV | MX_CLSONLY);             // ViT closing add of the CLS rows when x + dO is in the stream already
        default: iisan_set_error("layernorm768_mixed: operand set %d not instantiated", V); return IISAN_EBADSHAPE;
    }
#undef MX_CASE
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

int launch_add_layernorm768(int dtype16, const float* x, const void* delta16, const float* g, const float* b, float eps,
                            float* sum32, void* out16, float* out32, int64_t rows, hipStream_t s) {
    return launch_add2_layernorm768(dtype16, x, delta16, nullptr, g, b, eps, sum32, out16, out32, rows, s);
}

int launch_layernorm768(int dtype16, const float* x, const float* g, const float* b, float eps, void* out16,
                        float* out32, int64_t rows, hipStream_t s) {
    return launch_add_layernorm768(dtype16, x, nullptr, g, b, eps, nullptr, out16, out32, rows, s);
}

int launch_vit_im2col(int dtype16, const void* img, int img_u8, void* out, int64_t M, int C, int R, int p, hipStream_t s) {
    const int64_t total = M * (R / p) * (R / p) * (C * p * p / 8);
    const unsigned grid = (unsigned)(ceil_div(total, 256) < 262144 ? ceil_div(total, 256) : 262144);
    if (img_u8) {
        if (dtype16 == IISAN_BF16)
            hipLaunchKernelGGL((vit_im2col_kernel<BF16, uint8_t>), dim3(grid), dim3(256), 0, s, (const uint8_t*)img, (__bf16*)out, M, C, R, p);
        else
            hipLaunchKernelGGL((vit_im2col_kernel<F16, uint8_t>), dim3(grid), dim3(256), 0, s, (const uint8_t*)img, (_Float16*)out, M, C, R, p);
    } else {
        if (dtype16 == IISAN_BF16)
            hipLaunchKernelGGL((vit_im2col_kernel<BF16, float>), dim3(grid), dim3(256), 0, s, (const float*)img, (__bf16*)out, M, C, R, p);
        else
            hipLaunchKernelGGL((vit_im2col_kernel<F16, float>), dim3(grid), dim3(256), 0, s, (const float*)img, (_Float16*)out, M, C, R, p);
    }
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

int launch_vit_cls_rows(float* X, const float* cls, const float* pos, int64_t M, int T, int D, hipStream_t s) {
    hipLaunchKernelGGL(vit_cls_rows_kernel, dim3((unsigned)ceil_div(M * D, 256)), dim3(256), 0, s, X, cls, pos, M, T, D);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

int launch_bert_embed_ln(int dtype16, const int64_t* text, const float* word, const float* pos, const float* type0,
                         const float* g, const float* b, float eps, float* X, void* H, float* key_bias, int64_t M,
                         int W, int vocab, hipStream_t s, void* X16, float* Xc) {       // X == null: mixed stream (X16 + Xc)
    dim3 grid((unsigned)ceil_div(M * W, 4)), block(256);
    if (dtype16 == IISAN_BF16)
        hipLaunchKernelGGL(bert_embed_ln_kernel<BF16>, grid, block, 0, s, text, word, pos, type0, g, b, eps, X, (__bf16*)H, key_bias, M, W, vocab, (_Float16*)X16, Xc);
    else
        hipLaunchKernelGGL(bert_embed_ln_kernel<F16>, grid, block, 0, s, text, word, pos, type0, g, b, eps, X, (_Float16*)H, key_bias, M, W, vocab, (_Float16*)X16, Xc);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

int launch_gather_rows16(const void* H, void* out, int64_t M, int T, int D, hipStream_t s) {
    const int D8 = D / 8;
    hipLaunchKernelGGL(gather_rows16_kernel, dim3((unsigned)ceil_div(M * D8, 256)), dim3(256), 0, s, (const uint4*)H, (uint4*)out, M, T, D8);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

int launch_gather_cls(const float* X, float* taps, int64_t M, int T, int D, int n_taps, int k, hipStream_t s) {
    hipLaunchKernelGGL(gather_cls_kernel, dim3((unsigned)ceil_div(M * (D / 4), 256)), dim3(256), 0, s, X, taps, M, T, D, n_taps, k);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

extern "C" int iisan_layernorm768(int32_t dtype16, const float* x, const float* g, const float* b, float eps,
                                  void* out16, float* out32, int64_t rows, void* stream) {
    return launch_layernorm768(dtype16, x, g, b, eps, out16, out32, rows, (hipStream_t)stream);
}

// ---- packed tap store gather: out[m, :] = fp32(table[ids[m], :]), rows of `row_elems` (multiple of 8) ----
// one thread = 8 consecutive elements (16-B loads from a 16-bit store, 2x16-B from an fp32 one; two 16-B stores)
template <typename TS>
__global__ __launch_bounds__(256) void gather_taps_kernel(const TS* __restrict__ table, const int64_t* __restrict__ ids,
                                                          float* __restrict__ out, int64_t M, int64_t row_elems, int64_t rows) {
    const int64_t per_row = row_elems / 8;
    const int64_t total = M * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / per_row, c = (i - m * per_row) * 8;
        int64_t id = ids[m];
        id = id < 0 ? 0 : (id >= rows ? rows - 1 : id);
        const TS* src = table + id * row_elems + c;
        f4 lo, hi;
        if constexpr (sizeof(TS) == 4) {
            lo = *(const f4*)src;
            hi = *(const f4*)(src + 4);
        } else {
            typedef TS V8 __attribute__((ext_vector_type(8)));
            const V8 v = *(const V8*)src;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = (float)v[e]; hi[e] = (float)v[4 + e]; }
        }
        *(f4*)(out + m * row_elems + c) = lo;
        *(f4*)(out + m * row_elems + c + 4) = hi;
    }
}

extern "C" int iisan_gather_taps(int32_t store_dtype, const void* table, int64_t rows, const int64_t* ids, float* out,
                                 int64_t M, int64_t row_elems, void* stream) {
    if (M <= 0) return IISAN_OK;
    IISAN_CHECK_SHAPE(row_elems > 0 && row_elems % 8 == 0 && rows > 0, "gather_taps: row length %lld must be a positive multiple of 8", (long long)row_elems);
    const int64_t total = M * (row_elems / 8);
    const unsigned grid = (unsigned)(ceil_div(total, 256) < 262144 ? ceil_div(total, 256) : 262144);
    hipStream_t s = (hipStream_t)stream;
    if (store_dtype == IISAN_F32)
        hipLaunchKernelGGL(gather_taps_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)table, ids, out, M, row_elems, rows);
    else if (store_dtype == IISAN_BF16)
        hipLaunchKernelGGL(gather_taps_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)table, ids, out, M, row_elems, rows);
    else if (store_dtype == IISAN_F16)
        hipLaunchKernelGGL(gather_taps_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)table, ids, out, M, row_elems, rows);
    else {
        iisan_set_error("gather_taps: unknown store dtype %d", store_dtype);
        return IISAN_EBADSHAPE;
    }
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

extern "C" int iisan_cast16(int32_t dtype16, const float* src, void* dst, int64_t n, void* stream) {
    if (n <= 0) return IISAN_OK;
    const unsigned grid = (unsigned)(ceil_div(n, 256) < 65536 ? ceil_div(n, 256) : 65536);
    if (dtype16 == IISAN_BF16)
        hipLaunchKernelGGL(cast16_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, n);
    else
        hipLaunchKernelGGL(cast16_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, n);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

// ---- LayerNorm folded into the weights of the product that consumes it (Gemm16Args::rowstat) ----
// LN(x) W^T + b = rstd (x Wf^T) + bf with Wf[n][k] = gamma[k] W[n][k] - (1/K) sum_j gamma[j] W[n][j] — CENTRED rows: sum_k x[k] Wf[n][k] =
// sum_k (x[k] - mean(x)) gamma[k] W[n][k] for every x, so the product of the un-normalised row needs no mean correction — and
// bf = b + W beta.  The fp16 rounding of Wf leaves a row sum d_n = sum_k (Wf16 - Wf) != 0 and with it an error mean(x) rstd d_n that
// grows with |mean| / std of the row (CPU emulation: 1.7e-4 -> 5.1e-4 relative at |mean| = 3 std with round-to-nearest, |d_n| ~ 8 ulp).
// SUM-PRESERVING ROUNDING: round to nearest, then move the few elements whose rounding was the closest call (residual nearest +-1/2
// ulp, on the side d_n needs) to their other neighbour — a bisection over the residual threshold finds the set whose ulps add up to
// d_n; each such move costs next to nothing ((1 - 2|r|) ulp^2 of squared error at residual r ~ 1/2), so the weights keep the noise of
// plain rounding (the first version diffused every element's error into its neighbour: row sums as good, 1.4 x the noise) — and what
// the bisection leaves (< one ulp) goes into one element.  |d_n| ends below one ulp of one weight, ~1e-5 of the ~4e-4 plain rounding
// leaves.  W is read from the caller's fp32 master when the weights struct has one (iisan_layer_weights.qkv_w32 / fc1_w32): ONE rounding
// of gamma W - mean instead of a second one on top of the 16-bit copy's.  The weights are frozen, but the ABI is stateless: folded once
// per forward call (200-300 MB of traffic for ViT-B, ~40-60 us) into the workspace.
// Half a wave per weight row (K = 768: three 8-element pieces per lane), 8 rows per workgroup, every job in one launch.
struct LnFoldArgs { LnFoldJob j[32]; int32_t row0[33]; int32_t n; };

__global__ __launch_bounds__(256) void fold_ln_weights_kernel(LnFoldArgs a) {
    const int lane = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= a.row0[a.n]) return;
    int ji = 0;
    while (row >= a.row0[ji + 1]) ++ji;
    const LnFoldJob& J = a.j[ji];
    const int n = row - a.row0[ji];
    _Float16* wf = (_Float16*)J.Wf + (int64_t)n * 768;
    auto sum32 = [](float t) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        return t;
    };
    float v[3][8];
    float cs = 0.f, bw = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = i * 256 + lane * 8;
        float x[8];
        if (J.w32) {
            const float* w = (const float*)J.W + (int64_t)n * 768 + c;
            const f4 x0 = *(const f4*)w, x1 = *(const f4*)(w + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = x0[e]; x[4 + e] = x1[e]; }
        } else {
            const h8 xh = *(const h8*)((const _Float16*)J.W + (int64_t)n * 768 + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (float)xh[e];
        }
        const f4 g0 = *(const f4*)(J.g + c), g1 = *(const f4*)(J.g + c + 4), b0 = *(const f4*)(J.b + c), b1 = *(const f4*)(J.b + c + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[i][e] = x[e] * (e < 4 ? g0[e] : g1[e - 4]);
            cs += v[i][e];
            bw += x[e] * (e < 4 ? b0[e] : b1[e - 4]);
        }
    }
    cs = sum32(cs); bw = sum32(bw);
    const float mu = cs * (1.0f / 768.0f);
    // round to nearest; r = residual in units of the element's OWN ulp towards the side the row sum needs (dir = sign of the total residual)
    _Float16 o[3][8];
    float res = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[i][e] -= mu;
            o[i][e] = (_Float16)v[i][e];
            res += v[i][e] - (float)o[i][e];
        }
    const float R = sum32(res);                       // what the emitted row is short of (target - emitted)
    const float dir = R >= 0.f ? 1.f : -1.f;
    // the neighbour of o on the `dir` side and what moving there adds to the row sum (its ulp on that side)
    float step[3][8], frac[3][8];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float of = (float)o[i][e];
            const unsigned short bits = __builtin_bit_cast(unsigned short, o[i][e]);
            // next representable half away from / towards zero: +-1 on the magnitude bits (zero: the smallest subnormal of sign dir)
            const bool up_mag = (of > 0.f) == (dir > 0.f) || of == 0.f;
            unsigned short nb = of == 0.f ? (unsigned short)(dir > 0.f ? 1 : 0x8001) : (unsigned short)(up_mag ? bits + 1 : bits - 1);
            const float nf = (float)__builtin_bit_cast(_Float16, nb);
            step[i][e] = (nf - of) * dir;                                    // > 0
            frac[i][e] = (v[i][e] - of) * dir / step[i][e];                  // in [-1/2, 1/2]: how far the target lies towards that neighbour
        }
    // bisection over the threshold t: move every element with frac > t; moved(t) decreases from ~sum of all steps (t = -1/2) to 0 (t = 1/2)
    float lo = 0.f, hi = 0.5f;                        // elements with frac <= 0 were rounded away from that side: never moved
    const float need = R * dir;                       // >= 0
#pragma unroll 1
    for (int it = 0; it < 12; ++it) {
        const float t = 0.5f * (lo + hi);
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) m += frac[i][e] > t ? step[i][e] : 0.f;
        m = sum32(m);
        if (m > need) lo = t; else hi = t;            // too much moved: raise the threshold
    }
    float moved = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (frac[i][e] > hi) {
                o[i][e] = (_Float16)((float)o[i][e] + dir * step[i][e]);
                moved += step[i][e];
            }
    const float rest = (need - sum32(moved)) * dir;   // |rest| < one ulp of the last element not moved: into one element, re-rounded
    if (lane == 0) o[0][0] = (_Float16)((float)o[0][0] + rest);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        h8 o8;
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = o[i][e];
        *(h8*)(wf + i * 256 + lane * 8) = o8;
    }
    if (lane == 0) J.bf[n] = (J.bias ? J.bias[n] : 0.f) + bw;
}

int launch_fold_ln_weights(const LnFoldJob* jobs, int n, hipStream_t s) {
    IISAN_CHECK_SHAPE(n >= 0 && n <= 32, "fold_ln_weights: %d jobs (at most 32 per launch)", n);
    if (n == 0) return IISAN_OK;
    LnFoldArgs a{};
    a.n = n;
    for (int i = 0; i < n; ++i) { a.j[i] = jobs[i]; a.row0[i + 1] = a.row0[i] + jobs[i].N; }
    hipLaunchKernelGGL(fold_ln_weights_kernel, dim3((unsigned)ceil_div((int64_t)a.row0[n], 8)), dim3(256), 0, s, a);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

// ---- after an EPI_STREAM16 product (gemm16_h256.hip): the rows' rstd from the per-slice partial sums, and the CLS rows ----
// Blocks [0, ceil(rows / 256)): one thread per row sums the `nslots` (sum, sum of squares) pairs in slice order — a fixed order, so the
// statistics are bit-reproducible — and writes rstd; single-pass variance in fp32 (E[x^2] - mean^2: the stream's rows have |mean| well
// under their spread; relative error ~1e-7 (1 + mean^2 / var)).  CLS rows are skipped there and done by the blocks behind: half a wave
// per item adds the fp16 delta the product left in the stream's CLS slot to the item's fp32 stream, writes the rounded sum back (the
// next product's A operand) and the rstd of the ROUNDED row, two-pass.
__global__ __launch_bounds__(256) void stream_stats_finalize_kernel(const f2* __restrict__ part, int nslots, int64_t Mpad, _Float16* x16,
                                                                    float* xc, float* __restrict__ rstat, float eps, int64_t items,
                                                                    int Ttok, unsigned row_blocks) {
    const int64_t rows = items * Ttok;
    if (blockIdx.x < row_blocks) {
        const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (row >= rows || row % Ttok == 0) return;
        float s = 0.f, q = 0.f;
        if (nslots == 12) {
            // D = 768: all twelve slice partials requested at once, same summation order, same bits (round 5; no measurable change —
            // 8.2 - 10.3 us per launch either way: the launch is its boundary + the CLS blocks, not this loop)
            f2 v[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = part[(int64_t)j * Mpad + row];
#pragma unroll
            for (int j = 0; j < 12; ++j) { s += v[j][0]; q += v[j][1]; }
        } else {
            for (int j = 0; j < nslots; ++j) {
                const f2 v = part[(int64_t)j * Mpad + row];
                s += v[0]; q += v[1];
            }
        }
        const float mean = s * (1.0f / 768.0f);
        const float var = fmaxf(q * (1.0f / 768.0f) - mean * mean, 0.f);
        rstat[row] = rsqrtf(var + eps);
        return;
    }
    const int lane = threadIdx.x & 31;
    const int64_t item = (int64_t)(blockIdx.x - row_blocks) * 8 + (threadIdx.x >> 5);
    if (item >= items) return;
    const int64_t row = item * Ttok;
    float v[3][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = i * 256 + lane * 8;
        const h8 d = *(const h8*)(x16 + row * 768 + c);
        const f4 a = *(const f4*)(xc + item * 768 + c), b = *(const f4*)(xc + item * 768 + c + 4);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[i][e] = (e < 4 ? a[e] : b[e - 4]) + (float)d[e];
            o[e] = (_Float16)v[i][e];
        }
        *(f4*)(xc + item * 768 + c) = (f4){v[i][0], v[i][1], v[i][2], v[i][3]};
        *(f4*)(xc + item * 768 + c + 4) = (f4){v[i][4], v[i][5], v[i][6], v[i][7]};
        *(h8*)(x16 + row * 768 + c) = o;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[i][e] = (float)o[e]; s += v[i][e]; }
    }
    auto sum32 = [](float t) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        return t;
    };
    const float mean = sum32(s) * (1.0f / 768.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    const float rstd = rsqrtf(sum32(q) * (1.0f / 768.0f) + eps);
    if (lane == 0) rstat[row] = rstd;
}

int launch_stream_stats_finalize(const float* rowpart, int nslots, int64_t Mpad, void* x16, float* xc, float* rstat, float eps,
                                 int64_t items, int Ttok, hipStream_t s) {
    if (items <= 0) return IISAN_OK;
    const int64_t row_blocks = ceil_div(items * Ttok, 256), cls_blocks = ceil_div(items, 8);
    IISAN_CHECK_SHAPE(row_blocks + cls_blocks < (1ll << 31), "stream_stats_finalize: grid too large");
    hipLaunchKernelGGL(stream_stats_finalize_kernel, dim3((unsigned)(row_blocks + cls_blocks)), dim3(256), 0, s, (const f2*)rowpart, nslots, Mpad,
                       (_Float16*)x16, xc, rstat, eps, items, Ttok, (unsigned)row_blocks);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}
