// Multi-head self-attention of the frozen encoders (head_dim 64): softmax(Q K^T / 8 + key_bias) V.
// Replaces HF ViTAttention / BertSelfAttention as reached from Code_Uncached/model/encoders.py:30,86
// (SURVEY.md §8a U1/U2).  ≈4 % of the hot path's FLOPs; sequences are short (197 / 30 tokens), so one workgroup
// owns one (item, head): K and V^T of that head live in LDS, each wave walks 16-query blocks.
//
// MFMA mapping (v_mfma_f32_16x16x32, wave64):
//   * scores are computed TRANSPOSED, S^T = K · Q^T (K rows as the A operand, Q rows as B): a lane then holds, for
//     ONE query (lane&15), keys 16t + 4(lane>>4) + r — the softmax row reduction is a per-lane loop plus two
//     xor-shuffles (16, 32), and the exponentiated registers are ALREADY in B-operand layout for the second
//     product O^T = V^T · P^T (k-slot e of lane group g <-> key 32kb + 4g + e | 32kb + 16 + 4g + (e-4));
//   * V is transposed into LDS while staging (V^T[d][key], row stride padded to 16*NT16+4 elements) so the A
//     operand of the second product is two 8-byte LDS reads per fragment;
//   * K rows are 128 bytes, 16-byte slots XOR-swizzled with (row&7): conflict-free ds_read_b128;
//   * fp32 softmax statistics; masked keys (BERT attention_mask == 0) take the constant fp32-min score exactly like
//     HF's additive mask, so an all-masked padding item attends uniformly; structural pad keys get -inf.
#include "common.h"

namespace {

constexpr float MASK_MIN = -3.4028234663852886e38f;

template <typename T, int NT16>
__global__ __launch_bounds__(256) void attention16_kernel(const typename T::elem* __restrict__ qkv,
                                                          const float* __restrict__ key_bias,
                                                          typename T::elem* __restrict__ ctx, int S, int heads) {
    typedef typename T::elem E;
    typedef typename T::v8 V8;
    typedef typename T::v4 V4;
    constexpr int SP = NT16 * 16;
    constexpr int VT_LD = SP + 4;
    __shared__ __attribute__((aligned(16))) char smem[SP * 128 + 64 * VT_LD * 2 + SP * 4];
    char* sK = smem;
    E* sVt = (E*)(smem + SP * 128);
    float* sKB = (float*)(smem + SP * 128 + 64 * VT_LD * 2);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int item = blockIdx.x / heads, h = blockIdx.x - item * heads;
    const int D = heads * 64, ld = 3 * D;
    const E* base = qkv + (int64_t)item * S * ld + h * 64;

    // ---- stage K (swizzled rows) and V^T ----
    {
        const int c = tid & 7;
        for (int r = tid >> 3; r < SP; r += 32) {
            V8 kv, vv;
            if (r < S) {
                kv = *(const V8*)(base + (int64_t)r * ld + D + c * 8);
                vv = *(const V8*)(base + (int64_t)r * ld + 2 * D + c * 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    kv[e] = (E)0.f;
                    vv[e] = (E)0.f;
                }
            }
            *(V8*)(sK + r * 128 + ((c ^ (r & 7)) << 4)) = kv;
#pragma unroll
            for (int e = 0; e < 8; ++e) sVt[(c * 8 + e) * VT_LD + r] = vv[e];
        }
        for (int r = tid; r < SP; r += 256)
            sKB[r] = r >= S ? -2.0f : (key_bias ? key_bias[(int64_t)item * S + r] : 0.0f);
    }
    __syncthreads();

    const int j = lane & 15, g = lane >> 4;
    const int nqb = (S + 15) >> 4;
    for (int qb = wave; qb < nqb; qb += 4) {
        const int sq = qb * 16 + j;
        const int sqc = sq < S ? sq : S - 1;
        V8 qf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const V8*)(base + (int64_t)sqc * ld + kk * 32 + g * 8);

        // S^T tiles: lane holds query j, keys 16t + 4g + r
        f4 sc[NT16];
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const V8 kf = *(const V8*)(sK + (t * 16 + j) * 128 + (((kk * 4 + g) ^ (j & 7)) << 4));
                acc = T::mfma(kf, qf[kk], acc);
            }
            const f4 kb = *(const f4*)(sKB + t * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = acc[r] * 0.125f;
                sc[t][r] = kb[r] < -1.5f ? -INFINITY : (kb[r] < 0.f ? MASK_MIN : s);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(sc[t][r] - mx);
                sc[t][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;

        // O^T = V^T · P^T
        f4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NT16 / 2; ++kb) {
            V8 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf[e] = T::from_f32(sc[2 * kb][e]);
                pf[4 + e] = T::from_f32(sc[2 * kb + 1][e]);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const E* vr = sVt + (dt * 16 + j) * VT_LD + kb * 32 + g * 4;
                const V4 lo = *(const V4*)vr, hi = *(const V4*)(vr + 16);
                V8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vf[e] = lo[e];
                    vf[4 + e] = hi[e];
                }
                o[dt] = T::mfma(vf, pf, o[dt]);
            }
        }
        if (sq < S) {
            E* op = ctx + ((int64_t)item * S + sq) * D + h * 64 + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                V4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = T::from_f32(o[dt][r] * inv);
                *(V4*)(op + dt * 16) = ov;
            }
        }
    }
}

template <typename T>
int launch_t(const void* qkv, const float* key_bias, void* ctx, int64_t items, int S, int heads, hipStream_t s) {
    typedef typename T::elem E;
    dim3 grid((unsigned)(items * heads)), block(256);
#define IISAN_ATTN_CASE(NT)                                                                                        \
    hipLaunchKernelGGL((attention16_kernel<T, NT>), grid, block, 0, s, (const E*)qkv, key_bias, (E*)ctx, S, heads)
    if (S <= 32) IISAN_ATTN_CASE(2);
    else if (S <= 64) IISAN_ATTN_CASE(4);
    else if (S <= 128) IISAN_ATTN_CASE(8);
    else if (S <= 224) IISAN_ATTN_CASE(14);
    else {
        iisan_set_error("attention16: sequence length %d > 224 not supported", S);
        return IISAN_EBADSHAPE;
    }
#undef IISAN_ATTN_CASE
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

}  // namespace

int launch_attention16(int dtype16, const void* qkv, const float* key_bias, void* ctx, int64_t items, int S, int heads,
                       hipStream_t s) {
    IISAN_CHECK_SHAPE(items > 0 && S > 0 && heads > 0, "attention16: empty problem");
    IISAN_CHECK_SHAPE(items * heads < (1ll << 31), "attention16: grid too large");
    return dtype16 == IISAN_BF16 ? launch_t<BF16>(qkv, key_bias, ctx, items, S, heads, s)
                                 : launch_t<F16>(qkv, key_bias, ctx, items, S, heads, s);
}

extern "C" int iisan_attention16(int32_t dtype16, const void* qkv, const float* key_bias, void* ctx, int64_t items,
                                 int32_t S, int32_t heads, void* stream) {
    return launch_attention16(dtype16, qkv, key_bias, ctx, items, S, heads, (hipStream_t)stream);
}
