// Multi-head self-attention of the frozen encoders (head_dim 64): softmax(Q K^T / 8 + key_bias) V.
// Input is the head-major QKV tensor [item][head][q|k|v][S][64] the QKV GEMM epilogue writes (EPI_QKVH16), so a
// workgroup streams three contiguous ~25 KB blocks (measured: the token-major layout's 128-byte pieces at a 4.6 KB
// stride capped the whole kernel at 1.85 TB/s of HBM reads).
// Replaces HF ViTAttention / BertSelfAttention as reached from Code_Uncached/model/encoders.py:30,86
// (SURVEY.md §8a U1/U2).  ≈4 % of the hot path's FLOPs; sequences are short (197 / 30 tokens), so one workgroup
// owns one (item, head): K and V^T of that head live in LDS, each wave walks 16-query blocks.
//
// MFMA mapping (v_mfma_f32_16x16x32, wave64):
//   * scores are computed TRANSPOSED, S^T = K · Q^T (K rows as the A operand, Q rows as B): a lane then holds, for
//     ONE query (lane&15), keys 16t + 4(lane>>4) + r — the softmax row reduction is a per-lane loop plus two
//     xor-shuffles (16, 32), and the exponentiated registers are ALREADY in B-operand layout for the second
//     product O^T = V^T · P^T (k-slot e of lane group g <-> key 32kb + 4g + e | 32kb + 16 + 4g + (e-4));
//   * V is transposed into LDS while staging (V^T[d][slot], keys permuted inside groups of 32: see VT_LD) so the A
//     operand of the second product is ONE 16-byte LDS read per fragment;
//   * K rows are 128 bytes, 16-byte slots XOR-swizzled with (row&7): conflict-free ds_read_b128;
//   * fp32 softmax statistics; masked keys (BERT attention_mask == 0) take the constant fp32-min score exactly like
//     HF's additive mask, so an all-masked padding item attends uniformly; structural pad keys get -inf.
#include "common.h"
#ifndef ATTN_KBATCH
#define ATTN_KBATCH 4
#endif

#ifdef ATTN_DEBUG_BITS
static int g_attn_dbg = 0;            // ablation builds only (tools/attn_time.py): bits 0..4 = kernel ablations, bits 8..11 = heads per workgroup
IISAN_DEV_KNOB(attn_debug, g_attn_dbg);
#else
static constexpr int g_attn_dbg = 0;
#endif

namespace {

// masked keys carry this RAW score (before the log2(e)/8 scaling): every real score is absorbed by it, like HF's
// additive fp32-min mask, and MASK_RAW * c2 stays finite
constexpr float MASK_RAW = -0x1p126f;      // a power of two: MASK_RAW * c2 is exact, so the fused scale-and-shift below is exactly 0 on all-masked rows

// One workgroup = one item x HPW consecutive heads, software-pipelined: while head h is being computed out of LDS,
// the Q/K/V registers for head h+1 are already being filled from HBM (measured on the unpipelined version: the load
// phase and the compute phase of a workgroup did not overlap at all and the kernel ran at 2.4 TB/s).
// 16-key tail product of an odd tile count (ViT: 13 tiles): v_mfma_f32_16x16x16 with 4 contraction slots per lane (key base + 4 g + e)
template <typename T> struct Mfma16k16;
template <> struct Mfma16k16<F16> {
    static __device__ __forceinline__ f4 run(h4 a, h4 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16k16<BF16> {
    typedef short s4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f4 run(b4 a, b4 b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0);
    }
};

// PF (prefetch): true (the product) — the next head's Q / K / V registers fill from HBM during this head's compute (92 registers held
// across it: 227 VGPRs, two waves per SIMD, two workgroups per CU); false (round 4 experiment, no longer instantiated) =
// no register prefetch, 136 VGPRs and 54.6 KB of LDS so that THREE workgroups share a CU: one loads and stages while two compute.
// Measured 4 % SLOWER (415-421 against 398-411 us per ViT layer): a third wave per SIMD does not make up for the load phase a
// workgroup now waits out.
// MASKALL (round 5): true = the per-key limit is applied to every key tile (key_bias present, or a sequence that leaves more than the last
// tile padded); false = compile-time knowledge that only the LAST tile can hold pad slots (no key_bias and S > 16 (NT16 - 1): ViT) — the
// run-time test per tile made every tile its own basic block (~100 branches in the unrolled block loop).
template <typename T, int NT16, bool PF, bool MASKALL>
__global__ __launch_bounds__(256, PF ? 2 : 3) void attention16_kernel(const typename T::elem* __restrict__ qkv,
                                                             const float* __restrict__ key_bias,
                                                             typename T::elem* __restrict__ ctx, int S, int heads, int hpw,
                                                             int dbg_arg) {
    typedef typename T::elem E;
    typedef typename T::v8 V8;
    typedef typename T::v4 V4;
#ifndef ATTN_DEBUG_BITS
    // the ablation bits of tools/attn_time.py (dev switch attn_debug: 1 = no V^T write, 8 = no store, 16 = no K write, 32 = no global loads, 64 = no
    // query blocks) are compiled in
    // only with -DATTN_DEBUG_BITS; the product build folds the tests away
    (void)dbg_arg;
    constexpr int dbg = 0;
#else
    const int dbg = dbg_arg;
#endif
    constexpr int SP = NT16 * 16;
    // V^T[d][slot] in LDS, keys PERMUTED inside every group of 32 so that the eight contraction values a lane needs for one P.V step — keys
    // 32 kb + 4 g + (0..3) of score tile 2 kb and 32 kb + 16 + 4 g + (0..3) of tile 2 kb + 1 — are contiguous: key 16 t + 4 g' + r sits at slot
    // 32 (t >> 1) + 8 g' + 4 (t & 1) + r, and a fragment is ONE ds_read_b128 (round 6; before: two 8-byte pieces 32 bytes apart, read as a
    // ds_read2_b64 — 8 LDS cycles per wave-instruction instead of 4, MI355X_MICROARCH.md LDS table; the reads were a third of the kernel's LDS time).
    // Row stride: the slots an odd tile count leaves half-filled count, + 16 elements: rows 16 (mod 32) elements apart put the 16 rows of a
    // ds_read_b128 lane group on 16 distinct bank quads (208 -> 240: K + V^T + key limits = 58,176 B, two workgroups per CU as the registers allow).
    constexpr int VT_LD = 32 * ((NT16 + 1) / 2) + 16;
    constexpr int MAXQB = (NT16 + 3) / 4;            // 16-query blocks per wave
    constexpr int KP = (SP + 31) / 32;                // K passes: 32 rows per pass (the last may be partial: NT16 odd)
    constexpr int VP = (SP / 4 + 31) / 32;            // V passes: 32 four-key groups per pass
    __shared__ __attribute__((aligned(16))) char smem[SP * 128 + 64 * VT_LD * 2 + SP * 4];
    char* sK = smem;
    E* sVt = (E*)(smem + SP * 128);
    float* sKB = (float*)(smem + SP * 128 + 64 * VT_LD * 2);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int groups = heads / hpw;
    const int item = blockIdx.x / groups, h0 = (blockIdx.x - item * groups) * hpw;
    const int D = heads * 64;
    const int j = lane & 15, g = lane >> 4;
    const int nqb = (S + 15) >> 4;
    const int c = tid & 7, r0 = tid >> 3;

    V8 kreg[KP], vreg[VP][4], qnext[MAXQB][2];
    auto load_head = [&](int h) {
        if (dbg & 32) return;                   // ablation (debug builds): no global loads — compute, staging and stores on whatever the registers hold
        // head-major input [item][head][q|k|v][S][64]: three contiguous blocks per (item, head)
        const E* qb_ = qkv + ((int64_t)item * heads + h) * 3 * S * 64;
        const E* kb_ = qb_ + (int64_t)S * 64;
        const E* vb_ = kb_ + (int64_t)S * 64;
        // The lane-dependent offsets are RECOMPUTED here from a laundered thread id: hoisted out of the head loop they
        // stayed live across it, were spilled (the kernel sits at 256 VGPRs) and every scratch reload is followed by an
        // `s_waitcnt vmcnt(0)` — which also waits for the prefetch loads in flight and for all earlier context stores.
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int wave = t2 >> 6, j = t2 & 15, g = (t2 >> 4) & 3, c = t2 & 7, r0 = t2 >> 3;
#pragma unroll
        for (int i = 0; i < MAXQB; ++i) {
            int sq = (wave + 4 * i) * 16 + j;
            sq = sq < S ? sq : S - 1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) qnext[i][kk] = *(const V8*)(qb_ + (unsigned)(sq * 64 + kk * 32 + g * 8));   // 32-bit lane offsets from a uniform base: as
                                                                                              // 64-bit addresses they were spilled (see the store below)
        }
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            const int r = r0 + 32 * p;
            // K and V of a head are read by exactly one workgroup, once: non-temporal (634 -> 619 us; nt on the Q loads or
            // on the context stores made it slower)
            // pad slots (r >= S) re-read the last real row instead of being zero-filled (a clamp instead of four selects
            // per fragment): their scores are forced to -inf by the per-key limit below and their P is exactly 0
            kreg[p] = __builtin_nontemporal_load((const V8*)(kb_ + (unsigned)((r < S ? r : S - 1) * 64 + c * 8)));
        }
#pragma unroll
        for (int p = 0; p < VP; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = (r0 + 32 * p) * 4 + r;
                // key groups beyond the padded length are never written to LDS: no load for them (with short sequences
                // — BERT, S = 30 — most lanes are in that case, and their clamped re-reads cost 10 %; for long sequences the
                // test itself costs 2 %, so it is compiled in for SP <= 64 only)
                if (SP > 64 || r0 + 32 * p < SP / 4) vreg[p][r] = __builtin_nontemporal_load((const V8*)(vb_ + (unsigned)((key < S ? key : S - 1) * 64 + c * 8)));
            }
    };

    if (PF) load_head(h0);
    for (int r = tid; r < SP; r += 256)
        sKB[r] = r >= S ? -INFINITY : ((key_bias && key_bias[(int64_t)item * S + r] < 0.f) ? MASK_RAW : INFINITY);   // per-key upper limit of the score

    // exp(s/8 - m) = exp2(acc * c2 - m2),  c2 = log2(e) / 8
    // (an odd tile count ends with a 16-key product of its own, Mfma16k16: no zero-filled half step, no LDS for it)
    const float c2 = 0.18033688011112042f;
#pragma unroll 1
    for (int hi = 0; hi < hpw; ++hi) {
        const int h = h0 + hi;
        if (!PF) load_head(h);                // (the other workgroups of the CU compute meanwhile)
        if (hi > 0) __syncthreads();          // every wave is done reading the previous head's K / V^T
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            const int r = r0 + 32 * p;
            if ((SP % 32 == 0 || r < SP) && !(dbg & 16)) *(V8*)(sK + r * 128 + ((c ^ (r & 7)) << 4)) = kreg[p];
        }
        // V^T: a thread owns 4 consecutive keys x 8 head dims -> eight 8-byte LDS writes per pass
#pragma unroll
        for (int p = 0; p < VP; ++p) {
            const int kg = r0 + 32 * p;
            if (kg < SP / 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    V4 t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] = vreg[p][r][e];
                    if (!(dbg & 1)) *(V4*)(sVt + (c * 8 + e) * VT_LD + 32 * (kg >> 3) + 8 * (kg & 3) + 4 * ((kg >> 2) & 1)) = t;
                }
            }
        }
        V8 qf[MAXQB][2];
#pragma unroll
        for (int i = 0; i < MAXQB; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) qf[i][kk] = qnext[i][kk];
        __syncthreads();
        if (PF && hi + 1 < hpw) load_head(h + 1);   // in flight during this head's compute

#pragma unroll
        for (int i = 0; i < MAXQB; ++i) {
            const int qb = wave + 4 * i;
            if (qb >= nqb || (dbg & 64)) break;  // (ablation bit 64, debug builds: loads and staging alone)
            const int sq = qb * 16 + j;

            // S^T tiles: lane holds query j, keys 16t + 4g + r (raw dot products; the 1/8 scale is folded into exp2)
            // VALU diet (PMC: this kernel is VALU-bound, 1088 VALU instructions per 16-query block before): the
            // mask select runs only on tiles that can contain masked/padded keys, scale+subtract is one fma, and
            // exp2 is the bare v_exp_f32 (arguments <= 0, results in [0,1]: no range fix-up needed).
            // The K fragments of tile t+1 are requested before the MFMAs of tile t (no branch inside this loop: with the
            // per-tile `if`s of the first version every tile was its own basic block and each pair of MFMAs waited a
            // full LDS round trip — the ISA showed read, s_waitcnt, mfma, read, s_waitcnt, mfma ... and PMC 49 % of
            // the wave time in waits).
            f4 sc[NT16];
            {
                constexpr int KBATCH = NT16 >= 13 ? ATTN_KBATCH : (NT16 % 4 == 0 ? 4 : NT16);        // tiles whose K fragments are requested together
#pragma unroll
                for (int t0 = 0; t0 < NT16; t0 += KBATCH) {
                    V8 kf[KBATCH][2];
#pragma unroll
                    for (int u = 0; u < KBATCH; ++u)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
                            if (t0 + u < NT16) kf[u][kk] = *(const V8*)(sK + ((t0 + u) * 16 + j) * 128 + (((kk * 4 + g) ^ (j & 7)) << 4));
                    __builtin_amdgcn_sched_barrier(0);      // unfenced, hipcc sinks the reads back to one or two MFMAs before their use
#pragma unroll
                    for (int u = 0; u < KBATCH; ++u) {
                        if (t0 + u >= NT16) continue;
                        f4 acc = {0.f, 0.f, 0.f, 0.f};
                        acc = T::mfma(kf[u][0], qf[i][0], acc);
                        acc = T::mfma(kf[u][1], qf[i][1], acc);
                        sc[t0 + u] = acc;
                    }
                }
            }
            // masks afterwards, only on the tiles that can hold masked / padded keys (wave-uniform conditions)
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                if (MASKALL || t == NT16 - 1) {
                    const f4 kb = *(const f4*)(sKB + t * 16 + g * 4);
#pragma unroll
                    // +inf keeps, MASK_RAW replaces (every real score is above it), -inf removes a pad slot: min(score, limit) as ONE
                    // instruction the compiler can see — med3(score, limit, -inf).  (fminf() adds two canonicalising v_max; rounds 2-4 used an
                    // inline-asm v_min_f32 here, which was only safe because a branch stood between it and the MFMAs: the hazard recognizer
                    // does not look inside inline asm, and with the per-tile branches gone (round 5) the v_min read its MFMA result before
                    // the matrix pipe had written it — 12 of 18 attention cases wrong.)
                    for (int r = 0; r < 4; ++r) sc[t][r] = __builtin_amdgcn_fmed3f(sc[t][r], kb[r], -INFINITY);
                }
            }
            // (round 5: two independent max chains and four independent sum chains instead of one 26-deep v_max3 chain and one 52-deep
            //  v_add chain — the dependent-issue latency of those chains was exposed time, not arithmetic)
            float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                if (t & 1) mx1 = fmaxf(fmaxf(mx1, sc[t][0]), fmaxf(sc[t][1], fmaxf(sc[t][2], sc[t][3])));
                else mx0 = fmaxf(fmaxf(mx0, sc[t][0]), fmaxf(sc[t][1], fmaxf(sc[t][2], sc[t][3])));
            }
            float mx = fmaxf(mx0, mx1);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            f4 sum4 = {0.f, 0.f, 0.f, 0.f};
            const float mxs = -(mx * c2);
#pragma unroll
            for (int t = 0; t < NT16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = fmaf(sc[t][r], c2, mxs);    // one instruction; mxs = -(mx * c2).  At the maximum the result is the
                                                                  // rounding residue of mx*c2 (<= 1e-6 in magnitude: exp2 = 1 +- 7e-7), and it
                                                                  // is exactly 0 on all-masked rows because MASK_RAW is a power of two
                    const float p = __builtin_amdgcn_exp2f(a);          // (a run-time debug select here cost one v_cndmask per score)
                    sc[t][r] = p;
                    sum4[r] += p;
                }
            float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;

            // O^T = V^T · P^T
            f4 o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = (f4){0.f, 0.f, 0.f, 0.f};
            {
                auto vload = [&](int kb, V8 (&vf)[4]) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) vf[dt] = *(const V8*)(sVt + (dt * 16 + j) * VT_LD + kb * 32 + g * 8);
                };
                V8 vc[4];
                vload(0, vc);
                constexpr int NPV = NT16 / 2;                  // 32-key steps of P.V (an odd tile count: + one 16-key step below)
#pragma unroll
                for (int kb = 0; kb < NPV; ++kb) {
                    V8 vn[4];
                    if (kb + 1 < NPV) vload(kb + 1, vn);          // next step's V^T fragments: in flight during the MFMAs
                    V8 pf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pf[e] = T::from_f32(sc[2 * kb][e]);
                        pf[4 + e] = T::from_f32(sc[2 * kb + 1][e]);
                    }
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) o[dt] = T::mfma(vc[dt], pf, o[dt]);
                    if (kb + 1 < NPV) {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) vc[dt] = vn[dt];
                    }
                }
                if constexpr (NT16 % 2 == 1) {                 // keys SP-16 .. SP-1
                    V4 pt;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pt[e] = T::from_f32(sc[NT16 - 1][e]);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        o[dt] = Mfma16k16<T>::run(*(const V4*)(sVt + (dt * 16 + j) * VT_LD + (NT16 / 2) * 32 + g * 8), pt, o[dt]);
                }
            }
            // A lane holds 4 consecutive head dims (8 B) of its query per 16-dim tile; lanes 16 apart (g, g+1) hold the
            // neighbouring 8 B.  `v_permlane16_swap` trades the odd lane's piece of tile 2q for the even lane's piece of
            // tile 2q+1, so every lane owns 16 contiguous bytes and a store instruction writes 16 rows x 64 contiguous
            // bytes (two 16-byte stores per block instead of four 8-byte ones).
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 pk[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                V4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = T::from_f32(o[dt][r] * inv);
                pk[dt] = __builtin_bit_cast(u2, ov);
            }
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(pk[2 * q2][w], pk[2 * q2 + 1][w], false, false);
                    pk[2 * q2][w] = sw[0];
                    pk[2 * q2 + 1][w] = sw[1];
                }
            if (sq < S && !(dbg & 8)) {
                // 32-bit element offset from a wave-uniform base (the launcher checks the tensor is < 2^31 elements): as 64-bit
                // per-lane addresses these were spilled, and every scratch reload is an `s_waitcnt vmcnt(0)` — which also waits
                // for the next head's prefetch loads and for every earlier store
                int g2 = g;
                asm volatile("" : "+v"(g2));              // (recomputed per block for the same reason as in load_head)
                E* op = ctx + (size_t)item * S * D + (unsigned)(sq * D + h * 64 + g2 * 4 + ((g2 & 1) ? 12 : 0));
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
                    *(u4*)(op + q2 * 32) = (u4){pk[2 * q2][0], pk[2 * q2][1], pk[2 * q2 + 1][0], pk[2 * q2 + 1][1]};
            }
        }
    }
}

// ---- CLS-query attention for the LAST executed encoder block --------------------------------------------------------
// The hot path consumes only `hidden_states[i][:, 0]` (Code_Uncached/model/model.py:210-213), so in the last block the
// outputs of every non-CLS token are dead: K and V are still needed for all tokens, but attention, O, the MLP and the
// LayerNorms run for one query per item.  One wave = one (item, head): phase 1 lanes over keys (fp32 dot products,
// softmax statistics by wave shuffles), phase 2 lanes over (4 keys x 16 four-dim groups) streaming V rows.
// HBM-bound: reads K and V once (2 * S * 128 B per pair).  P is rounded to the 16-bit operand type before the PV
// product and the sum uses the unrounded values, as in the MFMA kernel above.
template <typename T>
__global__ __launch_bounds__(256) void attention_cls_kernel(const typename T::elem* __restrict__ qkv,
                                                            const float* __restrict__ key_bias,
                                                            typename T::elem* __restrict__ ctx, int64_t pairs, int S, int heads,
                                                            const typename T::elem* __restrict__ q_cls) {
    typedef typename T::elem E;
    typedef typename T::v8 V8;
    typedef typename T::v4 V4;
    __shared__ float sP[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
    if (pair >= pairs) return;
    const int64_t item = pair / heads;
    const int h = (int)(pair - item * heads);
    const E* qb_ = qkv + pair * 3 * S * 64;
    const E* kb_ = qb_ + (int64_t)S * 64;
    const E* vb_ = kb_ + (int64_t)S * 64;
    // Phase 1: 8 lanes per key (16 bytes of the 128-byte K row each), 8 keys per load instruction — every instruction
    // reads complete lines.  (The first version gave each lane a whole row: 64 lines touched per instruction, 16 bytes of
    // each; with 32 waves per CU doing that the lines fell out of L1 and L2 between a lane's 8 loads — PMC: 2.47 GB
    // fetched per launch for 0.85 GB of K and V.)
    const int kc = lane & 7, kr = lane >> 3;
    float q[8];
    {
        // q_cls: the CLS queries come from their own [items, heads*64] projection (the QKV GEMM wrote K and V only)
        const V8 t = q_cls ? *(const V8*)(q_cls + (item * heads + h) * 64 + kc * 8) : *(const V8*)(qb_ + kc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = T::to_f32(t[e]);
    }
    float mx = -INFINITY;
    // five K rows (and their mask values) per lane in flight: one load per iteration behind an `if` — what the plain loop
    // compiles to — made every one of the 25 iterations wait a whole memory round trip.  Keys past S re-read key S-1.
    constexpr int KB = 5;
    for (int k0 = 0; k0 < S; k0 += 8 * KB) {
        V8 t[KB];
        float kbv[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int key = k0 + 8 * u + kr;
            const int kcl = key < S ? key : S - 1;
            t[u] = *(const V8*)(kb_ + (int64_t)kcl * 64 + kc * 8);
            kbv[u] = key_bias ? key_bias[item * S + kcl] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int key = k0 + 8 * u + kr;
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(q[e], T::to_f32(t[u][e]), a);
            a += __shfl_xor(a, 1, 64);
            a += __shfl_xor(a, 2, 64);
            a += __shfl_xor(a, 4, 64);
            if (key >= S) a = -INFINITY;
            else if (kbv[u] < 0.f) a = MASK_RAW;
            if (kc == 0 && key < 256) sP[wave][key] = a;
            mx = fmaxf(mx, a);
        }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the raw scores are in sP (each wave uses only its own row)
    const float c2 = 0.18033688011112042f;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = lane + 64 * i;
        const float a = key < S ? sP[wave][key] : -INFINITY;
        const float p = __builtin_amdgcn_exp2f((a - mx) * c2);      // -inf -> 0 for the structural pad keys
        sum += p;
        sP[wave][key] = T::to_f32(T::from_f32(p));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): this wave's sP writes (each wave reads only its own row)
    const int kg = lane >> 4, dq = lane & 15;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int VB = 8;                      // V rows per lane in flight (same reason as the K rows above)
    for (int k0 = kg; k0 < S; k0 += 4 * VB) {
        V4 v[VB];
        float pr[VB];
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            const int key = k0 + 4 * u;
            const int kcl = key < S ? key : S - 1;
            v[u] = *(const V4*)(vb_ + (int64_t)kcl * 64 + dq * 4);
            pr[u] = key < S ? sP[wave][kcl] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < VB; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(pr[u], T::to_f32(v[u][e]), acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        acc[e] += __shfl_xor(acc[e], 16, 64);
        acc[e] += __shfl_xor(acc[e], 32, 64);
    }
    if (kg == 0) {
        V4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = T::from_f32(acc[e] * inv);
        *(V4*)(ctx + item * (int64_t)heads * 64 + h * 64 + dq * 4) = o;
    }
}

template <typename T>
int launch_t(const void* qkv, const float* key_bias, void* ctx, int64_t items, int S, int heads, hipStream_t s) {
    typedef typename T::elem E;
    // heads per workgroup: 2 (one head computing, the next one's Q/K/V in flight).  On the spill-free kernel 1 / 2 / 4 / 6
    // heads measured 420 / 421 / 428 / 437 us per ViT layer in isolation and 68.55 / 68.58 / 68.78 ms per step
    int hpw = heads % 2 == 0 ? 2 : 1;
#ifdef ATTN_DEBUG_BITS
    {   // ablation builds (tools/attn_pf_ab.py): bits 8..11 of dev switch attn_debug = heads per workgroup, when it divides the head count
        const int want = (g_attn_dbg >> 8) & 15;
        if (want > 0 && heads % want == 0) hpw = want;
    }
#endif
    dim3 grid((unsigned)(items * (heads / hpw))), block(256);
    // (round 4 measured a three-workgroups-per-CU instantiation without register prefetch, PF = false: 415-421 us against 398-411 us per
    //  ViT layer for this one; it is no longer instantiated — round 5 route retirement)
#define IISAN_ATTN_CASE(NT)                                                                                                       \
    if (key_bias == nullptr && S > 16 * (NT - 1))                                                                                 \
        hipLaunchKernelGGL((attention16_kernel<T, NT, true, false>), grid, block, 0, s, (const E*)qkv, key_bias, (E*)ctx, S, heads, hpw, g_attn_dbg & 127); \
    else                                                                                                                          \
        hipLaunchKernelGGL((attention16_kernel<T, NT, true, true>), grid, block, 0, s, (const E*)qkv, key_bias, (E*)ctx, S, heads, hpw, g_attn_dbg & 127)
    if (S <= 32) { IISAN_ATTN_CASE(2); }
    else if (S <= 64) { IISAN_ATTN_CASE(4); }
    else if (S <= 128) { IISAN_ATTN_CASE(8); }
    else if (S <= 208) { IISAN_ATTN_CASE(13); }   // ViT: 197 tokens = 13 tiles of 16 keys (a 14th would be pure padding)
    else if (S <= 224) { IISAN_ATTN_CASE(14); }
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
    IISAN_CHECK_SHAPE((int64_t)S * heads * 64 < (1ll << 31), "attention16: item of %d x %d elements too large", S, heads * 64);
    return dtype16 == IISAN_BF16 ? launch_t<BF16>(qkv, key_bias, ctx, items, S, heads, s)
                                 : launch_t<F16>(qkv, key_bias, ctx, items, S, heads, s);
}

int launch_attention_cls16(int dtype16, const void* qkv, const float* key_bias, void* ctx_cls, int64_t items, int S, int heads,
                           hipStream_t s, const void* q_cls) {
    IISAN_CHECK_SHAPE(items > 0 && S > 0 && S <= 256 && heads > 0, "attention_cls16: unsupported problem (S=%d)", S);
    const int64_t pairs = items * heads;
    dim3 grid((unsigned)ceil_div(pairs, 4)), block(256);
    if (dtype16 == IISAN_BF16)
        hipLaunchKernelGGL(attention_cls_kernel<BF16>, grid, block, 0, s, (const __bf16*)qkv, key_bias, (__bf16*)ctx_cls, pairs, S, heads, (const __bf16*)q_cls);
    else
        hipLaunchKernelGGL(attention_cls_kernel<F16>, grid, block, 0, s, (const _Float16*)qkv, key_bias, (_Float16*)ctx_cls, pairs, S, heads, (const _Float16*)q_cls);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

extern "C" int iisan_attention16(int32_t dtype16, const void* qkv, const float* key_bias, void* ctx, int64_t items,
                                 int32_t S, int32_t heads, void* stream) {
    return launch_attention16(dtype16, qkv, key_bias, ctx, items, S, heads, (hipStream_t)stream);
}

extern "C" int iisan_attention_cls16(int32_t dtype16, const void* qkv, const float* key_bias, void* ctx_cls, int64_t items,
                                     int32_t S, int32_t heads, void* stream) {
    return launch_attention_cls16(dtype16, qkv, key_bias, ctx_cls, items, S, heads, (hipStream_t)stream, nullptr);
}
